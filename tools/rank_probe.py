#!/usr/bin/env python3
"""One-GPU probe of the strong-scaling partition: per-rank kernel times (one lane, HIP events) and wall times (three lanes)
for the tile lists of an N-rank run, to see which kernel makes one rank's list slower than another's.

    python tools/rank_probe.py [N] [steps]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtrace_amd as pt
from pathtrace_amd.distributed import measure_tile_costs, tiles_for_rank

W, H, TILE = 1920, 1080, 128
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
strategy = os.environ.get("PT_BENCH_PARTITION", "lpt")
scene = pt.Scene(os.path.join(ROOT, "scenes", "cornell_box.json"), W, H)
spiral = pt.spiral_tiles(W, H, TILE, TILE)
planner = pt.Renderer(scene, max_paths_in_flight=W * H)
costs = measure_tile_costs(planner, spiral)
planner.close()
total_spp = 16 * steps
spp_launch = -(-total_spp // 6)
lists = [tiles_for_rank(W, H, TILE, TILE, q, n, costs, strategy) for q in range(n)]
extra = os.environ.get("PT_PROBE_EXTRA")
if extra == "noclip":      # every list without its clipped (top row) tiles
    lists = [[t for t in tl if t[3] - t[1] == TILE] for tl in lists]
pix = [sum((x1 - x0) * (y1 - y0) for (x0, y0, x1, y1) in tl) for tl in lists]
r = pt.Renderer(scene, max_paths_in_flight=max(pix) * spp_launch)


def run(tl):
    s = 0
    while s < total_spp:
        e = min(s + spp_launch, total_spp)
        r.render_tiles_async(tl, s, e)
        s = e
    r.wait()


run(lists[0])
out = []
for q, tl in enumerate(lists):
    r.clear()
    r.set_lanes(3)
    t0 = time.perf_counter()
    run(tl)
    wall3 = time.perf_counter() - t0
    r.clear()
    r.set_lanes(1)
    r.set_profiling(True)
    t0 = time.perf_counter()
    run(tl)
    wall1 = time.perf_counter() - t0
    kt = r.kernel_times()
    r.set_profiling(False)
    c = r.counters()
    out.append({"rank": q, "tiles": len(tl), "pixels": pix[q], "clipped": sum(1 for t in tl if t[3] - t[1] != TILE or t[2] - t[0] != TILE),
                "wall3_ms": round(wall3 * 1e3, 3), "wall1_ms": round(wall1 * 1e3, 3), "rays_traced": c["rays_traced"],
                "kernel_ms": {k: round(v["ms"], 3) for k, v in kt.items()}})
    print(json.dumps(out[-1]), flush=True)
