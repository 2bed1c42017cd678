#!/usr/bin/env python3
"""One-GPU probe of the launch plan of a rank of an N-rank strong-scaling run: the same tile lists and K steps rendered with
different numbers of equal batches (at most 66 M paths each), three lanes in flight.  N = 1 is the whole frame.

    python tools/plan_probe.py [N] [steps]      -> one JSON line per batch count: slowest rank ms, N x slowest
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PATHTRACE_HIP_PLAN"] = "caller"   # this probe times plans OTHER than the library's: every call below is one batch
import pathtrace_amd as pt
from pathtrace_amd.distributed import measure_tile_costs, tiles_for_rank

W, H, TILE = 1920, 1080, 128
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
scene = pt.Scene(os.path.join(ROOT, "scenes", "cornell_box.json"), W, H)
spiral = pt.spiral_tiles(W, H, TILE, TILE)
planner = pt.Renderer(scene, max_paths_in_flight=W * H)
costs = measure_tile_costs(planner, spiral)
planner.close()
total = 16 * steps
lists = [tiles_for_rank(W, H, TILE, TILE, q, n, costs, "lpt") for q in range(n)]
pix = [sum((x1 - x0) * (y1 - y0) for (x0, y0, x1, y1) in tl) for tl in lists]


def cut(weights):
    """total spp cut in proportion to the weights (largest remainder), zero-size batches dropped"""
    s = float(sum(weights))
    raw = [total * w / s for w in weights]
    sizes = [int(x) for x in raw]
    order = sorted(range(len(raw)), key=lambda i: raw[i] - sizes[i], reverse=True)
    for i in order[: total - sum(sizes)]:
        sizes[i] += 1
    return [x for x in sizes if x > 0]


CAP = int(os.environ.get("PT_PLAN_CAP", "66400000"))   # path slots per batch (the library's PT_PLAN_MAX_PATHS is 199 M)
COUNTS = [int(x) for x in os.environ.get("PT_PLAN_COUNTS", "3,4,5,6,9,12,15,18,24,33").split(",")]
WEIGHTS = [[float(x) for x in w.split(":")] for w in os.environ.get("PT_PLAN_WEIGHTS", "").split(",") if w]   # e.g. "5:3,3:5,2:1:1": unequal batches
for plan in COUNTS + WEIGHTS:
    count = plan if isinstance(plan, int) else len(plan)
    sizes = cut([1] * count if isinstance(plan, int) else plan)
    if max(pix) * max(sizes) > CAP or len(sizes) < count:
        continue
    r = pt.Renderer(scene, max_paths_in_flight=max(pix) * max(sizes))
    r.spec_wait()

    def run(tl):
        s = 0
        for k in sizes:
            r.render_tiles_async(tl, s, s + k)
            s += k
        r.wait()

    run(lists[0])
    times = []
    for rep in range(2):
        tt = []
        for q, tl in enumerate(lists):
            r.clear()
            t0 = time.perf_counter()
            run(tl)
            tt.append(time.perf_counter() - t0)
        times.append(tt)
    r.close()
    best = [min(a, b) for a, b in zip(*times)]
    print(json.dumps({"n": n, "steps": steps, "batches": count, "spp": sizes[0], "sizes": sizes, "paths_M": round(max(pix) * sizes[0] / 1e6, 1),
                      "max_ms": round(max(best) * 1e3, 3), "sum_ms": round(sum(best) * 1e3, 3), "n_x_max_ms": round(n * max(best) * 1e3, 3)}), flush=True)
