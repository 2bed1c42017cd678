#!/usr/bin/env python3
"""Does a batch's pixel count (its alignment to the queue segments) change the kernel times?  Renders the centre WxH window
of cornell_box 1080p for a few (W, H, spp per batch) and prints per-kernel times (one lane, HIP events) per Mray."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtrace_amd as pt

W, H = 1920, 1080
scene = pt.Scene(os.path.join(ROOT, "scenes", "cornell_box.json"), W, H)
cases = [(512, 512, 54), (512, 500, 54), (512, 512, 53), (520, 512, 54), (1024, 256, 54), (512, 512, 27), (1024, 1024, 16), (1024, 1000, 16)]
for (w, h, spp) in cases:
    x0, y0 = (W - w) // 2, (H - h) // 2
    r = pt.Renderer(scene, max_paths_in_flight=w * h * spp)
    r.render_async(0, spp, (x0, y0, x0 + w, y0 + h))
    r.wait()
    r.clear()
    r.set_lanes(1)
    r.set_profiling(True)
    t0 = time.perf_counter()
    for i in range(4):
        r.render_async(i * spp, (i + 1) * spp, (x0, y0, x0 + w, y0 + h))
    r.wait()
    dt = time.perf_counter() - t0
    kt = r.kernel_times()
    c = r.counters()
    mr = c["rays_traced"] / 1e6
    print(json.dumps({"w": w, "h": h, "spp": spp, "npix": w * h, "Mrays": round(mr, 1), "wall_ms": round(dt * 1e3, 2), "Grays_per_s": round(mr / dt / 1e3, 2),
                      "us_per_Mray": {k: round(v["ms"] * 1e3 / mr, 2) for k, v in kt.items()}}), flush=True)
    r.close()
