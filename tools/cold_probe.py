#!/usr/bin/env python3
"""How long a fresh process's first render takes against its later ones (the drop-in surface starts cold: pth_main renders once).

    python tools/cold_probe.py [spp] [prime]     prime = none | calls:N (N one-sample calls over the frame before the first render) | ms:T
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtrace_amd as pt

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prime = sys.argv[2] if len(sys.argv) > 2 else "none"
W, H = 1920, 1080
t0 = time.perf_counter()
sc = pt.Scene(os.path.join(ROOT, "scenes", "cornell_box.json"), W, H)
r = pt.Renderer(sc, seed=0)
r.reserve(W * H, spp)
t1 = time.perf_counter()
r.spec_wait()
t2 = time.perf_counter()
if prime.startswith("calls:"):
    for k in range(int(prime.split(":")[1])):
        r.render_async(0, 1)
    r.wait()
    r.clear()
elif prime.startswith("ms:"):
    end = time.perf_counter() + float(prime.split(":")[1]) * 1e-3
    while time.perf_counter() < end:
        r.render_async(0, 3)
        r.wait()
    r.clear()
t3 = time.perf_counter()
secs = []
for k in range(5):
    r.clear()
    r.render_async(0, spp)
    r.wait()
    secs.append(round(r.render_seconds() * 1e3, 3))
c = r.counters()
print(json.dumps({"spp": spp, "prime": prime, "create_reserve_ms": round((t1 - t0) * 1e3, 1), "spec_wait_ms": round((t2 - t1) * 1e3, 1), "prime_ms": round((t3 - t2) * 1e3, 1),
                  "render_ms": secs, "first_over_best": round(secs[0] / min(secs), 4), "plan": r.plan(), "Grays_traced_best": round(c["rays_traced"] / min(secs) / 1e6, 2)}))
