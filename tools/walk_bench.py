#!/usr/bin/env python3
"""Throughput of the traversal choices on generated scenes of many instances (tests/scene_gen.py): the per-lane walk
(default above 48 instances), the fast sweep over the tree program (PATHTRACE_HIP_TRAVERSAL=sweep) and the general sweep
(PATHTRACE_HIP_TRAVERSAL=general).  Prints Mrays/s per scene size and checks that the three images are identical."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

import pathtrace_amd as pt
from scene_gen import random_scene

out = {}
for n_inst in (30, 60, 120, 200, 300):
    js = random_scene(7, n_inst=n_inst, volume=False)
    sc = pt.Scene(text=json.dumps(js), width=1024, height=768)
    imgs = {}
    row = {}
    for name, env in (("walk", {"PATHTRACE_HIP_TRAVERSAL": "walk"}), ("fast_sweep", {"PATHTRACE_HIP_TRAVERSAL": "sweep"}), ("general_sweep", {"PATHTRACE_HIP_TRAVERSAL": "general"})):
        os.environ.pop("PATHTRACE_HIP_TRAVERSAL", None)
        os.environ.update(env)
        r = pt.Renderer(sc, seed=1, max_paths_in_flight=1024 * 768 * 8)
        r.render_async(0, 8); r.wait(); r.clear()
        r.set_profiling(True)
        t0 = time.perf_counter()
        r.render_async(0, 32)
        r.wait()
        dt = time.perf_counter() - t0
        c = r.counters()
        kt = r.kernel_times()
        imgs[name] = r.framebuffer()
        r.close()
        row[name] = {"Mrays/s": round(c["rays"] / dt / 1e6, 1), "extend_ms": round(kt["extend"]["ms"], 1), "connect_ms": round(kt["connect"]["ms"], 1),
                     "shade_ms": round(kt["shade"]["ms"], 1)}
    row["identical"] = bool(all(np.array_equal(imgs["walk"].view(np.uint32), v.view(np.uint32)) for v in imgs.values()))
    out[f"{sc.desc.n_instances} instances"] = row
    print(n_inst, row, file=sys.stderr, flush=True)
print(json.dumps(out))
