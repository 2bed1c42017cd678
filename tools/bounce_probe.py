#!/usr/bin/env python3
"""Time of every launch of a batch, per kernel and bounce: PATHTRACE_HIP_TRACE_LAUNCH=1 puts one stream synchronisation and one
stderr line after each launch (about 30 us of every figure is that synchronisation).

    python tools/bounce_probe.py [spp_per_batch] [batches] [scene.json]  -> one JSON object (median us per launch, n_seg, share of the batch)
"""
import json
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 1920, 1080

if os.environ.get("PT_BOUNCE_PROBE_CHILD"):
    sys.path.insert(0, ROOT)
    import pathtrace_amd as pt
    spp, batches, scene_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    scene = pt.Scene(scene_file, W, H)
    r = pt.Renderer(scene, max_paths_in_flight=W * H * spp)
    r.set_lanes(1)
    r.spec_wait()
    s = 0
    for _ in range(batches):
        r.render_async(s, s + spp)
        s += spp
    r.wait()
    print(json.dumps({"module": r.spec_status() == 1, "rays": r.counters()["rays"] if hasattr(r, "counters") else None}))
    r.close()
    sys.exit(0)

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 96
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
scene_file = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "scenes", "cornell_box.json")
env = dict(os.environ, PT_BOUNCE_PROBE_CHILD="1", PATHTRACE_HIP_TRACE_LAUNCH="1")
p = subprocess.run([sys.executable, os.path.abspath(__file__), str(spp), str(batches), scene_file], env=env, capture_output=True, text=True)
if p.returncode != 0:
    sys.stderr.write(p.stderr[-2000:])
    sys.exit(1)
pat = re.compile(r"\[pt launch\] lane \d+ (\w+) bounce (-?\d+): n_seg (\d+) x (\d+) -> .*?([0-9.]+) us since the previous line")
rows = {}
for line in p.stderr.splitlines():
    m = pat.search(line)
    if m:
        rows.setdefault((m.group(1), int(m.group(2))), []).append((float(m.group(5)), int(m.group(3))))
out = {"what": f"PATHTRACE_HIP_TRACE_LAUNCH=1, {os.path.basename(scene_file)} {W}x{H}, one lane, {batches} batches of {spp} spp "
               f"({W * H * spp / 1e6:.0f} M paths); median us per launch over the batches after the first (about 30 us of each is the synchronisation)",
       "child": json.loads(p.stdout.strip().splitlines()[-1]), "launches": {}}
total = 0.0
for (k, b), v in sorted(rows.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    us = statistics.median([x[0] for x in v[1:]] or [v[0][0]])
    out["launches"][f"{k} bounce {b}"] = {"median_us": round(us, 1), "n_seg": v[-1][1]}
    total += us
out["batch_us"] = round(total, 1)
late = sum(d["median_us"] for n, d in out["launches"].items() if int(n.split()[-1]) >= 5)
out["bounces_5_up_us"] = round(late, 1)
out["bounces_5_up_share"] = round(late / total, 4)
out["bounces_5_up_launches"] = sum(1 for n in out["launches"] if int(n.split()[-1]) >= 5)
print(json.dumps(out))
