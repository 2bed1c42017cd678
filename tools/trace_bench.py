#!/usr/bin/env python3
"""Micro-benchmark of World::hit through pt_trace_rays (k_trace): N interior rays of a scene, fast sweep then general
sweep (PATHTRACE_HIP_TRAVERSAL=general), for rays_per_origin 1 and 2.  Run under `rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU
SQ_INSTS_SALU` and read the k_trace dispatches in order with tools/trace_bench.py --summarize <dir>: instructions per
wave = per ray (one origin per lane)."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarize(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[0]
    per = {}
    for row in csv.DictReader(open(f)):
        if "k_trace" not in row["Kernel_Name"]:
            continue
        k = int(row["Dispatch_Id"])
        per.setdefault(k, {"name": row["Kernel_Name"].split("(")[0][-24:], "dur_us": (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3})
        per[k][row["Counter_Name"]] = per[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    for k in sorted(per):
        c = per[k]
        w = max(c.get("SQ_WAVES", 1.0), 1.0)
        print(k, c["name"], "waves", int(w), "valu/wave", round(c.get("SQ_INSTS_VALU", 0) / w, 1), "salu/wave", round(c.get("SQ_INSTS_SALU", 0) / w, 1),
              "us", round(c["dur_us"], 1))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        return summarize(sys.argv[2])
    import numpy as np
    import pathtrace_amd as pt
    scene = sys.argv[1] if len(sys.argv) > 1 else "cornell_box.json"
    n = 1 << 21
    rng = np.random.default_rng(5)
    sc = pt.Scene(os.path.join(ROOT, "scenes", scene), 64, 64)
    bb = sc.nodes()[0][0].astype(np.float64)
    o = (bb[:3] + 1.0 + rng.random((n, 3)) * (bb[3:] - bb[:3] - 2.0)).astype(np.float32)
    d1 = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d2 = rng.normal(0, 1, (n, 2, 3)).astype(np.float32)
    for env in (None, "1"):
        if env:
            os.environ["PATHTRACE_HIP_TRAVERSAL"] = "general"
        else:
            os.environ.pop("PATHTRACE_HIP_TRAVERSAL", None)
        r = pt.Renderer(sc, max_paths_in_flight=4096)
        t1, i1 = r.trace_rays(o, d1)
        t2, i2 = r.trace_rays(o, d2)
        print("fast" if not env else "general", "hits", int((i1 >= 0).sum()), int((i2 >= 0).sum()))
        r.close()


if __name__ == "__main__":
    main()
