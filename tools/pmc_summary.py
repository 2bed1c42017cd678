#!/usr/bin/env python3
"""Sum the counters of rocprofv3 --pmc passes per kernel.

    python tools/pmc_summary.py <out.json> <dir1> [<dir2> ...]

Every <dir> holds one pass's counter_collection.csv.  Output: {kernel: {counter: sum over its dispatches, ...,
"dispatches": n, "seconds": summed dispatch time of the first pass that saw the kernel}}.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNELS = ("generate", "extend", "shade", "connect", "accumulate")


def kname(full):
    for k in KERNELS:
        if "k_" + k in full:
            return k
    return None


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    res = defaultdict(dict)
    for d in dirs:
        hits = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
        if not hits:
            continue
        acc = defaultdict(lambda: defaultdict(float))
        dur = defaultdict(dict)
        with open(hits[0]) as fh:
            for row in csv.DictReader(fh):
                k = kname(row.get("Kernel_Name", ""))
                if not k:
                    continue
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                dur[k][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
        for k, c in acc.items():
            res[k].update(c)
            res[k].setdefault("dispatches", len(dur[k]))
            res[k].setdefault("seconds", round(sum(dur[k].values()), 6))
    json.dump(res, open(out, "w"), indent=1)
    for k, c in res.items():
        print(k, json.dumps(c))


if __name__ == "__main__":
    main()
