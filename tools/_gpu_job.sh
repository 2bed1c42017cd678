#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for g in 1536 3072 4096 4608 6144 9216; do
for l in 3; do
PATHTRACE_HIP_GRID=$g PATHTRACE_HIP_LANES=$l timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > gpurun_out/g_${g}_$l.json 2> gpurun_out/g.err || { tail gpurun_out/g.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/g_${g}_$l.json").read().strip().splitlines()[-1])
print("grid $g lanes $l", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
done; done
