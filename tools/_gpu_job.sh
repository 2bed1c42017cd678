#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1 || { tail -30 gpurun_out/final_tests.log; exit 1; }
tail -2 gpurun_out/final_tests.log
timeout -k 10 200 python __graft_entry__.py --smoke > gpurun_out/final_smoke.log 2>&1 || { tail gpurun_out/final_smoke.log; exit 1; }
tail -1 gpurun_out/final_smoke.log
timeout -k 10 400 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err || { tail gpurun_out/bench_full.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_full.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["cpu_baseline"])
for c in d["configs"]: print(c["config"], c["value"], c["dominant_kernel"])
PY
