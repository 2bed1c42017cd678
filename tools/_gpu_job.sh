#!/bin/bash
set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_traversal.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for i in 1 2 3; do
PATHTRACE_HIP_LIB=$R/pathtrace_amd/lib/libpathtrace_hip_old.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > gpurun_out/old_$i.json 2> gpurun_out/spec.err || { tail gpurun_out/spec.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > gpurun_out/new_$i.json 2> gpurun_out/spec.err || { tail gpurun_out/spec.err; exit 1; }
done
python - <<'PY'
import json
for f in ("old_1","new_1","old_2","new_2","old_3","new_3"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["config"]["framebuffer_sum"], d["config"]["rays"], d["roofline"]["kernel_ms"]["extend"])
PY
