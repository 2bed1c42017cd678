set -o pipefail
mkdir -p gpurun_out
R=$(pwd); OUT=$R/gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_traversal.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q > $OUT/test14.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/test14.log
PATHTRACE_HIP_LANES=1 timeout -k 10 900 python tools/walk_bench.py > $OUT/walk_bench.json 2> $OUT/walk_bench.err; echo "walk rc=$?"; cat $OUT/walk_bench.err | tail -6
for i in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > $OUT/b15_$i.json 2>/dev/null; done; python - <<'PY'
import json
for i in (1,2):
    d=json.loads(open(f"gpurun_out/b15_{i}.json").read().strip().splitlines()[-1]); print("bench", d["value"], d["ms_per_step"])
PY
