set -o pipefail
mkdir -p gpurun_out
R=$(pwd); OUT=$R/gpurun_out; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/test11.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/test11.log
timeout -k 10 400 python bench.py > $OUT/bench_r02b.json 2> $OUT/bench_r02b.err; echo "bench rc=$?"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_r02b_k20.json 2> /dev/null; echo "bench k20 rc=$?"
PATHTRACE_HIP_CONNECT_NR=4 timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > $OUT/bench_nr4.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_r02b","bench_r02b_k20","bench_nr4"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d.get("parity"))
    for c in (d.get("configs") or []): print("   ", c.get("config"), c.get("value"), c.get("error"))
PY
