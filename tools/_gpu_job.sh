set -o pipefail
mkdir -p gpurun_out
R=$(pwd); OUT=$R/gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/test6.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/test6.log
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > $OUT/ab_pk_$i.json 2> $OUT/ab_pk_$i.err; echo "pk $i rc=$?"
  PATHTRACE_HIP_LIB=$R/pathtrace_amd/lib/libpathtrace_hip_nopk.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > $OUT/ab_nopk_$i.json 2> $OUT/ab_nopk_$i.err; echo "nopk $i rc=$?"
done
cd /tmp
rm -rf $OUT/pmcT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d $OUT/pmcT -- python3 $R/tools/trace_bench.py > $OUT/pmcT.log 2>&1; echo rc=$?
cd $R
python3 tools/trace_bench.py --summarize $OUT/pmcT
python - <<'PY'
import json
for f in ("ab_pk_1","ab_nopk_1","ab_pk_2","ab_nopk_2"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
