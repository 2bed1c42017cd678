set -o pipefail
mkdir -p gpurun_out
R=$(pwd); OUT=$R/gpurun_out; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/test10.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/test10.log
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > $OUT/ab_flat_$i.json 2> $OUT/ab_flat_$i.err; echo "flat $i rc=$?"
  PATHTRACE_HIP_NO_FLAT=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > $OUT/ab_noflat_$i.json 2> $OUT/ab_noflat_$i.err; echo "noflat $i rc=$?"
done
PATHTRACE_HIP_LANES=3 timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > $OUT/ab_flat_l3.json 2>/dev/null
cd /tmp
B="python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs"
PMC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU"
rm -rf $OUT/pmcX
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmcX -- $B > $OUT/pmcX.log 2>&1 && echo X ok
cd $R
python3 tools/pmc_summary.py $OUT/pmc_X.json $OUT/pmcX > /dev/null
find $OUT/pmcX -name "*.csv" -size +2M -delete
python - <<'PY'
import json
for f in ("ab_flat_1","ab_noflat_1","ab_flat_2","ab_noflat_2","ab_flat_l3"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
d=json.load(open("gpurun_out/pmc_X.json"))
for k in ("extend","connect","shade"):
    c=d[k]; print(k, "valu/wave", round(c["SQ_INSTS_VALU"]/c["SQ_WAVES"],1), "salu/wave", round(c["SQ_INSTS_SALU"]/c["SQ_WAVES"],1), "sec", c["seconds"])
PY
