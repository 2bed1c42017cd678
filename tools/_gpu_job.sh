set -o pipefail
mkdir -p gpurun_out
R=$(pwd); OUT=$R/gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/test7.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/test7.log
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > $OUT/ab_sort_$i.json 2> $OUT/ab_sort_$i.err; echo "sort $i rc=$?"
  PATHTRACE_HIP_NO_SORT=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > $OUT/ab_nosort_$i.json 2> $OUT/ab_nosort_$i.err; echo "nosort $i rc=$?"
done
for sc in cornell_box_with_volume textured_room light_test cornell_box_small_lights three_orbs; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs --steps 16 --scene scenes/$sc.json > $OUT/sc_${sc}_sort.json 2>/dev/null
  PATHTRACE_HIP_NO_SORT=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs --steps 16 --scene scenes/$sc.json > $OUT/sc_${sc}_nosort.json 2>/dev/null
done
cd /tmp
B="python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs"
PMC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU"
rm -rf $OUT/pmcX $OUT/pmcY
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmcX -- $B > $OUT/pmcX.log 2>&1 && echo X ok
PATHTRACE_HIP_NO_SORT=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmcY -- $B > $OUT/pmcY.log 2>&1 && echo Y ok
cd $R
python3 tools/pmc_summary.py $OUT/pmc_X.json $OUT/pmcX > /dev/null; python3 tools/pmc_summary.py $OUT/pmc_Y.json $OUT/pmcY > /dev/null
find $OUT/pmcX $OUT/pmcY -name "*.csv" -size +2M -delete
python - <<'PY'
import json,glob
for f in ("ab_sort_1","ab_nosort_1","ab_sort_2","ab_nosort_2"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
for f in sorted(glob.glob("gpurun_out/sc_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["value"], d["config"]["rays_per_sample"])
    except Exception as e: print(f, "ERR", e)
for v in "XY":
    d=json.load(open(f"gpurun_out/pmc_{v}.json"))
    for k in ("extend","connect","shade"):
        c=d[k]; print(v, k, "valu/wave", round(c["SQ_INSTS_VALU"]/c["SQ_WAVES"],1), "salu/wave", round(c["SQ_INSTS_SALU"]/c["SQ_WAVES"],1), "sec", c["seconds"],
                 "lane_util", round(c["SQ_THREAD_CYCLES_VALU"]/c["SQ_ACTIVE_INST_VALU"]/64,3))
PY
