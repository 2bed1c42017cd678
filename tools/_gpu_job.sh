set -o pipefail
mkdir -p gpurun_out
R=$(pwd); OUT=$R/gpurun_out; export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs"
PMC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA"
rm -rf $OUT/pmcX
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmcX -- $B > $OUT/pmcX.log 2>&1 && echo X ok
cd $R
python3 tools/pmc_summary.py $OUT/pmc_X.json $OUT/pmcX > /dev/null
find $OUT/pmcX -name "*.csv" -size +2M -delete
python - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_X.json"))
for k in ("extend","connect","shade"):
    c=d[k]; print(k, "valu/wave", round(c["SQ_INSTS_VALU"]/c["SQ_WAVES"],1), "salu/wave", round(c["SQ_INSTS_SALU"]/c["SQ_WAVES"],1), "sec", c["seconds"],
                 "wait_any", round(c["SQ_WAIT_ANY"]/c["SQ_WAVE_CYCLES"],3), "wait_inst", round(c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"],3), "active_valu/insts", round(c["SQ_ACTIVE_INST_VALU"]/c["SQ_INSTS_VALU"],3))
PY
PATHTRACE_HIP_LANES=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > $OUT/l1.json 2>/dev/null
PATHTRACE_HIP_LANES=3 timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > $OUT/l3.json 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > $OUT/l2.json 2>/dev/null
python - <<'PY'
import json
for f in ("l1","l2","l3"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
