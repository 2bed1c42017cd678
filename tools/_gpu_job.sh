set -o pipefail
mkdir -p gpurun_out
R=$(pwd); OUT=$R/gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > $OUT/test12.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/test12.log
run() { name=$1; scene=$2; shift; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs --steps 32 --scene scenes/$scene.json > $OUT/knob_$name.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/knob_$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], d["ms_per_step"])
PY
}
L=$R/pathtrace_amd/lib
run vol_base cornell_box_with_volume A=1
run vol_ga5 cornell_box_with_volume PATHTRACE_HIP_LIB=$L/libpathtrace_hip_ga5.so
run vol_base2 cornell_box_with_volume A=1
run vol_ga5b cornell_box_with_volume PATHTRACE_HIP_LIB=$L/libpathtrace_hip_ga5.so
run small cornell_box_small_lights A=1
run small2 cornell_box_small_lights A=1
run orbs three_orbs A=1
run ltest light_test A=1
run tex textured_room A=1
