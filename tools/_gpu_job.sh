set -o pipefail
mkdir -p gpurun_out
R=$(pwd); OUT=$R/gpurun_out; export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > $OUT/knob_$name.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/knob_$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], d["ms_per_step"])
PY
}
L=$R/pathtrace_amd/lib
run base A=1
run s5 PATHTRACE_HIP_LIB=$L/libpathtrace_hip_s5.so
run s7 PATHTRACE_HIP_LIB=$L/libpathtrace_hip_s7.so
run c5 PATHTRACE_HIP_LIB=$L/libpathtrace_hip_c5.so
run c7 PATHTRACE_HIP_LIB=$L/libpathtrace_hip_c7.so
run base2 A=1
