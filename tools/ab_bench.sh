#!/bin/bash
# Same-box A/B of library variants (run through gpurun from the repo root):
#   tools/ab_bench.sh TAG REPS STEPS VARIANT [VARIANT ...]
# VARIANT = "main" (pathtrace_amd/lib/libpathtrace_hip.so) or NAME for lib/libpathtrace_hip_NAME.so (python -m pathtrace_amd.build
# --variant NAME -D...).  An optional ":ENV=VAL,ENV2=VAL2" suffix sets environment knobs for that variant.  The variants
# alternate REPS times; every run is the default bench line without the side blocks -> gpurun_out/TAG_<variant>_<i>.json
set -o pipefail
TAG=$1; REPS=$2; STEPS=$3; shift 3
R=$(pwd); OUT=$R/gpurun_out; mkdir -p "$OUT"
for i in $(seq 1 "$REPS"); do
  for spec in "$@"; do
    v=${spec%%:*}; envs=""
    if [[ "$spec" == *:* ]]; then envs=${spec#*:}; fi
    label=$(echo "$spec" | tr ':=,/ ' '_____')
    (
      if [ "$v" != "main" ]; then export PATHTRACE_HIP_LIB=$R/pathtrace_amd/lib/libpathtrace_hip_$v.so; fi
      if [ -n "$envs" ]; then IFS=',' read -ra kv <<< "$envs"; for e in "${kv[@]}"; do export "$e"; done; fi
      SC=""; if [ -n "$AB_SCENE" ]; then SC="--scene $R/scenes/$AB_SCENE"; fi   # AB_SCENE=cornell_box_with_volume.json: another scene
      timeout -k 10 300 python3 "$R/bench.py" $SC --steps "$STEPS" --warmup 2 --no-cpu-baseline --no-configs --no-scaling-proxy --no-plugin-path > "$OUT/${TAG}_${label}_$i.json" 2> "$OUT/${TAG}_${label}_$i.err"
    ) || { echo "run $label $i failed"; tail -3 "$OUT/${TAG}_${label}_$i.err"; exit 1; }
    python3 - "$OUT/${TAG}_${label}_$i.json" "$label" "$i" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["roofline"]["serialised"]["kernels"]
print(sys.argv[2], sys.argv[3], "traced Mrays/s", d["value"], "ms/step", d["ms_per_step"], "one-lane ms:", {n: k[n]["ms"] for n in ("extend", "shade", "connect")}, flush=True)
PY
  done
done
