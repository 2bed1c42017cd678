#!/bin/bash
# north_star's "per-bounce material-sorted shading queues" (VERDICT N2): what the sort inside a 256-path chunk recovers.
# For three divergent scenes and the headline one, k_shade's VALU lane utilisation (SQ_THREAD_CYCLES_VALU / (64 x
# SQ_ACTIVE_INST_VALU): the share of lanes doing work in the vector instructions it issues) and its time, unsorted / sorted, one
# lane (every launch alone on the chip), plus the hit / miss / emitter mix that bounds what class-pure waves could reach.
#   tools/n2_probe.sh TAG      -> gpurun_out/TAG_n2_<scene>_<sort>.json (tools/pmc_summary.py), gpurun_out/TAG_n2_mix.json
set -o pipefail
TAG=${1:-r04}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for scene in cornell_box three_orbs textured_room light_test; do
  for sort in nosort sort; do
    d=$OUT/${TAG}_n2_${scene}_${sort}
    rm -rf "$d"
    export PATHTRACE_HIP_LANES=1 PT_BENCH_MAX_PATHS=66355200
    export PATHTRACE_HIP_SHADE=${sort}
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --output-format csv -d "$d" -- \
      python3 "$R/bench.py" --scene "$R/scenes/$scene.json" --steps 4 --warmup 1 --no-cpu-baseline --no-configs --no-scaling-proxy --no-plugin-path > "$d.log" 2>&1 || { echo "$scene $sort failed"; tail -3 "$d.log"; exit 1; }
    python3 "$R/tools/pmc_summary.py" "$d.json" "$d" > /dev/null
    python3 - "$d.json" "$scene" "$sort" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["shade"]
print(sys.argv[2], sys.argv[3], "k_shade lane utilisation %.4f" % (d["SQ_THREAD_CYCLES_VALU"] / (64.0 * d["SQ_ACTIVE_INST_VALU"])), "ms %.3f" % (d["seconds"] * 1e3),
      "VALU wave-instructions %.3e" % d["SQ_INSTS_VALU"], flush=True)
PY
    rm -rf "$d"
  done
done
