#!/bin/bash
# Regenerates the measurement evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   tools/profile_gpu.sh r03a [scene.json]
#   1. default bench.py line (headline scene only)              -> gpurun_out/<tag>_bench_default.json
#   2. rocprofv3 --kernel-trace --stats of the same command    -> gpurun_out/<tag>_kernel_stats.csv
#   2b. the same with one batch in flight (PATHTRACE_HIP_LANES=1) -> gpurun_out/<tag>_kernel_stats_one_lane.csv
#   3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), one lane -> gpurun_out/<tag>_pmc_{fetch,write}/
#   4. one --pmc pass of SQ instruction counters               -> gpurun_out/<tag>_pmc_insts/
#   5. one --pmc pass of SQ wait-state counters, one lane      -> gpurun_out/<tag>_pmc_wait/
# tools/summarize_profiles.py then writes the summaries (it runs here too, so the merged gpurun_out/ already holds them;
# copy them into profiles/ and commit).  Every summary carries the hash of the kernel sources it was taken on.
set -o pipefail
TAG=${1:-r01x}
SCENE=${2:-}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
SC=""
if [ -n "$SCENE" ]; then SC="--scene $R/scenes/$SCENE --no-configs --no-scaling-proxy"; fi
timeout -k 10 800 python3 "$R/bench.py" $SC > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench.err" || { echo "bench failed"; tail -5 "$OUT/${TAG}_bench.err"; exit 1; }
cd /tmp
Q="--no-cpu-baseline --no-configs --no-scaling-proxy --no-plugin-path"
if [ -n "$SCENE" ]; then Q="$Q --scene $R/scenes/$SCENE"; fi
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_prof" -- python3 "$R/bench.py" $Q > "$OUT/${TAG}_prof.log" 2>&1 || { echo "kernel-trace failed"; tail -5 "$OUT/${TAG}_prof.log"; exit 1; }
export PATHTRACE_HIP_LANES=1
export PT_BENCH_MAX_PATHS=66355200   # 1920 x 1080 x 32 slots: 32 spp per launch, comparable from round to round
# 2b. the same kernel-trace statistics with ONE batch in flight: every launch alone on the chip, so the table's average duration of a
#     kernel is the one bench.py's roofline block uses (its own one-lane pass)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_prof1" -- python3 "$R/bench.py" --steps 8 --warmup 1 $Q > "$OUT/${TAG}_prof1.log" 2>&1 || { echo "one-lane kernel-trace failed"; tail -5 "$OUT/${TAG}_prof1.log"; exit 1; }
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_pmc_fetch" -- python3 "$R/bench.py" --steps 4 --warmup 1 $Q > "$OUT/${TAG}_pmc_fetch.log" 2>&1 || { echo "pmc fetch failed"; tail -5 "$OUT/${TAG}_pmc_fetch.log"; exit 1; }
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_pmc_write" -- python3 "$R/bench.py" --steps 4 --warmup 1 $Q > "$OUT/${TAG}_pmc_write.log" 2>&1 || { echo "pmc write failed"; tail -5 "$OUT/${TAG}_pmc_write.log"; exit 1; }
# 4. instruction counters (own pass, serialised dispatches)
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${TAG}_pmc_insts" -- python3 "$R/bench.py" --steps 4 --warmup 1 $Q > "$OUT/${TAG}_pmc_insts.log" 2>&1 || { echo "pmc insts failed"; tail -5 "$OUT/${TAG}_pmc_insts.log"; exit 1; }
# 5. where a wave's life goes, every kernel alone on the chip (one lane): parked on s_waitcnt / barriers, stalled at issue, executing
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${TAG}_pmc_wait" -- python3 "$R/bench.py" --steps 4 --warmup 1 $Q > "$OUT/${TAG}_pmc_wait.log" 2>&1 || { echo "pmc wait failed"; tail -5 "$OUT/${TAG}_pmc_wait.log"; exit 1; }
unset PATHTRACE_HIP_LANES PT_BENCH_MAX_PATHS
cd "$R"
python3 tools/summarize_profiles.py "$TAG" "$OUT"
