#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1 || { tail -30 gpurun_out/final_tests.log; exit 1; }
tail -3 gpurun_out/final_tests.log
timeout -k 10 200 python __graft_entry__.py --smoke > gpurun_out/final_smoke.log 2>&1 || { tail gpurun_out/final_smoke.log; exit 1; }
tail -1 gpurun_out/final_smoke.log
bash tools/profile_gpu.sh r02b
