#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1 || { tail -30 gpurun_out/final_tests.log; exit 1; }
tail -2 gpurun_out/final_tests.log
bash tools/profile_gpu.sh r02d
