set -o pipefail
mkdir -p gpurun_out
R=$(pwd); OUT=$R/gpurun_out; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/test9.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/test9.log
