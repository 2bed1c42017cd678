set -o pipefail
bash tools/profile_gpu.sh r02a 2>&1 | tail -40
