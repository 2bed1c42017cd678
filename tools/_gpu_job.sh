set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python tools/fuzz_gpu.py 1000 240 > gpurun_out/fuzz_r02.json 2> gpurun_out/fuzz_r02.err; echo rc=$?
cat gpurun_out/fuzz_r02.json; grep -c ok gpurun_out/fuzz_r02.err; grep MISMATCH gpurun_out/fuzz_r02.err | head
