#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (GPU sanitizers are not available on this pool):
#   1. the C++ host front end (JSON, scene builder, BVH build, PNG reader) on every scene, 80 generated scenes and
#      malformed inputs                                           -> load_scenes
#   2. the PNG reader on 7500 mutated copies of assets/*.png      -> png_fuzz
#   3. the oracle (CPU restatement) through its own test file     -> pytest with a sanitized libpt_oracle.so
#   4. the per-scene build's compile helper (pt_spec_cc) and its file protocol -> rtc_roundtrip
# The device entry points the front end refers to are stubbed.  Run from the repo root: bash tools/sanitize/run.sh
set -e
R=$(pwd)
W=$(mktemp -d)
FLAGS="-std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I$R/include -I$R/tools/sanitize"
g++ $FLAGS tools/sanitize/load_scenes.cpp pathtrace_amd/csrc/host/pt_host.cpp -o $W/load_scenes
g++ $FLAGS tools/sanitize/png_fuzz.cpp pathtrace_amd/csrc/host/pt_host.cpp -o $W/png_fuzz
python3 - "$W" <<'PY'
import json, os, sys
sys.path.insert(0, "tests")
from scene_gen import random_scene
d = os.path.join(sys.argv[1], "rs", "scenes")
os.makedirs(d)
for seed in range(80):
    json.dump(random_scene(seed), open(os.path.join(d, f"r{seed}.json"), "w"))
open(os.path.join(d, "bad1.json"), "w").write('{"camera": {"look_from": [0,0,0], "look_at": [0,0,1]}, "instances": []}')
open(os.path.join(d, "bad2.json"), "w").write("{ not json")
open(os.path.join(d, "bad3.json"), "w").write("[1,2,3]")
open(os.path.join(d, "bad4.json"), "w").write('{"camera": {}, "materials": [{"id":"a"}], "primitives": [{"type":"rect"}], "instances": [{"type":"direct","primitive":{"type":"sphere"}}]}')
PY
$W/load_scenes scenes/*.json $W/rs/scenes/*.json | tail -6
$W/png_fuzz assets/*.png
cp oracle/libpt_oracle.so $W/libpt_oracle.orig.so
trap 'cp $W/libpt_oracle.orig.so oracle/libpt_oracle.so; touch oracle/libpt_oracle.so' EXIT
gcc -std=c11 -O1 -g -fPIC -shared -pthread -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -o oracle/libpt_oracle.so oracle/pt_oracle.c -lm
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python3 -m pytest tests/test_oracle_golden.py -x -q | tail -2
# 4. the per-scene build's helper process and its request / result files
g++ $FLAGS pathtrace_amd/csrc/device/pt_spec_cc.cpp -o $W/pt_spec_cc_asan -ldl
g++ $FLAGS tools/sanitize/rtc_roundtrip.cpp -o $W/rtc_roundtrip -ldl
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 $W/rtc_roundtrip $W/pt_spec_cc_asan $W
echo "sanitizers: clean"
