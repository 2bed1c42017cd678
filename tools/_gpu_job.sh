#!/bin/bash
bash tools/profile_gpu.sh r02c
