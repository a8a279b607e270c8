#!/usr/bin/env bash
python tools/bench_conv_patch.py 30 2>&1 | grep -v amdgpu.ids
bash tools/gpu_ab.sh ${1:-ab}
