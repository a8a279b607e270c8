#!/usr/bin/env bash
# round 5, last change: six hardware queues only for ranks; driver launches the pixel regimes eagerly
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_two_ranks.py tests/test_gpu_train.py -q -x -s -k "one_rank or two_rank or pixel or driver or graphs or skip or shard" > $OUT/r05t_pytest.log 2>&1
echo "pytest exit $?"; grep -E "one-rank exchange|passed|failed" $OUT/r05t_pytest.log | tail -4
bash tools/gpu_head_line.sh r05t | tail -4
python -c "
import json
d = json.load(open('gpurun_out/r05t_bench_default_head.json'))
print(d['config']['launch_probe'], d['config']['launch'][:60])"
