#!/usr/bin/env bash
# round 5: how many hardware queues?  the 1-rank RCCL step of both regimes under GPU_MAX_HW_QUEUES = 4, 5, 6, 8
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_two_ranks.py tests/test_gpu_train.py -q -x > $OUT/r05f_pytest.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/r05f_pytest.log
for q in 4 5 6 8; do
for d in 1 0; do
GPU_MAX_HW_QUEUES=$q LIPREADING_BENCH_FORCE_DIST=$d timeout 300 python bench.py --regime both --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 > $OUT/r05f_q${q}_dist$d.json
python -c "
import json; d=json.load(open('$OUT/r05f_q${q}_dist$d.json')); print('queues $q dist$d pixels', d['ms_per_step'], 'landmarks', d['regimes']['landmarks']['ms_per_step'])"
done
done
