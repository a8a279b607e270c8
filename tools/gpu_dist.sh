#!/usr/bin/env bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_two_ranks.py -m gpu -q -x > $OUT/dist_pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/dist_pytest.log | tail -4
LIPREADING_BENCH_FORCE_DIST=1 timeout 600 python bench.py --regime pixels --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/dist_bench.json
python -c "
import json; d=json.load(open('gpurun_out/dist_bench.json')); print(d['value'], d['ms_per_step'], d.get('rccl_ranks'), d.get('all_reduce_buckets'))"
