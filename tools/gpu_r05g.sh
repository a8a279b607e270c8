#!/usr/bin/env bash
# round 5: fgemm second form (two LDS stage buffers, buffer loads, LDS-staged epilogue); dist tests with the folded words
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fgemm.py tests/test_gpu_transformer.py tests/test_gpu_distributed.py tests/test_gpu_two_ranks.py tests/test_gpu_train.py -q -x > $OUT/r05g_pytest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/r05g_pytest.log
timeout 300 python tools/bench_fgemm.py > $OUT/r05g_fgemm.txt 2>&1; grep -v amdgpu.ids $OUT/r05g_fgemm.txt
timeout 300 python bench.py --regime pixels_tfm --no-cpu-baseline 2>$OUT/r05g_tfm.err | tail -1 > $OUT/r05g_tfm.json
python -c "
import json; d=json.load(open('$OUT/r05g_tfm.json')); print('pixels_tfm', d['ms_per_step'], d.get('final_loss'))"
bash tools/gpu_timeline.sh r05g_tfm conv1_fwd --regime pixels_tfm > /dev/null
head -3 $OUT/r05g_tfm_step_timeline.txt
