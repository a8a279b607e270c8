#!/usr/bin/env bash
# round 5, second visit: the products of lr_fgemm.hip and the transformer stack built from them
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fgemm.py tests/test_gpu_transformer.py -q -x --durations=5 > $OUT/r05b_pytest.log 2>&1
echo "pytest exit $?"; tail -15 $OUT/r05b_pytest.log
timeout 300 python bench.py --regime pixels_tfm --no-cpu-baseline 2>$OUT/r05b_tfm.err | tail -1 > $OUT/r05b_tfm.json
python -c "
import json; d=json.load(open('$OUT/r05b_tfm.json')); print('pixels_tfm', d['ms_per_step'], d.get('final_loss'))"
tail -5 $OUT/r05b_tfm.err
bash tools/gpu_timeline.sh r05b_tfm conv1_fwd --regime pixels_tfm > /dev/null
head -3 $OUT/r05b_tfm_step_timeline.txt
