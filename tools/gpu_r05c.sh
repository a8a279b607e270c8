#!/usr/bin/env bash
# round 5, third visit: fgemm with unpredicated loads, the encoder tests after the pair / persist removal, tfm timeline
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fgemm.py tests/test_gpu_transformer.py tests/test_gpu_encoder.py tests/test_gpu_decoder.py -q -x --durations=5 > $OUT/r05c_pytest.log 2>&1
echo "pytest exit $?"; tail -12 $OUT/r05c_pytest.log
timeout 300 python bench.py --regime pixels_tfm --no-cpu-baseline 2>$OUT/r05c_tfm.err | tail -1 > $OUT/r05c_tfm.json
python -c "
import json; d=json.load(open('$OUT/r05c_tfm.json')); print('pixels_tfm', d['ms_per_step'], d.get('final_loss'))"
bash tools/gpu_timeline.sh r05c_tfm conv1_fwd --regime pixels_tfm > /dev/null
head -3 $OUT/r05c_tfm_step_timeline.txt
GPU_MAX_HW_QUEUES=8 LIPREADING_BENCH_FORCE_DIST=1 bash tools/gpu_timeline.sh r05c_pxdist_q8 conv1_fwd --regime pixels > /dev/null
head -3 $OUT/r05c_pxdist_q8_step_timeline.txt
