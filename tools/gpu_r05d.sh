#!/usr/bin/env bash
# round 5, fourth visit: fgemm with job fields in registers + two stages of loads in flight; 8 hardware queues by default
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fgemm.py tests/test_gpu_transformer.py -q -x > $OUT/r05d_pytest.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/r05d_pytest.log
bash tools/gpu_timeline.sh r05d_tfm conv1_fwd --regime pixels_tfm > /dev/null
head -3 $OUT/r05d_tfm_step_timeline.txt
timeout 300 python bench.py --regime pixels_tfm --no-cpu-baseline 2>$OUT/r05d_tfm.err | tail -1 > $OUT/r05d_tfm.json
python -c "
import json; d=json.load(open('$OUT/r05d_tfm.json')); print('pixels_tfm', d['ms_per_step'], d.get('final_loss'))"
for d in 0 1; do
LIPREADING_BENCH_FORCE_DIST=$d timeout 300 python bench.py --regime both --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r05d_both_dist$d.json
python -c "
import json; d=json.load(open('$OUT/r05d_both_dist$d.json')); print('dist$d pixels', d['ms_per_step'], 'landmarks', d['regimes']['landmarks']['ms_per_step'])"
done
