#!/usr/bin/env bash
# round 5: regime R's weight half as split-K lr_fgemm jobs (bias gradients = column sums): tests, A/B, timeline
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_fgemm.py tests/test_gpu_encoder.py tests/test_gpu_train.py tests/test_gpu_two_ranks.py tests/test_gpu_decoder.py -q -x > $OUT/r05m_pytest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/r05m_pytest.log
line() {   # tag, env...
  local tag=$1; shift
  env "$@" timeout 300 python bench.py --regime landmarks --model gru256 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$tag', d['ms_per_step'], d['timing']['ms_per_step_min'], 'loss', d['final_loss'])"
}
line split_fgemm A=1
line fp32_grouped LIPREADING_RNN_DEBUG=4
line split_fgemm_again A=1
for m in lstm512 lstm768; do
  for e in 0 4; do
    LIPREADING_RNN_DEBUG=$e timeout 300 python bench.py --regime landmarks --model $m --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$m debug=$e', d['ms_per_step'], d['timing']['ms_per_step_min'])"
  done
done


