#!/usr/bin/env bash
# round 5: a recurrent layer's whole split-bf16 weight half as one lr_fgemm launch (row-shifted B for dW_hh, bias gradients
# from the column sums); A/B against the packed path (LIPREADING_RNN_DEBUG=8), in-tree first and last
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fgemm.py tests/test_gpu_encoder.py tests/test_gpu_frontend.py tests/test_gpu_train.py tests/test_gpu_two_ranks.py -q -x > $OUT/r05k_pytest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/r05k_pytest.log
line() {   # tag, env...
  local tag=$1; shift
  env "$@" timeout 300 python bench.py --regime pixels --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$tag', d['ms_per_step'], d['timing']['ms_per_step_min'], 'loss', d['final_loss'])"
}
line fgemm_weight_half A=1
line packed LIPREADING_RNN_DEBUG=8
line fgemm_weight_half_again A=1
bash tools/gpu_timeline.sh r05k_px conv1_fwd --regime pixels > /dev/null
cat $OUT/r05k_px_step_timeline.txt | cut -c1-100
