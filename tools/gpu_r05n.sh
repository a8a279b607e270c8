#!/usr/bin/env bash
# round 5: backward fragments packed by the forward prologue, CTC batch reduction riding the recursion launch, fp32
# forward projection through lr_fgemm; side-stream priority experiment
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_ctc.py tests/test_gpu_encoder.py tests/test_gpu_train.py tests/test_gpu_decoder.py tests/test_gpu_two_ranks.py -q -x > $OUT/r05n_pytest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/r05n_pytest.log
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
line() {   # tag, env...
  local tag=$1; shift
  env "$@" timeout 300 python bench.py --regime both --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$tag', d['ms_per_step'], d['timing']['ms_per_step_min'], 'loss', d['final_loss'], 'landmarks', d['regimes']['landmarks']['ms_per_step'], d['regimes']['landmarks'].get('final_loss'))"
}
line default A=1
line old_fwd_proj LIPREADING_RNN_DEBUG=8
line side_low LIPREADING_SIDE_PRIORITY=1
line side_high LIPREADING_SIDE_PRIORITY=-1
line default_again A=1
for m in lstm512 lstm768; do
  for e in 0 8; do
    LIPREADING_RNN_DEBUG=$e timeout 300 python bench.py --regime landmarks --model $m --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$m debug=$e', d['ms_per_step'], d['timing']['ms_per_step_min'])"
  done
done
TL_STEP=8 bash tools/gpu_timeline.sh r05n_lm step_begin --regime landmarks --model gru256 > /dev/null
cut -c1-110 $OUT/r05n_lm_step_timeline.txt
