#!/usr/bin/env bash
# round 5: early sum of squares, the upper layer's forward projection through lr_fgemm; A/B; regime-R timeline
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_encoder.py tests/test_gpu_frontend.py -q -x -k "early_sum or bf16x3 or pixel or ctc_step or graphs" > $OUT/r05l_pytest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/r05l_pytest.log
line() {   # tag, env...
  local tag=$1; shift
  env "$@" timeout 300 python bench.py --regime both --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$tag', d['ms_per_step'], d['timing']['ms_per_step_min'], 'loss', d['final_loss'], 'landmarks', d['regimes']['landmarks']['ms_per_step'])"
}
line default A=1
line no_early_sumsq LIPREADING_SUMSQ_EARLY=0
line default_again A=1
TL_STEP=-12 bash tools/gpu_timeline.sh r05l_lm step_begin --regime landmarks --model gru256 > /dev/null
cat $OUT/r05l_lm_step_timeline.txt | cut -c1-100
