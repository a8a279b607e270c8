#!/usr/bin/env bash
# round 5: recurrent layers' prologues hoisted under the conv forward: tests, A/B, timeline
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_train.py tests/test_gpu_encoder.py tests/test_gpu_two_ranks.py tests/test_gpu_distributed.py -q -x > $OUT/r05q_pytest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/r05q_pytest.log
line() {   # tag, env...
  local tag=$1; shift
  env "$@" timeout 300 python bench.py --regime pixels --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$tag', d['ms_per_step'], d['timing']['ms_per_step_min'], 'loss', d['final_loss'])"
}
line hoisted A=1
line in_place LIPREADING_PREPARE_STEP=0
line hoisted_again A=1
line in_place_again LIPREADING_PREPARE_STEP=0
bash tools/gpu_timeline.sh r05q_px conv1_fwd --regime pixels > /dev/null
cut -c1-120 $OUT/r05q_px_step_timeline.txt | head -24
