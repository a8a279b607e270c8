#!/usr/bin/env bash
# round 5: full GPU suite + smoke; pixel-regime weight halves cut along K (A/B); pixel timeline
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/r05o_pytest.log 2>&1
echo "pytest exit $?"; tail -14 $OUT/r05o_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
line() {   # tag, env...
  local tag=$1; shift
  env "$@" timeout 300 python bench.py --regime pixels --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$tag', d['ms_per_step'], d['timing']['ms_per_step_min'], 'loss', d['final_loss'])"
}
line cut A=1
line uncut LIPREADING_RNN_DEBUG=4
line cut_again A=1
line uncut_again LIPREADING_RNN_DEBUG=4
bash tools/gpu_timeline.sh r05o_px conv1_fwd --regime pixels > /dev/null
cut -c1-120 $OUT/r05o_px_step_timeline.txt
