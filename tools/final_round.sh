#!/usr/bin/env bash
# One GPU-box visit at the end of a round: smoke(), the whole -m gpu suite, then the round's artefacts
# (tools/refresh_profiles.sh).   usage: gpurun --timeout 3000 -- 'bash tools/final_round.sh r04'
set -u
TAG=${1:-r04}
OUT=gpurun_out
mkdir -p $OUT
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $OUT/${TAG}_smoke.log 2>&1
echo "smoke exit $?"; tail -2 $OUT/${TAG}_smoke.log
timeout 1800 python -m pytest tests -m gpu -q --durations=10 > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|FAILED|Error" $OUT/${TAG}_pytest_gpu.log | tail -5
bash tools/refresh_profiles.sh $TAG > $OUT/${TAG}_refresh.log 2>&1
echo "refresh exit $?"
