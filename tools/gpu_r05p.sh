#!/usr/bin/env bash
# round 5: full GPU suite (no -x) + smoke
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -q --durations=8 > $OUT/r05p_pytest.log 2>&1
echo "pytest exit $?"; tail -16 $OUT/r05p_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
