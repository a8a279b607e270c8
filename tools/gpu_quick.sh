#!/usr/bin/env bash
# One short GPU-box visit: the given tests (pytest args in $1), then the pixel regime's line and per-kernel times.
#   gpurun --timeout 900 -- 'bash tools/gpu_quick.sh "tests/test_gpu_frontend.py" [regime args...]'
set -u
OUT=gpurun_out; mkdir -p $OUT
T=$1; shift
if [ -n "$T" ]; then timeout 900 python -m pytest -m gpu -q -x $T 2>&1 | tail -15; fi
line() {
  timeout 300 python bench.py "$@" --no-cpu-baseline 2>$OUT/quick_err.log | tail -1 | python -c "
import json, sys
try:
  d = json.loads(sys.stdin.read())
  r = d.get('roofline') or {}
  print('$*', d['ms_per_step'], 'min', d['timing']['ms_per_step_min'], 'errs', d.get('pair_errors'), d['config'].get('launch_probe'),
        {k: round(v, 1) for k, v in (r.get('avg_launch_us_by_kernel') or {}).items()})
except Exception as e:
  print('$*', 'FAILED', e); print(open('$OUT/quick_err.log').read()[-2000:])"
}
if [ $# -eq 0 ]; then line --regime pixels; else
  for r in "$@"; do line $r; done
fi
