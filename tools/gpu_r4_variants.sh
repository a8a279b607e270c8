#!/usr/bin/env bash
# Round 4, first GPU visit: every build-time variant that round 3 left untimed (tools/build_prepared_variants.sh) runs
# the tests of the code it changes, then is timed — pixel regime for all of them, the landmark regimes for the
# recurrence / CTC variants.  In-tree first and last: the drift between the two is the box's noise.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r4_variants.sh'
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r04_variants.txt; : > $LOG
ALT=$GRAFT_REPO_ROOT/lipreading_amd/_lib/alt
line() {   # tag, bench args...
  local tag=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
try:
  d = json.loads(sys.stdin.read())
  r = d.get('roofline') or {}
  c = (d.get('ctc') or {}).get('kernels')
  print('$tag', '$*', d['ms_per_step'], 'min', d['timing']['ms_per_step_min'], 'errs', d.get('pair_errors'),
        {k: round(v, 1) for k, v in (r.get('avg_launch_us_by_kernel') or {}).items()}, c)
except Exception as e:
  print('$tag', '$*', 'FAILED', e)" | tee -a $LOG
}
tests() {   # tag, pytest args
  local tag=$1; shift
  echo "$tag tests: $(timeout 600 python -m pytest -m gpu -q -x "$@" 2>&1 | tail -1)" | tee -a $LOG
}
use() { if [ "$1" = in-tree ]; then unset LIPREADING_HIP_LIB; else export LIPREADING_HIP_LIB=$ALT/$1.so; fi; }

use in-tree
line in-tree --regime pixels
line in-tree --regime landmarks --model gru256
line in-tree --regime landmarks --model lstm768
for tag in p2order p3wg3 noring wreg; do
  use $tag; tests $tag tests/test_gpu_frontend.py; line $tag --regime pixels
done
use xbk64; tests xbk64 tests/test_gpu_frontend.py tests/test_gpu_transformer.py tests/test_gpu_encoder.py -k "pixel or transformer or split or proj or xgemm or oracle"
line xbk64 --regime pixels
timeout 120 python tools/bench_xgemm.py 2>&1 | tail -12 | sed 's/^/xbk64 /' | tee -a $LOG
use in-tree
timeout 120 python tools/bench_xgemm.py 2>&1 | tail -12 | sed 's/^/in-tree /' | tee -a $LOG
for tag in prewait prewait2; do
  use $tag; tests $tag tests/test_gpu_encoder.py tests/test_gpu_decoder.py
  line $tag --regime pixels
  line $tag --regime landmarks --model gru256
  line $tag --regime landmarks --model lstm768
done
use ctcpin; tests ctcpin tests/test_gpu_ctc.py
line ctcpin --regime landmarks --model gru256
line ctcpin --regime pixels
use in-tree
line in-tree --regime pixels
line in-tree --regime landmarks --model gru256
line in-tree --regime landmarks --model lstm768
