#!/usr/bin/env bash
# A/B of kernel variants in ONE GPU-box visit: the frontend (or any) parity tests on the in-tree build, then the pixel
# regime's per-kernel times (HIP events, bench.py's roofline leg) for the in-tree build and for every variant library
# built beforehand with tools/build_variant.sh (they travel with the snapshot: lipreading_amd/_lib/alt/<tag>.so).
#   usage (through gpurun): bash tools/gpu_ab_libs.sh "<pytest -k expression or ''>" <tag> [<tag> ...]
#   e.g.  gpurun --timeout 300 -- 'bash tools/gpu_ab_libs.sh "frontend" noring lb3'
# Every variant also runs the same tests (AB_TEST_VARIANTS=0 to skip).
# Output: one line of kernel times per build, in-tree first and last (the box's drift between the two is the noise).
set -u
OUT=gpurun_out; mkdir -p $OUT
EXPR=$1; shift
if [ -n "$EXPR" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x -k "$EXPR" 2>&1 | tail -2
fi
line() {
  timeout 300 python bench.py --regime pixels --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
print('$1', d['ms_per_step'], {k: round(v, 1) for k, v in (r.get('avg_launch_us_by_kernel') or {}).items()})"
}
unset LIPREADING_HIP_LIB
line in-tree
for tag in "$@"; do
  export LIPREADING_HIP_LIB=$GRAFT_REPO_ROOT/lipreading_amd/_lib/alt/$tag.so
  [ -f "$LIPREADING_HIP_LIB" ] || { echo "$tag: no such variant library"; continue; }
  if [ -n "$EXPR" ] && [ "${AB_TEST_VARIANTS:-1}" = 1 ]; then   # a variant is only worth timing if it is RIGHT
    echo "$tag: $(timeout 900 python -m pytest tests -m gpu -q -x -k "$EXPR" 2>&1 | tail -1)"
  fi
  line $tag
done
unset LIPREADING_HIP_LIB
line in-tree
