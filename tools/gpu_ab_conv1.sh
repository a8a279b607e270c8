#!/usr/bin/env bash
# A/B of first-layer kernel variants in ONE GPU-box visit: the frontend parity tests on the in-tree build, then
# tools/bench_conv1.py (stand-alone times + checksums of the outputs) for the in-tree build and every variant library
# given (lipreading_amd/_lib/alt/<tag>.so), then the pixel regime's line for in-tree, the fastest variant, in-tree.
#   usage (through gpurun): bash tools/gpu_ab_conv1.sh <tag> [<tag> ...]
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_frontend.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -2
one() { timeout 200 python tools/bench_conv1.py 40 2>&1 | grep -E "median|checksum" | sed "s/^/$1: /"; }
unset LIPREADING_HIP_LIB
one in-tree | tee $OUT/ab_conv1.txt
for tag in "$@"; do
  export LIPREADING_HIP_LIB=$GRAFT_REPO_ROOT/lipreading_amd/_lib/alt/$tag.so
  [ -f "$LIPREADING_HIP_LIB" ] || { echo "$tag: no such variant library"; continue; }
  one $tag | tee -a $OUT/ab_conv1.txt
done
unset LIPREADING_HIP_LIB
one in-tree | tee -a $OUT/ab_conv1.txt
BEST=$(grep "weight gradient" $OUT/ab_conv1.txt | grep -v "^in-tree" | sed 's/^\([^:]*\):.*median \([0-9.]*\) us.*/\2 \1/' | sort -n | head -1 | cut -d' ' -f2)
echo "fastest weight gradient: $BEST"
line() {
  timeout 300 python bench.py --regime pixels --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
print('$1', d['ms_per_step'], 'min', d['timing']['ms_per_step_min'], {k: round(v, 1) for k, v in (r.get('avg_launch_us_by_kernel') or {}).items()})"
}
line in-tree
if [ -n "$BEST" ]; then
  export LIPREADING_HIP_LIB=$GRAFT_REPO_ROOT/lipreading_amd/_lib/alt/$BEST.so
  echo "$BEST: $(timeout 900 python -m pytest tests/test_gpu_frontend.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -1)"
  line $BEST
  unset LIPREADING_HIP_LIB
  line in-tree
fi
