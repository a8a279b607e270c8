#!/usr/bin/env bash
# A/B of an environment switch in ONE GPU-box visit: the pixel regime's line with VAR=a, VAR=b, a, b (the drift between
# equal settings is the box's noise).   usage (through gpurun): bash tools/gpu_ab_env.sh VAR a b ["pytest -k expression"]
set -u
V=$1; A=$2; B=$3; EXPR=${4:-}
line() {
  env $V=$1 timeout 300 python bench.py --regime pixels --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$V=$1', d['ms_per_step'], 'min', d['timing']['ms_per_step_min'], d['config'].get('launch_probe'))"
}
if [ -n "$EXPR" ]; then for x in $A $B; do echo "$V=$x: $(env $V=$x timeout 900 python -m pytest tests -m gpu -q -x -k "$EXPR" 2>&1 | grep -E 'passed|failed' | tail -1)"; done; fi
line $A; line $B; line $A; line $B
