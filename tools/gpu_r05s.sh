#!/usr/bin/env bash
# round 5: the lr_fgemm weight half for the LARGE regime-R layers (G*H >= 1536, round 4's packed lr_xgemm path): A/B
# (LIPREADING_RNN_DEBUG=8 keeps them on the packed path)
set -u
OUT=gpurun_out; mkdir -p $OUT
for m in lstm768 lstm512 lstm700 gru800; do
  for e in 0 8 0 8; do
    LIPREADING_RNN_DEBUG=$e timeout 300 python bench.py --regime landmarks --model $m --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$m debug=$e', d['ms_per_step'], d['timing']['ms_per_step_min'], 'loss', d['final_loss'])"
  done
done 2>&1 | tee $OUT/r05s_ab.txt
