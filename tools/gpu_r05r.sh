#!/usr/bin/env bash
# round 5: half-batch pipelining experiment (timing only, tools/exp_half_batch.py) + its timelines
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/exp_half_batch.py --steps 20 2>&1 | grep "ms per step" | tee $OUT/r05r_half_batch.txt
timeout 600 python tools/exp_half_batch.py --steps 20 --graph --only full,serial 2>&1 | grep "ms per step\|Error\|error" | tee -a $OUT/r05r_half_batch.txt
for v in full serial streams; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d "$OUT/tl_hb_$v" -o kt -- python "$R/tools/exp_half_batch.py" --steps 4 --only $v > $OUT/r05r_prof_$v.log 2>&1)
  DB=$(find "$OUT/tl_hb_$v" -name '*.db' | head -1)
  python tools/rocpd_timeline.py "$DB" adam_kernel -3 > "$OUT/r05r_half_batch_${v}_timeline.txt"
  rm -rf "$OUT/tl_hb_$v"
  head -2 "$OUT/r05r_half_batch_${v}_timeline.txt"
done
cut -c1-130 $OUT/r05r_half_batch_streams_timeline.txt
