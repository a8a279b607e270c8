#!/usr/bin/env bash
# round 5, first visit: where the 1-rank distributed path's extra 0.21 ms sits, and the ecd family on the step kernels
set -u
OUT=gpurun_out; mkdir -p $OUT
bash tools/gpu_timeline.sh r05a_px conv1_fwd --regime pixels > /dev/null
LIPREADING_BENCH_FORCE_DIST=1 bash tools/gpu_timeline.sh r05a_pxdist conv1_fwd --regime pixels > /dev/null
bash tools/gpu_timeline.sh r05a_lm step_begin --regime landmarks --model gru256 > /dev/null
LIPREADING_BENCH_FORCE_DIST=1 bash tools/gpu_timeline.sh r05a_lmdist step_begin --regime landmarks --model gru256 > /dev/null
for b in 32 128; do
  bash tools/gpu_kt.sh r05a_ecd$b --regime landmarks_attn --model lstm768 --attention none --char-dim 256 --batch $b > /dev/null
  python bench.py --regime landmarks_attn --model lstm768 --attention none --char-dim 256 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r05a_ecd$b.json
done
ls -la $OUT | grep r05a
