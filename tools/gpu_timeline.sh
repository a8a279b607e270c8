#!/usr/bin/env bash
# One GPU-box visit: a kernel-trace timeline of one step of a regime (bench args in "$@") and its per-kernel statistics.
#   gpurun --timeout 600 -- 'bash tools/gpu_timeline.sh r04a conv1_fwd --regime pixels'
# The step shown is one of the TIMED region's: with --steps 8 the trace ends with bench.py's roofline leg (5 eager steps
# in the pixel regimes, 8 elsewhere), the timed steps sit right in front of it (TL_STEP: index from the end, default
# -9 = inside the timed region of a pixel regime; landmarks: TL_STEP=8, from the start — their trace ends with the recurrence = f32 option run).  (Rounds 2-4 showed step 8 from the start —
# a step of the untimed eager-versus-replay probe, whose replays still copy the inputs into the graph's buffers: the
# "staging copies" and the idle gap in front of them in those timelines are not part of the timed step.)
set -u
R=$PWD; TAG=$1; ANCHOR=$2; shift 2
OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace -d "$OUT/tl_$TAG" -o kt -- python "$R/bench.py" "$@" --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
DB=$(find "$OUT/tl_$TAG" -name '*.db' | head -1)
python tools/rocpd_timeline.py "$DB" "$ANCHOR" ${TL_STEP:--9} > "$OUT/${TAG}_step_timeline.txt"
python tools/rocpd_summary.py "$DB" 60 > "$OUT/${TAG}_kernel_stats.txt"
rm -rf "$OUT/tl_$TAG"
head -5 "$OUT/${TAG}_step_timeline.txt"
