#!/usr/bin/env bash
# One GPU-box visit: a kernel-trace timeline of one step of a regime (bench args in "$@") and its per-kernel statistics.
#   gpurun --timeout 600 -- 'bash tools/gpu_timeline.sh r04a conv1_fwd --regime pixels'
set -u
R=$PWD; TAG=$1; ANCHOR=$2; shift 2
OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace -d "$OUT/tl_$TAG" -o kt -- python "$R/bench.py" "$@" --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
DB=$(find "$OUT/tl_$TAG" -name '*.db' | head -1)
python tools/rocpd_timeline.py "$DB" "$ANCHOR" 8 > "$OUT/${TAG}_step_timeline.txt"
python tools/rocpd_summary.py "$DB" 60 > "$OUT/${TAG}_kernel_stats.txt"
rm -rf "$OUT/tl_$TAG"
head -5 "$OUT/${TAG}_step_timeline.txt"
