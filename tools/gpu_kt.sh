#!/usr/bin/env bash
# kernel trace of one bench configuration: bash tools/gpu_kt.sh <tag> <bench args...>
set -u
OUT=gpurun_out; mkdir -p $OUT; R=$GRAFT_REPO_ROOT; TAG=$1; shift
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$OUT/kt_$TAG" -o kt -- \
  python "$R/bench.py" "$@" --no-graph --no-cpu-baseline --steps 10 --repeats 1 > /dev/null 2>&1)
python tools/rocpd_summary.py "$(find "$OUT/kt_$TAG" -name '*.db' | head -1)" 60 > $OUT/${TAG}_kernel_stats.txt
rm -rf $OUT/kt_$TAG
head -40 $OUT/${TAG}_kernel_stats.txt | cut -c1-150
