#!/usr/bin/env bash
# SQ / GRBM counters of the conv kernels of the pixel regime (one --pmc pass per counter group)
set -u
OUT=gpurun_out; mkdir -p $OUT; R=$GRAFT_REPO_ROOT; TAG=${1:-pmc}
export TMPDIR=/tmp
pmc() { # tag counters
  (cd /tmp && LIPREADING_CONV_WGRAD_SIDE=0 timeout 300 rocprofv3 --pmc $2 -d "$R/$OUT/pmc_$1" -o pm -- \
     python "$R/bench.py" --regime pixels --no-graph --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
  python tools/rocpd_pmc.py "$(find "$OUT/pmc_$1" -name '*.db' | head -1)" conv > "$OUT/${TAG}_pmc_$1.txt"
  rm -rf "$OUT/pmc_$1"
  cut -c1-60,64-130 "$OUT/${TAG}_pmc_$1.txt"
}
pmc a "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY"
(cd /tmp && LIPREADING_CONV_WGRAD_SIDE=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$R/$OUT/kt" -o kt -- \
  python "$R/bench.py" --regime pixels --no-graph --no-cpu-baseline --steps 10 --repeats 1 > /dev/null 2>&1)
python tools/rocpd_summary.py "$(find "$OUT/kt" -name '*.db' | head -1)" 60 > $OUT/${TAG}_kernel_stats.txt
grep -i "conv" $OUT/${TAG}_kernel_stats.txt | cut -c1-160 | head -12
rm -rf $OUT/kt
