#!/usr/bin/env bash
# round 5, HEAD: smoke + the whole -m gpu suite once more (python-side changes since the refresh), ecd B=128 statistics
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $OUT/r05_smoke.log 2>&1
echo "smoke exit $?"; tail -1 $OUT/r05_smoke.log
timeout 1800 python -m pytest tests -m gpu -q --durations=10 > $OUT/r05_pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|FAILED|Error" $OUT/r05_pytest_gpu.log | tail -5
(cd /tmp && LIPREADING_CONV_WGRAD_SIDE=0 LIPREADING_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats -d "$OUT/kt_ecd128" -o kt -- \
   python "$R/bench.py" --regime landmarks_attn --model lstm768 --attention none --char-dim 256 --batch 128 --no-graph --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/rocpd_summary.py "$(find "$OUT/kt_ecd128" -name '*.db' | head -1)" 40 > "$OUT/r05_ecd_lstm768_b128_kernel_stats.txt"
rm -rf "$OUT/kt_ecd128"
head -14 "$OUT/r05_ecd_lstm768_b128_kernel_stats.txt" | cut -c1-130
