#!/usr/bin/env bash
# Regenerates the round's measurement artefacts on a GPU box (run from the repo root, e.g. through
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r01'
# ); writes under gpurun_out/, copy what should be judged into profiles/.
#   <round>_bench_default.json            the line `python bench.py` prints
#   <round>_pixels_kernel_stats.txt       rocprofv3 --kernel-trace --stats, summarised per kernel
#   <round>_pixels_pmc_{FETCH,WRITE}_SIZE.txt   HBM/fabric bytes per launch (separate --pmc passes)
#   <round>_pixels_pmc_SQ_pass{1,2}.txt   matrix-pipe / LDS counters of the conv, recurrence and xgemm kernels
# PMC passes never share a run with trace domains other than the kernel trace rocprofv3 adds itself.
set -u
R=$PWD
TAG=${1:-r01}
OUT=$R/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp

python bench.py 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_default.json"

(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o px -- \
   python "$R/bench.py" --regime pixels --no-graph --steps 15 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
python tools/rocpd_summary.py "$(find "$OUT/kt" -name '*.db' | head -1)" > "$OUT/${TAG}_pixels_kernel_stats.txt"
rm -rf "$OUT/kt"

for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c -d "$OUT/pmc_$c" -o px -- \
     python "$R/bench.py" --regime pixels --no-graph --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
  python tools/rocpd_pmc.py "$(find "$OUT/pmc_$c" -name '*.db' | head -1)" > "$OUT/${TAG}_pixels_pmc_$c.txt"
  rm -rf "$OUT/pmc_$c"
done

i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i + 1))
  (cd /tmp && rocprofv3 --pmc $c -d "$OUT/pmc_sq$i" -o px -- \
     python "$R/bench.py" --regime pixels --no-graph --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
  python tools/rocpd_pmc.py "$(find "$OUT/pmc_sq$i" -name '*.db' | head -1)" conv_ conv1_ gru256 xgemm \
     > "$OUT/${TAG}_pixels_pmc_SQ_pass$i.txt"
  rm -rf "$OUT/pmc_sq$i"
done
ls -la "$OUT" | grep "${TAG}_"
