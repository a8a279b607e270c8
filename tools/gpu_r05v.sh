#!/usr/bin/env bash
# round 5: matrix-pipe counters of the transformer stage's kernels (lr_fgemm, fused attention, layer norm) — one --pmc pass
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && LIPREADING_CONV_WGRAD_SIDE=0 LIPREADING_OVERLAP_WGRAD=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d "$OUT/pmc_tfm" -o pm -- \
   python "$R/bench.py" --regime pixels_tfm --no-graph --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/rocpd_pmc.py "$(find "$OUT/pmc_tfm" -name '*.db' | head -1)" fgemm attn_ ln_ layernorm > "$OUT/r05_pixels_tfm_pmc_SQ_pass1.txt"
rm -rf "$OUT/pmc_tfm"
cut -c1-150 "$OUT/r05_pixels_tfm_pmc_SQ_pass1.txt" | head -40
