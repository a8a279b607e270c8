#!/usr/bin/env bash
# SQ counters of the conv weight-gradient kernels (second form)
set -u
OUT=gpurun_out; mkdir -p $OUT; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
pmc() { # tag counters
  (cd /tmp && LIPREADING_CONV_WGRAD_SIDE=0 timeout 300 rocprofv3 --pmc $2 -d "$R/$OUT/pmc_$1" -o pm -- \
     python "$R/bench.py" --regime pixels --no-graph --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
  python tools/rocpd_pmc.py "$(find "$OUT/pmc_$1" -name '*.db' | head -1)" conv_wgrad_tr conv_patch_kernel > "$OUT/r3i_pmc_$1.txt"
  rm -rf "$OUT/pmc_$1"
  cut -c1-60,64-130 "$OUT/r3i_pmc_$1.txt"
}
pmc a "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"
pmc b "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
pmc c "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"
