#!/usr/bin/env bash
# Regenerates a round's measurement artefacts on a GPU box (run from the repo root, e.g. through
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh r02'
# ); writes under gpurun_out/, copy what should be judged into profiles/ (then run
# `python tools/make_pmc_traffic.py r02` to rebuild the traffic JSON bench.py reads).
# A second argument "pixels" regenerates only what a change to the conv kernels moves: the default bench line and the
# pixel regimes' kernel statistics, traffic, timeline and counters, most important first.
#   <round>_bench_default.json / _bench_{gru256,lstm768,lstm700,lstm512,gru800}.json / _bench_forcedist.json   bench lines
#   <round>_{pixels,gru256,lstm768,lstm700,landmarks_attn,pixels_tfm}_kernel_stats.txt   rocprofv3 --kernel-trace --stats per kernel
#   <round>_{pixels,gru256,lstm768,lstm700}_pmc_{FETCH,WRITE}_SIZE.txt   HBM/fabric bytes per launch (separate --pmc passes)
#   <round>_{pixels,gru256,lstm768}_pmc_SQ_pass1.txt, <round>_pixels_pmc_SQ_pass2.txt   matrix-pipe / LDS counters of the
#                                                                       conv, recurrence and GEMM kernels
#   <round>_pixels_pmc_SQ_pass3.txt   the conv kernels' wave cycles by state (parked / issue-stalled / issuing per pipe)
#   <round>_{pixels,pixels_tfm,gru256,lstm768}_step_timeline.txt   every dispatch of one step of the TIMED region with start
#                                                                       offset and queue
#   <round>_bench_ecd_lstm768_b{32,128}.json, <round>_ecd_lstm768_b{32,128}_kernel_stats.txt   the reference's ecd flag-file family
#   round 6: <round>_bench_{pixels_b8,pixels_b64,gru256_b64,gru256_b128,lstm512_b64,lstm768_b64,lstm768_b128}[_ns8].json + <round>_pixels_b{8,64}_step_timeline.txt
#            (the per-rank shapes of BASELINE configs[3]), <round>_bench_scaling_model.json (bench.py --model-scaling),
#            <round>_ecd_lstm768_pmc_{FETCH,WRITE}_SIZE.txt / _pmc_SQ_pass1.txt (the grid recurrence's traffic and matrix pipe),
#            <round>_pixels_pmc_clock.txt (GRBM_GUI_ACTIVE: the effective shader clock of the conv kernels),
#            <round>_grid_phase_timing.txt / <round>_conv_patch_phase_timing.txt (shader-clock stamps of the timing variants,
#            when lipreading_amd/_lib/alt/{gridtime,patchtime}.so were built before the visit)
#            <round>_bench_pixels_tfm.json, <round>_pixels_tfm_pmc_{FETCH,WRITE}_SIZE.txt / _pmc_SQ_pass1.txt (the transformer stage's
#            row-block, attention and product kernels), <round>_rowblock_phase_timing.txt (alt/rbtime.so: the forward
#            row-block launch by phase)
# PMC passes never share a run with trace domains other than the kernel trace rocprofv3 adds itself.
set -u
R=$PWD
TAG=${1:-r06}
ONLY=${2:-all}
OUT=$R/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp

kt() {   # name, bench args...
  local name=$1; shift
  # (per-kernel statistics with EVERY side stream off, as bench.py's roofline leg: the default step runs a layer's weight
  # gradient beside its data gradient and the recurrent layers' weight-gradient GEMMs beside the conv backward, and two
  # kernels sharing the chip have no durations of their own)
  (cd /tmp && LIPREADING_CONV_WGRAD_SIDE=0 LIPREADING_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats -d "$OUT/kt_$name" -o kt -- \
     python "$R/bench.py" "$@" --no-graph --steps 15 --warmup 2 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
  python tools/rocpd_summary.py "$(find "$OUT/kt_$name" -name '*.db' | head -1)" 60 > "$OUT/${TAG}_${name}_kernel_stats.txt"
  rm -rf "$OUT/kt_$name"
}
tl() {   # name, anchor kernel, step index, bench args...: one step of the TIMED region as a timeline.  The trace ends with
         # bench.py's roofline leg (5 eager steps in the pixel regimes, 8 elsewhere); the 8 timed steps sit in front of it:
         # index -9 from the end in the pixel regimes.  The landmark regimes' trace ends with the
         # recurrence = 'f32' option run instead, and their probe phase has no staging copies: index 8 from the start.  (Rounds 2-4 took step 8 from the START: a step of the
         # untimed eager-versus-replay probe, with staging copies the timed steps do not have.)
  local name=$1 anchor=$2 which=$3; shift 3
  (cd /tmp && rocprofv3 --kernel-trace -d "$OUT/tl_$name" -o kt -- \
     python "$R/bench.py" "$@" --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
  python tools/rocpd_timeline.py "$(find "$OUT/tl_$name" -name '*.db' | head -1)" "$anchor" $which > "$OUT/${TAG}_${name}_step_timeline.txt"
  rm -rf "$OUT/tl_$name"
}
pmc() {   # name, counter list, filters..., -- bench args
  local name=$1 counters=$2 suffix=$3; shift 3
  local filters=()
  while [ "$1" != "--" ]; do filters+=("$1"); shift; done
  shift
  (cd /tmp && LIPREADING_CONV_WGRAD_SIDE=0 LIPREADING_OVERLAP_WGRAD=0 rocprofv3 --pmc $counters -d "$OUT/pmc_$name" -o pm -- \
     python "$R/bench.py" "$@" --no-graph --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
  python tools/rocpd_pmc.py "$(find "$OUT/pmc_$name" -name '*.db' | head -1)" "${filters[@]}" > "$OUT/${TAG}_${suffix}.txt"
  rm -rf "$OUT/pmc_$name"
}

# which kernels these artefacts are of (bench.py quotes the PMC figures only on a build with the same fingerprint)
python -c "from lipreading_amd import _build; print(_build._fingerprint())" > "$OUT/${TAG}_source_fingerprint.txt"
python bench.py 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_default.json"
kt pixels --regime pixels
for c in FETCH_SIZE WRITE_SIZE; do pmc px_$c $c pixels_pmc_$c -- --regime pixels; done
tl pixels conv1_fwd -9 --regime pixels
# the per-rank shapes of a data-parallel run of BASELINE configs[3] (64 clips): the whole batch on one GPU, and 8 per rank
for b in 64 8; do
  python bench.py --regime pixels --batch $b --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_pixels_b$b.json"
  tl pixels_b$b conv1_fwd -9 --regime pixels --batch $b
done
python bench.py --regime pixels --model-scaling --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_scaling_model.json"
pmc clk "GRBM_GUI_ACTIVE GRBM_COUNT" pixels_pmc_clock conv_ conv1_ rnnc_ xgemm -- --regime pixels
tl pixels_tfm conv1_fwd -9 --regime pixels_tfm
kt pixels_tfm --regime pixels_tfm
python bench.py --regime pixels_tfm --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_pixels_tfm.json"
for c in FETCH_SIZE WRITE_SIZE; do pmc tfm_$c $c pixels_tfm_pmc_$c tfm_rb attn_fused fgemm -- --regime pixels_tfm; done
pmc sq6 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" pixels_tfm_pmc_SQ_pass1 tfm_rb attn_fused fgemm -- --regime pixels_tfm
if [ -f lipreading_amd/_lib/alt/rbtime.so ]; then
  LIPREADING_HIP_LIB=$R/lipreading_amd/_lib/alt/rbtime.so python tools/probes/rb_timing.py > "$OUT/${TAG}_rowblock_phase_timing.txt" 2>&1
fi
LIPREADING_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_forcedist.json"
pmc sq1 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" pixels_pmc_SQ_pass1 conv_ conv1_ rnnc_ xgemm -- --regime pixels
pmc sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" pixels_pmc_SQ_pass2 conv_ conv1_ rnnc_ xgemm -- --regime pixels
# where the conv kernels' wave cycles go: parked (s_waitcnt / barrier), issue-stalled, or issuing — and in which pipe
pmc sq2b "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" pixels_pmc_SQ_pass3 conv_ conv1_ -- --regime pixels
if [ "$ONLY" != "pixels" ]; then
  for m in gru256 lstm768 lstm700 lstm512 gru800; do
    python bench.py --regime landmarks --model $m 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_$m.json"
  done
  kt gru256 --regime landmarks --model gru256
  kt lstm768 --regime landmarks --model lstm768
  kt lstm700 --regime landmarks --model lstm700
  kt landmarks_attn --regime landmarks_attn
  # the reference's dominant flag-file family as shipped (config/archive/experiments/ecd/*: BiLSTM-768, char_dim 256,
  # attention none, rnn_dropout 0.3) at 32 and at the files' own batch of 128
  for b in 32 128; do
    python bench.py --regime landmarks_attn --model lstm768 --attention none --char-dim 256 --rnn-dropout 0.3 --batch $b 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_ecd_lstm768_b$b.json"
  done
  # config/train/attn/attention_type as shipped (BiLSTM-512 -> LSTM-1024, 1_layer_nn, char_dim 256) at B = 32
  python bench.py --regime landmarks_attn --model lstm512 --char-dim 256 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_attn_lstm512_b32.json"
  kt ecd_lstm768_b32 --regime landmarks_attn --model lstm768 --attention none --char-dim 256 --batch 32
  kt ecd_lstm768_b128 --regime landmarks_attn --model lstm768 --attention none --char-dim 256 --batch 128
  for c in FETCH_SIZE WRITE_SIZE; do
    pmc ecd_$c $c ecd_lstm768_pmc_$c rnng_ rnnc_ -- --regime landmarks_attn --model lstm768 --attention none --char-dim 256 --batch 32
  done
  pmc sq5 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" ecd_lstm768_pmc_SQ_pass1 rnng_ rnnc_ fgemm -- --regime landmarks_attn --model lstm768 --attention none --char-dim 256 --batch 32
  for a in "gru256 64" "gru256 128" "lstm512 64"; do
    set -- $a
    python bench.py --regime landmarks --model $1 --batch $2 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_$1_b$2.json"
  done
  # sixteen samples per cluster (BiLSTM-768 / 700 past B = 32) against the 8-sample form (test hook bit 4), same box
  for b in 64 128; do
    python bench.py --regime landmarks --model lstm768 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_lstm768_b$b.json"
    LIPREADING_RNN_DEBUG=16 python bench.py --regime landmarks --model lstm768 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_lstm768_b${b}_ns8.json"
  done
  tl ecd_lstm768_b32 step_begin_ctc 8 --regime landmarks_attn --model lstm768 --attention none --char-dim 256 --batch 32
  if [ -f lipreading_amd/_lib/alt/gridtime.so ]; then
    (LIPREADING_HIP_LIB=$R/lipreading_amd/_lib/alt/gridtime.so python tools/probes/grid_timing.py 32 31; LIPREADING_HIP_LIB=$R/lipreading_amd/_lib/alt/gridtime.so python tools/probes/grid_timing.py 64 31) > "$OUT/${TAG}_grid_phase_timing.txt" 2>&1
  fi
  if [ -f lipreading_amd/_lib/alt/patchtime.so ]; then
    LIPREADING_HIP_LIB=$R/lipreading_amd/_lib/alt/patchtime.so python tools/probes/conv_patch_timing.py > "$OUT/${TAG}_conv_patch_phase_timing.txt" 2>&1
  fi
  tl gru256 step_begin 8 --regime landmarks --model gru256
  tl lstm768 step_begin 8 --regime landmarks --model lstm768
  for c in FETCH_SIZE WRITE_SIZE; do
    pmc gru_$c $c gru256_pmc_$c -- --regime landmarks --model gru256
    pmc lstm_$c $c lstm768_pmc_$c -- --regime landmarks --model lstm768
    pmc lstm7_$c $c lstm700_pmc_$c -- --regime landmarks --model lstm700
  done
  pmc sq3 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" gru256_pmc_SQ_pass1 rnnc_ sgemm xgemm -- --regime landmarks --model gru256
  pmc sq4 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" lstm768_pmc_SQ_pass1 rnnc_ sgemm xgemm -- --regime landmarks --model lstm768
fi
ls -la "$OUT" | grep "${TAG}_"
