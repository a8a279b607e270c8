#!/usr/bin/env bash
# Builds every variant library that round 3 prepared but could not time into
# lipreading_amd/_lib/alt/: run it HERE (hipcc cross-compiles), after `python -c 'import __graft_entry__ as g; g.build()'`,
# then time them in one GPU-box visit, e.g.
#   gpurun --timeout 900 -- 'bash tools/gpu_ab_libs.sh "frontend or encoder" prewait prewait2 p2order noring p3wg3 wreg'
# (tools/gpu_ab_libs.sh prints the pixel regime's per-kernel times per library; for the landmark regimes and the GEMMs use
#  LIPREADING_HIP_LIB=.../alt/<tag>.so with tools/gpu_ab_landmarks.sh / tools/bench_xgemm.py.)
set -e
cd "$(dirname "$0")/.."
C=lipreading_amd/csrc
VARIANT_DEFS="-DLR_RNNC_PREWAIT=1" bash tools/build_variant.sh prewait $C/lr_rnn_cluster.hip
VARIANT_DEFS="-DLR_RNNC_PREWAIT=1 -DLR_RNNC_UNCOND_FETCH=1" bash tools/build_variant.sh prewait2 $C/lr_rnn_cluster.hip
VARIANT_DEFS="-DLR_P2_PROLOGUE_ORDER=1" bash tools/build_variant.sh p2order $C/lr_conv_patch.hip
VARIANT_DEFS="-DLR_P3_MIN_WGS=3" bash tools/build_variant.sh p3wg3 $C/lr_conv.hip
VARIANT_DEFS="-DLR_C1_WREG=1" bash tools/build_variant.sh wreg $C/lr_conv1.hip
VARIANT_DEFS="-DLR_XBK=64" bash tools/build_variant.sh xbk64 $C/lr_xgemm.hip
VARIANT_DEFS="-DLR_CTC_PIN_LOADS=1" bash tools/build_variant.sh ctcpin $C/lr_ctc.hip
# layer 3 before its fragment ring (the form that was timed in round 3)
if git rev-parse --verify -q 709ee36 > /dev/null; then
  mkdir -p /tmp/lr_noring && git show 709ee36:$C/lr_conv.hip > /tmp/lr_noring/lr_conv.hip
  bash tools/build_variant.sh noring /tmp/lr_noring/lr_conv.hip lr_conv
fi
ls -la lipreading_amd/_lib/alt/
