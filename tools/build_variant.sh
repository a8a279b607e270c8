#!/bin/bash
# Build a variant library for A/B timing: tools/build_variant.sh <tag> <path/to/variant of a csrc/*.hip> [name of the
# translation unit it replaces, default: the variant's own file name]
# -> lipreading_amd/_lib/alt/<tag>.so (the other objects come from the current in-tree build).
# VARIANT_DEFS: extra compiler flags, e.g. VARIANT_DEFS="-DLR_C1_WREG=1" tools/build_variant.sh wreg lipreading_amd/csrc/lr_conv1.hip
set -e
TAG=$1; SRC=$2; UNIT=${3:-$(basename $SRC .hip)}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=/tmp/variant_$TAG.o
EXTRA=""   # the unit's flags of lipreading_amd/_build.py (UNIT_FLAGS)
if [ "$UNIT" = lr_conv1 ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $EXTRA ${VARIANT_DEFS:-} -I$ROOT/lipreading_amd/csrc -I$ROOT/include -c $SRC -o $OBJ
OTHERS=$(ls $ROOT/lipreading_amd/_lib/obj/*.o | grep -v "/$UNIT.o")
mkdir -p $ROOT/lipreading_amd/_lib/alt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/lipreading_amd/_lib/alt/$TAG.so $OBJ $OTHERS
echo built $ROOT/lipreading_amd/_lib/alt/$TAG.so
