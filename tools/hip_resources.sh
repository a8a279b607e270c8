#!/usr/bin/env bash
# Compile one csrc/*.hip for gfx950 and print one line of register / scratch usage per kernel (plus any
# error or warning).   usage: tools/hip_resources.sh lipreading_amd/csrc/lr_rnn_cluster.hip [name filter]
set -u
SRC=$1; FILT=${2:-.}
EXTRA=""   # the unit's flags of lipreading_amd/_build.py (UNIT_FLAGS)
if [ "$(basename $SRC .hip)" = lr_conv1 ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EXTRA \
  -I$(dirname $0)/../include -Rpass-analysis=kernel-resource-usage -c "$SRC" -o /tmp/hip_resources.o 2>&1 | python3 -c '
import re, sys
name = None; row = {}
for line in sys.stdin:
  if re.search(r"error|warning:", line): print(line.rstrip())
  m = re.search(r"Function Name: (\S+)", line)
  if m:
    if name: print(name, row)
    name, row = m.group(1), {}
    continue
  m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", line)
  if m and name: row[m.group(1).split()[0] + ("Spill" if "Spill" in m.group(1) else "")] = int(m.group(2))
if name: print(name, row)
' | grep -E "error|warning|$FILT"
