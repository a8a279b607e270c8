#!/usr/bin/env bash
# One GPU-box visit: the -m gpu suite, the default bench line, the forced-distributed single-GPU line.
# usage (from the repo root): gpurun --timeout 1800 -- 'bash tools/gpu_round.sh r02a'
set -u
TAG=${1:-r02}
OUT=gpurun_out
mkdir -p $OUT
python -m pytest tests -m gpu -x -q --durations=15 > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/${TAG}_pytest_gpu.log
tail -25 $OUT/${TAG}_pytest_gpu.log
python bench.py > $OUT/${TAG}_bench_default.log 2>&1
tail -1 $OUT/${TAG}_bench_default.log > $OUT/${TAG}_bench_default.json
LIPREADING_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --repeats 3 > $OUT/${TAG}_bench_forcedist.log 2>&1
tail -1 $OUT/${TAG}_bench_forcedist.log > $OUT/${TAG}_bench_forcedist.json
python -c "
import json,sys
for f in ('$OUT/${TAG}_bench_default.json','$OUT/${TAG}_bench_forcedist.json'):
  try:
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d.get('parity'), {k:(v['value'],v['ms_per_step'],v.get('parity')) for k,v in d.get('regimes',{}).items()})
  except Exception as e:
    print(f, 'unreadable', e); print(open(f.replace('.json','.log')).read()[-3000:])
"
