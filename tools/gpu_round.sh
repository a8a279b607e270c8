#!/usr/bin/env bash
# One GPU-box visit: the -m gpu suite, the default bench line, the forced-distributed single-GPU line.
# usage (from the repo root): gpurun --timeout 1800 -- 'bash tools/gpu_round.sh r02a'
set -u
TAG=${1:-r02}
OUT=gpurun_out
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -s --durations=15 > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/${TAG}_pytest_gpu.log
grep -E "parity|vs oracle|split vs|passed|failed|FAILED|pytest exit" $OUT/${TAG}_pytest_gpu.log | tail -25
timeout 900 python bench.py > $OUT/${TAG}_bench_default.log 2>&1
tail -1 $OUT/${TAG}_bench_default.log > $OUT/${TAG}_bench_default.json
LIPREADING_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --repeats 3 > $OUT/${TAG}_bench_forcedist.log 2>&1
tail -1 $OUT/${TAG}_bench_forcedist.log > $OUT/${TAG}_bench_forcedist.json
python -c "
import json,sys
for f in ('$OUT/${TAG}_bench_default.json','$OUT/${TAG}_bench_forcedist.json'):
  try:
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], 'pair_errors', d.get('pair_errors'), (d.get('parity') or {}).get('abs_diff'))
    for k,v in d.get('regimes',{}).items():
      print('  ', k, v['value'], v['ms_per_step'], (v.get('parity') or {}).get('abs_diff'), {n:(o['ms_per_step'], o['final_loss_delta_vs_default']) for n,o in v.get('other_recurrences',{}).items()})
      r=v.get('roofline') or {}
      print('      ', r.get('kernel'), r.get('avg_launch_us'), r.get('us_per_step'), r.get('frac'), r.get('avg_launch_us_by_kernel'))
  except Exception as e:
    print(f, 'unreadable', e); print(open(f.replace('.json','.log')).read()[-3000:])
"
