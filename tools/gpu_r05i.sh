#!/usr/bin/env bash
# round 5: the whole -m gpu suite + smoke + default bench line on the current tree
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $OUT/r05i_smoke.log 2>&1
echo "smoke exit $?"; tail -2 $OUT/r05i_smoke.log
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/r05i_pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|FAILED|Error" $OUT/r05i_pytest_gpu.log | tail -8
timeout 900 python bench.py > $OUT/r05i_bench_default.log 2>&1
tail -1 $OUT/r05i_bench_default.log > $OUT/r05i_bench_default.json
python -c "
import json
d=json.load(open('$OUT/r05i_bench_default.json')); print('default', d['value'], d['ms_per_step'], (d.get('parity') or {}).get('abs_diff'), d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline'].get('thread_sweep_ms_per_step'))
for k,v in d.get('regimes',{}).items():
  print('  ', k, v['ms_per_step'], (v.get('parity') or {}).get('abs_diff'), (v.get('cpu_baseline') or {}).get('value'), (v.get('cpu_baseline') or {}).get('cores'), (v.get('cpu_baseline') or {}).get('thread_sweep_ms_per_step'))
" || tail -20 $OUT/r05i_bench_default.log
