#!/bin/bash
# scratch experiment: LSTM-768 cluster backward
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "lstm768_cluster" -s > gpurun_out/exp_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/exp_pytest.log
grep -E "passed|failed|worst|Error|error" gpurun_out/exp_pytest.log | tail -20
for rec in f32 split; do
LIPREADING_RECURRENCE=$rec timeout 300 python bench.py --regime landmarks --model lstm768 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/exp_bench_$rec.log 2>&1
echo "bench $rec exit $?"
tail -1 gpurun_out/exp_bench_$rec.log | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
  j = json.loads(l)
  print({k: j[k] for k in ('value', 'ms_per_step', 'final_loss') if k in j}); print(json.dumps(j.get('roofline', {}).get('avg_launch_us_by_kernel'))); print(j.get('pair_errors'))
except Exception as e:
  print('no json', e, l[-500:])
"
done
