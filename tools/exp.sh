#!/bin/bash
# scratch experiment: fused head with deeper prefetch: encoder tests, landmark benches
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_train.py -x -q -m gpu > gpurun_out/exp_pytest.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/exp_pytest.log | tail -3
for m in gru256 lstm768; do
timeout 600 python bench.py --regime landmarks --model $m --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$m', j['value'], j['ms_per_step'], j['timing']['ms_per_step_min'], j['pair_errors'], j['final_loss'])"
done
