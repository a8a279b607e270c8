#!/bin/bash
# scratch experiment: conv weight gradients on a side stream beside the data gradients
mkdir -p gpurun_out
LIPREADING_CONV_WGRAD_SIDE=1 timeout 600 python -m pytest tests/test_gpu_frontend.py -x -q -m gpu > gpurun_out/exp_pytest.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/exp_pytest.log | tail -3
for round in 1 2 3; do
for v in 0 1; do
LIPREADING_CONV_WGRAD_SIDE=$v timeout 600 python bench.py --regime pixels --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('side=$v', j['value'], j['ms_per_step'], j['timing']['ms_per_step_min'], j['config']['launch_probe'], j['final_loss'])"
done
done
