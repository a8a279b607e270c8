#!/bin/bash
# scratch experiment: conv1 forward branch-free / weight gradient predicated: frontend tests, pixel A/B
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_frontend.py -x -q -m gpu > gpurun_out/exp_pytest.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/exp_pytest.log | tail -3
for round in 1 2; do
for v in pool1 c1c; do
LIPREADING_HIP_LIB=$(pwd)/lipreading_amd/_lib/alt/$v.so timeout 600 python bench.py --regime pixels --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); k = j['roofline']['avg_launch_us_by_kernel']; print('$v', j['value'], j['ms_per_step'], j['timing']['ms_per_step_min'], k.get('conv1_fwd'), k.get('conv1_wgrad'))"
done
done
