#!/bin/bash
# scratch experiment: all 96 weight fragments of the persistent GRU forward kernel in registers
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "persist or bf16" > gpurun_out/exp_pytest.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/exp_pytest.log | tail -3
for round in 1 2; do
for v in head pks8; do
LIPREADING_HIP_LIB=$(pwd)/lipreading_amd/_lib/alt/$v.so timeout 600 python bench.py --regime pixels --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); k = j['roofline']['avg_launch_us_by_kernel']; print('$v', j['value'], j['ms_per_step'], j['timing']['ms_per_step_min'], {a: b for a, b in k.items() if 'persist' in a})"
done
done
