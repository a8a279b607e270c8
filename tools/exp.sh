#!/bin/bash
# scratch experiment: 128-tiles in the grouped weight-gradient GEMM (LSTM-768)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_transformer.py -x -q -m gpu > gpurun_out/exp_pytest.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/exp_pytest.log | tail -3
for round in 1 2; do
for v in g128 g128b; do
LIPREADING_HIP_LIB=$(pwd)/lipreading_amd/_lib/alt/$v.so timeout 600 python bench.py --regime landmarks --model lstm768 --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$v', j['value'], j['ms_per_step'], j['timing']['ms_per_step_min'], j['pair_errors'], j['final_loss'])"
done
done
