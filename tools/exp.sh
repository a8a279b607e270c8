#!/bin/bash
# scratch experiment: XCD-local exchange in the GRU-256 pair kernels
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "split or pair or oracle" -s > gpurun_out/exp_pytest.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|worst" gpurun_out/exp_pytest.log | tail -8
for round in 1 2; do
for v in c1c pairx; do
LIPREADING_HIP_LIB=$(pwd)/lipreading_amd/_lib/alt/$v.so timeout 600 python bench.py --regime landmarks --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); k = j['roofline']['avg_launch_us_by_kernel']; print('$v', j['value'], j['ms_per_step'], j['timing']['ms_per_step_min'], j['pair_errors'], {a: b for a, b in k.items() if 'pair' in a})"
done
done
