#!/bin/bash
# scratch experiment: fused projection head + pack-kernel clears + cached ones: full suite, benches
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/exp_pytest.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|Error" gpurun_out/exp_pytest.log | tail -5
for rg in landmarks pixels; do
for v in pairx projf; do
LIPREADING_HIP_LIB=$(pwd)/lipreading_amd/_lib/alt/$v.so timeout 600 python bench.py --regime $rg --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$rg $v', j['value'], j['ms_per_step'], j['timing']['ms_per_step_min'], j['pair_errors'], j['final_loss'])"
done
done
