#!/bin/bash
# scratch experiment: A/B of two library builds on the pixel regime, alternating
for round in 1 2 3; do
for v in e135c3 cur; do
LIPREADING_HIP_LIB=$(pwd)/lipreading_amd/_lib/alt/$v.so timeout 600 python bench.py --regime pixels --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$v', j['value'], j['ms_per_step'], j['timing']['ms_per_step_min'])"
done
done
