#!/usr/bin/env bash
# round 5, fifth visit: fgemm launch-time anatomy; the folded skip / fault words of the data-parallel step
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python tools/bench_fgemm.py > $OUT/r05e_fgemm.txt 2>&1; cat $OUT/r05e_fgemm.txt | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_two_ranks.py tests/test_gpu_train.py -q -x > $OUT/r05e_pytest.log 2>&1
echo "pytest exit $?"; tail -5 $OUT/r05e_pytest.log
for d in 0 1; do
LIPREADING_BENCH_FORCE_DIST=$d timeout 300 python bench.py --regime both --no-cpu-baseline 2>$OUT/r05e_dist$d.err | tail -1 > $OUT/r05e_both_dist$d.json
python -c "
import json; d=json.load(open('$OUT/r05e_both_dist$d.json')); print('dist$d pixels', d['ms_per_step'], 'landmarks', d['regimes']['landmarks']['ms_per_step'])" || tail -5 $OUT/r05e_dist$d.err
done
LIPREADING_BENCH_FORCE_DIST=1 bash tools/gpu_timeline.sh r05e_lmdist step_begin --regime landmarks --model gru256 > /dev/null
cat $OUT/r05e_lmdist_step_timeline.txt | cut -c1-100
