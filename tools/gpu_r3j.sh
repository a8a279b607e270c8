#!/usr/bin/env bash
set -u
OUT=gpurun_out; mkdir -p $OUT
python tools/tr2_stamps.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_frontend.py -m gpu -q -x -k "stride1 or oracle or side_stream" > $OUT/r3j_pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/r3j_pytest.log | tail -5
timeout 600 python bench.py --regime pixels --no-cpu-baseline > $OUT/r3j_bench.log 2>&1
tail -1 $OUT/r3j_bench.log > $OUT/r3j_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3j_bench.json"))
r = d.get("roofline") or {}
print(d["value"], d["ms_per_step"], (d.get("timing") or {}).get("ms_per_step_min"), r.get("kernel"), r.get("avg_launch_us"), r.get("frac"), r.get("avg_launch_us_by_kernel"))
PY
