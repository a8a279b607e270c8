#!/usr/bin/env bash
# quick loop for conv-kernel experiments: frontend parity tests, then the pixel regime's bench line (per-kernel event times)
set -u
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-ab}
timeout 900 python -m pytest tests/test_gpu_frontend.py -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/${TAG}_pytest.log | tail -5
timeout 600 python bench.py --regime pixels --no-cpu-baseline > $OUT/${TAG}_bench.log 2>&1
tail -1 $OUT/${TAG}_bench.log > $OUT/${TAG}_bench.json
python - $TAG <<'PY'
import json, sys
d = json.load(open("gpurun_out/%s_bench.json" % sys.argv[1]))
r = d.get("roofline") or {}
print(d["value"], d["ms_per_step"], (d.get("timing") or {}).get("ms_per_step_min"), r.get("kernel"), r.get("avg_launch_us"), r.get("frac"))
print({k: round(v, 1) for k, v in (r.get("avg_launch_us_by_kernel") or {}).items()})
p = d.get("parity") or {}
print("parity", p.get("abs_diff"), p.get("greedy_strings_equal"), p.get("argmax_flips"))
PY
