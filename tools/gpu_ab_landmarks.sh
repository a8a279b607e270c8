#!/usr/bin/env bash
set -u
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-ab}
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_ctc.py tests/test_gpu_train.py -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/${TAG}_pytest.log | tail -5
for m in gru256 lstm768; do
  timeout 300 python bench.py --regime landmarks --model $m --no-cpu-baseline > $OUT/${TAG}_bench_$m.log 2>&1
  tail -1 $OUT/${TAG}_bench_$m.log > $OUT/${TAG}_bench_$m.json
done
python - $TAG <<'PY'
import json, sys
for m in ("gru256", "lstm768"):
  d = json.load(open("gpurun_out/%s_bench_%s.json" % (sys.argv[1], m)))
  p = d.get("parity") or {}
  print(m, d["value"], d["ms_per_step"], (d.get("timing") or {}).get("ms_per_step_min"), "parity", p.get("abs_diff"), p.get("greedy_strings_equal"))
PY
