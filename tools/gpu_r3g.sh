#!/usr/bin/env bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/r3g_pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/r3g_pytest.log | tail -12
timeout 900 python bench.py --no-cpu-baseline > $OUT/r3g_bench_default.log 2>&1
tail -1 $OUT/r3g_bench_default.log > $OUT/r3g_bench_default.json
for m in lstm768 lstm700 gru800; do
  timeout 300 python bench.py --regime landmarks --model $m --no-cpu-baseline > $OUT/r3g_bench_$m.log 2>&1
  tail -1 $OUT/r3g_bench_$m.log > $OUT/r3g_bench_$m.json
done
python - <<'PY'
import json
for m in ("default", "lstm768", "lstm700", "gru800"):
  try:
    d = json.load(open("gpurun_out/r3g_bench_%s.json" % m))
    print(m, d["value"], d["ms_per_step"], "faults", d.get("pair_errors"), (d.get("roofline") or {}).get("avg_launch_us_by_kernel"))
    for k, v in d.get("regimes", {}).items():
      print("  ", k, v["value"], v["ms_per_step"])
  except Exception as e:
    print(m, "unreadable", e); print(open("gpurun_out/r3g_bench_%s.log" % m).read()[-1500:])
PY
