#!/usr/bin/env bash
# round 3, first box visit: the new recurrence tests first (fail fast), then the rest of the suite, then the
# landmark regimes' bench lines.
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q -s -k "cluster or time_out or matches_oracle or split" > $OUT/r3a_enc.log 2>&1
echo "enc exit $?"; grep -E "cluster vs|passed|failed|FAILED|Error|error" $OUT/r3a_enc.log | tail -30
timeout 600 python -m pytest tests/test_gpu_decoder.py -m gpu -x -q -s > $OUT/r3a_dec.log 2>&1
echo "dec exit $?"; grep -E "cluster vs|passed|failed|FAILED|Error" $OUT/r3a_dec.log | tail -10
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_decoder.py > $OUT/r3a_all.log 2>&1
echo "all exit $?"; grep -E "passed|failed|FAILED|Error" $OUT/r3a_all.log | tail -15
for m in lstm768 gru256; do
  timeout 300 python bench.py --regime landmarks --model $m --no-cpu-baseline > $OUT/r3a_bench_$m.log 2>&1
  tail -1 $OUT/r3a_bench_$m.log > $OUT/r3a_bench_$m.json
done
timeout 300 python bench.py --regime landmarks_attn --no-cpu-baseline > $OUT/r3a_bench_attn.log 2>&1
tail -1 $OUT/r3a_bench_attn.log > $OUT/r3a_bench_attn.json
python - <<'PY'
import json
for m in ("lstm768", "gru256", "attn"):
  try:
    d = json.load(open("gpurun_out/r3a_bench_%s.json" % m))
    for k, v in d.get("regimes", {}).items():
      r = v.get("roofline") or {}
      print(m, k, v["ms_per_step"], (v.get("parity") or {}).get("abs_diff"), r.get("avg_launch_us_by_kernel"), v.get("pair_errors"))
  except Exception as e:
    print(m, "unreadable", e)
    print(open("gpurun_out/r3a_bench_%s.log" % m).read()[-1500:])
PY
