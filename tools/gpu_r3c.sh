#!/usr/bin/env bash
# round 3, third visit: the whole -m gpu suite (new: two ranks on one GPU, pixel pipeline end to end, configs[2] at
# BiLSTM-768, decoder on the cluster kernels, forced time-outs), then the default bench line and the landmark lines.
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -s --durations=12 > $OUT/r3c_pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|pixel pipeline|cluster vs step" $OUT/r3c_pytest.log | tail -30
timeout 900 python bench.py > $OUT/r3c_bench_default.log 2>&1
tail -1 $OUT/r3c_bench_default.log > $OUT/r3c_bench_default.json
for m in lstm768 gru256; do
  timeout 400 python bench.py --regime landmarks --model $m > $OUT/r3c_bench_$m.log 2>&1
  tail -1 $OUT/r3c_bench_$m.log > $OUT/r3c_bench_$m.json
done
LIPREADING_RNN_DEBUG=4 timeout 300 python bench.py --regime landmarks --model lstm768 --no-cpu-baseline > $OUT/r3c_bench_lstm768_f32wgrad.log 2>&1
tail -1 $OUT/r3c_bench_lstm768_f32wgrad.log > $OUT/r3c_bench_lstm768_f32wgrad.json
LIPREADING_RNN_DEBUG=4 timeout 300 python bench.py --regime landmarks --model gru256 --no-cpu-baseline > $OUT/r3c_bench_gru256_f32wgrad.log 2>&1
tail -1 $OUT/r3c_bench_gru256_f32wgrad.log > $OUT/r3c_bench_gru256_f32wgrad.json
python - <<'PY'
import json
for m in ("default", "lstm768", "gru256", "lstm768_f32wgrad", "gru256_f32wgrad"):
  try:
    d = json.load(open("gpurun_out/r3c_bench_%s.json" % m))
    print(m, d["value"], d["ms_per_step"], "faults", d.get("pair_errors"))
    p = d.get("parity") or {}
    print("   parity", {k: p.get(k) for k in ("abs_diff", "greedy_strings_equal", "flip_fraction", "max_oracle_margin_at_flips", "max_abs_log_prob_diff", "flips_outside_2x_diff", "loss_oracle_fp32conv", "abs_diff_vs_fp32conv_oracle")})
    print("   split", p.get("recurrence_split")); print("   trained", p.get("after_training"))
    for k, v in d.get("regimes", {}).items():
      print("  ", k, v["value"], v["ms_per_step"], (v.get("parity") or {}).get("abs_diff"), (v.get("parity") or {}).get("greedy_strings_equal"))
  except Exception as e:
    print(m, "unreadable", e); print(open("gpurun_out/r3c_bench_%s.log" % m).read()[-2500:])
PY
