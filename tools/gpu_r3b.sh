#!/usr/bin/env bash
# round 3, second visit: cluster tests with the 16-byte backward exchange, the decoder on the cluster kernels, and
# the landmark regimes' bench lines on the reference's shapes.
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q -s -k "cluster or time_out or matches_oracle or split" > $OUT/r3b_enc.log 2>&1
echo "enc exit $?"; grep -E "cluster vs|passed|failed|FAILED|Error|error" $OUT/r3b_enc.log | tail -24
timeout 600 python -m pytest tests/test_gpu_decoder.py -m gpu -x -q -s > $OUT/r3b_dec.log 2>&1
echo "dec exit $?"; grep -E "cluster vs|passed|failed|FAILED|Error" $OUT/r3b_dec.log | tail -10
for m in lstm768 lstm700 lstm512 gru800 gru256; do
  timeout 300 python bench.py --regime landmarks --model $m --no-cpu-baseline > $OUT/r3b_bench_$m.log 2>&1
  tail -1 $OUT/r3b_bench_$m.log > $OUT/r3b_bench_$m.json
done
LIPREADING_RNN_DEBUG=2 timeout 300 python bench.py --regime landmarks --model gru256 --no-cpu-baseline > $OUT/r3b_bench_gru256c.log 2>&1
tail -1 $OUT/r3b_bench_gru256c.log > $OUT/r3b_bench_gru256c.json
timeout 300 python bench.py --regime landmarks_attn --no-cpu-baseline > $OUT/r3b_bench_attn.log 2>&1
tail -1 $OUT/r3b_bench_attn.log > $OUT/r3b_bench_attn.json
python - <<'PY'
import json
for m in ("lstm768", "lstm700", "lstm512", "gru800", "gru256", "gru256c", "attn"):
  try:
    d = json.load(open("gpurun_out/r3b_bench_%s.json" % m))
    r = d.get("roofline") or {}
    print(m, d["ms_per_step"], d.get("pair_errors"), r.get("us_per_step_by_direction"), r.get("avg_launch_us_by_kernel"),
          {k: v["ms_per_step"] for k, v in (d.get("other_recurrences") or {}).items()})
  except Exception as e:
    print(m, "unreadable", e)
    print(open("gpurun_out/r3b_bench_%s.log" % m).read()[-1500:])
PY
