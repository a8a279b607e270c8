#!/usr/bin/env bash
# round 3, fourth visit: the files of the -m gpu suite the third visit did not reach, the time-out tests again (fixed:
# no asm load left in flight on a thread that has given up), the default bench line with the parity variants, and a
# kernel trace of the BiLSTM-768 landmark step.
set -u
OUT=gpurun_out; mkdir -p $OUT
R=$PWD
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q -k "time_out" > $OUT/r3d_timeout.log 2>&1
echo "timeout tests exit $?"; tail -3 $OUT/r3d_timeout.log
timeout 1500 python -m pytest tests -m gpu -q -s --durations=12 > $OUT/r3d_pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|pixel pipeline|pixel parity|cluster vs step" $OUT/r3d_pytest.log | tail -40
timeout 900 python bench.py > $OUT/r3d_bench_default.log 2>&1
tail -1 $OUT/r3d_bench_default.log > $OUT/r3d_bench_default.json
LIPREADING_INPUT_PROJECTION=bf16x1 timeout 400 python bench.py --regime pixels --no-cpu-baseline > $OUT/r3d_bench_x1.log 2>&1
tail -1 $OUT/r3d_bench_x1.log > $OUT/r3d_bench_x1.json
LIPREADING_RECURRENCE=bf16 timeout 400 python bench.py --regime pixels --no-cpu-baseline > $OUT/r3d_bench_recbf16.log 2>&1
tail -1 $OUT/r3d_bench_recbf16.log > $OUT/r3d_bench_recbf16.json
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$R/$OUT/kt_lstm768" -o kt -- python "$R/bench.py" --regime landmarks --model lstm768 --no-graph --steps 15 --warmup 2 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/rocpd_summary.py "$(find $OUT/kt_lstm768 -name '*.db' | head -1)" 40 > $OUT/r3d_lstm768_kernel_stats.txt; rm -rf $OUT/kt_lstm768
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$R/$OUT/kt_attn" -o kt -- python "$R/bench.py" --regime landmarks_attn --no-graph --steps 15 --warmup 2 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/rocpd_summary.py "$(find $OUT/kt_attn -name '*.db' | head -1)" 40 > $OUT/r3d_attn_kernel_stats.txt; rm -rf $OUT/kt_attn
python - <<'PY'
import json
for m in ("default", "x1", "recbf16"):
  try:
    d = json.load(open("gpurun_out/r3d_bench_%s.json" % m))
    print(m, d["value"], d["ms_per_step"], "faults", d.get("pair_errors"))
    p = d.get("parity") or {}
    print("   parity", {k: p.get(k) for k in ("abs_diff", "greedy_strings_equal", "flip_fraction", "max_abs_log_prob_diff", "loss_oracle_fp32conv", "abs_diff_vs_fp32conv_oracle")})
    for k, v in (p.get("other_paths_vs_the_same_oracle") or {}).items():
      print("     ", k, {q: v.get(q) for q in ("abs_diff", "argmax_flips", "max_abs_log_prob_diff", "greedy_strings_equal")})
    print("   trained", p.get("after_training"))
    print("   cpu", {k: (d.get("cpu_baseline") or {}).get(k) for k in ("value", "full_batch_step")})
    for k, v in d.get("regimes", {}).items():
      print("  ", k, v["value"], v["ms_per_step"], (v.get("parity") or {}).get("abs_diff"), (v.get("parity") or {}).get("greedy_strings_equal"))
  except Exception as e:
    print(m, "unreadable", e); print(open("gpurun_out/r3d_bench_%s.log" % m).read()[-2500:])
PY
head -30 $OUT/r3d_lstm768_kernel_stats.txt
