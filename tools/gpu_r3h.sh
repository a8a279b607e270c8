#!/usr/bin/env bash
# second form of the stride-1 weight gradient: parity tests, then the pixel regime with the first (0) and second form
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_frontend.py -m gpu -q -x --durations=5 > $OUT/r3h_pytest.log 2>&1
echo "pytest exit $?"; grep -E "^(FAILED|ERROR)|passed|failed|Error|error" $OUT/r3h_pytest.log | tail -12
for v in 0 2; do
  LIPREADING_CONV_WGRAD_TR=$v timeout 600 python bench.py --regime pixels --no-cpu-baseline > $OUT/r3h_bench_v$v.log 2>&1
  tail -1 $OUT/r3h_bench_v$v.log > $OUT/r3h_bench_v$v.json
done
python - <<'PY'
import json
for v in (0, 2):
  try:
    d = json.load(open("gpurun_out/r3h_bench_v%d.json" % v))
    r = d.get("roofline") or {}
    print(v, d["value"], d["ms_per_step"], (d.get("timing") or {}).get("ms_per_step_min"), r.get("kernel"), r.get("avg_launch_us"), r.get("frac"), r.get("avg_launch_us_by_kernel"))
    p = d.get("parity") or {}
    print("   parity", p.get("abs_diff"), p.get("greedy_strings_equal"), p.get("argmax_flips"))
  except Exception as e:
    print(v, "unreadable", e); print(open("gpurun_out/r3h_bench_v%d.log" % v).read()[-1500:])
PY
export TMPDIR=/tmp
(cd /tmp && LIPREADING_CONV_WGRAD_SIDE=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$OUT/r3h_prof" -o kt -- \
  python "$GRAFT_REPO_ROOT/bench.py" --regime pixels --no-graph --no-cpu-baseline --steps 10 --repeats 1 > "$GRAFT_REPO_ROOT/$OUT/r3h_prof.log" 2>&1)
python tools/rocpd_summary.py "$(find "$OUT/r3h_prof" -name '*.db' | head -1)" 60 > $OUT/r3h_pixels_kernel_stats.txt
grep -i "conv\|unpool" $OUT/r3h_pixels_kernel_stats.txt | cut -c1-160 | head -20
rm -rf $OUT/r3h_prof
