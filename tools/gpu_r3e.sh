#!/usr/bin/env bash
# where do the +130 us of the fp32-faithful recurrence in the pixel step come from?
set -u
OUT=gpurun_out; mkdir -p $OUT
R=$PWD
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_train.py -m gpu -q -s -k "pixel_pipeline" > $OUT/r3e_pipeline.log 2>&1
echo "pipeline exit $?"; grep -E "pixel pipeline|passed|failed|^E " $OUT/r3e_pipeline.log | tail -12
for rec in split bf16; do for ov in 1 0; do
  LIPREADING_RECURRENCE=$rec LIPREADING_OVERLAP_WGRAD=$ov timeout 300 python bench.py --regime pixels --no-cpu-baseline --repeats 3 > $OUT/r3e_px_${rec}_$ov.log 2>&1
  tail -1 $OUT/r3e_px_${rec}_$ov.log > $OUT/r3e_px_${rec}_$ov.json
done; done
for rec in split bf16; do
  (cd /tmp && LIPREADING_RECURRENCE=$rec rocprofv3 --kernel-trace -d "$R/$OUT/tl_$rec" -o kt -- python "$R/bench.py" --regime pixels --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
  python tools/rocpd_timeline.py "$(find $OUT/tl_$rec -name '*.db' | head -1)" conv1_fwd 8 > $OUT/r3e_pixels_${rec}_timeline.txt
  rm -rf $OUT/tl_$rec
done
python - <<'PY'
import json
for rec in ("split", "bf16"):
  for ov in (1, 0):
    try:
      d = json.load(open("gpurun_out/r3e_px_%s_%d.json" % (rec, ov)))
      print(rec, "overlap", ov, d["ms_per_step"], d["config"].get("launch_probe"), d["config"]["launch"][:40])
    except Exception as e:
      print(rec, ov, "unreadable", e)
PY
