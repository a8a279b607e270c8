#!/usr/bin/env bash
# The reference's own shipped steps as bench lines (parity + cpu_baseline on each) and their per-kernel statistics:
#   config/train/attn/attention_type  BiLSTM-512 + CTC + CharDecodingStep (LSTM-1024, char_dim 256), batch 4 (and B=32)
#   config/defaults.txt               LSTM-700 unidirectional, enable_ctc False, decoder LSTM-700, char_dim 300, batch 4 (and 32)
#   gpurun --timeout 1500 -- 'bash tools/gpu_attn_lines.sh r04'
set -u
R=$PWD; TAG=${1:-r04}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A1="--regime landmarks_attn --model lstm512 --char-dim 256"
A2="--regime landmarks_attn --model lstm700uni --char-dim 300 --no-ctc"
for b in 32 4; do
  timeout 600 python bench.py $A1 --batch $b 2>$OUT/attn_err.log | tail -1 > $OUT/${TAG}_bench_attn_lstm512_b$b.json || tail -5 $OUT/attn_err.log
  timeout 600 python bench.py $A2 --batch $b 2>$OUT/attn_err.log | tail -1 > $OUT/${TAG}_bench_attn_lstm700uni_b$b.json || tail -5 $OUT/attn_err.log
done
kt() {
  local name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/kt_$name" -o kt -- python "$R/bench.py" "$@" --no-graph --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
  python tools/rocpd_summary.py "$(find "$OUT/kt_$name" -name '*.db' | head -1)" 40 > "$OUT/${TAG}_${name}_kernel_stats.txt"
  rm -rf "$OUT/kt_$name"
}
kt attn_lstm512 $A1
kt attn_lstm700uni $A2
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/${TAG}_bench_attn_*.json")):
  try:
    d = json.load(open(f))
    p = d.get("parity") or {}
    print(f.split("/")[-1], d["ms_per_step"], "ms", d["value"], "frames/s | parity", p.get("abs_diff"), p.get("ok"), "| cpu", (d.get("cpu_baseline") or {}).get("value"),
          "|", d["config"].get("decoder_recurrence"))
  except Exception as e:
    print(f, "unreadable", e); print(open("$OUT/attn_err.log").read()[-1500:])
PY
