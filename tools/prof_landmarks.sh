set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for m in gru256 lstm768; do
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/kt_$m" -o lm -- python "$R/bench.py" --regime landmarks --model $m --no-graph --steps 15 --warmup 2 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/rocpd_summary.py "$(find "$OUT/kt_$m" -name '*.db' | head -1)" > "$OUT/r02_${m}_kernel_stats.txt"
rm -rf "$OUT/kt_$m"
head -40 "$OUT/r02_${m}_kernel_stats.txt"
done
