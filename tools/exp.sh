set -u
R=$PWD; OUT=$R/gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_transformer.py -x -q 2>&1 | tail -3
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/kt_t" -o kt -- python "$R/bench.py" --regime pixels_tfm --no-graph --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/rocpd_summary.py "$(find "$OUT/kt_t" -name '*.db' | head -1)" 60 > "$OUT/r02_pixels_tfm_kernel_stats.txt"; rm -rf "$OUT/kt_t"
grep -n "attn\|sgemm\|xgemm\|layernorm\|all kernels" "$OUT/r02_pixels_tfm_kernel_stats.txt" | cut -c1-150
