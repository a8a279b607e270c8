set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_frontend.py -m gpu -q -x 2>&1 | tail -3
for v in main c1w768 c1old main; do
  if [ $v != main ]; then export LIPREADING_HIP_LIB=$GRAFT_REPO_ROOT/lipreading_amd/_lib/alt/$v.so; else unset LIPREADING_HIP_LIB; fi
  echo "== $v"; timeout 120 python tools/bench_conv1.py 30 2>&1 | tail -2
done
