set -u
timeout 600 python -m pytest -m gpu -q -x tests/test_gpu_frontend.py 2>&1 | tail -4
for f in 2 0 1 2 0; do
  echo "LIPREADING_FUSE_UNPOOL=$f"
  LIPREADING_FUSE_UNPOOL=$f bash tools/gpu_quick.sh "" 
done
