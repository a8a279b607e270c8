timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_train.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --regime landmarks_attn --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('landmarks_attn', d['ms_per_step'], d['pair_errors'])"
