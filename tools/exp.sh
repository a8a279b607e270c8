timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_train.py tests/test_gpu_decoder.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --regime landmarks --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('landmarks gru256', d['ms_per_step'], d['pair_errors'])"
timeout 600 python bench.py --regime landmarks --model lstm768 --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('landmarks lstm768', d['ms_per_step'], d['pair_errors'], {k:v['ms_per_step'] for k,v in d.get('other_recurrences',{}).items()})"
