for dbg in 0 1 2 3 7; do
LR_CLUSTER_DBG=$dbg timeout 600 python bench.py --regime landmarks --model lstm768 --no-cpu-baseline --repeats 1 --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dbg $dbg', d['ms_per_step'], {k:v for k,v in d['roofline']['avg_launch_us_by_kernel'].items() if 'cluster' in k})"
done
