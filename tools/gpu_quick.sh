# GPU suite + step timings of both stages + the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -m gpu -q -x 2>&1 | tail -2
python bench.py --workload base64 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); u=d['unet_eval']; print('base64', round(d['value']), 'steps/s', round(d['ms_per_step'],2), 'ms/call; graph step', round(u.get('graph_step_ms',0),4))"
python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); u=d['unet_eval']; print('cascade', round(d['value']), 'steps/s', round(d['ms_per_step'],2), 'ms/call; SR graph step', round(u.get('graph_step_ms',0),4), {k: round(v,4) for k,v in u['by_kernel_ms'].items()})"
