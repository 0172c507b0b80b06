cd "${GRAFT_REPO_ROOT:-/root/repo}"
for b in 8 16 32 64; do python bench.py --workload base64 --batch $b --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-pipeline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); u=d['unet_eval']; print('B=$b', round(d['value']), 'steps/s', round(d['ms_per_step'],2), 'ms/call; graph step', round(u.get('graph_step_ms',0),4), {k: round(v,4) for k,v in u['by_kernel_ms'].items()})"; done
