# step timing under environment-variable variants:  VARIANTS="A=1 B=2|C=3" bash tools/gpu_env_ab.sh   ('|' separates variants; '-' = defaults)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
IFS='|' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  [ "$v" = "-" ] && v=""
  echo "== ${v:-defaults}"
  env $v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); u=d['unet_eval']; print('  SR graph step', round(u.get('graph_step_ms',0),4), 'ms;', {k: round(v,4) for k,v in u['by_kernel_ms'].items()})"
  env $v python bench.py --workload base64 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); u=d['unet_eval']; print('  base graph step', round(u.get('graph_step_ms',0),4), 'ms;', {k: round(v,4) for k,v in u['by_kernel_ms'].items()})"
done
