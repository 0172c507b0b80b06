# CrossEmbed on the matrix cores: kernel tests on the device, then the step timing (MINIMAGEN_HIP_LIB variants side by side)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_kernels.py -q -x -k "crossembed and gpu" 2>&1 | tail -2
run() { echo "== $*"; env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); u=d['unet_eval']; print(round(d['value']), 'steps/s; SR graph step', round(u.get('graph_step_ms',0),4), 'ms; by kernel', {k: round(v,4) for k,v in u['by_kernel_ms'].items()})"; }
for lib in $LIBS; do run MINIMAGEN_HIP_LIB=$(pwd)/minimagen_amd/$lib; done
