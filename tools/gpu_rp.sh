#!/bin/bash
# rp-conv loop: kernel parity on the GPU, then cascade bench lines with A/B knobs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels.py tests/test_unet.py -m gpu -q -x --timeout 300 2>&1 | tail -3
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --breakdown-out $OUT/bd_$name.json > $OUT/bench_$name.log 2>&1
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/bench_$name.log").read().strip().splitlines()[-1])
    print("$name cascade", round(r["value"]), "steps/s | SR step ms", round(r["unet_eval"]["graph_step_ms"], 3), "hbm", round(r["unet_eval"]["hbm_frac_graph_step"], 3), {k: round(v, 3) for k, v in r["unet_eval"]["by_kernel_ms"].items()})
except Exception as e:
    print("$name FAILED", e); print(open("$OUT/bench_$name.log").read()[-1500:])
PY
}
run rp0 MINIMAGEN_CONV_RP=0
run rp1 MINIMAGEN_CONV_RP=1
run rp1_n1 MINIMAGEN_CONV_RP=1 MINIMAGEN_RP_NTILE=1
run rp1_n2 MINIMAGEN_CONV_RP=1 MINIMAGEN_RP_NTILE=2
run rp1_n4 MINIMAGEN_CONV_RP=1 MINIMAGEN_RP_NTILE=4
run rp1_S6 MINIMAGEN_CONV_RP=1 MINIMAGEN_RP_TILE_S=6
