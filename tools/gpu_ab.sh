#!/bin/bash
# quick A/B: GPU kernel+unet parity, then cascade bench with the given env settings (one per argument, "-" = defaults)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels.py tests/test_unet.py -m gpu -q -x --timeout 300 2>&1 | tail -2
i=0
for SET in "$@"; do
  i=$((i+1)); name=ab$i
  if [ "$SET" = "-" ]; then ENVS=""; else ENVS="$SET"; fi
  env $ENVS timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --breakdown-out $OUT/bd_$name.json > $OUT/bench_$name.log 2>&1
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/bench_$name.log").read().strip().splitlines()[-1])
    print("[$SET] cascade", round(r["value"]), "steps/s | SR step ms", round(r["unet_eval"]["graph_step_ms"], 3), "hbm", round(r["unet_eval"]["hbm_frac_graph_step"], 3), {k: round(v, 3) for k, v in r["unet_eval"]["by_kernel_ms"].items()})
except Exception as e:
    print("[$SET] FAILED", e); print(open("$OUT/bench_$name.log").read()[-1500:])
PY
done
