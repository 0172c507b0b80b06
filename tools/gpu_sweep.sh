#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for T64 in 2 0 1; do for SP in 0 1; do for T128 in 0 1; do
  export MINIMAGEN_TILE64=$T64 MINIMAGEN_CONV_SPLIT16=$SP MINIMAGEN_TILE128=$T128
  timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --breakdown-out $OUT/bd_t${T64}_s${SP}_u${T128}.json > $OUT/bench_t${T64}_s${SP}_u${T128}.log 2>&1
  python - <<PY
import json
r = json.loads(open("$OUT/bench_t${T64}_s${SP}_u${T128}.log").read().strip().splitlines()[-1])
print("tile64=$T64 split16=$SP tile128=$T128", round(r["value"]), "steps/s  SR eval ms", round(r["unet_eval"]["sum_kernel_ms"], 3), {k: round(v, 3) for k, v in r["unet_eval"]["by_kernel_ms"].items()})
PY
done; done; done
