#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q --timeout 300 -k attention 2>&1 | tail -3
MINIMAGEN_ATTN_VARIANT=6 timeout 600 python -m pytest tests/test_unet.py tests/test_sampler.py -m gpu -q --timeout 300 2>&1 | tail -3
for AV in 3 6; do
  export MINIMAGEN_ATTN_VARIANT=$AV
  timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --breakdown-out $OUT/bd_av${AV}.json > $OUT/bench_av${AV}.log 2>&1
  python - <<PY
import json
r = json.loads(open("$OUT/bench_av${AV}.log").read().strip().splitlines()[-1])
print("attn_variant=$AV", round(r["value"]), "steps/s  SR eval ms", round(r["unet_eval"]["sum_kernel_ms"], 3), {k: round(v, 3) for k, v in r["unet_eval"]["by_kernel_ms"].items()})
PY
done
