#!/bin/bash
# A/B of the matrix-core conv workgroup shape (4 waves x 8 pixel-tiles vs 8 x 4) + the GPU parity tests of the conv family
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q --timeout 300 -k "conv" 2>&1 | tail -3
for W8 in 0 1; do
  export MINIMAGEN_CONV_WAVES8=$W8
  timeout 600 python -m pytest tests/test_unet.py tests/test_sampler.py -m gpu -q --timeout 300 2>&1 | tail -2
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --breakdown-out $OUT/bd_w8_${W8}.json > $OUT/bench_w8_${W8}.log 2>&1
  timeout 300 python bench.py --workload base64 --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown > $OUT/bench_base_w8_${W8}.log 2>&1
  python - <<PY
import json
r = json.loads(open("$OUT/bench_w8_${W8}.log").read().strip().splitlines()[-1])
rb = json.loads(open("$OUT/bench_base_w8_${W8}.log").read().strip().splitlines()[-1])
print("waves8=$W8 cascade", round(r["value"]), "steps/s; base", round(rb["value"]), " SR eval ms", round(r["unet_eval"]["sum_kernel_ms"], 3), {k: round(v, 3) for k, v in r["unet_eval"]["by_kernel_ms"].items()})
rows = json.load(open("$OUT/bd_w8_${W8}.json"))
print("  ", [(x["op"].replace("conv k3s1 ", "")[:28], round(x["ms"] * 1e3, 1)) for x in rows if "@64x64" in x["op"]][:8])
PY
done
