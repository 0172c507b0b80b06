#!/bin/bash
# A/B of library build variants: tools/gpu_ab_lib.sh "" _variant ...   (suffixes of minimagen_amd/libminimagen_hip<suffix>.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
for V in "$@"; do
  export MINIMAGEN_HIP_LIB=$(pwd)/minimagen_amd/libminimagen_hip$V.so
  timeout 300 python -m pytest tests/test_kernels.py tests/test_unet.py -m gpu -q --timeout 300 -k "cross_attention or golden or half" 2>&1 | tail -1
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/ab_c$V.log 2>&1
  timeout 300 python bench.py --workload base64 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/ab_b$V.log 2>&1
  python - <<PY
import json
r = json.loads(open("$OUT/ab_c$V.log").read().strip().splitlines()[-1])
rb = json.loads(open("$OUT/ab_b$V.log").read().strip().splitlines()[-1])
print("lib$V cascade", round(r["value"]), "base", round(rb["value"]), "| SR step ms", round(r["unet_eval"]["graph_step_ms"], 3), "attn", round(r["unet_eval"]["by_kernel_ms"]["cross_attn"], 3), "| base step ms", round(rb["unet_eval"]["graph_step_ms"], 3), "attn", round(rb["unet_eval"]["by_kernel_ms"]["cross_attn"], 3))
PY
done
