#!/bin/bash
# A/B of library build variants: tools/gpu_ab_lib.sh "" _variant ...   (suffixes of minimagen_amd/libminimagen_hip<suffix>.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
for V in "$@"; do
  if [ "$V" = "-" ]; then V=""; fi
  export MINIMAGEN_HIP_LIB=$(pwd)/minimagen_amd/libminimagen_hip$V.so
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/ab_c$V.log 2>&1
  python - <<PY
import json
r = json.loads(open("$OUT/ab_c$V.log").read().strip().splitlines()[-1])
print("lib$V cascade", round(r["value"]), "| SR step ms", round(r["unet_eval"]["graph_step_ms"], 3), {k: round(v, 3) for k, v in r["unet_eval"]["by_kernel_ms"].items()})
PY
done
