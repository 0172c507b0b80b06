#!/bin/bash
# A/B of one engine environment knob: tools/gpu_ab_env.sh NAME v1 v2 ...   (cascade + base bench lines per value)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
NAME=$1; shift
for V in "$@"; do
  export $NAME=$V
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/ab_c_$V.log 2>&1
  timeout 300 python bench.py --workload base64 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/ab_b_$V.log 2>&1
  python - <<PY
import json
r = json.loads(open("$OUT/ab_c_$V.log").read().strip().splitlines()[-1])
rb = json.loads(open("$OUT/ab_b_$V.log").read().strip().splitlines()[-1])
print("$NAME=$V cascade", round(r["value"]), "base", round(rb["value"]), "| SR step ms", round(r["unet_eval"]["graph_step_ms"], 3), r["unet_eval"]["by_kernel_ms"]["conv"].__round__(3), "| base step ms", round(rb["unet_eval"]["graph_step_ms"], 3), rb["unet_eval"]["by_kernel_ms"]["conv"].__round__(3))
PY
done
