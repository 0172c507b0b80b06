#!/bin/bash
# A/B of environment knobs on the GPU box: usage  bash tools/gpu_env_ab.sh "" "KNOB=1" "KNOB=2 OTHER=3" ...   (one bench.py run per argument)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out; mkdir -p $OUT
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 400 python bench.py --no-cpu-baseline --no-secondary --no-t5 --breakdown-out $OUT/env_bd_$i.json > $OUT/env_bench_$i.log 2>&1
  python - <<PY
import json
d=json.loads(open("$OUT/env_bench_$i.log").read().strip().splitlines()[-1])
r=d["roofline"]
print("[$e]", round(d["value"]), "one lane", round(d.get("value_one_lane",0)), "sync", round(d["value_no_pipeline"]), "| graph step", round(d["unet_eval"]["graph_step_ms"], 4), "ms |", r["kernel"], round(r["kernel_ms"]*1e3,1), "us")
PY
done
