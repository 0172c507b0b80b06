#!/bin/bash
# A/B of environment knobs on ONE box, back to back: one short bench.py line (+ per-launch breakdown of the SR stage) per variant.
# usage: bash tools/gpu_ab_env.sh TAG "NAME1:ENV1=V ENV2=V" "NAME2:..." ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); TAG=$1; shift; OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  env $envs timeout 400 python bench.py --steps ${AB_STEPS:-8} --warmup 2 --no-cpu-baseline --no-secondary --no-t5 --breakdown-out $OUT/bd_$name.json > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open("$OUT/bench_$name.json") if l.startswith("{")][-1])
    ue = j.get("unet_eval", {})
    print("$name", "value %.0f  sync %.0f  one_lane %.0f  graph_step %.4f ms  conv_only %.1f GB/s  levels %s" % (j["value"], j["value_no_pipeline"], j.get("value_one_lane", 0), ue.get("graph_step_ms", 0), ue["conv_only"]["alg_GBps"], {k: round(v["ms"] * 1e3, 1) for k, v in ue["conv_only"]["by_level"].items()}))
except Exception as e:
    print("$name failed", e, open("$OUT/bench_$name.err").read()[-800:])
PY
done
