#!/bin/bash
# cross-attention variants A/B: MINIMAGEN_ATTN_SWP = 0 (round-2 kernel), 1 / 2 / 3 (software-pipelined at 4 / 3 / 2 waves per SIMD)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -x -k "cross_attention" 2>&1 | tail -3
for m in 0 1 2 3; do
  for wl in cascade64_256 base64; do
    MINIMAGEN_ATTN_SWP=$m timeout 300 python bench.py --workload $wl --steps 4 --no-cpu-baseline --no-secondary --no-t5 --no-pipeline --breakdown-out $OUT/bd_attn_${m}_$wl.json > $OUT/bench_attn_${m}_$wl.log 2>&1
    python - <<PY
import json
rows = json.load(open("$OUT/bd_attn_${m}_$wl.json"))
line = json.loads(open("$OUT/bench_attn_${m}_$wl.log").read().strip().splitlines()[-1])
at = [r["ms"] * 1e3 for r in rows if r["kernel"] == "cross_attn"]
print("SWP=$m $wl: cross_attn launches %s us; graph step %.4f ms; value(no pipeline) %.0f" % (["%.1f" % a for a in at], line["unet_eval"].get("graph_step_ms", 0), line["value"]))
PY
  done
done
