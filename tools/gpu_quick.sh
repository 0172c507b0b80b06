#!/bin/bash
# quick GPU loop: kernel + U-Net + sampler parity, then the cascade / base bench lines with the SR-stage breakdown
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_generate.py -m gpu -q --timeout 600 2>&1 | tail -1
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --breakdown-out $OUT/bd_q.json > $OUT/bench_q.log 2>&1
timeout 300 python bench.py --workload base64 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --breakdown-out $OUT/bd_qb.json > $OUT/bench_qb.log 2>&1
timeout 300 python bench.py --precision half --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --breakdown-out $OUT/bd_qh.json > $OUT/bench_qh.log 2>&1
python -c "import json; r=json.loads(open(\"$OUT/bench_qh.log\").read().strip().splitlines()[-1]); print(\"HALF cascade\", round(r[\"value\"]), r[\"unet_eval\"][\"by_kernel_ms\"], r[\"roofline\"][\"frac\"])"
python - <<PY
import json
r = json.loads(open("$OUT/bench_q.log").read().strip().splitlines()[-1])
rb = json.loads(open("$OUT/bench_qb.log").read().strip().splitlines()[-1])
print("cascade", round(r["value"]), "steps/s; base", round(rb["value"]), " SR eval ms", round(r["unet_eval"]["sum_kernel_ms"], 3), {k: round(v, 3) for k, v in r["unet_eval"]["by_kernel_ms"].items()}, "hbm_frac", round(r["unet_eval"]["hbm_frac_whole_forward"], 3))
print("base eval ms", round(rb["unet_eval"]["sum_kernel_ms"], 3), {k: round(v, 3) for k, v in rb["unet_eval"]["by_kernel_ms"].items()}, "hbm_frac", round(rb["unet_eval"]["hbm_frac_whole_forward"], 3))
print("graph_step_ms", r["unet_eval"].get("graph_step_ms"), "hbm_frac_graph_step", r["unet_eval"].get("hbm_frac_graph_step"))
PY
