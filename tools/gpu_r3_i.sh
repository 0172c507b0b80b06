#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
python tools/wide_unet_forward.py 16 2>&1 | grep -v amdgpu.ids | tail -1
MI_GEMM_EXACT_F32=1 python tools/wide_unet_forward.py 16 2>&1 | grep -v amdgpu.ids | tail -1
timeout 900 python bench.py --breakdown-out $OUT/bd_cascade.json > $OUT/bench_cascade.log 2>&1; echo "bench rc=$?"; python - <<PY
import json
j = json.loads([l for l in open("$OUT/bench_cascade.log").read().splitlines() if l.startswith("{")][-1])
print({k: j[k] for k in ("value", "ms_per_step", "value_no_pipeline", "pipelined_equals_synchronous")})
print("t5", j["t5_encode"]["ms"], "roofline", j["roofline"]["frac"], j["roofline"]["kernel_ms"], "graph_step", j["unet_eval"]["graph_step_ms"], j["unet_eval"]["hbm_frac_graph_step"])
print({k: (v.get("denoising_steps_per_s"), v.get("denoising_steps_per_s_no_pipeline"), v.get("error")) for k, v in j["secondary"].items()})
print(j["cpu_baseline"]["value"], j["cpu_baseline"]["kind"])
PY
