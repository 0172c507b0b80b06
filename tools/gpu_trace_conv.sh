#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels.py tests/test_unet.py -m gpu -q --timeout 300 2>&1 | tail -2
export MINIMAGEN_HIP_LIB=$(pwd)/minimagen_amd/libminimagen_hip_trace.so
TRACE_FLAGS=0x200 python tools/trace_conv.py 2>&1 | grep -v amdgpu.ids
unset MINIMAGEN_HIP_LIB
for CM in 1 3; do
  export MINIMAGEN_CONV_MFMA=$CM
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --breakdown-out $OUT/bd_cm${CM}.json > $OUT/bench_cm${CM}.log 2>&1
  timeout 300 python bench.py --workload base64 --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown > $OUT/bench_base_cm${CM}.log 2>&1
  python - <<PY
import json
r = json.loads(open("$OUT/bench_cm${CM}.log").read().strip().splitlines()[-1])
rb = json.loads(open("$OUT/bench_base_cm${CM}.log").read().strip().splitlines()[-1])
print("conv_mfma=$CM cascade", round(r["value"]), "steps/s; base", round(rb["value"]), " SR eval ms", round(r["unet_eval"]["sum_kernel_ms"], 3), {k: round(v, 3) for k, v in r["unet_eval"]["by_kernel_ms"].items()})
rows = json.load(open("$OUT/bd_cm${CM}.json"))
print("  ", [(x["op"].replace("conv ", "")[:30], round(x["ms"] * 1e3, 1)) for x in rows if x["kernel"] == "conv"])
PY
done
timeout 600 python -m pytest tests/test_sampler.py -m gpu -q --timeout 300 2>&1 | tail -2
