#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
{
echo "== small sampler under load (rebuilt library)"; LOAD=crossembed timeout 300 python tools/gpu_small_sampler_race.py 2>&1 | grep "load:\|under load\|Error\|error" | tail -3
echo "== U-Net evaluations as victims"; timeout 600 python tools/gpu_concurrency_stress.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|bias.abs" | tail -12
echo "== same, reduced precision (bf16 storage)"; PRECISION=half timeout 600 python tools/gpu_concurrency_stress.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|bias.abs" | tail -12
echo "== pipelined vs synchronous sample()"; timeout 300 python tools/gpu_async_diag.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|bias.abs" | tail -8
} > $OUT/concurrency_stress.log 2>&1
cat $OUT/concurrency_stress.log
timeout 900 python bench.py --breakdown-out $OUT/bd_cascade.json > $OUT/bench_cascade.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench_cascade.log | cut -c1-2500
