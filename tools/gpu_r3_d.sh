#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
{
echo "== baseline library"; timeout 300 python tools/gpu_small_sampler_race.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|bias.abs" | tail -8
for v in 1 2 3; do
  echo "== SS_FENCE=$v (1: acquire at kernel start, 2: release at kernel end, 3: both)"
  MINIMAGEN_HIP_LIB=$(pwd)/minimagen_amd/libminimagen_hip_f$v.so timeout 300 python tools/gpu_small_sampler_race.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|bias.abs" | tail -4
done
} > $OUT/small_sampler_race.log 2>&1
cat $OUT/small_sampler_race.log
