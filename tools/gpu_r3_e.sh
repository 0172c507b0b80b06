#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
{
for v in 512 256; do
  echo "== SS_THREADS=$v"
  MINIMAGEN_HIP_LIB=$(pwd)/minimagen_amd/libminimagen_hip_t$v.so timeout 300 python tools/gpu_small_sampler_race.py 2>&1 | grep "idle\|under load\|Error\|error" | tail -4
done
} > $OUT/small_sampler_threads.log 2>&1
cat $OUT/small_sampler_threads.log
