#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
{
echo "== baseline"; LOAD=crossembed timeout 300 python tools/gpu_small_sampler_race.py 2>&1 | grep "load:\|under load\|Error\|error" | tail -3
echo "== sampler.hip built with -fno-slp-vectorize (no v_pk_*_f32)"; MINIMAGEN_HIP_LIB=$(pwd)/minimagen_amd/libminimagen_hip_noslp.so LOAD=crossembed timeout 300 python tools/gpu_small_sampler_race.py 2>&1 | grep "load:\|under load\|Error\|error" | tail -3
MINIMAGEN_HIP_LIB=$(pwd)/minimagen_amd/libminimagen_hip_noslp.so LOAD=all timeout 300 python tools/gpu_small_sampler_race.py 2>&1 | grep "load:\|under load\|Error\|error" | tail -3
} > $OUT/small_sampler_noslp.log 2>&1
cat $OUT/small_sampler_noslp.log
