#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
run() { echo "== $*"; env "${@:3}" timeout 200 python tools/gpu_race_bisect.py $1 $2 2>&1 | grep -v amdgpu.ids | tail -5; }
{
run sr 1
run torch 1
run sr 0
run sr 1 MINIMAGEN_SAMPLER_FUSED=0
run sr 1 MINIMAGEN_CONV_RP=0
run sr 1 MINIMAGEN_ATTN_VARIANT=0
run sr 1 MINIMAGEN_CE_MFMA=0
run sr 1 MINIMAGEN_STEPS_PER_GRAPH=1
} > $OUT/race_bisect.log 2>&1
cat $OUT/race_bisect.log
