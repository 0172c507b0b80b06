#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
{
for l in crossembed cross_attn conv conv@256 conv@128 conv@64; do
  LOAD=$l timeout 300 python tools/gpu_small_sampler_race.py 2>&1 | grep "load:\|under load\|Error\|error" | tail -3
done
} > $OUT/small_sampler_loadkind.log 2>&1
cat $OUT/small_sampler_loadkind.log
