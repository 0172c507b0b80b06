#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
i=0
for e in "" "" "MINIMAGEN_SAMPLER_GROUP=0" "MINIMAGEN_CONV_REVERSE=0"; do
  i=$((i+1))
  env $e timeout 400 python bench.py --workload cascade64_256_1024 --batch 8 --precision half --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-t5 --no-breakdown > $OUT/c5_$i.log 2>&1
  echo "[$e] rc=$? $(tail -1 $OUT/c5_$i.log | cut -c1-420)"
done
