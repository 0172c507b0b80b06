#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
./tools/ubench/valu_rate > $OUT/valu_rate.txt 2>&1; cat $OUT/valu_rate.txt
timeout 600 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
export MINIMAGEN_ATTN_VARIANT=1 MINIMAGEN_CONV_SPLIT16=1
for V in "" _fastsilu _noslp _both; do
  export MINIMAGEN_HIP_LIB=$(pwd)/minimagen_amd/libminimagen_hip$V.so
  timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --breakdown-out $OUT/bd_cascade_lib$V.json > $OUT/bench_cascade_lib$V.log 2>&1
  echo "== lib$V"; tail -1 $OUT/bench_cascade_lib$V.log | cut -c1-200
  timeout 300 python -m pytest tests/test_unet.py tests/test_sampler.py -m gpu -q --timeout 300 2>&1 | tail -2
done
