#!/bin/bash
# A/B of the kernel-shape knobs on the GPU box: per-launch breakdown of both stages for each setting.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
for AV in 0 1; do for SP in 0 1; do
  export MINIMAGEN_ATTN_VARIANT=$AV MINIMAGEN_CONV_SPLIT16=$SP
  timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --breakdown-out $OUT/bd_cascade_a${AV}_s${SP}.json > $OUT/bench_cascade_a${AV}_s${SP}.log 2>&1
  timeout 300 python bench.py --workload base64 --steps 2 --warmup 1 --no-cpu-baseline --breakdown-out $OUT/bd_base_a${AV}_s${SP}.json > $OUT/bench_base_a${AV}_s${SP}.log 2>&1
  echo "== attn_variant=$AV split16=$SP"; tail -1 $OUT/bench_cascade_a${AV}_s${SP}.log | cut -c1-250; tail -1 $OUT/bench_base_a${AV}_s${SP}.log | cut -c1-250
done; done
