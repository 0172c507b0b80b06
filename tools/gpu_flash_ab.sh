#!/bin/bash
# the multi-query self-attention launches of the default Unet() (tools/bench_flash.py: 4096 / 1024 / 256 tokens) per variant + the parity tests per variant
# (default: ping-pong kernel; MI_FLASH_MQ_QT=2 / 1: all waves in step, two / one query tile per wave; PREP=0: self-staging kernel)
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "flash" -x 2>&1 | tail -1
for q in 1 2; do MI_FLASH_MQ_QT=$q timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "flash" -x 2>&1 | tail -1; done
for hw in 4096 1024 256; do
  timeout 100 python tools/bench_flash.py 32 $hw 20 2>&1 | tail -1
  for q in 1 2; do MI_FLASH_MQ_QT=$q timeout 100 python tools/bench_flash.py 32 $hw 20 2>&1 | tail -1; done
  PREP=0 timeout 100 python tools/bench_flash.py 32 $hw 20 2>&1 | tail -1
done
timeout 600 python -m pytest tests/test_unet.py -q -m gpu -k "wide or default_unet or preset" -x 2>&1 | tail -2
timeout 300 python tools/gpu_wide_sample.py 16 25 2>&1 | tail -1
