#!/bin/bash
# A/B of the prepared-K/V multi-query attention on the default Unet(): parity tests, per-step time per variant
# (MINIMAGEN_FLASH_KV_PREP=0: self-staging kernel; MI_FLASH_MQ_QT: 16-query tiles per wave, 3 = two tiles at four waves per SIMD), kernel stats
R=$GRAFT_REPO_ROOT
cd $R
for q in 1 2 3 4; do MI_FLASH_MQ_QT=$q timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "flash" -x 2>&1 | tail -1; done
timeout 600 python -m pytest tests/test_unet.py -q -m gpu -k "wide or default_unet or preset" -x 2>&1 | tail -2
MINIMAGEN_FLASH_KV_PREP=0 timeout 300 python tools/gpu_wide_sample.py 16 25 2>&1 | tail -1 | sed "s/^/KV_PREP=0: /"
for q in 1 2 3 4; do MI_FLASH_MQ_QT=$q timeout 300 python tools/gpu_wide_sample.py 16 25 2>&1 | tail -1 | sed "s/^/QT=$q: /"; done
bash tools/gpu_wide_profile.sh 2>&1 | cut -c1-160
