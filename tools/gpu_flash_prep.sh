#!/bin/bash
# A/B of the prepared-K/V multi-query attention (MINIMAGEN_FLASH_KV_PREP) on the default Unet(): parity tests, per-step time, kernel stats
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_kernels.py tests/test_unet.py -q -m gpu -k "flash or wide or default_unet or preset" -x 2>&1 | tail -4
for v in 1 0; do MINIMAGEN_FLASH_KV_PREP=$v timeout 300 python tools/gpu_wide_sample.py 16 25 2>&1 | tail -1 | sed "s/^/KV_PREP=$v: /"; done
bash tools/gpu_wide_profile.sh 2>&1 | cut -c1-160
