#!/bin/bash
# one iteration on the wide regime: parity tests, standalone conv times + phase traces, per-step time of sampling with the default Unet()
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_kernels.py tests/test_unet.py -q -m gpu -k "wide or default_unet or preset" -x 2>&1 | tail -3
bash tools/gpu_wide_conv_trace.sh
timeout 300 python tools/gpu_wide_sample.py 16 25 2>&1 | tail -1
