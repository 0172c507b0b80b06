#!/bin/bash
# one iteration on the wide regime: parity tests, per-step time of sampling with the default Unet() (A/B lines from $AB), time per (kernel, grid)
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_kernels.py tests/test_unet.py tests/test_t5.py -q -m gpu -k "wide or default_unet or preset or attention_bearing or flash or tokens or gemm or crossembed" -x 2>&1 | tail -3
[ -n "$CONVS" ] && bash tools/gpu_wide_conv_trace.sh
for e in "" $AB; do env $e timeout 300 python tools/gpu_wide_sample.py 16 25 2>&1 | tail -1 | sed "s/^/[$e] /"; done
bash tools/gpu_wide_trace.sh 2>&1 | head -${ROWS:-16}
