#!/bin/bash
# standalone times of the wide convs of the default Unet(): row-paired wide regime (w6 / w7 / w10) against the wide GEMM kernel (w11);
# with the -DMI_TRACE build present also the per-workgroup phase cycles of the row-paired kernel
R=$GRAFT_REPO_ROOT; cd $R
T=$R/minimagen_amd/libminimagen_hip_trace.so
for spec in "32 512 512 16 16 1 id w10" "32 512 512 16 16 1 id w11" "32 768 512 16 16 1 none w10" "32 768 512 16 16 1 none w11" "32 256 256 32 32 1 id w7" "32 256 256 32 32 1 id w11" "32 128 128 64 64 1 id w6" "32 128 128 64 64 1 id w11" "32 256 128 64 64 1 none w6" "32 256 128 64 64 1 none w11" "32 128 128 64 64 1 conv w6 128" "32 128 128 64 64 1 conv w11 128"; do
  timeout 120 python tools/bench_conv.py $spec 2>&1 | grep "TFLOP\|conv_prep"
  [ -f $T ] && MINIMAGEN_HIP_LIB=$T timeout 120 python tools/bench_conv.py $spec 2>&1 | grep -v 'Warning\|amdgpu.ids\|ret = \|return _methods' | sed -n 2,10p
done
