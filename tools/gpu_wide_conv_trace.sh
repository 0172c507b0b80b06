#!/bin/bash
# phase breakdown (-DMI_TRACE build) + standalone times of the wide-regime convs of the default Unet()
R=$GRAFT_REPO_ROOT; cd $R
T=$R/minimagen_amd/libminimagen_hip_trace.so
for spec in "32 512 512 16 16 1 id w10" "32 512 512 16 16 1 id w7" "32 256 256 32 32 1 id w7" "32 128 128 64 64 1 id w6" "32 256 128 64 64 1 none w6"; do
  timeout 120 python tools/bench_conv.py $spec 2>&1 | grep "TFLOP"
  [ -f $T ] && MINIMAGEN_HIP_LIB=$T timeout 120 python tools/bench_conv.py $spec 2>&1 | grep -v 'Warning\|amdgpu.ids\|ret = \|return _methods' | sed -n 2,10p
done
