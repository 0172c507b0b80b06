#!/bin/bash
# phase + wall-clock trace of the row-paired conv kernel at the SR U-Net's <= 128^2 shapes (needs the -DMI_TRACE build, see gpu_rp_trace.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MINIMAGEN_HIP_LIB=$(pwd)/minimagen_amd/libminimagen_hip_trace.so
python tools/bench_conv.py 64 16 16 64 64 1 id rp6 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 16 16 64 64 1 id rp7 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 16 16 64 64 1 none rp6 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 8 8 128 128 1 id rp6 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 8 8 128 128 1 conv rp6 8 2>&1 | grep -v amdgpu.ids
NTILE=4 python tools/bench_conv.py 64 8 8 256 256 1 id rp6 2>&1 | grep -v amdgpu.ids
