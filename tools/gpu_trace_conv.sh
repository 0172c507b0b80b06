#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MINIMAGEN_HIP_LIB=$(pwd)/minimagen_amd/libminimagen_hip_trace.so
TRACE_FLAGS=0x200 python tools/trace_conv.py 2>&1 | grep -v amdgpu.ids
