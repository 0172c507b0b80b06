cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MINIMAGEN_HIP_LIB=$(pwd)/minimagen_amd/libminimagen_hip_trace.so
export NTILE=1
python tools/bench_conv.py 64 16 16 64 64 1 id rp7 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 16 16 64 64 1 id rp6 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 16 16 64 64 1 none rp6 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 8 8 64 64 1 id rp6 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 8 16 16 64 64 1 id rp6 2>&1 | grep -v amdgpu.ids
