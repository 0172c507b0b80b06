cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MINIMAGEN_HIP_LIB=$(pwd)/minimagen_amd/libminimagen_hip_trace.so
for NT in ${NTS:-1 8}; do
export NTILE=$NT
echo "== NTILE $NT"
python tools/bench_conv.py 64 8 3 256 256 0 none rp6 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 8 8 256 256 1 id rp6 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 16 16 64 64 1 id rp7 2>&1 | grep -v amdgpu.ids
done
