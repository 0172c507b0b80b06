cd "${GRAFT_REPO_ROOT:-/root/repo}"
for NT in 1 2 4 8; do export NTILE=$NT; echo NTILE $NT
python tools/bench_conv.py 64 8 8 256 256 1 id rp6 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 8 8 256 256 1 none rp6 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 8 3 256 256 0 none rp6 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 8 8 128 128 1 id rp6 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 8 8 128 128 1 conv rp6 8 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 16 16 64 64 1 id rp7 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 16 16 64 64 1 conv rp7 16 2>&1 | grep -v amdgpu.ids
done
