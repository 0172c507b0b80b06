# phase + wall-clock trace of the row-paired conv kernel at the 256^2 shapes (needs the -DMI_TRACE build:
#   make -C minimagen_amd/csrc LIBNAME=libminimagen_hip_trace.so BUILD=build_trace VARIANT_FLAGS=-DMI_TRACE)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MINIMAGEN_HIP_LIB=$(pwd)/minimagen_amd/libminimagen_hip_trace.so
for NT in 4 8; do export NTILE=$NT; echo "== NTILE $NT"
python tools/bench_conv.py 64 8 8 256 256 1 id rp6 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 8 3 256 256 0 none rp6 2>&1 | grep -v amdgpu.ids
done
