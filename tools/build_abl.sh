#!/bin/bash
# Development aid: timing-only variants of the library with one phase of conv_rp_dma_kernel dropped (-DRP_ABL=n, conv_rp.hip) -> minimagen_amd/libminimagen_hip_abl<n>.so
cd "$(dirname "$0")/../minimagen_amd/csrc" || exit 1
make -j8 >/dev/null || exit 1
mkdir -p build_abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -DMI_BACKEND_STRING=\"hip-gfx950\" -Wno-unused-function"
for n in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DRP_ABL=$n -c conv_rp.hip -o build_abl/conv_rp_$n.o 2>/dev/null && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v conv_rp.o) build_abl/conv_rp_$n.o -o ../libminimagen_hip_abl$n.so && echo "built abl$n" ) &
done
wait
