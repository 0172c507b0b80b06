#!/bin/bash
# Where does a 256^2 / 128^2 8 -> 8 conv spend its time?  Timing-only library variants (tools/build_abl.sh: one phase of the LDS-DMA kernel
# dropped each) on the shapes of the SR U-Net, isolated launches back to back (tools/bench_conv.py).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/${1:-r05c}; mkdir -p $OUT
run() { # name lib cfgx
  echo "== $1" >> $OUT/ablation.txt
  for shape in "64 8 8 256 256 1 id rp6" "64 8 8 256 256 1 none rp6" "64 8 3 256 256 0 none rp6" "64 8 8 128 128 1 id rp6"; do
    MINIMAGEN_HIP_LIB=$2 CFGX=$3 NTILE=${NT:-0} timeout 120 python tools/bench_conv.py $shape 2>&1 | grep "us " >> $OUT/ablation.txt
  done
}
: > $OUT/ablation.txt
run "register form (product)" $ROOTDIR/minimagen_amd/libminimagen_hip.so 0
run "LDS-DMA form (product build)" $ROOTDIR/minimagen_amd/libminimagen_hip.so 0x10000
for n in ${ABLS:-1 2 4 8 16 32 18 63}; do
  run "LDS-DMA form, RP_ABL=$n" $ROOTDIR/minimagen_amd/libminimagen_hip_abl$n.so 0x10000
done
cat $OUT/ablation.txt
