# strip-length (NTILE) sweep of the row-paired conv on the layer shapes of the SR U-Net
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for NT in 0 1 2 4 8; do export NTILE=$NT; echo "== NTILE $NT"
python tools/bench_conv.py 64 16 8 128 128 1 none rp6 2>&1 | grep rp6
python tools/bench_conv.py 64 8 8 128 128 1 id rp6 2>&1 | grep rp6
python tools/bench_conv.py 64 8 8 128 128 1 none rp6 2>&1 | grep rp6
python tools/bench_conv.py 64 16 16 64 64 1 id rp6 2>&1 | grep rp6
python tools/bench_conv.py 64 32 16 64 64 1 none rp6 2>&1 | grep rp6
python tools/bench_conv.py 64 8 8 256 256 1 none rp6 2>&1 | grep rp6
done
