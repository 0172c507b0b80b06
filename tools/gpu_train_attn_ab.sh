#!/bin/bash
# one box: the folded cross-attention kernels of the training path (forward isolated; forward + dq + dkv inside the step) at the SR U-Net's shape (B 32, 4096 tokens, 8 heads, C 16, 261 context rows) on
# the matrix cores against the fp32 VALU kernel (MI_FOLDED_ATTN_VALU=1), isolated (50 launches) and inside the SR training step; GPU tests first
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/train_attn; out=gpurun_out/train_attn
timeout 1200 python -m pytest tests/test_training.py -x -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
cat > /tmp/ub.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from minimagen_amd import train_ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for B, n in ((32, 4096), (32, 1024)):
    H, C, J = 8, 16, 261
    q = torch.randn(B, n, C, generator=g).to(dev); kf = (torch.randn(B, H, J, C, generator=g) * 0.5).to(dev); vf = torch.randn(B, H, J, C, generator=g).to(dev)
    mask = (torch.arange(J)[None, :] < torch.tensor([J - (r % 24) for r in range(B)])[:, None]).to(dev)
    with torch.no_grad():
        for _ in range(3): o = train_ops.folded_attention(q, kf, vf, mask)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): o = train_ops.folded_attention(q, kf, vf, mask)
        torch.cuda.synchronize(); print(f"B {B} tokens {n}: forward {(time.perf_counter() - t0) / 50 * 1e6:.1f} us, checksum {float(o.abs().sum()):.1f}")
PY
for tag in valu mfma valu_b mfma_b; do
  unset MI_FOLDED_ATTN_VALU
  case $tag in valu*) export MI_FOLDED_ATTN_VALU=1;; esac
  echo "== $tag"; timeout 120 python /tmp/ub.py 2>&1 | tail -2
  timeout 600 python bench.py --train-step-only > $out/train_$tag.json 2> $out/train_$tag.err
  python - <<PY
import json
d = json.loads([l for l in open("$out/train_$tag.json") if l.startswith("{")][-1])
h = d["hip_kernels"]
print("$tag", "fwd+bwd", round(h["ms_per_fwd_bwd"], 2), "with clip + Adam", round(h["ms_per_step_with_clip_and_adam"], 2))
PY
done
