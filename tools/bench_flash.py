"""Development aid: mi_flash_attn_fwd at the default Unet()'s 64 x 64 self-attention shape (multi-query, 8 heads, 4096 tokens + null row,
B rows), HIP events; MI_FLASH_MQ_QT / PREP=0 in the environment select the variant.  usage: python tools/bench_flash.py [B] [HW] [reps]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minimagen_amd import _lib as L
from minimagen_amd import packing as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
heads = 8
lib = L.lib()
dev = torch.device("cuda:0")
q, kv, null = torch.randn(B, HW, heads * 64, device=dev), torch.randn(B, HW, 128, device=dev), torch.randn(2, 64, device=dev)
out = torch.empty(B, HW, heads * 64, device=dev)
p = L.MiFlashAttnParams()
p.B, p.HW, p.heads, p.kv_heads, p.q, p.q_scale = B, HW, heads, 1, L.ptr(q), 64 ** -0.5 * P.LOG2E
p.null_k, p.null_v = L.ptr(null), L.ptr(null) + 4 * 64
p.k0, p.v0, p.n0, p.ld0, p.bs0 = L.ptr(kv), L.ptr(kv) + 4 * 64, HW, 128, HW * 128
p.out = L.ptr(out)
if os.environ.get("PREP", "1") != "0":
    nbytes = lib.mi_flash_kv_prep_bytes(B, HW + 1)
    prep = torch.empty((nbytes + 3) // 4, device=dev)
    p.kv_prep, p.kv_prep_bytes = L.ptr(prep), nbytes
st = L.current_stream()
for _ in range(2):
    L.check(lib.mi_flash_attn_fwd(C.byref(p), st))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    L.check(lib.mi_flash_attn_fwd(C.byref(p), st))
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 4.0 * B * heads * HW * (HW + 1) * 64
print(f"flash mq B={B} HW={HW} heads={heads} PREP={os.environ.get('PREP', '1')} QT={os.environ.get('MI_FLASH_MQ_QT', 'default')}: {ms:.3f} ms  {fl / ms * 1e-9:.1f} TFLOP/s algorithmic ({3 * fl / ms * 1e-9:.0f} issued)", flush=True)
