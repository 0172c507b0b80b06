"""Development aid: mi_gemm_f32 at the T5-small encoder's shapes (B=32, L=64 -> M=2048), HIP events, f16x3 block-scaled kernel vs the
exact-fp32 MFMA kernel (MI_GEMM_EXACT_F32=1 in the environment selects the latter for the whole process)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minimagen_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda:0")
for name, M, N, K, gated, act in (("qkv", 2048, 1536, 512, False, 0), ("o", 2048, 512, 512, False, 0), ("wi relu", 2048, 2048, 512, False, 1),
                                  ("wi gated-gelu", 2048, 1024, 512, True, 2), ("wo", 2048, 512, 2048, False, 0)):
    A, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05
    G = torch.randn(N, K, device=dev) * 0.05 if gated else None
    R = torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev)
    st = L.current_stream()
    for _ in range(3):
        L.check(lib.mi_gemm_f32(L.ptr(A), L.ptr(W), L.ptr(G), L.ptr(R), L.ptr(out), M, N, K, act, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        L.check(lib.mi_gemm_f32(L.ptr(A), L.ptr(W), L.ptr(G), L.ptr(R), L.ptr(out), M, N, K, act, st))
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    fl = 2.0 * M * N * K * (2 if gated else 1)
    ref = (A.double() @ W.double().t())
    print(f"{name:14s} M{M} N{N} K{K}: {us:7.1f} us  {fl / us * 1e-6:7.1f} TFLOP/s algorithmic", flush=True)
