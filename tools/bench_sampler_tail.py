"""Sampler tail of one denoising step at a super-resolution stage's size: the separate kernels (mi_cfg_x0_fwd + mi_quantile_fwd (4 launches)
+ mi_posterior_fwd) against the one-launch grouped kernel (mi_sampler_step_group_fwd).  usage: python tools/bench_sampler_tail.py [side] [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minimagen_amd import _lib as L                               # noqa: E402
from minimagen_amd.diffusion_model import GaussianDiffusion      # noqa: E402
from minimagen_amd.helpers import quantile_rank                  # noqa: E402


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda:0")
    lib = L.lib()
    T, n = 1000, 3 * side * side
    coef = GaussianDiffusion(timesteps=T).sampler_coef_table().to(dev)
    ts = torch.tensor([500], dtype=torch.int32, device=dev)
    pred2, xt = torch.randn(2 * B, n, device=dev), torch.randn(B, n, device=dev)
    x0, s, v = torch.zeros(B, n, device=dev), torch.zeros(B, device=dev), torch.zeros(B, 2, device=dev)
    hist = torch.zeros(3 * B * 2 * 2048, dtype=torch.int32, device=dev)
    sync = torch.zeros(lib.mi_sampler_group_sync_bytes(B, n), dtype=torch.uint8, device=dev)
    k_lo, k_hi, w = quantile_rank(n, 0.9)
    xa, xb = xt.clone(), xt.clone()
    st = L.current_stream()
    cp = L.MiCfgX0Params(B, n, pred2.data_ptr(), 1, 3.0, xa.data_ptr(), coef.data_ptr(), ts.data_ptr(), 0, x0.data_ptr(), hist.data_ptr())
    qp = L.MiQuantileParams(B, n, x0.data_ptr(), k_lo, k_hi, w, hist.data_ptr(), s.data_ptr(), v.data_ptr(), 1, 1)
    pp = L.MiPosteriorParams(B, n, T, x0.data_ptr(), s.data_ptr(), xa.data_ptr(), coef.data_ptr(), ts.data_ptr(), 0, 77, 0, 1 << 20)
    cf = L.MiCfgX0Params(B, n, pred2.data_ptr(), 1, 3.0, xb.data_ptr(), coef.data_ptr(), ts.data_ptr(), 0, 0, 0)
    qf = L.MiQuantileParams(B, n, 0, k_lo, k_hi, w, 0, s.data_ptr(), v.data_ptr(), 0, 0)
    pf = L.MiPosteriorParams(B, n, T, 0, 0, xb.data_ptr(), coef.data_ptr(), ts.data_ptr(), 0, 77, 0, 1 << 20)

    def separate():
        L.check(lib.mi_cfg_x0_fwd(C.byref(cp), st)); L.check(lib.mi_quantile_fwd(C.byref(qp), st)); L.check(lib.mi_posterior_fwd(C.byref(pp), st))

    def grouped():
        L.check(lib.mi_sampler_step_group_fwd(C.byref(cf), C.byref(qf), C.byref(pf), sync.data_ptr(), st))

    separate(); grouped()
    torch.cuda.synchronize()
    assert torch.equal(xa, xb), "grouped tail differs from the separate kernels"
    for name, fn in (("separate kernels (6 launches)", separate), ("grouped (1 launch)", grouped)):
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 300
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        alg = B * n * 4 * 4                     # pred2 (both halves) + x_t read, x written
        print(f"{side}x{side} B={B} {name}: {us:.1f} us per step tail ({alg / us / 1e3:.0f} GB/s of the algorithmic 16 B per element)")
    err = int(sync[8:12].cpu().view(torch.int32).item())
    print(f"error word {err:#x}; workgroups per image {lib.mi_sampler_group_size(n)}")


if __name__ == "__main__":
    main()
