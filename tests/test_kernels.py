"""Op-level parity of every HIP kernel, called through the C ABI, against the oracle / plain torch fp32 ops
(the arithmetic spec the reference itself uses).  Tolerances: fp32, op-level atol 2e-5 (+4e-6 relative for
the long k15 sums); bit-exact for the quantile selection and the elementwise sampler math."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from minimagen_amd import _lib as L, packing as P
from minimagen_amd.helpers import quantile_rank
from oracle import restated as R
from tests._backend import BACKENDS, GPU_ONLY, setup


def chan_stats(x):
    st = torch.zeros(x.shape[0], x.shape[1], 1, 2, dtype=torch.float64)          # fp64 partial (sum, sumsq): include/minimagen_hip.h
    st[:, :, 0, 0] = x.double().sum((2, 3))
    st[:, :, 0, 1] = (x.double() ** 2).sum((2, 3))
    return st.contiguous()


def tile_nt(lib, cfg, H, W):
    th, tw = C.c_int(), C.c_int()
    lib.mi_conv_tile_shape(cfg, C.byref(th), C.byref(tw))
    return -(-H // th.value) * -(-W // tw.value)


def check_stats(ost, ref, rtol=1e-5):
    assert ost.dtype == torch.float64
    ref = ref.double()
    dims = tuple(range(2, ref.dim()))
    es = (ost[..., 0].sum(-1) - ref.sum(dims)).abs().max().item() / max(ref.abs().sum(dims).max().item(), 1e-6)
    eq = (ost[..., 1].sum(-1) - (ref ** 2).sum(dims)).abs().max().item() / max((ref ** 2).sum(dims).max().item(), 1e-6)
    assert es < rtol and eq < rtol, (es, eq)


CONV_CASES = [
    # B, C0, C1, Cout, H, W, ks, stride, up2, gn, ss, res, tile_cfg
    (2, 8, 0, 8, 16, 64, 3, 1, 0, True, True, 'none', 0),
    (2, 16, 8, 16, 32, 32, 3, 1, 0, True, True, 'conv2', 1),
    (1, 8, 8, 8, 20, 36, 3, 1, 0, True, False, 'conv', 2),          # ragged tile edges
    (2, 16, 0, 16, 32, 32, 3, 1, 0, True, True, 'id', 2),
    (1, 8, 0, 3, 24, 64, 3, 1, 0, False, False, 'none', 0),          # final conv (Cout 3 -> padded tile)
    (2, 16, 0, 8, 32, 64, 3, 1, 1, False, False, 'none', 0),         # nearest x2 + conv
    (2, 8, 0, 16, 16, 32, 4, 2, 0, False, False, 'none', 1),         # downsample k4 s2
    (1, 32, 0, 16, 16, 64, 3, 1, 0, True, True, 'none', 0),
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_family(backend, case):
    dev = setup(backend)
    lib = L.lib()
    B, C0, C1, Cout, H, W, ks, stride, up2, gn, ss, res, cfg = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    rn = lambda *s: torch.randn(*s, generator=g)
    Hin, Win = (H // 2, W // 2) if up2 else (H * stride, W * stride)
    x0 = rn(B, C0, Hin, Win) * 1.5 + 0.3
    x1 = rn(B, C1, Hin, Win) if C1 else None
    Cin = C0 + C1
    w, bias = rn(Cout, Cin, ks, ks) * 0.2, rn(Cout)
    gamma, beta = 1 + 0.2 * rn(Cin), 0.1 * rn(Cin)
    sst = rn(B, 7 + 2 * Cin) * 0.3 if ss else None
    sk = 2 ** -0.5
    xin = torch.cat((x0, x1 * sk), 1) if C1 else x0
    h = xin
    if gn:
        h = F.group_norm(h, 8, gamma, beta, 1e-5)
        if ss:
            h = h * (sst[:, 7:7 + Cin, None, None] + 1) + sst[:, 7 + Cin:7 + 2 * Cin, None, None]
        h = F.silu(h)
    if up2:
        h = F.interpolate(h, scale_factor=2, mode='nearest')
    ref = F.conv2d(h, w, bias, stride=stride, padding=1)
    ct = lib.mi_conv_cout_tile(Cout)
    keep = {}
    d = lambda name, t: keep.setdefault(name, t.to(dev).contiguous())
    p = L.MiConvParams()
    p.B, p.H, p.W = B, H, W
    p.in0 = L.MiAct(d("x0", x0).data_ptr(), C0, d("s0", chan_stats(x0)).data_ptr(), 1, 1.0, 0)
    if C1:
        p.in1 = L.MiAct(d("x1", x1).data_ptr(), C1, d("s1", chan_stats(x1)).data_ptr(), 1, sk, 0)
    p.Cout, p.ksize, p.stride, p.up2 = Cout, ks, stride, up2
    p.w, p.bias = d("w", P.pack_conv_weight(w, ct)).data_ptr(), d("b", bias).data_ptr()
    if gn:
        p.gn_groups, p.gn_gamma, p.gn_beta, p.gn_eps = 8, d("g", gamma).data_ptr(), d("be", beta).data_ptr(), 1e-5
        if ss:
            p.scale_shift, p.ss_stride, p.ss_off = d("ss", sst).data_ptr(), sst.shape[1], 7
    if res != 'none':
        r0 = rn(B, Cout if res == 'id' else 5, H, W)
        r1 = rn(B, 3, H, W) if res == 'conv2' else None
        p.res0 = L.MiAct(d("r0", r0).data_ptr(), r0.shape[1], 0, 0, 1.0, 0)
        if res == 'id':
            ref = ref + r0
        else:
            rin = torch.cat((r0, r1 * sk), 1) if r1 is not None else r0
            rw, rb = rn(Cout, rin.shape[1], 1, 1) * 0.3, rn(Cout)
            ref = ref + F.conv2d(rin, rw, rb)
            p.res_w = d("rw", P.pack_conv_weight(rw, ct).reshape(rin.shape[1], -1)).data_ptr()
            p.res_b = d("rb", rb).data_ptr()
            if r1 is not None:
                p.res1 = L.MiAct(d("r1", r1).data_ptr(), 3, 0, 0, sk, 0)
    nt = tile_nt(lib, cfg, H, W)
    out = torch.full((B, Cout, H, W), float('nan'), device=dev)
    ost = torch.zeros(B, Cout, nt, 2, dtype=torch.float64, device=dev)
    p.out, p.out_stats, p.tile_cfg = out.data_ptr(), ost.data_ptr(), cfg
    L.check(lib.mi_conv_fwd(C.byref(p), L.current_stream()), "conv")
    assert (out.cpu() - ref).abs().max().item() < 2e-5
    check_stats(ost.cpu(), ref)


RP_CASES = [
    # B, C0, C1, Cout, H, W, gn, ss, res, tile_cfg, xscale, wscale
    (2, 8, 0, 8, 16, 64, True, True, 'id', 6, 1.0, 1.0),
    (1, 8, 8, 8, 24, 72, True, True, 'conv2', 6, 1.0, 1.0),          # ragged tile edges, concat input, 1x1 residual over a concat
    (8, 16, 0, 16, 16, 32, True, True, 'id', 7, 1.0, 1.0),           # B % 8 == 0: the XCD-aware workgroup -> image map
    (1, 16, 16, 16, 20, 64, True, False, 'conv', 6, 1.0, 1.0),
    (1, 8, 0, 3, 16, 64, False, False, 'none', 6, 1.0, 1.0),          # final conv (Cout 3 -> one padded N tile)
    (1, 32, 0, 32, 8, 32, True, True, 'none', 7, 1.0, 1.0),           # four N tiles
    (1, 8, 0, 8, 16, 64, True, True, 'id', 6, 256.0, 256.0),          # range safety of the fp16 split: large / small operands
    (1, 8, 8, 8, 16, 32, True, False, 'conv', 7, 1.0 / 256, 1.0 / 256),
    (1, 8, 0, 8, 16, 32, False, False, 'none', 7, 1.0 / 256, 300.0),
    (1, 8, 0, 8, 16, 32, False, False, 'id', 7, 4096.0, 1.0 / 300),
    (2, 8, 0, 8, 40, 128, True, True, 'id', 6 | (3 << 12), 1.0, 1.0),        # strips of 3 tiles per workgroup (ragged last strip)
    (1, 16, 16, 16, 24, 64, True, True, 'conv2', 7 | (4 << 12), 1.0, 1.0),   # strips x several rounds (concat input + 1x1 residual)
    (2, 16, 8, 16, 32, 64, True, True, 'none', 6 | (2 << 12), 1.0, 1.0),     # 16 + 8 skip channels -> 16 (base U-Net up path, Unet.py:166-178)
    (2, 16, 0, 16, 32, 32, True, True, 'conv2', 7, 1.0, 1.0),                # ... and its second conv with the 1x1 residual over the 24
    (1, 16, 0, 16, 16, 64, True, False, 'conv2', 6, 1.0 / 256, 1.0),
    # more 8x64-tile shapes of the 8-channel layers
    (8, 8, 0, 8, 24, 200, True, True, 'id', 6, 1.0, 1.0),                    # B % 8 == 0, ragged tile edges in both directions
    (1, 8, 0, 3, 24, 136, False, False, 'none', 6, 1.0, 1.0),                # the final conv: no GroupNorm, 3 output channels
    (2, 8, 0, 8, 16, 64, True, True, 'none', 6 | (1 << 12), 1.0, 1.0),       # one-tile strips
    (1, 8, 0, 8, 72, 64, True, False, 'id', 6 | (5 << 12), 1.0 / 64, 30.0),  # odd strip length, ragged last strip, scaled operands
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", RP_CASES)
def test_conv_row_paired_path(backend, case):
    """k3 s1 conv on v_mfma_f32_16x16x32_f16 with N = (row parity, 8 channels) and power-of-two scaled fp16x3 operand splits vs torch
    fp32: same tolerance as the VALU path, relative to the magnitude of the output for the scaled cases"""
    dev = setup(backend)
    lib = L.lib()
    B, C0, C1, Cout, H, W, gn, ss, res, cfg, xs, wsc = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    rn = lambda *s_: torch.randn(*s_, generator=g)
    x0 = (rn(B, C0, H, W) * 1.5 + 0.3) * xs
    x1 = rn(B, C1, H, W) * xs if C1 else None
    Cin = C0 + C1
    w, bias = rn(Cout, Cin, 3, 3) * 0.2 * wsc, rn(Cout) * wsc * (1.0 if gn else xs)
    gamma, beta = 1 + 0.2 * rn(Cin), 0.1 * rn(Cin)
    sst = rn(B, 7 + 2 * Cin) * 0.3 if ss else None
    if ss and xs != 1.0:
        sst[:, 7] = 50.0                    # a large scale and shift on some channels (layers.py:141-143)
        sst[:, 7 + Cin + 1] = -50.0
    sk = 2 ** -0.5
    h = torch.cat((x0, x1 * sk), 1) if C1 else x0
    if gn:
        h = F.group_norm(h, 8, gamma, beta, 1e-5)
        if ss:
            h = h * (sst[:, 7:7 + Cin, None, None] + 1) + sst[:, 7 + Cin:7 + 2 * Cin, None, None]
        h = F.silu(h)
    ref = F.conv2d(h.double(), w.double(), bias.double(), padding=1)
    keep = {}
    d = lambda name, t: keep.setdefault(name, t.to(dev).contiguous())
    p = L.MiConvParams()
    p.B, p.H, p.W = B, H, W
    p.in0 = L.MiAct(d("x0", x0).data_ptr(), C0, d("s0", chan_stats(x0)).data_ptr(), 1, 1.0, 0)
    if C1:
        p.in1 = L.MiAct(d("x1", x1).data_ptr(), C1, d("s1", chan_stats(x1)).data_ptr(), 1, sk, 0)
    p.Cout, p.ksize, p.stride, p.up2 = Cout, 3, 1, 0
    wf, wexp = P.pack_conv_weight_rp(w)
    p.w_rp, p.w_rp_exp, p.bias = d("wf", wf).data_ptr(), wexp, d("b", bias).data_ptr()
    if gn:
        p.gn_groups, p.gn_gamma, p.gn_beta, p.gn_eps = 8, d("g", gamma).data_ptr(), d("be", beta).data_ptr(), 1e-5
        if ss:
            p.scale_shift, p.ss_stride, p.ss_off = d("ss", sst).data_ptr(), sst.shape[1], 7
    if res != 'none':
        rsc = xs if not gn else 1.0
        r0 = rn(B, Cout if res == 'id' else 8, H, W) * rsc
        r1 = rn(B, 16, H, W) * rsc if res == 'conv2' else None
        p.res0 = L.MiAct(d("r0", r0).data_ptr(), r0.shape[1], d("rs0", chan_stats(r0)).data_ptr(), 1, 1.0, 0)
        if res == 'id':
            ref = ref + r0
        else:
            rin = torch.cat((r0, r1 * sk), 1) if r1 is not None else r0
            rw, rb = rn(Cout, rin.shape[1], 1, 1) * 0.3, rn(Cout)
            ref = ref + F.conv2d(rin.double(), rw.double(), rb.double())
            p.res_w = 1          # non-null marker: the residual is a 1x1 conv
            rwf, rwexp = P.pack_conv_weight_rp(rw)
            p.res_w_rp, p.res_w_rp_exp = d("rwf", rwf).data_ptr(), rwexp
            p.res_b = d("rb", rb).data_ptr()
            if r1 is not None:
                p.res1 = L.MiAct(d("r1", r1).data_ptr(), 16, d("rs1", chan_stats(r1)).data_ptr(), 1, sk, 0)
    nt = tile_nt(lib, cfg & 0xff, H, W)
    out = torch.full((B, Cout, H, W), float('nan'), device=dev)
    ost = torch.zeros(B, Cout, nt, 2, dtype=torch.float64, device=dev)
    p.out, p.out_stats, p.tile_cfg = out.data_ptr(), ost.data_ptr(), cfg
    L.check(lib.mi_conv_fwd(C.byref(p), L.current_stream()), "conv rp")
    scale = max(1.0, ref.abs().max().item() / 8.0)          # unit-scale cases: outputs of magnitude ~8
    err = (out.cpu().double() - ref).abs().max().item()
    print(f"rp conv {case}: max|d| = {err:.2e} (gate {2e-5 * scale:.2e}, |ref|max {ref.abs().max().item():.3g})")
    assert err < 2e-5 * scale
    check_stats(ost.cpu(), ref.float())


BF16_CASES = [
    # B, C0, C1, Cout, H, W, res, tile_cfg, output storage (1 = bf16, 0 = fp32: the final conv feeds the fp32 sampler)
    (2, 8, 0, 8, 16, 64, 'id', 6, 1),
    (1, 8, 8, 8, 24, 72, 'conv2', 6, 1),          # concat input + 1x1 residual over a concat, all stored as bf16
    (2, 16, 0, 16, 16, 32, 'id', 7, 1),
    (1, 8, 0, 3, 16, 64, 'none', 6, 0),
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", BF16_CASES)
def test_conv_row_paired_bf16_storage(backend, case):
    """reduced-precision configuration (BASELINE configs 3-5): the single-term kernels (tile_cfg | 0x400) read and write activations stored
    as bf16 (mi_act.st / out_st); statistics and accumulation stay fp32.  Against torch fp32 on the same bf16-rounded inputs: the error
    budget is the single fp16 term (2^-11 per operand) plus the bf16 rounding of the output (2^-9)"""
    dev = setup(backend)
    lib = L.lib()
    B, C0, C1, Cout, H, W, res, cfg, ost = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    rn = lambda *s_: torch.randn(*s_, generator=g)
    bf = lambda t: t.to(torch.bfloat16)
    x0 = bf(rn(B, C0, H, W) * 1.5 + 0.3)
    x1 = bf(rn(B, C1, H, W)) if C1 else None
    Cin = C0 + C1
    w, bias = rn(Cout, Cin, 3, 3) * 0.2, rn(Cout)
    gamma, beta = 1 + 0.2 * rn(Cin), 0.1 * rn(Cin)
    sk = 2 ** -0.5
    h = torch.cat((x0.float(), x1.float() * sk), 1) if C1 else x0.float()
    h = F.silu(F.group_norm(h, 8, gamma, beta, 1e-5))
    ref = F.conv2d(h.double(), w.double(), bias.double(), padding=1)
    keep = {}
    d = lambda name, t: keep.setdefault(name, t.to(dev).contiguous())
    p = L.MiConvParams()
    p.B, p.H, p.W = B, H, W
    p.in0 = L.MiAct(d("x0", x0).data_ptr(), C0, d("s0", chan_stats(x0.float())).data_ptr(), 1, 1.0, 0, 1)
    if C1:
        p.in1 = L.MiAct(d("x1", x1).data_ptr(), C1, d("s1", chan_stats(x1.float())).data_ptr(), 1, sk, 0, 1)
    p.Cout, p.ksize, p.stride, p.up2 = Cout, 3, 1, 0
    wf, wexp = P.pack_conv_weight_rp(w)
    p.w_rp, p.w_rp_exp, p.bias = d("wf", wf).data_ptr(), wexp, d("b", bias).data_ptr()
    p.gn_groups, p.gn_gamma, p.gn_beta, p.gn_eps = 8, d("g", gamma).data_ptr(), d("be", beta).data_ptr(), 1e-5
    if res != 'none':
        r0 = bf(rn(B, Cout if res == 'id' else 8, H, W))
        r1 = bf(rn(B, 16, H, W)) if res == 'conv2' else None
        p.res0 = L.MiAct(d("r0", r0).data_ptr(), r0.shape[1], d("rs0", chan_stats(r0.float())).data_ptr(), 1, 1.0, 0, 1)
        if res == 'id':
            ref = ref + r0.double()
        else:
            rin = torch.cat((r0.float(), r1.float() * sk), 1)
            rw, rb = rn(Cout, rin.shape[1], 1, 1) * 0.3, rn(Cout)
            ref = ref + F.conv2d(rin.double(), rw.double(), rb.double())
            p.res_w = 1
            rwf, p.res_w_rp_exp = P.pack_conv_weight_rp(rw)
            p.res_w_rp, p.res_b = d("rwf", rwf).data_ptr(), d("rb", rb).data_ptr()
            p.res1 = L.MiAct(d("r1", r1).data_ptr(), 16, d("rs1", chan_stats(r1.float())).data_ptr(), 1, sk, 0, 1)
    nt = tile_nt(lib, cfg, H, W)
    out = torch.full((B, Cout, H, W), float('nan'), device=dev).to(torch.bfloat16 if ost else torch.float32)
    ostat = torch.zeros(B, Cout, nt, 2, dtype=torch.float64, device=dev)
    p.out, p.out_st, p.out_stats, p.tile_cfg = out.data_ptr(), ost, ostat.data_ptr(), cfg | 0x400
    L.check(lib.mi_conv_fwd(C.byref(p), L.current_stream()), "conv rp (bf16 storage)")
    err = (out.cpu().double() - ref).abs()
    tol = 2e-3 * ref.abs().max().item() + (2.0 ** -8) * ref.abs() * ost            # single fp16 term + bf16 rounding of the stored result
    print(f"rp conv bf16 storage {case}: max|d| = {err.max().item():.2e} (|ref|max {ref.abs().max().item():.3g})")
    assert bool((err <= tol).all())
    check_stats(ostat.cpu(), ref.float(), rtol=2e-3)


RP_MODE_CASES = [
    # B, Cin, Cout, H, W (OUTPUT), ksize, stride, up2, tile_cfg, xscale
    (2, 16, 8, 32, 64, 3, 1, 1, 6, 1.0),            # nearest x2 + conv (Upsample, layers.py:512-515)
    (8, 8, 8, 16, 128, 3, 1, 1, 6 | (2 << 12), 1.0),
    (1, 8, 8, 24, 72, 3, 1, 1, 6, 1.0 / 256),        # ragged tile edges
    (2, 8, 16, 16, 32, 4, 2, 0, 7, 1.0),            # Downsample k4 s2 (layers.py:319)
    (1, 8, 8, 20, 40, 4, 2, 0, 7 | (2 << 12), 300.0),
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", RP_MODE_CASES)
def test_conv_row_paired_resampling(backend, case):
    """the nearest-x2 + k3 and the k4 s2 members of the row-paired matrix-core kernel vs torch fp32"""
    dev = setup(backend)
    lib = L.lib()
    B, Cin, Cout, H, W, ks, stride, up2, cfg, xs = case
    g = torch.Generator().manual_seed(sum(int(v * 7) for v in case) & 0xffff)
    rn = lambda *s_: torch.randn(*s_, generator=g)
    Hin, Win = (H // 2, W // 2) if up2 else (H * stride, W * stride)
    x0 = (rn(B, Cin, Hin, Win) * 1.5 + 0.3) * xs
    w, bias = rn(Cout, Cin, ks, ks) * 0.2, rn(Cout) * xs
    h = F.interpolate(x0, scale_factor=2, mode='nearest') if up2 else x0
    ref = F.conv2d(h.double(), w.double(), bias.double(), stride=stride, padding=1)
    keep = {}
    d = lambda name, t: keep.setdefault(name, t.to(dev).contiguous())
    p = L.MiConvParams()
    p.B, p.H, p.W = B, H, W
    p.in0 = L.MiAct(d("x0", x0).data_ptr(), Cin, d("s0", chan_stats(x0)).data_ptr(), 1, 1.0, 0)
    p.Cout, p.ksize, p.stride, p.up2 = Cout, ks, stride, up2
    wf, wexp = P.pack_conv_weight_rp(w)
    p.w_rp, p.w_rp_exp, p.bias = d("wf", wf).data_ptr(), wexp, d("b", bias).data_ptr()
    nt = tile_nt(lib, cfg, H, W)
    out = torch.full((B, Cout, H, W), float('nan'), device=dev)
    ost = torch.zeros(B, Cout, nt, 2, dtype=torch.float64, device=dev)
    p.out, p.out_stats, p.tile_cfg = out.data_ptr(), ost.data_ptr(), cfg
    L.check(lib.mi_conv_fwd(C.byref(p), L.current_stream()), "conv rp")
    scale = max(1.0, ref.abs().max().item() / 8.0)
    err = (out.cpu().double() - ref).abs().max().item()
    print(f"rp resampling conv {case}: max|d| = {err:.2e} (gate {2e-5 * scale:.2e})")
    assert err < 2e-5 * scale
    check_stats(ost.cpu(), ref.float())


WIDE_CASES = [
    # B, C0, C1, Cout, H, W (output), ks, stride, up2, gn, ss, res
    (1, 128, 0, 128, 16, 32, 3, 1, 0, True, True, 'id'),
    (2, 96, 64, 72, 8, 32, 3, 1, 0, True, True, 'conv'),          # concat input, 1x1 residual over the concat, Cout not a multiple of 32
    (2, 96, 64, 72, 8, 32, 3, 1, 0, True, True, 'none'),
    (2, 96, 64, 64, 8, 32, 3, 1, 0, True, True, 'conv'),
    (2, 160, 0, 64, 8, 32, 3, 1, 0, True, True, 'none'),
    (2, 160, 0, 64, 8, 32, 3, 1, 0, False, False, 'none'),
    (1, 72, 0, 3, 16, 32, 3, 1, 0, False, False, 'none'),         # final conv of a wide net
    (1, 128, 0, 64, 16, 32, 3, 1, 1, False, False, 'none'),       # nearest x2 + conv
    (1, 64, 0, 136, 8, 32, 4, 2, 0, False, False, 'none'),        # downsample k4 s2
    (1, 128, 0, 64, 16, 128, 3, 1, 0, True, True, 'id', 6),       # 8 x 64 tiles (tile_cfg 6: the wide k3 s1 member where the image is a multiple of 64 wide)
    (2, 96, 64, 72, 8, 64, 3, 1, 0, False, False, 'conv', 6),     # ... concat input, 1x1 residual over the concat, ragged N tiles, no GroupNorm
    (2, 128, 0, 64, 16, 16, 3, 1, 0, True, True, 'id', 10),       # 16 x 16 tiles (tile_cfg 10: images no wider than 16)
    (1, 96, 64, 72, 24, 12, 3, 1, 0, True, False, 'conv', 10),    # ... ragged in both directions, concat input + 1x1 residual, ragged N tiles
    (2, 128, 0, 128, 16, 16, 3, 1, 0, True, True, 'id', 11),      # the wide GEMM kernel (conv_wide.hip: prepared operand planes, 8 x 16 x 128-channel tiles)
    (1, 96, 64, 128, 24, 12, 3, 1, 0, True, False, 'conv', 11),   # ... ragged in both directions, concat input, 1x1 residual over a concat
    (2, 64, 0, 256, 8, 32, 3, 1, 0, False, False, 'none', 11),    # ... no GroupNorm, two tiles across, two channel groups
    (1, 64, 0, 192, 16, 16, 3, 1, 0, True, True, 'none', 11),     # ... 192 output channels: three 64-channel workgroups
    (2, 128, 0, 64, 16, 32, 3, 1, 1, False, False, 'none', 11),   # ... nearest x2 + conv: the operand planes are written up-sampled
    (8, 32, 32, 64, 8, 32, 3, 1, 0, True, True, 'id', 11),        # ... a batch of eight: the XCD-aware workgroup -> image map
]


# the level shapes of the `Super` preset (dim 128, memory_efficient: Unet.py:637-692) at their REAL sizes for a 256^2 input -- GPU only (the
# emulator would take minutes): first level 128 channels @128^2 (ResnetBlock conv with identity residual; the skip-concat conv of the up
# path 256 -> 128 with its 1x1 residual), second level 256 channels @64^2, the k4 s2 Downsample 128 -> 256, nearest x2 + conv 256 -> 128
WIDE_REAL_CASES = [
    (1, 128, 0, 128, 128, 128, 3, 1, 0, True, True, 'id'),
    (1, 128, 128, 128, 128, 128, 3, 1, 0, True, True, 'conv'),
    (1, 256, 0, 256, 64, 64, 3, 1, 0, True, True, 'id'),
    (1, 128, 0, 256, 64, 64, 4, 2, 0, False, False, 'none'),
    (1, 256, 0, 128, 128, 128, 3, 1, 1, False, False, 'none'),
    (1, 128, 0, 128, 128, 128, 3, 1, 0, True, True, 'id', 11),    # the same shapes on the wide GEMM kernel (tile_cfg 11)
    (1, 96, 64, 128, 128, 128, 3, 1, 0, True, True, 'conv', 11),
    (1, 256, 0, 256, 64, 64, 3, 1, 0, True, True, 'id', 11),
    (1, 256, 0, 128, 128, 128, 3, 1, 1, False, False, 'none', 11),
]


@pytest.mark.parametrize("backend", GPU_ONLY)
@pytest.mark.parametrize("case", WIDE_REAL_CASES)
def test_conv_wide_regime_at_the_super_presets_level_shapes(backend, case):
    test_conv_wide_regime(backend, case)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", WIDE_CASES)
def test_conv_wide_regime(backend, case):
    """the wide-channel regime of the row-paired matrix-core conv (output channels tiled over the grid, per-channel affine and operand
    exponents from mi_gn_coef_fwd) vs torch fp32"""
    dev = setup(backend)
    lib = L.lib()
    B, C0, C1, Cout, H, W, ks, stride, up2, gn, ss, res = case[:12]
    tcfg = case[12] if len(case) > 12 else 7
    g = torch.Generator().manual_seed(sum(int(v) if not isinstance(v, str) else len(v) for v in case))
    rn = lambda *s_: torch.randn(*s_, generator=g)
    Hin, Win = (H // 2, W // 2) if up2 else (H * stride, W * stride)
    x0 = rn(B, C0, Hin, Win) * 1.5 + 0.3
    x1 = rn(B, C1, Hin, Win) if C1 else None
    Cin = C0 + C1
    w, bias = rn(Cout, Cin, ks, ks) * (0.5 / (Cin * ks * ks) ** 0.5), rn(Cout)
    gamma, beta = 1 + 0.2 * rn(Cin), 0.1 * rn(Cin)
    sst = rn(B, 7 + 2 * Cin) * 0.3 if ss else None
    sk = 2 ** -0.5
    h = torch.cat((x0, x1 * sk), 1) if C1 else x0
    if gn:
        h = F.group_norm(h, 8, gamma, beta, 1e-5)
        if ss:
            h = h * (sst[:, 7:7 + Cin, None, None] + 1) + sst[:, 7 + Cin:7 + 2 * Cin, None, None]
        h = F.silu(h)
    if up2:
        h = F.interpolate(h, scale_factor=2, mode='nearest')
    ref = F.conv2d(h.double(), w.double(), bias.double(), stride=stride, padding=1)
    keep = {}
    d = lambda name, t: keep.setdefault(name, t.to(dev).contiguous())
    p = L.MiConvParams()
    p.B, p.H, p.W = B, H, W
    p.in0 = L.MiAct(d("x0", x0).data_ptr(), C0, d("s0", chan_stats(x0)).data_ptr(), 1, 1.0, 0)
    if C1:
        p.in1 = L.MiAct(d("x1", x1).data_ptr(), C1, d("s1", chan_stats(x1)).data_ptr(), 1, sk, 0)
    p.Cout, p.ksize, p.stride, p.up2 = Cout, ks, stride, up2
    pack = P.pack_conv_weight_ig if tcfg == 11 else P.pack_conv_weight_rp
    wf, wexp = pack(w)
    p.w_rp, p.w_rp_exp, p.bias = d("wf", wf).data_ptr(), wexp, d("b", bias).data_ptr()
    if gn:
        p.gn_groups, p.gn_gamma, p.gn_beta, p.gn_eps = 8, d("g", gamma).data_ptr(), d("be", beta).data_ptr(), 1e-5
        if ss:
            p.scale_shift, p.ss_stride, p.ss_off = d("ss", sst).data_ptr(), sst.shape[1], 7
    if res == 'id':
        r0 = rn(B, Cout, H, W)
        p.res0 = L.MiAct(d("r0", r0).data_ptr(), Cout, 0, 0, 1.0, 0)
        ref = ref + r0
    elif res == 'conv':
        r0, r1 = rn(B, 96, H, W), rn(B, 64, H, W)
        rin = torch.cat((r0, r1 * sk), 1)
        rw, rb = rn(Cout, 160, 1, 1) * 0.1, rn(Cout)
        ref = ref + F.conv2d(rin.double(), rw.double(), rb.double())
        p.res0 = L.MiAct(d("r0", r0).data_ptr(), 96, d("rs0", chan_stats(r0)).data_ptr(), 1, 1.0, 0)
        p.res1 = L.MiAct(d("r1", r1).data_ptr(), 64, d("rs1", chan_stats(r1)).data_ptr(), 1, sk, 0)
        p.res_w = 1
        rwf, p.res_w_rp_exp = pack(rw)
        p.res_w_rp, p.res_b = d("rwf", rwf).data_ptr(), d("rb", rb).data_ptr()
    coef = torch.zeros(B, Cin, 4, device=dev)
    exps = torch.zeros(B, 2, dtype=torch.int32, device=dev)
    p.gn_coef, p.gn_exps = coef.data_ptr(), exps.data_ptr()
    nt = tile_nt(lib, tcfg, H, W)
    out = torch.full((B, Cout, H, W), float('nan'), device=dev)
    ost = torch.zeros(B, Cout, nt, 2, dtype=torch.float64, device=dev)
    p.out, p.out_stats, p.tile_cfg = out.data_ptr(), ost.data_ptr(), tcfg
    L.check(lib.mi_gn_coef_fwd(C.byref(p), L.current_stream()), "gn coef")
    if tcfg == 11:
        nbytes = lib.mi_conv_prep_bytes(B, Cin, 160 if res == 'conv' else 0, H, W)
        prep = torch.full((nbytes // 4,), float('nan'), device=dev)
        p.act_prep, p.act_prep_bytes = prep.data_ptr(), nbytes - 16
        assert lib.mi_conv_prep_fwd(C.byref(p), L.current_stream()) != 0 and b"too small" in lib.mi_last_error()
        p.act_prep_bytes = nbytes
        L.check(lib.mi_conv_prep_fwd(C.byref(p), L.current_stream()), "conv prep")
    L.check(lib.mi_conv_fwd(C.byref(p), L.current_stream()), "conv wide")
    if tcfg == 11 and Cout % 128 == 0:
        # small launches take 64 output channels per workgroup; the 128-channel form (forced here) gives the same bits
        out64, ost64 = out.clone(), ost.clone()
        out.fill_(float('nan')); ost.zero_()
        os.environ["MI_CONV_WIDE_N128"] = "1"
        try:
            L.check(lib.mi_conv_fwd(C.byref(p), L.current_stream()), "conv wide (128 channels per workgroup)")
        finally:
            del os.environ["MI_CONV_WIDE_N128"]
        assert torch.equal(out.cpu(), out64.cpu()) and torch.equal(ost.cpu(), ost64.cpu())
    scale = max(1.0, ref.abs().max().item() / 8.0)
    err = (out.cpu().double() - ref).abs().max().item()
    print(f"wide conv {case}: max|d| = {err:.2e} (gate {2e-5 * scale:.2e}, exps {exps.cpu().tolist()})")
    assert err < 2e-5 * scale
    check_stats(ost.cpu(), ref.float())


def test_conv_rejects_bad_arguments():
    setup("emu")
    lib = L.lib()
    p = L.MiConvParams()
    p.B, p.H, p.W, p.Cout, p.ksize, p.stride = 1, 8, 8, 8, 5, 1
    p.in0 = L.MiAct(1, 8, 0, 0, 1.0, 0)
    assert lib.mi_conv_fwd(C.byref(p), None) == -3 and b"unsupported" in lib.mi_last_error()
    p.ksize, p.gn_groups = 3, 8          # GroupNorm without statistics
    assert lib.mi_conv_fwd(C.byref(p), None) == -1
    p.gn_groups, p.B = 0, 0              # empty batch
    assert lib.mi_conv_fwd(C.byref(p), None) == -1
    p.B, p.out_st = 1, 1                 # bf16 storage outside the single-term row-paired kernels
    assert lib.mi_conv_fwd(C.byref(p), None) == -3 and b"bf16" in lib.mi_last_error()


CE_CASES = [(2, 3, 0, 64, 64, (3, 7, 15), (4, 2, 2), 0, 0), (2, 3, 3, 40, 72, (3, 7, 15), (4, 2, 2), 1, 0),
            (4, 3, 3, 16, 32, (3, 7, 15), (4, 2, 2), 2, 2), (1, 3, 0, 24, 40, (3, 7, 15), (8, 4, 4), 2, 0),
            (1, 3, 3, 16, 64, (3, 5), (4, 4), 0, 0),
            (2, 3, 3, 24, 40, (3, 7, 15), (16, 8, 8), 2, 1), (1, 3, 0, 16, 64, (3, 7), (24, 8), 0, 0)]      # members in multiples of 8: eight channels per work-item


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", CE_CASES)
def test_crossembed(backend, case):
    dev = setup(backend)
    lib = L.lib()
    B, C0, C1, H, W, ks, cout, cfg, mod = case
    g = torch.Generator().manual_seed(3)
    rn = lambda *s: torch.randn(*s, generator=g)
    Bx = mod if mod else B
    x0, x1 = rn(Bx, C0, H, W), (rn(B, C1, H, W) if C1 else None)
    ws = [rn(co, C0 + C1, k, k) * 0.1 for k, co in zip(ks, cout)]
    bs = [rn(co) for co in cout]
    x0e = x0.repeat(B // Bx, 1, 1, 1)
    xin = torch.cat((x0e, x1), 1) if C1 else x0e
    ref = torch.cat([F.conv2d(xin, w, b, padding=(k - 1) // 2) for w, b, k in zip(ws, bs, ks)], 1)
    p = L.MiCrossEmbedParams()
    p.B, p.H, p.W = B, H, W
    x0d = x0.to(dev); x1d = x1.to(dev) if C1 else None
    p.in0, p.C0, p.in0_batch_mod = x0d.data_ptr(), C0, mod
    if C1:
        p.in1, p.C1 = x1d.data_ptr(), C1
    p.n_kernels = len(ks)
    wp = [w.permute(1, 2, 3, 0).contiguous().to(dev) for w in ws]
    bd = [b.to(dev) for b in bs]
    for i in range(len(ks)):
        p.ksize[i], p.cout[i], p.w[i], p.bias[i] = ks[i], cout[i], wp[i].data_ptr(), bd[i].data_ptr()
    nt = tile_nt(lib, cfg, H, W)
    out = torch.full(ref.shape, float('nan'), device=dev)
    ost = torch.zeros(B, ref.shape[1], nt, 2, dtype=torch.float64, device=dev)
    p.out, p.out_stats, p.tile_cfg = out.data_ptr(), ost.data_ptr(), cfg
    L.check(lib.mi_crossembed_fwd(C.byref(p), L.current_stream()), "crossembed")
    assert (out.cpu() - ref).abs().max().item() < 2e-5 + 4e-6 * ref.abs().max().item()
    check_stats(ost.cpu(), ref)


CE_MFMA_CASES = [
    # B, Cin, channels of the full weight, first channel, H, W, tile_cfg, in0_batch_mod, addend, xscale, wscale
    (2, 3, 3, 0, 64, 64, 9, 0, False, 1.0, 1.0),
    (2, 3, 6, 3, 40, 72, 8, 0, True, 1.0, 1.0),           # ragged tile edges; the low-res half of a 6-channel weight; addend
    (4, 3, 3, 0, 16, 32, 9, 2, False, 300.0, 1.0 / 64),   # shared input rows (guidance halves); range safety of the fp16 split
    (1, 4, 4, 0, 32, 64, 8, 0, True, 1.0 / 256, 30.0),
    (1, 1, 3, 1, 24, 40, 9, 0, False, 1.0, 1.0),
    (1, 3, 3, 0, 72, 136, 8, 0, False, 1.0, 1.0),         # several tiles per image in both directions
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", CE_MFMA_CASES)
def test_crossembed_matrix_core(backend, case):
    """CrossEmbedLayer (layers.py:298-305) as a Toeplitz GEMM on the matrix cores (3-term fp16 splits) vs torch: same gate as the
    VALU kernel, relative to the output magnitude for the scaled cases; an all-zero image must give exactly bias (+ addend)"""
    dev = setup(backend)
    lib = L.lib()
    B, Cin, Cw, c0, H, W, cfg, mod, with_add, xs, wsc = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    rn = lambda *s_: torch.randn(*s_, generator=g)
    ks, cout = (3, 7, 15), (4, 2, 2)
    Bx = mod if mod else B
    x = rn(Bx, Cin, H, W) * xs
    ws = [rn(co, Cw, k, k) * 0.1 * wsc for k, co in zip(ks, cout)]
    bs = [rn(co) * xs * wsc for co in cout]
    add = rn(B, 8, H, W) * xs * wsc if with_add else None
    tab, exps = P.pack_crossembed_mfma(ws, c0, Cin)
    for zero in (False, True):
        xe = (torch.zeros_like(x) if zero else x).repeat(B // Bx, 1, 1, 1)
        ref = torch.cat([F.conv2d(xe.double(), w[:, c0:c0 + Cin].double(), b.double(), padding=(k - 1) // 2) for w, b, k in zip(ws, bs, ks)], 1)
        if with_add:
            ref = ref + add.double()
        p = L.MiCrossEmbedParams()
        p.B, p.H, p.W = B, H, W
        xd, tabd = (torch.zeros_like(x) if zero else x).to(dev), tab.to(dev)
        p.in0, p.C0, p.in0_batch_mod = xd.data_ptr(), Cin, mod
        p.n_kernels = 3
        bd = [b.to(dev) for b in bs]
        for i in range(3):
            p.ksize[i], p.cout[i], p.bias[i], p.w_mfma_exp[i] = ks[i], cout[i], bd[i].data_ptr(), exps[i]
        p.w_mfma = tabd.data_ptr()
        addd = add.to(dev) if with_add else None
        p.addend = addd.data_ptr() if with_add else 0
        nt = tile_nt(lib, cfg, H, W)
        out = torch.full(ref.shape, float('nan'), device=dev)
        ost = torch.zeros(B, 8, nt, 2, dtype=torch.float64, device=dev)
        p.out, p.out_stats, p.tile_cfg = out.data_ptr(), ost.data_ptr(), cfg
        L.check(lib.mi_crossembed_fwd(C.byref(p), L.current_stream()), "crossembed (matrix cores)")
        err = (out.cpu().double() - ref).abs().max().item()
        if zero:
            assert torch.equal(out.cpu(), ref.float())
            continue
        scale = max(1.0, ref.abs().max().item() / 4.0)
        print(f"crossembed mfma {case}: max|d| = {err:.2e} (gate {2e-5 * scale:.2e}, |ref|max {ref.abs().max().item():.3g})")
        assert err < 2e-5 * scale
        check_stats(ost.cpu(), ref.float())


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(2, 3, 3, 0, 64, 64, 9, False), (2, 3, 6, 3, 40, 72, 8, True), (1, 3, 3, 0, 72, 136, 8, True)])
def test_crossembed_bf16_output(backend, case):
    """reduced-precision configuration of CrossEmbed (tile_cfg | 0x400: single fp16 term; out_st = 1: the result is STORED as bf16, the
    addend -- the hoisted low-res half -- is READ as bf16); statistics stay fp32 and are taken before the rounding.  Error budget vs
    torch fp64 on the same inputs: single fp16 term (2^-11 per operand) + bf16 rounding of the stored result (2^-9 relative)"""
    dev = setup(backend)
    lib = L.lib()
    B, Cin, Cw, c0, H, W, cfg, with_add = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    rn = lambda *s_: torch.randn(*s_, generator=g)
    ks, cout = (3, 7, 15), (4, 2, 2)
    x = rn(B, Cin, H, W)
    ws = [rn(co, Cw, k, k) * 0.1 for k, co in zip(ks, cout)]
    bs = [rn(co) for co in cout]
    add = (rn(B, 8, H, W)).to(torch.bfloat16) if with_add else None
    tab, exps = P.pack_crossembed_mfma(ws, c0, Cin)
    ref = torch.cat([F.conv2d(x.double(), w[:, c0:c0 + Cin].double(), b.double(), padding=(k - 1) // 2) for w, b, k in zip(ws, bs, ks)], 1)
    if with_add:
        ref = ref + add.double()
    p = L.MiCrossEmbedParams()
    p.B, p.H, p.W = B, H, W
    xd, tabd = x.to(dev), tab.to(dev)
    p.in0, p.C0, p.in0_batch_mod = xd.data_ptr(), Cin, 0
    p.n_kernels = 3
    bd = [b.to(dev) for b in bs]
    for i in range(3):
        p.ksize[i], p.cout[i], p.bias[i], p.w_mfma_exp[i] = ks[i], cout[i], bd[i].data_ptr(), exps[i]
    p.w_mfma = tabd.data_ptr()
    addd = add.to(dev) if with_add else None
    p.addend = addd.data_ptr() if with_add else 0
    nt = tile_nt(lib, cfg, H, W)
    out = torch.full(ref.shape, float('nan'), device=dev).to(torch.bfloat16)
    ost = torch.zeros(B, 8, nt, 2, dtype=torch.float64, device=dev)
    p.out, p.out_stats, p.out_st, p.tile_cfg = out.data_ptr(), ost.data_ptr(), 1, cfg | 0x400
    L.check(lib.mi_crossembed_fwd(C.byref(p), L.current_stream()), "crossembed (bf16 storage)")
    err = (out.cpu().double() - ref).abs()
    tol = 2e-3 * ref.abs().max().item() + (2.0 ** -8) * ref.abs()
    print(f"crossembed bf16 out {case}: max|d| = {err.max().item():.2e} (|ref|max {ref.abs().max().item():.3g})")
    assert bool((err <= tol).all())
    assert (out.cpu().float() - ref.float()).abs().max() > 1e-5          # the output really is bf16-rounded
    check_stats(ost.cpu(), ref.float(), rtol=2e-3)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(2, 16, 256, 8, 2, 6), (8, 16, 200, 8, 4, 6), (2, 16, 200, 8, 4, 6), (1, 8, 128, 8, 2, 6), (1, 8, 200, 8, 2, 6), (1, 32, 128, 16, 2, 6),
                                  # range safety of the fp16 split (variant 6): checkpoint weights / context rows far from unit scale
                                  (1, 16, 128, 8, 2, 6, 256.0, 1.0 / 256, 1.0), (1, 16, 128, 8, 2, 6, 1.0 / 64, 300.0, 40.0), (1, 8, 128, 8, 4, 6, 30.0, 30.0, 0.01)])
def test_cross_attention_folded(backend, case):
    """K9 against the oracle's unfolded CrossAttention (+ residual), incl. a ragged token count and both context lengths."""
    dev = setup(backend)
    lib = L.lib()
    B2, Cc, HW, cd, ntok, variant = case[:6]
    q_scale, v_scale, c_scale = case[6:] if len(case) > 6 else (1.0, 1.0, 1.0)
    heads, J = 8, 1 + ntok + 256
    g = torch.Generator().manual_seed(1)
    rn = lambda *s: torch.randn(*s, generator=g)
    kv_w = rn(2 * heads * 64, cd) * cd ** -0.5
    kv_w[heads * 64:] *= v_scale                                 # the value rows
    sd = {"a.norm.gamma": 1 + 0.2 * rn(Cc), "a.norm.beta": 0.1 * rn(Cc),
          "a.to_q.weight": rn(heads * 64, Cc) * Cc ** -0.5 * q_scale, "a.to_kv.weight": kv_w,
          "a.null_kv": rn(2, 64), "a.to_out.0.weight": rn(Cc, heads * 64) * (heads * 64) ** -0.5,
          "a.to_out.1.gamma": 1 + 0.2 * rn(Cc), "a.to_out.1.beta": 0.1 * rn(Cc)}
    if q_scale != 1.0:
        sd["a.to_q.weight"] = sd["a.to_q.weight"] / (1.0 + 0.3 * q_scale)     # keep the logits of a usable size: the test is about the operands' range
    x, c = rn(B2, Cc, HW) * 1.3, rn(B2, J - 1, cd) * c_scale
    xt = x.permute(0, 2, 1)
    ref = (R.cross_attention(xt, c, sd, "a") + xt).permute(0, 2, 1).contiguous()
    mg, mv, g0, v0 = [t.to(dev) for t in P.fold_cross_attention(sd["a.to_q.weight"], sd["a.to_kv.weight"], sd["a.to_out.0.weight"], sd["a.null_kv"], heads)]
    FR = lib.mi_attn_fragment_floats(Cc)
    gv = torch.zeros(B2, heads, 18 if variant == 6 else 17, 64, FR, device=dev)      # fp16 fragments: tile count padded to even (V chunks per tile pair)
    fp = L.MiAttnFoldParams()
    fp.B2, fp.C, fp.cd, fp.heads, fp.JT, fp.n_blocks = B2, Cc, cd, heads, 17, 1
    fp.frag_f16 = 1 if variant == 6 else 0
    x_exp, g_exp, v_exp = P.attn_f16_exponents(mg, mv, g0, v0, cmax=float(c.abs().max()), xmax=P.layernorm_bound(sd["a.norm.gamma"], sd["a.norm.beta"], Cc))
    if variant == 6:
        fp.blk[0].g_exp, fp.blk[0].v_exp = g_exp, v_exp
    fp.blk[0].mg, fp.blk[0].mv, fp.blk[0].g0, fp.blk[0].v0, fp.blk[0].gv = mg.data_ptr(), mv.data_ptr(), g0.data_ptr(), v0.data_ptr(), gv.data_ptr()
    ct, cx = c[:, :ntok].contiguous().to(dev), c[:, ntok:].contiguous().to(dev)
    fp.c_rows, fp.c_stride_b, fp.row0, fp.nrows, fp.write_null = cx.data_ptr(), 256 * cd, 1 + ntok, 256, 1
    L.check(lib.mi_attn_fold_rows(C.byref(fp), L.current_stream()))
    fp.c_rows, fp.c_stride_b, fp.row0, fp.nrows, fp.write_null = ct.data_ptr(), ntok * cd, 1, ntok, 0
    L.check(lib.mi_attn_fold_rows(C.byref(fp), L.current_stream()))
    ap = L.MiCrossAttnParams()
    ap.B2, ap.C, ap.HW, ap.heads, ap.J = B2, Cc, HW, heads, J
    xd = x.to(dev)
    sdd = {k: v.to(dev) for k, v in sd.items()}
    ap.x, ap.gv = L.MiAct(xd.data_ptr(), Cc, 0, 0, 1.0, 0), gv.data_ptr()
    ap.n1_g, ap.n1_b = sdd["a.norm.gamma"].data_ptr(), sdd["a.norm.beta"].data_ptr()
    ap.n2_g, ap.n2_b = sdd["a.to_out.1.gamma"].data_ptr(), sdd["a.to_out.1.beta"].data_ptr()
    nt = -(-HW // (128 if variant in (0, 5) else 64))
    out = torch.full(x.shape, float('nan'), device=dev)
    ost = torch.zeros(B2, Cc, nt, 2, dtype=torch.float64, device=dev)
    ap.out, ap.out_stats, ap.variant = out.data_ptr(), ost.data_ptr(), variant
    if variant == 6:
        ap.x_exp, ap.g_exp, ap.v_exp = x_exp, g_exp, v_exp
    L.check(lib.mi_cross_attn_fwd(C.byref(ap), L.current_stream()))
    assert torch.isfinite(out).all()
    assert (out.cpu() - ref).abs().max().item() < 3e-5
    check_stats(ost.cpu(), ref)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(2, 16, 256, 8, 4), (1, 16, 200, 8, 2), (1, 8, 128, 8, 2), (1, 32, 128, 16, 2)])
def test_cross_attention_bf16_io(backend, case):
    """reduced-precision configuration of K9 (variant 7: single fp16 term) with the tokens READ as bf16 (x.st = 1) and the result
    STORED as bf16 (out_st = 1) -- what the BASELINE U-Nets run under _precision="half".  Against the oracle's unfolded fp32
    CrossAttention + residual on the same bf16-rounded tokens: single-term products (2^-11 per operand) + bf16 rounding of the output"""
    dev = setup(backend)
    lib = L.lib()
    B2, Cc, HW, cd, ntok = case
    heads, J = 8, 1 + ntok + 256
    g = torch.Generator().manual_seed(3)
    rn = lambda *s: torch.randn(*s, generator=g)
    sd = {"a.norm.gamma": 1 + 0.2 * rn(Cc), "a.norm.beta": 0.1 * rn(Cc),
          "a.to_q.weight": rn(heads * 64, Cc) * Cc ** -0.5, "a.to_kv.weight": rn(2 * heads * 64, cd) * cd ** -0.5,
          "a.null_kv": rn(2, 64), "a.to_out.0.weight": rn(Cc, heads * 64) * (heads * 64) ** -0.5,
          "a.to_out.1.gamma": 1 + 0.2 * rn(Cc), "a.to_out.1.beta": 0.1 * rn(Cc)}
    x16 = (rn(B2, Cc, HW) * 1.3).to(torch.bfloat16)
    x, c = x16.float(), rn(B2, J - 1, cd)
    xt = x.permute(0, 2, 1)
    ref = (R.cross_attention(xt, c, sd, "a") + xt).permute(0, 2, 1).contiguous()
    mg, mv, g0, v0 = [t.to(dev) for t in P.fold_cross_attention(sd["a.to_q.weight"], sd["a.to_kv.weight"], sd["a.to_out.0.weight"], sd["a.null_kv"], heads)]
    FR = lib.mi_attn_fragment_floats(Cc)
    gv = torch.zeros(B2, heads, 18, 64, FR, device=dev)
    fp = L.MiAttnFoldParams()
    fp.B2, fp.C, fp.cd, fp.heads, fp.JT, fp.n_blocks, fp.frag_f16 = B2, Cc, cd, heads, 17, 1, 1
    x_exp, g_exp, v_exp = P.attn_f16_exponents(mg, mv, g0, v0, cmax=float(c.abs().max()), xmax=P.layernorm_bound(sd["a.norm.gamma"], sd["a.norm.beta"], Cc))
    fp.blk[0].g_exp, fp.blk[0].v_exp = g_exp, v_exp
    fp.blk[0].mg, fp.blk[0].mv, fp.blk[0].g0, fp.blk[0].v0, fp.blk[0].gv = mg.data_ptr(), mv.data_ptr(), g0.data_ptr(), v0.data_ptr(), gv.data_ptr()
    ct, cx = c[:, :ntok].contiguous().to(dev), c[:, ntok:].contiguous().to(dev)
    fp.c_rows, fp.c_stride_b, fp.row0, fp.nrows, fp.write_null = cx.data_ptr(), 256 * cd, 1 + ntok, 256, 1
    L.check(lib.mi_attn_fold_rows(C.byref(fp), L.current_stream()))
    fp.c_rows, fp.c_stride_b, fp.row0, fp.nrows, fp.write_null = ct.data_ptr(), ntok * cd, 1, ntok, 0
    L.check(lib.mi_attn_fold_rows(C.byref(fp), L.current_stream()))
    ap = L.MiCrossAttnParams()
    ap.B2, ap.C, ap.HW, ap.heads, ap.J = B2, Cc, HW, heads, J
    xd = x16.to(dev)
    sdd = {k: v.to(dev) for k, v in sd.items()}
    ap.x, ap.gv = L.MiAct(xd.data_ptr(), Cc, 0, 0, 1.0, 0, 1), gv.data_ptr()
    ap.n1_g, ap.n1_b = sdd["a.norm.gamma"].data_ptr(), sdd["a.norm.beta"].data_ptr()
    ap.n2_g, ap.n2_b = sdd["a.to_out.1.gamma"].data_ptr(), sdd["a.to_out.1.beta"].data_ptr()
    out = torch.full(x.shape, float('nan'), device=dev).to(torch.bfloat16)
    ost = torch.zeros(B2, Cc, -(-HW // 64), 2, dtype=torch.float64, device=dev)
    ap.out, ap.out_stats, ap.out_st, ap.variant = out.data_ptr(), ost.data_ptr(), 1, 7
    ap.x_exp, ap.g_exp, ap.v_exp = x_exp, g_exp, v_exp
    L.check(lib.mi_cross_attn_fwd(C.byref(ap), L.current_stream()))
    o = out.cpu().float()
    assert torch.isfinite(o).all()
    err = (o - ref).abs()
    tol = 4e-3 * ref.abs().max().item() + (2.0 ** -8) * ref.abs()
    print(f"cross-attention bf16 I/O {case}: max|d| = {err.max().item():.2e} (|ref|max {ref.abs().max().item():.3g})")
    assert bool((err <= tol).all())
    assert err.max() > 1e-5                                              # reduced precision really ran
    check_stats(ost.cpu(), ref, rtol=4e-3)


FLASH_CASES = [
    # B, HW, heads, kv_heads, n0 (first segment), n1 (second segment), null row, q scale, k scale, v scale
    (2, 100, 2, 2, 4, 40, True, 1.0, 1.0, 1.0),            # CrossAttention: null + time tokens + text rows, per-head k / v, ragged HW and J
    (1, 64, 3, 1, 64, 0, True, 1.0, 1.0, 1.0),             # multi-query self-attention: one shared k / v head, context = the tokens themselves
    (2, 150, 8, 1, 150, 0, True, 1.0, 1.0, 1.0),           # multi-query with heads % 4 == 0: four heads per workgroup share the staged chunks
    (1, 70, 4, 1, 70, 60, False, 30.0, 1.0 / 64, 300.0),   # ... three chunks over two segments, no null row, far from unit scale
    (1, 80, 2, 2, 130, 0, False, 30.0, 1.0 / 64, 300.0),   # three chunks, no null row, operands far from unit scale
    (1, 16, 1, 1, 3, 0, True, 1.0 / 256, 256.0, 1.0 / 512),
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", FLASH_CASES)
def test_flash_attention_unfolded(backend, case):
    """mi_flash_attn_fwd (dim_head 64, online softmax over 64-row chunks, null k / v + two context segments, per-head or multi-query k / v)
    against softmax(q k^T * scale) v in fp64"""
    dev = setup(backend)
    lib = L.lib()
    B, HW, heads, kvh, n0, n1, has_null, qs, ks, vs = case
    g = torch.Generator().manual_seed(11)
    rn = lambda *s_: torch.randn(*s_, generator=g)
    inner, kin = heads * 64, kvh * 64
    q = rn(B, HW, inner) * qs
    kv0 = rn(B, n0, 2 * kin); kv0[..., :kin] *= ks; kv0[..., kin:] *= vs
    kv1 = rn(B, max(n1, 1), 2 * kin); kv1[..., :kin] *= ks; kv1[..., kin:] *= vs
    null = rn(2, 64); null[0] *= ks; null[1] *= vs
    scale = 64 ** -0.5 / (qs * ks)                          # keeps the logits O(1) whatever the operand scales: the test is about range, not about saturation
    ref = torch.zeros(B, HW, inner, dtype=torch.float64)
    for b in range(B):
        for h in range(heads):
            kh = h if kvh > 1 else 0
            ksegs, vsegs = [], []
            if has_null:
                ksegs.append(null[0:1]); vsegs.append(null[1:2])
            ksegs.append(kv0[b, :, kh * 64:(kh + 1) * 64]); vsegs.append(kv0[b, :, kin + kh * 64:kin + (kh + 1) * 64])
            if n1:
                ksegs.append(kv1[b, :, kh * 64:(kh + 1) * 64]); vsegs.append(kv1[b, :, kin + kh * 64:kin + (kh + 1) * 64])
            K_, V_ = torch.cat(ksegs).double(), torch.cat(vsegs).double()
            sim = q[b, :, h * 64:(h + 1) * 64].double() @ K_.t() * scale
            ref[b, :, h * 64:(h + 1) * 64] = torch.softmax(sim, -1) @ V_
    qd, kv0d, kv1d, nulld = q.to(dev), kv0.to(dev), kv1.to(dev), null.to(dev)
    out = torch.full((B, HW, inner), float('nan'), device=dev)
    p = L.MiFlashAttnParams()
    p.B, p.HW, p.heads, p.kv_heads, p.q, p.q_scale = B, HW, heads, kvh, L.ptr(qd), scale * P.LOG2E
    if has_null:
        p.null_k, p.null_v = L.ptr(nulld), L.ptr(nulld) + 4 * 64
    p.k0, p.v0, p.n0, p.ld0, p.bs0 = L.ptr(kv0d), L.ptr(kv0d) + 4 * kin, n0, 2 * kin, n0 * 2 * kin
    if n1:
        p.k1, p.v1, p.n1, p.ld1, p.bs1 = L.ptr(kv1d), L.ptr(kv1d) + 4 * kin, n1, 2 * kin, n1 * 2 * kin
    p.out = L.ptr(out)
    L.check(lib.mi_flash_attn_fwd(C.byref(p), L.current_stream()), "flash")
    err = (out.cpu().double() - ref).abs().max().item()
    gate = 2e-5 * max(1.0, ref.abs().max().item())
    print(f"flash attention {case}: max|d| = {err:.2e} (gate {gate:.2e})")
    assert torch.isfinite(out).all() and err < gate
    if (kvh == 1 and heads % 4 == 0) or kvh == heads:
        # the same launch with its K / V operands prepared once (flash_kv_prep_kernel + LDS-DMA): the same arithmetic, the same bits
        J = (1 if has_null else 0) + n0 + n1
        nbytes = lib.mi_flash_kv_prep_bytes(B * kvh, J)
        assert nbytes == B * kvh * ((J + 63) // 64) * (4 * 64 * 8 * 16 + 8)
        prep = torch.full(((nbytes + 3) // 4,), float('nan'), device=dev)
        out2 = torch.full((B, HW, inner), float('nan'), device=dev)
        p.out, p.kv_prep, p.kv_prep_bytes = L.ptr(out2), L.ptr(prep), nbytes
        L.check(lib.mi_flash_attn_fwd(C.byref(p), L.current_stream()), "flash (prepared k / v)")
        assert torch.equal(out2.cpu(), out.cpu())
        p.kv_prep_bytes = nbytes - 1
        assert lib.mi_flash_attn_fwd(C.byref(p), L.current_stream()) != 0


@pytest.mark.parametrize("backend", BACKENDS)
def test_quantile_bit_exact(backend):
    """K12: exact order statistics + torch's fp32 rank arithmetic + fused lerp == torch.quantile, incl. ties and n = 3*256^2."""
    dev = setup(backend)
    lib = L.lib()
    g = torch.Generator().manual_seed(5)
    sizes = (48, 1000, 1001, 12288, 196608) + ((3145728,) if backend == "gpu" else ())      # 1001: the 4-byte access path (n % 4 != 0)
    for n in sizes:
        for ties in (False, True):
            B = 3
            x0 = torch.randn(B, n, generator=g) * 2
            if ties:
                x0 = (x0 * 4).round() / 4
            k_lo, k_hi, w = quantile_rank(n, 0.9)
            assert (k_lo, w) == tuple(float(v) if i else v for i, v in enumerate(R.quantile_rank(n, 0.9)))
            x0d = x0.to(dev)
            hist = torch.zeros(3 * B * 2 * 2048, dtype=torch.int32, device=dev)
            s, v = torch.zeros(B, device=dev), torch.zeros(B, 2, device=dev)
            p = L.MiQuantileParams(B, n, x0d.data_ptr(), k_lo, k_hi, w, hist.data_ptr(), s.data_ptr(), v.data_ptr())
            L.check(lib.mi_quantile_fwd(C.byref(p), L.current_stream()))
            srt = x0.abs().sort(-1).values
            assert torch.equal(s.cpu(), torch.quantile(x0.abs(), 0.9, dim=-1)), (n, ties)
            assert torch.equal(v.cpu()[:, 0], srt[:, k_lo]) and torch.equal(v.cpu()[:, 1], srt[:, k_hi])


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("side", [16, 15])
def test_sampler_elementwise_bit_exact(backend, side):
    """K11 epilogue + K13 + schedule look-ups against the oracle, bit for bit (t = 13 and the t = 0 no-noise step); side 15: an image
    whose element count is not a multiple of 4 (the kernels' 4-byte access path)"""
    dev = setup(backend)
    lib = L.lib()
    from minimagen_amd.diffusion_model import GaussianDiffusion
    T, B, n = 25, 2, 3 * side * side
    sched = R.Schedule(T)
    gd = GaussianDiffusion(timesteps=T)
    for k in ("sqrt_recip_alphas_cumprod", "posterior_mean_coef1", "posterior_log_variance_clipped", "sqrt_alphas_cumprod"):
        assert torch.equal(getattr(gd, k), getattr(sched, k))
    coef = gd.sampler_coef_table().to(dev)
    g = torch.Generator().manual_seed(9)
    for t in (13, 0):
        tstate = torch.tensor([t], dtype=torch.int32, device=dev)
        pred2, xt, noise = torch.randn(2 * B, n, generator=g), torch.randn(B, n, generator=g), torch.randn(T, B, n, generator=g)
        pred2d, xtd, noised = pred2.to(dev), xt.to(dev), noise.to(dev)
        x0, pg = torch.zeros(B, n, device=dev), torch.zeros(B, n, device=dev)
        p = L.MiCfgX0Params(B, n, pred2d.data_ptr(), 1, 3.0, xtd.data_ptr(), coef.data_ptr(), tstate.data_ptr(), pg.data_ptr(), x0.data_ptr())
        L.check(lib.mi_cfg_x0_fwd(C.byref(p), L.current_stream()))
        pred = pred2[B:] + (pred2[:B] - pred2[B:]) * 3.0
        assert torch.equal(pg.cpu(), pred)
        assert torch.equal(x0.cpu(), sched.predict_start_from_noise(xt, t, pred))
        k_lo, k_hi, w = quantile_rank(n, 0.9)
        hist = torch.zeros(3 * B * 2 * 2048, dtype=torch.int32, device=dev)
        s = torch.zeros(B, device=dev)
        qp = L.MiQuantileParams(B, n, x0.data_ptr(), k_lo, k_hi, w, hist.data_ptr(), s.data_ptr(), None)
        L.check(lib.mi_quantile_fwd(C.byref(qp), L.current_stream()))
        # the fused form the sampler uses: pass 0 of the radix select rides on the kernel that writes x0, and the histograms
        # (zero on entry) are left zeroed -> same threshold bits, twice in a row without a memset
        hist2 = torch.zeros_like(hist)
        s2, x0b = torch.zeros(B, device=dev), torch.zeros(B, n, device=dev)
        for _ in range(2):
            pf = L.MiCfgX0Params(B, n, pred2d.data_ptr(), 1, 3.0, xtd.data_ptr(), coef.data_ptr(), tstate.data_ptr(), 0, x0b.data_ptr(), hist2.data_ptr())
            L.check(lib.mi_cfg_x0_fwd(C.byref(pf), L.current_stream()))
            s2.zero_()
            qf = L.MiQuantileParams(B, n, x0b.data_ptr(), k_lo, k_hi, w, hist2.data_ptr(), s2.data_ptr(), None, 1, 1)
            L.check(lib.mi_quantile_fwd(C.byref(qf), L.current_stream()))
            assert torch.equal(x0b, x0) and torch.equal(s2, s) and int(hist2.abs().sum()) == 0
        x = xtd.clone()
        pp = L.MiPosteriorParams(B, n, T, x0.data_ptr(), s.data_ptr(), x.data_ptr(), coef.data_ptr(), tstate.data_ptr(), noised.data_ptr(), 0, 0, 0)
        L.check(lib.mi_posterior_fwd(C.byref(pp), L.current_stream()))
        xr, _ = R.p_sample(None, sched, xt.reshape(B, 3, side, side), t, noise[T - 1 - t].reshape(B, 3, side, side), pred=pred.reshape(B, 3, side, side))
        assert torch.equal(x.cpu().reshape(B, 3, side, side), xr)
        # the whole tail in one launch (images that fit a workgroup's registers: the base stage), addressed as step *t_state - t_off:
        # same bits as the three launches, with the injected noise and with the on-device generator
        for off in (0, 2):
            ts2 = torch.tensor([t + off], dtype=torch.int32, device=dev)
            for use_noise in (True, False):
                xa, xb = xtd.clone(), xtd.clone()
                sf, vf, pgf = torch.zeros(B, device=dev), torch.zeros(B, 2, device=dev), torch.zeros(B, n, device=dev)
                nzp = noised.data_ptr() if use_noise else 0
                ppa = L.MiPosteriorParams(B, n, T, x0.data_ptr(), s.data_ptr(), xa.data_ptr(), coef.data_ptr(), tstate.data_ptr(), nzp, 77, 5, 3 << 20)
                L.check(lib.mi_posterior_fwd(C.byref(ppa), L.current_stream()))
                cf = L.MiCfgX0Params(B, n, pred2d.data_ptr(), 1, 3.0, xb.data_ptr(), coef.data_ptr(), ts2.data_ptr(), pgf.data_ptr(), 0, 0, off)
                qf = L.MiQuantileParams(B, n, 0, k_lo, k_hi, w, 0, sf.data_ptr(), vf.data_ptr(), 0, 0)
                ppf = L.MiPosteriorParams(B, n, T, 0, 0, xb.data_ptr(), coef.data_ptr(), ts2.data_ptr(), nzp, 77, 5, 3 << 20, 0, off)
                L.check(lib.mi_sampler_step_small_fwd(C.byref(cf), C.byref(qf), C.byref(ppf), L.current_stream()), "fused tail")
                assert torch.equal(xb, xa) and torch.equal(sf, s) and torch.equal(pgf, pg), (off, use_noise)
    # timestep bookkeeping (bit-exact integers)
    times = torch.zeros(5, dtype=torch.int64, device=dev)
    ts = torch.zeros(1, dtype=torch.int32, device=dev)
    lib.mi_step_set(ts.data_ptr(), times.data_ptr(), 5, 24, L.current_stream())
    assert ts.item() == 24 and times.tolist() == [24] * 5
    lib.mi_step_advance(ts.data_ptr(), times.data_ptr(), 5, L.current_stream())
    assert ts.item() == 23 and times.tolist() == [23] * 5
    lib.mi_step_advance_by(ts.data_ptr(), times.data_ptr(), 5, 5, L.current_stream())
    assert ts.item() == 18 and times.tolist() == [18] * 5
    xx = torch.randn(1000, generator=g) * 2
    oo = torch.zeros(1000, device=dev)
    lib.mi_finalize_images(xx.to(dev).data_ptr(), oo.data_ptr(), 1000, 1, L.current_stream())
    assert torch.equal(oo.cpu(), (xx.clamp(-1, 1) + 1) * 0.5)


GROUP_SIDES = [96, 160]


def _group_tail_case(dev, side, B=2, T=25, rounds=3, nan_row=False, two=1, ties=False):
    """mi_sampler_step_group_fwd against mi_cfg_x0_fwd + mi_quantile_fwd + mi_posterior_fwd on the same inputs: bit for bit, several launches in
    a row on one sync buffer (both parities of the double-buffered histograms, the step offset of multi-step graphs)."""
    lib = L.lib()
    from minimagen_amd.diffusion_model import GaussianDiffusion
    n = 3 * side * side
    G = lib.mi_sampler_group_size(n)
    assert G >= 2
    coef = GaussianDiffusion(timesteps=T).sampler_coef_table().to(dev)
    g = torch.Generator().manual_seed(side)
    k_lo, k_hi, w = quantile_rank(n, 0.9)
    sync = torch.zeros(lib.mi_sampler_group_sync_bytes(B, n), dtype=torch.uint8, device=dev)
    hist = torch.zeros(3 * B * 2 * 2048, dtype=torch.int32, device=dev)
    noise = torch.randn(T, B, n, generator=g).to(dev)
    for r in range(rounds):
        t, off = (13, 0) if r == 0 else ((0, 2) if r == 1 else (7, 1))
        use_noise = r != 2
        pred2 = (torch.randn(2 * B, n, generator=g) * (1.0 + r)).to(dev)
        xt = torch.randn(B, n, generator=g).to(dev)
        if ties:                    # heavy ties around the order statistics (and a constant image): the rank-inside-the-bin bookkeeping of all three passes
            pred2, xt = (pred2 * 2).round() / 2, (xt * 2).round() / 2
            if r == 2:
                pred2[:, :] = 0.25
                xt[0, :] = -1.5
        if nan_row and r == 1:
            pred2[0, 5] = float("nan")
        ts, ts2 = torch.tensor([t], dtype=torch.int32, device=dev), torch.tensor([t + off], dtype=torch.int32, device=dev)
        nzp = noise.data_ptr() if use_noise else 0
        # the separate kernels
        xa, x0, pg, s, v = xt.clone(), torch.zeros(B, n, device=dev), torch.zeros(B, n, device=dev), torch.zeros(B, device=dev), torch.zeros(B, 2, device=dev)
        cp = L.MiCfgX0Params(B, n, pred2.data_ptr(), two, 3.0, xa.data_ptr(), coef.data_ptr(), ts.data_ptr(), pg.data_ptr(), x0.data_ptr(), hist.data_ptr())
        L.check(lib.mi_cfg_x0_fwd(C.byref(cp), L.current_stream()))
        qp = L.MiQuantileParams(B, n, x0.data_ptr(), k_lo, k_hi, w, hist.data_ptr(), s.data_ptr(), v.data_ptr(), 1, 1)
        L.check(lib.mi_quantile_fwd(C.byref(qp), L.current_stream()))
        pp = L.MiPosteriorParams(B, n, T, x0.data_ptr(), s.data_ptr(), xa.data_ptr(), coef.data_ptr(), ts.data_ptr(), nzp, 77, 5, 3 << 20)
        L.check(lib.mi_posterior_fwd(C.byref(pp), L.current_stream()))
        # one launch
        xb, x0g, pgg, sg, vg = xt.clone(), torch.zeros(B, n, device=dev), torch.zeros(B, n, device=dev), torch.zeros(B, device=dev), torch.zeros(B, 2, device=dev)
        cf = L.MiCfgX0Params(B, n, pred2.data_ptr(), two, 3.0, xb.data_ptr(), coef.data_ptr(), ts2.data_ptr(), pgg.data_ptr() if r else 0, x0g.data_ptr() if r else 0, 0, off)
        qf = L.MiQuantileParams(B, n, 0, k_lo, k_hi, w, 0, sg.data_ptr(), vg.data_ptr(), 0, 0)
        pf = L.MiPosteriorParams(B, n, T, 0, 0, xb.data_ptr(), coef.data_ptr(), ts2.data_ptr(), nzp, 77, 5, 3 << 20, 0, off)
        L.check(lib.mi_sampler_step_group_fwd(C.byref(cf), C.byref(qf), C.byref(pf), sync.data_ptr(), L.current_stream()), "grouped tail")
        eq = (lambda a, b: torch.equal(a, b)) if not (nan_row and r == 1) else (lambda a, b: torch.equal(torch.nan_to_num(a, nan=7e7), torch.nan_to_num(b, nan=7e7)))
        assert eq(sg, s) and eq(vg, v), (side, r, sg, s)
        assert eq(xb, xa), (side, r)
        if r:
            assert eq(x0g, x0) and eq(pgg, pg)
        if nan_row and r == 1:
            assert torch.isnan(sg[0]) and not torch.isnan(sg[1])
        words = sync[:16].cpu().view(torch.int64)
        assert int(words[0]) == (r + 1) * B * G and int(words[1]) == 0            # tickets taken, error word clear
    lay_hist = sync[64 + ((B * 8 + 63) & ~63):].view(torch.int32).view(2, B, 5, 2048)
    assert int(lay_hist[rounds & 1].abs().sum()) == 0                            # the parity the next launch uses is clean


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("side", GROUP_SIDES)
def test_sampler_group_tail_bit_exact(backend, side):
    dev = setup(backend)
    _group_tail_case(dev, side, nan_row=(side == 96))
    if side == 96:
        _group_tail_case(dev, side, ties=True)
        _group_tail_case(dev, side, B=1, rounds=2, two=0)


@pytest.mark.gpu
def test_sampler_group_tail_bit_exact_at_the_sr_stage_size():
    """the 256^2 stage of the headline cascade (B = 32, G = 8: 256 workgroups), without guidance too"""
    dev = setup("gpu")
    _group_tail_case(dev, 256, B=32, rounds=3)
    _group_tail_case(dev, 256, B=3, rounds=3, two=0)
    lib = L.lib()
    assert lib.mi_sampler_group_size(3 * 256 * 256) == 8 and lib.mi_sampler_group_size(3 * 1024 * 1024) == 128
    assert lib.mi_sampler_group_size(3 * 15 * 15) == 0


@pytest.mark.parametrize("backend", BACKENDS)
def test_randn_keyed_by_global_sample(backend):
    dev = setup(backend)
    lib = L.lib()
    z = torch.zeros(4, 50001, device=dev)
    lib.mi_randn_fill(z.data_ptr(), 4, 50001, 1234, 0, 7, L.current_stream())
    z2 = torch.zeros(2, 50001, device=dev)
    lib.mi_randn_fill(z2.data_ptr(), 2, 50001, 1234, 2, 7, L.current_stream())
    assert torch.equal(z2, z[2:])                      # a shard starting at global row 2 reproduces rows 2..3
    zc = z.cpu()
    assert abs(zc.mean()) < 0.01 and abs(zc.std() - 1) < 0.01 and abs((zc ** 4).mean() / zc.var() ** 2 - 3) < 0.1


# The reflect-padded border of resize_right's cubic x4 up-sampling (64 -> 256), derived BY HAND from the published algorithm and not
# from oracle/resize_restated.py: grid[o] = o/4 - 3/8, so output 0 sits 0.375 left of input 0 and its four taps are the inputs
# -2, -1, 0, 1 at distances 1.625, 0.625, 0.375, 1.375; output 1 (grid -0.125): distances 1.875, 0.875, 0.125, 1.125.  Keys' cubic
# (a = -1/2) at those distances, all exact in binary; 'reflect' maps input -1 -> 1 and -2 -> 2 (no edge repeat); the far end mirrors.
BORDER_TAPS_X4 = {
    0: ((2, 1, 0, 1), (-0.0439453125, 0.3896484375, 0.7275390625, -0.0732421875)),
    1: ((2, 1, 0, 1), (-0.0068359375, 0.0908203125, 0.9638671875, -0.0478515625)),
}


def border_closed_form_check(dev, lib):
    """the 4 x 4 corner outputs of a 64 -> 256 resize against the hand-derived border taps (both passes closed-form: no restatement involved)"""
    from minimagen_amd.helpers import cubic_taps
    n, N = 64, 256
    _, idx, w = cubic_taps(n, N)
    taps = {}
    for o, (ii, ww) in BORDER_TAPS_X4.items():
        taps[o] = (ii, ww)
        taps[N - 1 - o] = (tuple(n - 1 - i for i in reversed(ii)), tuple(reversed(ww)))       # mirror image at the far end
    for o, (ii, ww) in taps.items():       # the host tap tables first: indices after reflection and weights, exactly
        assert tuple(idx[o].tolist()) == ii and tuple(w[o].tolist()) == ww, (o, idx[o], w[o])
    g = torch.Generator().manual_seed(9)
    img = torch.rand(1, 2, n, n, generator=g)
    tabs = [t.to(dev) for t in (idx, w, idx, w)]
    imgd, up = img.to(dev), torch.zeros(1, 2, N, N, device=dev)
    rp = L.MiResizeParams(2, n, n, N, N, idx.shape[1], idx.shape[1], imgd.data_ptr(), up.data_ptr(),
                          tabs[0].data_ptr(), tabs[1].data_ptr(), tabs[2].data_ptr(), tabs[3].data_ptr())
    L.check(lib.mi_resize_fwd(C.byref(rp), L.current_stream()))
    up = up.cpu().double()
    x = img.double()
    for oy, (iy, wy) in taps.items():
        for ox, (ix, wx) in taps.items():
            ref = sum(a * b * x[0, :, r, c] for r, a in zip(iy, wy) for c, b in zip(ix, wx))
            assert (up[0, :, oy, ox] - ref).abs().max() < 5e-7, (oy, ox)


@pytest.mark.parametrize("backend", BACKENDS)
def test_resize_and_lowres_augment(backend):
    """K14 against the oracle's restatement of resize_right and q_sample, bit for bit; the restatement itself is pinned in the interior
    against Pillow's bicubic (tests/test_oracle.py) and at the reflect-padded border against hand-derived closed-form taps (below)."""
    dev = setup(backend)
    lib = L.lib()
    from minimagen_amd.helpers import cubic_taps
    from oracle import resize_restated as RR
    g = torch.Generator().manual_seed(2)
    img = torch.rand(2, 3, 16, 16, generator=g)
    for out_sz in (32, 64):
        _, ih, wh = cubic_taps(16, out_sz)
        ref = RR.resize(img, scale_factors=out_sz / 16, pad_mode='reflect')
        tabs = [t.to(dev) for t in (ih, wh, ih, wh)]
        imgd, up = img.to(dev), torch.zeros(2, 3, out_sz, out_sz, device=dev)
        rp = L.MiResizeParams(6, 16, 16, out_sz, out_sz, ih.shape[1], ih.shape[1], imgd.data_ptr(), up.data_ptr(),
                              tabs[0].data_ptr(), tabs[1].data_ptr(), tabs[2].data_ptr(), tabs[3].data_ptr())
        L.check(lib.mi_resize_fwd(C.byref(rp), L.current_stream()))
        assert torch.equal(up.cpu(), ref)
    border_closed_form_check(dev, lib)
    sched = R.Schedule(100)
    noise = torch.randn(2, 3, 64, 64, generator=g)
    out = torch.zeros(2, 3, 64, 64, device=dev)
    a, b = float(sched.sqrt_alphas_cumprod[20]), float(sched.sqrt_one_minus_alphas_cumprod[20])
    lib.mi_lowres_augment(up.data_ptr(), noise.to(dev).data_ptr(), out.data_ptr(), up.numel(), a, b, 1, L.current_stream())
    assert torch.equal(out.cpu(), sched.q_sample(ref, 20, noise) * 2 - 1)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(130, 128, True), (7, 200, False), (64, 512, True)])
def test_ln_rows(backend, case):
    """mi_ln_rows_fwd (ChanLayerNorm of ChanFeedForward in token layout, layers.py:148-161, 322-343): LayerNorm over the last dimension of [rows][dim],
    a wave per row; ragged row counts and widths, with and without beta, vs torch fp64"""
    dev = setup(backend)
    lib = L.lib()
    rows, dim, has_beta = case
    g = torch.Generator().manual_seed(23)
    x = torch.randn(rows, dim, generator=g) * 3.0 - 1.0
    gamma, beta = 1 + 0.2 * torch.randn(dim, generator=g), 0.1 * torch.randn(dim, generator=g)
    xd = x.double()
    ref = (xd - xd.mean(-1, keepdim=True)) / torch.sqrt(xd.var(-1, unbiased=False, keepdim=True) + 1e-5) * gamma.double() + (beta.double() if has_beta else 0.0)
    xg, gg, bg = x.to(dev), gamma.to(dev), beta.to(dev)
    out = torch.full((rows, dim), float('nan'), device=dev)
    L.check(lib.mi_ln_rows_fwd(xg.data_ptr(), gg.data_ptr(), bg.data_ptr() if has_beta else None, out.data_ptr(), rows, dim, 1e-5, L.current_stream()), "mi_ln_rows_fwd")
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(2, 100, 96, 1.0), (1, 256, 512, 0.7071), (3, 37, 40, 1.0)])
def test_ln_tokens(backend, case):
    """mi_ln_tokens_fwd (the LayerNorm in front of the wide attention blocks, layers.py:322-343 / 14-104): NCHW activation (x its scale) -> per-token
    LayerNorm over the channels -> token rows; ragged token / channel counts, few-token launches (channel blocks split over the grid) vs torch fp64"""
    dev = setup(backend)
    lib = L.lib()
    B, HW, Cc, scale = case
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, Cc, HW, generator=g) * 2.0 + 0.5
    gamma, beta = 1 + 0.2 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    xs = x.double().permute(0, 2, 1) * scale
    mean, var = xs.mean(-1, keepdim=True), xs.var(-1, unbiased=False, keepdim=True)
    ref = (xs - mean) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
    xd, gd, bd = x.to(dev).contiguous(), gamma.to(dev), beta.to(dev)
    out = torch.full((B, HW, Cc), float('nan'), device=dev)
    a = L.MiAct(xd.data_ptr(), Cc, 0, 0, scale, 0)
    L.check(lib.mi_ln_tokens_fwd(C.byref(a), B, HW, gd.data_ptr(), bd.data_ptr(), out.data_ptr(), L.current_stream()), "mi_ln_tokens_fwd")
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(2, 100, 96, True, True), (1, 64, 128, True, False), (3, 37, 40, False, True), (1, 256, 512, True, True)])
def test_tokens_to_nchw(backend, case):
    """mi_tokens_to_nchw_fwd (to_out.1 LayerNorm + residual + NCHW + statistics of the wide attention blocks, layers.py:14-104): token rows read
    channel-contiguous, transposed through LDS; ragged token / channel counts, with and without LayerNorm and residual, against torch fp64"""
    dev = setup(backend)
    lib = L.lib()
    B, HW, Cc, ln, res = case
    g = torch.Generator().manual_seed(11)
    tok = torch.randn(B, HW, Cc, generator=g) * 1.3 + 0.4
    gamma, beta = 1 + 0.2 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    r = torch.randn(B, Cc, HW, generator=g)
    ref = tok.double()
    if ln:
        mu = ref.mean(-1, keepdim=True)
        ref = (ref - mu) / torch.sqrt(((ref - mu) ** 2).mean(-1, keepdim=True) + 1e-5) * gamma.double() + beta.double()
    ref = ref.transpose(1, 2)
    if res:
        ref = ref + 0.5 * r.double()
    keep = [t.to(dev).contiguous() for t in (tok, gamma, beta, r)]
    nt = -(-HW // 64)
    out = torch.full((B, Cc, HW), float('nan'), device=dev)
    ost = torch.full((B, Cc, nt, 2), float('nan'), dtype=torch.float64, device=dev)
    p = L.MiTokensToNchwParams()
    p.B, p.HW, p.C, p.tokens = B, HW, Cc, keep[0].data_ptr()
    if ln:
        p.gamma, p.beta, p.eps = keep[1].data_ptr(), keep[2].data_ptr(), 1e-5
    if res:
        p.res = L.MiAct(keep[3].data_ptr(), Cc, 0, 0, 0.5, 0, 0)
    p.out, p.out_stats = out.data_ptr(), ost.data_ptr()
    L.check(lib.mi_tokens_to_nchw_fwd(C.byref(p), L.current_stream()), "mi_tokens_to_nchw_fwd")
    assert (out.cpu().double() - ref).abs().max() < 2e-5
    st = ost.cpu().double()
    assert (st[..., 0].sum(-1) - ref.sum(-1)).abs().max() < 1e-3 and (st[..., 1].sum(-1) - (ref ** 2).sum(-1)).abs().max() < 1e-2
