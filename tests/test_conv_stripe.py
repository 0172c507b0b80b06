"""The full-width-stripe, wave-specialised form of the narrow 3x3 convs (csrc/conv_stripe.hip, tile_cfg 12; round 6) through the C ABI:
against torch fp64 (Block = GroupNorm -> scale/shift -> SiLU -> Conv3x3 + identity / 1x1 residual, layers.py:131-145, 415-439) with the
row-paired family's gates, BIT FOR BIT against the tile kernel it replaces (conv_rp.hip, tile_cfg 6 / 7: same arithmetic, other data
movement; statistics blocks of W / 8 rows), statistics against fp64, and independence of the result from the stripe length a workgroup takes (speed-only knob)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from minimagen_amd import _lib as L, packing as P
from tests._backend import BACKENDS, setup
from tests.test_kernels import chan_stats, check_stats, tile_nt

SK = 2 ** -0.5


def build(case, dev, keep):
    """inputs + fp64 reference + a parameter struct without tile_cfg / out / out_stats"""
    B, C0, C1, Cout, H, W, gn, ss, res, xs, wsc = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    rn = lambda *s_: torch.randn(*s_, generator=g)
    x0 = (rn(B, C0, H, W) * 1.5 + 0.3) * xs
    x1 = rn(B, C1, H, W) * xs if C1 else None
    Cin = C0 + C1
    w, bias = rn(Cout, Cin, 3, 3) * 0.2 * wsc, rn(Cout) * wsc * (1.0 if gn else xs)
    gamma, beta = 1 + 0.2 * rn(Cin), 0.1 * rn(Cin)
    sst = rn(B, 7 + 2 * Cin) * 0.3 if ss else None
    h = torch.cat((x0, x1 * SK), 1) if C1 else x0
    if gn:
        h = F.group_norm(h, 8, gamma, beta, 1e-5)
        if ss:
            h = h * (sst[:, 7:7 + Cin, None, None] + 1) + sst[:, 7 + Cin:7 + 2 * Cin, None, None]
        h = F.silu(h)
    ref = F.conv2d(h.double(), w.double(), bias.double(), padding=1)
    d = lambda name, t: keep.setdefault(name, t.to(dev).contiguous())
    p = L.MiConvParams()
    p.B, p.H, p.W = B, H, W
    p.in0 = L.MiAct(d("x0", x0).data_ptr(), C0, d("s0", chan_stats(x0)).data_ptr(), 1, 1.0, 0)
    if C1:
        p.in1 = L.MiAct(d("x1", x1).data_ptr(), C1, d("s1", chan_stats(x1)).data_ptr(), 1, SK, 0)
    p.Cout, p.ksize, p.stride, p.up2 = Cout, 3, 1, 0
    wf, wexp = P.pack_conv_weight_rp(w)
    p.w_rp, p.w_rp_exp, p.bias = d("wf", wf).data_ptr(), wexp, d("b", bias).data_ptr()
    if gn:
        p.gn_groups, p.gn_gamma, p.gn_beta, p.gn_eps = 8, d("g", gamma).data_ptr(), d("be", beta).data_ptr(), 1e-5
        if ss:
            p.scale_shift, p.ss_stride, p.ss_off = d("ss", sst).data_ptr(), sst.shape[1], 7
    if res != 'none':
        rsc = xs if not gn else 1.0
        if res == 'id':
            r0 = rn(B, Cout, H, W) * rsc
            p.res0 = L.MiAct(d("r0", r0).data_ptr(), Cout, 0, 0, 0.5, 0)            # (a scaled identity residual: mi_act.scale)
            ref = ref + 0.5 * r0
        else:
            cr0, cr1 = res
            r0, r1 = rn(B, cr0, H, W) * rsc, (rn(B, cr1, H, W) * rsc if cr1 else None)
            rin = torch.cat((r0, r1 * SK), 1) if r1 is not None else r0
            rw, rb = rn(Cout, rin.shape[1], 1, 1) * 0.3, rn(Cout)
            ref = ref + F.conv2d(rin.double(), rw.double(), rb.double())
            p.res0 = L.MiAct(d("r0", r0).data_ptr(), cr0, d("rs0", chan_stats(r0)).data_ptr(), 1, 1.0, 0)
            p.res_w = 1
            rwf, rwexp = P.pack_conv_weight_rp(rw)
            p.res_w_rp, p.res_w_rp_exp = d("rwf", rwf).data_ptr(), rwexp
            p.res_b = d("rb", rb).data_ptr()
            if r1 is not None:
                p.res1 = L.MiAct(d("r1", r1).data_ptr(), cr1, d("rs1", chan_stats(r1)).data_ptr(), 1, SK, 0)
    return p, ref


def run(lib, p, cfg, nt, dev):
    out = torch.full((p.B, p.Cout, p.H, p.W), float('nan'), device=dev)
    ost = torch.full((p.B, p.Cout, nt, 2), float('nan'), dtype=torch.float64, device=dev)
    p.out, p.out_stats, p.tile_cfg = out.data_ptr(), ost.data_ptr(), cfg
    L.check(lib.mi_conv_fwd(C.byref(p), L.current_stream()), "mi_conv_fwd")
    return out.cpu(), ost.cpu()


STRIPE_CASES = [
    # B, C0, C1, Cout, H, W, gn, ss, res ('none' | 'id' | (Cres0, Cres1): 1x1 conv over their concat), xscale, wscale
    (1, 8, 0, 8, 64, 256, True, True, 'id', 1.0, 1.0),               # the 256^2 layers of the SR U-Net (final_res_block)
    (1, 8, 0, 3, 32, 256, False, False, 'none', 1.0, 1.0),           # final conv: no GroupNorm, 3 output channels
    (2, 8, 0, 8, 32, 128, True, True, 'id', 1.0, 1.0),
    (1, 8, 8, 8, 16, 128, True, True, 'none', 1.0, 1.0),             # 8 + 8 skip channels -> 8 (SR ups.1 block1)
    (1, 8, 0, 8, 16, 128, True, True, (8, 8), 1.0, 1.0),             # ... its block2 with the 1x1 residual over the 16
    (8, 8, 0, 8, 32, 128, True, True, (8, 8), 1.0 / 64, 30.0),       # ... scaled operands, B % 8 == 0
    (8, 16, 0, 16, 16, 64, True, True, 'id', 1.0, 1.0),              # B % 8 == 0: the XCD-aware workgroup -> image map
    (1, 16, 16, 16, 16, 64, True, True, 'none', 1.0, 1.0),           # 32 -> 16 (SR ups.0)
    (1, 16, 0, 16, 16, 64, True, False, (16, 16), 1.0, 1.0),         # ... and the 1x1 residual over the 32
    (2, 8, 0, 8, 24, 64, True, True, 'none', 1.0, 1.0),              # base U-Net, 64^2 level (H not a power of two)
    (1, 8, 8, 8, 8, 64, True, True, 'none', 1.0, 1.0),
    (1, 8, 0, 8, 16, 64, True, True, (8, 8), 1.0, 1.0),
    (1, 8, 0, 3, 8, 64, False, False, 'none', 1.0, 1.0),
    (2, 8, 0, 8, 32, 32, True, True, 'id', 1.0, 1.0),                # base U-Net, 32^2 level
    (1, 8, 0, 16, 16, 32, False, False, 'none', 1.0, 1.0),           # the folded Parallel(3x3, 1x1) conv 8 -> 16, no GroupNorm
    (1, 16, 0, 16, 12, 32, True, True, 'id', 1.0, 1.0),
    (1, 16, 8, 16, 16, 32, True, True, 'none', 1.0, 1.0),            # 16 + 8 skip channels -> 16
    (1, 16, 0, 16, 8, 32, True, True, (16, 8), 1.0, 1.0),            # ... 1x1 residual over the 24
    (1, 8, 0, 8, 16, 64, True, True, 'id', 256.0, 256.0),            # range safety of the fp16 split: large / small operands
    (1, 16, 0, 16, 16, 64, True, False, (16, 16), 1.0 / 256, 1.0 / 256),
    (1, 8, 0, 8, 8, 128, False, False, 'id', 4096.0, 1.0 / 300),
    (1, 8, 0, 8, 16, 128, False, False, 'none', 4096.0, 1.0 / 300),
    (2, 16, 0, 16, 16, 64, True, True, 'none', 1.0, 1.0),
    (1, 8, 0, 8, 32, 256, True, False, 'none', 1.0, 1.0),
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", STRIPE_CASES)
def test_conv_full_width_stripes(backend, case):
    dev = setup(backend)
    lib = L.lib()
    keep = {}
    p, ref = build(case, dev, keep)
    B, C0, C1, Cout, H, W = case[:6]
    rows = lib.mi_conv_stripe_rows(C.byref(p))
    rconv_ok = (case[8] == (8, 8) and C0 + C1 == 8 and Cout == 8 and W in (64, 128)) or \
        (case[8] == (16, 16) and C0 + C1 == 16 and Cout == 16 and W == 64)      # the instantiated 1x1-residual members (ups.1 of both U-Nets, ups.0 of the SR U-Net)
    if (isinstance(case[8], tuple) and not rconv_ok) or (case[8] == 'id' and not case[6]):
        # launches with a bigger 1x1 residual conv, or an identity residual without a Block in front (no such layer in the U-Nets), stay on the
        # tile kernel: the library says so, the engine asks
        assert rows == 0
        return
    assert rows == W // 8, f"the stripe kernel does not take {case}"
    nt = H // rows
    out, ost = run(lib, p, 12, nt, dev)
    scale = max(1.0, ref.abs().max().item() / 8.0)
    err = (out.double() - ref).abs().max().item()
    print(f"stripe conv {case}: max|d| = {err:.2e} (gate {2e-5 * scale:.2e}, |ref|max {ref.abs().max().item():.3g})")
    assert err < 2e-5 * scale
    check_stats(ost, ref.float())
    # the stripe length a workgroup takes is a speed-only knob: same outputs, same partial statistics, bit for bit
    for nblk in (1, 2, nt):
        if nt % nblk == 0 and nblk <= 15:
            o2, s2 = run(lib, p, 12 | (nblk << 12), nt, dev)
            assert torch.equal(o2, out) and torch.equal(s2, ost), f"{nblk} statistics blocks per workgroup change the result"
    o3, s3 = run(lib, p, 12 | 0x200, nt, dev)                        # MI_CONV_REVERSE: image groups in reverse order (placement only)
    assert torch.equal(o3, out) and torch.equal(s3, ost)
    # the tile kernel it replaces computes the same bits (the same affine, operand split, MFMA order, epilogue); 32-wide images reduce the
    # producers' statistics with 128 instead of 256 work-items (another fp64 summation order): there a last-bit tolerance
    rp_cfg = 6 if W >= 64 else 7
    o_rp, s_rp = run(lib, p, rp_cfg, tile_nt(lib, rp_cfg, H, W), dev)
    if W >= 64:
        assert torch.equal(o_rp, out), f"stripe and tile kernels differ: max|d| = {(o_rp - out).abs().max().item():.3e}"
    else:
        assert (o_rp - out).abs().max().item() < 2e-6 * scale
    assert torch.equal(o_rp == o_rp, out == out)
    assert (s_rp.sum(2) - ost.sum(2)).abs().max().item() <= 1e-6 * s_rp.sum(2).abs().max().item()


UP_CASES = [
    # B, Cin, H, W (output size), xscale, wscale: nearest x2 + Conv3x3 to 8 channels (Upsample, layers.py:512-515), no GroupNorm
    (2, 16, 32, 128, 1.0, 1.0),         # SR U-Net: 16 -> 8 up to 128^2
    (8, 16, 16, 64, 1.0, 1.0),          # base U-Net: 16 -> 8 up to 64^2 (B % 8 == 0)
    (1, 8, 16, 128, 300.0, 1.0 / 64),   # 8 -> 8, scaled operands
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", UP_CASES)
def test_conv_full_width_stripes_upsampling(backend, case):
    """tile_cfg 12 with up2: the ring holds source rows of half the width; against torch fp64 and bit for bit against conv_rp's MODE 1"""
    dev = setup(backend)
    lib = L.lib()
    B, Cin, H, W, xs, wsc = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    rn = lambda *s_: torch.randn(*s_, generator=g)
    x = (rn(B, Cin, H // 2, W // 2) * 1.5 + 0.3) * xs
    w, bias = rn(8, Cin, 3, 3) * 0.2 * wsc, rn(8) * wsc * xs
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), bias.double(), padding=1)
    keep = {}
    d = lambda name, t: keep.setdefault(name, t.to(dev).contiguous())
    p = L.MiConvParams()
    p.B, p.H, p.W = B, H, W
    p.in0 = L.MiAct(d("x", x).data_ptr(), Cin, d("s", chan_stats(x)).data_ptr(), 1, 1.0, 0)
    p.Cout, p.ksize, p.stride, p.up2 = 8, 3, 1, 1
    wf, wexp = P.pack_conv_weight_rp(w)
    p.w_rp, p.w_rp_exp, p.bias = d("wf", wf).data_ptr(), wexp, d("b", bias).data_ptr()
    rows = lib.mi_conv_stripe_rows(C.byref(p))
    assert rows == W // 8
    nt = H // rows
    out, ost = run(lib, p, 12, nt, dev)
    scale = max(1.0, ref.abs().max().item() / 8.0)
    err = (out.double() - ref).abs().max().item()
    print(f"stripe up-sampling conv {case}: max|d| = {err:.2e} (gate {2e-5 * scale:.2e})")
    assert err < 2e-5 * scale
    check_stats(ost, ref.float())
    for nblk in (2, nt):
        if nt % nblk == 0 and nblk <= 15:
            o2, s2 = run(lib, p, 12 | (nblk << 12), nt, dev)
            assert torch.equal(o2, out) and torch.equal(s2, ost)
    o_rp, _ = run(lib, p, 6, tile_nt(lib, 6, H, W), dev)
    assert torch.equal(o_rp, out), f"stripe and tile kernels differ: max|d| = {(o_rp - out).abs().max().item():.3e}"


DOWN_CASES = [
    # B, Cout, H, W (OUTPUT size), xscale, wscale: Conv 4x4 stride 2 over 8 channels (Downsample, layers.py:308-319), no GroupNorm
    (2, 8, 32, 128, 1.0, 1.0),          # SR U-Net: 8 -> 8, 256^2 -> 128^2
    (8, 16, 16, 64, 1.0, 1.0),          # SR U-Net: 8 -> 16, 128^2 -> 64^2 (B % 8 == 0)
    (1, 8, 24, 64, 300.0, 1.0 / 64),    # scaled operands, H not a power of two
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", DOWN_CASES)
def test_conv_full_width_stripes_downsampling(backend, case):
    """tile_cfg 12 with the 4x4 stride-2 conv: the ring holds source rows of twice the width as two column-parity planes, one output row per step;
    against torch fp64 and bit for bit against conv_rp's MODE 2"""
    dev = setup(backend)
    lib = L.lib()
    B, Cout, H, W, xs, wsc = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    rn = lambda *s_: torch.randn(*s_, generator=g)
    x = (rn(B, 8, 2 * H, 2 * W) * 1.5 + 0.3) * xs
    w, bias = rn(Cout, 8, 4, 4) * 0.2 * wsc, rn(Cout) * wsc * xs
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride=2, padding=1)
    keep = {}
    d = lambda name, t: keep.setdefault(name, t.to(dev).contiguous())
    p = L.MiConvParams()
    p.B, p.H, p.W = B, H, W
    p.in0 = L.MiAct(d("x", x).data_ptr(), 8, d("s", chan_stats(x)).data_ptr(), 1, 1.0, 0)
    p.Cout, p.ksize, p.stride, p.up2 = Cout, 4, 2, 0
    wf, wexp = P.pack_conv_weight_rp(w)
    p.w_rp, p.w_rp_exp, p.bias = d("wf", wf).data_ptr(), wexp, d("b", bias).data_ptr()
    rows = lib.mi_conv_stripe_rows(C.byref(p))
    assert rows == W // 8
    nt = H // rows
    out, ost = run(lib, p, 12, nt, dev)
    scale = max(1.0, ref.abs().max().item() / 8.0)
    err = (out.double() - ref).abs().max().item()
    print(f"stripe stride-2 conv {case}: max|d| = {err:.2e} (gate {2e-5 * scale:.2e})")
    assert err < 2e-5 * scale
    check_stats(ost, ref.float())
    for nblk in (2, nt):
        if nt % nblk == 0 and nblk <= 15:
            o2, s2 = run(lib, p, 12 | (nblk << 12), nt, dev)
            assert torch.equal(o2, out) and torch.equal(s2, ost)
    o_rp, _ = run(lib, p, 7, tile_nt(lib, 7, H, W), dev)            # (the tile kernel's stride-2 member has 8 x 32 tiles)
    assert torch.equal(o_rp, out), f"stripe and tile kernels differ: max|d| = {(o_rp - out).abs().max().item():.3e}"


def test_stripe_eligibility_is_declared_by_the_library():
    """mi_conv_stripe_rows: what tile_cfg 12 takes (the engine asks before planning a launch) -- no compute, runs without a GPU"""
    setup("emu")
    lib = L.lib()
    p = L.MiConvParams()
    p.B, p.H, p.W, p.Cout, p.ksize, p.stride = 2, 64, 64, 8, 3, 1
    p.in0 = L.MiAct(1, 8, 0, 0, 1.0, 0)
    p.w_rp = 1
    assert lib.mi_conv_stripe_rows(C.byref(p)) == 8
    p.W = 32
    assert lib.mi_conv_stripe_rows(C.byref(p)) == 4
    for field, val in (("W", 48), ("H", 60), ("stride", 2), ("gn_groups", 3), ("Cout", 32), ("out_st", 1), ("tile_cfg", 0x400)):
        q = L.MiConvParams.from_buffer_copy(p)
        q.W = 64
        setattr(q, field, val)
        if field == "stride":
            q.ksize, q.W = 4, 32                                     # (4x4 stride 2 is taken at 64 / 128 output columns)
        assert lib.mi_conv_stripe_rows(C.byref(q)) == 0, field
    q = L.MiConvParams.from_buffer_copy(p)
    q.W, q.ksize, q.stride = 64, 4, 2
    assert lib.mi_conv_stripe_rows(C.byref(q)) == 8
    q.Cout = 16
    assert lib.mi_conv_stripe_rows(C.byref(q)) == 8
    q.gn_groups = 8
    assert lib.mi_conv_stripe_rows(C.byref(q)) == 0
    q = L.MiConvParams.from_buffer_copy(p)
    q.W, q.in0 = 256, L.MiAct(1, 16, 0, 0, 1.0, 0)                    # 16 input channels at 256 wide: not instantiated (LDS)
    assert lib.mi_conv_stripe_rows(C.byref(q)) == 0
