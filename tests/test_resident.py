"""Resident conv chain (csrc/conv_resident.hip, mi_resident_convs_fwd): runs of Block / ResnetBlock convolutions of one U-Net level in
ONE launch (layers.py:131-145, 417-439), against torch fp64 -- same gates as the per-layer row-paired kernel (tests/test_kernels.py)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from minimagen_amd import _lib as L
from minimagen_amd import packing as P
from tests._backend import BACKENDS, setup

SK = 2 ** -0.5


def chan_stats(x, nt=1):
    """per-channel partial (sum, sum of squares) [B][C][nt][2]: everything in tile 0 (a producer with one tile)"""
    B, Cc = x.shape[:2]
    st = torch.zeros(B, Cc, nt, 2)
    st[:, :, 0, 0] = x.double().sum((2, 3)).float()
    st[:, :, 0, 1] = (x.double() ** 2).sum((2, 3)).float()
    return st


class Chain:
    """Builds mi_resident_params layer by layer together with the torch fp64 reference of the same chain."""

    def __init__(self, dev, B, H, W, seed, half=False):
        self.dev, self.B, self.H, self.W = dev, B, H, W
        self.g = torch.Generator().manual_seed(seed)
        self.keep = []
        self.p = L.MiResidentParams()
        self.p.B, self.p.H, self.p.W = B, H, W
        self.p.half = 1 if half else 0
        self.S = L.lib().mi_resident_slabs(H, W)
        assert self.S > 0
        self.sync = torch.zeros(L.lib().mi_resident_sync_bytes(B, H, W), dtype=torch.uint8, device=dev)
        self.p.sync = self.sync.data_ptr()
        self.ss = self.rn(B, 400) * 0.3
        self.p.scale_shift, self.p.ss_stride = self.d(self.ss).data_ptr(), 400
        self.ss_next = 3
        self.cur = None          # reference value of the resident tensor (double)
        self.X = None
        self.n = 0
        self.outs = []           # (device tensor, device stats, reference)

    def rn(self, *s):
        return torch.randn(*s, generator=self.g)

    def d(self, t):
        t = t.to(self.dev).contiguous()
        self.keep.append(t)
        return t

    def act(self, x, scale=1.0):
        return L.MiAct(self.d(x).data_ptr(), x.shape[1], self.d(chan_stats(x)).data_ptr(), 1, scale, 0, 0)

    def layer(self, Cout, src=None, skip=None, gn=True, ss=False, res='none', res_glob=None, res_skip=None, save_x=False, store=False, wscale=1.0):
        """src: a global input tensor (None: the resident tensor); skip: second conv input (x 2^-1/2);
        res: none | idX | idG (res_glob) | convX (over X ++ res_skip) | convG (over res_glob ++ res_skip)"""
        Lr = self.p.layer[self.n]
        parts = []
        if src is None:
            assert self.cur is not None
            Lr.src = 0
            parts.append(self.cur)
        else:
            Lr.src = 1
            Lr.in0 = self.act(src)
            parts.append(src.double())
        if skip is not None:
            Lr.in1 = self.act(skip, SK)
            parts.append(skip.double() * SK)
        h = torch.cat(parts, 1)
        Cin = h.shape[1]
        w, bias = self.rn(Cout, Cin, 3, 3) * 0.2 * wscale, self.rn(Cout) * wscale
        if gn:
            gamma, beta = 1 + 0.2 * self.rn(Cin), 0.1 * self.rn(Cin)
            Lr.gn_groups, Lr.gn_gamma, Lr.gn_beta, Lr.gn_eps = 8, self.d(gamma).data_ptr(), self.d(beta).data_ptr(), 1e-5
            h = F.group_norm(h, 8, gamma.double(), beta.double(), 1e-5)
            if ss:
                off = self.ss_next
                self.ss_next += 2 * Cin
                Lr.ss_off = off
                sst = self.ss.double()
                h = h * (sst[:, off:off + Cin, None, None] + 1) + sst[:, off + Cin:off + 2 * Cin, None, None]
            else:
                Lr.ss_off = -1
            h = F.silu(h)
        else:
            Lr.ss_off = -1
        y = F.conv2d(h, w.double(), bias.double(), padding=1)
        wf, wexp = P.pack_conv_weight_rp(w)
        Lr.w_rp, Lr.w_rp_exp, Lr.bias, Lr.Cout = self.d(wf).data_ptr(), wexp, self.d(bias).data_ptr(), Cout
        if res == 'idX':
            Lr.res = 1
            y = y + self.X
        elif res == 'idG':
            Lr.res = 2
            Lr.res0 = self.act(res_glob)
            y = y + res_glob.double()
        elif res in ('convX', 'convG'):
            rparts = [self.X] if res == 'convX' else [res_glob.double()]
            Lr.res = 3 if res == 'convX' else 4
            if res == 'convG':
                Lr.res0 = self.act(res_glob)
            if res_skip is not None:
                Lr.res1 = self.act(res_skip, SK)
                rparts.append(res_skip.double() * SK)
            rin = torch.cat(rparts, 1)
            rw, rb = self.rn(Cout, rin.shape[1], 1, 1) * 0.3, self.rn(Cout)
            y = y + F.conv2d(rin, rw.double(), rb.double())
            rwf, rwexp = P.pack_conv_weight_rp(rw)
            Lr.res_w_rp, Lr.res_w_rp_exp, Lr.res_b = self.d(rwf).data_ptr(), rwexp, self.d(rb).data_ptr()
        Lr.save_x = 1 if save_x else 0
        if save_x:
            self.X = y
        if store:
            out = self.d(torch.full((self.B, Cout, self.H, self.W), float('nan')))
            ost = self.d(torch.zeros(self.B, Cout, self.S, 2))
            Lr.out, Lr.out_stats, Lr.out_nt = out.data_ptr(), ost.data_ptr(), self.S
            self.outs.append((out, ost, y))
        self.cur = y
        self.n += 1
        self.p.n_layers = self.n

    def run(self, reps=1):
        for _ in range(reps):
            L.check(L.lib().mi_resident_convs_fwd(C.byref(self.p), L.current_stream()), "resident")
        if self.dev.type == "cuda":
            torch.cuda.synchronize()
        err = int(self.sync[L.lib().mi_resident_error_offset():L.lib().mi_resident_error_offset() + 4].cpu().view(torch.int32).item())
        assert err == 0, f"resident kernel reported {err:#x}"

    def check(self, gate=2e-5):
        for k, (out, ost, ref) in enumerate(self.outs):
            scale = max(1.0, ref.abs().max().item() / 8.0)
            e = (out.cpu().double() - ref).abs().max().item()
            print(f"stored output {k}: max|d| = {e:.2e} (gate {gate * scale:.2e}, |ref|max {ref.abs().max().item():.3g})")
            assert e < gate * scale
            s = ost.cpu().double().sum(2)
            rs, rq = ref.sum((2, 3)), (ref ** 2).sum((2, 3))
            assert (s[..., 0] - rs).abs().max().item() < 1e-4 * max(1.0, rs.abs().max().item())
            assert ((s[..., 1] - rq).abs() / rq.clamp_min(1.0)).max().item() < 1e-4


def sr_level_chain(ch, B, H, W, variant):
    """the three segments of the 64^2 level of the super-resolution U-Net (Unet.py:419-465 around the two cross-attention launches)"""
    rn = ch.rn
    if variant == "down":                       # ResnetBlocks of the down path: block1 / block2 + identity residual; skips are stored
        x = rn(B, 16, H, W) * 1.5 + 0.3
        ch.layer(16, src=x, gn=True)
        ch.layer(16, gn=True, ss=True, res='idG', res_glob=x, save_x=True)
        ch.layer(16, gn=True)
        ch.layer(16, gn=True, ss=True, res='idX', save_x=True, store=True)
        ch.layer(16, gn=True)
        ch.layer(16, gn=True, ss=True, res='idX', save_x=True, store=True)
        ch.layer(16, gn=True, store=True)       # block1 of mid_block1: feeds the cross-attention launch
    elif variant == "mid":                      # block2 of mid_block1 (+ identity residual from memory), block1 of mid_block2
        a, x = rn(B, 16, H, W), rn(B, 16, H, W) * 2
        ch.layer(16, src=a, gn=True, ss=True, res='idG', res_glob=x, store=True)
        ch.layer(16, gn=True, store=True)
    else:                                       # block2 of mid_block2, then the up path: concat inputs, 1x1 residual convs over the concat
        a, x, s1, s0 = rn(B, 16, H, W), rn(B, 16, H, W) * 2, rn(B, 16, H, W) * 1.3, rn(B, 16, H, W) + 0.5
        ch.layer(16, src=a, gn=True, ss=True, res='idG', res_glob=x, save_x=True)
        ch.layer(16, skip=s1, gn=True)
        ch.layer(16, gn=True, ss=True, res='convX', res_skip=s1, save_x=True)
        ch.layer(16, skip=s1, gn=True)
        ch.layer(16, gn=True, ss=True, res='convX', res_skip=s1, save_x=True, store=True)
        ch.layer(16, skip=s0, gn=True)
        ch.layer(16, gn=True, ss=True, res='convX', res_skip=s0, store=True)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [("down", 2, 32, 64), ("mid", 3, 64, 64), ("up", 2, 64, 64), ("up", 1, 16, 64)])
def test_resident_chain_sr_level(backend, case):
    dev = setup(backend)
    variant, B, H, W = case
    ch = Chain(dev, B, H, W, seed=hash(case) & 0xffff)
    sr_level_chain(ch, B, H, W, variant)
    ch.run()
    ch.check()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(2, 32, 32), (3, 16, 32), (1, 64, 32)])
def test_resident_chain_base_level(backend, case):
    """32-wide slabs and the channel counts of the base U-Net's 32^2 level: 8-channel blocks, the folded Parallel(3x3, 1x1) conv to 16
    channels without GroupNorm (Unet.py:233-234), 16 + 8 skip channels into 16 with the 1x1 residual conv over the 24"""
    dev = setup(backend)
    B, H, W = case
    ch = Chain(dev, B, H, W, seed=hash(case) & 0xffff)
    rn = ch.rn
    x, s8 = rn(B, 8, H, W) * 1.5 + 0.3, rn(B, 8, H, W)
    ch.layer(8, src=x, gn=True)
    ch.layer(8, gn=True, ss=True, res='idG', res_glob=x, save_x=True, store=True)
    ch.layer(16, gn=False, store=True)                       # Parallel fold: plain conv, operands scaled from the exchanged sum of squares
    ch.layer(16, gn=True, save_x=False)
    ch.layer(16, skip=s8, gn=True, save_x=True)
    ch.layer(16, skip=s8, gn=True)
    ch.layer(16, gn=True, ss=True, res='convX', res_skip=s8, store=True)
    ch.run()
    ch.check()


@pytest.mark.parametrize("backend", BACKENDS)
def test_resident_chain_replays_and_scaled_operands(backend):
    """the ticket / flag words only ever grow: the same parameters run again and again (as from a captured graph) must reproduce
    the first launch bit for bit; operands far from unit scale stay inside the fp16 split's range"""
    dev = setup(backend)
    B, H, W = 2, 32, 64
    ch = Chain(dev, B, H, W, seed=77)
    x = (ch.rn(B, 16, H, W) * 1.5 + 0.3) * 256.0
    s1 = ch.rn(B, 16, H, W) / 256.0
    ch.layer(16, src=x, gn=False, save_x=True, wscale=1 / 300.0)
    ch.layer(16, skip=s1, gn=True, ss=True, wscale=200.0)
    ch.layer(16, gn=True, res='convX', res_skip=s1, save_x=True, store=True, wscale=200.0)
    ch.layer(16, gn=True, res='idX', store=True)
    ch.run()
    first = [o.clone() for o, _, _ in ch.outs]
    ch.check()
    ch.run(reps=3)
    for a, (o, _, _) in zip(first, ch.outs):
        assert torch.equal(a, o)


@pytest.mark.parametrize("backend", BACKENDS)
def test_resident_rejects_bad_arguments(backend):
    dev = setup(backend)
    ch = Chain(dev, 1, 16, 64, seed=1)
    ch.layer(16, src=ch.rn(1, 16, 16, 64), gn=True)
    p = L.MiResidentParams.from_buffer_copy(ch.p)
    p.W = 48
    assert L.lib().mi_resident_convs_fwd(C.byref(p), L.current_stream()) != 0
    p = L.MiResidentParams.from_buffer_copy(ch.p)
    p.layer[0].src = 0
    assert L.lib().mi_resident_convs_fwd(C.byref(p), L.current_stream()) != 0
    p = L.MiResidentParams.from_buffer_copy(ch.p)
    p.layer[0].res = 1
    assert L.lib().mi_resident_convs_fwd(C.byref(p), L.current_stream()) != 0
    assert L.lib().mi_resident_slabs(48, 48) == 0 and L.lib().mi_resident_slabs(64, 64) in (4, 8) and L.lib().mi_resident_slabs(32, 32) == 4


# ---------------------------------------------------------------------------------------------------------------------------------------
# the engine's resident-chain pass (engine._fuse_resident, MINIMAGEN_RESIDENT=1): whole U-Net evaluations against the reference goldens
@pytest.fixture
def resident_on(monkeypatch):
    from minimagen_amd import engine
    monkeypatch.setattr(engine, "RESIDENT", 1)
    return engine


def _golden_unet(which, dev):
    from tests.test_unet import make_unet
    return make_unet(which, dev)


@pytest.mark.parametrize("backend", BACKENDS)
def test_engine_base_unet_with_resident_chains_matches_reference_golden(backend, resident_on):
    """unet_0 @64^2 with classifier-free guidance: 26 conv launches become 5 resident chains + 5 single convs; same gate as the
    launch-per-layer plan (tests/test_unet.py::test_forward_A_golden)"""
    from tests import _inputs as I
    dev = setup(backend)
    u0 = _golden_unet("unet0", dev)
    g = I.load("fwdA.pt"); m = g["meta"]
    emb, mask = I.text(m)
    x = I.seeded((2, 3, 64, 64), m["x_seed"]).to(dev)
    tm = torch.tensor(m["time"]).to(dev)
    og = u0.forward_with_cond_scale(x, tm, text_embeds=emb.to(dev), text_mask=mask.to(dev), cond_scale=3.)
    eng = u0.engine()
    ws = eng.workspace(2, 4, 64, 64)
    names = [n for _, _, n in ws.prog]
    assert names.count("resident") >= 4 and names.count("conv") <= 6, names
    eng.check_resident(ws)
    ref = g["out_null"] + (g["out_cond"] - g["out_null"]) * 3.
    d = (og.cpu() - ref).abs().max().item()
    print(f"unet_0 forward with {names.count('resident')} resident chains: max|d| = {d:.2e}")
    assert d < 3 * 2e-5
    og2 = u0.forward_with_cond_scale(x, tm, text_embeds=emb.to(dev), text_mask=mask.to(dev), cond_scale=3.)
    assert torch.equal(og, og2)                  # run to run bit-identical (fixed reduction orders, whatever the workgroup placement)


@pytest.mark.parametrize("backend", [pytest.param("gpu", marks=pytest.mark.gpu)])
def test_engine_sr_unet_256_with_resident_chains_matches_reference_golden(backend, resident_on):
    """unet_1 @256^2 (low-res conditioning): the 17 conv launches of its 64^2 level become 3 resident chains around the two
    cross-attention launches; fp32 and the reduced-precision configuration (bf16 storage of the tensors that still go to memory)"""
    from tests import _inputs as I
    dev = setup(backend)
    u1 = _golden_unet("unet1", dev)
    g = I.load("fwdC.pt"); m = g["meta"]
    emb, mask = I.text(m)
    x = I.seeded((1, 3, 256, 256), m["x_seed"]).to(dev)
    lr = I.seeded((1, 3, 256, 256), m["lr_seed"]).to(dev)
    tm = torch.tensor(m["time"]).to(dev)
    kw = dict(lowres_cond_img=lr, lowres_noise_times=torch.tensor(m["ltime"]).to(dev), text_embeds=emb[:1].to(dev), text_mask=mask[:1].to(dev))
    o = u1(x, tm, **kw)
    eng = u1.engine()
    ws = eng.workspace(1, 1, 256, 256)
    names = [n for _, _, n in ws.prog]
    assert names.count("resident") == 3, names
    eng.check_resident(ws)
    ref = g["out_cond"]
    d = (o.cpu() - ref).abs().max().item()
    print(f"unet_1 @256 forward with resident chains: max|d| = {d:.2e}")
    assert d < 2e-5
    eng.precision = "half"
    oh = u1(x, tm, **kw)
    eng.precision = "fp32"
    wsh = eng.workspace(1, 1, 256, 256, precision="half")
    assert wsh.store16 and [n for _, _, n in wsh.prog].count("resident") == 3
    eng.check_resident(wsh)
    dh = (oh.cpu() - ref).abs()
    scale = ref.abs().max()
    print(f"  reduced precision: max|d| = {dh.max():.2e}, mean {dh.mean():.2e} (|ref|max {scale:.3g})")
    assert dh.max() < 3e-2 * scale and dh.mean() < 3e-3 * scale


@pytest.mark.parametrize("backend", BACKENDS)
def test_sampling_loop_with_resident_chains(backend, resident_on):
    """the base stage's denoising loop through captured graphs with resident chains inside: the ticket / tag words advance across
    replays; the image matches the oracle's sampling loop and a second call reproduces it bit for bit"""
    from oracle import restated as R
    from tests import _inputs as I
    from tests.test_sampler import make_imagen
    dev = setup(backend)
    T = 25
    im = make_imagen([64], T, dev)
    emb, mask = R.synthetic_text(2, length=32, seed=5)
    a = im.sample(text_embeds=emb.to(dev), text_masks=mask.to(dev), cond_scale=3., _noise=R.make_randn(11))
    b = im.sample(text_embeds=emb.to(dev), text_masks=mask.to(dev), cond_scale=3., _noise=R.make_randn(11))
    eng = im.unets[0].engine()
    for ws in eng._ws.values():
        assert any(n == "resident" for _, _, n in ws.prog)
        eng.check_resident(ws)
    assert torch.equal(a, b)
    ref = R.sample([I.load("unet0_sd.pt")], [64], T, text_embeds=emb, text_masks=mask, cond_scale=3., randn=R.make_randn(11))
    d = (a.cpu() - ref).abs()
    print(f"base stage T={T} with resident chains vs oracle: max|d| = {d.max():.2e}, mean {d.mean():.2e}")
    assert d.max() < 1e-4 and d.mean() < 1e-5


@pytest.mark.parametrize("backend", [pytest.param("gpu", marks=pytest.mark.gpu)])
def test_resident_chains_of_two_streams_side_by_side(backend):
    """Two chains (64 images each: 2 x 256 workgroups of 512 work-items, more than the GPU holds at once) launched back to back on two
    streams, 40 rounds: the slabs of an image are claimed by ticket after a workgroup is resident, so the launches interleave on the CUs
    without deadlock; every launch reproduces the chain's stand-alone result bit for bit and no workgroup reports a timeout."""
    dev = setup(backend)
    chains, streams, refs = [], [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)], []
    for k, variant in enumerate(("down", "up")):
        ch = Chain(dev, 64, 64, 64, seed=40 + k)
        sr_level_chain(ch, 64, 64, 64, variant)
        ch.run()
        ch.check()
        refs.append([o.clone() for o, _, _ in ch.outs])
        chains.append(ch)
    torch.cuda.synchronize()
    lib = L.lib()
    for rnd in range(40):
        for ch, st in zip(chains, streams):
            with torch.cuda.stream(st):
                L.check(lib.mi_resident_convs_fwd(C.byref(ch.p), st.cuda_stream), "resident")
        if rnd % 8 == 7:
            torch.cuda.synchronize()
            for ch, ref in zip(chains, refs):
                off = lib.mi_resident_error_offset()
                assert int(ch.sync[off:off + 4].cpu().view(torch.int32).item()) == 0
                assert all(torch.equal(a, o) for a, (o, _, _) in zip(ref, ch.outs))
