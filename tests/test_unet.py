"""Unet.forward / forward_with_cond_scale through the HIP engine vs the reference golden vectors and the oracle.
Tolerance (fp32): forward atol 2e-5."""
import pytest
import torch

from minimagen_amd.Unet import Unet, Base, Super, BaseTest, SuperTest
from oracle import restated as R
from tests import _inputs as I
from tests._backend import BACKENDS, GPU_ONLY, setup

FWD_ATOL = 2e-5


def make_unet(which, dev):
    p = I.unet_params()
    u = Unet(**p[which])
    u.load_state_dict(I.load(f"{which}_sd.pt"), strict=True)
    return u.to(dev).eval()          # inference: the HIP engine (train mode + autograd would take the differentiable torch-op path)


@pytest.mark.parametrize("backend", BACKENDS)
def test_forward_A_golden(backend):
    dev = setup(backend)
    u0 = make_unet("unet0", dev)
    g = I.load("fwdA.pt"); m = g["meta"]
    emb, mask = I.text(m)
    x = I.seeded((2, 3, 64, 64), m["x_seed"]).to(dev)
    tm = torch.tensor(m["time"]).to(dev)
    oc = u0(x, tm, text_embeds=emb.to(dev), text_mask=mask.to(dev), cond_drop_prob=0.)
    on = u0(x, tm, text_embeds=emb.to(dev), text_mask=mask.to(dev), cond_drop_prob=1.)
    og = u0.forward_with_cond_scale(x, tm, text_embeds=emb.to(dev), text_mask=mask.to(dev), cond_scale=3.)
    assert (oc.cpu() - g["out_cond"]).abs().max() < FWD_ATOL
    assert (on.cpu() - g["out_null"]).abs().max() < FWD_ATOL
    ref = g["out_null"] + (g["out_cond"] - g["out_null"]) * 3.
    assert (og.cpu() - ref).abs().max() < 3 * FWD_ATOL        # guidance amplifies the difference of the halves 3x


@pytest.mark.parametrize("backend", BACKENDS)
def test_forward_A_half_precision(backend):
    """the reduced-precision configuration (single fp16 term on the matrix cores, fp32 accumulate / softmax / statistics):
    within the half-precision gate of SURVEY.md 8(c), and measurably different from the fp32-grade path (i.e. really active)"""
    dev = setup(backend)
    u0 = make_unet("unet0", dev)
    g = I.load("fwdA.pt"); m = g["meta"]
    emb, mask = I.text(m)
    x = I.seeded((2, 3, 64, 64), m["x_seed"]).to(dev)
    tm = torch.tensor(m["time"]).to(dev)
    u0.engine().precision = "half"
    oc = u0(x, tm, text_embeds=emb.to(dev), text_mask=mask.to(dev), cond_drop_prob=0.)
    u0.engine().precision = "fp32"
    of = u0(x, tm, text_embeds=emb.to(dev), text_mask=mask.to(dev), cond_drop_prob=0.)
    d = (oc.cpu() - g["out_cond"]).abs()
    scale = g["out_cond"].abs().max()
    assert d.max() < 3e-2 * scale and d.mean() < 3e-3 * scale, (d.max(), d.mean(), scale)
    assert d.max() > 3 * FWD_ATOL                                    # not the fp32-grade path
    assert (of.cpu() - g["out_cond"]).abs().max() < FWD_ATOL         # switching back restores it


@pytest.mark.parametrize("backend", BACKENDS)
def test_forward_B_golden_lowres(backend):
    dev = setup(backend)
    u1 = make_unet("unet1", dev)
    g = I.load("fwdB.pt"); m = g["meta"]
    emb, mask = I.text(m)
    x = I.seeded((2, 3, 128, 128), m["x_seed"]).to(dev)
    lr = I.seeded((2, 3, 128, 128), m["lr_seed"]).to(dev)
    kw = dict(lowres_cond_img=lr, lowres_noise_times=torch.tensor(m["ltime"]).to(dev), text_embeds=emb.to(dev), text_mask=mask.to(dev))
    og = u1.forward_with_cond_scale(x, torch.tensor(m["time"]).to(dev), cond_scale=3., **kw)
    ref = g["out_null"] + (g["out_cond"] - g["out_null"]) * 3.
    assert (og.cpu() - ref).abs().max() < 3 * FWD_ATOL
    with pytest.raises(AssertionError):
        u1(x, torch.tensor(m["time"]).to(dev), text_embeds=emb.to(dev))      # Unet.py:384-387


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_forward_C_golden_256(backend):
    dev = setup(backend)
    u1 = make_unet("unet1", dev)
    g = I.load("fwdC.pt"); m = g["meta"]
    emb, mask = I.text(m)
    x = I.seeded((1, 3, 256, 256), m["x_seed"]).to(dev)
    lr = I.seeded((1, 3, 256, 256), m["lr_seed"]).to(dev)
    o = u1(x, torch.tensor(m["time"]).to(dev), lowres_cond_img=lr, lowres_noise_times=torch.tensor(m["ltime"]).to(dev),
           text_embeds=emb[:1].to(dev), text_mask=mask[:1].to(dev))
    assert (o.cpu() - g["out_cond"]).abs().max() < FWD_ATOL


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_forward_full_size_vs_oracle(backend):
    """BASELINE config sizes: base 64^2 at B=32 and SR 256^2 at B=4 against the oracle run on the host."""
    dev = setup(backend)
    u0, u1 = make_unet("unet0", dev), make_unet("unet1", dev)
    sd0, sd1 = I.load("unet0_sd.pt"), I.load("unet1_sd.pt")
    emb, mask = R.synthetic_text(32, length=64, seed=7)
    x = I.seeded((32, 3, 64, 64), 101)
    tm = torch.randint(0, 100, (32,), generator=torch.Generator().manual_seed(4))
    o = u0.forward_with_cond_scale(x.to(dev), tm.to(dev), text_embeds=emb.to(dev), text_mask=mask.to(dev), cond_scale=3.)
    ref = R.unet_forward_with_cond_scale(sd0, x, tm, cond_scale=3., text_embeds=emb, text_mask=mask)
    assert (o.cpu() - ref).abs().max() < 3 * FWD_ATOL
    xb, lr = I.seeded((4, 3, 256, 256), 102), I.seeded((4, 3, 256, 256), 103)
    tb, lt = torch.tensor([99, 50, 1, 0]), torch.full((4,), 20)
    o = u1(xb.to(dev), tb.to(dev), lowres_cond_img=lr.to(dev), lowres_noise_times=lt.to(dev), text_embeds=emb[:4].to(dev), text_mask=mask[:4].to(dev))
    ref = R.unet_forward(sd1, xb, tb, lowres_cond_img=lr, lowres_noise_times=lt, text_embeds=emb[:4], text_mask=mask[:4])
    assert (o.cpu() - ref).abs().max() < FWD_ATOL


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_forward_1024_vs_oracle(backend):
    """BASELINE config 5's third stage: the SR U-Net (unet_1 parameters) at 1024 x 1024, B = 1, against the oracle run on the host"""
    dev = setup(backend)
    u1, sd1 = make_unet("unet1", dev), I.load("unet1_sd.pt")
    emb, mask = R.synthetic_text(1, length=40, seed=11)
    x, lr = I.seeded((1, 3, 1024, 1024), 201), I.seeded((1, 3, 1024, 1024), 202)
    tm, lt = torch.tensor([37]), torch.tensor([20])
    o = u1(x.to(dev), tm.to(dev), lowres_cond_img=lr.to(dev), lowres_noise_times=lt.to(dev), text_embeds=emb.to(dev), text_mask=mask.to(dev))
    ref = R.unet_forward(sd1, x, tm, lowres_cond_img=lr, lowres_noise_times=lt, text_embeds=emb, text_mask=mask)
    d = (o.cpu() - ref).abs().max().item()
    print(f"Unet.forward 1024^2 B=1 vs oracle: max|d| = {d:.2e}")
    assert o.shape == (1, 3, 1024, 1024) and d < FWD_ATOL


WIDE_SMALL = dict(dim=64, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True),
                  attend_at_middle=True, memory_efficient=False)


@pytest.mark.parametrize("backend", BACKENDS)
def test_wide_channel_unet_vs_oracle(backend):
    """the wide-channel regime end to end on a small net (64 / 128 channels: wide row-paired convs with grid-tiled output channels,
    unfolded flash cross-attention, multi-query self-attention, token-layout ChanFeedForward, attend_at_middle) vs the oracle"""
    dev = setup(backend)
    torch.manual_seed(5)
    u = Unet(**WIDE_SMALL)
    sd = {k: v.clone() for k, v in u.state_dict().items()}
    u = u.to(dev).eval()
    B, S = (2, 32) if backend == "gpu" else (1, 16)
    emb, mask = R.synthetic_text(B, length=12, seed=4)
    x, tm = I.seeded((B, 3, S, S), 31), torch.tensor([77, 5][:B])
    o = u(x.to(dev), tm.to(dev), text_embeds=emb.to(dev), text_mask=mask.to(dev))
    ref = R.unet_forward(sd, x, tm, text_embeds=emb, text_mask=mask)
    d = (o.cpu() - ref).abs().max().item()
    print(f"wide small U-Net {S}x{S} B={B}: max|d| = {d:.2e} (|ref| max {ref.abs().max():.2f})")
    assert d < FWD_ATOL * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("backend", BACKENDS)
def test_reversed_image_order_of_alternate_convs_is_bit_exact(backend, monkeypatch):
    """MINIMAGEN_CONV_REVERSE (default on, batches that are multiples of 8): a row-paired conv walks the image groups opposite to its
    producer -- a placement choice only: same bits as with the knob off, and the oracle's values"""
    from minimagen_amd import engine as E
    dev = setup(backend)
    torch.manual_seed(6)
    kw = dict(dim=8, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=False, layer_cross_attns=(False, True), memory_efficient=True)
    sd = {k: v.clone() for k, v in Unet(**kw).state_dict().items()}
    B, S = 8, (64 if backend == "gpu" else 16)
    emb, mask = R.synthetic_text(B, length=9, seed=4)
    x, tm = I.seeded((B, 3, S, S), 33), torch.arange(B) * 3 + 1
    outs = []
    for rev in (1, 0):
        monkeypatch.setattr(E, "CONV_REVERSE", rev)
        u = Unet(**kw)
        u.load_state_dict(sd)
        u = u.to(dev).eval()
        outs.append(u(x.to(dev), tm.to(dev), text_embeds=emb.to(dev), text_mask=mask.to(dev)).cpu())
        plan = next(iter(u.engine()._ws.values()))
        flagged = sum(1 for _, prm, name in plan.prog if hasattr(prm, "tile_cfg") and hasattr(prm, "w_rp") and (prm.tile_cfg & 0x200))
        assert (flagged > 0) == bool(rev), flagged
    assert torch.equal(outs[0], outs[1])
    ref = R.unet_forward(sd, x, tm, text_embeds=emb, text_mask=mask)
    assert (outs[0] - ref).abs().max() < FWD_ATOL * max(1.0, ref.abs().max().item())


def _fp64_oracle(sd, x, tm, **kw):
    """the oracle's algorithm evaluated in fp64 (oracle/restated.py COMPUTE_DTYPE): the exact value fp32 implementations scatter around"""
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    kw64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in kw.items()}
    R.COMPUTE_DTYPE = torch.float64
    try:
        return R.unet_forward_with_cond_scale(sd64, x.double(), tm, **kw64)
    finally:
        R.COMPUTE_DTYPE = torch.float32


HOSTILE = [("unet0", 3, "noise"), ("unet0", 4, "constant"), ("unet1", 5, "noise"), ("unet1", 6, "constant"), ("unet1", 7, "noise")]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("which,seed,image", HOSTILE)
def test_hostile_weight_ranges_vs_fp64_oracle(backend, which, seed, image):
    """The fp32 label of the matrix-core path rests on power-of-two operand scalings derived from BOUNDS (weights' maxima, GroupNorm /
    LayerNorm gains, per-channel statistics).  All other parity evidence uses random-init weights; here the BASELINE U-Nets carry
    trained-like / adversarial ranges (tests/_inputs.hostile_state_dict: per-layer scales 2^-10 .. 2^6, one 2^8 outlier weight per filter,
    outlier normalisation channels), text embeddings x 100, and either a noise image or an ALL-CONSTANT one (GroupNorm of a nearly
    constant tensor: sigma ~ 0 away from the borders).  Gate: the HIP result is as close to the fp64 value of the algorithm as the
    reference's own fp32 arithmetic is -- |hip - ref64| <= max(3 * 2e-5 * max|ref64|, 4 * |oracle32 - ref64|) -- with classifier-free
    guidance (x 3), at the BASELINE sizes on the GPU."""
    dev = setup(backend)
    p = I.unet_params()[which]
    torch.manual_seed(seed)
    sd = I.hostile_state_dict(I.load(f"{which}_sd.pt"), seed)
    u = Unet(**p)
    u.load_state_dict(sd, strict=True)
    u = u.to(dev).eval()
    S = ({"unet0": 64, "unet1": 256} if backend == "gpu" else {"unet0": 32, "unet1": 32})[which]
    B = 2
    emb, mask = R.synthetic_text(B, length=24, seed=seed)
    emb = emb * 100.0
    x = I.seeded((B, 3, S, S), 50 + seed) if image == "noise" else torch.full((B, 3, S, S), 0.37)
    tm = torch.tensor([77, 2])
    kw = dict(text_embeds=emb, text_mask=mask, cond_scale=3.)
    if p["lowres_cond"]:
        kw.update(lowres_cond_img=(I.seeded((B, 3, S, S), 60 + seed) if image == "noise" else torch.full((B, 3, S, S), -0.81)), lowres_noise_times=torch.tensor([20, 20]))
    o = u.forward_with_cond_scale(x.to(dev), tm.to(dev), **{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()}).cpu().double()
    ref64 = _fp64_oracle(sd, x, tm, **kw)
    o32 = R.unet_forward_with_cond_scale(sd, x, tm, **kw).double()
    e_hip, e_o32, mag = (o - ref64).abs().max().item(), (o32 - ref64).abs().max().item(), ref64.abs().max().item()
    print(f"hostile {which} seed {seed} {image} @{S}: |ref64|max {mag:.3g}; |hip - ref64| {e_hip:.3g}, |oracle32 - ref64| {e_o32:.3g}")
    assert torch.isfinite(ref64).all() and torch.isfinite(o).all()
    assert e_hip <= max(3 * FWD_ATOL * max(1.0, mag), 4 * e_o32), (e_hip, e_o32, mag)


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_default_unet_vs_oracle(backend):
    """``Unet()`` with the reference's default arguments (dim 128, dim_mults (1, 2, 4), self- and cross-attention at every level,
    Unet.py:31-48) at 64 x 64, B = 2, with classifier-free guidance, against the oracle run on the host; and the reference's Base /
    Super presets construct and run at a reduced width"""
    dev = setup(backend)
    torch.manual_seed(6)
    u = Unet()
    sd = {k: v.clone() for k, v in u.state_dict().items()}
    u = u.to(dev).eval()
    emb, mask = R.synthetic_text(2, length=20, seed=8)
    x, tm = I.seeded((2, 3, 64, 64), 41), torch.tensor([90, 11])
    o = u.forward_with_cond_scale(x.to(dev), tm.to(dev), text_embeds=emb.to(dev), text_mask=mask.to(dev), cond_scale=3.)
    ref = R.unet_forward_with_cond_scale(sd, x, tm, cond_scale=3., text_embeds=emb, text_mask=mask)
    d = (o.cpu() - ref).abs().max().item()
    print(f"Unet() default 64x64 B=2 cond_scale 3: max|d| = {d:.2e} (|ref| max {ref.abs().max():.2f})")
    assert d < 3 * FWD_ATOL * max(1.0, ref.abs().max().item())
    for klass, kw in ((Base, dict(dim=64)), (Super, dict(dim=32, lowres_cond=True))):
        torch.manual_seed(7)
        m = klass(**kw)
        sdm = {k: v.clone() for k, v in m.state_dict().items()}
        m = m.to(dev).eval()
        xm = I.seeded((1, 3, 64, 64), 43)
        extra = dict(lowres_cond_img=I.seeded((1, 3, 64, 64), 44), lowres_noise_times=torch.tensor([20])) if m.lowres_cond else {}
        om = m(xm.to(dev), torch.tensor([50]).to(dev), text_embeds=emb[:1].to(dev), text_mask=mask[:1].to(dev),
               **{k: v.to(dev) for k, v in extra.items()})
        refm = R.unet_forward(sdm, xm, torch.tensor([50]), text_embeds=emb[:1], text_mask=mask[:1], **extra)
        dm = (om.cpu() - refm).abs().max().item()
        print(f"{klass.__name__}({kw}) 64x64: max|d| = {dm:.2e} (|ref| max {refm.abs().max():.2f})")
        assert dm < 3 * FWD_ATOL * max(1.0, refm.abs().max().item())


@pytest.mark.parametrize("backend", BACKENDS)
def test_forward_without_text_vs_oracle(backend):
    """Unet.py:572: text conditioning is optional -- ``text_embeds=None`` runs with the context [null | time tokens] and no text hiddens"""
    dev = setup(backend)
    u0, sd0 = make_unet("unet0", dev), I.load("unet0_sd.pt")
    x, tm = I.seeded((2, 3, 32, 32), 17), torch.tensor([40, 3])
    o = u0(x.to(dev), tm.to(dev))
    ref = R.unet_forward(sd0, x, tm, text_embeds=None, text_mask=None)
    assert (o.cpu() - ref).abs().max() < FWD_ATOL
    emb, mask = R.synthetic_text(2, length=12, seed=5)                 # and the same module still serves text-conditioned calls
    o = u0(x.to(dev), tm.to(dev), text_embeds=emb.to(dev), text_mask=mask.to(dev))
    assert (o.cpu() - R.unet_forward(sd0, x, tm, text_embeds=emb, text_mask=mask)).abs().max() < FWD_ATOL


@pytest.mark.parametrize("backend", BACKENDS)
def test_repack_after_data_mutation(backend):
    """in-place updates through ``p.data`` (EMA, ``.data.copy_``) do not bump ``p._version``: the content fingerprint must catch them"""
    dev = setup(backend)
    u0, sd0 = make_unet("unet0", dev), I.load("unet0_sd.pt")
    m = I.load("fwdA.pt")["meta"]
    emb, mask = I.text(m)
    x, tm = I.seeded((2, 3, 32, 32), 1), torch.tensor([3, 4])
    a = u0(x.to(dev), tm.to(dev), text_embeds=emb.to(dev), text_mask=mask.to(dev))
    for prm in u0.parameters():
        prm.data.mul_(1.05)
    b = u0(x.to(dev), tm.to(dev), text_embeds=emb.to(dev), text_mask=mask.to(dev))
    sd = {k: (v * 1.05 if v.is_floating_point() and k in dict(u0.named_parameters()) else v) for k, v in sd0.items()}
    ref = R.unet_forward(sd, x, tm, text_embeds=emb, text_mask=mask)
    assert (b.cpu() - ref).abs().max() < FWD_ATOL and (a - b).abs().max() > 1e-3


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("which", ["unet0", "unet1"])
def test_conditioning_ops_vs_oracle(backend, which):
    """op-level: mi_cond_step_fwd (Unet._generate_t_tokens, Unet.py:508-536, + norm_cond of the time tokens + every time_mlp) and
    mi_text_cond_fwd (Unet._text_condition, Unet.py:571-634) against the oracle's functions -- ragged masks, a caption longer than one
    row's mask, the conditional and the null half of a guidance batch"""
    dev = setup(backend)
    u, sd = make_unet(which, dev), I.load(f"{which}_sd.pt")
    eng = u.engine()
    B = 3
    emb, mask = R.synthetic_text(B, length=20, seed=5)
    mask[1, 7:] = False
    ws = eng.workspace(B, 2 * B, 32, 32)
    keep = torch.cat((torch.ones(B, dtype=torch.bool), torch.zeros(B, dtype=torch.bool)))
    tm = torch.tensor([99, 13, 0])
    lt = torch.tensor([20, 20, 20])
    ws.times.copy_(tm)
    if u.lowres_cond:
        ws.lowres_times.copy_(lt)
    eng.set_text(ws, emb.to(dev), mask.to(dev), keep)
    import ctypes as C
    from minimagen_amd import _lib as L
    fn, cp, _ = ws.prog_cond[0]
    L.check(fn(C.byref(cp), L.current_stream()), "mi_cond_step_fwd")
    emb2, mask2, tm2 = emb.repeat(2, 1, 1), mask.repeat(2, 1), tm.repeat(2)
    t_ref, tok_ref = R.generate_t_tokens(sd, tm2, lt.repeat(2) if u.lowres_cond else None)
    t_ref, c_ref = R.text_condition(sd, emb2, mask2, keep, t_ref, tok_ref)
    ntot = ws.ntot
    assert (ws.t_out.cpu() - t_ref).abs().max() < 2e-5                                       # t = time cond + text hiddens
    assert (ws.c_time.cpu() - c_ref[:, :ntot]).abs().max() < 2e-5                            # norm_cond(time tokens)
    assert (ws.c_text.cpu() - c_ref[:, ntot:]).abs().max() < 2e-5                            # norm_cond(text / null tokens), 256 rows
    # every ResnetBlock's time_mlp (layers.py:395-399): Linear(SiLU(t)) stacked in the engine's order
    pk = eng.pack()
    ss_ref = torch.nn.functional.silu(t_ref) @ pk.tm_w.cpu().t() + pk.tm_b.cpu()
    assert (ws.ss.cpu() - ss_ref).abs().max() < 2e-5


@pytest.mark.parametrize("backend", BACKENDS)
def test_repack_after_weight_update(backend):
    """packed / folded weight copies must follow load_state_dict and in-place parameter updates"""
    dev = setup(backend)
    u0 = make_unet("unet0", dev)
    m = I.load("fwdA.pt")["meta"]
    emb, mask = I.text(m)
    x, tm = I.seeded((2, 3, 64, 64), 1).to(dev), torch.tensor([3, 4]).to(dev)
    a = u0(x, tm, text_embeds=emb.to(dev), text_mask=mask.to(dev))
    with torch.no_grad():
        u0.mid_block1.cross_attn.fn.to_q.weight.mul_(1.5)
        u0.final_conv.weight.add_(0.01)
    b = u0(x, tm, text_embeds=emb.to(dev), text_mask=mask.to(dev))
    sd = {k: v.cpu() for k, v in u0.state_dict().items()}
    ref = R.unet_forward(sd, x.cpu(), tm.cpu(), text_embeds=emb, text_mask=mask)
    assert (a - b).abs().max() > 1e-3 and (b.cpu() - ref).abs().max() < FWD_ATOL


def test_no_cpu_path():
    """the product has no CPU fallback: host tensors are rejected when the gfx950 library is the backend"""
    import os
    from minimagen_amd import _lib as L
    if not os.path.exists(L.DEFAULT_LIB):
        pytest.skip("libminimagen_hip.so not built")
    L.use_library(L.DEFAULT_LIB)
    u0 = make_unet("unet0", "cpu")
    with pytest.raises(L.MinImagenHipError):
        u0(torch.zeros(1, 3, 64, 64), torch.tensor([1]), text_embeds=torch.zeros(1, 4, 512))


def test_constructor_api_and_state_dict_contract():
    p = I.unet_params()
    for which, n in (("unet0", 104115), ("unet1", 156099)):
        u = Unet(**p[which])
        assert sum(t.numel() for t in u.parameters()) == n
        sd = I.load(f"{which}_sd.pt")
        assert list(u.state_dict().keys()) == list(sd.keys())                 # same keys, same order as the reference
        assert all(u.state_dict()[k].shape == v.shape for k, v in sd.items())
    u = Unet(**p["unet0"])
    assert u._cast_model_parameters(lowres_cond=False, text_embed_dim=512, channels=3, channels_out=3) is u
    u2 = u._cast_model_parameters(lowres_cond=True, text_embed_dim=512, channels=3, channels_out=3)
    assert u2 is not u and u2.lowres_cond and u2.time_cond_dim == 64 and u2.init_conv.convs[0].in_channels == 6
    assert BaseTest.defaults["dim"] == 8 and SuperTest.defaults["memory_efficient"] and Base.defaults["dim"] == 512 and Super.defaults["dim"] == 128
    ua = Unet(dim=16, dim_mults=(1, 2), layer_attns=(False, True), layer_cross_attns=(False, True), attend_at_middle=True)
    keys = set(ua.state_dict().keys())
    for k in ("downs.1.3.attn.fn.to_kv.weight", "downs.1.3.ff.0.g", "downs.1.3.ff.4.weight", "mid_attn.fn.fn.null_kv",
              "downs.1.1.cross_attn.fn.to_out.1.beta", "ups.0.2.attn.fn.norm.gamma", "downs.1.4.fns.1.weight", "ups.0.3.1.weight"):
        assert k in keys, k


@pytest.mark.parametrize("backend", BACKENDS)
def test_attention_bearing_unet_vs_oracle(backend):
    """K10 + cross-attention at every level: layer_attns / layer_cross_attns / attend_at_middle on a narrow U-Net
    (self-attention TransformerBlocks, ChanFeedForward, mid attention) against the oracle."""
    dev = setup(backend)
    torch.manual_seed(3)
    u = Unet(dim=16, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(True, True),
             attend_at_middle=True, memory_efficient=False)
    with torch.no_grad():
        for n, p in u.named_parameters():
            if n.endswith(("gamma", ".g")) or "norm" in n and n.endswith("weight"):
                p.copy_(1 + 0.2 * torch.randn(p.shape))
    sd = {k: v.clone() for k, v in u.state_dict().items()}
    u = u.to(dev).eval()
    emb, mask = R.synthetic_text(2, length=12, seed=3)
    x = I.seeded((2, 3, 32, 32), 6)
    tm = torch.tensor([3, 9])
    ref = R.unet_forward_with_cond_scale(sd, x, tm, cond_scale=2., text_embeds=emb, text_mask=mask)
    out = u.forward_with_cond_scale(x.to(dev), tm.to(dev), text_embeds=emb.to(dev), text_mask=mask.to(dev), cond_scale=2.)
    assert (out.cpu() - ref).abs().max() < 4e-5


_NARROW = dict(dim=8, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=False, layer_cross_attns=(False, True), memory_efficient=False)
SWEEP = {
    # non-memory-efficient levels run at dim_in: cross-attention at C = 8 AND C = 16 in one U-Net (they share a fragment size)
    "ragged_48": (_NARROW, 48, {}),
    "ragged_36": (_NARROW, 36, {}),
    "three_levels": (dict(dim=8, dim_mults=(1, 2, 4), num_resnet_blocks=(1, 2, 1), layer_attns=False, layer_cross_attns=(False, True, True), memory_efficient=True), 64, {}),
    "t5_base_width": ({**_NARROW, "text_embed_dim": 768}, 32, {"E": 768}),
    "cond_dim_16": ({**_NARROW, "cond_dim": 16, "num_resnet_blocks": 2}, 32, {}),
    "one_channel": ({**_NARROW, "channels": 1}, 32, {"ch": 1}),
    "four_heads": ({**_NARROW, "attn_heads": 4}, 32, {}),
    "no_text_mask": (_NARROW, 32, {"nomask": True}),            # Unet.py:581-603: the zero padding after the projection stays zero
    "dim16": ({**_NARROW, "dim": 16, "num_resnet_blocks": 2}, 32, {}),
    "lowres_96": (dict(dim=8, dim_mults=(1, 2), num_resnet_blocks=(1, 2), layer_attns=False, layer_cross_attns=False, memory_efficient=True, lowres_cond=True), 96, {"lowres": True}),
}


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", sorted(SWEEP))
def test_config_sweep_vs_oracle(backend, case):
    """constructor arguments away from the two BASELINE parameter files (ragged image sizes, three levels, other text /
    conditioning widths, head counts, channel counts), each against the oracle (which equals the reference bit for bit on these)"""
    dev = setup(backend)
    kw, S, extra = SWEEP[case]
    torch.manual_seed(1)
    u = Unet(**kw)
    sd = {k: v.clone() for k, v in u.state_dict().items()}
    u = u.to(dev).eval()
    E, ch = extra.get("E", 512), extra.get("ch", 3)
    emb, mask = R.synthetic_text(2, length=10, seed=3)
    if E != 512:
        emb = torch.randn(2, 10, E, generator=torch.Generator().manual_seed(3)).masked_fill(~mask[:, :, None], 0.)
    x, tm = I.seeded((2, ch, S, S), 6), torch.tensor([3, 9])
    kwargs = dict(text_embeds=emb, text_mask=None if extra.get("nomask") else mask)
    if extra.get("lowres"):
        kwargs.update(lowres_cond_img=I.seeded((2, ch, S, S), 7), lowres_noise_times=torch.tensor([5, 5]))
    ref = R.unet_forward(sd, x, tm, **kwargs)
    out = u(x.to(dev), tm.to(dev), **{k: (v.to(dev) if v is not None else None) for k, v in kwargs.items()})
    assert (out.cpu() - ref).abs().max() < FWD_ATOL


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_self_attention_4096_tokens(backend):
    """K10 at full token count: layer_attns at the 64x64 level (4096 tokens + null row -> 17 context chunks of 256)"""
    dev = setup(backend)
    torch.manual_seed(5)
    u = Unet(dim=8, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(True, True), layer_cross_attns=False, memory_efficient=True, lowres_cond=True)
    sd = {k: v.clone() for k, v in u.state_dict().items()}
    u = u.to(dev).eval()
    emb, mask = R.synthetic_text(2, length=20, seed=3)
    x, lr = I.seeded((2, 3, 128, 128), 6), I.seeded((2, 3, 128, 128), 7)
    tm, lt = torch.tensor([3, 90]), torch.tensor([20, 20])
    ref = R.unet_forward(sd, x, tm, lowres_cond_img=lr, lowres_noise_times=lt, text_embeds=emb, text_mask=mask)
    out = u(x.to(dev), tm.to(dev), lowres_cond_img=lr.to(dev), lowres_noise_times=lt.to(dev), text_embeds=emb.to(dev), text_mask=mask.to(dev))
    assert (out.cpu() - ref).abs().max() < 4e-5
