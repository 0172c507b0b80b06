"""The oracle (oracle/restated.py) against the golden vectors produced by the unmodified
reference, and -- when /root/reference is present -- against the reference directly."""
import numpy as np
import pytest
import torch

from oracle import restated as R
from oracle import ref_loader
from tests import _inputs as I

torch.set_num_threads(8)


@pytest.fixture(scope="module")
def sds():
    return I.load("unet0_sd.pt"), I.load("unet1_sd.pt")


def test_forward_A(sds):
    g = I.load("fwdA.pt"); m = g["meta"]
    emb, mask = I.text(m)
    x = I.seeded((2, 3, 64, 64), m["x_seed"])
    time = torch.tensor(m["time"])
    oc = R.unet_forward(sds[0], x, time, text_embeds=emb, text_mask=mask, cond_drop_prob=0.)
    on = R.unet_forward(sds[0], x, time, text_embeds=emb, text_mask=mask, cond_drop_prob=1.)
    assert torch.allclose(oc, g["out_cond"], atol=2e-6, rtol=1e-6)
    assert torch.allclose(on, g["out_null"], atol=2e-6, rtol=1e-6)


def test_forward_B(sds):
    g = I.load("fwdB.pt"); m = g["meta"]
    emb, mask = I.text(m)
    x = I.seeded((2, 3, 128, 128), m["x_seed"]); lr = I.seeded((2, 3, 128, 128), m["lr_seed"])
    kw = dict(lowres_cond_img=lr, lowres_noise_times=torch.tensor(m["ltime"]), text_embeds=emb, text_mask=mask)
    oc = R.unet_forward(sds[1], x, torch.tensor(m["time"]), cond_drop_prob=0., **kw)
    on = R.unet_forward(sds[1], x, torch.tensor(m["time"]), cond_drop_prob=1., **kw)
    assert torch.allclose(oc, g["out_cond"], atol=2e-6, rtol=1e-6)
    assert torch.allclose(on, g["out_null"], atol=2e-6, rtol=1e-6)


def test_lowres_inputs_required(sds):
    with pytest.raises(AssertionError):
        R.unet_forward(sds[1], torch.zeros(1, 3, 64, 64), torch.tensor([1]), text_embeds=torch.zeros(1, 4, 512))


def test_step(sds):
    g = I.load("step.pt"); m = g["meta"]
    emb, mask = I.text(m)
    sched = R.Schedule(m["T"])
    for t, st in g["steps"].items():
        x = I.seeded((2, 3, 64, 64), st["x_seed"])
        noise = R.make_randn(st["noise_seed"])((2, 3, 64, 64))
        xp, aux = R.p_sample(sds[0], sched, x, t, noise, text_embeds=emb, text_mask=mask, cond_scale=m["cond_scale"])
        assert torch.allclose(aux["pred"], st["pred"], atol=5e-6, rtol=1e-6)
        # quantile of the oracle's own x_start must equal torch.quantile bit-for-bit
        tq = torch.quantile(aux["x_start"].reshape(2, -1).abs(), 0.9, dim=-1)
        sq = R.dynamic_threshold_quantile(aux["x_start"].reshape(2, -1).abs(), 0.9)[0]
        assert torch.equal(tq, sq)
        assert torch.allclose(xp, st["x_prev"], atol=2e-5, rtol=1e-5)


def test_quantile_kats_bit_exact():
    for kat in I.load("quantile.pt"):
        v = I.seeded((3, kat["n"]), kat["seed"]).abs()
        if kat["ties"]:
            v = (v * 4).round() / 4
        s, v_lo, v_hi, lo, w = R.dynamic_threshold_quantile(v, kat["q"])
        assert torch.equal(s, kat["out"]), (kat["n"], s, kat["out"])
        srt = v.sort(dim=-1).values
        assert torch.equal(v_lo, srt[:, lo]) and torch.equal(v_hi, srt[:, lo + 1])


def test_quantile_rank_constants():
    # SURVEY.md Appendix B-1
    assert R.quantile_rank(12288, 0.9) == (11058, np.float32(0.2998046875))
    assert R.quantile_rank(196608, 0.9) == (176946, np.float32(0.296875))


def test_quantile_fuzz_vs_torch():
    g = torch.Generator().manual_seed(99)
    for n in (17, 100, 1000, 12288):
        for _ in range(5):
            v = (torch.randn(8, n, generator=g) * 3).abs()
            s = R.dynamic_threshold_quantile(v, 0.9)[0]
            assert torch.equal(s, torch.quantile(v, 0.9, dim=-1))


def test_schedule_tables_bit_exact():
    tabs = I.load("schedule.pt")
    for T, bufs in tabs.items():
        s = R.Schedule(T)
        for k, v in bufs.items():
            assert torch.equal(getattr(s, k), v), (T, k)
    assert R.Schedule(100).get_times(3, 0.2).tolist() == [20, 20, 20]
    assert R.Schedule(25).get_times(1, 0.2).tolist() == [5]
    assert R.Schedule(25).sampling_timesteps()[0] == 24 and R.Schedule(25).sampling_timesteps()[-1] == 0


@pytest.mark.parametrize("name", ["sample_base_cs1.pt", "sample_base_cs3.pt", "sample_cascade.pt"])
def test_sample(sds, name):
    g = I.load(name); m = g["meta"]
    emb, mask = I.text(m)
    out = R.sample(sds[:len(m["sizes"])], m["sizes"], m["T"], text_embeds=emb, text_masks=mask, cond_scale=m["cond_scale"],
                   randn=R.make_randn(m["noise_seed"]))
    d = (out - g["out"]).abs()
    assert d.max() < 1e-4 and d.mean() < 1e-5, (d.max(), d.mean())


def test_resize_properties():
    # PARITY UNPINNED (no resize_right in the container): property checks only.
    from oracle import resize_restated as RR
    out_sz, pad, fov, w = RR.taps_for_dim(64, 4.0)
    assert out_sz == 256 and pad == (2, 2) and fov.shape == (256, 4)
    assert torch.allclose(w.sum(1), torch.ones(256))
    assert torch.equal(w[0], w[4]) and torch.equal(w[1], w[5])     # 4 repeating phases
    x = torch.full((1, 1, 8, 8), 0.37)
    assert torch.allclose(RR.resize(x, scale_factors=4.0, pad_mode='reflect'), torch.full((1, 1, 32, 32), 0.37), atol=1e-6)
    ramp = torch.arange(16.).reshape(1, 1, 1, 16).expand(1, 1, 16, 16).contiguous()
    up = RR.resize(ramp, scale_factors=2.0, pad_mode='reflect')
    mid = up[0, 0, 8, 8:24]
    assert torch.allclose(mid[1:] - mid[:-1], torch.full((15,), 0.5), atol=1e-5)   # cubic reproduces linear ramps


@pytest.mark.skipif(not ref_loader.reference_available(), reason="reference only exists in the build container")
def test_against_reference_directly(sds):
    ref = ref_loader.load_reference()
    p = I.unet_params()
    u0 = ref.Unet(**p["unet0"]); u0.load_state_dict(sds[0]); u0.eval()
    emb, mask = R.synthetic_text(3, length=40, seed=3)
    x = I.seeded((3, 3, 64, 64), 5)
    time = torch.tensor([0, 24, 7])
    with torch.no_grad():
        a = u0.forward_with_cond_scale(x, time, text_embeds=emb, text_mask=mask, cond_scale=3.)
    b = R.unet_forward_with_cond_scale(sds[0], x, time, cond_scale=3., text_embeds=emb, text_mask=mask)
    assert torch.allclose(a, b, atol=5e-6, rtol=1e-6)
    # an attention-bearing config (self-attention TransformerBlocks + mid attention)
    torch.manual_seed(3)
    ua = ref.Unet(dim=16, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True),
                  attend_at_middle=True, memory_efficient=False).eval()
    xa = I.seeded((2, 3, 32, 32), 6)
    with torch.no_grad():
        a = ua(xa, torch.tensor([3, 9]), text_embeds=emb[:2], text_mask=mask[:2])
    b = R.unet_forward(ua.state_dict(), xa, torch.tensor([3, 9]), text_embeds=emb[:2], text_mask=mask[:2])
    assert torch.allclose(a, b, atol=5e-6, rtol=1e-6)


@pytest.mark.skipif(not ref_loader.reference_available(), reason="reference only exists in the build container")
def test_sweep_configs_against_reference_directly():
    """the constructor-argument sweep of tests/test_unet.py::test_config_sweep_vs_oracle: the oracle equals the unmodified reference"""
    from tests.test_unet import SWEEP
    ref = ref_loader.load_reference()
    for case, (kw, S, extra) in sorted(SWEEP.items()):
        torch.manual_seed(1)
        ru = ref.Unet(**kw).eval()
        sd = {k: v.clone() for k, v in ru.state_dict().items()}
        E, ch = extra.get("E", 512), extra.get("ch", 3)
        emb, mask = R.synthetic_text(2, length=10, seed=3)
        if E != 512:
            emb = torch.randn(2, 10, E, generator=torch.Generator().manual_seed(3)).masked_fill(~mask[:, :, None], 0.)
        x, tm = I.seeded((2, ch, S, S), 6), torch.tensor([3, 9])
        kwargs = dict(text_embeds=emb, text_mask=None if extra.get("nomask") else mask)
        if extra.get("lowres"):
            kwargs.update(lowres_cond_img=I.seeded((2, ch, S, S), 7), lowres_noise_times=torch.tensor([5, 5]))
        with torch.no_grad():
            a = ru(x, tm, **kwargs)
        b = R.unet_forward(sd, x, tm, **kwargs)
        assert torch.allclose(a, b, atol=5e-6, rtol=1e-6), case


def test_cubic_resize_interior_matches_pillow_bicubic():
    """resize-right is neither installed nor vendored, so its restatement cannot be pinned on the package itself.  Pillow's
    BICUBIC is an independent implementation of the same Keys (a = -1/2) kernel with the same pixel-centre convention: away
    from the border (where Pillow renormalises the taps and resize-right reflect-pads) the two must agree to fp32 rounding."""
    from PIL import Image
    from oracle import resize_restated as RR
    g = torch.Generator().manual_seed(0)
    for shape, f, border in (((64, 64), 4.0, 12), ((32, 48), 2.0, 6)):
        x = torch.randn(1, 1, *shape, generator=g)
        up = RR.resize(x, scale_factors=f, pad_mode='reflect')[0, 0].numpy()
        pil = np.asarray(Image.fromarray(x[0, 0].numpy(), mode='F').resize((int(shape[1] * f), int(shape[0] * f)), resample=Image.BICUBIC))
        assert up.shape == pil.shape
        assert np.abs(up - pil)[border:-border, border:-border].max() < 5e-6
