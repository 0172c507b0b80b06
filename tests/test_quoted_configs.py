"""Every configuration a `bench.py` leg or a profile QUOTES A NUMBER ON, value-checked at the batch size it is quoted at (VERDICT r05
"next round" item 2; the headline and config 3 are in tests/test_benched_configs.py).  The benched batch takes code paths the small parity
cases never reach -- 32-row launches take the XCD-aware image map and the 64-channel-per-workgroup form of `conv_wide`, the grouped sampler
tail runs with many images, the 1024^2 stage keeps the separate tail kernels -- so each leg gets: rows of the full-batch call against the
same rows sampled as B = 2 shards (Imagen.py:424-510 has no coupling between samples), and two rows against the oracle on the same
injected noise (the oracle draws every noise tensor at the full batch size and keeps its rows).

  (a) `secondary.wide_unet_default_B16`: ``Unet()`` default (Unet.py:31-48) @64^2, B = 16, cond_scale 3, T = 25;
  (b) `profiles/r05_wide_presets_base_super.txt`: ``Base()`` / ``Super()`` (Unet.py:637-692) at FULL width, one forward each, B = 1;
  (c) `secondary.config5_...`: three-stage cascade 64 -> 256 -> 1024, B = 8, cond_scale 3, reduced precision, T = 25;
  (d) one hostile-weights CASCADE (64 -> 256, T = 25, B = 2): the fp32 label on trained-like weight ranges over 50 dependent steps."""
import gc

import pytest
import torch

from minimagen_amd.Imagen import Imagen
from minimagen_amd.Unet import Base, Super, Unet
from oracle import restated as R
from tests import _inputs as I
from tests._backend import GPU_ONLY, setup
from tests.test_sampler import make_imagen
from tests.test_unet import FWD_ATOL


def _rows_of_full_stream(seed, rows, b_full):
    full = R.make_randn(seed)
    return lambda shape: full((b_full,) + tuple(shape[1:]))[rows].contiguous()


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_wide_unet_default_batch16_guided_vs_shards_and_oracle(backend):
    """(a) the `wide_unet_default_B16` leg of bench.py: 32-row launches of the wide regime (conv_wide 64-/128-channel forms, prepared-K/V
    multi-query attention, stacked time-MLP GEMM over the batch rows)"""
    dev = setup(backend)
    B, T, ROWS = 16, 25, [3, 12]
    torch.manual_seed(6)
    u = Unet()
    sd = {k: v.clone() for k, v in u.state_dict().items()}
    im = Imagen((u,), text_encoder_name="t5_small", image_sizes=(64,), timesteps=T, cond_drop_prob=0.1).to(dev).eval()
    emb, mask = R.synthetic_text(B, length=64, seed=7)
    embd, maskd = emb.to(dev), mask.to(dev)
    full = im.sample(text_embeds=embd, text_masks=maskd, cond_scale=3., _seed=1234)
    assert full.shape == (B, 3, 64, 64) and torch.isfinite(full).all()
    for r0 in (0, 6, 14):
        shard = im.sample(text_embeds=embd[r0:r0 + 2].contiguous(), text_masks=maskd[r0:r0 + 2].contiguous(), cond_scale=3., _seed=1234, _sample_offset=r0)
        assert torch.equal(shard, full[r0:r0 + 2]), f"rows {r0}..{r0 + 1} of the B = 16 call differ from the B = 2 shard"
    out = im.sample(text_embeds=embd, text_masks=maskd, cond_scale=3., _noise=R.make_randn(55)).cpu()
    ref = R.sample([sd], [64], T, text_embeds=emb[ROWS].contiguous(), text_masks=mask[ROWS].contiguous(), cond_scale=3.,
                   randn=_rows_of_full_stream(55, ROWS, B))
    d = (out[ROWS] - ref).abs()
    print(f"Unet() default @64, B={B}, T={T}, cond_scale 3: rows {ROWS} vs oracle: max|d| = {d.max():.2e}, mean|d| = {d.mean():.2e}")
    assert d.max() < 1e-4 and d.mean() < 1e-5, (d.max(), d.mean())
    im.check_device_status()


@pytest.mark.parametrize("backend", GPU_ONLY)
@pytest.mark.parametrize("preset", ["Base", "Super"])
def test_full_width_presets_one_forward_vs_oracle(backend, preset):
    """(b) the reference's presets at their real widths (Base: dim 512, 1.6 B parameters, 64^2; Super: dim 128, dim_mults (1, 2, 4, 8),
    256^2, low-res conditioned), one guided evaluation at B = 1 against the oracle on the host (slow: the oracle is ~1 TFLOP on CPU)"""
    dev = setup(backend)
    klass, S = (Base, 64) if preset == "Base" else (Super, 256)
    torch.manual_seed(7)
    m = klass()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    emb, mask = R.synthetic_text(1, length=20, seed=8)
    x, tm = I.seeded((1, 3, S, S), 43), torch.tensor([50])
    extra = dict(lowres_cond_img=I.seeded((1, 3, S, S), 44), lowres_noise_times=torch.tensor([20])) if m.lowres_cond else {}
    ref = R.unet_forward_with_cond_scale(sd, x, tm, cond_scale=3., text_embeds=emb, text_mask=mask, **extra)
    del sd
    gc.collect()
    m = m.to(dev).eval()
    o = m.forward_with_cond_scale(x.to(dev), tm.to(dev), text_embeds=emb.to(dev), text_mask=mask.to(dev), cond_scale=3.,
                                  **{k: v.to(dev) for k, v in extra.items()}).cpu()
    d = (o - ref).abs().max().item()
    print(f"{preset}() full width @{S}, B=1, cond_scale 3: max|d| = {d:.2e} (|ref| max {ref.abs().max():.2f}, "
          f"{sum(p.numel() for p in m.parameters()) / 1e6:.0f} M parameters)")
    assert torch.isfinite(o).all() and d < 3 * FWD_ATOL * max(1.0, ref.abs().max().item())
    del m
    gc.collect()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_config5_batch8_guided_reduced_precision_vs_shards_and_oracle(backend):
    """(c) bench.py's config-5 leg as it is benched: 64 -> 256 -> 1024, B = 8, cond_scale 3, reduced precision (bf16 storage), noise
    augmentation on both SR stages, T = 25: shard bit identity (the statistics partition and every tile depend on the image size only)
    and two rows against the fp32 oracle under the half-precision gate of SURVEY 8(c) (3e-2 / 3e-3)"""
    dev = setup(backend)
    B, T, ROWS = 8, 25, [2, 7]
    im = make_imagen([64, 256, 1024], T, dev)
    emb, mask = R.synthetic_text(B, length=64, seed=7)
    embd, maskd = emb.to(dev), mask.to(dev)
    kw = dict(cond_scale=3., lowres_sample_noise_level=0.2, _precision="half")
    full = im.sample(text_embeds=embd, text_masks=maskd, _seed=99, **kw)
    assert full.shape == (B, 3, 1024, 1024) and torch.isfinite(full).all()
    for r0 in (0, 6):
        shard = im.sample(text_embeds=embd[r0:r0 + 2].contiguous(), text_masks=maskd[r0:r0 + 2].contiguous(), _seed=99, _sample_offset=r0, **kw)
        assert torch.equal(shard, full[r0:r0 + 2]), f"rows {r0}..{r0 + 1} of the B = 8 call differ from the B = 2 shard"
    del full, shard
    for u, S in zip(im.unets, (64, 256, 1024)):
        ws = u.engine().workspace(B, 2 * B, S, S, precision="half")
        assert ws.half and ws.store16, f"stage {S}: the reduced-precision plan fell back to fp32 storage"
    out = im.sample(text_embeds=embd, text_masks=maskd, _noise=R.make_randn(31), **kw)[ROWS].cpu()
    im.check_device_status()
    sd0, sd1 = I.load("unet0_sd.pt"), I.load("unet1_sd.pt")
    ref = R.sample([sd0, sd1, sd1], [64, 256, 1024], T, text_embeds=emb[ROWS].contiguous(), text_masks=mask[ROWS].contiguous(), cond_scale=3.,
                   randn=_rows_of_full_stream(31, ROWS, B), lowres_sample_noise_level=0.2)
    d = (out - ref).abs()
    print(f"config 5 (64->256->1024, B={B}, cond_scale 3, bf16 storage), T={T}: rows {ROWS} vs the fp32 oracle: max|d| = {d.max():.2e}, mean|d| = {d.mean():.2e}")
    assert d.max() < 3e-2 and d.mean() < 3e-3, (d.max(), d.mean())
    assert d.max() > 2e-6


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_hostile_weights_cascade_vs_oracle(backend, monkeypatch):
    """(d) the hostile weight ranges of test_hostile_weight_ranges_vs_fp64_oracle (per-layer scales 2^-10 .. 2^6, outlier weights, outlier
    normalisation channels) through a whole CASCADE -- 64 -> 256, T = 25 per stage, B = 2, cond_scale 3: 50 dependent denoising steps, the
    dynamic threshold and the low-res hand-over on trained-like ranges.  Gate: the cascade gate of SURVEY 8(c) (1e-4 / 1e-5 on [0,1]
    images), unless the reference's own fp32 arithmetic is further than that from the fp64 value of the algorithm on these weights --
    then 4 x its distance (the yardstick of the single-forward hostile test)."""
    dev = setup(backend)
    T, B = 25, 2
    p = I.unet_params()
    sds = [I.hostile_state_dict(I.load("unet0_sd.pt"), 21), I.hostile_state_dict(I.load("unet1_sd.pt"), 22)]
    im = Imagen([Unet(**p["unet0"]), Unet(**p["unet1"])], text_encoder_name="t5_small", image_sizes=[64, 256], timesteps=T, cond_drop_prob=0.15)
    for u, sd in zip(im.unets, sds):
        u.load_state_dict(sd)
    im = im.to(dev)
    emb, mask = R.synthetic_text(B, length=24, seed=5)
    out = im.sample(text_embeds=emb.to(dev), text_masks=mask.to(dev), cond_scale=3., _noise=R.make_randn(88)).cpu()
    im.check_device_status()
    ref = R.sample(sds, [64, 256], T, text_embeds=emb, text_masks=mask, cond_scale=3., randn=R.make_randn(88))
    # yardstick: the same sampler with every U-Net evaluation done in fp64 (rounded to fp32 for the fp32 sampler arithmetic the reference has)
    sds64 = {id(sd): {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()} for sd in sds}
    fwd32 = R.unet_forward_with_cond_scale

    def fwd64(sd, x, time, cond_scale=1., **kw):
        kw64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in kw.items()}
        R.COMPUTE_DTYPE = torch.float64
        try:
            return fwd32(sds64[id(sd)], x.double(), time, cond_scale=cond_scale, **kw64).float()
        finally:
            R.COMPUTE_DTYPE = torch.float32
    monkeypatch.setattr(R, "unet_forward_with_cond_scale", fwd64)
    ref64 = R.sample(sds, [64, 256], T, text_embeds=emb, text_masks=mask, cond_scale=3., randn=R.make_randn(88)).double()
    monkeypatch.undo()
    d, d32 = (out - ref).abs(), (ref.double() - ref64).abs()
    d64 = (out.double() - ref64).abs()
    print(f"hostile-weights cascade 64->256, T={T}, B={B}, cond_scale 3: |hip - oracle32| max {d.max():.2e} mean {d.mean():.2e}; "
          f"|hip - ref64| max {d64.max():.2e}; |oracle32 - ref64| max {d32.max():.2e} mean {d32.mean():.2e}")
    assert torch.isfinite(out).all() and torch.isfinite(ref).all()
    assert d64.max() <= max(1e-4, 4 * d32.max().item()) and d64.mean() <= max(1e-5, 4 * d32.mean().item()), (d64.max(), d64.mean(), d32.max(), d32.mean())
