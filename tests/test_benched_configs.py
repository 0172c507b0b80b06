"""The configurations bench.py quotes, VALUE-checked at their own batch sizes (VERDICT r04 "what's weak" #1: every other full-cascade
parity test runs B = 2, T = 25; B = 32 / 64-row guidance launches take code paths B = 2 does not -- the XCD-aware image map at
B % 8 == 0, shared conditioning-independent rows, strips over many images, the grouped sampler tail with 32 images x 8 workgroups).

  * BASELINE's headline configuration: cascade 64 -> 256, per-GPU batch 32, T = 100 per stage, cond_scale 3, fp32 --
      - rows of the B = 32 call == the same rows sampled as B = 2 shards (`_sample_offset`), bit for bit (on-device Philox noise),
      - two rows of the B = 32 call against the oracle run with the same injected noise: max|d| <= 1e-4, mean|d| <= 1e-5 on [0,1] images;
  * BASELINE config 3 as named: the same cascade in the reduced-precision configuration (bf16 activation storage, single fp16 term on
    the matrix cores), batch 16, 256^2 output, against the SAME fp32 oracle rows: max|d| <= 3e-2, mean|d| <= 3e-3."""
import pytest
import torch

from oracle import restated as R
from tests import _inputs as I
from tests._backend import GPU_ONLY, setup
from tests.test_sampler import make_imagen

T, B_FULL, ROWS, NOISE_SEED = 100, 32, [5, 13], 77


def _rows_of_full_stream(seed, rows):
    """the oracle's noise source for a SUBSET of the batch: every draw is made at the full batch size (same generator, same order as the
    B_FULL-row call) and the subset's rows are handed out"""
    full = R.make_randn(seed)
    return lambda shape: full((B_FULL,) + tuple(shape[1:]))[rows].contiguous()


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_headline_batch32_T100_and_config3_batch16_bf16_vs_oracle(backend):
    dev = setup(backend)
    im = make_imagen([64, 256], T, dev)
    emb, mask = R.synthetic_text(B_FULL, length=64, seed=7)            # bench.py's synthetic captions
    embd, maskd = emb.to(dev), mask.to(dev)

    # ---- B = 32 == B = 2 shards, bit for bit (Philox keyed by the global row)
    full = im.sample(text_embeds=embd, text_masks=maskd, cond_scale=3., _seed=1234)
    assert full.shape == (B_FULL, 3, 256, 256) and torch.isfinite(full).all() and full.min() >= 0 and full.max() <= 1
    for r0 in (0, 14, 30):
        shard = im.sample(text_embeds=embd[r0:r0 + 2].contiguous(), text_masks=maskd[r0:r0 + 2].contiguous(), cond_scale=3., _seed=1234, _sample_offset=r0)
        assert torch.equal(shard, full[r0:r0 + 2]), f"rows {r0}..{r0 + 1} of the B = 32 call differ from the B = 2 shard"
    im.check_device_status()

    # ---- two rows of the B = 32 call vs the oracle, T = 100 per stage, same injected noise
    out32 = im.sample(text_embeds=embd, text_masks=maskd, cond_scale=3., _noise=R.make_randn(NOISE_SEED)).cpu()
    sds = [I.load("unet0_sd.pt"), I.load("unet1_sd.pt")]
    ref = R.sample(sds, [64, 256], T, text_embeds=emb[ROWS].contiguous(), text_masks=mask[ROWS].contiguous(), cond_scale=3.,
                   randn=_rows_of_full_stream(NOISE_SEED, ROWS))
    d = (out32[ROWS] - ref).abs()
    print(f"cascade 64->256, B=32, T={T}/stage, cond_scale 3, fp32: rows {ROWS} vs oracle: max|d| = {d.max():.2e}, mean|d| = {d.mean():.2e}")
    assert d.max() < 1e-4 and d.mean() < 1e-5, (d.max(), d.mean())

    # ---- BASELINE config 3: bf16, B = 16, 256^2.  Rows 0..15 of the same noise stream and captions -> the same oracle rows apply
    B3 = 16
    full_stream = R.make_randn(NOISE_SEED)
    first16 = lambda shape: full_stream((B_FULL,) + tuple(shape[1:]))[:B3].contiguous()
    out16 = im.sample(text_embeds=embd[:B3].contiguous(), text_masks=maskd[:B3].contiguous(), cond_scale=3., _noise=first16, _precision="half").cpu()
    for u, S in zip(im.unets, (64, 256)):
        ws = u.engine().workspace(B3, 2 * B3, S, S, precision="half")
        assert ws.half and ws.store16, f"stage {S}: the reduced-precision plan fell back to fp32 storage"
    d3 = (out16[ROWS] - ref).abs()
    print(f"config 3 (bf16 storage, single fp16 term), B=16, T={T}/stage: rows {ROWS} vs the fp32 oracle: max|d| = {d3.max():.2e}, mean|d| = {d3.mean():.2e}")
    assert d3.max() < 3e-2 and d3.mean() < 3e-3, (d3.max(), d3.mean())
    assert d3.max() > 2e-6                                   # the reduced-precision kernels really ran
