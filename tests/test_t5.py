"""K16: the HIP T5 encoder against transformers' T5EncoderModel (the third-party code minimagen/t5.py calls) with
random-init weights and synthetic ids (no checkpoints / tokenizer offline: parity unpinned for pretrained weights)."""
import pytest
import torch

from tests._backend import BACKENDS, setup


def hf_model(ff="relu", layers=2, seed=0):
    from transformers import T5Config, T5EncoderModel
    torch.manual_seed(seed)
    cfg = T5Config(num_layers=layers, feed_forward_proj=ff, d_ff=1024 if ff != "relu" else 2048)    # other fields = t5-small
    m = T5EncoderModel(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "layer_norm" in n:
                p.copy_(1 + 0.2 * torch.randn(p.shape))
            elif "relative_attention_bias" in n:
                p.copy_(torch.randn(p.shape))
            elif "shared" not in n and "embed" not in n:
                p.mul_(0.5)
    return m


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [("relu", 3, 40), ("gated-gelu", 2, 70), ("relu", 1, 200)])
def test_t5_encoder_matches_transformers(backend, case):
    ff, B, Lq = case
    if backend == "emu" and Lq > 100:
        pytest.skip("long sequences only on the GPU (emulator time)")
    dev = setup(backend)
    from minimagen_amd.t5 import T5EncoderHIP
    m = hf_model(ff)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 32128, (B, Lq), generator=g)
    mask = torch.arange(Lq)[None, :] < torch.tensor([Lq - 7 * r for r in range(B)])[:, None]
    with torch.no_grad():
        ref = m(input_ids=ids, attention_mask=mask.long()).last_hidden_state
    ref = ref.masked_fill(~mask[:, :, None], 0.)                      # minimagen/t5.py:82
    enc = T5EncoderHIP.from_hf(m, device=dev)
    out, mk = enc.encode(ids.to(dev), mask.to(dev))
    assert torch.equal(mk.cpu(), mask)
    d = (out.cpu() - ref).abs()
    assert d.max() < 2e-4 and d.mean() < 2e-5, (d.max(), d.mean(), ref.abs().max())


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(100, 96, 64, False, 0, 1.0, 1.0), (70, 128, 96, True, 2, 1.0, 1.0), (64, 64, 256, False, 1, 2.0 ** 12, 2.0 ** -14),
                                  (33, 64, 64, False, 3, 2.0 ** -18, 2.0 ** 9)])
def test_gemm_f16x3_block_scaled(backend, case):
    """mi_gemm_f32 on the f16 matrix-core instruction (3-term fp16 splits, per-K-slice block scaling) against fp64: C = act(A W^T) [* A G^T] + R,
    ragged M / N, operands far from unit scale and rows / columns whose magnitudes differ by 2^20 inside one tile -- the error budget is a few
    fp32 roundings of sum_k |a||w| (what an fp32 GEMM itself is allowed)"""
    import ctypes as C
    from minimagen_amd import _lib as L
    dev = setup(backend)
    lib = L.lib()
    M, N, K, gated, act, sa, sw = case
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g) * sa
    W = torch.randn(N, K, generator=g) * sw
    A[::7] *= 2.0 ** -20                       # tiny rows next to large ones in the same 64 x 32 tile
    W[1::5] *= 2.0 ** -12
    A[3, 5] = 0.0
    G = torch.randn(N, K, generator=g) if gated else None
    Rm = torch.randn(M, N, generator=g) * sa * sw
    pre = A.double() @ W.double().t()
    ref = pre.clamp(min=0) if act == 1 else (torch.nn.functional.gelu(pre, approximate="tanh") if act == 2 else (torch.nn.functional.gelu(pre) if act == 3 else pre))
    bound = A.abs().double() @ W.abs().double().t()
    if gated:
        gate = A.double() @ G.double().t()
        bound = bound * gate.abs() + ref.abs() * (A.abs().double() @ G.abs().double().t())
        ref = ref * gate
    ref = ref + Rm.double()
    Ad, Wd, Rd = A.to(dev), W.to(dev), Rm.to(dev)
    Gd = G.to(dev) if gated else None
    out = torch.full((M, N), float("nan"), device=dev)
    L.check(lib.mi_gemm_f32(L.ptr(Ad), L.ptr(Wd), L.ptr(Gd), L.ptr(Rd), L.ptr(out), M, N, K, act, L.current_stream()), "mi_gemm_f32")
    err = (out.cpu().double() - ref).abs()
    tol = 4e-6 * bound + 1e-6 * ref.abs() + 1e-30
    assert torch.isfinite(out).all() and bool((err <= tol).all()), (err / tol).max()


@pytest.mark.gpu
def test_t5_encoder_bench_shape():
    """the shape bench.py's t5_encode leg runs: T5Config() = t5-small, 6 layers, B=32, L=64, ragged masks"""
    from transformers import T5Config, T5EncoderModel
    from minimagen_amd.t5 import T5EncoderHIP
    dev = setup("gpu")
    torch.manual_seed(0)
    m = T5EncoderModel(T5Config()).eval()
    B, Lq = 32, 64
    ids = torch.randint(0, 32128, (B, Lq), generator=torch.Generator().manual_seed(1))
    keep = torch.tensor([max(1, Lq - (r % 24)) for r in range(B)])
    mask = torch.arange(Lq)[None, :] < keep[:, None]
    with torch.no_grad():
        ref = m(input_ids=ids, attention_mask=mask.long()).last_hidden_state
    ref = ref.masked_fill(~mask[:, :, None], 0.)
    out, mk = T5EncoderHIP.from_hf(m, device=dev).encode(ids.to(dev), mask.to(dev))
    d = (out.cpu() - ref).abs()
    print(f"T5 6 layers B=32 L=64: max|d| = {d.max():.2e}, mean|d| = {d.mean():.2e}, |ref|max = {ref.abs().max():.3g}")
    assert torch.equal(mk.cpu(), mask) and d.max() < 2e-4 * max(1.0, ref.abs().max().item()) and d.mean() < 2e-5


def test_relative_position_bucket_matches_transformers():
    from transformers.models.t5.modeling_t5 import T5Attention
    from minimagen_amd.t5 import relative_position_bucket
    rel = torch.arange(-255, 256)
    assert torch.equal(relative_position_bucket(rel), T5Attention._relative_position_bucket(rel, bidirectional=True, num_buckets=32, max_distance=128))


def test_t5_api_surface():
    from minimagen_amd import t5
    assert t5.get_encoded_dim("t5_small") == 512 and t5.get_encoded_dim("xxl1.1") == 4096 and t5.MAX_LENGTH == 256
    with pytest.raises(Exception):          # no tokenizer / checkpoint files offline: must fail loudly, never silently stub
        t5.t5_encode_text(["a cat"], name="t5_small")
