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


# ---------------------------------------------------------------------- the text front-end (minimagen/t5.py:24-84), offline
WORDS = "a photo of the cat dog red blue green house tree sitting on under near big small painting oil water color bright dark happy sad bird fish sky sea mountain river".split()


def _fabricate_checkpoint(path, ff="relu", seed=0):
    """A Hugging Face checkpoint DIRECTORY made here, offline: config.json + model.safetensors of a randomised T5EncoderModel (t5-small's width
    so that it can stand in for 't5_small' / 'small1.1') and the files of a Unigram T5Tokenizer over a small word list."""
    from transformers import T5Config, T5EncoderModel, T5Tokenizer
    vocab = [("<pad>", 0.0), ("</s>", 0.0), ("<unk>", 0.0), ("▁", -2.0)] + [("▁" + w, -3.0 - 0.01 * i) for i, w in enumerate(WORDS)] \
        + [(c, -6.0) for c in "abcdefghijklmnopqrstuvwxyz"]
    tok = T5Tokenizer(vocab=vocab, extra_ids=0)
    torch.manual_seed(seed)
    cfg = T5Config(vocab_size=len(tok), d_model=512, d_kv=64, num_heads=8, d_ff=1024, num_layers=2, feed_forward_proj=ff,
                   tie_word_embeddings=(ff == "relu"))
    m = T5EncoderModel(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "layer_norm" in n:
                p.copy_(1 + 0.2 * torch.randn(p.shape))
            elif "relative_attention_bias" in n:
                p.copy_(torch.randn(p.shape))
            elif "shared" not in n and "embed" not in n:
                p.mul_(0.5)
    m.save_pretrained(path)
    tok.save_pretrained(path)
    return m, tok


def _hf_reference(m, tok, texts, max_length=256):
    enc = tok(texts, padding='longest', max_length=max_length, truncation=True, return_tensors="pt")
    with torch.no_grad():
        ref = m(input_ids=enc.input_ids, attention_mask=enc.attention_mask).last_hidden_state
    return ref.masked_fill(~enc.attention_mask.bool()[:, :, None], 0.), enc.attention_mask.bool()


@pytest.fixture
def clean_t5_registry():
    from minimagen_amd import t5 as T
    saved = {k: dict(v) for k, v in T.T5_VERSIONS.items()}
    T._embed_cache_clear()
    yield T
    T.T5_VERSIONS.clear()
    T.T5_VERSIONS.update(saved)
    T._embed_cache_clear()


@pytest.mark.parametrize("backend", BACKENDS)
def test_text_front_end_from_local_checkpoints(backend, tmp_path, monkeypatch, clean_t5_registry):
    """t5_encode_text(list[str]) end to end without a network: checkpoint directories under $MINIMAGEN_T5_DIR are read directly
    (T5EncoderHIP.from_directory: config.json + safetensors, no from_pretrained / hub call), the tokenizer comes from the same directory;
    original (ReLU, tied embedding) and v1.1 (gated-GELU, untied) checkpoints; values against transformers' own forward of the same
    checkpoint and tokenizer (minimagen/t5.py:63-84); the caption cache returns the very same bits."""
    T = clean_t5_registry
    dev = setup(backend)
    m0, tok0 = _fabricate_checkpoint(str(tmp_path / "t5-small"), "relu", seed=0)
    m1, tok1 = _fabricate_checkpoint(str(tmp_path / "google--t5-v1_1-small"), "gated-gelu", seed=1)
    monkeypatch.setenv(T.T5_LOCAL_DIR_ENV, str(tmp_path))
    monkeypatch.setattr("huggingface_hub.snapshot_download", lambda *a, **k: (_ for _ in ()).throw(AssertionError("hub call")), raising=False)
    texts = ["a photo of the cat", "big red house near the sea under dark sky", "oil painting of a happy zebra fish on the mountain river near a small tree"]
    for name, m, tok in (("t5_small", m0, tok0), ("small1.1", m1, tok1)):
        for v in ("model", "tokenizer"):
            T.T5_VERSIONS[name][v] = None
        emb, mask = T.t5_encode_text(texts, name=name)
        ref, rmask = _hf_reference(m, tok, texts)
        assert emb.shape == ref.shape and torch.equal(mask.cpu(), rmask)
        d = (emb.cpu() - ref).abs()
        assert d.max() < 2e-4 and d.mean() < 2e-5, (name, d.max(), d.mean())
        assert isinstance(T.T5_VERSIONS[name]["model"], T.T5EncoderHIP) and T.T5_VERSIONS[name]["model"].gated == (name == "small1.1")
        # the caption cache: same captions again -> no encoder launch, the very same bits; a re-ordered, differently padded batch with one new
        # caption -> cached and fresh rows agree with a cold encode of that batch to fp32 rounding (the block-scaled GEMM shares one
        # power-of-two scale among the 64 rows of a tile, so the last bits of a row depend on its batch neighbours)
        h0 = T.t5_cache_stats["hits"]
        emb2, _ = T.t5_encode_text(texts, name=name)
        assert T.t5_cache_stats["hits"] == h0 + 3 and torch.equal(emb2, emb)
        mixed = [texts[1], "the dog", texts[0]]
        warm, wmask = T.t5_encode_text(mixed, name=name)
        T._embed_cache_clear()
        cold, cmask = T.t5_encode_text(mixed, name=name)
        assert torch.equal(wmask, cmask) and (warm - cold).abs().max() < 2e-5
        direct, _ = T.T5_VERSIONS[name]["model"].encode(*[t.to(dev) for t in T._tokenize(tok, mixed, 256)])
        assert (direct - cold).abs().max() < 2e-5                # ... and with the uncached encoder call
    # max_length truncation (t5.py:65-66) and a caller-supplied tokenizer callable
    emb_t, mask_t = T.t5_encode_text(texts, name="t5_small", max_length=6)
    ref_t, rmask_t = _hf_reference(m0, tok0, texts, max_length=6)
    assert emb_t.shape[1] == 6 and torch.equal(mask_t.cpu(), rmask_t) and (emb_t.cpu() - ref_t).abs().max() < 2e-4

    def my_tok(caps, max_length):
        e = tok0(caps, padding='longest', max_length=max_length, truncation=True, return_tensors="pt")
        return e.input_ids, e.attention_mask
    emb_c, _ = T.t5_encode_text(texts, name="t5_small", tokenizer=my_tok)
    assert (emb_c - T.t5_encode_text(texts, name="t5_small")[0]).abs().max() < 2e-5


@pytest.mark.parametrize("backend", BACKENDS)
def test_imagen_sample_from_captions(backend, tmp_path, monkeypatch, clean_t5_registry):
    """Imagen.sample(texts=[...]) -- the reference's caption entry point (Imagen.py:455-458 -> t5.py:31-84) -- end to end on the device path,
    against the oracle fed with transformers' embeddings of the same captions"""
    from minimagen_amd.Imagen import Imagen
    from minimagen_amd.Unet import Unet
    from oracle import restated as R
    T = clean_t5_registry
    dev = setup(backend)
    m0, tok0 = _fabricate_checkpoint(str(tmp_path / "t5-small"), "relu", seed=3)
    T.register_t5("t5_small", model_dir=str(tmp_path / "t5-small"), tokenizer=tok0, device=dev)
    torch.manual_seed(5)
    kw = dict(dim=8, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=False, layer_cross_attns=(False, True), memory_efficient=False)
    im = Imagen([Unet(**kw)], text_encoder_name="t5_small", image_sizes=[32], timesteps=25, cond_drop_prob=0.15)
    sd = {k: v.clone() for k, v in im.unets[0].state_dict().items()}
    im = im.to(dev)
    texts = ["a photo of the cat", "big red house near the sea under dark sky"]
    out = im.sample(texts=texts, cond_scale=2., _noise=R.make_randn(9))
    ref_emb, ref_mask = _hf_reference(m0, tok0, texts)
    ref = R.sample([sd], [32], 25, text_embeds=ref_emb, text_masks=ref_mask, cond_scale=2., randn=R.make_randn(9))
    d = (out.cpu() - ref).abs()
    print(f"Imagen.sample(texts=...) 32x32 T=25 vs oracle on transformers' embeddings: max|d| = {d.max():.2e}")
    assert out.shape == (2, 3, 32, 32) and d.max() < 2e-4 and d.mean() < 2e-5, (d.max(), d.mean())
