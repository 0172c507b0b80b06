"""Text-encoder front-end with the reference's names (minimagen/t5.py), T5 encoder stack on HIP kernels (K16)."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch

from . import _lib as L

MAX_LENGTH = 256
DEFAULT_T5_NAME = 't5_small'

# minimagen/t5.py:10-21
T5_VERSIONS = {
    't5_small': {'tokenizer': None, 'model': None, 'handle': 't5-small', 'dim': 512, 'size': .24},
    't5_base': {'tokenizer': None, 'model': None, 'handle': 't5-base', 'dim': 768, 'size': .890},
    't5_large': {'tokenizer': None, 'model': None, 'handle': 't5-large', 'dim': 1024, 'size': 2.75},
    't5_3b': {'tokenizer': None, 'model': None, 'handle': 't5-3b', 'dim': 1024, 'size': 10.6},
    't5_11b': {'tokenizer': None, 'model': None, 'handle': 't5-11b', 'dim': 1024, 'size': 42.1},
    'small1.1': {'tokenizer': None, 'model': None, 'handle': 'google/t5-v1_1-small', 'dim': 512, 'size': .3},
    'base1.1': {'tokenizer': None, 'model': None, 'handle': 'google/t5-v1_1-base', 'dim': 768, 'size': .99},
    'large1.1': {'tokenizer': None, 'model': None, 'handle': 'google/t5-v1_1-large', 'dim': 1024, 'size': 3.13},
    'xl1.1': {'tokenizer': None, 'model': None, 'handle': 'google/t5-v1_1-xl', 'dim': 2048, 'size': 11.4},
    'xxl1.1': {'tokenizer': None, 'model': None, 'handle': 'google/t5-v1_1-xxl', 'dim': 4096, 'size': 44.5},
}


def get_encoded_dim(name: str) -> int:
    """minimagen/t5.py:87-90"""
    return T5_VERSIONS[name]['dim']


def relative_position_bucket(relative_position: torch.Tensor, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """T5's bidirectional bucket function (transformers T5Attention._relative_position_bucket), same integer / fp32
    operations so the bucket indices are identical."""
    num_buckets //= 2
    buckets = (relative_position > 0).to(torch.long) * num_buckets
    rp = torch.abs(relative_position)
    max_exact = num_buckets // 2
    is_small = rp < max_exact
    if_large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)).to(torch.long)
    if_large = torch.min(if_large, torch.full_like(if_large, num_buckets - 1))
    return buckets + torch.where(is_small, rp, if_large)


class T5EncoderHIP:
    """The T5 encoder stack (embedding -> N x {RMSNorm, self-attention with shared relative-position bias, RMSNorm,
    feed-forward} -> final RMSNorm) executed by the kernels of csrc/t5.hip from a transformers ``T5EncoderModel``
    state dict.  fp32; no biases; attention is unscaled (T5)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], *, d_model: int, d_kv: int, num_heads: int, d_ff: int, num_layers: int,
                 relative_attention_num_buckets: int = 32, relative_attention_max_distance: int = 128,
                 layer_norm_epsilon: float = 1e-6, feed_forward_proj: str = "relu", device="cuda"):
        if d_kv != 64:
            raise NotImplementedError("T5 attention kernel is instantiated for d_kv = 64")
        self.cfg = dict(d_model=d_model, d_kv=d_kv, heads=num_heads, d_ff=d_ff, layers=num_layers, nb=relative_attention_num_buckets,
                        md=relative_attention_max_distance, eps=layer_norm_epsilon)
        self.gated = feed_forward_proj.startswith("gated")
        self.act = 1 if feed_forward_proj == "relu" else 2
        dev = torch.device(device)
        f = lambda k: state_dict[k].detach().to(dev, torch.float32).contiguous()
        self.emb = f("shared.weight") if "shared.weight" in state_dict else f("encoder.embed_tokens.weight")
        self.rel_emb_cpu = state_dict["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].detach().to("cpu", torch.float32)
        self.layers = []
        for i in range(num_layers):
            p = f"encoder.block.{i}.layer."
            ff = p + "1.DenseReluDense."
            # T5LayerNorm's weight is folded into the projection behind it (y = rsqrt(mean x^2 + eps) * (x . (W diag w)^T)): the norm itself is the
            # row scaling of mi_gemm_rms_f32's epilogue, fed by the sums of squares the previous projection's epilogue left (no rmsnorm launches)
            ln0, ln1 = f(p + "0.layer_norm.weight"), f(p + "1.layer_norm.weight")
            self.layers.append(dict(
                wqkv=(torch.cat([f(p + f"0.SelfAttention.{n}.weight") for n in "qkv"], 0) * ln0[None, :]).contiguous(),
                wo=f(p + "0.SelfAttention.o.weight"),
                wi=(f(ff + ("wi_0.weight" if self.gated else "wi.weight")) * ln1[None, :]).contiguous(),
                wg=(f(ff + "wi_1.weight") * ln1[None, :]).contiguous() if self.gated else None,
                wo2=f(ff + "wo.weight")))
        self.final_ln = f("encoder.final_layer_norm.weight")
        self.dev = dev
        L.require_device(self.emb)

    @classmethod
    def from_hf(cls, model, device="cuda"):
        c = model.config
        return cls(model.state_dict(), d_model=c.d_model, d_kv=c.d_kv, num_heads=c.num_heads, d_ff=c.d_ff, num_layers=c.num_layers,
                   relative_attention_num_buckets=c.relative_attention_num_buckets,
                   relative_attention_max_distance=getattr(c, "relative_attention_max_distance", 128),
                   layer_norm_epsilon=c.layer_norm_epsilon, feed_forward_proj=c.feed_forward_proj, device=device)

    def bias_table(self, Lq: int) -> torch.Tensor:
        """[heads][2L-1]: relative_attention_bias[bucket(j - i)] for j - i = -(L-1) .. L-1 (layer 0's table, shared by all layers)"""
        rel = torch.arange(-(Lq - 1), Lq)
        b = relative_position_bucket(rel, self.cfg["nb"], self.cfg["md"])
        return self.rel_emb_cpu[b].t().contiguous().to(self.dev)

    def _plan(self, B: int, Lq: int):
        """static buffers + (later) the captured HIP graph of one (batch, length) shape: 32 launches become one replay"""
        plans = self.__dict__.setdefault("_plans", {})
        pl = plans.get((B, Lq))
        if pl is None:
            c = self.cfg
            M, d, inner = B * Lq, c["d_model"], c["heads"] * c["d_kv"]
            e = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.dev)
            pl = dict(ids=torch.zeros(B, Lq, dtype=torch.int64, device=self.dev), mask=torch.ones(B, Lq, dtype=torch.uint8, device=self.dev),
                      h=e(M, d), sq=e(M, -(-d // 64)), sq2=e(M, -(-d // 64)), qkv=e(M, 3 * inner), ctx=e(M, inner), ff=e(M, c["d_ff"]), h2=e(M, d), out=e(M, d),
                      bias=self.bias_table(Lq), graph=None)
            while len(plans) >= 8:                         # bounded: one plan per shape
                old = plans.pop(next(iter(plans)))
                if old["graph"] is not None:
                    if L.backend() == "hip-gfx950":
                        torch.cuda.synchronize(self.dev)
                    L.lib().mi_graph_destroy(old["graph"])
            plans[(B, Lq)] = pl
        return pl

    def _launch(self, pl, B: int, Lq: int, st):
        lib, c = L.lib(), self.cfg
        M, d, inner = B * Lq, c["d_model"], c["heads"] * c["d_kv"]
        h, qkv, ctx, ff, h2, bias, mask = pl["h"], pl["qkv"], pl["ctx"], pl["ff"], pl["h2"], pl["bias"], pl["mask"]
        sq, sq2, np_ = pl["sq"], pl["sq2"], -(-d // 64)           # per-row sums of squares of h / h2, one partial per 64-column tile of their producer
        eps = c["eps"]
        L.check(lib.mi_embed_rows_sq(L.ptr(pl["ids"]), L.ptr(self.emb), L.ptr(h), L.ptr(sq), np_, M, d, st), "mi_embed_rows_sq")
        for ly in self.layers:        # five launches per layer: RMSNorm + QKV, attention, O + residual, RMSNorm + FF-in (+ gate, activation), FF-out + residual
            L.check(lib.mi_gemm_rms_f32(L.ptr(h), L.ptr(ly["wqkv"]), None, None, L.ptr(qkv), M, 3 * inner, d, 0, L.ptr(sq), np_, eps, None, st), "mi_gemm_rms_f32 qkv")
            L.check(lib.mi_t5_attention(L.ptr(qkv), L.ptr(bias), L.ptr(mask), L.ptr(ctx), B, Lq, c["heads"], st), "mi_t5_attention")
            L.check(lib.mi_gemm_rms_f32(L.ptr(ctx), L.ptr(ly["wo"]), None, L.ptr(h), L.ptr(h2), M, d, inner, 0, None, 0, 0.0, L.ptr(sq2), st), "mi_gemm_rms_f32 o")
            L.check(lib.mi_gemm_rms_f32(L.ptr(h2), L.ptr(ly["wi"]), L.ptr(ly["wg"]), None, L.ptr(ff), M, c["d_ff"], d, self.act, L.ptr(sq2), np_, eps, None, st), "mi_gemm_rms_f32 wi")
            L.check(lib.mi_gemm_rms_f32(L.ptr(ff), L.ptr(ly["wo2"]), None, L.ptr(h2), L.ptr(h), M, d, c["d_ff"], 0, None, 0, 0.0, L.ptr(sq), st), "mi_gemm_rms_f32 wo")
        L.check(lib.mi_rmsnorm(L.ptr(h), L.ptr(self.final_ln), L.ptr(pl["out"]), M, d, c["eps"], L.ptr(mask), st), "mi_rmsnorm final")

    @torch.no_grad()
    def encode(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, _use_graph: bool = True):
        """-> (last_hidden_state with masked rows zeroed (t5.py:82), bool mask).  The launches of one (batch, length) shape are captured in a
        HIP graph on first use and replayed afterwards (static buffers; the result is copied out)."""
        lib = L.lib()
        B, Lq = input_ids.shape
        if Lq > MAX_LENGTH:
            raise ValueError(f"sequence length {Lq} exceeds MAX_LENGTH {MAX_LENGTH} (t5.py:5)")
        ids = input_ids.to(self.dev, torch.int64).contiguous()
        mask = None if attention_mask is None else attention_mask.to(self.dev).to(torch.uint8).contiguous()
        L.require_device(ids, mask)
        pl = self._plan(B, Lq)

        def stage():
            pl["ids"].copy_(ids)
            if mask is None:
                pl["mask"].fill_(1)
            else:
                pl["mask"].copy_(mask)

        if _use_graph and L.backend() == "hip-gfx950":
            # Everything that touches the plan's static buffers -- the staging copies, the replay, the copy-out -- runs on the encoder's ONE
            # private stream: encode() calls of the same shape from different caller streams are then ordered among themselves (a second
            # call cannot overwrite ids / mask while the first replay still reads them), each caller stream only waits for its own result.
            # (The captured graph addresses the encoder's own weight tensors; they are never re-homed: T5EncoderHIP owns private copies.)
            cur = torch.cuda.current_stream(self.dev)
            side = self.__dict__.get("_stream")
            if side is None:
                side = self._stream = torch.cuda.Stream(device=self.dev)     # (graphs cannot be captured on the legacy default stream)
            side.wait_stream(cur)
            for t_in in (ids, mask):
                if t_in is not None:
                    t_in.record_stream(side)
            with torch.cuda.stream(side):
                stage()
                st = side.cuda_stream
                if pl["graph"] is None:
                    L.check(lib.mi_graph_begin(st), "mi_graph_begin")
                    try:
                        self._launch(pl, B, Lq, st)
                    finally:
                        g = C.c_void_p()
                        rc = lib.mi_graph_end(st, C.byref(g))
                    L.check(rc, "mi_graph_end")
                    pl["graph"] = g
                L.check(lib.mi_graph_launch(pl["graph"], st), "mi_graph_launch")
                out, mk = pl["out"].clone().reshape(B, Lq, self.cfg["d_model"]), pl["mask"].bool()
            cur.wait_stream(side)
            out.record_stream(cur)
            mk.record_stream(cur)
            return out, mk
        stage()
        self._launch(pl, B, Lq, L.current_stream())
        return pl["out"].clone().reshape(B, Lq, self.cfg["d_model"]), pl["mask"].bool()

    # ------------------------------------------------------------------ checkpoints from disk (no network)
    @classmethod
    def from_directory(cls, path: str, device="cuda"):
        """A T5 encoder from a Hugging Face checkpoint DIRECTORY -- ``config.json`` + ``model.safetensors`` (or ``pytorch_model.bin``, also
        sharded ``*.safetensors`` / ``*.bin`` with their index) -- read directly: no ``from_pretrained``, no hub / network call.  Covers the
        original checkpoints (ReLU feed-forward, tied ``shared.weight``) and the t5-v1.1 family (gated-GELU, ``wi_0`` / ``wi_1``).  Decoder
        and LM-head tensors of full T5 checkpoints are ignored.  (minimagen/t5.py:24-28 obtains the same tensors through
        ``T5EncoderModel.from_pretrained``.)"""
        import glob
        import json
        import os
        with open(os.path.join(path, "config.json")) as f:
            c = json.load(f)
        sd = {}
        files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if files:
            from safetensors.torch import load_file
            for fn in files:
                sd.update(load_file(fn, device="cpu"))
        else:
            files = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
            if not files:
                raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin under {path}")
            for fn in files:
                sd.update(torch.load(fn, map_location="cpu", weights_only=True))
        sd = {k: v for k, v in sd.items() if k.startswith("encoder.") or k == "shared.weight"}
        if "shared.weight" not in sd and "encoder.embed_tokens.weight" not in sd:
            raise KeyError(f"{path}: neither shared.weight nor encoder.embed_tokens.weight in the checkpoint")
        ff = c.get("feed_forward_proj", "relu")
        if ff not in ("relu", "gated-gelu", "gelu", "gated-relu"):
            raise NotImplementedError(f"feed_forward_proj {ff!r}")
        return cls(sd, d_model=c["d_model"], d_kv=c["d_kv"], num_heads=c["num_heads"], d_ff=c["d_ff"], num_layers=c["num_layers"],
                   relative_attention_num_buckets=c.get("relative_attention_num_buckets", 32),
                   relative_attention_max_distance=c.get("relative_attention_max_distance", 128),
                   layer_norm_epsilon=c.get("layer_norm_epsilon", 1e-6), feed_forward_proj=ff, device=device)


# ---------------------------------------------------------------------- the reference's front-end (minimagen/t5.py:24-84), offline-capable
T5_LOCAL_DIR_ENV = "MINIMAGEN_T5_DIR"        # a directory holding one Hugging Face checkpoint directory per handle ('t5-small', 'google/t5-v1_1-base', ...)


def _local_dir(handle: str):
    """Where the files of ``handle`` lie on this machine, or None: $MINIMAGEN_T5_DIR/<handle> (also with '/' -> '--' or only the last path
    component), else the Hugging Face hub cache (``snapshot_download(local_files_only=True)``: never touches the network)."""
    import os
    root = os.environ.get(T5_LOCAL_DIR_ENV)
    if root:
        for cand in (handle, handle.replace("/", "--"), handle.split("/")[-1]):
            d = os.path.join(root, cand)
            if os.path.isfile(os.path.join(d, "config.json")):
                return d
    try:
        from huggingface_hub import snapshot_download
        return snapshot_download(handle, local_files_only=True)
    except Exception:
        return None


def register_t5(name: str, *, model=None, model_dir: Optional[str] = None, tokenizer=None, dim: Optional[int] = None, device=None):
    """Install the encoder and / or tokenizer of ``name`` (an entry of T5_VERSIONS, or a new name with ``dim``) from local objects instead of
    the hub: ``model`` = a T5EncoderHIP or a transformers T5EncoderModel, ``model_dir`` = a checkpoint directory (T5EncoderHIP.from_directory),
    ``tokenizer`` = a Hugging Face tokenizer object or any callable ``(list[str], max_length) -> (input_ids [B, L] int64, attention_mask [B, L])``
    (padding to the longest caption, truncation to max_length: what minimagen/t5.py:63-69 asks the tokenizer for)."""
    if name not in T5_VERSIONS:
        if dim is None:
            raise KeyError(f"unknown T5 name {name!r}: pass dim= to register a new one")
        T5_VERSIONS[name] = {'tokenizer': None, 'model': None, 'handle': name, 'dim': dim, 'size': 0.}
    dev = device if device is not None else ("cuda" if L.backend() == "hip-gfx950" else "cpu")
    if model_dir is not None:
        model = T5EncoderHIP.from_directory(model_dir, device=dev)
    if model is not None:
        if not isinstance(model, T5EncoderHIP):
            model = T5EncoderHIP.from_hf(model.eval(), device=dev)
        if model.cfg["d_model"] != T5_VERSIONS[name]['dim']:
            raise ValueError(f"{name}: encoder width {model.cfg['d_model']} != registered dim {T5_VERSIONS[name]['dim']}")
        T5_VERSIONS[name]['model'] = model
    if tokenizer is not None:
        T5_VERSIONS[name]['tokenizer'] = tokenizer
    _embed_cache_clear(name)


def _check_downloads(name):
    """minimagen/t5.py:24-28.  Order: whatever register_t5() installed; the checkpoint directory on this machine (_local_dir: read without
    any hub call); last the reference's own route, ``from_pretrained(handle)``, which needs the network on first use."""
    ver = T5_VERSIONS[name]
    if ver['tokenizer'] is not None and ver['model'] is not None:
        return
    d = _local_dir(ver['handle'])
    dev = "cuda" if L.backend() == "hip-gfx950" else "cpu"
    if ver['tokenizer'] is None:
        from transformers import T5Tokenizer
        ver['tokenizer'] = T5Tokenizer.from_pretrained(d, local_files_only=True) if d else T5Tokenizer.from_pretrained(ver['handle'])
    if ver['model'] is None:
        if d:
            ver['model'] = T5EncoderHIP.from_directory(d, device=dev)
        else:
            from transformers import T5EncoderModel
            ver['model'] = T5EncoderHIP.from_hf(T5EncoderModel.from_pretrained(ver['handle']).eval(), device=dev)


def _tokenize(tokenizer, text, max_length):
    """minimagen/t5.py:63-69: pad to the longest caption, truncate to max_length -> (input_ids, attention_mask) int64 on the host"""
    if hasattr(tokenizer, "batch_encode_plus") or hasattr(tokenizer, "pad"):
        enc = (tokenizer.batch_encode_plus if hasattr(tokenizer, "batch_encode_plus") else tokenizer)(
            text, padding='longest', max_length=max_length, truncation=True, return_tensors="pt")
        return enc.input_ids, enc.attention_mask
    ids, mask = tokenizer(list(text), max_length)
    return torch.as_tensor(ids, dtype=torch.int64), torch.as_tensor(mask, dtype=torch.int64)


# Caption -> embedding cache (not in the reference, which re-encodes every caption of every call): the encoder is frozen, so a caption's
# rows are a pure function of (encoder, caption, max_length).  Entries hold the UNPADDED rows on the encoder's device; a call encodes only
# its misses (one batch) and assembles the padded [B, L, dim] result from the cache.  Masked keys contribute exact zeros and every other
# operation is per row, so a caption's rows do not depend on its batch beyond fp32 rounding (the block-scaled GEMM shares a power-of-two
# scale among the rows of a tile): cached and fresh results agree to ~1e-6 (tests/test_t5.py) -- so with the cache on, a caption's rows
# depend on which call first encoded it; set MINIMAGEN_T5_CACHE=0 where bit-reproducible encodings matter.  MINIMAGEN_T5_CACHE = entries
# kept per process (default 4096, 0 = off); MINIMAGEN_T5_CACHE_MB bounds the device memory the entries hold (default 256 MB: an entry is
# up to 256 x d_model fp32 values, 0.8 MB for t5_base, 4 MB for the xxl encoders).  Keys carry a per-encoder serial, not id().
import collections as _collections
import os as _os

_EMBED_CACHE = _collections.OrderedDict()
T5_CACHE_ENTRIES = int(_os.environ.get("MINIMAGEN_T5_CACHE", "4096"))
T5_CACHE_BYTES = int(float(_os.environ.get("MINIMAGEN_T5_CACHE_MB", "256")) * (1 << 20))
_cache_bytes = 0
_encoder_serial = iter(range(1, 1 << 62))


def _serial_of(model):
    s = getattr(model, "_cache_serial", None)
    if s is None:
        s = model._cache_serial = next(_encoder_serial)
    return s


def _cache_pop_oldest():
    global _cache_bytes
    _, v = _EMBED_CACHE.popitem(last=False)
    _cache_bytes -= v.numel() * v.element_size()
t5_cache_stats = {"hits": 0, "misses": 0}


def _embed_cache_clear(name=None):
    global _cache_bytes
    for k in [k for k in _EMBED_CACHE if name is None or k[0] == name]:
        v = _EMBED_CACHE.pop(k)
        _cache_bytes -= v.numel() * v.element_size()


def t5_encode_text(text, name: str = 't5_base', max_length=MAX_LENGTH, tokenizer=None):
    """minimagen/t5.py:31-84: tokenise (pad longest, truncate to max_length), encode with the HIP T5 stack, zero the masked positions,
    return (embeds [B, L, dim] fp32, bool mask [B, L]).  ``tokenizer`` overrides the registered one for this call (see register_t5)."""
    if tokenizer is None or T5_VERSIONS[name]['model'] is None:
        _check_downloads(name)
    model = T5_VERSIONS[name]['model']
    tok = tokenizer if tokenizer is not None else T5_VERSIONS[name]['tokenizer']
    text = list(text)
    input_ids, attention_mask = _tokenize(tok, text, max_length)
    B, Lq = input_ids.shape
    lens = attention_mask.sum(1).tolist()
    prefix = bool((attention_mask.bool() == (torch.arange(Lq)[None, :] < attention_mask.sum(1, keepdim=True))).all())      # right-padded, as every T5 tokenizer pads
    if T5_CACHE_ENTRIES <= 0 or tokenizer is not None or not prefix or min(lens) == 0:
        return model.encode(input_ids, attention_mask)
    global _cache_bytes
    keys = [(name, cap, max_length, _serial_of(model)) for cap in text]
    rows = [None] * B
    miss = []
    for b, k in enumerate(keys):
        hit = _EMBED_CACHE.get(k)
        if hit is not None and hit.shape[0] == lens[b]:
            _EMBED_CACHE.move_to_end(k)
            rows[b] = hit
            t5_cache_stats["hits"] += 1
        elif k not in [keys[m] for m in miss]:
            miss.append(b)
            t5_cache_stats["misses"] += 1
    if miss:
        lm = max(lens[b] for b in miss)                     # (re-pad the misses among themselves: a shorter launch)
        emb, _ = model.encode(input_ids[miss, :lm].contiguous(), attention_mask[miss, :lm].contiguous())
        for j, b in enumerate(miss):
            old = _EMBED_CACHE.pop(keys[b], None)
            if old is not None:
                _cache_bytes -= old.numel() * old.element_size()
            v = _EMBED_CACHE[keys[b]] = emb[j, :lens[b]].clone()
            _cache_bytes += v.numel() * v.element_size()
        while _EMBED_CACHE and (len(_EMBED_CACHE) > T5_CACHE_ENTRIES or _cache_bytes > T5_CACHE_BYTES):
            _cache_pop_oldest()
    out = torch.zeros(B, Lq, model.cfg["d_model"], dtype=torch.float32, device=model.dev)
    for b, k in enumerate(keys):
        r = rows[b] if rows[b] is not None else _EMBED_CACHE.get(k)
        if r is None:                                       # (evicted within this very call: capacity below the batch size)
            r = model.encode(input_ids[b:b + 1, :lens[b]].contiguous(), attention_mask[b:b + 1, :lens[b]].contiguous())[0][0]
        out[b, :lens[b]] = r
    return out, attention_mask.to(model.dev).bool()
