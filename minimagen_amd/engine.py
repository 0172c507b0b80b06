"""Launch-plan builder for ``Unet``: turns the module tree into a fixed sequence of C-ABI kernel calls.

For one (batch, resolution) the engine allocates every intermediate tensor once (a ``Workspace``),
pre-builds every parameter struct, and ``run`` just replays the list -- no allocation, no host
synchronisation, pointer-stable, so the sampler can capture it in a HIP graph.

Classifier-free guidance runs as ONE batch of 2B rows (conditional rows then null rows).  Tensors that
do not depend on the conditioning (the input image, CrossEmbed output, pre-downsample convs, the first
Block of a level) are computed once for B rows and read by both halves (``mi_act.bmod``).

The op order follows minimagen/Unet.py:355-472 and minimagen/layers.py:417-439.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch
from torch import nn

from . import _lib as L
from . import packing as P
from .layers import Attention, CrossAttention, EinopsToAndFrom, Identity, Parallel, ResnetBlock, TransformerBlock

MAX_TEXT_LEN = 256
# kernel-shape tuning knobs (A/B measurements; the defaults are what profiles/ was measured with)
CONV_SPLIT8 = int(os.environ.get("MINIMAGEN_CONV_SPLIT8", "64"))        # 8-channel outputs of images up to SPLIT8^2 pixels as two 4-channel workgroups (0 = off)
CONV_RP = int(os.environ.get("MINIMAGEN_CONV_RP", "2"))                 # 1: narrow k3 s1 convs (channels in multiples of 8, <= 64 in) on the row-paired matrix-core kernel; 2: also nearest-x2 + k3 and k4 s2
CONV_WIDE_GEMM = os.environ.get("MINIMAGEN_CONV_WIDE_GEMM", "1") != "0"   # wide k3 s1 convs (channels in multiples of 32 in / 64 out) on the GEMM kernel with prepared operand planes (conv_wide.hip)
COND_GEMM = int(os.environ.get("MINIMAGEN_COND_GEMM", "2048"))           # stacked time-MLPs with at least this many rows run as one GEMM per step (0 = always inside cond_step_kernel)
FLASH_KV_PREP = os.environ.get("MINIMAGEN_FLASH_KV_PREP", "1") != "0"     # multi-query self-attention of the wide presets: K / V prepared once per launch, LDS-DMA into the workgroups
CE_MFMA = os.environ.get("MINIMAGEN_CE_MFMA", "1") != "0"                  # CrossEmbed on the matrix cores (0: the fp32 VALU kernel)
CONV_REVERSE = int(os.environ.get("MINIMAGEN_CONV_REVERSE", "1"))        # a row-paired conv walks the image groups opposite to its producer (0 = off)
RP_NTILE = int(os.environ.get("MINIMAGEN_RP_NTILE", "0"))               # tiles per workgroup of the row-paired kernel (0 = the library's choice)
RP_NTILE_BY = {k: int(os.environ.get("MINIMAGEN_RP_NTILE_" + k, "0")) for k in ("L", "M", "S")}     # ... per image-size class (> 128^2 / > 64^2 / smaller)
# ... and for workspaces of PIPELINED calls (Imagen.sample(_async=True) with two call lanes in flight): longer strips = fewer, longer workgroups
# per launch leave the other lane's kernels room on the CUs.  Measured on one box, back to back (profiles/r05_summary.md): two tiles per workgroup
# at <= 64^2 and four at 128^2 give 44.1 K against 42.3 K denoising-steps/s with two lanes -- and 31.0 K against 33.5 K for synchronous calls, which
# therefore keep the library's choice.  Speed only: the per-tile statistics (hence every bit of the result) do not depend on the strip length.
RP_NTILE_PIPE = {k: int(os.environ.get("MINIMAGEN_RP_NTILE_PIPE_" + k, d)) for k, d in (("L", "0"), ("M", "4"), ("S", "2"))}
# round 6: the narrow k3 s1 convs on 32 / 64 / 128 / 256-wide images as full-width row stripes with specialised loader / MFMA waves
# (csrc/conv_stripe.hip, tile_cfg 12; same bits per output as the tile kernel, statistics per block of 8 (4) rows).  0 = the tile kernel
# everywhere; a string of size classes ("LMS" = all, the default) selects where: L > 128^2, M > 64^2, S smaller
CONV_STRIPE = os.environ.get("MINIMAGEN_CONV_STRIPE", "LMS").upper().replace("0", "").replace("1", "LMS")
ST_NBLK = {k: int(os.environ.get("MINIMAGEN_ST_NBLK_" + k, "0")) for k in ("L", "M", "S")}              # statistics blocks per workgroup (0 = the library's choice; speed only)
# ... for workspaces of PIPELINED calls: four blocks (half an image) per workgroup at <= 64^2 while that leaves >= 128 workgroups -- fewer, longer
# workgroups leave the other lane's kernels room, as with RP_NTILE_PIPE: 46.7 K against 45.8 K steps/s, same box back to back; L / M: no effect
# (profiles/r06_summary.md)
ST_NBLK_PIPE = {k: int(os.environ.get("MINIMAGEN_ST_NBLK_PIPE_" + k, d)) for k, d in (("L", "0"), ("M", "0"), ("S", "4"))}
WEIGHT_FINGERPRINT = os.environ.get("MINIMAGEN_WEIGHT_FINGERPRINT", "1") != "0"   # content fingerprint of the weights at every public call (see pack())
JT = 17     # context tiles of 16 rows: 1 null + (2|4) time tokens + 256 text rows <= 272


class Act:
    """An NCHW activation (fp32, or bf16 in the reduced-precision configuration) plus the per-channel partial statistics its producer
    emitted (always fp64)."""
    __slots__ = ("t", "stats", "nt", "C", "H", "W", "batch", "uses", "rev")

    def __init__(self, t, stats, nt, C_, H, W, batch):
        self.t, self.stats, self.nt, self.C, self.H, self.W, self.batch = t, stats, nt, C_, H, W, batch
        self.uses = 0
        self.rev = False            # the producing launch walked the image groups in reverse order (conv(): the consumer goes the other way)

    @property
    def st(self) -> int:
        return 1 if self.t.dtype == torch.bfloat16 else 0

    def c(self, consumer_batch: int, scale: float = 1.0) -> L.MiAct:
        bmod = self.batch if self.batch != consumer_batch else 0
        self.uses += 1
        return L.MiAct(L.ptr(self.t), self.C, L.ptr(self.stats), self.nt, scale, bmod, self.st)


class _Store16Unsupported(Exception):
    """raised while a launch plan is built with bf16 activation storage and a layer needs a kernel that only reads fp32"""


def _lin(m: Optional[nn.Linear]) -> L.MiLinear:
    if m is None:
        return L.MiLinear(0, 0, 0, 0)
    return L.MiLinear(L.ptr(m.weight), L.ptr(m.bias) if m.bias is not None else 0, m.in_features, m.out_features)


class Workspace:
    pass


def _close_workspace(ws):
    """Destroy the HIP graph execs cached on a workspace's sampler state (Imagen._p_sample_loop): they are raw handles, not tensors."""
    lib = L.lib()
    for st in getattr(ws, "sampler_state", {}).values():
        for entry in getattr(st, "graphs", {}).values():
            if entry.get("graph") is not None:
                # sample() never host-syncs its stage streams: replays of this exec (and the workspace buffers it addresses, dropped
                # together with it) may still be queued -- drain the device first (invalidations are rare: new weights / .to())
                if L.backend() == "hip-gfx950" and torch.cuda.is_available():
                    torch.cuda.synchronize(ws.dev)
                lib.mi_graph_destroy(entry["graph"])
                entry["graph"] = None
        if hasattr(st, "graphs"):
            st.graphs.clear()


class UnetEngine:
    def __init__(self, unet):
        self.unet = unet
        self._pack = None
        self._pack_key = None
        self._pack_fp = None
        self._ws = {}
        # "fp32": every contraction fp32-grade (3-term fp16 splits on the matrix cores, fp32 VALU elsewhere);
        # "half": the matrix-core contractions use a single fp16 term (fp32 accumulate / softmax / statistics / storage) --
        #         the reduced-precision BASELINE configurations; parity gate 3e-2 / 3e-3 instead of 1e-4 / 1e-5
        self.precision = os.environ.get("MINIMAGEN_PRECISION", "fp32")

    # ------------------------------------------------------------------ weights
    def _param_key(self):
        """Identity of the weights the packed copies were derived from: device, every tensor's storage pointer and version counter."""
        ts = self._tensors()
        return (str(ts[0].device), tuple(t.data_ptr() for t in ts), sum(t._version for t in ts))

    def _tensors(self):
        """every parameter and buffer of the U-Net (the module tree is walked once; nn.Module._apply / load_state_dict re-home tensors
        through Unet._apply -> invalidate(), which drops this list)"""
        ts = self.__dict__.get("_tensor_list")
        if ts is None:
            ts = self._tensor_list = list(self.unet.parameters()) + list(self.unet.buffers())
        return ts

    def _fingerprint(self):
        """Cheap content fingerprint (L1 and L2 norm of every parameter, two fused multi-tensor launches): in-place updates through
        ``p.data`` (EMA, ``.data.copy_``) do not bump ``p._version``, so identity alone would leave the packed copies stale."""
        ts = [t.detach() for t in self._tensors() if t.is_floating_point()]
        return torch.stack(torch._foreach_norm(ts, 1) + torch._foreach_norm(ts, 2))

    def invalidate(self):
        """Drop every packed / folded weight copy, workspace and captured step graph (they are rebuilt on the next call).  Called
        automatically when the weights' identity or content fingerprint changes, by ``Unet.load_state_dict`` and ``Unet._apply``."""
        for ws in self._ws.values():
            _close_workspace(ws)
        self._ws = {}
        self._pack = self._pack_key = self._pack_fp = None
        self.__dict__.pop("_tensor_list", None)

    def _check_params(self):
        for name, t in list(self.unet.named_parameters()) + list(self.unet.named_buffers()):
            if t.dtype != torch.float32:
                raise L.MinImagenHipError(f"parameter {name} is {t.dtype}; the HIP path computes in fp32 like the reference")
            if not t.is_contiguous():
                raise L.MinImagenHipError(f"parameter {name} is not contiguous")
        L.require_device(next(self.unet.parameters()))

    def pack(self):
        """Validate / rebuild the packed weights.  Identity (storage pointer + version counter of every parameter) is a host-only check;
        the content fingerprint catches ``p.data`` updates and costs two fused multi-tensor launches plus ONE small device -> host
        comparison ON THE CALLER'S STREAM (never on a stage stream: a sampling call in flight is not waited for).  Once per public call;
        MINIMAGEN_WEIGHT_FINGERPRINT=0 trusts identity alone (no device work, no synchronisation at all).  Imagen.sample() uses the split
        form pack_begin() / pack_changed() instead, which takes the host wait off the front of the call."""
        key = self._param_key()
        if not WEIGHT_FINGERPRINT:
            if self._pack is not None and key == self._pack_key:
                return self._pack
            fp = None
        else:
            fp = self._fingerprint()
            if self._pack is not None and key == self._pack_key and torch.equal(fp, self._pack_fp):
                return self._pack
        return self._repack(key, fp)

    def pack_identity(self):
        """Host-only half of the validation (storage pointers + version counters: what every optimiser step, load_state_dict and .to()
        change): re-packs right here when it fails.  Imagen.sample() calls it up front and runs the content fingerprint (pack_begin /
        pack_changed) BEHIND the work it enqueues: with the fingerprint in front, a synchronous call kept the host behind the whole
        previous call (one device -> host comparison on the caller's stream) and the GPU idle for the host's ~2 ms of list walking and
        launch time (profiles/r05_sync_call_gaps.txt)."""
        if self._pack is None or self._param_key() != self._pack_key:
            self.pack()

    def pack_begin(self):
        """Launch the content fingerprint; its comparison with the packed state goes to pinned host memory WITHOUT a host wait.  Returns a
        token for pack_changed(), or None when there is nothing to compare (MINIMAGEN_WEIGHT_FINGERPRINT=0).  Call after pack_identity()."""
        # (the module tree is walked HERE, where the host has slack: a parameter object replaced since the list was cached -- module.weight =
        # nn.Parameter(...) -- counts as a change)
        fresh = list(self.unet.parameters()) + list(self.unet.buffers())
        cached = self._tensors()
        if len(fresh) != len(cached) or any(a is not b for a, b in zip(fresh, cached)):
            self.__dict__.pop("_tensor_list", None)
            return (None, None, None)
        if not WEIGHT_FINGERPRINT or self._pack is None or self._pack_fp is None:
            return None
        fp = self._fingerprint()
        flag = self.__dict__.get("_fp_flag")
        if flag is None or flag.device != torch.device("cpu"):
            flag = self._fp_flag = torch.zeros(1, dtype=torch.int32).pin_memory() if fp.is_cuda else torch.zeros(1, dtype=torch.int32)
        flag.copy_((fp != self._pack_fp).any().to(torch.int32).reshape(1), non_blocking=True)
        ev = torch.cuda.current_stream(fp.device).record_event() if fp.is_cuda else None
        return (flag, ev, fp)

    def pack_changed(self, token) -> bool:
        """Second half: True when the weights' VALUES differ from the packed copies (``p.data`` updates behind the version counter).  The
        caller must then discard what it enqueued, let the device drain, and call pack() (Imagen.sample does)."""
        if token is None:
            return False
        flag, ev, _ = token
        if flag is None:
            return True
        if ev is not None:
            ev.synchronize()
        return bool(flag.item())

    def _repack(self, key, fp):
        self.invalidate()
        self._pack_fp = fp
        self._check_params()
        u = self.unet
        dev = next(u.parameters()).device
        lib = L.lib()
        pk = Workspace()
        pk.keep = []           # keeps packed tensors alive
        pk.freq = P.sinusoid_freq(u.dim, dev)
        pk.conv = {}
        pk.conv_rp = {}
        pk.conv_ig = {}
        pk.attn = {}
        pk.attn_exp = {}

        def conv_pack(mod: nn.Conv2d, w=None, b=None):
            w = mod.weight if w is None else w
            ct = lib.mi_conv_cout_tile(w.shape[0])
            wp = P.pack_conv_weight(w.to(dev), ct)
            pk.keep.append(wp)
            if w.shape[-1] in (3, 4) and w.shape[1] % 8 == 0:
                pk.conv_rp[id(wp)] = P.pack_conv_weight_rp(w.to(dev))
            if CONV_WIDE_GEMM and w.shape[-1] == 3 and w.shape[1] % 32 == 0 and w.shape[0] % 64 == 0:
                pk.conv_ig[id(wp)] = P.pack_conv_weight_ig(w.to(dev))
            return wp

        resblocks: List[ResnetBlock] = [m for m in u.modules() if isinstance(m, ResnetBlock)]
        # every time_mlp stacked into one [R][tcd] matrix (layers.py:395-399)
        rows, biases, off = [], [], 0
        pk.ss_off = {}
        for rb in resblocks:
            if rb.time_mlp is not None:
                lin = rb.time_mlp[1]
                pk.ss_off[id(rb)] = off
                rows.append(lin.weight.detach())
                biases.append(lin.bias.detach())
                off += lin.out_features
        pk.R = off
        pk.tm_w = torch.cat(rows, 0).contiguous() if rows else torch.zeros(1, u.time_cond_dim, device=dev)
        pk.tm_b = torch.cat(biases, 0).contiguous() if rows else torch.zeros(1, device=dev)
        for rb in resblocks:
            pk.conv[id(rb.block1.project)] = conv_pack(rb.block1.project)
            pk.conv[id(rb.block2.project)] = conv_pack(rb.block2.project)
            if isinstance(rb.res_conv, nn.Conv2d):
                ct = lib.mi_conv_cout_tile(rb.res_conv.weight.shape[0])
                rw = P.pack_conv_weight(rb.res_conv.weight, ct).reshape(rb.res_conv.weight.shape[1], -1).contiguous()
                pk.keep.append(rw)
                if rb.res_conv.weight.shape[1] % 8 == 0:
                    pk.conv_rp[id(rw)] = P.pack_conv_weight_rp(rb.res_conv.weight)
                if CONV_WIDE_GEMM and rb.res_conv.weight.shape[1] % 32 == 0 and rb.res_conv.weight.shape[0] % 64 == 0:
                    pk.conv_ig[id(rw)] = P.pack_conv_weight_ig(rb.res_conv.weight)
                pk.conv[id(rb.res_conv)] = rw
            if rb.cross_attn is not None:
                ca: CrossAttention = rb.cross_attn.fn
                Cc = ca.to_q.in_features
                if ca.dim_head != 64:
                    raise NotImplementedError("cross-attention with dim_head != 64")
                if Cc not in (8, 16, 32):
                    if Cc % 16 or u.cond_dim % 16:
                        raise NotImplementedError(f"cross-attention over {Cc} channels / {u.cond_dim}-wide context: the wide path needs multiples of 16")
                    continue                 # wide path (unfolded, attention_wide.hip): works on the module's own weights
                if not isinstance(ca.norm_context, Identity):
                    raise NotImplementedError("norm_context=True cross-attention is not on the MinImagen hot path")
                mg, mv, g0, v0 = P.fold_cross_attention(ca.to_q.weight, ca.to_kv.weight, ca.to_out[0].weight, ca.null_kv, ca.heads, ca.dim_head)
                pk.attn[id(ca)] = (mg, mv, g0, v0)
                # power-of-two operand scalings of the fp16x3 kernel from magnitude bounds of its inputs (both are LayerNorm outputs)
                pk.attn_exp[id(ca)] = P.attn_f16_exponents(mg, mv, g0, v0, cmax=P.layernorm_bound(u.norm_cond.weight, u.norm_cond.bias, u.cond_dim),
                                                           xmax=P.layernorm_bound(ca.norm.gamma, ca.norm.beta, Cc))
        # K10: multi-query self-attention (one shared 64-wide k/v head, layers.py:42) folded like K9 with the k/v rows repeated per head
        for m in u.modules():
            if isinstance(m, Attention):
                Cc = m.to_q.in_features
                dh = m.to_kv.out_features // 2
                if dh != 64:
                    raise NotImplementedError("self-attention with dim_head != 64")
                if Cc not in (8, 16, 32):
                    if Cc % 16:
                        raise NotImplementedError(f"self-attention over {Cc} channels: the wide path needs a multiple of 16")
                    continue
                kv = m.to_kv.weight.detach()
                kv_full = torch.cat((kv[:dh].repeat(m.heads, 1), kv[dh:].repeat(m.heads, 1)), 0)
                pk.attn[id(m)] = P.fold_cross_attention(m.to_q.weight, kv_full, m.to_out[0].weight, m.null_kv, m.heads, dh)
        # plain convs: down/up-sample, final
        for level in u.downs:
            pre, _, _, _, post = level
            if isinstance(pre, nn.Conv2d):
                pk.conv[id(pre)] = conv_pack(pre)
            if isinstance(post, nn.Conv2d):
                pk.conv[id(post)] = conv_pack(post)
            elif isinstance(post, Parallel):
                w, b = P.fold_parallel_1x1(post.fns[0].weight, post.fns[0].bias, post.fns[1].weight, post.fns[1].bias)
                b = b.contiguous()
                pk.keep.append(b)
                pk.conv[id(post)] = (conv_pack(None, w=w), b)
        for level in u.ups:
            up = level[3]
            if isinstance(up, nn.Sequential):
                pk.conv[id(up[1])] = conv_pack(up[1])
        pk.conv[id(u.final_conv)] = conv_pack(u.final_conv)
        pk.ce_w = [c.weight.detach().permute(1, 2, 3, 0).contiguous() for c in u.init_conv.convs]
        # matrix-core CrossEmbed (the BASELINE configuration: dim_scales (4, 2, 2), kernels (3, 7, 15), <= 4 image channels): one Toeplitz
        # table per input half (x / low-res conditioning image)
        pk.ce_mfma = {}
        cvs = u.init_conv.convs
        if CE_MFMA and [c.kernel_size[0] for c in cvs] == [3, 7, 15] and [c.out_channels for c in cvs] == [4, 2, 2] and u.channels <= 4:
            for chan0 in range(0, cvs[0].in_channels, u.channels):
                pk.ce_mfma[chan0] = P.pack_crossembed_mfma([c.weight for c in cvs], chan0, u.channels)
        self._pack, self._pack_key = pk, key
        return pk

    # ------------------------------------------------------------------ workspace / program
    def packed(self):
        """The packed weights as validated by the last ``pack()`` (one validation -- a device-side fingerprint compare, i.e. a host
        synchronisation -- per public call, not one per internal step: in the pipelined sampler a sync on a stage's stream would stall the
        host behind that stage's queued work)."""
        return self._pack if self._pack is not None else self.pack()

    def workspace(self, B: int, B2: int, H: int, W: int, precision: Optional[str] = None, has_text: bool = True, lane: int = 0,
                  pipelined: bool = False) -> Workspace:
        """``lane``: independent workspaces (buffers, step tables, captured graphs) of one shape, so that two sample() calls can be in
        flight side by side (Imagen.sample(_async=True) alternates lanes); ``pipelined``: the launch plan of calls that share the GPU with
        another lane (strip lengths RP_NTILE_PIPE; same results, another workspace)"""
        pk = self.packed()
        dev = next(self.unet.parameters()).device
        precision = self.precision if precision is None else precision
        assert precision in ("fp32", "half"), precision
        key = (B, B2, H, W, str(dev), precision, has_text) + ((lane,) if lane else ()) + (("pipe",) if pipelined else ())
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        u = self.unet
        ws = Workspace()
        ws.B, ws.B2, ws.H, ws.W, ws.dev = B, B2, H, W, dev
        ws.pipelined = pipelined
        ws.half = precision == "half"
        f = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        ws.x = f(B, u.channels, H, W)
        ws.lowres = f(B, u.channels, H, W) if u.lowres_cond else None
        ws.times = torch.zeros(B, dtype=torch.int64, device=dev)
        ws.lowres_times = torch.zeros(B, dtype=torch.int64, device=dev) if u.lowres_cond else None
        ws.keep = torch.ones(B2, dtype=torch.uint8, device=dev)
        ws.c_text = f(B2, MAX_TEXT_LEN, u.cond_dim)
        ws.text_hiddens = f(B2, u.time_cond_dim)
        ws.ntot = u.num_time_tokens * (2 if u.lowres_cond else 1)
        ws.has_text = has_text                       # Unet.py:572: text is optional -- without it the context is the null row + the time tokens
        ws.J = 1 + ws.ntot + (MAX_TEXT_LEN if has_text else 0)
        ws.JT = -(-ws.J // 16)
        ws.ss = f(B2, max(pk.R, 1))
        ws.c_time = f(B2, ws.ntot, u.cond_dim)
        ws.t_out = f(B2, u.time_cond_dim)
        ws.gv = {}
        ws.tensors = []
        ws.text_L = None
        # reduced-precision configuration: activations in bf16 (BASELINE configs 3-5) when every layer runs on a kernel that reads them
        # (row-paired convs, matrix-core CrossEmbed, fp16 cross-attention: the BASELINE U-Nets); otherwise fp32 storage
        for store16 in ((True, False) if ws.half else (False,)):
            ws.store16 = store16
            ws.gv, ws.tensors, ws.prog, ws.prog_text, ws.wide_attn = {}, [], [], [], False     # (prog_text: the wide cross-attentions' text keys / values)
            ws.ig_convs = []                 # wide GEMM convs: their operand planes share ONE buffer (the launches of a workspace are ordered)
            try:
                self._build_program(ws, pk)
                break
            except _Store16Unsupported:
                continue
        if ws.ig_convs:
            nbytes = max(n for _, n in ws.ig_convs)
            ws.ig_prep = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
            for p_, _ in ws.ig_convs:
                p_.act_prep, p_.act_prep_bytes = L.ptr(ws.ig_prep), nbytes
        self._ws[key] = ws
        return ws

    def _tile_cfg(self, H, W, batch=None, cz=1):
        """Tile shape by image size ONLY: the per-tile partial statistics must reduce in the same order whatever the
        batch is, so that a sharded batch reproduces the unsharded rows bit for bit."""
        lib = L.lib()
        best = 0 if (W >= 64 and H * W > 64 * 64) else 2
        th, tw = C.c_int(), C.c_int()
        lib.mi_conv_tile_shape(best, C.byref(th), C.byref(tw))
        return best, -(-H // th.value) * -(-W // tw.value)

    def _stripe_rows(self, batch, Ho, Wo, in0, in1, Cout, gn, res, up2=0, ksize=3, stride=1) -> int:
        """rows per statistics block if csrc/conv_stripe.hip (tile_cfg 12) takes this narrow fp32 k3 s1 (or k4 s2) launch, else 0: asked of the library
        (mi_conv_stripe_rows) on a probe of the launch's shape -- pointers only need to be non-null where the real launch has them"""
        q = L.MiConvParams()
        q.B, q.H, q.W, q.Cout, q.ksize, q.stride, q.up2 = batch, Ho, Wo, Cout, ksize, stride, int(bool(up2))
        q.in0 = L.MiAct(1, in0.C, 1 if in0.stats is not None else 0, in0.nt, 1.0, 0)
        if in1 is not None:
            q.in1 = L.MiAct(1, in1.C, 1 if in1.stats is not None else 0, in1.nt, 1.0, 0)
        q.w_rp = 1
        if gn is not None:
            q.gn_groups = gn.num_groups
        if res is not None:
            r0, r1, rw, _ = res
            q.res0 = L.MiAct(1, r0.C, 1 if r0.stats is not None else 0, r0.nt, 1.0, 0)
            if r1 is not None:
                q.res1 = L.MiAct(1, r1.C, 1 if r1.stats is not None else 0, r1.nt, 1.0, 0)
            if rw is not None:
                q.res_w, q.res_w_rp = 1, 1
        return int(L.lib().mi_conv_stripe_rows(C.byref(q)))

    def _new_act(self, ws, batch, Cc, H, W, nt, fp32: bool = False) -> Act:
        t = torch.empty(batch, Cc, H, W, dtype=torch.float32 if (fp32 or not ws.store16) else torch.bfloat16, device=ws.dev)
        st = torch.zeros(batch, Cc, max(nt, 1), 2, dtype=torch.float64, device=ws.dev) if nt else None      # fp64 partial (sum, sumsq): include/minimagen_hip.h
        ws.tensors += [t, st]
        return Act(t, st, nt, Cc, H, W, batch)

    def _emit_conv(self, ws, pk, in0: Act, in1: Optional[Act], *, wpack, bias, Cout, ksize=3, stride=1, up2=0,
                   gn: Optional[nn.GroupNorm] = None, ss_off: Optional[int] = None, res=None, conditioned=False,
                   want_stats=True, skip_scale=1.0, out_fp32=False) -> Act:
        lib = L.lib()
        Ho = in0.H * 2 if up2 else in0.H // stride
        Wo = in0.W * 2 if up2 else in0.W // stride
        if stride == 2 and ((in0.H | in0.W) & 1):
            raise NotImplementedError("stride-2 conv on an odd-sized image")
        ins = [a for a in (in0, in1) if a is not None] + ([r for r in res[:2] if r is not None] if res else [])
        batch = ws.B2 if (conditioned or any(a.batch == ws.B2 for a in ins)) else ws.B
        ct = lib.mi_conv_cout_tile(Cout)
        cfg, nt = self._tile_cfg(Ho, Wo, batch, -(-Cout // ct))
        cin_tot = in0.C + (in1.C if in1 is not None else 0)
        # row-paired matrix-core path (conv_rp.hip): every conv whose channel counts come in octets.  Narrow layers (<= 64 in, <= 32 out: the
        # BASELINE U-Nets) run the tuned single-kernel form; wider ones (Unet() default, Base, Super) the wide regime: output channels
        # tiled over the grid, GroupNorm affine + operand exponents from a small per-image launch (mi_gn_coef_fwd) ahead of the conv
        cres = 0 if (res is None or res[2] is None) else res[0].C + (res[1].C if res[1] is not None else 0)
        rp = bool(CONV_RP) and ((ksize == 3 and stride == 1) or (ksize == 4 and stride == 2 and not up2 and CONV_RP >= 2)) \
            and (not up2 or CONV_RP >= 2) and Wo % 4 == 0 and id(wpack) in pk.conv_rp \
            and in0.C % 8 == 0 and (in1 is None or in1.C % 8 == 0) \
            and (res is None or res[2] is None or (id(res[2]) in pk.conv_rp and res[0].C % 8 == 0 and (res[1] is None or res[1].C % 8 == 0)))
        narrow = cin_tot <= 64 and cres <= 64 and Cout <= (16 if stride == 2 else (8 if up2 else 32))
        wide = rp and not narrow
        stripe = False
        if rp:
            cfg = 6                      # 8 x 64 tiles
            if Ho * Wo <= 64 * 64 and (Wo <= 32 or Cout <= 8):
                cfg = 7                  # measured: 8x32 tiles for 8-channel layers at <= 64^2 (twice the workgroups) and for images no
                                         # wider than a tile; 16 output channels at 64^2 stay on 8x64 (B fragments staged per workgroup)
            if Cout > 16:
                cfg = 7                  # four N tiles are only instantiated for the 8x32 tile
            if up2:
                cfg = 6                  # the up- / down-sampling members have one tile shape each
            if stride == 2 or wide:
                cfg = 7
            if wide and stride == 1 and not up2 and Wo % 64 == 0 and Cout >= 32 and not ws.half:
                cfg = 6                  # 8 x 64 tiles for the wide k3 s1 convs
            if wide and stride == 1 and not up2 and Wo <= 16 and Ho > 8 and Cout >= 32 and not ws.half:
                cfg = 10                 # 16 x 16 tiles for images no wider than 16
            gemm = wide and ksize == 3 and stride == 1 and (not up2 or res is None) and not ws.half and cin_tot % 32 == 0 and cres % 32 == 0 and Cout % 64 == 0 \
                and id(wpack) in pk.conv_ig and (cres == 0 or id(res[2]) in pk.conv_ig)
            if gemm:
                cfg = 11                 # the wide GEMM kernel (conv_wide.hip): 8 x 16 pixels x 128 (or 64) channels per workgroup
            th, tw = {6: (8, 64), 7: (8, 32), 10: (16, 16), 11: (8, 16)}[cfg]
            nt = -(-Ho // th) * -(-Wo // tw)
            cls = "L" if Ho * Wo > 128 * 128 else ("M" if Ho * Wo > 64 * 64 else "S")
            if narrow and ((ksize == 3 and stride == 1) or (ksize == 4 and stride == 2)) and not ws.half and cls in CONV_STRIPE:
                rows = self._stripe_rows(batch, Ho, Wo, in0, in1, Cout, gn, res, up2, ksize, stride)
                if rows:
                    cfg, nt, stripe = 12, Ho // rows, True
        if ws.store16 and (not rp or wide):
            raise _Store16Unsupported()          # only the narrow row-paired kernels read bf16 activations
        out = self._new_act(ws, batch, Cout, Ho, Wo, nt if want_stats else 0, fp32=out_fp32)
        p = L.MiConvParams()
        p.B, p.H, p.W = batch, Ho, Wo
        p.in0 = in0.c(batch)
        if in1 is not None:
            p.in1 = in1.c(batch, skip_scale)
        p.Cout, p.ksize, p.stride, p.up2 = Cout, ksize, stride, up2
        p.w, p.bias = L.ptr(wpack), L.ptr(bias)
        if gn is not None:
            p.gn_groups, p.gn_gamma, p.gn_beta, p.gn_eps = gn.num_groups, L.ptr(gn.weight), L.ptr(gn.bias), gn.eps
            if ss_off is not None:
                p.scale_shift, p.ss_stride, p.ss_off = L.ptr(ws.ss), ws.ss.shape[1], ss_off
        if res is not None:
            r0, r1, rw, rb = res
            p.res0 = r0.c(batch)
            if r1 is not None:
                p.res1 = r1.c(batch, skip_scale)
            p.res_w, p.res_b = L.ptr(rw), L.ptr(rb)
        p.out_st = out.st
        p.out, p.out_stats, p.tile_cfg = L.ptr(out.t), L.ptr(out.stats), cfg | 0x100 | (0x400 if ws.half else 0) | (0x800 if (CONV_SPLIT8 and Ho * Wo <= CONV_SPLIT8 * CONV_SPLIT8) else 0)
        if wide:
            coef = torch.zeros(batch, cin_tot, 4, dtype=torch.float32, device=ws.dev)
            exps = torch.zeros(batch, 2, dtype=torch.int32, device=ws.dev)
            ws.tensors += [coef, exps]
            p.gn_coef, p.gn_exps = L.ptr(coef), L.ptr(exps)
        if rp:
            # the consumer walks the image groups in the opposite order of its producer: what was written last (still in the memory-side
            # cache / L2) is read first (speed only; measured -1.2 % on the SR step)
            out.rev = bool(CONV_REVERSE) and batch % 8 == 0 and not in0.rev
            if out.rev:
                p.tile_cfg |= 0x200
            cls = "L" if Ho * Wo > 128 * 128 else ("M" if Ho * Wo > 64 * 64 else "S")
            nt_knob = RP_NTILE_BY[cls] or RP_NTILE
            if not nt_knob and ws.pipelined and RP_NTILE_PIPE[cls] and batch * nt // RP_NTILE_PIPE[cls] >= 256:
                nt_knob = RP_NTILE_PIPE[cls]        # (only while the launch still has a workgroup per CU: smaller batches keep the library's choice -- config 3 at B = 16: 39.1 K with, 40.7 K without)
            if stripe:                              # statistics blocks per workgroup of the stripe kernel (must divide the blocks of an image)
                nt_knob = ST_NBLK[cls]
                if ws.pipelined and ST_NBLK_PIPE[cls] and nt % ST_NBLK_PIPE[cls] == 0 and batch * nt // ST_NBLK_PIPE[cls] >= 128 \
                        and self.unet.lowres_cond:       # (super-resolution U-Nets only: on the base U-Net it costs the base stage 7 %: 97.1 against 104.8 K steps/s, and gives the cascade nothing)
                    nt_knob = ST_NBLK_PIPE[cls]
                if nt_knob and (nt % nt_knob or nt_knob > 15):
                    nt_knob = 0
            p.tile_cfg |= (nt_knob & 0xf) << 12
            frags = pk.conv_ig if gemm else pk.conv_rp
            frag, p.w_rp_exp = frags[id(wpack)]
            p.w_rp = L.ptr(frag)
            if res is not None and res[2] is not None:
                rfrag, p.res_w_rp_exp = frags[id(res[2])]
                p.res_w_rp = L.ptr(rfrag)
        if wide:
            ws.prog.append((lib.mi_gn_coef_fwd, p, "gn_coef"))
            if gemm:
                ws.ig_convs.append((p, lib.mi_conv_prep_bytes(batch, cin_tot, cres, Ho, Wo)))
                ws.prog.append((lib.mi_conv_prep_fwd, p, "conv_prep"))
        ws.prog.append((lib.mi_conv_fwd, p, "conv"))
        return out

    def _emit_resnet(self, ws, pk, rb: ResnetBlock, in0: Act, in1: Optional[Act]) -> Act:
        """layers.py:417-439"""
        u = self.unet
        s = u.skip_connect_scale
        Cout = rb.block1.project.out_channels
        has_cross = rb.cross_attn is not None
        h = self._emit_conv(ws, pk, in0, in1, wpack=pk.conv[id(rb.block1.project)], bias=rb.block1.project.bias, Cout=Cout,
                            gn=rb.block1.groupnorm, want_stats=not has_cross, skip_scale=s)
        if has_cross:
            h = self._emit_cross_attn(ws, pk, rb.cross_attn.fn, h)
        ss_off = pk.ss_off.get(id(rb))
        if isinstance(rb.res_conv, nn.Conv2d):
            res = (in0, in1, pk.conv[id(rb.res_conv)], rb.res_conv.bias)
        else:
            assert in1 is None
            res = (in0, None, None, None)
        return self._emit_conv(ws, pk, h, None, wpack=pk.conv[id(rb.block2.project)], bias=rb.block2.project.bias, Cout=Cout,
                               gn=rb.block2.groupnorm, ss_off=ss_off, res=res, conditioned=ss_off is not None, skip_scale=s)

    # ------------------------------------------------------------------ wide-channel attention (C > 32): token-major chains
    @staticmethod
    def _call(fn, name, *args):
        """program entry for a positional-argument C-ABI function (the struct-taking ones are (fn, params, name))"""
        return (lambda _p, st, fn=fn, args=args: fn(*args, st)), None, name

    def _buf(self, ws, *shape):
        t = torch.empty(*shape, dtype=torch.float32, device=ws.dev)
        ws.tensors.append(t)
        return t

    def _emit_tokens_out(self, ws, tokens, batch, Cc, H, W, ln, res: Act, want_stats: bool) -> Act:
        if ws.store16:
            raise _Store16Unsupported()
        lib = L.lib()
        HW = H * W
        out = self._new_act(ws, batch, Cc, H, W, -(-HW // 64) if want_stats else 0)
        p = L.MiTokensToNchwParams()
        p.B, p.HW, p.C, p.tokens = batch, HW, Cc, L.ptr(tokens)
        if ln is not None:
            p.gamma, p.beta, p.eps = L.ptr(ln.gamma), L.ptr(ln.beta), 1e-5
        p.res = res.c(batch)
        p.out, p.out_stats = L.ptr(out.t), L.ptr(out.stats)
        ws.prog.append((lib.mi_tokens_to_nchw_fwd, p, "tokens_to_nchw"))
        return out

    def _emit_cross_attn_wide(self, ws, ca: CrossAttention, h: Act) -> Act:
        """layers.py:220-251 unfolded: LN(x) -> to_q | context -> to_kv -> flash attention over [null | time tokens | text tokens] -> to_out.0
        -> to_out.1 LayerNorm + residual.  The text rows' keys / values are step-invariant: projected once per sample() (ws.prog_text)."""
        if ws.store16:
            raise _Store16Unsupported()
        lib, u = L.lib(), self.unet
        Cc, HW, B2, inner = h.C, h.H * h.W, ws.B2, ca.heads * 64
        xh, q, o, t = self._buf(ws, B2, HW, Cc), self._buf(ws, B2, HW, inner), self._buf(ws, B2, HW, inner), self._buf(ws, B2, HW, Cc)
        kv_time = self._buf(ws, B2, ws.ntot, 2 * inner)
        kv_text = self._buf(ws, B2, MAX_TEXT_LEN, 2 * inner) if ws.has_text else None
        xa = h.c(B2)
        ws.tensors.append(xa)
        ws.prog.append(self._call(lib.mi_ln_tokens_fwd, "ln_tokens", C.byref(xa), B2, HW, L.ptr(ca.norm.gamma), L.ptr(ca.norm.beta), L.ptr(xh)))
        ws.prog.append(self._call(lib.mi_gemm_f32, "gemm_q", L.ptr(xh), L.ptr(ca.to_q.weight), 0, 0, L.ptr(q), B2 * HW, inner, Cc, 0))
        ws.prog.append(self._call(lib.mi_gemm_f32, "gemm_kv_time", L.ptr(ws.c_time), L.ptr(ca.to_kv.weight), 0, 0, L.ptr(kv_time), B2 * ws.ntot, 2 * inner, u.cond_dim, 0))
        if ws.has_text:
            ws.prog_text.append(self._call(lib.mi_gemm_f32, "gemm_kv_text", L.ptr(ws.c_text), L.ptr(ca.to_kv.weight), 0, 0, L.ptr(kv_text), B2 * MAX_TEXT_LEN, 2 * inner, u.cond_dim, 0))
        p = L.MiFlashAttnParams()
        p.B, p.HW, p.heads, p.kv_heads, p.q, p.q_scale = B2, HW, ca.heads, ca.heads, L.ptr(q), 64 ** -0.5 * P.LOG2E
        p.null_k, p.null_v = L.ptr(ca.null_kv), L.ptr(ca.null_kv) + 4 * 64
        p.k0, p.v0, p.n0, p.ld0, p.bs0 = L.ptr(kv_time), L.ptr(kv_time) + 4 * inner, ws.ntot, 2 * inner, ws.ntot * 2 * inner
        if ws.has_text:
            p.k1, p.v1, p.n1, p.ld1, p.bs1 = L.ptr(kv_text), L.ptr(kv_text) + 4 * inner, MAX_TEXT_LEN, 2 * inner, MAX_TEXT_LEN * 2 * inner
        p.out = L.ptr(o)
        if FLASH_KV_PREP:
            nbytes = lib.mi_flash_kv_prep_bytes(B2 * ca.heads, 1 + ws.ntot + (MAX_TEXT_LEN if ws.has_text else 0))
            prep = self._buf(ws, (nbytes + 3) // 4)
            p.kv_prep, p.kv_prep_bytes = L.ptr(prep), nbytes
        ws.prog.append((lib.mi_flash_attn_fwd, p, "flash_attn"))
        ws.prog.append(self._call(lib.mi_gemm_f32, "gemm_out", L.ptr(o), L.ptr(ca.to_out[0].weight), 0, 0, L.ptr(t), B2 * HW, Cc, inner, 0))
        ws.wide_attn = True
        return self._emit_tokens_out(ws, t, B2, Cc, h.H, h.W, ca.to_out[1], h, True)

    def _emit_self_attn_wide(self, ws, at: Attention, x: Act, want_stats: bool) -> Act:
        """layers.py:52-104 (multi-query: one shared 64-wide key / value head) + the residual"""
        if ws.store16:
            raise _Store16Unsupported()
        lib = L.lib()
        Cc, HW, batch, inner = x.C, x.H * x.W, x.batch, at.heads * 64
        xh, q, kv = self._buf(ws, batch, HW, Cc), self._buf(ws, batch, HW, inner), self._buf(ws, batch, HW, 128)
        o, t = self._buf(ws, batch, HW, inner), self._buf(ws, batch, HW, Cc)
        xa = x.c(batch)
        ws.tensors.append(xa)
        ws.prog.append(self._call(lib.mi_ln_tokens_fwd, "ln_tokens", C.byref(xa), batch, HW, L.ptr(at.norm.gamma), L.ptr(at.norm.beta), L.ptr(xh)))
        ws.prog.append(self._call(lib.mi_gemm_f32, "gemm_q", L.ptr(xh), L.ptr(at.to_q.weight), 0, 0, L.ptr(q), batch * HW, inner, Cc, 0))
        ws.prog.append(self._call(lib.mi_gemm_f32, "gemm_kv", L.ptr(xh), L.ptr(at.to_kv.weight), 0, 0, L.ptr(kv), batch * HW, 128, Cc, 0))
        p = L.MiFlashAttnParams()
        p.B, p.HW, p.heads, p.kv_heads, p.q, p.q_scale = batch, HW, at.heads, 1, L.ptr(q), 64 ** -0.5 * P.LOG2E
        p.null_k, p.null_v = L.ptr(at.null_kv), L.ptr(at.null_kv) + 4 * 64
        p.k0, p.v0, p.n0, p.ld0, p.bs0 = L.ptr(kv), L.ptr(kv) + 4 * 64, HW, 128, HW * 128
        p.out = L.ptr(o)
        if FLASH_KV_PREP and at.heads % 4 == 0:
            # K / V split + transposed once per launch, streamed to LDS by DMA (attention_wide.hip: flash_kv_prep_kernel)
            nbytes = lib.mi_flash_kv_prep_bytes(batch, HW + 1)
            prep = self._buf(ws, (nbytes + 3) // 4)
            p.kv_prep, p.kv_prep_bytes = L.ptr(prep), nbytes
        ws.prog.append((lib.mi_flash_attn_fwd, p, "flash_attn"))
        ws.prog.append(self._call(lib.mi_gemm_f32, "gemm_out", L.ptr(o), L.ptr(at.to_out[0].weight), 0, 0, L.ptr(t), batch * HW, Cc, inner, 0))
        return self._emit_tokens_out(ws, t, batch, Cc, x.H, x.W, at.to_out[1], x, want_stats)

    def _emit_chan_ff_wide(self, ws, tb: TransformerBlock, y: Act) -> Act:
        """layers.py:148-161 + the residual of :498 in token layout: ChanLayerNorm -> 1x1 conv -> GELU -> ChanLayerNorm -> 1x1 conv"""
        if ws.store16:
            raise _Store16Unsupported()
        lib = L.lib()
        Cc, HW, batch, Chid = y.C, y.H * y.W, y.batch, tb.ff[1].out_channels
        t0, h1, h2, t = self._buf(ws, batch, HW, Cc), self._buf(ws, batch, HW, Chid), self._buf(ws, batch, HW, Chid), self._buf(ws, batch, HW, Cc)
        zeros = torch.zeros(Cc, dtype=torch.float32, device=ws.dev)
        ya = y.c(batch)
        ws.tensors += [zeros, ya]
        ws.prog.append(self._call(lib.mi_ln_tokens_fwd, "ln_tokens", C.byref(ya), batch, HW, L.ptr(tb.ff[0].g), L.ptr(zeros), L.ptr(t0)))
        ws.prog.append(self._call(lib.mi_gemm_f32, "gemm_ff1", L.ptr(t0), L.ptr(tb.ff[1].weight), 0, 0, L.ptr(h1), batch * HW, Chid, Cc, 3))
        ws.prog.append(self._call(lib.mi_ln_rows_fwd, "ln_rows", L.ptr(h1), L.ptr(tb.ff[3].g), 0, L.ptr(h2), batch * HW, Chid, 1e-5))
        ws.prog.append(self._call(lib.mi_gemm_f32, "gemm_ff2", L.ptr(h2), L.ptr(tb.ff[4].weight), 0, 0, L.ptr(t), batch * HW, Cc, Chid, 0))
        return self._emit_tokens_out(ws, t, batch, Cc, y.H, y.W, None, y, True)

    def _emit_cross_attn(self, ws, pk, ca: CrossAttention, h: Act) -> Act:
        lib = L.lib()
        Cc, HW = h.C, h.H * h.W
        if Cc not in (8, 16, 32):
            return self._emit_cross_attn_wide(ws, ca, h)
        FR = lib.mi_attn_fragment_floats(Cc)
        jts = ws.JT + (ws.JT & 1)            # fp16 fragments: V chunks live per PAIR of context tiles
        gv = torch.zeros(ws.B2, ca.heads, jts, 64, FR, dtype=torch.float32, device=ws.dev)        # zero-filled: padded context rows must read as finite
        ws.gv[id(ca)] = gv
        nt = -(-HW // 64)
        out = self._new_act(ws, ws.B2, Cc, h.H, h.W, nt)
        p = L.MiCrossAttnParams()
        p.B2, p.C, p.HW, p.heads, p.J = ws.B2, Cc, HW, ca.heads, ws.J
        p.x = h.c(ws.B2)
        p.gv = L.ptr(gv)
        p.n1_g, p.n1_b = L.ptr(ca.norm.gamma), L.ptr(ca.norm.beta)
        p.n2_g, p.n2_b = L.ptr(ca.to_out[1].gamma), L.ptr(ca.to_out[1].beta)
        p.out_st = out.st
        p.out, p.out_stats, p.variant = L.ptr(out.t), L.ptr(out.stats), (7 if ws.half else 6)       # 3-term fp16 split (fp32-grade) / single term
        p.x_exp, p.g_exp, p.v_exp = pk.attn_exp[id(ca)]
        ws.prog.append((lib.mi_cross_attn_fwd, p, "cross_attn"))
        return out

    def _emit_self_attn(self, ws, pk, at: Attention, x: Act, want_stats: bool) -> Act:
        """layers.py:52-104 + residual: LayerNorm tokens -> fold them as their own context -> chunked online-softmax attention"""
        if ws.store16:
            raise _Store16Unsupported()
        lib = L.lib()
        Cc, HW, batch = x.C, x.H * x.W, x.batch
        if Cc not in (8, 16, 32):
            return self._emit_self_attn_wide(ws, at, x, want_stats)
        xh = torch.empty(batch, HW, Cc, dtype=torch.float32, device=ws.dev)
        J = HW + 1
        jt = -(-J // 16)
        FR = lib.mi_attn_fragment_floats(Cc)
        gv = torch.zeros(batch, at.heads, jt, 64, FR, dtype=torch.float32, device=ws.dev)
        ws.tensors += [xh, gv]
        xa = x.c(batch)
        ws.tensors.append(xa)

        def ln_call(_p, st, xa=xa, xh=xh):
            return lib.mi_ln_tokens_fwd(C.byref(xa), batch, HW, L.ptr(at.norm.gamma), L.ptr(at.norm.beta), L.ptr(xh), st)
        ws.prog.append((lambda p_, st: ln_call(p_, st), None, "ln_tokens"))
        mg, mv, g0, v0 = pk.attn[id(at)]
        fp = L.MiAttnFoldParams()
        fp.B2, fp.C, fp.cd, fp.heads, fp.JT = batch, Cc, Cc, at.heads, jt
        fp.c_rows, fp.c_stride_b, fp.row0, fp.nrows, fp.write_null, fp.n_blocks = L.ptr(xh), HW * Cc, 1, HW, 1, 1
        fp.blk[0].mg, fp.blk[0].mv, fp.blk[0].g0, fp.blk[0].v0, fp.blk[0].gv = L.ptr(mg), L.ptr(mv), L.ptr(g0), L.ptr(v0), L.ptr(gv)
        ws.prog.append((lib.mi_attn_fold_rows, fp, "self_fold"))
        out = self._new_act(ws, batch, Cc, x.H, x.W, -(-HW // 64) if want_stats else 0)
        p = L.MiSelfAttnParams()
        p.B2, p.C, p.HW, p.heads, p.J = batch, Cc, HW, at.heads, J
        p.x, p.gv = xa, L.ptr(gv)
        p.n1_g, p.n1_b = L.ptr(at.norm.gamma), L.ptr(at.norm.beta)
        p.n2_g, p.n2_b = L.ptr(at.to_out[1].gamma), L.ptr(at.to_out[1].beta)
        p.out, p.out_stats = L.ptr(out.t), L.ptr(out.stats)
        ws.prog.append((lib.mi_self_attn_fwd, p, "self_attn"))
        return out

    def _emit_transformer_block(self, ws, pk, tb: TransformerBlock, x: Act) -> Act:
        """layers.py:496-499: x = attn(x) + x ; x = ff(x) + x"""
        if ws.store16:
            raise _Store16Unsupported()
        lib = L.lib()
        y = self._emit_self_attn(ws, pk, tb.attn.fn, x, want_stats=False)
        if y.C not in (8, 16, 32):
            return self._emit_chan_ff_wide(ws, tb, y)
        HW = y.H * y.W
        out = self._new_act(ws, y.batch, y.C, y.H, y.W, -(-HW // 256))
        p = L.MiChanFFParams()
        p.B, p.C, p.Chid, p.HW = y.batch, y.C, tb.ff[1].out_channels, HW
        p.x = y.c(y.batch)
        p.g1, p.w1, p.g2, p.w2 = L.ptr(tb.ff[0].g), L.ptr(tb.ff[1].weight), L.ptr(tb.ff[3].g), L.ptr(tb.ff[4].weight)
        p.out, p.out_stats = L.ptr(out.t), L.ptr(out.stats)
        ws.prog.append((lib.mi_chan_ff_fwd, p, "chan_ff"))
        return out

    def _fold_params(self, ws, pk, rows_t, stride_b, row0, nrows, write_null):
        """one mi_attn_fold_rows launch covering every cross-attention block (they share C in practice; else one per C)"""
        lib = L.lib()
        calls = []
        cas = {id(m): m for m in self.unet.modules() if isinstance(m, CrossAttention)}
        by_c = {}                       # one launch per channel count C (NOT per fragment size: C = 8 and C = 16 share FR = 8 floats)
        for ca_id in ws.gv:
            if ca_id in cas:
                by_c.setdefault(cas[ca_id].to_q.in_features, []).append(ca_id)
        for _, ids in by_c.items():
            for i0 in range(0, len(ids), 8):
                chunk = ids[i0:i0 + 8]
                ca0 = cas[chunk[0]]
                p = L.MiAttnFoldParams()
                p.B2, p.C, p.cd, p.heads, p.JT = ws.B2, ca0.to_q.in_features, self.unet.cond_dim, ca0.heads, ws.JT
                p.c_rows, p.c_stride_b, p.row0, p.nrows, p.write_null = L.ptr(rows_t), stride_b, row0, nrows, write_null
                p.frag_f16 = 1
                p.n_blocks = len(chunk)
                for k, cid in enumerate(chunk):
                    mg, mv, g0, v0 = pk.attn[cid]
                    _, p.blk[k].g_exp, p.blk[k].v_exp = pk.attn_exp[cid]
                    p.blk[k].mg, p.blk[k].mv, p.blk[k].g0, p.blk[k].v0 = L.ptr(mg), L.ptr(mv), L.ptr(g0), L.ptr(v0)
                    p.blk[k].gv = L.ptr(ws.gv[cid])
                calls.append((lib.mi_attn_fold_rows, p, "fold"))
        return calls

    def _build_program(self, ws, pk):
        u = self.unet
        lib = L.lib()
        B, B2, H, W = ws.B, ws.B2, ws.H, ws.W
        # ---- K1/K5: per-step conditioning
        cp = L.MiCondStepParams()
        cp.B2, cp.B, cp.dim, cp.cd, cp.tcd, cp.ntok = B2, B, u.dim, u.cond_dim, u.time_cond_dim, u.num_time_tokens
        cp.time, cp.lowres_time, cp.freq = L.ptr(ws.times), L.ptr(ws.lowres_times), L.ptr(pk.freq)
        cp.th, cp.tc, cp.tt = _lin(u.to_time_hiddens[1]), _lin(u.to_time_cond[0]), _lin(u.to_time_tokens[0])
        if u.lowres_cond:
            cp.lth, cp.ltc, cp.ltt = _lin(u.to_lowres_time_hiddens[1]), _lin(u.to_lowres_time_cond[0]), _lin(u.to_lowres_time_tokens[0])
        cp.text_hiddens = L.ptr(ws.text_hiddens) if ws.has_text else 0
        cp.norm_w, cp.norm_b = L.ptr(u.norm_cond.weight), L.ptr(u.norm_cond.bias)
        cp.time_mlps = L.MiLinear(L.ptr(pk.tm_w), L.ptr(pk.tm_b), u.time_cond_dim, pk.R)
        cp.ss, cp.c_time, cp.t_out = L.ptr(ws.ss), L.ptr(ws.c_time), L.ptr(ws.t_out)
        # everything that depends only on (timestep, text): one step at a time for Unet.forward (ws.prog_cond), all T steps at once for
        # the sampling loop (prepare_step_tables / ws.prog_stage)
        ws.cond_params = cp
        ws.prog_cond = [(lib.mi_cond_step_fwd, cp, "cond_step")]
        if COND_GEMM and pk.R >= COND_GEMM and u.time_cond_dim % 32 == 0:
            # wide nets: the stacked time-MLPs are a [R ~ 10 K][tcd] matrix -- one mat-vec per batch row inside cond_step_kernel (one workgroup per
            # row, a weight row per work-item) took 0.54 ms per step of Unet(); as ONE GEMM over all rows the weights are read once
            cg = L.MiCondStepParams.from_buffer_copy(cp)
            ws.silu_t = self._buf(ws, B2, u.time_cond_dim)
            ws.tm_bias = pk.tm_b.unsqueeze(0).expand(B2, -1).contiguous()
            ws.tensors.append(ws.tm_bias)
            cg.ss, cg.silu_out = 0, L.ptr(ws.silu_t)
            ws.prog_cond = [(lib.mi_cond_step_fwd, cg, "cond_step"),
                            self._call(lib.mi_gemm_f32, "gemm_time_mlps", L.ptr(ws.silu_t), L.ptr(pk.tm_w), 0, L.ptr(ws.tm_bias), L.ptr(ws.ss), B2, pk.R, u.time_cond_dim, 0)]

        # ---- K3: init conv.  For super-resolution U-Nets the low-res conditioning image is constant over the T steps and
        # the convolution is linear, so conv(cat(x, lr)) = conv_x(x) + conv_lr(lr): the lr half runs once per sample()
        # (ws.prog_pre, see prepare_lowres) and is added back by the per-step kernel.
        cin = u.init_conv.convs[0].in_channels
        assert cin == u.channels * (2 if u.lowres_cond else 1)
        cfg, nt = self._tile_cfg(H, W, B)
        ce_mfma = bool(pk.ce_mfma) and W % 4 == 0
        if ws.store16 and not ce_mfma:
            raise _Store16Unsupported()
        if ce_mfma:                       # its own tile shapes: 32 x 64 for large images, 16 x 32 below (enough workgroups at 64 x 64)
            cfg = 8 if H * W >= 128 * 128 else 9
            th, tw = C.c_int(), C.c_int()
            lib.mi_conv_tile_shape(cfg, C.byref(th), C.byref(tw))
            nt = -(-H // th.value) * -(-W // tw.value)
        ctot = sum(u.init_conv.dim_scales)
        cur = self._new_act(ws, B, ctot, H, W, nt)
        if len(u.init_conv.convs) > 3:
            raise NotImplementedError("CrossEmbedLayer with more than 3 kernel sizes")

        def ce_params(src, chan0, with_bias, out_t, out_stats, addend):
            ce = L.MiCrossEmbedParams()
            ce.B, ce.H, ce.W = B, H, W
            ce.in0, ce.C0 = L.ptr(src), u.channels
            ce.n_kernels = len(u.init_conv.convs)
            for i, cv in enumerate(u.init_conv.convs):
                k, co = cv.kernel_size[0], cv.out_channels
                ce.ksize[i], ce.cout[i] = k, co
                ce.w[i] = L.ptr(pk.ce_w[i]) + 4 * chan0 * k * k * co        # packed [Cin][k][k][co]: a channel offset is a pointer offset
                ce.bias[i] = L.ptr(cv.bias) if with_bias else 0
            ce.out, ce.out_stats, ce.tile_cfg, ce.addend = L.ptr(out_t), L.ptr(out_stats), cfg, L.ptr(addend)
            ce.out_st = 1 if ws.store16 else 0
            if ce_mfma:
                tab, exps = pk.ce_mfma[chan0]
                ce.w_mfma, ce.tile_cfg = L.ptr(tab), cfg | (0x400 if ws.half else 0)
                for i in range(3):
                    ce.w_mfma_exp[i] = exps[i]
            return ce

        ws.prog_pre = []
        if u.lowres_cond:
            ws.ce_lr = torch.zeros(B, ctot, H, W, dtype=torch.bfloat16 if ws.store16 else torch.float32, device=ws.dev)
            ws.prog_pre.append((lib.mi_crossembed_fwd, ce_params(ws.lowres, u.channels, False, ws.ce_lr, None, None), "crossembed_lowres"))
            ws.prog.append((lib.mi_crossembed_fwd, ce_params(ws.x, 0, True, cur.t, cur.stats, ws.ce_lr), "crossembed"))
        else:
            ws.prog.append((lib.mi_crossembed_fwd, ce_params(ws.x, 0, True, cur.t, cur.stats, None), "crossembed"))

        hiddens: List[Act] = []
        for pre, init_block, resnet_blocks, attn_block, post in u.downs:
            if isinstance(pre, nn.Conv2d):
                cur = self._emit_conv(ws, pk, cur, None, wpack=pk.conv[id(pre)], bias=pre.bias, Cout=pre.out_channels, ksize=4, stride=2)
            cur = self._emit_resnet(ws, pk, init_block, cur, None)
            for rb in resnet_blocks:
                cur = self._emit_resnet(ws, pk, rb, cur, None)
                hiddens.append(cur)
            if isinstance(attn_block, TransformerBlock):
                cur = self._emit_transformer_block(ws, pk, attn_block, cur)
            hiddens.append(cur)
            if isinstance(post, nn.Conv2d):
                cur = self._emit_conv(ws, pk, cur, None, wpack=pk.conv[id(post)], bias=post.bias, Cout=post.out_channels, ksize=4, stride=2)
            elif isinstance(post, Parallel):
                wp, bsum = pk.conv[id(post)]
                cur = self._emit_conv(ws, pk, cur, None, wpack=wp, bias=bsum, Cout=post.fns[0].out_channels)
        cur = self._emit_resnet(ws, pk, u.mid_block1, cur, None)
        if u.mid_attn is not None:            # EinopsToAndFrom(Residual(Attention)), Unet.py:272-274
            cur = self._emit_self_attn(ws, pk, u.mid_attn.fn.fn, cur, want_stats=True)
        cur = self._emit_resnet(ws, pk, u.mid_block2, cur, None)
        for init_block, resnet_blocks, attn_block, upsample in u.ups:
            cur = self._emit_resnet(ws, pk, init_block, cur, hiddens.pop())
            for rb in resnet_blocks:
                cur = self._emit_resnet(ws, pk, rb, cur, hiddens.pop())
            if isinstance(attn_block, TransformerBlock):
                cur = self._emit_transformer_block(ws, pk, attn_block, cur)
            if isinstance(upsample, nn.Sequential):
                cv = upsample[1]
                cur = self._emit_conv(ws, pk, cur, None, wpack=pk.conv[id(cv)], bias=cv.bias, Cout=cv.out_channels, up2=1)
        cur = self._emit_resnet(ws, pk, u.final_res_block, cur, None)
        out = self._emit_conv(ws, pk, cur, None, wpack=pk.conv[id(u.final_conv)], bias=u.final_conv.bias, Cout=u.channels_out,
                              want_stats=False, conditioned=True, out_fp32=True)       # the prediction feeds the (fp32, bit-exact) sampler
        ws.pred = out.t
        if (out.H, out.W) != (H, W):
            raise L.MinImagenHipError(f"U-Net output is {out.H}x{out.W} for a {H}x{W} input (image size must be divisible by the down-sampling factor)")
        # time-token rows of the folded context, every step
        if ws.gv:
            ws.prog_cond += self._fold_params(ws, pk, ws.c_time, ws.ntot * u.cond_dim, 1, ws.ntot, 0)

    # ------------------------------------------------------------------ execution
    def set_text(self, ws, text_embeds: torch.Tensor, text_mask: Optional[torch.Tensor], keep: torch.Tensor):
        """K2 + the step-invariant part of the context fold.  Once per ``sample()`` / ``forward``."""
        u, pk, lib = self.unet, self.packed(), L.lib()
        st = L.current_stream()
        if text_embeds is None:
            # Unet.py:572-634 without text: no text hiddens are added to t, the context is [null | time tokens]; only the null row is written here
            assert not ws.has_text, "workspace was built for a text-conditioned call"
            for fn, fp, name in self._fold_params(ws, pk, ws.c_time, ws.ntot * u.cond_dim, 1, 0, 1):
                L.check(fn(C.byref(fp), st), name)
            return
        assert ws.has_text, "workspace was built for a call without text"
        text_embeds = text_embeds.to(device=ws.dev, dtype=torch.float32).contiguous()
        assert text_embeds.shape[0] == ws.B and text_embeds.shape[-1] == u.text_embed_dim
        mask8 = None if text_mask is None else text_mask.to(device=ws.dev).to(torch.uint8).contiguous()
        # a host -> device copy from pageable memory blocks the host behind everything queued on this stream (the previous call's stage,
        # in the pipelined sampler): skip it when the guidance pattern is the one already on the device (every sampling call)
        k8 = keep.to(device='cpu', dtype=torch.uint8)
        if getattr(ws, "keep_host", None) is None or not torch.equal(ws.keep_host, k8):
            ws.keep.copy_(k8)
            ws.keep_host = k8.clone()
        ws.text_keepalive = (text_embeds, mask8)
        L.require_device(text_embeds, mask8)
        p = L.MiTextCondParams()
        p.B2, p.B, p.L, p.E, p.cd, p.tcd, p.max_len = ws.B2, ws.B, text_embeds.shape[1], u.text_embed_dim, u.cond_dim, u.time_cond_dim, MAX_TEXT_LEN
        p.text_embeds, p.text_mask, p.keep = L.ptr(text_embeds), L.ptr(mask8), L.ptr(ws.keep)
        p.text_to_cond = _lin(u.text_to_cond)
        p.null_text_embed = L.ptr(u.null_text_embed)
        ln = u.to_text_non_attn_cond[0]
        p.ln_w, p.ln_b = L.ptr(ln.weight), L.ptr(ln.bias)
        p.h1, p.h2 = _lin(u.to_text_non_attn_cond[1]), _lin(u.to_text_non_attn_cond[3])
        p.null_text_hidden = L.ptr(u.null_text_hidden)
        p.norm_w, p.norm_b = L.ptr(u.norm_cond.weight), L.ptr(u.norm_cond.bias)
        p.c_text, p.text_hiddens = L.ptr(ws.c_text), L.ptr(ws.text_hiddens)
        L.check(lib.mi_text_cond_fwd(C.byref(p), st), "mi_text_cond_fwd")
        for fn, fp, name in self._fold_params(ws, pk, ws.c_text, MAX_TEXT_LEN * u.cond_dim, 1 + ws.ntot, MAX_TEXT_LEN, 1):
            L.check(fn(C.byref(fp), st), name)
        for fn, fp, name in ws.prog_text:
            L.check(fn(None, st), name)

    def prepare_lowres(self, ws, stream=None):
        """Once per ``sample()`` stage / ``forward``: everything that depends only on ws.lowres (the low-res half of CrossEmbed)."""
        st = L.current_stream() if stream is None else stream
        for fn, p, name in ws.prog_pre:
            L.check(fn(C.byref(p), st), name)

    def prepare_step_tables(self, ws, T: int, t_state: torch.Tensor, stream=None):
        """Once per ``sample()`` stage: the per-step conditioning (every ResnetBlock's scale/shift, the folded time-token rows of the
        cross-attention context) for ALL T timesteps -- the sequence T-1 .. 0 is known in advance and identical for every sample -- so
        that the denoising step only scatters row ``t`` of the tables (one small launch instead of cond_step + fold).  Same kernels,
        same arithmetic, same bits as the step-at-a-time path."""
        u, pk, lib = self.unet, self.packed(), L.lib()
        st = L.current_stream() if stream is None else stream
        B2, dev = ws.B2, ws.dev
        if ws.wide_attn:                 # the wide cross-attention projects the time tokens per step from ws.c_time: keep the per-step conditioning
            ws.prog_stage = ws.prog_cond
            return
        key = (T, t_state.data_ptr())
        tb = ws.__dict__.setdefault("step_tables", {}).get(key)
        if tb is None:
            tb = Workspace()
            n = T * B2
            # per (timestep, batch row): the scale/shift rows (they see the sample's pooled text through t).  Per timestep only: the time
            # tokens of the cross-attention context and their folded (g, v) rows -- identical for every sample, so one row per t.
            tb.times = torch.arange(T, dtype=torch.int64, device=dev).repeat_interleave(B2).contiguous()
            tb.lowres_times = torch.zeros(n, dtype=torch.int64, device=dev) if u.lowres_cond else None
            tb.text_hiddens = torch.empty(n, u.time_cond_dim, dtype=torch.float32, device=dev)
            tb.ss = torch.empty(n, max(pk.R, 1), dtype=torch.float32, device=dev)
            cp = L.MiCondStepParams.from_buffer_copy(ws.cond_params)
            cp.B2 = cp.B = n
            cp.time, cp.lowres_time = L.ptr(tb.times), L.ptr(tb.lowres_times)
            cp.text_hiddens, cp.ss, cp.c_time, cp.t_out = (L.ptr(tb.text_hiddens) if ws.has_text else 0), L.ptr(tb.ss), 0, 0
            tb.cond = cp
            tb.times_t = torch.arange(T, dtype=torch.int64, device=dev)
            tb.lowres_t = torch.zeros(T, dtype=torch.int64, device=dev) if u.lowres_cond else None
            tb.c_time_t = torch.empty(T, ws.ntot, u.cond_dim, dtype=torch.float32, device=dev)
            ct = L.MiCondStepParams.from_buffer_copy(ws.cond_params)
            ct.B2 = ct.B = T
            ct.time, ct.lowres_time = L.ptr(tb.times_t), L.ptr(tb.lowres_t)
            ct.text_hiddens, ct.ss, ct.c_time, ct.t_out = 0, 0, L.ptr(tb.c_time_t), 0
            tb.cond_t = ct
            tb.fold, tb.tables = [], []
            stage_blocks = []
            for fn, fp, name in self._fold_params(ws, pk, tb.c_time_t, ws.ntot * u.cond_dim, 1, ws.ntot, 0):
                f1 = L.MiAttnFoldParams.from_buffer_copy(fp)
                f1.B2, f1.mode = T, 1
                for k in range(fp.n_blocks):
                    t = torch.empty(T, ws.ntot, fp.heads, fp.C, 2, dtype=torch.float32, device=dev)
                    tb.tables.append(t)
                    f1.blk[k].table = L.ptr(t)
                tb.fold.append(f1)
                f2 = L.MiAttnFoldParams.from_buffer_copy(f1)            # per-step scatter of the same blocks into the B2 real rows
                f2.B2, f2.mode, f2.t_state = B2, 3, L.ptr(t_state)
                stage_blocks.append(f2)
            if not stage_blocks:                                       # no cross-attention anywhere: the scale/shift rows only
                f2 = L.MiAttnFoldParams()
                f2.B2, f2.mode, f2.t_state, f2.n_blocks = B2, 3, L.ptr(t_state), 0
                stage_blocks.append(f2)
            stage_blocks[0].ss_all, stage_blocks[0].ss, stage_blocks[0].ss_n = L.ptr(tb.ss), L.ptr(ws.ss), ws.ss.shape[1]
            tb.stage = [(lib.mi_attn_fold_rows, f2, "stage_step") for f2 in stage_blocks]
            tb.keep = t_state
            ws.step_tables[key] = tb
        tb.text_hiddens.view(T, B2, -1).copy_(ws.text_hiddens.unsqueeze(0).expand(T, -1, -1))
        if u.lowres_cond:
            tb.lowres_times.view(T, B2)[:, :ws.B].copy_(ws.lowres_times.unsqueeze(0).expand(T, -1))
            if B2 != ws.B:
                tb.lowres_times.view(T, B2)[:, ws.B:].copy_(ws.lowres_times.unsqueeze(0).expand(T, -1))
            tb.lowres_t.copy_(ws.lowres_times[:1].expand(T))           # the augmentation level is one scalar per sample() (Imagen.py:479-485)
        L.check(lib.mi_cond_step_fwd(C.byref(tb.cond), st), "mi_cond_step_fwd (all steps: scale/shift rows)")
        L.check(lib.mi_cond_step_fwd(C.byref(tb.cond_t), st), "mi_cond_step_fwd (all steps: time tokens)")
        for f1 in tb.fold:
            L.check(lib.mi_attn_fold_rows(C.byref(f1), st), "mi_attn_fold_rows (all steps)")
        ws.prog_stage = tb.stage

    def stage_prog(self, ws, t_off: int = 0):
        """The per-step conditioning launches for the step ``*t_state - t_off`` (the steps of one captured graph share one advance of
        the device-resident timestep); only the table scatter takes an offset -- the per-step ``prog_cond`` of the wide presets reads
        ``ws.times`` and needs t_off == 0."""
        if t_off == 0:
            return ws.prog_stage
        assert ws.prog_stage is not ws.prog_cond, "the step-at-a-time conditioning cannot address a timestep offset"
        cache = ws.__dict__.setdefault("prog_stage_off", {})
        key = (id(ws.prog_stage), t_off)
        if key not in cache:
            prog = []
            for fn, p, name in ws.prog_stage:
                q = L.MiAttnFoldParams.from_buffer_copy(p)
                q.t_off = t_off
                prog.append((fn, q, name))
            cache[key] = prog
        return cache[key]

    def step_offsets_supported(self, ws) -> bool:
        return ws.prog_stage is not ws.prog_cond

    def run_step(self, ws, stream=None, t_off: int = 0):
        """One denoising step's U-Net evaluation inside the sampling loop: scatter the current step's conditioning (see
        prepare_step_tables), then the image kernels."""
        st = L.current_stream() if stream is None else stream
        for fn, p, name in self.stage_prog(ws, t_off) + ws.prog:
            rc = fn(C.byref(p), st) if p is not None else fn(None, st)
            if rc != 0:
                L.check(rc, name)

    def run(self, ws, stream=None):
        """Enqueue one U-Net evaluation (all conditioning rows) on the current stream: ws.x / ws.times /
        ws.lowres / ws.lowres_times -> ws.pred."""
        st = L.current_stream() if stream is None else stream
        for fn, p, name in ws.prog_cond + ws.prog:
            rc = fn(C.byref(p), st) if p is not None else fn(None, st)
            if rc != 0:
                L.check(rc, name)

    def forward_once(self, x, time, *, lowres_cond_img, lowres_noise_times, text_embeds, text_mask, keep, cond_scale=None):
        u, lib = self.unet, L.lib()
        self.pack()
        L.require_device(x)
        B, Cc, H, W = x.shape
        assert Cc == u.channels
        two = cond_scale is not None
        ws = self.workspace(B, 2 * B if two else B, H, W, has_text=text_embeds is not None)
        ws.x.copy_(x)
        ws.times.copy_(time.to(torch.int64))
        if u.lowres_cond:
            ws.lowres.copy_(lowres_cond_img)
            ws.lowres_times.copy_(lowres_noise_times.to(torch.int64))
            self.prepare_lowres(ws)
        if two:
            keep = torch.cat((torch.ones(B, dtype=torch.bool), torch.zeros(B, dtype=torch.bool)))
        self.set_text(ws, text_embeds, text_mask, keep)
        self.run(ws)
        if not two:
            return ws.pred.clone()
        out = torch.empty(B, u.channels_out, H, W, dtype=torch.float32, device=ws.dev)
        p = L.MiCfgX0Params(B, u.channels_out * H * W, L.ptr(ws.pred), 1, float(cond_scale), 0, 0, 0, L.ptr(out), 0)
        L.check(lib.mi_cfg_x0_fwd(C.byref(p), L.current_stream()), "mi_cfg_x0_fwd")
        return out
