"""Training on the device: the convolutions of ``Imagen.forward`` -> ``Unet.forward`` in train mode
(minimagen/Imagen.py:512-573, Unet.py:355-472) on the HIP kernels, forward AND backward.

``Block`` (GroupNorm -> [scale / shift] -> SiLU -> Conv3x3, layers.py:107-145) and the plain 3x3 convolutions (``final_conv``, the conv of
``Upsample``) become ``torch.autograd.Function``s:

* forward: ONE ``mi_conv_fwd`` launch -- the fused inference kernel (GroupNorm-apply + scale/shift + SiLU + conv, conv_rp.hip); only the
  block's INPUT is kept for the backward (the normalised / activated tensors are never stored: a third of the activation memory of the
  torch-op graph);
* backward, data gradient: the same kernel on the transposed, flipped weights (a 3x3 stride-1 convolution's adjoint is a 3x3 stride-1
  convolution), range-scaled by the gradient's own channel statistics;
* backward, weight / bias gradient: ``mi_conv_wgrad`` (conv_wgrad.hip: split-K fp32 matrix-core GEMM, deterministic);
* backward, pointwise part: ``mi_block_bwd`` (train_bwd.hip) -- GroupNorm / scale-shift / SiLU backward with the activation recomputed in
  registers from the saved input (row sums, apply, parameter gradients: three launches); the weight-gradient kernel applies the same
  activation while it stages its operand tiles, so the activated tensor is never written to memory in either direction.

Weights are re-packed into matrix-core fragments (``mi_pack_conv3``: one launch per weight and direction) when their version counter changes,
i.e. once per optimiser step; the exponents of all layers come back with one host round trip (``begin_step``).  ``CrossEmbedLayer``: matrix-core forward + ``mi_crossembed_wgrad``.
``CrossAttention`` (C < dim_head): the sampler's fold with the core on ``mi_folded_attn_fwd`` /
``mi_folded_attn_bwd`` (no score tensor, layers.CrossAttention._forward_folded).  Everything else of the training graph (self-attention,
conditioning, 1x1 / k4s2 convs, LayerNorms) stays on torch ops.  ``gradient all-reduce``: minimagen_amd/distributed.py::allreduce_gradients.
MINIMAGEN_TRAIN_HIP=0 switches the whole thing off (torch ops only)."""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib as L
from . import packing as P

ENABLED = os.environ.get("MINIMAGEN_TRAIN_HIP", "1") != "0"
FORCE = False                   # tests: take the HIP path for host tensors too (emulator build of the kernels)
WGRAD_NWG = 1024         # workgroups of a conv weight-gradient launch (split-K partials are added in a fixed order)
CE_WGRAD_NWG = int(os.environ.get("MINIMAGEN_CE_WGRAD_NWG", "512"))      # workgroups of the CrossEmbed weight-gradient launch


FINGERPRINT_EVERY = int(os.environ.get("MINIMAGEN_TRAIN_FINGERPRINT", "1"))     # content fingerprint of the conv weights every N-th training step (begin_step)
# weights on the GPU: begin_step re-packs EVERY conv weight from its current values on the device and takes the power-of-two fragment scaling
# from the maxima that the PREVIOUS step sent to the host (an asynchronous copy, a whole step old when it is read) -- no host round trip per
# step.  0 = the synchronous fingerprint path above (also what host tensors / the emulator take)
LAGGED = os.environ.get("MINIMAGEN_TRAIN_LAGGED_SCALES", "1") != "0"


def active(x: torch.Tensor) -> bool:
    """the HIP training path runs for fp32 GPU tensors under autograd (and for host tensors when a test forces the emulator)"""
    return ENABLED and x.dtype == torch.float32 and x.dim() == 4 and (x.is_cuda or FORCE)


class _Pack:
    """one conv weight (or its adjoint) in the layouts mi_conv_fwd reads: [Cin][3][3][Cout_pad] fp32 (direct-conv family) and the row-paired
    fp16 hi|lo fragments + their exponent -- written by ONE launch of mi_pack_conv3 on the device"""
    __slots__ = ("generic", "frag", "exp", "cout", "cin", "rp")

    def __init__(self, w: torch.Tensor, exp: int, adjoint: bool, launch: bool = True):
        lib = L.lib()
        Cout, Cin = w.shape[0], w.shape[1]
        self.cout, self.cin = (Cin, Cout) if adjoint else (Cout, Cin)
        ct = lib.mi_conv_cout_tile(self.cout)
        pad = -(-self.cout // ct) * ct
        self.generic = torch.empty(lib.mi_pack_conv3_floats(Cout, Cin, int(adjoint), pad, 1), dtype=torch.float32, device=w.device)
        self.frag = torch.empty(lib.mi_pack_conv3_floats(Cout, Cin, int(adjoint), pad, 0), dtype=torch.float16, device=w.device)
        self.exp, self.rp = exp, self.cin % 8 == 0
        if launch:                # (launch=False: the buffers only -- begin_step's lagged mode fills every pack of a U-Net with ONE mi_pack_conv3_multi)
            L.check(lib.mi_pack_conv3(w.data_ptr(), Cout, Cin, int(adjoint), exp, self.frag.data_ptr(), self.generic.data_ptr(), pad, L.current_stream()),
                    "mi_pack_conv3")

    def desc(self, w: torch.Tensor, adjoint: bool):
        """this pack's row of mi_pack_conv3_multi's descriptor table"""
        return (w.data_ptr(), self.frag.data_ptr(), self.generic.data_ptr(), w.shape[0], w.shape[1], int(adjoint), self.exp,
                self.generic.numel() // (self.cin * 9), 0)


def _pack_key(weight: torch.Tensor):
    """identity of the values a pack was made from: in-place updates bump the version counter (every torch optimiser), ``p.data = t`` and
    ``.to()`` change the storage pointer / device"""
    return (weight._version, weight.data_ptr(), str(weight.device))


def _packs(weight: torch.Tensor, exp=None):
    """(forward pack, data-gradient pack) of a [Cout][Cin][3][3] parameter, cached on the parameter until it is updated in place"""
    cached = getattr(weight, "_mi_train_packs", None)
    if cached is not None and cached[0] == _pack_key(weight):
        return cached[1], cached[2]
    w = weight.detach()
    if exp is None:
        exp = P.rp_weight_exponent(float(w.abs().max()))
    w = w.contiguous()
    L.require_device(w)
    fwd = _Pack(w, exp, False)
    bwd = _Pack(w, exp, True)             # adjoint: W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]
    weight._mi_train_packs = (_pack_key(weight), fwd, bwd)
    return fwd, bwd


def begin_step(module: torch.nn.Module):
    """Once per training step.  Weights on the GPU (default): ``_begin_step_lagged`` -- every pack rebuilt from the current values by ONE launch, the
    fragment scaling from the previous step's asynchronously copied maxima, no host / GPU synchronisation.  Host tensors (the emulator) and
    MINIMAGEN_TRAIN_LAGGED_SCALES=0: re-pack every conv weight under ``module`` whose VALUES changed.  The version counter and the storage pointer
    (``_pack_key``) miss updates made through ``p.data`` (EMA copy-in, ``.data.mul_`` / ``.data.copy_``, hand-written SGD, an optimiser step
    replayed from a captured graph), so every weight also carries a content fingerprint (max |w|, ||w||_2): two fused multi-tensor launches
    and ONE device -> host copy for all layers -- the same copy that brings the exponents of the weights that need a new pack."""
    convs = [m for m in module.modules() if isinstance(m, torch.nn.Conv2d) and m.weight.is_floating_point()]
    if not convs:
        return
    ws = [m.weight for m in convs]
    if LAGGED and all(w.is_cuda for w in ws):
        return _begin_step_lagged(module, convs, ws)
    # The device -> host copy waits for the work queued before it (measured: 17.8 instead of 15.4 ms per SR step at B = 32, the host no longer
    # runs ahead of the GPU).  MINIMAGEN_TRAIN_FINGERPRINT=N checks every N-th step (0: never -- identity key only, round 3's behaviour);
    # weights whose identity key changed are always re-packed.
    counter = module.__dict__["_mi_step_counter"] = module.__dict__.get("_mi_step_counter", -1) + 1
    stale_id = any(getattr(w, "_mi_train_packs", None) is None or w._mi_train_packs[0] != _pack_key(w) for m, w in zip(convs, ws)
                   if m.kernel_size == (3, 3) and m.stride == (1, 1))
    check = FINGERPRINT_EVERY > 0 and counter % FINGERPRINT_EVERY == 0
    fp = None
    if check or stale_id:
        with torch.no_grad():
            det = [w.detach() for w in ws]
            # (max |w|, ||w||_2, ||w + r||_2) with r a fixed pseudo-random pattern per weight: the third term moves when values change sign or
            # place with both norms kept (w.data.neg_(), a channel permutation, an EMA copy of equal norm)
            pats = []
            for w in ws:                  # (cached on the parameter object itself)
                r = getattr(w, "_mi_fp_pattern", None)
                if r is None or r.shape != w.shape or r.device != w.device:
                    g = torch.Generator().manual_seed(w.numel() * 2654435761 % (2 ** 31))
                    r = (torch.rand(w.shape, generator=g) - 0.5).to(w.device)
                    w._mi_fp_pattern = r
                pats.append(r)
            fp = torch.stack(torch._foreach_norm(det, float("inf")) + torch._foreach_norm(det, 2) + torch._foreach_norm(torch._foreach_add(det, pats), 2)).tolist()
    n = len(ws)
    for k, (m, w) in enumerate(zip(convs, ws)):
        if fp is None:
            break
        mark = (fp[k], fp[n + k], fp[2 * n + k])
        if getattr(w, "_mi_fingerprint", None) != mark:           # values changed behind the version counter: drop everything derived from them
            w._mi_fingerprint = mark
            for attr in ("_mi_train_packs", "_mi_ce_tables"):
                if hasattr(w, attr):
                    delattr(w, attr)
        if m.kernel_size == (3, 3) and m.stride == (1, 1):
            cached = getattr(w, "_mi_train_packs", None)
            if cached is None or cached[0] != _pack_key(w):
                with torch.no_grad():
                    _packs(w, P.rp_weight_exponent(fp[k]))


_DESC_DTYPE = np.dtype([("w", "u8"), ("frag", "u8"), ("generic", "u8"), ("Cout", "i4"), ("Cin", "i4"), ("adjoint", "i4"), ("exp", "i4"),
                        ("cout_pad", "i4"), ("reserved", "i4")])          # mi_pack_conv3_desc


def _begin_step_lagged(module, convs, ws):
    """begin_step without a host round trip.  Every pack is rebuilt from the weights' CURRENT values (so no update can be missed, whichever way
    it was made); what comes from the host is only the exponent of the fragment scaling, and that tolerates stale maxima: a scale taken from the
    previous step's max |w| leaves 2^8 of head room to the fp16 range (max |w'| in [128, 256) against 65504), far more than an optimiser step
    moves a weight.  Per step: one fused max launch + one asynchronous copy into pinned memory, read at the NEXT call -- by then the copy is a
    whole step old, so the host keeps running ahead of the GPU.  The first call (no history) waits for its own maxima, once."""
    st = module.__dict__.get("_mi_lagged")
    sig = tuple(id(w) for w in ws)
    if st is None or st["sig"] != sig or st["dev"] != ws[0].device:
        st = module.__dict__["_mi_lagged"] = dict(sig=sig, dev=ws[0].device, k=0, ev=[None, None],
                                                  host=[torch.zeros(len(ws), dtype=torch.float32).pin_memory() for _ in range(2)])
    slot = st["k"] & 1
    with torch.no_grad():
        mx = torch.stack(torch._foreach_norm([w.detach() for w in ws], float("inf")))
        known = None
        if st["ev"][1 - slot] is not None:
            st["ev"][1 - slot].synchronize()
            known = st["host"][1 - slot].tolist()
        st["host"][slot].copy_(mx, non_blocking=True)
        ev = st["ev"][slot] = torch.cuda.Event()
        ev.record()
        if known is None or not all(math.isfinite(v) for v in known):
            ev.synchronize()
            known = st["host"][slot].tolist()
        st["k"] += 1
        rows = []
        packs = st.setdefault("packs", {})
        for m, w, mxk in zip(convs, ws, known):
            w._mi_fingerprint = (mxk, st["k"])                 # _weight_exp / _ce_tables: the scale, and a mark that is new every step
            for attr in ("_mi_train_packs", "_mi_ce_tables"):
                if hasattr(w, attr):
                    delattr(w, attr)
            if m.kernel_size == (3, 3) and m.stride == (1, 1):
                exp = P.rp_weight_exponent(mxk)
                if not w.is_contiguous():
                    _packs(w, exp)
                    continue
                # the pack buffers live as long as the parameter keeps its storage: the descriptor table then only changes when an exponent does
                pk = packs.get(id(w))
                if pk is None or pk[0] != (w.data_ptr(), tuple(w.shape)):
                    wd = w.detach()
                    pk = packs[id(w)] = ((w.data_ptr(), tuple(w.shape)), _Pack(wd, exp, False, launch=False), _Pack(wd, exp, True, launch=False))
                pk[1].exp = pk[2].exp = exp
                rows += [pk[1].desc(w, False), pk[2].desc(w, True)]
                w._mi_train_packs = (_pack_key(w), pk[1], pk[2])
        if rows:
            key = tuple(rows)
            if st.get("desc_key") != key:                      # (first step, or an exponent moved: a small asynchronous upload through pinned memory)
                arr = np.array(rows, dtype=_DESC_DTYPE)
                host = torch.from_numpy(arr.view(np.uint8).reshape(-1)).pin_memory()
                if st.get("desc_dev") is None or st["desc_dev"].numel() != host.numel():
                    st["desc_dev"] = torch.empty(host.numel(), dtype=torch.uint8, device=ws[0].device)
                st["desc_dev"].copy_(host, non_blocking=True)
                st["desc_key"] = key
            L.require_device(ws[0])
            L.check(L.lib().mi_pack_conv3_multi(st["desc_dev"].data_ptr(), len(rows), 16, L.current_stream()), "mi_pack_conv3_multi")


def invalidate(module: torch.nn.Module):
    """Drop every pack / table derived from the parameters under ``module`` (they are rebuilt on the next use).  ``begin_step`` notices
    changed values by itself; this is for callers that replace parameters outside a training step (``load_state_dict``, ``_apply``)."""
    module.__dict__.pop("_mi_lagged", None)
    for prm in module.parameters():
        for attr in ("_mi_train_packs", "_mi_ce_tables", "_mi_fingerprint"):
            if hasattr(prm, attr):
                delattr(prm, attr)


def _chan_stats(x: torch.Tensor) -> torch.Tensor:
    """per-(image, channel) sum and sum of squares, [B][C][1][2] (one statistics tile): what a producing kernel's epilogue would have left"""
    B, Cc, H, W = x.shape
    st = torch.empty(B, Cc, 1, 2, dtype=torch.float64, device=x.device)
    L.check(L.lib().mi_chan_stats_fwd(x.data_ptr(), st.data_ptr(), B * Cc, H * W, L.current_stream()), "mi_chan_stats_fwd")
    return st


_ZEROS = {}


def _zero_bias(n: int, dev) -> torch.Tensor:
    key = (n, str(dev))
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(n, dtype=torch.float32, device=dev)
    return _ZEROS[key]


def _conv3x3(x: torch.Tensor, pack: _Pack, bias, gn=None, ss=None, stats=None, want_stats: bool = False, res=None):
    """one mi_conv_fwd launch: out = conv3x3(act(x)) + bias with act = SiLU(GroupNorm(x) * (scale + 1) + shift) when ``gn`` is given"""
    lib = L.lib()
    B, Cin, H, W = x.shape
    Cout = pack.cout
    assert Cin == pack.cin
    L.require_device(x)
    out = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    if stats is None:
        stats = _chan_stats(x)
    if bias is None:
        bias = _zero_bias(Cout, x.device)
    keep = [stats, bias, out]
    p = L.MiConvParams()
    p.B, p.H, p.W = B, H, W
    p.in0 = L.MiAct(x.data_ptr(), Cin, stats.data_ptr(), stats.shape[2], 1.0, 0, 0)
    p.Cout, p.ksize, p.stride, p.up2 = Cout, 3, 1, 0
    p.w, p.bias, p.out = pack.generic.data_ptr(), bias.data_ptr(), out.data_ptr()
    if gn is not None:
        gamma, beta, groups, eps = gn
        p.gn_groups, p.gn_gamma, p.gn_beta, p.gn_eps = groups, gamma.data_ptr(), beta.data_ptr(), eps
        if ss is not None:
            p.scale_shift, p.ss_stride, p.ss_off = ss.data_ptr(), ss.shape[1], 0
    rp = pack.rp and W % 4 == 0
    if res is not None:             # identity residual added in the epilogue (row-paired narrow family only: residual_fusable); out_stats describe the SUM
        assert rp and Cin <= 64 and Cout <= 32 and res.shape == out.shape and res.is_contiguous()
        p.res0 = L.MiAct(res.data_ptr(), Cout, 0, 0, 1.0, 0, 0)
        keep.append(res)
    if rp:
        wide = not (Cin <= 64 and Cout <= 32)
        cfg = 6
        if wide or Cout > 16 or (H * W <= 64 * 64 and (W <= 32 or Cout <= 8)):
            cfg = 7
        p.w_rp, p.w_rp_exp = pack.frag.data_ptr(), pack.exp
        if wide:
            coef = torch.empty(B, Cin, 4, dtype=torch.float32, device=x.device)
            exps = torch.empty(B, 2, dtype=torch.int32, device=x.device)
            keep += [coef, exps]
            p.gn_coef, p.gn_exps = coef.data_ptr(), exps.data_ptr()
        p.tile_cfg = cfg
        if wide:
            L.check(lib.mi_gn_coef_fwd(C.byref(p), L.current_stream()), "mi_gn_coef_fwd (training)")
    else:
        p.tile_cfg = 0 if (W >= 64 and H * W > 64 * 64) else 2
    out_stats = None
    if want_stats:                  # the epilogue's per-tile partial statistics of the OUTPUT: the next Block reads them instead of a statistics pass
        th, tw = C.c_int(), C.c_int()
        lib.mi_conv_tile_shape(p.tile_cfg & 0xff, C.byref(th), C.byref(tw))
        out_stats = torch.empty(B, Cout, (-(-H // th.value)) * (-(-W // tw.value)), 2, dtype=torch.float64, device=x.device)
        p.out_stats = out_stats.data_ptr()
    L.check(lib.mi_conv_fwd(C.byref(p), L.current_stream()), "mi_conv_fwd (training)")
    return (out, out_stats) if want_stats else out


def _wgrad(a: torch.Tensor, dy: torch.Tensor, want_bias: bool, act=None):
    """dW, db of a 3x3 conv from its input ``a`` and the output gradient; ``act`` = (stats, gamma, beta, groups, eps, ss): ``a`` is the raw
    Block input and the kernel applies GroupNorm -> scale/shift -> SiLU while staging the tiles"""
    lib = L.lib()
    B, Cin, H, W = a.shape
    Cout = dy.shape[1]
    tiles = B * (-(-H // 8)) * (-(-W // 32))
    nblk = (-(-Cin // 16)) * (-(-Cout // 16))
    nwg = max(1, min(tiles, max(64, WGRAD_NWG // nblk)))
    part = torch.empty(lib.mi_conv_wgrad_workspace(Cin, Cout, nwg), dtype=torch.float32, device=a.device)
    dw = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=a.device)
    db = torch.empty(Cout, dtype=torch.float32, device=a.device) if want_bias else None
    p = L.MiConvWgradParams()
    p.B, p.Cin, p.Cout, p.H, p.W = B, Cin, Cout, H, W
    p.a, p.dy, p.dw, p.db, p.partial, p.nwg = a.data_ptr(), dy.data_ptr(), dw.data_ptr(), L.ptr(db), part.data_ptr(), nwg
    if act is not None:
        stats, gamma, beta, groups, eps, ss = act
        p.a_stats, p.a_nt, p.gamma, p.beta, p.groups, p.eps = stats.data_ptr(), stats.shape[2], gamma.data_ptr(), beta.data_ptr(), groups, eps
        if ss is not None:
            p.ss, p.ss_stride, p.ss_off = ss.data_ptr(), ss.shape[1], 0
    L.check(lib.mi_conv_wgrad(C.byref(p), L.current_stream()), "mi_conv_wgrad")
    return dw, db


def _block_bwd(x, da, stats, gamma, beta, groups, eps, ss):
    """dx, dgamma, dbeta, d(scale|shift) of GroupNorm -> scale/shift -> SiLU from the activation's gradient (mi_block_bwd: three launches)"""
    lib = L.lib()
    B, Cc, H, W = x.shape
    HW = H * W
    nchunk = max(1, min(HW // 1024, -(-2048 // (B * Cc))))
    uv = torch.empty(B, Cc, nchunk, 2, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
    dss = torch.empty(B, 2 * Cc, dtype=torch.float32, device=x.device) if ss is not None else None
    p = L.MiBlockBwdParams()
    p.B, p.C, p.HW, p.groups, p.nt, p.nchunk, p.eps = B, Cc, HW, groups, stats.shape[2], nchunk, eps
    p.x, p.da, p.x_stats, p.gamma, p.beta = x.data_ptr(), da.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    if ss is not None:
        p.ss, p.ss_stride, p.ss_off = ss.data_ptr(), ss.shape[1], 0
    dx_stats = torch.empty(B, Cc, nchunk, 2, dtype=torch.float64, device=x.device)
    p.uv, p.dx, p.dgamma, p.dbeta, p.dss, p.dx_stats = uv.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), L.ptr(dss), dx_stats.data_ptr()
    L.check(lib.mi_block_bwd(C.byref(p), L.current_stream()), "mi_block_bwd")
    dx._mi_stats = (dx_stats, dx._version)        # the previous Block's backward receives this very tensor as its output gradient
    return dx, dgamma, dbeta, dss


class _BlockFn(torch.autograd.Function):
    """layers.py:131-145: fused HIP forward; backward = data-gradient conv (the forward kernel on the adjoint weights) -> mi_block_bwd
    (GroupNorm / scale-shift / SiLU backward, recomputing the activation in registers) + mi_conv_wgrad with the activation fused into its
    operand staging.  Saved for the backward: the block's input, its channel statistics and the scale|shift table -- nothing else."""

    @staticmethod
    def forward(ctx, x, gamma, beta, scale, shift, weight, bias, groups, eps, x_stats, residual=None):
        x = x.contiguous()
        fwd, _ = _packs(weight)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        ss = None
        if scale is not None:
            B = x.shape[0]
            ss = torch.cat((scale.detach().reshape(B, -1), shift.detach().reshape(B, -1)), 1).contiguous()
        stats = x_stats if x_stats is not None else _chan_stats(x)        # left by the producing Block's epilogue, or one statistics pass
        out, out_stats = _conv3x3(x, fwd, None if bias is None else bias.detach(), gn=(g, b, groups, eps), ss=ss, stats=stats, want_stats=True,
                                  res=None if residual is None else residual.detach())
        ctx.save_for_backward(x, g, b, ss, stats, weight)
        ctx.has_res = residual is not None
        ctx.groups, ctx.eps, ctx.has_bias, ctx.ss_shape = groups, eps, bias is not None, (None if scale is None else scale.shape)
        ctx.mark_non_differentiable(out_stats)
        return out, out_stats

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        x, gamma, beta, ss, stats, weight = ctx.saved_tensors
        dy = dy.contiguous()
        _, bwd = _packs(weight)
        need = ctx.needs_input_grad
        dx = dgamma = dbeta = dscale = dshift = dw = db = None
        if any(need[:5]):
            tag = getattr(dy, "_mi_stats", None)          # left by the consumer Block's backward when dy is its dx, unmodified
            dy_stats = tag[0] if (tag is not None and tag[1] == dy._version and tag[0].shape[:2] == dy.shape[:2]) else None
            da = _conv3x3(dy, bwd, None, stats=dy_stats)
            dx, dgamma, dbeta, dss = _block_bwd(x, da, stats, gamma, beta, ctx.groups, ctx.eps, ss)
            if dss is not None:
                Cc = x.shape[1]
                dscale, dshift = dss[:, :Cc].reshape(ctx.ss_shape), dss[:, Cc:].reshape(ctx.ss_shape)
        if need[5] or (ctx.has_bias and need[6]):
            dw, db = _wgrad(x, dy, ctx.has_bias and need[6], act=(stats, gamma, beta, ctx.groups, ctx.eps, ss))
        pick = lambda g, n: g if n else None
        return (pick(dx, need[0]), pick(dgamma, need[1]), pick(dbeta, need[2]), pick(dscale, need[3]), pick(dshift, need[4]),
                pick(dw, need[5]), db, None, None, None, (dy if (ctx.has_res and need[10]) else None))


class _ConvFn(torch.autograd.Function):
    """a plain 3x3 stride-1 convolution (Unet.py final_conv, layers.py:512-515 after the nearest x2): forward, data gradient and
    parameter gradients on the HIP kernels"""

    @staticmethod
    def forward(ctx, x, weight, bias, exp=None):
        x = x.contiguous()
        fwd, _ = _packs(weight, exp)
        out = _conv3x3(x, fwd, None if bias is None else bias.detach())
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.exp = exp
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        _, bwd = _packs(weight, ctx.exp)
        need = ctx.needs_input_grad
        dx = _conv3x3(dy, bwd, None) if need[0] else None
        dw = db = None
        if need[1] or (ctx.has_bias and need[2]):
            dw, db = _wgrad(x, dy, ctx.has_bias and need[2])
        return dx, (dw if need[1] else None), db, None


def _ce_tables(convs, channels: int):
    """Toeplitz fragment tables of the matrix-core CrossEmbed kernel (packing.pack_crossembed_mfma), one per input half, cached on the first
    member's weight until any member is updated; ONE device->host copy of the (small) weights per optimiser step"""
    key = tuple((_pack_key(c.weight), getattr(c.weight, "_mi_fingerprint", None)) for c in convs)
    cached = getattr(convs[0].weight, "_mi_ce_tables", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    dev = convs[0].weight.device
    marks = [getattr(c.weight, "_mi_fingerprint", None) for c in convs]
    if LAGGED and dev.type == "cuda" and all(m is not None and len(m) == 2 for m in marks):
        # begin_step's lagged mode: tables gathered on the device from the current weights, scaled by the exponents of the previous step's
        # whole-tensor maxima (>= the packed channel slice's own: the head room only grows) -- no device -> host copy
        idxs = getattr(convs[0].weight, "_mi_ce_index", None)
        if idxs is None:
            idxs = convs[0].weight._mi_ce_index = {}
        exps = [P.rp_weight_exponent(m[0]) for m in marks]
        tabs = {}
        with torch.no_grad():
            for chan0 in range(0, convs[0].in_channels, channels):
                ik = (chan0, channels, tuple(tuple(c.weight.shape) for c in convs), str(dev))
                if ik not in idxs:
                    idxs[ik] = P.crossembed_mfma_gather_index([c.weight.shape for c in convs], chan0, channels).to(dev)
                tabs[chan0] = (P.pack_crossembed_mfma_device([c.weight for c in convs], idxs[ik], exps), exps)
        convs[0].weight._mi_ce_tables = (key, tabs)
        return tabs
    with torch.no_grad():
        flat = torch.cat([c.weight.detach().reshape(-1) for c in convs]).cpu()
    ws, off = [], 0
    for c in convs:
        n = c.weight.numel()
        ws.append(flat[off:off + n].reshape(c.weight.shape))
        off += n
    tabs = {}
    for chan0 in range(0, convs[0].in_channels, channels):
        tab, exps = P.pack_crossembed_mfma(ws, chan0, channels)
        tabs[chan0] = (tab.to(dev, non_blocking=True), exps)
    convs[0].weight._mi_ce_tables = (key, tabs)
    return tabs


def crossembed_supported(layer, x: torch.Tensor, lowres) -> bool:
    """the matrix-core forward + the shared-correlation weight gradient cover the reference's configuration: kernel sizes (3, 7, 15),
    dim_scales (4, 2, 2), stride 1, <= 4 image channels per half and <= 6 in total (mi_crossembed_wgrad), W % 4 == 0; the input must not need a gradient (it is the first layer)"""
    cv = layer.convs
    return (active(x) and layer.stride == 1 and [c.kernel_size[0] for c in cv] == [3, 7, 15] and [c.out_channels for c in cv] == [4, 2, 2]
            and x.shape[1] <= 4 and cv[0].in_channels <= 6 and x.shape[3] % 4 == 0 and not x.requires_grad and (lowres is None or (not lowres.requires_grad and lowres.shape == x.shape))
            and cv[0].in_channels == x.shape[1] * (2 if lowres is not None else 1) and all(c.bias is not None for c in cv))


class _CrossEmbedFn(torch.autograd.Function):
    """CrossEmbedLayer (layers.py:298-305) of the image (| low-res conditioning image): forward on the matrix-core Toeplitz kernel (one launch
    per input half, the second as addend of the first), weight / bias gradients by mi_crossembed_wgrad"""

    @staticmethod
    def forward(ctx, x, lowres, tabs, w0, b0, w1, b1, w2, b2):
        lib = L.lib()
        x = x.contiguous()
        B, Cc, H, W = x.shape
        ws_, bs_ = (w0, w1, w2), (b0, b1, b2)
        cfg = 8 if H * W >= 128 * 128 else 9
        out = torch.empty(B, 8, H, W, dtype=torch.float32, device=x.device)
        halves = [(x, 0)] + ([(lowres.contiguous(), Cc)] if lowres is not None else [])
        addend = None
        for src, chan0 in reversed(halves):             # the low-res half first (no bias), then the image half adds it and the biases
            dst = out if chan0 == 0 else torch.empty_like(out)
            p = L.MiCrossEmbedParams()
            p.B, p.H, p.W, p.in0, p.C0, p.n_kernels = B, H, W, src.data_ptr(), Cc, 3
            tab, exps = tabs[chan0]
            for i in range(3):
                p.ksize[i], p.cout[i], p.w_mfma_exp[i] = ws_[i].shape[-1], ws_[i].shape[0], exps[i]
                p.bias[i] = bs_[i].data_ptr() if chan0 == 0 else 0
            p.w_mfma, p.out, p.tile_cfg, p.addend = tab.data_ptr(), dst.data_ptr(), cfg, L.ptr(addend)
            L.check(lib.mi_crossembed_fwd(C.byref(p), L.current_stream()), "mi_crossembed_fwd (training)")
            addend = dst
        ctx.save_for_backward(x if lowres is None else torch.cat((x, lowres), 1))
        ctx.shapes = [w.shape for w in ws_]
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        (xin,) = ctx.saved_tensors
        dy = dy.contiguous()
        B, Cin, H, W = xin.shape
        tiles = B * (-(-H // 8)) * (-(-W // 32))
        nwg = max(1, min(tiles, CE_WGRAD_NWG))
        part = torch.empty(lib.mi_crossembed_wgrad_workspace(Cin, 15, nwg), dtype=torch.float32, device=xin.device)
        dws = [torch.empty(sh, dtype=torch.float32, device=xin.device) for sh in ctx.shapes]
        dbs = [torch.empty(sh[0], dtype=torch.float32, device=xin.device) for sh in ctx.shapes]
        p = L.MiCrossEmbedWgradParams()
        p.B, p.Cin, p.H, p.W, p.x, p.dy, p.n_kernels, p.partial, p.nwg = B, Cin, H, W, xin.data_ptr(), dy.data_ptr(), 3, part.data_ptr(), nwg
        for i in range(3):
            p.ksize[i], p.cout[i], p.dw[i], p.db[i] = ctx.shapes[i][-1], ctx.shapes[i][0], dws[i].data_ptr(), dbs[i].data_ptr()
        L.check(lib.mi_crossembed_wgrad(C.byref(p), L.current_stream()), "mi_crossembed_wgrad")
        return None, None, None, dws[0], dbs[0], dws[1], dbs[1], dws[2], dbs[2]


def crossembed_forward(layer, x: torch.Tensor, lowres) -> torch.Tensor:
    cv = layer.convs
    tabs = _ce_tables(cv, x.shape[1])
    return _CrossEmbedFn.apply(x, lowres, tabs, cv[0].weight, cv[0].bias, cv[1].weight, cv[1].bias, cv[2].weight, cv[2].bias)


def folded_attention_supported(q: torch.Tensor, kf: torch.Tensor) -> bool:
    """mi_folded_attn_fwd / _bwd: fp32, C in {8, 16, 32}, J * C <= 6144 (the BASELINE U-Nets: C = 16, J = 259 / 261)"""
    return (ENABLED and q.dtype == torch.float32 and (q.is_cuda or FORCE) and q.shape[-1] in (8, 16, 32) and kf.shape[2] * kf.shape[3] <= 6144
            and kf.shape[2] <= 1024)


class _FoldedAttnFn(torch.autograd.Function):
    """out[b, i] = sum_h sum_j softmax_j(q[b, i] . kf[b, h, j]) vf[b, h, j] (the core of layers.CrossAttention._forward_folded) on the HIP
    kernels of attn_train.hip: the [tokens x heads x context] score tensor is never materialised, forward or backward; saved for the
    backward: q, kf, vf and the per-(token, head) logsumexp"""

    @staticmethod
    def forward(ctx, q, kf, vf, mask):
        lib = L.lib()
        q, kf, vf = q.contiguous(), kf.contiguous(), vf.contiguous()
        L.require_device(q, kf, vf)
        B, n, Cc = q.shape
        H, J = kf.shape[1], kf.shape[2]
        m8 = None if mask is None else mask.to(torch.uint8).contiguous()
        out = torch.empty_like(q)
        lse = torch.empty(B, n, H, dtype=torch.float32, device=q.device)
        oh = torch.empty(B, n, H, Cc, dtype=torch.float32, device=q.device)       # per-head outputs (8 x the output: 67 MB at 64 x 64, B = 32)
        p = L.MiFoldedAttnParams()
        p.B, p.n, p.H, p.J, p.C, p.nchunk = B, n, H, J, Cc, 1
        p.q, p.kf, p.vf, p.mask, p.out, p.lse, p.oh = q.data_ptr(), kf.data_ptr(), vf.data_ptr(), L.ptr(m8), out.data_ptr(), lse.data_ptr(), oh.data_ptr()
        L.check(lib.mi_folded_attn_fwd(C.byref(p), L.current_stream()), "mi_folded_attn_fwd")
        ctx.save_for_backward(q, kf, vf, m8, lse, oh)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.lib()
        q, kf, vf, m8, lse, oh = ctx.saved_tensors
        dout = dout.contiguous()
        B, n, Cc = q.shape
        H, J = kf.shape[1], kf.shape[2]
        nchunk = max(1, min(n // 128, -(-2048 // (B * H))))
        dq = torch.empty_like(q)
        dsum = torch.empty_like(lse)
        dkf = torch.empty(nchunk, B, H, J, Cc, dtype=torch.float32, device=q.device)
        dvf = torch.empty_like(dkf)
        p = L.MiFoldedAttnParams()
        p.B, p.n, p.H, p.J, p.C, p.nchunk = B, n, H, J, Cc, nchunk
        p.q, p.kf, p.vf, p.mask, p.lse = q.data_ptr(), kf.data_ptr(), vf.data_ptr(), L.ptr(m8), lse.data_ptr()
        p.dout, p.dsum, p.dq, p.dkf, p.dvf, p.oh = dout.data_ptr(), dsum.data_ptr(), dq.data_ptr(), dkf.data_ptr(), dvf.data_ptr(), oh.data_ptr()
        L.check(lib.mi_folded_attn_bwd(C.byref(p), L.current_stream()), "mi_folded_attn_bwd")
        return dq, (dkf[0] if nchunk == 1 else dkf.sum(0)), (dvf[0] if nchunk == 1 else dvf.sum(0)), None


def folded_attention(q: torch.Tensor, kf: torch.Tensor, vf: torch.Tensor, mask) -> torch.Tensor:
    return _FoldedAttnFn.apply(q, kf, vf, mask)


def layer_norm_supported(x: torch.Tensor, weight) -> bool:
    """mi_layernorm_fwd / _bwd: fp32, last dimension <= 1024, an affine weight"""
    return (ENABLED and torch.is_grad_enabled() and x.dtype == torch.float32 and (x.is_cuda or FORCE) and weight is not None and x.dim() >= 2
            and 0 < x.shape[-1] <= 1024 and x.numel() > 0)


class _LayerNormFn(torch.autograd.Function):
    """LayerNorm over the last dimension (the reference's LayerNorm, layers.py:333-343, and the nn.LayerNorm members of the conditioning stack) on
    train_ln.hip, forward and backward; saved for the backward: the input and (mean, rstd) per row -- not the normalised tensor"""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        lib = L.lib()
        x2 = x.contiguous().view(-1, x.shape[-1])
        L.require_device(x2)
        rows, dim = x2.shape
        y = torch.empty_like(x2)
        stat = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
        w = weight.detach().contiguous()
        b = None if bias is None else bias.detach().contiguous()
        L.check(lib.mi_layernorm_fwd(x2.data_ptr(), w.data_ptr(), L.ptr(b), y.data_ptr(), stat.data_ptr(), rows, dim, float(eps), L.current_stream()), "mi_layernorm_fwd")
        ctx.save_for_backward(x2, w, stat)
        ctx.has_bias = bias is not None
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        x2, w, stat = ctx.saved_tensors
        rows, dim = x2.shape
        dy2 = dy.contiguous().view(rows, dim)
        dx = torch.empty_like(x2)
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        partial = dgamma = dbeta = None
        if need_w:
            partial = torch.empty(lib.mi_layernorm_bwd_nwg(rows, dim), 2, dim, dtype=torch.float32, device=x2.device)
            dgamma = torch.empty(dim, dtype=torch.float32, device=x2.device)
            dbeta = torch.empty(dim, dtype=torch.float32, device=x2.device) if ctx.has_bias else None
        L.check(lib.mi_layernorm_bwd(dy2.data_ptr(), x2.data_ptr(), w.data_ptr(), stat.data_ptr(), dx.data_ptr(), L.ptr(partial), L.ptr(dgamma), L.ptr(dbeta),
                                     rows, dim, L.current_stream()), "mi_layernorm_bwd")
        return dx.view(dy.shape), dgamma, dbeta, None


def layer_norm(x: torch.Tensor, weight, bias, eps: float = 1e-5) -> torch.Tensor:
    """``F.layer_norm(x, x.shape[-1:], weight, bias, eps)`` on the HIP kernels where they apply (training on the device), else torch's"""
    if layer_norm_supported(x, weight):
        return _LayerNormFn.apply(x, weight, bias, eps)
    return torch.nn.functional.layer_norm(x, x.shape[-1:], weight, bias, eps)


def residual_fusable(block, x: torch.Tensor, residual) -> bool:
    """the conv epilogue can add ``residual`` (ResnetBlock: h + res_conv(x), layers.py:439): narrow row-paired family, same shape, fp32, contiguous"""
    conv = block.project
    return (residual is not None and isinstance(block.groupnorm, torch.nn.GroupNorm) and conv.in_channels % 8 == 0 and x.shape[-1] % 4 == 0
            and conv.in_channels <= 64 and conv.out_channels <= 32 and residual.dtype == torch.float32 and residual.is_contiguous()
            and tuple(residual.shape) == (x.shape[0], conv.out_channels, x.shape[2], x.shape[3]))


def block_forward(block, x: torch.Tensor, scale_shift=None, residual=None) -> torch.Tensor:
    """``Block.forward`` (layers.py:131-145) through _BlockFn; ``residual`` (optional) is added to the output -- in the conv's epilogue where the
    kernel family allows (one launch and one statistics pass less per ResnetBlock: the next Block reads the statistics of the SUM)"""
    gnm = block.groupnorm
    conv = block.project
    scale, shift = scale_shift if scale_shift is not None else (None, None)
    if not isinstance(gnm, torch.nn.GroupNorm):             # Block(norm=False): not a layer of the reference's U-Nets; keep the semantics
        h = x if scale is None else x * (scale + 1) + shift
        out = _ConvFn.apply(F.silu(h), conv.weight, conv.bias)
        return out if residual is None else out + residual
    # statistics handed over by the Block that produced x (block1 -> block2 of a ResnetBlock without cross-attention): valid for this very tensor
    # object while it has not been written since
    tag = getattr(x, "_mi_stats", None)
    x_stats = tag[0] if (tag is not None and tag[1] == x._version and x.is_contiguous() and tag[0].shape[:2] == x.shape[:2]) else None
    fuse = residual_fusable(block, x, residual)
    out, out_stats = _BlockFn.apply(x, gnm.weight, gnm.bias, scale, shift, conv.weight, conv.bias, gnm.num_groups, gnm.eps, x_stats, residual if fuse else None)
    if residual is not None and not fuse:
        return out + residual
    out._mi_stats = (out_stats, out._version)
    return out


def conv3x3_forward(conv: torch.nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    return _ConvFn.apply(x, conv.weight, conv.bias)


def _weight_exp(weight: torch.Tensor):
    """exponent of the fragment scaling from the max |w| that begin_step() already brought to the host (None: _packs fetches it)"""
    mark = getattr(weight, "_mi_fingerprint", None)
    return P.rp_weight_exponent(mark[0]) if mark is not None else None


def is_conv1x1(m, x=None) -> bool:
    ok = isinstance(m, torch.nn.Conv2d) and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0) and m.groups == 1
    if ok and x is not None:
        ok = conv_shape_supported(m.in_channels, m.out_channels, x.shape[-1])
    return ok


def conv1x1_forward(conv: torch.nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """A 1x1 convolution (ResnetBlock.res_conv, layers.py:415; Parallel's second member, :346-356) as the centre tap of a 3x3 one: forward, data
    gradient and weight gradient on the HIP 3x3 kernels (the zero taps cost matrix-core work on an 8..32-channel layer that is bound by its
    activation traffic); autograd takes the centre of the 3x3 weight gradient back to the parameter."""
    w3 = F.pad(conv.weight, (1, 1, 1, 1))
    return _ConvFn.apply(x, w3, conv.bias, _weight_exp(conv.weight))


def is_conv4x4s2(m, x=None) -> bool:
    ok = isinstance(m, torch.nn.Conv2d) and m.kernel_size == (4, 4) and m.stride == (2, 2) and m.padding == (1, 1) and m.groups == 1 \
        and m.dilation == (1, 1) and m.padding_mode == "zeros"
    if ok and x is not None:
        ok = x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0 and conv_shape_supported(4 * m.in_channels, m.out_channels, x.shape[-1] // 2)
    return ok


def conv4x4s2_forward(conv: torch.nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """Downsample (layers.py:308-319: Conv2d k4 s2 p1) as a 3x3 stride-1 convolution over the space-to-depth image: with x'[4c + 2py + px][Y][X] =
    x[c][2Y + py][2X + px] the tap ky of the stride-2 kernel reads row 2y + ky - 1 = 2 (y + a) + py, (a, py) = (-1, 1), (0, 0), (0, 1), (1, 0)
    for ky = 0..3 -- a 3x3 kernel over 4 Cin channels in which 20 of the 36 (a, py, b, px) combinations are zero.  pixel_unshuffle is one copy;
    forward, data gradient and weight gradient then run on the HIP 3x3 kernels, and autograd scatters / gathers between the two weight layouts
    (16 slices of a [Cout, Cin, 4, 4] tensor)."""
    w = conv.weight
    Cout, Cin = w.shape[0], w.shape[1]
    # ONE gather for the re-layout (and one index_add in its backward): 16 sliced assignments were ~50 small launches per Downsample and step
    # on a training step that is bound by its launch count
    w3 = F.pad(w.reshape(Cout, Cin, 16), (0, 1)).index_select(2, _k4_index(w.device)).reshape(Cout, 4 * Cin, 3, 3)
    return _ConvFn.apply(F.pixel_unshuffle(x, 2), w3, conv.bias, _weight_exp(w))


_K4_INDEX = {}


def _k4_index(dev) -> torch.Tensor:
    """[py, px, a + 1, b + 1] flattened -> 4 ky + kx of the stride-2 kernel's tap, 16 (a zero column appended to the taps) where there is none"""
    key = str(dev)
    if key not in _K4_INDEX:
        idx = torch.full((2, 2, 3, 3), 16, dtype=torch.long)
        tap = ((0, 1), (1, 0), (1, 1), (2, 0))                        # ky -> (a + 1, py)
        for ky in range(4):
            for kx in range(4):
                (ay, py), (ax, px) = tap[ky], tap[kx]
                idx[py, px, ay, ax] = 4 * ky + kx
        _K4_INDEX[key] = idx.reshape(-1).to(dev)
    return _K4_INDEX[key]


def conv_shape_supported(cin: int, cout: int, W: int, groups: int = 0) -> bool:
    """the limits of mi_conv_fwd / mi_conv_wgrad as the kernels check them: the row-paired matrix-core family needs channel octets and
    W % 4 == 0 and goes up to 4096 input channels (wide regime); everything else runs the direct-conv family (<= 256 input channels);
    GroupNorm with at most 32 groups.  Both directions must fit (the data gradient is the same conv with the channel counts swapped)."""
    def one(ci, co):
        rp = ci % 8 == 0 and W % 4 == 0
        return ci <= (4096 if rp else 256)
    return one(cin, cout) and one(cout, cin) and groups <= 32


def is_plain_conv3x3(m, x=None) -> bool:
    ok = isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and m.groups == 1 \
        and m.dilation == (1, 1) and m.padding_mode == "zeros"
    if ok and x is not None:
        ok = conv_shape_supported(m.in_channels, m.out_channels, x.shape[-1])
    return ok


def block_supported(block, x: torch.Tensor) -> bool:
    gnm = block.groupnorm
    groups = gnm.num_groups if isinstance(gnm, torch.nn.GroupNorm) else 0
    return conv_shape_supported(block.project.in_channels, block.project.out_channels, x.shape[-1], groups)
