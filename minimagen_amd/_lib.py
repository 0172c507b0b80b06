"""ctypes binding of the C ABI in include/minimagen_hip.h (libminimagen_hip.so, gfx950).

There is NO CPU fallback: if the HIP library is missing, or a tensor is not on the GPU,
every op raises.  (The dev-only SIMT emulator under tools/hipemu builds the same C ABI for
x86; tests install it explicitly with ``use_library`` -- the product never does.)
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MINIMAGEN_HIP_LIB selects another gfx950 build of the same C ABI (kernel A/B experiments); never a CPU library
DEFAULT_LIB = os.environ.get("MINIMAGEN_HIP_LIB") or os.path.join(_HERE, "libminimagen_hip.so")

c_float_p = C.c_void_p   # raw device pointers travel as integers


class MiAct(C.Structure):
    _fields_ = [("data", C.c_void_p), ("C", C.c_int), ("stats", C.c_void_p), ("nt", C.c_int), ("scale", C.c_float), ("bmod", C.c_int),
                ("st", C.c_int)]


class MiConvParams(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("in0", MiAct), ("in1", MiAct),
        ("Cout", C.c_int), ("ksize", C.c_int), ("stride", C.c_int), ("up2", C.c_int),
        ("w", C.c_void_p), ("bias", C.c_void_p),
        ("gn_groups", C.c_int), ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p), ("gn_eps", C.c_float),
        ("scale_shift", C.c_void_p), ("ss_stride", C.c_int), ("ss_off", C.c_int),
        ("res0", MiAct), ("res1", MiAct), ("res_w", C.c_void_p), ("res_b", C.c_void_p),
        ("out", C.c_void_p), ("out_st", C.c_int), ("out_stats", C.c_void_p), ("tile_cfg", C.c_int),
        ("w_rp", C.c_void_p), ("res_w_rp", C.c_void_p), ("w_rp_exp", C.c_int), ("res_w_rp_exp", C.c_int),
        ("gn_coef", C.c_void_p), ("gn_exps", C.c_void_p),
        ("act_prep", C.c_void_p), ("act_prep_bytes", C.c_longlong),
    ]


class MiCrossEmbedParams(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("in0", C.c_void_p), ("C0", C.c_int), ("in1", C.c_void_p), ("C1", C.c_int),
        ("in1_batch_mod", C.c_int), ("in0_batch_mod", C.c_int), ("n_kernels", C.c_int),
        ("ksize", C.c_int * 3), ("cout", C.c_int * 3), ("w", C.c_void_p * 3), ("bias", C.c_void_p * 3),
        ("out", C.c_void_p), ("out_stats", C.c_void_p), ("out_st", C.c_int), ("tile_cfg", C.c_int), ("addend", C.c_void_p),
        ("w_mfma", C.c_void_p), ("w_mfma_exp", C.c_int * 3),
    ]


class MiLinear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("in_", C.c_int), ("out", C.c_int)]


class MiTextCondParams(C.Structure):
    _fields_ = [
        ("B2", C.c_int), ("B", C.c_int), ("L", C.c_int), ("E", C.c_int), ("cd", C.c_int), ("tcd", C.c_int), ("max_len", C.c_int),
        ("text_embeds", C.c_void_p), ("text_mask", C.c_void_p), ("keep", C.c_void_p),
        ("text_to_cond", MiLinear), ("null_text_embed", C.c_void_p), ("ln_w", C.c_void_p), ("ln_b", C.c_void_p),
        ("h1", MiLinear), ("h2", MiLinear), ("null_text_hidden", C.c_void_p), ("norm_w", C.c_void_p), ("norm_b", C.c_void_p),
        ("c_text", C.c_void_p), ("text_hiddens", C.c_void_p),
    ]


class MiCondStepParams(C.Structure):
    _fields_ = [
        ("B2", C.c_int), ("B", C.c_int), ("dim", C.c_int), ("cd", C.c_int), ("tcd", C.c_int), ("ntok", C.c_int),
        ("time", C.c_void_p), ("lowres_time", C.c_void_p), ("freq", C.c_void_p),
        ("th", MiLinear), ("tc", MiLinear), ("tt", MiLinear), ("lth", MiLinear), ("ltc", MiLinear), ("ltt", MiLinear),
        ("text_hiddens", C.c_void_p), ("norm_w", C.c_void_p), ("norm_b", C.c_void_p), ("time_mlps", MiLinear),
        ("ss", C.c_void_p), ("c_time", C.c_void_p), ("t_out", C.c_void_p), ("silu_out", C.c_void_p),
    ]


class MiAttnFoldBlk(C.Structure):
    _fields_ = [("mg", C.c_void_p), ("mv", C.c_void_p), ("g0", C.c_void_p), ("v0", C.c_void_p), ("gv", C.c_void_p), ("table", C.c_void_p),
                ("g_exp", C.c_int), ("v_exp", C.c_int)]


class MiAttnFoldParams(C.Structure):
    _fields_ = [
        ("B2", C.c_int), ("C", C.c_int), ("cd", C.c_int), ("heads", C.c_int), ("JT", C.c_int),
        ("c_rows", C.c_void_p), ("c_stride_b", C.c_int), ("row0", C.c_int), ("nrows", C.c_int), ("write_null", C.c_int), ("frag_f16", C.c_int),
        ("n_blocks", C.c_int), ("blk", MiAttnFoldBlk * 8),
        ("mode", C.c_int), ("t_state", C.c_void_p), ("t_off", C.c_int), ("ss_all", C.c_void_p), ("ss", C.c_void_p), ("ss_n", C.c_int),
    ]


class MiCrossAttnParams(C.Structure):
    _fields_ = [
        ("B2", C.c_int), ("C", C.c_int), ("HW", C.c_int), ("heads", C.c_int), ("J", C.c_int),
        ("x", MiAct), ("gv", C.c_void_p), ("n1_g", C.c_void_p), ("n1_b", C.c_void_p), ("n2_g", C.c_void_p), ("n2_b", C.c_void_p),
        ("out", C.c_void_p), ("out_stats", C.c_void_p), ("out_st", C.c_int), ("x_exp", C.c_int), ("g_exp", C.c_int), ("v_exp", C.c_int), ("variant", C.c_int),
    ]


class MiSelfAttnParams(C.Structure):
    _fields_ = [("B2", C.c_int), ("C", C.c_int), ("HW", C.c_int), ("heads", C.c_int), ("J", C.c_int), ("x", MiAct), ("gv", C.c_void_p),
                ("n1_g", C.c_void_p), ("n1_b", C.c_void_p), ("n2_g", C.c_void_p), ("n2_b", C.c_void_p), ("out", C.c_void_p), ("out_stats", C.c_void_p)]


class MiChanFFParams(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("Chid", C.c_int), ("HW", C.c_int), ("x", MiAct), ("g1", C.c_void_p), ("w1", C.c_void_p),
                ("g2", C.c_void_p), ("w2", C.c_void_p), ("out", C.c_void_p), ("out_stats", C.c_void_p)]


class MiFlashAttnParams(C.Structure):
    _fields_ = [("B", C.c_int), ("HW", C.c_int), ("heads", C.c_int), ("kv_heads", C.c_int), ("q", C.c_void_p), ("q_scale", C.c_float),
                ("null_k", C.c_void_p), ("null_v", C.c_void_p),
                ("k0", C.c_void_p), ("v0", C.c_void_p), ("n0", C.c_int), ("ld0", C.c_int), ("bs0", C.c_longlong),
                ("k1", C.c_void_p), ("v1", C.c_void_p), ("n1", C.c_int), ("ld1", C.c_int), ("bs1", C.c_longlong), ("out", C.c_void_p),
                ("kv_prep", C.c_void_p), ("kv_prep_bytes", C.c_longlong)]


class MiTokensToNchwParams(C.Structure):
    _fields_ = [("B", C.c_int), ("HW", C.c_int), ("C", C.c_int), ("tokens", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("eps", C.c_float), ("res", MiAct), ("out", C.c_void_p), ("out_stats", C.c_void_p)]


class MiCfgX0Params(C.Structure):
    _fields_ = [("B", C.c_int), ("n", C.c_int), ("pred2", C.c_void_p), ("two", C.c_int), ("cond_scale", C.c_float),
                ("x_t", C.c_void_p), ("coef", C.c_void_p), ("t_state", C.c_void_p), ("pred_out", C.c_void_p), ("x0", C.c_void_p),
                ("hist0", C.c_void_p), ("t_off", C.c_int)]


class MiQuantileParams(C.Structure):
    _fields_ = [("B", C.c_int), ("n", C.c_int), ("x0", C.c_void_p), ("k_lo", C.c_int), ("k_hi", C.c_int), ("w", C.c_float),
                ("hist", C.c_void_p), ("s_out", C.c_void_p), ("v_out", C.c_void_p), ("pass0_done", C.c_int), ("self_cleaning", C.c_int)]


class MiPosteriorParams(C.Structure):
    _fields_ = [("B", C.c_int), ("n", C.c_int), ("T", C.c_int), ("x0", C.c_void_p), ("s_q", C.c_void_p), ("x", C.c_void_p),
                ("coef", C.c_void_p), ("t_state", C.c_void_p), ("noise", C.c_void_p),
                ("seed", C.c_uint64), ("sample0", C.c_int), ("stream_base", C.c_int), ("seed_dev", C.c_void_p), ("t_off", C.c_int)]


class MiResizeParams(C.Structure):
    _fields_ = [("planes", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int), ("KH", C.c_int), ("KW", C.c_int),
                ("in_", C.c_void_p), ("out", C.c_void_p), ("idx_h", C.c_void_p), ("w_h", C.c_void_p), ("idx_w", C.c_void_p), ("w_w", C.c_void_p)]


class MiConvWgradParams(C.Structure):
    _fields_ = [("B", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("H", C.c_int), ("W", C.c_int), ("a", C.c_void_p), ("dy", C.c_void_p),
                ("dw", C.c_void_p), ("db", C.c_void_p), ("partial", C.c_void_p), ("nwg", C.c_int),
                ("a_stats", C.c_void_p), ("a_nt", C.c_int), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("groups", C.c_int), ("eps", C.c_float),
                ("ss", C.c_void_p), ("ss_stride", C.c_int), ("ss_off", C.c_int)]


class MiBlockBwdParams(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("HW", C.c_int), ("groups", C.c_int), ("nt", C.c_int), ("nchunk", C.c_int), ("eps", C.c_float),
                ("x", C.c_void_p), ("da", C.c_void_p), ("x_stats", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("ss", C.c_void_p), ("ss_stride", C.c_int), ("ss_off", C.c_int), ("uv", C.c_void_p), ("dx", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dss", C.c_void_p), ("dx_stats", C.c_void_p)]


class MiCrossEmbedWgradParams(C.Structure):
    _fields_ = [("B", C.c_int), ("Cin", C.c_int), ("H", C.c_int), ("W", C.c_int), ("x", C.c_void_p), ("dy", C.c_void_p),
                ("n_kernels", C.c_int), ("ksize", C.c_int * 3), ("cout", C.c_int * 3), ("dw", C.c_void_p * 3), ("db", C.c_void_p * 3),
                ("partial", C.c_void_p), ("nwg", C.c_int)]


class MiFoldedAttnParams(C.Structure):
    _fields_ = [("B", C.c_int), ("n", C.c_int), ("H", C.c_int), ("J", C.c_int), ("C", C.c_int), ("nchunk", C.c_int), ("q", C.c_void_p),
                ("kf", C.c_void_p), ("vf", C.c_void_p), ("mask", C.c_void_p), ("out", C.c_void_p), ("lse", C.c_void_p), ("dout", C.c_void_p),
                ("dsum", C.c_void_p), ("dq", C.c_void_p), ("dkf", C.c_void_p), ("dvf", C.c_void_p), ("oh", C.c_void_p)]


class MiAdamTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_longlong)]


class MiAdamParams(C.Structure):
    _fields_ = [("tensors", C.c_void_p), ("chunk_tensor", C.c_void_p), ("chunk_off", C.c_void_p), ("nchunks", C.c_int), ("chunk", C.c_int),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
                ("bias_correction1", C.c_float), ("bias_correction2", C.c_float), ("one_minus_beta1", C.c_float), ("one_minus_beta2", C.c_float),
                ("grad_scale", C.c_void_p)]


class MiPackConv3Desc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("frag", C.c_void_p), ("generic", C.c_void_p), ("Cout", C.c_int), ("Cin", C.c_int), ("adjoint", C.c_int),
                ("exp", C.c_int), ("cout_pad", C.c_int), ("reserved", C.c_int)]


_STRUCTS = {0: MiAct, 1: MiConvParams, 2: MiCrossEmbedParams, 3: MiLinear, 4: MiTextCondParams, 5: MiCondStepParams,
            6: MiAttnFoldParams, 7: MiCrossAttnParams, 8: MiCfgX0Params, 9: MiQuantileParams, 10: MiPosteriorParams,
            11: MiResizeParams, 12: MiSelfAttnParams, 13: MiChanFFParams, 14: MiFlashAttnParams, 15: MiTokensToNchwParams, 16: MiConvWgradParams, 17: MiBlockBwdParams, 18: MiCrossEmbedWgradParams, 19: MiFoldedAttnParams, 20: MiAdamTensor, 21: MiAdamParams, 22: MiPackConv3Desc}

_lib = None
_backend = None


class MinImagenHipError(RuntimeError):
    pass


def _bind(lib):
    lib.mi_abi_version.restype = C.c_int
    lib.mi_last_error.restype = C.c_char_p
    lib.mi_backend.restype = C.c_char_p
    lib.mi_struct_size.argtypes = [C.c_int]
    vp, i32, i64, u64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float
    for name in ("mi_conv_fwd", "mi_gn_coef_fwd", "mi_crossembed_fwd", "mi_text_cond_fwd", "mi_cond_step_fwd", "mi_attn_fold_rows", "mi_cross_attn_fwd",
                 "mi_cfg_x0_fwd", "mi_quantile_fwd", "mi_posterior_fwd", "mi_resize_fwd", "mi_self_attn_fwd", "mi_chan_ff_fwd",
                 "mi_flash_attn_fwd", "mi_conv_prep_fwd", "mi_tokens_to_nchw_fwd", "mi_conv_wgrad", "mi_block_bwd", "mi_crossembed_wgrad", "mi_folded_attn_fwd", "mi_folded_attn_bwd", "mi_adam_step"):
        getattr(lib, name).argtypes = [vp, vp]
        getattr(lib, name).restype = i32
    lib.mi_conv_prep_bytes.argtypes = [i32, i32, i32, i32, i32]
    lib.mi_conv_prep_bytes.restype = C.c_longlong
    lib.mi_flash_kv_prep_bytes.argtypes = [i32, i32]
    lib.mi_flash_kv_prep_bytes.restype = C.c_longlong
    lib.mi_step_advance.argtypes = [vp, vp, i32, vp]
    lib.mi_step_advance_by.argtypes = [vp, vp, i32, i32, vp]
    lib.mi_sampler_step_small_fwd.argtypes = [vp, vp, vp, vp]
    lib.mi_sampler_step_small_fwd.restype = i32
    lib.mi_sampler_step_group_fwd.argtypes = [vp, vp, vp, vp, vp]
    lib.mi_sampler_step_group_fwd.restype = i32
    lib.mi_sampler_group_size.argtypes = [i32]
    lib.mi_sampler_group_sync_bytes.argtypes = [i32, i32]
    lib.mi_sampler_group_sync_bytes.restype = C.c_longlong
    lib.mi_step_set.argtypes = [vp, vp, i32, i32, vp]
    lib.mi_randn_fill.argtypes = [vp, i32, i32, u64, i32, i32, vp]
    lib.mi_finalize_images.argtypes = [vp, vp, i64, i32, vp]
    lib.mi_lowres_augment.argtypes = [vp, vp, vp, i64, f32, f32, i32, vp]
    lib.mi_gemm_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.mi_rmsnorm.argtypes = [vp, vp, vp, i32, i32, f32, vp, vp]
    lib.mi_embed_rows.argtypes = [vp, vp, vp, i32, i32, vp]
    lib.mi_embed_rows_sq.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    lib.mi_gemm_rms_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, f32, vp, vp]
    lib.mi_t5_attention.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    lib.mi_ln_tokens_fwd.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    lib.mi_ln_rows_fwd.argtypes = [vp, vp, vp, vp, i32, i32, f32, vp]
    lib.mi_graph_begin.argtypes = [vp]
    lib.mi_graph_end.argtypes = [vp, C.POINTER(vp)]
    lib.mi_graph_launch.argtypes = [vp, vp]
    lib.mi_graph_destroy.argtypes = [vp]
    lib.mi_conv_tile_shape.argtypes = [i32, C.POINTER(i32), C.POINTER(i32)]
    lib.mi_conv_stripe_rows.argtypes = [vp]
    lib.mi_conv_stripe_rows.restype = i32
    lib.mi_conv_cout_tile.argtypes = [i32]
    lib.mi_conv_wgrad_workspace.argtypes = [i32, i32, i32]
    lib.mi_conv_wgrad_workspace.restype = C.c_longlong
    lib.mi_crossembed_wgrad_workspace.argtypes = [i32, i32, i32]
    lib.mi_crossembed_wgrad_workspace.restype = C.c_longlong
    lib.mi_chan_stats_fwd.argtypes = [vp, vp, i32, i32, vp]
    lib.mi_layernorm_fwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, C.c_float, vp]
    lib.mi_layernorm_bwd_nwg.argtypes = [i32, i32]
    lib.mi_layernorm_bwd_nwg.restype = i32
    lib.mi_layernorm_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.mi_pack_conv3_floats.argtypes = [i32, i32, i32, i32, i32]
    lib.mi_pack_conv3_floats.restype = C.c_longlong
    lib.mi_pack_conv3.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, vp]
    lib.mi_pack_conv3_multi.argtypes = [vp, i32, i32, vp]
    lib.mi_attn_fragment_floats.argtypes = [i32]
    for which, st in _STRUCTS.items():
        n = lib.mi_struct_size(which)
        if n != C.sizeof(st):
            raise MinImagenHipError(f"ABI mismatch: struct {st.__name__} is {n} bytes in the library, {C.sizeof(st)} in the binding")
    return lib


def use_library(path: str):
    """Load a specific build of the C ABI (tests use this for the emulator build)."""
    global _lib, _backend
    if not os.path.exists(path):
        raise MinImagenHipError(f"{path} not found -- build it first (python -c 'import __graft_entry__ as g; g.build()')")
    _lib = _bind(C.CDLL(path))
    _backend = _lib.mi_backend().decode()
    return _lib


def lib():
    global _lib
    if _lib is None:
        use_library(DEFAULT_LIB)
    return _lib


def backend() -> str:
    lib()
    return _backend


def require_device(*tensors):
    """Every tensor handed to a kernel must live where the loaded backend computes (the GPU)."""
    be = backend()
    for t in tensors:
        if t is None:
            continue
        if t.dtype not in (torch.float32, torch.int64, torch.int32, torch.uint8, torch.bool, torch.float64):
            raise MinImagenHipError(f"unsupported dtype {t.dtype}")
        if not t.is_contiguous():
            raise MinImagenHipError("tensor must be contiguous")
        if be == "hip-gfx950" and not t.is_cuda:
            raise MinImagenHipError("minimagen_amd has no CPU path: tensor is not on the GPU")
        if be == "hipemu" and t.is_cuda:
            raise MinImagenHipError("emulator build needs host tensors")


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream() -> int:
    """the raw HIP stream of torch's current stream on the current device (one C call: this is asked once per launch, ~1 000 times per training step)"""
    if backend() == "hip-gfx950":
        if _RAW_STREAM is not None:
            return _RAW_STREAM(torch.cuda.current_device())
        return torch.cuda.current_stream().cuda_stream
    return 0


def check(rc: int, what: str = ""):
    if rc != 0:
        raise MinImagenHipError(f"{what} failed ({rc}): {lib().mi_last_error().decode()}")
