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
DEFAULT_LIB = os.path.join(_HERE, "libminimagen_hip.so")

c_float_p = C.c_void_p   # raw device pointers travel as integers


class MiAct(C.Structure):
    _fields_ = [("data", C.c_void_p), ("C", C.c_int), ("stats", C.c_void_p), ("nt", C.c_int), ("scale", C.c_float)]


class MiConvParams(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("in0", MiAct), ("in1", MiAct),
        ("Cout", C.c_int), ("ksize", C.c_int), ("stride", C.c_int), ("up2", C.c_int),
        ("w", C.c_void_p), ("bias", C.c_void_p),
        ("gn_groups", C.c_int), ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p), ("gn_eps", C.c_float),
        ("scale_shift", C.c_void_p), ("ss_stride", C.c_int), ("ss_off", C.c_int),
        ("res0", MiAct), ("res1", MiAct), ("res_w", C.c_void_p), ("res_b", C.c_void_p),
        ("out", C.c_void_p), ("out_stats", C.c_void_p), ("tile_cfg", C.c_int),
    ]


class MiCrossEmbedParams(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("in0", C.c_void_p), ("C0", C.c_int), ("in1", C.c_void_p), ("C1", C.c_int),
        ("in1_batch_mod", C.c_int), ("in0_batch_mod", C.c_int), ("n_kernels", C.c_int),
        ("ksize", C.c_int * 3), ("cout", C.c_int * 3), ("w", C.c_void_p * 3), ("bias", C.c_void_p * 3),
        ("out", C.c_void_p), ("out_stats", C.c_void_p), ("tile_cfg", C.c_int),
    ]


_STRUCTS = {0: MiAct, 1: MiConvParams, 2: MiCrossEmbedParams}

_lib = None
_backend = None


class MinImagenHipError(RuntimeError):
    pass


def _bind(lib):
    lib.mi_abi_version.restype = C.c_int
    lib.mi_last_error.restype = C.c_char_p
    lib.mi_backend.restype = C.c_char_p
    lib.mi_struct_size.argtypes = [C.c_int]
    for name in dir(lib.__class__):
        pass
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


def current_stream() -> int:
    if backend() == "hip-gfx950":
        return torch.cuda.current_stream().cuda_stream
    return 0


def check(rc: int, what: str = ""):
    if rc != 0:
        raise MinImagenHipError(f"{what} failed ({rc}): {lib().mi_last_error().decode()}")
