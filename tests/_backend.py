"""Select where the C ABI runs for a test: the real gfx950 library on a GPU ("gpu", marked @pytest.mark.gpu)
or the dev-only SIMT emulator build of the same kernel sources on the host ("emu")."""
import os
import subprocess

import pytest
import torch

from minimagen_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_LIB = os.path.join(ROOT, "tools", "hipemu", "build", "libminimagen_emu.so")

BACKENDS = [pytest.param("gpu", marks=pytest.mark.gpu), pytest.param("emu", marks=pytest.mark.emu)]
GPU_ONLY = [pytest.param("gpu", marks=pytest.mark.gpu)]


_EMU_BUILT = False


def setup(kind: str) -> torch.device:
    if kind == "gpu":
        assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
        L.use_library(L.DEFAULT_LIB)
        assert L.backend() == "hip-gfx950"
        return torch.device("cuda:0")
    global _EMU_BUILT
    if not _EMU_BUILT:            # once per test process: a no-op when up to date, a rebuild when a kernel source changed
        subprocess.run(["make", "-j8", "-C", os.path.join(ROOT, "tools", "hipemu")], check=True, stdout=subprocess.DEVNULL)
        _EMU_BUILT = True
    L.use_library(EMU_LIB)
    assert L.backend() == "hipemu"
    return torch.device("cpu")
