"""Disassembles every gfx950 code object of a built libminimagen_hip.so and rejects packed-fp32 VALU instructions with a scalar operand
(profiles/r03_pk_f32_hazard.txt, tools/ubench/pk_f32_hazard.hip).  Used by __graft_entry__.build() and tests/test_host.py."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def check_library(so_path: str, workdir: str = None):
    """-> (number of gfx950 code objects, number of packed-fp32 instructions); raises AssertionError on an SGPR-operand packed instruction"""
    own = workdir is None
    workdir = tempfile.mkdtemp(prefix="mi_codecheck_") if own else workdir
    try:
        so = os.path.join(workdir, "lib.so")
        shutil.copy(so_path, so)
        subprocess.run([OBJDUMP, "--offloading", so], cwd=workdir, check=True, capture_output=True)
        bundles = sorted(f for f in os.listdir(workdir) if "gfx950" in f)
        n_pk = 0
        for f in bundles:
            asm = subprocess.run([OBJDUMP, "-d", os.path.join(workdir, f)], capture_output=True, text=True, check=True).stdout
            pk = [ln for ln in asm.splitlines() if re.search(r"\bv_pk_(mul|add|fma)_f32\b", ln)]
            n_pk += len(pk)
            bad = [ln for ln in pk if re.search(r"\bs\[\d+:\d+\]|\bs\d+\b", ln.split("//")[0])]
            assert not bad, f"{f}: packed fp32 instruction with a scalar operand: {bad[0].strip()}"
        return len(bundles), n_pk
    finally:
        if own:
            shutil.rmtree(workdir, ignore_errors=True)


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, k = check_library(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "minimagen_amd", "libminimagen_hip.so"))
    print(f"{n} gfx950 code objects, {k} packed-fp32 instructions, none with a scalar operand")
