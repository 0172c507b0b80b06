"""Disassembles every gfx950 code object of a built libminimagen_hip.so and rejects packed-fp32 VALU instructions with a scalar operand
(profiles/r03_pk_f32_hazard.txt, tools/ubench/pk_f32_hazard.hip); reads the kernels' metadata and rejects SCRATCH (register spills, by-value
structs indexed per lane) in any kernel of the default sampling path -- a spill reload's s_waitcnt also waits for every prefetch in flight
(profiles/r05_summary.md: 9.5 % on the SR U-Net's 64^2 level).  Used by __graft_entry__.build() and tests/test_host.py."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"

# kernels of the default fp32 sampling path of the BASELINE U-Nets (engine launch plan + sampler tail): no scratch allowed.  The stripe conv family
# and the folded attention are listed whole; of the tile conv family the compile-time-round members (what the BASELINE layers instantiate)
DEFAULT_PATH = (r"conv_stripe_kernel", r"cross_attn_f16x3_kernelILi(8|16)E", r"crossembed_mfma_kernel", r"sampler_small_kernel", r"sampler_group_kernel", r"cfg_x0_kernel",
                r"posterior_kernel", r"quantile_hist_kernel", r"quantile_finish_kernel", r"cond_step_kernel", r"text_cond_kernel", r"attn_fold_rows_kernel",
                r"conv_rp_kernelINS_5RpCfgILi8ELi(64|32)ELi[12]ELb[01]ELb0E(Li0ELi[1-4]ELi[0-4]|Li1ELi[12]ELi0|Li2ELi1ELi0)ELb0E")      # k3 s1 | nearest x2 | k4 s2 (8 input channels)


def kernel_scratch(code_object: str):
    """{kernel name: scratch bytes per work-item} from the code object's metadata note"""
    txt = subprocess.run([READELF, "--notes", code_object], capture_output=True, text=True, check=True).stdout
    out, name = {}, None
    for ln in txt.splitlines():
        m = re.search(r"\.name:\s+(\S+)", ln)
        if m:
            name = m.group(1)
        m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", ln)
        if m and name:
            out[name] = int(m.group(1))
    return out


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
            if os.path.exists(READELF):
                for name, scratch in kernel_scratch(os.path.join(workdir, f)).items():
                    if scratch and any(re.search(pat, name) for pat in DEFAULT_PATH):
                        raise AssertionError(f"{f}: default-path kernel {name} uses {scratch} bytes of scratch per work-item")
        return len(bundles), n_pk
    finally:
        if own:
            shutil.rmtree(workdir, ignore_errors=True)


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, k = check_library(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "minimagen_amd", "libminimagen_hip.so"))
    print(f"{n} gfx950 code objects, {k} packed-fp32 instructions, none with a scalar operand; no scratch in the default-path kernels")
