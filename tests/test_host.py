"""Host logic that needs no kernel: tap tables, rank arithmetic, sharding + gather (gloo, world_size 2),
and that the product library loads and exports every symbol declared in include/minimagen_hip.h."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

from minimagen_amd import _lib as L
from minimagen_amd.distributed import shard_bounds
from minimagen_amd.helpers import cubic_taps, quantile_rank
from oracle import resize_restated as RR
from oracle import restated as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    so = L.DEFAULT_LIB
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(so)           # loads without a GPU; no compute call is made
    hdr = open(os.path.join(ROOT, "include", "minimagen_hip.h")).read()
    names = sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in minimagen_hip.h but not exported"
    lib.mi_backend.restype = ctypes.c_char_p
    assert lib.mi_backend() == b"hip-gfx950" and lib.mi_abi_version() == 12


def test_struct_layouts_match_the_library():
    L.use_library(L.DEFAULT_LIB)    # _bind() compares sizeof() of every parameter struct


def test_integration_md_binding_example_matches_the_abi():
    """INTEGRATION.md shows the ctypes binding a maintainer would write: the struct it spells out must have the library's layout (a
    binder copying the document must pass mi_struct_size) and the ABI version it quotes must be the header's"""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class MiAct\(C\.Structure\):[^\n]*\n\s*_fields_ = (\[.*?\])\n", doc, re.S)
    assert m, "INTEGRATION.md no longer shows the MiAct binding"
    fields = eval(m.group(1), {"C": ctypes})
    doc_act = type("DocMiAct", (ctypes.Structure,), {"_fields_": fields})
    lib = ctypes.CDLL(L.DEFAULT_LIB)
    lib.mi_struct_size.argtypes = [ctypes.c_int]
    assert ctypes.sizeof(doc_act) == lib.mi_struct_size(0)
    assert [(n, t) for n, t in fields] == [(n, t) for n, t in L.MiAct._fields_]
    nargs = len(fields)
    for call in re.findall(r"MiAct\(([^#\n]*?)\)\s", doc):
        depth, count = 0, 1
        for ch in call:
            depth += ch in "([" 
            depth -= ch in ")]"
            count += (ch == "," and depth == 0)
        assert count == nargs, f"MiAct({call}) in INTEGRATION.md has {count} arguments, the struct has {nargs} fields"
    ver = int(re.search(r"#define MI_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "minimagen_hip.h")).read()).group(1))
    assert f"(currently {ver})" in doc and lib.mi_abi_version() == ver


def test_no_packed_fp32_with_scalar_operands_in_the_code_objects(tmp_path):
    """Round 3 (profiles/r03_pk_f32_hazard.txt; stand-alone reproducer: tools/ubench/pk_f32_hazard.hip, profiles/r04_pk_f32_hazard_ubench.txt):
    a packed-fp32 VALU instruction with an SGPR operand returns wrong products for a 16-lane pass while matrix-core waves of ANOTHER
    stream's kernel share its SIMD -- and the stage-pipelined sampler runs two streams side by side.  The library is built with
    -fno-slp-vectorize; the hand-placed packed operations (attention softmax) use VGPR operands only.  Disassemble every gfx950 code
    object of the built library and check (the same check runs inside __graft_entry__.build() and, below, in the GPU tier)."""
    from tools.check_code_objects import check_library, OBJDUMP
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    n_objects, n_pk = check_library(L.DEFAULT_LIB, str(tmp_path))
    assert n_objects >= 6 and n_pk < 2000


@pytest.mark.gpu
def test_no_packed_fp32_with_scalar_operands_in_the_library_the_gpu_tests_load(tmp_path):
    """the same disassembly check on the GPU box, on the very .so the -m gpu tests load"""
    from tools.check_code_objects import check_library, OBJDUMP
    assert os.path.exists(OBJDUMP), "the ROCm image ships llvm-objdump"
    n_objects, n_pk = check_library(L.DEFAULT_LIB, str(tmp_path))
    assert n_objects >= 6 and n_pk < 2000


def test_cubic_taps_match_oracle_restatement():
    for in_sz, out_sz in ((64, 256), (64, 128), (256, 1024), (16, 64)):
        osz, idx, w = cubic_taps(in_sz, out_sz)
        o2, pad, fov, w2 = RR.taps_for_dim(in_sz, out_sz / in_sz)
        assert osz == o2 == out_sz and torch.equal(w, w2)
        ref_idx = fov - pad[0]
        ref_idx = torch.where(ref_idx < 0, -ref_idx, ref_idx)
        ref_idx = torch.where(ref_idx >= in_sz, 2 * (in_sz - 1) - ref_idx, ref_idx)
        assert torch.equal(idx.long(), ref_idx)
    assert cubic_taps(64, 256)[2].shape == (256, 4)


def test_quantile_rank_is_torch_fp32_arithmetic():
    assert quantile_rank(12288, 0.9) == (11058, 11059, 0.2998046875)
    assert quantile_rank(196608, 0.9) == (176946, 176947, 0.296875)
    for n in (48, 3145728, 12288):
        lo, hi, w = quantile_rank(n, 0.9)
        rl, rw = R.quantile_rank(n, 0.9)
        assert (lo, w) == (rl, float(rw)) and hi == lo + 1


def test_shard_bounds_cover_the_batch():
    for B in (1, 7, 32, 128):
        for ws in (1, 2, 4, 8):
            spans = [shard_bounds(B, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from minimagen_amd.distributed import shard_bounds, gather_samples
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=2)
for B in (5, 4, 1):                          # ragged shards, even shards (one collective straight into the result), a rank with NO rows
    full = torch.arange(B * 3 * 4 * 4, dtype=torch.float32).reshape(B, 3, 4, 4)
    lo, hi = shard_bounds(B, 2, dist.get_rank())
    out = gather_samples(full[lo:hi].clone(), B)
    assert torch.equal(out, full), f"gathered batch differs (B={B})"
dist.destroy_process_group()
print("ok")
'''


def test_gather_samples_world_size_2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


_SAMPLE_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from tests._backend import setup
from minimagen_amd.Imagen import Imagen
from minimagen_amd.Unet import Unet
from minimagen_amd.distributed import sample_distributed
from oracle import restated as R
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=2)
setup("emu")                                 # the kernels run through the SIMT emulator in the GPU-less container
torch.manual_seed(4)                         # same weights on both ranks (the product loads the same state_dict)
u = Unet(dim=8, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=False, layer_cross_attns=False, memory_efficient=True)
im = Imagen([u], text_encoder_name="t5_small", image_sizes=[16], timesteps=21, cond_drop_prob=0.15)
B = 3                                        # uneven shards: rank 0 samples rows 0-1, rank 1 row 2
emb, mask = R.synthetic_text(B, length=9, seed=3)
out = sample_distributed(im, text_embeds=emb, text_masks=mask, cond_scale=1., _seed=77)
assert out.shape == (B, 3, 16, 16)
if dist.get_rank() == 0:
    whole = im.sample(text_embeds=emb, text_masks=mask, cond_scale=1., _seed=77)
    assert torch.equal(out, whole), "2-rank sharded sampling differs from the single-process batch"
# more ranks than samples: rank 1 has an empty shard and must still take part in the collective (no hang, no raise)
one = sample_distributed(im, text_embeds=emb[:1], text_masks=mask[:1], cond_scale=1., _seed=77)
assert one.shape == (1, 3, 16, 16)
if dist.get_rank() == 0:
    assert torch.equal(one, whole[:1])
dist.barrier()
dist.destroy_process_group()
print("ok")
'''


def test_sample_distributed_world_size_2_gloo(tmp_path):
    """SURVEY.md 8(e) end to end on CPU: two ranks shard the batch, sample their rows (noise keyed by the global row), all_gather;
    the result equals the single-process batch bit for bit"""
    script = tmp_path / "ws.py"
    script.write_text(_SAMPLE_WORKER)
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_bench_contract_helpers():
    """bench.py pieces that need no GPU: the CLI parses, the synthetic text follows SURVEY.md 8(d) (ragged masks, masked rows zeroed,
    rows keyed by the global index), and the roofline's PMC traffic is found in the committed profile of the dominant launch shape"""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout
    emb, mask = bench.synthetic_text(30, length=64)
    assert emb.shape == (30, 64, 512) and mask.dtype == torch.bool
    assert [int(m.sum()) for m in mask[:3]] == [64, 63, 62] and int(mask[24].sum()) == 64           # L - (r mod 24)
    assert float(emb[1, 63].abs().max()) == 0.0 and float(emb[1, 62].abs().max()) > 0.0
    emb2, mask2 = bench.synthetic_text(4, length=64, row0=26)                                      # rank offset = global rows 26..29
    assert torch.equal(emb2, emb[26:30]) and torch.equal(mask2, mask[26:30])
    t = bench.pmc_traffic({"kernel": "cross_attn"}, 64, 256)
    assert t["traffic"] is None or 3e7 < t["traffic"] < 1e8


def test_minimagen_import_path_is_the_amd_implementation():
    """the reference's callers import ``minimagen.*`` (reference inference.py:2, generate.py:8-9): the alias package must resolve every
    one of those names to the MI355X implementation (same objects, not copies)"""
    import minimagen
    import minimagen_amd.Imagen, minimagen_amd.Unet, minimagen_amd.generate, minimagen_amd.t5, minimagen_amd.diffusion_model
    from minimagen.Imagen import Imagen
    from minimagen.Unet import Unet, Base, Super, BaseTest, SuperTest
    from minimagen.generate import load_minimagen, load_params, sample_and_save
    from minimagen.t5 import t5_encode_text, get_encoded_dim
    from minimagen.diffusion_model import GaussianDiffusion
    from minimagen.helpers import cast_tuple, default, exists
    assert Imagen is minimagen_amd.Imagen.Imagen and Unet is minimagen_amd.Unet.Unet and Super is minimagen_amd.Unet.Super
    assert load_minimagen is minimagen_amd.generate.load_minimagen and GaussianDiffusion is minimagen_amd.diffusion_model.GaussianDiffusion
    assert minimagen.Imagen is minimagen_amd.Imagen and get_encoded_dim("t5_small") == 512
    # a reference-style construction through the alias path (reference README / main.py usage)
    u = BaseTest.defaults
    im = Imagen(unets=(Unet(**BaseTest.defaults), Unet(**SuperTest.defaults)), image_sizes=(32, 64), timesteps=25, text_encoder_name="t5_small")
    assert len(im.unets) == 2 and im.unets[1].lowres_cond and u["dim"] == 8


def test_gaussian_diffusion_public_methods():
    """q_sample / q_posterior / predict_start_from_noise (diffusion_model.py:89-162) on the bit-identical tables: against the live
    reference when it is present, always against the oracle's scalar forms"""
    from minimagen_amd.diffusion_model import GaussianDiffusion
    T = 100
    gd = GaussianDiffusion(timesteps=T)
    g = torch.Generator().manual_seed(3)
    x0, eps = torch.randn(4, 3, 8, 8, generator=g), torch.randn(4, 3, 8, 8, generator=g)
    t = torch.tensor([0, 1, 57, 99])
    xt = gd.q_sample(x0, t, eps)
    mean, var, logvar = gd.q_posterior(x0, xt, t)
    x0_hat = gd.predict_start_from_noise(xt, t, eps)
    assert mean.shape == x0.shape and var.shape == logvar.shape == (4, 1, 1, 1)
    sched = R.Schedule(T)
    for i, ti in enumerate(t.tolist()):
        assert torch.equal(xt[i], sched.q_sample(x0[i:i + 1], ti, eps[i:i + 1])[0])
        assert torch.equal(x0_hat[i], sched.predict_start_from_noise(xt[i:i + 1], ti, eps[i:i + 1])[0])
        m_ref = sched.q_posterior(x0[i:i + 1], xt[i:i + 1], ti)
        assert torch.equal(mean[i], m_ref[0][0])
    assert gd.q_sample(x0, t).shape == x0.shape                     # noise defaults to randn_like
    if os.path.isdir("/root/reference/minimagen"):
        from oracle import ref_loader
        ref = ref_loader.load_reference()
        rgd = ref.diffusion_model.GaussianDiffusion(timesteps=T)
        assert torch.equal(xt, rgd.q_sample(x0, t, eps)) and torch.equal(x0_hat, rgd.predict_start_from_noise(xt, t, eps))
        for a, b in zip((mean, var, logvar), rgd.q_posterior(x0, xt, t)):
            assert torch.equal(a, b)
