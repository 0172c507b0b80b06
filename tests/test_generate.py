"""the training-directory caller of Imagen.sample (minimagen/generate.py) -- parameter files,
checkpoint selection (state_dicts first, tmp as fallback), output layout and the PIL conversion."""
import json
import os

import numpy as np
import pytest
import torch

from minimagen_amd import Imagen as imagen_module
from minimagen_amd import generate as G
from oracle import restated as R
from tests import _inputs as I
from tests._backend import BACKENDS, setup

IMAGEN_PARAMS = {"text_embed_dim": None, "channels": 3, "timesteps": 25, "cond_drop_prob": 0.15, "loss_type": "l2",
                 "lowres_sample_noise_level": 0.2, "auto_normalize_img": True, "dynamic_thresholding_percentile": 0.9,
                 "only_train_unet_number": None, "image_sizes": [64], "text_encoder_name": "t5_small"}


def make_training_dir(root, where="state_dicts", image_size=64):
    for sub in ("parameters", "state_dicts", "tmp"):
        os.makedirs(os.path.join(root, sub))
    stamp = "20220816_165729"
    json.dump(I.unet_params()["unet0"], open(os.path.join(root, "parameters", f"unet_0_params_{stamp}.json"), "w"))
    json.dump(dict(IMAGEN_PARAMS, image_sizes=[image_size]), open(os.path.join(root, "parameters", f"imagen_params_{stamp}.json"), "w"))
    open(os.path.join(root, "parameters", f"training_parameters_{stamp}.txt"), "w").write("--BATCH_SIZE=2\n")
    if where == "state_dicts":
        torch.save(I.load("unet0_sd.pt"), os.path.join(root, "state_dicts", "unet_0_state_0_2_0.512.pth"))
    elif where == "tmp":
        torch.save(I.load("unet0_sd.pt"), os.path.join(root, "tmp", "unet_0_tmp.pth"))
    return str(root)


def test_load_params_and_checkpoint_selection(tmp_path):
    setup("emu")
    d = make_training_dir(tmp_path / "a")
    unets, imagen = G.load_params(d)
    assert unets == [I.unet_params()["unet0"]] and imagen == IMAGEN_PARAMS
    m = G.load_minimagen(d)
    sd = I.load("unet0_sd.pt")
    assert all(torch.equal(v, sd[k]) for k, v in m.unets[0].state_dict().items())
    assert m.noise_schedulers[0].num_timesteps == 25 and list(m.image_sizes) == [64]
    m2 = G.load_minimagen(make_training_dir(tmp_path / "b", where="tmp"))           # generate.py:105-119
    assert all(torch.equal(v, sd[k]) for k, v in m2.unets[0].state_dict().items())
    with pytest.raises(ValueError):
        G.load_minimagen(make_training_dir(tmp_path / "c", where="nowhere"))         # generate.py:110-111


@pytest.mark.parametrize("backend", BACKENDS)
def test_sample_and_save_layout_and_pixels(backend, tmp_path, monkeypatch):
    dev = setup(backend)
    size = 64 if backend == "gpu" else 32                                           # the emulator is slow: one caption and a 32 x 32 image there
    d = make_training_dir(tmp_path / "train", image_size=size)
    captions = ["a happy dog", "a blue house"][:2 if backend == "gpu" else 1]
    n = len(captions)
    emb, mask = R.synthetic_text(n, length=8, seed=3)
    monkeypatch.setattr(imagen_module, "t5_encode_text", lambda texts, name=None: (emb.clone(), mask.clone()))   # no T5 files offline
    out = tmp_path / "out"
    G.sample_and_save(captions, training_directory=d, save_directory=str(out), sample_args={"cond_scale": 1.})
    assert open(out / "captions.txt").read() == "".join(c + "\n" for c in captions)
    assert open(out / "imagen_training_directory.txt").read() == d
    assert sorted(os.listdir(out / "generated_images")) == [f"image_{i}.png" for i in range(n)]
    from PIL import Image
    m = G.load_minimagen(d).to(dev)
    ref = m.sample(text_embeds=emb.to(dev), text_masks=mask.to(dev), cond_scale=1.).cpu()
    for i in range(n):
        px = np.asarray(Image.open(out / "generated_images" / f"image_{i}.png"))
        assert px.shape == (size, size, 3) and px.dtype == np.uint8
        want = ref[i].mul(255).to(torch.uint8).permute(1, 2, 0).numpy()              # ToPILImage: scale then truncate
        assert np.array_equal(px, want)
    with pytest.raises(FileExistsError):                                             # generate.py:21-22
        G.sample_and_save(captions, minimagen=m, save_directory=str(out))
    with pytest.raises(AssertionError):
        G.sample_and_save(captions)
    with pytest.raises(AssertionError):
        G.sample_and_save(captions, minimagen=m, training_directory=d, save_directory=str(tmp_path / "z"))



@pytest.mark.gpu
def test_bench_two_ranks_control_flow_on_one_gpu(tmp_path):
    """The N > 1 path of bench.py as the driver launches it (torch.distributed.run, one process per rank): sharded text rows, noise keyed
    by the global row, the gather, the MAX-reduced timing and rank 0's JSON line.  MINIMAGEN_BENCH_ONE_GPU=1 puts both ranks on cuda:0
    with gloo collectives (RCCL refuses two ranks on one device; a single-GPU box is all the test tier has) -- a control-flow check,
    not a performance number."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MINIMAGEN_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = str(29000 + os.getpid() % 1500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "2", "--timesteps", "25",
           "--no-breakdown", "--no-cpu-baseline", "--no-secondary", "--no-t5"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4 and line["config"]["parallelism"] == "dp2"
    assert line["value"] > 0 and line["scaling"] == "weak" and line["steps"] == 1
    assert line["pipelined"] and line["pipelined_equals_synchronous"] and line["value_no_pipeline"] > 0
    assert [r["rows"] for r in line["per_rank"]] == [2, 2]
    # strong scaling (BASELINE config 4): a fixed global batch sharded over the ranks
    out = subprocess.run(cmd + ["--global-batch", "4"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["config"]["global_batch"] == 4 and [r["rows"] for r in line["per_rank"]] == [2, 2]


@pytest.mark.gpu
def test_bench_two_ranks_over_rccl():
    """the real multi-GPU branch (one rank per GPU, backend "nccl" = RCCL, all_gather_into_tensor of the finished images on its own
    stream): runs only where two devices are visible (the single-GPU test tier skips it)"""
    import json
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MINIMAGEN_BENCH_ONE_GPU", None)
    port = str(27000 + os.getpid() % 1500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--timesteps", "25",
           "--no-breakdown", "--no-cpu-baseline", "--no-secondary", "--no-t5"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["all_gather_ms"] > 0 and line["pipelined_equals_synchronous"]


@pytest.mark.gpu
def test_bench_rccl_calls_with_a_group_of_one():
    """every RCCL call of bench.py's N > 1 control flow on the single-GPU tier: MINIMAGEN_DIST_SINGLE=1 makes a one-rank launch take it
    (init_process_group("nccl", device_id), barriers, the all_gather_into_tensor on its own stream behind the last stage's event, the
    device-side max-over-ranks all_reduce, all_gather_object) -- what RCCL is asked to do is the same for one rank and for eight"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MINIMAGEN_DIST_SINGLE="1")
    env.pop("MINIMAGEN_BENCH_ONE_GPU", None)
    port = str(25000 + os.getpid() % 1500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2", "--timesteps", "25",
           "--no-breakdown", "--no-cpu-baseline", "--no-secondary", "--no-t5"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["all_gather_ms"] > 0 and line["pipelined_equals_synchronous"]
    assert [r["rows"] for r in line["per_rank"]] == [2]


_RCCL_ONE = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ["MINIMAGEN_DIST_SINGLE"] = "1"
from minimagen_amd import _lib as L
from minimagen_amd.Imagen import Imagen
from minimagen_amd.Unet import Unet
from minimagen_amd.distributed import sample_distributed, gather_samples, allreduce_gradients, GradientBucketReducer
from oracle import restated as R
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=0, world_size=1, device_id=dev)
L.use_library(L.DEFAULT_LIB)
torch.manual_seed(4)
u = Unet(dim=8, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=False, layer_cross_attns=False, memory_efficient=True)
im = Imagen([u], text_encoder_name="t5_small", image_sizes=[16], timesteps=21, cond_drop_prob=0.15).to(dev)
emb, mask = R.synthetic_text(3, length=9, seed=3)
emb, mask = emb.to(dev), mask.to(dev)
whole = im.sample(text_embeds=emb, text_masks=mask, cond_scale=1., _seed=77)
out = sample_distributed(im, text_embeds=emb, text_masks=mask, cond_scale=1., _seed=77)      # sample + all_gather_into_tensor over RCCL
assert out.data_ptr() != whole.data_ptr() and torch.equal(out, whole)
g = torch.randn(5, 3, 16, 16, device=dev)
assert torch.equal(gather_samples(g, 5), g)
# gradient all-reduce: flat buckets, asynchronous all_reduce, averaged by the world size
ps = [torch.nn.Parameter(torch.randn(n, device=dev)) for n in (1000, 70000, 33)]
for p in ps:
    p.grad = torch.randn_like(p)
want = [p.grad.clone() for p in ps]
assert allreduce_gradients(ps, bucket_mb=0.1) == 3
assert all(torch.equal(p.grad, w) for p, w in zip(ps, want))
red = GradientBucketReducer(ps, bucket_mb=0.1)
for p in ps:
    p.grad = None
loss = sum((p * p).sum() for p in ps)
loss.backward()
want = [2 * p.detach() for p in ps]
assert red.finish() == 3 and red.launched_in_backward >= 1
assert all(torch.equal(p.grad, w) for p, w in zip(ps, want))
dist.barrier()
dist.destroy_process_group()
print("ok")
'''


@pytest.mark.gpu
def test_distributed_module_over_rccl_with_a_group_of_one(tmp_path):
    """minimagen_amd.distributed on RCCL (backend "nccl") with one rank: sample_distributed's all_gather_into_tensor, allreduce_gradients'
    bucketed asynchronous all_reduce and GradientBucketReducer's hook-launched buckets run on the device and leave the values unchanged"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "r1.py"
    script.write_text(_RCCL_ONE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script), root, str(23000 + os.getpid() % 1500)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
