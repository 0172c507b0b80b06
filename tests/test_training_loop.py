"""SURVEY 8(f) rank 3, the driver side: the multi-tensor Adam kernel against torch.optim.Adam, and the reference's train.py flow
(minimagen/training.py:344-478, train.py:23-102) through ``minimagen.training`` with an offline dataset."""
import os

import pytest
import torch

from tests._backend import BACKENDS, setup


@pytest.mark.parametrize("backend", BACKENDS)
def test_adam_kernel_matches_torch_adam(backend):
    """mi_adam_step (one launch for all tensors) vs torch.optim.Adam's single-tensor update: parameters after 6 steps with fresh gradients each
    step (a tensor without a gradient in some steps, weight decay in a second group); the two optimisers' state dicts are interchangeable"""
    from minimagen_amd.optim import Adam
    dev = setup(backend)
    g = torch.Generator().manual_seed(3)
    shapes = [(16, 8, 3, 3), (16,), (5000,), (3, 7), (1,), (33, 129)]
    base = [torch.randn(s, generator=g) for s in shapes]
    mine = [torch.nn.Parameter(b.clone().to(dev)) for b in base]
    ref = [torch.nn.Parameter(b.clone()) for b in base]
    groups = lambda ps: [dict(params=ps[:4]), dict(params=ps[4:], weight_decay=0.01, lr=3e-3)]
    om, orf = Adam(groups(mine), lr=1e-2), torch.optim.Adam(groups(ref), lr=1e-2, foreach=False)
    for step in range(6):
        for k, (a, b) in enumerate(zip(mine, ref)):
            if k == 3 and step % 2:                      # a parameter that gets no gradient in some steps: its own step count
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, generator=g) * (10.0 ** (k - 3))
            a.grad, b.grad = gr.clone().to(dev), gr.clone()
        om.step(); orf.step()
    for a, b in zip(mine, ref):
        assert (a.detach().cpu() - b.detach()).abs().max() < 2e-6 * max(1.0, float(b.abs().max())), (a.shape, float((a.detach().cpu() - b.detach()).abs().max()))
    sm, sr = om.state_dict(), orf.state_dict()
    assert sm["state"].keys() == sr["state"].keys() and all(set(sm["state"][k]) == set(sr["state"][k]) for k in sm["state"])
    for k in sm["state"]:
        assert float(sm["state"][k]["step"]) == float(sr["state"][k]["step"])
        assert (sm["state"][k]["exp_avg_sq"].cpu() - sr["state"][k]["exp_avg_sq"]).abs().max() < 1e-6 * max(1.0, float(sr["state"][k]["exp_avg_sq"].abs().max()))
    # torch's optimiser continues from this one's state and vice versa
    o2 = torch.optim.Adam(groups([torch.nn.Parameter(a.detach().cpu().clone()) for a in mine]), lr=1e-2, foreach=False)
    o2.load_state_dict({"state": {k: {n: (v.cpu() if torch.is_tensor(v) else v) for n, v in s.items()} for k, s in sm["state"].items()}, "param_groups": sr["param_groups"]})
    om2 = Adam(groups([torch.nn.Parameter(b.detach().clone().to(dev)) for b in ref]), lr=1e-2)
    om2.load_state_dict(sr)
    ps2, pm2 = [p for gr_ in o2.param_groups for p in gr_["params"]], [p for gr_ in om2.param_groups for p in gr_["params"]]
    for a, b in zip(pm2, ps2):
        gr = torch.randn(a.shape, generator=g)
        a.grad, b.grad = gr.clone().to(dev), gr.clone()
    om2.step(); o2.step()
    for a, b in zip(pm2, ps2):
        assert (a.detach().cpu() - b.detach()).abs().max() < 4e-6 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("backend", BACKENDS)
def test_reference_train_flow_offline(backend, tmp_path, monkeypatch):
    """What the reference's ``train.py -test`` does (train.py:23-102), through the ``minimagen.*`` import path, with SyntheticCaptions in place
    of the Conceptual-Captions download: parser -> testing parameters -> training directory -> BaseTest / SuperTest U-Nets -> Imagen ->
    save_training_info -> Adam -> MinimagenTrain (two epochs: training batches, checkpoints, validation, best state dicts) -> the directory
    loads back with minimagen.generate.load_minimagen.  On the GPU the loop runs the HIP training graph and the one-launch Adam step."""
    from minimagen.Imagen import Imagen
    from minimagen.Unet import Unet, BaseTest, SuperTest
    from minimagen.generate import load_minimagen, load_params
    from minimagen.t5 import get_encoded_dim
    from minimagen.training import (get_minimagen_parser, get_minimagen_dl_opts, create_directory, get_model_size, save_training_info,
                                    get_default_args, MinimagenTrain, load_testing_parameters, load_restart_training_parameters,
                                    get_model_params, SyntheticCaptions, ConceptualCaptions)
    from minimagen_amd import optim, train_ops
    dev = setup(backend)
    monkeypatch.chdir(tmp_path)
    args = load_testing_parameters(get_minimagen_parser().parse_args(["-test", "-cn", "2"]))
    assert (args.BATCH_SIZE, args.EPOCHS, args.T5_NAME, args.TIMESTEPS, args.CHCKPT_NUM) == (2, 2, "t5_small", 25, 2)
    if backend == "emu":
        args.IMG_SIDE_LEN = 32                     # (emulator time; the GPU run keeps the reference's 128)
    with pytest.raises(NotImplementedError):
        ConceptualCaptions(args, smalldata=True)
    timestamp = "20260101_000000"
    training_dir = create_directory(f"./training_{timestamp}")
    data = SyntheticCaptions(8, args.IMG_SIDE_LEN, get_encoded_dim(args.T5_NAME), max_words=args.MAX_NUM_WORDS, seed=1)
    train_ds, valid_ds = torch.utils.data.random_split(data, [6, 2], generator=torch.Generator().manual_seed(0))
    dl_opts = {**get_minimagen_dl_opts(dev), "batch_size": args.BATCH_SIZE, "num_workers": args.NUM_WORKERS}
    train_dl, valid_dl = torch.utils.data.DataLoader(train_ds, **dl_opts), torch.utils.data.DataLoader(valid_ds, **dl_opts)
    imagen_params = dict(image_sizes=(args.IMG_SIDE_LEN // 2, args.IMG_SIDE_LEN), timesteps=args.TIMESTEPS, cond_drop_prob=0.15, text_encoder_name=args.T5_NAME)
    unets_params = [get_default_args(BaseTest), get_default_args(SuperTest)]
    torch.manual_seed(0)
    unets = [Unet(**p).to(dev) for p in unets_params]
    imagen = Imagen(unets=unets, **imagen_params).to(dev)
    before = [p.detach().clone() for p in imagen.parameters()]
    unets_params = [{**get_default_args(Unet), **p} for p in unets_params]
    imagen_params = {**get_default_args(Imagen), **imagen_params}
    save_training_info(args, timestamp, unets_params, imagen_params, get_model_size(imagen), training_dir)
    optimizer = optim.Adam(imagen.parameters(), lr=args.OPTIM_LR)
    train_ops.FORCE = backend == "emu"
    try:
        MinimagenTrain(timestamp, args, unets, imagen, train_dl, valid_dl, training_dir, optimizer, timeout=600)
    finally:
        train_ops.FORCE = False
    root = tmp_path / f"training_{timestamp}"
    text = (root / "training_progess.txt").read_text()
    assert "model size:" in text and text.count("EPOCH") == 2 and "TRAINING ABORTED" not in text
    assert text.count("Checkpoint created at batch number") == 4 and "U-Nets Avg Valid Losses" in text and "U-Nets Best Valid Losses" in text
    assert sorted(os.listdir(root / "tmp")) == ["unet_0_tmp.pth", "unet_1_tmp.pth"]
    assert sorted(os.listdir(root / "state_dicts")) == [f"unet_0_state_{timestamp}.pth", f"unet_1_state_{timestamp}.pth"]
    moved = max(float((p.detach() - b).abs().max()) for p, b in zip(imagen.parameters(), before))
    assert 0 < moved < 0.05 and all(torch.isfinite(p).all() for p in imagen.parameters())
    # the directory is the on-disk format of generate.py / a restart
    up, ip = load_params(str(root))
    assert up == get_model_params(str(root / "parameters"))[0] and tuple(ip["image_sizes"]) == (args.IMG_SIDE_LEN // 2, args.IMG_SIDE_LEN)
    again = load_minimagen(str(root))
    best = torch.load(root / "state_dicts" / f"unet_1_state_{timestamp}.pth", map_location="cpu")
    assert all(torch.equal(v.cpu(), best[k]) for k, v in again.unets[1].state_dict().items())
    rargs = get_minimagen_parser().parse_args(["-rd", str(root), "-s", "999", "-t", "7"])
    rargs = load_restart_training_parameters(rargs)
    assert (rargs.IMG_SIDE_LEN, rargs.TIMESTEPS, rargs.T5_NAME, rargs.MAX_NUM_WORDS) == (args.IMG_SIDE_LEN, 25, "t5_small", 32)
