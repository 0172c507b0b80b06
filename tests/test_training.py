"""The training path: Imagen.forward -> _p_losses -> Unet.forward in train mode = the differentiable torch-op
forms of the layers (minimagen_amd/layers.py), against the oracle, against the unmodified reference (when present) and against the HIP
inference engine running the same module tree."""
import os

import pytest
import torch

from minimagen_amd.Imagen import Imagen
from minimagen_amd.Unet import Unet
from oracle import restated as R
from tests._backend import GPU_ONLY, setup

NARROW_ATTN = dict(dim=8, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True), attend_at_middle=True)
BASE = dict(dim=8, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=False, layer_cross_attns=False, memory_efficient=False)
SR = dict(dim=8, dim_mults=(1, 2), num_resnet_blocks=(1, 2), layer_attns=False, layer_cross_attns=False, memory_efficient=True)


def test_train_mode_forward_is_the_oracle_forward():
    """every layer's torch-op ``forward`` composed by Unet._forward_train == the oracle's restatement of minimagen/Unet.py:355-472"""
    torch.manual_seed(0)
    for kw, lowres in ((NARROW_ATTN, False), (dict(SR, lowres_cond=True), True)):
        u = Unet(**kw).train()
        sd = {k: v.clone() for k, v in u.state_dict().items()}
        x, tm = torch.randn(2, 3, 16, 16), torch.tensor([5, 80])
        emb, mask = R.synthetic_text(2, length=9, seed=1)
        extra = dict(lowres_cond_img=torch.randn(2, 3, 16, 16), lowres_noise_times=torch.tensor([10, 10])) if lowres else {}
        out = u(x, tm, text_embeds=emb, text_mask=mask, **extra)
        ref = R.unet_forward(sd, x, tm, text_embeds=emb, text_mask=mask, **extra)
        assert out.requires_grad and (out - ref).abs().max() < 1e-6
        null = u(x, tm, text_embeds=emb, text_mask=mask, cond_drop_prob=1., **extra)            # every sample dropped -> the null branch
        assert (null - R.unet_forward(sd, x, tm, text_embeds=emb, text_mask=mask, cond_drop_prob=1., **extra)).abs().max() < 1e-6


def test_imagen_forward_loss_and_gradients():
    torch.manual_seed(1)
    im = Imagen((Unet(**BASE), Unet(**SR)), text_encoder_name="t5_small", image_sizes=(16, 32), timesteps=50, loss_type='l2').train()
    imgs = torch.rand(3, 3, 40, 40)
    emb, mask = R.synthetic_text(3, length=7, seed=2)
    with pytest.raises(AssertionError):
        im(imgs, text_embeds=emb, text_masks=mask)                      # Imagen.py:597-599: a cascade needs unet_number
    for n in (1, 2):
        loss = im(imgs, text_embeds=emb, text_masks=mask, unet_number=n)
        assert loss.dim() == 0 and torch.isfinite(loss) and 0.2 < loss.item() < 5.0           # an untrained net predicts ~unit-variance noise
        loss.backward()
        grads = [p.grad for p in im.unets[n - 1].parameters()]
        assert all(g is not None and torch.isfinite(g).all() for g in grads) and sum(float(g.abs().sum()) for g in grads) > 0
        other = [p.grad for p in im.unets[2 - n].parameters()]
        assert n == 2 or all(g is None for g in other)                  # only the U-Net being trained receives gradients
    for lt in ('l1', 'huber'):
        assert torch.isfinite(Imagen((Unet(**BASE),), text_encoder_name="t5_small", image_sizes=(16,), timesteps=50, loss_type=lt).train()(
            imgs, text_embeds=emb, text_masks=mask))
    # a few SGD steps on one fixed batch lower the loss (the graph is connected end to end)
    torch.manual_seed(3)
    im1 = Imagen((Unet(**BASE),), text_encoder_name="t5_small", image_sizes=(16,), timesteps=50, cond_drop_prob=0.).train()
    opt = torch.optim.Adam(im1.parameters(), lr=3e-3)
    first = last = None
    for step in range(30):
        torch.manual_seed(100)                                          # same timesteps / noise every step: a deterministic objective
        loss = im1(imgs, text_embeds=emb, text_masks=mask)
        opt.zero_grad(); loss.backward(); opt.step()
        first, last = (loss.item() if first is None else first), loss.item()
    assert last < 0.9 * first, (first, last)


@pytest.mark.skipif(not os.path.isdir("/root/reference/minimagen"), reason="the unmodified reference is only present in the build container")
def test_imagen_forward_matches_the_reference_loss():
    """same weights, same global RNG seed -> the reference's Imagen.forward draws the same timesteps / noise / dropout mask in the same
    order: the loss and every parameter gradient agree to fp32 rounding"""
    from oracle import ref_loader
    ref = ref_loader.load_reference()
    torch.manual_seed(7)
    ours = Imagen((Unet(**BASE), Unet(**SR)), text_encoder_name="t5_small", image_sizes=(16, 32), timesteps=60).train()
    theirs = ref.Imagen(unets=(ref.Unet(**BASE), ref.Unet(**SR)), text_encoder_name="t5_small", image_sizes=(16, 32), timesteps=60).train()
    theirs.load_state_dict(ours.state_dict())
    imgs = torch.rand(2, 3, 48, 48)
    emb, mask = R.synthetic_text(2, length=11, seed=5)
    for n in (1, 2):
        torch.manual_seed(11)
        la = ours(imgs, text_embeds=emb, text_masks=mask, unet_number=n)
        torch.manual_seed(11)
        lb = theirs(imgs, text_embeds=emb, text_masks=mask, unet_number=n)
        assert abs(float(la) - float(lb)) < 1e-5 * max(1.0, abs(float(lb))), (float(la), float(lb))
        la.backward(); lb.backward()
        ga = dict(ours.unets[n - 1].named_parameters())
        for name, pb in theirs.unets[n - 1].named_parameters():
            assert (ga[name].grad - pb.grad).abs().max() < 1e-4 * max(1.0, float(pb.grad.abs().max())), name


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_training_path_on_the_gpu_agrees_with_the_hip_engine(backend):
    """the two execution paths of one module tree on the MI355X: train-mode torch ops vs the HIP inference engine; and one optimiser step
    invalidates the engine's packed weights (the next evaluation sees the updated parameters)"""
    dev = setup(backend)
    torch.manual_seed(2)
    u = Unet(**NARROW_ATTN).to(dev)
    x, tm = torch.randn(2, 3, 32, 32, device=dev), torch.tensor([5, 80], device=dev)
    emb, mask = R.synthetic_text(2, length=9, seed=1)
    emb, mask = emb.to(dev), mask.to(dev)
    a = u.train()(x, tm, text_embeds=emb, text_mask=mask)
    b = u.eval()(x, tm, text_embeds=emb, text_mask=mask)
    assert a.requires_grad and not b.requires_grad and (a - b).abs().max() < 2e-5 * max(1.0, float(b.abs().max()))
    im = Imagen((u,), text_encoder_name="t5_small", image_sizes=(32,), timesteps=50).train()
    opt = torch.optim.SGD(im.parameters(), lr=0.05)
    loss = im(torch.rand(2, 3, 32, 32, device=dev), text_embeds=emb, text_masks=mask)
    opt.zero_grad(); loss.backward(); opt.step()
    c = u.eval()(x, tm, text_embeds=emb, text_mask=mask)
    d = u.train()(x, tm, text_embeds=emb, text_mask=mask)
    assert (c - b).abs().max() > 1e-4 and (c - d).abs().max() < 2e-5 * max(1.0, float(c.abs().max()))
