"""The training path: Imagen.forward -> _p_losses -> Unet.forward in train mode = the differentiable torch-op
forms of the layers (minimagen_amd/layers.py), against the oracle, against the unmodified reference (when present) and against the HIP
inference engine running the same module tree."""
import os

import pytest
import torch
import torch.nn.functional as F

from minimagen_amd.Imagen import Imagen
from minimagen_amd.Unet import Unet
from oracle import restated as R
from tests._backend import BACKENDS, GPU_ONLY, setup

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


# ---------------------------------------------------------------------------------------------------------------------------------------
# training on the device (minimagen_amd/train_ops.py): HIP forward + HIP data / weight gradients for the 3x3 convolutions


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(2, 8, 8, 16, 32, 3), (1, 16, 24, 13, 40, 5), (3, 3, 5, 9, 7, 2), (1, 40, 8, 8, 32, 1), (2, 8, 8, 64, 64, 64)])
def test_conv_wgrad_kernel(backend, case):
    """mi_conv_wgrad (split-K fp32 matrix-core GEMM over pixel tiles, partials added in a fixed order) against torch autograd in fp64: ragged
    tiles, channel counts off the 16-blocks, fewer workgroups than tiles, and run-to-run bit equality"""
    import ctypes as C
    import torch.nn.functional as F
    from minimagen_amd import _lib as L
    B, Cin, Cout, H, W, nwg = case
    if backend == "emu" and H * W > 1024:
        pytest.skip("emulator time")
    dev = setup(backend)
    lib = L.lib()
    g = torch.Generator().manual_seed(1)
    a, dy = torch.randn(B, Cin, H, W, generator=g), torch.randn(B, Cout, H, W, generator=g)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    F.conv2d(a.double(), w, b, padding=1).backward(dy.double())
    ad, dyd = a.to(dev), dy.to(dev)
    outs = []
    for rep in range(2):
        part = torch.full((lib.mi_conv_wgrad_workspace(Cin, Cout, nwg),), float('nan'), device=dev)
        dw, db = torch.full((Cout, Cin, 3, 3), float('nan'), device=dev), torch.full((Cout,), float('nan'), device=dev)
        p = L.MiConvWgradParams(B, Cin, Cout, H, W, ad.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr(), part.data_ptr(), nwg)
        L.check(lib.mi_conv_wgrad(C.byref(p), L.current_stream()), "mi_conv_wgrad")
        outs.append((dw.cpu(), db.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    tol = 2e-6 * (B * H * W) ** 0.5 * 4          # fp32 accumulation over B*H*W products of unit-variance factors
    assert (outs[0][0].double() - w.grad).abs().max() < tol and (outs[0][1].double() - b.grad).abs().max() < tol


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("shape", [(16, 8), (8, 24), (3, 8), (40, 16)])
def test_pack_conv3_on_the_device_matches_the_host_packing(backend, shape):
    """mi_pack_conv3 (the training path re-packs every weight after every optimiser step) writes the values packing.pack_conv_weight_rp /
    pack_conv_weight write on the host, for a weight and for its adjoint"""
    from minimagen_amd import _lib as L, packing as P, train_ops
    dev = setup(backend)
    lib = L.lib()
    Cout, Cin = shape
    w = (torch.randn(Cout, Cin, 3, 3, generator=torch.Generator().manual_seed(3)) * 0.07).to(dev)
    exp = P.rp_weight_exponent(float(w.abs().max()))
    for adjoint in (False, True):
        pk = train_ops._Pack(w, exp, adjoint)
        wl = w.flip(2, 3).transpose(0, 1).contiguous() if adjoint else w
        assert torch.equal(pk.generic.cpu().reshape(-1), P.pack_conv_weight(wl.cpu(), lib.mi_conv_cout_tile(wl.shape[0])).reshape(-1))
        if wl.shape[1] % 8 == 0:
            frag, e = P.pack_conv_weight_rp(wl.cpu())
            assert e == exp and torch.equal(pk.frag.cpu().reshape(-1), frag.reshape(-1))        # values (the host form writes -0 for dead taps)


@pytest.mark.parametrize("backend", BACKENDS)
def test_pack_conv3_multi_equals_the_single_launches(backend):
    """mi_pack_conv3_multi (ABI 12: every conv weight of a U-Net, both directions, in ONE launch from a device-resident descriptor table --
    begin_step's lagged mode) writes exactly what one mi_pack_conv3 per weight and direction writes"""
    import numpy as np
    from minimagen_amd import _lib as L, packing as P, train_ops
    dev = setup(backend)
    lib = L.lib()
    g = torch.Generator().manual_seed(5)
    ws = [(torch.randn(co, ci, 3, 3, generator=g) * sc).to(dev) for co, ci, sc in ((8, 8, 0.1), (16, 32, 3.0), (3, 8, 1e-3), (16, 24, 0.5), (8, 3, 0.2))]
    single, multi, rows = [], [], []
    for w in ws:
        exp = P.rp_weight_exponent(float(w.abs().max()))
        for adjoint in (False, True):
            single.append(train_ops._Pack(w, exp, adjoint))
            pk = train_ops._Pack(w, exp, adjoint, launch=False)
            pk.frag.fill_(7.0); pk.generic.fill_(7.0)
            multi.append(pk)
            rows.append(pk.desc(w, adjoint))
    table = torch.from_numpy(np.array(rows, dtype=train_ops._DESC_DTYPE).view(np.uint8).reshape(-1).copy()).to(dev)
    L.check(lib.mi_pack_conv3_multi(table.data_ptr(), len(rows), 16, L.current_stream()), "mi_pack_conv3_multi")
    for a, b in zip(single, multi):
        assert torch.equal(a.generic, b.generic) and torch.equal(a.frag.view(torch.int16), b.frag.view(torch.int16))
    assert lib.mi_pack_conv3_multi(0, 1, 16, L.current_stream()) != 0 and lib.mi_pack_conv3_multi(table.data_ptr(), 0, 16, L.current_stream()) != 0


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(2, 6, 16, 32, (3, 7, 15), (4, 2, 2), 3), (1, 3, 11, 40, (3, 7, 15), (4, 2, 2), 2), (1, 4, 9, 9, (3, 5), (6, 2), 1),
                                  (2, 6, 64, 64, (3, 7, 15), (4, 2, 2), 64)])
def test_crossembed_wgrad_kernel(backend, case):
    """mi_crossembed_wgrad: one K x K correlation for all members of a CrossEmbedLayer (the smaller kernels' gradients are centre windows)
    against torch autograd in fp64; ragged tiles, fewer workgroups than tiles, other member sets, run-to-run bit equality"""
    import ctypes as C
    import torch.nn.functional as F
    from minimagen_amd import _lib as L
    B, Cin, H, W, ks, cs, nwg = case
    if backend == "emu" and H * W > 1024:
        pytest.skip("emulator time")
    dev = setup(backend)
    lib = L.lib()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, H, W, generator=g)
    ws = [torch.zeros(c, Cin, k, k, dtype=torch.float64, requires_grad=True) for k, c in zip(ks, cs)]
    bs = [torch.zeros(c, dtype=torch.float64, requires_grad=True) for c in cs]
    y = torch.cat([F.conv2d(x.double(), w, b, padding=k // 2) for w, b, k in zip(ws, bs, ks)], 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    xd, dyd = x.to(dev), dy.to(dev)
    runs = []
    for rep in range(2):
        part = torch.full((lib.mi_crossembed_wgrad_workspace(Cin, max(ks), nwg),), float('nan'), device=dev)
        dws = [torch.full(w.shape, float('nan'), device=dev) for w in ws]
        dbs = [torch.full(b.shape, float('nan'), device=dev) for b in bs]
        p = L.MiCrossEmbedWgradParams()
        p.B, p.Cin, p.H, p.W, p.x, p.dy, p.n_kernels, p.partial, p.nwg = B, Cin, H, W, xd.data_ptr(), dyd.data_ptr(), len(ks), part.data_ptr(), nwg
        for i in range(len(ks)):
            p.ksize[i], p.cout[i], p.dw[i], p.db[i] = ks[i], cs[i], dws[i].data_ptr(), dbs[i].data_ptr()
        L.check(lib.mi_crossembed_wgrad(C.byref(p), L.current_stream()), "mi_crossembed_wgrad")
        runs.append([t.cpu() for t in dws + dbs])
    assert all(torch.equal(a, b) for a, b in zip(*runs))
    tol = 2e-6 * (B * H * W) ** 0.5 * 4
    for got, ref in zip(runs[0], [w.grad for w in ws] + [b.grad for b in bs]):
        assert (got.double() - ref).abs().max() < tol


def _block_grads(blk, x, ss, gy, hip):
    from minimagen_amd import train_ops
    train_ops.FORCE, train_ops.ENABLED = hip, hip
    try:
        for t in list(blk.parameters()) + [x] + list(ss or ()):
            t.grad = None
        y = blk(x, ss)
        y.backward(gy)
        return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in blk.parameters()] + [t.grad.clone() for t in (ss or ())]
    finally:
        train_ops.FORCE, train_ops.ENABLED = False, True


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(8, 16, 16, 32, True), (16, 8, 9, 12, False), (24, 3, 8, 8, True), (3, 8, 8, 16, False), (8, 8, 64, 64, True),
                                  (128, 64, 16, 16, True)])
def test_block_function_gradients(backend, case):
    """Block (layers.py:131-145) through train_ops._BlockFn -- fused HIP forward, HIP data gradient (the forward kernel on the adjoint
    weights), mi_conv_wgrad, recomputed pointwise part -- against the same module on torch ops: output and the gradients of the input,
    GroupNorm affine, conv weight / bias and the scale / shift, on the row-paired kernels (narrow and wide) and the direct-conv family"""
    from minimagen_amd.layers import Block
    Cin, Cout, H, W, with_ss = case
    if backend == "emu" and (H * W > 1024 or Cin > 64):
        pytest.skip("emulator time")
    dev = setup(backend)
    torch.manual_seed(0)
    blk = Block(Cin, Cout, groups=8 if Cin % 8 == 0 else 3).train().to(dev)
    with torch.no_grad():
        blk.groupnorm.weight.add_(0.3 * torch.randn(Cin, device=dev))
        blk.groupnorm.bias.add_(0.2 * torch.randn(Cin, device=dev))
    x = (torch.randn(2, Cin, H, W, device=dev) * 1.5 + 0.2).requires_grad_()
    ss = tuple((torch.randn(2, Cin, 1, 1, device=dev) * 0.3).requires_grad_() for _ in range(2)) if with_ss else None
    gy = torch.randn(2, Cout, H, W, device=dev) * 1e-3                # loss gradients are small: the data-gradient conv range-scales them
    ref = _block_grads(blk, x, ss, gy, False)
    got = _block_grads(blk, x, ss, gy, True)
    for a, b in zip(got, ref):
        assert (a - b).abs().max() <= 2e-5 * max(1e-6, float(b.abs().max())), (case, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [((2, 300, 16), True), ((3, 77, 8), False), ((5, 32), True), ((2, 33, 512), True), ((1, 700, 16), False), ((4, 9, 100), True),
                                  ((2, 3, 1024), True)])
def test_layernorm_kernels_forward_and_backward(backend, case):
    """train_ln.hip through train_ops.layer_norm (the reference's LayerNorm, layers.py:333-343, and nn.LayerNorm of the conditioning stack): output,
    dx, dgamma, dbeta against torch's F.layer_norm in fp64 -- the work-item-per-row form (dim 8 / 16 / 32), the wave-per-row form, ragged row
    counts, with and without a bias, rows with a large common offset; run-to-run bit equality of the parameter gradients (fixed-order reduction)"""
    from minimagen_amd import train_ops
    shape, with_bias = case
    dev = setup(backend)
    torch.manual_seed(3)
    dim = shape[-1]
    x = (torch.randn(*shape, device=dev) * 2.0 + torch.randn(*shape[:-1], 1, device=dev) * 30.0).requires_grad_()
    w = (1.0 + 0.3 * torch.randn(dim, device=dev)).requires_grad_()
    b = (0.2 * torch.randn(dim, device=dev)).requires_grad_() if with_bias else None
    gy = torch.randn(*shape, device=dev)
    train_ops.FORCE, train_ops.ENABLED = True, True
    try:
        assert train_ops.layer_norm_supported(x, w)
        outs = []
        for rep in range(2):
            y = train_ops.layer_norm(x, w, b, 1e-5)
            g = torch.autograd.grad(y, [x, w] + ([b] if with_bias else []), gy)
            outs.append((y.detach(),) + g)
    finally:
        train_ops.FORCE, train_ops.ENABLED = False, True
    x64, w64 = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    b64 = b.detach().double().requires_grad_() if with_bias else None
    y64 = torch.nn.functional.layer_norm(x64, (dim,), w64, b64, 1e-5)
    g64 = torch.autograd.grad(y64, [x64, w64] + ([b64] if with_bias else []), gy.double())
    for a, r in zip(outs[0], (y64.detach(),) + g64):
        assert (a.double() - r).abs().max() <= 3e-5 * max(1.0, float(r.abs().max())), (case, float((a.double() - r).abs().max()), float(r.abs().max()))
    for a, c in zip(outs[0], outs[1]):
        assert torch.equal(a, c)


def _unet_loss_grads(im, imgs, emb, mask, hip, unet_number):
    from minimagen_amd import train_ops
    train_ops.FORCE, train_ops.ENABLED = hip, hip
    try:
        im.zero_grad(set_to_none=True)
        torch.manual_seed(11)                                           # the same timesteps / noise / dropout mask on both paths
        loss = im(imgs, text_embeds=emb, text_masks=mask, unet_number=unet_number)
        loss.backward()
        return loss.item(), {n: p.grad.clone() for n, p in im.unets[unet_number - 1].named_parameters()}
    finally:
        train_ops.FORCE, train_ops.ENABLED = False, True


@pytest.mark.parametrize("backend", BACKENDS)
def test_training_step_on_the_device_path_equals_the_torch_op_path(backend):
    """Imagen.forward -> loss.backward() with every Block / Upsample conv / final_conv of the U-Net on the HIP kernels (forward, data
    gradient, weight gradient) against the torch-op graph whose loss and gradients test_imagen_forward_matches_the_reference_loss pins
    on the unmodified reference: base U-Net and the lowres-conditioned SR U-Net of a cascade"""
    dev = setup(backend)
    torch.manual_seed(7)
    size = (16, 32) if backend == "emu" else (32, 64)
    im = Imagen((Unet(**BASE), Unet(**SR)), text_encoder_name="t5_small", image_sizes=size, timesteps=60).train().to(dev)
    imgs = torch.rand(2, 3, size[1] + 8, size[1] + 8, device=dev)
    emb, mask = R.synthetic_text(2, length=11, seed=5)
    emb, mask = emb.to(dev), mask.to(dev)
    for n in (1, 2):
        la, ga = _unet_loss_grads(im, imgs, emb, mask, False, n)
        lb, gb = _unet_loss_grads(im, imgs, emb, mask, True, n)
        assert abs(la - lb) < 1e-5 * max(1.0, abs(la)), (la, lb)
        for name, g in ga.items():
            assert (gb[name] - g).abs().max() < 1e-4 * max(1e-3, float(g.abs().max())), (n, name, float((gb[name] - g).abs().max()), float(g.abs().max()))
    # a U-Net with self- and cross-attention: the cross-attention takes the folded training form next to the HIP convolutions
    torch.manual_seed(8)
    im2 = Imagen((Unet(**NARROW_ATTN),), text_encoder_name="t5_small", image_sizes=(size[0],), timesteps=60).train().to(dev)
    la, ga = _unet_loss_grads(im2, imgs, emb, mask, False, 1)
    lb, gb = _unet_loss_grads(im2, imgs, emb, mask, True, 1)
    assert abs(la - lb) < 1e-5 * max(1.0, abs(la)), (la, lb)
    for name, g in ga.items():
        assert (gb[name] - g).abs().max() < 1e-4 * max(1e-3, float(g.abs().max())), (name, float((gb[name] - g).abs().max()), float(g.abs().max()))


_GRAD_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from minimagen_amd.Imagen import Imagen
from minimagen_amd.Unet import Unet
from minimagen_amd.distributed import allreduce_gradients, shard_bounds
from oracle import restated as R
rank = int(sys.argv[3])
backend = sys.argv[4] if len(sys.argv) > 4 else "gloo"
dev = torch.device("cpu")
if backend == "nccl":                      # RCCL: one rank per GPU; the training graph then runs its convolutions / attention on the HIP kernels
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=rank, world_size=2, device_id=dev)
else:
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=rank, world_size=2)
torch.manual_seed(3)
im = Imagen((Unet(dim=8, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=False, layer_cross_attns=False),), text_encoder_name="t5_small",
            image_sizes=(16,), timesteps=40, cond_drop_prob=0.).train().to(dev)
B = 4
imgs = torch.rand(B, 3, 16, 16).to(dev)
emb, mask = R.synthetic_text(B, length=7, seed=2)
emb, mask = emb.to(dev), mask.to(dev)
t_all = torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(5)).to(dev)
noise = torch.randn(B, 3, 16, 16, generator=torch.Generator().manual_seed(6)).to(dev)
def loss_of(rows):
    # the per-sample objective of Imagen._p_losses with fixed timesteps / noise (so that shards and the whole batch see the same draws)
    x0 = imgs[rows] * 2 - 1
    nd = im.noise_schedulers[0]
    xt = nd.q_sample(x0, t_all[rows], noise[rows])
    pred = im.unets[0](xt, t_all[rows], text_embeds=emb[rows], text_mask=mask[rows], cond_drop_prob=0.)
    return ((pred - noise[rows]) ** 2).mean(dim=(1, 2, 3))
lo, hi = shard_bounds(B, 2, rank)
im.zero_grad(set_to_none=True)
(loss_of(slice(lo, hi)).sum() / (hi - lo)).backward()
n = allreduce_gradients(im.unets[0].parameters(), bucket_mb=0.01)          # several buckets
assert n > 1
mine = [p.grad.clone() for p in im.unets[0].parameters()]
im.zero_grad(set_to_none=True)
loss_of(slice(0, B)).mean().backward()
for g, p in zip(mine, im.unets[0].parameters()):
    assert (g - p.grad).abs().max() < (1e-6 if backend == "gloo" else 2e-5) * max(1.0, float(p.grad.abs().max())), "averaged shard gradients differ from the full-batch gradient"
# the same reduction OVERLAPPED with the backward (GradientBucketReducer: post-accumulate hooks launch a bucket's all-reduce as soon as its
# gradients exist, in bucket order): same averaged gradients, and collectives really were in flight before backward() returned
from minimagen_amd.distributed import GradientBucketReducer
red = GradientBucketReducer(im.unets[0].parameters(), bucket_mb=0.01)
assert len(red.buckets) > 2
im.zero_grad(set_to_none=True)
(loss_of(slice(lo, hi)).sum() / (hi - lo)).backward()
early = red.launched_in_backward
assert early >= len(red.buckets) - 1, (early, len(red.buckets))
assert red.finish() == len(red.buckets)
for g, p in zip(mine, im.unets[0].parameters()):
    assert torch.equal(g, p.grad) or (g - p.grad).abs().max() < 1e-6 * max(1.0, float(g.abs().max())), "hook-overlapped reduction differs from allreduce_gradients"
with red.no_sync():                         # an accumulation step: no collectives, local gradients untouched
    im.zero_grad(set_to_none=True)
    (loss_of(slice(lo, hi)).sum() / (hi - lo)).backward()
    local = [p.grad.clone() for p in im.unets[0].parameters()]
assert red.finish() == 0 and all(torch.equal(a, p.grad) for a, p in zip(local, im.unets[0].parameters()))
red.remove()
# ---- a FULL data-parallel training step on the two ranks (train.py:99-102 / training.py:344-478 semantics: backward -> gradient-norm clip at 50 ->
# Adam): shard backward with the hook-overlapped reduction, clip, minimagen_amd.optim.Adam.step -- against the same step taken on the whole batch by
# one process' worth of arithmetic; afterwards both ranks hold the same parameters
import copy
from minimagen_amd.optim import Adam
params = list(im.unets[0].parameters())
start = [p.detach().clone() for p in params]
def one_step(shard):
    for p, s0 in zip(params, start):
        p.data.copy_(s0)
    opt = Adam(params, lr=1e-2, eps=1e-3)          # (a large eps: with Adam's default a gradient of ~1e-12 whose rounding noise flips sign moves its parameter by +-lr)
    r2 = GradientBucketReducer(params, bucket_mb=0.01) if shard else None
    for it in range(2):                      # two steps: the second sees the first one's moments
        im.zero_grad(set_to_none=True)
        if shard:
            (loss_of(slice(lo, hi)).sum() / (hi - lo)).backward()
            r2.finish()
        else:
            loss_of(slice(0, B)).mean().backward()
        torch.nn.utils.clip_grad_norm_(params, 50)
        opt.step()
    if r2 is not None:
        r2.remove()
    return [p.detach().clone() for p in params]
after_dp = one_step(True)
after_full = one_step(False)
for a, f, s0 in zip(after_dp, after_full, start):
    assert (a - f).abs().max() < 1e-5 * max(1.0, float(f.abs().max())), "the two-rank step differs from the full-batch step"
assert any((a - s0).abs().max() > 1e-4 for a, s0 in zip(after_dp, start)), "the step did not move the parameters"
flat = torch.cat([a.reshape(-1).double() for a in after_dp]).cpu()
both = [torch.zeros_like(flat) for _ in range(2)]
dist.all_gather(both, flat) if backend == "gloo" else dist.all_gather([b.to(dev) for b in both], flat.to(dev))
if backend == "gloo":
    assert torch.equal(both[0], both[1]), "the ranks' parameters diverged after a data-parallel step"
dist.barrier()
dist.destroy_process_group()
print("ok")
'''


def test_allreduce_gradients_world_size_2_gloo(tmp_path):
    """data-parallel training on two ranks: each differentiates its contiguous shard, allreduce_gradients (bucketed, averaged) leaves every rank
    with the gradient of the whole batch; the hook-overlapped GradientBucketReducer gives the same gradients; and two FULL steps (backward ->
    overlapped reduction -> clip -> Adam) move the parameters exactly as the full-batch step does, identically on both ranks"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "g.py"
    script.write_text(_GRAD_WORKER)
    port = str(33500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), root, port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_training_packs_follow_the_weights():
    """the fragment packs cached on a conv weight are rebuilt after an in-place optimiser update, after ``p.data = ...`` and stay put otherwise"""
    from minimagen_amd import train_ops
    setup("emu")
    w = torch.nn.Parameter(torch.randn(8, 8, 3, 3) * 0.1)
    a, _ = train_ops._packs(w)
    assert train_ops._packs(w)[0] is a
    with torch.no_grad():
        w.add_(0.5)
    b, _ = train_ops._packs(w)
    assert b is not a and not torch.equal(b.generic, a.generic)
    w.data = torch.randn(8, 8, 3, 3)
    c, _ = train_ops._packs(w)
    assert c is not b and torch.equal(c.generic.reshape(8, 3, 3, 8), w.detach().permute(1, 2, 3, 0))


def test_training_packs_notice_updates_made_through_data():
    """``p.data.mul_()`` / ``.data.copy_()`` (EMA copy-in, hand-written SGD) bump neither the version counter nor the storage pointer: the
    content fingerprint taken by ``begin_step`` must still drop the stale packs (forward and data-gradient convs otherwise keep
    the old weights while the weight-gradient kernel and the torch ops see the new ones)"""
    from minimagen_amd import train_ops
    setup("emu")
    conv = torch.nn.Conv2d(8, 8, 3, padding=1)
    train_ops.begin_step(conv)
    a, _ = train_ops._packs(conv.weight)
    train_ops.begin_step(conv)
    assert train_ops._packs(conv.weight)[0] is a                       # untouched weights keep their packs
    v, ptr = conv.weight._version, conv.weight.data_ptr()
    conv.weight.data.mul_(2.0)
    assert (conv.weight._version, conv.weight.data_ptr()) == (v, ptr)   # invisible to the identity key ...
    train_ops.begin_step(conv)
    b, _ = train_ops._packs(conv.weight)
    assert b is not a                                                  # ... but not to the fingerprint
    assert torch.equal(b.generic.reshape(8, 3, 3, -1)[..., :8], conv.weight.detach().permute(1, 2, 3, 0))
    train_ops.invalidate(conv)
    assert not hasattr(conv.weight, "_mi_train_packs")


def test_cross_attention_folded_training_form_equals_the_reference_form():
    """CrossAttention._forward_folded (the sampler's fold as differentiable torch ops, taken by the training graph on the GPU for C < dim_head)
    against the layer's reference-order forward: output and every gradient (token input, context, null_kv, to_q, to_kv, to_out, both norms)"""
    from minimagen_amd import train_ops
    from minimagen_amd.layers import CrossAttention
    setup("emu")                                  # the folded form's core runs on the HIP kernels (emulator build here)
    torch.manual_seed(4)
    ca = CrossAttention(dim=16, context_dim=24, norm_context=True).train()
    with torch.no_grad():
        for p_ in ca.parameters():
            p_.add_(0.1 * torch.randn_like(p_))
    x = torch.randn(2, 50, 16, requires_grad=True)
    ctx = torch.randn(2, 9, 24, requires_grad=True)
    mask = torch.arange(9)[None, :] < torch.tensor([9, 4])[:, None]
    gy = torch.randn(2, 50, 16)
    res = {}
    for folded in (False, True):
        train_ops.FORCE = folded
        try:
            for t in list(ca.parameters()) + [x, ctx]:
                t.grad = None
            y = ca(x, ctx, mask=mask)
            y.backward(gy)
            res[folded] = [y.detach().clone(), x.grad.clone(), ctx.grad.clone()] + [p_.grad.clone() for p_ in ca.parameters()]
        finally:
            train_ops.FORCE = False
    for a, b in zip(res[True], res[False]):
        assert (a - b).abs().max() <= 2e-5 * max(1e-3, float(b.abs().max())), (float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(2, 300, 8, 16, 37, True), (1, 70, 2, 8, 261, False), (3, 129, 4, 32, 20, True), (2, 4096, 8, 16, 261, True)])
def test_folded_attention_kernels(backend, case):
    """mi_folded_attn_fwd / _bwd (out = sum_h softmax(q kf_h^T) vf_h without the score tensor; dq, dkf, dvf with per-chunk partials) against
    torch autograd in fp64: ragged token counts, masked context rows, every instantiated channel count, several token chunks"""
    from minimagen_amd import train_ops
    B, n, H, Cc, J, with_mask = case
    if backend == "emu" and n > 1024:
        pytest.skip("emulator time")
    dev = setup(backend)
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B, n, Cc, generator=g)
    kf = torch.randn(B, H, J, Cc, generator=g) * 0.5
    vf = torch.randn(B, H, J, Cc, generator=g)
    mask = (torch.arange(J)[None, :] < torch.tensor([J - (5 * r) % J for r in range(B)])[:, None]) if with_mask else None
    gy = torch.randn(B, n, Cc, generator=g)
    qd, kd, vd = (t.double().requires_grad_() for t in (q, kf, vf))
    sim = torch.einsum('bnc,bhjc->bhnj', qd, kd)
    if mask is not None:
        sim = sim.masked_fill(~mask[:, None, None, :], -torch.finfo(torch.float32).max)
    ref = torch.einsum('bhnj,bhjc->bnc', sim.softmax(-1), vd)
    ref.backward(gy.double())
    train_ops.FORCE = True
    try:
        qh, kh, vh = (t.to(dev).requires_grad_() for t in (q, kf, vf))
        out = train_ops.folded_attention(qh, kh, vh, None if mask is None else mask.to(dev))
        out.backward(gy.to(dev))
    finally:
        train_ops.FORCE = False
    for got, want in ((out, ref), (qh.grad, qd.grad), (kh.grad, kd.grad), (vh.grad, vd.grad)):
        assert (got.detach().cpu().double() - want.detach()).abs().max() < 3e-5 * max(1.0, float(want.abs().max())), (case, float((got.detach().cpu().double() - want.detach()).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("B", [32, 16])
def test_folded_attention_matrix_core_forms_at_the_benched_shapes(B):
    """the launch forms the SR training step takes at its benched batch (eight waves x two token tiles at B = 32, four waves x two at B = 16; the
    fp64 case above reaches only the one-tile form): forward, dq, dkf, dvf of the matrix-core kernels against the fp32 VALU kernels of the same
    library (MI_FOLDED_ATTN_VALU, read per call) on the SR U-Net's shape -- same arithmetic class, another summation order"""
    from minimagen_amd import train_ops
    dev = setup("gpu")
    n, H, Cc, J = 4096, 8, 16, 261
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, n, Cc, generator=g).to(dev)
    kf = (torch.randn(B, H, J, Cc, generator=g) * 0.5).to(dev)
    vf = torch.randn(B, H, J, Cc, generator=g).to(dev)
    mask = (torch.arange(J)[None, :] < torch.tensor([J - (7 * r) % 40 for r in range(B)])[:, None]).to(dev)
    gy = torch.randn(B, n, Cc, generator=g).to(dev)
    res = {}
    for valu in (False, True):
        if valu:
            os.environ["MI_FOLDED_ATTN_VALU"] = "1"
        try:
            qh, kh, vh = (t.clone().requires_grad_() for t in (q, kf, vf))
            out = train_ops.folded_attention(qh, kh, vh, mask)
            out.backward(gy)
            torch.cuda.synchronize()
            res[valu] = [t.detach().clone() for t in (out, qh.grad, kh.grad, vh.grad)]
        finally:
            os.environ.pop("MI_FOLDED_ATTN_VALU", None)
    for name, a, b in zip(("out", "dq", "dkf", "dvf"), res[False], res[True]):
        assert torch.isfinite(a).all()
        assert (a - b).abs().max() <= 2e-5 * max(1.0, float(b.abs().max())), (name, float((a - b).abs().max()), float(b.abs().max()))
    assert not torch.equal(res[False][0], res[True][0])          # (two different kernels did run)


@pytest.mark.gpu
def test_allreduce_gradients_over_rccl(tmp_path):
    """the same data-parallel step with one rank per GPU over RCCL (backend "nccl"), the training graph on the HIP kernels: runs only where
    two devices are visible (the single-GPU test tier skips it)"""
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "g.py"
    script.write_text(_GRAD_WORKER)
    port = str(35500 + os.getpid() % 2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), root, port, str(r), "nccl"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


@pytest.mark.parametrize("backend", BACKENDS)
def test_optimiser_steps_on_the_device_path_track_the_torch_op_path(backend):
    """four Adam steps from the same initial weights: the device path (weights re-packed after every in-place update, CrossEmbed tables rebuilt,
    folded attention) and the torch-op path end at the same parameters and losses"""
    import copy
    from minimagen_amd import train_ops
    dev = setup(backend)
    torch.manual_seed(12)
    im0 = Imagen((Unet(**NARROW_ATTN),), text_encoder_name="t5_small", image_sizes=(16,), timesteps=50, cond_drop_prob=0.1).train().to(dev)
    imgs = torch.rand(2, 3, 16, 16, device=dev)
    emb, mask = R.synthetic_text(2, length=9, seed=4)
    emb, mask = emb.to(dev), mask.to(dev)
    runs = {}
    for hip in (False, True):
        im = copy.deepcopy(im0)
        opt = torch.optim.Adam(im.parameters(), lr=2e-3)
        train_ops.FORCE, train_ops.ENABLED = hip, hip
        try:
            losses = []
            for k in range(4):
                torch.manual_seed(100 + k)
                loss = im(imgs, text_embeds=emb, text_masks=mask)
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
                losses.append(loss.item())
        finally:
            train_ops.FORCE, train_ops.ENABLED = False, True
        runs[hip] = (losses, [p.detach().clone() for p in im.parameters()])
    for a, b in zip(*[runs[h][0] for h in (True, False)]):
        assert abs(a - b) < 2e-4 * max(1.0, abs(b)), (runs[True][0], runs[False][0])
    # Adam normalises the update: a parameter whose gradient is tiny can move by lr in either direction -- compare where the update is well defined
    worst = max(float((a - b).abs().max()) for a, b in zip(runs[True][1], runs[False][1]))
    assert worst < 4 * 2e-3, worst
    close = sum(int(((a - b).abs() < 2e-4).sum()) for a, b in zip(runs[True][1], runs[False][1])) / sum(a.numel() for a in runs[True][1])
    assert close > 0.97, close


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [("k4s2", 8, 8, 2, 16, 32), ("k4s2", 8, 16, 1, 24, 16), ("1x1", 16, 8, 2, 16, 32), ("1x1", 32, 16, 1, 8, 16)])
def test_downsample_and_1x1_convs_on_the_hip_3x3_kernels(backend, case):
    """Downsample (Conv2d k4 s2 p1, layers.py:308-319) as a 3x3 conv over the space-to-depth image and the 1x1 res_conv (layers.py:415) as the centre
    tap of a 3x3 one (train_ops.conv4x4s2_forward / conv1x1_forward): output, input gradient, weight and bias gradients against torch's conv2d"""
    from minimagen_amd import train_ops
    kind, Cin, Cout, B, H, W = case
    dev = setup(backend)
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    conv = torch.nn.Conv2d(Cin, Cout, 4, 2, 1) if kind == "k4s2" else torch.nn.Conv2d(Cin, Cout, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.3)
        conv.bias.copy_(torch.randn(Cout, generator=g))
    conv = conv.to(dev)
    x = (torch.randn(B, Cin, H, W, generator=g) * 1.3 + 0.2).to(dev).requires_grad_(True)
    gy = torch.randn(B, Cout, H // (2 if kind == "k4s2" else 1), W // (2 if kind == "k4s2" else 1), generator=g).to(dev)
    ref = conv(x)
    rgx, rgw, rgb = torch.autograd.grad(ref, (x, conv.weight, conv.bias), gy)
    train_ops.FORCE = backend == "emu"
    try:
        assert (train_ops.is_conv4x4s2(conv, x) if kind == "k4s2" else train_ops.is_conv1x1(conv, x)) and train_ops.active(x)
        out = train_ops.conv4x4s2_forward(conv, x) if kind == "k4s2" else train_ops.conv1x1_forward(conv, x)
        gx, gw, gb = torch.autograd.grad(out, (x, conv.weight, conv.bias), gy)
    finally:
        train_ops.FORCE = False
    for name, a, b in (("out", out, ref), ("dx", gx, rgx), ("dw", gw, rgw), ("db", gb, rgb)):
        err, mag = float((a - b).abs().max()), float(b.abs().max())
        assert a.shape == b.shape and err < 3e-5 * max(1.0, mag), (case, name, err, mag)


@pytest.mark.gpu
def test_lagged_scales_never_leave_a_stale_pack():
    """weights on the GPU: ``begin_step`` re-packs from the CURRENT values on the device with the fragment scaling of the PREVIOUS step's maxima
    (no host round trip per step).  Updates made behind the version counter (``p.data.mul_``) reach the very next forward; a 100-fold
    out-of-band rescale stays inside the scale's head room; a training step agrees with the synchronous fingerprint mode."""
    import copy
    from minimagen_amd import train_ops
    dev = setup("gpu")
    assert train_ops.LAGGED
    torch.manual_seed(21)
    conv = torch.nn.Conv2d(8, 8, 3, padding=1).to(dev)
    x = torch.randn(2, 8, 16, 16, device=dev)
    for factor in (1.0, 2.0, 100.0, 0.01):
        conv.weight.data.mul_(factor)                                   # invisible to (version, pointer)
        train_ops.begin_step(conv)
        a, _ = train_ops._packs(conv.weight)
        assert torch.equal(a.generic.reshape(8, 3, 3, -1)[..., :8], conv.weight.detach().permute(1, 2, 3, 0))
        y = train_ops.conv3x3_forward(conv, x)
        ref = F.conv2d(x.double(), conv.weight.detach().double(), conv.bias.detach().double(), padding=1)
        assert (y.double() - ref).abs().max() < 2e-6 * float(ref.abs().max()), factor
    st = conv.__dict__["_mi_lagged"]
    assert st["k"] == 4 and conv.weight._mi_fingerprint[1] == 4
    # a whole training step, lagged against synchronous scales: same loss, same gradients (the scale is an exact power of two either way)
    torch.manual_seed(9)
    im = Imagen((Unet(**BASE), Unet(**SR)), text_encoder_name="t5_small", image_sizes=(32, 64), timesteps=60).train().to(dev)
    imgs = torch.rand(2, 3, 72, 72, device=dev)
    emb, mask = R.synthetic_text(2, length=11, seed=5)
    emb, mask = emb.to(dev), mask.to(dev)
    res = {}
    for lagged in (True, False):
        train_ops.LAGGED = lagged
        try:
            m = copy.deepcopy(im)
            opt = torch.optim.Adam(m.parameters(), lr=1e-3)
            losses = []
            for step in range(3):
                for n in (1, 2):
                    torch.manual_seed(100 + step)
                    loss = m(imgs, text_embeds=emb, text_masks=mask, unet_number=n)
                    opt.zero_grad(set_to_none=True)
                    loss.backward()
                    opt.step()
                    losses.append(float(loss))
            res[lagged] = (losses, [p.detach().clone() for p in m.parameters()])
        finally:
            train_ops.LAGGED = True
    for la, lb in zip(*[res[k][0] for k in (True, False)]):
        assert abs(la - lb) < 1e-5 * max(1.0, abs(lb)), (res[True][0], res[False][0])
    # Adam normalises the update: a parameter whose gradient is tiny can move by lr in either direction -- compare where the update is well defined
    worst = max(float((a - b).abs().max()) for a, b in zip(res[True][1], res[False][1]))
    assert worst < 3 * 2e-3, worst
    close = sum(int(((a - b).abs() < 2e-4).sum()) for a, b in zip(res[True][1], res[False][1])) / sum(a.numel() for a in res[True][1])
    assert close > 0.97, close
