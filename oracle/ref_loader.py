"""Import the *unmodified* reference (``/root/reference``) on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Works only in the build
container, where ``/root/reference`` exists; the GPU box uses the committed
golden vectors instead.

The reference imports three packages that are not installed here
(SURVEY.md section 8(c) / Appendix C); this module installs minimal stand-ins
for them in ``sys.modules`` *before* importing ``minimagen``:

  * ``einops_exts``            (check_shape, rearrange_many, repeat_many,
                                torch.EinopsToAndFrom)  -- pure re-layout helpers
  * ``resize_right.resize``    -- restated in oracle/resize_restated.py
  * ``torchvision.transforms`` -- names only (ToPILImage/Compose/ToTensor)
"""
from __future__ import annotations

import contextlib
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MINIMAGEN_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "minimagen"))


def _install_shims():
    import transformers  # noqa: F401  (must be imported before the torchvision stub exists)
    import torch
    from torch import nn
    from einops import rearrange, repeat

    if "einops_exts" not in sys.modules:
        ee = types.ModuleType("einops_exts")

        def check_shape(tensor, pattern, **kwargs):
            return rearrange(tensor, f"{pattern} -> {pattern}", **kwargs)

        def rearrange_many(tensors, pattern, **kwargs):
            return (rearrange(t, pattern, **kwargs) for t in tensors)

        def repeat_many(tensors, pattern, **kwargs):
            return (repeat(t, pattern, **kwargs) for t in tensors)

        ee.check_shape = check_shape
        ee.rearrange_many = rearrange_many
        ee.repeat_many = repeat_many

        eet = types.ModuleType("einops_exts.torch")

        class EinopsToAndFrom(nn.Module):
            def __init__(self, from_einops, to_einops, fn):
                super().__init__()
                self.from_einops = from_einops
                self.to_einops = to_einops
                self.fn = fn  # attribute name is part of the state-dict contract (".fn.")

            def forward(self, x, **kwargs):
                names = self.from_einops.split(" ")
                dims = dict(zip(names, x.shape))
                x = rearrange(x, f"{self.from_einops} -> {self.to_einops}")
                x = self.fn(x, **kwargs)
                return rearrange(x, f"{self.to_einops} -> {self.from_einops}", **dims)

        eet.EinopsToAndFrom = EinopsToAndFrom
        ee.torch = eet
        sys.modules["einops_exts"] = ee
        sys.modules["einops_exts.torch"] = eet

    if "resize_right" not in sys.modules:
        from . import resize_restated
        rr = types.ModuleType("resize_right")
        rr.resize = resize_restated.resize
        sys.modules["resize_right"] = rr

    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")

        class ToPILImage:
            def __call__(self, img):
                import numpy as np
                from PIL import Image
                arr = (img.detach().cpu().clamp(0, 1) * 255).round().to(torch.uint8)
                return Image.fromarray(np.ascontiguousarray(arr.permute(1, 2, 0).numpy()))

        class Compose:
            def __init__(self, transforms):
                self.transforms = transforms

            def __call__(self, x):
                for t in self.transforms:
                    x = t(x)
                return x

        class ToTensor:
            def __call__(self, pic):
                import numpy as np
                a = torch.from_numpy(np.asarray(pic).copy())
                return a.permute(2, 0, 1).float() / 255.0

        tvt.ToPILImage, tvt.Compose, tvt.ToTensor = ToPILImage, Compose, ToTensor
        tv.transforms = tvt
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tvt


_ref = None


def load_reference():
    """Returns a namespace with the reference's modules (Imagen, Unet, layers, ...)."""
    global _ref
    if _ref is not None:
        return _ref
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    _install_shims()
    # The reference package is called ``minimagen`` -- and so is the repo's drop-in alias package (minimagen/__init__.py re-exports
    # minimagen_amd).  Load the reference under a PRIVATE package name from its own directory (its modules import each other
    # relatively), so that "the reference" can never silently resolve to the implementation under test.
    import importlib.util
    pkg_dir = os.path.join(REFERENCE_ROOT, "minimagen")
    name = "_reference_minimagen"
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules[name] = pkg
    spec.loader.exec_module(pkg)
    ns = types.SimpleNamespace()
    ns.Imagen_mod = importlib.import_module(name + ".Imagen")
    ns.Unet_mod = importlib.import_module(name + ".Unet")
    ns.layers = importlib.import_module(name + ".layers")
    ns.helpers = importlib.import_module(name + ".helpers")
    ns.diffusion_model = importlib.import_module(name + ".diffusion_model")
    ns.t5 = importlib.import_module(name + ".t5")
    for m in (ns.Imagen_mod, ns.Unet_mod, ns.layers, ns.helpers, ns.diffusion_model, ns.t5):
        assert os.path.abspath(m.__file__).startswith(os.path.abspath(pkg_dir)), f"{m.__name__} was not loaded from the reference tree"
    ns.Imagen = ns.Imagen_mod.Imagen
    ns.Unet = ns.Unet_mod.Unet
    ns.GaussianDiffusion = ns.diffusion_model.GaussianDiffusion
    _ref = ns
    return ns


@contextlib.contextmanager
def injected_noise(seed: int):
    """Make ``torch.randn`` / ``torch.randn_like`` draw from one seeded CPU generator.

    The reference has no generator argument (Imagen.py:361,400,485); parity runs
    need the same noise stream in both implementations.  Draw order is the
    reference's own call order.
    """
    import torch
    gen = torch.Generator(device="cpu").manual_seed(seed)
    orig_randn, orig_randn_like = torch.randn, torch.randn_like

    def randn(*size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        dev = kw.get("device", None)
        out = orig_randn(*size, generator=gen, dtype=kw.get("dtype", torch.float32))
        return out.to(dev) if dev is not None else out

    def randn_like(t, **kw):
        return orig_randn(tuple(t.shape), generator=gen, dtype=t.dtype).to(t.device)

    torch.randn, torch.randn_like = randn, randn_like
    try:
        yield gen
    finally:
        torch.randn, torch.randn_like = orig_randn, orig_randn_like
