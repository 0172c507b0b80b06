"""Drop-in import path: ``minimagen.*`` resolves to the MI355X implementation in ``minimagen_amd`` so that code written against the
reference (``from minimagen.Imagen import Imagen``, ``from minimagen.Unet import Unet, Base, Super``, ``from minimagen.generate import
load_minimagen, sample_and_save``, ``from minimagen.training import get_minimagen_parser, MinimagenTrain, ...`` -- the reference's
inference.py / generate.py / train.py imports) runs unchanged.  No code lives here."""
import importlib
import sys

import minimagen_amd

__version__ = minimagen_amd.__version__
for _name in ("Imagen", "Unet", "diffusion_model", "generate", "helpers", "layers", "t5", "training", "optim"):
    _mod = importlib.import_module(f"minimagen_amd.{_name}")
    sys.modules[f"{__name__}.{_name}"] = _mod
    setattr(sys.modules[__name__], _name, _mod)
del _name, _mod
