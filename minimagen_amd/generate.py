"""The real-world caller of ``Imagen.sample``: read a MinImagen *training directory*
(``parameters/*.json`` + ``state_dicts/`` or ``tmp/`` checkpoints), build the model, sample captions and write image files.
Mirrors the observable behaviour of minimagen/generate.py:12-173 (directory layout, file names, errors); sampling itself
runs on the HIP path of :class:`minimagen_amd.Imagen`."""
import json
import os
from datetime import datetime
from typing import List, Optional, Tuple

import torch

from .Imagen import Imagen
from .Unet import Unet


def _unet_index(name: str) -> int:
    return int(name.split("_")[1])


def load_params(directory: str) -> Tuple[List[dict], dict]:
    """generate.py:49-67: ``(unets_params, imagen_params)`` from ``<directory>/parameters``; the i-th list entry is the
    keyword dict of U-Net i (files ``unet_<i>_params_*.json``), the Imagen dict comes from ``imagen_params_*.json``."""
    pdir = os.path.join(directory, "parameters")
    names = os.listdir(pdir)

    def read(name):
        with open(os.path.join(pdir, name), "r") as fh:
            return json.load(fh)

    unet_files = sorted((n for n in names if n.startswith("unet_")), key=_unet_index)
    imagen_files = [n for n in names if n.startswith("imagen_")]
    if not imagen_files:
        raise IndexError(f"no imagen_*.json under {pdir}")          # generate.py:66 indexes [0] of an empty list
    return [read(n) for n in unet_files], read(imagen_files[0])


def _checkpoint_files(directory: str):
    """generate.py:95-119: ``state_dicts/`` (one file per U-Net, ``unet_<i>_state_*.pth``) wins; when it is empty fall back
    to the rolling checkpoints in ``tmp/`` (``unet_<i>_tmp.pth``); both empty is a ValueError."""
    for sub in ("state_dicts", "tmp"):
        folder = os.path.join(directory, sub)
        names = os.listdir(folder) if os.path.isdir(folder) else []
        if names:
            if sub == "tmp":
                print(f"\n\"state_dicts\" folder in {directory} is empty, using the most recent checkpoint from \"tmp\".\n")
            per_unet = {}
            for n in names:
                if n.startswith("unet_"):
                    per_unet.setdefault(_unet_index(n), n)       # first listing hit per U-Net, as the reference takes [0]
            count = max(per_unet) + 1
            return [os.path.join(folder, per_unet[i]) for i in range(count)]
    raise ValueError(f"Both \"/state_dicts\" and \"/tmp\" in {directory} are empty. "
                     f"Train the model to acquire state dictionaries for inference. ")


def load_minimagen(directory: str) -> Imagen:
    """generate.py:79-121: instantiate from the parameter files, then load each U-Net's checkpoint."""
    unets_params, imagen_params = load_params(directory)
    model = Imagen(unets=[Unet(**p) for p in unets_params], **imagen_params)
    where = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    for idx, path in enumerate(_checkpoint_files(directory)):
        model.unets[idx].load_state_dict(torch.load(path, map_location=where))
    # the reference leaves the module where it was built and samples there; this implementation has no CPU sampling path,
    # so the loaded model goes to the device the checkpoints were mapped to
    return model.to(where)


def _prepare_output(save_directory: str) -> str:
    """generate.py:12-30: ``<save_directory>/generated_images`` must be new or empty."""
    root = os.path.abspath(save_directory)
    img_dir = os.path.join(root, "generated_images")
    if not os.path.exists(img_dir):
        os.makedirs(img_dir)
    elif os.listdir(img_dir):
        raise FileExistsError(f"The directory {img_dir} already exists and is nonempty")
    return root


def sample_and_save(captions: list, *, minimagen: Optional[Imagen] = None, training_directory: Optional[str] = None,
                    sample_args: dict = {}, save_directory: Optional[str] = None, filetype: str = "png"):
    """generate.py:124-173: writes ``captions.txt`` (+ ``imagen_training_directory.txt``) into ``save_directory`` and
    ``generated_images/image_<caption index>.<filetype>``; exactly one of ``minimagen`` / ``training_directory``."""
    assert not (minimagen is None and training_directory is None), \
        "Must supply either a training directory or MinImagen instance."
    assert (minimagen is not None) ^ (training_directory is not None), \
        "Cannot supply both a MinImagen instance and a training directory"
    if save_directory is None:
        save_directory = datetime.now().strftime("generated_images_%Y%m%d_%H%M%S")
    root = _prepare_output(save_directory)
    with open(os.path.join(root, "captions.txt"), "w") as fh:
        fh.writelines(f"{c}\n" for c in captions)
    if training_directory is not None:
        with open(os.path.join(root, "imagen_training_directory.txt"), "w") as fh:
            fh.write(training_directory)
        minimagen = load_minimagen(training_directory)
    images = minimagen.sample(texts=captions, return_pil_images=True, **sample_args)
    for idx, im in enumerate(images):
        im.save(os.path.join(root, "generated_images", f"image_{idx}.{filetype}"))
