"""The reference's training driver surface (minimagen/training.py) over the device training path: command-line parser, training /
validation loop with its on-disk layout, and the helpers train.py calls.  What is kept is the OBSERVABLE behaviour -- flag names and defaults,
``training_<timestamp>/{parameters,state_dicts,tmp}``, ``training_progess.txt`` [sic] and its lines, the checkpoint file names
``minimagen_amd.generate.load_minimagen`` reads back.  The reference's ``train.py`` does NOT run unchanged against this package: it calls
``ConceptualCaptions`` unconditionally (train.py:43-47; a network download, SURVEY 2: out of scope) and that raises here, and the
reference's ``MinimagenDataset`` / ``_Rescale`` are not rebuilt.  Everything else train.py imports from ``minimagen.training`` resolves;
``tests/test_training_loop.py::test_reference_train_flow_offline`` is that script's flow with ``SyntheticCaptions`` as the dataset.

The loop body is ``imagen(images, text_embeds=..., unet_number=k).backward()``: with the U-Nets on the GPU that is the HIP training graph of
``minimagen_amd.train_ops`` (every convolution, CrossEmbed and the folded cross-attention core forward and backward on the kernels of
csrc/), and ``minimagen_amd.optim.Adam`` steps all parameters in one launch.  Data: the Conceptual-Captions pipeline of the reference
(Hugging Face ``datasets`` + image download by URL, training.py:214-313) needs the network and is not rebuilt; ``MinimagenCollator`` (pure
tensor work) is, and ``SyntheticCaptions`` stands in for the ``-test`` flow offline."""
import inspect
import json
import os
import signal
from argparse import ArgumentParser
from contextlib import contextmanager

import torch
import torch.nn.functional as F
import torch.utils.data

from . import Unet as _Unet
from .helpers import exists


def get_minimagen_parser() -> ArgumentParser:
    """training.py:178-211: the same flags, destinations, types and defaults"""
    ap = ArgumentParser()
    for short, name, typ, default, text in (
            ("-p", "PARAMETERS", str, None, "Parameters directory to load Imagen from"),
            ("-n", "NUM_WORKERS", int, 0, "Number of workers for DataLoader"),
            ("-b", "BATCH_SIZE", int, 2, "Batch size"),
            ("-mw", "MAX_NUM_WORDS", int, 64, "Maximum number of words allowed in a caption"),
            ("-s", "IMG_SIDE_LEN", int, 128, "Side length of square Imagen output images"),
            ("-e", "EPOCHS", int, 5, "Number of training epochs"),
            ("-t5", "T5_NAME", str, "t5_base", "Name of T5 encoder to use"),
            ("-f", "TRAIN_VALID_FRAC", float, 0.9, "Fraction of dataset to use for training (vs. validation)"),
            ("-t", "TIMESTEPS", int, 1000, "Number of timesteps in Diffusion process"),
            ("-lr", "OPTIM_LR", float, 0.0001, "Learning rate for Adam optimizer"),
            ("-ai", "ACCUM_ITER", int, 1, "Number of batches for gradient accumulation"),
            ("-cn", "CHCKPT_NUM", int, 500, "Checkpointing batch number interval"),
            ("-vn", "VALID_NUM", int, None, "Number of validation images to use. If None, uses full amount from train/valid split"),
            ("-rd", "RESTART_DIRECTORY", str, None, "Training directory to resume training from if restarting.")):
        ap.add_argument(short, "--" + name, dest=name, help=text, default=default, type=typ)
    ap.add_argument("-test", "--TESTING", dest="TESTING", help="Whether to test with smaller dataset", action="store_true")
    ap.set_defaults(TESTING=False)
    return ap


class MinimagenCollator:
    """training.py:59-90: drop items whose image could not be fetched, pad every caption's encoding / mask to the longest of the batch,
    move to ``device``, collate.  Returns None for an empty batch (the training loop skips it)."""

    def __init__(self, device):
        self.device = device

    def __call__(self, batch):
        batch = [b for b in batch if b is not None and b.get("image") is not None]
        if not batch:
            return None
        longest = max(b["mask"].shape[1] for b in batch)
        out = []
        for b in batch:
            pad = longest - b["mask"].shape[1]
            mask, enc = torch.squeeze(b["mask"]), torch.squeeze(b["encoding"])
            if pad > 0:
                mask, enc = F.pad(mask, (0, pad), "constant", 0), F.pad(enc, (0, 0, 0, pad), "constant", 0.)
            out.append({"image": b["image"].to(self.device), "encoding": enc.to(self.device), "mask": mask.to(self.device)})
        return torch.utils.data.dataloader.default_collate(out)


class SyntheticCaptions(torch.utils.data.Dataset):
    """Offline stand-in for ConceptualCaptions in the ``-test`` flow: ``n`` smooth random images of ``side`` pixels with random caption
    embeddings of the encoder's width (items shaped like MinimagenDataset's, training.py:236-269)."""

    def __init__(self, n: int, side: int, embed_dim: int, max_words: int = 32, seed: int = 0):
        g = torch.Generator().manual_seed(seed)
        low = torch.rand(n, 3, 8, 8, generator=g)
        self.images = F.interpolate(low, size=(side, side), mode="bilinear", align_corners=False).clamp(0, 1)
        self.lengths = torch.randint(3, max_words + 1, (n,), generator=g)
        self.enc = [torch.randn(1, int(L_), embed_dim, generator=g) for L_ in self.lengths]

    def __len__(self):
        return self.images.shape[0]

    def __getitem__(self, i):
        return {"image": self.images[i], "encoding": self.enc[i], "mask": torch.ones(1, int(self.lengths[i]), dtype=torch.bool)}


def ConceptualCaptions(args, smalldata=False, testset=False):
    """training.py:272-313 downloads Conceptual Captions through ``datasets`` and fetches every image by URL.  That pipeline is network I/O
    outside the accelerated path; it is not rebuilt here."""
    raise NotImplementedError("the Conceptual-Captions download pipeline (minimagen/training.py:214-313) is not part of this package: "
                              "build a torch Dataset of {'image', 'encoding', 'mask'} items (see SyntheticCaptions, MinimagenCollator)")


def get_minimagen_dl_opts(device):
    """training.py:316-322"""
    return {"shuffle": True, "drop_last": True, "collate_fn": MinimagenCollator(device)}


class _Timeout:
    """training.py:325-341: SIGALRM-based time limit around one batch (POSIX main thread only; elsewhere it is inert)"""

    class _Timeout(Exception):
        pass

    def __init__(self, seconds):
        self.seconds = int(seconds)

    def _handler(self, *a):
        raise _Timeout._Timeout()

    def __enter__(self):
        self.armed = hasattr(signal, "SIGALRM")
        if self.armed:
            try:
                signal.signal(signal.SIGALRM, self._handler)
                signal.alarm(self.seconds)
            except ValueError:                 # not the main thread
                self.armed = False

    def __exit__(self, *a):
        if self.armed:
            signal.alarm(0)


def _progress(training_dir, text):
    with training_dir():
        with open("training_progess.txt", "a") as fh:
            fh.write(text)


def _save_tmp(training_dir, imagen, n_unets):
    with training_dir("tmp"):
        for i in range(n_unets):
            torch.save(imagen.unets[i].state_dict(), f"unet_{i}_tmp.pth")


def MinimagenTrain(timestamp, args, unets, imagen, train_dataloader, valid_dataloader, training_dir, optimizer, timeout=60, fail_fast=False):
    """training.py:344-478.  Per batch: for every U-Net of the cascade ``imagen(images, text_embeds, text_masks, unet_number)`` ->
    ``backward`` -> gradient-norm clip at 50 over ALL parameters; optimiser step every ACCUM_ITER batches (and at the last batch); every
    CHCKPT_NUM batches: rolling checkpoints in ``tmp/``, running / batch losses, a validation pass, best-so-far state dicts in ``state_dicts/``.
    A batch that raises leaves a note and the latest state dicts in ``tmp/`` and the loop goes on with the next batch, as the reference's
    does (``fail_fast=True``, not in the reference, re-raises instead); one that exceeds ``timeout`` seconds is skipped."""
    n = len(unets)
    best = [torch.tensor(9999999.) for _ in range(n)]
    params = [p for p in imagen.parameters()]

    def validate(epoch, batch_num, running, losses):
        _progress(training_dir, f'{"-" * 10}Checkpoint created at batch number {batch_num}{"-" * 10}\n')
        _save_tmp(training_dir, imagen, n)
        avg = [r / max(batch_num, 1) for r in running]
        _progress(training_dir, f"U-Nets Avg Train Losses Epoch {epoch + 1} Batch {batch_num}: {[round(float(v), 3) for v in avg]}\n"
                                f"U-Nets Batch Train Losses Epoch {epoch + 1} Batch {batch_num}: {[round(float(v), 3) for v in losses]}\n")
        imagen.train(False)
        vsum = [0. for _ in range(n)]
        print(f'\n{"-" * 10}Validation...{"-" * 10}')
        with torch.no_grad():
            for vb in valid_dataloader:
                if not vb:
                    continue
                for k in range(n):
                    vsum[k] = vsum[k] + imagen(vb["image"], text_embeds=vb["encoding"], text_masks=vb["mask"], unet_number=k + 1).detach()
        vavg = [torch.as_tensor(v / max(len(valid_dataloader), 1)).cpu() for v in vsum]
        for k, v in enumerate(vavg):
            print(f"Unet {k} avg validation loss: ", v)
            if v < best[k]:
                best[k] = v
                with training_dir("state_dicts"):
                    torch.save(imagen.unets[k].state_dict(), f"unet_{k}_state_{timestamp}.pth")
        _progress(training_dir, f"U-Nets Avg Valid Losses: {[round(float(v), 3) for v in vavg]}\n"
                                f"U-Nets Best Valid Losses: {[round(float(v), 3) for v in best]}\n\n")
        imagen.train(True)

    for epoch in range(args.EPOCHS):
        print(f'\n{"-" * 20} EPOCH {epoch + 1} {"-" * 20}')
        _progress(training_dir, f'{"-" * 20} EPOCH {epoch + 1} {"-" * 20}\n')
        imagen.train(True)
        running = [0. for _ in range(n)]
        print(f'\n{"-" * 10}Training...{"-" * 10}')
        for batch_num, batch in enumerate(train_dataloader):
            if not batch:
                continue
            try:
                with _Timeout(timeout):
                    losses = [0. for _ in range(n)]
                    for k in range(n):
                        loss = imagen(batch["image"], text_embeds=batch["encoding"], text_masks=batch["mask"], unet_number=k + 1)
                        losses[k] = loss.detach()
                        running[k] = running[k] + loss.detach()
                        loss.backward()
                        torch.nn.utils.clip_grad_norm_(params, 50)
                    if args.ACCUM_ITER == 1 or (batch_num % args.ACCUM_ITER == 0) or (batch_num + 1 == len(train_dataloader)):
                        optimizer.step()
                        optimizer.zero_grad()
                    if batch_num % args.CHCKPT_NUM == 0:
                        validate(epoch, batch_num, running, losses)
            except _Timeout._Timeout:
                pass
            except Exception as exc:
                _progress(training_dir, f"\n\nTRAINING ABORTED AT EPOCH {epoch}, BATCH NUMBER {batch_num} with exception {exc}. MOST RECENT STATE "
                                        f"DICTS SAVED TO ./tmp IN TRAINING FOLDER")
                _save_tmp(training_dir, imagen, n)          # training.py:470-478: note, tmp state dicts, then ON to the next batch
                if fail_fast:
                    raise


def _parse_training_file(params_dir, keep):
    name = [f for f in os.listdir(params_dir) if f.startswith("training_")][0]
    found = {}
    with open(os.path.join(params_dir, name)) as fh:
        for line in fh:
            if line.startswith("--") and "=" in line:
                key, val = line[2:].rstrip("\n").split("=", 1)
                if key in keep:
                    found[key] = int(val) if val.lstrip("-").isdigit() else val
    return found


def load_restart_training_parameters(args, justparams=False):
    """training.py:481-517: MAX_NUM_WORDS, IMG_SIDE_LEN, T5_NAME, TIMESTEPS of the ORIGINAL training override the command line when a
    training is resumed (``--RESTART_DIRECTORY``) or parameters are loaded (``--PARAMETERS``)"""
    where = args.PARAMETERS if justparams else os.path.join(args.RESTART_DIRECTORY, "parameters")
    args.__dict__.update(_parse_training_file(where, ("MAX_NUM_WORDS", "IMG_SIDE_LEN", "T5_NAME", "TIMESTEPS")))
    return args


def load_testing_parameters(args):
    """training.py:520-556: the low-cost settings of ``-test``"""
    args.__dict__.update(dict(BATCH_SIZE=2, MAX_NUM_WORDS=32, IMG_SIDE_LEN=128, EPOCHS=2, T5_NAME="t5_small", TRAIN_VALID_FRAC=0.5,
                              TIMESTEPS=25, OPTIM_LR=0.0001))
    return args


def create_directory(dir_path):
    """training.py:559-581: make ``dir_path`` with ``parameters`` / ``state_dicts`` / ``tmp`` and return a context manager that enters it (or a
    sub-directory) for the duration of a ``with`` block"""
    home = os.getcwd()
    dir_path = os.path.abspath(dir_path)
    if not os.path.exists(dir_path):
        for sub in ("", "parameters", "state_dicts", "tmp"):
            os.makedirs(os.path.join(dir_path, sub), exist_ok=True)

    @contextmanager
    def enter(subpath=""):
        os.chdir(os.path.join(dir_path, subpath))
        try:
            yield
        finally:
            os.chdir(home)

    return enter


def get_model_size(imagen) -> float:
    """training.py:584-593: parameters + buffers in MB"""
    return sum(t.nelement() * t.element_size() for t in list(imagen.parameters()) + list(imagen.buffers())) / 1024 ** 2


def save_training_info(args, timestamp, unets_params, imagen_params, model_size, training_dir):
    """training.py:596-625: ``parameters/training_parameters_<ts>.txt`` (one ``--NAME=value`` line per argument), the model size in the
    progress file, ``parameters/unet_<i>_params_<ts>.json`` and ``parameters/imagen_params_<ts>.json``"""
    with training_dir("parameters"):
        with open(f"training_parameters_{timestamp}.txt", "w") as fh:
            fh.writelines(f"--{k}={v}\n" for k, v in args.__dict__.items())
    head = f"STARTED FROM CHECKPOINT {args.RESTART_DIRECTORY}\n" if getattr(args, "RESTART_DIRECTORY", None) is not None else ""
    _progress(training_dir, head + f"model size: {model_size:.3f}MB\n\n")
    with training_dir("parameters"):
        for i, prm in enumerate(unets_params):
            with open(f"unet_{i}_params_{timestamp}.json", "w") as fh:
                json.dump(prm, fh, indent=4)
        with open(f"imagen_params_{timestamp}.json", "w") as fh:
            json.dump(imagen_params, fh, indent=4)


def get_model_params(parameters_dir):
    """training.py:628-657: ``(unets_params, imagen_params)`` from a ``parameters`` directory"""
    names = os.listdir(parameters_dir)
    unet_files = sorted((f for f in names if f.startswith("unet_")), key=lambda f: int(f.split("_")[1]))
    imagen_file = [f for f in names if f.startswith("imagen")][-1]

    def read(f):
        with open(os.path.join(parameters_dir, f)) as fh:
            return json.load(fh)
    return [read(f) for f in unet_files], read(imagen_file)


def get_default_args(obj):
    """training.py:660-672: keyword defaults of a callable; for the U-Net presets (Base, Super, BaseTest, SuperTest) the Unet defaults
    overlaid with the preset's ``defaults``"""
    if inspect.isclass(obj) and issubclass(obj, _Unet.Unet) and obj is not _Unet.Unet:
        return {**get_default_args(_Unet.Unet), **obj.defaults}
    return {k: v.default for k, v in inspect.signature(obj).parameters.items() if v.default is not inspect.Parameter.empty}
