"""``Imagen`` with the reference's constructor and ``sample`` API (minimagen/Imagen.py), whose cascaded
reverse-diffusion loop runs as HIP kernels replayed from a HIP graph on MI355X."""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Literal, Tuple, Union

import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib as L
from .Unet import Unet
from .diffusion_model import GaussianDiffusion
from .helpers import (cast_tuple, cubic_taps, default, eval_decorator, exists, module_device, normalize_neg_one_to_one, quantile_rank,
                      resize_image_to)
from .t5 import get_encoded_dim, t5_encode_text

# the sampler tail of images too large for one workgroup (the super-resolution stages) as ONE launch of cooperating workgroups
# (mi_sampler_step_group_fwd) instead of five launches; 0: the separate kernels
SAMPLER_GROUP = int(os.environ.get("MINIMAGEN_SAMPLER_GROUP", "1"))
# ... for at most this many workgroups per image (8: up to 256^2).  The workgroups of an image wait for each other, so a launch is only safe
# next to OTHER launches of the same kind while the partial groups of all of them fit the chip beside everything else that is resident: two
# 1024^2 tails (128 workgroups of 1024 work-items per image, one per CU) in flight on two call lanes starved each other's last image until the
# bounded spin gave up (profiles/r04_sampler_group_config5.txt) -- large images keep the separate kernels
SAMPLER_GROUP_MAX = int(os.environ.get("MINIMAGEN_SAMPLER_GROUP_MAX", "8"))
SAMPLE_LANES = max(1, int(os.environ.get("MINIMAGEN_SAMPLE_LANES", "2")))     # independent call lanes of sample(_async=True)
# 1: a synchronous sample() waits on the HOST for its last stage and checks the cooperative kernels' status words before it returns (the
# default defers the check to the next API entry: the failed call's images are NaN -- fail-stop -- so nothing plausible-but-wrong escapes)
STRICT_STATUS = os.environ.get("MINIMAGEN_STRICT_STATUS", "0") != "0"
_STAGE_STREAMS = {}          # (device, lanes, stages, priority mode) -> [lane][stage] HIP streams, process-wide (see sample())


class Imagen(nn.Module):
    """minimagen/Imagen.py:22-131."""

    def __init__(
            self,
            unets: Union[Unet, List[Unet], Tuple[Unet, ...]],
            *,
            text_encoder_name: str,
            image_sizes: Union[int, List[int], Tuple[int, ...]],
            text_embed_dim: int = None,
            channels: int = 3,
            timesteps: Union[int, List[int], Tuple[int, ...]] = 1000,
            cond_drop_prob: float = 0.1,
            loss_type: Literal["l1", "l2", "huber"] = 'l2',
            lowres_sample_noise_level: float = 0.2,
            auto_normalize_img: bool = True,
            dynamic_thresholding_percentile: float = 0.9,
            only_train_unet_number: int = None
    ):
        super().__init__()
        if loss_type not in ('l1', 'l2', 'huber'):
            raise NotImplementedError()
        self.loss_type = loss_type
        self.channels = channels
        unets = cast_tuple(unets)
        num_unets = len(unets)
        self.noise_schedulers = nn.ModuleList([GaussianDiffusion(timesteps=t) for t in cast_tuple(timesteps, num_unets)])
        # built from the RAW argument like Imagen.py:78 (so only an int works, as in the reference)
        self.lowres_noise_schedule = GaussianDiffusion(timesteps=timesteps)
        self.text_encoder_name = text_encoder_name
        self.text_embed_dim = default(text_embed_dim, lambda: get_encoded_dim(text_encoder_name))
        self.unet_being_trained_index = -1
        self.only_train_unet_number = only_train_unet_number
        self.unets = nn.ModuleList([])
        for ind, one_unet in enumerate(unets):
            assert isinstance(one_unet, Unet)
            one_unet = one_unet._cast_model_parameters(lowres_cond=not ind == 0, text_embed_dim=self.text_embed_dim,
                                                       channels=self.channels, channels_out=self.channels)
            self.unets.append(one_unet)
        self.image_sizes = cast_tuple(image_sizes)
        assert num_unets == len(image_sizes), f'you did not supply the correct number of u-nets ({len(self.unets)}) for resolutions {image_sizes}'
        self.sample_channels = cast_tuple(self.channels, num_unets)
        self.lowres_sample_noise_level = lowres_sample_noise_level
        self.cond_drop_prob = cond_drop_prob
        self.can_classifier_guidance = cond_drop_prob > 0.
        self.auto_normalize_img = auto_normalize_img
        self.input_image_range = (0. if auto_normalize_img else -1., 1.)
        self.dynamic_thresholding_percentile = dynamic_thresholding_percentile
        self.register_buffer('_temp', torch.tensor([0.]), persistent=False)
        self.to(next(self.unets.parameters()).device)
        self._sampler_state = {}

    @property
    def device(self) -> torch.device:
        return self._temp.device

    def _reset_unets_all_one_device(self, device: torch.device = None):
        """Imagen.py:205-219.  All U-Nets stay resident on the GPU (288 GB of HBM: nothing is swapped to the host)."""
        device = default(device, self.device)
        self.unets = nn.ModuleList([*self.unets])
        self.unets.to(device)
        self.unet_being_trained_index = -1

    def state_dict(self, *args, **kwargs):
        self._reset_unets_all_one_device()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._reset_unets_all_one_device()
        return super().load_state_dict(*args, **kwargs)

    # ------------------------------------------------------------------ training (Imagen.py:512-650)
    def _get_unet(self, unet_number: int) -> Unet:
        """Imagen.py:221-259.  All U-Nets stay on the GPU (288 GB of HBM); only the bookkeeping of the reference is kept."""
        assert 0 < unet_number <= len(self.unets)
        self.unet_being_trained_index = unet_number - 1
        return self.unets[unet_number - 1]

    def _p_losses(self, unet: Unet, x_start, times, *, noise_scheduler: GaussianDiffusion, lowres_cond_img=None, lowres_aug_times=None,
                  text_embeds=None, text_mask=None, noise=None):
        """Imagen.py:512-573: corrupt x_0 with q_sample, predict the noise, loss against the true noise"""
        noise = default(noise, lambda: torch.randn_like(x_start))
        norm = normalize_neg_one_to_one if self.auto_normalize_img else (lambda v: v)
        x_start = norm(x_start)
        x_noisy = noise_scheduler.q_sample(x_start=x_start, t=times, noise=noise)
        lowres_noisy = None
        if exists(lowres_cond_img):
            lowres_aug_times = default(lowres_aug_times, times)
            lowres_cond_img = norm(lowres_cond_img)
            lowres_noisy = self.lowres_noise_schedule.q_sample(x_start=lowres_cond_img, t=lowres_aug_times, noise=torch.randn_like(lowres_cond_img))
        pred = unet.forward(x_noisy, times, text_embeds=text_embeds, text_mask=text_mask, lowres_noise_times=lowres_aug_times,
                            lowres_cond_img=lowres_noisy, cond_drop_prob=self.cond_drop_prob)
        return {'l1': F.l1_loss, 'l2': F.mse_loss, 'huber': F.smooth_l1_loss}[self.loss_type](pred, noise)

    def forward(self, images, texts: List[str] = None, text_embeds: torch.Tensor = None, text_masks: torch.Tensor = None, unet_number: int = None):
        """Imagen.py:575-650: the training loss of ONE U-Net of the cascade on a batch of images + captions.  The U-Net runs through its
        differentiable training graph when it is in train mode with autograd on (Unet._forward_train: HIP kernels for the convolution stack,
        forward and backward, torch ops for the rest -- minimagen_amd/train_ops.py) and through the HIP inference engine otherwise
        (evaluation of the loss)."""
        assert not (len(self.unets) > 1 and not exists(unet_number)), \
            f'you must specify which unet you want trained, from a range of 1 to {len(self.unets)}, if you are training cascading DDPM (multiple unets)'
        unet_number = default(unet_number, 1)
        assert not exists(self.only_train_unet_number) or self.only_train_unet_number == unet_number, \
            f'you can only train on unet #{self.only_train_unet_number}'
        k = unet_number - 1
        unet, noise_scheduler, target = self._get_unet(unet_number), self.noise_schedulers[k], self.image_sizes[k]
        prev = self.image_sizes[k - 1] if k > 0 else None
        assert images.dim() == 4 and images.shape[1] == self.channels, f'images must be (b, {self.channels}, h, w)'
        b, _, h, w = images.shape
        assert h >= target and w >= target
        times = noise_scheduler._sample_random_times(b, device=images.device)
        if exists(texts) and not exists(text_embeds):
            assert len(texts) == len(images), 'number of text captions does not match up with the number of images given'
            text_embeds, text_masks = t5_encode_text(texts, name=self.text_encoder_name)
            text_embeds, text_masks = text_embeds.to(images.device), text_masks.to(images.device)
        assert exists(text_embeds), 'text or text encodings must be passed into decoder'
        assert text_embeds.shape[-1] == self.text_embed_dim, f'invalid text embedding dimension being passed in (should be {self.text_embed_dim})'
        lowres_cond_img = lowres_aug_times = None
        if exists(prev):
            lowres_cond_img = resize_image_to(images, prev, clamp_range=self.input_image_range, pad_mode='reflect')
            lowres_cond_img = resize_image_to(lowres_cond_img, target, clamp_range=self.input_image_range, pad_mode='reflect')
            lowres_aug_times = self.lowres_noise_schedule._sample_random_times(1, device=images.device).expand(b)
        images = resize_image_to(images, target)
        return self._p_losses(unet, images, times, text_embeds=text_embeds, text_mask=text_masks, noise_scheduler=noise_scheduler,
                              lowres_cond_img=lowres_cond_img, lowres_aug_times=lowres_aug_times)

    # ------------------------------------------------------------------ sampling
    def _stage_state(self, ws, sched: GaussianDiffusion, B: int, n: int):
        # the state (and the step graphs cached on it) lives on the workspace, so it dies with the buffers it points into
        store = ws.__dict__.setdefault("sampler_state", {})
        key = sched.num_timesteps
        st = store.get(key)
        if st is None:
            dev = ws.dev
            st = type("StageState", (), {})()
            st.coef = sched.sampler_coef_table().to(dev).contiguous()
            st.t_state = torch.zeros(1, dtype=torch.int32, device=dev)
            st.x0 = torch.empty(B, n, dtype=torch.float32, device=dev)
            st.hist = torch.zeros(3 * B * 2 * 2048, dtype=torch.int32, device=dev)
            st.s_q = torch.zeros(B, dtype=torch.float32, device=dev)
            st.v_q = torch.zeros(B, 2, dtype=torch.float32, device=dev)
            store[key] = st
        return st

    def _stage_begin(self, unet: Unet, shape, *, noise_scheduler: GaussianDiffusion, ws, noise_fn: Callable = None, seed: int = 0,
                     sample0: int = 0, stage: int = 0):
        """Everything of a stage's loop that does not depend on the PREVIOUS stage's image: x_T (Imagen.py:400), the device-resident
        timestep, the per-step conditioning tables of all T steps.  sample() issues it for every stage before the first stage's loop,
        so that a later stage's stream has it done while it waits for its low-resolution input (on-device noise only: injected noise
        must be drawn in the reference's order)."""
        lib = L.lib()
        stream = L.current_stream()
        eng = unet.engine()
        B, Cc, H, W = shape
        n = Cc * H * W
        T = noise_scheduler.num_timesteps
        st = self._stage_state(ws, noise_scheduler, B, n)
        noise_dev = None
        if noise_fn is not None:
            ws.x.copy_(noise_fn(shape))                                          # Imagen.py:400
            noise_dev = torch.stack([noise_fn(shape) for _ in range(T)]).to(ws.dev).contiguous()   # Imagen.py:361, in step order
        else:
            L.check(lib.mi_randn_fill(L.ptr(ws.x), B, n, seed, sample0, (stage << 20) | (1 << 19) | 1, stream), "mi_randn_fill")
        L.check(lib.mi_step_set(L.ptr(st.t_state), L.ptr(ws.times), B, T - 1, stream), "mi_step_set")
        eng.prepare_step_tables(ws, T, st.t_state, stream)       # (timestep, text)-only conditioning of all T steps, once
        return st, noise_dev

    def _p_sample_loop(self, unet: Unet, shape, *, noise_scheduler: GaussianDiffusion, ws, cond_scale: float,
                       noise_fn: Callable = None, seed: int = 0, sample0: int = 0, stage: int = 0, use_graph: bool = True, begun=None):
        """Imagen.py:373-420 + :329-370 + :261-326: T replays of
        [U-Net (both guidance halves) -> CFG combine + x0 -> dynamic-threshold quantile -> posterior draw -> t -= 1]."""
        lib = L.lib()
        stream = L.current_stream()
        eng = unet.engine()
        B, Cc, H, W = shape
        n = Cc * H * W
        T = noise_scheduler.num_timesteps
        two = ws.B2 != ws.B
        if begun is None:
            begun = self._stage_begin(unet, shape, noise_scheduler=noise_scheduler, ws=ws, noise_fn=noise_fn, seed=seed, sample0=sample0, stage=stage)
        st, noise_dev = begun

        k_lo, k_hi, w = quantile_rank(n, self.dynamic_thresholding_percentile)
        fused = os.environ.get("MINIMAGEN_SAMPLER_FUSED", "1") != "0"
        small = n <= 16384 and fused                                    # MI_SAMPLER_SMALL_N: the whole tail in one launch of one workgroup per image
        # ... or of <= SAMPLER_GROUP_MAX cooperating workgroups per image.  A stage state whose grouped launch ever failed (fail-stop:
        # NaN images + the sticky error word, see _poll_status) keeps the separate kernels from then on
        group = (not small) and fused and bool(SAMPLER_GROUP) and 0 < lib.mi_sampler_group_size(n) <= SAMPLER_GROUP_MAX \
            and not getattr(st, "group_failed", False)
        if getattr(st, "group_heal", False):
            st.group_sync.zero_()               # stream-ordered behind every launch queued on this lane: ticket, counters, histograms, error word
            st.group_heal = False
        # the captured graph of one denoising step is cached per (workspace, guidance, threshold, noise mode, shard offset, tail kind):
        # the Philox seed lives in device memory, so replays of later sample() calls need no re-capture
        gkey = (float(cond_scale), two, k_lo, k_hi, w, sample0, stage, T, noise_dev is None, group)
        cached = getattr(st, "graphs", None)
        if cached is None:
            cached = st.graphs = {}
        entry = cached.get(gkey) if (use_graph and noise_dev is None) else None
        if noise_dev is None:
            if not hasattr(st, "seed_dev"):
                st.seed_dev = torch.zeros(1, dtype=torch.int64, device=ws.dev)
            st.seed_dev.fill_(int(seed) & 0x7FFFFFFFFFFFFFFF)
        if entry is None:
            # the radix select's first pass rides on the kernel that produces x0, and the histograms clean themselves: st.hist is
            # zero on allocation and every mi_quantile_fwd leaves it zeroed again
            cp = L.MiCfgX0Params(B, n, L.ptr(ws.pred), 1 if two else 0, float(cond_scale), L.ptr(ws.x), L.ptr(st.coef), L.ptr(st.t_state), 0, L.ptr(st.x0),
                                 L.ptr(st.hist))
            qp = L.MiQuantileParams(B, n, L.ptr(st.x0), k_lo, k_hi, w, L.ptr(st.hist), L.ptr(st.s_q), L.ptr(st.v_q), 1, 1)
            pp = L.MiPosteriorParams(B, n, T, L.ptr(st.x0), L.ptr(st.s_q), L.ptr(ws.x), L.ptr(st.coef), L.ptr(st.t_state),
                                     L.ptr(noise_dev), int(seed) & 0x7FFFFFFFFFFFFFFF, sample0, stage << 20,
                                     L.ptr(st.seed_dev) if noise_dev is None else 0)

            if group and not hasattr(st, "group_sync"):
                st.group_sync = torch.zeros(lib.mi_sampler_group_sync_bytes(B, n), dtype=torch.uint8, device=ws.dev)   # this workspace's launches only
                st.group_err_host = torch.zeros(1, dtype=torch.int32).pin_memory() if L.backend() == "hip-gfx950" else torch.zeros(1, dtype=torch.int32)
            offsets = eng.step_offsets_supported(ws)          # the k-th step of a graph addresses *t_state - k; one advance per graph

            def tail_params(k):
                c_, p_ = L.MiCfgX0Params.from_buffer_copy(cp), L.MiPosteriorParams.from_buffer_copy(pp)
                c_.t_off = p_.t_off = k
                if small or group:
                    c_.x0 = c_.hist0 = 0                      # x0 stays in registers, the histograms in LDS (and per-image counters)
                return c_, p_
            tails = {}

            def one_step(k=0, advance=1):
                eng.run_step(ws, stream, t_off=k)
                if k not in tails:
                    tails[k] = tail_params(k)
                c_, p_ = tails[k]
                if small:
                    L.check(lib.mi_sampler_step_small_fwd(C.byref(c_), C.byref(qp), C.byref(p_), stream), "mi_sampler_step_small_fwd")
                elif group:
                    L.check(lib.mi_sampler_step_group_fwd(C.byref(c_), C.byref(qp), C.byref(p_), L.ptr(st.group_sync), stream), "mi_sampler_step_group_fwd")
                else:
                    L.check(lib.mi_cfg_x0_fwd(C.byref(c_), stream), "mi_cfg_x0_fwd")
                    L.check(lib.mi_quantile_fwd(C.byref(qp), stream), "mi_quantile_fwd")
                    L.check(lib.mi_posterior_fwd(C.byref(p_), stream), "mi_posterior_fwd")
                if advance == 1:
                    L.check(lib.mi_step_advance(L.ptr(st.t_state), L.ptr(ws.times), B, stream), "mi_step_advance")
                elif advance > 1:
                    L.check(lib.mi_step_advance_by(L.ptr(st.t_state), L.ptr(ws.times), B, advance, stream), "mi_step_advance_by")
            # several denoising steps per captured graph: one replay boundary (~9 us of idle GPU) per `per` steps instead of per step
            cap = int(os.environ.get("MINIMAGEN_STEPS_PER_GRAPH", "5"))
            per = next(k for k in (5, 4, 3, 2, 1) if k <= cap and T % k == 0)
            entry = dict(step=one_step, graph=None, keep=(cp, qp, pp, tails), per=per)
            if use_graph:
                L.check(lib.mi_graph_begin(stream), "mi_graph_begin")
                try:
                    for k in range(per):
                        if offsets:
                            one_step(k, per if k == per - 1 else 0)
                        else:
                            one_step()
                finally:
                    g = C.c_void_p()
                    rc = lib.mi_graph_end(stream, C.byref(g))
                L.check(rc, "mi_graph_end")
                entry["graph"] = g
                if noise_dev is None:
                    while len(cached) >= 8:                      # bounded: one exec per (guidance, threshold, shard offset) combination
                        old = cached.pop(next(iter(cached)))
                        # replays of the evicted exec may still be queued on a stage stream (sample() never host-syncs): drain the
                        # device before the handle goes (rare: a 9th distinct (guidance, threshold, shard) combination)
                        if L.backend() == "hip-gfx950":
                            torch.cuda.synchronize(ws.dev)
                        lib.mi_graph_destroy(old["graph"])
                    cached[gkey] = entry
        if use_graph:
            try:
                for _ in range(T // entry["per"]):
                    L.check(lib.mi_graph_launch(entry["graph"], stream), "mi_graph_launch")
            finally:
                if noise_dev is not None:          # one-off graph (injected noise buffer): the exec must outlive its replays
                    if L.backend() == "hip-gfx950":
                        torch.cuda.current_stream().synchronize()
                    lib.mi_graph_destroy(entry["graph"])
        else:
            for _ in range(T):
                entry["step"]()
        img = torch.empty(shape, dtype=torch.float32, device=ws.dev)
        L.check(lib.mi_finalize_images(L.ptr(ws.x), L.ptr(img), B * n, 1 if self.auto_normalize_img else 0, stream), "mi_finalize_images")
        if group:
            # the grouped tail's sticky error word travels to pinned host memory behind the stage's last launch (no host synchronisation):
            # _poll_status reads it once this call's completion event has fired
            st.group_err_host.copy_(st.group_sync[8:12].view(torch.int32), non_blocking=True)
            self.__dict__.setdefault("_status_stages", []).append((st, stage, (B, H, W)))
        return img

    def _lowres_conditioning(self, img, image_size: int, ws, lowres_noise_level: float, noise_fn, seed, sample0, stage):
        """Imagen.py:479-485 + :393: cubic resize (reflect pad) -> q_sample at int(T*level) -> *2-1."""
        lib = L.lib()
        stream = L.current_stream()
        B, Cc, Hin, Win = img.shape
        t_low = int(self.lowres_noise_schedule.num_timesteps * lowres_noise_level)          # diffusion_model.py:68-69
        ws.lowres_times.fill_(t_low)
        if Hin != image_size:
            # tap tables: built and uploaded once per (workspace, source size) -- an upload from pageable host memory per call would
            # block the host behind the previous call still running on this stage's stream
            cache = ws.__dict__.setdefault("resize_tabs", {})
            if (Hin, Win) not in cache:
                _, idx_h, w_h = cubic_taps(Hin, image_size)
                _, idx_w, w_w = cubic_taps(Win, image_size)
                cache[(Hin, Win)] = ([t.to(ws.dev) for t in (idx_h, w_h, idx_w, w_w)], idx_h.shape[1], idx_w.shape[1])
            tabs, kh, kw = cache[(Hin, Win)]
            up = torch.empty(B, Cc, image_size, image_size, dtype=torch.float32, device=ws.dev)
            rp = L.MiResizeParams(B * Cc, Hin, Win, image_size, image_size, kh, kw, L.ptr(img), L.ptr(up),
                                  L.ptr(tabs[0]), L.ptr(tabs[1]), L.ptr(tabs[2]), L.ptr(tabs[3]))
            L.check(lib.mi_resize_fwd(C.byref(rp), stream), "mi_resize_fwd")
            ws.resize_keepalive = tabs
        else:
            up = img
        n = Cc * image_size * image_size
        if noise_fn is not None:
            noise = noise_fn(up.shape).to(ws.dev).contiguous()                  # Imagen.py:485 randn_like
        else:
            noise = torch.empty_like(up)
            L.check(lib.mi_randn_fill(L.ptr(noise), B, n, seed, sample0, (stage << 20) | (1 << 19), stream), "mi_randn_fill")
        a = self.lowres_noise_schedule._host_sqrt_alphas_cumprod[t_low]
        b = self.lowres_noise_schedule._host_sqrt_one_minus_alphas_cumprod[t_low]
        L.check(lib.mi_lowres_augment(L.ptr(up), L.ptr(noise), L.ptr(ws.lowres), B * n, a, b, 1 if self.auto_normalize_img else 0, stream), "mi_lowres_augment")
        ws.lowres_keepalive = (up, noise)
        unet = [u for u in self.unets if u.engine()._ws and ws in u.engine()._ws.values()][0]
        unet.engine().prepare_lowres(ws, stream)

    @torch.no_grad()
    @eval_decorator
    def sample(self, texts: List[str] = None, text_masks: torch.Tensor = None, text_embeds: torch.Tensor = None,
               cond_scale: float = 1., lowres_sample_noise_level: float = None, return_pil_images: bool = False,
               device: torch.device = None, *, _noise: Callable = None, _seed: int = 1234, _sample_offset: int = 0,
               _use_graph: bool = True, _precision: str = None, _async: bool = False, _revalidated: bool = False):
        """minimagen/Imagen.py:424-510.  Private keyword-only extras (not in the reference): ``_noise(shape)`` injects a
        host noise stream in the reference's draw order (parity runs); otherwise noise is Philox keyed by
        (``_seed``, ``_sample_offset`` + row, stage, step, element) so a sharded batch reproduces the unsharded one;
        ``_precision`` = "fp32" (default) or "half" (single-fp16-term matrix-core contractions, see engine.UnetEngine.precision);
        ``_async=True`` returns without making the caller's stream wait (``self.last_sample_done`` / the returned tensor's ``sample_done`` is THIS call's completion event; ``wait_pending_samples()`` covers every lane): successive
        calls then pipeline across the per-stage streams (the base stage of the next batch under the super-resolution stage of this one)."""
        call_args = dict(texts=texts, text_masks=text_masks, text_embeds=text_embeds, cond_scale=cond_scale, lowres_sample_noise_level=lowres_sample_noise_level,
                         return_pil_images=return_pil_images, device=device, _noise=_noise, _seed=_seed, _sample_offset=_sample_offset,
                         _use_graph=_use_graph, _precision=_precision, _async=_async)
        device = default(device, self.device)
        self._poll_status()                  # a cooperative launch of an EARLIER call gave up (its images are NaN): raise here, never silently
        self._status_stages = []
        self._reset_unets_all_one_device(device=device)
        if exists(texts) and not exists(text_embeds):
            text_embeds, text_masks = t5_encode_text(texts, name=self.text_encoder_name)
            text_embeds, text_masks = map(lambda t: t.to(device), (text_embeds, text_masks))
        assert exists(text_embeds), 'text or text encodings must be passed into Imagen'
        assert not (exists(text_embeds) and text_embeds.shape[-1] != self.text_embed_dim), \
            f'invalid text embedding dimension being passed in (should be {self.text_embed_dim})'
        assert not (cond_scale != 1. and not self.can_classifier_guidance), \
            'imagen was not trained with conditional dropout, and thus one cannot use classifier free guidance (cond_scale anything other than 1)'
        batch_size = text_embeds.shape[0]
        device = next(self.parameters()).device
        lowres_sample_noise_level = default(lowres_sample_noise_level, self.lowres_sample_noise_level)
        two = cond_scale != 1.
        B2 = 2 * batch_size if two else batch_size
        keep = torch.cat((torch.ones(batch_size, dtype=torch.bool), torch.zeros(B2 - batch_size, dtype=torch.bool)))
        text_embeds = text_embeds.to(device)
        text_masks = text_masks.to(device) if exists(text_masks) else None

        # One HIP stream PER STAGE (graphs cannot be captured on the legacy default stream anyway).  Within a call stage s + 1 waits for
        # stage s through an event; ACROSS calls the stages form a pipeline: with ``_async=True`` the caller's stream is never made to
        # wait, so the (small, latency-bound) base stage of call k + 1 runs on its stream while the super-resolution stage of call k
        # still occupies the other -- each stage owns its workspace, so nothing is shared but the finished image handed down the cascade.
        on_gpu = L.backend() == "hip-gfx950"
        # Packed-weight validation, once per call.  Identity (pointers, version counters: host only) up front; the content fingerprint -- device
        # work on the caller's stream + one small device -> host copy -- is launched and read AFTER this call's work is enqueued (see
        # engine.pack_identity).  A rare positive verdict discards the enqueued work and runs the call again on fresh packs.  Injected noise
        # (a stateful host generator: the call cannot be repeated) keeps the whole check up front.
        for unet in self.unets:
            if _noise is not None or _revalidated:
                unet.engine().pack()
            else:
                unet.engine().pack_identity()
        # LANES: asynchronous calls alternate between SAMPLE_LANES independent sets of (stage streams, workspaces, graphs), so that two
        # calls are in flight side by side -- the kernels of one call's stages fill the launch floors and tails of the other's (measured:
        # 42.8 K vs 37.7 K steps/s for the B = 32 cascade).  Calls on one lane stay ordered by its streams; lanes share
        # only read-only state (weights, tables).  Synchronous calls use lane 0.
        lane = 0
        if _async and on_gpu and SAMPLE_LANES > 1:
            lane = self._lane_rr = (getattr(self, "_lane_rr", -1) + 1) % SAMPLE_LANES
        if on_gpu:
            all_streams = getattr(self, "_stage_streams", None)
            if all_streams is None or len(all_streams[0]) != len(self.unets) or all_streams[0][0].device != device or len(all_streams) != max(1, SAMPLE_LANES):
                # earlier (smaller-image, latency-bound) stages get the higher stream priority: their short kernels then slot in between the
                # workgroups of the later stages' large ones instead of queueing behind them
                prio = int(os.environ.get("MINIMAGEN_STAGE_PRIORITY", "1"))
                if all_streams is not None:
                    torch.cuda.synchronize(device)       # (rare: lane count / device changed) work queued on the old lanes' streams still owns the workspaces
                # ONE set of stage streams per process and shape, shared by every Imagen instance.  HIP multiplexes its streams onto
                # GPU_MAX_HW_QUEUES (default 4) hardware queues, and torch hands out pool streams round robin: a second Imagen built in the
                # same process got streams whose hardware queue was shared with the stream feeding them, and every node of a graph launched
                # there paid ~16 us (2.7 ms per 170-node launch: a 150 ms sample() took 261 ms; tools/gpu_realloc_slowdown.py,
                # profiles/r04_second_instance_slowdown.txt -- the device buffers had nothing to do with it).  The first set a process creates
                # has never shown the sharing; keeping it for all instances makes the mapping a constant.
                key = (str(device), max(1, SAMPLE_LANES), len(self.unets), prio)
                all_streams = _STAGE_STREAMS.get(key)
                if all_streams is None:
                    all_streams = _STAGE_STREAMS[key] = [[torch.cuda.Stream(device=device, priority=(-1 if (prio and k + 1 < len(self.unets)) else 0))
                                                          for k in range(len(self.unets))] for _ in range(max(1, SAMPLE_LANES))]
                self._stage_streams = all_streams
            streams = all_streams[lane]
            self._stream = all_streams[0][-1]            # (benchmarks time the last stage's captured graph on its own stream)
            caller_stream = torch.cuda.current_stream(device)
            inputs_ready = caller_stream.record_event()
        from .helpers import null_context
        precision = _precision if _precision is not None else os.environ.get("MINIMAGEN_PRECISION", "fp32")
        stages = list(enumerate(zip(self.unets, self.sample_channels, self.image_sizes, self.noise_schedulers)))
        # ---- pass 1, every stage on its own stream: what depends on the CAPTIONS only (text conditioning, the folded context rows, x_T, the
        # step tables) -- issued for all stages up front, so a later stage has it behind it when its low-resolution input arrives
        wss, begun = {}, {}
        for stage, (unet, channel, image_size, noise_scheduler) in stages:
            if on_gpu:
                streams[stage].wait_event(inputs_ready)
                # the stage streams read the caller's tensors after sample() has returned (_async) / after the caller may have dropped
                # them: tell the caching allocator, or a block freed on the caller's stream could be handed out again while a stage's
                # text_cond launch is still queued
                for t_in in (text_embeds, text_masks):
                    if t_in is not None and t_in.is_cuda:
                        t_in.record_stream(streams[stage])
            with (torch.cuda.stream(streams[stage]) if on_gpu else null_context()):
                eng = unet.engine()
                # per call, never sticky engine state: a later Unet.forward stays on the engine's default precision
                ws = wss[stage] = eng.workspace(batch_size, B2, image_size, image_size, precision=precision, lane=lane,
                                                pipelined=bool(_async and on_gpu and SAMPLE_LANES > 1))
                eng.set_text(ws, text_embeds, text_masks, keep)
                if unet.lowres_cond:             # the augmentation level's timestep feeds the step tables (diffusion_model.py:68-69)
                    ws.lowres_times.fill_(int(self.lowres_noise_schedule.num_timesteps * lowres_sample_noise_level))
                if _noise is None:
                    begun[stage] = self._stage_begin(unet, (batch_size, self.channels, image_size, image_size), noise_scheduler=noise_scheduler,
                                                     ws=ws, seed=_seed, sample0=_sample_offset, stage=stage)
        # ---- pass 2: the cascade
        img, prev_done = None, None
        for stage, (unet, channel, image_size, noise_scheduler) in stages:
            if on_gpu and prev_done is not None:
                streams[stage].wait_event(prev_done)
            with (torch.cuda.stream(streams[stage]) if on_gpu else null_context()):
                ws = wss[stage]
                if unet.lowres_cond:
                    if on_gpu:
                        img.record_stream(streams[stage])
                    self._lowres_conditioning(img, image_size, ws, lowres_sample_noise_level, _noise, _seed, _sample_offset, stage)
                img = self._p_sample_loop(unet, (batch_size, self.channels, image_size, image_size), noise_scheduler=noise_scheduler,
                                          ws=ws, cond_scale=cond_scale, noise_fn=_noise, seed=_seed, sample0=_sample_offset,
                                          stage=stage, use_graph=_use_graph, begun=begun.get(stage))
                if on_gpu:
                    prev_done = streams[stage].record_event()
        pack_tokens = [] if (_noise is not None or _revalidated) else [(unet.engine(), unet.engine().pack_begin()) for unet in self.unets]
        if any(eng.pack_changed(tok) for eng, tok in pack_tokens):
            # the weights' values had changed behind the version counters (p.data updates): what was enqueued ran on stale packs
            if on_gpu:
                torch.cuda.synchronize(device)
            self._status_stages = []
            call_args["_revalidated"] = True
            return self.sample(**call_args)
        if self._status_stages:
            self.__dict__.setdefault("_status_pending", []).append((prev_done, self._status_stages))
            self._status_stages = []
        if STRICT_STATUS and on_gpu and not _async:
            prev_done.synchronize()
            self._poll_status()
        if on_gpu:
            if _async:
                # pipelined use: the result is ready when ``done`` is (the caller synchronises / waits on it before touching the images)
                # per CALL: the event of this very call.  A later call on the OTHER lane does not wait for it -- capture it right after
                # the call that produced the tensor (it also travels on the tensor), or use wait_pending_samples() to cover every lane
                self.last_sample_done = prev_done
                self.__dict__.setdefault("_lane_done", {})[lane] = prev_done
                if not return_pil_images:
                    img.sample_done = prev_done
                    return img
                caller_stream.wait_event(prev_done)          # the device -> host copy below runs on the caller's stream
                img.record_stream(caller_stream)
                pil = _to_pil_images(img)
                self.check_device_status()
                return pil
            caller_stream.wait_event(prev_done)
            img.record_stream(caller_stream)
        if not return_pil_images:
            return img
        pil = _to_pil_images(img)
        self.check_device_status()
        return pil


def _poll_status(self, block: bool = False):
    """Status of the kernels whose workgroups wait for each other (the grouped sampler tail).  Such a launch is fail-stop: when a wait
    runs out it sets a sticky error word and turns the image -- and everything sampled from it afterwards -- into NaN.  Every sample()
    call copies the word to pinned host memory behind its last launch; this looks at the calls whose completion event has fired
    (``block=True``: waits for all of them) and raises MinImagenHipError for a failed one.  Called at every sample() entry, by
    wait_pending_samples() and check_device_status().  Recovery is automatic: the stage state re-zeroes its sync buffer before its next
    launch and keeps the separate (non-cooperative) kernels from then on."""
    pending = self.__dict__.get("_status_pending", [])
    failed, rest = [], []
    for done, stages in pending:
        if done is not None:
            if block:
                done.synchronize()
            elif not done.query():
                rest.append((done, stages))
                continue
        for st, stage, shape in stages:
            err = int(st.group_err_host.item())
            if err and not getattr(st, "group_failed", False):
                st.group_failed, st.group_heal = True, True
                failed.append(f"stage {stage} {shape}: {err:#x}")
            elif err:
                failed.append(f"stage {stage} {shape}: {err:#x} (call queued behind the failed one)")
    self._status_pending = rest
    if failed:
        raise L.MinImagenHipError("grouped sampler tail: a workgroup timed out waiting for its image's other workgroups (" + "; ".join(failed) +
                                  "); the images of that sample() call are NaN.  The stage falls back to the separate kernels from the next call on.")


Imagen._poll_status = _poll_status


def _check_device_status(self):
    """Host-side check (waits for every sample() call in flight): no kernel whose workgroups wait for each other gave up waiting."""
    self._poll_status(block=True)


Imagen.check_device_status = _check_device_status


def _wait_pending_samples(self, stream=None):
    """Make ``stream`` (default: the caller's current stream) wait for the latest ``sample(_async=True)`` call of EVERY call lane; raises
    if a call that has already completed reported a failed cooperative launch (check_device_status() waits on the host and covers all)."""
    for ev in self.__dict__.get("_lane_done", {}).values():
        (stream if stream is not None else torch.cuda.current_stream()).wait_event(ev)
    self._poll_status()


Imagen.wait_pending_samples = _wait_pending_samples


def _to_pil_images(img: torch.Tensor):
    """torchvision.transforms.ToPILImage on a float CHW tensor (Imagen.py:508): mul(255) then truncate to uint8."""
    import numpy as np
    from PIL import Image
    arr = img.detach().to('cpu').mul(255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()
    return [Image.fromarray(np.ascontiguousarray(a)) for a in arr]
