"""Noise schedule with the reference's class name and buffers (minimagen/diffusion_model.py)."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn


class GaussianDiffusion(nn.Module):
    """Linear-beta DDPM schedule.  The twelve tables are computed in fp64 and stored as fp32
    non-persistent buffers exactly as diffusion_model.py:28-66 does, so every look-up value is
    bit-identical to the reference's.  On the sampling path the per-timestep arithmetic that consumes them
    (predict_start_from_noise / q_posterior / q_sample) runs inside the HIP sampler kernels; the methods of the
    same names below are the tensor forms of the class API."""

    def __init__(self, *, timesteps: int):
        super().__init__()
        assert not timesteps < 20, f'timsteps must be at least 20'
        self.num_timesteps = timesteps
        scale = 1000 / timesteps
        betas = torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)
        alphas = 1. - betas
        alphas_cumprod = torch.cumprod(alphas, dim=0)
        alphas_cumprod_prev = F.pad(alphas_cumprod[:-1], (1, 0), value=1.)
        reg = lambda name, val: self.register_buffer(name, val.to(torch.float32), persistent=False)
        reg('betas', betas)
        reg('alphas_cumprod', alphas_cumprod)
        reg('alphas_cumprod_prev', alphas_cumprod_prev)
        reg('sqrt_alphas_cumprod', torch.sqrt(alphas_cumprod))
        reg('sqrt_one_minus_alphas_cumprod', torch.sqrt(1. - alphas_cumprod))
        reg('log_one_minus_alphas_cumprod', torch.log(1. - alphas_cumprod))
        reg('sqrt_recip_alphas_cumprod', torch.sqrt(1. / alphas_cumprod))
        reg('sqrt_recipm1_alphas_cumprod', torch.sqrt(1. / alphas_cumprod - 1))
        posterior_variance = betas * (1. - alphas_cumprod_prev) / (1. - alphas_cumprod)
        reg('posterior_variance', posterior_variance)
        reg('posterior_log_variance_clipped', torch.log(posterior_variance.clamp(min=1e-20)))
        reg('posterior_mean_coef1', betas * torch.sqrt(alphas_cumprod_prev) / (1. - alphas_cumprod))
        reg('posterior_mean_coef2', (1. - alphas_cumprod_prev) * torch.sqrt(alphas) / (1. - alphas_cumprod))
        # host copies of the two q_sample tables (same fp32 values): the sampler reads one scalar of each per stage, and reading it from
        # the device buffer would synchronise the host with that stage's stream -- i.e. with the previous call still running on it
        self._host_sqrt_alphas_cumprod = self.sqrt_alphas_cumprod.tolist()
        self._host_sqrt_one_minus_alphas_cumprod = self.sqrt_one_minus_alphas_cumprod.tolist()

    def _get_times(self, batch_size: int, noise_level: float, *, device) -> torch.Tensor:
        """diffusion_model.py:68-69"""
        return torch.full((batch_size,), int(self.num_timesteps * noise_level), device=device, dtype=torch.long)

    def _sample_random_times(self, batch_size: int, *, device) -> torch.Tensor:
        return torch.randint(0, self.num_timesteps, (batch_size,), device=device, dtype=torch.long)

    def _get_sampling_timesteps(self, batch: int, *, device):
        """diffusion_model.py:81-87"""
        return [torch.full((batch,), i, device=device, dtype=torch.long) for i in reversed(range(self.num_timesteps))]

    def sampler_coef_table(self) -> torch.Tensor:
        """[T][8] table consumed by mi_cfg_x0_fwd / mi_posterior_fwd: per-timestep scalars gathered from the
        buffers above; column 4 is [t != 0] * exp(0.5 * posterior_log_variance_clipped) (Imagen.py:364-370)."""
        T = self.num_timesteps
        tab = torch.zeros(T, 8, dtype=torch.float32)
        cpu = lambda v: v.detach().to('cpu', torch.float32)
        tab[:, 0] = cpu(self.sqrt_recip_alphas_cumprod)
        tab[:, 1] = cpu(self.sqrt_recipm1_alphas_cumprod)
        tab[:, 2] = cpu(self.posterior_mean_coef1)
        tab[:, 3] = cpu(self.posterior_mean_coef2)
        nonzero = torch.ones(T)
        nonzero[0] = 0.
        tab[:, 4] = nonzero * (0.5 * cpu(self.posterior_log_variance_clipped)).exp()
        return tab

    # ---- the per-timestep helpers of the reference's public API (diffusion_model.py:89-162).  The sampling hot path has them fused
    # into the HIP sampler kernels (mi_lowres_augment, mi_cfg_x0_fwd, mi_posterior_fwd); these tensor forms serve callers of the class
    # API and the training loss (Imagen.forward), on whatever device the tables live, and are differentiable.
    @staticmethod
    def _at(table: torch.Tensor, t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
        """table[t] broadcast over an image batch: (b,) -> (b, 1, 1, 1) (helpers.extract, helpers.py:48-59)"""
        return table.gather(-1, t).reshape(t.shape[0], *((1,) * (like.dim() - 1)))

    def q_sample(self, x_start: torch.Tensor, t: torch.Tensor, noise: torch.Tensor = None) -> torch.Tensor:
        """x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) eps (diffusion_model.py:127-147)"""
        if noise is None:
            noise = torch.randn_like(x_start)
        return self._at(self.sqrt_alphas_cumprod, t, x_start) * x_start + self._at(self.sqrt_one_minus_alphas_cumprod, t, x_start) * noise

    def q_posterior(self, x_start: torch.Tensor, x_t: torch.Tensor, t: torch.Tensor):
        """mean, variance and clipped log-variance of q(x_{t-1} | x_t, x_0) (diffusion_model.py:89-125)"""
        mean = self._at(self.posterior_mean_coef1, t, x_t) * x_start + self._at(self.posterior_mean_coef2, t, x_t) * x_t
        return mean, self._at(self.posterior_variance, t, x_t), self._at(self.posterior_log_variance_clipped, t, x_t)

    def predict_start_from_noise(self, x_t: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        """x_0 = sqrt(1 / abar_t) x_t - sqrt(1 / abar_t - 1) eps (diffusion_model.py:149-162)"""
        return self._at(self.sqrt_recip_alphas_cumprod, t, x_t) * x_t - self._at(self.sqrt_recipm1_alphas_cumprod, t, x_t) * noise
