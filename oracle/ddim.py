"""CPU restatement of the DDPM / DDIM scheduler arithmetic used by stage-1 guided distillation (TEST INFRASTRUCTURE
ONLY): diffusers/schedulers/scheduling_ddpm.py:132-183,420-443 (tables, add_noise), scheduling_ddim.py:218-241,243-370
(set_timesteps, step with eta = 0) in the reference's batched-timestep versions, and the min-SNR loss weights of
models/audio_guided_model.py:92-117.  SD-2.1 scheduler config: scaled_linear betas, v_prediction, clip_sample false,
set_alpha_to_one false."""
import numpy as np
import torch


def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def ddpm_timesteps(num_train_timesteps=1000):
    return torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())


def ddim_timesteps(num_inference_steps, num_train_timesteps=1000):
    ratio = num_train_timesteps // num_inference_steps
    return torch.from_numpy((np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64))


def add_noise(x0, noise, t, ac=None):
    ac = alphas_cumprod() if ac is None else ac
    a = (ac[t] ** 0.5).reshape(-1, 1, 1, 1)
    b = ((1 - ac[t]) ** 0.5).reshape(-1, 1, 1, 1)
    return a * x0 + b * noise


def ddim_step(v, t, sample, num_inference_steps, ac=None, num_train_timesteps=1000, clip=None):
    """eta = 0, v-prediction; t (B,) int64; final_alpha_cumprod = alphas_cumprod[0] (set_alpha_to_one false)."""
    ac = alphas_cumprod() if ac is None else ac
    t = torch.as_tensor(t).reshape(-1)
    prev = t - num_train_timesteps // num_inference_steps
    a_t = ac[t]
    a_prev = torch.where(prev >= 0, ac[prev.clamp(min=0)], ac[0]).reshape(-1, 1, 1, 1)
    b_t = 1 - a_t
    sa, sb = (a_t ** 0.5).reshape(-1, 1, 1, 1), (b_t ** 0.5).reshape(-1, 1, 1, 1)
    x0 = sa * sample - sb * v
    eps = sa * v + sb * sample
    if clip:
        x0 = x0.clamp(-clip, clip)
    return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps


def gdm_loss_weights(t, snr_gamma, ac=None):
    """compute_snr (audio_distilled_model.py) = alpha^2 / sigma^2 = ac / (1 - ac); v-prediction weights
    min(snr, gamma) / (snr + 1)."""
    ac = alphas_cumprod() if ac is None else ac
    snr = ((ac[t] ** 0.5) / ((1.0 - ac[t]) ** 0.5)) ** 2          # audio_distilled_model.py:165-191
    return torch.clamp(snr, max=snr_gamma) / (snr + 1)
