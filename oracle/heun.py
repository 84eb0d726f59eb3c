"""numpy/torch restatement of the reference's batched-timestep Heun scheduler, the CFG
teacher query combine, the SNR-weighted loss and the EMA update (test oracle).

Reference: diffusers/schedulers/scheduling_heun_discrete.py (set_timesteps :174-227,
scale_model_input :151-172, step :273-362, add_noise :364-385),
models/audio_distilled_model.py:286-322 (_query_teacher),
models/audio_consistency_model.py:215-219,250-266 (compute_snr / get_loss),
tools/losses.py:28-33 (MSELoss 'instance'), tools/train_utils.py:255-282 (do_ema_update).
The scheduler config is Stable-Diffusion-2.1's (train.sh:5, fetched from the HF hub by the
reference): 1000 train steps, scaled_linear betas 0.00085..0.012, v_prediction.
"""
import numpy as np
import torch


def sigma_table(num_train=1000, beta_start=0.00085, beta_end=0.012):
    """float64 array of the 1000 training sigmas (computed like the reference: betas and
    alphas_cumprod in float32 torch, then numpy)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train,
                           dtype=torch.float32) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    return np.array(((1 - ac) / ac) ** 0.5)


def set_timesteps(n, num_train=1000):
    """Returns (timesteps float64 [2n-1], sigmas float32 [2n]) as set_timesteps :174-227."""
    ts = np.linspace(0, num_train - 1, n, dtype=float)[::-1].copy()
    s = sigma_table(num_train)
    sig = np.interp(ts, np.arange(0, len(s)), s)
    sig = np.concatenate([sig, [0.0]]).astype(np.float32)
    sig = np.concatenate([sig[:1], np.repeat(sig[1:-1], 2), sig[-1:]])
    ts = np.concatenate([ts[:1], np.repeat(ts[1:], 2)])
    return ts, sig


def scale_model_input(x, sigma):
    s = sigma.reshape(-1, 1, 1, 1)
    return x / ((s ** 2 + 1) ** 0.5)


def add_noise(x0, noise, sigma):
    return x0 + noise * sigma.reshape(-1, 1, 1, 1)


def pred_x0_v(sample, v, sigma_in):
    s = sigma_in.reshape(-1, 1, 1, 1)
    alpha_prod = 1 / (s ** 2 + 1)
    return sample * alpha_prod - v * (s * alpha_prod ** 0.5)


def step_first(v, sample, sigma, sigma_next):
    """1st-order half of `step` (state_in_first_order).  Returns (prev_sample, derivative, dt)."""
    x0 = pred_x0_v(sample, v, sigma)
    d = (sample - x0) / sigma.reshape(-1, 1, 1, 1)
    dt = (sigma_next - sigma).reshape(-1, 1, 1, 1)
    return sample + d * dt, d, dt


def step_second(v, sample_hat, sigma_next, stored_sample, prev_d, dt):
    """2nd-order half: derivative at the predicted point, averaged, applied to the STORED sample."""
    x0 = pred_x0_v(sample_hat, v, sigma_next)
    d = (sample_hat - x0) / sigma_next.reshape(-1, 1, 1, 1)
    d = (prev_d + d) / 2
    return stored_sample + d * dt


def cfg_combine(v_uncond, v_cond, w):
    """_query_teacher :313-319: (1 - w) * uncond + w * cond with per-sample w."""
    w = w.reshape(-1, 1, 1, 1)
    return (1 - w) * v_uncond + w * v_cond


def snr_mse_loss(pred, target, sigma, snr_gamma=5.0):
    inst = ((pred.float() - target.float()) ** 2).mean(dim=tuple(range(1, pred.ndim)))
    wgt = torch.clamp(sigma.float() ** (-2), max=snr_gamma)
    return (inst * wgt).mean()


def ema_update(shadow, param, decay):
    """do_ema_update :277-282: shadow -= (1-decay)*(shadow-param), in place semantics."""
    return shadow - (1.0 - decay) * (shadow - param)
