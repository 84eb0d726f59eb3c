"""HeunDiscreteScheduler with per-sample (batched) timesteps, backed by the fused HIP
elementwise kernels.  Mirrors diffusers/schedulers/scheduling_heun_discrete.py of the
reference (which rewrote the stock scheduler for batched timesteps): same constructor
arguments, `set_timesteps`, `timesteps` / `sigmas` / `init_noise_sigma`,
`scale_model_input`, `add_noise`, `step(...).prev_sample`, `state_in_first_order` and the
resettable `prev_derivative` / `dt` / `sample` state (audio_consistency_model.py:392-395).

Difference by design: the sigma/timestep tables also live on the HOST, so looking a Python /
CPU timestep up costs no device->host sync (the reference's `index_for_timestep` does
`mask.cpu()` on every call, :141-149).  CUDA-tensor timesteps still work (one sync, as in the
reference).
"""
from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np
import torch

from . import _native as N

# Stable-Diffusion-2.1 scheduler config, which the reference fetches from the HF hub
# (scheduler_name="stabilityai/stable-diffusion-2-1", train.sh:5)
SD21_SCHEDULER_CONFIG = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                             beta_schedule="scaled_linear", prediction_type="v_prediction")


@dataclass
class SchedulerOutput:
    prev_sample: torch.Tensor


class HeunDiscreteScheduler:
    order = 2

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                 trained_betas=None, prediction_type="epsilon", use_karras_sigmas=False):
        if trained_betas is not None:
            self.betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} does is not implemented for {self.__class__}")
        if prediction_type != "v_prediction":
            # the HIP step kernels implement the path the reference actually runs (SD-2.1 config)
            raise NotImplementedError("only prediction_type='v_prediction' is built (SD-2.1 scheduler config)")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, beta_schedule=beta_schedule,
                                      prediction_type=prediction_type, use_karras_sigmas=use_karras_sigmas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.use_karras_sigmas = use_karras_sigmas
        self.set_timesteps(num_train_timesteps, None, num_train_timesteps)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, subfolder=None, **kwargs):
        """No network on the build/GPU boxes: any name resolves to the SD-2.1 scheduler config
        the reference's scripts use."""
        cfg = dict(SD21_SCHEDULER_CONFIG)
        cfg.update(kwargs)
        return cls(**cfg)

    # ---- tables (set_timesteps :174-227)
    def set_timesteps(self, num_inference_steps, device=None, num_train_timesteps=None):
        self.num_inference_steps = num_inference_steps
        num_train_timesteps = num_train_timesteps or self.config.num_train_timesteps
        timesteps = np.linspace(0, num_train_timesteps - 1, num_inference_steps, dtype=float)[::-1].copy()
        sigmas = np.array((((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).tolist(), dtype=np.float32)
        sigmas = np.interp(timesteps, np.arange(0, len(sigmas)), sigmas)
        if self.use_karras_sigmas:
            raise NotImplementedError("use_karras_sigmas is not built (unused by train.sh / inference.sh)")
        sigmas = np.concatenate([sigmas, [0.0]]).astype(np.float32)
        sig = np.concatenate([sigmas[:1], np.repeat(sigmas[1:-1], 2), sigmas[-1:]])
        ts = np.concatenate([timesteps[:1], np.repeat(timesteps[1:], 2)])
        self._sigmas_host = sig
        self._timesteps_host = ts
        if device is not None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            device = None   # inside hipGraph capture: no host->device copies; the kernels use the host tables anyway
        self.sigmas = torch.from_numpy(sig).to(device=device)
        self.init_noise_sigma = self.sigmas.max()
        self.timesteps = torch.from_numpy(ts).to(device=device)
        self.prev_derivative = None
        self.dt = None
        self.sample = None

    @property
    def state_in_first_order(self):
        return self.dt is None

    def index_for_timestep(self, timestep):
        """LAST index with timesteps == t (argmax of mask*arange, :145-149) in first-order state,
        that index - 1 in second-order state."""
        if torch.is_tensor(timestep):
            t = timestep.detach().reshape(-1).cpu().numpy().astype(np.float64)
        else:
            t = np.asarray(timestep, dtype=np.float64).reshape(-1)
        eq = self._timesteps_host[None, :] == t[:, None]
        assert eq.any(axis=1).all(), f"timestep: {t.tolist()}"
        idx = (eq * np.arange(eq.shape[1])[None, :]).argmax(axis=1)
        return idx if self.state_in_first_order else idx - 1

    def _sigma_dev(self, idx, B, device):
        s = np.asarray(self._sigmas_host[idx], dtype=np.float32).reshape(-1)
        if s.shape[0] == 1 or bool((s == s[0]).all()):
            # one sigma for the whole batch: a device-side fill (no host->device copy, safe inside hipGraph capture)
            return torch.full((B,), float(s[0]), dtype=torch.float32, device=device)
        return torch.from_numpy(np.ascontiguousarray(s)).to(device, non_blocking=True)

    @staticmethod
    def _check(x):
        if not x.is_cuda:
            raise N.CttaError("tensor on %s: the HIP scheduler kernels have no CPU path" % x.device)
        return x.detach().to(torch.float32).contiguous()

    def scale_model_input(self, sample, timestep):
        x = self._check(sample)
        B = x.shape[0]
        sig = self._sigma_dev(self.index_for_timestep(timestep), B, x.device)
        out = torch.empty_like(x)
        N.check(N.lib().ctta_heun_scale_model_input(N.ptr(x), N.ptr(sig), N.ptr(out), B, x[0].numel(), N.stream_ptr()))
        return out

    def add_noise(self, original_samples, noise, timesteps):
        x = self._check(original_samples)
        nz = self._check(noise)
        B = x.shape[0]
        sig = self._sigma_dev(self.index_for_timestep(timesteps), B, x.device)
        out = torch.empty_like(x)
        N.check(N.lib().ctta_heun_add_noise(N.ptr(x), N.ptr(nz), N.ptr(sig), N.ptr(out), B, x[0].numel(),
                                            N.stream_ptr()))
        return out

    def step(self, model_output, timestep, sample, return_dict=True):
        v = self._check(model_output)
        x = self._check(sample)
        B = x.shape[0]
        idx = self.index_for_timestep(timestep)
        n = x[0].numel()
        prev = torch.empty_like(x)
        L_ = N.lib()
        if self.state_in_first_order:
            sig = self._sigma_dev(idx, B, x.device)
            sig_next = self._sigma_dev(idx + 1, B, x.device)
            deriv = torch.empty_like(x)
            N.check(L_.ctta_heun_step_first(N.ptr(v), N.ptr(x), N.ptr(sig), N.ptr(sig_next), N.ptr(prev),
                                            N.ptr(deriv), B, n, N.stream_ptr()))
            self.prev_derivative = deriv
            self.dt = (sig, sig_next)      # kept as the two sigmas; dt = sigma_next - sigma
            self.sample = x
        else:
            # :309-311: sigma = sigmas[idx-1], sigma_next = sigmas[idx] of THIS call; the stored dt of
            # the first half equals sigma_next - sigma in every call pattern of the reference
            sig = self._sigma_dev(idx - 1, B, x.device)
            sig_next = self._sigma_dev(idx, B, x.device)
            N.check(L_.ctta_heun_step_second(N.ptr(v), N.ptr(x), N.ptr(self.sample), N.ptr(self.prev_derivative),
                                             N.ptr(sig), N.ptr(sig_next), N.ptr(prev), B, n, N.stream_ptr()))
            self.prev_derivative = None
            self.dt = None
            self.sample = None
        return SchedulerOutput(prev_sample=prev) if return_dict else (prev,)

    def __len__(self):
        return self.config.num_train_timesteps


# ------------------------------------------------------------------------------------------------------------------
# Stage-1 guided distillation: the reference's batched-timestep DDPM / DDIM schedulers
# (diffusers/schedulers/scheduling_ddpm.py, scheduling_ddim.py).  Coefficient tables live on the host; the per-sample
# linear combinations run on the HIP elementwise kernel ctta_lincomb2_rows.
SD21_DDIM_EXTRA = dict(clip_sample=False, set_alpha_to_one=False)


def _scaled_linear_alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas):
    if trained_betas is not None:
        betas = torch.tensor(trained_betas, dtype=torch.float32)
    elif beta_schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    elif beta_schedule == "scaled_linear":
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    else:
        raise NotImplementedError(f"{beta_schedule} does is not implemented for this scheduler")
    return betas, torch.cumprod(1.0 - betas, dim=0)


class _LinCombMixin:
    @staticmethod
    def _lincomb(x, y, a, b, clamp=0.0):
        """out = a[b]*x + b[b]*y with per-sample coefficients given as host tensors / arrays."""
        if not x.is_cuda:
            raise N.CttaError("scheduler inputs are on %s: the HIP kernels have no CPU path" % x.device)
        x = x.detach().to(torch.float32).contiguous()
        y = y.detach().to(device=x.device, dtype=torch.float32).contiguous()
        B = x.shape[0]
        ad = torch.as_tensor(np.asarray(a, dtype=np.float32)).reshape(-1).expand(B).contiguous().to(x.device)
        bd = torch.as_tensor(np.asarray(b, dtype=np.float32)).reshape(-1).expand(B).contiguous().to(x.device)
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            N.check(N.lib().ctta_lincomb2_rows(N.ptr(x), N.ptr(y), N.ptr(ad), N.ptr(bd), N.ptr(out), B, x[0].numel(),
                                               float(clamp), N.stream_ptr()))
        return out

    def _t_index(self, timesteps, B):
        """(B,) int64 host indices from a Python int / CPU tensor (no device sync) or a CUDA tensor (one sync)."""
        t = torch.as_tensor(timesteps).detach().reshape(-1).to("cpu", torch.int64)
        return t.expand(B).contiguous() if t.numel() == 1 else t

    def scale_model_input(self, sample, timestep=None):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        """sqrt(ac[t]) * x0 + sqrt(1 - ac[t]) * noise  (scheduling_ddpm.py:420-443, scheduling_ddim.py:372-393)."""
        t = self._t_index(timesteps, original_samples.shape[0])
        ac = self.alphas_cumprod[t]
        return self._lincomb(original_samples, noise, (ac ** 0.5).numpy(), ((1 - ac) ** 0.5).numpy())

    def __len__(self):
        return self.config.num_train_timesteps


class DDPMScheduler(_LinCombMixin):
    """Training-side noising schedule of AudioGDM (models/audio_guided_model.py:43-47): timesteps 999..0, add_noise,
    init_noise_sigma = 1, identity scale_model_input, alphas_cumprod for compute_snr, and the ancestral `step`
    (scheduling_ddpm.py:285-418; unused by the reference's scripts -- inference runs DDIM or Heun -- kept for parity)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon",
                 thresholding=False, clip_sample_range=1.0, **kw):
        if thresholding:
            raise NotImplementedError("dynamic thresholding is unused by ConsistencyTTA and not built")
        self.betas, self.alphas_cumprod = _scaled_linear_alphas_cumprod(num_train_timesteps, beta_start, beta_end,
                                                                       beta_schedule, trained_betas)
        self.alphas = 1.0 - self.betas
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, prediction_type=prediction_type,
                                      clip_sample=clip_sample, clip_sample_range=clip_sample_range,
                                      variance_type=variance_type, thresholding=False)
        self.variance_type = variance_type
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, subfolder=None, **kwargs):
        return cls(**dict(SD21_SCHEDULER_CONFIG, clip_sample=False, **kwargs))

    def set_timesteps(self, num_inference_steps, device=None):
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                             f"`self.config.train_timesteps`: {self.config.num_train_timesteps}")
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        self.timesteps = torch.from_numpy((np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64))

    def _coeffs(self, t):
        """Host-side per-sample tables of `step` (scheduling_ddpm.py:319-333,232-265): x0 / x_t coefficients of the
        posterior mean and the posterior variance, float32 like the reference's tensor arithmetic."""
        n_inf = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        prev = t - self.config.num_train_timesteps // n_inf
        one = torch.tensor(1.0)
        a_t = self.alphas_cumprod[t]
        a_prev = torch.where(prev >= 0, self.alphas_cumprod[prev.clamp(min=0)], one)
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        c_x0 = (a_prev ** 0.5 * cur_b) / b_t
        c_xt = cur_a ** 0.5 * b_prev / b_t
        var = b_prev / b_t * cur_b
        vt = self.config.variance_type
        if vt == "fixed_small":
            std = torch.clamp(var, min=1e-20) ** 0.5
        elif vt == "fixed_small_log":
            std = torch.exp(0.5 * torch.log(torch.clamp(var, min=1e-20)))
        elif vt == "fixed_large":
            std = cur_b ** 0.5
        else:
            raise NotImplementedError("variance_type %r: only the fixed variances are built (the ConsistencyTTA U-Nets "
                                      "predict no variance)" % (vt,))
        return a_t, b_t, c_x0, c_xt, std

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, variance_noise=None):
        """Ancestral sampling step in the reference's batched-timestep form (scheduling_ddpm.py:285-418): x0 from the
        prediction, posterior mean (formula 7 of Ho et al.), plus sqrt(variance) * noise for the samples with t > 0.
        Noise is drawn like the reference (one randn of the t > 0 sub-batch's shape on the sample's device) unless
        `variance_noise` (full batch shape; rows with t == 0 ignored) is given."""
        B = sample.shape[0]
        t = self._t_index(timestep, B)
        a_t, b_t, c_x0, c_xt, std = self._coeffs(t)
        clamp = float(getattr(self.config, "clip_sample_range", 1.0)) if self.config.clip_sample else 0.0
        pt = self.config.prediction_type
        if pt == "epsilon":
            x0 = self._lincomb(sample, model_output, (a_t ** -0.5).numpy(), (-(b_t ** 0.5) / a_t ** 0.5).numpy(), clamp)
        elif pt == "sample":
            x0 = self._lincomb(model_output, sample, np.ones(B, np.float32), np.zeros(B, np.float32), clamp)
        elif pt == "v_prediction":
            x0 = self._lincomb(sample, model_output, (a_t ** 0.5).numpy(), (-(b_t ** 0.5)).numpy(), clamp)
        else:
            raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, `sample` or `v_prediction` "
                             "for the DDPMScheduler.")
        mean = self._lincomb(x0, sample, c_x0.numpy(), c_xt.numpy())
        # reference quirk kept for parity (:381-407): the noise rows are picked by indexing the batch with
        # (timestep > 0).nonzero() of the UN-expanded timestep, so a shared scalar timestep perturbs batch row 0 only
        t_raw = torch.as_tensor(timestep).detach().reshape(-1).to("cpu", torch.int64)
        pos = (t_raw > 0).nonzero().reshape(-1)
        if variance_noise is None:
            variance_noise = torch.zeros_like(mean)
            if pos.numel():
                variance_noise[pos.to(mean.device)] = torch.randn((pos.numel(),) + tuple(mean.shape[1:]), generator=generator,
                                                                  device=mean.device, dtype=mean.dtype)
        scale = torch.zeros_like(std)
        scale[pos] = std[pos]
        prev = self._lincomb(mean, variance_noise, np.ones(B, np.float32), scale.numpy())
        if not return_dict:
            return (prev,)
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)


class DDIMScheduler(_LinCombMixin):
    """scheduling_ddim.py in the reference's batched-timestep version: set_timesteps (:218-241, the steps_offset line is
    commented out upstream), step (:243-370) for v_prediction / epsilon with eta = 0, add_noise."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon",
                 clip_sample_range=1.0, **kw):
        self.betas, self.alphas_cumprod = _scaled_linear_alphas_cumprod(num_train_timesteps, beta_start, beta_end,
                                                                       beta_schedule, trained_betas)
        self.alphas = 1.0 - self.betas
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        if prediction_type not in ("v_prediction", "epsilon"):
            raise ValueError(f"prediction_type given as {prediction_type} must be one of `epsilon` or `v_prediction`")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, prediction_type=prediction_type,
                                      clip_sample=clip_sample, clip_sample_range=clip_sample_range,
                                      set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, subfolder=None, **kwargs):
        return cls(**dict(SD21_SCHEDULER_CONFIG, **SD21_DDIM_EXTRA, **kwargs))

    def set_timesteps(self, num_inference_steps, device=None):
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                             f"`self.config.train_timesteps`: {self.config.num_train_timesteps}")
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        self.timesteps = torch.from_numpy((np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64))

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if eta != 0.0:
            raise NotImplementedError("only the deterministic DDIM step (eta = 0) the reference uses is built")
        B = sample.shape[0]
        t = self._t_index(timestep, B)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = torch.where(prev >= 0, self.alphas_cumprod[prev.clamp(min=0)], self.final_alpha_cumprod)
        sa, sb = a_t ** 0.5, (1 - a_t) ** 0.5
        clamp = float(self.config.clip_sample_range) if self.config.clip_sample else 0.0
        if self.config.prediction_type == "v_prediction":
            x0 = self._lincomb(sample, model_output, sa.numpy(), (-sb).numpy(), clamp)
            eps = self._lincomb(model_output, sample, sa.numpy(), sb.numpy())
        else:  # epsilon
            x0 = self._lincomb(sample, model_output, (1.0 / sa).numpy(), (-sb / sa).numpy(), clamp)
            eps = model_output
        prev_sample = self._lincomb(x0, eps, (a_prev ** 0.5).numpy(), ((1 - a_prev) ** 0.5).numpy())
        if not return_dict:
            return (prev_sample,)
        return SimpleNamespace(prev_sample=prev_sample, pred_original_sample=x0)
