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
        s = self._sigmas_host[idx]
        if s.shape[0] == 1 and B > 1:
            s = np.repeat(s, B)
        return torch.from_numpy(np.ascontiguousarray(s, dtype=np.float32)).to(device, non_blocking=True)

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
