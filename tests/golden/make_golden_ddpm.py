"""Fixture for `DDPMScheduler.step` (SURVEY.md §8f rank 3 leftover), produced by the REFERENCE's own batched-timestep
`diffusers/schedulers/scheduling_ddpm.py:285-418` (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ddpm.py

The noise the reference draws inside `step` (one randn over the t > 0 sub-batch) is recorded and scattered to the full
batch shape, so other implementations replay it through `variance_noise=`."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
from consistencytta_amd import spec  # noqa: E402
from make_golden_gdm import SD21, load_reference_schedulers  # noqa: E402


def main():
    DDPM, _, _ = load_reference_schedulers()
    import diffusers.schedulers.scheduling_ddpm as M
    out = {}
    B = 4
    x = cases.t(spec.det_uniform("ddpm.x", (B, 8, 16, 4), 1)) * 2
    v = cases.t(spec.det_uniform("ddpm.v", (B, 8, 16, 4), 3))
    rec = {}
    orig = M.randn_tensor

    def randn_tensor(shape, **k):
        n = orig(shape, **k)
        rec["noise"] = n.clone()
        return n
    M.randn_tensor = randn_tensor
    try:
        with torch.no_grad():
            for tag, kw, steps, t in (
                    ("v_full", dict(SD21), None, torch.tensor([999, 400, 0, 1])),
                    ("v_50", dict(SD21), 50, torch.tensor([980, 0, 20, 500])),
                    # epsilon: the reference leaves the (B,) coefficients un-reshaped (scheduling_ddpm.py:330-333), so only a
                    # single shared timestep broadcasts correctly there
                    ("eps_clip", dict(SD21, prediction_type="epsilon", clip_sample=True), 20, torch.tensor([350])),
                    ("v_large", dict(SD21, variance_type="fixed_large"), 10, torch.tensor([900, 0, 100, 300]))):
                sch = DDPM(**kw)
                if steps:
                    sch.set_timesteps(steps)
                torch.manual_seed(11)
                r = sch.step(v, t, x)
                noise = torch.zeros_like(x)
                if t.numel() == 1:      # a shared timestep indexes batch row 0 only (:381-407): one row gets noise
                    noise[0] = rec["noise"][0]
                else:
                    noise[(t > 0).nonzero().reshape(-1)] = rec["noise"]
                out[tag + "_t"] = t.numpy()
                out[tag + "_noise"] = noise.numpy()
                out[tag + "_prev"] = r.prev_sample.numpy()
                out[tag + "_x0"] = r.pred_original_sample.numpy()
            sch = DDPM(**SD21)
            sch.set_timesteps(50)
            out["scalar_t_prev"] = sch.step(v, 0, x).prev_sample.numpy()      # t = 0: no noise term at all
    finally:
        M.randn_tensor = orig
    path = os.path.join(HERE, "ddpm_step.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
