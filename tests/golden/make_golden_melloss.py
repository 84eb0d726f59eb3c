"""Fixture for the perceptual training loss built on the differentiable decode, produced by
the REFERENCE's own tools.losses.MelLoss over its own AutoencoderKL (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_melloss.py

Inputs / weights are regenerated from names and seeds (cases.py); the fixture holds the per-instance losses and the
gradient of their SNR-weighted mean with respect to the predicted latent.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
from consistencytta_amd import spec  # noqa: E402
import make_golden  # noqa: E402
import make_golden_distill  # noqa: E402


def main():
    ns, _, _ = make_golden_distill.load_reference_audiolcm()
    import tools.losses as RL
    vae, sf = make_golden.ref_vae(ns, cases.TINY_VAE_DD, cases.TINY_HIFIGAN)
    pred = (cases.vae_inputs(2, 16, 16, "melloss.pred") * 0.5).requires_grad_(True)
    target = cases.vae_inputs(2, 16, 16, "melloss.target") * 0.5
    weights = torch.tensor([0.75, 2.5])
    loss = RL.MelLoss(vae=vae, reduction="instance")
    inst = loss(pred, target, None, None)
    (inst * weights).mean().backward()
    path = os.path.join(HERE, "melloss_tiny.npz")
    np.savez_compressed(path, instance_loss=inst.detach().numpy(), weights=weights.numpy(), grad_pred=pred.grad.numpy(),
                        scale_factor=sf)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    main()
