#!/usr/bin/env python3
"""How far do two bf16 runs of the light U-Net drift apart when only the fp32 accumulation ORDER changes?
Runs one clip (B=1) and writes the latent; run it under different CTTA_OPT_SPLITK / CTTA_OPT_XCD settings (library options, applied by _native at load) and
compare the files (tools/order_noise.py out.pt [compare.pt])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases  # noqa: E402
from consistencytta_amd import modules, spec  # noqa: E402
from consistencytta_amd.models import ConsistencyTTA  # noqa: E402

DEV = torch.device("cuda", 0)
cfg = spec.LIGHT_UNET_CONFIG
vae = modules.AutoencoderKL(ddconfig=spec.VAE_DDCONFIG, embed_dim=8, scale_factor=0.9227914214134216)
pipe = ConsistencyTTA(unet_config=cfg, vae=vae)
pipe.to(DEV)
pipe.unet.load_state_dict({k: v.to(DEV) for k, v in cases.unet_weights(cfg, True).items()})
pipe.eval().requires_grad_(False)
B = int(os.environ.get("ORDER_B", "1"))
gen = torch.Generator().manual_seed(3)
enc = torch.randn(32, 32, 1024, generator=gen) * 0.25
lens = torch.randint(6, 33, (32,), generator=gen)
mask = torch.arange(32)[None, :] < lens[:, None]
noise = torch.randn(32, 8, 256, 16, generator=gen)
lat = pipe.generate_latent(enc[:B].to(DEV), mask[:B].to(DEV), noise[:B].to(DEV), cfg_scale_input=4.0, cfg_scale_post=1.0,
                           num_steps=1)[:1].float().cpu()
torch.save(lat, sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    print("%s vs %s: rel_l2 %.3e  max %.3e" % (sys.argv[1], sys.argv[2], float((lat - ref).norm() / ref.norm()),
                                               float((lat - ref).abs().max())))
