#!/usr/bin/env python3
"""One rank, real RCCL process group, CTTA_FORCE_COLLECTIVES=1: the segmented (+ pipelined) distillation step at the real size
(light U-Nets, batch 9) with every bucket going through ncclAllReduce between graph replays -- step time against the eager
train_step under the same forced collectives.  (The 2-rank gloo rehearsal on one GPU cannot time this: two processes share
the device and gloo moves 2.2 GB per step through the host.)"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistencytta_amd import spec  # noqa: E402
from consistencytta_amd.models import AudioLCM  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ["CTTA_FORCE_COLLECTIVES"] = "1"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
B, L = 9, 32
m = AudioLCM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
             unet_model_config_path="tango_diffusion_light.json", unet_config=spec.LIGHT_UNET_CONFIG, snr_gamma=5.0,
             use_edm=True, teacher_guidance_scale=-1, num_diffusion_steps=18, vae=None, loss_type="mse",
             target_ema_decay=0.95, ema_decay=0.999)
m.to(dev)
m.teacher_unet.init_random_(seed=10)
m.student_unet.init_random_(seed=11)
with torch.no_grad():
    for dst in (m.student_target_unet, m.student_ema_unet):
        for p, q in zip(dst.parameters(), m.student_unet.parameters()):
            p.copy_(q)
m.train()
opt = m.prepare_training(lr=1e-5, weight_decay=1e-4, broadcast=True)
g = torch.Generator().manual_seed(5)
z0 = (torch.randn(B, 8, 256, 16, generator=g) * 0.9).to(dev)
enc = (torch.randn(B, L, 1024, generator=g) * 0.25).to(dev)
mask = torch.ones(B, L, dtype=torch.bool, device=dev)
unc, um = torch.zeros_like(enc), torch.zeros_like(mask)
um[:, 0] = True
P = {"embeds_cf": torch.cat([unc, enc]), "mask_cf": torch.cat([um, mask]), "embeds": enc, "mask": mask}


def timed(fn, n=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("eager train_step, forced RCCL buckets: %.1f ms" % timed(lambda: m.train_step(z0, P, opt, None)))
for pipe in (False, True):
    gs = m.capture_train_graph(opt, z0, P, pipeline_teacher=pipe)
    assert gs.segmented and len(gs.segments) == 8
    print("segmented%s graph step, forced RCCL buckets: %.1f ms" % (" + pipelined" if pipe else "", timed(lambda: gs.step(z0, None))))
    del gs
dist.destroy_process_group()
