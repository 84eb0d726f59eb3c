"""Worker of tests/test_dist_gpu.py: one rank of a 2-rank data-parallel distillation step, both ranks sharing GPU 0
with the collectives routed through the host (gloo) -- the same `AudioLCM.train_step` code path that runs over RCCL on
an 8-GPU node (block-wise backward, bucketed asynchronous all-reduce, NaN flag, fused AdamW, EMA).  Started as a fresh
child process per rank; writes its results to `<out>/rank<r>.pt`.

    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dist_gpu_worker.py <mode> <out_dir>
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import cases  # noqa: E402
from consistencytta_amd import dist_util as du  # noqa: E402
from consistencytta_amd import spec  # noqa: E402
from consistencytta_amd.models import AudioLCM  # noqa: E402

GLOBAL_B, L = 4, 6


def build(dev, seed_shift=0):
    cfg = cases.TINY_UNET
    m = AudioLCM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                 unet_model_config_path="tiny_light.json", unet_config=cfg, snr_gamma=5.0, use_edm=True,
                 teacher_guidance_scale=-1, num_diffusion_steps=18, vae=None, loss_type="mse",
                 target_ema_decay=0.95, ema_decay=0.999)
    m.teacher_unet.load_state_dict(cases.unet_weights(cfg, False, 0))
    # rank-dependent student weights: prepare_training must replace them by rank 0's (DDP's wrap-time broadcast)
    m.student_unet.load_state_dict(cases.unet_weights(cfg, True, 1 + seed_shift))
    m.student_target_unet.load_state_dict(cases.unet_weights(cfg, True, 2 + seed_shift))
    m.student_ema_unet.load_state_dict(cases.unet_weights(cfg, True, 3 + seed_shift))
    return m.to(dev)


def global_batch():
    cfg = cases.TINY_UNET
    P = cases.prompt_states(cfg, GLOBAL_B, L, "dist")
    z0 = cases.t(spec.det_uniform("dist.z0", (GLOBAL_B, 8, 32, 8), 14)) * 0.9
    g = torch.Generator().manual_seed(77)
    draws = dict(time_inds=torch.randint(0, 17, (GLOBAL_B,), generator=g) * 2,
                 gaussian_noise=torch.randn(GLOBAL_B, 8, 32, 8, generator=g),
                 guidance_scale=torch.rand(GLOBAL_B, generator=g) * 6)
    return P, z0, draws


def shard(P, z0, draws, lo, hi, dev):
    n = GLOBAL_B
    Ps = {"embeds_cf": torch.cat([P["embeds_cf"][:n][lo:hi], P["embeds_cf"][n:][lo:hi]]).to(dev),
          "mask_cf": torch.cat([P["mask_cf"][:n][lo:hi], P["mask_cf"][n:][lo:hi]]).to(dev),
          "embeds": P["embeds"][lo:hi].to(dev), "mask": P["mask"][lo:hi].to(dev)}
    kw = dict(time_inds=draws["time_inds"][lo:hi], gaussian_noise=draws["gaussian_noise"][lo:hi].to(dev),
              guidance_scale=draws["guidance_scale"][lo:hi])
    return Ps, z0[lo:hi].to(dev), kw


def run_step(m, Ps, z, kw, lr=1e-4, compress=None, graph=False):
    """One train_step (graph=True: the segmented hipGraph replay of it, `capture_train_graph`); returns (loss, the
    all-reduced flat gradient AdamW consumed x 1/world, flat params after)."""
    opt = m.prepare_training(lr=lr, weight_decay=0.0, broadcast=True)
    m.allreduce_dtype = compress
    gs = None
    if graph:   # captured with OTHER draws than the step's: the replay must pick up the refreshed static tensors
        g = torch.Generator().manual_seed(5)
        gs = m.capture_train_graph(opt, z, Ps, time_inds=torch.randint(0, 17, (z.shape[0],), generator=g) * 2,
                                   gaussian_noise=torch.randn(z.shape, generator=g).to(z.device),
                                   guidance_scale=torch.rand(z.shape[0], generator=g) * 6, bucket_min_elems=1)
        assert gs.segmented and len(gs.segments) == 2 * len(cases.TINY_UNET["block_out_channels"]) + 3
    seen = {}
    orig = opt.step

    def step(grad_scale=1.0):
        seen["grad"] = (opt.grad.detach() * grad_scale).clone()
        return orig(grad_scale=grad_scale)
    opt.step = step
    before = opt.flat.detach().clone()
    loss = gs.step(z, None, **kw) if gs is not None else m.train_step(z, Ps, opt, None, **kw)
    torch.cuda.synchronize()
    return loss, seen.get("grad"), opt.flat.detach().clone(), before, opt


def tiny_vocoder(dev):
    from consistencytta_amd import modules
    sd = dict(cases.vae_weights(cases.TINY_VAE_DD))
    sd.update(cases.hifigan_weights(cases.TINY_HIFIGAN))
    v = modules.AutoencoderKL(ddconfig=cases.TINY_VAE_DD, embed_dim=8, scale_factor=0.9, hifigan_config=cases.TINY_HIFIGAN)
    v.load_state_dict(sd)
    return v.to(dev).eval().requires_grad_(False)


def wav_batch():
    mel = cases.t(spec.det_uniform("dist.mel", (4, 1, 64, cases.TINY_HIFIGAN["num_mels"]), 33)) * 2.0 - 5.0
    mel[3] += 1.5          # the loudest clip lives on rank 1: rank 0 cannot centre correctly without the exchange
    return mel


def main():
    mode, out = sys.argv[1], sys.argv[2]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    world, rank = du.init("gloo")
    assert world == 2
    if mode == "wav":     # clip-sharded generation with the opt-in batch-global centring (hifigan/utilities.py:85)
        v = tiny_vocoder(dev)
        mel = wav_batch()[rank * 2:(rank + 1) * 2].to(dev)
        res = {"world": torch.from_numpy(v.decode_to_waveform(mel, world_extrema=True)),
               "local": torch.from_numpy(v.decode_to_waveform(mel))}
        torch.save(res, os.path.join(out, "rank%d.pt" % rank))
        du.finish()
        return
    m = build(dev, seed_shift=10 * rank)
    m.train()
    P, z0, draws = global_batch()
    per = GLOBAL_B // world
    Ps, z, kw = shard(P, z0, draws, rank * per, (rank + 1) * per, dev)
    if mode == "nan" and rank == 1:
        z = z.clone()
        z[0, 0, 0, 0] = float("nan")
    compress = torch.bfloat16 if mode == "bf16" else None
    loss, grad, after, before, opt = run_step(m, Ps, z, kw, compress=compress, graph=(mode == "graph"))
    res = {"loss": loss, "grad": None if grad is None else grad.cpu(), "after": after.cpu(), "before": before.cpu(),
           "target_after": m.student_target_unet._flat.detach().cpu(), "step_count": opt.step_count}
    torch.save(res, os.path.join(out, "rank%d.pt" % rank))
    du.finish()


if __name__ == "__main__":
    main()
