#!/usr/bin/env python3
"""Predicted data-parallel scaling of the distillation step from a MEASURED single-GPU backward timeline (VERDICT r2 #6;
no 8-GPU node is available to the build, so this is a model, not a measurement).

On the GPU box: one batch-9 distillation micro-step at the light widths with a HIP event recorded each time the engine
reports a finished U-Net block (`on_block_done`: out head, up blocks, mid, down blocks, conv_in + embeddings); the flat
gradient slice of each block is what `dist_util.GradientBuckets` all-reduces at that moment.  The model then replays the
bucket schedule against an xGMI all-reduce of the MI355X node (8 GPUs, 7 links x 153 GB/s per direction per GPU):
    ring   : one logical ring, per-link bound:  t = 2 * (N-1)/N * bytes / 153 GB/s
    direct : reduce-scatter + all-gather over all 7 links at once: t = 2 * (N-1)/N * bytes / (7 * 153 GB/s)
plus a fixed launch latency per collective, collectives serialised on RCCL's stream, each starting when its last block is
done.  Exposed time = what is left of the last collective after the backward pass has finished.

    python tools/overlap_model.py profiles/overlap_model_r03.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consistencytta_amd import spec  # noqa: E402
from consistencytta_amd.models import AudioLCM  # noqa: E402

LINK_GBPS, LINKS, LAT_US = 153.0, 7, 25.0


def simulate(ready_ms, nbytes, bwd_end_ms, n_gpus, gbps):
    t = 0.0
    for r, b in zip(ready_ms, nbytes):
        start = max(t, r)
        t = start + LAT_US * 1e-3 + 2.0 * (n_gpus - 1) / n_gpus * b / (gbps * 1e9) * 1e3
    return max(0.0, t - bwd_end_ms), t


def main():
    dev = "cuda:0"
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
    opt = m.prepare_training(lr=1e-5, weight_decay=1e-4, broadcast=False)
    g = torch.Generator(device="cpu").manual_seed(5)
    z0 = (torch.randn(B, 8, 256, 16, generator=g) * 0.9).to(dev)
    enc = (torch.randn(B, L, 1024, generator=g) * 0.25).to(dev)
    mask = torch.ones(B, L, dtype=torch.bool, device=dev)
    unc, umask = torch.zeros_like(enc), torch.zeros_like(mask)
    umask[:, 0] = True
    P = {"embeds_cf": torch.cat([unc, enc]), "mask_cf": torch.cat([umask, mask]), "embeds": enc, "mask": mask}
    for _ in range(2):
        m.train_step(z0, P, opt)
    ranges = dict(m.student_unet.block_ranges())
    runs = []
    for _ in range(3):
        torch.cuda.synchronize()
        t_step0 = torch.cuda.Event(enable_timing=True)
        t_step0.record()
        with torch.no_grad():
            loss, pred, target, sig, gamma = m._forward_impl(z0, None, P, False, True, None, None, None, True)
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
            marks = []

            def done(block):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append((block, e))
            m._student_backward(pred, target, sig, gamma, 1.0, done)
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
        opt.step()
        opt.zero_grad()
        m.update_ema()
        ev2 = torch.cuda.Event(enable_timing=True)
        ev2.record()
        torch.cuda.synchronize()
        runs.append({"forward_ms": t_step0.elapsed_time(ev0), "backward_ms": ev0.elapsed_time(ev1), "step_ms": t_step0.elapsed_time(ev2),
                     "blocks": [(b, ev0.elapsed_time(e)) for b, e in marks]})
    r = sorted(runs, key=lambda x: x["step_ms"])[1]
    blocks = [{"block": int(b), "done_ms_after_backward_start": round(t, 3), "grad_elems": int(ranges[b][1] - ranges[b][0])}
              for b, t in r["blocks"] if b in ranges]
    out = {"what": "MODEL of the 8-GPU distillation step from a measured 1-GPU backward timeline (no multi-GPU measurement exists)",
           "batch_per_gpu": B, "step_ms_1gpu": round(r["step_ms"], 3), "forward_ms": round(r["forward_ms"], 3),
           "backward_ms": round(r["backward_ms"], 3), "blocks": blocks,
           "link_GBps": LINK_GBPS, "links_per_gpu": LINKS, "collective_latency_us": LAT_US, "predictions": []}
    for min_elems in (4 << 20, 16 << 20, 64 << 20):
        # GradientBuckets: blocks merged until a bucket holds >= min_elems elements
        ready, sizes, acc = [], [], 0
        for blk in blocks:
            acc += blk["grad_elems"]
            if acc >= min_elems:
                ready.append(blk["done_ms_after_backward_start"])
                sizes.append(acc)
                acc = 0
        if acc:
            ready.append(blocks[-1]["done_ms_after_backward_start"])
            sizes.append(acc)
        for dtype, bpe in (("fp32", 4), ("bf16", 2)):
            for name, gbps in (("ring (1 link)", LINK_GBPS), ("direct (7 links)", LINK_GBPS * LINKS)):
                row = {"min_elems": min_elems, "collectives": len(sizes), "dtype": dtype, "algorithm": name}
                for n in (2, 4, 8):
                    exposed, _ = simulate(ready, [s * bpe for s in sizes], r["backward_ms"], n, gbps)
                    step = r["step_ms"] + exposed
                    row["n%d" % n] = {"exposed_ms": round(exposed, 3), "step_ms": round(step, 3),
                                      "scaling_vs_1gpu": round(n * r["step_ms"] / step, 3)}
                out["predictions"].append(row)
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
        open(sys.argv[1], "w").write(text)
    print(text[:3000])


if __name__ == "__main__":
    main()
