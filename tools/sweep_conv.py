#!/usr/bin/env python3
"""Times every conv_gemm tile variant on the layer shapes of the B=32 pipeline (GPU box only).

    python tools/sweep_conv.py [out.json]

Prints one line per shape with the TFLOP/s of each variant (executed FLOPs, random data) and
writes the table as JSON; the result drives pick_variant() in csrc/conv_gemm.hip."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistencytta_amd import _native as N  # noqa: E402

DEV = "cuda:0"

# (tag, batch, H, W, Cin, Cout, kh, kw, dil_w)   -- NHWC, stride 1, "same" padding
SHAPES = [
    ("vae512 256x16 3x3", 32, 256, 16, 512, 512, 3, 3, 1),
    ("vae 512x32 512>256", 32, 512, 32, 512, 256, 3, 3, 1),
    ("vae 512x32 256>256", 32, 512, 32, 256, 256, 3, 3, 1),
    ("vae 1024x64 256>128", 32, 1024, 64, 256, 128, 3, 3, 1),
    ("vae 1024x64 128>128", 32, 1024, 64, 128, 128, 3, 3, 1),
    ("vae 1024x64 256>256", 32, 1024, 64, 256, 256, 3, 3, 1),
    ("unet 256x16 256>256", 32, 256, 16, 256, 256, 3, 3, 1),
    ("unet 256x16 512>256", 32, 256, 16, 512, 256, 3, 3, 1),
    ("unet 128x8 512>512", 32, 128, 8, 512, 512, 3, 3, 1),
    ("unet 64x4 1024>1024", 32, 64, 4, 1024, 1024, 3, 3, 1),
    ("unet 32x2 2048>1024", 32, 32, 2, 2048, 1024, 3, 3, 1),
    ("hifi L5121 C512 k11", 32, 1, 5121, 512, 512, 1, 11, 1),
    ("hifi L5121 C512 k3", 32, 1, 5121, 512, 512, 1, 3, 3),
    ("hifi L20484 C256 k11", 32, 1, 20484, 256, 256, 1, 11, 5),
    ("hifi L20484 C256 k3", 32, 1, 20484, 256, 256, 1, 3, 1),
    ("hifi L40968 C128 k11", 32, 1, 40968, 128, 128, 1, 11, 1),
    ("hifi L40968 C128 k3", 32, 1, 40968, 128, 128, 1, 3, 1),
    ("hifi L81936 C64 k11", 32, 1, 81936, 64, 64, 1, 11, 3),
    ("hifi L81936 C64 k3", 32, 1, 81936, 64, 64, 1, 3, 1),
    ("hifi L163872 C32 k11", 32, 1, 163872, 32, 32, 1, 11, 1),
    ("hifi L163872 C32 k3", 32, 1, 163872, 32, 32, 1, 3, 5),
    ("lin M131072 256>256", 32, 4096, 1, 256, 256, 1, 1, 1),
    ("lin M131072 256>640", 32, 4096, 1, 256, 640, 1, 1, 1),
    ("lin M131072 256>2048", 32, 4096, 1, 256, 2048, 1, 1, 1),
    ("lin M131072 1024>256", 32, 4096, 1, 1024, 256, 1, 1, 1),
    ("lin M32768 512>4096", 32, 1024, 1, 512, 4096, 1, 1, 1),
    ("lin M32768 2048>512", 32, 1024, 1, 2048, 512, 1, 1, 1),
    ("lin M8192 1024>8192", 32, 256, 1, 1024, 8192, 1, 1, 1),
    ("lin M8192 4096>1024", 32, 256, 1, 4096, 1024, 1, 1, 1),
    ("lin M2048 1024>8192", 32, 64, 1, 1024, 8192, 1, 1, 1),
    # round 3: the transformer linears of the batch-32 U-Net that the first list lacked (SWEEP_FILTER=u32)
    ("u32 M131072 320>256", 32, 4096, 1, 320, 256, 1, 1, 1),
    ("u32 M131072 256>320", 32, 4096, 1, 256, 320, 1, 1, 1),
    ("u32 M32768 512>512", 32, 1024, 1, 512, 512, 1, 1, 1),
    ("u32 M32768 640>512", 32, 1024, 1, 640, 512, 1, 1, 1),
    ("u32 M32768 512>1280", 32, 1024, 1, 512, 1280, 1, 1, 1),
    ("u32 M8192 1024>1024", 32, 256, 1, 1024, 1024, 1, 1, 1),
    ("u32 M8192 1280>1024", 32, 256, 1, 1280, 1024, 1, 1, 1),
    ("u32 M8192 1024>2560", 32, 256, 1, 1024, 2560, 1, 1, 1),
    ("u32 conv 64x4 1024>1024", 32, 64, 4, 1024, 1024, 3, 3, 1),
    ("u32 conv 32x2 1024>1024", 32, 32, 2, 1024, 1024, 3, 3, 1),
    # the Heun teacher's batch (8 clips x 2 CFG halves = 16): SWEEP_FILTER=t16
    ("t16 conv 128x8 512>512", 16, 128, 8, 512, 512, 3, 3, 1),
    ("t16 conv 128x8 1024>512", 16, 128, 8, 1024, 512, 3, 3, 1),
    ("t16 conv 64x4 1024>1024", 16, 64, 4, 1024, 1024, 3, 3, 1),
    ("t16 conv 64x4 2048>1024", 16, 64, 4, 2048, 1024, 3, 3, 1),
    ("t16 conv 32x2 1024>1024", 16, 32, 2, 1024, 1024, 3, 3, 1),
    ("t16 lin M16384 512>512", 16, 1024, 1, 512, 512, 1, 1, 1),
    ("t16 lin M16384 2048>512", 16, 1024, 1, 2048, 512, 1, 1, 1),
    ("t16 lin M16384 640>512", 16, 1024, 1, 640, 512, 1, 1, 1),
    ("t16 lin M4096 1024>1024", 16, 256, 1, 1024, 1024, 1, 1, 1),
    ("t16 lin M4096 4096>1024", 16, 256, 1, 4096, 1024, 1, 1, 1),
    ("t16 lin M4096 1280>1024", 16, 256, 1, 1280, 1024, 1, 1, 1),
    ("t16 lin M4096 1024>2560", 16, 256, 1, 1024, 2560, 1, 1, 1),
    # the distillation teacher's batch (9 latents x 2 CFG halves = 18): SWEEP_FILTER=t18
    ("t18 conv 256x16 256>256", 18, 256, 16, 256, 256, 3, 3, 1),
    ("t18 conv 256x16 512>256", 18, 256, 16, 512, 256, 3, 3, 1),
    ("t18 conv 128x8 512>512", 18, 128, 8, 512, 512, 3, 3, 1),
    ("t18 conv 128x8 1024>512", 18, 128, 8, 1024, 512, 3, 3, 1),
    ("t18 conv 64x4 1024>1024", 18, 64, 4, 1024, 1024, 3, 3, 1),
    ("t18 conv 32x2 1024>1024", 18, 32, 2, 1024, 1024, 3, 3, 1),
    ("t18 lin M73728 256>256", 18, 4096, 1, 256, 256, 1, 1, 1),
    ("t18 lin M73728 256>640", 18, 4096, 1, 256, 640, 1, 1, 1),
    ("t18 lin M73728 1024>256", 18, 4096, 1, 1024, 256, 1, 1, 1),
    ("t18 lin M73728 320>256", 18, 4096, 1, 320, 256, 1, 1, 1),
    ("t18 lin M18432 512>512", 18, 1024, 1, 512, 512, 1, 1, 1),
    ("t18 lin M18432 2048>512", 18, 1024, 1, 2048, 512, 1, 1, 1),
    ("t18 lin M4608 1024>1024", 18, 256, 1, 1024, 1024, 1, 1, 1),
    ("t18 lin M4608 4096>1024", 18, 256, 1, 4096, 1024, 1, 1, 1),
    ("t9 conv 32x2 1024>1024", 9, 32, 2, 1024, 1024, 3, 3, 1),
    ("t9 conv 64x4 2048>1024", 9, 64, 4, 2048, 1024, 3, 3, 1),
    ("t9 conv 256x16 512>256", 9, 256, 16, 512, 256, 3, 3, 1),
    # round 4: the thin K-heavy rows of launch_table_distill (SWEEP_FILTER=thin)
    ("thin conv 32x2 b9 1024>1024", 9, 32, 2, 1024, 1024, 3, 3, 1),
    ("thin conv 32x2 b18 1024>1024", 18, 32, 2, 1024, 1024, 3, 3, 1),
    ("thin conv 64x4 b9 1024>1024", 9, 64, 4, 1024, 1024, 3, 3, 1),
    ("thin conv 64x4 b18 1024>1024", 18, 64, 4, 1024, 1024, 3, 3, 1),
    ("thin conv 128x8 b9 512>512", 9, 128, 8, 512, 512, 3, 3, 1),
    ("thin conv 128x8 b18 512>512", 18, 128, 8, 512, 512, 3, 3, 1),
    ("thin conv 256x16 b9 256>256", 9, 256, 16, 256, 256, 3, 3, 1),
    ("thin conv 256x16 b18 256>256", 18, 256, 16, 256, 256, 3, 3, 1),
    ("thin lin M2304 1280>1024", 9, 256, 1, 1280, 1024, 1, 1, 1),
    ("thin lin M2304 1024>1024", 9, 256, 1, 1024, 1024, 1, 1, 1),
    ("thin lin M2304 4096>1024", 9, 256, 1, 4096, 1024, 1, 1, 1),
    ("thin lin M9216 640>512", 9, 1024, 1, 640, 512, 1, 1, 1),
    ("thin lin M9216 512>512", 9, 1024, 1, 512, 512, 1, 1, 1),
    ("thin lin M36864 320>256", 9, 4096, 1, 320, 256, 1, 1, 1),
    ("thin lin M36864 256>256", 9, 4096, 1, 256, 256, 1, 1, 1),
    ("thin lin M73728 320>256", 18, 4096, 1, 320, 256, 1, 1, 1),
    # fused GEGLU (SWEEP_GEGLU=1 SWEEP_FILTER=ff1): N counts value + gate columns
    ("ff1 M131072 256>2048", 32, 4096, 1, 256, 2048, 1, 1, 1),
    ("ff1 M32768 512>4096", 32, 1024, 1, 512, 4096, 1, 1, 1),
    ("ff1 M8192 1024>8192", 32, 256, 1, 1024, 8192, 1, 1, 1),
    ("ff1 M36864 256>2048", 9, 4096, 1, 256, 2048, 1, 1, 1),
    ("ff1 M65536 256>2048", 16, 4096, 1, 256, 2048, 1, 1, 1),
    # distillation micro-batch (B = 9 student / 18 teacher-CFG): SWEEP_FILTER=d9
    ("d9 conv 128x8 512>512", 9, 128, 8, 512, 512, 3, 3, 1),
    ("d9 conv 256x16 256>256", 9, 256, 16, 256, 256, 3, 3, 1),
    ("d9 conv 64x4 1024>1024", 9, 64, 4, 1024, 1024, 3, 3, 1),
    ("d9 lin M9216 640>512", 9, 1024, 1, 640, 512, 1, 1, 1),
    ("d9 lin M36864 320>256", 9, 4096, 1, 320, 256, 1, 1, 1),
    ("d9 lin M36864 256>256", 9, 4096, 1, 256, 256, 1, 1, 1),
    ("d9 lin M9216 512>512", 9, 1024, 1, 512, 512, 1, 1, 1),
    ("d9 lin M2304 1280>1024", 9, 256, 1, 1280, 1024, 1, 1, 1),
    ("d9 lin M4608 1024>1024", 18, 256, 1, 1024, 1024, 1, 1, 1),
    ("d9 lin M36864 256>2048", 9, 4096, 1, 256, 2048, 1, 1, 1),
    ("d9 lin M9216 2048>512", 9, 1024, 1, 2048, 512, 1, 1, 1),
]


def main():
    # SWEEP_FILTER=substring restricts the shapes; SWEEP_VARIANTS=17,18,... the variants; SWEEP_EPI=1 adds the
    # HiFi-GAN ResBlock epilogue (residual read + second, LeakyReLU'd output) to every launch
    L = N.lib()
    nvar = L.ctta_conv_gemm_num_variants()
    names = [L.ctta_conv_gemm_variant_name(i + 1).decode() for i in range(nvar)]
    flt = os.environ.get("SWEEP_FILTER", "")
    only = [int(v) for v in os.environ.get("SWEEP_VARIANTS", "").split(",") if v]
    epi = os.environ.get("SWEEP_EPI", "0") == "1"
    geglu = os.environ.get("SWEEP_GEGLU", "0") == "1"
    print("variants:", names if not only else [names[v - 1] for v in only])
    results = []
    for (tag, B, H, W, Cin, Cout, kh, kw, dil) in SHAPES:
        if flt and flt not in tag:
            continue
        x = (torch.randn(B, H, W, Cin, device=DEV) * 0.5).to(torch.bfloat16)
        K = kh * kw * Cin
        k_pad = (K + 63) // 64 * 64
        # SWEEP_COLD=1: every launch reads ANOTHER copy of the weights (> 512 MB of copies in rotation: past the 256 MB
        # Infinity Cache), as inside the pipeline where a layer's weights arrive cold from HBM
        ncopy = 1
        if os.environ.get("SWEEP_COLD", "0") == "1":
            ncopy = max(2, min(64, (600 << 20) // (Cout * k_pad * 2) + 1))
        ws = [(torch.randn(Cout, k_pad, device=DEV) * 0.05).to(torch.bfloat16) for _ in range(ncopy)]
        w = ws[0]
        bias = torch.randn(Cout, device=DEV)
        out = torch.empty(B, H, W, Cout // 2 if geglu else Cout, dtype=torch.bfloat16, device=DEV)
        # SWEEP_COLD=2: the ACTIVATIONS rotate too (input and output copies past the Infinity Cache together), as in the
        # pipeline where a layer's input was written once by its producer and its output goes to arena memory
        xs, outs = [x], [out]
        if os.environ.get("SWEEP_COLD", "0") == "2":
            per = x.numel() * 2 + out.numel() * 2
            nact = max(2, min(16, (600 << 20) // per + 1))
            ncopy = max(2, min(64, (600 << 20) // (Cout * k_pad * 2) + 1))
            ws = ws + [ws[0].clone() for _ in range(ncopy - 1)]
            xs = [x] + [x.clone() for _ in range(nact - 1)]
            outs = [out] + [torch.empty_like(out) for _ in range(nact - 1)]
        M = B * H * W
        flops = 2.0 * M * Cout * K
        row = {"tag": tag, "M": M, "N": Cout, "K": K, "tflops": {}}
        res = torch.randn(B, H, W, Cout, device=DEV).to(torch.bfloat16) if epi else None
        out2 = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=DEV) if epi else None
        for v in ([0] + only if only else range(0, nvar + 1)):
            d = N.ConvDesc()
            d.x0, d.c0 = x.data_ptr(), Cin
            d.batch, d.hi, d.wi, d.ho, d.wo = B, H, W, H, W
            d.kh, d.kw, d.stride_h, d.stride_w = kh, kw, 1, 1
            d.dil_h, d.dil_w = 1, dil
            d.pad_h, d.pad_w = (kh - 1) // 2, (kw - 1) * dil // 2
            d.w, d.k_pad, d.n = w.data_ptr(), k_pad, Cout
            d.bias = bias.data_ptr()
            d.alpha, d.groups = 1.0, 1
            if epi:
                d.res, d.res_ld, d.out2, d.out2_slope = res.data_ptr(), Cout, out2.data_ptr(), 0.1
            d.out, d.ldc, d.tile = out.data_ptr(), (Cout // 2 if geglu else Cout), v
            if geglu:
                d.out_act = 4
            st = N.stream_ptr()
            vname = names[v - 1] if v > 0 else "auto"
            if L.ctta_conv_gemm(ctypes.byref(d), st) != 0:   # variant not eligible for this shape
                row["tflops"][vname] = 0.0
                continue
            N.check(L.ctta_conv_gemm(ctypes.byref(d), st))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5 if ncopy == 1 else 2 * ncopy
            e0.record()
            for r_ in range(reps):
                d.w = ws[r_ % ncopy].data_ptr()
                d.x0, d.out = xs[r_ % len(xs)].data_ptr(), outs[r_ % len(outs)].data_ptr()
                N.check(L.ctta_conv_gemm(ctypes.byref(d), st))
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            row["tflops"][vname] = round(flops / (ms * 1e-3) / 1e12, 1)
        best = max(row["tflops"], key=row["tflops"].get)
        row["best"] = best
        results.append(row)
        if os.environ.get("SWEEP_BRIEF", "0") == "1":
            print("%-24s M=%8d N=%5d K=%6d  auto %6.0f  best %-22s %6.0f" % (tag, M, Cout, K, row["tflops"].get("auto", 0.0),
                  best, row["tflops"][best]), flush=True)
        else:
            print("%-24s M=%8d N=%5d K=%6d  best %-22s %s" % (tag, M, Cout, K, best,
                  " ".join("%6.0f" % t for t in row["tflops"].values())), flush=True)
        del x, w, ws, out, xs, outs
    if len(sys.argv) > 1:
        json.dump({"variants": names, "shapes": results}, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
