"""Helpers for the GPU parity tests: raw C-ABI calls on torch-owned device memory."""
import ctypes

import numpy as np
import torch

from consistencytta_amd import _native as N

DEV = "cuda:0"


def sync():
    torch.cuda.synchronize()


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def nhwc_bf16(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def from_nhwc(x_nhwc):
    return x_nhwc.to(torch.float32).permute(0, 3, 1, 2).contiguous().cpu()


def pack_conv_weight(w, cin_pad=None, k_mult=64):
    """(cout, cin, kh, kw) fp32 -> bf16 [cout][k_pad] with k = (kh, kw, c) on the host."""
    cout, cin, kh, kw = w.shape
    cin_pad = cin_pad or cin
    wp = torch.zeros(cout, kh, kw, cin_pad)
    wp[:, :, :, :cin] = w.permute(0, 2, 3, 1)
    wp = wp.reshape(cout, kh * kw * cin_pad)
    k_pad = (wp.shape[1] + k_mult - 1) // k_mult * k_mult
    out = torch.zeros(cout, k_pad)
    out[:, :wp.shape[1]] = wp
    return out.to(torch.bfloat16).to(DEV), k_pad


def conv_desc(**kw):
    d = N.ConvDesc()
    d.kh = d.kw = 1
    d.stride_h = d.stride_w = 1
    d.dil_h = d.dil_w = 1
    d.alpha = 1.0
    d.groups = 1
    for k, v in kw.items():
        if torch.is_tensor(v):
            v = v.data_ptr()
        setattr(d, k, v)
    return d


def run_conv(d):
    N.check(N.lib().ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
    sync()


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def rel_l2(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def det(name, shape, seed=0, scale=1.0):
    from consistencytta_amd import spec
    return torch.from_numpy(spec.det_uniform(name, shape, seed)) * scale
