"""CLAP on the HIP kernels: the audio tower (HTSAT-base Swin transformer with its log-mel front end), the RoBERTa text
tower and the projection heads that `tools/losses.py:259-316` (`CLAPLoss`) reads through
`laion_clap.CLAP_Module(enable_fusion=False, amodel='HTSAT-base')` (`laion_clap/hook.py:21-217`,
`clap_module/model.py:420-744`, `clap_module/htsat.py:615-1013`).

The towers are FROZEN in the reference (`self.clap.requires_grad_(False)`), so only INPUT gradients exist: the loss
back-propagates through the audio tower into the generated waveform (and from there through the vocoder and the VAE
decoder into the student U-Net).  Every contraction runs on `ctta_conv_gemm`, window attention on the flash kernels with
a full bias table (`ctta_attention_fullbias[_bwd]`), LayerNorm / GELU / resampling / log-mel / image folding on their
own HIP kernels, all through the C ABI.  The host side is torch plumbing: parameters (reference key names), bf16
activation buffers, `torch.autograd.Function`s that pair each forward kernel with its backward kernel (autograd only
sequences them and sums the two gradients that meet at each residual fork), and the index tables of the window
partition / cyclic shift / patch merging, which are row permutations of the token matrix.
"""
import ctypes
import math
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from . import _native as N
from . import spec
from .modules import _ParamTree

LOG2E = 1.4426950408889634


def _rup(a, b):
    return (a + b - 1) // b * b


def _check_cuda(t, what):
    if not t.is_cuda:
        raise N.CttaError("%s is on %s: the HIP kernels have no CPU path" % (what, t.device))


# ------------------------------------------------------------------------------------------------ low-level wrappers
def _desc(**kw):
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


def _conv(d):
    N.check(N.lib().ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))


class PackedLinear:
    """bf16 operands of y = x W^T (+ b): the forward pack [n_pad][k_pad] and its transpose for the data gradient.
    `row_map[r]` / `col_map[k]` give the source row / column of packed row r / packed column k (-1 = zero padding),
    which is how heads are padded from their real width to the attention kernels' 64 lanes."""

    def __init__(self, weight, bias, row_map, col_map, need_grad=True):
        dev = weight.device
        n_src, k_src = weight.shape
        n_pad, k_pad = len(row_map), _rup(len(col_map), 64)
        rm = torch.as_tensor(row_map, dtype=torch.int64)
        cm = torch.full((k_pad,), -1, dtype=torch.int64)
        cm[:len(col_map)] = torch.as_tensor(col_map, dtype=torch.int64)
        self.n, self.k_pad = n_pad, k_pad
        w32 = weight.detach().to(torch.float32).contiguous()
        ro = torch.where(rm >= 0, rm * k_src, torch.full_like(rm, -1)).to(torch.int32).to(dev)
        co = cm.to(torch.int32).to(dev)
        self.w = torch.empty(n_pad, k_pad, dtype=torch.bfloat16, device=dev)
        L_ = N.lib()
        N.check(L_.ctta_pack_weight(N.ptr(w32), N.ptr(ro), N.ptr(co), None, None, 0, n_pad, k_pad, N.ptr(self.w), N.stream_ptr()))
        self.bias = None
        if bias is not None:
            b = torch.zeros(n_pad, dtype=torch.float32, device=dev)
            ok = rm >= 0
            b[ok.to(dev)] = bias.detach().to(torch.float32)[rm[ok].to(dev)]
            self.bias = b
        self.wt = None
        if need_grad:   # W^T: rows = packed input columns (k_pad), K = packed output rows padded to 64
            self.nt_pad = _rup(n_pad, 64)
            rot = torch.where(cm >= 0, cm, torch.full_like(cm, -1)).to(torch.int32).to(dev)
            cmt = torch.full((self.nt_pad,), -1, dtype=torch.int64)
            cmt[:n_pad] = torch.where(rm >= 0, rm * k_src, torch.full_like(rm, -1))
            self.wt = torch.empty(k_pad, self.nt_pad, dtype=torch.bfloat16, device=dev)
            N.check(L_.ctta_pack_weight(N.ptr(w32), N.ptr(rot), N.ptr(cmt.to(torch.int32).to(dev)), None, None, 0, k_pad,
                                        self.nt_pad, N.ptr(self.wt), N.stream_ptr()))
        torch.cuda.current_stream().synchronize()   # the int32 maps may be freed once the packs have run


def _linear_fwd(x, P, res=None, out_act=0):
    """x bf16 [M][k_pad] -> bf16 [M][n]  (bias, optional residual of the same shape, optional tanh)."""
    M = x.shape[0]
    assert x.shape[1] == P.k_pad and x.is_contiguous(), (tuple(x.shape), P.k_pad)
    y = torch.empty(M, P.n, dtype=torch.bfloat16, device=x.device)
    d = _desc(x0=x, c0=P.k_pad, batch=1, hi=M, wi=1, ho=M, wo=1, w=P.w, k_pad=P.k_pad, n=P.n, bias=P.bias,
              out=y, ldc=P.n, out_act=out_act)
    if res is not None:
        d.res, d.res_ld = res.data_ptr(), res.shape[1]
    _conv(d)
    return y


def _linear_bwd(dy, P, out_f32=False):
    """dy bf16 [M][n] -> dx [M][k_pad] = dy W."""
    M = dy.shape[0]
    dyp = dy
    if dy.shape[1] != P.nt_pad:   # K of the transposed operand is padded to 64
        dyp = torch.zeros(M, P.nt_pad, dtype=torch.bfloat16, device=dy.device)
        dyp[:, :dy.shape[1]] = dy
    dx = torch.empty(M, P.k_pad, dtype=torch.float32 if out_f32 else torch.bfloat16, device=dy.device)
    _conv(_desc(x0=dyp.contiguous(), c0=P.nt_pad, batch=1, hi=M, wi=1, ho=M, wo=1, w=P.wt, k_pad=P.nt_pad, n=P.k_pad,
                out=dx, ldc=P.k_pad, out_f32=1 if out_f32 else 0))
    return dx


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, P):
        ctx.P, ctx.has_res = P, res is not None
        return _linear_fwd(x.contiguous(), P, res.contiguous() if res is not None else None)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return _linear_bwd(dy, ctx.P), (dy if ctx.has_res else None), None


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, d, eps):
        x = x.contiguous()
        y = torch.empty_like(x)
        N.check(N.lib().ctta_layernorm(N.ptr(x), N.ptr(y), x.shape[0], d, x.shape[1], N.ptr(gamma), N.ptr(beta), float(eps),
                                       N.stream_ptr()))
        ctx.save_for_backward(x, gamma)
        ctx.d, ctx.eps = d, eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        scratch = torch.zeros(2, x.shape[1], dtype=torch.float32, device=x.device)   # d gamma / d beta of a frozen layer
        nf = int(N.lib().ctta_layernorm_bwd_scratch_floats(x.shape[0], x.shape[1]))   # the caller owns the partial table
        part = torch.empty(nf, dtype=torch.float32, device=x.device) if nf else None
        N.check(N.lib().ctta_layernorm_bwd_ws(N.ptr(x), N.ptr(dy), None, N.ptr(dx), x.shape[0], ctx.d, x.shape[1], N.ptr(gamma),
                                              float(ctx.eps), N.ptr(scratch[0]), N.ptr(scratch[1]), N.ptr(part), nf,
                                              N.stream_ptr()))
        return dx, None, None, None, None


def _gather(x, idx, row_elems=None):
    x = x.contiguous()
    n = idx.numel()
    re = row_elems or x.shape[1]
    src = x.view(-1, re)
    out = torch.empty(n, re, dtype=torch.bfloat16, device=x.device)
    N.check(N.lib().ctta_gather_rows(N.ptr(src), re, N.ptr(idx), N.ptr(out), re, n, re, N.stream_ptr()))
    return out


class _Permute(torch.autograd.Function):
    """rows of a token matrix through a permutation (`idx`) -- the backward is the gather through its inverse."""

    @staticmethod
    def forward(ctx, x, idx, inv, row_elems, out_cols):
        ctx.inv, ctx.row_elems, ctx.in_shape = inv, row_elems, tuple(x.shape)
        return _gather(x, idx, row_elems).view(-1, out_cols)

    @staticmethod
    def backward(ctx, dy):
        return _gather(dy, ctx.inv, ctx.row_elems).view(ctx.in_shape), None, None, None, None


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        N.check(N.lib().ctta_gelu(N.ptr(x), N.ptr(y), x.numel(), N.stream_ptr()))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        N.check(N.lib().ctta_gelu_bwd(N.ptr(x), N.ptr(dy.contiguous()), N.ptr(dx), x.numel(), N.stream_ptr()))
        return dx


class _WindowAttention(torch.autograd.Function):
    """softmax(q k^T * scale + bias) v per (window, head) on head-padded operands.  qkv bf16 [windows * n][3 * hp]
    (q | k | v blocks of heads * 64 columns); bias_log2 fp32 [nbb][heads][n][n]; returns [windows * n][hp]."""

    @staticmethod
    def forward(ctx, qkv, bias_log2, nbb, heads, n, scale):
        qkv = qkv.contiguous()
        hp = heads * 64
        T = qkv.shape[0]
        nw = T // n
        dev = qkv.device
        L_ = N.lib()
        vt_ld = max(8, _rup(n, 8))
        vt = torch.empty(nw, hp, vt_ld, dtype=torch.bfloat16, device=dev)
        N.check(L_.ctta_transpose_bf16(N.ptr(qkv), n * 3 * hp, n, hp, 3 * hp, 2 * hp, N.ptr(vt), hp * vt_ld, vt_ld, nw,
                                       N.stream_ptr()))
        out = torch.empty(T, hp, dtype=torch.bfloat16, device=dev)
        need = qkv.requires_grad
        lse = torch.empty(nw, heads, n, dtype=torch.float32, device=dev)
        kptr = N.c_void_p(qkv.data_ptr() + hp * 2)
        N.check(L_.ctta_attention_fullbias(N.ptr(qkv), 3 * hp, kptr, 3 * hp, n, N.ptr(vt), vt_ld, N.ptr(bias_log2), nbb,
                                           N.ptr(out), hp, nw, heads, n, n, float(scale), N.ptr(lse), N.stream_ptr()))
        ctx.save_for_backward(qkv, out, lse, bias_log2)
        ctx.cfg = (nbb, heads, n, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, bias_log2 = ctx.saved_tensors
        nbb, heads, n, scale = ctx.cfg
        hp = heads * 64
        T = qkv.shape[0]
        nw = T // n
        dev = qkv.device
        L_ = N.lib()
        dout = dout.contiguous()
        s = N.stream_ptr()

        def tpose(src, col0, ld):
            dst = torch.empty(nw, hp, 64, dtype=torch.bfloat16, device=dev)
            N.check(L_.ctta_transpose_bf16(N.ptr(src), n * ld, n, hp, ld, col0, N.ptr(dst), hp * 64, 64, nw, s))
            return dst
        qt, kt, dot = tpose(qkv, 0, 3 * hp), tpose(qkv, hp, 3 * hp), tpose(dout, 0, hp)
        dqkv = torch.empty_like(qkv)
        dsum = torch.empty(nw, heads, n, dtype=torch.float32, device=dev)
        base = qkv.data_ptr()
        dbase = dqkv.data_ptr()
        N.check(L_.ctta_attention_fullbias_bwd(
            N.c_void_p(base), 3 * hp, N.c_void_p(base + hp * 2), 3 * hp, n, N.c_void_p(base + 4 * hp), 3 * hp, n,
            N.ptr(kt), 64, N.ptr(qt), N.ptr(dot), 64, N.ptr(bias_log2), nbb, N.ptr(out), hp, N.ptr(dout), hp, N.ptr(lse),
            N.ptr(dsum), N.c_void_p(dbase), 3 * hp, N.c_void_p(dbase + hp * 2), 3 * hp, N.c_void_p(dbase + 4 * hp), 3 * hp,
            nw, heads, n, n, float(scale), s))
        return dqkv, None, None, None, None, None


class _MeanTokens(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, batch, tokens, channels):
        x = x.contiguous()
        y = torch.empty(batch, channels, dtype=torch.float32, device=x.device)
        N.check(N.lib().ctta_mean_tokens(N.ptr(x), batch, tokens, channels, x.shape[1], N.ptr(y), N.stream_ptr()))
        ctx.cfg = (batch, tokens, channels, x.shape[1])
        return y

    @staticmethod
    def backward(ctx, dy):
        batch, tokens, channels, ld = ctx.cfg
        dx = torch.zeros(batch * tokens, ld, dtype=torch.bfloat16, device=dy.device)
        N.check(N.lib().ctta_mean_tokens_bwd(N.ptr(dy.contiguous().float()), batch, tokens, channels, ld, N.ptr(dx),
                                             N.stream_ptr()))
        return dx, None, None, None


class _LinearF32(torch.autograd.Function):
    """fp32 y = x W^T + b on the small projection heads (rows = batch)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous().float()
        y = torch.empty(x.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
        N.check(N.lib().ctta_linear_f32(N.ptr(x), N.ptr(w), N.ptr(b), N.ptr(y), x.shape[0], w.shape[0], w.shape[1], 0, 0,
                                        N.stream_ptr()))
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous().float()
        dx = torch.empty_like(x)
        N.check(N.lib().ctta_linear_f32_bwd(N.ptr(x), N.ptr(w), N.ptr(dy), dy.shape[1], None, N.ptr(dx), None, None,
                                            x.shape[0], w.shape[0], w.shape[1], 0, 0, N.stream_ptr()))
        return dx, None, None


# ------------------------------------------------------------------------------------------------ resampler
def sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99, beta=None):
    """Host table of torchaudio's `_get_sinc_resample_kernel(..., "sinc_interp_kaiser")` (float64 arithmetic, float32
    result): returns (kernels [new][taps], width, orig, new) for the gcd-reduced rates."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = np.clip(t * base, -lowpass_filter_width, lowpass_filter_width)
    beta = 14.769656459379492 if beta is None else float(beta)
    window = np.i0(beta * np.sqrt(1 - (t / lowpass_filter_width) ** 2)) / np.i0(beta)
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    return (k * window * (base / orig)).astype(np.float32), width, orig, new


class _Resample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wav, kernels, up, down, width):
        wav = wav.contiguous().float()
        B, L = wav.shape
        out_len = int(math.ceil(up * L / down))
        y = torch.empty(B, out_len, dtype=torch.float32, device=wav.device)
        N.check(N.lib().ctta_resample_poly(N.ptr(wav), B, L, N.ptr(kernels), up, down, width, kernels.shape[1], N.ptr(y),
                                           out_len, N.stream_ptr()))
        ctx.save_for_backward(kernels)
        ctx.cfg = (B, L, up, down, width, out_len)
        return y

    @staticmethod
    def backward(ctx, dy):
        (kernels,) = ctx.saved_tensors
        B, L, up, down, width, out_len = ctx.cfg
        dx = torch.empty(B, L, dtype=torch.float32, device=dy.device)
        N.check(N.lib().ctta_resample_poly_bwd(N.ptr(dy.contiguous().float()), B, out_len, N.ptr(kernels), up, down, width,
                                               kernels.shape[1], N.ptr(dx), L, N.stream_ptr()))
        return dx, None, None, None, None


class Resampler(nn.Module):
    """`torchaudio.functional.resample(wav, orig, new, lowpass_filter_width, rolloff, "sinc_interp_kaiser", beta)` as a
    polyphase FIR on the GPU (tools/losses.py:299-303), differentiable with respect to the waveform."""

    def __init__(self, orig_freq=16000, new_freq=48000, lowpass_filter_width=64, rolloff=0.9475937167399596,
                 beta=14.769656459379492):
        super().__init__()
        k, self.width, self.down, self.up = sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width, rolloff, beta)
        self.register_buffer("kernels", torch.from_numpy(k), persistent=False)

    def forward(self, wav):
        _check_cuda(wav, "waveform")
        if self.kernels.device != wav.device:
            self.kernels = self.kernels.to(wav.device)
        shape = wav.shape
        y = _Resample.apply(wav.reshape(-1, shape[-1]), self.kernels, self.up, self.down, self.width)
        return y.reshape(shape[:-1] + (y.shape[-1],))


# ------------------------------------------------------------------------------------------------ HTSAT front end
def bicubic_taps(t_in, t_out):
    """Source indices / weights [t_out][4] of F.interpolate(mode="bicubic", align_corners=True) along one axis (cubic
    convolution, A = -0.75, border indices clamped), evaluated in float32 like torch does for float32 inputs."""
    A = np.float32(-0.75)
    scale = np.float32(t_in - 1) / np.float32(t_out - 1) if t_out > 1 else np.float32(0)
    real = scale * np.arange(t_out, dtype=np.float32)
    base = np.floor(real)
    t = (real - base).astype(np.float32)

    def c1(x):
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def c2(x):
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    w = np.stack([c2(t + 1), c1(t), c1(1 - t), c2(2 - t)], 1).astype(np.float32)
    idx = np.clip(base[:, None].astype(np.int64) + np.arange(-1, 3)[None, :], 0, t_in - 1).astype(np.int32)
    return idx, w


class _Frontend(torch.autograd.Function):
    """waveform (B, L) fp32 at the tower's rate -> patch tokens bf16 [B * (S/ps)^2][C]: power STFT, dB log-mel, bn0,
    bicubic frame stretch, fold into the (S x S) image (htsat.py:911-925), then the patch-embedding convolution
    (ps x ps, stride ps, htsat.py:105-108,146-151) as an implicit GEMM over the NHWC image.  Backward: the conv's data
    gradient in token layout [B * patches][ps * ps], the adjoint of fold / stretch / bn0, log-mel and STFT backward."""

    @staticmethod
    def forward(ctx, wav, fe, P, Pt):
        wav = wav.contiguous().float()
        B, L = wav.shape
        grad = bool(ctx.needs_input_grad[0])
        h = fe.handle(B, L, grad)
        frames = L // fe.hop + 1
        S, F, ps = fe.spec_size, fe.mel_bins, fe.patch
        TT = S * (S // F)
        if frames > TT:
            raise ValueError("the waveform gives %d frames, more than the %d the tower's image holds" % (frames, TT))
        lm = torch.empty(B, frames, F, dtype=torch.float32, device=wav.device)
        L_ = N.lib()
        N.check(L_.ctta_wav_to_logmel_db(h, N.ptr(wav), B, L, 1e-10, N.ptr(lm), N.stream_ptr()))
        tabs = fe.tap_tables(frames, TT, wav.device)
        img = torch.empty(B, S, S, 8, dtype=torch.bfloat16, device=wav.device)
        N.check(L_.ctta_htsat_image(N.ptr(lm), B, frames, F, N.ptr(fe.bn_scale), N.ptr(fe.bn_shift), N.ptr(tabs["idx"]),
                                    N.ptr(tabs["w"]), TT, S, 8, N.ptr(img), N.stream_ptr()))
        g = S // ps
        y = torch.empty(B * g * g, P.n, dtype=torch.bfloat16, device=wav.device)
        _conv(_desc(x0=img, c0=8, batch=B, hi=S, wi=S, ho=g, wo=g, kh=ps, kw=ps, stride_h=ps, stride_w=ps, w=P.w,
                    k_pad=P.k_pad, n=P.n, bias=P.bias, out=y, ldc=P.n))
        ctx.fe, ctx.cfg, ctx.h, ctx.Pt = fe, (B, L, frames), h, Pt
        if grad:
            fe._pending = ctx.token = object()
        fe.last_image = img
        return y

    @staticmethod
    def backward(ctx, dy):
        fe, Pt = ctx.fe, ctx.Pt
        B, L, frames = ctx.cfg
        if fe._pending is not ctx.token:
            raise N.CttaError("the CLAP front end ran another differentiable pass before this backward: only the latest "
                              "one can be back-propagated (its STFT lives in the handle)")
        dy = dy.contiguous()
        M = dy.shape[0]
        dtok = torch.empty(M, Pt.n, dtype=torch.float32, device=dy.device)
        _conv(_desc(x0=dy, c0=Pt.k_pad, batch=1, hi=M, wi=1, ho=M, wo=1, w=Pt.w, k_pad=Pt.k_pad, n=Pt.n, out=dtok,
                    ldc=Pt.n, out_f32=1))
        S, F = fe.spec_size, fe.mel_bins
        tabs = fe.tap_tables(frames, S * (S // F), dy.device)
        dlm = torch.empty(B, frames, F, dtype=torch.float32, device=dy.device)
        L_ = N.lib()
        N.check(L_.ctta_htsat_image_bwd(N.ptr(dtok), Pt.n, fe.patch, B, frames, F, N.ptr(fe.bn_scale),
                                        N.ptr(tabs["rt_ptr"]), N.ptr(tabs["rt_tt"]), N.ptr(tabs["rt_w"]), S, N.ptr(dlm),
                                        N.stream_ptr()))
        dwav = torch.empty(B, L, dtype=torch.float32, device=dy.device)
        N.check(L_.ctta_wav_to_logmel_db_bwd(ctx.h, N.ptr(dlm), B, L, 1e-10, N.ptr(dwav), N.stream_ptr()))
        fe._pending = None
        return dwav, None, None, None


class _FrontendState:
    """Per-tower front-end state: two ctta_mel_frontend handles (the differentiable pass keeps its STFT in the handle
    until the backward, so a plain pass in between must not share it), bn0 folded to scale / shift, tap tables."""

    def __init__(self, cfg):
        self.n_fft, self.hop, self.mel_bins = cfg["n_fft"], cfg["hop"], cfg["mel_bins"]
        self.sr, self.fmin, self.fmax = cfg["sample_rate"], cfg["fmin"], cfg["fmax"]
        self.spec_size, self.patch = cfg["spec_size"], cfg["patch_size"]
        self._h = {False: None, True: None}
        self._key = {False: None, True: None}
        self._tabs = {}
        self._pending = None
        self.last_image = None
        self.bn_scale = self.bn_shift = None

    def handle(self, B, L, grad):
        key = self._key[grad]
        if self._h[grad] is None or B > key[0] or L > key[1]:
            self.release(grad)
            Bm, Lm = (max(B, key[0]), max(L, key[1])) if key else (B, L)
            h = N.c_void_p()
            N.check(N.lib().ctta_mel_frontend_create(self.n_fft, self.hop, self.n_fft, self.mel_bins, self.sr,
                                                     float(self.fmin), float(self.fmax), Bm, Lm, h))
            self._h[grad], self._key[grad] = h, (Bm, Lm)
        return self._h[grad]

    def release(self, grad=None):
        for g in ((False, True) if grad is None else (grad,)):
            if self._h[g] is not None:
                N.lib().ctta_mel_frontend_destroy(self._h[g])
                self._h[g] = None

    def tap_tables(self, t_in, t_out, dev):
        key = (t_in, t_out, str(dev))
        if key not in self._tabs:
            if t_in < t_out:
                idx, w = bicubic_taps(t_in, t_out)
            else:   # no stretch: identity taps
                idx = np.repeat(np.arange(t_out, dtype=np.int32)[:, None], 4, 1)
                w = np.tile(np.array([0, 1, 0, 0], np.float32), (t_out, 1))
            rows = [[] for _ in range(t_in)]
            for tt in range(t_out):
                for k in range(4):
                    rows[int(idx[tt, k])].append((tt, float(w[tt, k])))
            ptr = np.zeros(t_in + 1, np.int32)
            for t in range(t_in):
                ptr[t + 1] = ptr[t] + len(rows[t])
            rt_tt = np.array([e[0] for r in rows for e in r], np.int32)
            rt_w = np.array([e[1] for r in rows for e in r], np.float32)
            self._tabs[key] = {k: torch.from_numpy(v).to(dev) for k, v in
                               dict(idx=idx.reshape(-1), w=w.reshape(-1), rt_ptr=ptr, rt_tt=rt_tt, rt_w=rt_w).items()}
        return self._tabs[key]


# ------------------------------------------------------------------------------------------------ HTSAT tower
def _relative_position_index(ws):
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0) + (ws - 1)
    return rel[:, :, 0] * (2 * ws - 1) + rel[:, :, 1]


def _shift_mask(H, W, ws, shift):
    img = torch.zeros(H, W)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = img.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    d = mw[:, None, :] - mw[:, :, None]
    return torch.where(d != 0, torch.tensor(-100.0), torch.tensor(0.0))


def _window_perm(B, res, ws, shift):
    """Row r of the window-partitioned (and cyclically shifted) token matrix <- token index (htsat.py:471-482)."""
    y = torch.arange(res)
    ys = (y + shift) % res                        # shifted[y] = x[(y + shift) % res]  (torch.roll by -shift)
    grid = (ys[:, None] * res + ys[None, :])      # source token of shifted position (y, x)
    nw = res // ws
    win = grid.view(nw, ws, nw, ws).permute(0, 2, 1, 3).reshape(-1)
    idx = (torch.arange(B)[:, None] * res * res + win[None, :]).reshape(-1)
    inv = torch.empty_like(idx)
    inv[idx] = torch.arange(idx.numel())
    return idx.to(torch.int32), inv.to(torch.int32)


def _merge_perm(B, res):
    """PatchMerging's 2x2 concat (htsat.py:526-532) as a row gather: merged row (b, y2, x2), slot j <- token."""
    y2, x2 = torch.meshgrid(torch.arange(res // 2), torch.arange(res // 2), indexing="ij")
    src = torch.stack([(2 * y2) * res + 2 * x2, (2 * y2 + 1) * res + 2 * x2, (2 * y2) * res + 2 * x2 + 1,
                       (2 * y2 + 1) * res + 2 * x2 + 1], -1).reshape(-1)
    idx = (torch.arange(B)[:, None] * res * res + src[None, :]).reshape(-1)
    inv = torch.empty_like(idx)
    inv[idx] = torch.arange(idx.numel())
    return idx.to(torch.int32), inv.to(torch.int32)


class HTSAT(_ParamTree):
    """`HTSAT_Swin_Transformer` (htsat.py:615-1013), the path `encode_audio(...)["embedding"]` takes in eval mode without
    fusion: parameters under the reference's key names, forward = waveform -> (B, num_features) embedding."""

    def __init__(self, config=None):
        super().__init__()
        cfg = dict(spec.HTSAT_BASE_CONFIG)
        cfg.update(config or {})
        if cfg["embed_dim"] % 64:
            raise ValueError("embed_dim=%d: the engine keeps token rows in 64-column blocks" % cfg["embed_dim"])
        if cfg["patch_size"] != cfg["patch_stride"]:
            raise ValueError("only non-overlapping patches (patch_size == patch_stride, every released HTSAT) are built")
        for i, h in enumerate(cfg["num_heads"]):
            if cfg["embed_dim"] * 2 ** i // h > 64:
                raise ValueError("head width above 64 lanes is not supported by the attention kernels")
        self.cfg = cfg
        self._register(spec.htsat_param_spec(cfg))
        self.requires_grad_(False)
        self._packed = None
        self._packed_ver = None
        self._fe = _FrontendState(cfg)
        self._perm_cache = {}

    @property
    def device(self):
        return self.get_parameter("norm.weight").device

    def load_state_dict(self, state_dict, strict=True):
        """Reference checkpoints also hold the integer buffers (relative_position_index, num_batches_tracked) and the
        shift masks, which are structural here."""
        keep = OrderedDict((k, v) for k, v in state_dict.items()
                           if not k.endswith(("relative_position_index", "num_batches_tracked", "attn_mask")))
        return super().load_state_dict(keep, strict=strict)

    def __del__(self):
        try:
            self._fe.release()
        except Exception:
            pass

    # ---- operands
    def _pack(self):
        ver = self._weights_version()
        if self._packed is not None and self._packed_ver == ver:
            return self._packed
        cfg = self.cfg
        sd = {k: p.detach() for k, p in self.named_parameters()}
        for k, p in sd.items():
            _check_cuda(p, "parameter '%s'" % k)
        dev = self.device
        P = {}
        C, ps = cfg["embed_dim"], cfg["patch_size"]
        # patch embedding: (C, 1, ps, ps) conv over the NHWC image with the channel padded to 8 -> K = ps*ps*8
        colmap = [-1] * (ps * ps * 8)
        for t in range(ps * ps):
            colmap[t * 8] = t
        P["patch"] = PackedLinear(sd["patch_embed.proj.weight"].reshape(C, ps * ps), sd["patch_embed.proj.bias"],
                                  list(range(C)), colmap, need_grad=False)
        # its data gradient in token layout: d tok[16] = d y[C] . W  -> rows = taps, K = C
        P["patch_t"] = PackedLinear(sd["patch_embed.proj.weight"].reshape(C, ps * ps).t().contiguous(), None,
                                    list(range(ps * ps)), list(range(C)), need_grad=False)
        res = cfg["spec_size"] // cfg["patch_stride"]
        n_layers = len(cfg["depths"])
        for i, (depth, heads) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
            dim = C * 2 ** i
            hd = dim // heads
            ws = min(cfg["window_size"], res)
            hp = heads * 64
            pad_rows = []      # packed row -> source row of [q | k | v], head h at columns h*64 .. h*64 + hd
            for part in range(3):
                for h in range(heads):
                    pad_rows += [part * dim + h * hd + e for e in range(hd)] + [-1] * (64 - hd)
            pad_cols = []
            for h in range(heads):
                pad_cols += [h * hd + e for e in range(hd)] + [-1] * (64 - hd)
            rpi = _relative_position_index(ws).view(-1)
            for j in range(depth):
                p = "layers.%d.blocks.%d." % (i, j)
                shift = 0 if (j % 2 == 0 or res <= cfg["window_size"]) else cfg["window_size"] // 2
                n = ws * ws
                bias = sd[p + "attn.relative_position_bias_table"].float().cpu()[rpi].view(n, n, heads).permute(2, 0, 1)
                if shift:
                    bias = bias[None] + _shift_mask(res, res, ws, shift)[:, None]
                else:
                    bias = bias[None]
                P[p] = dict(
                    qkv=PackedLinear(sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"], pad_rows, list(range(dim))),
                    proj=PackedLinear(sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], list(range(dim)), pad_cols),
                    fc1=PackedLinear(sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], list(range(int(dim * cfg["mlp_ratio"]))),
                                     list(range(dim))),
                    fc2=PackedLinear(sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"], list(range(dim)),
                                     list(range(int(dim * cfg["mlp_ratio"])))),
                    bias=(bias * LOG2E).contiguous().to(dev), nbb=bias.shape[0], heads=heads, ws=ws, shift=shift, res=res,
                    scale=hd ** -0.5, dim=dim)
            if i < n_layers - 1:
                p = "layers.%d.downsample." % i
                P[p] = PackedLinear(sd[p + "reduction.weight"], None, list(range(2 * dim)), list(range(4 * dim)))
                res //= 2
        self._fe.bn_scale = (sd["bn0.weight"].float() / torch.sqrt(sd["bn0.running_var"].float() + 1e-5)).contiguous()
        self._fe.bn_shift = (sd["bn0.bias"].float() - sd["bn0.running_mean"].float() * self._fe.bn_scale).contiguous()
        self._packed, self._packed_ver = P, ver
        return P

    def _perm(self, kind, B, res, ws=0, shift=0):
        key = (kind, B, res, ws, shift, str(self.device))
        if key not in self._perm_cache:
            idx, inv = _window_perm(B, res, ws, shift) if kind == "win" else _merge_perm(B, res)
            self._perm_cache[key] = (idx.to(self.device), inv.to(self.device))
        return self._perm_cache[key]

    # ---- forward
    def forward(self, wav, taps=None):
        """wav (B, L) fp32 at cfg['sample_rate'] -> embedding (B, num_features) fp32; differentiable in `wav`."""
        _check_cuda(wav, "waveform")
        P = self._pack()
        cfg = self.cfg
        sd = dict(self.named_parameters())
        B = wav.shape[0]
        S, ps, C = cfg["spec_size"], cfg["patch_size"], cfg["embed_dim"]
        x = _Frontend.apply(wav, self._fe, P["patch"], P["patch_t"])             # tokens [B * (S/ps)^2][C] bf16
        if taps is not None:
            taps["image"] = self._fe.last_image[..., 0].float()
        x = _LayerNorm.apply(x, sd["patch_embed.norm.weight"], sd["patch_embed.norm.bias"], C, 1e-5)
        res = S // cfg["patch_stride"]
        n_layers = len(cfg["depths"])
        for i, depth in enumerate(cfg["depths"]):
            dim = C * 2 ** i
            for j in range(depth):
                p = "layers.%d.blocks.%d." % (i, j)
                L_ = P[p]
                idx, inv = self._perm("win", B, res, L_["ws"], L_["shift"])
                h = _LayerNorm.apply(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], dim, 1e-5)
                hw = _Permute.apply(h, idx, inv, dim, dim)
                qkv = _Linear.apply(hw, None, L_["qkv"])
                a = _WindowAttention.apply(qkv, L_["bias"], L_["nbb"], L_["heads"], L_["ws"] ** 2, L_["scale"])
                a = _Permute.apply(a, inv, idx, a.shape[1], a.shape[1])
                x = _Linear.apply(a, x, L_["proj"])
                h = _LayerNorm.apply(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], dim, 1e-5)
                f = _Gelu.apply(_Linear.apply(h, None, L_["fc1"]))
                x = _Linear.apply(f, x, L_["fc2"])
            if taps is not None:
                taps["layer%d" % i] = x.detach().float().view(B, -1, dim)
            if i < n_layers - 1:
                p = "layers.%d.downsample." % i
                idx, inv = self._perm("merge", B, res)
                m = _Permute.apply(x, idx, inv, dim, 4 * dim)
                m = _LayerNorm.apply(m, sd[p + "norm.weight"], sd[p + "norm.bias"], 4 * dim, 1e-5)
                x = _Linear.apply(m, None, P[p])
                res //= 2
        nf = C * 2 ** (n_layers - 1)
        x = _LayerNorm.apply(x, sd["norm.weight"], sd["norm.bias"], nf, 1e-5)
        return _MeanTokens.apply(x, B, res * res, nf)


# ------------------------------------------------------------------------------------------------ RoBERTa text tower
class RobertaModel(_ParamTree):
    """`transformers.RobertaModel` as the reference's text branch uses it (`text_branch(input_ids=, attention_mask=)
    ["pooler_output"]`, model.py:633-641): same parameter names, forward only (frozen, no gradient reaches the text)."""

    def __init__(self, config=None):
        super().__init__()
        cfg = dict(spec.ROBERTA_BASE_CONFIG)
        cfg.update(config or {})
        if cfg["hidden_size"] // cfg["num_attention_heads"] != 64:
            raise ValueError("head width must be 64 (every released RoBERTa)")
        self.cfg = cfg
        self._register(spec.roberta_param_spec(cfg))
        self.requires_grad_(False)
        self._packed = self._packed_ver = None

    @property
    def device(self):
        return self.get_parameter("pooler.dense.weight").device

    def load_state_dict(self, state_dict, strict=True):
        keep = OrderedDict((k, v) for k, v in state_dict.items() if not k.endswith(("position_ids", "token_type_ids")))
        return super().load_state_dict(keep, strict=strict)

    def _pack(self):
        ver = self._weights_version()
        if self._packed is not None and self._packed_ver == ver:
            return self._packed
        sd = {k: p.detach() for k, p in self.named_parameters()}
        H, I = self.cfg["hidden_size"], self.cfg["intermediate_size"]
        P = {}
        for i in range(self.cfg["num_hidden_layers"]):
            p = "encoder.layer.%d." % i
            wqkv = torch.cat([sd[p + "attention.self.%s.weight" % t] for t in ("query", "key", "value")])
            bqkv = torch.cat([sd[p + "attention.self.%s.bias" % t] for t in ("query", "key", "value")])
            P[p] = dict(qkv=PackedLinear(wqkv, bqkv, list(range(3 * H)), list(range(H)), need_grad=False),
                        out=PackedLinear(sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"],
                                         list(range(H)), list(range(H)), need_grad=False),
                        fc1=PackedLinear(sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"],
                                         list(range(I)), list(range(H)), need_grad=False),
                        fc2=PackedLinear(sd[p + "output.dense.weight"], sd[p + "output.dense.bias"], list(range(H)),
                                         list(range(I)), need_grad=False))
        P["pooler"] = PackedLinear(sd["pooler.dense.weight"], sd["pooler.dense.bias"], list(range(H)), list(range(H)),
                                   need_grad=False)
        self._packed, self._packed_ver = P, ver
        return P

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, **kwargs):
        _check_cuda(input_ids, "input_ids")
        cfg = self.cfg
        B, L = input_ids.shape
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        P = self._pack()
        sd = dict(self.named_parameters())
        H, heads, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
        pad = cfg["pad_token_id"]
        # create_position_ids_from_input_ids: cumulative count of non-pad tokens, offset by the padding index
        keep = (input_ids != pad).to(torch.int64)
        pos = torch.cumsum(keep, 1) * keep + pad
        emb = (sd["embeddings.word_embeddings.weight"][input_ids] + sd["embeddings.position_embeddings.weight"][pos]
               + sd["embeddings.token_type_embeddings.weight"][0])
        x = emb.reshape(B * L, H).to(torch.bfloat16).contiguous()
        x = _LayerNorm.apply(x, sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], H, eps)
        key_bias = ((1.0 - attention_mask.to(torch.float32)) * -30000.0).contiguous().to(x.device)   # exp underflows to 0
        L_ = N.lib()
        vt_ld = max(8, _rup(L, 8))
        for i in range(cfg["num_hidden_layers"]):
            p = "encoder.layer.%d." % i
            qkv = _linear_fwd(x, P[p]["qkv"])
            vt = torch.empty(B, H, vt_ld, dtype=torch.bfloat16, device=x.device)
            N.check(L_.ctta_transpose_bf16(N.ptr(qkv), L * 3 * H, L, H, 3 * H, 2 * H, N.ptr(vt), H * vt_ld, vt_ld, B,
                                           N.stream_ptr()))
            a = torch.empty(B * L, H, dtype=torch.bfloat16, device=x.device)
            N.check(L_.ctta_attention(N.ptr(qkv), 3 * H, N.c_void_p(qkv.data_ptr() + 2 * H), 3 * H, L, N.ptr(vt), vt_ld,
                                      N.ptr(key_bias), N.ptr(a), H, B, heads, L, L, 0.125, N.stream_ptr()))
            h = _linear_fwd(a, P[p]["out"], res=x)
            x = _LayerNorm.apply(h, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], H, eps)
            f = _Gelu.apply(_linear_fwd(x, P[p]["fc1"]))
            h = _linear_fwd(f, P[p]["fc2"], res=x)
            x = _LayerNorm.apply(h, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], H, eps)
        last = x.view(B, L, H)
        pooled = _linear_fwd(last[:, 0].contiguous(), P["pooler"], out_act=2).float()      # tanh epilogue
        return {"last_hidden_state": last.float(), "pooler_output": pooled}


# ------------------------------------------------------------------------------------------------ CLAP
class CLAP(nn.Module):
    """`clap_module.model.CLAP` reduced to the two embedding paths (model.py:688-744)."""

    def __init__(self, audio_cfg=None, text_cfg=None, joint=spec.CLAP_JOINT_DIM):
        super().__init__()
        self.audio_branch = HTSAT(audio_cfg)
        self.text_branch = RobertaModel(text_cfg)
        nf = self.audio_branch.cfg["embed_dim"] * 2 ** (len(self.audio_branch.cfg["depths"]) - 1)

        def mlp(width):
            seq = nn.Sequential(nn.Linear(width, joint), nn.ReLU(), nn.Linear(joint, joint))
            return seq
        self.text_projection = mlp(self.text_branch.cfg["hidden_size"])
        self.audio_projection = mlp(nf)
        self.requires_grad_(False)

    def init_random_(self, seed=0):
        """On-device random init for benchmarks (no CLAP checkpoint offline): the same scale rules as the deterministic
        test weights (spec.clap_det_weight), values not reproducible across boxes."""
        gen = torch.Generator(device=self.audio_branch.device)
        gen.manual_seed(seed)
        with torch.no_grad():
            for k, p in self.named_parameters():
                u = torch.rand(p.shape, generator=gen, device=p.device) * 2 - 1
                if k.endswith("running_var"):
                    p.copy_(1.0 + 0.5 * u.abs())
                elif k.endswith("relative_position_bias_table"):
                    p.copy_(0.5 * u)
                elif k.endswith("bn0.weight"):
                    p.copy_(0.1 + 0.02 * u)
                elif k.endswith("LayerNorm.weight"):
                    p.copy_(1.0 + 0.2 * u)
                elif "embeddings." in k and k.endswith("weight"):
                    p.copy_(u)
                else:
                    off, amp = spec.weight_rule(k, tuple(p.shape))
                    p.copy_(off + amp * u)
        return self

    @staticmethod
    def _project(seq, x):
        h = _LinearF32.apply(x, seq[0].weight, seq[0].bias)
        h = torch.clamp_min(h, 0.0)
        return _LinearF32.apply(h, seq[2].weight, seq[2].bias)

    def get_audio_embedding(self, wav48k):
        e = self._project(self.audio_projection, self.audio_branch(wav48k))
        return torch.nn.functional.normalize(e, dim=-1)

    @torch.no_grad()
    def get_text_embedding(self, data):
        out = self.text_branch(input_ids=data["input_ids"], attention_mask=data["attention_mask"])["pooler_output"]
        return torch.nn.functional.normalize(self._project(self.text_projection, out), dim=-1)

    def load_state_dict(self, state_dict, strict=False):
        """Released checkpoints also carry logit scales, the *_transform MLPs and the classification head, which the
        embedding paths never read."""
        own = self.state_dict()
        keep = OrderedDict((k, v) for k, v in state_dict.items() if k in own)
        self.audio_branch.load_state_dict(OrderedDict((k[len("audio_branch."):], v) for k, v in state_dict.items()
                                                      if k.startswith("audio_branch.")), strict=False)
        rest = OrderedDict((k, v) for k, v in keep.items() if not k.startswith("audio_branch."))
        return super().load_state_dict(rest, strict=False)


class CLAP_Module(nn.Module):
    """`laion_clap.CLAP_Module(enable_fusion=False, amodel='HTSAT-base')` (hook.py:21-217): `load_ckpt`,
    `get_audio_embedding_from_data(x, use_tensor=True)`, `get_text_embedding(captions, use_tensor=True)`."""

    def __init__(self, enable_fusion=False, device=None, amodel="HTSAT-base", tmodel="roberta", audio_cfg=None,
                 text_cfg=None, tokenizer=None, clip_samples=480000):
        super().__init__()
        if enable_fusion:
            raise NotImplementedError("the fusion front end is not used by ConsistencyTTA (enable_fusion=False)")
        if amodel != "HTSAT-base" or tmodel != "roberta":
            raise NotImplementedError("only HTSAT-base + roberta (tools/losses.py:270) is built")
        self.enable_fusion = False
        self.model = CLAP(audio_cfg, text_cfg)
        self.clip_samples = int(clip_samples)      # max_len of get_audio_features (hook.py:178): 10 s at 48 kHz
        self.model_cfg = {"audio_cfg": dict(clip_samples=self.clip_samples, sample_rate=48000)}
        self.tokenize = tokenizer

    def load_ckpt(self, ckpt=None, model_id=-1, verbose=False):
        if ckpt is None:
            raise RuntimeError("downloading CLAP checkpoints needs network access; pass ckpt=<path>")
        sd = torch.load(ckpt, map_location="cpu", weights_only=False)   # a full training checkpoint dict (epoch, optimizer, ...), as laion_clap loads it
        sd = sd.get("state_dict", sd)
        sd = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in sd.items())
        sd.pop("text_branch.embeddings.position_ids", None)
        return self.model.load_state_dict(sd)

    def tokenizer(self, text):
        if self.tokenize is None:
            from transformers import RobertaTokenizer
            self.tokenize = RobertaTokenizer.from_pretrained("roberta-base")
        result = self.tokenize(text, padding="max_length", truncation=True, max_length=77, return_tensors="pt")
        return {k: v.squeeze(0) for k, v in result.items()}

    @staticmethod
    def _fit(wav, max_len=480000):
        """get_audio_features(data_truncating='rand_trunc', data_filling='repeatpad') (training/data.py:402-492) for a
        batch of equal-length clips: repeat-and-zero-pad short clips, randomly crop long ones."""
        L = wav.shape[-1]
        if L > max_len:   # one offset PER CLIP, like the reference's loop over waveforms (hook.py:174-186)
            flat = wav.reshape(-1, L)
            offs = [int(np.random.randint(0, L - max_len + 1)) for _ in range(flat.shape[0])]
            return torch.stack([flat[j, i:i + max_len] for j, i in enumerate(offs)]).reshape(*wav.shape[:-1], max_len)
        if L < max_len:
            rep = wav.repeat(1, max_len // L)
            return torch.nn.functional.pad(rep, (0, max_len - rep.shape[-1]))
        return wav

    def get_audio_embedding_from_data(self, x, use_tensor=False):
        if not use_tensor:
            raise NotImplementedError("numpy waveforms (with int16 quantisation) are not used on the training path")
        self.model.eval()
        return self.model.get_audio_embedding(self._fit(x, self.clip_samples))

    def get_text_embedding(self, x, tokenizer=None, use_tensor=False):
        self.model.eval()
        tok = tokenizer(x) if tokenizer is not None else self.tokenizer(x)
        dev = self.model.text_branch.device
        data = {k: (v if v.ndim == 2 else v[None]).to(dev) for k, v in tok.items()}
        emb = self.model.get_text_embedding(data)
        return emb if use_tensor else emb.detach().cpu().numpy()
