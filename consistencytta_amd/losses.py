"""Training losses of tools/losses.py on top of the HIP engines.

The latent MSE ('mse', train.sh's setting) never goes through this module: AudioLCM computes it and its gradient with
ctta_snr_mse_loss / ctta_snr_mse_grad.  The perceptual variants decode the predicted latent with
AutoencoderKL.decode_first_stage / decode_to_waveform(allow_grad=True) -- ctta_vae_decode_with_grad and
ctta_hifigan_forward_with_grad on the device -- and only the few reductions after that run as torch ops:

  MelLoss                  tools/losses.py:36-64    0.3 * mse(mel(input), mel(target)) + 0.7 * mse(input, target)
  MultiResolutionSTFTLoss  tools/losses.py:187-256  spectral convergence + log-magnitude over 3 STFT resolutions (the
                                                    magnitudes and their input gradient on csrc/stft_loss.hip)
  CLAPLoss                 tools/losses.py:259-316  mse + cosine terms of CLAP embeddings of the decoded waveform (clap.py:
                                                    resampler, HTSAT-base audio tower with input gradient, RoBERTa)
All return one value per instance for reduction='instance' (the only mode AudioLCM uses, audio_consistency_model.py:93-102).
"""
import torch
import torch.nn.functional as F
from torch import nn


def reduce(loss, reduction):
    if reduction == "instance":
        return loss
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    raise ValueError("Unknown loss reduction option.")


def _instance_mse(a, b):
    d = (a.float() - b.float()) ** 2
    return d.reshape(d.shape[0], -1).mean(dim=1)


class MSELoss(nn.Module):
    def __init__(self, reduction="instance"):
        super().__init__()
        self.reduction = reduction

    def forward(self, input, target, gt_wav=None, gt_text=None):
        return reduce(_instance_mse(input, target), self.reduction)


class MelLoss(nn.Module):
    def __init__(self, vae, reduction="instance", mse_weight=.7, mel_weight=.3):
        super().__init__()
        object.__setattr__(self, "vae", vae)   # not a submodule: the VAE belongs to the model (and stays frozen)
        self.reduction = reduction
        self.mse_weight, self.mel_weight = mse_weight, mel_weight

    def forward(self, input, target, gt_wav=None, gt_text=None):
        input_mel = self.vae.decode_first_stage(input.float(), allow_grad=True)
        target_mel = self.vae.decode_first_stage(target.float(), allow_grad=True)
        inst = _instance_mse(input_mel, target_mel) * self.mel_weight + _instance_mse(input, target) * self.mse_weight
        return reduce(inst, self.reduction)


class _STFTMagnitudeFn(torch.autograd.Function):
    """(B, T) fp32 waveform -> |STFT| (B, frames, fft_size // 2 + 1) fp32 on ctta_stft_magnitude / _bwd."""

    @staticmethod
    def forward(ctx, x, owner):
        from . import _native as N
        B, T = x.shape
        grad = bool(ctx.needs_input_grad[0])     # (grad mode is off inside forward)
        h = owner._handle(B, T, x.device, grad)
        xc = x.detach().contiguous().float()
        frames = T // owner.shift_size + 1
        mag = torch.empty(B, frames, owner.fft_size // 2 + 1, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            N.check(N.lib().ctta_stft_magnitude(h, N.ptr(xc), B, T, N.ptr(mag), N.stream_ptr()))
        ctx.owner, ctx.h, ctx.shape = owner, h, (B, T)
        if grad:
            owner._pending = ctx.token = object()   # the handle keeps (re, im) of this call until its backward
        return mag

    @staticmethod
    def backward(ctx, dmag):
        from . import _native as N
        owner, (B, T) = ctx.owner, ctx.shape
        if owner._pending is not ctx.token:
            raise RuntimeError("_STFTMagnitude: another differentiable forward ran on this module before backward; use one "
                               "module instance per differentiable input")
        dwav = torch.empty(B, T, dtype=torch.float32, device=dmag.device)
        g = dmag.contiguous().float()
        with torch.cuda.device(dmag.device):
            N.check(N.lib().ctta_stft_magnitude_bwd(ctx.h, N.ptr(g), B, T, N.ptr(dwav), N.stream_ptr()))
        owner._pending = None
        return dwav, None


class _STFTMagnitude(nn.Module):
    """|STFT| of tools/losses.py:146-169 (torch.stft in float64, periodic Hann window centred in fft_size, reflect padding,
    sqrt(clamp(power, 1e-8)), transposed to (B, frames, bins), float32) on the HIP kernels of csrc/stft_loss.hip: split-bf16
    GEMM against the windowed DFT basis (fp32-grade), bf16 GEMM + overlap-add for the input gradient.  Two native handles:
    the differentiable call keeps its spectrum until the backward, a plain call (the target's) must not overwrite it."""

    def __init__(self, fft_size, shift_size, win_length, window):
        super().__init__()
        if window != "hann_window":
            raise ValueError("_STFTMagnitude: only hann_window (the reference's setting) is built")
        self.fft_size, self.shift_size, self.win_length = fft_size, shift_size, win_length
        self._h = {False: None, True: None}
        self._key = {False: None, True: None}
        self._pending = None

    def _handle(self, B, T, dev, grad):
        from . import _native as N
        key = self._key[grad]
        if self._h[grad] is None or key[0] < B or key[1] < T or key[2] != dev:
            self._release(grad)
            Bm, Tm = (max(B, key[0]), max(T, key[1])) if key and key[2] == dev else (B, T)
            h = N.c_void_p()
            with torch.cuda.device(dev):
                N.check(N.lib().ctta_stft_create(self.fft_size, self.shift_size, self.win_length, Bm, Tm, h))
            self._h[grad], self._key[grad] = h, (Bm, Tm, dev)
        return self._h[grad]

    def _release(self, grad):
        if self._h[grad] is not None:
            from . import _native as N
            N.lib().ctta_stft_destroy(self._h[grad])
            self._h[grad] = None

    def __del__(self):
        try:
            self._release(False)
            self._release(True)
        except Exception:
            pass

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("_STFTMagnitude runs on the HIP kernels: the waveform must be a CUDA(ROCm) tensor")
        return _STFTMagnitudeFn.apply(x, self)


class MultiResolutionSTFTLoss(nn.Module):
    """factor_mse * mse(latents) + factor_mag * mean_r log-magnitude L1 + factor_sc * mean_r spectral convergence.
    The reference reads an undefined `self.sr` (tools/losses.py:241); 16 kHz -- CLAPLoss's value and the vocoder's
    rate -- is used here so that the loss is usable."""

    def __init__(self, vae, reduction="instance", fft_sizes=(1024, 2048, 512), hop_sizes=(120, 240, 50),
                 win_lengths=(600, 1200, 240), window="hann_window", factor_sc=0.2, factor_mag=0.2, factor_mse=1):
        super().__init__()
        assert len(fft_sizes) == len(hop_sizes) == len(win_lengths)
        object.__setattr__(self, "vae", vae)
        self.reduction = reduction
        self.sr = 16000
        self.stfts = nn.ModuleList(_STFTMagnitude(f, h, w, window) for f, h, w in zip(fft_sizes, hop_sizes, win_lengths))
        self.factor_sc, self.factor_mag, self.factor_mse = factor_sc, factor_mag, factor_mse

    def _wav(self, latent):
        mel = self.vae.decode_first_stage(latent.float(), allow_grad=True)
        return self.vae.decode_to_waveform(mel.float(), allow_grad=True)[:, :int(self.sr * 10)]

    def forward(self, input, target, gt_wav=None, gt_text=None):
        mse = reduce(_instance_mse(input, target), self.reduction)
        x, y = self._wav(input), self._wav(target)
        sc = mag = 0.
        for stft in self.stfts:
            xm, ym = stft(x), stft(y)
            diff = (ym - xm).reshape(xm.shape[0], -1)
            sc = sc + reduce(diff.norm(dim=1) / ym.reshape(ym.shape[0], -1).norm(dim=1), self.reduction)
            l1 = (torch.log(ym) - torch.log(xm)).abs()
            mag = mag + reduce(l1.reshape(l1.shape[0], -1).mean(dim=1), self.reduction)
        n = len(self.stfts)
        return self.factor_mse * mse + self.factor_mag * (mag / n) + self.factor_sc * (sc / n)


class CLAPLoss(nn.Module):
    """tools/losses.py:259-316: mse_weight * mse(latents) + clap_weight * (2 - cos(audio(input), text) - cos(audio(input),
    audio(gt))).  The predicted latent is decoded to a waveform with allow_grad=True, cut to 10 s, resampled 16 -> 48 kHz
    (Kaiser-windowed sinc, `clap.Resampler`) and embedded by the frozen CLAP towers (`clap.CLAP_Module`: HTSAT-base audio
    tower with input gradient, RoBERTa text tower) -- every stage on the HIP kernels.

    The reference loads `ckpt/music_audioset_epoch_15_esc_90.14.pt` in its constructor; pass `clap=` (a ready
    `clap.CLAP_Module`) or `ckpt=` here, there is no download path offline.  `tokenizer=` may replace the RoBERTa
    tokenizer (host-side string work, stays transformers')."""

    def __init__(self, vae, reduction="instance", mse_weight=1., clap_weight=1., clap=None, ckpt="ckpt/music_audioset_epoch_15_esc_90.14.pt",
                 tokenizer=None):
        super().__init__()
        from . import clap as C
        object.__setattr__(self, "vae", vae)
        self.reduction = reduction
        self.sr = 16000
        if clap is None:
            import os
            if not os.path.exists(ckpt):
                raise RuntimeError("CLAPLoss: the CLAP checkpoint %r is not on this box (tools/losses.py:271 loads it in the "
                                   "constructor); pass clap=<clap.CLAP_Module with weights> or ckpt=<path>" % ckpt)
            clap = C.CLAP_Module(enable_fusion=False, amodel="HTSAT-base", tokenizer=tokenizer)
            clap.load_ckpt(ckpt)
        self.clap = clap
        self.clap.eval()
        self.clap.requires_grad_(False)
        self.resample = C.Resampler(16000, 48000, lowpass_filter_width=64, rolloff=0.9475937167399596,
                                    beta=14.769656459379492)
        self.mse_weight, self.clap_weight = mse_weight, clap_weight

    def forward(self, input, target, gt_wav, captions, use_ema=False):
        mse_loss = reduce(_instance_mse(input, target), self.reduction)
        input_mel = self.vae.decode_first_stage(input.float(), allow_grad=True, use_ema=use_ema)
        input_wav = self.vae.decode_to_waveform(input_mel.float(), allow_grad=True)
        input_wav = input_wav[:, :int(self.sr * 10)]
        input_wav, gt48 = (self.resample(w[:, :int(self.sr * 10)].float()) for w in (input_wav, gt_wav.to(input_wav.device)))
        input_feat = self.clap.get_audio_embedding_from_data(input_wav, use_tensor=True)
        with torch.no_grad():
            gt_wav_feat = self.clap.get_audio_embedding_from_data(gt48, use_tensor=True)
            if isinstance(captions, dict):                  # pre-computed prompt states (benchmarks / tests): CLAP text
                captions = captions["clap_text_features"]    # features ride along under this key
            caption_feat = (captions if torch.is_tensor(captions)     # (offline boxes have no tokenizer files)
                            else self.clap.get_text_embedding(captions, use_tensor=True))
        gen_text_similarity = F.cosine_similarity(input_feat, caption_feat, dim=1)
        gen_gt_similarity = F.cosine_similarity(input_feat, gt_wav_feat, dim=1)
        instance_loss = self.mse_weight * mse_loss + self.clap_weight * (2 - gen_text_similarity - gen_gt_similarity)
        return reduce(instance_loss, self.reduction)
