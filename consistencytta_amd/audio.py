"""Waveform -> log-mel front-end mirrors: `audioldm.audio.stft.TacotronSTFT` (stft.py:132-186) and
`tools.torch_tools.wav_to_fbank` (torch_tools.py:126-135), backed by the HIP front-end (csrc/mel_frontend.hip):
reflect pad, STFT as a split-bf16 MFMA GEMM against the windowed DFT basis, magnitude, Slaney mel filterbank,
log(clamp(x, 1e-5)), frame padding.  Same call signatures as the reference; the arithmetic runs on the GPU only."""
import torch
from torch import nn

from . import _native as N


class TacotronSTFT(nn.Module):
    def __init__(self, filter_length=1024, hop_length=160, win_length=1024, n_mel_channels=64, sampling_rate=16000,
                 mel_fmin=0.0, mel_fmax=8000.0):
        super().__init__()
        self.filter_length, self.hop_length, self.win_length = int(filter_length), int(hop_length), int(win_length)
        self.n_mel_channels, self.sampling_rate = int(n_mel_channels), int(sampling_rate)
        self.mel_fmin, self.mel_fmax = float(mel_fmin), float(mel_fmax)
        self.register_buffer("_device_anchor", torch.zeros(1), persistent=False)
        self._h = None
        self._h_key = None

    @property
    def device(self):
        return self._device_anchor.device

    def _release(self):
        if getattr(self, "_h", None):
            N.lib().ctta_mel_frontend_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure(self, B, T):
        key = self._h_key
        if self._h is None or B > key[0] or T > key[1] or key[2] != self.device:
            self._release()
            Bm = max(B, key[0]) if key else B
            Tm = max(T, key[1]) if key else T
            h = N.c_void_p()
            with torch.cuda.device(self.device):
                N.check(N.lib().ctta_mel_frontend_create(self.filter_length, self.hop_length, self.win_length,
                                                         self.n_mel_channels, self.sampling_rate, self.mel_fmin,
                                                         self.mel_fmax, Bm, Tm, h))
            self._h, self._h_key = h, (Bm, Tm, self.device)
        return self._h

    def fbank(self, y, target_length=None, want_logmag=True):
        """y (B, T) -> fbank (B, frames|target_length, n_mels) [, log-magnitudes (B, ., filter_length/2)]."""
        if self.device.type != "cuda":
            raise N.CttaError("TacotronSTFT is on %s: the HIP front-end has no CPU path (call .cuda())" % self.device)
        if y.ndim != 2:
            raise ValueError("waveforms must be (batch, samples), got %s" % (tuple(y.shape),))
        y = y.detach().to(device=self.device, dtype=torch.float32).contiguous()
        B, T = y.shape
        frames = T // self.hop_length + 1
        target = int(target_length) if target_length is not None else frames
        h = self._ensure(B, T)
        fb = torch.empty(B, target, self.n_mel_channels, dtype=torch.float32, device=self.device)
        lm = torch.empty(B, target, self.filter_length // 2, dtype=torch.float32, device=self.device) if want_logmag else None
        with torch.cuda.device(self.device):
            N.check(N.lib().ctta_wav_to_fbank(h, N.ptr(y), B, T, target, N.ptr(fb), N.ptr(lm), N.stream_ptr()))
        return fb, lm

    def mel_spectrogram(self, y, normalize_fun=torch.log):
        """(B, T) in [-1, 1] -> mel_output (B, n_mel_channels, frames), log-magnitudes (B, filter_length/2, frames).
        The reference also returns the per-frame energy and the 513th (Nyquist) log-magnitude row; neither is used on
        the ConsistencyTTA path (`_pad_spec` drops the odd bin) and they are not produced here (energy is None)."""
        if normalize_fun is not torch.log:
            raise NotImplementedError("only the natural-log compression the reference uses is built")
        assert torch.min(y.data) >= -1, torch.min(y.data)
        assert torch.max(y.data) <= 1, torch.max(y.data)
        fb, lm = self.fbank(y)
        return fb.transpose(1, 2), lm.transpose(1, 2), None


def wav_to_fbank(waveforms, target_length=1024, fn_STFT=None):
    """tools/torch_tools.py:126-135: clip / nan_to_num, mel spectrogram, transpose to (B, frames, n_mels), pad or cut to
    `target_length` frames.  Returns (fbank, log_magnitudes_stft)."""
    assert fn_STFT is not None
    return fn_STFT.fbank(waveforms, target_length)
