"""CPU restatement of the waveform -> log-mel front-end of the training step (TEST INFRASTRUCTURE ONLY):
`tools/torch_tools.py:126-135` (wav_to_fbank) -> `:78-82` (get_mel_from_wav: clip to [-1,1], nan_to_num) ->
`audioldm/audio/stft.py:165-186` (TacotronSTFT.mel_spectrogram) -> `:52-84` (STFT.transform: reflect pad n_fft/2,
conv1d with the windowed Fourier basis at hop 160, magnitude) -> mel_basis @ magnitude ->
`audio_processing.py:85-91` (log(clamp(x, 1e-5))) -> `torch_tools.py:38-51` (_pad_spec to 1024 frames).

Third-party pieces the reference imports and this image lacks -- librosa==0.10.0.post2 (environment.yml:106):
  * `librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` with its defaults htk=False, norm="slaney": restated below from
    the published algorithm (Slaney's Auditory Toolbox mel scale: linear below 1 kHz at 200/3 Hz per mel, logarithmic
    above with step ln(6.4)/27; triangular filters on the rfft bin centres; area normalisation 2/(f[i+2]-f[i])).
  * `librosa.util.pad_center` (no-op here: win_length == filter_length) and `scipy.signal.get_window('hann',
    fftbins=True)` (periodic Hann), which scipy provides.
The golden fixtures are produced by the reference's own STFT / TacotronSTFT / wav_to_fbank code with `librosa.filters.mel`
and `librosa.util.pad_center` bound to these restatements (tests/golden/make_golden_mel.py)."""
import numpy as np
import torch
import torch.nn.functional as F


def hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr=16000, n_fft=1024, n_mels=64, fmin=0.0, fmax=8000.0):
    """librosa.filters.mel (0.10): (n_mels, 1 + n_fft/2) float32, Slaney scale, slaney (area) normalisation."""
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights.astype(np.float32)


def pad_center(data, size):
    """librosa.util.pad_center along the last axis."""
    n = data.shape[-1]
    lpad = int((size - n) // 2)
    return np.pad(data, [(0, 0)] * (data.ndim - 1) + [(lpad, int(size - n - lpad))], mode="constant")


def stft_basis(filter_length=1024, win_length=1024):
    """STFT.__init__ (stft.py:17-48): rows [Re(F[:cutoff]) ; Im(F[:cutoff])] of the DFT matrix times the periodic Hann
    window -> (2*cutoff, filter_length) float32, cutoff = filter_length/2 + 1."""
    fourier = np.fft.fft(np.eye(filter_length))
    cutoff = filter_length // 2 + 1
    basis = np.vstack([np.real(fourier[:cutoff, :]), np.imag(fourier[:cutoff, :])])
    n = np.arange(win_length)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)          # scipy get_window('hann', fftbins=True)
    window = pad_center(window, filter_length)
    return (torch.FloatTensor(basis) * torch.from_numpy(window).float()).float()


def wav_to_fbank(waveforms, target_length=1024, filter_length=1024, hop_length=160, n_mels=64, sr=16000, fmin=0.0,
                 fmax=8000.0):
    """waveforms (B, T) in [-1, 1] -> log-mel (B, target_length, n_mels) and log-magnitude (B, target_length, 512)."""
    audio = torch.nan_to_num(torch.clip(waveforms.float(), -1, 1))
    x = F.pad(audio[:, None, None, :], (filter_length // 2, filter_length // 2, 0, 0), mode="reflect")[:, 0]
    ft = F.conv1d(x, stft_basis(filter_length)[:, None, :], stride=hop_length)
    cutoff = filter_length // 2 + 1
    mag = torch.sqrt(ft[:, :cutoff] ** 2 + ft[:, cutoff:] ** 2)
    mel = torch.matmul(torch.from_numpy(mel_filterbank(sr, filter_length, n_mels, fmin, fmax)), mag)
    fbank = torch.log(torch.clamp(mel, min=1e-5)).transpose(1, 2)
    logmag = torch.log(torch.clamp(mag, min=1e-5)).transpose(1, 2)

    def pad_spec(s):
        p = target_length - s.shape[1]
        if p > 0:
            s = torch.cat([s, torch.zeros(s.shape[0], p, s.shape[2])], 1)
        elif p < 0:
            s = s[:, :target_length]
        return s[:, :, :-1] if s.shape[2] % 2 else s
    return pad_spec(fbank), pad_spec(logmag)
