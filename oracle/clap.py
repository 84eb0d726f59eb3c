"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

fp32 PyTorch-CPU restatement of the CLAP loss path of the CLAP fine-tuning stage (tools/losses.py:259-316):

  wav (16 kHz) -> torchaudio.functional.resample(16 k -> 48 k, sinc_interp_kaiser)         tools/losses.py:299-303
      -> laion_clap HTSAT audio tower (clap_module/htsat.py:615-1013, `create_htsat_model("base")`)
      -> CLAP.audio_projection, cosine similarity                                           clap_module/model.py:537-541,669

Third-party arithmetic that is NOT vendored under /root/reference, restated from its published algorithm:
  * torchaudio==2.0.2 (environment.yml) `functional.resample` / `_get_sinc_resample_kernel` / `_apply_sinc_resample_kernel`:
    polyphase windowed-sinc interpolation, Kaiser window; pinned by known answers (DC gain, a band-limited sine,
    tests/test_oracle_golden.py).
  * torchlibrosa==0.1.0 `stft.Spectrogram` (conv1d against a windowed DFT basis, centre reflect padding, power 2) and
    `stft.LogmelFilterBank` (librosa Slaney mel basis, 10*log10(clamp(x, amin)) - 10*log10(max(amin, ref))), used by
    htsat.py:684-697; `librosa.filters.mel` itself is restated in oracle/mel.py.
The HTSAT network is pinned against the reference's OWN htsat.py imported with these two classes bound to the
restatement (tests/golden/make_golden_clap.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import mel as omel


# ------------------------------------------------------------------------------------------ torchaudio resample
def sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99, beta=None):
    """torchaudio.functional.functional._get_sinc_resample_kernel(resampling_method="sinc_interp_kaiser"):
    returns (kernels (new, 1, 2*width + orig) float32, width) for the gcd-reduced rates."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    if beta is None:
        beta = 14.769656459379492
    beta_t = torch.tensor(float(beta), dtype=torch.float64)
    window = torch.i0(beta_t * torch.sqrt(1 - (t / lowpass_filter_width) ** 2)) / torch.i0(beta_t)
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels = kernels * window * scale
    return kernels.to(torch.float32), width, orig, new


def resample(wav, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99, beta=None):
    """_apply_sinc_resample_kernel: pad (width, width + orig), strided conv1d with the `new` phase filters, interleave
    the phases, cut to ceil(new * length / orig)."""
    kernels, width, orig, new = sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width, rolloff, beta)
    shape = wav.shape
    x = wav.reshape(-1, shape[-1])
    length = x.shape[1]
    x = F.pad(x, (width, width + orig))
    y = F.conv1d(x[:, None], kernels, stride=orig)
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = int(math.ceil(new * length / orig))
    return y[..., :target].reshape(shape[:-1] + (target,))


# ------------------------------------------------------------------------------------------ torchlibrosa front-end
def dft_basis(n_fft, win_length):
    """Spectrogram's conv kernels: rows = n_fft // 2 + 1 frequencies, periodic Hann window centred in n_fft."""
    n = np.arange(n_fft)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)   # scipy get_window('hann', fftbins=True)
    lpad = (n_fft - win_length) // 2
    w = np.zeros(n_fft)
    w[lpad:lpad + win_length] = win
    k = np.arange(n_fft // 2 + 1)
    ang = 2.0 * np.pi * np.outer(k, n) / n_fft
    return (torch.from_numpy((np.cos(ang) * w).astype(np.float32)),
            torch.from_numpy((-np.sin(ang) * w).astype(np.float32)))


def spectrogram_power(wav, n_fft=1024, hop=480, win_length=1024):
    """(B, L) -> (B, 1, frames, n_fft // 2 + 1): |STFT|^2, centre = True, reflect padding."""
    real_w, imag_w = dft_basis(n_fft, win_length)
    x = F.pad(wav[:, None, :], (n_fft // 2, n_fft // 2), mode="reflect")
    re = F.conv1d(x, real_w[:, None, :], stride=hop)
    im = F.conv1d(x, imag_w[:, None, :], stride=hop)
    return (re ** 2 + im ** 2).transpose(1, 2)[:, None]


def logmel(power, sr=48000, n_fft=1024, n_mels=64, fmin=50, fmax=14000, amin=1e-10, ref=1.0):
    melw = torch.from_numpy(omel.mel_filterbank(sr, n_fft, n_mels, fmin, fmax).T.astype(np.float32))   # (freq, mel)
    m = power @ melw
    return 10.0 * torch.log10(torch.clamp(m, min=amin)) - 10.0 * math.log10(max(amin, ref))


# ------------------------------------------------------------------------------------------ HTSAT (Swin) tower
HTSAT_BASE = dict(spec_size=256, patch_size=4, patch_stride=4, embed_dim=128, depths=[2, 2, 12, 2],
                  num_heads=[4, 8, 16, 32], window_size=8, mlp_ratio=4.0, mel_bins=64, sample_rate=48000,
                  n_fft=1024, hop=480, fmin=50, fmax=14000, num_classes=527)


def relative_position_index(ws):
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0) + (ws - 1)
    return rel[:, :, 0] * (2 * ws - 1) + rel[:, :, 1]


def shift_attn_mask(H, W, ws, shift):
    """htsat.py:429-447: 0 / -100 mask per window for the cyclically shifted partition."""
    img = torch.zeros(H, W)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = img.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    d = mw[:, None, :] - mw[:, :, None]
    return torch.where(d != 0, torch.tensor(-100.0), torch.tensor(0.0))


def window_partition(x, ws):
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)


def window_reverse(w, ws, H, W):
    B = w.shape[0] // ((H // ws) * (W // ws))
    return w.view(B, H // ws, W // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def swin_block(sd, p, x, res, heads, ws, shift):
    H = W = res
    if res <= ws:
        ws, shift = res, 0
    B, L, C = x.shape
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"]).view(B, H, W, C)
    if shift:
        h = torch.roll(h, (-shift, -shift), (1, 2))
    xw = window_partition(h, ws)
    N = ws * ws
    qkv = F.linear(xw, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).view(-1, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    attn = (qkv[0] * (C // heads) ** -0.5) @ qkv[1].transpose(-2, -1)
    bias = sd[p + "attn.relative_position_bias_table"][relative_position_index(ws).view(-1)].view(N, N, heads).permute(2, 0, 1)
    attn = attn + bias[None]
    if shift:
        m = shift_attn_mask(H, W, ws, shift)
        attn = (attn.view(-1, m.shape[0], heads, N, N) + m[None, :, None]).view(-1, heads, N, N)
    o = (attn.softmax(-1) @ qkv[2]).transpose(1, 2).reshape(-1, N, C)
    o = F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    h = window_reverse(o, ws, H, W)
    if shift:
        h = torch.roll(h, (shift, shift), (1, 2))
    x = x + h.view(B, L, C)
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"],
                 sd[p + "mlp.fc2.bias"])
    return x + h


def patch_merging(sd, p, x, res):
    B, L, C = x.shape
    x = x.view(B, res, res, C)
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)
    x = F.layer_norm(x, (4 * C,), sd[p + "norm.weight"], sd[p + "norm.bias"])
    return F.linear(x, sd[p + "reduction.weight"])


def wav_to_image(cfg, sd, wav, prefix="audio_branch."):
    """htsat.py:911-925,856-878 in eval mode: log-mel, bn0 over mel bins, bicubic stretch of the frame axis to
    spec_size * freq_ratio, fold into a (spec_size, spec_size) image."""
    P = prefix
    x = logmel(spectrogram_power(wav, cfg["n_fft"], cfg["hop"], cfg["n_fft"]), cfg["sample_rate"], cfg["n_fft"],
               cfg["mel_bins"], cfg["fmin"], cfg["fmax"])
    x = F.batch_norm(x.transpose(1, 3), sd[P + "bn0.running_mean"], sd[P + "bn0.running_var"], sd[P + "bn0.weight"],
                     sd[P + "bn0.bias"], False, 0.0, 1e-5).transpose(1, 3)
    ratio = cfg["spec_size"] // cfg["mel_bins"]
    target_t = cfg["spec_size"] * ratio
    B, C, T, Fq = x.shape
    assert T <= target_t and Fq <= cfg["spec_size"] // ratio
    if T < target_t:
        x = F.interpolate(x, (target_t, Fq), mode="bicubic", align_corners=True)
    x = x.permute(0, 1, 3, 2).reshape(B, C, Fq, ratio, target_t // ratio).permute(0, 1, 3, 2, 4)
    return x.reshape(B, C, Fq * ratio, target_t // ratio)


def htsat_embedding(cfg, sd, wav, prefix="audio_branch.", taps=None):
    """`encode_audio(...)["embedding"]` (htsat.py:795-823): mean over the final 8 x 8 token grid of the normalised
    last-stage features.  wav (B, L) at cfg['sample_rate']."""
    P = prefix
    x = wav_to_image(cfg, sd, wav, prefix)
    if taps is not None:
        taps["image"] = x.clone()
    ps = cfg["patch_size"]
    x = F.conv2d(x, sd[P + "patch_embed.proj.weight"], sd[P + "patch_embed.proj.bias"], stride=cfg["patch_stride"],
                 padding=(ps - cfg["patch_stride"]) // 2).flatten(2).transpose(1, 2)
    C = cfg["embed_dim"]
    x = F.layer_norm(x, (C,), sd[P + "patch_embed.norm.weight"], sd[P + "patch_embed.norm.bias"])
    res = cfg["spec_size"] // cfg["patch_stride"]
    ws = cfg["window_size"]
    for i, (depth, heads) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        for j in range(depth):
            x = swin_block(sd, P + "layers.%d.blocks.%d." % (i, j), x, res, heads, ws, 0 if j % 2 == 0 else ws // 2)
        if taps is not None:
            taps["layer%d" % i] = x.clone()
        if i < len(cfg["depths"]) - 1:
            x = patch_merging(sd, P + "layers.%d.downsample." % i, x, res)
            res //= 2
    x = F.layer_norm(x, (x.shape[-1],), sd[P + "norm.weight"], sd[P + "norm.bias"])
    return x.mean(1)


def audio_features(cfg, sd, wav48k):
    """CLAP.get_audio_embedding (model.py:726-744): projection MLP (Linear, ReLU, Linear), then L2 normalisation."""
    e = htsat_embedding(cfg, sd, wav48k)
    e = F.linear(F.relu(F.linear(e, sd["audio_projection.0.weight"], sd["audio_projection.0.bias"])),
                 sd["audio_projection.2.weight"], sd["audio_projection.2.bias"])
    return F.normalize(e, dim=-1)


def clap_instance_loss(cfg, sd, input_wav16k, gt_wav16k, text_features, mse_instance, mse_weight=1.0, clap_weight=0.1):
    """tools/losses.py:299-315 after the decode: resample both waveforms to 48 kHz, audio embeddings, cosine terms."""
    kw = dict(lowpass_filter_width=64, rolloff=0.9475937167399596, beta=14.769656459379492)
    fin = audio_features(cfg, sd, resample(input_wav16k[:, :160000], 16000, 48000, **kw))
    fgt = audio_features(cfg, sd, resample(gt_wav16k[:, :160000], 16000, 48000, **kw))
    sim_t = F.cosine_similarity(fin, text_features, dim=1)
    sim_g = F.cosine_similarity(fin, fgt, dim=1)
    return mse_weight * mse_instance + clap_weight * (2 - sim_t - sim_g)
