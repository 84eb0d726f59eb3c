"""Fixtures for the CLAP audio tower (tools/losses.py:259-316), produced by the REFERENCE's own
`laion_clap/clap_module/htsat.py` (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_clap.py

`htsat.py` is loaded from /root/reference into a stub `clap_module` package (its real __init__ pulls in the whole
training stack).  Third-party modules this image lacks are bound as follows:
  * torchlibrosa.stft.Spectrogram / LogmelFilterBank (htsat.py:20, used at :684-697) -> the restatement in
    oracle/clap.py (the same move as librosa.filters.mel for the TacotronSTFT fixtures);
  * torchlibrosa.augmentation.SpecAugmentation (training only), torchvision.ops.misc.FrozenBatchNorm2d, h5py -> inert stubs.
Weights are the build's deterministic generator over the reference module's own state-dict keys and shapes (stored in
the fixture as data, so that the GPU box can rebuild them without the reference).  The CLAP checkpoint
(music_audioset_epoch_15_esc_90.14.pt) is not available offline.

Two cases: HTSAT-base (`create_htsat_model` "base": embed 128, depths 2/2/12/2, heads 4/8/16/32, window 8) on 10 s of
48 kHz audio -- embedding in full, the input gradient as norm + strided sample -- and a small tower (spec 64, embed 64,
depths 2/2/2/2) whose stages cover shifted 8x8 windows, one full window, and windows shrunk to 4x4 and 2x2, with every
tensor stored in full.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
from consistencytta_amd import spec  # noqa: E402
from oracle import clap as oclap  # noqa: E402

REF = os.environ.get("CTTA_REFERENCE_ROOT", "/root/reference")


def load_reference_htsat():
    sys.dont_write_bytecode = True

    class Spectrogram(torch.nn.Module):
        def __init__(self, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True, pad_mode="reflect",
                     power=2.0, freeze_parameters=True):
            super().__init__()
            assert window == "hann" and center and pad_mode == "reflect" and power == 2.0
            self.n_fft, self.hop, self.win = n_fft, hop_length, win_length

        def forward(self, x):
            return oclap.spectrogram_power(x, self.n_fft, self.hop, self.win)

    class LogmelFilterBank(torch.nn.Module):
        def __init__(self, sr=22050, n_fft=2048, n_mels=64, fmin=0.0, fmax=None, is_log=True, ref=1.0, amin=1e-10,
                     top_db=80.0, freeze_parameters=True):
            super().__init__()
            assert is_log and top_db is None
            self.kw = dict(sr=sr, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax, amin=amin, ref=ref)

        def forward(self, x):
            return oclap.logmel(x, **self.kw)

    class SpecAugmentation(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()

        def forward(self, x):
            raise RuntimeError("training-only")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    mod("torchlibrosa")
    mod("torchlibrosa.stft", Spectrogram=Spectrogram, LogmelFilterBank=LogmelFilterBank)
    mod("torchlibrosa.augmentation", SpecAugmentation=SpecAugmentation)
    if "torchvision" not in sys.modules:
        mod("torchvision")
        mod("torchvision.ops")
        mod("torchvision.ops.misc", FrozenBatchNorm2d=type("FrozenBatchNorm2d", (torch.nn.Module,), {}))
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    pkg = types.ModuleType("clap_module")
    pkg.__path__ = [os.path.join(REF, "laion_clap", "clap_module")]
    sys.modules["clap_module"] = pkg
    out = {}
    for name in ("utils", "feature_fusion", "htsat"):
        sp = importlib.util.spec_from_file_location("clap_module." + name, os.path.join(pkg.__path__[0], name + ".py"))
        m = importlib.util.module_from_spec(sp)
        sys.modules[sp.name] = m
        sp.loader.exec_module(m)
        out[name] = m
    return out["htsat"]


def det_weights(model, tag, seed):
    """Deterministic values for every floating-point entry of the reference module's state dict (structural integer
    buffers -- relative_position_index, num_batches_tracked -- and the 0 / -100 shift masks keep their own values)."""
    sd = model.state_dict()
    new, keys, shapes = {}, [], []
    for k, v in sd.items():
        if not v.dtype.is_floating_point or k.endswith("attn_mask"):
            continue
        keys.append(k)
        shapes.append(list(v.shape))
        w = torch.from_numpy(spec.clap_det_weight(tag + k, tuple(v.shape), seed))
        new[k] = w
    model.load_state_dict(new, strict=False)
    return new, keys, shapes


def run_case(H, cfg_obj, ctor_kw, wav, tag, seed, out, full):
    torch.manual_seed(0)
    m = H.HTSAT_Swin_Transformer(config=cfg_obj, **ctor_kw).eval()
    sd, keys, shapes = det_weights(m, tag + ".", seed)
    out[tag + "_keys"] = np.array(keys)
    out[tag + "_shapes"] = np.array([",".join(str(d) for d in s) for s in shapes])
    w = wav.clone().requires_grad_(True)
    emb = m({"waveform": w}, device="cpu")["embedding"]
    direction = cases.t(spec.det_uniform(tag + ".dir", tuple(emb.shape), 9))
    (emb * direction).sum().backward()
    out[tag + "_embedding"] = emb.detach().numpy()
    g = w.grad.detach()
    if full:
        out[tag + "_grad"] = g.numpy()
    else:
        out[tag + "_grad_norm"] = np.float64(float(g.double().norm()))
        idx = torch.from_numpy(cases.sample_index(g[0].numel(), 4096))
        out[tag + "_grad_sample"] = g[:, idx].numpy()
    print(tag, "embedding", tuple(emb.shape), "|emb|", float(emb.norm()), "|grad|", float(g.norm()), "params", len(keys))
    return m, sd


def main():
    H = load_reference_htsat()
    out = {}
    base = types.SimpleNamespace(mel_bins=64, window_size=1024, hop_size=480, sample_rate=48000, fmin=50, fmax=14000,
                                 class_num=527, model_name="base")
    # ---- small tower, every tensor in full
    B, L = 2, 28800
    wav = cases.t(spec.det_uniform("clap.tiny.wav", (B, L), 3)) * 0.4
    tiny_kw = dict(spec_size=64, patch_size=4, patch_stride=(4, 4), num_classes=11, embed_dim=64, depths=[2, 2, 2, 2],
                   num_heads=[2, 4, 8, 16], window_size=8)
    m, sd = run_case(H, base, tiny_kw, wav, "tiny", 5, out, full=True)
    with torch.no_grad():   # the oracle's own front half against the reference's (bn0 + bicubic + fold)
        x = m.logmel_extractor(m.spectrogram_extractor(wav)).transpose(1, 3)
        out["tiny_image"] = m.reshape_wav2img(m.bn0(x).transpose(1, 3)).numpy()
    # ---- HTSAT-base on 10 s
    wav = cases.t(spec.det_uniform("clap.base.wav", (2, 480000), 4)) * 0.4
    model = H.create_htsat_model(base)
    base_kw = dict(spec_size=256, patch_size=4, patch_stride=(4, 4), num_classes=527, embed_dim=128, depths=[2, 2, 12, 2],
                   num_heads=[4, 8, 16, 32], window_size=8)
    assert sum(p.numel() for p in model.parameters()) == sum(
        p.numel() for p in H.HTSAT_Swin_Transformer(config=base, **base_kw).parameters())
    run_case(H, base, base_kw, wav, "base", 6, out, full=False)
    path = os.path.join(HERE, "clap_htsat.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    main()
