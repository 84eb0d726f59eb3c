"""Evaluation suite of the reference (`audioldm_eval/`): the PANNs Cnn14 classifier on the HIP path and the metric
arithmetic on its features.

  * `Cnn14` -- audioldm_eval/feature_extractors/panns/models.py:168-323 in eval mode: waveform -> power STFT -> dB log-mel
    -> bn0 -> six ConvBlocks (conv3x3, BatchNorm, ReLU, twice; 2x2 average pooling) -> mean over frequency, max + mean over
    time -> fc1 + ReLU (the "2048" embedding) -> fc_audioset ("logits").  The twelve convolutions (40 GFLOP per 10 s clip)
    run on ctta_conv_gemm with BatchNorm folded into weights / bias and ReLU in the epilogue, the front end on
    ctta_wav_to_logmel_db (the torchlibrosa Spectrogram + LogmelFilterBank the CLAP tower already uses); state-dict keys are
    the reference's, so `Cnn14_16k_mAP=0.438.pth` loads as it is.
  * `calculate_fid` / `calculate_isc` / `calculate_kid` / `calculate_kl` -- audioldm_eval/metrics/{fid,isc,kid,kl}.py: same
    names, arguments, dictionary keys and random streams (np.random.RandomState(rng_seed)); host arithmetic on (N, 2048) /
    (N, 527) feature matrices, as in the reference (numpy / scipy there too).
  * `EvaluationHelper` -- audioldm_eval/eval.py:58-349: `get_featuresdict`, `calculate_metrics` on two directories of .wav
    files (or two lists of waveforms), the CLAP scores through `consistencytta_amd.clap.CLAP_Module`.

Not rebuilt, and reported as NaN exactly like a key the reference leaves out (eval.py:297-299 `out.get(key, nan)`):
frechet_audio_distance (VGGish through `torch.hub`, metrics/fad.py:53 -- its sources are not in the reference tree), lsd / ssim_stft
(`ssr_eval`, a pip dependency) and psnr / ssim (`skimage`).  Resampling of files whose rate is not an integer multiple of the
target (`resampy`, load_mel.py:25-28) is refused loudly.
"""
import os
from collections import OrderedDict

import numpy as np
import scipy.linalg
import torch

from . import _native as N
from . import spec
from .clap import PackedLinear, _check_cuda, _conv, _desc
from .modules import _ParamTree

__all__ = ["Cnn14", "EvaluationHelper", "calculate_fid", "calculate_isc", "calculate_kid", "calculate_kl",
           "read_centered_wav", "pad_short_audio"]


# ------------------------------------------------------------------------------------------------ classifier
class Cnn14(_ParamTree):
    """`Cnn14(features_list, sample_rate, window_size, hop_size, mel_bins, fmin, fmax, classes_num)` of the reference,
    eval mode (no SpecAugment, mixup or dropout: eval.py:83 calls `.eval()`), without the checkpoint download of its
    constructor (models.py:236-253): load the released state dict with `load_state_dict(torch.load(...)["model"])`."""

    def __init__(self, features_list=("2048", "logits"), sample_rate=16000, window_size=512, hop_size=160, mel_bins=64,
                 fmin=50, fmax=8000, classes_num=527):
        super().__init__()
        self.features_list = list(features_list)
        self.cfg = dict(sample_rate=sample_rate, n_fft=window_size, hop=hop_size, mel_bins=mel_bins, fmin=fmin, fmax=fmax,
                        classes_num=classes_num, widths=list(spec.CNN14_16K_CONFIG["widths"]))
        self._register(spec.cnn14_param_spec(self.cfg))
        self.requires_grad_(False)
        self._packed = self._packed_ver = None
        self._fe = self._fe_key = None

    @property
    def device(self):
        return self.get_parameter("fc1.weight").device

    def load_state_dict(self, state_dict, strict=True):
        """A released checkpoint also holds `num_batches_tracked` counters (structural here); torchlibrosa's frozen STFT /
        mel matrices are accepted and ignored -- the front end derives its own."""
        keep = OrderedDict((k, v) for k, v in state_dict.items() if not k.endswith("num_batches_tracked"))
        if strict:
            for k in spec.CNN14_STRUCTURAL:
                keep.setdefault(k, self.get_parameter(k).detach())
        return super().load_state_dict(keep, strict=strict)

    def init_deterministic(self, seed=0, prefix="cnn14."):
        with torch.no_grad():
            for k, p in self.named_parameters():
                if k not in spec.CNN14_STRUCTURAL:
                    p.copy_(torch.from_numpy(spec.cnn14_det_weight(prefix + k, tuple(p.shape), seed)).to(p.device))
        return self

    def __del__(self):
        try:
            self._release_frontend()
        except Exception:
            pass

    def _release_frontend(self):
        if self._fe is not None:
            N.lib().ctta_mel_frontend_destroy(self._fe)
            self._fe = self._fe_key = None

    def _frontend(self, B, L):
        key = self._fe_key
        if self._fe is None or B > key[0] or L > key[1]:
            Bm, Lm = (max(B, key[0]), max(L, key[1])) if key else (B, L)
            self._release_frontend()
            c = self.cfg
            h = N.c_void_p()
            N.check(N.lib().ctta_mel_frontend_create(c["n_fft"], c["hop"], c["n_fft"], c["mel_bins"], c["sample_rate"],
                                                     float(c["fmin"]), float(c["fmax"]), Bm, Lm, h))
            self._fe, self._fe_key = h, (Bm, Lm)
        return self._fe

    def _pack(self):
        ver = self._weights_version()
        if self._packed is not None and self._packed_ver == ver:
            return self._packed
        sd = {k: p.detach() for k, p in self.named_parameters()}
        for k, p in sd.items():
            _check_cuda(p, "parameter '%s'" % k)

        def bn_affine(p):        # BatchNorm2d in eval mode, eps 1e-5 (nn.BatchNorm2d default, models.py:53-54,224)
            scale = sd[p + "weight"].float() / torch.sqrt(sd[p + "running_var"].float() + 1e-5)
            return scale, sd[p + "bias"].float() - sd[p + "running_mean"].float() * scale

        P = {}
        P["bn0"] = tuple(t.contiguous() for t in bn_affine("bn0."))
        cin = 1
        for i, c in enumerate(self.cfg["widths"]):
            p = "conv_block%d." % (i + 1)
            for j, ci in ((1, cin), (2, c)):
                cp = max(8, ci)                              # NHWC channel count of the layer's input (1 -> 8)
                colmap = [-1] * (9 * cp)
                for t in range(9):
                    for ch in range(ci):
                        colmap[t * cp + ch] = ch * 9 + t     # (cout, cin, kh, kw) rows -> (kh, kw, c) columns
                scale, shift = bn_affine(p + "bn%d." % j)
                w = sd[p + "conv%d.weight" % j].float().reshape(c, ci * 9) * scale[:, None]
                P[p + "conv%d" % j] = PackedLinear(w, shift, list(range(c)), colmap, need_grad=False)
            cin = c
        self._packed, self._packed_ver = P, ver
        return P

    def forward(self, input, mixup_lambda=None):
        """input: (batch_size, data_length) waveform at cfg['sample_rate'] on the GPU -> {"logits", "2048",
        "clipwise_output"} fp32, the reference's output dictionary (models.py:315-321)."""
        if self.training:
            raise N.CttaError("Cnn14 is built for evaluation only (the reference calls .eval(), eval.py:83): SpecAugment, "
                              "mixup and dropout of the training mode are not implemented")
        wav = input.contiguous().float()
        _check_cuda(wav, "waveform")
        P = self._pack()
        c = self.cfg
        B, L = wav.shape
        L_ = N.lib()
        s = N.stream_ptr()
        frames, F = L // c["hop"] + 1, c["mel_bins"]
        if frames < 32 or F < 32:
            raise ValueError("%d frames x %d mel bins: five 2x2 poolings need at least 32 of each" % (frames, F))
        lm = torch.empty(B, frames, F, dtype=torch.float32, device=wav.device)
        N.check(L_.ctta_wav_to_logmel_db(self._frontend(B, L), N.ptr(wav), B, L, 1e-10, N.ptr(lm), s))
        x = torch.empty(B, frames, F, 8, dtype=torch.bfloat16, device=wav.device)
        N.check(L_.ctta_logmel_to_image(N.ptr(lm), B, frames, F, N.ptr(P["bn0"][0]), N.ptr(P["bn0"][1]), N.ptr(x), s))
        H, W, cp = frames, F, 8
        for i, width in enumerate(c["widths"]):
            p = "conv_block%d." % (i + 1)
            for j in (1, 2):
                W_ = P[p + "conv%d" % j]
                y = torch.empty(B, H, W, width, dtype=torch.bfloat16, device=wav.device)
                _conv(_desc(x0=x, c0=cp, batch=B, hi=H, wi=W, ho=H, wo=W, kh=3, kw=3, pad_h=1, pad_w=1, w=W_.w,
                            k_pad=W_.k_pad, n=W_.n, bias=W_.bias, out=y, ldc=width, out_act=3, out_slope=0.0))   # ReLU
                x, cp = y, width
            if i < len(c["widths"]) - 1:      # pool_size (2, 2); conv_block6 pools (1, 1) = identity (models.py:302)
                y = torch.empty(B, H // 2, W // 2, width, dtype=torch.bfloat16, device=wav.device)
                N.check(L_.ctta_avgpool2(N.ptr(x), N.ptr(y), B, H, W, width, s))
                x, H, W = y, H // 2, W // 2
        pooled = torch.empty(B, cp, dtype=torch.float32, device=wav.device)
        N.check(L_.ctta_cnn14_head(N.ptr(x), B, H, W, cp, N.ptr(pooled), s))
        sd = dict(self.named_parameters())
        emb = torch.empty(B, cp, dtype=torch.float32, device=wav.device)
        logits = torch.empty(B, c["classes_num"], dtype=torch.float32, device=wav.device)
        for r0 in range(0, B, 1024):          # ctta_linear_f32 takes <= 1024 rows
            r1 = min(B, r0 + 1024)
            N.check(L_.ctta_linear_f32(N.ptr(pooled[r0:r1]), N.ptr(sd["fc1.weight"]), N.ptr(sd["fc1.bias"]), N.ptr(emb[r0:r1]),
                                       r1 - r0, cp, cp, 0, 0, s))
        emb.clamp_(min=0)                     # F.relu_ (models.py:313)
        for r0 in range(0, B, 1024):
            r1 = min(B, r0 + 1024)
            N.check(L_.ctta_linear_f32(N.ptr(emb[r0:r1]), N.ptr(sd["fc_audioset.weight"]), N.ptr(sd["fc_audioset.bias"]),
                                       N.ptr(logits[r0:r1]), r1 - r0, c["classes_num"], cp, 0, 0, s))
        return {"logits": logits, "2048": emb, "clipwise_output": torch.sigmoid(logits)}


# ------------------------------------------------------------------------------------------------ metrics
def calculate_fid(featuresdict_1, featuresdict_2, feat_layer_name):
    """metrics/fid.py:7-67: Frechet distance between the Gaussians fitted to two (N, D) feature sets."""
    eps = 1e-6
    features_1, features_2 = featuresdict_1[feat_layer_name], featuresdict_2[feat_layer_name]
    assert torch.is_tensor(features_1) and features_1.dim() == 2
    assert torch.is_tensor(features_2) and features_2.dim() == 2
    f1, f2 = features_1.cpu().numpy(), features_2.cpu().numpy()
    mu1, sigma1 = np.atleast_1d(np.mean(f1, axis=0)), np.atleast_2d(np.cov(f1, rowvar=False))
    mu2, sigma2 = np.atleast_1d(np.mean(f2, axis=0)), np.atleast_2d(np.cov(f2, rowvar=False))
    assert mu1.shape == mu2.shape, "Training and test mean vectors have different lengths"
    assert sigma1.shape == sigma2.shape, "Training and test covariances have different dimensions"
    diff = mu1 - mu2
    covmean, _ = scipy.linalg.sqrtm(sigma1.dot(sigma2), disp=False)        # the product might be almost singular
    if not np.isfinite(covmean).all():
        print("WARNING: fid calculation produces singular product; adding %g to the covariance diagonal" % eps)
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = scipy.linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):                                           # numerical error: slight imaginary component
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise AssertionError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    fid = diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)
    return {"frechet_distance": float(fid)}


def calculate_isc(featuresdict, feat_layer_name, rng_seed, samples_shuffle, splits):
    """metrics/isc.py:5-32: inception score of (N, C) logits over `splits` consecutive chunks (float64)."""
    features = featuresdict[feat_layer_name]
    assert torch.is_tensor(features) and features.dim() == 2
    n = features.shape[0]
    features = features.cpu()
    if samples_shuffle:
        rng = np.random.RandomState(rng_seed)
        features = features[rng.permutation(n), :]
    features = features.double()
    p, log_p = features.softmax(dim=1), features.log_softmax(dim=1)
    scores = []
    for i in range(splits):
        lo, hi = i * n // splits, (i + 1) * n // splits
        p_chunk, log_p_chunk = p[lo:hi, :], log_p[lo:hi, :]
        q_chunk = p_chunk.mean(dim=0, keepdim=True)
        kl = p_chunk * (log_p_chunk - q_chunk.log())
        scores.append(kl.sum(dim=1).mean().exp().item())
    return {"inception_score_mean": float(np.mean(scores)), "inception_score_std": float(np.std(scores))}


def polynomial_kernel(X, Y, degree=3, gamma=None, coef0=1):
    if gamma in [None, "none", "null", "None"]:
        gamma = 1.0 / X.shape[1]
    return (np.matmul(X, Y.T) * gamma + coef0) ** degree


def polynomial_mmd(features_1, features_2, degree, gamma, coef0):
    """metrics/kid.py:68-104: unbiased MMD^2 estimate under the polynomial kernel."""
    k_xx = polynomial_kernel(features_1, features_1, degree=degree, gamma=gamma, coef0=coef0)
    k_yy = polynomial_kernel(features_2, features_2, degree=degree, gamma=gamma, coef0=coef0)
    k_xy = polynomial_kernel(features_1, features_2, degree=degree, gamma=gamma, coef0=coef0)
    m = k_xx.shape[0]
    assert k_xx.shape == (m, m) and k_xy.shape == (m, m) and k_yy.shape == (m, m)
    kt_xx_sum = (k_xx.sum(axis=1) - np.diagonal(k_xx)).sum()
    kt_yy_sum = (k_yy.sum(axis=1) - np.diagonal(k_yy)).sum()
    k_xy_sum = k_xy.sum(axis=0).sum()
    mmd2 = (kt_xx_sum + kt_yy_sum) / (m * (m - 1))
    mmd2 -= 2 * k_xy_sum / (m * m)
    return mmd2


def calculate_kid(featuresdict_1, featuresdict_2, subsets, subset_size, degree, gamma, coef0, rng_seed, feat_layer_name):
    """metrics/kid.py:8-60: kernel inception distance over `subsets` random subsets drawn without replacement."""
    features_1, features_2 = featuresdict_1[feat_layer_name], featuresdict_2[feat_layer_name]
    assert torch.is_tensor(features_1) and features_1.dim() == 2
    assert torch.is_tensor(features_2) and features_2.dim() == 2
    assert features_1.shape[1] == features_2.shape[1]
    for other in (features_2, features_1):
        if subset_size > len(other):
            print("WARNING: subset size (%d) is larger than feature length (%d). Using %d for both datasets"
                  % (subset_size, len(other), len(other)))
            subset_size = len(other)
    features_1, features_2 = features_1.cpu().numpy(), features_2.cpu().numpy()
    mmds = np.zeros(subsets)
    rng = np.random.RandomState(rng_seed)
    for i in range(subsets):
        f1 = features_1[rng.choice(len(features_1), subset_size, replace=False)]
        f2 = features_2[rng.choice(len(features_2), subset_size, replace=False)]
        mmds[i] = polynomial_mmd(f1, f2, degree, gamma, coef0)
    return {"kernel_inception_distance_mean": float(np.mean(mmds)), "kernel_inception_distance_std": float(np.std(mmds))}


def calculate_kl(featuresdict_1, featuresdict_2, feat_layer_name, same_name=True):
    """metrics/kl.py:36-114: KL(target || prediction) of the classifier's distributions, files paired by base name;
    returns (metrics, per-file reference KL, paths of the first set) like the reference."""
    if not same_name:
        return ({"kullback_leibler_divergence_sigmoid": float(-1), "kullback_leibler_divergence_softmax": float(-1)},
                None, None)
    EPS = 1e-6
    features_1, features_2 = featuresdict_1[feat_layer_name].cpu(), featuresdict_2[feat_layer_name].cpu()
    paths_1 = [os.path.basename(x) for x in featuresdict_1["file_path_"]]
    paths_2 = [os.path.basename(x) for x in featuresdict_2["file_path_"]]
    key_to_feats_1 = {p: f for p, f in zip(paths_1, features_1)}
    key_to_feats_2 = {p: f for p, f in zip(paths_2, features_2)}
    f1, f2 = [], []
    for key, feat_2 in key_to_feats_2.items():
        if key not in key_to_feats_1:
            print("%s is not in the generation result" % key)
            continue
        f1.append(key_to_feats_1[key])
        f2.append(feat_2)
    features_1, features_2 = torch.stack(f1, dim=0), torch.stack(f2, dim=0)
    kl_div = torch.nn.functional.kl_div
    kl_ref = kl_div((features_1.softmax(dim=1) + EPS).log(), features_2.softmax(dim=1), reduction="none") / len(features_1)
    kl_ref = torch.mean(kl_ref, dim=-1)
    kl_softmax = kl_div((features_1.softmax(dim=1) + EPS).log(), features_2.softmax(dim=1), reduction="sum") / len(features_1)
    kl_sigmoid = kl_div((features_1.sigmoid() + EPS).log(), features_2.sigmoid(), reduction="sum") / len(features_1)
    return ({"kullback_leibler_divergence_sigmoid": float(kl_sigmoid),
             "kullback_leibler_divergence_softmax": float(kl_softmax)}, kl_ref, paths_1)


# ------------------------------------------------------------------------------------------------ files
def pad_short_audio(audio, min_samples=32000):
    """datasets/load_mel.py:9-14."""
    if audio.shape[-1] < min_samples:
        audio = torch.nn.functional.pad(audio, (0, min_samples - audio.shape[-1]), mode="constant", value=0.0)
    return audio


def read_centered_wav(audio_file, target_sr):
    """datasets/load_mel.py:17-29 with scipy.io.wavfile in place of soundfile: first channel mix-down, integer-ratio
    decimation by plain striding (as the reference does), mean removed; PCM widths scaled to [-1, 1) like soundfile."""
    from scipy.io import wavfile
    orig_sr, audio = wavfile.read(audio_file)
    if audio.dtype.kind == "i":
        audio = audio.astype(np.float64) / float(2 ** (8 * audio.dtype.itemsize - 1))
    elif audio.dtype.kind == "u":                            # 8-bit PCM is unsigned
        audio = (audio.astype(np.float64) - 128.0) / 128.0
    else:
        audio = audio.astype(np.float64)
    if audio.ndim > 1:
        audio = audio.mean(axis=1)                           # librosa.to_mono
    if orig_sr != target_sr and orig_sr % target_sr == 0:
        audio = audio[..., ::(orig_sr // target_sr)]
    elif orig_sr != target_sr:
        raise N.CttaError("%s: %d Hz is not an integer multiple of %d Hz; the reference resamples such files with resampy "
                          "(kaiser_best), which is not rebuilt -- resample the directory first" % (audio_file, orig_sr, target_sr))
    return audio - audio.mean()


class WaveDataset:
    """datasets/load_mel.py:123-151: sorted .wav files of a directory -> (waveform (1, n) fp32, base name)."""

    def __init__(self, datadir, sr=16000, target_length=1000, limit_num=None):
        self.datalist = sorted(os.path.join(datadir, x) for x in os.listdir(datadir))
        self.datalist = [x for x in self.datalist if x.endswith(".wav")]
        if limit_num is not None:
            self.datalist = self.datalist[:limit_num]
        self.sr, self.target_length = sr, target_length

    def __len__(self):
        return len(self.datalist)

    def __getitem__(self, index):
        filename = self.datalist[index]
        audio = torch.from_numpy(read_centered_wav(filename, self.sr)).float()[None]
        audio = pad_short_audio(audio[..., :int(self.sr * self.target_length / 100)], min_samples=32000)
        if audio.shape[-1] < 1:
            raise ValueError("empty file %s" % filename)
        return audio, os.path.basename(filename)


# ------------------------------------------------------------------------------------------------ driver
class EvaluationHelper:
    """eval.py:58-349 for the metrics this build computes.  `clap_model`: a `consistencytta_amd.clap.CLAP_Module` (or None
    to skip the three CLAP scores); the reference constructs one from `ckpt/music_audioset_epoch_15_esc_90.14.pt`."""

    KEYS = ["frechet_distance", "frechet_audio_distance", "lsd", "psnr", "kullback_leibler_divergence_sigmoid",
            "kullback_leibler_divergence_softmax", "ssim", "ssim_stft", "inception_score_mean", "inception_score_std",
            "kernel_inception_distance_mean", "kernel_inception_distance_std", "gt_text_clap_score", "gen_text_clap_score",
            "gen_gt_clap_score"]

    def __init__(self, sampling_rate, device, backbone="cnn14", mel_model=None, clap_model=None):
        self.device, self.backbone, self.sampling_rate = device, backbone, sampling_rate
        if sampling_rate not in (16000, 32000):
            raise ValueError("We only support the evaluation on 16kHz and 32kHz sampling rates.")
        if mel_model is None:
            c = spec.CNN14_16K_CONFIG if sampling_rate == 16000 else spec.CNN14_32K_CONFIG
            mel_model = Cnn14(features_list=["2048", "logits"], sample_rate=c["sample_rate"], window_size=c["n_fft"],
                              hop_size=c["hop"], mel_bins=c["mel_bins"], fmin=c["fmin"], fmax=c["fmax"],
                              classes_num=c["classes_num"]).to(device)
        self.mel_model = mel_model.eval()
        self.clap_model = clap_model

    def file_init_check(self, dir):
        assert os.path.exists(dir), "The path does not exist %s" % dir
        assert len(os.listdir(dir)) > 1, "There is no files in %s" % dir

    def get_filename_intersection_ratio(self, dir1, dir2, threshold=0.99, limit_num=None):
        k1 = {os.path.basename(x) for x in os.listdir(dir1) if x.endswith(".wav")}
        k2 = {os.path.basename(x) for x in os.listdir(dir2) if x.endswith(".wav")}
        both = k1 & k2
        return len(both) / len(k1) > threshold and len(both) / len(k2) > threshold

    def get_featuresdict(self, dataloader):
        """eval.py:310-330: an iterable of (waveform (1, n) or (n,), file name) -> {"2048", "logits", "clipwise_output":
        (N, .) CPU tensors, "file_path_": names}.  Clips of equal length go through the classifier together."""
        items = [(w.reshape(-1).float(), name) for w, name in dataloader]
        feats = {}
        order = sorted(range(len(items)), key=lambda i: items[i][0].numel())
        i = 0
        with torch.no_grad():
            while i < len(order):
                n = items[order[i]][0].numel()
                j = i
                while j < len(order) and j - i < 32 and items[order[j]][0].numel() == n:
                    j += 1
                out = self.mel_model(torch.stack([items[order[k]][0] for k in range(i, j)]).to(self.device))
                for k in range(i, j):
                    feats[order[k]] = {key: v[k - i].cpu() for key, v in out.items()}
                i = j
        res = {key: torch.stack([feats[i][key] for i in range(len(items))]) for key in ("2048", "logits", "clipwise_output")}
        res["file_path_"] = [name for _, name in items]
        return res

    @staticmethod
    def captions_from_dataset_json(dataset_json_path):
        """{generated file name: caption} as `T2APairedDataset` pairs them (tools/t2a_dataset.py:79-87,118-119): line i of the
        json-lines file holds the caption of `output_<i>.wav`."""
        import json
        if not os.path.isfile(dataset_json_path):
            raise AssertionError("%s is not a file." % dataset_json_path)
        with open(dataset_json_path) as f:
            rows = [json.loads(line) for line in f if line.strip()]
        return {"output_%d.wav" % i: r["captions"] for i, r in enumerate(rows)}

    def calculate_metrics(self, dataset_json_path, generate_files_path, groundtruth_path, mel_path=None, same_name=True,
                          target_length=1000, limit_num=None, captions=None, subset_size=None):
        """eval.py:181-308 (same positional order) on two directories of identically named .wav files.  The captions of the
        CLAP scores come from `dataset_json_path` as in the reference (or from `captions`: {file name: text}); `mel_path`
        (pre-computed generated mels for the reference's optional mel metrics) is accepted and unused.  Returns the
        reference's dictionary, rounded to 4 digits; metrics whose third-party model is not rebuilt are NaN."""
        if captions is None and dataset_json_path is not None:
            captions = self.captions_from_dataset_json(dataset_json_path)
        gen_files = sorted(f for f in os.listdir(generate_files_path) if f.endswith(".wav"))
        gt_files = sorted(f for f in os.listdir(groundtruth_path) if f.endswith(".wav"))
        if gen_files != gt_files:
            raise ValueError("Generated and groundtruth diretories have different files.\nGenerated: %s;\nGround truth: %s."
                             % (gen_files, gt_files))
        sr = self.sampling_rate
        gen = WaveDataset(generate_files_path, sr, limit_num=limit_num, target_length=target_length)
        gt = WaveDataset(groundtruth_path, sr, limit_num=limit_num, target_length=1000)
        featuresdict_2 = self.get_featuresdict(gt[i] for i in range(len(gt)))
        featuresdict_1 = self.get_featuresdict(gen[i] for i in range(len(gen)))
        out = {}
        if self.clap_model is not None and captions is not None:
            out.update(self.clap_scores([gt[i] for i in range(len(gt))], [gen[i] for i in range(len(gen))], captions))
        metric_kl, _, _ = calculate_kl(featuresdict_1, featuresdict_2, "logits", same_name)
        out.update(metric_kl)
        out.update(calculate_isc(featuresdict_1, feat_layer_name="logits", splits=10, samples_shuffle=True, rng_seed=2020))
        out.update(calculate_kid(featuresdict_1, featuresdict_2, feat_layer_name="2048", degree=3, gamma=None, subsets=100,
                                 subset_size=len(gen) if subset_size is None else subset_size, coef0=1, rng_seed=2020))
        out.update(calculate_fid(featuresdict_1, featuresdict_2, feat_layer_name="2048"))
        return {key: round(out.get(key, float("nan")), 4) for key in self.KEYS}

    def _clap_wave(self, w, seconds=10.0):
        """What `T2APairedDataset` hands the CLAP tower (tools/t2a_dataset.py:111-125 -> tools/torch_tools.py:54-75): the clip at
        48 kHz, mean removed, scaled to peak 0.5, cut / zero-padded to the segment length, scaled again.  The reference
        resamples with `resampy` (kaiser_best), which is not in its tree; here the Kaiser-windowed sinc resampler of the CLAP loss
        (tools/losses.py:299-303 = `clap.Resampler`) does it -- a stated deviation of the filter, not of the pipeline."""
        from .clap import Resampler
        w = w.reshape(1, -1).float().to(self.device)
        if self.sampling_rate != 48000:
            if getattr(self, "_to48k", None) is None:
                self._to48k = Resampler(orig_freq=self.sampling_rate, new_freq=48000)
            w = self._to48k(w)
        w = w - w.mean()
        w = w / (w.abs().max() + 1e-8) / 2
        seg = int(round(seconds * 48000))
        w = w[:, :seg] if w.shape[1] >= seg else torch.nn.functional.pad(w, (0, seg - w.shape[1]))
        return w / (w.abs().max() + 1e-8) / 2

    def clap_scores(self, gt_items, gen_items, captions, seconds=10.0):
        """eval.py:29-55,238-253: clamped cosine similarities of CLAP embeddings (audio at 48 kHz as `_clap_wave` prepares
        it, captions through the text tower), x 100."""
        cos = torch.nn.functional.cosine_similarity
        sims = {"gt_text": [], "gen_text": [], "gen_gt": []}
        with torch.no_grad():
            for (gw, name), (xw, _) in zip(gt_items, gen_items):
                g = self.clap_model.get_audio_embedding_from_data(x=self._clap_wave(gw, seconds), use_tensor=True)
                x = self.clap_model.get_audio_embedding_from_data(x=self._clap_wave(xw, seconds), use_tensor=True)
                t = self.clap_model.get_text_embedding([captions[name]], use_tensor=True)
                sims["gt_text"].append(cos(g, t, dim=1).clamp(min=0))
                sims["gen_text"].append(cos(x, t, dim=1).clamp(min=0))
                sims["gen_gt"].append(cos(x, g, dim=1).clamp(min=0))
        return {k + "_clap_score": torch.cat(v).mean().item() * 100.0 for k, v in sims.items()}

    def main(self, dataset_json_path, generated_files_path, groundtruth_path, mel_path=None, target_length=1000,
             limit_num=None, captions=None):
        """eval.py:336-349, keyword- and position-compatible with the reference's callers (inference.py:230,
        evaluate_existing.py:54); `dataset_json_path=None` skips the CLAP scores unless `captions` is given."""
        self.file_init_check(generated_files_path)
        self.file_init_check(groundtruth_path)
        same_name = self.get_filename_intersection_ratio(generated_files_path, groundtruth_path, limit_num=limit_num)
        return self.calculate_metrics(dataset_json_path, generated_files_path, groundtruth_path, mel_path, same_name,
                                      target_length, limit_num, captions)
