"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

fp32 PyTorch-CPU / numpy restatement of the reference's evaluation suite (audioldm_eval/):

  * `cnn14_forward`: PANNs Cnn14 in eval mode, audioldm_eval/feature_extractors/panns/models.py:30-80 (ConvBlock) and
    :269-323 (forward): torchlibrosa Spectrogram + LogmelFilterBank (restated in oracle/clap.py, shared with the HTSAT
    tower) -> bn0 over mel bins -> six ConvBlocks -> mean over frequency, max + mean over time -> fc1, ReLU -> fc_audioset.
  * `fid`, `isc`, `kid`, `kl`: audioldm_eval/metrics/{fid,isc,kid,kl}.py in plain numpy (float64 where the reference is).

Pinned by tests/golden/eval_suite.npz: the reference's OWN `Cnn14.forward` (its front end bound to the restatement above,
torchlibrosa being absent) and the reference's OWN four metric functions on seeded features
(tests/golden/make_golden_eval.py).
"""
import numpy as np
import scipy.linalg
import torch
import torch.nn.functional as F

from . import clap as oclap


def cnn14_forward(cfg, sd, wav, taps=None):
    """cfg: sample_rate, n_fft, hop, mel_bins, fmin, fmax, widths; sd: reference key names -> fp32 tensors; wav (B, L)."""
    x = oclap.logmel(oclap.spectrogram_power(wav, cfg["n_fft"], cfg["hop"], cfg["n_fft"]), cfg["sample_rate"], cfg["n_fft"],
                     cfg["mel_bins"], cfg["fmin"], cfg["fmax"])                       # (B, 1, T, mel)
    x = F.batch_norm(x.transpose(1, 3), sd["bn0.running_mean"], sd["bn0.running_var"], sd["bn0.weight"], sd["bn0.bias"],
                     False, 0.0, 1e-5).transpose(1, 3)                                # models.py:276-278
    n = len(cfg["widths"])
    for i in range(n):
        p = "conv_block%d." % (i + 1)
        for j in (1, 2):                                                              # models.py:65-66
            x = F.conv2d(x, sd[p + "conv%d.weight" % j], None, 1, 1)
            b = p + "bn%d." % j
            x = F.relu(F.batch_norm(x, sd[b + "running_mean"], sd[b + "running_var"], sd[b + "weight"], sd[b + "bias"],
                                    False, 0.0, 1e-5))
        if i < n - 1:
            x = F.avg_pool2d(x, kernel_size=(2, 2))                                   # block 6 pools (1, 1)
        if taps is not None:
            taps["block%d" % (i + 1)] = x
    x = torch.mean(x, dim=3)
    x = torch.max(x, dim=2)[0] + torch.mean(x, dim=2)                                 # models.py:305-309
    emb = F.relu(F.linear(x, sd["fc1.weight"], sd["fc1.bias"]))
    logits = F.linear(emb, sd["fc_audioset.weight"], sd["fc_audioset.bias"])
    return {"2048": emb, "logits": logits, "clipwise_output": torch.sigmoid(logits)}


def fid(f1, f2):
    """metrics/fid.py: f1, f2 (N, D) float arrays -> Frechet distance."""
    mu1, mu2 = f1.mean(0), f2.mean(0)
    s1, s2 = np.cov(f1, rowvar=False), np.cov(f2, rowvar=False)
    covmean = scipy.linalg.sqrtm(s1.dot(s2), disp=False)[0]
    if not np.isfinite(covmean).all():
        off = np.eye(s1.shape[0]) * 1e-6
        covmean = scipy.linalg.sqrtm((s1 + off).dot(s2 + off))
    covmean = covmean.real
    d = mu1 - mu2
    return float(d.dot(d) + np.trace(s1) + np.trace(s2) - 2.0 * np.trace(covmean))


def _log_softmax(x):
    x = x - x.max(1, keepdims=True)
    return x - np.log(np.exp(x).sum(1, keepdims=True))


def isc(logits, splits=10, rng_seed=2020, shuffle=True):
    """metrics/isc.py: (mean, std) of exp(mean_i KL(p_i || mean_j p_j)) over `splits` consecutive chunks."""
    n = logits.shape[0]
    x = logits[np.random.RandomState(rng_seed).permutation(n)] if shuffle else logits
    lp = _log_softmax(x.astype(np.float64))
    p = np.exp(lp)
    scores = []
    for i in range(splits):
        sl = slice(i * n // splits, (i + 1) * n // splits)
        q = p[sl].mean(0, keepdims=True)
        scores.append(float(np.exp((p[sl] * (lp[sl] - np.log(q))).sum(1).mean())))
    return float(np.mean(scores)), float(np.std(scores))


def kid(f1, f2, subsets=100, subset_size=None, degree=3, coef0=1, rng_seed=2020):
    """metrics/kid.py: (mean, std) of the unbiased polynomial-kernel MMD^2 over random subsets."""
    m = min(len(f1), len(f2)) if subset_size is None else min(subset_size, len(f1), len(f2))
    gamma = 1.0 / f1.shape[1]
    rng = np.random.RandomState(rng_seed)
    out = np.zeros(subsets)
    for i in range(subsets):
        a = f1[rng.choice(len(f1), m, replace=False)]
        b = f2[rng.choice(len(f2), m, replace=False)]
        kxx, kyy, kxy = [(np.matmul(u, v.T) * gamma + coef0) ** degree for u, v in ((a, a), (b, b), (a, b))]
        out[i] = ((kxx.sum() - np.trace(kxx)) + (kyy.sum() - np.trace(kyy))) / (m * (m - 1)) - 2.0 * kxy.sum() / (m * m)
    return float(out.mean()), float(out.std())


def kl(logits_pred, logits_target):
    """metrics/kl.py on already paired rows: (sigmoid form, softmax form) of KL(target || prediction + 1e-6) / N, in the
    fp32 arithmetic of torch.nn.functional.kl_div (target * (log target - input), 0 where target == 0)."""
    eps = np.float32(1e-6)
    a, b = logits_pred.astype(np.float32), logits_target.astype(np.float32)

    def kldiv(inp_log, tgt):
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where(tgt > 0, tgt * (np.log(tgt) - inp_log), np.float32(0))
        return float(t.astype(np.float32).sum(dtype=np.float64))

    sm = lambda x: np.exp(_log_softmax(x.astype(np.float64))).astype(np.float32)
    sg = lambda x: (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)
    n = len(a)
    return kldiv(np.log(sg(a) + eps), sg(b)) / n, kldiv(np.log(sm(a) + eps), sm(b)) / n
