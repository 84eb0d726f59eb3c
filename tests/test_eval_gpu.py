"""Evaluation suite on the HIP path (audioldm_eval/eval.py:181-308): the PANNs Cnn14 classifier against
the fixture the reference's own `Cnn14.forward` produced, its glue kernels against torch, and the metric driver end to end on
directories of .wav files against the oracle's features."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import cases  # noqa: E402
from consistencytta_amd import _native as N  # noqa: E402
from consistencytta_amd import audioldm_eval as E  # noqa: E402
from consistencytta_amd import spec  # noqa: E402
from gpu_util import DEV, bf16_round, det, rel_l2, sync  # noqa: E402
from oracle import evalsuite as oe  # noqa: E402

pytestmark = pytest.mark.gpu

# bf16 operands through twelve 3x3 convolutions (fp32 accumulation, bf16 activations between layers) against the fp32 reference
CNN14_REL_L2 = 2e-2


def lib():
    return N.lib()


def _model(g):
    m = E.Cnn14(features_list=["2048", "logits"]).to(DEV)
    sd = cases.cnn14_weights(g["cnn14_keys"], g["cnn14_shapes"])
    m.load_state_dict(sd, strict=True)                    # reference keys only: the structural front-end entries are optional
    return m.eval(), sd


def test_cnn14_glue_kernels():
    """ctta_logmel_to_image (bn0 + cast), ctta_avgpool2 (odd sizes: the trailing row / column is dropped like F.avg_pool2d) and
    ctta_cnn14_head (mean over frequency, max + mean over time) against torch on the same bf16-rounded values."""
    L_ = lib()
    s = N.stream_ptr()
    B, T, Fq = 2, 37, 64
    lm = det("ev.lm", (B, T, Fq), 1) * 40 - 50
    sc, sh = det("ev.sc", (Fq,), 2) * 0.02 + 0.1, det("ev.sh", (Fq,), 3)
    img = torch.empty(B, T, Fq, 8, dtype=torch.bfloat16, device=DEV)
    lm_d, sc_d, sh_d = lm.to(DEV), sc.to(DEV), sh.to(DEV)
    N.check(L_.ctta_logmel_to_image(N.ptr(lm_d), B, T, Fq, N.ptr(sc_d), N.ptr(sh_d), N.ptr(img), s))
    sync()
    assert torch.equal(img[..., 0].float().cpu(), bf16_round(lm * sc + sh)) and float(img[..., 1:].float().abs().max()) == 0.0
    for (H, W, C) in ((37, 64, 64), (1001, 64, 64), (12, 5, 2048), (2, 2, 8)):
        x = bf16_round(det("ev.pool", (B, C, H, W), 4))
        y = torch.empty(B, H // 2, W // 2, C, dtype=torch.bfloat16, device=DEV)
        x_d = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
        N.check(L_.ctta_avgpool2(N.ptr(x_d), N.ptr(y), B, H, W, C, s))
        sync()
        assert torch.equal(y.float().permute(0, 3, 1, 2).cpu(), bf16_round(F.avg_pool2d(x, 2))), (H, W, C)
    with pytest.raises(RuntimeError):
        N.check(L_.ctta_avgpool2(N.ptr(img), N.ptr(img), B, 4, 4, 12, s))
    for (T2, F2, C) in ((31, 2, 2048), (6, 2, 100), (1, 1, 8)):
        x = bf16_round(det("ev.head", (B, C, T2, F2), 5))
        y = torch.empty(B, C, device=DEV)
        x_d = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
        N.check(L_.ctta_cnn14_head(N.ptr(x_d), B, T2, F2, C, N.ptr(y), s))
        sync()
        m = x.mean(3)
        np.testing.assert_allclose(y.cpu().numpy(), (m.max(2)[0] + m.mean(2)).numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag,B,L", [("short", 2, 32000), ("clip", 1, 160000)])
def test_cnn14_matches_the_reference_module(golden, tag, B, L):
    """`Cnn14.forward` (panns/models.py:269-323) on the HIP path vs the reference's own module (fixture): the 2048-wide
    embedding, the 527 logits and the sigmoid outputs, 2 s clips in a batch and one 10 s clip (1001 frames: odd sizes at
    every pooling)."""
    g = golden("eval_suite")
    m, _ = _model(g)
    wav = cases.eval_waves("evalsuite." + tag, B, L).to(DEV)
    with torch.no_grad():
        out = m(wav)
    sync()
    assert set(out) == {"logits", "2048", "clipwise_output"}
    for key, ref in (("2048", g[tag + "_2048"]), ("logits", g[tag + "_logits"]), ("clipwise_output", g[tag + "_clipwise"])):
        assert rel_l2(out[key].cpu(), torch.from_numpy(ref)) < CNN14_REL_L2, key
    # the same clips one by one give the same rows (no cross-sample coupling in the batch)
    if B > 1:
        with torch.no_grad():
            one = m(wav[1:2])
        assert torch.equal(one["2048"][0], out["2048"][1])


def test_cnn14_rejects_what_it_does_not_build(golden):
    m, _ = _model(golden("eval_suite"))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 4000, device=DEV))               # 26 frames: fewer than the five poolings need
    m.train()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 32000, device=DEV))
    m.eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 32000))                          # a host tensor: there is no CPU path


def test_evaluation_helper_end_to_end(golden, tmp_path):
    """EvaluationHelper.main (eval.py:332-349) on two directories of int16 .wav files (16 kHz and 48 kHz, the latter decimated
    by striding as the reference does): features through the HIP classifier, metrics on the host; against the same metrics on
    the oracle's fp32 features of the same files."""
    from scipy.io import wavfile
    g = golden("eval_suite")
    m, sd = _model(g)
    gen_dir, gt_dir = tmp_path / "gen", tmp_path / "gt"
    gen_dir.mkdir()
    gt_dir.mkdir()
    n = 12
    gen = cases.eval_waves("evalsuite.e2e.gen", n, 32000).numpy()
    gt = cases.eval_waves("evalsuite.e2e.gt", n, 96000, sr=48000).numpy()
    for i in range(n):
        wavfile.write(str(gen_dir / ("clip_%02d.wav" % i)), 16000, (gen[i] * 32767).astype(np.int16))
        wavfile.write(str(gt_dir / ("clip_%02d.wav" % i)), 48000, (gt[i] * 32767).astype(np.int16))
    helper = E.EvaluationHelper(16000, DEV, mel_model=m)
    res = helper.main(None, str(gen_dir), str(gt_dir))
    assert list(res) == E.EvaluationHelper.KEYS
    for k in ("frechet_audio_distance", "lsd", "psnr", "ssim", "ssim_stft", "gt_text_clap_score"):
        assert np.isnan(res[k]), k                        # third-party models the build does not have: reported like a missing key
    # the same files through the oracle classifier
    ds_gen, ds_gt = E.WaveDataset(str(gen_dir), 16000), E.WaveDataset(str(gt_dir), 16000)
    names = [ds_gen[i][1] for i in range(n)]
    with torch.no_grad():
        o_gen = oe.cnn14_forward(spec.CNN14_16K_CONFIG, sd, torch.cat([ds_gen[i][0] for i in range(n)]))
        o_gt = oe.cnn14_forward(spec.CNN14_16K_CONFIG, sd, torch.cat([ds_gt[i][0] for i in range(n)]))
    f_gen = helper.get_featuresdict(ds_gen[i] for i in range(n))
    assert f_gen["file_path_"] == names and rel_l2(f_gen["2048"], o_gen["2048"]) < CNN14_REL_L2
    d1 = {"2048": o_gen["2048"], "logits": o_gen["logits"], "file_path_": names}
    d2 = {"2048": o_gt["2048"], "logits": o_gt["logits"], "file_path_": names}
    ref = {}
    ref.update(E.calculate_kl(d1, d2, "logits", True)[0])
    ref.update(E.calculate_isc(d1, feat_layer_name="logits", splits=10, samples_shuffle=True, rng_seed=2020))
    ref.update(E.calculate_kid(d1, d2, feat_layer_name="2048", degree=3, gamma=None, subsets=100, subset_size=n, coef0=1, rng_seed=2020))
    for k, v in ref.items():
        # metrics of 12 clips on features that differ by ~1e-2: the same order of agreement, absolute floor for the near-zero ones
        assert abs(res[k] - v) <= 5e-2 * abs(v) + 5e-3, (k, res[k], v)
    assert np.isfinite(res["frechet_distance"])           # 12 samples in 2048 dimensions: singular covariances, value not compared
    # directories that do not hold the same files are refused like the reference (eval.py:196-204)
    os.remove(str(gen_dir / "clip_00.wav"))
    with pytest.raises(ValueError):
        helper.main(None, str(gen_dir), str(gt_dir))


def test_clap_scores_feed_the_tower_what_the_reference_dataset_does(golden):
    """EvaluationHelper.clap_scores (eval.py:29-55,238-253): every clip reaches the CLAP tower as `read_wav_file` prepares it
    (tools/torch_tools.py:54-75: 48 kHz, mean removed, peak 0.5, 10 s segment, peak 0.5 again) and the three scores are the
    clamped cosines x 100; checked with a stand-in tower that records its inputs and embeds by simple statistics."""
    m, _ = _model(golden("eval_suite"))
    seen = []

    class Tower:
        def get_audio_embedding_from_data(self, x, use_tensor=False):
            assert use_tensor and x.is_cuda
            seen.append(x)
            f = x.reshape(1, 100, -1)
            return torch.nn.functional.normalize(f.abs().mean(2) + f.std(2), dim=-1)

        def get_text_embedding(self, texts, use_tensor=False):
            gen = torch.Generator().manual_seed(len(texts[0]))
            return torch.nn.functional.normalize(torch.rand(1, 100, generator=gen), dim=-1).to(DEV)

    helper = E.EvaluationHelper(16000, DEV, mel_model=m, clap_model=Tower())
    gt = [(cases.eval_waves("evalsuite.cs.gt", 1, 32000)[0:1], "a.wav"), (cases.eval_waves("evalsuite.cs.gt2", 1, 200000)[0:1], "b.wav")]
    gen = [(cases.eval_waves("evalsuite.cs.gen", 1, 32000)[0:1], "a.wav"), (cases.eval_waves("evalsuite.cs.gen2", 1, 160000)[0:1], "b.wav")]
    res = helper.clap_scores(gt, gen, {"a.wav": "a dog barks", "b.wav": "rain on a roof"})
    assert set(res) == {"gt_text_clap_score", "gen_text_clap_score", "gen_gt_clap_score"}
    assert all(0.0 <= v <= 100.0 for v in res.values()) and res["gen_gt_clap_score"] > 50.0
    assert len(seen) == 4
    for x in seen:
        assert tuple(x.shape) == (1, 480000) and abs(float(x.abs().max()) - 0.5) < 1e-4
    # the 2 s clips are zero-padded behind 96 000 samples, the 12.5 s clip is cut at 10 s
    assert float(seen[0][0, 96100:].abs().max()) == 0.0 and float(seen[2][0, -100:].abs().max()) > 0.0
    # against a direct evaluation of the same recipe
    w = helper._clap_wave(gen[0][0])
    ref = torch.nn.functional.cosine_similarity(Tower().get_audio_embedding_from_data(w, True),
                                                Tower().get_text_embedding(["a dog barks"], True), dim=1).clamp(min=0)
    w2 = helper._clap_wave(gen[1][0])
    ref2 = torch.nn.functional.cosine_similarity(Tower().get_audio_embedding_from_data(w2, True),
                                                 Tower().get_text_embedding(["rain on a roof"], True), dim=1).clamp(min=0)
    assert abs(res["gen_text_clap_score"] - float((ref + ref2) / 2) * 100.0) < 1e-3
