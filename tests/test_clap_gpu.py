"""CLAP fine-tuning stage on the HIP path (tools/losses.py:259-316): operator parity of the new
kernels against the CPU oracle, the HTSAT audio tower (forward AND input gradient) against fixtures produced by the
reference's own laion_clap/clap_module/htsat.py, the RoBERTa text tower against transformers' RobertaModel, and the
CLAP loss end to end (latent -> VAE decoder -> HiFi-GAN -> resampler -> tower -> cosine terms) against the oracle.

Tolerances: bf16 activations with fp32 accumulation against fp32 references -- relative L2 2.5e-2 on embeddings,
5e-2 on input gradients (24 residual blocks deep); fp32 kernels (resampler, front end) a few 1e-4."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import cases  # noqa: E402
from consistencytta_amd import _native as N  # noqa: E402
from consistencytta_amd import clap as C  # noqa: E402
from consistencytta_amd import spec  # noqa: E402
from gpu_util import DEV, bf16_round, det, rel_err, rel_l2, sync  # noqa: E402
from oracle import clap as oclap  # noqa: E402

KAISER = dict(lowpass_filter_width=64, rolloff=0.9475937167399596, beta=14.769656459379492)
# Sampled-gradient budget of the real-size CLAP fine-tuning step (per block and overall).  Measured on MI355X: loss 0.547995
# vs the reference's 0.547909, sampled gradient rel-L2 5.1e-3 over all blocks, worst block 9.0e-3, worst per-tensor norm
# deviation 2.2e-3 -- the MSE term dominates the gradient at this size, so the vocoder's mask flips (LABNOTES.md 4b: 0.13..0.19 on
# the vocoder's own input gradient) stay below the bf16 noise of the U-Net backward.  Budget = the distillation step's.
CLAPFT_GRAD_REL_L2 = 4e-2


# ------------------------------------------------------------------------------------------------ operators
@pytest.mark.parametrize("orig,new,L", [(16000, 48000, 5000), (16000, 48000, 333), (48000, 16000, 4001), (2, 3, 1000)])
def test_resampler_forward_and_adjoint(orig, new, L):
    x = (det("rs.x", (2, L), 1) * 0.8).requires_grad_(True)
    ref = oclap.resample(x, orig, new, **KAISER)
    g = det("rs.g", tuple(ref.shape), 2)
    (ref * g).sum().backward()
    rs = C.Resampler(orig, new, **KAISER)
    xd = x.detach().to(DEV).requires_grad_(True)
    y = rs(xd)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < 2e-5
    (y * g.to(DEV)).sum().backward()
    assert rel_err(xd.grad, x.grad) < 2e-5
    with pytest.raises(RuntimeError):
        rs(x.detach())                                     # CPU tensor: no CPU path


def test_gather_gelu_mean_tokens():
    L_ = N.lib()
    x = bf16_round(det("g.x", (37, 64), 1) * 3).to(torch.bfloat16).to(DEV)
    idx = torch.randperm(37, generator=torch.Generator().manual_seed(1)).to(torch.int32)
    idx[5] = -1
    out = C._gather(x, idx.to(DEV))
    ref = x.cpu()[idx.clamp(min=0).long()]
    ref[5] = 0
    assert torch.equal(out.cpu(), ref)
    xf = x.float().cpu().requires_grad_(True)
    yr = F.gelu(xf)
    gy = bf16_round(det("g.gy", (37, 64), 2))
    (yr * gy).sum().backward()
    xg = x.clone().requires_grad_(True)
    y = C._Gelu.apply(xg)
    assert rel_err(y, yr) < 2.0 ** -8
    (y.float() * gy.to(DEV)).sum().backward()
    assert rel_err(xg.grad, xf.grad) < 2 * 2.0 ** -8
    t = bf16_round(det("g.t", (3 * 16, 128), 3)).to(torch.bfloat16).to(DEV).requires_grad_(True)
    m = C._MeanTokens.apply(t, 3, 16, 128)
    assert rel_err(m, t.detach().float().cpu().view(3, 16, 128).mean(1)) < 1e-6
    m.sum().backward()
    assert rel_err(t.grad, torch.full((48, 128), 1.0 / 16)) < 2.0 ** -8


@pytest.mark.parametrize("L", [28800, 9000])
def test_htsat_front_end_against_oracle(L):
    """wav -> power STFT -> dB log-mel -> bn0 -> bicubic frame stretch -> fold, and its input gradient, vs the oracle
    restatement (itself bit-identical to the reference module's own front half, tests/test_oracle_golden.py)."""
    cfg = cases.TINY_HTSAT
    m = C.HTSAT(cfg)
    g = np.load(cases.__file__.replace("cases.py", "clap_htsat.npz"))
    sd = cases.clap_weights(g["tiny_keys"], g["tiny_shapes"], "tiny", 5)
    m.load_state_dict(sd, strict=False)
    m.to(DEV)
    wav = (det("fe.wav", (2, L), 7) * 0.4)
    with torch.no_grad():
        ref = oclap.wav_to_image(cfg, sd, wav, prefix="")                 # (B, 1, 64, 64)
    P = m._pack()
    wd = wav.to(DEV).requires_grad_(True)
    tok = C._Frontend.apply(wd, m._fe, P["patch"], P["patch_t"])
    img = m._fe.last_image[..., 0].float().cpu()
    assert rel_err(img, ref[:, 0]) < 3 * 2.0 ** -8                       # bf16 image of O(1) values
    # patch tokens vs conv2d on the bf16 image
    tref = F.conv2d(bf16_round(ref.detach()), bf16_round(sd["patch_embed.proj.weight"]), sd["patch_embed.proj.bias"], stride=4)
    assert rel_err(tok.float().cpu().view(2, 16, 16, -1).permute(0, 3, 1, 2), tref) < 3 * 2.0 ** -8
    (tok.float() * 0).sum().backward()                                     # exercises the plumbing (zero gradient)
    assert float(wd.grad.abs().max()) == 0.0


def test_window_attention_fullbias_forward_backward():
    """Swin window attention (htsat.py:336-361) on the flash kernels with a full bias table vs plain torch."""
    heads, hd, n, nw, nbb = 4, 32, 64, 8, 4
    hp = heads * 64
    qkv = torch.zeros(nw * n, 3 * hp)
    raw = bf16_round(det("wa.qkv", (nw * n, 3, heads, hd), 1) * 1.5)
    for part in range(3):
        for h in range(heads):
            qkv[:, part * hp + h * 64: part * hp + h * 64 + hd] = raw[:, part, h]
    bias = det("wa.bias", (nbb, heads, n, n), 2) * 2.0
    bias[1, :, :, 40:] = -100.0                                              # a shifted-window style mask
    scale = hd ** -0.5
    q, k, v = (raw[:, i].view(nw, n, heads, hd).permute(0, 2, 1, 3).clone().requires_grad_(True) for i in range(3))
    att = (q * scale) @ k.transpose(-1, -2) + bias[torch.arange(nw) % nbb]
    o_ref = att.softmax(-1) @ v                                              # (nw, heads, n, hd)
    go = bf16_round(det("wa.go", tuple(o_ref.shape), 3))
    (o_ref * go).sum().backward()
    qd = qkv.to(torch.bfloat16).to(DEV).requires_grad_(True)
    out = C._WindowAttention.apply(qd, (bias * C.LOG2E).contiguous().to(DEV), nbb, heads, n, scale)
    got = out.float().cpu().view(nw, n, heads, 64)[..., :hd].permute(0, 2, 1, 3)
    assert rel_err(got, o_ref) < 3 * 2.0 ** -8
    assert float(out.float().cpu().view(nw, n, heads, 64)[..., hd:].abs().max()) == 0.0     # pad lanes stay zero
    gpad = torch.zeros(nw * n, hp)
    gp = go.permute(0, 2, 1, 3).reshape(nw * n, heads, hd)
    for h in range(heads):
        gpad[:, h * 64: h * 64 + hd] = gp[:, h]
    out.backward(gpad.to(torch.bfloat16).to(DEV))
    dq = qd.grad.float().cpu()
    for part, ref in enumerate((q.grad, k.grad, v.grad)):
        got = dq[:, part * hp:(part + 1) * hp].view(nw, n, heads, 64)[..., :hd].permute(0, 2, 1, 3)
        assert rel_l2(got, ref) < 2e-2, part


# ------------------------------------------------------------------------------------------------ towers
def _tower(cfg, keys, shapes, tag, seed):
    m = C.HTSAT(cfg)
    sd = cases.clap_weights(keys, shapes, tag, seed)
    missing = m.load_state_dict(sd, strict=False)
    assert all(k.startswith(("spectrogram_extractor", "logmel_extractor")) for k in missing.missing_keys), missing
    assert not missing.unexpected_keys
    return m.to(DEV).eval(), sd


def test_htsat_tiny_tower_embedding_taps_and_input_gradient(golden):
    g = golden("clap_htsat")
    cfg = cases.TINY_HTSAT
    m, sd = _tower(cfg, g["tiny_keys"], g["tiny_shapes"], "tiny", 5)
    wav = cases.t(spec.det_uniform("clap.tiny.wav", (2, 28800), 3)) * 0.4
    otaps = {}
    with torch.no_grad():
        oclap.htsat_embedding(cfg, sd, wav, prefix="", taps=otaps)
    wd = wav.to(DEV).requires_grad_(True)
    taps = {}
    emb = m(wd, taps=taps)
    assert rel_err(taps["image"], torch.from_numpy(g["tiny_image"])[:, 0]) < 3 * 2.0 ** -8
    for i in range(4):
        l2 = rel_l2(taps["layer%d" % i], otaps["layer%d" % i])
        print("tiny tower layer %d rel_l2 %.3e" % (i, l2))
        assert l2 < 2.5e-2
    l2 = rel_l2(emb, torch.from_numpy(g["tiny_embedding"]))
    print("tiny tower embedding rel_l2 %.3e" % l2)
    assert l2 < 2.5e-2
    direction = cases.t(spec.det_uniform("tiny.dir", tuple(emb.shape), 9)).to(DEV)
    (emb * direction).sum().backward()
    gl2 = rel_l2(wd.grad, torch.from_numpy(g["tiny_grad"]))
    print("tiny tower d/d wav rel_l2 %.3e" % gl2)
    assert gl2 < 5e-2
    # a plain (no-grad) pass between a differentiable forward and its backward must not disturb it
    wd2 = wav.to(DEV).requires_grad_(True)
    e2 = m(wd2)
    with torch.no_grad():
        m(wav.to(DEV) * 0.5)
    (e2 * direction).sum().backward()
    assert torch.equal(wd2.grad, wd.grad)


def test_htsat_base_tower_on_ten_seconds(golden):
    """HTSAT-base (72 M parameters, 24 Swin blocks) on 10 s at 48 kHz: the shape CLAPLoss runs (tools/losses.py:305)."""
    g = golden("clap_htsat")
    m, sd = _tower(spec.HTSAT_BASE_CONFIG, g["base_keys"], g["base_shapes"], "base", 6)
    wav = cases.t(spec.det_uniform("clap.base.wav", (2, 480000), 4)) * 0.4
    wd = wav.to(DEV).requires_grad_(True)
    emb = m(wd)
    l2 = rel_l2(emb, torch.from_numpy(g["base_embedding"]))
    print("HTSAT-base embedding rel_l2 %.3e" % l2)
    assert l2 < 2.5e-2
    direction = cases.t(spec.det_uniform("base.dir", tuple(emb.shape), 9)).to(DEV)
    (emb * direction).sum().backward()
    gn = float(wd.grad.double().norm())
    idx = torch.from_numpy(cases.sample_index(480000, 4096)).to(DEV)
    gs = rel_l2(wd.grad[:, idx], torch.from_numpy(g["base_grad_sample"]))
    print("HTSAT-base d/d wav: norm %.4e (ref %.4e), sampled rel_l2 %.3e" % (gn, float(g["base_grad_norm"]), gs))
    assert abs(gn - float(g["base_grad_norm"])) <= 5e-2 * float(g["base_grad_norm"]) and gs < 6e-2


@pytest.mark.parametrize("tag,cfg", [("tiny", cases.TINY_ROBERTA), ("wide", cases.WIDE_ROBERTA)])
def test_roberta_text_tower_matches_transformers(golden, tag, cfg):
    g = golden("roberta")
    m = C.RobertaModel(cfg)
    m.load_state_dict(cases.roberta_weights(cfg))
    m.to(DEV).eval()
    ids, mask = cases.roberta_inputs(cfg, 3, 20, tag)
    out = m(input_ids=ids.to(DEV), attention_mask=mask.to(DEV))
    valid = mask.bool()
    l2 = rel_l2(out["last_hidden_state"].cpu()[valid], torch.from_numpy(g[tag + "_last"])[valid])
    p2 = rel_l2(out["pooler_output"], torch.from_numpy(g[tag + "_pooler"]))
    print("RoBERTa %s: last_hidden_state rel_l2 %.3e, pooler_output rel_l2 %.3e" % (tag, l2, p2))
    assert l2 < 2.5e-2 and p2 < 2.5e-2


def test_clap_loss_end_to_end_matches_oracle():
    """CLAPLoss (tools/losses.py:276-316): latent -> VAE decoder -> HiFi-GAN (allow_grad) -> 16 -> 48 kHz -> CLAP audio
    embedding; cosine terms against text / ground-truth-audio embeddings; value and latent gradient vs the oracle chain
    (oracle VAE / vocoder / resampler / tower under torch autograd)."""
    from consistencytta_amd import losses, modules
    from oracle import nets as onets
    g = np.load(cases.__file__.replace("cases.py", "clap_htsat.npz"))
    cfg = cases.TINY_HTSAT
    clap = C.CLAP_Module(audio_cfg=cfg, text_cfg=cases.TINY_ROBERTA)
    hsd = cases.clap_weights(g["tiny_keys"], g["tiny_shapes"], "tiny", 5)
    clap.model.audio_branch.load_state_dict(hsd, strict=False)
    proj = {k: cases.t(spec.clap_det_weight("clap." + k, tuple(v.shape), 1)) for k, v in clap.model.state_dict().items()
            if k.startswith("audio_projection")}
    clap.model.load_state_dict(proj)
    vsd = dict(cases.vae_weights(cases.TINY_VAE_DD))
    vsd.update(cases.hifigan_weights(cases.TINY_HIFIGAN))
    vae = modules.AutoencoderKL(ddconfig=cases.TINY_VAE_DD, embed_dim=8, scale_factor=0.9, hifigan_config=cases.TINY_HIFIGAN)
    vae.load_state_dict(vsd)
    vae.to(DEV).eval().requires_grad_(False)
    loss = losses.CLAPLoss(vae, reduction="instance", mse_weight=1.0, clap_weight=0.1, clap=clap).to(DEV)
    B = 2
    z = (cases.vae_inputs(B, 12, 16, "claploss") * 0.5)
    zt = z + 0.1 * cases.vae_inputs(B, 12, 16, "claploss.t")
    gt = det("claploss.gt", (B, 7000), 3) * 0.3                 # shorter than the clip: repeat-padded
    text = F.normalize(det("claploss.text", (B, 512), 4), dim=-1)
    # ---- oracle chain
    zo = z.clone().requires_grad_(True)
    mel = onets.vae_decode(cases.TINY_VAE_DD, vsd, zo, 0.9)
    wav = onets.hifigan_forward(cases.TINY_HIFIGAN, vsd, mel[:, 0].transpose(1, 2))[:, 0]
    wav = wav - (wav.max() + wav.min()) / 2
    clap.clip_samples = 3 * wav.shape[1]          # the small tower's clip length stands in for the 480 000 of HTSAT-base
    zd = z.to(DEV).requires_grad_(True)
    inst = loss(zd, zt.to(DEV), gt.to(DEV), text.to(DEV))
    inst.sum().backward()
    sd = dict(hsd)
    sd.update(proj)
    kw = KAISER

    fin48 = oclap.resample(wav[:, :160000], 16000, 48000, **kw)
    gt48 = C.CLAP_Module._fit(oclap.resample(gt, 16000, 48000, **kw), fin48.shape[1])
    osd = {("audio_branch." + k if not k.startswith("audio_projection") else k): v for k, v in sd.items()}
    e_in = oclap.audio_features(cfg, osd, fin48)
    with torch.no_grad():
        e_gt = oclap.audio_features(cfg, osd, gt48)
    mse = ((zo - zt) ** 2).reshape(B, -1).mean(1)
    ref = 1.0 * mse + 0.1 * (2 - F.cosine_similarity(e_in, text, dim=1) - F.cosine_similarity(e_in, e_gt, dim=1))
    ref.sum().backward()
    print("CLAP loss hip", inst.detach().cpu().numpy(), "oracle", ref.detach().numpy())
    np.testing.assert_allclose(inst.detach().cpu().numpy(), ref.detach().numpy(), rtol=3e-2)
    gl2 = rel_l2(zd.grad, zo.grad)
    print("CLAP loss d/d latent rel_l2 %.3e" % gl2)
    assert gl2 < 4e-2     # measured 3.0e-2 on MI355X (round 6) + 30 %: the piecewise-linear vocoder's mask flips dominate it
                          # (LABNOTES.md 4b: 0.13..0.19 on the vocoder's OWN input gradient); the bound stood at 0.3 for three rounds


def test_audiolcm_clap_finetune_step_runs_and_moves_the_student():
    """BASELINE configs[4] at toy size through the public path: AudioLCM(loss_type='clap').train_step with ground-truth
    audio and pre-computed CLAP text features -- consistency generation, differentiable decode, resampler, CLAP tower
    forward + input gradient, U-Net backward, AdamW, EMA.  Checks the loss against a manual evaluation of the same
    CLAPLoss on the step's own prediction, that the student moved and the frozen towers did not."""
    from consistencytta_amd import losses, modules
    from consistencytta_amd.models import AudioLCM
    ucfg = cases.TINY_UNET
    hcfg = dict(cases.TINY_HTSAT, spec_size=128)                       # 256 frames: room for the 64-frame mel of a 16x16 latent
    clap = C.CLAP_Module(audio_cfg=hcfg, text_cfg=cases.TINY_ROBERTA, clip_samples=3 * (64 * 160 + 32))
    clap.to(DEV)
    clap.model.init_random_(seed=3)
    vsd = dict(cases.vae_weights(cases.TINY_VAE_DD))
    vsd.update(cases.hifigan_weights(cases.TINY_HIFIGAN))
    vae = modules.AutoencoderKL(ddconfig=cases.TINY_VAE_DD, embed_dim=8, scale_factor=0.9, hifigan_config=cases.TINY_HIFIGAN)
    vae.load_state_dict(vsd)
    vae.to(DEV).eval().requires_grad_(False)
    m = AudioLCM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                 unet_model_config_path="tiny_light.json", unet_config=ucfg, snr_gamma=5.0, use_edm=True,
                 teacher_guidance_scale=-1, num_diffusion_steps=18, vae=vae, loss_type="clap", clap_module=clap)
    m.teacher_unet.load_state_dict(cases.unet_weights(ucfg, False, 0))
    for net, seed in ((m.student_unet, 1), (m.student_target_unet, 2), (m.student_ema_unet, 3)):
        net.load_state_dict(cases.unet_weights(ucfg, True, seed))
    m.to(DEV)
    m.train()
    assert isinstance(m.loss, losses.CLAPLoss) and not any(p.requires_grad for p in m.loss.clap.parameters())
    B = 2
    P = {k: v.to(DEV) for k, v in cases.prompt_states(ucfg, B, 6, "clapft").items()}
    ids, mask = cases.roberta_inputs(cases.TINY_ROBERTA, B, 12, "clapft")
    P["clap_text_features"] = clap.model.get_text_embedding({"input_ids": ids.to(DEV), "attention_mask": mask.to(DEV)})
    assert abs(float(P["clap_text_features"].norm(dim=1)[0]) - 1.0) < 1e-3
    z0 = (cases.t(spec.det_uniform("clapft.z0", (B, 8, 16, 16), 14)) * 0.9).to(DEV)
    gt = (det("clapft.gt", (B, 9000), 5) * 0.3).to(DEV)
    opt = m.prepare_training(lr=1e-4, weight_decay=0.0, broadcast=False)
    before = opt.flat.detach().clone()
    tower_before = m.loss.clap.model.audio_branch.get_parameter("norm.weight").detach().clone()
    gen = torch.Generator().manual_seed(3)
    kw = dict(time_inds=torch.randint(0, 17, (B,), generator=gen) * 2,
              gaussian_noise=torch.randn(B, 8, 16, 16, generator=gen).to(DEV), guidance_scale=torch.rand(B, generator=gen) * 6)
    v1 = m.train_step(z0, P, opt, None, gt_wav=gt, **kw)
    torch.cuda.synchronize()
    assert np.isfinite(v1) and v1 > 0
    moved = float((opt.flat - before).abs().max())
    print("CLAP fine-tuning step: loss %.5f, max parameter change %.3e" % (v1, moved))
    assert moved > 0 and torch.equal(m.loss.clap.model.audio_branch.get_parameter("norm.weight"), tower_before)
    v2 = m.train_step(z0, P, opt, None, gt_wav=gt, **kw)
    assert np.isfinite(v2) and v2 != v1


@pytest.mark.parametrize("loss_type", ["mel", "stft", "clap"])
def test_perceptual_losses_decrease_over_twenty_fixed_draw_steps(loss_type):
    """with the random draws of AudioLCM.forward held fixed (timestep indices, noise, guidance), twenty
    optimizer steps on the waveform- / mel-domain losses must lower the loss -- the bench's `loss_first_last` draws a new
    timestep every step and cannot show this."""
    from consistencytta_amd import modules
    from consistencytta_amd.models import AudioLCM
    ucfg = cases.TINY_UNET
    vsd = dict(cases.vae_weights(cases.TINY_VAE_DD))
    vsd.update(cases.hifigan_weights(cases.TINY_HIFIGAN))
    vae = modules.AutoencoderKL(ddconfig=cases.TINY_VAE_DD, embed_dim=8, scale_factor=0.9, hifigan_config=cases.TINY_HIFIGAN)
    vae.load_state_dict(vsd)
    vae.to(DEV).eval().requires_grad_(False)
    kw_model = {}
    B = 2
    P = {k: v.to(DEV) for k, v in cases.prompt_states(ucfg, B, 6, "decr").items()}
    if loss_type == "clap":
        clap = C.CLAP_Module(audio_cfg=dict(cases.TINY_HTSAT, spec_size=128), text_cfg=cases.TINY_ROBERTA,
                             clip_samples=3 * (64 * 160 + 32))
        clap.to(DEV)
        clap.model.init_random_(seed=3)
        kw_model["clap_module"] = clap
        P["clap_text_features"] = F.normalize(det("decr.text", (B, 512), 4), dim=-1).to(DEV)
    m = AudioLCM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                 unet_model_config_path="tiny_light.json", unet_config=ucfg, snr_gamma=5.0, use_edm=True,
                 teacher_guidance_scale=-1, num_diffusion_steps=18, vae=vae, loss_type=loss_type, **kw_model)
    m.teacher_unet.load_state_dict(cases.unet_weights(ucfg, False, 0))
    for net, seed in ((m.student_unet, 1), (m.student_target_unet, 1), (m.student_ema_unet, 1)):
        net.load_state_dict(cases.unet_weights(ucfg, True, seed))     # target = EMA = student, as after load_state_dict_from_tango
    m.to(DEV)
    m.train()
    z0 = (cases.t(spec.det_uniform("decr.z0", (B, 8, 16, 16), 14)) * 0.9).to(DEV)
    gt = (det("decr.gt", (B, 9000), 5) * 0.3).to(DEV)
    opt = m.prepare_training(lr=2e-4, weight_decay=0.0, broadcast=False)
    gen = torch.Generator().manual_seed(11)
    kw = dict(time_inds=torch.randint(2, 15, (B,), generator=gen) * 2,
              gaussian_noise=torch.randn(B, 8, 16, 16, generator=gen).to(DEV), guidance_scale=torch.rand(B, generator=gen) * 6)
    losses = [m.train_step(z0, P, opt, None, gt_wav=gt, **kw) for _ in range(20)]
    print("%s loss over 20 fixed-draw steps: %.5f -> %.5f (min %.5f)" % (loss_type, losses[0], losses[-1], min(losses)))
    assert all(np.isfinite(v) for v in losses)
    assert np.mean(losses[-3:]) < 0.9 * np.mean(losses[:3])


# ------------------------------------------------------------------------------------------------
# BASELINE configs[4] at its REAL size: light U-Nets, full-width VAE decoder + HiFi-GAN, HTSAT-base, one
# 10.24 s clip -- loss and student gradients against the reference's own AudioLCM + CLAPLoss under torch autograd
# (tests/golden/make_golden_clapft_light.py).
def _clapft_light(golden):
    from consistencytta_amd import modules
    from consistencytta_amd.models import AudioLCM
    g = golden("clapft_light")
    g0 = golden("clap_htsat")
    cfg = spec.LIGHT_UNET_CONFIG
    vsd = dict(cases.vae_weights(spec.VAE_DDCONFIG))
    vsd.update(cases.hifigan_weights(spec.HIFIGAN_16K_64))
    vae = modules.AutoencoderKL(ddconfig=spec.VAE_DDCONFIG, embed_dim=8, scale_factor=float(g["scale_factor"]),
                                hifigan_config=spec.HIFIGAN_16K_64)
    vae.load_state_dict(vsd)
    vae.to(DEV).eval().requires_grad_(False)
    clap = C.CLAP_Module(enable_fusion=False, amodel="HTSAT-base")
    clap.model.audio_branch.load_state_dict(cases.clap_weights(g0["base_keys"], g0["base_shapes"], "base", 6), strict=False)
    clap.model.load_state_dict({k: cases.t(spec.clap_det_weight("clap." + k, tuple(v.shape), 1))
                                for k, v in clap.model.state_dict().items() if k.startswith("audio_projection")}, strict=False)
    clap.to(DEV)
    m = AudioLCM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                 unet_model_config_path="tango_diffusion_light.json", unet_config=cfg, snr_gamma=5.0, use_edm=True,
                 teacher_guidance_scale=-1, num_diffusion_steps=18, vae=vae, loss_type="clap", clap_module=clap,
                 target_ema_decay=0.95, ema_decay=0.999)
    m.teacher_unet.load_state_dict(cases.unet_weights(cfg, False, 0))
    m.student_unet.load_state_dict(cases.unet_weights(cfg, True, 1))
    m.student_target_unet.load_state_dict(cases.unet_weights(cfg, True, 2))
    m.student_ema_unet.load_state_dict(cases.unet_weights(cfg, True, 3))
    m.to(DEV)
    m.train()
    B = 1
    P = {k: v.to(DEV) for k, v in cases.prompt_states(cfg, B, 16, "clapft_light").items()}
    P["clap_text_features"] = F.normalize(cases.t(spec.det_uniform("clapft_light.text", (B, 512), 4)), dim=-1).to(DEV)
    z0 = (cases.t(spec.det_uniform("clapft_light.z0", (B, 8, 256, 16), 14)) * 0.9).to(DEV)
    gt = (cases.t(spec.det_uniform("clapft_light.gt", (B, 160000), 5)) * 0.3).to(DEV)
    kw = dict(time_inds=torch.from_numpy(g["time_inds"]) * 2, gaussian_noise=torch.from_numpy(g["noise"]).to(DEV),
              guidance_scale=torch.from_numpy(g["guidance"]))
    return g, m, P, z0, gt, kw


def _block_of(key):
    head, _, rest = key.partition(".")
    return head + "." + rest.split(".")[0] if head in ("down_blocks", "up_blocks") else head


def test_clap_finetune_step_at_light_widths_matches_reference(golden):
    """configs[4] end to end at the real widths vs the reference: loss within 5e-2; the student's gradient, block by
    block on the fixture's strided samples, within the budget of the plain distillation step (4e-2; measured 5e-3)."""
    g, m, P, z0, gt, kw = _clapft_light(golden)
    loss = m(z0, gt, P, **kw)
    ref_loss = float(g["train_loss"])
    print("light-width CLAP fine-tuning loss hip %.6f ref %.6f (instance %s, mse part %s)"
          % (float(loss.detach()), ref_loss, g["instance_loss"], g["instance_mse"]))
    assert abs(float(loss) - ref_loss) <= 5e-2 * abs(ref_loss)
    loss.backward()
    torch.cuda.synchronize()
    names = [str(k) for k in g["grad_names"]]
    params = dict(m.student_unet.named_parameters())
    assert names == [k for k, p in params.items() if p.requires_grad]
    off, samples, norms = g["grad_offsets"], g["grad_samples"], g["grad_norms"]
    blocks, worst_norm = {}, ("", 0.0)
    for i, k in enumerate(names):
        gr = params[k].grad.detach().reshape(-1)
        idx = torch.from_numpy(cases.sample_index(gr.numel())).to(DEV)
        got = gr[idx].double().cpu().numpy()
        ref = samples[off[i]:off[i + 1]].astype(np.float64)
        b = blocks.setdefault(_block_of(k), [0.0, 0.0])
        b[0] += float(((got - ref) ** 2).sum())
        b[1] += float((ref ** 2).sum())
        nrel = abs(float(gr.double().norm()) - norms[i]) / max(norms[i], 1e-30)
        if nrel > worst_norm[1] and norms[i] > 1e-3 * float(norms.max()):
            worst_norm = (k, nrel)
    tot_e, tot_n = sum(b[0] for b in blocks.values()), sum(b[1] for b in blocks.values())
    for name, (e, n) in blocks.items():
        print("  %-28s sampled grad rel_l2 %.3e" % (name, (e / max(n, 1e-300)) ** 0.5))
    print("all blocks: sampled rel_l2 %.3e ; worst per-tensor norm deviation %s %.3e"
          % ((tot_e / tot_n) ** 0.5, worst_norm[0], worst_norm[1]))
    assert (tot_e / tot_n) ** 0.5 <= CLAPFT_GRAD_REL_L2
    for name, (e, n) in blocks.items():
        assert (e / max(n, 1e-300)) ** 0.5 <= CLAPFT_GRAD_REL_L2, name
    assert worst_norm[1] <= CLAPFT_GRAD_REL_L2, worst_norm
    for name in ("teacher_unet", "student_target_unet", "student_ema_unet"):
        assert all(p.grad is None for p in getattr(m, name).parameters())


def test_clap_finetune_step_as_a_replayed_graph_equals_the_eager_step(golden):
    """`capture_train_graph(..., pipeline_teacher=True, main_eager=False)` with loss_type='clap' -- target network,
    student forward, VAE decode + vocoder + resampler + CLAP towers and torch's backward through them down to the latent, then
    the engine's backward, recorded as ONE hipGraph (the teacher phase as its own graph one batch ahead): the replay gives the
    eager step's loss and the student's gradient (same kernels, same order; fp32 atomics in the loss modules leave round-off)."""
    g, m, P, z0, gt, kw = _clapft_light(golden)
    opt = m.prepare_training(lr=2e-5, weight_decay=0.0, broadcast=False)
    with torch.no_grad():
        loss, pred, target, sig, gamma = m._forward_impl(z0, gt, P, False, True, kw["time_inds"], kw["gaussian_noise"],
                                                         kw["guidance_scale"], True)
        m._student_backward(pred, target, sig, gamma, 1.0, None)
    torch.cuda.synchronize()
    g_eager, l_eager = opt.grad.clone(), float(loss)
    opt.zero_grad()
    gs = m.capture_train_graph(opt, z0, P, segmented=False, pipeline_teacher=True, main_eager=False, gt_wav=gt, **kw)
    assert gs.main_eager is False and gs.graph is not None
    assert gs.feed(z0, prompt=P, gt_wav=gt, **kw) is False
    for rnd in range(2):                       # the second replay reads the rotated static copies of gt / text features
        assert gs.feed(z0, prompt=P, gt_wav=gt, **kw) is True
        gs.replay()
        torch.cuda.synchronize()
        rel = float((opt.grad - g_eager).norm() / g_eager.norm())
        print("replayed CLAP fine-tuning step %d: loss %.7f (eager %.7f), gradient rel diff %.2e" % (rnd, float(gs.loss.item()), l_eager, rel))
        assert abs(float(gs.loss.item()) - l_eager) <= 1e-5 * abs(l_eager) and rel <= 1e-4
        opt.zero_grad()
    # a batch with OTHER ground-truth audio and caption features changes the loss: the static copies are what the graph reads
    P2 = dict(P, clap_text_features=F.normalize(P["clap_text_features"].flip(-1) + 0.1, dim=-1))
    assert gs.feed(z0, prompt=P2, gt_wav=gt.flip(-1) * 0.5, **kw) is True
    gs.replay()                                # still the first batch's inputs (fed one call ahead)
    assert abs(float(gs.loss.item()) - l_eager) <= 1e-5 * abs(l_eager)
    opt.zero_grad()
    assert gs.feed(z0, prompt=P, gt_wav=gt, **kw) is True
    gs.replay()                                # now batch 2
    torch.cuda.synchronize()
    assert abs(float(gs.loss.item()) - l_eager) > 1e-4 * abs(l_eager)
    opt.zero_grad()


def test_clap_finetune_loss_decreases_at_light_widths(golden):
    """The fixed-draw decrease test of the toy size, at configs[4]'s real size: twenty optimizer steps on ONE clip with the
    fixture's draws held fixed must lower the CLAP fine-tuning loss."""
    g, m, P, z0, gt, kw = _clapft_light(golden)
    with torch.no_grad():   # target = EMA = student, as after load_state_dict_from_tango
        for dst in (m.student_target_unet, m.student_ema_unet):
            for p, q in zip(dst.parameters(), m.student_unet.parameters()):
                p.copy_(q)
    opt = m.prepare_training(lr=2e-5, weight_decay=0.0, broadcast=False)
    losses = [m.train_step(z0, P, opt, None, gt_wav=gt, **kw) for _ in range(20)]
    print("light-width CLAP loss over 20 fixed-draw steps: %.5f -> %.5f (min %.5f)" % (losses[0], losses[-1], min(losses)))
    assert all(np.isfinite(v) for v in losses)
    assert np.mean(losses[-3:]) < 0.9 * np.mean(losses[:3])
