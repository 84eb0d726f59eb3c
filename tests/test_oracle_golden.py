"""Pins the CPU oracle against fixtures produced by the reference's own modules
(tests/golden/make_golden.py).  fp32 vs fp32 on the same CPU kernels: tolerance is
round-off only."""
import numpy as np
import pytest
import torch

import cases
from consistencytta_amd import spec
from oracle import heun, nets


def close(a, b, rtol=2e-4, atol=2e-5):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(1e-6, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= atol + rtol * scale, "max abs err %.3e (scale %.3e)" % (err, scale)


def test_param_specs_match_reference_key_order(golden):
    g = golden("unet_tiny")
    assert list(spec.unet_param_spec(cases.TINY_UNET, True).keys()) == [str(k) for k in g["keys"]]
    light = spec.unet_param_spec(spec.LIGHT_UNET_CONFIG, True)
    n = sum(int(np.prod(s)) for s in light.values())
    assert n == int(golden("unet_light")["n_params"]) == 559209676


def test_heun_tables(golden):
    g = golden("heun")
    for n in (1, 2, 18, 200):
        ts, sig = heun.set_timesteps(n)
        np.testing.assert_array_equal(ts, g["timesteps_%d" % n])
        # fp32 torch.linspace/cumprod on the host: last-bit differences between CPU models allowed
        np.testing.assert_allclose(sig, g["sigmas_%d" % n], rtol=1e-6, atol=0)
        assert abs(float(sig.max()) - float(g["init_sigma_%d" % n])) <= 1e-6 * float(sig.max())
    ts, sig = heun.set_timesteps(18)
    assert len(ts) == 35 and len(sig) == 36
    assert abs(float(sig[0]) - 14.6146) < 1e-3


def test_heun_steps(golden):
    g = golden("heun")
    _, sig = heun.set_timesteps(18)
    sig = torch.from_numpy(sig)
    idx = torch.from_numpy(g["idx"])
    x = cases.t(spec.det_uniform("heun.x", (3, 8, 16, 4), 1)) * 3
    v1 = cases.t(spec.det_uniform("heun.v1", (3, 8, 16, 4), 2))
    v2 = cases.t(spec.det_uniform("heun.v2", (3, 8, 16, 4), 3))
    noise = cases.t(spec.det_uniform("heun.n", (3, 8, 16, 4), 4))
    close(heun.scale_model_input(x, sig[idx]), g["scaled"], 1e-6, 1e-7)
    close(heun.add_noise(x, noise, sig[idx]), g["noised"], 1e-6, 1e-7)
    first, d, dt = heun.step_first(v1, x, sig[idx], sig[idx + 1])
    close(first, g["step1"], 1e-6, 1e-7)
    # 2nd-order call happens at timestep index idx+2, in second-order state:
    # index_for_timestep returns (last match) - 1 = idx+1; sigma=sig[idx], sigma_next=sig[idx+1]
    close(heun.scale_model_input(first, sig[idx + 2]), g["scaled2"], 1e-6, 1e-7)
    second = heun.step_second(v2, first, sig[idx + 1], x, d, dt)
    close(second, g["step2"], 1e-6, 1e-7)


def test_unet_tiny(golden):
    g = golden("unet_tiny")
    cfg = cases.TINY_UNET
    with torch.no_grad():
        sd = cases.unet_weights(cfg, True)
        x, ts, gs, enc, mask = cases.unet_inputs(cfg, 2, 32, 8, 7, "unet_tiny")
        close(nets.unet_forward(cfg, sd, x, ts, gs, enc, mask), g["guided"])
        x2, _, _, enc2, mask2 = cases.unet_inputs(cfg, 2, 16, 8, 5, "unet_tiny_s", False)
        close(nets.unet_forward(cfg, sd, x2, 999.0, 4.0, enc2, mask2), g["guided_scalar"])
        sdt = cases.unet_weights(cfg, False)
        close(nets.unet_forward(cfg, sdt, x, ts, None, enc, mask), g["teacher"])


@pytest.mark.slow
def test_unet_light(golden):
    g = golden("unet_light")
    cfg = spec.LIGHT_UNET_CONFIG
    with torch.no_grad():
        sd = cases.unet_weights(cfg, True)
        x, _, _, enc, mask = cases.unet_inputs(cfg, 1, 256, 16, 16, "unet_light", False)
        x = x / 1.7 * cases.SIGMA_MAX / ((cases.SIGMA_MAX ** 2 + 1) ** 0.5)
        close(nets.unet_forward(cfg, sd, x, 999.0, 4.0, enc, mask), g["out"])


def test_vae_hifigan_tiny(golden):
    g = golden("vae_tiny")
    with torch.no_grad():
        sd = dict(cases.vae_weights(cases.TINY_VAE_DD))
        sd.update(cases.hifigan_weights(cases.TINY_HIFIGAN))
        z = cases.vae_inputs(2, 16, 8, "vae_tiny")
        close(nets.vae_decode(cases.TINY_VAE_DD, sd, z, float(g["scale_factor"])), g["mel"])
        mel_in = cases.mel_inputs(2, 24, 64, "hifigan_tiny")
        wav, centred, pcm = nets.mel_to_waveform(cases.TINY_HIFIGAN, sd, mel_in)
        close(wav, g["wav"])
        assert np.abs(pcm.astype(np.int64) - g["pcm"].astype(np.int64)).max() <= 1


def test_differentiable_decode_gradients(golden):
    """decode_first_stage / decode_to_waveform with allow_grad=True (tools/losses.py:294-298): waveform, mel and the
    latent / mel gradients of the reference's autograd graph."""
    g = golden("vae_grad_tiny")
    sd = dict(cases.vae_weights(cases.TINY_VAE_DD))
    sd.update(cases.hifigan_weights(cases.TINY_HIFIGAN))
    z = cases.vae_inputs(2, 16, 16, "vae_grad").clone().requires_grad_(True)
    mel = nets.vae_decode(cases.TINY_VAE_DD, sd, z, float(g["scale_factor"]))
    mel.retain_grad()
    _, centred, _ = nets.mel_to_waveform(cases.TINY_HIFIGAN, sd, mel)
    direction = cases.t(spec.det_uniform("vae_grad.direction", tuple(centred.shape), 21))
    (centred * direction).sum().backward()
    close(mel.detach(), g["mel"])
    close(centred.detach(), g["wav"])
    close(mel.grad, g["grad_mel"], rtol=1e-4, atol=1e-6)
    close(z.grad, g["grad_z"], rtol=1e-4, atol=1e-6)


def test_mel_loss_and_its_latent_gradient(golden):
    """tools.losses.MelLoss over the reference VAE: instance losses and d(weighted mean)/d(predicted latent)."""
    from oracle import distill
    g = golden("melloss_tiny")
    sd = dict(cases.vae_weights(cases.TINY_VAE_DD))
    pred = (cases.vae_inputs(2, 16, 16, "melloss.pred") * 0.5).requires_grad_(True)
    target = cases.vae_inputs(2, 16, 16, "melloss.target") * 0.5
    inst = distill.mel_loss_instances(cases.TINY_VAE_DD, sd, pred, target, float(g["scale_factor"]))
    (inst * torch.from_numpy(g["weights"])).mean().backward()
    close(inst.detach(), g["instance_loss"])
    close(pred.grad, g["grad_pred"], rtol=1e-4, atol=1e-7)


def test_t5_encoder_restatement_against_transformers(golden):
    """oracle.t5 vs transformers.T5EncoderModel (the reference's text encoder, audio_distilled_model.py:97-98,208-214):
    padded batches, a sequence longer than max_distance (all 32 relative-position buckets), FLAN-T5-large's widths."""
    from oracle import t5
    g = golden("t5_encoder")
    for name, cfg, B, L, tag in (("tiny", cases.TINY_T5, 3, 13, "t5_tiny"), ("tiny_long", cases.TINY_T5, 2, 150, "t5_long"),
                                 ("wide", cases.WIDE_T5, 2, 16, "t5_wide")):
        ids, mask = cases.t5_inputs(cfg, B, L, tag)
        with torch.no_grad():
            out = t5.t5_encode(cfg, cases.t5_weights(cfg), ids, mask)
        close(out, g[name])
    # known answers of the bucket function (modeling_t5.py _relative_position_bucket, bidirectional, 32 / 128)
    rp = torch.tensor([0, 1, -1, 7, 8, -8, 15, 16, 64, 127, 128, 1000, -1000])
    assert t5.relative_position_bucket(rp).tolist() == [0, 17, 1, 23, 24, 8, 25, 26, 30, 31, 31, 31, 15]


@pytest.mark.slow
def test_vae_hifigan_full_width(golden):
    g = golden("vae_full")
    with torch.no_grad():
        sd = dict(cases.vae_weights(spec.VAE_DDCONFIG))
        sd.update(cases.hifigan_weights(spec.HIFIGAN_16K_64))
        z = cases.vae_inputs(1, 64, 16, "vae_full")
        close(nets.vae_decode(spec.VAE_DDCONFIG, sd, z, float(g["scale_factor"])), g["mel"])
        mel_in = cases.mel_inputs(1, 64, 64, "hifigan_full")
        wav, _, _ = nets.mel_to_waveform(spec.HIFIGAN_16K_64, sd, mel_in)
        close(wav, g["wav"])


def _distill_setup():
    prompt_states = cases.prompt_states
    from oracle import distill
    cfg = cases.TINY_UNET
    n = distill.Nets(cfg, cases.unet_weights(cfg, False, 0), cases.unet_weights(cfg, True, 1),
                     cases.unet_weights(cfg, True, 2), cases.unet_weights(cfg, True, 3))
    P = prompt_states(cfg, 3, 6, "distill")
    z0 = cases.t(spec.det_uniform("distill.z0", (3, 8, 32, 8), 14)) * 0.9
    return distill, n, P, z0


def test_distillation_loss_and_teacher_query(golden):
    """oracle.distill vs the reference's own AudioLCM.forward / _query_teacher (a13, a15, a16)."""
    g = golden("distill_tiny")
    distill, n, P, z0 = _distill_setup()
    with torch.no_grad():
        inds = torch.from_numpy(g["time_inds"]) * 2
        loss = distill.distill_loss(n, P, z0, torch.from_numpy(g["noise"]), inds, torch.from_numpy(g["guidance"]))
        assert abs(float(loss) - float(g["train_loss"])) <= 2e-4 * float(g["train_loss"])
        _, sig = heun.set_timesteps(18)
        ts, _ = heun.set_timesteps(18)
        zq = cases.t(spec.det_uniform("distill.zq", (3, 8, 32, 8), 15))
        tq = torch.from_numpy(ts[[0, 6, 32]])
        close(distill.query_teacher(n, zq, tq, P["embeds_cf"], P["mask_cf"], torch.tensor([0.5, 3.0, 5.5])),
              g["query_teacher"])
        vl = distill.validation_losses(n, P, z0, torch.from_numpy(g["val_noise"]), 2,
                                       torch.from_numpy(g["val_guidance"]))
        np.testing.assert_allclose([float(v) for v in vl], g["val_losses"], rtol=5e-4)


@pytest.mark.slow
def test_inference_student_and_teacher_loops(golden):
    """oracle.distill inference vs AudioLCM.inference (a14): 1-step, 2-step + post-CFG, Heun teacher."""
    g = golden("distill_tiny")
    distill, n, P, _ = _distill_setup()
    noise = cases.t(spec.det_uniform("distill.inf_noise", (3, 8, 256, 16), 16)) * np.float32(np.sqrt(3.0))
    with torch.no_grad():
        close(distill.inference_student(n, n.ema, P, noise, 4.0, 1.0, 1), g["inf_student_1step"])
        close(distill.inference_student(n, n.target, P, noise, 3.0, 2.0, 2, torch.from_numpy(g["inf_renoise"])),
              g["inf_student_2step_cfg"])
        close(distill.inference_teacher(n, P, noise, 4.0, 3), g["inf_teacher_3steps"], 5e-4, 5e-5)


def test_vae_encoder_tiny(golden):
    """oracle.nets.vae_encode / posterior_sample vs the reference's AutoencoderKL.encode_first_stage and
    get_first_stage_encoding (recorded noise draw)."""
    g = golden("vae_encoder_tiny")
    with torch.no_grad():
        sd = cases.vae_encoder_weights(cases.TINY_VAE_DD)
        assert list(sd.keys()) == list(spec.vae_encoder_param_spec(cases.TINY_VAE_DD).keys())
        mel = cases.mel_inputs(2, 64, 16, "vaeenc_tiny") * 2.0 - 4.0
        mom = nets.vae_encode(cases.TINY_VAE_DD, sd, mel)
        close(mom, g["moments"])
        z = nets.posterior_sample(mom, torch.from_numpy(g["noise"]), float(g["scale_factor"]))
        close(z, g["z"])


@pytest.mark.slow
def test_vae_encoder_full_width(golden):
    g = golden("vae_encoder_full")
    with torch.no_grad():
        sd = cases.vae_encoder_weights(spec.VAE_DDCONFIG)
        mel = cases.mel_inputs(1, 128, 64, "vaeenc_full") * 2.0 - 4.0
        close(nets.vae_encode(spec.VAE_DDCONFIG, sd, mel), g["moments"])


def test_mel_filterbank_known_answers():
    """librosa.filters.mel (absent here) restated in oracle/mel.py: published constants of the Slaney mel scale and the
    structural properties of the area-normalised triangular filterbank."""
    from oracle import mel as omel
    assert abs(float(omel.hz_to_mel(1000.0)) - 15.0) < 1e-12 and abs(float(omel.mel_to_hz(15.0)) - 1000.0) < 1e-9
    assert abs(float(omel.hz_to_mel(8000.0)) - (15.0 + 27.0 * np.log(8.0) / np.log(6.4))) < 1e-9
    assert abs(float(omel.hz_to_mel(500.0)) - 7.5) < 1e-12                    # linear region: 200/3 Hz per mel
    w = omel.mel_filterbank(16000, 1024, 64, 0.0, 8000.0)
    assert w.shape == (64, 513) and w.dtype == np.float32 and float(w.min()) >= 0.0
    # slaney normalisation: every triangle has unit area in Hz (bin width 15.625 Hz), up to bin sampling
    area = w.sum(1) * (16000.0 / 1024)
    assert np.all(np.abs(area[8:] - 1.0) < 0.05), area
    # filters are ordered, overlap only with neighbours, and tile the band edge to edge
    peaks = w.argmax(1)
    assert np.all(np.diff(peaks) > 0) and peaks[0] >= 1 and peaks[-1] <= 512
    assert float((w[:-2] * w[2:]).sum()) == 0.0
    assert float(w[:, 0].sum()) == 0.0 and float(w[:, 512].sum()) == 0.0


def test_mel_frontend(golden):
    """oracle.mel.wav_to_fbank vs the reference's TacotronSTFT + tools.torch_tools.wav_to_fbank."""
    import make_golden_mel as mg
    from oracle import mel as omel
    g = golden("mel_frontend")
    basis = omel.stft_basis(1024)
    close(basis[[0, 1, 7, 512, 513, 520, 1025]], g["basis_rows"], 1e-6, 1e-7)
    close(omel.mel_filterbank().sum(1), g["mel_basis_sum"], 1e-6, 1e-8)
    with torch.no_grad():
        fb, lm = omel.wav_to_fbank(mg.test_wave(2, 40000, "mel"), 256)
        assert tuple(fb.shape) == (2, 256, 64) and tuple(lm.shape) == (2, 256, 512)
        close(fb, g["fbank"], 1e-5, 2e-4)          # log domain: absolute tolerance
        close(lm[:, ::8, ::8], g["logmag_sub"], 1e-5, 2e-4)
        assert float(fb[:, 251:].abs().max()) == 0.0                 # _pad_spec pads with zeros, not log(1e-5)
        fb2, _ = omel.wav_to_fbank(mg.test_wave(1, 163840, "mel_full"), 1024)
        close(fb2, g["fbank_full"], 1e-5, 2e-4)


def test_stage1_schedulers_and_guided_distillation(golden):
    """oracle.ddim / oracle.distill.gdm_* vs the reference's DDPMScheduler, DDIMScheduler and AudioGDM.forward
    (SURVEY §8f rank 3)."""
    from oracle import ddim, distill
    g = golden("gdm_tiny")
    ac = ddim.alphas_cumprod()
    np.testing.assert_allclose(ac.numpy(), g["alphas_cumprod"], rtol=1e-6)
    np.testing.assert_array_equal(ddim.ddpm_timesteps()[:5].numpy(), g["ddpm_timesteps_head"])
    for n in (5, 50):
        np.testing.assert_array_equal(ddim.ddim_timesteps(n).numpy(), g["ddim_timesteps_%d" % n])
    x = cases.t(spec.det_uniform("gdm.x", (3, 8, 16, 4), 1)) * 2
    noise = cases.t(spec.det_uniform("gdm.n", (3, 8, 16, 4), 2))
    v = cases.t(spec.det_uniform("gdm.v", (3, 8, 16, 4), 3))
    t_train = torch.tensor([999, 400, 0])
    close(ddim.add_noise(x, noise, t_train, ac), g["ddpm_add_noise"], 1e-6, 1e-7)
    close(ddim.add_noise(x, noise, t_train, ac), g["ddim_add_noise"], 1e-6, 1e-7)
    close(ddim.ddim_step(v, torch.from_numpy(g["ddim_t"]), x, 5, ac), g["ddim_step"], 1e-6, 1e-7)
    close(ddim.ddim_step(v, torch.full((3,), 600), x, 5, ac), g["ddim_step_scalar_t"], 1e-6, 1e-7)
    cfg = cases.TINY_UNET
    n = distill.Nets(cfg, cases.unet_weights(cfg, False, 0), cases.unet_weights(cfg, True, 1), None,
                     cases.unet_weights(cfg, True, 3))
    P = cases.prompt_states(cfg, 3, 6, "distill")
    z0 = cases.t(spec.det_uniform("distill.z0", (3, 8, 32, 8), 14)) * 0.9
    with torch.no_grad():
        loss = distill.gdm_loss(n, P, z0, torch.from_numpy(g["gdm_noise"]), torch.from_numpy(g["gdm_time_inds"]),
                                torch.from_numpy(g["gdm_guidance"]))
    assert abs(float(loss) - float(g["gdm_loss"])) <= 2e-4 * float(g["gdm_loss"])


@pytest.mark.slow
def test_stage1_inference(golden):
    from oracle import distill
    g = golden("gdm_tiny")
    cfg = cases.TINY_UNET
    n = distill.Nets(cfg, cases.unet_weights(cfg, False, 0), cases.unet_weights(cfg, True, 1), None,
                     cases.unet_weights(cfg, True, 3))
    P = cases.prompt_states(cfg, 3, 6, "distill")
    lat = cases.t(spec.det_uniform("gdm.inf_noise", (3, 8, 256, 16), 16)) * np.float32(np.sqrt(3.0))
    with torch.no_grad():
        close(distill.gdm_inference(n, n.ema, P, lat, 3.0, 4), g["gdm_inference_4steps"], 5e-4, 5e-5)


# ------------------------------------------------------------------------------------------------ CLAP audio tower
def test_clap_oracle_matches_reference_htsat(golden):
    """oracle/clap.py (resampler-free part: log-mel front end + HTSAT Swin tower) against the fixtures the reference's
    own laion_clap/clap_module/htsat.py produced (tests/golden/make_golden_clap.py): image, embedding, input gradient."""
    from oracle import clap as oclap
    g = golden("clap_htsat")
    cfg = cases.TINY_HTSAT
    sd = cases.clap_weights(g["tiny_keys"], g["tiny_shapes"], "tiny", 5)
    wav = (cases.t(spec.det_uniform("clap.tiny.wav", (2, 28800), 3)) * 0.4).requires_grad_(True)
    taps = {}
    emb = oclap.htsat_embedding(cfg, sd, wav, prefix="", taps=taps)
    np.testing.assert_allclose(taps["image"].detach().numpy(), g["tiny_image"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(emb.detach().numpy(), g["tiny_embedding"], rtol=2e-4, atol=2e-4)
    (emb * cases.t(spec.det_uniform("tiny.dir", tuple(emb.shape), 9))).sum().backward()
    ref = torch.from_numpy(g["tiny_grad"])
    assert float((wav.grad - ref).norm() / ref.norm()) < 2e-4
    # the build's own parameter table reproduces the reference module's key list (order included)
    mine = [k for k in spec.htsat_param_spec(spec.HTSAT_BASE_CONFIG) if not k.startswith(("spectrogram_extractor", "logmel_extractor"))]
    assert mine == [str(k) for k in g["base_keys"]]
    assert [",".join(str(d) for d in spec.htsat_param_spec()[k]) for k in mine] == [str(s) for s in g["base_shapes"]]


@pytest.mark.slow
def test_clap_oracle_htsat_base_embedding(golden):
    from oracle import clap as oclap
    g = golden("clap_htsat")
    sd = cases.clap_weights(g["base_keys"], g["base_shapes"], "base", 6)
    wav = cases.t(spec.det_uniform("clap.base.wav", (2, 480000), 4)) * 0.4
    with torch.no_grad():
        emb = oclap.htsat_embedding(spec.HTSAT_BASE_CONFIG, sd, wav, prefix="")
    np.testing.assert_allclose(emb.numpy(), g["base_embedding"], rtol=2e-4, atol=2e-4)


def test_kaiser_sinc_resampler_known_answers():
    """torchaudio's published sinc_interp_kaiser algorithm (absent here) restated in oracle/clap.py: properties any correct
    implementation has -- unit DC gain per phase, a band-limited sine is reproduced on the new grid, the output length is
    ceil(new * len / orig), up then down returns the signal -- plus the host table of the product path being identical."""
    from consistencytta_amd import clap as C
    from oracle import clap as oclap
    kw = dict(lowpass_filter_width=64, rolloff=0.9475937167399596, beta=14.769656459379492)
    k, width, orig, new = oclap.sinc_resample_kernel(16000, 48000, **kw)
    assert (orig, new, width) == (1, 3, 68) and tuple(k.shape) == (3, 1, 137)
    np.testing.assert_allclose(k.sum(-1).numpy().ravel(), 1.0, atol=2e-6)            # DC gain of every phase filter
    kp, wp, op, npp = C.sinc_resample_kernel(16000, 48000, **kw)
    assert (wp, op, npp) == (68, 1, 3) and np.array_equal(kp, k[:, 0].numpy())
    t16 = torch.arange(4000, dtype=torch.float64) / 16000
    x = torch.sin(2 * np.pi * 1000.0 * t16).float()[None]
    y = oclap.resample(x, 16000, 48000, **kw)
    assert y.shape == (1, 12000)
    t48 = torch.arange(12000, dtype=torch.float64) / 48000
    ref = torch.sin(2 * np.pi * 1000.0 * t48).float()
    assert float((y[0, 300:-300] - ref[300:-300]).abs().max()) < 2e-3
    back = oclap.resample(y, 48000, 16000, **kw)
    assert back.shape == x.shape and float((back[0, 300:-300] - x[0, 300:-300]).abs().max()) < 3e-3
    assert oclap.resample(torch.zeros(2, 333), 16000, 48000, **kw).shape == (2, 999)


# ------------------------------------------------------------------------------------------------ evaluation suite
def test_eval_suite_oracle_matches_reference(golden):
    """oracle/evalsuite.py against tests/golden/eval_suite.npz, which the reference's own `Cnn14.forward`
    (audioldm_eval/feature_extractors/panns/models.py:269-323) and its own metric functions
    (audioldm_eval/metrics/{fid,isc,kid,kl}.py) produced (tests/golden/make_golden_eval.py)."""
    from oracle import evalsuite as oe
    g = golden("eval_suite")
    cfg = spec.CNN14_16K_CONFIG
    # the build's parameter table reproduces the reference module's key list (order and shapes included)
    mine = [k for k in spec.cnn14_param_spec(cfg) if k not in spec.CNN14_STRUCTURAL]
    assert mine == [str(k) for k in g["cnn14_keys"]]
    assert [",".join(str(d) for d in spec.cnn14_param_spec(cfg)[k]) for k in mine] == [str(s) for s in g["cnn14_shapes"]]
    sd = cases.cnn14_weights(g["cnn14_keys"], g["cnn14_shapes"])
    for tag, B, L in (("short", 2, 32000), ("clip", 1, 160000)):
        taps = {}
        with torch.no_grad():
            out = oe.cnn14_forward(cfg, sd, cases.eval_waves("evalsuite." + tag, B, L), taps)
        np.testing.assert_allclose(out["2048"].numpy(), g[tag + "_2048"], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(out["logits"].numpy(), g[tag + "_logits"], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(out["clipwise_output"].numpy(), g[tag + "_clipwise"], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(taps["block6"].numpy()[:, ::16], g[tag + "_block6"], rtol=2e-4, atol=2e-4)
        rms = [float(taps["block%d" % i].double().pow(2).mean().sqrt()) for i in range(1, 7)]
        np.testing.assert_allclose(rms, g[tag + "_block_rms"], rtol=1e-4)
    X = {k: v.numpy() for k, v in cases.eval_metric_inputs().items()}
    np.testing.assert_allclose(oe.fid(X["fid1"].astype(np.float32), X["fid2"].astype(np.float32)), float(g["fid"]), rtol=1e-6)
    np.testing.assert_allclose(oe.isc(X["isc"]), g["isc"], rtol=1e-9)
    np.testing.assert_allclose(oe.kid(X["kid1"], X["kid2"], subsets=100, subset_size=90), g["kid"], rtol=2e-5)   # float32 kernel sums, another order
    names, perm = cases.eval_kl_names(50)
    np.testing.assert_allclose(oe.kl(X["kl1"], X["kl2"]), g["kl"], rtol=2e-5)
