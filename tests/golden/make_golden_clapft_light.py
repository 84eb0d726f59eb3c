"""Fixture pinning BASELINE configs[4] -- the CLAP fine-tuning step -- at its REAL size: the reference's
own `models.AudioLCM` (light U-Nets, 559 M parameters each) with the reference's own `tools.losses.CLAPLoss`
(tools/losses.py:259-316) over the reference's own full-width `AutoencoderKL` + HiFi-GAN, B = 1, one 10.24 s latent, the
internal random draws recorded; loss and the student's gradients from torch autograd (build container only, ~5 min of CPU):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_clapft_light.py

What is the reference's code and what is bound to a restatement (stated because it bounds what the fixture pins):
  * AudioLCM.forward, CLAPLoss.forward, AutoencoderKL.decode_first_stage / decode_to_waveform(allow_grad=True), the HiFi-GAN
    generator and the HTSAT-base audio tower (`laion_clap/clap_module/htsat.py`, loaded as in make_golden_clap.py): reference;
  * `torchaudio.functional.resample` (not installed): `oracle.clap.resample`, pinned by known answers in
    tests/test_oracle_golden.py;
  * `laion_clap.CLAP_Module` (its hook.py pulls in the whole training stack and a checkpoint download): a 12-line stand-in
    that does what `get_audio_embedding_from_data(use_tensor=True)` does for a clip of exactly 480 000 samples
    (hook.py:163-195 -> model.py:718-744: tower "embedding" -> audio_projection MLP -> L2 normalisation) and returns fixed
    unit-norm caption features from `get_text_embedding` (the RoBERTa tower is frozen, carries no gradient, and is pinned
    separately against transformers);
  * torchlibrosa's STFT / log-mel inside the tower: oracle/clap.py, as in make_golden_clap.py.
Weights are the build's deterministic generators (no checkpoint exists offline): the U-Nets of distill_light.npz, the VAE
and vocoder of vae_full.npz, the HTSAT-base tower of clap_htsat.npz ("base", seed 6), audio_projection from its key names.

The fixture keeps the loss, the per-instance CLAP terms, the recorded draws and, for every student tensor, the gradient's
L2 norm and a strided sample of up to 512 entries (cases.sample_index), like distill_light.npz.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
from consistencytta_amd import spec  # noqa: E402
from oracle import clap as oclap  # noqa: E402
import make_golden  # noqa: E402
import make_golden_clap  # noqa: E402
from make_golden_distill import load_reference_audiolcm  # noqa: E402

B, H, W, L = 1, 256, 16, 16
sample_index = cases.sample_index


class RefClapStandIn(torch.nn.Module):
    """laion_clap.CLAP_Module as CLAPLoss uses it (tools/losses.py:305-307), around the REFERENCE's HTSAT tower."""

    def __init__(self, tower, proj, text_features):
        super().__init__()
        self.tower, self.proj = tower, proj
        self.register_buffer("text_features", text_features)

    def get_audio_embedding_from_data(self, x, use_tensor=False):
        assert use_tensor and x.shape[-1] == 480000          # hook.py:174-186: neither truncation nor padding happens
        self.tower.eval()                                     # hook.py:163 `self.model.eval()` on every call
        e = self.tower({"waveform": x}, device="cpu")["embedding"]           # model.py:587-601 encode_audio
        e = F.linear(F.relu(F.linear(e, self.proj["audio_projection.0.weight"], self.proj["audio_projection.0.bias"])),
                     self.proj["audio_projection.2.weight"], self.proj["audio_projection.2.bias"])   # model.py:537-541,738
        return F.normalize(e, dim=-1)                                                                 # model.py:739

    def get_text_embedding(self, x, tokenizer=None, use_tensor=False):
        assert use_tensor
        return self.text_features


def main():
    ns, AudioLCM, TU = load_reference_audiolcm()
    import tools.losses as RL
    RL = importlib.reload(RL)                 # load_reference_audiolcm stubbed CLAPLoss out; this is the real class again
    RL.resample = lambda wav, orig_freq, new_freq, lowpass_filter_width, rolloff, resampling_method, beta: oclap.resample(
        wav, orig_freq, new_freq, lowpass_filter_width=lowpass_filter_width, rolloff=rolloff, beta=beta)
    HT = make_golden_clap.load_reference_htsat()
    base = types.SimpleNamespace(mel_bins=64, window_size=1024, hop_size=480, sample_rate=48000, fmin=50, fmax=14000,
                                 class_num=527, model_name="base")
    tower = HT.create_htsat_model(base).eval()
    _, hkeys, hshapes = make_golden_clap.det_weights(tower, "base.", 6)
    g0 = np.load(os.path.join(HERE, "clap_htsat.npz"))
    assert list(g0["base_keys"]) == hkeys, "the tower of this fixture is the tower of clap_htsat.npz"
    tower.requires_grad_(False)
    proj = {k: cases.t(spec.clap_det_weight("clap." + k, shp, 1)) for k, shp in (
        ("audio_projection.0.weight", (512, 1024)), ("audio_projection.0.bias", (512,)),
        ("audio_projection.2.weight", (512, 512)), ("audio_projection.2.bias", (512,)))}
    text = F.normalize(cases.t(spec.det_uniform("clapft_light.text", (B, 512), 4)), dim=-1)
    clap = RefClapStandIn(tower, proj, text)

    vae, sf = make_golden.ref_vae(ns, spec.VAE_DDCONFIG, spec.HIFIGAN_16K_64)
    cfg = spec.LIGHT_UNET_CONFIG
    torch.manual_seed(0)
    model = AudioLCM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                     unet_model_config_path=ns.light_config_path, snr_gamma=5.0, use_edm=True,
                     teacher_guidance_scale=-1, num_diffusion_steps=18, vae=vae, loss_type="mse",
                     target_ema_decay=0.95, ema_decay=0.999)
    loss_mod = RL.CLAPLoss.__new__(RL.CLAPLoss)               # the constructor would build laion_clap and load a checkpoint
    torch.nn.Module.__init__(loss_mod)
    loss_mod.vae, loss_mod.reduction, loss_mod.sr = vae, "instance", 16000
    loss_mod.clap, loss_mod.mse_weight, loss_mod.clap_weight = clap, 1.0, 0.1      # audio_consistency_model.py:100-102
    model.loss_type, model.loss = "clap", loss_mod
    model.teacher_unet.load_state_dict(cases.unet_weights(cfg, False, 0))
    model.student_unet.load_state_dict(cases.unet_weights(cfg, True, 1))
    model.student_target_unet.load_state_dict(cases.unet_weights(cfg, True, 2))
    model.student_ema_unet.load_state_dict(cases.unet_weights(cfg, True, 3))
    P = cases.prompt_states(cfg, B, L, "clapft_light")
    model.get_prompt_embeds = lambda prompt, use_cf, num_samples_per_prompt=1: (
        P["embeds_cf"], P["mask_cf"], P["embeds"], P["mask"])
    model.encode_text_classifier_free = lambda prompt, n: (P["embeds_cf"], P["mask_cf"], P["embeds"], P["mask"])
    z0 = cases.t(spec.det_uniform("clapft_light.z0", (B, 8, H, W), 14)) * 0.9
    gt = cases.t(spec.det_uniform("clapft_light.gt", (B, 160000), 5)) * 0.3

    rec = {}
    o_randint, o_randn_like, o_rand = torch.randint, torch.randn_like, torch.rand

    def randint(*a, **k):
        v = o_randint(*a, **k)
        rec.setdefault("randint", v.clone())
        return v

    def randn_like(x, *a, **k):
        v = o_randn_like(x, *a, **k)
        rec.setdefault("randn_like", v.clone())
        return v

    def rand(*a, **k):
        v = o_rand(*a, **k)
        rec.setdefault("rand", v.clone())
        return v

    inst_rec = {}
    o_forward = loss_mod.forward

    def loss_forward(inp, tgt, gw, cap, use_ema=False):
        v = o_forward(inp, tgt, gw, cap, use_ema=use_ema)
        inst_rec["instance"] = v.detach().clone()
        inst_rec["mse"] = ((inp.float() - tgt.float()) ** 2).reshape(B, -1).mean(1).detach().clone()
        return v
    loss_mod.forward = loss_forward

    model.train()
    torch.manual_seed(4321)
    torch.randint, torch.randn_like, torch.rand = randint, randn_like, rand
    try:
        loss = model(z0, gt, ["a"] * B)
    finally:
        torch.randint, torch.randn_like, torch.rand = o_randint, o_randn_like, o_rand
    loss.backward()

    out = dict(train_loss=np.float64(float(loss)), instance_loss=inst_rec["instance"].numpy(), instance_mse=inst_rec["mse"].numpy(),
               time_inds=rec["randint"].numpy(), noise=rec["randn_like"].numpy(), guidance=rec["rand"].numpy() * 6,
               scale_factor=np.float64(sf))
    names, norms, samples, offsets = [], [], [], [0]
    for k, p in model.student_unet.named_parameters():
        if p.grad is None:
            assert not p.requires_grad, k
            continue
        g = p.grad.detach().reshape(-1)
        names.append(k)
        norms.append(float(g.double().norm()))
        samples.append(g[torch.from_numpy(sample_index(g.numel()))].numpy())
        offsets.append(offsets[-1] + samples[-1].size)
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, dtype=np.float64)
    out["grad_samples"] = np.concatenate(samples).astype(np.float32)
    out["grad_offsets"] = np.array(offsets, dtype=np.int64)
    for p in list(model.teacher_unet.parameters()) + list(vae.parameters()) + list(tower.parameters()):
        assert p.grad is None
    path = os.path.join(HERE, "clapft_light.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; loss", float(loss), "instance", out["instance_loss"], "mse",
          out["instance_mse"], "tensors", len(names), "time_inds", out["time_inds"], "guidance", out["guidance"])


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    main()
