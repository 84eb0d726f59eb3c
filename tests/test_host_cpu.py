"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol that
include/ctta.h declares, the nn.Module mirrors carry the reference's state-dict keys, the
scheduler's host tables match the reference, and the product path fails LOUDLY (no CPU
fallback) when it is handed CPU tensors or the library is missing."""
import os
import re

import numpy as np
import pytest
import torch

import cases
from consistencytta_amd import _native as N
from consistencytta_amd import modules, scheduler, spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ctta.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ctta_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(N.LIB_PATH):
        N.build()
    return N.lib()


def test_library_exports_every_declared_symbol(built_lib):
    declared = _declared_symbols()
    assert len(declared) >= 45
    for name in declared:
        assert hasattr(built_lib, name), "libctta_hip.so does not export %s" % name
        assert name in N.SIGNATURES, "%s has no ctypes signature" % name
    assert sorted(N.SIGNATURES) == declared
    assert built_lib.ctta_version() == 100
    assert built_lib.ctta_conv_gemm_num_variants() >= 4


def test_struct_layouts_match_header_field_order():
    text = open(os.path.join(ROOT, "include", "ctta.h")).read()
    end = text.index("} ctta_conv_desc;")
    body = text[text.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?(void|float|int64_t|int)\s*\*?", "", decl).strip()
        names += [n.strip().lstrip("*") for n in decl.split(",")]
    assert names == [f[0] for f in N.ConvDesc._fields_]
    # ctta_ffn_desc (round 6): same field order, and the same byte size as the C compiler gives it
    end = text.index("} ctta_ffn_desc;")
    body = re.sub(r"/\*.*?\*/", "", text[text.rindex("typedef struct {", 0, end) + len("typedef struct {"):end], flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = re.sub(r"^(const\s+)?(void|float|int64_t|int)\s*\*?", "", decl.strip()).strip()
        if decl:
            names += [n.strip().lstrip("*") for n in decl.split(",")]
    assert names == [f[0] for f in N.FfnDesc._fields_]
    import ctypes
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "s.c")
        open(src, "w").write('#include <stdio.h>\n#include "ctta.h"\nint main(void) { printf("%zu %zu\\n", sizeof(ctta_ffn_desc), sizeof(ctta_conv_desc)); return 0; }\n')
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        ffn_size, conv_size = (int(v) for v in subprocess.check_output([exe]).split())
    assert ffn_size == ctypes.sizeof(N.FfnDesc) and conv_size == ctypes.sizeof(N.ConvDesc)


def test_mirror_state_dict_keys_are_the_references(golden):
    g = golden("unet_tiny")
    m = modules.UNet2DConditionGuidedModel.from_config(cases.TINY_UNET)
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    assert not m.get_parameter("guidance_proj.weight").requires_grad       # embeddings.py:229
    t = modules.UNet2DConditionModel.from_config(cases.TINY_UNET)
    assert all(not k.startswith("guidance") for k in t.state_dict())
    v = modules.AutoencoderKL(ddconfig=cases.TINY_VAE_DD, embed_dim=8, hifigan_config=cases.TINY_HIFIGAN)
    keys = list(v.state_dict().keys())
    # the reference AutoencoderKL's own key list and order (encoder, decoder, quant convs, vocoder)
    assert keys == [str(k) for k in golden("vae_encoder_tiny")["keys"]]
    assert keys[:2] == ["encoder.conv_in.weight", "encoder.conv_in.bias"]
    assert "vocoder.resblocks.14.convs2.2.weight" in keys and "post_quant_conv.bias" in keys
    # decoder-only checkpoints (generation) load; the encoder then refuses to run; unknown keys still raise
    sd = dict(cases.vae_weights(cases.TINY_VAE_DD))
    sd.update(cases.hifigan_weights(cases.TINY_HIFIGAN))
    v.load_state_dict(sd)
    with pytest.raises(RuntimeError, match="decoder-only"):
        v.encode_first_stage(torch.zeros(1, 1, 8, 8))
    v.load_state_dict(dict(sd, **cases.vae_encoder_weights(cases.TINY_VAE_DD)))
    with pytest.raises(RuntimeError):
        v.load_state_dict(dict(sd, bogus=torch.zeros(1)))
    import copy
    m2 = copy.deepcopy(m)                                                   # audio_consistency_model.py:65
    assert list(m2.state_dict().keys()) == list(m.state_dict().keys())


def test_config_validation_mirrors_reference_errors():
    with pytest.raises(ValueError):
        modules.UNet2DConditionGuidedModel(**dict(cases.TINY_UNET, block_out_channels=[32, 64]))
    with pytest.raises(ValueError):
        modules.UNet2DConditionGuidedModel(**dict(cases.TINY_UNET, down_block_types=["Nope"] * 4))


def test_scheduler_host_tables_match_reference(golden):
    g = golden("heun")
    s = scheduler.HeunDiscreteScheduler.from_pretrained("stabilityai/stable-diffusion-2-1", subfolder="scheduler")
    assert s.config.prediction_type == "v_prediction" and s.order == 2
    for n in (1, 2, 18, 200):
        s.set_timesteps(n)
        assert np.array_equal(s.timesteps.numpy(), g["timesteps_%d" % n])
        np.testing.assert_allclose(s.sigmas.numpy(), g["sigmas_%d" % n], rtol=1e-6, atol=0)  # host-CPU fp32
    s.set_timesteps(18)
    assert len(s.timesteps) == 35 and len(s.sigmas) == 36 and s.state_in_first_order
    # last-match semantics of index_for_timestep (mask * arange argmax)
    assert list(s.index_for_timestep(s.timesteps[[0, 1, 2, 34]])) == [0, 2, 2, 34]
    with pytest.raises(AssertionError):
        s.index_for_timestep(123.456)                                       # :143 membership assert


def test_product_path_has_no_cpu_fallback(built_lib):
    m = modules.UNet2DConditionGuidedModel.from_config(cases.TINY_UNET).init_deterministic()
    x, ts, gs, enc, mask = cases.unet_inputs(cases.TINY_UNET, 1, 16, 8, 4, "cpu")
    with pytest.raises(RuntimeError):
        m(x, ts, guidance=gs, encoder_hidden_states=enc, encoder_attention_mask=mask)
    v = modules.AutoencoderKL(ddconfig=cases.TINY_VAE_DD, embed_dim=8, hifigan_config=cases.TINY_HIFIGAN)
    with pytest.raises(RuntimeError):
        v.decode_first_stage(torch.zeros(1, 8, 16, 8))
    s = scheduler.HeunDiscreteScheduler.from_pretrained("x")
    s.set_timesteps(18)
    with pytest.raises(RuntimeError):
        s.scale_model_input(torch.zeros(1, 8, 4, 4), 999.0)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 16, 8), 1.0, guidance=1.0, encoder_hidden_states=enc)   # wrong channels


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no non-HIP fallback"):
        N.lib()


def test_det_generator_is_stable():
    a = spec.det_uniform("some.key", (4, 3), seed=7)
    assert a.dtype == np.float32 and a.shape == (4, 3)
    assert np.array_equal(a, spec.det_uniform("some.key", (4, 3), seed=7))
    assert not np.array_equal(a, spec.det_uniform("some.key", (4, 3), seed=8))
    # frozen values: a change here invalidates every committed fixture
    np.testing.assert_allclose(spec.det_uniform("x", (3,), 0), [-0.597583532333374, 0.6170535087585449, 0.2866472005844116], rtol=0, atol=1e-6)


# ------------------------------------------------------------------------------------------------
# checkpoint loaders of AudioLCM (audio_consistency_model.py:107-204) -- host logic only, CPU tensors
def _lcm():
    from consistencytta_amd.models import AudioLCM
    return AudioLCM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                    unet_model_config_path="tiny_light.json", unet_config=cases.TINY_UNET, snr_gamma=5.0, use_edm=True,
                    teacher_guidance_scale=-1, num_diffusion_steps=18, vae=None, loss_type="mse")


def _fill(m, value):
    with torch.no_grad():
        for p in m.parameters():
            p.fill_(value)


def test_load_state_dict_from_tango_with_and_without_stage1(capsys):
    m = _lcm()
    _fill(m, -1.0)
    tango = {"unet." + k: torch.full_like(v, 2.0) for k, v in m.teacher_unet.state_dict().items()}
    tango["text_encoder.shared.weight"] = torch.zeros(3)            # parked: FLAN-T5 is built lazily
    info = m.load_state_dict_from_tango(tango)
    # the teacher has no guidance branch: those student keys are reported as not loaded, nothing is redundant
    printed = capsys.readouterr().out
    assert "Keys that are not loaded" in printed and "guidance_embedding.linear_1.weight" in printed
    assert all("guidance" in k for k in info.missing_keys)
    assert m._pending_text_encoder_sd.keys() == {"shared.weight"}
    for net in (m.teacher_unet, m.student_unet, m.student_target_unet, m.student_ema_unet):
        for k, p in net.named_parameters():
            assert float(p.flatten()[0]) == (-1.0 if k.startswith("guidance") else 2.0), k
    assert not any(p.requires_grad for p in m.student_target_unet.parameters())
    # stage 1 given: teacher from TANGO, all three students from stage 1's student_ema_* (guidance branch included)
    _fill(m, -1.0)
    stage1 = {"student_ema_unet." + k: torch.full_like(v, 5.0) for k, v in m.student_unet.state_dict().items()}
    stage1.update({"student_unet." + k: torch.full_like(v, 9.0) for k, v in m.student_unet.state_dict().items()})
    m.load_state_dict_from_tango({k: v for k, v in tango.items() if k.startswith("unet.")}, stage1)
    assert all(float(p.flatten()[0]) == 2.0 for p in m.teacher_unet.parameters())
    for net in (m.student_unet, m.student_target_unet, m.student_ema_unet):
        assert all(float(p.flatten()[0]) == 5.0 for p in net.parameters())
    # an unknown key is refused like in the reference
    with pytest.raises(AssertionError, match="Redundant keys"):
        m.load_state_dict_from_tango(dict(tango, **{"unet.bogus.weight": torch.zeros(1)}))


def test_load_pretrained_converts_legacy_names():
    m = _lcm()
    _fill(m, -1.0)
    stu = m.student_unet.state_dict()
    legacy = {"consistency_unet." + k: torch.full_like(v, 1.0) for k, v in stu.items()}
    legacy.update({"consistency_ema_unet." + k: torch.full_like(v, 2.0) for k, v in stu.items()})
    legacy.update({"diffusion_unet." + k: torch.full_like(v, 4.0) for k, v in m.teacher_unet.state_dict().items()})
    legacy["vae.decoder.conv_in.weight"] = torch.zeros(1)             # never loaded
    m.load_pretrained(legacy)
    val = lambda net: {float(p.flatten()[0]) for p in net.parameters()}
    # no slow EMA in the checkpoint: the fast EMA seeds BOTH shadows; every student_ema tensor was overwritten
    assert val(m.student_unet) == {1.0} and val(m.student_target_unet) == {2.0} and val(m.student_ema_unet) == {2.0}
    assert val(m.teacher_unet) == {4.0}
    # with a slow EMA, it wins for student_ema_* whatever the key order
    _fill(m, -1.0)
    slow = {"consistency_slow_ema_unet." + k: torch.full_like(v, 3.0) for k, v in stu.items()}
    m.load_pretrained(dict(legacy, **slow))
    assert val(m.student_ema_unet) == {3.0} and val(m.student_target_unet) == {2.0}
    _fill(m, -1.0)
    m.load_pretrained(dict(slow, **legacy))
    assert val(m.student_ema_unet) == {3.0}
    # current names pass through; unknown names are refused
    m.load_pretrained({k: torch.full_like(v, 6.0) for k, v in m.state_dict().items()})
    assert val(m.student_ema_unet) == {6.0}
    with pytest.raises(AssertionError, match="Redundant keys"):
        m.load_pretrained(dict(legacy, **{"consistency_unet.nope": torch.zeros(1)}))


def test_flat_alias_checks_and_schedule_state():
    from consistencytta_amd import optim
    m = modules.UNet2DConditionGuidedModel.from_config(cases.TINY_UNET).init_deterministic()
    flat = m.flatten_parameters_()
    g = m.flat_grad_()
    assert m.flat_is_current() and m.grads_alias_flat()
    for p in m.parameters():
        p.grad = None                              # torch's default zero_grad(set_to_none=True)
    assert not m.grads_alias_flat()
    m.realias_grads_()
    assert m.grads_alias_flat() and m._flat_grad is g
    # the per-step sampled form (FusedAdamW.step): first call walks everything, later ones sample -- and a parameter
    # frozen AFTER the cache was built does not make every n / k-th step raise: the sample defers to the full walk, which filters
    # on requires_grad at call time; a hand-assigned foreign gradient is found by the full walk of every 64th call at the latest
    assert m.grads_alias_flat_sampled() and m.grads_alias_flat_sampled()
    ps = [p for p in m.parameters() if p.requires_grad]
    ps[3].requires_grad_(False)
    ps[3].grad = None
    assert all(m.grads_alias_flat_sampled() for _ in range(2 * len(ps) // 8 + 2))
    ps[3].requires_grad_(True)
    m.realias_grads_()
    ps[5].grad = torch.zeros_like(ps[5])           # one gradient re-pointed by hand
    seen = [m.grads_alias_flat_sampled() for _ in range(70)]
    assert not all(seen) and seen.index(False) < 70
    m.realias_grads_()
    assert m.grads_alias_flat()
    m.double().float()                             # re-homes p.data: the flat buffer is stale now
    assert not m.flat_is_current()
    assert m.flatten_parameters_() is not flat and m.flat_is_current()

    class _Opt:
        param_groups = [dict(lr=1e-3)]
    sch = optim.WarmupSchedule(_Opt(), "linear", num_warmup_steps=4, num_training_steps=20)
    for _ in range(7):
        sch.step()
    o2 = _Opt()
    o2.param_groups = [dict(lr=1e-3)]
    s2 = optim.WarmupSchedule(o2, "constant")
    s2.load_state_dict(sch.state_dict())
    assert s2.get_last_lr() == sch.get_last_lr() and s2.last_step == 7
    s2.step(), sch.step()
    assert s2.get_last_lr() == sch.get_last_lr()


def test_optimizer_and_scheduler_files_are_torch_layout(tmp_path):
    """optimizer.bin / scheduler.bin must be what the reference writes -- torch.optim.AdamW.state_dict()
    over student_unet.parameters() (tools/train_utils.py:38-39,59-63) and LambdaLR.state_dict() (get_scheduler, :77-81) --
    so that a reference run resumes here and the other way round.  Both directions against the real torch classes."""
    from transformers import get_scheduler
    from consistencytta_amd import optim
    m = _lcm()
    net = m.student_unet
    net.init_deterministic(seed=3)
    params = list(net.parameters())
    names = [k for k, _ in net.named_parameters()]
    frozen = set(net._frozen_keys)
    assert frozen and all(not p.requires_grad for k, p in net.named_parameters() if k in frozen)
    # --- a reference-side optimizer after two updates (parameters without a gradient get no state, as in the reference)
    ref = torch.optim.AdamW(params, lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-4, eps=1e-8)
    rs = get_scheduler(name="linear", optimizer=ref, num_warmup_steps=2, num_training_steps=40)
    g = torch.Generator().manual_seed(0)
    for _ in range(2):
        for k, p in zip(names, params):
            p.grad = None if k in frozen else torch.randn(p.shape, generator=g) * 1e-2
        ref.step()
        rs.step()
    torch.save(ref.state_dict(), tmp_path / "optimizer.bin")
    torch.save(rs.state_dict(), tmp_path / "scheduler.bin")
    for p in params:
        p.grad = None
    ours = optim.FusedAdamW(net, lr=3e-5)
    sch = optim.WarmupSchedule(ours, "linear", num_warmup_steps=2, num_training_steps=40)
    ours.load_state_dict(torch.load(tmp_path / "optimizer.bin"))
    sch.load_state_dict(torch.load(tmp_path / "scheduler.bin"))
    assert ours.step_count == 2 and sch.last_step == 2 and sch.get_last_lr() == rs.get_last_lr()
    assert ours.param_groups[0]["weight_decay"] == 1e-4 and ours.param_groups[0]["betas"] == (0.9, 0.999)
    off = 0
    for i, (k, p) in enumerate(net.named_parameters()):   # the flat moments are the per-parameter ones, trainable order
        if k in frozen:
            continue
        st = ref.state_dict()["state"][i]
        assert torch.equal(ours.exp_avg[off:off + p.numel()].view(p.shape), st["exp_avg"]), k
        assert torch.equal(ours.exp_avg_sq[off:off + p.numel()].view(p.shape), st["exp_avg_sq"]), k
        off += p.numel()
    assert off == ours.n
    # --- and back: our files load into the torch classes and continue identically
    torch.save(ours.state_dict(), tmp_path / "optimizer2.bin")
    torch.save(sch.state_dict(), tmp_path / "scheduler2.bin")
    ref2 = torch.optim.AdamW(params, lr=7e-5)
    rs2 = get_scheduler(name="linear", optimizer=ref2, num_warmup_steps=2, num_training_steps=40)
    ref2.load_state_dict(torch.load(tmp_path / "optimizer2.bin"))
    rs2.load_state_dict(torch.load(tmp_path / "scheduler2.bin"))
    a, b = ref.state_dict(), ref2.state_dict()
    assert sorted(a["state"]) == sorted(b["state"]) and a["param_groups"][0]["params"] == b["param_groups"][0]["params"]
    for i in a["state"]:
        for key in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(torch.as_tensor(a["state"][i][key]), torch.as_tensor(b["state"][i][key])), (i, key)
    assert rs2.last_epoch == 2 and rs2.get_last_lr() == rs.get_last_lr()
    rs.step(), rs2.step(), sch.step()
    assert rs2.get_last_lr() == rs.get_last_lr() == sch.get_last_lr()
    # a state dict for a different model is refused like torch refuses it
    bad = ours.state_dict()
    bad["param_groups"][0]["params"] = bad["param_groups"][0]["params"][:-1]
    with pytest.raises(ValueError, match="doesn't match the size"):
        ours.load_state_dict(bad)
    # the private layout of rounds 1-2 still loads
    ours.load_state_dict({"step": 5, "exp_avg": torch.zeros(ours.n), "exp_avg_sq": torch.ones(ours.n),
                          "param_groups": [dict(lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)]})
    assert ours.step_count == 5 and float(ours.exp_avg_sq[3]) == 1.0


def test_ddpm_step_host_tables_match_reference(golden):
    """DDPMScheduler.step's per-sample coefficient tables (scheduling_ddpm.py:232-265,319-333) vs the reference's own
    step outputs: the HIP kernel only evaluates a*x + b*y with these numbers (GPU twin: test_models_gpu.py)."""
    g = golden("ddpm_step")
    x = cases.t(spec.det_uniform("ddpm.x", (4, 8, 16, 4), 1)) * 2
    v = cases.t(spec.det_uniform("ddpm.v", (4, 8, 16, 4), 3))
    sd21 = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                prediction_type="v_prediction", clip_sample=False)
    for tag, kw, steps in (("v_full", sd21, None), ("v_50", sd21, 50),
                           ("eps_clip", dict(sd21, prediction_type="epsilon", clip_sample=True), 20),
                           ("v_large", dict(sd21, variance_type="fixed_large"), 10)):
        s = scheduler.DDPMScheduler(**kw)
        if steps:
            s.set_timesteps(steps)
        t = torch.from_numpy(g[tag + "_t"])
        a_t, b_t, c_x0, c_xt, std = (q.reshape(-1, 1, 1, 1) for q in s._coeffs(t))
        if kw["prediction_type"] == "epsilon":
            x0 = ((x - b_t ** 0.5 * v) / a_t ** 0.5).clamp(-1, 1)
        else:
            x0 = a_t ** 0.5 * x - b_t ** 0.5 * v
        prev = c_x0 * x0 + c_xt * x + std * torch.from_numpy(g[tag + "_noise"])     # the fixture's noise is 0 where none is added
        np.testing.assert_allclose(x0.numpy(), g[tag + "_x0"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(prev.numpy(), g[tag + "_prev"], rtol=2e-5, atol=2e-6)
    with pytest.raises(NotImplementedError):
        scheduler.DDPMScheduler(**dict(sd21, variance_type="learned_range"))._coeffs(torch.tensor([5]))
    with pytest.raises(RuntimeError):
        scheduler.DDPMScheduler(**sd21).step(v, 3, x)                       # CPU tensors: no CPU path


def test_run_directory_formats_round_trip(tmp_path):
    """summary.jsonl (train.py:304-305, tools/train_utils.py:240-241, inference.py:114) and the accelerate
    `save_state` layout with the distilled model in pytorch_model_2.bin (train.py:380,498-505; inference.py:152-153)."""
    import argparse
    import json
    from consistencytta_amd import checkpoint as ck
    from consistencytta_amd import audio, optim
    run = str(tmp_path / "run")
    args = argparse.Namespace(stage=2, text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                              unet_model_name=None, unet_model_config="tiny_light.json", snr_gamma=5.0,
                              freeze_text_encoder=True, uncondition=False, use_edm=True, use_karras=False, use_lora=False,
                              target_ema_decay=0.95, ema_decay=0.999, num_diffusion_steps=18, teacher_guidance_scale=-1,
                              loss_type="mse", finetune_vae=False, output_dir=run)
    ck.write_args_summary(run, args)
    ck.append_summary(run, {"epoch": 1, "step": 10, "validation_loss": 0.5, "train_loss": 0.25})
    text = open(os.path.join(run, "summary.jsonl")).read()
    assert text.split("\n")[0] == json.dumps(dict(vars(args))) and text.endswith("}\n\n")
    assert os.path.isdir(os.path.join(run, "outputs"))
    ta = ck.read_original_args(os.path.join(run, "summary.jsonl"))
    assert ta.stage == 2 and ta.hf_model is None and ta.unet_model_config == "tiny_light.json" and ta.nonexistent is None
    assert ck.read_summary(os.path.join(run, "summary.jsonl"))[1]["validation_loss"] == 0.5

    m = _lcm()
    for i, net in enumerate((m.teacher_unet, m.student_unet, m.student_target_unet, m.student_ema_unet)):
        net.init_deterministic(seed=10 + i)
    vae = modules.AutoencoderKL(ddconfig=cases.TINY_VAE_DD, embed_dim=8, hifigan_config=cases.TINY_HIFIGAN).init_deterministic(3)
    stft = audio.TacotronSTFT()
    opt = optim.FusedAdamW(m.student_unet, lr=1e-5)
    opt.step_count = 7
    opt.exp_avg.fill_(0.5)
    sch = optim.WarmupSchedule(opt, "linear", num_warmup_steps=3, num_training_steps=30)
    for _ in range(5):
        sch.step()
    ckdir = ck.save_state(os.path.join(run, "epoch_1"), (vae, stft, m), opt, sch)
    assert sorted(os.listdir(ckdir)) == ["optimizer.bin", "pytorch_model.bin", "pytorch_model_1.bin", "pytorch_model_2.bin",
                                         "random_states_0.pkl", "scheduler.bin"]
    saved = torch.load(os.path.join(ckdir, "pytorch_model_2.bin"))
    assert list(saved.keys()) == list(m.state_dict().keys())          # plain state dict, the model's own key order
    assert list(saved)[0].startswith("teacher_unet.") and any(k.startswith("student_target_unet.") for k in saved)

    # inference.py's path: arguments from summary.jsonl + pytorch_model_2.bin through load_pretrained
    m2, ta2 = ck.build_model_from_run(os.path.join(ckdir, "pytorch_model_2.bin"), os.path.join(run, "summary.jsonl"),
                                      vae=None, stage=2, unet_config=cases.TINY_UNET)
    assert not m2.training and ta2.loss_type == "mse"
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(AssertionError, match="Stage mismatch"):
        ck.build_model_from_run(os.path.join(ckdir, "pytorch_model_2.bin"), os.path.join(run, "summary.jsonl"), stage=1,
                                unet_config=cases.TINY_UNET)

    # train.py --resume_from_checkpoint: everything comes back, RNG streams included
    m3 = _lcm()
    vae3 = modules.AutoencoderKL(ddconfig=cases.TINY_VAE_DD, embed_dim=8, hifigan_config=cases.TINY_HIFIGAN)
    opt3 = optim.FusedAdamW(m3.student_unet, lr=3e-5)
    sch3 = optim.WarmupSchedule(opt3, "constant")
    ck.load_state(ckdir, (vae3, audio.TacotronSTFT(), m3), opt3, sch3)
    assert opt3.step_count == 7 and float(opt3.exp_avg[0]) == 0.5 and sch3.last_step == 5
    assert sch3.get_last_lr() == sch.get_last_lr()
    for (k, a), (_, b) in zip(vae.state_dict().items(), vae3.state_dict().items()):
        assert torch.equal(a, b), k
    assert torch.equal(m3.student_ema_unet.state_dict()["conv_in.weight"], m.student_ema_unet.state_dict()["conv_in.weight"])
    torch.manual_seed(5)
    ck.save_state(os.path.join(run, "rng"), ())
    want = torch.rand(4)
    torch.manual_seed(6)
    ck.load_state(os.path.join(run, "rng"), ())
    assert torch.equal(torch.rand(4), want)


def test_device_code_has_no_swizzled_packed_fp32():
    """`v_pk_*_f32 ... op_sel:[...]` (low lane fed from the high dword of a source pair) misbehaves on the MI355X boxes
    when another kernel's MFMA waves share the CU (tools/pk_hazard.py); build.sh compiles without the SLP vectoriser that
    emits it and this scan of the built library's ISA keeps hand-written vector code from bringing it back."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sp = importlib.util.spec_from_file_location("check_isa", os.path.join(root, "tools", "check_isa.py"))
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    hits, n_insn, n_obj = mod.scan(os.path.join(root, "consistencytta_amd", "libctta_hip.so"))
    assert n_obj >= 10 and n_insn > 100000, "the scan did not find the device code (%d objects, %d instructions)" % (n_obj, n_insn)
    assert not hits, hits[:5]


def test_no_kernel_lost_occupancy_against_the_committed_table():
    """tests/golden/kernel_occupancy.json holds, per kernel of the built library, [VGPRs, AGPRs, spilled VGPRs, waves per
    SIMD] (`python tools/check_isa.py --regs consistencytta_amd/libctta_hip.so tests/golden/kernel_occupancy.json`).  A
    kernel that fits FEWER waves per SIMD than the table says, or spills more, fails here: round 4 lost 1.3 % of the
    generation step to ONE register (an epilogue block compiled into every conv tile took the 128x128x32 and the 8-wave
    256x128x32 tiles from 128 to 129 VGPRs = one workgroup less per CU; fused GEGLU 796 -> 557 TFLOP/s).  Regenerate the
    table when a change is meant to trade occupancy.  The three tiles built to sit at exactly 128 are named explicitly."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sp = importlib.util.spec_from_file_location("check_isa", os.path.join(root, "tools", "check_isa.py"))
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    now = mod.occupancy_table(os.path.join(root, "consistencytta_amd", "libctta_hip.so"))
    want = json.load(open(os.path.join(root, "tests", "golden", "kernel_occupancy.json")))
    assert len(now) >= 190 and mod.waves_per_simd(128) == 4 and mod.waves_per_simd(129) == 3 and mod.waves_per_simd(100, 32) == 4
    lost = {k: (want[k], now[k]) for k in want if k in now and (now[k][3] < want[k][3] or now[k][2] > want[k][2])}
    assert not lost, "kernels that lost waves per SIMD or spill more than the committed table: %r" % lost
    tile = lambda *t: "_Z16conv_gemm_kernelI" + "".join("Li%dE" % x for x in t) + "Ev10ConvParams"
    for t in ((128, 128, 32, 2, 2, 1, 2), (128, 128, 32, 2, 2, 2, 2), (256, 128, 32, 4, 2, 2, 2)):
        assert now[tile(*t)][0] <= 128 and now[tile(*t)][3] == 4, (t, now[tile(*t)])
    conv = {k: v for k, v in now.items() if k.startswith("_Z16conv_gemm_kernelILi")}
    assert len(conv) >= 36 and all(v[2] == 0 for v in conv.values()), {k: v for k, v in conv.items() if v[2]}


def test_no_kernel_gained_conservative_store_waits():
    """tests/golden/kernel_vmcnt0.json holds the number of `s_waitcnt vmcnt(0)` instructions of every kernel of the built
    library (`python tools/isa_vmcnt0_scan.py --json consistencytta_amd/libctta_hip.so tests/golden/kernel_vmcnt0.json`).
    Global stores inside divergent `if (row < M)` blocks of a straight-line epilogue make the compiler wait vmcnt(0) in front
    of every later read-back, i.e. every row sweep waits for the previous store's acknowledgement (the bf16 wide-store
    epilogue of conv_gemm in round 2, its fp32 twin in round 5: 0.7 ms of a distillation step).  A kernel whose count grows
    by more than a tenth (and by more than 4) over the committed table fails here; regenerate the table when it is meant."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sp = importlib.util.spec_from_file_location("isa_vmcnt0_scan", os.path.join(root, "tools", "isa_vmcnt0_scan.py"))
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    now = {nm: w for w, _, _, nm in mod.scan(os.path.join(root, "consistencytta_amd", "libctta_hip.so"))}
    want = json.load(open(os.path.join(root, "tests", "golden", "kernel_vmcnt0.json")))
    assert len(now) >= 190
    grown = {k: (want[k], now[k]) for k in want if k in now and now[k] > want[k] + max(4, want[k] // 10)}
    assert not grown, "kernels with more `s_waitcnt vmcnt(0)` than the committed table: %r" % grown


def test_eval_metrics_match_the_reference_functions(golden):
    """consistencytta_amd.audioldm_eval.calculate_{fid,isc,kid,kl} (host arithmetic, as in the reference) against the values
    the reference's own audioldm_eval/metrics/*.py returned on the same seeded features (tests/golden/make_golden_eval.py):
    same dictionary keys, same random streams, KL pairing by base name whatever the directory order."""
    from consistencytta_amd import audioldm_eval as E
    g = golden("eval_suite")
    X = cases.eval_metric_inputs()
    fd = lambda f: {"f": f}
    assert E.calculate_fid(fd(X["fid1"]), fd(X["fid2"]), "f") == {"frechet_distance": pytest.approx(float(g["fid"]), rel=1e-9)}
    r = E.calculate_isc(fd(X["isc"]), "f", rng_seed=2020, samples_shuffle=True, splits=10)
    assert [r["inception_score_mean"], r["inception_score_std"]] == pytest.approx(list(g["isc"]), rel=1e-12)
    r = E.calculate_kid(fd(X["kid1"]), fd(X["kid2"]), subsets=100, subset_size=90, degree=3, gamma=None, coef0=1, rng_seed=2020,
                        feat_layer_name="f")
    assert [r["kernel_inception_distance_mean"], r["kernel_inception_distance_std"]] == pytest.approx(list(g["kid"]), rel=1e-12)
    names, perm = cases.eval_kl_names(50)
    d1 = {"f": X["kl1"], "file_path_": ["/gen/" + n for n in names]}
    d2 = {"f": X["kl2"][perm], "file_path_": ["/gt/" + names[i] for i in perm]}
    r, kl_ref, paths = E.calculate_kl(d1, d2, "f", True)
    assert [r["kullback_leibler_divergence_sigmoid"], r["kullback_leibler_divergence_softmax"]] == pytest.approx(list(g["kl"]), rel=1e-6)
    np.testing.assert_allclose(kl_ref.numpy(), g["kl_ref"], rtol=1e-5)
    assert paths == names
    r, a, b = E.calculate_kl(d1, d2, "f", same_name=False)      # unpaired directories: the reference reports -1
    assert r == {"kullback_leibler_divergence_sigmoid": -1.0, "kullback_leibler_divergence_softmax": -1.0} and a is None
    # subset larger than a set: clipped with a warning, like the reference
    r = E.calculate_kid(fd(X["kid1"]), fd(X["kid2"]), subsets=3, subset_size=1000, degree=3, gamma=None, coef0=1, rng_seed=1,
                        feat_layer_name="f")
    assert np.isfinite(r["kernel_inception_distance_mean"])


def test_eval_wav_reader(tmp_path):
    """datasets/load_mel.py:17-29 with scipy.io.wavfile: int16 scaling, stereo mix-down, integer-ratio decimation by striding,
    mean removal, the 2 s zero padding; a non-integer rate ratio (resampy in the reference) is refused."""
    from scipy.io import wavfile
    from consistencytta_amd import audioldm_eval as E
    rng = np.random.RandomState(0)
    x = (rng.rand(48000, 2) * 2 - 1) * 0.5
    wavfile.write(str(tmp_path / "a.wav"), 48000, (x * 32767).astype(np.int16))
    got = E.read_centered_wav(str(tmp_path / "a.wav"), 16000)
    mono = ((x * 32767).astype(np.int16).astype(np.float64) / 32768.0).mean(1)[::3]
    np.testing.assert_allclose(got, mono - mono.mean(), atol=1e-12)
    wavfile.write(str(tmp_path / "b.wav"), 22050, (x[:, 0] * 32767).astype(np.int16))
    with pytest.raises(RuntimeError):
        E.read_centered_wav(str(tmp_path / "b.wav"), 16000)
    ds = E.WaveDataset(str(tmp_path), sr=48000, target_length=50)
    w, name = ds[0]
    assert name == "a.wav" and tuple(w.shape) == (1, 32000) and float(w[0, 24000:].abs().max()) == 0.0


def test_cnn14_loads_a_released_checkpoint_layout():
    """`Cnn14.load_state_dict` takes the key set of the released `Cnn14_16k_mAP=0.438.pth` (`["model"]`): the module's own
    parameters and BatchNorm buffers, torchlibrosa's frozen STFT / mel matrices, `num_batches_tracked` counters -- strictly,
    with or without the structural front-end entries; an unknown key is refused."""
    from consistencytta_amd import audioldm_eval as E
    m = E.Cnn14(features_list=["2048", "logits"])
    table = spec.cnn14_param_spec(spec.CNN14_16K_CONFIG)
    assert list(k for k, _ in m.named_parameters()) == list(table)
    sd = {k: torch.from_numpy(spec.cnn14_det_weight("ck." + k, shape, 1)) for k, shape in table.items()}
    for k in list(sd):
        if k.endswith("running_var"):
            sd[k.replace("running_var", "num_batches_tracked")] = torch.tensor(7)
    m.load_state_dict(sd, strict=True)
    assert torch.equal(m.get_parameter("conv_block3.bn2.running_var"), sd["conv_block3.bn2.running_var"])
    core = {k: v for k, v in sd.items() if k not in spec.CNN14_STRUCTURAL}
    m.load_state_dict(core, strict=True)                       # a state dict without the frozen front-end matrices
    with pytest.raises(RuntimeError):
        m.load_state_dict(dict(core, **{"conv_block7.conv1.weight": torch.zeros(1)}), strict=True)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 32000))                               # CPU tensors: there is no CPU path


def test_scheduler_resumes_an_n_process_reference_run():
    """accelerate's AcceleratedScheduler steps the LambdaLR `num_processes` times per optimizer update, and
    tools/train_utils.py:77 multiplies the warm-up by num_processes for that reason; an N-GPU reference scheduler.bin
    therefore holds last_epoch = N x updates.  WarmupSchedule(steps_per_update=N) continues exactly that curve."""
    from transformers import get_scheduler
    from consistencytta_amd import optim
    N_PROC, warm, total = 4, 3, 50
    p = torch.nn.Parameter(torch.zeros(2))
    ref = torch.optim.AdamW([p], lr=3e-5)
    rs = get_scheduler(name="linear", optimizer=ref, num_warmup_steps=warm * N_PROC, num_training_steps=total * N_PROC)
    for _ in range(5):                       # 5 optimizer updates of the 4-process reference run
        for _ in range(N_PROC):
            rs.step()
    sd = rs.state_dict()
    assert sd["last_epoch"] == 5 * N_PROC

    class _Opt:
        param_groups = [{"lr": 3e-5}]
    sch = optim.WarmupSchedule(_Opt(), "linear", num_warmup_steps=warm * N_PROC, num_training_steps=total * N_PROC,
                               steps_per_update=N_PROC)
    sch.load_state_dict(sd)
    assert sch.last_step == 20 and sch.get_last_lr() == rs.get_last_lr()
    for _ in range(7):                       # both continue: one update = N_PROC LambdaLR steps there, one step() here
        for _ in range(N_PROC):
            rs.step()
        sch.step()
        assert abs(sch.get_last_lr()[0] - rs.get_last_lr()[0]) <= 1e-12 * 3e-5 + 1e-18
    assert sch.state_dict()["last_epoch"] == rs.state_dict()["last_epoch"] == 48


def test_eval_captions_come_from_the_dataset_json_like_the_reference(tmp_path):
    """`EvaluationHelper.main(dataset_json_path, generated_files_path, groundtruth_path, mel_path=...)` keeps the
    reference's positional order (audioldm_eval/eval.py:336-349; callers inference.py:230, evaluate_existing.py:54) and reads
    the captions itself: line i of the json-lines file belongs to output_<i>.wav (tools/t2a_dataset.py:79-87,118-119)."""
    import inspect
    import json
    from consistencytta_amd import audioldm_eval as E
    path = tmp_path / "test.json"
    with open(path, "w") as f:
        for i, c in enumerate(["a dog barks", "rain on a roof", "a car passes"]):
            f.write(json.dumps({"captions": c, "location": "/data/%d.wav" % i, "dataset": "audiocaps"}) + "\n")
    caps = E.EvaluationHelper.captions_from_dataset_json(str(path))
    assert caps == {"output_0.wav": "a dog barks", "output_1.wav": "rain on a roof", "output_2.wav": "a car passes"}
    assert list(inspect.signature(E.EvaluationHelper.main).parameters)[:7] == [
        "self", "dataset_json_path", "generated_files_path", "groundtruth_path", "mel_path", "target_length", "limit_num"]
    assert list(inspect.signature(E.EvaluationHelper.calculate_metrics).parameters)[:8] == [
        "self", "dataset_json_path", "generate_files_path", "groundtruth_path", "mel_path", "same_name", "target_length",
        "limit_num"]
    with pytest.raises(AssertionError):
        E.EvaluationHelper.captions_from_dataset_json(str(tmp_path / "missing.json"))
