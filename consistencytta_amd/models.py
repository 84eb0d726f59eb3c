"""Task-level API mirrors: `AudioLCM` (models/audio_consistency_model.py:19-548 +
models/audio_distilled_model.py) and the inference-only twin `ConsistencyTTA`
(easy_inference/consistencytta.py:12-200).

Everything numerical below the text encoder runs in the HIP engines
(`modules.UNet2DCondition*Model`, `modules.AutoencoderKL`, `scheduler.HeunDiscreteScheduler`,
fused CFG / loss / EMA kernels).  The FLAN-T5 text encoder stays a PyTorch module, as in the
reference (it is frozen and outside the hot path); because neither box has
network access it can be injected (`text_encoder=`, `tokenizer=`) instead of downloaded.

Scope: `inference` (student, multi-step, teacher Heun loop), `_query_teacher`, `update_ema`,
`forward` (distillation loss; in training mode the returned loss carries a grad_fn whose backward
runs the HIP engine's own backward pass and fills `student_unet` `.grad`s, so the reference loop
`loss = model(...); loss.backward(); optimizer.step()` works unchanged on one GPU), and
`train_step` = the fused multi-GPU step (backward, RCCL gradient all-reduce over the flat gradient
buffer, fused AdamW, two-shadow EMA) that replaces accelerate's DDP wrapper (train.py:377-379),
whose autograd hooks cannot see gradients produced outside autograd.
"""
import os
from copy import deepcopy
from time import time

import numpy as np
import torch
from torch import nn

from . import _native as N
from . import dist_util
from .modules import AutoencoderKL, UNet2DConditionGuidedModel, UNet2DConditionModel
from .scheduler import DDPMScheduler, HeunDiscreteScheduler


def randn_tensor(shape, generator=None, device=None, dtype=None):
    """diffusers.utils.randn_tensor: noise on `device` from the global (or given) RNG."""
    return torch.randn(shape, generator=generator, device=device, dtype=dtype)


def optimizer_tail(source_model, shadow_models, decay_consts, optimizer, lr_scheduler, grad_scale, do_step):
    """The end of a training step, tools/train_utils.py:177-183: optimizer.step() (skipped on a NaN loss, :167-172) ->
    lr_scheduler.step() -> optimizer.zero_grad() -> update_ema().  With the fused AdamW on `source_model` and every network
    still in the flat buffers `do_ema_update` validated on an earlier step, the three device passes are ONE launch
    (FusedAdamW.step_zero_ema; bit-identical state, tests/test_train_gpu.py); anything else -- a torch optimizer, the first
    step, re-homed parameters, CTTA_FUSED_TAIL=0 -- takes the three calls."""
    nets = [source_model] + list(shadow_models)
    sig = tuple((id(m), getattr(m, "_rehome_count", 0)) for m in nets)
    flats = [getattr(m, "_flat", None) for m in nets]
    if (hasattr(optimizer, "step_zero_ema") and optimizer.module is source_model and os.environ.get("CTTA_FUSED_TAIL", "1") != "0"
            and getattr(source_model, "_ema_validated", None) == sig and all(f is not None for f in flats)
            and all(m.flat_is_current() for m in nets) and optimizer.can_fuse_tail(flats[1:])):
        for d in decay_consts:
            assert 0 <= d <= 1
        optimizer.step_zero_ema(flats[1:], decay_consts, grad_scale=grad_scale, do_step=do_step)
        if do_step and lr_scheduler is not None:
            lr_scheduler.step()
        for m in shadow_models:
            m.mark_weights_changed()
        return
    if do_step:
        optimizer.step(grad_scale=grad_scale)
        if lr_scheduler is not None:
            lr_scheduler.step()
    optimizer.zero_grad()
    do_ema_update(source_model, shadow_models, decay_consts)


def do_ema_update(source_model, shadow_models, decay_consts):
    """tools/train_utils.py:255-282 with one fused pass per parameter for up to two shadows:
    shadow += (1 - decay) * (param - shadow).  Buffers: the U-Net mirrors have none."""
    assert len(shadow_models) == len(decay_consts)
    assert 1 <= len(shadow_models) <= 2, "the fused kernel updates one or two shadows"
    L_ = N.lib()
    nets = [source_model] + list(shadow_models)
    flats = [getattr(m, "_flat", None) for m in nets]
    # steady state of the training loop: the same networks, still living in their flat buffers (O(1) checks: the key
    # comparison below ran when this tuple was recorded, and flat_is_current() only walks after a re-homing event)
    sig = tuple((id(m), getattr(m, "_rehome_count", 0)) for m in nets)
    if (getattr(source_model, "_ema_validated", None) == sig and all(f is not None for f in flats)
            and all(m.flat_is_current() for m in nets)):
        for d in decay_consts:
            assert 0 <= d <= 1
        with torch.cuda.device(flats[0].device):
            N.check(L_.ctta_ema_update2(N.ptr(flats[0]), N.ptr(flats[1]), float(decay_consts[0]),
                                        N.ptr(flats[2]) if len(flats) > 2 else N.c_void_p(0),
                                        float(decay_consts[1]) if len(flats) > 2 else 0.0, flats[0].numel(),
                                        N.stream_ptr()))
        for m in shadow_models:
            m.mark_weights_changed()
        return
    src = dict(source_model.named_parameters())
    shadows = [dict(m.named_parameters()) for m in shadow_models]
    for sh, d in zip(shadows, decay_consts):
        assert 0 <= d <= 1
        assert src.keys() == sh.keys()
    # the one-launch path needs every network to STILL live in its flat buffer (a `.to()` / `.float()` after
    # flatten_parameters_ re-homes p.data and would leave the kernel updating stale memory)
    if all(f is not None and f.numel() == flats[0].numel() for f in flats) and all(m.flat_is_current() for m in nets):
        # every network lives in one flat buffer with the same layout: one launch for the whole model
        fa = flats[1]
        fb = flats[2] if len(flats) > 2 else None
        with torch.cuda.device(flats[0].device):
            N.check(L_.ctta_ema_update2(N.ptr(flats[0]), N.ptr(fa), float(decay_consts[0]),
                                        N.ptr(fb) if fb is not None else N.c_void_p(0),
                                        float(decay_consts[1]) if fb is not None else 0.0, flats[0].numel(),
                                        N.stream_ptr()))
        for m in shadow_models:
            m.mark_weights_changed()
        source_model._ema_validated = sig
        return
    for name, p in src.items():
        a = shadows[0][name]
        b = shadows[1][name] if len(shadows) > 1 else None
        if not p.is_cuda:
            raise N.CttaError("EMA update needs CUDA(ROCm) parameters; there is no CPU path")
        N.check(L_.ctta_ema_update2(N.ptr(p.detach()), N.ptr(a.detach()), float(decay_consts[0]),
                                    N.ptr(b.detach()) if b is not None else N.c_void_p(0),
                                    float(decay_consts[1]) if b is not None else 0.0, p.numel(), N.stream_ptr()))
        # bump version counters so the engines re-pack the shadows
        a.detach().add_(0)
        if b is not None:
            b.detach().add_(0)


class _DistillLoss(torch.autograd.Function):
    """Gives the distillation loss a grad_fn: autograd's only job is to call the engine's backward."""

    @staticmethod
    def forward(ctx, anchor, loss, model, pred, target, sig, gamma):
        ctx.model, ctx.saved = model, (pred, target, sig, gamma)
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out):
        pred, target, sig, gamma = ctx.saved
        ctx.model._student_backward(pred, target, sig, gamma, float(grad_out))
        return (None,) * 7


class AudioDistilledModel(nn.Module):
    """models/audio_distilled_model.py: text encoding, CFG teacher query, EMA bookkeeping."""

    def __init__(self, text_encoder_name, scheduler_name, unet_model_name=None, unet_model_config_path=None,
                 snr_gamma=None, freeze_text_encoder=True, use_lora=False, ema_decay=0.999,
                 teacher_guidance_scale=3, unet_config=None, text_encoder=None, tokenizer=None, **kwargs):
        super().__init__()
        assert unet_model_name is None, "pretrained hub U-Nets need network access"
        assert not use_lora, "LoRA is unused by train.sh and not built"
        self.text_encoder_name = text_encoder_name
        self.scheduler_name = scheduler_name
        self.unet_model_config_path = unet_model_config_path
        self.snr_gamma = snr_gamma
        self.freeze_text_encoder = freeze_text_encoder
        self.ema_decay = ema_decay
        self.teacher_guidance_scale = teacher_guidance_scale
        self.max_rand_guidance_scale = 6
        self.use_teacher_cf_guidance = teacher_guidance_scale == -1 or teacher_guidance_scale > 1.0

        if unet_config is None:
            unet_config = UNet2DConditionModel.load_config(unet_model_config_path)
        self.teacher_unet = UNet2DConditionModel.from_config(unet_config, subfolder="unet")
        self.student_unet = UNet2DConditionGuidedModel.from_config(unet_config, subfolder="unet")
        self.student_ema_unet = deepcopy(self.student_unet)
        self.teacher_unet.eval().requires_grad_(False)
        self.student_ema_unet.eval().requires_grad_(False)

        self.tokenizer = tokenizer
        self.text_encoder = text_encoder
        if self.text_encoder is not None:
            self.text_encoder.eval()
            self.text_encoder.requires_grad_(False)

    @property
    def device(self):
        return next(self.student_unet.parameters()).device

    def _require_text_encoder(self):
        if self.text_encoder is None or self.tokenizer is None:
            try:
                from transformers import AutoTokenizer
                from .text_encoder import T5EncoderModel   # HIP engine behind transformers' T5EncoderModel interface
                self.tokenizer = AutoTokenizer.from_pretrained(self.text_encoder_name)
                self.text_encoder = T5EncoderModel.from_pretrained(self.text_encoder_name).to(self.device)
                self.text_encoder.eval().requires_grad_(False)
                pending = getattr(self, "_pending_text_encoder_sd", None)
                if pending:         # text_encoder.* entries a checkpoint brought before the encoder existed
                    self.text_encoder.load_state_dict(pending)
                    self._pending_text_encoder_sd = None
            except Exception as e:  # no network / no cache
                raise RuntimeError(
                    "FLAN-T5 (%s) is not available offline; pass text_encoder= and tokenizer= "
                    "or call the *_from_embeds entry points: %s" % (self.text_encoder_name, e))

    # audio_distilled_model.py:194-248
    @torch.no_grad()
    def encode_text(self, prompt, max_length=None, padding=True):
        self._require_text_encoder()
        device = self.device
        if max_length is None:
            max_length = self.tokenizer.model_max_length
        batch = self.tokenizer(prompt, max_length=max_length, padding=padding, truncation=True,
                               return_tensors="pt")
        input_ids = batch.input_ids.to(device)
        attention_mask = batch.attention_mask.to(device)
        prompt_embeds = self.text_encoder(input_ids=input_ids, attention_mask=attention_mask)[0]
        return prompt_embeds, (attention_mask == 1).to(device)

    @torch.no_grad()
    def encode_text_classifier_free(self, prompt, num_samples_per_prompt):
        cond_embeds, cond_mask = self.encode_text(prompt)
        cond_embeds = cond_embeds.repeat_interleave(num_samples_per_prompt, 0)
        cond_mask = cond_mask.repeat_interleave(num_samples_per_prompt, 0)
        uncond_embeds, uncond_mask = self.encode_text([""] * len(prompt), max_length=cond_embeds.shape[1],
                                                      padding="max_length")
        uncond_embeds = uncond_embeds.repeat_interleave(num_samples_per_prompt, 0)
        uncond_mask = uncond_mask.repeat_interleave(num_samples_per_prompt, 0)
        return (torch.cat([uncond_embeds, cond_embeds]), torch.cat([uncond_mask, cond_mask]),
                cond_embeds, cond_mask)

    def get_prompt_embeds(self, prompt, use_cf_guidance, num_samples_per_prompt=1):
        if isinstance(prompt, dict):  # pre-computed text states (synthetic benchmarks / tests)
            return prompt["embeds_cf"], prompt["mask_cf"], prompt["embeds"], prompt["mask"]
        return self.encode_text_classifier_free(prompt, num_samples_per_prompt)

    def check_eval_mode(self):
        for model, name in ((self.teacher_unet, "teacher_unet"), (self.student_ema_unet, "student_ema_unet")):
            assert model.training is False, f"The {name} is not in eval mode."
            for p in model.parameters():
                assert p.requires_grad is False, f"The {name} is not frozen."

    # audio_distilled_model.py:286-322
    def _query_teacher(self, z_scaled, t, prompt_embeds, prompt_mask, guidance_scale=None, reuse_text=None):
        if not torch.is_tensor(t):
            t = torch.tensor(t)
        if len(t.reshape(-1)) != 1 and self.use_teacher_cf_guidance:
            t = torch.cat([t] * 2)
        z_cat = torch.cat([z_scaled] * 2) if self.use_teacher_cf_guidance else z_scaled
        # reuse_text: the cross-attention K / V of the text states come from the handle's text cache when the caller (or, in
        # eager mode, the module itself: same tensor objects, unmodified) knows they are those of the previous query
        pred = self.teacher_unet(z_cat, t, prompt_embeds, encoder_attention_mask=prompt_mask, reuse_text=reuse_text).sample
        if self.use_teacher_cf_guidance:
            B = z_scaled.shape[0]
            if self.teacher_guidance_scale == -1:
                w = guidance_scale if torch.is_tensor(guidance_scale) else torch.tensor(guidance_scale)
                w = w.to(device=pred.device, dtype=torch.float32).reshape(-1)
                if w.numel() == 1:
                    w = w.expand(B)
            else:
                w = torch.full((B,), float(self.teacher_guidance_scale), dtype=torch.float32, device=pred.device)
            w = w.contiguous()
            out = torch.empty_like(pred[:B])
            N.check(N.lib().ctta_cfg_combine(N.ptr(pred[:B]), N.ptr(pred[B:]), N.ptr(w), N.ptr(out), B,
                                             out[0].numel(), N.stream_ptr()))
            pred = out
        # the reference asserts `not noise_pred.isnan().any()` here (a device->host sync per query);
        # enable with CTTA_NAN_CHECKS=1
        return pred


class AudioLCM(AudioDistilledModel):
    def __init__(self, text_encoder_name, scheduler_name, unet_model_name=None, unet_model_config_path=None,
                 snr_gamma=None, freeze_text_encoder=True, uncondition=False, use_edm=False, use_karras=False,
                 use_lora=False, target_ema_decay=.95, ema_decay=.999, num_diffusion_steps=18,
                 teacher_guidance_scale=1, vae=None, loss_type="mse", clap_module=None, **kwargs):
        super().__init__(text_encoder_name=text_encoder_name, scheduler_name=scheduler_name,
                         unet_model_name=unet_model_name, unet_model_config_path=unet_model_config_path,
                         snr_gamma=snr_gamma, freeze_text_encoder=freeze_text_encoder, use_lora=use_lora,
                         ema_decay=ema_decay, teacher_guidance_scale=teacher_guidance_scale, **kwargs)
        assert use_edm, "only the Heun/EDM path (use_edm, train.sh:33) is built; DDIM is §8f rank 3"
        assert not use_karras, "Karras sigmas are unused by the shipped scripts and not built"
        assert loss_type in ("mse", "mel", "stft", "clap"), "unknown loss_type %r" % (loss_type,)
        self.uncondition = uncondition
        self.use_edm = use_edm
        self.target_ema_decay = target_ema_decay
        self.num_diffusion_steps = num_diffusion_steps
        self.lightweight = "light" in (unet_model_config_path or "light")
        self.student_target_unet = deepcopy(self.student_unet)
        self.student_target_unet.eval().requires_grad_(False)
        self.noise_scheduler = HeunDiscreteScheduler.from_pretrained(self.scheduler_name, subfolder="scheduler")
        self.noise_scheduler.set_timesteps(self.num_diffusion_steps)
        self.vae = vae
        self.loss_type = loss_type
        # audio_consistency_model.py:92-105.  'mse' stays on the fused kernels (ctta_snr_mse_loss / _grad); the
        # perceptual losses run through the differentiable decode (losses.py).
        from . import losses as L
        if loss_type == "mel":
            self.loss = L.MelLoss(vae=self.vae, reduction="instance")
        elif loss_type == "stft":
            self.loss = L.MultiResolutionSTFTLoss(vae=self.vae, reduction="instance", fft_sizes=[1024, 2048, 512],
                                                  hop_sizes=[120, 240, 50], win_lengths=[600, 1200, 240],
                                                  window="hann_window", factor_sc=0.1, factor_mag=0.1, factor_mse=.8)
        elif loss_type == "clap":
            # the reference loads ckpt/music_audioset_epoch_15_esc_90.14.pt here; `clap_module=` supplies ready towers
            self.loss = L.CLAPLoss(vae=self.vae, reduction="instance", mse_weight=1., clap_weight=.1, clap=clap_module)
        else:
            self.loss = None
        if self.loss is not None and self.vae is None:
            raise ValueError("loss_type=%r decodes latents: pass vae=" % (loss_type,))

    def _perceptual_loss(self, pred, target, gt_wav, prompt, sig, gamma):
        """get_loss (audio_consistency_model.py:250-266) for loss_type != 'mse': instance losses through the
        differentiable decode, SNR weights clamp(sigma^-2, max=snr_gamma), batch mean."""
        inst = self.loss(pred, target, gt_wav, prompt)
        if gamma and gamma > 0:
            inst = inst * torch.clamp(sig.to(inst.device, torch.float32) ** -2, max=float(gamma))
        return inst.mean()

    def train(self, mode=True):
        super().train(mode)
        self.teacher_unet.eval()
        self.student_ema_unet.eval()
        self.student_target_unet.eval()
        if self.vae is not None:
            self.vae.eval()
        if self.text_encoder is not None:
            self.text_encoder.eval()
        return self

    def eval(self):
        return self.train(False)

    def compute_snr(self, timesteps, t_indices):
        return self.noise_scheduler.sigmas[t_indices] ** (-2)

    def update_ema(self):
        assert self.training, "EMA update should only be called during training"
        do_ema_update(self.student_unet, [self.student_target_unet, self.student_ema_unet],
                      [self.target_ema_decay, self.ema_decay])

    def _optimizer_tail(self, optimizer, lr_scheduler, grad_scale, do_step):
        assert self.training, "EMA update should only be called during training"
        optimizer_tail(self.student_unet, [self.student_target_unet, self.student_ema_unet],
                       [self.target_ema_decay, self.ema_decay], optimizer, lr_scheduler, grad_scale, do_step)

    def check_eval_mode(self):
        super().check_eval_mode()
        assert self.student_target_unet.training is False, "The student_target_unet is not in eval mode."
        for p in self.student_target_unet.parameters():
            assert p.requires_grad is False, "The student_target_unet is not frozen."

    def _load_converted(self, new_sd):
        """Tail shared by both loaders (audio_consistency_model.py:129-147,187-204): a strict load first; when it
        fails, a non-strict one that prints the keys it could not fill and refuses unknown ones (vae / loss keys
        exempt).  FLAN-T5 is constructed lazily here (`_require_text_encoder`), so `text_encoder.*` entries of a
        checkpoint are parked until the encoder exists instead of being reported as redundant."""
        if self.text_encoder is None:
            parked = {k[len("text_encoder."):]: v for k, v in new_sd.items() if k.startswith("text_encoder.")}
            if parked:
                self._pending_text_encoder_sd = parked
                new_sd = {k: v for k, v in new_sd.items() if not k.startswith("text_encoder.")}
        try:
            return self.load_state_dict(new_sd, strict=True)
        except Exception:
            print("Strict loading failed. The loaded state_dict may not match the target model. "
                  "This is okay if 'Keys that are not loaded' is an empty list.")
            info = self.load_state_dict(new_sd, strict=False)
            missing = [k for k in info.missing_keys if "vae" not in k and "loss." not in k]
            redundant = [k for k in info.unexpected_keys if "vae" not in k and "loss." not in k]
            print(f"Keys that are not loaded: {missing}")
            assert len(redundant) == 0, f"Redundant keys in state_dict: {info.unexpected_keys}"
            return info

    def load_pretrained(self, state_dict, strict=True):
        """audio_consistency_model.py:160-204: checkpoints written by older implementations name the networks
        consistency_unet / consistency_ema_* / consistency_slow_ema_* / diffusion_unet and the STFT loss `loss.*`;
        the fast EMA (`consistency_ema_`) also seeds `student_ema_*` unless the checkpoint brings its own slow EMA;
        `vae.*` entries are never loaded.  (`strict` is accepted and, as in the reference, not consulted.)"""
        new_sd = {}
        for key, val in state_dict.items():
            if "consistency_unet" in key:
                new_sd["student_unet" + key.split("consistency_unet")[-1]] = val
            elif "consistency_ema_" in key:
                aft = key.split("consistency_ema_")[-1]
                new_sd["student_target_" + aft] = val
                new_sd.setdefault("student_ema_" + aft, val)
            elif "consistency_slow_ema_" in key:
                new_sd["student_ema_" + key.split("consistency_slow_ema_")[-1]] = val
            elif "diffusion_unet" in key:
                new_sd["teacher_unet" + key.split("diffusion_unet")[-1]] = val
            elif "loss." in key and "vae." not in key:
                new_sd["stft_loss." + key.split("loss.")[-1]] = val
            elif "vae." not in key:
                new_sd[key] = val
        return self._load_converted(new_sd)

    def load_state_dict_from_tango(self, tango_state_dict, stage1_state_dict=None):
        """audio_consistency_model.py:107-158: TANGO's `unet.*` becomes the teacher; the three students start from
        the teacher when there is no stage-1 checkpoint, otherwise all three start from stage 1's `student_ema_*`
        weights.  Every other TANGO key (text encoder, ...) passes through under its own name."""
        students = ("student", "student_target", "student_ema")
        new_sd = {}
        for key, val in tango_state_dict.items():
            if "unet" in key and "_unet" not in key:
                new_sd["teacher_" + key] = val
                if stage1_state_dict is None:
                    for m in students:
                        new_sd[m + "_" + key] = val
            else:
                new_sd[key] = val
        if stage1_state_dict is not None:
            for key, val in stage1_state_dict.items():
                if "student_ema" in key:
                    aft = key.split("student_ema_")[-1]
                    for m in students:
                        new_sd[m + "_" + aft] = val
        info = self._load_converted(new_sd)
        self.student_target_unet.requires_grad_(False)
        self.student_ema_unet.requires_grad_(False)
        self.teacher_unet.requires_grad_(False)
        return info

    # ---- distillation loss, audio_consistency_model.py:239-427
    def forward(self, z_0, gt_wav, prompt, validation_mode=False, run_teacher=True, time_inds=None,
                gaussian_noise=None, guidance_scale=None, **kwargs):
        """Training mode (validation_mode == 0) with grad enabled: the loss has a grad_fn; `.backward()`
        accumulates the student U-Net's parameter gradients through the engine's backward pass."""
        want_grad = (not validation_mode and self.training and torch.is_grad_enabled()
                     and any(p.requires_grad for p in self.student_unet.parameters()))
        with torch.no_grad():
            out = self._forward_impl(z_0, gt_wav, prompt, validation_mode, run_teacher, time_inds, gaussian_noise,
                                     guidance_scale, want_grad)
        if not want_grad:
            return out
        loss, pred, target, sig, gamma = out
        return _DistillLoss.apply(self._grad_anchor(), loss, self, pred, target, sig, gamma)

    def _grad_anchor(self):
        a = getattr(self, "_anchor", None)
        if a is None or a.device != self.device:
            a = self._anchor = torch.zeros((), device=self.device, requires_grad=True)
        return a

    def _side_stream(self, dev):
        st = getattr(self, "_side", None)
        if st is None or st.device != torch.device(dev):
            st = self._side = torch.cuda.Stream(device=dev)
        return st

    def _student_backward(self, pred, target, sig, gamma, loss_scale=1.0, on_block_done=None):
        """d loss / d pred of get_loss (audio_consistency_model.py:250-266) -> engine backward."""
        if isinstance(target, tuple):   # perceptual loss: torch differentiates loss(pred) down to the latent
            leaf, graph = target
            with torch.enable_grad():
                (g,) = torch.autograd.grad(graph, leaf, torch.full_like(graph, float(loss_scale)))
            self.student_unet.backward(grad_output=g, on_block_done=on_block_done)
            return
        B, C, H, W = pred.shape
        d = torch.empty(B, H * W, 8, dtype=torch.bfloat16, device=pred.device)
        with torch.cuda.device(pred.device):
            N.check(N.lib().ctta_snr_mse_grad(N.ptr(pred), N.ptr(target.contiguous()), N.ptr(sig), float(gamma),
                                              float(loss_scale), B, C, H * W, 8, N.ptr(d), N.stream_ptr()))
        self.student_unet.backward(grad_output_nhwc=d, on_block_done=on_block_done)

    def prepare_training(self, lr=3e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, broadcast=True):
        """Flat parameter buffers for the student family, rank-0 weights on every rank (DDP's wrap-time
        broadcast) and the fused AdamW over the student (tools/train_utils.py:59-63)."""
        from .optim import FusedAdamW
        self.student_unet.enable_training = True
        for m in (self.student_unet, self.student_target_unet, self.student_ema_unet):
            flat = m.flatten_parameters_()
            if broadcast:
                dist_util.broadcast_(flat)
                m.mark_weights_changed()
        return FusedAdamW(self.student_unet, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    def train_step(self, z_0, prompt, optimizer, lr_scheduler=None, gt_wav=None, skip_nan=True,
                   accumulation_steps=1, **fw):
        """One micro-step of tools/train_utils.py:150-190: loss -> backward, and on every `accumulation_steps`-th
        call (accelerator.accumulate / sync_gradients, train.py:269) gradient all-reduce (RCCL, overlapped with
        that backward) -> AdamW -> lr schedule -> zero_grad -> EMA.  In between, gradients only accumulate locally
        (DDP's no_sync).  The loss of every micro-step is scaled by 1/accumulation_steps like accelerate does.
        Returns the (unscaled) loss as a Python float (the reference reads `loss.item()` every step too)."""
        assert self.training, "train_step needs model.train()"
        self._micro = getattr(self, "_micro", 0) + 1
        boundary = self._micro % max(1, int(accumulation_steps)) == 0
        if not boundary:
            with torch.no_grad():
                loss, pred, target, sig, gamma = self._forward_impl(
                    z_0, gt_wav, prompt, False, True, fw.pop("time_inds", None), fw.pop("gaussian_noise", None),
                    fw.pop("guidance_scale", None), True)
                self._student_backward(pred, target, sig, gamma, 1.0 / accumulation_steps)
            return float(loss.item())
        with torch.no_grad():
            loss, pred, target, sig, gamma = self._forward_impl(
                z_0, gt_wav, prompt, False, True, fw.pop("time_inds", None), fw.pop("gaussian_noise", None),
                fw.pop("guidance_scale", None), True)
            # gradient all-reduce (RCCL) overlapped with the backward pass, block by block
            cdt = getattr(self, "allreduce_dtype", None)
            buckets = dist_util.GradientBuckets(optimizer.grad, self.student_unet.block_ranges(), compress=cdt,
                                                twin=optimizer.grad_twin(cdt) if hasattr(optimizer, "grad_twin") else None)
            nan_any = dist_util.AnyRankFlag(torch.isnan(loss))     # all ranks skip together (or none)
            self._student_backward(pred, target, sig, gamma, 1.0 / max(1, int(accumulation_steps)),
                                   buckets.ready if buckets.enabled else None)
            world = buckets.wait()
            value = float(loss.item())
            # train_utils.py:167-172: a NaN loss skips the update (not the zero_grad, not the EMA)
            self._optimizer_tail(optimizer, lr_scheduler, 1.0 / world, not (skip_nan and nan_any.result()))
        return value

    def _forward_impl(self, z_0, gt_wav, prompt, validation_mode, run_teacher, time_inds, gaussian_noise,
                      guidance_scale, want_grad):
        self.check_eval_mode()
        assert validation_mode >= 0
        sch = self.noise_scheduler
        dev = z_0.device
        B = z_0.shape[0]
        embeds_cf, mask_cf, embeds, mask = self.get_prompt_embeds(prompt, self.use_teacher_cf_guidance, 1)
        avail = sch._timesteps_host
        order = 2
        if validation_mode != 0:
            ti = len(avail) - 1 - int(validation_mode * order)
            assert ti >= 0
            inds = torch.full((B,), ti, dtype=torch.int64)
        elif time_inds is not None:
            inds = time_inds.to("cpu", torch.int64)
        else:
            inds = torch.randint(0, (len(avail) - 1) // order, (B,)) * order
        t_np1 = torch.from_numpy(avail[inds.numpy()])
        t_n = torch.from_numpy(avail[(inds + order).numpy()])
        noise = gaussian_noise if gaussian_noise is not None else torch.randn_like(z_0)
        z_noisy = sch.add_noise(z_0, noise, t_np1)
        z_gauss = noise * sch.init_noise_sigma.to(dev)
        last_mask = (t_np1 == float(avail.max())).reshape(-1, 1, 1, 1).to(dev)
        z_np1 = torch.where(last_mask, z_gauss, z_noisy)
        z_np1_scaled = sch.scale_model_input(z_np1, t_np1)
        assert sch.state_in_first_order
        if self.teacher_guidance_scale == -1:
            if guidance_scale is None:
                guidance_scale = torch.rand(B) * self.max_rand_guidance_scale
            guidance_scale = guidance_scale.to(dev)
        else:
            guidance_scale = None
        # The student's training forward depends only on the noised input, not on the teacher: enqueue it on a second
        # HIP stream so that it runs beside the two teacher queries and the target network (the thin deep-level
        # launches of a batch-9/18 forward leave CUs idle).  Every engine handle owns its arena and split-K workspace,
        # so the two streams share no scratch memory.  CTTA_TWO_STREAM=0 keeps everything on one stream.
        # (Two kernels of different handles on one CU once made the teacher's output vary from run to run: an
        # SLP-generated `v_pk_fma_f32 ... op_sel:[0,1,0]` in the fp32 MLP kernel reads a wrong dword beside another
        # kernel's MFMA waves -- tools/pk_hazard.py.  The library is built without that instruction form and
        # tools/check_isa.py keeps it out; test_distillation_step_full_batch_is_deterministic... guards the overlap.)
        side_pred = None
        if (want_grad and validation_mode == 0 and z_0.is_cuda and os.environ.get("CTTA_TWO_STREAM", "1") != "0"):
            cur = torch.cuda.current_stream(dev)
            side = self._side_stream(dev)
            w_stu = guidance_scale if guidance_scale is not None else float(self.teacher_guidance_scale)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                side_pred = self.student_unet.forward_train(z_np1_scaled, t_np1, w_stu, embeds, mask)
            side_pred.record_stream(cur)
        v1 = self._query_teacher(z_np1_scaled, t_np1, embeds_cf, mask_cf, guidance_scale)
        zhat_n = sch.step(v1, t_np1, z_np1).prev_sample
        zhat_n_scaled = sch.scale_model_input(zhat_n, t_n)
        v2 = self._query_teacher(zhat_n_scaled, t_n, embeds_cf, mask_cf, guidance_scale)
        zhat_n = sch.step(v2, t_n, zhat_n).prev_sample
        zhat_n_scaled = sch.scale_model_input(zhat_n, t_n)
        assert sch.state_in_first_order
        w = guidance_scale if guidance_scale is not None else float(self.teacher_guidance_scale)
        target = self.student_target_unet(zhat_n_scaled, t_n, guidance=w, encoder_hidden_states=embeds,
                                          encoder_attention_mask=mask).sample
        sig = torch.from_numpy(sch._sigmas_host[inds.numpy()]).to(dev)

        def mse(a, b, sigma=None, gamma=0.0):
            if self.loss is not None and sigma is not None:   # the consistency loss proper (get_loss); plain MSEs stay MSE
                with torch.no_grad():
                    return self._perceptual_loss(a, b, gt_wav, prompt, sigma, gamma)
            inst = torch.empty(B, dtype=torch.float32, device=dev)
            out = torch.empty(1, dtype=torch.float32, device=dev)
            N.check(N.lib().ctta_snr_mse_loss(N.ptr(a.contiguous()), N.ptr(b.contiguous()), N.ptr(sigma), float(gamma),
                                              N.ptr(inst), N.ptr(out), B, a[0].numel(), N.stream_ptr()))
            return out[0]

        if validation_mode != 0:   # :354-405
            from_np1 = self.student_target_unet(z_np1_scaled, t_np1, guidance=w, encoder_hidden_states=embeds,
                                                encoder_attention_mask=mask).sample
            if run_teacher:        # the teacher continues from t_n down to 0 with the full Heun schedule
                for j in range(int(inds[0]) + order, len(avail)):
                    t = float(avail[j])
                    z_in = sch.scale_model_input(zhat_n, t)
                    pred = self._query_teacher(z_in, t, embeds_cf, mask_cf, guidance_scale)
                    zhat_n = sch.step(pred, t, zhat_n).prev_sample
                sch.prev_derivative = sch.dt = sch.sample = None
            loss_w_gt = mse(from_np1, z_0)
            loss_w_teacher = mse(from_np1, zhat_n)
            loss_consis = mse(from_np1, target, sig, self.snr_gamma or 0.0)
            loss_teacher = mse(zhat_n, z_0)
            return loss_w_gt, loss_w_teacher, loss_consis, loss_teacher
        target = torch.where((t_n == 0).reshape(-1, 1, 1, 1).to(dev), z_0, target)
        if want_grad:
            if side_pred is not None:
                torch.cuda.current_stream(dev).wait_stream(self._side_stream(dev))
                pred = side_pred
            else:
                pred = self.student_unet.forward_train(z_np1_scaled, t_np1, w, embeds, mask)
            gamma = self.snr_gamma or 0.0
            if self.loss is not None:   # keep the graph from pred to the loss; _student_backward differentiates it
                leaf = pred.detach().requires_grad_(True)
                with torch.enable_grad():
                    graph = self._perceptual_loss(leaf, target, gt_wav, prompt, sig, gamma)
                return graph.detach(), pred, (leaf, graph), sig, gamma
            return mse(pred, target, sig, gamma), pred, target.contiguous(), sig, gamma
        pred = self.student_unet(z_np1_scaled, t_np1, guidance=w, encoder_hidden_states=embeds,
                                 encoder_attention_mask=mask).sample
        return mse(pred, target, sig, self.snr_gamma or 0.0)

    # ---- generation, audio_consistency_model.py:429-548
    @torch.no_grad()
    def inference(self, prompt, inference_scheduler, guidance_scale_input=3, guidance_scale_post=1, num_steps=20,
                  use_edm=False, num_samples=1, use_ema=True, query_teacher=False, num_teacher_steps=18,
                  return_all=False, noise=None, graph_teacher=False):
        """`graph_teacher=True` runs the Heun teacher loop as replays of ONE captured hipGraph (a full 2nd-order
        step = 2 CFG teacher queries); results are identical to the eager loop."""
        self.check_eval_mode()
        sch = inference_scheduler
        use_cf = guidance_scale_post > 1.
        t0 = time()
        embeds_cf, mask_cf, embeds, mask = self.get_prompt_embeds(prompt, True, num_samples)
        enc_stu, mask_stu = (embeds_cf, mask_cf) if use_cf else (embeds, mask)
        enc_tea, mask_tea = (embeds_cf, mask_cf) if self.use_teacher_cf_guidance else (embeds, mask)
        dev = embeds.device
        B = embeds.shape[0]
        C = self.student_target_unet.config.in_channels
        if noise is None:
            noise = randn_tensor((B, C, 256, 16), device=dev, dtype=torch.float32)
        time_embed = time() - t0
        unet = self.student_ema_unet if use_ema else self.student_target_unet

        def calc_zhat_0(z_n, t):
            z_in = torch.cat([z_n] * 2) if use_cf else z_n
            z_in = sch.scale_model_input(z_in, t)
            zh = unet(z_in, t, guidance=guidance_scale_input, encoder_hidden_states=enc_stu,
                      encoder_attention_mask=mask_stu).sample
            if use_cf:
                u, c = zh.chunk(2)
                zh = (1 - guidance_scale_post) * u + guidance_scale_post * c
            return zh

        t1 = time()
        sch.set_timesteps(18, device=dev)
        z_N = noise * sch.init_noise_sigma
        zhat_0 = calc_zhat_0(z_N, float(sch._timesteps_host[0]))
        sch.set_timesteps(num_steps, device=dev)
        for t in sch._timesteps_host[1::2]:
            zhat_n = sch.add_noise(zhat_0, torch.randn_like(zhat_0), float(t))
            zhat_0 = calc_zhat_0(zhat_n, float(t))
        time_stu = time() - t1
        zhat_tea, time_tea = None, None
        if query_teacher:
            t2 = time()
            sch.set_timesteps(num_teacher_steps, device=dev)
            zhat_tea = noise * sch.init_noise_sigma
            if graph_teacher:
                zhat_tea = self._teacher_loop_graphed(sch, zhat_tea, enc_tea, mask_tea, guidance_scale_input)
            else:
                for t in sch._timesteps_host:
                    z_in = sch.scale_model_input(zhat_tea, float(t))
                    pred = self._query_teacher(z_in, float(t), enc_tea, mask_tea, guidance_scale_input)
                    zhat_tea = sch.step(pred, float(t), zhat_tea).prev_sample
            sch.prev_derivative = sch.dt = sch.sample = None
            time_tea = time() - t2 + time_embed
        if return_all:
            return zhat_0, zhat_tea, time_stu + time_embed, time_tea
        return zhat_0


class _WeightedMSELoss(torch.autograd.Function):
    """grad_fn of AudioGDM's loss: backward = d(weighted MSE)/d(pred) -> the student engine's backward pass."""

    @staticmethod
    def forward(ctx, anchor, loss, model, pred, target, weights):
        ctx.model, ctx.saved = model, (pred, target, weights)
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out):
        pred, target, weights = ctx.saved
        ctx.model._student_backward(pred, target, weights, float(grad_out))
        return (None,) * 6


class AudioGDM(AudioDistilledModel):
    """Stage-1 guided distillation (models/audio_guided_model.py:16-244): the CFG teacher is
    distilled into the guidance-conditioned student at DDPM noise levels; loss = MSE to the teacher's prediction with
    min-SNR-gamma weights.  `forward` returns a loss with a grad_fn (engine backward), `train_step` is the fused
    data-parallel step, `inference` runs the reference's DDIM loop."""

    def __init__(self, text_encoder_name, scheduler_name, unet_model_name=None, unet_model_config_path=None,
                 snr_gamma=None, freeze_text_encoder=True, use_lora=False, ema_decay=.999, teacher_guidance_scale=3,
                 **kwargs):
        super().__init__(text_encoder_name=text_encoder_name, scheduler_name=scheduler_name,
                         unet_model_name=unet_model_name, unet_model_config_path=unet_model_config_path,
                         snr_gamma=snr_gamma, freeze_text_encoder=freeze_text_encoder, use_lora=use_lora,
                         ema_decay=ema_decay, teacher_guidance_scale=teacher_guidance_scale, **kwargs)
        self.noise_scheduler = DDPMScheduler.from_pretrained(self.scheduler_name, subfolder="scheduler")

    def train(self, mode=True):
        """audio_distilled_model.py:154-163: the frozen teacher / EMA / text encoder stay in eval mode."""
        super().train(mode)
        self.teacher_unet.eval()
        self.student_ema_unet.eval()
        if self.text_encoder is not None:
            self.text_encoder.eval()
        return self

    def eval(self):
        return self.train(False)

    def compute_snr(self, timesteps):
        """alpha^2 / sigma^2 with alpha = sqrt(alphas_cumprod[t]) (audio_distilled_model.py:165-191), host side."""
        ac = self.noise_scheduler.alphas_cumprod[torch.as_tensor(timesteps).reshape(-1).to("cpu", torch.int64)]
        return ((ac ** 0.5) / ((1.0 - ac) ** 0.5)) ** 2

    def update_ema(self):
        assert self.training, "EMA update should only be called during training"
        do_ema_update(self.student_unet, [self.student_ema_unet], [self.ema_decay])

    def _optimizer_tail(self, optimizer, lr_scheduler, grad_scale, do_step):
        assert self.training, "EMA update should only be called during training"
        optimizer_tail(self.student_unet, [self.student_ema_unet], [self.ema_decay], optimizer, lr_scheduler, grad_scale, do_step)

    def _grad_anchor(self):
        a = getattr(self, "_anchor", None)
        if a is None or a.device != self.device:
            a = self._anchor = torch.zeros((), device=self.device, requires_grad=True)
        return a

    def _loss_weights(self, t_n, dev):
        if self.snr_gamma is None:
            return None
        snr = self.compute_snr(t_n).reshape(-1)
        trunc = torch.clamp(snr, max=self.snr_gamma)
        if self.noise_scheduler.config.prediction_type == "v_prediction":
            w = trunc / (snr + 1)
        elif self.noise_scheduler.config.prediction_type == "epsilon":
            w = trunc / snr
        else:
            raise ValueError("Unknown prediction type.")
        return w.to(device=dev, dtype=torch.float32).contiguous()

    def _student_backward(self, pred, target, weights, loss_scale=1.0, on_block_done=None):
        B, C, H, W = pred.shape
        d = torch.empty(B, H * W, 8, dtype=torch.bfloat16, device=pred.device)
        with torch.cuda.device(pred.device):
            N.check(N.lib().ctta_weighted_mse_grad(N.ptr(pred), N.ptr(target.contiguous()), N.ptr(weights),
                                                   float(loss_scale), B, C, H * W, 8, N.ptr(d), N.stream_ptr()))
        self.student_unet.backward(grad_output_nhwc=d, on_block_done=on_block_done)

    def _forward_impl(self, z_0, prompt, want_grad, time_inds=None, gaussian_noise=None, guidance_scale=None):
        self.check_eval_mode()
        sch = self.noise_scheduler
        dev = z_0.device
        B = z_0.shape[0]
        embeds_cf, mask_cf, embeds, mask = self.get_prompt_embeds(prompt, self.use_teacher_cf_guidance, 1)
        avail = sch.timesteps                                               # host int64, 999..0
        inds = time_inds.to("cpu", torch.int64) if time_inds is not None else torch.randint(0, len(avail), (B,))
        t_n = avail[inds]
        noise = gaussian_noise if gaussian_noise is not None else torch.randn_like(z_0)
        z_noisy = sch.add_noise(z_0, noise, t_n)
        z_gauss = noise * sch.init_noise_sigma
        last = (t_n == int(avail.max())).reshape(-1, 1, 1, 1).to(dev)
        z_n = torch.where(last, z_gauss, z_noisy)
        z_n_scaled = sch.scale_model_input(z_n, t_n)
        if self.teacher_guidance_scale == -1:
            if guidance_scale is None:
                guidance_scale = torch.rand(B) * self.max_rand_guidance_scale
            guidance_scale = guidance_scale.to(dev)
        else:
            guidance_scale = None
        t_f = t_n.to(torch.float32)
        teacher = self._query_teacher(z_n_scaled, t_f, embeds_cf, mask_cf, guidance_scale)
        w = guidance_scale if guidance_scale is not None else float(self.teacher_guidance_scale)
        if want_grad:
            pred = self.student_unet.forward_train(z_n_scaled, t_f, w, embeds, mask)
        else:
            pred = self.student_unet(z_n_scaled, t_f, guidance=w, encoder_hidden_states=embeds,
                                     encoder_attention_mask=mask).sample
        weights = self._loss_weights(t_n, dev)
        inst = torch.empty(B, dtype=torch.float32, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        N.check(N.lib().ctta_weighted_mse_loss(N.ptr(pred.contiguous()), N.ptr(teacher.contiguous()), N.ptr(weights),
                                               N.ptr(inst), N.ptr(out), B, pred[0].numel(), N.stream_ptr()))
        return out[0], pred, teacher.contiguous(), weights

    def forward(self, z_0, prompt, time_inds=None, gaussian_noise=None, guidance_scale=None, **kwargs):
        want_grad = (self.training and torch.is_grad_enabled()
                     and any(p.requires_grad for p in self.student_unet.parameters()))
        with torch.no_grad():
            loss, pred, target, weights = self._forward_impl(z_0, prompt, want_grad, time_inds, gaussian_noise,
                                                             guidance_scale)
        if not want_grad:
            return loss
        return _WeightedMSELoss.apply(self._grad_anchor(), loss, self, pred, target, weights)

    def prepare_training(self, lr=3e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, broadcast=True):
        from .optim import FusedAdamW
        self.student_unet.enable_training = True
        for m in (self.student_unet, self.student_ema_unet):
            flat = m.flatten_parameters_()
            if broadcast:
                dist_util.broadcast_(flat)
                m.mark_weights_changed()
        return FusedAdamW(self.student_unet, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    def train_step(self, z_0, prompt, optimizer, lr_scheduler=None, skip_nan=True, **fw):
        """loss -> backward (+ overlapped RCCL gradient all-reduce) -> AdamW -> schedule -> zero_grad -> EMA."""
        assert self.training, "train_step needs model.train()"
        with torch.no_grad():
            loss, pred, target, weights = self._forward_impl(z_0, prompt, True, fw.pop("time_inds", None),
                                                             fw.pop("gaussian_noise", None), fw.pop("guidance_scale", None))
            cdt = getattr(self, "allreduce_dtype", None)
            buckets = dist_util.GradientBuckets(optimizer.grad, self.student_unet.block_ranges(), compress=cdt,
                                                twin=optimizer.grad_twin(cdt) if hasattr(optimizer, "grad_twin") else None)
            nan_any = dist_util.AnyRankFlag(torch.isnan(loss))
            self._student_backward(pred, target, weights, 1.0, buckets.ready if buckets.enabled else None)
            world = buckets.wait()
            value = float(loss.item())
            self._optimizer_tail(optimizer, lr_scheduler, 1.0 / world, not (skip_nan and nan_any.result()))
        return value

    @torch.no_grad()
    def inference(self, prompt, inference_scheduler, guidance_scale_input=3, guidance_scale_post=1, num_steps=20,
                  use_edm=False, num_samples=1, use_ema=True, query_teacher=False, noise=None, **kwargs):
        self.check_eval_mode()
        sch = inference_scheduler
        use_cf = guidance_scale_post > 1.
        embeds_cf, mask_cf, embeds, mask = self.get_prompt_embeds(prompt, True, num_samples)
        enc_stu, mask_stu = (embeds_cf, mask_cf) if use_cf else (embeds, mask)
        enc_tea, mask_tea = (embeds_cf, mask_cf) if self.use_teacher_cf_guidance else (embeds, mask)
        dev = embeds.device
        B = embeds.shape[0]
        sch.set_timesteps(num_steps, device=dev)
        tea_sch = deepcopy(sch) if query_teacher else None
        if noise is None:
            noise = randn_tensor((B, self.student_unet.config.in_channels, 256, 16), device=dev, dtype=torch.float32)
        z_stu = z_tea = noise * sch.init_noise_sigma
        unet = self.student_ema_unet if use_ema else self.student_unet
        for t in [float(v) for v in sch.timesteps]:
            z_in = torch.cat([z_stu] * 2) if use_cf else z_stu
            z_in = sch.scale_model_input(z_in, t)
            v = unet(z_in, t, guidance=guidance_scale_input, encoder_hidden_states=enc_stu,
                     encoder_attention_mask=mask_stu).sample
            if use_cf:
                u, c = v.chunk(2)
                v = u + guidance_scale_post * (c - u)
            z_stu = sch.step(v, int(t), z_stu).prev_sample
            if query_teacher:
                vt = self._query_teacher(sch.scale_model_input(z_tea, t), t, enc_tea, mask_tea, guidance_scale_input)
                z_tea = tea_sch.step(vt, int(t), z_tea).prev_sample
        return z_stu


def _teacher_loop_graphed(self, sch, z, enc, mask, guidance_scale):
    """The teacher loop of `inference` (audio_consistency_model.py:513-531) with its launch sequence captured
    once: timesteps and sigmas live in device tables indexed by a device-side counter, the Heun state
    (sample, derivative) in static buffers, so one graph = one full 2nd-order Heun step (scale, CFG teacher
    query, 1st-order half, scale, CFG teacher query, 2nd-order half) and the loop is N-1 replays plus the
    final 1st-order half.  No host sync, no per-iteration Python work beyond `replay()`."""
    dev = z.device
    B = z.shape[0]
    n = z[0].numel()
    ts_host, sig_host = sch._timesteps_host, sch._sigmas_host
    nt = len(ts_host)                                   # 2N - 1
    ts_dev = torch.tensor(ts_host, dtype=torch.float32, device=dev)
    sig_dev = torch.tensor(sig_host, dtype=torch.float32, device=dev)
    L_ = N.lib()
    w = torch.full((B,), float(guidance_scale), dtype=torch.float32, device=dev)
    x = z.clone().contiguous()
    idx = torch.zeros(1, dtype=torch.int64, device=dev)
    bufs = {k: torch.empty_like(x) for k in ("zin", "xhat", "deriv", "xnew")}
    pred = torch.empty_like(x)

    first = [True]

    def query(zin, t_b):
        # the text states are the same tensors for the whole loop: only the very first query projects their K / V
        out = self.teacher_unet(torch.cat([zin] * 2), torch.cat([t_b] * 2), enc, encoder_attention_mask=mask,
                                reuse_text=not first[0]).sample
        first[0] = False
        N.check(L_.ctta_cfg_combine(N.ptr(out[:B]), N.ptr(out[B:]), N.ptr(w), N.ptr(pred), B, n, N.stream_ptr()))
        return pred

    def half_first():
        s_a = sig_dev.index_select(0, idx).expand(B).contiguous()
        s_n = sig_dev.index_select(0, idx + 1).expand(B).contiguous()
        t_a = ts_dev.index_select(0, idx).expand(B).contiguous()
        N.check(L_.ctta_heun_scale_model_input(N.ptr(x), N.ptr(s_a), N.ptr(bufs["zin"]), B, n, N.stream_ptr()))
        v = query(bufs["zin"], t_a)
        N.check(L_.ctta_heun_step_first(N.ptr(v), N.ptr(x), N.ptr(s_a), N.ptr(s_n), N.ptr(bufs["xhat"]),
                                        N.ptr(bufs["deriv"]), B, n, N.stream_ptr()))
        return s_a, s_n

    def pair():
        s_a, s_n = half_first()
        t_b = ts_dev.index_select(0, idx + 1).expand(B).contiguous()
        N.check(L_.ctta_heun_scale_model_input(N.ptr(bufs["xhat"]), N.ptr(s_n), N.ptr(bufs["zin"]), B, n, N.stream_ptr()))
        v = query(bufs["zin"], t_b)
        N.check(L_.ctta_heun_step_second(N.ptr(v), N.ptr(bufs["xhat"]), N.ptr(x), N.ptr(bufs["deriv"]), N.ptr(s_a),
                                         N.ptr(s_n), N.ptr(bufs["xnew"]), B, n, N.stream_ptr()))
        x.copy_(bufs["xnew"])
        idx.add_(2)

    n_pairs = (nt - 1) // 2
    if n_pairs > 0:
        # one eager pair on a side stream first: handles, kernel attributes and the allocator pool must exist
        # before capture; then rewind the state and capture
        x0 = x.clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            pair()
        torch.cuda.current_stream(dev).wait_stream(side)
        x.copy_(x0)
        idx.zero_()
        graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: with a process group alive, RCCL's watchdog thread may touch the HIP runtime
        # during the capture; only this thread's calls belong to it
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            pair()
        x.copy_(x0)
        idx.zero_()
        for _ in range(n_pairs):
            graph.replay()
    half_first()                # last timestep: sigma_next = 0, a plain 1st-order (Euler) step
    return bufs["xhat"].clone()


AudioLCM._teacher_loop_graphed = _teacher_loop_graphed


class _DistillStepGraph:
    """The distillation micro-step of `AudioLCM.train_step` (tools/train_utils.py:150-183) with its launch sequence
    captured ONCE into a hipGraph: noising, the two CFG teacher queries + Heun, the target network, the student's training
    forward (on its side stream) and backward, and the loss -- about 5 600 launches -- replay as one submission.  AdamW,
    the LR schedule, zero_grad and the EMA stay eager behind the replay (3 launches; their scalars -- lr, step count --
    are kernel arguments and change every step).

    What makes the sequence replayable: every per-step quantity lives in a STATIC device tensor that the host refreshes
    before the replay -- the latents, the noise, the guidance scales, the two timestep vectors and the eight per-sample
    sigma vectors the Heun scheduler's methods look up (computed by the scheduler's own host logic, `_sigma_plan`, in the
    order `_forward_impl` calls them) -- and the networks whose weights change every step (student after AdamW, target after
    the EMA) have their bf16 re-pack captured at the head of their forward.
    Results are bit-identical to the eager `train_step` (same kernels, same arguments; asserted by
    tests/test_train_gpu.py and by bench.py before it times the replays).

    `segmented=True` is the form the DATA-PARALLEL step runs (tools/train_utils.py:152-183 under accelerate's DDP): the
    sequence is captured as several hipGraphs -- forward, loss, d loss / d pred and the blocks of the first all-reduce
    bucket, then one graph per further bucket of the block-wise backward (`ctta_unet_backward_next`: every call joins its
    weight-gradient side stream before it returns, so any run of calls is a closed sub-sequence; blocks are merged by
    `GradientBuckets`' own rule).  `step` replays them in order and hands every finished block to
    `dist_util.GradientBuckets` between two replays, so the bucketed asynchronous all-reduce (RCCL on
    its own stream) overlaps the blocks still to come exactly as in the eager `train_step`; the graphs share one memory
    pool and the engine's arena, so the kernels and their arguments are those of the monolithic capture."""

    def __init__(self, model, optimizer, z_shape, P, segmented=False, accumulation_steps=1, pipeline_teacher=False,
                 bucket_min_elems=16 << 20, main_eager=None):
        assert model.training and model.use_teacher_cf_guidance
        self.m, self.opt = model, optimizer
        self.segmented = bool(segmented)
        self.pipelined = bool(pipeline_teacher)
        # main_eager: only the teacher phase is a hipGraph (on its own stream); target network, student forward, loss and
        # backward run as eager launches -- the form the waveform-domain losses take (loss_type 'mel' / 'stft' / 'clap': torch
        # autograd differentiates the loss down to the latent, which a capture cannot hold)
        # (round 6: `main_eager=False` with a waveform-domain loss captures that part too -- the decode, the loss modules and
        # torch's backward through them are recorded like any other launch sequence; the ground-truth audio and the prompt's
        # extra tensors then live in static buffers of the input set.  Opt-in: the default stays eager.)
        self.main_eager = bool(model.loss is not None) if main_eager is None else bool(main_eager)
        assert not self.main_eager or self.pipelined, "main_eager without a pipelined teacher is the plain train_step"
        self.accum = max(1, int(accumulation_steps))
        self.bucket_min_elems = int(bucket_min_elems)     # = GradientBuckets' merge rule (dist_util)
        self.segments = []       # [(graph, (block ids it completes, ...))]
        self._micro = 0
        dev = model.device
        B = z_shape[0]
        self.B, self.dev = B, dev
        # Static inputs of one batch, carved out of ONE flat buffer (so that the pipelined step rotates a whole set with a
        # single device copy): the latents, the noise, guidance, the two timestep vectors, the eight sigma vectors, and the
        # two tensors the teacher phase hands to the rest of the step (student input, target-network input).
        self.cur = self._input_set(z_shape, P)
        self.nxt = self._input_set(z_shape, P) if self.pipelined else self.cur      # what the teacher graph reads / writes
        self.loss = None
        self.graph = None
        self.teacher_graph = None
        self._pinned = torch.zeros(11, B, dtype=torch.float32).pin_memory()
        self._primed = False
        if self.pipelined:
            self._tstream = torch.cuda.Stream(device=dev)
            self._ev_teacher = torch.cuda.Event()
            self._ev_h2d = torch.cuda.Event()

    _P_KEYS = ("embeds_cf", "mask_cf", "embeds", "mask")

    def _input_set(self, z_shape, P):
        B = z_shape[0]
        nz = int(np.prod(z_shape))
        flat = torch.zeros(4 * nz + 11 * B, dtype=torch.float32, device=self.dev)
        # the batch's text states ride with it (the teacher graph reads the CFG pair of the NEXT batch while the main
        # part still reads the current batch's): static copies per set; anything else in the prompt dict (CLAP caption
        # features) and the ground-truth waveforms of a waveform-domain loss are carried by reference
        S = {"flat": flat, "P": {k: P[k].detach().clone() for k in self._P_KEYS},
             "extra": {k: v for k, v in P.items() if k not in self._P_KEYS}, "gt_wav": None}
        for i, k in enumerate(("z0", "noise", "z_in", "tgt_in")):
            S[k] = flat[i * nz:(i + 1) * nz].view(z_shape)
        small = flat[4 * nz:].view(11, B)
        S["small"] = small
        S["t_np1"], S["t_n"], S["w"] = small[0], small[1], small[2]
        S["sig"] = [small[3 + i] for i in range(8)]
        return S

    # the names the tests and bench.py use for the CURRENT batch's static tensors
    z0 = property(lambda self: self.cur["z0"])
    noise = property(lambda self: self.cur["noise"])
    w = property(lambda self: self.cur["w"])
    t_np1 = property(lambda self: self.cur["t_np1"])
    t_n = property(lambda self: self.cur["t_n"])
    sig = property(lambda self: self.cur["sig"])
    P = property(lambda self: self.cur["P"])

    def _rotate(self):
        """next -> current: one device copy for the numeric inputs, four for the text states, references for the rest."""
        self.cur["flat"].copy_(self.nxt["flat"])
        for k in self._P_KEYS:
            self.cur["P"][k].copy_(self.nxt["P"][k])
        if self.main_eager or self.m.loss is None:
            self.cur["extra"], self.cur["gt_wav"] = self.nxt["extra"], self.nxt["gt_wav"]
            return
        # captured main part: the graph reads the CURRENT set's own buffers
        def own(dst, src):
            if torch.is_tensor(src):
                if torch.is_tensor(dst) and dst.shape == src.shape and dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
                    return dst
                return src.detach().clone()
            return src
        self.cur["extra"] = {k: own(self.cur["extra"].get(k), v) for k, v in self.nxt["extra"].items()}
        self.cur["gt_wav"] = own(self.cur.get("gt_wav"), self.nxt["gt_wav"])

    # -- host side: what the Heun scheduler would look up for this draw, in `_forward_impl`'s call order
    def _sigma_plan(self, inds):
        sch = self.m.noise_scheduler
        avail, sig = sch._timesteps_host, sch._sigmas_host
        t_np1, t_n = avail[inds], avail[inds + 2]

        def idx(t, first_order):
            saved = sch.dt
            sch.dt = None if first_order else (0, 0)
            try:
                return sch.index_for_timestep(t)
            finally:
                sch.dt = saved
        i1 = idx(t_np1, True)         # add_noise, scale_model_input, step (1st order) at t_{n+1}
        i2 = idx(t_n, False)          # scale_model_input and step (2nd order) at t_n
        i3 = idx(t_n, True)           # scale_model_input at t_n, back in first-order state
        plan = [sig[i1], sig[i1 + 1], sig[i2], sig[i2 - 1], sig[i2], sig[i3], sig[inds]]
        return t_np1, t_n, plan

    def _refresh(self, z_0, time_inds, gaussian_noise, guidance_scale, S=None, prompt=None, gt_wav=None):
        """Fills an input set (default: the one the teacher phase reads) on the CURRENT stream.  `prompt`: the batch's
        pre-computed text states (dict); None keeps the set's (a fixed prompt batch)."""
        S = self.nxt if S is None else S
        self._refresh_tensors(S, z_0, gaussian_noise, prompt, gt_wav)
        self._refresh_scalars(S, time_inds, guidance_scale)

    def _refresh_tensors(self, S, z_0, gaussian_noise, prompt, gt_wav):
        """The device-side half: copies out of the CALLER's tensors (latents, noise, text states), on the current stream --
        which must be the stream the caller produced them on (`feed` runs this before it hands over to the teacher stream,
        so the caching allocator never sees a caller tensor read on a stream it does not know about)."""
        if prompt is not None:
            for k in self._P_KEYS:
                S["P"][k].copy_(prompt[k])
            extra = {k: v for k, v in prompt.items() if k not in self._P_KEYS}
            if self.main_eager or self.m.loss is None:
                S["extra"] = extra
            else:                              # captured main part: static copies (the graph holds their addresses)
                for k, v in extra.items():
                    if torch.is_tensor(v):
                        if k not in S["extra"] or not torch.is_tensor(S["extra"][k]) or S["extra"][k].shape != v.shape or S["extra"][k] is v:
                            S["extra"][k] = v.detach().clone()
                        else:
                            S["extra"][k].copy_(v)
                    else:
                        S["extra"][k] = v
        if gt_wav is not None and not self.main_eager and self.m.loss is not None:
            if S.get("gt_wav") is None or S["gt_wav"].shape != gt_wav.shape or S["gt_wav"] is gt_wav:
                S["gt_wav"] = gt_wav.detach().clone()
            else:
                S["gt_wav"].copy_(gt_wav)
        else:
            S["gt_wav"] = gt_wav
        S["z0"].copy_(z_0)
        S["noise"].copy_(gaussian_noise if gaussian_noise is not None else torch.randn_like(z_0))

    def _refresh_scalars(self, S, time_inds, guidance_scale):
        """The host-side half: timesteps, sigma vectors and guidance scales, computed by the scheduler's own host logic and
        copied from the pinned staging buffer on the current stream."""
        m, B = self.m, self.B
        avail = m.noise_scheduler._timesteps_host
        order = 2
        if time_inds is not None:
            inds = time_inds.to("cpu", torch.int64)
        else:
            inds = torch.randint(0, (len(avail) - 1) // order, (B,)) * order
        if m.teacher_guidance_scale != -1:
            # a fixed teacher scale conditions student and target on that same w (`_forward_impl`: guidance_scale = None
            # -> w = float(teacher_guidance_scale)); `_query_teacher` uses the fixed scale by itself
            guidance_scale = torch.full((B,), float(m.teacher_guidance_scale))
        elif guidance_scale is None:
            guidance_scale = torch.rand(B) * m.max_rand_guidance_scale
        t_np1, t_n, plan = self._sigma_plan(inds.numpy())
        if self.pipelined:
            self._ev_h2d.synchronize()       # the previous batch's host -> device copies have left the pinned buffer
        host = self._pinned
        host[0].copy_(torch.from_numpy(np.asarray(t_np1, dtype=np.float32)))
        host[1].copy_(torch.from_numpy(np.asarray(t_n, dtype=np.float32)))
        host[2].copy_(guidance_scale.detach().to("cpu", torch.float32).reshape(-1).expand(B))
        for i, s_ in enumerate(plan):
            host[3 + i].copy_(torch.from_numpy(np.asarray(s_, dtype=np.float32).reshape(-1)))
        # ONE host -> device copy: the pinned block has the layout of the set's (11, B) scalar block (round 4 issued eleven
        # copies, ~65 us apart on the stream: 1.4 ms per unpipelined step by the idle-gap table of profiles/gaps_distill_r04.txt)
        S["small"].copy_(host, non_blocking=True)
        if self.pipelined:
            self._ev_h2d.record()

    # -- device side: `_forward_impl` (training branch) + `_student_backward`, on static tensors only
    def _body(self):
        out, pred, target, s_loss, gamma = self._forward_part()
        self.m._student_backward(pred, target, s_loss, gamma, 1.0 / self.accum, None)
        return out

    def _teacher_part(self, S):
        """Noising, the two CFG teacher queries and the Heun step between them (audio_consistency_model.py:268-311) on the
        input set S: writes the student's input S['z_in'] and the target network's input S['tgt_in'].  Nothing here
        depends on the student's weights: the pipelined step runs it for batch i + 1 beside the rest of batch i."""
        m, B = self.m, self.B
        sch = m.noise_scheduler
        L_ = N.lib()
        z0, noise = S["z0"], S["noise"]
        n = z0[0].numel()
        s_add, s_next1, s_scale2, s_prev2, s_cur2, s_scale3 = S["sig"][:6]
        embeds_cf, mask_cf = S["P"]["embeds_cf"], S["P"]["mask_cf"]

        def scale(x, sg, out=None):
            out = torch.empty_like(x) if out is None else out
            N.check(L_.ctta_heun_scale_model_input(N.ptr(x), N.ptr(sg), N.ptr(out), B, n, N.stream_ptr()))
            return out
        z_noisy = torch.empty_like(z0)
        N.check(L_.ctta_heun_add_noise(N.ptr(z0), N.ptr(noise), N.ptr(s_add), N.ptr(z_noisy), B, n, N.stream_ptr()))
        z_gauss = noise * float(sch.init_noise_sigma)
        t_max = float(sch._timesteps_host.max())
        z_np1 = torch.where((S["t_np1"] == t_max).reshape(-1, 1, 1, 1), z_gauss, z_noisy)
        z_np1_scaled = scale(z_np1, s_add, S["z_in"])
        if self._fork_student is not None:
            self._fork_student()      # unpipelined: the student's forward starts here, beside the teacher queries
        w_t = S["w"] if m.teacher_guidance_scale == -1 else None      # None: `_query_teacher` takes the model's fixed scale
        v1 = m._query_teacher(z_np1_scaled, S["t_np1"], embeds_cf, mask_cf, w_t, reuse_text=False)
        zhat = torch.empty_like(z0)
        deriv = torch.empty_like(z0)
        N.check(L_.ctta_heun_step_first(N.ptr(v1.contiguous()), N.ptr(z_np1), N.ptr(s_add), N.ptr(s_next1), N.ptr(zhat),
                                        N.ptr(deriv), B, n, N.stream_ptr()))
        v2 = m._query_teacher(scale(zhat, s_scale2), S["t_n"], embeds_cf, mask_cf, w_t, reuse_text=True)   # K / V of query 1
        zhat2 = torch.empty_like(z0)
        N.check(L_.ctta_heun_step_second(N.ptr(v2.contiguous()), N.ptr(zhat), N.ptr(z_np1), N.ptr(deriv), N.ptr(s_prev2),
                                         N.ptr(s_cur2), N.ptr(zhat2), B, n, N.stream_ptr()))
        scale(zhat2, s_scale3, S["tgt_in"])

    _fork_student = None

    def _main_part(self, S, teacher_first):
        """Target network, the student's training forward (on its side stream) and the loss, on the input set S whose
        teacher phase is done (`teacher_first=False`) or runs first in this same sequence (the unpipelined step)."""
        m, B, dev = self.m, self.B, self.dev
        L_ = N.lib()
        w = S["w"]
        embeds, mask = S["P"]["embeds"], S["P"]["mask"]
        two_stream = os.environ.get("CTTA_TWO_STREAM", "1") != "0"
        box = {}

        def fork():
            if not two_stream:
                return
            cur = torch.cuda.current_stream(dev)
            side = m._side_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                box["pred"] = m.student_unet.forward_train(S["z_in"], S["t_np1"], w, embeds, mask)
        if teacher_first:
            self._fork_student = fork
            try:
                self._teacher_part(S)
            finally:
                self._fork_student = None
        else:
            fork()
        target = m.student_target_unet(S["tgt_in"], S["t_n"], guidance=w, encoder_hidden_states=embeds,
                                       encoder_attention_mask=mask).sample
        target = torch.where((S["t_n"] == 0).reshape(-1, 1, 1, 1), S["z0"], target).contiguous()
        if "pred" in box:
            torch.cuda.current_stream(dev).wait_stream(m._side_stream(dev))
            pred = box["pred"]
        else:
            pred = m.student_unet.forward_train(S["z_in"], S["t_np1"], w, embeds, mask)
        gamma = m.snr_gamma or 0.0
        s_loss = S["sig"][6]
        if m.loss is not None:     # waveform-domain loss (eager only): keep the graph from pred to the loss, as `_forward_impl` does
            leaf = pred.detach().requires_grad_(True)
            prompt = dict(S["P"], **S["extra"])
            with torch.enable_grad():
                graph = m._perceptual_loss(leaf, target, S["gt_wav"], prompt, s_loss, gamma)
            return graph.detach().reshape(1), pred, (leaf, graph), s_loss, gamma
        inst = torch.empty(B, dtype=torch.float32, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        N.check(L_.ctta_snr_mse_loss(N.ptr(pred.contiguous()), N.ptr(target), N.ptr(s_loss), float(gamma), N.ptr(inst),
                                     N.ptr(out), B, pred[0].numel(), N.stream_ptr()))
        return out, pred, target, s_loss, gamma

    def _forward_part(self):
        return self._main_part(self.cur, teacher_first=not self.pipelined)

    def _capture_segments(self, warm):
        """forward + loss + the first bucket's blocks | one graph per further BUCKET of the gradient all-reduce, all in one
        memory pool.  Blocks are merged until they hold `bucket_min_elems` gradient elements -- GradientBuckets' own rule, so
        every replay ends exactly where a collective can start: 8 graphs instead of 12 for the light U-Net (each graph
        launch, and each graph's own branch streams, take hardware-queue slots the pipelined teacher graph competes for)."""
        m = self.m
        unet = m.student_unet
        pool = torch.cuda.graph_pool_handle()
        kw = dict(pool=pool, stream=warm, capture_error_mode="thread_local")
        ranges = unet.block_ranges()
        min_e = self.bucket_min_elems
        state = {"fin": False}

        def more_blocks(blocks, pending):
            while not state["fin"] and pending < min_e:
                blk, state["fin"] = unet.backward_next()
                blocks.append(blk)
                pending += (ranges[blk][1] - ranges[blk][0]) if blk in ranges else 0
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, **kw):
            out, pred, target, s_loss, gamma = self._forward_part()
            Bn, C, H, W = pred.shape
            d = torch.empty(Bn, H * W, 8, dtype=torch.bfloat16, device=pred.device)
            N.check(N.lib().ctta_snr_mse_grad(N.ptr(pred), N.ptr(target), N.ptr(s_loss), float(gamma), 1.0 / self.accum,
                                              Bn, C, H * W, 8, N.ptr(d), N.stream_ptr()))
            first = unet.backward_begin(d)
            blocks = [first]
            more_blocks(blocks, (ranges[first][1] - ranges[first][0]) if first in ranges else 0)
        self._keep = (pred, target, d)        # read by the later graphs: their pool blocks must stay allocated
        self.segments = [(self.graph, tuple(blocks))]
        while not state["fin"]:
            g, blocks = torch.cuda.CUDAGraph(), []
            with torch.cuda.graph(g, **kw):
                more_blocks(blocks, 0)
            self.segments.append((g, tuple(blocks)))
        return out

    def capture(self, z_0, time_inds=None, gaussian_noise=None, guidance_scale=None, gt_wav=None):
        """One eager pass on a side stream (handles, kernel attributes, allocator pool), then the capture.  Both passes
        ACCUMULATE into the gradient buffer like any backward; the caller zeroes it (train_step does after its update)."""
        m = self.m
        self._refresh(z_0, time_inds, gaussian_noise, guidance_scale, gt_wav=gt_wav)
        with torch.no_grad():
            warm = torch.cuda.Stream(device=self.dev)
            warm.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(warm):
                if self.pipelined:
                    self._teacher_part(self.nxt)
                    self._rotate()
                self._body()
            torch.cuda.current_stream(self.dev).wait_stream(warm)
            torch.cuda.synchronize(self.dev)
            for net in (m.student_unet, m.student_target_unet):   # their bf16 re-pack belongs to every replay
                net._h_version = None
            if self.pipelined:
                self.teacher_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.teacher_graph, stream=warm, capture_error_mode="thread_local"):
                    self._teacher_part(self.nxt)
            if self.main_eager:
                pass                                  # the main part stays eager launches (see __init__)
            elif self.segmented:
                self.loss = self._capture_segments(warm)
            else:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=warm, capture_error_mode="thread_local"):
                    self.loss = self._body()
            if self.pipelined:
                self._place_teacher_stream()
        self.opt.zero_grad()
        return self

    def _place_teacher_stream(self, candidates=10):
        """Which stream the teacher graph is replayed on decides whether it overlaps anything: HIP maps its streams (torch's
        pool streams, the hipGraphs' own branch streams, the handles' side streams) onto 4 hardware queues, and two streams
        on one queue run back to back.  The mapping is not queryable, so it is MEASURED once per capture: (teacher graph on
        the candidate) beside (main graph on the current stream), timed for a handful of pool streams; the fastest stays
        (measured on MI355X: 86 ms per step on a free queue against 99 ms on the main graph's own queue)."""
        dev = self.dev
        cur = torch.cuda.current_stream(dev)
        self.placement_ms = []
        best = (None, float("inf"))
        # high-priority pool streams sit on hardware queues of their own (no normal-priority stream -- the main graph's launch
        # stream, its branch streams -- can share them); the teacher has a whole step of slack, so its priority only matters
        # for the placement
        cands = [self._tstream] + [torch.cuda.Stream(device=dev, priority=-1) for _ in range(2)] + \
                [torch.cuda.Stream(device=dev) for _ in range(max(0, candidates - 3))]
        for s_ in cands:
            ms = float("inf")
            for _ in range(2):          # the second pass is the measurement (the first pays first-use costs of the stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(dev)
                e0.record(cur)
                s_.wait_stream(cur)
                with torch.cuda.stream(s_):
                    self.teacher_graph.replay()
                self.replay()
                cur.wait_stream(s_)
                e1.record(cur)
                torch.cuda.synchronize(dev)
                ms = e0.elapsed_time(e1)
            self.placement_ms.append(round(ms, 2))
            if ms < best[1]:
                best = (s_, ms)
        self._tstream = best[0]

    def replay(self, on_block_done=None):
        """The device work of one micro-step on the CURRENT input set.  Segmented: `on_block_done(block id)` runs between
        two replays, when that block's gradients are final on the stream (what `ctta_unet_backward_next` reports to
        `train_step`).  Pipelined captures: the teacher phase of the current set is NOT part of this (see `feed`)."""
        if self.main_eager:
            with torch.no_grad():
                self.loss, pred, target, s_loss, gamma = self._forward_part()
                self.m._student_backward(pred, target, s_loss, gamma, 1.0 / self.accum, on_block_done)
            return
        if not self.segmented:
            self.graph.replay()
            return
        for g, blks in self.segments:
            g.replay()
            if on_block_done is not None:
                for blk in blks:
                    on_block_done(blk)

    def feed(self, z_0, time_inds=None, gaussian_noise=None, guidance_scale=None, prompt=None, gt_wav=None):
        """Pipelined captures: hands the NEXT batch (latents + draws) to the teacher stream -- inputs refreshed and the
        teacher graph replayed there, beside whatever the main stream does -- after moving the batch fed before into the
        current set.  `step` calls it; a training loop may call it directly to prime the pipeline with its first batch."""
        assert self.pipelined
        cur = torch.cuda.current_stream(self.dev)
        if self._primed:
            cur.wait_event(self._ev_teacher)                 # the teacher phase of the batch fed last time is complete
            self._rotate()                                   # ... and that batch becomes the current one
        # the caller's tensors are read HERE, on the caller's stream (read from the teacher stream without a
        # record_stream, the caching allocator could recycle them under the copies); only the pinned-buffer host -> device
        # copies and the graph replay run on the teacher stream
        self._refresh_tensors(self.nxt, z_0, gaussian_noise, prompt, gt_wav)
        self._tstream.wait_stream(cur)
        with torch.cuda.stream(self._tstream):
            self._refresh_scalars(self.nxt, time_inds, guidance_scale)
            self.teacher_graph.replay()
            self._ev_teacher.record(self._tstream)
        was, self._primed = self._primed, True
        self._fed_untrained = True
        return was

    def drain(self, lr_scheduler=None, skip_nan=True):
        """Pipelined captures: one more optimizer step on the batch fed LAST (whose teacher phase is done or in flight)
        WITHOUT feeding a new one -- the end of an epoch.  A loop `for z in batches: g.step(z)` trains on batches
        [0, 0, 1, ..., N-2] (the first call primes the pipeline with its batch and trains on it; every later call trains on
        the batch fed one call earlier); `drain()` then trains on batch N-1, so that every batch is trained on (the
        reference trains each batch exactly once, tools/train_utils.py:150-183; to make batch 0 count once too, prime with
        `feed(batch 0)` instead of `step(batch 0)` -- see INTEGRATION.md).  Returns the loss, or None when nothing is
        waiting."""
        assert self.pipelined
        if not self._primed or not getattr(self, "_fed_untrained", False):
            return None
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(self._ev_teacher)
        self._rotate()
        self._primed = self._fed_untrained = False      # the next feed() starts a fresh pipeline (no teacher phase in flight)
        return self._train_current(lr_scheduler, skip_nan)

    def step(self, z_0, lr_scheduler=None, time_inds=None, gaussian_noise=None, guidance_scale=None, skip_nan=True,
             prompt=None, gt_wav=None):
        """`AudioLCM.train_step`: refresh the static inputs, replay, then -- on every `accumulation_steps`-th call -- the
        eager tail (gradient all-reduce joined, AdamW, LR schedule, zero_grad, EMA).  With a process group the capture
        must be `segmented` so that the all-reduce of a finished block is issued before the next block's replay.
        Returns the (unscaled) loss as a Python float.

        Pipelined captures (`pipeline_teacher=True`): `z_0` and the draws are those of the NEXT batch -- their teacher
        phase is queued on the teacher stream and overlaps this call's student / target / backward work, which trains on
        the batch fed by the PREVIOUS call (the first call primes the pipeline with its batch and trains on it too)."""
        m = self.m
        draw = dict(time_inds=time_inds, gaussian_noise=gaussian_noise, guidance_scale=guidance_scale, prompt=prompt, gt_wav=gt_wav)
        if self.pipelined:
            if not self.feed(z_0, **draw):
                self.feed(z_0, **draw)      # first call: the same batch is also the next one
        else:
            self._refresh(z_0, time_inds, gaussian_noise, guidance_scale, prompt=prompt, gt_wav=gt_wav)
        return self._train_current(lr_scheduler, skip_nan)

    def _train_current(self, lr_scheduler=None, skip_nan=True):
        """Replay on the CURRENT input set + (at an accumulation boundary) the eager tail."""
        m = self.m
        self._micro += 1
        if self._micro % self.accum != 0:      # DDP's no_sync: gradients only accumulate locally
            self.replay()
            return float(self.loss.item())
        cdt = getattr(m, "allreduce_dtype", None)
        buckets = dist_util.GradientBuckets(self.opt.grad, m.student_unet.block_ranges(), min_elems=self.bucket_min_elems,
                                            compress=cdt, twin=self.opt.grad_twin(cdt) if hasattr(self.opt, "grad_twin") else None)
        if buckets.enabled and not self.segmented and not self.main_eager:
            raise N.CttaError("a monolithic step graph cannot interleave the gradient all-reduce with the backward pass: "
                              "capture with segmented=True under a process group")
        if self.main_eager:
            self.replay(buckets.ready if buckets.enabled else None)
            nan_any = dist_util.AnyRankFlag(torch.isnan(self.loss))
        elif not buckets.enabled:
            self.replay()
            nan_any = dist_util.AnyRankFlag(torch.isnan(self.loss))
        else:
            g0, first = self.segments[0]
            g0.replay()
            nan_any = dist_util.AnyRankFlag(torch.isnan(self.loss))     # all ranks skip together (or none)
            for blk in first:
                buckets.ready(blk)
            for g, blks in self.segments[1:]:
                g.replay()
                for blk in blks:
                    buckets.ready(blk)
        world = buckets.wait()
        value = float(self.loss.item())
        m._optimizer_tail(self.opt, lr_scheduler, 1.0 / world, not (skip_nan and nan_any.result()))
        return value


def _capture_train_graph(self, optimizer, z_0, prompt, segmented=None, accumulation_steps=1, pipeline_teacher=False,
                         bucket_min_elems=16 << 20, main_eager=None, **draws):
    """hipGraph-captured distillation step (loss_type='mse'): returns a `_DistillStepGraph` whose
    `.step(z_0, lr_scheduler, ...)` replaces `train_step(z_0, prompt, optimizer, lr_scheduler, ...)` for fixed shapes and a
    prompt shape (`prompt` must be the dict of pre-computed text states; `step(..., prompt=next_states)` / `feed(..., prompt=)`
    hands a batch's own text states over with it -- they are double-buffered like the latents).  With a waveform-domain loss
    (loss_type 'mel' / 'stft' / 'clap') pass pipeline_teacher=True: the teacher phase is a hipGraph on its own stream and the
    rest of the step stays eager launches (`main_eager`, the default) or -- `main_eager=False`, round 6 -- becomes a second
    hipGraph that holds the decode, the loss modules and torch's backward through them as well (the ground-truth audio and the
    prompt's extra tensors are then copied into static buffers of the input set); `step(..., gt_wav=)` carries the audio.
    `segmented` (default: whenever a process group with more than one rank exists) captures the backward block by block
    so that the data-parallel gradient all-reduce overlaps it as in the eager `train_step`.
    `pipeline_teacher=True`: the frozen teacher's two CFG queries + Heun step (a third of the step's device time, and
    independent of the student's weights) are captured as their own graph and run for batch i + 1 on a second stream
    beside the student / target / backward work of batch i -- `step(z_next)` then trains on the batch fed one call
    earlier (software pipelining of the training loop; same arithmetic per batch)."""
    if segmented is None:
        segmented = (dist_util.dist.is_initialized() and dist_util.dist.get_world_size() > 1) or \
            os.environ.get("CTTA_FORCE_COLLECTIVES", "0") == "1"
    if not isinstance(prompt, dict):
        raise N.CttaError("capture_train_graph needs the pre-computed text states (dict), not prompt strings")
    return _DistillStepGraph(self, optimizer, tuple(z_0.shape), prompt, segmented=segmented,
                             accumulation_steps=accumulation_steps, pipeline_teacher=pipeline_teacher,
                             bucket_min_elems=bucket_min_elems, main_eager=main_eager).capture(z_0, **draws)


AudioLCM.capture_train_graph = _capture_train_graph


class ConsistencyTTA(nn.Module):
    """easy_inference/consistencytta.py: prompts -> int16 waveforms in one module."""

    def __init__(self, unet_config=None, vae=None, text_encoder=None, tokenizer=None,
                 text_encoder_name="google/flan-t5-large"):
        super().__init__()
        from . import spec
        self.unet = UNet2DConditionGuidedModel.from_config(unet_config or spec.LIGHT_UNET_CONFIG, subfolder="unet")
        self.vae = vae if vae is not None else AutoencoderKL(embed_dim=8, scale_factor=1.0)
        self.text_encoder, self.tokenizer, self.text_encoder_name = text_encoder, tokenizer, text_encoder_name
        self.scheduler = HeunDiscreteScheduler.from_pretrained("stabilityai/stable-diffusion-2-1", subfolder="scheduler")
        self.unet.eval().requires_grad_(False)
        self.vae.eval().requires_grad_(False)

    @classmethod
    def from_checkpoint_dir(cls, ckpt_dir="consistencytta_clapft_ckpt", unet_config=None, text_encoder=None,
                            tokenizer=None, device=None):
        """The reference's released layout (easy_inference/consistencytta.py:22-42): `unet_state_dict.pt` (the guided
        U-Net's state dict) and `vae_state_dict.pt` = {"state_dict": AutoencoderKL incl. vocoder, "scale_factor"}."""
        import os
        unet_sd = torch.load(os.path.join(ckpt_dir, "unet_state_dict.pt"), map_location="cpu")
        raw = torch.load(os.path.join(ckpt_dir, "vae_state_dict.pt"), map_location="cpu")
        vae = AutoencoderKL(embed_dim=8, scale_factor=float(raw["scale_factor"]))
        vae.load_state_dict(raw["state_dict"])
        pipe = cls(unet_config=unet_config, vae=vae, text_encoder=text_encoder, tokenizer=tokenizer)
        pipe.unet.load_state_dict(unet_sd)
        if device is not None:
            pipe.to(device)
        return pipe.eval().requires_grad_(False)

    def check_eval_mode(self):
        for model, name in ((self.vae, "vae"), (self.unet, "unet")):
            assert model.training is False, f"The {name} is not in eval mode."
            for p in model.parameters():
                assert p.requires_grad is False, f"The {name} is not frozen."

    @torch.no_grad()
    def generate_latent(self, encoder_states, encoder_mask, noise, cfg_scale_input=3., cfg_scale_post=1.,
                        num_steps=1, uncond_states=None, uncond_mask=None):
        """consistencytta.py:152-197 from pre-computed text states (T5 excluded)."""
        sch = self.scheduler
        use_cf = cfg_scale_post > 1.
        if use_cf:
            encoder_states = torch.cat([uncond_states, encoder_states])
            encoder_mask = torch.cat([uncond_mask, encoder_mask])
        sch.set_timesteps(18, device=noise.device)
        z_N = noise * sch.init_noise_sigma

        def calc(z_n, t):
            z_in = torch.cat([z_n] * 2) if use_cf else z_n
            z_in = sch.scale_model_input(z_in, t)
            zh = self.unet(z_in, t, guidance=cfg_scale_input, encoder_hidden_states=encoder_states,
                           encoder_attention_mask=encoder_mask).sample
            if use_cf:
                u, c = zh.chunk(2)
                zh = (1 - cfg_scale_post) * u + cfg_scale_post * c
            return zh

        zhat_0 = calc(z_N, float(sch._timesteps_host[0]))
        sch.set_timesteps(num_steps, device=noise.device)
        for t in sch._timesteps_host[1::2]:
            zhat_n = sch.add_noise(zhat_0, torch.randn_like(zhat_0), float(t))
            zhat_0 = calc(zhat_n, float(t))
        return zhat_0

    @torch.no_grad()
    def capture_graph(self, batch, text_len, cfg_scale_input=3., cross_attention_dim=1024, latent=(8, 256, 16)):
        """Captures the whole single-step pipeline (U-Net query -> VAE decoder -> HiFi-GAN -> int16) for a fixed
        (batch, text_len) into ONE hipGraph and returns a callable `(encoder_states, encoder_mask, noise) -> pcm` that
        copies its arguments into the graph's static inputs and replays it.  The ~1500 kernel launches of a clip cost
        more host time than GPU time at batch 1 (demo.py / easy_inference latency); a replay is a single launch.
        Results are identical to `forward_from_embeds(..., cfg_scale_post=1, num_steps=1)`."""
        self.check_eval_mode()
        dev = self.unet.device
        st = {"enc": torch.zeros(batch, text_len, cross_attention_dim, device=dev),
              "mask": torch.ones(batch, text_len, dtype=torch.bool, device=dev),
              "noise": torch.zeros((batch,) + tuple(latent), device=dev)}
        scratch = torch.empty(4, dtype=torch.float32, device=dev)

        def run():
            lat = self.generate_latent(st["enc"], st["mask"], st["noise"], cfg_scale_input, 1.0, 1)
            mel = self.vae.decode_first_stage(lat)
            wav = self.vae.vocode(mel)
            pcm = torch.empty(wav.shape, dtype=torch.int16, device=dev)
            N.check(N.lib().ctta_wav_finalize(N.ptr(wav), wav.numel(), N.ptr(scratch), None, N.ptr(pcm), N.stream_ptr()))
            return lat, mel, pcm

        side = torch.cuda.Stream(device=dev)        # eager warm-up: engine handles, kernel attributes, allocator pool
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            out = run()

        def replay(encoder_states, encoder_mask, noise):
            st["enc"].copy_(encoder_states)
            st["mask"].copy_(encoder_mask)
            st["noise"].copy_(noise)
            graph.replay()
            return out[2]

        replay.graph, replay.outputs = graph, out
        return replay

    def capture_pipeline(self, batch, text_len, cfg_scale_input=3., cross_attention_dim=1024, latent=(8, 256, 16),
                         candidates=8):
        """The same single-step pipeline as a 3-deep SOFTWARE PIPELINE over batches for a throughput-bound server: three
        hipGraphs -- U-Net query, VAE decoder, HiFi-GAN + int16 -- replayed per call on three streams for three DIFFERENT
        batches (the batch of this call, of the previous call, of the call before).  The stages of one batch stay strictly
        ordered; stages of different batches share nothing, and each one's thin launches (the U-Net's deep levels, the
        vocoder's C <= 64 stages) fill the CUs the others leave idle.  Returns `f(encoder_states, encoder_mask, noise) ->
        int16 waveforms of the batch fed TWO calls earlier` (the buffer is overwritten by the next call; the first two calls
        return warm-up garbage -- feed two extra batches, e.g. the last one again, to drain).  Results are bit-identical to
        `capture_graph` / the eager path.  Which streams the U-Net and decoder graphs run on is measured at capture: HIP maps
        streams onto 4 hardware queues and two graphs on one queue do not overlap (see `_DistillStepGraph._place_teacher_stream`)."""
        self.check_eval_mode()
        dev = self.unet.device
        st = {"enc": torch.zeros(batch, text_len, cross_attention_dim, device=dev),
              "mask": torch.ones(batch, text_len, dtype=torch.bool, device=dev),
              "noise": torch.zeros((batch,) + tuple(latent), device=dev)}
        scratch = torch.empty(4, dtype=torch.float32, device=dev)

        def graphed(fn):
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                fn()                                  # eager warm-up: engine handles, kernel attributes, allocator pool
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side, capture_error_mode="thread_local"):
                out = fn()
            return gr, out

        g_u, lat = graphed(lambda: self.generate_latent(st["enc"], st["mask"], st["noise"], cfg_scale_input, 1.0, 1))
        lat_in = lat.clone()
        g_v, mel = graphed(lambda: self.vae.decode_first_stage(lat_in))
        mel_in = mel.clone()

        def voc():
            wav = self.vae.vocode(mel_in)
            pcm = torch.empty(wav.shape, dtype=torch.int16, device=dev)
            N.check(N.lib().ctta_wav_finalize(N.ptr(wav), wav.numel(), N.ptr(scratch), None, N.ptr(pcm), N.stream_ptr()))
            return pcm
        g_h, pcm = graphed(voc)
        streams = {"u": torch.cuda.Stream(device=dev), "v": torch.cuda.Stream(device=dev)}

        def iteration():
            cur = torch.cuda.current_stream(dev)
            lat_in.copy_(lat)                          # hand-over at the call boundary: batch i-1 to the decoder, i-2 to the vocoder
            mel_in.copy_(mel)
            for k, g_ in (("v", g_v), ("u", g_u)):
                streams[k].wait_stream(cur)
                with torch.cuda.stream(streams[k]):
                    g_.replay()
            g_h.replay()                               # the largest stage on the caller's stream
            for k in ("v", "u"):
                cur.wait_stream(streams[k])

        def timed():
            ms = 0.0
            for _ in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(dev)
                e0.record()
                iteration()
                e1.record()
                torch.cuda.synchronize(dev)
                ms = e0.elapsed_time(e1)
            return ms
        placement = {}
        for k in ("v", "u"):                           # decoder first, then the U-Net beside both
            best = (streams[k], float("inf"))
            tried = []
            for s_ in [streams[k]] + [torch.cuda.Stream(device=dev) for _ in range(max(0, candidates - 1))]:
                streams[k] = s_
                ms = timed()
                tried.append(round(ms, 2))
                if ms < best[1]:
                    best = (s_, ms)
            streams[k] = best[0]
            placement[k] = tried

        def replay(encoder_states, encoder_mask, noise):
            st["enc"].copy_(encoder_states)
            st["mask"].copy_(encoder_mask)
            st["noise"].copy_(noise)
            iteration()
            return pcm

        replay.depth, replay.placement_ms, replay.graphs = 2, placement, (g_u, g_v, g_h)
        return replay

    # ---- text side, easy_inference/consistencytta.py:82-132 (FLAN-T5 on the HIP engine, tokenizer on the host)
    @property
    def device(self):
        return self.unet.device

    _require_text_encoder = AudioDistilledModel._require_text_encoder
    encode_text = AudioDistilledModel.encode_text
    encode_text_classifier_free = AudioDistilledModel.encode_text_classifier_free

    @torch.no_grad()
    def forward(self, prompt, cfg_scale_input=3., cfg_scale_post=1., num_steps=1, num_samples=1, sr=16000):
        """easy_inference/consistencytta.py:135-200: prompts -> int16 waveforms (numpy), truncated to 9.5 s."""
        self.check_eval_mode()
        embeds_cf, mask_cf, embeds, mask = self.encode_text_classifier_free(prompt, num_samples)
        B = embeds.shape[0]
        noise = randn_tensor((B, self.unet.config.in_channels, 256, 16), device=embeds.device, dtype=torch.float32)
        kw = {}
        if cfg_scale_post > 1.:
            kw = dict(uncond_states=embeds_cf[:B], uncond_mask=mask_cf[:B])
        return self.forward_from_embeds(embeds, mask, noise, cfg_scale_input, cfg_scale_post, num_steps, sr, **kw)

    @torch.no_grad()
    def forward_from_embeds(self, encoder_states, encoder_mask, noise, cfg_scale_input=3., cfg_scale_post=1.,
                            num_steps=1, sr=16000, **kw):
        self.check_eval_mode()
        lat = self.generate_latent(encoder_states, encoder_mask, noise, cfg_scale_input, cfg_scale_post, num_steps, **kw)
        mel = self.vae.decode_first_stage(lat.float())
        return self.vae.decode_to_waveform(mel)[:, :int(sr * 9.5)]
