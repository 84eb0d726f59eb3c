"""On-disk run formats of the reference's training / evaluation scripts.

* `summary.jsonl` in the run directory: the first line is `json.dumps(vars(args))` of the training
  command (train.py:304-305), every later record one JSON object per epoch (tools/train_utils.py:240-241);
  records are separated by a blank line.  `inference.py:114` / `demo.py:59` rebuild the model from the FIRST
  line (`dotdict(json.loads(open(path).readlines()[0]))`).
* Checkpoint directories `best/`, `epoch_N/`, `step_N/` written by `accelerator.save_state` (train.py:498-505,
  tools/train_utils.py:197) with accelerate 0.19's naming: the i-th prepared model's state dict is
  `pytorch_model.bin` (i = 0) or `pytorch_model_{i}.bin`; the reference prepares (vae, stft, model) in that
  order (train.py:380), so the distilled model is `pytorch_model_2.bin` -- the file `inference.py --model` and
  `demo.py` load with `torch.load(...)` + `model.load_pretrained(...)` (inference.py:152-153); `optimizer.bin`,
  `scheduler.bin` and `random_states_0.pkl` sit beside it and are read back by `accelerator.load_state`
  (train.py:424-427).  `optimizer.bin` holds `torch.optim.AdamW.state_dict()` over `student_unet.parameters()` and
  `scheduler.bin` `LambdaLR.state_dict()` (optim.FusedAdamW / WarmupSchedule emit and accept exactly those layouts:
  tests/test_host_cpu.py::test_optimizer_and_scheduler_files_are_torch_layout loads files written by the torch
  classes and the other way round), so a reference run resumes here and vice versa.

Host-side file plumbing only: tensors are moved to the CPU for writing and land on each module's own device when read.
"""
import json
import os
import random

import numpy as np
import torch

MODEL_NAME = "pytorch_model"
OPTIMIZER_NAME = "optimizer"
SCHEDULER_NAME = "scheduler"
RNG_STATE_NAME = "random_states"


class dotdict(dict):
    """inference.py:18-22: attribute access to the training arguments; a missing key reads as None."""
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__


# ------------------------------------------------------------------------------------------------ summary.jsonl
def write_args_summary(output_dir, args):
    """train.py:302-305: creates `<output_dir>/outputs`, appends the argument record."""
    os.makedirs(os.path.join(output_dir, "outputs"), exist_ok=True)
    rec = dict(vars(args)) if not isinstance(args, dict) else dict(args)
    with open(os.path.join(output_dir, "summary.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n\n")


def append_summary(output_dir, result):
    """tools/train_utils.py:240-241: one JSON object per evaluation, blank-line separated."""
    with open(os.path.join(output_dir, "summary.jsonl"), "a") as f:
        f.write(json.dumps(result) + "\n\n")


def read_original_args(path):
    """inference.py:114-116: the first line of summary.jsonl as a dotdict, `hf_model` defaulted to None."""
    train_args = dotdict(json.loads(open(path).readlines()[0]))
    if "hf_model" not in train_args:
        train_args["hf_model"] = None
    return train_args


def read_summary(path):
    """Every record of a summary.jsonl (arguments first, then the per-epoch results)."""
    return [json.loads(line) for line in open(path).read().split("\n") if line.strip()]


# ------------------------------------------------------------------------------------------------ save / load state
def _model_file(i):
    return "%s.bin" % MODEL_NAME if i == 0 else "%s_%d.bin" % (MODEL_NAME, i)


def _cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().to("cpu").clone()
    if isinstance(obj, dict):
        return type(obj)((k, _cpu(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_cpu(v) for v in obj)
    return obj


def save_state(output_dir, models, optimizer=None, lr_scheduler=None, process_index=0):
    """`accelerator.save_state(output_dir)` for the objects the reference prepares: `models` in preparation order
    ((vae, stft, model) in train.py:380), then the optimizer, the LR scheduler and the RNG states."""
    os.makedirs(output_dir, exist_ok=True)
    for i, m in enumerate(models):
        torch.save(_cpu(m.state_dict()), os.path.join(output_dir, _model_file(i)))
    if optimizer is not None:
        torch.save(_cpu(optimizer.state_dict()), os.path.join(output_dir, OPTIMIZER_NAME + ".bin"))
    if lr_scheduler is not None:
        torch.save(lr_scheduler.state_dict(), os.path.join(output_dir, SCHEDULER_NAME + ".bin"))
    states = {"random_state": random.getstate(), "numpy_random_seed": np.random.get_state(),
              "torch_manual_seed": torch.get_rng_state()}
    if torch.cuda.is_available():
        states["torch_cuda_manual_seed"] = torch.cuda.get_rng_state_all()
    torch.save(states, os.path.join(output_dir, "%s_%d.pkl" % (RNG_STATE_NAME, process_index)))
    return output_dir


def load_state(input_dir, models, optimizer=None, lr_scheduler=None, process_index=0, map_location="cpu",
               restore_rng=True):
    """`accelerator.load_state(input_dir, map_location='cpu')` (train.py:424-427).  Module mirrors take the state
    dicts through their own `load_state_dict` (packed engine operands re-sync on the next call)."""
    for i, m in enumerate(models):
        sd = torch.load(os.path.join(input_dir, _model_file(i)), map_location=map_location)
        m.load_state_dict(sd)
    if optimizer is not None:
        optimizer.load_state_dict(torch.load(os.path.join(input_dir, OPTIMIZER_NAME + ".bin"), map_location=map_location))
    if lr_scheduler is not None:
        lr_scheduler.load_state_dict(torch.load(os.path.join(input_dir, SCHEDULER_NAME + ".bin")))
    path = os.path.join(input_dir, "%s_%d.pkl" % (RNG_STATE_NAME, process_index))
    if restore_rng and os.path.exists(path):
        states = torch.load(path, weights_only=False)
        random.setstate(states["random_state"])
        np.random.set_state(states["numpy_random_seed"])
        torch.set_rng_state(states["torch_manual_seed"])
        if torch.cuda.is_available() and "torch_cuda_manual_seed" in states:
            try:
                torch.cuda.set_rng_state_all(states["torch_cuda_manual_seed"])
            except Exception:
                pass    # saved on a box with a different GPU count
    return input_dir


# ------------------------------------------------------------------------------------------------ inference.py:112-158
def build_model_from_run(model_path, original_args_path, vae=None, stage=2, device=None, **overrides):
    """What inference.py / demo.py do between `parse_args` and the generation loop: read the training arguments
    from summary.jsonl, construct AudioGDM (stage 1) or AudioLCM (stage 2) from them, load `pytorch_model_2.bin`
    through `load_pretrained` (legacy key names included) and switch to eval mode.  `overrides` (e.g. `unet_config=`,
    `text_encoder=`, `tokenizer=`) are passed to the constructor: offline boxes cannot fetch FLAN-T5."""
    from .models import AudioGDM, AudioLCM
    train_args = read_original_args(original_args_path)
    assert stage == train_args.stage, "Stage mismatch between training and eval."
    assert not train_args.finetune_vae, "AudioLCM_FTVAE (VAE fine-tuning) is outside the built path"
    cls = AudioGDM if stage == 1 else AudioLCM
    kw = dict(text_encoder_name=train_args.text_encoder_name, scheduler_name=train_args.scheduler_name,
              unet_model_name=train_args.unet_model_name, unet_model_config_path=train_args.unet_model_config,
              snr_gamma=train_args.snr_gamma, freeze_text_encoder=train_args.freeze_text_encoder,
              uncondition=train_args.uncondition, use_edm=train_args.use_edm, use_karras=train_args.use_karras,
              use_lora=train_args.use_lora, target_ema_decay=train_args.target_ema_decay,
              ema_decay=train_args.ema_decay, num_diffusion_steps=train_args.num_diffusion_steps,
              teacher_guidance_scale=train_args.teacher_guidance_scale, loss_type=train_args.loss_type, vae=vae)
    kw.update(overrides)
    model = cls(**{k: v for k, v in kw.items() if v is not None or k in ("unet_model_name", "snr_gamma", "vae")})
    if device is not None:
        model.to(device)
    model.load_pretrained(torch.load(model_path, map_location="cpu"))
    model.eval()
    return model, train_args
