"""Optimizer of the distillation step: torch.optim.AdamW semantics (tools/train_utils.py:59-63) as ONE
fused launch over the student's flat parameter / gradient / moment buffers, plus the constant-with-
warmup / linear schedules the reference builds with transformers.get_scheduler (train_utils.py:77-81).
"""
import math

import torch

from . import _native as N


class FusedAdamW:
    """AdamW over a `_ParamTree` whose parameters were re-homed into one flat fp32 buffer.

    The trainable prefix of the flat buffers is updated by `ctta_adamw_step` (decoupled weight decay,
    bias correction by step count, no amsgrad); frozen parameters (guidance_proj.weight) are never
    touched, exactly as torch skips parameters without a gradient."""

    def __init__(self, module, lr=3e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.module = module
        self.flat = module.flatten_parameters_()
        self.grad = module.flat_grad_()
        self.n = module.n_trainable()
        self.exp_avg = torch.zeros(self.n, dtype=torch.float32, device=self.flat.device)
        self.exp_avg_sq = torch.zeros(self.n, dtype=torch.float32, device=self.flat.device)
        self.param_groups = [dict(lr=float(lr), betas=tuple(betas), eps=float(eps), weight_decay=float(weight_decay))]
        self.step_count = 0
        self._grad_twin = None

    def grad_twin(self, dtype):
        """The persistent low-precision twin of the flat gradient buffer that a compressed gradient all-reduce stages its
        buckets in (`dist_util.GradientBuckets(..., compress=dtype, twin=...)`): owned by the optimizer that owns the
        gradient buffer -- allocated on first use, replaced when the dtype changes, freed with the optimizer."""
        if dtype is None:
            return None
        t = self._grad_twin
        if t is None or t.dtype != dtype or t.numel() != self.grad.numel() or t.device != self.grad.device:
            t = self._grad_twin = torch.empty(self.grad.numel(), dtype=dtype, device=self.grad.device)
        return t

    def zero_grad(self, set_to_none=False):
        self.grad.zero_()

    def step(self, grad_scale=1.0):
        """One update; `grad_scale` multiplies the gradient inside the kernel (1/world_size after a
        SUM all-reduce, 1/accumulation_steps, ...)."""
        g = self.param_groups[0]
        if not self.module.flat_is_current() or self.module._flat is not self.flat:
            raise N.CttaError("FusedAdamW: the module's parameters no longer alias the flat buffer this optimizer "
                              "was built on (model.to()/.float() after prepare_training?) -- build a new optimizer")
        if not self.module.grads_alias_flat_sampled():
            self.module.realias_grads_()
            raise N.CttaError("FusedAdamW: parameter gradients were not views of the flat gradient buffer (zero_grad("
                              "set_to_none=True) + a foreign backward?); they have been re-aliased -- redo the backward")
        self.step_count += 1
        with torch.cuda.device(self.flat.device):
            N.check(N.lib().ctta_adamw_step(N.ptr(self.flat), N.ptr(self.grad), N.ptr(self.exp_avg),
                                            N.ptr(self.exp_avg_sq), self.n, g["lr"], g["betas"][0], g["betas"][1],
                                            g["eps"], g["weight_decay"], self.step_count, float(grad_scale),
                                            N.stream_ptr()))
        self.module.mark_weights_changed()

    def can_fuse_tail(self, shadow_flats):
        """Whether `step_zero_ema` may replace step() + zero_grad() + the EMA launch: shadows with this optimizer's flat
        layout, vector-aligned extents (the fused kernel has no scalar tail)."""
        n_all = self.flat.numel()
        return (self.n % 4 == 0 and n_all % 4 == 0 and self.grad.numel() == n_all and 1 <= len(shadow_flats) <= 2
                and all(f is not None and f.numel() == n_all and f.device == self.flat.device and f.dtype == torch.float32
                        and f.data_ptr() % 16 == 0 for f in shadow_flats)
                and all(t.data_ptr() % 16 == 0 for t in (self.flat, self.grad, self.exp_avg, self.exp_avg_sq)))

    def step_zero_ema(self, shadow_flats, decays, grad_scale=1.0, do_step=True):
        """optimizer.step() -> optimizer.zero_grad() -> EMA of up to two shadow networks (tools/train_utils.py:177-183,
        255-282) as ONE pass over the training state (`ctta_adamw_ema2_zero`: 12 fp32 streams instead of 7 + 1 + 5);
        bit-identical to the three launches.  `do_step=False`: the update is skipped (NaN loss), the gradients are still
        zeroed and the shadows still move, as in the reference.  The caller has checked `can_fuse_tail` and that the
        shadows still live in `shadow_flats`."""
        g = self.param_groups[0]
        if not self.module.flat_is_current() or self.module._flat is not self.flat:
            raise N.CttaError("FusedAdamW: the module's parameters no longer alias the flat buffer this optimizer "
                              "was built on (model.to()/.float() after prepare_training?) -- build a new optimizer")
        if do_step:
            if not self.module.grads_alias_flat_sampled():
                self.module.realias_grads_()
                raise N.CttaError("FusedAdamW: parameter gradients were not views of the flat gradient buffer (zero_grad("
                                  "set_to_none=True) + a foreign backward?); they have been re-aliased -- redo the backward")
            self.step_count += 1
        with torch.cuda.device(self.flat.device):
            N.check(N.lib().ctta_adamw_ema2_zero(
                N.ptr(self.flat), N.ptr(self.grad), N.ptr(self.exp_avg), N.ptr(self.exp_avg_sq), self.n, self.flat.numel(),
                N.ptr(shadow_flats[0]), float(decays[0]), N.ptr(shadow_flats[1]) if len(shadow_flats) > 1 else N.c_void_p(0),
                float(decays[1]) if len(shadow_flats) > 1 else 0.0, 1 if do_step else 0, g["lr"], g["betas"][0], g["betas"][1],
                g["eps"], g["weight_decay"], max(1, self.step_count), float(grad_scale), N.stream_ptr()))
        if do_step:
            self.module.mark_weights_changed()

    # checkpoint / resume (accelerator.save_state stores the optimizer state, train.py:497-505): the layout is
    # torch.optim.AdamW.state_dict() over `student_unet.parameters()` (tools/train_utils.py:38-39,59-63), so that
    # optimizer.bin written here resumes a reference run and vice versa:
    #   {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [{..., 'params': [0..n_params)}]}
    # with i the parameter's position in `parameters()` order; parameters without a gradient (guidance_proj.weight)
    # have no state entry, exactly as torch leaves them.
    def _segments(self):
        """[(index in parameters() order, parameter, offset in the flat moment buffers)] of the trainable parameters."""
        frozen = set(getattr(self.module, "_frozen_keys", ()))
        segs, off = [], 0
        for i, (k, p) in enumerate(self.module.named_parameters()):
            if k in frozen:
                continue
            segs.append((i, p, off))
            off += p.numel()
        assert off == self.n
        return segs

    def state_dict(self):
        step = torch.tensor(float(self.step_count))
        state = {}
        if self.step_count > 0:   # torch creates a parameter's state at its first update
            for i, p, off in self._segments():
                n = p.numel()
                state[i] = {"step": step.clone(), "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
        n_params = sum(1 for _ in self.module.parameters())
        groups = []
        for g in self.param_groups:
            t = dict(g)
            t.setdefault("amsgrad", False)
            t.setdefault("maximize", False)
            t.setdefault("foreach", None)
            t.setdefault("capturable", False)
            t.setdefault("differentiable", False)
            t.setdefault("fused", None)
            t["params"] = list(range(n_params))
            groups.append(t)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if "state" not in sd:   # rounds 1-2 private layout {step, exp_avg, exp_avg_sq, param_groups} over the flat buffer
            self.step_count = int(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            self.param_groups = [dict(g) for g in sd["param_groups"]]
            return
        if len(sd["param_groups"]) != 1:
            raise ValueError("loaded state dict has a different number of parameter groups")
        n_params = sum(1 for _ in self.module.parameters())
        if len(sd["param_groups"][0]["params"]) != n_params:
            raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
        segs = {i: (p, off) for i, p, off in self._segments()}
        state = sd["state"]
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, st in state.items():
            i = int(i)
            if i not in segs:
                raise ValueError("optimizer state for parameter %d, which this model does not train" % i)
            p, off = segs[i]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("optimizer state %d: shape %s, parameter %s" % (i, tuple(st["exp_avg"].shape), tuple(p.shape)))
            n = p.numel()
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if state and len(state) != len(segs):
            raise ValueError("optimizer state covers %d of %d trainable parameters" % (len(state), len(segs)))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ (%s): the fused update keeps one" % sorted(steps))
        self.step_count = steps.pop() if steps else 0
        g = dict(sd["param_groups"][0])
        g.pop("params", None)
        g["betas"] = tuple(g["betas"])
        for k in ("amsgrad", "maximize"):
            if g.get(k):
                raise ValueError("FusedAdamW does not implement %s=True" % k)
        self.param_groups = [g]


class WarmupSchedule:
    """transformers.get_scheduler('linear' | 'constant' | 'constant_with_warmup') on a FusedAdamW."""

    def __init__(self, optimizer, name="linear", num_warmup_steps=0, num_training_steps=None, steps_per_update=1):
        """`steps_per_update`: accelerate's AcceleratedScheduler steps the wrapped LambdaLR `num_processes` times per
        optimizer update (which is why tools/train_utils.py:77 multiplies the warm-up by `accelerator.num_processes`).  To
        continue an N-GPU reference run -- its scheduler.bin holds last_epoch = N x updates and its warm-up / total counts
        are in those units -- pass steps_per_update=N and the reference's (already multiplied) step counts; the default 1
        is a single-process run, where the two conventions coincide."""
        self.steps_per_update = max(1, int(steps_per_update))
        if name not in ("linear", "constant", "constant_with_warmup"):
            raise ValueError("lr schedule '%s' is not built (linear, constant, constant_with_warmup)" % name)
        if name == "linear" and not num_training_steps:
            raise ValueError("linear schedule needs num_training_steps")
        self.opt, self.name, self.warm, self.total = optimizer, name, int(num_warmup_steps), num_training_steps
        self.base_lr = optimizer.param_groups[0]["lr"]
        self.last_step = 0
        self._apply()

    def _factor(self, s):
        if self.name == "constant":
            return 1.0
        if s < self.warm:
            return float(s) / float(max(1, self.warm))
        if self.name == "constant_with_warmup":
            return 1.0
        return max(0.0, float(self.total - s) / float(max(1, self.total - self.warm)))

    def _apply(self):
        self.opt.param_groups[0]["lr"] = self.base_lr * self._factor(self.last_step)

    def step(self):
        self.last_step += self.steps_per_update
        self._apply()

    def get_last_lr(self):
        return [self.opt.param_groups[0]["lr"]]

    # resume (accelerator.save_state stores the scheduler too, train.py:497-505): torch.optim.lr_scheduler.LambdaLR's
    # own keys (what transformers.get_scheduler returns), so scheduler.bin is interchangeable with the reference's;
    # the schedule's shape (name / warm-up / total) is a constructor argument there too and rides along as extra keys.
    def state_dict(self):
        return {"base_lrs": [self.base_lr], "last_epoch": self.last_step, "_step_count": self.last_step + 1,
                "_get_lr_called_within_step": False, "_last_lr": self.get_last_lr(), "lr_lambdas": [None],
                "name": self.name, "num_warmup_steps": self.warm, "num_training_steps": self.total}

    def load_state_dict(self, sd):
        if "last_epoch" in sd:
            self.base_lr, self.last_step = float(sd["base_lrs"][0]), int(sd["last_epoch"])
            if "name" in sd:
                self.name, self.warm, self.total = sd["name"], int(sd["num_warmup_steps"]), sd["num_training_steps"]
        else:   # rounds 1-2 private layout
            self.name, self.warm, self.total = sd["name"], int(sd["num_warmup_steps"]), sd["num_training_steps"]
            self.base_lr, self.last_step = float(sd["base_lr"]), int(sd["last_step"])
        self._apply()
