"""Drop-in nn.Module mirrors of the reference's numerical modules, backed by libctta_hip.so.

Same constructor configs, `forward` signatures, `state_dict` key names / order and error
behaviour as
  * diffusers.UNet2DConditionGuidedModel  (diffusers/models/unet_2d_condition_guided.py:51)
  * diffusers.UNet2DConditionModel        (diffusers/models/unet_2d_condition.py)  [teacher]
  * audioldm AutoencoderKL                (audioldm/variational_autoencoder/autoencoder.py:10)
  * audioldm.hifigan.Generator            (audioldm/hifigan/models.py:72)
so the reference's L4 callers (`models/audio_consistency_model.py`, `inference.py`,
`easy_inference/consistencytta.py`) can switch by changing an import.  The modules only OWN
parameters (fp32, reference layout); all arithmetic happens in the HIP engines, which keep
their own packed bf16 copies and are re-synced when parameter versions change.
"""
import json
import math
from collections import OrderedDict
from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

from . import _native as N
from . import spec


@dataclass
class UNet2DConditionOutput:
    sample: torch.Tensor


class _ParamTree(nn.Module):
    """Registers parameters under dotted reference key names (nested anonymous modules)."""

    def _register(self, table, frozen=()):
        self._frozen_keys = tuple(frozen)
        for key, shape in table.items():
            mod = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, nn.Module())
                mod = mod._modules[p]
            param = nn.Parameter(torch.empty(*shape, dtype=torch.float32), requires_grad=key not in frozen)
            mod.register_parameter(parts[-1], param)

    def init_deterministic(self, seed=0, prefix=""):
        """Random init from the build's deterministic generator (spec.det_weight)."""
        with torch.no_grad():
            for k, p in self.named_parameters():
                p.copy_(torch.from_numpy(spec.det_weight(prefix + k, tuple(p.shape), seed)).to(p.device))
        return self

    def init_random_(self, seed=0, prefix=""):
        """Fast on-device random init with the same scale rule (benchmarks: values need not be
        reproducible across boxes, only well-conditioned)."""
        gen = torch.Generator(device=self.device if hasattr(self, "device") else "cpu")
        gen.manual_seed(seed)
        with torch.no_grad():
            for k, p in self.named_parameters():
                off, amp = spec.weight_rule(prefix + k, tuple(p.shape))
                p.copy_(off + amp * (torch.rand(p.shape, generator=gen, device=p.device) * 2 - 1))
        return self

    def _weights_version(self):
        return sum(p._version for p in self.parameters()) + 1000003 * sum(
            p.data_ptr() % 1000003 for p in self.parameters()) + 7 * getattr(self, "_manual_version", 0)

    def mark_weights_changed(self):
        """Call after parameters were updated through raw pointers (fused AdamW / EMA kernels)."""
        self._manual_version = getattr(self, "_manual_version", 0) + 1

    def flatten_parameters_(self):
        """Re-homes every parameter into ONE contiguous fp32 buffer (views keep names/shapes), so the
        optimizer, EMA and gradient all-reduce run as single launches over the whole model.
        Returns the flat buffer; `flat_grad_()` gives the matching gradient buffer."""
        params = self._flat_order()
        flat0 = getattr(self, "_flat", None)
        if flat0 is not None and all(
                flat0.data_ptr() <= p.data_ptr() < flat0.data_ptr() + flat0.numel() * 4 for p in params):
            return flat0
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, dtype=torch.float32, device=params[0].device)
        off = 0
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            off += n
        self._flat = flat
        self._flat_grad = None
        self._flat_checked_at = getattr(self, "_rehome_count", 0)
        return flat

    # Everything that re-homes `p.data` through the nn.Module protocol (.to() / .float() / .cuda() / ... and
    # load_state_dict(assign=True)) bumps a counter, so `flat_is_current()` is O(1) on the steady-state training step
    # instead of a walk over 690 parameters per network (0.9 ms of host time per EMA update in round 2's bench).
    def _apply(self, fn, recurse=True):
        self._rehome_count = getattr(self, "_rehome_count", 0) + 1
        return super()._apply(fn, recurse)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        if assign:
            self._rehome_count = getattr(self, "_rehome_count", 0) + 1
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def flat_is_current(self):
        """True while every parameter still aliases `_flat` (a later `.to()`, `.float()`, `load_state_dict(assign=True)`
        ... re-homes `p.data` and would leave the fused optimizer / EMA kernels updating a stale buffer).  The walk over
        the parameters runs once per re-homing event (see `_apply`); assigning `p.data` by hand is not tracked."""
        flat = getattr(self, "_flat", None)
        if flat is None:
            return False
        epoch = getattr(self, "_rehome_count", 0)
        if getattr(self, "_flat_checked_at", None) == epoch:
            return True
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        ok = all(lo <= p.data_ptr() < hi for p in self.parameters())
        if ok:
            self._flat_checked_at = epoch
        return ok

    def grads_alias_flat(self):
        """True while every trainable parameter's `.grad` is a view of `_flat_grad` (`zero_grad(set_to_none=True)`
        followed by a backward that allocates fresh gradients would break this)."""
        g = getattr(self, "_flat_grad", None)
        if g is None:
            return False
        lo, hi = g.data_ptr(), g.data_ptr() + g.numel() * 4
        return all(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in self.parameters() if p.requires_grad)

    def grads_alias_flat_sampled(self, k=8):
        """The per-step form of `grads_alias_flat` (called by FusedAdamW.step): the first and the last trainable parameter
        plus `k` more, rotating from call to call.  What breaks the aliasing in practice -- `zero_grad(set_to_none=True)`
        followed by a backward that allocates fresh gradients -- breaks it for EVERY parameter and is seen by the very
        next call; a single hand-assigned `p.grad` is seen within n / k calls.  The full walk costs 690 `p.grad` look-ups =
        1.5-3 ms of host time per optimizer step, more than the AdamW kernel itself on a slow host (round 4: 3.48 vs
        2.78 ms for the same launch on two boxes -- it was this walk inside the timed bracket)."""
        g = getattr(self, "_flat_grad", None)
        if g is None:
            return False
        tr = getattr(self, "_trainable_cache", None)
        key = (getattr(self, "_rehome_count", 0), tuple(getattr(self, "_frozen_keys", ())))
        if tr is None or tr[0] != key:
            tr = self._trainable_cache = (key, [p for p in self.parameters() if p.requires_grad])
            self._alias_cursor = 0
            self._alias_calls = 0
            return self.grads_alias_flat()
        ps = tr[1]
        n = len(ps)
        if n == 0:
            return True
        # every 64th call walks everything (a single hand-assigned p.grad is then seen within 64 steps whatever n / k is)
        self._alias_calls = getattr(self, "_alias_calls", 0) + 1
        if self._alias_calls % 64 == 0:
            return self.grads_alias_flat()
        lo, hi = g.data_ptr(), g.data_ptr() + g.numel() * 4
        cur = self._alias_cursor
        idx = [0, n - 1] + [(cur + i) % n for i in range(k)]
        self._alias_cursor = (cur + k) % n
        for i in idx:
            p = ps[i]
            if not p.requires_grad:       # frozen after the cache was built (requires_grad_(False) by hand): rebuild, walk
                self._trainable_cache = None
                return self.grads_alias_flat()
            gr = p.grad
            if gr is None or not (lo <= gr.data_ptr() < hi):
                # the sample says no: let the full walk decide (it filters on requires_grad at call time) before step() raises
                return self.grads_alias_flat()
        return True

    def realias_grads_(self):
        """Points every trainable `p.grad` back at its slice of the flat gradient buffer (after set_to_none)."""
        g = getattr(self, "_flat_grad", None)
        if g is None:
            return
        off = 0
        for p in self._flat_order():
            n = p.numel()
            if p.requires_grad and (p.grad is None or p.grad.data_ptr() != g.data_ptr() + off * 4):
                p.grad = g[off:off + n].view(p.shape)
            off += n

    def flat_grad_(self):
        """Flat fp32 gradient buffer aliased by every `p.grad` (zero-initialised on first use)."""
        flat = self.flatten_parameters_()
        if getattr(self, "_flat_grad", None) is None:
            g = torch.zeros_like(flat)
            off = 0
            for p in self._flat_order():
                n = p.numel()
                if p.requires_grad:
                    p.grad = g[off:off + n].view(p.shape)
                off += n
            self._flat_grad = g
        return self._flat_grad

    def _flat_order(self):
        """Trainable parameters first (state-dict order), frozen ones last, so optimizer / all-reduce
        kernels cover exactly the prefix `[0, n_trainable())` of the flat buffers."""
        frozen = set(getattr(self, "_frozen_keys", ()))
        named = list(self.named_parameters())
        return [p for k, p in named if k not in frozen] + [p for k, p in named if k in frozen]

    def n_trainable(self):
        frozen = set(getattr(self, "_frozen_keys", ()))
        return sum(p.numel() for k, p in self.named_parameters() if k not in frozen)

    def _table(self, prefix_filter=None, strip=""):
        sd = OrderedDict()
        for k, p in self.named_parameters():
            if prefix_filter and not k.startswith(prefix_filter):
                continue
            if not p.is_cuda:
                raise N.CttaError(
                    "parameter '%s' is on %s: the HIP engine needs CUDA(ROCm) tensors; there is no CPU path"
                    % (k, p.device))
            sd[k[len(strip):] if strip and k.startswith(strip) else k] = p.detach()
        return sd

    # handles are not copyable / picklable: a deepcopy (audio_consistency_model.py:65) gets fresh ones
    def __getstate__(self):
        d = self.__dict__.copy()
        for k in list(d):
            if k.startswith("_h_"):
                d[k] = None
        if "_eng" in d:
            d["_eng"] = {"vae": [None, None], "voc": [None, None]}
            d["_vae_pending"] = d["_voc_pending"] = None
        return d

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_eng":
                new.__dict__[k] = {"vae": [None, None], "voc": [None, None]}
            elif k in ("_vae_pending", "_voc_pending"):
                new.__dict__[k] = None
            else:
                new.__dict__[k] = None if k.startswith(("_h_", "_flat")) else copy.deepcopy(v, memo)
        return new


def _cfg_get(cfg, key, default=None):
    return cfg[key] if key in cfg else default


class _UNetBase(_ParamTree):
    _guided = True

    def __init__(self, **config):
        super().__init__()
        cfg = dict(spec.LIGHT_UNET_CONFIG)
        cfg.update({k: v for k, v in config.items() if not k.startswith("_")})
        for k in ("down_block_types", "up_block_types"):
            for t in cfg[k]:
                if t not in ("CrossAttnDownBlock2D", "DownBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"):
                    raise ValueError(f"{t} does not exist.")  # unet_2d_blocks.py get_down_block
        if len(cfg["down_block_types"]) != len(cfg["up_block_types"]):
            raise ValueError("Must provide the same number of `down_block_types` as `up_block_types`.")
        if len(cfg["block_out_channels"]) != len(cfg["down_block_types"]):
            raise ValueError("Must provide the same number of `block_out_channels` as `down_block_types`.")
        if not cfg.get("use_linear_projection", True):
            raise ValueError("use_linear_projection=False (conv proj_in/out) is not supported by the HIP engine")
        self.config = SimpleNamespace(**cfg)
        self._cfg = cfg
        frozen = ("guidance_proj.weight",)  # embeddings.py:229 requires_grad=False
        self._register(spec.unet_param_spec(cfg, self._guided), frozen)
        self._h_unet = None
        self._h_key = None
        self._h_version = None
        self.debug_taps = False
        self.enable_training = False   # True: the native handle also carries the backward pass

    # ---- config plumbing (configuration_utils.py:161,256)
    @classmethod
    def load_config(cls, path, **kwargs):
        with open(path) as f:
            return json.load(f)

    @classmethod
    def from_config(cls, config, subfolder=None, **kwargs):
        return cls(**{k: v for k, v in dict(config).items() if not k.startswith("_")})

    @property
    def device(self):
        return next(self.parameters()).device

    # ---- native handle management
    def _native_config(self, B, H, W, L):
        cfg = self._cfg
        boc, heads, layers = spec.unet_levels(cfg)
        c = N.UNetConfig()
        c.in_channels, c.out_channels, c.n_levels = cfg["in_channels"], cfg["out_channels"], len(boc)
        for i in range(len(boc)):
            c.block_out_channels[i] = boc[i]
            c.heads[i] = heads[i]
            c.layers_per_block[i] = layers[i]
            c.down_cross[i] = int(cfg["down_block_types"][i] == "CrossAttnDownBlock2D")
            c.up_cross[i] = int(cfg["up_block_types"][i] == "CrossAttnUpBlock2D")
        c.cross_attention_dim = cfg["cross_attention_dim"]
        c.norm_num_groups = cfg["norm_num_groups"]
        c.norm_eps = cfg["norm_eps"]
        c.flip_sin_to_cos = int(cfg["flip_sin_to_cos"])
        c.freq_shift = float(cfg["freq_shift"])
        c.guided = int(self._guided)
        c.max_batch, c.height, c.width, c.max_text_len = B, H, W, L
        c.debug_taps = int(self.debug_taps)
        c.enable_training = int(self.enable_training)
        return c

    def _release(self):
        if getattr(self, "_h_unet", None):
            N.lib().ctta_unet_destroy(self._h_unet)
        self._h_unet = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure(self, B, H, W, L):
        L_ = N.lib()
        key = self._h_key
        need_new = (self._h_unet is None or key is None or key[1:3] != (H, W) or B > key[0] or L > key[3]
                    or key[4] != (self.debug_taps, self.enable_training) or key[5] != self.device)
        ver = self._weights_version()
        if need_new:
            self._release()
            Bm = max(B, key[0]) if key and key[1:3] == (H, W) else B
            Lm = max(L, key[3], 32) if key else max(L, 32)
            table, keep = N.tensor_table(self._table())
            h = N.c_void_p()
            cfg = self._native_config(Bm, H, W, Lm)
            with torch.cuda.device(self.device):
                N.check(L_.ctta_unet_create(cfg, table, len(table), N.stream_ptr(), h))
            self._h_unet, self._h_key, self._h_version = (
                h, (Bm, H, W, Lm, (self.debug_taps, self.enable_training), self.device), ver)
        elif ver != self._h_version:  # parameters changed (optimizer / EMA / load_state_dict)
            table, keep = N.tensor_table(self._table())
            N.check(L_.ctta_unet_load_weights(self._h_unet, table, len(table), N.stream_ptr()))
            self._h_version = ver

    @staticmethod
    def _per_sample(v, B, device, dtype):
        """_prepare_tensor + expand (unet_2d_condition_guided.py:699-714,803,810)."""
        if not torch.is_tensor(v):
            return torch.full((B,), float(v), dtype=dtype, device=device)
        v = v.to(device=device, dtype=dtype)
        if v.ndim == 0:
            v = v[None]
        return v.expand(B).contiguous()

    def _text_unchanged(self, enc, mask, B, reuse_text):
        """Whether this forward may take the cross-attention K / V projections of the text states from the handle's text
        cache (ctta_unet_reuse_text): the SAME tensor objects as in the previous forward on this module, unmodified since
        (identity + version counters; the module keeps a reference, so the storage cannot have been recycled).  Inside a
        hipGraph capture only on the caller's explicit word (`reuse_text=True`): a replay cannot re-decide."""
        key = (enc, enc._version, mask, None if mask is None else mask._version, B)
        prev = getattr(self, "_text_key", None)
        same = (prev is not None and prev[0] is key[0] and prev[1] == key[1] and prev[2] is key[2] and prev[3] == key[3]
                and prev[4] == key[4])
        # the key of THIS forward is committed by `_forward` after the native call has returned OK (a failed forward must
        # not leave its text states registered as "the cached ones").  Identity + `_version` cannot see writes through raw
        # pointers (how this library's own kernels fill tensors): a caller that refills a persistent text-state buffer
        # natively must pass reuse_text=False.
        self._text_key_pending = key
        if reuse_text is False or not same:
            return False
        if reuse_text is True:
            return True
        return not torch.cuda.is_current_stream_capturing()

    def _forward(self, sample, timestep, guidance, encoder_hidden_states, encoder_attention_mask, train=False,
                 reuse_text=None):
        if sample.ndim != 4 or sample.shape[1] != self._cfg["in_channels"]:
            raise ValueError("sample must be (batch, %d, height, width), got %s"
                             % (self._cfg["in_channels"], tuple(sample.shape)))
        if encoder_hidden_states is None or encoder_hidden_states.ndim != 3:
            raise ValueError("encoder_hidden_states must be (batch, sequence_length, feature_dim)")
        B, _, H, W = sample.shape
        L = encoder_hidden_states.shape[1]
        if encoder_hidden_states.shape[0] != B or encoder_hidden_states.shape[2] != self._cfg["cross_attention_dim"]:
            raise ValueError("encoder_hidden_states shape %s does not match batch %d / cross_attention_dim %d"
                             % (tuple(encoder_hidden_states.shape), B, self._cfg["cross_attention_dim"]))
        dev = self.device
        if not sample.is_cuda:
            raise N.CttaError("sample is on %s: the HIP engine has no CPU path" % sample.device)
        had_handle, ver0 = self._h_unet is not None, self._h_version
        self._ensure(B, H, W, L)
        reuse = self._text_unchanged(encoder_hidden_states, encoder_attention_mask, B, reuse_text)
        reuse = reuse and not train and had_handle and ver0 == self._h_version       # same handle, same packed weights
        x = sample.detach().to(device=dev, dtype=torch.float32).contiguous()
        enc = encoder_hidden_states.detach().to(device=dev, dtype=torch.float32).contiguous()
        t = self._per_sample(timestep, B, dev, torch.float32)  # get_timestep_embedding casts to fp32
        g = self._per_sample(guidance, B, dev, torch.float64) if self._guided else None
        m = None
        if encoder_attention_mask is not None:
            m = encoder_attention_mask.to(device=dev).reshape(B, L).to(torch.uint8).contiguous()
        out = torch.empty((B, self._cfg["out_channels"], H, W), dtype=torch.float32, device=dev)
        fn = N.lib().ctta_unet_forward_train if train else N.lib().ctta_unet_forward
        self._text_key = None
        with torch.cuda.device(dev):
            if reuse:
                N.check(N.lib().ctta_unet_reuse_text(self._h_unet, 1))
            N.check(fn(self._h_unet, N.ptr(x), N.ptr(t), N.ptr(g), N.ptr(enc), N.ptr(m), B, L, N.ptr(out),
                       N.stream_ptr()))
        self._text_key = getattr(self, "_text_key_pending", None)
        return out

    # ---- distillation step: the reference differentiates `forward` with torch autograd
    # (train.py:332-346); here the engine keeps the activations and runs its own backward pass.
    def forward_train(self, sample, timestep, guidance, encoder_hidden_states, encoder_attention_mask=None):
        """`forward` that keeps what `backward` needs.  Returns the prediction (B,C,H,W) fp32, detached."""
        if not self.enable_training:
            self.enable_training = True
        return self._forward(sample, timestep, guidance if self._guided else None, encoder_hidden_states,
                             encoder_attention_mask, train=True)

    def backward(self, grad_output=None, grad_output_nhwc=None, on_block_done=None):
        """Accumulates dL/d(param) of the last `forward_train` into `p.grad` (fp32, allocated on first
        use).  `grad_output`: dL/d(prediction) (B,C,H,W) fp32, or `grad_output_nhwc`: the same as the
        (B, H*W, 8) bf16 tensor ctta_snr_mse_grad writes."""
        dev = self.device
        if grad_output_nhwc is None:
            B, C, H, W = grad_output.shape
            g = torch.zeros(B, H * W, 8, dtype=torch.bfloat16, device=dev)
            g[:, :, :C] = grad_output.detach().to(dev).permute(0, 2, 3, 1).reshape(B, H * W, C).to(torch.bfloat16)
            grad_output_nhwc = g
        tab, keep = self._grad_table()
        g = grad_output_nhwc.contiguous()
        with torch.cuda.device(dev):
            if on_block_done is None:
                N.check(N.lib().ctta_unet_backward(self._h_unet, N.ptr(g), tab, len(tab), N.stream_ptr()))
                return
        # block-wise: after each step the gradients of one block are final; the callback typically starts
        # the all-reduce of that block's slice of the flat gradient buffer (dist_util.GradientBuckets)
        on_block_done(self.backward_begin(g))
        fin = False
        while not fin:
            blk, fin = self.backward_next()
            on_block_done(blk)

    def _grad_table(self):
        table = OrderedDict()
        if getattr(self, "_flat_grad", None) is not None and not self.grads_alias_flat():
            self.realias_grads_()    # e.g. after zero_grad(set_to_none=True): the fused optimizer reads the flat buffer
        for k, p in self.named_parameters():
            if not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = torch.zeros_like(p.data)
            table[k] = p.grad
        return N.tensor_table(table)

    def backward_begin(self, grad_output_nhwc):
        """First step of the block-wise backward (ctta_unet_backward_begin): the out head.  Returns its block id.  Each
        begin / next call is self-contained on the calling stream (the weight-gradient side stream is joined before it
        returns), so a caller may capture every call into its own hipGraph (`_DistillStepGraph(segmented=True)`)."""
        tab, keep = self._bwd_table = self._grad_table()     # kept for the `backward_next` calls of THIS backward pass
        with torch.cuda.device(self.device):
            N.check(N.lib().ctta_unet_backward_begin(self._h_unet, N.ptr(grad_output_nhwc), tab, len(tab), N.stream_ptr()))
        return 2 * len(self._cfg["block_out_channels"]) + 2

    def backward_next(self):
        """One more block (ctta_unet_backward_next) -> (block id, finished)."""
        # the table `backward_begin` built (690 named gradient tensors: rebuilding it for each of the ~12 calls of one
        # block-wise backward was host time on the data-parallel step's critical path; the gradient tensors cannot change
        # between the calls of one pass)
        tab, keep = getattr(self, "_bwd_table", None) or self._grad_table()
        blk, fin = N.c_int(0), N.c_int(0)
        with torch.cuda.device(self.device):
            N.check(N.lib().ctta_unet_backward_next(self._h_unet, tab, len(tab), N.stream_ptr(), N.byref(blk), N.byref(fin)))
        if fin.value:
            self._bwd_table = None
        return blk.value, bool(fin.value)

    def block_ranges(self):
        """block id (see ctta_unet_backward_next) -> (start, end) element range of that block's parameters in the
        flat parameter / gradient buffers (trainable parameters only; state-dict order keeps blocks contiguous)."""
        n_levels = len(self._cfg["block_out_channels"])

        def block_of(key):
            head, _, rest = key.partition(".")
            if head == "down_blocks":
                return 1 + int(rest.split(".")[0])
            if head == "mid_block":
                return 1 + n_levels
            if head == "up_blocks":
                return 2 + n_levels + int(rest.split(".")[0])
            if head in ("conv_norm_out", "conv_out"):
                return 2 * n_levels + 2
            return 0   # conv_in, time_embedding, guidance_embedding
        frozen = set(getattr(self, "_frozen_keys", ()))
        ranges, off = {}, 0
        for k, p in self.named_parameters():
            if k in frozen:
                continue
            b = block_of(k)
            lo, hi = ranges.get(b, (off, off))
            if hi != off:
                raise RuntimeError("parameters of block %d are not contiguous in the flat buffer" % b)
            ranges[b] = (lo, off + p.numel())
            off += p.numel()
        return ranges

    def read_taps(self):
        """name -> NCHW fp32 tensor of every recorded intermediate (debug_taps=True only)."""
        L_ = N.lib()
        out = OrderedDict()
        for i in range(L_.ctta_unet_num_taps(self._h_unet)):
            name = N.c_char_p()
            dims = (N.c_int * 4)()
            N.check(L_.ctta_unet_tap_info(self._h_unet, i, name, dims))
            t = torch.empty(tuple(dims), dtype=torch.float32, device=self.device)
            N.check(L_.ctta_unet_tap_read(self._h_unet, i, N.ptr(t), N.stream_ptr()))
            out[name.value.decode()] = t
        return out


class UNet2DConditionGuidedModel(_UNetBase):
    """Student / consistency U-Net with the guidance-strength Fourier input."""
    _guided = True

    def forward(self, sample, timestep, guidance, encoder_hidden_states, class_labels=None, timestep_cond=None,
                guidance_cond=None, attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None,
                encoder_attention_mask=None, return_dict=True, **kwargs):
        for name, v in (("class_labels", class_labels), ("timestep_cond", timestep_cond),
                        ("guidance_cond", guidance_cond), ("attention_mask", attention_mask),
                        ("down_block_additional_residuals", down_block_additional_residuals),
                        ("mid_block_additional_residual", mid_block_additional_residual)):
            if v is not None:
                raise NotImplementedError("%s is not used on the ConsistencyTTA path and is not supported" % name)
        out = self._forward(sample, timestep, guidance, encoder_hidden_states, encoder_attention_mask,
                            reuse_text=kwargs.get("reuse_text"))
        return UNet2DConditionOutput(sample=out) if return_dict else (out,)


class UNet2DConditionModel(_UNetBase):
    """Teacher diffusion U-Net; `forward(**kwargs)` swallows `guidance=` like the reference
    (unet_2d_condition.py:668-690)."""
    _guided = False

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, timestep_cond=None,
                attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None,
                encoder_attention_mask=None, return_dict=True, **kwargs):
        out = self._forward(sample, timestep, None, encoder_hidden_states, encoder_attention_mask,
                            reuse_text=kwargs.get("reuse_text"))
        return UNet2DConditionOutput(sample=out) if return_dict else (out,)


# ------------------------------------------------------------------------------------------
class DiagonalGaussianDistribution(object):
    """audioldm/variational_autoencoder/distributions.py:24-74 (elementwise torch ops on the moments the HIP
    encoder produced; sampling draws from the global RNG exactly like the reference)."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean).to(device=self.parameters.device)

    def sample(self):
        return self.mean + self.std * torch.randn(self.mean.shape).to(device=self.parameters.device)

    def kl(self, other=None):
        if self.deterministic:
            return torch.Tensor([0.0])
        if other is None:
            return 0.5 * torch.mean(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=[1, 2, 3])
        return 0.5 * torch.mean(torch.pow(self.mean - other.mean, 2) / other.var + self.var / other.var - 1.0
                                - self.logvar + other.logvar, dim=[1, 2, 3])

    def mode(self):
        return self.mean


class _VaeDecodeWithGrad(torch.autograd.Function):
    """decode_first_stage under torch.set_grad_enabled(True) (autoencoder.py:103-106) for a frozen decoder: the HIP
    engine keeps its own saved tensors between the two calls; one forward may be pending per AutoencoderKL."""

    @staticmethod
    def forward(ctx, z, vae):
        B, _, T, F = z.shape
        h = vae._ensure_vae(B, T, F, grad=True)
        zz = z.detach().to(device=vae.device, dtype=torch.float32).contiguous()
        up = 2 ** (len(vae.ddconfig["ch_mult"]) - 1)
        mel = torch.empty((B, vae.ddconfig["out_ch"], T * up, F * up), dtype=torch.float32, device=vae.device)
        with torch.cuda.device(vae.device):
            N.check(N.lib().ctta_vae_decode_with_grad(h, N.ptr(zz), B, N.ptr(mel), N.stream_ptr()))
        ctx.vae, ctx.shape, ctx.dtype = vae, tuple(z.shape), z.dtype
        ctx.token = vae._vae_pending = object()
        return mel

    @staticmethod
    def backward(ctx, grad_mel):
        vae = ctx.vae
        if vae._vae_pending is not ctx.token:
            raise N.CttaError("decode_first_stage(allow_grad=True): the decoder ran again (or its handle was rebuilt) "
                              "before this backward -- only the latest differentiable decode can be back-propagated")
        g = grad_mel.to(device=vae.device, dtype=torch.float32).contiguous()
        gz = torch.empty(ctx.shape, dtype=torch.float32, device=vae.device)
        with torch.cuda.device(vae.device):
            N.check(N.lib().ctta_vae_decode_backward(vae._eng["vae"][1]["h"], N.ptr(g), ctx.shape[0], N.ptr(gz), N.stream_ptr()))
        vae._vae_pending = None
        return gz.to(ctx.dtype), None


class _VocodeWithGrad(torch.autograd.Function):
    """vocoder(mels) of vocoder_infer(allow_grad=True) (hifigan/utilities.py:79-80) for the frozen generator."""

    @staticmethod
    def forward(ctx, mel, vae):
        if mel.ndim != 4 or mel.shape[1] != 1 or mel.shape[3] != vae.vocoder.h["num_mels"]:
            raise ValueError("mel must be (batch, 1, T, %d), got %s" % (vae.vocoder.h["num_mels"], tuple(mel.shape)))
        if not mel.is_cuda:
            raise N.CttaError("mel is on %s: the HIP engine has no CPU path" % mel.device)
        B, _, T, F = mel.shape
        h = vae._ensure_voc(B, T, grad=True)
        m = mel.detach().to(device=vae.device, dtype=torch.float32).contiguous()
        n = N.lib().ctta_hifigan_out_len(h, T)
        wav = torch.empty((B, n), dtype=torch.float32, device=vae.device)
        with torch.cuda.device(vae.device):
            N.check(N.lib().ctta_hifigan_forward_with_grad(h, N.ptr(m), B, T, N.ptr(wav), N.stream_ptr()))
        ctx.vae, ctx.shape, ctx.dtype = vae, tuple(mel.shape), mel.dtype
        ctx.token = vae._voc_pending = object()
        ctx.save_for_backward(wav)
        return wav

    @staticmethod
    def backward(ctx, grad_wav):
        vae = ctx.vae
        if vae._voc_pending is not ctx.token:
            raise N.CttaError("decode_to_waveform(allow_grad=True): the vocoder ran again (or its handle was rebuilt) "
                              "before this backward -- only the latest differentiable call can be back-propagated")
        (wav,) = ctx.saved_tensors
        B, _, T, F = ctx.shape
        g = grad_wav.to(device=vae.device, dtype=torch.float32).contiguous()
        gm = torch.empty(ctx.shape, dtype=torch.float32, device=vae.device)
        with torch.cuda.device(vae.device):
            N.check(N.lib().ctta_hifigan_backward(vae._eng["voc"][1]["h"], N.ptr(g), N.ptr(wav), B, T, N.ptr(gm), N.stream_ptr()))
        vae._voc_pending = None
        return gm.to(ctx.dtype), None


class Generator(_ParamTree):
    """HiFi-GAN parameter holder (weight_norm already removed, hifigan/utilities.py:71)."""

    def __init__(self, h=None):
        super().__init__()
        self.h = dict(spec.HIFIGAN_16K_64 if h is None else h)
        self._register(spec.hifigan_param_spec(self.h, prefix=""))


class AutoencoderKL(_ParamTree):
    """AudioLDM VAE (autoencoder.py:10-132): `decode_first_stage` / `decode_to_waveform` (generation) and
    `encode_first_stage` / `get_first_stage_encoding` (the training-side latent encoder, tools/train_utils.py:155-162)."""

    def __init__(self, ddconfig=None, lossconfig=None, image_key="fbank", embed_dim=None, time_shuffle=1,
                 subband=1, ckpt_path=None, reload_from_ckpt=None, ignore_keys=(), colorize_nlabels=None,
                 monitor=None, base_learning_rate=1e-5, scale_factor=1, hifigan_config=None, **ignored):
        super().__init__()
        self.ddconfig = dict(spec.VAE_DDCONFIG if ddconfig is None else ddconfig)
        self.embed_dim = int(embed_dim if embed_dim is not None else 8)
        if int(subband) != 1:
            raise NotImplementedError("subband decomposition is unused by ConsistencyTTA (subband=1)")
        if self.ddconfig.get("attn_resolutions"):
            raise NotImplementedError("attn_resolutions must be empty (audioldm-s-full)")
        self.subband = 1
        self.image_key = image_key
        self.scale_factor = scale_factor
        # reference state-dict order: encoder.*, decoder.*, quant_conv.*, post_quant_conv.*, vocoder.*
        enc = spec.vae_encoder_param_spec(self.ddconfig, self.embed_dim)
        dec = spec.vae_decoder_param_spec(self.ddconfig, self.embed_dim)
        table = OrderedDict((k, v) for k, v in enc.items() if k.startswith("encoder."))
        table.update((k, v) for k, v in dec.items() if k.startswith("decoder."))
        table.update((k, v) for k, v in enc.items() if k.startswith("quant_conv."))
        table.update((k, v) for k, v in dec.items() if k.startswith("post_quant_conv."))
        self._register(table)
        self.vocoder = Generator(hifigan_config)
        self._h_enc = self._h_enc_key = self._h_enc_ver = None
        self.ema_decoder = None
        # engine handles: [plain, differentiable] per stage -- separate handles, so that a plain decode (e.g. of the
        # target latent in MelLoss) never disturbs the tensors a pending differentiable decode saved for its backward
        self._eng = {"vae": [None, None], "voc": [None, None]}
        self._h_vae = self._h_voc = None     # the handle each stage used last (debug taps)
        self._vae_pending = self._voc_pending = None
        self.debug_taps = False

    @property
    def device(self):
        return next(self.parameters()).device

    def load_state_dict(self, state_dict, strict=True):
        """Reference checkpoints also carry loss.* (discriminator) and ema_* keys, which this path never uses.
        A decoder-only state dict (generation checkpoints without the encoder) loads too: missing encoder.* /
        quant_conv.* keys are tolerated and `encode*` then raises until they are provided."""
        own = {k: v for k, v in state_dict.items() if not k.startswith(("loss.", "ema_"))}
        has_enc = any(k.startswith("encoder.") for k in own)
        self._encoder_loaded = has_enc or getattr(self, "_encoder_loaded", False)
        if not has_enc:
            mine = self.state_dict()
            own.update({k: v for k, v in mine.items() if k.startswith(("encoder.", "quant_conv."))})
        return super().load_state_dict(own, strict=strict)

    def init_deterministic(self, seed=0, prefix=""):
        self._encoder_loaded = True
        return super().init_deterministic(seed, prefix)

    def init_random_(self, seed=0, prefix=""):
        self._encoder_loaded = True
        return super().init_random_(seed, prefix)

    # ---- encoder, autoencoder.py:80-89,123-132
    def _ensure_enc(self, B, T, F):
        L_ = N.lib()
        nres = len(self.ddconfig["ch_mult"])
        s = 1 << (nres - 1)
        if T % s or F % s:
            raise ValueError("mel extent %dx%d must be a multiple of %d" % (T, F, s))
        key = (T, F, self.debug_taps, self.device)
        ver = sum(p._version for k, p in self.named_parameters() if k.startswith(("encoder.", "quant_conv.")))
        if self._h_enc is None or self._h_enc_key[1:] != key or B > self._h_enc_key[0] or ver != self._h_enc_ver:
            if self._h_enc is not None:
                L_.ctta_vae_encoder_destroy(self._h_enc)
                self._h_enc = None
            dd = self.ddconfig
            c = N.VAEConfig()
            c.z_channels, c.embed_dim, c.ch, c.out_ch = dd["z_channels"], self.embed_dim, dd["ch"], dd["in_channels"]
            c.n_levels, c.num_res_blocks = nres, dd["num_res_blocks"]
            for i, m in enumerate(dd["ch_mult"]):
                c.ch_mult[i] = m
            c.scale_factor = float(self.scale_factor)
            c.max_batch, c.latent_h, c.latent_w = B, T // s, F // s
            c.debug_taps = int(self.debug_taps)
            sd = OrderedDict((k, p.detach()) for k, p in self.named_parameters()
                             if k.startswith(("encoder.", "quant_conv.")))
            for k, p in sd.items():
                if not p.is_cuda:
                    raise N.CttaError("parameter '%s' is on %s: the HIP engine has no CPU path" % (k, p.device))
            table, keep = N.tensor_table(sd)
            h = N.c_void_p()
            with torch.cuda.device(self.device):
                N.check(L_.ctta_vae_encoder_create(c, table, len(table), N.stream_ptr(), h))
            self._h_enc, self._h_enc_key, self._h_enc_ver = h, (B,) + key, ver
        return self._h_enc

    @torch.no_grad()
    def encode(self, x):
        """mel (B,1,T,F) -> DiagonalGaussianDistribution over the latent (B, embed_dim, T/4, F/4)."""
        if not getattr(self, "_encoder_loaded", False):
            raise RuntimeError("this AutoencoderKL was loaded from a decoder-only state dict: no encoder weights")
        if x.ndim != 4 or x.shape[1] != self.ddconfig["in_channels"]:
            raise ValueError("mel must be (batch, %d, T, F), got %s" % (self.ddconfig["in_channels"], tuple(x.shape)))
        if not x.is_cuda:
            raise N.CttaError("mel is on %s: the HIP engine has no CPU path" % x.device)
        B, _, T, F = x.shape
        h = self._ensure_enc(B, T, F)
        s = 1 << (len(self.ddconfig["ch_mult"]) - 1)
        mel = x.detach().to(device=self.device, dtype=torch.float32).contiguous()
        moments = torch.empty((B, 2 * self.embed_dim, T // s, F // s), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            N.check(N.lib().ctta_vae_encode(h, N.ptr(mel), B, N.ptr(moments), N.stream_ptr()))
        return DiagonalGaussianDistribution(moments)

    encode_first_stage = encode

    def get_first_stage_encoding(self, encoder_posterior):
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            z = encoder_posterior.sample()
        elif isinstance(encoder_posterior, torch.Tensor):
            z = encoder_posterior
        else:
            raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")
        return self.scale_factor * z

    def read_encoder_taps(self):
        L_ = N.lib()
        out = OrderedDict()
        for i in range(L_.ctta_vae_encoder_num_taps(self._h_enc)):
            name = N.c_char_p()
            dims = (N.c_int * 4)()
            N.check(L_.ctta_vae_encoder_tap_info(self._h_enc, i, name, dims))
            t = torch.empty(tuple(dims), dtype=torch.float32, device=self.device)
            N.check(L_.ctta_vae_encoder_tap_read(self._h_enc, i, N.ptr(t), N.stream_ptr()))
            out[name.value.decode()] = t
        return out

    def _release(self):
        eng = getattr(self, "_eng", None) or {"vae": [], "voc": []}
        live = [e for v in eng.values() for e in v if e]
        L_ = N.lib() if (live or getattr(self, "_h_enc", None)) else None
        if getattr(self, "_h_enc", None):
            L_.ctta_vae_encoder_destroy(self._h_enc)
            self._h_enc = None
        for which, destroy in (("vae", "ctta_vae_destroy"), ("voc", "ctta_hifigan_destroy")):
            for i, e in enumerate(eng[which]):
                if e:
                    getattr(L_, destroy)(e["h"])
                    eng[which][i] = None
        self._h_vae = self._h_voc = None
        self._vae_pending = self._voc_pending = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _decoder_version(self):
        return sum(p._version for k, p in self.named_parameters() if k.startswith(("decoder.", "post_quant_conv.")))

    def _ensure_vae(self, B, T, F, grad=False):
        L_ = N.lib()
        sf = float(self.scale_factor)
        key = (T, F, sf, self.debug_taps, self.device)
        ver = self._decoder_version()
        e = self._eng["vae"][int(grad)]
        if e is None or e["key"] != key or B > e["B"] or ver != e["ver"]:
            if e is not None:
                L_.ctta_vae_destroy(e["h"])
                self._eng["vae"][int(grad)] = None
            dd = self.ddconfig
            c = N.VAEConfig()
            c.z_channels, c.embed_dim, c.ch, c.out_ch = dd["z_channels"], self.embed_dim, dd["ch"], dd["out_ch"]
            c.n_levels, c.num_res_blocks = len(dd["ch_mult"]), dd["num_res_blocks"]
            for i, m in enumerate(dd["ch_mult"]):
                c.ch_mult[i] = m
            c.scale_factor = sf
            c.max_batch, c.latent_h, c.latent_w = B, T, F
            c.debug_taps = int(self.debug_taps)
            c.enable_grad = int(grad)
            if grad:
                self._vae_pending = None
            sd = OrderedDict((k, p.detach()) for k, p in self.named_parameters()
                             if k.startswith(("decoder.", "post_quant_conv.")))
            for k, p in sd.items():
                if not p.is_cuda:
                    raise N.CttaError("parameter '%s' is on %s: the HIP engine has no CPU path" % (k, p.device))
            table, keep = N.tensor_table(sd)
            h = N.c_void_p()
            with torch.cuda.device(self.device):
                N.check(L_.ctta_vae_create(c, table, len(table), N.stream_ptr(), h))
            e = self._eng["vae"][int(grad)] = {"h": h, "key": key, "B": B, "ver": ver}
        self._h_vae = e["h"]
        return e["h"]

    def _ensure_voc(self, B, frames, grad=False):
        L_ = N.lib()
        ver = sum(p._version for p in self.vocoder.parameters())
        key = (self.debug_taps, self.device)
        e = self._eng["voc"][int(grad)]
        if e is None or e["key"] != key or B > e["B"] or frames > e["frames"] or ver != e["ver"]:
            if e is not None:
                L_.ctta_hifigan_destroy(e["h"])
                self._eng["voc"][int(grad)] = None
            h_ = self.vocoder.h
            c = N.HifiganConfig()
            c.num_mels, c.upsample_initial_channel = h_["num_mels"], h_["upsample_initial_channel"]
            c.n_ups, c.n_kernels = len(h_["upsample_rates"]), len(h_["resblock_kernel_sizes"])
            for i, (u, k) in enumerate(zip(h_["upsample_rates"], h_["upsample_kernel_sizes"])):
                c.upsample_rates[i], c.upsample_kernel_sizes[i] = u, k
            for j, (k, dil) in enumerate(zip(h_["resblock_kernel_sizes"], h_["resblock_dilation_sizes"])):
                c.resblock_kernel_sizes[j] = k
                for m in range(3):
                    c.resblock_dilations[j][m] = dil[m]
            c.max_batch, c.max_frames, c.debug_taps = B, frames, int(self.debug_taps)
            c.enable_grad = int(grad)
            if grad:
                self._voc_pending = None
            sd = OrderedDict(("vocoder." + k, p.detach()) for k, p in self.vocoder.named_parameters())
            for k, p in sd.items():
                if not p.is_cuda:
                    raise N.CttaError("parameter '%s' is on %s: the HIP engine has no CPU path" % (k, p.device))
            table, keep = N.tensor_table(sd)
            h = N.c_void_p()
            with torch.cuda.device(self.device):
                N.check(L_.ctta_hifigan_create(c, table, len(table), N.stream_ptr(), h))
            e = self._eng["voc"][int(grad)] = {"h": h, "key": key, "B": B, "frames": frames, "ver": ver}
        self._h_voc = e["h"]
        return e["h"]

    # ---- reference API
    def decode(self, z, use_ema=False):
        return self.decode_first_stage(z * self.scale_factor, use_ema=use_ema)

    def decode_first_stage(self, z, allow_grad=False, use_ema=False):
        """z (B,8,T,F) -> mel (B,1,4T,4F); z is divided by scale_factor first (autoencoder.py:105).
        allow_grad=True (CLAPLoss, tools/losses.py:294-296) returns a mel that back-propagates into z through the
        frozen decoder (ctta_vae_decode_with_grad / ctta_vae_decode_backward)."""
        if use_ema and self.ema_decoder is None:
            print("VAE does not have EMA modules, but specified use_ema. Using the none-EMA modules instead.")
        if z.ndim != 4 or z.shape[1] != self.embed_dim:
            raise ValueError("z must be (batch, %d, T, F), got %s" % (self.embed_dim, tuple(z.shape)))
        if not z.is_cuda:
            raise N.CttaError("z is on %s: the HIP engine has no CPU path" % z.device)
        if allow_grad and torch.is_grad_enabled() and z.requires_grad:
            return _VaeDecodeWithGrad.apply(z, self)
        B, _, T, F = z.shape
        h = self._ensure_vae(B, T, F)
        zz = z.detach().to(device=self.device, dtype=torch.float32).contiguous()
        up = 2 ** (len(self.ddconfig["ch_mult"]) - 1)
        mel = torch.empty((B, self.ddconfig["out_ch"], T * up, F * up), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            N.check(N.lib().ctta_vae_decode(h, N.ptr(zz), B, N.ptr(mel), N.stream_ptr()))
        return mel

    def vocode(self, mel):
        """mel (B,1,T,F) -> float waveform (B, L) before centring (Generator.forward)."""
        if mel.ndim != 4 or mel.shape[1] != 1 or mel.shape[3] != self.vocoder.h["num_mels"]:
            raise ValueError("mel must be (batch, 1, T, %d), got %s" % (self.vocoder.h["num_mels"], tuple(mel.shape)))
        if not mel.is_cuda:
            raise N.CttaError("mel is on %s: the HIP engine has no CPU path" % mel.device)
        B, _, T, F = mel.shape
        h = self._ensure_voc(B, T)
        m = mel.detach().to(device=self.device, dtype=torch.float32).contiguous()
        n = N.lib().ctta_hifigan_out_len(h, T)
        wav = torch.empty((B, n), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            N.check(N.lib().ctta_hifigan_forward(h, N.ptr(m), B, T, N.ptr(wav), N.stream_ptr()))
        return wav

    def decode_to_waveform(self, dec, allow_grad=False, return_float=False, world_extrema=False):
        """vocoder_infer (hifigan/utilities.py:76-91): batch-global (max+min)/2 centring, then
        int16 numpy on the host (the reference's return type); return_float keeps the centred
        float tensor on the device.  allow_grad=True returns the centred float waveform with a graph back to `dec`
        (utilities.py:79-81; the centring's max/min run as torch ops on the vocoder's differentiable output).
        world_extrema=True (clip-sharded generation under a process group): the extrema are MAX-reduced over all ranks
        first, so every rank centres with the whole batch's pair and the shards equal a single-process run of the
        unsharded batch bit for bit; the default centres per shard (no data-path collective)."""
        if allow_grad:
            need_graph = torch.is_grad_enabled() and dec.requires_grad
            wavs = (_VocodeWithGrad.apply(dec, self) if need_graph else self.vocode(dec)).float()
            return wavs - (wavs.max() + wavs.min()) / 2
        wav = self.vocode(dec)
        scratch = torch.empty(4, dtype=torch.float32, device=wav.device)
        centred = torch.empty_like(wav) if return_float else None
        pcm = None if return_float else torch.empty(wav.shape, dtype=torch.int16, device=wav.device)
        with torch.cuda.device(self.device):
            if world_extrema:
                from . import dist_util
                mm = torch.empty(2, dtype=torch.float32, device=wav.device)
                N.check(N.lib().ctta_wav_extrema(N.ptr(wav), wav.numel(), N.ptr(scratch), N.ptr(mm), N.stream_ptr()))
                dist_util.global_wav_extrema_(mm)
                N.check(N.lib().ctta_wav_center(N.ptr(wav), wav.numel(), N.ptr(mm), N.ptr(scratch), N.ptr(centred), N.ptr(pcm),
                                                N.stream_ptr()))
            else:
                N.check(N.lib().ctta_wav_finalize(N.ptr(wav), wav.numel(), N.ptr(scratch), N.ptr(centred), N.ptr(pcm),
                                                  N.stream_ptr()))
        return centred if return_float else pcm.cpu().numpy()

    def _read_taps(self, which):
        L_ = N.lib()
        h = self._h_vae if which == "vae" else self._h_voc
        num = getattr(L_, "ctta_%s_num_taps" % ("vae" if which == "vae" else "hifigan"))
        info = getattr(L_, "ctta_%s_tap_info" % ("vae" if which == "vae" else "hifigan"))
        read = getattr(L_, "ctta_%s_tap_read" % ("vae" if which == "vae" else "hifigan"))
        out = OrderedDict()
        for i in range(num(h)):
            name = N.c_char_p()
            dims = (N.c_int * 4)()
            N.check(info(h, i, name, dims))
            t = torch.empty(tuple(dims), dtype=torch.float32, device=self.device)
            N.check(read(h, i, N.ptr(t), N.stream_ptr()))
            out[name.value.decode()] = t
        return out
