"""HIP text encoder: the mirror of `transformers.T5EncoderModel` the reference builds with
`T5EncoderModel.from_pretrained("google/flan-t5-large")` (models/audio_distilled_model.py:97-98) and calls as
`self.text_encoder(input_ids=input_ids, attention_mask=attention_mask)[0]` (:208-214, 236-240).

Same constructor config (a T5Config-like dict), same `state_dict()` keys and order (`shared.weight`, the tied
`encoder.embed_tokens.weight`, `encoder.block.N...`), same call signature and output indexing; the arithmetic runs in
`ctta_t5_encode` (csrc/engine_t5.hip).  Frozen by construction: the reference never trains it
(`freeze_text_encoder=True`, train.sh).  The tokenizer stays `transformers.AutoTokenizer` (host-side string work)."""
from collections import OrderedDict

import torch

from . import _native as N
from . import spec
from .modules import _ParamTree


class BaseModelOutput(tuple):
    """`out[0]` / `out.last_hidden_state`, like transformers' ModelOutput for the one field the reference reads."""

    def __new__(cls, last_hidden_state):
        return super().__new__(cls, (last_hidden_state,))

    @property
    def last_hidden_state(self):
        return self[0]


class _Config(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)


class T5EncoderModel(_ParamTree):
    def __init__(self, config=None):
        super().__init__()
        cfg = dict(spec.T5_LARGE_CONFIG)
        if config is not None:
            cfg.update(config if isinstance(config, dict) else config.to_dict())
        if cfg.get("feed_forward_proj", "gated-gelu") != "gated-gelu":
            raise ValueError("only the gated-gelu feed-forward of T5 v1.1 / FLAN-T5 is built, got %r" % cfg["feed_forward_proj"])
        if cfg["d_kv"] != 64:
            raise ValueError("d_kv=%d: every released T5 uses 64 and the attention kernel is built for it" % cfg["d_kv"])
        self.config = _Config(cfg)
        full = spec.t5_encoder_param_spec(cfg)
        self._register(OrderedDict([("shared.weight", full["shared.weight"])]))
        self.add_module("encoder", torch.nn.Module())
        self.encoder.add_module("embed_tokens", torch.nn.Module())
        self.encoder.embed_tokens.weight = self.shared.weight          # tied, as in T5EncoderModel
        self._register(OrderedDict((k, s) for k, s in full.items() if k not in ("shared.weight", "encoder.embed_tokens.weight")))
        self.requires_grad_(False)
        self._h_t5 = self._h_t5_key = self._h_t5_ver = None

    @classmethod
    def from_pretrained(cls, name, **kwargs):
        """Loads a FLAN-T5 checkpoint through transformers when one is reachable (cache or network), keeping this
        class's HIP forward; raises the underlying error otherwise (there is no checkpoint offline)."""
        from transformers import T5EncoderModel as HFT5
        hf = HFT5.from_pretrained(name, **kwargs)
        m = cls(hf.config.to_dict())
        m.load_state_dict(hf.state_dict())
        return m

    @property
    def device(self):
        return self.shared.weight.device

    def _release(self):
        if getattr(self, "_h_t5", None):
            N.lib().ctta_t5_destroy(self._h_t5)
        self._h_t5 = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure(self, B, L):
        ver = self._weights_version()
        key = (self.device,)
        if (self._h_t5 is None or self._h_t5_key[2:] != key or B > self._h_t5_key[0] or L > self._h_t5_key[1]
                or ver != self._h_t5_ver):
            self._release()
            cfg = self.config
            c = N.T5Config()
            c.vocab_size, c.d_model, c.d_kv, c.d_ff = cfg["vocab_size"], cfg["d_model"], cfg["d_kv"], cfg["d_ff"]
            c.num_layers, c.num_heads = cfg["num_layers"], cfg["num_heads"]
            c.rel_buckets, c.rel_max_distance = cfg["relative_attention_num_buckets"], cfg["relative_attention_max_distance"]
            c.eps = cfg["layer_norm_epsilon"]
            c.max_batch, c.max_len = B, max(L, 8)
            table, keep = N.tensor_table(self._table())
            h = N.c_void_p()
            with torch.cuda.device(self.device):
                N.check(N.lib().ctta_t5_create(c, table, len(table), N.stream_ptr(), h))
            self._h_t5, self._h_t5_key, self._h_t5_ver = h, (B, max(L, 8)) + key, ver
        return self._h_t5

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, **kwargs):
        if kwargs.get("inputs_embeds") is not None or kwargs.get("head_mask") is not None:
            raise NotImplementedError("inputs_embeds / head_mask are not used by the reference and not built")
        if input_ids is None or input_ids.ndim != 2:
            raise ValueError("input_ids must be (batch, length)")
        if not input_ids.is_cuda:
            raise N.CttaError("input_ids is on %s: the HIP engine has no CPU path" % input_ids.device)
        B, L = input_ids.shape
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if tuple(attention_mask.shape) != (B, L):
            raise ValueError("attention_mask must match input_ids, got %s" % (tuple(attention_mask.shape),))
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        if int(ids.min()) < 0 or int(ids.max()) >= self.config["vocab_size"]:
            raise IndexError("input_ids outside [0, %d)" % self.config["vocab_size"])
        mask = (attention_mask != 0).to(device=self.device, dtype=torch.uint8).contiguous()
        if not bool(mask.any(dim=1).all()):
            raise ValueError("every row of attention_mask needs at least one token")
        h = self._ensure(B, L)
        out = torch.empty((B, L, self.config["d_model"]), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            N.check(N.lib().ctta_t5_encode(h, N.ptr(ids), N.ptr(mask), B, L, N.ptr(out), N.stream_ptr()))
        return BaseModelOutput(out)
