"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

fp32 restatement of the FLAN-T5 encoder the reference calls at models/audio_distilled_model.py:208-214
(`self.text_encoder(input_ids=..., attention_mask=...)[0]`).  The arithmetic lives in a third-party dependency that is
not vendored under /root/reference: transformers==4.29.2 (environment.yml:157), `T5EncoderModel` / `T5Stack` /
`T5Attention` / `T5LayerNorm` / `T5DenseGatedActDense` of models/t5/modeling_t5.py.  The published algorithm, restated:

  h = embedding(ids)
  per block:  n = rmsnorm(h);  q,k,v = n Wq^T, n Wk^T, n Wv^T (heads x d_kv, NO 1/sqrt(d) scaling)
              scores = q k^T + position_bias + mask,   position_bias[h][i][j] = rel_emb[bucket(j - i)][h]
              (bidirectional log-spaced buckets, shared by all blocks; block 0 owns the table),
              mask = (1 - attention_mask) * finfo.min;   h += softmax(scores) v Wo^T
              n = rmsnorm(h);  h += (gelu_new(n Wi0^T) * (n Wi1^T)) Wo^T
  out = rmsnorm(h)                      rmsnorm(x) = x * rsqrt(mean(x^2) + eps) * weight   (no mean subtraction, no bias)

Pinned against the installed transformers' own T5EncoderModel with the build's deterministic weights
(tests/golden/make_golden_t5.py -> tests/golden/t5_encoder.npz)."""
import math

import torch
import torch.nn.functional as F


def relative_position_bucket(relative_position, num_buckets=32, max_distance=128):
    """Bidirectional bucket of (memory position - query position): half the buckets per sign; the first half of each
    exact, the rest log-spaced up to max_distance."""
    num_buckets //= 2
    ret = (relative_position > 0).to(torch.long) * num_buckets
    n = relative_position.abs()
    max_exact = num_buckets // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return ret + torch.where(n < max_exact, n, large)


def rmsnorm(x, weight, eps):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * weight


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def t5_encode(cfg, sd, input_ids, attention_mask, taps=None):
    H, dk, eps = cfg["num_heads"], cfg["d_kv"], cfg["layer_norm_epsilon"]
    B, L = input_ids.shape
    h = sd["shared.weight"][input_ids]
    pos = torch.arange(L)
    bucket = relative_position_bucket(pos[None, :] - pos[:, None], cfg["relative_attention_num_buckets"],
                                      cfg["relative_attention_max_distance"])
    bias = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"][bucket].permute(2, 0, 1)[None]
    bias = bias + (1.0 - attention_mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    for i in range(cfg["num_layers"]):
        p = "encoder.block.%d.layer." % i
        n = rmsnorm(h, sd[p + "0.layer_norm.weight"], eps)
        q, k, v = (F.linear(n, sd[p + "0.SelfAttention.%s.weight" % t]).view(B, L, H, dk).transpose(1, 2) for t in "qkv")
        a = torch.softmax(q @ k.transpose(2, 3) + bias, dim=-1) @ v
        h = h + F.linear(a.transpose(1, 2).reshape(B, L, H * dk), sd[p + "0.SelfAttention.o.weight"])
        n = rmsnorm(h, sd[p + "1.layer_norm.weight"], eps)
        g = gelu_new(F.linear(n, sd[p + "1.DenseReluDense.wi_0.weight"])) * F.linear(n, sd[p + "1.DenseReluDense.wi_1.weight"])
        h = h + F.linear(g, sd[p + "1.DenseReluDense.wo.weight"])
        if taps is not None:
            taps["block.%d" % i] = h.detach().clone()
    return rmsnorm(h, sd["encoder.final_layer_norm.weight"], eps)
