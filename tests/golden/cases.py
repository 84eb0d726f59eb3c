"""Shared definitions of the parity cases: configs, deterministic inputs, weights.

Inputs and weights are pure functions of (name, shape, seed) via
`consistencytta_amd.spec.det_uniform/det_weight`, so fixtures only need to hold the
reference's OUTPUTS.  Used by make_golden.py (reference side), the oracle tests and the
GPU parity tests.
"""
import numpy as np
import torch

from consistencytta_amd import spec

# Tiny U-Net: exercises inner_dim != channels (heads that do not divide C: 40//3=13 -> 39),
# GroupNorm group sizes that are not multiples of 8 (5, 10, 15), concat widths, 4 levels.
TINY_UNET = dict(spec.LIGHT_UNET_CONFIG,
                 block_out_channels=[40, 80, 80, 80],
                 attention_head_dim=[3, 3, 6, 6],
                 cross_attention_dim=48,
                 norm_num_groups=8)

TINY_VAE_DD = dict(spec.VAE_DDCONFIG, ch=32)
TINY_VAE_GROUPS = 32  # Normalize() hard-codes 32 groups (modules.py:38-41): ch=32 -> 1 ch/group

TINY_HIFIGAN = dict(spec.HIFIGAN_16K_64, upsample_initial_channel=256)  # 128,64,32,16,8 channels

SIGMA_MAX = 14.6146


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def unet_weights(cfg, guided=True, seed=0):
    sp = spec.unet_param_spec(cfg, guided)
    return {k: t(v) for k, v in spec.det_state_dict(sp, seed, prefix="unet.").items()}


def vae_weights(dd, seed=0):
    sp = spec.vae_decoder_param_spec(dd)
    return {k: t(v) for k, v in spec.det_state_dict(sp, seed, prefix="vae.").items()}


def vae_encoder_weights(dd, seed=0):
    sp = spec.vae_encoder_param_spec(dd)
    return {k: t(v) for k, v in spec.det_state_dict(sp, seed, prefix="vae.").items()}


def hifigan_weights(h, seed=0):
    sp = spec.hifigan_param_spec(h)
    return {k: t(v) for k, v in spec.det_state_dict(sp, seed, prefix="vae.").items()}


def unet_inputs(cfg, B, H, W, L, tag, per_sample=True):
    """sample, timestep, guidance, encoder states, mask for a case called `tag`."""
    C = cfg["in_channels"]
    X = cfg["cross_attention_dim"]
    sample = t(spec.det_uniform(tag + ".sample", (B, C, H, W), 1)) * 1.7
    enc = t(spec.det_uniform(tag + ".enc", (B, L, X), 2)) * 0.5
    lens = (np.abs(spec.det_uniform(tag + ".len", (B,), 3)) * (L - 1)).astype(np.int64) + 1
    lens[0] = L
    mask = torch.arange(L)[None, :] < t(lens)[:, None]
    if per_sample:
        ts = t(np.abs(spec.det_uniform(tag + ".t", (B,), 4)).astype(np.float64) * 999.0)
        gs = t(np.abs(spec.det_uniform(tag + ".w", (B,), 5)) * 6.0)
    else:
        ts, gs = 999.0, 4.0
    return sample, ts, gs, enc, mask


def vae_inputs(B, T, Fq, tag):
    return t(spec.det_uniform(tag + ".z", (B, 8, T, Fq), 6)) * 2.0


def mel_inputs(B, T, nmel, tag):
    return t(spec.det_uniform(tag + ".mel", (B, 1, T, nmel), 7)) * 1.5


def prompt_states(cfg, B, L, tag):
    """cond / uncond text states + masks (what encode_text_classifier_free would return)."""
    X = cfg["cross_attention_dim"]
    cond = t(spec.det_uniform(tag + ".cond", (B, L, X), 11)) * 0.5
    uncond = t(spec.det_uniform(tag + ".uncond", (B, L, X), 12)) * 0.5
    lens = (np.abs(spec.det_uniform(tag + ".len", (B,), 13)) * (L - 1)).astype(np.int64) + 1
    lens[0] = L
    cmask = torch.arange(L)[None, :] < t(lens)[:, None]
    umask = torch.zeros(B, L, dtype=torch.bool)
    umask[:, 0] = True        # "" pads to the cond length: one valid token (audio_distilled_model.py:230-233)
    return dict(embeds_cf=torch.cat([uncond, cond]), mask_cf=torch.cat([umask, cmask]), embeds=cond, mask=cmask)


def sample_index(numel, n=512):
    """Deterministic strided sample positions inside a flattened tensor: what `distill_light.npz` keeps of every
    gradient tensor (the full gradient is 2.2 GB)."""
    n = min(n, numel)
    return (np.arange(n, dtype=np.int64) * numel) // n


# Text encoder: a small T5 (d_kv stays 64 like every T5 size) and FLAN-T5-large's widths at 2 layers.
TINY_T5 = dict(spec.T5_LARGE_CONFIG, vocab_size=512, d_model=256, d_ff=512, num_layers=3, num_heads=4)
WIDE_T5 = dict(spec.T5_LARGE_CONFIG, vocab_size=1024, num_layers=2)


def t5_weights(cfg, seed=0):
    sp = spec.t5_encoder_param_spec(cfg)
    sd = {k: t(v) for k, v in spec.det_state_dict(sp, seed, prefix="t5.").items() if k != "encoder.embed_tokens.weight"}
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    return sd


def t5_inputs(cfg, B, L, tag):
    ids = (np.abs(spec.det_uniform(tag + ".ids", (B, L), 31)) * (cfg["vocab_size"] - 1)).astype(np.int64)
    lens = (np.abs(spec.det_uniform(tag + ".len", (B,), 32)) * (L - 1)).astype(np.int64) + 1
    lens[0] = L
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    return t(ids), t(mask)


# CLAP text tower: a small RoBERTa and roberta-base's widths at one layer (head width 64 like every released size)
TINY_ROBERTA = dict(spec.ROBERTA_BASE_CONFIG, vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                    intermediate_size=256, max_position_embeddings=90)
WIDE_ROBERTA = dict(spec.ROBERTA_BASE_CONFIG, vocab_size=2000, num_hidden_layers=1, max_position_embeddings=90)


def roberta_weights(cfg, seed=0):
    sd = {}
    for k, s in spec.roberta_param_spec(cfg).items():
        if k.endswith("LayerNorm.weight"):
            sd[k] = t(1.0 + 0.2 * spec.det_uniform("roberta." + k, s, seed))
        elif "embeddings" in k and k.endswith("weight"):
            sd[k] = t(spec.det_uniform("roberta." + k, s, seed))          # O(1) embedding tables
        else:
            sd[k] = t(spec.clap_det_weight("roberta." + k, s, seed))
    return sd


def roberta_inputs(cfg, B, L, tag):
    ids = (np.abs(spec.det_uniform(tag + ".rids", (B, L), 41)) * (cfg["vocab_size"] - 3)).astype(np.int64) + 2
    lens = (np.abs(spec.det_uniform(tag + ".rlen", (B,), 42)) * (L - 2)).astype(np.int64) + 2
    lens[0] = L
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    ids = np.where(mask == 1, ids, cfg["pad_token_id"])
    return t(ids), t(mask)


def clap_weights(keys, shapes, tag, seed):
    """HTSAT weights of the `clap_htsat.npz` cases: the reference module's own key / shape lists (fixture data) through
    the deterministic generator."""
    sd = {}
    for k, s in zip(keys, shapes):
        shape = tuple(int(x) for x in str(s).split(",")) if str(s) else ()
        sd[str(k)] = t(spec.clap_det_weight(tag + "." + str(k), shape, seed))
    return sd


TINY_HTSAT = dict(spec.HTSAT_BASE_CONFIG, spec_size=64, embed_dim=64, depths=[2, 2, 2, 2], num_heads=[2, 4, 8, 16],
                  num_classes=11)


# ------------------------------------------------------------------------------------------------ evaluation suite
def cnn14_weights(keys, shapes, seed=3):
    """Deterministic Cnn14 weights over the reference module's own state-dict keys (stored in eval_suite.npz as data)."""
    return {str(k): t(spec.cnn14_det_weight("cnn14." + str(k), tuple(int(d) for d in str(s).split(",")), seed))
            for k, s in zip(keys, shapes)}


def eval_waves(tag, B, L, sr=16000):
    """Seeded test audio: band-limited noise plus two sines per clip, amplitude about 0.3."""
    t = np.arange(L, dtype=np.float64) / sr
    out = []
    for b in range(B):
        n = spec.det_uniform("%s.noise.%d" % (tag, b), (L,), 11).astype(np.float64)
        n = np.convolve(n, np.ones(8) / 8.0, mode="same")
        w = 0.25 * n + 0.15 * np.sin(2 * np.pi * (220.0 * (b + 1)) * t) + 0.1 * np.sin(2 * np.pi * (1330.0 + 517.0 * b) * t + 0.3)
        out.append(w)
    return torch.from_numpy(np.stack(out).astype(np.float32))


def eval_metric_inputs():
    """Seeded feature matrices, the same on both sides (tests rebuild them from spec.det_uniform)."""
    u = lambda name, shape: spec.det_uniform("evalsuite." + name, shape, 21).astype(np.float64)
    mix = u("fid.mix", (48, 48)) * 0.3 + np.eye(48)
    fid1 = (u("fid.1", (200, 48)) * 1.7) @ mix + 0.2
    fid2 = (u("fid.2", (180, 48)) * 1.4) @ mix.T - 0.1
    isc = u("isc", (203, 527)) * 4.0
    kid1 = np.maximum(u("kid.1", (120, 64)) * 2.0 + 0.5, 0.0)
    kid2 = np.maximum(u("kid.2", (100, 64)) * 1.5 + 0.8, 0.0)
    kl1, kl2 = u("kl.1", (50, 527)) * 3.0, u("kl.2", (50, 527)) * 3.0
    f32 = lambda a: torch.from_numpy(a.astype(np.float32))
    return dict(fid1=f32(fid1), fid2=f32(fid2), isc=f32(isc), kid1=f32(kid1), kid2=f32(kid2), kl1=f32(kl1), kl2=f32(kl2))


def eval_kl_names(n):
    names = ["clip_%03d.wav" % i for i in range(n)]
    perm = np.random.RandomState(5).permutation(n)
    return names, perm
