"""Parameter tables (name -> shape) of the three networks on the hot path, in the
reference's own state-dict order, plus a deterministic, library-version-independent
weight generator used for random-init benchmarks and parity fixtures.

Key names follow the reference so that its checkpoints load unchanged:
  * U-Net: diffusers/models/unet_2d_condition_guided.py:256-567 (module registration order),
    unet_2d_blocks.py:825-906,503-586,1934-2015,2081-2127, resnet.py:418-547,
    transformer_2d.py:145-216, attention.py:226-274,
  * VAE decoder: audioldm/variational_autoencoder/modules.py:546-648, autoencoder.py:36-37,
  * HiFi-GAN: audioldm/hifigan/models.py:72-99 (weight_norm removed, utilities.py:71).
"""
from collections import OrderedDict

import numpy as np

LIGHT_UNET_CONFIG = {
    "act_fn": "silu",
    "attention_head_dim": [5, 10, 20, 20],
    "block_out_channels": [256, 512, 1024, 1024],
    "center_input_sample": False,
    "cross_attention_dim": 1024,
    "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D",
                         "CrossAttnDownBlock2D", "DownBlock2D"],
    "downsample_padding": 1,
    "flip_sin_to_cos": True,
    "freq_shift": 0,
    "in_channels": 8,
    "layers_per_block": 2,
    "mid_block_scale_factor": 1,
    "norm_eps": 1e-5,
    "norm_num_groups": 32,
    "out_channels": 8,
    "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D",
                       "CrossAttnUpBlock2D"],
    "use_linear_projection": True,
    "upcast_attention": True,
}

FULL_UNET_CONFIG = dict(LIGHT_UNET_CONFIG, block_out_channels=[320, 640, 1280, 1280])

# audioldm/utils.py:75-88 (first_stage_config.params.ddconfig) + embed_dim
VAE_DDCONFIG = {
    "double_z": True, "z_channels": 8, "resolution": 256, "in_channels": 1, "out_ch": 1,
    "ch": 128, "ch_mult": [1, 2, 4], "num_res_blocks": 2, "attn_resolutions": [],
    "dropout": 0.0,
}

# audioldm/hifigan/utilities.py:9-39
HIFIGAN_16K_64 = {
    "upsample_rates": [5, 4, 2, 2, 2],
    "upsample_kernel_sizes": [16, 16, 8, 4, 4],
    "upsample_initial_channel": 1024,
    "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    "num_mels": 64,
}


def _as_list(v, n):
    return list(v) if isinstance(v, (list, tuple)) else [v] * n


def unet_levels(cfg):
    """Per-level derived quantities shared by the table below, the oracle and the engine."""
    boc = list(cfg["block_out_channels"])
    n = len(boc)
    heads = _as_list(cfg.get("num_attention_heads") or cfg["attention_head_dim"], n)
    layers = _as_list(cfg["layers_per_block"], n)
    return boc, heads, layers


def _resnet(sd, p, cin, cout, temb):
    sd[p + "norm1.weight"] = (cin,)
    sd[p + "norm1.bias"] = (cin,)
    sd[p + "conv1.weight"] = (cout, cin, 3, 3)
    sd[p + "conv1.bias"] = (cout,)
    sd[p + "time_emb_proj.weight"] = (cout, temb)
    sd[p + "time_emb_proj.bias"] = (cout,)
    sd[p + "norm2.weight"] = (cout,)
    sd[p + "norm2.bias"] = (cout,)
    sd[p + "conv2.weight"] = (cout, cout, 3, 3)
    sd[p + "conv2.bias"] = (cout,)
    if cin != cout:
        sd[p + "conv_shortcut.weight"] = (cout, cin, 1, 1)
        sd[p + "conv_shortcut.bias"] = (cout,)


def _transformer(sd, p, c, heads, xdim):
    inner = heads * (c // heads)  # unet_2d_blocks.py:873-875: dim_head = out_channels // heads
    sd[p + "norm.weight"] = (c,)
    sd[p + "norm.bias"] = (c,)
    sd[p + "proj_in.weight"] = (inner, c)
    sd[p + "proj_in.bias"] = (inner,)
    t = p + "transformer_blocks.0."
    sd[t + "attn1.to_q.weight"] = (inner, inner)
    sd[t + "attn1.to_k.weight"] = (inner, inner)
    sd[t + "attn1.to_v.weight"] = (inner, inner)
    sd[t + "attn1.to_out.0.weight"] = (inner, inner)
    sd[t + "attn1.to_out.0.bias"] = (inner,)
    sd[t + "ff.net.0.proj.weight"] = (inner * 8, inner)
    sd[t + "ff.net.0.proj.bias"] = (inner * 8,)
    sd[t + "ff.net.2.weight"] = (inner, inner * 4)
    sd[t + "ff.net.2.bias"] = (inner,)
    sd[t + "attn2.to_q.weight"] = (inner, inner)
    sd[t + "attn2.to_k.weight"] = (inner, xdim)
    sd[t + "attn2.to_v.weight"] = (inner, xdim)
    sd[t + "attn2.to_out.0.weight"] = (inner, inner)
    sd[t + "attn2.to_out.0.bias"] = (inner,)
    for k in ("norm1", "norm2", "norm3"):
        sd[t + k + ".weight"] = (inner,)
        sd[t + k + ".bias"] = (inner,)
    sd[p + "proj_out.weight"] = (c, inner)
    sd[p + "proj_out.bias"] = (c,)


def unet_param_spec(cfg, guided=True):
    """OrderedDict name -> shape for UNet2DConditionGuidedModel (guided=True) or
    UNet2DConditionModel (teacher, guided=False)."""
    boc, heads, layers = unet_levels(cfg)
    n = len(boc)
    temb = boc[0] * 4
    xdim = cfg["cross_attention_dim"]
    down_types = cfg["down_block_types"]
    up_types = cfg["up_block_types"]
    sd = OrderedDict()
    sd["conv_in.weight"] = (boc[0], cfg["in_channels"], 3, 3)
    sd["conv_in.bias"] = (boc[0],)
    if guided:
        sd["guidance_proj.weight"] = (temb // 2,)
    sd["time_embedding.linear_1.weight"] = (temb, boc[0])
    sd["time_embedding.linear_1.bias"] = (temb,)
    sd["time_embedding.linear_2.weight"] = (temb, temb)
    sd["time_embedding.linear_2.bias"] = (temb,)
    if guided:
        sd["guidance_embedding.linear_1.weight"] = (temb, temb)
        sd["guidance_embedding.linear_1.bias"] = (temb,)
        sd["guidance_embedding.linear_2.weight"] = (temb, temb)
        sd["guidance_embedding.linear_2.bias"] = (temb,)
    out_c = boc[0]
    for i in range(n):
        in_c, out_c = out_c, boc[i]
        p = "down_blocks.%d." % i
        if down_types[i] == "CrossAttnDownBlock2D":
            for j in range(layers[i]):
                _transformer(sd, p + "attentions.%d." % j, out_c, heads[i], xdim)
        for j in range(layers[i]):
            _resnet(sd, p + "resnets.%d." % j, in_c if j == 0 else out_c, out_c, temb)
        if i != n - 1:
            sd[p + "downsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            sd[p + "downsamplers.0.conv.bias"] = (out_c,)
    rboc, rheads, rlayers = boc[::-1], heads[::-1], layers[::-1]
    out_c = rboc[0]
    for i in range(n):
        prev_c, out_c = out_c, rboc[i]
        in_c = rboc[min(i + 1, n - 1)]
        nl = rlayers[i] + 1
        p = "up_blocks.%d." % i
        if up_types[i] == "CrossAttnUpBlock2D":
            for j in range(nl):
                _transformer(sd, p + "attentions.%d." % j, out_c, rheads[i], xdim)
        for j in range(nl):
            skip_c = in_c if j == nl - 1 else out_c
            rin = prev_c if j == 0 else out_c
            _resnet(sd, p + "resnets.%d." % j, rin + skip_c, out_c, temb)
        if i != n - 1:
            sd[p + "upsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            sd[p + "upsamplers.0.conv.bias"] = (out_c,)
    _transformer(sd, "mid_block.attentions.0.", boc[-1], heads[-1], xdim)
    _resnet(sd, "mid_block.resnets.0.", boc[-1], boc[-1], temb)
    _resnet(sd, "mid_block.resnets.1.", boc[-1], boc[-1], temb)
    sd["conv_norm_out.weight"] = (boc[0],)
    sd["conv_norm_out.bias"] = (boc[0],)
    sd["conv_out.weight"] = (cfg["out_channels"], boc[0], 3, 3)
    sd["conv_out.bias"] = (cfg["out_channels"],)
    return sd


def _vae_resblock(sd, p, cin, cout):
    sd[p + "norm1.weight"] = (cin,)
    sd[p + "norm1.bias"] = (cin,)
    sd[p + "conv1.weight"] = (cout, cin, 3, 3)
    sd[p + "conv1.bias"] = (cout,)
    sd[p + "norm2.weight"] = (cout,)
    sd[p + "norm2.bias"] = (cout,)
    sd[p + "conv2.weight"] = (cout, cout, 3, 3)
    sd[p + "conv2.bias"] = (cout,)
    if cin != cout:
        sd[p + "nin_shortcut.weight"] = (cout, cin, 1, 1)
        sd[p + "nin_shortcut.bias"] = (cout,)


def vae_decoder_param_spec(dd=VAE_DDCONFIG, embed_dim=8):
    """decoder.* and post_quant_conv.* keys of AutoencoderKL (modules.py:546-648)."""
    ch, mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    nres = len(mult)
    block_in = ch * mult[-1]
    sd = OrderedDict()
    sd["decoder.conv_in.weight"] = (block_in, dd["z_channels"], 3, 3)
    sd["decoder.conv_in.bias"] = (block_in,)
    _vae_resblock(sd, "decoder.mid.block_1.", block_in, block_in)
    a = "decoder.mid.attn_1."
    sd[a + "norm.weight"] = (block_in,)
    sd[a + "norm.bias"] = (block_in,)
    for k in ("q", "k", "v", "proj_out"):
        sd[a + k + ".weight"] = (block_in, block_in, 1, 1)
        sd[a + k + ".bias"] = (block_in,)
    _vae_resblock(sd, "decoder.mid.block_2.", block_in, block_in)
    per_level = {}
    for lvl in reversed(range(nres)):
        block_out = ch * mult[lvl]
        entries = OrderedDict()
        for b in range(nrb + 1):
            _vae_resblock(entries, "decoder.up.%d.block.%d." % (lvl, b), block_in, block_out)
            block_in = block_out
        if lvl != 0:
            entries["decoder.up.%d.upsample.conv.weight" % lvl] = (block_in, block_in, 3, 3)
            entries["decoder.up.%d.upsample.conv.bias" % lvl] = (block_in,)
        per_level[lvl] = entries
    for lvl in range(nres):  # self.up.insert(0, up): state-dict order is level 0 first
        sd.update(per_level[lvl])
    sd["decoder.norm_out.weight"] = (block_in,)
    sd["decoder.norm_out.bias"] = (block_in,)
    sd["decoder.conv_out.weight"] = (dd["out_ch"], block_in, 3, 3)
    sd["decoder.conv_out.bias"] = (dd["out_ch"],)
    sd["post_quant_conv.weight"] = (dd["z_channels"], embed_dim, 1, 1)
    sd["post_quant_conv.bias"] = (dd["z_channels"],)
    return sd


def vae_encoder_param_spec(dd=VAE_DDCONFIG, embed_dim=8):
    """encoder.* and quant_conv.* keys of AutoencoderKL (modules.py:419-543, autoencoder.py:37), reference order:
    conv_in, down.{lvl}.block.{b} (+ downsample.conv), mid, norm_out, conv_out; attn_resolutions is empty in the
    AudioLDM config, so only the mid block carries attention."""
    ch, mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    nres = len(mult)
    sd = OrderedDict()
    sd["encoder.conv_in.weight"] = (ch, dd["in_channels"], 3, 3)
    sd["encoder.conv_in.bias"] = (ch,)
    block_in = ch
    for lvl in range(nres):
        block_out = ch * mult[lvl]
        for b in range(nrb):
            _vae_resblock(sd, "encoder.down.%d.block.%d." % (lvl, b), block_in, block_out)
            block_in = block_out
        if lvl != nres - 1:
            sd["encoder.down.%d.downsample.conv.weight" % lvl] = (block_in, block_in, 3, 3)
            sd["encoder.down.%d.downsample.conv.bias" % lvl] = (block_in,)
    _vae_resblock(sd, "encoder.mid.block_1.", block_in, block_in)
    a = "encoder.mid.attn_1."
    sd[a + "norm.weight"] = (block_in,)
    sd[a + "norm.bias"] = (block_in,)
    for k in ("q", "k", "v", "proj_out"):
        sd[a + k + ".weight"] = (block_in, block_in, 1, 1)
        sd[a + k + ".bias"] = (block_in,)
    _vae_resblock(sd, "encoder.mid.block_2.", block_in, block_in)
    sd["encoder.norm_out.weight"] = (block_in,)
    sd["encoder.norm_out.bias"] = (block_in,)
    zc2 = 2 * dd["z_channels"] if dd.get("double_z", True) else dd["z_channels"]
    sd["encoder.conv_out.weight"] = (zc2, block_in, 3, 3)
    sd["encoder.conv_out.bias"] = (zc2,)
    sd["quant_conv.weight"] = (2 * embed_dim, 2 * dd["z_channels"], 1, 1)
    sd["quant_conv.bias"] = (2 * embed_dim,)
    return sd


def hifigan_param_spec(h=HIFIGAN_16K_64, prefix="vocoder."):
    """Generator keys after remove_weight_norm (bias registered before weight)."""
    sd = OrderedDict()
    c0 = h["upsample_initial_channel"]
    sd[prefix + "conv_pre.bias"] = (c0,)
    sd[prefix + "conv_pre.weight"] = (c0, h["num_mels"], 7)
    nk = len(h["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        sd[prefix + "ups.%d.bias" % i] = (c0 // 2 ** (i + 1),)
        sd[prefix + "ups.%d.weight" % i] = (c0 // 2 ** i, c0 // 2 ** (i + 1), k)
    ch = c0
    for i in range(len(h["upsample_rates"])):
        ch = c0 // 2 ** (i + 1)
        for j, k in enumerate(h["resblock_kernel_sizes"]):
            p = prefix + "resblocks.%d." % (i * nk + j)
            for grp in ("convs1", "convs2"):
                for d in range(3):
                    sd[p + "%s.%d.bias" % (grp, d)] = (ch,)
                    sd[p + "%s.%d.weight" % (grp, d)] = (ch, ch, k)
    sd[prefix + "conv_post.bias"] = (1,)
    sd[prefix + "conv_post.weight"] = (1, ch, 7)
    return sd


# FLAN-T5-large encoder (google/flan-t5-large config.json; the reference's text encoder,
# models/audio_distilled_model.py:97-98).  Weights are not available offline: shapes and key order only.
T5_LARGE_CONFIG = {
    "vocab_size": 32128, "d_model": 1024, "d_kv": 64, "d_ff": 2816, "num_layers": 24, "num_heads": 16,
    "relative_attention_num_buckets": 32, "relative_attention_max_distance": 128, "layer_norm_epsilon": 1e-6,
    "feed_forward_proj": "gated-gelu", "tie_word_embeddings": False,
}


def t5_encoder_param_spec(cfg=T5_LARGE_CONFIG):
    """transformers.T5EncoderModel.state_dict() keys in order (`encoder.embed_tokens.weight` is the same tensor as
    `shared.weight`)."""
    sd = OrderedDict()
    d, inner, ff = cfg["d_model"], cfg["num_heads"] * cfg["d_kv"], cfg["d_ff"]
    sd["shared.weight"] = (cfg["vocab_size"], d)
    sd["encoder.embed_tokens.weight"] = (cfg["vocab_size"], d)
    for i in range(cfg["num_layers"]):
        p = "encoder.block.%d.layer." % i
        for n in ("q", "k", "v"):
            sd[p + "0.SelfAttention.%s.weight" % n] = (inner, d)
        sd[p + "0.SelfAttention.o.weight"] = (d, inner)
        if i == 0:
            sd[p + "0.SelfAttention.relative_attention_bias.weight"] = (cfg["relative_attention_num_buckets"], cfg["num_heads"])
        sd[p + "0.layer_norm.weight"] = (d,)
        sd[p + "1.DenseReluDense.wi_0.weight"] = (ff, d)
        sd[p + "1.DenseReluDense.wi_1.weight"] = (ff, d)
        sd[p + "1.DenseReluDense.wo.weight"] = (d, ff)
        sd[p + "1.layer_norm.weight"] = (d,)
    sd["encoder.final_layer_norm.weight"] = (d,)
    return sd


# ----------------------------------------------------------------------------------------
# Deterministic generator: value i of tensor `name` is a pure function of (seed, name, i).
# splitmix64 counter stream -> 24-bit uniform -> float32; only IEEE basic ops, so the
# result is identical under any numpy/torch version.
# ----------------------------------------------------------------------------------------
_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(s):
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def det_uniform(name, shape, seed=0):
    """float32 array in [-1, 1)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((_fnv1a(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        z = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + base
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)  # [0,1)
    return (u * np.float32(2.0) - np.float32(1.0)).reshape(shape)


def weight_rule(name, shape):
    """(offset, amplitude) of the random-init value `offset + amplitude * u`, u ~ U[-1,1), by
    name class.  Scales are chosen so that activations stay O(1) through ~100 layers (unit-gain
    fan-in init, damped residual branches); there is no checkpoint on either box (no network),
    see BASELINE.md §3."""
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if name.endswith("guidance_proj.weight"):
        return 0.0, float(np.sqrt(3.0))  # ~ unit variance
    is_norm = any(t in name for t in (".norm", "norm1.", "norm2.", "norm3.", "norm_out",
                                      "conv_norm_out", "norm."))
    if len(shape) == 1:
        if is_norm and leaf == "weight":
            return 1.0, 0.2
        return 0.0, 0.05  # biases
    if "vocoder.ups." in name:
        # ConvTranspose1d weight (Cin, Cout, k): each output sees Cin * k/u taps
        fan_in = shape[0] * max(1, shape[2] // 2)
    else:
        fan_in = int(np.prod(shape[1:]))
    gain = 1.0
    if any(t in name for t in ("conv2.", "to_out.0.", "ff.net.2.", "proj_out.", "convs2.")):
        gain = 0.5  # residual-branch outputs
    if "SelfAttention.q." in name:
        gain = 0.125  # T5 folds the missing 1/sqrt(d_kv) score scaling into q's init: std (d_model * d_kv)^-0.5
    return 0.0, float(gain * np.sqrt(3.0 / fan_in))


def det_weight(name, shape, seed=0):
    """Deterministic random-init value for a parameter (see weight_rule)."""
    off, amp = weight_rule(name, shape)
    u = det_uniform(name, tuple(shape), seed)
    return (np.float32(off) + np.float32(amp) * u).astype(np.float32)


def det_state_dict(spec, seed=0, prefix=""):
    """name -> float32 numpy array for every entry of a spec table."""
    return OrderedDict((k, det_weight(prefix + k, s, seed)) for k, s in spec.items())


# ------------------------------------------------------------------------------------------------------------------
# CLAP (CLAP fine-tuning stage, tools/losses.py:259-316): laion_clap.CLAP_Module(amodel='HTSAT-base', tmodel='roberta')
HTSAT_BASE_CONFIG = dict(spec_size=256, patch_size=4, patch_stride=4, embed_dim=128, depths=[2, 2, 12, 2],
                         num_heads=[4, 8, 16, 32], window_size=8, mlp_ratio=4.0, mel_bins=64, sample_rate=48000,
                         n_fft=1024, hop=480, fmin=50, fmax=14000, num_classes=527)
ROBERTA_BASE_CONFIG = dict(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                           intermediate_size=3072, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5,
                           pad_token_id=1)
CLAP_JOINT_DIM = 512


def htsat_param_spec(cfg=HTSAT_BASE_CONFIG):
    """State-dict keys / shapes of `HTSAT_Swin_Transformer` (laion_clap/clap_module/htsat.py:615-775) in the module's
    own registration order, floating-point entries only (the integer buffers relative_position_index /
    num_batches_tracked and the constant shift masks are structural).  torchlibrosa's frozen STFT / mel matrices are
    listed too so that a released checkpoint loads; the engine derives them itself."""
    sd = OrderedDict()
    n_fft, F = cfg["n_fft"], cfg["n_fft"] // 2 + 1
    sd["spectrogram_extractor.stft.conv_real.weight"] = (F, 1, n_fft)
    sd["spectrogram_extractor.stft.conv_imag.weight"] = (F, 1, n_fft)
    sd["logmel_extractor.melW"] = (F, cfg["mel_bins"])
    for k in ("weight", "bias", "running_mean", "running_var"):
        sd["bn0." + k] = (cfg["mel_bins"],)
    C, ps = cfg["embed_dim"], cfg["patch_size"]
    sd["patch_embed.proj.weight"] = (C, 1, ps, ps)
    sd["patch_embed.proj.bias"] = (C,)
    sd["patch_embed.norm.weight"] = (C,)
    sd["patch_embed.norm.bias"] = (C,)
    res = cfg["spec_size"] // cfg["patch_stride"]
    n_layers = len(cfg["depths"])
    for i, (depth, heads) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        dim = C * 2 ** i
        ws = min(cfg["window_size"], res)
        hidden = int(dim * cfg["mlp_ratio"])
        for j in range(depth):
            p = "layers.%d.blocks.%d." % (i, j)
            sd[p + "norm1.weight"] = (dim,)
            sd[p + "norm1.bias"] = (dim,)
            sd[p + "attn.relative_position_bias_table"] = ((2 * ws - 1) ** 2, heads)
            sd[p + "attn.qkv.weight"] = (3 * dim, dim)
            sd[p + "attn.qkv.bias"] = (3 * dim,)
            sd[p + "attn.proj.weight"] = (dim, dim)
            sd[p + "attn.proj.bias"] = (dim,)
            sd[p + "norm2.weight"] = (dim,)
            sd[p + "norm2.bias"] = (dim,)
            sd[p + "mlp.fc1.weight"] = (hidden, dim)
            sd[p + "mlp.fc1.bias"] = (hidden,)
            sd[p + "mlp.fc2.weight"] = (dim, hidden)
            sd[p + "mlp.fc2.bias"] = (dim,)
        if i < n_layers - 1:
            p = "layers.%d.downsample." % i
            sd[p + "reduction.weight"] = (2 * dim, 4 * dim)
            sd[p + "norm.weight"] = (4 * dim,)
            sd[p + "norm.bias"] = (4 * dim,)
            res //= 2
    nf = C * 2 ** (n_layers - 1)
    sd["norm.weight"] = (nf,)
    sd["norm.bias"] = (nf,)
    sf = cfg["spec_size"] // (2 ** (n_layers - 1)) // cfg["patch_stride"] // (cfg["spec_size"] // cfg["mel_bins"])
    sd["tscam_conv.weight"] = (cfg["num_classes"], nf, sf, 3)
    sd["tscam_conv.bias"] = (cfg["num_classes"],)
    sd["head.weight"] = (cfg["num_classes"], cfg["num_classes"])
    sd["head.bias"] = (cfg["num_classes"],)
    return sd


def roberta_param_spec(cfg=ROBERTA_BASE_CONFIG):
    """transformers.RobertaModel(add_pooling_layer=True) parameters in its own order."""
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    sd = OrderedDict()
    sd["embeddings.word_embeddings.weight"] = (cfg["vocab_size"], H)
    sd["embeddings.position_embeddings.weight"] = (cfg["max_position_embeddings"], H)
    sd["embeddings.token_type_embeddings.weight"] = (cfg["type_vocab_size"], H)
    sd["embeddings.LayerNorm.weight"] = (H,)
    sd["embeddings.LayerNorm.bias"] = (H,)
    for i in range(cfg["num_hidden_layers"]):
        p = "encoder.layer.%d." % i
        for t in ("query", "key", "value"):
            sd[p + "attention.self.%s.weight" % t] = (H, H)
            sd[p + "attention.self.%s.bias" % t] = (H,)
        sd[p + "attention.output.dense.weight"] = (H, H)
        sd[p + "attention.output.dense.bias"] = (H,)
        sd[p + "attention.output.LayerNorm.weight"] = (H,)
        sd[p + "attention.output.LayerNorm.bias"] = (H,)
        sd[p + "intermediate.dense.weight"] = (I, H)
        sd[p + "intermediate.dense.bias"] = (I,)
        sd[p + "output.dense.weight"] = (H, I)
        sd[p + "output.dense.bias"] = (H,)
        sd[p + "output.LayerNorm.weight"] = (H,)
        sd[p + "output.LayerNorm.bias"] = (H,)
    sd["pooler.dense.weight"] = (H, H)
    sd["pooler.dense.bias"] = (H,)
    return sd


def clap_param_spec(audio_cfg=HTSAT_BASE_CONFIG, text_cfg=ROBERTA_BASE_CONFIG, joint=CLAP_JOINT_DIM):
    """`clap_module.model.CLAP` (model.py:420-560) with an HTSAT audio branch and a RoBERTa text branch: the entries the
    embedding paths read (`get_audio_embedding` / `get_text_embedding`, model.py:688-744), reference key names."""
    sd = OrderedDict()
    for k, s in htsat_param_spec(audio_cfg).items():
        sd["audio_branch." + k] = s
    for k, s in roberta_param_spec(text_cfg).items():
        sd["text_branch." + k] = s
    nf = audio_cfg["embed_dim"] * 2 ** (len(audio_cfg["depths"]) - 1)
    for name, width in (("text_projection", text_cfg["hidden_size"]), ("audio_projection", nf)):
        sd[name + ".0.weight"] = (joint, width)
        sd[name + ".0.bias"] = (joint,)
        sd[name + ".2.weight"] = (joint, joint)
        sd[name + ".2.bias"] = (joint,)
    return sd


# ---------------------------------------------------------------------------------------------- evaluation classifier
# PANNs Cnn14 as audioldm_eval/eval.py:68-82 builds it for 16 kHz / 32 kHz audio
CNN14_16K_CONFIG = dict(sample_rate=16000, n_fft=512, hop=160, mel_bins=64, fmin=50, fmax=8000, classes_num=527,
                        widths=[64, 128, 256, 512, 1024, 2048])
CNN14_32K_CONFIG = dict(sample_rate=32000, n_fft=1024, hop=320, mel_bins=64, fmin=50, fmax=14000, classes_num=527,
                        widths=[64, 128, 256, 512, 1024, 2048])


def cnn14_param_spec(cfg=CNN14_16K_CONFIG):
    """State-dict keys / shapes of `Cnn14` (audioldm_eval/feature_extractors/panns/models.py:168-234) in the module's own
    registration order, floating-point entries only (num_batches_tracked is structural).  torchlibrosa's frozen STFT / mel
    matrices are listed too so that the released `Cnn14_16k_mAP=0.438.pth` loads; the engine derives them itself."""
    sd = OrderedDict()
    n_fft, F = cfg["n_fft"], cfg["n_fft"] // 2 + 1
    sd["spectrogram_extractor.stft.conv_real.weight"] = (F, 1, n_fft)
    sd["spectrogram_extractor.stft.conv_imag.weight"] = (F, 1, n_fft)
    sd["logmel_extractor.melW"] = (F, cfg["mel_bins"])
    for k in ("weight", "bias", "running_mean", "running_var"):
        sd["bn0." + k] = (cfg["mel_bins"],)
    cin = 1
    for i, c in enumerate(cfg["widths"]):
        p = "conv_block%d." % (i + 1)
        sd[p + "conv1.weight"] = (c, cin, 3, 3)
        sd[p + "conv2.weight"] = (c, c, 3, 3)
        for bn in ("bn1.", "bn2."):
            for k in ("weight", "bias", "running_mean", "running_var"):
                sd[p + bn + k] = (c,)
        cin = c
    sd["fc1.weight"] = (cin, cin)
    sd["fc1.bias"] = (cin,)
    sd["fc_audioset.weight"] = (cfg["classes_num"], cin)
    sd["fc_audioset.bias"] = (cfg["classes_num"],)
    return sd


CNN14_STRUCTURAL = ("spectrogram_extractor.stft.conv_real.weight", "spectrogram_extractor.stft.conv_imag.weight",
                    "logmel_extractor.melW")


def cnn14_det_weight(name, shape, seed=0):
    """Deterministic Cnn14 test weights (no checkpoint exists offline): positive BatchNorm variances, BatchNorm gains around
    1 (bn0: 0.1, which brings the dB-scaled log-mel back to O(1)), He-scaled convolutions so that twelve ReLU layers keep
    their activations O(1)."""
    shape = tuple(shape)
    u = det_uniform(name, shape, seed)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "running_var":
        return (1.0 + 0.5 * np.abs(u)).astype(np.float32)
    if name.endswith("bn0.weight"):
        return (0.1 + 0.02 * u).astype(np.float32)
    if ".bn" in name and leaf == "weight":
        return (1.0 + 0.2 * u).astype(np.float32)
    if len(shape) == 1:
        return (0.05 * u).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    return (np.sqrt(6.0 / fan_in) * u).astype(np.float32)


def clap_det_weight(name, shape, seed=0):
    """Deterministic CLAP test weights: det_weight with the few distribution tweaks the tower needs to stay well
    conditioned at random init (positive BatchNorm variances, O(1) relative-position biases, a bn0 gain that brings the
    dB-scaled log-mel values back to O(1))."""
    shape = tuple(shape)
    if name.endswith("running_var"):
        return (1.0 + 0.5 * np.abs(det_uniform(name, shape, seed))).astype(np.float32)
    if name.endswith("relative_position_bias_table"):
        return (0.5 * det_uniform(name, shape, seed)).astype(np.float32)
    if name.endswith("bn0.weight"):
        return (0.1 + 0.02 * det_uniform(name, shape, seed)).astype(np.float32)
    return det_weight(name, shape, seed)
