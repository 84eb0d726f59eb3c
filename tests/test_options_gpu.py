"""Every library option (include/ctta.h: ctta_set_option) in its NON-default position, through the C ABI.

The library reads no environment variable; these eight ints are its only switches.  Each test sets one option away from its
default, runs the path the option governs, compares with the default position (bit-identical where the option only moves
work between streams / launches / XCDs, within the bf16 tolerance where it changes a summation order), and restores it."""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import cases  # noqa: E402
from consistencytta_amd import _native as N, modules, spec  # noqa: E402
from gpu_util import DEV, bf16_round, conv_desc, det, from_nhwc, nhwc_bf16, pack_conv_weight, rel_err, run_conv  # noqa: E402

BF16_TOL = 1.5 * 2.0 ** -8


@pytest.fixture
def option():
    """set(name, value) with the defaults restored afterwards."""
    before = N.options()

    def set_(name, value):
        N.set_option(name, value)
    yield set_
    for k, v in before.items():
        N.set_option(k, v)


def test_option_table_is_complete_and_rejects_what_it_does_not_know():
    L = N.lib()
    names = [L.ctta_option_name(i).decode() for i in range(L.ctta_num_options())]
    assert names == ["xcd", "splitk", "streamk", "streamk_grid", "wgrad_stream", "gn_fuse", "fused_res", "ffn_fuse"]
    for i, n in enumerate(names):
        assert N.get_option(n) == L.ctta_option_default(i), n      # nothing in the test process has moved one
    with pytest.raises(N.CttaError, match="unknown option"):
        N.set_option("no_such_option", 1)
    with pytest.raises(N.CttaError, match="outside"):
        N.set_option("xcd", 7)


def _conv(B, Cin, H, W, Cout, tile=0, tag="opt"):
    x = bf16_round(det(tag + ".x", (B, Cin, H, W), 1))
    w = bf16_round(det(tag + ".w", (Cout, Cin, 3, 3), 2) * (1.0 / math.sqrt(Cin * 9)))
    b = det(tag + ".b", (Cout,), 3) * 0.1
    ref = F.conv2d(x, w, b, padding=1)
    wp, k_pad = pack_conv_weight(w)
    xa, bd = nhwc_bf16(x), b.to(DEV)
    out = torch.zeros(B, H, W, Cout, dtype=torch.bfloat16, device=DEV)
    run_conv(conv_desc(x0=xa, c0=Cin, batch=B, hi=H, wi=W, ho=H, wo=W, kh=3, kw=3, pad_h=1, pad_w=1, w=wp, k_pad=k_pad, n=Cout,
                       bias=bd, out=out, ldc=Cout, tile=tile))
    return out, ref


def test_xcd_off_moves_tiles_not_numbers(option):
    """"xcd" = 0: plain 2-D grids instead of the XCD-aware tile order.  Which workgroup computes a tile never enters its
    arithmetic: bit-identical outputs on a launch that takes the M-range mapping (many row tiles) and on one that takes the
    weight-slab mapping (few row tiles, heavy weights)."""
    for shape in ((8, 128, 64, 64, 256), (2, 512, 16, 8, 512)):
        a, ref = _conv(*shape)
        option("xcd", 0)
        b, _ = _conv(*shape)
        option("xcd", 1)
        assert torch.equal(a, b), shape
        assert rel_err(from_nhwc(a), ref) < BF16_TOL


def test_splitk_off_walks_the_whole_k_in_one_workgroup(option):
    """"splitk" = 0: deep thin launches are neither split over K (two passes) nor taken as stream-K; "streamk" = 0 keeps the
    two-pass split.  Three summation orders of the same products: each within the bf16 tolerance of F.conv2d, and the
    unsplit walk differs from the split one by fp32 rounding only."""
    shape = (9, 1024, 8, 4, 512)          # M = 288, K = 9216: eight splits by default
    a, ref = _conv(*shape)
    option("splitk", 0)
    b, _ = _conv(*shape)
    option("splitk", 1)
    option("streamk", 0)
    c, _ = _conv(*shape)
    for o in (a, b, c):
        assert rel_err(from_nhwc(o), ref) < 2 * BF16_TOL
    assert rel_err(from_nhwc(a), from_nhwc(b)) < BF16_TOL
    assert torch.equal(a, c)              # this shape takes the two-pass split either way


def test_streamk_off_takes_the_two_pass_split_on_the_same_tile(option):
    """The rule's own territory (K >= 8192, 64 .. 191 big tiles: the distillation teacher's 4608 x 1024 x 9216): by default the
    launch is stream-K on the 256x256x64 tile = what tile 41 gives bit for bit; "streamk" = 0 leaves the same tile with the
    two-pass split-K = tile 29 bit for bit; both within the bf16 tolerance of F.conv2d and of each other."""
    shape = (18, 1024, 64, 4, 1024)
    a, ref = _conv(*shape)
    a41, _ = _conv(*shape, tile=41)
    option("streamk", 0)
    b, _ = _conv(*shape)
    b29, _ = _conv(*shape, tile=29)
    option("streamk", 1)
    assert torch.equal(a, a41) and torch.equal(b, b29) and not torch.equal(a, b)
    assert rel_err(from_nhwc(a), ref) < 2 * BF16_TOL and rel_err(from_nhwc(b), ref) < 2 * BF16_TOL
    assert rel_err(from_nhwc(a), from_nhwc(b)) < BF16_TOL


@pytest.mark.parametrize("grid", [3, 8, 24, 100])
def test_streamk_grid_changes_the_decomposition_not_the_result(option, grid):
    """"streamk_grid" = G: a stream-K launch (tile 41) on G workgroups instead of one per CU slot.  Every G gives another cut
    of the (tile, K step) items -- other owners, other partner counts, 1 / 2 / 4 / 8 XCD chunks -- and the same sums up to fp32
    rounding; G = the default must reproduce the default bit for bit."""
    shape = (4, 256, 16, 16, 512)         # M = 1024: 4 x 2 tiles of 36 K steps
    a, ref = _conv(*shape, tile=41)
    option("streamk_grid", grid)
    b, _ = _conv(*shape, tile=41)
    option("streamk_grid", 0)
    c, _ = _conv(*shape, tile=41)
    assert rel_err(from_nhwc(b), ref) < BF16_TOL and rel_err(from_nhwc(a), from_nhwc(b)) < BF16_TOL
    assert torch.equal(a, c)


def test_wgrad_stream_off_gives_the_same_gradients(option):
    """"wgrad_stream" = 0 at a backward call: the weight-gradient jobs run on the caller's stream in the same scratch slots
    (what bench.py's profiled step needs).  Same kernels, same operands: the gradients agree to run-to-run round-off; and a
    handle CREATED with the option off (no side stream at all, the backward releases its arena blocks) agrees too."""
    cfg = cases.TINY_UNET
    sd = cases.unet_weights(cfg, True, 1)
    x, ts, gs, enc, mask = cases.unet_inputs(cfg, 2, 32, 8, 7, "train_tiny")
    dout = bf16_round(cases.t(spec.det_uniform("train.dout", (2, cfg["out_channels"], 32, 8), 21))) * 0.01

    def grads(m):
        for p in m.parameters():
            p.grad = None
        m.forward_train(x.to(DEV), ts.to(DEV), gs.to(DEV), enc.to(DEV), mask.to(DEV))
        m.backward(dout.to(DEV))
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad}

    m = modules.UNet2DConditionGuidedModel.from_config(cfg)
    m.load_state_dict(sd)
    m = m.to(DEV)
    g_on = grads(m)
    option("wgrad_stream", 0)
    g_off = grads(m)                       # the same handle, side stream bypassed per call
    m2 = modules.UNet2DConditionGuidedModel.from_config(cfg)
    m2.load_state_dict(sd)
    m2 = m2.to(DEV)
    g_never = grads(m2)                    # a handle that never had one
    option("wgrad_stream", 1)
    g_back = grads(m)
    # (not bit-for-bit: LayerNorm's d gamma / d beta and the embedding MLPs fold per-block partials with fp32 atomics, whose
    # order differs from run to run even with every setting equal -- test_train_gpu.py states 1e-7 for two runs)
    def close(a, b_):
        return float((a - b_).norm() / b_.norm().clamp_min(1e-20)) < 2e-6
    for k in g_on:
        assert close(g_off[k], g_on[k]), k
        assert close(g_never[k], g_on[k]), k
        assert close(g_back[k], g_on[k]), k


def test_gn_fuse_and_fused_res_off_agree_with_the_defaults(option):
    """"gn_fuse" = 0: GroupNorm statistics by their own pass instead of the producing convolution's epilogue; "fused_res" = 0:
    one conv_gemm launch per HiFi-GAN ResBlock convolution instead of the fused pair kernels.  Both change where fp32 values
    are rounded, not what is computed: the tiny VAE decoder + vocoder agree within the engines' stated tolerance."""
    def build():
        v = modules.AutoencoderKL(ddconfig=cases.TINY_VAE_DD, embed_dim=8, scale_factor=1.0, hifigan_config=cases.TINY_HIFIGAN)
        sd = dict(cases.vae_weights(cases.TINY_VAE_DD))
        sd.update(cases.hifigan_weights(cases.TINY_HIFIGAN))
        v.load_state_dict(sd)
        return v.to(DEV).eval().requires_grad_(False)

    z = cases.vae_inputs(2, 16, 8, "vae_tiny").to(DEV)
    mel_in = cases.mel_inputs(2, 24, 64, "hifigan_tiny").to(DEV)
    v = build()
    mel0, wav0 = v.decode_first_stage(z).clone(), v.vocode(mel_in).clone()
    option("gn_fuse", 0)
    option("fused_res", 0)
    v2 = build()                           # "fused_res" is read when the handle is built
    mel1, wav1 = v2.decode_first_stage(z), v2.vocode(mel_in)
    assert N.get_option("gn_fuse") == 0 and N.lib().ctta_get_gn_fuse() == 0
    l2 = lambda a, b: float((a - b).norm() / b.norm())
    print("gn_fuse / fused_res off vs on: mel rel_l2 %.2e, wav rel_l2 %.2e" % (l2(mel1, mel0), l2(wav1, wav0)))
    assert l2(mel1, mel0) < 2.5e-2 and l2(wav1, wav0) < 2.5e-2
    assert not torch.equal(wav1, wav0)     # the other path really ran


def test_ffn_fuse_off_runs_the_two_launches_and_gives_the_same_bits(option):
    """"ffn_fuse" = 0 (read when a handle is built): the inference forward runs every transformer feed-forward as ff1 (+ GEGLU
    epilogue) and ff2 (+ residual); by default the 256- and 512-wide blocks take ONE ctta_ffn_geglu launch each when their token
    count fills its row tiles, with attn2.to_out + norm3 in front and proj_out (+ the block's input) behind.  A U-Net with
    256 / 512 / 64 / 64 channels at 128 x 64 latents: ten blocks fuse (four conv_gemm launches and the LayerNorm become one:
    thirty MFMA launches fewer), the 64-wide ones do not, and the output is the same bit for bit."""
    cfg = dict(spec.LIGHT_UNET_CONFIG, block_out_channels=[256, 512, 64, 64], attention_head_dim=[5, 10, 2, 2], cross_attention_dim=48)
    sd = cases.unet_weights(cfg, True, 3)
    x, ts, gs, enc, mask = cases.unet_inputs(cfg, 2, 128, 64, 7, "ffn_opt")
    L = N.lib()

    def run():
        m = modules.UNet2DConditionGuidedModel.from_config(cfg)
        m.load_state_dict(sd)
        m = m.to(DEV).eval().requires_grad_(False)
        with torch.no_grad():
            m(x.to(DEV), ts.to(DEV), gs.to(DEV), enc.to(DEV), encoder_attention_mask=mask.to(DEV))          # sizing + warm-up
            L.ctta_prof_enable(1)
            out = m(x.to(DEV), ts.to(DEV), gs.to(DEV), enc.to(DEV), encoder_attention_mask=mask.to(DEV)).sample.clone()
            torch.cuda.synchronize()
            L.ctta_prof_enable(0)
        ms, fl, cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        N.check(L.ctta_prof_collect(0, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt), None))
        return out, cnt.value
    a, n_on = run()
    option("ffn_fuse", 0)
    b, n_off = run()
    option("ffn_fuse", 1)
    assert n_off - n_on == 30, (n_on, n_off)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
