"""Per-operator parity of the HIP kernels against the CPU oracle, through the C ABI.

bf16 kernels are checked against the fp32 oracle evaluated on the SAME bf16-rounded inputs
and weights, so the only differences are fp32 accumulation order and the final bf16 rounding
of the output: tolerance 1.5 * 2^-8 of the output's max magnitude (BF16_TOL).  fp32 kernels
(Heun / EMA / loss / embeddings) are checked to a few ulp (F32_TOL).
"""
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from consistencytta_amd import _native as N  # noqa: E402
from gpu_util import (DEV, bf16_round, conv_desc, det, from_nhwc, nhwc_bf16, pack_conv_weight,  # noqa: E402
                      rel_err, run_conv, sync)
from oracle import heun as oheun  # noqa: E402
from oracle import nets as onets  # noqa: E402

BF16_TOL = 1.5 * 2.0 ** -8
F32_TOL = 2e-6


def lib():
    return N.lib()


# ------------------------------------------------------------------------------------ conv / gemm
def _conv_case(B, Cin, H, W, Cout, k, stride, pad, tile=0, upsample=False, c_split=None, epilogue=False,
               tag="c"):
    x = bf16_round(det(tag + ".x", (B, Cin, H, W), 1))
    w = bf16_round(det(tag + ".w", (Cout, Cin, k, k), 2) * (1.0 / math.sqrt(Cin * k * k)))
    bias = det(tag + ".b", (Cout,), 3) * 0.1
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if upsample else x
    ref = F.conv2d(xin, w, bias, stride=stride, padding=pad)
    Ho, Wo = ref.shape[2], ref.shape[3]
    wp, k_pad = pack_conv_weight(w)
    out = torch.zeros(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=DEV)
    kw = dict(batch=B, hi=xin.shape[2], wi=xin.shape[3], upsample=int(upsample), ho=Ho, wo=Wo, kh=k, kw=k,
              stride_h=stride, stride_w=stride, pad_h=pad, pad_w=pad, w=wp, k_pad=k_pad, n=Cout,
              bias=bias.to(DEV), out=out, ldc=Cout, tile=tile)
    keep = [wp, out]
    if c_split:
        xa, xb = nhwc_bf16(x[:, :c_split]), nhwc_bf16(x[:, c_split:])
        kw.update(x0=xa, c0=c_split, x1=xb, c1=Cin - c_split)
        keep += [xa, xb]
    else:
        xa = nhwc_bf16(x)
        kw.update(x0=xa, c0=Cin)
        keep.append(xa)
    if epilogue:
        rowvec = det(tag + ".rv", (B, Cout), 4)
        res = bf16_round(det(tag + ".res", (B, Cout, Ho, Wo), 5))
        ref = ref + rowvec[:, :, None, None] + res
        rv, rs = rowvec.to(DEV), nhwc_bf16(res)
        kw.update(rowvec=rv, rowvec_ld=Cout, res=rs, res_ld=Cout)
        keep += [rv, rs]
    kw["bias"] = kw["bias"].contiguous()
    keep.append(kw["bias"])
    run_conv(conv_desc(**kw))
    return rel_err(from_nhwc(out), ref)


@pytest.mark.parametrize("tile", list(range(0, 44)))
def test_conv3x3_all_tiles(tile):
    """41-43: the stream-K forms of tiles 29 / 31 / 17 (M = 768: three row tiles of nine K steps on six workgroups -- every
    tile is split, one of them over three workgroups; six tiles on thirteen workgroups for the 128-row tile)."""
    assert _conv_case(2, 64, 24, 16, 96, 3, 1, 1, tile=tile, tag="t%d" % tile) < BF16_TOL


SK_TILES = [41, 42, 43]


@pytest.mark.parametrize("tile", SK_TILES)
def test_conv_geometries_on_streamk_tiles(tile):
    assert _conv_case(2, 64, 8, 4, 64, 3, 1, 1, upsample=True, tile=tile, tag="g_up") < BF16_TOL
    assert _conv_case(2, 64, 16, 8, 72, 3, 2, 1, tile=tile, tag="g_s2") < BF16_TOL
    assert _conv_case(1, 128, 9, 7, 24, 1, 1, 0, tile=tile, tag="g_1x1") < BF16_TOL
    assert _conv_case(3, 256, 4, 2, 256, 3, 1, 1, tile=tile, tag="g_deepk") < BF16_TOL
    assert _conv_case(2, 64, 8, 8, 64, 3, 1, 1, epilogue=True, tile=tile, tag="g_epi") < BF16_TOL
    assert _conv_case(5, 128, 32, 16, 320, 3, 1, 1, epilogue=True, tile=tile, tag="g_big") < BF16_TOL   # M = 2560, two column tiles


@pytest.mark.parametrize("tile", SK_TILES)
@pytest.mark.parametrize("B,Cin,Cout,H,W,f32", [(9, 1024, 1024, 32, 2, False), (18, 512, 512, 16, 8, False), (9, 1024, 512, 8, 4, True),
                                                (7, 256, 320, 37, 3, False)])
def test_streamk_fold_is_exact_ordered_and_repeatable(tile, B, Cin, Cout, H, W, f32):
    """conv_gemm_sk_kernel (ConvParams::sk_hdr): ONE persistent launch whose workgroups walk equal shares of the (tile, K step)
    items; split tiles are folded inside the launch, in K order, by the workgroup that holds the tile's first K step.
    Checked: (a) against F.conv2d with the whole fused epilogue (bias, per-sample row vector, residual, SiLU; or an fp32
    output), on the distillation step's deep thin shapes (M = 576 x N = 1024 x K = 9216 ...) and on a ragged one (M = 777,
    N = 320: partial row and column tiles); (b) 40 launches in a row on one workspace give BIT-identical outputs (the fold order
    is fixed by the decomposition, the epoch / ticket words are left consistent by every launch); (c) the header afterwards:
    tickets and finished count back at zero, no poll ever timed out, the epoch advanced once per launch."""
    L_ = lib()
    x = bf16_round(det("skf.x", (B, Cin, H, W), 1))
    w = bf16_round(det("skf.w", (Cout, Cin, 3, 3), 2) * (1.0 / math.sqrt(Cin * 9)))
    b = det("skf.b", (Cout,), 3) * 0.1
    rv = det("skf.rv", (B, Cout), 4) * 0.2
    res = bf16_round(det("skf.r", (B, Cout, H, W), 5))
    conv = F.conv2d(x, w, b, padding=1)
    ref = conv if f32 else F.silu(conv + rv[:, :, None, None] + res)
    wp, k_pad = pack_conv_weight(w)
    xa, ra = nhwc_bf16(x), nhwc_bf16(res)
    bd, rvd = b.to(DEV), rv.to(DEV).contiguous()
    ws = torch.zeros(L_.ctta_conv_workspace_bytes(), dtype=torch.uint8, device=DEV)
    L_.ctta_conv_bind_workspace_ex(N.ptr(ws), ws.numel(), 1)
    try:
        outs = []
        for rep in range(40):
            out = torch.full((B, H, W, Cout), float("nan"), dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
            kw = dict(x0=xa, c0=Cin, batch=B, hi=H, wi=W, ho=H, wo=W, kh=3, kw=3, pad_h=1, pad_w=1, w=wp, k_pad=k_pad, n=Cout,
                      bias=bd, out=out, ldc=Cout, tile=tile)
            if f32:
                kw.update(out_f32=1)
            else:
                kw.update(rowvec=rvd, rowvec_ld=Cout, res=ra, res_ld=Cout, out_act=1)
            dsc = conv_desc(**kw)
            N.check(L_.ctta_conv_gemm(ctypes.byref(dsc), N.stream_ptr()))      # back to back: launch i + 1 queues behind launch i
            outs.append(out)
        sync()
        got = outs[0].float().permute(0, 3, 1, 2).cpu()
        assert rel_err(got, ref) < 2 * BF16_TOL
        for o in outs[1:]:
            assert torch.equal(o, outs[0])
        hdr = ws[:16].view(torch.int32).cpu().tolist()
        assert hdr[0] == 0 and hdr[1] == 0 and hdr[3] == 0 and hdr[2] == 40, hdr
    finally:
        L_.ctta_conv_bind_workspace(None, 0)


def test_two_streamk_launches_on_two_streams_share_the_device_without_waiting_for_each_other():
    """Two persistent launches at once, each wanting every CU (the pipelined distillation step does this: the teacher graph's
    deep GEMMs beside the main graph's): whatever share of its workgroups is resident, a stream-K launch must finish -- its only
    waits are bounded, the writer of a tile's last part never waits -- and give the bits it gives alone.  30 rounds of
    (launch A on stream 1 | launch B on stream 2), each launch with its own workspace, an elementwise kernel streaming beside
    them for uneven load; every output compared with the launch run alone."""
    L_ = lib()
    shapes = [(18, 1024, 64, 4, 1024), (16, 1024, 64, 4, 1024)]
    cases_ = []
    for si, (B, Cin, H, W, Cout) in enumerate(shapes):
        x = bf16_round(det("sk2.x%d" % si, (B, Cin, H, W), 1))
        w = bf16_round(det("sk2.w%d" % si, (Cout, Cin, 3, 3), 2) * (1.0 / math.sqrt(Cin * 9)))
        wp, k_pad = pack_conv_weight(w)
        xa = nhwc_bf16(x)
        ws = torch.zeros(L_.ctta_conv_workspace_bytes(), dtype=torch.uint8, device=DEV)
        out = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=DEV)
        d = conv_desc(x0=xa, c0=Cin, batch=B, hi=H, wi=W, ho=H, wo=W, kh=3, kw=3, pad_h=1, pad_w=1, w=wp, k_pad=k_pad, n=Cout,
                      out=out, ldc=Cout, tile=41)
        cases_.append(dict(d=d, ws=ws, out=out, keep=(xa, wp)))
    try:
        alone = []
        for c in cases_:
            L_.ctta_conv_bind_workspace_ex(N.ptr(c["ws"]), c["ws"].numel(), 1)
            N.check(L_.ctta_conv_gemm(ctypes.byref(c["d"]), N.stream_ptr()))
            sync()
            alone.append(c["out"].clone())
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        noise = torch.zeros(64 << 20, device=DEV)
        for rnd in range(30):
            for c in cases_:
                c["out"].fill_(float("nan"))
            sync()
            noise.add_(1.0)                                  # something else on the device while both launch
            for c, st in zip(cases_, streams):
                L_.ctta_conv_bind_workspace_ex(N.ptr(c["ws"]), c["ws"].numel(), 1)
                with torch.cuda.stream(st):
                    N.check(L_.ctta_conv_gemm(ctypes.byref(c["d"]), N.stream_ptr()))
            noise.mul_(0.5)
            sync()
            for c, ref in zip(cases_, alone):
                assert torch.equal(c["out"], ref), rnd
        for c in cases_:
            hdr = c["ws"][:16].view(torch.int32).cpu().tolist()
            assert hdr[0] == 0 and hdr[1] == 0 and hdr[2] == 31, hdr
    finally:
        L_.ctta_conv_bind_workspace(None, 0)


def test_streamk_needs_a_workspace_with_a_zeroed_header():
    """A buffer bound with ctta_conv_bind_workspace() promises nothing about its first bytes: a stream-K tile is refused there
    (and never chosen automatically); the two-pass split-K keeps working on it."""
    L_ = lib()
    mine = torch.full((L_.ctta_conv_workspace_bytes(),), 0x5A, dtype=torch.uint8, device=DEV)
    L_.ctta_conv_bind_workspace(N.ptr(mine), mine.numel())
    try:
        with pytest.raises(RuntimeError, match="stream-K needs a workspace"):
            _conv_case(3, 256, 4, 2, 256, 3, 1, 1, tile=41, tag="skh")
        assert _conv_case(9, 512, 8, 2, 256, 3, 1, 1, tag="skh2") < BF16_TOL      # split over K, slabs behind the header
        assert bool((mine[:L_.ctta_conv_workspace_header_bytes()] == 0x5A).all())    # the header bytes were left alone
    finally:
        L_.ctta_conv_bind_workspace(None, 0)


def test_conv_shapes():
    assert _conv_case(2, 32, 16, 8, 40, 3, 2, 1, tag="s2") < BF16_TOL            # Downsample2D
    assert _conv_case(1, 72, 9, 7, 24, 1, 1, 0, tag="1x1") < BF16_TOL            # ragged extent, 1x1
    assert _conv_case(2, 48, 8, 4, 64, 3, 1, 1, upsample=True, tag="up") < BF16_TOL   # Upsample2D fused
    assert _conv_case(2, 120, 8, 8, 80, 3, 1, 1, c_split=80, tag="cat") < BF16_TOL    # concat gather
    assert _conv_case(2, 64, 8, 8, 64, 3, 1, 1, epilogue=True, tag="epi") < BF16_TOL  # bias+temb+residual
    assert _conv_case(3, 256, 4, 2, 256, 3, 1, 1, tag="deepk") < BF16_TOL        # K = 2304


@pytest.mark.parametrize("tile", [0, 9, 17, 18, 22, 24, 27, 29, 31, 32, 37, 38])
def test_conv_geometries_on_direct_to_lds_paths(tile):
    """Upsample-fused, stride-2, ragged and deep-K geometries through the generic (mode 1) and
    descriptor (mode 2) direct-to-LDS paths (Cin multiples of 64 so that mode 2 is eligible)."""
    assert _conv_case(2, 64, 8, 4, 64, 3, 1, 1, upsample=True, tile=tile, tag="g_up") < BF16_TOL
    assert _conv_case(2, 64, 16, 8, 72, 3, 2, 1, tile=tile, tag="g_s2") < BF16_TOL
    assert _conv_case(1, 128, 9, 7, 24, 1, 1, 0, tile=tile, tag="g_1x1") < BF16_TOL
    assert _conv_case(3, 256, 4, 2, 256, 3, 1, 1, tile=tile, tag="g_deepk") < BF16_TOL
    assert _conv_case(2, 64, 8, 8, 64, 3, 1, 1, epilogue=True, tile=tile, tag="g_epi") < BF16_TOL


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 256, 128, 8, 8), (9, 512, 256, 8, 2), (1, 1024, 64, 4, 4)])
def test_conv_split_k(B, Cin, Cout, H, W):
    """Deep, narrow problems (few output tiles, K >= 2048) are split over K automatically: partial slabs + a second
    pass that runs the fused epilogue (bias, per-sample row vector, residual, SiLU) -- checked against F.conv2d."""
    x = bf16_round(det("sk.x", (B, Cin, H, W), 1))
    w = bf16_round(det("sk.w", (Cout, Cin, 3, 3), 2) * (1.0 / math.sqrt(Cin * 9)))
    b = det("sk.b", (Cout,), 3) * 0.1
    rv = det("sk.rv", (B, Cout), 4) * 0.2
    res = bf16_round(det("sk.r", (B, Cout, H, W), 5))
    ref = F.silu(F.conv2d(x, w, b, padding=1) + rv[:, :, None, None] + res)
    wp, k_pad = pack_conv_weight(w)
    xa, ra = nhwc_bf16(x), nhwc_bf16(res)
    bd, rvd = b.to(DEV), rv.to(DEV).contiguous()
    out = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=DEV)
    run_conv(conv_desc(x0=xa, c0=Cin, batch=B, hi=H, wi=W, ho=H, wo=W, kh=3, kw=3, pad_h=1, pad_w=1, w=wp, k_pad=k_pad,
                       n=Cout, bias=bd, rowvec=rvd, rowvec_ld=Cout, res=ra, res_ld=Cout, out_act=1, out=out, ldc=Cout))
    assert rel_err(from_nhwc(out), ref) < 2 * BF16_TOL


@pytest.mark.parametrize("rows,k,hid,tile", [(300, 256, 1024, 0), (70, 512, 2048, 0), (129, 64, 96, 0), (300, 256, 1024, 21),
                                             (300, 256, 1024, 24), (300, 256, 1024, 12)])
def test_linear_with_fused_geglu_epilogue(rows, k, hid, tile):
    """out_act=4: rows of the packed weight interleaved in 16-blocks [16 value][16 gate]; the epilogue writes
    value * gelu(gate) at half the GEMM width (GEGLU, attention.py:430-432)."""
    x = bf16_round(det("fg.x", (rows, k), 1))
    w = bf16_round(det("fg.w", (2 * hid, k), 2) * (1.5 / math.sqrt(k)))
    b = det("fg.b", (2 * hid,), 3) * 0.2
    proj = F.linear(x, w, b)
    ref = proj[:, :hid] * F.gelu(proj[:, hid:])
    hidp = (hid + 15) // 16 * 16
    wi, bi = torch.zeros(2 * hidp, k), torch.zeros(2 * hidp)
    for j in range(hid):
        wi[(j // 16) * 32 + j % 16], wi[(j // 16) * 32 + 16 + j % 16] = w[j], w[hid + j]
        bi[(j // 16) * 32 + j % 16], bi[(j // 16) * 32 + 16 + j % 16] = b[j], b[hid + j]
    wp, k_pad = pack_conv_weight(wi[:, :, None, None])
    xd = x.to(torch.bfloat16).to(DEV)
    out = torch.empty(rows, hidp, dtype=torch.bfloat16, device=DEV)
    run_conv(conv_desc(x0=xd, c0=k, batch=1, hi=rows, wi=1, ho=rows, wo=1, w=wp, k_pad=k_pad, n=2 * hidp, bias=bi.to(DEV),
                       out=out, ldc=hidp, out_act=4, tile=tile))
    got = out.float().cpu()
    assert rel_err(got[:, :hid], ref) < 2 * BF16_TOL
    assert float(got[:, hid:].abs().max()) == 0.0 if hidp > hid else True


def _ffn_case(cp, hid_valid, M, seed):
    """weights of one GEGLU feed-forward in the engine's packed layouts + the two-launch form as a closure"""
    ffp = 4 * cp
    d = cp - cp // 256                                     # 255 -> 256, 510 -> 512: the light U-Net's inner widths
    w1 = bf16_round(det("ffn.w1.%d" % cp, (2 * hid_valid, d), seed) * (1.5 / math.sqrt(d)))
    b1 = det("ffn.b1.%d" % cp, (2 * hid_valid,), seed + 1) * 0.2
    w2 = bf16_round(det("ffn.w2.%d" % cp, (d, hid_valid), seed + 2) * (1.0 / math.sqrt(hid_valid)))
    b2 = det("ffn.b2.%d" % cp, (d,), seed + 3) * 0.1
    wi, bi = torch.zeros(2 * ffp, cp), torch.zeros(2 * ffp)          # ff1 rows in 16-blocks [16 value][16 gate], K padded to cp
    for j in range(hid_valid):
        wi[(j // 16) * 32 + j % 16, :d], wi[(j // 16) * 32 + 16 + j % 16, :d] = w1[j], w1[hid_valid + j]
        bi[(j // 16) * 32 + j % 16], bi[(j // 16) * 32 + 16 + j % 16] = b1[j], b1[hid_valid + j]
    w2i, b2i = torch.zeros(cp, ffp), torch.zeros(cp)
    w2i[:d, :hid_valid], b2i[:d] = w2, b2
    w1p, k1 = pack_conv_weight(wi[:, :, None, None])
    w2p, k2 = pack_conv_weight(w2i[:, :, None, None])
    L = lib()
    packed = torch.empty(L.ctta_ffn_pack_bytes(cp, ffp), dtype=torch.uint8, device=DEV)
    N.check(L.ctta_ffn_pack(N.ptr(w1p), k1, N.ptr(w2p), k2, cp, ffp, N.ptr(packed), N.stream_ptr()))
    return dict(cp=cp, ffp=ffp, d=d, w1=w1, b1=b1, w2=w2, b2=b2, w1p=w1p, k1=k1, w2p=w2p, k2=k2, b1d=bi.to(DEV), b2d=b2i.to(DEV),
                packed=packed, hid=hid_valid)


def _ffn_two_launches(c, n3, s2, out):
    """(with "splitk" off: at a few hundred rows the K = 2048 ff2 launch of the 512-wide block would be cut over K -- another
    summation order than the one walk the engine's launches take at their sizes)"""
    M, cp, ffp = n3.shape[0], c["cp"], c["ffp"]
    gg = torch.empty(M, ffp, dtype=torch.bfloat16, device=DEV)
    N.set_option("splitk", 0)
    run_conv(conv_desc(x0=n3, c0=cp, batch=1, hi=M, wi=1, ho=M, wo=1, w=c["w1p"], k_pad=c["k1"], n=2 * ffp, bias=c["b1d"], out=gg,
                       ldc=ffp, out_act=4))
    run_conv(conv_desc(x0=gg, c0=ffp, batch=1, hi=M, wi=1, ho=M, wo=1, w=c["w2p"], k_pad=c["k2"], n=cp, bias=c["b2d"], res=s2,
                       res_ld=cp, out=out, ldc=cp))
    N.set_option("splitk", 1)


@pytest.mark.parametrize("cp,M", [(256, 300), (256, 1), (256, 128 * 5 + 17), (512, 300), (512, 48 * 7 + 5), (512, 16)])
def test_fused_geglu_feed_forward_equals_the_two_launches_bit_for_bit(cp, M):
    """ctta_ffn_geglu (csrc/ffn_fused.hip): ff1 -> chunk -> value * gelu(gate) -> ff2 + bias + residual of one transformer block
    (attention.py:276-334, 383-386, 430-432) in ONE launch, the hidden activations in LDS.  Same products in the same K order with
    the same two bf16 roundings as ctta_conv_gemm(out_act = 4) + ctta_conv_gemm(res): equal bit for bit, with EVERY row tile of the
    width (ragged last tiles included), and within the bf16 tolerance of the fp32 torch expression.  The output's pad columns
    and the rows past M are left alone."""
    L = lib()
    c = _ffn_case(cp, 4 * (cp - cp // 256), M, 11)
    d, ffp = c["d"], c["ffp"]
    n3 = torch.zeros(M, cp)
    n3[:, :d] = bf16_round(det("ffn.n3", (M, d), 5))
    s2 = torch.zeros(M, cp)
    s2[:, :d] = bf16_round(det("ffn.s2", (M, d), 6))
    proj = F.linear(n3[:, :d], c["w1"], c["b1"])
    ref = s2[:, :d] + F.linear(bf16_round(proj[:, :c["hid"]] * F.gelu(proj[:, c["hid"]:])), c["w2"], c["b2"])
    n3d, s2d = n3.to(torch.bfloat16).to(DEV), s2.to(torch.bfloat16).to(DEV)
    two = torch.full((M, cp), 3.0, dtype=torch.bfloat16, device=DEV)
    _ffn_two_launches(c, n3d, s2d, two)
    assert rel_err(two[:, :d].float().cpu(), ref) < 2 * BF16_TOL
    for bm in ((0, 128, 144) if cp == 256 else (0, 48, 64, 80)):
        L.ctta_ffn_debug_rows(bm)
        big = torch.full((M + 3, cp + 8), 3.0, dtype=torch.bfloat16, device=DEV)          # a wider, longer destination
        N.check(L.ctta_ffn_geglu(N.ptr(n3d), cp, M, cp, ffp, N.ptr(c["packed"]), N.ptr(c["b1d"]), N.ptr(c["b2d"]), N.ptr(s2d), cp,
                                 N.ptr(big), cp + 8, cp, None, None, 0, 0.0, N.stream_ptr()))
        sync()
        L.ctta_ffn_debug_rows(0)
        assert torch.equal(big[:M, :cp], two), (cp, M, bm)
        assert bool((big[M:] == 3.0).all()) and bool((big[:, cp:] == 3.0).all()), (cp, M, bm)
    # n_valid < cp: only the first n_valid columns are written (what a destination narrower than the padded width needs)
    part = torch.full((M, cp), 3.0, dtype=torch.bfloat16, device=DEV)
    N.check(L.ctta_ffn_geglu(N.ptr(n3d), cp, M, cp, ffp, N.ptr(c["packed"]), N.ptr(c["b1d"]), N.ptr(c["b2d"]), N.ptr(s2d), cp,
                             N.ptr(part), cp, cp - 12, None, None, 0, 0.0, N.stream_ptr()))
    sync()
    assert torch.equal(part[:, :cp - 12], two[:, :cp - 12]) and bool((part[:, cp - 12:] == 3.0).all())


@pytest.mark.parametrize("cp,M", [(256, 777), (512, 333)])
def test_fused_feed_forward_with_layernorm_on_load(cp, M):
    """ln_gamma != NULL: x is norm3's INPUT, the rows are normalised while the workgroup stages them (the lane mapping, the
    summation order and the arithmetic of ctta_layernorm's kernel for the width): bit-identical to ctta_layernorm followed by the
    fused call -- and therefore to the three launches."""
    L = lib()
    c = _ffn_case(cp, 4 * (cp - cp // 256), M, 21)
    d, ffp = c["d"], c["ffp"]
    s2 = torch.zeros(M, cp)
    s2[:, :d] = bf16_round(det("ffnln.s2", (M, d), 6) * 1.3 + 0.2)
    s2d = s2.to(torch.bfloat16).to(DEV)
    gamma, beta = (1 + 0.2 * det("ffnln.g", (d,), 7)).to(DEV), (0.1 * det("ffnln.b", (d,), 8)).to(DEV)
    n3d = torch.empty_like(s2d)
    N.check(L.ctta_layernorm(N.ptr(s2d), N.ptr(n3d), M, d, cp, N.ptr(gamma), N.ptr(beta), 1e-5, N.stream_ptr()))
    three = torch.empty(M, cp, dtype=torch.bfloat16, device=DEV)
    _ffn_two_launches(c, n3d, s2d, three)
    got = torch.empty(M, cp, dtype=torch.bfloat16, device=DEV)
    N.check(L.ctta_ffn_geglu(N.ptr(s2d), cp, M, cp, ffp, N.ptr(c["packed"]), N.ptr(c["b1d"]), N.ptr(c["b2d"]), N.ptr(s2d), cp,
                             N.ptr(got), cp, cp, N.ptr(gamma), N.ptr(beta), d, 1e-5, N.stream_ptr()))
    sync()
    assert torch.equal(got, three)
    n_ref = F.layer_norm(s2[:, :d], (d,), gamma.cpu(), beta.cpu(), 1e-5)
    proj = F.linear(bf16_round(n_ref), c["w1"], c["b1"])
    ref = s2[:, :d] + F.linear(bf16_round(proj[:, :c["hid"]] * F.gelu(proj[:, c["hid"]:])), c["w2"], c["b2"])
    assert rel_err(got[:, :d].float().cpu(), ref) < 3 * BF16_TOL


@pytest.mark.parametrize("cp,M", [(256, 500), (256, 144 * 3), (512, 333)])
def test_fused_feed_forward_with_the_tail_projection(cp, M):
    """ctta_ffn_block with proj_packed: proj_out of the Transformer2DModel (+ bias + the block's input as residual,
    transformer_2d.py) as one more GEMM inside the workgroup, on the bf16-rounded feed-forward result that the separate launch
    would have stored: bit-identical to ff1 + ff2 + proj_out as three ctta_conv_gemm launches, every row tile, a destination
    narrower than the padded width."""
    L = lib()
    c = _ffn_case(cp, 4 * (cp - cp // 256), M, 31)
    d, ffp = c["d"], c["ffp"]
    n_out = cp - 4
    wp_ = bf16_round(det("ffnp.w", (n_out, d), 9) * (1.0 / math.sqrt(d)))
    bp_ = det("ffnp.b", (n_out,), 10) * 0.1
    wpi, bpi = torch.zeros(cp, cp), torch.zeros(cp)
    wpi[:n_out, :d], bpi[:n_out] = wp_, bp_
    w3p, k3 = pack_conv_weight(wpi[:, :, None, None])
    assert k3 == cp
    pstream = torch.empty(L.ctta_ffn_proj_pack_bytes(cp, cp), dtype=torch.uint8, device=DEV)
    N.check(L.ctta_ffn_proj_pack(N.ptr(w3p), k3, cp, cp, N.ptr(pstream), N.stream_ptr()))
    n3 = torch.zeros(M, cp)
    n3[:, :d] = bf16_round(det("ffnp.n3", (M, d), 5))
    s2 = torch.zeros(M, cp)
    s2[:, :d] = bf16_round(det("ffnp.s2", (M, d), 6))
    xin = bf16_round(det("ffnp.x", (M, cp), 7))
    n3d, s2d, xd = n3.to(torch.bfloat16).to(DEV), s2.to(torch.bfloat16).to(DEV), xin.to(torch.bfloat16).to(DEV)
    s3 = torch.empty(M, cp, dtype=torch.bfloat16, device=DEV)
    _ffn_two_launches(c, n3d, s2d, s3)
    three = torch.full((M, cp), 3.0, dtype=torch.bfloat16, device=DEV)
    bpd = bpi.to(DEV)
    run_conv(conv_desc(x0=s3, c0=cp, batch=1, hi=M, wi=1, ho=M, wo=1, w=w3p, k_pad=k3, n=n_out, bias=bpd, res=xd, res_ld=cp, out=three,
                       ldc=cp))
    ref = xin[:, :n_out] + F.linear(s3[:, :d].float().cpu(), wp_, bp_)
    assert rel_err(three[:, :n_out].float().cpu(), ref) < 2 * BF16_TOL
    fd = N.FfnDesc()
    L.ctta_ffn_desc_init(ctypes.byref(fd))
    fd.x, fd.ld_x, fd.M, fd.cp, fd.ffp = n3d.data_ptr(), cp, M, cp, ffp
    fd.packed, fd.b1, fd.b2 = c["packed"].data_ptr(), c["b1d"].data_ptr(), c["b2d"].data_ptr()
    fd.res, fd.res_ld = s2d.data_ptr(), cp
    fd.proj_packed, fd.proj_bias, fd.proj_res, fd.proj_res_ld = pstream.data_ptr(), bpd.data_ptr(), xd.data_ptr(), cp
    for bm in ((0, 128, 144) if cp == 256 else (0, 48, 64, 80)):
        got = torch.full((M + 2, cp), 3.0, dtype=torch.bfloat16, device=DEV)
        fd.out, fd.ldc, fd.n_valid = got.data_ptr(), cp, n_out
        L.ctta_ffn_debug_rows(bm)
        N.check(L.ctta_ffn_block(ctypes.byref(fd), N.stream_ptr()))
        sync()
        L.ctta_ffn_debug_rows(0)
        assert torch.equal(got[:M], three), (cp, M, bm)
        assert bool((got[M:] == 3.0).all())
    fd.proj_bias = None
    with pytest.raises(RuntimeError, match="tail projection needs"):
        N.check(L.ctta_ffn_block(ctypes.byref(fd), N.stream_ptr()))


@pytest.mark.parametrize("cp,M,tail", [(256, 500, True), (256, 144 * 2 + 9, False), (512, 333, True), (512, 80 * 3, False)])
def test_fused_feed_forward_with_the_front_projection(cp, M, tail):
    """ctta_ffn_block with front_packed: attn2.to_out + bias + residual (attention.py:318-327) as a GEMM of the workgroup in
    front of the feed-forward -- its result s2 is stored (the residual of ff2), normalised in LDS (norm3) and fed on; with the
    tail projection that is the transformer block from the cross-attention's output to the Transformer2DModel's output in one
    launch.  Bit-identical to ctta_conv_gemm (to_out) + ctta_layernorm + ff1 + ff2 (+ proj_out) as separate launches, every
    row tile, s2 included."""
    L = lib()
    c = _ffn_case(cp, 4 * (cp - cp // 256), M, 41)
    d, ffp = c["d"], c["ffp"]
    hp, hv = cp // 4 * 5, cp // 256 * 255                    # 320 / 640 head-padded attention width: heads x 64, 51 of 64 live
    w0 = bf16_round(det("ffnf.w0", (d, hp), 3) * (1.0 / math.sqrt(hv)))
    w0.view(d, hp // 64, 64)[:, :, 51:] = 0
    b0 = det("ffnf.b0", (d,), 4) * 0.1
    w0i, b0i = torch.zeros(cp, hp), torch.zeros(cp)
    w0i[:d], b0i[:d] = w0, b0
    w0p, k0 = pack_conv_weight(w0i[:, :, None, None], k_mult=32)
    assert k0 == hp
    fstream = torch.empty(L.ctta_ffn_proj_pack_bytes(cp, hp), dtype=torch.uint8, device=DEV)
    N.check(L.ctta_ffn_proj_pack(N.ptr(w0p), k0, hp, cp, N.ptr(fstream), N.stream_ptr()))
    att = bf16_round(det("ffnf.att", (M, hp), 5))
    s1 = torch.zeros(M, cp)
    s1[:, :d] = bf16_round(det("ffnf.s1", (M, d), 6))
    xin = bf16_round(det("ffnf.x", (M, cp), 7))
    attd, s1d, xd = att.to(torch.bfloat16).to(DEV), s1.to(torch.bfloat16).to(DEV), xin.to(torch.bfloat16).to(DEV)
    gamma, beta = (1 + 0.2 * det("ffnf.g", (d,), 8)).to(DEV), (0.1 * det("ffnf.be", (d,), 9)).to(DEV)
    b0d = b0i.to(DEV)
    # the separate launches
    s2 = torch.full((M, cp), 3.0, dtype=torch.bfloat16, device=DEV)
    run_conv(conv_desc(x0=attd, c0=hp, batch=1, hi=M, wi=1, ho=M, wo=1, w=w0p, k_pad=k0, n=cp, bias=b0d, res=s1d, res_ld=cp, out=s2,
                       ldc=cp))
    assert rel_err(s2[:, :d].float().cpu(), s1[:, :d] + F.linear(att, w0, b0)) < 2 * BF16_TOL
    n3 = torch.empty_like(s2)
    N.check(L.ctta_layernorm(N.ptr(s2), N.ptr(n3), M, d, cp, N.ptr(gamma), N.ptr(beta), 1e-5, N.stream_ptr()))
    s3 = torch.empty(M, cp, dtype=torch.bfloat16, device=DEV)
    _ffn_two_launches(c, n3, s2, s3)
    want = s3
    fd = N.FfnDesc()
    L.ctta_ffn_desc_init(ctypes.byref(fd))
    fd.M, fd.cp, fd.ffp = M, cp, ffp
    fd.packed, fd.b1, fd.b2 = c["packed"].data_ptr(), c["b1d"].data_ptr(), c["b2d"].data_ptr()
    fd.front_packed, fd.front_bias, fd.att, fd.att_ld, fd.front_k = fstream.data_ptr(), b0d.data_ptr(), attd.data_ptr(), hp, hp
    fd.front_res, fd.front_res_ld = s1d.data_ptr(), cp
    fd.ln_gamma, fd.ln_beta, fd.ln_d, fd.ln_eps = gamma.data_ptr(), beta.data_ptr(), d, 1e-5
    n_out = cp
    if tail:
        n_out = cp - 8
        wp_ = bf16_round(det("ffnf.wp", (n_out, d), 10) * (1.0 / math.sqrt(d)))
        wpi, bpi = torch.zeros(cp, cp), torch.zeros(cp)
        wpi[:n_out, :d], bpi[:n_out] = wp_, det("ffnf.bp", (n_out,), 11) * 0.1
        w3p, k3 = pack_conv_weight(wpi[:, :, None, None])
        pstream = torch.empty(L.ctta_ffn_proj_pack_bytes(cp, cp), dtype=torch.uint8, device=DEV)
        N.check(L.ctta_ffn_proj_pack(N.ptr(w3p), k3, cp, cp, N.ptr(pstream), N.stream_ptr()))
        bpd = bpi.to(DEV)
        want = torch.full((M, cp), 3.0, dtype=torch.bfloat16, device=DEV)
        run_conv(conv_desc(x0=s3, c0=cp, batch=1, hi=M, wi=1, ho=M, wo=1, w=w3p, k_pad=k3, n=n_out, bias=bpd, res=xd, res_ld=cp, out=want,
                           ldc=cp))
        fd.proj_packed, fd.proj_bias, fd.proj_res, fd.proj_res_ld = pstream.data_ptr(), bpd.data_ptr(), xd.data_ptr(), cp
    for bm in ((0, 128, 144) if cp == 256 else (0, 48, 64, 80)):
        got = torch.full((M + 2, cp), 3.0, dtype=torch.bfloat16, device=DEV)
        s2g = torch.full((M + 2, cp), 5.0, dtype=torch.bfloat16, device=DEV)
        fd.out, fd.ldc, fd.n_valid = got.data_ptr(), cp, n_out
        fd.s2_out, fd.s2_ld = s2g.data_ptr(), cp
        L.ctta_ffn_debug_rows(bm)
        N.check(L.ctta_ffn_block(ctypes.byref(fd), N.stream_ptr()))
        sync()
        L.ctta_ffn_debug_rows(0)
        assert torch.equal(s2g[:M], s2) and bool((s2g[M:] == 5.0).all()), (cp, M, bm)
        assert torch.equal(got[:M], want), (cp, M, bm)
        assert bool((got[M:] == 3.0).all())
    fd.s2_out = None
    with pytest.raises(RuntimeError, match="front projection needs"):
        N.check(L.ctta_ffn_block(ctypes.byref(fd), N.stream_ptr()))


def test_fused_feed_forward_refuses_what_it_cannot_run_and_picks_tiles_by_rounds():
    L = lib()
    assert L.ctta_ffn_geglu_supported(256, 1024) == 1 and L.ctta_ffn_geglu_supported(512, 2048) == 1
    assert L.ctta_ffn_geglu_supported(1024, 4096) == 0 and L.ctta_ffn_geglu_supported(256, 128) == 0 and L.ctta_ffn_geglu_supported(320, 1280) == 0
    assert L.ctta_ffn_pack_bytes(320, 1280) == 0
    # the rule the engines ask (256 CUs): full rounds and single rounds of >= 64 tiles are wanted, a second round left > 40 % empty is not
    for cp, M, want in ((256, 131072, 1), (256, 73728, 1), (256, 36864, 1), (256, 8192, 1), (256, 4096, 0), (256, 37000, 0),
                        (512, 32768, 1), (512, 18432, 1), (512, 9216, 1), (512, 1024, 0), (1024, 8192, 0)):
        assert L.ctta_ffn_geglu_wanted(cp, 4 * cp, M) == want, (cp, M)
    x = torch.zeros(64, 256, dtype=torch.bfloat16, device=DEV)
    pk = torch.zeros(L.ctta_ffn_pack_bytes(256, 1024), dtype=torch.uint8, device=DEV)
    b = torch.zeros(2048, device=DEV)
    with pytest.raises(RuntimeError, match="outside the fused kernel's range"):
        N.check(L.ctta_ffn_geglu(N.ptr(x), 256, 64, 320, 1280, N.ptr(pk), N.ptr(b), N.ptr(b), N.ptr(x), 256, N.ptr(x), 256, 256, None, None, 0,
                                 0.0, N.stream_ptr()))
    with pytest.raises(RuntimeError, match="bad extents"):
        N.check(L.ctta_ffn_geglu(N.ptr(x), 250, 64, 256, 1024, N.ptr(pk), N.ptr(b), N.ptr(b), N.ptr(x), 256, N.ptr(x), 256, 256, None, None, 0,
                                 0.0, N.stream_ptr()))


def test_conv1d_dilated_lrelu_and_accumulate():
    B, C, L, k, d = 2, 32, 200, 7, 3
    x = bf16_round(det("c1d.x", (B, C, L), 1))
    w = bf16_round(det("c1d.w", (C, C, k), 2) * (1.0 / math.sqrt(C * k)))
    b = det("c1d.b", (C,), 3) * 0.1
    res = bf16_round(det("c1d.r", (B, C, L), 4))
    old = bf16_round(det("c1d.o", (B, C, L), 5))
    pad = (k * d - d) // 2
    ref = (F.conv1d(F.leaky_relu(x, 0.1), w, b, dilation=d, padding=pad) + res + old) / 3
    wp, k_pad = pack_conv_weight(w[:, :, None, :])
    out = old.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    xa = x.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    rs = res.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    bd = b.to(DEV)
    run_conv(conv_desc(x0=xa, c0=C, batch=B, hi=1, wi=L, ho=1, wo=L, kh=1, kw=k, pad_w=pad, dil_w=d, w=wp,
                       k_pad=k_pad, n=C, bias=bd, in_act=1, in_slope=0.1, res=rs, res_ld=C, accumulate=1,
                       alpha=1.0 / 3.0, out=out, ldc=C))
    got = out.to(torch.float32).permute(0, 2, 1).cpu()
    assert rel_err(got, ref) < 2 * BF16_TOL   # `old` is itself bf16 and re-rounded


@pytest.mark.parametrize("C,k,d,L", [(32, 11, 5, 777), (32, 3, 1, 256), (32, 7, 3, 1300), (32, 11, 1, 90), (64, 7, 3, 500)])
def test_conv1d_halo_kernel(C, k, d, L):
    """Stride-1 1-D convs with 32 channels (last HiFi-GAN stage) take the LDS-halo kernel automatically (C=64 stays on
    the generic path); same epilogue options as the generic implicit GEMM (forced with tile=13), both checked against
    F.conv1d."""
    B = 2
    x = bf16_round(det("halo.x", (B, C, L), 1))
    w = bf16_round(det("halo.w", (C, C, k), 2) * (1.0 / math.sqrt(C * k)))
    b = det("halo.b", (C,), 3) * 0.1
    res = bf16_round(det("halo.r", (B, C, L), 4))
    old = bf16_round(det("halo.o", (B, C, L), 5))
    pad = (k * d - d) // 2
    ref = F.leaky_relu((F.conv1d(x, w, b, dilation=d, padding=pad) + res + old) * 0.5, 0.1)
    wp, k_pad = pack_conv_weight(w[:, :, None, :])
    xa = x.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    rs = res.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    bd = b.to(DEV)
    outs = []
    for tile in (0, 13):
        out = old.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
        out2 = torch.empty_like(out)
        run_conv(conv_desc(x0=xa, c0=C, batch=B, hi=1, wi=L, ho=1, wo=L, kh=1, kw=k, pad_w=pad, dil_w=d, w=wp, k_pad=k_pad,
                           n=C, bias=bd, res=rs, res_ld=C, accumulate=1, alpha=0.5, out_act=3, out_slope=0.1, out=out, ldc=C,
                           out2=out2, out2_slope=0.1, tile=tile))
        got = out.to(torch.float32).permute(0, 2, 1).cpu()
        assert rel_err(got, ref) < 2 * BF16_TOL, tile
        assert torch.equal(out2.float().cpu(), bf16_round(F.leaky_relu(out.float().cpu(), 0.1)))
        outs.append(got)
    assert rel_err(outs[0], outs[1]) < BF16_TOL      # halo kernel vs generic kernel


@pytest.mark.parametrize("C,k,d,L,B", [(128, 11, 5, 700, 2), (128, 3, 1, 129, 1), (128, 7, 3, 128, 2), (64, 11, 5, 1000, 2),
                                       (64, 3, 3, 255, 1), (64, 7, 1, 513, 2), (32, 11, 5, 1500, 2), (32, 3, 1, 40, 3),
                                       (32, 7, 5, 1025, 1), (128, 11, 1, 50, 1), (256, 3, 1, 700, 2), (256, 3, 5, 129, 1),
                                       (256, 7, 3, 128, 3), (256, 11, 5, 300, 1), (512, 3, 5, 200, 2), (512, 3, 1, 64, 1),
                                       (512, 7, 5, 200, 2), (512, 11, 3, 97, 1), (512, 11, 5, 245, 2), (512, 11, 5, 80, 1)])
def test_fused_resblock_unit(C, k, d, L, B):
    """hifigan/models.py:56-63 as ONE launch (resunit.hip): x + conv2(lrelu(conv1(lrelu(x)))) with the intermediate in
    LDS, vs F.conv1d on the same bf16-rounded operands (intermediate rounded to bf16 like the two-launch path), for
    every channel width the kernel serves, sequence ends inside / on / beyond tile borders, and the stage-fold epilogue
    (accumulate, scale, final leaky_relu).  Also checked against the two generic conv_gemm launches it replaces."""
    L_ = lib()
    assert L_.ctta_resunit_supported(C, k, d) == 1
    x = bf16_round(det("ru.x", (B, C, L), 1))
    w1 = bf16_round(det("ru.w1", (C, C, k), 2) * (1.0 / math.sqrt(C * k)))
    w2 = bf16_round(det("ru.w2", (C, C, k), 3) * (1.0 / math.sqrt(C * k)))
    b1, b2 = det("ru.b1", (C,), 4) * 0.1, det("ru.b2", (C,), 5) * 0.1
    old = bf16_round(det("ru.o", (B, C, L), 6))
    mid = bf16_round(F.leaky_relu(F.conv1d(F.leaky_relu(x, 0.1), w1, b1, dilation=d, padding=(k * d - d) // 2), 0.1))
    unit = x + F.conv1d(mid, w2, b2, padding=(k - 1) // 2)
    xa = x.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    frags = []
    for w in (w1, w2):
        wp, k_pad = pack_conv_weight(w[:, :, None, :])
        f = torch.empty(C * k * C, dtype=torch.bfloat16, device=DEV)
        N.check(L_.ctta_frag_pack(N.ptr(wp), C, k_pad, k * C, N.ptr(f), N.stream_ptr()))
        frags.append((f, wp, k_pad))
    b1d, b2d = b1.to(DEV), b2.to(DEV)
    # plain unit
    out = torch.empty_like(xa)
    N.check(L_.ctta_resunit_conv1d(N.ptr(xa), B, L, C, k, d, N.ptr(frags[0][0]), N.ptr(b1d), N.ptr(frags[1][0]), N.ptr(b2d),
                                   0.1, N.ptr(out), 0, 1.0, 0.0, N.stream_ptr()))
    sync()
    got = out.float().permute(0, 2, 1).cpu()
    assert rel_err(got, unit) < 2 * BF16_TOL
    # the two generic launches it replaces (activated input, activated bf16 intermediate, residual epilogue)
    xact = bf16_round(F.leaky_relu(x, 0.1)).permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    t1, t2 = torch.empty_like(xa), torch.empty_like(xa)
    run_conv(conv_desc(x0=xact, c0=C, batch=B, hi=1, wi=L, ho=1, wo=L, kh=1, kw=k, pad_w=(k * d - d) // 2, dil_w=d,
                       w=frags[0][1], k_pad=frags[0][2], n=C, bias=b1d, out_act=3, out_slope=0.1, out=t1, ldc=C, tile=13))
    run_conv(conv_desc(x0=t1, c0=C, batch=B, hi=1, wi=L, ho=1, wo=L, kh=1, kw=k, pad_w=(k - 1) // 2, w=frags[1][1],
                       k_pad=frags[1][2], n=C, bias=b2d, res=xa, res_ld=C, out=t2, ldc=C, tile=13))
    # two bf16-rounded results whose fp32 sums were taken in different orders (round 3: the generic launch starts its
    # accumulators from the bias): they may differ by one bf16 ulp of the largest magnitude
    assert rel_err(got, t2.float().permute(0, 2, 1).cpu()) <= 2.0 ** -7
    # stage fold: (old + unit) / 3 then leaky_relu(0.01), accumulated in place
    out = old.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    N.check(L_.ctta_resunit_conv1d(N.ptr(xa), B, L, C, k, d, N.ptr(frags[0][0]), N.ptr(b1d), N.ptr(frags[1][0]), N.ptr(b2d),
                                   0.1, N.ptr(out), 1, 1.0 / 3.0, 0.01, N.stream_ptr()))
    sync()
    ref = F.leaky_relu((old + unit) / 3.0, 0.01)
    assert rel_err(out.float().permute(0, 2, 1).cpu(), ref) < 2 * BF16_TOL
    # unsupported shapes are refused loudly
    assert L_.ctta_resunit_supported(1024, 3, 1) == 0 and L_.ctta_resunit_supported(64, 4, 1) == 0 and L_.ctta_resunit_supported(512, 5, 1) == 0
    with pytest.raises(RuntimeError):
        N.check(L_.ctta_resunit_conv1d(N.ptr(xa), B, L, 1024, k, d, N.ptr(frags[0][0]), N.ptr(b1d), N.ptr(frags[1][0]),
                                       N.ptr(b2d), 0.1, N.ptr(out), 0, 1.0, 0.0, N.stream_ptr()))


@pytest.mark.parametrize("C,k,dils,L,B", [(64, 3, (1, 3, 5), 1000, 2), (64, 7, (1, 3, 5), 513, 2), (64, 7, (1, 3, 5), 128, 1),
                                          (64, 3, (1, 3, 5), 20, 3), (32, 3, (1, 3, 5), 1500, 2), (32, 7, (1, 3, 5), 1025, 1),
                                          (32, 7, (1, 3, 5), 256, 2), (32, 5, (2, 1, 4), 700, 1), (64, 5, (1, 1, 1), 129, 2),
                                          (32, 7, (1, 3, 5), 30, 1)])
def test_chained_resblock(C, k, dils, L, B):
    """hifigan/models.py:42-63, one whole ResBlock (three units, dilations `dils`) as ONE launch (resunit.hip:
    reschain_kernel): the residual stream stays in LDS between units.  Must be BIT-identical to three
    ctta_resunit_conv1d launches (same MFMA order, same bf16 rounding of the stream after every unit) -- plain, and with
    the stage-fold epilogue -- and agree with F.conv1d on bf16-rounded operands; sequences shorter than, equal to and
    ragged against the tile and its 12..36-row halo."""
    L_ = lib()
    dil_arr = (ctypes.c_int * 3)(*dils)
    assert L_.ctta_reschain_supported(C, k, dil_arr) == 1
    x = bf16_round(det("rc.x", (B, C, L), 1))
    old = bf16_round(det("rc.o", (B, C, L), 6))
    xa = x.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    units, ref = [], x
    for u, d in enumerate(dils):
        w1 = bf16_round(det("rc.w1.%d" % u, (C, C, k), 2) * (1.0 / math.sqrt(C * k)))
        w2 = bf16_round(det("rc.w2.%d" % u, (C, C, k), 3) * (1.0 / math.sqrt(C * k)))
        b1, b2 = det("rc.b1.%d" % u, (C,), 4) * 0.1, det("rc.b2.%d" % u, (C,), 5) * 0.1
        mid = bf16_round(F.leaky_relu(F.conv1d(F.leaky_relu(ref, 0.1), w1, b1, dilation=d, padding=(k * d - d) // 2), 0.1))
        ref = ref + F.conv1d(mid, w2, b2, padding=(k - 1) // 2)
        if u < 2:
            ref = bf16_round(ref)
        fr = []
        for w in (w1, w2):
            wp, k_pad = pack_conv_weight(w[:, :, None, :])
            f = torch.empty(C * k * C, dtype=torch.bfloat16, device=DEV)
            N.check(L_.ctta_frag_pack(N.ptr(wp), C, k_pad, k * C, N.ptr(f), N.stream_ptr()))
            fr.append(f)
        units.append((fr[0], b1.to(DEV), fr[1], b2.to(DEV)))
    vp = lambda ts: (ctypes.c_void_p * 3)(*[N.ptr(t) for t in ts])
    args = (dil_arr, vp([u[0] for u in units]), vp([u[1] for u in units]), vp([u[2] for u in units]), vp([u[3] for u in units]))

    def three_launches(dst, accumulate, alpha, out_slope):
        cur = xa
        for u, d in enumerate(dils):
            last = u == 2
            nxt = dst if last else torch.empty_like(xa)
            N.check(L_.ctta_resunit_conv1d(N.ptr(cur), B, L, C, k, d, N.ptr(units[u][0]), N.ptr(units[u][1]),
                                           N.ptr(units[u][2]), N.ptr(units[u][3]), 0.1, N.ptr(nxt), accumulate if last else 0,
                                           alpha if last else 1.0, out_slope if last else 0.0, N.stream_ptr()))
            cur = nxt
        sync()
        return dst

    out = torch.empty_like(xa)
    N.check(L_.ctta_reschain_conv1d(N.ptr(xa), B, L, C, k, *args, 0.1, N.ptr(out), 0, 1.0, 0.0, N.stream_ptr()))
    sync()
    assert rel_err(out.float().permute(0, 2, 1).cpu(), ref) < 3 * BF16_TOL
    assert torch.equal(out, three_launches(torch.empty_like(xa), 0, 1.0, 0.0))
    # stage fold: (old + block) / 3 then leaky_relu(0.1), accumulated in place
    o1 = old.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    o2 = o1.clone()
    N.check(L_.ctta_reschain_conv1d(N.ptr(xa), B, L, C, k, *args, 0.1, N.ptr(o1), 1, 1.0 / 3.0, 0.1, N.stream_ptr()))
    sync()
    assert torch.equal(o1, three_launches(o2, 1, 1.0 / 3.0, 0.1))
    assert rel_err(o1.float().permute(0, 2, 1).cpu(), F.leaky_relu((old + ref) / 3.0, 0.1)) < 3 * BF16_TOL
    # outside the kernel's range: refused loudly
    assert L_.ctta_reschain_supported(128, 3, dil_arr) == 0 and L_.ctta_reschain_supported(64, 11, dil_arr) == 0
    with pytest.raises(RuntimeError):
        N.check(L_.ctta_reschain_conv1d(N.ptr(xa), B, L, C, k, *args, 0.1, N.ptr(xa), 0, 1.0, 0.0, N.stream_ptr()))


@pytest.mark.parametrize("with_res", [True, False])
@pytest.mark.parametrize("B,C,H,W,Cout,tile", [(2, 64, 16, 16, 128, 0), (3, 128, 32, 8, 256, 0), (4, 256, 64, 64, 512, 0),
                                               (1, 64, 64, 64, 128, 0), (2, 64, 16, 16, 128, 17), (2, 128, 16, 16, 256, 29),
                                               (2, 128, 64, 16, 1024, 29), (2, 64, 64, 32, 128, 28), (2, 64, 64, 32, 128, 36)])
def test_groupnorm_statistics_from_the_conv_epilogue(B, C, H, W, Cout, tile, with_res):
    """ctta_conv_desc.gn_part: the convolution's wide-store epilogue writes per-tile (sum, sum of squares) of its OUTPUT per
    channel group; ctta_groupnorm_from_partials then normalises without its own statistics pass.  Must agree with the
    three-pass GroupNorm on the same tensor (same bf16 values, only the fp32 summation order differs) and with the oracle.
    Round 3: the statistics ride in the STRAIGHT-LINE epilogue (wide_epilogue_fast<..., GN>), with and without a residual
    (two instantiations), for groups narrower than, equal to and wider than a lane's 4 channels / a wave's 64 columns; the
    finalize + apply pair also runs as ONE launch (gn_apply_fused_kernel) when no (mean, rstd) table is asked for."""
    L_ = lib()
    G = 32
    x = bf16_round(det("gnf.x", (B, C, H, W), 1))
    w = bf16_round(det("gnf.w", (Cout, C, 3, 3), 2) * (1.0 / math.sqrt(C * 9)))
    b = det("gnf.b", (Cout,), 3) * 0.1
    res = bf16_round(det("gnf.r", (B, Cout, H, W), 4))
    gamma, beta = 1.0 + 0.2 * det("gnf.g", (Cout,), 5), 0.1 * det("gnf.be", (Cout,), 6)
    wp, k_pad = pack_conv_weight(w)
    out = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=DEV)
    part = torch.full((B * (H * W // 16 + 1) * G * 2,), float("nan"), dtype=torch.float32, device=DEV)
    xd, bdev, resd = nhwc_bf16(x), b.to(DEV), nhwc_bf16(res)
    d = conv_desc(x0=xd, c0=C, batch=B, hi=H, wi=W, ho=H, wo=W, kh=3, kw=3, pad_h=1, pad_w=1, w=wp, k_pad=k_pad,
                  n=Cout, bias=bdev, out=out, ldc=Cout, tile=tile, gn_part=part,
                  gn_groups=G, gn_hw=H * W, gn_part_floats=part.numel(), **(dict(res=resd, res_ld=Cout) if with_res else {}))
    run_conv(d)
    chunks = L_.ctta_conv_last_gn_chunks()
    assert chunks >= 1 and (H * W) % chunks == 0, "this launch should have produced GroupNorm partials"
    assert bool(torch.isfinite(part[:B * chunks * G * 2]).all())
    gd, bd = gamma.to(DEV), beta.to(DEV)
    scratch = torch.empty(L_.ctta_groupnorm_scratch_floats(B, H * W, Cout, G) + B * 2 * Cout, dtype=torch.float32, device=DEV)
    y_f, y_s = torch.empty_like(out), torch.empty_like(out)
    st_f, st_s = (torch.empty(B, G, 2, dtype=torch.float32, device=DEV) for _ in range(2))
    N.check(L_.ctta_groupnorm_from_partials(N.ptr(out), N.ptr(y_f), B, H * W, Cout, G, N.ptr(gd), N.ptr(bd),
                                            1e-5, 1, N.ptr(part), chunks, N.ptr(scratch), N.ptr(st_f), N.stream_ptr()))
    N.check(L_.ctta_groupnorm_stats_out(N.ptr(out), N.ptr(y_s), B, H * W, Cout, G, N.ptr(gd), N.ptr(bd),
                                        1e-5, 1, N.ptr(scratch), N.ptr(st_s), N.stream_ptr()))
    sync()
    # (mean, rstd): the epilogue sums the fp32 values BEFORE their bf16 rounding (what the reference's fp32 GroupNorm sees),
    # the three-pass kernel the stored bf16 values: zero-mean rounding noise, 2^-9 relative per element
    # -> the means agree to a thousandth of a standard deviation, the reciprocal deviations to 1e-3 (1024-element groups in the smallest case)
    assert float(((st_f[..., 0] - st_s[..., 0]) * st_s[..., 1]).abs().max()) < 1e-3 and rel_err(st_f[..., 1], st_s[..., 1]) < 1e-3
    assert rel_err(y_f.float(), y_s.float()) <= 2.0 ** -7                                                 # at most an ulp apart
    # no statistics table -> one launch folds the partials in every block: bit-identical to the two-launch form
    y_1 = torch.empty_like(out)
    N.check(L_.ctta_groupnorm_from_partials(N.ptr(out), N.ptr(y_1), B, H * W, Cout, G, N.ptr(gd), N.ptr(bd),
                                            1e-5, 1, N.ptr(part), chunks, N.ptr(scratch), None, N.stream_ptr()))
    sync()
    assert torch.equal(y_1, y_f)
    ref = F.silu(F.group_norm(from_nhwc(out), G, gamma, beta, 1e-5))
    assert rel_err(from_nhwc(y_f), ref) < BF16_TOL
    # launches that cannot provide the statistics say so instead of writing garbage: split-K (deep and narrow) ...
    x2, w2 = nhwc_bf16(bf16_round(det("gnf.x2", (1, 1024, 4, 4), 7))), pack_conv_weight(bf16_round(det("gnf.w2", (256, 1024, 3, 3), 8) * 0.01))[0]
    o2 = torch.empty(1, 4, 4, 256, dtype=torch.bfloat16, device=DEV)
    d2 = conv_desc(x0=x2, c0=1024, batch=1, hi=4, wi=4, ho=4, wo=4, kh=3, kw=3, pad_h=1, pad_w=1, w=w2, k_pad=9216, n=256,
                   out=o2, ldc=256, gn_part=part, gn_groups=G, gn_hw=16, gn_part_floats=part.numel())
    run_conv(d2)
    assert L_.ctta_conv_last_gn_chunks() == 0


@pytest.mark.parametrize("with_res", [False, True])
def test_ragged_last_row_tile_runs_as_a_second_launch(with_res):
    """M = 256 * 256 + 40 rows, one column tile: 257 workgroups of the 256x256 tile = two rounds of the 256 CUs for 40 rows.
    The host cuts the last row tile off and runs it with 64x64 tiles through ConvParams.m_off (round 3; HiFi-GAN's
    M = 32 * 5121).  Same numbers as one launch with a forced tile, and as F.conv1d."""
    B, C, Co, L, k = 1, 64, 256, 256 * 256 + 40, 11
    x = bf16_round(det("tail.x", (B, C, L), 1))
    w = bf16_round(det("tail.w", (Co, C, k), 2) * (1.0 / math.sqrt(C * k)))
    b = det("tail.b", (Co,), 3) * 0.1
    res = bf16_round(det("tail.r", (B, Co, L), 4))
    ref = F.conv1d(x, w, b, padding=k // 2) + (res if with_res else 0.0)
    wp, k_pad = pack_conv_weight(w[:, :, None, :])
    xa = x.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    ra = res.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    bd = b.to(DEV)
    outs = []
    for tile in (0, 29):
        out = torch.full((B, L, Co), float("nan"), dtype=torch.bfloat16, device=DEV)
        run_conv(conv_desc(x0=xa, c0=C, batch=B, hi=1, wi=L, ho=1, wo=L, kh=1, kw=k, pad_w=k // 2, w=wp, k_pad=k_pad, n=Co,
                           bias=bd, out=out, ldc=Co, tile=tile, **(dict(res=ra, res_ld=Co) if with_res else {})))
        assert bool(torch.isfinite(out.float()).all())
        outs.append(out.float().permute(0, 2, 1).cpu())
    assert rel_err(outs[0], ref) < BF16_TOL and rel_err(outs[1], ref) < BF16_TOL
    assert torch.equal(outs[0][:, :, :256 * 256], outs[1][:, :, :256 * 256])     # the big tiles are the same launch
    assert rel_err(outs[0][:, :, 256 * 256:], outs[1][:, :, 256 * 256:]) <= 2.0 ** -7


def test_conv_fused_output_activations_and_scalar_store():
    """out_act=leaky_relu, the second (activated) output, and element-wise stores for Cout=1."""
    B, C, L, k = 2, 64, 150, 3
    x = bf16_round(det("oa.x", (B, C, L), 1))
    w = bf16_round(det("oa.w", (C, C, k), 2) * (1.0 / math.sqrt(C * k)))
    b = det("oa.b", (C,), 3) * 0.1
    ref = F.conv1d(x, w, b, padding=1)
    wp, k_pad = pack_conv_weight(w[:, :, None, :])
    xa = x.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    bd = b.to(DEV)
    out = torch.empty(B, L, C, dtype=torch.bfloat16, device=DEV)
    out2 = torch.empty(B, L, C, dtype=torch.bfloat16, device=DEV)
    run_conv(conv_desc(x0=xa, c0=C, batch=B, hi=1, wi=L, ho=1, wo=L, kh=1, kw=k, pad_w=1, w=wp, k_pad=k_pad, n=C,
                       bias=bd, out=out, ldc=C, out2=out2, out2_slope=0.1))
    o = out.to(torch.float32).permute(0, 2, 1).cpu()
    assert rel_err(o, ref) < BF16_TOL
    assert torch.equal(out2.to(torch.float32).cpu(), bf16_round(F.leaky_relu(out.to(torch.float32).cpu(), 0.1)))
    run_conv(conv_desc(x0=xa, c0=C, batch=B, hi=1, wi=L, ho=1, wo=L, kh=1, kw=k, pad_w=1, w=wp, k_pad=k_pad, n=C,
                       bias=bd, out=out, ldc=C, out_act=3, out_slope=0.01))
    assert rel_err(out.to(torch.float32).permute(0, 2, 1).cpu(), F.leaky_relu(ref, 0.01)) < BF16_TOL
    # Cout = 1, fp32 planar output (the VAE's conv_out): ldc = 1 -> element-wise stores
    x2 = bf16_round(det("oa.x2", (B, 32, 12, 10), 4))
    w2 = bf16_round(det("oa.w2", (1, 32, 3, 3), 5) * 0.1)
    b2 = det("oa.b2", (1,), 6)
    ref2 = F.conv2d(x2, w2, b2, padding=1)
    wp2, kp2 = pack_conv_weight(w2)
    x2a, b2d = nhwc_bf16(x2), b2.to(DEV)
    o2 = torch.empty(B, 1, 12, 10, dtype=torch.float32, device=DEV)
    run_conv(conv_desc(x0=x2a, c0=32, batch=B, hi=12, wi=10, ho=12, wo=10, kh=3, kw=3, pad_h=1, pad_w=1, w=wp2,
                       k_pad=kp2, n=1, bias=b2d, out=o2, ldc=1, out_f32=1))
    assert rel_err(o2.cpu(), ref2) < 1e-4


@pytest.mark.parametrize("k,u", [(16, 5), (16, 4), (8, 2), (4, 2)])
def test_conv_transpose1d_as_phase_gemm(k, u):
    """ConvTranspose1d(k, stride u, padding (k-u)//2) == u phase convolutions written through
    the output remap (engine_vae.hip make_convt1d); weights packed here with the same maps."""
    B, Cin, Cout, L = 2, 64, 32, 37
    pad = (k - u) // 2
    x = bf16_round(det("ct.x", (B, Cin, L), 1))
    w = bf16_round(det("ct.w", (Cin, Cout, k), 2) * (1.0 / math.sqrt(Cin * k / u)))
    b = det("ct.b", (Cout,), 3) * 0.1
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=u, padding=pad)
    Lout = ref.shape[2]
    taps = (k + u - 1) // u
    K = taps * Cin
    k_pad = (K + 63) // 64 * 64
    n = u * Cout
    ro = torch.tensor([o * k + r for r in range(u) for o in range(Cout)], dtype=torch.int32)
    ra = torch.tensor([r for r in range(u) for o in range(Cout)], dtype=torch.int32)
    co = torch.full((k_pad,), -1, dtype=torch.int32)
    ca = torch.zeros(k_pad, dtype=torch.int32)
    for t in range(taps):
        m = taps - 1 - t
        for c in range(Cin):
            co[t * Cin + c] = c * Cout * k + m * u
            ca[t * Cin + c] = m * u
    wd = w.contiguous().to(DEV)
    wp = torch.empty(n, k_pad, dtype=torch.bfloat16, device=DEV)
    ro, co, ra, ca = ro.to(DEV), co.to(DEV), ra.to(DEV), ca.to(DEV)
    N.check(lib().ctta_pack_weight(N.ptr(wd), N.ptr(ro), N.ptr(co), N.ptr(ra), N.ptr(ca), k, n, k_pad,
                                   N.ptr(wp), N.stream_ptr()))
    bias = b.repeat(u).contiguous().to(DEV)
    Q = (Lout - 1 + pad) // u + 1
    out = torch.full((B, Lout, Cout), 7.0, dtype=torch.bfloat16, device=DEV)
    xa = x.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    run_conv(conv_desc(x0=xa, c0=Cin, batch=B, hi=1, wi=L, ho=1, wo=Q, kh=1, kw=taps, pad_w=taps - 1, w=wp,
                       k_pad=k_pad, n=n, bias=bias, in_act=1, in_slope=0.1, out=out, ldc=u * Cout,
                       out_batch_stride=Lout * Cout, out_offset=-pad * Cout, out_limit=Lout * Cout))
    got = out.to(torch.float32).permute(0, 2, 1).cpu()
    assert rel_err(got, ref) < BF16_TOL


@pytest.mark.parametrize("tile", [0, 17, 29, 31, 36])
@pytest.mark.parametrize("with_out2", [False, True])
def test_conv_transpose1d_rows_leave_through_the_straight_line_epilogue(tile, with_out2):
    """The HiFi-GAN upsamplers at a size where the 128x64-per-wave tiles run (B = 3, L = 333 -> 1332, 128 -> 64 channels, k = 8,
    u = 4: one GEMM row = 4 output positions x 64 channels, per-sample stride, shifted by the padding, clipped at both ends).
    Since round 6 a wave whose rows lie in one sample stores through ONE buffer descriptor over that sample -- the bounds check
    does the clipping, no divergent `if (row < M)` around a store -- with or without the second (LeakyReLU'd) output; waves that
    straddle two samples (Q = 334 rows per sample against 64..128-row wave blocks) take the generic loop.  Both must give
    conv_transpose1d, and the poisoned bytes around every sample must stay untouched."""
    B, Cin, Cout, L, k, u = 3, 128, 64, 333, 8, 4
    pad = (k - u) // 2
    x = bf16_round(det("ctf.x", (B, Cin, L), 1))
    w = bf16_round(det("ctf.w", (Cin, Cout, k), 2) * (1.0 / math.sqrt(Cin * k / u)))
    b = det("ctf.b", (Cout,), 3) * 0.1
    ref = F.conv_transpose1d(x, w, b, stride=u, padding=pad)
    Lout = ref.shape[2]
    taps = (k + u - 1) // u
    K = taps * Cin
    k_pad = (K + 63) // 64 * 64
    n = u * Cout
    ro = torch.tensor([o * k + r for r in range(u) for o in range(Cout)], dtype=torch.int32)
    ra = torch.tensor([r for r in range(u) for o in range(Cout)], dtype=torch.int32)
    co = torch.full((k_pad,), -1, dtype=torch.int32)
    ca = torch.zeros(k_pad, dtype=torch.int32)
    for t in range(taps):
        m = taps - 1 - t
        for c in range(Cin):
            co[t * Cin + c] = c * Cout * k + m * u
            ca[t * Cin + c] = m * u
    wd = w.contiguous().to(DEV)
    wp = torch.empty(n, k_pad, dtype=torch.bfloat16, device=DEV)
    ro, co, ra, ca = ro.to(DEV), co.to(DEV), ra.to(DEV), ca.to(DEV)
    N.check(lib().ctta_pack_weight(N.ptr(wd), N.ptr(ro), N.ptr(co), N.ptr(ra), N.ptr(ca), k, n, k_pad, N.ptr(wp), N.stream_ptr()))
    bias = b.repeat(u).contiguous().to(DEV)
    Q = (Lout - 1 + pad) // u + 1
    guard = 256                                       # poisoned elements in front of / behind the whole output
    buf = torch.full((guard + B * Lout * Cout + guard,), 7.0, dtype=torch.bfloat16, device=DEV)
    buf2 = torch.full_like(buf, 9.0)
    out, out2 = buf[guard:guard + B * Lout * Cout], buf2[guard:guard + B * Lout * Cout]
    xa = x.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)
    kw = dict(x0=xa, c0=Cin, batch=B, hi=1, wi=L, ho=1, wo=Q, kh=1, kw=taps, pad_w=taps - 1, w=wp, k_pad=k_pad, n=n, bias=bias,
              out=out, ldc=u * Cout, out_batch_stride=Lout * Cout, out_offset=-pad * Cout, out_limit=Lout * Cout, tile=tile)
    if with_out2:
        kw.update(out2=out2, out2_slope=0.1)
    run_conv(conv_desc(**kw))
    got = out.view(B, Lout, Cout).to(torch.float32).permute(0, 2, 1).cpu()
    assert rel_err(got, ref) < BF16_TOL
    assert bool((buf[:guard] == 7.0).all()) and bool((buf[-guard:] == 7.0).all())
    if with_out2:
        got2 = out2.view(B, Lout, Cout).to(torch.float32).permute(0, 2, 1).cpu()
        assert torch.equal(got2, bf16_round(F.leaky_relu(got, 0.1)))      # leaky_relu of the SAME (bf16-rounded) values
        assert bool((buf2[:guard] == 9.0).all()) and bool((buf2[-guard:] == 9.0).all())


def test_batched_gemm_f32_and_transposed_product():
    """q k^T per batch with fp32 output (VAE AttnBlock scores) and V^T = Wv X^T + bv[row]."""
    B, Nt, C = 2, 128, 64
    q = bf16_round(det("bg.q", (B, Nt, C), 1))
    kk = bf16_round(det("bg.k", (B, Nt, C), 2))
    qd, kd = q.to(torch.bfloat16).to(DEV), kk.to(torch.bfloat16).to(DEV)
    s = torch.empty(B, Nt, Nt, dtype=torch.float32, device=DEV)
    run_conv(conv_desc(x0=qd, c0=C, batch=1, hi=Nt, wi=1, ho=Nt, wo=1, w=kd, k_pad=C, n=Nt, out=s, ldc=Nt,
                       out_f32=1, groups=B, x_group_stride=Nt * C, w_group_stride=Nt * C,
                       out_group_stride=Nt * Nt))
    assert rel_err(s.cpu(), torch.bmm(q, kk.transpose(1, 2))) < 1e-4
    # transposed product with odd token count (n not a multiple of 4) and per-row bias
    T = 6
    wv = bf16_round(det("bg.wv", (C, C), 3) * 0.2)
    bv = det("bg.bv", (C,), 4)
    x = bf16_round(det("bg.x", (B, T, C), 5))
    wvd, xd, bvd = wv.to(torch.bfloat16).to(DEV), x.to(torch.bfloat16).to(DEV), bv.to(DEV)
    ld = 8
    vt = torch.zeros(B, C, ld, dtype=torch.bfloat16, device=DEV)
    run_conv(conv_desc(x0=wvd, c0=C, batch=1, hi=C, wi=1, ho=C, wo=1, w=xd, k_pad=C, n=T, bias_m=bvd, out=vt,
                       ldc=ld, groups=B, w_group_stride=T * C, out_group_stride=C * ld))
    ref = torch.einsum("ck,btk->bct", wv, x) + bv[None, :, None]
    assert rel_err(vt.to(torch.float32).cpu()[:, :, :T], ref) < BF16_TOL


def test_conv_small_n():
    for (C, n, kh, kw_, H, W, act_in, act_out) in [(40, 8, 3, 3, 16, 8, 0, 0), (32, 1, 3, 3, 12, 6, 0, 0),
                                                   (32, 1, 1, 7, 1, 300, 1, 2)]:
        B = 2
        x = bf16_round(det("sn.x", (B, C, H, W), 1))
        w = det("sn.w", (n, C, kh, kw_), 2) * (1.0 / math.sqrt(C * kh * kw_))
        b = det("sn.b", (n,), 3) * 0.1
        xin = F.leaky_relu(x, 0.01) if act_in else x
        ref = F.conv2d(xin, w, b, padding=(kh // 2, kw_ // 2))
        if act_out == 2:
            ref = torch.tanh(ref)
        wr = w.permute(0, 2, 3, 1).contiguous().to(DEV)
        out = torch.empty(B, n, H, W, dtype=torch.float32, device=DEV)
        xa, bd = nhwc_bf16(x), b.to(DEV)
        N.check(lib().ctta_conv_small_n(N.ptr(xa), C, B, H, W, kh, kw_, kh // 2, kw_ // 2, N.ptr(wr), N.ptr(bd), n,
                                        act_in, 0.01, act_out, N.ptr(out), None, N.stream_ptr()))
        sync()
        assert rel_err(out.cpu(), ref) < 1e-5


# ------------------------------------------------------------------------------------ norms etc.
@pytest.mark.parametrize("B,C,H,W,G,silu,eps", [(2, 40, 32, 8, 8, True, 1e-5), (2, 256, 16, 16, 32, False, 1e-6),
                                                (1, 120, 8, 4, 8, True, 1e-5), (3, 2048, 4, 2, 32, True, 1e-5),
                                                (1, 128, 64, 64, 32, True, 1e-6), (2, 32, 9, 5, 32, False, 1e-6)])
def test_groupnorm(B, C, H, W, G, silu, eps):
    x = bf16_round(det("gn.x", (B, C, H, W), 1) * 2 + 0.3)
    gamma = 1 + 0.2 * det("gn.g", (C,), 2)
    beta = 0.1 * det("gn.b", (C,), 3)
    ref = F.group_norm(x, G, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    xa = nhwc_bf16(x)
    y = torch.empty_like(xa)
    scratch = torch.empty(lib().ctta_groupnorm_scratch_floats(B, H * W, C, G) + 16, dtype=torch.float32, device=DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    N.check(lib().ctta_groupnorm(N.ptr(xa), N.ptr(y), B, H * W, C, G, N.ptr(gd), N.ptr(bd), eps, int(silu),
                                 N.ptr(scratch), N.stream_ptr()))
    sync()
    assert rel_err(from_nhwc(y), ref) < BF16_TOL


@pytest.mark.parametrize("B,C1,C2,H,W,G,stats", [(2, 256, 256, 64, 16, 32, False), (3, 1024, 512, 16, 4, 32, False),
                                                 (2, 512, 256, 32, 8, 32, True), (1, 1024, 1024, 8, 2, 32, False),
                                                 (2, 40, 24, 9, 5, 8, False)])
def test_concat_with_groupnorm_statistics(B, C1, C2, H, W, G, stats):
    """ctta_concat_channels_gn: torch.cat([h, skip], 1) (unet_2d_blocks.py:2053) that also sums the result for the resnet's
    norm1: the concatenation is exact, and ctta_groupnorm_from_partials on its partials is BIT-identical to
    ctta_groupnorm on the concatenated tensor (same chunking and order) -- group boundaries that straddle the two
    sources (1024 + 512 in 32 groups) included."""
    L_ = lib()
    s = N.stream_ptr()
    C, hw = C1 + C2, H * W
    a = nhwc_bf16(bf16_round(det("cg.a", (B, C1, H, W), 1) * 2 + 0.3))
    b = nhwc_bf16(bf16_round(det("cg.b", (B, C2, H, W), 2) - 0.2))
    gd, bd = (1 + 0.2 * det("cg.g", (C,), 3)).to(DEV), (0.1 * det("cg.be", (C,), 4)).to(DEV)
    cat = torch.empty(B, H, W, C, dtype=torch.bfloat16, device=DEV)
    part = torch.full((B * (hw // 16 + 1) * G * 2,), float("nan"), device=DEV)
    nchunk = ctypes.c_int(0)
    N.check(L_.ctta_concat_channels_gn(N.ptr(a), C1, N.ptr(b), C2, N.ptr(cat), B, hw, G, N.ptr(part), part.numel(),
                                       ctypes.byref(nchunk), s))
    sync()
    assert torch.equal(cat, torch.cat([a, b], dim=3)) and nchunk.value >= 1
    scratch = torch.empty(L_.ctta_groupnorm_scratch_floats(B, hw, C, G) + 16, dtype=torch.float32, device=DEV)
    y_ref, y = torch.empty_like(cat), torch.empty_like(cat)
    st_ref, st = torch.zeros(B, G, 2, device=DEV), torch.zeros(B, G, 2, device=DEV)
    N.check(L_.ctta_groupnorm_stats_out(N.ptr(cat), N.ptr(y_ref), B, hw, C, G, N.ptr(gd), N.ptr(bd), 1e-5, 1, N.ptr(scratch),
                                        N.ptr(st_ref) if stats else None, s))
    scratch2 = torch.empty(B * 2 * C + 16, device=DEV)
    N.check(L_.ctta_groupnorm_from_partials(N.ptr(cat), N.ptr(y), B, hw, C, G, N.ptr(gd), N.ptr(bd), 1e-5, 1, N.ptr(part),
                                            nchunk.value, N.ptr(scratch2), N.ptr(st) if stats else None, s))
    sync()
    ref = F.silu(F.group_norm(from_nhwc(cat), G, gd.cpu(), bd.cpu(), 1e-5))
    assert rel_err(from_nhwc(y), ref) < BF16_TOL
    if hw * (C // G) > 16384:          # below that ctta_groupnorm takes its single-launch form (another summation order)
        assert torch.equal(y, y_ref) and torch.equal(st, st_ref)
    else:
        assert rel_err(y.float().cpu(), y_ref.float().cpu()) < 2.0 ** -7
    with pytest.raises(RuntimeError):   # a partials buffer that cannot hold the chunks is refused
        N.check(L_.ctta_concat_channels_gn(N.ptr(a), C1, N.ptr(b), C2, N.ptr(cat), B, hw, G, N.ptr(part), 2, ctypes.byref(nchunk), s))


@pytest.mark.parametrize("rows,d,ld", [(37, 39, 64), (64, 255, 256), (16, 1020, 1024), (5, 1275, 1280), (131, 255, 256),
                                        (70, 510, 512), (9, 1020, 1024), (33, 512, 512)])
def test_layernorm(rows, d, ld):
    x = torch.zeros(rows, ld)
    x[:, :d] = bf16_round(det("ln.x", (rows, d), 1) * 3 + 0.5)
    gamma = 1 + 0.2 * det("ln.g", (d,), 2)
    beta = 0.1 * det("ln.b", (d,), 3)
    ref = F.layer_norm(x[:, :d], (d,), gamma, beta, 1e-5)
    xd = x.to(torch.bfloat16).to(DEV)
    y = torch.full((rows, ld), 9.0, dtype=torch.bfloat16, device=DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    N.check(lib().ctta_layernorm(N.ptr(xd), N.ptr(y), rows, d, ld, N.ptr(gd), N.ptr(bd), 1e-5, N.stream_ptr()))
    sync()
    yy = y.to(torch.float32).cpu()
    assert rel_err(yy[:, :d], ref) < BF16_TOL
    assert float(yy[:, d:].abs().max()) == 0.0 if d < ld else True   # pad columns are zeroed


def test_geglu_and_softmax():
    rows, hp = 33, 128
    x = bf16_round(det("gg.x", (rows, 2 * hp), 1) * 3)
    ref = x[:, :hp] * F.gelu(x[:, hp:])
    xd = x.to(torch.bfloat16).to(DEV)
    y = torch.empty(rows, hp, dtype=torch.bfloat16, device=DEV)
    N.check(lib().ctta_geglu(N.ptr(xd), N.ptr(y), rows, hp, 0, N.stream_ptr()))
    sync()
    assert rel_err(y.to(torch.float32).cpu(), ref) < BF16_TOL
    # the engine's layout: 16-column blocks [16 value][16 gate]
    xi = torch.empty_like(x)
    for j in range(hp):
        xi[:, (j // 16) * 32 + j % 16] = x[:, j]
        xi[:, (j // 16) * 32 + 16 + j % 16] = x[:, hp + j]
    N.check(lib().ctta_geglu(N.ptr(xi.to(torch.bfloat16).to(DEV)), N.ptr(y), rows, hp, 1, N.stream_ptr()))
    sync()
    assert rel_err(y.to(torch.float32).cpu(), ref) < BF16_TOL
    s = det("sm.s", (19, 4096), 2) * 30
    sd = s.to(DEV)
    p = torch.empty(19, 4096, dtype=torch.bfloat16, device=DEV)
    N.check(lib().ctta_softmax_rows(N.ptr(sd), N.ptr(p), 19, 4096, 0.044, N.stream_ptr()))
    sync()
    assert rel_err(p.to(torch.float32).cpu(), torch.softmax(s * 0.044, dim=-1)) < BF16_TOL


@pytest.mark.parametrize("B,heads,dh,nq,nk,masked", [(2, 3, 13, 256, 256, False), (2, 5, 51, 200, 200, False),
                                                     (2, 6, 13, 4, 4, False), (1, 2, 64, 130, 70, False),
                                                     (2, 3, 13, 64, 7, True), (3, 5, 51, 300, 32, True)])
def test_attention(B, heads, dh, nq, nk, masked):
    """Flash attention with head dim padded to 64 vs softmax(q k^T / sqrt(dh) + bias) v."""
    q = bf16_round(det("at.q", (B, heads, nq, dh), 1))
    k = bf16_round(det("at.k", (B, heads, nk, dh), 2))
    v = bf16_round(det("at.v", (B, heads, nk, dh), 3))
    bias = None
    if masked:
        lens = [nk, max(1, nk // 2), max(1, nk - 3)][:B]
        keep = torch.arange(nk)[None, :] < torch.tensor(lens)[:, None]
        bias = (1 - keep.float()) * -10000.0
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
    if bias is not None:
        s = s + bias[:, None, None, :]
    ref = torch.matmul(torch.softmax(s, dim=-1), v)          # (B, heads, nq, dh)
    hp = heads * 64
    k_rows = nk + 5                                           # k buffer taller than nk
    vt_ld = (nk + 7) // 8 * 8
    qd = torch.zeros(B, nq, hp)
    kd = torch.zeros(B, k_rows, hp)
    vtd = torch.full((B, hp, vt_ld), float("nan"))            # padding must be ignored, even NaN
    for h in range(heads):
        qd[:, :, h * 64:h * 64 + dh] = q[:, h]
        kd[:, :nk, h * 64:h * 64 + dh] = k[:, h]
        vtd[:, h * 64:h * 64 + 64, :nk] = 0
        vtd[:, h * 64:h * 64 + dh, :nk] = v[:, h].transpose(1, 2)
    qd, kd, vtd = (t.to(torch.bfloat16).to(DEV) for t in (qd, kd, vtd))
    out = torch.zeros(B, nq, hp, dtype=torch.bfloat16, device=DEV)
    bd = bias.contiguous().to(DEV) if bias is not None else None
    N.check(lib().ctta_attention(N.ptr(qd), hp, N.ptr(kd), hp, k_rows, N.ptr(vtd), vt_ld, N.ptr(bd), N.ptr(out), hp,
                                 B, heads, nq, nk, 1.0 / math.sqrt(dh), N.stream_ptr()))
    sync()
    o = out.to(torch.float32).cpu().reshape(B, nq, heads, 64)[..., :dh].permute(0, 2, 1, 3)
    # P is rounded to bf16 before the PV product: 2^-8 relative on each probability
    assert rel_err(o, ref) < 2.5 * BF16_TOL


def test_small_fp32_kernels():
    B, K, Nn = 5, 96, 70
    x, w, b = det("l.x", (B, K), 1), det("l.w", (Nn, K), 2) * 0.2, det("l.b", (Nn,), 3)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y = torch.empty(B, Nn, device=DEV)
    N.check(lib().ctta_linear_f32(N.ptr(xd), N.ptr(wd), N.ptr(bd), N.ptr(y), B, Nn, K, 0, 1, N.stream_ptr()))
    sync()
    assert rel_err(y.cpu(), F.silu(F.linear(x, w, b))) < 1e-5
    # sinusoid + Fourier features
    t = torch.tensor([999.0, 0.0, 58.7647, 470.1176, 940.2353])
    dim = 40
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / half
    freqs = torch.exp(exponent)
    ref = onets.timestep_embedding(t, dim, True, 0.0)
    td, fd = t.to(DEV), freqs.to(DEV)
    o = torch.empty(5, dim, device=DEV)
    N.check(lib().ctta_time_features(N.ptr(td), N.ptr(fd), dim, 1, N.ptr(o), 5, N.stream_ptr()))
    sync()
    assert float((o.cpu() - ref).abs().max()) < 2e-4   # sin/cos of arguments up to 999 in fp32
    wv = torch.tensor([4.0, 0.3, 5.9], dtype=torch.float64)
    W = det("f.w", (24,), 4) * 1.7
    ref = onets.fourier_embedding(wv, W.double(), True).float()
    wd_, Wd = wv.to(DEV), W.to(DEV)
    o = torch.empty(3, 48, device=DEV)
    N.check(lib().ctta_fourier_features(N.ptr(wd_), N.ptr(Wd), 24, 1, N.ptr(o), 3, N.stream_ptr()))
    sync()
    assert float((o.cpu() - ref).abs().max()) < 1e-6


def test_layout_packs():
    x = det("lp.x", (2, 8, 6, 5), 1)
    xd = x.to(DEV)
    y = torch.empty(2, 6, 5, 32, dtype=torch.bfloat16, device=DEV)
    N.check(lib().ctta_nchw_f32_to_nhwc_bf16(N.ptr(xd), N.ptr(y), 2, 8, 6, 5, 32, 0.5, N.stream_ptr()))
    sync()
    yy = y.to(torch.float32).cpu()
    assert torch.equal(yy[..., :8], bf16_round(x * 0.5).permute(0, 2, 3, 1))
    assert float(yy[..., 8:].abs().max()) == 0.0
    z = torch.empty(2, 8, 6, 5, device=DEV)
    N.check(lib().ctta_nhwc_bf16_to_nchw_f32(N.ptr(y), N.ptr(z), 2, 8, 6, 5, 32, N.stream_ptr()))
    sync()
    assert torch.equal(z.cpu(), bf16_round(x * 0.5))
    a = torch.arange(2 * 3 * 16, dtype=torch.float32).reshape(6, 16).to(torch.bfloat16).to(DEV)
    b = (torch.arange(6 * 8, dtype=torch.float32).reshape(6, 8) + 500).to(torch.bfloat16).to(DEV)
    c = torch.empty(6, 24, dtype=torch.bfloat16, device=DEV)
    N.check(lib().ctta_concat_channels(N.ptr(a), 16, N.ptr(b), 8, N.ptr(c), 6, N.stream_ptr()))
    sync()
    assert torch.equal(c.cpu(), torch.cat([a.cpu(), b.cpu()], dim=1))


# ------------------------------------------------------------------------------------ fp32 solver kernels
def test_heun_cfg_loss_ema_wav(golden):
    g = golden("heun")
    _, sig = oheun.set_timesteps(18)
    sig = torch.from_numpy(sig)
    idx = torch.from_numpy(g["idx"])
    B, n = 3, 8 * 16 * 4
    x = det("heun.x", (B, 8, 16, 4), 1) * 3
    v1, v2, noise = det("heun.v1", (B, 8, 16, 4), 2), det("heun.v2", (B, 8, 16, 4), 3), det("heun.n", (B, 8, 16, 4), 4)
    xd, v1d, v2d, nd = x.to(DEV), v1.to(DEV), v2.to(DEV), noise.to(DEV)
    s0, s1 = sig[idx].contiguous().to(DEV), sig[idx + 1].contiguous().to(DEV)
    L = lib()
    st = N.stream_ptr()
    out = torch.empty_like(xd)
    N.check(L.ctta_heun_scale_model_input(N.ptr(xd), N.ptr(s0), N.ptr(out), B, n, st)); sync()
    assert rel_err(out.cpu(), torch.from_numpy(g["scaled"])) < F32_TOL
    N.check(L.ctta_heun_add_noise(N.ptr(xd), N.ptr(nd), N.ptr(s0), N.ptr(out), B, n, st)); sync()
    assert rel_err(out.cpu(), torch.from_numpy(g["noised"])) < F32_TOL
    prev, deriv = torch.empty_like(xd), torch.empty_like(xd)
    N.check(L.ctta_heun_step_first(N.ptr(v1d), N.ptr(xd), N.ptr(s0), N.ptr(s1), N.ptr(prev), N.ptr(deriv), B, n, st)); sync()
    assert rel_err(prev.cpu(), torch.from_numpy(g["step1"])) < F32_TOL
    prev2 = torch.empty_like(xd)
    N.check(L.ctta_heun_step_second(N.ptr(v2d), N.ptr(prev), N.ptr(xd), N.ptr(deriv), N.ptr(s0), N.ptr(s1),
                                    N.ptr(prev2), B, n, st)); sync()
    assert rel_err(prev2.cpu(), torch.from_numpy(g["step2"])) < 4 * F32_TOL
    # CFG combine
    w = torch.tensor([0.0, 3.0, 5.5])
    wd = w.to(DEV)
    N.check(L.ctta_cfg_combine(N.ptr(v1d), N.ptr(v2d), N.ptr(wd), N.ptr(out), B, n, st)); sync()
    assert rel_err(out.cpu(), oheun.cfg_combine(v1, v2, w)) < F32_TOL
    # SNR-weighted instance MSE, including sigma = 0 (clamped weight)
    sg = torch.tensor([14.6146, 0.5, 0.0])
    sgd = sg.to(DEV)
    inst, loss = torch.empty(B, device=DEV), torch.empty(1, device=DEV)
    N.check(L.ctta_snr_mse_loss(N.ptr(v1d), N.ptr(v2d), N.ptr(sgd), 5.0, N.ptr(inst), N.ptr(loss), B, n, st)); sync()
    assert abs(float(loss.cpu()) - float(oheun.snr_mse_loss(v1, v2, sg, 5.0))) < 1e-6
    # two-shadow EMA: bit-exact against the reference arithmetic
    p = det("ema.p", (1003,), 1)
    a, b = det("ema.a", (1003,), 2), det("ema.b", (1003,), 3)
    pd, ad, bd = p.to(DEV), a.to(DEV), b.to(DEV)
    N.check(L.ctta_ema_update2(N.ptr(pd), N.ptr(ad), 0.95, N.ptr(bd), 0.999, 1003, st)); sync()
    ra = a.clone(); ra.add_((1. - 0.95) * (p - ra))
    rb = b.clone(); rb.add_((1. - 0.999) * (p - rb))
    assert torch.equal(ad.cpu(), ra) and torch.equal(bd.cpu(), rb)
    # vocoder post-processing
    wav = torch.tanh(det("wav", (2, 5000), 5) * 2) * 0.9 + 0.03
    wd_ = wav.to(DEV)
    scratch = torch.empty(4, device=DEV)
    cen = torch.empty_like(wd_)
    pcm = torch.empty(2, 5000, dtype=torch.int16, device=DEV)
    N.check(L.ctta_wav_finalize(N.ptr(wd_), wav.numel(), N.ptr(scratch), N.ptr(cen), N.ptr(pcm), st)); sync()
    ref = wav - (wav.max() + wav.min()) / 2
    assert torch.equal(cen.cpu(), ref)
    assert np.array_equal(pcm.cpu().numpy(), (ref.numpy() * 32768).astype("int16"))
    # the two halves a clip-sharded batch uses (extrema -> all-reduce -> centre): shard A and shard B centred with the
    # pair of the WHOLE batch reproduce the one-call result bit for bit (hifigan/utilities.py:85)
    mmA, mmB = torch.empty(2, device=DEV), torch.empty(2, device=DEV)
    N.check(L.ctta_wav_extrema(N.ptr(wd_[0]), 5000, N.ptr(scratch), N.ptr(mmA), st))
    N.check(L.ctta_wav_extrema(N.ptr(wd_[1]), 5000, N.ptr(scratch), N.ptr(mmB), st)); sync()
    assert mmA.tolist() == [float(wav[0].max()), float(wav[0].min())]
    both = torch.stack([torch.maximum(mmA[0], mmB[0]), torch.minimum(mmA[1], mmB[1])]).contiguous()
    cenS, pcmS = torch.empty_like(wd_), torch.empty_like(pcm)
    for i in range(2):
        N.check(L.ctta_wav_center(N.ptr(wd_[i]), 5000, N.ptr(both), N.ptr(scratch), N.ptr(cenS[i]), N.ptr(pcmS[i]), st))
    sync()
    assert torch.equal(cenS, cen) and torch.equal(pcmS, pcm)
    # the float4 min / max walk: a base pointer 4 bytes off 16-byte alignment and a length that is not a multiple of 4, with
    # the extrema placed in the unaligned head and in the tail
    for n, lo_at, hi_at in [(9999, 0, 9998), (4097, 4096, 1), (5, 4, 0), (3, 1, 2)]:
        w2 = torch.tanh(det("wav2.%d" % n, (n + 1,), 6)) * 0.5
        w2[1 + lo_at], w2[1 + hi_at] = -0.97, 0.91
        buf = w2.to(DEV)
        view = buf[1:]                      # 4-byte offset
        assert view.data_ptr() % 16 == 4
        cen2 = torch.empty(n, device=DEV)
        pcm2 = torch.empty(n, dtype=torch.int16, device=DEV)
        N.check(L.ctta_wav_finalize(N.ptr(view), n, N.ptr(scratch), N.ptr(cen2), N.ptr(pcm2), st)); sync()
        r2 = w2[1:] - (w2[1:].max() + w2[1:].min()) / 2
        assert torch.equal(cen2.cpu(), r2), n
        assert np.array_equal(pcm2.cpu().numpy(), (r2.numpy() * 32768).astype("int16")), n


def test_errors_are_loud():
    d = conv_desc()
    with pytest.raises(N.CttaError):
        N.check(lib().ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))


class _PackJob(ctypes.Structure):
    """ctta_pack_job of include/ctta.h."""
    _fields_ = [("src", ctypes.c_void_p), ("row_off", ctypes.c_void_p), ("col_off", ctypes.c_void_p),
                ("row_aux", ctypes.c_void_p), ("col_aux", ctypes.c_void_p), ("aux_limit", ctypes.c_int),
                ("n_rows", ctypes.c_int), ("k_pad", ctypes.c_int), ("block0", ctypes.c_int),
                ("dst", ctypes.c_void_p), ("src_row_len", ctypes.c_int), ("rows_per_block", ctypes.c_int)]


def _pack_case(name, cout, cin, taps, n_rows, aux_limit, seed):
    """A conv-weight-like job: src (cout, cin, taps) fp32 -> dst [n_rows][k_pad] with k = (tap, cin); some rows
    are zero rows (row_off < 0), the K tail is padding (col_off < 0), optional aux masking."""
    g = torch.Generator().manual_seed(seed)
    L = cin * taps
    K = taps * cin
    k_pad = (K + 63) // 64 * 64
    src = det(name, (cout, cin, taps), seed)
    ro = torch.full((n_rows,), -1, dtype=torch.int32)
    perm = torch.randperm(cout, generator=g)
    for r in range(min(n_rows, cout)):
        if r % 7 != 5:
            ro[r] = int(perm[r]) * L
    co = torch.full((k_pad,), -1, dtype=torch.int32)
    for t in range(taps):
        for c in range(cin):
            co[t * cin + c] = c * taps + t
    ra = torch.randint(0, 3, (n_rows,), generator=g, dtype=torch.int32)
    ca = torch.randint(0, 3, (k_pad,), generator=g, dtype=torch.int32)
    flat = src.reshape(-1)
    want = torch.zeros(n_rows, k_pad)
    for r in range(n_rows):
        if ro[r] < 0:
            continue
        ok = co >= 0
        if aux_limit > 0:
            ok = ok & ((ra[r] + ca) < aux_limit)
        idx = (ro[r] + co.clamp(min=0)).long()
        want[r] = torch.where(ok, flat[idx], torch.zeros(()))
    return dict(src=src, ro=ro, co=co, ra=ra, ca=ca, aux_limit=aux_limit, n_rows=n_rows, k_pad=k_pad, L=L,
                want=want.to(torch.bfloat16))


@pytest.mark.parametrize("threads,lds_floats", [(256, 9216), (1024, 36 * 1024)])
def test_pack_weight_tables_match_gather(threads, lds_floats):
    """ctta_pack_weight_multi (generic gather) and ctta_pack_weight_rows_multi (LDS-staged rows) against a host
    gather, bit-exact (round-to-nearest-even bf16): several rows per block, a ragged last block, zero rows, K padding,
    aux masking, a row length that is not a multiple of 4 (scalar staging), a row that fills the LDS class."""
    cases = [_pack_case("pk.a", 40, 32, 9, 37, 0, 1),      # 3x3 conv, rows of 288 floats, many rows per block
             _pack_case("pk.b", 70, 51, 1, 70, 0, 2),      # linear with odd width: scalar staging
             _pack_case("pk.c", 16, 24, 4, 20, 4, 3),      # aux masking, more destination rows than source rows
             _pack_case("pk.d", 3, lds_floats // 9, 9, 3, 0, 4)]   # one row per block, LDS filled
    keep = []
    for variant in ("rows", "generic"):
        jobs = (_PackJob * len(cases))()
        blocks = 0
        outs = []
        for i, c in enumerate(cases):
            dev = {k: c[k].contiguous().to(DEV) for k in ("src", "ro", "co", "ra", "ca")}
            out = torch.full((c["n_rows"], c["k_pad"]), 7.0, dtype=torch.bfloat16, device=DEV)
            keep.append(dev)
            outs.append(out)
            j = jobs[i]
            j.src, j.row_off, j.col_off = N.ptr(dev["src"]), N.ptr(dev["ro"]), N.ptr(dev["co"])
            j.row_aux, j.col_aux = N.ptr(dev["ra"]), N.ptr(dev["ca"])
            j.aux_limit, j.n_rows, j.k_pad, j.dst = c["aux_limit"], c["n_rows"], c["k_pad"], N.ptr(out)
            j.src_row_len, j.block0 = c["L"], blocks
            if variant == "rows":
                rpb = max(1, min(64, lds_floats // c["L"], c["n_rows"]))
                j.rows_per_block = rpb
                blocks += (c["n_rows"] + rpb - 1) // rpb
            else:
                blocks += (c["n_rows"] * c["k_pad"] + 2047) // 2048
        table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(DEV)
        if variant == "rows":
            N.check(lib().ctta_pack_weight_rows_multi(N.ptr(table), len(cases), blocks, lds_floats, threads,
                                                      N.stream_ptr()))
        else:
            N.check(lib().ctta_pack_weight_multi(N.ptr(table), len(cases), blocks, N.stream_ptr()))
        torch.cuda.synchronize()
        for c, out in zip(cases, outs):
            assert torch.equal(out.cpu().view(torch.int16), c["want"].view(torch.int16)), variant


@pytest.mark.parametrize("fft,hop,win", [(1024, 120, 600), (2048, 240, 1200), (512, 50, 240)])
def test_stft_magnitude_and_input_gradient_against_torch_stft_float64(fft, hop, win):
    """losses._STFTMagnitude (csrc/stft_loss.hip) against the reference's own recipe, tools/losses.py:146-169:
    torch.stft(x.double(), ...) -> sqrt(clamp(re^2 + im^2, 1e-8)) -> (B, frames, bins) float32, forward and d/dx."""
    from consistencytta_amd import losses
    B, T = 2, 9000
    gen = torch.Generator().manual_seed(fft)
    x = (torch.randn(B, T, generator=gen) * 0.2 * torch.linspace(0.05, 1.0, T)).float()
    direction = torch.randn(B, T // hop + 1, fft // 2 + 1, generator=gen)
    xr = x.clone().double().requires_grad_(True)
    spec = torch.stft(xr, fft, hop, win, torch.hann_window(win).double(), return_complex=True)
    ref = torch.clamp(spec.real ** 2 + spec.imag ** 2, min=1e-8).sqrt().transpose(2, 1)
    (ref * direction.double()).sum().backward()
    m = losses._STFTMagnitude(fft, hop, win, "hann_window")
    xg = x.to(DEV).requires_grad_(True)
    # a raw ctta_conv_gemm caller's split-K workspace binding survives the entry point (WsBind used to unbind)
    L_ = lib()
    mine = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    L_.ctta_conv_bind_workspace(N.ptr(mine), mine.numel())
    got = m(xg)
    ws, nb = ctypes.c_void_p(), ctypes.c_size_t()
    L_.ctta_conv_bound_workspace(ctypes.byref(ws), ctypes.byref(nb))
    L_.ctta_conv_bind_workspace(None, 0)
    assert ws.value == mine.data_ptr() and nb.value == mine.numel()
    assert got.shape == ref.shape and got.dtype == torch.float32
    (got * direction.to(DEV)).sum().backward()
    plain = m(x.to(DEV))                               # the no-grad handle gives the same numbers
    assert torch.equal(plain, got.detach())
    e_f = float((got.detach().cpu().double() - ref.detach()).norm() / ref.detach().norm())
    e_b = float((xg.grad.cpu().double() - xr.grad).norm() / xr.grad.norm())
    print("stft %d/%d/%d: magnitude rel_l2 %.2e, input gradient rel_l2 %.2e" % (fft, hop, win, e_f, e_b))
    assert e_f <= 2e-6 and e_b <= 1e-2      # forward: three-way bf16 split = fp32-grade; backward: one bf16 GEMM


def test_fp32_mlp_kernel_is_exact_beside_a_concurrent_conv_gemm():
    """Two engine handles on two streams put kernels of different kinds on one CU.  The fp32 MLP kernel once returned wrong
    rows there: an SLP-generated `v_pk_fma_f32 ... op_sel:[0,1,0]` reads a wrong dword beside another kernel's MFMA waves
    (tools/pk_hazard.py, LABNOTES.md 5).  The library is built without that form (tools/check_isa.py is the static guard);
    this is the dynamic one: a dependent pair of ctta_linear_f32 launches on fresh inputs beside three big conv_gemm
    launches on a second stream must match torch every time."""
    L = N.lib()
    M, K, Nn, C = 18, 256, 1024, 256
    W = torch.randn(Nn, K, device=DEV) * 0.05
    b = torch.randn(Nn, device=DEV)
    W2 = torch.randn(Nn, Nn, device=DEV) * 0.03
    xs = (torch.randn(16, 1, 20484, C, device=DEV) * 0.5).to(torch.bfloat16)
    outs = torch.empty_like(xs)
    ws = (torch.randn(C, 11 * C, device=DEV) * 0.05).to(torch.bfloat16)
    bs = torch.randn(C, device=DEV)
    d = N.ConvDesc()
    d.x0, d.c0 = xs.data_ptr(), C
    d.batch, d.hi, d.wi, d.ho, d.wo = 16, 1, 20484, 1, 20484
    d.kh, d.kw, d.stride_h, d.stride_w, d.dil_h, d.dil_w = 1, 11, 1, 1, 1, 1
    d.pad_h, d.pad_w = 0, 5
    d.w, d.k_pad, d.n, d.bias = ws.data_ptr(), 11 * C, C, bs.data_ptr()
    d.alpha, d.groups, d.out, d.ldc = 1.0, 1, outs.data_ptr(), C
    side = torch.cuda.Stream()
    worst = 0.0
    for _ in range(12):
        x = torch.randn(M, K, device=DEV)
        ref_h = F.silu(x @ W.t() + b)
        ref_y = ref_h @ W2.t() + b
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
        h = torch.empty(M, Nn, device=DEV)
        y = torch.empty(M, Nn, device=DEV)
        N.check(L.ctta_linear_f32(N.ptr(x), N.ptr(W), N.ptr(b), N.ptr(h), M, Nn, K, 0, 1, N.stream_ptr()))
        N.check(L.ctta_linear_f32(N.ptr(h), N.ptr(W2), N.ptr(b), N.ptr(y), M, Nn, Nn, 0, 0, N.stream_ptr()))
        torch.cuda.synchronize()
        worst = max(worst, float((h - ref_h).abs().max()), float((y - ref_y).abs().max()))
    print("linear_f32 chain beside conv_gemm: worst abs error %.2e" % worst)
    assert worst <= 2e-4


@pytest.mark.parametrize("tile", [0, 17, 18, 21, 22, 24, 28, 29, 32, 35, 36])
@pytest.mark.parametrize("mode", ["bias", "res", "res_out2", "acc", "res_acc_lrelu", "rowvec_res"])
def test_straight_line_epilogue_classes(tile, mode):
    """Every class of the straight-line wide-store epilogue (wide_epilogue_fast<RES, OUT2, ACC>: bias / per-sample row vector /
    residual / second LeakyReLU output / accumulate + scale + LeakyReLU) on 1-D convolutions whose row count is NOT a multiple
    of any tile (the last row tile is partial: rows past M must be dropped by the buffer bounds, and on the 256-row tiles whole
    waves lie past M and skip their MFMAs) and whose length is not a multiple of the tile either, so that with a row vector
    some tiles straddle two samples (generic path) and others do not (fast path)."""
    B, C, L, k = 3, 256, 1111, 3
    x = bf16_round(det("sle.x", (B, C, L), 1))
    w = bf16_round(det("sle.w", (C, C, k), 2) * (1.0 / math.sqrt(C * k)))
    b = det("sle.b", (C,), 3) * 0.1
    res = bf16_round(det("sle.r", (B, C, L), 4))
    old = bf16_round(det("sle.o", (B, C, L), 5))
    rv = det("sle.rv", (B, C), 6)
    ref = F.conv1d(x, w, b, padding=1)
    nlc = lambda t: t.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV)     # noqa: E731
    wp, k_pad = pack_conv_weight(w[:, :, None, :])
    out = nlc(old)
    kw = dict(x0=nlc(x), c0=C, batch=B, hi=1, wi=L, ho=1, wo=L, kh=1, kw=k, pad_w=1, w=wp, k_pad=k_pad, n=C, bias=b.to(DEV),
              out=out, ldc=C, tile=tile)
    keep, ref2, out2 = [wp, kw["x0"], kw["bias"]], None, None
    if mode in ("res", "res_out2", "res_acc_lrelu", "rowvec_res"):
        rs = nlc(res)
        kw.update(res=rs, res_ld=C)
        keep.append(rs)
        ref = ref + res
    if mode == "rowvec_res":
        rvd = rv.to(DEV)
        kw.update(rowvec=rvd, rowvec_ld=C)
        keep.append(rvd)
        ref = ref + rv[:, :, None]
    if mode in ("acc", "res_acc_lrelu"):
        kw.update(accumulate=1)
        ref = ref + old
    if mode == "res_acc_lrelu":
        kw.update(alpha=1.0 / 3.0, out_act=3, out_slope=0.1)
        ref = F.leaky_relu(ref / 3.0, 0.1)
    if mode == "res_out2":
        out2 = torch.zeros_like(out)
        kw.update(out2=out2, out2_slope=0.1)
        ref2 = F.leaky_relu(bf16_round(ref), 0.1)
    run_conv(conv_desc(**kw))
    got = out.to(torch.float32).permute(0, 2, 1).cpu()
    assert rel_err(got, ref) < 2 * BF16_TOL
    if ref2 is not None:
        assert rel_err(out2.to(torch.float32).permute(0, 2, 1).cpu(), ref2) < 2 * BF16_TOL
