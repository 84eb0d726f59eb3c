"""Backward-pass operators vs torch autograd on the CPU oracle ops (bf16-rounded inputs, fp32
reference): data gradients through conv_gemm with flipped weights, weight / bias / per-sample
gradients through the transposed-im2col GEMM, GroupNorm / LayerNorm / GEGLU / softmax backward,
the SNR-weighted loss gradient and AdamW."""
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from consistencytta_amd import _native as N  # noqa: E402
from gpu_util import DEV, bf16_round, conv_desc, det, from_nhwc, nhwc_bf16, pack_conv_weight, rel_err, run_conv, sync  # noqa: E402

BF16_TOL = 1.5 * 2.0 ** -8


def lib():
    return N.lib()


def rup(a, b):
    return (a + b - 1) // b * b


def wgrad(x_nhwc, dy_nhwc, B, H, W, Cin, Cout, k, stride, pad, ups, Ho, Wo, splits=2):
    """dW (Cout,Cin,k,k), db (Cout), per-sample sums (B,Cout) via im2col^T GEMM, as the engine does."""
    L = lib()
    st = N.stream_ptr()
    M = B * Ho * Wo
    mp = rup(M, 64 * splits)
    K = k * k * Cin
    R = K + 1 + B
    q = torch.empty(R, mp, dtype=torch.bfloat16, device=DEV)
    N.check(L.ctta_im2col_t(N.ptr(x_nhwc), Cin, B, H * (2 if ups else 1), W * (2 if ups else 1), int(ups), Ho, Wo, k, k,
                            stride, pad, pad, 1, N.ptr(q), mp, B, st))
    pt = torch.empty(Cout, mp, dtype=torch.bfloat16, device=DEV)
    N.check(L.ctta_transpose_bf16(N.ptr(dy_nhwc), 0, M, Cout, Cout, 0, N.ptr(pt), 0, mp, 1, st))
    ld = rup(R, 4)
    slabs = torch.empty(splits, Cout, ld, dtype=torch.float32, device=DEV)
    seg = mp // splits
    # slabs[s][n][r] = sum_m dY^T[n][m] Q[r][m]: one slab row per weight row, k contiguous
    run_conv(conv_desc(x0=pt, c0=seg, x_stride=mp, batch=1, hi=Cout, wi=1, ho=Cout, wo=1, w=q, k_pad=mp, n=R, out=slabs,
                       ldc=ld, out_f32=1, groups=splits, x_group_stride=seg, w_group_stride=seg,
                       out_group_stride=Cout * ld))
    ro = (torch.arange(Cout, dtype=torch.int32) * (Cin * k * k)).to(DEV)
    # im2col_t rows are (cin, kh, kw)-ordered like a conv weight row: identity column map (NULL)
    dw = torch.zeros(Cout, Cin, k, k, device=DEV)
    N.check(L.ctta_wgrad_scatter_rows(N.ptr(slabs), splits, Cout * ld, ld, K, Cout, N.ptr(ro), None, N.ptr(dw), 0, st))
    db = torch.zeros(Cout, device=DEV)
    N.check(L.ctta_col_scatter(N.ptr(slabs), splits, Cout * ld, ld, K, 1, Cout, None, N.ptr(db), 0, 0, st))
    ps = torch.zeros(B, Cout, device=DEV)
    N.check(L.ctta_col_scatter(N.ptr(slabs), splits, Cout * ld, ld, K + 1, B, Cout, None, N.ptr(ps), Cout, 0, st))
    sync()
    return dw.cpu(), db.cpu(), ps.cpu()


def dgrad(dy_nhwc, w, B, Ho, Wo, Cin, Cout, k, pad, H, W):
    """dX = conv(dY, W rotated by 180 degrees with in/out channels swapped), stride 1."""
    wf = w.flip(2, 3).permute(1, 0, 2, 3).contiguous()      # (Cin, Cout, k, k)
    wp, k_pad = pack_conv_weight(wf)
    dx = torch.empty(B, H, W, Cin, dtype=torch.bfloat16, device=DEV)
    run_conv(conv_desc(x0=dy_nhwc, c0=Cout, batch=B, hi=Ho, wi=Wo, ho=H, wo=W, kh=k, kw=k, pad_h=k - 1 - pad,
                       pad_w=k - 1 - pad, w=wp, k_pad=k_pad, n=Cin, out=dx, ldc=Cin))
    return dx


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,stride,ups", [(2, 64, 96, 8, 8, 3, 1, False), (3, 40, 24, 6, 4, 3, 1, False),
                                                         (2, 64, 64, 8, 8, 1, 1, False), (2, 32, 40, 8, 8, 3, 2, False),
                                                         (2, 48, 64, 4, 4, 3, 1, True)])
def test_conv_backward(B, Cin, Cout, H, W, k, stride, ups):
    pad = k // 2
    x = bf16_round(det("cb.x", (B, Cin, H, W), 1)).requires_grad_(True)
    w = bf16_round(det("cb.w", (Cout, Cin, k, k), 2) / math.sqrt(Cin * k * k)).requires_grad_(True)
    b = (det("cb.b", (Cout,), 3) * 0.1).requires_grad_(True)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    y = F.conv2d(xin, w, b, stride=stride, padding=pad)
    Ho, Wo = y.shape[2], y.shape[3]
    dy = bf16_round(det("cb.dy", tuple(y.shape), 4))
    y.backward(dy)
    xa, dya = nhwc_bf16(x.detach()), nhwc_bf16(dy)
    dw, db, ps = wgrad(xa, dya, B, H, W, Cin, Cout, k, stride, pad, ups, Ho, Wo)
    assert rel_err(dw, w.grad) < 2e-3          # fp32 accumulation of bf16 products
    assert rel_err(db, b.grad) < 2e-3
    assert rel_err(ps, dy.sum(dim=(2, 3))) < 2e-3
    # data gradient
    L = lib()
    st = N.stream_ptr()
    if stride == 2:
        hz, wz = H, W
        dz = torch.empty(B, hz, wz, Cout, dtype=torch.bfloat16, device=DEV)
        N.check(L.ctta_zero_insert2(N.ptr(dya), N.ptr(dz), B, Ho, Wo, hz, wz, Cout, st))
        dx = dgrad(dz, w.detach(), B, hz, wz, Cin, Cout, k, pad, H, W)
    elif ups:
        dup = dgrad(dya, w.detach(), B, Ho, Wo, Cin, Cout, k, pad, 2 * H, 2 * W)
        dx = torch.empty(B, H, W, Cin, dtype=torch.bfloat16, device=DEV)
        N.check(L.ctta_pool2_sum(N.ptr(dup), N.ptr(dx), B, H, W, Cin, 0, st))
    else:
        dx = dgrad(dya, w.detach(), B, Ho, Wo, Cin, Cout, k, pad, H, W)
    sync()
    assert rel_err(from_nhwc(dx), x.grad) < (2.5 if ups else 1.0) * BF16_TOL


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,splits,nb", [(2, 64, 96, 8, 8, 3, 1, 2), (3, 40, 24, 32, 2, 3, 2, 3),
                                                        (2, 128, 72, 16, 16, 3, 4, 0), (1, 320, 64, 8, 16, 3, 1, 1),
                                                        (2, 64, 64, 8, 32, 3, 2, 2), (1, 256, 248, 200, 1, 1, 1, 0),
                                                        (1, 1280, 40, 77, 1, 1, 2, 0), (4, 72, 520, 16, 4, 1, 4, 0)])
def test_implicit_weight_gradient(B, Cin, Cout, H, W, k, splits, nb):
    """ctta_wgrad_implicit (csrc/wgrad_gemm.hip): dW of F.conv2d(3x3, stride 1, pad 1) / F.linear read from the NHWC input
    through LDS transpose reads, against torch autograd and against the im2col^T route it replaces: 3x3 at every width the
    U-Net has (2 .. 32), ragged channel counts on both sides, split positions, bias and per-sample columns."""
    L = lib()
    st = N.stream_ptr()
    pad = k // 2
    x = bf16_round(det("wi.x", (B, Cin, H, W), 1)).requires_grad_(True)
    w = bf16_round(det("wi.w", (Cout, Cin, k, k), 2) / math.sqrt(Cin * k * k)).requires_grad_(True)
    b = (det("wi.b", (Cout,), 3) * 0.1).requires_grad_(True)
    y = F.conv2d(x, w, b, padding=pad)
    dy = bf16_round(det("wi.dy", tuple(y.shape), 4))
    y.backward(dy)
    xa, dya = nhwc_bf16(x.detach()), nhwc_bf16(dy)
    M = B * H * W
    taps = k * k
    assert L.ctta_wgrad_implicit_supported(taps, Cin, H, W, Cin, Cout) == 1
    mp = rup(M, 64 * splits)
    pt = torch.empty(Cout, mp, dtype=torch.bfloat16, device=DEV)
    N.check(L.ctta_transpose_bf16(N.ptr(dya), 0, M, Cout, Cout, 0, N.ptr(pt), 0, mp, 1, st))
    K = taps * Cin
    ld = rup(K + 1 + nb, 4)
    slabs = torch.full((splits, Cout, ld), float("nan"), device=DEV)
    N.check(L.ctta_wgrad_implicit(N.ptr(pt), Cout, mp, N.ptr(xa), Cin, Cin, B, H, W, taps, M, splits, K, nb, N.ptr(slabs),
                                  Cout * ld, ld, st))
    sync()
    got = slabs[:, :, :K + 1 + nb].sum(0).cpu()
    assert torch.isfinite(got).all()
    assert rel_err(got[:, :K].reshape(Cout, Cin, k, k), w.grad) < 2e-3
    assert rel_err(got[:, K], b.grad) < 2e-3
    if nb:
        assert rel_err(got[:, K + 1:K + 1 + nb].t(), dy.sum(dim=(2, 3))[:nb]) < 2e-3
    # the route it replaces gives the same numbers up to the fp32 summation order
    dw_old, db_old, _ = wgrad(xa, dya, B, H, W, Cin, Cout, k, 1, pad, False, H, W)
    assert rel_err(got[:, :K].reshape(Cout, Cin, k, k), dw_old) < 1e-4
    if k == 3 and Cout % 8 == 0:   # round 4: the same product with dY read where it lies (no transposed copy, sums in the kernel)
        slabs2 = torch.full((splits, Cout, ld), float("nan"), device=DEV)
        N.check(L.ctta_wgrad_implicit_inplace(N.ptr(dya), Cout, Cout, mp, N.ptr(xa), Cin, Cin, B, H, W, taps, M, splits, K, nb,
                                              N.ptr(slabs2), Cout * ld, ld, st))
        sync()
        got2 = slabs2[:, :, :K + 1 + nb].sum(0).cpu()
        assert torch.isfinite(got2).all()
        assert torch.equal(got2[:, :K], got[:, :K])                       # the same MFMA products in the same order
        assert rel_err(got2[:, K], b.grad) < 2e-3
        if nb:
            assert rel_err(got2[:, K + 1:K + 1 + nb].t(), dy.sum(dim=(2, 3))[:nb]) < 2e-3
        # round 5: ONE split, the tile added straight into the gradient tensors through the pack map (no slab, no scatter):
        # bit-identical to slab + ctta_wgrad_scatter_rows_bias(accumulate = 1), padded rows and bias slots dropped
        mp1 = rup(M, 64)
        ld1 = rup(K + 1, 4)
        slab1 = torch.full((1, Cout, ld1), float("nan"), device=DEV)
        N.check(L.ctta_wgrad_implicit_inplace(N.ptr(dya), Cout, Cout, mp1, N.ptr(xa), Cin, Cin, B, H, W, taps, M, 1, K, 0,
                                              N.ptr(slab1), Cout * ld1, ld1, st))
        ro = torch.arange(Cout, dtype=torch.int32) * K
        bidx = torch.arange(Cout, dtype=torch.int32)
        ro[1], bidx[2] = -1, -1
        ro_d, bi_d = ro.to(DEV), bidx.to(DEV)
        gw0, gb0 = det("wi.gw", (Cout * K,), 8).to(DEV), det("wi.gb", (Cout,), 9).to(DEV)
        gw_s, gb_s, gw_d, gb_d = gw0.clone(), gb0.clone(), gw0.clone(), gb0.clone()
        N.check(L.ctta_wgrad_scatter_rows_bias(N.ptr(slab1), 1, Cout * ld1, ld1, K, Cout, N.ptr(ro_d), None, N.ptr(gw_s), K, Cout,
                                               N.ptr(bi_d), N.ptr(gb_s), 1, st))
        N.check(L.ctta_wgrad_implicit_direct(N.ptr(dya), Cout, Cout, mp1, N.ptr(xa), Cin, Cin, B, H, W, M, K, Cout, N.ptr(ro_d),
                                             N.ptr(gw_d), Cout, N.ptr(bi_d), N.ptr(gb_d), st))
        sync()
        assert torch.equal(gw_d, gw_s) and torch.equal(gb_d, gb_s)
        assert torch.equal(gw_d.view(Cout, K)[1], gw0.view(Cout, K)[1]) and float(gb_d[2]) == float(gb0[2])
        assert rel_err((gw_d - gw0).cpu().view(Cout, K)[2:].reshape(Cout - 2, Cin, k, k), w.grad[2:]) < 2e-3
    # outside the kernel's range: refused loudly
    assert L.ctta_wgrad_implicit_supported(9, 64, 8, 48, 64, 64) == 0 and L.ctta_wgrad_implicit_supported(9, 64, 6, 4, 64, 64) == 0
    assert L.ctta_wgrad_implicit_supported(4, 64, 8, 8, 64, 64) == 0 and L.ctta_wgrad_implicit_supported(1, 8, 8, 8, 8, 64) == 0
    with pytest.raises(RuntimeError):
        N.check(L.ctta_wgrad_implicit(N.ptr(pt), Cout, mp, N.ptr(xa), Cin, Cin, B, H, W, 4, M, splits, K, nb, N.ptr(slabs),
                                      Cout * ld, ld, st))


@pytest.mark.parametrize("M,Cin,Cout,splits,x_ld,dy_ld", [(200, 256, 248, 1, 256, 248), (77, 1280, 40, 2, 1280, 40),
                                                          (1000, 72, 520, 4, 72, 520), (4096, 320, 256, 2, 384, 512),
                                                          (333, 136, 264, 1, 136, 264), (9216, 512, 1024, 8, 512, 1024)])
def test_weight_gradient_with_both_operands_in_place(M, Cin, Cout, splits, x_ld, dy_ld):
    """ctta_wgrad_tn (csrc/wgrad_gemm.hip, round 4): dW / db of F.linear with dY [M][N] and X [M][K] read WHERE THEY LIE --
    both MFMA operands through transposing LDS reads, no transposed copies (autograd of attention.py:276-334's linears).
    Ragged row counts (zero rows up to the 64 * splits padding), column counts that are not multiples of the 128-wide
    tiles, operands that are column slices of wider matrices (x_ld / dy_ld > width), splits; against torch autograd."""
    L = lib()
    st = N.stream_ptr()
    x = bf16_round(det("tn.x", (M, Cin), 1)).requires_grad_(True)
    w = bf16_round(det("tn.w", (Cout, Cin), 2) / math.sqrt(Cin)).requires_grad_(True)
    b = (det("tn.b", (Cout,), 3) * 0.1).requires_grad_(True)
    y = F.linear(x, w, b)
    dy = bf16_round(det("tn.dy", (M, Cout), 4))
    y.backward(dy)
    xa = torch.full((M, x_ld), 7.0, dtype=torch.bfloat16, device=DEV)          # the columns beyond Cin / Cout must not leak in
    xa[:, :Cin] = x.detach().to(torch.bfloat16).to(DEV)
    dya = torch.full((M, dy_ld), -3.0, dtype=torch.bfloat16, device=DEV)
    dya[:, :Cout] = dy.to(torch.bfloat16).to(DEV)
    mp = rup(M, 64 * splits)
    ld = rup(Cin + 1, 4)
    slabs = torch.full((splits, Cout, ld), float("nan"), device=DEV)
    N.check(L.ctta_wgrad_tn(N.ptr(dya), dy_ld, Cout, N.ptr(xa), x_ld, Cin, M, mp, splits, Cin, N.ptr(slabs), Cout * ld, ld, st))
    sync()
    got = slabs[:, :, :Cin + 1].sum(0).cpu()
    assert torch.isfinite(got).all()
    assert rel_err(got[:, :Cin], w.grad) < 2e-3
    assert rel_err(got[:, Cin], b.grad) < 2e-3
    with pytest.raises(RuntimeError):       # widths that are not multiples of 8 are refused loudly
        N.check(L.ctta_wgrad_tn(N.ptr(dya), dy_ld, Cout - 1, N.ptr(xa), x_ld, Cin, M, mp, splits, Cin, N.ptr(slabs), Cout * ld, ld, st))
    # round 5: one split, added straight into the gradient tensors -- with a column map that drops two padding columns (the
    # transformer's inner width 255 -> 256) and with the identity map; bit-identical to slab + scatter
    mp1 = rup(M, 64)
    slab1 = torch.full((1, Cout, ld), float("nan"), device=DEV)
    N.check(L.ctta_wgrad_tn(N.ptr(dya), dy_ld, Cout, N.ptr(xa), x_ld, Cin, M, mp1, 1, Cin, N.ptr(slab1), Cout * ld, ld, st))
    for use_map in (False, True):
        Kd = Cin - 2 if use_map else Cin
        co = None
        if use_map:
            co = torch.arange(Cin, dtype=torch.int32) - 1
            co[0], co[Cin - 1] = -1, -1                    # columns 0 and Cin - 1 are padding: dropped
            co = co.to(DEV)
        ro = (torch.arange(Cout, dtype=torch.int32) * Kd)
        bidx = torch.arange(Cout, dtype=torch.int32)
        ro[0], bidx[1] = -1, -1
        ro_d, bi_d = ro.to(DEV), bidx.to(DEV)
        gw0, gb0 = det("tn.gw", (Cout * Kd,), 8).to(DEV), det("tn.gb", (Cout,), 9).to(DEV)
        gw_s, gb_s, gw_d, gb_d = gw0.clone(), gb0.clone(), gw0.clone(), gb0.clone()
        N.check(L.ctta_wgrad_scatter_rows_bias(N.ptr(slab1), 1, Cout * ld, ld, Cin, Cout, N.ptr(ro_d), N.ptr(co), N.ptr(gw_s), Cin, Cout,
                                               N.ptr(bi_d), N.ptr(gb_s), 1, st))
        N.check(L.ctta_wgrad_tn_direct(N.ptr(dya), dy_ld, Cout, N.ptr(xa), x_ld, Cin, M, mp1, Cin, Cout, N.ptr(ro_d), N.ptr(co),
                                       N.ptr(gw_d), Cout, N.ptr(bi_d), N.ptr(gb_d), st))
        sync()
        assert torch.equal(gw_d, gw_s) and torch.equal(gb_d, gb_s), use_map
        assert bool(torch.isfinite(gw_d).all()) and not torch.equal(gw_d, gw0)
        # a gradient tensor that starts 1 .. 3 floats past a 16-byte boundary (a caller's own sub-view of a flat buffer): the
        # four-wide add is chosen on the ABSOLUTE address, so the result is the aligned one bit for bit and the floats either
        # side of the view are untouched
        for off in (1, 2, 3):
            buf = torch.full((Cout * Kd + 8,), 123.0, device=DEV)
            gw_o, gb_o = buf[off:off + Cout * Kd], gb0.clone()
            gw_o.copy_(gw0)
            assert N.ptr(gw_o).value % 16 == 4 * off
            N.check(L.ctta_wgrad_tn_direct(N.ptr(dya), dy_ld, Cout, N.ptr(xa), x_ld, Cin, M, mp1, Cin, Cout, N.ptr(ro_d), N.ptr(co),
                                           N.ptr(gw_o), Cout, N.ptr(bi_d), N.ptr(gb_o), st))
            sync()
            assert torch.equal(gw_o, gw_d) and torch.equal(gb_o, gb_d), (use_map, off)
            assert bool((buf[:off] == 123.0).all()) and bool((buf[off + Cout * Kd:] == 123.0).all())


def test_slab_scatter_with_the_bias_column_in_the_same_launch():
    """ctta_wgrad_scatter_rows_bias (round 4): grad_w[row_off[n] + k] += sum_s slab[s][n][k] AND grad_b[bias_idx[n]] += sum_s
    slab[s][n][bias_col] in one launch, against ctta_wgrad_scatter_rows + ctta_col_scatter (the two launches it replaces);
    ctta_wgrad_rowsum: the row sums of dY^T (bias / per-sample columns) against torch."""
    L = lib()
    st = N.stream_ptr()
    S, n_rows, K, ld = 5, 37, 64, 68
    slabs = det("sc.slab", (S, n_rows, ld), 1).to(DEV).contiguous()
    ro = torch.arange(n_rows, dtype=torch.int32) * K
    ro[3] = -1                                            # a padded row: dropped
    bidx = torch.arange(n_rows, dtype=torch.int32)
    bidx[3] = -1
    ro_d, bi_d = ro.to(DEV), bidx.to(DEV)
    gw0, gb0 = det("sc.gw", (n_rows * K,), 2).to(DEV), det("sc.gb", (n_rows,), 3).to(DEV)
    gw1, gb1 = gw0.clone(), gb0.clone()
    N.check(L.ctta_wgrad_scatter_rows_bias(N.ptr(slabs), S, n_rows * ld, ld, K, n_rows, N.ptr(ro_d), None, N.ptr(gw1), K, n_rows,
                                           N.ptr(bi_d), N.ptr(gb1), 1, st))
    gw2, gb2 = gw0.clone(), gb0.clone()
    N.check(L.ctta_wgrad_scatter_rows(N.ptr(slabs), S, n_rows * ld, ld, K, n_rows, N.ptr(ro_d), None, N.ptr(gw2), 1, st))
    N.check(L.ctta_col_scatter(N.ptr(slabs), S, n_rows * ld, ld, K, 1, n_rows, N.ptr(bi_d), N.ptr(gb2), 0, 1, st))
    sync()
    assert torch.equal(gw1, gw2)
    assert rel_err(gb1.cpu(), gb2.cpu()) < 1e-6 and float(gb1[3]) == float(gb0[3])
    ref_w = gw0.cpu().view(n_rows, K) + slabs.cpu()[:, :, :K].sum(0)
    ref_w[3] = gw0.cpu().view(n_rows, K)[3]
    assert rel_err(gw1.cpu().view(n_rows, K), ref_w) < 1e-6
    # many slabs: the eight-loads-in-flight loop and its ragged tail, narrow and wide layers, against the fp64 sum
    for S2, rows2, K2 in ((64, 19, 256), (33, 7, 320), (16, 5, 512), (21, 3, 1280)):
        ld2 = K2 + 4
        sl2 = det("sc.slab2.%d" % K2, (S2, rows2, ld2), 1).to(DEV).contiguous()
        ro2 = (torch.arange(rows2, dtype=torch.int32) * K2).to(DEV)
        g0 = det("sc.g2.%d" % K2, (rows2 * K2,), 2).to(DEV)
        g1 = g0.clone()
        N.check(L.ctta_wgrad_scatter_rows_bias(N.ptr(sl2), S2, rows2 * ld2, ld2, K2, rows2, N.ptr(ro2), None, N.ptr(g1), -1, 0,
                                               None, None, 1, st))
        sync()
        ref2 = g0.cpu().double().view(rows2, K2) + sl2.cpu().double()[:, :, :K2].sum(0)
        assert rel_err(g1.cpu().view(rows2, K2), ref2.float()) < 1e-6, (S2, K2)
    # row sums of dY^T with per-sample columns
    Nn, B, hw, splits = 24, 3, 64, 2
    M = B * hw
    mp = rup(M, 64 * splits)
    dyt = torch.zeros(Nn, mp, dtype=torch.bfloat16, device=DEV)
    dyt[:, :M] = bf16_round(det("sc.dyt", (Nn, M), 4)).to(torch.bfloat16).to(DEV)
    ld2 = rup(8 + 1 + B, 4)
    out = torch.full((splits, Nn, ld2), float("nan"), device=DEV)
    N.check(L.ctta_wgrad_rowsum(N.ptr(dyt), Nn, mp, M, splits, hw, B, N.ptr(out), Nn * ld2, ld2, 8, st))
    sync()
    got = out[:, :, 8:8 + 1 + B].sum(0).cpu()
    d = dyt[:, :M].float().cpu()
    assert rel_err(got[:, 0], d.sum(1)) < 1e-5
    assert rel_err(got[:, 1:], d.view(Nn, B, hw).sum(2)) < 1e-5


@pytest.mark.parametrize("B,C,H,W,G,silu,eps", [(2, 40, 16, 8, 8, True, 1e-5), (3, 256, 8, 8, 32, False, 1e-6),
                                                (2, 120, 4, 2, 8, True, 1e-5),
                                                # many chunks per sample: the chunk walk of gn_bwd_fold split over the block's
                                                # threads (C <= 512), with C not dividing 1024 and C = 512
                                                (2, 320, 64, 16, 32, True, 1e-5), (2, 512, 32, 16, 32, True, 1e-5),
                                                (1, 1024, 16, 16, 32, False, 1e-5)])
def test_groupnorm_backward(B, C, H, W, G, silu, eps):
    x = bf16_round(det("gb.x", (B, C, H, W), 1) * 2 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * det("gb.g", (C,), 2)).requires_grad_(True)
    beta = (0.1 * det("gb.b", (C,), 3)).requires_grad_(True)
    y = F.group_norm(x, G, gamma, beta, eps)
    if silu:
        y = F.silu(y)
    dy = bf16_round(det("gb.dy", (B, C, H, W), 4))
    y.backward(dy)
    L = lib()
    st = N.stream_ptr()
    xa, dya = nhwc_bf16(x.detach()), nhwc_bf16(dy)
    stats = torch.empty(B, G, 2, device=DEV)
    N.check(L.ctta_groupnorm_stats(N.ptr(xa), B, H * W, C, G, eps, N.ptr(stats), st))
    scratch = torch.empty(L.ctta_groupnorm_bwd_scratch_floats(B, H * W, C, G), device=DEV)
    dx = torch.empty_like(xa)
    dg, dbt = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    gd, bd = gamma.detach().to(DEV), beta.detach().to(DEV)
    N.check(L.ctta_groupnorm_bwd(N.ptr(xa), N.ptr(dya), N.ptr(dx), B, H * W, C, G, N.ptr(stats), N.ptr(gd), N.ptr(bd),
                                 int(silu), 0, N.ptr(dg), N.ptr(dbt), 0, N.ptr(scratch), st))
    sync()
    assert rel_err(from_nhwc(dx), x.grad) < 2 * BF16_TOL
    assert rel_err(dg.cpu(), gamma.grad) < 2e-3 and rel_err(dbt.cpu(), beta.grad) < 2e-3


@pytest.mark.parametrize("rows,d,ld", [(70, 39, 64), (130, 255, 256), (9, 1020, 1024), (4100, 255, 256), (2304, 510, 512)])
def test_layernorm_geglu_backward(rows, d, ld):
    x = bf16_round(det("lb.x", (rows, d), 1) * 3 + 0.5).requires_grad_(True)
    gamma = (1 + 0.2 * det("lb.g", (d,), 2)).requires_grad_(True)
    beta = (0.1 * det("lb.b", (d,), 3)).requires_grad_(True)
    y = F.layer_norm(x, (d,), gamma, beta, 1e-5)
    dy = bf16_round(det("lb.dy", (rows, d), 4))
    y.backward(dy)
    xp, dyp = torch.zeros(rows, ld), torch.zeros(rows, ld)
    xp[:, :d], dyp[:, :d] = x.detach(), dy
    xd, dyd = xp.to(torch.bfloat16).to(DEV), dyp.to(torch.bfloat16).to(DEV)
    dx = torch.empty_like(xd)
    dg, dbt = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    gd = gamma.detach().to(DEV)
    N.check(lib().ctta_layernorm_bwd(N.ptr(xd), N.ptr(dyd), N.ptr(dx), rows, d, ld, N.ptr(gd), 1e-5, 0, N.ptr(dg),
                                     N.ptr(dbt), N.stream_ptr()))
    sync()
    assert rel_err(dx.float().cpu()[:, :d], x.grad) < 2 * BF16_TOL
    assert rel_err(dg.cpu(), gamma.grad) < 2e-3 and rel_err(dbt.cpu(), beta.grad) < 2e-3
    # round 4: dx_out = dx_add + dL/dx with a SEPARATE output (the transformer's token-stream gradient is never rewritten in
    # place): equals the in-place accumulate bit for bit and leaves dx_add untouched
    add = bf16_round(det("lb.add", (rows, ld), 7)).to(torch.bfloat16).to(DEV)
    keep = add.clone()
    out2, dg2, db2 = torch.empty_like(xd), torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    N.check(lib().ctta_layernorm_bwd_add(N.ptr(xd), N.ptr(dyd), N.ptr(add), N.ptr(out2), rows, d, ld, N.ptr(gd), 1e-5, N.ptr(dg2),
                                         N.ptr(db2), N.stream_ptr()))
    inpl, dg3, db3 = add.clone(), torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    N.check(lib().ctta_layernorm_bwd(N.ptr(xd), N.ptr(dyd), N.ptr(inpl), rows, d, ld, N.ptr(gd), 1e-5, 1, N.ptr(dg3),
                                     N.ptr(db3), N.stream_ptr()))
    sync()
    assert torch.equal(add, keep) and torch.equal(out2, inpl)
    assert rel_err(out2.float().cpu()[:, :d], add.float().cpu()[:, :d] + x.grad) < 2 * BF16_TOL
    # round 5: the per-block partial table of d gamma / d beta is the CALLER's (no workspace inside the library): with it
    # the fold is per-block sums + a sliced reduce, without it one atomic per block and column -- same dx bit for bit,
    # same parameter gradients up to the order of the fp32 additions
    nf = int(lib().ctta_layernorm_bwd_scratch_floats(rows, ld))
    assert (nf > 0) if rows >= 2000 else (nf == 0 or rows >= 121)      # >= 16 blocks of >= 8 rows take the two-pass fold
    part = torch.empty(max(nf, 1), device=DEV)
    out4, dg4, db4 = torch.empty_like(xd), torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    N.check(lib().ctta_layernorm_bwd_ws(N.ptr(xd), N.ptr(dyd), None, N.ptr(out4), rows, d, ld, N.ptr(gd), 1e-5, N.ptr(dg4),
                                        N.ptr(db4), N.ptr(part), nf, N.stream_ptr()))
    sync()
    assert torch.equal(out4, dx)
    assert rel_err(dg4.cpu(), gamma.grad) < 2e-3 and rel_err(db4.cpu(), beta.grad) < 2e-3
    assert rel_err(dg4.cpu(), dg.cpu()) < 1e-5 and rel_err(db4.cpu(), dbt.cpu()) < 1e-5
    # GEGLU
    hp = ld
    f = bf16_round(det("gg.f", (rows, 2 * hp), 5) * 2).requires_grad_(True)
    out = f[:, :hp] * F.gelu(f[:, hp:])
    do = bf16_round(det("gg.do", (rows, hp), 6))
    out.backward(do)
    fd, dod = f.detach().to(torch.bfloat16).to(DEV), do.to(torch.bfloat16).to(DEV)
    df = torch.empty_like(fd)
    N.check(lib().ctta_geglu_bwd(N.ptr(fd), N.ptr(dod), N.ptr(df), rows, hp, 0, N.stream_ptr()))
    sync()
    assert rel_err(df.float().cpu(), f.grad) < 2 * BF16_TOL
    # interleaved 16-blocks (engine layout): same numbers at permuted columns
    perm = torch.empty(2 * hp, dtype=torch.long)
    for j in range(hp):
        perm[(j // 16) * 32 + j % 16], perm[(j // 16) * 32 + 16 + j % 16] = j, hp + j
    fi = fd[:, perm.to(DEV)].contiguous()
    dfi = torch.empty_like(fi)
    N.check(lib().ctta_geglu_bwd(N.ptr(fi), N.ptr(dod), N.ptr(dfi), rows, hp, 1, N.stream_ptr()))
    sync()
    assert torch.equal(dfi.cpu(), df[:, perm.to(DEV)].cpu())


def test_softmax_forward_bias_and_backward():
    rows, cols, ldp, rpb = 12, 37, 40, 4
    s = (det("sb.s", (rows, cols), 1) * 8).requires_grad_(True)
    bias = torch.zeros(rows // rpb, cols)
    bias[1, 20:] = -10000.0
    p = torch.softmax(s * 0.14 + bias.repeat_interleave(rpb, 0), dim=-1)
    dp = det("sb.dp", (rows, cols), 2)
    p.backward(dp)
    L = lib()
    st = N.stream_ptr()
    sd, bd, dpd = s.detach().to(DEV), bias.to(DEV), dp.to(DEV)
    pd = torch.empty(rows, ldp, dtype=torch.bfloat16, device=DEV)
    N.check(L.ctta_softmax_bias_rows(N.ptr(sd), cols, N.ptr(bd), rpb, N.ptr(pd), rows, cols, ldp, 0.14, st))
    ds = torch.empty(rows, ldp, dtype=torch.bfloat16, device=DEV)
    N.check(L.ctta_softmax_bwd_rows(N.ptr(pd), N.ptr(dpd), cols, N.ptr(ds), rows, cols, ldp, 0.14, st))
    sync()
    assert rel_err(pd.float().cpu()[:, :cols], p.detach()) < BF16_TOL
    assert float(pd.float().cpu()[:, cols:].abs().max()) == 0.0
    assert rel_err(ds.float().cpu()[:, :cols], s.grad) < 3 * BF16_TOL


def test_transpose_and_add_slices():
    g, rows, cols, ld = 3, 50, 24, 40
    x = bf16_round(det("tr.x", (g, rows, ld), 1))
    xd = x.to(torch.bfloat16).to(DEV)
    dst = torch.full((g, cols, 56), 7.0, dtype=torch.bfloat16, device=DEV)
    N.check(lib().ctta_transpose_bf16(N.ptr(xd), rows * ld, rows, cols, ld, 8, N.ptr(dst), cols * 56, 56, g, N.stream_ptr()))
    sync()
    got = dst.float().cpu()
    assert torch.equal(got[:, :, :rows], x[:, :, 8:8 + cols].transpose(1, 2))
    assert float(got[:, :, rows:].abs().max()) == 0.0
    a, b = bf16_round(det("as.a", (20, 32), 2)), bf16_round(det("as.b", (20, 48), 3))
    ad, bd = a.to(torch.bfloat16).to(DEV), b.to(torch.bfloat16).to(DEV)
    o = torch.empty(20, 16, dtype=torch.bfloat16, device=DEV)
    N.check(lib().ctta_add_slices(N.ptr(ad[:, 16:]), 32, N.ptr(bd[:, 8:]), 48, N.ptr(o), 16, 20, 16, N.stream_ptr()))
    sync()
    assert torch.equal(o.float().cpu(), bf16_round(a[:, 16:] + b[:, 8:24]))


def test_embedding_mlp_backward_loss_grad_and_adamw():
    M, Kd, Nn = 5, 48, 36
    x = det("lf.x", (M, Kd), 1).requires_grad_(True)
    w = (det("lf.w", (Nn, Kd), 2) * 0.3).requires_grad_(True)
    b = det("lf.b", (Nn,), 3).requires_grad_(True)
    dy = det("lf.dy", (M, Nn), 4)
    F.linear(F.silu(x), w, b).backward(dy)
    L = lib()
    st = N.stream_ptr()
    xs = F.silu(x.detach()).to(DEV)
    dx, dw, db = torch.empty(M, Kd, device=DEV), torch.empty(Nn, Kd, device=DEV), torch.empty(Nn, device=DEV)
    xpre, wd, dyd = x.detach().to(DEV), w.detach().to(DEV), dy.to(DEV)
    N.check(L.ctta_linear_f32_bwd(N.ptr(xs), N.ptr(wd), N.ptr(dyd), Nn, N.ptr(xpre), N.ptr(dx), N.ptr(dw), N.ptr(db), M, Nn,
                                  Kd, 0, 0, st))
    sync()
    assert rel_err(dx.cpu(), x.grad) < 1e-5 and rel_err(dw.cpu(), w.grad) < 1e-5 and rel_err(db.cpu(), b.grad) < 1e-5
    # a column slice of a wider dY (the per-resnet rows of the concatenated time_emb_proj table)
    wide = torch.zeros(M, Nn + 11, device=DEV)
    wide[:, 7:7 + Nn] = dyd
    dw2, db2 = torch.zeros_like(dw), torch.zeros_like(db)
    N.check(L.ctta_linear_f32_bwd(N.ptr(xs), N.ptr(wd), N.ptr(wide[:, 7:]), Nn + 11, None, None, N.ptr(dw2), N.ptr(db2), M,
                                  Nn, Kd, 0, 1, st))
    sync()
    assert rel_err(dw2.cpu(), w.grad) < 1e-5 and rel_err(db2.cpu(), b.grad) < 1e-5
    # loss gradient
    B, C, H, W = 3, 8, 6, 4
    pred = det("lg.p", (B, C, H, W), 5).requires_grad_(True)
    tgt = det("lg.t", (B, C, H, W), 6)
    sig = torch.tensor([14.6, 0.4, 0.0])
    inst = ((pred - tgt) ** 2).mean(dim=(1, 2, 3))
    (inst * torch.clamp(sig ** -2, max=5.0)).mean().backward()
    out = torch.empty(B, H * W, 8, dtype=torch.bfloat16, device=DEV)
    pd, td, sd = pred.detach().to(DEV), tgt.to(DEV), sig.to(DEV)
    N.check(L.ctta_snr_mse_grad(N.ptr(pd), N.ptr(td), N.ptr(sd), 5.0, 1.0, B, C, H * W, 8, N.ptr(out), st))
    sync()
    got = out.float().cpu().reshape(B, H, W, 8).permute(0, 3, 1, 2)
    assert rel_err(got, pred.grad) < BF16_TOL
    # AdamW: two steps against torch.optim.AdamW
    p0, g1, g2 = det("aw.p", (1000,), 7), det("aw.g1", (1000,), 8) * 0.1, det("aw.g2", (1000,), 9) * 0.1
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    for g in (g1, g2):
        pt.grad = g.clone()
        opt.step()
    pd, m, v = p0.clone().to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    for i, g in enumerate((g1, g2)):
        gd = g.to(DEV)
        N.check(L.ctta_adamw_step(N.ptr(pd), N.ptr(gd), N.ptr(m), N.ptr(v), 1000, 1e-3, 0.9, 0.999, 1e-8, 1e-2, i + 1, 1.0, st))
    sync()
    assert rel_err(pd.cpu(), pt.detach()) < 2e-6


@pytest.mark.parametrize("B,heads,dh,nq,nk,krows,use_bias", [(2, 3, 51, 200, 200, 200, False), (2, 2, 64, 256, 256, 256, False),
                                                            (3, 3, 13, 70, 7, 8, True), (1, 5, 51, 130, 31, 32, True),
                                                            (1, 2, 51, 2100, 20, 24, True),
                                                            (2, 6, 40, 4, 4, 4, False)])
def test_flash_attention_backward(B, heads, dh, nq, nk, krows, use_bias):
    """ctta_attention_lse + ctta_attention_bwd vs autograd of softmax(q k^T * scale + bias) v (fp32 on the
    bf16-rounded operands); heads padded to 64 lanes as in the engine."""
    L = lib()
    st = N.stream_ptr()
    hp = heads * 64
    scale = 1.0 / math.sqrt(dh)
    q = bf16_round(det("fa.q", (B, heads, nq, dh), 1)).requires_grad_(True)
    k = bf16_round(det("fa.k", (B, heads, nk, dh), 2)).requires_grad_(True)
    v = bf16_round(det("fa.v", (B, heads, nk, dh), 3)).requires_grad_(True)
    bias = None
    if use_bias:
        keep = (det("fa.m", (B, nk), 4) > -0.3).float()
        keep[:, 0] = 1.0
        bias = (1.0 - keep) * -10000.0
    s = torch.einsum("bhqd,bhkd->bhqk", q, k) * scale
    if bias is not None:
        s = s + bias[:, None, None, :]
    o = torch.einsum("bhqk,bhkd->bhqd", torch.softmax(s, dim=-1), v)
    do = bf16_round(det("fa.do", (B, heads, nq, dh), 5))
    o.backward(do)

    def pad_heads(x, rows):          # (B, heads, n, dh) -> [B][rows][hp] bf16
        out = torch.zeros(B, rows, hp)
        for h in range(heads):
            out[:, :x.shape[2], h * 64:h * 64 + dh] = x[:, h]
        return out.to(torch.bfloat16).to(DEV)

    qd, kd, vd, dod = pad_heads(q.detach(), nq), pad_heads(k.detach(), krows), pad_heads(v.detach(), krows), pad_heads(do, nq)
    vt_ld = rup(krows, 8)
    vt = torch.zeros(B, hp, vt_ld, dtype=torch.bfloat16, device=DEV)
    vt[:, :, :krows] = vd.transpose(1, 2)
    bd = bias.to(DEV).contiguous() if bias is not None else None
    out = torch.empty(B, nq, hp, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, heads, nq, device=DEV)
    N.check(L.ctta_attention_lse(N.ptr(qd), hp, N.ptr(kd), hp, krows, N.ptr(vt), vt_ld, N.ptr(bd), N.ptr(out), hp, B, heads,
                                 nq, nk, scale, N.ptr(lse), st))
    # transposed operands, as the engine builds them
    nk64, nq64 = rup(nk, 64), rup(nq, 64)
    vn = torch.empty(B, vt_ld, hp, dtype=torch.bfloat16, device=DEV)
    kt = torch.empty(B, hp, nk64, dtype=torch.bfloat16, device=DEV)
    qt = torch.empty(B, hp, nq64, dtype=torch.bfloat16, device=DEV)
    dot = torch.empty(B, hp, nq64, dtype=torch.bfloat16, device=DEV)
    N.check(L.ctta_transpose_bf16(N.ptr(vt), hp * vt_ld, hp, vt_ld, vt_ld, 0, N.ptr(vn), vt_ld * hp, hp, B, st))
    N.check(L.ctta_transpose_bf16(N.ptr(kd), krows * hp, nk, hp, hp, 0, N.ptr(kt), hp * nk64, nk64, B, st))
    N.check(L.ctta_transpose_bf16(N.ptr(qd), nq * hp, nq, hp, hp, 0, N.ptr(qt), hp * nq64, nq64, B, st))
    N.check(L.ctta_transpose_bf16(N.ptr(dod), nq * hp, nq, hp, hp, 0, N.ptr(dot), hp * nq64, nq64, B, st))
    dsum = torch.empty(B, heads, nq, device=DEV)
    dq = torch.full((B, nq, hp), float("nan"), dtype=torch.bfloat16, device=DEV)
    dk = torch.zeros(B, krows, hp, dtype=torch.bfloat16, device=DEV)
    dv = torch.zeros(B, krows, hp, dtype=torch.bfloat16, device=DEV)
    part = torch.empty(32 * 2 * B * krows * hp, device=DEV) if nk <= 128 else None   # query-split scratch (few keys)
    N.check(L.ctta_attention_bwd(N.ptr(qd), hp, N.ptr(kd), hp, krows, N.ptr(vn), hp, vt_ld, N.ptr(kt), nk64, N.ptr(qt),
                                 N.ptr(dot), nq64, N.ptr(bd), N.ptr(out), hp, N.ptr(dod), hp, N.ptr(lse), N.ptr(dsum),
                                 N.ptr(dq), hp, N.ptr(dk), hp, N.ptr(dv), hp, B, heads, nq, nk, scale, N.ptr(part),
                                 part.numel() if part is not None else 0, st))
    sync()

    def unpad(x, n):
        return torch.stack([x[:, :n, h * 64:h * 64 + dh].float().cpu() for h in range(heads)], dim=1)

    # log-sum-exp of the forward (log2 domain) and the forward output
    ref_lse = torch.logsumexp(s.detach(), dim=-1) * math.log2(math.e)
    assert float((lse.cpu() - ref_lse).abs().max()) < 2e-2
    assert rel_err(unpad(out, nq), o.detach()) < 2 * BF16_TOL
    for name, got, ref in (("dq", unpad(dq, nq), q.grad), ("dk", unpad(dk, nk), k.grad), ("dv", unpad(dv, nk), v.grad)):
        e = rel_err(got, ref)
        print("%s rel err %.3e" % (name, e))
        assert e < 3 * BF16_TOL, (name, e)
    # head-padding lanes stay exactly zero (they feed the projection data-gradient GEMMs)
    if dh < 64:
        assert float(dq[:, :, dh:64].float().abs().max()) == 0.0 and float(dk[:, :nk, dh:64].float().abs().max()) == 0.0
    # round 5: the same backward with q / k / dout read where they lie and V as the forward took it (no K^T, Q^T, dO^T,
    # natural-V copies): same MFMA products in the same order -> bit-identical gradients
    dq2 = torch.full((B, nq, hp), float("nan"), dtype=torch.bfloat16, device=DEV)
    dk2 = torch.zeros(B, krows, hp, dtype=torch.bfloat16, device=DEV)
    dv2 = torch.zeros(B, krows, hp, dtype=torch.bfloat16, device=DEV)
    if vt_ld > nk:
        vt[:, :, nk:] = float("nan")           # whatever lies beyond the valid keys must never reach a product
    dsum2 = torch.empty(B, heads, nq, device=DEV)
    N.check(L.ctta_attention_bwd_inplace(N.ptr(qd), hp, N.ptr(kd), hp, krows, N.ptr(vt), vt_ld, N.ptr(bd), N.ptr(out), hp,
                                         N.ptr(dod), hp, N.ptr(lse), N.ptr(dsum2), N.ptr(dq2), hp, N.ptr(dk2), hp, N.ptr(dv2), hp,
                                         B, heads, nq, nk, scale, N.ptr(part), part.numel() if part is not None else 0, st))
    sync()
    for name, a, b_ in (("dq", dq2, dq), ("dk", dk2[:, :nk], dk[:, :nk]), ("dv", dv2[:, :nk], dv[:, :nk])):
        assert bool(torch.isfinite(a.float()).all()), name
        e = rel_err(a.float().cpu(), b_.float().cpu())
        print("in-place %s vs transposed-copies route: rel diff %.3e" % (name, e))
        assert e <= 2e-3, (name, e)     # dP contracts over d in another slot order: fp32 sums differ in the last bits


@pytest.mark.parametrize("n_train,n_all,two", [(4096 * 37 + 64, 4096 * 37 + 320, True), (1 << 22, 1 << 22, True), (8192, 8200, False)])
def test_optimizer_tail_in_one_pass_is_bit_identical_to_the_three_launches(n_train, n_all, two):
    """ctta_adamw_ema2_zero = optimizer.step() -> optimizer.zero_grad() -> update_ema() (tools/train_utils.py:177-183, 255-282) as
    ONE pass over the training state: 12 fp32 streams instead of 7 + 1 + 5.  Same fp32 operations in the same order per element
    as ctta_adamw_step + fill + ctta_ema_update2, so after three steps (the middle one with do_step = 0: the NaN-loss skip)
    every buffer is BIT-identical; the frozen suffix [n_train, n_all) only moves its shadows and loses its gradient."""
    L = N.lib()
    st = N.stream_ptr()
    gen = torch.Generator().manual_seed(n_all)
    def mk(scale=1.0):
        return (torch.randn(n_all, generator=gen) * scale).to(DEV)
    p0, sa0, sb0 = mk(), mk(), mk()
    A = dict(p=p0.clone(), m=torch.zeros(n_train, device=DEV), v=torch.zeros(n_train, device=DEV), sa=sa0.clone(), sb=sb0.clone())
    B = {k: t.clone() for k, t in A.items()}
    for step in range(1, 4):
        g = mk(0.1)
        do = step != 2
        ga, gb = g.clone(), g.clone()
        if do:
            N.check(L.ctta_adamw_step(N.ptr(A["p"]), N.ptr(ga), N.ptr(A["m"]), N.ptr(A["v"]), n_train, 1e-3, 0.9, 0.999, 1e-8, 1e-2,
                                      step if step < 2 else step - 1, 0.5, st))
        ga.zero_()
        N.check(L.ctta_ema_update2(N.ptr(A["p"]), N.ptr(A["sa"]), 0.95, N.ptr(A["sb"]) if two else N.c_void_p(0), 0.999, n_all, st))
        N.check(L.ctta_adamw_ema2_zero(N.ptr(B["p"]), N.ptr(gb), N.ptr(B["m"]), N.ptr(B["v"]), n_train, n_all, N.ptr(B["sa"]), 0.95,
                                       N.ptr(B["sb"]) if two else N.c_void_p(0), 0.999, 1 if do else 0, 1e-3, 0.9, 0.999, 1e-8, 1e-2,
                                       step if step < 2 else step - 1, 0.5, st))
        torch.cuda.synchronize()
        assert not bool(gb.any())
        for k in A:
            if k == "sb" and not two:
                assert torch.equal(B[k], sb0)
                continue
            assert torch.equal(A[k], B[k]), (k, step)
    assert not torch.equal(A["p"], p0) and not torch.equal(A["sa"], sa0)
