"""GPU tier: the bf16x3 split-MFMA variants (RCOT_PREC_BF16X3) of the three GEMM-shaped entry points, against the
same fp64 statements as the exact-fp32 kernels (tests/test_kernels_gpu.py), plus evidence that the split kernels are
the ones that ran (results differ from the fp32 kernels' in the last bits, yet agree with fp64 to ~1e-5)."""
import pytest
import torch

import test_kernels_gpu as K
from conftest import relerr, seeded_tensor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hipx3():
    from rcot_amd import lib
    from rcot_amd.ops import HipBackend
    be = HipBackend()
    be.prec = lib.PREC_BF16X3
    return be


@pytest.mark.parametrize("B,Ci,Co,N", [(2, 96, 288, 1024), (1, 96, 510, 16384), (2, 255, 96, 256), (2, 48, 144, 2048),
                                       (1, 1021, 384, 256), (2, 384, 2042, 256), (2, 510, 96, 512), (3, 96, 96, 128),
                                       (8, 192, 510, 1024), (2, 127, 48, 384), (8, 96, 96, 4096), (2, 96, 288, 16384)])
@pytest.mark.parametrize("ln,res", [(False, False), (True, True)])
def test_x3_kmajor_conv1x1(hipx3, B, Ci, Co, N, ln, res):
    K.test_kmajor_conv1x1(hipx3, B, Ci, Co, N, ln, res)


@pytest.mark.parametrize("B,Ci,Co,N", [(2, 96, 288, 1024), (1, 96, 510, 16384), (2, 255, 96, 256), (2, 48, 144, 2048),
                                       (1, 1021, 384, 256), (2, 384, 2042, 256), (2, 510, 96, 512), (8, 192, 510, 1024),
                                       (8, 96, 96, 4096), (2, 96, 288, 16384), (8, 384, 1152, 256), (3, 100, 130, 768),
                                       (8, 96, 288, 1152)])
@pytest.mark.parametrize("ln,res", [(False, False), (True, False), (False, True), (True, True)])
def test_x3_presplit_conv1x1(hipx3, B, Ci, Co, N, ln, res):
    """the same statements with the PRE-SPLIT weight packs: N % 256 == 0 and more than 64 output rows run on the producer /
    consumer kernel (gemm_x3w.hip) — forward with the LN fold, residual + beta epilogues, K tails (255, 1021, 100), padded
    row tiles (510, 2042, 130 rows), split-K (256-pixel planes), the 128-column form (N = 1152) — everything else falls back to
    gemm_x3.hip."""
    K.test_kmajor_conv1x1(hipx3, B, Ci, Co, N, ln, res, split=True)


@pytest.mark.parametrize("B,heads,c,N", [(2, 1, 96, 16384), (2, 2, 48, 4096), (2, 4, 48, 1024), (1, 8, 48, 256), (2, 4, 96, 256),
                                         (2, 1, 48, 16384)])
def test_x3_kmajor_mdta_products(hipx3, B, heads, c, N):
    K.test_kmajor_mdta_products(hipx3, B, heads, c, N)


@pytest.mark.parametrize("B,Ci,Co,N", [(2, 48, 144, 256), (1, 255, 96, 1024), (2, 384, 2042, 64), (2, 96, 510, 4096),
                                       (8, 96, 288, 16384), (4, 192, 510, 1024), (2, 510, 96, 4096)])
def test_x3_conv1x1_wgrad(hipx3, B, Ci, Co, N):
    K.test_conv1x1_dgrad_wgrad(hipx3, B, Ci, Co, N)


@pytest.mark.parametrize("B,heads,c,N", [(2, 1, 48, 1024), (2, 2, 48, 256), (1, 4, 24, 256), (1, 1, 96, 4096), (2, 4, 96, 64),
                                         (2, 1, 96, 16384), (2, 8, 48, 256)])
def test_x3_gram_products(hipx3, B, heads, c, N):
    K.test_mdta_products(hipx3, B, heads, c, N)


def test_x3_is_the_split_kernel_and_close_to_fp32(hipx3):
    """Same inputs through prec = fp32 and prec = bf16x3: not bit-identical (the split kernels ran), 1e-5-close."""
    from rcot_amd import lib
    B, Ci, Co, N = 2, 96, 510, 4096
    W, X = seeded_tensor(1, (Co, Ci), scale=0.1).cuda(), seeded_tensor(2, (B, Ci, N)).cuda()
    dY = seeded_tensor(3, (B, Co, N)).cuda()
    WT, WP = (torch.zeros(*s, device="cuda") for s in hipx3.pack_shapes(Co, Ci))
    hipx3.pack_weight(W, WT, WP)
    outs = {}
    for name, prec in (("fp32", lib.PREC_FP32), ("x3", lib.PREC_BF16X3)):
        hipx3.prec = prec
        Y, dW = torch.zeros(B, Co, N, device="cuda"), torch.zeros(Co, Ci, device="cuda")
        hipx3.conv1x1_fwd(W, X, Y, packed=(WT, WP))
        hipx3.conv1x1_wgrad(dY, X, dW, beta=0.0)
        G = torch.zeros(B, 1, Ci, Ci, device="cuda")
        hipx3.bmm_nt(X.unsqueeze(1), X.unsqueeze(1), G)
        outs[name] = (Y, dW, G)
    hipx3.prec = lib.PREC_BF16X3
    torch.cuda.synchronize()
    for a, b in zip(outs["x3"], outs["fp32"]):
        assert not torch.equal(a, b)
        assert relerr(a, b) < 4e-5


@pytest.mark.parametrize("ratio", [1.0, 20.0])
def test_x3_ln_fold_with_large_pixel_mean(hipx3, ratio):
    """The LN fold of the bf16x3 kernels evaluates rs*(W' x) - rs*mu*c1 + c2: the two products cancel when a pixel's mean over
    channels is large against its spread.  Error against fp64 for |mu|/sigma = 1 and 20 (qkv shape of the 128x128 level, pre-split
    packs -> producer/consumer kernel): it grows with the ratio (~5e-6 * (1 + ratio)) and stays far inside the north_star's
    1e-3; the 188 LayerNorm inputs of the transport map measure |mu|/sigma <= 0.8 on every pixel at the seeded
    initialisation and after 10 iterations (scripts/ln_mean_ratio.py)."""
    B, Ci, Co, N = 2, 96, 288, 4096
    W, lw, lb = seeded_tensor(1, (Co, Ci), scale=0.1), 1 + 0.1 * seeded_tensor(3, (Ci,)), 0.1 * seeded_tensor(4, (Ci,))
    X = seeded_tensor(2, (B, Ci, N)) + ratio * (1 + 0.2 * seeded_tensor(12, (B, 1, N)))
    Xd = X.double()
    mu = Xd.mean(1, keepdim=True)
    xh = (Xd - mu) / (Xd.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt()
    ref = torch.einsum("oc,bcn->bon", W.double(), xh * lw.double().view(1, Ci, 1) + lb.double().view(1, Ci, 1))
    be = hipx3
    g = lambda t: t.cuda()
    Wg, Xg = g(W), g(X)
    WT, WP = (torch.zeros(*s, device="cuda") for s in be.pack_shapes(Co, Ci))
    WTf, c12 = (torch.zeros(*s, device="cuda") for s in be.fold_shapes(Co, Ci))
    (st,), (sp,) = be.split_shapes(Co, Ci)
    WTs, WPs, WTfs = torch.zeros(st, device="cuda"), torch.zeros(sp, device="cuda"), torch.zeros(st, device="cuda")
    be.pack_weight(Wg, WT, WP, (g(lw), g(lb), WTf, c12), (WTs, WPs, WTfs))
    mu_, rs_ = torch.zeros(B, N, device="cuda"), torch.zeros(B, N, device="cuda")
    be.ln_stats(Xg, mu_, rs_)
    Y = torch.zeros(B, Co, N, device="cuda")
    be.conv1x1_fwd(Wg, Xg, Y, ln=(mu_, rs_, g(lw), g(lb)), packed=(WT, WP, (WTf, c12), (WTs, WPs, WTfs)))
    torch.cuda.synchronize()
    e = relerr(Y, ref)
    print(f"LN-folded bf16x3 projection, |mu|/sigma = {ratio}: rel err {e:.2e}")
    assert e < 1e-5 * (1 + ratio)


@pytest.mark.parametrize("B,Ci,Co,N,ratio,fused", [(2, 96, 288, 16384, 1.0, True), (8, 192, 510, 1024, 1.0, True), (8, 96, 288, 1152, 1.0, True),
                                                   (2, 48, 144, 2048, 1.0, True), (1, 96, 510, 16384, 20.0, True), (8, 384, 1152, 256, 1.0, False),
                                                   (3, 100, 130, 768, 1.0, False), (16, 96, 288, 4096, 0.3, True)])
def test_x3_ln_statistics_made_by_the_projection(hipx3, B, Ci, Co, N, ratio, fused):
    """rcot_gemm_kmajor(ln_compute = 1): the producer wavefronts of the bf16x3 projection kernel form the per-pixel LayerNorm
    statistics from the rows they split; the epilogue uses them and writes them out.  mu / rstd and the projection against fp64
    (mean comparable to and 20x the spread; 256- and 128-column tiles; several row tiles; more tiles than resident workgroups).
    Split reductions (16x16 level) and K % 16 != 0 are declined and conv1x1_fwd runs rcot_ln_stats first — same results."""
    W, lw, lb = seeded_tensor(1, (Co, Ci), scale=0.1), 1 + 0.1 * seeded_tensor(3, (Ci,)), 0.1 * seeded_tensor(4, (Ci,))
    X = seeded_tensor(2, (B, Ci, N)) + ratio * (1 + 0.2 * seeded_tensor(12, (B, 1, N)))
    Xd = X.double()
    mu = Xd.mean(1, keepdim=True)
    rstd = (Xd.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    ref = torch.einsum("oc,bcn->bon", W.double(), (Xd - mu) * rstd * lw.double().view(1, Ci, 1) + lb.double().view(1, Ci, 1))
    be = hipx3
    g = lambda t: t.cuda()
    Wg, Xg = g(W), g(X)
    WT, WP = (torch.zeros(*s, device="cuda") for s in be.pack_shapes(Co, Ci))
    WTf, c12 = (torch.zeros(*s, device="cuda") for s in be.fold_shapes(Co, Ci))
    (st,), (sp,) = be.split_shapes(Co, Ci)
    WTs, WPs, WTfs = torch.zeros(st, device="cuda"), torch.zeros(sp, device="cuda"), torch.zeros(st, device="cuda")
    be.pack_weight(Wg, WT, WP, (g(lw), g(lb), WTf, c12), (WTs, WPs, WTfs))
    mu_, rs_ = torch.full((B, N), float("nan"), device="cuda"), torch.full((B, N), float("nan"), device="cuda")
    Y = torch.full((B, Co, N), float("nan"), device="cuda")
    calls = []
    orig = be.ln_stats
    be.ln_stats = lambda *a: (calls.append(1), orig(*a))
    try:
        be.conv1x1_fwd(Wg, Xg, Y, ln=(mu_, rs_, g(lw), g(lb)), packed=(WT, WP, (WTf, c12), (WTs, WPs, WTfs)), ln_compute=True)
    finally:
        del be.ln_stats
    torch.cuda.synchronize()
    assert (len(calls) == 0) == fused, "which path made the statistics"
    e_mu = float((mu_.double().cpu() - mu[:, 0]).abs().max() / mu.abs().max())
    e_rs = float((rs_.double().cpu() / rstd[:, 0] - 1).abs().max())
    e = relerr(Y, ref)
    print(f"fused={fused} |mu|/sigma={ratio}: mu {e_mu:.1e} rstd {e_rs:.1e} projection {e:.2e}")
    assert e_mu < 2e-6 and e_rs < 2e-5 * (1 + ratio) and e < 1e-5 * (1 + ratio)


@pytest.mark.parametrize("B,Ci,Co,N", [(8, 96, 510, 4096), (8, 255, 96, 4096), (8, 192, 1020, 1024), (8, 510, 192, 1024), (8, 384, 2042, 256),
                                       (8, 1021, 384, 256), (8, 384, 1152, 256), (8, 96, 288, 2048), (8, 96, 255, 1152), (2, 48, 144, 1024)])
@pytest.mark.parametrize("ln", [False, True])
def test_x3_paired_dgrad_wgrad(hipx3, B, Ci, Co, N, ln):
    """rcot_conv1x1_dgrad_wgrad_slabs: the data gradient and the weight-gradient slabs of one dY from ONE launch (workgroups of both
    products in one grid) against fp64 — every projection shape of the 64x64 / 32x32 / 16x16 levels (row tiles of 255 / 510 / 1021
    / 2042 rows, K tails, split-K on both sides), a 128-column plane (N = 1152); (48, 144): the data
    gradient has 48 rows, no paired kernel -> None and the separate entry points."""
    be = hipx3
    W, dY, X = seeded_tensor(1, (Co, Ci), scale=0.1), seeded_tensor(2, (B, Co, N)), seeded_tensor(3, (B, Ci, N)) + 0.5
    lw, lb = 1 + 0.1 * seeded_tensor(4, (Ci,)), 0.1 * seeded_tensor(5, (Ci,))
    g = lambda t: t.cuda()
    Wg, dYg, Xg = g(W), g(dY), g(X)
    WT, WP = (torch.zeros(*s, device="cuda") for s in be.pack_shapes(Co, Ci))
    (st,), (sp,) = be.split_shapes(Co, Ci)
    WTs, WPs = torch.zeros(st, device="cuda"), torch.zeros(sp, device="cuda")
    be.pack_weight(Wg, WT, WP, None, (WTs, WPs, None))
    mu, rs = torch.zeros(B, N, device="cuda"), torch.ones(B, N, device="cuda")
    be.ln_stats(Xg, mu, rs)
    lnarg = (mu, rs, g(lw), g(lb)) if ln else None
    dX = torch.full((B, Ci, N), float("nan"), device="cuda")
    gW0 = seeded_tensor(6, (Co, Ci))
    gW = g(gW0)
    d = be.conv1x1_dgrad_wgrad_slabs(Wg, dYg, dX, Xg, gW, ln=lnarg, packed=(WT, WP, None, (WTs, WPs, None)), region=(1, 3))
    if Ci <= 64:
        assert d is None
        return
    assert d is not None
    # the launch that closes a block sums the slabs into gW (one row of zero LayerNorm partials, zero dW_o / dtau parts)
    zc = lambda *sh: torch.zeros(*sh, device="cuda")
    for sc in be._ln_scratch[be._gen]:
        sc[:2 * Ci].zero_()
    be._ln_rows = 1
    be.block_param_reduce(Ci, zc(Ci), zc(Ci), zc(Ci), zc(Ci), zc(B, Ci, Ci), zc(Ci, Ci), zc(B, 1), zc(1), [d])
    torch.cuda.synchronize()
    Xd = X.double()
    if ln:
        m = Xd.mean(1, keepdim=True)
        r = (Xd.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
        Xd = (Xd - m) * r * lw.double().view(1, Ci, 1) + lb.double().view(1, Ci, 1)
    ref_dX = torch.einsum("oc,bon->bcn", W.double(), dY.double())
    ref_gW = gW0.double() + torch.einsum("bon,bcn->oc", dY.double(), Xd)
    e1, e2 = relerr(dX, ref_dX), relerr(gW, ref_gW)
    print(f"paired dgrad/wgrad B={B} {Co}x{Ci} N={N} ln={ln}: dX {e1:.2e}  dW {e2:.2e}")
    assert e1 < 4e-5 and e2 < 4e-5


# ----------------------------------------------------------------------------- bf16x6: fp32-class results from the bf16 pipe
@pytest.fixture(scope="module")
def hipx6():
    from rcot_amd import lib
    from rcot_amd.ops import HipBackend
    be = HipBackend()
    be.prec = lib.PREC_BF16X6
    be.x6_packs = True
    return be


@pytest.mark.parametrize("B,Ci,Co,N", [(2, 96, 288, 1024), (1, 96, 510, 16384), (2, 255, 96, 256), (1, 1021, 384, 256), (2, 384, 2042, 256),
                                       (8, 192, 510, 1024), (8, 96, 96, 4096), (2, 96, 288, 16384), (8, 384, 1152, 256), (3, 100, 130, 768),
                                       (8, 96, 288, 1152)])
@pytest.mark.parametrize("ln,res", [(False, False), (True, False), (False, True), (True, True)])
def test_x6_presplit_conv1x1(hipx6, B, Ci, Co, N, ln, res):
    """RCOT_PREC_BF16X6 (three-term split, six products) on the producer / consumer kernel, at the tolerance of the EXACT-fp32
    kernels (2e-5 of max|C| vs fp64 — the bf16x3 bar is 4e-5): forward with the LN fold, residual + beta, K tails, padded row
    tiles, split-K, 128-column planes."""
    K.test_kmajor_conv1x1(hipx6, B, Ci, Co, N, ln, res, split=True, six=True, tol=K.TOL)


@pytest.mark.parametrize("B,Ci,Co,N", [(2, 48, 144, 256), (1, 255, 96, 1024), (2, 384, 2042, 64), (2, 96, 510, 4096),
                                       (8, 96, 288, 16384), (4, 192, 510, 1024), (2, 510, 96, 4096)])
def test_x6_conv1x1_wgrad(hipx6, B, Ci, Co, N):
    """the pixel-reduction products (gemm_nt_kernel<.., X3, X6>: weight gradients with the LayerNorm recomputed in the loop), fp32 bar"""
    K.test_conv1x1_dgrad_wgrad(hipx6, B, Ci, Co, N)


@pytest.mark.parametrize("B,heads,c,N", [(2, 1, 48, 1024), (2, 2, 48, 256), (1, 4, 24, 256), (1, 1, 96, 4096), (2, 4, 96, 64),
                                         (2, 1, 96, 16384), (2, 8, 48, 256)])
def test_x6_gram_products(hipx6, B, heads, c, N):
    K.test_mdta_products(hipx6, B, heads, c, N)


def test_x6_pixel_reductions_are_the_split_kernels_and_fp32_class(hipx6):
    """weight gradient and Gram matrix in the three arithmetics against fp64: bf16x6 differs from the exact-fp32 kernel in the
    last bits (the split kernel ran), is as close to fp64 as it, and several times closer than bf16x3."""
    from rcot_amd import lib
    be = hipx6
    B, Ci, Co, N = 4, 96, 510, 4096
    X, dY = seeded_tensor(2, (B, Ci, N)) + 0.5, seeded_tensor(3, (B, Co, N))
    ref_w = torch.einsum("bon,bcn->oc", dY.double(), X.double())
    ref_g = torch.einsum("bcn,bdn->bcd", X.double(), X.double()).unsqueeze(1)
    Xg, dYg = X.cuda(), dY.cuda()
    errs, outs = {}, {}
    for name, prec in (("fp32", lib.PREC_FP32), ("bf16x3", lib.PREC_BF16X3), ("bf16x6", lib.PREC_BF16X6)):
        be.prec = prec
        dW, G = torch.zeros(Co, Ci, device="cuda"), torch.zeros(B, 1, Ci, Ci, device="cuda")
        be.conv1x1_wgrad(dYg, Xg, dW, beta=0.0)
        be.bmm_nt(Xg.unsqueeze(1), Xg.unsqueeze(1), G)
        torch.cuda.synchronize()
        errs[name], outs[name] = (relerr(dW, ref_w), relerr(G, ref_g)), (dW, G)
    be.prec = lib.PREC_BF16X6
    print("max|C - fp64| / max|C| (weight gradient, Gram): " + "  ".join(f"{k} {v[0]:.2e} {v[1]:.2e}" for k, v in errs.items()))
    for i in range(2):
        assert errs["bf16x6"][i] <= 1.5 * errs["fp32"][i] + 5e-8
        assert errs["bf16x3"][i] > 3 * errs["bf16x6"][i]
        assert not torch.equal(outs["bf16x6"][i], outs["fp32"][i]) and not torch.equal(outs["bf16x6"][i], outs["bf16x3"][i])


@pytest.mark.parametrize("B,Ci,Co,N", [(8, 96, 510, 4096), (8, 192, 1020, 1024), (2, 96, 288, 16384), (8, 1021, 384, 256)])
def test_x6_is_as_accurate_as_the_fp32_kernel(hipx6, B, Ci, Co, N):
    """The claim behind "fp32-class": on the same operands the bf16x6 product is no further from fp64 than the exact-fp32 MFMA
    kernel's (whose own error is the rounding of its fp32 accumulation), and ~10x closer than bf16x3; and it really is the
    split kernel that ran (its result differs from the fp32 kernel's in the last bits)."""
    from rcot_amd import lib
    be = hipx6
    W, X = seeded_tensor(1, (Co, Ci), scale=0.1), seeded_tensor(2, (B, Ci, N)) + 0.5
    ref = torch.einsum("oc,bcn->bon", W.double(), X.double())
    Wg, Xg = W.cuda(), X.cuda()
    WT, WP = (torch.zeros(*s, device="cuda") for s in be.pack_shapes(Co, Ci))
    (st,), (sp,) = be.split_shapes(Co, Ci)
    (st6,), (sp6,) = be.split6_shapes(Co, Ci)
    s3 = (torch.zeros(st, device="cuda"), torch.zeros(sp, device="cuda"), None)
    s6 = (torch.zeros(st6, device="cuda"), torch.zeros(sp6, device="cuda"), None)
    be.pack_weight(Wg, WT, WP, None, s3, s6)
    errs, outs = {}, {}
    for name, prec in (("fp32", lib.PREC_FP32), ("bf16x3", lib.PREC_BF16X3), ("bf16x6", lib.PREC_BF16X6)):
        be.prec = prec
        Y = torch.full((B, Co, N), float("nan"), device="cuda")
        be.conv1x1_fwd(Wg, Xg, Y, packed=(WT, WP, None, s3, s6))
        torch.cuda.synchronize()
        errs[name], outs[name] = relerr(Y, ref), Y
    be.prec = lib.PREC_BF16X6
    print(f"{Co}x{Ci} N={N}: max|C - fp64| / max|C|  fp32 kernel {errs['fp32']:.2e}  bf16x6 {errs['bf16x6']:.2e}  bf16x3 {errs['bf16x3']:.2e}")
    assert errs["bf16x6"] <= 1.5 * errs["fp32"] + 5e-8
    assert errs["bf16x3"] > 4 * errs["bf16x6"]
    assert not torch.equal(outs["bf16x6"], outs["fp32"]) and not torch.equal(outs["bf16x6"], outs["bf16x3"])


@pytest.mark.parametrize("prec_name", ["bf16x6", "bf16x3"])
@pytest.mark.parametrize("B,Ci,Co,N,ln", [(8, 96, 510, 16384, True), (2, 96, 288, 4096, True), (2, 255, 96, 4096, False), (2, 48, 144, 1024, True),
                                          (2, 127, 48, 1024, False)])
def test_cooperative_split_changes_no_bit(hipx6, prec_name, B, Ci, Co, N, ln):
    """Round 5: the pixel-reduction kernel normalises / splits the operand that all four wavefronts of a tile share once per workgroup
    (gemm_nt_body.h COOP; tiles 128 x 96, 96 x 128, 128 x 64, 64 x 128).  The same operations on the same values as the per-wavefront
    split: the weight gradient (with and without the LayerNorm in the loop) and a Gram product are the SAME BITS with the split off
    (rcot_debug_nt_coop(0): what RCOT_NT_COOP = 0 selects)."""
    from rcot_amd import lib
    be = hipx6
    be.prec = {"bf16x6": lib.PREC_BF16X6, "bf16x3": lib.PREC_BF16X3}[prec_name]
    try:
        X, dY = (seeded_tensor(2, (B, Ci, N)) + 0.5).cuda(), seeded_tensor(3, (B, Co, N)).cuda()
        lw, lb = (1 + 0.1 * seeded_tensor(4, (Ci,))).cuda(), (0.1 * seeded_tensor(5, (Ci,))).cuda()
        mu, rs = torch.empty(B, N, device="cuda"), torch.empty(B, N, device="cuda")
        be.ln_stats(X, mu, rs)
        outs = []
        try:
            for c in (3, 0):
                assert be.L.rcot_debug_nt_coop(c) == 0       # (the environment's RCOT_NT_COOP is read once per process: the test hook switches)
                dW, G = torch.zeros(Co, Ci, device="cuda"), torch.zeros(B, 1, Ci, Ci, device="cuda")
                be.conv1x1_wgrad(dY, X, dW, ln=(mu, rs, lw, lb) if ln else None, beta=0.0)
                be.bmm_nt(X.unsqueeze(1), X.unsqueeze(1), G)
                torch.cuda.synchronize()
                outs.append((dW, G))
        finally:
            be.L.rcot_debug_nt_coop(-1)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert bool(torch.isfinite(outs[0][0]).all()) and float(outs[0][0].abs().max()) > 0
    finally:
        be.prec = lib.PREC_BF16X6


# ----------------------------------------------------------------------------- bf16x1: the single product (opt-in, round 5)
def _bf(t):
    """rne to bfloat16, back in fp64: the operand a single-product kernel multiplies"""
    return t.float().bfloat16().double()


@pytest.fixture(scope="module")
def hipx1():
    from rcot_amd import lib
    from rcot_amd.ops import HipBackend
    be = HipBackend()
    be.prec = lib.PREC_BF16X1
    return be


@pytest.mark.parametrize("B,Ci,Co,N", [(2, 96, 288, 1024), (1, 96, 510, 16384), (8, 384, 2042, 256), (2, 255, 96, 256), (8, 192, 510, 1024), (2, 48, 144, 2048)])
def test_x1_projection_is_the_single_bf16_product(hipx1, hipx3, B, Ci, Co, N):
    """RCOT_PREC_BF16X1 on the producer / consumer and general split kernels: forward and data gradient equal the fp64 product of the
    bf16-ROUNDED operands to fp32-accumulation accuracy (so exactly one product term is evaluated), differ from the three-product
    result, and sit ~2^-9 from the exact product."""
    be = hipx1
    W, X, dY = seeded_tensor(1, (Co, Ci), scale=0.1), seeded_tensor(2, (B, Ci, N)), seeded_tensor(3, (B, Co, N))
    g = lambda t: t.cuda()
    Wg, Xg, dYg = g(W), g(X), g(dY)
    WT, WP = (torch.zeros(*s, device="cuda") for s in be.pack_shapes(Co, Ci))
    (st,), (sp,) = be.split_shapes(Co, Ci)
    WTs, WPs = torch.zeros(st, device="cuda"), torch.zeros(sp, device="cuda")
    be.pack_weight(Wg, WT, WP, None, (WTs, WPs, None))
    packed = (WT, WP, None, (WTs, WPs, None))
    outs = {}
    for name, b in (("x1", hipx1), ("x3", hipx3)):
        Y, dX = torch.full((B, Co, N), float("nan"), device="cuda"), torch.full((B, Ci, N), float("nan"), device="cuda")
        b.conv1x1_fwd(Wg, Xg, Y, packed=packed)
        b.conv1x1_dgrad(Wg, dYg, dX, packed=packed)
        torch.cuda.synchronize()
        outs[name] = (Y, dX)
    emu_y = torch.einsum("oc,bcn->bon", _bf(W), _bf(X))
    emu_dx = torch.einsum("oc,bon->bcn", _bf(W), _bf(dY))
    ex_y = torch.einsum("oc,bcn->bon", W.double(), X.double())
    kmajor = be.kmajor_worth(Co, N, B)
    for got, emu, exact, got3 in ((outs["x1"][0], emu_y, ex_y, outs["x3"][0]),):
        if not kmajor:
            continue                                  # (shapes below the K-major kernels' limits run exact fp32 in every arithmetic)
        assert relerr(got, emu) < 2e-5, relerr(got, emu)
        assert 2e-4 < relerr(got, exact) < 2e-2, relerr(got, exact)
        assert not torch.equal(got, got3)
    if be.kmajor_worth(Ci, N, B):
        assert relerr(outs["x1"][1], emu_dx) < 2e-5


@pytest.mark.parametrize("B,Ci,Co,N", [(8, 96, 510, 4096), (2, 96, 288, 16384), (8, 384, 1152, 256)])
def test_x1_weight_gradient_and_gram_are_single_products(hipx1, B, Ci, Co, N):
    be = hipx1
    X, dY = seeded_tensor(2, (B, Ci, N)), seeded_tensor(3, (B, Co, N))
    dW = torch.zeros(Co, Ci, device="cuda")
    be.conv1x1_wgrad(dY.cuda(), X.cuda(), dW, beta=0.0)
    G = torch.full((B, 1, Ci, Ci), float("nan"), device="cuda")
    be.bmm_nt(X.cuda().view(B, 1, Ci, N), X.cuda().view(B, 1, Ci, N), G)
    torch.cuda.synchronize()
    assert relerr(dW, torch.einsum("bon,bcn->oc", _bf(dY), _bf(X))) < 2e-5
    assert relerr(G[:, 0], torch.einsum("bin,bjn->bij", _bf(X), _bf(X))) < 2e-5
