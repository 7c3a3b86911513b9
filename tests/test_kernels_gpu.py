"""GPU tier: every C-ABI entry point of librcot_hip.so, called through rcot_amd.ops.HipBackend, against an
independent fp64 CPU statement of the same operation (tests/host_double.py, itself verified against the
oracle by the CPU tier).  Shapes are the ones the transport map / critic actually use (odd hidden sizes
127/255/510/1021, heads 1/2/4/8, 24/48/96 channels per head, k5s1/k4s2/k3s1 convs).
Tolerance: fp32 rounding only (the MFMA path is an exact fp32 fmaf chain): 2e-5 relative to max|ref|.
"""
import os

import numpy as np
import pytest
import torch

from conftest import relerr, seeded_tensor
from host_double import TorchDouble

pytestmark = pytest.mark.gpu
TOL = 2e-5
X3_TOL = 4e-5        # bf16x3 split products: ~5e-6 of max|C| expected (scripts/micro/bf16x3_error.py), 1e-3 is the north_star bar


@pytest.fixture(scope="module")
def hip():
    from rcot_amd.ops import HipBackend
    return HipBackend()


DBL = TorchDouble(torch.float64)


def T(seed, *shape, scale=1.0):
    return seeded_tensor(seed, shape, scale=scale)


def both(hip, fn, arrays, outs, tol=TOL):
    """Run fn(backend, *tensors) on the double (fp64 CPU) and on HIP (fp32 GPU); compare tensors[outs]."""
    if getattr(hip, "prec", 0) == 1:
        tol = max(tol, X3_TOL)
    cpu = [None if a is None else a.double().clone() for a in arrays]
    gpu = [None if a is None else a.cuda() for a in arrays]
    fn(DBL, *cpu)
    fn(hip, *gpu)
    torch.cuda.synchronize()
    for i in outs:
        e = relerr(gpu[i], cpu[i])
        assert e < tol, (i, e)


# ----------------------------------------------------------------------------- 1x1 projections
@pytest.mark.parametrize("B,Ci,Co,N", [(2, 48, 144, 256), (1, 96, 510, 1024), (2, 255, 96, 320), (1, 384, 2042, 64),
                                       (2, 1021, 384, 64), (3, 96, 96, 16384)])
@pytest.mark.parametrize("ln,res", [(False, False), (True, True)])
def test_conv1x1_fwd(hip, B, Ci, Co, N, ln, res):
    def fn(be, W, X, Y, mu, rs, lw, lb, R):
        if ln:
            be.ln_stats(X, mu, rs)
        be.conv1x1_fwd(W, X, Y, ln=(mu, rs, lw, lb) if ln else None, R=R if res else None)
    arrs = [T(1, Co, Ci, scale=0.1), T(2, B, Ci, N), torch.zeros(B, Co, N), torch.zeros(B, N), torch.zeros(B, N),
            1 + 0.1 * T(3, Ci), 0.1 * T(4, Ci), T(5, B, Co, N)]
    both(hip, fn, arrs, [2])


def test_conv1x1_two_source_and_slices(hip):
    """cat-free reduce: W[:, :C1] x1 + W[:, C1:] x2 with beta accumulation and channel-sliced operands."""
    B, C1, C2, Co, N = 2, 96, 192, 192, 256

    def fn(be, W, big, Y):
        x1, x2 = big[:, :C1], big[:, C1:]
        be.conv1x1_fwd(W[:, :C1], x1, Y)
        be.conv1x1_fwd(W[:, C1:], x2, Y, beta=1.0)
    both(hip, fn, [T(1, Co, C1 + C2, scale=0.1), T(2, B, C1 + C2, N), torch.zeros(B, Co, N)], [2])


@pytest.mark.parametrize("B,Ci,Co,N", [(2, 48, 144, 256), (1, 255, 96, 1024), (2, 384, 2042, 64), (2, 96, 510, 4096),
                                       (8, 96, 288, 16384)])      # the last: 96-row big-tile dispatch of the unpacked dgrad
def test_conv1x1_dgrad_wgrad(hip, B, Ci, Co, N):
    def fn(be, W, dY, X, dX, dW, mu, rs, lw, lb, dX2, dW2):
        be.conv1x1_dgrad(W, dY, dX)
        be.conv1x1_dgrad(W, dY, dX2, beta=1.0)
        be.conv1x1_wgrad(dY, X, dW, beta=0.0)
        be.ln_stats(X, mu, rs)
        be.conv1x1_wgrad(dY, X, dW2, ln=(mu, rs, lw, lb), beta=1.0)
    arrs = [T(1, Co, Ci, scale=0.1), T(2, B, Co, N), T(3, B, Ci, N), torch.zeros(B, Ci, N), torch.zeros(Co, Ci),
            torch.zeros(B, N), torch.zeros(B, N), 1 + 0.1 * T(4, Ci), 0.1 * T(5, Ci), T(6, B, Ci, N), T(7, Co, Ci)]
    both(hip, fn, arrs, [3, 4, 9, 10])


# ----------------------------------------------------------------------------- batched small-matrix products
@pytest.mark.parametrize("B,heads,c,N", [(2, 1, 48, 1024), (2, 2, 48, 256), (1, 4, 24, 256), (2, 8, 48, 64), (1, 1, 96, 4096), (2, 4, 96, 64)])
def test_mdta_products(hip, B, heads, c, N):
    C = heads * c

    def fn(be, u, G, M, y, x, dM, du, Eq, Dq):
        uu = u.view(B, 3, heads, c, N)
        Q, K, V = uu[:, 0], uu[:, 1], u.view(B, 3, C, N)[:, 2].unsqueeze(1)
        be.bmm_nt(Q, K, G)                                                   # Gram over pixels per head
        be.bmm_nn(M.unsqueeze(1), V, y.view(B, 1, C, N), R=x.view(B, 1, C, N))   # attention apply + residual
        be.bmm_nt(y.view(B, 1, C, N), V, dM.unsqueeze(1))                     # dM = dY V^T
        dd = du.view(B, 3, heads, c, N)
        be.bmm_nn(M.unsqueeze(1), y.view(B, 1, C, N), du.view(B, 3, C, N)[:, 2].unsqueeze(1), transA=True)
        be.bmm_nn(Eq, K, dd[:, 0], R=Q, rowscale=Dq.view(B, heads, c))
        be.bmm_nn(Eq, Q, dd[:, 1], transA=True, R=K, rowscale=Dq.view(B, heads, c))
    arrs = [T(1, B, 3 * C, N), torch.zeros(B, heads, c, c), T(2, B, C, C, scale=0.2), torch.zeros(B, C, N), T(3, B, C, N),
            torch.zeros(B, C, C), torch.zeros(B, 3 * C, N), T(4, B, heads, c, c, scale=0.2), T(5, B, C)]
    both(hip, fn, arrs, [1, 3, 5, 6])


@pytest.mark.parametrize("B,heads,c", [(2, 1, 48), (2, 2, 48), (1, 4, 24), (2, 8, 48), (2, 1, 96), (1, 4, 96)])
def test_attn_small(hip, B, heads, c):
    """softmax core + the W_o folds done as per-(image, head) GEMMs on strided head-column views."""
    C = heads * c

    def fn(be, Graw, sq, temp, Wo, Gn, A, Mf, dM, dWp, dtp, Eq, Dq, Dk, dWo, dtemp, dA, EqT):
        wo = Wo.view(C, heads, c).permute(1, 0, 2).unsqueeze(0).expand(B, -1, -1, -1)
        cols = lambda M: M.view(B, C, heads, c).permute(0, 2, 1, 3)
        be.attn_softmax(Graw, sq, temp, Gn, A)
        be.bmm_nn(wo, A, cols(Mf))
        be.bmm_nn(wo, cols(dM), dA, transA=True)
        be.bmm_nt(cols(dM), A, cols(dWp))
        be.attn_bwd_small(dA, A, Gn, sq, temp, dtp, Eq, EqT, Dq, Dk)
        be.batch_reduce(dWp, dWo, beta=1.0)
        be.batch_reduce(dtp, dtemp, beta=0.0)
    sq = T(2, B, 2 * C).abs() * 50 + 1.0
    arrs = [T(1, B, heads, c, c, scale=5.0), sq, 1 + 0.2 * T(3, heads), T(4, C, C, scale=0.1)] + \
        [torch.zeros(B, heads, c, c), torch.zeros(B, heads, c, c), torch.zeros(B, C, C), T(5, B, C, C),
         torch.zeros(B, C, C), torch.zeros(B, heads), torch.zeros(B, heads, c, c), torch.zeros(B, C), torch.zeros(B, C),
         T(6, C, C), torch.zeros(heads), torch.zeros(B, heads, c, c), torch.zeros(B, heads, c, c)]
    both(hip, fn, arrs, [4, 5, 6, 10, 11, 12, 13, 14, 15, 16], tol=5e-5)


@pytest.mark.parametrize("B,heads,c,N", [(2, 1, 48, 16384), (2, 2, 48, 4096), (1, 8, 48, 256), (2, 1, 96, 4096), (2, 4, 24, 1024),
                                         (8, 1, 48, 16384)])
def test_gram_slabs_into_softmax(hip, B, heads, c, N):
    """q k^T left as split-K slabs and summed inside the softmax kernel == reduce launch + dense softmax input (c = 24 has no
    slab kernel: the caller's fallback)."""
    C = heads * c

    def fn(be, u, sq, temp, Gn, A):
        uu = u.view(B, 3, heads, c, N)
        Q, K = uu[:, 0], uu[:, 1]
        be.row_sumsq(u[:, :2 * C], sq)
        G = be.bmm_nt_slabs(Q, K)
        if c >= 33:
            assert G is not None
        if G is None:
            G = torch.zeros_like(Gn)
            be.bmm_nt(Q, K, G)
        be.attn_softmax(G, sq, temp, Gn, A)
    arrs = [T(1, B, 3 * C, N), torch.zeros(B, 2 * C), 1 + 0.2 * T(3, heads), torch.zeros(B, heads, c, c), torch.zeros(B, heads, c, c)]
    both(hip, fn, arrs, [3, 4], tol=5e-5)


@pytest.mark.parametrize("B,heads,c,N", [(2, 8, 48, 256), (8, 8, 48, 256), (1, 4, 96, 256), (2, 4, 48, 1024), (2, 2, 48, 4096),
                                         (2, 4, 24, 4096), (1, 1, 96, 1024), (2, 1, 48, 16384), (1, 4, 48, 1280), (1, 2, 48, 3840), (1, 2, 48, 2304),
                                         (8, 1, 96, 16384), (1, 4, 24, 16384), (1, 1, 96, 65536), (1, 1, 48, 15360), (1, 2, 48, 4352)])
def test_attn_core_fwd(hip, B, heads, c, N):
    """sq, Gn, A and the folded operand (W_o blockdiag(A))^T from u in one or two launches == the four separate launches' results;
    the kernels also take the 128x128 level (32 pixel ranges of 512) and 256x256 patches (ranges of 2048) — measured slower there than
    the caller's four-launch route, so HipBackend.attn_core_fwd stops at ``attn_core_maxn`` = 4096 pixels by default: raised here."""
    C = heads * c

    def fn(be, u, temp, WoT, sq, Gn, A, MfT):
        be.attn_core_maxn = 65536
        try:
            ok = be.attn_core_fwd(u.view(B, 3 * C, 16, N // 16), temp, WoT, sq, Gn, A, MfT)
        finally:
            be.attn_core_maxn = 4096
        assert ok == (N <= 4096 or N % 512 == 0)
        if not ok:
            for t in (sq, Gn, A, MfT):
                t.zero_()
    r16, r4 = (C + 15) // 16 * 16, (C + 3) // 4 * 4
    WoT = torch.zeros(r16, r4)
    WoT[:C, :C] = T(4, C, C, scale=0.1)
    arrs = [T(1, B, 3 * C, N), 1 + 0.2 * T(3, heads), WoT, torch.zeros(B, 2 * C), torch.zeros(B, heads, c, c),
            torch.zeros(B, heads, c, c), torch.zeros(B, C, C)]
    both(hip, fn, arrs, [3, 4, 5, 6], tol=2e-5)


@pytest.mark.parametrize("B,heads,c,N", [(8, 8, 48, 256), (8, 4, 48, 1024), (8, 2, 48, 4096), (2, 1, 48, 16384), (2, 1, 96, 16384), (8, 4, 96, 256)])
def test_kmajor_multi_equals_three_launches(hip, B, heads, c, N):
    """rcot_gemm_kmajor_multi (round 5): dV = Mf^T dY, dQ = Eq K + Dq.Q, dK = Eq^T Q + Dk.K of one MDTA block from ONE launch, each
    bit-identical to its own rcot_gemm_kmajor launch (all three tile shapes), and the launch really is one kernel."""
    import ctypes
    be, C = hip, heads * c
    g = lambda seed, *sh: seeded_tensor(seed, sh).cuda()
    u, dy = g(1, B, 3 * C, N), g(2, B, C, N)
    Mf, Eq, EqT = g(3, B, C, C), g(4, B, heads, c, c), g(5, B, heads, c, c)
    Dq, Dk = g(6, B, C), g(7, B, C)
    uu = u.view(B, 3, heads, c, N)
    Q, K = uu[:, 0], uu[:, 1]
    dy4 = dy.view(B, 1, C, N)
    outs = []
    for multi in (True, False):
        du = torch.full((B, 3 * C, N), float("nan"), device="cuda")
        dd = du.view(B, 3, heads, c, N)
        dQ, dK, dV = dd[:, 0], dd[:, 1], du.view(B, 3, C, N)[:, 2].unsqueeze(1)
        if multi:
            assert be.gemm_kmajor_multi([(Mf.unsqueeze(1), dy4, dV, C, C, None, None), (EqT, K, dQ, c, c, Q, Dq.view(B, heads, c)),
                                         (Eq, Q, dK, c, c, K, Dk.view(B, heads, c))])
            buf = ctypes.create_string_buffer(192)
            be.L.rcot_last_kernel(buf, 192)
            assert buf.value.decode().startswith("gemm_xx_multi_kernel"), buf.value
        else:
            be.gemm_kmajor(Mf.unsqueeze(1), dy4, dV, C, C)
            be.gemm_kmajor(EqT, K, dQ, c, c, R=Q, rowscale=Dq.view(B, heads, c))
            be.gemm_kmajor(Eq, Q, dK, c, c, R=K, rowscale=Dk.view(B, heads, c))
        torch.cuda.synchronize()
        outs.append(du)
    assert torch.equal(outs[0], outs[1])
    ref = torch.einsum("bkm,bkn->bmn", Mf.double().cpu(), dy.double().cpu())
    assert relerr(outs[0].view(B, 3, C, N)[:, 2], ref) < TOL


@pytest.mark.parametrize("Z,M,K,N,res", [(2, 96, 96, 1024, True), (8, 48, 127, 16384, True), (3, 96, 255, 128, True), (2, 48, 48, 4096, False),
                                        (2, 80, 100, 256, True), (8, 96, 255, 16384, True)])
def test_product_that_makes_the_layernorm_statistics_of_its_output(hip, Z, M, K, N, res):
    """round 6 (rcot_gemm_kmajor_stats): C = A B + R from the exact-fp32 kernel AND the per-pixel LayerNorm statistics of C over its
    rows from the same launch == the product followed by rcot_ln_stats; pixel means comparable to the spread (the shifted sums must
    not cancel), M = 80: rows that fill neither a 32-row MFMA tile nor the 96-row workgroup tile."""
    def fn(be, At, Bm, Cc, R, mu, rs):
        be.gemm_kmajor_stats(At, Bm, Cc, M, K, R if res else None, (mu, rs))
    rows, ld = (K + 15) // 16 * 16, (M + 3) // 4 * 4
    A = seeded_tensor(1, (Z, M, K), scale=0.1)
    At = torch.zeros(Z, 1, rows, ld)
    At[:, 0, :K, :M] = A.transpose(1, 2)
    R = seeded_tensor(3, (Z, 1, M, N)) + 2.0 * seeded_tensor(13, (Z, 1, 1, N))
    both(hip, fn, [At, seeded_tensor(2, (Z, 1, K, N)), torch.zeros(Z, 1, M, N), R, torch.zeros(Z, N), torch.zeros(Z, N)], [2, 4, 5])


@pytest.mark.parametrize("Z,M,K,N", [(2, 100, 528, 256), (3, 64, 1021, 64), (1, 384, 2042, 256), (2, 37, 520, 128)])
def test_kgroup_kernel_odd_slab_counts_and_ragged_rows(hip, Z, M, K, N):
    """the eight-wavefront k-group form of the 64 x 64 exact-fp32 kernel (gemm_xx_kg_kernel: K >= 512 on <= 512 workgroups)
    with an ODD number of 16-row slabs (K = 528: 33; 1021: 64 with a ragged last slab; 520: 33 with a half-filled one) — the two
    k-groups then differ by one slab — and row counts that are no multiple of the 64-row tile; the network's own shapes all give even
    counts and full tiles.  Against the fp64 product; the symbol is asserted."""
    import ctypes
    from rcot_amd import lib
    be = hip
    p0 = be.prec
    be.prec = lib.PREC_FP32
    try:
        rows, ld = (K + 15) // 16 * 16, (M + 3) // 4 * 4
        A = seeded_tensor(1, (Z, M, K), scale=0.1)
        At = torch.zeros(Z, 1, rows, ld)
        At[:, 0, :K, :M] = A.transpose(1, 2)
        Bm, R = seeded_tensor(2, (Z, 1, K, N)), seeded_tensor(3, (Z, 1, M, N))
        Cc = torch.full((Z, 1, M, N), float("nan")).cuda()
        be.gemm_kmajor(At.cuda(), Bm.cuda(), Cc, M, K, R=R.cuda())
        buf = ctypes.create_string_buffer(192)
        be.L.rcot_last_kernel(buf, 192)
        assert buf.value.decode() == "gemm_xx_kg_kernel", buf.value
        torch.cuda.synchronize()
        ref = A.double() @ Bm[:, 0].double() + R[:, 0].double()
        assert relerr(Cc[:, 0], ref) < TOL
    finally:
        be.prec = p0


@pytest.mark.parametrize("B,heads,c,N", [(8, 8, 48, 256), (8, 4, 48, 1024)])
def test_attn_core_bwd_takes_dM_as_slabs(hip, B, heads, c, N):
    """the slab form of dM = dY V^T (rcot_bmm_nt_slabs, S <= 8) that rcot_attn_core_bwd is documented to take (include/rcot_hip.h;
    the schedule hands over the dense tensor: the slab route saved a launch and no time) against the dense form: same results up to
    the order of the slab sum."""
    be, C = hip, heads * c
    g = lambda seed, *sh, **kw: seeded_tensor(seed, sh, **kw).cuda()
    dy, V = g(1, B, 1, C, N), g(2, B, 1, C, N)
    Wo, temp = g(3, C, C, scale=0.1), 1 + 0.2 * g(4, heads)
    Gn = torch.tanh(g(5, B, heads, c, c))
    A = torch.softmax(Gn * temp.view(1, heads, 1, 1), -1).contiguous()
    sq = 1 + g(6, B, 2 * C).abs()
    res = []
    for slabs in (True, False):
        outs = [torch.full(sh, float("nan"), device="cuda") for sh in ((B, C, C), (B, C, C), (B, heads), (B, heads, c, c), (B, heads, c, c), (B, C), (B, C))]
        if slabs:
            d = be.bmm_nt_slabs(dy, V)
            assert isinstance(d, tuple) and d[1] >= 1, d
            if d[1] > 8:
                pytest.skip(f"split factor {d[1]} > 8 on this shape")
        else:
            d = torch.empty(B, C, C, device="cuda")
            be.bmm_nt(dy, V, d.unsqueeze(1))
        assert be.attn_core_bwd(d, Wo, A, Gn, sq, temp, *outs)
        torch.cuda.synchronize()
        res.append(outs)
    for a, b in zip(*res):
        assert relerr(a, b) < 1e-5


@pytest.mark.parametrize("B,heads,c", [(2, 1, 48), (2, 2, 48), (8, 8, 48), (2, 1, 96), (1, 4, 96), (3, 2, 96), (2, 4, 24), (8, 4, 48)])
def test_attn_core_bwd(hip, B, heads, c):
    """the attention-matrix backward in ONE launch (fp32 MFMA) == attn_bwd_fused's results"""
    C = heads * c

    def fn(be, dM, Wo, A, Gn, sq, temp, Mf, dWp, dtp, Eq, EqT, Dq, Dk):
        if be is hip and c > 48:                                      # the wrapper routes c = 96 to the two-launch form: call the entry point
            from rcot_amd import lib
            lib.check(be.L.rcot_attn_core_bwd(dM.data_ptr(), 0, C, Wo.data_ptr(), A.data_ptr(), Gn.data_ptr(), sq.data_ptr(), temp.data_ptr(),
                                              Mf.data_ptr(), dWp.data_ptr(), dtp.data_ptr(), Eq.data_ptr(), EqT.data_ptr(), Dq.data_ptr(),
                                              Dk.data_ptr(), B, heads, c, be._st()), "rcot_attn_core_bwd")
        else:
            assert be.attn_core_bwd(dM, Wo, A, Gn, sq, temp, Mf, dWp, dtp, Eq, EqT, Dq, Dk)
    A = torch.softmax(T(1, B, heads, c, c, scale=2.0), -1)
    arrs = [T(5, B, C, C), T(4, C, C, scale=0.1), A, T(7, B, heads, c, c, scale=0.3), T(2, B, 2 * C).abs() * 50 + 1.0, 1 + 0.2 * T(3, heads),
            torch.zeros(B, C, C), torch.zeros(B, C, C), torch.zeros(B, heads), torch.zeros(B, heads, c, c), torch.zeros(B, heads, c, c),
            torch.zeros(B, C), torch.zeros(B, C)]
    both(hip, fn, arrs, [6, 7, 8, 9, 10, 11, 12], tol=5e-5)


@pytest.mark.parametrize("B,heads,c", [(2, 1, 48), (2, 2, 48), (1, 8, 48), (2, 1, 96), (1, 4, 96), (3, 2, 96)])
def test_attn_bwd_fused(hip, B, heads, c):
    """One-launch backward of the attention-matrix chain == the four separate launches (double reference)."""
    C = heads * c

    def fn(be, Graw, sq, temp, Wo, dM, Gn, A, Mf, dWp, dtp, Eq, EqT, Dq, Dk):
        be.attn_softmax(Graw, sq, temp, Gn, A)
        be.attn_bwd_fused(dM, Wo, A, Gn, sq, temp, Mf, dWp, dtp, Eq, EqT, Dq, Dk)
    sq = T(2, B, 2 * C).abs() * 50 + 1.0
    arrs = [T(1, B, heads, c, c, scale=5.0), sq, 1 + 0.2 * T(3, heads), T(4, C, C, scale=0.1), T(5, B, C, C)] + \
        [torch.zeros(B, heads, c, c), torch.zeros(B, heads, c, c), torch.zeros(B, C, C), torch.zeros(B, C, C),
         torch.zeros(B, heads), torch.zeros(B, heads, c, c), torch.zeros(B, heads, c, c), torch.zeros(B, C), torch.zeros(B, C)]
    both(hip, fn, arrs, [7, 8, 9, 10, 11, 12, 13], tol=5e-5)


def test_row_sumsq(hip):
    B, C, N = 2, 96, 1024

    def fn(be, u, out):
        be.row_sumsq(u[:, :2 * C], out)
    both(hip, fn, [T(1, B, 3 * C, N), torch.zeros(B, 2 * C)], [1])


# ----------------------------------------------------------------------------- LayerNorm / stencils
@pytest.mark.parametrize("B,C,H,W", [(2, 48, 16, 16), (1, 96, 32, 64), (2, 384, 8, 8), (1, 192, 24, 40)])
def test_layernorm(hip, B, C, H, W):
    def fn(be, x, mu, rs, g, w, dres, dx, dw, db):
        be.ln_stats(x, mu, rs)
        be.ln_bwd(g, x, mu, rs, w, dres, dx, dw, db)
    arrs = [T(1, B, C, H, W) + 0.5, torch.zeros(B, H * W), torch.zeros(B, H * W), T(2, B, C, H, W), 1 + 0.1 * T(3, C),
            T(4, B, C, H, W), torch.zeros(B, C, H, W), T(5, C), T(6, C)]
    both(hip, fn, arrs, [1, 2, 6, 7, 8])


@pytest.mark.parametrize("B,C,H,W", [(2, 144, 16, 16), (1, 288, 32, 64), (2, 1152, 8, 8), (1, 48, 128, 128), (1, 6, 256, 256), (2, 12, 64, 64),
                                     (1, 5, 64, 256), (2, 3, 20, 16)])
def test_dwconv(hip, B, C, H, W):
    def fn(be, x, w, y, yf, dy, dw):
        be.dwconv3x3(x, w, y)
        be.dwconv3x3(x, w, yf, flip=True)
        be.dwconv3x3_wgrad(dy, x, dw)
    both(hip, fn, [T(1, B, C, H, W), T(2, C, 9), torch.zeros(B, C, H, W), torch.zeros(B, C, H, W), T(3, B, C, H, W), T(4, C, 9)],
         [2, 3, 5])


@pytest.mark.parametrize("B,C,N,heads", [(2, 96, 4096, 1), (8, 48, 16384, 1), (2, 192, 1024, 4), (3, 384, 256, 8)])
def test_block_param_reduce(hip, B, C, N, heads):
    """Deferred LayerNorm partials (two slots) + dW_o / dtau batch sums in one launch == the separate reductions."""
    def fn(be, g1, x1, g2, x2, w, dx1, dx2, gw1, gb1, gw2, gb2, dWp, gWo, dtp, gtemp):
        mu1, rs1 = torch.zeros_like(x1[:, 0]), torch.ones_like(x1[:, 0])
        mu2, rs2 = torch.zeros_like(x2[:, 0]), torch.ones_like(x2[:, 0])
        be.ln_stats(x1, mu1, rs1)
        be.ln_stats(x2, mu2, rs2)
        be.ln_bwd(g1, x1, mu1, rs1, w, None, dx1, None, None, slot=0)
        be.ln_bwd(g2, x2, mu2, rs2, w, None, dx2, None, None, slot=1)
        be.block_param_reduce(C, gw1, gb1, gw2, gb2, dWp, gWo, dtp, gtemp)
    arrs = [T(1, B, C, N), T(2, B, C, N), T(3, B, C, N), T(4, B, C, N), 1 + 0.1 * T(5, C), torch.zeros(B, C, N), torch.zeros(B, C, N),
            T(6, C), T(7, C), T(8, C), T(9, C), T(10, B, C, C), T(11, C, C), T(12, B, heads), T(13, heads)]
    both(hip, fn, arrs, [5, 6, 7, 8, 9, 10, 12, 14], tol=1e-4)


@pytest.mark.parametrize("B,C,hid,N", [(2, 96, 255, 4096), (8, 48, 127, 16384), (2, 192, 510, 1024), (2, 384, 1021, 256),
                                       (2, 24, 40, 1024)])
def test_block_param_reduce_with_weight_gradient_slabs(hip, B, C, hid, N):
    """The three 1x1 weight gradients of a block left as split-K slabs (three thirds of the workspace) and summed by the
    launch that closes the block == accumulating conv1x1_wgrad calls; (24, 40) channels have no slab kernel: fallback."""
    heads = 1

    def fn(be, g1, x1, w, dx1, gw1, gb1, gw2, gb2, dWp, gWo, dtp, gtemp, dY1, X1, gW1, dY2, gW2, dY3, X3, gW3):
        mu, rs = torch.zeros_like(x1[:, 0]), torch.ones_like(x1[:, 0])
        be.ln_stats(x1, mu, rs)
        be.ln_bwd(g1, x1, mu, rs, w, None, dx1, None, None, slot=0)
        be.ln_bwd(g1, x1, mu, rs, w, None, dx1, None, None, slot=1)
        slabs = []
        for part, (dY, X, gW, ln) in enumerate(((dY1, X1, gW1, None), (dY2, x1, gW2, (mu, rs, w, gb1 * 0 + 0.1)),
                                                (dY3, X3, gW3, None))):
            d = be.conv1x1_wgrad_slabs(dY, X, gW, ln=ln, region=(part, 3))
            if d is None:
                be.conv1x1_wgrad(dY, X, gW, ln=ln, beta=1.0)
            slabs.append(d)
        if min(C, hid) >= 33:
            assert all(d is not None for d in slabs)
        be.block_param_reduce(C, gw1, gb1, gw2, gb2, dWp, gWo, dtp, gtemp, slabs)
    arrs = [T(1, B, C, N), T(2, B, C, N), 1 + 0.1 * T(5, C), torch.zeros(B, C, N), T(6, C), T(7, C), T(8, C), T(9, C),
            T(10, B, C, C), T(11, C, C), T(12, B, heads), T(13, heads),
            T(14, B, C, N), T(15, B, hid, N), T(16, C, hid), T(17, B, 2 * hid, N), T(18, 2 * hid, C), T(19, B, hid, N),
            T(20, B, C, N), T(21, hid, C)]
    both(hip, fn, arrs, [4, 5, 6, 7, 9, 11, 14, 16, 19], tol=1e-4)


@pytest.mark.parametrize("B,C,H,W", [(1, 9, 128, 128), (2, 18, 64, 64), (2, 144, 16, 16), (1, 288, 32, 64), (2, 1152, 8, 8),
                                     (2, 21, 16, 24), (3, 5, 4, 4), (1, 6, 256, 256), (1, 4, 64, 256), (2, 3, 20, 16), (1, 3, 24, 128)])
def test_dwconv_bwd_one_pass(hip, B, C, H, W):
    def fn(be, dy, x, w, dx, dw):
        be.dwconv3x3_bwd(dy, x, w, dx, dw)
    both(hip, fn, [T(1, B, C, H, W), T(2, B, C, H, W), T(3, C, 9), torch.zeros(B, C, H, W), T(4, C, 9)], [3, 4])


@pytest.mark.parametrize("B,hid,H,W", [(2, 127, 16, 16), (1, 255, 32, 32), (2, 1021, 8, 8), (1, 5, 128, 128), (1, 3, 256, 256), (2, 9, 64, 64),
                                       (1, 4, 64, 256), (2, 3, 20, 16)])
def test_gdfn_gate(hip, B, hid, H, W):
    def fn(be, p, w, g, dg, dd):
        be.gdfn_gate_fwd(p, w, g)
        be.gdfn_gate_bwd(p, w, dg, dd)
    both(hip, fn, [T(1, B, 2 * hid, H, W), T(2, 2 * hid, 9, scale=0.5), torch.zeros(B, hid, H, W), T(3, B, hid, H, W),
                   torch.zeros(B, 2 * hid, H, W)], [2, 4])


# plane sizes covering every lane-grouping of the fused depthwise weight gradient: 4096/1024 blocks of 4x4 per plane
# (whole workgroup), 64 (one wavefront), 16 and 4 (sub-wave groups), 24 (odd: separate pass), and a ragged tail
@pytest.mark.parametrize("B,hid,H,W", [(1, 5, 128, 128), (2, 9, 64, 64), (2, 31, 32, 32), (2, 127, 16, 16), (3, 37, 8, 8),
                                       (2, 11, 16, 24), (1, 3, 32, 64)])
def test_gdfn_gate_bwd_fused_wgrad(hip, B, hid, H, W):
    def fn(be, p, w, dg, dd, dw):
        be.gdfn_gate_bwd(p, w, dg, dd, dw=dw)
    both(hip, fn, [T(1, B, 2 * hid, H, W), T(2, 2 * hid, 9, scale=0.5), T(3, B, hid, H, W), torch.zeros(B, 2 * hid, H, W),
                   T(4, 2 * hid, 9)], [3, 4])


# fused (dd on chip) for W/4 | 64 with every lane grouping; (16, 24) and (8, 40) take the two-kernel route through scratch
@pytest.mark.parametrize("B,hid,H,W", [(1, 5, 128, 128), (2, 9, 64, 64), (2, 31, 32, 32), (2, 127, 16, 16), (3, 37, 8, 8),
                                       (2, 11, 16, 24), (1, 3, 32, 64), (2, 7, 8, 40), (1, 4, 64, 256), (2, 3, 20, 16)])
def test_gdfn_bwd_one_pass(hip, B, hid, H, W):
    def fn(be, p, w, dg, dp, dw):
        be.gdfn_bwd(p, w, dg, dp, dw)
    both(hip, fn, [T(1, B, 2 * hid, H, W), T(2, 2 * hid, 9, scale=0.5), T(3, B, hid, H, W), torch.zeros(B, 2 * hid, H, W),
                   T(4, 2 * hid, 9)], [3, 4])


# ----------------------------------------------------------------------------- dense convolutions
CONVS = [(2, 3, 48, 16, 16, 3, 1, 1), (2, 48, 24, 16, 16, 3, 1, 1), (1, 192, 384, 8, 8, 3, 1, 1), (2, 96, 3, 16, 24, 3, 1, 1),
         (2, 3, 64, 32, 32, 5, 1, 2), (2, 64, 64, 32, 32, 4, 2, 1), (2, 64, 128, 16, 16, 3, 1, 1), (2, 512, 512, 4, 4, 4, 2, 1),
         (2, 256, 512, 8, 8, 3, 1, 1),
         # thin (RGB output side) direct kernels: every lane grouping of the weight gradient, ragged tile edges, 5x5 taps
         (1, 96, 3, 128, 128, 3, 1, 1), (2, 96, 3, 64, 64, 3, 1, 1), (2, 48, 3, 16, 16, 3, 1, 1), (2, 3, 48, 40, 72, 3, 1, 1),
         (1, 3, 64, 128, 128, 5, 1, 2), (2, 3, 64, 24, 36, 5, 1, 2)]


@pytest.mark.parametrize("B,Ci,Co,H,W,k,s,p", CONVS)
def test_conv2d_fwd_dgrad_wgrad(hip, B, Ci, Co, H, W, k, s, p):
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1

    def fn(be, X, Wt, bias, Y, Y2, dY, dX, dW, R):
        be.conv2d_fwd(X, Wt, bias, Y, s, p, lrelu=0.2)
        be.conv2d_fwd(X, Wt, None, Y2, s, p, lrelu=1.0, R=R)
        be.conv2d_dgrad(dY, Wt, dX, s, p, beta=1.0)
        be.conv2d_wgrad(dY, X, dW, s, p, beta=1.0)
    arrs = [T(1, B, Ci, H, W), T(2, Co, Ci, k, k, scale=0.1), T(3, Co), torch.zeros(B, Co, OH, OW), torch.zeros(B, Co, OH, OW),
            T(4, B, Co, OH, OW), T(5, B, Ci, H, W), T(6, Co, Ci, k, k), T(7, B, Co, OH, OW)]
    both(hip, fn, arrs, [3, 4, 6, 7])


@pytest.mark.parametrize("B,Ci,Co,H,W,k", [(2, 64, 128, 16, 16, 3), (2, 64, 64, 32, 32, 4), (16, 512, 512, 8, 8, 3), (3, 128, 256, 32, 16, 3),
                                          (4, 512, 512, 8, 8, 4), (2, 128, 128, 64, 64, 4), (2, 256, 512, 16, 16, 3), (1, 64, 128, 64, 64, 3)])
def test_conv_pcm(hip, B, Ci, Co, H, W, k):
    """k3s1p1 / k4s2p1 convolutions and their data gradients as bf16x3 K-major products over padded channel-major operands
    (rcot_conv_pcm_*) against torch's conv2d in fp64; bias + LeakyReLU in the epilogue."""
    import torch.nn.functional as F
    s = 1 if k == 3 else 2
    Ho, Wo = H // s, W // s
    assert hip.conv_pcm_ok(Ci, Co, k, s, 1, H, W)
    Wt, X, bias = T(1, Co, Ci, k, k, scale=0.05), T(2, B, Ci, H, W), T(3, Co, scale=0.1)
    dZ = T(4, B, Co, Ho, Wo)
    ref = F.leaky_relu(F.conv2d(X.double(), Wt.double(), bias.double(), s, 1), 0.2)
    ref_dx = torch.nn.grad.conv2d_input(X.shape, Wt.double(), dZ.double(), s, 1)
    Wg = Wt.cuda()
    pf, pd = hip.conv_pcm_pack(Wg, "fwd"), hip.conv_pcm_pack(Wg, "dgrad")
    Y, dX = torch.full((B, Co, Ho, Wo), float("nan"), device="cuda"), torch.full((B, Ci, H, W), float("nan"), device="cuda")
    hip.conv_pcm_fwd(X.cuda(), pf, bias.cuda(), Y, k, lrelu=0.2)
    hip.conv_pcm_dgrad(dZ.cuda(), pd, dX, k)
    torch.cuda.synchronize()
    e1, e2 = relerr(Y, ref), relerr(dX, ref_dx)
    assert e1 < X3_TOL and e2 < X3_TOL, (e1, e2)


@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 96, 192, 64, 64), (8, 384, 768, 16, 16), (3, 192, 96, 32, 16), (2, 96, 48, 64, 64), (1, 48, 64, 20, 24)])
@pytest.mark.parametrize("x3", [False, True])
def test_conv_pcm_wgrad(hip, B, Ci, Co, H, W, x3):
    """3x3 weight gradient as one pixel-reduction product over padded planes with per-tap shifted rows (rcot_conv_pcm_wgrad),
    both arithmetics, accumulating (beta = 1), against torch's conv2d_weight in fp64; then the data gradient from the padded dZ
    it left behind."""
    from rcot_amd import lib
    X, dZ, W0 = T(1, B, Ci, H, W), T(2, B, Co, H, W), T(3, Co, Ci, 3, 3)
    Wt = T(4, Co, Ci, 3, 3, scale=0.05)
    ref = W0.double() + torch.nn.grad.conv2d_weight(X.double(), W0.shape, dZ.double(), stride=1, padding=1)
    ref_dx = torch.nn.grad.conv2d_input(X.shape, Wt.double(), dZ.double(), 1, 1)
    old = hip.prec
    hip.prec = lib.PREC_BF16X3 if x3 else lib.PREC_FP32
    try:
        dW = W0.cuda().clone()
        ok = hip.conv_pcm_wgrad(dZ.cuda(), X.cuda(), dW, 1.0)
        assert ok
        dX = torch.full((B, Ci, H, W), float("nan"), device="cuda")
        hip.conv_pcm_dgrad(dZ.cuda(), hip.conv_pcm_pack(Wt.cuda(), "dgrad"), dX, 3, prepped=True)
        torch.cuda.synchronize()
    finally:
        hip.prec = old
    e, e2 = relerr(dW, ref), relerr(dX, ref_dx)
    assert e < (X3_TOL if x3 else TOL) and e2 < X3_TOL, (e, e2)


@pytest.mark.parametrize("cmap", [1, 2])
def test_conv_pixel_shuffle_epilogue(hip, cmap):
    B, Ci, Co, H, W = 2, 48, (24 if cmap == 1 else 96), 16, 16
    oshape = (B, 4 * Co, H // 2, W // 2) if cmap == 1 else (B, Co // 4, 2 * H, 2 * W)

    def fn(be, X, Wt, Y, back):
        be.conv2d_fwd(X, Wt, None, Y, 1, 1, 1.0, cmap, None)
        be.pixel_shuffle(Y, back, 2 if cmap == 1 else 1)
    both(hip, fn, [T(1, B, Ci, H, W), T(2, Co, Ci, 3, 3, scale=0.1), torch.zeros(*oshape), torch.zeros(B, Co, H, W)], [2, 3])


# ----------------------------------------------------------------------------- Linear
@pytest.mark.parametrize("B,i,o", [(2, 2048, 512), (8, 8192, 2048), (4, 2048, 64), (4, 64, 1), (16, 512, 64)])
def test_linear(hip, B, i, o):
    def fn(be, X, W, b, Y, Y2, dY, dX, dW):
        be.linear_fwd(X, W, b, Y)
        be.linear_fwd(X, W, None, Y2, lrelu=0.2)
        be.linear_dgrad(dY, W, dX)
        be.linear_wgrad(dY, X, dW, beta=1.0)
    arrs = [T(1, B, i), T(2, o, i, scale=0.05), T(3, o), torch.zeros(B, o), torch.zeros(B, o), T(4, B, o), torch.zeros(B, i), T(5, o, i)]
    both(hip, fn, arrs, [3, 4, 6, 7])


# ----------------------------------------------------------------------------- elementwise / critic pieces
def test_elementwise(hip):
    B, C, H, W = 3, 8, 16, 16

    def fn(be, x, y, o1, cat, a, dz, db, al, lo, norms, u0, gp):
        be.axpby(x, y, o1, 1.0, -0.8)
        be.axpby(x, None, cat[:, :C], 1.0, 0.0)
        be.axpby(y, None, cat[:, C:], 1.0, 0.0)
        be.axpby(cat[:, C:], x, x, 2.0, 1.0)
        be.lrelu_bwd(y, a, dz)
        be.bias_grad(dz, db)
        be.lerp(x, y, al, lo)
        be.gp_penalty(y, norms, u0, gp, 1.0 / (2 * B))
    arrs = [T(1, B, C, H, W), T(2, B, C, H, W), torch.zeros(B, C, H, W), torch.zeros(B, 2 * C, H, W), T(3, B, C, H, W),
            torch.zeros(B, C, H, W), T(4, C), T(5, B).abs().clamp(0, 1), torch.zeros(B, C, H, W), torch.zeros(B),
            torch.zeros(B, C, H, W), torch.zeros(1)]
    both(hip, fn, arrs, [0, 2, 3, 5, 6, 8, 9, 10, 11])


# ----------------------------------------------------------------------------- OT cost
@pytest.mark.parametrize("P_,paired", [(32, False), (64, True), (128, True)])
def test_ot_cost(hip, P_, paired):
    B = 4
    de = [0, 2, 3, 7]

    def fn(be, deg, out, tgt, dout, sums, spec, scal, gF):
        d = torch.tensor(de, dtype=torch.int32, device=deg.device)
        be.ot_reduce(deg, out, tgt if paired else None, sums)
        be.ot_spectrum(deg, out, d, gF, spec)
        be.ot_grad(deg, out, tgt if paired else None, d, gF, sums, spec, dout, scal, 1.0, 10000.0, B)
    deg, out = T(1, B, 3, P_, P_, scale=0.3), T(2, B, 3, P_, P_, scale=0.3)
    out[2, 0] = deg[2, 0]               # a plane with an exactly-zero spectrum
    out[3, 1] = deg[3, 1] - 0.25        # constant residual: one non-zero bin
    arrs = [deg, out, T(3, B, 3, P_, P_, scale=0.3), T(4, B, 3, P_, P_, scale=0.01), torch.zeros(2 * B + 2), torch.zeros(B),
            torch.zeros(3), torch.zeros(B, 3, P_, P_)]
    cpu = [a.double().clone() for a in arrs]
    gpu = [a.cuda() for a in arrs]
    fn(DBL, *cpu)
    fn(hip, *gpu)
    torch.cuda.synchronize()
    assert relerr(gpu[4], cpu[4]) < 1e-5 and relerr(gpu[6], cpu[6]) < 1e-5
    assert relerr(gpu[5][2:], cpu[5][2:]) < 1e-5
    # gradient: the |F|=0 / single-bin planes are degenerate for F/|F| in fp32 (tiny bins flip phase); compare the rest
    # tightly and those planes loosely via the loss they induce
    m = torch.ones(B, 3, 1, 1)
    m[2, 0] = 0
    m[3, 1] = 0
    assert relerr(gpu[3].cpu() * m, cpu[3] * m) < 5e-5


def test_ot_cost_golden(hip, gold):
    """The trainer's inline expression evaluated by the REFERENCE (fixture otcost.npz)."""
    fx = gold("otcost.npz")
    res = torch.from_numpy(fx["res"]).cuda()
    B = res.shape[0]
    de = torch.tensor(fx["de_id"], dtype=torch.int32).cuda()
    deg, out = res.clone(), torch.zeros_like(res)
    sums, spec, scal, gF, dout = hip.empty(2 * B + 2), hip.empty(B), hip.empty(3), hip.empty(*res.shape), hip.zeros(*res.shape)
    hip.ot_reduce(deg, out, None, sums)
    hip.ot_spectrum(deg, out, de, gF, spec)
    hip.ot_grad(deg, out, None, de, gF, sums, spec, dout, scal, 1.0, 0.0, B)
    s = scal.cpu().double()
    assert abs(float(s[0]) - float(fx["rmse"])) < 1e-5 * float(fx["rmse"])
    assert abs(float(s[1]) - float(fx["per_sample"].sum())) < 1e-5 * float(fx["per_sample"].sum())
    # d/d(out) = -d/d(res); degenerate planes (zero / constant) excluded as above
    m = torch.ones(B, 3, 1, 1)
    m[1, 0] = 0
    m[3, 1] = 0
    assert relerr(-dout.cpu() * m, torch.from_numpy(fx["dres"]) * m) < 5e-5


# ----------------------------------------------------------------------------- optimizers
def test_optimizers(hip):
    n = 64 * 1000

    def fn(be, p, g, sq, p2, m, v):
        for _ in range(3):
            be.rmsprop_step(p, g, sq, n, 1e-3)
        for t in range(1, 4):
            be.adam_step(p2, g, m, v, n, 1e-3, t)
    g = T(2, n)
    g[:100] = 0.0
    both(hip, fn, [T(1, n), g, torch.zeros(n), T(3, n), torch.zeros(n), torch.zeros(n)], [0, 2, 3, 4, 5], tol=1e-5)


# ----------------------------------------------------------------------------- K-major LDS-DMA GEMM
@pytest.mark.parametrize("B,Ci,Co,N", [(2, 96, 288, 1024), (1, 96, 510, 16384), (2, 255, 96, 256), (2, 48, 144, 2048),
                                       (1, 1021, 384, 256), (2, 384, 2042, 256), (2, 510, 96, 512), (3, 96, 96, 128),
                                       (2, 384, 1152, 64), (8, 192, 510, 1024), (2, 127, 48, 192),
                                       (8, 384, 2042, 256), (8, 1021, 384, 256), (8, 384, 1152, 256), (8, 192, 1020, 1024), (6, 300, 777, 512)])
@pytest.mark.parametrize("ln,res", [(False, False), (True, True)])
def test_kmajor_conv1x1(hip, B, Ci, Co, N, ln, res, split=False, six=False, tol=TOL):
    """packed 1x1 projections on the LDS-DMA ring kernel: forward (+LN prologue, +residual, beta) and data gradient.
    ``split``: also make and hand over the pre-split fragment packs (the bf16x3 producer / consumer kernel takes them);
    ``six``: the three-term packs of the bf16x6 arithmetic instead."""
    def fn(be, W, X, Y, mu, rs, lw, lb, R, dY, dX, WT, WP, WTf, c12, WTs, WPs, WTfs):
        sp3 = (WTs, WPs, WTfs if ln else None) if split else None
        be.pack_weight(W, WT, WP, (lw, lb, WTf, c12) if ln else None, None if six else sp3, sp3 if six else None)
        if ln:
            be.ln_stats(X, mu, rs)
        pk4 = (None, sp3) if six else (sp3, None)
        be.conv1x1_fwd(W, X, Y, ln=(mu, rs, lw, lb) if ln else None, R=R if res else None, beta=1.0 if res else 0.0,
                       packed=(WT, WP, (WTf, c12) if ln else None) + pk4)
        be.conv1x1_dgrad(W, dY, dX, beta=1.0 if res else 0.0, packed=(WT, WP, None) + pk4)
    st, sp = DBL.pack_shapes(Co, Ci)
    sf, sc = DBL.fold_shapes(Co, Ci)
    ss, sq = (DBL.split6_shapes if six else DBL.split_shapes)(Co, Ci)
    # activations with a per-pixel mean comparable to their spread (the LN fold subtracts mu c1 AFTER the product)
    X = T(2, B, Ci, N) + 0.7 * T(12, B, 1, N)
    arrs = [T(1, Co, Ci, scale=0.1), X, T(8, B, Co, N), torch.zeros(B, N), torch.zeros(B, N),
            1 + 0.1 * T(3, Ci), 0.1 * T(4, Ci), T(5, B, Co, N), T(6, B, Co, N), T(7, B, Ci, N), torch.zeros(*st), torch.zeros(*sp),
            torch.zeros(*sf), torch.zeros(*sc), torch.zeros(*ss), torch.zeros(*sq), torch.zeros(*ss)]
    both(hip, fn, arrs, [2, 9, 10, 11] + ([12, 13] if ln else []), tol=tol)


@pytest.mark.parametrize("B,Ci,Co,N,ratio", [(8, 384, 1152, 256, 1.0), (8, 384, 2042, 256, 1.0), (8, 192, 576, 1024, 1.0), (8, 96, 510, 4096, 1.0),
                                              (2, 96, 288, 16384, 1.0), (8, 48, 254, 4096, 20.0), (3, 100, 130, 768, 0.3), (2, 384, 1152, 64, 1.0),
                                              (8, 100, 254, 1024, 1.0)])
def test_fp32_ln_statistics_made_by_the_projection(hip, B, Ci, Co, N, ratio):
    """rcot_gemm_kmajor(ln_compute = 1) in the exact-fp32 arithmetic (round 5): every workgroup of gemm_xx_kernel makes the per-pixel
    LayerNorm statistics of its own columns before its slab loop — no rcot_ln_stats launch.  Same formula and summation order as
    ln_stats_kernel: (mu, rstd) and the projection are BIT-identical to the two-launch form (all three tile shapes: 64x64, 96x128,
    128x128; K % 16 != 0; a mean 20x the spread), and agree with fp64.  Planes above 4096 pixels keep the rcot_ln_stats launch
    (measured faster there)."""
    from rcot_amd import lib
    be = hip
    assert be.prec == lib.PREC_FP32
    W, lw, lb = seeded_tensor(1, (Co, Ci), scale=0.1), 1 + 0.1 * seeded_tensor(3, (Ci,)), 0.1 * seeded_tensor(4, (Ci,))
    X = seeded_tensor(2, (B, Ci, N)) + ratio * (1 + 0.2 * seeded_tensor(12, (B, 1, N)))
    R = seeded_tensor(5, (B, Co, N))
    Xd = X.double()
    mu = Xd.mean(1, keepdim=True)
    rstd = (Xd.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    ref = R.double() + torch.einsum("oc,bcn->bon", W.double(), (Xd - mu) * rstd * lw.double().view(1, Ci, 1) + lb.double().view(1, Ci, 1))
    g = lambda t: t.cuda()
    Wg, Xg, Rg, lwg, lbg = g(W), g(X), g(R), g(lw), g(lb)
    WT, WP = (torch.zeros(*s, device="cuda") for s in be.pack_shapes(Co, Ci))
    WTf, c12 = (torch.zeros(*s, device="cuda") for s in be.fold_shapes(Co, Ci))
    be.pack_weight(Wg, WT, WP, (lwg, lbg, WTf, c12))
    outs = []
    for fused in (True, False):
        mu_, rs_ = torch.full((B, N), float("nan"), device="cuda"), torch.full((B, N), float("nan"), device="cuda")
        Y = torch.full((B, Co, N), float("nan"), device="cuda")
        calls = []
        orig, be.ln_fused = be.ln_stats, fused
        be.ln_stats = lambda *a: (calls.append(1), orig(*a))
        try:
            be.conv1x1_fwd(Wg, Xg, Y, ln=(mu_, rs_, lwg, lbg), R=Rg, packed=(WT, WP, (WTf, c12)), ln_compute=True)
        finally:
            del be.ln_stats
            be.ln_fused = True
        torch.cuda.synchronize()
        kmajor = be.kmajor_worth(Co, N, B)
        assert (len(calls) == 0) == (fused and kmajor and N <= 4096), "which path made the statistics"
        outs.append((mu_, rs_, Y))
    (m1, r1, Y1), (m2, r2, Y2) = outs
    assert torch.equal(m1, m2) and torch.equal(r1, r2) and torch.equal(Y1, Y2)
    e_mu = float((m1.double().cpu() - mu[:, 0]).abs().max() / mu.abs().max())
    e_rs = float((r1.double().cpu() / rstd[:, 0] - 1).abs().max())
    assert e_mu < 2e-6 and e_rs < 2e-5 * (1 + ratio) and relerr(Y1, ref) < TOL * (1 + ratio)


@pytest.mark.parametrize("B,Ci,Co,N,ln,res", [(8, 96, 288, 16384, True, False), (8, 255, 96, 16384, False, True), (3, 100, 330, 16384, True, True)])
def test_streaming_tile_stores_change_no_bit(hip, B, Ci, Co, N, ln, res):
    """Round 5: outputs of the exact-fp32 projection kernel that are larger than the L2s (>= 32 MiB, RCOT_XX_NTS_MB) leave by non-temporal
    stores, so the operand panels the neighbouring row tiles re-read stay in cache (-0.2 ms per iteration, profiles/r05_nts_stores.txt).
    A store hint: the results are the same bits, with and without the LayerNorm prologue, the residual and a row tail."""
    import os
    from rcot_amd import lib
    be = hip
    assert be.prec == lib.PREC_FP32
    g = lambda t: t.cuda()
    Wg, Xg, Rg = g(seeded_tensor(1, (Co, Ci), scale=0.1)), g(seeded_tensor(2, (B, Ci, N))), g(seeded_tensor(5, (B, Co, N)))
    lwg, lbg = g(1 + 0.1 * seeded_tensor(3, (Ci,))), g(0.1 * seeded_tensor(4, (Ci,)))
    WT, WP = (torch.zeros(*s, device="cuda") for s in be.pack_shapes(Co, Ci))
    WTf, c12 = (torch.zeros(*s, device="cuda") for s in be.fold_shapes(Co, Ci))
    be.pack_weight(Wg, WT, WP, (lwg, lbg, WTf, c12))
    mu_, rs_ = torch.empty(B, N, device="cuda"), torch.empty(B, N, device="cuda")
    be.ln_stats(Xg, mu_, rs_)
    outs = []
    old = os.environ.get("RCOT_XX_NTS_MB")
    try:
        for mb in ("32", "0"):
            os.environ["RCOT_XX_NTS_MB"] = mb
            Y = torch.full((B, Co, N), float("nan"), device="cuda")
            be.conv1x1_fwd(Wg, Xg, Y, ln=(mu_, rs_, lwg, lbg) if ln else None, R=Rg if res else None, packed=(WT, WP, (WTf, c12)))
            torch.cuda.synchronize()
            outs.append(Y)
    finally:
        if old is None:
            os.environ.pop("RCOT_XX_NTS_MB", None)
        else:
            os.environ["RCOT_XX_NTS_MB"] = old
    assert torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[0]).all())


@pytest.mark.parametrize("B,heads,c,N", [(2, 1, 96, 16384), (2, 2, 48, 4096), (2, 4, 48, 1024), (1, 8, 48, 256), (2, 4, 96, 256), (8, 8, 48, 256)])
def test_kmajor_mdta_products(hip, B, heads, c, N):
    C = heads * c

    def fn(be, u, MfT, y, x, du, Eq, EqT, Dq):
        uu = u.view(B, 3, heads, c, N)
        Q, K, V = uu[:, 0], uu[:, 1], u.view(B, 3, C, N)[:, 2].unsqueeze(1)
        dd = du.view(B, 3, heads, c, N)
        EqT.copy_(Eq.transpose(-1, -2))
        be.gemm_kmajor(MfT.unsqueeze(1), V, y.view(B, 1, C, N), C, C, R=x.view(B, 1, C, N))
        be.gemm_kmajor(MfT.unsqueeze(1), y.view(B, 1, C, N), du.view(B, 3, C, N)[:, 2].unsqueeze(1), C, C)
        be.gemm_kmajor(EqT, K, dd[:, 0], c, c, R=Q, rowscale=Dq.view(B, heads, c))
        be.gemm_kmajor(Eq, Q, dd[:, 1], c, c, R=K, rowscale=Dq.view(B, heads, c))
    arrs = [T(1, B, 3 * C, N), T(2, B, C, C, scale=0.2), torch.zeros(B, C, N), T(3, B, C, N), torch.zeros(B, 3 * C, N),
            T(4, B, heads, c, c, scale=0.2), torch.zeros(B, heads, c, c), T(5, B, C)]
    both(hip, fn, arrs, [2, 4])


@pytest.mark.parametrize("raw", [True, False])
def test_hand_overs_between_the_two_streams_order_the_kernels(raw):
    """round 6: side_run / side_join on fence-free HIP events (HipBackend._Handover; ``raw`` False = torch's events): a long kernel on the
    calling stream fills a 256 MiB tensor with i, the side stream copies a strided sample of it as soon as it has been handed over, the
    calling stream overwrites the source after the join — 64 rounds, eagerly and from a recorded launch plan; every sample must hold
    its round's value (a consumer that started early would see the previous round's, a producer that did not wait the next round's)."""
    import os
    from rcot_amd.ops import HipBackend
    from rcot_amd.plan import LaunchPlan
    old = os.environ.get("RCOT_RAW_EVENTS")
    os.environ["RCOT_RAW_EVENTS"] = "1" if raw else "0"
    try:
        be = HipBackend()
    finally:
        if old is None:
            os.environ.pop("RCOT_RAW_EVENTS", None)
        else:
            os.environ["RCOT_RAW_EVENTS"] = old
    assert be._raw_events == raw
    R = 64
    src = torch.empty(4096, 16384, device="cuda")
    out = torch.zeros(R, 4096, 4, device="cuda")

    def rounds():
        for i in range(R):
            be.fill(src, float(i + 1))
            be.side_run(lambda i=i: be.axpby(src[:, :4], None, out[i], 1.0, 0.0), src)
            be.side_join()
            be.fill(src, -1.0)
    rounds()
    torch.cuda.synchronize()
    want = torch.arange(1, R + 1, device="cuda", dtype=torch.float32).view(R, 1, 1).expand(R, 4096, 4)
    assert torch.equal(out, want)
    out.zero_()
    plan = LaunchPlan(be).record(rounds)
    out.zero_()
    for _ in range(3):
        plan.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
