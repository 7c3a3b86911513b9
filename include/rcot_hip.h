/* rcot_hip.h — C ABI of librcot_hip.so: the MI355X (gfx950) kernels of the RCOT training hot path.
 *
 * The reference (xl-tang3/RCOT) has no FFI; its hot path sits behind Python nn.Module calls that end in
 * ATen/cuDNN/cuBLAS/cuFFT kernels.  Each entry point below replaces one such group of reference operations;
 * the reference site is cited as file:line (paths relative to the reference root).
 *
 * Conventions
 *  - All tensors are fp32, NCHW, contiguous unless a stride argument says otherwise; "N" = H*W pixels.
 *  - Pointers are DEVICE pointers owned by the caller (PyTorch's allocator); the library never allocates.
 *  - Every function is asynchronous on `stream` (a hipStream_t), reentrant and thread-safe; no hidden syncs.
 *  - Return value: 0 ok; RCOT_EINVAL (-1) bad shape/alignment/null; RCOT_EWORKSPACE (-2) workspace too small;
 *    RCOT_EUNSUPPORTED (-3) this entry point has no kernel for the shape (use the general one it names);
 *    >0 a hipError_t from the launch.  Nothing throws across the ABI.
 *  - "ws/ws_bytes": caller-provided scratch for split-K slabs / FFT lines (256 MiB is plenty for every call).
 *  - Gradients of weights ACCUMULATE when beta = 1 (dW = beta*dW + contribution); weights shared by the two
 *    passes of T_net rely on this.
 */
#ifndef RCOT_HIP_H
#define RCOT_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped on EVERY signature change; returned by rcot_abi_version() (csrc/api.hip) and compared by the loader
 * (rcot_amd/lib.py ABI_VERSION, tests/test_abi.py) so that a stale prebuilt .so is refused, not mis-called. */
#define RCOT_ABI_VERSION 25

/* Arithmetic of the MFMA products of the three GEMM-shaped entry points that take `prec` (rcot_gemm_kmajor,
 * rcot_conv1x1_wgrad, rcot_bmm_nt); operands and results are fp32 in memory either way.
 *   RCOT_PREC_FP32   : v_mfma_f32_32x32x2_f32, bit-exact fp32 fmaf chains (the reference's dtype).
 *   RCOT_PREC_BF16X3 : each fp32 operand is split on chip into two bfloat16 terms and every product is evaluated as
 *                      hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (~1e-5 relative per
 *                      product; BASELINE config 5 "reduced-precision MFMA pointwise projections").  Shapes the split
 *                      kernels do not cover silently use the fp32 kernels (never the reverse).
 *   RCOT_PREC_BF16X6 : fp32-CLASS results from the bf16 pipe (rcot_gemm_kmajor with a three-term Asplit pack — without one it
 *                      runs the exact-fp32 kernels — and the pixel reductions rcot_conv1x1_wgrad* / rcot_bmm_nt*).  Each fp32 operand is split into THREE bfloat16 terms (8 + 8 + 8
 *                      significand bits = fp32's 24) and a product is the six partial products of order <= 2
 *                      (t0 t0' + t0 t1' + t1 t0' + t0 t2' + t2 t0' + t1 t1') with fp32 accumulation.  What is dropped is <= 2^-24
 *                      relative: measured 6e-9 of max|C| against 4e-7 for the rounding of the fp32 ACCUMULATION that every fp32
 *                      GEMM (the reference's included) carries — the results are as accurate as v_mfma_f32_32x32x2_f32's
 *                      (tests/test_x3_gpu.py::test_x6_*: every fp32 tolerance of the suite holds), at 6 x 32 MFMA cycles per
 *                      32x32x16 block instead of 8 x 64. */
#define RCOT_PREC_FP32 0
#define RCOT_PREC_BF16X3 1
#define RCOT_PREC_BF16X6 2
/*   RCOT_PREC_BF16X1 : ONE bf16 MFMA product per fp32 product, fp32 accumulation (BASELINE configs[4] "16-bit MFMA pointwise
 *                      projections" taken literally): the kernels, packs and contracts of RCOT_PREC_BF16X3 with the two cross products
 *                      skipped, i.e. C = rne_bf16(A) rne_bf16(B).  ~2^-9 relative per operand: the transport map's output at 128x128
 *                      differs from the reference's by more than the north_star's 1e-3 (measured figures: DESIGN.md section 5), so
 *                      this arithmetic is OPT-IN and is never the default of anything.  Products without a split kernel (the paired
 *                      data + weight gradient launch, shapes below the split kernels' limits) run as under RCOT_PREC_BF16X3 / fp32. */
#define RCOT_PREC_BF16X1 3

int rcot_abi_version(void);
/* Debugging / test hook (round 6, ABI 23): the pixel-reduction kernel's cooperative operand split (csrc/gemm_nt_body.h COOP) per
 * arithmetic — bit 0 bf16x6, bit 1 bf16x3; -1 = as the environment's RCOT_NT_COOP says (read once per process; default 3).  Both
 * forms give the same bits (tests/test_x3_gpu.py); process-wide host state, no launch, no stream argument. */
int rcot_debug_nt_coop(int mask);
/* Measurement aid (bench.py): per-launch DEVICE time stamps.  Between rcot_profile_begin() and rcot_profile_end() every kernel the
 * calling thread launches through this library goes out with a start and a stop event of its own (hipExtLaunchKernelGGL): the dispatch's
 * begin / end times, i.e. the durations `rocprofv3 --kernel-trace` lists, in situ, with nothing inserted between the kernels.  After a
 * device synchronisation rcot_profile_end writes one line per kernel symbol — "demangled symbol|launches|total ms", largest first — into
 * out[0..n) and returns the number of launches collected.  Not for use while a HIP graph is being captured. */
int rcot_profile_begin(void);
int rcot_profile_end(char* out, int n);
/* Measurement aid (bench.py): the symbol of the kernel the calling thread's last dispatcher launched — several kernel families serve one
 * entry point (x3p / gemm_x3 / gemm_xx behind rcot_gemm_kmajor, ...) — copied into out[0..n); returns a counter that advances with every
 * recorded launch decision (unchanged counter: the last entry point did not pass a tagged dispatcher).  Host only, no stream. */
int rcot_last_kernel(char* out, int n);

/* ---- 1x1 projections (true dense GEMMs, fp32 MFMA) -------------------------------------------------------
 * Y[b] (Co x N) = W (Co x Ci, leading dim ldw) * LN?(X[b]) (Ci x N) [+ R[b]] [+ beta*Y[b]]
 * replaces nn.Conv2d(k=1) at Net_Restormer.py:25,27 (qkv, project_out), :73,:78 (GDFN project_in/out),
 * :282,:291,:299,:304,:316 (reduce_*); with ln_* non-null the WithBias LayerNorm of :186-189 (stats from
 * rcot_ln_stats) is applied to X while it is staged, i.e. norm1/norm2 of :211-212 never hit HBM.
 * R (optional, same layout as Y, batch stride sRb) is the residual add of :211-212.
 * sXb / sYb / sRb: batch strides in floats (lets callers pass channel slices, e.g. the two halves of a cat). */
int rcot_conv1x1_fwd(const float* W, long ldw, const float* X, long sXb, float* Y, long sYb, int B, int Ci, int Co,
                     int N, const float* ln_mu, const float* ln_rs, const float* ln_w, const float* ln_b,
                     const float* R, long sRb, float beta, void* stream);
/* dX[b] (Ci x N) = W^T * dY[b] (+ beta*dX[b]) — autograd's conv data-gradient for the same layers. */
int rcot_conv1x1_dgrad(const float* W, long ldw, const float* dY, long sdYb, float* dX, long sdXb, int B, int Ci,
                       int Co, int N, float beta, void* stream);
/* dW (Co x Ci, ld ldw) = beta*dW + sum_b dY[b] * LN?(X[b])^T — conv weight-gradient; the reduction over
 * batch*pixels is split across workgroups into slabs in ws and summed deterministically.  N % 16 == 0. */
int rcot_conv1x1_wgrad(const float* dY, long sdYb, const float* X, long sXb, float* dW, long ldw, int B, int Ci,
                       int Co, int N, const float* ln_mu, const float* ln_rs, const float* ln_w, const float* ln_b,
                       float beta, float* ws, size_t ws_bytes, int prec, void* stream);
/* The same product WITHOUT the final sum: the split-K slabs [S][Co][ldws] are left in ws (S and ldws returned) for
 * rcot_block_param_reduce, which adds them to dW together with the other parameter reductions that close a transformer
 * block — three reduce launches per block become none.  RCOT_EUNSUPPORTED when the LDS-DMA kernel does not take the shape
 * (fewer than 33 channels on either side, unaligned views): call rcot_conv1x1_wgrad instead. */
int rcot_conv1x1_wgrad_slabs(const float* dY, long sdYb, const float* X, long sXb, int B, int Ci, int Co, int N,
                             const float* ln_mu, const float* ln_rs, const float* ln_w, const float* ln_b, float* ws,
                             size_t ws_bytes, int prec, int* S, int* ldws, void* stream);
/* BOTH products of one incoming gradient dY of a 1x1 projection from ONE launch (bf16x3 arithmetic only):
 *   dX[b] (Ci x N) = W^T dY[b]         — rcot_conv1x1_dgrad on the packs WP / WPs of rcot_pack_weight (the K-major operand of the
 *                                        data gradient and its pre-split form), and
 *   the split-K slabs of dW            — exactly what rcot_conv1x1_wgrad_slabs leaves in ws_slabs (S, ldws returned).
 * autograd's conv2d backward computes the two from the same grad_output (Net_Restormer.py:25,27,73,78 in backward); they are
 * independent, and on the small levels of T_net each alone is a launch of one or two tiles per CU.  Workgroups of both
 * products share one grid here, so the pair lasts as long as the longer product instead of their sum, and dY is read from HBM
 * once.  ws: split-K scratch of the data gradient (as rcot_gemm_kmajor).  RCOT_EUNSUPPORTED: prec is not RCOT_PREC_BF16X3 or one
 * of the products has no kernel of its family for the shape — run rcot_conv1x1_dgrad / rcot_conv1x1_wgrad_slabs instead. */
int rcot_conv1x1_dgrad_wgrad_slabs(const float* WP, long ldp, const void* WPs, const float* dY, long sdYb, float* dX, long sdXb,
                                   const float* X, long sXb, int B, int Ci, int Co, int N, const float* ln_mu,
                                   const float* ln_rs, const float* ln_w, const float* ln_b, float* ws, size_t ws_bytes,
                                   float* ws_slabs, size_t ws_slabs_bytes, int prec, int* S, int* ldws, void* stream);

/* ---- batched small-matrix x activation products of MDTA ---------------------------------------------------
 * z = zo*Zi + zi (image, head).  C[z] (M x N) = op(A[z]) (M x K) * Bm[z] (K x N) + rowscale[z][m]*R[z] + beta*C[z]
 * replaces `attn @ v` + project_out (Net_Restormer.py:45,49: A = W_o*blockdiag(softmax), Bm = v) and, in
 * backward, dV = M^T dY, dQ = Eq K + Dq.Q, dK = Eq^T Q + Dk.K. */
int rcot_bmm_nn(const float* A, long lda, long sAo, long sAi, int transA, const float* Bm, long ldb, long sBo,
                long sBi, float* C, long ldc, long sCo, long sCi, const float* R, long ldr, long sRo, long sRi,
                const float* rowscale, long sSo, long sSi, int Zo, int Zi, int M, int N, int K, float beta,
                void* stream);
/* C[z] (M x N) = A[z] (M x K) * Bm[z]^T (N x K), K = pixels (split-K through ws).
 * replaces `q @ k.transpose(-2,-1)` (Net_Restormer.py:42) on un-normalised q,k, and dM = dY V^T in backward. */
int rcot_bmm_nt(const float* A, long lda, long sAo, long sAi, const float* Bm, long ldb, long sBo, long sBi,
                float* C, long ldc, long sCo, long sCi, int Zo, int Zi, int M, int N, int K, float* ws,
                size_t ws_bytes, int prec, void* stream);

/* The same product left as split-K slabs [Zo*Zi][S][M][ldws] in ws (S, ldws returned) for a consumer that sums them
 * (rcot_attn_softmax); RCOT_EUNSUPPORTED when the LDS-DMA kernel does not take the shape: call rcot_bmm_nt. */
int rcot_bmm_nt_slabs(const float* A, long lda, long sAo, long sAi, const float* Bm, long ldb, long sBo, long sBi, int Zo, int Zi,
                      int M, int N, int K, float* ws, size_t ws_bytes, int prec, int* S, int* ldws, void* stream);

/* ---- K-major fast path of the same products (LDS-DMA ring, see csrc/gemm_glds.hip) --------------------------
 * C[z] (M x N) = A[z] * LN?(Bm[z]) + rowscale[z][m]*R[z] + beta*C[z] with A given TRANSPOSED: At[k][m], leading dim
 * lda, `a_rows` readable rows of which rows >= K are zero (a_rows >= ceil16(K)).  N % 128 == 0.  Serves
 * rcot_conv1x1_fwd / _dgrad (At = packs from rcot_pack_weight) and the MDTA apply / dV / dQ / dK products. */
int rcot_gemm_kmajor(const float* At, long lda, long sAo, long sAi, int a_rows, const float* Bm, long ldb, long sBo,
                     long sBi, float* C, long ldc, long sCo, long sCi, const float* R, long ldr, long sRo, long sRi,
                     const float* rowscale, long sSo, long sSi, float* ln_mu, float* ln_rs, long sLN, int ln_compute,
                     const float* ln_w, const float* ln_b, const float* AtF, const float* ln_c12, const void* Asplit, int Zo,
                     int Zi, int M, int N, int K, float beta, float* ws, size_t ws_bytes, int prec, void* stream);
/* AtF / ln_c12 (optional, used with ln_* and prec = RCOT_PREC_BF16X3): the LN-FOLDED operand (W diag(ln_w))^T, same
 * leading dim and strides as At, and [c1 = W ln_w | c2 = W ln_b] (2 x ceil4(M) floats), both made by rcot_pack_weight.
 * The split kernel then evaluates  rs[n] (AtF^T X)[m][n] - rs[n] mu[n] c1[m] + c2[m]  ( == W LN(X) ): the per-pixel
 * statistics enter in the epilogue and the slab loop carries no normalisation arithmetic.  Without them a LayerNorm
 * prologue runs on the exact-fp32 kernel whatever `prec` says.
 * Asplit (optional, prec = RCOT_PREC_BF16X3, batch-invariant A): the PRE-SPLIT fragment pack of the operand actually
 * multiplied (At, or AtF when ln_* is given) from rcot_pack_weight (WTs / WPs / WTfs).  With it and N % 256 == 0 the
 * product runs on the producer / consumer kernel of csrc/gemm_x3w.hip (no per-row scale on that path).
 * ws / ws_bytes (optional): scratch for the split-K pieces of the bf16x3 kernel (few output tiles, long reductions: the
 * 16x16 / 32x32 levels); without it such products run unsplit.
 * ln_compute = 1 (with ln_*, AtF, ln_c12, Asplit, prec = RCOT_PREC_BF16X3, Zi = 1, K % 16 == 0): ln_mu / ln_rs are OUTPUTS.  The
 * producer wavefronts of csrc/gemm_x3w.hip see every fp32 row of X when they split it: they form the per-pixel statistics
 * (the shifted sums of rcot_ln_stats) on the way, the fold epilogue takes them from LDS and row tile 0 writes them out for
 * rcot_ln_bwd: norm1/norm2 of Net_Restormer.py:211-212 cost no pass over x and no launch.  RCOT_EUNSUPPORTED when that kernel
 * does not run the shape unsplit (nothing is launched: call rcot_ln_stats, then this entry point with ln_compute = 0). */
/* Up to three INDEPENDENT plain products of rcot_gemm_kmajor (no LayerNorm prologue, no split pack, beta = 0) sharing the pixel count N,
 * from ONE launch: the data gradients of one MDTA block — dV = Mf^T dY, dQ = Eq K + Dq.Q, dK = Eq^T Q + Dk.K (autograd's
 * backward of Net_Restormer.py:42-45) — are independent of each other and, below the 128x128 level, a launch of 8-50 workgroups each.
 * `d`: HOST array of n descriptors (copied at the call), fields as the arguments of rcot_gemm_kmajor.  prec: RCOT_PREC_FP32 or
 * RCOT_PREC_BF16X6 (both run these products on the exact-fp32 kernel); RCOT_EUNSUPPORTED for RCOT_PREC_BF16X3 (nothing launched:
 * call rcot_gemm_kmajor per product).  ONE tile shape serves the grid — the one rcot_gemm_kmajor would choose for product 0 — and the
 * eight-wavefront k-group kernel is never used: a product with K < 512 (RCOT_XX_KG_MINK; every MDTA product: K = c or C <= 384) is
 * bit-identical to its own rcot_gemm_kmajor launch (the per-element summation order of gemm_xx_kernel does not depend on the tile);
 * with K >= 512 on <= 512 workgroups the single launch runs gemm_xx_kg_kernel, which adds two partial chains: equal to fp32
 * rounding, not to the bit. */
typedef struct rcot_kmajor_desc {
    const float* At; long lda, sAo, sAi; int a_rows;
    const float* Bm; long ldb, sBo, sBi;
    float* C; long ldc, sCo, sCi;
    const float* R; long ldr, sRo, sRi;
    const float* rowscale; long sSo, sSi;
    int Zo, Zi, M, K;
} rcot_kmajor_desc;
int rcot_gemm_kmajor_multi(const rcot_kmajor_desc* d, int n, int N, int prec, void* stream);
/* A plain exact-fp32 product of rcot_gemm_kmajor (no LayerNorm prologue, no rowscale, beta = 0, Zi = 1) whose OUTPUT is the input of a
 * WithBias LayerNorm (Net_Restormer.py:211-212: y = x + attn(norm1(x)) feeds norm2, the block's result feeds the next block's norm1):
 *     C[z] = A[z] Bm[z] + R[z]   and   st_mu[z][n], st_rs[z][n] = mean and 1/sqrt(biased variance + 1e-5) over the M rows of C[z][:, n]
 * — rcot_ln_stats of the stored tensor, made by the epilogue of the product that stores it (round 6): no pass over the tensor, no launch.
 * RCOT_EUNSUPPORTED unless M <= 96 and N % 128 == 0 (one row tile must hold every channel of its pixels; the 48- and 96-channel
 * levels): the caller runs rcot_gemm_kmajor and lets the consumer make its statistics.  Same values as rcot_ln_stats to fp32 rounding
 * (same shifted-sum formula, another summation order). */
int rcot_gemm_kmajor_stats(const float* At, long lda, long sAo, long sAi, int a_rows, const float* Bm, long ldb, long sBo, long sBi,
                           float* C, long ldc, long sCo, long sCi, const float* R, long ldr, long sRo, long sRi, float* st_mu,
                           float* st_rs, long sST, int Zo, int Zi, int M, int N, int K, void* stream);
/* sizeof(rcot_kmajor_desc) as the library was compiled: a binding checks its own struct layout against it (tests/test_abi.py). */
int rcot_kmajor_desc_size(void);
/* Private repack of a 1x1 weight W [Co][Ci] (native OIHW layout, leading dim ldw), refreshed after every optimizer
 * step: WT [ceil16(Ci)][ceil4(Co)] = W^T zero padded (forward), WP [ceil16(Co)][ceil4(Ci)] = W zero padded (dgrad),
 * and — when the projection follows a LayerNorm (ln_w, ln_b, WTf, c12 non-null; Net_Restormer.py:211-212 norm1 -> qkv,
 * norm2 -> project_in) — WTf [ceil16(Ci)][ceil4(Co)] = (W diag(ln_w))^T and c12 = [W ln_w | W ln_b] (2 x ceil4(Co)).
 * WTs / WPs / WTfs (each optional): the same three operands PRE-SPLIT for the bf16x3 kernels, as MFMA fragments:
 * [ceil(K/16)][ceil(M/32)][hi | lo][64 lanes][8 bf16] bytes (M x K = Co x Ci for WTs / WTfs, Ci x Co for WPs), lane
 * (lm, kg) of row tile mt holding A[32 mt + lm][16 slab + 8 kg + (0..7)], hi = rne_bf16(a), lo = rne_bf16(a - hi).
 * WTs6 / WPs6 / WTfs6 (each optional): the THREE-term form of the same packs for RCOT_PREC_BF16X6:
 * [ceil(K/16)][ceil(M/32)][t0 | t1 | t2][64 lanes][8 bf16], t2 = rne_bf16(a - t0 - t1). */
int rcot_pack_weight(const float* W, long ldw, int Co, int Ci, float* WT, float* WP, const float* ln_w, const float* ln_b,
                     float* WTf, float* c12, void* WTs, void* WPs, void* WTfs, void* WTs6, void* WPs6, void* WTfs6, void* stream);
/* The same repack for MANY weights in one launch (after an optimizer step): `table` is a DEVICE array of n rows of
 * 20 int64 { W, ldw, Co, Ci, WT, WP, first chunk, ln_w, ln_b, WTf, c12, WTs, WPs, WTfs, WTs6, WPs6, WTfs6, 0, 0, 0 } (0 for
 * what is absent); the pack space nt + np (+ nt with a fold) elements, nt = ceil16(Ci)*ceil4(Co), np = ceil16(Co)*ceil4(Ci), plus
 * one unit per pre-split record — rt = ceil(Ci/16)*ceil(Co/32)*64 for WTs (and again for WTfs, WTs6, WTfs6), rp = ceil(Co/16)*
 * ceil(Ci/32)*64 for WPs (and WPs6) — of every weight is cut into 1024-unit chunks, followed for a fold by ceil(ceil4(Co)/64)
 * row-sum chunks (c1, c2); the DEVICE int32 array chunk2desc[nchunks] names each chunk's row. */
int rcot_pack_weights(const long long* table, const int* chunk2desc, int nchunks, void* stream);

/* ---- critic Linear layers (Net_Restormer.py:494-496, 513-520) ---------------------------------------------
 * Y[B,out] = act(X[B,in] W^T + bias), act = LeakyReLU(slope lrelu) or identity (lrelu = 1). */
int rcot_linear_fwd(const float* X, const float* W, const float* bias, float* Y, int B, int in, int out, float lrelu,
                    float* ws, size_t ws_bytes, void* stream);
int rcot_linear_dgrad(const float* dY, const float* W, float* dX, int B, int in, int out, float* ws, size_t ws_bytes,
                      void* stream);
int rcot_linear_wgrad(const float* dY, const float* X, float* dW, int B, int in, int out, float beta, void* stream);

/* ---- dense convolutions (implicit GEMM, fp32 MFMA) --------------------------------------------------------
 * Y = act(conv2d(X, Wt, stride, pad) + bias) [+ R]; act = LeakyReLU(lrelu) (1 = identity).
 * cmap 1/2 folds nn.PixelUnshuffle(2)/nn.PixelShuffle(2) into the store (Net_Restormer.py:90-91, 107-108).
 * replaces nn.Conv2d at Net_Restormer.py:117 (patch embed), :90,:107 (Down/Upsample), :326 (+inp_img, :375),
 * and F_net.features :443-487 (k5s1p2, k4s2p1, k3s1p1 + LeakyReLU 0.2). */
int rcot_conv2d_fwd(const float* X, const float* Wt, const float* bias, float* Y, int B, int Ci, int H, int W,
                    int Co, int KH, int KW, int stride, int pad, float lrelu, int cmap, const float* R, const float* mask,
                    float mslope, float* ws, size_t ws_bytes, void* stream);
/* stride 2 (k4 only) is decomposed by output parity into 4 dense sub-problems (no multiplications by zero). */
int rcot_conv2d_dgrad(const float* dY, const float* Wt, float* dX, int B, int Ci, int H, int W, int Co, int KH,
                      int KW, int stride, int pad, float beta, const float* mask, float mslope, float* ws, size_t ws_bytes,
                      void* stream);
/* mask / mslope (optional, ABI 14): the result is stored as  mask > 0 ? out : out * mslope  with `mask` laid out like the output —
 * rcot_lrelu_bwd of the critic (Net_Restormer.py:445-487: the LeakyReLU(0.2) behind every conv) folded into the store of the
 * data gradient that feeds it, and into the linearised forward sweep of the gradient penalty.  Same expression as the separate
 * launch, so results are bit-identical to it. */
int rcot_conv2d_wgrad(const float* dY, const float* X, float* dWt, int B, int Ci, int H, int W, int Co, int KH,
                      int KW, int stride, int pad, float beta, float* ws, size_t ws_bytes, void* stream);
/* ---- k3 s1 p1 / k4 s2 p1 convolutions and their data gradients as bf16x3 K-major products (csrc/conv_pcm.hip) ------------------
 * The critic's convolutions (Net_Restormer.py:447-487, Ci % 16 == 0) over a PADDED, CHANNEL-MAJOR copy of the input: plane geometry
 * (Ho + 2) x (Wo + 8) with the image at rows 1.., columns 4.. (Ho, Wo = the conv's OUTPUT size), tensor [channels][B planes] flat
 * with channel stride ldo.  A GEMM column n is a position of that flat space; each reduction slab is 16 channel rows at a per-tap
 * offset (contiguous, any 4-byte alignment), so the product runs on the producer / consumer kernel with no gather arithmetic.
 *  rcot_conv_pcm_prep : mode 0  out[c][b][y+1][x+4] = X[b][c][y][x]  (k3 s1 operand; also dZ of either data gradient);
 *                       mode 1  the four parity planes Q_ij[c](y, x) = X[b][c][2y+i-1][2x+j-1] as channels ij*C + c in the
 *                       geometry of the H/2 x W/2 output (k4 s2 forward operand).  Every position is written (zeros outside).
 *  rcot_conv_pcm_pack : A[m][k] = W[rowoff[m] + koff[k]] (DEVICE int tables) as the pre-split fragment pack of rcot_pack_weight
 *                       (K % 16 == 0): forward / data-gradient operand orders of a conv weight without a transposed copy.
 *  rcot_conv_pcm      : Y[m][colmap[n/4] ..+3] = lrelu(sum_{tap, c} A[m][(tap, c)] Xp[c][n + tapoff[tap]] + bias[m]) for the 4-column
 *                       groups with colmap >= 0 (DEVICE int table, dense NCHW offsets; -1 = padding position); tapoff: HOST array of
 *                       ntaps <= 16 element offsets; bias (optional) padded to a multiple of 4 rows; N % 128 == 0; Xp must be
 *                       readable from n + min(tapoff) to n + max(tapoff) (guard bands).  Split-K through ws when the tiles are few.
 *  rcot_conv_pcm_merge: dX[b][c][iy][ix] from the parity planes dQ[(ij, c)] (channel stride ldq) of a k4 s2 data gradient. */
int rcot_conv_pcm_prep(const float* X, float* out, long ldo, int B, int C, int H, int W, int mode, void* stream);
/* 3x3 (s1 p1) WEIGHT gradient over two padded planes of rcot_conv_pcm_prep mode 0 (same geometry, row stride ld, plane pitch Wp):
 * dW[co][ci][ky][kx] = beta dW + sum_n dZp[co][n] Xp[ci][n + (ky-1) Wp + (kx-1)]  over the N (% 16 == 0) plane positions — the
 * pixel-reduction GEMM of rcot_conv1x1_wgrad with the B rows shifted per tap (csrc/gemm_nt_glds.hip), split-K through ws, `prec` as
 * there.  Used for the transport map's Down/Upsample convolutions (Net_Restormer.py:86-111) under bf16x3. */
int rcot_conv_pcm_wgrad(const float* dZp, const float* Xp, long ld, int N, int Wp, int Co, int Ci, float* dW, float beta, float* ws,
                        size_t ws_bytes, int prec, void* stream);
int rcot_conv_pcm_merge(const float* dQ, long ldq, float* dX, int B, int C, int H, int W, void* stream);
int rcot_conv_pcm_pack(const float* W, const int* rowoff, const int* koff, int M, int K, void* Apk, void* stream);
int rcot_conv_pcm(const void* Apk, int M, int K, const float* Xp, long ldb, int N, const int* tapoff, int ntaps, const float* bias,
                  float lrelu, const int* colmap, float* Y, long ldy, float* ws, size_t ws_bytes, void* stream);
/* mode 1: PixelUnshuffle(2) [planes][H][W] -> [4*planes][H/2][W/2]; mode 2: PixelShuffle(2) (inverse). */
int rcot_pixel_shuffle(const float* in, float* out, long planes, int H, int W, int mode, void* stream);

/* ---- per-pixel LayerNorm over channels (Net_Restormer.py:186-189, 198-200) ------------------------------- */
int rcot_ln_stats(const float* x, float* mu, float* rs, int B, int C, int N, void* stream);
/* dx = dres + LN'(g); dw += sum g*xhat; db += sum g  (SURVEY.md A.1); C <= 512.  ws: >= 8 KiB * C scratch for the
   per-workgroup partial sums of dw/db, which are added up in a fixed order (deterministic, no atomics).  dw == db == NULL
   defers that sum: the rcot_ln_bwd_rows(B, C, N) partial rows [rows][2C] stay in ws for rcot_block_param_reduce. */
int rcot_ln_bwd(const float* g, const float* x, const float* mu, const float* rs, const float* w, const float* dres,
                float* dx, float* dw, float* db, int B, int C, int N, void* ws, long ws_bytes, void* stream);
int rcot_ln_bwd_rows(int B, int C, int N);
/* Closes the backward of one transformer block in one launch: gw1/gb1 += columns of part1, gw2/gb2 += columns of part2
 * (deferred rcot_ln_bwd partials, same rows and C), gWo += sum_b dWo_part[b], gtemp += sum_b dtemp_part[b], and for
 * each of the n_sets (<= 4) HOST rows { ws, S, M, N, ldws, dst, ldd } of slab_sets (from rcot_conv1x1_wgrad_slabs):
 * dst (M x N, leading dim ldd) += sum_s ws[s][M][ldws], fixed summation order. */
int rcot_block_param_reduce(const float* part1, const float* part2, int rows, int C, float* gw1, float* gb1, float* gw2,
                            float* gb2, const float* dWo_part, float* gWo, const float* dtemp_part, float* gtemp, int B,
                            int heads, const long long* slab_sets, int n_sets, void* stream);

/* ---- depthwise 3x3 stencils (Net_Restormer.py:26, 75-76, 82-83) ------------------------------------------ */
/* y = dwconv3x3(x, w[C][3][3], pad 1); flip=1 correlates with the rotated filter (= data gradient). */
int rcot_dwconv3x3(const float* x, const float* w, float* y, int B, int C, int H, int W, int flip, void* stream);
/* g[b][j] = gelu_erf(dw(p)[b][j]) * dw(p)[b][j+hid]   (p has 2*hid channels)  — Net_Restormer.py:82-83 */
int rcot_gdfn_gate_fwd(const float* p, const float* w, float* g, int B, int hid, int H, int W, void* stream);
/* dd = d(loss)/d(dw(p)) from dg, recomputing dw(p).  dwg (optional, [2*hid][3][3]): also accumulates the
   depthwise weight gradient dwg += sum dd (*) p in the same pass (Net_Restormer.py:75-76 backward). */
int rcot_gdfn_gate_bwd(const float* p, const float* w, const float* dg, float* dd, float* dwg, int B, int hid, int H,
                       int W, void* stream);
/* The whole depthwise part of the GDFN backward in one pass: dp = dw3x3(dd, rotated w) with dd = gate backward of dg
 * (dd is never written) and dwg += sum dd (*) p.  Fused when W/4 divides 64 and the plane/strip grouping is regular;
 * otherwise it runs rcot_gdfn_gate_bwd + rcot_dwconv3x3(flip) through dd_scratch [B][2*hid][H][W] (RCOT_EWORKSPACE if
 * that is needed and null). */
int rcot_gdfn_bwd(const float* p, const float* w, const float* dg, float* dp, float* dwg, float* dd_scratch, int B, int hid,
                  int H, int W, void* stream);
/* dw[c][i][j] += sum_{b,y,x} dy * x(shifted) */
int rcot_dwconv3x3_wgrad(const float* dy, const float* x, float* dw, int B, int C, int H, int W, void* stream);
/* Depthwise 3x3 backward in ONE pass: dx = dw3x3(dy, rotated w) and dwg += sum dy (*) x  (dy is read once). */
int rcot_dwconv3x3_bwd(const float* dy, const float* x, const float* w, float* dx, float* dwg, int B, int C, int H, int W,
                       void* stream);

/* ---- MDTA small-matrix core (Net_Restormer.py:39-43) -------------------------------------- */
/* out[b*R + r] = sum_n x[b*sXb + r*N + n]^2   (|q|^2, |k|^2 rows for F.normalize, :39-40) */
int rcot_row_sumsq(const float* x, float* out, int B, int R, int N, long sXb, void* stream);
/* Gn = Graw/(nq nk^T); A = softmax_rows(tau*Gn)  (Net_Restormer.py:39-43).  c = C/heads <= 96.
 * The fold Mf[b] = W_o * blockdiag_h(A[b,h]) is an rcot_bmm_nn call over (image, head). */
int rcot_attn_softmax(const float* Graw, int S, int ld, const float* sq, const float* temp, float* Gn, float* A, int B,
                      int heads, int c, void* stream);
/* S = 0: Graw is the finished Gram product [B][heads][c][c].  S > 0: Graw is the slab set of rcot_bmm_nt_slabs
 * ([B*heads][S][c][ld]) and is summed here in a fixed order: q k^T then needs no reduce launch of its own. */
/* from dA[b,h] = W_o[:,h]^T dMf[b][:,h] (rcot_bmm_nn): dtau partials [B][heads], Eq and its transpose EqT
 * [B][heads][c][c] (operands of dQ = Eq K + Dq.Q and dK = Eq^T Q + Dk.K), Dq/Dk [B][C]. */
int rcot_attn_bwd_small(const float* dA, const float* A, const float* Gn, const float* sq, const float* temp,
                        float* dtemp_part, float* Eq, float* EqT, float* Dq, float* Dk, int B, int heads, int c,
                        void* stream);
/* The backward of the attention-matrix chain of one block in TWO launches instead of four (c = 48 or 96): from dM
 * [B][C][C] (= dY V^T), W_o [C][C], A, Gn [B][heads][c][c], sq, temp it forms, per head block and row chunk, Mf =
 * W_o blockdiag(A) [B][C][C], the per-image dW_o [B][C][C] and partial dA = W_o^T dM (in ws), then everything
 * rcot_attn_bwd_small returns.  ws >= B*heads*ceil(C*c/3072)*c*c floats. */
int rcot_attn_bwd_fused(const float* dM, const float* Wo, const float* A, const float* Gn, const float* sq, const float* temp,
                        float* Mf, float* dWo_part, float* dtemp_part, float* Eq, float* EqT, float* Dq, float* Dk, int B,
                        int heads, int c, void* ws, long ws_bytes, void* stream);
/* The whole forward attention-matrix chain of one block for SMALL images (N = H*W a multiple of 256, N <= 4096: the 64x64,
 * 32x32 and 16x16 levels; c = 24, 48 or 96; C = heads*c a multiple of 16): from u = [q | k | v] [B][3C][N] (batch stride sUb)
 * it writes sq [B][2C] (|q_i|^2, |k_j|^2: F.normalize, Net_Restormer.py:39-40), Gn and A [B][heads][c][c] (:42-43) and the
 * K-major operand of y = x + (W_o blockdiag(A)) v (:45,49):  MfT[b][h c + j][m] = sum_i A[b,h][i][j] WoT[h c + i][m], with
 * WoT = W_o^T (the WT pack of rcot_pack_weight, leading dim ldwt) and MfT rows ldm apart, images sMb apart.  One launch
 * when N == 256 and c <= 48, two otherwise (partial Gram blocks through ws: B*heads*(N/256)*(c*c + 2c) floats).  Replaces
 * rcot_row_sumsq + rcot_bmm_nt[_slabs] + rcot_attn_softmax + rcot_bmm_nn; exact fp32 (v_mfma_f32_16x16x4_f32).
 * RCOT_EUNSUPPORTED for other shapes (the 128x128 level): use those four. */
int rcot_attn_core_fwd(const float* u, long sUb, const float* temp, const float* WoT, long ldwt, float* sq, float* Gn, float* A,
                       float* MfT, long ldm, long sMb, int B, int heads, int c, int N, float* ws, size_t ws_bytes, void* stream);
/* The backward of the attention-matrix chain of one block in ONE launch (c = 24, 48 or 96; C = heads*c a multiple of 16), one
 * workgroup per (head, image), the three small products on v_mfma_f32_16x16x4_f32 (exact fp32): everything rcot_attn_bwd_fused
 * returns — Mf = W_o blockdiag(A) [B][C][C], the per-image dW_o [B][C][C], dtau partials [B][heads], Eq / Eq^T [B][heads][c][c],
 * Dq / Dk [B][C] — from dM = dY V^T given dense (S = 0: [B][C][C]) or as S <= 8 split-K slabs [B][S][C][ldd] of
 * rcot_bmm_nt_slabs, summed here in a fixed order.  RCOT_EUNSUPPORTED for other head widths: rcot_attn_bwd_fused. */
int rcot_attn_core_bwd(const float* dM, int S, int ldd, const float* Wo, const float* A, const float* Gn, const float* sq,
                       const float* temp, float* Mf, float* dWo_part, float* dtemp_part, float* Eq, float* EqT, float* Dq, float* Dk, int B,
                       int heads, int c, void* stream);
/* dst = beta*dst + sum_b src[b][0..n) */
int rcot_batch_reduce(const float* src, float* dst, int B, long n, float beta, void* stream);

/* ---- critic / minimax-step elementwise pieces (trainer.py:268-308) ---------------------------------------- */
int rcot_lrelu_bwd(const float* dy, const float* a, float* dz, long n, float slope, void* stream);
int rcot_bias_grad(const float* dz, float* db, int B, int C, int P, void* stream);
/* out[r][c] = a*x[r][c] + b*y[r][c] over `rows` rows of `cols` contiguous floats with row strides sx/sy/so
 * (y may be NULL).  Covers residual-conditioning `latent += 0.8*reslatent` (Net_Restormer.py:401),
 * `res = inp - out` (:377), and the channel-slice copies that replace torch.cat (:369). */
int rcot_axpby2d(const float* x, long sx, const float* y, long sy, float* out, long so, long rows, long cols, float a,
                 float b, void* stream);
/* p[0:n) = v (any 4-byte alignment).  The loss seeds of the three half-steps (-1/B, +1/B: the `.mean()` / `-.mean()` of
 * trainer.py:269,274,319 seen from backward) and the zeroing of the flat gradient buffers (`zero_grad()`, :267,:283,:312):
 * with it an iteration issues no launch that is not this library's (host-side launch plans record rcot_* calls only). */
int rcot_fill(float* p, long n, float v, void* stream);
/* out[b] = alpha[b]*t[b] + (1-alpha[b])*f[b]   (trainer.py:286) */
int rcot_lerp(const float* t, const float* f, const float* alpha, float* out, int B, long per, void* stream);
/* norms[b] = ||g_b||; u0 = d/dg [10/Bg * sum_b (||g_b||-1)^2]; *gp_out = 10/Bg * sum_b (||g_b||-1)^2
 * (trainer.py:300-305; Bg = global batch, inv_global_batch = 1/Bg) */
int rcot_gp_penalty(const float* g, float* norms, float* u0, float* gp_out, int B, long per, float inv_global_batch,
                    void* stream);

/* ---- Fourier residual-guided OT cost (trainer.py:320-343) ---------------------------------
 * sums: [2B+2] floats: sum res^2 per sample, sum|out-target| per sample, and the two totals (all-reduce the
 * totals across ranks between rcot_ot_reduce and rcot_ot_grad for data parallelism). */
int rcot_ot_reduce(const float* degraded, const float* restored, const float* target, float* sums, int B, long per,
                   void* stream);
/* L1-spectrum branch (de_id >= 3): gF = d mean|FFT2(res)| / d res, spec[b] = sum |FFT2(res_b)|.  H, W powers of 2. */
int rcot_ot_spectrum(const float* degraded, const float* restored, const int* de_id, float* gF, float* spec, float* ws,
                     size_t ws_bytes, int B, int H, int W, void* stream);
/* dout += d/d(restored) [ sigma*(rmse + sum_i f_i) + Sigma*mean|restored-target| ]; scal = {rmse, sum_i f_i (local),
 * local sum|restored-target| / (global elements)}. */
int rcot_ot_grad(const float* degraded, const float* restored, const float* target, const int* de_id, const float* gF,
                 const float* sums, const float* spec, float* dout, float* scal, int B, long per, float sigma,
                 float Sigma, long global_batch, void* stream);

/* ---- training-patch preparation (the data contract of util/dataset_utils.py:215-281) ----------------------------
 * One sample of TrainDataset.__getitem__ after file decoding, on the device: crop the P x P window at (y0, x0) of the
 * uint8 HWC RGB image(s) [H][W][3], apply dihedral map `mode` (0..7 = util/image_utils.py:133-163 data_augmentation),
 * and write CHW float / 255 (ToTensor, :264-265) into deg_out / clean_out [3][P][P].  deg_img == NULL: synthetic
 * denoising sample, degraded = clip(clean + N(0, noise_sigma^2), 0, 255).astype(uint8) of the AUGMENTED clean patch
 * (util/degradation_utils.py:21-27; counter-based generator seeded by `seed`); otherwise the paired degraded image
 * (derain / dehaze / deblur / lowlight / single) gets the same crop and map. */
int rcot_patch_prep(const unsigned char* deg_img, const unsigned char* clean_img, int H, int W, int y0, int x0, int P,
                    int mode, float noise_sigma, unsigned long long seed, float* deg_out, float* clean_out, void* stream);

/* ---- fused flat-buffer optimizers (trainer.py:121-126) ---------------------------------------------------- */
int rcot_rmsprop_step(float* p, const float* g, float* sq, long n, double lr, double alpha, double eps,
                      double grad_scale, void* stream);
int rcot_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps,
                   int step, double grad_scale, void* stream);

/* ---- the reference's OLDER transport map: MPRNet-style Net.T_net (Net.py:179-216) ----------------------------
 * Its 3x3 / 1x1 convolutions are rcot_conv2d_fwd / _dgrad / _wgrad (KH = KW = 3 or 1, no bias); these are the pieces between them
 * (csrc/mprnet_ops.hip).  Every tensor is contiguous NCHW fp32; "rows" = B*C image planes of N = H*W pixels.
 *  rcot_prelu_fwd / _bwd : nn.PReLU() with ONE slope shared by every CAB (Net.py:185, device scalar `slope`):
 *                          y = x > 0 ? x : slope x;  dx = x > 0 ? dy : slope dy;  dslope[0] += sum_{x <= 0} x dy (fixed summation
 *                          order: per-workgroup partials through ws, >= 4 KiB).
 *  rcot_row_dot          : out[row] = scale * sum_n a[row][n] * (b ? b[row][n] : 1) — the global average pool of CALayer
 *                          (Net.py:40,50; b = NULL, scale = 1/N) and its backward's  dgate[row] = sum_n dout r  (scale = 1).
 *  rcot_row_scale_add    : out[row][n] = a[row][n] * s[row] + (x ? x[row][n] : 0) + (t ? t[row] * tscale : 0) — the gated residual
 *                          of a CAB, res * y + x (Net.py:52,70-72), and its backward dres = dout * gate + dmean / N.
 *  rcot_ca_gate_fwd / _bwd : CALayer.conv_du (Net.py:42-47) on the pooled means [B][C]: hid = relu(W1 mean) [B][Cr], gate =
 *                          sigmoid(W2 hid) [B][C], W1 [Cr][C], W2 [C][Cr] (the 1x1 convolutions' OIHW weights); backward from dgate:
 *                          dmean [B][C], dW1 += , dW2 += (batch summed in image order; ws >= B (C + Cr) floats).  C <= 1024, Cr <= 256 else
 *                          RCOT_EUNSUPPORTED.
 *  rcot_bilinear_down2 / _bwd : nn.Upsample(scale_factor=0.5, bilinear, align_corners=False) of DownSample (Net.py:149): [planes][H][W]
 *                          -> [planes][H/2][W/2] (H, W even: the mean of each 2 x 2 cell) and its adjoint dx = beta dx + 0.25 dy.
 *  rcot_bilinear_up2 / _bwd : scale_factor=2 of SkipUpSample (Net.py:167-176): y [planes][2H][2W] = up(x) + (skip ? skip : 0) with
 *                          PyTorch's source-coordinate rule (clamped at 0, neighbour clamped at the border) and the adjoint gather
 *                          dx [planes][H][W] from dy [planes][2H][2W].
 *  rcot_conv_weight_flip : Wf[ci][co][KH-1-ky][KW-1-kx] = W[co][ci][ky][kx] for n weights of one shape at the element offsets
 *                          table[2 j] (in src) / table[2 j + 1] (in dst) — `table` a DEVICE array of 2 n int64.  rcot_conv2d_fwd(dY, Wf,
 *                          stride 1, pad K/2) then IS rcot_conv2d_dgrad(dY, W): how the 80-channel level takes the forward kernel's
 *                          16-row form for its data gradients (refreshed after every optimizer step, one launch). */
int rcot_prelu_fwd(const float* x, const float* slope, float* y, long n, void* stream);
int rcot_prelu_bwd(const float* dy, const float* x, const float* slope, float* dx, float* dslope, long n, float* ws, size_t ws_bytes,
                   void* stream);
int rcot_row_dot(const float* a, const float* b, float* out, long rows, int N, float scale, void* stream);
int rcot_row_scale_add(const float* a, const float* s, const float* x, const float* t, float tscale, float* out, long rows, int N,
                       void* stream);
int rcot_ca_gate_fwd(const float* mean, const float* W1, const float* W2, float* hid, float* gate, int B, int C, int Cr, void* stream);
int rcot_ca_gate_bwd(const float* dgate, const float* gate, const float* hid, const float* mean, const float* W1, const float* W2,
                     float* dW1, float* dW2, float* dmean, int B, int C, int Cr, float* ws, size_t ws_bytes, void* stream);
int rcot_bilinear_down2(const float* x, float* y, long planes, int H, int W, void* stream);
int rcot_bilinear_down2_bwd(const float* dy, float* dx, long planes, int H, int W, float beta, void* stream);
int rcot_bilinear_up2(const float* x, const float* skip, float* y, long planes, int H, int W, void* stream);
int rcot_bilinear_up2_bwd(const float* dy, float* dx, long planes, int H, int W, void* stream);
int rcot_conv_weight_flip(const float* src, float* dst, const long long* table, int n, int Co, int Ci, int KH, int KW, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RCOT_HIP_H */
