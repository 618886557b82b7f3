/*
 * istvt_hip.h — C ABI of libistvt_hip.so, the MI355X (gfx950) kernels behind the ISTVT
 * video-clip forward/backward hot path.
 *
 * The reference (Vill-Lab/2023-TIFS-ISTVT) is pure PyTorch: it has no FFI for this path; its
 * "interface" is the chain of torch ops inside network/xception.py, network/vivit/module.py
 * and network/vivit/vivit.py.  Each entry point below names the reference ops it replaces
 * (file:line into /root/reference).  INTEGRATION.md shows the ctypes stubs that bind them.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocates everything);
 *     kernels never allocate, free, or retain pointers, and only enqueue on `stream`
 *     (hipStream_t passed as void*; no host synchronisation).
 *   - `dtype`: storage type of activations: 0 = float32 (parity mode), 1 = bfloat16.
 *     Parameters, statistics, parameter gradients and all accumulation are float32.
 *   - return value: 0 = ok; -2 = bad dtype; -3 = bad shape/argument; -(1000+e) = hipError_t e
 *     raised by the launch.  Nothing throws, nothing exits.
 *   - "accumulate" outputs (parameter gradients) are added to, so the caller zeroes them
 *     (they are the .grad buffers).
 *   - row-major everywhere; activations of the transformer are [M = B*F*P][D] with rows
 *     ordered (clip b, frame f, token p); stem activations are NHWC [frames][H][W][C].
 *   - `ld*` arguments are row strides in ELEMENTS (>= the row width, multiples of 8).  The host
 *     keeps transformer activations and the bf16 GEMM operand copies of the weights with rows
 *     padded so that every row starts on a 128-byte line and spans an ODD number of lines
 *     (ops.pad_ld: 728 -> 832, 2912 -> 3008, 512 -> 576, 1536 -> 1600 elements): the LDS-DMA
 *     staging of the GEMMs is priced per cache line touched (tools/dma_probe.hip), and with an
 *     even line count the rows of a column panel fall on half of the memory channels.  The
 *     kernels only require 16-byte aligned rows; pad columns are never read as data and never
 *     written.
 */
#ifndef ISTVT_HIP_H
#define ISTVT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* istvt_stream_t; /* hipStream_t */

#define ISTVT_F32 0
#define ISTVT_BF16 1

/* ---- GEMM with fused epilogue -------------------------------------------------------------
 * C[m][n] = epi(alpha * sum_k A(m,k) B(n,k)) ; a_kc/b_kc = 1: operand stored [rows][K]
 * (K contiguous), 0: stored [K][rows].
 * Replaces nn.Linear (module.py:27,30,74,77,182,183,186; vivit.py:129), 1x1 convs
 * (xception.py:44,57) and the im2col'd 3x3 convs (xception.py:118,122): forward (1,1),
 * input gradient (1,0) and weight gradient (0,0).
 * bias: float[N] or NULL.  residual: T[M][ldr] or NULL (added after activation handling).
 * epi: 0 none; 1 GELU forward (exact erf, module.py:28): C = pre-activation, C2 = gelu;
 *      2 GELU backward: C = acc * gelu'(C2).
 * out_mode: 0 store T; 1 store float; 2 atomicAdd into float C; 3 split z stores its float partial
 *           at C + z*M*ldc (caller sums them with istvt_splitk_reduce).  splitk > 1 needs 2 or 3.
 * col_sum / col_sumsq (both or neither; NULL = off): replica-0 rows of a double[R][2][N] statistics accumulator (R =
 *           istvt_stats_replicas()); the kernel adds the per-column sum and sum of squares of the values it stores --
 *           the train-mode BatchNorm statistics of a 1x1 convolution's output (xception.py:44,57 -> :58,69,75) without a
 *           second pass.  Only the persistent bf16 NT kernel does this (forward of a >= 64-wide, 16-byte aligned
 *           problem with epi 0, no residual, out_mode 0); any other combination returns -3.
 *           col_sum alone (col_sumsq NULL) with epi 2: only the column sums -- the bias gradient of FeedForward's hidden
 *           layer (module.py:27), whose dy is exactly this GEMM's output; fold with istvt_stats_reduce_add.
 * flags:    bit 0 (float32 only): blocked summation -- every 32-deep step of the reduction is summed from zero and added
 *           to the running total, so the rounding error grows with 32 + K/32 terms instead of K (the transformer's
 *           Linears and every weight gradient use it: 10x closer to the float64 reference run on golden G5).  0 = one
 *           sequential fp32 FMA chain: the order that reproduces the reference CPU convolutions' ReLU / arg-max
 *           decisions in the Xception stem (forward and input gradient of its convolutions).
 *           bit 1 (bfloat16 forward on the persistent NT kernel only, -3 otherwise): A is TWO planes of M rows, the
 *           second directly behind the first (A + M * lda elements); output columns at or past 64 * (flags >> 16), a
 *           multiple of 256, take their rows from the second plane.  TemporalResidualAttention (module.py:193-196): q | k
 *           are projections of the frame-differenced LayerNorm output, v of the plain one -- one 728 -> 1536 GEMM.
 *           bit 4 (with epi 1 or 2, -3 otherwise): the GELU pair exchanges the DERIVATIVE instead of the pre-activation.
 *           epi 1 then stores C = gelu'(u) (computed from the fp32 u with the erf / exp of gelu itself), C2 = gelu(u); epi 2
 *           computes C = acc * C2 with C2 = that saved derivative.  FeedForward's backward (module.py:27-30) needs u for
 *           nothing else, so the same bytes carry what the backward would recompute per element; in float32 the result is
 *           bit-identical to the u form.
 *           bits 8..15: CUs, in units of 8 (at most 24 = 192 CUs, -3 otherwise), that a launch of the persistent NT kernel
 *           leaves free: set while a collective's kernels occupy CUs (the data-parallel gradient all-reduce that overlaps
 *           the stem backward, train_CNN.py:185-186 -> parallel.GradBucket), so that every persistent workgroup is resident
 *           at once instead of queueing behind a whole tile list.  An argument of each launch: the library keeps no state
 *           (results are bit-identical for every value: one workgroup computes an output tile in one fixed order). */
int istvt_gemm(const void* A, long lda, int a_kc, const void* B, long ldb, int b_kc, void* C, long ldc, int M, int N,
               int K, const float* bias, const void* residual, long ldr, void* C2, int epi, int out_mode, int splitk,
               float alpha, double* col_sum, double* col_sumsq, int flags, int dtype, istvt_stream_t stream);

/* out[i] += sum_z ws[z*n + i]: second pass of a split-K weight gradient written as partial slabs */
int istvt_splitk_reduce(const float* ws, int splits, long n, float* out, istvt_stream_t stream);

/* Weight gradients of up to 8 nn.Linear layers in one launch pair (the eight of one transformer layer: module.py:27,30,
 * 74,77,182,183,186 as they come out of the backward pass): out_i[N_i][K_i] += dy_i^T x_i, dy_i bf16 [M][N_i] (row
 * stride lddy_i), x_i bf16 [M][K_i] (row stride ldx_i), out_i contiguous float.  The problems share the reduction
 * length M, so one grid of (tile, split) workgroups covers all of them with few reduction splits; ws: scratch of at
 * least splits * sum(N_i K_i) floats.  splits <= 0: chosen so that the grid fills the chip
 * (istvt_wgrad_group_splits returns that count, for sizing ws).  ISTVT_ERR_SHAPE when a problem is not one the
 * 256x256 bf16 weight-gradient kernel takes (N_i, K_i >= 64 and multiples of 8, 16-byte aligned rows). */
int istvt_wgrad_group(int count, const void* const* dy, const long* lddy, const void* const* x, const long* ldx,
                      float* const* out, const int* N, const int* K, int M, int splits, float* ws, long ws_elems,
                      istvt_stream_t stream);
int istvt_wgrad_group_splits(int count, const int* N, const int* K, int M);

/* ---- LayerNorm (module.py:15-21 PreNorm; vivit.py:89,128) ---------------------------------- */
int istvt_layernorm_fwd(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy, float* mean,
                        float* rstd, long M, int D, float eps, int dtype, istvt_stream_t stream);
/* The same plus the frame difference of module.py:193 taken in fp32 before the rounding to the storage type:
 * diff[(b, f, p)] = y[(b, f, p)] for f < 2, y[(b, f, p)] - y[(b, f - 1, p)] for f >= 2; rows ordered (b, f, p). */
int istvt_layernorm_fwd_diff(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy, void* diff,
                             long ldd, float* mean, float* rstd, int B, int F, int P, int D, float eps, int dtype,
                             istvt_stream_t stream);
/* dres (may be NULL) = gradient arriving through the residual connection, added to dx.  dgamma / dbeta accumulate.
 * dcol (may be NULL) accumulates the column sums of dx: the bias gradient of the nn.Linear whose output this LayerNorm
 * normalises (module.py:30,77,186).  ws: float scratch of >= istvt_layernorm_bwd_ws_elems(M, D) elements for the
 * per-workgroup partial sums; the parameter gradients are reduced in a fixed order (bit-reproducible, no atomics). */
int istvt_layernorm_bwd(const void* dy, long ld_dy, const void* x, long ld_x, const float* mean, const float* rstd,
                        const float* gamma, const void* dres, long ld_res, void* dx, long ld_dx, float* dgamma,
                        float* dbeta, float* dcol, float* ws, long ws_elems, long M, int D, int dtype,
                        istvt_stream_t stream);
int istvt_layernorm_bwd_ws_elems(long M, int D);
/* istvt_layernorm_bwd in two halves, for a caller that folds the partial rows off its critical path (another stream, or
 * later): _partial = the row kernel alone (dx, and the per-workgroup partial rows in ws; with_dcol: a third accumulator,
 * the column sums of dx), _reduce = the fixed-order fold of that ws into dgamma / dbeta (and dcol, non-NULL exactly when the
 * partial call had with_dcol).  Same arithmetic and order as the one-call form: bit-identical results. */
int istvt_layernorm_bwd_partial(const void* dy, long ld_dy, const void* x, long ld_x, const float* mean, const float* rstd,
                                const float* gamma, const void* dres, long ld_res, void* dx, long ld_dx, int with_dcol,
                                float* ws, long ws_elems, long M, int D, int dtype, istvt_stream_t stream);
int istvt_layernorm_bwd_reduce(const float* ws, long ws_elems, long M, int D, float* dgamma, float* dbeta, float* dcol,
                               istvt_stream_t stream);

/* ---- spatial attention (SpatialOnlyAttention.forward core, module.py:84-91) ---------------
 * qkv [BF*P][3*heads*dh] (q|k|v, 'b n (h d)') with row stride ldqkv (elements; also dqkv's), out / dout
 * [BF*P][heads*dh] with row stride ldo; lse [BF*P][heads][2] = softmax statistics (row max in the log2 domain,
 * 1/row sum).  Row strides are multiples of 8 elements; line-aligned, odd-line-count rows (ops.pad_ld) are what the
 * GEMMs on either side want. */
int istvt_attn_spatial_fwd(const void* qkv, long ldqkv, void* out, long ldo, float* lse, int BF, int P, int heads,
                           int dh, float scale, int dtype, istvt_stream_t stream);
int istvt_attn_spatial_bwd(const void* qkv, long ldqkv, const void* out, const void* dout, long ldo, const float* lse,
                           float* delta_scratch, void* dqkv, int BF, int P, int heads, int dh, float scale, int dtype,
                           istvt_stream_t stream);
/* fp8 variant of the two entry points above (BASELINE.json configs[4]; bfloat16 storage only): Q, K, V and the
 * softmax probabilities enter the attention MFMAs as OCP e4m3 (v_mfma_f32_16x16x32_fp8_fp8); softmax, statistics,
 * accumulation and the remaining backward products are unchanged.  Same arguments. */
int istvt_attn_spatial_fwd_fp8(const void* qkv, long ldqkv, void* out, long ldo, float* lse, int BF, int P, int heads,
                               int dh, float scale, int dtype, istvt_stream_t stream);
int istvt_attn_spatial_bwd_fp8(const void* qkv, long ldqkv, const void* out, const void* dout, long ldo, const float* lse,
                               float* delta, void* dqkv, int BF, int P, int heads, int dh, float scale, int dtype,
                               istvt_stream_t stream);

/* ---- temporal attention (TemporalResidualAttention.forward core, module.py:192-205) --------
 * qk [B*F*P][2*heads*dh] (q|k) with row stride ldqk (also dqk's); v / dv [B*F*P][heads*dh] with row stride ldv;
 * out / dout [B*F*P][heads*dh] with row stride ldo; rows (b,f,p); F <= 17 (float32), <= 32 (bfloat16).
 * qk and v may be column ranges of ONE [B*F*P][3*heads*dh] projection (v = qk + 2*heads*dh, ldv = ldqk).
 * diff != 0: q and k are the projections of the UN-differenced LayerNorm output; the kernels take the frame
 * difference of module.py:193 on them (q'[f] = q[f] - q[f-1] for f >= 2; exact because to_qk has no bias,
 * module.py:182) and the backward returns gradients with respect to the un-differenced rows.  diff == 0: plain
 * attention over frames (TemporalOnlyAttention, module.py:145-172).  diff == 2 (bfloat16 only, -3 otherwise): q and k
 * ARRIVE differenced (projections of istvt_layernorm_fwd_diff's second output: the reference's own order, the rounding
 * to bfloat16 at the magnitude of the difference); the forward is plain attention, the backward returns dq, dk with
 * respect to the UN-differenced projections (d q[f] = d q'[f] - d q'[f+1] for f >= 1) so that one input-gradient GEMM
 * over [dq | dk | dv] follows as with diff == 1. */
int istvt_attn_temporal_fwd(const void* qk, long ldqk, const void* v, long ldv, void* out, long ldo, int B, int F, int P,
                            int heads, int dh, float scale, int diff, int dtype, istvt_stream_t stream);
int istvt_attn_temporal_bwd(const void* qk, long ldqk, const void* v, long ldv, const void* dout, long ldo, void* dqk,
                            void* dv, int B, int F, int P, int heads, int dh, float scale, int diff, int dtype,
                            istvt_stream_t stream);

/* ---- token assembly (DSTTr.forward, vivit.py:133-142) -------------------------------------- */
int istvt_tokens_fwd(const void* feats, const float* space, const float* temporal, const float* pos, void* x, long ldx,
                     int B, int F, int P, int D, int pos_rows, int dtype, istvt_stream_t stream);
/* dspace / dtemporal / dpos accumulate; ws = float scratch of (P + F - 1) * D elements (per-workgroup partial rows of the
 * two token gradients, folded in a fixed order: no floating-point atomics) */
int istvt_tokens_bwd(const void* dx, long lddx, void* dfeats, float* dspace, float* dtemporal, float* dpos, float* ws,
                     int B, int F, int P, int D, int pos_rows, int dtype, istvt_stream_t stream);

/* frame difference of module.py:193 (adjoint = 1: its transpose, for the backward) */
int istvt_frame_diff(const void* x, void* out, int B, int F, int P, int D, int adjoint, int dtype,
                     istvt_stream_t stream);

/* ==== Xception entry flow (network/xception.py:193-206), NHWC activations [frames][H][W][C] ====
 * A finalized BatchNorm travels as ONE device pointer `bnp` to a float[4][C] pack
 * {mean, rstd, scale = gamma*rstd, beta}; consumers apply z = (u - mean)*scale + beta. */

/* Per-channel statistics accumulators are double[R][2][C] with R = istvt_stats_replicas(): producers
 * (bn_stats, bn_bwd_stats, the fused sums of dwconv3x3) take pointers to replica 0's two rows and add
 * into replica (workgroup % R) to spread same-address atomics; istvt_stats_reduce folds replicas
 * 1..R-1 into replica 0, which finalize / bwd_apply then read. */
int istvt_stats_replicas(void);
int istvt_stats_reduce(double* acc, int C, istvt_stream_t stream);
/* out[c] += sum over the replicas of row 0: folds column sums accumulated by istvt_gemm(col_sum, NULL) into a float gradient */
int istvt_stats_reduce_add(const double* acc, int C, float* out, istvt_stream_t stream);

/* train-mode nn.BatchNorm2d (xception.py:58,69,75,119,123): sum/sumsq = rows 0/1 of replica 0 */
int istvt_bn_stats(const void* x, double* sum, double* sumsq, long M, int C, int dtype, istvt_stream_t stream);
/* use_batch=1: batch statistics (+ running-stat update, momentum/unbiased var as torch);
 * use_batch=0: running statistics (eval mode).  Writes the pack. */
int istvt_bn_finalize(const double* sum, const double* sumsq, double count, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, float* bnp, int C,
                      int use_batch, int update_running, istvt_stream_t stream);
int istvt_bn_apply(const void* x, const float* bnp, void* y, long M, int C, int relu, int dtype,
                   istvt_stream_t stream);
/* s1 += sum dz, s2 += sum dz*xhat (double[C]);  du = gamma*rstd*(dz - s1/M - xhat*s2/M), dgamma += s2, dbeta += s1 */
int istvt_bn_bwd_stats(const void* dz, const void* u, const float* bnp, double* s1, double* s2, long M, int C,
                       int dtype, istvt_stream_t stream);
/* batch_stats = 0: eval-mode BatchNorm (running statistics are constants): du = gamma*rstd*dz */
int istvt_bn_bwd_apply(const void* dz, const void* u, const float* bnp, const float* gamma, const double* s1,
                       const double* s2, void* du, float* dgamma, float* dbeta, long M, int C, int batch_stats,
                       int dtype, istvt_stream_t stream);
/* tail of a stride-1 Block (xception.py:91-100 without the MaxPool): out = bn_x(x) + (bns ? bn_s(skip) : skip) */
int istvt_bn_add_fwd(const void* x, const float* bnx, const void* skip, const float* bns, void* out, long M, int C,
                     int dtype, istvt_stream_t stream);

/* conv1 (3->32, 3x3, s2, p0; xception.py:118): NCHW float clip -> col[M][32] ((dy,dx,ci) + 5 zero cols); adjoint */
int istvt_im2col_conv1(const float* x, void* col, int frames, int S, int dtype, istvt_stream_t stream);
int istvt_col2im_conv1(const void* dcol, float* dx, int frames, int S, int dtype, istvt_stream_t stream);
/* conv2 (32->64, 3x3, p0; xception.py:122): NHWC source with optional BN pack (+ReLU) on load -> col[M][9*C];
 * adjoint gathers dcol and masks by relu'(bn(u)) */
int istvt_im2col3x3(const void* src, const float* bnp, int relu, void* col, int frames, int H, int W, int C,
                    int dtype, istvt_stream_t stream);
int istvt_col2im3x3(const void* dcol, const void* u, const float* bnp, void* dz, int frames, int H, int W, int C,
                    int dtype, istvt_stream_t stream);

/* The same two convolutions computed directly (no im2col matrix in HBM); xception.py:118-123,193-199.
 * conv1_fwd: x float [frames][3][S][S], w float [32][3][3][3] (conv1.weight as stored) -> u1 [frames*Ho*Wo][32] (dtype).
 * conv2_* are bf16 only: u1 = raw conv1 output [frames][H][W][32], bnp = bn1's finalized pack (ReLU implied),
 * w = bf16 [64][(dy,dx,ci)].  fwd -> u2 [frames][H-2][W-2][64]; dgrad: du2 -> dz1 [frames][H][W][32] masked by
 * relu'(bn1(u1)); wgrad: dw float [64][(dy,dx,ci)] += sum over pixels, slabs = caller-owned float workspace of
 * istvt_conv2_wgrad_slabs() * 64 * 288 elements. */
int istvt_conv1_fwd(const float* x, const float* w, void* u1, int frames, int S, int dtype, istvt_stream_t stream);
/* conv1 weight gradient: du1 [frames*Ho*Wo][32] (dtype; rounded to bf16 for the MFMA), x as above ->
 * dw float [32][32] +=, column k = ci*9 + dy*3 + dx (conv1.weight's own order; columns 27..31 unused);
 * slabs = caller-owned float workspace of istvt_conv1_wgrad_slabs() * 1024 elements.  Ho <= 128. */
int istvt_conv1_wgrad(const void* du1, const float* x, float* slabs, float* dw, int frames, int S, int dtype,
                      istvt_stream_t stream);
int istvt_conv1_wgrad_slabs(void);
int istvt_conv2_fwd(const void* u1, const float* bnp, const void* w, void* u2, int frames, int H, int W,
                    istvt_stream_t stream);
int istvt_conv2_dgrad(const void* du2, const void* w, const void* u1, const float* bnp, void* dz1, int frames, int H,
                      int W, istvt_stream_t stream);
int istvt_conv2_wgrad(const void* du2, const void* u1, const float* bnp, float* slabs, float* dw, int frames, int H,
                      int W, istvt_stream_t stream);
int istvt_conv2_wgrad_slabs(void);

/* depthwise 3x3 s1 p1 (SeparableConv2d.conv1, xception.py:43), LDS-tiled.  w: float[9][C] (tap-major);
 * the weight gradient dw is float[C][9] (PyTorch order).
 * forward: in_bn (+in_relu) = the preceding BatchNorm+ReLU applied on load (xception.py:67,73).
 * input gradient (flip=1): epilogue = ReLU mask of msrc (optionally through m_bn) before (mask_pre) /
 * after (mask_post) adding the stride-2 skip-path gradient addsrc[f][y/2][x/2] (Ha = (H-1)/2+1, Wa likewise) or, for
 * the stride-1 blocks, the full-resolution one addsrc[f][y][x] (Ha = H, Wa = W), plus fused
 * BatchNorm-backward sums st_s1/st_s2 (double[C]) of the result w.r.t. m_bn. */
int istvt_dwconv3x3(const void* in, const float* w, void* out, int frames, int H, int W, int C, const float* in_bn,
                    int in_relu, int flip, const void* msrc, const float* m_bn, int mask_pre, int mask_post,
                    const void* addsrc, int Ha, int Wa, double* st_s1, double* st_s2, int dtype,
                    istvt_stream_t stream);
/* dw float [C][9] accumulates; ws = float scratch of istvt_dwconv3x3_wgrad_ws_elems(...) elements (one partial slab
 * per workgroup slot, folded in slot order: bit-reproducible) */
int istvt_dwconv3x3_wgrad(const void* in, const float* in_bn, int in_relu, const void* dout, float* dw, float* ws,
                          long ws_elems, int frames, int H, int W, int C, int dtype, istvt_stream_t stream);
int istvt_dwconv3x3_wgrad_ws_elems(int frames, int H, int W, int C);

/* Block tail (xception.py:88,91-100): out = maxpool3x3s2p1(bn_x(x)) + bn_s(skip); argmax: uint8 per output element */
int istvt_pool_add_fwd(const void* x, const float* bnx, const void* skip, const float* bns, void* out,
                       unsigned char* argmax, int frames, int H, int W, int C, int dtype, istvt_stream_t stream);
/* u (NULL = off): the input of the block's last BatchNorm (xception.py:75), whose output the pooling consumed; with it
 * bnp (that BatchNorm's pack) and s1 / s2 (replica-0 rows of a double[R][2][C] accumulator): the kernel adds the
 * BatchNorm-backward sums  sum dz,  sum dz * xhat  of the values it stores -- no istvt_bn_bwd_stats pass over dz and u */
int istvt_pool_bwd(const void* dout, const unsigned char* argmax, void* dz, int frames, int H, int W, int C,
                   const void* u, const float* bnp, double* s1, double* s2, int dtype, istvt_stream_t stream);
/* input of the stride-2 1x1 skip conv (xception.py:57): out[f][y][x] = in[f][2y][2x] */
int istvt_subsample2(const void* in, void* out, int frames, int H, int W, int C, int dtype, istvt_stream_t stream);

/* ==== next rows: Xception exit flow head, ablation transformers, dropout ==== */
/* Xception.logits (xception.py:208-213): out[f][c] = mean_hw relu(x[f][hw][c]) on NHWC features; relu = 0: plain mean */
int istvt_relu_avgpool_fwd(const void* x, void* out, int frames, int HW, int C, int relu, int dtype,
                           istvt_stream_t stream);
int istvt_relu_avgpool_bwd(const void* x, const void* dout, void* dx, int frames, int HW, int C, int relu, int dtype,
                           istvt_stream_t stream);
/* Token assembly of ViViT / VanillaTr (vivit.py:60-67,74-75,180-186): S sequences of n rows -> n + 1 rows,
 * out[s][0] = tok (+ pos[s % period][0]), out[s][1+i] = src[s][i] (+ pos[s % period][1+i]); pos float
 * [period][pos_rows][D] or NULL.  bwd: dsrc (may be NULL), dtok / dpos accumulate (float). */
int istvt_prepend_fwd(const void* src, const float* tok, const float* pos, void* out, long ldo, long S, int n, int D,
                      int period, int pos_rows, int dtype, istvt_stream_t stream);
/* ws (needed with dtok) = float scratch of istvt_prepend_bwd_ws_rows(S, period, dpos != NULL) * D elements */
int istvt_prepend_bwd(const void* dout, long ldd, void* dsrc, float* dtok, float* dpos, float* ws, long S, int n, int D,
                      int period, int pos_rows, int dtype, istvt_stream_t stream);
int istvt_prepend_bwd_ws_rows(long S, int period, int has_pos);
/* x.mean(dim=1) of [S][n][D] (ViViT pool='mean', vivit.py:79) and its adjoint */
int istvt_seq_mean_fwd(const void* x, long ldx, void* out, long S, int n, int D, int dtype, istvt_stream_t stream);
int istvt_seq_mean_bwd(const void* dout, void* dx, long ldx, long S, int n, int D, int dtype, istvt_stream_t stream);
/* nn.Dropout(p) in training mode (module.py:29,31,78,187; models_copy.py:41-44): Philox4x32-10 keyed by `seed`,
 * mask = one byte per element (1 = kept), y = mask ? x / (1 - p) : 0; backward applies the stored mask. */
int istvt_dropout_fwd(const void* x, long ldx, void* y, long ldy, unsigned char* mask, long M, int D, float p,
                      unsigned long long seed, int dtype, istvt_stream_t stream);
int istvt_dropout_bwd(const void* dy, long ldy, const unsigned char* mask, void* dx, long ldx, long M, int D, float p,
                      int dtype, istvt_stream_t stream);
/* out = a + b on row-strided [M][D] views: the residual add behind an active Dropout (module.py:78,187 with p > 0) */
int istvt_add(const void* a, long lda, const void* b, long ldb, void* out, long ldo, long M, int D, int dtype,
              istvt_stream_t stream);

/* ---- helpers -------------------------------------------------------------------------------- */
/* out[n] += sum_m x[m][n]  (bias gradients); ws = float scratch of istvt_colsum_ws_elems(M, N) elements (one partial row
 * per row block, folded in a fixed order) */
int istvt_colsum(const void* x, float* out, long M, int N, long ld, float* ws, long ws_elems, int dtype,
                 istvt_stream_t stream);
int istvt_colsum_ws_elems(long M, int N);
/* out[i] += sum_{r < rows} ws[r * n + i], rows added in index order by one writer per element: the second stage of every
 * per-column sum of this library (fp32 split-K slabs of narrow weight gradients, bias / token / BatchNorm-affine
 * gradients).  No kernel adds float32 values with atomics; the only atomics left are the double-precision statistics
 * accumulators (BatchNorm sums, the GELU-backward column sums), whose addends are float32 partial sums: a 53-bit
 * accumulator adds 24-bit addends exactly over any realistic spread of magnitudes, so their order does not change
 * the result.  A training step gives the same bits on every run (tests/test_model_gpu.py). */
int istvt_rows_reduce(const float* ws, int rows, long n, float* out, istvt_stream_t stream);
int istvt_cast(const void* in, int in_dtype, void* out, int out_dtype, long n, istvt_stream_t stream);
/* rows x cols cast between row-strided buffers (bf16 operand copies of fp32 weights with line-aligned rows) */
int istvt_cast2d(const void* in, int in_dtype, long ldi, void* out, int out_dtype, long ldo, long rows, int cols,
                 istvt_stream_t stream);
/* fp32 [R][C] -> bf16 [R][C] (row stride ldo) and its transpose bf16 [C][R] (row stride ldt) in one pass: the
 * operand copies of a Linear weight for the forward and the input-gradient GEMMs (R, C multiples of 8) */
int istvt_cast_transpose(const float* in, long ldi, void* out, long ldo, void* outT, long ldt, int R, int C,
                         istvt_stream_t stream);
/* the same for `count` weights in one launch (groups of 32): the bf16 operand copies of every nn.Linear weight the
 * optimizer has just changed, refreshed at the start of a step instead of one launch per weight */
int istvt_cast_transpose_group(int count, const float* const* in, const long* ldi, void* const* out, const long* ldo,
                               void* const* outT, const long* ldt, const int* R, const int* C, istvt_stream_t stream);

/* ---- fused optimizer steps over flat float buffers (train_CNN.py:196-201: torch.optim.SGD(momentum) / AdamW) ----
 * p, g, state: n floats each, 16-byte aligned (parallel.GradBucket(flatten_params=True)).  Semantics are torch.optim's:
 * sgd: g' = g + wd p; buf = first_step ? g' : momentum buf + (1 - dampening) g'; p -= lr (nesterov ? g' + momentum buf : buf)
 * adamw (amsgrad off): p *= 1 - lr wd; m, v moments; p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)
 * zero_grad != 0 also writes zeros over g (the next step's zero-grad pass).  grad_scale multiplies g on load: the
 * 1 / world_size of the data-parallel gradient mean (the all-reduce then only sums; no separate scaling pass).
 * momentum == 0 ignores dampening, as torch does. */
int istvt_sgd_momentum(float* p, float* g, float* buf, long n, float lr, float momentum, float dampening,
                       float weight_decay, int nesterov, int first_step, int zero_grad, float grad_scale,
                       istvt_stream_t stream);
int istvt_adamw(float* p, float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                float weight_decay, long step, int zero_grad, float grad_scale, istvt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ISTVT_HIP_H */
