// Xception entry-flow kernels (reference: network/xception.py:39-101,118-128,193-206), NHWC
// activations [frames][H][W][C] in the storage dtype T, fp32 statistics and parameters.
//
//   im2col (conv1: NCHW fp32 clip -> [M][32]; conv2: NHWC -> [M][9*C]) feeding the MFMA GEMM
//   col2im (conv2 input gradient)
//   train-mode BatchNorm: per-channel statistics (column reductions, fp64 atomics), finalize
//       (mean/rstd/scale/shift + running-stat update), apply, backward statistics / apply
//   depthwise 3x3 (pad 1) with an LDS input tile (halo loaded once per workgroup, the
//       preceding BN-apply+ReLU fused into the tile load): forward, input gradient (flipped
//       taps, ReLU mask / strided skip-gradient add / BN-backward statistics in the epilogue),
//       weight gradient
//   MaxPool(3,2,1) fused with both BN-applies and the skip add; its backward
//   stride-2 pixel subsample for the 1x1 stride-2 skip convolutions
//
// All of these are HBM-bound (AI <= 4.5 flop/B): coalesced 16-byte accesses along C.
#include "common.h"
#include <cstdlib>

// A finalized BatchNorm is passed around as ONE pointer `bnp` to a float [4][C] pack:
//   row 0 mean, row 1 rstd, row 2 scale = gamma * rstd, row 3 beta.
// Consumers apply it in the centred form  z = (u - mean) * scale + beta  (NOT u*scale + shift:
// with |mean| >> std the folded shift loses ~eps*|mean|/std and flips ReLU masks near zero).
__device__ __forceinline__ void bn_affine8(float (&v)[8], const float* bnp, int C, int c) {
    float mu[8], sc[8], be[8];
    load8(bnp + c, mu);
    load8(bnp + 2 * C + c, sc);
    load8(bnp + 3 * C + c, be);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (v[j] - mu[j]) * sc[j] + be[j];
}

// ============================================================================================
// column reductions over a [M][C] matrix, C % 8 == 0
// ============================================================================================
// Per-channel statistics are accumulated with fp64 atomics.  Thousands of workgroups adding to the
// SAME C addresses serialise at the memory side (measured: every bn_stats launch took ~400 us
// whatever its size), so accumulators are replicated: a statistics buffer is double[R][2][C],
// workgroup b adds into replica b % R, and istvt_stats_reduce folds replicas 1..R-1 into replica 0,
// which is what the consumers (finalize, backward-apply) read.  The C ABI passes pointers to
// replica 0's two rows.
constexpr int STAT_REPLICAS = ISTVT_STAT_REPLICAS;

__global__ void stats_reduce_kernel(double* acc, int n2c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2c) return;
    double s = acc[i];
    for (int r = 1; r < STAT_REPLICAS; ++r) s += acc[(long)r * n2c + i];
    acc[i] = s;
}

// out[c] += sum over the replicas of row 0 (the column sums a GEMM epilogue accumulated: a bias gradient)
__global__ void stats_reduce_add_kernel(const double* acc, int C, float* out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int r = 0; r < STAT_REPLICAS; ++r) s += acc[(long)r * 2 * C + c];
    out[c] += (float)s;
}

extern "C" int istvt_stats_replicas() { return STAT_REPLICAS; }

extern "C" int istvt_stats_reduce_add(const double* acc, int C, float* out, hipStream_t stream) {
    if (C <= 0 || !acc || !out) return ISTVT_ERR_SHAPE;
    hipLaunchKernelGGL(stats_reduce_add_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, acc, C, out);
    return istvt_check_launch();
}

extern "C" int istvt_stats_reduce(double* acc, int C, hipStream_t stream) {
    if (C <= 0) return ISTVT_ERR_SHAPE;
    hipLaunchKernelGGL(stats_reduce_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, stream, acc, 2 * C);
    return istvt_check_launch();
}
template <int NACC, typename F>
__device__ __forceinline__ void colreduce_block(F f, double* const (&out)[NACC], long M, int C, int rows_per_block) {
    __shared__ float red[NACC][256][8];
    const int VC = C / 8;
    const int vc0 = blockIdx.x * 256;
    const int vcg = min(256, VC - vc0);
    const int RS = 256 / vcg;
    const int tcol = threadIdx.x % vcg, trow = threadIdx.x / vcg;
    float acc[NACC][8];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[a][i] = 0.f;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(M, r0 + (long)rows_per_block);
    const int c0 = (vc0 + tcol) * 8;
    if (trow < RS) {
#pragma unroll 4
        for (long m = r0 + trow; m < r1; m += RS) f(m, c0, acc);      // 4 rows of loads in flight per thread
    }
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < 8; ++i) red[a][threadIdx.x][i] = acc[a][i];
    __syncthreads();
    if (threadIdx.x < vcg) {
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float s = 0.f;
                for (int t = 0; t < RS; ++t) s += red[a][threadIdx.x + t * vcg][i];
                atomicAdd(out[a] + (long)(blockIdx.y % STAT_REPLICAS) * 2 * C + c0 + i, (double)s);
            }
    }
}

static void colreduce_grid(long M, int C, dim3& grid, int& rpb) {
    const int gx = (C / 8 + 255) / 256;
    static const long wg_target = istvt_tune("ISTVT_COLRED_BLOCKS", 512);   // 2048: +0.6 ms per step (fp64 atomics tail per workgroup); 256: +0.03
    long target = wg_target / gx;
    if (target < 1) target = 1;
    rpb = (int)((M + target - 1) / target);
    if (rpb < 32) rpb = 32;
    grid = dim3(gx, (unsigned)((M + rpb - 1) / rpb));
}

// sum[c] += sum_m x ; sumsq[c] += sum_m x^2       (nn.BatchNorm2d batch statistics)
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, double* sum, double* sumsq, long M,
                                                       int C, int rpb) {
    double* const outs[2] = {sum, sumsq};
    colreduce_block<2>([&](long m, int c0, float (&acc)[2][8]) {
        float v[8];
        load8(x + m * C + c0, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[0][i] += v[i]; acc[1][i] += v[i] * v[i]; }
    }, outs, M, C, rpb);
}

// s1[c] += sum_m dz ; s2[c] += sum_m dz * xhat,  xhat = (u - mean) * rstd
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const T* __restrict__ dz, const T* __restrict__ u,
                                                           const float* __restrict__ bnp, double* s1, double* s2,
                                                           long M, int C, int rpb) {
    const float* mean = bnp;
    const float* rstd = bnp + C;
    double* const outs[2] = {s1, s2};
    colreduce_block<2>([&](long m, int c0, float (&acc)[2][8]) {
        float d[8], uv[8], mu[8], rs[8];
        load8(dz + m * C + c0, d);
        load8(u + m * C + c0, uv);
        load8(mean + c0, mu);
        load8(rstd + c0, rs);
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[0][i] += d[i]; acc[1][i] += d[i] * (uv[i] - mu[i]) * rs[i]; }
    }, outs, M, C, rpb);
}

// finalize: batch (use_batch=1) or running statistics -> pack {mean, rstd, scale = g*rstd, beta};
// updates running stats like torch (momentum, unbiased variance) when update_running.
__global__ void bn_finalize_kernel(const double* sum, const double* sumsq, double count, const float* gamma,
                                   const float* beta, float* rmean, float* rvar, float momentum, float eps,
                                   float* bnp, int C, int use_batch, int update_running) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double mean, var;
    if (use_batch) {
        mean = sum[c] / count;
        var = sumsq[c] / count - mean * mean;
        if (var < 0) var = 0;
        if (update_running) {
            const double unb = count > 1 ? var * count / (count - 1) : var;
            rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
            rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
        }
    } else {
        mean = rmean[c];
        var = rvar[c];
    }
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    bnp[c] = (float)mean;
    bnp[C + c] = rstd;
    bnp[2 * C + c] = sc;
    bnp[3 * C + c] = beta[c];
}

// y = (x - mean) * scale + beta (optionally ReLU)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ bnp,
                                                       T* __restrict__ y, long M, int C, int relu) {
    const int vpr = C / 8;
    const long nvec = M * vpr, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        const int c0 = (int)(i % vpr) * 8;
        float v[8];
        load8(x + i * 8, v);
        bn_affine8(v, bnp, C, c0);
        if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        store8(y + i * 8, v);
    }
}

// du = gamma*rstd * (dz - s1/M - xhat * s2/M);  dgamma += s2 ; dbeta += s1 (block 0 only)
// relu_mask: dz is first masked by (scale*u + shift > 0) -- only legal when s1/s2 were computed
// from the masked dz, so the mask is applied by the producer instead; kept out of this kernel.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ u,
                                                           const float* __restrict__ bnp,
                                                           const float* __restrict__ gamma, const double* s1,
                                                           const double* s2, T* __restrict__ du, float* dgamma,
                                                           float* dbeta, long M, int C, int batch_stats) {
    const int vpr = C / 8;
    const long nvec = M * vpr;
    // eval mode (running statistics are constants): du = gamma * rstd * dz, no mean / projection terms
    const float invM = batch_stats ? 1.0f / (float)M : 0.0f;
    const float* mean = bnp;
    const float* rstd = bnp + C;
    // A thread keeps ONE channel chunk for all its vectors (the stride is a multiple of the chunks per row), so the
    // per-channel constants are loaded once: read per vector they were 224 bytes of L1 traffic beside 32 bytes of
    // payload, and the kernel ran at the rate of the load path (3.7 TB/s), not of HBM.
    const long nthr = ((long)gridDim.x * 256) / vpr * vpr;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t < nthr) {
        const int c0 = (int)(t % vpr) * 8;
        float mu[8], rs[8], g[8], k1[8], k2[8];
        load8(mean + c0, mu);
        load8(rstd + c0, rs);
        load8(gamma + c0, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) { k1[j] = (float)s1[c0 + j] * invM; k2[j] = (float)s2[c0 + j] * invM; }
        long i = t;
        for (; i + nthr < nvec; i += 2 * nthr) {                 // two vectors in flight
            float d0[8], u0[8], d1[8], u1[8];
            load8(dz + i * 8, d0);
            load8(u + i * 8, u0);
            load8(dz + (i + nthr) * 8, d1);
            load8(u + (i + nthr) * 8, u1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x0 = (u0[j] - mu[j]) * rs[j], x1 = (u1[j] - mu[j]) * rs[j];
                d0[j] = g[j] * rs[j] * (d0[j] - k1[j] - x0 * k2[j]);
                d1[j] = g[j] * rs[j] * (d1[j] - k1[j] - x1 * k2[j]);
            }
            store8(du + i * 8, d0);
            store8(du + (i + nthr) * 8, d1);
        }
        if (i < nvec) {
            float d0[8], u0[8];
            load8(dz + i * 8, d0);
            load8(u + i * 8, u0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x0 = (u0[j] - mu[j]) * rs[j];
                d0[j] = g[j] * rs[j] * (d0[j] - k1[j] - x0 * k2[j]);
            }
            store8(du + i * 8, d0);
        }
    }
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += 256) {
            if (dgamma) dgamma[c] += (float)s2[c];
            if (dbeta) dbeta[c] += (float)s1[c];
        }
    }
}

// out = bn_x(x) + (bns ? bn_s(skip) : skip): the tail of a stride-1 Block (xception.py:91-100, no MaxPool)
template <typename T>
__global__ __launch_bounds__(256) void bn_add_kernel(const T* __restrict__ x, const float* __restrict__ bnx,
                                                     const T* __restrict__ skip, const float* __restrict__ bns,
                                                     T* __restrict__ out, long M, int C) {
    const int vpr = C / 8;
    const long nvec = M * vpr, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        const int c0 = (int)(i % vpr) * 8;
        float v[8], sv[8];
        load8(x + i * 8, v);
        load8(skip + i * 8, sv);
        bn_affine8(v, bnx, C, c0);
        if (bns) bn_affine8(sv, bns, C, c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = to_f32(from_f32<T>(v[j])) + sv[j];     // the value a separate BN pass would store
        store8(out + i * 8, v);
    }
}

static int ew_grid(long nvec) {
    static const long cap = istvt_tune("ISTVT_EW_BLOCKS", 65536);   // measured: 4096 -> 65536 workgroups = -6 % on the BN-backward apply, -4 % on the pools
    long b = (nvec + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int istvt_bn_stats(const void* x, double* sum, double* sumsq, long M, int C, int dtype, hipStream_t stream) {
    if (M <= 0 || C <= 0 || C % 8 != 0) return ISTVT_ERR_SHAPE;
    dim3 grid; int rpb;
    colreduce_grid(M, C, grid, rpb);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_stats_kernel<T>), grid, dim3(256), 0, stream, (const T*)x, sum, sumsq, M, C, rpb));
    return istvt_check_launch();
}

extern "C" int istvt_bn_finalize(const double* sum, const double* sumsq, double count, const float* gamma,
                                 const float* beta, float* rmean, float* rvar, float momentum, float eps, float* bnp,
                                 int C, int use_batch, int update_running, hipStream_t stream) {
    if (C <= 0) return ISTVT_ERR_SHAPE;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, sum, sumsq, count, gamma, beta,
                       rmean, rvar, momentum, eps, bnp, C, use_batch, update_running);
    return istvt_check_launch();
}

extern "C" int istvt_bn_apply(const void* x, const float* bnp, void* y, long M, int C, int relu, int dtype,
                              hipStream_t stream) {
    if (M <= 0 || C % 8 != 0) return ISTVT_ERR_SHAPE;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(ew_grid(M * (C / 8))), dim3(256), 0, stream,
                                             (const T*)x, bnp, (T*)y, M, C, relu));
    return istvt_check_launch();
}

extern "C" int istvt_bn_add_fwd(const void* x, const float* bnx, const void* skip, const float* bns, void* out, long M,
                                int C, int dtype, hipStream_t stream) {
    if (M <= 0 || C % 8 != 0 || !bnx) return ISTVT_ERR_SHAPE;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_add_kernel<T>), dim3(ew_grid(M * (C / 8))), dim3(256), 0, stream,
                                             (const T*)x, bnx, (const T*)skip, bns, (T*)out, M, C));
    return istvt_check_launch();
}

extern "C" int istvt_bn_bwd_stats(const void* dz, const void* u, const float* bnp, double* s1, double* s2, long M,
                                  int C, int dtype, hipStream_t stream) {
    if (M <= 0 || C % 8 != 0) return ISTVT_ERR_SHAPE;
    dim3 grid; int rpb;
    colreduce_grid(M, C, grid, rpb);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_bwd_stats_kernel<T>), grid, dim3(256), 0, stream, (const T*)dz,
                                             (const T*)u, bnp, s1, s2, M, C, rpb));
    return istvt_check_launch();
}

extern "C" int istvt_bn_bwd_apply(const void* dz, const void* u, const float* bnp,
                                  const float* gamma, const double* s1, const double* s2, void* du, float* dgamma,
                                  float* dbeta, long M, int C, int batch_stats, int dtype, hipStream_t stream) {
    if (M <= 0 || C % 8 != 0) return ISTVT_ERR_SHAPE;
    // few vectors per thread would load the per-channel constants as often as before: a grid of resident size
    static const long bnb_cap = istvt_tune("ISTVT_BNB_BLOCKS", 2048);   // sweep 1024 .. 16384: 2.04 .. 2.19 ms per step (flat); 65536: 2.56
    long bnb = (M * (C / 8) + 255) / 256;
    if (bnb > bnb_cap) bnb = bnb_cap;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3((unsigned)bnb), dim3(256), 0,
                                             stream, (const T*)dz, (const T*)u, bnp, gamma, s1, s2, (T*)du,
                                             dgamma, dbeta, M, C, batch_stats));
    return istvt_check_launch();
}

// ============================================================================================
// im2col / col2im for the two dense 3x3 convolutions (conv1: 3->32 s2 p0, conv2: 32->64 s1 p0)
// column order is (dy, dx, ci): the GEMM weight is W.permute(0,2,3,1).reshape(Cout, 9*Cin)
// ============================================================================================
// conv1: x NCHW fp32 [Fr][3][S][S] -> col [Fr*Ho*Wo][32] (27 taps + 5 zero columns)
template <typename T>
__global__ __launch_bounds__(256) void im2col_c3s2_kernel(const float* __restrict__ x, T* __restrict__ col, long Mo,
                                                          int S, int Ho, int Wo) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= Mo) return;
    const int xo = (int)(m % Wo), yo = (int)((m / Wo) % Ho);
    const long f = m / ((long)Wo * Ho);
    float v[32];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
                v[(dy * 3 + dx) * 3 + ci] = x[((f * 3 + ci) * S + 2 * yo + dy) * S + 2 * xo + dx];
#pragma unroll
    for (int i = 27; i < 32; ++i) v[i] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = v[c * 8 + i];
        store8(col + m * 32 + c * 8, t);
    }
}

// NHWC source, stride 1, pad 0, optional BatchNorm pack (+ReLU) applied on load
template <typename T>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const T* __restrict__ src, const float* __restrict__ bnp,
                                                        int relu, T* __restrict__ col, long Mo, int H, int W, int C,
                                                        int Ho, int Wo) {
    const int vpr = C / 8;
    const long nitems = Mo * 9 * vpr, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nitems; i += stride) {
        const int ch = (int)(i % vpr);
        const int tap = (int)((i / vpr) % 9);
        const long m = i / (9 * vpr);
        const int xo = (int)(m % Wo), yo = (int)((m / Wo) % Ho);
        const long f = m / ((long)Wo * Ho);
        const int dy = tap / 3, dx = tap % 3;
        float v[8];
        load8(src + ((f * H + yo + dy) * W + xo + dx) * C + ch * 8, v);
        if (bnp) bn_affine8(v, bnp, C, ch * 8);
        if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        store8(col + m * 9 * C + tap * C + ch * 8, v);
    }
}

// dx[f,yi,xi,c] = sum_{dy,dx} dcol[(f,yi-dy,xi-dx)][(dy,dx,c)], then masked by relu'(bn(u))
template <typename T>
__global__ __launch_bounds__(256) void col2im3x3_kernel(const T* __restrict__ dcol, const T* __restrict__ u,
                                                        const float* __restrict__ bnp, T* __restrict__ dz, long Mi,
                                                        int H, int W, int C, int Ho, int Wo) {
    const int vpr = C / 8;
    const long nitems = Mi * vpr, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nitems; i += stride) {
        const int ch = (int)(i % vpr);
        const long m = i / vpr;
        const int xi = (int)(m % W), yi = (int)((m / W) % H);
        const long f = m / ((long)W * H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yo = yi - dy;
            if (yo < 0 || yo >= Ho) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xo = xi - dx;
                if (xo < 0 || xo >= Wo) continue;
                float v[8];
                load8(dcol + ((f * Ho + yo) * Wo + xo) * 9 * C + (dy * 3 + dx) * C + ch * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
        }
        if (u) {
            float uv[8];
            load8(u + m * C + ch * 8, uv);
            bn_affine8(uv, bnp, C, ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = (uv[j] > 0.f) ? acc[j] : 0.f;
        }
        store8(dz + m * C + ch * 8, acc);
    }
}

// conv1 input gradient (only needed when the clip itself requires grad): NCHW fp32
// dx[f][ci][y][x] = sum_{dy,dx} dcol[(f,(y-dy)/2,(x-dx)/2)][(dy*3+dx)*3+ci] over even offsets in range
template <typename T>
__global__ __launch_bounds__(256) void col2im_c3s2_kernel(const T* __restrict__ dcol, float* __restrict__ dx, long n,
                                                          int S, int Ho, int Wo) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % S), y = (int)((i / S) % S);
    const int ci = (int)((i / ((long)S * S)) % 3);
    const long f = i / ((long)3 * S * S);
    float acc = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int t = y - dy;
        if (t < 0 || (t & 1) || (t >> 1) >= Ho) continue;
#pragma unroll
        for (int dxx = 0; dxx < 3; ++dxx) {
            const int s = x - dxx;
            if (s < 0 || (s & 1) || (s >> 1) >= Wo) continue;
            acc += to_f32(dcol[((f * Ho + (t >> 1)) * Wo + (s >> 1)) * 32 + (dy * 3 + dxx) * 3 + ci]);
        }
    }
    dx[i] = acc;
}

extern "C" int istvt_col2im_conv1(const void* dcol, float* dx, int Fr, int S, int dtype, hipStream_t stream) {
    if (Fr <= 0 || S < 3) return ISTVT_ERR_SHAPE;
    const int Ho = (S - 3) / 2 + 1;
    const long n = (long)Fr * 3 * S * S;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((col2im_c3s2_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                                             stream, (const T*)dcol, dx, n, S, Ho, Ho));
    return istvt_check_launch();
}

extern "C" int istvt_im2col_conv1(const float* x, void* col, int Fr, int S, int dtype, hipStream_t stream) {
    if (Fr <= 0 || S < 3) return ISTVT_ERR_SHAPE;
    const int Ho = (S - 3) / 2 + 1;
    const long Mo = (long)Fr * Ho * Ho;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((im2col_c3s2_kernel<T>), dim3((unsigned)((Mo + 255) / 256)), dim3(256), 0,
                                             stream, x, (T*)col, Mo, S, Ho, Ho));
    return istvt_check_launch();
}

extern "C" int istvt_im2col3x3(const void* src, const float* bnp, int relu, void* col, int Fr, int H, int W, int C,
                               int dtype, hipStream_t stream) {
    if (Fr <= 0 || H < 3 || W < 3 || C % 8 != 0) return ISTVT_ERR_SHAPE;
    const int Ho = H - 2, Wo = W - 2;
    const long Mo = (long)Fr * Ho * Wo;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((im2col3x3_kernel<T>), dim3(ew_grid(Mo * 9 * (C / 8))), dim3(256), 0,
                                             stream, (const T*)src, bnp, relu, (T*)col, Mo, H, W, C, Ho, Wo));
    return istvt_check_launch();
}

extern "C" int istvt_col2im3x3(const void* dcol, const void* u, const float* bnp, void* dz, int Fr, int H, int W,
                               int C, int dtype, hipStream_t stream) {
    if (Fr <= 0 || H < 3 || W < 3 || C % 8 != 0) return ISTVT_ERR_SHAPE;
    const long Mi = (long)Fr * H * W;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((col2im3x3_kernel<T>), dim3(ew_grid(Mi * (C / 8))), dim3(256), 0, stream,
                                             (const T*)dcol, (const T*)u, bnp, (T*)dz, Mi, H, W, C, H - 2,
                                             W - 2));
    return istvt_check_launch();
}

// ============================================================================================
// depthwise 3x3, stride 1, pad 1 (SeparableConv2d.conv1, xception.py:43) with an LDS tile
// ============================================================================================
// output tile: 8 x 16 pixels x 64 channels.  64 channels = one 128-byte line per pixel (bf16): with
// 32 (half lines) every line crossed the L2->CU path twice and the kernel ran at 1.3-1.5 TB/s.
constexpr int DW_TH = 8, DW_TW = 16, DW_CC = 64;
constexpr int DW_NCH = DW_CC / 8;                           // 8-channel chunks per tile
constexpr int DW_PIXSTEP = 256 / DW_NCH;                    // pixels covered by one pass of the 256 threads
constexpr int DW_ITEMS = DW_TH * DW_TW / DW_PIXSTEP;        // (pixel, chunk) items per thread
constexpr int DW_LH = DW_TH + 2, DW_LW = DW_TW + 2;
constexpr int DW_TILE_ELEMS = DW_LH * DW_LW * DW_CC;

struct DwArgs {
    const void* in; const float* w; void* out;
    int Fr, H, W, C;
    const float* in_bn; int in_relu;                                 // BatchNorm pack (+ReLU) applied on load
    int flip;                                                        // 1: correlate with flipped taps (input gradient)
    const void* msrc; const float* m_bn;                             // ReLU mask source (+ optional BatchNorm pack)
    int mask_pre, mask_post;
    const void* addsrc; int Ha, Wa;                                  // += addsrc[f][y/2][x/2] at even (y,x); Ha == H && Wa == W: += addsrc[f][y][x]
    double* st_s1; double* st_s2;                                    // fused BN-backward statistics (of m_bn)
    unsigned long long* dbg;                                         // diagnostic builds only (-DISTVT_DW_DIAG)
};

// load the (TH+2)x(TW+2)xCC input tile (zero outside the image; the on-load transform only
// touches in-image pixels, i.e. the conv's zero padding is applied AFTER BN/ReLU as in the reference)
// TT = element type of the LDS tile: T for the forward / input-gradient kernels, float for the weight-gradient kernel
// (see there).  SWZ (float tiles only): the two 16-byte halves of a pixel's 8-channel chunk are swapped in odd LDS pixels.
// A float pixel is 256 bytes = all 64 banks, and a 16-byte access of 16 lanes covers TWO adjacent pixels x 8 chunks x
// one half: without the swap both pixels hit the same 32 banks (a 2-way conflict on every store and every tap read:
// SQ_LDS_BANK_CONFLICT was 48 % of the LDS cycles of the weight-gradient kernel), with it they split the banks.
// ReLU as ONE instruction: fmaxf(v, 0) on a value the compiler cannot prove canonical (a bf16 widened by a shift) is
// v_max v, v, v + v_max 0, v -- sixteen instead of eight per 8-channel chunk in the issue-bound tile fills.
// (inline assembly: the compiler folds fmed3(v, 0, inf) and every other spelling back into the canonicalising pair)
__device__ __forceinline__ float relu1(float v) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
}

// descriptor over ONE frame of an NHWC bf16 tensor: 32-bit offsets, lanes outside the image / past C ask for offset 2^31
// (out of range: zeros, no branch, no 64-bit address arithmetic)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dw_frame_rsrc(const bf16_t* in, long f, int H, int W, int C) {
    const unsigned long long u = (unsigned long long)(in + f * H * W * C);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, H * W * C * 2, 0x00020000u);
}

template <typename T, typename TT = T, bool SWZ = false>
__device__ __forceinline__ void dw_load_tile(TT* tile, const T* __restrict__ in, long f, int y0, int x0, int c0, int H,
                                             int W, int C, const float* bnp, int relu, int tid) {
    constexpr int NVEC = DW_LH * DW_LW * DW_NCH;
    constexpr int NIT = (NVEC + 255) / 256;
#ifdef ISTVT_DW_CACHE_DIAG
    f = 0; y0 = 0; x0 = 0;          // diagnostic: every workgroup stages the first tile (cache hits only)
#endif
    // phase 1: issue every global load of the tile before touching any result (one dependent
    // load->LDS-store round per loop iteration made the tile fill a chain of HBM latencies)
    const int ch = tid % DW_NCH, c = c0 + ch * 8;          // 256 % DW_NCH == 0: same chunk every iteration
    typename Mma<T>::frag raw[NIT];
    bool ok[NIT];
    if constexpr (sizeof(T) == 2) {
        // bf16: buffer loads through a per-frame descriptor (frames are far below 2 GiB; the host checks the tile count)
        const __amdgpu_buffer_rsrc_t rs = dw_frame_rsrc((const bf16_t*)in, f, H, W, C);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + 256 * it;
            const int px = (i / DW_NCH) % DW_LW, py = i / (DW_NCH * DW_LW);
            const int y = y0 - 1 + py, x = x0 - 1 + px;
            ok[it] = i < NVEC && y >= 0 && y < H && x >= 0 && x < W && c < C;
            const unsigned voff = ok[it] ? (unsigned)(((y * W + x) * C + c) * 2) : 0x80000000u;
            raw[it] = __builtin_bit_cast(typename Mma<T>::frag, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
        }
    } else {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + 256 * it;
            const int px = (i / DW_NCH) % DW_LW, py = i / (DW_NCH * DW_LW);
            const int y = y0 - 1 + py, x = x0 - 1 + px;
            ok[it] = i < NVEC && y >= 0 && y < H && x >= 0 && x < W && c < C;
            if (ok[it]) raw[it] = frag_load(in + ((f * H + y) * W + x) * C + c);
        }
    }
    float mu[8], sc[8], be[8];
    if (bnp && c < C) { load8(bnp + c, mu); load8(bnp + 2 * C + c, sc); load8(bnp + 3 * C + c, be); }
    // phase 2: transform (BatchNorm pack, ReLU) and store
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + 256 * it;
        if (i >= NVEC) continue;
        const int px = (i / DW_NCH) % DW_LW, py = i / (DW_NCH * DW_LW);
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ok[it]) {
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (float)raw[it][j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = raw[it].v[j];
            }
            if (bnp) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (v[j] - mu[j]) * sc[j] + be[j];
            }
            if (relu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = relu1(v[j]);
            }
        }
        if constexpr (SWZ) {
            static_assert(sizeof(TT) == 4 && DW_LW % 2 == 0, "swizzled tile: float, even row pitch");
            float* dst = reinterpret_cast<float*>(tile) + (py * DW_LW + px) * DW_CC + ch * 8;
            const int o = (px & 1) * 4;
            *reinterpret_cast<float4*>(dst + o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(dst + 4 - o) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            store8(tile + (py * DW_LW + px) * DW_CC + ch * 8, v);
        }
    }
}

// The tile fill WITHOUT a transform (no BatchNorm pack, no ReLU: every input-gradient launch -- the operand is a plain
// gradient -- and a stand-alone SeparableConv2d), bfloat16: LDS-DMA.  A wave instruction moves 64 lanes x 16 bytes =
// eight consecutive LDS pixels (a pixel's 64 channels are 128 bytes = 8 lanes); pixels outside the image, channel chunks
// past C and the pixels past the tile's end are out of range for the frame's descriptor and arrive as zeros -- the
// convolution's zero padding.  23 instructions per tile (6 per wavefront) replace the register fill's 6 loads + 6 x
// (8 conversions up, 8 down, the LDS store and its address arithmetic) per thread.  The caller waits (vmcnt(0)) before
// its barrier.  The LDS tile needs DW_DMA_PAD elements of slack (the last instruction covers pixels 176 .. 183 of 180).
constexpr int DW_DMA_INSTR = (DW_LH * DW_LW + 7) / 8;
constexpr int DW_DMA_PAD = (DW_DMA_INSTR * 8 - DW_LH * DW_LW) * DW_CC;
__device__ __forceinline__ void dw_load_tile_dma(bf16_t* tile, const bf16_t* __restrict__ in, long f, int y0, int x0, int c0,
                                                 int H, int W, int C, int tid) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned long long u = (unsigned long long)(in + f * H * W * C);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                                                        H * W * C * 2, 0x00020000u);
    const unsigned lds0 = (unsigned)(__SIZE_TYPE__)(__attribute__((address_space(3))) void*)tile;
    const int c = c0 + (lane & 7) * 8;
#pragma unroll
    for (int k = 0; k < (DW_DMA_INSTR + 3) / 4; ++k) {
        const int it = wave + 4 * k;
        if (it >= DW_DMA_INSTR) break;                       // wave-uniform
        const int i = it * 8 + (lane >> 3);
        const int py = i / DW_LW, px = i - py * DW_LW;
        const int y = y0 - 1 + py, x = x0 - 1 + px;
        const bool ok = i < DW_LH * DW_LW && y >= 0 && y < H && x >= 0 && x < W && c < C;
        dma16_lds(rs, lds0 + it * 1024, ok ? (unsigned)(((y * W + x) * C + c) * 2) : 0x80000000u, 0);
    }
}

// dw_load_tile in two halves, for a caller that keeps the NEXT tile's global loads in flight under the current tile's
// arithmetic (dwconv3x3_wgrad_kernel, round 4): fetch = every global load of the tile into registers, commit = BatchNorm
// pack / ReLU / conversion and the (swizzled, float) LDS stores.
constexpr int DW_NIT = (DW_LH * DW_LW * DW_NCH + 255) / 256;
template <typename T> struct DwTileRegs { typename Mma<T>::frag raw[DW_NIT]; unsigned ok; };
template <typename T>
__device__ __forceinline__ void dw_tile_fetch(DwTileRegs<T>& rg, const T* __restrict__ in, long f, int y0, int x0, int c0,
                                              int H, int W, int C, int tid) {
    constexpr int NVEC = DW_LH * DW_LW * DW_NCH;
    const int ch = tid % DW_NCH, c = c0 + ch * 8;
    rg.ok = 0u;
    if constexpr (sizeof(T) == 2) {
        const __amdgpu_buffer_rsrc_t rs = dw_frame_rsrc((const bf16_t*)in, f, H, W, C);      // (see dw_load_tile)
#pragma unroll
        for (int it = 0; it < DW_NIT; ++it) {
            const int i = tid + 256 * it;
            const int px = (i / DW_NCH) % DW_LW, py = i / (DW_NCH * DW_LW);
            const int y = y0 - 1 + py, x = x0 - 1 + px;
            const bool ok = i < NVEC && y >= 0 && y < H && x >= 0 && x < W && c < C;
            rg.raw[it] = __builtin_bit_cast(typename Mma<T>::frag,
                                            __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(((y * W + x) * C + c) * 2) : 0x80000000u, 0, 0));
            rg.ok |= (ok ? 1u : 0u) << it;
        }
    } else {
#pragma unroll
        for (int it = 0; it < DW_NIT; ++it) {
            const int i = tid + 256 * it;
            const int px = (i / DW_NCH) % DW_LW, py = i / (DW_NCH * DW_LW);
            const int y = y0 - 1 + py, x = x0 - 1 + px;
            const bool ok = i < NVEC && y >= 0 && y < H && x >= 0 && x < W && c < C;
            if (ok) { rg.raw[it] = frag_load(in + ((f * H + y) * W + x) * C + c); rg.ok |= 1u << it; }
        }
    }
}
template <typename T>
__device__ __forceinline__ void dw_tile_commit(float* tile, const DwTileRegs<T>& rg, bool bnp, const float (&mu)[8],
                                               const float (&sc)[8], const float (&be)[8], int relu, int tid) {
    constexpr int NVEC = DW_LH * DW_LW * DW_NCH;
    static_assert(DW_LW % 2 == 0, "swizzled tile: even row pitch");
    const int ch = tid % DW_NCH;
#pragma unroll
    for (int it = 0; it < DW_NIT; ++it) {
        const int i = tid + 256 * it;
        if (i >= NVEC) continue;
        const int px = (i / DW_NCH) % DW_LW, py = i / (DW_NCH * DW_LW);
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (rg.ok & (1u << it)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = Mma<T>::get(rg.raw[it], j);
            if (bnp) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (v[j] - mu[j]) * sc[j] + be[j];
            }
            if (relu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = relu1(v[j]);
            }
        }
        float* dst = tile + (py * DW_LW + px) * DW_CC + ch * 8;
        const int o = (px & 1) * 4;
        *reinterpret_cast<float4*>(dst + o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + 4 - o) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// Items of a thread in dwconv3x3_kernel: FOUR VERTICALLY ADJACENT pixels of one column (a 4 x 1 strip) and one 8-channel
// chunk.  The strip's 12 (item, vertical tap) pairs touch only 6 distinct input rows, so per horizontal tap a thread
// reads (and converts to float) 6 chunks instead of 12: half the LDS reads and half the bf16 -> float conversions of a
// kernel that is issue-bound (8 of the ~13 vector instructions per (item, tap) were conversions).
static_assert(DW_TH == 8 && DW_TW == 16 && DW_ITEMS == 4, "strip mapping: 2 strips of 4 rows x 16 columns");
__device__ __forceinline__ void dw_strip(int tid, int& strip, int& px) {
    const int pix0 = tid / DW_NCH;              // 0 .. 31
    strip = pix0 / DW_TW;
    px = pix0 % DW_TW;
}

// EPI = false: plain convolution (the forward launches).  The mask / skip-add / statistics epilogue of the input-
// gradient launches costs ~90 registers; compiled into one kernel it left the forward at two workgroups per CU too.
// DMA: the tile is filled by LDS-DMA (bf16 operands that need no transform: the host decides); a template parameter, so that
// neither instantiation carries the other's fill.
template <typename T, bool EPI, bool DMA = false>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(DwArgs p) {
    static_assert(!DMA || sizeof(T) == 2, "the DMA fill moves bf16");
    __shared__ __attribute__((aligned(16))) T tile[DW_TILE_ELEMS + DW_DMA_PAD];
    __shared__ float sred[2][4][DW_CC];   // [s1|s2][wave][channel]
    __shared__ __attribute__((aligned(16))) float wsm[9][DW_CC];
    __shared__ __attribute__((aligned(16))) float bnsm[EPI ? 4 : 1][DW_CC];   // the mask source's BatchNorm pack (mean, rstd, scale, beta)
    const int tid = threadIdx.x;
    const int tiles_x = (p.W + DW_TW - 1) / DW_TW, tiles_y = (p.H + DW_TH - 1) / DW_TH;
    // 1-D grid, channel chunk fastest: the workgroups that share a pixel's 128-byte lines (its other
    // channel chunks) are dispatched back to back and meet in L2 instead of re-fetching from HBM
    const int nch = (p.C + DW_CC - 1) / DW_CC;
    const int bid = xcd_chunk(blockIdx.x, gridDim.x);    // an XCD's workgroups: a contiguous run of tiles (shared halos)
    // 32-bit tile arithmetic (the host checks the tile count): as `long` the three divisions were ~130 instructions each
    const unsigned t = (unsigned)bid / (unsigned)nch;
    const int c0 = (int)((unsigned)bid - t * nch) * DW_CC;
    const unsigned tpf = (unsigned)(tiles_x * tiles_y), fu = t / tpf, tr = t - fu * tpf;
    const int ty = (int)(tr / (unsigned)tiles_x), tx = (int)(tr - ty * (unsigned)tiles_x);
    const long f = fu;
    const int y0 = ty * DW_TH, x0 = tx * DW_TW;
#ifdef ISTVT_DW_DIAG
    unsigned long long tstamp[5];
    tstamp[0] = __builtin_amdgcn_s_memtime();
#endif
    // epilogue operands (ReLU-mask source, stride-2 skip gradient) of this thread's four outputs: requested BEFORE the
    // tile is staged -- loaded per item after the barrier they were dependent HBM latencies inside the convolution
    typename Mma<T>::frag mraw[DW_ITEMS], araw[DW_ITEMS];
    // stride-1 blocks (identity or 1x1 skip, xception.py:97-100) add the skip-path gradient at every pixel
    const bool addfull = EPI && p.Ha == p.H && p.Wa == p.W;
    const int ashift = addfull ? 0 : 1;
    if constexpr (EPI) {
        const int cq = c0 + (tid % DW_NCH) * 8;
        int strip_, px_;
        dw_strip(tid, strip_, px_);
#pragma unroll
        for (int k = 0; k < DW_ITEMS; ++k) {
            const int y = y0 + strip_ * DW_ITEMS + k, x = x0 + px_;
            mraw[k] = Mma<T>::zero(); araw[k] = Mma<T>::zero();
            if (y < p.H && x < p.W && cq < p.C) {
                if (p.msrc) mraw[k] = frag_load((const T*)p.msrc + ((f * p.H + y) * p.W + x) * p.C + cq);
                if (p.addsrc && (addfull || (!(y & 1) && !(x & 1))))
                    araw[k] = frag_load((const T*)p.addsrc + ((f * p.Ha + (y >> ashift)) * p.Wa + (x >> ashift)) * p.C + cq);
            }
        }
    }
    constexpr bool dma_fill = DMA;
    if constexpr (DMA) dw_load_tile_dma((bf16_t*)tile, (const bf16_t*)p.in, f, y0, x0, c0, p.H, p.W, p.C, tid);
    else dw_load_tile<T>(tile, (const T*)p.in, f, y0, x0, c0, p.H, p.W, p.C, p.in_bn, p.in_relu, tid);
#ifdef ISTVT_DW_DIAG
    tstamp[1] = __builtin_amdgcn_s_memtime();
#endif

    const int ch = tid % DW_NCH;                  // 8-channel chunk of this thread (same for all its items)
    const int c = c0 + ch * 8;
    // weights are passed tap-major ([9][C]).  They live in LDS, not in registers: 72 weight registers kept the kernel at
    // 186 VGPRs = two workgroups per CU, and its phases (tile fetch 5.6 k cycles, convolution 5.3 k) only overlap
    // ACROSS workgroups (tools/dw_diag.py); the two extra LDS reads per tap are cheaper than that.
    if (tid < 9 * DW_NCH) {
        const int tap = tid / DW_NCH, wc = (tid % DW_NCH) * 8;
        const int src_tap = p.flip ? 8 - tap : tap;
        float w8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (c0 + wc < p.C) load8(p.w + (long)src_tap * p.C + c0 + wc, w8);
        store8(&wsm[tap][wc], w8);
    }
    if constexpr (EPI) {
        // per-channel constants of the epilogue, staged like the weights: read from global memory per item they were
        // 160 bytes of L1 traffic per item beside 48 bytes of payload (mask source, skip gradient, store)
        if (p.m_bn && tid >= 128 && tid < 128 + 4 * DW_NCH) {
            const int row = (tid - 128) / DW_NCH, wc = ((tid - 128) % DW_NCH) * 8;
            float b8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c0 + wc < p.C) load8(p.m_bn + (long)row * p.C + c0 + wc, b8);
            store8(&bnsm[row][wc], b8);
        }
    }
    if (dma_fill) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's DMA pieces have landed
    __syncthreads();
#ifdef ISTVT_DW_DIAG
    tstamp[2] = __builtin_amdgcn_s_memtime();
#endif

    float st1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // Taps outside, this thread's items inside: a tap's weights are read from LDS once for the four items (the weights
    // were two thirds of the kernel's LDS reads: 27 x the output bytes), the accumulators of all items stay live.
    float accs[DW_ITEMS][8];
#pragma unroll
    for (int k = 0; k < DW_ITEMS; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) accs[k][j] = 0.f;
    {
        int strip, px0;
        dw_strip(tid, strip, px0);
        const T* tbase = tile + ((strip * DW_ITEMS) * DW_LW + px0) * DW_CC + ch * 8;
        const float* wbase = &wsm[0][ch * 8];
        if constexpr (sizeof(T) == 4) {
            // float32 parity mode keeps the tap-major order of every item's nine products (dy outer, dx inner): it is the
            // order of the reference's CPU convolution, and with the goldens' structured weights another order flips ReLU /
            // arg-max decisions at |z| ~ 1e-7 (stem_oracle_139_f32 holds 2e-4 only in this order)
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - 3 * dy;
                float w8[8];
                load8(wbase + tap * DW_CC, w8);
#pragma unroll
                for (int k = 0; k < DW_ITEMS; ++k) {
                    float v[8];
                    load8(tbase + ((k + dy) * DW_LW + dx) * DW_CC, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) accs[k][j] += v[j] * w8[j];
                }
            }
        } else {
#pragma unroll 1
            for (int dx = 0; dx < 3; ++dx) {                // a real loop: nothing of column tap dx + 1 is live during dx
                float w8[3][8];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) load8(wbase + (dy * 3 + dx) * DW_CC, w8[dy]);
#pragma unroll
                for (int r = 0; r < DW_ITEMS + 2; ++r) {    // input row r of the strip feeds item k through tap dy = r - k
                    float v[8];
                    load8(tbase + (r * DW_LW + dx) * DW_CC, v);
#pragma unroll
                    for (int k = 0; k < DW_ITEMS; ++k) {
                        const int dy = r - k;
                        if (dy < 0 || dy > 2) continue;
#pragma unroll
                        for (int j = 0; j < 8; ++j) accs[k][j] += v[j] * w8[dy][j];
                    }
                }
            }
        }
    }
    auto item = [&](const int k) {
        int strip, px;
        dw_strip(tid, strip, px);
        const int y = y0 + strip * DW_ITEMS + k, x = x0 + px;
        if (y >= p.H || x >= p.W || c >= p.C) return;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = accs[k][j];
        int wo = ch * 8;
        asm volatile("" : "+v"(wo));              // opaque per item: the constants must not be hoisted across items
        const long off = ((f * p.H + y) * p.W + x) * p.C + c;
        if constexpr (EPI) {
        // epilogue order: ReLU mask of the rep path (pre) -> add the skip-path gradient at the
        // stride-2 positions -> ReLU mask that covers both paths (post)
        float mv[8], z[8];
        const bool have_m = p.msrc != nullptr;
        if (have_m) {
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) mv[j] = (float)mraw[k][j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) mv[j] = mraw[k].v[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = mv[j];
            if (p.m_bn) {
                float mu[8], sc[8], be[8];
                load8(&bnsm[0][wo], mu); load8(&bnsm[2][wo], sc); load8(&bnsm[3][wo], be);
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = (z[j] - mu[j]) * sc[j] + be[j];
            }
        }
        if (p.mask_pre) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = z[j] > 0.f ? acc[j] : 0.f;
        }
        if (p.addsrc && (addfull || (!(y & 1) && !(x & 1)))) {
            float a[8];
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = (float)araw[k][j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = araw[k].v[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += a[j];
        }
        if (p.mask_post) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = z[j] > 0.f ? acc[j] : 0.f;
        }
        if (p.st_s1 && have_m) {
            // statistics of the (rounded) value that is stored, so they match a separate pass
            float mu[8], rs[8];
            load8(&bnsm[0][wo], mu);
            load8(&bnsm[1][wo], rs);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = to_f32(from_f32<T>(acc[j]));
                st1[j] += d;
                st2[j] += d * (mv[j] - mu[j]) * rs[j];
            }
        }
        }
        store8((T*)p.out + off, acc);
    };
    if constexpr (EPI) {        // unrolled by hand: the prefetched fragments are indexed by k
        static_assert(DW_ITEMS == 4, "item(0..3)");
        item(0); item(1); item(2); item(3);
    } else {
#pragma unroll
        for (int k = 0; k < DW_ITEMS; ++k) item(k);
    }
#ifdef ISTVT_DW_DIAG
    tstamp[3] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tstamp[4] = __builtin_amdgcn_s_memtime();
    if (p.dbg && (blockIdx.x % 997) == 0 && blockIdx.x / 997 < 32 && (tid & 63) == 0) {
        unsigned long long* d = p.dbg + ((blockIdx.x / 997) * 4 + (tid >> 6)) * 5;
        for (int i = 0; i < 5; ++i) d[i] = tstamp[i];
    }
#endif
    if (EPI && p.st_s1) {
        // reduce over the pixels of a wave that share this channel chunk (lanes with equal tid % DW_NCH)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int o = DW_NCH; o < 64; o <<= 1) {
                st1[j] += __shfl_xor(st1[j], o, 64);
                st2[j] += __shfl_xor(st2[j], o, 64);
            }
        }
        const int lane = tid & 63, wid = tid >> 6;
        if (lane < DW_NCH) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { sred[0][wid][lane * 8 + j] = st1[j]; sred[1][wid][lane * 8 + j] = st2[j]; }
        }
        __syncthreads();
        if (tid < DW_CC && c0 + tid < p.C) {
            const long rep = (long)(blockIdx.x % STAT_REPLICAS) * 2 * p.C;
            atomicAdd(p.st_s1 + rep + c0 + tid, (double)(sred[0][0][tid] + sred[0][1][tid] + sred[0][2][tid] + sred[0][3][tid]));
            atomicAdd(p.st_s2 + rep + c0 + tid, (double)(sred[1][0][tid] + sred[1][1][tid] + sred[1][2][tid] + sred[1][3][tid]));
        }
    }
}

// dw[c][tap] += sum_pix dout[pix][c] * a[pix + tap][c],  a = on-load transform of the forward input
#ifdef ISTVT_DW_STAMP
// diagnostic build (-DISTVT_DW_STAMP): per-phase s_memtime sums of the depthwise weight-gradient kernel per wavefront,
// [workgroup][wavefront][8] u64 = {top barrier, issue loads, tile landed + transformed + stored, barrier, FMAs, tiles, -, -}
__device__ unsigned long long* g_dw_stamps = nullptr;
extern "C" int istvt_diag_dw_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_dw_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -4;
}
#define DW_STAMP(i)                                                                                   \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if ((i) > 0) dws_seg[(i) - 1] += (unsigned)(t_ - dws_prev);                                   \
        dws_prev = t_;                                                                                \
    } while (0)
#else
#define DW_STAMP(i) do { } while (0)
#endif

template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const T* __restrict__ in, const float* in_bn,
                                                              int in_relu,
                                                              const T* __restrict__ dout, float* __restrict__ dw,
                                                              int Fr, int H, int W, int C) {
    // The LDS tile of THIS kernel holds floats whatever the storage dtype: every input element is used by nine taps, and
    // with a bf16 tile each use converted it again -- in a kernel that is issue-bound at two wavefronts per SIMD
    // (SQ_ACTIVE_INST_ANY 0.47 of the wave cycles).  Measured at C2 (tools/stem_bench.py, same box): 280 -> 251, 476 ->
    // 429, 278 -> 253, 233 -> 218 us for the four shapes.  The forward / input-gradient kernels keep a bf16 tile: a float
    // tile takes them from 5 / 4 to 3 workgroups per CU and they lose 10-25 %.
    __shared__ __attribute__((aligned(16))) float tile[DW_TILE_ELEMS];
    __shared__ float sred[4][DW_NCH][72];
    const int tid = threadIdx.x;
    const int tiles_x = (W + DW_TW - 1) / DW_TW, tiles_y = (H + DW_TH - 1) / DW_TH;
    // 32-bit tile arithmetic (the host checks the tile count): as `long`, five divisions per tile were ~650 of the tile
    // loop's ~1800 instructions
    const unsigned tpf = (unsigned)(tiles_x * tiles_y), ntiles = (unsigned)Fr * tpf;
    const int nch = (C + DW_CC - 1) / DW_CC;                 // 1-D grid, channel chunk fastest (L2 sharing)
    const int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const unsigned slot = (unsigned)bid / (unsigned)nch;
    const int c0 = (int)((unsigned)bid - slot * nch) * DW_CC;
    const int ch = tid % DW_NCH, c = c0 + ch * 8;
    const int o0 = ((tid / DW_NCH) & 1) * 4;         // parity of this thread's pixel column (DW_PIXSTEP and DW_TW are even)
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
#ifdef ISTVT_DW_STAMP
    unsigned dws_seg[5] = {0, 0, 0, 0, 0}, dws_n = 0;
    unsigned long long dws_prev = 0;
#endif
    // Round 4 (stamps: tools/dw_stamps.py, profiles/r04_e_dwconv_wgrad_stamps.txt).  A tile cost a wavefront ~7.4 k
    // cycles: 4.2 k from issuing the tile's global loads to its LDS stores (HBM round trip + a per-tile re-read of the
    // BatchNorm pack, each waited for; nothing overlapped inside a workgroup and two workgroups per CU), 2.3 k of FMAs.
    // Three changes, measured one at a time on one box (H=109 C=64 / H=109 C=128 / H=55 C=256 / H=28 C=728, us):
    //   next tile's loads (input tile AND output-gradient rows) requested BEFORE the current tile's FMAs, waiting in
    //   registers (+40 VGPRs) -- alone: 224 -> 231 (the round trip moved, the BatchNorm re-read stayed exposed);
    //   BatchNorm pack read once per workgroup (its channels never change):           226 / 434 / 245 / 207 -> 222 / 420 / 236 / 196;
    //   vertical strips (18 LDS positions per thread and tile instead of 36):         -> 208 / 404 / 227 / 190.
    // (Strips alone, without the prefetch, had measured neutral: the exposed round trip hid the LDS reads.)  A tile now
    // costs ~6.4 k cycles: 2.0 k transform + LDS stores, 4.1 k next-tile load issue + FMAs + the end-of-tile wait.
    const unsigned tstep = gridDim.x / (unsigned)nch;
    int wstrip, wpx;
    dw_strip(tid, wstrip, wpx);
    // the BatchNorm pack of the workgroup's channels: the same for every tile, read once (it was re-read, and waited for,
    // per tile)
    float bn_mu[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bn_sc[8] = {1, 1, 1, 1, 1, 1, 1, 1}, bn_be[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (in_bn && c < C) { load8(in_bn + c, bn_mu); load8(in_bn + 2 * C + c, bn_sc); load8(in_bn + 3 * C + c, bn_be); }
    DwTileRegs<T> nxt;
    typename Mma<T>::frag ndraw[DW_ITEMS];
    unsigned nlive = 0;                                      // bit k: item k of the fetched tile is inside the image
    auto fetch = [&](unsigned t) {
        const unsigned fu = t / tpf, tr = t - fu * tpf;
        const int ty = (int)(tr / (unsigned)tiles_x), tx = (int)(tr - ty * (unsigned)tiles_x);
        const long f = fu;
        const int y0 = ty * DW_TH, x0 = tx * DW_TW;
        nlive = 0;
#pragma unroll
        for (int k = 0; k < DW_ITEMS; ++k) {
            const int y = y0 + wstrip * DW_ITEMS + k, x = x0 + wpx;
            if (y < H && x < W && c < C) { ndraw[k] = frag_load(dout + ((f * H + y) * W + x) * C + c); nlive |= 1u << k; }
        }
#ifdef ISTVT_DW_CACHE_DIAG
        dw_tile_fetch<T>(nxt, in, 0, 0, 0, c0, H, W, C, tid);
#else
        dw_tile_fetch<T>(nxt, in, f, y0, x0, c0, H, W, C, tid);
#endif
    };
    if (slot < ntiles) fetch(slot);
    for (unsigned t = slot; t < ntiles; t += tstep) {
        DW_STAMP(0);
        __syncthreads();                                   // the previous tile's FMAs are done reading the LDS tile
        DW_STAMP(1);
        typename Mma<T>::frag draw[DW_ITEMS];
#pragma unroll
        for (int k = 0; k < DW_ITEMS; ++k) draw[k] = ndraw[k];
        const unsigned dlive = nlive;
        dw_tile_commit<T>(tile, nxt, in_bn != nullptr, bn_mu, bn_sc, bn_be, in_relu, tid);
        DW_STAMP(2);
        __syncthreads();
        DW_STAMP(3);
        if (t + tstep < ntiles) fetch(t + tstep);          // in flight under the FMAs below (t + tstep < 2^32: host check)
        DW_STAMP(4);
        {
            // vertical strips (dw_strip, as in dwconv3x3_kernel): a thread's four items are four consecutive rows of one
            // column, so the 12 (item, vertical tap) pairs of a horizontal tap share 6 input rows -- 18 LDS positions per
            // thread and tile instead of 36
            float d[DW_ITEMS][8];
#pragma unroll
            for (int k = 0; k < DW_ITEMS; ++k) {
                const bool live = (dlive >> k) & 1u;
#pragma unroll
                for (int j = 0; j < 8; ++j) d[k][j] = live ? Mma<T>::get(draw[k], j) : 0.f;
            }
            const float* tbase = tile + ((wstrip * DW_ITEMS) * DW_LW + wpx) * DW_CC + ch * 8;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int o = (dx & 1) ? 4 - o0 : o0;       // swizzled tile: halves swapped in odd LDS pixels
#pragma unroll
                for (int r = 0; r < DW_ITEMS + 2; ++r) {
                    const float* tp = tbase + (r * DW_LW + dx) * DW_CC;
                    const float4 lo = *reinterpret_cast<const float4*>(tp + o);
                    const float4 hi = *reinterpret_cast<const float4*>(tp + 4 - o);
                    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                    for (int k = 0; k < DW_ITEMS; ++k) {
                        const int dy = r - k;
                        if (dy < 0 || dy > 2) continue;
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[dy * 3 + dx][j] += d[k][j] * v[j];
                    }
                }
            }
        }
        DW_STAMP(5);
#ifdef ISTVT_DW_STAMP
        ++dws_n;
#endif
    }
#ifdef ISTVT_DW_STAMP
    if ((tid & 63) == 0 && g_dw_stamps && blockIdx.x < 2048) {
        unsigned long long* dd = g_dw_stamps + ((long)blockIdx.x * 4 + (tid >> 6)) * 8;
        for (int j = 0; j < 5; ++j) dd[j] = dws_seg[j];
        dd[5] = dws_n;
    }
#endif
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int o = DW_NCH; o < 64; o <<= 1) acc[t][j] += __shfl_xor(acc[t][j], o, 64);
        }
    const int lane = tid & 63, wid = tid >> 6;
    if (lane < DW_NCH) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) sred[wid][lane][t * 8 + j] = acc[t][j];
    }
    __syncthreads();
    for (int o = tid; o < DW_NCH * 72; o += 256) {
        const int chunk = o / 72, rem = o % 72, tap = rem / 8, j = rem % 8;
        const int cc = c0 + chunk * 8 + j;
        // one partial [C][9] slab per slot (the workgroups of a slot cover the channel chunks): folded in slot order by
        // istvt_rows_reduce_add -- no float atomics
        if (cc < C)
            dw[(long)slot * C * 9 + (long)cc * 9 + tap] = (sred[0][chunk][rem] + sred[1][chunk][rem]) + (sred[2][chunk][rem] + sred[3][chunk][rem]);
    }
}

extern "C" int istvt_dwconv3x3(const void* in, const float* w, void* out, int Fr, int H, int W, int C,
                               const float* in_bn, int in_relu, int flip, const void* msrc, const float* m_bn,
                               int mask_pre, int mask_post, const void* addsrc, int Ha, int Wa, double* st_s1,
                               double* st_s2, int dtype, hipStream_t stream) {
    if (Fr <= 0 || H <= 0 || W <= 0 || C % 8 != 0) return ISTVT_ERR_SHAPE;
    if ((long)H * W * C * 4 >= 0x7fffffffL) return ISTVT_ERR_SHAPE;           // one frame behind a 32-bit buffer descriptor
    if ((mask_pre || mask_post || st_s1) && !msrc) return ISTVT_ERR_SHAPE;
    if (st_s1 && !m_bn) return ISTVT_ERR_SHAPE;
    if (addsrc && !(Ha == H && Wa == W) && (Ha != (H - 1) / 2 + 1 || Wa != (W - 1) / 2 + 1)) return ISTVT_ERR_SHAPE;
    DwArgs a;
    a.in = in; a.w = w; a.out = out; a.Fr = Fr; a.H = H; a.W = W; a.C = C;
    a.in_bn = in_bn; a.in_relu = in_relu; a.flip = flip;
    a.msrc = msrc; a.m_bn = m_bn; a.mask_pre = mask_pre; a.mask_post = mask_post;
    a.addsrc = addsrc; a.Ha = Ha; a.Wa = Wa; a.st_s1 = st_s1; a.st_s2 = st_s2;
    a.dbg = nullptr;
#ifdef ISTVT_DW_DIAG
    if (getenv("ISTVT_DW_DBGPTR")) a.dbg = (unsigned long long*)strtoull(getenv("ISTVT_DW_DBGPTR"), nullptr, 0);
#endif
    const long tiles = (long)Fr * ((H + DW_TH - 1) / DW_TH) * ((W + DW_TW - 1) / DW_TW);
    const long nblk = tiles * ((C + DW_CC - 1) / DW_CC);
    if (nblk > 0x7fffffffL) return ISTVT_ERR_SHAPE;
    dim3 grid((unsigned)nblk);
    const bool epi = msrc || addsrc || st_s1;
#ifndef ISTVT_DW_NO_DMA_FILL
    if (dtype == DT_BF16 && !in_bn && !in_relu) {        // no transform on load: the tile arrives by LDS-DMA
        if (epi) hipLaunchKernelGGL((dwconv3x3_kernel<bf16_t, true, true>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((dwconv3x3_kernel<bf16_t, false, true>), grid, dim3(256), 0, stream, a);
        return istvt_check_launch();
    }
#endif
    if (epi) DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dwconv3x3_kernel<T, true>), grid, dim3(256), 0, stream, a));
    else DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dwconv3x3_kernel<T, false>), grid, dim3(256), 0, stream, a));
    return istvt_check_launch();
}

// slots (workgroups per channel chunk) of the weight-gradient launch: each stores one partial [C][9] slab
static long dww_slots(int Fr, int H, int W, int C) {
    const long tiles = (long)Fr * ((H + DW_TH - 1) / DW_TH) * ((W + DW_TW - 1) / DW_TW);
    const int cy = (C + DW_CC - 1) / DW_CC;
    static const long wg_cap = istvt_tune("ISTVT_DWW_BLOCKS", 512);   // = resident workgroups
    long bx = wg_cap / cy;
    if (bx < 1) bx = 1;
    return bx > tiles ? tiles : bx;
}
extern "C" int istvt_dwconv3x3_wgrad_ws_elems(int Fr, int H, int W, int C) {
    if (Fr <= 0 || H <= 0 || W <= 0 || C % 8 != 0) return ISTVT_ERR_SHAPE;
    return (int)(dww_slots(Fr, H, W, C) * C * 9);
}

// dw [C][9] accumulates (+=); ws: float scratch of istvt_dwconv3x3_wgrad_ws_elems(...) elements
extern "C" int istvt_dwconv3x3_wgrad(const void* in, const float* in_bn, int in_relu, const void* dout, float* dw,
                                     float* ws, long ws_elems, int Fr, int H, int W, int C, int dtype, hipStream_t stream) {
    if (Fr <= 0 || H <= 0 || W <= 0 || C % 8 != 0 || !ws) return ISTVT_ERR_SHAPE;
    if ((long)H * W * C * 4 >= 0x7fffffffL) return ISTVT_ERR_SHAPE;           // one frame behind a 32-bit buffer descriptor
    const int cy = (C + DW_CC - 1) / DW_CC;
    const long bx = dww_slots(Fr, H, W, C);
    if (ws_elems < bx * C * 9) return ISTVT_ERR_SHAPE;
    if ((long)Fr * ((H + DW_TH - 1) / DW_TH) * ((W + DW_TW - 1) / DW_TW) + bx > 0x7fffffffL) return ISTVT_ERR_SHAPE;   // 32-bit tile index
    dim3 grid((unsigned)(bx * cy));                        // slot-major, channel chunk fastest
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<T>), grid, dim3(256), 0, stream, (const T*)in,
                                             in_bn, in_relu, (const T*)dout, ws, Fr, H, W, C));
    int rc = istvt_check_launch();
    if (rc) return rc;
    return istvt_rows_reduce_add(ws, (int)bx, 1, C * 9, dw, nullptr, nullptr, stream);
}

// ============================================================================================
// MaxPool2d(3, 2, 1) fused with both BatchNorm applies and the skip add (Block.forward,
// xception.py:88,91-100):   out = maxpool(bn(x)) + bn_skip(skip)
// ============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void pool_add_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bnx,
                                                           const T* __restrict__ skip, const float* __restrict__ bns,
                                                           T* __restrict__ out, uint8_t* __restrict__ argmax, long Mo,
                                                           int H, int W, int C, int Ho, int Wo) {
    // Item -> (frame, row, column, chunk) in 32-bit unsigned arithmetic (the launcher refuses 2^31 items or more): as
    // `long` the four divisions were the compiler's 64-bit expansion with a run-time test for the 32-bit case, per item,
    // in a kernel whose issue port is as busy as its memory path.
    const unsigned vpr = C / 8;
    const unsigned nitems = (unsigned)Mo * vpr, stride = gridDim.x * 256u;
    for (unsigned i = (unsigned)xcd_chunk(blockIdx.x, gridDim.x) * 256u + threadIdx.x; i < nitems; i += stride) {   // windows of neighbouring rows overlap: keep them in one XCD's L2
        const unsigned mu_ = i / vpr, ru_ = mu_ / (unsigned)Wo, fu_ = ru_ / (unsigned)Ho;
        const int ch = (int)(i - mu_ * vpr);
        const long m = mu_;
        const int xo = (int)(mu_ - ru_ * (unsigned)Wo), yo = (int)(ru_ - fu_ * (unsigned)Ho);
        const long f = fu_;
        // All nine taps (clamped addresses) and the skip row are requested before any is used: with each load inside its
        // bounds test the taps were nine dependent global-memory latencies per output (2.2 TB/s of algorithmic traffic).
        typename Mma<T>::frag raw[9];
        bool ok[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int y = 2 * yo - 1 + dy;
            const int yc = min(max(y, 0), H - 1);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = 2 * xo - 1 + dx;
                const int xc = min(max(xx, 0), W - 1);
                ok[dy * 3 + dx] = y >= 0 && y < H && xx >= 0 && xx < W;
                raw[dy * 3 + dx] = frag_load(x + ((f * H + yc) * W + xc) * C + ch * 8);
            }
        }
        const typename Mma<T>::frag sraw = frag_load(skip + m * C + ch * 8);
        float mu[8], sc[8], be[8];
        load8(bnx + ch * 8, mu); load8(bnx + 2 * C + ch * 8, sc); load8(bnx + 3 * C + ch * 8, be);
        float best[8];
        int bi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bi[j] = 0; }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (!ok[t]) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float z = to_f32(from_f32<T>((Mma<T>::get(raw[t], j) - mu[j]) * sc[j] + be[j]));   // the value a separate BN pass would store
                if (z > best[j]) { best[j] = z; bi[j] = t; }                                  // first maximum wins (torch)
            }
        }
        float sv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[j] = Mma<T>::get(sraw, j);
        bn_affine8(sv, bns, C, ch * 8);
        uint64_t packed = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            best[j] += sv[j];
            packed |= (uint64_t)bi[j] << (8 * j);
        }
        store8(out + m * C + ch * 8, best);
        *reinterpret_cast<uint64_t*>(argmax + m * C + ch * 8) = packed;
    }
}

// dz[f,y,x,c] = sum over the (<= 4) windows containing (y,x) of dout[window] * [argmax(window) == (y,x)]
// One thread = one 8-channel chunk of a 2x2 pixel quad {2a-1, 2a} x {2b-1, 2b}: the quad's pixels lie in the windows
// (a-1|a, b-1|b) only, so four window loads (gradient row + argmax bytes) serve four outputs.  One thread per PIXEL
// read its <= 4 windows itself: 96 bytes loaded per 16 written, and the kernel ran at the rate of the load path
// (2.3 TB/s of algorithmic traffic) although its HBM traffic was the algorithmic 645 MB per launch.
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const T* __restrict__ dout, const uint8_t* __restrict__ argmax,
                                                       T* __restrict__ dz, long nquads, int H, int W, int C, int Ho,
                                                       int Wo, const T* __restrict__ u, const float* __restrict__ bnp,
                                                       double* st_s1, double* st_s2) {
    // u != null: dz is the output gradient of the BatchNorm whose input was u (the block's last BatchNorm, xception.py:
    // 75 -> 88): its backward sums  s1 = sum dz,  s2 = sum dz * xhat  are taken here, of the rounded values that are
    // stored, instead of by a pass over dz and u (istvt_bn_bwd_stats).  A thread keeps one channel chunk for all its
    // quads (the thread count in use is a multiple of the chunks per pixel): constants and partial sums in registers.
    __shared__ float part[2][256][8];                   // the threads' partial sums (u != null)
    const int vpr = C / 8;
    const int QH = H / 2 + 1, QW = W / 2 + 1;
    const long nthr = ((long)gridDim.x * 256) / vpr * vpr;
    const long t = (long)xcd_chunk(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const bool stats = u != nullptr;
    const int ch = (int)(t % vpr);
    float mu[8], rs[8], a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu[j] = 0.f; rs[j] = 0.f; a1[j] = 0.f; a2[j] = 0.f; }
    if (stats && t < nthr) { load8(bnp + ch * 8, mu); load8(bnp + C + ch * 8, rs); }
    // quad -> (frame, a, b) in 32-bit unsigned arithmetic (the launcher refuses 2^31 quads or more), see pool_add_fwd_kernel
    const unsigned qstep = (unsigned)(nthr / vpr), nq = (unsigned)nquads;
    for (unsigned q = t < nthr ? (unsigned)(t / vpr) : nq; q < nq; q += qstep) {
        const unsigned ru_ = q / (unsigned)QW, fu_ = ru_ / (unsigned)QH;
        const int b = (int)(q - ru_ * (unsigned)QW), a = (int)(ru_ - fu_ * (unsigned)QH);
        const long f = fu_;
        // windows (a-1+wa, b-1+wb), wa, wb in {0, 1}
        typename Mma<T>::frag draw[4], uraw[4];
        unsigned lo[4], hi[4];
        bool okw[4];
#pragma unroll
        for (int wa = 0; wa < 2; ++wa)
#pragma unroll
            for (int wb = 0; wb < 2; ++wb) {
                const int yo = a - 1 + wa, xo = b - 1 + wb;
                okw[wa * 2 + wb] = yo >= 0 && yo < Ho && xo >= 0 && xo < Wo;
                const long mo = (f * Ho + min(max(yo, 0), Ho - 1)) * Wo + min(max(xo, 0), Wo - 1);
                const uint64_t pk = *reinterpret_cast<const uint64_t*>(argmax + mo * C + ch * 8);
                lo[wa * 2 + wb] = (unsigned)pk; hi[wa * 2 + wb] = (unsigned)(pk >> 32);
                draw[wa * 2 + wb] = frag_load(dout + mo * C + ch * 8);
            }
        if (stats) {
#pragma unroll
            for (int pa = 0; pa < 2; ++pa)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    const int y = min(max(2 * a - 1 + pa, 0), H - 1), x = min(max(2 * b - 1 + pb, 0), W - 1);
                    uraw[pa * 2 + pb] = frag_load(u + ((f * H + y) * W + x) * C + ch * 8);
                }
        }
        // pixel (2a-1+pa, 2b-1+pb): position dy = 2 - 2 wa + pa inside window row a-1+wa, valid for (pa, wa) in
        // {(0,0): 2, (0,1): 0, (1,1): 1}; the same for dx
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {
            const int y = 2 * a - 1 + pa;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const int x = 2 * b - 1 + pb;
                if (y < 0 || y >= H || x < 0 || x >= W) continue;
                float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int wa = pa; wa < 2; ++wa)              // pa = 1 (even row) lies in window row a only
#pragma unroll
                    for (int wb = pb; wb < 2; ++wb) {
                        const int w = wa * 2 + wb;
                        const unsigned want = okw[w] ? (unsigned)((2 - 2 * wa + pa) * 3 + (2 - 2 * wb + pb)) : 255u;   // 255 never matches (codes 0..8)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const unsigned code = __builtin_amdgcn_ubfe(j < 4 ? lo[w] : hi[w], 8 * (j & 3), 8);
                            acc[j] += code == want ? Mma<T>::get(draw[w], j) : 0.f;
                        }
                    }
                store8(dz + ((f * H + y) * W + x) * C + ch * 8, acc);
                if (stats) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float d = to_f32(from_f32<T>(acc[j]));
                        a1[j] += d;
                        a2[j] += d * (Mma<T>::get(uraw[pa * 2 + pb], j) - mu[j]) * rs[j];
                    }
                }
            }
        }
    }
    if (stats) {
        // the threads of this workgroup that hold channel chunk k are tid = k0, k0 + vpr, ...: summed in that fixed order
        // (fp32 atomics into LDS would make the sums -- and through them every gradient upstream -- depend on timing)
#pragma unroll
        for (int j = 0; j < 8; ++j) { part[0][threadIdx.x][j] = a1[j]; part[1][threadIdx.x][j] = a2[j]; }
        __syncthreads();
        const long t0 = t - threadIdx.x;                          // this workgroup's first thread index
        const long rep = (long)(blockIdx.x % STAT_REPLICAS) * 2 * C;
        for (int c = threadIdx.x; c < C; c += 256) {
            const int k = c >> 3, j = c & 7;
            const int first = (int)(((long)k - t0 % vpr + vpr) % vpr);
            float s1 = 0.f, s2 = 0.f;
            for (int tt = first; tt < 256 && t0 + tt < nthr; tt += vpr) { s1 += part[0][tt][j]; s2 += part[1][tt][j]; }
            atomicAdd(st_s1 + rep + c, (double)s1);
            atomicAdd(st_s2 + rep + c, (double)s2);
        }
    }
}

// out[f,yo,xo,:] = in[f,2yo,2xo,:]   (input of the stride-2 1x1 skip conv, xception.py:57)
template <typename T>
__global__ __launch_bounds__(256) void subsample2_kernel(const T* __restrict__ in, T* __restrict__ out, long Mo, int H,
                                                         int W, int C, int Ho, int Wo) {
    const unsigned vpr = C / 8;
    const unsigned nitems = (unsigned)Mo * vpr, stride = gridDim.x * 256u;      // 32-bit: the launcher checks the count
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < nitems; i += stride) {
        const unsigned mu_ = i / vpr, ru_ = mu_ / (unsigned)Wo, fu_ = ru_ / (unsigned)Ho;
        const int ch = (int)(i - mu_ * vpr);
        const long m = mu_;
        const int xo = (int)(mu_ - ru_ * (unsigned)Wo), yo = (int)(ru_ - fu_ * (unsigned)Ho);
        const long f = fu_;
        float v[8];
        load8(in + ((f * H + 2 * yo) * W + 2 * xo) * C + ch * 8, v);
        store8(out + m * C + ch * 8, v);
    }
}

extern "C" int istvt_pool_add_fwd(const void* x, const float* bnx, const void* skip, const float* bns, void* out,
                                  uint8_t* argmax, int Fr, int H, int W, int C, int dtype, hipStream_t stream) {
    if (Fr <= 0 || H <= 0 || W <= 0 || C % 8 != 0) return ISTVT_ERR_SHAPE;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long Mo = (long)Fr * Ho * Wo;
    if (Mo * (C / 8) + 65536L * 256 > 0x7fffffffL) return ISTVT_ERR_SHAPE;      // 32-bit item index (+ one grid stride)
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pool_add_fwd_kernel<T>), dim3(ew_grid(Mo * (C / 8))), dim3(256), 0,
                                             stream, (const T*)x, bnx, (const T*)skip, bns, (T*)out, argmax, Mo,
                                             H, W, C, Ho, Wo));
    return istvt_check_launch();
}

extern "C" int istvt_pool_bwd(const void* dout, const uint8_t* argmax, void* dz, int Fr, int H, int W, int C,
                              const void* u, const float* bnp, double* st_s1, double* st_s2, int dtype,
                              hipStream_t stream) {
    if (Fr <= 0 || H <= 0 || W <= 0 || C % 8 != 0) return ISTVT_ERR_SHAPE;
    if (u && (!bnp || !st_s1 || !st_s2)) return ISTVT_ERR_SHAPE;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long nquads = (long)Fr * (H / 2 + 1) * (W / 2 + 1);      // 2x2 pixel quads, see the kernel
    // with the statistics every workgroup ends in 2 C fp64 atomics: a grid of resident size
    static const long cap = istvt_tune("ISTVT_POOLB_BLOCKS", 2048);
    long blocks = (nquads * (C / 8) + 255) / 256;
    if (blocks > cap) blocks = cap;
    if (nquads + blocks * 256 > 0x7fffffffL) return ISTVT_ERR_SHAPE;            // 32-bit quad index (+ one grid stride)
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pool_bwd_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, stream,
                                             (const T*)dout, argmax, (T*)dz, nquads, H, W, C, Ho, Wo, (const T*)u, bnp,
                                             st_s1, st_s2));
    return istvt_check_launch();
}

extern "C" int istvt_subsample2(const void* in, void* out, int Fr, int H, int W, int C, int dtype, hipStream_t stream) {
    if (Fr <= 0 || H <= 0 || W <= 0 || C % 8 != 0) return ISTVT_ERR_SHAPE;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long Mo = (long)Fr * Ho * Wo;
    if (Mo * (C / 8) + 65536L * 256 > 0x7fffffffL) return ISTVT_ERR_SHAPE;      // 32-bit item index (+ one grid stride)
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((subsample2_kernel<T>), dim3(ew_grid(Mo * (C / 8))), dim3(256), 0, stream,
                                             (const T*)in, (T*)out, Mo, H, W, C, Ho, Wo));
    return istvt_check_launch();
}
