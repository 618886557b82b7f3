// LayerNorm forward / backward over rows of width D (reference: nn.LayerNorm inside PreNorm,
// network/vivit/module.py:15-21; STTransformer.norm vivit.py:89; mlp_head[0] vivit.py:128).
//
// One wavefront per row: the row lives in registers (8-element chunks, 16 B bf16 / 32 B f32 per
// lane access), mean and variance are two wave reductions (two-pass, fp32), gamma/beta are fp32.
// HBM-bound: algorithmic traffic = one read + one write of the row (forward).
//
// (The frame difference of module.py:193 is not taken here: TemporalResidualAttention projects the LayerNorm output
// once and the temporal attention kernels difference q and k, see attn_temporal.hip.)
#include "common.h"
#include <cstdlib>

constexpr int LN_MAXCH = 2;   // chunks of 8 per lane -> D <= 1024

template <typename T>
__device__ __forceinline__ void ln_row_load(const T* row, int D, int lane, float (&v)[LN_MAXCH][8]) {
#pragma unroll
    for (int c = 0; c < LN_MAXCH; ++c) {
        const int e = (lane + 64 * c) * 8;
        if (e < D) load8(row + e, v[c]);
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[c][i] = 0.f;
        }
    }
}

__device__ __forceinline__ void ln_stats(const float (&v)[LN_MAXCH][8], int D, int lane, float eps, float& mean,
                                         float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXCH; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[c][i];
    mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXCH; ++c) {
        const int e = (lane + 64 * c) * 8;
        if (e < D) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mean; q += d * d; }
        }
    }
    const float var = wave_sum(q) / (float)D;
    rstd = rsqrtf(var + eps);
}

// ------------------------------------------------------------------------------------------
// forward: rows independent
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     long M, int D, float eps, long ldx, long ldy) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    float gm[LN_MAXCH][8], bt[LN_MAXCH][8];
    ln_row_load<float>(gamma, D, lane, gm);
    ln_row_load<float>(beta, D, lane, bt);
    // the next row of this wavefront is requested before the current one is reduced: one row per (memory latency +
    // two wave reductions) per wavefront otherwise
    typename Mma<T>::frag nxt[LN_MAXCH];
    auto fetch = [&](long m) {
#pragma unroll
        for (int c = 0; c < LN_MAXCH; ++c) {
            const int e = (lane + 64 * c) * 8;
            if (e < D) nxt[c] = frag_load(x + m * ldx + e);
        }
    };
    if (wave < M) fetch(wave);
    for (long m = wave; m < M; m += nwaves) {
        float v[LN_MAXCH][8];
#pragma unroll
        for (int c = 0; c < LN_MAXCH; ++c) {
            const int e = (lane + 64 * c) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[c][i] = e < D ? Mma<T>::get(nxt[c], i) : 0.f;
        }
        if (m + nwaves < M) fetch(m + nwaves);
        float mean, rstd;
        ln_stats(v, D, lane, eps, mean, rstd);
#pragma unroll
        for (int c = 0; c < LN_MAXCH; ++c) {
            const int e = (lane + 64 * c) * 8;
            if (e < D) {
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (v[c][i] - mean) * rstd * gm[c][i] + bt[c][i];
                store8(y + m * ldy + e, o);
            }
        }
        if (lane == 0) { mean_out[m] = mean; rstd_out[m] = rstd; }
    }
}

// ------------------------------------------------------------------------------------------
// backward.  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
//            dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy
// dx (+= dres when given: gradient arriving through the residual connection)
// The loop is software-pipelined by one row: the loads of row m + nwaves are issued (as raw 16-byte chunks) before
// row m is reduced, so a wavefront always has a row in flight behind the two dependent wave reductions.
// Parameter gradients are reduced in a FIXED order: per lane over the rows of its wavefront, over the four wavefronts
// of a workgroup in LDS, then every workgroup stores its partial row to ws[workgroup][accumulator][D] and
// istvt_rows_reduce_add sums the workgroups in index order -- no floating-point atomics, so two runs give the same bits
// (and no tail of 3 x 728 same-address atomics per workgroup, which is what capped the grid at 512 workgroups).
template <typename T> struct LnRaw;
template <> struct LnRaw<bf16_t> { bf16x8 v; };
template <> struct LnRaw<float> { float4 a, b; };
__device__ __forceinline__ void raw_load(const bf16_t* p, LnRaw<bf16_t>& r) { r.v = *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void raw_load(const float* p, LnRaw<float>& r) {
    r.a = *reinterpret_cast<const float4*>(p);
    r.b = *reinterpret_cast<const float4*>(p + 4);
}
__device__ __forceinline__ void raw_f32(const LnRaw<bf16_t>& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r.v[i];
}
__device__ __forceinline__ void raw_f32(const LnRaw<float>& r, float (&v)[8]) {
    v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
}
template <typename T> struct LnBwdRow {
    LnRaw<T> dy[LN_MAXCH], x[LN_MAXCH], rs[LN_MAXCH];
    float mean, rstd;
};

// DCOL: also accumulate the column sums of dx (in fp32, before rounding) -- dx is the output gradient of the
// Linear that produced this LayerNorm's input, so this IS that Linear's bias gradient and saves its own pass over dx.
template <typename T, bool DCOL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy1, const T* __restrict__ x,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ ws, long M, int D,
                                                     long ld_dy, long ld_x, long ld_res, long ld_dx) {
    constexpr int NACC = DCOL ? 3 : 2;
    __shared__ float red[NACC][4][LN_MAXCH * 64 * 8];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long wave = (long)blockIdx.x * 4 + wid;
    const long nwaves = (long)gridDim.x * 4;
    float gm[LN_MAXCH][8];
    ln_row_load<float>(gamma, D, lane, gm);
    float ag[LN_MAXCH][8], ab[LN_MAXCH][8], ac[DCOL ? LN_MAXCH : 1][8];
#pragma unroll
    for (int c = 0; c < LN_MAXCH; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) { ag[c][i] = 0.f; ab[c][i] = 0.f; if (DCOL) ac[c][i] = 0.f; }
    bool on[LN_MAXCH];
#pragma unroll
    for (int c = 0; c < LN_MAXCH; ++c) on[c] = (lane + 64 * c) * 8 < D;

    auto fetch = [&](LnBwdRow<T>& r, long m) {
#pragma unroll
        for (int c = 0; c < LN_MAXCH; ++c) {
            if (!on[c]) continue;
            const int e = (lane + 64 * c) * 8;
            raw_load(dy1 + m * ld_dy + e, r.dy[c]);
            raw_load(x + m * ld_x + e, r.x[c]);
            if (dres) raw_load(dres + m * ld_res + e, r.rs[c]);
        }
        r.mean = mean_in[m];
        r.rstd = rstd_in[m];
    };

    LnBwdRow<T> cur, nxt;
    if (wave < M) fetch(cur, wave);
    for (long m = wave; m < M; m += nwaves) {
        const bool more = m + nwaves < M;
        if (more) fetch(nxt, m + nwaves);
        float dy[LN_MAXCH][8], xv[LN_MAXCH][8];
        const float mean = cur.mean, rstd = cur.rstd;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < LN_MAXCH; ++c) {
            if (on[c]) {
                raw_f32(cur.dy[c], dy[c]);
                raw_f32(cur.x[c], xv[c]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float xh = (xv[c][i] - mean) * rstd;
                    xv[c][i] = xh;
                    const float gdy = dy[c][i] * gm[c][i];
                    s1 += gdy;
                    s2 += gdy * xh;
                    ag[c][i] += dy[c][i] * xh;
                    ab[c][i] += dy[c][i];
                }
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int c = 0; c < LN_MAXCH; ++c) {
            if (on[c]) {
                const int e = (lane + 64 * c) * 8;
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = rstd * (dy[c][i] * gm[c][i] - s1 - xv[c][i] * s2);
                if (dres) {
                    float rr[8];
                    raw_f32(cur.rs[c], rr);
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] += rr[i];
                }
                if (DCOL) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) ac[c][i] += o[i];
                }
                store8(dx + m * ld_dx + e, o);
            }
        }
        if (more) cur = nxt;
    }
    // the four wavefronts' sums through LDS, in wavefront order, then this workgroup's partial row per accumulator
#pragma unroll
    for (int c = 0; c < LN_MAXCH; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            red[0][wid][(c * 64 + lane) * 8 + i] = ag[c][i];
            red[1][wid][(c * 64 + lane) * 8 + i] = ab[c][i];
            if (DCOL) red[NACC - 1][wid][(c * 64 + lane) * 8 + i] = ac[c][i];
        }
    __syncthreads();
    float* mine = ws + (long)blockIdx.x * NACC * D;
    for (int col = threadIdx.x; col < D; col += 256) {
#pragma unroll
        for (int a = 0; a < NACC; ++a)
            mine[a * D + col] = (red[a][0][col] + red[a][1][col]) + (red[a][2][col] + red[a][3][col]);
    }
}

static int ln_grid(long rows) {
    static const long cap = istvt_tune("ISTVT_LN_BLOCKS", 4096);   // 2048 -> 4096: -8 % on the forward LayerNorm
    long blocks = (rows + 3) / 4;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// ld*: row strides in elements (>= D, multiples of 8): activations are kept with line-aligned rows (ops.py)
extern "C" int istvt_layernorm_fwd(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy,
                                   float* mean, float* rstd, long M, int D, float eps, int dtype, hipStream_t stream) {
    if (D % 8 != 0 || D > LN_MAXCH * 512 || M <= 0 || ldx < D || ldy < D || ldx % 8 || ldy % 8) return ISTVT_ERR_SHAPE;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_fwd_kernel<T>), dim3(ln_grid(M)), dim3(256), 0, stream, (const T*)x,
                                             gamma, beta, (T*)y, mean, rstd, M, D, eps, ldx, ldy));
    return istvt_check_launch();
}

// Workgroups of the backward launch for M rows = rows of the partial-sum workspace per accumulator.  Measured at C2
// (M = 56 736, bf16, with the residual input; tools/ln_bench.py): 256 -> 94.8 us, 512 -> 66.3, 1024 -> 74.1, 2048 -> 72.7
// (the atomics version of round 2: 65.6 at 512).
static long ln_bwd_blocks(long M) {
    static const long cap = istvt_tune("ISTVT_LN_BWD_BLOCKS", 512);
    long blocks = (M + 3) / 4;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}
extern "C" int istvt_layernorm_bwd_ws_elems(long M, int D) { return (int)(ln_bwd_blocks(M) * 3 * D); }

// dres may be null.  dgamma / dbeta accumulate (+=); dcol (may be null) accumulates the column sums of dx.
// ws: float workspace of at least istvt_layernorm_bwd_ws_elems(M, D) elements (per-workgroup partial sums; contents
// are scratch).  Bit-reproducible: no floating-point atomics.
extern "C" int istvt_layernorm_bwd(const void* dy, long ld_dy, const void* x, long ld_x, const float* mean,
                                   const float* rstd, const float* gamma, const void* dres, long ld_res, void* dx,
                                   long ld_dx, float* dgamma, float* dbeta, float* dcol, float* ws, long ws_elems, long M,
                                   int D, int dtype, hipStream_t stream) {
    if (D % 8 != 0 || D > LN_MAXCH * 512 || M <= 0 || !ws) return ISTVT_ERR_SHAPE;
    if (ld_dy < D || ld_x < D || ld_dx < D || ld_dy % 8 || ld_x % 8 || ld_dx % 8) return ISTVT_ERR_SHAPE;
    if (dres && (ld_res < D || ld_res % 8)) return ISTVT_ERR_SHAPE;
    const long blocks = ln_bwd_blocks(M);
    const int nacc = dcol ? 3 : 2;
    if (ws_elems < blocks * nacc * D) return ISTVT_ERR_SHAPE;
    if (dcol)
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, true>), dim3((int)blocks), dim3(256), 0, stream,
                                                 (const T*)dy, (const T*)x, mean, rstd, gamma, (const T*)dres, (T*)dx, ws, M,
                                                 D, ld_dy, ld_x, ld_res, ld_dx));
    else
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, false>), dim3((int)blocks), dim3(256), 0, stream,
                                                 (const T*)dy, (const T*)x, mean, rstd, gamma, (const T*)dres, (T*)dx, ws, M,
                                                 D, ld_dy, ld_x, ld_res, ld_dx));
    int rc = istvt_check_launch();
    if (rc) return rc;
    return istvt_rows_reduce_add(ws, (int)blocks, nacc, D, dgamma, dbeta, dcol, stream);
}
