// LayerNorm forward / backward over rows of width D (reference: nn.LayerNorm inside PreNorm,
// network/vivit/module.py:15-21; STTransformer.norm vivit.py:89; mlp_head[0] vivit.py:128).
//
// One wavefront per row: the row lives in registers (8-element chunks, 16 B bf16 / 32 B f32 per
// lane access), mean and variance are two wave reductions (two-pass, fp32), gamma/beta are fp32.
// HBM-bound: algorithmic traffic = one read + one write of the row (forward).
//
// The "temporal" variants also produce / consume the frame difference that
// TemporalResidualAttention feeds to to_qk (module.py:193): rows are ordered (b, f, p) and
//   diff[b,f,p] = y[b,f,p]                 f < 2
//               = y[b,f,p] - y[b,f-1,p]    f >= 2
// so the forward writes both y and diff in one pass (one wavefront walks the F frames of a
// position), and the backward folds  dy = g_y + g_diff[f] - g_diff[f+1]  into its load.
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

// forward, temporal variant: one wave per (b, p), walks f = 0..F-1; rows at b*F*P + f*P + p
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_diff_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ y,
                                                          T* __restrict__ diff, float* __restrict__ mean_out,
                                                          float* __restrict__ rstd_out, int Bn, int F, int P, int D,
                                                          float eps, long ldx, long ldy, long ldd) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    float gm[LN_MAXCH][8], bt[LN_MAXCH][8];
    ln_row_load<float>(gamma, D, lane, gm);
    ln_row_load<float>(beta, D, lane, bt);
    const long npos = (long)Bn * P;
    // the next row of the walk (next frame, or frame 0 of this wavefront's next position) is requested before the
    // current one is reduced (see ln_fwd_kernel)
    typename Mma<T>::frag nxt[LN_MAXCH];
    auto fetch = [&](long m) {
#pragma unroll
        for (int c = 0; c < LN_MAXCH; ++c) {
            const int e = (lane + 64 * c) * 8;
            if (e < D) nxt[c] = frag_load(x + m * ldx + e);
        }
    };
    if (wave < npos) fetch(((wave / P) * F) * P + wave % P);
    for (long w = wave; w < npos; w += nwaves) {
        const long b = w / P, pp = w % P;
        float prev[LN_MAXCH][8];
        for (int f = 0; f < F; ++f) {
            const long m = (b * F + f) * P + pp;
            float v[LN_MAXCH][8];
#pragma unroll
            for (int c = 0; c < LN_MAXCH; ++c) {
                const int e = (lane + 64 * c) * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[c][i] = e < D ? Mma<T>::get(nxt[c], i) : 0.f;
            }
            if (f + 1 < F) fetch(m + P);
            else if (w + nwaves < npos) fetch((((w + nwaves) / P) * F) * P + (w + nwaves) % P);
            float mean, rstd;
            ln_stats(v, D, lane, eps, mean, rstd);
#pragma unroll
            for (int c = 0; c < LN_MAXCH; ++c) {
                const int e = (lane + 64 * c) * 8;
                if (e < D) {
                    float o[8], dd[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        o[i] = (v[c][i] - mean) * rstd * gm[c][i] + bt[c][i];
                        dd[i] = (f >= 2) ? o[i] - prev[c][i] : o[i];
                        prev[c][i] = o[i];
                    }
                    store8(y + m * ldy + e, o);
                    store8(diff + m * ldd + e, dd);
                }
            }
            if (lane == 0) { mean_out[m] = mean; rstd_out[m] = rstd; }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward.  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
//            dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy        (fp32 atomics per block)
// dy = dy1 (+ dy2[m] - dy2[m + P] when frame(m)+1 in [2, F-1])   [temporal variant, dy2 != null]
// dx (+= dres when given: gradient arriving through the residual connection)
// The loop is software-pipelined by one row: the loads of row m + nwaves are issued (as raw 16-byte chunks) before
// row m is reduced, so a wavefront always has a row in flight behind the two dependent wave reductions.
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
    LnRaw<T> dy[LN_MAXCH], x[LN_MAXCH], t0[LN_MAXCH], t1[LN_MAXCH], rs[LN_MAXCH];
    float mean, rstd;
    bool sub;          // temporal variant: subtract dy2 of the next frame
};

// DCOL: also accumulate the column sums of dx (in fp32, before rounding) into dcol -- dx is the output gradient of the
// Linear that produced this LayerNorm's input, so this IS that Linear's bias gradient and saves its own pass over dx.
template <typename T, bool DCOL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy1, const T* __restrict__ dy2,
                                                     const T* __restrict__ x, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, const float* __restrict__ gamma,
                                                     const T* __restrict__ dres, T* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ dcol, long M,
                                                     int D, int F, int P, long ld_dy, long ld_dy2, long ld_x, long ld_res,
                                                     long ld_dx) {
    __shared__ float red[2][4][LN_MAXCH * 64 * 8];
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
    const long N = (long)F * P;
    bool on[LN_MAXCH];
#pragma unroll
    for (int c = 0; c < LN_MAXCH; ++c) on[c] = (lane + 64 * c) * 8 < D;

    auto fetch = [&](LnBwdRow<T>& r, long m) {
        r.sub = false;
        if (dy2) {
            const int f = (int)((m % N) / P);
            r.sub = f + 1 >= 2 && f + 1 < F;
        }
#pragma unroll
        for (int c = 0; c < LN_MAXCH; ++c) {
            if (!on[c]) continue;
            const int e = (lane + 64 * c) * 8;
            raw_load(dy1 + m * ld_dy + e, r.dy[c]);
            raw_load(x + m * ld_x + e, r.x[c]);
            if (dy2) {
                raw_load(dy2 + m * ld_dy2 + e, r.t0[c]);
                if (r.sub) raw_load(dy2 + (m + P) * ld_dy2 + e, r.t1[c]);
            }
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
                if (dy2) {
                    float t[8];
                    raw_f32(cur.t0[c], t);
#pragma unroll
                    for (int i = 0; i < 8; ++i) dy[c][i] += t[i];
                    if (cur.sub) {
                        raw_f32(cur.t1[c], t);
#pragma unroll
                        for (int i = 0; i < 8; ++i) dy[c][i] -= t[i];
                    }
                }
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
    // block reduction of the parameter gradients, then one atomic per column per block
#pragma unroll
    for (int c = 0; c < LN_MAXCH; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            red[0][wid][(c * 64 + lane) * 8 + i] = ag[c][i];
            red[1][wid][(c * 64 + lane) * 8 + i] = ab[c][i];
        }
    __syncthreads();
    for (int col = threadIdx.x; col < LN_MAXCH * 512; col += 256) {
        if (col < D) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a += red[0][w][col]; b += red[1][w][col]; }
            atomicAdd(dgamma + col, a);
            atomicAdd(dbeta + col, b);
        }
    }
    if (DCOL) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < LN_MAXCH; ++c)
#pragma unroll
            for (int i = 0; i < 8; ++i) red[0][wid][(c * 64 + lane) * 8 + i] = ac[c][i];
        __syncthreads();
        for (int col = threadIdx.x; col < LN_MAXCH * 512; col += 256) {
            if (col < D) atomicAdd(dcol + col, (red[0][0][col] + red[0][1][col]) + (red[0][2][col] + red[0][3][col]));
        }
    }
}

static int ln_grid(long rows) {
    static const long cap = getenv("ISTVT_LN_BLOCKS") ? atol(getenv("ISTVT_LN_BLOCKS")) : 4096;   // 2048 -> 4096: -8 % on the forward LayerNorm
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

extern "C" int istvt_layernorm_fwd_diff(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy,
                                        void* diff, long ldd, float* mean, float* rstd, int B, int F, int P, int D,
                                        float eps, int dtype, hipStream_t stream) {
    if (D % 8 != 0 || D > LN_MAXCH * 512 || B <= 0 || F <= 0 || P <= 0) return ISTVT_ERR_SHAPE;
    if (ldx < D || ldy < D || ldd < D || ldx % 8 || ldy % 8 || ldd % 8) return ISTVT_ERR_SHAPE;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_fwd_diff_kernel<T>), dim3(ln_grid((long)B * P)), dim3(256), 0, stream,
                                             (const T*)x, gamma, beta, (T*)y, (T*)diff, mean, rstd, B, F, P, D, eps, ldx,
                                             ldy, ldd));
    return istvt_check_launch();
}

// dy2 == null: plain LayerNorm backward (F, P ignored).  dres may be null.  dgamma/dbeta accumulate; dcol (may be
// null) accumulates the column sums of dx.
extern "C" int istvt_layernorm_bwd(const void* dy, long ld_dy, const void* dy2, long ld_dy2, const void* x, long ld_x,
                                   const float* mean, const float* rstd, const float* gamma, const void* dres,
                                   long ld_res, void* dx, long ld_dx, float* dgamma, float* dbeta, float* dcol, long M,
                                   int D, int F, int P, int dtype, hipStream_t stream) {
    if (D % 8 != 0 || D > LN_MAXCH * 512 || M <= 0) return ISTVT_ERR_SHAPE;
    if (ld_dy < D || ld_x < D || ld_dx < D || ld_dy % 8 || ld_x % 8 || ld_dx % 8) return ISTVT_ERR_SHAPE;
    if ((dy2 && (ld_dy2 < D || ld_dy2 % 8)) || (dres && (ld_res < D || ld_res % 8))) return ISTVT_ERR_SHAPE;
    if (dy2 && (F <= 0 || P <= 0 || M % ((long)F * P) != 0)) return ISTVT_ERR_SHAPE;
    if (!dy2) { F = 1; P = 1; }
    long blocks = (M + 3) / 4;
    // every workgroup ends with one atomic per column per accumulator into the SAME 728 (x2, x3) addresses; they all
    // finish together, so the tail grows with the workgroup count: measured at C2 (38 launches / step)
    // 256 -> 2.95 ms, 512 -> 2.83 ms, 1024 -> 3.28 ms, 2048 -> 3.84 ms
    static const long cap = getenv("ISTVT_LN_BWD_BLOCKS") ? atol(getenv("ISTVT_LN_BWD_BLOCKS")) : 512;
    if (blocks > cap) blocks = cap;
    if (dcol)
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, true>), dim3((int)blocks), dim3(256), 0, stream,
                                                 (const T*)dy, (const T*)dy2, (const T*)x, mean, rstd, gamma,
                                                 (const T*)dres, (T*)dx, dgamma, dbeta, dcol, M, D, F, P, ld_dy, ld_dy2,
                                                 ld_x, ld_res, ld_dx));
    else
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, false>), dim3((int)blocks), dim3(256), 0, stream,
                                                 (const T*)dy, (const T*)dy2, (const T*)x, mean, rstd, gamma,
                                                 (const T*)dres, (T*)dx, dgamma, dbeta, dcol, M, D, F, P, ld_dy, ld_dy2,
                                                 ld_x, ld_res, ld_dx));
    return istvt_check_launch();
}
