// Kernels of the next-row components, all HBM-bound, 16-byte accesses along the feature dim:
//   * prepend: token assembly of the ablation transformers ViViT / VanillaTr (reference network/vivit/vivit.py:60-67,
//     74-75, 180-186): one learned token in front of every sequence, plus an optional positional embedding;
//   * seq_take / seq_mean: x[:, 0] and x.mean(dim=1) of a [S][n][D] batch of sequences (vivit.py:71,79,189);
//   * relu_avgpool: Xception.logits' ReLU + adaptive_avg_pool2d((1,1)) on NHWC features (network/xception.py:208-213);
//   * dropout: nn.Dropout in training mode (module.py:29,31,78,187; models_copy.py:41-44) with a Philox4x32-10 stream
//     keyed by (seed, element index / 4): the mask is stored (one byte per element) for the backward.
#include "common.h"

// ------------------------------------------------------------------------------------------ prepend
// out[s][0] = tok (+ pos[s % period][0]);  out[s][1+i] = src[s][i] (+ pos[s % period][1+i]);   pos: float [period][pos_rows][D]
template <typename T>
__global__ __launch_bounds__(128) void prepend_fwd_kernel(const T* __restrict__ src, const float* __restrict__ tok,
                                                          const float* __restrict__ pos, T* __restrict__ out, long ldo,
                                                          int n, int D, int period, int pos_rows) {
    const long row = blockIdx.x;                 // over S * (n + 1)
    const int i = (int)(row % (n + 1));
    const long s = row / (n + 1);
    for (int e = threadIdx.x * 8; e < D; e += 128 * 8) {
        float o[8];
        if (i == 0) load8(tok + e, o);
        else load8(src + (s * n + (i - 1)) * D + e, o);
        if (pos) {
            float pe[8];
            load8(pos + ((long)(s % period) * pos_rows + i) * D + e, pe);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += pe[j];
        }
        store8(out + row * ldo + e, o);
    }
}

// grid (n + 1, period): dsrc[s][i] = dout[s][1+i];  dpos[q][i] += sum_{s % period == q} dout[s][i];
// dtok += sum_s dout[s][0]  (atomics over the `period` workgroups of row 0)
template <typename T>
__global__ __launch_bounds__(128) void prepend_bwd_kernel(const T* __restrict__ dout, long ldd, T* __restrict__ dsrc,
                                                          float* __restrict__ dtok, float* __restrict__ dpos, long S,
                                                          int n, int D, int period, int pos_rows) {
    const int i = blockIdx.x, q = blockIdx.y;
    for (int e = threadIdx.x * 8; e < D; e += 128 * 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (long s = q; s < S; s += period) {
            float v[8];
            load8(dout + (s * (n + 1) + i) * ldd + e, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
            if (i > 0 && dsrc) store8(dsrc + (s * n + (i - 1)) * D + e, v);
        }
        if (dpos) {
            float* dp = dpos + ((long)q * pos_rows + i) * D + e;
#pragma unroll
            for (int j = 0; j < 8; ++j) dp[j] += acc[j];
        }
        if (i == 0 && dtok) {                   // dtok here = workspace [period][D]: one partial row per q
#pragma unroll
            for (int j = 0; j < 8; ++j) dtok[(long)q * D + e + j] = acc[j];
        }
    }
}

extern "C" int istvt_prepend_fwd(const void* src, const float* tok, const float* pos, void* out, long ldo, long S, int n,
                                 int D, int period, int pos_rows, int dtype, hipStream_t stream) {
    if (S <= 0 || n <= 0 || D <= 0 || D % 8 || ldo < D || ldo % 8 || !tok) return ISTVT_ERR_SHAPE;
    if (pos && (period <= 0 || pos_rows < n + 1 || S % period)) return ISTVT_ERR_SHAPE;
    if (S * (n + 1) > 0x7fffffffL) return ISTVT_ERR_SHAPE;
    dim3 grid((unsigned)(S * (n + 1))), block(128);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((prepend_fwd_kernel<T>), grid, block, 0, stream, (const T*)src, tok, pos, (T*)out,
                                             ldo, n, D, pos ? period : 1, pos_rows));
    return istvt_check_launch();
}

// rows of the float workspace istvt_prepend_bwd needs when dtok is given (D floats each)
extern "C" int istvt_prepend_bwd_ws_rows(long S, int period, int has_pos) {
    if (S <= 0) return ISTVT_ERR_SHAPE;
    return has_pos ? period : (int)(S < 64 ? S : 64);
}

// dtok (may be null) accumulates; ws: float scratch of istvt_prepend_bwd_ws_rows(...) * D elements (needed with dtok):
// the class-token gradient is summed from per-workgroup partial rows in a fixed order
extern "C" int istvt_prepend_bwd(const void* dout, long ldd, void* dsrc, float* dtok, float* dpos, float* ws, long S, int n,
                                 int D, int period, int pos_rows, int dtype, hipStream_t stream) {
    if (S <= 0 || n <= 0 || D <= 0 || D % 8 || ldd < D || ldd % 8) return ISTVT_ERR_SHAPE;
    if (dpos && (period <= 0 || pos_rows < n + 1 || S % period)) return ISTVT_ERR_SHAPE;
    if (dtok && !ws) return ISTVT_ERR_SHAPE;
    if (!dpos) period = (int)(S < 64 ? S : 64);          // no per-period sums: spread the sequences over <= 64 workgroups per row
    dim3 grid(n + 1, period), block(128);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((prepend_bwd_kernel<T>), grid, block, 0, stream, (const T*)dout, ldd, (T*)dsrc,
                                             dtok ? ws : nullptr, dpos, S, n, D, period, pos_rows));
    int rc = istvt_check_launch();
    if (rc || !dtok) return rc;
    return istvt_rows_reduce_add(ws, period, 1, D, dtok, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------------------------------ seq_mean
// out[s] = mean_i x[s][i]   (adjoint: dx[s][i] = dout[s] / n)
template <typename T>
__global__ __launch_bounds__(128) void seq_mean_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ out, int n, int D) {
    const long s = blockIdx.x;
    for (int e = threadIdx.x * 8; e < D; e += 128 * 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            float v[8];
            load8(x + (s * n + i) * ldx + e, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
        const float inv = 1.0f / (float)n;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= inv;
        store8(out + s * D + e, acc);
    }
}

template <typename T>
__global__ __launch_bounds__(128) void seq_mean_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dx, long ldx, int n, int D) {
    const long row = blockIdx.x;
    const long s = row / n;
    const float inv = 1.0f / (float)n;
    for (int e = threadIdx.x * 8; e < D; e += 128 * 8) {
        float v[8];
        load8(dout + s * D + e, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= inv;
        store8(dx + row * ldx + e, v);
    }
}

extern "C" int istvt_seq_mean_fwd(const void* x, long ldx, void* out, long S, int n, int D, int dtype, hipStream_t stream) {
    if (S <= 0 || n <= 0 || D % 8 || ldx < D || ldx % 8 || S > 0x7fffffffL) return ISTVT_ERR_SHAPE;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((seq_mean_fwd_kernel<T>), dim3((unsigned)S), dim3(128), 0, stream, (const T*)x, ldx,
                                             (T*)out, n, D));
    return istvt_check_launch();
}

extern "C" int istvt_seq_mean_bwd(const void* dout, void* dx, long ldx, long S, int n, int D, int dtype, hipStream_t stream) {
    if (S <= 0 || n <= 0 || D % 8 || ldx < D || ldx % 8 || S * n > 0x7fffffffL) return ISTVT_ERR_SHAPE;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((seq_mean_bwd_kernel<T>), dim3((unsigned)(S * n)), dim3(128), 0, stream,
                                             (const T*)dout, (T*)dx, ldx, n, D));
    return istvt_check_launch();
}

// ------------------------------------------------------------------------------------------ relu + global average pool
// x NHWC [Fr][HW][C] -> out [Fr][C] = mean_hw relu(x);  backward: dx = (x > 0) * dout / HW
template <typename T>
__global__ __launch_bounds__(128) void relu_avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, int HW, int C, int relu) {
    const long f = blockIdx.x;
    for (int e = (blockIdx.y * 128 + threadIdx.x) * 8; e < C; e += gridDim.y * 128 * 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < HW; ++i) {
            float v[8];
            load8(x + (f * HW + i) * C + e, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += relu ? fmaxf(v[j], 0.f) : v[j];
        }
        const float inv = 1.0f / (float)HW;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= inv;
        store8(out + f * C + e, acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void relu_avgpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dout,
                                                               T* __restrict__ dx, long M, int HW, int C, int relu) {
    const int vpr = C / 8;
    const long nvec = M * vpr, stride = (long)gridDim.x * 256;
    const float inv = 1.0f / (float)HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        const long m = i / vpr;
        const int e = (int)(i % vpr) * 8;
        float v[8], d[8];
        load8(x + m * C + e, v);
        load8(dout + (m / HW) * C + e, d);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = (!relu || v[j] > 0.f) ? d[j] * inv : 0.f;
        store8(dx + m * C + e, d);
    }
}

extern "C" int istvt_relu_avgpool_fwd(const void* x, void* out, int Fr, int HW, int C, int relu, int dtype, hipStream_t stream) {
    if (Fr <= 0 || HW <= 0 || C <= 0 || C % 8) return ISTVT_ERR_SHAPE;
    dim3 grid(Fr, (C / 8 + 127) / 128), block(128);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((relu_avgpool_fwd_kernel<T>), grid, block, 0, stream, (const T*)x, (T*)out, HW, C, relu));
    return istvt_check_launch();
}

extern "C" int istvt_relu_avgpool_bwd(const void* x, const void* dout, void* dx, int Fr, int HW, int C, int relu, int dtype,
                                      hipStream_t stream) {
    if (Fr <= 0 || HW <= 0 || C <= 0 || C % 8) return ISTVT_ERR_SHAPE;
    const long M = (long)Fr * HW;
    long blocks = (M * (C / 8) + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((relu_avgpool_bwd_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, stream,
                                             (const T*)x, (const T*)dout, (T*)dx, M, HW, C, relu));
    return istvt_check_launch();
}

// ------------------------------------------------------------------------------------------ dropout
// Philox4x32-10 (Salmon et al.): counter = (group index, 0, 0, 0), key = seed; four 32-bit outputs per group of four
// consecutive elements.  keep = u >= p * 2^32 (so P(keep) = 1 - p);  y = keep ? x / (1 - p) : 0.
__device__ __forceinline__ void philox4x32_10(unsigned long long ctr, unsigned long long seed, unsigned (&r)[4]) {
    unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0u, c3 = 0u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy,
                                                          unsigned char* __restrict__ mask, long M, int D, float p,
                                                          unsigned long long seed) {
    const int vpr = D / 8;
    const long nvec = M * vpr, stride = (long)gridDim.x * 256;
    const unsigned thr = (unsigned)fminf(p * 4294967296.0f, 4294967295.0f);
    const float sc = 1.0f / (1.0f - p);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        const long m = i / vpr;
        const int e = (int)(i % vpr) * 8;
        float v[8];
        load8(x + m * ldx + e, v);
        unsigned r[8];
        unsigned ra[4], rb[4];
        philox4x32_10((unsigned long long)(2 * i), seed, ra);
        philox4x32_10((unsigned long long)(2 * i + 1), seed, rb);
#pragma unroll
        for (int j = 0; j < 4; ++j) { r[j] = ra[j]; r[4 + j] = rb[j]; }
        unsigned long long packed = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool keep = r[j] >= thr;
            v[j] = keep ? v[j] * sc : 0.f;
            packed |= (unsigned long long)(keep ? 1u : 0u) << (8 * j);
        }
        store8(y + m * ldy + e, v);
        *reinterpret_cast<unsigned long long*>(mask + m * D + e) = packed;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const T* __restrict__ dy, long ldy, const unsigned char* __restrict__ mask,
                                                          T* __restrict__ dx, long ldx, long M, int D, float p) {
    const int vpr = D / 8;
    const long nvec = M * vpr, stride = (long)gridDim.x * 256;
    const float sc = 1.0f / (1.0f - p);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        const long m = i / vpr;
        const int e = (int)(i % vpr) * 8;
        float v[8];
        load8(dy + m * ldy + e, v);
        const unsigned long long packed = *reinterpret_cast<const unsigned long long*>(mask + m * D + e);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ((packed >> (8 * j)) & 1u) ? v[j] * sc : 0.f;
        store8(dx + m * ldx + e, v);
    }
}

extern "C" int istvt_dropout_fwd(const void* x, long ldx, void* y, long ldy, unsigned char* mask, long M, int D, float p,
                                 unsigned long long seed, int dtype, hipStream_t stream) {
    if (M <= 0 || D <= 0 || D % 8 || ldx < D || ldy < D || ldx % 8 || ldy % 8 || !(p >= 0.f && p < 1.f)) return ISTVT_ERR_SHAPE;
    long blocks = (M * (D / 8) + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dropout_fwd_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, stream, (const T*)x,
                                             ldx, (T*)y, ldy, mask, M, D, p, seed));
    return istvt_check_launch();
}

extern "C" int istvt_dropout_bwd(const void* dy, long ldy, const unsigned char* mask, void* dx, long ldx, long M, int D, float p,
                                 int dtype, hipStream_t stream) {
    if (M <= 0 || D <= 0 || D % 8 || ldx < D || ldy < D || ldx % 8 || ldy % 8 || !(p >= 0.f && p < 1.f)) return ISTVT_ERR_SHAPE;
    long blocks = (M * (D / 8) + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dropout_bwd_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, stream, (const T*)dy,
                                             ldy, mask, (T*)dx, ldx, M, D, p));
    return istvt_check_launch();
}

// ------------------------------------------------------------------------------------------ out = a + b (row-strided)
// the residual add of a block whose output projection is followed by an active Dropout (module.py:78,187 with p > 0:
// the add can then no longer ride in the projection GEMM's epilogue)
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, long lda, const T* __restrict__ b, long ldb,
                                                  T* __restrict__ out, long ldo, long M, int D) {
    const int vpr = D / 8;
    const long nvec = M * vpr, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        const long m = i / vpr;
        const int e = (int)(i % vpr) * 8;
        float x[8], y[8];
        load8(a + m * lda + e, x);
        load8(b + m * ldb + e, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
        store8(out + m * ldo + e, x);
    }
}

extern "C" int istvt_add(const void* a, long lda, const void* b, long ldb, void* out, long ldo, long M, int D, int dtype,
                         hipStream_t stream) {
    if (M <= 0 || D <= 0 || D % 8 || lda < D || ldb < D || ldo < D || lda % 8 || ldb % 8 || ldo % 8) return ISTVT_ERR_SHAPE;
    long blocks = (M * (D / 8) + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((add_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, stream, (const T*)a, lda,
                                             (const T*)b, ldb, (T*)out, ldo, M, D));
    return istvt_check_launch();
}
