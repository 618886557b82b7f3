// Fused per-frame spatial self-attention (reference: SpatialOnlyAttention.forward,
// network/vivit/module.py:81-93): for every (clip b, frame f, head h)
//     O = softmax(Q K^T * DH^-1/2) V        Q,K,V: [P tokens][DH]
// read straight out of the packed projection buffer qkv[M][3*inner] (q | k | v, heads h-major
// inside each third: 'b n (h d)'), written as out[M][inner]; scores never leave registers.
//
// MFMA formulation (16x16 tiles, K=32 logical steps, see common.h):
//   S^T = K Q^T      keys on accumulator rows, queries on lanes  (A = K rows from LDS,
//                    B = Q rows held in registers)
//   softmax over keys = in-lane over the tile registers + 2 cross-lane-group shuffles
//   O^T = V^T P^T    the S^T accumulators ARE the B operand (k-slot (g,j) <-> key
//                    32s + 16(j>>2) + 4g + (j&3)); A = V read transposed from its row-major LDS
//                    image (ds_read_b64_tr_b16 for bf16).
// One wavefront owns 32 queries for the whole kernel; the 4 wavefronts of a workgroup share the
// K/V images; keys stream through LDS in chunks of 128 with online-softmax rescaling, so any P
// works (197 -> 2 chunks, 362 -> 3).
//
// Backward = two kernels that recompute P from the saved log-sum-exp:
//   sattn_bwd_dq : same geometry as forward; dS^T = P^T o (dP^T - delta) ; dQ^T += K^T dS^T
//   sattn_bwd_dkv: one wavefront owns 32 keys; queries stream through LDS;
//                  dV^T += dO^T P ; dK^T += Q^T dS
#include "common.h"
#include <type_traits>
#include <cstdlib>

constexpr int CHUNK = 128;          // rows of an LDS image
constexpr int NT = CHUNK / 16;      // 16-row tiles per chunk
constexpr int IPAD = 8;             // row padding (elements)
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// copy rows [row0, row0+CHUNK) x DH columns of a [rows][ld] matrix into an LDS image, zero-filling
// rows >= nrows
template <typename T, int DH, int NTHR = 256>
__device__ __forceinline__ void stage_img(T* img, const T* __restrict__ src, long ld, int row0, int nrows, int tid) {
    constexpr int LDI = DH + IPAD;
    constexpr int VPR = DH / 8;
    constexpr int NV = CHUNK * VPR / NTHR;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + NTHR * i;
        const int row = v / VPR, col = (v % VPR) * 8;
        typename Mma<T>::frag f;
        if (row0 + row < nrows) f = frag_load(src + (long)(row0 + row) * ld + col);
        else f = Mma<T>::zero();
        if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<bf16x8*>(img + row * LDI + col) = f;
        } else {
            float* p = (float*)(img + row * LDI + col);
            *reinterpret_cast<float4*>(p) = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
            *reinterpret_cast<float4*>(p + 4) = make_float4(f.v[4], f.v[5], f.v[6], f.v[7]);
        }
    }
}

// the same copy split in two: global -> registers (issued a chunk ahead, in flight under the current chunk's MFMAs)
// and registers -> LDS image
template <typename T, int DH, int NTHR>
struct StageRegs { typename Mma<T>::frag f[CHUNK * (DH / 8) / NTHR]; };
template <typename T, int DH, int NTHR>
__device__ __forceinline__ void stage_fetch(StageRegs<T, DH, NTHR>& rg, const T* __restrict__ src, long ld, int row0,
                                            int nrows, int tid) {
    constexpr int VPR = DH / 8;
    constexpr int NV = CHUNK * VPR / NTHR;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + NTHR * i;
        const int row = v / VPR, col = (v % VPR) * 8;
        if (row0 + row < nrows) rg.f[i] = frag_load(src + (long)(row0 + row) * ld + col);
        else rg.f[i] = Mma<T>::zero();
    }
}
template <typename T, int DH, int NTHR>
__device__ __forceinline__ void stage_commit(T* img, const StageRegs<T, DH, NTHR>& rg, int tid) {
    constexpr int LDI = DH + IPAD;
    constexpr int VPR = DH / 8;
    constexpr int NV = CHUNK * VPR / NTHR;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + NTHR * i;
        const int row = v / VPR, col = (v % VPR) * 8;
        if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<bf16x8*>(img + row * LDI + col) = rg.f[i];
        } else {
            float* p = (float*)(img + row * LDI + col);
            *reinterpret_cast<float4*>(p) = make_float4(rg.f[i].v[0], rg.f[i].v[1], rg.f[i].v[2], rg.f[i].v[3]);
            *reinterpret_cast<float4*>(p + 4) = make_float4(rg.f[i].v[4], rg.f[i].v[5], rg.f[i].v[6], rg.f[i].v[7]);
        }
    }
}

// fragment (8 consecutive d) of one row of a global matrix, zero if the row is out of range
template <typename T>
__device__ __forceinline__ typename Mma<T>::frag row_frag(const T* __restrict__ src, long ld, int row, int nrows,
                                                           int col) {
    if (row < nrows) return frag_load(src + (long)row * ld + col);
    return Mma<T>::zero();
}

template <typename T>
__device__ __forceinline__ typename Mma<T>::frag acc_frag(const f32x4& lo, const f32x4& hi) {
    typename Mma<T>::frag f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { Mma<T>::set(f, i, lo[i]); Mma<T>::set(f, 4 + i, hi[i]); }
    return f;
}

// ---- fp8 (OCP e4m3) operands for the attention MFMAs (BASELINE config 5): the bf16 fragments are converted in
// registers and fed to v_mfma_f32_16x16x32_fp8_fp8, whose 8-bytes-per-lane K = 32 operand layout is the bf16 one
__device__ __forceinline__ long pack_fp8(const float (&v)[8]) {
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ long to_fp8(const bf16x8& f) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)f[i];
    return pack_fp8(v);
}
__device__ __forceinline__ long to_fp8(const f32frag& f) { return pack_fp8(f.v); }
__device__ __forceinline__ void mma8(f32x4& c, long a, long b) { c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c, 0, 0, 0); }
constexpr float P8_SCALE = 256.f;          // probabilities (<= 1) are scaled into e4m3's normal range before conversion

__device__ __forceinline__ float group_max(float v) {      // over the 4 lane groups (same r)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------------------
// U = query tiles (of 16) per wavefront, 128 / (16 U) wavefronts per workgroup.  U = 1 (8 wavefronts) halves the
// registers of a wavefront: 4 instead of 2 wavefronts per SIMD, whose softmax (VALU) and MFMA phases then overlap and
// whose staging latencies hide each other (the kernel is bound by neither pipe: it waits).
template <typename T, int DH, int U, bool FP8 = false>
__global__ __launch_bounds__(512 / U, U == 1 ? 4 : 1) void sattn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                        float* __restrict__ lse, int P, int heads, float scale) {
    constexpr int LDI = DH + IPAD, KS = DH / 32, DT = DH / 16;
    __shared__ __attribute__((aligned(16))) T smem[2 * CHUNK * LDI];
    T* Kimg = smem;
    T* Vimg = smem + CHUNK * LDI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
    const int prob = blockIdx.y, h = prob % heads, bf = prob / heads;
    const int inner = heads * DH;
    const long ld = 3L * inner;
    const T* base = qkv + (long)bf * P * ld;
    const T* qp = base + h * DH;
    const T* kp = base + inner + h * DH;
    const T* vp = base + 2 * inner + h * DH;
    const int q0 = blockIdx.x * 128 + wave * 16 * U;
    const bool active = q0 < P;
    const float c = scale * LOG2E;

    typename Mma<T>::frag qf[U][KS];
    long q8[U][KS];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[u][ks] = row_frag<T>(qp, ld, q0 + 16 * u + r, P, 32 * ks + 8 * g);
            if constexpr (FP8) q8[u][ks] = to_fp8(qf[u][ks]);
        }

    f32x4 o[DT][U];
    float m_run[U], l_run[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        m_run[u] = -INFINITY; l_run[u] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt][u] = f32x4{0, 0, 0, 0};
    }

    // One chunk of 128 keys.  TAIL = the chunk that holds the row end: keys >= P are masked and tiles wholly past P
    // skipped there; every other chunk runs the lean straight-line form.  The softmax is the kernel's critical
    // resource (vector ALU, the exponential at quarter rate): the row maximum is taken on the raw scores (c > 0)
    // and the scale rides in the exponent's FMA.
    auto chunk = [&](const int c0, auto tail_c) {
        constexpr bool TAIL = decltype(tail_c)::value;
        f32x4 s[NT][U];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int u = 0; u < U; ++u) s[t][u] = f32x4{0, 0, 0, 0};
            if (!TAIL || c0 + 16 * t < P) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    typename Mma<T>::frag kf = frag_load(Kimg + (16 * t + r) * LDI + 32 * ks + 8 * g);
                    if constexpr (FP8) {
                        const long k8 = to_fp8(kf);
#pragma unroll
                        for (int u = 0; u < U; ++u) mma8(s[t][u], k8, q8[u][ks]);
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u) Mma<T>::mma(s[t][u], kf, qf[u][ks]);
                    }
                }
            }
        }
        // online softmax in the log2 domain; lane owns query r of sub-tile u, keys 4g+j of each tile
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (TAIL) s[t][u][j] = (c0 + 16 * t + 4 * g + j < P) ? s[t][u][j] : -INFINITY;
                    mx = fmaxf(mx, s[t][u][j]);
                }
            mx = group_max(mx) * c;
            const float m_new = fmaxf(m_run[u], mx);
            const float alpha = fast_exp2(m_run[u] - m_new);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float pv = fast_exp2(fmaf(s[t][u][j], c, -m_new));
                    s[t][u][j] = pv;
                    sum += pv;
                }
            sum = group_sum(sum);
            l_run[u] = l_run[u] * alpha + sum;
            m_run[u] = m_new;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt][u] *= alpha;
        }
        // O^T += V^T P^T
#pragma unroll
        for (int ss = 0; ss < NT / 2; ++ss) {
            if (!TAIL || c0 + 32 * ss < P) {
                typename Mma<T>::frag pf[U];
                long p8[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if constexpr (FP8) {
                        const f32x4 &lo = s[2 * ss][u], &hi = s[2 * ss + 1][u];
                        const float pv[8] = {lo[0] * P8_SCALE, lo[1] * P8_SCALE, lo[2] * P8_SCALE, lo[3] * P8_SCALE,
                                             hi[0] * P8_SCALE, hi[1] * P8_SCALE, hi[2] * P8_SCALE, hi[3] * P8_SCALE};
                        p8[u] = pack_fp8(pv);
                    } else {
                        pf[u] = acc_frag<T>(s[2 * ss][u], s[2 * ss + 1][u]);
                    }
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    typename Mma<T>::frag vf =
                        frag_load_tr(Vimg, LDI, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, 16 * dt, r);
                    if constexpr (FP8) {
                        const long v8 = to_fp8(vf);
#pragma unroll
                        for (int u = 0; u < U; ++u) mma8(o[dt][u], v8, p8[u]);
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u) Mma<T>::mma(o[dt][u], vf, pf[u]);
                    }
                }
            }
        }
    };
    StageRegs<T, DH, 512 / U> kreg, vreg;
    stage_fetch(kreg, kp, ld, 0, P, tid);
    stage_fetch(vreg, vp, ld, 0, P, tid);
    for (int c0 = 0; c0 < P; c0 += CHUNK) {
        if (c0) __syncthreads();
        stage_commit(Kimg, kreg, tid);
        stage_commit(Vimg, vreg, tid);
        __syncthreads();
        if (c0 + CHUNK < P) {                       // next chunk's rows: in flight under this chunk's MFMAs
            stage_fetch(kreg, kp, ld, c0 + CHUNK, P, tid);
            stage_fetch(vreg, vp, ld, c0 + CHUNK, P, tid);
        }
        if (!active) continue;
        if (c0 + CHUNK > P) chunk(c0, std::true_type{});
        else chunk(c0, std::false_type{});
    }
    if (!active) return;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = q0 + 16 * u + r;
        if (q >= P) continue;
        const float inv = 1.0f / l_run[u];
        const float oinv = FP8 ? inv * (1.0f / P8_SCALE) : inv;
        T* op = out + ((long)bf * P + q) * inner + h * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            float v[4] = {o[dt][u][0] * oinv, o[dt][u][1] * oinv, o[dt][u][2] * oinv, o[dt][u][3] * oinv};
            store4(op + 16 * dt + 4 * g, v);
        }
        if (g == 0) {
            // softmax statistics for backward: running max (log2 domain) and 1/sum, kept separate so
            // that recomputed probabilities still sum to 1 to fp32 rounding (a fused log-sum-exp
            // loses eps*|lse|, which p*(dP - delta) amplifies for peaked rows)
            float2* st = reinterpret_cast<float2*>(lse) + ((long)bf * P + q) * heads + h;
            *st = make_float2(m_run[u], inv);
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward part 1: delta = rowsum(dO o O), dQ
template <typename T, int DH, int U, bool FP8 = false>
__global__ __launch_bounds__(512 / U) void sattn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                           const T* __restrict__ dout, const float* __restrict__ lse,
                                                           float* __restrict__ delta, T* __restrict__ dqkv, int P,
                                                           int heads, float scale) {
    constexpr int LDI = DH + IPAD, KS = DH / 32, DT = DH / 16;
    __shared__ __attribute__((aligned(16))) T smem[2 * CHUNK * LDI];
    T* Kimg = smem;
    T* Vimg = smem + CHUNK * LDI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
    const int prob = blockIdx.y, h = prob % heads, bf = prob / heads;
    const int inner = heads * DH;
    const long ld = 3L * inner;
    const T* base = qkv + (long)bf * P * ld;
    const T* qp = base + h * DH;
    const T* kp = base + inner + h * DH;
    const T* vp = base + 2 * inner + h * DH;
    const T* op = out + (long)bf * P * inner + h * DH;
    const T* dop = dout + (long)bf * P * inner + h * DH;
    const int q0 = blockIdx.x * 128 + wave * 16 * U;
    const bool active = q0 < P;
    const float c = scale * LOG2E;

    typename Mma<T>::frag qf[U][KS], dof[U][KS];
    long q8[U][KS];
    float lq[U], li[U], dl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = q0 + 16 * u + r;
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[u][ks] = row_frag<T>(qp, ld, q, P, 32 * ks + 8 * g);
            if constexpr (FP8) q8[u][ks] = to_fp8(qf[u][ks]);        // S is recomputed exactly as the forward computed it
            dof[u][ks] = row_frag<T>(dop, inner, q, P, 32 * ks + 8 * g);
            if (q < P) {
                float a[8], b[8];
                load8(dop + (long)q * inner + 32 * ks + 8 * g, a);
                load8(op + (long)q * inner + 32 * ks + 8 * g, b);
#pragma unroll
                for (int i = 0; i < 8; ++i) part += a[i] * b[i];
            }
        }
        dl[u] = group_sum(part);
        const float2 st = q < P ? reinterpret_cast<const float2*>(lse)[((long)bf * P + q) * heads + h] : make_float2(0.f, 0.f);
        // exponent offset with 1/rowsum folded in: p = exp2(s c - (max - log2(1/sum))); rows past P have 1/sum = 0 ->
        // offset +inf -> p = 0
        lq[u] = st.x - __builtin_amdgcn_logf(st.y); li[u] = st.y;
        if (q < P && g == 0) delta[((long)bf * P + q) * heads + h] = dl[u];
    }

    f32x4 dq[DT][U];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int u = 0; u < U; ++u) dq[dt][u] = f32x4{0, 0, 0, 0};

    StageRegs<T, DH, 512 / U> kreg, vreg;
    stage_fetch(kreg, kp, ld, 0, P, tid);
    stage_fetch(vreg, vp, ld, 0, P, tid);
    for (int c0 = 0; c0 < P; c0 += CHUNK) {
        if (c0) __syncthreads();
        stage_commit(Kimg, kreg, tid);
        stage_commit(Vimg, vreg, tid);
        __syncthreads();
        if (c0 + CHUNK < P) {
            stage_fetch(kreg, kp, ld, c0 + CHUNK, P, tid);
            stage_fetch(vreg, vp, ld, c0 + CHUNK, P, tid);
        }
        if (!active) continue;
        const bool tail = c0 + CHUNK > P;
#pragma unroll
        for (int ss = 0; ss < NT / 2; ++ss) {
            if (c0 + 32 * ss >= P) continue;
            f32x4 s[2][U], dp[2][U];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int u = 0; u < U; ++u) { s[tt][u] = f32x4{0, 0, 0, 0}; dp[tt][u] = f32x4{0, 0, 0, 0}; }
                const int krow = 32 * ss + 16 * tt + r;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    typename Mma<T>::frag kf = frag_load(Kimg + krow * LDI + 32 * ks + 8 * g);
                    typename Mma<T>::frag vf = frag_load(Vimg + krow * LDI + 32 * ks + 8 * g);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if constexpr (FP8) mma8(s[tt][u], to_fp8(kf), q8[u][ks]);
                        else Mma<T>::mma(s[tt][u], kf, qf[u][ks]);
                        Mma<T>::mma(dp[tt][u], vf, dof[u][ks]);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float pv = fast_exp2(fmaf(s[tt][u][j], c, -lq[u]));
                        if (tail && c0 + 32 * ss + 16 * tt + 4 * g + j >= P) pv = 0.f;
                        s[tt][u][j] = pv * (dp[tt][u][j] - dl[u]);               // dS^T / scale (applied to dQ below)
                    }
            }
            typename Mma<T>::frag dsf[U];
#pragma unroll
            for (int u = 0; u < U; ++u) dsf[u] = acc_frag<T>(s[0][u], s[1][u]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                typename Mma<T>::frag kt = frag_load_tr(Kimg, LDI, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, 16 * dt, r);
#pragma unroll
                for (int u = 0; u < U; ++u) Mma<T>::mma(dq[dt][u], kt, dsf[u]);
            }
        }
    }
    if (!active) return;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = q0 + 16 * u + r;
        if (q >= P) continue;
        T* dqp = dqkv + ((long)bf * P + q) * ld + h * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            float v[4] = {dq[dt][u][0] * scale, dq[dt][u][1] * scale, dq[dt][u][2] * scale, dq[dt][u][3] * scale};
            store4(dqp + 16 * dt + 4 * g, v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward part 2: dK, dV.  Wave owns 32 keys; queries stream through LDS.
template <typename T, int DH, int U, bool FP8 = false>
__global__ __launch_bounds__(512 / U) void sattn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                            const float* __restrict__ lse,
                                                            const float* __restrict__ delta, T* __restrict__ dqkv,
                                                            int P, int heads, float scale) {
    constexpr int LDI = DH + IPAD, KS = DH / 32, DT = DH / 16;
    __shared__ __attribute__((aligned(16))) T smem[2 * CHUNK * LDI];
    __shared__ __attribute__((aligned(16))) float stat[2][CHUNK];
    T* Qimg = smem;
    T* Dimg = smem + CHUNK * LDI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
    const int prob = blockIdx.y, h = prob % heads, bf = prob / heads;
    const int inner = heads * DH;
    const long ld = 3L * inner;
    const T* base = qkv + (long)bf * P * ld;
    const T* qp = base + h * DH;
    const T* kp = base + inner + h * DH;
    const T* vp = base + 2 * inner + h * DH;
    const T* dop = dout + (long)bf * P * inner + h * DH;
    const int k0 = blockIdx.x * 128 + wave * 16 * U;
    const bool active = k0 < P;
    const float c = scale * LOG2E;

    typename Mma<T>::frag kf[U][KS], vf[U][KS];
    long k8[U][KS];
#pragma unroll
    for (int kt = 0; kt < U; ++kt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[kt][ks] = row_frag<T>(kp, ld, k0 + 16 * kt + r, P, 32 * ks + 8 * g);
            if constexpr (FP8) k8[kt][ks] = to_fp8(kf[kt][ks]);
            vf[kt][ks] = row_frag<T>(vp, ld, k0 + 16 * kt + r, P, 32 * ks + 8 * g);
        }
    f32x4 dk[DT][U], dv[DT][U];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int kt = 0; kt < U; ++kt) { dk[dt][kt] = f32x4{0, 0, 0, 0}; dv[dt][kt] = f32x4{0, 0, 0, 0}; }

    StageRegs<T, DH, 512 / U> qreg, dreg;
    float2 streg = make_float2(0.f, 0.f);
    float dlreg = 0.f;
    auto fetch = [&](int c0) {
        stage_fetch(qreg, qp, ld, c0, P, tid);
        stage_fetch(dreg, dop, (long)inner, c0, P, tid);
        if (tid < CHUNK) {
            const int q = c0 + tid;
            streg = q < P ? reinterpret_cast<const float2*>(lse)[((long)bf * P + q) * heads + h]
                          : make_float2(0.f, 0.f);                  // 1/l = 0 masks padded query rows
            dlreg = q < P ? delta[((long)bf * P + q) * heads + h] : 0.f;
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < P; c0 += CHUNK) {
        if (c0) __syncthreads();
        stage_commit(Qimg, qreg, tid);
        stage_commit(Dimg, dreg, tid);
        if (tid < CHUNK) {
            stat[0][tid] = streg.x - __builtin_amdgcn_logf(streg.y);    // exponent offset incl. log2(1/sum); +inf for padding
            stat[1][tid] = dlreg;
        }
        __syncthreads();
        if (c0 + CHUNK < P) fetch(c0 + CHUNK);
        if (!active) continue;
#pragma unroll
        for (int ss = 0; ss < NT / 2; ++ss) {
            if (c0 + 32 * ss >= P) continue;
            f32x4 s[2][U], dp[2][U];      // [query tile tt][key tile kt]; rows = queries 4g+j, cols = keys r
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int kt = 0; kt < U; ++kt) { s[tt][kt] = f32x4{0, 0, 0, 0}; dp[tt][kt] = f32x4{0, 0, 0, 0}; }
                const int qrow = 32 * ss + 16 * tt + r;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    typename Mma<T>::frag qa = frag_load(Qimg + qrow * LDI + 32 * ks + 8 * g);
                    typename Mma<T>::frag da = frag_load(Dimg + qrow * LDI + 32 * ks + 8 * g);
#pragma unroll
                    for (int kt = 0; kt < U; ++kt) {
                        if constexpr (FP8) mma8(s[tt][kt], to_fp8(qa), k8[kt][ks]);
                        else Mma<T>::mma(s[tt][kt], qa, kf[kt][ks]);
                        Mma<T>::mma(dp[tt][kt], da, vf[kt][ks]);
                    }
                }
                const int qb = 32 * ss + 16 * tt + 4 * g;
                const float4 l4 = *reinterpret_cast<const float4*>(&stat[0][qb]);
                const float4 d4 = *reinterpret_cast<const float4*>(&stat[1][qb]);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int kt = 0; kt < U; ++kt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float pv = fast_exp2(fmaf(s[tt][kt][j], c, -lv[j]));
                        s[tt][kt][j] = pv;                                         // P
                        dp[tt][kt][j] = pv * (dp[tt][kt][j] - dv4[j]);             // dS / scale (applied to dK below)
                    }
            }
            typename Mma<T>::frag pf[U], dsf[U];
#pragma unroll
            for (int kt = 0; kt < U; ++kt) { pf[kt] = acc_frag<T>(s[0][kt], s[1][kt]); dsf[kt] = acc_frag<T>(dp[0][kt], dp[1][kt]); }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                typename Mma<T>::frag dot_ = frag_load_tr(Dimg, LDI, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, 16 * dt, r);
                typename Mma<T>::frag qt_ = frag_load_tr(Qimg, LDI, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, 16 * dt, r);
#pragma unroll
                for (int kt = 0; kt < U; ++kt) {
                    Mma<T>::mma(dv[dt][kt], dot_, pf[kt]);
                    Mma<T>::mma(dk[dt][kt], qt_, dsf[kt]);
                }
            }
        }
    }
    if (!active) return;
#pragma unroll
    for (int kt = 0; kt < U; ++kt) {
        const int key = k0 + 16 * kt + r;
        if (key >= P) continue;
        T* dkp = dqkv + ((long)bf * P + key) * ld + inner + h * DH;
        T* dvp = dqkv + ((long)bf * P + key) * ld + 2 * inner + h * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            float a[4] = {dk[dt][kt][0] * scale, dk[dt][kt][1] * scale, dk[dt][kt][2] * scale, dk[dt][kt][3] * scale};
            float b[4] = {dv[dt][kt][0], dv[dt][kt][1], dv[dt][kt][2], dv[dt][kt][3]};
            store4(dkp + 16 * dt + 4 * g, a);
            store4(dvp + 16 * dt + 4 * g, b);
        }
    }
}

// ------------------------------------------------------------------------------------------
#define DISPATCH_DH(DHV, ...)                                   \
    do {                                                        \
        if ((DHV) == 64) { constexpr int DH = 64; __VA_ARGS__; } \
        else if ((DHV) == 32) { constexpr int DH = 32; __VA_ARGS__; } \
        else return ISTVT_ERR_SHAPE;                            \
    } while (0)

// qkv: [BF*P][3*heads*dh]; out: [BF*P][heads*dh]; lse: [BF*P][heads][2] = (row max in the log2 domain, 1/rowsum)
extern "C" int istvt_attn_spatial_fwd(const void* qkv, void* out, float* lse, int BF, int P, int heads, int dh,
                                      float scale, int dtype, hipStream_t stream) {
    if (BF <= 0 || P <= 0 || heads <= 0) return ISTVT_ERR_SHAPE;
    dim3 grid((P + 127) / 128, BF * heads);
    static const int u1 = getenv("ISTVT_SATTN_U") ? atoi(getenv("ISTVT_SATTN_U")) : 1;
    // bf16: 8 wavefronts x 16 queries (more wavefronts per SIMD); fp32 keeps 4 x 32 (its LDS image fills the CU)
    if (dtype == DT_BF16 && u1 == 1) {
        DISPATCH_DH(dh, hipLaunchKernelGGL((sattn_fwd_kernel<bf16_t, DH, 1>), grid, dim3(512), 0, stream,
                                           (const bf16_t*)qkv, (bf16_t*)out, lse, P, heads, scale));
        return istvt_check_launch();
    }
    dim3 block(256);
    DISPATCH_DTYPE(dtype, DISPATCH_DH(dh, hipLaunchKernelGGL((sattn_fwd_kernel<T, DH, 2>), grid, block, 0, stream,
                                                             (const T*)qkv, (T*)out, lse, P, heads, scale)));
    return istvt_check_launch();
}

// delta: scratch [BF*P][heads] fp32 (written here, consumed by the dK/dV kernel); dqkv: [BF*P][3*inner]
extern "C" int istvt_attn_spatial_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                                      float* delta, void* dqkv, int BF, int P, int heads, int dh, float scale,
                                      int dtype, hipStream_t stream) {
    if (BF <= 0 || P <= 0 || heads <= 0) return ISTVT_ERR_SHAPE;
    dim3 grid((P + 127) / 128, BF * heads);
    static const int u1 = getenv("ISTVT_SATTN_U") ? atoi(getenv("ISTVT_SATTN_U")) : 1;
    if (dtype == DT_BF16 && u1 == 1) {          // 8 wavefronts x 16 rows, see sattn_fwd_kernel
        DISPATCH_DH(dh, {
            hipLaunchKernelGGL((sattn_bwd_dq_kernel<bf16_t, DH, 1>), grid, dim3(512), 0, stream, (const bf16_t*)qkv,
                               (const bf16_t*)out, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, P, heads, scale);
            hipLaunchKernelGGL((sattn_bwd_dkv_kernel<bf16_t, DH, 1>), grid, dim3(512), 0, stream, (const bf16_t*)qkv,
                               (const bf16_t*)dout, lse, (const float*)delta, (bf16_t*)dqkv, P, heads, scale);
        });
        return istvt_check_launch();
    }
    dim3 block(256);
    DISPATCH_DTYPE(dtype, DISPATCH_DH(dh, {
        hipLaunchKernelGGL((sattn_bwd_dq_kernel<T, DH, 2>), grid, block, 0, stream, (const T*)qkv, (const T*)out,
                           (const T*)dout, lse, delta, (T*)dqkv, P, heads, scale);
        hipLaunchKernelGGL((sattn_bwd_dkv_kernel<T, DH, 2>), grid, block, 0, stream, (const T*)qkv, (const T*)dout, lse,
                           (const float*)delta, (T*)dqkv, P, heads, scale);
    }));
    return istvt_check_launch();
}

// ---- fp8 variant (bfloat16 storage only): Q, K, V and the probabilities enter the attention MFMAs as OCP e4m3
// (v_mfma_f32_16x16x32_fp8_fp8); softmax, statistics, accumulation and every other product stay as above.  The
// backward recomputes S with the same fp8 operands, so the saved (max, 1/sum) match its probabilities.
extern "C" int istvt_attn_spatial_fwd_fp8(const void* qkv, void* out, float* lse, int BF, int P, int heads, int dh,
                                          float scale, int dtype, hipStream_t stream) {
    if (BF <= 0 || P <= 0 || heads <= 0) return ISTVT_ERR_SHAPE;
    if (dtype != DT_BF16) return ISTVT_ERR_DTYPE;
    dim3 grid((P + 127) / 128, BF * heads);
    DISPATCH_DH(dh, hipLaunchKernelGGL((sattn_fwd_kernel<bf16_t, DH, 1, true>), grid, dim3(512), 0, stream,
                                       (const bf16_t*)qkv, (bf16_t*)out, lse, P, heads, scale));
    return istvt_check_launch();
}

extern "C" int istvt_attn_spatial_bwd_fp8(const void* qkv, const void* out, const void* dout, const float* lse,
                                          float* delta, void* dqkv, int BF, int P, int heads, int dh, float scale,
                                          int dtype, hipStream_t stream) {
    if (BF <= 0 || P <= 0 || heads <= 0) return ISTVT_ERR_SHAPE;
    if (dtype != DT_BF16) return ISTVT_ERR_DTYPE;
    dim3 grid((P + 127) / 128, BF * heads);
    DISPATCH_DH(dh, {
        hipLaunchKernelGGL((sattn_bwd_dq_kernel<bf16_t, DH, 1, true>), grid, dim3(512), 0, stream, (const bf16_t*)qkv,
                           (const bf16_t*)out, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, P, heads, scale);
        hipLaunchKernelGGL((sattn_bwd_dkv_kernel<bf16_t, DH, 1, true>), grid, dim3(512), 0, stream, (const bf16_t*)qkv,
                           (const bf16_t*)dout, lse, (const float*)delta, (bf16_t*)dqkv, P, heads, scale);
    });
    return istvt_check_launch();
}
