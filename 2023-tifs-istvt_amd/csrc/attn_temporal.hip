// Per-position temporal self-attention over the frame axis (reference:
// TemporalResidualAttention.forward, network/vivit/module.py:197-205).  For every
// (clip b, position p, head h):  O = softmax(Q K^T * DH^-1/2) V with Q,K,V of shape [F][DH],
// F = T+1 <= 17.  diff != 0 (TemporalResidualAttention): the frame differencing of module.py:193 happens HERE, on the
// projected rows.  to_qk has no bias (module.py:182), so to_qk(x[f] - x[f-1]) = to_qk(x[f]) - to_qk(x[f-1]): the caller
// projects the un-differenced LayerNorm output ONCE with [to_qk | to_v] stacked (one GEMM, one activation tensor) and
// the kernels take q'[f] = q[f] - q[f-1], k'[f] = k[f] - k[f-1] for f >= 2 (frames 0, 1 unchanged) in registers -- all
// F frames of a position are in the wavefront anyway.  The backward kernels return the gradient with respect to the
// UN-differenced rows (the adjoint: d q[f] = d q'[f] - d q'[f+1] for f + 1 >= 2), so one input-gradient GEMM follows.
//
// 0.1 % of the model's FLOPs and F x F tiles far below an MFMA tile: this is a data-movement
// kernel, bound by HBM.  A cluster of CL lanes owns one (b,p,h); each lane keeps EPL = DH/CL
// consecutive head-dim elements of every frame's k/v row in registers, so a row segment
// (DH elements) is one fully-coalesced cluster access, dot products finish with log2(CL) DPP /
// swizzle adds, no LDS.
//   F <= 9  : CL = DH/4 (4 elements per lane), every row of q,k,v(,dO) in registers (tattn_*_kernel)
//   F <= 17 : CL = DH/2 (2 elements per lane) and the q / dO rows are fetched one frame ahead (tattn_*2_kernel)
//             instead of held: with 4 elements per lane the F = 17 kernels needed 356 registers
//             (560 bytes of scratch per lane, one wavefront per SIMD) and ran 13x slower for 2x
//             the data (T = 16, BASELINE config 4).
//
// qk : [B*F*P][2*inner]  (q | k), v : [B*F*P][inner], out : [B*F*P][inner], rows ordered (b,f,p).
// Nothing is saved for backward: the F x F probabilities are recomputed from q,k.
#include "common.h"
#include <cstdlib>
#include "attn_temporal_mfma.h"

template <int CL> __device__ __forceinline__ float cluster_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    if (CL >= 16)
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));  // row_mirror
    if (CL == 32)   // lanes i <-> i ^ 16 (ds_swizzle bit mode: and 0x1f, or 0, xor 0x10)
        v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
    return v;
}

__device__ __forceinline__ float dot4(const float (&a)[4], const float (&b)[4]) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}

// q'[f] = q[f] - q[f-1], k'[f] = k[f] - k[f-1] for f >= 2, in place on rows held in registers (top frame first)
template <int FMAX, int EPL>
__device__ __forceinline__ void frame_diff_rows(float (&q)[FMAX][EPL], float (&k)[FMAX][EPL], const int F) {
#pragma unroll
    for (int f = FMAX - 1; f >= 2; --f) {
        if (f < F) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) { q[f][e] -= q[f - 1][e]; k[f][e] -= k[f - 1][e]; }
        }
    }
}
template <int FMAX, int EPL>
__device__ __forceinline__ void frame_diff_rows1(float (&k)[FMAX][EPL], const int F) {
#pragma unroll
    for (int f = FMAX - 1; f >= 2; --f) {
        if (f < F) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) k[f][e] -= k[f - 1][e];
        }
    }
}
// the adjoint on the key gradients: d k[f] = d k'[f] - d k'[f+1] for f + 1 >= 2 (in place, bottom frame first)
template <int FMAX, int EPL>
__device__ __forceinline__ void frame_diff_adjoint_rows(float (&dk)[FMAX][EPL], const int F) {
#pragma unroll
    for (int f = 1; f + 1 < FMAX; ++f) {
        if (f + 1 < F) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) dk[f][e] -= dk[f + 1][e];
        }
    }
}

template <typename T, int DH, int FMAX>
__global__ __launch_bounds__(256) void tattn_fwd_kernel(const T* __restrict__ qk, const T* __restrict__ v,
                                                        T* __restrict__ out, int B, int F, int P, int heads,
                                                        float scale, long ldqk, long ldv, long ldo, int diff) {
    constexpr int CL = DH / 4, GW = 64 / CL;
    const int lane = threadIdx.x & 63;
    const long ngroups = (long)B * P * heads;
    long gid = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * GW + lane / CL;
    const bool valid = gid < ngroups;
    if (!valid) gid = ngroups - 1;            // keep every lane alive for the DPP reductions
    const int cl = lane % CL;
    const int h = (int)(gid % heads);
    const long bp = gid / heads;
    const long b = bp / P, p = bp % P;
    const int inner = heads * DH;
    const long row0 = b * F * P + p;          // frame f lives at row0 + f*P
    const int col = h * DH + cl * 4;

    float q[FMAX][4], k[FMAX][4], vv[FMAX][4];
#pragma unroll
    for (int f = 0; f < FMAX; ++f) {
        if (f < F) {
            const long m = row0 + (long)f * P;
            load4(qk + m * ldqk + col, q[f]);
            load4(qk + m * ldqk + inner + col, k[f]);
            load4(v + m * ldv + col, vv[f]);
        }
    }
    if (diff) frame_diff_rows<FMAX, 4>(q, k, F);
#pragma unroll
    for (int i = 0; i < FMAX; ++i) {
        if (i < F) {
            float s[FMAX];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) {
                    s[j] = cluster_sum<CL>(dot4(q[i], k[j])) * scale;
                    mx = fmaxf(mx, s[j]);
                }
            }
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) { s[j] = __expf(s[j] - mx); sum += s[j]; }
            }
            const float inv = 1.0f / sum;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) {
                    const float pj = s[j] * inv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += pj * vv[j][e];
                }
            }
            if (valid) {
                const long m = row0 + (long)i * P;
                store4(out + m * ldo + col, o);
            }
        }
    }
}

// Softmax backward without the cancellation of  dp_j - sum_l p_l dp_l : with sum_l p_l = 1 that difference is
// sum_l p_l (dp_j - dp_l), a sum of terms that are each small when the softmax is peaked (p_max -> 1 makes dp_max - delta
// the difference of two nearly equal numbers whose rounding errors, ~eps |dO||V|, are then amplified by 1 / (1 - p_max):
// with the golden recipe's saturated temporal softmaxes that was per cents of the to_qk gradients).  F <= 17 keys:
// F^2 subtractions per row are nothing next to the loads.
template <int FMAX>
__device__ __forceinline__ float softmax_bwd_pairwise(const float (&pr)[FMAX], const float (&dp)[FMAX], const int j, const int F) {
    float acc = 0.f;
#pragma unroll
    for (int l = 0; l < FMAX; ++l) {
        if (l < F) acc += pr[l] * (dp[j] - dp[l]);
    }
    return acc;
}

// backward: dqk [B*F*P][2*inner] (dq | dk), dv [B*F*P][inner]
template <typename T, int DH, int FMAX>
__global__ __launch_bounds__(256) void tattn_bwd_kernel(const T* __restrict__ qk, const T* __restrict__ v,
                                                        const T* __restrict__ dout, T* __restrict__ dqk,
                                                        T* __restrict__ dv, int B, int F, int P, int heads,
                                                        float scale, long ldqk, long ldv, long ldo, int diff) {
    constexpr int CL = DH / 4, GW = 64 / CL;
    const int lane = threadIdx.x & 63;
    const long ngroups = (long)B * P * heads;
    long gid = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * GW + lane / CL;
    const bool valid = gid < ngroups;
    if (!valid) gid = ngroups - 1;
    const int cl = lane % CL;
    const int h = (int)(gid % heads);
    const long bp = gid / heads;
    const long b = bp / P, p = bp % P;
    const int inner = heads * DH;
    const long row0 = b * F * P + p;
    const int col = h * DH + cl * 4;

    float q[FMAX][4], k[FMAX][4], vv[FMAX][4], dO[FMAX][4], dk[FMAX][4], dvv[FMAX][4];
#pragma unroll
    for (int f = 0; f < FMAX; ++f) {
        if (f < F) {
            const long m = row0 + (long)f * P;
            load4(qk + m * ldqk + col, q[f]);
            load4(qk + m * ldqk + inner + col, k[f]);
            load4(v + m * ldv + col, vv[f]);
            load4(dout + m * ldo + col, dO[f]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { dk[f][e] = 0.f; dvv[f][e] = 0.f; }
        }
    }
    if (diff) frame_diff_rows<FMAX, 4>(q, k, F);
    float dq_held[4] = {0.f, 0.f, 0.f, 0.f};         // diff: d q'[i-1], stored once d q'[i] is known
#pragma unroll
    for (int i = 0; i < FMAX; ++i) {
        if (i < F) {
            const long m = row0 + (long)i * P;
            // probabilities are recomputed exactly as the forward computes them (max, exp, sum,
            // divide): a saved log-sum-exp would leave sum(p) != 1 by eps*|lse| and that error is
            // amplified in p*(dp - delta) when the softmax is peaked.
            float pr[FMAX], dp[FMAX];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) {
                    pr[j] = cluster_sum<CL>(dot4(q[i], k[j])) * scale;
                    mx = fmaxf(mx, pr[j]);
                    dp[j] = cluster_sum<CL>(dot4(dO[i], vv[j]));
                }
            }
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) { pr[j] = __expf(pr[j] - mx); sum += pr[j]; }
            }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) pr[j] *= inv;
            }
            float dq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) {
                    const float ds = pr[j] * softmax_bwd_pairwise<FMAX>(pr, dp, j, F) * scale;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dq[e] += ds * k[j][e];
                        dk[j][e] += ds * q[i][e];
                        dvv[j][e] += pr[j] * dO[i][e];
                    }
                }
            }
            if (!diff) {
                if (valid) store4(dqk + m * ldqk + col, dq);
            } else {
                if (i >= 1) {
                    if (i >= 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) dq_held[e] -= dq[e];
                    }
                    if (valid) store4(dqk + (m - P) * ldqk + col, dq_held);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) dq_held[e] = dq[e];
            }
        }
    }
    if (diff) {
        if (valid) store4(dqk + (row0 + (long)(F - 1) * P) * ldqk + col, dq_held);
        frame_diff_adjoint_rows<FMAX, 4>(dk, F);
    }
    if (valid) {
#pragma unroll
        for (int f = 0; f < FMAX; ++f) {
            if (f < F) {
                const long m = row0 + (long)f * P;
                store4(dqk + m * ldqk + inner + col, dk[f]);
                store4(dv + m * ldv + col, dvv[f]);
            }
        }
    }
}

// ---- F <= 17: two elements per lane, q / dO rows fetched one frame ahead ---------------------------------
// EPL (2 or 4) consecutive elements
template <int EPL> __device__ __forceinline__ void loadE(const float* p, float (&v)[EPL]) {
    if constexpr (EPL == 4) { const float4 a = *reinterpret_cast<const float4*>(p); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
    else { const float2 a = *reinterpret_cast<const float2*>(p); v[0] = a.x; v[1] = a.y; }
}
template <int EPL> __device__ __forceinline__ void loadE(const bf16_t* p, float (&v)[EPL]) {
    if constexpr (EPL == 4) {
        const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
    } else {
        const unsigned u = *reinterpret_cast<const unsigned*>(p);
        v[0] = __uint_as_float(u << 16);
        v[1] = __uint_as_float(u & 0xffff0000u);
    }
}
template <int EPL> __device__ __forceinline__ void storeE(float* p, const float (&v)[EPL]) {
    if constexpr (EPL == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
}
template <int EPL> __device__ __forceinline__ void storeE(bf16_t* p, const float (&v)[EPL]) {
    if constexpr (EPL == 4) {
        bf16x4 a;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = (bf16_t)v[i];
        *reinterpret_cast<bf16x4*>(p) = a;
    } else {
        typedef bf16_t bf16x2v __attribute__((ext_vector_type(2)));
        bf16x2v a;
        a[0] = (bf16_t)v[0]; a[1] = (bf16_t)v[1];
        *reinterpret_cast<bf16x2v*>(p) = a;
    }
}
template <int EPL> __device__ __forceinline__ float dotE(const float (&a)[EPL], const float (&b)[EPL]) {
    float s = a[0] * b[0];
#pragma unroll
    for (int e = 1; e < EPL; ++e) s += a[e] * b[e];
    return s;
}

template <typename T, int DH, int FMAX, int EPL>
__global__ __launch_bounds__(256) void tattn_fwd2_kernel(const T* __restrict__ qk, const T* __restrict__ v,
                                                        T* __restrict__ out, int B, int F, int P, int heads,
                                                        float scale, long ldqk, long ldv, long ldo, int diff) {
    constexpr int CL = DH / EPL, GW = 64 / CL;
    const int lane = threadIdx.x & 63;
    const long ngroups = (long)B * P * heads;
    long gid = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * GW + lane / CL;
    const bool valid = gid < ngroups;
    if (!valid) gid = ngroups - 1;            // keep every lane alive for the DPP reductions
    const int cl = lane % CL;
    const int h = (int)(gid % heads);
    const long bp = gid / heads;
    const long b = bp / P, p = bp % P;
    const int inner = heads * DH;
    const long row0 = b * F * P + p;          // frame f lives at row0 + f*P
    const int col = h * DH + cl * EPL;

    float k[FMAX][EPL], vv[FMAX][EPL];
#pragma unroll
    for (int f = 0; f < FMAX; ++f) {
        if (f < F) {
            const long m = row0 + (long)f * P;
            loadE<EPL>(qk + m * ldqk + inner + col, k[f]);
            loadE<EPL>(v + m * ldv + col, vv[f]);
        }
    }
    if (diff) frame_diff_rows1<FMAX, EPL>(k, F);
    float qn[EPL], qraw[EPL];
    loadE<EPL>(qk + row0 * ldqk + col, qn);
#pragma unroll
    for (int e = 0; e < EPL; ++e) qraw[e] = 0.f;
#pragma unroll
    for (int i = 0; i < FMAX; ++i) {
        if (i < F) {
            float q[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                q[e] = (diff && i >= 2) ? qn[e] - qraw[e] : qn[e];
                qraw[e] = qn[e];
            }
            if (i + 1 < F) loadE<EPL>(qk + (row0 + (long)(i + 1) * P) * ldqk + col, qn);     // one frame ahead
            float s[FMAX];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) {
                    s[j] = cluster_sum<CL>(dotE<EPL>(q, k[j])) * scale;
                    mx = fmaxf(mx, s[j]);
                }
            }
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) { s[j] = __expf(s[j] - mx); sum += s[j]; }
            }
            const float inv = 1.0f / sum;
            float o[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[e] = 0.f;
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) {
                    const float pj = s[j] * inv;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) o[e] += pj * vv[j][e];
                }
            }
            if (valid) {
                const long m = row0 + (long)i * P;
                storeE<EPL>(out + m * ldo + col, o);
            }
        }
    }
}

// backward: dqk [B*F*P][2*inner] (dq | dk), dv [B*F*P][inner]
// HOLD: every q / dO row in registers (F <= 9); otherwise they are fetched one frame ahead of their use
template <typename T, int DH, int FMAX, int EPL, bool HOLD = false>
__global__ __launch_bounds__(256) void tattn_bwd2_kernel(const T* __restrict__ qk, const T* __restrict__ v,
                                                        const T* __restrict__ dout, T* __restrict__ dqk,
                                                        T* __restrict__ dv, int B, int F, int P, int heads,
                                                        float scale, long ldqk, long ldv, long ldo, int diff) {
    constexpr int CL = DH / EPL, GW = 64 / CL;
    const int lane = threadIdx.x & 63;
    const long ngroups = (long)B * P * heads;
    long gid = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * GW + lane / CL;
    const bool valid = gid < ngroups;
    if (!valid) gid = ngroups - 1;
    const int cl = lane % CL;
    const int h = (int)(gid % heads);
    const long bp = gid / heads;
    const long b = bp / P, p = bp % P;
    const int inner = heads * DH;
    const long row0 = b * F * P + p;
    const int col = h * DH + cl * EPL;

    float k[FMAX][EPL], vv[FMAX][EPL], dk[FMAX][EPL], dvv[FMAX][EPL];
#pragma unroll
    for (int f = 0; f < FMAX; ++f) {
        if (f < F) {
            const long m = row0 + (long)f * P;
            loadE<EPL>(qk + m * ldqk + inner + col, k[f]);
            loadE<EPL>(v + m * ldv + col, vv[f]);
#pragma unroll
            for (int e = 0; e < EPL; ++e) { dk[f][e] = 0.f; dvv[f][e] = 0.f; }
        }
    }
    if (diff) frame_diff_rows1<FMAX, EPL>(k, F);
    constexpr int NH = HOLD ? FMAX : 1;
    float qa[NH][EPL], da[NH][EPL];
    float qn[EPL], don[EPL], qraw[EPL], dq_held[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { qraw[e] = 0.f; dq_held[e] = 0.f; }
    if (HOLD) {
#pragma unroll
        for (int f = 0; f < NH; ++f) {
            if (f < F) {
                loadE<EPL>(qk + (row0 + (long)f * P) * ldqk + col, qa[f]);
                loadE<EPL>(dout + (row0 + (long)f * P) * ldo + col, da[f]);
            }
        }
        if (diff) frame_diff_rows1<NH, EPL>(qa, F);
    } else {
        loadE<EPL>(qk + row0 * ldqk + col, qn);
        loadE<EPL>(dout + row0 * ldo + col, don);
    }
#pragma unroll
    for (int i = 0; i < FMAX; ++i) {
        if (i < F) {
            const long m = row0 + (long)i * P;
            float q[EPL], dO[EPL];
            if (HOLD) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) { q[e] = qa[HOLD ? i : 0][e]; dO[e] = da[HOLD ? i : 0][e]; }
            } else {
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    q[e] = (diff && i >= 2) ? qn[e] - qraw[e] : qn[e];
                    qraw[e] = qn[e];
                    dO[e] = don[e];
                }
                if (i + 1 < F) {                                               // one frame ahead
                    loadE<EPL>(qk + (m + P) * ldqk + col, qn);
                    loadE<EPL>(dout + (m + P) * ldo + col, don);
                }
            }
            // probabilities are recomputed exactly as the forward computes them (max, exp, sum,
            // divide): a saved log-sum-exp would leave sum(p) != 1 by eps*|lse| and that error is
            // amplified in p*(dp - delta) when the softmax is peaked.
            float pr[FMAX], dp[FMAX];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) {
                    pr[j] = cluster_sum<CL>(dotE<EPL>(q, k[j])) * scale;
                    mx = fmaxf(mx, pr[j]);
                    dp[j] = cluster_sum<CL>(dotE<EPL>(dO, vv[j]));
                }
            }
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) { pr[j] = __expf(pr[j] - mx); sum += pr[j]; }
            }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) pr[j] *= inv;
            }
            float dq[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) dq[e] = 0.f;
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) {
                    const float ds = pr[j] * softmax_bwd_pairwise<FMAX>(pr, dp, j, F) * scale;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        dq[e] += ds * k[j][e];
                        dk[j][e] += ds * q[e];
                        dvv[j][e] += pr[j] * dO[e];
                    }
                }
            }
            if (!diff) {
                if (valid) storeE<EPL>(dqk + m * ldqk + col, dq);
            } else {
                if (i >= 1) {
                    if (i >= 2) {
#pragma unroll
                        for (int e = 0; e < EPL; ++e) dq_held[e] -= dq[e];
                    }
                    if (valid) storeE<EPL>(dqk + (m - P) * ldqk + col, dq_held);
                }
#pragma unroll
                for (int e = 0; e < EPL; ++e) dq_held[e] = dq[e];
            }
        }
    }
    if (diff) {
        if (valid) storeE<EPL>(dqk + (row0 + (long)(F - 1) * P) * ldqk + col, dq_held);
        frame_diff_adjoint_rows<FMAX, EPL>(dk, F);
    }
    if (valid) {
#pragma unroll
        for (int f = 0; f < FMAX; ++f) {
            if (f < F) {
                const long m = row0 + (long)f * P;
                storeE<EPL>(dqk + m * ldqk + inner + col, dk[f]);
                storeE<EPL>(dv + m * ldv + col, dvv[f]);
            }
        }
    }
}

#define DISPATCH_TATTN(KERNEL, KERNEL2, ...)                                                             \
    do {                                                                                                 \
        const long ngroups = (long)B * P * heads;                                                        \
        const int epl = F <= 9 ? 4 : 2;                                                                  \
        const int gw = 64 / (dh / epl);                                                                  \
        const long blocks = (ngroups + 4 * gw - 1) / (4 * gw);                                           \
        dim3 grid((unsigned)blocks), block(256);                                                         \
        if (dh == 64 && F <= 9) hipLaunchKernelGGL((KERNEL<T, 64, 9>), grid, block, 0, stream, __VA_ARGS__);           \
        else if (dh == 64 && F <= 17) hipLaunchKernelGGL((KERNEL2<T, 64, 17, 2>), grid, block, 0, stream, __VA_ARGS__); \
        else if (dh == 32 && F <= 9) hipLaunchKernelGGL((KERNEL<T, 32, 9>), grid, block, 0, stream, __VA_ARGS__);      \
        else if (dh == 32 && F <= 17) hipLaunchKernelGGL((KERNEL2<T, 32, 17, 2>), grid, block, 0, stream, __VA_ARGS__); \
        else return ISTVT_ERR_SHAPE;                                                                     \
    } while (0)

extern "C" int istvt_attn_temporal_fwd(const void* qk, long ldqk, const void* v, long ldv, void* out, long ldo, int B, int F,
                                       int P, int heads, int dh, float scale, int diff, int dtype, hipStream_t stream) {
    if (B <= 0 || F <= 0 || P <= 0 || heads <= 0 || diff < 0 || diff > 2) return ISTVT_ERR_SHAPE;
    if (ldqk < 2L * heads * dh || ldv < (long)heads * dh || ldo < (long)heads * dh || ldqk % 8 || ldv % 8 || ldo % 8) return ISTVT_ERR_SHAPE;
    static const int use_mfma = istvt_tune("ISTVT_TATTN_MFMA", 1);
    if (use_mfma && dtype == DT_BF16 && F <= 32 && (dh == 64 || dh == 32)) {     // one wavefront per (b, p, h), MFMA tiles
        const long nprob = (long)B * P * heads;
        if (nprob > 0x3fffffffL) return ISTVT_ERR_SHAPE;      // 32-bit problem index in the kernels
        dim3 grid((unsigned)((nprob + 3) / 4)), block(256);
#define TATTN_F(DHV, NTLV) hipLaunchKernelGGL((tattn_mfma_fwd_kernel<DHV, NTLV>), grid, block, 0, stream, (const bf16_t*)qk, (const bf16_t*)v, (bf16_t*)out, B, F, P, heads, scale, ldqk, ldv, ldo, diff)
        if (dh == 64) { if (F <= 16) TATTN_F(64, 1); else TATTN_F(64, 2); }
        else { if (F <= 16) TATTN_F(32, 1); else TATTN_F(32, 2); }
#undef TATTN_F
        return istvt_check_launch();
    }
    if (diff == 2) return ISTVT_ERR_SHAPE;      // pre-differenced operands: the bfloat16 MFMA kernels only
    DISPATCH_DTYPE(dtype, DISPATCH_TATTN(tattn_fwd_kernel, tattn_fwd2_kernel, (const T*)qk, (const T*)v, (T*)out, B, F, P,
                                         heads, scale, ldqk, ldv, ldo, diff));
    return istvt_check_launch();
}

extern "C" int istvt_attn_temporal_bwd(const void* qk, long ldqk, const void* v, long ldv, const void* dout, long ldo,
                                       void* dqk, void* dv, int B, int F, int P, int heads, int dh, float scale, int diff,
                                       int dtype, hipStream_t stream) {
    if (B <= 0 || F <= 0 || P <= 0 || heads <= 0 || diff < 0 || diff > 2) return ISTVT_ERR_SHAPE;
    if (ldqk < 2L * heads * dh || ldv < (long)heads * dh || ldo < (long)heads * dh || ldqk % 8 || ldv % 8 || ldo % 8) return ISTVT_ERR_SHAPE;
    static const int use_mfma = istvt_tune("ISTVT_TATTN_MFMA", 1);
    // measured at C2 / C4 (tools/tattn_bench.py): F = 9 lane-cluster 170 us vs MFMA 203 us (its three 32-row LDS images
    // allow 8 wavefronts per CU); F = 17 850 us vs 336 us
    static const int mfma_bwd_min = istvt_tune("ISTVT_TATTN_MFMA_BWD_MINF", 1);
    if (use_mfma && dtype == DT_BF16 && F >= mfma_bwd_min && F <= 32 && (dh == 64 || dh == 32)) {
        const long nprob = (long)B * P * heads;
        if (nprob > 0x3fffffffL) return ISTVT_ERR_SHAPE;      // 32-bit problem index in the kernel (+ one grid stride)
        long nwg = (nprob + 3) / 4;
        const long resident = 256L * (F <= 16 ? ISTVT_TB_WPE : 2);      // workgroups per CU by registers (105 / 213 VGPRs at dh 64)
        if (nwg > resident) nwg = resident;                   // wavefronts loop over problems
        dim3 grid((unsigned)nwg), block(256);
#define TATTN_B(DHV, NTLV) hipLaunchKernelGGL((tattn_mfma_bwd_kernel<DHV, NTLV>), grid, block, 0, stream, (const bf16_t*)qk, (const bf16_t*)v, (const bf16_t*)dout, (bf16_t*)dqk, (bf16_t*)dv, B, F, P, heads, scale, ldqk, ldv, ldo, diff)
        if (dh == 64) { if (F <= 16) TATTN_B(64, 1); else TATTN_B(64, 2); }
        else { if (F <= 16) TATTN_B(32, 1); else TATTN_B(32, 2); }
#undef TATTN_B
        return istvt_check_launch();
    }
    if (diff == 2) return ISTVT_ERR_SHAPE;
    DISPATCH_DTYPE(dtype, DISPATCH_TATTN(tattn_bwd_kernel, tattn_bwd2_kernel, (const T*)qk, (const T*)v, (const T*)dout,
                                         (T*)dqk, (T*)dv, B, F, P, heads, scale, ldqk, ldv, ldo, diff));
    return istvt_check_launch();
}
