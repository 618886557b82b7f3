// Per-position temporal self-attention over the frame axis (reference:
// TemporalResidualAttention.forward, network/vivit/module.py:197-205).  For every
// (clip b, position p, head h):  O = softmax(Q K^T * DH^-1/2) V with Q,K,V of shape [F][DH],
// F = T+1 <= 17.  The frame differencing of module.py:193 is NOT done here: q,k arrive already
// projected from the differenced LayerNorm output (see layernorm.hip).
//
// 0.1 % of the model's FLOPs and F x F tiles far below an MFMA tile: this is a data-movement
// kernel, bound by HBM.  A cluster of CL = DH/4 lanes owns one (b,p,h); each lane keeps 4
// consecutive head-dim elements of every frame's q/k/v row in registers, so a row segment
// (DH elements) is one fully-coalesced cluster access, a wavefront touches 64/CL whole
// DH-segments per instruction, dot products finish with log2(CL) DPP adds, no LDS.
//
// qk : [B*F*P][2*inner]  (q | k), v : [B*F*P][inner], out : [B*F*P][inner], rows ordered (b,f,p).
// Nothing is saved for backward: the F x F probabilities are recomputed from q,k.
#include "common.h"

template <int CL> __device__ __forceinline__ float cluster_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    if (CL == 16)
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));  // row_mirror
    return v;
}

__device__ __forceinline__ float dot4(const float (&a)[4], const float (&b)[4]) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}

template <typename T, int DH, int FMAX>
__global__ __launch_bounds__(256) void tattn_fwd_kernel(const T* __restrict__ qk, const T* __restrict__ v,
                                                        T* __restrict__ out, int B, int F, int P, int heads,
                                                        float scale) {
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
            load4(qk + m * 2 * inner + col, q[f]);
            load4(qk + m * 2 * inner + inner + col, k[f]);
            load4(v + m * inner + col, vv[f]);
        }
    }
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
                store4(out + m * inner + col, o);
            }
        }
    }
}

// backward: dqk [B*F*P][2*inner] (dq | dk), dv [B*F*P][inner]
template <typename T, int DH, int FMAX>
__global__ __launch_bounds__(256) void tattn_bwd_kernel(const T* __restrict__ qk, const T* __restrict__ v,
                                                        const T* __restrict__ dout, T* __restrict__ dqk,
                                                        T* __restrict__ dv, int B, int F, int P, int heads,
                                                        float scale) {
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
            load4(qk + m * 2 * inner + col, q[f]);
            load4(qk + m * 2 * inner + inner + col, k[f]);
            load4(v + m * inner + col, vv[f]);
            load4(dout + m * inner + col, dO[f]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { dk[f][e] = 0.f; dvv[f][e] = 0.f; }
        }
    }
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
            float delta = 0.f;
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) { pr[j] *= inv; delta += pr[j] * dp[j]; }
            }
            float dq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < FMAX; ++j) {
                if (j < F) {
                    const float ds = pr[j] * (dp[j] - delta) * scale;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dq[e] += ds * k[j][e];
                        dk[j][e] += ds * q[i][e];
                        dvv[j][e] += pr[j] * dO[i][e];
                    }
                }
            }
            if (valid) store4(dqk + m * 2 * inner + col, dq);
        }
    }
    if (valid) {
#pragma unroll
        for (int f = 0; f < FMAX; ++f) {
            if (f < F) {
                const long m = row0 + (long)f * P;
                store4(dqk + m * 2 * inner + inner + col, dk[f]);
                store4(dv + m * inner + col, dvv[f]);
            }
        }
    }
}

#define DISPATCH_TATTN(KERNEL, ...)                                                                      \
    do {                                                                                                 \
        const long ngroups = (long)B * P * heads;                                                        \
        const int cl = dh / 4, gw = 64 / cl;                                                             \
        const long blocks = (ngroups + 4 * gw - 1) / (4 * gw);                                           \
        dim3 grid((unsigned)blocks), block(256);                                                         \
        if (dh == 64 && F <= 9) hipLaunchKernelGGL((KERNEL<T, 64, 9>), grid, block, 0, stream, __VA_ARGS__);        \
        else if (dh == 64 && F <= 17) hipLaunchKernelGGL((KERNEL<T, 64, 17>), grid, block, 0, stream, __VA_ARGS__); \
        else if (dh == 32 && F <= 9) hipLaunchKernelGGL((KERNEL<T, 32, 9>), grid, block, 0, stream, __VA_ARGS__);   \
        else if (dh == 32 && F <= 17) hipLaunchKernelGGL((KERNEL<T, 32, 17>), grid, block, 0, stream, __VA_ARGS__); \
        else return ISTVT_ERR_SHAPE;                                                                     \
    } while (0)

extern "C" int istvt_attn_temporal_fwd(const void* qk, const void* v, void* out, int B, int F, int P, int heads,
                                       int dh, float scale, int dtype, hipStream_t stream) {
    if (B <= 0 || F <= 0 || P <= 0 || heads <= 0) return ISTVT_ERR_SHAPE;
    DISPATCH_DTYPE(dtype, DISPATCH_TATTN(tattn_fwd_kernel, (const T*)qk, (const T*)v, (T*)out, B, F, P, heads, scale));
    return istvt_check_launch();
}

extern "C" int istvt_attn_temporal_bwd(const void* qk, const void* v, const void* dout, void* dqk, void* dv, int B,
                                       int F, int P, int heads, int dh, float scale, int dtype, hipStream_t stream) {
    if (B <= 0 || F <= 0 || P <= 0 || heads <= 0) return ISTVT_ERR_SHAPE;
    DISPATCH_DTYPE(dtype, DISPATCH_TATTN(tattn_bwd_kernel, (const T*)qk, (const T*)v, (const T*)dout, (T*)dqk, (T*)dv,
                                         B, F, P, heads, scale));
    return istvt_check_launch();
}
