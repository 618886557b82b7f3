// LayerNorm forward / backward over rows of width D (reference: nn.LayerNorm inside PreNorm,
// network/vivit/module.py:15-21; STTransformer.norm vivit.py:89; mlp_head[0] vivit.py:128).
//
// One wavefront per row: the row lives in registers (8-element chunks, 16 B bf16 / 32 B f32 per
// lane access), mean and variance are two wave reductions (two-pass, fp32), gamma/beta are fp32.
// HBM-bound: algorithmic traffic = one read + one write of the row (forward).
//
// ln_fwd_diff (bfloat16 path of TemporalResidualAttention, round 4): also writes the frame difference of module.py:193,
//   diff[b,f,p] = y[b,f,p] (f < 2),  y[b,f,p] - y[b,f-1,p] (f >= 2),
// taken in fp32 BEFORE the rounding to the storage type: the q / k projection then sees operands rounded at the
// magnitude of the difference, as the reference's own order of operations does.  (Rounds 2-3 projected the
// un-differenced rows and differenced bf16 q / k afterwards: for correlated consecutive frames -- what a face video
// is -- the rounding error then scales with |q| instead of |q'|, 11 % of q' at a 3 % frame difference.)  One wavefront
// walks the F frames of a position and keeps the previous normalised row in registers.  float32 keeps the in-kernel
// difference of attn_temporal.hip (no rounding to lose).
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

// forward, temporal variant: one wavefront per (b, p) walks f = 0..F-1; rows at (b*F + f)*P + p
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_diff_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ y,
                                                          T* __restrict__ diff, float* __restrict__ mean_out,
                                                          float* __restrict__ rstd_out, int Bn, int F, int P, int D,
                                                          float eps, long ldx, long ldy, long ldd) {
    const int lane = threadIdx.x & 63;
    // position arithmetic on the scalar unit, in 32 bits (the launcher checks B * P): as `long` from the lane-valued
    // threadIdx.x >> 6 every position cost four 64-bit vector divisions
    const unsigned wave = blockIdx.x * 4u + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nwaves = gridDim.x * 4u;
    float gm[LN_MAXCH][8], bt[LN_MAXCH][8];
    ln_row_load<float>(gamma, D, lane, gm);
    ln_row_load<float>(beta, D, lane, bt);
    const unsigned npos = (unsigned)Bn * (unsigned)P, Pu = (unsigned)P;
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
    auto first_row = [&](unsigned w) { const unsigned b = w / Pu; return ((long)b * F) * P + (w - b * Pu); };
    if (wave < npos) fetch(first_row(wave));
    for (unsigned w = wave; w < npos; w += nwaves) {
        const long b = w / Pu, pp = w - (w / Pu) * Pu;
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
            else if (w + nwaves < npos) fetch(first_row(w + nwaves));
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

// The same backward for bf16 rows with the operands staged by LDS-DMA (round 3).  The register version above keeps ONE
// row ahead per wavefront (its raw chunks cost 24 registers and the kernel already sits at two wavefronts per SIMD):
// 35 KB in flight per CU, and SQ_WAIT_ANY 0.64 -- it waits on HBM latency, not on bandwidth.  Here every wavefront owns
// a ring of LN_RING row slots in LDS; the rows LN_RING - 1 ahead are on their way while it reduces the current one,
// with no register cost (buffer_load ... lds), and the wait is a counted vmcnt: all VMEM operations of the loop are
// issued by hand in a fixed order (per row: the DMAs of dy, x [, dres], mean, rstd; then the dx stores), so
// "row i has landed" is "all but the (LN_RING-1) * NLD + min(i, LN_RING) * NST youngest operations are done".
// Rows past M and the lanes past D are sent out of range (no traffic, zeros in LDS), never branched around.
constexpr int LN_RING = 3;
constexpr int LN_SLOT_BYTES = 3 * 2048 + 512;          // dy, x, dres: 2 chunks of 1 KiB each; mean, rstd: 256 B each

template <int N> __device__ __forceinline__ void ln_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool DCOL, bool HAS_RES, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_dma_kernel(const bf16_t* __restrict__ dy1, const bf16_t* __restrict__ x,
                                                         const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                         const float* __restrict__ gamma, const bf16_t* __restrict__ dres,
                                                         bf16_t* __restrict__ dx, float* __restrict__ ws, long M, int D,
                                                         long ld_dy, long ld_x, long ld_res, long ld_dx) {
    constexpr int NACC = DCOL ? 3 : 2;
    constexpr int NARR = HAS_RES ? 3 : 2;
    constexpr int NLD = NARR * NCH + 2, NST = NCH;     // VMEM operations per row: loads (DMA), stores
    static_assert(NACC * 4 * LN_MAXCH * 512 * 4 <= 4 * LN_RING * LN_SLOT_BYTES, "the final reduction aliases the ring");
    __shared__ __attribute__((aligned(16))) char smem[4 * LN_RING * LN_SLOT_BYTES];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    unsigned voff[LN_MAXCH];
#pragma unroll
    for (int c = 0; c < LN_MAXCH; ++c) {
        on[c] = c < NCH && (lane + 64 * c) * 8 < D;
        voff[c] = on[c] ? (unsigned)((lane + 64 * c) * 16) : 0x80000000u;
    }
    auto uni = [](const void* q) -> char* {
        const unsigned long long u = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return (char*)(((unsigned long long)hi << 32) | lo);
    };
    constexpr unsigned FLAGS = 0x00020000u;
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(uni(dy1), 0, (int)(M * ld_dy * 2), FLAGS);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(uni(x), 0, (int)(M * ld_x * 2), FLAGS);
    const __amdgpu_buffer_rsrc_t rs_rs = __builtin_amdgcn_make_buffer_rsrc(uni(HAS_RES ? dres : x), 0, (int)(M * (HAS_RES ? ld_res : ld_x) * 2), FLAGS);
    const __amdgpu_buffer_rsrc_t rs_mu = __builtin_amdgcn_make_buffer_rsrc(uni(mean_in), 0, (int)(M * 4), FLAGS);
    const __amdgpu_buffer_rsrc_t rs_rd = __builtin_amdgcn_make_buffer_rsrc(uni(rstd_in), 0, (int)(M * 4), FLAGS);
    typedef __attribute__((address_space(3))) void lds_v;
    const unsigned ring0 = (unsigned)(__SIZE_TYPE__)(lds_v*)smem + wid * (LN_RING * LN_SLOT_BYTES);
    const char* ringp = smem + wid * (LN_RING * LN_SLOT_BYTES);

    // row i of this wavefront -> slot i % LN_RING; rows past M go out of range as a whole (scalar offset past num_records)
    auto issue = [&](long i) {
        const long m = wave + i * nwaves;
        const bool ok = m < M;
        const unsigned dst = ring0 + (unsigned)(i % LN_RING) * LN_SLOT_BYTES;
        const int dead = ok ? 0 : 0x7fffffff;
        const int s_dy = ok ? (int)(m * ld_dy * 2) : dead, s_x = ok ? (int)(m * ld_x * 2) : dead;
        const int s_rs = ok ? (int)(m * ld_res * 2) : dead, s_m = ok ? (int)(m * 4) : dead;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            dma16_lds(rs_dy, dst + c * 1024, voff[c], s_dy);
            dma16_lds(rs_x, dst + 2048 + c * 1024, voff[c], s_x);
            if (HAS_RES) dma16_lds(rs_rs, dst + 4096 + c * 1024, voff[c], s_rs);
        }
        dma4_lds(rs_mu, dst + 6144, 0u, s_m);               // every lane gets its own copy of the row's mean / rstd
        dma4_lds(rs_rd, dst + 6144 + 256, 0u, s_m);
    };
    const long n_rows = wave < M ? (M - wave + nwaves - 1) / nwaves : 0;
#pragma unroll
    for (int i = 0; i < LN_RING; ++i) issue(i);
    for (long i = 0; i < n_rows; ++i) {
        // row i landed: younger are the loads of rows i+1 .. i+LN_RING-1 and the stores of rows max(0, i-LN_RING) .. i-1
        if (i >= LN_RING) ln_wait_vm<(LN_RING - 1) * NLD + LN_RING * NST>();
        else if (i == 2) ln_wait_vm<(LN_RING - 1) * NLD + 2 * NST>();
        else if (i == 1) ln_wait_vm<(LN_RING - 1) * NLD + 1 * NST>();
        else ln_wait_vm<(LN_RING - 1) * NLD>();
        static_assert(LN_RING == 3, "the three warm-up waits above");
        const char* slot = ringp + (i % LN_RING) * LN_SLOT_BYTES;
        bf16x8 rdy[LN_MAXCH], rx[LN_MAXCH], rr[LN_MAXCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            rdy[c] = *reinterpret_cast<const bf16x8*>(slot + c * 1024 + lane * 16);
            rx[c] = *reinterpret_cast<const bf16x8*>(slot + 2048 + c * 1024 + lane * 16);
            if (HAS_RES) rr[c] = *reinterpret_cast<const bf16x8*>(slot + 4096 + c * 1024 + lane * 16);
        }
        const float mean = *reinterpret_cast<const float*>(slot + 6144 + lane * 4);
        const float rstd = *reinterpret_cast<const float*>(slot + 6144 + 256 + lane * 4);
        // the slot is free once these reads have returned: refill it with row i + LN_RING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue(i + LN_RING);
        const long m = wave + i * nwaves;
        float dy[LN_MAXCH][8], xv[LN_MAXCH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (on[c]) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    dy[c][k] = (float)rdy[c][k];
                    const float xh = ((float)rx[c][k] - mean) * rstd;
                    xv[c][k] = xh;
                    const float gdy = dy[c][k] * gm[c][k];
                    s1 += gdy;
                    s2 += gdy * xh;
                    ag[c][k] += dy[c][k] * xh;
                    ab[c][k] += dy[c][k];
                }
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = on[c] ? rstd * (dy[c][k] * gm[c][k] - s1 - xv[c][k] * s2) : 0.f;
            if (HAS_RES) {
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] += on[c] ? (float)rr[c][k] : 0.f;
            }
            if (DCOL) {
#pragma unroll
                for (int k = 0; k < 8; ++k) ac[c][k] += o[k];
            }
            // exactly one store instruction per chunk (the count the waits above assume; chunk c < NCH always has active
            // lanes).  A plain global store: the buffer form with the row in an SGPR offset hit the store-data hazard of
            // gemm_shared.h (its data registers are overwritten by the next row's LDS reads a few instructions later).
            if (on[c]) store8(dx + m * ld_dx + (lane + 64 * c) * 8, o);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the out-of-range refills of the last rows still target the ring
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);            // [NACC][4][LN_MAXCH * 512]
    constexpr int RW = LN_MAXCH * 512;
#pragma unroll
    for (int c = 0; c < LN_MAXCH; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[(0 * 4 + wid) * RW + (c * 64 + lane) * 8 + k] = ag[c][k];
            red[(1 * 4 + wid) * RW + (c * 64 + lane) * 8 + k] = ab[c][k];
            if (DCOL) red[((NACC - 1) * 4 + wid) * RW + (c * 64 + lane) * 8 + k] = ac[c][k];
        }
    __syncthreads();
    float* mine = ws + (long)blockIdx.x * NACC * D;
    for (int col = threadIdx.x; col < D; col += 256) {
#pragma unroll
        for (int a = 0; a < NACC; ++a)
            mine[a * D + col] = (red[(a * 4 + 0) * RW + col] + red[(a * 4 + 1) * RW + col]) + (red[(a * 4 + 2) * RW + col] + red[(a * 4 + 3) * RW + col]);
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

// y and diff (see ln_fwd_diff_kernel): rows ordered (b, f, p), B clips x F frames x P positions
extern "C" int istvt_layernorm_fwd_diff(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy,
                                        void* diff, long ldd, float* mean, float* rstd, int B, int F, int P, int D,
                                        float eps, int dtype, hipStream_t stream) {
    if (D % 8 != 0 || D > LN_MAXCH * 512 || B <= 0 || F <= 0 || P <= 0) return ISTVT_ERR_SHAPE;
    if (ldx < D || ldy < D || ldd < D || ldx % 8 || ldy % 8 || ldd % 8) return ISTVT_ERR_SHAPE;
    if ((long)B * P > 0x3fffffffL) return ISTVT_ERR_SHAPE;            // 32-bit position index in the kernel
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_fwd_diff_kernel<T>), dim3(ln_grid((long)B * P)), dim3(256), 0, stream,
                                             (const T*)x, gamma, beta, (T*)y, (T*)diff, mean, rstd, B, F, P, D, eps, ldx,
                                             ldy, ldd));
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
// the row kernel alone: dx and the per-workgroup partial rows in ws (dcol: a third accumulator, the column sums of dx)
static int ln_bwd_launch(const void* dy, long ld_dy, const void* x, long ld_x, const float* mean, const float* rstd,
                         const float* gamma, const void* dres, long ld_res, void* dx, long ld_dx, bool dcol, float* ws,
                         long ws_elems, long M, int D, int dtype, hipStream_t stream) {
    if (D % 8 != 0 || D > LN_MAXCH * 512 || M <= 0 || !ws) return ISTVT_ERR_SHAPE;
    if (ld_dy < D || ld_x < D || ld_dx < D || ld_dy % 8 || ld_x % 8 || ld_dx % 8) return ISTVT_ERR_SHAPE;
    if (dres && (ld_res < D || ld_res % 8)) return ISTVT_ERR_SHAPE;
    const long blocks = ln_bwd_blocks(M);
    const int nacc = dcol ? 3 : 2;
    if (ws_elems < blocks * nacc * D) return ISTVT_ERR_SHAPE;
    // bf16 rows whose operands fit 32-bit buffer offsets: the LDS-DMA ring kernel; everything else the register kernel
    static const int dma_on = istvt_tune("ISTVT_LN_BWD_DMA", 1);
    const long lim = 0x7fffffffL / 2;
    const bool dma = dma_on && dtype == DT_BF16 && M * ld_dy < lim && M * ld_x < lim && M * ld_dx < lim && (!dres || M * ld_res < lim) &&
                     ((uintptr_t)dy % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dx % 16) == 0 && (!dres || ((uintptr_t)dres % 16) == 0);
    if (dma) {
#define LN_DMA(DCOLV, RESV, NCHV)                                                                                          \
    hipLaunchKernelGGL((ln_bwd_dma_kernel<DCOLV, RESV, NCHV>), dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)dy, \
                       (const bf16_t*)x, mean, rstd, gamma, (const bf16_t*)dres, (bf16_t*)dx, ws, M, D, ld_dy, ld_x, ld_res, ld_dx)
        const int sel = (dcol ? 4 : 0) | (dres ? 2 : 0) | (D > 512 ? 1 : 0);
        switch (sel) {
            case 0: LN_DMA(false, false, 1); break;
            case 1: LN_DMA(false, false, 2); break;
            case 2: LN_DMA(false, true, 1); break;
            case 3: LN_DMA(false, true, 2); break;
            case 4: LN_DMA(true, false, 1); break;
            case 5: LN_DMA(true, false, 2); break;
            case 6: LN_DMA(true, true, 1); break;
            default: LN_DMA(true, true, 2); break;
        }
#undef LN_DMA
    } else if (dcol)
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, true>), dim3((int)blocks), dim3(256), 0, stream,
                                                 (const T*)dy, (const T*)x, mean, rstd, gamma, (const T*)dres, (T*)dx, ws, M,
                                                 D, ld_dy, ld_x, ld_res, ld_dx));
    else
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, false>), dim3((int)blocks), dim3(256), 0, stream,
                                                 (const T*)dy, (const T*)x, mean, rstd, gamma, (const T*)dres, (T*)dx, ws, M,
                                                 D, ld_dy, ld_x, ld_res, ld_dx));
    return istvt_check_launch();
}

extern "C" int istvt_layernorm_bwd(const void* dy, long ld_dy, const void* x, long ld_x, const float* mean,
                                   const float* rstd, const float* gamma, const void* dres, long ld_res, void* dx,
                                   long ld_dx, float* dgamma, float* dbeta, float* dcol, float* ws, long ws_elems, long M,
                                   int D, int dtype, hipStream_t stream) {
    const int rc = ln_bwd_launch(dy, ld_dy, x, ld_x, mean, rstd, gamma, dres, ld_res, dx, ld_dx, dcol != nullptr, ws, ws_elems,
                                 M, D, dtype, stream);
    if (rc) return rc;
    return istvt_rows_reduce_add(ws, (int)ln_bwd_blocks(M), dcol ? 3 : 2, D, dgamma, dbeta, dcol, stream);
}

// The two halves of istvt_layernorm_bwd for a caller that runs the fold of the partial rows elsewhere (another stream,
// later): _partial writes dx and ws (with_dcol: also the column sums of dx), _reduce folds ws into dgamma / dbeta (/ dcol:
// non-null exactly when the partial call had with_dcol) in the same fixed order.  Nothing on the caller's critical path
// reads the parameter gradients, but 38 five-microsecond reduce launches per training step sat on it.
extern "C" int istvt_layernorm_bwd_partial(const void* dy, long ld_dy, const void* x, long ld_x, const float* mean,
                                           const float* rstd, const float* gamma, const void* dres, long ld_res, void* dx,
                                           long ld_dx, int with_dcol, float* ws, long ws_elems, long M, int D, int dtype,
                                           hipStream_t stream) {
    return ln_bwd_launch(dy, ld_dy, x, ld_x, mean, rstd, gamma, dres, ld_res, dx, ld_dx, with_dcol != 0, ws, ws_elems, M, D,
                         dtype, stream);
}
extern "C" int istvt_layernorm_bwd_reduce(const float* ws, long ws_elems, long M, int D, float* dgamma, float* dbeta,
                                          float* dcol, hipStream_t stream) {
    if (D % 8 != 0 || D > LN_MAXCH * 512 || M <= 0 || !ws || !dgamma || !dbeta) return ISTVT_ERR_SHAPE;
    const long blocks = ln_bwd_blocks(M);
    const int nacc = dcol ? 3 : 2;
    if (ws_elems < blocks * nacc * D) return ISTVT_ERR_SHAPE;
    return istvt_rows_reduce_add(ws, (int)blocks, nacc, D, dgamma, dbeta, dcol, stream);
}
