// MFMA formulation of the per-position temporal attention for bfloat16 (included by attn_temporal.hip).
//
// The lane-cluster kernels of attn_temporal.hip spend ~28 vector instructions per (query frame, key frame) pair and
// are VALU-bound, badly so at F = 17 (T = 16: 1 TB/s).  Here ONE WAVEFRONT owns one (clip b, position p, head h):
// F <= 32 frames are padded to two 16-row MFMA tiles and every product is a v_mfma_f32_16x16x32_bf16, exactly the
// dataflow of the spatial kernels (attn_spatial.hip) with a 32-row "chunk":
//   forward   S^T = K Q^T (keys on accumulator rows, queries on lanes) -> softmax over keys in registers ->
//             O^T = V^T P^T with the S^T accumulators as the B operand and V read transposed from LDS
//   backward  part 1 (per query tile): S^T, dP^T = V dO^T, p, delta = sum_j p dp, dS^T -> dQ^T = K^T dS^T;
//                    (max, 1/sum, delta) per query go to a 3 x 32 LDS table
//             part 2 (per key tile):   S = Q K^T, dP = dO V^T (queries on accumulator rows, keys on lanes),
//                    p, dS from the table -> dV^T = dO^T P, dK^T = Q^T dS
// Rows of one problem are F rows P apart (row stride P * ld), so every operand fragment that is not transposed is
// loaded straight from global memory in MFMA layout (16 bytes per lane, one row per lane); the operands that are
// read TRANSPOSED (V forward; K, Q, dO backward) are staged through a wave-private LDS image of 32 rows.
// No workgroup barrier anywhere: a wavefront's LDS operations execute in order.
#pragma once

namespace tmf {
constexpr int ROWS = 32, IPAD = 8;
#define TM_LOG2E 1.4426950408889634f

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (bf16_t)0.0f;
    return f;
}
// 8 consecutive elements of row `row` (rows >= nrows read as zeros); rstride = elements between rows
__device__ __forceinline__ bf16x8 row_frag(const bf16_t* __restrict__ src, long rstride, int row, int nrows, int col) {
    if (row < nrows) return *reinterpret_cast<const bf16x8*>(src + (long)row * rstride + col);
    return zero8();
}
__device__ __forceinline__ bf16x8 acc_frag(const f32x4& lo, const f32x4& hi) {
    bf16x8 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = (bf16_t)lo[i]; f[4 + i] = (bf16_t)hi[i]; }
    return f;
}
__device__ __forceinline__ void mma(f32x4& c, const bf16x8& a, const bf16x8& b) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// K = 16 step (F <= 16: the contraction over frames fits ONE 16-row tile): k-slot (g, i) <-> frame 4g + i, which is
// exactly the accumulator row of the S tile -> the B operand is the accumulator converted to bf16, and the transposed
// A operand is ONE ds_read_b64_tr_b16 (rows 4g .. 4g+3, lane r receives column col16 + r)
typedef short short4k __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mma16(f32x4& c, const short4k& a, const short4k& b) {
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ short4k acc_frag4(const f32x4& v) {
    bf16x4 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = (bf16_t)v[i];
    return __builtin_bit_cast(short4k, f);
}
__device__ __forceinline__ short4k tr4(const bf16_t* img, int ld, int k0, int col16, int r) {
    typedef short4v __attribute__((address_space(3))) * lds_ptr;
    const int q = r >> 2, p = r & 3;
    const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + (size_t)(k0 + q) * ld + col16 + 4 * p));
    return __builtin_bit_cast(short4k, v);
}
// one product "acc += X^T(frames x 16 columns at col16)^T * Y" over the frames of NTL tiles: ya / yb are the
// accumulator tiles whose rows are the frames (yb unused when NTL == 1)
template <int NTL>
__device__ __forceinline__ void mma_frames(f32x4& acc, const bf16_t* img, int ld, int col16, int g, int r, const f32x4& ya,
                                           const f32x4& yb) {
    if constexpr (NTL == 1) mma16(acc, tr4(img, ld, 4 * g, col16, r), acc_frag4(ya));
    else mma(acc, frag_load_tr(img, ld, 4 * g, 16 + 4 * g, col16, r), acc_frag(ya, yb));
}
__device__ __forceinline__ float group_max(float v) { return xgroup_max(v); }      // over the 4 lane groups (same r)
__device__ __forceinline__ float group_sum(float v) { return xgroup_sum(v); }
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
// rows 0 .. 16 NTL - 1 x DH of a strided matrix -> wave-private LDS image [16 NTL][DH + IPAD]; rows >= nrows are zero-filled
template <int DH, int NTL>
__device__ __forceinline__ void stage32(bf16_t* img, const bf16_t* __restrict__ src, long rstride, int nrows, int lane) {
    constexpr int LDI = DH + IPAD, VPR = DH / 8, RPI = 64 / VPR;      // rows per wave-instruction
#pragma unroll
    for (int it = 0; it < 16 * NTL / RPI; ++it) {
        const int row = it * RPI + lane / VPR, col = (lane % VPR) * 8;
        *reinterpret_cast<bf16x8*>(img + row * LDI + col) = row_frag(src, rstride, row, nrows, col);
    }
}
// ---- the frame difference of module.py:193 on projected rows (see attn_temporal.hip) ---------------------------------
// Fragment layout: lane (g, r) holds row 16 t + r of a tile.  Row r - 1 is lane r - 1 of the same 16-lane DPP row
// (row_shr:1); lane r = 0 receives row 15 of the tile BELOW (row_ror:1 of that tile's fragment).
__device__ __forceinline__ bf16x8 prev_row_frag(const bf16x8& cur, const bf16x8& below) {
    const u32x4 c = __builtin_bit_cast(u32x4, cur), b = __builtin_bit_cast(u32x4, below);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned wrap = __builtin_amdgcn_update_dpp(0u, b[i], 0x121 /* row_ror:1 */, 0xf, 0xf, true);
        o[i] = __builtin_amdgcn_update_dpp(wrap, c[i], 0x111 /* row_shr:1: lane 0 keeps `wrap` */, 0xf, 0xf, false);
    }
    return __builtin_bit_cast(bf16x8, o);
}
// x'[f] = x[f] - x[f-1] for f >= 2 (rounded once to bf16, the MFMA operand type)
__device__ __forceinline__ bf16x8 diff_frag(const bf16x8& cur, const bf16x8& prev, const int f) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f >= 2 ? (bf16_t)((float)cur[i] - (float)prev[i]) : cur[i];
    return o;
}
// accumulator layout: lane (g, r) holds row 16 u + r.  Row r + 1 is lane r + 1 (row_shl:1); lane 15 receives row 0 of
// the tile ABOVE.  The adjoint of the difference: d x[f] = d x'[f] - d x'[f+1] for f + 1 >= 2.
__device__ __forceinline__ f32x4 diff_adjoint_acc(const f32x4& cur, const f32x4& above, const int f) {
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned wrap = __builtin_amdgcn_update_dpp(0u, __float_as_uint(above[i]), 0x12F /* row_ror:15 */, 0xf, 0xf, true);
        const unsigned nx = __builtin_amdgcn_update_dpp(wrap, __float_as_uint(cur[i]), 0x101 /* row_shl:1: lane 15 keeps `wrap` */, 0xf, 0xf, false);
        o[i] = f >= 1 ? cur[i] - __uint_as_float(nx) : cur[i];
    }
    return o;
}
// the difference applied in place to a staged [16 NTL][DH + IPAD] image (one wavefront): every lane holds its 8-element
// chunk of row `row` in `mine` (as staged), reads the chunk of the row above it in the frame order from the image,
// and -- after every lane has read -- overwrites its own
template <int DH, int NTL>
__device__ __forceinline__ void diff_image(bf16_t* img, const bf16x8 (&mine)[16 * NTL / (64 / (DH / 8))], int lane) {
    constexpr int LDI = DH + IPAD, VPR = DH / 8, RPI = 64 / VPR, NIT = 16 * NTL / RPI;
    bf16x8 prev[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = it * RPI + lane / VPR, col = (lane % VPR) * 8;
        prev[it] = *reinterpret_cast<const bf16x8*>(img + (row >= 1 ? row - 1 : 0) * LDI + col);
    }
    wave_lds_fence();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = it * RPI + lane / VPR, col = (lane % VPR) * 8;
        *reinterpret_cast<bf16x8*>(img + row * LDI + col) = diff_frag(mine[it], prev[it], row);
    }
}
}  // namespace tmf

// NTL = 16-row tiles that hold the F frames (1: F <= 16, 2: F <= 32)
template <int DH, int NTL>
__global__ __launch_bounds__(256) void tattn_mfma_fwd_kernel(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ v,
                                                             bf16_t* __restrict__ out, int B, int F, int P, int heads,
                                                             float scale, long ldqk, long ldv, long ldo, int diff) {
    constexpr int LDI = DH + tmf::IPAD, KS = DH / 32, DT = DH / 16;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4][16 * NTL * LDI];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r = lane & 15;
    const long prob = (long)blockIdx.x * 4 + wave;
    if (prob >= (long)B * P * heads) return;              // no workgroup-level synchronisation below
    const int h = (int)(prob % heads);
    const long bp = prob / heads, b = bp / P, p = bp % P;
    const int inner = heads * DH;
    const long row0 = b * F * P + p;
    const long sq = (long)P * ldqk, sv = (long)P * ldv, so = (long)P * ldo;
    const bf16_t* qp = qk + row0 * ldqk + h * DH;
    const bf16_t* kp = qp + inner;
    const bf16_t* vp = v + row0 * ldv + h * DH;
    bf16_t* op = out + row0 * ldo + h * DH;
    bf16_t* Vimg = smem[wave];
    const float c = scale * TM_LOG2E;

    tmf::stage32<DH, NTL>(Vimg, vp, sv, F, lane);
    bf16x8 kf[NTL][KS];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[t][ks] = tmf::row_frag(kp, sq, 16 * t + r, F, 32 * ks + 8 * g);
    if (diff) {                                             // top tile first: the tile below is still un-differenced
#pragma unroll
        for (int t = NTL - 1; t >= 0; --t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                kf[t][ks] = tmf::diff_frag(kf[t][ks], tmf::prev_row_frag(kf[t][ks], kf[t > 0 ? t - 1 : 0][ks]), 16 * t + r);
    }
    tmf::wave_lds_fence();

    bf16x8 qbelow[KS];                                      // the un-differenced query rows of the previous tile
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qbelow[ks] = tmf::zero8();
#pragma unroll
    for (int u = 0; u < NTL; ++u) {
        if (16 * u >= F) break;
        bf16x8 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = tmf::row_frag(qp, sq, 16 * u + r, F, 32 * ks + 8 * g);
        if (diff) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 raw = qf[ks];
                qf[ks] = tmf::diff_frag(raw, tmf::prev_row_frag(raw, qbelow[ks]), 16 * u + r);
                qbelow[ks] = raw;
            }
        }
        f32x4 s[2];
        s[1] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            s[t] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) tmf::mma(s[t], kf[t][ks], qf[ks]);
        }
        // softmax over keys: lane owns query r, keys 16t + 4g + j
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = (16 * t + 4 * g + j) < F ? s[t][j] * c : -INFINITY;
                s[t][j] = z;
                mx = fmaxf(mx, z);
            }
        mx = tmf::group_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pv = fast_exp2(s[t][j] - mx);
                s[t][j] = pv;
                sum += pv;
            }
        sum = tmf::group_sum(sum);
        const float inv = 1.0f / sum;
        const int q = 16 * u + r;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 o = f32x4{0, 0, 0, 0};
            tmf::mma_frames<NTL>(o, Vimg, LDI, 16 * dt, g, r, s[0], s[1]);
            if (q < F) {
                float ov[4] = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
                store4(op + (long)q * so + 16 * dt + 4 * g, ov);
            }
        }
    }
}

template <int DH, int NTL>
__global__ __launch_bounds__(256) void tattn_mfma_bwd_kernel(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ v,
                                                             const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqk,
                                                             bf16_t* __restrict__ dv, int B, int F, int P, int heads,
                                                             float scale, long ldqk, long ldv, long ldo, int diff) {
    constexpr int LDI = DH + tmf::IPAD, KS = DH / 32, DT = DH / 16;
    constexpr int IMG = 16 * NTL * LDI;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4][3 * IMG];
    __shared__ __attribute__((aligned(16))) float stat[4][3][16 * NTL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r = lane & 15;
    const long total = (long)B * P * heads, nwaves = (long)gridDim.x * 4;
    long prob = (long)blockIdx.x * 4 + wave;
    if (prob >= total) return;                            // no workgroup-level synchronisation below
    const int inner = heads * DH;
    const long sq = (long)P * ldqk, sv = (long)P * ldv, so = (long)P * ldo;
    bf16_t* Qimg = smem[wave];
    bf16_t* Kimg = Qimg + IMG;
    bf16_t* Dimg = Kimg + IMG;
    float (*st)[16 * NTL] = stat[wave];
    const float c = scale * TM_LOG2E;
    // A wavefront walks problems prob, prob + nwaves, ...: everything the next problem needs from HBM (the Q, K, dO
    // rows for the three LDS images and the V fragments) is requested before the current one is computed and waits
    // in registers -- one problem per wavefront left five wavefronts per SIMD each idling through its own latency.
    constexpr int VPR = DH / 8, RPI = 64 / VPR, NIT = 16 * NTL / RPI;
    bf16x8 nq[NIT], nk[NIT], nd[NIT], nv[NTL][KS];
    auto fetch = [&](long pr) {
        const int h = (int)(pr % heads);
        const long bp = pr / heads, b = bp / P, pp = bp % P;
        const long row0 = b * F * P + pp;
        const bf16_t* qp = qk + row0 * ldqk + h * DH;
        const bf16_t* kp = qp + inner;
        const bf16_t* vp = v + row0 * ldv + h * DH;
        const bf16_t* dop = dout + row0 * ldo + h * DH;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RPI + lane / VPR, col = (lane % VPR) * 8;
            nq[it] = tmf::row_frag(qp, sq, row, F, col);
            nk[it] = tmf::row_frag(kp, sq, row, F, col);
            nd[it] = tmf::row_frag(dop, so, row, F, col);
        }
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) nv[t][ks] = tmf::row_frag(vp, sv, 16 * t + r, F, 32 * ks + 8 * g);
    };
    fetch(prob);
    for (; prob < total; prob += nwaves) {
    const int h = (int)(prob % heads);
    const long bp = prob / heads, b = bp / P, p = bp % P;
    const long row0 = b * F * P + p;
    bf16_t* dqp = dqk + row0 * ldqk + h * DH;
    bf16_t* dkp = dqp + inner;
    bf16_t* dvp = dv + row0 * ldv + h * DH;
    tmf::wave_lds_fence();                                // the previous problem's image reads are done
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = it * RPI + lane / VPR, col = (lane % VPR) * 8;
        *reinterpret_cast<bf16x8*>(Qimg + row * LDI + col) = nq[it];
        *reinterpret_cast<bf16x8*>(Kimg + row * LDI + col) = nk[it];
        *reinterpret_cast<bf16x8*>(Dimg + row * LDI + col) = nd[it];
    }
    bf16x8 kf[NTL][KS], vf[NTL][KS];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vf[t][ks] = nv[t][ks];
    tmf::wave_lds_fence();
    if (diff) {                                           // Q, K images -> differenced in place (nq / nk still hold the rows)
        tmf::diff_image<DH, NTL>(Qimg, nq, lane);
        tmf::diff_image<DH, NTL>(Kimg, nk, lane);
        tmf::wave_lds_fence();
    }
    if (prob + nwaves < total) fetch(prob + nwaves);
    // K rows in fragment layout (row 16t + r, columns 32ks + 8g), from the image
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            kf[t][ks] = *reinterpret_cast<const bf16x8*>(Kimg + (16 * t + r) * LDI + 32 * ks + 8 * g);

    // ---- part 1: per query tile u -- S^T, dP^T (keys on rows, queries on lanes), statistics, dQ
    f32x4 above[DT];                                      // diff: d q' of the tile above (its row 0 closes row 15 below)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) above[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int u = NTL - 1; u >= 0; --u) {
        if (16 * u >= F) continue;
        const int q = 16 * u + r;
        bf16x8 qf[KS], dof[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(Qimg + q * LDI + 32 * ks + 8 * g);
            dof[ks] = *reinterpret_cast<const bf16x8*>(Dimg + q * LDI + 32 * ks + 8 * g);
        }
        f32x4 s[2], dp[2];
        s[1] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            s[t] = f32x4{0, 0, 0, 0}; dp[t] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                tmf::mma(s[t], kf[t][ks], qf[ks]);
                tmf::mma(dp[t], vf[t][ks], dof[ks]);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = (16 * t + 4 * g + j) < F ? s[t][j] * c : -INFINITY;
                s[t][j] = z;
                mx = fmaxf(mx, z);
            }
        mx = tmf::group_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pv = fast_exp2(s[t][j] - mx);
                s[t][j] = pv;
                sum += pv;
            }
        sum = tmf::group_sum(sum);
        const float inv = 1.0f / sum;
        float dl = 0.f;
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[t][j] *= inv; dl += s[t][j] * dp[t][j]; }
        dl = tmf::group_sum(dl);
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[t][j] = s[t][j] * (dp[t][j] - dl) * scale;        // dS^T
        if (g == 0) { st[0][q] = mx; st[1][q] = inv; st[2][q] = dl; }                       // rows q >= F: never used (p = 0 there)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 dq = f32x4{0, 0, 0, 0};
            tmf::mma_frames<NTL>(dq, Kimg, LDI, 16 * dt, g, r, s[0], s[1]);
            if (diff) {                                   // rows >= F carry zero gradients (their dO rows are zero)
                const f32x4 raw = dq;
                dq = tmf::diff_adjoint_acc(raw, above[dt], q);
                above[dt] = raw;
            }
            if (q < F) {
                float o[4] = {dq[0], dq[1], dq[2], dq[3]};
                store4(dqp + (long)q * sq + 16 * dt + 4 * g, o);
            }
        }
    }
    tmf::wave_lds_fence();

    // ---- part 2: per key tile kt -- S, dP (queries on rows, keys on lanes) -> dV, dK
    // B operands of this orientation are the K / V rows of the tile: kf[kt], vf[kt] as loaded above
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) above[dt] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int kt = NTL - 1; kt >= 0; --kt) {
        if (16 * kt >= F) continue;
        const int key = 16 * kt + r;
        f32x4 s[2], dp[2];
        s[1] = f32x4{0, 0, 0, 0}; dp[1] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int tt = 0; tt < NTL; ++tt) {
            s[tt] = f32x4{0, 0, 0, 0}; dp[tt] = f32x4{0, 0, 0, 0};
            const int qrow = 16 * tt + r;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 qa = *reinterpret_cast<const bf16x8*>(Qimg + qrow * LDI + 32 * ks + 8 * g);
                const bf16x8 da = *reinterpret_cast<const bf16x8*>(Dimg + qrow * LDI + 32 * ks + 8 * g);
                tmf::mma(s[tt], qa, kf[kt][ks]);
                tmf::mma(dp[tt], da, vf[kt][ks]);
            }
            const int qb = 16 * tt + 4 * g;
            const float4 m4 = *reinterpret_cast<const float4*>(&st[0][qb]);
            const float4 i4 = *reinterpret_cast<const float4*>(&st[1][qb]);
            const float4 d4 = *reinterpret_cast<const float4*>(&st[2][qb]);
            const float mv[4] = {m4.x, m4.y, m4.z, m4.w}, iv[4] = {i4.x, i4.y, i4.z, i4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = (qb + j) < F && key < F;
                const float pv = ok ? fast_exp2(s[tt][j] * c - mv[j]) * iv[j] : 0.f;
                s[tt][j] = pv;                                            // P
                dp[tt][j] = pv * (dp[tt][j] - dv4[j]) * scale;            // dS
            }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 dvv = f32x4{0, 0, 0, 0}, dkk = f32x4{0, 0, 0, 0};
            tmf::mma_frames<NTL>(dvv, Dimg, LDI, 16 * dt, g, r, s[0], s[1]);
            tmf::mma_frames<NTL>(dkk, Qimg, LDI, 16 * dt, g, r, dp[0], dp[1]);
            if (diff) {
                const f32x4 raw = dkk;
                dkk = tmf::diff_adjoint_acc(raw, above[dt], key);
                above[dt] = raw;
            }
            if (key < F) {
                float a[4] = {dkk[0], dkk[1], dkk[2], dkk[3]}, bb[4] = {dvv[0], dvv[1], dvv[2], dvv[3]};
                store4(dkp + (long)key * sq + 16 * dt + 4 * g, a);
                store4(dvp + (long)key * sv + 16 * dt + 4 * g, bb);
            }
        }
    }
    }   // problems of this wavefront
}
