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
// ---- the frame difference of module.py:193, taken on the SCORES --------------------------------------------------
// q'[f] = q[f] - q[f-1], k'[f] = k[f] - k[f-1] for f >= 2 (frames 0, 1 unchanged) means S' = D S D^T with S = Q K^T of the
// UN-differenced rows and D the F x F difference operator: the wavefront differences its F x F score tile along both
// axes in fp32 (a handful of DPP moves and subtractions) instead of ~100 conversions on the bf16 operand fragments, the
// difference is exact (no second rounding of q' / k' to bf16), and the backward needs no adjoint on [F][DH] outputs: with
// dS = D^T dS' D the gradients dQ = dS K, dK = dS^T Q come out with respect to the un-differenced rows directly.
// A tile set x[NA][NB] of 16x16 accumulator tiles: element (a, b, j) of lane (g, r) has row index rho = 16a + 4g + j (the
// accumulator-row axis) and column index kappa = 16b + r (the lane axis).
__device__ __forceinline__ float lane_from(float v, int src_lane) {          // value of `v` in lane src_lane (LDS crossbar, no memory)
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
// y = D x along the lane axis: y[kappa] = x[kappa] - x[kappa - 1] for kappa >= 2
template <int NA, int NB>
__device__ __forceinline__ void diff_lanes(f32x4 (&x)[NA][NB], int r) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = NB - 1; b >= 0; --b)                      // top tile first: the tile below is still un-differenced
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned below = __float_as_uint(x[a][b > 0 ? b - 1 : 0][j]);
                const unsigned wrap = __builtin_amdgcn_update_dpp(0u, below, 0x121 /* row_ror:1: lane 0 <- lane 15 */, 0xf, 0xf, true);
                const unsigned pv = __builtin_amdgcn_update_dpp(wrap, __float_as_uint(x[a][b][j]), 0x111 /* row_shr:1; lane 0 keeps wrap */, 0xf, 0xf, false);
                if (16 * b + r >= 2) x[a][b][j] -= __uint_as_float(pv);
            }
}
// y = D^T x along the lane axis: y[kappa] = x[kappa] - x[kappa + 1] for kappa >= 1 (x beyond the last tile is zero)
template <int NA, int NB>
__device__ __forceinline__ void adj_lanes(f32x4 (&x)[NA][NB], int r) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)                           // bottom tile first: the tile above is still untouched
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned above = b + 1 < NB ? __float_as_uint(x[a][b + 1 < NB ? b + 1 : b][j]) : 0u;
                const unsigned wrap = __builtin_amdgcn_update_dpp(0u, above, 0x12F /* row_ror:15: lane 15 <- lane 0 */, 0xf, 0xf, true);
                const unsigned nx = __builtin_amdgcn_update_dpp(wrap, __float_as_uint(x[a][b][j]), 0x101 /* row_shl:1; lane 15 keeps wrap */, 0xf, 0xf, false);
                if (16 * b + r >= 1) x[a][b][j] -= __uint_as_float(nx);
            }
}
// y = D x along the accumulator-row axis: y[rho] = x[rho] - x[rho - 1] for rho >= 2.  Row rho - 1 of register j = 0 is
// register 3 of the lane 16 below (lane group g - 1), or -- for g = 0 -- register 3 of group 3 in the tile below.
template <int NA, int NB>
__device__ __forceinline__ void diff_rows(f32x4 (&x)[NA][NB], int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int a = NA - 1; a >= 0; --a) {
            const float send = (g == 3 && a > 0) ? x[a > 0 ? a - 1 : 0][b][3] : x[a][b][3];
            const float recv = lane_from(send, (lane - 16) & 63);
            const int rho0 = 16 * a + 4 * g;
            f32x4 y = x[a][b];
#pragma unroll
            for (int j = 3; j >= 1; --j)
                if (rho0 + j >= 2) y[j] = x[a][b][j] - x[a][b][j - 1];
            if (rho0 >= 2) y[0] = x[a][b][0] - recv;
            x[a][b] = y;
        }
}
// y = D^T x along the accumulator-row axis: y[rho] = x[rho] - x[rho + 1] for rho >= 1
template <int NA, int NB>
__device__ __forceinline__ void adj_rows(f32x4 (&x)[NA][NB], int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const float up0 = a + 1 < NA ? x[a + 1 < NA ? a + 1 : a][b][0] : 0.f;
            const float send = g == 0 ? up0 : x[a][b][0];           // group 0 answers group 3 of the tile below it
            const float recv = lane_from(send, (lane + 16) & 63);
            const int rho0 = 16 * a + 4 * g;
            f32x4 y = x[a][b];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (rho0 + j >= 1) y[j] = x[a][b][j] - x[a][b][j + 1];
            y[3] = x[a][b][3] - recv;                                // rho0 + 3 >= 1 always
            x[a][b] = y;
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
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    // the wavefront's problem -> (clip, position, head) on the SCALAR unit, in 32 bits (the host checks the count): from the
    // lane-valued threadIdx.x >> 6 in `long` these were three 64-bit divisions per lane on the vector ALU
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned prob = blockIdx.x * 4u + (unsigned)wave;
    if (prob >= (unsigned)B * P * heads) return;          // no workgroup-level synchronisation below
    const unsigned bpu = prob / (unsigned)heads, bu = bpu / (unsigned)P;
    const int h = (int)(prob - bpu * heads);
    const long b = bu, p = bpu - bu * (unsigned)P;
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
    bf16x8 kf[NTL][KS], qf[NTL][KS];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[t][ks] = tmf::row_frag(kp, sq, 16 * t + r, F, 32 * ks + 8 * g);
            qf[t][ks] = tmf::row_frag(qp, sq, 16 * t + r, F, 32 * ks + 8 * g);
        }
    tmf::wave_lds_fence();

    // S^T = K Q^T for every (key tile t, query tile u): rows = keys 16t + 4g + j, lanes = queries 16u + r
    f32x4 s[NTL][NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int u = 0; u < NTL; ++u) {
            s[t][u] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) tmf::mma(s[t][u], kf[t][ks], qf[u][ks]);
        }
    if (diff == 1) {                                       // S' = D S D^T (see the helpers above)
        tmf::diff_lanes<NTL, NTL>(s, r);
        tmf::diff_rows<NTL, NTL>(s, lane);
    }
    float inv_u[NTL];
#pragma unroll
    for (int u = 0; u < NTL; ++u) {
        inv_u[u] = 0.f;
        if (16 * u >= F) break;
        // softmax over keys: lane owns query r, keys 16t + 4g + j
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = (16 * t + 4 * g + j) < F ? s[t][u][j] * c : -INFINITY;
                s[t][u][j] = z;
                mx = fmaxf(mx, z);
            }
        mx = tmf::group_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pv = fast_exp2(s[t][u][j] - mx);
                s[t][u][j] = pv;
                sum += pv;
            }
        sum = tmf::group_sum(sum);
        inv_u[u] = 1.0f / sum;
    }
    // O^T = V^T P^T by 16-column blocks.  The output rows leave through the V image (column block 16 dt is dead once every
    // query tile's product has read it) and are stored as whole 16-byte chunks, 64 per instruction: F = 9, DH = 64 is two
    // stores (1 KiB + 128 B) instead of four 8-byte-per-lane stores with 36 lanes active (see the backward kernel).
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        f32x4 o[NTL];
#pragma unroll
        for (int u = 0; u < NTL; ++u) {
            o[u] = f32x4{0, 0, 0, 0};
            if (16 * u < F) tmf::mma_frames<NTL>(o[u], Vimg, LDI, 16 * dt, g, r, s[0][u], s[NTL - 1][u]);
        }
#pragma unroll
        for (int u = 0; u < NTL; ++u) {
            float ov[4] = {o[u][0] * inv_u[u], o[u][1] * inv_u[u], o[u][2] * inv_u[u], o[u][3] * inv_u[u]};
            store4(Vimg + (16 * u + r) * LDI + 16 * dt + 4 * g, ov);
        }
    }
    tmf::wave_lds_fence();
    {
        constexpr int VPR = DH / 8;
        const int n1 = F * VPR;
        for (int i0 = 0; i0 < n1; i0 += 64) {
            const int idx = i0 + lane;
            if (idx < n1) {
                const int row = idx / VPR, ch = idx % VPR;
                *reinterpret_cast<bf16x8*>(op + (long)row * so + ch * 8) = *reinterpret_cast<const bf16x8*>(Vimg + row * LDI + ch * 8);
            }
        }
    }
}

// Diagnostic builds of the backward kernel (tools/build_variant.sh <out.so> attn_temporal.hip -D...; tools/tattn_bench.py):
//   ISTVT_TATTN_NOSTORE  the output stores sit behind a condition that is never true at run time (the arithmetic stays):
//                        the floor without stores, 50-55 us at F = 9 -- how the 8-byte partial stores of round 3 were found;
//   ISTVT_TATTN_NT       non-temporal output stores: 85 -> 73 us alone, +0.08 ms per step in the model (not the default);
//   ISTVT_TB_WPE=n       wavefronts per SIMD the single-tile kernel is compiled and its grid is sized for (4; 3 and 2 measured
//                        within 4 %).  Stating it at all also stops the compiler parking values in AGPRs (176 v_accvgpr moves
//                        per problem with the default heuristic; removing them changed nothing measurable).
#ifdef ISTVT_TATTN_NOSTORE
#define TB_STORE8(p, v) do { if (scale < -1e30f) *reinterpret_cast<bf16x8*>(p) = (v); } while (0)
#elif defined(ISTVT_TATTN_NT)
#define TB_STORE8(p, v) __builtin_nontemporal_store((v), reinterpret_cast<bf16x8*>(p))
#else
#define TB_STORE8(p, v) (*reinterpret_cast<bf16x8*>(p) = (v))
#endif
#ifndef ISTVT_TB_WPE
#define ISTVT_TB_WPE 4
#endif
template <int DH, int NTL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NTL == 1 ? ISTVT_TB_WPE : 2, NTL == 1 ? ISTVT_TB_WPE : 2)))
void tattn_mfma_bwd_kernel(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ v,
                                                             const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqk,
                                                             bf16_t* __restrict__ dv, int B, int F, int P, int heads,
                                                             float scale, long ldqk, long ldv, long ldo, int diff) {
    constexpr int LDI = DH + tmf::IPAD, KS = DH / 32, DT = DH / 16;
    constexpr int IMG = 16 * NTL * LDI;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4][3 * IMG];
    __shared__ __attribute__((aligned(16))) float stat[4][3][16 * NTL];
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    // problem arithmetic on the scalar unit, in 32 bits (see the forward kernel): six 64-bit vector divisions per problem
    // otherwise, in a kernel that is bound by its vector issue
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned total = (unsigned)B * P * heads, nwaves = gridDim.x * 4u;
    unsigned prob = blockIdx.x * 4u + (unsigned)wave;
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
    auto fetch = [&](unsigned pr) {
        const unsigned bpu = pr / (unsigned)heads, bu = bpu / (unsigned)P;
        const int h = (int)(pr - bpu * heads);
        const long b = bu, pp = bpu - bu * (unsigned)P;
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
    const unsigned bpu = prob / (unsigned)heads, bu = bpu / (unsigned)P;
    const int h = (int)(prob - bpu * heads);
    const long b = bu, p = bpu - bu * (unsigned)P;
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
    bf16x8 kf[NTL][KS], vf[NTL][KS], qf[NTL][KS], dof[NTL][KS];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vf[t][ks] = nv[t][ks];
    tmf::wave_lds_fence();
    if (prob + nwaves < total) fetch(prob + nwaves);
    // Q, K, dO rows in fragment layout (row 16t + r, columns 32ks + 8g), from the images
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[t][ks] = *reinterpret_cast<const bf16x8*>(Kimg + (16 * t + r) * LDI + 32 * ks + 8 * g);
            qf[t][ks] = *reinterpret_cast<const bf16x8*>(Qimg + (16 * t + r) * LDI + 32 * ks + 8 * g);
            dof[t][ks] = *reinterpret_cast<const bf16x8*>(Dimg + (16 * t + r) * LDI + 32 * ks + 8 * g);
        }

    // ---- part 1: S^T, dP^T (keys on rows, queries on lanes) for every (key tile t, query tile u): statistics, dS, dQ
    {
        f32x4 s[NTL][NTL], dp[NTL][NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int u = 0; u < NTL; ++u) {
                s[t][u] = f32x4{0, 0, 0, 0}; dp[t][u] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    tmf::mma(s[t][u], kf[t][ks], qf[u][ks]);
                    tmf::mma(dp[t][u], vf[t][ks], dof[u][ks]);
                }
            }
        if (diff == 1) {
            tmf::diff_lanes<NTL, NTL>(s, r);
            tmf::diff_rows<NTL, NTL>(s, lane);
        }
#pragma unroll
        for (int u = 0; u < NTL; ++u) {
            const int q = 16 * u + r;
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < NTL; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float z = (16 * t + 4 * g + j) < F ? s[t][u][j] * c : -INFINITY;
                    s[t][u][j] = z;
                    mx = fmaxf(mx, z);
                }
            mx = tmf::group_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NTL; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float pv = fast_exp2(s[t][u][j] - mx);
                    s[t][u][j] = pv;
                    sum += pv;
                }
            sum = tmf::group_sum(sum);
            const float inv = 1.0f / sum;
            float dl = 0.f;
#pragma unroll
            for (int t = 0; t < NTL; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[t][u][j] *= inv; dl += s[t][u][j] * dp[t][u][j]; }
            dl = tmf::group_sum(dl);
#pragma unroll
            for (int t = 0; t < NTL; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[t][u][j] = s[t][u][j] * (dp[t][u][j] - dl) * scale;      // dS'^T (query rows >= F: dO = 0 -> 0)
            if (g == 0) { st[0][q] = mx; st[1][q] = inv; st[2][q] = dl; }                           // rows q >= F: never used (p = 0 there)
        }
        if (diff == 1) {                                   // dS = D^T dS' D: the gradient w.r.t. the UN-differenced scores
            tmf::adj_lanes<NTL, NTL>(s, r);
            tmf::adj_rows<NTL, NTL>(s, lane);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 dq[1][NTL];
#pragma unroll
            for (int u = 0; u < NTL; ++u) {
                dq[0][u] = f32x4{0, 0, 0, 0};
                if (16 * u < F) tmf::mma_frames<NTL>(dq[0][u], Kimg, LDI, 16 * dt, g, r, s[0][u], s[NTL - 1][u]);
            }
            // diff == 2: q', k' arrived differenced (K' is what the images hold), dQ' = dS' K' is the gradient w.r.t. q';
            // the caller wants it w.r.t. the un-differenced projection: dQ = D^T dQ' along the frames (= the lanes here)
            if (diff == 2) tmf::adj_lanes<1, NTL>(dq, r);
            // Output rows leave through the images (see the end of the problem): this column block of K is dead once
            // every query tile's product has read it, and part 2 does not read the K image at all
#pragma unroll
            for (int u = 0; u < NTL; ++u) {
                float o[4] = {dq[0][u][0], dq[0][u][1], dq[0][u][2], dq[0][u][3]};
                store4(Kimg + (16 * u + r) * LDI + 16 * dt + 4 * g, o);
            }
        }
    }
    tmf::wave_lds_fence();

    // ---- part 2: S, dP (queries on rows, keys on lanes) for every (query tile tt, key tile kt) -> dV, dK
    {
        f32x4 s[NTL][NTL], dp[NTL][NTL];
#pragma unroll
        for (int tt = 0; tt < NTL; ++tt)
#pragma unroll
            for (int kt = 0; kt < NTL; ++kt) {
                s[tt][kt] = f32x4{0, 0, 0, 0}; dp[tt][kt] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    tmf::mma(s[tt][kt], qf[tt][ks], kf[kt][ks]);
                    tmf::mma(dp[tt][kt], dof[tt][ks], vf[kt][ks]);
                }
            }
        if (diff == 1) {
            tmf::diff_lanes<NTL, NTL>(s, r);
            tmf::diff_rows<NTL, NTL>(s, lane);
        }
#pragma unroll
        for (int tt = 0; tt < NTL; ++tt) {
            const int qb = 16 * tt + 4 * g;
            const float4 m4 = *reinterpret_cast<const float4*>(&st[0][qb]);
            const float4 i4 = *reinterpret_cast<const float4*>(&st[1][qb]);
            const float4 d4 = *reinterpret_cast<const float4*>(&st[2][qb]);
            const float mv[4] = {m4.x, m4.y, m4.z, m4.w}, iv[4] = {i4.x, i4.y, i4.z, i4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int kt = 0; kt < NTL; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = (qb + j) < F && (16 * kt + r) < F;
                    const float pv = ok ? fast_exp2(s[tt][kt][j] * c - mv[j]) * iv[j] : 0.f;
                    s[tt][kt][j] = pv;                                            // P
                    dp[tt][kt][j] = pv * (dp[tt][kt][j] - dv4[j]) * scale;        // dS'
                }
        }
        if (diff == 1) {
            tmf::adj_lanes<NTL, NTL>(dp, r);
            tmf::adj_rows<NTL, NTL>(dp, lane);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 dkk[1][NTL], dvv[NTL];
#pragma unroll
            for (int kt = 0; kt < NTL; ++kt) {
                dkk[0][kt] = f32x4{0, 0, 0, 0};
                dvv[kt] = f32x4{0, 0, 0, 0};
                if (16 * kt >= F) continue;
                tmf::mma_frames<NTL>(dvv[kt], Dimg, LDI, 16 * dt, g, r, s[0][kt], s[NTL - 1][kt]);
                tmf::mma_frames<NTL>(dkk[0][kt], Qimg, LDI, 16 * dt, g, r, dp[0][kt], dp[NTL - 1][kt]);
            }
            if (diff == 2) tmf::adj_lanes<1, NTL>(dkk, r);         // dK = D^T dK' (see dQ above)
            // column block 16 dt of the dO and Q images is dead now (every key tile's products have read it): dV and dK
            // rows take its place
#pragma unroll
            for (int kt = 0; kt < NTL; ++kt) {
                float bb[4] = {dvv[kt][0], dvv[kt][1], dvv[kt][2], dvv[kt][3]};
                float a[4] = {dkk[0][kt][0], dkk[0][kt][1], dkk[0][kt][2], dkk[0][kt][3]};
                store4(Dimg + (16 * kt + r) * LDI + 16 * dt + 4 * g, bb);
                store4(Qimg + (16 * kt + r) * LDI + 16 * dt + 4 * g, a);
            }
        }
    }
    // ---- the three gradients leave as whole 16-byte chunks of their rows: 3 F rows x DH / 8 chunks, 64 per instruction
    // (F = 9, DH = 64: 4 stores of 1 KiB, 1 KiB, 1 KiB, 384 B).  Straight from the accumulator layout they were 12 stores
    // of 8 bytes per lane with 36 lanes active (288 B each), and the kernel spent more than half its time on them:
    // 109.6 us with, 50.2 us without its stores (tools/tattn_bench.py, -DISTVT_TATTN_NOSTORE).
    tmf::wave_lds_fence();
    {
        const int n1 = F * VPR;                            // chunks per matrix; image m: 0 = Q (dK), 1 = K (dQ), 2 = dO (dV)
        // (a real loop: unrolled, with every LDS read ahead of the first store, it measured 85 -> 105 us at F = 9)
        for (int i0 = 0; i0 < 3 * n1; i0 += 64) {
            const int idx = i0 + lane;
            if (idx < 3 * n1) {
                const int m = (idx >= n1) + (idx >= 2 * n1), rem = idx - m * n1, row = rem / VPR, ch = rem % VPR;
                const bf16x8 val = *reinterpret_cast<const bf16x8*>(Qimg + m * IMG + row * LDI + ch * 8);
                bf16_t* dst = (m == 0 ? dkp : m == 1 ? dqp : dvp) + (long)row * (m == 2 ? sv : sq) + ch * 8;
                TB_STORE8(dst, val);
            }
        }
    }
    }   // problems of this wavefront
}
