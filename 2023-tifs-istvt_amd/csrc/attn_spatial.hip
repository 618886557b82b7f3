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
// Kernels in this file (bfloat16; float32 runs the same templates with 4 wavefronts x 32 queries):
//   * sattn_fwd_kernel: 8 wavefronts x 16 queries per workgroup, keys stream through two LDS images in chunks of 128 with
//     online-softmax rescaling (any P: 197 -> 2 chunks, 362 -> 3).  RES variant (128 < P <= 256): ONE workgroup per
//     (frame, head) keeps all keys / values resident and walks the query blocks itself.
//   * attn_spatial_pers.h (round 4): the persistent forward for 128 < P <= 240 -- one 16-wavefront workgroup per CU walks
//     the (frame, head) problems, the next problem's K / V arriving by LDS-DMA under the current one's arithmetic.
//   * backward, recomputing P from the saved (row max, 1 / row sum):
//       sattn_bwd_dq  : the forward's geometry; dS^T = P^T o (dP^T - delta); dQ^T += K^T dS^T
//       sattn_bwd_dkv : one wavefront owns 32 keys, queries stream through LDS; dV^T += dO^T P; dK^T += Q^T dS
//       sattn_bwd_fused_kernel (RES, round 3): both parts in ONE workgroup per (frame, head) -- dQ with K / V in LDS, a
//       barrier, dK / dV with Q / dO in the same bytes, the second part's images written from the fragments the first part
//       loaded: every operand is read from global memory once.
#include "common.h"
#include <type_traits>
#include <cstdlib>

constexpr int CHUNK = 128;          // rows of an LDS image
constexpr int NT = CHUNK / 16;      // 16-row tiles per chunk
// Row pitch of an LDS image, in elements.  bfloat16: DH + 16 (DH = 64: 160 bytes = 40 banks): eight consecutive rows
// then start on the eight distinct multiples of 8 banks, so a ds_read_b64_tr_b16 half-wave (8 rows x 32 bytes) and a
// ds_read_b128 lane group (16 rows x 16 bytes) each touch all 64 banks exactly once.  DH + 8 (144 bytes = 36 banks) put
// row 7 on top of row 0's first four banks: every transposed read and most row reads were 2-way conflicts (PMC round 2:
// SQ_LDS_BANK_CONFLICT = 43 % of the LDS cycles of these kernels).  float32 keeps DH + 8 (72 dwords = 8 mod 64).
#ifdef ISTVT_SATTN_NOSTORE         // diagnostic: what do the backward's 8-byte output stores cost?
#define SB_STORE4(p, v) do { if (scale < -1e30f) store4(p, v); } while (0)
#else
#define SB_STORE4(p, v) store4(p, v)
#endif
template <typename T, int DH> struct Pitch { static constexpr int v = DH + (sizeof(T) == 2 ? 16 : 8); };
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// copy rows [row0, row0+CHUNK) x DH columns of a [rows][ld] matrix into an LDS image, zero-filling
// rows >= nrows
template <typename T, int DH, int NTHR = 256>
__device__ __forceinline__ void stage_img(T* img, const T* __restrict__ src, long ld, int row0, int nrows, int tid) {
    constexpr int LDI = Pitch<T, DH>::v;
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
__device__ __forceinline__ void stage_commit(T* img, const StageRegs<T, DH, NTHR>& rg, int tid, int rlim = CHUNK) {
    constexpr int LDI = Pitch<T, DH>::v;
    constexpr int VPR = DH / 8;
    constexpr int NV = CHUNK * VPR / NTHR;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + NTHR * i;
        const int row = v / VPR, col = (v % VPR) * 8;
        if (row >= rlim) continue;                       // rows the image does not have (RES: images of img_rows rows)
        if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<bf16x8*>(img + row * LDI + col) = rg.f[i];
        } else {
            float* p = (float*)(img + row * LDI + col);
            *reinterpret_cast<float4*>(p) = make_float4(rg.f[i].v[0], rg.f[i].v[1], rg.f[i].v[2], rg.f[i].v[3]);
            *reinterpret_cast<float4*>(p + 4) = make_float4(rg.f[i].v[4], rg.f[i].v[5], rg.f[i].v[6], rg.f[i].v[7]);
        }
    }
}

// RES mode staging: rows [0, P) of two matrices into NCH x 128-row images each, every global load issued before the
// first LDS store (stage_img's load -> store pairs made the two matrices two dependent memory round trips)
template <typename T, int DH, int NTHR, int NCH>
__device__ __forceinline__ void stage_all2(T* imgA, const T* __restrict__ srcA, long ldA, T* imgB, const T* __restrict__ srcB,
                                           long ldB, int P, int tid, int img_rows) {
    constexpr int LDI = Pitch<T, DH>::v;
    StageRegs<T, DH, NTHR> ra[NCH], rb[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch * CHUNK < P) {
            stage_fetch(ra[ch], srcA, ldA, ch * CHUNK, P, tid);
            stage_fetch(rb[ch], srcB, ldB, ch * CHUNK, P, tid);
        }
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch * CHUNK < P) {
            stage_commit(imgA + ch * CHUNK * LDI, ra[ch], tid, img_rows - ch * CHUNK);
            stage_commit(imgB + ch * CHUNK * LDI, rb[ch], tid, img_rows - ch * CHUNK);
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

__device__ __forceinline__ float group_max(float v) { return xgroup_max(v); }      // over the 4 lane groups (same r)
__device__ __forceinline__ float group_sum(float v) { return xgroup_sum(v); }

// ------------------------------------------------------------------------------------------
// U = query tiles (of 16) per wavefront, 128 / (16 U) wavefronts per workgroup.  U = 1 (8 wavefronts) halves the
// registers of a wavefront: 4 instead of 2 wavefronts per SIMD, whose softmax (VALU) and MFMA phases then overlap and
// whose staging latencies hide each other (the kernel is bound by neither pipe: it waits).
// RES (resident keys, P <= RES_CHUNKS * 128): ONE workgroup per (frame, head) stages every key / value row once and walks
// the query blocks itself, instead of ceil(P / 128) workgroups that each stage all keys (P = 197: two workgroups, K and
// V fetched twice, 1.57x the algorithmic bytes by the PMC pass; the kernel is bound by that traffic).  After the one
// barrier behind the staging the wavefronts run independently: no chunk barriers, no re-staging.
// Round 5: RES is the number of resident 128-row chunks, 2 (P <= 256: 224^2 inputs, P = 197) or 3 (P <= 384: the reference's
// own 300^2 geometry, P = 362 -- two images of 384 rows = 120 KiB, one workgroup per CU).  Three chunks are BUILT and tested
// (-DISTVT_SATTN_RES_MAX=3) but not dispatched: measured at the reference's geometry (112 frames x 8 heads, P = 362, same
// box, alternating; profiles/r05_b_*): forward 90.4 us resident against 81.2 us on the chunked kernels (three workgroups per
// problem, K / V fetched three times but 4 workgroups per CU instead of 1), backward 236 against 233 us; P = 300: 68 vs 61,
// 180 vs 189.  The chunked kernels already run P = 362 at the TFLOP/s of P = 197 (330-380 forward, 320 backward).
#ifndef ISTVT_SATTN_RES_MAX
#define ISTVT_SATTN_RES_MAX 2
#endif
constexpr int RES_CHUNKS_MAX = ISTVT_SATTN_RES_MAX;
// RES kernels: the two LDS images hold img_rows = P rounded up to 32 rows (zero-filled past P), in dynamic shared memory
// sized by the host -- P = 197: 2 x 224 rows x 160 bytes = 70 KiB, two workgroups per CU (256-row images at this pitch
// would be 80 KiB + statistics: one).
extern __shared__ __attribute__((aligned(16))) char sattn_dyn[];
__host__ __device__ inline int res_img_rows(int P) { return (P + 31) & ~31; }

template <typename T, int DH, int U, bool FP8 = false, int RES = 0>
__global__ __launch_bounds__(512 / U, (U == 1 && RES < 3) ? 4 : 1) void sattn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                        float* __restrict__ lse, int P, int heads, float scale, long ldqkv, long ldo) {
    constexpr int LDI = Pitch<T, DH>::v, KS = DH / 32, DT = DH / 16;
    __shared__ __attribute__((aligned(16))) T smem_st[RES ? 8 : 2 * CHUNK * LDI];
    const int img_rows = RES ? res_img_rows(P) : CHUNK;
    T* Kimg = RES ? reinterpret_cast<T*>(sattn_dyn) : smem_st;
    T* Vimg = Kimg + img_rows * LDI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
    const int prob = blockIdx.y, h = prob % heads, bf = prob / heads;
    const int inner = heads * DH;
    const long ld = ldqkv;
    const T* base = qkv + (long)bf * P * ld;
#ifdef ISTVT_SATTN_DIAG
    // diagnostic: every workgroup reads the operands of (frame 0, head 0) -- all cache hits: what the kernel takes
    // when no operand comes from HBM (1: K and V, 2: also Q)
    const T* base_kv = qkv;
    const T* qp = (ISTVT_SATTN_DIAG >= 2 ? qkv : base + h * DH);
    const T* kp = base_kv + inner;
    const T* vp = base_kv + 2 * inner;
#else
    const T* qp = base + h * DH;
    const T* kp = base + inner + h * DH;
    const T* vp = base + 2 * inner + h * DH;
#endif
    const float c = scale * LOG2E;
    constexpr int NBLK = RES ? RES : 1;
    // the query rows of EVERY block this wavefront will own are requested first: in RES mode their latency then hides
    // behind the staging (a global-memory round trip under load is ~2 us: one per block was a third of a wavefront's life)
    typename Mma<T>::frag qf_pre[NBLK][U][KS];
#pragma unroll
    for (int qi = 0; qi < NBLK; ++qi)
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qf_pre[qi][u][ks] = row_frag<T>(qp, ld, (RES ? qi : (int)blockIdx.x) * 128 + wave * 16 * U + 16 * u + r, P,
                                                32 * ks + 8 * g);
    if constexpr (RES) {
        stage_all2<T, DH, 512 / U, RES>(Kimg, kp, ld, Vimg, vp, ld, P, tid, img_rows);
        __syncthreads();
    }
#pragma unroll
  for (int qi = 0; qi < NBLK; ++qi) {
    const int qblk = RES ? qi : (int)blockIdx.x;
    const int q0 = qblk * 128 + wave * 16 * U;
    const bool active = q0 < P;
    if (RES && !active) break;                       // wave-uniform; nothing below synchronises in RES mode

    typename Mma<T>::frag qf[U][KS];
    long q8[U][KS];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[u][ks] = qf_pre[qi][u][ks];
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
        const T* const Kimg_c = RES ? Kimg + (c0 / CHUNK) * CHUNK * LDI : Kimg;
        const T* const Vimg_c = RES ? Vimg + (c0 / CHUNK) * CHUNK * LDI : Vimg;
        f32x4 s[NT][U];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int u = 0; u < U; ++u) s[t][u] = f32x4{0, 0, 0, 0};
            if (!TAIL || c0 + 16 * t < P) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    typename Mma<T>::frag kf = frag_load(Kimg_c + (16 * t + r) * LDI + 32 * ks + 8 * g);
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
            // (TAIL: tiles wholly past P take no part -- their s stays 0 for the PV product --, and only the ONE tile that
            //  straddles P is masked: both conditions are wave-uniform, so the row end costs branches, not a compare and
            //  a select per score of the whole chunk)
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (TAIL && c0 + 16 * t >= P) continue;
                if (TAIL && c0 + 16 * t + 16 > P) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[t][u][j] = (c0 + 16 * t + 4 * g + j < P) ? s[t][u][j] : -INFINITY;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, s[t][u][j]);
            }
            mx = group_max(mx) * c;
            const float m_new = fmaxf(m_run[u], mx);
            const float alpha = fast_exp2(m_run[u] - m_new);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (TAIL && c0 + 16 * t >= P) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float pv = fast_exp2(fmaf(s[t][u][j], c, -m_new));
                    s[t][u][j] = pv;
                    sum += pv;
                }
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
                        frag_load_tr(Vimg_c, LDI, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, 16 * dt, r);
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
    if constexpr (RES) {
        for (int c0 = 0; c0 < P; c0 += CHUNK) {
            if (c0 + CHUNK > P) chunk(c0, std::true_type{});
            else chunk(c0, std::false_type{});
        }
    } else {
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
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = q0 + 16 * u + r;
        if (q >= P) continue;
        const float inv = 1.0f / l_run[u];
        const float oinv = FP8 ? inv * (1.0f / P8_SCALE) : inv;
        T* op = out + ((long)bf * P + q) * ldo + h * DH;
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
  }   // query blocks
}

// ------------------------------------------------------------------------------------------
// backward part 1: delta = rowsum(dO o O), dQ
// The query / dO row fragments a wavefront loaded for its dQ blocks: together the wavefronts of a workgroup hold every
// row of Q and dO, so the fused backward writes the Q / dO images of its second part from them instead of reading global
// memory a second time.
template <typename T, int DH, int U, int NBLK> struct SattnKeep {
    typename Mma<T>::frag q[NBLK][U][DH / 32], d[NBLK][U][DH / 32];
};

// FUSED (sattn_bwd_fused_kernel): delta goes to the LDS row the dK / dV part reads it from instead of to global memory
template <typename T, int DH, int U, bool FP8 = false, int RES = 0, bool FUSED = false>
__device__ __forceinline__ void sattn_dq_body(const T* __restrict__ qkv, const T* __restrict__ out,
                                              const T* __restrict__ dout, const float* __restrict__ lse,
                                              float* __restrict__ delta, T* __restrict__ dqkv, int P,
                                              int heads, float scale, long ldqkv, long ldo,
                                              SattnKeep<T, DH, U, RES ? RES : 1>* keep = nullptr) {
    static_assert(!FUSED || RES, "the fused backward is the keys-resident form");
    constexpr int LDI = Pitch<T, DH>::v, KS = DH / 32, DT = DH / 16;
    __shared__ __attribute__((aligned(16))) T smem_st[RES ? 8 : 2 * CHUNK * LDI];
    const int img_rows = RES ? res_img_rows(P) : CHUNK;
    T* Kimg = RES ? reinterpret_cast<T*>(sattn_dyn) : smem_st;
    T* Vimg = Kimg + img_rows * LDI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
    const int prob = blockIdx.y, h = prob % heads, bf = prob / heads;
    const int inner = heads * DH;
    const long ld = ldqkv;
    const T* base = qkv + (long)bf * P * ld;
    const T* qp = base + h * DH;
    const T* kp = base + inner + h * DH;
    const T* vp = base + 2 * inner + h * DH;
    const T* op = out + (long)bf * P * ldo + h * DH;
    const T* dop = dout + (long)bf * P * ldo + h * DH;
    const float c = scale * LOG2E;
    constexpr int NBLK = RES ? RES : 1;
    // rows of every query block this wavefront will own, requested before the staging (see sattn_fwd_kernel)
    typename Mma<T>::frag qf_pre[NBLK][U][KS], dof_pre[NBLK][U][KS], of_pre[NBLK][U][KS];
    float2 st_pre[NBLK][U];
#pragma unroll
    for (int qi = 0; qi < NBLK; ++qi)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = (RES ? qi : (int)blockIdx.x) * 128 + wave * 16 * U + 16 * u + r;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                qf_pre[qi][u][ks] = row_frag<T>(qp, ld, q, P, 32 * ks + 8 * g);
                dof_pre[qi][u][ks] = row_frag<T>(dop, ldo, q, P, 32 * ks + 8 * g);
                of_pre[qi][u][ks] = row_frag<T>(op, ldo, q, P, 32 * ks + 8 * g);
                if constexpr (FUSED) { keep->q[qi][u][ks] = qf_pre[qi][u][ks]; keep->d[qi][u][ks] = dof_pre[qi][u][ks]; }
            }
            st_pre[qi][u] = q < P ? reinterpret_cast<const float2*>(lse)[((long)bf * P + q) * heads + h] : make_float2(0.f, 0.f);
        }
    if constexpr (RES) {                               // every key / value row staged once (see sattn_fwd_kernel)
        stage_all2<T, DH, 512 / U, RES>(Kimg, kp, ld, Vimg, vp, ld, P, tid, img_rows);
        __syncthreads();
    }
#pragma unroll
  for (int qi = 0; qi < NBLK; ++qi) {
    const int qblk = RES ? qi : (int)blockIdx.x;
    const int q0 = qblk * 128 + wave * 16 * U;
    const bool active = q0 < P;
    if (RES && !active) break;

    typename Mma<T>::frag qf[U][KS], dof[U][KS];
    long q8[U][KS];
    float lq[U], li[U], dl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = q0 + 16 * u + r;
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[u][ks] = qf_pre[qi][u][ks];
            if constexpr (FP8) q8[u][ks] = to_fp8(qf[u][ks]);        // S is recomputed exactly as the forward computed it
            dof[u][ks] = dof_pre[qi][u][ks];
#pragma unroll
            for (int i = 0; i < 8; ++i) part += Mma<T>::get(dof[u][ks], i) * Mma<T>::get(of_pre[qi][u][ks], i);   // rows past P are zero
        }
        dl[u] = group_sum(part);
        const float2 st = st_pre[qi][u];
        // exponent offset with 1/rowsum folded in: p = exp2(s c - (max - log2(1/sum))); rows past P have 1/sum = 0 ->
        // offset +inf -> p = 0
        lq[u] = st.x - __builtin_amdgcn_logf(st.y); li[u] = st.y;
        if constexpr (FUSED) {
            // (rows P .. of an active wavefront are zero rows: dl = 0, as the dK / dV part wants them)
            float* stat1 = reinterpret_cast<float*>(Kimg + 2 * img_rows * LDI) + img_rows;
            if (g == 0 && q < img_rows) stat1[q] = dl[u];
        } else {
            if (q < P && g == 0) delta[((long)bf * P + q) * heads + h] = dl[u];
        }
    }

    f32x4 dq[DT][U];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int u = 0; u < U; ++u) dq[dt][u] = f32x4{0, 0, 0, 0};

    StageRegs<T, DH, 512 / U> kreg, vreg;
    if constexpr (!RES) {
        stage_fetch(kreg, kp, ld, 0, P, tid);
        stage_fetch(vreg, vp, ld, 0, P, tid);
    }
    for (int c0 = 0; c0 < P; c0 += CHUNK) {
        if constexpr (!RES) {
            if (c0) __syncthreads();
            stage_commit(Kimg, kreg, tid);
            stage_commit(Vimg, vreg, tid);
            __syncthreads();
            if (c0 + CHUNK < P) {
                stage_fetch(kreg, kp, ld, c0 + CHUNK, P, tid);
                stage_fetch(vreg, vp, ld, c0 + CHUNK, P, tid);
            }
            if (!active) continue;
        }
        const T* const Kimg_c = RES ? Kimg + (c0 / CHUNK) * CHUNK * LDI : Kimg;
        const T* const Vimg_c = RES ? Vimg + (c0 / CHUNK) * CHUNK * LDI : Vimg;
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
                // only the 16-key tile that holds the row end needs its scores masked (wave-uniform): a compare and a
                // select per score of the whole tail chunk were 12 % of this kernel's instructions
                const bool straddle = tail && c0 + 32 * ss + 16 * tt + 16 > P;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    typename Mma<T>::frag kf = frag_load(Kimg_c + krow * LDI + 32 * ks + 8 * g);
                    typename Mma<T>::frag vf = frag_load(Vimg_c + krow * LDI + 32 * ks + 8 * g);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if constexpr (FP8) mma8(s[tt][u], to_fp8(kf), q8[u][ks]);
                        else Mma<T>::mma(s[tt][u], kf, qf[u][ks]);
                        Mma<T>::mma(dp[tt][u], vf, dof[u][ks]);
                    }
                }
                auto softmax_bwd = [&](auto masked) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float pv = fast_exp2(fmaf(s[tt][u][j], c, -lq[u]));
                            if (decltype(masked)::value && c0 + 32 * ss + 16 * tt + 4 * g + j >= P) pv = 0.f;
                            s[tt][u][j] = pv * (dp[tt][u][j] - dl[u]);           // dS^T / scale (applied to dQ below)
                        }
                };
                if (straddle) softmax_bwd(std::true_type{});
                else softmax_bwd(std::false_type{});
            }
            typename Mma<T>::frag dsf[U];
#pragma unroll
            for (int u = 0; u < U; ++u) dsf[u] = acc_frag<T>(s[0][u], s[1][u]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                typename Mma<T>::frag kt = frag_load_tr(Kimg_c, LDI, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, 16 * dt, r);
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
            SB_STORE4(dqp + 16 * dt + 4 * g, v);
        }
    }
  }   // query blocks
}

template <typename T, int DH, int U, bool FP8 = false, int RES = 0>
__global__ __launch_bounds__(512 / U) void sattn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                           const T* __restrict__ dout, const float* __restrict__ lse,
                                                           float* __restrict__ delta, T* __restrict__ dqkv, int P,
                                                           int heads, float scale, long ldqkv, long ldo) {
    sattn_dq_body<T, DH, U, FP8, RES, false>(qkv, out, dout, lse, delta, dqkv, P, heads, scale, ldqkv, ldo);
}

// ------------------------------------------------------------------------------------------
// backward part 2: dK, dV.  Wave owns 32 keys; queries stream through LDS.
template <typename T, int DH, int U, bool FP8 = false, int RES = 0, bool FUSED = false>
__device__ __forceinline__ void sattn_dkv_body(const T* __restrict__ qkv, const T* __restrict__ dout,
                                               const float* __restrict__ lse,
                                               const float* __restrict__ delta, T* __restrict__ dqkv,
                                               int P, int heads, float scale, long ldqkv, long ldo,
                                               const SattnKeep<T, DH, U, RES ? RES : 1>* keep = nullptr) {
    constexpr int LDI = Pitch<T, DH>::v, KS = DH / 32, DT = DH / 16;
    __shared__ __attribute__((aligned(16))) T smem_st[RES ? 8 : 2 * CHUNK * LDI];
    __shared__ __attribute__((aligned(16))) float stat_st[RES ? 4 : 2 * CHUNK];
    const int img_rows = RES ? res_img_rows(P) : CHUNK;
    T* Qimg = RES ? reinterpret_cast<T*>(sattn_dyn) : smem_st;
    T* Dimg = Qimg + img_rows * LDI;
    float* stat0 = RES ? reinterpret_cast<float*>(Dimg + img_rows * LDI) : stat_st;     // exponent offsets, then deltas
    float* stat1 = stat0 + img_rows;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
    const int prob = blockIdx.y, h = prob % heads, bf = prob / heads;
    const int inner = heads * DH;
    const long ld = ldqkv;
    const T* base = qkv + (long)bf * P * ld;
    const T* qp = base + h * DH;
    const T* kp = base + inner + h * DH;
    const T* vp = base + 2 * inner + h * DH;
    const T* dop = dout + (long)bf * P * ldo + h * DH;
    const float c = scale * LOG2E;
    constexpr int NBLK = RES ? RES : 1;
    // key / value rows of every block this wavefront will own, requested before the staging (see sattn_fwd_kernel)
    typename Mma<T>::frag kf_pre[NBLK][U][KS], vf_pre[NBLK][U][KS];
#pragma unroll
    for (int ki = 0; ki < NBLK; ++ki)
#pragma unroll
        for (int kt = 0; kt < U; ++kt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int key = (RES ? ki : (int)blockIdx.x) * 128 + wave * 16 * U + 16 * kt + r;
                if constexpr (FUSED) {
                    // the dQ part left the K / V images of this (frame, head) in the bytes Q / dO are about to take:
                    // the wavefront's own key rows come from there, not from global memory again
                    const int kr = min(key, img_rows - 1);
                    typename Mma<T>::frag kk = frag_load(Qimg + kr * LDI + 32 * ks + 8 * g);
                    typename Mma<T>::frag vv = frag_load(Dimg + kr * LDI + 32 * ks + 8 * g);
                    if (key >= P) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { Mma<T>::set(kk, i, 0.f); Mma<T>::set(vv, i, 0.f); }
                    }
                    kf_pre[ki][kt][ks] = kk;
                    vf_pre[ki][kt][ks] = vv;
                } else {
                    kf_pre[ki][kt][ks] = row_frag<T>(kp, ld, key, P, 32 * ks + 8 * g);
                    vf_pre[ki][kt][ks] = row_frag<T>(vp, ld, key, P, 32 * ks + 8 * g);
                }
            }
    if constexpr (FUSED) __syncthreads();              // all key rows are in registers: the images may be overwritten
    if constexpr (RES) {                               // every query / dO row and its statistics staged once
        if constexpr (FUSED) {
            // from the registers of the dQ part (rows past P are zero fragments): no second read of q and dO
#pragma unroll
            for (int qi = 0; qi < NBLK; ++qi)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int row = qi * 128 + wave * 16 * U + 16 * u + r;
                    if (row < img_rows) {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            *reinterpret_cast<typename Mma<T>::frag*>(Qimg + row * LDI + 32 * ks + 8 * g) = keep->q[qi][u][ks];
                            *reinterpret_cast<typename Mma<T>::frag*>(Dimg + row * LDI + 32 * ks + 8 * g) = keep->d[qi][u][ks];
                        }
                    }
                }
        } else {
            stage_all2<T, DH, 512 / U, RES>(Qimg, qp, ld, Dimg, dop, ldo, P, tid, img_rows);
        }
        for (int q = tid; q < img_rows; q += 512 / U) {
            const float2 st = q < P ? reinterpret_cast<const float2*>(lse)[((long)bf * P + q) * heads + h] : make_float2(0.f, 0.f);
            stat0[q] = st.x - __builtin_amdgcn_logf(st.y);       // exponent offset incl. log2(1/sum); +inf for padding
            if constexpr (FUSED) { if (q >= P) stat1[q] = 0.f; }       // rows < P: written by the dQ part of this workgroup
            else stat1[q] = q < P ? delta[((long)bf * P + q) * heads + h] : 0.f;
        }
        __syncthreads();
    }
#pragma unroll
  for (int ki = 0; ki < NBLK; ++ki) {
    const int kblk = RES ? ki : (int)blockIdx.x;
    const int k0 = kblk * 128 + wave * 16 * U;
    const bool active = k0 < P;
    if (RES && !active) break;

    typename Mma<T>::frag kf[U][KS], vf[U][KS];
    long k8[U][KS];
#pragma unroll
    for (int kt = 0; kt < U; ++kt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[kt][ks] = kf_pre[ki][kt][ks];
            if constexpr (FP8) k8[kt][ks] = to_fp8(kf[kt][ks]);
            vf[kt][ks] = vf_pre[ki][kt][ks];
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
        stage_fetch(dreg, dop, ldo, c0, P, tid);
        if (tid < CHUNK) {
            const int q = c0 + tid;
            streg = q < P ? reinterpret_cast<const float2*>(lse)[((long)bf * P + q) * heads + h]
                          : make_float2(0.f, 0.f);                  // 1/l = 0 masks padded query rows
            dlreg = q < P ? delta[((long)bf * P + q) * heads + h] : 0.f;
        }
    };
    if constexpr (!RES) fetch(0);
    for (int c0 = 0; c0 < P; c0 += CHUNK) {
        if constexpr (!RES) {
            if (c0) __syncthreads();
            stage_commit(Qimg, qreg, tid);
            stage_commit(Dimg, dreg, tid);
            if (tid < CHUNK) {
                stat0[tid] = streg.x - __builtin_amdgcn_logf(streg.y);    // exponent offset incl. log2(1/sum); +inf for padding
                stat1[tid] = dlreg;
            }
            __syncthreads();
            if (c0 + CHUNK < P) fetch(c0 + CHUNK);
            if (!active) continue;
        }
        const T* const Qimg_c = RES ? Qimg + (c0 / CHUNK) * CHUNK * LDI : Qimg;
        const T* const Dimg_c = RES ? Dimg + (c0 / CHUNK) * CHUNK * LDI : Dimg;
        const float* const st0 = RES ? stat0 + c0 : stat0;
        const float* const st1 = RES ? stat1 + c0 : stat1;
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
                    typename Mma<T>::frag qa = frag_load(Qimg_c + qrow * LDI + 32 * ks + 8 * g);
                    typename Mma<T>::frag da = frag_load(Dimg_c + qrow * LDI + 32 * ks + 8 * g);
#pragma unroll
                    for (int kt = 0; kt < U; ++kt) {
                        if constexpr (FP8) mma8(s[tt][kt], to_fp8(qa), k8[kt][ks]);
                        else Mma<T>::mma(s[tt][kt], qa, kf[kt][ks]);
                        Mma<T>::mma(dp[tt][kt], da, vf[kt][ks]);
                    }
                }
                const int qb = 32 * ss + 16 * tt + 4 * g;
                const float4 l4 = *reinterpret_cast<const float4*>(st0 + qb);
                const float4 d4 = *reinterpret_cast<const float4*>(st1 + qb);
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
                typename Mma<T>::frag dot_ = frag_load_tr(Dimg_c, LDI, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, 16 * dt, r);
                typename Mma<T>::frag qt_ = frag_load_tr(Qimg_c, LDI, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, 16 * dt, r);
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
            SB_STORE4(dkp + 16 * dt + 4 * g, a);
            SB_STORE4(dvp + 16 * dt + 4 * g, b);
        }
    }
  }   // key blocks
}

template <typename T, int DH, int U, bool FP8 = false, int RES = 0>
__global__ __launch_bounds__(512 / U) void sattn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                            const float* __restrict__ lse,
                                                            const float* __restrict__ delta, T* __restrict__ dqkv,
                                                            int P, int heads, float scale, long ldqkv, long ldo) {
    sattn_dkv_body<T, DH, U, FP8, RES, false>(qkv, dout, lse, delta, dqkv, P, heads, scale, ldqkv, ldo);
}

// Both parts in ONE workgroup per (frame, head) (keys-resident form): dQ with K / V staged in LDS, a barrier, then dK / dV
// with Q / dO staged over the same bytes.  The second part's operands were read by this CU microseconds earlier -- they
// stay on the CU -- the wavefronts' own key rows are read from the K / V images before they are overwritten, the Q / dO
// images are written from the row fragments the dQ part loaded -- and delta passes through LDS: every operand is read from
// global memory once (HBM-side traffic 1.69 x -> 1.0 x of the algorithmic bytes), one launch instead of two.
template <typename T, int DH, bool FP8 = false, int NCH = 2>
__global__ __launch_bounds__(512) void sattn_bwd_fused_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                              const T* __restrict__ dout, const float* __restrict__ lse,
                                                              T* __restrict__ dqkv, int P, int heads, float scale,
                                                              long ldqkv, long ldo) {
    SattnKeep<T, DH, 1, NCH> keep;
    sattn_dq_body<T, DH, 1, FP8, NCH, true>(qkv, out, dout, lse, nullptr, dqkv, P, heads, scale, ldqkv, ldo, &keep);
    __syncthreads();                    // every wavefront is done with the K / V images
    sattn_dkv_body<T, DH, 1, FP8, NCH, true>(qkv, dout, lse, nullptr, dqkv, P, heads, scale, ldqkv, ldo, &keep);
}

#include "attn_spatial_pers.h"

// ------------------------------------------------------------------------------------------
#define DISPATCH_DH(DHV, ...)                                   \
    do {                                                        \
        if ((DHV) == 64) { constexpr int DH = 64; __VA_ARGS__; } \
        else if ((DHV) == 32) { constexpr int DH = 32; __VA_ARGS__; } \
        else return ISTVT_ERR_SHAPE;                            \
    } while (0)

// RES launches: dynamic LDS = two images of res_img_rows(P) rows (+ two float statistics rows for the dK / dV kernel);
// above 64 KiB the kernel needs its limit raised once (hipFuncSetAttribute).
template <typename K>
static int res_lds(K kernel, int P, int dh, bool with_stats, size_t* bytes) {
    const size_t b = 2ul * res_img_rows(P) * (dh + 16) * sizeof(bf16_t) + (with_stats ? 2ul * res_img_rows(P) * sizeof(float) : 0);
    *bytes = b;
    if (b > 65536 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b) != hipSuccess)
        return ISTVT_ERR_LAUNCH;
    return ISTVT_OK;
}
// RC = resident chunks for this P (2 or 3) as a constant expression
#define DISPATCH_RC(PV, ...)                                     \
    do {                                                         \
        if ((PV) <= 2 * CHUNK) { constexpr int RC = 2; __VA_ARGS__; } \
        else { constexpr int RC = 3; __VA_ARGS__; }              \
    } while (0)
#define LAUNCH_RES(KERNEL, STATS, ...)                                                            \
    do {                                                                                          \
        size_t lds_ = 0;                                                                          \
        const int rc_ = res_lds(KERNEL, P, dh, STATS, &lds_);                                     \
        if (rc_) return rc_;                                                                      \
        hipLaunchKernelGGL(KERNEL, dim3(1, BF * heads), dim3(512), lds_, stream, __VA_ARGS__);    \
    } while (0)

// qkv: [BF*P][3*heads*dh]; out: [BF*P][heads*dh]; lse: [BF*P][heads][2] = (row max in the log2 domain, 1/rowsum)
extern "C" int istvt_attn_spatial_fwd(const void* qkv, long ldqkv, void* out, long ldo, float* lse, int BF, int P, int heads, int dh,
                                      float scale, int dtype, hipStream_t stream) {
    if (BF <= 0 || P <= 0 || heads <= 0) return ISTVT_ERR_SHAPE;
    if (ldqkv < 3L * heads * dh || ldo < (long)heads * dh || ldqkv % 8 || ldo % 8) return ISTVT_ERR_SHAPE;
    dim3 grid((P + 127) / 128, BF * heads);
    // bf16: 8 wavefronts x 16 queries (more wavefronts per SIMD); fp32 keeps 4 x 32 (its LDS image fills the CU)
    if (dtype == DT_BF16) {
        static const int pers = istvt_tune("ISTVT_SATTN_PERS", 1);
        // (P <= 240: at least one of the 16 wavefronts owns no query tile and does the staging; with all 16 busy -- P = 256:
        //  114 us against 93 -- the keys-resident kernel below is faster)
        if (pers && dh == 64 && P > CHUNK && P <= 240 && ldqkv * 2 * P < 0x7fffffffL) {
            // persistent form: one 16-wavefront workgroup per CU walks the (frame, head) problems, the next problem's K / V
            // arriving by LDS-DMA under the current one's arithmetic (attn_spatial_pers.h)
            const int cus = istvt_device_cus();
            const int nprob = BF * heads, ntt = (P - CHUNK + 15) / 16;
            const dim3 pgrid(nprob < cus ? nprob : cus);
#define SP_LAUNCH(NTTV)                                                                                                   \
            case NTTV: {                                                                                                  \
                static std::atomic<unsigned long long> lds_raised{0};                                                      \
                if (istvt_raise_lds_limit(lds_raised, reinterpret_cast<const void*>(sattn_fwd_pers_kernel<64, NTTV>),      \
                                          4 * spers::IMG_B) != ISTVT_OK)                                                   \
                    return ISTVT_ERR_LAUNCH;                                                                               \
                hipLaunchKernelGGL((sattn_fwd_pers_kernel<64, NTTV>), pgrid, dim3(1024), 4 * spers::IMG_B, stream,          \
                                   (const bf16_t*)qkv, (bf16_t*)out, lse, nprob, P, heads, scale, ldqkv, ldo);            \
            } break;
            switch (ntt) {
                SP_LAUNCH(1) SP_LAUNCH(2) SP_LAUNCH(3) SP_LAUNCH(4) SP_LAUNCH(5) SP_LAUNCH(6) SP_LAUNCH(7) SP_LAUNCH(8)
                default: return ISTVT_ERR_SHAPE;
            }
#undef SP_LAUNCH
            return istvt_check_launch();
        }
        if (P > CHUNK && P <= RES_CHUNKS_MAX * CHUNK) {      // one workgroup per (frame, head), keys resident (P = 197 at 224^2)
            DISPATCH_DH(dh, DISPATCH_RC(P, LAUNCH_RES((sattn_fwd_kernel<bf16_t, DH, 1, false, RC>), false, (const bf16_t*)qkv, (bf16_t*)out, lse,
                                       P, heads, scale, ldqkv, ldo)));
            return istvt_check_launch();
        }
        DISPATCH_DH(dh, hipLaunchKernelGGL((sattn_fwd_kernel<bf16_t, DH, 1>), grid, dim3(512), 0, stream,
                                           (const bf16_t*)qkv, (bf16_t*)out, lse, P, heads, scale, ldqkv, ldo));
        return istvt_check_launch();
    }
    dim3 block(256);
    DISPATCH_DTYPE(dtype, DISPATCH_DH(dh, hipLaunchKernelGGL((sattn_fwd_kernel<T, DH, 2>), grid, block, 0, stream,
                                                             (const T*)qkv, (T*)out, lse, P, heads, scale, ldqkv, ldo)));
    return istvt_check_launch();
}

// delta: scratch [BF*P][heads] fp32 (written here, consumed by the dK/dV kernel); dqkv: [BF*P][3*inner]
extern "C" int istvt_attn_spatial_bwd(const void* qkv, long ldqkv, const void* out, const void* dout, long ldo, const float* lse,
                                      float* delta, void* dqkv, int BF, int P, int heads, int dh, float scale,
                                      int dtype, hipStream_t stream) {
    if (BF <= 0 || P <= 0 || heads <= 0) return ISTVT_ERR_SHAPE;
    if (ldqkv < 3L * heads * dh || ldo < (long)heads * dh || ldqkv % 8 || ldo % 8) return ISTVT_ERR_SHAPE;
    dim3 grid((P + 127) / 128, BF * heads);
    static const int fused = istvt_tune("ISTVT_SATTN_FUSED_BWD", 1);
    if (fused && dtype == DT_BF16 && P > CHUNK && P <= RES_CHUNKS_MAX * CHUNK) {
        DISPATCH_DH(dh, DISPATCH_RC(P, LAUNCH_RES((sattn_bwd_fused_kernel<bf16_t, DH, false, RC>), true, (const bf16_t*)qkv, (const bf16_t*)out,
                                   (const bf16_t*)dout, lse, (bf16_t*)dqkv, P, heads, scale, ldqkv, ldo)));
        return istvt_check_launch();
    }
    if (dtype == DT_BF16 && P > CHUNK && P <= RES_CHUNKS_MAX * CHUNK) {
        DISPATCH_DH(dh, DISPATCH_RC(P, {
            LAUNCH_RES((sattn_bwd_dq_kernel<bf16_t, DH, 1, false, RC>), false, (const bf16_t*)qkv, (const bf16_t*)out,
                       (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, P, heads, scale, ldqkv, ldo);
            LAUNCH_RES((sattn_bwd_dkv_kernel<bf16_t, DH, 1, false, RC>), true, (const bf16_t*)qkv, (const bf16_t*)dout, lse,
                       (const float*)delta, (bf16_t*)dqkv, P, heads, scale, ldqkv, ldo);
        }));
        return istvt_check_launch();
    }
    if (dtype == DT_BF16) {                     // 8 wavefronts x 16 rows, see sattn_fwd_kernel
        DISPATCH_DH(dh, {
            hipLaunchKernelGGL((sattn_bwd_dq_kernel<bf16_t, DH, 1>), grid, dim3(512), 0, stream, (const bf16_t*)qkv,
                               (const bf16_t*)out, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, P, heads, scale, ldqkv, ldo);
            hipLaunchKernelGGL((sattn_bwd_dkv_kernel<bf16_t, DH, 1>), grid, dim3(512), 0, stream, (const bf16_t*)qkv,
                               (const bf16_t*)dout, lse, (const float*)delta, (bf16_t*)dqkv, P, heads, scale, ldqkv, ldo);
        });
        return istvt_check_launch();
    }
    dim3 block(256);
    DISPATCH_DTYPE(dtype, DISPATCH_DH(dh, {
        hipLaunchKernelGGL((sattn_bwd_dq_kernel<T, DH, 2>), grid, block, 0, stream, (const T*)qkv, (const T*)out,
                           (const T*)dout, lse, delta, (T*)dqkv, P, heads, scale, ldqkv, ldo);
        hipLaunchKernelGGL((sattn_bwd_dkv_kernel<T, DH, 2>), grid, block, 0, stream, (const T*)qkv, (const T*)dout, lse,
                           (const float*)delta, (T*)dqkv, P, heads, scale, ldqkv, ldo);
    }));
    return istvt_check_launch();
}

// ---- fp8 variant (bfloat16 storage only): Q, K, V and the probabilities enter the attention MFMAs as OCP e4m3
// (v_mfma_f32_16x16x32_fp8_fp8); softmax, statistics, accumulation and every other product stay as above.  The
// backward recomputes S with the same fp8 operands, so the saved (max, 1/sum) match its probabilities.
extern "C" int istvt_attn_spatial_fwd_fp8(const void* qkv, long ldqkv, void* out, long ldo, float* lse, int BF, int P, int heads, int dh,
                                          float scale, int dtype, hipStream_t stream) {
    if (BF <= 0 || P <= 0 || heads <= 0) return ISTVT_ERR_SHAPE;
    if (ldqkv < 3L * heads * dh || ldo < (long)heads * dh || ldqkv % 8 || ldo % 8) return ISTVT_ERR_SHAPE;
    if (dtype != DT_BF16) return ISTVT_ERR_DTYPE;
    dim3 grid((P + 127) / 128, BF * heads);
    if (P > CHUNK && P <= RES_CHUNKS_MAX * CHUNK) {
        DISPATCH_DH(dh, DISPATCH_RC(P, LAUNCH_RES((sattn_fwd_kernel<bf16_t, DH, 1, true, RC>), false, (const bf16_t*)qkv, (bf16_t*)out, lse, P,
                                   heads, scale, ldqkv, ldo)));
        return istvt_check_launch();
    }
    DISPATCH_DH(dh, hipLaunchKernelGGL((sattn_fwd_kernel<bf16_t, DH, 1, true>), grid, dim3(512), 0, stream,
                                       (const bf16_t*)qkv, (bf16_t*)out, lse, P, heads, scale, ldqkv, ldo));
    return istvt_check_launch();
}

extern "C" int istvt_attn_spatial_bwd_fp8(const void* qkv, long ldqkv, const void* out, const void* dout, long ldo, const float* lse,
                                          float* delta, void* dqkv, int BF, int P, int heads, int dh, float scale,
                                          int dtype, hipStream_t stream) {
    if (BF <= 0 || P <= 0 || heads <= 0) return ISTVT_ERR_SHAPE;
    if (ldqkv < 3L * heads * dh || ldo < (long)heads * dh || ldqkv % 8 || ldo % 8) return ISTVT_ERR_SHAPE;
    if (dtype != DT_BF16) return ISTVT_ERR_DTYPE;
    dim3 grid((P + 127) / 128, BF * heads);
    static const int fused = istvt_tune("ISTVT_SATTN_FUSED_BWD", 1);
    if (fused && P > CHUNK && P <= RES_CHUNKS_MAX * CHUNK) {
        DISPATCH_DH(dh, DISPATCH_RC(P, LAUNCH_RES((sattn_bwd_fused_kernel<bf16_t, DH, true, RC>), true, (const bf16_t*)qkv, (const bf16_t*)out,
                                   (const bf16_t*)dout, lse, (bf16_t*)dqkv, P, heads, scale, ldqkv, ldo)));
        return istvt_check_launch();
    }
    if (P > CHUNK && P <= RES_CHUNKS_MAX * CHUNK) {
        DISPATCH_DH(dh, DISPATCH_RC(P, {
            LAUNCH_RES((sattn_bwd_dq_kernel<bf16_t, DH, 1, true, RC>), false, (const bf16_t*)qkv, (const bf16_t*)out,
                       (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, P, heads, scale, ldqkv, ldo);
            LAUNCH_RES((sattn_bwd_dkv_kernel<bf16_t, DH, 1, true, RC>), true, (const bf16_t*)qkv, (const bf16_t*)dout, lse,
                       (const float*)delta, (bf16_t*)dqkv, P, heads, scale, ldqkv, ldo);
        }));
        return istvt_check_launch();
    }
    DISPATCH_DH(dh, {
        hipLaunchKernelGGL((sattn_bwd_dq_kernel<bf16_t, DH, 1, true>), grid, dim3(512), 0, stream, (const bf16_t*)qkv,
                           (const bf16_t*)out, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, P, heads, scale, ldqkv, ldo);
        hipLaunchKernelGGL((sattn_bwd_dkv_kernel<bf16_t, DH, 1, true>), grid, dim3(512), 0, stream, (const bf16_t*)qkv,
                           (const bf16_t*)dout, lse, (const float*)delta, (bf16_t*)dqkv, P, heads, scale, ldqkv, ldo);
    });
    return istvt_check_launch();
}
