// MFMA GEMM with fused epilogues for gfx950.
//
//   C[m][n] = epi( sum_k A(m,k) * B(n,k) )          m < M, n < N, k < K
//
// Replaces every nn.Linear of the transformer (reference network/vivit/module.py:27,30,74,77,
// 182,183,186; network/vivit/vivit.py:129) and every 1x1 / im2col'd convolution of the stem
// (network/xception.py:44,57,118,122), forward, input-gradient and weight-gradient:
//   forward  y = x W^T      : A = x  [M][K] k-contiguous,  B = W  [N][K] k-contiguous
//   dgrad    dx = dy W      : A = dy [M][K] k-contiguous,  B = W  [K][N] n-contiguous (a_kc=1,b_kc=0)
//   wgrad    dW = dy^T x    : A = dy [K][M] m-contiguous,  B = x  [K][N] n-contiguous (a_kc=0,b_kc=0)
//
// Tile: 128x128 per 256-thread workgroup (4 wavefronts as 2x2, 64x64 each = 4x4 MFMA 16x16 tiles),
// BK = 64 (bf16) / 32 (f32), register-prefetched global->LDS staging, padded LDS rows.
// MFMA operands are swapped (weights as A-operand) so a lane owns 4 consecutive output columns
// -> 8/16-byte epilogue accesses.  bf16: v_mfma_f32_16x16x32_bf16, transposed operands read with
// ds_read_b64_tr_b16; f32: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
#include "common.h"
#include <cstdlib>
#include <algorithm>

#define EPI_NONE 0
#define EPI_GELU_FWD 1   // C = u (pre-activation), C2 = gelu(u)
#define EPI_GELU_BWD 2   // C = acc * gelu'(U), U given in C2

struct GemmArgs {
    const void* A; const void* B; void* C; void* C2;
    const float* bias; const void* residual;
    long lda, ldb, ldc, ldr;
    int M, N, K;
    int epi;
    int atomic_f32;       // accumulate into float C with atomics (split-K weight gradients)
    int out_f32;          // C is float regardless of T (plain store)
    int kper;             // K range per blockIdx.z
    int a_vec, b_vec;     // 16-byte vector loads legal for the operand
    float alpha;
    long slab;            // out_mode 3: blockIdx.z writes its fp32 partial at C + z * slab (elements)
    int gm;               // gemm256q: row-panels per tile group (L2 locality of the tile walk)
    int band;             // gemm256q: > 0: column bands of this many tiles instead (wide outputs, see tile_origin)
    int blocked;          // float32 generic kernel: sum every 32-deep step from zero, then add it to the running total
    int walk;             // gemm256q: 0 = XCD-contiguous eighths of the (group, row, column) list, 1 = row-panel slab per XCD
    double* st_sum;       // gemm256q<.., STATS = 1>: per-column sum / sum of squares of the STORED outputs, replica 0's rows
    double* st_sumsq;     //   (double[R][2][N] accumulators of stem.hip; train-mode BatchNorm statistics of a 1x1 conv)
                          // gemm256q<.., STATS = 2>: st_sum only (a bias gradient; folded by istvt_stats_reduce_add)
    int a_sel_col;        // gemm256q: > 0: column tiles at or past this column take their A rows from a SECOND plane,
    int a2_off;           //   a2_off bytes behind A (istvt_gemm flags bit 1: the plane follows the first one, M rows of lda)
    void* dbg;            // diagnostic builds (-DISTVT_GEMM_DIAG): where the in-kernel stamps go when C2 is a real operand
    int gelu_d;           // istvt_gemm flags bit 4: epi 1 stores gelu'(u) in C instead of u; epi 2 multiplies by the saved derivative in C2
};

__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float u) {
    const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * u * u);
    return cdf + u * pdf;
}

// Branch-free erf for the bf16 epilogues (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 absolute: 4 orders below
// one bf16 ulp of gelu / gelu').  libm's erff is two divergent polynomial branches (~50 VALU operations when a
// wavefront takes both); in the GELU epilogues that was ~0.17 ms of VALU per launch with the MFMA pipe idle.
// exp(-x^2) is shared with the Gaussian of gelu'.  The fp32 parity kernels keep erff.
__device__ __forceinline__ float erfc_pos_fast(float ax, float e /* exp(-ax*ax) */) {
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    return p * t * e;                                   // erfc(ax), ax >= 0
}
__device__ __forceinline__ float gelu_fast(float u) {
    const float x = u * 0.70710678118654752440f, ax = fabsf(x);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float erf_abs = 1.0f - erfc_pos_fast(ax, e);
    return 0.5f * u * (1.0f + copysignf(erf_abs, x));
}
__device__ __forceinline__ float gelu_grad_fast(float u) {
    const float x = u * 0.70710678118654752440f, ax = fabsf(x);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);       // exp(-u*u/2)
    const float erf_abs = 1.0f - erfc_pos_fast(ax, e);
    return fmaf(u * 0.39894228040143267794f, e, 0.5f * (1.0f + copysignf(erf_abs, x)));
}

// Two elements at a time: the multiplies / FMAs become v_pk_mul_f32 / v_pk_fma_f32 (two lanes-worth per issue), only
// the two transcendentals and the sign transfer stay scalar.  The GELU epilogues are pure VALU time with the MFMA
// pipe idle (~15 k cycles per 256x256 tile against 32 k for its K loop), so instruction count is what they cost.
typedef float gf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gf2 erf_abs2(gf2 ax, gf2& e) {
    const gf2 arg = ax * ax * gf2{-1.4426950408889634f, -1.4426950408889634f};
    e = gf2{__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};                  // exp(-ax^2)
    const gf2 d = ax * gf2{0.3275911f, 0.3275911f} + gf2{1.0f, 1.0f};
    const gf2 t = gf2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    gf2 p = t * gf2{1.061405429f, 1.061405429f} + gf2{-1.453152027f, -1.453152027f};
    p = p * t + gf2{1.421413741f, 1.421413741f};
    p = p * t + gf2{-0.284496736f, -0.284496736f};
    p = p * t + gf2{0.254829592f, 0.254829592f};
    return gf2{1.0f, 1.0f} - p * t * e;                                                      // erf(ax), ax >= 0
}
__device__ __forceinline__ gf2 gelu_fast2(gf2 u) {
    const gf2 x = u * gf2{0.70710678118654752440f, 0.70710678118654752440f};
    gf2 e;
    const gf2 ea = erf_abs2(gf2{fabsf(x.x), fabsf(x.y)}, e);
    const gf2 er = gf2{copysignf(ea.x, x.x), copysignf(ea.y, x.y)};
    const gf2 hu = u * gf2{0.5f, 0.5f};
    return hu * er + hu;
}
// gelu(u) and gelu'(u) from ONE erf / exp pair (the Gaussian of gelu' is the exponential of the erf approximation):
// what the forward epilogue stores when the backward is to multiply by a saved derivative (istvt_gemm flags bit 4)
__device__ __forceinline__ gf2 gelu_both_fast2(gf2 u, gf2& grad) {
    const gf2 x = u * gf2{0.70710678118654752440f, 0.70710678118654752440f};
    gf2 e;                                                                                   // exp(-u*u/2)
    const gf2 ea = erf_abs2(gf2{fabsf(x.x), fabsf(x.y)}, e);
    const gf2 er = gf2{copysignf(ea.x, x.x), copysignf(ea.y, x.y)};
    grad = (u * gf2{0.39894228040143267794f, 0.39894228040143267794f}) * e + (er * gf2{0.5f, 0.5f} + gf2{0.5f, 0.5f});
    const gf2 hu = u * gf2{0.5f, 0.5f};
    return hu * er + hu;
}
__device__ __forceinline__ gf2 gelu_grad_fast2(gf2 u) {
    const gf2 x = u * gf2{0.70710678118654752440f, 0.70710678118654752440f};
    gf2 e;                                                                                   // exp(-u*u/2)
    const gf2 ea = erf_abs2(gf2{fabsf(x.x), fabsf(x.y)}, e);
    const gf2 er = gf2{copysignf(ea.x, x.x), copysignf(ea.y, x.y)};
    return (u * gf2{0.39894228040143267794f, 0.39894228040143267794f}) * e + (er * gf2{0.5f, 0.5f} + gf2{0.5f, 0.5f});
}

#include "gemm_shared.h"
#include "gemm256q.h"
#include "gemm256t.h"

// Smallest output edge sent to the 256x256 DMA kernels.  Narrow outputs (the stem's 64/128-channel
// pointwise convs over ~3 M pixels, K <= 288) waste MFMA lanes in a 256-wide tile, but those GEMMs
// are HBM-bound: what matters is streaming A once with full-line DMA and storing C in 16-byte
// row segments, which the 128x128 register-staged kernel does not do.
constexpr int ISTVT_G256_MIN = 64;

template <typename T> struct Tile { static constexpr int BK = 32; };
template <> struct Tile<bf16_t> { static constexpr int BK = 64; };

constexpr int BM = 128, BN = 128;
constexpr int KPAD = 8;      // row padding (elements) of a k-contiguous LDS image
constexpr int RPAD = 8;      // row padding of a rows-contiguous LDS image

// one staged operand tile held in registers between the global load and the LDS store
template <typename T, int BK> struct Stage {
    static constexpr int NV = (128 * BK / 8) / 256;     // 8-element vectors per thread
    typename Mma<T>::frag v[NV];
};

__device__ __forceinline__ void frag_store(bf16_t* p, const bf16x8& f) { *reinterpret_cast<bf16x8*>(p) = f; }
__device__ __forceinline__ void frag_store(float* p, const f32frag& f) {
    *reinterpret_cast<float4*>(p) = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(f.v[4], f.v[5], f.v[6], f.v[7]);
}

// load a 128(rows) x BK(k) tile.  KC: src[row][k]; else src[k][row].
template <typename T, int BK, bool KC>
__device__ __forceinline__ void stage_load(Stage<T, BK>& st, const T* __restrict__ src, long ld, int row0, int nrows,
                                           int k0, int kend, bool vec_ok, int tid) {
    constexpr int NV = Stage<T, BK>::NV;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + 256 * i;
        int row, k;
        if (KC) { row = v / (BK / 8); k = (v % (BK / 8)) * 8; }
        else    { k = v / 16; row = (v % 16) * 8; }
        const int grow = row0 + row, gk = k0 + k;
        if (KC) {
            if (grow < nrows && gk + 8 <= kend && vec_ok) {
                st.v[i] = frag_load(src + (long)grow * ld + gk);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    Mma<T>::set(st.v[i], e, (grow < nrows && gk + e < kend) ? to_f32(src[(long)grow * ld + gk + e]) : 0.f);
            }
        } else {
            if (gk < kend && grow + 8 <= nrows && vec_ok) {
                st.v[i] = frag_load(src + (long)gk * ld + grow);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    Mma<T>::set(st.v[i], e, (gk < kend && grow + e < nrows) ? to_f32(src[(long)gk * ld + grow + e]) : 0.f);
            }
        }
    }
}

template <typename T, int BK, bool KC>
__device__ __forceinline__ void stage_store(const Stage<T, BK>& st, T* img, int tid) {
    constexpr int NV = Stage<T, BK>::NV;
    constexpr int LDK = BK + KPAD, LDR = 128 + RPAD;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + 256 * i;
        if (KC) {
            const int row = v / (BK / 8), k = (v % (BK / 8)) * 8;
            frag_store(img + row * LDK + k, st.v[i]);
        } else {
            const int k = v / 16, row = (v % 16) * 8;
            frag_store(img + k * LDR + row, st.v[i]);
        }
    }
}

template <typename T, int BK, bool KC>
__device__ __forceinline__ typename Mma<T>::frag frag_get(const T* img, int row16, int ks, int g, int r) {
    constexpr int LDK = BK + KPAD, LDR = 128 + RPAD;
    if (KC) return frag_load(img + (row16 + r) * LDK + ks * 32 + 8 * g);
    return frag_load_tr(img, LDR, ks * 32 + 8 * g, ks * 32 + 8 * g + 4, row16, r);
}

template <typename T, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    constexpr int BK = Tile<T>::BK;
    constexpr int KS = BK / 32;
    constexpr int A_ELEMS = A_KC ? 128 * (BK + KPAD) : BK * (128 + RPAD);
    constexpr int B_ELEMS = B_KC ? 128 * (BK + KPAD) : BK * (128 + RPAD);
    __shared__ __attribute__((aligned(16))) T smem[A_ELEMS + B_ELEMS];
    T* As = smem;
    T* Bs = smem + A_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: blocks that share an XCD (id % 8) get consecutive tiles, tiles run
    // n-fastest so one XCD's L2 keeps an A row-panel while sweeping the (few) column tiles.
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_n * tiles_m;
    int id = blockIdx.x;
    {
        const int xcd = id & 7, q = nwg >> 3, rem = nwg & 7;
        id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (id >> 3);
    }
    const int bm0 = (id / tiles_n) * BM, bn0 = (id % tiles_n) * BN;
    const int k_begin = blockIdx.z * p.kper;
    const int k_end = min(p.K, k_begin + p.kper);

    const T* A = (const T*)p.A;
    const T* B = (const T*)p.B;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Stage<T, BK> sa, sb;
    const bool avec = p.a_vec != 0, bvec = p.b_vec != 0;
    stage_load<T, BK, A_KC>(sa, A, p.lda, bm0, p.M, k_begin, k_end, avec, tid);
    stage_load<T, BK, B_KC>(sb, B, p.ldb, bn0, p.N, k_begin, k_end, bvec, tid);
    stage_store<T, BK, A_KC>(sa, As, tid);
    stage_store<T, BK, B_KC>(sb, Bs, tid);
    __syncthreads();

    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = k0 + BK < k_end;
        if (more) {
            stage_load<T, BK, A_KC>(sa, A, p.lda, bm0, p.M, k0 + BK, k_end, avec, tid);
            stage_load<T, BK, B_KC>(sb, B, p.ldb, bn0, p.N, k0 + BK, k_end, bvec, tid);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            typename Mma<T>::frag af[4], bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = frag_get<T, BK, A_KC>(As, wm * 64 + t * 16, ks, g, r);
                bf[t] = frag_get<T, BK, B_KC>(Bs, wn * 64 + t * 16, ks, g, r);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if (sizeof(T) == 4 && p.blocked) {
                        // float32 parity mode, flags & 1: every 32-deep step is summed from zero and then added to the running
                        // total (blocked summation).  One sequential fp32 FMA chain over K = 728 ... 2912 -- what eight
                        // v_mfma_f32_16x16x4_f32 per step on ONE accumulator amount to -- left the gradients of the golden
                        // model G5 6-8x further from the float64 reference run than torch's CPU float32 (whose GEMM keeps
                        // 8-16 partial sums); the error now grows with 32 + K / 32 terms instead of K.
                        f32x4 part = f32x4{0.f, 0.f, 0.f, 0.f};
                        Mma<T>::mma(part, bf[nt], af[mt]);
                        acc[mt][nt] += part;
                    } else {
                        Mma<T>::mma(acc[mt][nt], bf[nt], af[mt]);
                    }
                }
        }
        __syncthreads();
        if (more) {
            stage_store<T, BK, A_KC>(sa, As, tid);
            stage_store<T, BK, B_KC>(sb, Bs, tid);
            __syncthreads();
        }
    }

    // ---- epilogue: lane owns C[m][n..n+3], m = tile row r, n = 4g..4g+3 of each 16x16 tile
    const bool nvec = (p.N % 4 == 0) && (p.ldc % 4 == 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = bm0 + wm * 64 + mt * 16 + r;
        if (m >= p.M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = bn0 + wn * 64 + nt * 16 + 4 * g;
            if (n >= p.N) continue;
            gemm_epilogue4<T>(p, m, n, acc[mt][nt], nvec);
        }
    }
}

template <typename T>
static int launch_gemm(const GemmArgs& a, int a_kc, int b_kc, int splitk, hipStream_t stream) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    dim3 grid(tiles, 1, splitk), block(256);
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_kernel<T, true, true>), grid, block, 0, stream, a);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_kernel<T, true, false>), grid, block, 0, stream, a);
    else if (!a_kc && !b_kc) hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, block, 0, stream, a);
    else return ISTVT_ERR_SHAPE;
    return istvt_check_launch();
}

// CUs the persistent NT kernel leaves alone: bits 8..15 of istvt_gemm's `flags`, in units of 8 CUs -- an argument of the
// launch, not state of the library (the entry points hold no global mutable state; the autograd thread and
// the main thread launch concurrently).  While a collective's kernels hold CUs, a persistent workgroup dealt to one of them
// would start only when another workgroup has finished its whole tile list and the launch would take up to twice as
// long; with the grid cut to the CUs that are really free every workgroup is resident at once (666 tiles on 224
// workgroups are the same three rounds as on 256).
extern "C" int istvt_gemm(const void* A, long lda, int a_kc, const void* B, long ldb, int b_kc, void* C, long ldc,
                          int M, int N, int K, const float* bias, const void* residual, long ldr, void* C2, int epi,
                          int out_mode, int splitk, float alpha, double* col_sum, double* col_sumsq, int flags, int dtype,
                          hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return ISTVT_ERR_SHAPE;
    if (out_mode < 0 || out_mode > 3 || epi < 0 || epi > 2) return ISTVT_ERR_SHAPE;
    if (splitk < 1) splitk = 1;
    if (splitk > 1 && out_mode < 2) return ISTVT_ERR_SHAPE;        // split-K needs atomics (2) or partial slabs (3)
    const int esz = dtype == DT_F32 ? 4 : 2;
    const int bk = dtype == DT_F32 ? 32 : 64;
    GemmArgs a{};
    a.A = A; a.B = B; a.C = C; a.C2 = C2; a.bias = bias; a.residual = residual;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr; a.M = M; a.N = N; a.K = K; a.epi = epi;
    a.out_f32 = (out_mode == 1 || out_mode == 3); a.atomic_f32 = out_mode == 2; a.alpha = alpha;
    a.slab = out_mode == 3 ? (long)M * ldc : 0;
    a.st_sum = col_sum; a.st_sumsq = col_sumsq;
    a.blocked = flags & 1;
    a.gelu_d = (flags >> 4) & 1;
    if (a.gelu_d && epi == EPI_NONE) return ISTVT_ERR_SHAPE;
    const bool a_sel = (flags & 2) != 0;
    const int cu_reserve = ((flags >> 8) & 0xff) * 8;
    if (cu_reserve > 192) return ISTVT_ERR_SHAPE;
    if (a_sel) {
        a.a_sel_col = ((flags >> 16) & 0xffff) * 64;
        a.a2_off = (int)((long)M * lda * 2);
        if (a.a_sel_col <= 0 || a.a_sel_col % T256 != 0 || a.a_sel_col >= N || dtype != DT_BF16 || !a_kc ||
            2 * (long)M * lda * 2 >= 0x7fffffffL) return ISTVT_ERR_SHAPE;
    }
    if (col_sumsq && !col_sum) return ISTVT_ERR_SHAPE;
    a.gm = 4;             // sweep at the model's shapes: 4 row-panels per tile group is best or neutral everywhere
    {
        // column bands for wide outputs (tile_origin in gemm256q.h).  Measured in the model at N = 2912 (12 column tiles):
        // bands of 6 tiles: GELU-forward GEMM 4.12 -> 3.96 ms per step, GELU-backward 4.05 -> 3.98; bands of 3: -3 % / +1.5 %;
        // bands of 2 or 4: the plain epilogue gains 7..9 %, the two GELU epilogues lose 8..16 %.  ISTVT_GEMM_BAND overrides.
        static const int band_env = istvt_tune("ISTVT_GEMM_BAND", 6);
        const int tn = (N + T256 - 1) / T256;
        a.band = tn >= 8 ? band_env : 0;
    }
    int kper = (K + splitk - 1) / splitk;
    kper = ((kper + bk - 1) / bk) * bk;
    a.kper = kper;
    splitk = (K + kper - 1) / kper;
    a.a_vec = (((uintptr_t)A % 16) == 0 && (lda * esz) % 16 == 0) ? 1 : 0;
    a.b_vec = (((uintptr_t)B % 16) == 0 && (ldb * esz) % 16 == 0) ? 1 : 0;
    // Large bf16 problems run on the two 256x256 LDS-DMA kernels: NT (forward, input gradient over the cached W^T) on
    // the persistent gemm256q, TN (weight gradient, fp32 split-K slabs) on gemm256t.  Everything they do not take --
    // float32, narrow outputs, misaligned operands, operands >= 2 GiB (32-bit buffer offsets), K < 32, atomic outputs --
    // runs on the generic 128x128 kernel below, which handles every mode.
    const bool out16 = N % 8 == 0 && ldc % 8 == 0 && ((uintptr_t)C % 16) == 0 &&
                       (!residual || (ldr % 8 == 0 && ((uintptr_t)residual % 16) == 0)) &&
                       (!C2 || ((uintptr_t)C2 % 16) == 0) && (!bias || ((uintptr_t)bias % 16) == 0);
    if (dtype == DT_BF16 && a.a_vec && a.b_vec && out16 && M >= ISTVT_G256_MIN && N >= ISTVT_G256_MIN && a_kc == b_kc &&
        (a_kc ? (K % 8 == 0) : (M % 8 == 0 && N % 8 == 0))) {
        const int tiles = ((M + T256 - 1) / T256) * ((N + T256 - 1) / T256);
        const dim3 block(512);
        const bool q_ok = a_kc && (long)M * lda * 2 * (a_sel ? 2 : 1) < 0x7fffffffL && (long)N * ldb * 2 < 0x7fffffffL && K >= 32 &&
                          out_mode == 0 && splitk == 1 && (!bias || alpha == 1.0f) && !(epi != EPI_NONE && residual);
        const bool t_ok = !a_kc && out_mode == 3 && !bias && !residual && epi == EPI_NONE &&
                          (long)K * lda * 2 < 0x7fffffffL && (long)K * ldb * 2 < 0x7fffffffL;
        if (q_ok) {
            // persistent NT kernel: one workgroup per CU walks its tiles with the LDS ring kept full across tiles
            const int cus_dev = istvt_device_cus() & ~7;
            const int cus = std::max(8, (cus_dev - cu_reserve) & ~7);      // the CUs this launch may fill (flags bits 8..15)
            // Balanced rounds: 666 tiles on 256 CUs are three rounds whichever way they are dealt; dealing them to 224
            // workgroups (3 tiles each) takes the same time and leaves 32 CUs free for the whole launch -- for the
            // weight-gradient GEMM running on the side stream -- instead of 102 CUs free for the last round only.
            auto balanced = [&](int ntiles) {
                int G = ntiles < cus ? ntiles : cus;
                if (ntiles > cus) {
                    const int rounds = (ntiles + cus - 1) / cus;
                    const int g8 = (((ntiles + rounds - 1) / rounds) + 7) & ~7;     // multiple of 8: blockIdx & 7 stays the XCD
                    if (g8 < G) G = g8;
                }
                return G;
            };
            // The tile walk (gemm256q.h): row-panel slabs per XCD for the tall GEMMs of the model, the XCD-contiguous list
            // walk for everything else (few row panels; the BatchNorm-statistics epilogue, which wants to stay on one
            // column tile as long as possible).
            const int tiles_m256 = (M + T256 - 1) / T256;
            // Measured (tools/gemm_q_sweep.py + rocprofv3 --pmc FETCH_SIZE, M = 56 736): the slab walk fetches 10-22 % fewer
            // bytes than the list walk on the plain / residual epilogues (N = 1536, K = 728: 264 -> 212 MB per launch;
            // N = 728, K = 2912: 534 -> 461) and 2-5 % MORE on the two GELU epilogues (N = 2912 column bands); launch time
            // and the step are the same within the run-to-run noise either way (diagnostic builds that serve every A
            // panel from L2 are only 0-5 % faster: the K loop is not bound by where the operands come from).
            static const int walk_env = istvt_tune("ISTVT_GEMM_WALK", 1), gm_env = istvt_tune("ISTVT_GEMM_GM", -1),
                             g_env = istvt_tune("ISTVT_GEMM_G", 0);
            const bool slab = walk_env == 1 && tiles_m256 >= 64 && !col_sumsq && epi == EPI_NONE;
            auto pick_grid = [&](int ntiles, int tm_rows) {
                if (!slab) return balanced(ntiles);
                // slab walk: the XCD with the most row panels sets the rounds; as few workgroups per XCD as fill them
                const int tiles_m = (M + tm_rows - 1) / tm_rows, tiles_n = (N + T256 - 1) / T256;
                const int per_xcd = ((tiles_m + 7) / 8) * tiles_n, cux = cus / 8;
                const int rounds = (per_xcd + cux - 1) / cux;
                int G = 8 * ((per_xcd + rounds - 1) / rounds);
                a.walk = 1;
                a.gm = gm_env > 0 ? gm_env : 1;
                if (g_env > 0) G = g_env & ~7;
                return G;
            };
#ifdef ISTVT_GEMM_DIAG
            static const int qdbg = istvt_tune("ISTVT_GEMM_QDBG", 0);
            if (qdbg && (epi == EPI_GELU_FWD || epi == EPI_GELU_BWD)) {
                // the GELU epilogues' stamps (round 5): 1024 = per-tile spans only, 1152 = + output stores out of range,
                // 3072 = + no GELU arithmetic (bit 2048), 3200 = neither; stamps go to the buffer named by ISTVT_GEMM_DBGPTR
                const char* dp = getenv("ISTVT_GEMM_DBGPTR");
                a.dbg = dp ? (void*)strtoull(dp, nullptr, 0) : nullptr;
                if (!a.dbg) return ISTVT_ERR_SHAPE;
                const int G = pick_grid(tiles, 256);
                const bool kh = K > 64 && (K & 63) != 0 && (K & 63) <= 32;
                if (!kh) return ISTVT_ERR_SHAPE;
#define QG(n) case n: if (epi == EPI_GELU_FWD) hipLaunchKernelGGL((gemm256q_kernel<EPI_GELU_FWD, false, n, 256, 0, true>), dim3(G), block, 0, stream, a); \
                      else if (col_sum) hipLaunchKernelGGL((gemm256q_kernel<EPI_GELU_BWD, false, n, 256, 2, true>), dim3(G), block, 0, stream, a); \
                      else hipLaunchKernelGGL((gemm256q_kernel<EPI_GELU_BWD, false, n, 256, 0, true>), dim3(G), block, 0, stream, a); break;
                switch (qdbg) {
                    QG(1024) QG(1152) QG(3072) QG(3200)
                    default: return ISTVT_ERR_SHAPE;
                }
#undef QG
                return istvt_check_launch();
            }
            if (qdbg && epi == 0 && !residual) {
                const int G = pick_grid(tiles, 256);
#define QD(n) case n: hipLaunchKernelGGL((gemm256q_kernel<0, false, n>), dim3(G), block, 0, stream, a); break;
                switch (qdbg) {
                    QD(1) QD(2) QD(4) QD(6) QD(8) QD(16) QD(22) QD(24) QD(48) QD(54) QD(128) QD(136) QD(129) QD(256) QD(257) QD(258) QD(260) QD(262) QD(304) QD(768) QD(774) QD(1024) QD(1152)
                    default: return ISTVT_ERR_SHAPE;
                }
#undef QD
                return istvt_check_launch();
            }
#endif
            // 224-row tiles (gemm256q.h) fill the rounds better at the model's shapes (cost = rounds x tile height),
            // but a tile's time does not shrink with its row count -- the load slot, not the MFMA count, sets the K
            // loop, and the epilogue is per tile: measured at C2, 254 x 3 tiles of 224 rows against 222 x 3 of 256:
            // N=728/K=2912 +2 %, N=512/K=728 +2.5 %, the GELU epilogue GEMMs -5..-8 %, the step +0.5 ms.  Off by
            // default: ISTVT_GEMM_TM=224 forces it (tests/test_model_gpu.py runs the GEMM checks that way), =-1 picks
            // by rounds x height.
            // (read from the environment in EVERY build, unlike the sweep knobs behind -DISTVT_TUNE: it selects another
            //  kernel instantiation, which tests/test_model_gpu.py::test_gemm_224_row_tile_variant must be able to reach
            //  in the shipped library, and ops.gemm_kernel_name reads the same variable)
            static const int tm_env = [] { const char* v = getenv("ISTVT_GEMM_TM"); return v ? atoi(v) : 0; }();
            const int tiles224 = ((M + 223) / 224) * ((N + T256 - 1) / T256);
            const long cost256 = (long)((tiles + cus - 1) / cus) * 256, cost224 = (long)((tiles224 + cus - 1) / cus) * 224;
            if (!col_sum && (tm_env == 224 || (tm_env == -1 && cost224 < cost256))) {
                const dim3 grid(pick_grid(tiles224, 224));
                if (epi == EPI_GELU_FWD) hipLaunchKernelGGL((gemm256q_kernel<EPI_GELU_FWD, false, 0, 224>), grid, block, 0, stream, a);
                else if (epi == EPI_GELU_BWD) hipLaunchKernelGGL((gemm256q_kernel<EPI_GELU_BWD, false, 0, 224>), grid, block, 0, stream, a);
                else if (residual) hipLaunchKernelGGL((gemm256q_kernel<0, true, 0, 224>), grid, block, 0, stream, a);
                else hipLaunchKernelGGL((gemm256q_kernel<0, false, 0, 224>), grid, block, 0, stream, a);
                return istvt_check_launch();
            }
            const dim3 grid(pick_grid(tiles, 256));
            // K % 64 in 1..32 (728, 2912): the last K tile of every output tile runs half its MFMAs (gemm256q.h, KHALF)
            static const bool khalf_on = istvt_tune("ISTVT_GEMM_KHALF", 1) != 0;
            const bool kh = khalf_on && K > 64 && (K & 63) != 0 && (K & 63) <= 32;
#define QL(...) do { if (kh) hipLaunchKernelGGL((gemm256q_kernel<__VA_ARGS__, true>), grid, block, 0, stream, a); \
                     else hipLaunchKernelGGL((gemm256q_kernel<__VA_ARGS__, false>), grid, block, 0, stream, a); } while (0)
            if (col_sum && col_sumsq) { // fused column statistics: the plain epilogue only (the stem's 1x1 convolutions)
                if (epi != EPI_NONE || residual) return ISTVT_ERR_SHAPE;
                QL(0, false, 0, 256, 1);
                return istvt_check_launch();
            }
            if (col_sum) {              // fused column sums of the output: the GELU-backward epilogue (the hidden layer's bias gradient)
                if (epi != EPI_GELU_BWD) return ISTVT_ERR_SHAPE;
                QL(EPI_GELU_BWD, false, 0, 256, 2);
                return istvt_check_launch();
            }
            if (epi == EPI_GELU_FWD) QL(EPI_GELU_FWD, false, 0, 256, 0);
            else if (epi == EPI_GELU_BWD) QL(EPI_GELU_BWD, false, 0, 256, 0);
            else if (residual) QL(0, true, 0, 256, 0);
            else QL(0, false, 0, 256, 0);
#undef QL
            return istvt_check_launch();
        }
        if (a_sel) return ISTVT_ERR_SHAPE;          // only the persistent NT kernel selects its A plane by column tile
        if (t_ok && !col_sum) {
            // weight gradient: unit / ping-pong structure with transposing fragment reads (gemm256t.h)
            hipLaunchKernelGGL(gemm256t_kernel, dim3(tiles * splitk), block, 0, stream, a);
            return istvt_check_launch();
        }
    }
    if (col_sum || a_sel) return ISTVT_ERR_SHAPE;   // only the persistent NT kernel accumulates statistics / selects A planes: the host checks first
    DISPATCH_DTYPE(dtype, return launch_gemm<T>(a, a_kc, b_kc, splitk, stream));
    return ISTVT_OK;
}


// out[i] += sum_z ws[z][i]   (second pass of split-K weight gradients written as partial slabs:
// plain stores run ~4-5x the contended-atomic rate, and the sum order is fixed -> reproducible)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, long n,
                                                            float* __restrict__ out) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        float4 acc = *reinterpret_cast<const float4*>(out + i);
#pragma unroll 8
        for (int z = 0; z < splits; ++z) {       // eight slabs' loads in flight per thread (1.55 -> 1.13 ms per step; 16: same)
            const float4 v = *reinterpret_cast<const float4*>(ws + (long)z * n + i);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(out + i) = acc;
    }
}

extern "C" int istvt_splitk_reduce(const float* ws, int splits, long n, float* out, hipStream_t stream) {
    if (splits < 1 || n <= 0 || n % 4 != 0) return ISTVT_ERR_SHAPE;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ws, splits, n, out);
    return istvt_check_launch();
}


// ---- grouped weight gradients ---------------------------------------------------------------------------------------
struct ReduceGroupArgs {
    const float* ws[ISTVT_WGRAD_GROUP_MAX];
    float* out[ISTVT_WGRAD_GROUP_MAX];
    long n[ISTVT_WGRAD_GROUP_MAX];
    int start[ISTVT_WGRAD_GROUP_MAX + 1];
    int count, splits;
};

// out_i[e] += sum_z ws_i[z][e] for every problem of a group: problem i owns blocks start[i] .. start[i+1]-1
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(ReduceGroupArgs g) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < ISTVT_WGRAD_GROUP_MAX; ++i)
        if (i < g.count && (int)blockIdx.x >= g.start[i]) pi = i;
    const float* __restrict__ ws = g.ws[pi];
    float* __restrict__ out = g.out[pi];
    const long n = g.n[pi];
    const int blocks = g.start[pi + 1] - g.start[pi], b = blockIdx.x - g.start[pi];
    const long stride = (long)blocks * 256 * 4;
    for (long i = ((long)b * 256 + threadIdx.x) * 4; i < n; i += stride) {
        float4 acc = *reinterpret_cast<const float4*>(out + i);
#pragma unroll 8
        for (int z = 0; z < g.splits; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(ws + (long)z * n + i);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(out + i) = acc;
    }
}

// out_i[N_i][K_i] += dy_i^T x_i for i < count: dy_i [M][N_i], x_i [M][K_i] bf16 with row strides lddy_i / ldx_i, out_i
// contiguous float.  All problems share the reduction length M and the split count; ws holds splits * sum(N_i K_i)
// floats (problem i's slabs start at splits * sum_{j<i} N_j K_j).  splits <= 0: as many as fill 256 workgroups.
// Returns ISTVT_ERR_SHAPE for problems the 256x256 weight-gradient kernel does not take (the caller then launches
// them one by one through istvt_gemm).
extern "C" int istvt_wgrad_group(int count, const void* const* dy, const long* lddy, const void* const* x, const long* ldx,
                                 float* const* out, const int* N, const int* K, int M, int splits, float* ws,
                                 long ws_elems, hipStream_t stream) {
    if (count < 1 || count > ISTVT_WGRAD_GROUP_MAX || M <= 0 || !ws) return ISTVT_ERR_SHAPE;
    GemmGroupArgs g;
    ReduceGroupArgs rg;
    int tiles_total = 0;
    long elems = 0;
    for (int i = 0; i < count; ++i) {
        if (N[i] < ISTVT_G256_MIN || K[i] < ISTVT_G256_MIN || N[i] % 8 || K[i] % 8) return ISTVT_ERR_SHAPE;
        if (((uintptr_t)dy[i] % 16) || ((uintptr_t)x[i] % 16) || (lddy[i] * 2) % 16 || (ldx[i] * 2) % 16 ||
            ((uintptr_t)out[i] % 16)) return ISTVT_ERR_SHAPE;
        if ((long)M * lddy[i] * 2 >= 0x7fffffffL || (long)M * ldx[i] * 2 >= 0x7fffffffL) return ISTVT_ERR_SHAPE;
        tiles_total += ((N[i] + T256 - 1) / T256) * ((K[i] + T256 - 1) / T256);
        elems += (long)N[i] * K[i];
    }
    if (splits <= 0) splits = 256 / tiles_total;
    if (splits < 1) splits = 1;
    if (splits > (M + 63) / 64) splits = (M + 63) / 64;
    int kper = (M + splits - 1) / splits;
    kper = ((kper + 63) / 64) * 64;
    splits = (M + kper - 1) / kper;
    if (ws_elems < (long)splits * elems) return ISTVT_ERR_SHAPE;
    int start = 0, rstart = 0;
    long off = 0;
    for (int i = 0; i < count; ++i) {
        GemmArgs& a = g.p[i];
        a = GemmArgs{};
        // the kernel's (M, N, K) are (output rows, output columns, reduction length)
        a.A = dy[i]; a.B = x[i]; a.C = ws + off; a.lda = lddy[i]; a.ldb = ldx[i]; a.ldc = K[i];
        a.M = N[i]; a.N = K[i]; a.K = M; a.kper = kper; a.alpha = 1.0f; a.out_f32 = 1; a.a_vec = a.b_vec = 1;
        a.slab = (long)N[i] * K[i]; a.gm = 4;
        g.start[i] = start;
        const int tiles = ((N[i] + T256 - 1) / T256) * ((K[i] + T256 - 1) / T256);
        start += tiles * splits;
        rg.ws[i] = ws + off; rg.out[i] = out[i]; rg.n[i] = (long)N[i] * K[i];
        rg.start[i] = rstart;
        long blocks = (rg.n[i] / 4 + 255) / 256;
        if (blocks > 512) blocks = 512;
        rstart += (int)blocks;
        off += (long)splits * N[i] * K[i];
    }
    for (int i = count; i <= ISTVT_WGRAD_GROUP_MAX; ++i) { g.start[i] = start; rg.start[i] = rstart; }
    g.n = count; rg.count = count; rg.splits = splits;
    hipLaunchKernelGGL(gemm256t_group_kernel, dim3(start), dim3(512), 0, stream, g);
    int rc = istvt_check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(rstart), dim3(256), 0, stream, rg);
    return istvt_check_launch();
}

// the split count istvt_wgrad_group(splits <= 0) would use: sizes the workspace
extern "C" int istvt_wgrad_group_splits(int count, const int* N, const int* K, int M) {
    if (count < 1 || count > ISTVT_WGRAD_GROUP_MAX || M <= 0) return ISTVT_ERR_SHAPE;
    int tiles_total = 0;
    for (int i = 0; i < count; ++i) tiles_total += ((N[i] + T256 - 1) / T256) * ((K[i] + T256 - 1) / T256);
    int splits = 256 / tiles_total;
    if (splits < 1) splits = 1;
    if (splits > (M + 63) / 64) splits = (M + 63) / 64;
    int kper = (((M + splits - 1) / splits + 63) / 64) * 64;
    return (M + kper - 1) / kper;
}
