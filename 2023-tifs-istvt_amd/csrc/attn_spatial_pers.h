// Persistent spatial attention forward for bfloat16, 128 < P <= 256 keys per frame (P = 197 at 224^2), included by
// attn_spatial.hip.  Reference: SpatialOnlyAttention.forward, network/vivit/module.py:84-91.
//
// Round 3's keys-resident kernel ran one workgroup per (frame, head): stage K / V through registers into LDS, barrier,
// compute, exit -- two lock-step workgroups per CU, each idling through its own staging latency (the r03 diagnostic build
// that served every operand from cache took 57 us per layer against 68; the instruction stream itself needs ~25).
// Here ONE workgroup of 16 wavefronts per CU walks the (frame, head) problems dealt to it and the operands of problem
// i + 1 arrive by LDS-DMA while problem i computes:
//   * two LDS buffers of [K image | V image], 256 rows x 128 bytes each (128 KiB in all), filled by
//     `buffer_load_dwordx4 ... lds` (64 pieces of 8 rows x 128 B per problem, four per wavefront); rows past P are out of
//     range for the problem's descriptor and arrive as zeros, so no masking on the way in;
//   * 128-byte row pitch, conflict-free by swizzles applied on the DMA SOURCE address (the images are lane-linear):
//     K (row reads, ds_read_b128): 16-byte chunk c of row k sits at c ^ ((k >> 1) & 7), the GEMM units' swizzle;
//     V (transposed reads, ds_read_b64_tr_b16): 32-byte pair d of row k sits at d ^ ((k >> 1) & 3) -- a half-wave reads
//     rows 8n .. 8n + 7 at one pair: the two halves of four 256-byte bank rows x four distinct pairs;
//   * wavefront w owns query tile w (rows 16 w .. 16 w + 15) in ONE pass: 13 of 16 wavefronts work at P = 197 (the
//     two-block walk of 8 wavefronts used 13 of 16 slots as well, but in two dependent halves); its Q rows for problem
//     i + 1 are requested (inline-assembly buffer loads) with that problem's DMA;
//   * per problem: s_waitcnt vmcnt (counted: the younger operations are this problem's 5 stores... see the loop),
//     barrier, compute, stores, barrier.
// The arithmetic -- S^T = K Q^T with keys on accumulator rows, online softmax over two 128-key chunks in the log2
// domain, O^T += V^T P^T with the score accumulators as the B operand, statistics (max, 1 / sum) -- is sattn_fwd_kernel's.
#pragma once

#ifdef ISTVT_SATTN_STAMP
// diagnostic build (-DISTVT_SATTN_STAMP): per-phase s_memtime sums of every wavefront, [workgroup][wavefront][8] u64 =
// {issue next problem's DMA + Q, vmcnt wait, barrier A, compute, stores, barrier B, problems, -}; tools/sattn_stamps.py
__device__ unsigned long long* g_sattn_stamps = nullptr;
extern "C" int istvt_diag_sattn_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sattn_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -4;
}
#define SP_STAMP(i)                                                                                   \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if ((i) > 0) sp_seg[(i) - 1] += (unsigned)(t_ - sp_prev);                                     \
        sp_prev = t_;                                                                                 \
    } while (0)
#else
#define SP_STAMP(i) do { } while (0)
#endif

namespace spers {
constexpr int IMG_ROWS = 256, ROW_B = 128, IMG_B = IMG_ROWS * ROW_B;      // one image: 32 KiB
constexpr unsigned OOB = 0x80000000u, RSRC_FLAGS = 0x00020000u;

__device__ __forceinline__ const char* uni_ptr(const void* q) {
    const unsigned long long u = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
// The Q fragments of the NEXT problem are requested a whole problem ahead and sit in registers meanwhile.  They are
// compiler-visible loads on purpose: an inline-assembly load's destination counts as written at once, and the first
// version of this kernel (asm loads + hand-counted vmcnt) came out with the in-flight registers COPIED at the loop's back
// edge -- every second problem computed on stale Q.  hipcc tracks a builtin load's registers and places the wait itself;
// the only operations it cannot see are the LDS-DMA pieces, which can only make its counted waits cover more.
__device__ __forceinline__ u32x4 q_load16(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
}
}  // namespace spers

// NTT = 16-key tiles of the SECOND chunk that hold keys (P = 197: 5): a template parameter, so that which tiles exist is
// known at compile time -- with P a run-time value every tile of the tail chunk carried its own range tests and masks
// (v_cndmask was the most frequent vector instruction of the kernel, and the kernel is bound by vector issue: stamps).
template <int DH, int NTT>
__global__ __launch_bounds__(1024) void sattn_fwd_pers_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                              float* __restrict__ lse, int nprob, int P, int heads, float scale,
                                                              long ldqkv, long ldo) {
    static_assert(DH == 64, "128-byte rows");
    using namespace spers;
    constexpr int KS = DH / 32, DT = DH / 16, NT = 8;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, r = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int inner = heads * DH;
    const int ld2 = (int)ldqkv * 2, ldo2 = (int)ldo * 2;
    const float c = scale * LOG2E;
    const unsigned lds0 = (unsigned)(__SIZE_TYPE__)(__attribute__((address_space(3))) void*)sattn_dyn;
    const char* smem = sattn_dyn;

    // ---- producer: piece j of this wavefront = piece q = 16 j + wave of the problem's 64 (K image: 0..31, V image: 32..63)
    unsigned dvoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = 16 * j + wave, pi = q & 31, row = 8 * pi + (lane >> 3), pc = lane & 7;
        const int chunk = (q < 32) ? (pc ^ ((row >> 1) & 7)) : (2 * ((pc >> 1) ^ ((row >> 1) & 3)) + (pc & 1));
        dvoff[j] = (unsigned)(row * ld2 + chunk * 16);
    }
    // this wavefront's query rows: tile `wave`, row 16 wave + r, columns 32 ks + 8 g (rows >= P: out of range, zeros)
    const unsigned qvoff = (unsigned)((16 * wave + r) * ld2 + 16 * g);
    const int q_row = 16 * wave + r;
    const bool tile_live = 16 * wave < P;                        // wave-uniform
    // this lane's output / statistics offsets inside the problem's windows (rows >= P end up past num_records)
    const unsigned ovoff = (unsigned)(q_row * ldo2 + 8 * g);
    const unsigned svoff = (g == 0) ? (unsigned)(q_row * heads * 8) : OOB;

    // Who stages: at P = 197 three of the 16 wavefronts own no query tile.  They have nothing else to do, so THEY send all
    // 64 DMA pieces of the next problem -- while the thirteen others compute -- and the live wavefronts only request their
    // two Q fragments (stamps: with every wavefront sending four pieces at the head of a problem the CU's address path
    // serialised them, 650 -> 2600 cycles before the last wavefront could start).  With no idle wavefront (P > 240) every
    // wavefront sends four pieces as before.
    const int nlive = (P + 15) >> 4, ndead = 16 - nlive;
    const bool by_dead = ndead > 0;                              // workgroup-uniform
    auto prob_rsrc = [&](const int prob, int& h) {
        const bool dead = prob >= nprob;
        h = __builtin_amdgcn_readfirstlane(prob % heads);
        const int bf = __builtin_amdgcn_readfirstlane(prob / heads);
        const char* base = (const char*)qkv + (long)bf * P * ld2;
        return __builtin_amdgcn_make_buffer_rsrc((void*)uni_ptr(base), 0, dead ? 0 : P * ld2, RSRC_FLAGS);
    };
    auto issue_q = [&](const int prob, u32x4 (&qn)[KS]) {
        int h;
        const __amdgpu_buffer_rsrc_t rs = prob_rsrc(prob, h);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qn[ks] = q_load16(rs, qvoff, (h * DH + 32 * ks) * 2);
    };
    auto issue_all_pieces = [&](const int prob, const int buf) {        // a wavefront without a query tile: pieces dw, dw + ndead, ...
        int h;
        const __amdgpu_buffer_rsrc_t rs = prob_rsrc(prob, h);
        const int lr = lane >> 3, pc = lane & 7;
        for (int q = wave - nlive; q < 64; q += ndead) {
            const int pi = q & 31, row = 8 * pi + lr;
            const int chunk = (q < 32) ? (pc ^ ((row >> 1) & 7)) : (2 * ((pc >> 1) ^ ((row >> 1) & 3)) + (pc & 1));
            const unsigned dst = lds0 + buf * 2 * IMG_B + (q >> 5) * IMG_B + pi * 1024;
            dma16_lds(rs, dst, (unsigned)(row * ld2 + chunk * 16), (inner * (1 + (q >> 5)) + h * DH) * 2);
        }
    };
    auto issue = [&](const int prob, const int buf, u32x4 (&qn)[KS]) {
        // (prob >= nprob: every lane out of range -- nothing is fetched, the counts of the vmcnt waits stay the same)
        const bool dead = prob >= nprob;
        const int h = __builtin_amdgcn_readfirstlane(prob % heads), bf = __builtin_amdgcn_readfirstlane(prob / heads);
        const char* base = (const char*)qkv + (long)bf * P * ld2;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)uni_ptr(base), 0, dead ? 0 : P * ld2, RSRC_FLAGS);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = 16 * j + wave;
            const unsigned dst = lds0 + buf * 2 * IMG_B + (q >> 5) * IMG_B + (q & 31) * 1024;
            dma16_lds(rs, dst, dvoff[j], (inner * (1 + (q >> 5)) + h * DH) * 2);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qn[ks] = q_load16(rs, qvoff, (h * DH + 32 * ks) * 2);
    };

    // One problem: `qc` holds its Q rows (requested a problem ago), `qn` receives the next problem's.
#ifdef ISTVT_SATTN_STAMP
    unsigned sp_seg[6] = {0, 0, 0, 0, 0, 0}, sp_n = 0;
    unsigned long long sp_prev = 0;
#endif
    u32x4 qc[KS], qn[KS];
    auto problem = [&](const int prob, const int it, const int buf) {
        SP_STAMP(0);
        if (!by_dead) {
            issue(prob + (int)gridDim.x, buf ^ 1, qn);                 // the other buffer was released by the barrier that closed problem it - 1
            SP_STAMP(1);
            // this problem's four DMA pieces landed: younger are [5 stores of problem it - 1] + 4 pieces + 2 Q loads of it + 1
            if (it == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
        } else if (tile_live) {
            issue_q(prob + (int)gridDim.x, qn);                        // (the compiler waits for qc where it is first used)
            SP_STAMP(1);
        } else {
            SP_STAMP(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this problem's pieces (sent during the previous problem)
        }
        SP_STAMP(2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        SP_STAMP(3);
        if (by_dead && !tile_live) issue_all_pieces(prob + (int)gridDim.x, buf ^ 1);     // under the others' arithmetic

        const int h = __builtin_amdgcn_readfirstlane(prob % heads), bf = __builtin_amdgcn_readfirstlane(prob / heads);
        const char* Kimg = smem + buf * 2 * IMG_B;
        const char* Vimg = Kimg + IMG_B;
        f32x4 o[DT];
        float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0, 0, 0, 0};
        if (tile_live) {
            bf16x8 qf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[ks] = __builtin_bit_cast(bf16x8, qc[ks]);
            auto chunk = [&](const int c0, auto tail_c) {
                constexpr bool TAIL = decltype(tail_c)::value;
                f32x4 s[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    s[t] = f32x4{0, 0, 0, 0};
                    if (!TAIL || t < NTT) {
                        const int row = c0 + 16 * t + r;
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kimg + row * ROW_B + (((4 * ks + g) ^ ((row >> 1) & 7)) << 4));
                            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[t], 0, 0, 0);
                        }
                    }
                }
                float mx = -INFINITY;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (TAIL && t >= NTT) continue;
                    if (TAIL && t == NTT - 1) {                 // the one tile that can straddle the row end
#pragma unroll
                        for (int j = 0; j < 4; ++j) s[t][j] = (c0 + 16 * t + 4 * g + j < P) ? s[t][j] : -INFINITY;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) mx = fmaxf(mx, s[t][j]);
                }
                mx = group_max(mx) * c;
                const float m_new = fmaxf(m_run, mx);
                const float alpha = fast_exp2(m_run - m_new);
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (TAIL && t >= NTT) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float pv = fast_exp2(fmaf(s[t][j], c, -m_new));
                        s[t][j] = pv;
                        sum += pv;
                    }
                }
                sum = group_sum(sum);
                l_run = l_run * alpha + sum;
                m_run = m_new;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;
#pragma unroll
                for (int ss = 0; ss < NT / 2; ++ss) {
                    if (!TAIL || 2 * ss < NTT) {
                        const bf16x8 pf = acc_frag<bf16_t>(s[2 * ss], s[2 * ss + 1]);
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            // V^T fragment: rows c0 + 32 ss + 4 g + q (elements 0..3) and + 16 (elements 4..7), pair dt
                            typedef short4v __attribute__((address_space(3))) * lds_ptr;
                            typedef short short8v __attribute__((ext_vector_type(8)));
                            const int ra = c0 + 32 * ss + 4 * g + (r >> 2), rb = ra + 16;
                            const char* pa = Vimg + ra * ROW_B + ((dt ^ ((ra >> 1) & 3)) << 5) + ((r & 3) << 3);
                            const char* pb = Vimg + rb * ROW_B + ((dt ^ ((rb >> 1) & 3)) << 5) + ((r & 3) << 3);
                            const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pa));
                            const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pb));
                            short8v sv;
                            sv[0] = lo[0]; sv[1] = lo[1]; sv[2] = lo[2]; sv[3] = lo[3]; sv[4] = hi[0]; sv[5] = hi[1]; sv[6] = hi[2]; sv[7] = hi[3];
                            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sv), pf, o[dt], 0, 0, 0);
                        }
                    }
                }
            };
            chunk(0, std::false_type{});
            chunk(128, std::true_type{});
        }
        SP_STAMP(4);
        // ---- stores: 4 x 8 bytes of O per lane and the statistics (lanes of group 0), always five operations per wavefront
        const float inv = tile_live ? 1.0f / l_run : 0.f;
        {
            const char* obase = (const char*)out + ((long)bf * P * ldo + h * DH) * 2;
            const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)uni_ptr(obase), 0, (P - 1) * ldo2 + DH * 2, RSRC_FLAGS);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16x4 ov;
#pragma unroll
                for (int j = 0; j < 4; ++j) ov[j] = (bf16_t)(o[dt][j] * inv);
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ov), o_rs, ovoff + 32 * dt, 0, 0);
            }
            const char* sbase = (const char*)lse + ((long)bf * P * heads + h) * 8;
            const __amdgpu_buffer_rsrc_t s_rs = __builtin_amdgcn_make_buffer_rsrc((void*)uni_ptr(sbase), 0, ((P - 1) * heads + 1) * 8, RSRC_FLAGS);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 st = {__float_as_uint(m_run), __float_as_uint(inv)};
            __builtin_amdgcn_raw_buffer_store_b64(st, s_rs, svoff, 0, 0);
        }
        SP_STAMP(5);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                                   // every wavefront is done reading this buffer
        __builtin_amdgcn_sched_barrier(0);
        SP_STAMP(6);
#ifdef ISTVT_SATTN_STAMP
        ++sp_n;
#endif
    };
    int prob = blockIdx.x;
    if (!by_dead) issue(prob, 0, qc);
    else if (tile_live) issue_q(prob, qc);
    else issue_all_pieces(prob, 0);
    for (int it = 0; prob < nprob; ++it, prob += gridDim.x) {
        problem(prob, it, it & 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qc[ks] = qn[ks];           // (requested a whole problem ago: landed long since)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the out-of-range requests of the problem past the last
#ifdef ISTVT_SATTN_STAMP
    if (lane == 0 && g_sattn_stamps) {
        unsigned long long* d = g_sattn_stamps + ((long)blockIdx.x * 16 + wave) * 8;
#pragma unroll
        for (int j = 0; j < 6; ++j) d[j] = sp_seg[j];
        d[6] = sp_n;
    }
#endif
}
