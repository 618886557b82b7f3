// gemm256t: the weight-gradient ("TN") 256x256 bf16 GEMM on the unit / ping-pong structure of gemm256q.h.
//
//   C[m][n] (+ split z) = sum_{k in split z} A[k][m] * B[k][n]       A = dy [Ktok][M], B = x [Ktok][N], C fp32
//
// Both operands are ROW-contiguous in the reduction index (k = token), so a unit is [64 k][128 columns] (256-byte
// rows, 16 KiB) and the MFMA fragments (8 consecutive k for one column) are read with ds_read_b64_tr_b16.
// One workgroup = one (tile, reduction split); the splits write fp32 partial slabs that istvt_splitk_reduce sums.
//
//   unit stream per K tile kt (u = 4 kt + j):  AL (A columns 0..127), BL, BH, AH (A columns 128..255); slot u & 7
//   DMA piece = 4 k-rows x 256 B; wave w stages k-rows 4w..4w+3 and 32+4w..32+4w+3 of every unit
//   image: 16-byte chunk c of k-row k at position c ^ 2 f(k), f(k) = (k & 3) | ((k >> 3) & 1) << 2: the 32 lanes of
//   one ds_read_b64_tr_b16 group read k-rows {k0..k0+3} and {k0+8..k0+11} at one column offset -> 8 distinct
//   32-byte bank groups
//   rows past the end of the reduction range / of the matrix are out of range for the whole-matrix descriptor and
//   read as zeros; columns past M / N of the last tile read neighbouring (finite or not) data that only reaches
//   output rows / columns which are never stored.
// Phases, slots, stagger, hazards and vmcnt counts are those of gemm256q.h (two phases of 32 MFMA per K tile).
#pragma once

// -DISTVT_T_DIAG=n (diagnostic builds only, tools/build_variant.sh + tools/gemm_bench.py with GB_LIB): bit 0 drops the
// MFMAs, bit 1 the LDS fragment reads, bit 2 the operand DMA of the K loop -- which of the three the K tile's time follows.
// -DISTVT_T_ORDER=1: the round-1 order inside a load slot (fragment reads, then the DMA issue) for A/B runs.
#ifndef ISTVT_T_DIAG
#define ISTVT_T_DIAG 0
#endif
// -DISTVT_T_RING10=0: the 8-unit LDS ring of round 1 (two K tiles) instead of 10 units (two and a half)
#ifndef ISTVT_T_RING10
#define ISTVT_T_RING10 1
#endif
#ifndef ISTVT_T_ORDER
#define ISTVT_T_ORDER 0
#endif
// ISTVT_T_SCHED: 0 = the schedule of rounds 2-3 (every fragment read in the load slots, ring of ten units).  1 = round 4:
// the schedule of gemm256q.h's ISTVT_Q_SCHED 1 -- a load slot of this kernel carried 32 transposing fragment reads besides
// its four DMA pieces (twice the LDS instructions of the NT kernel's), against 32 MFMAs in the other group's slot:
//      L_A(k)   DMA AL(k+1), AH(k+1)    reads B(k) (8 fragments)            wait: AH(k) landed       vmcnt(8)
//      C_A(k)   32 MFMA AL(k) x B(k), the 8 fragments of AL(k) read between them
//      L_B(k)   DMA BL(k+2), BH(k+2)    reads AH(k) (8 fragments)           wait: B(k+1) landed      vmcnt(8)
//      C_B(k)   32 MFMA AH(k) x B(k)                                        wait: AL(k+1) landed     vmcnt(6)
// unit X(k) in ring slot 4 (k & 1) + {AL 0, BL 1, BH 2, AH 3} (eight units; the epilogue slabs are separate again); WAR / RAW
// / vmcnt reasoning as in gemm256q.h.  K tiles past the split's end are requested like any other: their rows are out of
// range for the descriptors, so they cost no traffic, write zeros into slots nobody reads and keep the vmcnt counts
// constant; the kernel drains them before its epilogue.
#ifndef ISTVT_T_SCHED
#define ISTVT_T_SCHED 1
#endif

#ifdef ISTVT_T_STAMP
// diagnostic build (-DISTVT_T_STAMP): per-segment s_memtime sums of the round-4 schedule, [workgroup][wavefront][16] u64 =
// 12 segments as in gemm256q.h's DBG 256, K-loop cycles, K-loop 100 MHz ticks, K tiles, -; tools/gemm_t_slots.py
__device__ unsigned long long* g_t_stamps = nullptr;
extern "C" int istvt_diag_t_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_t_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -4;
}
#define T_STAMP(i)                                                                                    \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if ((i) > 0 || kt > 0) ts_seg[((i) + 11) % 12] += (unsigned)(t_ - ts_prev);                   \
        ts_prev = t_;                                                                                 \
    } while (0)
#else
#define T_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ int tswz(int k) { return 2 * ((k & 3) | (((k >> 3) & 1) << 2)); }

// fragment of 8 consecutive k (k0 .. k0+7) for column col16 + r out of a [64][128] unit image
__device__ __forceinline__ bf16x8 t_frag(const char* img, int k0, int col16, int r) {
    typedef short4v __attribute__((address_space(3))) * lds_ptr;
    typedef short short8v __attribute__((ext_vector_type(8)));
    const int q = r >> 2, pp = r & 3;
    const int ka = k0 + q, kb = k0 + 4 + q;
    const int chunk = (col16 >> 3) + (pp >> 1);
    const char* pa = img + ka * 256 + ((chunk ^ tswz(ka)) << 4) + ((pp & 1) << 3);
    const char* pb = img + kb * 256 + ((chunk ^ tswz(kb)) << 4) + ((pp & 1) << 3);
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pa));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pb));
    short8v s;
    s[0] = lo[0]; s[1] = lo[1]; s[2] = lo[2]; s[3] = lo[3]; s[4] = hi[0]; s[5] = hi[1]; s[6] = hi[2]; s[7] = hi[3];
    return __builtin_bit_cast(bf16x8, s);
}

__device__ __forceinline__ bf16x8 diag_frag(int v) {
    typedef short short8v __attribute__((ext_vector_type(8)));
    const short x = (short)(0x3c00 | (v & 63));
    return __builtin_bit_cast(bf16x8, short8v{x, x, x, x, x, x, x, x});
}

// Workgroup placement.  The hardware deals workgroup ids round-robin to the 8 XCDs (id & 7), each with its own L2.
// xcd_chunk() (common.h) renumbers the grid so that the workgroups of one XCD hold a CONTIGUOUS run of the (split z, tile) list:
// with ~32 workgroups resident per XCD that run is one split's tiles in row-major order -- for a 3 x 12-tile
// gradient 3 row panels + 12 column panels feed 32 tiles (L2 hit rate ~75 % of the panel reads).  The earlier map
// ((tile & 7) taken as the XCD with the split on grid.z, which it is not when the tile count is not a multiple
// of 8) left every XCD with tiles of all splits: rocprof FETCH_SIZE 720 MB per launch against 206 MB of operands.

// One (tile, reduction split) of problem p: wg = the tile's index (row-major over the 256x256 output tiles), z = the
// reduction split.
__device__ __forceinline__ void gemm256t_body(const GemmArgs& p, const int wg, const int z) {
    // The ring: TNP slots of one unit PAIR each ((AL, BL) or (BH, AH): 32 KiB).  The K loop is bound by the latency of
    // the operand DMA times the bytes the ring lets be in flight, and the LDS is full -- so the epilogue staging (which
    // the K loop never touches: one tile per workgroup) aliases the ring and the ring gets a fifth slot: the producer
    // runs 8..11 units ahead of the consumer instead of 6..9.  Five is not a power of two: slot indices are counters
    // that wrap, the two pair bases of a K tile are scalars.
    constexpr bool R10 = ISTVT_T_RING10 != 0 && ISTVT_T_SCHED == 0;
    constexpr int TNP = R10 ? 5 : 4;
    __shared__ __attribute__((aligned(16))) char smem[TNP * 2 * QU_BYTES + (R10 ? 0 : 8 * PSLAB_BYTES)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int wm = wave >> 2, wn = wave & 3;
    const int lda = (int)p.lda, ldb = (int)p.ldb, ldc = (int)p.ldc;

    const int tiles_n = (p.N + T256 - 1) / T256;
    const int id = wg;
    const int bm0 = (id / tiles_n) * T256, bn0 = (id % tiles_n) * T256;
    const int k_begin = z * p.kper;
    const int k_end = min(p.K, k_begin + p.kper);
    const int nkt = (k_end - k_begin + 63) >> 6;
    const int total_u = nkt * 4;

    constexpr unsigned OOB = 0x80000000u, WINDOW = 0x7fffffffu, RSRC_FLAGS = 0x00020000u;
    auto uni_ptr = [](const void* q) -> char* {
        const unsigned long long u = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return (char*)(((unsigned long long)hi << 32) | lo);
    };
    // ---- producer ------------------------------------------------------------------------------------
    // piece i of a unit = k-rows i*32 + wave*4 + (lane >> 4), 16-byte chunk lane & 15 (swizzled on the source).
    // The descriptors end at the split's last k-row: the reduction tail reads zeros.
    unsigned va[2], vb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kr = i * 32 + wave * 4 + (lane >> 4);
        const int chunk = (lane & 15) ^ tswz(kr);
        va[i] = (unsigned)(kr * lda * 2 + chunk * 16);
        vb[i] = (unsigned)(kr * ldb * 2 + chunk * 16);
    }
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(p.A), 0, k_end * lda * 2, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(p.B), 0, k_end * ldb * 2, RSRC_FLAGS);
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#if ISTVT_T_SCHED == 1
    {
        const unsigned ring0 = (unsigned)(__SIZE_TYPE__)(lds_void*)smem + wave * 1024;
        int ka = 0, kb = 0;                                // the K tiles whose A / B units go out next
        auto issue_a = [&](const int hi) {                 // AL (hi = 0) / AH (hi = 1) of K tile ka: two pieces
            if (ISTVT_T_DIAG & 4) return;
            const int sa = (k_begin + ka * 64) * lda * 2 + (bm0 + (hi ? 128 : 0)) * 2;
            const unsigned dst = ring0 + ((ka & 1) * 4 + (hi ? 3 : 0)) * QU_BYTES;
            dma16_lds(a_rs, dst, va[0], sa);
            dma16_lds(a_rs, dst + 8192, va[1], sa);
        };
        auto issue_b = [&](const int hi) {                 // BL / BH of K tile kb
            if (ISTVT_T_DIAG & 4) return;
            const int sb = (k_begin + kb * 64) * ldb * 2 + (bn0 + (hi ? 128 : 0)) * 2;
            const unsigned dst = ring0 + ((kb & 1) * 4 + 1 + hi) * QU_BYTES;
            dma16_lds(b_rs, dst, vb[0], sb);
            dma16_lds(b_rs, dst + 8192, vb[1], sb);
        };
        issue_b(0); issue_b(1); ++kb;                      // B(0)
        issue_a(0); issue_a(1); ++ka;                      // AL(0), AH(0)
        issue_b(0); issue_b(1); ++kb;                      // B(1)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // B(0), AL(0) landed (own pieces)
        slot_barrier();
        if (wm == 1) slot_barrier();                       // stagger: waves 4..7 run one slot behind
#ifdef ISTVT_T_STAMP
        unsigned ts_seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long ts_prev = 0;
        const unsigned long long ts_c0 = __builtin_amdgcn_s_memtime(), ts_r0 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
        for (int kt = 0; kt < nkt; ++kt) {
            const char* ubase = smem + (kt & 1) * 4 * QU_BYTES;
            const char* ua_lo = ubase;
            const char* ua_hi = ubase + 3 * QU_BYTES;
            const char* ub = ubase + (1 + (wn >> 1)) * QU_BYTES;
            bf16x8 af[4][2], bq[4][2];
            auto mma4 = [&](const int mt, const int t, const int kh) {
                if (ISTVT_T_DIAG & 1) return;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[nt][kh], af[t][kh], acc[mt][nt], 0, 0, 0);
            };
            auto fa = [&](const char* base, const int t, const int kh) -> bf16x8 {
                return (ISTVT_T_DIAG & 2) ? diag_frag(lane + t + kh) : t_frag(base, kh * 32 + 8 * g, wm * 64 + t * 16, r);
            };
            // ---- L_A
            T_STAMP(0);
            issue_a(0); issue_a(1); ++ka;
            T_STAMP(1);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
                    bq[t][kh] = (ISTVT_T_DIAG & 2) ? diag_frag(lane - t - kh) : t_frag(ub, kh * 32 + 8 * g, (wn & 1) * 64 + t * 16, r);
            T_STAMP(2);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            // AH(k) landed
            T_STAMP(3);
            slot_barrier();
            T_STAMP(4);
            // ---- C_A: the A fragments (i: t = i & 3, kh = i >> 2) two ahead of the MFMAs that use them
            af[0][0] = fa(ua_lo, 0, 0);
            af[1][0] = fa(ua_lo, 1, 0);
#define TSTEP(i)                                                                             \
            if ((i) + 2 < 8) af[((i) + 2) & 3][((i) + 2) >> 2] = fa(ua_lo, ((i) + 2) & 3, ((i) + 2) >> 2); \
            mma4((i) & 3, (i) & 3, (i) >> 2);
            TSTEP(0) TSTEP(1) TSTEP(2) TSTEP(3) TSTEP(4) TSTEP(5) TSTEP(6) TSTEP(7)
#undef TSTEP
            if (!(ISTVT_T_DIAG & 3)) {
                // a fragment is two ds_read_b64_tr_b16: 6 reads, then 4 MFMAs + 2 reads five times, then the MFMAs left
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            }
            T_STAMP(5);
            slot_barrier();
            T_STAMP(6);
            // ---- L_B
            issue_b(0); issue_b(1); ++kb;
            T_STAMP(7);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) af[t][kh] = fa(ua_hi, t, kh);
            T_STAMP(8);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            // B(k+1) landed
            T_STAMP(9);
            slot_barrier();
            T_STAMP(10);
            // ---- C_B
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int t = 0; t < 4; ++t) mma4(4 + t, t, kh);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");            // AL(k+1) landed
            T_STAMP(11);
            slot_barrier();
        }
#ifdef ISTVT_T_STAMP
        {
            const unsigned long long ts_c1 = __builtin_amdgcn_s_memtime(), ts_r1 = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (lane == 0 && g_t_stamps && blockIdx.x < 256) {
                unsigned long long* d = g_t_stamps + ((long)blockIdx.x * 8 + wave) * 16;
#pragma unroll
                for (int j = 0; j < 12; ++j) d[j] = ts_seg[j];
                d[12] = ts_c1 - ts_c0; d[13] = ts_r1 - ts_r0; d[14] = nkt; d[15] = 1;
            }
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the out-of-range pieces past the split's end
    }
#else
    int P = 0, p_s = 0, pp = 0;                         // pp: the producer's pair slot
    auto issue_pair = [&](const int J0) {              // J0 = 0 -> (AL, BL), J0 = 2 -> (BH, AH) of K tile p_s
        if (P >= total_u) return;
        char* img = smem + pp * (2 * QU_BYTES) + wave * 1024;
        pp = pp + 1 == TNP ? 0 : pp + 1;
        const int k0 = k_begin + p_s * 64;
        const int sa = k0 * lda * 2 + (bm0 + (J0 == 2 ? 128 : 0)) * 2;
        const int sb = k0 * ldb * 2 + (bn0 + (J0 == 2 ? 128 : 0)) * 2;
        if (ISTVT_T_DIAG & 4) { P += 2; if (J0 == 2) ++p_s; return; }
        const unsigned dst = (unsigned)(__SIZE_TYPE__)(lds_void*)img;
        if (J0 == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16_lds(a_rs, dst + i * 8192, va[i], sa);
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16_lds(b_rs, dst + QU_BYTES + i * 8192, vb[i], sb);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16_lds(b_rs, dst + i * 8192, vb[i], sb);
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16_lds(a_rs, dst + QU_BYTES + i * 8192, va[i], sa);
        }
        P += 2;
        if (J0 == 2) ++p_s;
    };
    issue_pair(0); issue_pair(2); issue_pair(0);           // units 0..5
    if (R10) issue_pair(2);                                // .. 7
    wait_vm_n(min(R10 ? 10 : 6, 2 * max(0, total_u - 3))); // units 0..2 landed (own pieces)
    slot_barrier();

    int U0 = 0, c0 = 0, c1 = 1;                 // c0 / c1: pair slots of the current K tile's (AL, BL) / (BH, AH)

    if (wm == 1) slot_barrier();               // stagger: waves 4..7 run one slot behind

    for (int kt = 0; kt < nkt; ++kt) {
        const char* ua_lo = smem + c0 * (2 * QU_BYTES);
        const char* ua_hi = smem + c1 * (2 * QU_BYTES) + QU_BYTES;
        const char* ub = (wn >> 1) ? smem + c1 * (2 * QU_BYTES) : ua_lo + QU_BYTES;
        c0 = c0 + 2 >= TNP ? c0 + 2 - TNP : c0 + 2;
        c1 = c1 + 2 >= TNP ? c1 + 2 - TNP : c1 + 2;
        bf16x8 af[4][2], bq[4][2];
        auto mma = [&](const int mt0) {
            if (ISTVT_T_DIAG & 1) return;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt0 + t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[nt][kh], af[t][kh], acc[mt0 + t][nt], 0, 0, 0);
        };
        auto load_a = [&](const char* base) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
                    af[t][kh] = (ISTVT_T_DIAG & 2) ? diag_frag(lane + t + kh) : t_frag(base, kh * 32 + 8 * g, wm * 64 + t * 16, r);
        };
        // ---- phase A: AL x B
        // The DMA of the units 6..7 ahead goes out FIRST (its slots were released by the barrier that opened this load
        // slot): the loop is bound by the latency of these requests, and issued after the fragment reads they started
        // ~0.2 us later in every phase.  Only possible with the opaque DMA form (dma16_lds, gemm_shared.h).
        if (ISTVT_T_ORDER == 0) issue_pair(R10 ? 0 : 2);                 // units U0+6, U0+7 (ring of 10: U0+8, U0+9)
        load_a(ua_lo);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
                bq[t][kh] = (ISTVT_T_DIAG & 2) ? diag_frag(lane - t - kh) : t_frag(ub, kh * 32 + 8 * g, (wn & 1) * 64 + t * 16, r);
        if (ISTVT_T_ORDER == 1) issue_pair(R10 ? 0 : 2);
        // unit U0+3 (AH) landed: all but the younger units' pieces (4 units, 6 with the ring of 10)
        if (total_u - 1 - (U0 + 3) >= (R10 ? 6 : 4)) wait_vm_n(R10 ? 12 : 8);
        else wait_vm_n(2 * max(0, total_u - 1 - (U0 + 3)));
        slot_barrier();
        mma(0);
        slot_barrier();
        // ---- phase B: AH x B
        if (ISTVT_T_ORDER == 0) issue_pair(R10 ? 2 : 0);                 // units U0+8, U0+9 (ring of 10: U0+10, U0+11)
        load_a(ua_hi);
        if (ISTVT_T_ORDER == 1) issue_pair(R10 ? 2 : 0);
        // units <= U0+6 (the next K tile's AL, BL, BH) landed
        if (total_u - 1 - (U0 + 6) >= (R10 ? 5 : 3)) wait_vm_n(R10 ? 10 : 6);
        else wait_vm_n(2 * max(0, total_u - 1 - (U0 + 6)));
        slot_barrier();
        mma(4);
        slot_barrier();
        U0 += 4;
    }
#endif
    if (wm == 0) slot_barrier();               // re-align the two groups

    // ---- epilogue: fp32 partial tile, wave-local 16-row passes through this wave's slab -----------------
    // (ring of 10: the slabs alias the ring -- every unit issued has been consumed and the re-aligning barrier above
    //  is behind every wavefront's last fragment read)
    float* slab = reinterpret_cast<float*>(smem + (R10 ? 0 : QNU * QU_BYTES) + wave * PSLAB_BYTES);
    const float alpha = p.alpha;
    const int colc = (lane & 7) * 8, erow = lane >> 3;
    const int row_w = wm * 64 + erow, col_w = wn * 64 + colc;
    const bool n_ok = bn0 + col_w < p.N;                   // N % 8 == 0: a chunk of 8 columns is all in or all out
    const int rows_left = p.M - bm0 - row_w;
    const unsigned c_off = n_ok ? (unsigned)((row_w * ldc + col_w) * 4) : OOB;
    const long c_org = ((long)z * p.slab + (long)bm0 * ldc + bn0) * 4;
    const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr((char*)p.C + c_org), 0, WINDOW, RSRC_FLAGS);
    const int re = lane & 15, ge = lane >> 4, l7 = lane & 7;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            *reinterpret_cast<f32x4*>(slab + re * 64 + (((nt * 4 + ge) ^ re) << 2)) = acc[pass][nt];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        u32x4 held[4];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = it * 8 + erow;
            const int rb = (pass >> 2) * 128 + (pass & 3) * 16 + it * 8;
            f32x4 lo = *reinterpret_cast<const f32x4*>(slab + row * 64 + (((2 * l7) ^ row) << 2));
            f32x4 hi = *reinterpret_cast<const f32x4*>(slab + row * 64 + (((2 * l7 + 1) ^ row) << 2));
            lo *= alpha; hi *= alpha;
            held[2 * it] = __builtin_bit_cast(u32x4, lo);
            held[2 * it + 1] = __builtin_bit_cast(u32x4, hi);
            const unsigned voff = rb < rows_left ? c_off : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(held[2 * it], c_rs, voff, rb * ldc * 4, 0);
            __builtin_amdgcn_raw_buffer_store_b128(held[2 * it + 1], c_rs, voff + 16, rb * ldc * 4, 0);
        }
        // STORE-DATA HAZARD, see gemm_shared.h: the data registers stay allocated and padded until the stores have read them
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(held[0]), "+v"(held[1]), "+v"(held[2]), "+v"(held[3])::"memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    }
}

// grid.x = tiles * splits
__global__ __launch_bounds__(512, 2) void gemm256t_kernel(GemmArgs p) {
    const int tiles = ((p.N + T256 - 1) / T256) * ((p.M + T256 - 1) / T256);
    const int L = xcd_chunk(blockIdx.x, gridDim.x);
    const int z = L / tiles;
    gemm256t_body(p, L - z * tiles, z);
}

// Several weight gradients with the same reduction length in ONE launch (all eight of a transformer layer: 120 tiles,
// so two reduction splits fill the chip instead of 7..42 per GEMM launched alone -- the fp32 slab traffic falls from
// 512 MB to 60 MB per layer and the launch has one ramp and one tail).  Problem i owns entries start[i] ..
// start[i+1]-1 (= tiles_i * splits of them, split-major) of the renumbered grid.
struct GemmGroupArgs {
    GemmArgs p[ISTVT_WGRAD_GROUP_MAX];
    int start[ISTVT_WGRAD_GROUP_MAX + 1];
    int n;
};

__global__ __launch_bounds__(512, 2) void gemm256t_group_kernel(GemmGroupArgs g) {
    const int L = xcd_chunk(blockIdx.x, gridDim.x);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < ISTVT_WGRAD_GROUP_MAX; ++i)
        if (i < g.n && L >= g.start[i]) pi = i;
    const GemmArgs& p = g.p[pi];
    const int l = L - g.start[pi];
    const int tiles = ((p.N + T256 - 1) / T256) * ((p.M + T256 - 1) / T256);
    const int z = l / tiles;
    gemm256t_body(p, l - z * tiles, z);
}
