// gemm256q: persistent 256x256 bf16 NT GEMM with 64-deep K tiles staged as four 16 KiB UNITS of 128-byte rows,
// consumed by two wave groups that alternate between a load slot and a 32-MFMA slot (ping-pong).
//
// Why (measured with tools/dma_probe.hip on MI355X, one workgroup per CU streaming the FF1 panels into LDS,
// no MFMA, no LDS reads):
//      row piece / row stride            TB/s chip-wide
//      64 B  (32-deep step) / 1456 B         10.4        <- the earlier 32-deep ring kernels at K = 728
//      64 B                 / 1536 B         15.0
//      128 B (64-deep)      / 1456 B         14.6
//      128 B                / 1536 B         22.5
// The LDS-DMA is priced per cache line touched in the CU's address path, not per byte and not by latency (ring
// depth changes nothing): a 64-byte piece of a 1456-byte-stride row touches up to two lines for half a line of
// data, and a 256x256 tile needs 19.6 TB/s of staging at the MFMA peak.  Hence whole 128-byte lines per row and
// step (BK = 64) and operands with line-aligned row strides of an odd line count (ops.pad_ld: 728 -> 832, 2912 -> 3008).
// In-kernel stamps of the first version (16-MFMA slots) then showed the slot, not the DMA, as the limit: a slot
// took ~445 cycles for 256 cycles of MFMA because ONE in-order wavefront needs ~5 cycles per instruction and the
// load slot carried ~45 of them plus the barrier round trip; so the slots are now 32 MFMA deep and the producer
// works from two whole-matrix buffer descriptors (tile / half / k position in the scalar offset, which gfx950
// includes in the range check -- tools/rc_probe) instead of rebuilding a descriptor per unit.
//
// Units of K tile kt, in stream order u = 4 kt + j, slot u & 7 (8 slots x 16 KiB = 128 KiB):
//      j = 0  AL  A rows   0..127        j = 1  BL  B rows (C columns)   0..127
//      j = 2  BH  B rows 128..255        j = 3  AH  A rows 128..255
// image [128 rows][64 k], 16-byte chunk c of row r at position c ^ ((r >> 1) & 7) (swizzle on the DMA source).
// Wavefront (wm, wn) = (wave >> 2, wave & 3) owns C rows {h*128 + wm*64 + 0..63, h = 0, 1} and C columns
// wn*64 + 0..63 (inside ONE B half): 8 x 4 accumulator tiles of 16x16.  A K tile is two phases of 32 MFMA:
//      phase A (p = 2 kt)      AL x B   needs AL and BL|BH    reads 8 A + 8 B fragments
//      phase B (p = 2 kt + 1)  AH x B   needs AH              reads 8 A fragments (B stays in registers)
// Every phase is a load slot L (the phase's ds_reads, the DMA of the unit pair 2p+6, 2p+7, the counted vmcnt for
// the NEXT phase, lgkmcnt(0)) and an MFMA slot C, each closed by an s_barrier.  Waves 4..7 run one barrier behind
// waves 0..3 inside a tile, so on every SIMD one wave is in L while its partner is in C; one extra barrier at either
// end of the tile re-aligns the groups so that both run the epilogue together.
//
// Hazards (unit U is issued in L of phase floor((U-6)/2) into the slot of unit U-8):
//   * WAR: unit U-8 is last read in phase r: AL, BL, BH of kt in 2kt, AH in 2kt+1.  The later group reads in slot
//     2r+1 and waits lgkmcnt(0) before that slot's barrier; the earlier group overwrites in slot 2p' >= 2r+2:
//     AL(kt+2), BL(kt+2) in phase 2kt+1, BH(kt+1), AH(kt+1) in phase 2kt.
//   * RAW: phase q's units are waited for (own pieces, counted vmcnt) at the end of L of phase q-1 by BOTH groups,
//     i.e. before the barriers that precede slot 2q, the first slot in which anybody reads them.
//   * vmcnt: phase 2kt+1 needs AH = 4kt+3, issued by then <= 4kt+7: 8 pieces may stay in flight; phase 2kt+2 needs
//     units <= 4kt+6, issued <= 4kt+9: 6 pieces.  Fewer exist only at the end of the stream.
#pragma once
#include <type_traits>

// -DISTVT_Q_ORDER=1: the round-1 order inside a load slot (fragment reads, then the DMA issue) for A/B runs
#ifndef ISTVT_Q_ORDER
#define ISTVT_Q_ORDER 0
#endif
// ISTVT_Q_SCHED: 0 = the schedule above (rounds 2-3).  1 = round 4, after the slot stamps (profiles/r04_a_*): a load slot
// took 930 / 790 cycles against 520 of MFMA, ~400 of them the issue of the slot's 16 DMA pieces (the CU's address path
// takes ~25 cycles per 1 KiB piece whoever issues it, with or without MFMAs or LDS reads beside it) and ~230 / ~100 the
// fragment reads, and a wavefront's own DMA issue and LDS reads do not overlap in any order (reads first, interleaved, or
// split over the wavefronts: same sum).  So the A fragments of phase A are read INSIDE the MFMA slot, between the MFMAs
// (one ds_read_b128 per four MFMAs, two ahead), and a load slot carries four DMA pieces + eight fragment reads:
//      L_A(k)   DMA AL(k+1), AH(k+1)    reads B(k) (8)            wait: AH(k) landed       vmcnt(8)
//      C_A(k)   32 MFMA AL(k) x B(k), the 8 reads of AL(k) between them
//      L_B(k)   DMA BL(k+2), BH(k+2)    reads AH(k) (8)           wait: B(k+1) landed      vmcnt(8)
//      C_B(k)   32 MFMA AH(k) x B(k)                              wait: AL(k+1) landed     vmcnt(6)
// Unit X(k) lives in ring slot 4 (k & 1) + {AL 0, BL 1, BH 2, AH 3}, k = the stream-wide K tile index.  With waves 4..7 one
// slot behind (g1 = g0 + 1; g0's L_A(k) is slot 4k):
//   * WAR: AL(k-1) is last read in g1's C_A(k-1) (slot 4k-2), AH(k-1) in g1's L_B(k-1) (4k-1): both before the barrier that
//     opens g0's L_A(k) (4k), which overwrites them; B(k) is last read in g1's L_A(k) (4k+1), overwritten from g0's L_B(k) (4k+2).
//   * RAW: a wavefront waits for its own pieces; every wavefront must have waited before the barrier that opens the first
//     slot in which anybody reads the unit.  AH(k): waited at the end of L_A(k) (g0 4k, g1 4k+1), first read in g0's L_B(k)
//     (4k+2).  B(k+1): waited at the end of L_B(k) (4k+2, 4k+3), first read in g0's L_A(k+1) (4k+4).  AL(k+1): waited at the
//     end of C_B(k) (4k+3, 4k+4), first read in g0's C_A(k+1) (4k+5).
//   * vmcnt: the issue order of a wavefront is ... AL(k), AH(k) | BL(k+1), BH(k+1) | AL(k+1), AH(k+1) | BL(k+2), BH(k+2) ...,
//     two pieces each; after AH(k) come 8 pieces by the end of L_A(k), after BH(k+1) 8 by the end of L_B(k), after AL(k+1) 6
//     by the end of C_B(k).  Past the end of the stream the producers keep issuing pieces that are out of range for every
//     lane (no traffic, zeros into ring slots nobody reads), so the counts never change; anything else in flight
//     (epilogue stores, side loads, the bias) only makes a wait cover more.  The kernel drains vmcnt before it ends.
//   Every unit is requested >= 3 slots before its wait (round 3: BH 2 slots).
#ifndef ISTVT_Q_SCHED
#define ISTVT_Q_SCHED 1
#endif

constexpr int QU_BYTES = 16384;
constexpr int QNU = 8;

__device__ __forceinline__ void wait_vm_n(int n) {      // uniform n; vmcnt(n) for the counts this kernel uses
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void slot_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
}

// EPI: 0 plain, 1 GELU forward (C = u, C2 = gelu(u)), 2 GELU backward (C = acc * gelu'(C2)).  SIDE: EPI 0 adds the
// residual rows.  Output is bf16; bias optional at run time (needs alpha == 1: it is added to the accumulators).
// Requires 16-byte aligned rows and operands
// smaller than 2 GiB (32-bit buffer offsets).
// DBG (diagnostic builds only, -DISTVT_GEMM_DIAG + ISTVT_GEMM_QDBG=n): 1 = no DMA inside the K loop, 2 = no MFMA,
// 4 = no LDS fragment reads, 8 = s_memtime stamps of block 0 (tile start / K loop end / epilogue end) into C2, 16 / 32 = every
// tile reads the FIRST A / B panel (all L2 hits: the staging rate without HBM misses), 128 = no output stores,
// 256 = slot stamps: s_memtime at every boundary inside a K tile (DMA issued / fragment reads back / vmcnt wait over /
// barrier passed / MFMAs issued / barrier passed, for both phases), summed per segment over all K tiles of the workgroup
// in scalar registers, plus the s_memtime and s_memrealtime span of every K loop (in-kernel clock = cycles / ticks x
// 100 MHz); every wavefront stores its sums (plus the epilogue's and
// the tile-to-tile gap's cycles) into C2 at [(workgroup * 8 + wavefront) * 24] once, at the end.  The
// stamps' own lgkmcnt(0) puts the fragment reads in front of the vmcnt wait: read the SHARES, not the run time
// (tools/gemm_slots.py; a stamp costs ~40 cycles, which every segment includes).  1024 = only the per-tile spans (K loop,
// epilogue, tile-to-tile gap) and the clock, three stamps per tile: the K tile's time in an otherwise unperturbed kernel.
// TM = rows of a C tile: 256, or 224 = AL unit (128 rows) + 96 rows of the AH unit (its last four DMA pieces are sent
// out of range: no traffic, zeros in LDS, same instruction and vmcnt counts), phase B then runs 3 instead of 4 row tiles
// (24 MFMA).  At M = 56 736 this turns 222 row tiles into 254: an N = 728 GEMM is 762 tiles = 2.98 rounds of 256 CUs of
// tiles that are 12.5 % shorter, instead of 666 = 2.6 rounds that cost 3.  The host picks per launch (gemm.hip).
// STATS: also accumulates, per output column, the sum and the sum of squares of the values it STORES (rounded to bf16) into
// the replicated double accumulators p.st_sum / p.st_sumsq: train-mode BatchNorm statistics of a 1x1 convolution's output
// without the separate pass that re-reads it (xception.py:44,57 -> :58,69,75).  Per lane 16 float partial sums (its 8
// columns), kept across the tiles of a workgroup while the column tile stays the same (N <= 256: the whole launch), then
// reduced over the 8 lanes that share the columns and added with 16 fp64 atomics per lane group.
// STATS = 2: only the sums (same replicated double accumulator, row 0): the bias gradient of the feed-forward's hidden layer,
// whose dy is this kernel's GELU-backward output (module.py:27) -- instead of a column-sum pass over the 330 MB tensor.
// (N = 2912 is 12 column tiles and a workgroup changes column with almost every tile, so it flushes per tile: float
// atomics straight into the 2912 addresses of the gradient made every launch 60 us longer -- same-address contention --,
// 32 replicas + istvt_stats_reduce_add do not.)
// KHALF (host: K % 64 in 1..32 and more than one K tile -- the model's 728 and 2912): the second 32-deep MFMA step of an
// output tile's LAST K tile multiplies zero padding only; that tile runs half its MFMAs and fragment reads (ktile1's
// HALF): 1/24 of the MFMAs at K = 728.  A template parameter, not a run-time branch: two copies of the last K tile behind
// a branch made the register allocator spill (300 bytes of scratch per lane).
// Cache-policy bits of the output stores: 2 = nt (non-temporal: the output lines stream through L2 instead of displacing
// the operand panels every tile of a column re-reads).  Measured at the model's shapes, same box (tools/gemm_bench.py):
// N=2912 K=728 plain 277..283 -> 269 us, GELU 336 -> 318, N=1536 131 -> 125, bias+residual N=1024 112 -> 97; K=512 N=728
// 51 -> 53; the train step 52.4..52.7 -> 52.1 ms (the consumers of the outputs lose nothing measurable).  sc1 (16) alone
// +1..3 %, nt + sc1 as nt.  The stem's statistics launches (64..728-column outputs) measured 3 % slower with nt: they keep 0.
// ISTVT_Q_DIRECT_EPI=1: the plain epilogue without its trip through LDS (see DIRECT_EPI in the kernel).  Built to test what
// bounds the epilogue; measured no faster (the CU's store path is the floor either way), so it is OFF by default.
#ifndef ISTVT_Q_DIRECT_EPI
#define ISTVT_Q_DIRECT_EPI 0
#endif
#ifndef ISTVT_Q_SIDE_NT
#define ISTVT_Q_SIDE_NT 0
#endif
#ifndef ISTVT_Q_STORE_AUX
#define ISTVT_Q_STORE_AUX (STATS == 1 ? 0 : 2)
#endif
// the GELU-forward epilogue's SECOND output, gelu(u): the A operand of the very next GEMM (module.py:28-30), while u is
// read a whole layer later (the backward pass)
#ifndef ISTVT_Q_STORE_AUX2
#define ISTVT_Q_STORE_AUX2 ISTVT_Q_STORE_AUX
#endif
template <int EPI, bool SIDE, int DBG = 0, int TM = 256, int STATS = 0, bool KHALF = false>
__global__ __launch_bounds__(512, 2) void gemm256q_kernel(GemmArgs p) {
    static_assert(TM == 256 || TM == 224, "row tile");
    static_assert(STATS != 1 || (EPI == 0 && !SIDE && TM == 256), "BatchNorm statistics ride in the plain epilogue only");
    static_assert(STATS != 2 || (EPI == EPI_GELU_BWD && TM == 256), "column sums ride in the GELU-backward epilogue only");
    constexpr int NB = (TM - 128) / 32;         // 16-row tiles of the AH unit per wavefront: 4 or 3
    constexpr int HI_HALF = (TM - 128) / 2;     // AH rows per wm half: 64 or 48
    __shared__ __attribute__((aligned(16))) char smem[QNU * QU_BYTES + 8 * PSLAB_BYTES];
    constexpr bool HAS_SIDE = SIDE || EPI == EPI_GELU_BWD;
    constexpr bool LATE_Q3 = EPI == EPI_GELU_BWD;
    constexpr bool DIRECT_EPI = ISTVT_Q_DIRECT_EPI != 0 && EPI == 0 && !SIDE && STATS == 0 && TM == 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int wm = wave >> 2, wn = wave & 3;
    const int lda = (int)p.lda, ldb = (int)p.ldb, ldc = (int)p.ldc;

    const int tiles_n = (p.N + T256 - 1) / T256, tiles_m = (p.M + TM - 1) / TM;
    const int nwg = tiles_n * tiles_m;
    const int G = gridDim.x;
    const int nkt = (p.K + 63) >> 6;
    // ---- the tile walk ---------------------------------------------------------------------------------------------
    // p.walk == 1 (slab walk; the host sends G % 8 == 0 and tiles_m >= 8): the hardware deals workgroup ids round-robin to
    // the 8 XCDs (id & 7, speed only), each with its own 4 MiB L2.  XCD x owns a contiguous SLAB of row panels
    // [sr0, sr0 + snr) x all column tiles and its G / 8 workgroups walk the slab's tiles in one shared order, workgroup
    // j taking entries j, j + G/8, ...: what runs at any moment is a window of G/8 consecutive entries.  The order is
    // column bands of p.band tiles (all of them when p.band == 0), inside a band groups of p.gm row panels, inside a
    // group row-fastest -- so the column tiles of a row panel sit within gm * band entries of each other and its A rows
    // are fetched from HBM once per band instead of once per column tile, and no A panel is ever shared by two XCDs.
    const int sxcd = (int)blockIdx.x & 7, sj = (int)blockIdx.x >> 3, sG = G >> 3;
    const int sq = tiles_m >> 3, srem = tiles_m & 7;
    const int sr0 = sxcd * sq + min(sxcd, srem), snr = sq + (sxcd < srem ? 1 : 0);
    const int my_tiles = p.walk == 1 ? max(0, (snr * tiles_n - sj + sG - 1) / sG) : (nwg - (int)blockIdx.x + G - 1) / G;
    const int total_u = my_tiles * nkt * 4;

    auto tile_origin_calc = [&](int i, int& bm0, int& bn0) {
        if (p.walk == 1) {
            const int e = sj + i * sG;
            const int bw = p.band > 0 ? min(p.band, tiles_n) : tiles_n, nb = (tiles_n + bw - 1) / bw;
            const int b = min(e / (snr * bw), nb - 1);
            const int el = e - b * snr * bw, wb = min(bw, tiles_n - b * bw);
            const int gm = p.gm > 0 ? p.gm : 1;
            const int grp = el / (gm * wb), idl = el - grp * gm * wb;
            const int rows_here = min(gm, snr - grp * gm);
            bm0 = (sr0 + grp * gm + idl % rows_here) * TM;
            bn0 = (b * bw + idl / rows_here) * T256;
            return;
        }
        int id = (int)blockIdx.x + i * G;
        const int xcd = id & 7, q = nwg >> 3, rem = nwg & 7;
        id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (id >> 3);
        if (p.band > 0) {
            // Wide outputs (N = 2912: 12 column tiles, W = 4.2 MB > one XCD's L2): column bands of p.band tiles, a band's
            // tiles row by row.  An XCD's contiguous eighth of the list then stays inside ONE band: its slice of W
            // (3 x 373 KB) is L2-resident and an A row panel is shared by the band's column tiles running side by side.
            const int bw = p.band, nb = (tiles_n + bw - 1) / bw;
            const int b = min(id / (tiles_m * bw), nb - 1);
            const int idl = id - b * tiles_m * bw, wb = min(bw, tiles_n - b * bw);
            bm0 = (idl / wb) * TM;
            bn0 = (b * bw + idl % wb) * T256;
            return;
        }
        const int gm = p.gm > 0 ? p.gm : 1;
        const int per_group = gm * tiles_n;
        const int grp = id / per_group, idl = id % per_group;
        const int rows_here = min(gm, tiles_m - grp * gm);
        bm0 = (grp * gm + idl % rows_here) * TM;
        bn0 = (idl / rows_here) * T256;
    };
    // The origins of this workgroup's first 64 tiles, one per lane, computed ONCE (all lanes at a time on the vector ALU) and
    // looked up with v_readlane.  The walk's integer divisions cost ~400 instructions per call and used to run three times
    // per tile inside the load slots and at the head of every tile: the r04 slot stamps showed ~125 cycles per load slot of
    // this bookkeeping.
    int tab_m, tab_n;
    tile_origin_calc(lane, tab_m, tab_n);
    auto tile_origin = [&](int i, int& bm0, int& bn0) {
        if (i < 64) {
            bm0 = __builtin_amdgcn_readlane(tab_m, i);
            bn0 = __builtin_amdgcn_readlane(tab_n, i);
        } else {
            int m, n;
            tile_origin_calc(i, m, n);
            bm0 = __builtin_amdgcn_readfirstlane(m);
            bn0 = __builtin_amdgcn_readfirstlane(n);
        }
    };

    constexpr unsigned OOB = 0x80000000u, WINDOW = 0x7fffffffu, RSRC_FLAGS = 0x00020000u;
    auto uni_ptr = [](const void* q) -> char* {          // provably wave-uniform to the compiler (no waterfall loops)
        const unsigned long long u = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return (char*)(((unsigned long long)hi << 32) | lo);
    };
    // ---- producer: the unit stream -------------------------------------------------------------------
    // piece i of a unit = half-rows i*64 + wave*8 + (lane >> 3); both pieces share the swizzled chunk.
    // One descriptor per operand over the whole matrix: rows past M / N are past num_records and read as zeros.
    const int hr0 = wave * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((hr0 >> 1) & 7);
    unsigned va[2], vb[2];
    const unsigned ah_dead = (TM == 224 && wave >= 4) ? 0x80000000u : 0u;    // AH rows 96..127 do not exist in a 224-row tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        va[i] = (unsigned)((hr0 + 64 * i) * lda * 2 + chunk * 16);
        vb[i] = (unsigned)((hr0 + 64 * i) * ldb * 2 + chunk * 16);
    }
    // (p.a_sel_col > 0: two planes of M rows; a row past M of the first plane then reads the second plane's first rows
    //  instead of zeros -- it only reaches output rows past M, which are never stored)
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(p.A), 0, p.M * lda * 2 + (p.a_sel_col > 0 ? p.a2_off : 0), RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(p.B), 0, p.N * ldb * 2, RSRC_FLAGS);
    int P = 0, p_s = 0, p_i = 0;
    int a_org = 0, b_org = 0;                     // byte offsets of the producer's tile inside A / B
    int KTQ = 0;                                  // K tiles consumed so far (diagnostic builds: DBG 1 stops the DMA after the first)
    auto p_setup = [&](int i) {
        int bm0, bn0;
        tile_origin(i, bm0, bn0);
        a_org = (DBG & 16) ? 0 : bm0 * lda * 2 + ((p.a_sel_col > 0 && bn0 >= p.a_sel_col) ? p.a2_off : 0);        // DBG 16 / 32: every tile streams the first A / B panel (L2 hits only)
        b_org = (DBG & 32) ? 0 : bn0 * ldb * 2;
    };
    // units P, P+1 of K tile p_s: J0 = 0 -> (AL, BL), J0 = 2 -> (BH, AH)
    auto issue_pair = [&](const int J0) {
        if (P >= total_u) return;
        if ((DBG & 1) && P >= 6) { P += 2; return; }
        char* img = smem + (P & (QNU - 1)) * QU_BYTES + wave * 1024;
        const int k0 = p_s * 64;
        // reduction tail: chunks past K get the top offset bit -> out of range -> zeros (a select, never a branch)
        const unsigned deadbit = (chunk * 8 >= p.K - k0) ? OOB : 0u;
        const int sa = a_org + k0 * 2 + (J0 == 2 ? 128 * lda * 2 : 0);
        const int sb = b_org + k0 * 2 + (J0 == 2 ? 128 * ldb * 2 : 0);
        const unsigned dst = (unsigned)(__SIZE_TYPE__)(lds_void*)img;
        if (J0 == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16_lds(a_rs, dst + i * 8192, va[i] | deadbit, sa);
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16_lds(b_rs, dst + QU_BYTES + i * 8192, vb[i] | deadbit, sb);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16_lds(b_rs, dst + i * 8192, vb[i] | deadbit, sb);
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16_lds(a_rs, dst + QU_BYTES + i * 8192, va[i] | deadbit | (i == 1 ? ah_dead : 0u), sa);
        }
        P += 2;
        if (J0 == 2 && ++p_s == nkt) {
            p_s = 0;
            if (++p_i < my_tiles) p_setup(p_i);
        }
    };
    // ---- SCHED 1: an A-side and a B-side producer, each at its own K tile of the stream ----
    constexpr int SCHED = ISTVT_Q_SCHED;
    int pa_i = 0, pa_s = 0, pa_par = 0, pb_i = 0, pb_s = 0, pb_par = 0;
    auto a_setup = [&](int i) {
        int bm0, bn0;
        tile_origin(i, bm0, bn0);
        a_org = (DBG & 16) ? 0 : bm0 * lda * 2 + ((p.a_sel_col > 0 && bn0 >= p.a_sel_col) ? p.a2_off : 0);
    };
    auto b_setup = [&](int i) { int bm0, bn0; tile_origin(i, bm0, bn0); b_org = (DBG & 32) ? 0 : bn0 * ldb * 2; };
    const unsigned ring0 = (unsigned)(__SIZE_TYPE__)(lds_void*)smem + wave * 1024;
    auto issue_a = [&](const int hi) {                 // the two pieces of AL (hi = 0) / AH (hi = 1) of the A producer's K tile
        if ((DBG & 1) && KTQ >= 1) return;
        if ((DBG & 512) && (wave & 3) != 0 && KTQ >= 1) return;      // timing probe: one wavefront in four issues DMA (wrong results)
        const int k0 = pa_s * 64;
        const unsigned dead = (pa_i >= my_tiles || chunk * 8 >= p.K - k0) ? OOB : 0u;
        const int sa = a_org + k0 * 2 + (hi ? 128 * lda * 2 : 0);
        const unsigned dst = ring0 + (pa_par * 4 + (hi ? 3 : 0)) * QU_BYTES;
        dma16_lds(a_rs, dst, va[0] | dead, sa);
        dma16_lds(a_rs, dst + 8192, va[1] | dead | (hi ? ah_dead : 0u), sa);
    };
    auto a_next = [&]() {
        pa_par ^= 1;
        if (++pa_s == nkt) { pa_s = 0; if (++pa_i < my_tiles) a_setup(pa_i); }
    };
    auto issue_b = [&](const int hi) {                 // BL (hi = 0) / BH (hi = 1) of the B producer's K tile
        if ((DBG & 1) && KTQ >= 1) return;
        if ((DBG & 512) && (wave & 3) != 0 && KTQ >= 1) return;      // timing probe: one wavefront in four issues DMA (wrong results)
        const int k0 = pb_s * 64;
        const unsigned dead = (pb_i >= my_tiles || chunk * 8 >= p.K - k0) ? OOB : 0u;
        const int sb = b_org + k0 * 2 + (hi ? 128 * ldb * 2 : 0);
        const unsigned dst = ring0 + (pb_par * 4 + 1 + hi) * QU_BYTES;
        dma16_lds(b_rs, dst, vb[0] | dead, sb);
        dma16_lds(b_rs, dst + 8192, vb[1] | dead, sb);
    };
    auto b_next = [&]() {
        pb_par ^= 1;
        if (++pb_s == nkt) { pb_s = 0; if (++pb_i < my_tiles) b_setup(pb_i); }
    };
    if constexpr (SCHED == 1) {
        if (my_tiles > 0) { a_setup(0); b_setup(0); }
        issue_b(0); issue_b(1); b_next();                  // B(0)
        issue_a(0); issue_a(1); a_next();                  // AL(0), AH(0)
        issue_b(0); issue_b(1); b_next();                  // B(1)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // B(0), AL(0) landed (own pieces)
        slot_barrier();
    } else {
        if (my_tiles > 0) p_setup(0);
        issue_pair(0); issue_pair(2); issue_pair(0);           // units 0..5
        wait_vm_n(min(6, 2 * max(0, total_u - 3)));            // units 0..2 landed (own pieces)
        slot_barrier();
    }

    float* slab = reinterpret_cast<float*>(smem + QNU * QU_BYTES + wave * PSLAB_BYTES);
    const float alpha = p.alpha;
    const bool gelu_d = p.gelu_d != 0;          // uniform: the GELU epilogues exchange gelu'(u) instead of u (flags bit 4)
    const int colc = (lane & 7) * 8;
    const int erow = lane >> 3;
    const char* side = EPI == EPI_GELU_BWD ? (const char*)p.C2 : (const char*)p.residual;
    const int lds_ = EPI == EPI_GELU_BWD ? ldc : (int)p.ldr;
    const int row_w = wm * 64 + erow, col_w = wn * 64 + colc;       // this lane's first row / column inside the tile
    // fragment address inside a unit: row (.. + r) * 128 + ((kh*4 + g) ^ ((r >> 1) & 7)) * 16; kh = 1 flips bit 6
    const unsigned la0 = (unsigned)(r * 128 + ((g ^ ((r >> 1) & 7)) << 4));
    const unsigned la_a[2] = {la0 + wm * 64 * 128, (la0 ^ 64u) + wm * 64 * 128};          // + unit base + t * 2048
    const unsigned la_h[2] = {la0 + wm * HI_HALF * 128, (la0 ^ 64u) + wm * HI_HALF * 128};  // the same inside the AH unit
    const unsigned la_b[2] = {la0 + (wn & 1) * 64 * 128, (la0 ^ 64u) + (wn & 1) * 64 * 128};
    const int b_unit = 1 + (wn >> 1);

    float st1[8], st2[STATS == 1 ? 8 : 1];
    int st_bn0 = -1;                            // column tile the partial sums belong to
#pragma unroll
    for (int j = 0; j < 8; ++j) st1[j] = 0.f;
#pragma unroll
    for (int j = 0; j < (STATS == 1 ? 8 : 1); ++j) st2[j] = 0.f;
    auto stats_flush = [&]() {
        if (st_bn0 < 0) return;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            st1[j] = xsum_lanes_8_16_32(st1[j]);
            if constexpr (STATS == 1) st2[j] = xsum_lanes_8_16_32(st2[j]);
        }
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));        // opaque per call: the lane predicate / column are not kept live (or spilled) across tiles
        const int col = st_bn0 + wn * 64 + (lane_s & 7) * 8;
        if ((lane_s >> 3) == 0 && col < p.N) {
            if constexpr (STATS == 1) {
                const long rep = (long)(blockIdx.x % ISTVT_STAT_REPLICAS) * 2 * p.N;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    atomicAdd(p.st_sum + rep + col + j, (double)st1[j]);
                    atomicAdd(p.st_sumsq + rep + col + j, (double)st2[j]);
                }
            } else {
                const long rep = (long)((blockIdx.x + wm) % ISTVT_STAT_REPLICAS) * 2 * p.N;
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(p.st_sum + rep + col + j, (double)st1[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) st1[j] = 0.f;
#pragma unroll
        for (int j = 0; j < (STATS == 1 ? 8 : 1); ++j) st2[j] = 0.f;
    };

    // DBG 256: per-segment cycle sums (wave-uniform, scalar registers)
    unsigned seg[12];
    unsigned long long t_prev = 0, clk_c = 0, clk_r = 0, ep_c = 0, gap_c = 0, t_ep_end = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) seg[j] = 0u;
    auto stamp = [&](const int i, const bool open) {      // closes segment i - 1 (unless `open`: the first stamp of a tile)
        if constexpr ((DBG & 256) != 0) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (!open) seg[(i + 11) % 12] += (unsigned)(t - t_prev);
            t_prev = t;
        }
    };

    int KT = 0;                                // K tiles consumed so far (stream-wide): slot parity
    int U0 = 0;                                // first unit of the current K tile
    for (int ti = 0; ti < my_tiles; ++ti) {
        int bm0, bn0;
        tile_origin(ti, bm0, bn0);
        if constexpr (STATS != 0) {
            if (bn0 != st_bn0) { stats_flush(); st_bn0 = bn0; }
        }
        // (the output / side descriptors are built where they are used: live across the K loop they cost 12 SGPRs)
        auto side_rs = [&]() {
            return __builtin_amdgcn_make_buffer_rsrc(
                uni_ptr(HAS_SIDE ? side + ((long)bm0 * lds_ + bn0) * 2 : (const char*)p.C), 0, WINDOW, RSRC_FLAGS);
        };
        bool n_ok; int rows_left; unsigned c_off, s_off;
        auto lane_offsets = [&]() {
            int row_o = row_w, col_o = col_w;
            asm volatile("" : "+v"(row_o), "+v"(col_o));
            n_ok = bn0 + col_o < p.N;
            rows_left = p.M - bm0 - row_o;
            c_off = n_ok ? (unsigned)((row_o * ldc + col_o) * 2) : OOB;
            s_off = n_ok ? (unsigned)((row_o * lds_ + col_o) * 2) : OOB;
        };
        // rows of epilogue piece idx = pass*2 + it (pass = accumulator row tile mt, it = 8-row half of it), relative
        // to row_w: (mt >> 2) * 128 + (mt & 3) * 16 + it * 8
#define QRB(idx) ((((idx) >> 3) * (128 - (TM == 224 ? wm * 16 : 0))) + ((((idx) >> 1) & 3) * 16) + (((idx) & 1) * 8))
        u32x4 sv[16];
        auto fetch_side = [&](int quarter) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = quarter * 4 + q;
                const int rb = QRB(idx);
                sv[idx] = buf_load16<ISTVT_Q_SIDE_NT != 0>(side_rs(), (rb < rows_left && !(TM == 224 && idx >= 14)) ? s_off : OOB, rb * lds_ * 2);
            }
        };
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        unsigned long long t_stamp[3];
        if (DBG & 8) t_stamp[0] = __builtin_amdgcn_s_memtime();
        if (wm == 1) slot_barrier();           // stagger: waves 4..7 run one slot behind inside the tile
        unsigned long long k0_c = 0, k0_r = 0;
        if constexpr ((DBG & 1280) != 0) {
            k0_c = __builtin_amdgcn_s_memtime(); k0_r = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (ti > 0) gap_c += k0_c - t_ep_end;
        }

        auto ktile1 = [&](const bool first, const bool last, auto half_tag) {
            constexpr bool HALF = decltype(half_tag)::value;
            const char* ubase = smem + (KT & 1) * 4 * QU_BYTES;
            const char* ua_lo = ubase;
            const char* ua_hi = ubase + 3 * QU_BYTES;
            const char* ub = ubase + b_unit * QU_BYTES;
            bf16x8 af[4][2], bq[4][2];
            auto rd = [&](const char* q, const int tag) -> bf16x8 {
                if (DBG & 4) return __builtin_bit_cast(bf16x8, make_uint4(KT + tag, lane, tag, 1));
                return *reinterpret_cast<const bf16x8*>(q);
            };
            auto mma4 = [&](const int mt, const int t, const int kh) {
                if (DBG & 2) { asm volatile("" ::"v"(af[t][kh])); asm volatile("" ::"v"(bq[t][kh])); return; }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[nt][kh], af[t][kh], acc[mt][nt], 0, 0, 0);
            };
            // ---- L_A: DMA AL(k+1), AH(k+1); the B fragments of this K tile
            stamp(0, first);
            if (first && p.bias) {
                // this wavefront's 64 bias values -> slab[0..63] by one 4-byte-per-lane LDS-DMA (columns past N clamped); it is
                // older than the 8 pieces of this K tile's two load slots, so the wait that closes L_B covers it
                int lane_b = lane;
                asm volatile("" : "+v"(lane_b));    // opaque per tile, as in stats_flush
                const int col = min(bn0 + wn * 64 + lane_b, p.N - 1);
                dma4_lds(__builtin_amdgcn_make_buffer_rsrc(uni_ptr(p.bias), 0, p.N * 4, RSRC_FLAGS),
                         (unsigned)(__SIZE_TYPE__)(lds_void*)slab, (unsigned)(col * 4), 0);
            }
            issue_a(0); issue_a(1); a_next();
            stamp(1, false);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bq[t][0] = rd(ub + la_b[0] + t * 2048, t);
                if constexpr (!HALF) bq[t][1] = rd(ub + la_b[1] + t * 2048, t + 4);
            }
            stamp(2, false);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            // AH(k) landed
            stamp(3, false);
            slot_barrier();
            stamp(4, false);
            // ---- C_A: AL x B, the A fragments read between the MFMAs (fragment i: t = i & 3, kh = i >> 2), two ahead
            af[0][0] = rd(ua_lo + la_a[0], 8);
            af[1][0] = rd(ua_lo + la_a[0] + 2048, 9);
#define QSTEP(i)                                                                                                        \
            if ((i) + 2 < 8) af[((i) + 2) & 3][((i) + 2) >> 2] = rd(ua_lo + la_a[((i) + 2) >> 2] + (((i) + 2) & 3) * 2048, 10 + (i)); \
            mma4((i) & 3, (i) & 3, (i) >> 2);
#define QSTEPH(i)                                                                                                       \
            if ((i) + 2 < 4) af[(i) + 2][0] = rd(ua_lo + la_a[0] + ((i) + 2) * 2048, 10 + (i));                           \
            mma4((i), (i), 0);
            if constexpr (HALF) {
                QSTEPH(0) QSTEPH(1) QSTEPH(2) QSTEPH(3)
                if (!(DBG & 6)) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                }
            } else {
            QSTEP(0) QSTEP(1) QSTEP(2) QSTEP(3) QSTEP(4) QSTEP(5) QSTEP(6) QSTEP(7)
            }
#undef QSTEP
#undef QSTEPH
            if (!HALF && !(DBG & 6)) {
                // the order the scheduler is to emit: 3 reads, then 4 MFMAs + 1 read five times, then the MFMAs left
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            }
            stamp(5, false);
            slot_barrier();
            stamp(6, false);
            // ---- L_B: DMA BL(k+2), BH(k+2); the fragments of AH
            issue_b(0); issue_b(1); b_next();
            stamp(7, false);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t >= NB) continue;
                af[t][0] = rd(ua_hi + la_h[0] + t * 2048, 20 + t);
                if constexpr (!HALF) af[t][1] = rd(ua_hi + la_h[1] + t * 2048, 24 + t);
            }
            if (last) { lane_offsets(); if (HAS_SIDE) fetch_side(0); }
            stamp(8, false);
            // B(k+1) landed; the side loads just issued are younger still
            if (last && HAS_SIDE) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            stamp(9, false);
            slot_barrier();
            stamp(10, false);
            // ---- C_B: AH x B
#pragma unroll
            for (int kh = 0; kh < (HALF ? 1 : 2); ++kh)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t >= NB) continue;
                    mma4(4 + t, t, kh);
                }
            // AL(k+1) landed
            if (last && HAS_SIDE) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            stamp(11, false);
            slot_barrier();
            if (last) stamp(12, false);
            ++KT;
            ++KTQ;
        };
        auto ktile0 = [&](const bool first, const bool last) {
            const char* ubase = smem + (KT & 1) * 4 * QU_BYTES;
            const char* ua_lo = ubase;
            const char* ua_hi = ubase + 3 * QU_BYTES;
            const char* ub = ubase + b_unit * QU_BYTES;
            bf16x8 af[4][2], bq[4][2];
            // (no s_setprio around the MFMAs: raising the computing wave's priority starved its SIMD partner's load slot,
            //  measured -4..6 % on the model's shapes; raising the loader's instead -4 %)
            auto mma = [&](const int mt0, const int nmt) {
                if (DBG & 2) {
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                        for (int t = 0; t < 4; ++t) { asm volatile("" ::"v"(af[t][kh])); asm volatile("" ::"v"(bq[t][kh])); }
                    return;
                }
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (t >= nmt) continue;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt0 + t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[nt][kh], af[t][kh], acc[mt0 + t][nt], 0, 0, 0);
                    }
            };
            auto load_a = [&](const char* base, const unsigned (&la)[2], const int nmt) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t >= nmt) continue;
                    if (DBG & 4) {
                        af[t][0] = __builtin_bit_cast(bf16x8, make_uint4(KT + t, lane, t, 1));
                        af[t][1] = __builtin_bit_cast(bf16x8, make_uint4(KT - t, lane, t, 2));
                    } else {
                        af[t][0] = *reinterpret_cast<const bf16x8*>(base + la[0] + t * 2048);
                        af[t][1] = *reinterpret_cast<const bf16x8*>(base + la[1] + t * 2048);
                    }
                }
            };
            // ---- phase A: AL x B
            // The DMA of the units 6..7 ahead goes out FIRST -- its slots were released by the barrier that opened this
            // load slot, and the loop is bound by how early these requests start -- and only then the fragment reads.
            // Needs the opaque DMA form (dma16_lds, gemm_shared.h): after the builtin the compiler drains vmcnt before
            // every LDS read.
            stamp(0, first);
            if (DBG & 64) __builtin_amdgcn_s_setprio(1);
            if (first && p.bias) {
                // this wavefront's 64 bias values -> slab[0..63] by one 4-byte-per-lane LDS-DMA (columns past N clamped);
                // issued BEFORE this slot's units, so that the wait at the end of phase B covers it
                int lane_b = lane;
                asm volatile("" : "+v"(lane_b));    // opaque per tile, as in stats_flush
                const int col = min(bn0 + wn * 64 + lane_b, p.N - 1);
                dma4_lds(__builtin_amdgcn_make_buffer_rsrc(uni_ptr(p.bias), 0, p.N * 4, RSRC_FLAGS),
                         (unsigned)(__SIZE_TYPE__)(lds_void*)slab, (unsigned)(col * 4), 0);
            }
            if (ISTVT_Q_ORDER == 0) issue_pair(2);                           // units U0+6, U0+7
            stamp(1, false);
            load_a(ua_lo, la_a, 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (DBG & 4) {
                    bq[t][0] = __builtin_bit_cast(bf16x8, make_uint4(KT + t, lane, t, 3));
                    bq[t][1] = __builtin_bit_cast(bf16x8, make_uint4(KT - t, lane, t, 4));
                } else {
                    bq[t][0] = *reinterpret_cast<const bf16x8*>(ub + la_b[0] + t * 2048);
                    bq[t][1] = *reinterpret_cast<const bf16x8*>(ub + la_b[1] + t * 2048);
                }
            }
            if (ISTVT_Q_ORDER == 1) issue_pair(2);
            stamp(2, false);
            // unit U0+3 (AH) landed: all but the 4 younger units (8 pieces); fewer exist only when the stream ends
            if (total_u - 1 - (U0 + 3) >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else wait_vm_n(2 * max(0, total_u - 1 - (U0 + 3)));
            if (DBG & 64) __builtin_amdgcn_s_setprio(0);
            stamp(3, false);
            slot_barrier();
            stamp(4, false);
            mma(0, 4);
            stamp(5, false);
            slot_barrier();
            stamp(6, false);
            // ---- phase B: AH x B
            if (DBG & 64) __builtin_amdgcn_s_setprio(1);
            if (ISTVT_Q_ORDER == 0) issue_pair(0);                           // units U0+8, U0+9
            stamp(7, false);
            load_a(ua_hi, la_h, NB);
            if (ISTVT_Q_ORDER == 1) issue_pair(0);
            if (last) { lane_offsets(); if (HAS_SIDE) fetch_side(0); }
            stamp(8, false);
            // units <= U0+6 (the next K tile's AL, BL, BH) landed; the side loads just issued are younger still
            if (total_u - 1 - (U0 + 6) >= 3) {
                if (last && HAS_SIDE) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                wait_vm_n(2 * max(0, total_u - 1 - (U0 + 6)) + ((last && HAS_SIDE) ? 4 : 0));
            }
            if (DBG & 64) __builtin_amdgcn_s_setprio(0);
            stamp(9, false);
            slot_barrier();
            stamp(10, false);
            mma(4, NB);
            stamp(11, false);
            slot_barrier();
            if (last) stamp(12, false);
            ++KT;
            U0 += 4;
        };
#if ISTVT_Q_SCHED == 1
#define ktile(f, l) ktile1(f, l, std::false_type{})
#else
#define ktile(f, l) ktile0(f, l)
#endif
        ktile(true, KHALF ? false : nkt == 1);      // (KHALF: the host guarantees more than one K tile)
        if (p.bias) {
            // the bias DMA was issued in phase A of the first K tile, before units U0-4+6..9: phase B's wait covered it
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 bqv = *reinterpret_cast<const f32x4*>(slab + nt * 16 + 4 * g);
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) acc[mt][nt] += bqv;
            }
        }
        for (int s = 1; s < nkt - 1; ++s) ktile(false, false);
#if ISTVT_Q_SCHED == 1
        if constexpr (KHALF) ktile1(false, true, std::true_type{});
        else
#endif
        if (nkt > 1) ktile(false, true);
        if constexpr ((DBG & 1280) != 0) {
            const unsigned long long k1_c = __builtin_amdgcn_s_memtime(), k1_r = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            clk_c += k1_c - k0_c; clk_r += k1_r - k0_r;
            t_ep_end = k1_c;                    // (the epilogue's start; overwritten by its end below)
        }

        if (wm == 0) slot_barrier();           // re-align the two groups: both run the epilogue together
        if (DBG & 8) t_stamp[1] = __builtin_amdgcn_s_memtime();

        // ---- epilogue: wave-local, 8 passes of 16 rows through this wave's slab ----
        const long c_org = ((long)bm0 * ldc + bn0) * 2;
        const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr((char*)p.C + c_org), 0, WINDOW, RSRC_FLAGS);
        const __amdgpu_buffer_rsrc_t c2_rs = __builtin_amdgcn_make_buffer_rsrc(
            uni_ptr(EPI == EPI_GELU_FWD ? (char*)p.C2 + c_org : (char*)p.C), 0, WINDOW, RSRC_FLAGS);
        if (HAS_SIDE) { fetch_side(1); if (!LATE_Q3) { fetch_side(2); fetch_side(3); } }
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int re = lane_e & 15, ge = lane_e >> 4, rowe = lane_e >> 3, l7 = lane_e & 7;
        // The slab round trip of pass p + 1 (four 16-byte writes, four reads back as output rows) is issued BEFORE pass p is
        // converted and stored: a wavefront's LDS operations execute in order, so the reads of pass p are older than the
        // writes of pass p + 1 that overwrite the slab, and each pass's LDS latency (twice per pass when the reads of its
        // second half followed the arithmetic of its first) hides under the previous pass's arithmetic and stores.
        constexpr int NPASS = TM == 224 ? 7 : 8;        // the AH unit holds three row tiles per wavefront at 224 rows
        // DIRECT (plain epilogue): no trip through LDS at all.  In the accumulator layout a lane (ge, re) holds row re,
        // columns 16 nt + 4 ge .. + 3 of block nt -- 8 bytes of bf16 per block; v_permlane16_swap of the block pairs (0, 1)
        // and (2, 3) leaves it with 8 CONTIGUOUS columns, 16 (2 ntp + (ge & 1)) + 8 (ge >> 1) .. + 7: one 16-byte store per
        // pair (a wave-instruction = 16 rows x 64 contiguous bytes; the two pairs complete the rows' 128-byte lines).
        // Why it was built (DBG 1024 / 1152 stamps, K = 728, N = 1536): the epilogue through LDS takes 6.5 k cycles per tile,
        // 6.0 k with its stores out of range and 5.6 k with no store instruction at all -- the LDS transposition alone is
        // 8 wavefronts x 8 passes x (4 ds_write_b128 at 13 cycles + 4 ds_read_b128 at 8) = 5.4 k cycles of the CU's LDS pipe.
        // What it measured: 4.1-4.7 k cycles with its stores out of range, but 7.5 k with them (7.9 k as 16 rows x 64 bytes
        // per instruction) -- whether 32 or 224 workgroups run.  128 KiB of output leave a CU at 17-20 bytes per clock
        // however they are issued; paced by the LDS round trips (6.5 k) or queued back to back (7.5 k) that is the
        // epilogue's floor, and launch times are equal within noise (tools/gemm_bench.py).  Kept for the record, not enabled.
        if constexpr (DIRECT_EPI) {
            int lane_d = lane;
            asm volatile("" : "+v"(lane_d));
            const int re = lane_d & 15, ge = lane_d >> 4, r1 = re & 7, hh = re >> 3;
            const int seg = 16 * (ge & 1) + 8 * (ge >> 1);                  // this lane's 8 columns inside a 32-column half
            // A second exchange makes every store instruction 8 rows x 128 contiguous bytes (whole lines; as 16 rows x 64
            // bytes the stores retired slowly: epilogue 7.9 k cycles against 4.1 k with the same stores out of range):
            // store 1 = rows 0..7 of the pass -- lanes re < 8 write their own columns 0..31 part, lanes re >= 8 the
            // columns 32..63 part of row re - 8, fetched from that lane (DPP row_ror:8); store 2 = rows 8..15, roles swapped.
            const int row_d = wm * 64 + r1;
            const int rows_left_d = p.M - bm0 - row_d;
            const int col1 = wn * 64 + seg + 32 * hh, col2 = wn * 64 + seg + 32 * (1 - hh);
            const unsigned voff1 = bn0 + col1 < p.N ? (unsigned)((row_d * ldc + col1) * 2) : OOB;
            const unsigned voff2 = bn0 + col2 < p.N ? (unsigned)((row_d * ldc + col2) * 2) : OOB;
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int rb = (pass >> 2) * 128 + (pass & 3) * 16;
                u32x2_t pk[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(acc[pass][nt][j] * alpha);
                    pk[nt] = __builtin_bit_cast(u32x2_t, o);
                }
                u32x4 half[2];                                              // [0]: columns 0..31 part, [1]: columns 32..63 part (row re)
#pragma unroll
                for (int ntp = 0; ntp < 2; ++ntp) {
                    const auto s0 = __builtin_amdgcn_permlane16_swap(pk[2 * ntp][0], pk[2 * ntp + 1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(pk[2 * ntp][1], pk[2 * ntp + 1][1], false, false);
                    half[ntp] = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
                u32x4 held[2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // the partner lane (re ^ 8, same ge) gives the part this lane stores for it
                    const unsigned x1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)half[1][q], 0x128, 0xf, 0xf, false);
                    held[0][q] = hh ? x1 : half[0][q];                    // rows 0..7: own left part | partner's right part
                    held[1][q] = hh ? half[0][q] : x1;                    // rows 8..15: own left part (re >= 8) | partner's right part
                }
                const unsigned v1 = (rb < rows_left_d && !(DBG & 128)) ? voff1 : OOB;
                const unsigned v2 = (rb + 8 < rows_left_d && !(DBG & 128)) ? voff2 : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(held[0], c_rs, v1, rb * ldc * 2, ISTVT_Q_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(held[1], c_rs, v2, (rb + 8) * ldc * 2, ISTVT_Q_STORE_AUX);
                asm volatile("s_nop 15\n\ts_nop 15" : "+v"(held[0]), "+v"(held[1])::"memory");      // STORE-DATA HAZARD, gemm_shared.h
            }
        } else {
        f32x4 xlo[2][2], xhi[2][2];
        auto slab_trip = [&](const int pass) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<f32x4*>(slab + re * 64 + (((nt * 4 + ge) ^ re) << 2)) = acc[pass][nt];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 8 + rowe;
                xlo[pass & 1][it] = *reinterpret_cast<const f32x4*>(slab + row * 64 + (((2 * l7) ^ row) << 2));
                xhi[pass & 1][it] = *reinterpret_cast<const f32x4*>(slab + row * 64 + (((2 * l7 + 1) ^ row) << 2));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        };
        slab_trip(0);
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (pass + 1 < NPASS) slab_trip(pass + 1);
            if (LATE_Q3 && pass == 2) fetch_side(2);
            if (LATE_Q3 && pass == 4) fetch_side(3);
            if (HAS_SIDE && (pass & 1) == 0) {
                u32x4 &s0 = sv[pass * 2], &s1 = sv[pass * 2 + 1], &s2 = sv[pass * 2 + 2], &s3 = sv[pass * 2 + 3];
                if (!LATE_Q3)
                    asm volatile("s_waitcnt vmcnt(12)" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)::"memory");
                else if (pass == 0 || pass == 6)
                    asm volatile("s_waitcnt vmcnt(4)" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)::"memory");
                else
                    asm volatile("s_waitcnt vmcnt(8)" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)::"memory");
            }
            u32x4 held[2], held2[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int rb = QRB(pass * 2 + it);
                const f32x4 lo = xlo[pass & 1][it], hi = xhi[pass & 1][it];
                float v[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = lo[j] * alpha; v[4 + j] = hi[j] * alpha; }
                const unsigned voff = (rb < rows_left && !(DBG & 128)) ? c_off : OOB;      // DBG 128: no output stores
                const int soff = rb * ldc * 2;
                if (EPI == EPI_GELU_BWD) {
                    const bf16x8 u = __builtin_bit_cast(bf16x8, sv[pass * 2 + it]);
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        if constexpr ((DBG & 2048) != 0) { v[j] *= (float)u[j]; v[j + 1] *= (float)u[j + 1]; continue; }   // no GELU arithmetic
                        if (gelu_d) { v[j] *= (float)u[j]; v[j + 1] *= (float)u[j + 1]; continue; }      // C2 holds gelu'(u), saved by the forward
                        const gf2 gg = gelu_grad_fast2(gf2{(float)u[j], (float)u[j + 1]});
                        v[j] *= gg.x; v[j + 1] *= gg.y;
                    }
                } else if (SIDE) {
                    const bf16x8 u = __builtin_bit_cast(bf16x8, sv[pass * 2 + it]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += (float)u[j];
                }
                bf16x8 o, o2e;
                if (EPI == EPI_GELU_FWD && gelu_d) {
                    // the forward keeps gelu'(u) instead of u (all the backward needs of it): one erf / exp pair gives both
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        gf2 gg;
                        const gf2 gv = gelu_both_fast2(gf2{v[j], v[j + 1]}, gg);
                        o[j] = (bf16_t)gg.x; o[j + 1] = (bf16_t)gg.y;
                        o2e[j] = (bf16_t)gv.x; o2e[j + 1] = (bf16_t)gv.y;
                    }
                } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
                }
                if constexpr (STATS != 0) {
                    if (voff != OOB) {                  // a row and a column chunk inside the matrix
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float t = (float)o[j];
                            st1[j] += t;
                            if constexpr (STATS == 1) st2[j] = fmaf(t, t, st2[j]);
                        }
                    }
                }
                held[it] = __builtin_bit_cast(u32x4, o);
                __builtin_amdgcn_raw_buffer_store_b128(held[it], c_rs, voff, soff, ISTVT_Q_STORE_AUX);
                if (EPI == EPI_GELU_FWD) {
                    bf16x8 o2;
                    if (gelu_d) o2 = o2e;
                    else
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        if constexpr ((DBG & 2048) != 0) { o2[j] = (bf16_t)(v[j] + 1.0f); o2[j + 1] = (bf16_t)(v[j + 1] + 1.0f); continue; }
                        const gf2 gv = gelu_fast2(gf2{v[j], v[j + 1]});
                        o2[j] = (bf16_t)gv.x; o2[j + 1] = (bf16_t)gv.y;
                    }
                    held2[it] = __builtin_bit_cast(u32x4, o2);
                    __builtin_amdgcn_raw_buffer_store_b128(held2[it], c2_rs, voff, soff, ISTVT_Q_STORE_AUX2);
                }
            }
            // STORE-DATA HAZARD, see gemm_shared.h
            if (EPI == EPI_GELU_FWD)
                asm volatile("s_nop 15\n\ts_nop 15" : "+v"(held[0]), "+v"(held[1]), "+v"(held2[0]), "+v"(held2[1])::"memory");
            else
                asm volatile("s_nop 15\n\ts_nop 15" : "+v"(held[0]), "+v"(held[1])::"memory");
        }
        }   // !DIRECT_EPI
#undef QRB
        if constexpr ((DBG & 1280) != 0) {
            // the epilogue as the wavefront sees it: its last store ISSUED (no drain: the stores retire under the next tile)
            const unsigned long long e1 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            ep_c += e1 - t_ep_end;
            t_ep_end = e1;
        }
        if (DBG & 8) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t_stamp[2] = __builtin_amdgcn_s_memtime();
            if (blockIdx.x == 0 && lane == 0 && ti < 16) {
                unsigned long long* d = (unsigned long long*)p.C2 + (wave * 16 + ti) * 3;
                d[0] = t_stamp[0]; d[1] = t_stamp[1]; d[2] = t_stamp[2];
            }
        }
    }
    if constexpr (STATS != 0) stats_flush();
    if constexpr (SCHED == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the out-of-range pieces past the stream's end
    if constexpr ((DBG & 1280) != 0) {
        if (lane == 0) {
            unsigned long long* d = (unsigned long long*)(p.dbg ? p.dbg : p.C2) + ((long)blockIdx.x * 8 + wave) * 24;
#pragma unroll
            for (int j = 0; j < 12; ++j) d[j] = seg[j];
            d[12] = clk_c; d[13] = clk_r; d[14] = (unsigned long long)my_tiles * nkt; d[15] = 1;
            d[16] = ep_c; d[17] = gap_c; d[18] = my_tiles;
        }
    }
}
