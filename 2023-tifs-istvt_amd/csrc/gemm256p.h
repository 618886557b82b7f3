// gemm256p: PERSISTENT 256x256 bf16 NT GEMM -- one workgroup per CU walks its tiles and feeds ONE continuous
// stream of 32-deep K steps through the 4-slot LDS ring of gemm256r.h.
//
// Why: with one launch-workgroup per tile every tile paid ~12-16 us that no MFMA ran in (K = 728 is only
// 23 steps = ~20 us of K loop): a prologue in which all 256 CUs fetch their first 96 KiB at once, and an epilogue
// in which all 256 CUs load the residual and store 128 KiB at once, HBM idle in between.  Here the ring never
// drains between tiles: the DMAs of the next tile's first three steps are issued during the last three steps of
// the current tile and land while the epilogue runs; the epilogue's stores retire under the next tile's K loop;
// the bias is folded into the accumulators early in the K loop; the residual / GELU-input rows are requested
// 32-row quarters ahead of their use.
//
//   LDS: 4 ring slots x 32 KiB + 8 wavefronts x 4 KiB epilogue slab = 160 KiB (the whole CU).
//   epilogue slab: [16 rows][64 f32], 16-byte chunk c of row r at position c ^ r  (conflict-free for the
//   MFMA-layout ds_write_b128 and for the row-major ds_read_b128; no padding fits in 4 KiB).
//
// vmcnt discipline.  vmcnt counts LDS-DMA, loads and stores together and retires them in issue order, so
// `s_waitcnt vmcnt(n)` means "everything but the n youngest operations is done".  All waits in this kernel are
// written by hand with n = the number of operations GUARANTEED to have been issued after the one that is needed
// (anything else that happens to be in flight only makes the wait longer, never too short):
//   * step S needs stage S: guaranteed younger = stages S+1, S+2 where they exist (4 DMA instructions each);
//   * the epilogue's loads are inline-asm buffer loads, tied to their wait through "+v" operands.  They must not
//     be compiler-visible loads: while an LDS-DMA is pending the compiler's own bookkeeping gives up and emits
//     vmcnt(0) before the first use of any loaded value, which drains the ring once per tile;
//   * lanes outside the matrix use the buffer instructions' range check (offset >= num_records: loads return 0,
//     stores are dropped) instead of branches, so every lane issues the same number of operations.
#pragma once

constexpr int PSLAB_BYTES = 4096;

__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
    u32x4 v;
    // s_nop 4: the compiler does not know this statement is a VMEM instruction, so it does not pad the 5 wait states a
    // VMEM read of an SGPR needs after a VALU wrote it (v_readfirstlane of a descriptor word, v_readlane of a spilled
    // scalar offset): without them the load went out with the PREVIOUS value of the scalar offset (seen on gfx950).
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    return v;
}
// wait until at most 4*n operations are outstanding (n = 0, 1, 2; uniform)
__device__ __forceinline__ void wait_stages(int n) {
    if (n >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// EPI: 0 plain, 1 GELU forward (C = u, C2 = gelu(u)), 2 GELU backward (C = acc * gelu'(C2)).  SIDE: EPI 0 adds the
// residual rows.  Output is bf16; bias optional at run time (needs alpha == 1: it is added to the accumulators).
// Requires K > 96 (four or more K steps).  Everything else (fp32 / atomic outputs, split-K, short K) stays on
// gemm256r_kernel.
template <int EPI, bool SIDE>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[NSLOT * SLOT_BYTES + 8 * PSLAB_BYTES];
    constexpr bool HAS_SIDE = SIDE || EPI == EPI_GELU_BWD;
    constexpr bool LATE_Q3 = EPI == EPI_GELU_BWD;    // gelu' needs the registers: its last side quarter is requested after pass 1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int wm = wave >> 2, wn = wave & 3;

    const int tiles_n = (p.N + T256 - 1) / T256, tiles_m = (p.M + T256 - 1) / T256;
    const int nwg = tiles_n * tiles_m;
    const int G = gridDim.x;                                   // multiple of 8 (or == nwg): workgroup b stays on XCD b % 8
    const int my_tiles = (nwg - (int)blockIdx.x + G - 1) / G;
    const int nsteps = (p.K + RBK - 1) / RBK;
    const int total = my_tiles * nsteps;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;

    // i-th tile of this workgroup: the launch order b + i*G is remapped so that an XCD's workgroups walk
    // consecutive tiles of a contiguous range (their A panels and B share that XCD's L2), in groups of gm row panels
    auto tile_origin = [&](int i, int& bm0, int& bn0) {
        int id = (int)blockIdx.x + i * G;
        const int xcd = id & 7, q = nwg >> 3, rem = nwg & 7;
        id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (id >> 3);
        const int gm = p.gm > 0 ? p.gm : 1;
        const int per_group = gm * tiles_n;
        const int grp = id / per_group, idl = id % per_group;
        const int rows_here = min(gm, tiles_m - grp * gm);
        bm0 = (grp * gm + idl % rows_here) * T256;
        bn0 = (idl / rows_here) * T256;
    };

    // ---- producer: the stage stream ----------------------------------------------------------------
    unsigned oa[2], ob[2];
    const bf16_t *a_tile = A, *b_tile = B;
    int P = 0, p_s = 0, p_i = 0;
    auto p_setup = [&](int i) {
        int bm0, bn0;
        tile_origin(i, bm0, bn0);
        r_offs_nt(oa, p.lda, bm0, p.M, wave, lane);
        r_offs_nt(ob, p.ldb, bn0, p.N, wave, lane);
        a_tile = A + (long)bm0 * p.lda;
        b_tile = B + (long)bn0 * p.ldb;
    };
    auto issue = [&]() {
        if (P >= total) return;
        char* a_img = smem + (P & (NSLOT - 1)) * SLOT_BYTES;
        char* b_img = a_img + SLOT_BYTES / 2;
        const int k0 = p_s * RBK;
        const int krem = p.K - k0;
        if (krem >= RBK) { r_stage_nt(a_img, a_tile + k0, oa, wave); r_stage_nt(b_img, b_tile + k0, ob, wave); }
        else { r_stage_nt_tail(a_img, a_tile + k0, oa, krem, wave, lane); r_stage_nt_tail(b_img, b_tile + k0, ob, krem, wave, lane); }
        ++P;
        if (++p_s == nsteps) {
            p_s = 0;
            if (++p_i < my_tiles) p_setup(p_i);
        }
    };
    if (my_tiles > 0) p_setup(0);
    issue(); issue(); issue();

    float* slab = reinterpret_cast<float*>(smem + NSLOT * SLOT_BYTES + wave * PSLAB_BYTES);
    const float alpha = p.alpha;
    const int colc = (lane & 7) * 8;
    const int erow = lane >> 3;                // row within an 8-row half pass
    // offsets of the buffer instructions are relative to the tile origin (a 2 GiB window); OOB >= num_records
    constexpr unsigned OOB = 0x80000000u, WINDOW = 0x7fffffffu, RSRC_FLAGS = 0x00020000u;
    const char* side = EPI == EPI_GELU_BWD ? (const char*)p.C2 : (const char*)p.residual;     // rows read beside C
    const long lds_ = EPI == EPI_GELU_BWD ? p.ldc : p.ldr;
    const int row_w = wm * 128 + erow, col_w = wn * 64 + colc;      // this lane's first row / column inside the tile

    int S = 0;                                 // global step index of the stream
    for (int ti = 0; ti < my_tiles; ++ti) {
        int bm0, bn0;
        tile_origin(ti, bm0, bn0);
        const long c_org = ((long)bm0 * p.ldc + bn0) * 2;
        const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((char*)p.C + c_org, 0, WINDOW, RSRC_FLAGS);
        const __amdgpu_buffer_rsrc_t c2_rs =
            __builtin_amdgcn_make_buffer_rsrc(EPI == EPI_GELU_FWD ? (char*)p.C2 + c_org : (char*)p.C, 0, WINDOW, RSRC_FLAGS);
        const __amdgpu_buffer_rsrc_t s_rs = __builtin_amdgcn_make_buffer_rsrc(
            HAS_SIDE ? const_cast<char*>(side) + ((long)bm0 * lds_ + bn0) * 2 : (char*)p.C, 0, WINDOW, RSRC_FLAGS);
        // per-lane offsets, (re)computed where they are first needed -- kept live across the K loop they cost the
        // registers that made the compiler spill (and a spill reload is a vmcnt(0) in the middle of the DMA stream)
        bool n_ok; int rows_left; unsigned c_off, s_off;
        auto lane_offsets = [&]() {
            int row_o = row_w, col_o = col_w;
            asm volatile("" : "+v"(row_o), "+v"(col_o));            // opaque: not hoistable above this point
            n_ok = bn0 + col_o < p.N;
            rows_left = p.M - bm0 - row_o;                          // row rb of this lane is valid while rb < rows_left
            c_off = n_ok ? (unsigned)(((long)row_o * p.ldc + col_o) * 2) : OOB;
            s_off = n_ok ? (unsigned)(((long)row_o * lds_ + col_o) * 2) : OOB;
        };
        // side rows of the tile (residual / GELU input), 16 x 16 B per lane in four quarters of 32 rows: quarter 0
        // is requested before the MFMAs of the last K step, 1..3 when the epilogue starts (the fragment registers
        // are free by then) -- before the first store where the registers allow it, because a load issued behind
        // a store cannot be consumed before that store is acknowledged
        u32x4 sv[16];
        auto fetch_side = [&](int quarter) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rb = (quarter * 4 + q) * 8;
                sv[quarter * 4 + q] = buf_load16(s_rs, rb < rows_left ? s_off : OOB, rb * (int)lds_ * 2);
            }
        };
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int p_at_bias = 0;

        // one 32-deep K step.  FIRST also requests the bias; LAST (peeled) requests side quarter 0.
        auto kstep = [&](bool first, bool last) {
            wait_stages(min(2, total - 1 - S));
            __builtin_amdgcn_s_barrier();      // stage S visible to all; every wave is done reading slot (S-1)&3
            asm volatile("" ::: "memory");
            issue();                           // stage S+3 -> slot (S+3)&3 == (S-1)&3
            if (first && p.bias) {
                // this wavefront's 64 bias values -> slab[0..63] by one 4-byte-per-lane LDS-DMA (columns past N are
                // clamped: they feed accumulator columns that are never stored); no registers held across steps
                const int col = min(bn0 + wn * 64 + lane, p.N - 1);
                __builtin_amdgcn_global_load_lds((glb_void*)(p.bias + col), (lds_void*)slab, 4, 0, 0);
                p_at_bias = P;
            }
            if (last) { lane_offsets(); if (HAS_SIDE) fetch_side(0); }
            const char* a_img = smem + (S & (NSLOT - 1)) * SLOT_BYTES;
            const char* b_img = a_img + SLOT_BYTES / 2;
            bf16x8 bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) bf[t] = r_frag_nt(b_img, wn * 64 + t * 16 + r, g);
#pragma unroll
            for (int h = 0; h < 2; ++h) {      // A fragments in two halves: 16 fewer live registers than all eight at once
                bf16x8 af[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) af[t] = r_frag_nt(a_img, wm * 128 + (h * 4 + t) * 16 + r, g);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[h * 4 + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[nt], af[mt], acc[h * 4 + mt][nt], 0, 0, 0);
            }
            ++S;
        };
        kstep(true, false);
        kstep(false, false);
        kstep(false, false);
        if (p.bias) {
            // requested three steps ago; guaranteed younger: the stages issued since (two, unless the stream is ending)
            wait_stages(min(2, P - p_at_bias));
            asm volatile("" ::: "memory");
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(slab + nt * 16 + 4 * g);
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) acc[mt][nt] += bq;
            }
        }
        for (int s = 3; s < nsteps - 1; ++s) kstep(false, false);
        kstep(false, true);

        // ---- epilogue: wave-local, 8 passes of 16 rows through this wave's slab ------------------------
        if (HAS_SIDE) { fetch_side(1); if (!LATE_Q3) { fetch_side(2); fetch_side(3); } }
        // slab addresses are recomputed per tile from an opaque copy of the lane id: as loop invariants they would
        // be hoisted out of the tile loop and held (or spilled) across every K step
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int re = lane_e & 15, ge = lane_e >> 4, rowe = lane_e >> 3, l7 = lane_e & 7;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            if (LATE_Q3 && pass == 2) fetch_side(2);       // into the registers quarter 0 leaves
            if (LATE_Q3 && pass == 4) fetch_side(3);
            if (HAS_SIDE && (pass & 1) == 0) {
                // quarter pass/2 is needed.  Issued after it (guaranteed): the later quarters and two stores per
                // finished pass -- 12 operations at every even pass; with late requests (one quarter ahead, two passes
                // before its use) 4 / 8 / 8 / 4 at passes 0 / 2 / 4 / 6
                u32x4 &s0 = sv[pass * 2], &s1 = sv[pass * 2 + 1], &s2 = sv[pass * 2 + 2], &s3 = sv[pass * 2 + 3];
                if (!LATE_Q3)
                    asm volatile("s_waitcnt vmcnt(12)" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)::"memory");
                else if (pass == 0 || pass == 6)
                    asm volatile("s_waitcnt vmcnt(4)" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)::"memory");
                else
                    asm volatile("s_waitcnt vmcnt(8)" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)::"memory");
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<f32x4*>(slab + re * 64 + (((nt * 4 + ge) ^ re) << 2)) = acc[pass][nt];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
            u32x4 held[2], held2[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 8 + rowe;
                const int rb = pass * 16 + it * 8;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + row * 64 + (((2 * l7) ^ row) << 2));
                const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + row * 64 + (((2 * l7 + 1) ^ row) << 2));
                float v[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = lo[j] * alpha; v[4 + j] = hi[j] * alpha; }
                const unsigned voff = rb < rows_left ? c_off : OOB;
                const int soff = rb * (int)p.ldc * 2;
                if (EPI == EPI_GELU_BWD) {
                    const bf16x8 u = __builtin_bit_cast(bf16x8, sv[pass * 2 + it]);
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const gf2 gg = gelu_grad_fast2(gf2{(float)u[j], (float)u[j + 1]});
                        v[j] *= gg.x; v[j + 1] *= gg.y;
                    }
                } else if (SIDE) {
                    const bf16x8 u = __builtin_bit_cast(bf16x8, sv[pass * 2 + it]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += (float)u[j];
                }
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
                held[it] = __builtin_bit_cast(u32x4, o);
                __builtin_amdgcn_raw_buffer_store_b128(held[it], c_rs, voff, soff, 0);
                if (EPI == EPI_GELU_FWD) {
                    bf16x8 o2;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const gf2 gv = gelu_fast2(gf2{v[j], v[j + 1]});
                        o2[j] = (bf16_t)gv.x; o2[j + 1] = (bf16_t)gv.y;
                    }
                    held2[it] = __builtin_bit_cast(u32x4, o2);
                    __builtin_amdgcn_raw_buffer_store_b128(held2[it], c2_rs, voff, soff, 0);
                }
            }
            // STORE-DATA HAZARD (observed on gfx950, not padded by the compiler): a VALU write to the data registers
            // of a buffer_store_dwordx4 with an SGPR soffset a few instructions after the store reached memory instead
            // of the store data (one dword, lanes 12..15 of every 16).  The data registers are therefore kept allocated
            // -- tied to this asm -- until the end of the pass, and padded with wait states before they can be reused.
            if (EPI == EPI_GELU_FWD)
                asm volatile("s_nop 15\n\ts_nop 15" : "+v"(held[0]), "+v"(held[1]), "+v"(held2[0]), "+v"(held2[1])::"memory");
            else
                asm volatile("s_nop 15\n\ts_nop 15" : "+v"(held[0]), "+v"(held[1])::"memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        }
    }
}
