// gemm256r: the 256x256 bf16 GEMM with a 4-slot LDS ring of 32-deep K steps.
//
// Why: in gemm256_kernel (2 stages x 64-deep) the DMA of tile t+1 is issued after the barrier of
// tile t and must land before the next barrier, so only one tile's latency is covered by one tile
// of MFMA: measured the K loop ran at the DMA's latency-bound 31 GB/s per CU (~8 TB/s chip)
// whatever the tile order, with the MFMAs alone needing half that time.  Here three 32 KiB steps
// (96 KiB) stay in flight behind a COUNTED s_waitcnt vmcnt(8|4|0) and a raw s_barrier, so a step's
// DMA has three steps of MFMA to land.
//
//   slot = 32 KiB = A image [256 rows][32 k] (64 B rows) + B image, 4 slots = 128 KiB
//   NT swizzle: 16-byte chunk c of row r lives at position c ^ F[(r >> 2) & 3], F = {0,2,3,1}
//        (conflict-free for ds_read_b128's interleaved 16-lane groups on 64-byte rows)
//   TN: image [32 k][256 rows] (512 B rows), chunk ^ 2*(k & 7) as in gemm256.h
//   per step and wavefront: 12 ds_read_b128 (or 24 tr-reads), 32 MFMA 16x16x32, 4 DMA instructions
#pragma once

constexpr int RBK = 32;
constexpr int SLOT_BYTES = 2 * T256 * RBK * 2;      // 32 KiB
constexpr int NSLOT = 4;

__device__ __forceinline__ int swz4(int row) {      // F[(row>>2)&3], F = {0,2,3,1}
    return (0x78 >> (((row >> 2) & 3) * 2)) & 3;     // 0b01_11_10_00 -> entries 0,2,3,1
}

// NT: one DMA piece = 16 rows x 64 B; wave w, round i -> rows i*128 + w*16 .. +16
__device__ __forceinline__ void r_offs_nt(unsigned (&o)[2], long ld, int row0, int nrows, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = i * 128 + wave * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ swz4(row);
        const int rr = min(row, nrows - 1 - row0);
        o[i] = (unsigned)(((long)rr * ld + chunk * 8) * 2);
    }
}
__device__ __forceinline__ void r_stage_nt(char* img, const bf16_t* base_k, const unsigned (&o)[2], int wave) {
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16((const char*)base_k + o[i], img + (i * 128 + wave * 16) * 64);
}
__device__ __forceinline__ void r_stage_nt_tail(char* img, const bf16_t* base_k, const unsigned (&o)[2], int krem, int wave,
                                                int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = i * 128 + wave * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ swz4(row);
        const char* s = (chunk * 8 < krem) ? (const char*)base_k + o[i] : (const char*)&g_zero16;
        glds16(s, img + (i * 128 + wave * 16) * 64);
    }
}
__device__ __forceinline__ bf16x8 r_frag_nt(const char* img, int row, int g) {
    return *reinterpret_cast<const bf16x8*>(img + row * 64 + ((g ^ swz4(row)) << 4));
}

// TN: image [32 k][256 cols]; one DMA piece = 2 k-rows x 512 B; wave w, round i -> k-rows 2*(i*8+w) ..+2
__device__ __forceinline__ void r_offs_tn(unsigned (&o)[2], long ld, int col0, int ncols, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = 2 * (i * 8 + wave) + (lane >> 5);
        const int chunk = (lane & 31) ^ (2 * (k & 7));
        const int cc = min(chunk * 8, ncols - 8 - col0);
        o[i] = (unsigned)(((long)k * ld + cc) * 2);
    }
}
__device__ __forceinline__ void r_stage_tn(char* img, const bf16_t* base_k, const unsigned (&o)[2], int wave) {
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16((const char*)base_k + o[i], img + 2 * (i * 8 + wave) * 512);
}
__device__ __forceinline__ void r_stage_tn_tail(char* img, const bf16_t* base_k, const unsigned (&o)[2], int krem, int wave,
                                                int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = 2 * (i * 8 + wave) + (lane >> 5);
        const char* s = (k < krem) ? (const char*)base_k + o[i] : (const char*)&g_zero16;
        glds16(s, img + 2 * (i * 8 + wave) * 512);
    }
}

template <bool TN, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm256r_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[NSLOT * SLOT_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int wm = wave >> 2, wn = wave & 3;

    const int tiles_n = (p.N + T256 - 1) / T256, tiles_m = (p.M + T256 - 1) / T256;
    const int nwg = tiles_n * tiles_m;
    int id = blockIdx.x;
    {
        const int xcd = id & 7, q = nwg >> 3, rem = nwg & 7;
        id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (id >> 3);
    }
    int tm, tn;
    {
        const int gm = p.gm > 0 ? p.gm : 1;
        const int per_group = gm * tiles_n;
        const int grp = id / per_group, idl = id % per_group;
        const int rows_here = min(gm, tiles_m - grp * gm);
        tm = grp * gm + idl % rows_here;
        tn = idl / rows_here;
    }
    const int bm0 = tm * T256, bn0 = tn * T256;
    const int k_begin = blockIdx.z * p.kper;
    const int k_end = min(p.K, k_begin + p.kper);
    const int nsteps = (k_end - k_begin + RBK - 1) / RBK;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    unsigned oa[2], ob[2];
    if (TN) { r_offs_tn(oa, p.lda, bm0, p.M, wave, lane); r_offs_tn(ob, p.ldb, bn0, p.N, wave, lane); }
    else { r_offs_nt(oa, p.lda, bm0, p.M, wave, lane); r_offs_nt(ob, p.ldb, bn0, p.N, wave, lane); }
    const bf16_t* a_tile = TN ? A + bm0 : A + (long)bm0 * p.lda;
    const bf16_t* b_tile = TN ? B + bn0 : B + (long)bn0 * p.ldb;

    auto stage = [&](int s) {
        char* a_img = smem + (s & (NSLOT - 1)) * SLOT_BYTES;
        char* b_img = a_img + SLOT_BYTES / 2;
        const int k0 = k_begin + s * RBK;
        const int krem = k_end - k0;
        const bf16_t* ab = TN ? a_tile + (long)k0 * p.lda : a_tile + k0;
        const bf16_t* bb = TN ? b_tile + (long)k0 * p.ldb : b_tile + k0;
        if (krem >= RBK) {
            if (TN) { r_stage_tn(a_img, ab, oa, wave); r_stage_tn(b_img, bb, ob, wave); }
            else { r_stage_nt(a_img, ab, oa, wave); r_stage_nt(b_img, bb, ob, wave); }
        } else {
            if (TN) { r_stage_tn_tail(a_img, ab, oa, krem, wave, lane); r_stage_tn_tail(b_img, bb, ob, krem, wave, lane); }
            else { r_stage_nt_tail(a_img, ab, oa, krem, wave, lane); r_stage_nt_tail(b_img, bb, ob, krem, wave, lane); }
        }
    };

    // prologue: three steps in flight (4 DMA instructions per thread and step)
    stage(0);
    if (nsteps > 1) stage(1);
    if (nsteps > 2) stage(2);

    for (int s = 0; s < nsteps; ++s) {
        // step s has landed when at most the DMAs of the (up to two) younger steps are outstanding
        const int younger = min(2, nsteps - 1 - s);
        if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // + every wave is done reading slot (s-1)&3 == (s+3)&3
        asm volatile("" ::: "memory");     // compiler-only fence: no LDS read may be hoisted above the barrier
        if (s + 3 < nsteps) stage(s + 3);
        const char* a_img = smem + (s & (NSLOT - 1)) * SLOT_BYTES;
        const char* b_img = a_img + SLOT_BYTES / 2;
        bf16x8 bf[4], af[8];
        if (DBG & 4) {
#pragma unroll
            for (int t = 0; t < 4; ++t) bf[t] = __builtin_bit_cast(bf16x8, make_uint4(s + t, lane, t, 1));
#pragma unroll
            for (int t = 0; t < 8; ++t) af[t] = __builtin_bit_cast(bf16x8, make_uint4(s - t, lane, t, 2));
        } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            bf[t] = TN ? frag_tn(b_img, 8 * g, wn * 64 + t * 16, r) : r_frag_nt(b_img, wn * 64 + t * 16 + r, g);
#pragma unroll
        for (int t = 0; t < 8; ++t)
            af[t] = TN ? frag_tn(a_img, 8 * g, wm * 128 + t * 16, r) : r_frag_nt(a_img, wm * 128 + t * 16 + r, g);
        }
        if (DBG & 2) {
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(bf[t]));
#pragma unroll
            for (int t = 0; t < 8; ++t) asm volatile("" ::"v"(af[t]));
        } else {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[nt], af[mt], acc[mt][nt], 0, 0, 0);
        }
    }

    // ---- epilogue (identical to gemm256_kernel): 32-row slabs transposed through LDS
    __syncthreads();
    constexpr int ELD = 68;
    float* slab = reinterpret_cast<float*>(smem) + wave * (32 * ELD);
    const float alpha = p.alpha;
    const float* bias = (p.bias && blockIdx.z == 0) ? p.bias : nullptr;
    const int colc = (lane & 7) * 8;
    const int n = bn0 + wn * 64 + colc;
    const bool n_ok = n < p.N;
    float bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (bias && n_ok) load8(bias + n, bv);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<f32x4*>(slab + (h * 16 + r) * ELD + nt * 16 + 4 * g) = acc[2 * pass + h][nt];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3);
            const int m = bm0 + wm * 128 + pass * 32 + row;
            if (m < p.M && n_ok) {
                float v[8];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + row * ELD + colc);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + row * ELD + colc + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = lo[j] * alpha + bv[j]; v[4 + j] = hi[j] * alpha + bv[4 + j]; }
                const long off = (long)m * p.ldc + n;
                if (p.atomic_f32) {
                    float* c = (float*)p.C + off;
#pragma unroll
                    for (int j = 0; j < 8; ++j) atomicAdd(c + j, v[j]);
                } else if (p.epi == EPI_GELU_FWD) {
                    float gv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) gv[j] = gelu_f(v[j]);
                    store8((bf16_t*)p.C + off, v);
                    store8((bf16_t*)p.C2 + off, gv);
                } else {
                    if (p.epi == EPI_GELU_BWD) {
                        float uv[8];
                        load8((const bf16_t*)p.C2 + off, uv);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(uv[j]);
                    }
                    if (p.residual) {
                        float rv[8];
                        load8((const bf16_t*)p.residual + (long)m * p.ldr + n, rv);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += rv[j];
                    }
                    if (p.out_f32) store8((float*)p.C + (long)blockIdx.z * p.slab + off, v);
                    else store8((bf16_t*)p.C + off, v);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
