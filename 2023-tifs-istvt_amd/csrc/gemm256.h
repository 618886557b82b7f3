// 256x256-tile bf16 MFMA GEMM for gfx950 with direct global->LDS staging (global_load_lds,
// 16 B/lane) and double-buffered LDS: one barrier per 64-deep K step, the next tile's DMA in
// flight under the current tile's 64 MFMAs per wavefront.  Included by gemm.hip (shares GemmArgs
// and the epilogue with the generic kernel).
//
//   workgroup = 512 threads = 8 wavefronts as 2(M) x 4(N); wavefront tile 128x64 = 8x4 MFMA
//   16x16x32 tiles (128 accumulator VGPRs); LDS = 2 stages x (A 32 KiB + B 32 KiB) = 128 KiB,
//   one workgroup per CU.  Per K step a CU moves 64 KiB into LDS for 2 x 256x256x64 flops:
//   128 flop/B against the ~56 B/clk/CU the L2 delivers, so the step is MFMA-bound.
//
// LDS images are written lane-linearly by the DMA (1 KiB per wavefront instruction), so bank
// conflicts are avoided by permuting the SOURCE address and applying the same XOR on the read:
//   NT (k-contiguous operands, image [256 rows][64 k], 128 B rows, 8 chunks of 16 B):
//        chunk' = chunk ^ ((row >> 1) & 7)      -> ds_read_b128 fragment reads conflict-free
//   TN (row-contiguous operands, image [64 k][256 rows], 512 B rows, 32 chunks):
//        chunk' = chunk ^ (2 * (k & 7))         -> ds_read_b64_tr_b16 transposed reads conflict-free
// Out-of-range rows / K tails (K = 728 = 11*64 + 24) are filled by pointing the lane's source at
// a 16-byte zero constant, never by reading out of bounds.
#pragma once

__device__ uint4 g_zero16 = {0u, 0u, 0u, 0u};   // deliberately non-const: keeps it in the GLOBAL address space so the source select is one v_cndmask

constexpr int T256 = 256;           // tile edge
constexpr int BK256 = 64;           // K step
constexpr int STAGE_BYTES = 2 * T256 * BK256 * 2;       // A + B images of one stage (64 KiB)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)lds_wave_base, 16, 0, 0);
}

// ---- one epilogue for every GEMM kernel: lane owns C[m][n .. n+3] -------------------------
template <typename T>
__device__ __forceinline__ void gemm_epilogue4(const GemmArgs& p, int m, int n, const f32x4& a, bool nvec) {
    float v[4] = {a[0] * p.alpha, a[1] * p.alpha, a[2] * p.alpha, a[3] * p.alpha};
    const int nv = min(4, p.N - n);
    if (p.bias && blockIdx.z == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < nv) v[j] += p.bias[n + j];
    }
    if (p.atomic_f32) {
        float* c = (float*)p.C + (long)m * p.ldc + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < nv) atomicAdd(c + j, v[j]);
        return;
    }
    const long off = (long)m * p.ldc + n;
    if (p.epi == EPI_GELU_FWD) {
        float gv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) gv[j] = gelu_f(v[j]);
        T* c = (T*)p.C + off;
        T* c2 = (T*)p.C2 + off;
        if (nvec) { store4(c, v); store4(c2, gv); }
        else { for (int j = 0; j < nv; ++j) { c[j] = from_f32<T>(v[j]); c2[j] = from_f32<T>(gv[j]); } }
        return;
    }
    if (p.epi == EPI_GELU_BWD) {
        const T* u = (const T*)p.C2 + off;
        float uv[4] = {0.f, 0.f, 0.f, 0.f};
        if (nvec) load4(u, uv); else { for (int j = 0; j < nv; ++j) uv[j] = to_f32(u[j]); }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= gelu_grad_f(uv[j]);
    }
    if (p.residual) {
        const T* rp = (const T*)p.residual + (long)m * p.ldr + n;
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        if (nvec && (p.ldr % 4 == 0)) load4(rp, rv); else { for (int j = 0; j < nv; ++j) rv[j] = to_f32(rp[j]); }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += rv[j];
    }
    if (p.out_f32) {
        float* c = (float*)p.C + (long)blockIdx.z * p.slab + off;
        if (nvec) store4(c, v); else { for (int j = 0; j < nv; ++j) c[j] = v[j]; }
    } else {
        T* c = (T*)p.C + off;
        if (nvec) store4(c, v); else { for (int j = 0; j < nv; ++j) c[j] = from_f32<T>(v[j]); }
    }
}

// ---- staging ---------------------------------------------------------------------------------
// Steady state: every lane's source is  uniform tile base (SGPRs, advanced by the K loop)  +  a
// loop-invariant 32-bit byte offset (one VGPR per DMA), so the K loop issues its 8 DMAs without
// touching a vector register (rewriting an address VGPR of an in-flight LDS-DMA costs a
// vmcnt(0)).  Rows / columns outside the matrix are CLAMPED to the last valid one: they only
// feed output rows / columns that are never stored.  Only the reduction-dim tail needs zeros;
// that single partial K step takes the *_tail path, which points out-of-range lanes at g_zero16.
//
// NT image: rows x 64 k.  One DMA piece = 8 rows x 128 B; wave w, round i -> rows i*64 + w*8 ..+8.
struct StageOffs { unsigned a[4], b[4]; };

__device__ __forceinline__ void offs_nt(unsigned (&o)[4], long ld, int row0, int nrows, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 64 + wave * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);          // data chunk that lives at LDS position lane&7
        const int rr = min(row, nrows - 1 - row0);                // clamp inside the matrix (tile-relative)
        o[i] = (unsigned)(((long)rr * ld + chunk * 8) * 2);
    }
}
__device__ __forceinline__ void stage_nt(char* img, const bf16_t* tile_base_k, const unsigned (&o)[4], int wave) {
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16((const char*)tile_base_k + o[i], img + (i * 64 + wave * 8) * 128);
}
__device__ __forceinline__ void stage_nt_tail(char* img, const bf16_t* tile_base_k, const unsigned (&o)[4], int krem,
                                              int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 64 + wave * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const char* s = (chunk * 8 < krem) ? (const char*)tile_base_k + o[i] : (const char*)&g_zero16;
        glds16(s, img + (i * 64 + wave * 8) * 128);
    }
}
// TN image: 64 k x 256 rows.  One DMA piece = 2 k-rows x 512 B; wave w, round i -> k-rows 2*(i*8+w) ..+2.
__device__ __forceinline__ void offs_tn(unsigned (&o)[4], long ld, int col0, int ncols, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = 2 * (i * 8 + wave) + (lane >> 5);
        const int chunk = (lane & 31) ^ (2 * (k & 7));
        const int cc = min(chunk * 8, ncols - 8 - col0);          // clamp the 8-column chunk inside the matrix
        o[i] = (unsigned)(((long)k * ld + cc) * 2);
    }
}
__device__ __forceinline__ void stage_tn(char* img, const bf16_t* tile_base_k, const unsigned (&o)[4], int wave) {
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16((const char*)tile_base_k + o[i], img + 2 * (i * 8 + wave) * 512);
}
__device__ __forceinline__ void stage_tn_tail(char* img, const bf16_t* tile_base_k, const unsigned (&o)[4], int krem,
                                              int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = 2 * (i * 8 + wave) + (lane >> 5);
        const char* s = (k < krem) ? (const char*)tile_base_k + o[i] : (const char*)&g_zero16;
        glds16(s, img + 2 * (i * 8 + wave) * 512);
    }
}

__device__ __forceinline__ bf16x8 frag_nt(const char* img, int row, int kchunk) {
    return *reinterpret_cast<const bf16x8*>(img + row * 128 + ((kchunk ^ ((row >> 1) & 7)) << 4));
}
// transposed fragment from a TN image: element i = img[k0 + i (i<4) | k0 + 4 + (i-4)][col16 + r]
__device__ __forceinline__ bf16x8 frag_tn(const char* img, int k0, int col16, int r) {
    const int q = r >> 2, p = r & 3;
    typedef short4v __attribute__((address_space(3))) * lds_ptr;
    const int ka = k0 + q, kb = k0 + 4 + q;
    const int chunk = (col16 >> 3) + (p >> 1);
    const char* pa = img + ka * 512 + ((chunk ^ (2 * (ka & 7))) << 4) + ((p & 1) << 3);
    const char* pb = img + kb * 512 + ((chunk ^ (2 * (kb & 7))) << 4) + ((p & 1) << 3);
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pa));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pb));
    typedef short short8v __attribute__((ext_vector_type(8)));
    short8v s;
    s[0] = lo[0]; s[1] = lo[1]; s[2] = lo[2]; s[3] = lo[3]; s[4] = hi[0]; s[5] = hi[1]; s[6] = hi[2]; s[7] = hi[3];
    return __builtin_bit_cast(bf16x8, s);
}

// ---- the kernel --------------------------------------------------------------------------------
// TN = false: A [M][K], B [N][K] (k-contiguous).   TN = true: A [K][M], B [K][N] (row-contiguous).
// DBG (diagnostic builds only, selected with ISTVT_GEMM_DBG): 1 = no DMA inside the K loop,
// 2 = no MFMA, 4 = no LDS fragment reads.  DBG = 0 is the product kernel.
template <bool TN, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int wm = wave >> 2, wn = wave & 3;                  // 2 x 4 wavefronts, 128 x 64 each

    const int tiles_n = (p.N + T256 - 1) / T256, tiles_m = (p.M + T256 - 1) / T256;
    const int nwg = tiles_n * tiles_m;
    int id = blockIdx.x;
    int zsplit = blockIdx.z;
    if (p.flat_splits > 0) {
        // split-K weight gradients: every tile of one reduction slice reads the same rows of dy and
        // x, so a whole slice is given to ONE XCD (slice s -> XCD s % 8) and its tiles run together:
        // the panels are fetched from HBM once per slice instead of once per tile row/column
        // (PMC: 3x the algorithmic bytes with slices interleaved over XCDs).
        const int xcd = id & 7, j = id >> 3;
        zsplit = xcd + 8 * (j / nwg);
        id = j % nwg;
        if (zsplit >= p.flat_splits) return;
    } else {   // XCD-aware order (bijective): workgroups that share an XCD walk consecutive tiles
        const int xcd = id & 7, q = nwg >> 3, rem = nwg & 7;
        id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (id >> 3);
    }
    // tiles are walked in groups of `gm` row-panels, row-fastest inside a group: the ~32 workgroups an
    // XCD runs concurrently then touch gm A-panels x 32/gm B-panels instead of 32/tiles_n x tiles_n
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
    const int k_begin = zsplit * p.kper;
    const int k_end = min(p.K, k_begin + p.kper);
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    StageOffs so;
    if (TN) { offs_tn(so.a, p.lda, bm0, p.M, wave, lane); offs_tn(so.b, p.ldb, bn0, p.N, wave, lane); }
    else { offs_nt(so.a, p.lda, bm0, p.M, wave, lane); offs_nt(so.b, p.ldb, bn0, p.N, wave, lane); }
    // uniform tile bases; the K loop advances them (NT: along the row, TN: by whole rows)
    const bf16_t* a_tile = TN ? A + bm0 : A + (long)bm0 * p.lda;
    const bf16_t* b_tile = TN ? B + bn0 : B + (long)bn0 * p.ldb;
    auto stage = [&](int buf, int k0) {
        char* a_img = smem + buf * STAGE_BYTES;
        char* b_img = a_img + STAGE_BYTES / 2;
        const int krem = k_end - k0;
        const bf16_t* ab = TN ? a_tile + (long)k0 * p.lda : a_tile + k0;
        const bf16_t* bb = TN ? b_tile + (long)k0 * p.ldb : b_tile + k0;
        if (krem >= BK256) {
            if (TN) { stage_tn(a_img, ab, so.a, wave); stage_tn(b_img, bb, so.b, wave); }
            else { stage_nt(a_img, ab, so.a, wave); stage_nt(b_img, bb, so.b, wave); }
        } else {
            if (TN) { stage_tn_tail(a_img, ab, so.a, krem, wave, lane); stage_tn_tail(b_img, bb, so.b, krem, wave, lane); }
            else { stage_nt_tail(a_img, ab, so.a, krem, wave, lane); stage_nt_tail(b_img, bb, so.b, krem, wave, lane); }
        }
    };

    stage(0, k_begin);
    int buf = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK256, buf ^= 1) {
        __syncthreads();        // tile k0 has landed (vmcnt(0) is part of the barrier); the other stage is free
        if (!(DBG & 1) && k0 + BK256 < k_end) stage(buf ^ 1, k0 + BK256);
        const char* a_img = smem + buf * STAGE_BYTES;
        const char* b_img = a_img + STAGE_BYTES / 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bf[4], af[8];
            if (DBG & 4) {
#pragma unroll
                for (int t = 0; t < 4; ++t) bf[t] = __builtin_bit_cast(bf16x8, make_uint4(k0 + t, lane, t, ks));
#pragma unroll
                for (int t = 0; t < 8; ++t) af[t] = __builtin_bit_cast(bf16x8, make_uint4(k0 - t, lane, t, ks));
            } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                bf[t] = TN ? frag_tn(b_img, ks * 32 + 8 * g, wn * 64 + t * 16, r)
                           : frag_nt(b_img, wn * 64 + t * 16 + r, ks * 4 + g);
#pragma unroll
            for (int t = 0; t < 8; ++t)
                af[t] = TN ? frag_tn(a_img, ks * 32 + 8 * g, wm * 128 + t * 16, r)
                           : frag_nt(a_img, wm * 128 + t * 16 + r, ks * 4 + g);
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
    }

    // ---- epilogue through LDS: the accumulator layout (lane = one row, 4 columns) would store
    // 8-byte pieces of 16 different rows per instruction; transposing 32-row slabs through the
    // (now idle) LDS gives every lane 8 consecutive columns of one row: 16-byte accesses, 8 lanes
    // = one 128-byte row segment, for the output, the residual, the GELU operand and the atomics.
    __syncthreads();
    constexpr int ELD = 68;                                     // floats per slab row (64 + pad)
    float* slab = reinterpret_cast<float*>(smem) + wave * (32 * ELD);
    const float alpha = p.alpha;
    const float* bias = (p.bias && zsplit == 0) ? p.bias : nullptr;
    const int colc = (lane & 7) * 8;
    const int n = bn0 + wn * 64 + colc;
    const bool n_ok = n < p.N;                                  // N % 8 == 0: a chunk is all in or all out
    float bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (bias && n_ok) load8(bias + n, bv);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<f32x4*>(slab + (h * 16 + r) * ELD + nt * 16 + 4 * g) = acc[2 * pass + h][nt];
        // the slab is private to this wavefront and a wavefront's DS operations execute in order, so
        // a compiler-level fence is all the write->read (and the read->next-write) hand-off needs:
        // the 8 wavefronts drift apart and one's stores overlap another's LDS transposes
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
                    if (p.out_f32) store8((float*)p.C + (long)zsplit * p.slab + off, v);
                    else store8((bf16_t*)p.C + off, v);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
