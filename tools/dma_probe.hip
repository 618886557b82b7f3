// dma_probe: how fast can ONE workgroup per CU stream GEMM operand panels into LDS with global_load_lds?
// Stand-alone diagnostic (hipcc --offload-arch=gfx950 tools/dma_probe.hip -o tools/dma_probe.bin), run on the
// GPU box.  It walks the same tiles in the same order as gemm256p_kernel (256x256 tiles of an [M][K] x [N][K]
// NT GEMM, XCD remap, groups of gm row panels) and issues the same LDS-DMA traffic through a ring of LDS
// slots with counted vmcnt waits and one barrier per operand stage -- but never reads LDS and runs no MFMA.
// Variants: bytes per row per stage (64 = the 32-deep step of gemm256r/p, 128, 256), stages in flight, row
// stride (1456 B = K 728, or padded to 1536), private (unshared) panels, optional MFMA filler per stage.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct Args {
    const char* A; const char* B;
    long lda, ldb;          // bytes
    int M, N, Kb;           // Kb = bytes of one row that are streamed
    int gm, priv, mfma;     // mfma = MFMAs per wave and operand stage (filler)
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int ROWB, int DEPTH>
__global__ __launch_bounds__(512) void probe(Args p, float* sink) {
    constexpr int STAGE = 256 * ROWB;               // bytes of one operand stage
    constexpr int NSLOT = (128 * 1024) / STAGE;
    constexpr int IPS = ROWB / 32;                  // DMA instructions per wave and stage
    constexpr int RPI = 1024 / ROWB;                // rows per wave-instruction
    constexpr int CPR = ROWB / 16;                  // 16-byte chunks per row
    static_assert(DEPTH < NSLOT || NSLOT == DEPTH, "ring too shallow");
    __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_n = (p.N + 255) / 256, tiles_m = (p.M + 255) / 256;
    const int nwg = tiles_n * tiles_m;
    const int G = gridDim.x;
    const int my_tiles = (nwg - (int)blockIdx.x + G - 1) / G;
    const int ksteps = p.Kb / ROWB;
    const int total = my_tiles * ksteps * 2;

    auto origin = [&](int i, int& bm0, int& bn0) {
        int id = (int)blockIdx.x + i * G;
        if (p.priv) { bm0 = (id % tiles_m) * 256; bn0 = 0; return; }
        const int xcd = id & 7, q = nwg >> 3, rem = nwg & 7;
        id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (id >> 3);
        const int gm = p.gm;
        const int per_group = gm * tiles_n;
        const int grp = id / per_group, idl = id % per_group;
        const int rows_here = min(gm, tiles_m - grp * gm);
        bm0 = (grp * gm + idl % rows_here) * 256;
        bn0 = (idl / rows_here) * 256;
    };

    int P = 0, p_t = 0, p_s = 0, p_op = 0, bm0 = 0, bn0 = 0;
    if (my_tiles > 0) origin(0, bm0, bn0);
    auto issue = [&]() {
        if (P >= total) return;
        char* img = smem + (P % NSLOT) * STAGE;
        const char* base;
        long ld; int nrows, r0;
        if (p_op == 0 || p.priv) { base = p.A; ld = p.lda; nrows = p.M; r0 = bm0; }
        else { base = p.B; ld = p.ldb; nrows = p.N; r0 = bn0; }
#pragma unroll
        for (int i = 0; i < IPS; ++i) {
            const int row_in_tile = (i * 8 + wave) * RPI + lane / CPR;
            const int row = min(r0 + row_in_tile, nrows - 1);
            const char* src = base + (long)row * ld + (long)p_s * ROWB + (lane % CPR) * 16;
            __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(img + (i * 8 + wave) * 1024), 16, 0, 0);
        }
        ++P;
        if (++p_op == 2) {
            p_op = 0;
            if (++p_s == ksteps) { p_s = 0; if (++p_t < my_tiles) origin(p_t, bm0, bn0); }
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue();

    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa = __builtin_bit_cast(bf16x8, make_uint4(lane, 1, 2, 3)), fb = __builtin_bit_cast(bf16x8, make_uint4(3, lane, 1, 0));

    for (int q = 0; q < total; ++q) {
        wait_vm<IPS*(DEPTH - 1)>();                // stage q has landed (the tail over-waits nothing: fewer are outstanding)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue();
        for (int m = 0; m < p.mfma; m += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[j], 0, 0, 0);
        }
    }
    wait_vm<0>();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 12345.678f) sink[0] = s + smem[threadIdx.x];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int ROWB, int DEPTH>
static void run(const char* name, Args a, int grid, float* sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<ROWB, DEPTH>), dim3(grid), dim3(512), 0, 0, a, sink);
    CK(hipDeviceSynchronize());
    const int reps = 5;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<ROWB, DEPTH>), dim3(grid), dim3(512), 0, 0, a, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const long tiles = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    const double bytes = (double)tiles * (a.Kb / ROWB) * 2.0 * 256 * ROWB;
    printf("%-44s rowB %3d depth %d (%3d KiB in flight) grid %4d : %8.1f us  %6.2f TB/s  %6.1f GB/s/CU\n", name, ROWB, DEPTH,
           DEPTH * 256 * ROWB / 1024, grid, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / 256);
    fflush(stdout);
}

int main() {
    const int M = 56736, N = 2912;
    const long ldmax = 1536;
    char *A, *B; float* sink;
    CK(hipMalloc(&A, (size_t)M * ldmax)); CK(hipMalloc(&B, (size_t)N * ldmax)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(A, 0x3c, (size_t)M * ldmax)); CK(hipMemset(B, 0x3c, (size_t)N * ldmax));
    for (int pass = 0; pass < 2; ++pass) {
        const long ld = pass == 0 ? 1456 : 1536;
        Args a{A, B, ld, ld, M, N, 1408, 4, 0, 0};
        printf("---- row stride %ld B, FF1 shape (M %d, N %d), shared panels, gm 4\n", ld, M, N);
        run<64, 3>("64-B rows (gemm256r/p today)", a, 256, sink);
        run<64, 6>("64-B rows", a, 256, sink);
        run<64, 7>("64-B rows", a, 256, sink);
        run<128, 2>("128-B rows", a, 256, sink);
        run<128, 3>("128-B rows", a, 256, sink);
        run<256, 1>("256-B rows", a, 256, sink);
        Args b = a; b.gm = 8;
        run<64, 6>("64-B rows gm 8", b, 256, sink);
        run<128, 3>("128-B rows gm 8", b, 256, sink);
        b.gm = 2;
        run<128, 3>("128-B rows gm 2", b, 256, sink);
        Args c = a; c.priv = 1;
        run<64, 6>("64-B rows, private panels (A only)", c, 256, sink);
        run<128, 3>("128-B rows, private panels (A only)", c, 256, sink);
        Args d = a; d.mfma = 16;     // 16 MFMA per wave and operand stage = what the GEMM runs per 16 KiB (64-B rows)
        run<64, 3>("64-B rows + 16 MFMA/stage", d, 256, sink);
        run<64, 6>("64-B rows + 16 MFMA/stage", d, 256, sink);
        d.mfma = 32;
        run<128, 3>("128-B rows + 32 MFMA/stage", d, 256, sink);
        run<128, 2>("128-B rows + 32 MFMA/stage", d, 256, sink);
    }
    // narrow N (to_out / FF2: N = 728): 3 column tiles
    {
        Args a{A, B, 1456, 1456, M, 728, 1408, 4, 0, 0};
        printf("---- N = 728\n");
        run<64, 3>("64-B rows", a, 256, sink);
        run<128, 3>("128-B rows", a, 256, sink);
    }
    return 0;
}
