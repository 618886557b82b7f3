// dma_issue_probe: what does it cost ONE wavefront to issue an LDS-DMA piece (1 KiB = 64 lanes x 16 B), by instruction form?
// Stand-alone diagnostic (hipcc --offload-arch=gfx950 tools/dma_issue_probe.hip -o tools/dma_issue_probe.bin; run on the GPU box).
// Background (round 4): slot stamps of gemm256q showed ~100 cycles per piece per issuing wavefront, the same whether one or
// four wavefronts of the CU issue and with nothing else running -- a per-wavefront cost, not the CU's address path.
// Every workgroup (one per CU) has W wavefronts that each issue NP pieces back to back from an L2-resident 256 KiB source
// (8 rows x 128 B per piece, row stride LD), then wait for them; s_memtime around the issue loop and around issue + wait.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NP = 32;          // pieces per wavefront and round
constexpr int LD = 1664;        // bytes per source row (832 bf16: the model's padded 728)

template <int FORM>
__global__ __launch_bounds__(512) void probe(const char* src, unsigned long long* out, int rounds) {
    __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)src);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)src >> 32));
    const char* usrc = (const char*)(((unsigned long long)hi << 32) | lo);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)usrc, 0, 1 << 20, 0x00020000u);
    const unsigned voff = (unsigned)((lane >> 3) * LD + (lane & 7) * 16);
    const unsigned lds0 = (unsigned)(__SIZE_TYPE__)(lds_void*)smem + wave * (128 * 1024 / 8);
    unsigned long long t_issue = 0, t_all = 0;
    for (int r = 0; r < rounds; ++r) {
        __builtin_amdgcn_s_barrier();
        unsigned long long t0, t1, t2;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const unsigned dst = lds0 + (i & 15) * 1024;
            const int soff = ((i * 8 + wave * 64 + r * 16) & 127) * LD;            // scalar: which 8 rows
            if constexpr (FORM == 0) {          // what gemm256q issues: M0 write, s_nop 4, buffer_load offen + scalar offset
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(dst), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
            } else if constexpr (FORM == 1) {   // the same with s_nop 0
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(dst), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
            } else if constexpr (FORM == 2) {   // global_load_lds, scalar base + 32-bit lane offset
                const char* base = usrc + soff;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                             :: "s"(dst), "v"(voff), "s"(base) : "memory", "m0");
            } else if constexpr (FORM == 3) {   // global_load_lds, 64-bit lane address
                const char* a = usrc + soff + voff;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                             :: "s"(dst), "v"(a) : "memory", "m0");
            } else if constexpr (FORM == 4) {   // the compiler's builtin
                __builtin_amdgcn_global_load_lds((glb_void*)(usrc + soff + voff), (lds_void*)(smem + wave * (128 * 1024 / 8) + (i & 15) * 1024), 16, 0, 0);
            } else if constexpr (FORM == 5) {   // buffer form, M0 written ONCE per 4 pieces, pieces told apart by the instruction offset
                if ((i & 3) == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dst) : "memory", "m0");
                // inst offset advances BOTH the memory address and the LDS address by the same amount: 1024 per piece
                if ((i & 3) == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(rs), "s"(soff) : "memory");
                if ((i & 3) == 1) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds" :: "v"(voff), "s"(rs), "s"(soff) : "memory");
                if ((i & 3) == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:2048 lds" :: "v"(voff), "s"(rs), "s"(soff) : "memory");
                if ((i & 3) == 3) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:3072 lds" :: "v"(voff), "s"(rs), "s"(soff) : "memory");
            } else if constexpr (FORM == 6) {   // plain register loads for comparison (no LDS): 16 B per lane into a VGPR quad
                unsigned __attribute__((ext_vector_type(4))) v;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff) : "memory");
                asm volatile("" :: "v"(v));
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) :: "memory");
        if (r > 0) { t_issue += t1 - t0; t_all += t2 - t0; }
    }
    if (lane == 0) {
        out[(blockIdx.x * nw + wave) * 2] = t_issue;
        out[(blockIdx.x * nw + wave) * 2 + 1] = t_all;
    }
    if (out[0] == 0x1234567 && smem[threadIdx.x]) out[1] = 1;       // keep the LDS image alive
}

template <int FORM>
static void run(const char* name, const char* src, unsigned long long* out, int waves) {
    const int rounds = 41, grid = 256;
    CK(hipMemset(out, 0, grid * 8 * 2 * 8));
    hipLaunchKernelGGL((probe<FORM>), dim3(grid), dim3(64 * waves), 0, 0, src, out, rounds);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(grid * waves * 2);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> a, b;
    for (int i = 0; i < grid * waves; ++i) { a.push_back(h[2 * i] / double(rounds - 1) / NP); b.push_back(h[2 * i + 1] / double(rounds - 1) / NP); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-58s %d wave(s)/CU: issue %6.1f cycles per piece (median; min %5.1f), issue+landed %6.1f per piece -> %5.1f B/clk/CU\n",
           name, waves, a[a.size() / 2], a[0], b[b.size() / 2], 1024.0 * waves / b[b.size() / 2]);
    fflush(stdout);
}

int main() {
    char* src; unsigned long long* out;
    CK(hipMalloc(&src, 2 << 20)); CK(hipMalloc(&out, 256 * 8 * 2 * 8));
    CK(hipMemset(src, 0x3c, 2 << 20));
    for (int waves : {1, 4, 8}) {
        run<0>("buffer_load..lds, M0 + s_nop 4 (gemm256q)", src, out, waves);
        run<1>("buffer_load..lds, M0 + s_nop 0", src, out, waves);
        run<2>("global_load_lds saddr + voffset", src, out, waves);
        run<3>("global_load_lds 64-bit lane address", src, out, waves);
        run<4>("__builtin_amdgcn_global_load_lds", src, out, waves);
        run<5>("buffer_load..lds, M0 once per 4 pieces (inst offset)", src, out, waves);
        run<6>("buffer_load to VGPRs (no LDS)", src, out, waves);
    }
    return 0;
}
