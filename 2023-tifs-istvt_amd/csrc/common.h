// Shared device helpers for the ISTVT gfx950 (CDNA4) kernels.
// Conventions used across csrc/:
//   * storage dtype T is float (parity mode) or bf16_t (throughput mode); all arithmetic,
//     statistics and accumulation are fp32.
//   * a wavefront is 64 lanes; lane = threadIdx.x & 63; MFMA lane groups g = lane >> 4, r = lane & 15.
//   * "fragment" = the 8 consecutive reduction-dim elements k = 8g .. 8g+7 one lane feeds to
//     one logical K=32 MFMA step (bf16: one v_mfma_f32_16x16x32_bf16; f32: eight
//     v_mfma_f32_16x16x4_f32 in which lane group g supplies element 8g+i in step i).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define ISTVT_OK 0
#define ISTVT_ERR_DTYPE (-2)
#define ISTVT_ERR_SHAPE (-3)
#define ISTVT_ERR_LAUNCH (-4)

enum { DT_F32 = 0, DT_BF16 = 1 };

#define WAVE 64

// per-channel statistics accumulators are double[ISTVT_STAT_REPLICAS][2][C] (see stem.hip); producers add into replica
// (workgroup % ISTVT_STAT_REPLICAS)
constexpr int ISTVT_STAT_REPLICAS = 32;
// The hardware deals the workgroup ids of a 1-D grid round-robin to the 8 XCDs (id & 7), each with its own L2:
// neighbours in blockIdx.x never share an L2.  xcd_chunk() renumbers the grid (a bijection on 0..nwg-1) so that the
// workgroups of one XCD hold a CONTIGUOUS eighth of the work list -- tiles that share halo rows / operand panels then
// meet in one L2 instead of each fetching its own copy from HBM.
__device__ __forceinline__ int xcd_chunk(const int id, const int nwg) {
    const int xcd = id & 7, q = nwg >> 3, rem = nwg & 7;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (id >> 3);
}

// weight gradients launched together by istvt_wgrad_group (gemm256t.h)
constexpr int ISTVT_WGRAD_GROUP_MAX = 8;

// Launch-geometry constants (grid caps, band widths, tile heights) were each swept on the MI355X; the value and the
// measurement stand at the use.  The shipped library compiles them in.  A -DISTVT_TUNE build (tools/build_variant.sh)
// reads the named environment variable instead, so a sweep is one build and one process per value.
#ifdef ISTVT_TUNE
#include <cstdlib>
static inline long istvt_tune(const char* name, long dflt) {
    const char* v = getenv(name);
    return v ? atol(v) : dflt;
}
#else
static inline constexpr long istvt_tune(const char*, long dflt) { return dflt; }
#endif

// Fixed-order column reduction (defined in elementwise.hip), the second stage of every per-column sum that used to end in
// float atomics: out_a[c] += sum_{r < rows} ws[(r * nacc + a) * N + c] for a < nacc <= 3, rows summed in index order by a
// single writer per column.  Kernels store one partial row per workgroup into a caller-owned float workspace and this
// launch folds them: two runs give the same bits, and no workgroup ends in a tail of same-address atomics.
int istvt_rows_reduce_add(const float* ws, int rows, int nacc, int N, float* o0, float* o1, float* o2, hipStream_t stream);

static inline int istvt_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ISTVT_OK : -(1000 + (int)e);
}

// Compute units of the CALLING THREAD'S CURRENT DEVICE (the one a launch on the caller's stream runs on), 256 when the
// runtime cannot say.  A device's CU count never changes, so the per-device slots below are write-once caches of an
// immutable value (relaxed atomics: two threads that race store the same number); nothing about the process's state --
// which device is current, how many devices it uses -- is remembered between calls (re-entrant entry points).
#include <atomic>
static inline int istvt_device_cus() {
    constexpr int MAXDEV = 64;
    static std::atomic<int> cache[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 256;
    if (dev < MAXDEV) {
        const int hit = cache[dev].load(std::memory_order_relaxed);
        if (hit > 0) return hit;
    }
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    if (dev < MAXDEV) cache[dev].store(n, std::memory_order_relaxed);
    return n;
}
// Raise a kernel's dynamic-LDS limit above 64 KiB once per (kernel instantiation, device): `done` is that instantiation's
// bitmask of the devices already served (a write-once flag per device, idempotent under a race).
static inline int istvt_raise_lds_limit(std::atomic<unsigned long long>& done, const void* kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (dev < 64 && (done.load(std::memory_order_relaxed) & bit)) return ISTVT_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return ISTVT_ERR_LAUNCH;
    if (dev < 64) done.fetch_or(bit, std::memory_order_relaxed);
    return ISTVT_OK;
}

#define DISPATCH_DTYPE(dtype, ...)                         \
    do {                                                   \
        if ((dtype) == DT_F32) { typedef float T; __VA_ARGS__; }       \
        else if ((dtype) == DT_BF16) { typedef bf16_t T; __VA_ARGS__; } \
        else return ISTVT_ERR_DTYPE;                       \
    } while (0)

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// ---- 8-element chunks (16 B of bf16, 32 B of f32); p must be 16-byte aligned -------------
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
}
// ---- 4-element chunks (8 B of bf16, 16 B of f32) -------------------------------------------
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
    bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
    bf16x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x4*>(p) = a;
}

// ---- wave-level reductions (all 64 lanes get the result) ---------------------------------
// DPP within each 16-lane row (quad_perm x2, row_half_mirror, row_mirror: no LDS crossbar), then
// the four row totals are read through SGPRs (v_readlane) -- 6 ds_bpermute round trips per
// reduction were the critical path of the one-wavefront-per-row LayerNorm kernels.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// the builtin is the INTEGER readlane: passing a float converts its VALUE; move the bits instead
__device__ __forceinline__ float row_total(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);     // row_half_mirror
    v += dpp_mov<0x140>(v);     // row_mirror  -> every lane holds its row's sum
    return (row_total(v, 0) + row_total(v, 16)) + (row_total(v, 32) + row_total(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return fmaxf(fmaxf(row_total(v, 0), row_total(v, 16)), fmaxf(row_total(v, 32), row_total(v, 48)));
}

// ---- MFMA traits: one logical K=32 step on a 16x16 tile ----------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ void mma(f32x4& c, const frag& a, const frag& b) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ frag zero() {
        frag f;
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (bf16_t)0.0f;
        return f;
    }
    static __device__ __forceinline__ void set(frag& f, int i, float v) { f[i] = (bf16_t)v; }
    static __device__ __forceinline__ float get(const frag& f, int i) { return (float)f[i]; }
};
struct f32frag { float v[8]; };
template <> struct Mma<float> {
    typedef f32frag frag;
    static __device__ __forceinline__ void mma(f32x4& c, const frag& a, const frag& b) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[i], b.v[i], c, 0, 0, 0);
    }
    static __device__ __forceinline__ frag zero() {
        frag f;
#pragma unroll
        for (int i = 0; i < 8; ++i) f.v[i] = 0.f;
        return f;
    }
    static __device__ __forceinline__ void set(frag& f, int i, float v) { f.v[i] = v; }
    static __device__ __forceinline__ float get(const frag& f, int i) { return f.v[i]; }
};

// Reductions over the four 16-lane groups of a wavefront (lanes r, r+16, r+32, r+48: the lanes that share an MFMA
// accumulator column): two gfx950 half-swaps (v_permlane16_swap / v_permlane32_swap, vector-ALU latency) instead of two
// ds_bpermute round trips through the LDS (~120 cycles each, on every softmax's critical path).  Swapping a value with
// itself leaves {own half, other half} in the two results, so one op on them is the xor-16 / xor-32 butterfly step.
__device__ __forceinline__ float xgroup_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// sum over the 8 lanes that share lane & 7 (lane ^ 8 by a DPP rotate inside the 16-lane row, ^ 16 / ^ 32 by half swaps):
// no LDS round trip (ds_bpermute) on the way
__device__ __forceinline__ float xsum_lanes_8_16_32(float v) {
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128 /* row_ror:8 */, 0xf, 0xf, true));
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float xgroup_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// fragment from 8 contiguous elements in LDS/global (16-byte aligned)
__device__ __forceinline__ bf16x8 frag_load(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ f32frag frag_load(const float* p) {
    f32frag f;
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
    return f;
}

// Transposed fragment: element i = img[(k0 + i) * ld + col], i.e. the reduction index runs
// over ROWS of a row-major LDS image (ld in elements).  k0 must be a multiple of 4 and, for
// bf16, `col16` (the first of the 16 columns this lane group covers) a multiple of 4.
//   bf16: two ds_read_b64_tr_b16 (each: 4 rows x 16 cols block, lane i of the 16-lane group
//         receives column i of the 4 rows); lane 4q+p supplies row q, cols 4p..4p+3.
//   f32 : eight ds_read_b32.
// k0a / k0b are the first rows of the two 4-row halves (elements 0..3 and 4..7).
__device__ __forceinline__ bf16x8 frag_load_tr(const bf16_t* img, int ld, int k0a, int k0b, int col16, int r) {
    const int q = r >> 2, p = r & 3;
    typedef short4v __attribute__((address_space(3))) * lds_ptr;
    const bf16_t* pa = img + (size_t)(k0a + q) * ld + col16 + 4 * p;
    const bf16_t* pb = img + (size_t)(k0b + q) * ld + col16 + 4 * p;
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pa));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pb));
    typedef short short8v __attribute__((ext_vector_type(8)));
    short8v s;
    s[0] = lo[0]; s[1] = lo[1]; s[2] = lo[2]; s[3] = lo[3]; s[4] = hi[0]; s[5] = hi[1]; s[6] = hi[2]; s[7] = hi[3];
    return __builtin_bit_cast(bf16x8, s);
}
__device__ __forceinline__ f32frag frag_load_tr(const float* img, int ld, int k0a, int k0b, int col16, int r) {
    f32frag f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f.v[i] = img[(size_t)(k0a + i) * ld + col16 + r];
        f.v[4 + i] = img[(size_t)(k0b + i) * ld + col16 + r];
    }
    return f;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---- LDS-DMA ---------------------------------------------------------------------------------
// LDS-DMA as inline assembly: 16 bytes per lane from (descriptor, per-lane offset voff, scalar offset soff) to the LDS
// byte address lds_addr + 16 * lane.  The builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) is a store to LDS as far as
// the compiler knows, so every LDS read that FOLLOWS it in program order gets an s_waitcnt vmcnt(0) in front: a K loop can
// then only issue its DMA after the fragment reads of the same slot.  This form is opaque: ordering against the LDS
// reads is the kernel's own barrier / vmcnt protocol.  s_nop 4: one wait state between the write of M0 and the DMA,
// five between a VALU write of an SGPR operand (v_readfirstlane) and the VMEM instruction that reads it.
__device__ __forceinline__ void dma16_lds(__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, unsigned voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}

// the same, 4 bytes per lane (lds_addr + 4 * lane)
__device__ __forceinline__ void dma4_lds(__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, unsigned voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}

