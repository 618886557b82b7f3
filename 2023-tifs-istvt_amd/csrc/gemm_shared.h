// Definitions shared by the GEMM kernels of gemm.hip: the generic 128x128 kernel and the two 256x256 LDS-DMA kernels
// (gemm256q.h: persistent NT forward / input gradient; gemm256t.h: TN weight gradient).
#pragma once

constexpr int T256 = 256;           // tile edge of the LDS-DMA kernels
constexpr int PSLAB_BYTES = 4096;   // per-wavefront epilogue slab: [16 rows][64 f32], 16-byte chunk c of row r at position c ^ r

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

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
        if (p.gelu_d) {                 // C takes gelu'(u): all the backward needs of u
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = gelu_grad_f(v[j]);
        }
        if (nvec) { store4(c, v); store4(c2, gv); }
        else { for (int j = 0; j < nv; ++j) { c[j] = from_f32<T>(v[j]); c2[j] = from_f32<T>(gv[j]); } }
        return;
    }
    if (p.epi == EPI_GELU_BWD) {
        const T* u = (const T*)p.C2 + off;
        float uv[4] = {0.f, 0.f, 0.f, 0.f};
        if (nvec) load4(u, uv); else { for (int j = 0; j < nv; ++j) uv[j] = to_f32(u[j]); }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= p.gelu_d ? uv[j] : gelu_grad_f(uv[j]);
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


// vmcnt discipline of the LDS-DMA kernels.  vmcnt counts LDS-DMA, loads and stores together and retires them in issue
// order, so `s_waitcnt vmcnt(n)` means "everything but the n youngest operations is done".  All waits are written by
// hand with n = the number of operations GUARANTEED to have been issued after the one that is needed.  The epilogue's
// loads are inline-asm buffer loads, tied to their wait through "+v" operands: they must not be compiler-visible
// loads, because while an LDS-DMA is pending the compiler's own bookkeeping gives up and emits vmcnt(0) before the
// first use of any loaded value, which drains the ring once per tile.  Lanes outside the matrix use the buffer
// instructions' range check (offset >= num_records: loads return 0, stores are dropped) instead of branches, so every
// lane issues the same number of operations.
template <bool NT = false>
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
    u32x4 v;
    if constexpr (NT) {          // read-once data (see the s_nop note below)
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff) : "memory");
        return v;
    }
    // s_nop 4: the compiler does not know this statement is a VMEM instruction, so it does not pad the 5 wait states a
    // VMEM read of an SGPR needs after a VALU wrote it (v_readfirstlane of a descriptor word, v_readlane of a spilled
    // scalar offset): without them the load went out with the PREVIOUS value of the scalar offset (seen on gfx950).
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    return v;
}

// (dma16_lds / dma4_lds, the LDS-DMA in its opaque inline-assembly form, live in common.h: layernorm.hip uses them too)

// STORE-DATA HAZARD (observed on gfx950, not padded by the compiler): a VALU write to the data registers of a
// buffer_store_dwordx4 with an SGPR soffset a few instructions after the store reached memory instead of the store
// data (one dword, lanes 12..15 of every 16).  The epilogues therefore keep the data registers allocated -- tied to an
// asm statement -- until the end of the pass, and pad them with wait states before they can be reused.
