// rc_probe: does gfx950 include the SCALAR offset of a raw buffer access in the range check?  (It does: a lane whose
// voffset is in range but voffset + soffset is not reads 0.)  gemm256q / gemm256t rely on it: one descriptor per operand,
// tile / k position in the scalar offset, rows past the matrix read as zeros.
//   hipcc -O3 --offload-arch=gfx950 tools/rc_probe.hip -o tools/rc_probe.bin && ./tools/rc_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* a, float* out) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, 1024, 0x00020000);
    int soff = 2048;
    asm volatile("" : "+s"(soff));
    float v0 = __builtin_amdgcn_raw_buffer_load_b32(rs, threadIdx.x * 4, soff, 0);       // voff < num_records, voff+soff >= num_records
    float v1 = __builtin_amdgcn_raw_buffer_load_b32(rs, threadIdx.x * 4 + 2048, 0, 0);    // voff >= num_records
    float v2 = __builtin_amdgcn_raw_buffer_load_b32(rs, threadIdx.x * 4, 512, 0);         // inside with soff
    int s3 = 1000; asm volatile("" : "+s"(s3));
    float v3 = __builtin_amdgcn_raw_buffer_load_b32(rs, threadIdx.x * 4, s3, 0);          // lane 5: voff 20 + 1000 = 1020 < 1024 in; lane 6: 1024 out
    out[threadIdx.x] = v0; out[64 + threadIdx.x] = v1; out[128 + threadIdx.x] = v2; out[192 + threadIdx.x] = v3;
}
int main() {
    float *a, *o; hipMalloc(&a, 1 << 20); hipMalloc(&o, 1024);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i + 1;
    hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, o);
    float r[256]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("voff<n, voff+soff>=n : %g %g (0 => soffset IS range-checked; 513 => not)\n", r[0], r[1]);
    printf("voff>=n              : %g\n", r[64]);
    printf("inside with soff 512 : %g (expect 129)\n", r[128]);
    printf("soff 1000 lanes 4..7 : %g %g %g %g (expect 255 256 then 0 0 if checked)\n", r[192+4], r[192+5], r[192+6], r[192+7]);
    return 0;
}
