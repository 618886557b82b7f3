// Small HBM-bound helpers of the transformer path: column sums (bias gradients), token
// assembly (reference DSTTr.forward, network/vivit/vivit.py:133-142) and dtype casts.
#include "common.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------
// istvt_rows_reduce_add (common.h): one thread per (16-row group, column), a workgroup = 64 columns x 16 row groups; every
// group sums its rows in index order, the 16 group sums are added in group order by the group-0 thread.
__global__ __launch_bounds__(1024) void rows_reduce_add_kernel(const float* __restrict__ ws, int rows, int nacc, int N,
                                                               float* __restrict__ o0, float* __restrict__ o1,
                                                               float* __restrict__ o2) {
    __shared__ float part[16][64];
    const int a = blockIdx.y, cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const int per = (rows + 15) / 16, b0 = grp * per, b1 = min(rows, b0 + per);
    float s = 0.f;
    if (col < N) {
        const float* p = ws + (long)a * N + col;
        const long step = (long)nacc * N;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += p[b * step];
    }
    part[grp][cl] = s;
    __syncthreads();
    if (grp == 0 && col < N) {
        float t = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) t += part[g2][cl];
        float* out = a == 0 ? o0 : (a == 1 ? o1 : o2);
        out[col] += t;
    }
}

int istvt_rows_reduce_add(const float* ws, int rows, int nacc, int N, float* o0, float* o1, float* o2, hipStream_t stream) {
    if (rows <= 0 || nacc < 1 || nacc > 3 || N <= 0) return ISTVT_ERR_SHAPE;
    hipLaunchKernelGGL(rows_reduce_add_kernel, dim3((N + 63) / 64, nacc), dim3(1024), 0, stream, ws, rows, nacc, N, o0, o1, o2);
    return istvt_check_launch();
}

// the same through the C ABI: out[i] += sum_{r < rows} ws[r * n + i] in row order (any n; istvt_splitk_reduce is the
// 16-byte-vector form for n % 4 == 0)
extern "C" int istvt_rows_reduce(const float* ws, int rows, long n, float* out, hipStream_t stream) {
    if (!ws || !out || n <= 0 || n > 0x7fffffffL) return ISTVT_ERR_SHAPE;
    return istvt_rows_reduce_add(ws, rows, 1, (int)n, out, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------------------------------
// out[n] += sum_m x[m][n]   (bias gradient of a Linear: colsum of dY).  fp32 accumulate.  Thread owns 8 consecutive
// columns; 4 waves split the rows of a row block; every row block stores ONE partial row (ws[row block][N]) and
// istvt_rows_reduce_add folds them in index order.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ ws, long M, int N,
                                                     long ld, int rows_per_block) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + lane * 8;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < N) {
        // four rows in flight per wavefront: one dependent load per iteration left the kernel latency-bound
        long m = r0 + wid;
        for (; m + 12 < r1; m += 16) {
            float v0[8], v1[8], v2[8], v3[8];
            load8(x + m * ld + col, v0);
            load8(x + (m + 4) * ld + col, v1);
            load8(x + (m + 8) * ld + col, v2);
            load8(x + (m + 12) * ld + col, v3);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += (v0[i] + v1[i]) + (v2[i] + v3[i]);
        }
        for (; m < r1; m += 4) {
            float v[8];
            load8(x + m * ld + col, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[wid][lane * 8 + i] = acc[i];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int gc = blockIdx.x * 512 + c;
        if (gc < N) ws[(long)blockIdx.y * N + gc] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

static int colsum_rows_per_block(long M) {
    static const long rb = istvt_tune("ISTVT_COLSUM_ROWBLOCKS", 512);
    int rpb = (int)((M + rb - 1) / rb);
    return rpb < 64 ? 64 : rpb;
}
// float elements of the workspace istvt_colsum needs for an [M][N] input (one partial row per row block)
extern "C" int istvt_colsum_ws_elems(long M, int N) {
    if (M <= 0 || N <= 0) return ISTVT_ERR_SHAPE;
    const int rpb = colsum_rows_per_block(M);
    return (int)(((M + rpb - 1) / rpb) * N);
}

extern "C" int istvt_colsum(const void* x, float* out, long M, int N, long ld, float* ws, long ws_elems, int dtype,
                            hipStream_t stream) {
    if (M <= 0 || N <= 0 || N % 8 != 0 || ld % 8 != 0 || !ws) return ISTVT_ERR_SHAPE;
    const int rpb = colsum_rows_per_block(M);
    const int rblocks = (int)((M + rpb - 1) / rpb);
    if (ws_elems < (long)rblocks * N) return ISTVT_ERR_SHAPE;
    dim3 grid((N + 511) / 512, (unsigned)rblocks), block(256);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((colsum_kernel<T>), grid, block, 0, stream, (const T*)x, ws, M, N, ld, rpb));
    int rc = istvt_check_launch();
    if (rc) return rc;
    return istvt_rows_reduce_add(ws, rblocks, 1, N, out, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------------------------------
// Token assembly.  feats [B][T][hw][D] (the stem's NHWC output IS this layout), P = hw + 1,
// F = T + 1, x [B][F][P][D]:
//   x[b,0,p]       = temporal_token                          (vivit.py:139-140, no pos-emb)
//   x[b,1+t,0]     = space_token + pos[t][0]                 (vivit.py:136-138)
//   x[b,1+t,1+i]   = feats[b,t,i] + pos[t][1+i]
// pos is the parameter's leading [T][P] slice with row stride pos_ld (>= P) rows per frame.
template <typename T>
__global__ __launch_bounds__(128) void tokens_fwd_kernel(const T* __restrict__ feats, const float* __restrict__ space,
                                                         const float* __restrict__ temporal,
                                                         const float* __restrict__ pos, T* __restrict__ x, int B,
                                                         int F, int P, int D, int pos_rows, long ldx) {
    const long row = blockIdx.x;                 // over B*F*P
    const int p = (int)(row % P);
    const int f = (int)((row / P) % F);
    const long b = row / ((long)P * F);
    const int hw = P - 1, Tn = F - 1;
    for (int e = threadIdx.x * 8; e < D; e += 128 * 8) {
        float o[8];
        if (f == 0) {
            load8(temporal + e, o);
        } else {
            float pe[8];
            load8(pos + ((long)(f - 1) * pos_rows + p) * D + e, pe);
            if (p == 0) load8(space + e, o);
            else load8(feats + (((b * Tn + (f - 1)) * hw) + (p - 1)) * D + e, o);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] += pe[i];
        }
        store8(x + row * ldx + e, o);
    }
}

// backward of the assembly.  grid (P, F); loops over b.
//   dfeats[b,t,i] = dx[b,1+t,1+i];  dpos[t][p] += sum_b dx[b,1+t,p];
//   dspace += sum_{b,t} dx[b,1+t,0];  dtemporal += sum_{b,p} dx[b,0,p]   (per-block partial rows + the fixed-order reduce)
template <typename T>
__global__ __launch_bounds__(128) void tokens_bwd_kernel(const T* __restrict__ dx, T* __restrict__ dfeats,
                                                         float* __restrict__ ws, float* __restrict__ dpos, int B, int F,
                                                         int P, int D, int pos_rows, long lddx) {
    // ws: [P][D] partial rows of dtemporal (blocks f == 0) then [F - 1][D] partial rows of dspace (blocks p == 0, f > 0)
    const int p = blockIdx.x, f = blockIdx.y;
    const int hw = P - 1, Tn = F - 1;
    for (int e = threadIdx.x * 8; e < D; e += 128 * 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (long b = 0; b < B; ++b) {
            float v[8];
            load8(dx + ((b * F + f) * P + p) * lddx + e, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v[i];
            if (f > 0 && p > 0 && dfeats) store8(dfeats + (((b * Tn + (f - 1)) * hw) + (p - 1)) * D + e, v);
        }
        if (f == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ws[(long)p * D + e + i] = acc[i];
        } else {
            float* dp = dpos + ((long)(f - 1) * pos_rows + p) * D + e;
#pragma unroll
            for (int i = 0; i < 8; ++i) dp[i] += acc[i];
            if (p == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) ws[(long)(P + f - 1) * D + e + i] = acc[i];
            }
        }
    }
}

extern "C" int istvt_tokens_fwd(const void* feats, const float* space, const float* temporal, const float* pos,
                                void* x, long ldx, int B, int F, int P, int D, int pos_rows, int dtype,
                                hipStream_t stream) {
    if (B <= 0 || F < 2 || P < 2 || D % 8 != 0 || pos_rows < P || ldx < D || ldx % 8) return ISTVT_ERR_SHAPE;
    dim3 grid((unsigned)((long)B * F * P)), block(128);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((tokens_fwd_kernel<T>), grid, block, 0, stream, (const T*)feats, space,
                                             temporal, pos, (T*)x, B, F, P, D, pos_rows, ldx));
    return istvt_check_launch();
}

// dfeats may be null (features do not require grad).  dspace/dtemporal/dpos accumulate (fp32).
// ws: float scratch of (P + F - 1) * D elements
extern "C" int istvt_tokens_bwd(const void* dx, long lddx, void* dfeats, float* dspace, float* dtemporal, float* dpos,
                                float* ws, int B, int F, int P, int D, int pos_rows, int dtype, hipStream_t stream) {
    if (B <= 0 || F < 2 || P < 2 || D % 8 != 0 || pos_rows < P || lddx < D || lddx % 8 || !ws) return ISTVT_ERR_SHAPE;
    dim3 grid(P, F), block(128);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((tokens_bwd_kernel<T>), grid, block, 0, stream, (const T*)dx, (T*)dfeats,
                                             ws, dpos, B, F, P, D, pos_rows, lddx));
    int rc = istvt_check_launch();
    if (rc) return rc;
    rc = istvt_rows_reduce_add(ws, P, 1, D, dtemporal, nullptr, nullptr, stream);
    if (rc) return rc;
    return istvt_rows_reduce_add(ws + (long)P * D, F - 1, 1, D, dspace, nullptr, nullptr, stream);
}


// ------------------------------------------------------------------------------------------
// Frame difference of TemporalResidualAttention (module.py:193) for callers that hand the module
// an already-normalised input (the PreNorm path fuses this into layernorm.hip):
//   forward : out[f] = x[f] - (f >= 2 ? x[f-1] : 0)
//   backward: out[f] = g[f] - (1 <= f <= F-2 ? g[f+1] : 0)          (adjoint)
template <typename T>
__global__ __launch_bounds__(256) void frame_diff_kernel(const T* __restrict__ x, T* __restrict__ out, long M, int F,
                                                         int P, int D, int adjoint) {
    const long nvec = M * (D / 8);
    const long stride = (long)gridDim.x * 256;
    const int vpr = D / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        const long m = i / vpr;
        const int e = (int)(i % vpr) * 8;
        const int f = (int)((m / P) % F);
        float a[8];
        load8(x + m * D + e, a);
        const bool sub = adjoint ? (f >= 1 && f <= F - 2) : (f >= 2);
        if (sub) {
            float b[8];
            load8(x + (adjoint ? m + P : m - P) * D + e, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] -= b[j];
        }
        store8(out + m * D + e, a);
    }
}

extern "C" int istvt_frame_diff(const void* x, void* out, int B, int F, int P, int D, int adjoint, int dtype,
                                hipStream_t stream) {
    if (B <= 0 || F <= 0 || P <= 0 || D % 8 != 0) return ISTVT_ERR_SHAPE;
    const long M = (long)B * F * P;
    long blocks = (M * (D / 8) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks), block(256);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((frame_diff_kernel<T>), grid, block, 0, stream, (const T*)x, (T*)out, M, F,
                                             P, D, adjoint));
    return istvt_check_launch();
}

// ------------------------------------------------------------------------------------------
// dtype casts (fp32 master weights -> bf16 GEMM operands; T -> fp32 for the loss head)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * 256 * 8;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            float v[8];
            load8(in + i, v);
            store8(out + i, v);
        } else {
            for (long j = i; j < n; ++j) out[j] = from_f32<TO>(to_f32(in[j]));
        }
    }
}

extern "C" int istvt_cast(const void* in, int in_dtype, void* out, int out_dtype, long n, hipStream_t stream) {
    if (n <= 0) return ISTVT_ERR_SHAPE;
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    dim3 grid((unsigned)blocks), block(256);
    if (in_dtype == DT_F32 && out_dtype == DT_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, block, 0, stream, (const float*)in, (bf16_t*)out, n);
    else if (in_dtype == DT_BF16 && out_dtype == DT_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, block, 0, stream, (const bf16_t*)in, (float*)out, n);
    else if (in_dtype == DT_F32 && out_dtype == DT_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), grid, block, 0, stream, (const float*)in, (float*)out, n);
    else if (in_dtype == DT_BF16 && out_dtype == DT_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), grid, block, 0, stream, (const bf16_t*)in, (bf16_t*)out, n);
    else return ISTVT_ERR_DTYPE;
    return istvt_check_launch();
}

// rows x cols cast between row-strided buffers (the bf16 GEMM operand copies of the fp32 weights get line-aligned rows)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast2d_kernel(const TI* __restrict__ in, long ldi, TO* __restrict__ out, long ldo,
                                                     long rows, int cols) {
    const int vpr = cols / 8;
    const long nvec = rows * vpr;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        const long m = i / vpr;
        const int e = (int)(i % vpr) * 8;
        float v[8];
        load8(in + m * ldi + e, v);
        store8(out + m * ldo + e, v);
    }
}

extern "C" int istvt_cast2d(const void* in, int in_dtype, long ldi, void* out, int out_dtype, long ldo, long rows,
                            int cols, hipStream_t stream) {
    if (rows <= 0 || cols <= 0 || cols % 8 || ldi < cols || ldo < cols || ldi % 8 || ldo % 8) return ISTVT_ERR_SHAPE;
    long blocks = (rows * (cols / 8) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    dim3 grid((unsigned)blocks), block(256);
    if (in_dtype == DT_F32 && out_dtype == DT_BF16)
        hipLaunchKernelGGL((cast2d_kernel<float, bf16_t>), grid, block, 0, stream, (const float*)in, ldi, (bf16_t*)out, ldo, rows, cols);
    else if (in_dtype == DT_BF16 && out_dtype == DT_F32)
        hipLaunchKernelGGL((cast2d_kernel<bf16_t, float>), grid, block, 0, stream, (const bf16_t*)in, ldi, (float*)out, ldo, rows, cols);
    else if (in_dtype == DT_F32 && out_dtype == DT_F32)
        hipLaunchKernelGGL((cast2d_kernel<float, float>), grid, block, 0, stream, (const float*)in, ldi, (float*)out, ldo, rows, cols);
    else if (in_dtype == DT_BF16 && out_dtype == DT_BF16)
        hipLaunchKernelGGL((cast2d_kernel<bf16_t, bf16_t>), grid, block, 0, stream, (const bf16_t*)in, ldi, (bf16_t*)out, ldo, rows, cols);
    else return ISTVT_ERR_DTYPE;
    return istvt_check_launch();
}

// fp32 master weight [R][C] -> bf16 copy [R][C] AND its transpose [C][R], both with caller-given (line-aligned) row
// strides, in one pass over the fp32 data: the forward GEMM reads W as its B operand, the input-gradient GEMM reads
// W^T the same way.  (Separately: a cast kernel plus a strided torch copy per weight, 0.8 ms per step for the copies.)
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ in, long ldi,
                                                             bf16_t* __restrict__ out, long ldo,
                                                             bf16_t* __restrict__ outT, long ldt, int R, int C) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int row = threadIdx.x >> 2, cc = (threadIdx.x & 3) * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = c0 + cc + 8 * h;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool ok = r0 + row < R && col < C;
        if (ok) load8(in + (long)(r0 + row) * ldi + col, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[row][cc + 8 * h + j] = v[j];
        if (ok) store8(out + (long)(r0 + row) * ldo + col, v);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int tr = c0 + row;                 // row of the transpose = column of the input
        const int tc = r0 + cc + 8 * h;
        if (tr < C && tc < R) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[cc + 8 * h + j][row];
            store8(outT + (long)tr * ldt + tc, v);
        }
    }
}

// The same for up to ISTVT_CAST_GROUP_MAX weights in ONE launch: the 84 per-weight launches of a training step (one per
// nn.Linear weight whose fp32 master copy the optimizer has just changed) are each far below the ~5 us a launch occupies
// the queue for (0.45 ms per step at C2); grouped they are a handful.  Workgroup b belongs to the problem whose
// start[] range holds it; inside the problem the tile order is that of cast_transpose_kernel.
constexpr int ISTVT_CAST_GROUP_MAX = 32;
struct CastGroupArgs {
    const float* in[ISTVT_CAST_GROUP_MAX];
    bf16_t* out[ISTVT_CAST_GROUP_MAX];
    bf16_t* outT[ISTVT_CAST_GROUP_MAX];
    long ldi[ISTVT_CAST_GROUP_MAX], ldo[ISTVT_CAST_GROUP_MAX], ldt[ISTVT_CAST_GROUP_MAX];
    int R[ISTVT_CAST_GROUP_MAX], C[ISTVT_CAST_GROUP_MAX];
    int start[ISTVT_CAST_GROUP_MAX + 1];
    int count;
};
__global__ __launch_bounds__(256) void cast_transpose_group_kernel(CastGroupArgs g) {
    __shared__ float tile[64][65];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < ISTVT_CAST_GROUP_MAX; ++i)
        if (i < g.count && (int)blockIdx.x >= g.start[i]) pi = i;
    const float* __restrict__ in = g.in[pi];
    bf16_t* __restrict__ out = g.out[pi];
    bf16_t* __restrict__ outT = g.outT[pi];
    const long ldi = g.ldi[pi], ldo = g.ldo[pi], ldt = g.ldt[pi];
    const int R = g.R[pi], C = g.C[pi];
    const int b = (int)blockIdx.x - g.start[pi], tx = (C + 63) / 64;
    const int r0 = (b / tx) * 64, c0 = (b % tx) * 64;
    const int row = threadIdx.x >> 2, cc = (threadIdx.x & 3) * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = c0 + cc + 8 * h;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool ok = r0 + row < R && col < C;
        if (ok) load8(in + (long)(r0 + row) * ldi + col, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[row][cc + 8 * h + j] = v[j];
        if (ok) store8(out + (long)(r0 + row) * ldo + col, v);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int tr = c0 + row;
        const int tc = r0 + cc + 8 * h;
        if (tr < C && tc < R) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[cc + 8 * h + j][row];
            store8(outT + (long)tr * ldt + tc, v);
        }
    }
}

// in[i] float [R_i][C_i] (row stride ldi_i) -> out[i] bf16 [R_i][C_i] (ldo_i) and outT[i] bf16 [C_i][R_i] (ldt_i), i < count
// (any count: launched in groups of 32)
extern "C" int istvt_cast_transpose_group(int count, const float* const* in, const long* ldi, void* const* out,
                                          const long* ldo, void* const* outT, const long* ldt, const int* R, const int* C,
                                          hipStream_t stream) {
    if (count < 1) return ISTVT_ERR_SHAPE;
    for (int i = 0; i < count; ++i)
        if (R[i] <= 0 || C[i] <= 0 || R[i] % 8 || C[i] % 8 || ldi[i] < C[i] || ldo[i] < C[i] || ldt[i] < R[i] || ldi[i] % 4 ||
            ldo[i] % 8 || ldt[i] % 8 || !in[i] || !out[i] || !outT[i]) return ISTVT_ERR_SHAPE;
    for (int i0 = 0; i0 < count; i0 += ISTVT_CAST_GROUP_MAX) {
        CastGroupArgs g;
        g.count = count - i0 < ISTVT_CAST_GROUP_MAX ? count - i0 : ISTVT_CAST_GROUP_MAX;
        int start = 0;
        for (int j = 0; j < ISTVT_CAST_GROUP_MAX; ++j) {
            const int i = i0 + (j < g.count ? j : 0);
            g.in[j] = in[i]; g.out[j] = (bf16_t*)out[i]; g.outT[j] = (bf16_t*)outT[i];
            g.ldi[j] = ldi[i]; g.ldo[j] = ldo[i]; g.ldt[j] = ldt[i]; g.R[j] = R[i]; g.C[j] = C[i];
            g.start[j] = start;
            if (j < g.count) start += ((C[i] + 63) / 64) * ((R[i] + 63) / 64);
        }
        g.start[ISTVT_CAST_GROUP_MAX] = start;
        hipLaunchKernelGGL(cast_transpose_group_kernel, dim3(start), dim3(256), 0, stream, g);
        const int rc = istvt_check_launch();
        if (rc) return rc;
    }
    return ISTVT_OK;
}

extern "C" int istvt_cast_transpose(const float* in, long ldi, void* out, long ldo, void* outT, long ldt, int R, int C,
                                    hipStream_t stream) {
    if (R <= 0 || C <= 0 || R % 8 || C % 8 || ldi < C || ldo < C || ldt < R || ldi % 4 || ldo % 8 || ldt % 8) return ISTVT_ERR_SHAPE;
    dim3 grid((C + 63) / 64, (R + 63) / 64), block(256);
    hipLaunchKernelGGL(cast_transpose_kernel, grid, block, 0, stream, in, ldi, (bf16_t*)out, ldo, (bf16_t*)outT, ldt, R, C);
    return istvt_check_launch();
}
