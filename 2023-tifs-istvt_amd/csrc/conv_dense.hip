// The two dense 3x3 convolutions that open the Xception entry flow (reference: network/xception.py:118-123,
// 193-199), computed directly from the NHWC / NCHW activations -- no im2col matrix in HBM.
//
//   conv1  (3 -> 32, stride 2, pad 0):  fp32 NCHW clip -> u1 [frames][Ho][Wo][32].  27 x 32 multiply-adds per output
//          pixel: one thread per pixel on the vector ALU, weights read through the scalar cache.
//   conv2  (32 -> 64, stride 1, pad 0) on a1 = relu(bn1(u1)):
//          forward         D[co][px] = sum_{tap,ci} W[co][tap][ci] a1[px + tap][ci]      K = 9 x 32
//          input gradient  D[ci][px] = sum_{tap,co} W[co][tap][ci] du2[px - tap][co]     K = 9 x 64, masked by relu'
//          weight gradient D[co][tap][ci] = sum_px du2[px][co] a1[px + tap][ci]          K = pixels
//          all three on v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the row operand and 16 consecutive pixels as the
//          column operand, so a lane ends up with consecutive channels of one pixel and stores them as 16-byte pieces.
//          A pixel fragment is one 16-byte load straight from the activation (lane (r, g): pixel r, channels 8g..8g+7);
//          BatchNorm + ReLU of bn1 are applied to it in registers.
//
// These maps are the largest of the network (111 x 111 x 32 and 109 x 109 x 64 per frame) and every pass over them is
// HBM-bound; algorithmic bytes per frame at S = 224 (bf16): conv2 forward 0.79 MB in + 1.52 MB out.
#include "common.h"

namespace {

__device__ __forceinline__ bf16x8 ld_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 zero_frag() {
    bf16x8 f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (bf16_t)0.0f;
    return f;
}
__device__ __forceinline__ f32x4 mma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------ conv1 forward
template <typename T>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        T* __restrict__ u1, long Mo, int S, int Ho, int Wo) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= Mo) return;
    const int xo = (int)(m % Wo), yo = (int)((m / Wo) % Ho);
    const long f = m / ((long)Wo * Ho);
    float v[27];                                     // [ci][dy][dx]: the order of conv1.weight[co]
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const float* row = x + ((f * 3 + ci) * S + 2 * yo + dy) * S + 2 * xo;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) v[ci * 9 + dy * 3 + dx] = row[dx];
        }
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* wc = w + (c8 * 8 + j) * 27;     // uniform address: scalar loads
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 27; ++k) a = fmaf(wc[k], v[k], a);
            o[j] = a;
        }
        store8(u1 + m * 32 + c8 * 8, o);
    }
}

// ------------------------------------------------------------------------------------------ conv2 forward
// wf[tap][nt]: row operand fragments; MFMA row rho of n-tile nt is output channel 16 (rho >> 2) + 4 nt + (rho & 3)
// so that lane (r, g) accumulates channels 16g .. 16g+15 of pixel r over its four n-tiles.
__global__ __launch_bounds__(256, 2) void conv2_fwd_kernel(const bf16_t* __restrict__ u1, const float* __restrict__ bnp,
                                                           const bf16_t* __restrict__ w, bf16_t* __restrict__ u2, long Mo,
                                                           int H, int W, int Ho, int Wo, int ngroups) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    // 27 of the 36 row-operand fragments live in registers (108 VGPRs): all of them from LDS would cost 36 KiB of
    // reads per 16 pixels, twice the MFMA time; the fourth n-tile's nine are read from LDS every group (register budget)
    __shared__ bf16x8 wl3[9 * 64];
    bf16x8 wf[9][3];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
            wf[tap][nt] = ld_frag(w + (16 * (r >> 2) + 4 * nt + (r & 3)) * 288 + tap * 32 + 8 * g);
        if (wave == 0) wl3[tap * 64 + lane] = ld_frag(w + (16 * (r >> 2) + 4 * 3 + (r & 3)) * 288 + tap * 32 + 8 * g);
    }
    __syncthreads();
    float mu[8], sc[8], be[8];
    load8(bnp + 8 * g, mu);
    load8(bnp + 2 * 32 + 8 * g, sc);
    load8(bnp + 3 * 32 + 8 * g, be);
    // Work item = (frame, 16-pixel strip, 9 output rows).  Output row y reads input rows y, y+1, y+2; walking down the
    // rows only row y+2 is new: it is loaded and put through bn1 + ReLU ONCE (3 fragments instead of 9 per 16 pixels)
    // into a three-row register ring (slot = input row mod 3: the row loop is unrolled, segments start at multiples of 3).
    constexpr int RSEG = 9;
    const int nstrips = (Wo + 15) / 16, nsegs = (Ho + RSEG - 1) / RSEG;
    const long nitems = (long)(Mo / ((long)Ho * Wo)) * nstrips * nsegs;
    bf16x8 ring[3][3];                                  // [input row mod 3][dx], already normalised + rectified
    for (long it = (long)blockIdx.x * 4 + wave; it < nitems; it += (long)gridDim.x * 4) {
        const int seg = (int)(it % nsegs);
        const int strip = (int)((it / nsegs) % nstrips);
        const long f = it / ((long)nsegs * nstrips);
        const int xo = strip * 16 + r;
        const bool ok = xo < Wo;
        const int xc = ok ? xo : Wo - 1;                // lanes past the row end re-read its last pixel, never store
        const int y0 = seg * RSEG;
        auto load_row = [&](int yi, bf16x8 (&dst)[3]) {
            const int yc = yi < H ? yi : H - 1;         // rows past the image only feed output rows that are skipped
            const bf16_t* sp = u1 + ((f * H + yc) * W + xc) * 32 + 8 * g;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const bf16x8 raw = ld_frag(sp + dx * 32);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    dst[dx][j] = (bf16_t)fmaxf(((float)raw[j] - mu[j]) * sc[j] + be[j], 0.f);
            }
        };
        load_row(y0, ring[0]);
        load_row(y0 + 1, ring[1]);
#pragma unroll
        for (int k = 0; k < RSEG; ++k) {
            const int yo = y0 + k;
            if (yo < Ho) {                              // uniform (a break would keep the loop from unrolling)
                load_row(yo + 2, ring[(k + 2) % 3]);
                f32x4 acc[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                int lo = lane;
                asm volatile("" : "+v"(lo));            // keeps the LDS fragment reads inside the loop
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)          // rows y, y+1 are in registers while row y+2 arrives
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const bf16x8& xf = ring[(k + dy) % 3][dx];
#pragma unroll
                        for (int nt = 0; nt < 3; ++nt) acc[nt] = mma16(wf[dy * 3 + dx][nt], xf, acc[nt]);
                        acc[3] = mma16(wl3[(dy * 3 + dx) * 64 + lo], xf, acc[3]);
                    }
                if (ok) {
                    bf16x8 o0, o1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        o0[i] = (bf16_t)acc[0][i]; o0[4 + i] = (bf16_t)acc[1][i];
                        o1[i] = (bf16_t)acc[2][i]; o1[4 + i] = (bf16_t)acc[3][i];
                    }
                    bf16_t* dst = u2 + (((f * Ho + yo) * Wo) + xo) * 64 + 16 * g;
                    *reinterpret_cast<bf16x8*>(dst) = o0;
                    *reinterpret_cast<bf16x8*>(dst + 8) = o1;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ conv2 input gradient
// wf[tap][ks][nt]: row = input channel 8 (rho >> 2) + 4 nt + (rho & 3), k = output channel 32 ks + 8 g + j.
// dz1[pixel][ci] = relu'(bn1(u1)) * sum over the taps whose output pixel (y - dy, x - dx) exists.
__global__ __launch_bounds__(256, 2) void conv2_dgrad_kernel(const bf16_t* __restrict__ du2, const bf16_t* __restrict__ w,
                                                             const bf16_t* __restrict__ u1, const float* __restrict__ bnp,
                                                             bf16_t* __restrict__ dz1, long Mi, int H, int W, int Ho,
                                                             int Wo, int ngroups) {
    __shared__ bf16_t wt[9 * 32 * 64];                  // [tap][ci][co]: the weight transposed once per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    for (int i = tid; i < 64 * 36; i += 256) {          // 16-byte pieces of w [64][288]
        const int co = i / 36, k8 = (i % 36) * 8;
        const bf16x8 v = ld_frag(w + co * 288 + k8);
#pragma unroll
        for (int j = 0; j < 8; ++j) wt[(k8 + j) * 64 + co] = v[j];
    }
    __syncthreads();
    // all 36 row-operand fragments in registers.  The loads are what bounds this kernel: a du2 pixel is one 128-byte
    // line, so every fragment load of 16 pixels touches 16 lines.
    // taps of dy = 1, 2 (24 fragments) stay in registers; the 12 of dy = 0 are read from the LDS image at each use
    // (register budget: 144 weight + 72 ring registers spilled)
    bf16x8 wf[9][2][2];
#pragma unroll
    for (int tap = 3; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                wf[tap][ks][nt] = ld_frag(wt + (tap * 32 + 8 * (r >> 2) + 4 * nt + (r & 3)) * 64 + 32 * ks + 8 * g);
    const int wrow = 8 * (r >> 2) + (r & 3), wcol = 8 * g;
    // Work item = (frame, 16-pixel strip, 9 input rows).  Input row y needs the output-gradient rows y, y-1, y-2; walking
    // down the rows only row y is new (6 fragment loads = 96 lines per 16 pixels instead of 18 loads = 288 lines): the
    // three rows live in a register ring whose slot is the row index mod 3 -- compile-time, the row loop is unrolled by
    // three and segments start at multiples of three.
    constexpr int RSEG = 9;
    const int nstrips = (W + 15) / 16, nsegs = (H + RSEG - 1) / RSEG;
    const long nitems = (long)(Mi / ((long)H * W)) * nstrips * nsegs;
    bf16x8 ring[3][3][2];                               // [row mod 3][dx][ks]
    for (long it = (long)blockIdx.x * 4 + wave; it < nitems; it += (long)gridDim.x * 4) {
        const int seg = (int)(it % nsegs);
        const int strip = (int)((it / nsegs) % nstrips);
        const long f = it / ((long)nsegs * nstrips);
        const int xi = strip * 16 + r;
        const int y0 = seg * RSEG;
        auto load_row = [&](int yo, bf16x8 (&dst)[3][2]) {
            const bool rowok = yo >= 0 && yo < Ho;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xo = xi - dx;
                const bool in = rowok && xo >= 0 && xo < Wo;
                const bf16_t* sp = du2 + ((f * Ho + yo) * Wo + xo) * 64 + 8 * g;
                dst[dx][0] = in ? ld_frag(sp) : zero_frag();
                dst[dx][1] = in ? ld_frag(sp + 32) : zero_frag();
            }
        };
        load_row(y0 - 2, ring[(RSEG * 3 - 2) % 3]);     // y0 is a multiple of 3: rows y0-2, y0-1 sit in slots 1, 2
        load_row(y0 - 1, ring[(RSEG * 3 - 1) % 3]);
#pragma unroll
        for (int k = 0; k < RSEG; ++k) {
            const int yi = y0 + k;
            if (yi < H) {                               // uniform (a break would keep the loop from unrolling)
            load_row(yi, ring[k % 3]);
            const bool ok = xi < W;
            const long m = (f * H + yi) * W + (ok ? xi : W - 1);
            const bf16x8 uraw = ld_frag(u1 + m * 32 + 8 * g);
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            // the two older rows first: their fragments are already in registers while row yi's loads are in flight
#pragma unroll
            for (int dyo = 2; dyo >= 1; --dyo) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[nt] = mma16(wf[dyo * 3 + dx][ks][nt], ring[(k + 3 - dyo) % 3][dx][ks], acc[nt]);
            }
            int wo = wrow;
            asm volatile("" : "+v"(wo));                // keeps the LDS fragment reads inside the loop
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[nt] = mma16(ld_frag(wt + (dx * 32 + wo + 4 * nt) * 64 + 32 * ks + wcol), ring[k % 3][dx][ks], acc[nt]);
            if (ok) {                                   // lane (r, g): channels 8g + 4nt + i of pixel r
                float mu[8], sc[8], be[8];              // re-read per row (L1): no registers held across the MFMAs
                load8(bnp + 8 * g, mu);
                load8(bnp + 2 * 32 + 8 * g, sc);
                load8(bnp + 3 * 32 + 8 * g, be);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float z = ((float)uraw[j] - mu[j]) * sc[j] + be[j];
                    o[j] = (bf16_t)(z > 0.f ? acc[j >> 2][j & 3] : 0.f);
                }
                *reinterpret_cast<bf16x8*>(dz1 + m * 32 + 8 * g) = o;
            }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ conv2 weight gradient
// One workgroup walks chunks of 32 consecutive output pixels of one row.  Per chunk it stages, in LDS,
//   dimg [32 px][64 co]      the output gradients (zero rows past the end of the row), and
//   aimg [3][34 px][32 ci]   relu(bn1(u1)) for input rows y..y+2, pixels x0..x0+33 (zero past the row end),
// and every wave owns one 16-channel slice of co: its row operand is the transposed read of dimg (8 consecutive pixels
// of one channel, ds_read_b64_tr_b16), its column operands the transposed reads of aimg shifted by the tap.
// 18 accumulator tiles per wave ([tap][2 ci tiles]); at the end each workgroup writes its fp32 partial [64][288] to
// its slab and istvt_rows_reduce_add sums the slabs into the gradient in slab order (no atomics).
constexpr int WG_PX = 32;
constexpr int DIMG_PITCH = 64 * 2 + 16;       // bytes per pixel row (+16: the 4 k-rows of one tr read land in 4 bank groups)
constexpr int AIMG_PITCH = 32 * 2 + 16;
constexpr int AIMG_ROWS = WG_PX + 2;

// 8 consecutive k-rows (k0..k0+7) of column col16 + r from an image with `pitch` bytes per k-row
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int pitch, int k0, int col16, int r) {
    typedef short4v __attribute__((address_space(3))) * lds_ptr;
    typedef short short8v __attribute__((ext_vector_type(8)));
    const int q = r >> 2, pp = r & 3;
    const char* pa = img + (k0 + q) * pitch + (col16 + 4 * pp) * 2;
    const char* pb = pa + 4 * pitch;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pa));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(pb));
    short8v s;
    s[0] = lo[0]; s[1] = lo[1]; s[2] = lo[2]; s[3] = lo[3]; s[4] = hi[0]; s[5] = hi[1]; s[6] = hi[2]; s[7] = hi[3];
    return __builtin_bit_cast(bf16x8, s);
}

__global__ __launch_bounds__(256) void conv2_wgrad_kernel(const bf16_t* __restrict__ du2, const bf16_t* __restrict__ u1,
                                                          const float* __restrict__ bnp, float* __restrict__ slabs,
                                                          int Fr, int H, int W, int Ho, int Wo, int nchunks) {
    __shared__ __attribute__((aligned(16))) char dimg[WG_PX * DIMG_PITCH];
    __shared__ __attribute__((aligned(16))) char aimg[3 * AIMG_ROWS * AIMG_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int cpr = (Wo + WG_PX - 1) / WG_PX;          // chunks per output row
    f32x4 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging roles: thread -> (pixel, 16-byte piece)
    const int dpx = tid >> 3, dpc = tid & 7;            // dimg: 32 px x 8 pieces
    float mu[8], sc[8], be[8];                          // aimg pieces are channels 8 (tid & 3)
    load8(bnp + 8 * (tid & 3), mu);
    load8(bnp + 2 * 32 + 8 * (tid & 3), sc);
    load8(bnp + 3 * 32 + 8 * (tid & 3), be);
    // global -> registers for one chunk (the next chunk's are in flight during this chunk's MFMAs)
    bf16x8 dreg, areg[2];
    auto gload = [&](int ch) {
        const int x0 = (ch % cpr) * WG_PX;
        const int rowi = ch / cpr;                      // (frame, output row)
        const int yo = rowi % Ho;
        const long f = rowi / Ho;
        dreg = zero_frag();
        if (x0 + dpx < Wo) dreg = ld_frag(du2 + ((f * Ho + yo) * Wo + x0 + dpx) * 64 + dpc * 8);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + 256 * k;
            const int px = (i >> 2) % AIMG_ROWS, dy = (i >> 2) / AIMG_ROWS;
            areg[k] = zero_frag();
            if (i < 3 * AIMG_ROWS * 4 && x0 + px < W)
                areg[k] = ld_frag(u1 + ((f * H + yo + dy) * W + x0 + px) * 32 + (tid & 3) * 8);
        }
    };
    int ch = blockIdx.x;
    if (ch < nchunks) gload(ch);
    for (; ch < nchunks; ch += gridDim.x) {
        // ---- stage: registers -> LDS, bn1 + ReLU on the way (pixels past the row end stay zero)
        const int x0 = (ch % cpr) * WG_PX;
        *reinterpret_cast<bf16x8*>(dimg + dpx * DIMG_PITCH + dpc * 16) = dreg;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + 256 * k;
            const int px = (i >> 2) % AIMG_ROWS, dy = (i >> 2) / AIMG_ROWS;
            if (i < 3 * AIMG_ROWS * 4) {
                bf16x8 v = zero_frag();
                if (x0 + px < W) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)fmaxf(((float)areg[k][j] - mu[j]) * sc[j] + be[j], 0.f);
                }
                *reinterpret_cast<bf16x8*>(aimg + (dy * AIMG_ROWS + px) * AIMG_PITCH + (tid & 3) * 16) = v;
            }
        }
        __syncthreads();
        if (ch + (int)gridDim.x < nchunks) gload(ch + gridDim.x);
        // ---- 18 MFMA per wave: rows = co 16 wave + r, cols = ci, k = the chunk's 32 pixels
        const bf16x8 df = tr_frag(dimg, DIMG_PITCH, 8 * g, 16 * wave, r);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const char* ab = aimg + ((tap / 3) * AIMG_ROWS + (tap % 3)) * AIMG_PITCH;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[tap][ct] = mma16(df, tr_frag(ab, AIMG_PITCH, 8 * g, 16 * ct, r), acc[tap][ct]);
        }
        __syncthreads();
    }
    // lane (r, g) of tile (tap, ct): rows co = 16 wave + 4g + i, column ci = 16 ct + r
    float* out = slabs + (long)blockIdx.x * 64 * 288;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) out[(16 * wave + 4 * g + i) * 288 + tap * 32 + 16 * ct + r] = acc[tap][ct][i];
}

// ------------------------------------------------------------------------------------------ conv1 weight gradient
// dW1[co][k] = sum_px du1[px][co] * patch[px][k], k = (ci, dy, dx) as conv1.weight stores it (27, padded to 32).
// One chunk = one output row (Wo <= 128 pixels).  Staged per chunk: dimg [128 px][32 co] (the row of du1) and pimg
// [128 px][32 k] (the im2col patches of that row, gathered from the fp32 clip and rounded to bf16 -- the rounding the
// im2col + GEMM path applies); wave (mt, nt) owns the 16 x 16 tile (co tile mt, k tile nt): four k-steps of 32 pixels.
constexpr int C1_PX = 128;
constexpr int C1_PITCH = 32 * 2 + 16;

template <typename TD>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const TD* __restrict__ du1, const float* __restrict__ x,
                                                          float* __restrict__ slabs, int S, int Ho, int Wo, int nrows) {
    __shared__ __attribute__((aligned(16))) char dimg[C1_PX * C1_PITCH];
    __shared__ __attribute__((aligned(16))) char pimg[C1_PX * C1_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int mt = wave >> 1, nt = wave & 1;
    const int ppx = tid >> 1, pk0 = (tid & 1) * 16;          // patch role: pixel, first of 16 k
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 dreg[2];
    float xreg[16];
    auto gload = [&](int row) {
        const int yo = row % Ho;
        const long f = row / Ho;
#pragma unroll
        for (int k = 0; k < 2; ++k) {                        // 128 px x 4 pieces of 8 channels
            const int i = tid + 256 * k;
            const int px = i >> 2, pc = i & 3;
            dreg[k] = zero_frag();
            if (px < Wo) {
                float v[8];
                load8(du1 + ((f * Ho + yo) * Wo + px) * 32 + pc * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) dreg[k][j] = (bf16_t)v[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = pk0 + j;                           // (ci, dy, dx)
            const int ci = k / 9, dy = (k % 9) / 3, dx = k % 3;
            xreg[j] = 0.f;
            if (k < 27 && ppx < Wo) xreg[j] = x[((f * 3 + ci) * S + 2 * yo + dy) * S + 2 * ppx + dx];
        }
    };
    int row = blockIdx.x;
    if (row < nrows) gload(row);
    for (; row < nrows; row += gridDim.x) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + 256 * k;
            *reinterpret_cast<bf16x8*>(dimg + (i >> 2) * C1_PITCH + (i & 3) * 16) = dreg[k];
        }
        {
            bf16x8 p0, p1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { p0[j] = (bf16_t)xreg[j]; p1[j] = (bf16_t)xreg[8 + j]; }
            *reinterpret_cast<bf16x8*>(pimg + ppx * C1_PITCH + pk0 * 2) = p0;
            *reinterpret_cast<bf16x8*>(pimg + ppx * C1_PITCH + pk0 * 2 + 16) = p1;
        }
        __syncthreads();
        if (row + (int)gridDim.x < nrows) gload(row + gridDim.x);
#pragma unroll
        for (int ks = 0; ks < C1_PX / 32; ++ks)
            acc = mma16(tr_frag(dimg, C1_PITCH, 32 * ks + 8 * g, 16 * mt, r),
                        tr_frag(pimg, C1_PITCH, 32 * ks + 8 * g, 16 * nt, r), acc);
        __syncthreads();
    }
    float* out = slabs + (long)blockIdx.x * 32 * 32;         // rows co = 16 mt + 4g + i, column k = 16 nt + r
#pragma unroll
    for (int i = 0; i < 4; ++i) out[(16 * mt + 4 * g + i) * 32 + 16 * nt + r] = acc[i];
}

inline int wave_grid(int ngroups, int per_cu) {
    const int need = (ngroups + 3) / 4;
    const int cap = 256 * per_cu;
    return need < cap ? (need > 0 ? need : 1) : cap;
}

}  // namespace

// conv1 forward: x float [frames][3][S][S], w float [32][3][3][3] (conv1.weight as stored) -> u1 [frames*Ho*Wo][32]
extern "C" int istvt_conv1_fwd(const float* x, const float* w, void* u1, int Fr, int S, int dtype, hipStream_t stream) {
    if (Fr <= 0 || S < 3) return ISTVT_ERR_SHAPE;
    const int Ho = (S - 3) / 2 + 1;
    const long Mo = (long)Fr * Ho * Ho;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((conv1_fwd_kernel<T>), dim3((unsigned)((Mo + 255) / 256)), dim3(256), 0,
                                             stream, x, w, (T*)u1, Mo, S, Ho, Ho));
    return istvt_check_launch();
}

// conv2 forward (bf16): u1 [frames][H][W][32] raw conv1 output, bnp = bn1's pack (relu applied), w [64][(dy,dx,ci)]
extern "C" int istvt_conv2_fwd(const void* u1, const float* bnp, const void* w, void* u2, int Fr, int H, int W,
                               hipStream_t stream) {
    if (Fr <= 0 || H < 3 || W < 3 || !bnp) return ISTVT_ERR_SHAPE;
    const int Ho = H - 2, Wo = W - 2;
    const long Mo = (long)Fr * Ho * Wo;
    if (Mo / 16 + 1 > 0x7fffffffL) return ISTVT_ERR_SHAPE;
    const int ngroups = (int)((Mo + 15) / 16);
    const long items = (long)Fr * ((Wo + 15) / 16) * ((Ho + 8) / 9);
    hipLaunchKernelGGL(conv2_fwd_kernel, dim3(wave_grid((int)(items < 0x7fffffffL ? items : 0x7fffffffL), 2)), dim3(256), 0, stream, (const bf16_t*)u1, bnp,
                       (const bf16_t*)w, (bf16_t*)u2, Mo, H, W, Ho, Wo, ngroups);
    return istvt_check_launch();
}

// conv2 input gradient (bf16): du2 [frames][H-2][W-2][64] -> dz1 [frames][H][W][32], masked by relu'(bn1(u1))
extern "C" int istvt_conv2_dgrad(const void* du2, const void* w, const void* u1, const float* bnp, void* dz1, int Fr,
                                 int H, int W, hipStream_t stream) {
    if (Fr <= 0 || H < 3 || W < 3 || !bnp) return ISTVT_ERR_SHAPE;
    const long Mi = (long)Fr * H * W;
    if (Mi / 16 + 1 > 0x7fffffffL) return ISTVT_ERR_SHAPE;
    const int ngroups = (int)((Mi + 15) / 16);
    const long items = (long)Fr * ((W + 15) / 16) * ((H + 8) / 9);
    hipLaunchKernelGGL(conv2_dgrad_kernel, dim3(wave_grid((int)(items < 0x7fffffffL ? items : 0x7fffffffL), 2)), dim3(256), 0, stream, (const bf16_t*)du2,
                       (const bf16_t*)w, (const bf16_t*)u1, bnp, (bf16_t*)dz1, Mi, H, W, H - 2, W - 2, ngroups);
    return istvt_check_launch();
}

// conv2 weight gradient (bf16): dw float [64][(dy,dx,ci)] += sum over pixels; slabs = float workspace of
// istvt_conv2_wgrad_slabs() * 64 * 288 elements owned by the caller
extern "C" int istvt_conv2_wgrad_slabs() { return 768; }

extern "C" int istvt_conv2_wgrad(const void* du2, const void* u1, const float* bnp, float* slabs, float* dw, int Fr,
                                 int H, int W, hipStream_t stream) {
    if (Fr <= 0 || H < 3 || W < 3 || !bnp || !slabs) return ISTVT_ERR_SHAPE;
    const int Ho = H - 2, Wo = W - 2;
    const long nch = (long)Fr * Ho * ((Wo + WG_PX - 1) / WG_PX);
    if (nch > 0x7fffffffL) return ISTVT_ERR_SHAPE;
    const int cap = istvt_conv2_wgrad_slabs();
    const int grid = nch < cap ? (int)nch : cap;
    hipLaunchKernelGGL(conv2_wgrad_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)du2, (const bf16_t*)u1, bnp,
                       slabs, Fr, H, W, Ho, Wo, (int)nch);
    int rc = istvt_check_launch();
    if (rc != ISTVT_OK) return rc;
    const int n = 64 * 288;
    return istvt_rows_reduce_add(slabs, grid, 1, n, dw, nullptr, nullptr, stream);       // slabs in index order, no atomics
}

// conv1 weight gradient: du1 [frames*Ho*Wo][32] (dtype), x float [frames][3][S][S] -> dw float [32][32] +=, column
// k = ci*9 + dy*3 + dx (the order of conv1.weight[co]; columns 27..31 stay untouched zeros of the patches);
// slabs = caller-owned float workspace of istvt_conv1_wgrad_slabs() * 1024 elements.  Needs Wo <= 128 (S <= 257).
extern "C" int istvt_conv1_wgrad_slabs() { return 1024; }

extern "C" int istvt_conv1_wgrad(const void* du1, const float* x, float* slabs, float* dw, int Fr, int S, int dtype,
                                 hipStream_t stream) {
    if (Fr <= 0 || S < 3 || !slabs) return ISTVT_ERR_SHAPE;
    const int Ho = (S - 3) / 2 + 1;
    if (Ho > C1_PX) return ISTVT_ERR_SHAPE;
    const long nrows = (long)Fr * Ho;
    if (nrows > 0x7fffffffL) return ISTVT_ERR_SHAPE;
    const int cap = istvt_conv1_wgrad_slabs();
    const int grid = nrows < cap ? (int)nrows : cap;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((conv1_wgrad_kernel<T>), dim3(grid), dim3(256), 0, stream, (const T*)du1, x,
                                             slabs, S, Ho, Ho, (int)nrows));
    int rc = istvt_check_launch();
    if (rc != ISTVT_OK) return rc;
    return istvt_rows_reduce_add(slabs, grid, 1, 1024, dw, nullptr, nullptr, stream);
}
