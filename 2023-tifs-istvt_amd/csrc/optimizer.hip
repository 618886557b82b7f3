// Fused optimizer steps over the flat parameter / gradient buffers of parallel.GradBucket
// (reference: train_CNN.py:196-201 -- torch.optim.SGD(lr, momentum=0.9, weight_decay=0) or AdamW(betas, eps)).
// One pass over {p, g, state}: reads g once and (optionally) writes zeros back, so the separate zero-grad pass of the
// next step disappears.  HBM-bound: SGD-momentum moves 5 x 4 bytes per parameter (p r/w, buf r/w, g r) + the zero write.
#include "common.h"

namespace {

// torch.optim.SGD semantics: g' = g + wd * p;  first step buf = g', later buf = mu * buf + (1 - dampening) * g';
// d = nesterov ? g' + mu * buf : buf;  p -= lr * d
__global__ __launch_bounds__(256) void sgd_momentum_kernel(float* __restrict__ p, float* __restrict__ g,
                                                           float* __restrict__ buf, long n, float lr, float mu,
                                                           float dampening, float wd, int nesterov, int first,
                                                           int zero_grad, float gscale) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            float4 pv = *reinterpret_cast<float4*>(p + i), gv = *reinterpret_cast<float4*>(g + i);
            float4 bv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<float4*>(buf + i);
            float* pp = &pv.x; float* gg = &gv.x; float* bb = &bv.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gj = gg[j] * gscale + wd * pp[j];
                bb[j] = first ? gj : mu * bb[j] + (1.f - dampening) * gj;
                pp[j] -= lr * (nesterov ? gj + mu * bb[j] : bb[j]);
            }
            *reinterpret_cast<float4*>(p + i) = pv;
            *reinterpret_cast<float4*>(buf + i) = bv;
            if (zero_grad) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (long k = i; k < n; ++k) {
                const float gj = g[k] * gscale + wd * p[k];
                const float b = first ? gj : mu * buf[k] + (1.f - dampening) * gj;
                buf[k] = b;
                p[k] -= lr * (nesterov ? gj + mu * b : b);
                if (zero_grad) g[k] = 0.f;
            }
        }
    }
}

// torch.optim.AdamW semantics (amsgrad off): p *= 1 - lr * wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
// p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float b1, float b2,
                                                    float eps, float wd, float step_size, float inv_sqrt_bc2,
                                                    int zero_grad, float gscale) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gj = g[i] * gscale;
        float pj = p[i] * (1.f - lr * wd);
        const float mj = b1 * m[i] + (1.f - b1) * gj;
        const float vj = b2 * v[i] + (1.f - b2) * gj * gj;
        pj -= step_size * mj / (sqrtf(vj) * inv_sqrt_bc2 + eps);
        p[i] = pj; m[i] = mj; v[i] = vj;
        if (zero_grad) g[i] = 0.f;
    }
}

inline int opt_grid(long n, int per_thread) {
    long b = (n + 256L * per_thread - 1) / (256L * per_thread);
    if (b > 65536) b = 65536;
    return b < 1 ? 1 : (int)b;
}

}  // namespace

extern "C" int istvt_sgd_momentum(float* p, float* g, float* buf, long n, float lr, float momentum, float dampening,
                                  float weight_decay, int nesterov, int first_step, int zero_grad, float grad_scale,
                                  hipStream_t stream) {
    if (n <= 0 || !p || !g || !buf) return ISTVT_ERR_SHAPE;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) return ISTVT_ERR_SHAPE;
    if (momentum == 0.f) dampening = 0.f;          // torch.optim.SGD ignores dampening without momentum (d_p = g')
    hipLaunchKernelGGL(sgd_momentum_kernel, dim3(opt_grid(n, 4)), dim3(256), 0, stream, p, g, buf, n, lr, momentum,
                       dampening, weight_decay, nesterov, first_step, zero_grad, grad_scale);
    return istvt_check_launch();
}

extern "C" int istvt_adamw(float* p, float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                           float weight_decay, long step, int zero_grad, float grad_scale, hipStream_t stream) {
    if (n <= 0 || step < 1 || !p || !g || !m || !v) return ISTVT_ERR_SHAPE;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(opt_grid(n, 1)), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), zero_grad, grad_scale);
    return istvt_check_launch();
}
