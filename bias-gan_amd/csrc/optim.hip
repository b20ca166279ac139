// Adam / AdamW over a flat fp32 arena with the bf16 weight copy fused in
// (one read-modify-write pass over parameters, moments and gradients).
#include "common.h"

namespace {

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, bf16_t* __restrict__ p_lp, long long n, float lr, float b1, float b2,
                            float omb1, float omb2, float eps, float wd, int decoupled, float bc1, float bc2, float gscale) {
    // torch.optim.Adam: g += wd*p; m = b1*m+(1-b1)g; v = b2*v+(1-b2)g^2;
    //                   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
    const float step = lr / bc1;
    const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float pv = p[i];
        float gv = g[i] * gscale;
        if (decoupled) pv *= (1.f - lr * wd);
        else gv = fmaf(wd, pv, gv);
        const float mv = b1 * m[i] + omb1 * gv;
        const float vv = b2 * v[i] + omb2 * gv * gv;
        m[i] = mv;
        v[i] = vv;
        pv -= step * mv / (sqrtf(vv) * inv_sqrt_bc2 + eps);
        p[i] = pv;
        if (p_lp) p_lp[i] = (bf16_t)pv;
    }
}

// The same update with its step-dependent scalars read from device memory: hyper = {lr, bias_corr1, bias_corr2,
// grad_scale}.  A captured (hipGraph) training step replays this launch every step; the host rewrites the four floats
// before each replay instead of baking them into the kernel arguments.
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, bf16_t* __restrict__ p_lp, long long n, const float* __restrict__ hyper,
                                float b1, float b2, float omb1, float omb2, float eps, float wd, int decoupled) {
    const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2], gscale = hyper[3];
    const float step = lr / bc1;
    const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float pv = p[i];
        float gv = g[i] * gscale;
        if (decoupled) pv *= (1.f - lr * wd);
        else gv = fmaf(wd, pv, gv);
        const float mv = b1 * m[i] + omb1 * gv;
        const float vv = b2 * v[i] + omb2 * gv * gv;
        m[i] = mv;
        v[i] = vv;
        pv -= step * mv / (sqrtf(vv) * inv_sqrt_bc2 + eps);
        p[i] = pv;
        if (p_lp) p_lp[i] = (bf16_t)pv;
    }
}

__global__ void cast_bf16_kernel(const float* src, bf16_t* dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = (bf16_t)src[i];
}

inline unsigned ew_grid(long long total) {
    long long g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int bg_adam_step(float* p, const float* g, float* m, float* v, void* p_lp, int64_t n, float lr, double beta1,
                            double beta2, float eps, float weight_decay, int32_t decoupled, float bias_corr1,
                            float bias_corr2, float grad_scale, void* stream) {
    BG_CHECK_ARG(p && g && m && v && n > 0 && bias_corr1 > 0.f && bias_corr2 > 0.f, "bg_adam_step: bad args");
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_lp,
                       (long long)n, lr, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, weight_decay,
                       decoupled, bias_corr1, bias_corr2, grad_scale);
    BG_CHECK_LAUNCH("adam_kernel");
    return BG_OK;
}

__global__ void set_floats_kernel(float* dst, int n, float v0, float v1, float v2, float v3) {
    const float v[4] = {v0, v1, v2, v3};
    if (threadIdx.x < n) dst[threadIdx.x] = v[threadIdx.x];
}

extern "C" int bg_set_floats(float* dst, int32_t n, float v0, float v1, float v2, float v3, void* stream) {
    BG_CHECK_ARG(dst && n >= 1 && n <= 4, "bg_set_floats: 1 to 4 values");
    hipLaunchKernelGGL(set_floats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, n, v0, v1, v2, v3);
    BG_CHECK_LAUNCH("set_floats_kernel");
    return BG_OK;
}

extern "C" int bg_adam_step_dev(float* p, const float* g, float* m, float* v, void* p_lp, int64_t n, const float* hyper,
                                double beta1, double beta2, float eps, float weight_decay, int32_t decoupled, void* stream) {
    BG_CHECK_ARG(p && g && m && v && hyper && n > 0, "bg_adam_step_dev: bad args");
    hipLaunchKernelGGL(adam_dev_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_lp,
                       (long long)n, hyper, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps,
                       weight_decay, decoupled);
    BG_CHECK_LAUNCH("adam_dev_kernel");
    return BG_OK;
}

extern "C" int bg_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    BG_CHECK_ARG(src && dst && n > 0, "bg_cast_f32_to_bf16: bad args");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst,
                       (long long)n);
    BG_CHECK_LAUNCH("cast_bf16_kernel");
    return BG_OK;
}
