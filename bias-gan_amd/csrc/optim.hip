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

// ---- LAMB (apex.optimizers.FusedLAMB as the reference selects it, utils/parsing_helpers.py:13-14) over a flat arena cut
// into parameter tensors [seg[t], seg[t + 1]): the published two-stage form (apex csrc/multi_tensor_lamb.cu).
// Stage 1 (per element, adam_w_mode = 1): s = g / clip, clip = max(1, ||g||_global / max_grad_norm);
//   m = b1 m + b3 s;  v = b2 v + (1 - b2) s^2;  u = (m / bc1) / (sqrt(v / bc2) + eps) + wd p   (mode 0: s += wd p, no wd term in u)
//   u overwrites g; per tensor ||p||^2 and ||u||^2 are accumulated.
// Stage 2 (per tensor): ratio = lr * ||p|| / ||u|| where wd != 0 (or use_nvlamb) and both norms are non-zero, else lr;  p -= ratio u.
constexpr int LAMB_GX = 64;   // blocks along a tensor's elements; blockIdx.y = tensor

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, float scale, double* out) {
    float a = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i] * scale;
        a = fmaf(v, v, a);
    }
    a = wave_sum(a);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (double)part[0] + (double)part[1] + (double)part[2] + (double)part[3]);
}

__global__ __launch_bounds__(256) void lamb_stage1_kernel(const float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, const long long* __restrict__ seg,
                                                          const double* __restrict__ gsq, float max_norm, float b1, float b2,
                                                          float b3, float eps, float wd, int adam_w, float bc1, float bc2,
                                                          float gscale, float* __restrict__ pn, float* __restrict__ un) {
    const int t = blockIdx.y;
    const long long lo = seg[t], hi = seg[t + 1];
    const float gn = sqrtf((float)gsq[0]);
    const float clip = (max_norm > 0.f && gn > max_norm) ? gn / max_norm : 1.f;
    float ap = 0.f, au = 0.f;
    for (long long i = lo + (long long)blockIdx.x * 256 + threadIdx.x; i < hi; i += (long long)gridDim.x * 256) {
        const float pv = p[i];
        float sg = g[i] * gscale / clip;
        if (!adam_w) sg = fmaf(wd, pv, sg);
        const float mv = b1 * m[i] + b3 * sg;
        const float vv = b2 * v[i] + (1.f - b2) * sg * sg;
        m[i] = mv;
        v[i] = vv;
        float u = (mv / bc1) / (sqrtf(vv / bc2) + eps);
        if (adam_w) u = fmaf(wd, pv, u);
        g[i] = u;
        ap = fmaf(pv, pv, ap);
        au = fmaf(u, u, au);
    }
    ap = wave_sum(ap);
    au = wave_sum(au);
    __shared__ float part[2][4];
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = ap; part[1][threadIdx.x >> 6] = au; }
    __syncthreads();
    if (threadIdx.x == 0 && lo + (long long)blockIdx.x * 256 < hi) {
        atomicAdd(pn + t, part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        atomicAdd(un + t, part[1][0] + part[1][1] + part[1][2] + part[1][3]);
    }
}

__global__ __launch_bounds__(256) void lamb_stage2_kernel(float* __restrict__ p, const float* __restrict__ u, bf16_t* __restrict__ p_lp,
                                                          const long long* __restrict__ seg, const float* __restrict__ pn,
                                                          const float* __restrict__ un, float lr, float wd, int use_nvlamb) {
    const int t = blockIdx.y;
    const long long lo = seg[t], hi = seg[t + 1];
    float ratio = lr;
    if (use_nvlamb || wd != 0.f) {
        const float a = sqrtf(pn[t]), b = sqrtf(un[t]);
        if (a != 0.f && b != 0.f) ratio = lr * (a / b);
    }
    for (long long i = lo + (long long)blockIdx.x * 256 + threadIdx.x; i < hi; i += (long long)gridDim.x * 256) {
        const float pv = p[i] - ratio * u[i];
        p[i] = pv;
        if (p_lp) p_lp[i] = (bf16_t)pv;
    }
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

extern "C" int bg_sumsq_f32(const float* x, int64_t n, float scale, double* out, void* stream) {
    BG_CHECK_ARG(x && out && n > 0, "bg_sumsq_f32: bad args");
    hipLaunchKernelGGL(sumsq_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, scale, out);
    BG_CHECK_LAUNCH("sumsq_kernel");
    return BG_OK;
}

extern "C" int bg_lamb_stage1(const float* p, float* g, float* m, float* v, const int64_t* seg, int32_t nseg, const double* grad_sumsq,
                              float max_grad_norm, double beta1, double beta2, int32_t grad_averaging, float eps, float weight_decay,
                              int32_t adam_w_mode, float bias_corr1, float bias_corr2, float grad_scale, float* param_sumsq,
                              float* update_sumsq, void* stream) {
    BG_CHECK_ARG(p && g && m && v && seg && grad_sumsq && param_sumsq && update_sumsq && nseg > 0 && nseg <= 65535 && bias_corr1 > 0.f &&
                     bias_corr2 > 0.f, "bg_lamb_stage1: bad args (at most 65535 tensors per call)");
    hipLaunchKernelGGL(lamb_stage1_kernel, dim3(LAMB_GX, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (const long long*)seg, grad_sumsq, max_grad_norm, (float)beta1, (float)beta2,
                       grad_averaging ? 1.f - (float)beta1 : 1.f, eps, weight_decay, adam_w_mode, bias_corr1, bias_corr2, grad_scale,
                       param_sumsq, update_sumsq);
    BG_CHECK_LAUNCH("lamb_stage1_kernel");
    return BG_OK;
}

extern "C" int bg_lamb_stage2(float* p, const float* update, void* p_lp, const int64_t* seg, int32_t nseg, const float* param_sumsq,
                              const float* update_sumsq, float lr, float weight_decay, int32_t use_nvlamb, void* stream) {
    BG_CHECK_ARG(p && update && seg && param_sumsq && update_sumsq && nseg > 0 && nseg <= 65535, "bg_lamb_stage2: bad args");
    hipLaunchKernelGGL(lamb_stage2_kernel, dim3(LAMB_GX, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, p, update, (bf16_t*)p_lp,
                       (const long long*)seg, param_sumsq, update_sumsq, lr, weight_decay, use_nvlamb);
    BG_CHECK_LAUNCH("lamb_stage2_kernel");
    return BG_OK;
}
