// Discriminator head (Linear(2048*h*w, 1)) and the scalar losses.
// Reductions: per-thread fp32 partials -> wavefront shuffle tree -> one float
// atomic per wave.
#include "common.h"

namespace {

// logits[n] += sum_{p,c} x[n,p,c] * w[c*HW + p].  Lanes run along pixels so the
// weight reads (reference layout, c-major) are coalesced; a lane's activation
// reads walk its own pixel row 16 B at a time.
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* x, int ldx, const float* w, float* logits, int HW,
                                                       int C) {
    constexpr int VEC = Elem<T>::VEC;
    const int n = blockIdx.y;
    const int p = blockIdx.x * 64 + (threadIdx.x & 63);
    const int cpart = threadIdx.x >> 6;  // 4 waves split the channel range
    float acc = 0.f;
    if (p < HW) {
        const T* xr = x + ((long long)n * HW + p) * ldx;
        for (int c = cpart * VEC; c < C; c += 4 * VEC) {
            Chunk<T> v;
            v.load(xr + c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc = fmaf(v.get(e), w[(long long)(c + e) * HW + p], acc);
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(logits + n, acc);
}

__global__ void head_bias_kernel(float* logits, const float* b, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) logits[i] = b ? b[0] : 0.f;
}

// dx[n,p,c] = dl[n]*w[c*HW+p];  dw[c*HW+p] += sum_n dl[n]*x[n,p,c]
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* x, int ldx, const float* w, const float* dl, T* dx,
                                                       int lddx, float* dw, int N, int HW, int C) {
    constexpr int VEC = Elem<T>::VEC;
    const int p = blockIdx.x * 64 + (threadIdx.x & 63);
    const int cpart = threadIdx.x >> 6;
    if (p >= HW) return;
    for (int c = (blockIdx.y * 4 + cpart) * VEC; c < C; c += gridDim.y * 4 * VEC) {
        float wv[VEC], gw[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            wv[e] = w[(long long)(c + e) * HW + p];
            gw[e] = 0.f;
        }
        for (int n = 0; n < N; ++n) {
            const float g = dl[n];
            if (dw) {
                Chunk<T> v;
                v.load(x + ((long long)n * HW + p) * ldx + c);
#pragma unroll
                for (int e = 0; e < VEC; ++e) gw[e] = fmaf(g, v.get(e), gw[e]);
            }
            if (dx) {
                Chunk<T> o;
#pragma unroll
                for (int e = 0; e < VEC; ++e) o.set(e, g * wv[e]);
                o.store(dx + ((long long)n * HW + p) * lddx + c);
            }
        }
        if (dw) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) dw[(long long)(c + e) * HW + p] += gw[e];  // (c,p) owned by this thread
        }
    }
}

__global__ void head_db_kernel(const float* dl, float* db, int N) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < N; ++i) s += dl[i];
        db[0] += s;
    }
}

__global__ void bce_logits_kernel(const float* x, const float* y, int n, float* loss, float* dx) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = x[i], t = y[i];
        acc += fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));
        if (dx) dx[i] = (1.f / (1.f + expf(-v)) - t) / (float)n;
    }
    acc = wave_sum(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (part[0] + part[1] + part[2] + part[3]) / (float)n;
}

// kind: 0 = |d| (nn.L1Loss), 1 = SmoothL1 with beta = 1 (0.5 d^2 for |d| < 1, |d| - 0.5 beyond), 2 = d^2 (nn.MSELoss)
__device__ __forceinline__ float pix_loss(int kind, float d) {
    const float a = fabsf(d);
    return kind == 0 ? a : (kind == 1 ? (a < 1.f ? 0.5f * d * d : a - 0.5f) : d * d);
}
__device__ __forceinline__ float pix_loss_grad(int kind, float d) {
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    return kind == 0 ? sgn : (kind == 1 ? (fabsf(d) < 1.f ? d : sgn) : 2.f * d);
}

// One atomic per BLOCK: 8 192 per-wave atomics onto the one loss word serialised at ~12 ns each -- 100 us of the 117 us this
// kernel took on the 256 x 256 fields (67 MB) and of its 266 us at 1152 x 768; 16-byte loads where the tensors allow.
__global__ __launch_bounds__(256) void l1_fwd_kernel(int kind, const float* p, const float* t, const float* w, long long n, float inv_norm,
                                                     float* loss) {
    float acc = 0.f;
    const long long tid0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
    const bool v4 = ((((uintptr_t)p | (uintptr_t)t | (uintptr_t)w) & 15) == 0);
    const long long n4 = v4 ? n / 4 : 0;
    for (long long i = tid0; i < n4; i += nth) {
        const f32x4 a = reinterpret_cast<const f32x4*>(p)[i], b = reinterpret_cast<const f32x4*>(t)[i];
        f32x4 ww = {1.f, 1.f, 1.f, 1.f};
        if (w) ww = reinterpret_cast<const f32x4*>(w)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = fmaf(pix_loss(kind, a[e] - b[e]), ww[e], acc);
    }
    for (long long i = n4 * 4 + tid0; i < n; i += nth) {
        const float a = pix_loss(kind, p[i] - t[i]);
        acc += w ? a * w[i] : a;
    }
    acc = wave_sum(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (part[0] + part[1] + part[2] + part[3]) * inv_norm);
}

__global__ void l1_bwd_kernel(int kind, const float* p, const float* t, const float* w, long long n, float inv_norm,
                              const float* coef, float* dp) {
    const float k = coef[0] * inv_norm;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dp[i] = pix_loss_grad(kind, p[i] - t[i]) * (w ? w[i] : 1.f) * k;
}

__global__ void gp_kernel(const float* g, int N, int C, int HW, float inv_norm, float* loss) {
    const long long total = (long long)N * HW;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / HW);
        const int p = (int)(i - (long long)n * HW);
        const float* b = g + (long long)n * C * HW + p;
        float s = 0.f;
        for (int c = 0; c < C; ++c) {
            const float v = b[(long long)c * HW];
            s = fmaf(v, v, s);
        }
        const float d = sqrtf(s) - 1.f;
        acc = fmaf(d, d, acc);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc * inv_norm);
}

inline unsigned red_grid(long long total) {
    long long g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int bg_linear_head_fwd(int32_t dtype, const void* x, int32_t ldx, const float* wlin, const float* blin,
                                  float* logits, int32_t N, int32_t HW, int32_t C, void* stream) {
    BG_CHECK_ARG(dtype_ok(dtype) && x && wlin && logits && aligned16(x) && N > 0 && HW > 0 && C > 0,
                 "bg_linear_head_fwd: bad args");
    BG_CHECK_ARG(C % dtype_vec(dtype) == 0 && ldx >= C && ldx % dtype_vec(dtype) == 0 && N <= 65535,
                 "bg_linear_head_fwd: bad C/ld/N");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(head_bias_kernel, dim3((N + 63) / 64), dim3(64), 0, st, logits, blin, N);
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((head_fwd_kernel<T>), dim3((HW + 63) / 64, N), dim3(256), 0, st,
                                                   (const T*)x, ldx, wlin, logits, HW, C));
    BG_CHECK_LAUNCH("head_fwd_kernel");
    return BG_OK;
}

extern "C" int bg_linear_head_bwd(int32_t dtype, const void* x, int32_t ldx, const float* wlin, const float* dlogits,
                                  void* dx, int32_t lddx, float* dw, float* db, int32_t N, int32_t HW, int32_t C,
                                  void* stream) {
    BG_CHECK_ARG(dtype_ok(dtype) && wlin && dlogits && N > 0 && HW > 0 && C > 0 && (dx || dw || db),
                 "bg_linear_head_bwd: bad args");
    const int vec = dtype_vec(dtype);
    BG_CHECK_ARG(C % vec == 0, "bg_linear_head_bwd: bad C");
    if (dw) BG_CHECK_ARG(x && aligned16(x) && ldx >= C && ldx % vec == 0, "bg_linear_head_bwd: dw needs x");
    if (dx) BG_CHECK_ARG(aligned16(dx) && lddx >= C && lddx % vec == 0, "bg_linear_head_bwd: bad dx");
    hipStream_t st = (hipStream_t)stream;
    if (dx || dw) {
        int gy = C / (4 * vec);
        if (gy < 1) gy = 1;
        if (gy > 16) gy = 16;
        BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((head_bwd_kernel<T>), dim3((HW + 63) / 64, gy), dim3(256), 0, st,
                                                       (const T*)x, ldx, wlin, dlogits, (T*)dx, lddx, dw, N, HW, C));
        BG_CHECK_LAUNCH("head_bwd_kernel");
    }
    if (db) {
        hipLaunchKernelGGL(head_db_kernel, dim3(1), dim3(64), 0, st, dlogits, db, N);
        BG_CHECK_LAUNCH("head_db_kernel");
    }
    return BG_OK;
}

extern "C" int bg_bce_logits(const float* x, const float* y, int32_t n, float* loss, float* dx, void* stream) {
    BG_CHECK_ARG(x && y && loss && n > 0, "bg_bce_logits: bad args");
    hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, y, n, loss, dx);
    BG_CHECK_LAUNCH("bce_logits_kernel");
    return BG_OK;
}

extern "C" int bg_pixel_loss_fwd(int32_t kind, const float* p, const float* t, const float* w, int64_t n,
                                 float inv_norm, float* loss, void* stream) {
    BG_CHECK_ARG(p && t && loss && n > 0 && kind >= 0 && kind <= 2, "bg_pixel_loss_fwd: bad args");
    hipLaunchKernelGGL(l1_fwd_kernel, dim3(red_grid(n)), dim3(256), 0, (hipStream_t)stream, kind, p, t, w, (long long)n,
                       inv_norm, loss);
    BG_CHECK_LAUNCH("l1_fwd_kernel");
    return BG_OK;
}

extern "C" int bg_pixel_loss_bwd(int32_t kind, const float* p, const float* t, const float* w, int64_t n,
                                 float inv_norm, const float* coef, float* dp, void* stream) {
    BG_CHECK_ARG(p && t && coef && dp && n > 0 && kind >= 0 && kind <= 2, "bg_pixel_loss_bwd: bad args");
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(red_grid(n) * 4), dim3(256), 0, (hipStream_t)stream, kind, p, t, w,
                       (long long)n, inv_norm, coef, dp);
    BG_CHECK_LAUNCH("l1_bwd_kernel");
    return BG_OK;
}

extern "C" int bg_l1_loss_fwd(const float* p, const float* t, const float* w, int64_t n, float inv_norm, float* loss,
                              void* stream) {
    return bg_pixel_loss_fwd(0, p, t, w, n, inv_norm, loss, stream);
}

extern "C" int bg_l1_loss_bwd(const float* p, const float* t, const float* w, int64_t n, float inv_norm,
                              const float* coef, float* dp, void* stream) {
    return bg_pixel_loss_bwd(0, p, t, w, n, inv_norm, coef, dp, stream);
}

extern "C" int bg_gp_penalty(const float* g, int32_t N, int32_t C, int32_t HW, float inv_norm, float* loss,
                             void* stream) {
    BG_CHECK_ARG(g && loss && N > 0 && C > 0 && HW > 0, "bg_gp_penalty: bad args");
    hipLaunchKernelGGL(gp_kernel, dim3(red_grid((long long)N * HW)), dim3(256), 0, (hipStream_t)stream, g, N, C, HW,
                       inv_norm, loss);
    BG_CHECK_LAUNCH("gp_kernel");
    return BG_OK;
}
