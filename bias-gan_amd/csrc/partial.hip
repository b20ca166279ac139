// Kernels of the partial-convolution U-Net GAN (reference:
// architecture/common/partialconv3d.py, architecture/gpsro/infill3d.py).  Volumes are folded NHWC tensors
// [N*D, H, W, C] as in volume.hip; masks are tensors of the same layout holding exact 0/1 values.
//   * bg_mask_window: the mask half of PartialConv3d (partialconv3d.py:49-75): window sum of the mask over all
//     input channels (the "convolution with all-ones weights"), clamp to {0,1}, ratio = winsize/(sum+eps)*clamp.
//     Integer-valued sums in fp32: exact.  Every update_mask has Cout identical channels, so past the first layer
//     a mask is held as ONE fp32 value per pixel standing for c channels (up to two such segments: the U-Net's
//     concatenations) -- C times less mask traffic than the reference's Cout-channel mask tensors.
//   * bg_mul_rows / bg_scale_rows: input*mask before the convolution, raw_out*mask_ratio (+ bias*update_mask)
//     after it; the same kernels are their own adjoints (mask and ratio are constants).
//   * bg_resize_nearest3d_fwd/_bwd: F.interpolate(mode='nearest') to an explicit size (infill3d.py:217-222).
//   * bg_tv_loss_fwd/_bwd: utils/losses.py:40-44 as it acts on a 5-D tensor (shifts along dims 3 and 2).
#include "common.h"

namespace {

inline unsigned grid1d(long long total) {
    long long g = (total + 255) / 256;
    if (g > 0x7fffffffLL) g = 0x7fffffffLL;
    if (g < 1) g = 1;
    return (unsigned)g;
}

struct MaskWinParams {
    const void* m; int ld, C;                       // per-channel mask (may be null)
    const float* r0; float c0;                      // per-pixel masks standing for c0 / c1 identical channels
    const float* r1; float c1;
    int N, D, H, W, Do, Ho, Wo, k, stride, pad;
    int kd, sd, pd;                                 // depth window (1, 1, 0 for the 2-D PartialConv2d)
    float winsize, eps;
    float* upd; float* ratio;
};

template <typename T>
__global__ __launch_bounds__(256) void mask_window_kernel(MaskWinParams P) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = (P.C + VEC - 1) / VEC;   // C counts the real channels; pad lanes of the last chunk are skipped
    const long long total = (long long)P.N * P.Do * P.Ho * P.Wo;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long r = i;
        const int ow = (int)(r % P.Wo); r /= P.Wo;
        const int oh = (int)(r % P.Ho); r /= P.Ho;
        const int od = (int)(r % P.Do);
        const int n = (int)(r / P.Do);
        const long long vol = (long long)n * P.D * P.H * P.W;
        const T* base = reinterpret_cast<const T*>(P.m) + vol * P.ld;
        float s = 0.f, s0 = 0.f, s1 = 0.f;
        for (int kd = 0; kd < P.kd; ++kd) {
            const int id = od * P.sd - P.pd + kd;
            if ((unsigned)id >= (unsigned)P.D) continue;
            for (int kh = 0; kh < P.k; ++kh) {
                const int ih = oh * P.stride - P.pad + kh;
                if ((unsigned)ih >= (unsigned)P.H) continue;
                for (int kw = 0; kw < P.k; ++kw) {
                    const int iw = ow * P.stride - P.pad + kw;
                    if ((unsigned)iw >= (unsigned)P.W) continue;
                    const long long pix = ((long long)id * P.H + ih) * P.W + iw;
                    if (P.r0) s0 += P.r0[vol + pix];
                    if (P.r1) s1 += P.r1[vol + pix];
                    if (!P.m) continue;
                    const T* p = base + pix * P.ld;
                    for (int c = 0; c < cv; ++c) {
                        Chunk<T> v;
                        v.load(p + c * VEC);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) s += (c * VEC + e < P.C) ? v.get(e) : 0.f;
                    }
                }
            }
        }
        s += P.c0 * s0 + P.c1 * s1;     // integer-valued and < 2^24: exact in any order
        const float u = fminf(fmaxf(s, 0.f), 1.f);
        P.upd[i] = u;
        P.ratio[i] = P.winsize / (s + P.eps) * u;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void mul_rows_kernel(const T* x, int ldx, const T* m, int ldm, T* y, int ldy, long long rows, int C) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const long long total = rows * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cv;
        const int c = (int)(i - r * cv) * VEC;
        Chunk<T> a, b, o;
        a.load(x + r * ldx + c);
        b.load(m + r * ldm + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, a.get(e) * b.get(e));
        o.store(y + r * ldy + c);
    }
}

// y[r,c] = (x ? x[r,c] : 1) * s[r] + (bias ? bias[c] * t[r] : 0)
template <typename T>
__global__ __launch_bounds__(256) void scale_rows_kernel(const T* x, int ldx, const float* s, const float* bias, const float* t,
                                                         T* y, int ldy, long long rows, int C) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const long long total = rows * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cv;
        const int c = (int)(i - r * cv) * VEC;
        const float sr = s[r], tr = bias ? t[r] : 0.f;
        Chunk<T> a, o;
        if (x) a.load(x + r * ldx + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float v = (x ? a.get(e) : 1.f) * sr;
            if (bias) v = fmaf(bias[c + e], tr, v);
            o.set(e, v);
        }
        o.store(y + r * ldy + c);
    }
}

struct NearestParams {
    const void* x; void* y;
    int N, Di, Hi, Wi, Do, Ho, Wo, C, ldx, ldy;
};

__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
    const int s = (int)(((long long)dst * in) / out);   // floor(dst * in / out): torch's 'nearest' for these sizes
    return s < in - 1 ? s : in - 1;
}

template <typename T>
__global__ __launch_bounds__(256) void nearest3d_fwd_kernel(NearestParams P) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = P.C / VEC;
    const long long total = (long long)P.N * P.Do * P.Ho * P.Wo * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / cv;
        const int c = (int)(i - pix * cv) * VEC;
        long long r = pix;
        const int ow = (int)(r % P.Wo); r /= P.Wo;
        const int oh = (int)(r % P.Ho); r /= P.Ho;
        const int od = (int)(r % P.Do);
        const int n = (int)(r / P.Do);
        const int id = nearest_src(od, P.Di, P.Do), ih = nearest_src(oh, P.Hi, P.Ho), iw = nearest_src(ow, P.Wi, P.Wo);
        Chunk<T> v;
        v.load(reinterpret_cast<const T*>(P.x) + ((((long long)n * P.Di + id) * P.Hi + ih) * P.Wi + iw) * P.ldx + c);
        v.store(reinterpret_cast<T*>(P.y) + pix * P.ldy + c);
    }
}

__global__ __launch_bounds__(256) void nearest3d_rows_kernel(const float* x, float* y, int N, int Di, int Hi, int Wi, int Do, int Ho,
                                                             int Wo) {
    const long long total = (long long)N * Do * Ho * Wo;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long r = i;
        const int ow = (int)(r % Wo); r /= Wo;
        const int oh = (int)(r % Ho); r /= Ho;
        const int od = (int)(r % Do);
        const int n = (int)(r / Do);
        y[i] = x[(((long long)n * Di + nearest_src(od, Di, Do)) * Hi + nearest_src(oh, Hi, Ho)) * Wi + nearest_src(ow, Wi, Wo)];
    }
}

__device__ __forceinline__ void nearest_dst_range(int src, int in, int out, int& lo, int& hi) {
    // all dst with floor(dst*in/out) == src (the last source index also takes every dst that would map beyond it)
    lo = (int)(((long long)src * out + in - 1) / in);
    hi = src == in - 1 ? out : (int)(((long long)(src + 1) * out + in - 1) / in);
}

template <typename T>
__global__ __launch_bounds__(256) void nearest3d_bwd_kernel(NearestParams P) {   // x = dx (out), y = dy (in)
    constexpr int VEC = Elem<T>::VEC;
    const int cv = P.C / VEC;
    const long long total = (long long)P.N * P.Di * P.Hi * P.Wi * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / cv;
        const int c = (int)(i - pix * cv) * VEC;
        long long r = pix;
        const int iw = (int)(r % P.Wi); r /= P.Wi;
        const int ih = (int)(r % P.Hi); r /= P.Hi;
        const int id = (int)(r % P.Di);
        const int n = (int)(r / P.Di);
        int d0, d1, h0, h1, w0, w1;
        nearest_dst_range(id, P.Di, P.Do, d0, d1);
        nearest_dst_range(ih, P.Hi, P.Ho, h0, h1);
        nearest_dst_range(iw, P.Wi, P.Wo, w0, w1);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int od = d0; od < d1; ++od)
            for (int oh = h0; oh < h1; ++oh)
                for (int ow = w0; ow < w1; ++ow) {
                    Chunk<T> v;
                    v.load(reinterpret_cast<const T*>(P.y) + ((((long long)n * P.Do + od) * P.Ho + oh) * P.Wo + ow) * P.ldy + c);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += v.get(e);
                }
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
        o.store(reinterpret_cast<T*>(const_cast<void*>(P.x)) + pix * P.ldx + c);
    }
}

// total_variation_loss on a contiguous fp32 5-D tensor [A=N*C][D][H][rest=W]: mean |x[.., h+1, :] - x[.., h, :]| over
// (A, D, H-1, W) plus mean |x[.., d+1, :, :] - x[.., d, :, :]| over (A, D-1, H, W)   (dims 3 and 2 of the 5-D tensor)
__global__ __launch_bounds__(256) void tv_fwd_kernel(const float* x, long long A, int D, int H, int W, float inv_h, float inv_d,
                                                     float* loss) {
    const long long total = A * D * H * W;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int h = (int)((i / W) % H), d = (int)((i / ((long long)W * H)) % D);
        const float v = x[i];
        if (h + 1 < H) acc += fabsf(v - x[i + W]) * inv_h;
        if (d + 1 < D) acc += fabsf(v - x[i + (long long)W * H]) * inv_d;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc);
}

__global__ __launch_bounds__(256) void tv_bwd_kernel(const float* x, long long A, int D, int H, int W, float inv_h, float inv_d,
                                                     const float* coef, float* dx) {
    const long long total = A * D * H * W;
    const float k = coef[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int h = (int)((i / W) % H), d = (int)((i / ((long long)W * H)) % D);
        const float v = x[i];
        auto sgn = [](float a) { return a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f); };
        float g = 0.f;
        if (h + 1 < H) g += sgn(v - x[i + W]) * inv_h;
        if (h > 0) g -= sgn(x[i - W] - v) * inv_h;
        if (d + 1 < D) g += sgn(v - x[i + (long long)W * H]) * inv_d;
        if (d > 0) g -= sgn(x[i - (long long)W * H] - v) * inv_d;
        dx[i] = g * k;
    }
}

// y = m*a + (1-m)*b on flat fp32 arrays (output_comp of InpaintingLoss, utils/losses.py:71); with a == NULL: y = (1-m)*b,
// which is the adjoint w.r.t. b applied to an upstream gradient
__global__ __launch_bounds__(256) void blend_kernel(const float* m, const float* a, const float* b, float* y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float mm = m[i];
        y[i] = (a ? mm * a[i] : 0.f) + (1.f - mm) * b[i];
    }
}

}  // namespace

extern "C" int bg_blend_f32(const float* m, const float* a, const float* b, float* y, int64_t n, void* stream) {
    BG_CHECK_ARG(m && b && y && n > 0, "bg_blend_f32: bad args");
    hipLaunchKernelGGL(blend_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, m, a, b, y, (long long)n);
    BG_CHECK_LAUNCH("blend_kernel");
    return BG_OK;
}

extern "C" int bg_mask_window(int32_t dtype, const void* mask, int32_t ld, int32_t C, const float* rows0, int32_t c0,
                              const float* rows1, int32_t c1, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Do, int32_t Ho,
                              int32_t Wo, int32_t k, int32_t stride, int32_t pad, int32_t planar, float eps, float* update_mask,
                              float* ratio, void* stream) {
    BG_CHECK_ARG(dtype_ok(dtype) && (mask || rows0) && update_mask && ratio && (!mask || aligned16(mask)) && (rows0 || !rows1),
                 "bg_mask_window: bad dtype / pointer");
    const int vec = dtype_vec(dtype);
    if (!mask) C = 0;
    if (!rows0) c0 = 0;
    if (!rows1) c1 = 0;
    BG_CHECK_ARG(N > 0 && D > 0 && H > 0 && W > 0 && C >= 0 && c0 >= 0 && c1 >= 0 && C + c0 + c1 > 0 && k >= 1 && stride >= 1 &&
                     pad >= 0 && (!mask || (C > 0 && ld % vec == 0 && ld >= (C + vec - 1) / vec * vec)) &&
                     (double)(C + c0 + c1) * k * k * k < 16777216.0, "bg_mask_window: bad sizes");
    const int kd = planar ? 1 : k, sd = planar ? 1 : stride, pd = planar ? 0 : pad;
    BG_CHECK_ARG(Do == (D + 2 * pd - kd) / sd + 1 && Ho == (H + 2 * pad - k) / stride + 1 && Wo == (W + 2 * pad - k) / stride + 1,
                 "bg_mask_window: output size does not match the conv arithmetic");
    MaskWinParams P{mask, ld, C, rows0, (float)c0, rows1, (float)c1, N, D, H, W, Do, Ho, Wo, k, stride, pad, kd, sd, pd,
                    (float)(C + c0 + c1) * kd * k * k, eps, update_mask, ratio};
    const long long total = (long long)N * Do * Ho * Wo;
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((mask_window_kernel<T>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, P));
    BG_CHECK_LAUNCH("mask_window_kernel");
    return BG_OK;
}

extern "C" int bg_mul_rows(int32_t dtype, const void* x, int32_t ldx, const void* m, int32_t ldm, void* y, int32_t ldy,
                           int64_t rows, int32_t C, void* stream) {
    BG_CHECK_ARG(dtype_ok(dtype) && x && m && y && aligned16(x) && aligned16(m) && aligned16(y) && rows > 0 && C > 0,
                 "bg_mul_rows: bad args");
    const int vec = dtype_vec(dtype);
    BG_CHECK_ARG(C % vec == 0 && ldx % vec == 0 && ldm % vec == 0 && ldy % vec == 0 && ldx >= C && ldm >= C && ldy >= C,
                 "bg_mul_rows: C/ld must be multiples of %d", vec);
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((mul_rows_kernel<T>), dim3(grid1d(rows * (C / vec))), dim3(256), 0,
                                                   (hipStream_t)stream, (const T*)x, ldx, (const T*)m, ldm, (T*)y, ldy,
                                                   (long long)rows, C));
    BG_CHECK_LAUNCH("mul_rows_kernel");
    return BG_OK;
}

extern "C" int bg_scale_rows(int32_t dtype, const void* x, int32_t ldx, const float* s, const float* bias, const float* t,
                             void* y, int32_t ldy, int64_t rows, int32_t C, void* stream) {
    BG_CHECK_ARG(dtype_ok(dtype) && s && y && aligned16(y) && rows > 0 && C > 0 && (!x || aligned16(x)) && (!bias || t),
                 "bg_scale_rows: bad args");
    const int vec = dtype_vec(dtype);
    BG_CHECK_ARG(C % vec == 0 && ldy % vec == 0 && ldy >= C && (!x || (ldx % vec == 0 && ldx >= C)),
                 "bg_scale_rows: C/ld must be multiples of %d", vec);
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((scale_rows_kernel<T>), dim3(grid1d(rows * (C / vec))), dim3(256), 0,
                                                   (hipStream_t)stream, (const T*)x, ldx, s, bias, t, (T*)y, ldy,
                                                   (long long)rows, C));
    BG_CHECK_LAUNCH("scale_rows_kernel");
    return BG_OK;
}

static int check_nearest(int32_t dtype, const void* a, const void* b, int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C,
                         int lda, int ldb, const char* who) {
    BG_CHECK_ARG(dtype_ok(dtype) && a && b && aligned16(a) && aligned16(b), "%s: bad dtype / pointer", who);
    const int vec = dtype_vec(dtype);
    BG_CHECK_ARG(N > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0 && C > 0 && C % vec == 0 && lda % vec == 0 &&
                     ldb % vec == 0 && lda >= C && ldb >= C, "%s: bad sizes", who);
    return BG_OK;
}

extern "C" int bg_resize_nearest3d_fwd(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int32_t N, int32_t Di,
                                       int32_t Hi, int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, int32_t C, void* stream) {
    int rc = check_nearest(dtype, x, y, N, Di, Hi, Wi, Do, Ho, Wo, C, ldx, ldy, "bg_resize_nearest3d_fwd");
    if (rc) return rc;
    NearestParams P{x, y, N, Di, Hi, Wi, Do, Ho, Wo, C, ldx, ldy};
    const long long total = (long long)N * Do * Ho * Wo * (C / dtype_vec(dtype));
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((nearest3d_fwd_kernel<T>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, P));
    BG_CHECK_LAUNCH("nearest3d_fwd_kernel");
    return BG_OK;
}

extern "C" int bg_resize_nearest3d_bwd(int32_t dtype, const void* dy, int32_t lddy, void* dx, int32_t lddx, int32_t N, int32_t Di,
                                       int32_t Hi, int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, int32_t C, void* stream) {
    int rc = check_nearest(dtype, dy, dx, N, Di, Hi, Wi, Do, Ho, Wo, C, lddx, lddy, "bg_resize_nearest3d_bwd");
    if (rc) return rc;
    NearestParams P{dx, const_cast<void*>(dy), N, Di, Hi, Wi, Do, Ho, Wo, C, lddx, lddy};
    const long long total = (long long)N * Di * Hi * Wi * (C / dtype_vec(dtype));
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((nearest3d_bwd_kernel<T>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, P));
    BG_CHECK_LAUNCH("nearest3d_bwd_kernel");
    return BG_OK;
}

extern "C" int bg_resize_nearest3d_rows(const float* x, float* y, int32_t N, int32_t Di, int32_t Hi, int32_t Wi, int32_t Do,
                                        int32_t Ho, int32_t Wo, void* stream) {
    BG_CHECK_ARG(x && y && N > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0, "bg_resize_nearest3d_rows: bad args");
    hipLaunchKernelGGL(nearest3d_rows_kernel, dim3(grid1d((long long)N * Do * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, x, y, N,
                       Di, Hi, Wi, Do, Ho, Wo);
    BG_CHECK_LAUNCH("nearest3d_rows_kernel");
    return BG_OK;
}

extern "C" int bg_tv_loss_fwd(const float* x, int64_t A, int32_t D, int32_t H, int32_t W, float* loss, void* stream) {
    BG_CHECK_ARG(x && loss && A > 0 && D > 1 && H > 1 && W > 0, "bg_tv_loss_fwd: bad args (needs D > 1 and H > 1)");
    const float inv_h = 1.f / ((float)A * D * (H - 1) * W), inv_d = 1.f / ((float)A * (D - 1) * H * W);
    long long g = (A * D * H * W + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(tv_fwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, (long long)A, D, H, W, inv_h, inv_d, loss);
    BG_CHECK_LAUNCH("tv_fwd_kernel");
    return BG_OK;
}

extern "C" int bg_tv_loss_bwd(const float* x, int64_t A, int32_t D, int32_t H, int32_t W, const float* coef, float* dx, void* stream) {
    BG_CHECK_ARG(x && coef && dx && A > 0 && D > 1 && H > 1 && W > 0, "bg_tv_loss_bwd: bad args");
    const float inv_h = 1.f / ((float)A * D * (H - 1) * W), inv_d = 1.f / ((float)A * (D - 1) * H * W);
    hipLaunchKernelGGL(tv_bwd_kernel, dim3(grid1d(A * D * H * W)), dim3(256), 0, (hipStream_t)stream, x, (long long)A, D, H, W, inv_h,
                       inv_d, coef, dx);
    BG_CHECK_LAUNCH("tv_bwd_kernel");
    return BG_OK;
}
