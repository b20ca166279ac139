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


// PCDropout3d (infill3d.py:115-135) in training mode.  nn.Dropout3d zeroes whole (sample, channel) maps of the MASK
// with probability p; keep[n][c] in {0, 1} is that draw (the 1/(1-p) rescaling cancels against `* self.scale` and the
// round()).  mask_d = mask * keep; drop_vals = mask - mask_d; input_d = input * (1 - drop_vals) / scale.
// The mask comes as one fp32 value per pixel (`rows`: an update_mask, whose channels are equal) or per channel (`m`).
template <typename T>
__global__ __launch_bounds__(256) void pc_dropout_kernel(const T* x, int ldx, const float* rows, const T* m, int ldm,
                                                         const float* keep, int ldk, T* y, int ldy, T* mo, int ldmo,
                                                         long long nrows, long long rows_per_sample, int C, float scale) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const long long total = nrows * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cv;
        const int c = (int)(i - r * cv) * VEC;
        const float* kp = keep + (r / rows_per_sample) * ldk + c;
        Chunk<T> xv, mv, yo, mk;
        xv.load(x + r * ldx + c);
        if (m) mv.load(m + r * ldm + c);
        const float mr = rows ? rows[r] : 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float mm = rows ? mr : mv.get(e);
            const float md = mm * kp[e];
            yo.set(e, xv.get(e) * (1.f - (mm - md)) / scale);
            mk.set(e, md);
        }
        yo.store(y + r * ldy + c);
        if (mo) mk.store(mo + r * ldmo + c);
    }
}

// F.interpolate(size=..., mode='trilinear') as infill3d.py:217-220 calls it (align_corners unset = False): source
// coordinate max(scale * (o + 0.5) - 0.5, 0) with scale = in / out in fp32, i0 = min((int)src, in - 1),
// lambda1 = clamp(src - i0, 0, 1), i1 = i0 + (i0 < in - 1), and the identity where in == out (ATen's
// compute_source_index_and_lambda).  The product and the difference are rounded separately (no fused multiply-add).
__device__ __forceinline__ void tri_src(int o, int in, int out, float scale, int& i0, int& i1, float& l1) {
    if (in == out) { i0 = i1 = o; l1 = 0.f; return; }
    const float r = fmaxf(__fsub_rn(__fmul_rn(scale, (float)o + 0.5f), 0.5f), 0.f);
    i0 = min((int)r, in - 1);
    l1 = fminf(fmaxf(r - (float)i0, 0.f), 1.f);
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
}

struct Tri3Params {
    const void* x; void* y;        // fwd: x -> y; bwd: x = dy (output-sized), y = dx (input-sized)
    int N, Di, Hi, Wi, Do, Ho, Wo, C, ldx, ldy;
    float sd, sh, sw;
};

template <typename T>
__global__ __launch_bounds__(256) void trilinear3d_fwd_kernel(Tri3Params P) {
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
        int d0, d1, h0, h1, w0, w1;
        float ld1, lh1, lw1;
        tri_src(od, P.Di, P.Do, P.sd, d0, d1, ld1);
        tri_src(oh, P.Hi, P.Ho, P.sh, h0, h1, lh1);
        tri_src(ow, P.Wi, P.Wo, P.sw, w0, w1, lw1);
        const T* xb = reinterpret_cast<const T*>(P.x) + (long long)n * P.Di * P.Hi * P.Wi * P.ldx + c;
        auto at = [&](int d, int h, int w) {
            Chunk<T> v;
            v.load(xb + (((long long)d * P.Hi + h) * P.Wi + w) * P.ldx);
            return v;
        };
        const Chunk<T> v000 = at(d0, h0, w0), v001 = at(d0, h0, w1), v010 = at(d0, h1, w0), v011 = at(d0, h1, w1);
        const Chunk<T> v100 = at(d1, h0, w0), v101 = at(d1, h0, w1), v110 = at(d1, h1, w0), v111 = at(d1, h1, w1);
        const float ld0 = 1.f - ld1, lh0 = 1.f - lh1, lw0 = 1.f - lw1;
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float a = ld0 * (lh0 * (lw0 * v000.get(e) + lw1 * v001.get(e)) + lh1 * (lw0 * v010.get(e) + lw1 * v011.get(e))) +
                            ld1 * (lh0 * (lw0 * v100.get(e) + lw1 * v101.get(e)) + lh1 * (lw0 * v110.get(e) + lw1 * v111.get(e)));
            o.set(e, a);
        }
        o.store(reinterpret_cast<T*>(P.y) + pix * P.ldy + c);
    }
}

// the outputs along one axis that read input index i: a window around the inverse of the source map, every candidate
// checked with the forward's own arithmetic (so this is the exact adjoint); weight = (i0 == i) * l0 + (i1 == i) * l1
__device__ __forceinline__ void tri_range(int i, int in, int out, float scale, int& lo, int& hi) {
    if (in == out) { lo = hi = i; return; }
    const float inv = (float)out / (float)in;
    lo = max(0, (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1);
    hi = min(out - 1, (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1);
}
__device__ __forceinline__ float tri_weight(int o, int i, int in, int out, float scale) {
    int i0, i1;
    float l1;
    tri_src(o, in, out, scale, i0, i1, l1);
    return (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void trilinear3d_bwd_kernel(Tri3Params P) {   // gather per INPUT voxel: no atomics, deterministic
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
        int dlo, dhi, hlo, hhi, wlo, whi;
        tri_range(id, P.Di, P.Do, P.sd, dlo, dhi);
        tri_range(ih, P.Hi, P.Ho, P.sh, hlo, hhi);
        tri_range(iw, P.Wi, P.Wo, P.sw, wlo, whi);
        const T* gb = reinterpret_cast<const T*>(P.x) + (long long)n * P.Do * P.Ho * P.Wo * P.ldx + c;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int od = dlo; od <= dhi; ++od) {
            const float wd = tri_weight(od, id, P.Di, P.Do, P.sd);
            if (wd == 0.f) continue;
            for (int oh = hlo; oh <= hhi; ++oh) {
                const float wh = wd * tri_weight(oh, ih, P.Hi, P.Ho, P.sh);
                if (wh == 0.f) continue;
                for (int ow = wlo; ow <= whi; ++ow) {
                    const float ww = wh * tri_weight(ow, iw, P.Wi, P.Wo, P.sw);
                    if (ww == 0.f) continue;
                    Chunk<T> g;
                    g.load(gb + (((long long)od * P.Ho + oh) * P.Wo + ow) * P.ldx);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] = fmaf(ww, g.get(e), acc[e]);
                }
            }
        }
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
        o.store(reinterpret_cast<T*>(P.y) + pix * P.ldy + c);
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

extern "C" int bg_pc_dropout(int32_t dtype, const void* x, int32_t ldx, const float* rows, const void* mask, int32_t ldm,
                             const float* keep, int32_t ldk, void* y, int32_t ldy, void* mask_out, int32_t ldmo, int64_t nrows,
                             int64_t rows_per_sample, int32_t C, float scale, void* stream) {
    BG_CHECK_ARG(dtype_ok(dtype) && x && keep && y && aligned16(x) && aligned16(y) && ((rows != nullptr) != (mask != nullptr)) &&
                     (!mask || aligned16(mask)) && (!mask_out || aligned16(mask_out)),
                 "bg_pc_dropout: bad dtype / pointer (exactly one of rows / mask)");
    const int vec = dtype_vec(dtype);
    BG_CHECK_ARG(nrows > 0 && rows_per_sample > 0 && nrows % rows_per_sample == 0 && C > 0 && C % vec == 0 && ldx % vec == 0 &&
                     ldy % vec == 0 && ldx >= C && ldy >= C && ldk >= C && (!mask || (ldm % vec == 0 && ldm >= C)) &&
                     (!mask_out || (ldmo % vec == 0 && ldmo >= C)) && scale > 0.f && scale <= 1.f,
                 "bg_pc_dropout: bad sizes (C/ld multiples of %d, 0 < scale <= 1)", vec);
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((pc_dropout_kernel<T>), dim3(grid1d(nrows * (C / vec))), dim3(256), 0,
                                                   (hipStream_t)stream, (const T*)x, ldx, rows, (const T*)mask, ldm, keep, ldk, (T*)y,
                                                   ldy, (T*)mask_out, ldmo, (long long)nrows, (long long)rows_per_sample, C, scale));
    BG_CHECK_LAUNCH("pc_dropout_kernel");
    return BG_OK;
}

static int tri3_params(Tri3Params& P, int32_t dtype, const void* a, int32_t lda, void* b, int32_t ldb, int N, int Di, int Hi, int Wi,
                       int Do, int Ho, int Wo, int C, const char* who) {
    // a / lda: the tensor read (x forward, dy backward); b / ldb: the tensor written
    int rc = check_nearest(dtype, a, b, N, Di, Hi, Wi, Do, Ho, Wo, C, lda, ldb, who);
    if (rc) return rc;
    P = Tri3Params{a, b, N, Di, Hi, Wi, Do, Ho, Wo, C, lda, ldb, (float)Di / (float)Do, (float)Hi / (float)Ho, (float)Wi / (float)Wo};
    return BG_OK;
}

extern "C" int bg_resize_trilinear3d_fwd(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int32_t N, int32_t Di,
                                         int32_t Hi, int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, int32_t C, void* stream) {
    Tri3Params P;
    int rc = tri3_params(P, dtype, x, ldx, y, ldy, N, Di, Hi, Wi, Do, Ho, Wo, C, "bg_resize_trilinear3d_fwd");
    if (rc) return rc;
    const long long total = (long long)N * Do * Ho * Wo * (C / dtype_vec(dtype));
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((trilinear3d_fwd_kernel<T>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, P));
    BG_CHECK_LAUNCH("trilinear3d_fwd_kernel");
    return BG_OK;
}

extern "C" int bg_resize_trilinear3d_bwd(int32_t dtype, const void* dy, int32_t lddy, void* dx, int32_t lddx, int32_t N, int32_t Di,
                                         int32_t Hi, int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, int32_t C, void* stream) {
    Tri3Params P;
    int rc = tri3_params(P, dtype, dy, lddy, dx, lddx, N, Di, Hi, Wi, Do, Ho, Wo, C, "bg_resize_trilinear3d_bwd");
    if (rc) return rc;
    const long long total = (long long)N * Di * Hi * Wi * (C / dtype_vec(dtype));
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((trilinear3d_bwd_kernel<T>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, P));
    BG_CHECK_LAUNCH("trilinear3d_bwd_kernel");
    return BG_OK;
}
