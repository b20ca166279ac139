// Bilinear resize (align_corners=True), row broadcast, dtype/layout conversion.
// HBM-bound element-wise kernels: lanes along channels, 16 bytes per lane.
#include "common.h"

namespace {

// PyTorch's align_corners=True source index, evaluated in fp32 exactly as
// aten's area_pixel_compute_source_index does for float inputs.
__device__ __forceinline__ void src_index(int o, float ratio, int in_size, int& i0, int& i1, float& l0, float& l1) {
    const float r = ratio * (float)o;
    i0 = (int)r;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    l1 = r - (float)i0;
    l0 = 1.f - l1;
}
__host__ __device__ inline float ac_ratio(int in_size, int out_size) {
    return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
}

struct RsParams {
    const void* x; int ldx;
    void* y; int ldy;
    int N, Hi, Wi, Ho, Wo, C;
    float rh, rw;
};

// 4 channels of T as one 16-byte (f32) / 8-byte (bf16) access
template <typename T> struct Quad;
template <> struct Quad<float> {
    typedef f32x4 vec;
    __device__ static inline void load(const float* p, float (&v)[4]) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(p);
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
    }
    __device__ static inline void store(float* p, const float (&v)[4]) { *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]}; }
};
template <> struct Quad<bf16_t> {
    __device__ static inline void load(const bf16_t* p, float (&v)[4]) {
        const bf16x4 q = *reinterpret_cast<const bf16x4*>(p);
        v[0] = (float)q[0]; v[1] = (float)q[1]; v[2] = (float)q[2]; v[3] = (float)q[3];
    }
    __device__ static inline void store(bf16_t* p, const float (&v)[4]) {
        *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    }
};

// grid: one 1-D range of (output row, 256-item segment of the row's Wo * C/4 items): the row decomposition is one
// scalar 32-bit division per block, the per-thread one a 32-bit division by C/4 -- the flat 64-bit index of the first
// version cost three 64-bit divisions per item and held the kernel at 1.7 TB/s.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void resize_fwd_kernel(RsParams P) {
    constexpr int V = 4;
    const unsigned cv = P.C / V;
    const unsigned items = (unsigned)P.Wo * cv, segs = (items + 255u) / 256u;
    const unsigned row = blockIdx.x / segs, seg = blockIdx.x - row * segs;   // row = n * Ho + ho
    const unsigned it = seg * 256u + threadIdx.x;
    if (it >= items) return;
    const unsigned wo = it / cv;
    const int c = (int)(it - wo * cv) * V;
    const unsigned n = row / (unsigned)P.Ho, ho = row - n * (unsigned)P.Ho;
    int h0, h1, w0, w1;
    float lh0, lh1, lw0, lw1;
    src_index((int)ho, P.rh, P.Hi, h0, h1, lh0, lh1);
    src_index((int)wo, P.rw, P.Wi, w0, w1, lw0, lw1);
    const TI* b = reinterpret_cast<const TI*>(P.x) + (long long)n * P.Hi * P.Wi * P.ldx + c;
    float v00[4], v01[4], v10[4], v11[4], o[4];
    Quad<TI>::load(b + ((long long)h0 * P.Wi + w0) * P.ldx, v00);
    Quad<TI>::load(b + ((long long)h0 * P.Wi + w1) * P.ldx, v01);
    Quad<TI>::load(b + ((long long)h1 * P.Wi + w0) * P.ldx, v10);
    Quad<TI>::load(b + ((long long)h1 * P.Wi + w1) * P.ldx, v11);
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = lh0 * (lw0 * v00[e] + lw1 * v01[e]) + lh1 * (lw0 * v10[e] + lw1 * v11[e]);
    Quad<TO>::store(reinterpret_cast<TO*>(P.y) + ((long long)row * P.Wo + wo) * P.ldy + c, o);
}

// Adjoint as a gather: every source pixel collects from the output pixels whose
// forward stencil touches it, using the SAME fp32 index arithmetic as forward.
template <typename TG, typename TX>
__global__ __launch_bounds__(256) void resize_bwd_kernel(RsParams P /* x = dy [Ho,Wo], y = dx [Hi,Wi] */) {
    constexpr int V = 4;
    const unsigned cv = P.C / V;
    const TG* dy = reinterpret_cast<const TG*>(P.x);
    TX* dx = reinterpret_cast<TX*>(P.y);
    const unsigned items = (unsigned)P.Wi * cv, segs = (items + 255u) / 256u;   // grid as in resize_fwd_kernel, over dx rows
    const unsigned row = blockIdx.x / segs, seg = blockIdx.x - row * segs;
    const unsigned it = seg * 256u + threadIdx.x;
    if (it >= items) return;
    {
        const int wi = (int)(it / cv);
        const int c = (int)(it - (unsigned)wi * cv) * V;
        const int n = (int)(row / (unsigned)P.Hi), hi = (int)(row - (unsigned)n * (unsigned)P.Hi);
        // candidate output rows: those whose source coordinate lies in (hi-1, hi+1)
        int ho_lo = 0, ho_hi = P.Ho - 1, wo_lo = 0, wo_hi = P.Wo - 1;
        if (P.rh > 0.f) {
            ho_lo = (int)floorf((float)(hi - 1) / P.rh) - 1;
            ho_hi = (int)ceilf((float)(hi + 1) / P.rh) + 1;
            if (ho_lo < 0) ho_lo = 0;
            if (ho_hi > P.Ho - 1) ho_hi = P.Ho - 1;
        }
        if (P.rw > 0.f) {
            wo_lo = (int)floorf((float)(wi - 1) / P.rw) - 1;
            wo_hi = (int)ceilf((float)(wi + 1) / P.rw) + 1;
            if (wo_lo < 0) wo_lo = 0;
            if (wo_hi > P.Wo - 1) wo_hi = P.Wo - 1;
        }
        float acc[V] = {0.f, 0.f, 0.f, 0.f};
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            int h0, h1;
            float lh0, lh1;
            src_index(ho, P.rh, P.Hi, h0, h1, lh0, lh1);
            float wh = 0.f;
            if (h0 == hi) wh += lh0;
            if (h1 == hi) wh += lh1;
            if (wh == 0.f) continue;
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                int w0, w1;
                float lw0, lw1;
                src_index(wo, P.rw, P.Wi, w0, w1, lw0, lw1);
                float ww = 0.f;
                if (w0 == wi) ww += lw0;
                if (w1 == wi) ww += lw1;
                if (ww == 0.f) continue;
                float g[4];
                Quad<TG>::load(dy + (((long long)n * P.Ho + ho) * P.Wo + wo) * P.ldx + c, g);
                const float f = wh * ww;
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] = fmaf(f, g[e], acc[e]);
            }
        }
        Quad<TX>::store(dx + (((long long)n * P.Hi + hi) * P.Wi + wi) * P.ldy + c, acc);
    }
}

template <typename T>
__global__ void broadcast_rows_kernel(const float* v, float scale, T* y, int ldy, long long rows, int C,
                                      long long rpg) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const long long total = rows * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VEC;
        const long long row = i / cv;
        const float* s = v + (row / rpg) * C + c;
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, s[e] * scale);
        o.store(y + row * ldy + c);
    }
}

template <typename TS, typename TD>
__global__ void cast_rows_kernel(const TS* src, int lds, TD* dst, int ldd, long long rows, int C) {
    constexpr int V = 4;
    const unsigned cv = C / V;
    const long long total = rows * cv;
    // (32-bit row / channel split whenever the item index fits: the 64-bit pair cost more than the copy)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long row;
        int c;
        if (total <= 0x7fffffffLL) {
            const unsigned r = (unsigned)i / cv;
            row = r;
            c = (int)((unsigned)i - r * cv) * V;
        } else {
            row = i / cv;
            c = (int)(i - row * cv) * V;
        }
        float v[4];
        Quad<TS>::load(src + row * lds + c, v);
        Quad<TD>::store(dst + row * ldd + c, v);
    }
}

template <typename T>
__global__ void axpy_rows_kernel(const T* x, int ldx, T* y, int ldy, long long rows, int C) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const long long total = rows * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VEC;
        const long long row = i / cv;
        Chunk<T> a, b;
        a.load(x + row * ldx + c);
        b.load(y + row * ldy + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) b.set(e, a.get(e) + b.get(e));
        b.store(y + row * ldy + c);
    }
}

// NCHW fp32 -> NHWC T (zero-filled channel padding).  One thread per pixel:
// reads are coalesced along pixels for every channel, writes are Cp*sizeof(T)
// contiguous bytes per lane.  C is small on this path (the field channels).
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* src, T* dst, int N, int C, int HW, int Cp, int ldd) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / HW);
        const int p = (int)(i - (long long)n * HW);
        const float* s = src + (long long)n * C * HW + p;
        T* d = dst + i * ldd;
        for (int c = 0; c < C; ++c) d[c] = Elem<T>::from_f(s[(long long)c * HW]);
        for (int c = C; c < Cp; ++c) d[c] = Elem<T>::from_f(0.f);
    }
}

// Fast forms (16-byte aligned rows, Cp and ld multiples of the 16-byte vector): one pixel per
// thread, channel loads coalesced across the wave, 16-byte packed stores / loads on the NHWC side.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_vec_kernel(const float* src, T* dst, int C, int HW, int Cp, int ldd) {
    constexpr int VEC = Elem<T>::VEC;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int n = blockIdx.y;
    const float* s = src + (long long)n * C * HW + p;
    T* d = dst + ((long long)n * HW + p) * ldd;
    for (int c0 = 0; c0 < Cp; c0 += VEC) {
        float v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = (c0 + e < C) ? __builtin_nontemporal_load(s + (long long)(c0 + e) * HW) : 0.f;
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, v[e]);
        o.store(d + c0);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_vec_kernel(const T* src, int lds, float* dst, int C, int HW) {
    constexpr int VEC = Elem<T>::VEC;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int n = blockIdx.y;
    const T* s = src + ((long long)n * HW + p) * lds;
    float* d = dst + (long long)n * C * HW + p;
    for (int c0 = 0; c0 < C; c0 += VEC) {
        Chunk<T> v;
        v.load(s + c0);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            if (c0 + e < C) __builtin_nontemporal_store(v.get(e), d + (long long)(c0 + e) * HW);
    }
}

// Four pixels per thread (HW % 4 == 0, 16-byte aligned planes): float4 loads / stores on the NCHW side -- a
// quarter of the instructions of the one-pixel form for the same bytes.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_vec4_kernel(const float* src, T* dst, int C, int HW, int Cp, int ldd) {
    constexpr int VEC = Elem<T>::VEC;
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= HW) return;
    const int n = blockIdx.y;
    const float* s = src + (long long)n * C * HW + p;
    T* d = dst + ((long long)n * HW + p) * ldd;
    for (int c0 = 0; c0 < Cp; c0 += VEC) {
        f32x4 v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            v[e] = (c0 + e < C) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(s + (long long)(c0 + e) * HW))
                                : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Chunk<T> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o.set(e, v[e][q]);
            o.store(d + (long long)q * ldd + c0);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_vec4_kernel(const T* src, int lds, float* dst, int C, int HW) {
    constexpr int VEC = Elem<T>::VEC;
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= HW) return;
    const int n = blockIdx.y;
    const T* s = src + ((long long)n * HW + p) * lds;
    float* d = dst + (long long)n * C * HW + p;
    for (int c0 = 0; c0 < C; c0 += VEC) {
        Chunk<T> v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q].load(s + (long long)q * lds + c0);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            if (c0 + e < C) {
                f32x4 o = {v[0].get(e), v[1].get(e), v[2].get(e), v[3].get(e)};
                __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(d + (long long)(c0 + e) * HW));
            }
    }
}

// bf16 NHWC rows of 16 / 32 / 64 bytes (8 * CPP channels, dense) <-> fp32 NCHW through LDS: a block moves 1 024 pixels; the
// NHWC side is touched in whole consecutive kilobytes per wave instruction (the four-pixel form above reads / writes 16 bytes
// of every 128-byte line per instruction: 2.2-3.3 TB/s on the 1152 x 768 x 16 fields), the NCHW side in float4 per plane.
// Chunk g of the tile (16 bytes; g = pixel * CPP + chunk) sits at 16 * (8 * (g / 8) + ((g % 8) ^ ((g / 8) & 7))): linear
// writes and the transposed reads (lane stride 4 * CPP chunks) are both conflict-free for CPP <= 2, 2-way for CPP = 4.
constexpr int LT_PIX = 1024;
__device__ __forceinline__ int lt_swz(int g) { return ((g >> 3) << 3) + ((g & 7) ^ ((g >> 3) & 7)); }

template <int CPP>
__global__ __launch_bounds__(256) void nhwc_to_nchw_lds_kernel(const bf16_t* src, float* dst, int C, int HW) {
    __shared__ u32x4 tile[LT_PIX * CPP];
    const int n = blockIdx.y, tid = threadIdx.x;
    const long long p0 = (long long)blockIdx.x * LT_PIX;
    const int npix = (int)min((long long)LT_PIX, (long long)HW - p0);
    const u32x4* s = reinterpret_cast<const u32x4*>(src + ((long long)n * HW + p0) * (8 * CPP));
#pragma unroll
    for (int k = 0; k < 4 * CPP; ++k) {
        const int g = k * 256 + tid;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (g < npix * CPP) v = __builtin_nontemporal_load(s + g);
        tile[lt_swz(g)] = v;
    }
    __syncthreads();
    const int p = tid * 4;
    if (p >= npix) return;                      // HW % 4 == 0: a thread's four pixels are all inside or all outside
    float* d = dst + (long long)n * C * HW + p0 + p;
#pragma unroll
    for (int cc = 0; cc < CPP; ++cc) {
        u32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = tile[lt_swz((p + q) * CPP + cc)];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (cc * 8 + e >= C) break;
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned w = v[q][e >> 1];
                o[q] = __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
            }
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(d + (long long)(cc * 8 + e) * HW));
        }
    }
}

template <int CPP>
__global__ __launch_bounds__(256) void nchw_to_nhwc_lds_kernel(const float* src, bf16_t* dst, int C, int HW) {
    __shared__ u32x4 tile[LT_PIX * CPP];
    const int n = blockIdx.y, tid = threadIdx.x;
    const long long p0 = (long long)blockIdx.x * LT_PIX;
    const int npix = (int)min((long long)LT_PIX, (long long)HW - p0);
    const int p = tid * 4;
    if (p < npix) {
        const float* sp = src + (long long)n * C * HW + p0 + p;
#pragma unroll
        for (int cc = 0; cc < CPP; ++cc) {
            f32x4 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] = (cc * 8 + e < C) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + (long long)(cc * 8 + e) * HW))
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                Chunk<bf16_t> o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o.set(e, v[e][q]);
                tile[lt_swz((p + q) * CPP + cc)] = __builtin_bit_cast(u32x4, o.v);
            }
        }
    }
    __syncthreads();
    u32x4* d = reinterpret_cast<u32x4*>(dst + ((long long)n * HW + p0) * (8 * CPP));
#pragma unroll
    for (int k = 0; k < 4 * CPP; ++k) {
        const int g = k * 256 + tid;
        if (g < npix * CPP) d[g] = tile[lt_swz(g)];
    }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* src, int lds, float* dst, int N, int C, int HW) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / HW);
        const int p = (int)(i - (long long)n * HW);
        const T* s = src + i * lds;
        float* d = dst + (long long)n * C * HW + p;
        for (int c = 0; c < C; ++c) d[(long long)c * HW] = Elem<T>::to_f(s[c]);
    }
}

// AvgPool2d(kernel 2, stride 1, count_include_pad): y[n,h,w,:] = 0.25 * sum_{i,j in {0,1}}
// x[n, h+off+i, w+off+j, :], zero outside the input.  off = -padding is the forward pass,
// off = padding-1 with the roles of x and y exchanged is its adjoint.
template <typename T>
__global__ __launch_bounds__(256) void avgpool2x2_kernel(const T* x, int ldx, T* y, int ldy, int Hi, int Wi, int Ho, int Wo, int C,
                                                         int off) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const int idx = blockIdx.x * 256 + threadIdx.x;  // (wo, channel vector) of one output row
    if (idx >= Wo * cv) return;
    const int wo = idx / cv, c = (idx - wo * cv) * VEC;
    const int n = blockIdx.y / Ho, ho = blockIdx.y - n * Ho;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ih = ho + off + i;
        if ((unsigned)ih >= (unsigned)Hi) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int iw = wo + off + j;
            if ((unsigned)iw >= (unsigned)Wi) continue;
            Chunk<T> v;
            v.load(x + (((long long)n * Hi + ih) * Wi + iw) * ldx + c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += v.get(e);
        }
    }
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, 0.25f * acc[e]);
    o.store(y + (((long long)n * Ho + ho) * Wo + wo) * ldy + c);
}

__global__ void fill_kernel(float* p, float v, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        p[i] = v;
}

inline unsigned ew_grid(long long total) {
    long long g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}

int check_rs(int dt_a, int dt_b, const void* a, int lda, void* b, int ldb, int N, int Hi, int Wi, int Ho, int Wo,
             int C, const char* who) {
    BG_CHECK_ARG(dtype_ok(dt_a) && dtype_ok(dt_b), "%s: bad dtype", who);
    BG_CHECK_ARG(a && b && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "%s: bad args", who);
    BG_CHECK_ARG(C % 4 == 0 && lda >= C && ldb >= C && lda % 4 == 0 && ldb % 4 == 0, "%s: C/ld must be multiples of 4", who);
    BG_CHECK_ARG(((uintptr_t)a % (dt_a == BG_BF16 ? 8 : 16)) == 0 && ((uintptr_t)b % (dt_b == BG_BF16 ? 8 : 16)) == 0,
                 "%s: pointers must be aligned to four channels", who);
    return BG_OK;
}

}  // namespace

#define RS_DISPATCH(dta, dtb, KERNEL, ...)                                                            \
    do {                                                                                              \
        if ((dta) == BG_BF16 && (dtb) == BG_BF16) hipLaunchKernelGGL((KERNEL<bf16_t, bf16_t>), __VA_ARGS__); \
        else if ((dta) == BG_BF16 && (dtb) == BG_F32) hipLaunchKernelGGL((KERNEL<bf16_t, float>), __VA_ARGS__); \
        else if ((dta) == BG_F32 && (dtb) == BG_BF16) hipLaunchKernelGGL((KERNEL<float, bf16_t>), __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<float, float>), __VA_ARGS__);                                  \
    } while (0)

extern "C" int bg_resize_bilinear_fwd(int32_t in_dtype, int32_t out_dtype, const void* x, int32_t ldx, void* y,
                                      int32_t ldy, int32_t N, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                      int32_t C, void* stream) {
    int rc = check_rs(in_dtype, out_dtype, x, ldx, y, ldy, N, Hi, Wi, Ho, Wo, C, "bg_resize_bilinear_fwd");
    if (rc) return rc;
    RsParams P{x, ldx, y, ldy, N, Hi, Wi, Ho, Wo, C, ac_ratio(Hi, Ho), ac_ratio(Wi, Wo)};
    const long long blocks = (long long)N * Ho * (((long long)Wo * (C / 4) + 255) / 256);
    BG_CHECK_ARG(blocks <= 0x7fffffffLL && (long long)Wo * (C / 4) < 0x7fffffffLL, "bg_resize_bilinear_fwd: grid too large");
    RS_DISPATCH(in_dtype, out_dtype, resize_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, P);
    BG_CHECK_LAUNCH("resize_fwd_kernel");
    return BG_OK;
}

extern "C" int bg_resize_bilinear_bwd(int32_t dy_dtype, int32_t dx_dtype, const void* dy, int32_t lddy, void* dx,
                                      int32_t lddx, int32_t N, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                      int32_t C, void* stream) {
    int rc = check_rs(dy_dtype, dx_dtype, dy, lddy, dx, lddx, N, Hi, Wi, Ho, Wo, C, "bg_resize_bilinear_bwd");
    if (rc) return rc;
    RsParams P{dy, lddy, dx, lddx, N, Hi, Wi, Ho, Wo, C, ac_ratio(Hi, Ho), ac_ratio(Wi, Wo)};
    const long long blocks = (long long)N * Hi * (((long long)Wi * (C / 4) + 255) / 256);
    BG_CHECK_ARG(blocks <= 0x7fffffffLL && (long long)Wi * (C / 4) < 0x7fffffffLL, "bg_resize_bilinear_bwd: grid too large");
    RS_DISPATCH(dy_dtype, dx_dtype, resize_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, P);
    BG_CHECK_LAUNCH("resize_bwd_kernel");
    return BG_OK;
}

extern "C" int bg_broadcast_rows(int32_t dtype, const float* v, float scale, void* y, int32_t ldy, int64_t rows,
                                 int32_t C, int32_t groups, void* stream) {
    BG_CHECK_ARG(dtype_ok(dtype) && v && y && aligned16(y) && rows > 0 && groups > 0 && rows % groups == 0,
                 "bg_broadcast_rows: bad args");
    BG_CHECK_ARG(C % dtype_vec(dtype) == 0 && ldy >= C && ldy % dtype_vec(dtype) == 0, "bg_broadcast_rows: bad C/ld");
    const long long total = rows * (C / dtype_vec(dtype));
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((broadcast_rows_kernel<T>), dim3(ew_grid(total)), dim3(256), 0,
                                                   (hipStream_t)stream, v, scale, (T*)y, ldy, (long long)rows, C,
                                                   (long long)(rows / groups)));
    BG_CHECK_LAUNCH("broadcast_rows_kernel");
    return BG_OK;
}

extern "C" int bg_cast_rows(int32_t src_dtype, int32_t dst_dtype, const void* src, int32_t lds, void* dst, int32_t ldd,
                            int64_t rows, int32_t C, void* stream) {
    BG_CHECK_ARG(dtype_ok(src_dtype) && dtype_ok(dst_dtype) && src && dst && rows > 0 && C > 0, "bg_cast_rows: bad args");
    BG_CHECK_ARG(C % 4 == 0 && lds >= C && ldd >= C && lds % 4 == 0 && ldd % 4 == 0, "bg_cast_rows: C and ld must be multiples of 4, ld >= C");
    BG_CHECK_ARG(((uintptr_t)src % (src_dtype == BG_BF16 ? 8 : 16)) == 0 && ((uintptr_t)dst % (dst_dtype == BG_BF16 ? 8 : 16)) == 0,
                 "bg_cast_rows: pointers must be aligned to four channels");
    const long long total = rows * (C / 4);
#define CAST_ARGS(TS, TD) dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const TS*)src, lds, (TD*)dst, ldd, (long long)rows, C
    if (src_dtype == BG_BF16 && dst_dtype == BG_BF16) hipLaunchKernelGGL((cast_rows_kernel<bf16_t, bf16_t>), CAST_ARGS(bf16_t, bf16_t));
    else if (src_dtype == BG_BF16) hipLaunchKernelGGL((cast_rows_kernel<bf16_t, float>), CAST_ARGS(bf16_t, float));
    else if (dst_dtype == BG_BF16) hipLaunchKernelGGL((cast_rows_kernel<float, bf16_t>), CAST_ARGS(float, bf16_t));
    else hipLaunchKernelGGL((cast_rows_kernel<float, float>), CAST_ARGS(float, float));
#undef CAST_ARGS
    BG_CHECK_LAUNCH("cast_rows_kernel");
    return BG_OK;
}

extern "C" int bg_axpy_rows(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int64_t rows, int32_t C,
                            void* stream) {
    BG_CHECK_ARG(dtype_ok(dtype) && x && y && aligned16(x) && aligned16(y) && rows > 0, "bg_axpy_rows: bad args");
    BG_CHECK_ARG(C % dtype_vec(dtype) == 0 && ldx >= C && ldy >= C && ldx % dtype_vec(dtype) == 0 &&
                     ldy % dtype_vec(dtype) == 0, "bg_axpy_rows: bad C/ld");
    const long long total = rows * (C / dtype_vec(dtype));
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((axpy_rows_kernel<T>), dim3(ew_grid(total)), dim3(256), 0,
                                                   (hipStream_t)stream, (const T*)x, ldx, (T*)y, ldy, (long long)rows, C));
    BG_CHECK_LAUNCH("axpy_rows_kernel");
    return BG_OK;
}

static bool lt_on() {   // BGAMD_LAYOUT_LDS=0: the four-pixel register forms (A/B)
    static const bool on = !(getenv("BGAMD_LAYOUT_LDS") && atoi(getenv("BGAMD_LAYOUT_LDS")) == 0);
    return on;
}

extern "C" int bg_nchw_to_nhwc(int32_t dst_dtype, const float* src, void* dst, int32_t N, int32_t C, int32_t HW,
                               int32_t Cp, int32_t ldd, void* stream) {
    BG_CHECK_ARG(dtype_ok(dst_dtype) && src && dst && N > 0 && C > 0 && HW > 0 && Cp >= C && ldd >= Cp,
                 "bg_nchw_to_nhwc: bad args");
    const long long total = (long long)N * HW;
    const int vec = dtype_vec(dst_dtype);
    if (lt_on() && dst_dtype == BG_BF16 && aligned16(dst) && aligned16(src) && ldd == Cp && (Cp == 8 || Cp == 16 || Cp == 32) &&
        N <= 65535 && HW % 4 == 0) {
        const dim3 grid((HW + LT_PIX - 1) / LT_PIX, N);
        if (Cp == 8) hipLaunchKernelGGL((nchw_to_nhwc_lds_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, C, HW);
        else if (Cp == 16) hipLaunchKernelGGL((nchw_to_nhwc_lds_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, C, HW);
        else hipLaunchKernelGGL((nchw_to_nhwc_lds_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, C, HW);
        BG_CHECK_LAUNCH("nchw_to_nhwc_lds_kernel");
        return BG_OK;
    }
    if (aligned16(dst) && aligned16(src) && Cp % vec == 0 && ldd % vec == 0 && N <= 65535 && HW % 4 == 0) {
        BG_DISPATCH_DTYPE(dst_dtype, T, hipLaunchKernelGGL((nchw_to_nhwc_vec4_kernel<T>), dim3((HW / 4 + 255) / 256, N), dim3(256),
                                                           0, (hipStream_t)stream, src, (T*)dst, C, HW, Cp, ldd));
        BG_CHECK_LAUNCH("nchw_to_nhwc_vec4_kernel");
        return BG_OK;
    }
    if (aligned16(dst) && Cp % vec == 0 && ldd % vec == 0 && N <= 65535) {
        BG_DISPATCH_DTYPE(dst_dtype, T, hipLaunchKernelGGL((nchw_to_nhwc_vec_kernel<T>), dim3((HW + 255) / 256, N), dim3(256),
                                                           0, (hipStream_t)stream, src, (T*)dst, C, HW, Cp, ldd));
        BG_CHECK_LAUNCH("nchw_to_nhwc_vec_kernel");
        return BG_OK;
    }
    BG_DISPATCH_DTYPE(dst_dtype, T, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(ew_grid(total)), dim3(256), 0,
                                                       (hipStream_t)stream, src, (T*)dst, N, C, HW, Cp, ldd));
    BG_CHECK_LAUNCH("nchw_to_nhwc_kernel");
    return BG_OK;
}

extern "C" int bg_nhwc_to_nchw(int32_t src_dtype, const void* src, int32_t lds, float* dst, int32_t N, int32_t C,
                               int32_t HW, void* stream) {
    BG_CHECK_ARG(dtype_ok(src_dtype) && src && dst && N > 0 && C > 0 && HW > 0 && lds >= C, "bg_nhwc_to_nchw: bad args");
    const long long total = (long long)N * HW;
    const int vec = dtype_vec(src_dtype);
    // the vector form reads whole 16-byte chunks: the row must hold them (lds >= C rounded up)
    // (towards NCHW the staged form only pays on 64-byte rows: 16 channels 119 against 108 us at 8 x 1152 x 768, 32 channels 521
    // against 686 us at 4 x 2304 x 1536; towards NHWC it pays everywhere: 103 against 144, 498 against 831; scripts/bench_layout.py)
    static const bool lt_all = getenv("BGAMD_LAYOUT_LDS") && atoi(getenv("BGAMD_LAYOUT_LDS")) == 2;
    if (lt_on() && src_dtype == BG_BF16 && aligned16(src) && aligned16(dst) && (lds == 32 || (lt_all && (lds == 8 || lds == 16))) && C <= lds &&
        N <= 65535 && HW % 4 == 0) {
        const dim3 grid((HW + LT_PIX - 1) / LT_PIX, N);
        if (lds == 8) hipLaunchKernelGGL((nhwc_to_nchw_lds_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, C, HW);
        else if (lds == 16) hipLaunchKernelGGL((nhwc_to_nchw_lds_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, C, HW);
        else hipLaunchKernelGGL((nhwc_to_nchw_lds_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, C, HW);
        BG_CHECK_LAUNCH("nhwc_to_nchw_lds_kernel");
        return BG_OK;
    }
    if (aligned16(src) && aligned16(dst) && lds % vec == 0 && lds >= (C + vec - 1) / vec * vec && N <= 65535 && HW % 4 == 0) {
        BG_DISPATCH_DTYPE(src_dtype, T, hipLaunchKernelGGL((nhwc_to_nchw_vec4_kernel<T>), dim3((HW / 4 + 255) / 256, N), dim3(256),
                                                           0, (hipStream_t)stream, (const T*)src, lds, dst, C, HW));
        BG_CHECK_LAUNCH("nhwc_to_nchw_vec4_kernel");
        return BG_OK;
    }
    if (aligned16(src) && lds % vec == 0 && lds >= (C + vec - 1) / vec * vec && N <= 65535) {
        BG_DISPATCH_DTYPE(src_dtype, T, hipLaunchKernelGGL((nhwc_to_nchw_vec_kernel<T>), dim3((HW + 255) / 256, N), dim3(256),
                                                           0, (hipStream_t)stream, (const T*)src, lds, dst, C, HW));
        BG_CHECK_LAUNCH("nhwc_to_nchw_vec_kernel");
        return BG_OK;
    }
    BG_DISPATCH_DTYPE(src_dtype, T, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(ew_grid(total)), dim3(256), 0,
                                                       (hipStream_t)stream, (const T*)src, lds, dst, N, C, HW));
    BG_CHECK_LAUNCH("nhwc_to_nchw_kernel");
    return BG_OK;
}

extern "C" int bg_avgpool2x2(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int32_t N, int32_t Hi,
                             int32_t Wi, int32_t Ho, int32_t Wo, int32_t C, int32_t off, void* stream) {
    BG_CHECK_ARG(dtype_ok(dtype) && x && y && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "bg_avgpool2x2: bad args");
    const int vec = dtype_vec(dtype);
    BG_CHECK_ARG(C % vec == 0 && ldx % vec == 0 && ldy % vec == 0 && ldx >= C && ldy >= C && aligned16(x) && aligned16(y),
                 "bg_avgpool2x2: C/ld must be multiples of %d and pointers 16-byte aligned", vec);
    BG_CHECK_ARG((long long)N * Ho <= 65535, "bg_avgpool2x2: N*Ho too large for this launch shape");
    const int items = Wo * (C / vec);
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((avgpool2x2_kernel<T>), dim3((items + 255) / 256, N * Ho), dim3(256), 0,
                                                   (hipStream_t)stream, (const T*)x, ldx, (T*)y, ldy, Hi, Wi, Ho, Wo, C, off));
    BG_CHECK_LAUNCH("avgpool2x2_kernel");
    return BG_OK;
}

extern "C" int bg_fill_f32(float* p, float v, int64_t n, void* stream) {
    BG_CHECK_ARG(p && n > 0, "bg_fill_f32: bad args");
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, v, (long long)n);
    BG_CHECK_LAUNCH("fill_kernel");
    return BG_OK;
}
