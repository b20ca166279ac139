// Kernels of the 3-D DeepLab GAN path (reference: architecture/gpsro/deeplab3d.py).
//
// A volume [N, D, H, W, C] is held as the NHWC tensor [N*D, H, W, C] (depth folded into the batch), so every
// 2-D element-wise, normalisation and 1x1(x1) convolution kernel applies unchanged.  What the third dimension
// adds lives here:
//   * depth unfold / fold: a dense k x k x k convolution becomes a 2-D k x k convolution over KD*C channels
//     (channel block kd of output slice od = input slice od*stride - pad + kd*dil, zeros outside) on the MFMA
//     GEMM kernels; the fold is its adjoint.  KD = 1 with stride 2 is the depth subsampling of the 1x1x1
//     stride-2 skip convolutions.
//   * depthwise 3x3x3 with the "same" padding of SeparableConv3d_same (deeplab3d.py:22-43) folded in:
//     forward, data gradient, weight gradient.  HBM-bound; one thread per output pixel x 16-byte channel vector.
//   * linear resize along depth (align_corners): trilinear interpolation = this followed by the 2-D bilinear kernel.
#include "common.h"

namespace {

inline unsigned grid1d(long long total) {
    long long g = (total + 255) / 256;
    if (g > 0x7fffffffLL) g = 0x7fffffffLL;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ------------------------------------------------------------------ depth unfold / fold
struct UnfoldParams {
    const void* x; void* y;      // x: [N][D][HW][ldx] (C used), y: [N][Do][HW][ldy] (KD*C used)
    int N, D, Do, HW, C, KD, stride, pad, dil, ldx, ldy;
};

template <typename T>
__global__ __launch_bounds__(256) void depth_unfold_kernel(UnfoldParams P) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = P.C / VEC;
    const long long per_slice = (long long)P.HW * P.KD * cv;
    const long long total = (long long)P.N * P.Do * per_slice;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long sl = i / per_slice;               // (n, od)
        long long r = i - sl * per_slice;
        const int p = (int)(r / (P.KD * cv));
        r -= (long long)p * (P.KD * cv);
        const int kd = (int)(r / cv), c = (int)(r - (long long)kd * cv) * VEC;
        const int n = (int)(sl / P.Do), od = (int)(sl - (long long)n * P.Do);
        const int id = od * P.stride - P.pad + kd * P.dil;
        Chunk<T> v;
        if ((unsigned)id < (unsigned)P.D)
            v.load(reinterpret_cast<const T*>(P.x) + (((long long)n * P.D + id) * P.HW + p) * P.ldx + c);
        else
            v.zero();
        v.store(reinterpret_cast<T*>(P.y) + (sl * P.HW + p) * P.ldy + kd * P.C + c);
    }
}

// dx[n, id, p, c] = sum over (od, kd) with od*stride - pad + kd*dil == id of dy[n, od, p, kd*C + c]
template <typename T>
__global__ __launch_bounds__(256) void depth_fold_kernel(UnfoldParams P) {   // x = dx (out), y = dy (in)
    constexpr int VEC = Elem<T>::VEC;
    const int cv = P.C / VEC;
    const long long per_slice = (long long)P.HW * cv;
    const long long total = (long long)P.N * P.D * per_slice;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long sl = i / per_slice;               // (n, id)
        const long long r = i - sl * per_slice;
        const int p = (int)(r / cv), c = (int)(r - (long long)p * cv) * VEC;
        const int n = (int)(sl / P.D), id = (int)(sl - (long long)n * P.D);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int kd = 0; kd < P.KD; ++kd) {
            const int t = id + P.pad - kd * P.dil;
            if (t < 0) continue;
            const int od = t / P.stride;
            if (od * P.stride != t || od >= P.Do) continue;
            Chunk<T> v;
            v.load(reinterpret_cast<const T*>(P.y) + (((long long)n * P.Do + od) * P.HW + p) * P.ldy + kd * P.C + c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += v.get(e);
        }
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
        o.store(reinterpret_cast<T*>(const_cast<void*>(P.x)) + (sl * P.HW + p) * P.ldx + c);
    }
}

// ------------------------------------------------------------------ depthwise 3x3x3
struct Dw3Params {
    const void* a; const void* b; void* o;   // fwd: x, w, y | bwd_data: dy, w, dx | bwd_weight: x, dy, dw(float)
    int N, D, H, W, C, Do, Ho, Wo, stride, dil, ldx, ldy;
};

template <typename T>
__global__ __launch_bounds__(256) void dw3_fwd_kernel(Dw3Params P) {
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
        const T* x = reinterpret_cast<const T*>(P.a) + (long long)n * P.D * P.H * P.W * P.ldx + c;
        const T* w = reinterpret_cast<const T*>(P.b) + c;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int kd = 0; kd < 3; ++kd) {
            const int id = od * P.stride - P.dil + kd * P.dil;
            if ((unsigned)id >= (unsigned)P.D) continue;
            for (int kh = 0; kh < 3; ++kh) {
                const int ih = oh * P.stride - P.dil + kh * P.dil;
                if ((unsigned)ih >= (unsigned)P.H) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int iw = ow * P.stride - P.dil + kw * P.dil;
                    if ((unsigned)iw >= (unsigned)P.W) continue;
                    Chunk<T> xv, wv;
                    xv.load(x + (((long long)id * P.H + ih) * P.W + iw) * P.ldx);
                    wv.load(w + ((kd * 3 + kh) * 3 + kw) * P.C);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] = fmaf(xv.get(e), wv.get(e), acc[e]);
                }
            }
        }
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
        o.store(reinterpret_cast<T*>(P.o) + pix * P.ldy + c);
    }
}

// dx[n,id,ih,iw,c] = sum_taps dy[n,od,oh,ow,c] * w[kd,kh,kw,c] with od*stride - dil + kd*dil == id (etc.)
template <typename T>
__global__ __launch_bounds__(256) void dw3_bwd_data_kernel(Dw3Params P) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = P.C / VEC;
    const long long total = (long long)P.N * P.D * P.H * P.W * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / cv;
        const int c = (int)(i - pix * cv) * VEC;
        long long r = pix;
        const int iw = (int)(r % P.W); r /= P.W;
        const int ih = (int)(r % P.H); r /= P.H;
        const int id = (int)(r % P.D);
        const int n = (int)(r / P.D);
        const T* dy = reinterpret_cast<const T*>(P.a) + (long long)n * P.Do * P.Ho * P.Wo * P.ldy + c;
        const T* w = reinterpret_cast<const T*>(P.b) + c;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int kd = 0; kd < 3; ++kd) {
            const int td = id + P.dil - kd * P.dil;
            if (td < 0) continue;
            const int od = td / P.stride;
            if (od * P.stride != td || od >= P.Do) continue;
            for (int kh = 0; kh < 3; ++kh) {
                const int th = ih + P.dil - kh * P.dil;
                if (th < 0) continue;
                const int oh = th / P.stride;
                if (oh * P.stride != th || oh >= P.Ho) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int tw = iw + P.dil - kw * P.dil;
                    if (tw < 0) continue;
                    const int ow = tw / P.stride;
                    if (ow * P.stride != tw || ow >= P.Wo) continue;
                    Chunk<T> gv, wv;
                    gv.load(dy + (((long long)od * P.Ho + oh) * P.Wo + ow) * P.ldy);
                    wv.load(w + ((kd * 3 + kh) * 3 + kw) * P.C);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] = fmaf(gv.get(e), wv.get(e), acc[e]);
                }
            }
        }
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
        o.store(reinterpret_cast<T*>(P.o) + pix * P.ldx + c);
    }
}

// The same two passes with every tap's load issued before the first is used.  The loops above skip taps outside the
// volume with branches, so each tap's load waits for the one before it: 27 L2 round trips in a row, ~13 us per launch on
// the 3 x 2 x 3 maps of the middle flow however little data there is.  Here a tap outside the volume is a buffer load at
// the out-of-range offset (returns zeros, no branch), the three loops are unrolled and the compiler has a depth plane's
// 18 loads in flight at once.  Same taps in the same order into the same fp32 accumulators: results as above.
// 32-bit offsets over the whole tensor (the launcher keeps the pointer kernels for tensors of 2 GiB and more).
template <typename T, bool DATA>
__global__ __launch_bounds__(256) void dw3_taps_kernel(Dw3Params P, int a_bytes) {
    constexpr int VEC = Elem<T>::VEC, ES = (int)sizeof(T);
    typedef typename Elem<T>::vec_t vec_t;
    const int cv = P.C / VEC;
    // forward: the grid runs over the output (Do, Ho, Wo) and reads x (D, H, W); data gradient: over (D, H, W), reads dy
    const int GD = DATA ? P.D : P.Do, GH = DATA ? P.H : P.Ho, GW = DATA ? P.W : P.Wo;
    const int SD = DATA ? P.Do : P.D, SH = DATA ? P.Ho : P.H, SW = DATA ? P.Wo : P.W;
    const int ld_src = DATA ? P.ldy : P.ldx, ldo = DATA ? P.ldx : P.ldy;
    const long long total = (long long)P.N * GD * GH * GW * cv;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.a), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.b), 0, 27 * P.C * ES, 0x00020000);
    const int s_w = ld_src * ES, s_h = SW * s_w, s_d = SH * s_h;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / cv;
        const int c = (int)(i - pix * cv) * VEC;
        long long r = pix;
        const int gw = (int)(r % GW); r /= GW;
        const int gh = (int)(r % GH); r /= GH;
        const int gd = (int)(r % GD);
        const int n = (int)(r / GD);
        int off[3][3];      // [axis][tap]: byte offset along the axis, or < 0 for a tap outside the volume
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int sd, sh, sw;
            if (!DATA) {
                sd = gd * P.stride - P.dil + k * P.dil;
                sh = gh * P.stride - P.dil + k * P.dil;
                sw = gw * P.stride - P.dil + k * P.dil;
            } else {          // od * stride - dil + k * dil == id
                const int td = gd + P.dil - k * P.dil, th = gh + P.dil - k * P.dil, tw = gw + P.dil - k * P.dil;
                sd = td / P.stride; sh = th / P.stride; sw = tw / P.stride;
                if (td < 0 || sd * P.stride != td) sd = -1;
                if (th < 0 || sh * P.stride != th) sh = -1;
                if (tw < 0 || sw * P.stride != tw) sw = -1;
            }
            off[0][k] = (unsigned)sd < (unsigned)SD ? sd * s_d : -1;
            off[1][k] = (unsigned)sh < (unsigned)SH ? sh * s_h : -1;
            off[2][k] = (unsigned)sw < (unsigned)SW ? sw * s_w : -1;
        }
        const int base = n * SD * s_d + c * ES;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            Chunk<T> av[9], wv[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int kh = t / 3, kw = t % 3;
                const int o = (off[0][kd] | off[1][kh] | off[2][kw]) < 0 ? (int)0x80000000
                                                                         : base + off[0][kd] + off[1][kh] + off[2][kw];
                av[t].v = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs_a, o, 0, 0));
                wv[t].v = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, ((kd * 9 + t) * P.C + c) * ES, 0, 0));
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf(av[t].get(e), wv[t].get(e), acc[e]);
        }
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
        o.store(reinterpret_cast<T*>(P.o) + pix * ldo + c);
    }
}

// dw[kd,kh,kw,c] += sum dy * x.  Block = TX channel vectors x TY pixel lanes over a run of output (n,od,oh) rows;
// one depth tap plane (9 taps) at a time keeps 9*VEC partials per thread; LDS fold, one float atomic per (tap, channel).
struct Dw3WParams {
    const void* x; const void* dy; float* dw;
    int N, D, H, W, C, Do, Ho, Wo, stride, dil, ldx, ldy;
    int rows_total, rows_per_block, tx, log_tx;
};

// BUF: the nine x loads of a gradient chunk as buffer loads (a tap outside the plane = the out-of-range offset, zeros) issued
// together with it instead of one after the other behind branches; x_bytes / dy_bytes < 2 GiB (launcher).
template <typename T, bool BUF>
__global__ __launch_bounds__(256) void dw3_bwd_weight_kernel(Dw3WParams P, int x_bytes, int dy_bytes) {
    constexpr int VEC = Elem<T>::VEC, ES = (int)sizeof(T);
    typedef typename Elem<T>::vec_t vec_t;
    __shared__ float red[256 * VEC];
    const int lx = threadIdx.x & (P.tx - 1);
    const int ly = threadIdx.x >> P.log_tx;
    const int ty = 256 >> P.log_tx;
    const int c = (blockIdx.x * P.tx + lx) * VEC;
    const bool c_ok = c < P.C;
    const int kd = blockIdx.z;
    const T* x = reinterpret_cast<const T*>(P.x) + c;
    const T* dy = reinterpret_cast<const T*>(P.dy) + c;
    float acc[9][VEC];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
    const int row0 = blockIdx.y * P.rows_per_block;
    int row1 = row0 + P.rows_per_block;
    if (row1 > P.rows_total) row1 = P.rows_total;
    if (BUF) {
        // a block takes a run of rows_per_block output PIXELS (not rows: on the 3 x 2 x 3 maps a row is three pixels wide and
        // the block's pixel lanes would idle); a pixel whose depth tap falls outside the volume loads zeros like any other
        // out-of-range tap
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.x), 0, x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.dy), 0, dy_bytes, 0x00020000);
        const int s_w = P.ldx * ES, s_h = P.W * s_w;
        const int pix_total = P.rows_total * P.Wo;
        const int p0 = blockIdx.y * P.rows_per_block;
        const int p1 = min(p0 + P.rows_per_block, pix_total);
#pragma unroll 2
        for (int p = p0 + ly; p < p1; p += ty) {
            const int ow = p % P.Wo, row = p / P.Wo;
            const int oh = row % P.Ho, q = row / P.Ho;
            const int od = q % P.Do, n = q / P.Do;
            const int id = od * P.stride - P.dil + kd * P.dil;
            const bool live = c_ok && (unsigned)id < (unsigned)P.D;
            const int x_plane = (n * P.D + id) * P.H * s_h + c * ES;
            Chunk<T> gv, xv[9];
            gv.v = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs_g, live ? (p * P.ldy + c) * ES : (int)0x80000000, 0, 0));
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int kh = t / 3, kw = t % 3;
                const int ih = oh * P.stride - P.dil + kh * P.dil, iw = ow * P.stride - P.dil + kw * P.dil;
                const bool ok = live && (unsigned)ih < (unsigned)P.H && (unsigned)iw < (unsigned)P.W;
                xv[t].v = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? x_plane + ih * s_h + iw * s_w : (int)0x80000000, 0, 0));
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[t][e] = fmaf(gv.get(e), xv[t].get(e), acc[t][e]);
        }
    } else if (c_ok) {
        for (int row = row0; row < row1; ++row) {      // row = (n, od, oh), block-uniform
            const int oh = row % P.Ho;
            const int q = row / P.Ho;
            const int od = q % P.Do, n = q / P.Do;
            const int id = od * P.stride - P.dil + kd * P.dil;
            if ((unsigned)id >= (unsigned)P.D) continue;
            const T* dyr = dy + (long long)row * P.Wo * P.ldy;
            const T* xs = x + ((long long)n * P.D + id) * P.H * P.W * P.ldx;
            for (int ow = ly; ow < P.Wo; ow += ty) {
                Chunk<T> gv;
                gv.load(dyr + (long long)ow * P.ldy);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int ih = oh * P.stride - P.dil + kh * P.dil;
                    if ((unsigned)ih >= (unsigned)P.H) continue;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int iw = ow * P.stride - P.dil + kw * P.dil;
                        if ((unsigned)iw >= (unsigned)P.W) continue;
                        Chunk<T> xv;
                        xv.load(xs + ((long long)ih * P.W + iw) * P.ldx);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[kh * 3 + kw][e] = fmaf(gv.get(e), xv.get(e), acc[kh * 3 + kw][e]);
                    }
                }
            }
        }
    }
    const int row_w = P.tx * VEC;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < VEC; ++e) red[ly * row_w + lx * VEC + e] = acc[t][e];
        __syncthreads();
        for (int i = threadIdx.x; i < row_w; i += 256) {
            float a = 0.f;
            for (int y = 0; y < ty; ++y) a += red[y * row_w + i];
            const int ch = blockIdx.x * row_w + i;
            if (ch < P.C) atomicAdd(P.dw + (long long)(kd * 9 + t) * P.C + ch, a);
        }
    }
}

// ------------------------------------------------------------------ linear resize along depth
struct DResizeParams {
    const void* x; void* y;   // x: [N][Di][HW][ldx], y: [N][Do][HW][ldy]
    int N, Di, Do, HW, C, ldx, ldy;
};

__device__ __forceinline__ void depth_src(int od, int Di, int Do, int& d0, float& t) {
    // align_corners=True: src = od * (Di-1)/(Do-1); a one-slice output samples slice 0
    const float s = Do > 1 ? (float)od * ((float)(Di - 1) / (float)(Do - 1)) : 0.f;
    d0 = (int)s;
    if (d0 > Di - 1) d0 = Di - 1;
    t = s - (float)d0;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void depth_resize_fwd_kernel(DResizeParams P) {
    // 4 channels per thread (the common granule of both dtypes' 16-byte vectors is not needed here: scalar-ish)
    const int c4 = P.C / 4;
    const long long total = (long long)P.N * P.Do * P.HW * c4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / c4;
        const int c = (int)(i - pix * c4) * 4;
        const int p = (int)(pix % P.HW);
        const long long sl = pix / P.HW;
        const int od = (int)(sl % P.Do), n = (int)(sl / P.Do);
        int d0; float t;
        depth_src(od, P.Di, P.Do, d0, t);
        const int d1 = d0 + 1 < P.Di ? d0 + 1 : d0;
        const TI* a = reinterpret_cast<const TI*>(P.x) + (((long long)n * P.Di + d0) * P.HW + p) * P.ldx + c;
        const TI* b = reinterpret_cast<const TI*>(P.x) + (((long long)n * P.Di + d1) * P.HW + p) * P.ldx + c;
        TO* o = reinterpret_cast<TO*>(P.y) + pix * P.ldy + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = Elem<TO>::from_f((1.f - t) * Elem<TI>::to_f(a[e]) + t * Elem<TI>::to_f(b[e]));
    }
}

// adjoint: dx[n, id] = sum_od dy[n, od] * ((d0(od) == id) * (1 - t) + (d1(od) == id) * t); depths are small (<= a few dozen)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void depth_resize_bwd_kernel(DResizeParams P) {   // x = dx (out, TO), y = dy (in, TI)
    const int c4 = P.C / 4;
    const long long total = (long long)P.N * P.Di * P.HW * c4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / c4;
        const int c = (int)(i - pix * c4) * 4;
        const int p = (int)(pix % P.HW);
        const long long sl = pix / P.HW;
        const int id = (int)(sl % P.Di), n = (int)(sl / P.Di);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int od = 0; od < P.Do; ++od) {
            int d0; float t;
            depth_src(od, P.Di, P.Do, d0, t);
            const int d1 = d0 + 1 < P.Di ? d0 + 1 : d0;
            float wgt = 0.f;
            if (d0 == id) wgt += 1.f - t;
            if (d1 == id) wgt += t;
            if (wgt == 0.f) continue;
            const TI* g = reinterpret_cast<const TI*>(P.y) + (((long long)n * P.Do + od) * P.HW + p) * P.ldy + c;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(wgt, Elem<TI>::to_f(g[e]), acc[e]);
        }
        TO* o = reinterpret_cast<TO*>(const_cast<void*>(P.x)) + pix * P.ldx + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = Elem<TO>::from_f(acc[e]);
    }
}

int check_dw3(const bg_dwconv3d_desc* d, const char* who) {
    BG_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
    BG_CHECK_ARG(dtype_ok(d->dtype), "%s: bad dtype", who);
    const int vec = dtype_vec(d->dtype);
    BG_CHECK_ARG(d->N > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->C > 0, "%s: non-positive dimension", who);
    BG_CHECK_ARG(d->stride >= 1 && d->dil >= 1, "%s: bad stride/dil", who);
    BG_CHECK_ARG(d->C % vec == 0 && d->ldx % vec == 0 && d->ldy % vec == 0 && d->ldx >= d->C && d->ldy >= d->C,
                 "%s: C/ld must be multiples of %d", who, vec);
    const int s = d->stride;
    BG_CHECK_ARG(d->Do == (d->D + s - 1) / s && d->Ho == (d->H + s - 1) / s && d->Wo == (d->W + s - 1) / s,
                 "%s: Do/Ho/Wo must be ceil(D/stride), ceil(H/stride), ceil(W/stride)", who);
    return BG_OK;
}


// Depth half of nn.AvgPool3d(2, stride 1) (the 2 x 2 in-plane half is bg_avgpool2x2): y[n, od] = 0.5 * (x[n, od + off]
// + x[n, od + off + 1]), zero outside the volume (count_include_pad).  off = -padding is the forward pass,
// off = padding - 1 with the roles of x and y exchanged its adjoint.
template <typename T>
__global__ __launch_bounds__(256) void depth_avg2_kernel(const T* x, int ldx, T* y, int ldy, int N, int Di, int Do, int HW, int C,
                                                         int off) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const long long total = (long long)N * Do * HW * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / cv;
        const int c = (int)(i - pix * cv) * VEC;
        const int p = (int)(pix % HW);
        const long long sl = pix / HW;
        const int od = (int)(sl % Do), n = (int)(sl / Do);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int id = od + off + t;
            if ((unsigned)id >= (unsigned)Di) continue;
            Chunk<T> v;
            v.load(x + (((long long)n * Di + id) * HW + p) * ldx + c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += v.get(e);
        }
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, 0.5f * acc[e]);
        o.store(y + pix * ldy + c);
    }
}

}  // namespace

extern "C" int bg_depth_unfold(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int32_t N, int32_t D,
                               int32_t Do, int32_t HW, int32_t C, int32_t KD, int32_t stride, int32_t pad, int32_t dil,
                               void* stream) {
    BG_CHECK_ARG(dtype_ok(dtype) && x && y && aligned16(x) && aligned16(y), "bg_depth_unfold: bad dtype / pointer");
    const int vec = dtype_vec(dtype);
    BG_CHECK_ARG(N > 0 && D > 0 && Do > 0 && HW > 0 && C > 0 && KD >= 1 && stride >= 1 && dil >= 1 && pad >= 0,
                 "bg_depth_unfold: bad sizes");
    BG_CHECK_ARG(C % vec == 0 && ldx % vec == 0 && ldy % vec == 0 && ldx >= C && ldy >= KD * C,
                 "bg_depth_unfold: C/ld must be multiples of %d, ldy >= KD*C", vec);
    BG_CHECK_ARG(Do == (D + 2 * pad - dil * (KD - 1) - 1) / stride + 1, "bg_depth_unfold: Do does not match the conv arithmetic");
    UnfoldParams P{x, y, N, D, Do, HW, C, KD, stride, pad, dil, ldx, ldy};
    const long long total = (long long)N * Do * HW * KD * (C / vec);
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((depth_unfold_kernel<T>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, P));
    BG_CHECK_LAUNCH("depth_unfold_kernel");
    return BG_OK;
}

extern "C" int bg_depth_fold(int32_t dtype, const void* dy, int32_t lddy, void* dx, int32_t lddx, int32_t N, int32_t D,
                             int32_t Do, int32_t HW, int32_t C, int32_t KD, int32_t stride, int32_t pad, int32_t dil,
                             void* stream) {
    BG_CHECK_ARG(dtype_ok(dtype) && dy && dx && aligned16(dy) && aligned16(dx), "bg_depth_fold: bad dtype / pointer");
    const int vec = dtype_vec(dtype);
    BG_CHECK_ARG(N > 0 && D > 0 && Do > 0 && HW > 0 && C > 0 && KD >= 1 && stride >= 1 && dil >= 1 && pad >= 0,
                 "bg_depth_fold: bad sizes");
    BG_CHECK_ARG(C % vec == 0 && lddx % vec == 0 && lddy % vec == 0 && lddx >= C && lddy >= KD * C,
                 "bg_depth_fold: C/ld must be multiples of %d, lddy >= KD*C", vec);
    UnfoldParams P{dx, const_cast<void*>(dy), N, D, Do, HW, C, KD, stride, pad, dil, lddx, lddy};
    const long long total = (long long)N * D * HW * (C / vec);
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((depth_fold_kernel<T>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, P));
    BG_CHECK_LAUNCH("depth_fold_kernel");
    return BG_OK;
}

static bool dw3_taps_on() {   // BGAMD_DW3_TAPS=0: the pointer kernels (A/B)
    static const bool on = !(getenv("BGAMD_DW3_TAPS") && atoi(getenv("BGAMD_DW3_TAPS")) == 0);
    return on;
}

extern "C" int bg_dwconv3x3x3_fwd(const bg_dwconv3d_desc* d, const void* x, const void* w, void* y, void* stream) {
    int rc = check_dw3(d, "bg_dwconv3x3x3_fwd");
    if (rc) return rc;
    BG_CHECK_ARG(x && w && y && aligned16(x) && aligned16(w) && aligned16(y), "bg_dwconv3x3x3_fwd: null/unaligned pointer");
    Dw3Params P{x, w, y, d->N, d->D, d->H, d->W, d->C, d->Do, d->Ho, d->Wo, d->stride, d->dil, d->ldx, d->ldy};
    const long long total = (long long)d->N * d->Do * d->Ho * d->Wo * (d->C / dtype_vec(d->dtype));
    const long long x_bytes = (((long long)d->N * d->D * d->H * d->W - 1) * d->ldx + d->C) * dtype_size(d->dtype);
    if (dw3_taps_on() && x_bytes < 0x7fffffffLL)
        BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw3_taps_kernel<T, false>), dim3(grid1d(total)), dim3(256), 0,
                                                          (hipStream_t)stream, P, (int)x_bytes));
    else
        BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw3_fwd_kernel<T>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, P));
    BG_CHECK_LAUNCH("dw3_fwd_kernel");
    return BG_OK;
}

extern "C" int bg_dwconv3x3x3_bwd_data(const bg_dwconv3d_desc* d, const void* dy, const void* w, void* dx, void* stream) {
    int rc = check_dw3(d, "bg_dwconv3x3x3_bwd_data");
    if (rc) return rc;
    BG_CHECK_ARG(dy && w && dx && aligned16(dy) && aligned16(w) && aligned16(dx), "bg_dwconv3x3x3_bwd_data: null/unaligned pointer");
    Dw3Params P{dy, w, dx, d->N, d->D, d->H, d->W, d->C, d->Do, d->Ho, d->Wo, d->stride, d->dil, d->ldx, d->ldy};
    const long long total = (long long)d->N * d->D * d->H * d->W * (d->C / dtype_vec(d->dtype));
    const long long dy_bytes = (((long long)d->N * d->Do * d->Ho * d->Wo - 1) * d->ldy + d->C) * dtype_size(d->dtype);
    if (dw3_taps_on() && dy_bytes < 0x7fffffffLL)
        BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw3_taps_kernel<T, true>), dim3(grid1d(total)), dim3(256), 0,
                                                          (hipStream_t)stream, P, (int)dy_bytes));
    else
        BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw3_bwd_data_kernel<T>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, P));
    BG_CHECK_LAUNCH("dw3_bwd_data_kernel");
    return BG_OK;
}

extern "C" int bg_dwconv3x3x3_bwd_weight(const bg_dwconv3d_desc* d, const void* x, const void* dy, float* dw, void* stream) {
    int rc = check_dw3(d, "bg_dwconv3x3x3_bwd_weight");
    if (rc) return rc;
    BG_CHECK_ARG(x && dy && dw && aligned16(x) && aligned16(dy), "bg_dwconv3x3x3_bwd_weight: null/unaligned pointer");
    Dw3WParams P{x, dy, dw, d->N, d->D, d->H, d->W, d->C, d->Do, d->Ho, d->Wo, d->stride, d->dil, d->ldx, d->ldy, 0, 0, 0, 0};
    const int cv = d->C / dtype_vec(d->dtype);
    int best = 16, best_pad = 1 << 30;
    for (int tx = 16; tx <= 64; tx *= 2) {
        const int pad = (cv + tx - 1) / tx * tx;
        if (pad <= best_pad) { best_pad = pad; best = tx; }
    }
    P.tx = best;
    P.log_tx = best == 16 ? 4 : (best == 32 ? 5 : 6);
    const int gx = (cv + best - 1) / best;
    P.rows_total = d->N * d->Do * d->Ho;
    int gy = 512 / gx;
    if (gy < 1) gy = 1;
    if (gy > P.rows_total) gy = P.rows_total;
    P.rows_per_block = (P.rows_total + gy - 1) / gy;
    gy = (P.rows_total + P.rows_per_block - 1) / P.rows_per_block;
    const long long x_bytes = (((long long)d->N * d->D * d->H * d->W - 1) * d->ldx + d->C) * dtype_size(d->dtype);
    const long long dy_bytes = (((long long)d->N * d->Do * d->Ho * d->Wo - 1) * d->ldy + d->C) * dtype_size(d->dtype);
    if (dw3_taps_on() && x_bytes < 0x7fffffffLL && dy_bytes < 0x7fffffffLL) {
        // pixel runs: about 512 / gx blocks per depth tap, at least 4 pixels for each of a block's 256 / tx pixel lanes
        Dw3WParams Q = P;
        const long long pix = (long long)P.rows_total * d->Wo;
        const int ty = 256 / best;
        long long per = (pix + 512 / gx - 1) / std::max(1, 512 / gx);
        per = std::max<long long>(per, 4 * ty);
        per = (per + ty - 1) / ty * ty;
        Q.rows_per_block = (int)per;
        const unsigned gyp = (unsigned)((pix + per - 1) / per);
        BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw3_bwd_weight_kernel<T, true>), dim3(gx, gyp, 3), dim3(256), 0,
                                                          (hipStream_t)stream, Q, (int)x_bytes, (int)dy_bytes));
    }
    else
        BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw3_bwd_weight_kernel<T, false>), dim3(gx, (unsigned)gy, 3), dim3(256), 0,
                                                          (hipStream_t)stream, P, 0, 0));
    BG_CHECK_LAUNCH("dw3_bwd_weight_kernel");
    return BG_OK;
}

#define BG_DISPATCH2(dta, dtb, TA, TB, ...)                         \
    do {                                                            \
        if ((dta) == BG_BF16 && (dtb) == BG_BF16) { typedef bf16_t TA; typedef bf16_t TB; __VA_ARGS__; } \
        else if ((dta) == BG_BF16) { typedef bf16_t TA; typedef float TB; __VA_ARGS__; }               \
        else if ((dtb) == BG_BF16) { typedef float TA; typedef bf16_t TB; __VA_ARGS__; }               \
        else { typedef float TA; typedef float TB; __VA_ARGS__; }   \
    } while (0)

extern "C" int bg_depth_resize_fwd(int32_t in_dtype, int32_t out_dtype, const void* x, int32_t ldx, void* y, int32_t ldy,
                                   int32_t N, int32_t Di, int32_t Do, int32_t HW, int32_t C, void* stream) {
    BG_CHECK_ARG(dtype_ok(in_dtype) && dtype_ok(out_dtype) && x && y && N > 0 && Di > 0 && Do > 0 && HW > 0 && C > 0,
                 "bg_depth_resize_fwd: bad args");
    BG_CHECK_ARG(C % 4 == 0 && ldx >= C && ldy >= C, "bg_depth_resize_fwd: C must be a multiple of 4, ld >= C");
    DResizeParams P{x, y, N, Di, Do, HW, C, ldx, ldy};
    const long long total = (long long)N * Do * HW * (C / 4);
    BG_DISPATCH2(in_dtype, out_dtype, TI, TO,
                 hipLaunchKernelGGL((depth_resize_fwd_kernel<TI, TO>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, P));
    BG_CHECK_LAUNCH("depth_resize_fwd_kernel");
    return BG_OK;
}

extern "C" int bg_depth_resize_bwd(int32_t dy_dtype, int32_t dx_dtype, const void* dy, int32_t lddy, void* dx, int32_t lddx,
                                   int32_t N, int32_t Di, int32_t Do, int32_t HW, int32_t C, void* stream) {
    BG_CHECK_ARG(dtype_ok(dy_dtype) && dtype_ok(dx_dtype) && dy && dx && N > 0 && Di > 0 && Do > 0 && HW > 0 && C > 0,
                 "bg_depth_resize_bwd: bad args");
    BG_CHECK_ARG(C % 4 == 0 && lddx >= C && lddy >= C, "bg_depth_resize_bwd: C must be a multiple of 4, ld >= C");
    DResizeParams P{dx, const_cast<void*>(dy), N, Di, Do, HW, C, lddx, lddy};
    const long long total = (long long)N * Di * HW * (C / 4);
    BG_DISPATCH2(dy_dtype, dx_dtype, TI, TO,
                 hipLaunchKernelGGL((depth_resize_bwd_kernel<TI, TO>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, P));
    BG_CHECK_LAUNCH("depth_resize_bwd_kernel");
    return BG_OK;
}

extern "C" int bg_depth_avg2(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int32_t N, int32_t Di, int32_t Do,
                             int32_t HW, int32_t C, int32_t off, void* stream) {
    BG_CHECK_ARG(dtype_ok(dtype) && x && y && aligned16(x) && aligned16(y) && N > 0 && Di > 0 && Do > 0 && HW > 0 && C > 0,
                 "bg_depth_avg2: bad args");
    const int vec = dtype_vec(dtype);
    BG_CHECK_ARG(C % vec == 0 && ldx % vec == 0 && ldy % vec == 0 && ldx >= C && ldy >= C, "bg_depth_avg2: C/ld must be multiples of %d", vec);
    const long long total = (long long)N * Do * HW * (C / vec);
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((depth_avg2_kernel<T>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)x, ldx, (T*)y, ldy, N, Di, Do, HW, C, off));
    BG_CHECK_LAUNCH("depth_avg2_kernel");
    return BG_OK;
}
