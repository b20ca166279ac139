// Depthwise 3x3 convolution of SeparableConv2d_same with the "same" zero
// padding folded into the index arithmetic (no padded copy is materialised).
// HBM-bound: one thread owns one output pixel x one 16-byte channel vector;
// lanes run along channels first, so every load/store is a full 16 B per lane
// and consecutive lanes touch consecutive addresses.
#include "common.h"

namespace {

struct DwParams {
    const void* x;
    const void* w;  // [3][3][C]
    void* y;
    int N, H, W, C, Ho, Wo, stride, dil, ldx, ldy;
    long long total;  // N*Ho*Wo*(C/VEC)   (fwd)   or N*H*W*(C/VEC) (bwd_data)
};

template <typename T>
__global__ void dw_fwd_kernel(DwParams P) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = P.C / VEC;
    const T* x = reinterpret_cast<const T*>(P.x);
    const T* w = reinterpret_cast<const T*>(P.w);
    T* y = reinterpret_cast<T*>(P.y);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P.total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VEC;
        long long pix = i / cv;
        const int wo = (int)(pix % P.Wo);
        pix /= P.Wo;
        const int ho = (int)(pix % P.Ho);
        const int n = (int)(pix / P.Ho);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ih = ho * P.stride - P.dil + r * P.dil;
            if ((unsigned)ih >= (unsigned)P.H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int iw = wo * P.stride - P.dil + s * P.dil;
                if ((unsigned)iw >= (unsigned)P.W) continue;
                Chunk<T> xv, wv;
                xv.load(x + (((long long)n * P.H + ih) * P.W + iw) * P.ldx + c);
                wv.load(w + (r * 3 + s) * P.C + c);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf(xv.get(e), wv.get(e), acc[e]);
            }
        }
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
        o.store(y + (((long long)n * P.Ho + ho) * P.Wo + wo) * P.ldy + c);
    }
}

// dx[n,h,w,c] = sum_{r,s} dy[n,ho,wo,c] * w[r,s,c]  with ho*stride - dil + r*dil == h
template <typename T>
__global__ void dw_bwd_data_kernel(DwParams P) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = P.C / VEC;
    const T* dy = reinterpret_cast<const T*>(P.x);
    const T* w = reinterpret_cast<const T*>(P.w);
    T* dx = reinterpret_cast<T*>(P.y);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P.total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VEC;
        long long pix = i / cv;
        const int iw = (int)(pix % P.W);
        pix /= P.W;
        const int ih = (int)(pix % P.H);
        const int n = (int)(pix / P.H);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int th = ih + P.dil - r * P.dil;
            if (th < 0) continue;
            const int ho = th / P.stride;
            if (ho * P.stride != th || ho >= P.Ho) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int tw = iw + P.dil - s * P.dil;
                if (tw < 0) continue;
                const int wo = tw / P.stride;
                if (wo * P.stride != tw || wo >= P.Wo) continue;
                Chunk<T> gv, wv;
                gv.load(dy + (((long long)n * P.Ho + ho) * P.Wo + wo) * P.ldy + c);
                wv.load(w + (r * 3 + s) * P.C + c);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf(gv.get(e), wv.get(e), acc[e]);
            }
        }
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
        o.store(dx + (((long long)n * P.H + ih) * P.W + iw) * P.ldx + c);
    }
}

// dw[r,s,c] += sum_{n,ho,wo} dy[n,ho,wo,c] * x[n, ho*st-dil+r*dil, wo*st-dil+s*dil, c]
// block = 16 channel-vectors (x) by 16 pixel lanes (y); each thread walks a strip
// of output pixels with 9*VEC fp32 accumulators, then the 4 pixel lanes of a wave
// are folded with shuffles and every wave issues one float atomic per (tap, channel).
struct DwWParams {
    const void* x;
    const void* dy;
    float* dw;
    int N, H, W, C, Ho, Wo, stride, dil, ldx, ldy;
    long long M;  // N*Ho*Wo
    int pix_per_block;
};

template <typename T>
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(DwWParams P) {
    constexpr int VEC = Elem<T>::VEC;
    const int cvi = blockIdx.x * 16 + (threadIdx.x & 15);
    const int plane = threadIdx.x >> 4;  // 0..15
    const int c = cvi * VEC;
    const bool c_ok = c < P.C;
    const T* x = reinterpret_cast<const T*>(P.x);
    const T* dy = reinterpret_cast<const T*>(P.dy);
    float acc[9][VEC];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
    const long long p0 = (long long)blockIdx.y * P.pix_per_block;
    long long p1 = p0 + P.pix_per_block;
    if (p1 > P.M) p1 = P.M;
    if (c_ok) {
        for (long long p = p0 + plane; p < p1; p += 16) {
            long long t = p;
            const int wo = (int)(t % P.Wo);
            t /= P.Wo;
            const int ho = (int)(t % P.Ho);
            const int n = (int)(t / P.Ho);
            Chunk<T> gv;
            gv.load(dy + p * P.ldy + c);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int ih = ho * P.stride - P.dil + r * P.dil;
                if ((unsigned)ih >= (unsigned)P.H) continue;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int iw = wo * P.stride - P.dil + s * P.dil;
                    if ((unsigned)iw >= (unsigned)P.W) continue;
                    Chunk<T> xv;
                    xv.load(x + (((long long)n * P.H + ih) * P.W + iw) * P.ldx + c);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[r * 3 + s][e] = fmaf(gv.get(e), xv.get(e), acc[r * 3 + s][e]);
                }
            }
        }
    }
    // lanes l, l^16, l^32, l^48 of a wave hold the same channel vector
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float v = acc[t][e];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[t][e] = v;
        }
    if (c_ok && (threadIdx.x & 63) < 16) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < VEC; ++e) atomicAdd(P.dw + (long long)t * P.C + c + e, acc[t][e]);
    }
}

int check_dw(const bg_dwconv_desc* d, const char* who) {
    BG_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
    BG_CHECK_ARG(dtype_ok(d->dtype), "%s: bad dtype", who);
    const int vec = dtype_vec(d->dtype);
    BG_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0, "%s: non-positive dimension", who);
    BG_CHECK_ARG(d->stride >= 1 && d->dil >= 1, "%s: bad stride/dil", who);
    BG_CHECK_ARG(d->C % vec == 0 && d->ldx % vec == 0 && d->ldy % vec == 0 && d->ldx >= d->C && d->ldy >= d->C,
                 "%s: C/ld must be multiples of %d", who, vec);
    BG_CHECK_ARG(d->Ho == (d->H + d->stride - 1) / d->stride && d->Wo == (d->W + d->stride - 1) / d->stride,
                 "%s: Ho/Wo must be ceil(H/stride), ceil(W/stride)", who);
    return BG_OK;
}

inline unsigned grid_for(long long total, int block) {
    long long g = (total + block - 1) / block;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int bg_dwconv3x3_fwd(const bg_dwconv_desc* d, const void* x, const void* w, void* y, void* stream) {
    int rc = check_dw(d, "bg_dwconv3x3_fwd");
    if (rc) return rc;
    BG_CHECK_ARG(x && w && y && aligned16(x) && aligned16(w) && aligned16(y), "bg_dwconv3x3_fwd: null/unaligned pointer");
    DwParams P{x, w, y, d->N, d->H, d->W, d->C, d->Ho, d->Wo, d->stride, d->dil, d->ldx, d->ldy, 0};
    P.total = (long long)d->N * d->Ho * d->Wo * (d->C / dtype_vec(d->dtype));
    hipStream_t st = (hipStream_t)stream;
    BG_DISPATCH_DTYPE(d->dtype, T,
                      hipLaunchKernelGGL((dw_fwd_kernel<T>), dim3(grid_for(P.total, 256)), dim3(256), 0, st, P));
    BG_CHECK_LAUNCH("dw_fwd_kernel");
    return BG_OK;
}

extern "C" int bg_dwconv3x3_bwd_data(const bg_dwconv_desc* d, const void* dy, const void* w, void* dx, void* stream) {
    int rc = check_dw(d, "bg_dwconv3x3_bwd_data");
    if (rc) return rc;
    BG_CHECK_ARG(dy && w && dx && aligned16(dy) && aligned16(w) && aligned16(dx),
                 "bg_dwconv3x3_bwd_data: null/unaligned pointer");
    DwParams P{dy, w, dx, d->N, d->H, d->W, d->C, d->Ho, d->Wo, d->stride, d->dil, d->ldx, d->ldy, 0};
    P.total = (long long)d->N * d->H * d->W * (d->C / dtype_vec(d->dtype));
    hipStream_t st = (hipStream_t)stream;
    BG_DISPATCH_DTYPE(d->dtype, T,
                      hipLaunchKernelGGL((dw_bwd_data_kernel<T>), dim3(grid_for(P.total, 256)), dim3(256), 0, st, P));
    BG_CHECK_LAUNCH("dw_bwd_data_kernel");
    return BG_OK;
}

extern "C" int bg_dwconv3x3_bwd_weight(const bg_dwconv_desc* d, const void* x, const void* dy, float* dw,
                                       void* stream) {
    int rc = check_dw(d, "bg_dwconv3x3_bwd_weight");
    if (rc) return rc;
    BG_CHECK_ARG(x && dy && dw && aligned16(x) && aligned16(dy), "bg_dwconv3x3_bwd_weight: null/unaligned pointer");
    DwWParams P{x, dy, dw, d->N, d->H, d->W, d->C, d->Ho, d->Wo, d->stride, d->dil, d->ldx, d->ldy, 0, 0};
    P.M = (long long)d->N * d->Ho * d->Wo;
    const int vec = dtype_vec(d->dtype);
    const int gx = (d->C / vec + 15) / 16;
    // ~2048 blocks in total, at least 256 pixels each
    long long gy = 2048 / gx;
    if (gy < 1) gy = 1;
    long long ppb = (P.M + gy - 1) / gy;
    if (ppb < 256) ppb = 256;
    P.pix_per_block = (int)ppb;
    gy = (P.M + ppb - 1) / ppb;
    BG_CHECK_ARG(gy <= 65535, "bg_dwconv3x3_bwd_weight: grid too large");
    hipStream_t st = (hipStream_t)stream;
    BG_DISPATCH_DTYPE(d->dtype, T,
                      hipLaunchKernelGGL((dw_bwd_weight_kernel<T>), dim3(gx, (unsigned)gy), dim3(256), 0, st, P));
    BG_CHECK_LAUNCH("dw_bwd_weight_kernel");
    return BG_OK;
}
