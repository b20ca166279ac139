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
    int items;  // Wo*(C/VEC) (fwd) or W*(C/VEC) (bwd_data): work items of one image row
    int bx;     // 256-thread blocks per image row; the grid is 1-D: rows * bx blocks
};

// 1-D grid -> (image row, block inside the row).  Blocks with the same id % 8 share an
// XCD and its L2; give every XCD a CONTIGUOUS band of rows, so the three output rows
// that read one input row find it in the same L2 (with the hardware's round-robin
// order every input row was fetched from HBM by three different XCDs: measured
// 3.2x the algorithmic read traffic).
__device__ __forceinline__ void dw_block_to_row(int bx, int& row, int& xblk) {
    const int nblk = gridDim.x;
    int lin = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = lin & 7, k = lin >> 3;
    lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    row = lin / bx;
    xblk = lin - row * bx;
}

// grid: y = (n, output row), x = 256-thread blocks over (wo, channel vector) with
// the channel vector fastest.  One 32-bit division per thread, none per tap.
template <typename T>
__global__ __launch_bounds__(256) void dw_fwd_kernel(DwParams P) {
    constexpr int VEC = Elem<T>::VEC;
    const unsigned cv = P.C / VEC;
    int grow, xblk;
    dw_block_to_row(P.bx, grow, xblk);
    const unsigned idx = xblk * 256u + threadIdx.x;
    if (idx >= (unsigned)P.items) return;
    const unsigned wo = idx / cv;
    const int c = (int)(idx - wo * cv) * VEC;
    const int n = grow / P.Ho, ho = grow - n * P.Ho;
    const T* x = reinterpret_cast<const T*>(P.x) + (long long)n * P.H * P.W * P.ldx + c;
    const T* w = reinterpret_cast<const T*>(P.w) + c;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int ih = ho * P.stride - P.dil + r * P.dil;
        if ((unsigned)ih >= (unsigned)P.H) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int iw = (int)wo * P.stride - P.dil + s * P.dil;
            if ((unsigned)iw >= (unsigned)P.W) continue;
            Chunk<T> xv, wv;
            xv.load(x + ((long long)ih * P.W + iw) * P.ldx);
            wv.load(w + (r * 3 + s) * P.C);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = fmaf(xv.get(e), wv.get(e), acc[e]);
        }
    }
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
    o.store(reinterpret_cast<T*>(P.y) + (((long long)n * P.Ho + ho) * P.Wo + wo) * P.ldy + c);
}

// Register-blocked variant: one thread produces TW = 4 consecutive output pixels of
// a row for one channel vector and loads every input column it needs ONCE
// (stride 1, dilation 1: 18 loads for 4 outputs instead of 36).  FLIP = 1 applies the
// taps reversed, which for stride 1 is exactly the data gradient (dy in, dx out).
template <typename T, int S, int D, int FLIP>
__global__ __launch_bounds__(256) void dw_fwd_tw_kernel(DwParams P) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int TW = 4;
    constexpr int NCOL = (TW - 1) * S + 2 * D + 1;
    const unsigned cv = P.C / VEC;
    int grow, xblk;
    dw_block_to_row(P.bx, grow, xblk);
    const unsigned idx = xblk * 256u + threadIdx.x;
    if (idx >= (unsigned)P.items) return;
    const unsigned wq = idx / cv;
    const int c = (int)(idx - wq * cv) * VEC;
    const int wo0 = (int)wq * TW;
    const int n = grow / P.Ho, ho = grow - n * P.Ho;
    const T* x = reinterpret_cast<const T*>(P.x) + (long long)n * P.H * P.W * P.ldx + c;
    const T* w = reinterpret_cast<const T*>(P.w) + c;
    float acc[TW][VEC];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
    const int iw0 = wo0 * S - D;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int ih = ho * S - D + r * D;
        if ((unsigned)ih >= (unsigned)P.H) continue;
        Chunk<T> wv[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) wv[s].load(w + ((FLIP ? (2 - r) : r) * 3 + (FLIP ? (2 - s) : s)) * P.C);
        const T* xr = x + (long long)ih * P.W * P.ldx;
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
            const int iw = iw0 + j;
            if ((unsigned)iw >= (unsigned)P.W) continue;
            Chunk<T> xv;
            xv.load(xr + (long long)iw * P.ldx);
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    if (j == t * S + s * D) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[t][e] = fmaf(xv.get(e), wv[s].get(e), acc[t][e]);
                    }
        }
    }
    T* y = reinterpret_cast<T*>(P.y) + (((long long)n * P.Ho + ho) * P.Wo + wo0) * P.ldy + c;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        if (wo0 + t >= P.Wo) break;
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, acc[t][e]);
        o.store(y + (long long)t * P.ldy);
    }
}

// dx[n,h,w,c] = sum_{r,s} dy[n,ho,wo,c] * w[r,s,c]  with ho*stride - dil + r*dil == h
template <typename T>
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(DwParams P) {
    constexpr int VEC = Elem<T>::VEC;
    const unsigned cv = P.C / VEC;
    int grow, xblk;
    dw_block_to_row(P.bx, grow, xblk);
    const unsigned idx = xblk * 256u + threadIdx.x;
    if (idx >= (unsigned)P.items) return;
    const unsigned iw = idx / cv;
    const int c = (int)(idx - iw * cv) * VEC;
    const int n = grow / P.H, ih = grow - n * P.H;
    const T* dy = reinterpret_cast<const T*>(P.x) + (long long)n * P.Ho * P.Wo * P.ldy + c;
    const T* w = reinterpret_cast<const T*>(P.w) + c;
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
            const int tw = (int)iw + P.dil - s * P.dil;
            if (tw < 0) continue;
            const int wo = tw / P.stride;
            if (wo * P.stride != tw || wo >= P.Wo) continue;
            Chunk<T> gv, wv;
            gv.load(dy + ((long long)ho * P.Wo + wo) * P.ldy);
            wv.load(w + (r * 3 + s) * P.C);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = fmaf(gv.get(e), wv.get(e), acc[e]);
        }
    }
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
    o.store(reinterpret_cast<T*>(P.y) + (((long long)n * P.H + ih) * P.W + iw) * P.ldx + c);
}

// dw[r,s,c] += sum_{n,ho,wo} dy[n,ho,wo,c] * x[n, ho*st-dil+r*dil, wo*st-dil+s*dil, c]
// block = TX channel vectors x TY pixel lanes; a block owns a run of output rows,
// keeps 9*VEC fp32 partials per thread, folds the TY lanes through LDS one tap at a
// time and issues ONE float atomic per (tap, channel).
struct DwWParams {
    const void* x;
    const void* dy;
    float* dw;
    int N, H, W, C, Ho, Wo, stride, dil, ldx, ldy;
    int rows_total;      // N*Ho output rows
    int rows_per_block;
    int tx, log_tx;
};

template <typename T, int FAST /* stride 1, dilation 1: sliding 3-column window */>
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(DwWParams P) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float red[256 * VEC];
    const int lx = threadIdx.x & (P.tx - 1);
    const int ly = threadIdx.x >> P.log_tx;
    const int ty = 256 >> P.log_tx;
    const int c = (blockIdx.x * P.tx + lx) * VEC;
    const bool c_ok = c < P.C;
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
    if (c_ok && FAST) {
        // each row lane walks a contiguous strip of the output row and keeps the three
        // input columns of its stencil in registers: 3 new loads per output instead of 9
        const int strip = (P.Wo + ty - 1) / ty;
        const int w_begin = ly * strip;
        int w_end = w_begin + strip;
        if (w_end > P.Wo) w_end = P.Wo;
        for (int row = row0; row < row1 && w_begin < w_end; ++row) {
            const int n = row / P.Ho, ho = row - n * P.Ho;
            const T* dyr = dy + (long long)row * P.Wo * P.ldy;
            const T* xn = x + (long long)n * P.H * P.W * P.ldx;
            Chunk<T> c0[3], c1[3], c2[3];
            bool rok[3];
            const T* xr[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int ih = ho - 1 + r;
                rok[r] = (unsigned)ih < (unsigned)P.H;
                xr[r] = xn + (long long)(rok[r] ? ih : 0) * P.W * P.ldx;
                if (rok[r] && w_begin - 1 >= 0) c0[r].load(xr[r] + (long long)(w_begin - 1) * P.ldx); else c0[r].zero();
                if (rok[r]) c1[r].load(xr[r] + (long long)w_begin * P.ldx); else c1[r].zero();
            }
            for (int wo = w_begin; wo < w_end; ++wo) {
                Chunk<T> gv;
                gv.load(dyr + (long long)wo * P.ldy);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    if (rok[r] && wo + 1 < P.W) c2[r].load(xr[r] + (long long)(wo + 1) * P.ldx); else c2[r].zero();
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float g = gv.get(e);
                        acc[r * 3 + 0][e] = fmaf(g, c0[r].get(e), acc[r * 3 + 0][e]);
                        acc[r * 3 + 1][e] = fmaf(g, c1[r].get(e), acc[r * 3 + 1][e]);
                        acc[r * 3 + 2][e] = fmaf(g, c2[r].get(e), acc[r * 3 + 2][e]);
                    }
                    c0[r] = c1[r];
                    c1[r] = c2[r];
                }
            }
        }
    } else if (c_ok) {
        for (int row = row0; row < row1; ++row) {
            const int n = row / P.Ho, ho = row - n * P.Ho;  // wave-uniform
            const T* dyr = dy + (long long)row * P.Wo * P.ldy;
            const T* xn = x + (long long)n * P.H * P.W * P.ldx;
            for (int wo = ly; wo < P.Wo; wo += ty) {
                Chunk<T> gv;
                gv.load(dyr + (long long)wo * P.ldy);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int ih = ho * P.stride - P.dil + r * P.dil;
                    if ((unsigned)ih >= (unsigned)P.H) continue;
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const int iw = wo * P.stride - P.dil + s * P.dil;
                        if ((unsigned)iw >= (unsigned)P.W) continue;
                        Chunk<T> xv;
                        xv.load(xn + ((long long)ih * P.W + iw) * P.ldx);
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            acc[r * 3 + s][e] = fmaf(gv.get(e), xv.get(e), acc[r * 3 + s][e]);
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
            if (ch < P.C) atomicAdd(P.dw + (long long)t * P.C + ch, a);
        }
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

}  // namespace

extern "C" int bg_dwconv3x3_fwd(const bg_dwconv_desc* d, const void* x, const void* w, void* y, void* stream) {
    int rc = check_dw(d, "bg_dwconv3x3_fwd");
    if (rc) return rc;
    BG_CHECK_ARG(x && w && y && aligned16(x) && aligned16(w) && aligned16(y), "bg_dwconv3x3_fwd: null/unaligned pointer");
    DwParams P{x, w, y, d->N, d->H, d->W, d->C, d->Ho, d->Wo, d->stride, d->dil, d->ldx, d->ldy, 0, 0};
    const long long rows = (long long)d->N * d->Ho;
    hipStream_t st = (hipStream_t)stream;
    const int cv = d->C / dtype_vec(d->dtype);
    const int sd = d->stride * 10 + d->dil;
    if (sd == 11 || sd == 12 || sd == 21) {
        P.items = ((d->Wo + 3) / 4) * cv;
        P.bx = (P.items + 255) / 256;
        BG_CHECK_ARG(rows * P.bx <= 0x7fffffffLL, "bg_dwconv3x3_fwd: grid too large");
        dim3 grid((unsigned)(rows * P.bx));
        if (sd == 11) BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw_fwd_tw_kernel<T, 1, 1, 0>), grid, dim3(256), 0, st, P));
        else if (sd == 12) BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw_fwd_tw_kernel<T, 1, 2, 0>), grid, dim3(256), 0, st, P));
        else BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw_fwd_tw_kernel<T, 2, 1, 0>), grid, dim3(256), 0, st, P));
    } else {
        P.items = d->Wo * cv;
        P.bx = (P.items + 255) / 256;
        BG_CHECK_ARG(rows * P.bx <= 0x7fffffffLL, "bg_dwconv3x3_fwd: grid too large");
        dim3 grid((unsigned)(rows * P.bx));
        BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw_fwd_kernel<T>), grid, dim3(256), 0, st, P));
    }
    BG_CHECK_LAUNCH("dw_fwd_kernel");
    return BG_OK;
}

extern "C" int bg_dwconv3x3_bwd_data(const bg_dwconv_desc* d, const void* dy, const void* w, void* dx, void* stream) {
    int rc = check_dw(d, "bg_dwconv3x3_bwd_data");
    if (rc) return rc;
    BG_CHECK_ARG(dy && w && dx && aligned16(dy) && aligned16(w) && aligned16(dx),
                 "bg_dwconv3x3_bwd_data: null/unaligned pointer");
    DwParams P{dy, w, dx, d->N, d->H, d->W, d->C, d->Ho, d->Wo, d->stride, d->dil, d->ldx, d->ldy, 0, 0};
    const long long rows = (long long)d->N * d->H;
    hipStream_t st = (hipStream_t)stream;
    const int cv = d->C / dtype_vec(d->dtype);
    if (d->stride == 1 && (d->dil == 1 || d->dil == 2)) {
        // stride 1: the data gradient is the same stencil with the taps reversed
        // (dy plays the input, dx the output; both are H x W)
        DwParams Q{dy, w, dx, d->N, d->H, d->W, d->C, d->H, d->W, 1, d->dil, d->ldy, d->ldx, 0, 0};
        Q.items = ((d->W + 3) / 4) * cv;
        Q.bx = (Q.items + 255) / 256;
        BG_CHECK_ARG(rows * Q.bx <= 0x7fffffffLL, "bg_dwconv3x3_bwd_data: grid too large");
        dim3 grid((unsigned)(rows * Q.bx));
        if (d->dil == 1) BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw_fwd_tw_kernel<T, 1, 1, 1>), grid, dim3(256), 0, st, Q));
        else BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw_fwd_tw_kernel<T, 1, 2, 1>), grid, dim3(256), 0, st, Q));
        BG_CHECK_LAUNCH("dw_fwd_tw_kernel(flip)");
        return BG_OK;
    }
    P.items = d->W * cv;
    P.bx = (P.items + 255) / 256;
    BG_CHECK_ARG(rows * P.bx <= 0x7fffffffLL, "bg_dwconv3x3_bwd_data: grid too large");
    dim3 grid((unsigned)(rows * P.bx));
    BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw_bwd_data_kernel<T>), grid, dim3(256), 0, st, P));
    BG_CHECK_LAUNCH("dw_bwd_data_kernel");
    return BG_OK;
}

extern "C" int bg_dwconv3x3_bwd_weight(const bg_dwconv_desc* d, const void* x, const void* dy, float* dw,
                                       void* stream) {
    int rc = check_dw(d, "bg_dwconv3x3_bwd_weight");
    if (rc) return rc;
    BG_CHECK_ARG(x && dy && dw && aligned16(x) && aligned16(dy), "bg_dwconv3x3_bwd_weight: null/unaligned pointer");
    DwWParams P{x, dy, dw, d->N, d->H, d->W, d->C, d->Ho, d->Wo, d->stride, d->dil, d->ldx, d->ldy, 0, 0, 0, 0};
    const int cv = d->C / dtype_vec(d->dtype);
    int best = 16, best_pad = 1 << 30;
    for (int tx = 16; tx <= 64; tx *= 2) {
        const int pad = (cv + tx - 1) / tx * tx;
        if (pad <= best_pad) { best_pad = pad; best = tx; }
    }
    P.tx = best;
    P.log_tx = best == 16 ? 4 : (best == 32 ? 5 : 6);
    const int gx = (cv + best - 1) / best;
    P.rows_total = d->N * d->Ho;
    // ~1536 blocks in total; each block at least one output row
    int gy = 1536 / gx;
    if (gy < 1) gy = 1;
    if (gy > P.rows_total) gy = P.rows_total;
    P.rows_per_block = (P.rows_total + gy - 1) / gy;
    gy = (P.rows_total + P.rows_per_block - 1) / P.rows_per_block;
    hipStream_t st = (hipStream_t)stream;
    if (d->stride == 1 && d->dil == 1)
        BG_DISPATCH_DTYPE(d->dtype, T,
                          hipLaunchKernelGGL((dw_bwd_weight_kernel<T, 1>), dim3(gx, (unsigned)gy), dim3(256), 0, st, P));
    else
        BG_DISPATCH_DTYPE(d->dtype, T,
                          hipLaunchKernelGGL((dw_bwd_weight_kernel<T, 0>), dim3(gx, (unsigned)gy), dim3(256), 0, st, P));
    BG_CHECK_LAUNCH("dw_bwd_weight_kernel");
    return BG_OK;
}
