// Backward of the fused [BatchNorm2d -> LeakyReLU -> depthwise 3x3] unit (deeplab.py:66-143; forward:
// bg_dwconv3x3_fwd_pre_stats) in ONE pass over its two inputs.
//
// Round 2 ran three kernels per unit: the depthwise data gradient (read dA, write d(act)), the depthwise weight gradient
// on the recomputed activation (read z, read dA) and the first pass of the BatchNorm backward (read d(act), read z):
// 6 tensor passes.  All three consume the same 3x3 window of dA around a pixel q and the raw convolution output z[q]:
//     d(act)[q] = sum_t dA[q + (1-t) D] w[t]
//     dW[t]    += act[q] dA[q + (1-t) D]            act = bf16(LeakyReLU(z*scale + shift))
//     s1       += g,  s2 += g * xhat                g = d(act)[q] * LeakyReLU'(u),  xhat = (z - mean) rstd
// so here a workgroup stages a (rows x columns x 64-channel) tile of dA with its halo in LDS by LDS-DMA (every load of
// the tile issued at once), a thread owns 4 channels of a pixel, reads the nine taps from LDS (the window never lives
// in registers: that is what made the register-window form of this fusion spill) and z[q] from global
// memory, writes d(act)[q] and keeps dW (36 fp32), s1, s2 (8 fp32) in registers over all the tiles it walks: 3 tensor
// passes (dA + halo from L2, z, d(act)).  HBM-bound: 6 bytes per element, ~40 VALU operations per element.
#include "common.h"
#include <algorithm>

namespace {

constexpr int FB_OOB = (int)0x80000000;
constexpr int FB_SLAB = 64;        // channels per slab: 128 bytes per pixel
constexpr int FB_ITEMS = 16;       // 4-channel items per slab = lanes along channels
constexpr int FB_THREADS = 512;
constexpr int FB_PL = FB_THREADS / FB_ITEMS;   // pixel lanes per workgroup: 32

struct DwFusedParams {
    const bf16_t* g;     // dA: gradient w.r.t. the depthwise output [N, H, W, C], pixel stride ldg
    const bf16_t* x;     // z: raw output of the previous pointwise convolution [N, H, W, C], pixel stride ldx
    const bf16_t* w;     // [3][3][C]
    const float* scale;  // [groups][C] forward affine: u = z * scale + shift
    const float* shift;
    const float* mean;   // [groups][C]
    const float* rstd;
    bf16_t* da;          // out: gradient w.r.t. the activated tensor, pixel stride ldda
    float* dw;           // [3][3][C] fp32, accumulated (NULL: weights frozen)
    double* s1;          // [groups][C] += sum g
    double* s2;          // [groups][C] += sum g * xhat
    int N, H, W, C, ldg, ldx, ldda, D, ipg;
    float slope;
    int slabs, R, TW, tiles_h, tiles_w, tiles, bps;
    int buf_bytes;   // one tile buffer (two of them, then the statistics scratch)
    int dpad;        // pixels of the dA part of a buffer (tile + halo, rounded up to 8); the z part (R x TW pixels) follows
    int zpad;        // pixels of the z part, rounded up to 8
};

__device__ __forceinline__ void fb_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_dst, 16, voffset, 0, 0, 0);
}

__device__ __forceinline__ void unpack4(const s16x4 v, float (&f)[4]) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    const u32x2 p = __builtin_bit_cast(u32x2, v);
    f[0] = __uint_as_float(p[0] << 16);
    f[1] = __uint_as_float(p[0] & 0xffff0000u);
    f[2] = __uint_as_float(p[1] << 16);
    f[3] = __uint_as_float(p[1] & 0xffff0000u);
}

template <bool WG>
__global__ __launch_bounds__(FB_THREADS) void dw_bwd_fused_kernel(DwFusedParams P) {
    extern __shared__ __align__(16) char fb_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blocks sharing id % 8 share an XCD: give every XCD a contiguous range of (tile, slab) so that the slabs of one
    // pixel range (whose 128-byte pieces share cache lines) and adjacent row tiles (which share halo rows) meet in one L2
    int lin = blockIdx.x;
    {
        const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = lin & 7, k = lin >> 3;
        lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int slab = lin % P.slabs, tb = lin / P.slabs;
    const int it = tid & (FB_ITEMS - 1), pl = tid >> 4;
    const int c0 = slab * FB_SLAB + it * 4;
    const bool c_ok = c0 < P.C;   // C is a multiple of 8: a 4-channel item is all inside or all outside
    const int D = P.D, TWp = P.TW + 2 * D, THp = P.R + 2 * D;
    const int tile_pix = THp * TWp;
    const int per_img = P.tiles_h * P.tiles_w;
    constexpr int NWAVE = FB_THREADS / 64;

    float wf[9][4], dwa[9][4], a1[4], a2[4];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        s16x4 v = {0, 0, 0, 0};
        if (c_ok) v = *reinterpret_cast<const s16x4*>(P.w + (long long)t * P.C + c0);
        unpack4(v, wf[t]);
#pragma unroll
        for (int e = 0; e < 4; ++e) dwa[t][e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) a1[e] = a2[e] = 0.f;
    float sc[4], sh[4], rs[4], mo[4];
    int cur_g = -1;
    char* const buf0 = fb_smem;
    float* const red_s = reinterpret_cast<float*>(fb_smem + 2 * P.buf_bytes);   // statistics scratch, apart from the tile buffers

    // block reduction of per-thread partial sums over the pixel lanes, one atomic per channel and value
    auto flush_stats = [&](int g) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red_s[(0 * FB_PL + pl) * FB_SLAB + it * 4 + e] = a1[e];
            red_s[(1 * FB_PL + pl) * FB_SLAB + it * 4 + e] = a2[e];
            a1[e] = a2[e] = 0.f;
        }
        __syncthreads();
        if (tid < 2 * FB_SLAB) {
            const int s = tid >> 6, ch = tid & 63;
            float acc = 0.f;
#pragma unroll
            for (int y = 0; y < FB_PL; ++y) acc += red_s[(s * FB_PL + y) * FB_SLAB + ch];
            const int c = slab * FB_SLAB + ch;
            if (c < P.C) atomicAdd((s ? P.s2 : P.s1) + (long long)g * P.C + c, (double)acc);
        }
    };

    // Staging of one dA tile with its halo: 8 pixels (x 128 B) per wave instruction, lanes beyond the image, the tile or
    // C fetch zeros (out-of-range offsets).  A tile's pieces are issued ONE OR TWO PER PIXEL ITERATION of the tile before
    // it (into the other buffer), so that the loads the compute loop waits for are never queued behind a burst.
    // Every call issues exactly ONE wave instruction (once the tile is complete: an out-of-range fetch into a per-wave dump
    // area), so the number of operations younger than any load is the same on every path and the compiler's counted
    // s_waitcnt vmcnt(N) for that load never includes a piece issued after it.  (With a conditional issue the counter
    // logic must assume the shortest path and the wave then waits for the fresh piece too: measured, every pixel
    // iteration paid a memory round trip.)
    __amdgpu_buffer_rsrc_t d_rg, d_rx;
    const int stage_pix = P.dpad + P.zpad;   // piece index space of one tile: dA with its halo, then z
    int d_base = stage_pix, d_row0 = 0, d_col0 = 0;
    char* d_dst = buf0;
    char* const dump = fb_smem + 2 * P.buf_bytes + 2 * FB_PL * FB_SLAB * 4 + wave * 1024;
    const int piece = lane & 7;
    const bool ch_ok = slab * FB_SLAB + piece * 8 < P.C;
    const float inv_twp = 1.f / (float)TWp, inv_tw = 1.f / (float)P.TW;
    const unsigned g_bytes = (unsigned)(((long long)P.H * P.W - 1) * P.ldg + P.C) * 2u;
    const unsigned x_bytes = (unsigned)(((long long)P.H * P.W - 1) * P.ldx + P.C) * 2u;
    // (the descriptors must always be valid: the dump fetches use them too)
    d_rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(P.g), 0, (int)g_bytes, 0x00020000);
    d_rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(P.x), 0, (int)x_bytes, 0x00020000);
    auto dma_begin = [&](int tile_, char* dst) {
        const int n = tile_ / per_img, rem = tile_ - n * per_img;
        const int th = rem / P.tiles_w, tw = rem - th * P.tiles_w;
        d_rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(P.g + (long long)n * P.H * P.W * P.ldg), 0, (int)g_bytes, 0x00020000);
        d_rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(P.x + (long long)n * P.H * P.W * P.ldx), 0, (int)x_bytes, 0x00020000);
        d_row0 = th * P.R; d_col0 = tw * P.TW;
        d_base = wave * 8;
        d_dst = dst;
    };
    auto dma_piece = [&]() {
        const bool live = d_base < stage_pix;   // wave-uniform
        const bool isz = d_base >= P.dpad;      // wave-uniform: dpad is a multiple of 8
        int voff = FB_OOB;
        if (!isz) {
            const int pi = d_base + (lane >> 3);
            const int tr = (int)(((float)pi + 0.5f) * inv_twp), tc = pi - tr * TWp;   // exact: pi < 2^12
            const int ih = d_row0 - D + tr, iw = d_col0 - D + tc;
            const bool ok = live && ch_ok && tr < THp && (unsigned)ih < (unsigned)P.H && (unsigned)iw < (unsigned)P.W;
            if (ok) voff = ((ih * P.W + iw) * P.ldg + slab * FB_SLAB + piece * 8) * 2;
            fb_dma16(d_rg, live ? d_dst + d_base * 128 : dump, voff);
        } else {
            const int pi = d_base - P.dpad + (lane >> 3);
            const int tr = (int)(((float)pi + 0.5f) * inv_tw), tc = pi - tr * P.TW;
            const int ih = d_row0 + tr, iw = d_col0 + tc;
            const bool ok = live && ch_ok && tr < P.R && ih < P.H && iw < P.W;
            if (ok) voff = ((ih * P.W + iw) * P.ldx + slab * FB_SLAB + piece * 8) * 2;
            fb_dma16(d_rx, live ? d_dst + d_base * 128 : dump, voff);
        }
        d_base += 8 * NWAVE;
    };
    const int pieces_per_wave = (stage_pix + 8 * NWAVE - 1) / (8 * NWAVE);

    int tile = tb, cur = 0;
    if (tile < P.tiles) {
        dma_begin(tile, buf0);
        for (int k = 0; k < pieces_per_wave; ++k) dma_piece();
    }
    for (; tile < P.tiles; tile += P.bps) {
        const int n = tile / per_img, rem = tile - n * per_img;
        const int th = rem / P.tiles_w, tw = rem - th * P.tiles_w;
        const int g = n / P.ipg;
        if (g != cur_g) {   // block-uniform
            if (cur_g >= 0) flush_stats(cur_g);
            cur_g = g;
            if (c_ok) {
                const long long o = (long long)g * P.C + c0;
                const f32x4 a = *reinterpret_cast<const f32x4*>(P.scale + o), b = *reinterpret_cast<const f32x4*>(P.shift + o);
                const f32x4 m = *reinterpret_cast<const f32x4*>(P.mean + o), r = *reinterpret_cast<const f32x4*>(P.rstd + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[e] = a[e]; sh[e] = b[e]; rs[e] = r[e]; mo[e] = -m[e] * r[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[e] = 0.f; sh[e] = 0.f; rs[e] = 0.f; mo[e] = 0.f; }
            }
        }
        const int row0 = th * P.R, col0 = tw * P.TW;
        const int rows_out = min(P.R, P.H - row0), cols_out = min(P.TW, P.W - col0);
        const long long img_pix = (long long)n * P.H * P.W;
        // this tile has landed (this wave's pieces: the counter; every wave's: the barrier), and every wave has left the
        // previous tile, whose buffer the next tile's pieces may now overwrite
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* tbuf = buf0 + cur * P.buf_bytes;
        const bool has_next = tile + P.bps < P.tiles;
        if (has_next) dma_begin(tile + P.bps, buf0 + (cur ^ 1) * P.buf_bytes);
        else d_base = stage_pix;   // nothing to stage: the pieces go to the dump
        {
            const int npix = rows_out * cols_out;
            const int iters = (npix + FB_PL - 1) / FB_PL;   // the same for every thread: one load, one piece, one store per iteration
            const float inv_c = 1.f / (float)cols_out;
            const unsigned dbytes = (unsigned)(((long long)P.H * P.W - 1) * P.ldda + P.C) * 2u;
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(P.da + img_pix * P.ldda, 0, (int)dbytes, 0x00020000);
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
            const char* zbuf = tbuf + P.dpad * 128 + it * 8;
            int p = pl;
            int r = (int)(((float)p + 0.5f) * inv_c), c = p - r * cols_out;
            for (int i = 0; i < iters; ++i) {
                const bool ok = p < npix;
                const int pn = p + FB_PL;
                const int rn = (int)(((float)pn + 0.5f) * inv_c), cn = pn - rn * cols_out;
                dma_piece();   // two pieces of the next tile per iteration, always issued (static operation counts)
                dma_piece();
                if (!ok) { r = 0; c = 0; }   // stay inside the tile buffer; the contributions are masked below
                float z[4], u[4], av[4];
                unpack4(*reinterpret_cast<const s16x4*>(zbuf + (r * P.TW + c) * 128), z);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u[e] = fmaf(z[e], sc[e], sh[e]);
                    av[e] = fmaxf(u[e], P.slope * u[e]);
                }
                if (WG) {   // the activation as the forward pass stored it (bf16); nothing from a lane without a pixel
                    const bf16x4 ab = {(bf16_t)av[0], (bf16_t)av[1], (bf16_t)av[2], (bf16_t)av[3]};
                    unpack4(__builtin_bit_cast(s16x4, ab), av);
                    if (!ok) av[0] = av[1] = av[2] = av[3] = 0.f;
                }
                float dacc[4] = {0.f, 0.f, 0.f, 0.f};
                const char* wbase = tbuf + ((r + 2 * D) * TWp + c + 2 * D) * 128 + it * 8;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int tr = t / 3, tc = t - tr * 3;
                    float v[4];
                    unpack4(*reinterpret_cast<const s16x4*>(wbase - (tr * TWp + tc) * D * 128), v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dacc[e] = fmaf(v[e], wf[t][e], dacc[e]);
                        if (WG) dwa[t][e] = fmaf(av[e], v[e], dwa[t][e]);
                    }
                }
                const bf16x4 db = {(bf16_t)dacc[0], (bf16_t)dacc[1], (bf16_t)dacc[2], (bf16_t)dacc[3]};
                const int doff = (c_ok && ok) ? (((row0 + r) * P.W + col0 + c) * P.ldda + c0) * 2 : FB_OOB;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, db), rd, doff, 0, 0);
                float dr[4];
                unpack4(__builtin_bit_cast(s16x4, db), dr);   // the gradient as stored: what the second pass will read
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gg = ok ? dr[e] * (u[e] > 0.f ? 1.f : P.slope) : 0.f;
                    a1[e] += gg;
                    a2[e] = fmaf(gg, fmaf(z[e], rs[e], mo[e]), a2[e]);
                }
                p = pn; r = rn; c = cn;
            }
            for (int k = 2 * iters; k < pieces_per_wave; ++k) dma_piece();   // ragged tiles: whatever is left of the next tile
        }
        cur ^= 1;
    }
    if (cur_g >= 0) flush_stats(cur_g);
    if (WG) {
        float* red = reinterpret_cast<float*>(fb_smem);   // the tile buffers are free now
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(t * FB_PL + pl) * FB_SLAB + it * 4 + e] = dwa[t][e];
        __syncthreads();
        for (int j = tid; j < 9 * FB_SLAB; j += FB_THREADS) {
            const int t = j >> 6, ch = j & 63;
            float acc = 0.f;
#pragma unroll
            for (int y = 0; y < FB_PL; ++y) acc += red[(t * FB_PL + y) * FB_SLAB + ch];
            const int c = slab * FB_SLAB + ch;
            if (c < P.C) atomicAdd(P.dw + (long long)t * P.C + c, acc);
        }
    }
}

}  // namespace

extern "C" int bg_dwconv3x3_bwd_fused(const bg_dwconv_desc* d, const void* dy, const void* w, const void* x, const float* scale,
                                      const float* shift, const float* mean, const float* rstd, int32_t groups, int32_t act,
                                      void* da, int32_t ldda, float* dw, double* s1, double* s2, void* stream) {
    BG_CHECK_ARG(d && d->dtype == BG_BF16, "bg_dwconv3x3_bwd_fused: bf16 tensors only (the fp32 path runs the three separate kernels)");
    BG_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->C % 8 == 0 && d->Ho == d->H && d->Wo == d->W && d->stride == 1 &&
                     (d->dil == 1 || d->dil == 2), "bg_dwconv3x3_bwd_fused: stride 1, dilation 1 or 2, C a multiple of 8");
    BG_CHECK_ARG(d->ldx >= d->C && d->ldy >= d->C && ldda >= d->C && d->ldx % 8 == 0 && d->ldy % 8 == 0 && ldda % 8 == 0,
                 "bg_dwconv3x3_bwd_fused: bad pixel strides");
    BG_CHECK_ARG(dy && w && x && scale && shift && mean && rstd && da && s1 && s2 && aligned16(dy) && aligned16(w) && aligned16(x) &&
                     aligned16(da) && aligned16(scale) && aligned16(shift) && aligned16(mean) && aligned16(rstd),
                 "bg_dwconv3x3_bwd_fused: null/unaligned pointer");
    BG_CHECK_ARG(groups >= 1 && d->N % groups == 0 && act >= 0 && act <= 2, "bg_dwconv3x3_bwd_fused: groups / act");
    BG_CHECK_ARG((long long)d->H * d->W * std::max(std::max(d->ldx, d->ldy), ldda) * 2 < (1LL << 31), "bg_dwconv3x3_bwd_fused: an image beyond 2 GiB");
    DwFusedParams P{};
    P.g = (const bf16_t*)dy; P.x = (const bf16_t*)x; P.w = (const bf16_t*)w;
    P.scale = scale; P.shift = shift; P.mean = mean; P.rstd = rstd;
    P.da = (bf16_t*)da; P.dw = dw; P.s1 = s1; P.s2 = s2;
    P.N = d->N; P.H = d->H; P.W = d->W; P.C = d->C; P.ldg = d->ldy; P.ldx = d->ldx; P.ldda = ldda; P.D = d->dil;
    P.ipg = d->N / groups;
    P.slope = act == 0 ? 1.f : act == 2 ? 0.f : LRELU_SLOPE;
    P.slabs = (d->C + FB_SLAB - 1) / FB_SLAB;
    // tile shape: at most max_pix staged pixels (128 B each); the (rows, columns) with the least halo per output pixel
    // tile shape: dA tile with halo + z tile (128 B per pixel) in one buffer of at most max_pix pixels, two buffers;
    // the (rows, columns) with the least halo per output pixel
    static const int max_pix = getenv("BGAMD_FB_PIX") ? atoi(getenv("BGAMD_FB_PIX")) : 512;
    const int D = P.D;
    double best = 1e30;
    for (int tw = std::min(8, P.W); tw <= std::min(P.W, 96); ++tw) {
        int r = std::min(P.H, (max_pix - 2 * D * (tw + 2 * D)) / (2 * tw + 2 * D));   // (r + 2D)(tw + 2D) + r tw <= max_pix
        while (r >= 1 && ((r + 2 * D) * (tw + 2 * D) + 7) / 8 * 8 + (r * tw + 7) / 8 * 8 > max_pix) --r;
        if (r < 1) continue;
        const int th_ = (P.H + r - 1) / r, tw_ = (P.W + tw - 1) / tw;
        const double staged = (double)th_ * tw_ * (r + 2 * D) * (tw + 2 * D);   // dA pixels moved into LDS per image
        if (staged < best) { best = staged; P.R = r; P.TW = tw; P.tiles_h = th_; P.tiles_w = tw_; }
    }
    BG_CHECK_ARG(best < 1e30, "bg_dwconv3x3_bwd_fused: no tile shape");
    P.tiles = P.N * P.tiles_h * P.tiles_w;
    static const int target = getenv("BGAMD_FB_BLOCKS") ? atoi(getenv("BGAMD_FB_BLOCKS")) : 256;   // one workgroup per CU walking its tiles
    P.bps = std::max(1, std::min(P.tiles, target / P.slabs));
    P.dpad = ((P.R + 2 * D) * (P.TW + 2 * D) + 7) / 8 * 8;
    P.zpad = (P.R * P.TW + 7) / 8 * 8;
    P.buf_bytes = (P.dpad + P.zpad) * 128;
    const size_t lds = std::max<size_t>((size_t)2 * P.buf_bytes + (size_t)2 * FB_PL * FB_SLAB * 4 + (FB_THREADS / 64) * 1024, (size_t)9 * FB_PL * FB_SLAB * 4);
    BG_CHECK_ARG(lds <= 160 * 1024, "bg_dwconv3x3_bwd_fused: tile too large for the LDS");
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_bwd_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_bwd_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    const dim3 grid((unsigned)(P.slabs * P.bps));
    if (dw) hipLaunchKernelGGL(dw_bwd_fused_kernel<true>, grid, dim3(FB_THREADS), lds, (hipStream_t)stream, P);
    else hipLaunchKernelGGL(dw_bwd_fused_kernel<false>, grid, dim3(FB_THREADS), lds, (hipStream_t)stream, P);
    BG_CHECK_LAUNCH("dw_bwd_fused_kernel");
    return BG_OK;
}
