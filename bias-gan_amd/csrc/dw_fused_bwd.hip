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
constexpr int FB_NWAVE = FB_THREADS / 64;
constexpr int FB_PL = FB_THREADS / FB_ITEMS;   // pixel lanes per workgroup: 32 = 4 rows x 8 columns per iteration
constexpr int FB_TW = 24;          // output columns per tile (3 column groups of 8)
constexpr int FB_PITCH = 32;       // pixels per staged dA row: FB_TW + 2 D, padded to whole 8-pixel pieces

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
    int N, H, W, C, ldg, ldx, ldda, ipg;
    float slope;
    int slabs, tiles_h, tiles_w, tiles, bps;
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

// Geometry (compile time per dilation D): a tile is R output rows x 24 columns x 64 channels; its dA part is staged with
// the halo as (R + 2D) rows of 32 pixels (24 + 2D, padded to whole 8-pixel pieces), its z part as R rows of 24 pixels;
// two such buffers.  The 32 pixel lanes of the workgroup cover 4 rows x 8 columns per iteration, R/4 x 3 iterations per
// tile, fully unrolled: every LDS address is a per-thread base plus an immediate, every global address a per-thread
// offset into a descriptor the scalar unit rebases -- no integer multiply and no division on the vector unit inside the
// loop (the first version spent more than half of its issue slots there: SQ_ACTIVE_INST_ANY 0.49 per wave at two waves
// per SIMD, i.e. issue-bound at 2.4 TB/s).
template <int D> struct FbGeo {
    static constexpr int R = D == 1 ? 8 : 4;
    static constexpr int THP = R + 2 * D;
    static constexpr int DPIX = THP * FB_PITCH;         // staged dA pixels
    static constexpr int ZPIX = R * FB_TW;              // staged z pixels
    static constexpr int NPD = DPIX / 8, NPZ = ZPIX / 8;   // 8-pixel pieces
    static constexpr int PPW = (NPD + NPZ + FB_NWAVE - 1) / FB_NWAVE;   // pieces per wave and tile
    static constexpr int ITERS = (R / 4) * 3;
    static constexpr int PPI = (PPW + ITERS - 1) / ITERS;               // pieces issued per pixel iteration
    static constexpr int BUF_BYTES = (DPIX + ZPIX) * 128;
    static constexpr int RED_BYTES = 2 * FB_PL * FB_SLAB * 4;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES + RED_BYTES + FB_NWAVE * 1024;
    static_assert(FB_TW + 2 * D <= FB_PITCH && R % 4 == 0, "tile geometry");
    static_assert(LDS_BYTES <= 160 * 1024 && 9 * FB_PL * FB_SLAB * 4 <= 2 * BUF_BYTES, "LDS budget");
};

template <bool WG, int D>
__global__ __launch_bounds__(FB_THREADS) void dw_bwd_fused_kernel(DwFusedParams P) {
    typedef FbGeo<D> G;
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    extern __shared__ __align__(16) char fb_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blocks sharing id % 8 share an XCD: give every XCD a contiguous range of (tile, slab) so that the slabs of one
    // pixel range (whose 128-byte pieces share cache lines) and adjacent tiles (which share halo pixels) meet in one L2
    int lin = blockIdx.x;
    {
        const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = lin & 7, k = lin >> 3;
        lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int slab = lin % P.slabs, tb = lin / P.slabs;
    const int it = tid & (FB_ITEMS - 1), pl = tid >> 4;
    const int pr = pl >> 3, pc = pl & 7;                 // this thread's pixel inside an iteration's 4 x 8 group
    const int c0 = slab * FB_SLAB + it * 4;
    const bool c_ok = c0 < P.C;   // C is a multiple of 8: a 4-channel item is all inside or all outside
    const int per_img = P.tiles_h * P.tiles_w;

    float wf[9][4], dwa[9][4], a1[4], a2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) a1[e] = a2[e] = 0.f;
    float sc[4], sh[4], rs[4], mo[4];
    int cur_g = -1;
    char* const buf0 = fb_smem;
    float* const red_s = reinterpret_cast<float*>(fb_smem + 2 * G::BUF_BYTES);   // statistics scratch, apart from the tile buffers
    char* const dump = fb_smem + 2 * G::BUF_BYTES + G::RED_BYTES + wave * 1024;

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

    // ---- staging: piece k of a tile = 8 pixels x 128 B = one wave instruction; pieces 0 .. NPD-1 are the dA tile with
    // its halo (row k / 4, pixels (k % 4) * 8 ..), NPD .. NPD+NPZ-1 the z tile (row kz / 3, pixels (kz % 3) * 8 ..).
    // Wave w issues pieces w, w + 8, ...: row and column of a piece are wave-uniform (scalar unit), the descriptor is
    // rebased to the piece's first pixel, a lane adds its own constant offset and is switched off (out-of-range offset:
    // the DMA writes zeros) outside the image or beyond C.  Every call issues exactly ONE wave instruction -- once the
    // tile is complete an out-of-range fetch into a per-wave dump area -- so the number of operations younger than any
    // other is the same on every path (the compiler's counted waits never include a freshly issued piece).
    const int lp = lane >> 3, piece = lane & 7;
    const bool ch_ok = slab * FB_SLAB + piece * 8 < P.C;
    const int lane_g = ch_ok ? (lp * P.ldg + slab * FB_SLAB + piece * 8) * 2 : FB_OOB;   // byte offset of this lane inside a dA piece
    const int lane_x = ch_ok ? (lp * P.ldx + slab * FB_SLAB + piece * 8) * 2 : FB_OOB;
    const char* d_bg = reinterpret_cast<const char*>(P.g);   // first staged pixel of the tile being staged: (row0 - D, col0 - D) of dA
    const char* d_bx = reinterpret_cast<const char*>(P.x);   // ... (row0, col0) of z
    int d_row0 = 0, d_col0 = 0;
    bool d_on = false;
    char* d_dst = buf0;
    auto dma_begin = [&](int tile_, char* dst) {
        const int n = tile_ / per_img, rem = tile_ - n * per_img;
        const int th = rem / P.tiles_w, tw = rem - th * P.tiles_w;
        d_row0 = th * G::R; d_col0 = tw * FB_TW;
        d_bg = reinterpret_cast<const char*>(P.g) + (((long long)n * P.H + d_row0 - D) * P.W + d_col0 - D) * P.ldg * 2;
        d_bx = reinterpret_cast<const char*>(P.x) + (((long long)n * P.H + d_row0) * P.W + d_col0) * P.ldx * 2;
        d_dst = dst;
        d_on = true;
    };
    // Piece k of a part (dA: NPD pieces, row k / 4, pixels (k % 4) * 8 ..; z: NPZ pieces, row k / 3, pixels (k % 3) * 8 ..)
    // is issued by wave k % 8 as its slot k / 8; slots are compile-time constants at every call site, the rest is a
    // handful of scalar operations (32-bit: a tile spans < 2^31 bytes).
    int wv = wave;   // re-materialised per tile (below): the slot arithmetic stays inside the tile loop instead of 50 hoisted scalars
    auto dma_a = [&](int j) {
        const int k = wv + FB_NWAVE * j;       // wave-uniform
        const int tr = k >> 2, px = (k & 3) * 8;
        const bool live = d_on && k < G::NPD;
        const char* base = d_bg + (tr * P.W + px) * P.ldg * 2;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0x7fffffff, 0x00020000);
        const bool ok = live && (unsigned)(d_row0 - D + tr) < (unsigned)P.H && (unsigned)(d_col0 - D + px + lp) < (unsigned)P.W;
        fb_dma16(rs, live ? d_dst + k * 1024 : dump, ok ? lane_g : FB_OOB);
    };
    auto dma_z = [&](int j) {
        const int k = wv + FB_NWAVE * j;
        const int tr = (k * 43) >> 7, px = (k - tr * 3) * 8;          // k / 3 (k < 128)
        const bool live = d_on && k < G::NPZ;
        const char* base = d_bx + (tr * P.W + px) * P.ldx * 2;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0x7fffffff, 0x00020000);
        const bool ok = live && d_row0 + tr < P.H && d_col0 + px + lp < P.W;
        fb_dma16(rs, live ? d_dst + (G::NPD + k) * 1024 : dump, ok ? lane_x : FB_OOB);
    };
    constexpr int SLOTS_A = (G::NPD + FB_NWAVE - 1) / FB_NWAVE, SLOTS_Z = (G::NPZ + FB_NWAVE - 1) / FB_NWAVE;
    constexpr int APER = (SLOTS_A + G::ITERS - 1) / G::ITERS, ZPER = (SLOTS_Z + G::ITERS - 1) / G::ITERS;   // per pixel iteration

    int tile = tb, cur = 0;
    if (tile < P.tiles) {
        dma_begin(tile, buf0);
#pragma unroll
        for (int j = 0; j < SLOTS_A; ++j) dma_a(j);
#pragma unroll
        for (int j = 0; j < SLOTS_Z; ++j) dma_z(j);
    }
    // the filter taps AFTER the first tile's pieces are on their way (their round trip would otherwise precede the DMA's)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        s16x4 v = {0, 0, 0, 0};
        if (c_ok) v = *reinterpret_cast<const s16x4*>(P.w + (long long)t * P.C + c0);
        unpack4(v, wf[t]);
#pragma unroll
        for (int e = 0; e < 4; ++e) dwa[t][e] = 0.f;
    }
    // per-thread LDS bases inside a buffer (immediates do the rest)
    const int lds_tap = ((pr + 2 * D) * FB_PITCH + pc + 2 * D) * 128 + it * 8;
    const int lds_z = G::DPIX * 128 + (pr * FB_TW + pc) * 128 + it * 8;
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
        const int row0 = th * G::R, col0 = tw * FB_TW;
        // this tile has landed (this wave's pieces: the counter; every wave's: the barrier), and every wave has left the
        // previous tile, whose buffer the next tile's pieces may now overwrite
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        asm volatile("" : "+s"(wv));
        const char* tbuf = buf0 + cur * G::BUF_BYTES;
        const bool has_next = tile + P.bps < P.tiles;
        if (has_next) dma_begin(tile + P.bps, buf0 + (cur ^ 1) * G::BUF_BYTES);
        else d_on = false;   // nothing to stage: the pieces go to the dump
        // the output descriptor rebased to the tile's first pixel; a thread's offset inside it is fixed per iteration
        const long long tile_pix0 = ((long long)n * P.H + row0) * P.W + col0;
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(P.da + tile_pix0 * P.ldda, 0, 0x7fffffff, 0x00020000);
        const char* tap0 = tbuf + lds_tap;
        const char* z0 = tbuf + lds_z;
        const int rmax = c_ok ? P.H - row0 - pr : 0, cmax = P.W - col0 - pc;   // this thread's pixel exists while rg*4 < rmax, cg*8 < cmax
        const int d_thread = ((pr * P.W + pc) * P.ldda + c0) * 2;
        int nrg = G::R / 4;
        if (WG) asm volatile("" : "+s"(nrg));   // the weight-gradient variant keeps its row groups a loop: unrolled, its 36 more
                                                // accumulators push the allocation over 256 registers (measured: 125 spills)
#pragma unroll
        for (int rg = 0; rg < nrg; ++rg) {
#pragma unroll
            for (int cg = 0; cg < 3; ++cg) {
                __builtin_amdgcn_sched_barrier(0);   // one iteration's loads are not hoisted into the previous one (registers)
                const bool ok = rg * 4 < rmax && cg * 8 < cmax;
#pragma unroll
                for (int q = 0; q < APER; ++q) dma_a((rg * 3 + cg) * APER + q);   // slots beyond the part: dump fetches
#pragma unroll
                for (int q = 0; q < ZPER; ++q) dma_z((rg * 3 + cg) * ZPER + q);
                float z[4], u[4], av[4];
                unpack4(*reinterpret_cast<const s16x4*>(z0 + rg * (4 * FB_TW * 128) + cg * 8 * 128), z);
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
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int tr = t / 3, tc = t - tr * 3;
                    float v[4];
                    unpack4(*reinterpret_cast<const s16x4*>(tap0 + rg * (4 * FB_PITCH * 128) + ((-tr * D) * FB_PITCH + cg * 8 - tc * D) * 128), v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dacc[e] = fmaf(v[e], wf[t][e], dacc[e]);
                        if (WG) dwa[t][e] = fmaf(av[e], v[e], dwa[t][e]);
                    }
                }
                const bf16x4 db = {(bf16_t)dacc[0], (bf16_t)dacc[1], (bf16_t)dacc[2], (bf16_t)dacc[3]};
                const int doff = ok ? d_thread + (rg * 4 * P.W + cg * 8) * P.ldda * 2 : FB_OOB;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, db), rd, doff, 0, 0);
                float dr[4];
                unpack4(__builtin_bit_cast(s16x4, db), dr);   // the gradient as stored: what the second pass will read
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gg = ok ? dr[e] * (u[e] > 0.f ? 1.f : P.slope) : 0.f;
                    a1[e] += gg;
                    a2[e] = fmaf(gg, fmaf(z[e], rs[e], mo[e]), a2[e]);
                }
            }
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

template <bool WG, int D>
void fb_launch(const DwFusedParams& P, hipStream_t st) {
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_bwd_fused_kernel<WG, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    hipLaunchKernelGGL((dw_bwd_fused_kernel<WG, D>), dim3((unsigned)(P.slabs * P.bps)), dim3(FB_THREADS), FbGeo<D>::LDS_BYTES, st, P);
}


// ---------------------------------------------------------------------------------------------------------------------
// The fork at a Block's input (deeplab.py:134-141), backward, in ONE pass (round 4).  The Block's input a0 = y_k -- the
// activated output of the previous Block's final [BatchNorm + residual + LeakyReLU] node -- feeds the first depthwise
// convolution and the skip path.  Round 3 ran, per Block: bg_dwconv3x3_bwd_data_add (read dA, read the skip gradient, write
// d(a0)), bg_dwconv3x3_bwd_weight (read dA, read a0) and, for the PRODUCER of a0, the first pass of its BatchNorm
// backward (read d(a0), read y_k, read z) plus a second output of its apply pass (write d(a0) * act'(y_k) as the residual
// gradient): 8 + 2 tensor passes.  Here, per pixel q and channel:
//     t        = bf16( sum_t dA[q + (1-t)] w[t] + skip[q] )                what bg_dwconv3x3_bwd_data_add stores
//     gout[q]  = t * act'(a0[q])                                           the gradient BEHIND the producer's activation
//     dW[t]   += a0[q] dA[q + (1-t)]
//     s1      += gout,  s2 += gout * xhat,   xhat = (z[q] - mean) rstd     the producer's BatchNorm-backward sums
// 5 tensor passes (dA + halo from L2, a0, skip, z; gout).  The producer's apply pass then reads gout as an already
// activated gradient (act = 0), and its residual gradient IS gout (no copy).  Stride 1, dilation 1 (the entry, middle and
// exit Blocks at os = 16), bf16.  Geometry: tiles of 4 rows x 24 columns x 64 channels, all four inputs staged by LDS-DMA
// into one of two buffers (dA with its halo as 6 rows x 32 pixels, the three others as 4 x 24), 3 pixel iterations per tile.
struct DwForkParams {
    const bf16_t* g;      // dA: gradient w.r.t. the depthwise output [N, H, W, C], pixel stride ldg
    const bf16_t* x;      // a0: the Block's activated input, pixel stride ldx
    const bf16_t* skip;   // gradient w.r.t. the skip alias of a0, pixel stride lds
    const bf16_t* z;      // the producer BatchNorm's input, pixel stride ldz
    const bf16_t* w;      // [3][3][C]
    const float* mean;    // producer BatchNorm, [groups][C]
    const float* rstd;
    bf16_t* gout;         // out, pixel stride ldo
    float* dw;            // [3][3][C] fp32, accumulated (NULL: weights frozen)
    double* s1;
    double* s2;
    int N, H, W, C, ldg, ldx, lds, ldz, ldo, ipg;
    float slope;          // act'(y) for y < 0 (1: the producer has no activation)
    int slabs, tiles_h, tiles_w, tiles, bps;
};

struct FkGeo {
    static constexpr int R = 4, D = 1;
    static constexpr int THP = R + 2 * D;
    static constexpr int DPIX = THP * FB_PITCH;      // 192 staged dA pixels
    static constexpr int ZPIX = R * FB_TW;           // 96 staged pixels of each of a0 / skip / z
    static constexpr int NPD = DPIX / 8, NPZ = ZPIX / 8;             // 24 and 12 eight-pixel pieces
    static constexpr int SLOTS_A = NPD / FB_NWAVE;                    // 3: wave w issues dA pieces w, w + 8, w + 16
    static constexpr int SLOTS_Z = (NPZ + FB_NWAVE - 1) / FB_NWAVE;   // 2 per part: pieces w and (w < 4) w + 8
    static constexpr int SLOTS = SLOTS_A + 3 * SLOTS_Z;               // 9 per wave and tile
    static constexpr int ITERS = 3;
    static constexpr int PPI = SLOTS / ITERS;                         // 3 issued per pixel iteration
    static constexpr int BUF_BYTES = (DPIX + 3 * ZPIX) * 128;         // 61 440
    static constexpr int RED_BYTES = 2 * FB_PL * FB_SLAB * 4;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES + RED_BYTES + FB_NWAVE * 1024;
    static_assert(NPD % FB_NWAVE == 0 && SLOTS % ITERS == 0, "slot arithmetic");
    static_assert(LDS_BYTES <= 160 * 1024 && 9 * FB_PL * FB_SLAB * 4 <= 2 * BUF_BYTES, "LDS budget");
};

template <bool WG>
__global__ __launch_bounds__(FB_THREADS) void dw_fork_bwd_kernel(DwForkParams P) {
    typedef FkGeo G;
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    extern __shared__ __align__(16) char fb_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lin = blockIdx.x;
    {   // blocks sharing id % 8 share an XCD: a contiguous range of (tile, slab) per XCD (halo pixels and the 128-byte pieces
        // of one pixel range meet in one L2)
        const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = lin & 7, k = lin >> 3;
        lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int slab = lin % P.slabs, tb = lin / P.slabs;
    const int it = tid & (FB_ITEMS - 1), pl = tid >> 4;
    const int pr = pl >> 3, pc = pl & 7;
    const int c0 = slab * FB_SLAB + it * 4;
    const bool c_ok = c0 < P.C;
    const int per_img = P.tiles_h * P.tiles_w;

    float wf[9][4], dwa[9][4], a1[4], a2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) a1[e] = a2[e] = 0.f;
    float rs[4], mo[4];
    int cur_g = -1;
    char* const buf0 = fb_smem;
    float* const red_s = reinterpret_cast<float*>(fb_smem + 2 * G::BUF_BYTES);
    char* const dump = fb_smem + 2 * G::BUF_BYTES + G::RED_BYTES + wave * 1024;

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

    // ---- staging (the scheme of dw_bwd_fused_kernel): piece = 8 pixels x 128 B = one wave instruction; every call issues
    // exactly one -- a dead slot fetches out of range into the wave's dump area -- so the operation count is static.
    const int lp = lane >> 3, piece = lane & 7;
    const bool ch_ok = slab * FB_SLAB + piece * 8 < P.C;
    const int cb = slab * FB_SLAB + piece * 8;
    const int lane_g = ch_ok ? (lp * P.ldg + cb) * 2 : FB_OOB;
    const int lane_x = ch_ok ? (lp * P.ldx + cb) * 2 : FB_OOB;
    const int lane_s = ch_ok ? (lp * P.lds + cb) * 2 : FB_OOB;
    const int lane_z = ch_ok ? (lp * P.ldz + cb) * 2 : FB_OOB;
    const char* d_bg = reinterpret_cast<const char*>(P.g);
    const char* d_bx = reinterpret_cast<const char*>(P.x);
    const char* d_bs = reinterpret_cast<const char*>(P.skip);
    const char* d_bz = reinterpret_cast<const char*>(P.z);
    int d_row0 = 0, d_col0 = 0;
    bool d_on = false;
    char* d_dst = buf0;
    auto dma_begin = [&](int tile_, char* dst) {
        const int n = tile_ / per_img, rem = tile_ - n * per_img;
        const int th = rem / P.tiles_w, tw = rem - th * P.tiles_w;
        d_row0 = th * G::R; d_col0 = tw * FB_TW;
        const long long pix = ((long long)n * P.H + d_row0) * P.W + d_col0;
        d_bg = reinterpret_cast<const char*>(P.g) + (pix - P.W - 1) * P.ldg * 2;     // (row0 - 1, col0 - 1)
        d_bx = reinterpret_cast<const char*>(P.x) + pix * P.ldx * 2;
        d_bs = reinterpret_cast<const char*>(P.skip) + pix * P.lds * 2;
        d_bz = reinterpret_cast<const char*>(P.z) + pix * P.ldz * 2;
        d_dst = dst;
        d_on = true;
    };
    int wv = wave;
    auto dma_a = [&](int j) {        // dA piece wv + 8 j: row k / 4, pixels (k % 4) * 8 ..
        const int k = wv + FB_NWAVE * j;
        const int tr = k >> 2, px = (k & 3) * 8;
        const char* base = d_bg + (tr * P.W + px) * P.ldg * 2;
        const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0x7fffffff, 0x00020000);
        const bool ok = d_on && (unsigned)(d_row0 - 1 + tr) < (unsigned)P.H && (unsigned)(d_col0 - 1 + px + lp) < (unsigned)P.W;
        fb_dma16(rsd, d_on ? d_dst + k * 1024 : dump, ok ? lane_g : FB_OOB);
    };
    auto dma_p = [&](int part, int j) {   // piece wv + 8 j of part 0 (a0) / 1 (skip) / 2 (z): row k / 3, pixels (k % 3) * 8 ..
        const int k = wv + FB_NWAVE * j;
        const int tr = (k * 43) >> 7, px = (k - tr * 3) * 8;
        const bool live = d_on && k < G::NPZ;
        const char* b = part == 0 ? d_bx : part == 1 ? d_bs : d_bz;
        const int ld = part == 0 ? P.ldx : part == 1 ? P.lds : P.ldz;
        const int lo = part == 0 ? lane_x : part == 1 ? lane_s : lane_z;
        const char* base = b + (tr * P.W + px) * ld * 2;
        const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0x7fffffff, 0x00020000);
        const bool ok = live && d_row0 + tr < P.H && d_col0 + px + lp < P.W;
        fb_dma16(rsd, live ? d_dst + (G::NPD + part * G::NPZ + k) * 1024 : dump, ok ? lo : FB_OOB);
    };
    auto dma_slot = [&](int s_) {    // slot 0 .. 8 of a tile (compile-time at every call site)
        if (s_ < G::SLOTS_A) dma_a(s_);
        else dma_p((s_ - G::SLOTS_A) / G::SLOTS_Z, (s_ - G::SLOTS_A) % G::SLOTS_Z);
    };

    int tile = tb, cur = 0;
    if (tile < P.tiles) {
        dma_begin(tile, buf0);
#pragma unroll
        for (int s_ = 0; s_ < G::SLOTS; ++s_) dma_slot(s_);
    }
    // the filter taps AFTER the first tile's pieces are on their way (their round trip would otherwise precede the DMA's)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        s16x4 v = {0, 0, 0, 0};
        if (c_ok) v = *reinterpret_cast<const s16x4*>(P.w + (long long)t * P.C + c0);
        unpack4(v, wf[t]);
#pragma unroll
        for (int e = 0; e < 4; ++e) dwa[t][e] = 0.f;
    }
    const int lds_tap = ((pr + 2) * FB_PITCH + pc + 2) * 128 + it * 8;
    const int lds_p = G::DPIX * 128 + (pr * FB_TW + pc) * 128 + it * 8;
    for (; tile < P.tiles; tile += P.bps) {
        const int n = tile / per_img, rem = tile - n * per_img;
        const int th = rem / P.tiles_w, tw = rem - th * P.tiles_w;
        const int g = n / P.ipg;
        if (g != cur_g) {   // block-uniform
            if (cur_g >= 0) flush_stats(cur_g);
            cur_g = g;
            if (c_ok) {
                const long long o = (long long)g * P.C + c0;
                const f32x4 m = *reinterpret_cast<const f32x4*>(P.mean + o), r = *reinterpret_cast<const f32x4*>(P.rstd + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) { rs[e] = r[e]; mo[e] = -m[e] * r[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { rs[e] = 0.f; mo[e] = 0.f; }
            }
        }
        const int row0 = th * G::R, col0 = tw * FB_TW;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        asm volatile("" : "+s"(wv));
        const char* tbuf = buf0 + cur * G::BUF_BYTES;
        const bool has_next = tile + P.bps < P.tiles;
        if (has_next) dma_begin(tile + P.bps, buf0 + (cur ^ 1) * G::BUF_BYTES);
        else d_on = false;
        const long long tile_pix0 = ((long long)n * P.H + row0) * P.W + col0;
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(P.gout + tile_pix0 * P.ldo, 0, 0x7fffffff, 0x00020000);
        const char* tap0 = tbuf + lds_tap;
        const char* p0 = tbuf + lds_p;
        const bool r_ok = c_ok && pr < P.H - row0;
        const int cmax = P.W - col0 - pc;
        const int d_thread = ((pr * P.W + pc) * P.ldo + c0) * 2;
#pragma unroll
        for (int cg = 0; cg < 3; ++cg) {
            __builtin_amdgcn_sched_barrier(0);
            const bool ok = r_ok && cg * 8 < cmax;
#pragma unroll
            for (int q = 0; q < G::PPI; ++q) dma_slot(cg * G::PPI + q);
            float av[4], sk[4], zv[4];
            unpack4(*reinterpret_cast<const s16x4*>(p0 + cg * 8 * 128), av);                       // a0 (zero outside the image)
            unpack4(*reinterpret_cast<const s16x4*>(p0 + (G::ZPIX + cg * 8) * 128), sk);
            unpack4(*reinterpret_cast<const s16x4*>(p0 + (2 * G::ZPIX + cg * 8) * 128), zv);
            float dacc[4] = {sk[0], sk[1], sk[2], sk[3]};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int tr = t / 3, tc = t - tr * 3;
                float v[4];
                unpack4(*reinterpret_cast<const s16x4*>(tap0 + ((-tr) * FB_PITCH + cg * 8 - tc) * 128), v);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dacc[e] = fmaf(v[e], wf[t][e], dacc[e]);
                    if (WG) dwa[t][e] = fmaf(av[e], v[e], dwa[t][e]);
                }
            }
            // t as the separate data-gradient kernel stores it, then the producer's activation derivative from its output
            const bf16x4 tb_ = {(bf16_t)dacc[0], (bf16_t)dacc[1], (bf16_t)dacc[2], (bf16_t)dacc[3]};
            float tr_[4], gg[4];
            unpack4(__builtin_bit_cast(s16x4, tb_), tr_);
            const float m1 = ok ? 1.f : 0.f, ms = m1 * P.slope;
#pragma unroll
            for (int e = 0; e < 4; ++e) gg[e] = tr_[e] * (av[e] > 0.f ? m1 : ms);      // y > 0 ? 1 : slope, as the reduce / apply kernels
            const bf16x4 gb = {(bf16_t)gg[0], (bf16_t)gg[1], (bf16_t)gg[2], (bf16_t)gg[3]};
            const int doff = ok ? d_thread + cg * 8 * P.ldo * 2 : FB_OOB;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, gb), rd, doff, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a1[e] += gg[e];
                a2[e] = fmaf(gg[e], fmaf(zv[e], rs[e], mo[e]), a2[e]);
            }
        }
        cur ^= 1;
    }
    if (cur_g >= 0) flush_stats(cur_g);
    if (WG) {
        float* red = reinterpret_cast<float*>(fb_smem);
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

template <bool WG>
void fk_launch(const DwForkParams& P, hipStream_t st) {
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_fork_bwd_kernel<WG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    hipLaunchKernelGGL((dw_fork_bwd_kernel<WG>), dim3((unsigned)(P.slabs * P.bps)), dim3(FB_THREADS), FkGeo::LDS_BYTES, st, P);
}

}  // namespace

extern "C" int bg_dwconv3x3_bwd_fork(const bg_dwconv_desc* d, const void* dy, const void* w, const void* a0, const void* skip,
                                     int32_t ldskip, const void* z, int32_t ldz, const float* mean, const float* rstd,
                                     int32_t groups, int32_t act, void* gout, int32_t ldgout, float* dw, double* s1, double* s2,
                                     void* stream) {
    BG_CHECK_ARG(d && d->dtype == BG_BF16, "bg_dwconv3x3_bwd_fork: bf16 tensors only (the fp32 path runs the separate kernels)");
    BG_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->C % 8 == 0 && d->Ho == d->H && d->Wo == d->W && d->stride == 1 &&
                     d->dil == 1, "bg_dwconv3x3_bwd_fork: stride 1, dilation 1, C a multiple of 8");
    BG_CHECK_ARG(d->ldx >= d->C && d->ldy >= d->C && ldskip >= d->C && ldz >= d->C && ldgout >= d->C && d->ldx % 8 == 0 &&
                     d->ldy % 8 == 0 && ldskip % 8 == 0 && ldz % 8 == 0 && ldgout % 8 == 0, "bg_dwconv3x3_bwd_fork: bad pixel strides");
    BG_CHECK_ARG(dy && w && a0 && skip && z && mean && rstd && gout && s1 && s2 && aligned16(dy) && aligned16(w) && aligned16(a0) &&
                     aligned16(skip) && aligned16(z) && aligned16(gout) && aligned16(mean) && aligned16(rstd),
                 "bg_dwconv3x3_bwd_fork: null/unaligned pointer");
    BG_CHECK_ARG(groups >= 1 && d->N % groups == 0 && act >= 0 && act <= 2, "bg_dwconv3x3_bwd_fork: groups / act");
    const int ldmax = std::max(std::max(std::max(d->ldx, d->ldy), std::max(ldskip, ldz)), ldgout);
    BG_CHECK_ARG((long long)d->H * d->W * ldmax * 2 < (1LL << 31), "bg_dwconv3x3_bwd_fork: an image beyond 2 GiB");
    DwForkParams P{};
    P.g = (const bf16_t*)dy; P.x = (const bf16_t*)a0; P.skip = (const bf16_t*)skip; P.z = (const bf16_t*)z; P.w = (const bf16_t*)w;
    P.mean = mean; P.rstd = rstd; P.gout = (bf16_t*)gout; P.dw = dw; P.s1 = s1; P.s2 = s2;
    P.N = d->N; P.H = d->H; P.W = d->W; P.C = d->C; P.ldg = d->ldy; P.ldx = d->ldx; P.lds = ldskip; P.ldz = ldz; P.ldo = ldgout;
    P.ipg = d->N / groups;
    P.slope = act == 0 ? 1.f : act == 2 ? 0.f : LRELU_SLOPE;
    P.slabs = (d->C + FB_SLAB - 1) / FB_SLAB;
    P.tiles_h = (P.H + FkGeo::R - 1) / FkGeo::R;
    P.tiles_w = (P.W + FB_TW - 1) / FB_TW;
    P.tiles = P.N * P.tiles_h * P.tiles_w;
    static const int target = getenv("BGAMD_FB_BLOCKS") ? atoi(getenv("BGAMD_FB_BLOCKS")) : 256;   // one workgroup per CU walking its tiles
    P.bps = std::max(1, std::min(P.tiles, target / P.slabs));
    hipStream_t st = (hipStream_t)stream;
    if (dw) fk_launch<true>(P, st); else fk_launch<false>(P, st);
    BG_CHECK_LAUNCH("dw_fork_bwd_kernel");
    return BG_OK;
}

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
    P.N = d->N; P.H = d->H; P.W = d->W; P.C = d->C; P.ldg = d->ldy; P.ldx = d->ldx; P.ldda = ldda;
    P.ipg = d->N / groups;
    P.slope = act == 0 ? 1.f : act == 2 ? 0.f : LRELU_SLOPE;
    P.slabs = (d->C + FB_SLAB - 1) / FB_SLAB;
    // tile shape: at most max_pix staged pixels (128 B each); the (rows, columns) with the least halo per output pixel
    const int R = d->dil == 1 ? FbGeo<1>::R : FbGeo<2>::R;
    P.tiles_h = (P.H + R - 1) / R;
    P.tiles_w = (P.W + FB_TW - 1) / FB_TW;
    P.tiles = P.N * P.tiles_h * P.tiles_w;
    static const int target = getenv("BGAMD_FB_BLOCKS") ? atoi(getenv("BGAMD_FB_BLOCKS")) : 256;   // one workgroup per CU walking its tiles
    P.bps = std::max(1, std::min(P.tiles, target / P.slabs));
    hipStream_t st = (hipStream_t)stream;
    if (d->dil == 1) { if (dw) fb_launch<true, 1>(P, st); else fb_launch<false, 1>(P, st); }
    else { if (dw) fb_launch<true, 2>(P, st); else fb_launch<false, 2>(P, st); }
    BG_CHECK_LAUNCH("dw_bwd_fused_kernel");
    return BG_OK;
}
