// Dense convolution as implicit GEMM on the CDNA4 matrix cores.
//
//   forward   y[p, co]  = sum_{tap, ci} x[gather(p, tap), ci] * w[co, tap, ci]
//   bwd_data  dx[p, ci] = sum_{tap, co} dy[scatter^-1(p, tap), co] * wt[ci, tap, co]
//   bwd_wgt   dw[co, tap, ci] += sum_p dy[p, co] * x[gather(p, tap), ci]
//
// forward and bwd_data are ONE kernel (gemm_conv_kernel): a 128(out-channel) x
// 128(pixel) tile per 256-thread workgroup, 4 waves as 2x2, each wave a 64x64
// sub-tile of 4x4 MFMA 16x16 accumulators.  The out-channel side is the MFMA
// A operand and the pixel side the B operand, so each lane ends up with 4
// CONSECUTIVE output channels of one pixel -> 8/16-byte NHWC stores with no LDS
// transpose.  Both operands are K-contiguous in memory (NHWC activations, KRSC /
// CRSK weights), staged global -> registers -> LDS (the gather needs per-lane
// zero fill, so no LDS-DMA), double-buffered, one barrier per K-step, with an
// XOR swizzle on the 16-byte chunk index that makes the ds_read_b128 fragment
// reads bank-conflict free for the lane groups of MI355X_MICROARCH.md (LDS).
//
// bwd_wgt (wgrad_kernel) reduces over pixels, which is the SLOW dimension of
// both operands; tiles are staged pixel-major and the MFMA fragments are read
// with the gfx950 transposing LDS read ds_read_b64_tr_b16 (bf16) or plain
// ds_read_b32 columns (f32).  Pixels are split over blockIdx.y; partial tiles
// are accumulated into the fp32 gradient with float atomics shaped as two
// 128-byte row segments per wave instruction (32x32 accumulator layout).
//
// dtype BG_F32 runs the same kernels on the f32-input MFMA (16x16x4 / 32x32x2):
// exact fp32 products and accumulation, used as the parity path.
#include "igemm_fat.h"

namespace {


// One K-slab of MFMAs for a wave: acc[i][j] += A_i * B_j over the BKB bytes of K.
template <typename T, int BKB>
__device__ __forceinline__ void mma_slab(const char* sA, const char* sB, int wave_c, int wave_p, int lane,
                                         f32x4 (&acc)[4][4]) {
    const int r16 = lane & 15, q = lane >> 4;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int ks = 0; ks < BKB / 64; ++ks) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const bf16x8*>(sA + lds_off<BKB>(wave_c * 64 + i * 16 + r16, ks * 4 + q));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(sB + lds_off<BKB>(wave_p * 64 + j * 16 + r16, ks * 4 + q));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < BKB / 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const float*>(sA + lds_off<BKB>(wave_c * 64 + i * 16 + r16, kk) + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b[j] = *reinterpret_cast<const float*>(sB + lds_off<BKB>(wave_p * 64 + j * 16 + r16, kk) + q * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
}


// Epilogue shared by both staging variants.  A lane holds, per accumulator, 4
// consecutive output channels of one pixel -> 8/16-byte NHWC stores.  With
// P.stat_sum set it also reduces sum(y) and sum(y^2) of the values AS STORED
// (after rounding to T) over the tile's 128 pixels: 16-lane shuffle tree, the two
// pixel-waves are combined through LDS, one fp64 atomic per channel per tile.
template <typename T, int TCH = TILE, int TP = TILE>
__device__ __forceinline__ void conv_epilogue(const GemmConvParams& P, f32x4 (&acc)[4][4], long long p_base, int c_base,
                                              int wave_c, int wave_p, int lane, char* smem) {
    T* out = reinterpret_cast<T*>(P.out);
    const int r16 = lane & 15, q = lane >> 4;
    float s1[4][4], s2[4][4];
    if (P.stat_sum) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) s1[i][e] = s2[i][e] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long p = p_base + wave_p * 64 + j * 16 + r16;
        const bool p_ok = p < P.M;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = c_base + wave_c * 64 + i * 16 + q * 4;
            if (co >= P.NO) continue;
            f32x4 v = acc[i][j];
            if (P.bias) {
                v[0] += P.bias[co + 0];
                v[1] += P.bias[co + 1];
                v[2] += P.bias[co + 2];
                v[3] += P.bias[co + 3];
            }
            T o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f(v[e]);
            if (P.stat_sum && p_ok) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = Elem<T>::to_f(o[e]);
                    s1[i][e] += r;
                    s2[i][e] = fmaf(r, r, s2[i][e]);
                }
            }
            if (!p_ok) continue;
            T* dst = out + p * P.ldo + co;
            if constexpr (sizeof(T) == 2) {
                bf16x4 ov = {o[0], o[1], o[2], o[3]};
                *reinterpret_cast<bf16x4*>(dst) = ov;
            } else {
                f32x4 ov = {o[0], o[1], o[2], o[3]};
                *reinterpret_cast<f32x4*>(dst) = ov;
            }
        }
    }
    if (P.stat_sum) {  // wave-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[i][e] = row16_sum(s1[i][e]);
                s2[i][e] = row16_sum(s2[i][e]);
            }
        __syncthreads();  // every wave is done reading the staging ring: reuse it
        constexpr int PW = TP / 64;                   // pixel-waves of the tile
        float* red = reinterpret_cast<float*>(smem);  // [wave_p][TCH channels][2]
        if (r16 == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int cl = wave_c * 64 + i * 16 + q * 4 + e;
                    red[(wave_p * TCH + cl) * 2 + 0] = s1[i][e];
                    red[(wave_p * TCH + cl) * 2 + 1] = s2[i][e];
                }
        }
        __syncthreads();
        const int t = threadIdx.x;
        if (t < TCH && c_base + t < P.NO) {
            // statistic groups are contiguous pixel ranges (sub-batches normalised separately); a tile
            // never straddles two (checked on the host)
            const int grp = P.stat_group_pix ? (int)(p_base / P.stat_group_pix) : 0;
            const long long o = (long long)grp * P.NO + c_base + t;
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int wp = 0; wp < PW; ++wp) {
                a1 += red[(wp * TCH + t) * 2];
                a2 += red[(wp * TCH + t) * 2 + 1];
            }
            atomicAdd(P.stat_sum + o, (double)a1);
            atomicAdd(P.stat_sq + o, (double)a2);
        }
    }
}


template <typename T, int BKB>
__global__ __launch_bounds__(NTHREADS) void gemm_conv_kernel(GemmConvParams P) {
    constexpr int ES = (int)sizeof(T);
    constexpr int BK = BKB / ES;         // K elements per step
    constexpr int CPR = BKB / 16;        // 16-byte chunks per tile row
    constexpr int RPP = NTHREADS / CPR;  // rows per load pass
    constexpr int NPASS = TILE / RPP;
    constexpr int TILE_BYTES = TILE * BKB;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_c = wave >> 1, wave_p = wave & 1;

    // XCD-aware tile order: workgroups that share an XCD (same id % 8) walk
    // consecutive tiles, and the channel tile varies fastest, so the blocks that
    // re-read one pixel tile sit on one L2.
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int tile_c = bid % P.tiles_c;
    const int tile_p = bid / P.tiles_c;
    const long long p_base = (long long)tile_p * TILE;
    const int c_base = tile_c * TILE;

    const int chunk = tid % CPR;
    const int row0 = tid / CPR;
    const int RS = P.KH * P.KW;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.in), 0, P.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.w), 0, P.w_bytes, 0x00020000);

    // per-thread row bookkeeping, fixed for the whole K loop
    int pix_base[NPASS];  // byte offset of the tap-(0,0) source pixel (+ this thread's chunk); only meaningful when valid
    int pix_n[NPASS], pix_h[NPASS], pix_w[NPASS];
    bool pix_ok[NPASS];
    int w_base[NPASS];    // byte offset of (co, tap 0, c = chunk) or OOB
    const bool direct = !P.transposed || P.stride == 1;  // source pixel = base + uniform tap delta
    const int sgn = P.transposed ? -1 : 1;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const long long p = p_base + row0 + i * RPP;
        pix_ok[i] = p < P.M;
        const unsigned pp = pix_ok[i] ? (unsigned)p : 0u;  // M < 2^31 (checked on the host)
        const unsigned ohw = (unsigned)(P.OH * P.OW);
        const unsigned n = pp / ohw;
        const unsigned rem = pp - n * ohw;
        const unsigned q = rem / (unsigned)P.OW;
        const int oh = (int)q, ow = (int)(rem - q * (unsigned)P.OW);
        pix_n[i] = (int)n;
        if (!P.transposed) {
            pix_h[i] = oh * P.stride - P.pad;
            pix_w[i] = ow * P.stride - P.pad;
        } else {
            pix_h[i] = oh + P.pad;
            pix_w[i] = ow + P.pad;
        }
        pix_base[i] = (((int)n * P.IH + pix_h[i]) * P.IW + pix_w[i]) * P.ldi * ES + chunk * 16;
        const int co = c_base + row0 + i * RPP;
        w_base[i] = co < P.NO ? co * RS * P.CKp * ES + chunk * 16 : OOB;
    }

    Chunk<T> ra[NPASS], rb[NPASS];
    const int ksteps_per_tap = (P.CK + BK - 1) / BK;
    const int KT = RS * ksteps_per_tap;
    // The last K-step of a tap may reach past the CK real channels of a pixel.  The packed weights
    // are zero there, but the activation lanes are whatever sits behind the row (the next pixel,
    // or the never-written pad lanes of a row whose pixel stride is rounded up): 0 * NaN must not
    // happen, so those chunks are fetched as out-of-range zeros.
    const bool tail_cut = (ksteps_per_tap - 1) * BK + chunk * (16 / ES) >= P.CK;

    // load cursor: runs one K-step ahead of the MFMAs
    int l_tap_r = 0, l_tap_s = 0, l_ks = 0, l_tap = 0;
    int va[NPASS], vb[NPASS];  // current byte offsets (advance by BKB per K-step inside a tap)
    auto start_tap = [&]() {
        const int dh = sgn * l_tap_r * P.dil, dw_ = sgn * l_tap_s * P.dil;
        const int tap_delta = (dh * P.IW + dw_) * P.ldi * ES;  // wave-uniform
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            va[i] = w_base[i] == OOB ? OOB : w_base[i] + l_tap * P.CKp * ES;
            if (direct) {
                const int ih = pix_h[i] + dh, iw = pix_w[i] + dw_;
                const bool ok = pix_ok[i] && (unsigned)ih < (unsigned)P.IH && (unsigned)iw < (unsigned)P.IW;
                vb[i] = ok ? pix_base[i] + tap_delta : OOB;
            } else {  // strided data gradient: only source pixels on the stride lattice contribute
                const int th = pix_h[i] - l_tap_r * P.dil, tw = pix_w[i] - l_tap_s * P.dil;
                bool ok = pix_ok[i] && th >= 0 && tw >= 0;
                const int ih = th / P.stride, iw = tw / P.stride;
                ok = ok && (ih * P.stride == th) && (iw * P.stride == tw) && ih < P.IH && iw < P.IW;
                vb[i] = ok ? ((pix_n[i] * P.IH + ih) * P.IW + iw) * P.ldi * ES + chunk * 16 : OOB;
            }
        }
    };
    auto issue_loads = [&]() {
        if (l_ks == 0) start_tap();
        const bool cut = tail_cut && l_ks == ksteps_per_tap - 1;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            ra[i].v = __builtin_bit_cast(typename Elem<T>::vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, va[i], 0, 0));
            rb[i].v = __builtin_bit_cast(typename Elem<T>::vec_t,
                                         __builtin_amdgcn_raw_buffer_load_b128(rs_in, cut ? OOB : vb[i], 0, 0));
            va[i] += BKB;  // OOB + small stays out of range
            vb[i] += BKB;
        }
        if (++l_ks == ksteps_per_tap) {
            l_ks = 0;
            ++l_tap;
            if (++l_tap_s == P.KW) { l_tap_s = 0; ++l_tap_r; }
        }
    };
    int lds_w[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) lds_w[i] = lds_off<BKB>(row0 + i * RPP, chunk);
    auto write_lds = [&](int buf) {
        char* sA = smem + buf * 2 * TILE_BYTES;
        char* sB = sA + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            ra[i].store(reinterpret_cast<T*>(sA + lds_w[i]));
            rb[i].store(reinterpret_cast<T*>(sB + lds_w[i]));
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue_loads();
    write_lds(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) issue_loads();  // global loads in flight under the MFMAs
        const char* sA = smem + cur * 2 * TILE_BYTES;
        mma_slab<T, BKB>(sA, sA + TILE_BYTES, wave_c, wave_p, lane, acc);
        if (more) write_lds(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    conv_epilogue<T>(P, acc, p_base, c_base, wave_c, wave_p, lane, smem);
}


// TCH = out-channel rows of the tile: 128 (4 waves) or 256 (8 waves, each still a 64x64 sub-tile).
// The kernel is bound by operand delivery into LDS, not by the MFMAs (an ablation build without the
// MFMAs runs as fast, one without the loads 1.5-1.8x faster); 256 x 128 moves 25 % fewer operand bytes
// per FLOP than 128 x 128 and puts 16 waves on a CU (2 workgroups x 72 KiB of LDS).
template <typename T, int BKB, int NBUF, int TCH, int TP = TILE, bool SPLIT = false>
__global__ __launch_bounds__(TCH * TP / 64) void gemm_conv_dma_kernel(GemmConvParams P) {
    constexpr int ES = (int)sizeof(T);
    constexpr int BK = BKB / ES;
    constexpr int CPR = BKB / 16;   // chunks per row
    constexpr int RPG = 64 / CPR;   // tile rows covered by one wave-wide DMA (1 KiB)
    constexpr int PW = TP / 64;                 // pixel-waves; every wave owns a 64 x 64 sub-tile
    constexpr int NW = (TCH / 64) * PW;         // waves per workgroup
    constexpr int ROWS_A = TCH / NW, ROWS_B = TP / NW;  // tile rows a wave stages per operand
    constexpr int NGA = ROWS_A / RPG, NGB = ROWS_B / RPG;  // DMAs per wave per K-step
    constexpr int NG = NGA > NGB ? NGA : NGB;
    constexpr int GROUP = NGA + NGB;  // VMEM ops per wave per K-step
    constexpr int TILE_BYTES = TCH * BKB;               // A (weight) rows of one stage
    constexpr int STAGE_BYTES = (TCH + TP) * BKB;
    constexpr int DIST = NBUF - 1;  // K-steps in flight ahead of the MFMAs
    static_assert(NGA >= 1 && NGB >= 1, "a wave stages at least one DMA per operand");
    static_assert(DIST == 2 || DIST == 3, "counted waits are written for 2 or 3 K-steps of prefetch");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_c = wave / PW, wave_p = wave % PW;

    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    int split = 0;
    if (SPLIT) {   // split-K: the grid is (tiles x splits); a block owns K-steps [split * kt_per_split, ...)
        const int ntile = P.tiles_c * P.tiles_p;
        split = bid / ntile;
        bid -= split * ntile;
    }
    const int tile_c = bid % P.tiles_c, tile_p = bid / P.tiles_c;
    const long long p_base = (long long)tile_p * TP;
    const int c_base = tile_c * TCH;
    const int RS = P.KH * P.KW;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.in), 0, P.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.w), 0, P.w_bytes, 0x00020000);

    const int lr = lane / CPR, lc = lane % CPR;
    int pix_base[NGB], pix_n[NGB], pix_h[NGB], pix_w[NGB], w_base[NGA], chk16[NGB];
    bool pix_ok[NGB], tail_cut[NGB];  // tail_cut: see gemm_conv_kernel
    const bool direct = !P.transposed || P.stride == 1;
    const int sgn = P.transposed ? -1 : 1;
    auto src_chunk = [&](int row) { return ((BKB == 64) ? (lc ^ ((0 - (row >> 2)) & 3)) : (lc ^ ((row >> 1) & 7))) * 16; };
#pragma unroll
    for (int g = 0; g < NGA; ++g) {
        const int row = wave * ROWS_A + g * RPG + lr;  // weight row inside the tile
        const int co = c_base + row;
        w_base[g] = co < P.NO ? co * RS * P.CKp * ES + src_chunk(row) : OOB;
    }
#pragma unroll
    for (int g = 0; g < NGB; ++g) {
        const int row = wave * ROWS_B + g * RPG + lr;  // pixel row inside the tile
        chk16[g] = src_chunk(row);
        const long long p = p_base + row;
        pix_ok[g] = p < P.M;
        const unsigned pp = pix_ok[g] ? (unsigned)p : 0u;
        const unsigned ohw = (unsigned)(P.OH * P.OW);
        const unsigned n = pp / ohw;
        const unsigned rem = pp - n * ohw;
        const unsigned q = rem / (unsigned)P.OW;
        const int oh = (int)q, ow = (int)(rem - q * (unsigned)P.OW);
        pix_n[g] = (int)n;
        if (!P.transposed) {
            pix_h[g] = oh * P.stride - P.pad;
            pix_w[g] = ow * P.stride - P.pad;
        } else {
            pix_h[g] = oh + P.pad;
            pix_w[g] = ow + P.pad;
        }
        pix_base[g] = (((int)n * P.IH + pix_h[g]) * P.IW + pix_w[g]) * P.ldi * ES + chk16[g];
    }

    const int ksteps_per_tap = (P.CK + BK - 1) / BK;
#ifdef ABL_KT1   // ablation: one K-step only -> what a tile costs besides its K loop
    const int KT = 1;
#else
    int KT = RS * ksteps_per_tap;
#endif
#pragma unroll
    for (int g = 0; g < NGB; ++g) tail_cut[g] = (ksteps_per_tap - 1) * BK + chk16[g] / ES >= P.CK;

    int l_tap_r = 0, l_tap_s = 0, l_ks = 0, l_tap = 0;
    bool fresh = SPLIT;   // split-K: the first issue starts in the middle of a tap
    if (SPLIT) {
#ifndef ABL_KT1
        const int kt0 = split * P.kt_per_split;
        KT = min(P.kt_per_split, KT - kt0);   // >= 1 (launcher)
        l_tap = kt0 / ksteps_per_tap;
        l_ks = kt0 - l_tap * ksteps_per_tap;
        l_tap_r = l_tap / P.KW;
        l_tap_s = l_tap - l_tap_r * P.KW;
#endif
    }
    int va[NGA], vb[NGB];
    auto start_tap = [&]() {
        const int dh = sgn * l_tap_r * P.dil, dw_ = sgn * l_tap_s * P.dil;
        const int tap_delta = (dh * P.IW + dw_) * P.ldi * ES;
#pragma unroll
        for (int g = 0; g < NGA; ++g) va[g] = w_base[g] == OOB ? OOB : w_base[g] + l_tap * P.CKp * ES;
#pragma unroll
        for (int g = 0; g < NGB; ++g) {
            if (direct) {
                const int ih = pix_h[g] + dh, iw = pix_w[g] + dw_;
                const bool ok = pix_ok[g] && (unsigned)ih < (unsigned)P.IH && (unsigned)iw < (unsigned)P.IW;
                vb[g] = ok ? pix_base[g] + tap_delta : OOB;
            } else {
                const int th = pix_h[g] - l_tap_r * P.dil, tw = pix_w[g] - l_tap_s * P.dil;
                bool ok = pix_ok[g] && th >= 0 && tw >= 0;
                const int ih = th / P.stride, iw = tw / P.stride;
                ok = ok && (ih * P.stride == th) && (iw * P.stride == tw) && ih < P.IH && iw < P.IW;
                vb[g] = ok ? ((pix_n[g] * P.IH + ih) * P.IW + iw) * P.ldi * ES + chk16[g] : OOB;
            }
        }
    };
    auto issue = [&](int buf) {
        if (l_ks == 0) start_tap();
        else if (SPLIT && fresh) {
            start_tap();
#pragma unroll
            for (int g = 0; g < NGA; ++g) va[g] += l_ks * BKB;   // an out-of-range marker stays out of range
#pragma unroll
            for (int g = 0; g < NGB; ++g) vb[g] += l_ks * BKB;
        }
        fresh = false;
        char* stage_a = smem + buf * STAGE_BYTES + wave * ROWS_A * BKB;
        char* stage_b = smem + buf * STAGE_BYTES + TILE_BYTES + wave * ROWS_B * BKB;
        const bool last = l_ks == ksteps_per_tap - 1;
#pragma unroll
        for (int g = 0; g < NG; ++g) {  // A and B DMAs interleaved
            if (g < NGA) {
#ifdef ABL_NO_LOAD   // ablation: without the global->LDS half (every DMA out of range: no memory traffic)
                dma16(rs_w, stage_a + g * RPG * BKB, OOB);
#elif defined(ABL_HOT_LOAD)   // ablation: every DMA hits the same few KB (L1/L2-resident operands)
                dma16(rs_w, stage_a + g * RPG * BKB, va[g] == OOB ? OOB : (va[g] & 0xFFFF));
#else
                dma16(rs_w, stage_a + g * RPG * BKB, va[g]);
#endif
                va[g] += BKB;
            }
            if (g < NGB) {
#ifdef ABL_NO_LOAD
                dma16(rs_in, stage_b + g * RPG * BKB, OOB);
#elif defined(ABL_HOT_LOAD)
                dma16(rs_in, stage_b + g * RPG * BKB, vb[g] == OOB ? OOB : (vb[g] & 0xFFFF));
#else
                dma16(rs_in, stage_b + g * RPG * BKB, (last && tail_cut[g]) ? OOB : vb[g]);
#endif
                vb[g] += BKB;
            }
        }
        if (++l_ks == ksteps_per_tap) {
            l_ks = 0;
            ++l_tap;
            if (++l_tap_s == P.KW) { l_tap_s = 0; ++l_tap_r; }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(0);
    if (KT > 1) issue(1);
    if (DIST > 2 && KT > 2) issue(2);
    int buf = 0, nbuf = DIST % NBUF;
    __builtin_amdgcn_s_waitcnt(0xC07F);   // compiler-visible lgkmcnt(0): see gemm_conv_fat_kernel (counted LDS waits inside the loop)
    for (int kt = 0; kt < KT; ++kt) {
        // retire this wave's DMAs of step kt (leave the later steps' in flight), then meet the others
        const int ahead = KT - 1 - kt;  // steps issued beyond kt, capped by DIST-1
        if (DIST > 2 && ahead >= 2) wait_vmcnt<2 * GROUP>();
        else if (ahead >= 1) wait_vmcnt<GROUP>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        // stage kt+2: its ring slot was last read in iteration kt-1, which every wave has left
        if (kt + DIST < KT) issue(nbuf);
        const char* sA = smem + buf * STAGE_BYTES;
#ifndef ABL_NO_MMA   // ablation builds (scripts/ablate_conv.sh): what the K loop costs without its MFMA/LDS-read half
        mma_slab<T, BKB>(sA, sA + TILE_BYTES, wave_c, wave_p, lane, acc);
#endif
        buf = (buf + 1 == NBUF) ? 0 : buf + 1;
        nbuf = (nbuf + 1 == NBUF) ? 0 : nbuf + 1;
    }

#ifdef ABL_NO_EPI   // ablation: no stores / statistics (one never-taken store keeps the K loop alive)
    float chk = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) chk += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (chk == 1.2345e-33f) reinterpret_cast<float*>(P.out)[0] = chk;
#else
    if (SPLIT) {   // partial sums of this K range -> this split's slice of the workspace (plain stores: the sum over
                   // splits is taken in a fixed order by splitk_reduce_kernel, so results do not depend on scheduling)
        const int r16 = lane & 15, q = lane >> 4;
        float* slice = P.ws + (long long)split * P.M * P.NO;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long p = p_base + wave_p * 64 + j * 16 + r16;
            if (p >= P.M) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = c_base + wave_c * 64 + i * 16 + q * 4;
                if (co >= P.NO) continue;
                *reinterpret_cast<f32x4*>(slice + p * P.NO + co) = acc[i][j];
            }
        }
    } else {
        // the fat tile's epilogue (LDS-transposed full-run stores through a rebased descriptor, branch-free): the pointer
        // form above wrote every 128-byte line of the output in four 32-byte pieces from ~40 VALU instructions each
        const int rows_valid = (int)min((long long)TP, P.M - p_base);
        const int grp = P.stat_group_pix ? (int)(p_base / P.stat_group_pix) : 0;
        conv_epilogue_fat<T, 4, 4, TCH / 64, PW>(P, acc, (int)p_base, rows_valid, grp, c_base, wave_c, wave_p, lane, smem);
    }
#endif
}


// ------------------------------------------------------- fat-tile variant ----
// ONE 512-thread workgroup per CU whose staging ring is the CU's whole 160 KiB of LDS, 8 waves as WM x WN, each
// wave an (MI*16) x (NJ*16) sub-tile of MI x NJ accumulators (96 x 112 = 168 VGPRs at MI 6, NJ 7).  Compared with
// the 64 x 64-per-wave tiles above: 13 fragment reads for 42 MFMAs instead of 8 for 16 (LDS read traffic per FLOP
// 0.6x), 5 LDS-DMA pieces per wave for 42 MFMAs instead of 3 for 16 (DMA issue per FLOP 0.63x), operand bytes through
// the DMA path per FLOP 0.6x (384 x 224: 141 FLOP/B against 85 for 256 x 128), and one barrier per 42 MFMAs.
// The pixel width of a tile is a RUN-TIME number tn_valid <= NJ*WN*16 (rows beyond it are fetched as out-of-range
// zeros and not stored), chosen by the launcher so that the tile count is a whole number of rounds of the 256 CUs:
// 8 x 72 x 48 pixels x 728 channels = 2 x 128 tiles of 384 x 216 -- one per CU -- where 256 x 128 tiles needed two
// rounds for 1.27 rounds of work.  Staging: A rows [s*128 + wave*16, +16) and B rows likewise per slot s, so every
// wave issues the same SA + SB pieces per K-step (uniform counted waits); B slots beyond the tile are dummies.
// Epilogue of the fat tile.  A lane's accumulators are 4 consecutive channels of one pixel: stored as they stand, a
// wave instruction writes sixteen 32-byte pieces of sixteen pixel rows, and the memory side sees every 128-byte line of
// the output four times (measured with in-kernel stamps: 10 us of a 39 us launch for 40 MB; 16 us when the four pieces
// of a line were issued seven instructions apart).  So each wave transposes its sub-tile through a private LDS region,
// one 16-pixel block at a time (the staging ring is free by then): ds_write in accumulator layout, ds_read_b128 so that
// a lane holds 16 contiguous bytes of a pixel row and a pixel's MI*32 bytes are written as ONE run by consecutive lanes.
// Stores go through a buffer descriptor REBASED to the tile's first pixel (64-bit base per workgroup, small 32-bit
// offsets: no 2 GiB limit on the tensor); invalid rows / channels carry the out-of-range marker and are dropped by the
// hardware: no exec-mask branches, no 64-bit address arithmetic per store.  Statistics need no masks either: rows
// beyond the tile and channels beyond Cout accumulated zeros.
// y[r][c] = T(sum_s ws[s][r][c]), splits summed in index order (deterministic)
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* ws, int splits, long long rows, int C, T* y, int ldy) {
    const int c4 = C / 4;
    const long long total = rows * c4, slice = rows * (long long)C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        const float* p = ws + r * C + c;
        f32x4 a = *reinterpret_cast<const f32x4*>(p);
        for (int s_ = 1; s_ < splits; ++s_) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(p + s_ * slice);
            a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
        }
        T* o = y + r * ldy + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f(a[e]);
    }
}

// ------------------------------------------------ small-Cin data gradient ----
// Data gradient of the networks' FIRST convolution (3x3, stride 2, pad 1, Cin <= 16 field channels): the implicit
// GEMM above would spend a 128-row tile on 16 output channels and three of four taps on the stride lattice's
// holes (38 TFLOP/s).  Here a wave owns 16 output pixels of one row and one column parity -- their contributing
// (ho, wo) are then consecutive and the taps of the parity class are known (1, 2, 2 or 4 of the 9) -- and the 16
// input channels are exactly one MFMA 16x16x32 tile: A = W^T[ci][co] from the CRSK copy, B = dy[pixel][co], both
// K-contiguous 16-byte loads straight into registers (no LDS: nothing is shared between waves).
struct SmallCDgradParams {
    const bf16_t* dy; const bf16_t* wt; bf16_t* dx;
    int N, H, W, Ho, Wo, Co, CKp, ldy, ldx, Ci, tiles_w;
    long long tiles;
};

template <int CO>
__global__ __launch_bounds__(256) void dgrad_s2_smallc_kernel(SmallCDgradParams P) {
    constexpr int NK = CO / 32;
    const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
    // the four waves of a block take the four parity classes (row parity, column parity): a wave's tap set is fixed,
    // so its weight fragments are loaded once and stay in registers
    const int cls = threadIdx.x >> 6, ph = cls >> 1, pw = cls & 1;
    const int nr = ph ? 2 : 1, ns = pw ? 2 : 1;
    const bool a_ok = r16 < P.Ci;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 a[2][2][NK];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = ph ? 2 * i : 1, s_ = pw ? 2 * j : 1;
            const bf16_t* ap = P.wt + ((long long)(a_ok ? r16 : 0) * 9 + r * 3 + s_) * P.CKp + q * 8;
#pragma unroll
            for (int k = 0; k < NK; ++k) a[i][j][k] = (a_ok && i < nr && j < ns) ? *reinterpret_cast<const bf16x8*>(ap + k * 32) : zero8;
        }
    const int rows_c = (P.H - ph + 1) / 2;                       // rows of this parity
    const long long tiles_c = (long long)P.N * rows_c * P.tiles_w;
    for (long long tile = blockIdx.x; tile < tiles_c; tile += gridDim.x) {
        long long t = tile;
        const int tw = (int)(t % P.tiles_w); t /= P.tiles_w;
        const int h = 2 * (int)(t % rows_c) + ph;
        const int n = (int)(t / rows_c);
        const int w = tw * 32 + pw + 2 * r16;       // this lane's output column (B column / stored pixel)
        bf16x8 b[2][2][NK];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = ph ? 2 * i : 1, s_ = pw ? 2 * j : 1;
                const int ho = (h + 1 - r) >> 1, wo = (w + 1 - s_) >> 1;
                const bool ok = i < nr && j < ns && w < P.W && (unsigned)ho < (unsigned)P.Ho && (unsigned)wo < (unsigned)P.Wo;
                const bf16_t* bp = P.dy + (((long long)n * P.Ho + (ok ? ho : 0)) * P.Wo + (ok ? wo : 0)) * P.ldy + q * 8;
#pragma unroll
                for (int k = 0; k < NK; ++k) b[i][j][k] = ok ? *reinterpret_cast<const bf16x8*>(bp + k * 32) : zero8;
            }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (i < nr && j < ns) {              // wave-uniform
#pragma unroll
                    for (int k = 0; k < NK; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][j][k], b[i][j][k], acc, 0, 0, 0);
                }
            }
        const int ci = q * 4;
        if (w < P.W && ci < P.Ci) {
            bf16x4 o = {(bf16_t)acc[0], (bf16_t)acc[1], (bf16_t)acc[2], (bf16_t)acc[3]};
            *reinterpret_cast<bf16x4*>(P.dx + (((long long)n * P.H + h) * P.W + w) * P.ldx + ci) = o;
        }
    }
}

// ------------------------------------------------------------------ wgrad ----
struct WgradParams {
    const void* x;
    const void* dy;
    float* dw;
    int N, H, W, Ho, Wo;
    int Ci, Co;  // physical channel counts
    int ldx, ldy;
    int KH, KW, stride, pad, dil;
    long long M;
    long long pix_per_split;  // multiple of 32
    int tiles_co, tiles_ci;
    int x_bytes, dy_bytes;    // exact operand extents (buffer-load range check)
    int fold;                 // 1: the (tap, ci) pairs are ONE column dimension of RS*Ci (im2col'd x operand) -- the
                              // few-channel first layers, where a tap per tile would re-read dy nine times for
                              // 16 useful columns of 128
    float* ws;                // not NULL: split s of tile t stores its 128 x 128 partial sums at ws[(s * tiles + t)] (plain
                              // stores) and wgrad_ws_reduce_kernel adds them to dw in split order -- no atomics, and the
                              // result does not depend on the order in which the workgroups finish
    int splits;
};

constexpr int WG_PIX = 32;  // pixels per K-chunk

template <typename T>
__device__ __forceinline__ constexpr int wg_rowb() {
    return TILE * (int)sizeof(T) + 64;  // +64 B pad: conflict-free transposed / column reads
}

template <typename T>
__device__ __forceinline__ void wgrad_mma(const char* sA, const char* sB, int wave_m, int wave_n, int lane,
                                          f32x16 (&acc)[2][2]) {
    constexpr int ROWB = wg_rowb<T>();
    if constexpr (sizeof(T) == 2) {
        const int gi = lane >> 4, ii = lane & 15, qq = ii >> 2, pp = ii & 3;
        const int cb = (gi & 1) * 16, h = gi >> 1;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#pragma unroll
        for (int ks = 0; ks < WG_PIX / 16; ++ks) {
            bf16x8 a[2], b[2];
            const int rowp = ks * 16 + 8 * h + qq;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* pa = sA + rowp * ROWB + (wave_m * 64 + i * 32 + cb + 4 * pp) * 2;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + 4 * ROWB));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                a[i] = __builtin_bit_cast(bf16x8, t);
                const char* pb = sB + rowp * ROWB + (wave_n * 64 + i * 32 + cb + 4 * pp) * 2;
                s16x4 lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb));
                s16x4 hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb + 4 * ROWB));
                s16x8 t2 = {lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
                b[i] = __builtin_bit_cast(bf16x8, t2);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    } else {
        const int r32 = lane & 31, h = lane >> 5;
#pragma unroll 4
        for (int kk = 0; kk < WG_PIX / 2; ++kk) {
            float a[2], b[2];
            const int rowp = 2 * kk + h;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const float*>(sA + rowp * ROWB + (wave_m * 64 + i * 32 + r32) * 4);
                b[i] = *reinterpret_cast<const float*>(sB + rowp * ROWB + (wave_n * 64 + i * 32 + r32) * 4);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(NTHREADS) void wgrad_kernel(WgradParams P) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int ROWB = wg_rowb<T>();
    constexpr int CPR = TILE * sizeof(T) / 16;  // chunks per tile row (16 or 32)
    constexpr int RPP = NTHREADS / CPR;         // rows per pass (16 or 8)
    constexpr int NPASS = WG_PIX / RPP;         // 2 or 4
    constexpr int TILE_BYTES = WG_PIX * ROWB;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;

    const int RS = P.KH * P.KW;
    // 1-D grid of (split, tile) pairs.  Workgroups with the same id % 8 share an XCD (and its
    // L2); give each XCD a CONTIGUOUS range of the split-major order, so the tiles that re-read
    // one split's pixels (dy for every ci tile, x for every co tile) hit in one L2 instead of
    // being fetched from HBM by all eight.
    const int tiles = P.fold ? P.tiles_co * P.tiles_ci : P.tiles_co * P.tiles_ci * RS;
    int lin = blockIdx.x;
    {
        const int nblk = gridDim.x;
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = lin & 7, k = lin >> 3;
        lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int split = lin / tiles;
    int bid = lin - split * tiles;
    const int tile_ci = bid % P.tiles_ci;
    bid /= P.tiles_ci;
    const int tap = P.fold ? 0 : bid % RS;
    const int tile_co = P.fold ? bid : bid / RS;
    const int co_base = tile_co * TILE, ci_base = tile_ci * TILE;

    const long long p_begin = (long long)split * P.pix_per_split;
    long long p_end = p_begin + P.pix_per_split;
    if (p_end > P.M) p_end = P.M;
    if (p_begin >= p_end) return;  // uniform per block

    const int chunk = tid % CPR, row0 = tid / CPR;
    constexpr int ES = (int)sizeof(T);
    const bool co_ok = co_base + chunk * VEC < P.Co;
    // this thread's x chunk: channel offset and tap (per block normally; per chunk when the taps are folded in)
    int cix = ci_base + chunk * VEC, my_tap = tap;
    bool ci_ok = cix < P.Ci;
    if (P.fold) {
        my_tap = cix / P.Ci;
        cix -= my_tap * P.Ci;
        ci_ok = my_tap < RS;
    }
    const int r = my_tap / P.KW, s = my_tap - r * P.KW;
    const int ohw = P.Ho * P.Wo;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.x), 0, P.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.dy), 0, P.dy_bytes, 0x00020000);

    // Per-thread row cursors, advanced by WG_PIX pixels per chunk with adds only (the
    // (n, oh, ow) decomposition of the first pixel is the only division in the kernel).
    int row_p[NPASS];           // pixel index relative to p_begin
    int off_dy[NPASS];          // byte offset of dy[p, co_base + chunk]
    int r_n[NPASS], r_oh[NPASS], r_ow[NPASS];
    const int span = (int)(p_end - p_begin);
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        row_p[i] = row0 + i * RPP;
        const unsigned pu = (unsigned)(p_begin + row_p[i]);
        off_dy[i] = (int)(pu * (unsigned)P.ldy + (unsigned)(co_base + chunk * VEC)) * ES;
        const unsigned n = pu / (unsigned)ohw;
        const unsigned rem = pu - n * (unsigned)ohw;
        const unsigned q = rem / (unsigned)P.Wo;
        r_n[i] = (int)n;
        r_oh[i] = (int)q;
        r_ow[i] = (int)(rem - q * (unsigned)P.Wo);
    }
    const int dy_step = WG_PIX * P.ldy * ES;

    Chunk<T> ra[NPASS], rb[NPASS];
    auto issue_loads = [&]() {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const bool pv = row_p[i] < span;
#ifdef ABL_W_NO_LOAD
            ra[i].v = __builtin_bit_cast(typename Elem<T>::vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, OOB, 0, 0));
#else
            ra[i].v = __builtin_bit_cast(typename Elem<T>::vec_t,
                                         __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (pv && co_ok) ? off_dy[i] : OOB, 0, 0));
#endif
            int ox = OOB;
            if (pv && ci_ok) {
                const int ih = r_oh[i] * P.stride - P.pad + r * P.dil;
                const int iw = r_ow[i] * P.stride - P.pad + s * P.dil;
                if ((unsigned)ih < (unsigned)P.H && (unsigned)iw < (unsigned)P.W)
                    ox = (((r_n[i] * P.H + ih) * P.W + iw) * P.ldx + cix) * ES;
            }
#ifdef ABL_W_NO_LOAD
            ox = OOB;
#endif
            rb[i].v = __builtin_bit_cast(typename Elem<T>::vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs_x, ox, 0, 0));
            // advance this row's cursor to the next chunk
            row_p[i] += WG_PIX;
            off_dy[i] += dy_step;
            r_ow[i] += WG_PIX;
            while (r_ow[i] >= P.Wo) {
                r_ow[i] -= P.Wo;
                if (++r_oh[i] == P.Ho) { r_oh[i] = 0; ++r_n[i]; }
            }
        }
    };
    auto write_lds = [&](int buf) {
        char* sA = smem + buf * 2 * TILE_BYTES;
        char* sB = sA + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int row = row0 + i * RPP;
            ra[i].store(reinterpret_cast<T*>(sA + row * ROWB + chunk * 16));
            rb[i].store(reinterpret_cast<T*>(sB + row * ROWB + chunk * 16));
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    issue_loads();
    write_lds(0);
    __syncthreads();
    int cur = 0;
    for (long long pc = p_begin; pc < p_end; pc += WG_PIX) {
        const bool more = pc + WG_PIX < p_end;
        if (more) issue_loads();
        const char* sA = smem + cur * 2 * TILE_BYTES;
#ifndef ABL_W_NO_MMA
        wgrad_mma<T>(sA, sA + TILE_BYTES, wave_m, wave_n, lane, acc);
#endif
        if (more) write_lds(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // accumulate: 32x32 accumulator layout: col = lane&31 (ci), row = (e&3)+8*(e>>2)+4*(lane>>5) (co)
    const int c32 = lane & 31, h = lane >> 5;
    if (P.ws) {   // block-uniform: the whole tile as it stands, [co_local][ci_local], 128-byte runs per wave store
        float* slice = P.ws + ((long long)split * tiles + (lin - split * tiles)) * (TILE * TILE);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int col = wave_m * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, cil = wave_n * 64 + j * 32 + c32;
                    slice[col * TILE + cil] = acc[i][j][e];
                }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int ci = ci_base + wave_n * 64 + j * 32 + c32, tap_e = tap;
            if (P.fold) {
                tap_e = ci / P.Ci;
                ci -= tap_e * P.Ci;
                if (tap_e >= RS) continue;
            } else if (ci >= P.Ci) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co_base + wave_m * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (co >= P.Co) continue;
#ifndef ABL_W_NO_ATOMIC
                atomicAdd(P.dw + ((long long)co * RS + tap_e) * P.Ci + ci, acc[i][j][e]);
#else
                if (acc[i][j][e] == 12345.678f) P.dw[0] = 1.f;  // keeps the accumulators alive
#endif
            }
        }
}

// dw += sum over splits (in split order) of the tiles wgrad_kernel left in the workspace: one thread per element of dW
__global__ __launch_bounds__(256) void wgrad_ws_reduce_kernel(WgradParams P) {
    const int RS = P.KH * P.KW;
    const int tiles = P.fold ? P.tiles_co * P.tiles_ci : P.tiles_co * P.tiles_ci * RS;
    const long long total = (long long)tiles * TILE * TILE;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int t = (int)(i / (TILE * TILE)), r = (int)(i - (long long)t * (TILE * TILE));
        const int col = r / TILE, cil = r - col * TILE;
        int bid = t;
        const int tile_ci = bid % P.tiles_ci;
        bid /= P.tiles_ci;
        int tap = P.fold ? 0 : bid % RS;
        const int tile_co = P.fold ? bid : bid / RS;
        const int co = tile_co * TILE + col;
        int ci = tile_ci * TILE + cil;
        if (P.fold) {
            tap = ci / P.Ci;
            ci -= tap * P.Ci;
            if (tap >= RS) continue;
        } else if (ci >= P.Ci) continue;
        if (co >= P.Co) continue;
        const float* src = P.ws + (long long)t * (TILE * TILE) + r;
        float a = 0.f;
        for (int s_ = 0; s_ < P.splits; ++s_) a += src[(long long)s_ * tiles * (TILE * TILE)];
        P.dw[((long long)co * RS + tap) * P.Ci + ci] += a;
    }
}

// ------------------------------------------------- gang weight gradient ----
// Weight gradient of L pointwise convolutions of ONE shape (the 48 + 2 identical 728 -> 728 layers of the middle flow;
// any single 1x1 layer is the L = 1 case) as one launch, issued when the backward pass has produced all their
// gradients: dW_l[co][ci] += sum_p dy_l[p][co] * x_l[p][ci].  The reduction runs over pixels -- the slow dimension of both
// operands -- and the output is tiny (9 tiles of 256 x 256 for 728 x 728), so a single layer can only fill the chip by
// splitting its pixels over many workgroups that all add a whole copy of dW with float atomics (the per-layer kernel
// above: 14 splits, a third of its time in atomics, operands re-streamed per split).  Here the (layer, pixel) space of
// the whole group is cut into `gangs` equal ranges; gang g = the T workgroups of the T output tiles walk range g
// together (the T tiles read the same dy / x rows at the same time: one trip to HBM, the rest from L2 / Infinity
// Cache), each keeps its 256 x 256 tile in registers across a layer's pixels and adds it to dW_l when the layer (or the
// range) ends: with ranges at least one layer long every dW tile receives at most two adds -- fp32 addition commutes, so
// the result does not depend on their order: deterministic.  Staging: 32-pixel K-steps, both operands global -> LDS by
// DMA (a 1-KiB piece = 2 pixel rows x 256 channels) into a 4-stage ring, pieces issued between the MFMA groups;
// fragments by ds_read_b64_tr_b16 (the reduction index is the LDS row); the 32-byte granules of a row are XOR-swizzled
// by (row & 3) << 1 on the source side, which makes the transposing reads conflict-free without row padding.
constexpr int WGG_MAX_LAYERS = 64;
struct WgGangParams {
    long long tbl[WGG_MAX_LAYERS][4];   // x, dy, dw addresses per layer (+ its tap, when taps travel as layers): in the kernel
                                        // arguments (2 KB), no table in memory
    int L, M, ldx, ldy, Ci, Co;
    // k x k convolutions (stride 1, "same" padding; TAPS kernels): a tap (r, s) is a 1x1 weight gradient against x shifted
    // by ((r - KH/2) * dil) rows and ((s - KW/2) * dil) columns, border pixels masked.  With few tiles the RS taps are
    // MEMBERS of the gang (taps_in_members: the nine workgroups walk the same pixels together, dy and the nine shifted
    // windows of x come out of one L2), otherwise every tap is a layer of its own (tbl[l][3]).
    int N, H, W, KH, KW, dil, RS;
    int tpt;                // tiles per tap (= tiles_co * tiles_ci)
    int taps_in_members;
    int tiles_ci, tiles;    // tiles per layer
    int KS;                 // K-steps (32 pixels) per layer
    int R;                  // K-steps per gang
    int gangs;              // ranges of the (layer, pixel) space
    int use_map;            // 1: (gang, member) of a workgroup come from `map` (XCD-local placement), 0: blockIdx / tiles
    unsigned short map[256];   // blockIdx -> gang << 8 | member, 0xffff = idle.  Workgroup b runs on XCD b % 8: the launcher
                            // puts the T workgroups of a gang on ONE XCD, so the rows the tiles share (each dy / x row is wanted
                            // by 3 of the 9 tiles of a 728 x 728 layer) are fetched once into that XCD's L2 -- spread over the
                            // XCDs every copy went to the fabric: 7.3 GB fetched for 3.9 GB of operands (rocprofv3 FETCH_SIZE),
                            // XCD-local 3.86 GB.  The CUs an XCD has left over (32 - 3 x 9 = 5) form gangs that straddle two
                            // neighbouring XCDs, so 252 of the 256 CUs work.
};

constexpr int WGG_PIX = 32, WGG_ROWB = 512, WGG_OP = WGG_PIX * WGG_ROWB, WGG_STAGE = 2 * WGG_OP;

// One MFMA 32x32x16 operand fragment of 16-pixel half H: two transposing 8-byte reads (pixel rows r and r + 4 of the half).
template <int OFF>
__device__ __forceinline__ s16x4 wgg_tr16(unsigned addr) {
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int H>
__device__ __forceinline__ bf16x8 wgg_frag(unsigned addr, int kt, int lane) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
#ifdef ABL_G_NO_FRAG   // ablation: no LDS reads, the MFMAs take a register pattern
    s16x8 tt = {(short)addr, (short)kt, (short)lane, 1, 2, 3, 4, 5};
    asm volatile("" : "+v"(tt));
    return __builtin_bit_cast(bf16x8, tt);
#else
    (void)kt; (void)lane;
    const s16x4 lo = wgg_tr16<H * 16 * WGG_ROWB>(addr);
    const s16x4 hi = wgg_tr16<H * 16 * WGG_ROWB + 4 * WGG_ROWB>(addr);
    const s16x8 tt = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, tt);
#endif
}

template <int WGG_NBUF, bool TAPS>
__global__ __launch_bounds__(512) void wgrad_gang_kernel(WgGangParams P) {
    typedef __attribute__((address_space(3))) char* lds_char_p;
    constexpr int GROUP = 4, DIST = WGG_NBUF - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds_base = (unsigned)(size_t)(lds_char_p)smem;   // byte address of the ring inside the LDS
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2, wave_n = wave & 3;      // 2 x 4 waves: 128 (co) x 64 (ci) each
    int gang, member;
    if (P.use_map) {
        const unsigned gm = P.map[blockIdx.x];
        if (gm == 0xffffu) return;      // a left-over CU
        gang = gm >> 8;
        member = gm & 255;
    } else {
        gang = blockIdx.x / P.tiles;
        member = blockIdx.x - gang * P.tiles;
    }
    if (gang >= P.gangs) return;
    int tap_m = 0;
    if (TAPS && P.taps_in_members) {
        tap_m = member / P.tpt;
        member -= tap_m * P.tpt;
    }
    const int tile_co = member / P.tiles_ci, tile_ci = member - tile_co * P.tiles_ci;
    const int co_base = tile_co * 256, ci_base = tile_ci * 256;
    const long long total = (long long)P.L * P.KS;
    long long g = (long long)gang * P.R;
    const long long gend = min(g + (long long)P.R, total);

    // DMA geometry of this lane: piece k (0, 1) of an operand covers tile rows 2*wave + 16*k + (lane >> 5)
    const int rsub = lane >> 5, slot = lane & 31;
    const int sw = ((wave & 1) * 2 + rsub) & 3;             // (row & 3) of both pieces
    const int chunk = slot ^ (sw << 2);                     // source 16-byte chunk that lands in LDS slot `slot`
    const int row_a = 2 * wave + rsub;
    const bool a_ok = co_base + chunk * 8 < P.Co, b_ok = ci_base + chunk * 8 < P.Ci;
    // fragment read geometry (see wgrad_mma)
    const int gi = lane >> 4, ii = lane & 15, qq = ii >> 2, pp = ii & 3;
    const int cb = (gi & 1) * 16, hh = gi >> 1;
    // per-lane LDS byte offsets of the fragments inside an operand tile, K sub-step 0, low half (the XOR moves bits 6-7
    // only, where the fragment index i / j lives: one register per fragment, everything else is an immediate)
    int off_a[4], off_b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) off_a[i] = (8 * hh + qq) * WGG_ROWB + (((wave_m * 128 + i * 32 + cb + 4 * pp) * 2) ^ (qq << 6));
#pragma unroll
    for (int j = 0; j < 2; ++j) off_b[j] = WGG_OP + (8 * hh + qq) * WGG_ROWB + (((wave_n * 64 + j * 32 + cb + 4 * pp) * 2) ^ (qq << 6));

    while (g < gend) {
        const int l = (int)(g / P.KS);
        const int ks0 = (int)(g - (long long)l * P.KS);
        const int ks1 = (int)min((long long)P.KS, (long long)ks0 + (gend - g));
        const bf16_t* xp = reinterpret_cast<const bf16_t*>(P.tbl[l][0]);
        const bf16_t* dyp = reinterpret_cast<const bf16_t*>(P.tbl[l][1]);
        float* dw = reinterpret_cast<float*>(P.tbl[l][2]);
        // descriptors rebased to the segment's first pixel; rows beyond M are out of range = zeros
        const long long p0 = (long long)ks0 * WGG_PIX;
        int tap = 0, dr = 0, dc = 0;
        long long shift = 0;            // rows between a dy pixel and the x pixel its tap multiplies
        if (TAPS) {
            tap = P.taps_in_members ? tap_m : (int)P.tbl[l][3];
            const int r = tap / P.KW, s_ = tap - r * P.KW;
            dr = (r - (P.KH >> 1)) * P.dil;
            dc = (s_ - (P.KW >> 1)) * P.dil;
            shift = (long long)dr * P.W + dc;
        }
        const long long rem_a = (((long long)P.M - p0 - 1) * P.ldy + P.Co) * 2,
                        rem_b = (((long long)P.M - p0 - shift - 1) * P.ldx + P.Ci) * 2;
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<bf16_t*>(dyp + p0 * P.ldy), 0, bg_records(rem_a), 0x00020000);
        // (with a negative shift the base lies in front of the tensor for the first rows: those lanes are masked below)
        // (a segment near the end with a positive shift starts beyond the tensor: no records at all -- a negative count
        // would read as 4 G records and let the out-of-range marker through)
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<bf16_t*>(xp + (p0 + shift) * P.ldx), 0, bg_records(rem_b), 0x00020000);
        int va[2], vb[2];
        int xn[2], xh[2], xw[2];        // TAPS: image, row, column of the dy pixel each x piece row belongs to
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            va[k] = a_ok ? ((row_a + 16 * k) * P.ldy + co_base + chunk * 8) * 2 : OOB;
            vb[k] = b_ok ? ((row_a + 16 * k) * P.ldx + ci_base + chunk * 8) * 2 : OOB;
            if (TAPS) {
                const unsigned pu = (unsigned)(p0 + row_a + 16 * k), hw = (unsigned)(P.H * P.W);
                const unsigned n_ = pu / hw, rem = pu - n_ * hw, h_ = rem / (unsigned)P.W;
                xn[k] = (int)n_; xh[k] = (int)h_; xw[k] = (int)(rem - h_ * (unsigned)P.W);
            }
        }
        const int step_a = WGG_PIX * P.ldy * 2, step_b = WGG_PIX * P.ldx * 2;
        auto piece = [&](int buf, int pi) {   // pi compile-time: 0, 1 = dy pieces, 2, 3 = x pieces
            char* st = smem + buf * WGG_STAGE + (pi >= 2 ? WGG_OP : 0) + (2 * wave + 16 * (pi & 1)) * WGG_ROWB;
#if defined(ABL_G_SKIP_DMA)    // ablation builds (scripts/ablate_gang.sh): no DMA instruction at all
            (void)st;
#elif defined(ABL_G_NO_DMA)    // the instruction issues, every lane out of range: no memory traffic
            dma16(rs_a, st, OOB);
#elif defined(ABL_G_HOT)       // every piece re-reads the segment's first rows (L2-resident operands)
            if (pi < 2) dma16(rs_a, st, va[pi]); else dma16(rs_b, st, vb[pi - 2]);
#else
            if (pi < 2) { dma16(rs_a, st, va[pi]); va[pi] += step_a; }
            else if (!TAPS) { dma16(rs_b, st, vb[pi - 2]); vb[pi - 2] += step_b; }
            else {
                const int k = pi - 2;
                const bool in = xn[k] < P.N && (unsigned)(xh[k] + dr) < (unsigned)P.H && (unsigned)(xw[k] + dc) < (unsigned)P.W;
                dma16(rs_b, st, in ? vb[k] : OOB);
                vb[k] += step_b;
                xw[k] += WGG_PIX;
                while (xw[k] >= P.W) {
                    xw[k] -= P.W;
                    if (++xh[k] == P.H) { xh[k] = 0; ++xn[k]; }
                }
            }
#endif
        };
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int KT = ks1 - ks0;
        __syncthreads();   // the previous segment's readers are done with the ring
#pragma unroll
        for (int s_ = 0; s_ < DIST; ++s_)
            if (s_ < KT) {
#pragma unroll
                for (int pi = 0; pi < GROUP; ++pi) piece(s_, pi);
            }
        // K loop, software-pipelined across the K-steps: a step's barrier sits in its MIDDLE.  The fragments of a
        // 16-pixel half are read one half-step before their MFMAs (eight MFMAs of the wave cover the LDS latency), and the
        // wait for the next stage's pieces + the workgroup barrier come after the first half's MFMAs, when the second
        // half's operands are already in registers: a wave leaves the barrier straight into MFMAs.  (With the barrier at
        // the top of the step both waves of a SIMD stood behind it with empty hands: 12 reads and their latency, every
        // step, with the matrix pipe idle -- 1 500 cycles per step against 1 024 of MFMA even without any DMA.)
        //   ring safety: pieces issued in step kt go to the buffer of stage kt + DIST = the buffer read in step kt - 1,
        //   whose last reads every wave has waited for before the mid-step barrier of kt - 1;
        //   the reads of stage kt + 1 start after the mid-step barrier of kt, behind every wave's own counted vmcnt.
        int buf = 0, nbuf = DIST % WGG_NBUF;
        bf16x8 fa[2][4], fb[2][2];
        auto stage_addr = [&](int bf, unsigned (&ad_a)[4], unsigned (&ad_b)[2]) {
            const unsigned sbase = lds_base + bf * WGG_STAGE;
#pragma unroll
            for (int i = 0; i < 4; ++i) ad_a[i] = sbase + off_a[i];
#pragma unroll
            for (int j = 0; j < 2; ++j) ad_b[j] = sbase + off_b[j];
        };
        // Fragment reads are inline assembly ON PURPOSE: LLVM's waitcnt pass cannot prove that an LDS read carrying a memory
        // operand does not alias the destination of an outstanding `buffer_load ... lds`, so it put `s_waitcnt vmcnt(0)`
        // in front of every ds_read_b64_tr_b16 intrinsic -- each K-step waited for ALL pieces in flight, those issued a
        // moment before included, and the ring had no depth at all (scripts/find_dma_waits.py lists such waits).  The
        // ring's counted vmcnt + barrier order the DMA writes against these reads; lgkmcnt is counted by hand (the LDS
        // returns in order and the loop holds no scalar loads).
        {   // stage 0 has landed everywhere; first half of step 0
            if (KT > 2) wait_vmcnt<2 * GROUP>();
            else if (KT > 1) wait_vmcnt<GROUP>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            unsigned ad_a[4], ad_b[2];
            stage_addr(0, ad_a, ad_b);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[0][j] = wgg_frag<0>(ad_b[j], 0, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[0][i] = wgg_frag<0>(ad_a[i], 0, lane);
        }
        for (int kt = 0; kt < KT; ++kt) {
            const int ahead = KT - 1 - kt;
            const bool more = ahead >= DIST;
            unsigned ad_a[4], ad_b[2];
            stage_addr(buf, ad_a, ad_b);
            // second half's fragments of this step
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[1][j] = wgg_frag<1>(ad_b[j], kt, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[1][i] = wgg_frag<1>(ad_a[i], kt, lane);
            asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fb[0][0]), "+v"(fb[0][1]));
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks == 1) {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]), "+v"(fb[1][0]), "+v"(fb[1][1]));
                    if (ahead >= 1) {
                        // stage kt + 1 complete: stages kt + 2 .. kt + DIST - 1 and this step's first two pieces may stay in flight
                        if (ahead >= DIST) wait_vmcnt<(DIST - 2) * GROUP + GROUP / 2>();
                        else if (DIST >= 4 && ahead >= 3) wait_vmcnt<2 * GROUP>();
                        else if (ahead >= 2) wait_vmcnt<GROUP>();
                        else wait_vmcnt<0>();
                        __builtin_amdgcn_s_barrier();
                        const int b1 = (buf + 1 == WGG_NBUF) ? 0 : buf + 1;
                        stage_addr(b1, ad_a, ad_b);
#pragma unroll
                        for (int j = 0; j < 2; ++j) fb[0][j] = wgg_frag<0>(ad_b[j], kt, lane);
#pragma unroll
                        for (int i = 0; i < 4; ++i) fa[0][i] = wgg_frag<0>(ad_a[i], kt, lane);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more && (i & 1) == 0) piece(nbuf, ks * 2 + (i >> 1));   // 4 pieces over the 8 MFMA groups of a step
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#ifdef ABL_G_NO_MFMA   // ablation: the fragments are consumed by one VALU op each instead of the MFMAs
                        acc[i][j][0] += __builtin_bit_cast(f32x4, fa[ks][i])[0] * __builtin_bit_cast(f32x4, fb[ks][j])[1];
#else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
#endif
                    }
                    __builtin_amdgcn_s_setprio(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            buf = (buf + 1 == WGG_NBUF) ? 0 : buf + 1;
            nbuf = (nbuf + 1 == WGG_NBUF) ? 0 : nbuf + 1;
        }
        // flush: 32x32 accumulator layout: col = lane & 31 (ci), row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) (co).
        // Buffer atomics over a descriptor of exactly this layer's dW: rows beyond Cout fall out of its range, lanes
        // beyond Cin carry the out-of-range marker -- no exec-mask branches, one VALU add per atomic.
        {
            // dW is [Co][RS][Ci]: this tap's columns start at tap * Ci, a row is RS * Ci wide
            const int rowb = P.RS * P.Ci * 4;
            const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(dw + tap * P.Ci, 0, (P.Co - 1) * rowb + P.Ci * 4, 0x00020000);
            const int c32 = lane & 31, h2 = lane >> 5;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ci = ci_base + wave_n * 64 + j * 32 + c32;
                int lane_off = ci < P.Ci ? (co_base + wave_m * 128 + 4 * h2) * rowb + ci * 4 : OOB;
                asm volatile("" : "+v"(lane_off));   // keeps the 64 derived offsets out of the segment loop's invariants (they spilled)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[i][j][e], rs_w, lane_off + (i * 32 + (e & 3) + 8 * (e >> 2)) * rowb, 0, 0);
            }
        }
        g += KT;
    }
}

// ------------------------------------------------------- batched weight packing
// Per dense-conv layer: src [K][RS][C] (the flat bf16/fp32 copy of the master weights)
//   -> dst_k [K][RS][Cp]  forward operand, reduction dim zero-padded to the K-step granule
//   -> dst_t [C][RS][Kp]  data-gradient operand (CRSK), likewise padded
// so the GEMM K loop never needs a tail predicate.  tbl: 8 int64 per layer
// (src_off, dk_off, dt_off, K, RS, C, Cp, Kp), element units.
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const T* __restrict__ src, T* __restrict__ dst_k, T* __restrict__ dst_t,
                                                                const long long* tbl) {
    // Round 3: 16-byte rows for the KRSC copy and a 64 x 64 LDS tile for the transposed (CRSK) one; 32-bit index
    // arithmetic.  (The first version moved single elements behind 64-bit divisions and read the transposed copy with a
    // stride of RS * C elements: 326 us per step for 330 MB.)
    constexpr int VEC = 16 / (int)sizeof(T);
    typedef typename Elem<T>::vec_t vec_t;
    __shared__ T tile[64][64 + VEC];
    const long long* e = tbl + (long long)blockIdx.y * 8;
    const long long so = e[0], dk = e[1], dt = e[2];
    const unsigned K = (unsigned)e[3], RS = (unsigned)e[4], C = (unsigned)e[5], Cp = (unsigned)e[6], Kp = (unsigned)e[7];
    const T zero = Elem<T>::from_f(0.f);
    const T* s_ = src + so;
    T* k_ = dst_k + dk;
    T* t_ = dst_t + dt;
    const bool vec_ok = ((so | dk | dt | (long long)C | Cp | Kp) % VEC) == 0 && (long long)K * RS * Cp < (1LL << 31) && (long long)C * RS * Kp < (1LL << 31);
    if (!vec_ok) {      // odd channel counts / offsets: element by element (tests only)
        const long long nk = (long long)K * RS * Cp, nt = (long long)C * RS * Kp;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nk + nt; i += (long long)gridDim.x * blockDim.x) {
            if (i < nk) {
                const long long c = i % Cp, t = i / Cp;  // t = k*RS + rs
                k_[i] = c < C ? s_[t * C + c] : zero;
            } else {
                const long long j = i - nk;
                const long long k = j % Kp, t = j / Kp;
                const long long rs = t % RS, c = t / RS;
                t_[j] = k < K ? s_[(k * RS + rs) * C + c] : zero;
            }
        }
        return;
    }
    // forward operand [K][RS][Cp]: row t = k*RS + rs, 16-byte chunks
    {
        const unsigned cpv = Cp / VEC, rows = K * RS, n = rows * cpv;
        vec_t zv;
#pragma unroll
        for (int q = 0; q < VEC; ++q) zv[q] = zero;
        for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
            const unsigned t = i / cpv, cv = i - t * cpv;
            const vec_t v = cv * VEC < C ? *reinterpret_cast<const vec_t*>(s_ + (size_t)t * C + cv * VEC) : zv;
            *reinterpret_cast<vec_t*>(k_ + (size_t)t * Cp + cv * VEC) = v;
        }
    }
    // data-gradient operand [C][RS][Kp]: per tap a K x C matrix transposed through a 64 x 64 tile
    {
        const unsigned tk = (Kp + 63) / 64, tc = (C + 63) / 64, ntile = tk * tc * RS;
        const unsigned ty = threadIdx.x / (64 / VEC), tx = threadIdx.x % (64 / VEC);    // 256 threads: 64/VEC chunks x RPP rows
        constexpr unsigned RPP = 256 / (64 / VEC);                                      // rows per pass (32 for bf16, 16 for fp32)
        for (unsigned ti = blockIdx.x; ti < ntile; ti += gridDim.x) {
            const unsigned rs = ti % RS, r = ti / RS, ck = r % tk, cc = r / tk;
            const unsigned k0 = ck * 64, c0 = cc * 64;
            __syncthreads();
#pragma unroll
            for (unsigned pass = 0; pass < 64 / RPP; ++pass) {
                const unsigned k = k0 + pass * RPP + ty, c = c0 + tx * VEC;
                vec_t v;
#pragma unroll
                for (int q = 0; q < VEC; ++q) v[q] = zero;
                if (k < K && c < C) v = *reinterpret_cast<const vec_t*>(s_ + ((size_t)k * RS + rs) * C + c);
#pragma unroll
                for (int q = 0; q < VEC; ++q) tile[pass * RPP + ty][tx * VEC + q] = v[q];
            }
            __syncthreads();
#pragma unroll
            for (unsigned pass = 0; pass < 64 / RPP; ++pass) {
                const unsigned c = c0 + pass * RPP + ty, k = k0 + tx * VEC;
                if (c < C && k < Kp) {
                    vec_t v;
#pragma unroll
                    for (int q = 0; q < VEC; ++q) v[q] = tile[tx * VEC + q][pass * RPP + ty];
                    *reinterpret_cast<vec_t*>(t_ + ((size_t)c * RS + rs) * Kp + k) = v;
                }
            }
        }
    }
}


template <typename T, int BKB, int NBUF, int TCH, int TP>
int launch_small(GemmConvParams& P, hipStream_t st) {
    P.tiles_c = (P.NO + TCH - 1) / TCH;
    P.tiles_p = (int)((P.M + TP - 1) / TP);
    const size_t sh = (size_t)NBUF * (TCH + TP) * BKB;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_dma_kernel<T, BKB, NBUF, TCH, TP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        once = true;
    }
    hipLaunchKernelGGL((gemm_conv_dma_kernel<T, BKB, NBUF, TCH, TP>), dim3((unsigned)(P.tiles_c * P.tiles_p)),
                       dim3(TCH * TP / 64), sh, st, P);
    BG_CHECK_LAUNCH("gemm_conv_dma_kernel(small)");
    return BG_OK;
}

template <typename T>
int launch_gemm_conv(const GemmConvParams& P0, hipStream_t st, int splits = 1) {
    GemmConvParams P = P0;
    const int dt = sizeof(T) == 2 ? BG_BF16 : BG_F32;
    P.CKp = pad_k(P.CK, dt);
    const long long in_bytes = (((long long)P.N * P.IH * P.IW - 1) * P.ldi + P.CK) * (long long)sizeof(T);
    const long long w_bytes = (long long)P.NO * P.KH * P.KW * P.CKp * (long long)sizeof(T);
    const long long out_bytes = ((P.M - 1) * P.ldo + P.NO) * (long long)sizeof(T);
    // The fat-tile kernel rebases its activation / output descriptors per tile; the 64 x 64-per-wave kernels address
    // the whole activation with 32-bit offsets.
    const bool big = in_bytes >= (1LL << 31) || out_bytes >= (1LL << 31);
    if (w_bytes >= (1LL << 31) || (big && splits != 1)) {
        bg_set_error("conv: operand larger than 2 GiB (32-bit buffer offsets)");
        return BG_E_ARG;
    }
    P.in_bytes = big ? 0 : (int)in_bytes;
    P.w_bytes = (int)w_bytes;
    if (splits == 1) {
        const int plan = plan_fat<T>(P, big);
        if (plan) {
            const int tm = plan / 1000, tnp = plan % 1000;
            const long long nblk = (long long)P.tiles_c * P.tiles_p;
            const bool pw1 = P.KH * P.KW == 1;
            static const int fat_bkb = getenv("BGAMD_FAT_BKB") ? atoi(getenv("BGAMD_FAT_BKB")) : 128;   // A/B: 64 = half-line rows, 4 stages
            if (tm == 128) {
                if (pw1) return launch_fat<T, 128, 2, 7, 4, 2, 3, true>(P, nblk, st);
                return launch_fat<T, 128, 2, 7, 4, 2, 3, false>(P, nblk, st);
            }
            if (tnp == 112) {   // 256 x 112: 8 waves along the channels, every wave all 112 pixels
                if (pw1) return launch_fat<T, 128, 2, 7, 8, 1, 3, true>(P, nblk, st);
                return launch_fat<T, 128, 2, 7, 8, 1, 3, false>(P, nblk, st);
            }
            if (fat_bkb == 64) {
                if (tm == 384) return launch_fat<T, 64, 6, 7, 4, 2, 4, true>(P, nblk, st);
                if (pw1) return launch_fat<T, 64, 4, 7, 4, 2, 4, true>(P, nblk, st);
                return launch_fat<T, 64, 4, 7, 4, 2, 4, false>(P, nblk, st);
            }
            if (tm == 384) return launch_fat<T, 128, 6, 7, 4, 2, 2, true>(P, nblk, st);
            if (pw1) return launch_fat<T, 128, 4, 7, 4, 2, 2, true>(P, nblk, st);
            return launch_fat<T, 128, 4, 7, 4, 2, 2, false>(P, nblk, st);
        }
        if (big) {   // the kernels below address the whole activation with 32-bit offsets
            bg_set_error("conv: operand beyond 2 GiB and no fat-tile plan (statistic groups must divide the pixels; "
                         "pixels per group < 2^31)");
            return BG_E_ARG;
        }
    }
    if (splits > 1) {   // split-K (few tiles, long reduction): 128 x 128 tiles, (tiles x splits) blocks, one workspace slice each
        const int kt_all = P.KH * P.KW * ((P.CK + 64 / (int)sizeof(T) - 1) / (64 / (int)sizeof(T)));
        P.kt_per_split = (kt_all + splits - 1) / splits;
        if ((kt_all + P.kt_per_split - 1) / P.kt_per_split != splits) {   // every split must own >= 1 K-step (its slice is read)
            bg_set_error("conv split-K: %d splits do not divide the %d K-steps into non-empty ranges", splits, kt_all);
            return BG_E_ARG;
        }
        P.tiles_c = (P.NO + TILE - 1) / TILE;
        P.tiles_p = (int)((P.M + TILE - 1) / TILE);
        const long long nb = (long long)P.tiles_c * P.tiles_p * splits;
        if (nb > 0x7fffffffLL) {
            bg_set_error("conv: grid too large");
            return BG_E_ARG;
        }
        hipLaunchKernelGGL((gemm_conv_dma_kernel<T, 64, 3, TILE, TILE, true>), dim3((unsigned)nb), dim3(NTHREADS), 3 * 2 * TILE * 64, st, P);
        BG_CHECK_LAUNCH("gemm_conv_dma_kernel(split-K)");
        return BG_OK;
    }
    static const int dma_mode = getenv("BGAMD_DMA") ? atoi(getenv("BGAMD_DMA")) : 1;  // 0: register staging; 1: 3-stage ring, 64-byte rows (default); 2: 128-byte rows where they pad less; 3: 4-stage ring
    static const int tch_max = getenv("BGAMD_TCH") ? atoi(getenv("BGAMD_TCH")) : 256;  // A/B switch: 128 = old tile
    // 256 out-channel rows per tile wherever that does not add padding (NO <= 128 stays on 128 x 128)
    const bool tall = dma_mode == 1 && tch_max >= 256 && P.NO > TILE;
    const int tch = tall ? 256 : TILE;
    // 256 x 256 tile (one 1024-thread workgroup per CU instead of two 256 x 128 ones): a third fewer operand bytes
    // through the LDS-DMA path per FLOP, measured 1.1-1.2x faster per FLOP in the K loop -- but coarser rounds of the
    // chip and no second workgroup to hide prologue / epilogue behind.  Chosen per launch by a round count model
    // (units: one 256 x 128 tile alone on a CU): 512 co-resident 256 x 128 tiles cost 2 per full round, a remainder
    // of <= 256 tiles 1; a round of 256 x 256 tiles costs 2 / 1.13.  Short reductions (< 16 K-steps) stay narrow.
    // BGAMD_TP: 0 = never, 256 = wherever legal, unset = by the model.
    static const int tp_mode = getenv("BGAMD_TP") ? atoi(getenv("BGAMD_TP")) : -1;
    bool wide = tall && tp_mode != 0 && (P.stat_group_pix == 0 || P.stat_group_pix % 256 == 0);
    if (wide && tp_mode < 0) {
        const long long tc = (P.NO + 255) / 256, t128 = tc * ((P.M + 127) / 128), t256 = tc * ((P.M + 255) / 256);
        const long long rem = t128 % 512;
        const double cost128 = 2.0 * (double)(t128 / 512) + (rem == 0 ? 0.0 : rem <= 256 ? 1.0 : 2.0);
        const double cost256 = (double)((t256 + 255) / 256) * (2.0 / 1.13);
        const long long ksteps = (long long)P.KH * P.KW * ((P.CK + 31) / 32);
        wide = ksteps >= 16 && cost256 < 0.98 * cost128;
    }
    // NO <= 128 on very many pixels (the entry flow): 128 x 256 tiles halve the tile count -- these launches are
    // HBM-bound and their statistics epilogue contends on 2 * NO atomic addresses once per tile
    static const long long flat_min = getenv("BGAMD_FLAT_MIN") ? atoll(getenv("BGAMD_FLAT_MIN")) : 2048;
    // (also for the 3 x 3 128 -> 128 layers with their 36 K-steps: in the step's launch table the flat tile runs them in
    // 566 / 525 us forward / data gradient at batch 8 against 706 / 620 with 128 x 128 tiles -- a micro-benchmark of the
    // isolated launch had suggested the opposite by 4 %)
    const bool flat = !tall && dma_mode == 1 && tp_mode != 0 && (P.M + 127) / 128 >= flat_min &&
                      (P.stat_group_pix == 0 || P.stat_group_pix % 256 == 0);
    const int tp = (wide || flat) ? 256 : TILE;
    P.tiles_c = (P.NO + tch - 1) / tch;
    P.tiles_p = (int)((P.M + tp - 1) / tp);
    // Few tiles (the 16 x 16 maps of the 256 x 256 configuration: 728 -> 728 on 2 048 pixels is 48 tiles of 256 x 128
    // for 256 CUs): such a launch is bound by the latency of its K-steps, not by operand bytes.  128 x 128 tiles on
    // twice the CUs, 128-byte rows (half the K-steps) and a 4-stage ring (three steps in flight): 16.5 -> 12.7 us for
    // that launch with its statistics, 21.3 -> 16.7 for 1024 -> 1536, 29.5 -> 22.6 for 1536 -> 2048 (graph-replayed,
    // scripts/bench_conv_one.py); 128 x 64 and 64 x 64 tiles were no faster on 728 channels and slower on the wider
    // layers.  With a layer's weights cold (every launch another copy, beyond the Infinity Cache -- as in a training step)
    // the old tiles take 18.6 us and these 13.5; touching the tile's whole weight panel ahead of the first K-step made
    // both cases 0.8 us slower and the step 3 % (dropped).  Same K order per output element as every other tile shape.
    // BGAMD_SMALL=0 switches it off (A/B).
    static const bool small_on = !(getenv("BGAMD_SMALL") && atoi(getenv("BGAMD_SMALL")) == 0);
    static const long long small_max = getenv("BGAMD_SMALL_MAX") ? atoll(getenv("BGAMD_SMALL_MAX")) : 128;
    if (dma_mode == 1 && small_on && !wide && !flat && (long long)P.tiles_c * P.tiles_p <= small_max &&
        P.stat_group_pix % TILE == 0)
        return launch_small<T, 128, 4, TILE, TILE>(P, st);
    const long long nblk = (long long)P.tiles_c * P.tiles_p;
    if (nblk <= 0 || nblk > 0x7fffffffLL) {
        bg_set_error("conv: grid too large");
        return BG_E_ARG;
    }
    // K-step width: 128-byte rows (BK = 64 bf16 / 32 f32) unless the padded
    // reduction depth is smaller with 64-byte rows (e.g. Cin = 728 -> 736 vs 768).
    const int bk64 = 64 / (int)sizeof(T), bk128 = 128 / (int)sizeof(T);
    const int pad64 = (P.CK + bk64 - 1) / bk64 * bk64, pad128 = (P.CK + bk128 - 1) / bk128 * bk128;
    bool use128 = pad128 <= pad64 + pad64 / 32;
    if (const char* e = getenv("BGAMD_BKB")) use128 = atoi(e) == 128;  // tuning knob
    if (dma_mode) {
        if (wide) {
            static const int nbuf = getenv("BGAMD_TP_NBUF") ? atoi(getenv("BGAMD_TP_NBUF")) : 3;
            const size_t sh = (size_t)nbuf * (256 + 256) * 64;  // 96 / 128 KiB
            static bool once = false;
            if (!once) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_dma_kernel<T, 64, 3, 256, 256>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 512 * 64);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_dma_kernel<T, 64, 4, 256, 256>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 512 * 64);
                once = true;
            }
            if (nbuf == 3) hipLaunchKernelGGL((gemm_conv_dma_kernel<T, 64, 3, 256, 256>), dim3((unsigned)nblk), dim3(1024), sh, st, P);
            else hipLaunchKernelGGL((gemm_conv_dma_kernel<T, 64, 4, 256, 256>), dim3((unsigned)nblk), dim3(1024), sh, st, P);
        } else if (flat) {
            const size_t sh = 3 * (TILE + 256) * 64;  // 72 KiB
            static bool once = false;
            if (!once) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_dma_kernel<T, 64, 3, TILE, 256>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
                once = true;
            }
            hipLaunchKernelGGL((gemm_conv_dma_kernel<T, 64, 3, TILE, 256>), dim3((unsigned)nblk), dim3(512), sh, st, P);
        } else if (tall) {
            const size_t sh = 3 * (256 + TILE) * 64;  // 72 KiB
            static bool once = false;
            if (!once) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_dma_kernel<T, 64, 3, 256>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
                once = true;
            }
            hipLaunchKernelGGL((gemm_conv_dma_kernel<T, 64, 3, 256>), dim3((unsigned)nblk), dim3(512), sh, st, P);
        } else if (use128 && dma_mode == 2) {
            const size_t sh = 3 * 2 * TILE * 128;  // 96 KiB
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_dma_kernel<T, 128, 3, TILE>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            hipLaunchKernelGGL((gemm_conv_dma_kernel<T, 128, 3, TILE>), dim3((unsigned)nblk), dim3(NTHREADS), sh, st, P);
        } else if (dma_mode == 3) {
            const size_t sh = 4 * 2 * TILE * 64;  // 64 KiB
            hipLaunchKernelGGL((gemm_conv_dma_kernel<T, 64, 4, TILE>), dim3((unsigned)nblk), dim3(NTHREADS), sh, st, P);
        } else {
            const size_t sh = 3 * 2 * TILE * 64;  // 48 KiB
            hipLaunchKernelGGL((gemm_conv_dma_kernel<T, 64, 3, TILE>), dim3((unsigned)nblk), dim3(NTHREADS), sh, st, P);
        }
        BG_CHECK_LAUNCH("gemm_conv_dma_kernel");
        return BG_OK;
    }
    if (use128) {
        const size_t sh = 2 * 2 * TILE * 128;
        hipLaunchKernelGGL((gemm_conv_kernel<T, 128>), dim3((unsigned)nblk), dim3(NTHREADS), sh, st, P);
    } else {
        const size_t sh = 2 * 2 * TILE * 64;
        hipLaunchKernelGGL((gemm_conv_kernel<T, 64>), dim3((unsigned)nblk), dim3(NTHREADS), sh, st, P);
    }
    BG_CHECK_LAUNCH("gemm_conv_kernel");
    return BG_OK;
}

}  // namespace

extern "C" int bg_conv2d_fwd(const bg_conv_desc* d, const void* x, const void* w, const float* bias, void* y,
                             void* stream) {
    int rc = check_conv_desc(d, "bg_conv2d_fwd");
    if (rc) return rc;
    BG_CHECK_ARG(x && w && y && aligned16(x) && aligned16(w) && aligned16(y), "bg_conv2d_fwd: null/unaligned pointer");
    GemmConvParams P{};
    P.in = x; P.w = w; P.out = y; P.bias = bias;
    P.N = d->N; P.IH = d->H; P.IW = d->W; P.OH = d->Ho; P.OW = d->Wo;
    P.CK = d->Cin; P.NO = d->Cout; P.ldi = d->ldx; P.ldo = d->ldy;
    P.KH = d->KH; P.KW = d->KW; P.stride = d->stride; P.pad = d->pad; P.dil = d->dil;
    P.transposed = 0;
    P.M = (long long)d->N * d->Ho * d->Wo;
    if (d->dtype == BG_BF16) return launch_gemm_conv<bf16_t>(P, (hipStream_t)stream);
    return launch_gemm_conv<float>(P, (hipStream_t)stream);
}

extern "C" int bg_conv2d_fwd_splitk(const bg_conv_desc* d, const void* x, const void* w, float* ws, int32_t splits,
                                    void* stream) {
    int rc = check_conv_desc(d, "bg_conv2d_fwd_splitk");
    if (rc) return rc;
    BG_CHECK_ARG(x && w && ws && aligned16(x) && aligned16(w) && splits >= 2, "bg_conv2d_fwd_splitk: bad args");
    GemmConvParams P{};
    P.in = x; P.w = w; P.out = nullptr; P.bias = nullptr; P.ws = ws;
    P.N = d->N; P.IH = d->H; P.IW = d->W; P.OH = d->Ho; P.OW = d->Wo;
    P.CK = d->Cin; P.NO = d->Cout; P.ldi = d->ldx; P.ldo = d->ldy;
    P.KH = d->KH; P.KW = d->KW; P.stride = d->stride; P.pad = d->pad; P.dil = d->dil;
    P.transposed = 0;
    P.M = (long long)d->N * d->Ho * d->Wo;
    if (d->dtype == BG_BF16) return launch_gemm_conv<bf16_t>(P, (hipStream_t)stream, splits);
    return launch_gemm_conv<float>(P, (hipStream_t)stream, splits);
}

extern "C" int bg_conv2d_bwd_data_splitk(const bg_conv_desc* d, const void* dy, const void* wt, float* ws, int32_t splits,
                                         void* stream) {
    int rc = check_conv_desc(d, "bg_conv2d_bwd_data_splitk");
    if (rc) return rc;
    BG_CHECK_ARG(dy && wt && ws && aligned16(dy) && aligned16(wt) && splits >= 2, "bg_conv2d_bwd_data_splitk: bad args");
    GemmConvParams P{};
    P.in = dy; P.w = wt; P.out = nullptr; P.bias = nullptr; P.ws = ws;
    P.N = d->N; P.IH = d->Ho; P.IW = d->Wo; P.OH = d->H; P.OW = d->W;
    P.CK = d->Cout; P.NO = d->Cin; P.ldi = d->ldy; P.ldo = d->ldx;
    P.KH = d->KH; P.KW = d->KW; P.stride = d->stride; P.pad = d->pad; P.dil = d->dil;
    P.transposed = 1;
    P.M = (long long)d->N * d->H * d->W;
    if (d->dtype == BG_BF16) return launch_gemm_conv<bf16_t>(P, (hipStream_t)stream, splits);
    return launch_gemm_conv<float>(P, (hipStream_t)stream, splits);
}

extern "C" int bg_splitk_reduce(int32_t dtype, const float* ws, int32_t splits, int64_t rows, int32_t C, void* y, int32_t ldy,
                               void* stream) {
    BG_CHECK_ARG(dtype_ok(dtype) && ws && y && splits >= 1 && rows > 0 && C > 0 && C % 4 == 0 && ldy >= C && aligned16(ws),
                 "bg_splitk_reduce: bad args");
    long long g = (rows * (C / 4) + 255) / 256;
    if (g > 65535) g = 65535;
    BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, ws,
                                                   splits, (long long)rows, C, (T*)y, ldy));
    BG_CHECK_LAUNCH("splitk_reduce_kernel");
    return BG_OK;
}

extern "C" int bg_conv2d_fwd_stats(const bg_conv_desc* d, const void* x, const void* w, void* y, double* sum,
                                   double* sumsq, int32_t groups, void* stream) {
    int rc = check_conv_desc(d, "bg_conv2d_fwd_stats");
    if (rc) return rc;
    BG_CHECK_ARG(x && w && y && sum && sumsq && aligned16(x) && aligned16(w) && aligned16(y),
                 "bg_conv2d_fwd_stats: null/unaligned pointer");
    const long long M_ = (long long)d->N * d->Ho * d->Wo;
    BG_CHECK_ARG(groups >= 1 && M_ % groups == 0 && (groups == 1 || (M_ / groups) % TILE == 0),
                 "bg_conv2d_fwd_stats: the pixels of a statistic group must be a multiple of %d (or groups == 1)", TILE);
    GemmConvParams P{};
    P.in = x; P.w = w; P.out = y; P.bias = nullptr;
    P.N = d->N; P.IH = d->H; P.IW = d->W; P.OH = d->Ho; P.OW = d->Wo;
    P.CK = d->Cin; P.NO = d->Cout; P.ldi = d->ldx; P.ldo = d->ldy;
    P.KH = d->KH; P.KW = d->KW; P.stride = d->stride; P.pad = d->pad; P.dil = d->dil;
    P.transposed = 0;
    P.M = (long long)d->N * d->Ho * d->Wo;
    P.stat_sum = sum; P.stat_sq = sumsq;
    P.stat_group_pix = groups > 1 ? (int)(M_ / groups) : 0;
    if (d->dtype == BG_BF16) return launch_gemm_conv<bf16_t>(P, (hipStream_t)stream);
    return launch_gemm_conv<float>(P, (hipStream_t)stream);
}

extern "C" int bg_conv2d_bwd_data(const bg_conv_desc* d, const void* dy, const void* wt, void* dx, void* stream) {
    int rc = check_conv_desc(d, "bg_conv2d_bwd_data");
    if (rc) return rc;
    BG_CHECK_ARG(dy && wt && dx && aligned16(dy) && aligned16(wt) && aligned16(dx),
                 "bg_conv2d_bwd_data: null/unaligned pointer");
    static const bool smallc = !getenv("BGAMD_NO_SMALLC");
    if (smallc && d->dtype == BG_BF16 && d->Cin <= 16 && d->Cin % 4 == 0 && d->KH == 3 && d->KW == 3 && d->stride == 2 &&
        d->pad == 1 && d->dil == 1 && (d->Cout == 128 || d->Cout == 64) && d->ldx % 4 == 0) {
        SmallCDgradParams Q{(const bf16_t*)dy, (const bf16_t*)wt, (bf16_t*)dx, d->N, d->H, d->W, d->Ho, d->Wo, d->Cout,
                            pad_k(d->Cout, BG_BF16), d->ldy, d->ldx, d->Cin, (d->W + 31) / 32, 0};
        Q.tiles = (long long)d->N * ((d->H + 1) / 2) * Q.tiles_w;     // per parity class (the even rows: the larger count)
        const long long blocks = std::min<long long>(Q.tiles, 256 * 8);
        if (d->Cout == 128) hipLaunchKernelGGL(dgrad_s2_smallc_kernel<128>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, Q);
        else hipLaunchKernelGGL(dgrad_s2_smallc_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, Q);
        BG_CHECK_LAUNCH("dgrad_s2_smallc_kernel");
        return BG_OK;
    }
    GemmConvParams P{};
    P.in = dy; P.w = wt; P.out = dx; P.bias = nullptr;
    P.N = d->N; P.IH = d->Ho; P.IW = d->Wo; P.OH = d->H; P.OW = d->W;
    P.CK = d->Cout; P.NO = d->Cin; P.ldi = d->ldy; P.ldo = d->ldx;
    P.KH = d->KH; P.KW = d->KW; P.stride = d->stride; P.pad = d->pad; P.dil = d->dil;
    P.transposed = 1;
    P.M = (long long)d->N * d->H * d->W;
    if (d->dtype == BG_BF16) return launch_gemm_conv<bf16_t>(P, (hipStream_t)stream);
    return launch_gemm_conv<float>(P, (hipStream_t)stream);
}

static int wgrad_plan(const bg_conv_desc* d, WgradParams& P, long long& tiles, long long& splits, const char* who) {
    P.N = d->N; P.H = d->H; P.W = d->W; P.Ho = d->Ho; P.Wo = d->Wo;
    P.Ci = d->Cin; P.Co = d->Cout; P.ldx = d->ldx; P.ldy = d->ldy;
    P.KH = d->KH; P.KW = d->KW; P.stride = d->stride; P.pad = d->pad; P.dil = d->dil;
    P.M = (long long)d->N * d->Ho * d->Wo;
    P.tiles_co = (d->Cout + TILE - 1) / TILE;
    P.tiles_ci = (d->Cin + TILE - 1) / TILE;
    static const bool fold_ok = !getenv("BGAMD_NO_WGRAD_FOLD");
    const int vec_w = 16 / (int)dtype_size(d->dtype);
    P.fold = fold_ok && d->KH * d->KW > 1 && d->Cin <= 32 && d->Cin % vec_w == 0;
    if (P.fold) P.tiles_ci = (d->KH * d->KW * d->Cin + TILE - 1) / TILE;
    tiles = P.fold ? (long long)P.tiles_co * P.tiles_ci : (long long)P.tiles_co * P.tiles_ci * d->KH * d->KW;
    // Pixel splits: two 256-thread workgroups fit a CU (VGPRs), so one full wave of the chip
    // is 512 workgroups.  Fill about one wave: more splits only add a whole copy of dW each
    // (float atomics, or a workspace slice), fewer leave CUs idle.
    static const long long target_wgs = getenv("BGAMD_WGRAD_TARGET") ? atoll(getenv("BGAMD_WGRAD_TARGET")) : 512;
    splits = tiles < 256 ? target_wgs / tiles : (2 * target_wgs + tiles - 1) / tiles;  // many tiles: balance the tail
    const long long max_splits = (P.M + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    P.pix_per_split = ((P.M + splits - 1) / splits + WG_PIX - 1) / WG_PIX * WG_PIX;
    splits = (P.M + P.pix_per_split - 1) / P.pix_per_split;     // every split owns pixels
    BG_CHECK_ARG(tiles * splits <= 0x7fffffffLL, "%s: grid too large", who);
    const long long es = dtype_size(d->dtype);
    const long long xb = (((long long)d->N * d->H * d->W - 1) * d->ldx + d->Cin) * es;
    const long long yb = ((P.M - 1) * d->ldy + d->Cout) * es;
    BG_CHECK_ARG(xb < (1LL << 31) && yb < (1LL << 31), "%s: operand larger than 2 GiB", who);
    P.x_bytes = (int)xb;
    P.dy_bytes = (int)yb;
    P.splits = (int)splits;
    return BG_OK;
}

static int wgrad_launch(const bg_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias, float* ws, long long ws_bytes,
                        void* stream, const char* who) {
    int rc = check_conv_desc(d, who);
    if (rc) return rc;
    BG_CHECK_ARG(x && dy && dw && aligned16(x) && aligned16(dy), "%s: null/unaligned pointer", who);
    WgradParams P{};
    P.x = x; P.dy = dy; P.dw = dw;
    long long tiles, splits;
    rc = wgrad_plan(d, P, tiles, splits, who);
    if (rc) return rc;
    if (ws) {
        BG_CHECK_ARG(aligned16(ws) && ws_bytes >= tiles * splits * (long long)(TILE * TILE * 4), "%s: workspace of %lld bytes, %lld needed",
                     who, ws_bytes, tiles * splits * (long long)(TILE * TILE * 4));
        P.ws = ws;
    }
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == BG_BF16) {
        const size_t sh = 2 * 2 * WG_PIX * (TILE * 2 + 64);
        hipLaunchKernelGGL((wgrad_kernel<bf16_t>), dim3((unsigned)(tiles * splits)), dim3(NTHREADS), sh, st, P);
    } else {
        const size_t sh = 2 * 2 * WG_PIX * (TILE * 4 + 64);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipLaunchKernelGGL((wgrad_kernel<float>), dim3((unsigned)(tiles * splits)), dim3(NTHREADS), sh, st, P);
    }
    BG_CHECK_LAUNCH("wgrad_kernel");
    if (P.ws) {
        const long long total = tiles * TILE * TILE;
        hipLaunchKernelGGL(wgrad_ws_reduce_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, st, P);
        BG_CHECK_LAUNCH("wgrad_ws_reduce_kernel");
    }
    if (dbias) return bg_colsum(d->dtype, dy, d->ldy, P.M, d->Cout, 1, 1.0f, dbias, stream);
    return BG_OK;
}

extern "C" int bg_conv2d_bwd_weight(const bg_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                                    void* stream) {
    return wgrad_launch(d, x, dy, dw, dbias, nullptr, 0, stream, "bg_conv2d_bwd_weight");
}

extern "C" int bg_conv2d_bwd_weight_ws_bytes(const bg_conv_desc* d, int64_t* bytes) {
    int rc = check_conv_desc(d, "bg_conv2d_bwd_weight_ws_bytes");
    if (rc) return rc;
    BG_CHECK_ARG(bytes != nullptr, "bg_conv2d_bwd_weight_ws_bytes: null result pointer");
    WgradParams P{};
    long long tiles, splits;
    rc = wgrad_plan(d, P, tiles, splits, "bg_conv2d_bwd_weight_ws_bytes");
    if (rc) return rc;
    *bytes = tiles * splits * (long long)(TILE * TILE * 4);
    return BG_OK;
}

extern "C" int bg_conv2d_bwd_weight_ws(const bg_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias, float* ws,
                                       int64_t ws_bytes, void* stream) {
    BG_CHECK_ARG(ws != nullptr, "bg_conv2d_bwd_weight_ws: null workspace");
    return wgrad_launch(d, x, dy, dw, dbias, ws, ws_bytes, stream, "bg_conv2d_bwd_weight_ws");
}

#ifdef BG_STAMPS
extern "C" int bg_conv_debug_stamps(void* buf) { g_dbg_stamps = (unsigned long long*)buf; return BG_OK; }
#endif

extern "C" int bg_conv_set_variant(int32_t variant) {
    BG_CHECK_ARG(variant == -1 || variant == 0 || variant == 2, "bg_conv_set_variant: %d is not one of -1, 0, 2", variant);
    g_conv_variant = variant;
    return BG_OK;
}

// Geometry of a grouped weight-gradient launch: RS = 1 is the pointwise case (M pixels, no image structure).
struct WggGeom { long long M; int N, H, W, KH, KW, dil, Cin, Cout, ldx, ldy; };

static int launch_wgrad_gang(const char* who, const int64_t* tbl, int n_layers, const WggGeom& G, void* stream) {
    const int RS = G.KH * G.KW;
    const bool taps = RS > 1;
    static const int nbuf = getenv("BGAMD_WGG_NBUF") ? atoi(getenv("BGAMD_WGG_NBUF")) : 4;   // 4 or 5 stages of 32 KiB
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_gang_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WGG_STAGE);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_gang_kernel<5, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * WGG_STAGE);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_gang_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WGG_STAGE);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_gang_kernel<5, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * WGG_STAGE);
        once = true;
    }
    const int tiles_ci = (G.Cin + 255) / 256, tpt = tiles_ci * ((G.Cout + 255) / 256);
    BG_CHECK_ARG(tpt <= 256, "%s: more than 256 output tiles per layer", who);
    // few tiles: the taps are members of the gang; otherwise every (layer, tap) pair is a layer of the launch
    const bool taps_in_members = taps && RS * tpt <= 32;
    const int per_layer = taps && !taps_in_members ? RS : 1;             // launch layers per caller layer
    const int chunk = WGG_MAX_LAYERS / per_layer;                          // caller layers per launch
    for (int l0 = 0; l0 < n_layers; l0 += chunk) {     // the addresses travel in the kernel arguments: 64 layers per launch
        WgGangParams P{};
        const int nl = std::min(chunk, n_layers - l0);
        P.L = nl * per_layer;
        for (int l = 0; l < nl; ++l) {
            for (int k = 0; k < 3; ++k)
                BG_CHECK_ARG(tbl[(long long)(l0 + l) * 4 + k] != 0 && (tbl[(long long)(l0 + l) * 4 + k] & 15) == 0,
                             "%s: null/unaligned pointer in row %d", who, l0 + l);
            for (int t = 0; t < per_layer; ++t) {
                for (int k = 0; k < 3; ++k) P.tbl[l * per_layer + t][k] = tbl[(long long)(l0 + l) * 4 + k];
                P.tbl[l * per_layer + t][3] = t;
            }
        }
        P.M = (int)G.M; P.ldx = G.ldx; P.ldy = G.ldy; P.Ci = G.Cin; P.Co = G.Cout;
        P.N = G.N; P.H = G.H; P.W = G.W; P.KH = G.KH; P.KW = G.KW; P.dil = G.dil; P.RS = RS;
        P.tiles_ci = tiles_ci;
        P.tpt = tpt;
        P.taps_in_members = taps_in_members;
        P.tiles = taps_in_members ? RS * tpt : tpt;     // members of a gang
        P.KS = (int)((G.M + WGG_PIX - 1) / WGG_PIX);
        const long long total = (long long)P.L * P.KS;
        // Placement.  T <= 32 members: XCD-local gangs (struct comment) -- g = 32 / T whole gangs per XCD plus gangs made
        // of the left-over CUs of q neighbouring XCDs; otherwise gangs spread in blockIdx order.
        static const bool spread_only = getenv("BGAMD_WGG_SPREAD") != nullptr;
        const int T = P.tiles;
        long long gangs;
        int gpx = 0, q = 0;
        P.use_map = !spread_only && T > 1 && T <= 32;
        if (P.use_map) {
            gpx = 32 / T;
            const int left = 32 - gpx * T;
            if (left > 0) {
                q = 1;
                while (q * left < T) q *= 2;          // q in {1, 2, 4, 8, ...}: XCDs pooled per straddling gang
                if (q > 8) q = 0;
            }
            gangs = 8 * gpx + (q ? 8 / q : 0);
        } else
            gangs = std::max(1, 256 / T);            // one workgroup per CU
        // A group of at least as many layers as gangs gives every gang a range of at least one layer: every dW tile then
        // receives at most two adds (order-independent: bit-reproducible).  Smaller groups keep all the gangs -- several
        // ranges per layer, their adds land in any order.  At least 8 K-steps per range.
        if (total / gangs < 8) gangs = std::max<long long>(1, total / 8);
        P.R = (int)((total + gangs - 1) / gangs);
        gangs = (total + P.R - 1) / P.R;
        P.gangs = (int)gangs;
        long long grid = gangs * T;
        if (P.use_map) {
            for (int b = 0; b < 256; ++b) P.map[b] = 0xffff;
            int gi = 0;
            // whole gangs: gang (x, k) takes slots k*T .. k*T + T - 1 of XCD x; gangs numbered XCD-major so that a short
            // group (fewer ranges than gangs) still spreads over all XCDs
            for (int k = 0; k < gpx; ++k)
                for (int x = 0; x < 8; ++x, ++gi)
                    for (int m = 0; m < T; ++m) P.map[(k * T + m) * 8 + x] = (unsigned short)(gi << 8 | m);
            // straddling gangs: consecutive members (row-major tiles: neighbours share their dy rows) fill the left-over
            // slots of XCDs c*q .. c*q + q - 1 in order
            if (q)
                for (int c = 0; c < 8 / q; ++c, ++gi) {
                    int m = 0;
                    for (int x = c * q; x < (c + 1) * q && m < T; ++x)
                        for (int sl = gpx * T; sl < 32 && m < T; ++sl, ++m) P.map[sl * 8 + x] = (unsigned short)(gi << 8 | m);
                }
            grid = 256;
        }
        const dim3 gr((unsigned)grid), bl(512);
        hipStream_t st = (hipStream_t)stream;
        if (taps) {
            if (nbuf == 4) hipLaunchKernelGGL((wgrad_gang_kernel<4, true>), gr, bl, 4 * WGG_STAGE, st, P);
            else hipLaunchKernelGGL((wgrad_gang_kernel<5, true>), gr, bl, 5 * WGG_STAGE, st, P);
        } else {
            if (nbuf == 4) hipLaunchKernelGGL((wgrad_gang_kernel<4, false>), gr, bl, 4 * WGG_STAGE, st, P);
            else hipLaunchKernelGGL((wgrad_gang_kernel<5, false>), gr, bl, 5 * WGG_STAGE, st, P);
        }
        BG_CHECK_LAUNCH("wgrad_gang_kernel");
    }
    return BG_OK;
}

extern "C" int bg_conv2d_bwd_weight_grouped(int32_t dtype, const int64_t* tbl, int32_t n_layers, int64_t M, int32_t Cin,
                                           int32_t Cout, int32_t ldx, int32_t ldy, void* stream) {
    BG_CHECK_ARG(dtype == BG_BF16, "bg_conv2d_bwd_weight_grouped: bf16 operands only (the transposing LDS reads)");
    BG_CHECK_ARG(tbl && n_layers >= 1 && M >= 1 && M < (1LL << 31) && Cin >= 8 && Cout >= 8 && Cin % 8 == 0 && Cout % 8 == 0 &&
                 ldx >= Cin && ldy >= Cout && ldx % 8 == 0 && ldy % 8 == 0, "bg_conv2d_bwd_weight_grouped: bad arguments");
    // the kernel addresses a layer's operands with 32-bit byte offsets into per-layer descriptors
    BG_CHECK_ARG((M - 1) * (long long)ldx * 2 + (long long)Cin * 2 < (1LL << 31) && (M - 1) * (long long)ldy * 2 + (long long)Cout * 2 < (1LL << 31),
                 "bg_conv2d_bwd_weight_grouped: an operand beyond 2 GiB (M=%lld ldx=%d ldy=%d)", (long long)M, ldx, ldy);
    const WggGeom G{M, 1, 1, (int)M, 1, 1, 1, Cin, Cout, ldx, ldy};
    return launch_wgrad_gang("bg_conv2d_bwd_weight_grouped", tbl, n_layers, G, stream);
}

// Weight gradients of n_layers k x k convolutions of ONE geometry (stride 1, odd kernel, "same" padding pad = dil * (k - 1) / 2:
// the 3x3 convolutions of the decoder and the ASPP) through the gang kernel: tbl rows = (x, dy, dw, 0) as above, dw
// is [Cout][KH][KW][Cin] fp32, accumulated into.
extern "C" int bg_conv2d_bwd_weight_grouped_taps(const bg_conv_desc* d, const int64_t* tbl, int32_t n_layers, void* stream) {
    BG_CHECK_ARG(d && tbl && n_layers >= 1, "bg_conv2d_bwd_weight_grouped_taps: null argument");
    BG_CHECK_ARG(d->dtype == BG_BF16, "bg_conv2d_bwd_weight_grouped_taps: bf16 operands only (the transposing LDS reads)");
    BG_CHECK_ARG(d->KH == d->KW && (d->KH & 1) && d->KH >= 1 && d->KH <= 5 && d->stride == 1 && d->dil >= 1 &&
                 d->pad == d->dil * (d->KH - 1) / 2 && d->Ho == d->H && d->Wo == d->W,
                 "bg_conv2d_bwd_weight_grouped_taps: stride-1 odd-kernel 'same' convolutions only (k=%d stride=%d pad=%d dil=%d)",
                 d->KH, d->stride, d->pad, d->dil);
    const long long M = (long long)d->N * d->H * d->W;
    BG_CHECK_ARG(M >= 1 && d->Cin >= 8 && d->Cout >= 8 && d->Cin % 8 == 0 && d->Cout % 8 == 0 && d->ldx >= d->Cin && d->ldy >= d->Cout &&
                 d->ldx % 8 == 0 && d->ldy % 8 == 0, "bg_conv2d_bwd_weight_grouped_taps: bad channel counts / pitches");
    // 32-bit byte offsets, with room for the largest tap shift
    const long long reach = (long long)d->dil * (d->KH / 2) * ((long long)d->W + 1);
    BG_CHECK_ARG((M + reach) * (long long)d->ldx * 2 < (1LL << 31) && M * (long long)d->ldy * 2 < (1LL << 31) &&
                 (long long)d->Cout * d->KH * d->KW * d->Cin * 4 < (1LL << 31),
                 "bg_conv2d_bwd_weight_grouped_taps: an operand beyond 2 GiB");
    const WggGeom G{M, d->N, d->H, d->W, d->KH, d->KW, d->dil, d->Cin, d->Cout, d->ldx, d->ldy};
    return launch_wgrad_gang("bg_conv2d_bwd_weight_grouped_taps", tbl, n_layers, G, stream);
}

extern "C" int bg_conv_weight_kpad(int32_t dtype) { return dtype_ok(dtype) ? kpad_of(dtype) : BG_E_ARG; }

extern "C" int bg_pack_conv_weights(int32_t dtype, const void* src, void* dst_krsc, void* dst_crsk, const int64_t* tbl,
                                    int32_t n_layers, int64_t max_elems, void* stream) {
    BG_CHECK_ARG(dtype_ok(dtype) && src && dst_krsc && dst_crsk && tbl && n_layers > 0 && max_elems > 0,
                 "bg_pack_conv_weights: bad args");
    BG_CHECK_ARG(n_layers <= 65535, "bg_pack_conv_weights: too many layers");
    long long bx = (max_elems + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BG_BF16)
        hipLaunchKernelGGL((pack_conv_weights_kernel<bf16_t>), dim3((unsigned)bx, (unsigned)n_layers), dim3(256), 0, st,
                           (const bf16_t*)src, (bf16_t*)dst_krsc, (bf16_t*)dst_crsk, (const long long*)tbl);
    else
        hipLaunchKernelGGL((pack_conv_weights_kernel<float>), dim3((unsigned)bx, (unsigned)n_layers), dim3(256), 0, st,
                           (const float*)src, (float*)dst_krsc, (float*)dst_crsk, (const long long*)tbl);
    BG_CHECK_LAUNCH("pack_conv_weights_kernel");
    return BG_OK;
}
