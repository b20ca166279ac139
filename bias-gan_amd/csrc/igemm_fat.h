// The fat-tile implicit-GEMM kernel (one 512-thread workgroup per CU owning the CU's LDS), its epilogues, its plan and
// launcher, shared by the bf16 / fp32 entry points (igemm_conv.hip) and the fp8 ones (fp8_conv.hip).  Everything sits in
// an anonymous namespace: each translation unit instantiates only the variants it launches.
#pragma once
#include "common.h"
#include <algorithm>
#include <stdlib.h>
#include <type_traits>

namespace {

// fp8 operand elements (OCP e4m3 / e5m2, one byte).  The tag type of a kernel instantiation names the format of the
// ACTIVATION-side operand (forward: x in e4m3; data gradient: dy in e5m2 or e4m3); the weight side is always e4m3.
// Outputs and statistics of an fp8 launch are bf16 / fp64 like the bf16 kernels'.
struct f8e4_t { unsigned char v; };
struct f8e5_t { unsigned char v; };
template <typename T> struct OutOf { typedef T type; };
template <> struct OutOf<f8e4_t> { typedef bf16_t type; };
template <> struct OutOf<f8e5_t> { typedef bf16_t type; };
template <typename T> struct IsFp8 { static constexpr bool value = std::is_same<T, f8e4_t>::value || std::is_same<T, f8e5_t>::value; };
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int TILE = 128;     // both tile edges
constexpr int NTHREADS = 256;

struct GemmConvParams {
    const void* in;
    const void* w;
    void* out;
    const float* bias;
    int N, IH, IW, OH, OW;
    int CK;  // reduction channels per tap (physical, multiple of VEC)
    int NO;  // output channels (physical rows of w)
    int ldi, ldo;
    int KH, KW, stride, pad, dil;
    int transposed;
    long long M;  // N*OH*OW
    int tiles_c, tiles_p;
    double* stat_sum;  // optional per-output-channel sum / sum of squares of the STORED outputs
    double* stat_sq;   // (BatchNorm statistics fused into the epilogue); NULL = off
    int stat_group_pix;  // pixels per statistic group (sum/sumsq are [groups][NO]; a multiple of the pixel tile); 0 = one group
    int CKp;       // K stride of one tap inside the (zero-padded) weight copy
    int in_bytes;  // exact extent of the activation operand (buffer-load range check)
    int w_bytes;
    float* ws;         // split-K: fp32 [splits][M][NO]; every split stores its partial tile into its own slice
    int kt_per_split;  // split-K: K-steps per split
    // fat-tile variant: pixel tiles of tn_valid (<= the template's padded width) rows, tiles_per_group of them per
    // statistic group of group_pix pixels (one group = the whole tensor unless stat_group_pix is set)
    int tn_valid, tiles_per_group, group_pix;
    // fp8 launches: power-of-two exponents the operands were quantised with (q = round(v * 2^e)), one int each in device
    // memory (weights: per layer, written by the pack kernel; activations: per quantisation site, delayed scaling); the
    // MFMA's hardware block scales carry 2^-e, so the accumulators come out in real units
    const int* exp_w;
    const int* exp_act;
#ifdef BG_STAMPS   // diagnostic build only (scripts/stamps_fat.py): 8 time stamps per workgroup
    unsigned long long* dbg;
#endif
};
#ifdef BG_STAMPS
unsigned long long* g_dbg_stamps = nullptr;
#define BG_STAMP(i) do { if (P.dbg && threadIdx.x == 0) P.dbg[(long long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define BG_STAMP_CYC(i) do { if (P.dbg && threadIdx.x == 0) P.dbg[(long long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BG_STAMP(i)
#define BG_STAMP_CYC(i)
#endif

template <int BKB>
__device__ __forceinline__ int lds_off(int row, int chunk) {
    if constexpr (BKB == 64) {
        return row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4);
    } else {
        return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    }
}

// Sum over the 16 lanes of a DPP row (lanes sharing lane>>4), result in every lane of the row: four
// VALU adds with a row-rotate modifier instead of four ds_bpermute round trips through the LDS.
__device__ __forceinline__ float row16_sum(float v) {
#define BG_ROR(x, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + (n), 0xf, 0xf, false))
    v += BG_ROR(v, 8);
    v += BG_ROR(v, 4);
    v += BG_ROR(v, 2);
    v += BG_ROR(v, 1);
#undef BG_ROR
    return v;
}

// The same reduction for N values at once with the rotate folded into the add (v_add_f32_dpp: the compiler emits
// v_mov_b32_dpp + v_add_f32 for row16_sum, 8 instructions per value instead of 4).  Each step runs over all N values
// before the next step starts, so a value's dependent DPP read is N instructions behind its write (a DPP source written
// by the preceding VALU instructions needs two wait states that nothing inserts inside inline assembly): N >= 4.
template <int N>
__device__ __forceinline__ void row16_sum_all(float (&v)[N]) {
    static_assert(N >= 4, "dependent DPP reads must be at least three instructions apart");
#define BG_ROR_ADD(n)                                                                                              \
    _Pragma("unroll") for (int k = 0; k < N; ++k)                                                                  \
        asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:" #n " row_mask:0xf bank_mask:0xf" : "=v"(v[k]) : "v"(v[k]));
    asm volatile("s_nop 4");   // whatever wrote v[] or EXEC just before: a DPP read needs up to five wait states after it
    BG_ROR_ADD(8)
    BG_ROR_ADD(4)
    BG_ROR_ADD(2)
    BG_ROR_ADD(1)
#undef BG_ROR_ADD
}

// Offsets into the operands are 32-bit byte offsets fed to buffer loads: the
// hardware range check (num_records = exact byte size of the tensor) returns
// zeros for any offset outside, so padding taps, rows beyond M and weight rows
// beyond Cout simply carry the marker OOB -- no per-lane predication, no zero
// fill code, and no 64-bit address arithmetic in the K loop.
constexpr int OOB = (int)0x80000000;

// ------------------------------------------------------- LDS-DMA variant ----
// Same tiling, swizzle, MFMA schedule and epilogue, but the operands travel
// global -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction,
// out-of-range lanes deliver zeros) into a ring of NBUF stages that runs two
// K-steps ahead of the MFMAs: no staging VGPRs, no ds_write pass, one raw
// s_barrier per K-step with the loads kept in flight across it behind a COUNTED
// s_waitcnt vmcnt.  The LDS image of a DMA is lane-linear, so the XOR swizzle is
// applied to the SOURCE chunk each lane fetches.
// (non-template on purpose: hipcc's host pass rejects this target builtin inside a
// dependent context with a silent substitution failure)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_dst, 16, voffset, 0, 0, 0);
}

#ifdef ABL_FAT_PRODUCER
// Ablation (scripts/ablate_fat.sh, not shipped): a synthetic stand-in for an in-kernel depthwise PRODUCER of the B operand
// (the fused [BatchNorm + LeakyReLU -> depthwise -> pointwise] forward unit of DESIGN.md 7), to MEASURE how much of its
// instruction stream hides in the shadow of the MFMAs before the real kernel is written: per MFMA group ABL_FAT_PRODUCER
// vector instructions in the producer's mix (packed FMAs, bf16 unpacks, max, cvt) on eight scratch registers and
// ABL_FAT_PRODUCER_LDS 16-byte LDS reads of the stage being consumed.  Results are discarded.
#ifndef ABL_FAT_PRODUCER_LDS
#define ABL_FAT_PRODUCER_LDS 0
#endif
typedef __attribute__((ext_vector_type(2))) float abl_f32x2;
struct AblRegs { abl_f32x2 p0, p1, p2; unsigned u0, u1, u2; };
__device__ __forceinline__ void abl_init(AblRegs& r, int lane) {
    r.p0 = abl_f32x2{0.5f, 0.25f}; r.p1 = abl_f32x2{1e-3f, 2e-3f}; r.p2 = abl_f32x2{3e-3f, 1e-3f};
    r.u0 = 0x3f803f00u + lane; r.u1 = r.u2 = 0;
}
__device__ __forceinline__ void abl_producer(AblRegs& r, const char* lds) {
#pragma unroll
    for (int i = 0; i < ABL_FAT_PRODUCER_LDS; ++i) {
        i32x4 t;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"((int)(size_t)lds), "n"(i * 1024));
    }
#pragma unroll
    for (int i = 0; i < ABL_FAT_PRODUCER / 8; ++i) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n\tv_and_b32 %4, 0xffff0000, %3\n\tv_pk_fma_f32 %1, %2, %0, %1\n\tv_lshlrev_b32 %5, 16, %3\n\t"
                     "v_pk_fma_f32 %2, %0, %1, %2\n\tv_max_f32 %3, %4, %5\n\tv_pk_mul_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %3, %4, %5"
                     : "+v"(r.p0), "+v"(r.p1), "+v"(r.p2), "+v"(r.u0), "+v"(r.u1), "+v"(r.u2));
    }
}
__device__ __forceinline__ void abl_sink(const AblRegs& r, double* where) {   // keeps the scratch registers alive; never true
    if (r.p0[0] + r.p1[1] + r.p2[0] == 12345.678f && r.u0 == 7u) where[0] = 1.0;
}
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Workgroup barrier for data exchanged through the LDS only: waits for this wave's LDS traffic, not for its global
// stores (__syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: behind a tile's output stores it makes the
// workgroup wait until they have been acknowledged).  Used in the statistics exchange of the epilogues.  Measured: no
// change by itself on the HBM-bound 128 / 256-channel layers, whose statistics epilogue costs 18-30 us per launch over
// the plain forward (scripts/bench_fat.py, WITH_PLAIN_FWD=1).  The ablation build -DABL_STAT_NOATOM splits that cost:
// 8-18 us are the fp64 atomics (1 728 - 6 912 tiles adding into the same 2 x Cout addresses), 4-20 us the reduction's
// ~400 instructions and two barriers, during which a workgroup (two per CU) has no loads in flight.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Geometry of the epilogue's transposition buffer: one 16-pixel x (MI*16-channel) block per wave, UNPADDED rows of
// CHB bytes, bank-conflict free on both sides (scripts/lds_conflicts.py, the LDS model of MI355X_MICROARCH.md):
//   writer  lane (pixel r16, channel quad q) stores its 4 channels of fragment i (8 B bf16: ds_write_b64, serviced in
//           16-lane groups = 16 pixels at ONE channel offset; 16 B fp32: ds_write_b128, 8-lane groups);
//   reader  lane takes 16-byte chunk ci = t*64 + lane of the block, pixel-major (ds_read_b128).
// With padded rows (16 bytes, round 2) the bf16 writer hit every bank twice and the reader's lane groups wrapped onto
// each other (SQ_LDS_BANK_CONFLICT 9-21 % of the LDS cycles of every GEMM variant).  Here the chunk index inside a
// pixel's row is XORed with a function of the pixel -- pix & 7 where a row is a multiple of 8 chunks, (pix >> 1) & 3
// otherwise (the XOR then stays inside aligned groups of 4 chunks) -- and, for the 8-byte granules of bf16, the two
// halves of a chunk are exchanged for pixels 8-15 (the reader swaps them back): 16 pixels -> 16 distinct bank pairs.
template <int CPP, int ES>
struct EpiSwz {
    static constexpr bool WIDE = CPP % 8 == 0;
    static_assert(CPP % 4 == 0, "the XOR must stay inside a pixel's row: rows of whole 4-chunk groups");
    __device__ static __forceinline__ int f(int pix) { return WIDE ? (pix & 7) : ((pix >> 1) & 3); }
    __device__ static __forceinline__ bool flip(int pix) { return ES == 2 && (pix & 8); }
    // byte offset inside the region of (pixel, byte b of its row): b is a multiple of the writer's granule
    __device__ static __forceinline__ int wr(int pix, int b) {
        const int gr = ES == 2 ? 8 : 16;
        int within = b & 15;
        if (ES == 2 && flip(pix)) within ^= gr;
        return pix * (CPP * 16) + (((b >> 4) ^ f(pix)) << 4) + within;
    }
    __device__ static __forceinline__ int rd(int pix, int ch) { return pix * (CPP * 16) + ((ch ^ f(pix)) << 4); }
};

template <typename T, int MI, int NJ, int WM, int WN>
__device__ __forceinline__ void conv_epilogue_fat(const GemmConvParams& P, f32x4 (&acc)[MI][NJ], int p_base, int rows_valid,
                                                  int grp, int c_base, int wave_c, int wave_p, int lane, char* smem) {
    constexpr int TM = WM * MI * 16, ES = (int)sizeof(T);
    constexpr int CHB = MI * 16 * ES;      // bytes of one pixel's channels in this wave's sub-tile
    constexpr int ROWB = CHB;              // unpadded LDS row, swizzled (EpiSwz)
    constexpr int CPP = CHB / 16;          // 16-byte chunks per pixel
    constexpr int NST = 16 * CPP / 64;     // store instructions per 16-pixel block
    constexpr int REGION = 16 * ROWB;
    static_assert((16 * CPP) % 64 == 0, "a 16-pixel block must be whole wave instructions");
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    const int r16 = lane & 15, q = lane >> 4;
    const int wave = wave_c * WN + wave_p;
    T* out_tile = reinterpret_cast<T*>(P.out) + (long long)p_base * P.ldo;
    const long long rem = (((long long)P.M - p_base - 1) * P.ldo + P.NO) * ES;
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(out_tile, 0, bg_records(rem), 0x00020000);
    const bool stats = P.stat_sum != nullptr;
    char* region = smem + wave * REGION;                               // this wave's transposition buffer
    float* red = reinterpret_cast<float*>(smem + WM * WN * REGION);   // [wave_p][TM channels][2]
    static_assert(WM * WN * REGION + WN * TM * 8 <= 96 * 1024, "epilogue scratch must fit the smallest staging ring that uses it");
    __syncthreads();   // every wave is done reading the staging ring: reuse it

    // read-back geometry: chunk ci of the block's 16 * CPP is (pixel ci / CPP, 16-byte chunk ci % CPP)
    int rd_off[NST], st_coff[NST], st_pix[NST];
    bool rd_flip[NST];
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        const int ci = t * 64 + lane;
        const int pix = ci / CPP, ch = ci - pix * CPP;
        rd_off[t] = EpiSwz<CPP, ES>::rd(pix, ch);
        rd_flip[t] = EpiSwz<CPP, ES>::flip(pix);
        const int co = c_base + wave_c * MI * 16 + ch * (16 / ES);
        st_coff[t] = co < P.NO ? co * ES : OOB;      // Cout is a multiple of the 16-byte vector
        st_pix[t] = wave_p * NJ * 16 + pix;
    }
    f32x4 bv[MI];
    if (P.bias) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int co = c_base + wave_c * MI * 16 + i * 16 + q * 4;
            bv[i] = co < P.NO ? *reinterpret_cast<const f32x4*>(P.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    float s1[MI][4], s2[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[i][e] = s2[i][e] = 0.f;
    int wr_off[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) wr_off[i] = EpiSwz<CPP, ES>::wr(r16, i * 16 * ES + q * 4 * ES);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            f32x4 v = acc[i][j];
            if (P.bias) v += bv[i];
            if constexpr (sizeof(T) == 2) {
                const bf16x4 ov = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *reinterpret_cast<bf16x4*>(region + wr_off[i]) = ov;
                if (stats) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float r = (float)ov[e];
                        s1[i][e] += r;
                        s2[i][e] = fmaf(r, r, s2[i][e]);
                    }
                }
            } else {
                *reinterpret_cast<f32x4*>(region + wr_off[i]) = v;
                if (stats) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s1[i][e] += v[e];
                        s2[i][e] = fmaf(v[e], v[e], s2[i][e]);
                    }
                }
            }
        }
        // the region is private to the wave and a wave's LDS instructions execute in order: no barrier
#pragma unroll
        for (int t = 0; t < NST; ++t) {
            u32x4 w = *reinterpret_cast<const u32x4*>(region + rd_off[t]);
            if (ES == 2 && rd_flip[t]) w = u32x4{w[2], w[3], w[0], w[1]};
            const int row = st_pix[t] + j * 16;
            const int roff = row < rows_valid ? row * P.ldo * ES : OOB;
            const int off = (roff | st_coff[t]) < 0 ? OOB : roff + st_coff[t];
            __builtin_amdgcn_raw_buffer_store_b128(w, rs_out, off, 0, 0);
        }
    }
    if (stats) {  // wave-uniform
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[i][e] = row16_sum(s1[i][e]);
                s2[i][e] = row16_sum(s2[i][e]);
            }
        if (r16 == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int cl = wave_c * MI * 16 + i * 16 + q * 4 + e;
                    red[(wave_p * TM + cl) * 2 + 0] = s1[i][e];
                    red[(wave_p * TM + cl) * 2 + 1] = s2[i][e];
                }
        }
        lds_barrier();     // NOT __syncthreads(): its vmcnt(0) would wait for the tile's stores to be acknowledged
        const int t = threadIdx.x;
        if (t < TM && c_base + t < P.NO) {
            const long long o = (long long)grp * P.NO + c_base + t;
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int wp = 0; wp < WN; ++wp) {
                a1 += red[(wp * TM + t) * 2];
                a2 += red[(wp * TM + t) * 2 + 1];
            }
#ifdef ABL_STAT_NOATOM   // ablation (scripts/bench_fat.py): the whole reduction, no atomics (a never-taken store keeps it alive)
            if (a1 == 12345.678f && a2 == 0.f) P.stat_sum[o] = 1.0;
#else
            atomicAdd(P.stat_sum + o, (double)a1);
            atomicAdd(P.stat_sq + o, (double)a2);
#endif
        }
    }
}

template <typename T, int MI, int NJ, int WM, int WN, bool STATS>
__device__ __forceinline__ void conv_epilogue_fat_impl(const GemmConvParams& P, f32x4 (&acc)[MI][NJ], int p_base, int rows_valid,
                                                       int grp, int c_base, int wave_c, int wave_p, int lane, char* smem) {
    constexpr int TM = WM * MI * 16, ES = (int)sizeof(T);
    constexpr int CHB = MI * 16 * ES;      // bytes of one pixel's channels in this wave's sub-tile
    constexpr int ROWB = CHB;              // unpadded LDS row, swizzled (EpiSwz)
    constexpr int CPP = CHB / 16;          // 16-byte chunks per pixel
    constexpr int NST = 16 * CPP / 64;     // store instructions per 16-pixel block
    constexpr int REGION = 16 * ROWB;
    static_assert((16 * CPP) % 64 == 0, "a 16-pixel block must be whole wave instructions");
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    const int r16 = lane & 15, q = lane >> 4;
    const int wave = wave_c * WN + wave_p;
    T* out_tile = reinterpret_cast<T*>(P.out) + (long long)p_base * P.ldo;
    const long long rem = (((long long)P.M - p_base - 1) * P.ldo + P.NO) * ES;
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(out_tile, 0, bg_records(rem), 0x00020000);
    constexpr bool stats = STATS;   // compile-time: with run-time flags the per-value code carried 216 selects and 435 moves
    char* region = smem + wave * REGION;                               // this wave's transposition buffer
    float* red = reinterpret_cast<float*>(smem + WM * WN * REGION);   // [wave_p][TM channels][2]
    static_assert(WM * WN * REGION + WN * TM * 8 <= 96 * 1024, "epilogue scratch must fit the smallest staging ring that uses it");
    __syncthreads();   // every wave is done reading the staging ring: reuse it

    // read-back geometry: chunk ci of the block's 16 * CPP is (pixel ci / CPP, 16-byte chunk ci % CPP)
    int rd_off[NST], st_coff[NST], st_pix[NST];
    bool rd_flip[NST];
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        const int ci = t * 64 + lane;
        const int pix = ci / CPP, ch = ci - pix * CPP;
        rd_off[t] = EpiSwz<CPP, ES>::rd(pix, ch);
        rd_flip[t] = EpiSwz<CPP, ES>::flip(pix);
        const int co = c_base + wave_c * MI * 16 + ch * (16 / ES);
        st_coff[t] = co < P.NO ? co * ES : OOB;      // Cout is a multiple of the 16-byte vector
        st_pix[t] = wave_p * NJ * 16 + pix;
    }
    float s1[MI][4], s2[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[i][e] = s2[i][e] = 0.f;
    int wr_off[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) wr_off[i] = EpiSwz<CPP, ES>::wr(r16, i * 16 * ES + q * 4 * ES);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const f32x4 v = acc[i][j];
            if constexpr (sizeof(T) == 2) {
                const bf16x4 ov = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *reinterpret_cast<bf16x4*>(region + wr_off[i]) = ov;
                if (stats) {
                    // the stored values back as fp32 straight from the two packed dwords (one shift / one mask per
                    // value instead of a conversion and a shift), accumulated as pairs (v_pk_add_f32 / v_pk_fma_f32)
                    typedef __attribute__((ext_vector_type(2))) float f32x2;
                    const u32x2 pk = __builtin_bit_cast(u32x2, ov);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 r = {__uint_as_float(pk[h] << 16), __uint_as_float(pk[h] & 0xffff0000u)};
                        f32x2 a1 = {s1[i][2 * h], s1[i][2 * h + 1]}, a2 = {s2[i][2 * h], s2[i][2 * h + 1]};
                        a1 += r;
                        a2 = r * r + a2;
                        s1[i][2 * h] = a1[0]; s1[i][2 * h + 1] = a1[1];
                        s2[i][2 * h] = a2[0]; s2[i][2 * h + 1] = a2[1];
                    }
                }
            } else {
                *reinterpret_cast<f32x4*>(region + wr_off[i]) = v;
                if (stats) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s1[i][e] += v[e];
                        s2[i][e] = fmaf(v[e], v[e], s2[i][e]);
                    }
                }
            }
        }
        // the region is private to the wave and a wave's LDS instructions execute in order: no barrier
#pragma unroll
        for (int t = 0; t < NST; ++t) {
            u32x4 w = *reinterpret_cast<const u32x4*>(region + rd_off[t]);
            if (ES == 2 && rd_flip[t]) w = u32x4{w[2], w[3], w[0], w[1]};
            const int row = st_pix[t] + j * 16;
            const int roff = row < rows_valid ? row * P.ldo * ES : OOB;
            const int off = (roff | st_coff[t]) < 0 ? OOB : roff + st_coff[t];
            __builtin_amdgcn_raw_buffer_store_b128(w, rs_out, off, 0, 0);
        }
    }
    if (stats) {  // wave-uniform
        {
            float all[MI * 8];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    all[i * 8 + e] = s1[i][e];
                    all[i * 8 + 4 + e] = s2[i][e];
                }
            row16_sum_all(all);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1[i][e] = all[i * 8 + e];
                    s2[i][e] = all[i * 8 + 4 + e];
                }
        }
        if (r16 == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int cl = wave_c * MI * 16 + i * 16 + q * 4 + e;
                    red[(wave_p * TM + cl) * 2 + 0] = s1[i][e];
                    red[(wave_p * TM + cl) * 2 + 1] = s2[i][e];
                }
        }
        lds_barrier();     // NOT __syncthreads(): its vmcnt(0) would wait for the tile's stores to be acknowledged
        const int t = threadIdx.x;
        if (t < TM && c_base + t < P.NO) {
            const long long o = (long long)grp * P.NO + c_base + t;
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int wp = 0; wp < WN; ++wp) {
                a1 += red[(wp * TM + t) * 2];
                a2 += red[(wp * TM + t) * 2 + 1];
            }
#ifdef ABL_STAT_NOATOM   // ablation (scripts/bench_fat.py): the whole reduction, no atomics (a never-taken store keeps it alive)
            if (a1 == 12345.678f && a2 == 0.f) P.stat_sum[o] = 1.0;
#else
            atomicAdd(P.stat_sum + o, (double)a1);
            atomicAdd(P.stat_sq + o, (double)a2);
#endif
        }
    }
}

// The fat bf16 kernels' entry: bias (the few layers that have one) is added once up front under a wave-uniform branch and
// the statistics variant is a compile-time copy of the epilogue (2 352 -> ~1 500 / ~700 instructions per wave; 39.6 -> 37.7 us
// on the 728 -> 728 launch with statistics).  The 64 x 64-per-wave kernels and the fp32 fat tiles keep the run-time-flag
// version above: the two inlined copies cost them 12-30 registers, i.e. a wave of occupancy or spills.
template <typename T, int MI, int NJ, int WM, int WN>
__device__ __forceinline__ void conv_epilogue_fat_ct(const GemmConvParams& P, f32x4 (&acc)[MI][NJ], int p_base, int rows_valid,
                                                     int grp, int c_base, int wave_c, int wave_p, int lane, char* smem) {
    if (P.bias) {
        const int q = lane >> 4;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int co = c_base + wave_c * MI * 16 + i * 16 + q * 4;
            const f32x4 bv = co < P.NO ? *reinterpret_cast<const f32x4*>(P.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] += bv;
        }
    }
    if (P.stat_sum) conv_epilogue_fat_impl<T, MI, NJ, WM, WN, true>(P, acc, p_base, rows_valid, grp, c_base, wave_c, wave_p, lane, smem);
    else conv_epilogue_fat_impl<T, MI, NJ, WM, WN, false>(P, acc, p_base, rows_valid, grp, c_base, wave_c, wave_p, lane, smem);
}

// PW1: a single tap (1x1 convolutions): the address set-up runs once and its registers are free in the K loop.
// BKB: bytes of K per row and K-step.  128 = whole cache lines per row (a 1-KiB DMA piece is 8 rows x 128 B), two
// stages of 80 KiB: a stage is issued right after the barrier that retires its predecessor and has the 84 MFMAs per wave
// of that step to land.  64 = half lines (16 rows x 64 B per piece; the other half of every line is fetched again by the
// next K-step), four stages, two steps in flight across each barrier.
template <typename T, int BKB, int MI, int NJ, int WM, int WN, int NBUF, bool PW1>
__global__ __launch_bounds__(512) void gemm_conv_fat_kernel(GemmConvParams P) {
    constexpr int ES = (int)sizeof(T);
    constexpr int BK = BKB / ES;
    constexpr int CPR = BKB / 16;          // 16-byte chunks per row
    constexpr int RPG = 64 / CPR;          // rows per DMA piece: 16 or 8
    constexpr int PPS = 16 / RPG;          // pieces per 16-row slot: 1 or 2
    constexpr int TM = WM * MI * 16, TN = WN * NJ * 16;
    constexpr int SA = TM / 128, SB = (TN + 127) / 128;   // 16-row slots per wave and K-step
    constexpr int GROUP = (SA + SB) * PPS;                // DMA pieces per wave and K-step
    constexpr int A_BYTES = TM * BKB, STAGE_BYTES = (TM + SB * 128) * BKB;
    constexpr int DIST = NBUF - 1;
    static_assert(BKB == 64 || BKB == 128, "64- or 128-byte rows");
    static_assert(WM * WN == 8 && TM % 128 == 0, "8 waves; A rows in 128-row slots");
    static_assert(DIST >= 1 && DIST <= 3, "counted waits are written for 1 to 3 K-steps of prefetch");
    static_assert(NBUF * STAGE_BYTES <= 160 * 1024, "staging ring exceeds the LDS");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_c = wave / WN, wave_p = wave % WN;

    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    }
    const int tile_c = bid % P.tiles_c, tile_p = bid / P.tiles_c;
    const int grp = tile_p / P.tiles_per_group, tig = tile_p - grp * P.tiles_per_group;
    const int p_base = grp * P.group_pix + tig * P.tn_valid;
    const int rows_valid = min(P.tn_valid, P.group_pix - tig * P.tn_valid);   // >= 1 (launcher)
    const int c_base = tile_c * TM;
    const int RS = P.KH * P.KW;

    // Activation descriptor REBASED to the first source row this tile can touch (64-bit base per workgroup): offsets are
    // relative to it and small, so the operand may be larger than 2 GiB.  Every source pixel of a pixel p >= p_base lies
    // at or behind the start of row max(0, first tap row of p_base) of p_base's image.
    const unsigned ohw_ = (unsigned)(P.OH * P.OW);
    const int n0 = (int)((unsigned)p_base / ohw_);
    const int oh0 = (int)(((unsigned)p_base - (unsigned)n0 * ohw_) / (unsigned)P.OW);
    int row0c;
    if (!P.transposed) row0c = max(0, oh0 * P.stride - P.pad);
    else { const int t0 = oh0 + P.pad - (P.KH - 1) * P.dil; row0c = t0 <= 0 ? 0 : t0 / P.stride; }
    const long long base_pix = ((long long)n0 * P.IH + row0c) * P.IW;
    const long long in_rem = (((long long)P.N * P.IH * P.IW - base_pix - 1) * P.ldi + P.CK) * ES;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(reinterpret_cast<const T*>(P.in) + base_pix * P.ldi), 0, bg_records(in_rem), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(P.w), 0, P.w_bytes, 0x00020000);

    const int lr = lane / CPR, lc = lane % CPR;
    // Source-side swizzle (the LDS image of a DMA is lane-linear).  Slots start on multiples of 16 rows and pieces on
    // multiples of RPG, so a lane's source chunk depends on the piece's position inside its slot only.
    int chk[PPS];
    bool tail_cut[PPS];   // see gemm_conv_kernel
    const int ksteps_per_tap = (P.CK + BK - 1) / BK;
#pragma unroll
    for (int h = 0; h < PPS; ++h) {
        const int row = h * RPG + lr;   // row inside the 16-row slot
        chk[h] = ((BKB == 64) ? (lc ^ ((0 - (row >> 2)) & 3)) : (lc ^ ((row >> 1) & 7))) * 16;
        tail_cut[h] = (ksteps_per_tap - 1) * BK + chk[h] / ES >= P.CK;
    }
    int w_base[SA][PPS], pix_base[SB][PPS], pix_n[SB][PPS], pix_h[SB][PPS], pix_w[SB][PPS];
    bool pix_ok[SB][PPS];
    const bool direct = !P.transposed || P.stride == 1;
    const int sgn = P.transposed ? -1 : 1;
#pragma unroll
    for (int s = 0; s < SA; ++s)
#pragma unroll
        for (int h = 0; h < PPS; ++h) {
            const int co = c_base + s * 128 + wave * 16 + h * RPG + lr;
            w_base[s][h] = co < P.NO ? co * RS * P.CKp * ES + chk[h] : OOB;
        }
#pragma unroll
    for (int s = 0; s < SB; ++s)
#pragma unroll
        for (int h = 0; h < PPS; ++h) {
            const int row = s * 128 + wave * 16 + h * RPG + lr;
            pix_ok[s][h] = row < rows_valid;
            const unsigned pp = pix_ok[s][h] ? (unsigned)(p_base + row) : 0u;
            const unsigned ohw = (unsigned)(P.OH * P.OW);
            const unsigned n = pp / ohw;
            const unsigned rem = pp - n * ohw;
            const unsigned qq = rem / (unsigned)P.OW;
            const int oh = (int)qq, ow = (int)(rem - qq * (unsigned)P.OW);
            pix_n[s][h] = (int)n - n0;   // relative to the rebased descriptor
            if (!P.transposed) {
                pix_h[s][h] = oh * P.stride - P.pad;
                pix_w[s][h] = ow * P.stride - P.pad;
            } else {
                pix_h[s][h] = oh + P.pad;
                pix_w[s][h] = ow + P.pad;
            }
            pix_base[s][h] = ((pix_n[s][h] * P.IH + pix_h[s][h] - row0c) * P.IW + pix_w[s][h]) * P.ldi * ES + chk[h];
        }

    const int KT = RS * ksteps_per_tap;
    int l_tap_r = 0, l_tap_s = 0, l_ks = 0, l_tap = 0;
    int va[SA][PPS], vb[SB][PPS];
    auto start_tap = [&]() {
        const int dh = sgn * l_tap_r * P.dil, dw_ = sgn * l_tap_s * P.dil;
        const int tap_delta = (dh * P.IW + dw_) * P.ldi * ES;
#pragma unroll
        for (int s = 0; s < SA; ++s)
#pragma unroll
            for (int h = 0; h < PPS; ++h) va[s][h] = w_base[s][h] == OOB ? OOB : w_base[s][h] + l_tap * P.CKp * ES;
#pragma unroll
        for (int s = 0; s < SB; ++s)
#pragma unroll
            for (int h = 0; h < PPS; ++h) {
                if (direct) {
                    const int ih = pix_h[s][h] + dh, iw = pix_w[s][h] + dw_;
                    const bool ok = pix_ok[s][h] && (unsigned)ih < (unsigned)P.IH && (unsigned)iw < (unsigned)P.IW;
                    vb[s][h] = ok ? pix_base[s][h] + tap_delta : OOB;
                } else {
                    const int th = pix_h[s][h] - l_tap_r * P.dil, tw = pix_w[s][h] - l_tap_s * P.dil;
                    bool ok = pix_ok[s][h] && th >= 0 && tw >= 0;
                    const int ih = th / P.stride, iw = tw / P.stride;
                    ok = ok && (ih * P.stride == th) && (iw * P.stride == tw) && ih < P.IH && iw < P.IW;
                    vb[s][h] = ok ? ((pix_n[s][h] * P.IH + ih - row0c) * P.IW + iw) * P.ldi * ES + chk[h] : OOB;
                }
            }
    };
    if (PW1) start_tap();
    // One K-step's staging = GROUP pieces per wave.  issue_begin()/issue_end() carry the (scalar) tap bookkeeping,
    // piece(buf, pi) issues piece pi: the A slots' pieces first, then the B slots'.
    bool last_ks = false;
    auto issue_begin = [&]() {
        if (!PW1 && l_ks == 0) start_tap();
        last_ks = l_ks == ksteps_per_tap - 1;
    };
    auto piece = [&](int buf, int pi) {   // pi is a compile-time constant at every call site (unrolled loops)
        char* stage_a = smem + buf * STAGE_BYTES + wave * 16 * BKB;
        if (pi < SA * PPS) {
#ifndef ABL_FAT_NO_A   // ablation (scripts/ablate_fat.sh): no A-operand traffic at all -- the bound of "weights through VGPRs"
            const int s_ = pi / PPS, h = pi % PPS;
            dma16(rs_w, stage_a + (s_ * 128 + h * RPG) * BKB, va[s_][h]);
            va[s_][h] += BKB;   // an out-of-range marker stays out of range
#endif
        } else {
            const int s_ = (pi - SA * PPS) / PPS, h = (pi - SA * PPS) % PPS;
            dma16(rs_in, stage_a + A_BYTES + (s_ * 128 + h * RPG) * BKB, (last_ks && tail_cut[h]) ? OOB : vb[s_][h]);
            vb[s_][h] += BKB;
        }
    };
    auto issue_end = [&]() {
        if (++l_ks == ksteps_per_tap) {
            l_ks = 0;
            ++l_tap;
            if (++l_tap_s == P.KW) { l_tap_s = 0; ++l_tap_r; }
        }
    };
    auto issue = [&](int buf) {
        issue_begin();
#pragma unroll
        for (int pi = 0; pi < GROUP; ++pi) piece(buf, pi);
        issue_end();
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    BG_STAMP(0);
    issue(0);
    if (DIST >= 2 && KT > 1) issue(1);
    if (DIST >= 3 && KT > 2) issue(2);
    BG_STAMP(1);
    // In the loop the pieces of stage kt+DIST are issued BETWEEN the MFMA groups of step kt (one group = the NJ MFMAs
    // of one A fragment), PPG per group, instead of as a burst behind the barrier: a wave sits ~100-150 cycles in the
    // issue of one piece, and with the burst both waves of a SIMD did so at the same time with the matrix pipe idle
    // (measured with in-kernel stamps: a K-step took its MFMA time PLUS its issue time).  Spread out, one wave's issue
    // stall is the other's MFMA slot.
    constexpr int NGRP = (IsFp8<T>::value ? BKB / 128 : sizeof(T) == 2 ? BKB / 64 : BKB / 16) * MI;   // MFMA groups per K-step
    // fp8: E8M0 block scales 2^-e in byte 0 of the scale operands (the same for every lane and block)
    int scale_w = 127, scale_act = 127;
    if constexpr (IsFp8<T>::value) {
        scale_w = (127 - (P.exp_w ? *P.exp_w : 0)) & 0xff;
        scale_act = (127 - (P.exp_act ? *P.exp_act : 0)) & 0xff;
    }
    constexpr int PPG = NBUF == 2 ? (GROUP + NGRP / 2 - 1) / (NGRP / 2) : (GROUP + NGRP - 1) / NGRP;   // a two-stage ring's pieces
                                                                  // must land within this step: all in its first half
    static_assert(PPG * NGRP >= GROUP, "not enough MFMA groups to carry the pieces");
    const int r16 = lane & 15, q = lane >> 4;
    const int rowA = wave_c * MI * 16, rowB = wave_p * NJ * 16;
    int buf = 0, nbuf = DIST % NBUF;
#ifdef ABL_FAT_PRODUCER
    AblRegs abl;
    abl_init(abl, lane);
#endif
    // A real s_waitcnt lgkmcnt(0) the compiler can SEE (encoding: vmcnt 63, expcnt 7, lgkmcnt 0): a scalar load still pending on
    // the loop's entry edge makes its waitcnt pass treat lgkmcnt as out-of-order at the loop header, and then the first LDS
    // wait of EVERY iteration is lgkmcnt(0) -- all nine fragment reads before the first MFMA -- instead of a count.
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int kt = 0; kt < KT; ++kt) {
#ifdef BG_STAMPS
        if (kt == 1) { BG_STAMP(2); BG_STAMP_CYC(6); }
#endif
        // retire this wave's pieces of step kt (the later steps' stay in flight), then meet the other waves
        const int ahead = KT - 1 - kt;
        if (DIST >= 3 && ahead >= 2) wait_vmcnt<2 * GROUP>();
        else if (DIST >= 2 && ahead >= 1) wait_vmcnt<GROUP>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        // stage kt+DIST: its ring slot was last read in iteration kt-1, which every wave has left
        const bool more = kt + DIST < KT;
        if (more) issue_begin();
        const char* sA = smem + buf * STAGE_BYTES;
        const char* sB = sA + A_BYTES;
        if constexpr (IsFp8<T>::value) {
            // one MFMA 16x16x128 (block-scaled f8f6f4 form: twice the bf16 rate) covers the whole 128-byte row.  A lane
            // supplies 32 of its row's 128 K-elements; WHICH 32 is free as long as both operands agree (the sum over K
            // is order-independent), so a lane takes chunks q and 4 + q -- the two ds_read_b128 of the bf16 loop's two
            // K-substeps, conflict-free under the same swizzle.
            static_assert(BKB == 128, "fp8: 128-byte rows = one MFMA K");
            auto load_a = [&](int i) {
                const i32x4 lo = *reinterpret_cast<const i32x4*>(sA + lds_off<BKB>(rowA + i * 16 + r16, q));
                const i32x4 hi = *reinterpret_cast<const i32x4*>(sA + lds_off<BKB>(rowA + i * 16 + r16, 4 + q));
                return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            i32x8 a = load_a(0);        // the A fragment first: the first MFMA then waits for four reads, not sixteen (bf16 path below)
            i32x8 b[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const i32x4 lo = *reinterpret_cast<const i32x4*>(sB + lds_off<BKB>(rowB + j * 16 + r16, q));
                const i32x4 hi = *reinterpret_cast<const i32x4*>(sB + lds_off<BKB>(rowB + j * 16 + r16, 4 + q));
                b[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
#pragma unroll
                    for (int e = 0; e < PPG; ++e)
                        if (i * PPG + e < GROUP) piece(nbuf, i * PPG + e);
                }
                i32x8 an = a;
                if (i + 1 < MI) an = load_a(i + 1);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        a, b[j], acc[i][j], 0 /* A: e4m3 */, std::is_same<T, f8e5_t>::value ? 1 : 0 /* B: e5m2 / e4m3 */,
                        0, scale_w, 0, scale_act);
                __builtin_amdgcn_s_setprio(0);
                a = an;
            }
        } else if constexpr (sizeof(T) == 2) {
            // The fragments of K-substep ks + 1 are read during the LAST MFMA group of substep ks -- the A fragment as that
            // group's look-ahead read, each B fragment right after the MFMA that took its predecessor (same registers) -- so
            // a substep starts with its operands in flight instead of with nine reads and an idle matrix pipe.  The first
            // substep's reads stand behind the step's barrier: the A fragment first, then the B fragments in the order
            // the MFMAs take them (the LDS returns in order: the first MFMA waits for two reads, not nine); the first group's
            // pieces go out BEFORE the reads (a branch between the reads and their MFMAs cost the counted waits).
            constexpr int NKS = BKB / 64;
            auto rd_a = [&](int i, int ks) { return *reinterpret_cast<const bf16x8*>(sA + lds_off<BKB>(rowA + i * 16 + r16, ks * 4 + q)); };
            auto rd_b = [&](int j, int ks) { return *reinterpret_cast<const bf16x8*>(sB + lds_off<BKB>(rowB + j * 16 + r16, ks * 4 + q)); };
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int e = 0; e < PPG; ++e)
                    if (e < GROUP) piece(nbuf, e);
            }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 b[NJ];
#ifndef ABL_FAT_NO_A   // ablation (scripts/ablate_fat.sh): no A fragment reads
            bf16x8 a = rd_a(0, 0);
#endif
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = rd_b(j, 0);
#ifdef ABL_FAT_NO_A
            bf16x8 a = b[0];
#endif
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more && ks * MI + i > 0) {
#pragma unroll
                        for (int e = 0; e < PPG; ++e)
                            if ((ks * MI + i) * PPG + e < GROUP) piece(nbuf, (ks * MI + i) * PPG + e);
                    }
                    const bool carry = i == MI - 1 && ks + 1 < NKS;     // this group hands over to the next substep
#ifdef ABL_FAT_PRODUCER
                    abl_producer(abl, sB + lane * 16);
#endif
                    bf16x8 an = a;
#ifdef ABL_FAT_NO_A
                    an = b[(i + 1) % NJ];
#else
                    if (i + 1 < MI) an = rd_a(i + 1, ks);
                    else if (carry) an = rd_a(0, ks + 1);
#endif
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[j], acc[i][j], 0, 0, 0);
                        if (carry) {
                            __builtin_amdgcn_sched_barrier(0);      // the read may not move above the MFMA that frees its registers
                            b[j] = rd_b(j, ks + 1);
                        }
                    }
                    __builtin_amdgcn_s_setprio(0);
                    a = an;
                }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BKB / 16; ++kk) {
                float b[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const float*>(sB + lds_off<BKB>(rowB + j * 16 + r16, kk) + q * 4);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
#pragma unroll
                        for (int e = 0; e < PPG; ++e)
                            if ((kk * MI + i) * PPG + e < GROUP) piece(nbuf, (kk * MI + i) * PPG + e);
                    }
                    const float a = *reinterpret_cast<const float*>(sA + lds_off<BKB>(rowA + i * 16 + r16, kk) + q * 4);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue_end();
        buf = (buf + 1 == NBUF) ? 0 : buf + 1;
        nbuf = (nbuf + 1 == NBUF) ? 0 : nbuf + 1;
    }
    BG_STAMP(3);
    BG_STAMP_CYC(7);
#ifdef ABL_FAT_PRODUCER
    abl_sink(abl, P.stat_sum);
#endif
    typedef typename OutOf<T>::type TO;
    if constexpr (sizeof(TO) == 2) conv_epilogue_fat_ct<TO, MI, NJ, WM, WN>(P, acc, p_base, rows_valid, grp, c_base, wave_c, wave_p, lane, smem);
    else conv_epilogue_fat<TO, MI, NJ, WM, WN>(P, acc, p_base, rows_valid, grp, c_base, wave_c, wave_p, lane, smem);
#ifdef BG_STAMPS
    BG_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BG_STAMP(5);
#endif
}


inline int kpad_of(int dtype) { return dtype == BG_FP8 ? 128 : dtype == BG_BF16 ? 64 : 32; }   // elements per padded K granule
inline int pad_k(int c, int dtype) { const int g = kpad_of(dtype); return (c + g - 1) / g * g; }

int check_conv_desc(const bg_conv_desc* d, const char* who) {
    BG_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
    BG_CHECK_ARG(dtype_ok(d->dtype), "%s: bad dtype %d", who, d->dtype);
    const int vec = dtype_vec(d->dtype);
    BG_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->Ho > 0 && d->Wo > 0,
                 "%s: non-positive dimension", who);
    BG_CHECK_ARG(d->KH >= 1 && d->KW >= 1 && d->KH <= 7 && d->KW <= 7, "%s: kernel size %dx%d unsupported", who, d->KH,
                 d->KW);
    BG_CHECK_ARG(d->stride >= 1 && d->dil >= 1 && d->pad >= 0, "%s: bad stride/dil/pad", who);
    BG_CHECK_ARG(d->Cin % vec == 0 && d->Cout % vec == 0,
                 "%s: Cin=%d / Cout=%d must be multiples of %d (pad channels with zeros)", who, d->Cin, d->Cout, vec);
    BG_CHECK_ARG(d->ldx >= d->Cin && d->ldy >= d->Cout && d->ldx % vec == 0 && d->ldy % vec == 0,
                 "%s: bad pixel strides ldx=%d ldy=%d", who, d->ldx, d->ldy);
    const int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
    BG_CHECK_ARG(ho == d->Ho && wo == d->Wo, "%s: output %dx%d does not match conv arithmetic %dx%d", who, d->Ho, d->Wo,
                 ho, wo);
    BG_CHECK_ARG((long long)d->N * d->H * d->W * d->ldx < (1LL << 40) && (long long)d->N * d->Ho * d->Wo < (1LL << 31),
                 "%s: tensor too large", who);
    return BG_OK;
}

// Tuning / test hook (bg_conv_set_variant): -1 = by environment and heuristics, 0 = the 64 x 64-per-wave tiles only,
// 2 = the fat-tile kernel wherever it is legal (small test shapes included).
int g_conv_variant = -1;

template <typename T, int BKB, int MI, int NJ, int WM, int WN, int NBUF, bool PW1>
int launch_fat(const GemmConvParams& P, long long nblk, hipStream_t st) {
    constexpr int SB = (WN * NJ * 16 + 127) / 128;
    constexpr int SH = NBUF * (WM * MI * 16 + SB * 128) * BKB;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_fat_kernel<T, BKB, MI, NJ, WM, WN, NBUF, PW1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, SH);
        once = true;
    }
#ifdef BG_STAMPS
    GemmConvParams Q = P;
    Q.dbg = g_dbg_stamps;
    hipLaunchKernelGGL((gemm_conv_fat_kernel<T, BKB, MI, NJ, WM, WN, NBUF, PW1>), dim3((unsigned)nblk), dim3(512), SH, st, Q);
#else
    hipLaunchKernelGGL((gemm_conv_fat_kernel<T, BKB, MI, NJ, WM, WN, NBUF, PW1>), dim3((unsigned)nblk), dim3(512), SH, st, P);
#endif
    BG_CHECK_LAUNCH("gemm_conv_fat_kernel");
    return BG_OK;
}

// Fat-tile plan: out-channel tile 384 (1x1 convolutions only: its 168 accumulator registers leave no room for the
// tap bookkeeping), 256 or 128 rows (Cout <= 128), pixel tiles of at most 224 rows (112 for a 256-row tile when the
// wider one would leave half the chip idle).  The cost of a candidate is rounds-of-256-tiles x MFMA work per tile;
// the pixel width is then shrunk until the tiles fill their last round.  Returns 0 when the launch should stay on the
// 64 x 64-per-wave kernels, else 1000 * rows + padded pixel width.
template <typename T>
int plan_fat(GemmConvParams& P, bool big) {   // big: an operand beyond the classic kernels' 32-bit offsets -> fat tiles always
    static const int fat_env = getenv("BGAMD_FAT") ? atoi(getenv("BGAMD_FAT")) : 1;
    static const int fat128 = getenv("BGAMD_FAT128") ? atoi(getenv("BGAMD_FAT128")) : 0;   // A/B: 128-row tiles by heuristics too
    const int mode = g_conv_variant >= 0 ? g_conv_variant : fat_env;
    if (mode == 0 && !big) return 0;
    const int bk = 64 / (int)sizeof(T);
    const long long kt = (long long)P.KH * P.KW * ((P.CK + bk - 1) / bk);
    if (mode != 2 && !big && ((P.NO <= 128 && !fat128) || kt < 8 || P.M < 16384)) return 0;
    // measured (scripts/bench_fat.py, and in the step's launch table): on hundreds of thousands of pixels with Cout <= 256
    // (the decoder's 3x3 layers, the 256 -> 256 pointwise layers at 288x192) the classic tiles are level or ahead; on the
    // ASPP's few-pixel layers the fat tile is 13-15 % ahead.  Re-measured in round 3 after the K-loop work
    // (BGAMD_FAT_MANY_PIXELS=1): the fat tile gains 5-14 % on the 256-channel forward launches and loses 25 % on the
    // 304 -> 256 data gradient (Cout' = 304: one and a fifth 256-row tiles) -- level in the step, rule kept.
    static const bool fat_on_many = getenv("BGAMD_FAT_MANY_PIXELS") != nullptr;
    if (!fat_on_many && mode != 2 && !big && P.M >= 131072 && (P.KH * P.KW > 1 || P.NO <= 256)) return 0;
    const int groups = P.stat_group_pix ? (int)(P.M / P.stat_group_pix) : 1;
    const long long gp = P.M / groups;
    if (gp * groups != P.M || gp >= (1LL << 31)) return 0;
    const bool pw1 = P.KH * P.KW == 1;
    constexpr int NCU = 256;
    static const int cand[4][2] = {{384, 224}, {256, 224}, {256, 112}, {128, 224}};
    int best = -1;
    long long best_cost = 0, best_rounds = 0;
    for (int c = 0; c < 4; ++c) {
        const int tm = cand[c][0], tnp = cand[c][1];
        if (tm == 384 && (!pw1 || sizeof(T) == 1)) continue;   // fp8: 56 registers of B fragments beside the accumulators
        static const bool no384 = getenv("BGAMD_FAT_NO384") != nullptr;   // A/B (scripts/bench_corun.py): 256-row tiles leave room on the CU
        if (tm == 384 && no384) continue;
        if ((tm == 128) != (P.NO <= 128)) continue;
        const long long tc = (P.NO + tm - 1) / tm, tp0 = groups * ((gp + tnp - 1) / tnp);
        const long long rounds = (tc * tp0 + NCU - 1) / NCU, cost = rounds * tm * tnp;
        if (best < 0 || cost < best_cost) { best = c; best_cost = cost; best_rounds = rounds; }   // ties: the larger tile, listed first
    }
    const int tm = cand[best][0], tnp = cand[best][1];
    P.tiles_c = (P.NO + tm - 1) / tm;
    long long tpg = (gp + tnp - 1) / tnp;                            // pixel tiles per group, at least
    const long long fill = best_rounds * NCU / P.tiles_c / groups;   // ... and as many as the last round has room for
    if (fill > tpg) tpg = fill;
    long long tnv = (gp + tpg - 1) / tpg;
    tpg = (gp + tnv - 1) / tnv;                                      // no empty tiles
    P.tn_valid = (int)tnv;
    P.tiles_per_group = (int)tpg;
    P.group_pix = (int)gp;
    P.tiles_p = (int)(tpg * groups);
    return tm * 1000 + tnp;
}

}  // namespace
