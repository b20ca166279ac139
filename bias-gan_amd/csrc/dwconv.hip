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
    int rb, bands;  // dw_s1_kernel: output rows per band, bands per image
    int wgroups, bands_per_phase;  // ... per column phase / row phase (dilation D: D phases each way)
    // PRE = 1 (dw_s1_kernel): the input is a raw convolution output; the normalisation's affine and the activation
    // are applied to every loaded chunk (x -> act(x*scale + shift), out-of-image taps stay zero)
    const float* pre_scale;  // [groups, C]
    const float* pre_shift;
    int pre_act;
    int pre_ipg;  // images per statistic group
    // ... or, with pre_sum set (dw_ring_kernel only), the fp64 sums of the producer's statistics epilogue: every block
    // derives the affine of its image's group itself (bg_norm_finalize_affine's arithmetic, once per channel, through
    // LDS) and the first block of each group publishes mean / rstd / scale / shift for the backward pass; the block of
    // image 0 also applies the running-statistics update -- no finalize launch between the GEMM and this kernel
    const double* pre_sum;
    const double* pre_sumsq;
    const float* pre_gamma;
    const float* pre_beta;
    float pre_eps, pre_mom;
    float* pre_rmean;
    float* pre_rvar;
    float* pre_mean_out;
    float* pre_rstd_out;
    float* pre_scale_out;
    float* pre_shift_out;
    long long pre_rpg;  // rows (pixels) per statistic group
    int pre_groups;
    // ADD = 1 (dw_ring_kernel, data gradient): y = conv(x) + add, add an [N, Ho, Wo, C] tensor of pixel stride ldadd
    const void* add;
    int ldadd;
};

// Normalisation + activation of one loaded chunk, rounded to T exactly as bg_norm_act_fwd would have stored it (the
// fused and the unfused pipelines give identical bits).  keep = false: the tap lies outside the image ("same" zero
// padding applies to the ACTIVATED tensor, so an out-of-range load's zero must stay zero).
template <typename T>
__device__ __forceinline__ void dw_pre_apply(Chunk<T>& v, const float* sc, const float* sh, float slope, bool keep) {
    constexpr int VEC = Elem<T>::VEC;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < VEC; e += 2) {   // pairs: v_pk_fma_f32 / v_pk_mul_f32 (the same fused multiply-add per element)
        const f32x2 x = {v.get(e), v.get(e + 1)}, s = {sc[e], sc[e + 1]}, h = {sh[e], sh[e + 1]};
        const f32x2 z = __builtin_elementwise_fma(x, s, h);
        const f32x2 zs = z * slope;
        o.set(e, fmaxf(z[0], zs[0]));
        o.set(e + 1, fmaxf(z[1], zs[1]));
    }
    if (!keep) o.zero();
    v = o;
}

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

// Stride 1, dilation 1 (all middle-flow units): one thread owns TW = 4 output columns of one
// channel vector and walks RB consecutive output rows with the 3-row input window held in
// registers -- every input element is loaded by 1.5 threads instead of 4.5 (L2 -> L1 traffic was
// the limiter of the per-row kernel above).  Loads are buffer loads with one descriptor per input
// row: columns left/right of the image and rows above/below it fall outside the descriptor's range
// and read as zero, which IS the "same" padding -- no branches around the loads.  The next input
// row is fetched while the current one is being used.  FLIP = 1 reverses the taps (data gradient).
// Dilation D (exit flow, D = 2) is the same kernel on the D x D sub-lattices of the image: a thread's
// four output columns and the rows it walks are D apart, so its window is again 3 rows x 6 columns.
template <typename T, int FLIP, int PRE>
__global__ __launch_bounds__(256) void dw_s1_kernel(DwParams P) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int TW = 4, NCOL = TW + 2;
    typedef typename Elem<T>::vec_t vec_t;
    const unsigned cv = P.C / VEC;
    int band, xblk;
    dw_block_to_row(P.bx, band, xblk);
    const unsigned idx = xblk * 256u + threadIdx.x;
    if (idx >= (unsigned)P.items) return;
    const unsigned wq = idx / cv;
    const int c = (int)(idx - wq * cv) * VEC;
    const int D = P.dil;
    const int pc = (int)wq / P.wgroups, gi = (int)wq - pc * P.wgroups;  // column phase, group inside it
    const int wo0 = pc + gi * TW * D;
    const int n = band / P.bands, b = band - n * P.bands;
    const int pr = b / P.bands_per_phase, bi = b - pr * P.bands_per_phase;  // row phase, band inside it
    const int ho0 = pr + bi * P.rb * D;
    const int ho1 = min(ho0 + P.rb * D, P.Ho);
    const unsigned row_bytes = (unsigned)P.W * P.ldx * sizeof(T);
    const char* xn = reinterpret_cast<const char*>(P.x) + (long long)n * P.H * row_bytes;
    int voff[NCOL];  // negative (left of the image) -> huge unsigned -> out of range -> 0
#pragma unroll
    for (int j = 0; j < NCOL; ++j) voff[j] = ((wo0 + (j - 1) * D) * P.ldx + c) * (int)sizeof(T);
    Chunk<T> wv[9];
    {
        const T* w = reinterpret_cast<const T*>(P.w) + c;
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t].load(w + (FLIP ? 8 - t : t) * P.C);
    }
    auto load_row = [&](int ih, Chunk<T>(&dst)[NCOL]) {
        const bool ok = (unsigned)ih < (unsigned)P.H;  // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(xn + (long long)(ok ? ih : 0) * row_bytes), 0, ok ? row_bytes : 0u, 0x00020000);
#pragma unroll
        for (int j = 0; j < NCOL; ++j) dst[j].v = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j], 0, 0));
    };
    float psc[VEC], psh[VEC];
    bool cok[NCOL];
    const float pslope = act_max_slope(P.pre_act);
    if (PRE) {
        const long long o = (long long)(n / P.pre_ipg) * P.C + c;
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(P.pre_scale + o + e);
            const f32x4 b = *reinterpret_cast<const f32x4*>(P.pre_shift + o + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                psc[e + k] = a[k];
                psh[e + k] = b[k];
            }
        }
#pragma unroll
        for (int j = 0; j < NCOL; ++j) cok[j] = (unsigned)(wo0 + (j - 1) * D) < (unsigned)P.W;
    }
    auto pre_row = [&](int ih, Chunk<T>(&row)[NCOL]) {
        const bool rok = (unsigned)ih < (unsigned)P.H;
#pragma unroll
        for (int j = 0; j < NCOL; ++j) dw_pre_apply<T>(row[j], psc, psh, pslope, rok && cok[j]);
    };
    Chunk<T> r0[NCOL], r1[NCOL], r2[NCOL], nx[NCOL];
    load_row(ho0 - D, r0);
    load_row(ho0, r1);
    load_row(ho0 + D, r2);
    if (PRE) {
        pre_row(ho0 - D, r0);
        pre_row(ho0, r1);
        pre_row(ho0 + D, r2);
    }
    T* y = reinterpret_cast<T*>(P.y) + (((long long)n * P.Ho + ho0) * P.Wo + wo0) * P.ldy + c;
    for (int ho = ho0; ho < ho1; ho += D) {
        if (ho + D < ho1) load_row(ho + 2 * D, nx);  // prefetch: consumed in the next iteration
        float acc[TW][VEC];
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    acc[t][e] = fmaf(r0[t + s].get(e), wv[0 + s].get(e), acc[t][e]);
                    acc[t][e] = fmaf(r1[t + s].get(e), wv[3 + s].get(e), acc[t][e]);
                    acc[t][e] = fmaf(r2[t + s].get(e), wv[6 + s].get(e), acc[t][e]);
                }
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            if (wo0 + t * D < P.Wo) {
                Chunk<T> o;
#pragma unroll
                for (int e = 0; e < VEC; ++e) o.set(e, acc[t][e]);
                o.store(y + (long long)t * D * P.ldy);
            }
        }
        y += (long long)D * P.Wo * P.ldy;
        if (PRE && ho + D < ho1) pre_row(ho + 2 * D, nx);
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
            r0[j] = r1[j];
            r1[j] = r2[j];
            r2[j] = nx[j];
        }
    }
}

// dw_s1_kernel with the input rows prefetched through LDS instead of registers.  In dw_s1_kernel a wave has ONE input
// row (6 KiB) in flight while it computes -- all the registers allow -- and a block's life is a chain of exposed
// memory latencies: rocprofv3 put it at 3.1 TB/s where a copy of the same tensor runs at 4.5.  Here every wave owns a
// ring of PF row slots in LDS that its own lanes fill by LDS-DMA (buffer_load ... lds: no VGPRs, lane i's 16 bytes land at
// slot + 16 i, so a lane reads back exactly what it fetched and no barrier is needed) PF rows ahead of the row being
// computed; the wait is a COUNTED s_waitcnt vmcnt.  gfx9 has one counter for loads and stores and retires them in issue
// order, so the output stores are buffer stores that are ALWAYS issued (columns past the image edge fall outside the
// row descriptor's range and are dropped): the number of younger operations at every wait is then known statically.
// (non-template helper: see dma16 in igemm_conv.hip)
__device__ __forceinline__ void dw_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_dst, 16, voffset, 0, 0, 0);
}
template <int N>
__device__ __forceinline__ void dw_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
constexpr int DW_PF = 3;                       // rows in flight per wave
constexpr int DW_RING_LDS = 4 * DW_PF * 6 * 1024;  // 4 waves x PF slots x 6 columns x 1 KiB
constexpr int DW_RING_LDS_ADD = 4 * DW_PF * 10 * 1024;  // ... x (6 + 4) columns: the addend's four pixels ride in the slot

// ADD = 1: the stored value is conv(x) + add (the data gradient of a Block's first depthwise convolution plus the
// gradient that reaches the Block's input through the skip path: the sum autograd would form in a pass of its own).
// The addend's four chunks of the output row that slot q completes travel through the same ring slot.
template <typename T, int FLIP, int PRE, int ADD = 0>
__global__ __launch_bounds__(256) void dw_ring_kernel(DwParams P) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int TW = 4, NCOL = TW + 2, PF = DW_PF, NC = ADD ? NCOL + TW : NCOL;
    constexpr int RING_LDS = ADD ? DW_RING_LDS_ADD : DW_RING_LDS;
    typedef typename Elem<T>::vec_t vec_t;
    extern __shared__ __align__(16) char dw_smem[];
    const unsigned cv = P.C / VEC;
    int band, xblk;
    dw_block_to_row(P.bx, band, xblk);
    const unsigned idx = xblk * 256u + threadIdx.x;
    const int n = band / P.bands, b = band - n * P.bands;
    float* ptab = reinterpret_cast<float*>(dw_smem + RING_LDS);  // [2][C] scale, shift (PRE with pre_sum)
    // Order of the prologue: the ring's first rows are requested BEFORE the BatchNorm coefficients are formed.  The
    // coefficient table needs a round trip of its own (the producer's fp64 sums) and the DMA does not depend on it, so the
    // two latencies overlap instead of adding up.  Worth 0.3 % of the step (same-box A/B of two builds, scripts/gpu_ab_lib.sh);
    // what the prologue costs besides is its arithmetic: 4.7 us plain, 6.5 with scale / shift tables, 7.5 with the tables
    // derived here, on 16 x 16 maps with hot operands (scripts/bench_dw_small.py).  Threads beyond the tile's items
    // request nothing (out-of-range offsets) and leave after the table's barrier.
    const bool active = idx < (unsigned)P.items;
    const unsigned wq = idx / cv;
    const int c = (int)(idx - wq * cv) * VEC;
    const int D = P.dil;
    const int pc = (int)wq / P.wgroups, gi = (int)wq - pc * P.wgroups;  // column phase, group inside it
    const int wo0 = pc + gi * TW * D;
    const int pr = b / P.bands_per_phase, bi = b - pr * P.bands_per_phase;  // row phase, band inside it
    const int ho0 = pr + bi * P.rb * D;
    const int ho1 = min(ho0 + P.rb * D, P.Ho);
    const unsigned row_bytes = (unsigned)P.W * P.ldx * sizeof(T);
    const unsigned orow_bytes = (unsigned)P.Wo * P.ldy * sizeof(T);
    const char* xn = reinterpret_cast<const char*>(P.x) + (long long)n * P.H * row_bytes;
    char* yn = reinterpret_cast<char*>(P.y) + (long long)n * P.Ho * orow_bytes;
    int voff[NCOL], goff[TW], aoff[TW];  // negative (left of the image) -> huge unsigned -> out of range -> 0 / dropped
#pragma unroll
    for (int j = 0; j < NCOL; ++j) voff[j] = active ? ((wo0 + (j - 1) * D) * P.ldx + c) * (int)sizeof(T) : (int)0x80000000;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        goff[t] = ((wo0 + t * D) * P.ldy + c) * (int)sizeof(T);
        aoff[t] = ADD ? (active ? ((wo0 + t * D) * P.ldadd + c) * (int)sizeof(T) : (int)0x80000000) : 0;
    }
    const unsigned arow_bytes = ADD ? (unsigned)P.Wo * P.ldadd * sizeof(T) : 0u;
    const char* an = ADD ? reinterpret_cast<const char*>(P.add) + (long long)n * P.Ho * arow_bytes : nullptr;
    const int lane = threadIdx.x & 63;
    char* wb = dw_smem + (threadIdx.x >> 6) * (PF * NC * 1024);
    const int nout = (ho1 - ho0 + D - 1) / D;  // output rows of this band (on its row phase)
    const int Q = nout + 2;                    // input rows q = 0 .. Q-1 at image row ho0 + (q-1) D
    auto issue_row = [&](int q, int slot) {
        const int ih = ho0 + (q - 1) * D;
        const bool ok = (unsigned)ih < (unsigned)P.H;  // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(xn + (long long)(ok ? ih : 0) * row_bytes), 0, ok ? row_bytes : 0u, 0x00020000);
#pragma unroll
        for (int j = 0; j < NCOL; ++j) dw_dma16(rs, wb + (slot * NC + j) * 1024, voff[j]);
        if (ADD) {   // the addend of output row ho0 + (q-2) D, the row this slot completes
            const int oh = ho0 + (q - 2) * D;
            const bool aok = q >= 2 && oh < ho1;
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(an + (long long)(aok ? oh : 0) * arow_bytes), 0, aok ? arow_bytes : 0u, 0x00020000);
#pragma unroll
            for (int t = 0; t < TW; ++t) dw_dma16(ra, wb + (slot * NC + NCOL + t) * 1024, aoff[t]);
        }
    };
#pragma unroll
    for (int q = 0; q < PF; ++q)
        if (q < Q) issue_row(q, q);
    Chunk<T> wv[9];
    if (active) {
        const T* w = reinterpret_cast<const T*>(P.w) + c;
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t].load(w + (FLIP ? 8 - t : t) * P.C);
    }
    if (PRE && P.pre_sum) {   // block-uniform
        const int g = n / P.pre_ipg;
        const bool publish = b == 0 && xblk == 0 && n == g * P.pre_ipg;  // first block of the group's first image
        const double cnt = (double)P.pre_rpg, inv_n = 1.0 / cnt;
        for (int ch = threadIdx.x; ch < P.C; ch += 256) {
            const long long i = (long long)g * P.C + ch;
            const double m = P.pre_sum[i] * inv_n;
            double var = P.pre_sumsq[i] * inv_n - m * m;
            if (var < 0.0 || P.pre_rpg == 1) var = 0.0;
            const float r = rsqrtf((float)var + P.pre_eps);
            float a, sh_;
            const float gm = P.pre_gamma ? P.pre_gamma[ch] : 1.f, bt = P.pre_beta ? P.pre_beta[ch] : 0.f;
            // norm_affine of norm_act.hip: the backward kernels re-form the affine this way from mean / rstd
            a = gm * r;
            sh_ = fmaf(-((float)m * gm), r, bt);
            ptab[ch] = a;
            ptab[P.C + ch] = sh_;
            if (publish) {
                P.pre_mean_out[i] = (float)m;
                P.pre_rstd_out[i] = r;
                P.pre_scale_out[i] = a;
                P.pre_shift_out[i] = sh_;
                if (P.pre_rmean && g == 0) {   // one momentum update per statistic group, in group order
                    float rm = P.pre_rmean[ch], rv = P.pre_rvar[ch];
                    for (int gg = 0; gg < P.pre_groups; ++gg) {
                        const long long j = (long long)gg * P.C + ch;
                        const double mg = P.pre_sum[j] * inv_n;
                        double vg = P.pre_sumsq[j] * inv_n - mg * mg;
                        if (vg < 0.0 || P.pre_rpg == 1) vg = 0.0;
                        const double unb = P.pre_rpg > 1 ? vg * cnt / (cnt - 1.0) : vg;
                        rm = (1.f - P.pre_mom) * rm + P.pre_mom * (float)mg;
                        rv = (1.f - P.pre_mom) * rv + P.pre_mom * (float)unb;
                    }
                    P.pre_rmean[ch] = rm;
                    P.pre_rvar[ch] = rv;
                }
            }
        }
        __syncthreads();
    }
    if (!active) return;
    float psc[VEC], psh[VEC];
    bool cok[NCOL];
    const float pslope = act_max_slope(P.pre_act);
    if (PRE) {
        const long long o = (long long)(n / P.pre_ipg) * P.C + c;
        const float* tsc = P.pre_sum ? ptab + c : P.pre_scale + o;      // LDS table of this block / global table
        const float* tsh = P.pre_sum ? ptab + P.C + c : P.pre_shift + o;
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(tsc + e);
            const f32x4 bq = *reinterpret_cast<const f32x4*>(tsh + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                psc[e + k] = a[k];
                psh[e + k] = bq[k];
            }
        }
#pragma unroll
        for (int j = 0; j < NCOL; ++j) cok[j] = (unsigned)(wo0 + (j - 1) * D) < (unsigned)P.W;
    }
    // Scatter form: the new input row is unpacked ONCE and added into the three output rows it contributes to (taps
    // 0-2 of output row q, 3-5 of row q-1, 6-8 of row q-2, which is complete afterwards and stored).  The gather form --
    // three packed rows kept, all three unpacked again for every output row -- spent 144 of its 304 VALU instructions
    // per row on those conversions, and with one workgroup per CU the kernel is VALU-bound (SQ: 71 % of wave cycles
    // issuing).  The three accumulator rows rotate by renaming (the loop body is instantiated three times).
    float wf[9][VEC];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) wf[t][e] = wv[t].get(e);
    float acc0[TW][VEC], acc1[TW][VEC], acc2[TW][VEC];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc0[t][e] = acc1[t][e] = acc2[t][e] = 0.f;
    int slot = 0;
    auto consume = [&](int q, float (&aA)[TW][VEC], float (&aB)[TW][VEC], float (&aC)[TW][VEC]) {
        const int rem = Q - 1 - q;  // input rows after this one
        // younger operations than row q's six loads: steady state = this wave's 4 stores of iteration q-PF, then
        // (6 loads + 4 stores) of each of the PF-1 iterations since; otherwise count only the rows issued after q
        // (waiting for more than necessary is always safe)
        if (q >= PF + 2 && rem >= PF - 1) dw_wait_vmcnt<TW + (PF - 1) * (NC + TW)>();
        else if (rem >= 2) dw_wait_vmcnt<2 * NC>();
        else if (rem == 1) dw_wait_vmcnt<NC>();
        else dw_wait_vmcnt<0>();
        Chunk<T> row[NCOL], av[TW];
#pragma unroll
        for (int j = 0; j < NCOL; ++j) row[j].v = *reinterpret_cast<const vec_t*>(wb + (slot * NC + j) * 1024 + lane * 16);
        if (ADD) {
#pragma unroll
            for (int t = 0; t < TW; ++t) av[t].v = *reinterpret_cast<const vec_t*>(wb + (slot * NC + NCOL + t) * 1024 + lane * 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot is read before the DMA below refills it
        if (q + PF < Q) issue_row(q + PF, slot);
        slot = slot + 1 == PF ? 0 : slot + 1;
        if (PRE) {
            const bool rok = (unsigned)(ho0 + (q - 1) * D) < (unsigned)P.H;
#pragma unroll
            for (int j = 0; j < NCOL; ++j) dw_pre_apply<T>(row[j], psc, psh, pslope, rok && cok[j]);
        }
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
            float xf[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) xf[e] = row[j].get(e);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int t = j - s;   // output column (of this thread's four) that input column j feeds through tap s
                if (t < 0 || t >= TW) continue;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    aC[t][e] = s == 0 ? xf[e] * wf[0][e] : fmaf(xf[e], wf[s][e], aC[t][e]);   // tap row 0 opens row q
                    aB[t][e] = fmaf(xf[e], wf[3 + s][e], aB[t][e]);
                    aA[t][e] = fmaf(xf[e], wf[6 + s][e], aA[t][e]);
                }
            }
        }
        if (q < 2) return;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
            yn + (long long)(ho0 + (q - 2) * D) * orow_bytes, 0, orow_bytes, 0x00020000);
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            Chunk<T> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o.set(e, ADD ? aA[t][e] + av[t].get(e) : aA[t][e]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o.v), ro, goff[t], 0, 0);
        }
    };
    for (int q = 0; q < Q; q += 3) {
        consume(q, acc0, acc1, acc2);
        if (q + 1 < Q) consume(q + 1, acc1, acc2, acc0);
        if (q + 2 < Q) consume(q + 2, acc2, acc0, acc1);
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

// Stride-2 data gradient in gather form.  With the "same" padding of SeparableConv2d_same (one pixel
// in front) dx[2ho + a][2wo + b] only receives
//   a = 0: tap row 1 of dy row ho;          a = 1: tap row 0 of dy row ho+1 and tap row 2 of dy row ho
// (same for columns), so a thread that holds dy[ho..ho+1][wo0..wo0+2] of one channel vector writes the
// 2 x 4 block dx[2ho..2ho+1][2wo0..2wo0+3] and slides down one dy row per step: dy is read once
// (plus a one-column halo), dx written once, no divisibility tests.
template <typename T>
__global__ __launch_bounds__(256) void dw_bwd_data_s2_kernel(DwParams P) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int TW = 2;
    typedef typename Elem<T>::vec_t vec_t;
    const unsigned cv = P.C / VEC;
    int band, xblk;
    dw_block_to_row(P.bx, band, xblk);
    const unsigned idx = xblk * 256u + threadIdx.x;
    if (idx >= (unsigned)P.items) return;
    const unsigned wq = idx / cv;
    const int c = (int)(idx - wq * cv) * VEC;
    const int wo0 = (int)wq * TW;
    const int n = band / P.bands, b = band - n * P.bands;
    const int ho0 = b * P.rb;
    const int ho1 = min(ho0 + P.rb, P.Ho);
    const unsigned grow_bytes = (unsigned)P.Wo * P.ldy * sizeof(T);
    const char* gn = reinterpret_cast<const char*>(P.x) + (long long)n * P.Ho * grow_bytes;  // P.x = dy
    int goff[TW + 1];
#pragma unroll
    for (int j = 0; j <= TW; ++j) goff[j] = ((wo0 + j) * P.ldy + c) * (int)sizeof(T);
    float wf[9][VEC];
    {
        const T* w = reinterpret_cast<const T*>(P.w) + c;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            Chunk<T> wv;
            wv.load(w + t * P.C);
#pragma unroll
            for (int e = 0; e < VEC; ++e) wf[t][e] = wv.get(e);
        }
    }
    auto load_row = [&](int ho, Chunk<T>(&dst)[TW + 1]) {
        const bool ok = (unsigned)ho < (unsigned)P.Ho;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(gn + (long long)(ok ? ho : 0) * grow_bytes), 0, ok ? grow_bytes : 0u, 0x00020000);
#pragma unroll
        for (int j = 0; j <= TW; ++j) dst[j].v = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[j], 0, 0));
    };
    Chunk<T> g0[TW + 1], g1[TW + 1];
    load_row(ho0, g0);
    T* dx = reinterpret_cast<T*>(P.y) + (((long long)n * P.H + 2 * ho0) * P.W + 2 * wo0) * P.ldx + c;  // P.y = dx
    for (int ho = ho0; ho < ho1; ++ho) {
        load_row(ho + 1, g1);
        const bool row1 = 2 * ho + 1 < P.H;
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            Chunk<T> o00, o01, o10, o11;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float a = g0[t].get(e), bq = g0[t + 1].get(e), cq = g1[t].get(e), dq = g1[t + 1].get(e);
                o00.set(e, a * wf[4][e]);
                o01.set(e, fmaf(bq, wf[3][e], a * wf[5][e]));
                o10.set(e, fmaf(cq, wf[1][e], a * wf[7][e]));
                o11.set(e, fmaf(dq, wf[0][e], fmaf(cq, wf[2][e], fmaf(bq, wf[6][e], a * wf[8][e]))));
            }
            const int iw = 2 * (wo0 + t);
            if (iw < P.W) {
                o00.store(dx + (long long)(2 * t) * P.ldx);
                if (row1) o10.store(dx + ((long long)P.W + 2 * t) * P.ldx);
            }
            if (iw + 1 < P.W) {
                o01.store(dx + (long long)(2 * t + 1) * P.ldx);
                if (row1) o11.store(dx + ((long long)P.W + 2 * t + 1) * P.ldx);
            }
        }
        dx += 2LL * P.W * P.ldx;
#pragma unroll
        for (int j = 0; j <= TW; ++j) g0[j] = g1[j];
    }
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
    int gx, bands;  // dw_bwd_weight_s1_kernel: channel blocks, row bands per image
    int wgroups, bands_per_phase;  // ... column groups per column phase, bands per row phase (dilation)
    const float* pre_scale;  // PRE = 1 (dw_bwd_weight_s1_kernel): x is a raw convolution output, see DwParams
    const float* pre_shift;
    int pre_act;
    int pre_ipg;
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

// Stride 1 / dilation 1 weight gradient with the register blocking of dw_s1_kernel: a thread
// owns TW = 4 output columns of one channel vector, walks RB output rows of one image with the
// 3-row input window in registers (x: 1.5 loads per output instead of 3; branch-free buffer loads,
// out-of-image = 0), then moves to its next column group.  Block = TX channel vectors x TY column
// lanes over one band of rows; partials are folded through LDS one tap at a time.
template <typename T, int PRE>
__global__ __launch_bounds__(256) void dw_bwd_weight_s1_kernel(DwWParams P) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int TW = 4, NCOL = TW + 2;
    typedef typename Elem<T>::vec_t vec_t;
    __shared__ float red[256 * VEC];
    const int lx = threadIdx.x & (P.tx - 1);
    const int ly = threadIdx.x >> P.log_tx;
    const int ty = 256 >> P.log_tx;
    int band, bx;
    dw_block_to_row(P.gx, band, bx);
    const int c = (bx * P.tx + lx) * VEC;
    const bool c_ok = c < P.C;
    float acc[9][VEC];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
    const int D = P.dil;  // dilation D: the D x D sub-lattices of the image, see dw_s1_kernel
    const int n = band / P.bands, b = band - n * P.bands;
    const int pr = b / P.bands_per_phase, bi = b - pr * P.bands_per_phase;
    const int ho0 = pr + bi * P.rows_per_block * D;
    const int ho1 = min(ho0 + P.rows_per_block * D, P.Ho);
    const unsigned xrow_bytes = (unsigned)P.W * P.ldx * sizeof(T);
    const unsigned grow_bytes = (unsigned)P.Wo * P.ldy * sizeof(T);
    const char* xn = reinterpret_cast<const char*>(P.x) + (long long)n * P.H * xrow_bytes;
    const char* gn = reinterpret_cast<const char*>(P.dy) + (long long)n * P.Ho * grow_bytes;
    const int nwq = D * P.wgroups;
    float psc[VEC], psh[VEC];
    const float pslope = act_max_slope(P.pre_act);
    if (PRE && c_ok) {
        const long long o = (long long)(n / P.pre_ipg) * P.C + c;
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(P.pre_scale + o + e);
            const f32x4 b = *reinterpret_cast<const f32x4*>(P.pre_shift + o + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                psc[e + k] = a[k];
                psh[e + k] = b[k];
            }
        }
    }
    if (c_ok) {
        for (int wq = ly; wq < nwq; wq += ty) {
            const int pc = wq / P.wgroups, gi = wq - pc * P.wgroups;
            const int wo0 = pc + gi * TW * D;
            int voff[NCOL], goff[TW];
#pragma unroll
            for (int j = 0; j < NCOL; ++j) voff[j] = ((wo0 + (j - 1) * D) * P.ldx + c) * (int)sizeof(T);
#pragma unroll
            for (int t = 0; t < TW; ++t) goff[t] = ((wo0 + t * D) * P.ldy + c) * (int)sizeof(T);
            auto load_row = [&](int ih, Chunk<T>(&dst)[NCOL]) {
                const bool ok = (unsigned)ih < (unsigned)P.H;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char*>(xn + (long long)(ok ? ih : 0) * xrow_bytes), 0, ok ? xrow_bytes : 0u, 0x00020000);
#pragma unroll
                for (int j = 0; j < NCOL; ++j)
                    dst[j].v = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j], 0, 0));
            };
            auto pre_row = [&](int ih, Chunk<T>(&row)[NCOL]) {
                const bool rok = (unsigned)ih < (unsigned)P.H;
#pragma unroll
                for (int j = 0; j < NCOL; ++j)
                    dw_pre_apply<T>(row[j], psc, psh, pslope, rok && (unsigned)(wo0 + (j - 1) * D) < (unsigned)P.W);
            };
            Chunk<T> r0[NCOL], r1[NCOL], r2[NCOL], nx[NCOL], gv[TW];
            load_row(ho0 - D, r0);
            load_row(ho0, r1);
            load_row(ho0 + D, r2);
            if (PRE) {
                pre_row(ho0 - D, r0);
                pre_row(ho0, r1);
                pre_row(ho0 + D, r2);
            }
            for (int ho = ho0; ho < ho1; ho += D) {
                const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char*>(gn + (long long)ho * grow_bytes), 0, grow_bytes, 0x00020000);
#pragma unroll
                for (int t = 0; t < TW; ++t)  // columns past Wo are out of range: zero gradient
                    gv[t].v = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rg, goff[t], 0, 0));
                if (ho + D < ho1) load_row(ho + 2 * D, nx);
#pragma unroll
                for (int t = 0; t < TW; ++t)
#pragma unroll
                    for (int s = 0; s < 3; ++s)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const float g = gv[t].get(e);
                            acc[0 + s][e] = fmaf(g, r0[t + s].get(e), acc[0 + s][e]);
                            acc[3 + s][e] = fmaf(g, r1[t + s].get(e), acc[3 + s][e]);
                            acc[6 + s][e] = fmaf(g, r2[t + s].get(e), acc[6 + s][e]);
                        }
                if (PRE && ho + D < ho1) pre_row(ho + 2 * D, nx);
#pragma unroll
                for (int j = 0; j < NCOL; ++j) {
                    r0[j] = r1[j];
                    r1[j] = r2[j];
                    r2[j] = nx[j];
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
            const int ch = bx * row_w + i;
            if (ch < P.C) atomicAdd(P.dw + (long long)t * P.C + ch, a);
        }
    }
}

// Stride-2 weight gradient: a thread owns TW = 2 output columns of one channel vector (input columns
// 2*wo0-1 .. 2*wo0+3) and walks the output rows of a band; input row 2*ho+1 is shared by output rows ho
// and ho+1, so every step loads two new input rows -- each x element is read once along rows, 1.25 times
// along columns (the generic path reads it up to 2.25 times through nine predicated loads per output).
template <typename T>
__global__ __launch_bounds__(256) void dw_bwd_weight_s2_kernel(DwWParams P) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int TW = 2, NCOL = 2 * TW + 1;
    typedef typename Elem<T>::vec_t vec_t;
    __shared__ float red[256 * VEC];
    const int lx = threadIdx.x & (P.tx - 1);
    const int ly = threadIdx.x >> P.log_tx;
    const int ty = 256 >> P.log_tx;
    int band, bx;
    dw_block_to_row(P.gx, band, bx);
    const int c = (bx * P.tx + lx) * VEC;
    const bool c_ok = c < P.C;
    float acc[9][VEC];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
    const int n = band / P.bands, b = band - n * P.bands;
    const int ho0 = b * P.rows_per_block;
    const int ho1 = min(ho0 + P.rows_per_block, P.Ho);
    const unsigned xrow_bytes = (unsigned)P.W * P.ldx * sizeof(T);
    const unsigned grow_bytes = (unsigned)P.Wo * P.ldy * sizeof(T);
    const char* xn = reinterpret_cast<const char*>(P.x) + (long long)n * P.H * xrow_bytes;
    const char* gn = reinterpret_cast<const char*>(P.dy) + (long long)n * P.Ho * grow_bytes;
    const int nwq = (P.Wo + TW - 1) / TW;
    if (c_ok) {
        for (int wq = ly; wq < nwq; wq += ty) {
            const int wo0 = wq * TW;
            int voff[NCOL], goff[TW];
#pragma unroll
            for (int j = 0; j < NCOL; ++j) voff[j] = ((2 * wo0 - 1 + j) * P.ldx + c) * (int)sizeof(T);
#pragma unroll
            for (int t = 0; t < TW; ++t) goff[t] = ((wo0 + t) * P.ldy + c) * (int)sizeof(T);
            auto load_row = [&](int ih, Chunk<T>(&dst)[NCOL]) {
                const bool ok = (unsigned)ih < (unsigned)P.H;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char*>(xn + (long long)(ok ? ih : 0) * xrow_bytes), 0, ok ? xrow_bytes : 0u, 0x00020000);
#pragma unroll
                for (int j = 0; j < NCOL; ++j)
                    dst[j].v = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j], 0, 0));
            };
            Chunk<T> r0[NCOL], r1[NCOL], r2[NCOL], gv[TW];
            load_row(2 * ho0 - 1, r0);
            for (int ho = ho0; ho < ho1; ++ho) {
                const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char*>(gn + (long long)ho * grow_bytes), 0, grow_bytes, 0x00020000);
#pragma unroll
                for (int t = 0; t < TW; ++t)  // columns past Wo are out of range: zero gradient
                    gv[t].v = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rg, goff[t], 0, 0));
                load_row(2 * ho, r1);
                load_row(2 * ho + 1, r2);
#pragma unroll
                for (int t = 0; t < TW; ++t)
#pragma unroll
                    for (int s = 0; s < 3; ++s)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const float g = gv[t].get(e);
                            acc[0 + s][e] = fmaf(g, r0[2 * t + s].get(e), acc[0 + s][e]);
                            acc[3 + s][e] = fmaf(g, r1[2 * t + s].get(e), acc[3 + s][e]);
                            acc[6 + s][e] = fmaf(g, r2[2 * t + s].get(e), acc[6 + s][e]);
                        }
#pragma unroll
                for (int j = 0; j < NCOL; ++j) r0[j] = r2[j];
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
            const int ch = bx * row_w + i;
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

template <typename T, int FLIP, int PRE, int ADD = 0>
void launch_dw_ring_t(const DwParams& P, unsigned blocks, hipStream_t st) {
    constexpr int RING = ADD ? DW_RING_LDS_ADD : DW_RING_LDS;
    static const bool once = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_ring_kernel<T, FLIP, PRE, ADD>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, RING + 32 * 1024) == hipSuccess;
    }();
    (void)once;
    const size_t lds = RING + (PRE && P.pre_sum ? (size_t)2 * P.C * sizeof(float) : 0);
    hipLaunchKernelGGL((dw_ring_kernel<T, FLIP, PRE, ADD>), dim3(blocks), dim3(256), lds, st, P);
}

// stride 1 / dilation 1 launcher shared by forward (flip 0) and data gradient (flip 1)
int launch_dw_s1(int dtype, DwParams P, int flip, hipStream_t st, const char* who) {
    static const int k_rb = getenv("BGAMD_DW_RB") ? atoi(getenv("BGAMD_DW_RB")) : 0;  // tuning knob (0: chosen below)
    static const bool ring = !(getenv("BGAMD_DW_RING") && atoi(getenv("BGAMD_DW_RING")) == 0);  // A/B switch
    const int cv = P.C / dtype_vec(dtype);
    BG_CHECK_ARG((long long)P.W * P.ldx * 4 < 0x7fffffffLL && (long long)P.Wo * P.ldy * 4 < 0x7fffffffLL,
                 "%s: image row too large", who);
    const int D = P.dil;
    const int vrows = (P.Ho + D - 1) / D, vcols = (P.Wo + D - 1) / D;  // rows / columns of one sub-lattice
    P.wgroups = (vcols + 3) / 4;
    P.items = D * P.wgroups * cv;
    P.bx = (P.items + 255) / 256;
    int rb = k_rb > 0 ? k_rb : 4;
    if (ring && k_rb <= 0) {
        // rows per band: ONE round of at most one workgroup per CU wherever the launch allows it -- measured over band
        // lengths 3..144 on the step's shapes (scripts/gpu_rb.sh), the fastest launch was always the one with 190-256
        // workgroups (8 x 72 x 48 x 728: 23.7 us at 240 workgroups against 26.5 at 480 and 25.9 at 720; 8 x 144 x 96 x 728:
        // 67.9 at 216 against 78.2), i.e. long bands that keep the prefetch ring in its steady state and no second
        // round; tensors too large for that (the entry flow: > 256 workgroups even with one band per image) are
        // insensitive and keep short bands
        const long long per_band = (long long)P.N * D * P.bx;
        if (per_band <= 256) {
            const int nbands = (int)(256 / per_band);
            rb = (vrows + nbands - 1) / nbands;
        } else {
            rb = 6;
        }
    }
    P.rb = rb < vrows ? rb : vrows;
    P.bands_per_phase = (vrows + P.rb - 1) / P.rb;
    P.bands = D * P.bands_per_phase;
    const long long blocks = (long long)P.N * P.bands * P.bx;
    BG_CHECK_ARG(blocks <= 0x7fffffffLL, "%s: grid too large", who);
    const unsigned nb = (unsigned)blocks;
    if (ring) {
        if (flip && P.add) BG_DISPATCH_DTYPE(dtype, T, (launch_dw_ring_t<T, 1, 0, 1>(P, nb, st)));
        else if (flip) BG_DISPATCH_DTYPE(dtype, T, (launch_dw_ring_t<T, 1, 0>(P, nb, st)));
        else if (P.pre_scale || P.pre_sum) BG_DISPATCH_DTYPE(dtype, T, (launch_dw_ring_t<T, 0, 1>(P, nb, st)));
        else BG_DISPATCH_DTYPE(dtype, T, (launch_dw_ring_t<T, 0, 0>(P, nb, st)));
    } else {
        if (flip) BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((dw_s1_kernel<T, 1, 0>), dim3(nb), dim3(256), 0, st, P));
        else if (P.pre_scale) BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((dw_s1_kernel<T, 0, 1>), dim3(nb), dim3(256), 0, st, P));
        else BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((dw_s1_kernel<T, 0, 0>), dim3(nb), dim3(256), 0, st, P));
    }
    BG_CHECK_LAUNCH(who);
    return BG_OK;
}

}  // namespace

namespace {
struct DwPre {
    const float* scale;
    const float* shift;
    int groups, act;
};
int check_pre(const bg_dwconv_desc* d, const DwPre& pre, const char* who) {
    BG_CHECK_ARG(pre.scale && pre.shift && aligned16(pre.scale) && aligned16(pre.shift), "%s: null/unaligned scale/shift", who);
    BG_CHECK_ARG(pre.groups >= 1 && d->N % pre.groups == 0, "%s: batch of %d does not split into %d groups", who, d->N, pre.groups);
    BG_CHECK_ARG(pre.act >= 0 && pre.act <= 2, "%s: bad activation code", who);
    BG_CHECK_ARG(d->stride == 1 && (d->dil == 1 || d->dil == 2), "%s: stride 1 with dilation 1 or 2 only", who);
    BG_CHECK_ARG(d->C % 4 == 0, "%s: C must be a multiple of 4", who);
    return BG_OK;
}
int dw_fwd_impl(const bg_dwconv_desc* d, const void* x, const DwPre* pre, const void* w, void* y, void* stream);
int dw_bwd_weight_impl(const bg_dwconv_desc* d, const void* x, const DwPre* pre, const void* dy, float* dw, void* stream);
}  // namespace

extern "C" int bg_dwconv3x3_fwd(const bg_dwconv_desc* d, const void* x, const void* w, void* y, void* stream) {
    return dw_fwd_impl(d, x, nullptr, w, y, stream);
}
extern "C" int bg_dwconv3x3_fwd_pre(const bg_dwconv_desc* d, const void* x, const float* scale, const float* shift,
                                    int32_t groups, int32_t act, const void* w, void* y, void* stream) {
    int rc = check_dw(d, "bg_dwconv3x3_fwd_pre");
    if (rc) return rc;
    const DwPre pre{scale, shift, groups, act};
    rc = check_pre(d, pre, "bg_dwconv3x3_fwd_pre");
    if (rc) return rc;
    return dw_fwd_impl(d, x, &pre, w, y, stream);
}
extern "C" int bg_dwconv3x3_fwd_pre_stats(const bg_dwconv_desc* d, const void* x, const double* sum, const double* sumsq,
                                          const float* gamma, const float* beta, float eps, float momentum,
                                          float* running_mean, float* running_var, float* mean, float* rstd,
                                          float* scale, float* shift, int32_t groups, int32_t act, const void* w, void* y,
                                          void* stream) {
    int rc = check_dw(d, "bg_dwconv3x3_fwd_pre_stats");
    if (rc) return rc;
    BG_CHECK_ARG(x && w && y && aligned16(x) && aligned16(w) && aligned16(y), "bg_dwconv3x3_fwd_pre_stats: null/unaligned pointer");
    BG_CHECK_ARG(sum && sumsq && mean && rstd && scale && shift, "bg_dwconv3x3_fwd_pre_stats: null statistics pointer");
    BG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bg_dwconv3x3_fwd_pre_stats: running statistics come in pairs");
    BG_CHECK_ARG(groups >= 1 && d->N % groups == 0 && act >= 0 && act <= 2, "bg_dwconv3x3_fwd_pre_stats: bad groups / act");
    BG_CHECK_ARG(d->stride == 1 && (d->dil == 1 || d->dil == 2), "bg_dwconv3x3_fwd_pre_stats: stride 1 with dilation 1 or 2 only");
    BG_CHECK_ARG(d->C % 4 == 0 && (size_t)2 * d->C * sizeof(float) <= 32 * 1024, "bg_dwconv3x3_fwd_pre_stats: C = %d not supported", d->C);
    static const bool ring = !(getenv("BGAMD_DW_RING") && atoi(getenv("BGAMD_DW_RING")) == 0);
    BG_CHECK_ARG(ring, "bg_dwconv3x3_fwd_pre_stats needs the ring kernel (BGAMD_DW_RING=0 is set)");
    DwParams P{x, w, y, d->N, d->H, d->W, d->C, d->Ho, d->Wo, d->stride, d->dil, d->ldx, d->ldy, 0, 0};
    P.pre_act = act;
    P.pre_ipg = d->N / groups;
    P.pre_sum = sum; P.pre_sumsq = sumsq; P.pre_gamma = gamma; P.pre_beta = beta; P.pre_eps = eps; P.pre_mom = momentum;
    P.pre_rmean = running_mean; P.pre_rvar = running_var; P.pre_mean_out = mean; P.pre_rstd_out = rstd;
    P.pre_scale_out = scale; P.pre_shift_out = shift;
    P.pre_rpg = (long long)P.pre_ipg * d->H * d->W;
    P.pre_groups = groups;
    return launch_dw_s1(d->dtype, P, 0, (hipStream_t)stream, "dw_ring_kernel(pre, stats)");
}

extern "C" int bg_dwconv3x3_bwd_data_add(const bg_dwconv_desc* d, const void* dy, const void* w, const void* add,
                                         int32_t ldadd, void* dx, void* stream) {
    int rc = check_dw(d, "bg_dwconv3x3_bwd_data_add");
    if (rc) return rc;
    BG_CHECK_ARG(dy && w && dx && add && aligned16(dy) && aligned16(w) && aligned16(dx) && aligned16(add),
                 "bg_dwconv3x3_bwd_data_add: null/unaligned pointer");
    BG_CHECK_ARG(d->stride == 1 && (d->dil == 1 || d->dil == 2), "bg_dwconv3x3_bwd_data_add: stride 1 with dilation 1 or 2 only");
    BG_CHECK_ARG(ldadd >= d->C && ldadd % dtype_vec(d->dtype) == 0, "bg_dwconv3x3_bwd_data_add: bad pixel stride of the addend");
    BG_CHECK_ARG((long long)d->W * ldadd * 4 < 0x7fffffffLL, "bg_dwconv3x3_bwd_data_add: image row too large");
    static const bool ring = !(getenv("BGAMD_DW_RING") && atoi(getenv("BGAMD_DW_RING")) == 0);
    BG_CHECK_ARG(ring, "bg_dwconv3x3_bwd_data_add needs the ring kernel (BGAMD_DW_RING=0 is set)");
    // roles as in bg_dwconv3x3_bwd_data: dy plays the input (pixel stride ldy), dx the output (ldx); both H x W
    DwParams Q{dy, w, dx, d->N, d->H, d->W, d->C, d->H, d->W, 1, d->dil, d->ldy, d->ldx, 0, 0};
    Q.add = add;
    Q.ldadd = ldadd;
    return launch_dw_s1(d->dtype, Q, 1, (hipStream_t)stream, "dw_ring_kernel(flip, add)");
}

extern "C" int bg_dwconv3x3_bwd_weight(const bg_dwconv_desc* d, const void* x, const void* dy, float* dw,
                                       void* stream) {
    return dw_bwd_weight_impl(d, x, nullptr, dy, dw, stream);
}
extern "C" int bg_dwconv3x3_bwd_weight_pre(const bg_dwconv_desc* d, const void* x, const float* scale,
                                           const float* shift, int32_t groups, int32_t act, const void* dy, float* dw,
                                           void* stream) {
    int rc = check_dw(d, "bg_dwconv3x3_bwd_weight_pre");
    if (rc) return rc;
    const DwPre pre{scale, shift, groups, act};
    rc = check_pre(d, pre, "bg_dwconv3x3_bwd_weight_pre");
    if (rc) return rc;
    return dw_bwd_weight_impl(d, x, &pre, dy, dw, stream);
}

namespace {
int dw_fwd_impl(const bg_dwconv_desc* d, const void* x, const DwPre* pre, const void* w, void* y, void* stream) {
    int rc = check_dw(d, "bg_dwconv3x3_fwd");
    if (rc) return rc;
    BG_CHECK_ARG(x && w && y && aligned16(x) && aligned16(w) && aligned16(y), "bg_dwconv3x3_fwd: null/unaligned pointer");
    DwParams P{x, w, y, d->N, d->H, d->W, d->C, d->Ho, d->Wo, d->stride, d->dil, d->ldx, d->ldy, 0, 0};
    if (pre) {
        P.pre_scale = pre->scale;
        P.pre_shift = pre->shift;
        P.pre_act = pre->act;
        P.pre_ipg = d->N / pre->groups;
        return launch_dw_s1(d->dtype, P, 0, (hipStream_t)stream, "dw_s1_kernel(pre)");
    }
    const long long rows = (long long)d->N * d->Ho;
    hipStream_t st = (hipStream_t)stream;
    const int cv = d->C / dtype_vec(d->dtype);
    const int sd = d->stride * 10 + d->dil;
    static const bool old11 = getenv("BGAMD_DW_OLD") != nullptr;  // A/B switch
    static const bool old_sd = getenv("BGAMD_DW_SD_OLD") != nullptr;  // A/B switch: stride-2 / dilation-2 on the generic kernels
    if ((sd == 11 || (sd == 12 && !old_sd)) && !old11) return launch_dw_s1(d->dtype, P, 0, st, "dw_s1_kernel");
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
}  // namespace

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
        static const bool old11 = getenv("BGAMD_DW_OLD") != nullptr;  // A/B switch
        static const bool old_sd = getenv("BGAMD_DW_SD_OLD") != nullptr;
        if (!old11 && (d->dil == 1 || !old_sd)) return launch_dw_s1(d->dtype, Q, 1, st, "dw_s1_kernel(flip)");
        Q.items = ((d->W + 3) / 4) * cv;
        Q.bx = (Q.items + 255) / 256;
        BG_CHECK_ARG(rows * Q.bx <= 0x7fffffffLL, "bg_dwconv3x3_bwd_data: grid too large");
        dim3 grid((unsigned)(rows * Q.bx));
        if (d->dil == 1) BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw_fwd_tw_kernel<T, 1, 1, 1>), grid, dim3(256), 0, st, Q));
        else BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw_fwd_tw_kernel<T, 1, 2, 1>), grid, dim3(256), 0, st, Q));
        BG_CHECK_LAUNCH("dw_fwd_tw_kernel(flip)");
        return BG_OK;
    }
    static const bool old_s2 = getenv("BGAMD_DW_OLD") != nullptr || getenv("BGAMD_DW_SD_OLD") != nullptr;  // A/B switch
    if (d->stride == 2 && d->dil == 1 && !old_s2) {
        BG_CHECK_ARG((long long)d->Wo * d->ldy * 4 < 0x7fffffffLL, "bg_dwconv3x3_bwd_data: image row too large");
        P.rb = 4 < d->Ho ? 4 : d->Ho;
        P.bands = (d->Ho + P.rb - 1) / P.rb;
        P.items = ((d->Wo + 1) / 2) * cv;
        P.bx = (P.items + 255) / 256;
        const long long blocks = (long long)d->N * P.bands * P.bx;
        BG_CHECK_ARG(blocks <= 0x7fffffffLL, "bg_dwconv3x3_bwd_data: grid too large");
        BG_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw_bwd_data_s2_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, P));
        BG_CHECK_LAUNCH("dw_bwd_data_s2_kernel");
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

namespace {
int dw_bwd_weight_impl(const bg_dwconv_desc* d, const void* x, const DwPre* pre, const void* dy, float* dw, void* stream) {
    int rc = check_dw(d, "bg_dwconv3x3_bwd_weight");
    if (rc) return rc;
    BG_CHECK_ARG(x && dy && dw && aligned16(x) && aligned16(dy), "bg_dwconv3x3_bwd_weight: null/unaligned pointer");
    DwWParams P{x, dy, dw, d->N, d->H, d->W, d->C, d->Ho, d->Wo, d->stride, d->dil, d->ldx, d->ldy, 0, 0, 0, 0};
    if (pre) {
        P.pre_scale = pre->scale;
        P.pre_shift = pre->shift;
        P.pre_act = pre->act;
        P.pre_ipg = d->N / pre->groups;
    }
    const int cv = d->C / dtype_vec(d->dtype);
    int best = 16, best_pad = 1 << 30;
    for (int tx = 16; tx <= 64; tx *= 2) {
        const int pad = (cv + tx - 1) / tx * tx;
        if (pad <= best_pad) { best_pad = pad; best = tx; }
    }
    P.tx = best;
    P.log_tx = best == 16 ? 4 : (best == 32 ? 5 : 6);
    const int gx = (cv + best - 1) / best;
    hipStream_t st = (hipStream_t)stream;
    static const bool old11 = getenv("BGAMD_DW_OLD") != nullptr;  // A/B switch
    static const bool old_sd = getenv("BGAMD_DW_SD_OLD") != nullptr;
    if (d->stride == 1 && (d->dil == 1 || (d->dil == 2 && !old_sd)) && !old11) {
        static const int k_rb_env = getenv("BGAMD_DWW_RB") ? atoi(getenv("BGAMD_DWW_RB")) : 0;  // tuning knob
        // rows per band (scripts/gpu_rbw.sh, band lengths 4..72): 4 on the 72-row maps of the middle flow, 8-9 on
        // the taller maps of the entry flow (8 x 144 x 96 x 728: 80.6 -> 73.5 us, 8 x 288 x 192 x 256: 110 -> 98, 8 x 576 x 384 x 128:
        // 189 -> 170) -- a thread walks its band once per column group, so band length trades launch width for fewer
        // window refills
        const int vrows_ = (d->Ho + d->dil - 1) / d->dil;
        const int k_rb = k_rb_env > 0 ? k_rb_env : (vrows_ >= 256 ? 9 : (vrows_ >= 128 ? 8 : 4));
        BG_CHECK_ARG((long long)d->W * d->ldx * 4 < 0x7fffffffLL && (long long)d->Wo * d->ldy * 4 < 0x7fffffffLL,
                     "bg_dwconv3x3_bwd_weight: image row too large");
        const int D = d->dil;
        const int vrows = (d->Ho + D - 1) / D, vcols = (d->Wo + D - 1) / D;
        P.rows_per_block = k_rb < vrows ? k_rb : vrows;
        P.bands_per_phase = (vrows + P.rows_per_block - 1) / P.rows_per_block;
        P.bands = D * P.bands_per_phase;
        P.wgroups = (vcols + 3) / 4;
        P.gx = gx;
        const long long blocks = (long long)gx * d->N * P.bands;
        BG_CHECK_ARG(blocks <= 0x7fffffffLL, "bg_dwconv3x3_bwd_weight: grid too large");
        if (P.pre_scale)
            BG_DISPATCH_DTYPE(d->dtype, T,
                              hipLaunchKernelGGL((dw_bwd_weight_s1_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, st, P));
        else
            BG_DISPATCH_DTYPE(d->dtype, T,
                              hipLaunchKernelGGL((dw_bwd_weight_s1_kernel<T, 0>), dim3((unsigned)blocks), dim3(256), 0, st, P));
        BG_CHECK_LAUNCH("dw_bwd_weight_s1_kernel");
        return BG_OK;
    }
    BG_CHECK_ARG(!P.pre_scale, "bg_dwconv3x3_bwd_weight_pre: stride 1 with dilation 1 or 2 only");
    if (d->stride == 2 && d->dil == 1 && !old11 && !old_sd) {
        BG_CHECK_ARG((long long)d->W * d->ldx * 4 < 0x7fffffffLL && (long long)d->Wo * d->ldy * 4 < 0x7fffffffLL,
                     "bg_dwconv3x3_bwd_weight: image row too large");
        P.rows_per_block = 4 < d->Ho ? 4 : d->Ho;
        P.bands = (d->Ho + P.rows_per_block - 1) / P.rows_per_block;
        P.gx = gx;
        const long long blocks = (long long)gx * d->N * P.bands;
        BG_CHECK_ARG(blocks <= 0x7fffffffLL, "bg_dwconv3x3_bwd_weight: grid too large");
        BG_DISPATCH_DTYPE(d->dtype, T,
                          hipLaunchKernelGGL((dw_bwd_weight_s2_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, P));
        BG_CHECK_LAUNCH("dw_bwd_weight_s2_kernel");
        return BG_OK;
    }
    P.rows_total = d->N * d->Ho;
    // ~1536 blocks in total; each block at least one output row
    int gy = 1536 / gx;
    if (gy < 1) gy = 1;
    if (gy > P.rows_total) gy = P.rows_total;
    P.rows_per_block = (P.rows_total + gy - 1) / gy;
    gy = (P.rows_total + P.rows_per_block - 1) / P.rows_per_block;
    if (d->stride == 1 && d->dil == 1)
        BG_DISPATCH_DTYPE(d->dtype, T,
                          hipLaunchKernelGGL((dw_bwd_weight_kernel<T, 1>), dim3(gx, (unsigned)gy), dim3(256), 0, st, P));
    else
        BG_DISPATCH_DTYPE(d->dtype, T,
                          hipLaunchKernelGGL((dw_bwd_weight_kernel<T, 0>), dim3(gx, (unsigned)gy), dim3(256), 0, st, P));
    BG_CHECK_LAUNCH("dw_bwd_weight_kernel");
    return BG_OK;
}
}  // namespace
