// Normalisation + residual + LeakyReLU, forward and backward, NHWC.
//
// All kernels here are HBM-bound.  Element-wise passes move one 16-byte vector
// per lane with lanes along channels (fully coalesced); per-channel reductions
// keep fp32 partials in registers over a strip of rows, fold the row lanes of a
// wave with shuffles and finish with fp64 atomics (so var = E[x^2]-E[x]^2 is
// evaluated in double and the result does not depend on how rows were split).
#include "common.h"
#include <initializer_list>

namespace {

// ----------------------------------------------------------------- tiling ----
// A [rows, C] activation is walked by 256-thread blocks shaped TX (channel
// vectors, lanes along channels -> 16 B per lane, contiguous) x TY (rows).
// grid: x = channel-vector blocks, y = row chunks inside a group, z = group.
// No integer division anywhere on the per-element path.
struct Tiling {
    int tx, ty;          // channel vectors per block row, block rows (threads per block = tx * ty <= 256)
    int gx;              // blocks along channels
    int rows_per_block;  // rows walked by one block
    int gy;
};

inline int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// The pixel stride all operands of a launch share (0 if they differ or one is absent with another stride).
inline int common_ld(std::initializer_list<int> lds) {
    int ld = 0;
    for (int v : lds) {
        if (v <= 0) continue;
        if (ld && v != ld) return 0;
        ld = v;
    }
    return ld;
}

// ld_common > 0: every operand is a whole-row tensor of that pixel stride (no channel slices of wider buffers), so
// consecutive rows are consecutive in memory.  Then a block row is the WHOLE pixel row (tx = ld / VEC chunks, the pad
// chunk's lanes idle) and the block's ty rows are one contiguous span: every wave load / store is 1 KiB of consecutive
// bytes, like a plain copy.  With the power-of-two tiling below a wave touched four 256-byte pieces 1 472 bytes apart;
// measured on the middle flow's 40 MB tensors a copy ran 1.5x faster than these kernels (scripts/bench_ew.py).
inline Tiling make_tiling(int dtype, int C, long long rows_per_group, int groups, int min_rows_per_thread,
                          int target_blocks, int ld_common = 0) {
    const int vec = dtype_vec(dtype);
    const int cv = C / vec;
    static const int row_mode = env_int("BGAMD_EW_ROWMODE", 1);
    Tiling t;
    const int cvl = ld_common / vec;
    if (row_mode && ld_common > 0 && ld_common % vec == 0 && cvl > 16 && cvl <= 256 && ld_common - C < 64) {
        t.tx = cvl;
        t.ty = 256 / cvl;
        t.gx = 1;
    } else {
        int best = 16, best_pad = 1 << 30;
        for (int tx = 16; tx <= 64; tx *= 2) {
            const int pad = (cv + tx - 1) / tx * tx;
            if (pad <= best_pad) { best_pad = pad; best = tx; }
        }
        t.tx = best;
        t.ty = 256 / best;
        t.gx = (cv + best - 1) / best;
    }
    const int ty = t.ty;
    long long want = target_blocks / ((long long)t.gx * groups);
    if (want >= 8) want = want / 8 * 8;  // the grid is gx * roundup8(gy) blocks (BG_BLOCK_COORDS): stay within the target
    if (want < 1) want = 1;
    long long rpb = (rows_per_group + want - 1) / want;
    const long long min_rpb = (long long)ty * min_rows_per_thread;
    if (rpb < min_rpb) rpb = min_rpb;
    if (rpb > (1 << 30)) rpb = 1 << 30;
    t.rows_per_block = (int)rpb;
    long long gy = (rows_per_group + rpb - 1) / rpb;
    if (gy > 65535) {  // keep the grid legal for enormous inputs
        gy = 65535;
        t.rows_per_block = (int)((rows_per_group + gy - 1) / gy);
        gy = (rows_per_group + t.rows_per_block - 1) / t.rows_per_block;
    }
    t.gy = (int)gy;
    return t;
}

// Block -> (channel block bx, row block by).  The gx channel blocks of one row range touch the same
// 128-byte lines whenever the row pitch is not a multiple of 128 B (C = 728): consecutive block ids go
// to different XCDs, so the ids are dealt such that those gx blocks land on ONE XCD (one L2 fetches /
// merges the shared lines).  The grid is gx * roundup8(gy) blocks; surplus blocks exit.
#define BG_BLOCK_COORDS(P, bx, by)                         \
    int bx, by;                                            \
    {                                                      \
        const int id_ = blockIdx.x, slot_ = id_ >> 3;      \
        const int q_ = slot_ / (P).gx;                     \
        bx = slot_ - q_ * (P).gx;                          \
        by = q_ * 8 + (id_ & 7);                           \
        if (by >= (P).gy) return;                          \
    }
inline unsigned xcd_grid(int gx, int gy) { return (unsigned)(gx * ((gy + 7) / 8 * 8)); }

// scale/shift of the normalisation, written ONCE so that the backward kernels that recompute the
// pre-activation sign (y not re-read) evaluate bit-identical arithmetic to the forward pass
__device__ __forceinline__ void norm_affine(float gamma, float beta, float mean, float rstd, float& sc, float& sh) {
    sc = gamma * rstd;
    sh = fmaf(-(mean * gamma), rstd, beta);
}

// ------------------------------------------------------------ column reduce --
enum { RED_STATS = 0, RED_BWD = 1, RED_COLSUM = 2 };
constexpr int RED_U = 4;  // rows in flight per thread and operand

struct RedParams {
    const void* a;  // x (stats, colsum) | dy (bwd)
    const void* b;  // y (bwd)
    const void* c;  // x (bwd)
    int lda, ldb, ldc;
    const float* mean;
    const float* rstd;
    const float* gamma;  // bwd with b == NULL: LeakyReLU slope from the recomputed pre-activation
    const float* beta;
    int act;
    long long rows_per_group;
    int rows_per_block;
    int C;
    int tx, ty;
    double* o1;
    double* o2;
    float* of;  // colsum output
    float scale;
    int gx, gy;
};

// SIGN (RED_BWD only): 0 no activation, 1 branch from the stored output b, 2 recomputed from c*scale+shift;
// HASX (RED_BWD only): the g*xhat statistic is wanted (c given).  Compile-time: see norm_act_fwd_kernel.
template <typename T, int MODE, int SIGN, bool HASX>
__global__ __launch_bounds__(256) void colreduce_kernel(RedParams P) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int NS = (MODE == RED_COLSUM) ? 1 : 2;
    __shared__ float red[256 * VEC * NS];
    const int ly = (int)threadIdx.x / P.tx;
    const int lx = (int)threadIdx.x - ly * P.tx;
    const int ty = P.ty;
    BG_BLOCK_COORDS(P, bx, by);
    const int c = (bx * P.tx + lx) * VEC;
    const bool c_ok = c < P.C;
    const int g = blockIdx.z;
    const long long r0 = (long long)by * P.rows_per_block;
    long long r1 = r0 + P.rows_per_block;
    if (r1 > P.rows_per_group) r1 = P.rows_per_group;
    const long long gbase = (long long)g * P.rows_per_group;

    float s1[VEC], s2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s1[e] = s2[e] = 0.f;
    float mu[VEC], rs[VEC], sc[VEC], sh[VEC];
    if (MODE == RED_BWD) {
        // per-channel statistics once per block, lanes along channels, through LDS (see norm_act_fwd_kernel); the
        // buffer is the reduction scratch, free until the row loop is over
        const int row_w = P.tx * VEC;
        for (int ch = threadIdx.x; ch < row_w; ch += (int)blockDim.x) {
            const int cg = bx * row_w + ch;
            float m = 0.f, r = 1.f, a = 1.f, b = 0.f;
            if (cg < P.C) {
                m = P.mean ? P.mean[(long long)g * P.C + cg] : 0.f;
                r = P.rstd ? P.rstd[(long long)g * P.C + cg] : 1.f;
                norm_affine(P.gamma ? P.gamma[cg] : 1.f, P.beta ? P.beta[cg] : 0.f, m, r, a, b);
            }
            const int ti = (ch % VEC) * P.tx + ch / VEC;   // element-major: conflict-free reads, see norm_act_fwd_kernel
            red[ti] = m;
            red[row_w + ti] = r;
            red[2 * row_w + ti] = a;
            red[3 * row_w + ti] = b;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            mu[e] = red[e * P.tx + lx];
            rs[e] = red[row_w + e * P.tx + lx];
            sc[e] = red[2 * row_w + e * P.tx + lx];
            sh[e] = red[3 * row_w + e * P.tx + lx];
        }
        __syncthreads();
    }
    if (c_ok) {
        const T* a = reinterpret_cast<const T*>(P.a) + (gbase + r0 + ly) * P.lda + c;
        const T* b = P.b ? reinterpret_cast<const T*>(P.b) + (gbase + r0 + ly) * P.ldb + c : nullptr;
        const T* cc = P.c ? reinterpret_cast<const T*>(P.c) + (gbase + r0 + ly) * P.ldc + c : nullptr;
        const long long sa = (long long)ty * P.lda, sb = (long long)ty * P.ldb, s_c = (long long)ty * P.ldc;
        constexpr bool ld_b = MODE == RED_BWD && SIGN == 1, ld_c = MODE == RED_BWD && (HASX || SIGN == 2);
        const float slope = act_slope(P.act);
        auto row = [&](const Chunk<T>& va, const Chunk<T>& vy, const Chunk<T>& vx) {
            if (MODE == RED_STATS) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float v = va.get(e);
                    s1[e] += v;
                    s2[e] = fmaf(v, v, s2[e]);
                }
            } else if (MODE == RED_COLSUM) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) s1[e] += va.get(e);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float gg = va.get(e);
                    if (SIGN == 1) gg *= vy.get(e) > 0.f ? 1.f : slope;
                    if (SIGN == 2) gg *= fmaf(vx.get(e), sc[e], sh[e]) > 0.f ? 1.f : slope;
                    s1[e] += gg;
                    if (HASX) s2[e] = fmaf(gg, (vx.get(e) - mu[e]) * rs[e], s2[e]);
                }
            }
        };
        auto load = [&](Chunk<T>* va, Chunk<T>* vy, Chunk<T>* vx) {
#pragma unroll
            for (int u = 0; u < RED_U; ++u) {
                va[u].load(a + u * sa);
                if (ld_b) vy[u].load(b + u * sb);
                if (ld_c) vx[u].load(cc + u * s_c);
            }
        };
        // software-pipelined batches of RED_U rows per operand (the compiler's own unrolling of the one-row loop waited
        // for each row's loads before issuing the next row's)
        long long r = r0 + ly;
        const long long step = RED_U * (long long)ty;
        Chunk<T> va[RED_U], vy[RED_U], vx[RED_U];
        bool have = r + step - ty < r1;
        if (have) load(va, vy, vx);
        while (have) {
            Chunk<T> ca[RED_U], cy[RED_U], cx[RED_U];
#pragma unroll
            for (int u = 0; u < RED_U; ++u) {
                ca[u] = va[u];
                if (ld_b) cy[u] = vy[u];
                if (ld_c) cx[u] = vx[u];
            }
            r += step;
            a += RED_U * sa;
            if (ld_b) b += RED_U * sb;
            if (ld_c) cc += RED_U * s_c;
            have = r + step - ty < r1;
            if (have) load(va, vy, vx);
#pragma unroll
            for (int u = 0; u < RED_U; ++u) row(ca[u], cy[u], cx[u]);
        }
        for (; r < r1; r += ty) {
            Chunk<T> v1, v2, v3;
            v1.load(a);
            a += sa;
            if (ld_b) { v2.load(b); b += sb; }
            if (ld_c) { v3.load(cc); cc += s_c; }
            row(v1, v2, v3);
        }
    }
    // block reduction over the TY row lanes through LDS, then ONE atomic per (channel, statistic)
    // layout: red[stat][ly][lx*VEC + e]
    const int row_w = P.tx * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        red[(0 * ty + ly) * row_w + lx * VEC + e] = s1[e];
        if (NS == 2) red[(1 * ty + ly) * row_w + lx * VEC + e] = s2[e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NS * row_w; i += (int)blockDim.x) {
        const int st = i / row_w;  // 0 or 1 (uniform per 256-thread pass when row_w >= 256, cheap otherwise)
        const int col = i - st * row_w;
        float acc = 0.f;
        for (int y = 0; y < ty; ++y) acc += red[(st * ty + y) * row_w + col];
        const int ch = bx * row_w + col;
        if (ch < P.C) {
            const long long o = (long long)g * P.C + ch;
            if (MODE == RED_COLSUM) atomicAdd(P.of + o, acc * P.scale);
            else if (st == 0) atomicAdd(P.o1 + o, (double)acc);
            else if (P.o2) atomicAdd(P.o2 + o, (double)acc);
        }
    }
}

template <int MODE>
int launch_colreduce(int dtype, RedParams P, int groups, hipStream_t st, const char* who) {
    // 256 blocks = one round of one block per CU: 5-15 % faster than 1 024 on every shape of the step (scripts/gpu_ewb.sh)
    static const int k_rows = env_int("BGAMD_RED_ROWS", 16), k_blocks = env_int("BGAMD_RED_BLOCKS", 256);  // tuning knobs
    // the power-of-two channel tiling only: with whole rows per block every block would fold and add ALL channels
    // (measured: 31 -> 52 us on the 40 MB tensors)
    const Tiling t = make_tiling(dtype, P.C, P.rows_per_group, groups, k_rows, k_blocks);
    P.tx = t.tx; P.ty = t.ty; P.rows_per_block = t.rows_per_block; P.gx = t.gx; P.gy = t.gy;
    BG_CHECK_ARG(groups <= 65535, "%s: too many groups", who);
    dim3 grid(xcd_grid(t.gx, t.gy), 1, (unsigned)groups);
    const dim3 block(t.tx * t.ty);
    if (MODE != RED_BWD) {
        BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((colreduce_kernel<T, MODE, 0, false>), grid, block, 0, st, P));
    } else {
        const int sign = !P.act ? 0 : (P.b ? 1 : 2);
#define BG_RED_BWD(S, X) BG_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((colreduce_kernel<T, RED_BWD, S, X>), grid, block, 0, st, P))
        if (P.c) {
            if (sign == 0) BG_RED_BWD(0, true);
            else if (sign == 1) BG_RED_BWD(1, true);
            else BG_RED_BWD(2, true);
        } else {
            BG_CHECK_ARG(sign != 2, "%s: the recomputed activation branch needs x", who);
            if (sign == 0) BG_RED_BWD(0, false);
            else BG_RED_BWD(1, false);
        }
#undef BG_RED_BWD
    }
    BG_CHECK_LAUNCH(who);
    return BG_OK;
}

// ------------------------------------------------------------- finalize ----
__global__ void norm_finalize_kernel(const double* sum, const double* sumsq, long long rpg, int groups, int C,
                                     const float* gamma, const float* beta, float eps, float momentum, float* rmean,
                                     float* rvar, float* mean, float* rstd, float* scale, float* shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * C) return;
    const int c = i % C;
    const double n = (double)rpg;
    const double m = sum[i] / n;
    double var = sumsq[i] / n - m * m;
    if (var < 0.0 || rpg == 1) var = 0.0;
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    const float gm = gamma ? gamma[c] : 1.f;
    const float bt = beta ? beta[c] : 0.f;
    mean[i] = (float)m;
    rstd[i] = r;
    scale[i] = gm * r;
    shift[i] = bt - (float)m * gm * r;
    if (rmean && groups == 1) {
        // nn.BatchNorm2d: running = (1-mom)*running + mom*batch, unbiased variance
        const double unb = rpg > 1 ? var * n / (n - 1.0) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

// The fused finalize of norm_act_fwd_kernel as a launch of its own, for consumers that apply the affine themselves
// (bg_dwconv3x3_fwd_pre): the SAME arithmetic (fp64 mean / variance, fp32 rsqrt, norm_affine), so that the backward
// kernels' recomputed activation branch agrees with what the consumer computed.  One thread per channel; statistic
// groups are walked in order for the running-statistics update (one momentum step per group).
__global__ void norm_finalize_affine_kernel(const double* sum, const double* sumsq, long long rpg, int groups, int C,
                                            const float* gamma, const float* beta, float eps, float momentum,
                                            float* rmean, float* rvar, float* mean, float* rstd, float* scale,
                                            float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double n = (double)rpg;
    const double inv_n = 1.0 / n;
    const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    float rm = rmean ? rmean[c] : 0.f, rv = rvar ? rvar[c] : 0.f;
    for (int g = 0; g < groups; ++g) {
        const long long i = (long long)g * C + c;
        const double m = sum[i] * inv_n;
        double var = sumsq[i] * inv_n - m * m;
        if (var < 0.0 || rpg == 1) var = 0.0;
        const float r = rsqrtf((float)var + eps);
        mean[i] = (float)m;
        rstd[i] = r;
        norm_affine(gm, bt, (float)m, r, scale[i], shift[i]);
        const double unb = rpg > 1 ? var * n / (n - 1.0) : var;
        rm = (1.f - momentum) * rm + momentum * (float)m;
        rv = (1.f - momentum) * rv + momentum * (float)unb;
    }
    if (rmean) {
        rmean[c] = rm;
        rvar[c] = rv;
    }
}

__global__ void norm_eval_affine_kernel(int C, const float* gamma, const float* beta, const float* rmean,
                                        const float* rvar, float eps, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float r = 1.f / sqrtf(rvar[c] + eps);
    const float gm = gamma ? gamma[c] : 1.f;
    const float bt = beta ? beta[c] : 0.f;
    scale[c] = gm * r;
    shift[c] = bt - rmean[c] * gm * r;
}

__global__ void norm_bwd_finalize_kernel(const double* s1, const double* s2, long long rpg, int groups, int C,
                                         const float* gamma, const float* mean, const float* rstd, int train, float* A,
                                         float* B, float* Cc, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float gm = gamma ? gamma[c] : 1.f;
    double t1 = 0.0, t2 = 0.0;
    for (int g = 0; g < groups; ++g) {
        const long long i = (long long)g * C + c;
        const double a1 = s1[i], a2 = s2 ? s2[i] : 0.0;
        t1 += a1;
        t2 += a2;
        const double r = rstd[i], m = mean[i];
        if (train) {
            const double n = (double)rpg;
            // dx = gm*r*( g - a1/n - xhat*a2/n ),  xhat = (x-m)*r
            A[i] = (float)(gm * r);
            B[i] = (float)(-gm * r * r * (a2 / n));
            Cc[i] = (float)(-gm * r * (a1 / n) + gm * r * r * (a2 / n) * m);
        } else {
            A[i] = (float)(gm * r);
            B[i] = 0.f;
            Cc[i] = 0.f;
        }
    }
    if (dgamma) dgamma[c] += (float)t2;
    if (dbeta) dbeta[c] += (float)t1;
}

// ------------------------------------------------------------ element-wise --
constexpr int EW_U = 4;  // rows in flight per thread (16 B each per operand)

struct EwParams {
    const void* x; int ldx;
    const float* scale; const float* shift;
    const void* res; int ldres;
    void* y; int ldy;
    int C; long long rows_per_group; int rows_per_block; int act; int tx, ty;
    // fused finalize (training-mode statistics -> affine) when sum != NULL
    const double* sum; const double* sumsq;
    const float* gamma; const float* beta;
    float eps, momentum;
    float* rmean; float* rvar;      // running statistics (batch norm, groups == 1) or NULL
    float* mean_out; float* rstd_out;
    int gx, gy;
};

template <typename T, bool RES>
__global__ __launch_bounds__(256) void norm_act_fwd_kernel(EwParams P) {
    constexpr int VEC = Elem<T>::VEC;
    const int ly = (int)threadIdx.x / P.tx;
    const int lx = (int)threadIdx.x - ly * P.tx;
    const int ty = P.ty;
    BG_BLOCK_COORDS(P, bx, by);
    const int c = (bx * P.tx + lx) * VEC;
    const int g = blockIdx.z;
    const long long r0 = (long long)by * P.rows_per_block;
    long long r1 = r0 + P.rows_per_block;
    if (r1 > P.rows_per_group) r1 = P.rows_per_group;
    const long long row0 = (long long)g * P.rows_per_group + r0 + ly;
    // The first batch of rows is requested BEFORE the coefficient table is formed: the table costs a round trip of its own
    // (the producer's fp64 sums) on which the loads do not depend, and on small maps -- where a launch is a few microseconds
    // of latency, not bandwidth -- the two round trips otherwise add up.
    const bool c_ok = c < P.C;
    const T* x = reinterpret_cast<const T*>(P.x) + row0 * P.ldx + c;
    const T* res = RES ? reinterpret_cast<const T*>(P.res) + row0 * P.ldres + c : nullptr;
    const long long sx = (long long)ty * P.ldx, sr = (long long)ty * P.ldres, sy = (long long)ty * P.ldy;
    long long r = r0 + ly;
    const long long step = EW_U * (long long)ty;
    Chunk<T> vx[EW_U], vr[EW_U];
    bool have = c_ok && r + step - ty < r1;
    if (have) {
#pragma unroll
        for (int u = 0; u < EW_U; ++u) {
            vx[u].load(x + u * sx);
            if (RES) vr[u].load(res + u * sr);
        }
    }
    // Per-channel affine: the block derives it ONCE per channel, lanes along channels (coalesced table reads), and
    // hands it to the row lanes through LDS.  When every thread read its own 8 channels straight from the tables, each
    // wave-level load touched 32-64 cache lines for 4-8 useful bytes per lane: rocprofv3 counted 8x the L1 accesses of
    // a plain copy of the same tensor and the kernel ran 1.5x slower than that copy (scripts/ew_patterns.hip).
    extern __shared__ float ew_tab[];
    const int row_w = P.tx * VEC, nthr = P.tx * ty;
    float* s_sc = ew_tab;
    float* s_sh = ew_tab + row_w;
    {
        const double n = (double)P.rows_per_group;
        const double inv_n = 1.0 / n;  // one fp64 division per thread; the rest is fp64 mul/add + one fp32 rsqrt
        for (int ch = threadIdx.x; ch < row_w; ch += nthr) {
            const int cg = bx * row_w + ch;
            float a = 1.f, b = 0.f;
            if (cg < P.C) {
                const long long i = (long long)g * P.C + cg;
                if (P.sum) {
                    // fused finalize: the affine from the fp64 sums; the first row-block of each group also publishes
                    // mean / rstd for the backward pass and applies the BatchNorm running-statistics update
                    const double m = P.sum[i] * inv_n;
                    double var = P.sumsq[i] * inv_n - m * m;  // the cancellation-prone step stays in fp64
                    if (var < 0.0 || P.rows_per_group == 1) var = 0.0;
                    const float r = rsqrtf((float)var + P.eps);
                    norm_affine(P.gamma ? P.gamma[cg] : 1.f, P.beta ? P.beta[cg] : 0.f, (float)m, r, a, b);
                    if (by == 0) {
                        P.mean_out[i] = (float)m;
                        P.rstd_out[i] = r;
                        if (P.rmean && g == 0) {
                            // BatchNorm over several statistic groups (sub-batches the reference pushes through the
                            // layer in separate calls): one momentum update per group, in group order
                            float rm = P.rmean[cg], rv = P.rvar[cg];
                            for (int gg = 0; gg < (int)gridDim.z; ++gg) {
                                const long long j = (long long)gg * P.C + cg;
                                const double mg = P.sum[j] * inv_n;
                                double vg = P.sumsq[j] * inv_n - mg * mg;
                                if (vg < 0.0 || P.rows_per_group == 1) vg = 0.0;
                                const double unb = P.rows_per_group > 1 ? vg * n / (n - 1.0) : vg;
                                rm = (1.f - P.momentum) * rm + P.momentum * (float)mg;
                                rv = (1.f - P.momentum) * rv + P.momentum * (float)unb;
                            }
                            P.rmean[cg] = rm;
                            P.rvar[cg] = rv;
                        }
                    }
                } else if (P.scale) {
                    a = P.scale[i];
                    b = P.shift[i];
                }
            }
            // stored element-major ([e][lx]): the read below is then one conflict-free ds_read_b32 per element, lanes
            // consecutive (lane-major rows of 8 floats put 16 lanes on the same banks: SQ_LDS_BANK_CONFLICT 0.8 of the
            // few LDS cycles of these kernels in round 2)
            const int ti = (ch % VEC) * P.tx + ch / VEC;
            s_sc[ti] = a;
            s_sh[ti] = b;
        }
    }
    __syncthreads();
    if (!c_ok) return;
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        sc[e] = s_sc[e * P.tx + lx];
        sh[e] = s_sh[e * P.tx + lx];
    }
    T* y = reinterpret_cast<T*>(P.y) + row0 * P.ldy + c;
    // LeakyReLU / ReLU / identity as max(z, slope*z) with slope 0.2 / 0 / 1: the same values as the branch
    // (z >= 0 ? z : slope*z) for every slope in [0, 1], with no per-element selects on run-time flags.  These kernels
    // are HBM-bound only while the VALU work per 16-byte chunk stays small: with the act / residual flags tested per
    // element the bf16 kernel spent ~85 VALU instructions per chunk and ran 1.5x slower than a copy.
    const float slope = act_max_slope(P.act);
    auto row = [&](const Chunk<T>& vx, const Chunk<T>& vr, T* out) {
        Chunk<T> vo;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float z = fmaf(vx.get(e), sc[e], sh[e]);
            if (RES) z += vr.get(e);
            vo.set(e, fmaxf(z, slope * z));
        }
        vo.store(out);
    };
    // EW_U rows per thread are in flight before the first is used, and the next batch is requested before the current
    // one is computed and stored (the one-row "#pragma unroll 4" loop compiled to load -> s_waitcnt vmcnt(0) -> store).
    while (have) {
        Chunk<T> cx[EW_U], cr[EW_U];
#pragma unroll
        for (int u = 0; u < EW_U; ++u) {
            cx[u] = vx[u];
            if (RES) cr[u] = vr[u];
        }
        r += step;
        x += EW_U * sx;
        if (RES) res += EW_U * sr;
        have = r + step - ty < r1;
        if (have) {
#pragma unroll
            for (int u = 0; u < EW_U; ++u) {
                vx[u].load(x + u * sx);
                if (RES) vr[u].load(res + u * sr);
            }
        }
#pragma unroll
        for (int u = 0; u < EW_U; ++u) row(cx[u], cr[u], y + u * sy);
        y += EW_U * sy;
    }
    for (; r < r1; r += ty) {
        Chunk<T> v1, v2;
        v1.load(x);
        x += sx;
        if (RES) { v2.load(res); res += sr; }
        row(v1, v2, y);
        y += sy;
    }
}

struct EwBwdParams {
    const void* dy; int lddy;
    const void* y; int ldy;
    const void* x; int ldx;
    const float* A; const float* B; const float* Cc;
    void* dx; int lddx;
    void* dres; int lddres;
    int C; long long rows_per_group; int rows_per_block; int act; int tx, ty;
    // fused finalize when s1 != NULL: coefficients from the fp64 sums, dgamma/dbeta by the first row-block
    const double* s1; const double* s2;
    const float* gamma; const float* beta; const float* mean; const float* rstd;
    float* dgamma; float* dbeta;
    int train; int groups;
    int gx, gy;
    // Q8 (fp8 operand path): dx is ALSO written as e5m2 bytes -- the data-gradient GEMM of the pointwise convolution that
    // produced x reads it as its operand -- q = e5m2(clamp(dx_as_stored * 2^(*q_exp))), C rounded up to 16 with zero lanes,
    // and *q_amax = max(*q_amax, max |dx|) for the next step's exponent (bg_quant_fp8's contract, one pass saved)
    unsigned char* dxq; int lddxq; const int* q_exp; unsigned* q_amax;
};

// SIGN: where the activation's branch comes from (0 no activation, 1 the stored output y, 2 recomputed from
// x*scale+shift); DRES: also write the residual branch's gradient; USEB: training-mode statistics terms (needs x).
// Compile-time so that the per-element code carries no selects on run-time flags (see norm_act_fwd_kernel).
template <typename T, int SIGN, bool DRES, bool USEB, bool Q8 = false>
__global__ __launch_bounds__(256) void norm_act_bwd_apply_kernel(EwBwdParams P) {
    constexpr int VEC = Elem<T>::VEC;
    static_assert(!Q8 || sizeof(T) == 2, "the fp8 copy is taken from a bf16 gradient");
    __shared__ unsigned q_blk_amax;
    if (Q8 && threadIdx.x == 0) q_blk_amax = 0u;   // published by the __syncthreads() after the coefficient table
    const int ly = (int)threadIdx.x / P.tx;
    const int lx = (int)threadIdx.x - ly * P.tx;
    const int ty = P.ty;
    BG_BLOCK_COORDS(P, bx, by);
    const int c = (bx * P.tx + lx) * VEC;
    const int g = blockIdx.z;
    const long long r0 = (long long)by * P.rows_per_block;
    long long r1 = r0 + P.rows_per_block;
    if (r1 > P.rows_per_group) r1 = P.rows_per_group;
    const long long row0 = (long long)g * P.rows_per_group + r0 + ly;
    constexpr bool useB = USEB;
    // first batch of rows requested before the coefficient table (see norm_act_fwd_kernel)
    const bool c_ok = c < P.C;
    constexpr bool LDX = USEB || SIGN == 2;
    const T* dy = reinterpret_cast<const T*>(P.dy) + row0 * P.lddy + c;
    const T* y = SIGN == 1 ? reinterpret_cast<const T*>(P.y) + row0 * P.ldy + c : nullptr;
    const T* x = LDX ? reinterpret_cast<const T*>(P.x) + row0 * P.ldx + c : nullptr;
    const long long s_dy = (long long)ty * P.lddy, s_y = (long long)ty * P.ldy, s_x = (long long)ty * P.ldx;
    auto load = [&](Chunk<T>* vg, Chunk<T>* vy, Chunk<T>* vx) {
#pragma unroll
        for (int u = 0; u < EW_U; ++u) {
            vg[u].load(dy + u * s_dy);
            if (SIGN == 1) vy[u].load(y + u * s_y);
            if (LDX) vx[u].load(x + u * s_x);
        }
    };
    long long r = r0 + ly;
    const long long step = EW_U * (long long)ty;
    Chunk<T> vg[EW_U], vy[EW_U], vx[EW_U];
    bool have = c_ok && r + step - ty < r1;
    if (have) load(vg, vy, vx);
    // per-channel coefficients once per block, lanes along channels, handed over through LDS (see norm_act_fwd_kernel)
    extern __shared__ float ew_tab[];
    const int row_w = P.tx * VEC, nthr = P.tx * ty;
    {
        const float inv_n = 1.f / (float)P.rows_per_group;
        for (int ch = threadIdx.x; ch < row_w; ch += nthr) {
            const int cg = bx * row_w + ch;
            float a = 1.f, b = 0.f, cc0 = 0.f, s = 1.f, h = 0.f;
            if (cg < P.C) {
                const long long i = (long long)g * P.C + cg;
                if (P.s1) {
                    // fused finalize: coefficients from the fp64 sums; dgamma / dbeta by the first row-block
                    const float gm = P.gamma ? P.gamma[cg] : 1.f;
                    const float r = P.rstd[i], m = P.mean[i];
                    norm_affine(gm, P.beta ? P.beta[cg] : 0.f, m, r, s, h);
                    const float a1 = (float)P.s1[i] * inv_n, a2 = (float)P.s2[i] * inv_n;  // means of g and g*xhat
                    a = gm * r;
                    b = P.train ? -gm * r * r * a2 : 0.f;
                    cc0 = P.train ? gm * r * (r * a2 * m - a1) : 0.f;
                    if (P.dgamma && by == 0 && g == 0) {
                        // parameter gradients: sum over groups (one thread per channel, once)
                        double t1 = 0.0, t2 = 0.0;
                        for (int gg = 0; gg < P.groups; ++gg) {
                            t1 += P.s1[(long long)gg * P.C + cg];
                            t2 += P.s2[(long long)gg * P.C + cg];
                        }
                        P.dgamma[cg] += (float)t2;
                        P.dbeta[cg] += (float)t1;
                    }
                } else {
                    a = P.A ? P.A[i] : 1.f;
                    b = useB ? P.B[i] : 0.f;
                    cc0 = useB ? P.Cc[i] : 0.f;
                }
            }
            const int ti = (ch % VEC) * P.tx + ch / VEC;   // element-major, see norm_act_fwd_kernel
            ew_tab[ti] = a;
            ew_tab[row_w + ti] = b;
            ew_tab[2 * row_w + ti] = cc0;
            ew_tab[3 * row_w + ti] = s;
            ew_tab[4 * row_w + ti] = h;
        }
    }
    __syncthreads();
    if (!c_ok) return;
    float ca[VEC], cb[VEC], cc[VEC], sc[VEC], sh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        ca[e] = ew_tab[e * P.tx + lx];
        cb[e] = ew_tab[row_w + e * P.tx + lx];
        cc[e] = ew_tab[2 * row_w + e * P.tx + lx];
        sc[e] = ew_tab[3 * row_w + e * P.tx + lx];
        sh[e] = ew_tab[4 * row_w + e * P.tx + lx];
    }
    T* dx = P.dx ? reinterpret_cast<T*>(P.dx) + row0 * P.lddx + c : nullptr;
    T* dres = DRES ? reinterpret_cast<T*>(P.dres) + row0 * P.lddres + c : nullptr;
    const long long s_dx = (long long)ty * P.lddx, s_dr = (long long)ty * P.lddres;
    const float slope = act_slope(P.act);
    const float q_scale = Q8 ? ldexpf(1.f, *P.q_exp) : 0.f;
    float q_mx = 0.f;
    unsigned char* dxq = Q8 ? P.dxq + row0 * P.lddxq + c : nullptr;
    const long long s_dq = (long long)ty * P.lddxq;
    const bool q_pad = Q8 && c + VEC == P.C && (P.C & 8);   // this thread also owns the 8 zero lanes that round C up to 16
    auto row = [&](const Chunk<T>& vg, const Chunk<T>& vy, const Chunk<T>& vx, T* o_res, T* o_dx, unsigned char* o_q) {
        Chunk<T> vo;
        float gg[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            gg[e] = vg.get(e);
            if (SIGN == 1) gg[e] *= vy.get(e) > 0.f ? 1.f : slope;
            if (SIGN == 2) gg[e] *= fmaf(vx.get(e), sc[e], sh[e]) > 0.f ? 1.f : slope;
        }
        if (DRES) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) vo.set(e, gg[e]);
            vo.store(o_res);
        }
        if (dx) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float d = gg[e] * ca[e];
                if (useB) d += fmaf(cb[e], vx.get(e), cc[e]);
                vo.set(e, d);
            }
            vo.store(o_dx);
            if constexpr (Q8) {
                typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = vo.get(e);                       // the gradient as stored (bf16)
                    q_mx = fmaxf(q_mx, fabsf(v[e]));
                    v[e] = fminf(fmaxf(v[e] * q_scale, -57344.f), 57344.f);
                }
                u32x2 qv;
                int w0 = 0, w1 = 0;
                w0 = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], w0, false);
                w0 = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], w0, true);
                w1 = __builtin_amdgcn_cvt_pk_bf8_f32(v[4], v[5], w1, false);
                w1 = __builtin_amdgcn_cvt_pk_bf8_f32(v[6], v[7], w1, true);
                qv[0] = (unsigned)w0; qv[1] = (unsigned)w1;
                *reinterpret_cast<u32x2*>(o_q) = qv;
                if (q_pad) *reinterpret_cast<u32x2*>(o_q + 8) = u32x2{0u, 0u};
            }
        }
    };
    // software-pipelined batches of EW_U rows (see norm_act_fwd_kernel)
    while (have) {
        Chunk<T> cg[EW_U], cy[EW_U], cx[EW_U];
#pragma unroll
        for (int u = 0; u < EW_U; ++u) {
            cg[u] = vg[u];
            if (SIGN == 1) cy[u] = vy[u];
            if (LDX) cx[u] = vx[u];
        }
        r += step;
        dy += EW_U * s_dy;
        if (SIGN == 1) y += EW_U * s_y;
        if (LDX) x += EW_U * s_x;
        have = r + step - ty < r1;
        if (have) load(vg, vy, vx);
#pragma unroll
        for (int u = 0; u < EW_U; ++u) row(cg[u], cy[u], cx[u], dres + u * s_dr, dx + u * s_dx, Q8 ? dxq + u * s_dq : nullptr);
        if (DRES) dres += EW_U * s_dr;
        if (dx) dx += EW_U * s_dx;
        if (Q8) dxq += EW_U * s_dq;
    }
    for (; r < r1; r += ty) {
        Chunk<T> v1, v2, v3;
        v1.load(dy);
        dy += s_dy;
        if (SIGN == 1) { v2.load(y); y += s_y; }
        if (LDX) { v3.load(x); x += s_x; }
        row(v1, v2, v3, dres, dx, dxq);
        if (DRES) dres += s_dr;
        if (dx) dx += s_dx;
        if (Q8) dxq += s_dq;
    }
    if constexpr (Q8) {   // one atomic per workgroup (|v| as float bits orders like the unsigned integer)
        if (q_mx > 0.f) atomicMax(&q_blk_amax, __float_as_uint(q_mx));
        __syncthreads();    // threads that left at `c >= P.C` are whole inactive lanes of waves that still arrive here
        if (threadIdx.x == 0 && q_blk_amax) atomicMax(P.q_amax, q_blk_amax);
    }
}

template <typename T, int SIGN>
void launch_bwd_apply_s(const EwBwdParams& P, const dim3 grid, const dim3 block, bool useB, hipStream_t st) {
    const size_t lds = (size_t)5 * P.tx * Elem<T>::VEC * sizeof(float);
    if constexpr (sizeof(T) == 2) {
        if (P.dxq) {   // (the entry point admits the fp8 copy only with the training-mode terms)
            if (P.dres) hipLaunchKernelGGL((norm_act_bwd_apply_kernel<T, SIGN, true, true, true>), grid, block, lds, st, P);
            else hipLaunchKernelGGL((norm_act_bwd_apply_kernel<T, SIGN, false, true, true>), grid, block, lds, st, P);
            return;
        }
    }
    if (P.dres) {
        if (useB) hipLaunchKernelGGL((norm_act_bwd_apply_kernel<T, SIGN, true, true>), grid, block, lds, st, P);
        else hipLaunchKernelGGL((norm_act_bwd_apply_kernel<T, SIGN, true, false>), grid, block, lds, st, P);
    } else {
        if (useB) hipLaunchKernelGGL((norm_act_bwd_apply_kernel<T, SIGN, false, true>), grid, block, lds, st, P);
        else hipLaunchKernelGGL((norm_act_bwd_apply_kernel<T, SIGN, false, false>), grid, block, lds, st, P);
    }
}
template <typename T>
void launch_bwd_apply_t(const EwBwdParams& P, const Tiling& t, int groups, hipStream_t st) {
    const dim3 grid(xcd_grid(t.gx, t.gy), 1, groups), block(t.tx * t.ty);
    const bool useB = P.s1 ? (P.train != 0) : (P.A && P.B);
    const int sign = !P.act ? 0 : (P.y ? 1 : 2);
    if (sign == 0) launch_bwd_apply_s<T, 0>(P, grid, block, useB, st);
    else if (sign == 1) launch_bwd_apply_s<T, 1>(P, grid, block, useB, st);
    else launch_bwd_apply_s<T, 2>(P, grid, block, useB, st);
}
inline void launch_bwd_apply(int dtype, const EwBwdParams& P, const Tiling& t, int groups, hipStream_t st) {
    BG_DISPATCH_DTYPE(dtype, T, launch_bwd_apply_t<T>(P, t, groups, st));
}

template <typename T>
void launch_norm_act_fwd_t(const EwParams& P, const Tiling& t, int groups, hipStream_t st) {
    const dim3 grid(xcd_grid(t.gx, t.gy), 1, groups), block(t.tx * t.ty);
    const size_t lds = (size_t)2 * t.tx * Elem<T>::VEC * sizeof(float);
    if (P.res) hipLaunchKernelGGL((norm_act_fwd_kernel<T, true>), grid, block, lds, st, P);
    else hipLaunchKernelGGL((norm_act_fwd_kernel<T, false>), grid, block, lds, st, P);
}
inline void launch_norm_act_fwd(int dtype, const EwParams& P, const Tiling& t, int groups, hipStream_t st) {
    BG_DISPATCH_DTYPE(dtype, T, launch_norm_act_fwd_t<T>(P, t, groups, st));
}

inline unsigned ew_grid(long long total) {
    long long g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}

int check_rows(int dtype, long long rows, int C, int groups, const char* who) {
    BG_CHECK_ARG(dtype_ok(dtype), "%s: bad dtype", who);
    BG_CHECK_ARG(rows > 0 && C > 0 && groups > 0 && rows % groups == 0, "%s: rows=%lld C=%d groups=%d", who, rows, C,
                 groups);
    BG_CHECK_ARG(C % dtype_vec(dtype) == 0, "%s: C=%d must be a multiple of %d", who, C, dtype_vec(dtype));
    return BG_OK;
}
#define CHECK_LD(ld, who) \
    BG_CHECK_ARG((ld) >= C && (ld) % dtype_vec(dtype) == 0, "%s: bad pixel stride %d for C=%d", who, (int)(ld), C)

}  // namespace

extern "C" int bg_norm_stats(int32_t dtype, const void* x, int64_t rows, int32_t C, int32_t ldx, int32_t groups,
                             double* sum, double* sumsq, void* stream) {
    int rc = check_rows(dtype, rows, C, groups, "bg_norm_stats");
    if (rc) return rc;
    CHECK_LD(ldx, "bg_norm_stats");
    BG_CHECK_ARG(x && sum && sumsq && aligned16(x), "bg_norm_stats: null/unaligned pointer");
    RedParams P{};
    P.a = x; P.lda = ldx; P.C = C; P.rows_per_group = rows / groups; P.o1 = sum; P.o2 = sumsq;
    return launch_colreduce<RED_STATS>(dtype, P, groups, (hipStream_t)stream, "bg_norm_stats");
}

extern "C" int bg_colsum(int32_t dtype, const void* x, int32_t ldx, int64_t rows, int32_t C, int32_t groups,
                         float scale, float* out, void* stream) {
    int rc = check_rows(dtype, rows, C, groups, "bg_colsum");
    if (rc) return rc;
    CHECK_LD(ldx, "bg_colsum");
    BG_CHECK_ARG(x && out && aligned16(x), "bg_colsum: null/unaligned pointer");
    RedParams P{};
    P.a = x; P.lda = ldx; P.C = C; P.rows_per_group = rows / groups; P.of = out; P.scale = scale;
    return launch_colreduce<RED_COLSUM>(dtype, P, groups, (hipStream_t)stream, "bg_colsum");
}

extern "C" int bg_norm_act_bwd_reduce(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy,
                                      const void* x, int32_t ldx, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, int64_t rows, int32_t C, int32_t groups,
                                      int32_t act, double* s1, double* s2, void* stream) {
    int rc = check_rows(dtype, rows, C, groups, "bg_norm_act_bwd_reduce");
    if (rc) return rc;
    CHECK_LD(lddy, "bg_norm_act_bwd_reduce");
    BG_CHECK_ARG(dy && s1 && aligned16(dy), "bg_norm_act_bwd_reduce: null/unaligned pointer");
    if (act && y) {
        BG_CHECK_ARG(aligned16(y), "bg_norm_act_bwd_reduce: unaligned y");
        CHECK_LD(ldy, "bg_norm_act_bwd_reduce");
    }
    BG_CHECK_ARG(!act || y || x, "bg_norm_act_bwd_reduce: act needs y, or x to recompute the pre-activation");
    if (x) {
        BG_CHECK_ARG(aligned16(x) && s2 && mean && rstd, "bg_norm_act_bwd_reduce: x needs mean/rstd/s2");
        CHECK_LD(ldx, "bg_norm_act_bwd_reduce");
    }
    RedParams P{};
    P.a = dy; P.lda = lddy; P.b = y; P.ldb = ldy; P.c = x; P.ldc = ldx;
    P.mean = mean; P.rstd = rstd; P.gamma = gamma; P.beta = beta; P.act = act; P.C = C;
    P.rows_per_group = rows / groups;
    P.o1 = s1; P.o2 = x ? s2 : nullptr;
    return launch_colreduce<RED_BWD>(dtype, P, groups, (hipStream_t)stream, "bg_norm_act_bwd_reduce");
}

extern "C" int bg_norm_finalize(const double* sum, const double* sumsq, int64_t rows_per_group, int32_t groups,
                                int32_t C, const float* gamma, const float* beta, float eps, float momentum,
                                float* running_mean, float* running_var, float* mean, float* rstd, float* scale,
                                float* shift, void* stream) {
    BG_CHECK_ARG(sum && sumsq && mean && rstd && scale && shift && rows_per_group > 0 && groups > 0 && C > 0,
                 "bg_norm_finalize: bad args");
    BG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bg_norm_finalize: running stats must come in pairs");
    BG_CHECK_ARG(!(running_mean && groups != 1), "bg_norm_finalize: running stats only with batch statistics");
    const int n = groups * C;
    hipLaunchKernelGGL(norm_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, sum, sumsq,
                       (long long)rows_per_group, groups, C, gamma, beta, eps, momentum, running_mean, running_var, mean,
                       rstd, scale, shift);
    BG_CHECK_LAUNCH("norm_finalize_kernel");
    return BG_OK;
}

extern "C" int bg_norm_finalize_affine(const double* sum, const double* sumsq, int64_t rows_per_group, int32_t groups,
                                       int32_t C, const float* gamma, const float* beta, float eps, float momentum,
                                       float* running_mean, float* running_var, float* mean, float* rstd,
                                       float* scale, float* shift, void* stream) {
    BG_CHECK_ARG(sum && sumsq && mean && rstd && scale && shift && rows_per_group > 0 && groups > 0 && C > 0,
                 "bg_norm_finalize_affine: bad args");
    BG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr),
                 "bg_norm_finalize_affine: running stats must come in pairs");
    hipLaunchKernelGGL(norm_finalize_affine_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, sum, sumsq,
                       (long long)rows_per_group, groups, C, gamma, beta, eps, momentum, running_mean, running_var, mean,
                       rstd, scale, shift);
    BG_CHECK_LAUNCH("norm_finalize_affine_kernel");
    return BG_OK;
}

extern "C" int bg_norm_eval_affine(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, float* scale, float* shift, void* stream) {
    BG_CHECK_ARG(C > 0 && running_mean && running_var && scale && shift, "bg_norm_eval_affine: bad args");
    hipLaunchKernelGGL(norm_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, gamma, beta,
                       running_mean, running_var, eps, scale, shift);
    BG_CHECK_LAUNCH("norm_eval_affine_kernel");
    return BG_OK;
}

extern "C" int bg_norm_bwd_finalize(const double* s1, const double* s2, int64_t rows_per_group, int32_t groups,
                                    int32_t C, const float* gamma, const float* mean, const float* rstd, int32_t train,
                                    float* A, float* B, float* Cc, float* dgamma, float* dbeta, void* stream) {
    BG_CHECK_ARG(s1 && mean && rstd && A && B && Cc && rows_per_group > 0 && groups > 0 && C > 0,
                 "bg_norm_bwd_finalize: bad args");
    hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, s1, s2,
                       (long long)rows_per_group, groups, C, gamma, mean, rstd, train, A, B, Cc, dgamma, dbeta);
    BG_CHECK_LAUNCH("norm_bwd_finalize_kernel");
    return BG_OK;
}

extern "C" int bg_norm_act_fwd(int32_t dtype, const void* x, int32_t ldx, const float* scale, const float* shift,
                               const void* res, int32_t ldres, void* y, int32_t ldy, int64_t rows, int32_t C,
                               int32_t groups, int32_t act, void* stream) {
    int rc = check_rows(dtype, rows, C, groups, "bg_norm_act_fwd");
    if (rc) return rc;
    CHECK_LD(ldx, "bg_norm_act_fwd");
    CHECK_LD(ldy, "bg_norm_act_fwd");
    BG_CHECK_ARG(x && y && aligned16(x) && aligned16(y), "bg_norm_act_fwd: null/unaligned pointer");
    BG_CHECK_ARG((scale == nullptr) == (shift == nullptr), "bg_norm_act_fwd: scale/shift must come in pairs");
    if (res) {
        BG_CHECK_ARG(aligned16(res), "bg_norm_act_fwd: unaligned res");
        CHECK_LD(ldres, "bg_norm_act_fwd");
    }
    const Tiling t = make_tiling(dtype, C, rows / groups, groups, 4, 8192, common_ld({ldx, ldy, res ? ldres : 0}));
    BG_CHECK_ARG(groups <= 65535, "bg_norm_act_fwd: too many groups");
    EwParams P{x, ldx, scale, shift, res, ldres, y, ldy, C, rows / groups, t.rows_per_block, act, t.tx, t.ty,
               nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr};
    P.gx = t.gx; P.gy = t.gy;
    launch_norm_act_fwd(dtype, P, t, groups, (hipStream_t)stream);
    BG_CHECK_LAUNCH("norm_act_fwd_kernel");
    return BG_OK;
}

extern "C" int bg_norm_act_bwd_apply(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy,
                                     const void* x, int32_t ldx, const float* A, const float* B, const float* Cc,
                                     void* dx, int32_t lddx, void* dres, int32_t lddres, int64_t rows, int32_t C,
                                     int32_t groups, int32_t act, void* stream) {
    int rc = check_rows(dtype, rows, C, groups, "bg_norm_act_bwd_apply");
    if (rc) return rc;
    CHECK_LD(lddy, "bg_norm_act_bwd_apply");
    BG_CHECK_ARG(dy && aligned16(dy) && (dx || dres), "bg_norm_act_bwd_apply: null/unaligned pointer");
    if (act) {
        BG_CHECK_ARG(y && aligned16(y), "bg_norm_act_bwd_apply: act needs y");
        CHECK_LD(ldy, "bg_norm_act_bwd_apply");
    }
    BG_CHECK_ARG(!(B && !A) && ((B == nullptr) == (Cc == nullptr)), "bg_norm_act_bwd_apply: A/B/Cc inconsistent");
    if (A && B) {
        BG_CHECK_ARG(x && aligned16(x), "bg_norm_act_bwd_apply: B needs x");
        CHECK_LD(ldx, "bg_norm_act_bwd_apply");
    }
    if (dx) {
        BG_CHECK_ARG(aligned16(dx), "bg_norm_act_bwd_apply: unaligned dx");
        CHECK_LD(lddx, "bg_norm_act_bwd_apply");
    }
    if (dres) {
        BG_CHECK_ARG(aligned16(dres), "bg_norm_act_bwd_apply: unaligned dres");
        CHECK_LD(lddres, "bg_norm_act_bwd_apply");
    }
    const Tiling t = make_tiling(dtype, C, rows / groups, groups, 4, 8192,
                                 common_ld({lddy, (act && y) ? ldy : 0, (A && B) ? ldx : 0, dx ? lddx : 0, dres ? lddres : 0}));
    BG_CHECK_ARG(groups <= 65535, "bg_norm_act_bwd_apply: too many groups");
    EwBwdParams P{dy, lddy, y, ldy, x, ldx, A, B, Cc, dx, lddx, dres, lddres, C, rows / groups, t.rows_per_block, act,
                  t.tx, t.ty, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, groups};
    P.gx = t.gx; P.gy = t.gy;
    launch_bwd_apply(dtype, P, t, groups, (hipStream_t)stream);
    BG_CHECK_LAUNCH("norm_act_bwd_apply_kernel");
    return BG_OK;
}

extern "C" int bg_norm_act_fwd_stats(int32_t dtype, const void* x, int32_t ldx, const double* sum, const double* sumsq,
                                     const float* gamma, const float* beta, float eps, float momentum,
                                     float* running_mean, float* running_var, float* mean, float* rstd, const void* res,
                                     int32_t ldres, void* y, int32_t ldy, int64_t rows, int32_t C, int32_t groups,
                                     int32_t act, void* stream) {
    int rc = check_rows(dtype, rows, C, groups, "bg_norm_act_fwd_stats");
    if (rc) return rc;
    CHECK_LD(ldx, "bg_norm_act_fwd_stats");
    CHECK_LD(ldy, "bg_norm_act_fwd_stats");
    BG_CHECK_ARG(x && y && sum && sumsq && mean && rstd && aligned16(x) && aligned16(y),
                 "bg_norm_act_fwd_stats: null/unaligned pointer");
    BG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr),
                 "bg_norm_act_fwd_stats: running statistics come in pairs");
    if (res) {
        BG_CHECK_ARG(aligned16(res), "bg_norm_act_fwd_stats: unaligned res");
        CHECK_LD(ldres, "bg_norm_act_fwd_stats");
    }
    // fewer, longer-running blocks than the plain apply kernel (1 024 against 8 192): each block first derives its channels'
    // affine from the sums, which must be amortised over the rows it then walks.  The floor of rows per thread is 4, one
    // batch of loads: with 16 the 16 x 16 maps of the 256 x 256 configuration ran as 64 blocks of four dependent batches
    // (256 x 256 step 22.53 -> 22.12 ms, 1152 x 768 unchanged; scripts/gpu_ews.sh)
    static const int k_rows = env_int("BGAMD_EWS_ROWS", 4), k_blocks = env_int("BGAMD_EWS_BLOCKS", 1024);  // tuning knobs
    const Tiling t = make_tiling(dtype, C, rows / groups, groups, k_rows, k_blocks, common_ld({ldx, ldy, res ? ldres : 0}));
    BG_CHECK_ARG(groups <= 65535, "bg_norm_act_fwd_stats: too many groups");
    EwParams P{x, ldx, nullptr, nullptr, res, ldres, y, ldy, C, rows / groups, t.rows_per_block, act, t.tx, t.ty,
               sum, sumsq, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd};
    P.gx = t.gx; P.gy = t.gy;
    launch_norm_act_fwd(dtype, P, t, groups, (hipStream_t)stream);
    BG_CHECK_LAUNCH("norm_act_fwd_kernel(stats)");
    return BG_OK;
}

namespace {
int bwd_apply_stats_impl(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x, int32_t ldx,
                         const double* s1, const double* s2, const float* gamma, const float* beta, const float* mean,
                         const float* rstd, int32_t train, float* dgamma, float* dbeta, void* dx, int32_t lddx, void* dres,
                         int32_t lddres, int64_t rows, int32_t C, int32_t groups, int32_t act, void* dxq, int32_t lddxq,
                         const int32_t* q_exp, uint32_t* q_amax, void* stream);
}

extern "C" int bg_norm_act_bwd_apply_stats(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy,
                                           const void* x, int32_t ldx, const double* s1, const double* s2,
                                           const float* gamma, const float* beta, const float* mean,
                                           const float* rstd, int32_t train, float* dgamma, float* dbeta, void* dx,
                                           int32_t lddx, void* dres,
                                           int32_t lddres, int64_t rows, int32_t C, int32_t groups, int32_t act,
                                           void* stream) {
    return bwd_apply_stats_impl(dtype, dy, lddy, y, ldy, x, ldx, s1, s2, gamma, beta, mean, rstd, train, dgamma, dbeta, dx, lddx,
                                dres, lddres, rows, C, groups, act, nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int bg_norm_act_bwd_apply_stats_q8(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy,
                                              const void* x, int32_t ldx, const double* s1, const double* s2,
                                              const float* gamma, const float* beta, const float* mean, const float* rstd,
                                              int32_t train, float* dgamma, float* dbeta, void* dx, int32_t lddx, void* dres,
                                              int32_t lddres, int64_t rows, int32_t C, int32_t groups, int32_t act, void* dxq,
                                              int32_t lddxq, const int32_t* q_exp, uint32_t* q_amax, void* stream) {
    BG_CHECK_ARG(dtype == BG_BF16 && train && dx && dxq && q_exp && q_amax && aligned16(dxq) && lddxq % 16 == 0 &&
                     lddxq >= (C + 15) / 16 * 16,
                 "bg_norm_act_bwd_apply_stats_q8: bf16, training mode, dx and a 16-byte-aligned fp8 buffer of "
                 "pixel stride >= C rounded up to 16");
    return bwd_apply_stats_impl(dtype, dy, lddy, y, ldy, x, ldx, s1, s2, gamma, beta, mean, rstd, train, dgamma, dbeta, dx, lddx,
                                dres, lddres, rows, C, groups, act, dxq, lddxq, q_exp, q_amax, stream);
}

namespace {
int bwd_apply_stats_impl(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy,
                         const void* x, int32_t ldx, const double* s1, const double* s2,
                         const float* gamma, const float* beta, const float* mean,
                         const float* rstd, int32_t train, float* dgamma, float* dbeta, void* dx,
                         int32_t lddx, void* dres,
                         int32_t lddres, int64_t rows, int32_t C, int32_t groups, int32_t act, void* dxq, int32_t lddxq,
                         const int32_t* q_exp, uint32_t* q_amax, void* stream) {
    int rc = check_rows(dtype, rows, C, groups, "bg_norm_act_bwd_apply_stats");
    if (rc) return rc;
    CHECK_LD(lddy, "bg_norm_act_bwd_apply_stats");
    CHECK_LD(ldx, "bg_norm_act_bwd_apply_stats");
    BG_CHECK_ARG(dy && x && s1 && s2 && mean && rstd && aligned16(dy) && aligned16(x) && (dx || dres || dgamma),
                 "bg_norm_act_bwd_apply_stats: null/unaligned pointer");
    BG_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "bg_norm_act_bwd_apply_stats: dgamma/dbeta in pairs");
    if (act && y) {  // y == NULL: the LeakyReLU branch is recomputed from x*scale+shift (no residual)
        BG_CHECK_ARG(aligned16(y), "bg_norm_act_bwd_apply_stats: unaligned y");
        CHECK_LD(ldy, "bg_norm_act_bwd_apply_stats");
    }
    if (dx) {
        BG_CHECK_ARG(aligned16(dx), "bg_norm_act_bwd_apply_stats: unaligned dx");
        CHECK_LD(lddx, "bg_norm_act_bwd_apply_stats");
    }
    if (dres) {
        BG_CHECK_ARG(aligned16(dres), "bg_norm_act_bwd_apply_stats: unaligned dres");
        CHECK_LD(lddres, "bg_norm_act_bwd_apply_stats");
    }
    static const int k_rows = env_int("BGAMD_EWS_ROWS", 4), k_blocks = env_int("BGAMD_EWB_BLOCKS", 256);  // tuning knobs
    const Tiling t = make_tiling(dtype, C, rows / groups, groups, k_rows, k_blocks,
                                 common_ld({lddy, y ? ldy : 0, ldx, dx ? lddx : 0, dres ? lddres : 0}));
    BG_CHECK_ARG(groups <= 65535, "bg_norm_act_bwd_apply_stats: too many groups");
    EwBwdParams P{dy, lddy, y, ldy, x, ldx, nullptr, nullptr, nullptr, dx, lddx, dres, lddres, C, rows / groups,
                  t.rows_per_block, act, t.tx, t.ty, s1, s2, gamma, beta, mean, rstd, dgamma, dbeta, train, groups};
    P.gx = t.gx; P.gy = t.gy;
    P.dxq = (unsigned char*)dxq; P.lddxq = lddxq; P.q_exp = q_exp; P.q_amax = q_amax;
    launch_bwd_apply(dtype, P, t, groups, (hipStream_t)stream);
    BG_CHECK_LAUNCH("norm_act_bwd_apply_kernel(stats)");
    return BG_OK;
}
}  // namespace
