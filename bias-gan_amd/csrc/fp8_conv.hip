// fp8 operand path of the dense convolutions (BASELINE.json configs[4]): the fat-tile implicit GEMM of igemm_fat.h on
// the block-scaled MFMA 16x16x128 f8f6f4 (e4m3 weights x e4m3 / e5m2 activations, fp32 accumulation, bf16 outputs), the
// quantiser with delayed power-of-two scaling, and the e4m3 weight pack.
//
// Why this shape (MI355X_MICROARCH.md, Matrix cores): the plain fp8 MFMAs (16x16x32 / 32x32x16 _fp8_fp8) run at the
// bf16 rate; only the scaled f8f6f4 forms reach 2x.  Their per-lane E8M0 block scales are used here as ONE
// power-of-two scale per tensor (every lane passes the same byte), which costs nothing and leaves the accumulators in
// real units -- the bf16 kernel's epilogue (LDS transpose, statistics) is reused unchanged.  A K-step is still a
// 128-byte row per tile row, i.e. the same LDS-DMA and ds_read traffic per step as in bf16, for twice the K.
#include "igemm_fat.h"

namespace {

// largest e with amax * 2^e <= fmax, fmax = 0.875 * 2^top (e4m3: 448 = 0.875 * 2^9; e5m2: 57344 = 0.875 * 2^16)
__device__ __forceinline__ int exp_for(float amax, int top, int margin) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 0;
    int k;
    const float m = frexpf(amax, &k);   // amax = m * 2^k, m in [0.5, 1)
    int e = (m <= 0.875f ? top : top - 1) - k - margin;
    return e < -96 ? -96 : e > 96 ? 96 : e;
}
__device__ __forceinline__ int fmt_top(int fmt) { return fmt == BG_FP8_E5M2 ? 16 : 9; }
__device__ __forceinline__ float fmt_max(int fmt) { return fmt == BG_FP8_E5M2 ? 57344.f : 448.f; }

// two floats -> two fp8 bytes in the low / high half of a dword (v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32: round to
// nearest even, OCP encodings on gfx950); the clamp keeps finite values finite (e4m3 has no infinity)
template <int FMT>
__device__ __forceinline__ int cvt4(float a, float b, float c, float d, float lim) {
    a = fminf(fmaxf(a, -lim), lim); b = fminf(fmaxf(b, -lim), lim);
    c = fminf(fmaxf(c, -lim), lim); d = fminf(fmaxf(d, -lim), lim);
    int r = 0;
    if constexpr (FMT == BG_FP8_E5M2) {
        r = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, r, false);
        r = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, r, true);
    } else {
        r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
        r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    }
    return r;
}

template <typename T, int FMT>
__global__ __launch_bounds__(256) void quant_fp8_kernel(const T* __restrict__ x, int ldx, long long rows, int C, unsigned char* __restrict__ xq,
                                                        int ldq, int cpr /* 16-byte chunks per output row */, const int* __restrict__ exp,
                                                        unsigned* __restrict__ amax) {
    const float scale = ldexpf(1.f, exp ? *exp : 0);
    const float lim = fmt_max(FMT);
    const long long total = rows * cpr;
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cpr;
        const int c0 = (int)(i - r * cpr) * 16;
        float v[16];
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (c0 + h * 8 < C) {   // C is a multiple of 8: a bf16 vector is all inside or all outside
                    const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + r * ldx + c0 + h * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[h * 8 + e] = (float)t[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[h * 8 + e] = 0.f;
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                if (c0 + h * 4 < C) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(x + r * ldx + c0 + h * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[h * 4 + e] = t[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[h * 4 + e] = 0.f;
                }
            }
        }
        i32x4 o;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(v[g * 4 + e]));
            o[g] = cvt4<FMT>(v[g * 4] * scale, v[g * 4 + 1] * scale, v[g * 4 + 2] * scale, v[g * 4 + 3] * scale, lim);
        }
        *reinterpret_cast<i32x4*>(xq + r * ldq + c0) = o;
    }
    if (amax) {   // one atomic per workgroup (|x| as float bits orders like the unsigned integer)
        __shared__ float red[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            if (mx > 0.f) atomicMax(amax, __float_as_uint(mx));
        }
    }
}

__global__ void fp8_roll_kernel(int* exp, unsigned* amax, const int* fmt, int n, int margin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = __uint_as_float(amax[i]);
    if (a > 0.f) exp[i] = exp_for(a, fmt_top(fmt ? fmt[i] : BG_FP8_E4M3), margin);   // a site not visited this step keeps its exponent
    amax[i] = 0u;
}

// per-layer max |w| over the dense fp32 master layout [K][RS][C]
__global__ __launch_bounds__(256) void layer_amax_kernel(const float* __restrict__ src, const long long* __restrict__ tbl, unsigned* __restrict__ amax) {
    const long long* e = tbl + (long long)blockIdx.y * 8;
    const long long so = e[0], n = e[3] * e[4] * e[5];
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) mx = fmaxf(mx, fabsf(src[so + i]));
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (mx > 0.f) atomicMax(amax + blockIdx.y, __float_as_uint(mx));
    }
}

__device__ __forceinline__ unsigned char cvt1_e4m3(float v) {
    v = fminf(fmaxf(v, -448.f), 448.f);
    return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false) & 0xff);
}

__global__ void pack_conv_weights_fp8_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst_k, unsigned char* __restrict__ dst_t,
                                             const long long* __restrict__ tbl, const unsigned* __restrict__ amax, int* __restrict__ exps) {
    const long long* e = tbl + (long long)blockIdx.y * 8;
    const long long so = e[0], dk = e[1], dt = e[2], K = e[3], RS = e[4], C = e[5], Cp = e[6], Kp = e[7];
    const long long nk = K * RS * Cp, nt = C * RS * Kp;
    const int ex = exp_for(__uint_as_float(amax[blockIdx.y]), 9, 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) exps[blockIdx.y] = ex;
    const float scale = ldexpf(1.f, ex);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nk + nt; i += (long long)gridDim.x * blockDim.x) {
        if (i < nk) {
            const long long c = i % Cp, t = i / Cp;  // t = k*RS + rs
            dst_k[dk + i] = c < C ? cvt1_e4m3(src[so + t * C + c] * scale) : 0;
        } else {
            const long long j = i - nk;
            const long long k = j % Kp, t = j / Kp;
            const long long rs = t % RS, c = t / RS;
            dst_t[dt + j] = k < K ? cvt1_e4m3(src[so + (k * RS + rs) * C + c] * scale) : 0;
        }
    }
}

template <typename T>
int launch_fp8(GemmConvParams P, hipStream_t st) {
    P.CKp = pad_k(P.CK, BG_FP8);
    const long long w_bytes = (long long)P.NO * P.KH * P.KW * P.CKp;
    if (w_bytes >= (1LL << 31)) {
        bg_set_error("fp8 conv: weight copy larger than 2 GiB");
        return BG_E_ARG;
    }
    P.in_bytes = 0;   // the fat kernel rebases its activation descriptor per tile
    P.w_bytes = (int)w_bytes;
    const int plan = plan_fat<T>(P, true);
    if (!plan) {
        bg_set_error("fp8 conv: no fat-tile plan (statistic groups must divide the pixels)");
        return BG_E_ARG;
    }
    const int tm = plan / 1000, tnp = plan % 1000;
    const long long nblk = (long long)P.tiles_c * P.tiles_p;
    const bool pw1 = P.KH * P.KW == 1;
    if (tm == 128) {
        if (pw1) return launch_fat<T, 128, 2, 7, 4, 2, 3, true>(P, nblk, st);
        return launch_fat<T, 128, 2, 7, 4, 2, 3, false>(P, nblk, st);
    }
    if (tnp == 112) {
        if (pw1) return launch_fat<T, 128, 2, 7, 8, 1, 3, true>(P, nblk, st);
        return launch_fat<T, 128, 2, 7, 8, 1, 3, false>(P, nblk, st);
    }
    if (pw1) return launch_fat<T, 128, 4, 7, 4, 2, 2, true>(P, nblk, st);
    return launch_fat<T, 128, 4, 7, 4, 2, 2, false>(P, nblk, st);
}

int check_fp8_desc(const bg_conv_desc* d, const char* who, bool fwd) {
    BG_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
    BG_CHECK_ARG(d->dtype == BG_BF16, "%s: d->dtype names the bf16 output", who);
    BG_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->Ho > 0 && d->Wo > 0, "%s: non-positive dimension", who);
    BG_CHECK_ARG(d->KH >= 1 && d->KW >= 1 && d->KH <= 7 && d->KW <= 7 && d->stride >= 1 && d->dil >= 1 && d->pad >= 0, "%s: bad window", who);
    // the fp8 side: 16 one-byte elements per vector; the bf16 side: 8
    const int vq = 16, vo = 8;
    const int cq = fwd ? d->Cin : d->Cout, ldq = fwd ? d->ldx : d->ldy, co = fwd ? d->Cout : d->Cin, ldo = fwd ? d->ldy : d->ldx;
    BG_CHECK_ARG(cq % vq == 0 && ldq % vq == 0 && ldq >= cq, "%s: fp8 operand needs channels / pixel stride in multiples of 16 (got %d / %d)", who, cq, ldq);
    BG_CHECK_ARG(co % vo == 0 && ldo % vo == 0 && ldo >= co, "%s: bf16 result needs channels / pixel stride in multiples of 8 (got %d / %d)", who, co, ldo);
    const int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
    BG_CHECK_ARG(ho == d->Ho && wo == d->Wo, "%s: output %dx%d does not match conv arithmetic %dx%d", who, d->Ho, d->Wo, ho, wo);
    BG_CHECK_ARG((long long)d->N * d->H * d->W < (1LL << 31) && (long long)d->N * d->Ho * d->Wo < (1LL << 31), "%s: tensor too large", who);
    return BG_OK;
}

}  // namespace

extern "C" int bg_quant_fp8(int32_t src_dtype, const void* x, int32_t ldx, int64_t rows, int32_t C, void* xq, int32_t ldq, int32_t Cq,
                            int32_t fmt, const int32_t* exp, uint32_t* amax, void* stream) {
    BG_CHECK_ARG(dtype_ok(src_dtype) && x && xq && aligned16(x) && aligned16(xq) && rows > 0 && C > 0, "bg_quant_fp8: bad arguments");
    BG_CHECK_ARG(C % dtype_vec(src_dtype) == 0 && ldx % dtype_vec(src_dtype) == 0 && ldx >= C, "bg_quant_fp8: source channels / stride must be whole 16-byte vectors");
    BG_CHECK_ARG(Cq % 16 == 0 && Cq >= C && Cq < C + 16 && ldq % 16 == 0 && ldq >= Cq, "bg_quant_fp8: Cq must be C rounded up to 16, ldq a multiple of 16 (C=%d Cq=%d ldq=%d)", C, Cq, ldq);
    BG_CHECK_ARG(fmt == BG_FP8_E4M3 || fmt == BG_FP8_E5M2, "bg_quant_fp8: bad format %d", fmt);
    const int cpr = Cq / 16;
    const long long total = rows * cpr;
    const unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, 2048);
    hipStream_t st = (hipStream_t)stream;
#define BG_QUANT(T, F) hipLaunchKernelGGL((quant_fp8_kernel<T, F>), dim3(blocks), dim3(256), 0, st, (const T*)x, ldx, (long long)rows, C, \
                                          (unsigned char*)xq, ldq, cpr, exp, amax)
    if (src_dtype == BG_BF16) { if (fmt == BG_FP8_E4M3) BG_QUANT(bf16_t, BG_FP8_E4M3); else BG_QUANT(bf16_t, BG_FP8_E5M2); }
    else { if (fmt == BG_FP8_E4M3) BG_QUANT(float, BG_FP8_E4M3); else BG_QUANT(float, BG_FP8_E5M2); }
#undef BG_QUANT
    BG_CHECK_LAUNCH("quant_fp8_kernel");
    return BG_OK;
}

extern "C" int bg_fp8_roll(int32_t* exp, uint32_t* amax, const int32_t* fmt, int32_t n, int32_t margin, void* stream) {
    BG_CHECK_ARG(exp && amax && n > 0 && margin >= 0 && margin <= 8, "bg_fp8_roll: bad arguments");
    hipLaunchKernelGGL(fp8_roll_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, exp, amax, fmt, n, margin);
    BG_CHECK_LAUNCH("fp8_roll_kernel");
    return BG_OK;
}

extern "C" int bg_pack_conv_weights_fp8(const float* src, void* dst_krsc, void* dst_crsk, const int64_t* tbl, int32_t n_layers,
                                        int64_t max_elems, int32_t* exps, uint32_t* amax_ws, void* stream) {
    BG_CHECK_ARG(src && dst_krsc && dst_crsk && tbl && exps && amax_ws && n_layers > 0 && n_layers <= 65535 && max_elems > 0,
                 "bg_pack_conv_weights_fp8: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(amax_ws, 0, sizeof(uint32_t) * (size_t)n_layers, st) != hipSuccess) {
        bg_set_error("bg_pack_conv_weights_fp8: memset failed");
        return BG_E_LAUNCH;
    }
    long long bx = (max_elems + 255) / 256;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(layer_amax_kernel, dim3((unsigned)bx, (unsigned)n_layers), dim3(256), 0, st, src, (const long long*)tbl, amax_ws);
    BG_CHECK_LAUNCH("layer_amax_kernel");
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(pack_conv_weights_fp8_kernel, dim3((unsigned)bx, (unsigned)n_layers), dim3(256), 0, st, src, (unsigned char*)dst_krsc,
                       (unsigned char*)dst_crsk, (const long long*)tbl, amax_ws, exps);
    BG_CHECK_LAUNCH("pack_conv_weights_fp8_kernel");
    return BG_OK;
}

extern "C" int bg_conv2d_fwd_fp8(const bg_conv_desc* d, const void* xq, const void* wq, const int32_t* exp_x, const int32_t* exp_w,
                                 const float* bias, void* y, double* sum, double* sumsq, int32_t groups, void* stream) {
    int rc = check_fp8_desc(d, "bg_conv2d_fwd_fp8", true);
    if (rc) return rc;
    BG_CHECK_ARG(xq && wq && y && exp_x && exp_w && aligned16(xq) && aligned16(wq) && aligned16(y), "bg_conv2d_fwd_fp8: null/unaligned pointer");
    BG_CHECK_ARG((sum == nullptr) == (sumsq == nullptr) && !(sum && bias), "bg_conv2d_fwd_fp8: statistics need both accumulators and no bias");
    const long long M_ = (long long)d->N * d->Ho * d->Wo;
    BG_CHECK_ARG(groups >= 1 && M_ % groups == 0, "bg_conv2d_fwd_fp8: groups must divide the pixels");
    GemmConvParams P{};
    P.in = xq; P.w = wq; P.out = y; P.bias = bias;
    P.N = d->N; P.IH = d->H; P.IW = d->W; P.OH = d->Ho; P.OW = d->Wo;
    P.CK = d->Cin; P.NO = d->Cout; P.ldi = d->ldx; P.ldo = d->ldy;
    P.KH = d->KH; P.KW = d->KW; P.stride = d->stride; P.pad = d->pad; P.dil = d->dil;
    P.transposed = 0;
    P.M = M_;
    P.stat_sum = sum; P.stat_sq = sumsq;
    P.stat_group_pix = groups > 1 ? (int)(M_ / groups) : 0;
    P.exp_w = exp_w; P.exp_act = exp_x;
    return launch_fp8<f8e4_t>(P, (hipStream_t)stream);
}

extern "C" int bg_conv2d_bwd_data_fp8(const bg_conv_desc* d, const void* dyq, int32_t dy_fmt, const void* wtq, const int32_t* exp_dy,
                                      const int32_t* exp_w, void* dx, void* stream) {
    int rc = check_fp8_desc(d, "bg_conv2d_bwd_data_fp8", false);
    if (rc) return rc;
    BG_CHECK_ARG(dyq && wtq && dx && exp_dy && exp_w && aligned16(dyq) && aligned16(wtq) && aligned16(dx), "bg_conv2d_bwd_data_fp8: null/unaligned pointer");
    BG_CHECK_ARG(dy_fmt == BG_FP8_E4M3 || dy_fmt == BG_FP8_E5M2, "bg_conv2d_bwd_data_fp8: bad format %d", dy_fmt);
    GemmConvParams P{};
    P.in = dyq; P.w = wtq; P.out = dx; P.bias = nullptr;
    P.N = d->N; P.IH = d->Ho; P.IW = d->Wo; P.OH = d->H; P.OW = d->W;
    P.CK = d->Cout; P.NO = d->Cin; P.ldi = d->ldy; P.ldo = d->ldx;
    P.KH = d->KH; P.KW = d->KW; P.stride = d->stride; P.pad = d->pad; P.dil = d->dil;
    P.transposed = 1;
    P.M = (long long)d->N * d->H * d->W;
    P.exp_w = exp_w; P.exp_act = exp_dy;
    if (dy_fmt == BG_FP8_E5M2) return launch_fp8<f8e5_t>(P, (hipStream_t)stream);
    return launch_fp8<f8e4_t>(P, (hipStream_t)stream);
}
