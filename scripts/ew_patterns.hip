// Which streaming access pattern reaches copy speed on MI355X?  y = a*x + b over bf16 [rows, 736] tensors (40 MB and
// 160 MB), buffers rotated so nothing is served from the Infinity Cache.  Build: hipcc -O3 --offload-arch=gfx950
// scripts/ew_patterns.hip -o gpurun_out/ew_patterns ; run on the GPU box.  Experiment only, not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/bgamd.h"

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ inline u32x4 affine(u32x4 v, float a, float b) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = __uint_as_float(v[i] << 16), hi = __uint_as_float(v[i] & 0xffff0000u);
        const float zl = fmaf(lo, a, b), zh = fmaf(hi, a, b);
        const unsigned rl = (__float_as_uint(zl) + 0x7fffu + ((__float_as_uint(zl) >> 16) & 1)) >> 16;
        const unsigned rh = (__float_as_uint(zh) + 0x7fffu + ((__float_as_uint(zh) >> 16) & 1)) & 0xffff0000u;
        o[i] = rl | rh;
    }
    return o;
}

// V0: one 16-byte chunk per thread
__global__ __launch_bounds__(256) void v0(const u32x4* x, u32x4* y, long long n, float a, float b) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = affine(x[i], a, b);
}
// V1: U chunks per thread, block covers a contiguous U*4 KiB span, no loop
template <int U, bool NT>
__global__ __launch_bounds__(256) void v1(const u32x4* x, u32x4* y, long long n, float a, float b) {
    const long long i0 = (long long)blockIdx.x * 256 * U + threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long i = i0 + u * 256;
        if (i < n) v[u] = NT ? __builtin_nontemporal_load(x + i) : x[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long i = i0 + u * 256;
        if (i < n) {
            const u32x4 o = affine(v[u], a, b);
            if (NT) __builtin_nontemporal_store(o, y + i);
            else y[i] = o;
        }
    }
}
// V2: grid-stride (the whole grid sweeps a contiguous window), U chunks in flight
template <int U>
__global__ __launch_bounds__(256) void v2(const u32x4* x, u32x4* y, long long n, float a, float b) {
    const long long span = (long long)gridDim.x * 256 * U;
    for (long long base = (long long)blockIdx.x * 256 * U + threadIdx.x; base < n; base += span) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + u * 256 < n) v[u] = x[base + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + u * 256 < n) y[base + u * 256] = affine(v[u], a, b);
    }
}
// V3: every block marches through its own contiguous range (what norm_act.hip does), U chunks in flight
template <int U>
__global__ __launch_bounds__(256) void v3(const u32x4* x, u32x4* y, long long n, float a, float b) {
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per;
    long long hi = lo + per;
    if (hi > n) hi = n;
    for (long long base = lo + threadIdx.x; base < hi; base += 256 * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + u * 256 < hi) v[u] = x[base + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + u * 256 < hi) y[base + u * 256] = affine(v[u], a, b);
    }
}
// V4: per-thread prologue of 16 dependent-free table loads first (scale/shift per channel), then V1
template <int U>
__global__ __launch_bounds__(256) void v4(const u32x4* x, u32x4* y, long long n, const float* tab, int C) {
    const long long i0 = (long long)blockIdx.x * 256 * U + threadIdx.x;
    const int c = (int)((i0 * 8) % C);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a += tab[c + e];
        b += tab[C + c + e];
    }
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long i = i0 + u * 256;
        if (i < n) v[u] = x[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long i = i0 + u * 256;
        if (i < n) y[i] = affine(v[u], a, b);
    }
}

// V6: the depthwise kernel's READ pattern alone: thread = (4-pixel column group, channel chunk) of an image row band,
// reads 6 consecutive pixels' chunks of R+2 rows (ld 736 elements), no stores unless the sum is a magic value.
template <int R, int COLS>
__global__ __launch_bounds__(256) void v6(const u32x4* x, u32x4* y, int H, int W, float magic) {
    const int cv = 91, ldc = 92;  // chunks per pixel / per pixel stride
    const int wgroups = W / 4, items = wgroups * cv, bx = (items + 255) / 256;
    const int band = blockIdx.x / bx, xblk = blockIdx.x - band * bx;
    const int idx = xblk * 256 + threadIdx.x;
    if (idx >= items) return;
    const int wq = idx / cv, c = idx - wq * cv;
    const int bands = H / R, n = band / bands, b = band - n * bands;
    const long long row_chunks = (long long)W * ldc;
    const u32x4* xn = x + (long long)n * H * row_chunks;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = -1; r <= R; ++r) {
        const int ih = b * R + r;
        if (ih < 0 || ih >= H) continue;
#pragma unroll
        for (int j = 0; j < COLS; ++j) {
            const int col = wq * 4 + j - (COLS > 4 ? 1 : 0);
            if (col < 0 || col >= W) continue;
            acc ^= xn[ih * row_chunks + (long long)col * ldc + c];
        }
    }
    if (__uint_as_float(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == magic) y[idx] = acc;
}

// V7: V6 with every load of the band issued before the first use (clamped addresses instead of branches)
template <int R, int COLS>
__global__ __launch_bounds__(256) void v7(const u32x4* x, u32x4* y, int H, int W, float magic) {
    const int cv = 91, ldc = 92;
    const int wgroups = W / 4, items = wgroups * cv, bx = (items + 255) / 256;
    const int band = blockIdx.x / bx, xblk = blockIdx.x - band * bx;
    const int idx = xblk * 256 + threadIdx.x;
    if (idx >= items) return;
    const int wq = idx / cv, c = idx - wq * cv;
    const int bands = H / R, n = band / bands, b = band - n * bands;
    const long long row_chunks = (long long)W * ldc;
    const u32x4* xn = x + (long long)n * H * row_chunks;
    u32x4 v[R + 2][COLS];
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
        int ih = b * R + r - 1;
        ih = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
#pragma unroll
        for (int j = 0; j < COLS; ++j) {
            int col = wq * 4 + j - (COLS > 4 ? 1 : 0);
            col = col < 0 ? 0 : (col >= W ? W - 1 : col);
            v[r][j] = xn[ih * row_chunks + (long long)col * ldc + c];
        }
    }
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < R + 2; ++r)
#pragma unroll
        for (int j = 0; j < COLS; ++j) acc ^= v[r][j];
    if (__uint_as_float(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == magic) y[idx] = acc;
}

#define CK(e)                                                                   \
    do {                                                                        \
        hipError_t r_ = (e);                                                    \
        if (r_ != hipSuccess) {                                                 \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(r_)); \
            exit(1);                                                            \
        }                                                                       \
    } while (0)

int main() {
    const long long sizes[2] = {8LL * 72 * 48 * 736 * 2, 8LL * 144 * 96 * 736 * 2};
    for (int s = 0; s < 2; ++s) {
        const long long bytes = sizes[s], n = bytes / 16;
        const int nb = (int)(1.5e9 / bytes) < 2 ? 2 : (int)(1.5e9 / bytes);
        std::vector<void*> xs(nb), ys(nb);
        for (int i = 0; i < nb; ++i) {
            CK(hipMalloc(&xs[i], bytes));
            CK(hipMalloc(&ys[i], bytes));
            CK(hipMemset(xs[i], 0x3c, bytes));
        }
        float* tab;
        CK(hipMalloc(&tab, 2 * 736 * 4));
        CK(hipMemset(tab, 0, 2 * 736 * 4));
        hipStream_t st;
        CK(hipStreamCreate(&st));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        int ctr = 0;
        auto run = [&](const char* name, auto&& launch) {
            const int iters = 40;
            for (int i = 0; i < 3; ++i) { launch((const u32x4*)xs[ctr % nb], (u32x4*)ys[ctr % nb]); ++ctr; }
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) { launch((const u32x4*)xs[ctr % nb], (u32x4*)ys[ctr % nb]); ++ctr; }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            printf("%6.0f MB  %-28s %7.1f us  %6.0f GB/s\n", bytes / 1e6, name, us, 2.0 * bytes / us * 1e-3);
            CK(hipGetLastError());
        };
        run("hipMemcpyAsync D2D", [&](const u32x4* x, u32x4* y) { CK(hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, st)); });
        run("v0 1 chunk/thread", [&](const u32x4* x, u32x4* y) { hipLaunchKernelGGL(v0, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n, 1.5f, 0.25f); });
#define V1(U, NT) run("v1 U=" #U " nt=" #NT, [&](const u32x4* x, u32x4* y) { hipLaunchKernelGGL((v1<U, NT>), dim3((unsigned)((n + 256 * U - 1) / (256 * U))), dim3(256), 0, st, x, y, n, 1.5f, 0.25f); })
        V1(2, false); V1(4, false); V1(8, false); V1(4, true); V1(8, true);
#define V2(U, G) run("v2 grid-stride U=" #U " G=" #G, [&](const u32x4* x, u32x4* y) { hipLaunchKernelGGL((v2<U>), dim3(G), dim3(256), 0, st, x, y, n, 1.5f, 0.25f); })
        V2(4, 1024); V2(4, 2048); V2(4, 4096); V2(8, 2048); V2(2, 4096);
#define V3(U, G) run("v3 block-march U=" #U " G=" #G, [&](const u32x4* x, u32x4* y) { hipLaunchKernelGGL((v3<U>), dim3(G), dim3(256), 0, st, x, y, n, 1.5f, 0.25f); })
        V3(4, 1024); V3(4, 2048); V3(4, 4096); V3(8, 2048); V3(1, 1024);
#define V4(U) run("v4 table prologue U=" #U, [&](const u32x4* x, u32x4* y) { hipLaunchKernelGGL((v4<U>), dim3((unsigned)((n + 256 * U - 1) / (256 * U))), dim3(256), 0, st, x, y, n, tab, 736); })
        V4(4); V4(8);
        {   // the library's own entry points in the same harness (no Python in the loop)
            const long long rows = bytes / (736 * 2);
            double* sums;
            CK(hipMalloc(&sums, 4 * 736 * 8));
            CK(hipMemset(sums, 0, 4 * 736 * 8));
            float* f;
            CK(hipMalloc(&f, 8 * 736 * 4));
            CK(hipMemset(f, 0, 8 * 736 * 4));
            run("lib norm_act_fwd_stats", [&](const u32x4* x, u32x4* y) {
                if (bg_norm_act_fwd_stats(BG_BF16, x, 736, sums, sums + 736, f, f + 736, 1e-5f, 0.1f, nullptr, nullptr, f + 2 * 736,
                                          f + 3 * 736, nullptr, 0, y, 736, rows, 728, 1, 1, st)) exit(2);
            });
            run("lib norm_act_fwd (plain)", [&](const u32x4* x, u32x4* y) {
                if (bg_norm_act_fwd(BG_BF16, x, 736, f, f + 736, nullptr, 0, y, 736, rows, 728, 1, 1, st)) exit(2);
            });
            run("lib norm_act_fwd C=736", [&](const u32x4* x, u32x4* y) {
                if (bg_norm_act_fwd(BG_BF16, x, 736, f, f + 736, nullptr, 0, y, 736, rows, 736, 1, 1, st)) exit(2);
            });
            run("lib norm_act_fwd flat C=256", [&](const u32x4* x, u32x4* y) {
                if (bg_norm_act_fwd(BG_BF16, x, 256, f, f + 736, nullptr, 0, y, 256, rows * 736 / 256, 256, 1, 1, st)) exit(2);
            });
            {   // depthwise 3x3 on the same tensor viewed as 8 x H x W x 728 (ld 736)
                const int W = 48 * (s + 1), H = (int)(rows / 8 / W);
#define V6(R, COLS) run("v6 dw read pattern R=" #R " cols=" #COLS, [&](const u32x4* x, u32x4* y) { \
                    const int bx = ((W / 4) * 91 + 255) / 256; \
                    hipLaunchKernelGGL((v6<R, COLS>), dim3(8 * (H / R) * bx), dim3(256), 0, st, x, y, H, W, 1.2345e-30f); })
                V6(6, 6); V6(6, 4); V6(12, 6); V6(3, 6); V6(72, 4);
#define V7(R, COLS) run("v7 all loads upfront R=" #R " cols=" #COLS, [&](const u32x4* x, u32x4* y) { \
                    const int bx = ((W / 4) * 91 + 255) / 256; \
                    hipLaunchKernelGGL((v7<R, COLS>), dim3(8 * (H / R) * bx), dim3(256), 0, st, x, y, H, W, 1.2345e-30f); })
                V7(6, 6); V7(6, 4); V7(3, 6); V7(2, 6);
                bg_dwconv_desc d{BG_BF16, 8, H, W, 728, H, W, 1, 1, 736, 736};
                void* wdw;
                CK(hipMalloc(&wdw, 9 * 736 * 2));
                CK(hipMemset(wdw, 0x3c, 9 * 736 * 2));
                run("lib dwconv3x3_fwd", [&](const u32x4* x, u32x4* y) {
                    if (bg_dwconv3x3_fwd(&d, x, wdw, y, st)) exit(2);
                });
                run("lib dwconv3x3_fwd_pre", [&](const u32x4* x, u32x4* y) {
                    if (bg_dwconv3x3_fwd_pre(&d, x, f, f + 736, 1, 1, wdw, y, st)) exit(2);
                });
                CK(hipFree(wdw));
            }
            CK(hipFree(sums));
            CK(hipFree(f));
        }
        for (int i = 0; i < nb; ++i) { CK(hipFree(xs[i])); CK(hipFree(ys[i])); }
        CK(hipFree(tab));
    }
    return 0;
}
