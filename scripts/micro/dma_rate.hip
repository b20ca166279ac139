// Micro-benchmark (diagnostic, not shipped): how fast can ONE workgroup per CU stream L2-missing rows into the CU --
// through the LDS-DMA path (buffer_load ... lds, 16 B per lane) against plain buffer loads into VGPRs followed by
// ds_write_b128 -- with the access pattern of the grouped weight gradient (a piece = 2 rows x 512 B, row pitch 1456 B;
// 32 pieces = 32 KiB per K-step and workgroup).  Prints GB/s per CU and chip-wide for a grid of G workgroups.
//   hipcc -O3 --offload-arch=gfx950 dma_rate.hip -o dma_rate && ./dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_dst, 16, voffset, 0, 0, 0);
}
__device__ __forceinline__ i32x4 ld16(__amdgpu_buffer_rsrc_t rsrc, int voffset) { return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, 0, 0); }

struct Prm { const char* base; long long wg_stride; int steps; int pitch; int rowb; int rows_per_piece; int* sink; };

template <int NBUF, int PIECES>   // PIECES per wave and K-step; 8 waves
__global__ __launch_bounds__(512) void k_dma(Prm P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lanes_per_row = P.rowb / 16;                  // 32 for 512-B rows
    const int rsub = lane / lanes_per_row, slot = lane % lanes_per_row;
    const int rows_step = 8 * PIECES * P.rows_per_piece;    // rows per K-step
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.base + (long long)blockIdx.x * P.wg_stride), 0, 0x7fffffff, 0x00020000);
    int v[PIECES];
#pragma unroll
    for (int k = 0; k < PIECES; ++k) v[k] = ((wave * PIECES + k) * P.rows_per_piece + rsub) * P.pitch + slot * 16;
    const int step = rows_step * P.pitch;
    auto issue = [&](int buf) {
#pragma unroll
        for (int k = 0; k < PIECES; ++k) {
            dma16(rs, smem + buf * (8 * PIECES * 1024) + (wave * PIECES + k) * 1024, v[k]);
            v[k] += step;
        }
    };
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s) issue(s);
    int buf = NBUF - 1;
    int acc = 0;
    for (int kt = 0; kt < P.steps; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * PIECES) : "memory");
        __builtin_amdgcn_s_barrier();
        acc += *(volatile int*)(smem + ((kt % NBUF) * 8 * PIECES * 1024) + tid * 4);
        issue(buf);
        buf = buf + 1 == NBUF ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678) P.sink[0] = acc;
}

template <int DEPTH, int PIECES>   // DEPTH K-steps of loads in flight in registers; stored to LDS when they arrive
__global__ __launch_bounds__(512) void k_vgpr(Prm P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lanes_per_row = P.rowb / 16;
    const int rsub = lane / lanes_per_row, slot = lane % lanes_per_row;
    const int rows_step = 8 * PIECES * P.rows_per_piece;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.base + (long long)blockIdx.x * P.wg_stride), 0, 0x7fffffff, 0x00020000);
    int v[PIECES];
#pragma unroll
    for (int k = 0; k < PIECES; ++k) v[k] = ((wave * PIECES + k) * P.rows_per_piece + rsub) * P.pitch + slot * 16;
    const int step = rows_step * P.pitch;
    i32x4 r[DEPTH][PIECES];
    auto issue = [&](int d) {
#pragma unroll
        for (int k = 0; k < PIECES; ++k) {
            r[d][k] = ld16(rs, v[k]);
            v[k] += step;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d);
    int acc = 0;
    for (int kt = 0; kt < P.steps; kt += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PIECES) : "memory");
#pragma unroll
            for (int k = 0; k < PIECES; ++k)
                *(i32x4*)(smem + ((kt + d) & 1) * (8 * PIECES * 1024) + (wave * PIECES + k) * 1024 + lane * 16) = r[d][k];
            issue(d);
            __builtin_amdgcn_s_barrier();
            acc += *(volatile int*)(smem + ((kt + d) & 1) * 8 * PIECES * 1024 + tid * 4);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678) P.sink[0] = acc;
}

template <typename F>
static float time_it(F&& launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const int steps = 720;
    const long long wg_bytes = 128ll << 20;         // each workgroup streams its own 64 MiB window (47 MB used)
    char* buf; int* sink;
    CK(hipMalloc(&buf, 256 * wg_bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, 256 * wg_bytes));
    struct Pat { const char* name; int pitch, rowb, rpp; };
    const Pat pats[] = {{"2 rows x 512 B, pitch 1456", 1456, 512, 2}, {"2 rows x 512 B, pitch 1536", 1536, 512, 2},
                        {"1 row x 1024 B, contiguous ", 1024, 1024, 1}};
    for (const Pat& pt : pats)
        for (int grid : {256, 128, 32, 8}) {
            Prm P{buf, wg_bytes, steps, pt.pitch, pt.rowb, pt.rpp, sink};
            // a K-step covers 8 waves x 4 pieces x rpp rows; window check
            const long long span = (long long)steps * 32 * pt.rpp * pt.pitch;
            if (span + (1 << 20) > wg_bytes) { printf("window too small for %s\n", pt.name); continue; }
            const double bytes = (double)grid * steps * 32768.0;
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<5, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 32768));
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vgpr<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vgpr<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
            const float a = time_it([&] { hipLaunchKernelGGL((k_dma<4, 4>), dim3(grid), dim3(512), 4 * 32768, 0, P); });
            const float a5 = time_it([&] { hipLaunchKernelGGL((k_dma<5, 4>), dim3(grid), dim3(512), 5 * 32768, 0, P); });
            const float b = time_it([&] { hipLaunchKernelGGL((k_vgpr<2, 4>), dim3(grid), dim3(512), 4 * 32768, 0, P); });
            const float c = time_it([&] { hipLaunchKernelGGL((k_vgpr<4, 4>), dim3(grid), dim3(512), 4 * 32768, 0, P); });
            auto show = [&](const char* n, float ms) { printf("  %-22s %8.1f us  %6.1f GB/s per CU  %6.2f TB/s chip\n", n, ms * 1e3, bytes / grid / ms * 1e-6, bytes / ms * 1e-9); };
            printf("%s, %d workgroups of 512 (128 KiB LDS: one per CU), %d K-steps of 32 KiB:\n", pt.name, grid, steps);
            show("LDS-DMA, 3 ahead", a); show("LDS-DMA, 4 ahead", a5); show("VGPR, 2 ahead", b); show("VGPR, 4 ahead", c);
        }
    return 0;
}
