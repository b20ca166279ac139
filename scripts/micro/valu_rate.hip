// Micro-benchmark (diagnostic, not shipped): issue cost of the vector instructions an element-wise bf16 kernel is made
// of, at 1 / 2 / 4 waves per SIMD, every CU busy.  Decides the "instruction diet" of the depthwise kernels:
// is v_pk_fma_f32 cheaper than two v_fma_f32, is v_dot2_f32_bf16 (two bf16 products per lane, no unpack) full rate?
// Prints SIMD cycles per wave-instruction (in-kernel s_memtime over the loop / instructions / waves per SIMD).
//   hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum Op { FMA, PKFMA, PKMUL, PKADD, DOT2, DOT2C, CVTPK, PERM, AND, LSHL, MAX, MUL, CNDMASK, MIX8, DSREAD64, NOPS };
static const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_dot2_f32_bf16", "v_dot2c_f32_bf16",
                              "v_cvt_pk_bf16_f32", "v_perm_b32", "v_and_b32", "v_lshlrev_b32", "v_max_f32", "v_mul_f32",
                              "v_cndmask_b32", "mix: 4 perm + 4 dot2", "ds_read_b64"};

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ void k_rate(int iters, long long* cyc, float* sink) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x;
    float a[8], b = 1.0001f + tid * 1e-7f, c = 0.5f;
    f32x2 p[8], q = {1.0001f, 0.9999f}, r = {0.5f, 0.25f};
    unsigned u[8], sel = 0x05040100u;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = i + tid * 1e-3f; p[i] = f32x2{(float)i, (float)tid}; u[i] = tid * 2654435761u + i; }
    lds[tid] = b; lds[tid + 1024] = c;
    __syncthreads();
    const int ldsa = (tid & 63) * 8;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (OP == FMA) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        } else if (OP == PKFMA) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(q), "v"(r));
            REP8(S) REP8(S)
#undef S
        } else if (OP == PKMUL) {
#define S(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q));
            REP8(S) REP8(S)
#undef S
        } else if (OP == PKADD) {
#define S(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(r));
            REP8(S) REP8(S)
#undef S
        } else if (OP == DOT2) {
#define S(i) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(u[i]), "v"(sel));
            REP8(S) REP8(S)
#undef S
        } else if (OP == DOT2C) {
#define S(i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(u[i]), "v"(sel));
            REP8(S) REP8(S)
#undef S
        } else if (OP == CVTPK) {
#define S(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b));
            REP8(S) REP8(S)
#undef S
        } else if (OP == PERM) {
#define S(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(sel));
            REP8(S) REP8(S)
#undef S
        } else if (OP == AND) {
#define S(i) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(a[i]) : "v"(u[i]));
            REP8(S) REP8(S)
#undef S
        } else if (OP == LSHL) {
#define S(i) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(a[i]) : "v"(u[i]));
            REP8(S) REP8(S)
#undef S
        } else if (OP == MAX) {
#define S(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        } else if (OP == MUL) {
#define S(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        } else if (OP == CNDMASK) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            REP8(S) REP8(S)
#undef S
        } else if (OP == MIX8) {
#define S(i) asm volatile("v_perm_b32 %0, %2, %3, %4\n v_dot2_f32_bf16 %1, %0, %4, %1" : "+v"(u[i]), "+v"(a[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]), "v"(sel));
            REP8(S)
#undef S
        } else if (OP == DSREAD64) {
#define S(i) asm volatile("ds_read_b64 %0, %1 offset:" #i "*512" : "=v"(p[i]) : "v"(ldsa));
            REP8(S) REP8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1] + (float)u[i];
    if (s == 1.2345f) sink[0] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
}

template <int OP>
void run(int wps, long long* d_cyc, float* d_sink) {
    const int threads = 256 * wps > 1024 ? 1024 : 256 * wps, blocks_per_cu = (256 * wps) / threads, grid = 256 * blocks_per_cu;
    const int iters = 4096, nw = grid * threads / 64;
    hipLaunchKernelGGL(k_rate<OP>, dim3(grid), dim3(threads), 0, 0, 16, d_cyc, d_sink);   // warm
    hipLaunchKernelGGL(k_rate<OP>, dim3(grid), dim3(threads), 0, 0, iters, d_cyc, d_sink);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(nw);
    CK(hipMemcpy(h.data(), d_cyc, nw * sizeof(long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double per_wave = (double)h[nw / 2] / (iters * 16.0);       // s_memtime ticks (100 MHz-independent: shader cycles) per instruction, one wave
    printf("%-24s waves/SIMD %d : %.2f cyc/instr per wave, %.2f SIMD-cyc per wave-instr\n", names[OP], wps, per_wave, per_wave / wps);
}

int main() {
    long long* d_cyc; float* d_sink;
    CK(hipMalloc(&d_cyc, 1 << 20)); CK(hipMalloc(&d_sink, 64));
    for (int wps : {1, 2, 4}) {
        run<FMA>(wps, d_cyc, d_sink); run<PKFMA>(wps, d_cyc, d_sink); run<PKMUL>(wps, d_cyc, d_sink); run<PKADD>(wps, d_cyc, d_sink);
        run<DOT2>(wps, d_cyc, d_sink); run<DOT2C>(wps, d_cyc, d_sink); run<CVTPK>(wps, d_cyc, d_sink); run<PERM>(wps, d_cyc, d_sink);
        run<AND>(wps, d_cyc, d_sink); run<LSHL>(wps, d_cyc, d_sink); run<MAX>(wps, d_cyc, d_sink); run<MUL>(wps, d_cyc, d_sink);
        run<CNDMASK>(wps, d_cyc, d_sink); run<MIX8>(wps, d_cyc, d_sink); run<DSREAD64>(wps, d_cyc, d_sink);
        printf("\n");
    }
    return 0;
}
