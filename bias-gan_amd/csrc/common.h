// Shared device/host helpers for libbgamd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include "../../include/bgamd.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define LRELU_SLOPE 0.2f

// ---- error plumbing ---------------------------------------------------------
void bg_set_error(const char* fmt, ...);

#define BG_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            bg_set_error(__VA_ARGS__);     \
            return BG_E_ARG;               \
        }                                  \
    } while (0)

#define BG_CHECK_LAUNCH(name)                                                  \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            bg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return BG_E_LAUNCH;                                                \
        }                                                                      \
    } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int dtype_size(int dt) { return dt == BG_BF16 ? 2 : 4; }
static inline int dtype_vec(int dt) { return dt == BG_BF16 ? 8 : 4; }  // elements per 16-byte vector
static inline bool dtype_ok(int dt) { return dt == BG_BF16 || dt == BG_F32; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- element traits -----------------------------------------------------------
template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr int VEC = 4;  // elements per 16-byte chunk
    typedef f32x4 vec_t;
    __device__ static inline float to_f(float v) { return v; }
    __device__ static inline float from_f(float v) { return v; }
};
template <>
struct Elem<bf16_t> {
    static constexpr int VEC = 8;
    typedef bf16x8 vec_t;
    __device__ static inline float to_f(bf16_t v) { return (float)v; }
    __device__ static inline bf16_t from_f(float v) { return (bf16_t)v; }
};

// A 16-byte chunk of T viewed as floats.
template <typename T>
struct Chunk {
    static constexpr int VEC = Elem<T>::VEC;
    typename Elem<T>::vec_t v;
    __device__ inline void load(const T* p) { v = *reinterpret_cast<const typename Elem<T>::vec_t*>(p); }
    __device__ inline void store(T* p) const { *reinterpret_cast<typename Elem<T>::vec_t*>(p) = v; }
    __device__ inline float get(int i) const { return Elem<T>::to_f(v[i]); }
    __device__ inline void set(int i, float f) { v[i] = Elem<T>::from_f(f); }
    __device__ inline void zero() {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = Elem<T>::from_f(0.f);
    }
};

// num_records of a buffer descriptor whose base has been moved forward inside a tensor: the bytes left behind the base,
// clamped to [0, 2^31 - 1].  The LOWER clamp matters: a base at or beyond the tensor's end gives a negative remainder,
// which the hardware reads as ~4 G records -- every offset, the out-of-range marker 0x80000000 included, would then be
// in range (the fault fixed in 9d5668e; tests: test_descriptor_ranges_past_the_tensor_end).
__device__ __host__ inline int bg_records(long long rem) {
    return (int)(rem < 0 ? 0 : rem < 0x7fffffffLL ? rem : 0x7fffffffLL);
}

__device__ inline float lrelu_f(float z) { return z >= 0.f ? z : LRELU_SLOPE * z; }
// activation codes of the norm_act entry points: 0 none, 1 LeakyReLU(0.2), 2 ReLU (PCBActiv3d, infill3d.py:103-106)
__device__ inline float act_slope(int act) { return act == 2 ? 0.f : LRELU_SLOPE; }
// slope s such that act(z) == max(z, s*z): identity 1, LeakyReLU 0.2, ReLU 0
__device__ inline float act_max_slope(int act) { return act == 0 ? 1.f : act_slope(act); }
__device__ inline float lrelu_grad_from_out(float y) { return y >= 0.f ? 1.f : LRELU_SLOPE; }

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// dispatch helper
#define BG_DISPATCH_DTYPE(dt, T, ...)   \
    do {                                \
        if ((dt) == BG_BF16) {          \
            typedef bf16_t T;           \
            __VA_ARGS__;                \
        } else {                        \
            typedef float T;            \
            __VA_ARGS__;                \
        }                               \
    } while (0)
