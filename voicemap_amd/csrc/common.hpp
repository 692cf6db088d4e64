// Shared device/host helpers for libvoicemap_hip.so (gfx950 only -- no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/voicemap_hip.h"

namespace vm {

// ---- error plumbing -----------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define VM_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            vm::set_error(__VA_ARGS__);       \
            return VM_ERR_ARG;                \
        }                                     \
    } while (0)

// ---- storage types ------------------------------------------------------------------------------
typedef __bf16 bf16;
typedef _Float16 f16;  // IEEE half: the second 16-bit storage type (dtype VM_F16) -- 11 significand bits against bf16's 8
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// fp32 -> bf16 round-to-nearest-even (NaN kept quiet); bit-level so host and device agree.
__host__ __device__ inline uint16_t f2bf_bits(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__host__ __device__ inline float bf_bits2f(uint16_t h) {
    union { float f; uint32_t u; } v;
    v.u = ((uint32_t)h) << 16;
    return v.f;
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kVec = 4;  // elements per 16-byte vector
    __device__ static inline float to_f(float v) { return v; }
    __device__ static inline float from_f(float v) { return v; }
};
template <> struct Elem<bf16> {
    static constexpr int kVec = 8;
    __device__ static inline float to_f(bf16 v) { return (float)v; }
    __device__ static inline bf16 from_f(float v) { return (bf16)v; }
};

template <> struct Elem<f16> {
    static constexpr int kVec = 8;
    __device__ static inline float to_f(f16 v) { return (float)v; }
    __device__ static inline f16 from_f(float v) { return (f16)v; }  // v_cvt_f16_f32, round-to-nearest-even
};

// 16-byte vector of T <-> floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    f32x4 v;
    __device__ inline float get(int i) const { return v[i]; }
    __device__ inline void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16> {
    bf16x8 v;
    __device__ inline float get(int i) const { return (float)v[i]; }
    __device__ inline void set(int i, float x) { v[i] = (bf16)x; }
};

template <> struct Vec16<f16> {
    f16x8 v;
    __device__ inline float get(int i) const { return (float)v[i]; }
    __device__ inline void set(int i, float x) { v[i] = (f16)x; }
};

template <typename T>
__device__ inline Vec16<T> load16(const T* p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T>
__device__ inline void store16(T* p, const Vec16<T>& r) {
    *reinterpret_cast<decltype(r.v)*>(p) = r.v;
}

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

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

int f1_set_blocks(int v);  // conv1_fused.hip tuning
int f1_set_fwd_blocks(int v);

// 16-bit storage types only (bf16 / f16): the LDS-DMA kernels and the fused block-1 kernels
#define VM_DISPATCH_16(dtype, ...)                                  \
    do {                                                            \
        if ((dtype) == VM_BF16) {                                   \
            using T = vm::bf16;                                     \
            __VA_ARGS__;                                            \
        } else {                                                    \
            using T = vm::f16;                                      \
            __VA_ARGS__;                                            \
        }                                                           \
    } while (0)

// reduce.hip: out0[i] (i < n0) / out1[i - n0] = sum over `slabs` slabs of ws[k][i], two fixed-order stages.
// `part` needs slab_sum_part_bytes(nel) bytes of scratch.
constexpr int SLAB_RCH = 16;
int64_t slab_sum_part_bytes(int64_t nel);
int slab_sum(const float* ws, int64_t slabs, int64_t nel, float* out0, int64_t n0, float* out1, void* part, hipStream_t stream);

}  // namespace vm

#define VM_DISPATCH_DTYPE(dtype, ...)                               \
    do {                                                            \
        if ((dtype) == VM_F32 || (dtype) == VM_F32S) {              \
            using T = float;                                        \
            __VA_ARGS__;                                            \
        } else if ((dtype) == VM_BF16) {                            \
            using T = vm::bf16;                                     \
            __VA_ARGS__;                                            \
        } else if ((dtype) == VM_F16) {                             \
            using T = vm::f16;                                      \
            __VA_ARGS__;                                            \
        } else {                                                    \
            vm::set_error("unknown dtype %d", (int)(dtype));        \
            return VM_ERR_ARG;                                      \
        }                                                           \
    } while (0)
