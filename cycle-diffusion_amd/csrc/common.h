// Common device/host helpers for the gfx950 CycleDiffusion engine.
// Everything in csrc/ is written for CDNA4 (wave64, MFMA, 160 KiB LDS) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <stdexcept>

namespace cd {

typedef uint16_t bf16_t;  // raw bf16 bits; all activations inside the engine are NHWC bf16

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -------------------------------------
__host__ __device__ inline float bf2f(bf16_t v) {
  union { uint32_t u; float f; } x;
  x.u = ((uint32_t)v) << 16;
  return x.f;
}
__host__ __device__ inline bf16_t f2bf(float f) {
  union { uint32_t u; float f; } x;
  x.f = f;
  uint32_t u = x.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

struct alignas(16) bf16x8_u {
  bf16_t v[8];
};

__device__ inline void unpack8(const uint4& raw, float* f) {
  f[0] = __uint_as_float(raw.x << 16); f[1] = __uint_as_float(raw.x & 0xffff0000u);
  f[2] = __uint_as_float(raw.y << 16); f[3] = __uint_as_float(raw.y & 0xffff0000u);
  f[4] = __uint_as_float(raw.z << 16); f[5] = __uint_as_float(raw.z & 0xffff0000u);
  f[6] = __uint_as_float(raw.w << 16); f[7] = __uint_as_float(raw.w & 0xffff0000u);
}
__device__ inline uint32_t pack2(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ inline uint4 pack8(const float* f) {
  uint4 r;
  r.x = pack2(f[0], f[1]); r.y = pack2(f[2], f[3]);
  r.z = pack2(f[4], f[5]); r.w = pack2(f[6], f[7]);
  return r;
}

__device__ inline float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact-erf GELU (reference: F.gelu default, attention.py:44)
__device__ inline float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// ---- error handling: no exception crosses the C ABI -----------------------------------------
struct Error : public std::runtime_error {
  explicit Error(const std::string& m) : std::runtime_error(m) {}
};

#define CD_CHECK(cond, ...)                                                      \
  do {                                                                           \
    if (!(cond)) {                                                               \
      char _b[512];                                                              \
      snprintf(_b, sizeof(_b), __VA_ARGS__);                                     \
      throw ::cd::Error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + \
                        ": " + _b);                                              \
    }                                                                            \
  } while (0)

#define HIP_CHECK(expr)                                                              \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      throw ::cd::Error(std::string(__FILE__) + ":" + std::to_string(__LINE__) +     \
                        ": HIP error " + hipGetErrorString(_e) + " in " #expr);      \
    }                                                                                \
  } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace cd
