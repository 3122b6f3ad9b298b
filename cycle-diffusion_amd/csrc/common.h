// Common device/host helpers for the gfx950 CycleDiffusion engine.
// Everything in csrc/ is written for CDNA4 (wave64, MFMA, 160 KiB LDS) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <stdexcept>
#include <atomic>

namespace cd {

// 16-bit storage format of activations and packed weights. The identifier `bf16_t` is kept for the
// raw 16-bit word throughout csrc/; its FORMAT is chosen at build time:
//   CD_ACT_FP16=1 (default): IEEE fp16 - 10-bit mantissa. The DPM-Encoder divides by sigma_t in
//     [2e-3, 3e-2] and the DDIM chain rescales by sqrt(a_{t-1}/a_t), so storage round-off of eps_hat is
//     amplified 15-130x; fp16 keeps that 8x smaller than bf16 at the same MFMA rate
//     (v_mfma_f32_32x32x16_f16). Stores saturate at +-65504 instead of overflowing to inf.
//   CD_ACT_FP16=0: bfloat16 (v_mfma_f32_32x32x16_bf16).
// Accumulation, normalisation statistics, softmax and all scheduler math are fp32 either way.
#ifndef CD_ACT_FP16
#define CD_ACT_FP16 1
#endif
typedef uint16_t bf16_t;

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#if CD_ACT_FP16
__host__ __device__ inline float bf2f(bf16_t v) {
  _Float16 h;
  __builtin_memcpy(&h, &v, 2);
  return (float)h;
}
__host__ __device__ inline bf16_t f2bf(float f) {
  f = f > 65504.0f ? 65504.0f : (f < -65504.0f ? -65504.0f : f);  // saturate (NaN passes through)
  _Float16 h = (_Float16)f;  // round-to-nearest-even
  bf16_t v;
  __builtin_memcpy(&v, &h, 2);
  return v;
}
__device__ inline void unpack8(const uint4& raw, float* f) {
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bf2f((bf16_t)(w[i] & 0xffffu));
    f[2 * i + 1] = bf2f((bf16_t)(w[i] >> 16));
  }
}
#define CD_MFMA_32x32x16(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(::cd::f16x8, a), __builtin_bit_cast(::cd::f16x8, b), c, 0, 0, 0)
#define CD_MFMA_16x16x32(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(::cd::f16x8, a), __builtin_bit_cast(::cd::f16x8, b), c, 0, 0, 0)
#else
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
__device__ inline void unpack8(const uint4& raw, float* f) {
  f[0] = __uint_as_float(raw.x << 16); f[1] = __uint_as_float(raw.x & 0xffff0000u);
  f[2] = __uint_as_float(raw.y << 16); f[3] = __uint_as_float(raw.y & 0xffff0000u);
  f[4] = __uint_as_float(raw.z << 16); f[5] = __uint_as_float(raw.z & 0xffff0000u);
  f[6] = __uint_as_float(raw.w << 16); f[7] = __uint_as_float(raw.w & 0xffff0000u);
}
#define CD_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define CD_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif

__device__ inline uint32_t pack2(float lo, float hi) {
#if CD_ACT_FP16 && defined(__HIP_DEVICE_COMPILE__) && !defined(CD_PACK8_F2BF)  // two v_med3_f32 + one v_cvt_pk_f16_f32 (see pack8)
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const h2 v = {(_Float16)__builtin_amdgcn_fmed3f(lo, -65504.0f, 65504.0f),
                (_Float16)__builtin_amdgcn_fmed3f(hi, -65504.0f, 65504.0f)};
  return __builtin_bit_cast(uint32_t, v);
#else
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
#endif
}
// two softmax probabilities, known to lie in [0, 2^8] (deferred-maximum attention): one hardware convert, no
// saturation needed
__device__ inline uint32_t pack2_prob(float lo, float hi) {
#if CD_ACT_FP16
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const h2 v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
#else
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;  // plain casts: hipcc emits v_cvt_pk_bf16_f32 and pads its
  const b2 v = {(__bf16)lo, (__bf16)hi};                  // hazards itself (an asm statement would not be padded)
  return __builtin_bit_cast(uint32_t, v);
#endif
}
#if CD_ACT_FP16
constexpr uint32_t kOnePair = 0x3C003C00u;  // two 16-bit 1.0 values
#else
constexpr uint32_t kOnePair = 0x3F803F80u;
#endif
// eight fp32 values -> eight 16-bit storage words. fp16 build: one v_med3_f32 per value (the +-65504 saturation of f2bf)
// and one v_cvt_pk_f16_f32 per pair - 12 VALU operations; the f2bf form compiles to two compare / select pairs, a convert,
// an SDWA convert and an OR per pair, ~50 operations per vector, and the GEMM epilogues are VALU-issue-bound (round 6:
// every k_conv_gemm tile spends 13-40 % of its time in its row passes). Same bits as f2bf for every non-NaN input; a NaN
// saturates instead of passing through.
__device__ inline uint4 pack8(const float* f) {
#if CD_ACT_FP16 && defined(__HIP_DEVICE_COMPILE__) && !defined(CD_PACK8_F2BF)  // (-DCD_PACK8_F2BF: the old form, A/B builds only)
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const h2 v = {(_Float16)__builtin_amdgcn_fmed3f(f[2 * i], -65504.0f, 65504.0f),
                  (_Float16)__builtin_amdgcn_fmed3f(f[2 * i + 1], -65504.0f, 65504.0f)};
    w[i] = __builtin_bit_cast(uint32_t, v);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
#else
  uint4 r;
  r.x = pack2(f[0], f[1]); r.y = pack2(f[2], f[3]);
  r.z = pack2(f[4], f[5]); r.w = pack2(f[6], f[7]);
  return r;
#endif
}

// Sum over the 64 lanes of a wave, result in every lane, without the LDS crossbar: quad_perm / row_half_mirror /
// row_mirror DPP adds give every lane its row-of-16 total, row_bcast15 / row_bcast31 chain the four rows so that row 3
// holds the wave total, v_readlane broadcasts it. 7 VALU operations; a __shfl_xor butterfly is 6 ds_bpermute round
// trips. (v_permlane16/32_swap with both operands the same value is NOT usable from the builtin: the compiler folds
// its two results into one register - measured as a wrong LayerNorm before this form.)
#define CD_DPP_ADD(v, ctrl, row_mask) \
  (v) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), ctrl, row_mask, 0xf, false))
__device__ inline float wave_allsum(float v) {
  CD_DPP_ADD(v, 0xB1, 0xf);   // quad_perm [1 0 3 2]
  CD_DPP_ADD(v, 0x4E, 0xf);   // quad_perm [2 3 0 1]
  CD_DPP_ADD(v, 0x141, 0xf);  // row_half_mirror
  CD_DPP_ADD(v, 0x140, 0xf);  // row_mirror: every lane = total of its 16-lane row
  CD_DPP_ADD(v, 0x142, 0xa);  // row_bcast15 into rows 1 and 3 (the other rows add 0)
  CD_DPP_ADD(v, 0x143, 0xc);  // row_bcast31 into rows 2 and 3: row 3 = total of the wave
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef CD_DPP_ADD

__device__ inline float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// <= 1 ulp exp: the fp32 path and the fp32 time-embedding MLP (the pixel 'ddim' chain amplifies every deviation from
// the reference's fp32 arithmetic by 4-5 orders of magnitude)
__device__ inline float silu_acc(float x) { return x / (1.0f + expf(-x)); }
// exact-erf GELU (reference: F.gelu default, attention.py:44)
__device__ inline float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// exact-erf GELU for the 16-bit epilogues: erfc by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e. far below the
// 2^-11 relative step of the fp16 value the result is stored as): q = poly(t) t exp(-x^2 / 2), t = 1 / (1 + p |x| / sqrt 2), and
//   gelu(x) = 0.5 x (1 + erf(x / sqrt 2)) = max(x, 0) - 0.5 |x| q        (x >= 0: x - 0.5 x q;  x < 0: -0.5 |x| q)
// - no select, the 0.5 lives in the polynomial's coefficients, |x| is an operand modifier, and the negative tail does not
// cancel. Round 6: 13 VALU operations (two of them quarter-rate: v_rcp_f32, v_exp_f32) instead of 16 - the GEGLU
// projections are 19 % of the GEMM time and VALU-bound in their epilogues; ~35 for erff.
// Measured against fp64 erf over [-12, 12]: |error| <= 3.4e-7 absolute.
__device__ inline float gelu_fast(float x) {
#pragma clang fp contract(off)  // the same bits wherever it is inlined (conv_gemm.hip and lin_stream.hip must agree)
#ifdef CD_GELU_SELECT_FORM  // rounds 3-5: 0.5 x (x < 0 ? q : 2 - q), 16 operations (A/B builds only: build.py --geluold)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float tt = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float pp = __builtin_fmaf(tt, 1.061405429f, -1.453152027f);
  pp = __builtin_fmaf(tt, pp, 1.421413741f);
  pp = __builtin_fmaf(tt, pp, -0.284496736f);
  pp = __builtin_fmaf(tt, pp, 0.254829592f);
  const float q = pp * tt * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
  return 0.5f * x * (x < 0.0f ? q : 2.0f - q);
#endif
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.231641888f, ax, 1.0f));  // 0.3275911 / sqrt 2
  float poly = __builtin_fmaf(t, 0.5307027145f, -0.7265760135f);                   // A-S coefficients, halved
  poly = __builtin_fmaf(t, poly, 0.7107068705f);
  poly = __builtin_fmaf(t, poly, -0.142248368f);
  poly = __builtin_fmaf(t, poly, 0.127414796f);
  const float e = __builtin_amdgcn_exp2f(-(x * x) * 0.72134752f);                  // exp(-x^2 / 2)
  return fmaxf(x, 0.0f) - poly * t * e * ax;
}

// Two / eight of them at a time for the GEGLU epilogues (conv_gemm.hip, lin_stream.hip), which are VALU-issue-bound: the
// polynomial, the products and the final difference on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two
// lanes' worth per issue slot) - 15 full-rate operations + 4 transcendentals per PAIR instead of 22 + 4. Same operations in
// the same order as gelu_fast on each element: the same bits.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ inline f32x2 gelu_fast2(f32x2 x) {
#pragma clang fp contract(off)
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 d = __builtin_elementwise_fma((f32x2){0.231641888f, 0.231641888f}, ax, (f32x2){1.0f, 1.0f});
  const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2 poly = __builtin_elementwise_fma(t, (f32x2){0.5307027145f, 0.5307027145f}, (f32x2){-0.7265760135f, -0.7265760135f});
  poly = __builtin_elementwise_fma(t, poly, (f32x2){0.7107068705f, 0.7107068705f});
  poly = __builtin_elementwise_fma(t, poly, (f32x2){-0.142248368f, -0.142248368f});
  poly = __builtin_elementwise_fma(t, poly, (f32x2){0.127414796f, 0.127414796f});
  const f32x2 xx = x * x;
  const f32x2 arg = -xx * (f32x2){0.72134752f, 0.72134752f};
  const f32x2 e = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
  const f32x2 mx = {fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f)};
  return mx - poly * t * e * ax;
}
// out[e] = val[e] * gelu(gate[e]), e = 0 .. 7
__device__ inline void mul_gelu8(const float* val, const float* gate, float* out) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 g = {gate[2 * i], gate[2 * i + 1]}, v = {val[2 * i], val[2 * i + 1]};
    const f32x2 r = v * gelu_fast2(g);
    out[2 * i] = r[0];
    out[2 * i + 1] = r[1];
  }
}

// out[e] = val[e] * gelu(gate[e]), e = 0 .. 3 (the same operations per element as mul_gelu8)
__device__ inline void mul_gelu4(const float* val, const float* gate, float* out) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x2 g = {gate[2 * i], gate[2 * i + 1]}, v = {val[2 * i], val[2 * i + 1]};
    const f32x2 r = v * gelu_fast2(g);
    out[2 * i] = r[0];
    out[2 * i + 1] = r[1];
  }
}

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

// Once per DEVICE and process: kernel function attributes (hipFuncSetAttribute) belong to the device that is current when
// they are set, and one process may drive engines on several GPUs (and on several host threads). The body is idempotent,
// so two threads racing through it on the same device is harmless; the fast path is one relaxed device query and a load.
struct PerDeviceOnce {
  std::atomic<unsigned long long> done{0};
  template <class F>
  void operator()(F&& body) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    body();
    done.fetch_or(bit, std::memory_order_release);
  }
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace cd
