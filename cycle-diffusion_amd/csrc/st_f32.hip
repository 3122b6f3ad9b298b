// fp32 kernels of the SpatialTransformer blocks (CD_PREC_F32 / CD_PREC_F32X3 on the text-conditioned U-Nets).
//
// The reference evaluates Stable Diffusion / LDM text2img at `precision = "full"`
// (stable_diffusion_stochastic_text_wrapper.py:117): BasicTransformerBlock (attention.py:196-215) = LayerNorm ->
// self-attention -> LayerNorm -> cross-attention over the text context -> LayerNorm -> GEGLU feed-forward, all fp32. The
// 16-bit engine closes a 99-step encode -> decode cycle to 2e-2 rms (the reference: 1.6e-5); this file supplies what the
// fp32 execution path (f32_path.hip) lacked for these blocks:
//   k_flash_f32      softmax(q k^T) v without the score matrix, on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains)
//   k_layernorm_f32  fp32 rows; optionally written as the split mode's fp16 pair (the next GEMM is a three-term product)
//   k_geglu_f32      value * gelu(gate) on a materialised projection (erff: the reference's exact GELU)
#include "common.h"
#include "kernels.h"

namespace cd {

namespace st_f32_detail {

typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) _Float16 h4;

__device__ inline void split_pair(float v, _Float16& hi, _Float16& lo, int* overflow) {
  if (overflow && !(fabsf(v) <= 65504.f)) *overflow = 1;  // also NaN; the engine raises at its next API entry
  v = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
  hi = (_Float16)v;
  lo = (_Float16)(v - (float)hi);
}

// ---------------------------------------------------------------------------------------- attention
// One workgroup = 4 waves = 128 queries of one (image, head); a wave owns 32 queries. Keys arrive 32 at a time in LDS.
// The product is formed transposed, S^T = K Q^T (A operand = K rows from LDS, B operand = Q^T held in registers for the
// whole kernel), so a lane owns ONE query column of the 32 x 32 score block and 16 of its 32 keys: the running maximum and
// the row sum are per-lane registers plus one exchange with lane ^ 32. The accumulator registers of S^T - after exp2 - are
// directly the B operand of the second product O^T = V^T P^T: its 16 two-key steps take the keys in the order the
// accumulator layout holds them (key(s, half) = 8 (s / 4) + 4 half + s % 4) and the A operand reads V rows in the same
// order from LDS, so P never moves between lanes.
// q is expected in log2 units when q_log2 (scale * log2 e folded into the packed to_q rows, as on the 16-bit path).
// SPLIT: the output is written as the split mode's fp16 pair ([row][hi(C) | lo(C)], C = heads * D = ldo): the to_out
// projection that follows is then a three-term product on the 16-bit matrix cores like the normalised ones.
template <int DB, bool SPLIT>  // head width padded to DB blocks of 32
__global__ __launch_bounds__(256) void k_flash_f32(const float* __restrict__ q, const float* __restrict__ k,
                                                   const float* __restrict__ v, float* __restrict__ o, int Tq, int Tk,
                                                   int D, int ldq, int ldk, int ldv, int ldo, int64_t q_bs, int64_t k_bs,
                                                   int64_t v_bs, int64_t o_bs, float qmul, int* overflow,
                                                   const float* __restrict__ obias) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int DP = 32 * DB, LDK = DP + 1, KT = 32;
  __shared__ float Ks[KT * LDK];
  __shared__ float Vs[KT * DP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const float* qb = q + (int64_t)b * q_bs + h * D;
  const float* kb = k + (int64_t)b * k_bs + h * D;
  const float* vb = v + (int64_t)b * v_bs + h * D;
  // zero the padding columns once (staging only writes d < D)
  for (int i = tid; i < KT * LDK; i += 256) Ks[i] = 0.f;
  for (int i = tid; i < KT * DP; i += 256) Vs[i] = 0.f;
  // Q^T fragments: lane holds Q[q0 + col][2 s + half], s < D / 2
  float qf[DP / 2];
  {
    const int qi = q0 + col;
    const float* qr = qb + (int64_t)(qi < Tq ? qi : Tq - 1) * ldq;
#pragma unroll
    for (int s = 0; s < DP / 2; ++s) qf[s] = (2 * s + half < D) ? qr[2 * s + half] * qmul : 0.f;
  }
  f32x16 oacc[DB];
#pragma unroll
  for (int j = 0; j < DB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[j][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const int d4 = D >> 2;  // float4 chunks per row
  __syncthreads();
  for (int k0 = 0; k0 < Tk; k0 += KT) {
    // ---- stage 32 key rows of K and V (zero rows beyond Tk)
    for (int i = tid; i < KT * d4; i += 256) {
      const int row = i / d4, c4 = i - row * d4;
      f4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (k0 + row < Tk) {
        kv = *(const f4*)(kb + (int64_t)(k0 + row) * ldk + c4 * 4);
        vv = *(const f4*)(vb + (int64_t)(k0 + row) * ldv + c4 * 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) Ks[row * LDK + c4 * 4 + e] = kv[e];
      *(f4*)(Vs + row * DP + c4 * 4) = vv;
    }
    __syncthreads();
    // ---- S^T = K Q^T : rows = keys, columns = queries
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < DP / 2; ++st) {
      if (2 * st < D) s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[col * LDK + 2 * st + half], qf[st], s, 0, 0, 0);
    }
    // accumulator register r of this lane: key k0 + 8 (r / 4) + 4 half + r % 4, query q0 + col
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + 8 * (r >> 2) + 4 * half + (r & 3);
      if (key >= Tk) s[r] = -INFINITY;
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    const float alpha = (m == -INFINITY) ? 0.f : exp2f(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = (s[r] == -INFINITY) ? 0.f : exp2f(s[r] - mn);
      ps += s[r];
    }
    l = l * alpha + ps;
    m = mn;
#pragma unroll
    for (int j = 0; j < DB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[j][r] *= alpha;
    // ---- O^T += V^T P^T : rows = head dims, columns = queries, contraction over the 32 keys in accumulator order
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int key = 8 * (st >> 2) + 4 * half + (st & 3);
#pragma unroll
      for (int j = 0; j < DB; ++j)
        oacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * DP + 32 * j + col], s[st], oacc[j], 0, 0, 0);
    }
    __syncthreads();
  }
  const float lt = l + __shfl_xor(l, 32);
  const float inv = 1.0f / lt;
  const int qi = q0 + col;
  if (qi < Tq) {
    float* orow = o + (int64_t)b * o_bs + (int64_t)qi * ldo + h * D;
    _Float16* hrow = (_Float16*)o + 2 * ((int64_t)b * o_bs + (int64_t)qi * ldo) + h * D;  // split form: row of 2 * ldo halves
#pragma unroll
    for (int j = 0; j < DB; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * j + 8 * g + 4 * half;  // 4 consecutive head dims: accumulator registers 4 g .. 4 g + 3
        if (d < D) {
          f4 val = {oacc[j][4 * g] * inv, oacc[j][4 * g + 1] * inv, oacc[j][4 * g + 2] * inv, oacc[j][4 * g + 3] * inv};
          if (obias) {  // value-projection bias: the rows of P sum to 1 (QKVAttentionLegacy with a biased qkv conv)
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] += obias[h * D + d + e];
          }
          if (SPLIT) {
            h4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              _Float16 a_, b_;
              split_pair(val[e] * kX3ActScale, a_, b_, overflow);
              hi[e] = a_; lo[e] = b_;
            }
            *(h4*)(hrow + d) = hi;
            *(h4*)(hrow + ldo + d) = lo;
          } else {
            *(f4*)(orow + d) = val;
          }
        }
      }
  }
#endif
}

// ---------------------------------------------------------------------------------------- LayerNorm
// One wave per row, the row in registers: mean, then the variance of the centred values (two passes over registers, as
// torch's fp32 LayerNorm does in effect). C <= 2048.
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_layernorm_f32(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                       int64_t rows, int C, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int* overflow) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  const int nv = C >> 2;
  f4 v[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c4 = lane + 64 * i;
    v[i] = (f4){0.f, 0.f, 0.f, 0.f};
    if (c4 < nv) {
      v[i] = *(const f4*)(xr + c4 * 4);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  }
  for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)C;
  float qv = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[i][e] -= mean; qv += v[i][e] * v[i][e]; }
    }
  }
  for (int off = 32; off; off >>= 1) qv += __shfl_xor(qv, off);
  const float rstd = 1.0f / sqrtf(qv / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
      const f4 g = *(const f4*)(gamma + c4 * 4), bt = *(const f4*)(beta + c4 * 4);
      f4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = v[i][e] * rstd * g[e] + bt[e];
      if (SPLIT) {
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 a_, b_;
          split_pair(t[e] * kX3ActScale, a_, b_, overflow);
          hi[e] = a_; lo[e] = b_;
        }
        _Float16* yr = (_Float16*)y + row * (2 * (int64_t)C);
        *(h4*)(yr + c4 * 4) = hi;
        *(h4*)(yr + C + c4 * 4) = lo;
      } else {
        *(f4*)(y + row * C + c4 * 4) = t;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------- GEGLU
// h [rows][2 * Nout]: the GEGLU projection (attention.py:37-44) in the packed column order of its weights - blocks of 64 =
// [32 value | 32 gate] columns (k_pack_rows). y [rows][Nout] = value * gelu(gate), exact erf form (F.gelu default).
template <bool SPLIT>
__global__ void k_geglu_f32(const float* __restrict__ h, float* __restrict__ y, int64_t rows, int Nout, int* overflow) {
  const int n4 = Nout >> 2;
  const int64_t total = rows * n4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % n4) * 4;
    const int64_t r = i / n4;
    const int src = (c >> 5) * 64 + (c & 31);
    const f4 val = *(const f4*)(h + r * (2 * (int64_t)Nout) + src);
    const f4 gate = *(const f4*)(h + r * (2 * (int64_t)Nout) + src + 32);
    f4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = val[e] * (0.5f * gate[e] * (1.0f + erff(gate[e] * 0.70710678118654752440f)));
    if (SPLIT) {
      h4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 a_, b_;
        split_pair(out[e] * kX3ActScale, a_, b_, overflow);
        hi[e] = a_; lo[e] = b_;
      }
      _Float16* yr = (_Float16*)y + r * (2 * (int64_t)Nout);
      *(h4*)(yr + c) = hi;
      *(h4*)(yr + Nout + c) = lo;
    } else {
      *(f4*)(y + r * Nout + c) = out;
    }
  }
}

// fp32 rows (optionally the channel concat of two tensors) -> the split mode's fp16 pairs [row][hi(C) | lo(C)], C = C0 + C1:
// the residual stream ahead of proj_out, the inputs of the 1 x 1 skip projections (th.cat([h, hs.pop()]), openaimodel.py:736)
__global__ void k_split_rows_f32(const float* __restrict__ x0, int ld0, int C0, const float* __restrict__ x1, int ld1, int C1,
                                 _Float16* __restrict__ y, int64_t rows, int* overflow) {
  const int C = C0 + C1, n4 = C >> 2;
  const int64_t total = rows * n4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % n4) * 4;
    const int64_t r = i / n4;
    const f4 v = c < C0 ? *(const f4*)(x0 + r * ld0 + c) : *(const f4*)(x1 + r * ld1 + (c - C0));
    h4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 a_, b_;
      split_pair(v[e] * kX3ActScale, a_, b_, overflow);
      hi[e] = a_; lo[e] = b_;
    }
    *(h4*)(y + r * (2 * (int64_t)C) + c) = hi;
    *(h4*)(y + r * (2 * (int64_t)C) + C + c) = lo;
  }
}

}  // namespace st_f32_detail
using namespace st_f32_detail;

void launch_flash_f32(hipStream_t st, const float* q, int ldq, int64_t q_bs, const float* k, int ldk, int64_t k_bs,
                      const float* v, int ldv, int64_t v_bs, float* o, int ldo, int64_t o_bs, int B, int H, int Tq, int Tk,
                      int D, float qmul, int split, int* overflow, const float* obias) {
  CD_CHECK(D % 4 == 0 && D <= 160 && (ldq % 4) == 0 && (ldk % 4) == 0 && (ldv % 4) == 0 && (ldo % 4) == 0 && Tq > 0 && Tk > 0,
           "flash_f32: D=%d ld %d %d %d %d", D, ldq, ldk, ldv, ldo);
  const dim3 grid((Tq + 127) / 128, H, B), block(256);
  const int DB = (D + 31) / 32;
  CD_CHECK(!split || ldo == H * D, "flash_f32: the split output is a dense [rows][2 * heads * D] tensor");
#define CD_FLASH(N)                                                                                                        \
  do {                                                                                                                     \
    if (split) hipLaunchKernelGGL((k_flash_f32<N, true>), grid, block, 0, st, q, k, v, o, Tq, Tk, D, ldq, ldk, ldv, ldo,  \
                                  q_bs, k_bs, v_bs, o_bs, qmul, overflow, obias);                                         \
    else hipLaunchKernelGGL((k_flash_f32<N, false>), grid, block, 0, st, q, k, v, o, Tq, Tk, D, ldq, ldk, ldv, ldo, q_bs, \
                            k_bs, v_bs, o_bs, qmul, overflow, obias);                                                     \
  } while (0)
  switch (DB) {
    case 1: CD_FLASH(1); break;
    case 2: CD_FLASH(2); break;
    case 3: CD_FLASH(3); break;
    case 4: CD_FLASH(4); break;
    default: CD_FLASH(5); break;
  }
#undef CD_FLASH
}

void launch_layernorm_f32(hipStream_t st, const float* x, int ldx, float* y, int64_t rows, int C, const float* gamma,
                          const float* beta, float eps, int split, int* overflow) {
  CD_CHECK(C % 4 == 0 && C <= 2048 && (ldx % 4) == 0, "layernorm_f32: C=%d ld=%d", C, ldx);
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (split)
    hipLaunchKernelGGL((k_layernorm_f32<true>), dim3(grid), dim3(256), 0, st, x, ldx, y, rows, C, gamma, beta, eps, overflow);
  else
    hipLaunchKernelGGL((k_layernorm_f32<false>), dim3(grid), dim3(256), 0, st, x, ldx, y, rows, C, gamma, beta, eps, overflow);
}

void launch_geglu_f32(hipStream_t st, const float* h, float* y, int64_t rows, int Nout, int split, int* overflow) {
  CD_CHECK(Nout % 32 == 0, "geglu_f32: Nout=%d", Nout);
  const int64_t n = rows * (Nout / 4);
  int64_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  if (split) hipLaunchKernelGGL(k_geglu_f32<true>, dim3((unsigned)g), dim3(256), 0, st, h, y, rows, Nout, overflow);
  else hipLaunchKernelGGL(k_geglu_f32<false>, dim3((unsigned)g), dim3(256), 0, st, h, y, rows, Nout, overflow);
}

void launch_split_rows_f32(hipStream_t st, const float* x0, int ld0, int C0, const float* x1, int ld1, int C1, bf16_t* y,
                           int64_t rows, int* overflow) {
  CD_CHECK(C0 % 4 == 0 && C1 % 4 == 0 && (ld0 % 4) == 0 && (ld1 % 4) == 0, "split_rows_f32: C=%d+%d", C0, C1);
  const int64_t n = rows * ((C0 + C1) / 4);
  int64_t g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_split_rows_f32, dim3((unsigned)g), dim3(256), 0, st, x0, ld0, C0, x1, ld1, C1, (_Float16*)y, rows,
                     overflow);
}

}  // namespace cd
