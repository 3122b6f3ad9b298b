// Small HBM-bound helpers: layout changes at the fp32-NCHW boundary, timestep embeddings, the
// time-embedding MLP on [B][dim] vectors, pooling / upsampling, VAE posterior sampling.
#include "common.h"
#include <mutex>

#include "kernels.h"

namespace cd {

namespace {

__device__ inline void philox_round2(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ inline float philox_normal2(uint64_t seed, uint32_t stream, uint64_t idx) {
  uint32_t c[4] = {(uint32_t)(idx >> 1), (uint32_t)(idx >> 33), stream, 0x9E3779B9u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) { philox_round2(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  float rad = sqrtf(-2.0f * __logf(u1));
  float ang = 6.28318530717958647692f * u2;
  return (idx & 1) ? rad * __sinf(ang) : rad * __cosf(ang);
}

__global__ void k_nchw_to_nhwc(const float* __restrict__ x, bf16_t* __restrict__ y, int B, int C,
                               int HW, int Cpad, float scale, float shift, int dup) {
  const int64_t n = (int64_t)B * HW * Cpad;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const int64_t bp = i / Cpad;
    const int b = (int)(bp / HW);
    const int pix = (int)(bp - (int64_t)b * HW);
    float v = 0.f;
    if (c < C) v = x[((int64_t)b * C + c) * HW + pix] * scale + shift;
    const bf16_t h = f2bf(v);
    y[i] = h;
    if (dup) y[i + n] = h;
  }
}

__global__ void k_nhwc_to_nchw(const void* __restrict__ x, int x_f32, int ldx, float* __restrict__ y,
                               int B, int C, int HW, float scale, float shift) {
  const int64_t n = (int64_t)B * C * HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int pix = (int)(i % HW);
    const int64_t bc = i / HW;
    const int c = (int)(bc % C);
    const int b = (int)(bc / C);
    const int64_t src = ((int64_t)b * HW + pix) * ldx + c;
    const float v = x_f32 ? ((const float*)x)[src] : bf2f(((const bf16_t*)x)[src]);
    y[i] = v * scale + shift;
  }
}

__global__ void k_timestep_embedding(const StepCoef* tab, const int* step_ptr, int step,
                                     const float* t_explicit, float* out, int B, int dim, int mode) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i % half;
  float t;
  if (t_explicit) t = t_explicit[b];
  else t = (float)tab[step_ptr ? *step_ptr : step].t;
  float* o = out + (int64_t)b * dim;
  if (mode == 0) {
    // util.py:162-167: freqs = exp(-log(1e4) * arange(half)/half); cat([cos, sin])
    const float f = expf(-9.210340371976184f * (float)k / (float)half);
    const float a = t * f;
    o[k] = cosf(a);
    o[half + k] = sinf(a);
  } else {
    // ddpm/diffusion.py:16-21: emb = log(1e4)/(half-1); exp(arange(half) * -emb); cat([sin, cos])
    const float e = 9.210340371976184f / (float)(half - 1);
    const float f = expf((float)k * -e);
    const float a = t * f;
    o[k] = sinf(a);
    o[half + k] = cosf(a);
  }
  if ((dim & 1) && k == 0) o[dim - 1] = 0.f;
}

// y[b][n] = act_out(sum_k act_in(x[b][k]) * W[n][k] + bias[n]);   one wave per output element
__global__ __launch_bounds__(256) void k_vec_linear(const float* __restrict__ x, int ldx,
                                                    const float* __restrict__ W,
                                                    const float* __restrict__ bias, float* __restrict__ y,
                                                    int ldy, int B, int K, int N, int silu_in,
                                                    int silu_out) {
  const int lane = threadIdx.x & 63;
  const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= (int64_t)B * N) return;
  const int b = (int)(o / N), n = (int)(o % N);
  const float* xr = x + (int64_t)b * ldx;
  const float* wr = W + (int64_t)n * K;
  float acc = 0.f;
  if ((K & 3) == 0 && (ldx & 3) == 0) {
    // 16 B per lane and four independent row segments in flight: the 100 MB fused time-embedding
    // projection is a pure weight stream, one load at a time per wave leaves it latency-bound
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
    const int nv = K >> 2;
    for (int v0 = lane; v0 < nv; v0 += 256) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int v = v0 + u * 64;
        if (v < nv) {
          const f32x4 wv = __builtin_nontemporal_load((const f32x4*)wr + v);
          f32x4 xv = *((const f32x4*)xr + v);
          if (silu_in) { xv[0] = silu_acc(xv[0]); xv[1] = silu_acc(xv[1]); xv[2] = silu_acc(xv[2]); xv[3] = silu_acc(xv[3]); }
          a4[u] += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
        }
      }
    }
    acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  } else {
    for (int k = lane; k < K; k += 64) {
      float xv = xr[k];
      if (silu_in) xv = silu_acc(xv);
      acc += xv * wr[k];
    }
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
  if (lane == 0) {
    float v = acc + (bias ? bias[n] : 0.f);
    if (silu_out) v = silu_acc(v);
    y[(int64_t)b * ldy + n] = v;
  }
}

// CLIPTextEmbeddings.forward (transformers models/clip/modeling_clip.py): token_embedding(ids) + position_embedding
__global__ void k_embed_tokens(const int* __restrict__ ids, const float* __restrict__ tok,
                               const float* __restrict__ pos, bf16_t* __restrict__ out, int B, int L, int D, int vocab) {
  const int nvec = D / 8;
  const int64_t n = (int64_t)B * L * nvec;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % nvec);
    const int64_t bl = i / nvec;
    const int l = (int)(bl % L);
    int id = ids[bl];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float* tr = tok + (int64_t)id * D + v * 8;
    const float* pr = pos + (int64_t)l * D + v * 8;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = tr[e] + pr[e];
    *(uint4*)(out + bl * D + v * 8) = pack8(f);
  }
}

__global__ void k_vit_tokens(const bf16_t* __restrict__ patch, const float* __restrict__ cls,
                             const float* __restrict__ pos, bf16_t* __restrict__ out, int B, int T, int D) {
  const int nvec = D / 8;
  const int64_t n = (int64_t)B * (T + 1) * nvec;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % nvec);
    const int64_t bt = i / nvec;
    const int t = (int)(bt % (T + 1));
    const int b = (int)(bt / (T + 1));
    float f[8];
    if (t == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = cls[v * 8 + e];
    } else {
      unpack8(*(const uint4*)(patch + ((int64_t)b * T + (t - 1)) * D + v * 8), f);
    }
    const float* pr = pos + (int64_t)t * D + v * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += pr[e];
    *(uint4*)(out + bt * D + v * 8) = pack8(f);
  }
}

// x[b, argmax(ids[b])] (model.py encode_text: x[torch.arange(B), text.argmax(dim=-1)]; first maximum on ties)
__global__ void k_gather_eot(const bf16_t* __restrict__ x, const int* __restrict__ ids, bf16_t* __restrict__ out,
                             int L, int D) {
  const int b = blockIdx.x;
  int best = 0, bv = ids[(int64_t)b * L];
  for (int l = 1; l < L; ++l) {
    const int v = ids[(int64_t)b * L + l];
    if (v > bv) { bv = v; best = l; }
  }
  const bf16_t* src = x + ((int64_t)b * L + best) * D;
  for (int i = threadIdx.x; i < D; i += blockDim.x) out[(int64_t)b * D + i] = src[i];
}

__global__ void k_avgpool2(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, nvec = C / 8;
  const int64_t n = (int64_t)B * Ho * Wo * nvec;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % nvec);
    int64_t t = i / nvec;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float f[8];
        unpack8(*(const uint4*)(x + (((int64_t)b * H + oy * 2 + dy) * W + ox * 2 + dx) * C + v * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += f[e];
      }
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] *= 0.25f;
    *(uint4*)(y + (((int64_t)b * Ho + oy) * Wo + ox) * C + v * 8) = pack8(a);
  }
}

__global__ void k_upsample2(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int C) {
  const int Ho = H * 2, Wo = W * 2, nvec = C / 8;
  const int64_t n = (int64_t)B * Ho * Wo * nvec;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % nvec);
    int64_t t = i / nvec;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    *(uint4*)(y + (((int64_t)b * Ho + oy) * Wo + ox) * C + v * 8) =
        *(const uint4*)(x + (((int64_t)b * H + (oy >> 1)) * W + (ox >> 1)) * C + v * 8);
  }
}

__global__ void k_posterior_sample(const float* __restrict__ mom, int ld, const float* __restrict__ noise,
                                   uint64_t seed, float* __restrict__ z, int B, int zc, int HW,
                                   float scale, int use_mean) {
  const int64_t n = (int64_t)B * zc * HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int pix = (int)(i % HW);
    const int64_t bc = i / HW;
    const int c = (int)(bc % zc);
    const int b = (int)(bc / zc);
    const float* m = mom + ((int64_t)b * HW + pix) * ld;
    const float mean = m[c];
    float v = mean;
    if (!use_mean) {
      float lv = m[zc + c];
      lv = fminf(fmaxf(lv, -30.0f), 20.0f);       // distributions.py:30
      const float sd = expf(0.5f * lv);           // distributions.py:33
      const float nz = noise ? noise[i] : philox_normal2(seed, 0x7a65u, (uint64_t)i);
      v = mean + sd * nz;                         // distributions.py:36
    }
    z[i] = scale * v;                             // ddpm.py:543 scale_factor * z
  }
}

// one thread per latent vector, codebook staged in LDS ([n_embed][zc] fp32: 96 KB for the 8192 x 3 codebook of VQ-f4)
__global__ __launch_bounds__(256) void k_vq_quantize(const float* __restrict__ z, float in_mul,
                                                     const float* __restrict__ codebook, int n_embed, int zc, int B,
                                                     int HW, bf16_t* __restrict__ out, int Cpad) {
  extern __shared__ float cb[];
  for (int i = threadIdx.x; i < n_embed * zc; i += blockDim.x) cb[i] = codebook[i];
  __syncthreads();
  const int64_t n = (int64_t)B * HW;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(v / HW), pix = (int)(v - (int64_t)b * HW);
    float zv[8];
    float z2 = 0.f;
    for (int c = 0; c < zc; ++c) {
      zv[c] = z[((int64_t)b * zc + c) * HW + pix] * in_mul;
      z2 += zv[c] * zv[c];
    }
    int best = 0;
    float bd = INFINITY;
    for (int k = 0; k < n_embed; ++k) {
      const float* e = cb + k * zc;
      float e2 = 0.f, dot = 0.f;
      for (int c = 0; c < zc; ++c) { e2 += e[c] * e[c]; dot += zv[c] * e[c]; }
      const float d = (z2 + e2) - 2.0f * dot;
      if (d < bd) { bd = d; best = k; }  // strict: the first minimum wins, as torch.argmin
    }
    bf16_t* o = out + v * Cpad;
    for (int c = 0; c < Cpad; ++c) o[c] = f2bf(c < zc ? zv[c] + (cb[best * zc + c] - zv[c]) : 0.f);
  }
}

__global__ void k_fill_f32(float* p, float v, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void k_copy_strided_bf16(const bf16_t* __restrict__ src, int lds_, bf16_t* __restrict__ dst,
                                    int ldd, int64_t rows, int cols) {
  const int nvec = cols / 8;
  const int64_t n = rows * nvec;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nvec;
    const int v = (int)(i % nvec);
    *(uint4*)(dst + r * ldd + v * 8) = *(const uint4*)(src + r * lds_ + v * 8);
  }
}

inline int ew_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

void launch_nchw_to_nhwc(hipStream_t st, const float* x, bf16_t* y, int B, int C, int HW, int Cpad,
                         float scale, float shift, int dup) {
  const int64_t n = (int64_t)B * HW * Cpad;
  hipLaunchKernelGGL(k_nchw_to_nhwc, dim3(ew_grid(n)), dim3(256), 0, st, x, y, B, C, HW, Cpad, scale,
                     shift, dup);
}
void launch_nhwc_to_nchw(hipStream_t st, const void* x, int x_f32, int ldx, float* y, int B, int C,
                         int HW, float scale, float shift) {
  const int64_t n = (int64_t)B * C * HW;
  hipLaunchKernelGGL(k_nhwc_to_nchw, dim3(ew_grid(n)), dim3(256), 0, st, x, x_f32, ldx, y, B, C, HW,
                     scale, shift);
}
void launch_timestep_embedding(hipStream_t st, const StepCoef* tab, const int* step_ptr, int step,
                               const float* t_explicit, float* out, int B, int dim, int mode) {
  const int n = B * (dim / 2);
  hipLaunchKernelGGL(k_timestep_embedding, dim3(ceil_div(n, 256)), dim3(256), 0, st, tab, step_ptr,
                     step, t_explicit, out, B, dim, mode);
}
void launch_vec_linear(hipStream_t st, const float* x, int ldx, const float* W, const float* bias,
                       float* y, int ldy, int B, int K, int N, int silu_in, int silu_out) {
  const int64_t outs = (int64_t)B * N;
  hipLaunchKernelGGL(k_vec_linear, dim3((unsigned)ceil_div64(outs, 4)), dim3(256), 0, st, x, ldx, W,
                     bias, y, ldy, B, K, N, silu_in, silu_out);
}
void launch_avgpool2(hipStream_t st, const bf16_t* x, bf16_t* y, int B, int H, int W, int C) {
  const int64_t n = (int64_t)B * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(k_avgpool2, dim3(ew_grid(n)), dim3(256), 0, st, x, y, B, H, W, C);
}
void launch_upsample2(hipStream_t st, const bf16_t* x, bf16_t* y, int B, int H, int W, int C) {
  const int64_t n = (int64_t)B * H * 2 * W * 2 * (C / 8);
  hipLaunchKernelGGL(k_upsample2, dim3(ew_grid(n)), dim3(256), 0, st, x, y, B, H, W, C);
}
void launch_posterior_sample(hipStream_t st, const float* mom, int ld, const float* noise,
                             uint64_t seed, float* z, int B, int zc, int HW, float scale,
                             int use_mean) {
  const int64_t n = (int64_t)B * zc * HW;
  hipLaunchKernelGGL(k_posterior_sample, dim3(ew_grid(n)), dim3(256), 0, st, mom, ld, noise, seed, z,
                     B, zc, HW, scale, use_mean);
}
void launch_vq_quantize(hipStream_t st, const float* z, float in_mul, const float* codebook, int n_embed, int zc,
                        int B, int HW, bf16_t* out, int Cpad) {
  CD_CHECK(zc >= 1 && zc <= 8 && n_embed > 0, "vq_quantize: embed_dim %d / n_embed %d", zc, n_embed);
  const size_t lds = (size_t)n_embed * zc * sizeof(float);
  CD_CHECK(lds <= 150 * 1024, "vq_quantize: codebook of %zu bytes does not fit LDS", lds);
  static PerDeviceOnce attr_once;  // engines on several host threads / devices may reach this launch together
  attr_once([&]() {
    HIP_CHECK(hipFuncSetAttribute((const void*)k_vq_quantize, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  });
  const int64_t n = (int64_t)B * HW;
  int grid = (int)((n + 255) / 256);
  if (grid > 512) grid = 512;
  hipLaunchKernelGGL(k_vq_quantize, dim3(grid), dim3(256), lds, st, z, in_mul, codebook, n_embed, zc, B, HW, out, Cpad);
}
void launch_embed_tokens(hipStream_t st, const int* ids, const float* tok, const float* pos, bf16_t* out, int B,
                         int L, int D, int vocab) {
  CD_CHECK(D % 8 == 0, "embed_tokens: width %% 8");
  const int64_t n = (int64_t)B * L * (D / 8);
  hipLaunchKernelGGL(k_embed_tokens, dim3(ew_grid(n)), dim3(256), 0, st, ids, tok, pos, out, B, L, D, vocab);
}
void launch_vit_tokens(hipStream_t st, const bf16_t* patch, const float* cls, const float* pos, bf16_t* out, int B,
                       int T, int D) {
  CD_CHECK(D % 8 == 0, "vit_tokens: width %% 8");
  const int64_t n = (int64_t)B * (T + 1) * (D / 8);
  hipLaunchKernelGGL(k_vit_tokens, dim3(ew_grid(n)), dim3(256), 0, st, patch, cls, pos, out, B, T, D);
}
void launch_gather_eot(hipStream_t st, const bf16_t* x, const int* ids, bf16_t* out, int B, int L, int D) {
  hipLaunchKernelGGL(k_gather_eot, dim3(B), dim3(256), 0, st, x, ids, out, L, D);
}
void launch_fill_f32(hipStream_t st, float* p, float v, int64_t n) {
  hipLaunchKernelGGL(k_fill_f32, dim3(ew_grid(n)), dim3(256), 0, st, p, v, n);
}
void launch_copy_strided_bf16(hipStream_t st, const bf16_t* src, int lds_, bf16_t* dst, int ldd,
                              int64_t rows, int cols) {
  const int64_t n = rows * (cols / 8);
  hipLaunchKernelGGL(k_copy_strided_bf16, dim3(ew_grid(n)), dim3(256), 0, st, src, lds_, dst, ldd,
                     rows, cols);
}

}  // namespace cd
