// GroupNorm(32) (+SiLU, +FiLM), LayerNorm and row softmax for NHWC bf16 activations.
//
// Reference semantics:
//   GroupNorm32 / Normalize: fp32 statistics, biased variance, eps 1e-5 (ResBlock,
//     diffusionmodules/util.py:205-216) or 1e-6 (SpatialTransformer / VAE / Ho-DDPM,
//     attention.py Normalize, model.py:38-39, ddpm/diffusion.py:32-33)
//   SiLU written as x*sigmoid(x) (model.py:33-35); FiLM h = norm(h)*(1+scale)+shift
//     (improved_ddpm/unet.py:253-257)
//   nn.LayerNorm (attention.py:205-207), softmax over the last dim (model.py:193)
// All three are HBM-bound: 16-byte vector accesses, fp32 math, deterministic reductions (no
// atomics, so encode- and decode-time passes see bit-identical statistics for identical inputs).
#include "common.h"
#include "kernels.h"

namespace cd {

namespace {

constexpr int GN_MAX_VPT = 4;  // supports C up to 8*256*4 = 8192

__device__ inline const bf16_t* gn_src(const GroupNormParams& p, int b, int row, int c, int& off) {
  if (c < p.C0) { off = c; return p.x + ((int64_t)b * p.HW + row) * p.ld0; }
  off = c - p.C0;
  return p.x1 + ((int64_t)b * p.HW + row) * p.ld1;
}

// statistics, affine and FiLM folded into one per-(image, channel) multiply-add: y = x*al + be;
// coef[b][0][c] = al, coef[b][1][c] = be
__device__ inline void gn_write_coef(const GroupNormParams& p, int b, int c, float sum, float sumsq, int cpg) {
  const int C = p.C0 + p.C1;
  const float n = (float)cpg * (float)p.HW;
  const float mean = sum / n;
  float var = sumsq / n - mean * mean;
  var = var < 0.f ? 0.f : var;
  const float rstd = 1.0f / sqrtf(var + p.eps);
  float a = rstd * p.gamma[c];
  float bb = p.beta[c] - mean * a;
  if (p.film) {
    const float* fl = p.film + (int64_t)b * p.film_ld;
    const float sc = 1.0f + fl[c];
    a *= sc;
    bb = bb * sc + fl[C + c];
  }
  float* o = p.coef + (int64_t)b * 2 * C;
  o[c] = a; o[C + c] = bb;
}

// grid (S, B). Each block reduces a slab of rows for all groups; thread owns a fixed set of 8-channel
// vectors so per-channel sums stay in registers; fixed-order LDS reduction -> partial[b][s][g][2].
__global__ __launch_bounds__(256) void k_gn_stats(GroupNormParams p, int rows_per_slab) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int C = p.C0 + p.C1;
  const int nvec = C >> 3;
  const int tid = threadIdx.x;
  const int s = blockIdx.x, b = blockIdx.y;
  const int row_begin = s * rows_per_slab;
  const int row_end = min(p.HW, row_begin + rows_per_slab);
  int rif, vpt;
  if (nvec <= 256) { rif = 256 / nvec; vpt = 1; }
  else { rif = 1; vpt = (nvec + 255) >> 8; }
  const int r0 = (nvec <= 256) ? tid / nvec : 0;
  const int v0 = (nvec <= 256) ? tid % nvec : tid;
  const bool active = (nvec <= 256) ? (tid < nvec * rif) : true;

  float sum[GN_MAX_VPT][8], sq[GN_MAX_VPT][8];
#pragma unroll
  for (int j = 0; j < GN_MAX_VPT; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) { sum[j][e] = 0.f; sq[j][e] = 0.f; }

  if (active) {
    for (int row = row_begin + r0; row < row_end; row += rif) {
#pragma unroll
      for (int j = 0; j < GN_MAX_VPT; ++j) {
        if (j < vpt) {
          const int v = v0 + j * 256;
          if (v < nvec) {
            int off;
            const bf16_t* base = gn_src(p, b, row, v * 8, off);
            float f[8];
            unpack8(*(const uint4*)(base + off), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { sum[j][e] += f[e]; sq[j][e] += f[e] * f[e]; }
          }
        }
      }
    }
  }
  // LDS layout: sum[rif][C] then sq[rif][C]
  float* lsum = sh;
  float* lsq = sh + rif * C;
  if (active) {
#pragma unroll
    for (int j = 0; j < GN_MAX_VPT; ++j) {
      if (j < vpt) {
        const int v = v0 + j * 256;
        if (v < nvec) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            lsum[r0 * C + v * 8 + e] = sum[j][e];
            lsq[r0 * C + v * 8 + e] = sq[j][e];
          }
        }
      }
    }
  }
  __syncthreads();
  if (tid < p.G) {
    const int cpg = C / p.G;
    float a = 0.f, q = 0.f;
    for (int r = 0; r < rif; ++r)
      for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += lsum[r * C + c]; q += lsq[r * C + c]; }
    float* o = p.partial + (((int64_t)b * p.S + s) * p.G + tid) * 2;
    o[0] = a; o[1] = q;
  }
}

// grid (G, B): fold the per-32-row-block channel sums written by the producing conv's epilogue
// (ConvGemmParams::stats) into one (sum, sumsq) per image and group -> partial[b][0][g][2] (S = 1).
__global__ __launch_bounds__(256) void k_gn_fold(GroupNormParams p) {
  __shared__ float rs[256], rq[256];
  const int C = p.C0 + p.C1;
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / p.G;
  const int nb = p.HW >> 5;
  const int items = nb * cpg;
  float a = 0.f, q = 0.f;
  for (int it = tid; it < items; it += 256) {
    const int rb = it / cpg, c = g * cpg + (it - rb * cpg);
    const int64_t blk = (int64_t)b * nb + rb;
    if (c < p.C0) {
      const float* s = p.pre0 + blk * 2 * p.C0 + c;
      a += s[0]; q += s[p.C0];
    } else {
      const float* s = p.pre1 + blk * 2 * p.C1 + (c - p.C0);
      a += s[0]; q += s[p.C1];
    }
  }
  rs[tid] = a; rq[tid] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {  // fixed-order tree: deterministic
    if (tid < o) { rs[tid] += rs[tid + o]; rq[tid] += rq[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    float* o = p.partial + ((int64_t)b * p.G + g) * 2;
    o[0] = rs[0]; o[1] = rq[0];
  }
  if (tid < cpg) gn_write_coef(p, b, g * cpg + tid, rs[0], rq[0], cpg);
}

// grid (B): statistics partial[b][S][G][2] -> per-channel multiply-add coefficients (non-fused path)
__global__ __launch_bounds__(256) void k_gn_coef(GroupNormParams p) {
  __shared__ float s_a[64], s_q[64];
  const int C = p.C0 + p.C1;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int cpg = C / p.G;
  if (tid < p.G) {
    float a = 0.f, q = 0.f;
    for (int s = 0; s < p.S; ++s) {  // fixed order: deterministic
      const float* o = p.partial + (((int64_t)b * p.S + s) * p.G + tid) * 2;
      a += o[0]; q += o[1];
    }
    s_a[tid] = a; s_q[tid] = q;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) gn_write_coef(p, b, c, s_a[c / cpg], s_q[c / cpg], cpg);
}

// grid (row_chunks, B): y = x*al[c] + be[c] [SiLU]. A thread owns fixed 8-channel vectors (its 16
// coefficients stay in registers) and walks rows `rif` apart, four independent 16-byte loads in flight;
// blocks are small (a few KB) so every CU holds several and the pass runs at HBM speed.
__global__ __launch_bounds__(256) void k_gn_apply(GroupNormParams p, int rows_per_block) {
  const int C = p.C0 + p.C1;
  const int nvec = C >> 3;
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int row_begin = blockIdx.x * rows_per_block;
  const int row_end = min(p.HW, row_begin + rows_per_block);
  int rif, vpt;
  if (nvec <= 256) { rif = 256 / nvec; vpt = 1; }
  else { rif = 1; vpt = (nvec + 255) >> 8; }
  const int r0 = (nvec <= 256) ? tid / nvec : 0;
  const int v0 = (nvec <= 256) ? tid % nvec : tid;
  if (nvec <= 256 && tid >= nvec * rif) return;
  const float* coef = p.coef + (int64_t)b * 2 * C;
  for (int j = 0; j < vpt; ++j) {
    const int v = v0 + j * 256;
    if (v >= nvec) break;
    const int c = v * 8;
    float ca[8], cb[8];
    {
      const f32x4 a0 = *(const f32x4*)(coef + c), a1 = *(const f32x4*)(coef + c + 4);
      const f32x4 b0 = *(const f32x4*)(coef + C + c), b1 = *(const f32x4*)(coef + C + c + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { ca[e] = a0[e]; ca[4 + e] = a1[e]; cb[e] = b0[e]; cb[4 + e] = b1[e]; }
    }
    for (int row = row_begin + r0; row < row_end; row += 4 * rif) {
      uint4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = row + u * rif;
        raw[u] = make_uint4(0, 0, 0, 0);
        if (r < row_end) {
          int off;
          const bf16_t* base = gn_src(p, b, r, c, off);
          raw[u] = *(const uint4*)(base + off);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = row + u * rif;
        if (r < row_end) {
          float f[8];
          unpack8(raw[u], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float y = f[e] * ca[e] + cb[e];
            if (p.silu) y = silu_f(y);
            f[e] = y;
          }
          *(uint4*)(p.y + ((int64_t)b * p.HW + r) * C + c) = pack8(f);
        }
      }
    }
  }
}

// One wave per row, R rows per wave; C multiple of 8, C <= 64 * 8 * VPL. All 16-byte loads of the wave's R rows are
// issued before any arithmetic: a single 640-byte row per wave (C = 320) keeps only ~5 MB in flight over the whole
// chip, a third of what HBM latency x bandwidth needs; R = 4 puts the pass at the speed of the other streaming
// kernels. Row sums by DPP / permlane swaps (wave_allsum), no LDS traffic.
template <int VPL, int R>
__global__ __launch_bounds__(256) void k_layernorm(const bf16_t* __restrict__ x, int ldx,
                                                   bf16_t* __restrict__ y, int ldy, int rows, int C,
                                                   const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float eps) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  const int nvec = C >> 3;
  uint4 raw[R][VPL];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      // unconditional loads (clamped to a valid row / vector; the duplicates are never used): predicated ones
      // end up behind branches with a vmcnt(0) between them
      const int v = lane + j * 64 < nvec ? lane + j * 64 : 0;
      const int row = row0 + r < rows ? row0 + r : rows - 1;
      raw[r][j] = *(const uint4*)(x + (int64_t)row * ldx + v * 8);
    }
  float ga[VPL][8], be[VPL][8];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int v = lane + j * 64 < nvec ? lane + j * 64 : 0;
    {
      const f32x4 g0 = *(const f32x4*)(gamma + v * 8), g1 = *(const f32x4*)(gamma + v * 8 + 4);
      const f32x4 b0 = *(const f32x4*)(beta + v * 8), b1 = *(const f32x4*)(beta + v * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { ga[j][e] = g0[e]; ga[j][4 + e] = g1[e]; be[j][e] = b0[e]; be[j][4 + e] = b1[e]; }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = row0 + r;
    if (row >= rows) break;  // wave-uniform
    float f[VPL][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      unpack8(raw[r][j], f[j]);
      if (lane + j * 64 < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[j][e];
      }
    }
    s = wave_allsum(s);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      if (lane + j * 64 < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[j][e] - mean; q += d * d; }
      }
    }
    q = wave_allsum(q);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int v = lane + j * 64;
      if (v < nvec) {
        float o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = (f[j][e] - mean) * rstd * ga[j][e] + be[j][e];
        *(uint4*)(y + (int64_t)row * ldy + v * 8) = pack8(o8);
      }
    }
  }
}

// one block per row; three passes over an L2-resident fp32 row
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ s, int lds_,
                                                      bf16_t* __restrict__ p, int ldp, int cols) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const float* sr = s + row * lds_;
  bf16_t* pr = p + row * ldp;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float mx = -INFINITY;
  for (int i = tid; i < cols; i += 256) mx = fmaxf(mx, sr[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int i = tid; i < cols; i += 256) sum += __expf(sr[i] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) red[w] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  const float inv = 1.0f / sum;
  for (int i = tid; i < cols; i += 256) pr[i] = f2bf(__expf(sr[i] - mx) * inv);
}

}  // namespace

int groupnorm_slabs(int B, int HW, int C) {
  // The slab partition fixes the summation order of the statistics, so it must NOT depend on the
  // batch size: the B-sized encode pass and the 2B-sized CFG decode pass have to produce bit-identical
  // statistics for identical samples (error amplification in eps extraction).
  (void)B; (void)C;
  int S = HW / 64;
  if (S < 1) S = 1;
  if (S > 64) S = 64;
  return S;
}

void launch_groupnorm(hipStream_t st, const GroupNormParams& p) {
  const int C = p.C0 + p.C1;
  CD_CHECK(C % 8 == 0 && C % p.G == 0 && p.G <= 64, "groupnorm: C=%d G=%d unsupported", C, p.G);
  CD_CHECK(p.C0 % 8 == 0, "groupnorm: concat split must be a multiple of 8");
  CD_CHECK((C >> 3) <= 256 * GN_MAX_VPT, "groupnorm: C too large");
  CD_CHECK(p.partial && p.S > 0, "groupnorm: workspace missing");
  const int nvec = C >> 3;
  const int rif = nvec <= 256 ? 256 / nvec : 1;
  CD_CHECK(p.coef, "groupnorm: coefficient workspace missing");
  CD_CHECK(C / p.G <= 256, "groupnorm: more than 256 channels per group");
  GroupNormParams q = p;
  if (p.pre0 && (p.C1 == 0 || p.pre1) && (p.HW % 32) == 0) {
    q.S = 1;  // statistics came out of the producing convs' epilogues: fold them per (image, group)
    hipLaunchKernelGGL(k_gn_fold, dim3(p.G, p.B), dim3(256), 0, st, q);
  } else {
    const int rows_per_slab = ceil_div(p.HW, p.S);
    const size_t lds = (size_t)rif * C * 2 * sizeof(float);
    hipLaunchKernelGGL(k_gn_stats, dim3(p.S, p.B), dim3(256), lds, st, p, rows_per_slab);
    hipLaunchKernelGGL(k_gn_coef, dim3(p.B), dim3(256), 0, st, p);
  }
  // ~2048 blocks per launch where the tensor allows it; at least one unrolled batch of rows per thread
  int rows_per_block = (int)(((int64_t)p.HW * p.B + 2047) / 2048);
  const int unit = 4 * rif;
  rows_per_block = ceil_div(rows_per_block, unit) * unit;
  hipLaunchKernelGGL(k_gn_apply, dim3(ceil_div(p.HW, rows_per_block), p.B), dim3(256), 0, st, q, rows_per_block);
}

void launch_layernorm(hipStream_t st, const bf16_t* x, int ldx, bf16_t* y, int ldy, int rows,
                      int C, const float* gamma, const float* beta, float eps) {
  CD_CHECK(C % 8 == 0 && C <= 2048, "layernorm: C=%d unsupported", C);
  CD_CHECK((ldx & 7) == 0 && (ldy & 7) == 0, "layernorm: row strides must be multiples of 8 elements");
#define CD_LN(VPL, R)                                                                                              \
  hipLaunchKernelGGL((k_layernorm<VPL, R>), dim3(ceil_div(rows, 4 * R)), dim3(256), 0, st, x, ldx, y, ldy, rows, C, \
                     gamma, beta, eps)
  if (C <= 512) CD_LN(1, 4);
  else if (C <= 1024) CD_LN(2, 4);
  else if (C <= 1536) CD_LN(3, 2);
  else CD_LN(4, 2);
#undef CD_LN
}

void launch_softmax_rows(hipStream_t st, const float* s, int lds_, bf16_t* p, int ldp, int64_t rows,
                         int cols) {
  hipLaunchKernelGGL(k_softmax_rows, dim3((unsigned)rows), dim3(256), 0, st, s, lds_, p, ldp, cols);
}

}  // namespace cd
