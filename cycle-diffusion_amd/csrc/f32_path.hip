// fp32 execution path (cd_net_desc::precision = CD_PREC_F32): the same network executors, with fp32 NHWC
// activations, fp32 packed weights and exact-fp32 matrix instructions (v_mfma_f32_32x32x2_f32).
//
// Why it exists: the pixel-space wrapper in sample_type 'ddim' (ddpm_ddim_wrapper.py:283-307, 114-227) rescales
// x by sqrt(abar_{t-1}/abar_t) every step (x130 end to end on the DDPM linear schedule) and divides by sigma_t when
// it extracts eps; the reference runs fp32 (`use_fp16=False`, improved_ddpm/script_util.py:15) and the chain is only
// reproducible if eps_hat(x_t) is a smooth function of x_t at fp32 resolution. Any 16-bit quantiser inside the
// network turns a 1e-7 perturbation of x_t into 2^-11 jumps, which the chain amplifies to O(1) image error
// (DESIGN.md §5). These networks are small (1-94 M parameters), so the 157 TFLOP/s fp32 MFMA rate is ample.
//
// One tile shape per problem size and a fixed k-ascending accumulation order: results are bit-reproducible across
// processes (no autotuner, no split-K on this path).
#include <stdio.h>

#include "common.h"
#include "kernels.h"

namespace cd {

namespace f32_detail {

typedef __attribute__((ext_vector_type(4))) float f4;

// ---------------------------------------------------------------------------------------- implicit-GEMM conv
// A = activations fp32 NHWC (optional channel concat, nearest-x2 upsample, stride, asymmetric padding folded
// into the gather), B = weights fp32 [Npad][KH*KW*Ctot]. 256 threads = 2 x 2 waves, BK = 16, LDS tiles stored
// k-major so that the 32 lanes of an MFMA operand read 32 consecutive floats.
template <int BM, int BN>
__global__ __launch_bounds__(256) void k_conv_f32(ConvGemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BK = 16, TM = BM / 2, TN = BN / 2, MT = TM / 32, NT = TN / 32;
  constexpr int LDA = BM + 2, LDB = BN + 2;   // +2: the four k-quads of a staging store hit 4 distinct bank groups
  constexpr int A_V = BM * BK / 4 / 256;      // float4 staging loads per thread
  constexpr int B_V = BN * BK / 4 / 256;
  static_assert(A_V >= 1 && B_V >= 1, "tile too small for 256 threads");
  __shared__ float As[2][BK * LDA];
  __shared__ float Bs[2][BK * LDB];
  const float* src0 = (const float*)p.src0;
  const float* src1 = (const float*)p.src1;
  const float* wgt = (const float*)p.wgt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int Ctot = p.C0 + p.C1;
  const int HWo = p.Hout * p.Wout;

  // staging geometry: vector v of this thread covers row (idx / 4), k-quad (idx % 4)
  int a_iy0[A_V], a_ix0[A_V], a_boff[A_V];
#pragma unroll
  for (int v = 0; v < A_V; ++v) {
    const int row = (tid + v * 256) >> 2;
    const int m = m0 + row;
    if (m < p.M) {
      const int b = m / HWo, rem = m - b * HWo;
      const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
      a_iy0[v] = oy * p.stride - p.pad_t;
      a_ix0[v] = ox * p.stride - p.pad_l;
      a_boff[v] = b * p.Hs * p.Ws;
    } else {
      a_iy0[v] = -(1 << 28); a_ix0[v] = 0; a_boff[v] = 0;
    }
  }
  const int nk = p.Ktot / BK;
  f4 ra[A_V], rb[B_V];
  auto gload = [&](int kt) {
    const int k_el = kt * BK;
    const int tap = k_el / Ctot;
    const int kc = k_el - tap * Ctot;           // a K step never straddles a tap or the concat seam (C % 32 == 0)
    const int kr = tap / p.KW, ks = tap - kr * p.KW;
    const bool first = kc < p.C0;
    const float* base = first ? src0 : src1;
    const int ld = first ? p.ld0 : p.ld1;
    const int cc = first ? kc : kc - p.C0;
#pragma unroll
    for (int v = 0; v < A_V; ++v) {
      const int kq = (tid + v * 256) & 3;
      int iy = a_iy0[v] + kr, ix = a_ix0[v] + ks;
      const bool ok = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);
      if (p.up) { iy >>= 1; ix >>= 1; }
      f4 val = {0.f, 0.f, 0.f, 0.f};
      if (ok) val = *(const f4*)(base + (int64_t)(a_boff[v] + iy * p.Ws + ix) * ld + cc + kq * 4);
      ra[v] = val;
    }
#pragma unroll
    for (int v = 0; v < B_V; ++v) {
      const int idx = tid + v * 256;
      int n = n0 + (idx >> 2);
      if (n >= p.N) n = p.N - 1;                // duplicate of a valid row; its outputs are masked
      rb[v] = *(const f4*)(wgt + (int64_t)n * p.Ktot + k_el + (idx & 3) * 4);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int v = 0; v < A_V; ++v) {
      const int idx = tid + v * 256, row = idx >> 2, kq = idx & 3;
#pragma unroll
      for (int e = 0; e < 4; ++e) As[buf][(kq * 4 + e) * LDA + row] = ra[v][e];
    }
#pragma unroll
    for (int v = 0; v < B_V; ++v) {
      const int idx = tid + v * 256, row = idx >> 2, kq = idx & 3;
#pragma unroll
      for (int e = 0; e < 4; ++e) Bs[buf][(kq * 4 + e) * LDB + row] = rb[v][e];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5;

  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = As[cur][(kk * 2 + fhalf) * LDA + wm * TM + i * 32 + frow];
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = Bs[cur][(kk * 2 + fhalf) * LDB + wn * TN + j * 32 + frow];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }

  // epilogue straight from the accumulators: for a fixed register r the 32 lanes of a half-wave hold 32
  // consecutive columns of one row (128-byte stores)
  float* out = (float*)p.out;
  const float* resid = (const float*)p.resid;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * TN + j * 32 + frow;
    if (n >= p.N) continue;
    const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        if (m >= p.M) continue;
        float v = acc[i][j][r] * p.alpha + bias;
        if (p.rowvec) {
          const int rvi = (p.rows_per_vec >= p.M) ? 0 : m / p.rows_per_vec;
          v += p.rowvec[(int64_t)rvi * p.rowvec_ld + n];
        }
        if (p.act == ACT_SILU) v = silu_acc(v);
        else if (p.act == ACT_GELU) v = gelu_f(v);
        if (resid) v += resid[(int64_t)m * p.resid_ld + n];
        out[(int64_t)m * p.out_ld + n] = v;
      }
  }
#endif
}

// ---------------------------------------------------------------------------------------- GroupNorm(32)
// pass 1: per (image, pixel slab) per-channel sums in fp64 (coalesced sweep, thread-fixed channel quad)
__global__ __launch_bounds__(256) void k_gn_partial_f32(const float* __restrict__ x0, const float* __restrict__ x1,
                                                        int C0, int C1, int ld0, int ld1, int HW, int S,
                                                        double* __restrict__ part) {
  const int C = C0 + C1, C4 = C >> 2;
  const int b = blockIdx.y, s = blockIdx.x;
  const int ppb = (HW + S - 1) / S;
  const int p0 = s * ppb, p1 = min(HW, p0 + ppb);
  if (C4 > 256) {  // wide concat inputs (> 1024 channels): one pixel per sweep, thread t owns channel quads t, t + 256, ...
    for (int cv = threadIdx.x; cv < C4; cv += 256) {
      const int c = cv * 4;
      const bool first = c < C0;
      const float* base = first ? x0 : x1;
      const int ld = first ? ld0 : ld1;
      const int cc = first ? c : c - C0;
      double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int p = p0; p < p1; ++p) {
        const f4 v = *(const f4*)(base + ((int64_t)b * HW + p) * ld + cc);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] += (double)v[e]; a[4 + e] += (double)v[e] * (double)v[e]; }
      }
      double* o = part + (((int64_t)b * S + s) * C4 + cv) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = a[e];
    }
    return;
  }
  const int tpp = C4;                           // threads per pixel
  const int pix_per_iter = 256 / tpp;           // whole pixels per sweep; threads beyond that idle
  const int tid = threadIdx.x;
  const int cv = tid % tpp, pr = tid / tpp;
  double sum[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
  if (pr < pix_per_iter) {
    const int c = cv * 4;
    const bool first = c < C0;
    const float* base = first ? x0 : x1;
    const int ld = first ? ld0 : ld1;
    const int cc = first ? c : c - C0;
    for (int p = p0 + pr; p < p1; p += pix_per_iter) {
      const f4 v = *(const f4*)(base + ((int64_t)b * HW + p) * ld + cc);
#pragma unroll
      for (int e = 0; e < 4; ++e) { sum[e] += (double)v[e]; sq[e] += (double)v[e] * (double)v[e]; }
    }
  }
  __shared__ double sm[256 * 8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { sm[tid * 8 + e] = sum[e]; sm[tid * 8 + 4 + e] = sq[e]; }
  __syncthreads();
  if (tid < tpp) {
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < pix_per_iter; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += sm[(r * tpp + tid) * 8 + e];
    double* o = part + (((int64_t)b * S + s) * C4 + tid) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = a[e];
  }
}
// pass 2: fold slabs and the channels of a group -> per (image, channel) multiply-add pair
__global__ void k_gn_coef_f32(const double* __restrict__ part, int C, int HW, int S, int G, float eps,
                              const float* __restrict__ gamma, const float* __restrict__ beta,
                              const float* __restrict__ film, int film_ld, float* __restrict__ coef) {
  const int b = blockIdx.y, g = blockIdx.x;
  const int cpg = C / G, C4 = C >> 2;
  __shared__ double red[2];
  double s = 0, q = 0;  // fixed partition over the 64 lanes, fixed fold order: deterministic
  for (int i = threadIdx.x; i < cpg * S; i += 64) {
    const int c = g * cpg + i % cpg, sl = i / cpg;
    const double* o = part + (((int64_t)b * S + sl) * C4 + (c >> 2)) * 8;
    s += o[c & 3]; q += o[4 + (c & 3)];
  }
  for (int off = 32; off; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
  if (threadIdx.x == 0) {
    const double n = (double)cpg * HW;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0) var = 0;
    red[0] = mean; red[1] = 1.0 / sqrt(var + (double)eps);
  }
  __syncthreads();
  const float mean = (float)red[0], rstd = (float)red[1];
  for (int c = g * cpg + threadIdx.x; c < (g + 1) * cpg; c += blockDim.x) {
    float a = rstd * gamma[c];
    float d = beta[c] - mean * a;
    if (film) {  // y = gn(x) * (1 + scale) + shift  (improved_ddpm/unet.py:253-257)
      const float* fl = film + (int64_t)b * film_ld;
      const float sc = 1.f + fl[c];
      a *= sc; d = d * sc + fl[C + c];
    }
    coef[((int64_t)b * 2) * C + c] = a;
    coef[((int64_t)b * 2 + 1) * C + c] = d;
  }
}
// CD_PREC_F32X3: the producing convolutions' epilogues already wrote per-channel sums over every 32-row block of the
// tensor (ConvGemmParams::stats, fp32): grid (G, B) folds them in fp64, in a fixed order, straight into the per
// (image, channel) multiply-add pair - no pass over the tensor itself (replaces k_gn_partial_f32 + k_gn_coef_f32)
__global__ __launch_bounds__(256) void k_gn_fold_f32(const float* __restrict__ pre0, const float* __restrict__ pre1,
                                                     int C0, int C1, int HW, int G, float eps,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ film, int film_ld,
                                                     float* __restrict__ coef) {
  __shared__ double rs[256], rq[256];
  const int C = C0 + C1;
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / G;
  const int nb = HW >> 5;
  const int items = nb * cpg;
  double a = 0, q = 0;
  for (int it = tid; it < items; it += 256) {
    const int rb = it / cpg, c = g * cpg + (it - rb * cpg);
    const int64_t blk = (int64_t)b * nb + rb;
    if (c < C0) {
      const float* sp = pre0 + blk * 2 * C0 + c;
      a += (double)sp[0]; q += (double)sp[C0];
    } else {
      const float* sp = pre1 + blk * 2 * C1 + (c - C0);
      a += (double)sp[0]; q += (double)sp[C1];
    }
  }
  rs[tid] = a; rq[tid] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {  // fixed-order tree: deterministic
    if (tid < o) { rs[tid] += rs[tid + o]; rq[tid] += rq[tid + o]; }
    __syncthreads();
  }
  const double n = (double)cpg * HW;
  const double mean_d = rs[0] / n;
  double var = rq[0] / n - mean_d * mean_d;
  if (var < 0) var = 0;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (tid < cpg) {
    const int c = g * cpg + tid;
    float ca = rstd * gamma[c];
    float cd = beta[c] - mean * ca;
    if (film) {  // y = gn(x) * (1 + scale) + shift  (improved_ddpm/unet.py:253-257)
      const float* fl = film + (int64_t)b * film_ld;
      const float sc = 1.f + fl[c];
      ca *= sc; cd = cd * sc + fl[C + c];
    }
    coef[((int64_t)b * 2) * C + c] = ca;
    coef[((int64_t)b * 2 + 1) * C + c] = cd;
  }
}

// fp16 pair of a scaled fp32 value: hi = fp16(v), lo = fp16(v - hi) (the difference is exact in fp32); saturating
__device__ inline void split_f16(float v, _Float16& hi, _Float16& lo, int* overflow) {
  if (overflow && !(fabsf(v) <= 65504.f)) *overflow = 1;  // also NaN; reported by the engine (capi.hip check_overflow)
  v = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
  hi = (_Float16)v;
  lo = (_Float16)(v - (float)hi);
}
typedef __attribute__((ext_vector_type(4))) _Float16 h4;

template <bool SPLIT>
__global__ void k_gn_apply_f32(const float* __restrict__ x0, const float* __restrict__ x1, int C0, int C1, int ld0,
                               int ld1, int HW, const float* __restrict__ coef, int silu, float* __restrict__ y,
                               int64_t nvec, int* overflow) {
  const int C = C0 + C1, C4 = C >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % C4);
    const int64_t bp = i / C4;
    const int b = (int)(bp / HW);
    const int c = cv * 4;
    const bool first = c < C0;
    const f4 v = first ? *(const f4*)(x0 + bp * ld0 + c) : *(const f4*)(x1 + bp * ld1 + (c - C0));
    const f4 a = *(const f4*)(coef + (int64_t)b * 2 * C + c);
    const f4 d = *(const f4*)(coef + ((int64_t)b * 2 + 1) * C + c);
    f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = v[e] * a[e] + d[e];
      o[e] = silu ? silu_acc(t) : t;
    }
    if (SPLIT) {
      h4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 a_, b_;
        split_f16(o[e] * kX3ActScale, a_, b_, overflow);
        hi[e] = a_; lo[e] = b_;
      }
      _Float16* t = (_Float16*)y + bp * (2 * C);
      *(h4*)(t + c) = hi;
      *(h4*)(t + C + c) = lo;
    } else {
      *(f4*)(y + bp * C + c) = o;
    }
  }
}
__global__ void k_avgpool2_split(const _Float16* __restrict__ x, _Float16* __restrict__ y, int B, int H, int W, int C,
                                 int* overflow) {
  const int Ho = H / 2, Wo = W / 2, C4 = C >> 2;
  const int64_t n = (int64_t)B * Ho * Wo * C4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % C4);
    int64_t t = i / C4;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const _Float16* s = x + (((int64_t)b * H + oy * 2) * W + ox * 2) * (2 * C) + cv * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // F.avg_pool2d order: (a + b + c + d) * 0.25 on the 22-bit values hi + lo
      const _Float16* sp = s + ((int64_t)(q >> 1) * W + (q & 1)) * (2 * C);
      const h4 hi = *(const h4*)sp, lo = *(const h4*)(sp + C);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += (float)hi[e] + (float)lo[e];
    }
    h4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 a_, b_;
      split_f16(acc[e] * 0.25f, a_, b_, overflow);
      hi[e] = a_; lo[e] = b_;
    }
    _Float16* o = y + (((int64_t)b * Ho + oy) * Wo + ox) * (2 * C) + cv * 4;
    *(h4*)o = hi;
    *(h4*)(o + C) = lo;
  }
}
// fp32 packed weights -> [wh | wh | wl] per filter tap (kernels.h kX3WgtScale)
__global__ void k_pack_w3(const float* __restrict__ w, _Float16* __restrict__ w3, int64_t rows_taps, int Cpad,
                          int* overflow) {
  const int64_t n = rows_taps * Cpad;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const int64_t rt = i / Cpad;
    _Float16 hi, lo;
    split_f16(w[i] * kX3WgtScale, hi, lo, overflow);
    _Float16* o = w3 + rt * (3 * Cpad);
    o[c] = hi; o[Cpad + c] = hi; o[2 * Cpad + c] = lo;
  }
}

// ---------------------------------------------------------------------------------------- attention
// softmax(scale * q k^T) v for the pixel U-Nets' AttentionBlock / AttnBlock (<= 1024 tokens per image):
// one wave per (image, head, query); scores in LDS; fp32 throughout (QKVAttentionLegacy upcasts its softmax to
// fp32 as well, improved_ddpm/unet.py:357).
__global__ __launch_bounds__(64) void k_attention_f32(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, float* __restrict__ o, int H, int T,
                                                      int D, int ldq, int ldk, int ldv, int ldo, float scale,
                                                      const float* __restrict__ obias) {
  extern __shared__ float sm[];  // [D] query | [T] scores
  float* qs = sm;
  float* ps = sm + D;
  const int lane = threadIdx.x;
  const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const float* qr = q + ((int64_t)b * T + qi) * ldq + h * D;
  for (int d = lane; d < D; d += 64) qs[d] = qr[d];
  __syncthreads();
  float mx = -INFINITY;
  for (int j = lane; j < T; j += 64) {
    const float* kr = k + ((int64_t)b * T + j) * ldk + h * D;
    float s = 0.f;
    for (int d = 0; d < D; d += 4) {
      const f4 kv = *(const f4*)(kr + d);
      s += qs[d] * kv[0] + qs[d + 1] * kv[1] + qs[d + 2] * kv[2] + qs[d + 3] * kv[3];
    }
    s *= scale;
    ps[j] = s;
    mx = fmaxf(mx, s);
  }
  for (int off = 32; off; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float den = 0.f;
  for (int j = lane; j < T; j += 64) {
    const float e = expf(ps[j] - mx);
    ps[j] = e;
    den += e;
  }
  for (int off = 32; off; off >>= 1) den += __shfl_xor(den, off);
  __syncthreads();
  const float inv = 1.0f / den;
  float* orow = o + ((int64_t)b * T + qi) * ldo + h * D;
  for (int d = lane; d < D; d += 64) {
    float a = 0.f;
    const float* vc = v + (int64_t)b * T * ldv + h * D + d;
    for (int j = 0; j < T; ++j) a += ps[j] * vc[(int64_t)j * ldv];
    orow[d] = a * inv + (obias ? obias[h * D + d] : 0.f);
  }
}

// ---------------------------------------------------------------------------------------- layout / resampling
__global__ void k_nchw_to_nhwc_f32(const float* __restrict__ x, float* __restrict__ y, int B, int C, int HW,
                                   int Cpad, float scale, float shift) {
  const int64_t n = (int64_t)B * HW * Cpad;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const int64_t bp = i / Cpad;
    const int b = (int)(bp / HW);
    const int pix = (int)(bp - (int64_t)b * HW);
    y[i] = (c < C) ? x[((int64_t)b * C + c) * HW + pix] * scale + shift : 0.f;
  }
}
// the same into the split form [pixel][hi(Cpad) | lo(Cpad)] of CD_PREC_F32X3 (the U-Net input feeds conv_in as a
// three-term product too)
__global__ void k_nchw_to_nhwc_split(const float* __restrict__ x, _Float16* __restrict__ y, int B, int C, int HW,
                                     int Cpad, float scale, float shift, int* overflow) {
  const int64_t n = (int64_t)B * HW * Cpad;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const int64_t bp = i / Cpad;
    const int b = (int)(bp / HW);
    const int pix = (int)(bp - (int64_t)b * HW);
    const float v = (c < C) ? x[((int64_t)b * C + c) * HW + pix] * scale + shift : 0.f;
    _Float16 hi, lo;
    split_f16(v * kX3ActScale, hi, lo, overflow);
    y[bp * (2 * Cpad) + c] = hi;
    y[bp * (2 * Cpad) + Cpad + c] = lo;
  }
}
__global__ void k_avgpool2_f32(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, C4 = C >> 2;
  const int64_t n = (int64_t)B * Ho * Wo * C4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % C4);
    int64_t t = i / C4;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const float* s = x + (((int64_t)b * H + oy * 2) * W + ox * 2) * C + cv * 4;
    const f4 a = *(const f4*)s, bq = *(const f4*)(s + C), c = *(const f4*)(s + (int64_t)W * C),
             d = *(const f4*)(s + (int64_t)W * C + C);
    f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (a[e] + bq[e] + c[e] + d[e]) * 0.25f;  // F.avg_pool2d: sum then / 4
    *(f4*)(y + i * 4) = o;
  }
}
__global__ void k_upsample2_f32(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C) {
  const int Ho = H * 2, Wo = W * 2, C4 = C >> 2;
  const int64_t n = (int64_t)B * Ho * Wo * C4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % C4);
    int64_t t = i / C4;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    *(f4*)(y + i * 4) = *(const f4*)(x + (((int64_t)b * H + (oy >> 1)) * W + (ox >> 1)) * C + cv * 4);
  }
}

inline int ew_grid(int64_t n) {
  const int64_t g = (n + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace f32_detail
using namespace f32_detail;

void launch_conv_gemm_f32(hipStream_t st, const ConvGemmParams& p) {
  const int Ctot = p.C0 + p.C1;
  CD_CHECK(p.C0 % 32 == 0 && p.C1 % 32 == 0, "conv_f32: channels must be multiples of 32 (C0=%d C1=%d)", p.C0, p.C1);
  CD_CHECK(p.Ktot == p.KH * p.KW * Ctot, "conv_f32: Ktot mismatch");
  CD_CHECK(p.M > 0 && p.N > 0 && p.nbatch == 1, "conv_f32: empty or batched problem");
  CD_CHECK(p.act != ACT_GEGLU && p.act != ACT_QGELU && !p.stats && p.splitk <= 1, "conv_f32: unsupported epilogue");
  CD_CHECK((p.ld0 % 4) == 0 && (p.src1 == nullptr || (p.ld1 % 4) == 0), "conv_f32: ld must be a multiple of 4");
  CD_CHECK(((uintptr_t)p.src0 & 15) == 0 && ((uintptr_t)p.wgt & 15) == 0, "conv_f32: 16-B alignment");
  KernelProfiler* prof = g_conv_prof;  // bench.py's roofline leg: per-launch HIP events on the launch stream
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof && prof->enabled) {
    char what[96] = "";
    if (prof->verbose) snprintf(what, sizeof(what), "f32 M%d N%d K%d k%d s%d", p.M, p.N, p.Ktot, p.KH, p.stride);
    prof->next_pair(&e0, &e1, 2.0 * (double)p.M * (double)p.N * (double)p.Ktot, what);
    (void)hipEventRecord(e0, st);
  }
  struct Closer {
    hipEvent_t e; hipStream_t s;
    ~Closer() { if (e) (void)hipEventRecord(e, s); }
  } closer{e1, st};
  const int64_t big = (int64_t)ceil_div(p.M, 128) * ceil_div(p.N, 64);
  if (big >= 256) {
    hipLaunchKernelGGL((k_conv_f32<128, 64>), dim3((unsigned)big), dim3(256), 0, st, p);
  } else {
    const int tiles = ceil_div(p.M, 64) * ceil_div(p.N, 64);
    hipLaunchKernelGGL((k_conv_f32<64, 64>), dim3(tiles), dim3(256), 0, st, p);
  }
}

int groupnorm_f32_slabs(int B, int HW) {
  (void)B;  // the partition of an image must not depend on the batch it travels in (same bits alone or batched)
  int S = 256;
  if (S > HW / 16) S = HW / 16;
  return S < 1 ? 1 : S;
}
size_t groupnorm_f32_workspace(int B, int HW, int C) {  // bytes
  return (size_t)B * groupnorm_f32_slabs(B, HW) * (C / 4) * 8 * sizeof(double) + (size_t)B * 2 * C * sizeof(float) + 256;
}
void launch_groupnorm_f32(hipStream_t st, const GroupNormParams& p, void* workspace) {
  const int C = p.C0 + p.C1;
  CD_CHECK(C % 32 == 0 && C <= 4096 && p.G == 32, "groupnorm_f32: channels %d", C);
  CD_CHECK(p.C0 % 4 == 0 && (p.ld0 % 4) == 0 && (!p.x1 || (p.ld1 % 4) == 0), "groupnorm_f32: alignment");
  const int S = groupnorm_f32_slabs(p.B, p.HW);
  double* part = (double*)workspace;
  float* coef = (float*)(part + (size_t)p.B * S * (C / 4) * 8);
  const float* x0 = (const float*)p.x;
  const float* x1 = (const float*)p.x1;
  if (p.pre0 && (p.C1 == 0 || p.pre1) && (p.HW % 32) == 0 && C / p.G <= 256) {
    hipLaunchKernelGGL(k_gn_fold_f32, dim3(p.G, p.B), dim3(256), 0, st, p.pre0, p.pre1, p.C0, p.C1, p.HW, p.G, p.eps,
                       p.gamma, p.beta, p.film, p.film_ld, coef);
  } else {
    hipLaunchKernelGGL(k_gn_partial_f32, dim3(S, p.B), dim3(256), 0, st, x0, x1, p.C0, p.C1, p.ld0, p.ld1, p.HW, S, part);
    hipLaunchKernelGGL(k_gn_coef_f32, dim3(p.G, p.B), dim3(64), 0, st, part, C, p.HW, S, p.G, p.eps, p.gamma, p.beta,
                       p.film, p.film_ld, coef);
  }
  const int64_t nvec = (int64_t)p.B * p.HW * (C / 4);
  if (p.split_out)
    hipLaunchKernelGGL(k_gn_apply_f32<true>, dim3(ew_grid(nvec)), dim3(256), 0, st, x0, x1, p.C0, p.C1, p.ld0, p.ld1,
                       p.HW, coef, p.silu, (float*)p.y, nvec, p.overflow);
  else
    hipLaunchKernelGGL(k_gn_apply_f32<false>, dim3(ew_grid(nvec)), dim3(256), 0, st, x0, x1, p.C0, p.C1, p.ld0, p.ld1,
                       p.HW, coef, p.silu, (float*)p.y, nvec, nullptr);
}
void launch_pack_w3(hipStream_t st, const float* w, bf16_t* w3, int Npad, int taps, int Cpad, int* overflow) {
  CD_CHECK(CD_ACT_FP16, "the split-fp16 mode of the fp32 path needs the fp16 build of the library");
  const int64_t rt = (int64_t)Npad * taps;
  hipLaunchKernelGGL(k_pack_w3, dim3(ew_grid(rt * Cpad)), dim3(256), 0, st, w, (_Float16*)w3, rt, Cpad,
                     overflow);
}
void launch_avgpool2_split(hipStream_t st, const bf16_t* x, bf16_t* y, int B, int H, int W, int C, int* overflow) {
  CD_CHECK(C % 4 == 0 && H % 2 == 0 && W % 2 == 0, "avgpool2_split: shape");
  hipLaunchKernelGGL(k_avgpool2_split, dim3(ew_grid((int64_t)B * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, st,
                     (const _Float16*)x, (_Float16*)y, B, H, W, C, overflow);
}

void launch_attention_f32(hipStream_t st, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                          float* o, int ldo, int B, int H, int T, int D, float scale, const float* obias) {
  CD_CHECK(D % 4 == 0 && (ldk % 4) == 0 && (size_t)(D + T) * 4 <= 64 * 1024, "attention_f32: D=%d T=%d", D, T);
  hipLaunchKernelGGL(k_attention_f32, dim3(T, H, B), dim3(64), (size_t)(D + T) * 4, st, q, k, v, o, H, T, D, ldq, ldk,
                     ldv, ldo, scale, obias);
}

void launch_nchw_to_nhwc_f32(hipStream_t st, const float* x, float* y, int B, int C, int HW, int Cpad, float scale,
                             float shift, int split, int* overflow) {
  if (split)
    hipLaunchKernelGGL(k_nchw_to_nhwc_split, dim3(ew_grid((int64_t)B * HW * Cpad)), dim3(256), 0, st, x, (_Float16*)y,
                       B, C, HW, Cpad, scale, shift, overflow);
  else
    hipLaunchKernelGGL(k_nchw_to_nhwc_f32, dim3(ew_grid((int64_t)B * HW * Cpad)), dim3(256), 0, st, x, y, B, C, HW, Cpad,
                       scale, shift);
}
void launch_avgpool2_f32(hipStream_t st, const float* x, float* y, int B, int H, int W, int C) {
  hipLaunchKernelGGL(k_avgpool2_f32, dim3(ew_grid((int64_t)B * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, st, x, y, B,
                     H, W, C);
}
void launch_upsample2_f32(hipStream_t st, const float* x, float* y, int B, int H, int W, int C) {
  hipLaunchKernelGGL(k_upsample2_f32, dim3(ew_grid((int64_t)B * H * 2 * W * 2 * (C / 4))), dim3(256), 0, st, x, y, B, H,
                     W, C);
}

}  // namespace cd
