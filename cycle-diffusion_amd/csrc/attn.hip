// Fused softmax(Q K^T * scale) V for the U-Net attention layers (no S x S materialisation).
//
// Reference: CrossAttention.forward (ldm/modules/attention.py:170-193) - 8 heads, d_head 40/80/160,
// self-attention over 4096/1024/256/64 tokens and cross-attention over 77 context tokens;
// QKVAttentionLegacy (improved_ddpm/unet.py:318-345, 64-channel heads); Ho-DDPM AttnBlock
// (ddpm/diffusion.py:137-189, single head).
//
// CDNA4 mapping: 4 waves x 32 queries per workgroup, 64-key tiles staged in LDS.
//   S^T = K Q^T   via v_mfma_f32_32x32x16_bf16 with K as the A operand: every lane then owns ONE
//                 query column (lane&31) and 16 of the tile's 32 keys, so the online-softmax row
//                 statistics are per-lane registers plus one lane^32 exchange.
//   O^T = V^T P^T with V pre-transposed in HBM ([B][H][D][T], keys contiguous) so the A operand is
//                 two 8-byte LDS reads; P goes straight from the S accumulators to the B operand
//                 (the k-slot permutation of the accumulator layout is applied to V^T's reads).
// fp32 softmax statistics and accumulation, bf16 operands.
#include "common.h"
#include "kernels.h"

namespace cd {

namespace {

constexpr int KT = 64;  // keys per tile

template <int DQK, int DV>
__global__ __launch_bounds__(256) void k_attention(AttnParams p) {
  constexpr int KLD = DQK + 8;  // bf16 elements per K row in LDS (16 B pad)
  constexpr int VLD = KT + 8;   // bf16 elements per V^T row in LDS
  constexpr int NKS = DQK / 16;
  constexpr int NDT = DV / 32;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[KT * KLD];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[DV * VLD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int D = p.D;

  const bf16_t* qb = p.q + (int64_t)b * p.q_bs + h * D;
  const bf16_t* kb = p.k + (int64_t)b * p.k_bs + h * D;
  const bf16_t* vtb = p.vt + ((int64_t)b * p.H + h) * (int64_t)p.vt_dpad * p.vt_tpad;

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[q][ks*16 + 8*half .. +7]
  bf16x8 qf[NKS];
  {
    const int q = q0 + qi;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int d0 = ks * 16 + 8 * half;
      uint4 raw = make_uint4(0, 0, 0, 0);
      if (q < p.Tq && d0 < D) raw = *(const uint4*)(qb + (int64_t)q * p.ldq + d0);
      qf[ks] = *(bf16x8*)&raw;
    }
  }

  f32x16 o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -INFINITY, l_part = 0.f;
  const float sc = p.scale * 1.44269504088896340736f;  // fold log2(e): softmax via exp2

  const int ntiles = (p.Tk + KT - 1) / KT;
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * KT;
    __syncthreads();
    // ---- stage K tile [64][DQK] (zero padded) and V^T tile [DV][64]
    {
      constexpr int CPR = DQK / 8;
      for (int id = tid; id < KT * CPR; id += 256) {
        const int row = id / CPR, ch = id % CPR;
        uint4 raw = make_uint4(0, 0, 0, 0);
        const int key = key0 + row;
        if (key < p.Tk && ch * 8 < D) raw = *(const uint4*)(kb + (int64_t)key * p.ldk + ch * 8);
        *(uint4*)(Ks + row * KLD + ch * 8) = raw;
      }
      for (int id = tid; id < DV * 8; id += 256) {
        const int row = id >> 3, ch = id & 7;
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (row < p.vt_dpad) raw = *(const uint4*)(vtb + (int64_t)row * p.vt_tpad + key0 + ch * 8);
        *(uint4*)(Vs + row * VLD + ch * 8) = raw;
      }
    }
    __syncthreads();

    // ---- S^T = K Q^T for the two 32-key halves of the tile
    f32x16 s[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kh][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(Ks + (kh * 32 + qi) * KLD + ks * 16 + 8 * half);
        s[kh] = CD_MFMA_32x32x16(kf, qf[ks], s[kh]);
      }
    }
    // ---- scale, mask keys beyond Tk, running max
    float mx = -INFINITY;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = s[kh][r] * sc;
        v = key < p.Tk ? v : -INFINITY;
        s[kh][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);  // first tile: exp2(-inf) = 0
    m_run = m_new;
    float ps = 0.f;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = exp2f(s[kh][r] - m_new);
        s[kh][r] = e;
        ps += e;
      }
    l_part = l_part * alpha + ps;
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;

    // ---- O^T += V^T P^T
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (short)f2bf(s[kh][8 * s2 + j]);
        const int kbase = kh * 32 + 16 * s2 + 4 * half;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const bf16_t* vr = Vs + (dt * 32 + qi) * VLD + kbase;
          const bf16x4 lo = *(const bf16x4*)(vr);
          const bf16x4 hi = *(const bf16x4*)(vr + 8);
          bf16x8 vf;
          vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
          vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
          o[dt] = CD_MFMA_32x32x16(vf, pf, o[dt]);
        }
      }
  }

  // ---- normalise and store: lane owns query q0+qi and 4 consecutive d per register quad
  const float l_tot = l_part + __shfl_xor(l_part, 32);
  const float inv = 1.0f / l_tot;
  const int q = q0 + qi;
  if (q < p.Tq) {
    bf16_t* ob = p.o + (int64_t)b * p.o_bs + (int64_t)q * p.ldo + h * D;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int d0 = dt * 32 + 8 * rq + 4 * half;
        if (d0 < D) {
          float bb[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.obias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bb[e] = p.obias[h * D + d0 + e];
          }
          uint2 pk;
          pk.x = pack2(o[dt][rq * 4 + 0] * inv + bb[0], o[dt][rq * 4 + 1] * inv + bb[1]);
          pk.y = pack2(o[dt][rq * 4 + 2] * inv + bb[2], o[dt][rq * 4 + 3] * inv + bb[3]);
          *(uint2*)(ob + d0) = pk;
        }
      }
  }
}

// V [B][Tk][ldv] -> Vt [B][H][Dpad][Tpad]; grid (Tpad/64, H, B)
__global__ __launch_bounds__(256) void k_transpose_v(const bf16_t* __restrict__ v, int ldv, int64_t v_bs,
                                                     bf16_t* __restrict__ vt, int H, int Tk, int D,
                                                     int Dpad, int Tpad) {
  extern __shared__ __attribute__((aligned(16))) bf16_t tile[];  // [64][D+2]
  const int key0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int ldt = D + 2;
  const int cpr = D / 8;
  for (int id = threadIdx.x; id < 64 * cpr; id += 256) {
    const int row = id / cpr, ch = id % cpr;
    uint4 raw = make_uint4(0, 0, 0, 0);
    if (key0 + row < Tk) raw = *(const uint4*)(v + (int64_t)b * v_bs + (int64_t)(key0 + row) * ldv + h * D + ch * 8);
    bf16_t* d = tile + row * ldt + ch * 8;
    const bf16_t* rs = (const bf16_t*)&raw;
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = rs[e];
  }
  __syncthreads();
  bf16_t* ob = vt + ((int64_t)b * H + h) * (int64_t)Dpad * Tpad;
  for (int id = threadIdx.x; id < Dpad * 64; id += 256) {
    const int d = id >> 6, key = id & 63;
    bf16_t val = 0;
    if (d < D) val = tile[key * ldt + d];
    ob[(int64_t)d * Tpad + key0 + key] = val;
  }
}

}  // namespace

void launch_attention(hipStream_t st, const AttnParams& p) {
  CD_CHECK(p.D % 8 == 0 && p.D <= 160, "attention: head dim %d unsupported by the fused kernel", p.D);
  CD_CHECK(p.vt_tpad % KT == 0 && p.vt_tpad >= round_up(p.Tk, KT), "attention: V^T key padding");
  CD_CHECK((p.ldq % 8) == 0 && (p.ldk % 8) == 0 && (p.ldo % 4) == 0, "attention: leading dims");
  dim3 grid(ceil_div(p.Tq, 128), p.H, p.B);
#define CD_ATTN(DQK, DV) hipLaunchKernelGGL((k_attention<DQK, DV>), grid, dim3(256), 0, st, p)
  if (p.D <= 32) CD_ATTN(32, 32);
  else if (p.D <= 48) CD_ATTN(48, 64);
  else if (p.D <= 64) CD_ATTN(64, 64);
  else if (p.D <= 80) CD_ATTN(80, 96);
  else if (p.D <= 96) CD_ATTN(96, 96);
  else if (p.D <= 128) CD_ATTN(128, 128);
  else CD_ATTN(160, 160);
#undef CD_ATTN
}

void launch_transpose_v(hipStream_t st, const bf16_t* v, int ldv, int64_t v_bs, bf16_t* vt, int B,
                        int H, int Tk, int D, int Dpad, int Tpad) {
  CD_CHECK(D % 8 == 0 && Tpad % 64 == 0, "transpose_v: D %% 8, Tpad %% 64");
  const size_t lds = (size_t)64 * (D + 2) * sizeof(bf16_t);
  hipLaunchKernelGGL(k_transpose_v, dim3(Tpad / 64, H, B), dim3(256), lds, st, v, ldv, v_bs, vt, H,
                     Tk, D, Dpad, Tpad);
}

}  // namespace cd
