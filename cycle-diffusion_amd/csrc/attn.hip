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
// The loop is VALU-bound for small heads (d = 40: 160 MFMA flops per score against max + fma + exp2 +
// convert), so the softmax works on raw scores (scale folded into the exp2 argument), masks only the
// ragged last tile, packs P with one convert per pair and - where V^T has a spare padded row - gets
// the row sums from the PV MFMA itself. fp32 softmax statistics and accumulation, 16-bit operands.
#include "common.h"
#include "kernels.h"

namespace cd {

namespace {

constexpr int KT = 64;  // keys per tile

// DH > 0: exact head dim known at compile time and DV > DH, so row DH of the staged V^T tile is set to
// ones and the softmax denominator falls out of the PV MFMA (summed from the same rounded
// probabilities the numerator uses) instead of 32 VALU adds per tile.
// NBUF = 2: the next tile's K / V^T are fetched into registers before the current tile's math and
// written to the other LDS buffer after it - one barrier per tile.
template <int DQK, int DV, int DH, int NBUF>
__global__ __launch_bounds__(256, DV <= 96 ? 2 : 1) void k_attention(AttnParams p) {
  constexpr int KLD = DQK + 8;  // elements per K row in LDS (16 B pad)
  constexpr int VLD = KT + 8;   // elements per V^T row in LDS
  constexpr int NKS = DQK / 16;
  constexpr int NDT = DV / 32;
  constexpr int CPR = DQK / 8;
  constexpr int NKR = (KT * CPR + 255) / 256;  // staging registers (uint4) per thread
  constexpr int NVR = DV * 8 / 256;
  constexpr bool ONES = DH > 0;
  static_assert(!ONES || (DH < DV && (DH & 7) == 0 && ((DH >> 2) & 1) == 0), "ones row placement");
  __shared__ __attribute__((aligned(16))) bf16_t Ks[NBUF][KT * KLD];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[NBUF][DV * VLD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int D = ONES ? DH : p.D;

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

  // ---- staging plan (fixed per thread): K tile [64][DQK] zero padded, V^T tile [DV][64]
  int k_goff[NKR], k_loff[NKR], k_row[NKR];
  int v_loff[NVR], v_row[NVR];
  int64_t v_goff[NVR];
#pragma unroll
  for (int i = 0; i < NKR; ++i) {
    const int id = tid + i * 256;
    const int row = id / CPR, ch = id % CPR;
    const bool ok = id < KT * CPR && ch * 8 < D;
    k_row[i] = ok ? row : (1 << 30);  // never < Tk
    k_goff[i] = row * p.ldk + ch * 8;
    k_loff[i] = id < KT * CPR ? row * KLD + ch * 8 : -1;
  }
#pragma unroll
  for (int i = 0; i < NVR; ++i) {
    const int id = tid + i * 256;
    const int row = id >> 3, ch = id & 7;
    v_row[i] = row;
    v_goff[i] = (int64_t)row * p.vt_tpad + ch * 8;
    v_loff[i] = row * VLD + ch * 8;
  }
  uint4 kreg[NKR], vreg[NVR];
  auto fetch = [&](int key0) {
#pragma unroll
    for (int i = 0; i < NKR; ++i) {
      kreg[i] = make_uint4(0, 0, 0, 0);
      if (key0 + k_row[i] < p.Tk && k_row[i] < KT) kreg[i] = *(const uint4*)(kb + (int64_t)key0 * p.ldk + k_goff[i]);
    }
#pragma unroll
    for (int i = 0; i < NVR; ++i) {
      vreg[i] = make_uint4(0, 0, 0, 0);
      if (v_row[i] < p.vt_dpad) vreg[i] = *(const uint4*)(vtb + v_goff[i] + key0);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NKR; ++i)
      if (k_loff[i] >= 0) *(uint4*)(&Ks[buf][k_loff[i]]) = kreg[i];
#pragma unroll
    for (int i = 0; i < NVR; ++i) {
      uint4 v = vreg[i];  // the ones row is substituted here so the loads stay in flight during the math
      if (ONES && v_row[i] == DH) v = make_uint4(kOnePair, kOnePair, kOnePair, kOnePair);
      *(uint4*)(&Vs[buf][v_loff[i]]) = v;
    }
  };

  f32x16 o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -INFINITY, l_part = 0.f;  // m_run in raw (unscaled) score units
  const float sc = p.scale * 1.44269504088896340736f;  // fold log2(e): softmax via exp2

  const int ntiles = (p.Tk + KT - 1) / KT;
  fetch(0);
  commit(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * KT;
    const int buf = NBUF == 2 ? (t & 1) : 0;
    const bool more = t + 1 < ntiles;
    if (more) fetch(key0 + KT);
    const bf16_t* Kt = Ks[buf];
    const bf16_t* Vt = Vs[buf];

    // ---- S^T = K Q^T for the two 32-key halves of the tile
    f32x16 s[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kh][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(Kt + (kh * 32 + qi) * KLD + ks * 16 + 8 * half);
        s[kh] = CD_MFMA_32x32x16(kf, qf[ks], s[kh]);
      }
    }
    // ---- keys beyond Tk (last tile only), running max in raw units
    if (key0 + KT > p.Tk || p.causal) {
      const int kmax = p.causal ? min(p.Tk - 1, q0 + qi) : p.Tk - 1;  // last visible key of this lane's query
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          s[kh][r] = key <= kmax ? s[kh][r] : -INFINITY;
        }
    }
    // v_max3 directly: fmaxf() would first canonicalise every MFMA result (one extra VALU op each)
    float mx = s[0][0];
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(s[0][0]), "v"(s[1][0]), "v"(s[0][1]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(s[1][1]), "v"(s[0][2]));
#pragma unroll
    for (int r = 3; r < 16; ++r)
      asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(s[1][r - 1]), "v"(s[0][r]));
    asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(mx), "v"(s[1][15]));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);  // first tile: exp2(-inf) = 0
    m_run = m_new;
    const float mneg = -m_new * sc;
    uint32_t pw[2][8];
    float ps = 0.f;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kh][r], sc, mneg));
        const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kh][r + 1], sc, mneg));
        if (!ONES) ps += e0 + e1;
        pw[kh][r >> 1] = pack2_unit(e0, e1);
      }
    if (!ONES) l_part = l_part * alpha + ps;
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;

    // ---- O^T += V^T P^T
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const uint4 praw = make_uint4(pw[kh][4 * s2], pw[kh][4 * s2 + 1], pw[kh][4 * s2 + 2], pw[kh][4 * s2 + 3]);
        const bf16x8 pf = __builtin_bit_cast(bf16x8, praw);
        const int kbase = kh * 32 + 16 * s2 + 4 * half;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const bf16_t* vr = Vt + (dt * 32 + qi) * VLD + kbase;
          const uint2 lo = *(const uint2*)(vr);
          const uint2 hi = *(const uint2*)(vr + 8);
          const bf16x8 vf = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
          o[dt] = CD_MFMA_32x32x16(vf, pf, o[dt]);
        }
      }

    if (NBUF == 1) __syncthreads();
    if (more) commit(NBUF == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }

  // ---- normalise and store: lane owns query q0+qi and 4 consecutive d per register quad
  float l_tot;
  if (ONES) {
    // row DH of O^T holds sum(p): tile DH/32, local row DH%32 -> register (l&3) + 4*(l>>3) of half 0
    constexpr int L = DH % 32;
    const float lv = o[DH / 32][(L & 3) + 4 * (L >> 3)];
    l_tot = __shfl(lv, qi);
  } else {
    l_tot = l_part + __shfl_xor(l_part, 32);
  }
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
#define CD_ATTN(DQK, DV, DH, NBUF) \
  hipLaunchKernelGGL((k_attention<DQK, DV, DH, NBUF>), grid, dim3(256), 0, st, p)
  if (p.D == 40) CD_ATTN(48, 64, 40, 2);       // SD / LDM 320-channel level
  else if (p.D == 80) CD_ATTN(80, 96, 80, 2);  // 640-channel level
  else if (p.D <= 32) CD_ATTN(32, 32, 0, 2);
  else if (p.D <= 48) CD_ATTN(48, 64, 0, 2);
  else if (p.D <= 64) CD_ATTN(64, 64, 0, 2);
  else if (p.D <= 80) CD_ATTN(80, 96, 0, 2);
  else if (p.D <= 96) CD_ATTN(96, 96, 0, 2);
  else if (p.D <= 128) CD_ATTN(128, 128, 0, 1);
  else CD_ATTN(160, 160, 0, 1);
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
