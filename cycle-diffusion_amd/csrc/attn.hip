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
//   O^T = V^T P^T P goes straight from the S accumulators to the B operand; the k-slot permutation of the
//                 accumulator layout is applied to the reads of the A operand V^T. Two V layouts:
//                 * token-major V ([B][T][ldv], as it leaves a fused q|k|v projection GEMM; VTOK = true): the tile
//                   is staged [key][d] like K and the A fragments are gfx950 LDS transpose reads
//                   (ds_read_b64_tr_b16: 16 lanes fetch a 4-key x 16-d block and each receives one d column), so
//                   no V^T tensor and no transposing GEMM exist anywhere (round 2: the V^T = Wv.X^T GEMMs were
//                   10 % of a step);
//                 * V pre-transposed in HBM ([B][H][D][T], keys contiguous; VTOK = false), two 16-byte reads.
// The loop is VALU-bound for small heads (d = 40: 160 MFMA flops per score; round-2 counters: 178 VALU
// instructions / 846 VALU-busy cycles per 64-key tile and wave against 448 MFMA cycles), so the per-score VALU
// work is cut to max + exp2 + convert:
//   * Q arrives in log2 units (softmax scale * log2(e) folded into the packed to_q weights, or applied
//     to the Q fragments once per workgroup), so a score needs no multiply;
//   * the running reference maximum m is kept as a 16-register block of -m that is the C operand of the
//     first QK^T MFMA: the accumulators come out as s - m and go straight into v_exp_f32;
//   * deferred maximum: m is only moved (accumulators, row sums and the -m block rescaled) when some row
//     of the wave exceeds it by more than 2^8 - a wave-uniform, rare branch; between moves probabilities
//     lie in (0, 256], exact for the final o / l because numerator and denominator share m;
//   * the ragged last tile alone is masked, P is packed with one convert per pair and - where V^T has a
//     spare padded row - the row sums fall out of the PV MFMA itself.
// fp32 softmax statistics and accumulation, 16-bit operands.
#include <stdlib.h>

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
// NWV = waves per workgroup (32 queries each): 4, or 8 - a 256-query workgroup stages every K / V tile once for twice
// the queries (half the L2 -> LDS traffic and half the staging instructions per query; same waves per SIMD)
// DBG (probe build only, CD_ATTN_DBG: timing experiments whose RESULTS ARE WRONG - which part of a tile costs what):
//   1 no K / V fetch and commit after the first tile   2 no barrier in the loop   8 no exponentials (p = s)
//   16 no PV MFMAs   32 no QK^T MFMAs   64 no maximum / deferred-maximum test
template <int DQK, int DV, int DH, int NBUF, bool VTOK, int KG, int NWV = 4, int DBG = 0>
__global__ __launch_bounds__(64 * NWV, DV <= 96 ? (KG == 32 ? (DV <= 64 ? 4 : 3) : 2) : 1) void k_attention(AttnParams p) {
  constexpr int NT = 64 * NWV;  // threads per workgroup
  constexpr int KLD = DQK + 8;  // elements per K row in LDS (16 B pad)
  constexpr int VLD = KT + 8;   // elements per V^T row in LDS
  constexpr int NKS = DQK / 16;
  constexpr int NDT = DV / 32;
  constexpr int CPR = DQK / 8;
  constexpr int NKR = (KT * CPR + NT - 1) / NT;  // staging registers (uint4) per thread
  constexpr bool ONES = DH > 0;
  // token-major V tile in LDS: [key][VS]; the row stride is 16 or 48 dwords mod 64, so the four key rows x 64 bytes
  // that the 32 lanes of a transpose-read group touch fall on 64 distinct banks
  constexpr int VS = DV <= 32 ? 32 : (DV <= 96 ? 96 : 160);
  constexpr int CPRV = ONES ? DH / 8 + 1 : DV / 8;  // 16-byte chunks staged per key row (the last one: ones column)
  constexpr int NVR = VTOK ? (KT * CPRV + NT - 1) / NT : (DV * 8 + NT - 1) / NT;
  constexpr int VBUF = VTOK ? KT * VS : DV * VLD;
  static_assert(!ONES || (DH < DV && (DH & 7) == 0 && ((DH >> 2) & 1) == 0), "ones row placement");
  __shared__ __attribute__((aligned(16))) bf16_t Ks[NBUF][KT * KLD];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[NBUF][VBUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * (32 * NWV) + wave * 32;
  const int D = ONES ? DH : p.D;

  const bf16_t* qb = p.q + (int64_t)b * p.q_bs + h * D;
  const bf16_t* kb = p.k + (int64_t)b * p.k_bs + h * D;
  const bf16_t* vtb = VTOK ? p.v + (int64_t)b * p.v_bs + h * D
                           : p.vt + ((int64_t)b * p.H + h) * (int64_t)p.vt_dpad * p.vt_tpad;

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[q][ks*16 + 8*half .. +7], in log2 score units
  bf16x8 qf[NKS];
  const float qsc = p.scale * 1.44269504088896340736f;
  {
    const int q = q0 + qi;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int d0 = ks * 16 + 8 * half;
      uint4 raw = make_uint4(0, 0, 0, 0);
      if (q < p.Tq && d0 < D) raw = *(const uint4*)(qb + (int64_t)q * p.ldq + d0);
      if (!p.q_log2) {  // q not produced in log2 units: scale the fragment here (one extra 16-bit rounding)
        float f[8];
        unpack8(raw, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= qsc;
        raw = pack8(f);
      }
      qf[ks] = *(bf16x8*)&raw;
    }
  }

  // ---- staging plan (fixed per thread): K tile [64][DQK] zero padded, V^T tile [DV][64] / V tile [64][VS].
  // Global reads are buffer loads: a 32-bit per-thread byte offset (kNoLoad = out of range for inactive slots) plus the
  // tile's position, and a descriptor whose size ends at the last valid row - rows beyond Tk (beyond vt_dpad for
  // V^T) arrive as zeros from the bounds check, without branches or 64-bit address arithmetic.
  constexpr unsigned kNoLoad = 0x80000000u;
  const unsigned k_bytes = p.Tk > 0 ? (unsigned)(((int64_t)(p.Tk - 1) * p.ldk + D) * 2) : 0u;
  const unsigned v_bytes = VTOK ? (p.Tk > 0 ? (unsigned)(((int64_t)(p.Tk - 1) * p.ldv + D) * 2) : 0u)
                                : (unsigned)((int64_t)p.vt_dpad * p.vt_tpad * 2);
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vtb, 0, v_bytes, 0x00020000);
  unsigned k_voff[NKR], v_voff[NVR];
  int k_loff[NKR], v_loff[NVR];
  bool v_ones[NVR];
#pragma unroll
  for (int i = 0; i < NKR; ++i) {
    const int id = tid + i * NT;
    const int row = id / CPR, ch = id % CPR;
    const bool ok = id < KT * CPR && ch * 8 < D;  // columns D .. DQK-1 of the tile are zero padding
    k_voff[i] = ok ? (unsigned)((row * p.ldk + ch * 8) * 2) : kNoLoad;
    k_loff[i] = id < KT * CPR ? row * KLD + ch * 8 : -1;
  }
#pragma unroll
  for (int i = 0; i < NVR; ++i) {
    const int id = tid + i * NT;
    if (VTOK) {  // [key][d] like K: chunk ch of key row `row`; chunk CPRV-1 = the ones column when ONES
      const int row = id / CPRV, ch = id % CPRV;
      const bool data = id < KT * CPRV && ch * 8 < D;
      v_voff[i] = data ? (unsigned)((row * p.ldv + ch * 8) * 2) : kNoLoad;
      v_loff[i] = id < KT * CPRV ? row * VS + ch * 8 : -1;
      v_ones[i] = ONES && ch * 8 == DH;
    } else {
      const int row = id >> 3, ch = id & 7;
      v_voff[i] = row < DV ? (unsigned)((row * p.vt_tpad + ch * 8) * 2) : kNoLoad;  // rows >= vt_dpad: zeros
      v_loff[i] = row < DV ? row * VLD + (ch >> 1) * 16 + (ch & 1) * 4 : -1;
      v_ones[i] = ONES && row == DH;
    }
  }
  uint4 kreg[NKR], vreg[NVR];
  auto fetch = [&](int key0) {
    // the tile position is ADDED TO THE VECTOR OFFSET (one v_add per load): the hardware bounds check covers the
    // vector offset only - a scalar offset is excluded from it, and rows beyond Tk of a ragged last tile (cross-
    // attention: 77 keys) would be read from whatever follows the tensor instead of arriving as zeros
    const unsigned k_toff = (unsigned)(key0 * p.ldk * 2);
    const unsigned v_toff = (unsigned)(VTOK ? key0 * p.ldv * 2 : key0 * 2);
#pragma unroll
    for (int i = 0; i < NKR; ++i)
      kreg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_k, k_voff[i] + k_toff, 0, 0));
#pragma unroll
    for (int i = 0; i < NVR; ++i)
      vreg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, v_voff[i] + v_toff, 0, 0));
  };
  // VTOK = false: V^T rows are stored with the four 4-key pieces of every 16-key group in the order [0 2 1 3]: the
  // two pieces a lane feeds to one PV MFMA (keys 4*half.. and 8+4*half.., the accumulator layout of S^T) are then one
  // aligned 16-byte read (row stride 36 dwords: conflict-free ds_read_b128) instead of two 8-byte ones.
  // The ones row / column is substituted here so the loads stay in flight during the math.
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NKR; ++i)
      if (k_loff[i] >= 0) *(uint4*)(&Ks[buf][k_loff[i]]) = kreg[i];
#pragma unroll
    for (int i = 0; i < NVR; ++i) {
      uint4 v = vreg[i];
      if (VTOK) {
        if (v_ones[i]) v = make_uint4(kOnePair & 0xffffu, 0, 0, 0);  // column DH = 1, the rest of the chunk 0
        if (v_loff[i] >= 0) *(uint4*)(&Vs[buf][v_loff[i]]) = v;
      } else {
        if (v_ones[i]) v = make_uint4(kOnePair, kOnePair, kOnePair, kOnePair);
        if (v_loff[i] >= 0) {
          *(uint2*)(&Vs[buf][v_loff[i]]) = make_uint2(v.x, v.y);
          *(uint2*)(&Vs[buf][v_loff[i] + 8]) = make_uint2(v.z, v.w);
        }
      }
    }
  };
  // transpose-read geometry (VTOK): 16 lanes fetch the 4-key x 16-d block [4 * half + (i >> 2)][16 * g + 4 * (i & 3)..]
  // (i = lane & 15, g = (lane >> 4) & 1) and lane l receives d column 16 * g + i = lane & 31 of its 4 keys
  const int vtr_off = ((4 * half + ((lane & 15) >> 2)) * VS + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));

  f32x16 o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  f32x16 negm;  // -m of this lane's query in every register: C operand of the first QK^T MFMA of a tile
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.f;
  float l_part = 0.f;
  constexpr float kDefer = 8.0f;  // log2 units: probabilities stay <= 2^8 between moves of m

  const int ntiles = (p.Tk + KT - 1) / KT;
  fetch(0);
  commit(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * KT;
    const int buf = NBUF == 2 ? (t & 1) : 0;
    const bool more = t + 1 < ntiles;
    if (more && !(DBG & 1)) fetch(key0 + KT);
    const bf16_t* Kt = Ks[(DBG & 1) ? 0 : buf];
    const bf16_t* Vt = Vs[(DBG & 1) ? 0 : buf];

    // The tile is consumed in groups of KG keys (KG = 64: both 32-key halves at once; KG = 32: one half at a time -
    // half the score / probability registers live, one more wave per SIMD where that crosses an occupancy step).
#pragma unroll
    for (int g = 0; g < KT / KG; ++g) {
      constexpr int HPG = KG / 32;  // 32-key halves per group
      // ---- S^T - m = K Q^T - m
      f32x16 s[HPG];
      {
        bf16x8 kf[HPG][NKS];  // all K fragments first: the LDS latency is paid once, not per MFMA
#pragma unroll
        for (int kh = 0; kh < HPG; ++kh)
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks)
            kf[kh][ks] = *(const bf16x8*)(Kt + ((g * HPG + kh) * 32 + qi) * KLD + ks * 16 + 8 * half);
        if (KG == 64) __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs (else re-serialised)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
          for (int kh = 0; kh < HPG; ++kh) {  // the 32-key halves alternate: no back-to-back dependent MFMAs
            if constexpr (DBG & 32) {
              if (ks == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kh][r] = negm[r] + __builtin_bit_cast(float, (uint32_t)kf[kh][NKS - 1][r & 7] << 16) * 1e-30f;
              }
            } else {
              s[kh] = CD_MFMA_32x32x16(kf[kh][ks], qf[ks], ks == 0 ? negm : s[kh]);
            }
          }
      }
      // ---- keys beyond Tk (last tile only) / causal mask
      if (key0 + KT > p.Tk || p.causal) {
        const int kmax = p.causal ? min(p.Tk - 1, q0 + qi) : p.Tk - 1;  // last visible key of this lane's query
#pragma unroll
        for (int kh = 0; kh < HPG; ++kh)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = key0 + (g * HPG + kh) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            s[kh][r] = key <= kmax ? s[kh][r] : -INFINITY;
          }
      }
      // group maximum relative to m. v_max3 through asm: fmaxf() would first canonicalise every MFMA result (one extra
      // VALU op each). hipcc pads no hazards for an asm statement, and an MFMA's D needs 12 wait states before any
      // reader: the chain therefore STARTS with a compiler-visible VALU read (v_med3 with +inf = max) of the accumulator
      // the last MFMA wrote - hipcc pads that one, every asm op depends on it, and all earlier MFMAs have retired by then.
      // (Rounds 1-2a started with the asm op: a latent hazard, visible as run-to-run differences once the 32-key groups
      // put the reader right behind a dependent MFMA chain.)
      float mx = __builtin_amdgcn_fmed3f(s[HPG - 1][14], s[HPG - 1][15], INFINITY);
      if constexpr (!(DBG & 64)) {
      if (HPG == 2) {
#pragma unroll
        for (int r = 0; r < 14; r += 2)
          asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(s[1][r]), "v"(s[1][r + 1]));
      }
#pragma unroll
      for (int r = 0; r < (HPG == 2 ? 16 : 14); r += 2)
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(s[0][r]), "v"(s[0][r + 1]));
      }
      // ---- move m (rare, wave-uniform): always on the first group, later only if a row grew past 2^kDefer.
      // Textbook order: the decision precedes the exponentiation of the keys it covers, and everything
      // accumulated against the old m (o, the row sums inside o or l_part) is rescaled exactly once.
      const bool first = t == 0 && g == 0;
      if (first || (!(DBG & 64) && __any(mx > kDefer))) {
        const float mxq = fmaxf(mx, __shfl_xor(mx, 32));  // both half-lanes of a query agree
        float delta = first ? mxq : fmaxf(mxq, 0.f);
        delta = delta == -INFINITY ? 0.f : delta;
        // on the first group o and l_part are still zero: a reference point at or below -128 (log2 units) would
        // make exp2(-delta) = +inf and 0 * inf = NaN
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int kh = 0; kh < HPG; ++kh)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kh][r] -= delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] -= delta;
#pragma unroll
        for (int i = 0; i < NDT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        l_part *= alpha;
      }
      uint32_t pw[HPG][8];
      float ps = 0.f;
#pragma unroll
      for (int kh = 0; kh < HPG; ++kh)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float e0 = (DBG & 8) ? s[kh][r] : __builtin_amdgcn_exp2f(s[kh][r]);
          const float e1 = (DBG & 8) ? s[kh][r + 1] : __builtin_amdgcn_exp2f(s[kh][r + 1]);
          if (!ONES) ps += e0 + e1;
          pw[kh][r >> 1] = pack2_prob(e0, e1);
        }
      if (!ONES) l_part += ps;

      // ---- O^T += V^T P^T
#pragma unroll
      for (int kh = 0; kh < HPG; ++kh)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const uint4 praw = make_uint4(pw[kh][4 * s2], pw[kh][4 * s2 + 1], pw[kh][4 * s2 + 2], pw[kh][4 * s2 + 3]);
          const bf16x8 pf = __builtin_bit_cast(bf16x8, praw);
          const int k16 = (g * HPG + kh) * 32 + 16 * s2;  // first key of this 16-key MFMA slice
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            bf16x8 vf;
            if (VTOK) {
              typedef __attribute__((address_space(3))) bf16x4* lds4_t;
              const bf16_t* vr = Vt + vtr_off + k16 * VS + dt * 32;
              const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)vr);
              const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vr + 8 * VS));
              vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            } else {
              vf = *(const bf16x8*)(Vt + (dt * 32 + qi) * VLD + k16 + 8 * half);
            }
            if constexpr (DBG & 16) o[dt][0] += __builtin_bit_cast(float, (uint32_t)vf[0] << 16) * __builtin_bit_cast(float, praw.x);
            else o[dt] = CD_MFMA_32x32x16(vf, pf, o[dt]);
          }
        }
    }

    if (NBUF == 1 && !(DBG & 2)) __syncthreads();
    if (more && !(DBG & 1)) commit(NBUF == 2 ? (buf ^ 1) : 0);
    if (!(DBG & 2)) __syncthreads();
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

// ---------------------------------------------------------------------------------------------------------------------
// d = 40 self-attention over long sequences (the 320-channel level of the SD / LDM U-Nets: 4096 / 1024 tokens, 8 heads;
// attention.py:171-200) - round 5. Same algorithm and the same numerics as k_attention<48, 64, 40, 2, false, 32, 8> above
// (S^T = K Q^T with -m as the C operand, deferred maximum, ones row for the denominators, V^T pre-transposed in HBM), with
// three changes to what a 64-key tile costs:
//   * K and V^T tiles go HBM / L2 -> LDS by `buffer_load_dwordx4 ... lds` (as conv_gemm.hip stages its operands): no
//     staging registers, no ds_write, 13 wave-wide copies per tile and workgroup instead of 16 loads + 24 LDS stores.
//     LDS rows are 128 bytes with the 16-byte chunks XOR-swizzled on the SOURCE side (chunk c of row r sits at
//     c ^ ((r >> 1) & 7)): every fragment read below is the conflict-free ds_read_b128 pattern of conv_gemm.hip's BK = 64
//     tiles. The zero padding of K (d 40 .. 47) arrives as the zeros of an out-of-range offset.
//   * the K rows of a tile are gathered in PERMUTED key order - the two middle 4-row pieces of every 16 rows swapped - so
//     that the 8 probabilities a lane feeds to one PV MFMA (accumulator rows {0-3, 8-11} + 4 half of S^T) belong to 8
//     CONSECUTIVE keys: V^T is then read in its natural order (one aligned 16-byte read per fragment) and needs no
//     permuted copy.
//   * rows 32 .. 47 of O^T (d 32 .. 39, the ones row 40, 7 padding rows) are a 16-row block on v_mfma_f32_16x16x32 instead
//     of a second 32-row block: the P operand of the 16 x 16 form (lane = query n & 15, k group lane >> 4) comes out of the
//     32 x 32 layout by one v_permlane16_swap per register pair - X = keys 0 .. 15, Y = keys 16 .. 31 of the group:
//     swap(X, Y) leaves queries 0 .. 15 with the four k groups [X.h0 | Y.h0 | X.h1 | Y.h1] in X and queries 16 .. 31 in Y.
//     Matrix work per 32-key group: 3 (QK^T) + 2 (PV rows 0 .. 31) + 2 x 1/2 (rows 32 .. 47) = 6 instead of 7 32 x 32 x 16
//     equivalents, one LDS fragment read less, 8 accumulator registers less.
// 8 waves x 32 queries per workgroup, two LDS buffers, one barrier per tile.
template <int NWV>
__global__ __launch_bounds__(64 * NWV, 4) void k_attention_d40(AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(NWV == 8, "one K copy per wave and tile");
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr int D = 40, NKS = 3, ROWB = 128;
  constexpr int KBUF = KT * ROWB;   // K tile: [64 keys (permuted)][64 x 16 bit]: d 0 .. 39 | zeros
  constexpr int VBUF = 48 * ROWB;   // V^T tile: [48 d rows][64 keys]: rows 0 .. 39 data, row 40 ones, 41 .. 47 zeros
  constexpr int BUF = KBUF + VBUF;
  __shared__ __attribute__((aligned(1024))) char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * (32 * NWV) + wave * 32;

  const bf16_t* qb = p.q + (int64_t)b * p.q_bs + h * D;
  const bf16_t* kb = p.k + (int64_t)b * p.k_bs + h * D;
  const bf16_t* vtb = p.vt + ((int64_t)b * p.H + h) * (int64_t)p.vt_dpad * p.vt_tpad;

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[q][ks * 16 + 8 half .. + 7] in log2 score units
  bf16x8 qf[NKS];
  {
    const float qsc = p.scale * 1.44269504088896340736f;
    const int q = q0 + qi;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int d0 = ks * 16 + 8 * half;
      uint4 raw = make_uint4(0, 0, 0, 0);
      if (q < p.Tq && d0 < D) raw = *(const uint4*)(qb + (int64_t)q * p.ldq + d0);
      if (!p.q_log2) {
        float f[8];
        unpack8(raw, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= qsc;
        raw = pack8(f);
      }
      qf[ks] = *(bf16x8*)&raw;
    }
  }

  // ---- copy geometry: one wave-wide copy = 8 LDS rows x 128 B; lane l fills physical chunk l & 7 of row 8 x + (l >> 3)
  // from logical chunk (l & 7) ^ ((row >> 1) & 7). K: copy `wave` of 8; V^T: waves 0 .. 4 (d rows 0 .. 39).
  constexpr unsigned kNoLoad = 0x80000000u;
  const unsigned k_bytes = p.Tk > 0 ? (unsigned)(((int64_t)(p.Tk - 1) * p.ldk + D) * 2) : 0u;
  const unsigned v_bytes = (unsigned)((int64_t)p.vt_dpad * p.vt_tpad * 2);
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vtb, 0, v_bytes, 0x00020000);
  unsigned k_voff, v_voff;
  {
    const int row = 8 * wave + (lane >> 3);
    const int lc = (lane & 7) ^ ((row >> 1) & 7);
    const int pc = (row >> 2) & 3;                                             // 4-row piece within its 16 rows
    const int key = (row & ~12) | ((((pc & 1) << 1) | (pc >> 1)) << 2);        // pieces [0 2 1 3]
    k_voff = lc < D / 8 ? (unsigned)((key * p.ldk + lc * 8) * 2) : kNoLoad;    // chunks 5 .. 7: zeros
    v_voff = (row < D && row < p.vt_dpad) ? (unsigned)((row * p.vt_tpad + lc * 8) * 2) : kNoLoad;
  }
  auto fetch = [&](int key0, int buf) {
    // the tile position is added to the VECTOR offset: the descriptor's bounds check (rows beyond Tk -> zeros) covers it only
    char* kd = lds + buf * BUF + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lptr_t)kd, 16, k_voff + (unsigned)(key0 * p.ldk * 2), 0, 0, 0);
    if (wave < D / 8)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lptr_t)(kd + KBUF), 16, v_voff + (unsigned)(key0 * 2), 0, 0, 0);
  };
  // rows 40 .. 47 of both V^T buffers are never copied over: the ones row and its padding, written once
  if (wave == 5 || wave == 6) {
    const uint4 v = lane < 8 ? make_uint4(kOnePair, kOnePair, kOnePair, kOnePair) : make_uint4(0, 0, 0, 0);
    *(uint4*)(lds + (wave - 5) * BUF + KBUF + D * ROWB + lane * 16) = v;
  }

  // ---- fragment read offsets (bytes inside a buffer)
  const int sw = (qi >> 1) & 7;                    // rows qi and 32 + qi share it: (32 >> 1) & 7 == 0
  const int kf_off = qi * ROWB;                    // + 32 ROWB per key half; chunk (2 ks + half) ^ sw
  const int vf_off = KBUF + qi * ROWB;             // chunk (4 g + 2 s2 + half) ^ sw
  const int tm = lane & 15, tg = lane >> 4;        // 16 x 16 x 32 A operand: d row 32 + tm, k group tg = keys [0 2 1 3][tg] * 8
  const int vt_off = KBUF + (32 + tm) * ROWB;
  const int vt_sw = ((32 + tm) >> 1) & 7, vt_lc = ((tg & 1) << 1) | (tg >> 1);

  f32x16 o0;
  f32x4 ot0 = {0.f, 0.f, 0.f, 0.f}, ot1 = {0.f, 0.f, 0.f, 0.f};
  f32x16 negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; negm[r] = 0.f; }
  constexpr float kDefer = 8.0f;

  const int ntiles = (p.Tk + KT - 1) / KT;
  fetch(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * KT;
    const int buf = t & 1;
    if (t + 1 < ntiles) fetch(key0 + KT, buf ^ 1);
    const char* Bt = lds + buf * BUF;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      // ---- S^T - m = K Q^T - m over the 32 (permuted) keys of this group
      f32x16 s;
      {
        bf16x8 kf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
          kf[ks] = *(const bf16x8*)(Bt + kf_off + g * 32 * ROWB + (((2 * ks + half) ^ sw) << 4));
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) s = CD_MFMA_32x32x16(kf[ks], qf[ks], ks == 0 ? negm : s);
      }
      if (key0 + KT > p.Tk || p.causal) {
        const int kmax = p.causal ? min(p.Tk - 1, q0 + qi) : p.Tk - 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;  // LDS row of the group -> key through the piece swap
          const int pc = (row >> 2) & 3;
          const int key = key0 + g * 32 + ((row & ~12) | ((((pc & 1) << 1) | (pc >> 1)) << 2));
          s[r] = key <= kmax ? s[r] : -INFINITY;
        }
      }
      // group maximum relative to m (the chain starts with a compiler-visible read of the last MFMA's result: see above)
      float mx = __builtin_amdgcn_fmed3f(s[14], s[15], INFINITY);
#pragma unroll
      for (int r = 0; r < 14; r += 2) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(s[r]), "v"(s[r + 1]));
      const bool first = t == 0 && g == 0;
      if (first || __any(mx > kDefer)) {
        const float mxq = fmaxf(mx, __shfl_xor(mx, 32));
        float delta = first ? mxq : fmaxf(mxq, 0.f);
        delta = delta == -INFINITY ? 0.f : delta;
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] -= delta; negm[r] -= delta; o0[r] *= alpha; }
        // the 16-row block holds query (lane & 15) + 16 qh in every lane: its factor lives in lane (lane & 15) + 16 qh
        const float a0 = __shfl(alpha, lane & 15), a1 = __shfl(alpha, 16 + (lane & 15));
#pragma unroll
        for (int r = 0; r < 4; ++r) { ot0[r] *= a0; ot1[r] *= a1; }
      }
      uint32_t pw[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2)
        pw[r >> 1] = pack2_prob(__builtin_amdgcn_exp2f(s[r]), __builtin_amdgcn_exp2f(s[r + 1]));
      // ---- O^T rows 0 .. 31 += V^T P^T (two 16-key slices)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 vf = *(const bf16x8*)(Bt + vf_off + (((4 * g + 2 * s2 + half) ^ sw) << 4));
        const uint4 praw = make_uint4(pw[4 * s2], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]);
        o0 = CD_MFMA_32x32x16(vf, __builtin_bit_cast(bf16x8, praw), o0);
      }
      // ---- rows 32 .. 47: P^T regrouped for the 16 x 16 x 32 form, one A fragment for both query halves
      {
        uint32_t xa[4], ya[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const auto r2 = __builtin_amdgcn_permlane16_swap(pw[i], pw[4 + i], false, false);
          xa[i] = r2[0]; ya[i] = r2[1];
        }
        const bf16x8 va = *(const bf16x8*)(Bt + vt_off + (((4 * g + vt_lc) ^ vt_sw) << 4));
        ot0 = CD_MFMA_16x16x32(va, __builtin_bit_cast(bf16x8, make_uint4(xa[0], xa[1], xa[2], xa[3])), ot0);
        ot1 = CD_MFMA_16x16x32(va, __builtin_bit_cast(bf16x8, make_uint4(ya[0], ya[1], ya[2], ya[3])), ot1);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's copies of the next tile have landed
    __syncthreads();                                   // ... everyone's; and everyone is done reading this tile
  }

  // ---- normalise and store. Denominators: row d = 40 of O^T = register 0 of lanes 32 .. 47 of the 16-row blocks
  const float l0 = __shfl(ot0[0], 32 + (lane & 15)), l1 = __shfl(ot1[0], 32 + (lane & 15));
  {
    const int q = q0 + qi;
    const float inv = 1.0f / ((qi & 16) ? l1 : l0);
    if (q < p.Tq) {
      bf16_t* ob = p.o + (int64_t)b * p.o_bs + (int64_t)q * p.ldo + h * D;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int d0 = 8 * rq + 4 * half;
        float bb[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.obias) {
#pragma unroll
          for (int e = 0; e < 4; ++e) bb[e] = p.obias[h * D + d0 + e];
        }
        uint2 pk;
        pk.x = pack2(o0[rq * 4 + 0] * inv + bb[0], o0[rq * 4 + 1] * inv + bb[1]);
        pk.y = pack2(o0[rq * 4 + 2] * inv + bb[2], o0[rq * 4 + 3] * inv + bb[3]);
        *(uint2*)(ob + d0) = pk;
      }
    }
  }
  if (tg < 2) {  // d 32 .. 39: lane holds d = 32 + 4 tg + r of query tm (+ 16)
    const int d0 = 32 + 4 * tg;
    float bb[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.obias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bb[e] = p.obias[h * D + d0 + e];
    }
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
      const int q = q0 + tm + 16 * qh;
      const f32x4 ov = qh ? ot1 : ot0;
      const float inv = 1.0f / (qh ? l1 : l0);
      if (q < p.Tq) {
        uint2 pk;
        pk.x = pack2(ov[0] * inv + bb[0], ov[1] * inv + bb[1]);
        pk.y = pack2(ov[2] * inv + bb[2], ov[3] * inv + bb[3]);
        *(uint2*)(p.o + (int64_t)b * p.o_bs + (int64_t)q * p.ldo + h * D + d0) = pk;
      }
    }
  }
#endif
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
  CD_CHECK((p.v != nullptr) != (p.vt != nullptr), "attention: exactly one of v (token-major) / vt (transposed)");
  if (p.vt) CD_CHECK(p.vt_tpad % KT == 0 && p.vt_tpad >= round_up(p.Tk, KT), "attention: V^T key padding");
  else CD_CHECK((p.ldv % 8) == 0 && ((uintptr_t)p.v & 15) == 0, "attention: V row stride / alignment");
  CD_CHECK((p.ldq % 8) == 0 && (p.ldk % 8) == 0 && (p.ldo % 4) == 0, "attention: leading dims");
  dim3 grid(ceil_div(p.Tq, 128), p.H, p.B);
#define CD_ATTN_KG(DQK, DV, DH, NBUF, KG)                                                                    \
  do {                                                                                                      \
    if (p.v) hipLaunchKernelGGL((k_attention<DQK, DV, DH, NBUF, true, KG>), grid, dim3(256), 0, st, p);      \
    else hipLaunchKernelGGL((k_attention<DQK, DV, DH, NBUF, false, KG>), grid, dim3(256), 0, st, p);         \
  } while (0)
#define CD_ATTN(DQK, DV, DH, NBUF) CD_ATTN_KG(DQK, DV, DH, NBUF, 64)
  // d = 40 (SD / LDM 320-channel level): 32-key softmax groups fit 128 VGPRs = four waves per SIMD (2 % faster than
  // 64-key groups at three: profiles/r2b_attention_v_layouts.txt); CD_ATTN_KG32=0 selects the 64-key form for A/B runs
  static const bool half_groups = [] { const char* e = getenv("CD_ATTN_KG32"); return !(e && e[0] == '0'); }();
  // 256-query workgroups (8 waves) for the long self-attention of the 64 x 64 level: CD_ATTN_W8=0 selects 128 (A/B)
  static const bool wide_wg = [] { const char* e = getenv("CD_ATTN_W8"); return !(e && e[0] == '0'); }();
  if (p.D == 40 && half_groups && wide_wg && p.Tq >= 1024 && p.Tk >= 1024 && p.vt) {
#ifdef CD_PROBE
    static const int dbg = [] { const char* e = getenv("CD_ATTN_DBG"); return e ? atoi(e) : 0; }();
    const dim3 g8(ceil_div(p.Tq, 256), p.H, p.B);
    // CD_ATTN_TIME=1: kernel time of every launch by events on its stream (printed; the probe build is never the product)
    static const bool timed = getenv("CD_ATTN_TIME") != nullptr;
    struct Timer {
      hipStream_t st; bool on; hipEvent_t e0, e1;
      Timer(hipStream_t s, bool o) : st(s), on(o) { if (on) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, st); } }
      ~Timer() {
        if (!on) return;
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("[attn d40 w8 dbg %d] %.1f us\n", dbg, ms * 1e3f);
        hipEventDestroy(e0); hipEventDestroy(e1);
      }
    } timer(st, timed);
#define CD_ATTN_DBG_CASE(V) case V: hipLaunchKernelGGL((k_attention<48, 64, 40, 2, false, 32, 8, V>), g8, dim3(512), 0, st, p); return;
    switch (dbg) {
      CD_ATTN_DBG_CASE(1) CD_ATTN_DBG_CASE(2) CD_ATTN_DBG_CASE(3) CD_ATTN_DBG_CASE(8) CD_ATTN_DBG_CASE(16) CD_ATTN_DBG_CASE(32)
      CD_ATTN_DBG_CASE(48) CD_ATTN_DBG_CASE(64) CD_ATTN_DBG_CASE(72) CD_ATTN_DBG_CASE(75) CD_ATTN_DBG_CASE(51)
      default: break;
    }
#undef CD_ATTN_DBG_CASE
#endif
    // round 5: the LDS-DMA / 16-row-block form (k_attention_d40); CD_ATTN_D40=0 selects the round-3 kernel for A/B runs
    static const bool d40_new = [] { const char* e = getenv("CD_ATTN_D40"); return !(e && e[0] == '0'); }();
    // what the kernel's 16-byte accesses and its ragged last key tile assume (round-5 advisor): V^T rows padded to whole
    // 64-key tiles (a tighter pad would let the last tile of one d row read the next row's data), q / k / V^T row strides
    // and base pointers on 16-byte boundaries; anything else takes the round-3 kernel
    const bool aligned16 = ((((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.vt | (uintptr_t)p.o) & 15) == 0) &&
                           (p.ldq % 8) == 0 && (p.ldk % 8) == 0 && (p.ldo % 8) == 0 && (p.q_bs % 8) == 0 && (p.k_bs % 8) == 0 &&
                           (p.o_bs % 8) == 0;
    if (d40_new && p.vt_dpad >= 40 && aligned16 && (p.vt_tpad % 8) == 0 && p.vt_tpad >= ceil_div(p.Tk, 64) * 64 &&
        (int64_t)p.Tk * p.ldk * 2 < (1ll << 31) && (int64_t)p.vt_dpad * p.vt_tpad * 2 < (1ll << 31))
      hipLaunchKernelGGL((k_attention_d40<8>), dim3(ceil_div(p.Tq, 256), p.H, p.B), dim3(512), 0, st, p);
    else
      hipLaunchKernelGGL((k_attention<48, 64, 40, 2, false, 32, 8>), dim3(ceil_div(p.Tq, 256), p.H, p.B), dim3(512), 0, st,
                         p);
  } else if (p.D == 40 && half_groups) CD_ATTN_KG(48, 64, 40, 2, 32);
  else if (p.D == 40) CD_ATTN(48, 64, 40, 2);
  else if (p.D == 80 && half_groups) CD_ATTN_KG(80, 96, 80, 2, 32);  // 640-channel level: three waves per SIMD
  else if (p.D == 80) CD_ATTN(80, 96, 80, 2);
  else if (p.D <= 32) CD_ATTN(32, 32, 0, 2);
  else if (p.D <= 48) CD_ATTN(48, 64, 0, 2);
  else if (p.D <= 64) CD_ATTN(64, 64, 0, 2);
  else if (p.D <= 80) CD_ATTN(80, 96, 0, 2);
  else if (p.D <= 96) CD_ATTN(96, 96, 0, 2);
  else if (p.D <= 128) CD_ATTN(128, 128, 0, 1);
  else CD_ATTN(160, 160, 0, 1);
#undef CD_ATTN
#undef CD_ATTN_KG
}

void launch_transpose_v(hipStream_t st, const bf16_t* v, int ldv, int64_t v_bs, bf16_t* vt, int B,
                        int H, int Tk, int D, int Dpad, int Tpad) {
  CD_CHECK(D % 8 == 0 && Tpad % 64 == 0, "transpose_v: D %% 8, Tpad %% 64");
  const size_t lds = (size_t)64 * (D + 2) * sizeof(bf16_t);
  hipLaunchKernelGGL(k_transpose_v, dim3(Tpad / 64, H, B), dim3(256), lds, st, v, ldv, v_bs, vt, H,
                     Tk, D, Dpad, Tpad);
}

}  // namespace cd
