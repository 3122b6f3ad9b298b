// Epilogue of the implicit-GEMM kernels (conv_gemm.hip, conv_pp.hip): accumulators -> LDS (fp32, per-wave region, CW
// columns at a time) -> fused elementwise -> 16-byte stores. Shared so that every kernel of the family produces the
// same bits from the same accumulators.
#pragma once
#include "common.h"
#include "kernels.h"

namespace cd {
namespace gemm_detail {

// T: tile configuration (TM, TN, MT, NT, CW, EPI_LD); acc: the wave's MT x NT accumulator blocks (32x32 MFMA layout:
// lane owns column lane & 31 and rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of a block); smem: workgroup LDS (the K-loop
// ring must be drained and all waves past a barrier); (m0, n0): tile origin; (wm, wn): wave position in the tile
template <class T>
__device__ __forceinline__ void conv_epilogue(const ConvGemmParams& p, f32x16 (&acc)[T::MT][T::NT], char* smem, int m0,
                                              int n0, int zb, int wave, int lane, int wm, int wn) {
  constexpr int MT = T::MT, NT = T::NT, TM = T::TM, TN = T::TN;
  const int frow = lane & 31, fhalf = lane >> 5;
  // ---- epilogue: accumulators -> LDS (fp32, per-wave region, CW columns at a time) -> fused elementwise -> 16-B
  // stores (8 lanes cover one 128-byte row segment). The staging region is private to the wave and a wave's LDS
  // accesses execute in program order, so the chunks need no barrier between them (wave_barrier only pins the
  // compiler's order). A register-resident variant (transposed MFMA blocks + v_permlane32_swap, no LDS round trip)
  // measured SLOWER: its row-per-lane 16-byte stores touch 32-64 lines per instruction (DESIGN.md optimisation log).
  constexpr int CW = T::CW, CJ = CW / 32;  // chunk width in columns / in 32-column MFMA blocks
  float* E = (float*)smem + wave * (TM * T::EPI_LD);
  const bool geglu = (p.act == ACT_GEGLU);
  char* outp = (char*)p.out + (int64_t)zb * p.o_bs * (p.out_f32 ? 4 : 2);
#pragma unroll
  for (int jc = 0; jc < NT; jc += CJ) {
    const int cj = (NT - jc) < CJ ? (NT - jc) : CJ;  // blocks in this chunk (compile-time after unrolling)
    const int cw = cj * 32;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < CJ; ++j)
        if (j < cj) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
            E[row * T::EPI_LD + j * 32 + frow] = acc[i][jc + j][r] * p.alpha;
          }
        }
    __builtin_amdgcn_wave_barrier();
    // GEGLU: packed columns come in blocks of 64 = [32 value | 32 gate] (k_pack_rows) = one chunk; only the value
    // half produces output, at column (n/64)*32 + n%32.
    const int vpr = geglu ? 4 : (cw / 8);  // 8-wide vectors per row handled
    const int rpp = 64 / vpr;              // rows per pass
    const int vr = lane / vpr, vc = lane % vpr;
    // column-only quantities are the same for every row this lane handles: hoist them out of the row loop
    const int col = vc * 8;                            // column inside the chunk
    const int n = n0 + wn * TN + jc * 32 + col;        // packed column
    const int nvalid = (p.N - n) < 8 ? (p.N - n) : 8;
    float bias_v[8], bias_g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bias_v[e] = (p.bias && e < nvalid) ? p.bias[n + e] : 0.0f;
      bias_g[e] = (geglu && p.bias) ? p.bias[n + 32 + e] : 0.0f;
    }
    // optional fused GroupNorm statistics: per-channel sum / sum-of-squares of the FINAL values over each
    // 32-row block of the output, written to stats[rowblock][2][N] (no atomics; the 32-row granularity is
    // independent of the tile configuration and of the batch size)
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    for (int r0 = 0; r0 < TM; r0 += rpp) {
      const int row = r0 + vr;
      const int m = m0 + wm * TM + row;
      if (row < TM && m < p.M && n < p.N) {
      float v[8];
      {
        const f32x4 lo = *(const f32x4*)(E + row * T::EPI_LD + col);
        const f32x4 hi = *(const f32x4*)(E + row * T::EPI_LD + col + 4);
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bias_v[e];
      int on = n;  // output column
      if (geglu) {
        float gt[8];
        const f32x4 lo = *(const f32x4*)(E + row * T::EPI_LD + col + 32);
        const f32x4 hi = *(const f32x4*)(E + row * T::EPI_LD + col + 36);
        gt[0] = lo[0]; gt[1] = lo[1]; gt[2] = lo[2]; gt[3] = lo[3];
        gt[4] = hi[0]; gt[5] = hi[1]; gt[6] = hi[2]; gt[7] = hi[3];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float gg = gt[e] + bias_g[e];
          v[e] = v[e] * gelu_fast(gg);
        }
        on = (n / 64) * 32 + (n % 64);
      } else {
        if (p.rowvec) {
          const int rvi = (p.rows_per_vec >= p.M) ? 0 : m / p.rows_per_vec;  // shared timestep: one vector
          const float* rv = p.rowvec + (int64_t)rvi * p.rowvec_ld + n;
#pragma unroll
          for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] += rv[e];
        }
        if (p.act == ACT_SILU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
        } else if (p.act == ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
        } else if (p.act == ACT_QGELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] / (1.0f + __expf(-1.702f * v[e]));
        }
      }
      if (p.resid) {
        const bf16_t* rp = p.resid + (int64_t)zb * p.o_bs + (int64_t)m * p.resid_ld + on;
        if (nvalid == 8 && ((p.resid_ld & 7) == 0)) {
          float rr[8];
          unpack8(*(const uint4*)rp, rr);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rr[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] += bf2f(rp[e]);
        }
      }
      if (p.out_f32) {
        float* op = (float*)outp + (int64_t)m * p.out_ld + on;
        if (nvalid == 8 && ((p.out_ld & 3) == 0)) {
          *(f32x4*)op = (f32x4){v[0], v[1], v[2], v[3]};
          *(f32x4*)(op + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (e < nvalid) op[e] = v[e];
        }
      } else {
        bf16_t* op = (bf16_t*)outp + (int64_t)m * p.out_ld + on;
        if (nvalid == 8 && ((p.out_ld & 7) == 0)) {
          *(uint4*)op = pack8(v);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (e < nvalid) op[e] = f2bf(v[e]);
        }
      }
      if (p.stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
      }
      }  // valid row
      if (p.stats && ((r0 + rpp) & 31) == 0) {
        // end of a 32-row block: fold the lanes that share this column vector (same vc, different vr)
        for (int o = vpr; o < 64; o <<= 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { ssum[e] += __shfl_xor(ssum[e], o); ssq[e] += __shfl_xor(ssq[e], o); }
        }
        const int rb = (m0 + wm * TM + r0 + rpp - 32) >> 5;
        if (vr == 0 && n < p.N && (rb << 5) < p.M) {
          float* sp = p.stats + ((int64_t)zb * ((p.M + 31) >> 5) + rb) * 2 * p.N + n;
#pragma unroll
          for (int e = 0; e < 8; ++e) if (e < nvalid) { sp[e] = ssum[e]; sp[p.N + e] = ssq[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
      }
    }
  }
}

}  // namespace gemm_detail
}  // namespace cd
