// Implicit-GEMM convolution / GEMM, "ping-pong" schedule: the 256 x 320 block tile of conv_gemm.hip (tile 20) with the
// two waves of every SIMD running half a phase apart, so that one of them always feeds the matrix pipe.
//
// conv_gemm.hip's K loop lets both waves of a SIMD read fragments, issue LDS-DMA and issue MFMAs all the time; its SQ
// counters on the big 3x3 convolutions show the matrix pipe 46 % busy with a third of the wave cycles parked at the one
// barrier per K step (profiles/r2_conv_gemm_sq_counters.txt). Here the K step is cut into phases separated by workgroup
// barriers (MI355X_MICROARCH.md "Two waves per SIMD", cdna_hip_programming.md T3-T5):
//   * waves 0-3 (group 0) and 4-7 (group 1) sit pairwise on the four SIMDs (waves go to SIMDs round-robin); group 1 is
//     one barrier behind group 0 for the whole loop. A wave alternates LOAD(slice): 7 fragment reads (ds_read_b128)
//     of one 16-deep k slice + its share of the LDS-DMA of a future tile, and MFMA(slice): ten v_mfma_f32_32x32x16 at
//     raised priority. While group 0 computes, group 1 loads, and vice versa: 4 barriers per 32-deep K step.
//   * 32-deep K steps, 4-deep LDS ring (36 KiB per stage): three tiles in flight; a wave waits (counted vmcnt) for its
//     own loads of tile k+1 at the end of its last LOAD phase of tile k, one barrier before anyone reads it.
//   * roles: group 0 stages the A operand (activations, implicit im2col addressing: 4 instructions per wave and tile),
//     group 1 the weights (5 per wave and tile).
// Same operand layouts, XOR-swizzled LDS rows, k-ascending accumulation and epilogue (conv_epilogue.h) as the other
// tiles: bit-identical results.
#include <mutex>

#include "common.h"
#include "kernels.h"
#include "conv_epilogue.h"

namespace cd {
namespace gemm_detail {

typedef const __attribute__((address_space(1))) void* pp_gptr_t;
typedef __attribute__((address_space(3))) void* pp_lptr_t;

struct PPCfg {
  static constexpr int BM = 256, BN = 320, BK = 32, WM = 4, WN = 2, NW = 8, NSTAGE = 4;
  static constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32, KS = BK / 16;
  static constexpr int CPR = BK / 8, RPI = 64 / CPR;  // 4 chunks of 16 B per LDS row, 16 rows per DMA instruction
  static constexpr int A_IPW = BM / RPI / 4, B_IPW = BN / RPI / 4;  // per wave of the staging group: 4 / 5
  static constexpr int CW = 64, EPI_LD = CW + 4;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int EPI_BYTES = NW * TM * EPI_LD * 4;
  static constexpr int LDS_BYTES = STAGE_BYTES * NSTAGE > EPI_BYTES ? STAGE_BYTES * NSTAGE : EPI_BYTES;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__global__ __launch_bounds__(512, 2) void k_conv_pp(ConvGemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using T = PPCfg;
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, MT = T::MT, NT = T::NT, KS = T::KS, TM = T::TM, TN = T::TN;
  constexpr int A_IPW = T::A_IPW, B_IPW = T::B_IPW, NSTAGE = T::NSTAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int grp = wave >> 2, wg = wave & 3;  // ping-pong group / position inside it

  // ---- block -> tile (XCD-aware, as conv_gemm.hip)
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const bool nmajor = (int64_t)p.N * p.Ktot > (int64_t)p.B * p.Hs * p.Ws * (p.C0 + p.C1);
  int tm, tn;
  if (nmajor) { tn = tile / tiles_m; tm = tile - tn * tiles_m; }
  else { tm = tile / tiles_n; tn = tile - tm * tiles_n; }
  const int m0 = tm * BM, n0 = tn * BN;
  const int zb = blockIdx.z;

  constexpr unsigned kRange = 0x7fffffffu, kInvalid = 0x80000000u;
  const bf16_t* base0 = p.src0 + (int64_t)zb * p.a_bs;
  const bf16_t* base1 = p.src1 ? p.src1 + (int64_t)zb * p.a_bs : base0;
  const __amdgpu_buffer_rsrc_t rsw =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.wgt + (int64_t)zb * p.w_bs), 0, kRange, 0x00020000);

  // ---- staging geometry of this wave's instructions (group 0: A rows, group 1: weight rows). One instruction covers
  // 16 LDS rows x 64 B; lane l fills row l / 4, physical chunk l & 3, from logical chunk (l & 3) ^ ((row >> 2) & 3)
  const int srow = lane >> 2, pchunk = lane & 3;
  int a_iy0[A_IPW], a_ix0[A_IPW], a_boff[A_IPW], a_lc8[A_IPW];
  unsigned a_voff[A_IPW];
  unsigned b_voff[B_IPW];
  const int HWo = p.Hout * p.Wout;
  const bool pow2 = ((HWo & (HWo - 1)) == 0) && ((p.Wout & (p.Wout - 1)) == 0);
  const int sh_hw = 31 - __builtin_clz(HWo), sh_w = 31 - __builtin_clz(p.Wout);
#pragma unroll
  for (int i = 0; i < A_IPW; ++i) {
    const int row = (wg * A_IPW + i) * T::RPI + srow;
    const int m = m0 + row;
    a_lc8[i] = (pchunk ^ ((row >> 2) & 3)) * 8;
    a_voff[i] = kInvalid;
    if (m < p.M) {
      int b, oy, ox;
      if (pow2) {
        b = m >> sh_hw;
        const int rem = m & (HWo - 1);
        oy = rem >> sh_w; ox = rem & (p.Wout - 1);
      } else {
        b = m / HWo;
        const int rem = m - b * HWo;
        oy = rem / p.Wout; ox = rem - oy * p.Wout;
      }
      a_iy0[i] = oy * p.stride - p.pad_t;
      a_ix0[i] = ox * p.stride - p.pad_l;
      a_boff[i] = b * p.Hs * p.Ws;
    } else {
      a_iy0[i] = -(1 << 28);
      a_ix0[i] = 0;
      a_boff[i] = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < B_IPW; ++i) {
    const int row = (wg * B_IPW + i) * T::RPI + srow;
    const int lchunk = pchunk ^ ((row >> 2) & 3);
    int nrow = n0 + row;
    if (nrow >= p.N) nrow = p.N - 1;  // duplicates a valid row, its outputs are masked
    b_voff[i] = (unsigned)((nrow * (p.ldw ? p.ldw : p.Ktot) + lchunk * 8) * 2);
  }

  const int Ctot = p.C0 + p.C1;
  const int nk = p.Ktot / BK;
  int kr = 0, kss = 0, kc = 0;  // K-step cursor: tap (kr, kss), channel offset kc in the concatenated channels

  char* As = smem;
  char* Bs = smem + NSTAGE * T::A_BYTES;

  bool st_live = false;
  int st_soffa = 0, st_soffb = 0;
  const bf16_t* st_base = base0;
  auto prepare = [&](int kt) {  // control flow of one tile's staging: cursor advance, per-tap offset refresh
    st_live = kt < nk;
    const int c_kc = kc, c_kr = kr, c_kss = kss;
    kc += BK;
    if (kc >= Ctot) {
      kc = 0;
      if (++kss >= p.KW) { kss = 0; ++kr; }
    }
    if (st_live && grp == 0 && (c_kc == 0 || c_kc == p.C0)) {  // new filter tap / second concat source
      const int ld = (c_kc < p.C0) ? p.ld0 : p.ld1;
#pragma unroll
      for (int i = 0; i < A_IPW; ++i) {
        int iy = a_iy0[i] + c_kr, ix = a_ix0[i] + c_kss;
        const bool ok = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);
        if (p.up) { iy >>= 1; ix >>= 1; }
        const int pix = a_boff[i] + iy * p.Ws + ix;
        a_voff[i] = ok ? (unsigned)((pix * ld + a_lc8[i]) * 2) : kInvalid;
      }
    }
    const bool first = c_kc < p.C0;
    st_base = first ? base0 : base1;
    st_soffa = (first ? c_kc : c_kc - p.C0) * 2;
    st_soffb = st_live ? kt * (BK * 2) : 0;
  };
  // this wave's loads of the prepared tile, in two halves (one per LOAD phase): group 0 -> A instructions 2h, 2h + 1;
  // group 1 -> weight instructions 0-2 / 3-4. Tiles past the end are issued with out-of-range offsets (zero fill into a
  // dead ring slot): every iteration has the same vmcnt footprint.
  auto issue_half = [&](int h, int buf) {
    if (grp == 0) {
      const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)st_base, 0, kRange, 0x00020000);
#pragma unroll
      for (int i = 0; i < A_IPW; ++i)
        if ((i >> 1) == h) {
          char* l = As + buf * T::A_BYTES + ((wg * A_IPW + i) * T::RPI) * (BK * 2);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (pp_lptr_t)l, 16, st_live ? a_voff[i] : kInvalid, st_soffa, 0, 0);
        }
    } else {
#pragma unroll
      for (int i = 0; i < B_IPW; ++i)
        if ((i < 3 ? 0 : 1) == h) {
          char* l = Bs + buf * T::B_BYTES + ((wg * B_IPW + i) * T::RPI) * (BK * 2);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (pp_lptr_t)l, 16, st_live ? b_voff[i] : kInvalid, st_soffb, 0, 0);
        }
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int frow = lane & 31, fhalf = lane >> 5;

  // ---- prologue: three tiles in flight, the first one landed and published
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s) {
    prepare(s);
    issue_half(0, s);
    issue_half(1, s);
  }
  // own loads of tile 0 have landed when at most the two younger tiles (4 / 5 loads each) are in flight
  if (grp == 0) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier\n\ts_barrier" ::: "memory");  // + the stagger: one phase behind

  int cur = 0, nxt = NSTAGE - 1;
  for (int kt = 0; kt < nk; ++kt) {
    prepare(kt + NSTAGE - 1);
    const char* Ab = As + cur * T::A_BYTES;
    const char* Bb = Bs + cur * T::B_BYTES;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // ---- LOAD(ks): half of this wave's staging loads for tile kt + 3 (first: the compiler drains lgkmcnt before it
      // rewrites M0 for an LDS-DMA, which would serialise the loads behind the fragment reads), then the slice's fragments
      issue_half(ks, nxt);
      bf16x8 af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * TM + i * 32 + frow;
        const int ch = (ks * 2 + fhalf) ^ ((row >> 2) & 3);
        af[i] = *(const bf16x8*)(Ab + row * (BK * 2) + ch * 16);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * TN + j * 32 + frow;
        const int ch = (ks * 2 + fhalf) ^ ((row >> 2) & 3);
        bfr[j] = *(const bf16x8*)(Bb + row * (BK * 2) + ch * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      // the reads are retired before the barrier (the other group may refill this slot right behind it); in the
      // tile's last LOAD phase this wave's loads of tile kt + 1 must also have landed (younger: tiles kt + 2, kt + 3)
      if (ks == KS - 1) {
        if (grp == 0) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- MFMA(ks): ten independent accumulators, at raised priority (the SIMD's other wave is in its LOAD phase)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = CD_MFMA_32x32x16(af[i], bfr[j], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    nxt = cur;
    cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
  }
  if (grp == 0) asm volatile("s_barrier" ::: "memory");  // re-align the groups
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the dead tail tiles
  __syncthreads();

  conv_epilogue<T>(p, acc, smem, m0, n0, zb, wave, lane, wm, wn);
#endif
}

}  // namespace gemm_detail

bool conv_pp_supports(const ConvGemmParams& p) {
  if (p.out == nullptr || p.splitk > 1 || p.act == ACT_GEGLU) return false;  // wave tile 160 wide: no GEGLU pairing
  if (p.C0 % 32 != 0 || p.C1 % 32 != 0 || p.Ktot % 32 != 0 || p.Ktot < 96) return false;
  if (p.N % 320 != 0 || p.M < 256) return false;
  return true;
}

void launch_conv_pp(hipStream_t st, const ConvGemmParams& p) {
  using namespace gemm_detail;
  CD_CHECK(conv_pp_supports(p), "conv_pp: unsupported problem (M %d N %d K %d)", p.M, p.N, p.Ktot);
  static std::once_flag attr_once;
  std::call_once(attr_once, [&]() {
    HIP_CHECK(hipFuncSetAttribute((const void*)k_conv_pp, hipFuncAttributeMaxDynamicSharedMemorySize, PPCfg::LDS_BYTES));
  });
  const int tiles = ceil_div(p.M, PPCfg::BM) * ceil_div(p.N, PPCfg::BN);
  hipLaunchKernelGGL(k_conv_pp, dim3(tiles, 1, p.nbatch), dim3(512), PPCfg::LDS_BYTES, st, p);
}

}  // namespace cd
