// Implicit-GEMM convolution / GEMM for gfx950: 16-bit MFMA 32x32x16 (fp16 or bf16), fp32 accumulate.
//
// One kernel family covers every contraction on the hot path:
// 3x3 s1 p1 conv, 3x3 s2 conv (U-Net Downsample p=1, openaimodel.py:134-160; VAE pad(0,1,0,1),
// model.py:72-76), 1x1 conv, nn.Linear, and batched Q.K^T / P.V for the single-head AttnBlock
// (model.py:178-202). Activations are NHWC 16-bit, so a K-slice of one filter tap is a contiguous
// run of channels; the skip-connection concat (openaimodel.py:736) is a second source pointer in
// the K loop and nearest-x2 upsampling (openaimodel.py:115) is folded into the gather address.
//
// Structure (CDNA4): 4 or 8 waves per workgroup; BMxBN tile, BK-deep K steps; both operands are
// staged HBM/L2 -> LDS with `buffer_load_dwordx4 ... offen lds` (no VGPR round trip) through buffer
// descriptors - padding taps and rows beyond M carry an out-of-range offset and arrive as zeros - into an
// NSTAGE-deep LDS ring; the wait for a tile is a COUNTED s_waitcnt vmcnt(N) (younger tiles stay in flight
// across the single workgroup barrier per K step); LDS rows are XOR-swizzled on the *source* side so
// ds_read_b128 fragment reads are conflict free; accumulators go through LDS in the epilogue so
// global stores are 16 B per lane with bias / time-embedding / residual / SiLU / GELU / GEGLU and the
// consumer's GroupNorm statistics fused. Deep-K layers with few output tiles run split-K (gridDim.y).
// Per-element reduction order is k-ascending for every tile configuration, so results do not depend on
// the configuration; a split factor > 1 sums the K ranges in range order (fixed for a given shape).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>

#include "common.h"
#include "kernels.h"

namespace cd {

namespace gemm_detail {

// cache-policy bits of the fast epilogue's output stores / residual loads (2 = nt: streaming). A/B builds only
// (build.py --ntst / --ntld): measured round 6, see docs/optimisation_log.md
#ifndef CD_EPI_STORE_AUX
#define CD_EPI_STORE_AUX 0
#endif
#ifndef CD_EPI_RESID_AUX
#define CD_EPI_RESID_AUX 0
#endif

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BM, int BN, int BK, int WM, int WN, int NSTAGE>
struct TileCfg {
  static constexpr int NW = WM * WN;             // waves per workgroup
  static constexpr int CPR = BK / 8;             // 16-byte chunks per LDS row
  static constexpr int RPI = 64 / CPR;           // rows covered by one wave-wide glds
  static constexpr int A_IPW = BM / RPI / NW;    // glds instructions per wave per K step (A)
  static constexpr int B_IPW = BN / RPI / NW;
  static constexpr int LPT = A_IPW + B_IPW;      // loads per wave per tile (vmcnt unit)
  static constexpr int TM = BM / WM, TN = BN / WN;  // wave tile
  static constexpr int MT = TM / 32, NT = TN / 32;
  static constexpr int KS = BK / 16;
  static constexpr int SWZ_SHIFT = (CPR == 8) ? 1 : 2;
  // the epilogue stages the wave tile through LDS in column chunks of CW (so the 64x160 / 64x128 wave tiles of the
  // 320- and 256-wide block tiles fit in 160 KiB)
  static constexpr int CW = TN >= 64 ? 64 : TN;  // chunk width in columns
  static constexpr int EPI_LD = CW + 4;          // fp32 row stride of the epilogue staging chunk
  static constexpr int PF = (MT * NT >= 8) ? 1 : 2;  // k-slices of fragments read ahead (register budget)
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int EPI_BYTES = NW * TM * EPI_LD * 4;
  static constexpr int LDS_BYTES = STAGE_BYTES * NSTAGE > EPI_BYTES ? STAGE_BYTES * NSTAGE : EPI_BYTES;
  // behind the ring: the tile's bias row [BN] and the time-embedding rows of its BM / 32 row blocks [BM / 32][BN] (fp32), filled
  // by the prologue's first DMA copies (1 KB per wave-wide copy) for the fast epilogue
  static constexpr int TAB_BIAS_I = (BN / 4 + 63) / 64;              // wave-wide copies: the bias row
  static constexpr int TAB_RV_I = ((BM / 32) * (BN / 4) + 63) / 64;  // ... the embedding rows
  // (256x64 with a 4-deep ring fills the 160 KB by itself: no tables, generic epilogue - the shipped table never picks it)
  static constexpr bool HAS_TAB = LDS_BYTES + (TAB_BIAS_I + TAB_RV_I) * 1024 <= 160 * 1024;
  static constexpr int TAB_BYTES = HAS_TAB ? (TAB_BIAS_I + TAB_RV_I) * 1024 : 0;
  static_assert(NW == 4 || NW == 8 || NW == 16, "4, 8 or 16 waves");
  static_assert(A_IPW >= 1 && B_IPW >= 1, "tile too small for the wave count");
  static_assert(A_IPW * RPI * NW == BM && B_IPW * RPI * NW == BN, "tile rows must split evenly over waves");
  static_assert(MT >= 1 && NT >= 1, "wave tile");
  static_assert(NSTAGE >= 2 && NSTAGE <= 8, "ring depth");
  static_assert((NSTAGE - 2) * LPT <= 63, "vmcnt field");
  static_assert(LDS_BYTES + TAB_BYTES <= 160 * 1024, "LDS");
};

template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier() {
  // this wave's loads of the current tile have landed (N younger ones may still fly) -> workgroup barrier.
  // One asm statement with a memory clobber: the compiler can neither move LDS accesses across it nor
  // add its own vmcnt(0) drain in front of the barrier.
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

#ifdef CD_PROBE
// Phase-timing instrumentation (probe build only): s_memtime stamps kept in SGPRs, sums formed on the scalar unit; the
// stamp sits where no LDS read is outstanding (top / bottom of a K step), so its lgkmcnt(0) costs nothing extra.
__device__ __forceinline__ unsigned long long probe_time() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
// the 100 MHz reference clock: the same origin on every CU (s_memtime counters are per-CU), calibrates ticks -> us
__device__ __forceinline__ unsigned long long probe_realtime() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define CD_PROBE_ONLY(...) __VA_ARGS__
#else
#define CD_PROBE_ONLY(...)
#endif

// A buffer descriptor for the epilogue. The words go through v_readfirstlane: behind the split-K fix-up's data-dependent return
// the compiler no longer treats them as wave-uniform and would wrap every buffer operation in a waterfall loop.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, int bytes) {
  const uint64_t u = (uint64_t)base;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// the straight-line epilogue (16-bit output, no activation, 16-byte aligned rows): see the kernel's epilogue
__device__ __forceinline__ bool fast_epilogue(const ConvGemmParams& p) {
  return p.act == ACT_NONE && !p.out_f32 && !p.resid_f32 && (p.N & 7) == 0 && (p.out_ld & 7) == 0 &&
         (((uintptr_t)p.out | (uintptr_t)p.resid | (uintptr_t)p.rowvec) & 15) == 0 && (p.o_bs & 7) == 0 &&
         (!p.resid || (p.resid_ld & 7) == 0) &&
         (!p.rowvec || ((p.rowvec_ld & 3) == 0 && (p.rows_per_vec >= p.M || (p.rows_per_vec & 31) == 0)));
}

// CHM: channel-major K order (compile-time: the tap-major instantiation carries none of its state)
template <int BM, int BN, int BK, int WM, int WN, int NSTAGE, bool CHM = false>
// (4-wave workgroups with 32-deep K steps are the two-per-CU configurations: 2 waves per SIMD, so at most 256 registers)
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4 && BK == 32 && NSTAGE == 3 && BM * BN >= 128 * 256) ? 2 : 1) void k_conv_gemm(ConvGemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // device-only builtins below: the host pass only needs the launch stub
  using T = TileCfg<BM, BN, BK, WM, WN, NSTAGE>;
  constexpr int CPR = T::CPR, RPI = T::RPI, A_IPW = T::A_IPW, B_IPW = T::B_IPW, NW = T::NW;
  constexpr int MT = T::MT, NT = T::NT, KS = T::KS, TM = T::TM, TN = T::TN, LPT = T::LPT;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // (launch_bounds below) two workgroups per CU: de-phase them once per launch, see ConvGemmParams::dephase_ticks
  if constexpr (WM * WN == 4 && BK == 32 && NSTAGE == 3 && BM * BN >= 128 * 256) {
    if (p.dephase_ticks > 0 && (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) < 2 * p.num_cus) {
      unsigned hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      if (tid == 0) *(volatile int*)smem = (int)(hwid & 1u);  // wave slot of wave 0: the second workgroup of a SIMD sits in slot 1
      __syncthreads();
      const int odd = *(volatile int*)smem;
      __syncthreads();
      if (odd) {
        unsigned long long t0, t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        do {
          __builtin_amdgcn_s_sleep(32);
          asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        } while ((long long)(t1 - t0) < (long long)p.dephase_ticks);
      }
    }
  }
  CD_PROBE_ONLY(
  const unsigned long long pr_rt0 = probe_realtime();
  unsigned long long pr_t0 = probe_time(), pr_prol = 0, pr_first = 0, pr_wait = 0, pr_comp = 0, pr_maxw = 0, pr_loop = 0,
                     pr_drain = 0, pr_stage = 0, pr_rows = 0, pr_issued = 0, pr_last = 0;
  unsigned* pr_log = (unsigned*)(smem + T::LDS_BYTES + T::TAB_BYTES) + wave * 64;  // (arrive, pass) of the first 32 K steps
  if (lane < 64) pr_log[lane] = 0;)

  // ---- block -> tile, XCD-aware (block b runs on XCD b%8; give each XCD a contiguous tile range
  // so neighbouring n-tiles of one m-tile share the A panel in that XCD's L2)
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  // Walk order inside an XCD's contiguous tile range: the operand with the larger footprint is the one an XCD
  // should own a slice of - its L2 then pulls 1/8 of it over the fabric and all of the smaller one. Weights
  // dominate in the 1280-channel 8x8 / 16x16 layers (29-59 MB of weights against 1-5 MB of activations): n-major
  // there, so the 8 XCDs do not each fetch the whole weight matrix (same speed, ~20 % less HBM-side traffic).
  const bool nmajor = (int64_t)p.N * p.Ktot > (int64_t)p.B * p.Hs * p.Ws * (p.C0 + p.C1);
  int tm, tn;
  if (p.tile_group > 0) {  // grouped walk: G row tiles at a time, m fastest inside the group
    const int width = p.tile_group * tiles_n;
    const int grp = tile / width, within = tile - grp * width;
    const int first_m = grp * p.tile_group;
    const int gsz = (tiles_m - first_m) < p.tile_group ? (tiles_m - first_m) : p.tile_group;
    tn = within / gsz; tm = first_m + (within - tn * gsz);
  } else if (nmajor) { tn = tile / tiles_m; tm = tile - tn * tiles_m; }
  else { tm = tile / tiles_n; tn = tile - tm * tiles_n; }
  const int m0 = tm * BM, n0 = tn * BN;
  const int zb = blockIdx.z;

  // ---- operand descriptors. Every staging load is `buffer_load_dwordx4 voff, rsrc, soff offen lds`:
  // the per-lane byte offset `voff` is constant within one filter tap (recomputed only when the tap or
  // the concat source changes), the K position inside the tap is the scalar `soff`, and rows that fall
  // into the zero padding (or beyond M) carry an out-of-range offset, for which the buffer unit
  // returns zeros into LDS. The K loop therefore issues no per-lane address arithmetic.
  constexpr unsigned kRange = 0x7fffffffu, kInvalid = 0x80000000u;
  // A buffer offset has 31 usable bits, an activation tensor may be larger (the KL-f8 decoder's 256-channel 512 x 512
  // level is 128 MiB per sample: 2 GiB at batch 16). The descriptor base therefore starts at the first SAMPLE this tile
  // touches and the lane offsets are relative to it: a tile's rows span a few samples at most.
  const int HWo = p.Hout * p.Wout;
  const bool pow2 = ((HWo & (HWo - 1)) == 0) && ((p.Wout & (p.Wout - 1)) == 0);
  const int sh_hw = 31 - __builtin_clz(HWo), sh_w = 31 - __builtin_clz(p.Wout);
  int b_first = pow2 ? (m0 >> sh_hw) : (m0 / HWo);
  CD_PROBE_ONLY(if (p.dbg & 4) b_first = 0;)
  const int64_t samp0 = (int64_t)b_first * p.Hs * p.Ws;
  const bf16_t* base0 = p.src0 + (int64_t)zb * p.a_bs + samp0 * p.ld0;
  const bf16_t* base1 = p.src1 ? p.src1 + (int64_t)zb * p.a_bs + samp0 * p.ld1 : base0;
  const __amdgpu_buffer_rsrc_t rsw =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.wgt + (int64_t)zb * p.w_bs), 0, kRange, 0x00020000);

  // ---- per-lane staging geometry
  const int srow = lane / CPR;   // row within a glds instruction
  const int pchunk = lane % CPR; // physical 16-B chunk this lane fills
  int a_iy0[A_IPW], a_ix0[A_IPW], a_boff[A_IPW], a_lc8[A_IPW];
  unsigned a_voff[A_IPW];
  // channel-major order (CHM): a_mk = validity of the KH tap rows (bits 0-7) and KW tap columns (bits 8-15) at this output
  // position; a_base = byte offset of tap (0, 0) in the current source - a step's offset is a_base + a scalar tap
  // displacement, or out of range where the masks say so (a_iy0 / a_ix0 / a_lc8 are then dead after the first K step)
  unsigned a_mk[A_IPW], a_base[A_IPW];
#pragma unroll
  for (int i = 0; i < A_IPW; ++i) {
    const int row = (i * NW + wave) * RPI + srow;
    int m = m0 + row;
    CD_PROBE_ONLY(if (p.dbg & 4) m = (m0 & 0x3ff) + row;)  // timing experiment: every tile gathers A from the first images (L2-hot)
    a_lc8[i] = (pchunk ^ ((row >> T::SWZ_SHIFT) & (CPR - 1))) * 8;
    a_voff[i] = kInvalid;
    a_mk[i] = 0; a_base[i] = 0;
    if (m < p.M) {
      int b, oy, ox;
      if (pow2) {  // every layer of the reference networks: shifts instead of ~100-instruction divisions
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
      a_boff[i] = (b - b_first) * p.Hs * p.Ws;
    } else {
      a_iy0[i] = -(1 << 28);  // always out of range -> zeros
      a_ix0[i] = 0;
      a_boff[i] = 0;
    }
    if constexpr (CHM) {
      unsigned mk = 0;
      for (int r = 0; r < p.KH; ++r) mk |= ((unsigned)(a_iy0[i] + r) < (unsigned)p.Hin ? 1u : 0u) << r;
      for (int q = 0; q < p.KW; ++q) mk |= ((unsigned)(a_ix0[i] + q) < (unsigned)p.Win ? 1u : 0u) << (8 + q);
      a_mk[i] = mk;
    }
  }
  unsigned b_voff[B_IPW];
#pragma unroll
  for (int i = 0; i < B_IPW; ++i) {
    const int row = (i * NW + wave) * RPI + srow;
    const int lchunk = pchunk ^ ((row >> T::SWZ_SHIFT) & (CPR - 1));
    int nrow = n0 + row;
    if (nrow >= p.N) nrow = p.N - 1;  // clamp: duplicates a valid row, its outputs are masked
    b_voff[i] = (unsigned)((nrow * (p.ldw ? p.ldw : p.Ktot) + lchunk * 8) * 2);
  }

  const int Ctot = p.C0 + p.C1;
  const int nk = p.Ktot / BK;
  // split-K: blockIdx.y owns the K steps [kt0, kt1); partial tiles meet in the fix-up before the epilogue
  const int nsplit = gridDim.y, sidx = blockIdx.y;
  const int kper = (nk + nsplit - 1) / nsplit;
  const int kt0 = sidx * kper;
  const int kt1 = (kt0 + kper < nk) ? kt0 + kper : nk;
  // K-step cursor (uniform): tap (kr, kss) and channel offset kc within the concatenated channels
  int kr, kss, kc;
  if constexpr (CHM) {  // step kt = (channel slice kt / ntaps, tap kt % ntaps)
    const int ntaps = p.KH * p.KW;
    const int slice = kt0 / ntaps, tap = kt0 - slice * ntaps;
    kc = slice * BK;
    kr = tap / p.KW;
    kss = tap - kr * p.KW;
  } else {
    const int k_el = kt0 * BK;
    const int tap = k_el / Ctot;
    kc = k_el - tap * Ctot;
    kr = tap / p.KW;
    kss = tap - kr * p.KW;
  }
  bool st_force = true;  // the first live tile of a split may start in the middle of a tap

  char* As = smem;                       // [NSTAGE][BM*BK*2]
  char* Bs = smem + NSTAGE * T::A_BYTES; // [NSTAGE][BN*BK*2]

  // ---- staging of one K tile, split in two so the loads can be spread over the MFMA segments:
  // prepare() does all control flow (cursor advance, per-tap offset refresh), issue(idx) emits load #idx.
  // Tiles past the end are issued with out-of-range offsets (zero fill into a dead ring slot), so every
  // iteration has the same vmcnt footprint and the loop body is branch-free.
  bool st_live = false;
  int st_soffa = 0, st_soffb = 0;
  const bf16_t* st_base = base0;
  auto prepare = [&](int kt) {
    st_live = kt < kt1;
    const int c_kc = kc, c_kr = kr, c_kss = kss;
    if constexpr (CHM) {
      // channel-major (no upsampling: the launcher keeps those tap-major): the tap moves every step, by a displacement that
      // is the same for every row; cursor advance in select form (if / else stores to captured state end up in scratch)
      const bool wk = c_kss + 1 >= p.KW;
      const bool wr = wk && (c_kr + 1 >= p.KH);
      kss = wk ? 0 : c_kss + 1;
      kr = wr ? 0 : c_kr + (wk ? 1 : 0);
      kc = c_kc + (wr ? BK : 0);
      if (st_live) {
        const int ld = (c_kc < p.C0) ? p.ld0 : p.ld1;
        if (st_force || ((c_kr | c_kss) == 0 && (c_kc == 0 || c_kc == p.C0))) {  // first step in this source
          st_force = false;
#pragma unroll
          for (int i = 0; i < A_IPW; ++i) {
            const int pix = a_boff[i] + a_iy0[i] * p.Ws + a_ix0[i];  // may lie before the sample where the masks say invalid
            a_base[i] = (unsigned)((pix * ld + a_lc8[i]) * 2);
          }
        }
        const unsigned dt = (unsigned)((c_kr * p.Ws + c_kss) * ld * 2);
#pragma unroll
        for (int i = 0; i < A_IPW; ++i) {
          const unsigned ok = (a_mk[i] >> c_kr) & (a_mk[i] >> (8 + c_kss)) & 1u;
          a_voff[i] = ok ? a_base[i] + dt : kInvalid;
        }
      }
    } else {
      kc += BK;
      if (kc >= Ctot) {
        kc = 0;
        if (++kss >= p.KW) { kss = 0; ++kr; }
      }
      if (st_live && (c_kc == 0 || c_kc == p.C0 || st_force)) {  // new filter tap / second concat source
        st_force = false;
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
    }
    const bool first = c_kc < p.C0;
    st_base = first ? base0 : base1;
    st_soffa = (first ? c_kc : c_kc - p.C0) * 2;
    st_soffb = st_live ? (CHM ? ((c_kr * p.KW + c_kss) * Ctot + c_kc) * 2 : kt * (BK * 2)) : 0;
  };
  auto issue = [&](int idx, int buf) {  // idx is a compile-time constant after unrolling
    if (idx < A_IPW) {
      const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)st_base, 0, kRange, 0x00020000);
      char* l = As + buf * T::A_BYTES + ((idx * NW + wave) * RPI) * (BK * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lptr_t)l, 16, st_live ? a_voff[idx < A_IPW ? idx : 0] : kInvalid,
                                               st_soffa, 0, 0);
    } else {
      const int i = idx - A_IPW;
      char* l = Bs + buf * T::B_BYTES + ((i * NW + wave) * RPI) * (BK * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr_t)l, 16, st_live ? b_voff[i < B_IPW ? i : 0] : kInvalid,
                                               st_soffb, 0, 0);
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int frow = lane & 31;  // fragment row within a 32-row MFMA tile
  const int fhalf = lane >> 5; // which 8-wide k half this lane feeds

  // ---- K loop. The ring holds NSTAGE tiles; fragments are read T::PF k-slices ahead of their MFMAs, ACROSS tile
  // boundaries. The one workgroup barrier per K step therefore does not sit at the step boundary (where every wave would
  // restart cold: refill issue, first fragment reads and their LDS latency with the matrix pipe idle - round 4 phase timing,
  // profiles/r4_conv_tile_phase_timing_*.txt: 35 % of a step) but in the middle of the segment that issues the first reads
  // of the NEXT tile, where each wave still holds MFMAs whose operands are in registers:
  //   segment ks = KS - PF of step kt:  first half of the slice's MFMAs
  //                                     s_waitcnt lgkmcnt(0) vmcnt(..) + s_barrier: every wave has finished READING tile kt
  //                                       (its last slice was requested one segment earlier) and its pieces of tile kt+1
  //                                       have landed -> slot of tile kt is free, tile kt+1 is readable by everyone
  //                                     refill of that slot with tile kt+NSTAGE, second half of the MFMAs, reads of
  //                                       slice 0 of tile kt+1
  // 2-deep rings issue the whole refill right there (a full step to land, as before); deeper rings spread it over the KS
  // segments that follow.
  constexpr int PF = T::PF, SYNC = KS - PF, NMF = MT * NT, HALF = NMF / 2;
  constexpr int SPREAD = (NSTAGE == 2) ? 1 : KS;
  static_assert(SYNC >= 0 && (NSTAGE - 1) * LPT <= 63, "fragment prefetch depth / vmcnt field");
#pragma unroll
  for (int s = 0; s < NSTAGE; ++s) {
    prepare(kt0 + s);
#pragma unroll
    for (int l = 0; l < LPT; ++l) issue(l, s);
  }
  // ---- bias / time-embedding tables of the fast epilogue (T::TAB_BYTES behind the ring), by DMA like the operands. Issued AFTER
  // the ring's first tiles so that their set-up (three more kernel-argument fetches, two descriptors) runs while tile kt0 is in
  // flight, and the SAME NUMBER of copies from every wave on every path, so the counted waits below stay exact: a wave whose
  // share runs past the last copy repeats an earlier one (same bytes), a launch without the fast epilogue copies through
  // empty descriptors (zeros). The K loop's drain (vmcnt(0) + barrier) publishes the tables to the epilogue.
  constexpr int TAB_N = T::TAB_BIAS_I + T::TAB_RV_I;                  // wave-wide 1 KB copies in all
  constexpr int TABI = T::HAS_TAB ? (TAB_N + NW - 1) / NW : 0;        // ... per wave
  if constexpr (TABI > 0) {
    const bool fe = fast_epilogue(p);
    const bool tb = (fe || p.act == ACT_GEGLU) && p.bias != nullptr;  // the GEGLU block epilogue takes its two bias vectors from the table too
    char* tab = smem + T::LDS_BYTES;
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(tb ? (const void*)p.bias : (const void*)p.wgt), 0, tb ? p.N * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((fe && p.rowvec) ? (const void*)p.rowvec : (const void*)p.wgt), 0, (fe && p.rowvec) ? kRange : 0, 0x00020000);
    const unsigned rpv = (unsigned)p.rows_per_vec;
    const bool rpv_pow2 = (rpv & (rpv - 1)) == 0;  // image sizes are powers of two on every reference network: a shift
    const int rpv_sh = 31 - __builtin_clz(rpv | 1u);
#pragma unroll
    for (int i = 0; i < TABI; ++i) {
      const int j = (i * NW + wave) % TAB_N;  // (uniform) copy number, bias copies first; past the last copy: an earlier one again
      if (j < T::TAB_BIAS_I) {
        const int c4 = j * 64 + lane;  // 16-byte chunk of the bias row (beyond BN: the next tile's columns or zeros, unused)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lptr_t)(tab + j * 1024), 16, (unsigned)((n0 + c4 * 4) * 4), 0, 0, 0);
      } else {
        const int c = (j - T::TAB_BIAS_I) * 64 + lane;  // chunk c = (row block, 16-byte chunk of its embedding row)
        const int blk = c / (BN / 4), c4 = c - blk * (BN / 4);
        const int m = m0 + blk * 32, nn = n0 + c4 * 4;
        int rvi = 0;
        if (p.rows_per_vec < p.M) rvi = rpv_pow2 ? (int)((unsigned)m >> rpv_sh) : m / p.rows_per_vec;
        const unsigned vo = (m < p.M && nn < p.N) ? (unsigned)((rvi * p.rowvec_ld + nn) * 4) : kInvalid;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsv, (lptr_t)(tab + j * 1024), 16, vo, 0, 0, 0);
      }
    }
  }
  CD_PROBE_ONLY(pr_prol = probe_time(); pr_last = pr_prol;)
  static_assert((NSTAGE - 1) * LPT + TABI <= 63, "vmcnt field");
  wait_vmcnt_barrier<(NSTAGE - 1) * LPT + TABI>();  // tile kt0 has landed everywhere (younger: the other tiles, the table copies)
  CD_PROBE_ONLY(pr_first = probe_time(); pr_last = pr_first;)

  bf16x8 af[KS][MT], bfr[KS][NT];
  auto read_frags = [&](int slot, int ks) {
    const char* Ab = As + slot * T::A_BYTES;
    const char* Bb = Bs + slot * T::B_BYTES;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = wm * TM + i * 32 + frow;
      const int ch = (ks * 2 + fhalf) ^ ((row >> T::SWZ_SHIFT) & (CPR - 1));
      af[ks][i] = *(const bf16x8*)(Ab + row * (BK * 2) + ch * 16);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int row = wn * TN + j * 32 + frow;
      const int ch = (ks * 2 + fhalf) ^ ((row >> T::SWZ_SHIFT) & (CPR - 1));
      bfr[ks][j] = *(const bf16x8*)(Bb + row * (BK * 2) + ch * 16);
    }
  };
#pragma unroll
  for (int q = 0; q < PF; ++q) read_frags(0, q);
  __builtin_amdgcn_sched_barrier(0);

  int cur = 0;   // ring slot of tile kt
  int fill = 0;  // slot the spread refill goes to (the tile whose sync came last)
  for (int kt = kt0; kt < kt1; ++kt) {
    const int nslot = (cur + 1 == NSTAGE) ? 0 : cur + 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int jrel = (ks - SYNC + KS) % KS;  // segments since the sync that opened the current refill
      auto mfma_range = [&](int lo, int hi) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < NMF; ++n)
          if (n >= lo && n < hi) acc[n / NT][n % NT] = CD_MFMA_32x32x16(af[ks][n / NT], bfr[ks][n % NT], acc[n / NT][n % NT]);
      };
      if (ks == SYNC) {
        mfma_range(0, HALF);
        CD_PROBE_ONLY(const unsigned long long pr_arr = probe_time();)
        // own LDS reads of tile kt done, own pieces of tile kt+1 landed ((NSTAGE-2) younger tiles may fly) -> barrier
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NSTAGE - 2) * LPT) : "memory");
        CD_PROBE_ONLY({
          const unsigned long long now = probe_time();
          pr_comp += pr_arr - pr_last; pr_wait += now - pr_arr; if (now - pr_arr > pr_maxw) pr_maxw = now - pr_arr;
          if (kt - kt0 < 32 && lane == 0) { pr_log[2 * (kt - kt0)] = (unsigned)(pr_arr - pr_t0); pr_log[2 * (kt - kt0) + 1] = (unsigned)(now - pr_t0); }
          pr_last = now;
        })
        prepare(kt + NSTAGE);
        fill = cur;
#pragma unroll
        for (int l = 0; l < LPT / SPREAD; ++l) issue(l, fill);
        mfma_range(HALF, NMF);
      } else {
        if (SPREAD > 1 && (ks > SYNC || kt != kt0)) {  // this segment's share of the refill opened at the last sync
#pragma unroll
          for (int l = (jrel * LPT) / SPREAD; l < ((jrel + 1) * LPT) / SPREAD; ++l) issue(l, fill);
        }
        mfma_range(0, NMF);
      }
      // fragments PF slices ahead: of this tile, or (after the sync) of the next one
      if (ks + PF < KS) read_frags(cur, ks + PF);
      else read_frags(nslot, ks + PF - KS);
      __builtin_amdgcn_sched_barrier(0);
    }
    cur = nslot;
  }
  CD_PROBE_ONLY(pr_loop = probe_time(); pr_comp += pr_loop - pr_last;)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the dummy tail loads before LDS is reused
  __syncthreads();  // all waves done with the ring before the epilogue reuses LDS
  CD_PROBE_ONLY(pr_drain = probe_time();)

  // ---- split-K fix-up: every split stores its fp32 partial tile (register layout, coalesced), the last
  // one to arrive sums all partials in split order (so the result does not depend on arrival order) and
  // runs the epilogue; the tile's arrival counter is reset for the next launch.
  if (nsplit > 1) {
    // Partial tiles travel with sc1 (device-coherent, write-through / L2-bypassing) 16-byte stores and
    // loads, so no release/acquire fence is needed around the counter: a fence here would be a full L2
    // write-back + invalidate per block, which costs more than the K loop it saves.
    constexpr int TILE_BYTES = BM * BN * 4;
    constexpr int kSc1 = 16;  // buffer cache-policy bit: sc1
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    float* part0 = p.sk_scratch + ((int64_t)zb * ntiles + tile) * nsplit * (BM * BN);
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc((void*)part0, 0, kRange, 0x00020000);
    // register layout: quad q of accumulator tile (i, j) of every lane is one 16-byte vector
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          const int off = sidx * TILE_BYTES + (((i * NT + j) * 4 + q) * (64 * NW) + tid) * 16;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsp, off, 0, kSc1);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's partial is out before it is counted
    __syncthreads();
    int* flag = p.sk_flags + zb * ntiles + tile;
    if (tid == 0) *(volatile int*)smem = __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int arrived = *(volatile int*)smem;
    __syncthreads();
    if (arrived != nsplit - 1) return;
    if (tid == 0) __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    for (int sp = 0; sp < nsplit; ++sp) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int off = sp * TILE_BYTES + (((i * NT + j) * 4 + q) * (64 * NW) + tid) * 16;
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsp, off, 0, kSc1));
            acc[i][j][4 * q] += v[0]; acc[i][j][4 * q + 1] += v[1];
            acc[i][j][4 * q + 2] += v[2]; acc[i][j][4 * q + 3] += v[3];
            // at most one 32 x 32 block of partials in flight: with 160 accumulators live the scheduler would otherwise batch
            // enough loads to spill what the epilogue needs later
            if (q == 3 && MT * NT >= 8) __builtin_amdgcn_sched_barrier(0);
          }
    }
  }

  // ---- epilogue: accumulators -> LDS (fp32, per-wave region, CW columns at a time) -> fused elementwise -> 16-B
  // stores (8 lanes cover one 128-byte row segment). The staging region is private to the wave and a wave's LDS
  // accesses execute in program order, so the chunks need no barrier between them (wave_barrier only pins the
  // compiler's order). A register-resident variant (transposed MFMA blocks + v_permlane32_swap, no LDS round trip)
  // measured SLOWER: its row-per-lane 16-byte stores touch 32-64 lines per instruction (DESIGN.md optimisation log).
  constexpr int CW = T::CW, CJ = CW / 32;  // chunk width in columns / in 32-column MFMA blocks
  // the epilogue's lane geometry starts from an opaque copy of the lane id, so that none of it is computed (and kept in
  // registers) ahead of the K loop or the split-K fix-up, where the tiles with 128+ accumulators have nothing to spare
  int lane_ep = lane;
  asm volatile("" : "+v"(lane_ep));
  float* E = (float*)smem + wave * (TM * T::EPI_LD);
  const bool geglu = (p.act == ACT_GEGLU);
  char* outp = (char*)p.out + (int64_t)zb * p.o_bs * (p.out_f32 ? 4 : 2);
  if (p.alpha != 1.0f) {  // (uniform) the attention-score GEMMs of the first stage; everything else leaves its sums as they are
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= p.alpha;
  }
  if (T::HAS_TAB && fast_epilogue(p)) {
    // ---- the common flavour (16-bit output, no activation: every ResBlock / transformer projection of the U-Nets and first
    // stages): one 32-row block at a time with all its LDS reads issued before any arithmetic; GroupNorm statistics of the
    // block from per-lane partial sums over its passes, transposed through the LDS rows just consumed, column sums by the lane
    // that owns the column -> two dense 256-byte stores (round 4). Round 6 (profiles/r6_conv_tile_phase_timing.txt: the row
    // passes are VALU-issue-bound at ~45 operations per 8-value vector, and a tile with a residual spends 6 us of its 37
    // waiting for it) made it STRAIGHT-LINE code whose only vector-memory operations are the residual loads and the stores:
    //   * all global traffic goes through buffer descriptors that start at the wave tile's first row and END AT ITS LAST VALID
    //     ROW (statistics: row block): a row beyond M or a column vector beyond N (offset bit 31) is dropped / read as zeros by
    //     the bounds check - no per-row compare, no exec-mask juggling, no 64-bit address arithmetic, one v_add_u32 per vector;
    //   * bias and time-embedding rows come from the LDS tables the prologue filled by DMA (T::TAB_BYTES);
    //   * the residual is fetched ONE 32-ROW BLOCK AHEAD: vector-memory operations return in order on gfx9, so a load issued
    //     behind the previous block's stores waits for their acknowledgements. With no branch between issue and use (the
    //     loads are issued even without a residual: an empty descriptor returns zeros without touching memory) the compiler's
    //     counted s_waitcnt vmcnt leaves the younger loads and the previous block's stores in flight.
    // (16-byte vector accesses: row strides AND base pointers must be multiples of 16 bytes - the batch strides are element
    // counts that keep the alignment when the leading dimensions do; odd shapes take the generic loop)
    constexpr int LD = T::EPI_LD, NRB = TM / 32;  // 32-row blocks per chunk
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4e;
    const int mw = m0 + wm * TM;  // first row of the wave tile
    const bool has_res = p.resid != nullptr;
    int rows_ok = p.M - mw;
    rows_ok = rows_ok < 0 ? 0 : (rows_ok > TM ? TM : rows_ok);
    const int nblk = (p.M + 31) >> 5;
    int blk_ok = nblk - (mw >> 5);
    blk_ok = blk_ok < 0 ? 0 : (blk_ok > NRB ? NRB : blk_ok);
    const __amdgpu_buffer_rsrc_t rs_o = uniform_rsrc(outp + (int64_t)mw * p.out_ld * 2, rows_ok * p.out_ld * 2);
    const __amdgpu_buffer_rsrc_t rs_r =
        uniform_rsrc(has_res ? (const void*)(p.resid + (int64_t)zb * p.o_bs + (int64_t)mw * p.resid_ld) : (const void*)outp,
                     has_res ? rows_ok * p.resid_ld * 2 : 0);
    const __amdgpu_buffer_rsrc_t rs_s =
        uniform_rsrc(p.stats ? (const void*)(p.stats + ((int64_t)zb * nblk + (mw >> 5)) * 2 * p.N) : (const void*)outp,
                     p.stats ? blk_ok * 2 * p.N * 4 : 0);
    const float* TB = (const float*)(smem + T::LDS_BYTES) + wn * TN;              // bias row of the tile
    const float* TR = (const float*)(smem + T::LDS_BYTES) + T::TAB_BIAS_I * 256 + (wm * NRB) * BN + wn * TN;  // embedding rows
    // residual vectors of 32-row block `rbx` of the chunk starting at MFMA column block `jcx` (the chunk's own lane geometry)
    constexpr bool PIPE = NW <= 8;  // (16-wave tiles live on 128 registers: they fetch the residual at the top of its own block)
    u32x4e rq[PIPE ? 2 : 1][4];
    auto resid_issue = [&](int jcx, int rbx, u32x4e (&dst)[4]) __attribute__((always_inline)) {
      const int cwx = ((NT - jcx) < CJ ? (NT - jcx) : CJ) * 32, vprx = cwx / 8, rppx = 64 / vprx, npx = 32 / rppx;
      const int nx = n0 + wn * TN + jcx * 32 + (lane_ep % vprx) * 8;
      const unsigned vo = nx < p.N ? (unsigned)(((rbx * 32 + lane_ep / vprx) * p.resid_ld + nx) * 2) : kInvalid;
#pragma unroll
      for (int ps = 0; ps < 4; ++ps)
        if (ps < npx) dst[ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, vo + (unsigned)(ps * rppx * p.resid_ld * 2), 0, CD_EPI_RESID_AUX);
    };
    if constexpr (PIPE) resid_issue(0, 0, rq[0]);
#pragma unroll
    for (int jc = 0; jc < NT; jc += CJ) {
      const int cj = (NT - jc) < CJ ? (NT - jc) : CJ;  // blocks in this chunk (compile-time after unrolling)
      const int cw = cj * 32;
      const int RPP = 64 / (cw / 8), NP = 32 / RPP;    // rows per pass, passes per 32-row block
      const int vr = lane_ep / (cw / 8), col = (lane_ep % (cw / 8)) * 8;  // row within a pass, column inside the chunk
      const int n = n0 + wn * TN + jc * 32 + col;
      CD_PROBE_ONLY(const unsigned long long pr_c0 = probe_time();)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < CJ; ++j)
          if (j < cj) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
              E[row * LD + j * 32 + frow] = acc[i][jc + j][r];
            }
          }
      __builtin_amdgcn_wave_barrier();
      CD_PROBE_ONLY(const unsigned long long pr_c1 = probe_time(); pr_stage += pr_c1 - pr_c0;)
      const f32x4 b0 = *(const f32x4*)(TB + jc * 32 + col), b1 = *(const f32x4*)(TB + jc * 32 + col + 4);
      const unsigned vo_out = n < p.N ? (unsigned)((vr * p.out_ld + n) * 2) : kInvalid;  // + the pass's row offset
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        const int kb = (jc / CJ) * NRB + rb;  // running block number: the residual registers alternate
        const int mb = mw + rb * 32;          // first row of the block
        if constexpr (!PIPE) resid_issue(jc, rb, rq[0]);  // the NEXT block's residual, ahead of this block's stores
        else if (rb + 1 < NRB) resid_issue(jc, rb + 1, rq[(kb + 1) & 1]);
        else if (jc + CJ < NT) resid_issue(jc + CJ, 0, rq[(kb + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 lo[4], hi[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
          if (ps < NP) {
            const float* er = E + (rb * 32 + ps * RPP + vr) * LD + col;
            lo[ps] = *(const f32x4*)er;
            hi[ps] = *(const f32x4*)(er + 4);
          }
        const f32x4 a0 = b0 + *(const f32x4*)(TR + rb * BN + jc * 32 + col);  // one embedding row per image; a block lies inside one
        const f32x4 a1 = b1 + *(const f32x4*)(TR + rb * BN + jc * 32 + col + 4);
        // (tiles of 8 waves keep the block's 4 x 8 values for the statistics, so that layers without statistics pay nothing for
        // them; the 16-wave tiles live on 128 registers and accumulate pass by pass)
        constexpr bool KEEP = NW <= 8;
        float v[KEEP ? 4 : 1][8];
        float ssum[8], ssq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
        const unsigned vo_blk = vo_out + (unsigned)(rb * 32 * p.out_ld * 2);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
          if (ps < NP) {
            float* vp = v[KEEP ? ps : 0];
#pragma unroll
            for (int e = 0; e < 4; ++e) { vp[e] = lo[ps][e] + a0[e]; vp[4 + e] = hi[ps][e] + a1[e]; }
            if (has_res) {
              float rf[8];
              const u32x4e w = rq[PIPE ? (kb & 1) : 0][ps];
              unpack8(make_uint4(w[0], w[1], w[2], w[3]), rf);
#pragma unroll
              for (int e = 0; e < 8; ++e) vp[e] += rf[e];
            }
            const uint4 pk = pack8(vp);
            CD_PROBE_ONLY(if (!(p.dbg & 1)))
            __builtin_amdgcn_raw_buffer_store_b128((u32x4e){pk.x, pk.y, pk.z, pk.w}, rs_o,
                                                   vo_blk + (unsigned)(ps * RPP * p.out_ld * 2), 0, CD_EPI_STORE_AUX);
            if constexpr (!KEEP) {
              const float on = (mb + ps * RPP + vr) < p.M ? 1.0f : 0.0f;
#pragma unroll
              for (int e = 0; e < 8; ++e) { const float x = on * vp[e]; ssum[e] += x; ssq[e] += x * x; }
            }
          }
        // fused GroupNorm statistics: per-channel sum / sum of squares of the FINAL values over each 32-row block of the output,
        // written to stats[rowblock][2][N] (no atomics; the 32-row granularity is independent of the tile configuration and of
        // the batch size)
        float s_sum = 0.f, s_sq = 0.f;
        if (p.stats) {
          if constexpr (KEEP) {
            if (mb + 32 <= p.M) {  // (uniform) every row of the block exists
#pragma unroll
              for (int ps = 0; ps < 4; ++ps)
                if (ps < NP) {
#pragma unroll
                  for (int e = 0; e < 8; ++e) { ssum[e] += v[ps][e]; ssq[e] += v[ps][e] * v[ps][e]; }
                }
            } else {
#pragma unroll
              for (int ps = 0; ps < 4; ++ps)
                if (ps < NP) {
                  const float on = (mb + ps * RPP + vr) < p.M ? 1.0f : 0.0f;
#pragma unroll
                  for (int e = 0; e < 8; ++e) { const float x = on * v[ps][e]; ssum[e] += x; ssq[e] += x * x; }
                }
            }
          }
          float* S = E + rb * 32 * LD;  // the block's rows are in registers: 2 x 512 floats of them as scratch
          __builtin_amdgcn_wave_barrier();
          *(f32x4*)(S + vr * cw + col) = (f32x4){ssum[0], ssum[1], ssum[2], ssum[3]};
          *(f32x4*)(S + vr * cw + col + 4) = (f32x4){ssum[4], ssum[5], ssum[6], ssum[7]};
          *(f32x4*)(S + 512 + vr * cw + col) = (f32x4){ssq[0], ssq[1], ssq[2], ssq[3]};
          *(f32x4*)(S + 512 + vr * cw + col + 4) = (f32x4){ssq[4], ssq[5], ssq[6], ssq[7]};
          __builtin_amdgcn_wave_barrier();
          if (cw == 64) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { s_sum += S[k * 64 + lane_ep]; s_sq += S[512 + k * 64 + lane_ep]; }
          } else {  // 32-column chunk: lanes 0-31 own the sums, 32-63 the sums of squares
            const int arr = lane_ep >> 5, c = lane_ep & 31;
#pragma unroll
            for (int k = 0; k < 16; ++k) s_sum += S[arr * 512 + k * 32 + c];
          }
          __builtin_amdgcn_wave_barrier();
        }
        {  // (issued without statistics too - an empty descriptor drops them - so that every path carries the same store count)
          const int nc0 = n0 + wn * TN + jc * 32;  // first column of the chunk
          if (cw == 64) {
            const unsigned so = nc0 + lane_ep < p.N ? (unsigned)((rb * 2 * p.N + nc0 + lane_ep) * 4) : kInvalid;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s_sum), rs_s, so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s_sq), rs_s, so + (unsigned)(p.N * 4), 0, 0);
          } else {
            const int arr = lane_ep >> 5, c = lane_ep & 31;
            const unsigned so = nc0 + c < p.N ? (unsigned)((rb * 2 * p.N + arr * p.N + nc0 + c) * 4) : kInvalid;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s_sum), rs_s, so, 0, 0);
          }
        }
      }
      CD_PROBE_ONLY(pr_rows += probe_time() - pr_c1;)
    }
  } else {
#pragma unroll
    for (int jc = 0; jc < NT; jc += CJ) {
      const int cj = (NT - jc) < CJ ? (NT - jc) : CJ;  // blocks in this chunk (compile-time after unrolling)
      const int cw = cj * 32;
      CD_PROBE_ONLY(const unsigned long long pr_c0 = probe_time();)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < CJ; ++j)
          if (j < cj) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
              E[row * T::EPI_LD + j * 32 + frow] = acc[i][jc + j][r];
            }
          }
      __builtin_amdgcn_wave_barrier();
      CD_PROBE_ONLY(const unsigned long long pr_c1 = probe_time(); pr_stage += pr_c1 - pr_c0;)
      // GEGLU: packed columns come in blocks of 64 = [32 value | 32 gate] (k_pack_rows) = one chunk; only the value
      // half produces output, at column (n/64)*32 + n%32.
      const int vpr = geglu ? 4 : (cw / 8);  // 8-wide vectors per row handled
      const int rpp = 64 / vpr;              // rows per pass
      int lane_g = lane_ep;  // (opaque again: none of this path's per-lane pointers is formed ahead of the staging writes)
      asm volatile("" : "+v"(lane_g));
      const int vr = lane_g / vpr, vc = lane_g % vpr;
      // column-only quantities are the same for every row this lane handles: hoist them out of the row loop
      const int col = vc * 8;                            // column inside the chunk
      const int n = n0 + wn * TN + jc * 32 + col;        // packed column
      const int nvalid = (p.N - n) < 8 ? (p.N - n) : 8;
      // GEGLU (attention.py:37-44: value * gelu(gate)), the feed-forward input projection of every transformer block: the block
      // form of the fast epilogue. A 64-column chunk = [32 value | 32 gate] columns; 4 lanes cover a row's 32 outputs, 16 rows
      // per pass; both bias vectors from the prologue's LDS table, stores through a row-bounded descriptor (no row branches).
      if (geglu && cw == 64 && !p.out_f32 && !p.resid && !p.stats && !p.rowvec && (p.N & 63) == 0 && (p.out_ld & 7) == 0 &&
          (((uintptr_t)p.out) & 15) == 0 && (p.o_bs & 7) == 0) {
        constexpr int LD = T::EPI_LD;
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4e;
        float bias_v[8], bias_g[8];
        if constexpr (T::HAS_TAB) {
          const float* tb = (const float*)(smem + T::LDS_BYTES) + wn * TN + jc * 32 + col;
          const f32x4 x0 = *(const f32x4*)tb, x1 = *(const f32x4*)(tb + 4), g0 = *(const f32x4*)(tb + 32), g1 = *(const f32x4*)(tb + 36);
#pragma unroll
          for (int e = 0; e < 4; ++e) { bias_v[e] = x0[e]; bias_v[4 + e] = x1[e]; bias_g[e] = g0[e]; bias_g[4 + e] = g1[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) { bias_v[e] = p.bias ? p.bias[n + e] : 0.0f; bias_g[e] = p.bias ? p.bias[n + 32 + e] : 0.0f; }
        }
        const int mw = m0 + wm * TM;
        int rows_ok = p.M - mw;
        rows_ok = rows_ok < 0 ? 0 : (rows_ok > TM ? TM : rows_ok);
        const __amdgpu_buffer_rsrc_t rs_o = uniform_rsrc(outp + (int64_t)mw * p.out_ld * 2, rows_ok * p.out_ld * 2);
        const unsigned vo_out = n < p.N ? (unsigned)((vr * p.out_ld + (n / 64) * 32 + (n % 64)) * 2) : kInvalid;
#pragma unroll
        for (int rb = 0; rb < TM / 32; ++rb) {
          f32x4 vl[2], vh[2], gl[2], gh[2];
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            const float* er = E + (rb * 32 + ps * 16 + vr) * LD + col;
            vl[ps] = *(const f32x4*)er;        vh[ps] = *(const f32x4*)(er + 4);
            gl[ps] = *(const f32x4*)(er + 32); gh[ps] = *(const f32x4*)(er + 36);
          }
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            float v[8], va[8], ga[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              va[e] = vl[ps][e] + bias_v[e]; va[4 + e] = vh[ps][e] + bias_v[4 + e];
              ga[e] = gl[ps][e] + bias_g[e]; ga[4 + e] = gh[ps][e] + bias_g[4 + e];
            }
            mul_gelu8(va, ga, v);
            const uint4 pk = pack8(v);
            CD_PROBE_ONLY(if (!(p.dbg & 1)))
            __builtin_amdgcn_raw_buffer_store_b128((u32x4e){pk.x, pk.y, pk.z, pk.w}, rs_o,
                                                   vo_out + (unsigned)((rb * 32 + ps * 16) * p.out_ld * 2), 0, 0);
          }
        }
        CD_PROBE_ONLY(pr_rows += probe_time() - pr_c1;)
        continue;
      }
      float bias_v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bias_v[e] = (p.bias && e < nvalid) ? p.bias[n + e] : 0.0f;
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
            const float gg = gt[e] + (p.bias ? p.bias[n + 32 + e] : 0.0f);  // (this loop serves odd shapes only)
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
            for (int e = 0; e < 8; ++e) {
              v[e] = gelu_fast(v[e]);
              // (16-wave tiles live on 128 registers: eight interleaved evaluations spill - two groups of four, the second
              // one's inputs defined by the empty statement that consumes the first one's results)
              if constexpr (NW == 16) {
                if (e == 3) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
              }
            }
          } else if (p.act == ACT_QGELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] / (1.0f + __expf(-1.702f * v[e]));
          }
        }
        if (p.resid && p.resid_f32) {
          const float* rp = (const float*)p.resid + (int64_t)zb * p.o_bs + (int64_t)m * p.resid_ld + on;
          if (nvalid == 8 && ((p.resid_ld & 3) == 0)) {
            const f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] += rp[e];
          }
        } else if (p.resid) {
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
        CD_PROBE_ONLY(if (!(p.dbg & 1)))
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
        CD_PROBE_ONLY(if (p.dbg & 1) { if (v[0] == 1.2345e-33f) ((float*)outp)[0] = v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7]; })
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
      CD_PROBE_ONLY(pr_rows += probe_time() - pr_c1;)
    }
  }
  CD_PROBE_ONLY({
    pr_issued = probe_time();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long pr_done = probe_time();
    if (p.probe) {
      unsigned hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned long long* o = p.probe + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * NW + wave) * kProbeWords;
      if (lane == 0) {
        o[0] = pr_t0; o[1] = pr_prol; o[2] = pr_first; o[3] = pr_wait; o[4] = pr_comp; o[5] = pr_maxw; o[6] = pr_loop;
        o[7] = pr_drain; o[8] = pr_stage; o[9] = pr_rows; o[10] = pr_issued; o[11] = pr_done; o[12] = kt1 - kt0;
        o[13] = ((unsigned long long)xcc << 32) | hwid; o[14] = pr_rt0; o[15] = probe_realtime();
      }
      ((unsigned*)(o + 16))[lane] = pr_log[lane];  // the log lies beyond the tile's LDS: the epilogue did not touch it
    }
  })
#endif  // __HIP_DEVICE_COMPILE__
}


template <int BM, int BN, int BK, int WM, int WN, int NSTAGE>
int launch_cfg(hipStream_t st, const ConvGemmParams& p) {
  using T = TileCfg<BM, BN, BK, WM, WN, NSTAGE>;
  const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
  const int split = p.splitk > 1 ? p.splitk : 1;
#ifdef CD_PROBE
  constexpr int kLds = T::LDS_BYTES + T::TAB_BYTES + T::NW * 256;  // + the per-wave stamp log
#else
  constexpr int kLds = T::LDS_BYTES + T::TAB_BYTES;
#endif
  // channel-major K order (ConvGemmParams::korder): 3 x 3 .. 8 x 8 filters without upsampling; everything else - 1 x 1, the
  // CLIP patch embeddings, the nearest-x2 convs - runs the tap-major instantiation
  // ... and only where it pays: the tap re-reads of a LARGE activation (>= 64 x 64 per sample, >= 32 MiB in all: the 64 x 64
  // level of the U-Nets from 8 samples up, the first stages) miss the 4 MiB L2 in tap-major order - there the order cuts the
  // fabric traffic by 45 % for 2 % of the conv's time (per-step offset update); on the 32 x 32 .. 8 x 8 levels the
  // activations stay L2 / Infinity-Cache resident either way and the update costs 2-7 % (profiles/r4_k_order_in_situ.txt)
  const int64_t a_bytes = (int64_t)p.B * p.Hs * p.Ws * (p.C0 + p.C1) * 2;
  const bool chm = p.korder != 0 && p.KH * p.KW > 1 && p.KH <= 8 && p.KW <= 8 && !p.up &&
                   (p.korder == 2 || (p.Hs * p.Ws >= 4096 && a_bytes >= (32ll << 20)));
  static PerDeviceOnce attr_once;  // engines on several host threads / devices launch the same instantiation
  auto kern0 = k_conv_gemm<BM, BN, BK, WM, WN, NSTAGE, false>;
  auto kern1 = k_conv_gemm<BM, BN, BK, WM, WN, NSTAGE, true>;
  attr_once([&]() {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern0, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    HIP_CHECK(hipFuncSetAttribute((const void*)kern1, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  });
  auto kern = chm ? kern1 : kern0;
  if (split > 1) {
    const SplitKWorkspace& ws = g_conv_splitk;
    CD_CHECK(p.sk_scratch && p.sk_flags, "conv_gemm: split-K without workspace");
    CD_CHECK((size_t)tiles * p.nbatch * split * BM * BN * 4 <= ws.scratch_bytes && tiles * p.nbatch <= ws.nflags,
             "conv_gemm: split-K workspace too small");
  }
  hipLaunchKernelGGL(kern, dim3(tiles, split, p.nbatch), dim3(64 * T::NW), kLds, st, p);
  return 0;
}

struct CfgInfo { int id, BM, BN, TN; const char* name; };
// tile configurations (id = ConvGemmParams::tile); every id exists for BK=64 and BK=32
const CfgInfo kCfgs[] = {
    {1, 128, 128, 64, "128x128 w2x2 s2"},
    {2, 128, 64, 64, "128x64 w4x1 s2"},
    {3, 64, 64, 32, "64x64 w2x2 s2"},
    {4, 128, 128, 64, "128x128 w2x2 s3"},
    {5, 256, 128, 64, "256x128 w4x2 s3"},
    {6, 128, 256, 64, "128x256 w2x4 s3"},
    {7, 128, 64, 64, "128x64 w4x1 s4"},
    {8, 64, 64, 32, "64x64 w2x2 s4"},
    {9, 256, 64, 64, "256x64 w8x1 s4"},
    {10, 128, 128, 64, "128x128 w2x2 s4"},
    {11, 64, 128, 64, "64x128 w2x2 s4"},
    {12, 128, 64, 64, "128x64 w4x1 s3"},
    {13, 256, 64, 64, "256x64 w8x1 s3"},
    {14, 128, 128, 64, "128x128 w4x2 s3"},
    {15, 64, 64, 32, "64x64 w2x2 s8"},
    {16, 128, 64, 64, "128x64 w4x1 s6"},
    {17, 64, 128, 64, "64x128 w2x2 s6"},
    {18, 256, 128, 32, "256x128 w4x4 s3"},
    {19, 256, 128, 64, "256x128 w8x2 s3"},
    // 320-wide block tiles: N = 320 / 640 / 1280 / 2560 of the SD / LDM U-Nets split with no padded columns and the
    // A panel is read once per 320 output channels; BK = 64 only (a 320-row B tile does not split over BK = 32 rows)
    {20, 256, 320, 160, "256x320 w4x2 s2"},
    {21, 128, 320, 160, "128x320 w2x2 s2"},
    {22, 256, 256, 128, "256x256 w4x2 s2"},
    {23, 128, 320, 160, "128x320 w4x2 s2"},
    // two workgroups per CU (4 waves, 74 KB of LDS each, 32-deep K steps, 3-deep ring): one's prologue / epilogue runs under
    // the other's K loop - for the short-K projections whose tiles spend a third of their time outside the K loop
    {24, 128, 256, 128, "128x256 w2x2 s3 bk32 x2/CU"},
    {25, 256, 128, 128, "256x128 w4x1 s3 bk32 x2/CU"},
};
inline bool cfg_needs_bk64(int id) { return id == 20 || id == 21 || id == 23; }
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

}  // namespace gemm_detail
thread_local SplitKWorkspace g_conv_splitk;
thread_local unsigned long long* g_conv_probe = nullptr;
namespace gemm_detail {

template <int BK>
int dispatch(hipStream_t st, const ConvGemmParams& p, int id) {
  switch (id) {
    case 1: return launch_cfg<128, 128, BK, 2, 2, 2>(st, p);
    case 2: return launch_cfg<128, 64, BK, 4, 1, 2>(st, p);
    case 3: return launch_cfg<64, 64, BK, 2, 2, 2>(st, p);
    case 4: return launch_cfg<128, 128, BK, 2, 2, 3>(st, p);
    case 5: return launch_cfg<256, 128, BK, 4, 2, 3>(st, p);
    case 6: return launch_cfg<128, 256, BK, 2, 4, 3>(st, p);
    case 7: return launch_cfg<128, 64, BK, 4, 1, 4>(st, p);
    case 8: return launch_cfg<64, 64, BK, 2, 2, 4>(st, p);
    case 9:
      if constexpr (BK == 64) return launch_cfg<256, 64, 64, 8, 1, 4>(st, p);
      else return launch_cfg<128, 64, 32, 4, 1, 4>(st, p);  // 8x1 waves need >= 8 glds rows groups per B tile
     
    case 10: return launch_cfg<128, 128, BK, 2, 2, 4>(st, p);
    case 11: return launch_cfg<64, 128, BK, 2, 2, 4>(st, p);
    case 12: return launch_cfg<128, 64, BK, 4, 1, 3>(st, p);
    case 13:
      if constexpr (BK == 64) return launch_cfg<256, 64, 64, 8, 1, 3>(st, p);
      else return launch_cfg<128, 64, 32, 4, 1, 3>(st, p);
     
    case 14: return launch_cfg<128, 128, BK, 4, 2, 3>(st, p);
    case 15: return launch_cfg<64, 64, BK, 2, 2, 8>(st, p);
    case 16: return launch_cfg<128, 64, BK, 4, 1, 6>(st, p);
    case 17: return launch_cfg<64, 128, BK, 2, 2, 6>(st, p);
    case 18:
      if constexpr (BK == 64) return launch_cfg<256, 128, 64, 4, 4, 3>(st, p);
      else return launch_cfg<256, 128, 32, 4, 2, 3>(st, p);  // 16 waves need >= 16 row groups per B tile
     
    case 19:
      if constexpr (BK == 64) return launch_cfg<256, 128, 64, 8, 2, 3>(st, p);
      else return launch_cfg<256, 128, 32, 4, 2, 3>(st, p);
     
    case 20: case 21: case 22: case 23:
      if constexpr (BK == 64) {
        if (id == 20) return launch_cfg<256, 320, 64, 4, 2, 2>(st, p);
        else if (id == 21) return launch_cfg<128, 320, 64, 2, 2, 2>(st, p);
        else if (id == 22) return launch_cfg<256, 256, 64, 4, 2, 2>(st, p);
        else return launch_cfg<128, 320, 64, 4, 2, 2>(st, p);
      } else {
        // 256x256 also exists with 32-deep K steps and a 4-deep ring (3 stages in flight instead of 1: the short-K
        // projections wait on Infinity-Cache latency, not on bandwidth); the 320-wide B tiles do not split over 32 rows
        if (id == 22) return launch_cfg<256, 256, 32, 4, 2, 4>(st, p);
        CD_CHECK(false, "conv_gemm: tile configuration %d needs channel counts that are multiples of 64", id);
      }
      return 0;
    case 24: return launch_cfg<128, 256, 32, 2, 2, 3>(st, p);
    case 25: return launch_cfg<256, 128, 32, 4, 1, 3>(st, p);
    default: CD_CHECK(false, "conv_gemm: unknown tile configuration %d", id);
  }
  return 0;
}

thread_local const char* g_last_cfg = "";

// CU count of the current device (256 on an unpartitioned MI355X; a partitioned part has fewer): "does this configuration
// fill the chip" is asked against it, not against a literal
int device_cu_count() {
  static std::atomic<int> ncu_of[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int n = ncu_of[dev & 63].load(std::memory_order_relaxed);
  if (!n) {
    hipDeviceProp_t prop;
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    ncu_of[dev & 63].store(n, std::memory_order_relaxed);
  }
  return n;
}

// shapes the tile table does not know (and tuning is off): the largest of three tiles that still fills the chip - "fills" is
// asked against the device's CU count (256 on an unpartitioned MI355X), not a literal
int pick_config(const ConvGemmParams& p) {
  const int64_t ncu = device_cu_count();
  const int64_t t128 = (int64_t)ceil_div(p.M, 128) * ceil_div(p.N, 128) * p.nbatch;
  const int64_t t12864 = (int64_t)ceil_div(p.M, 128) * ceil_div(p.N, 64) * p.nbatch;
  if (p.act == ACT_GEGLU) return (2 * t128 >= 3 * ncu) ? 1 : 2;  // GEGLU needs a wave tile >= 64 columns wide
  if (2 * t128 >= 3 * ncu && p.N % 128 == 0) return 1;
  if (t128 >= 2 * ncu) return 1;
  if (t12864 >= ncu) return 2;
  return 3;
}

}  // namespace gemm_detail
using namespace gemm_detail;

const char* conv_gemm_last_config() { return g_last_cfg; }

thread_local KernelProfiler* g_conv_prof = nullptr;

void KernelProfiler::next_pair(hipEvent_t* a, hipEvent_t* b, double fl, const std::string& what) {
  labels.push_back(what);
  if (used + 2 > (int)events.size()) {
    const size_t old = events.size();
    events.resize(old + 1024);
    for (size_t i = old; i < events.size(); ++i) HIP_CHECK(hipEventCreate(&events[i]));
  }
  *a = events[used]; *b = events[used + 1];
  used += 2;
  flops.push_back(fl);
}
void KernelProfiler::collect(int* launches, double* total_ms, double* total_flops) {
  double ms = 0, fl = 0;
  for (int i = 0; i + 1 < used; i += 2) {
    HIP_CHECK(hipEventSynchronize(events[i + 1]));
    float t = 0;
    HIP_CHECK(hipEventElapsedTime(&t, events[i], events[i + 1]));
    ms += t; fl += flops[i / 2];
    if (verbose) {
      auto& e = per_shape[labels[i / 2]];
      e.n += 1; e.ms += t; e.flops += flops[i / 2];
    }
  }
  if (verbose) {
    std::vector<std::pair<std::string, Entry>> v(per_shape.begin(), per_shape.end());
    std::sort(v.begin(), v.end(), [](const auto& x, const auto& y) { return x.second.ms > y.second.ms; });
    fprintf(stderr, "[conv_gemm] %d launches %.3f ms %.1f TFLOP/s\n", used / 2, ms, fl / ms * 1e-9);
    for (auto& kv : v)
      fprintf(stderr, "  %-58s n=%4d %8.3f ms %7.1f us/launch %7.1f TF/s\n", kv.first.c_str(), kv.second.n,
              kv.second.ms, kv.second.ms / kv.second.n * 1e3, kv.second.flops / kv.second.ms * 1e-9);
    per_shape.clear();
  }
  *launches = used / 2; *total_ms = ms; *total_flops = fl;
  used = 0; flops.clear(); labels.clear();
}
KernelProfiler::~KernelProfiler() { for (auto e : events) (void)hipEventDestroy(e); }

// ---- online autotuner: the engine replays the same few dozen contraction shapes hundreds of times, so
// the first time a shape is seen every tile configuration is timed on it (HIP events, output redirected
// to a scratch buffer so in-place residual updates are not disturbed) and the fastest is remembered.
// Every tile configuration accumulates each output element in the same k order, so the choice of TILE never changes a
// result bit. A split-K factor does (it sums per-range fp32 partials): split factors therefore come only from the
// shipped / cached table (tune_gfx950.txt, CYCLEDIFF_TUNE_CACHE), identical for every process and rank; online tuning
// of an unseen shape tries split = 1 only, unless CYCLEDIFF_TUNE_SPLITK=1 (the mode the shipped table is made in).
ConvTuner g_conv_tuner;

namespace {
struct ShapeKey {
  int v[14];
  bool operator<(const ShapeKey& o) const { return memcmp(v, o.v, sizeof(v)) < 0; }
};
// CYCLEDIFF_TUNE_CACHE=<file>: choices are appended to / preloaded from a text file (15 ints per line), so a
// second process (e.g. a rocprofv3 run whose kernel statistics should not contain tuning launches) starts tuned.
const char* tune_cache_path() {
  static const char* p = getenv("CYCLEDIFF_TUNE_CACHE");
  return (p && p[0]) ? p : nullptr;
}
std::map<ShapeKey, int>& tune_table() {
  static std::map<ShapeKey, int> t;
  static bool loaded = false;
  if (!loaded) {
    loaded = true;
    // CYCLEDIFF_TUNE_DEFAULT: the table shipped with the package (tune_gfx950.txt, set by _ffi.load_library) - the
    // shapes of the reference networks at the BASELINE batch sizes start with fixed choices, so results and
    // timings do not depend on one-off timing noise; CYCLEDIFF_TUNE_CACHE (read + append) overrides / extends it
    const char* paths[2] = {getenv("CYCLEDIFF_TUNE_DEFAULT"), tune_cache_path()};
    for (const char* path : paths) {
      if (!path || !path[0]) continue;
      if (FILE* f = fopen(path, "r")) {
        ShapeKey k; int val;
        for (;;) {
          int got = 0;
          for (int i = 0; i < 14; ++i) got += fscanf(f, "%d", &k.v[i]);
          got += fscanf(f, "%d", &val);
          if (got != 15) break;
          t[k] = val;
        }
        fclose(f);
      }
    }
  }
  return t;
}
void tune_cache_append(const ShapeKey& k, int val) {
  const char* path = tune_cache_path();
  if (!path) return;
  if (FILE* f = fopen(path, "a")) {
    for (int i = 0; i < 14; ++i) fprintf(f, "%d ", k.v[i]);
    fprintf(f, "%d\n", val);
    fclose(f);
  }
}

std::mutex g_tune_mu;  // engines on different host threads share one table; tuning itself is serialised

int tuned_config(hipStream_t st, const ConvGemmParams& p, bool k64) {
  std::lock_guard<std::mutex> lock(g_tune_mu);
  ConvTuner& tu = g_conv_tuner;
  if (!tu.enabled || !tu.scratch) return pick_config(p);
  const int nout = (p.act == ACT_GEGLU) ? p.N / 2 : p.N;
  const size_t need = (size_t)p.M * nout * (p.out_f32 ? 4 : 2) * p.nbatch;
  if (need > tu.scratch_bytes) return pick_config(p);
  ShapeKey key = {{p.M, p.N, p.Ktot, p.KH, p.C0, p.C1, p.stride, p.up, p.act, p.nbatch, p.out_f32, p.Hout, p.Wout,
                   (p.resid ? 1 : 0) | (p.rowvec ? 2 : 0)}};
  auto& tab = tune_table();
  auto it = tab.find(key);
  if (it != tab.end()) return it->second;
  ConvGemmParams q = p;
  q.out = tu.scratch; q.out_ld = nout; q.o_bs = (int64_t)p.M * nout;
  if (p.resid && p.nbatch > 1) q.resid = nullptr;  // o_bs also strides the residual
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
  int best = pick_config(p);
  float best_ms = 1e30f;
  // split-K candidates only where 128x128 tiles cannot fill the 256 CUs (deep-K 3x3 convs at 8x8 / 16x16)
  const SplitKWorkspace& sk = g_conv_splitk;
  const int64_t t128 = (int64_t)ceil_div(p.M, 128) * ceil_div(p.N, 128) * p.nbatch;
  const int nk = p.Ktot / (k64 ? 64 : 32);
  static const int kSplits[] = {1, 2, 3, 4, 6, 8, 12, 16};
  static const bool tune_splitk = [] { const char* e = getenv("CYCLEDIFF_TUNE_SPLITK"); return e && e[0] == '1'; }();
  // short-K contractions (1x1 convs / linears) also try the BK=32 variants: half the LDS per stage, so
  // twice the resident blocks to hide the prologue / epilogue of a 5-10 step K loop
  const int nbk = (k64 && p.Ktot <= 1280) ? 2 : 1;
  for (int bi = 0; bi < nbk; ++bi)
  for (int i = 0; i < kNumCfgs; ++i)
  for (int si = 0; si < 8; ++si) {
    const CfgInfo& c = kCfgs[i];
    const int split = kSplits[si];
    const bool use64 = k64 && bi == 0;
    if (bi == 1 && split > 1) continue;
    if (split > 1 && !tune_splitk) continue;
    if (p.act == ACT_GEGLU && (c.TN % 64) != 0) continue;
    if (c.BM >= 256 && p.M < 256) continue;
    if (cfg_needs_bk64(c.id) && !use64) continue;
    if ((c.id == 24 || c.id == 25) && bi == 1) continue;  // one instantiation (32-deep K steps either way)
    if (c.BN == 320 && (p.N % 320) != 0) continue;
    const int64_t tiles = (int64_t)ceil_div(p.M, c.BM) * ceil_div(p.N, c.BN) * p.nbatch;
    if (split > 1) {
      // a split only where THIS tile configuration leaves CUs idle (round 4: the wide tiles on the 16 x 16 level have 128
      // tiles at B' = 32; their K loop is the fastest, two K ranges each fill the chip)
      if (!sk.scratch || tiles >= device_cu_count() || nk < 4 * split || tiles * split > 2048) continue;
      if ((size_t)tiles * split * c.BM * c.BN * 4 > sk.scratch_bytes || tiles > sk.nflags) continue;
    }
    q.splitk = split; q.sk_scratch = sk.scratch; q.sk_flags = sk.flags;
    auto run = [&]() { if (use64) dispatch<64>(st, q, c.id); else dispatch<32>(st, q, c.id); };
    auto timed = [&](int reps) {
      HIP_CHECK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) run();
      HIP_CHECK(hipEventRecord(e1, st));
      HIP_CHECK(hipEventSynchronize(e1));
      float t = 0;
      HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
      return t / reps;
    };
    run();  // warm
    float ms = timed(2);
    // short launches: event granularity and launch gaps dominate a 2-launch sample
    const int reps = (int)fminf(24.f, fmaxf(2.f, 0.4f / fmaxf(ms, 1e-3f)));
    ms = fminf(ms, fminf(timed(reps), timed(reps)));
    if (ms < best_ms) { best_ms = ms; best = c.id | (split << 8) | ((k64 && !use64) ? (1 << 16) : 0); }
  }
  if (lin_stream_supports(q)) {  // the streaming schedule for the 320-channel linears (lin_stream.hip): same bits
    auto timed = [&](int reps) {
      HIP_CHECK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) launch_lin_stream(st, q);
      HIP_CHECK(hipEventRecord(e1, st));
      HIP_CHECK(hipEventSynchronize(e1));
      float t = 0;
      HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
      return t / reps;
    };
    q.splitk = 0;
    launch_lin_stream(st, q);
    float ms = timed(2);
    const int reps = (int)fminf(24.f, fmaxf(2.f, 0.4f / fmaxf(ms, 1e-3f)));
    ms = fminf(ms, fminf(timed(reps), timed(reps)));
    if (ms < best_ms) { best_ms = ms; best = kLinStreamTile; }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  tab[key] = best;
  tune_cache_append(key, best);
  ++tu.shapes_tuned;
  return best;
}
}  // namespace

void launch_conv_gemm(hipStream_t st, const ConvGemmParams& p) {
  const int Ctot = p.C0 + p.C1;
  CD_CHECK(p.C0 % 32 == 0 && p.C1 % 32 == 0, "conv_gemm: channels must be multiples of 32 (C0=%d C1=%d)", p.C0, p.C1);
  CD_CHECK(p.Ktot == p.KH * p.KW * Ctot, "conv_gemm: Ktot mismatch");
  CD_CHECK(p.zeros != nullptr, "conv_gemm: zero page missing");
  CD_CHECK(p.M > 0 && p.N > 0, "conv_gemm: empty problem");
  CD_CHECK(((uintptr_t)p.src0 & 15) == 0 && ((uintptr_t)p.wgt & 15) == 0, "conv_gemm: 16-B alignment");
  CD_CHECK((p.ld0 % 8) == 0 && (p.src1 == nullptr || (p.ld1 % 8) == 0), "conv_gemm: ld must be a multiple of 8");
  if (p.act == ACT_GEGLU) CD_CHECK(p.N % 64 == 0, "GEGLU needs packed N %% 64 == 0");
  bool k64 = (p.C0 % 64 == 0) && (p.C1 % 64 == 0);
  if (p.ln_fold) CD_CHECK(p.tile == kLinStreamTile, "conv_gemm: a LayerNorm-folded layer runs on lin_stream only");
  int id = p.tile ? p.tile : tuned_config(st, p, k64);
  if (id & (1 << 16)) k64 = false;  // tuner (or an explicit tile | 1<<16) asks for the BK=32 variant
  ConvGemmParams pk = p;
  id &= 0xffff;
  if ((id >> 8) > 1) pk.splitk = id >> 8;  // from the tuner, or packed into an explicit `tile` (tests, sweeps)
  id &= 0xff;
  if (pk.splitk > 1 && !pk.sk_scratch) { pk.sk_scratch = g_conv_splitk.scratch; pk.sk_flags = g_conv_splitk.flags; }
  pk.probe = g_conv_probe;
  static const int korder_env = [] { const char* e = getenv("CYCLEDIFF_KORDER"); return e ? atoi(e) : -1; }();
  if (korder_env >= 0) pk.korder = korder_env;
  // grouped tile walk for wide-N contractions (>= CYCLEDIFF_TILE_GROUP_MIN_N column tiles of 256): both operands exceed an
  // XCD's 4 MiB L2 and a row-major walk re-streams one of them for every tile row / column (round-6 per-shape counters:
  // GEGLU 1280 -> 10240 at 16 x 16 reads 26x its algorithmic bytes, 640 -> 5120 at 32 x 32 14x)
  // Measured (profiles/r6_conv_gemm_traffic_by_shape_b64_group8.json, r6_tile_group_ab.json): groups of 8 cut those two layers'
  // fetch to 9x / 7x and the forward's GEMM reads from 2.31x to 2.03x of algorithmic at unchanged speed (the layers are bound by
  // their GEGLU epilogues, not by the fabric) - on by default for N >= 2048; CYCLEDIFF_TILE_GROUP=0 restores the row-major walk
  static const int tg_env = [] { const char* e = getenv("CYCLEDIFF_TILE_GROUP"); return e ? atoi(e) : 8; }();
  static const int tg_min_n = [] { const char* e = getenv("CYCLEDIFF_TILE_GROUP_MIN_N"); return e ? atoi(e) : 2048; }();
  if (tg_env > 0 && p.N >= tg_min_n && p.nbatch == 1) pk.tile_group = tg_env;
#ifdef CD_PROBE
  if (const char* e = getenv("CYCLEDIFF_PROBE_DBG")) pk.dbg = atoi(e);
  static const int dephase_env = [] { const char* e = getenv("CYCLEDIFF_DEPHASE_TICKS"); return e ? atoi(e) : 0; }();
  pk.dephase_ticks = dephase_env;
  pk.num_cus = device_cu_count();
  if (pk.dbg & 2) pk.stats = nullptr;
#endif
  static const CfgInfo kLinStreamCfg = {kLinStreamTile, 256, 64, 64, "lin_stream 256 x N, K = 320"};
  const CfgInfo* ci = nullptr;
  if (id == kLinStreamTile) {
    // a table entry made where the shape qualified (e.g. with 16-bit output) may meet a call that does not: fall back
    if (lin_stream_supports(pk)) ci = &kLinStreamCfg;
    else { CD_CHECK(!p.tile, "conv_gemm: lin_stream does not support this problem"); id = pick_config(p); }
  }
  for (int i = 0; i < kNumCfgs && !ci; ++i) if (kCfgs[i].id == id) ci = &kCfgs[i];
  CD_CHECK(ci, "conv_gemm: unknown tile configuration %d", id);
  if (p.act == ACT_GEGLU && (ci->TN % 64) != 0) { id = 2; ci = &kCfgs[1]; }
  if (cfg_needs_bk64(id)) CD_CHECK(k64, "conv_gemm: tile configuration %d needs channel counts that are multiples of 64", id);
  g_last_cfg = ci->name;
  // CYCLEDIFF_GEMM_TRACE=1: one line per launch, in launch order (scripts/pmc_traffic_by_shape.py matches them with the
  // dispatches of a rocprofv3 counter pass: M N K KH stride up cat nbatch act resid out_f32 chm | tile)
  static const bool trace = [] { const char* e = getenv("CYCLEDIFF_GEMM_TRACE"); return e && e[0] == '1'; }();
  if (trace) {
    const int64_t a_bytes = (int64_t)p.B * p.Hs * p.Ws * Ctot * 2;
    const bool chm = id != kLinStreamTile && pk.korder != 0 && p.KH * p.KW > 1 && p.KH <= 8 && p.KW <= 8 && !p.up &&
                     (pk.korder == 2 || (p.Hs * p.Ws >= 4096 && a_bytes >= (32ll << 20)));
    fprintf(stderr, "[gemm_trace] %d %d %d %d %d %d %d %d %d %d %d %d | %s x%d\n", p.M, p.N, p.Ktot, p.KH, p.stride, p.up,
            p.src1 ? 1 : 0, p.nbatch, p.act, p.resid ? 1 : 0, p.out_f32, chm ? 1 : 0, ci->name, pk.splitk > 1 ? pk.splitk : 1);
  }
  KernelProfiler* prof = g_conv_prof;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof && prof->enabled) {
    char what[112] = "";
    if (prof->verbose)
      snprintf(what, sizeof(what), "M%d N%d K%d k%d s%d%s%s z%d act%d | %s x%d", p.M, p.N, p.Ktot, p.KH, p.stride,
               p.up ? " up" : "", p.src1 ? " cat" : "", p.nbatch, p.act, ci->name, pk.splitk > 1 ? pk.splitk : 1);
    if (prof->verbose && !k64) strncat(what, " bk32", sizeof(what) - strlen(what) - 1);
    prof->next_pair(&e0, &e1, 2.0 * (double)p.M * (double)p.N * (double)p.Ktot * (double)p.nbatch * p.prof_flop_scale,
                    what);
    (void)hipEventRecord(e0, st);
  }
  struct Closer {  // record the stop event on every exit path
    hipEvent_t e; hipStream_t s;
    ~Closer() { if (e) (void)hipEventRecord(e, s); }
  } closer{e1, st};
  if (id == kLinStreamTile) launch_lin_stream(st, pk);
  else if (k64) dispatch<64>(st, pk, id);
  else dispatch<32>(st, pk, id);
}

int conv_gemm_num_configs() { return kNumCfgs; }
const char* conv_gemm_config_name(int id) {
  for (int i = 0; i < kNumCfgs; ++i) if (kCfgs[i].id == id) return kCfgs[i].name;
  return "?";
}

// ------------------------------------------------------------------------------------------------
// Weight repack: torch fp32 [N][Cin][KH][KW] -> 16-bit [Npad][KH][KW][Cpad], zero padded.
// geglu=1: source rows are [value(N/2) | gate(N/2)] (GEGLU.proj, attention.py:37-44); packed rows
// are interleaved in blocks of 32 so value n and gate n sit 32 columns apart in one wave tile.
__global__ void k_repack_weight(const float* __restrict__ w, bf16_t* __restrict__ out, int N, int Cin,
                                int KH, int KW, int Npad, int Cpad, int geglu, int64_t src_row_offset) {
  const int64_t total = (int64_t)Npad * KH * KW * Cpad;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    int64_t t = i / Cpad;
    const int s = (int)(t % KW); t /= KW;
    const int r = (int)(t % KH); t /= KH;
    const int np = (int)t;
    int n = np;
    if (geglu) {
      const int blk = np / 64, within = np % 64;
      n = (within < 32) ? blk * 32 + within : N / 2 + blk * 32 + (within - 32);
    }
    float v = 0.0f;
    if (np < N && c < Cin) v = w[(((int64_t)(n + src_row_offset) * Cin + c) * KH + r) * KW + s];
    out[i] = f2bf(v);
  }
}

void launch_repack_weight(hipStream_t st, const float* w, bf16_t* out, int N, int Cin, int KH,
                          int KW, int Npad, int Cpad, int geglu, int64_t src_row_offset) {
  const int64_t total = (int64_t)Npad * KH * KW * Cpad;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_repack_weight, dim3(grid), dim3(256), 0, st, w, out, N, Cin, KH, KW, Npad,
                     Cpad, geglu, src_row_offset);
}

}  // namespace cd
