// Implicit-GEMM convolution / GEMM for gfx950: bf16 MFMA 32x32x16, fp32 accumulate.
//
// One kernel family covers every contraction on the hot path:
// 3x3 s1 p1 conv, 3x3 s2 conv (U-Net Downsample p=1, openaimodel.py:134-160; VAE pad(0,1,0,1),
// model.py:72-76), 1x1 conv, nn.Linear, and batched Q.K^T / P.V for the single-head AttnBlock
// (model.py:178-202). Activations are NHWC bf16, so a K-slice of one filter tap is a contiguous
// run of channels; the skip-connection concat (openaimodel.py:736) is a second source pointer in
// the K loop and nearest-x2 upsampling (openaimodel.py:115) is folded into the gather address.
//
// Structure (CDNA4): 256 threads = 4 waves; BMxBN tile, BK-deep K steps; both operands are staged
// HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered, one barrier per
// K step; LDS rows are XOR-swizzled on the *source* side so ds_read_b128 fragment reads are
// conflict free; accumulators go through LDS in the epilogue so global stores are 16 B per lane
// with bias / time-embedding / residual / SiLU / GELU / GEGLU fused.
#include "common.h"
#include "kernels.h"

namespace cd {

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BM, int BN, int BK, int WM, int WN>
struct TileCfg {
  static constexpr int CPR = BK / 8;         // 16-byte chunks per LDS row
  static constexpr int RPI = 64 / CPR;       // rows covered by one wave-wide glds
  static constexpr int A_IPW = BM / RPI / 4; // glds instructions per wave per K step (A)
  static constexpr int B_IPW = BN / RPI / 4;
  static constexpr int TM = BM / WM, TN = BN / WN;  // wave tile
  static constexpr int MT = TM / 32, NT = TN / 32;
  static constexpr int KS = BK / 16;
  static constexpr int SWZ_SHIFT = (CPR == 8) ? 1 : 2;
  static constexpr int EPI_LD = TN + 4;      // fp32 row stride of the epilogue staging tile
  static constexpr int STAGE_BYTES = (BM + BN) * BK * 2 * 2;
  static constexpr int EPI_BYTES = 4 * TM * EPI_LD * 4;
  static constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(A_IPW >= 1 && B_IPW >= 1, "tile too small for 4 waves");
};

template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(256) void k_conv_gemm(ConvGemmParams p) {
  using T = TileCfg<BM, BN, BK, WM, WN>;
  constexpr int CPR = T::CPR, RPI = T::RPI, A_IPW = T::A_IPW, B_IPW = T::B_IPW;
  constexpr int MT = T::MT, NT = T::NT, KS = T::KS, TM = T::TM, TN = T::TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

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
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int zb = blockIdx.z;

  const bf16_t* src0 = p.src0 + (int64_t)zb * p.a_bs;
  const bf16_t* src1 = p.src1 ? p.src1 + (int64_t)zb * p.a_bs : nullptr;
  const bf16_t* wgt = p.wgt + (int64_t)zb * p.w_bs;

  // ---- per-lane staging geometry
  const int srow = lane / CPR;   // row within a glds instruction
  const int pchunk = lane % CPR; // physical 16-B chunk this lane fills
  int a_iy0[A_IPW], a_ix0[A_IPW], a_boff[A_IPW], a_lchunk[A_IPW];
  const int HWo = p.Hout * p.Wout;
#pragma unroll
  for (int i = 0; i < A_IPW; ++i) {
    const int row = (i * 4 + wave) * RPI + srow;
    const int m = m0 + row;
    a_lchunk[i] = pchunk ^ ((row >> T::SWZ_SHIFT) & (CPR - 1));
    if (m < p.M) {
      const int b = m / HWo;
      const int rem = m - b * HWo;
      const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
      a_iy0[i] = oy * p.stride - p.pad_t;
      a_ix0[i] = ox * p.stride - p.pad_l;
      a_boff[i] = b * p.Hs * p.Ws;
    } else {
      a_iy0[i] = -(1 << 28);  // always out of range -> zeros
      a_ix0[i] = 0;
      a_boff[i] = 0;
    }
  }
  const bf16_t* b_ptr[B_IPW];
#pragma unroll
  for (int i = 0; i < B_IPW; ++i) {
    const int row = (i * 4 + wave) * RPI + srow;
    const int lchunk = pchunk ^ ((row >> T::SWZ_SHIFT) & (CPR - 1));
    int nrow = n0 + row;
    if (nrow >= p.N) nrow = p.N - 1;  // clamp: duplicates a valid row, its outputs are masked
    b_ptr[i] = wgt + (int64_t)nrow * (p.ldw ? p.ldw : p.Ktot) + lchunk * 8;
  }

  const int Ctot = p.C0 + p.C1;
  const int nk = p.Ktot / BK;
  // K-step cursor (uniform): tap (kr, ks_) and channel offset kc within the concatenated channels
  int kr = 0, kss = 0, kc = 0;

  char* As = smem;
  char* Bs = smem + 2 * BM * BK * 2;

  auto stage = [&](int buf, int kt) {
    const bf16_t* sp;
    int ld, coff;
    if (kc < p.C0) { sp = src0; ld = p.ld0; coff = kc; }
    else { sp = src1; ld = p.ld1; coff = kc - p.C0; }
#pragma unroll
    for (int i = 0; i < A_IPW; ++i) {
      int iy = a_iy0[i] + kr, ix = a_ix0[i] + kss;
      const bool ok = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);
      if (p.up) { iy >>= 1; ix >>= 1; }
      const int pix = a_boff[i] + iy * p.Ws + ix;
      const bf16_t* g = ok ? sp + (int64_t)pix * ld + coff + a_lchunk[i] * 8 : p.zeros + a_lchunk[i] * 8;
      char* l = As + buf * (BM * BK * 2) + ((i * 4 + wave) * RPI) * (BK * 2);
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_IPW; ++i) {
      const bf16_t* g = b_ptr[i] + (int64_t)kt * BK;
      char* l = Bs + buf * (BN * BK * 2) + ((i * 4 + wave) * RPI) * (BK * 2);
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
    }
    // advance cursor
    kc += BK;
    if (kc >= Ctot) {
      kc = 0;
      if (++kss >= p.KW) { kss = 0; ++kr; }
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

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* Ab = As + cur * (BM * BK * 2);
    const char* Bb = Bs + cur * (BN * BK * 2);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * TM + i * 32 + frow;
        const int ch = (ks * 2 + fhalf) ^ ((row >> T::SWZ_SHIFT) & (CPR - 1));
        af[i] = *(const bf16x8*)(Ab + row * (BK * 2) + ch * 16);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * TN + j * 32 + frow;
        const int ch = (ks * 2 + fhalf) ^ ((row >> T::SWZ_SHIFT) & (CPR - 1));
        bfr[j] = *(const bf16x8*)(Bb + row * (BK * 2) + ch * 16);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = CD_MFMA_32x32x16(af[i], bfr[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS (fp32, per-wave region) -> fused elementwise -> 16-B stores
  float* E = (float*)smem + wave * (TM * T::EPI_LD);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        const int col = j * 32 + frow;
        E[row * T::EPI_LD + col] = acc[i][j][r] * p.alpha;
      }
  __syncthreads();

  const bool geglu = (p.act == ACT_GEGLU);
  // GEGLU: packed columns come in blocks of 64 = [32 value | 32 gate] (launch_repack_weight);
  // only the value half produces output, at column (n/64)*32 + n%32.
  constexpr int VPR_FULL = TN / 8;
  const int vpr = geglu ? (TN / 16) : VPR_FULL;  // 8-wide vectors per row handled
  const int rpp = 64 / vpr;                      // rows per pass
  const int vr = lane / vpr, vc = lane % vpr;
  char* outp = (char*)p.out + (int64_t)zb * p.o_bs * (p.out_f32 ? 4 : 2);
  for (int r0 = 0; r0 < TM; r0 += rpp) {
    const int row = r0 + vr;
    const int m = m0 + wm * TM + row;
    int col;  // column inside the wave tile
    if (geglu) col = (vc / 4) * 64 + (vc % 4) * 8;
    else col = vc * 8;
    const int n = n0 + wn * TN + col;  // packed column
    if (m >= p.M || n >= p.N) continue;
    float v[8];
    {
      const f32x4 lo = *(const f32x4*)(E + row * T::EPI_LD + col);
      const f32x4 hi = *(const f32x4*)(E + row * T::EPI_LD + col + 4);
      v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
      v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    }
    const int nvalid = (p.N - n) < 8 ? (p.N - n) : 8;
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] += p.bias[n + e];
    }
    int on = n;  // output column
    if (geglu) {
      float gt[8];
      const f32x4 lo = *(const f32x4*)(E + row * T::EPI_LD + col + 32);
      const f32x4 hi = *(const f32x4*)(E + row * T::EPI_LD + col + 36);
      gt[0] = lo[0]; gt[1] = lo[1]; gt[2] = lo[2]; gt[3] = lo[3];
      gt[4] = hi[0]; gt[5] = hi[1]; gt[6] = hi[2]; gt[7] = hi[3];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gg = gt[e] + (p.bias ? p.bias[n + 32 + e] : 0.0f);
        v[e] = v[e] * gelu_f(gg);
      }
      on = (n / 64) * 32 + (n % 64);
    } else {
      if (p.rowvec) {
        const float* rv = p.rowvec + (int64_t)(m / p.rows_per_vec) * p.rowvec_ld + n;
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nvalid) v[e] += rv[e];
      }
      if (p.act == ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
      } else if (p.act == ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
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
  }
}

template <int BM, int BN, int BK, int WM, int WN>
void launch_cfg(hipStream_t st, const ConvGemmParams& p) {
  using T = TileCfg<BM, BN, BK, WM, WN>;
  static bool attr_set = false;
  auto kern = k_conv_gemm<BM, BN, BK, WM, WN>;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  T::LDS_BYTES));
    attr_set = true;
  }
  const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
  hipLaunchKernelGGL(kern, dim3(tiles, 1, p.nbatch), dim3(256), T::LDS_BYTES, st, p);
}

thread_local const char* g_last_cfg = "";

}  // namespace

const char* conv_gemm_last_config() { return g_last_cfg; }

KernelProfiler* g_conv_prof = nullptr;

void KernelProfiler::next_pair(hipEvent_t* a, hipEvent_t* b, double fl) {
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
  }
  *launches = used / 2; *total_ms = ms; *total_flops = fl;
  used = 0; flops.clear();
}
KernelProfiler::~KernelProfiler() { for (auto e : events) (void)hipEventDestroy(e); }

void launch_conv_gemm(hipStream_t st, const ConvGemmParams& p) {
  const int Ctot = p.C0 + p.C1;
  CD_CHECK(p.C0 % 32 == 0 && p.C1 % 32 == 0, "conv_gemm: channels must be multiples of 32 (C0=%d C1=%d)", p.C0, p.C1);
  CD_CHECK(p.Ktot == p.KH * p.KW * Ctot, "conv_gemm: Ktot mismatch");
  CD_CHECK(p.zeros != nullptr, "conv_gemm: zero page missing");
  CD_CHECK(p.M > 0 && p.N > 0, "conv_gemm: empty problem");
  CD_CHECK(((uintptr_t)p.src0 & 15) == 0 && ((uintptr_t)p.wgt & 15) == 0, "conv_gemm: 16-B alignment");
  CD_CHECK((p.ld0 % 8) == 0 && (p.src1 == nullptr || (p.ld1 % 8) == 0), "conv_gemm: ld must be a multiple of 8");
  if (p.act == ACT_GEGLU) CD_CHECK(p.N % 64 == 0, "GEGLU needs packed N %% 64 == 0");
  const bool k64 = (p.C0 % 64 == 0) && (p.C1 % 64 == 0);
  // tile selection: fill >= ~1 wave of the 256 CUs when the problem allows it
  int tile = p.tile;
  if (tile == 0) {
    const int64_t t128 = (int64_t)ceil_div(p.M, 128) * ceil_div(p.N, 128) * p.nbatch;
    const int64_t t12864 = (int64_t)ceil_div(p.M, 128) * ceil_div(p.N, 64) * p.nbatch;
    if (p.act == ACT_GEGLU) tile = (t128 >= 384) ? 1 : 2;  // GEGLU needs TN >= 64
    else if (t128 >= 384 && p.N % 128 == 0) tile = 1;
    else if (t128 >= 512) tile = 1;
    else if (t12864 >= 256) tile = 2;
    else tile = 3;
  }
  if (p.act == ACT_GEGLU && tile == 3) tile = 2;
  KernelProfiler* prof = g_conv_prof;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof && prof->enabled) {
    prof->next_pair(&e0, &e1, 2.0 * (double)p.M * (double)p.N * (double)p.Ktot * (double)p.nbatch);
    (void)hipEventRecord(e0, st);
  }
  struct Closer {  // record the stop event on every exit path of the switch below
    hipEvent_t e; hipStream_t s;
    ~Closer() { if (e) (void)hipEventRecord(e, s); }
  } closer{e1, st};
  if (k64) {
    switch (tile) {
      case 1: g_last_cfg = "128x128x64"; launch_cfg<128, 128, 64, 2, 2>(st, p); break;
      case 2: g_last_cfg = "128x64x64"; launch_cfg<128, 64, 64, 4, 1>(st, p); break;
      default: g_last_cfg = "64x64x64"; launch_cfg<64, 64, 64, 2, 2>(st, p); break;
    }
  } else {
    switch (tile) {
      case 1: g_last_cfg = "128x128x32"; launch_cfg<128, 128, 32, 2, 2>(st, p); break;
      case 2: g_last_cfg = "128x64x32"; launch_cfg<128, 64, 32, 4, 1>(st, p); break;
      default: g_last_cfg = "64x64x32"; launch_cfg<64, 64, 32, 2, 2>(st, p); break;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight repack: torch fp32 [N][Cin][KH][KW] -> bf16 [Npad][KH][KW][Cpad], zero padded.
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
