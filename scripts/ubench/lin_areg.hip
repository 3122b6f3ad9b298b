// Prototype (NOT part of the product library): a different schedule for the short-K linear layers of the SD U-Net
// (out[M][N] = A[M][K] . W[N][K]^T, K = 320, M = B' * 4096, N = 320 ... 2560), which run at 250-500 TFLOP/s in
// conv_gemm.hip's tile loop: 5 K steps per tile with one stage in flight, an epilogue as long as
// the loop, and one 256x320 workgroup per CU so that the whole chip loads, computes and stores in lock-step.
//
// Idea measured here:
//   * the A operand never touches LDS: a wave owns 32 rows and keeps their whole K extent as MFMA fragments in registers
//     (K = 320: 20 fragments = 80 VGPRs), loaded once;
//   * W streams through a small LDS ring in FRAGMENT-READY order: a piece is 64 output columns x 160 k = 20 x 1 KiB blocks,
//     one block = one `buffer_load ... lds` wave instruction whose lane l fetches W[n0 + (l & 31)][k0 + 8 * (l >> 5) ..]
//     - so a fragment read is the lane-linear, conflict-free ds_read_b128 of that block, with no swizzle arithmetic;
//   * the ring runs across output tiles without draining (no prologue / epilogue bubble per tile), a workgroup is 4 waves
//     with 60 KB of ring + 18 KB of epilogue staging: two workgroups per CU, out of phase;
//   * D^T = W A^T, so a lane owns ONE output row; the tile leaves through a per-wave LDS transpose as 16-byte stores,
//     8 lanes per 128-byte row segment (EPI = 1), or directly as 8-byte stores (EPI = 0, the pattern DESIGN.md section 7
//     found slow: for comparison).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/ubench/lin_areg scripts/ubench/lin_areg.hip
//   scripts/ubench/lin_areg [M=131072] [N=320] [K=320|640] [iters=20]
// Prints the check against a CPU reference on sampled rows, then us / launch, TFLOP/s and effective TB/s per variant.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <vector>

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void* lptr_t;

#define HIP_OK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); }      \
  } while (0)

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int KSH = 10;             // k slices (of 16) per piece: a piece is 64 output columns x 160 k
constexpr int PIECE = 2 * KSH * 1024;  // bytes: 2 column blocks of 32 x KSH slices x 1 KiB
constexpr int EPI_LD = 72;          // fp16 elements per staged output row (64 + 8 pad: 36 dwords, conflict-free b128)

// K = 320 (20 A fragments = 80 VGPRs, 2 pieces per 64-column tile) or 640 (40 fragments = 160 VGPRs, 4 pieces).
// RESID: out[m][n] += bias-free product, read-modify-write of the same element by the same lane (the in-place
// residual update of the attention / feed-forward output projections).
template <int K, int NSTAGE, int EPI, bool RESID>
__global__ __launch_bounds__(256, 2) void k_lin_areg(const half_t* __restrict__ A, const half_t* __restrict__ W,
                                                     half_t* __restrict__ out, int M, int N) {
  constexpr int NKS = K / 16;       // 16-wide k slices = A fragments per wave
  constexpr int NPT = NKS / KSH;    // pieces per 64-column tile
  static_assert(NKS % KSH == 0, "K must be a multiple of 160");
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                // [NSTAGE][PIECE]
  half_t* stage = (half_t*)(smem + NSTAGE * PIECE);  // [4 waves][32][EPI_LD]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mi = lane & 31, half = lane >> 5;
  const int m = blockIdx.x * 128 + wave * 32 + mi;
  constexpr unsigned kNoLoad = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)M * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (unsigned)((size_t)N * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (unsigned)((size_t)M * N * 2), 0x00020000);

  // ---- A fragments (the MFMA B operand of D^T = W A^T): lane holds A[m][ks*16 + 8*half .. +7] for all 20 slices
  f16x8 af[NKS];
  {
    const unsigned base = m < M ? (unsigned)(((size_t)m * K + 8 * half) * 2) : kNoLoad;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      af[ks] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, base + ks * 32, 0, 0));
  }

  // ---- W pieces: piece p = output columns [64 * (p >> 1), +64) x k slices [KSH * (p & 1), +KSH), 20 blocks of 1 KiB:
  // block j = column block j / KSH (32 columns), slice j % KSH; this wave issues blocks wave, wave + 4, ... (5 of them)
  const int npieces = (N / 64) * NPT;
  const unsigned w_lane = (unsigned)(((size_t)mi * K + 8 * half) * 2);  // row (lane & 31), k half (lane >> 5)
  // per-wave block geometry (scalar): block j -> byte offset of its W sub-block relative to the piece origin
  unsigned blk_off[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int j = wave + 4 * u;
    blk_off[u] = (unsigned)((((j / KSH) * 32) * K + (j % KSH) * 16) * 2);
  }
  auto issue_piece = [&](int p, int slot_idx) {
    // pieces past the end are issued with an out-of-range offset (zero fill into a dead slot): every iteration has
    // the same vmcnt footprint and the loop body is branch-free (scalar select, no control flow)
    const unsigned origin = p < npieces ? (unsigned)((((p / NPT) * 64) * K + (p % NPT) * KSH * 16) * 2) : kNoLoad;
    char* slot = ring + slot_idx * PIECE + wave * 1024;
#pragma unroll
    for (int u = 0; u < 5; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(slot + u * 4096), 16, w_lane + origin + blk_off[u], 0, 0, 0);
  };

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s) issue_piece(s, s);

  int cur = 0, nxt = NSTAGE - 1;  // ring slot of piece p / slot the next prefetch goes to (= slot of piece p - 1)
  for (int nt = 0; nt < N / 64; ++nt) {
    // in-place residual: read at the START of the tile (older than the pieces issued during it), so that waiting for
    // it in the epilogue does not drain the ring (returns are in order). Same elements, same lane as the store.
    u32x4 rres[4];
    if (RESID) {
      static_assert(!RESID || EPI == 1, "the residual variant uses the LDS epilogue's store mapping");
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mm = blockIdx.x * 128 + wave * 32 + (lane >> 3) + 8 * i;
        const unsigned off = mm < M ? (unsigned)(((size_t)mm * N + nt * 64 + (lane & 7) * 8) * 2) : kNoLoad;
        rres[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_o, off, 0, 0);
      }
    }
    static_for<0, NPT>([&](auto khc) {  // compile-time k part: A fragments and wait counts are indexed statically
      constexpr int kh = decltype(khc)::value;
      const int p = nt * NPT + kh;
      // piece p has landed when at most (NSTAGE - 2) younger pieces (5 loads each per wave) are in flight; the A
      // fragment loads are older than every piece and are covered by the same count. vmcnt counts stores too, and
      // returns are in issue order: the operations issued at the tile boundary (the previous tile's NSTORE stores, this
      // tile's NRES residual reads) are younger than every piece issued before the boundary, i.e. than the first
      // NSTAGE - 1 pieces of the tile - for those the exact count is that much larger (a smaller one would still be
      // correct but would also drain the prefetched pieces)
      constexpr int NSTORE = EPI ? 4 : 8, NRES = RESID ? 4 : 0;
      constexpr bool bnd = kh <= NSTAGE - 2;
      constexpr int cnt0 = (NSTAGE - 2) * 5 + (bnd ? NRES : 0), cnt1 = cnt0 + (bnd ? NSTORE : 0);
      if (nt > 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(cnt1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(cnt0) : "memory");
      issue_piece(p + NSTAGE - 1, nxt);  // into the slot of piece p - 1: all waves are past the barrier, nobody reads it
      const char* slot = ring + cur * PIECE;
      nxt = cur;
      cur = cur + 1 == NSTAGE ? 0 : cur + 1;
#pragma unroll
      for (int ksl = 0; ksl < KSH; ++ksl) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const f16x8 wf = *(const f16x8*)(slot + (nb * KSH + ksl) * 1024 + lane * 16);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af[kh * KSH + ksl], acc[nb], 0, 0, 0);
        }
      }
    });
    {
      // the 64 columns of this tile are complete: lane owns row m, columns (r&3) + 8*(r>>2) + 4*half
      const int n0 = nt * 64;
      if (EPI == 0) {
        if (m < M) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              typedef __attribute__((ext_vector_type(4))) _Float16 h4;
              h4* op = (h4*)(out + (size_t)m * N + n0 + nb * 32 + 8 * q + 4 * half);
              h4 v = {(half_t)acc[nb][4 * q], (half_t)acc[nb][4 * q + 1], (half_t)acc[nb][4 * q + 2],
                      (half_t)acc[nb][4 * q + 3]};
              *op = v;
            }
        }
      } else {
        half_t* st = stage + wave * (32 * EPI_LD);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            typedef __attribute__((ext_vector_type(4))) _Float16 h4;
            const h4 v = {(half_t)acc[nb][4 * q], (half_t)acc[nb][4 * q + 1], (half_t)acc[nb][4 * q + 2],
                          (half_t)acc[nb][4 * q + 3]};
            *(h4*)(st + mi * EPI_LD + nb * 32 + 8 * q + 4 * half) = v;
          }
        __builtin_amdgcn_wave_barrier();  // the staging region is private to the wave: program order suffices
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // 32 rows x 8 chunks of 16 bytes: 8 lanes per 128-byte row segment
          const int row = (lane >> 3) + 8 * i, ch = lane & 7;
          const int mm = blockIdx.x * 128 + wave * 32 + row;
          u32x4 v = *(const u32x4*)(st + row * EPI_LD + ch * 8);
          if (mm < M) {
            u32x4* op = (u32x4*)(out + (size_t)mm * N + n0 + ch * 8);
            if (RESID) {  // the staged product was rounded to fp16 once; the sum is rounded again (prototype: timing)
              const f16x8 a = __builtin_bit_cast(f16x8, v), r = __builtin_bit_cast(f16x8, rres[i]);
              f16x8 sum;
#pragma unroll
              for (int e = 0; e < 8; ++e) sum[e] = (half_t)((float)a[e] + (float)r[e]);
              v = __builtin_bit_cast(u32x4, sum);
            }
            *op = v;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy tail pieces
#endif
}

template <int K, int NSTAGE, int EPI, bool RESID>
float run(const half_t* A, const half_t* W, half_t* out, int M, int N, int iters) {
  const size_t lds = (size_t)NSTAGE * PIECE + 4 * 32 * EPI_LD * sizeof(half_t);
  auto kern = k_lin_areg<K, NSTAGE, EPI, RESID>;
  HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const dim3 grid((M + 127) / 128);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, A, W, out, M, N);
  HIP_OK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, A, W, out, M, N);
  HIP_OK(hipEventRecord(e1));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

template <int K>
int bench(int M, int N, int iters) {
  std::vector<half_t> hA((size_t)M * K), hW((size_t)N * K), hO((size_t)M * N);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hA) v = (half_t)rnd();
  for (auto& v : hW) v = (half_t)(rnd() * 0.1f);
  half_t *dA, *dW, *dO;
  HIP_OK(hipMalloc(&dA, hA.size() * 2)); HIP_OK(hipMalloc(&dW, hW.size() * 2)); HIP_OK(hipMalloc(&dO, hO.size() * 2));
  HIP_OK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  const double flop = 2.0 * M * N * K, bytes = 2.0 * ((double)M * K + (double)N * K + (double)M * N);
  // the timed launches of a RESID variant keep accumulating into `out`: the check runs on ONE launch from zeros
  auto check = [&](const char* name, float (*fn)(const half_t*, const half_t*, half_t*, int, int, int)) {
    HIP_OK(hipMemset(dO, 0, hO.size() * 2));
    fn(dA, dW, dO, M, N, 0);  // iters = 0: the single untimed launch only
    HIP_OK(hipMemcpy(hO.data(), dO, hO.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 96; ++t) {
      const int mm = (int)(((size_t)t * 2654435761u) % (size_t)M);
      for (int n = 0; n < N; ++n) {
        float ref = 0;
        for (int k = 0; k < K; ++k) ref += (float)hA[(size_t)mm * K + k] * (float)hW[(size_t)n * K + k];
        const double e = fabs((double)(float)hO[(size_t)mm * N + n] - ref);
        if (e > worst) worst = e;
      }
    }
    printf("%-34s max |err| over 96 sampled rows: %.3e %s\n", name, worst, worst < 3e-2 ? "ok" : "MISMATCH");
  };
  struct V { const char* name; float (*fn)(const half_t*, const half_t*, half_t*, int, int, int); bool resid; };
  const V variants[] = {{"3-stage, LDS epilogue", run<K, 3, 1, false>, false},
                        {"3-stage, 8-B stores", run<K, 3, 0, false>, false},
                        {"2-stage, LDS epilogue", run<K, 2, 1, false>, false},
                        {"4-stage, LDS epilogue", run<K, 4, 1, false>, false},
                        {"3-stage, LDS epilogue, += out", run<K, 3, 1, true>, true}};
  printf("out[%d][%d] = A[%d][%d] . W[%d][%d]^T  (%.1f GFLOP, %.0f MB of operands; += out reads the output too)\n", M, N, M,
         K, N, K, flop * 1e-9, bytes * 1e-6);
  for (const V& v : variants) {
    check(v.name, v.fn);
    const float ms = v.fn(dA, dW, dO, M, N, iters);
    const double b = bytes + (v.resid ? 2.0 * M * N : 0.0);
    printf("%-34s %8.1f us  %7.1f TFLOP/s  %5.2f TB/s of operand traffic\n", v.name, ms * 1e3, flop / ms * 1e-9,
           b / ms * 1e-9);
  }
  HIP_OK(hipFree(dA)); HIP_OK(hipFree(dW)); HIP_OK(hipFree(dO));
  return 0;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 131072;
  const int N = argc > 2 ? atoi(argv[2]) : 320;
  const int Kd = argc > 3 ? atoi(argv[3]) : 320;
  const int iters = argc > 4 ? atoi(argv[4]) : 20;
  if (N % 64 != 0 || (Kd != 320 && Kd != 640)) { fprintf(stderr, "N %% 64 == 0, K in {320, 640}\n"); return 1; }
  return Kd == 320 ? bench<320>(M, N, iters) : bench<640>(M, N, iters);
}
