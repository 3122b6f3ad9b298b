// Microbenchmark: how fast can one CU fill LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA), as a function of the number
// of issuing waves and of where the data comes from (L2-resident / HBM-streamed)? conv_gemm.hip's 256x320 tile needs
// 73.7 KB per K step per CU; at its measured ~1.0-1.15 PFLOP/s that is 25-30 GB/s per CU - is that the ceiling?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/ubench/dma_fill scripts/ubench/dma_fill.hip
//   scripts/ubench/dma_fill            (prints GB/s per CU and aggregate TB/s per variant)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lptr_t;
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every wave copies `iters` x 8 KiB (8 instructions of 1 KiB) into its own 16 KiB LDS area; source offsets walk a window
// of `window` bytes (L2-resident when small) starting at a per-wave / per-block position; at most 16 instructions in
// flight per wave (counted vmcnt)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_fill(const char* src, size_t window, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffffu, 0x00020000);
  char* dst = smem + wave * 16384;
  size_t pos = ((size_t)blockIdx.x * WAVES + wave) * 8192 & (window - 1);  // window: a power of two
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + ((it & 1) * 8 + u) * 1024), 16,
                                               (unsigned)(pos + u * 1024 + lane * 16), 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    pos = (pos + (size_t)gridDim.x * WAVES * 8192) & (window - 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && smem[lane] == 123) sink[0] = 1;
}

template <int WAVES>
void run(const char* src, size_t window, const char* what, int* sink) {
  const int iters = 2048 / WAVES;  // 16 MiB per CU
  auto kern = k_fill<WAVES>;
  HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, WAVES * 16384));
  hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WAVES), WAVES * 16384, 0, src, window, iters, sink);
  HIP_OK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WAVES), WAVES * 16384, 0, src, window, iters, sink);
  HIP_OK(hipEventRecord(e1));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double bytes_per_cu = (double)iters * WAVES * 8192;
  printf("%-22s %d waves/CU: %7.1f GB/s per CU, %6.2f TB/s chip (%.3f ms)\n", what, WAVES, bytes_per_cu / ms * 1e-6,
         bytes_per_cu * 256 / ms * 1e-9, ms);
}

int main() {
  const size_t big = (size_t)1 << 30;  // 1 GiB: HBM-streamed
  char* src; int* sink;
  HIP_OK(hipMalloc(&src, big)); HIP_OK(hipMalloc(&sink, 4));
  HIP_OK(hipMemset(src, 1, big));
  // L2-resident: every CU re-reads a 2 MiB window; MALL-resident: 128 MiB; HBM: 1 GiB
  run<1>(src, 2 << 20, "L2 window 2 MiB", sink); run<2>(src, 2 << 20, "L2 window 2 MiB", sink);
  run<4>(src, 2 << 20, "L2 window 2 MiB", sink); run<8>(src, 2 << 20, "L2 window 2 MiB", sink);
  run<4>(src, (size_t)128 << 20, "MALL window 128 MiB", sink); run<8>(src, (size_t)128 << 20, "MALL window 128 MiB", sink);
  run<4>(src, big, "HBM 1 GiB", sink); run<8>(src, big, "HBM 1 GiB", sink);
  return 0;
}
