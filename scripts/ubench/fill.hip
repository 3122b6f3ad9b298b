// Micro-benchmark: per-CU fill rate of L2-resident data into LDS on gfx950, three ways:
//   0: buffer_load_dwordx4 ... lds   (LDS-DMA, what conv_gemm.hip uses)
//   1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   2: global_load_dwordx4 -> VGPR only
// hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/fill scripts/ubench/fill.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(256) void k_fill(const uint4* __restrict__ src, size_t nvec, int iters, unsigned* sink) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // every block streams the same `nvec` 16-byte vectors (L2-resident), offset by block to spread channels
  const size_t span = nvec;
  size_t pos = ((size_t)blockIdx.x * 4099 + wave * 64) % span;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < INFLIGHT; ++u) {
        char* l = smem + ((wave * INFLIGHT + u) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)l, 16, (unsigned)((pos + lane) * 16), 0, 0, 0);
        pos += 256; if (pos >= span) pos -= span;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      uint4 r[INFLIGHT];
#pragma unroll
      for (int u = 0; u < INFLIGHT; ++u) {
        r[u] = src[pos + lane];
        pos += 256; if (pos >= span) pos -= span;
      }
#pragma unroll
      for (int u = 0; u < INFLIGHT; ++u) {
        if (MODE == 1) *(uint4*)(smem + ((wave * INFLIGHT + u) * 1024) + lane * 16) = r[u];
        else { acc.x ^= r[u].x; acc.y ^= r[u].y; acc.z ^= r[u].z; acc.w ^= r[u].w; }
      }
    }
  }
  if (MODE != 2) { __syncthreads(); acc = *(uint4*)(smem + tid * 16); }
  if (acc.x == 0x12345678u) sink[0] = acc.y ^ acc.z ^ acc.w;
#endif
}

template <int MODE, int INFLIGHT>
void run(const uint4* src, size_t nvec, int blocks, unsigned* sink) {
  const int iters = 2000;
  const size_t lds = 4 * INFLIGHT * 1024;
  hipFuncSetAttribute((const void*)k_fill<MODE, INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_fill<MODE, INFLIGHT>), dim3(blocks), dim3(256), lds, 0, src, nvec, 10, sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_fill<MODE, INFLIGHT>), dim3(blocks), dim3(256), lds, 0, src, nvec, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * 4 * INFLIGHT * 1024.0 * iters;
  printf("mode %d inflight %2d blocks %4d (%.1f/CU): %7.2f TB/s  %6.1f B/clk/CU(@2.4GHz,256CU)\n", MODE, INFLIGHT,
         blocks, blocks / 256.0, bytes / ms * 1e-9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 2;
  const size_t nvec = mb * (1 << 20) / 16;
  uint4* src; unsigned* sink;
  hipMalloc(&src, nvec * 16 + 65536); hipMemset(src, 1, nvec * 16 + 65536); hipMalloc(&sink, 64);
  printf("footprint %zu MB\n", mb);
  for (int blocks : {256, 512, 768, 1024}) {
    run<0, 4>(src, nvec, blocks, sink);
    run<0, 8>(src, nvec, blocks, sink);
    run<0, 16>(src, nvec, blocks, sink);
    run<1, 4>(src, nvec, blocks, sink);
    run<1, 8>(src, nvec, blocks, sink);
    run<1, 16>(src, nvec, blocks, sink);
    run<2, 8>(src, nvec, blocks, sink);
    run<2, 16>(src, nvec, blocks, sink);
  }
  return 0;
}
