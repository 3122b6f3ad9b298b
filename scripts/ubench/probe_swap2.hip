#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(float* p, float alpha) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = p[threadIdx.x * 16 + r];
  constexpr int kReg[16] = {0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15};
  auto swap_halves = [&](f32x16& a) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      {
        const float x0 = a[e], y0 = a[8 + e];
        const auto r2 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x0),
                                                         __builtin_bit_cast(unsigned, y0), false, false);
        a[e] = __builtin_bit_cast(float, r2[0]); a[8 + e] = __builtin_bit_cast(float, r2[1]);
      }
      {
        const float x0 = a[4 + e], y0 = a[12 + e];
        const auto r2 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x0),
                                                         __builtin_bit_cast(unsigned, y0), false, false);
        a[4 + e] = __builtin_bit_cast(float, r2[0]); a[12 + e] = __builtin_bit_cast(float, r2[1]);
      }
    }
  };
  swap_halves(acc);
  float v[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) v[c] = acc[kReg[c]] * alpha;
  for (int c = 0; c < 16; ++c) p[1024 + threadIdx.x * 16 + c] = v[c];
}

int main() {
  float* d; (void)hipMalloc(&d, 2048 * 4);
  float h[2048];
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) h[l * 16 + r] = l * 100 + r;
  (void)hipMemcpy(d, h, 1024 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1.0f);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 32, 33}) { printf("lane %2d v:", l); for (int c = 0; c < 16; ++c) printf(" %g", h[1024 + l * 16 + c]); printf("\n"); }
  return 0;
}
