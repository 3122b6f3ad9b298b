// Probe: v_permlane32_swap semantics and the register layout of a 32x32x16 MFMA block (run on gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  unsigned a = 1000 + lane, b = 2000 + lane;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane] = (float)r[0];
  out[64 + lane] = (float)r[1];
  // MFMA: A[i][k] = i+1 for k == 0 (lane<32 element 0), B[k][j] = 1 for k==0 -> D[i][j] = i+1
  f16x8 fa = {0, 0, 0, 0, 0, 0, 0, 0}, fb = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 acc;
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  if (lane < 32) { fa[0] = (_Float16)(lane + 1); fb[0] = (_Float16)1.0f; }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
  for (int q = 0; q < 16; ++q) out[128 + lane * 16 + q] = acc[q];  // value = (A-operand row index i) + 1
}
int main() {
  float* d; hipMalloc(&d, 4096 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("r0 (new 1st operand): lane0=%g lane31=%g lane32=%g lane63=%g\n", h[0], h[31], h[32], h[63]);
  printf("r1 (new 2nd operand): lane0=%g lane31=%g lane32=%g lane63=%g\n", h[64], h[64 + 31], h[64 + 32], h[64 + 63]);
  for (int l : {0, 1, 31, 32, 33}) {
    printf("lane %2d acc (A-row index+1):", l);
    for (int q = 0; q < 16; ++q) printf(" %g", h[128 + l * 16 + q]);
    printf("\n");
  }
  return 0;
}
