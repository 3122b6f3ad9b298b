// Torch-free driver of two C-ABI test entry points of libcyclediff.so, for kernel A/B runs on a fresh GPU box
// (the first `import torch` there costs one to two minutes of a metered call; this starts in a second):
//
//   abi_bench conv  B H C0 C1 N k stride up act tile iters      -> cd_op_bench_conv (ms per launch, TFLOP/s)
//   abi_bench peak  [ms]                                        -> cd_op_bench_mfma_sustained (TFLOP/s, GHz)
//   abi_bench attn  B T H D v_transposed iters [Tk]             -> cd_op_attention  (ms per call incl. the fp32 <-> 16-bit
//                                                                   layout conversions of that entry point; kernel time:
//                                                                   rocprofv3 --kernel-trace -- abi_bench attn ...)
//
//   hipcc -O2 -I include -o scripts/ubench/abi_bench scripts/ubench/abi_bench.cpp \
//         -L cycle-diffusion_amd/lib -lcyclediff -Wl,-rpath,'$ORIGIN/../../cycle-diffusion_amd/lib'
// CYCLEDIFF_TUNE_DEFAULT=cycle-diffusion_amd/tune_gfx950.txt preloads the shipped tile table (as _ffi.py does).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "cyclediff.h"

#define HIP_OK(x)                                                                              \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)
#define CD_OK(x)                                                              \
  do {                                                                        \
    if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, cd_last_error()); exit(1); } \
  } while (0)

static float* device_random(size_t n, unsigned seed, float scale) {
  std::vector<float> h(n);
  unsigned s = seed;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = (((s >> 8) & 0xffff) / 32768.0f - 1.0f) * scale;
  }
  float* d = nullptr;
  HIP_OK(hipMalloc((void**)&d, n * sizeof(float)));
  HIP_OK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: abi_bench conv|attn ... (see the header of this file)\n"); return 2; }
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  cd_handle h = nullptr;
  CD_OK(cd_engine_create((void*)st, (size_t)8 << 30, &h));
  printf("libcyclediff version %d, 16-bit format %s\n", cd_version(), cd_act_format() == 1 ? "fp16" : "bf16");
  if (!strcmp(argv[1], "conv")) {
    if (argc < 13) { fprintf(stderr, "conv needs 11 arguments\n"); return 2; }
    int a[11];
    for (int i = 0; i < 11; ++i) a[i] = atoi(argv[2 + i]);
    const int B = a[0], H = a[1], C0 = a[2], C1 = a[3], N = a[4], k = a[5], stride = a[6], up = a[7], act = a[8], tile = a[9],
              iters = a[10];
    float ms = 0;
    CD_OK(cd_op_bench_conv(h, B, H, H, C0, C1, N, k, stride, up, act, tile, iters, &ms));
    const int ho = up ? H * 2 : H / stride;
    const double fl = 2.0 * B * ho * ho * (double)N * k * k * (C0 + C1);
    printf("conv B%d %dx%d C%d+%d -> N%d k%d s%d up%d act%d tile %d: %.3f ms  %.1f TFLOP/s\n", B, H, H, C0, C1, N, k, stride,
           up, act, tile, ms, fl / ms * 1e-9);
  } else if (!strcmp(argv[1], "peak")) {  // abi_bench peak [ms]: sustained 16-bit MFMA rate of this device (cd_op_bench_mfma_sustained)
    float tf = 0, ghz = 0;
    for (int r = 0; r < 3; ++r) {
      CD_OK(cd_op_bench_mfma_sustained(h, argc > 2 ? atoi(argv[2]) : 300, &tf, &ghz));
      printf("sustained MFMA rate: %.1f TFLOP/s at %.3f GHz\n", tf, ghz);
    }
  } else if (!strcmp(argv[1], "attn")) {
    if (argc < 8) { fprintf(stderr, "attn needs 6 arguments\n"); return 2; }
    const int B = atoi(argv[2]), T = atoi(argv[3]), H = atoi(argv[4]), D = atoi(argv[5]), vt = atoi(argv[6]),
              iters = atoi(argv[7]);
    const int Tk = argc > 8 ? atoi(argv[8]) : T;
    const size_t nq = (size_t)B * T * H * D, nk = (size_t)B * Tk * H * D;
    float* q = device_random(nq, 1, 1.0f);
    float* k = device_random(nk, 2, 1.0f);
    float* v = device_random(nk, 3, 1.0f);
    float* o = nullptr;
    HIP_OK(hipMalloc((void**)&o, nq * sizeof(float)));
    const float scale = 1.0f / sqrtf((float)D);
    CD_OK(cd_op_attention(h, q, k, v, B, H, T, Tk, D, scale, vt, o));
    HIP_OK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) CD_OK(cd_op_attention(h, q, k, v, B, H, T, Tk, D, scale, vt, o));
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> ho(16);
    HIP_OK(hipMemcpy(ho.data(), o, 16 * sizeof(float), hipMemcpyDeviceToHost));
    printf("attn B%d Tq%d Tk%d H%d d%d %s: %.3f ms per call incl. layout conversions (%.0f GFLOP of attention); o[0..3] = %g %g %g %g\n",
           B, T, Tk, H, D, vt ? "V^T" : "V token-major", ms / iters, 4.0 * B * H * (double)T * Tk * D * 1e-9, ho[0], ho[1], ho[2],
           ho[3]);
    // whole-output fingerprint (torch-free A/B of kernel variants: identical bits <=> identical xor / sums)
    std::vector<float> all(nq);
    HIP_OK(hipMemcpy(all.data(), o, nq * sizeof(float), hipMemcpyDeviceToHost));
    unsigned long long x = 0; double s1 = 0, s2 = 0; size_t bad = 0;
    for (size_t i = 0; i < nq; ++i) {
      unsigned u; memcpy(&u, &all[i], 4);
      x = (x << 1 | x >> 63) ^ u;
      if (!(all[i] == all[i]) || fabsf(all[i]) > 3e38f) { ++bad; continue; }
      s1 += all[i]; s2 += (double)all[i] * all[i];
    }
    printf("attn output fingerprint: rolling xor %016llx  sum %.9g  sum of squares %.9g  non-finite %zu\n", x, s1, s2, bad);
  } else {
    fprintf(stderr, "unknown mode %s\n", argv[1]);
    return 2;
  }
  CD_OK(cd_engine_destroy(h));
  return 0;
}
