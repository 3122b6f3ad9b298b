// Diagnostic kernel: what the matrix cores of THIS box sustain on 16-bit operands under its power cap (DESIGN.md section 7,
// round 4). Every SIMD runs a bare v_mfma_f32_32x32x16 loop - two waves, ten independent 32 x 32 accumulator blocks each (the
// register blocking of the 256 x 320 conv tile), operand fragments pseudo-random and resident in registers, no memory traffic.
// On an MI355X (1400 W package cap) this settles at 1.7-1.8 GHz = 1.7-1.8 PFLOP/s (fp16), not at 2.4 GHz x 1024 flop/cycle:
// `bench.py` reports it beside roofline.peak, so that a bench line carries the ceiling of the lease it was measured on.
// Register-exact inline asm (as scripts/ubench/gen_pipe_ubench.py): accumulators v[64:223], fragments v[8:35].
#include "common.h"
#include "kernels.h"

namespace cd {
namespace {

#if CD_ACT_FP16
#define CD_DIAG_MFMA "v_mfma_f32_32x32x16_f16"
#else
#define CD_DIAG_MFMA "v_mfma_f32_32x32x16_bf16"
#endif
#define MF(acc, a, b) CD_DIAG_MFMA " " acc ", " a ", " b ", " acc "\n"

__global__ __launch_bounds__(512) void k_mfma_sustained(int iters, unsigned long long* clk, float* sink) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) unsigned seed_words[64 * 4 * 7];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 4 * 7; i += blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
#if CD_ACT_FP16
    seed_words[i] = (0x3800u | (h & 0x83ffu)) | ((0x3800u | ((h >> 16) & 0x83ffu)) << 16);  // +-[0.5, 1) pairs
#else
    seed_words[i] = (0x3f00u | (h & 0x807fu)) | ((0x3f00u | ((h >> 16) & 0x807fu)) << 16);
#endif
  }
  __syncthreads();
  typedef __attribute__((address_space(3))) unsigned* lds_u32_t;
  const unsigned addr = (unsigned)(uintptr_t)(lds_u32_t)seed_words + lane * 16;
  unsigned long long t0, t1, r0, r1;
  float res;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  asm volatile(
      "ds_read_b128 v[8:11], %[addr]\n"
      "ds_read_b128 v[12:15], %[addr] offset:1024\n"
      "ds_read_b128 v[16:19], %[addr] offset:2048\n"
      "ds_read_b128 v[20:23], %[addr] offset:3072\n"
      "ds_read_b128 v[24:27], %[addr] offset:4096\n"
      "ds_read_b128 v[28:31], %[addr] offset:5120\n"
      "ds_read_b128 v[32:35], %[addr] offset:6144\n"
      "v_mov_b32 v64, 0\n"
      "v_mov_b32 v65, 0\n"
      "v_mov_b32 v66, 0\n"
      "v_mov_b32 v67, 0\n"
      "v_mov_b32 v68, 0\n"
      "v_mov_b32 v69, 0\n"
      "v_mov_b32 v70, 0\n"
      "v_mov_b32 v71, 0\n"
      "v_mov_b32 v72, 0\n"
      "v_mov_b32 v73, 0\n"
      "v_mov_b32 v74, 0\n"
      "v_mov_b32 v75, 0\n"
      "v_mov_b32 v76, 0\n"
      "v_mov_b32 v77, 0\n"
      "v_mov_b32 v78, 0\n"
      "v_mov_b32 v79, 0\n"
      "v_mov_b32 v80, 0\n"
      "v_mov_b32 v81, 0\n"
      "v_mov_b32 v82, 0\n"
      "v_mov_b32 v83, 0\n"
      "v_mov_b32 v84, 0\n"
      "v_mov_b32 v85, 0\n"
      "v_mov_b32 v86, 0\n"
      "v_mov_b32 v87, 0\n"
      "v_mov_b32 v88, 0\n"
      "v_mov_b32 v89, 0\n"
      "v_mov_b32 v90, 0\n"
      "v_mov_b32 v91, 0\n"
      "v_mov_b32 v92, 0\n"
      "v_mov_b32 v93, 0\n"
      "v_mov_b32 v94, 0\n"
      "v_mov_b32 v95, 0\n"
      "v_mov_b32 v96, 0\n"
      "v_mov_b32 v97, 0\n"
      "v_mov_b32 v98, 0\n"
      "v_mov_b32 v99, 0\n"
      "v_mov_b32 v100, 0\n"
      "v_mov_b32 v101, 0\n"
      "v_mov_b32 v102, 0\n"
      "v_mov_b32 v103, 0\n"
      "v_mov_b32 v104, 0\n"
      "v_mov_b32 v105, 0\n"
      "v_mov_b32 v106, 0\n"
      "v_mov_b32 v107, 0\n"
      "v_mov_b32 v108, 0\n"
      "v_mov_b32 v109, 0\n"
      "v_mov_b32 v110, 0\n"
      "v_mov_b32 v111, 0\n"
      "v_mov_b32 v112, 0\n"
      "v_mov_b32 v113, 0\n"
      "v_mov_b32 v114, 0\n"
      "v_mov_b32 v115, 0\n"
      "v_mov_b32 v116, 0\n"
      "v_mov_b32 v117, 0\n"
      "v_mov_b32 v118, 0\n"
      "v_mov_b32 v119, 0\n"
      "v_mov_b32 v120, 0\n"
      "v_mov_b32 v121, 0\n"
      "v_mov_b32 v122, 0\n"
      "v_mov_b32 v123, 0\n"
      "v_mov_b32 v124, 0\n"
      "v_mov_b32 v125, 0\n"
      "v_mov_b32 v126, 0\n"
      "v_mov_b32 v127, 0\n"
      "v_mov_b32 v128, 0\n"
      "v_mov_b32 v129, 0\n"
      "v_mov_b32 v130, 0\n"
      "v_mov_b32 v131, 0\n"
      "v_mov_b32 v132, 0\n"
      "v_mov_b32 v133, 0\n"
      "v_mov_b32 v134, 0\n"
      "v_mov_b32 v135, 0\n"
      "v_mov_b32 v136, 0\n"
      "v_mov_b32 v137, 0\n"
      "v_mov_b32 v138, 0\n"
      "v_mov_b32 v139, 0\n"
      "v_mov_b32 v140, 0\n"
      "v_mov_b32 v141, 0\n"
      "v_mov_b32 v142, 0\n"
      "v_mov_b32 v143, 0\n"
      "v_mov_b32 v144, 0\n"
      "v_mov_b32 v145, 0\n"
      "v_mov_b32 v146, 0\n"
      "v_mov_b32 v147, 0\n"
      "v_mov_b32 v148, 0\n"
      "v_mov_b32 v149, 0\n"
      "v_mov_b32 v150, 0\n"
      "v_mov_b32 v151, 0\n"
      "v_mov_b32 v152, 0\n"
      "v_mov_b32 v153, 0\n"
      "v_mov_b32 v154, 0\n"
      "v_mov_b32 v155, 0\n"
      "v_mov_b32 v156, 0\n"
      "v_mov_b32 v157, 0\n"
      "v_mov_b32 v158, 0\n"
      "v_mov_b32 v159, 0\n"
      "v_mov_b32 v160, 0\n"
      "v_mov_b32 v161, 0\n"
      "v_mov_b32 v162, 0\n"
      "v_mov_b32 v163, 0\n"
      "v_mov_b32 v164, 0\n"
      "v_mov_b32 v165, 0\n"
      "v_mov_b32 v166, 0\n"
      "v_mov_b32 v167, 0\n"
      "v_mov_b32 v168, 0\n"
      "v_mov_b32 v169, 0\n"
      "v_mov_b32 v170, 0\n"
      "v_mov_b32 v171, 0\n"
      "v_mov_b32 v172, 0\n"
      "v_mov_b32 v173, 0\n"
      "v_mov_b32 v174, 0\n"
      "v_mov_b32 v175, 0\n"
      "v_mov_b32 v176, 0\n"
      "v_mov_b32 v177, 0\n"
      "v_mov_b32 v178, 0\n"
      "v_mov_b32 v179, 0\n"
      "v_mov_b32 v180, 0\n"
      "v_mov_b32 v181, 0\n"
      "v_mov_b32 v182, 0\n"
      "v_mov_b32 v183, 0\n"
      "v_mov_b32 v184, 0\n"
      "v_mov_b32 v185, 0\n"
      "v_mov_b32 v186, 0\n"
      "v_mov_b32 v187, 0\n"
      "v_mov_b32 v188, 0\n"
      "v_mov_b32 v189, 0\n"
      "v_mov_b32 v190, 0\n"
      "v_mov_b32 v191, 0\n"
      "v_mov_b32 v192, 0\n"
      "v_mov_b32 v193, 0\n"
      "v_mov_b32 v194, 0\n"
      "v_mov_b32 v195, 0\n"
      "v_mov_b32 v196, 0\n"
      "v_mov_b32 v197, 0\n"
      "v_mov_b32 v198, 0\n"
      "v_mov_b32 v199, 0\n"
      "v_mov_b32 v200, 0\n"
      "v_mov_b32 v201, 0\n"
      "v_mov_b32 v202, 0\n"
      "v_mov_b32 v203, 0\n"
      "v_mov_b32 v204, 0\n"
      "v_mov_b32 v205, 0\n"
      "v_mov_b32 v206, 0\n"
      "v_mov_b32 v207, 0\n"
      "v_mov_b32 v208, 0\n"
      "v_mov_b32 v209, 0\n"
      "v_mov_b32 v210, 0\n"
      "v_mov_b32 v211, 0\n"
      "v_mov_b32 v212, 0\n"
      "v_mov_b32 v213, 0\n"
      "v_mov_b32 v214, 0\n"
      "v_mov_b32 v215, 0\n"
      "v_mov_b32 v216, 0\n"
      "v_mov_b32 v217, 0\n"
      "v_mov_b32 v218, 0\n"
      "v_mov_b32 v219, 0\n"
      "v_mov_b32 v220, 0\n"
      "v_mov_b32 v221, 0\n"
      "v_mov_b32 v222, 0\n"
      "v_mov_b32 v223, 0\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_mov_b32 s20, %[iters]\n"
      "L_loop_%=:\n"
      MF("v[64:79]", "v[8:11]", "v[16:19]")
      MF("v[80:95]", "v[8:11]", "v[20:23]")
      MF("v[96:111]", "v[8:11]", "v[24:27]")
      MF("v[112:127]", "v[8:11]", "v[28:31]")
      MF("v[128:143]", "v[8:11]", "v[32:35]")
      MF("v[144:159]", "v[12:15]", "v[16:19]")
      MF("v[160:175]", "v[12:15]", "v[20:23]")
      MF("v[176:191]", "v[12:15]", "v[24:27]")
      MF("v[192:207]", "v[12:15]", "v[28:31]")
      MF("v[208:223]", "v[12:15]", "v[32:35]")
      MF("v[64:79]", "v[8:11]", "v[16:19]")
      MF("v[80:95]", "v[8:11]", "v[20:23]")
      MF("v[96:111]", "v[8:11]", "v[24:27]")
      MF("v[112:127]", "v[8:11]", "v[28:31]")
      MF("v[128:143]", "v[8:11]", "v[32:35]")
      MF("v[144:159]", "v[12:15]", "v[16:19]")
      MF("v[160:175]", "v[12:15]", "v[20:23]")
      MF("v[176:191]", "v[12:15]", "v[24:27]")
      MF("v[192:207]", "v[12:15]", "v[28:31]")
      MF("v[208:223]", "v[12:15]", "v[32:35]")
      "s_sub_u32 s20, s20, 1\n"
      "s_cmp_lg_u32 s20, 0\n"
      "s_cbranch_scc1 L_loop_%=\n"
      "s_nop 15\n"
      "s_nop 15\n"
      "v_mov_b32 %[res], v64\n"
      : [res] "=v"(res)
      : [addr] "v"(addr), [iters] "s"(iters)
      : "memory", "scc", "s20", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223");
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  if (lane == 0) {
    atomicMax(&clk[2 * blockIdx.x], t1 - t0);
    atomicMax(&clk[2 * blockIdx.x + 1], r1 - r0);
  }
  if (res == 123.456f) sink[0] = res;  // never true: keeps the result live
#endif
}

}  // namespace

// Runs the loop on every CU for about `target_ms` after a warm-up launch; tflops = launched MFMA flops / event time,
// ghz = shader clock from s_memtime against the 100 MHz s_memrealtime (average over workgroups, last launch).
void launch_mfma_sustained(hipStream_t st, int target_ms, float* tflops, float* ghz) {
  int dev = 0;
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDevice(&dev));
  HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  unsigned long long* clk = nullptr;
  float* sink = nullptr;
  HIP_CHECK(hipMalloc((void**)&clk, (size_t)ncu * 16 + 64));
  sink = (float*)(clk + 2 * ncu);
  HIP_CHECK(hipMemsetAsync(clk, 0, (size_t)ncu * 16 + 64, st));
  const int iters = 20000;                                   // 20 MFMAs x 32 cycles x 2 waves per SIMD: ~14 ms at 1.8 GHz
  const double flop_per_launch = (double)ncu * 8 * iters * 20 * 32768.0;
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_mfma_sustained, dim3(ncu), dim3(512), 0, st, iters, clk, sink);  // clocks settle
  HIP_CHECK(hipMemsetAsync(clk, 0, (size_t)ncu * 16, st));
  int reps = target_ms / 14;
  if (reps < 2) reps = 2;
  if (reps > 200) reps = 200;
  HIP_CHECK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_mfma_sustained, dim3(ncu), dim3(512), 0, st, iters, clk, sink);
  HIP_CHECK(hipEventRecord(e1, st));
  HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0;
  HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)2 * ncu);
  HIP_CHECK(hipMemcpy(h.data(), clk, (size_t)ncu * 16, hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (int b = 0; b < ncu; ++b) { cyc += (double)h[2 * b]; rt += (double)h[2 * b + 1]; }
  *tflops = (float)(flop_per_launch * reps / (ms * 1e-3) * 1e-12);
  *ghz = rt > 0 ? (float)(cyc / rt * 0.1) : 0.f;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(clk);
}

}  // namespace cd
