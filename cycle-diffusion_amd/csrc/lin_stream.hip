// Streaming linear layer for the 320-channel level of the SD / LDM U-Nets (K = 320): out[M][N] = epi(A[M][320] . W[N][320]^T)
// (attention.py:37-44 GEGLU.proj, :171-200 to_q / to_k / to_out, :211-215 the residual adds; openaimodel.py proj_in /
// proj_out 1x1 convs). At 64 x 64 tokens these layers are HBM-bound (N = K = 320: 168-252 MB per launch against
// 26.8 GFLOP) and ran at 2.4 TB/s in conv_gemm.hip's tile loop: one 256x320 workgroup per CU, so the whole chip loads,
// computes and stores in lock-step, with a prologue and an epilogue bubble per tile.
//
// Schedule here (CDNA4, one persistent 8-wave workgroup per CU):
//   * a workgroup owns 256-row strips of A (wave w: rows 32 w ..), strip after strip. A wave keeps the whole K extent of
//     its 32 rows as MFMA B-operand fragments in registers (20 x 4 VGPRs) and, beside them, the fragments of its NEXT
//     strip (another 80): A is read from HBM exactly once, in full 128-byte lines, by `buffer_load ... lds` into a small
//     ring of 128-row x 64-k pieces (XOR-swizzled on the source side), and copied LDS -> registers while the current
//     strip computes. HBM therefore always has the next strip's loads and the current strip's stores in flight.
//   * W streams from L2 through a 3-deep LDS ring in FRAGMENT-MAJOR order (k_pack_wfrag below): a 1-KiB block is one
//     32-column x 16-k MFMA A operand, so a piece (64 columns x 160 k) is twenty linear 1-KiB copies and a fragment read
//     is the lane-linear, conflict-free ds_read_b128 of a block; the ring runs across tile and strip boundaries.
//   * D^T = W A^T: a lane owns one output row. A 64-column tile leaves through a per-wave fp32 LDS transpose (32 columns
//     at a time) as 16-byte stores with bias / residual / GEGLU / GroupNorm statistics fused, one rounding to 16 bits.
//   * roles: waves 0-3 issue the W pieces, waves 4-7 the A pieces, so the W waits (L2 latency) never queue behind A loads
//     (HBM latency) in a wave's in-order VMEM counter. Every wait is COUNTED: each wave keeps a running count of the
//     LOADS it has issued and remembers the count after each piece; the wait for a piece is s_waitcnt
//     vmcnt(now - then), rounded down to a multiple of 4 (stores are left out of the count: they may retire out of
//     order with respect to loads, and leaving them out makes a wait cover them as well). Residual loads go through inline asm (hipcc drains
//     vmcnt(0) for an ordinary load beside LDS-DMA) and are issued one tile ahead.
// Per-element reduction order is k-ascending, as in conv_gemm.hip: results are bit-identical to its tiles.
#include <stdlib.h>

#include <mutex>

#include "common.h"
#include "kernels.h"

namespace cd {
namespace lin_detail {

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int K = 320, NKS = K / 16, KSH = NKS / 2;  // k slices of 16 per row / per W piece
constexpr int WPIECE = 2 * KSH * 1024, WSLOTS = 3;   // 64 columns x 160 k
constexpr int APIECE = 128 * 128, ASLOTS = 3;        // 128 rows x 64 k; 2 pieces in flight + 1 being read
constexpr int NAP = 10;                              // A pieces per 256-row strip
constexpr int ST_LD = 36, ST_BYTES = 32 * ST_LD * 4; // per-wave fp32 transpose: 32 rows x (32 + 4) floats
constexpr int BIAS_MAX = 2560;                       // floats in the LDS bias table
constexpr int OFF_A = WSLOTS * WPIECE, OFF_ST = OFF_A + ASLOTS * APIECE, OFF_BIAS = OFF_ST + 8 * ST_BYTES;
constexpr int LDS_BYTES = OFF_BIAS + BIAS_MAX * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
constexpr unsigned kNoLoad = 0x80000000u;
constexpr int kGegluDirectDefault = 1;  // LinStreamParams::geglu_direct unless CYCLEDIFF_GEGLU_DIRECT says otherwise

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (rounded DOWN to a multiple of 4: waiting for a few more
// operations than necessary is always correct)
__device__ __forceinline__ void wait_vmcnt_le(int n) {
  n = n < 0 ? 0 : (n > 60 ? 60 : n);
  switch (n >> 2) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
  }
}

// 16-byte buffer load the compiler's wait-count pass does not see (it would wait vmcnt(0) for it while LDS-DMA is in
// flight); the caller waits by count and ties the registers to the wait with touch4().
__device__ __forceinline__ u32x4 load16_hidden(u32x4 rsrc, unsigned voff) {
  u32x4 r;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(voff), "s"(rsrc) : "memory");
  return r;
}
// eight fp32 values -> the 16-bit storage format, saturating like f2bf(): one v_med3 per value and one packed convert per
// pair (common.h's pack8 compiles to two compare / select pairs, a convert, an SDWA convert and an OR per pair: ~56
// VALU operations per vector, a fifth of this kernel's GEGLU epilogue). Same bits as pack8 for every non-NaN input
// (a NaN saturates here instead of passing through).
__device__ __forceinline__ uint4 pack8_sat(const float* f) {
#if CD_ACT_FP16
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const h2 v = {(_Float16)__builtin_amdgcn_fmed3f(f[2 * i], -65504.0f, 65504.0f),
                  (_Float16)__builtin_amdgcn_fmed3f(f[2 * i + 1], -65504.0f, 65504.0f)};
    w[i] = __builtin_bit_cast(uint32_t, v);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
#else
  return pack8(f);
#endif
}
__device__ __forceinline__ void touch4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}

// LNF: LayerNorm folded in. The rows of A are normalised in registers when a strip's fragments become current - a lane
// holds half of its row (160 values), the other half sits in lane ^ 32 - and the layer runs on weights that carry the
// LayerNorm's gain and bias (k_fold_ln): y = ((x - mean) rstd) . (W gamma)^T + (b + W beta), attention.py:211-215.
template <int ACT, bool RESID, bool STATS, bool LNF = false>
__global__ __launch_bounds__(512, 2) void k_lin_stream(LinStreamParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool GEGLU = (ACT == ACT_GEGLU);
  static_assert(!(GEGLU && (RESID || STATS)), "GEGLU has neither residual nor statistics");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wring = smem;
  char* aring = smem + OFF_A;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* stage = (float*)(smem + OFF_ST + wave * ST_BYTES);
  float* bias_s = (float*)(smem + OFF_BIAS);
  const int mi = lane & 31, half = lane >> 5;
  const bool w_issuer = wave < 4;
  const int wq = wave & 3;

  const int NT = p.N >> 6;    // 64-column tiles (packed columns)
  const int I = 2 * NT;       // W pieces = iterations per strip
  const int nstrips = (p.M + 255) >> 8;
  const int G = gridDim.x;
  const int my_count = (nstrips - (int)blockIdx.x + G - 1) / G;
  if (my_count <= 0) return;
  // A pieces of the NEXT strip are consumed at iterations I - 10 cs + j cs of the current one, issued 2 cs earlier
  const int cs = I >= 80 ? 4 : (I >= 40 ? 2 : 1);

  const __amdgpu_buffer_rsrc_t rs_a =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (unsigned)((size_t)p.M * p.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.wfrag, 0, (unsigned)((size_t)p.N * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.out, 0, (unsigned)((size_t)p.M * p.ldo * 2), 0x00020000);
  u32x4 rs_r = {0u, 0u, 0u, 0x00020000u};
  if (RESID) {
    const uint64_t ra = (uint64_t)p.resid;
    rs_r[0] = (unsigned)ra; rs_r[1] = (unsigned)(ra >> 32) & 0xffffu;
    rs_r[2] = (unsigned)((size_t)p.M * p.ldr * 2);
  }

  // ---- bias table (fp32, packed column order; zeros when the layer has none)
  for (int i = tid; i < p.N; i += 512) bias_s[i] = p.bias ? p.bias[i] : 0.f;

  // ---- A piece geometry: piece j of a strip = rows [128 (j & 1), +128) x k [64 (j >> 1), +64). An A-issuer wave
  // covers rows 32 wq .. +32 of the piece with 4 instructions of 8 rows x 128 B; lane l of instruction u fills LDS
  // row 8 x + (l >> 3), physical chunk l & 7, from logical chunk (l & 7) ^ ((row >> 1) & 7) (conflict-free b128 reads)
  auto issue_a = [&](int row0, int j, int slot) {
    const unsigned a_row = (unsigned)((lane >> 3) * p.lda * 2);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int x = wq * 4 + u;
      const unsigned a_lane = a_row + (unsigned)(((lane & 7) ^ ((4 * u + (lane >> 4)) & 7)) * 16);
      const unsigned base = (unsigned)(((row0 + 128 * (j & 1) + 8 * x) * p.lda + (j >> 1) * 64) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(aring + slot * APIECE + x * 1024), 16, a_lane + base, 0,
                                               0, 0);
    }
  };
  // ---- W piece geometry: piece (t, kh) = column blocks 2t, 2t + 1 x k slices [10 kh, +10): twenty 1-KiB blocks,
  // block x = column block x / 10, slice x % 10; W-issuer wave wq copies blocks wq, wq + 4, ... (5 of them)
  int w_boff[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int x = wq + 4 * u;
    w_boff[u] = ((x / KSH) * NKS + (x % KSH)) * 1024;
  }
  auto issue_w = [&](int piece, bool valid, int slot) {
    const int base = ((piece >> 1) * 2 * NKS + (piece & 1) * KSH) * 1024;
    const unsigned voff = valid ? (unsigned)(lane * 16) : kNoLoad;  // a dead piece zero-fills a dead slot
#pragma unroll
    for (int u = 0; u < 5; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(wring + slot * WPIECE + (wq + 4 * u) * 1024), 16, voff,
                                               base + w_boff[u], 0, 0);
  };

  // ---- fragment registers: this strip's and the next strip's A (MFMA B operand: lane = row mi, k half `half`)
  bf16x8 af[NKS], afn[NKS];
  // LayerNorm of the wave's 32 rows in place (fp32 statistics and arithmetic, one rounding back to 16 bits)
  auto normalise_rows = [&]() __attribute__((always_inline)) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      float f[8];
      unpack8(__builtin_bit_cast(uint4, af[ks]), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1 += f[e]; s2 = __builtin_fmaf(f[e], f[e], s2); }
      __builtin_amdgcn_sched_barrier(0);  // one fragment at a time (160 unpacked values would not fit)
    }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    const float mean = s1 * (1.0f / K);
    const float var = fmaxf(s2 * (1.0f / K) - mean * mean, 0.f);
    const float rstd = __builtin_amdgcn_rsqf(var + p.ln_eps);
    const float nmr = -mean * rstd;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      float f[8];
      unpack8(__builtin_bit_cast(uint4, af[ks]), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = __builtin_fmaf(f[e], rstd, nmr);
      af[ks] = __builtin_bit_cast(bf16x8, pack8_sat(f));
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // copy one landed A piece into the NEXT-strip registers (only the 4 waves whose rows it holds)
  auto take_a = [&](int j, int slot) {
    if ((j & 1) != (wave >> 2)) return;
    const int row = wq * 32 + mi;
    const int sw = (row >> 1) & 7;
    // chunk (2 f + half) ^ sw of a 128-byte row = base ^ (32 f): one base register, recomputed per piece (the asm
    // keeps the compiler from hoisting four per-lane offsets out of the strip loop - this kernel has no spare VGPRs)
    unsigned base = (unsigned)(OFF_A + slot * APIECE + row * 128 + ((half ^ sw) << 4));
    asm volatile("" : "+v"(base));
    const int kb = j >> 1;
    // five separately guarded copies with constant register indices (a switch over kb gets its stores sunk into one
    // dynamically indexed store by the compiler, which sends the array to scratch memory)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      if (kb == c) {
#pragma unroll
        for (int f = 0; f < 4; ++f) afn[4 * c + f] = *(const bf16x8*)(smem + (base ^ (unsigned)(32 * f)));
        asm volatile("" ::: "memory");
      }
    }
  };

  // ---- VMEM bookkeeping (all wave-uniform): seq = LOADS (LDS-DMA and residual loads; never stores, see the epilogue)
  // issued so far by this wave; a mark = seq right after the loads someone will wait for
  int seq = 0;
  int wm0 = 0, wm1 = 0;          // W pieces of the current and the next iteration
  int am0 = 0, am1 = 0, a_fly = 0;  // A pieces in flight (front, back)
  int rmark = 0;                 // residual rows of the coming tile

  // ---- the A stream: elements (strip ordinal o, piece j) in order; ordinal 0 is consumed by the prologue
  // (pseudo-iterations -10 .. -1, issued 2 earlier), ordinal o >= 1 during strip o - 1 at iteration
  // (o - 1) I + I - 10 cs + j cs, issued 2 cs earlier. All state advances incrementally (wave-uniform scalars: the
  // loop body is latency-critical and SALU divisions / the branches around them were ~150 instructions per iteration)
  int e_issue = 0, e_cons = 0;               // elements issued / consumed so far
  int is_o = 0, is_j = 0, is_gi = -NAP - 2;  // next element to issue: strip ordinal, piece, (pseudo-)iteration
  int is_slot = 0, is_row0 = (int)blockIdx.x * 256;
  int cn_o = 0, cn_j = 0, cn_gi = -NAP, cn_slot = 0;  // next element to consume
  // (updates are written with selects, not if / else pairs: the compiler sinks the two stores of such a pair into one
  // dynamically addressed store, which sends the whole state block to scratch memory and the loop to exec masking)
  auto a_step_issue = [&](int gi) {  // issue the element due at pseudo-iteration gi (at most one per call)
    if (is_o < my_count && is_gi <= gi) {
      if (!w_issuer) {
        issue_a(is_row0, is_j, is_slot);
        seq += 4;
        am0 = a_fly == 0 ? seq : am0;
        am1 = a_fly == 0 ? am1 : seq;
      }
      const bool wrap = is_j == NAP - 1;
      const int step_in = is_o == 0 ? 1 : cs, step_wrap = is_o == 0 ? I - 12 * cs + 3 : I - 9 * cs;
      ++a_fly;
      ++e_issue;
      is_slot = is_slot == ASLOTS - 1 ? 0 : is_slot + 1;
      is_gi += wrap ? step_wrap : step_in;
      is_j = wrap ? 0 : is_j + 1;
      is_o += wrap ? 1 : 0;
      is_row0 += wrap ? G * 256 : 0;
    }
  };
  auto a_due = [&](int gi) { return e_cons < e_issue && cn_gi <= gi; };
  int ce_j = 0, ce_slot = 0;  // the element being consumed in this iteration
  auto a_consume_step = [&]() {  // bookkeeping of one consumption (at most two pieces are ever in flight)
    const bool wrap = cn_j == NAP - 1;
    const int step_in = cn_o == 0 ? 1 : cs, step_wrap = cn_o == 0 ? I - 10 * cs + 1 : I - 9 * cs;
    ce_j = cn_j; ce_slot = cn_slot;
    am0 = am1; --a_fly; ++e_cons;
    cn_slot = cn_slot == ASLOTS - 1 ? 0 : cn_slot + 1;
    cn_gi += wrap ? step_wrap : step_in;
    cn_j = wrap ? 0 : cn_j + 1;
    cn_o += wrap ? 1 : 0;
  };

  // ---- prologue: W pieces 0 and 1; the first strip's A through the ring into `afn`, then afn -> af
  if (w_issuer) {
    issue_w(0, true, 0); seq += 5; wm0 = seq;
    issue_w(1, true, 1); seq += 5; wm1 = seq;
  }
  a_step_issue(-NAP - 2);
  a_step_issue(-NAP - 1);
  for (int gi = -NAP; gi < 0; ++gi) {
    const bool due = a_due(gi);
    if (due && !w_issuer) wait_vmcnt_le(seq - am0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (due) a_consume_step();
    a_step_issue(gi);
    if (due) take_a(ce_j, ce_slot);
  }
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) af[ks] = afn[ks];
  if (LNF) normalise_rows();

  f32x16 acc[2];
  // post-transpose lane mapping: pass ps covers rows (lane >> 2) + 16 ps, columns 8 (lane & 3) .. +8 of a 32-column
  // block. A wave's 32 rows are all inside M or all outside (M % 32 == 0): stores, residual loads and their counts sit
  // behind ONE wave-uniform test, and the row / tile part of every address travels in the scalar offset.
  u32x4 rres[4];  // residual rows of the coming tile: [column block][pass]
  auto issue_resid = [&](int row0, int t) {
    if (!RESID || row0 + wave * 32 >= p.M) return;
    int l2 = lane;
    asm volatile("" : "+v"(l2));  // lane-derived offsets are recomputed where they are used, not kept in registers
    const unsigned r_lane = (unsigned)(((wave * 32 + (l2 >> 2)) * p.ldr + (l2 & 3) * 8) * 2);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int soff = ((row0 + 16 * ps) * p.ldr + t * 64 + nb * 32) * 2;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen"
                     : "=v"(rres[nb * 2 + ps]) : "v"(r_lane), "s"(rs_r), "s"(soff) : "memory");
      }
    seq += 4;
    rmark = seq;
  };
  issue_resid((int)blockIdx.x * 256, 0);

  int gi = 0;       // global iteration (W piece) counter over all strips of this workgroup
  int wslot = 0;    // ring slot of the W piece of iteration gi
  for (int ord = 0; ord < my_count; ++ord) {
    const int row0 = ((int)blockIdx.x + ord * G) * 256;
    const bool has_next = ord + 1 < my_count;
    const bool rows_ok = row0 + wave * 32 < p.M;
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const int i = 2 * t + kh;
        // ---- this iteration's W piece (this wave's share) and, if one is due, the A piece have landed
        if (w_issuer) wait_vmcnt_le(seq - wm0);
        const bool due = a_due(gi);
        if (due && !w_issuer) wait_vmcnt_le(seq - am0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- refill: the W slot read in the previous iteration, the A slot copied out in an earlier one
        if (w_issuer) {
          const int nxt = i + 2;
          const int slot2 = wslot == 0 ? 2 : wslot - 1;  // (wslot + 2) % 3
          issue_w(nxt < I ? nxt : nxt - I, nxt < I || has_next, slot2);
          seq += 5;
          wm0 = wm1; wm1 = seq;
        }
        if (due) a_consume_step();
        a_step_issue(gi);
        // ---- 10 k slices x 2 column blocks. Fragment reads are pinned (the scheduler would otherwise hoist as many as
        // registers allow, and this kernel has none to spare): one slice ahead, or - residual variants, which also hold
        // 16 registers of residual rows - just in time, the SIMD's other wave covering the LDS latency.
        const char* wp = wring + wslot * WPIECE + lane * 16;
        if (RESID) {
#pragma unroll
          for (int ksl = 0; ksl < KSH; ++ksl) {
            const bf16x8 w0 = *(const bf16x8*)(wp + ksl * 1024), w1 = *(const bf16x8*)(wp + (KSH + ksl) * 1024);
            acc[0] = CD_MFMA_32x32x16(w0, af[kh * KSH + ksl], acc[0]);
            acc[1] = CD_MFMA_32x32x16(w1, af[kh * KSH + ksl], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          constexpr int PFD = 2;  // slices read ahead
          bf16x8 wf[PFD + 1][2];
#pragma unroll
          for (int d = 0; d < PFD; ++d) {
            wf[d][0] = *(const bf16x8*)(wp + d * 1024);
            wf[d][1] = *(const bf16x8*)(wp + (KSH + d) * 1024);
          }
#pragma unroll
          for (int ksl = 0; ksl < KSH; ++ksl) {
            if (ksl + PFD < KSH) {
              wf[(ksl + PFD) % (PFD + 1)][0] = *(const bf16x8*)(wp + (ksl + PFD) * 1024);
              wf[(ksl + PFD) % (PFD + 1)][1] = *(const bf16x8*)(wp + (KSH + ksl + PFD) * 1024);
            }
            acc[0] = CD_MFMA_32x32x16(wf[ksl % (PFD + 1)][0], af[kh * KSH + ksl], acc[0]);
            acc[1] = CD_MFMA_32x32x16(wf[ksl % (PFD + 1)][1], af[kh * KSH + ksl], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // the landed A piece goes to the next-strip registers after the MFMAs, when the fragment registers are free
        // (its ring slot is refilled one iteration later at the earliest, behind the next barrier's lgkmcnt(0))
        if (due) take_a(ce_j, ce_slot);
        wslot = wslot == 2 ? 0 : wslot + 1;
        ++gi;
      }
      // ---- epilogue of tile t: lane owns row mi of the wave's 32, columns (r & 3) + 8 (r >> 2) + 4 half of each block
      if (RESID && rows_ok) {
        wait_vmcnt_le(seq - rmark);
        touch4(rres[0], rres[1], rres[2], rres[3]);
      }
      float vv[2][8];  // GEGLU: the value half waits for its gate
      int l2 = lane;
      asm volatile("" : "+v"(l2));  // as above: nothing lane-derived stays live across the MFMA loop
      const int prow = l2 >> 2, pcol = (l2 & 3) * 8, mi2 = l2 & 31, half2 = l2 >> 5;
      const unsigned o_lane = (unsigned)(((wave * 32 + prow) * p.ldo + pcol) * 2);
      if (GEGLU && p.geglu_direct) {
        // GEGLU straight from the accumulators: value (acc[0]) and gate (acc[1]) of an output element sit in the same
        // register of the same lane, so the tile needs no transpose - the LDS round trip of the staged form (4 writes,
        // a wait, 4 reads per column block, each behind the other) is this epilogue's critical path, not its VALU count
        // (round 6: running the SIMD's two waves' epilogues one after the other instead of against each other LOST 5 %).
        // A lane stores 4 consecutive columns of its row per register quad (8 bytes; lanes l and l + 32 are neighbours).
        // Same operations per element as the staged form (bias add, gelu_fast2 on column pairs, product, one rounding).
        const unsigned od_lane = (unsigned)(((wave * 32 + mi2) * p.ldo + 4 * half2) * 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float* bp = bias_s + t * 64 + 8 * q + 4 * half2;
          const f32x4 bv = *(const f32x4*)bp, bg = *(const f32x4*)(bp + 32);
          float val[4], gate[4], o4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { val[j] = acc[0][4 * q + j] + bv[j]; gate[j] = acc[1][4 * q + j] + bg[j]; }
          mul_gelu4(val, gate, o4);
          if (rows_ok) {
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
            const u32x2 o = {pack2(o4[0], o4[1]), pack2(o4[2], o4[3])};
            __builtin_amdgcn_raw_buffer_store_b64(o, rs_o, od_lane, (row0 * p.ldo + t * 32 + 8 * q) * 2, 0);
          }
          __builtin_amdgcn_sched_barrier(0);  // one register quad at a time
        }
      } else {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v4 = {acc[nb][4 * q], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]};
          *(f32x4*)(stage + mi2 * ST_LD + 8 * q + 4 * half2) = v4;
        }
        // the read-back lands in the registers these stores take their data from: retire them first (see the
        // statistics write-back below for what happens otherwise)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int ncol0 = t * 64 + nb * 32;  // first packed column of this block
        const f32x4 b0 = *(const f32x4*)(bias_s + ncol0 + pcol), b1 = *(const f32x4*)(bias_s + ncol0 + pcol + 4);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          float* srow = stage + (prow + 16 * ps) * ST_LD + pcol;
          const f32x4 lo = *(const f32x4*)srow, hi = *(const f32x4*)(srow + 4);
          float v[8] = {lo[0] + b0[0], lo[1] + b0[1], lo[2] + b0[2], lo[3] + b0[3],
                        hi[0] + b1[0], hi[1] + b1[1], hi[2] + b1[2], hi[3] + b1[3]};
          if (GEGLU) {
            if (nb == 0) {
#pragma unroll
              for (int e = 0; e < 8; ++e) vv[ps][e] = v[e];
            } else {
              mul_gelu8(vv[ps], v, v);
              if (rows_ok) {
                const uint4 o = pack8_sat(v);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_o, o_lane,
                                                       ((row0 + 16 * ps) * p.ldo + t * 32) * 2, 0);
              }
            }
          } else {
            if (RESID && rows_ok) {
              float rr[8];
              unpack8(__builtin_bit_cast(uint4, rres[nb * 2 + ps]), rr);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += rr[e];
            }
            if (rows_ok) {
              const uint4 o = pack8_sat(v);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_o, o_lane,
                                                     ((row0 + 16 * ps) * p.ldo + ncol0) * 2, 0);
            }
            if (STATS) {  // the final fp32 values go back to the transpose buffer for the column sums below
              *(f32x4*)srow = (f32x4){v[0], v[1], v[2], v[3]};
              *(f32x4*)(srow + 4) = (f32x4){v[4], v[5], v[6], v[7]};
              // retire these stores before their source registers are reused: on hardware, without this wait, a few
              // waves per launch stored a wrong third dword in the NEXT pass (lanes 12-15 of every 16, this column
              // block only) although the instruction stream is correct - the following ds_read_b128 lands in the
              // registers the pending ds_write_b128 takes its data from (profiles/r3_lds_store_hazard_bisection.txt)
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
          }
          __builtin_amdgcn_sched_barrier(0);  // one (column block, pass) at a time: the temporaries must not pile up
        }
        if (STATS) {
          // per-channel sum / sum of squares over the wave's 32 rows (= one 32-row statistics block of
          // ConvGemmParams::stats): lane l sums column l & 31 over rows 16 (l >> 5) .. +16, the halves meet by DPP
          __builtin_amdgcn_wave_barrier();
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float x = stage[(16 * half2 + r) * ST_LD + mi2];
            s1 += x; s2 += x * x;
          }
          s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
          if (rows_ok && half2 == 0) {
            float* sp = p.stats + (int64_t)((row0 + wave * 32) >> 5) * 2 * p.N + ncol0 + mi2;
            sp[0] = s1; sp[p.N] = s2;
          }
        }
      }
      }  // staged epilogue
      // Stores are NOT counted in `seq`: gfx9-family hardware may retire stores out of order with respect to loads
      // (only loads return in order among themselves), so a count that allowed "the younger stores" to be outstanding
      // could be satisfied by early stores while the awaited load is still in flight - seen on hardware as stale A
      // pieces once a workgroup runs a second strip. Counting LOADS only makes every wait also cover older stores.
      // residual rows of the next tile (of the next strip after the last tile): a whole tile of lead
      if (RESID) {  // ONE issue site inside the loop: every asm statement defines its own set of result registers
        const bool last = t + 1 == NT;
        if (!last || has_next) issue_resid(last ? row0 + G * 256 : row0, last ? 0 : t + 1);
      }
    }
    if (has_next) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) af[ks] = afn[ks];
      if (LNF) normalise_rows();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dead tail pieces
#endif
}

// standard packed weights [N][K] (row n contiguous in k) -> fragment-major blocks [N / 32][K / 16][64 lanes][8]:
// lane l of block (nb, ks) holds W[32 nb + (l & 31)][16 ks + 8 (l >> 5) .. + 8]
__global__ void k_pack_wfrag(const bf16_t* __restrict__ w, int ldw, bf16_t* __restrict__ out, int N) {
  const int64_t total = (int64_t)(N / 32) * NKS * 64;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    const int ks = (int)((i >> 6) % NKS), nb = (int)((i >> 6) / NKS);
    const uint4 v = *(const uint4*)(w + (int64_t)(32 * nb + (l & 31)) * ldw + 16 * ks + 8 * (l >> 5));
    *(uint4*)(out + i * 8) = v;
  }
}

// LayerNorm gain / bias folded into the following linear layer: w_out[n][k] = w[n][k] gamma[k] (one more rounding to
// 16 bits), bias_out[n] = bias[n] + sum_k w[n][k] beta[k] (fp32). One wave per output row.
__global__ void k_fold_ln(const bf16_t* __restrict__ w, int ldw, const float* __restrict__ gamma,
                          const float* __restrict__ beta, const float* __restrict__ bias, bf16_t* __restrict__ w_out,
                          float* __restrict__ bias_out, int N, int Kc) {
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float acc = 0.f;
  for (int k = lane; k < Kc; k += 64) {
    const float x = bf2f(w[(int64_t)n * ldw + k]);
    acc = __builtin_fmaf(x, beta[k], acc);
    w_out[(int64_t)n * ldw + k] = f2bf(x * gamma[k]);
  }
  acc = wave_allsum(acc);
  if (lane == 0) bias_out[n] = (bias ? bias[n] : 0.f) + acc;
}

template <int ACT, bool RESID, bool STATS, bool LNF = false>
void launch_variant(hipStream_t st, const LinStreamParams& p, int grid) {
  static PerDeviceOnce attr_once;
  auto kern = k_lin_stream<ACT, RESID, STATS, LNF>;
  attr_once([&]() {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  });
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, st, p);
}

}  // namespace lin_detail

bool lin_stream_supports(const ConvGemmParams& p) {
  if (p.KH != 1 || p.KW != 1 || p.stride != 1 || p.up || p.src1 || p.C1 != 0) return false;
  if (p.Ktot != lin_detail::K || p.C0 != lin_detail::K || p.nbatch != 1 || p.out_f32 || p.rowvec) return false;
  if (p.alpha != 1.0f || p.splitk > 1 || !p.wgt_frag) return false;
  if (p.N % 64 != 0 || p.N > lin_detail::BIAS_MAX || p.N < 320) return false;
  if (p.act != ACT_NONE && p.act != ACT_GEGLU) return false;
  if (p.act == ACT_GEGLU && (p.resid || p.stats)) return false;
  if (p.M % 32 != 0) return false;
  if ((p.ld0 % 8) != 0 || (p.out_ld % 8) != 0 || (p.resid && (p.resid_ld % 8) != 0)) return false;
  const int nout = p.act == ACT_GEGLU ? p.N / 2 : p.N;
  // 32-bit byte offsets inside the buffer descriptors
  if ((int64_t)p.M * p.ld0 * 2 >= (1ll << 31) || (int64_t)p.M * p.out_ld * 2 >= (1ll << 31)) return false;
  if (p.resid && (int64_t)p.M * p.resid_ld * 2 >= (1ll << 31)) return false;
  (void)nout;
  return true;
}

void launch_lin_stream(hipStream_t st, const ConvGemmParams& c) {
  using namespace lin_detail;
  CD_CHECK(lin_stream_supports(c), "lin_stream: unsupported problem (M %d N %d K %d)", c.M, c.N, c.Ktot);
  LinStreamParams p;
  p.a = c.src0; p.lda = c.ld0; p.wfrag = c.wgt_frag; p.bias = c.bias;
  p.resid = c.resid; p.ldr = c.resid_ld; p.out = (bf16_t*)c.out; p.ldo = c.out_ld;
  p.stats = c.stats; p.M = c.M; p.N = c.N;
  static std::atomic<int> ncu_of[64];  // CU count per device (one persistent workgroup per CU)
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  int ncu = ncu_of[dev & 63].load(std::memory_order_relaxed);
  if (!ncu) {
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ncu_of[dev & 63].store(ncu, std::memory_order_relaxed);
  }
  const int nstrips = (p.M + 255) / 256;
  const int grid = nstrips < ncu ? nstrips : ncu;
  // CYCLEDIFF_GEGLU_DIRECT=0 / 1 (A/B runs): the staged / the straight-from-the-accumulators GEGLU epilogue
  static const int geglu_direct = [] { const char* e = getenv("CYCLEDIFF_GEGLU_DIRECT"); return e && e[0] ? atoi(e) : kGegluDirectDefault; }();
  p.geglu_direct = geglu_direct;
  if (c.ln_fold) {  // LayerNorm folded in: the feed-forward GEGLU projection and the plain projections behind norm1-3
    p.ln_eps = c.ln_eps;
    CD_CHECK(!c.resid && !c.stats, "lin_stream: a LayerNorm-folded layer has neither residual nor statistics");
    if (c.act == ACT_GEGLU) launch_variant<ACT_GEGLU, false, false, true>(st, p, grid);
    else launch_variant<ACT_NONE, false, false, true>(st, p, grid);
    return;
  }
  if (c.act == ACT_GEGLU) launch_variant<ACT_GEGLU, false, false>(st, p, grid);
  else if (c.resid && c.stats) launch_variant<ACT_NONE, true, true>(st, p, grid);
  else if (c.resid) launch_variant<ACT_NONE, true, false>(st, p, grid);
  else if (c.stats) launch_variant<ACT_NONE, false, true>(st, p, grid);
  else launch_variant<ACT_NONE, false, false>(st, p, grid);
}

void launch_fold_ln(hipStream_t st, const bf16_t* w, int ldw, const float* gamma, const float* beta, const float* bias,
                    bf16_t* w_out, float* bias_out, int N, int Kc) {
  hipLaunchKernelGGL(lin_detail::k_fold_ln, dim3((N + 3) / 4), dim3(256), 0, st, w, ldw, gamma, beta, bias, w_out,
                     bias_out, N, Kc);
}

void launch_pack_wfrag(hipStream_t st, const bf16_t* w, int ldw, bf16_t* out, int N) {
  CD_CHECK(N % 32 == 0 && ldw >= lin_detail::K, "pack_wfrag: N %% 32, ldw >= 320");
  const int64_t total = (int64_t)(N / 32) * lin_detail::NKS * 64;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(lin_detail::k_pack_wfrag, dim3(grid), dim3(256), 0, st, w, ldw, out, N);
}

}  // namespace cd
