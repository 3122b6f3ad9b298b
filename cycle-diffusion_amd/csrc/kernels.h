// Launcher declarations for the hand-written gfx950 kernels (implementation: *.hip in this dir).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace cd {

// ---------------------------------------------------------------- scheduler (sched.hip)
// Per-step scalar coefficients, evaluated on the host in fp32 in the reference's operation
// order (ddim.py:570-579). DDIM-eta form:
//   sa=sqrt(a_t) s1a=sqrt(1-a_t) sap=sqrt(a_prev) dirc=sqrt(1-a_prev-sigma^2) sigma r=sqrt(1-a_t)
// DDPM form (pixel 'ddpm'): see sched.hip k_encode_step_ddpm.
struct StepCoef {
  float sa, s1a, sap, dirc, sigma, r, t_mask;
  int t;  // integer timestep fed to the U-Net
};
enum { SCHED_DDIM = 0, SCHED_DDPM = 1 };

struct EpsHat {  // network output view: element (b,c,p) at p[b*sb + c*sc + p*sp]
  const float* p;
  int64_t sb, sc, sp;
  int cfg;   // 1: batch is [uncond B | cond B], combine with guidance scale g
  float g;
  const float* gvec = nullptr;  // per-sample guidance scales [B] (device) instead of g: ensemble members that differ only
                                // in their decoder scale share one launch set
};

void launch_init_xt(hipStream_t st, const float* x0, const float* noise, uint64_t seed,
                    uint32_t stream, float* xt, float* z, int64_t z_bstride, int B, int C, int HW,
                    const StepCoef* tab, int step, bf16_t* xin, int xin_cpad, int cfg_dup);
void launch_encode_step(hipStream_t st, int kind, const float* x0, float* xt, const EpsHat& eh,
                        const float* noise, uint64_t seed, uint32_t stream, float* z,
                        int64_t z_bstride, int B, int C, int HW, const StepCoef* tab,
                        const int* step_ptr, int step, int is_last, bf16_t* xin, int xin_cpad,
                        int cfg_dup_next);
void launch_decode_step(hipStream_t st, int kind, float* x, const EpsHat& eh, const float* eps,
                        int64_t eps_bstride, const float* noise, uint64_t seed, uint32_t stream,
                        int B, int C, int HW, const StepCoef* tab, const int* step_ptr, int step,
                        bf16_t* xin, int xin_cpad, int cfg_dup_next, float* x0_pred, int eps_bmod = 0);
// eps_bmod > 0: sample b takes its injected eps from slot sample b % eps_bmod (the coupled loop decodes several guidance scales
// of the same eps_bmod encoder samples in one batch)
void launch_set_int(hipStream_t st, int* p, int v);
void launch_add_int(hipStream_t st, int* p, int d);

// ---------------------------------------------------------------- implicit-GEMM conv / GEMM (conv_gemm.hip)
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2, ACT_GEGLU = 3, ACT_QGELU = 4 };  // QGELU: x*sigmoid(1.702x) (CLIP)

struct ConvGemmParams {
  // A operand: activations, NHWC bf16; optional second source = channel concat (th.cat, openaimodel.py:736)
  const bf16_t* src0 = nullptr;
  const bf16_t* src1 = nullptr;
  int C0 = 0, C1 = 0;        // channels taken from each source (multiples of 32)
  int ld0 = 0, ld1 = 0;      // pixel stride in elements
  int B = 1, Hs = 1, Ws = 1; // stored spatial dims of the sources
  int up = 0;                // 1: nearest x2 upsample folded into the gather (F.interpolate, openaimodel.py:115)
  int Hin = 1, Win = 1;      // logical input dims (= 2*Hs when up)
  int KH = 1, KW = 1, stride = 1, pad_t = 0, pad_l = 0;
  int Hout = 1, Wout = 1;
  int M = 0;                 // B*Hout*Wout
  // B operand: weights [Npad][Ktot] bf16, k = (r*KW+s)*(C0+C1) + c
  const bf16_t* wgt = nullptr;
  const bf16_t* wgt_frag = nullptr;  // the same weights in MFMA-fragment-major order (lin_stream.hip), or null
  int Ktot = 0, N = 0;
  int ldw = 0;               // weight row stride in elements (0 = Ktot)
  // batched GEMM (grid.z): element strides
  int nbatch = 1;
  int64_t a_bs = 0, w_bs = 0, o_bs = 0;
  // epilogue: out = act(alpha*acc + bias[n] + rowvec[m/rows_per_vec][n]) + resid[m][n]
  float alpha = 1.0f;
  const float* bias = nullptr;
  const float* rowvec = nullptr;
  int rowvec_ld = 0, rows_per_vec = 1;
  const bf16_t* resid = nullptr;
  int resid_ld = 0;
  int resid_f32 = 0;         // the residual is fp32 (the split-fp16 mode of the fp32 path, f32_path.hip)
  int act = ACT_NONE;
  void* out = nullptr;
  int out_ld = 0;
  int out_f32 = 0;
  // optional fused GroupNorm statistics of the output: [nbatch][ceil(M/32)][2][N] fp32 (sum | sum of squares
  // per channel over each 32-row block); null = off. Not available with GEGLU.
  float* stats = nullptr;
  const bf16_t* zeros = nullptr;  // >= 256 B of zeros (masked rows / padding taps)
  // the rows of A are LayerNorm-ed (no gain / bias: those are folded into wgt / bias) inside the kernel; lin_stream only
  int ln_fold = 0;
  float ln_eps = 1e-5f;
  int tile = 0;                   // 0 = auto (tile configuration AND split factor from the autotuner)
  float prof_flop_scale = 1.f;    // KernelProfiler books 2*M*N*Ktot times this (1/3 for the three-term split GEMMs)
  // split-K (deep-K layers whose output tiles cannot fill 256 CUs): K is cut in `splitk` ranges, fp32
  // partial tiles meet in sk_scratch and the last arrival sums them in split order. 0/1 = off.
  int splitk = 0;
  float* sk_scratch = nullptr;    // defaults to g_conv_splitk when splitk > 1
  int* sk_flags = nullptr;
  // phase-timing build only (-DCD_PROBE, lib/libcyclediff_probe.so; scripts/probe_report.py): every wave leaves
  // kProbeWords 64-bit words of s_memtime stamps here, [block][wave][kProbeWords]; null = off
  unsigned long long* probe = nullptr;
  // order of the K steps of a KH x KW > 1 convolution: tap-major (all channels of a filter tap, then the next tap) or
  // channel-major (the KH * KW taps of one BK-channel slice, then the next slice: the re-reads of an activation line by
  // neighbouring taps follow each other within KH * KW steps and hit the XCD's L2 - round 4: hit rate 54 -> 72 %, fabric fetch
  // of the 320-channel 3 x 3 conv at 64 x 64 750 -> 409 MB per launch). 1 (default) = channel-major where the activation is
  // large (launch_cfg), 2 = wherever the kernel supports it, 0 = never (CYCLEDIFF_KORDER for A/B runs)
  int korder = 1;
  // tile walk inside an XCD's contiguous range: 0 = row-major in the operand the launcher picked (m-major / n-major);
  // G > 0 = groups of G row tiles walked m-fastest (the 32 CUs of an XCD then run G row tiles x 32 / G column tiles at a time:
  // G A panels + 32 / G W panels in its L2 instead of 1 + 32) - for the wide-N layers whose operands both exceed the L2
  int tile_group = 0;
  // two-per-CU tile configurations (24, 25) only: the workgroups of the launch's first wave front that sit in an odd wave slot
  // wait this many shader cycles before they start, so that the two co-resident workgroups of a CU run half a tile apart (one's
  // prologue / epilogue under the other's K loop). 0 = off. Experiment of round 6 (CYCLEDIFF_DEPHASE_TICKS), docs/optimisation_log.md
  int dephase_ticks = 0;
  int num_cus = 256;
  int dbg = 0;  // probe build: 1 = the epilogue skips its global stores, 2 = skips the statistics, 4 = every tile gathers
                // its A rows from the first 1024 + BM rows (an L2-resident operand: what would the K loop do without misses?)
};
constexpr int kProbeWords = 48;
extern thread_local unsigned long long* g_conv_probe;  // picked up by launch_conv_gemm in the probe build
struct SplitKWorkspace {
  float* scratch = nullptr;
  size_t scratch_bytes = 0;
  int* flags = nullptr;  // zero-initialised arrival counters, one per output tile
  int nflags = 0;
};
extern thread_local SplitKWorkspace g_conv_splitk;  // the calling thread's engine (capi.hip enter_engine)
void launch_conv_gemm(hipStream_t st, const ConvGemmParams& p);
const char* conv_gemm_last_config();
int conv_gemm_num_configs();
const char* conv_gemm_config_name(int id);

// ---------------------------------------------------------------- streaming linear layer, K = 320 (lin_stream.hip)
// out[M][N] = epi(A[M][320] . W[N][320]^T): A strips resident in registers and prefetched a strip ahead, W streamed
// through LDS in fragment-major order, persistent workgroups. Part of the implicit-GEMM family: launch_conv_gemm
// dispatches to it (tile id kLinStreamTile) for the shapes lin_stream_supports() accepts.
struct LinStreamParams {
  const bf16_t* a = nullptr; int lda = 0;
  const bf16_t* wfrag = nullptr;
  const float* bias = nullptr;
  const bf16_t* resid = nullptr; int ldr = 0;
  bf16_t* out = nullptr; int ldo = 0;
  float* stats = nullptr;
  int M = 0, N = 0;
  float ln_eps = 1e-5f;  // LayerNorm-folded variants
  int geglu_direct = 0;  // GEGLU epilogue straight from the accumulators (no LDS transpose), lin_stream.hip
};
constexpr int kLinStreamTile = 30;
bool lin_stream_supports(const ConvGemmParams& p);
void launch_lin_stream(hipStream_t st, const ConvGemmParams& p);
// w_out = w . diag(gamma) (16-bit), bias_out = bias + w . beta (fp32): the weights of a LayerNorm-folded layer
void launch_fold_ln(hipStream_t st, const bf16_t* w, int ldw, const float* gamma, const float* beta, const float* bias,
                    bf16_t* w_out, float* bias_out, int N, int Kc);
// standard packed rows [N][ldw] (K = 320 used) -> fragment-major blocks [N / 32][20][64][8]
void launch_pack_wfrag(hipStream_t st, const bf16_t* w, int ldw, bf16_t* out, int N);

// Optional per-launch timing of the dominant kernel family with HIP events recorded on the launch
// stream (bench.py's roofline leg). flops = 2*M*N*K per launch (algorithmic, padding excluded except
// the 3/4 -> 32 input-channel pad of conv_in).
struct KernelProfiler {
  bool enabled = false;
  std::vector<hipEvent_t> events;
  std::vector<double> flops;
  std::vector<std::string> labels;
  int used = 0;
  bool verbose = false;  // CYCLEDIFF_GEMM_LOG=1: collect() prints a per-shape table to stderr
  struct Entry { int n = 0; double ms = 0, flops = 0; };
  std::map<std::string, Entry> per_shape;
  void next_pair(hipEvent_t* a, hipEvent_t* b, double fl, const std::string& what);
  void collect(int* launches, double* total_ms, double* total_flops);
  ~KernelProfiler();
};
extern thread_local KernelProfiler* g_conv_prof;

// online tile-configuration autotuner of the implicit-GEMM kernel (conv_gemm.hip)
struct ConvTuner {
  bool enabled = false;
  void* scratch = nullptr;   // device scratch for redirected outputs while timing
  size_t scratch_bytes = 0;
  int shapes_tuned = 0;
};
extern ConvTuner g_conv_tuner;

// weight repack: fp32 [N][Cin][KH][KW] (torch conv / linear with KH=KW=1) -> bf16 [Npad][KH*KW*Cpad]
// with zero padding; `geglu` interleaves value/gate rows in blocks of 32 (see conv_gemm.hip).
void launch_repack_weight(hipStream_t st, const float* w, bf16_t* out, int N, int Cin, int KH,
                          int KW, int Npad, int Cpad, int geglu, int64_t src_row_offset);

// ---------------------------------------------------------------- normalisation / softmax (norm.hip)
struct GroupNormParams {
  const bf16_t* x = nullptr;   // NHWC bf16 [B][HW][C] (optionally a channel concat of two tensors)
  const bf16_t* x1 = nullptr;
  int C0 = 0, C1 = 0, ld0 = 0, ld1 = 0;
  int B = 0, HW = 0, G = 32;
  float eps = 1e-5f;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  // FiLM (use_scale_shift_norm, improved_ddpm/unet.py:253-257): y = gn(x)*(1+scale[b][c]) + shift[b][c]
  const float* film = nullptr;  // [B][2*C]: scale | shift
  int film_ld = 0;
  int silu = 0;
  int split_out = 0;            // fp32 path only: y is the fp16 pair [B][HW][hi(C) | lo(C)] (CD_PREC_F32X3)
  int* overflow = nullptr;      // ... set to 1 by any value outside the fp16 range after scaling (host-visible word)
  bf16_t* y = nullptr;          // [B][HW][C] dense
  float* partial = nullptr;     // workspace [B][S][G][2]
  int S = 0;
  float* coef = nullptr;        // workspace [B][2][C]: per-channel multiply-add coefficients
  // statistics already produced by the conv epilogue that wrote x / x1 (ConvGemmParams::stats layout,
  // [B*HW/32][2][C]): when set, the stats pass is replaced by a small per-(image, group) fold
  const float* pre0 = nullptr;
  const float* pre1 = nullptr;
};
int groupnorm_slabs(int B, int HW, int C);
void launch_groupnorm(hipStream_t st, const GroupNormParams& p);

void launch_layernorm(hipStream_t st, const bf16_t* x, int ldx, bf16_t* y, int ldy, int rows,
                      int C, const float* gamma, const float* beta, float eps);
// row softmax over fp32 scores [rows][cols] -> bf16 probabilities (VAE AttnBlock, model.py:193)
void launch_softmax_rows(hipStream_t st, const float* s, int lds, bf16_t* p, int ldp, int64_t rows,
                         int cols);

// ---------------------------------------------------------------- attention (attn.hip)
struct AttnParams {
  const bf16_t* q = nullptr;   // [B][Tq][ldq], head h at column h*D
  const bf16_t* k = nullptr;   // [B][Tk][ldk]
  // values, exactly one of:
  const bf16_t* v = nullptr;   //   token-major [B][Tk][ldv], head h at column h*D (fused q|k|v projection output)
  const bf16_t* vt = nullptr;  //   transposed: [B][H][Dv_pad][Tk_pad]  (keys contiguous)
  int ldv = 0; int64_t v_bs = 0;
  bf16_t* o = nullptr;         // [B][Tq][ldo]
  int B = 0, H = 0, Tq = 0, Tk = 0, D = 0;
  int ldq = 0, ldk = 0, ldo = 0;
  int64_t q_bs = 0, k_bs = 0, o_bs = 0;
  int vt_dpad = 0, vt_tpad = 0;
  float scale = 1.0f;            // softmax scale; ignored when q_log2 is set
  int q_log2 = 0;                // 1: q already carries scale * log2(e) (folded into the packed to_q weights)
  const float* obias = nullptr;  // [H*D] added to the output (value-projection bias: rows of P sum to 1)
  int causal = 0;                // 1: key j is visible to query i only if j <= i (CLIP text transformer)
};
void launch_attention(hipStream_t st, const AttnParams& p);
// V [B][Tk][ldv] (head h at column h*D) -> Vt [B][H][Dpad][Tpad], zero padded
void launch_transpose_v(hipStream_t st, const bf16_t* v, int ldv, int64_t v_bs, bf16_t* vt, int B,
                        int H, int Tk, int D, int Dpad, int Tpad);

// ---------------------------------------------------------------- diagnostics (diag.hip)
// sustained 16-bit MFMA rate of this device under its power cap: bare MFMA loop on every CU for ~target_ms
void launch_mfma_sustained(hipStream_t st, int target_ms, float* tflops, float* ghz);

// ---------------------------------------------------------------- elementwise (elementwise.hip)
// fp32 NCHW [B][C][HW] -> bf16 NHWC [B][HW][Cpad] (pad channels zero), y = x*scale + shift
void launch_nchw_to_nhwc(hipStream_t st, const float* x, bf16_t* y, int B, int C, int HW, int Cpad,
                         float scale, float shift, int dup);
// fp32/bf16 NHWC (ld) -> fp32 NCHW, y = x*scale + shift
void launch_nhwc_to_nchw(hipStream_t st, const void* x, int x_f32, int ldx, float* y, int B, int C,
                         int HW, float scale, float shift);
// sinusoidal timestep embedding -> fp32 [B][dim]; mode 0: [cos|sin], freq = exp(-ln(1e4)*k/half)
// (util.py:152-172); mode 1: [sin|cos], divisor half-1 (ddpm/diffusion.py:6-24)
void launch_timestep_embedding(hipStream_t st, const StepCoef* tab, const int* step_ptr, int step,
                               const float* t_explicit, float* out, int B, int dim, int mode);
// small dense layer on fp32 vectors: y[b][n] = act_out(sum_k act_in(x[b][k])*W[n][k] + bias[n]); W bf16 or f32
void launch_vec_linear(hipStream_t st, const float* x, int ldx, const float* W, const float* bias,
                       float* y, int ldy, int B, int K, int N, int silu_in, int silu_out);
// 2x2 average pool on NHWC bf16 (improved_ddpm Downsample without conv, unet.py:132-135)
void launch_avgpool2(hipStream_t st, const bf16_t* x, bf16_t* y, int B, int H, int W, int C);
// nearest x2 upsample (materialised; only where it cannot be folded into a conv)
void launch_upsample2(hipStream_t st, const bf16_t* x, bf16_t* y, int B, int H, int W, int C);
// DiagonalGaussianDistribution sample (distributions.py:24-37) from moments NHWC fp32 [B][HW][2*zc]:
// z = (mean + exp(0.5*clamp(logvar,-30,20))*noise) * scale -> fp32 NCHW [B][zc][HW]
void launch_posterior_sample(hipStream_t st, const float* mom, int ld, const float* noise,
                             uint64_t seed, float* z, int B, int zc, int HW, float scale,
                             int use_mean);
// VectorQuantizer2.forward of taming-transformers (git master; not vendored by the reference, README.md:85-91) as
// VQModelInterface.decode calls it (autoencoder.py:274-282): per latent vector z (fp32 NCHW * in_mul) the nearest
// codebook row by d = (|z|^2 + |e|^2) - 2 z.e (first minimum), output z + (e - z) -> 16-bit NHWC padded to Cpad
void launch_vq_quantize(hipStream_t st, const float* z, float in_mul, const float* codebook, int n_embed, int zc,
                        int B, int HW, bf16_t* out, int Cpad);
void launch_fill_f32(hipStream_t st, float* p, float v, int64_t n);
// ViT token assembly (OpenAI CLIP VisionTransformer.forward): out[b][0] = cls + pos[0]; out[b][1+t] = patch[b][t] + pos[1+t]
void launch_vit_tokens(hipStream_t st, const bf16_t* patch, const float* cls, const float* pos, bf16_t* out, int B,
                       int T, int D);
// out[b][:] = x[b][argmax_l ids[b][l]][:]  (CLIP text pooling at the end-of-text token, the largest id)
void launch_gather_eot(hipStream_t st, const bf16_t* x, const int* ids, bf16_t* out, int B, int L, int D);
// token + position embedding lookup: out[b][l][:] = tok[ids[b][l]][:] + pos[l][:] (fp32 tables -> 16-bit rows)
void launch_embed_tokens(hipStream_t st, const int* ids, const float* tok, const float* pos, bf16_t* out, int B,
                         int L, int D, int vocab);
// ---------------------------------------------------------------- fp32 execution path (f32_path.hip)
// The same parameter structs with every activation / weight / output pointer holding fp32 data and every `ld` counted
// in floats (precision CD_PREC_F32: pixel-space DDPMs). Fixed tile shapes, no autotuner, no split-K.
void launch_conv_gemm_f32(hipStream_t st, const ConvGemmParams& p);
size_t groupnorm_f32_workspace(int B, int HW, int C);
void launch_groupnorm_f32(hipStream_t st, const GroupNormParams& p, void* workspace);
// softmax(scale * q k^T) v + obias on token-major fp32 tensors, head h at column h*D
void launch_attention_f32(hipStream_t st, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                          float* o, int ldo, int B, int H, int T, int D, float scale, const float* obias);
// split = 1: the fp16 pair form [pixel][hi(Cpad) | lo(Cpad)] of CD_PREC_F32X3 (same bytes), range guard in `overflow`
void launch_nchw_to_nhwc_f32(hipStream_t st, const float* x, float* y, int B, int C, int HW, int Cpad, float scale,
                             float shift, int split = 0, int* overflow = nullptr);
void launch_avgpool2_f32(hipStream_t st, const float* x, float* y, int B, int H, int W, int C);
// fp32 SpatialTransformer pieces (st_f32.hip): flash attention on fp32 MFMA (q, k, v token-major [B][T][ld] with the head
// at column h * D; qmul multiplies q - scale * log2 e, or 1 when the to_q rows already carry it), LayerNorm (optionally
// written as the split mode's fp16 pair), GEGLU on a materialised [rows][2 * Nout] projection in packed column order
void launch_flash_f32(hipStream_t st, const float* q, int ldq, int64_t q_bs, const float* k, int ldk, int64_t k_bs,
                      const float* v, int ldv, int64_t v_bs, float* o, int ldo, int64_t o_bs, int B, int H, int Tq, int Tk,
                      int D, float qmul, int split, int* overflow, const float* obias = nullptr);
void launch_layernorm_f32(hipStream_t st, const float* x, int ldx, float* y, int64_t rows, int C, const float* gamma,
                          const float* beta, float eps, int split, int* overflow);
void launch_geglu_f32(hipStream_t st, const float* h, float* y, int64_t rows, int Nout, int split, int* overflow);
void launch_split_rows_f32(hipStream_t st, const float* x0, int ld0, int C0, const float* x1, int ld1, int C1, bf16_t* y,
                           int64_t rows, int* overflow);
void launch_upsample2_f32(hipStream_t st, const float* x, float* y, int B, int H, int W, int C);
// Split-fp16 mode of the fp32 path (precision CD_PREC_F32X3): a GroupNorm-ed activation x is kept as the fp16 pair
// T = [hi | lo] per pixel, hi = fp16(s_a x), lo = fp16(s_a x - hi), and a weight as [wh | wh | wl] per filter tap
// (scaled by s_w), so that one 16-bit implicit GEMM over the channel list [hi | lo | hi] with alpha = 1 / (s_a s_w)
// yields x.w to 2^-22 (the lo.wl term is dropped). Exact powers of two; fp16 builds only.
constexpr float kX3ActScale = 16.f;    // |x| < 4095 representable, full 22 bits down to |x| = 2^-6, 2^-28 absolute below
constexpr float kX3WgtScale = 256.f;   // |w| < 255, full 22 bits down to |w| = 2^-10
// fp32 packed weights [Npad][taps][Cpad] -> 16-bit [Npad][taps][3 * Cpad]
// `overflow` (may be null): a host-visible word the kernels set to 1 when a scaled value leaves the fp16 range - the
// engine turns it into an error at its next synchronisation point instead of returning saturated results.
void launch_pack_w3(hipStream_t st, const float* w, bf16_t* w3, int Npad, int taps, int Cpad, int* overflow);
// 2 x 2 average pool of a split activation [B][H][W][2C] -> [B][H/2][W/2][2C]
void launch_avgpool2_split(hipStream_t st, const bf16_t* x, bf16_t* y, int B, int H, int W, int C, int* overflow);

void launch_copy_strided_bf16(hipStream_t st, const bf16_t* src, int lds, bf16_t* dst, int ldd,
                              int64_t rows, int cols);

}  // namespace cd
