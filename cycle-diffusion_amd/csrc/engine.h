// Engine core: device arena, parameter store keyed by the reference's state_dict names, and the
// network executors (static launch schedules over the kernels in kernels.h).
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "../../include/cyclediff.h"

namespace cd {

// ------------------------------------------------------------------ device memory
// Stack-discipline bump allocator over one slab (sized for 288 GB parts: activations of a whole
// forward pass stay resident; temporaries of a block are released at block exit so the hot
// working set keeps re-using the same lines of the 256 MiB Infinity Cache).
class Arena {
 public:
  ~Arena();
  void init(size_t bytes);
  void* alloc(size_t bytes);
  size_t mark() const { return off_; }
  void release(size_t m) { off_ = m; }
  void reset() { off_ = 0; }
  size_t capacity() const { return cap_; }
  size_t high_water() const { return high_; }
 private:
  char* base_ = nullptr;
  size_t cap_ = 0, off_ = 0, high_ = 0;
};

struct Act {  // NHWC activation view: 16-bit elements, or fp32 when f32 is set (precision CD_PREC_F32)
  bf16_t* p = nullptr;
  int B = 0, H = 0, W = 0, C = 0;
  int ld = 0;  // pixel stride (elements)
  bool f32 = false;
  // fp32 path in the split-fp16 mode (CD_PREC_F32X3): p holds the fp16 pair [hi(C) | lo(C)] per pixel, ld = 2 * C
  // (kernels.h kX3ActScale). Only GroupNorm outputs (and their 2x2 average pools) take this form; convs consume it
  bool split = false;
  float* pf() const { return (float*)p; }
  // GroupNorm statistics of this tensor, written by the conv epilogue that produced it
  // ([B*H*W/32][2][C] fp32, ConvGemmParams::stats); stats_buf = storage, stats = valid content
  float* stats = nullptr;
  float* stats_buf = nullptr;
  int64_t rows() const { return (int64_t)B * H * W; }
};

// ------------------------------------------------------------------ parameters
struct ConvW {  // packed conv / linear weight: 16-bit (or fp32 when f32) [Npad][KH*KW*Cpad], fp32 bias[N]
  bf16_t* w = nullptr;
  bf16_t* wfrag = nullptr;  // the same rows in MFMA-fragment-major order for lin_stream.hip (1x1, 320 input channels)
  bf16_t* w3 = nullptr;     // CD_PREC_F32X3: fp16 [Npad][KH*KW][wh | wh | wl] (3 * Cpad per tap), from the fp32 rows in w
  bool f32 = false;
  float* b = nullptr;
  int N = 0, Cin = 0, Cpad = 0, KH = 1, KW = 1, Npad = 0;
  bool geglu = false;
  int Ktot() const { return KH * KW * Cpad; }
};

struct PackTarget {
  enum Kind { MATRIX_BF16, VECTOR_F32, MATRIX_F32 } kind;
  // MATRIX_BF16: rows [dst_row0, dst_row0+rows) of dst ConvW come from source rows
  //   src_base + (j/grp)*grp_stride + j%grp, j in [0, rows)
  ConvW* conv = nullptr;
  int dst_row0 = 0, rows = 0, src_base = 0, grp = 0, grp_stride = 0;
  // VECTOR_F32 / MATRIX_F32: dst[dst_off + j(*K)] with the same row mapping; geglu interleave
  float* fdst = nullptr;
  int dst_off = 0, K = 1;
  bool geglu = false;
  int geglu_N = 0;
  float scale = 1.f;  // rows are multiplied by this before the one rounding to the packed format
};

struct ParamDecl {
  std::string name;
  std::vector<int64_t> shape;  // reference (torch) shape
  std::vector<PackTarget> targets;
  bool loaded = false;
  bool transposed = false;  // reference tensor is [K][N] (used as x @ W); rows are transposed on load
};

class ParamStore {
 public:
  ~ParamStore();
  bool f32 = false;  // matrices are packed as fp32 (set by the network before it declares anything)
  bool x3 = false;   // ... and additionally as three-term fp16 splits (CD_PREC_F32X3)
  int* overflow = nullptr;  // host-visible word set when a split weight leaves the fp16 range (engine-owned)
  int version = 0;   // bumped by every load(): derived weights (LayerNorm folds) are rebuilt when it moves
  // allocate packed storage
  ConvW* new_conv(int N, int Cin, int KH, int KW, bool bias, bool geglu = false);
  float* new_vec(int n, float init = 0.f);
  // declare reference tensors and where their rows go
  ParamDecl& declare(const std::string& name, std::vector<int64_t> shape);
  // whole tensor -> whole ConvW; ref_ndim = rank of the reference tensor (2 Linear, 3 Conv1d, 4 Conv2d; 0 = auto)
  // scale: folded into the rows before they are rounded (attention: softmax scale * log2(e) into to_q)
  void conv_weight(const std::string& name, ConvW* c, int ref_ndim = 0, float scale = 1.f);
  // reference tensor of shape [Cin][N] applied as `x @ W` (OpenAI CLIP `proj` / `text_projection`)
  void conv_weight_t(const std::string& name, ConvW* c);
  void conv_bias(const std::string& name, ConvW* c);
  void conv_rows(const std::string& name, std::vector<int64_t> shape, ConvW* c, int dst_row0, int rows,
                 int src_base, int grp, int grp_stride, float scale = 1.f);
  void bias_rows(const std::string& name, int64_t n_total, float* dst, int dst_off, int rows, int src_base,
                 int grp, int grp_stride);
  void vec(const std::string& name, float* dst, int n);
  void mat_f32(const std::string& name, float* dst, int N, int K, int dst_row0 = 0);

  void load(hipStream_t st, const std::string& name, const float* host, int ndim, const int64_t* shape);
  int missing(std::string* first = nullptr) const;
  const std::vector<std::unique_ptr<ParamDecl>>& decls() const { return decls_; }
  size_t device_bytes() const { return dev_bytes_; }
 private:
  std::vector<std::unique_ptr<ParamDecl>> decls_;
  std::map<std::string, ParamDecl*> by_name_;
  std::vector<std::unique_ptr<ConvW>> convs_;
  std::vector<void*> allocs_;
  float* staging_ = nullptr;
  size_t staging_bytes_ = 0;
  size_t dev_bytes_ = 0;
  void* dmalloc(size_t bytes);
};

// ------------------------------------------------------------------ execution context
struct Ctx {
  hipStream_t st = nullptr;
  Arena* arena = nullptr;
  const bf16_t* zeros = nullptr;
  float* gn_partial = nullptr;  // [B][S][G][2] scratch, sized for the largest GroupNorm
  size_t gn_partial_floats = 0;
  bool f32 = false;             // the running network executes in fp32 (Net::f32)
  bool x3 = false;              // ... with its GroupNorm-fed convolutions as three-term fp16 GEMMs (Net::x3)
  int* overflow = nullptr;      // host-visible word the split kernels set on a value outside the fp16 range
};

// shared building blocks -------------------------------------------------------------------
struct ConvOpts {
  int stride = 1;
  int pad = 1;          // symmetric top/left padding (bottom/right implied by the output size)
  bool asym = false;    // VAE/Ho Downsample: F.pad(x,(0,1,0,1)) + stride-2 conv (model.py:72-76)
  bool up = false;      // nearest x2 before the conv
  const float* rowvec = nullptr; int rowvec_ld = 0; int rows_per_vec = 1;
  const Act* resid = nullptr;
  int act = ACT_NONE;
  float alpha = 1.f;
  bool out_f32 = false;
  void* out = nullptr;  // optional preallocated output
  int out_ld = 0;
  // LayerNorm the input rows inside the kernel (statistics only; the weights carry gain and bias): the streaming
  // K = 320 linear kernel only - the caller checks conv_ln_fold_available() first
  bool ln_fold = false;
  float ln_eps = 1e-5f;
  bool raw_geglu = false;      // fp32 path: a GEGLU projection leaves its [value | gate] columns as they are (k_geglu_f32 follows)
  bool want_stats = false;     // output feeds a GroupNorm: emit its statistics from the epilogue
  float* out_stats = nullptr;  // storage for them when `out` is preallocated
  int tile = 0;
};
// fp32 path, transformer blocks (st_f32.hip): q [B][Tq][ldq], k / v [B][Tk][ld] fp32 with per-image strides
Act attention_flash_f32_fwd(Ctx& c, const float* q, int ldq, const float* k, int ldk, int64_t k_bs, const float* v, int ldv,
                            int64_t v_bs, int B, int H, int Tq, int Tk, int D, float scale, int Himg, int Wimg, bool q_log2);
Act geglu_f32_fwd(Ctx& c, const Act& h);
Act split_rows_f32_fwd(Ctx& c, const Act& x, const Act* x2 = nullptr);  // split mode: fp32 rows (or a channel concat) as fp16 pairs (range-guarded)  // [rows][2 * Nout] packed [32 value | 32 gate] blocks -> [rows][Nout]
Act alloc_act(Ctx& c, int B, int H, int W, int C, bool with_stats = false);
// y = conv(x [| x2]) with the fused epilogue; returns the output view (bf16 unless out_f32)
Act conv_fwd(Ctx& c, const ConvW& w, const Act& x, const Act* x2, const ConvOpts& o);
// whether a LayerNorm-folded call of this layer on `rows` tokens would run (streaming kernel applicable and the shape
// large enough for it to be the fastest choice: the 64 x 64 level from 16 images up)
bool conv_ln_fold_available(const Ctx& c, const ConvW& w, int64_t rows);

struct GNW { float* g = nullptr; float* b = nullptr; int C = 0; float eps = 1e-5f; };
Act groupnorm_fwd(Ctx& c, const GNW& w, const Act& x, const Act* x2, bool silu, const float* film = nullptr,
                  int film_ld = 0);
struct LNW { float* g = nullptr; float* b = nullptr; int C = 0; };
Act layernorm_fwd(Ctx& c, const LNW& w, const Act& x);

// 2x2 average pool / nearest x2 upsample of a dense activation (resblock_updown, improved_ddpm/unet.py:104-135)
Act avgpool2_fwd(Ctx& c, const Act& x);
Act upsample2_fwd(Ctx& c, const Act& x);
// fp32 path: softmax(scale q k^T) v + obias with q | k in one token-major tensor and v in another
Act attention_f32_fwd(Ctx& c, const Act& qk, const Act& v, int H, int D, float scale, const float* obias);

// multi-head attention on token-major activations: q [B][Tq][ldq], k [B][Tk][ldk], v [B][Tk][ldv] (head h at column
// h*D of each). q_log2: q already carries scale * log2(e) (folded into its projection weights); `scale` is then unused
Act attention_fwd(Ctx& c, const bf16_t* q, int ldq, const bf16_t* k, int ldk, const bf16_t* v, int ldv, int B,
                  int H, int Tq, int Tk, int D, float scale, int Himg, int Wimg, bool q_log2 = false);
// the same with V pre-transposed: vt [B][H*D][Tpad] (keys contiguous, zero padded to Tpad % 64 == 0)
Act attention_vt_fwd(Ctx& c, const bf16_t* q, int ldq, const bf16_t* k, int ldk, const bf16_t* vt, int B,
                     int H, int Tq, int Tk, int Tpad, int D, float scale, int Himg, int Wimg, bool q_log2 = false);

// ------------------------------------------------------------------ networks
class Net {
 public:
  virtual ~Net() {}
  ParamStore params;
  cd_net_desc desc;
  bool f32 = false;  // desc.precision is CD_PREC_F32 or CD_PREC_F32X3
  bool x3 = false;   // desc.precision == CD_PREC_F32X3
  virtual int kind() const = 0;
};

struct TimeEmb {  // shared by all U-Nets: sinusoid -> MLP -> per-ResBlock projections in one launch
  int mode = 0, dim = 0, hidden = 0;
  float *w0 = nullptr, *b0 = nullptr, *w1 = nullptr, *b1 = nullptr;  // fp32 [hidden][dim], [hidden][hidden]
  float* proj_w = nullptr; float* proj_b = nullptr;                   // fused emb_layers: [proj_total][hidden]
  int proj_total = 0;
};

struct UNetIO {
  const bf16_t* xin = nullptr;   // NHWC bf16 [B][H][W][Cpad_in]
  int B = 0;
  // timestep source: either a schedule table row (device step counter or immediate) or explicit floats
  const StepCoef* tab = nullptr; const int* step_ptr = nullptr; int step = 0;
  const float* t_explicit = nullptr;  // [B] device, or null
  bool t_shared = true;               // all samples share one timestep -> embed once
  // classifier-free-guidance batch [uncond B/2 | cond B/2] built from ONE x_t (ddim.py:553-559 th.cat([x] * 2)): rows
  // B/2 .. B-1 of xin repeat rows 0 .. B/2-1 and only the cross-attention context differs, so everything ahead of the
  // first cross-attention may be computed once for B/2 rows and duplicated
  bool cfg_dup = false;
  // generalisation for the coupled encode / decode loop (cd_cycle_translate): the batch is [rows 0 .. B - dup_tail - 1 unique |
  // dup_tail rows that repeat the dup_tail rows just ahead of them], e.g. [encoder rows | decoder uncond | decoder cond]
  // with dup_tail = the decoder's sample count. cfg_dup is the case dup_tail = B / 2. 0 = no repeated rows.
  int dup_tail = 0;
  float* out = nullptr;               // fp32 [B*H*W][out_ld]
  int out_ld = 0;
};

class UNet : public Net {
 public:
  virtual void forward(Ctx& c, const UNetIO& io) = 0;
  // cross-attention K / V^T of the context are step-invariant: computed once per call
  virtual void set_context(Ctx& c, const bf16_t* ctx, int B, int L) { (void)c; (void)ctx; (void)B; (void)L; }
  int in_cpad = 32;
  int out_channels = 0;
  int image_size = 0;
  virtual size_t workspace_hint(int B) const = 0;
};

std::unique_ptr<UNet> make_unet_openai(const cd_net_desc& d);
std::unique_ptr<UNet> make_unet_ho(const cd_net_desc& d);

class VAE : public Net {
 public:
  int kind() const override { return CD_NET_VAE_KL; }
  virtual void encode_moments(Ctx& c, const bf16_t* img_nhwc, int B, int R, float* moments) = 0;  // fp32 [B*h*w][2*zc]
  virtual void decode(Ctx& c, const bf16_t* z_nhwc, int B, int h, float* img) = 0;                 // fp32 [B*R*R][3]
  int z_channels = 4, factor = 8;
  int moments_channels = 8;          // channels of encode_moments' output: 2 * embed_dim (KL) or embed_dim (VQ)
  const float* codebook = nullptr;   // VQ first stage: [n_embed][embed_dim] fp32, else null
  int n_embed = 0;
};
std::unique_ptr<VAE> make_vae_kl(const cd_net_desc& d);

class TextEncoder : public Net {  // token ids -> conditioning sequence
 public:
  virtual void encode(Ctx& c, const int* ids_dev, int B, int L, float* out) = 0;  // fp32 [B][L][width]
  // OpenAI-CLIP towers (DirectionalCLIP ranker): pooled + projected features fp32 [B][embed]
  virtual void text_features(Ctx&, const int*, int, int, float*) { CD_CHECK(false, "net has no pooled text features"); }
  virtual void image_features(Ctx&, const float*, int, float*) { CD_CHECK(false, "net has no image tower"); }
  virtual int width() const = 0;
  virtual int max_positions() const = 0;
};
std::unique_ptr<TextEncoder> make_clip_text(const cd_net_desc& d);

}  // namespace cd
