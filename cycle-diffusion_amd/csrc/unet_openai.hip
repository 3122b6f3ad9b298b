// Executor for the guided-diffusion style U-Net family:
//   * Stable Diffusion v1 / LDM text2img-large:  UNetModel with SpatialTransformer blocks
//     (ldm/modules/diffusionmodules/openaimodel.py:413-742, ldm/modules/attention.py:152-261)
//   * improved-DDPM AFHQ: UNetModel with resblock_updown, FiLM (scale-shift) norm and
//     QKVAttentionLegacy (model/lib/ddpm_ddim/models/improved_ddpm/unet.py:401-668)
// Parameter names are the reference's state_dict keys.
#include <limits.h>

#include <algorithm>

#include "engine.h"

namespace cd {

namespace {

struct ResW {
  int cin = 0, cout = 0;
  bool up = false, down = false, film = false;
  GNW gn1, gn2;
  ConvW *conv1 = nullptr, *conv2 = nullptr, *skip = nullptr;
  int emb_off = 0;
};

struct STW {  // SpatialTransformer with one BasicTransformerBlock
  int C = 0, heads = 0, dh = 0, ctx = 0;
  GNW norm;
  ConvW *proj_in = nullptr, *proj_out = nullptr;
  LNW ln1, ln2, ln3;
  // self-attention projections: fused [Wq * scale * log2(e); Wk; Wv] (V consumed token-major), or - d_head = 40,
  // where the pre-transposed V^T layout is worth more to k_attention than the fusion - [Wq*; Wk] plus Wv kept as the
  // A operand of a V^T = Wv . X^T GEMM
  ConvW *qkv1 = nullptr, *qk1 = nullptr, *v1 = nullptr, *o1 = nullptr;
  ConvW *q2 = nullptr, *k2 = nullptr, *v2 = nullptr, *o2 = nullptr;
  ConvW *ff1 = nullptr, *ff2 = nullptr;
  // LayerNorm-folded copies (320-channel level only): norm2 -> to_q of the cross-attention, norm3 -> the GEGLU
  // projection. Rebuilt from the loaded tensors whenever the parameter store changes (UNetOpenAI::refresh_ln_folds)
  ConvW *q2_ln = nullptr, *ff1_ln = nullptr;
  // context cache (step invariant)
  bf16_t* k2c = nullptr;  // [B*L][C]
  bf16_t* v2c = nullptr;  // [B*L][C] (token-major: k_attention transposes V tiles with LDS transpose reads)
};

struct ABW {  // AttentionBlock (improved_ddpm/unet.py:268-315)
  int C = 0, heads = 0, dh = 0;
  GNW norm;
  ConvW *qk = nullptr, *v = nullptr, *proj = nullptr;
  float* vbias = nullptr;
};

struct Layer {
  enum Kind { CONV_IN, RES, ST, AB, DOWN_CONV, UP_CONV } kind;
  int idx = 0;        // index into the matching weight vector
  ConvW* conv = nullptr;
};
struct Block { std::vector<Layer> layers; };

class UNetOpenAI : public UNet {
 public:
  explicit UNetOpenAI(const cd_net_desc& d);
  int kind() const override { return CD_NET_UNET_OPENAI; }
  void forward(Ctx& c, const UNetIO& io) override;
  void set_context(Ctx& c, const bf16_t* ctx, int B, int L) override;
  size_t workspace_hint(int B) const override;

 private:
  int mc_, hidden_;
  TimeEmb te_;
  std::vector<ResW> res_;
  std::vector<STW> st_;
  std::vector<ABW> ab_;
  std::vector<Block> in_blocks_, out_blocks_;
  Block mid_;
  GNW out_norm_;
  ConvW* out_conv_ = nullptr;
  int ctx_B_ = 0, ctx_L_ = 0;
  std::vector<void*> ctx_allocs_;

  int add_res(const std::string& pfx, int cin, int cout, bool up, bool down);
  int add_st(const std::string& pfx, int C, int heads, int dh);
  int add_ab(const std::string& pfx, int C, int heads);
  int folded_version_ = -1;
  void refresh_ln_folds(Ctx& c);
  Act run_block(Ctx& c, const Block& b, Act h, const Act* skip, const float* proj, int proj_ld, bool t_shared);
  Act res_fwd(Ctx& c, const ResW& r, const Act& x, const Act* x2, const float* proj, int proj_ld, bool t_shared);
  // dup_tail > 0: x holds the UNIQUE rows of a batch whose last dup_tail samples repeat the dup_tail samples ahead of them
  // (UNetIO::dup_tail; a classifier-free-guidance batch is dup_tail = x.B); everything ahead of the cross-attention runs on
  // x, then the token stream and the block input get the repeated samples appended and the rest runs on x.B + dup_tail
  Act st_fwd(Ctx& c, STW& s, const Act& x, int dup_tail = 0);
  // the block on images [b0, b0 + x.B) of the batch the context was set for, into the caller's output rows
  void st_core16(Ctx& c, STW& s, const Act& x_in, const Act& x_resid, Act& out, int b0, int dup_tail);
  Act ab_fwd(Ctx& c, const ABW& a, const Act& x);
 public:
  ~UNetOpenAI() override { for (void* p : ctx_allocs_) (void)hipFree(p); }
};

GNW make_gn(ParamStore& ps, const std::string& pfx, int C, float eps) {
  GNW g; g.C = C; g.eps = eps;
  g.g = ps.new_vec(C, 1.f); g.b = ps.new_vec(C, 0.f);
  ps.vec(pfx + ".weight", g.g, C);
  ps.vec(pfx + ".bias", g.b, C);
  return g;
}
LNW make_ln(ParamStore& ps, const std::string& pfx, int C) {
  LNW g; g.C = C;
  g.g = ps.new_vec(C, 1.f); g.b = ps.new_vec(C, 0.f);
  ps.vec(pfx + ".weight", g.g, C);
  ps.vec(pfx + ".bias", g.b, C);
  return g;
}
// ref_ndim: rank of the reference weight tensor (2 = nn.Linear, 3 = Conv1d, 4 = Conv2d)
ConvW* make_conv(ParamStore& ps, const std::string& pfx, int N, int Cin, int k, bool bias, bool geglu = false,
                 int ref_ndim = 4, float wscale = 1.f) {
  CD_CHECK(!(bias && wscale != 1.f), "a folded weight scale needs the bias scaled too");
  ConvW* c = ps.new_conv(N, Cin, k, k, bias, geglu);
  ps.conv_weight(pfx + ".weight", c, ref_ndim, wscale);
  if (bias) ps.conv_bias(pfx + ".bias", c);
  return c;
}

int UNetOpenAI::add_res(const std::string& pfx, int cin, int cout, bool up, bool down) {
  ResW r; r.cin = cin; r.cout = cout; r.up = up; r.down = down; r.film = desc.use_scale_shift_norm != 0;
  r.gn1 = make_gn(params, pfx + ".in_layers.0", cin, 1e-5f);
  r.conv1 = make_conv(params, pfx + ".in_layers.2", cout, cin, 3, true);
  r.gn2 = make_gn(params, pfx + ".out_layers.0", cout, 1e-5f);
  r.conv2 = make_conv(params, pfx + ".out_layers.3", cout, cout, 3, true);
  if (cin != cout) r.skip = make_conv(params, pfx + ".skip_connection", cout, cin, 1, true);
  r.emb_off = te_.proj_total;
  te_.proj_total += r.film ? 2 * cout : cout;
  res_.push_back(r);
  // emb_layers.1 is declared after proj storage exists (constructor tail); remember the prefix
  return (int)res_.size() - 1;
}

int UNetOpenAI::add_st(const std::string& pfx, int C, int heads, int dh) {
  STW s; s.C = C; s.heads = heads; s.dh = dh; s.ctx = desc.context_dim;
  CD_CHECK(heads * dh == C, "spatial transformer: inner dim %d != channels %d", heads * dh, C);
  CD_CHECK(desc.transformer_depth == 1, "only transformer_depth=1 is supported");
  s.norm = make_gn(params, pfx + ".norm", C, 1e-6f);
  s.proj_in = make_conv(params, pfx + ".proj_in", C, C, 1, true);
  s.proj_out = make_conv(params, pfx + ".proj_out", C, C, 1, true);
  const std::string tb = pfx + ".transformer_blocks.0";
  s.ln1 = make_ln(params, tb + ".norm1", C);
  s.ln2 = make_ln(params, tb + ".norm2", C);
  s.ln3 = make_ln(params, tb + ".norm3", C);
  // to_q rows carry the softmax scale and log2(e) (attention.py:178: sim = q k^T * dim_head^-0.5), folded in
  // fp32 before the one rounding to 16 bits: q comes out of its GEMM in the log2 units k_attention consumes.
  const float qscale = 1.44269504088896340736f / sqrtf((float)dh);
  // Measured on one box, B' = 32 (profiles/r2b_attention_v_layouts.txt): with token-major V the kernel's PV operand
  // comes from LDS transpose reads - 7 % faster than the V^T layout at d = 80, 12 % slower at d = 40 (twice the LDS
  // instructions for 41 useful rows); the fused q|k|v GEMM saves 17-26 us per block against [q|k] + V^T GEMMs.
  if (dh == 40) {
    s.qk1 = params.new_conv(2 * C, C, 1, 1, false);
    params.conv_rows(tb + ".attn1.to_q.weight", {C, C}, s.qk1, 0, C, 0, C, 0, qscale);
    params.conv_rows(tb + ".attn1.to_k.weight", {C, C}, s.qk1, C, C, 0, C, 0);
    s.v1 = make_conv(params, tb + ".attn1.to_v", C, C, 1, false, false, 2);
  } else {
    s.qkv1 = params.new_conv(3 * C, C, 1, 1, false);
    params.conv_rows(tb + ".attn1.to_q.weight", {C, C}, s.qkv1, 0, C, 0, C, 0, qscale);
    params.conv_rows(tb + ".attn1.to_k.weight", {C, C}, s.qkv1, C, C, 0, C, 0);
    params.conv_rows(tb + ".attn1.to_v.weight", {C, C}, s.qkv1, 2 * C, C, 0, C, 0);
  }
  s.o1 = make_conv(params, tb + ".attn1.to_out.0", C, C, 1, true, false, 2);
  s.q2 = make_conv(params, tb + ".attn2.to_q", C, C, 1, false, false, 2, qscale);
  s.k2 = make_conv(params, tb + ".attn2.to_k", C, s.ctx, 1, false, false, 2);
  s.v2 = make_conv(params, tb + ".attn2.to_v", C, s.ctx, 1, false, false, 2);
  s.o2 = make_conv(params, tb + ".attn2.to_out.0", C, C, 1, true, false, 2);
  s.ff1 = make_conv(params, tb + ".ff.net.0.proj", 8 * C, C, 1, true, /*geglu=*/true, 2);
  s.ff2 = make_conv(params, tb + ".ff.net.2", C, 4 * C, 1, true, false, 2);
  if (s.q2->wfrag) {  // derived weights, not declared as parameters
    s.q2_ln = params.new_conv(C, C, 1, 1, true);
    s.ff1_ln = params.new_conv(8 * C, C, 1, 1, true, /*geglu=*/true);
  }
  st_.push_back(s);
  return (int)st_.size() - 1;
}

int UNetOpenAI::add_ab(const std::string& pfx, int C, int heads) {
  ABW a; a.C = C; a.heads = heads; a.dh = C / heads;
  a.norm = make_gn(params, pfx + ".norm", C, 1e-5f);
  const int ch = a.dh;
  // legacy layout: qkv rows are [head][q(ch) | k(ch) | v(ch)] (QKVAttentionLegacy, unet.py:333-336)
  a.qk = params.new_conv(2 * C, C, 1, 1, true);
  a.v = params.new_conv(C, C, 1, 1, false);
  a.vbias = params.new_vec(C);
  const std::string wn = pfx + ".qkv.weight", bn = pfx + ".qkv.bias";
  params.conv_rows(wn, {3 * C, C, 1}, a.qk, 0, C, 0, ch, 3 * ch);
  params.conv_rows(wn, {3 * C, C, 1}, a.qk, C, C, ch, ch, 3 * ch);
  params.conv_rows(wn, {3 * C, C, 1}, a.v, 0, C, 2 * ch, ch, 3 * ch);
  params.bias_rows(bn, 3 * C, a.qk->b, 0, C, 0, ch, 3 * ch);
  params.bias_rows(bn, 3 * C, a.qk->b, C, C, ch, ch, 3 * ch);
  params.bias_rows(bn, 3 * C, a.vbias, 0, C, 2 * ch, ch, 3 * ch);
  a.proj = make_conv(params, pfx + ".proj_out", C, C, 1, true, false, 3);
  ab_.push_back(a);
  return (int)ab_.size() - 1;
}

UNetOpenAI::UNetOpenAI(const cd_net_desc& d) {
  desc = d;
  f32 = d.precision == CD_PREC_F32 || d.precision == CD_PREC_F32X3;
  x3 = d.precision == CD_PREC_F32X3;
  params.f32 = f32; params.x3 = x3;

  mc_ = d.model_channels; hidden_ = 4 * mc_;
  image_size = d.image_size; out_channels = d.out_channels;
  in_cpad = round_up(d.in_channels, 32);
  CD_CHECK(mc_ % 32 == 0, "model_channels must be a multiple of 32");
  const bool use_st = d.use_spatial_transformer != 0;
  auto has_attn = [&](int ds) { for (int i = 0; i < d.n_attn; ++i) if (d.attn[i] == ds) return true; return false; };
  auto heads_for = [&](int ch, int& heads, int& dh) {
    if (d.num_head_channels == -1) { heads = d.num_heads; dh = ch / heads; }
    else { heads = ch / d.num_head_channels; dh = d.num_head_channels; }
  };
  std::vector<std::string> res_pfx;
  auto add_res_named = [&](const std::string& pfx, int cin, int cout, bool up, bool down) {
    res_pfx.push_back(pfx);
    return add_res(pfx, cin, cout, up, down);
  };
  auto add_attn = [&](Block& blk, const std::string& pfx, int ch) {
    int heads, dh; heads_for(ch, heads, dh);
    Layer l;
    if (use_st) { l.kind = Layer::ST; l.idx = add_st(pfx, ch, heads, dh); }
    else { l.kind = Layer::AB; l.idx = add_ab(pfx, ch, heads); }
    blk.layers.push_back(l);
  };

  // input_blocks.0.0 : conv3x3 in -> mc
  {
    Block b; Layer l; l.kind = Layer::CONV_IN;
    l.conv = make_conv(params, "input_blocks.0.0", mc_, d.in_channels, 3, true);
    b.layers.push_back(l); in_blocks_.push_back(b);
  }
  std::vector<int> chans{mc_};
  int ch = mc_, ds = 1, bi = 1;
  for (int level = 0; level < d.n_mult; ++level) {
    const int mult = d.channel_mult[level];
    for (int r = 0; r < d.num_res_blocks; ++r) {
      Block b; Layer l; l.kind = Layer::RES;
      const std::string pfx = "input_blocks." + std::to_string(bi);
      l.idx = add_res_named(pfx + ".0", ch, mult * mc_, false, false);
      b.layers.push_back(l);
      ch = mult * mc_;
      if (has_attn(ds)) add_attn(b, pfx + ".1", ch);
      in_blocks_.push_back(b); chans.push_back(ch); ++bi;
    }
    if (level != d.n_mult - 1) {
      Block b; Layer l;
      const std::string pfx = "input_blocks." + std::to_string(bi) + ".0";
      if (d.resblock_updown) { l.kind = Layer::RES; l.idx = add_res_named(pfx, ch, ch, false, true); }
      else {
        CD_CHECK(d.conv_resample, "Downsample without conv (avg-pool) only exists inside resblock_updown");
        l.kind = Layer::DOWN_CONV; l.conv = make_conv(params, pfx + ".op", ch, ch, 3, true);
      }
      b.layers.push_back(l); in_blocks_.push_back(b); chans.push_back(ch); ++bi; ds *= 2;
    }
  }
  // middle
  {
    Layer l; l.kind = Layer::RES; l.idx = add_res_named("middle_block.0", ch, ch, false, false);
    mid_.layers.push_back(l);
    add_attn(mid_, "middle_block.1", ch);
    Layer l2; l2.kind = Layer::RES; l2.idx = add_res_named("middle_block.2", ch, ch, false, false);
    mid_.layers.push_back(l2);
  }
  // output blocks
  int oi = 0;
  for (int level = d.n_mult - 1; level >= 0; --level) {
    const int mult = d.channel_mult[level];
    for (int i = 0; i <= d.num_res_blocks; ++i) {
      const int ich = chans.back(); chans.pop_back();
      Block b; Layer l; l.kind = Layer::RES;
      const std::string pfx = "output_blocks." + std::to_string(oi);
      l.idx = add_res_named(pfx + ".0", ch + ich, mc_ * mult, false, false);
      b.layers.push_back(l);
      ch = mc_ * mult;
      int sub = 1;
      if (has_attn(ds)) { add_attn(b, pfx + "." + std::to_string(sub), ch); ++sub; }
      if (level && i == d.num_res_blocks) {
        Layer u;
        const std::string up = pfx + "." + std::to_string(sub);
        if (d.resblock_updown) { u.kind = Layer::RES; u.idx = add_res_named(up, ch, ch, true, false); }
        else { u.kind = Layer::UP_CONV; u.conv = make_conv(params, up + ".conv", ch, ch, 3, true); }
        b.layers.push_back(u);
        ds /= 2;
      }
      out_blocks_.push_back(b); ++oi;
    }
  }
  out_norm_ = make_gn(params, "out.0", ch, 1e-5f);
  out_conv_ = make_conv(params, "out.2", d.out_channels, mc_, 3, true);
  CD_CHECK(ch == mc_, "channel bookkeeping: final channels %d != model_channels %d", ch, mc_);

  // time embedding MLP (fp32) + all emb_layers.1 projections fused into one matrix
  te_.mode = 0; te_.dim = mc_; te_.hidden = hidden_;
  te_.w0 = params.new_vec(hidden_ * mc_); te_.b0 = params.new_vec(hidden_);
  te_.w1 = params.new_vec(hidden_ * hidden_); te_.b1 = params.new_vec(hidden_);
  params.mat_f32("time_embed.0.weight", te_.w0, hidden_, mc_);
  params.vec("time_embed.0.bias", te_.b0, hidden_);
  params.mat_f32("time_embed.2.weight", te_.w1, hidden_, hidden_);
  params.vec("time_embed.2.bias", te_.b1, hidden_);
  te_.proj_w = params.new_vec(te_.proj_total * hidden_);
  te_.proj_b = params.new_vec(te_.proj_total);
  for (size_t i = 0; i < res_.size(); ++i) {
    const int n = res_[i].film ? 2 * res_[i].cout : res_[i].cout;
    params.mat_f32(res_pfx[i] + ".emb_layers.1.weight", te_.proj_w, n, hidden_, res_[i].emb_off);
    params.bias_rows(res_pfx[i] + ".emb_layers.1.bias", n, te_.proj_b, res_[i].emb_off, n, 0, n, 0);
  }
}

size_t UNetOpenAI::workspace_hint(int B) const {
  // generous: ~40 live full-resolution tensors of the widest layer
  const size_t hw = (size_t)image_size * image_size;
  size_t widest = 0;
  for (auto& r : res_) widest = std::max(widest, (size_t)std::max(r.cin, r.cout));
  return (size_t)B * hw * std::max(widest, (size_t)8 * mc_) * 2 * 24 + (64u << 20);
}

Act UNetOpenAI::res_fwd(Ctx& c, const ResW& r, const Act& x, const Act* x2, const float* proj, int proj_ld,
                        bool t_shared) {
  // output first (it outlives the block's temporaries)
  const int Ho = r.up ? x.H * 2 : (r.down ? x.H / 2 : x.H);
  const int Wo = r.up ? x.W * 2 : (r.down ? x.W / 2 : x.W);
  Act out = alloc_act(c, x.B, Ho, Wo, r.cout, /*with_stats=*/true);
  const int rpv = t_shared ? INT_MAX : Ho * Wo;  // output rows sharing one time-embedding vector
  const size_t mk = c.arena->mark();
  Act h = groupnorm_fwd(c, r.gn1, x, x2, /*silu=*/true);
  Act xs = x;           // skip-path input (identity path needs a single dense tensor)
  const Act* xs2 = x2;
  bool up_in_conv = false;
  if (r.down) {
    CD_CHECK(!x2, "resblock_updown with concat input is not produced by the reference");
    h = avgpool2_fwd(c, h);
    xs = avgpool2_fwd(c, x); xs2 = nullptr;
  } else if (r.up) {
    CD_CHECK(!x2, "resblock_updown with concat input is not produced by the reference");
    up_in_conv = true;
    xs = upsample2_fwd(c, x); xs2 = nullptr;
  }
  ConvOpts o1; o1.up = up_in_conv; o1.want_stats = true;
  const float* pr = proj + r.emb_off;
  if (!r.film) { o1.rowvec = pr; o1.rowvec_ld = proj_ld; o1.rows_per_vec = rpv; }
  Act h2 = conv_fwd(c, *r.conv1, h, nullptr, o1);
  Act h3;
  if (r.film) h3 = groupnorm_fwd(c, r.gn2, h2, nullptr, true, pr, t_shared ? 0 : proj_ld);
  else h3 = groupnorm_fwd(c, r.gn2, h2, nullptr, true);
  Act skip;
  if (r.skip) {
    ConvOpts os; os.pad = 0;
    // (Round 4 measured the split form of this projection - its unnormalised input, or the concat of two, first written as
    // fp16 pairs by split_rows_f32_fwd: on config 5 the conversion pass over the 256 x 256 tensors costs more than k_conv_f32
    // saves - 2.93 -> 2.82 images/s same box, profiles/r4_c5r_split_skip_projection_ab.txt - so it stays on the fp32 matrix
    // instructions.)
    skip = conv_fwd(c, *r.skip, xs, xs2, os);
  } else {
    CD_CHECK(!xs2, "identity skip with concat input");
    skip = xs;
  }
  ConvOpts o2; o2.resid = &skip; o2.out = out.p; o2.out_ld = out.ld; o2.out_stats = out.stats_buf;
  conv_fwd(c, *r.conv2, h3, nullptr, o2);
  out.stats = out.stats_buf;
  c.arena->release(mk);
  return out;
}

// V^T[b] = Wv . X[b]^T : weights as the A operand, tokens as the B operand -> [B][C][Tpad]
static void vt_gemm(Ctx& c, const ConvW& wv, const bf16_t* x, int ldx, int B, int T, int Tpad, bf16_t* vt) {
  ConvGemmParams p;
  p.src0 = wv.w; p.C0 = wv.Cpad; p.ld0 = wv.Cpad;
  p.B = 1; p.Hs = wv.N; p.Ws = 1; p.Hin = wv.N; p.Win = 1; p.Hout = wv.N; p.Wout = 1;
  p.M = wv.N;
  p.wgt = x; p.Ktot = wv.Cpad; p.N = T;
  CD_CHECK(ldx == wv.Cpad, "vt_gemm: token row stride %d must equal K %d", ldx, wv.Cpad);
  p.nbatch = B; p.a_bs = 0; p.w_bs = (int64_t)T * ldx; p.o_bs = (int64_t)wv.N * Tpad;
  p.out = vt; p.out_ld = Tpad; p.zeros = c.zeros;
  launch_conv_gemm(c.st, p);
}

void UNetOpenAI::set_context(Ctx& c, const bf16_t* ctx, int B, int L) {
  if (st_.empty()) return;
  if (B != ctx_B_ || L != ctx_L_) {
    for (void* p : ctx_allocs_) (void)hipFree(p);
    ctx_allocs_.clear();
    const size_t esz = f32 ? 4 : 2;  // fp32 networks take the context as fp32 and keep fp32 K / V
    for (auto& s : st_) {
      HIP_CHECK(hipMalloc((void**)&s.k2c, (size_t)B * L * s.C * esz + 256));
      HIP_CHECK(hipMalloc((void**)&s.v2c, (size_t)B * L * s.C * esz + 256));
      ctx_allocs_.push_back(s.k2c); ctx_allocs_.push_back(s.v2c);
    }
    ctx_B_ = B; ctx_L_ = L;
  }
  const bool keep_f32 = c.f32, keep_x3 = c.x3;
  c.f32 = f32; c.x3 = x3;  // (forward() sets these for its own duration; the context projections run outside it)
  Act cx; cx.p = (bf16_t*)ctx; cx.B = 1; cx.H = B * L; cx.W = 1; cx.C = st_[0].ctx; cx.ld = cx.C; cx.f32 = f32;
  for (auto& s : st_) {
    ConvOpts o; o.pad = 0; o.out = s.k2c; o.out_ld = s.C;
    conv_fwd(c, *s.k2, cx, nullptr, o);
    o.out = s.v2c;
    conv_fwd(c, *s.v2, cx, nullptr, o);
  }
  c.f32 = keep_f32; c.x3 = keep_x3;
}

// rows [0, n) of src -> rows [0, n) of dst, and its last `tail` rows again -> rows [n, n + tail) (dense, `cols` 16-bit
// elements per row)
static void dup_rows(Ctx& c, const bf16_t* src, bf16_t* dst, int64_t n, int64_t tail, int cols) {
  launch_copy_strided_bf16(c.st, src, cols, dst, cols, n, cols);
  launch_copy_strided_bf16(c.st, src + (n - tail) * cols, cols, dst + n * cols, cols, tail, cols);
}

Act UNetOpenAI::st_fwd(Ctx& c, STW& s, const Act& x_in, int dup_tail) {
  int B = x_in.B;
  const bool dup = dup_tail > 0;
  const int T = x_in.H * x_in.W, C = s.C;
  CD_CHECK(dup_tail >= 0 && dup_tail <= B, "transformer block: %d repeated samples of %d", dup_tail, B);
  Act out = alloc_act(c, B + dup_tail, x_in.H, x_in.W, C, /*with_stats=*/true);
  Act x = x_in;
  if (dup) {  // the block input at full batch: residual of proj_out (lives as long as the block's output)
    CD_CHECK(x_in.ld == C, "transformer block: dense input expected");
    x = alloc_act(c, B + dup_tail, x_in.H, x_in.W, C);
    dup_rows(c, x_in.p, x.p, (int64_t)B * T, (int64_t)dup_tail * T, C);
  }
  const size_t mk = c.arena->mark();
  CD_CHECK(s.k2c && ctx_B_ == B + dup_tail, "cross-attention context not set for batch %d", B + dup_tail);
  if (c.f32) {
    ConvOpts p0; p0.pad = 0;
    Act n = groupnorm_fwd(c, s.norm, x_in, nullptr, false);
    Act h = conv_fwd(c, *s.proj_in, n, nullptr, p0);  // tokens [B*T][C]
    const float scale = 1.0f / sqrtf((float)s.dh);
    // CD_PREC_F32 / F32X3 (st_f32.hip): the reference's own arithmetic for this block (`precision = "full"`,
    // stable_diffusion_stochastic_text_wrapper.py:117). LayerNorm / GroupNorm outputs feed their projections as fp32 (k_conv_f32)
    // or, in the split mode, as fp16 pairs (three-term products on the 16-bit matrix cores) - there the attention output, the
    // GEGLU output and the residual stream ahead of proj_out are written as pairs too (range-guarded like the norms: a value
    // beyond 4094 raises at the next API entry, and `precision = fp32` is the answer); attention is fp32 flash attention.
    CD_CHECK(!dup, "transformer block: prefix sharing is a 16-bit path feature");
    const float* ck = (const float*)s.k2c;
    const float* cv = (const float*)s.v2c;
    {  // self-attention
      const size_t m2 = c.arena->mark();
      Act n1 = layernorm_fwd(c, s.ln1, h);
      Act a;
      if (s.qkv1) {
        Act qkv = conv_fwd(c, *s.qkv1, n1, nullptr, p0);  // [B*T][3C] = q (log2 units) | k | v
        a = attention_flash_f32_fwd(c, qkv.pf(), qkv.ld, qkv.pf() + C, qkv.ld, (int64_t)T * qkv.ld, qkv.pf() + 2 * C, qkv.ld,
                                    (int64_t)T * qkv.ld, B, s.heads, T, T, s.dh, scale, x.H, x.W, /*q_log2=*/true);
      } else {
        Act qk = conv_fwd(c, *s.qk1, n1, nullptr, p0);  // [B*T][2C]
        Act vv = conv_fwd(c, *s.v1, n1, nullptr, p0);   // [B*T][C]
        a = attention_flash_f32_fwd(c, qk.pf(), qk.ld, qk.pf() + C, qk.ld, (int64_t)T * qk.ld, vv.pf(), vv.ld,
                                    (int64_t)T * vv.ld, B, s.heads, T, T, s.dh, scale, x.H, x.W, /*q_log2=*/true);
      }
      ConvOpts o; o.pad = 0; o.resid = &h; o.out = h.p; o.out_ld = h.ld;  // in place: read then written by the same lane
      conv_fwd(c, *s.o1, a, nullptr, o);
      c.arena->release(m2);
    }
    {  // cross-attention over the cached fp32 context K / V
      const size_t m2 = c.arena->mark();
      Act n2 = layernorm_fwd(c, s.ln2, h);
      Act q = conv_fwd(c, *s.q2, n2, nullptr, p0);
      Act a = attention_flash_f32_fwd(c, q.pf(), q.ld, ck, C, (int64_t)ctx_L_ * C, cv, C, (int64_t)ctx_L_ * C, B, s.heads, T,
                                      ctx_L_, s.dh, scale, x.H, x.W, /*q_log2=*/true);
      ConvOpts o; o.pad = 0; o.resid = &h; o.out = h.p; o.out_ld = h.ld;
      conv_fwd(c, *s.o2, a, nullptr, o);
      c.arena->release(m2);
    }
    {  // GEGLU feed-forward: projection materialised, exact-erf GELU in fp32 - in row chunks, so that the [rows][8C] fp32
       // projection stays below 1 GiB whatever the fold (at 64 x 64 and C = 320 a B' = 64 batch would need 2.7 GB for it on top
       // of the attention buffers; every row is independent, so chunking changes no value)
      const int64_t rows = h.rows();
      int64_t chunk = ((int64_t)1 << 30) / ((int64_t)8 * C * 4);
      chunk = std::max<int64_t>(4096, chunk & ~(int64_t)4095);
      for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
        const int64_t n = std::min(chunk, rows - r0);
        const size_t m2 = c.arena->mark();
        Act hv = h;  // rows [r0, r0 + n) of the token stream as an [1][n][1][C] activation
        hv.p = (bf16_t*)(h.pf() + r0 * h.ld); hv.B = 1; hv.H = (int)n; hv.W = 1;
        Act n3 = layernorm_fwd(c, s.ln3, hv);
        ConvOpts pg; pg.pad = 0; pg.raw_geglu = true;
        Act g8 = conv_fwd(c, *s.ff1, n3, nullptr, pg);  // [n][8C] in packed [32 value | 32 gate] blocks
        Act g = geglu_f32_fwd(c, g8);                   // [n][4C]
        ConvOpts o; o.pad = 0; o.resid = &hv; o.out = hv.p; o.out_ld = hv.ld;
        conv_fwd(c, *s.ff2, g, nullptr, o);
        c.arena->release(m2);
      }
    }
    ConvOpts po; po.pad = 0; po.resid = &x; po.out = out.p; po.out_ld = out.ld; po.out_stats = out.stats_buf;
    if (c.x3) {  // the residual stream as fp16 pairs: proj_out is a split product too, and its epilogue emits the statistics
      Act hs = split_rows_f32_fwd(c, h);
      Act y = conv_fwd(c, *s.proj_out, hs, nullptr, po);
      out.stats = y.stats;
    } else {
      conv_fwd(c, *s.proj_out, h, nullptr, po);
    }
    c.arena->release(mk);
    return out;
  }
  // 16-bit path. Every image is independent through the whole block: at the 64 x 64 level a large batch runs depth-first in
  // chunks of ST_CHUNK images, so that the chunk's token tensors (42 MB per 320-channel tensor and 16 images, 168 MB for the
  // GEGLU hidden tensor) stay in the 256 MB Infinity Cache between the block's ~15 kernels instead of making an HBM round
  // trip each - the N = K = 320 projections, the norm passes and the feed-forward pair are bound by exactly those trips
  // (DESIGN.md 8). Chunks of 16 images are 65 536 rows: 256 strips / row tiles, one full round of the chip per kernel.
  static const int st_chunk = [] { const char* e = getenv("CYCLEDIFF_ST_CHUNK"); return e ? atoi(e) : 0; }();
  c.arena->release(mk);
  if (!dup && st_chunk > 0 && T >= 4096 && B >= 2 * st_chunk) {
    for (int b0 = 0; b0 < B; b0 += st_chunk) {
      const int n = std::min(st_chunk, B - b0);
      auto view = [&](const Act& a) {
        Act v = a; v.B = n;
        v.p = a.p + (int64_t)b0 * T * a.ld;
        if (a.stats) v.stats = a.stats + ((int64_t)b0 * T / 32) * 2 * a.C;
        if (a.stats_buf) v.stats_buf = a.stats_buf + ((int64_t)b0 * T / 32) * 2 * a.C;
        return v;
      };
      Act xv = view(x_in), ov = view(out);
      st_core16(c, s, xv, xv, ov, b0, 0);
    }
  } else {
    st_core16(c, s, x_in, x, out, 0, dup_tail);
  }
  out.stats = out.stats_buf;
  return out;
}

void UNetOpenAI::st_core16(Ctx& c, STW& s, const Act& x_in, const Act& x, Act& out, int b0, int dup_tail) {
  const bool dup = dup_tail > 0;
  int B = x_in.B;
  const int T = x_in.H * x_in.W, C = s.C;
  const size_t mk = c.arena->mark();
  ConvOpts p0; p0.pad = 0;
  const float scale = 1.0f / sqrtf((float)s.dh);
  const bf16_t* k2c = s.k2c + (int64_t)b0 * ctx_L_ * C;
  const bf16_t* v2c = s.v2c + (int64_t)b0 * ctx_L_ * C;
  Act n = groupnorm_fwd(c, s.norm, x_in, nullptr, false);
  Act h = conv_fwd(c, *s.proj_in, n, nullptr, p0);  // tokens [B*T][C]
  {  // self-attention
    const size_t m2 = c.arena->mark();
    Act n1 = layernorm_fwd(c, s.ln1, h);
    Act a;
    if (s.qkv1) {
      Act qkv = conv_fwd(c, *s.qkv1, n1, nullptr, p0);  // [B*T][3C] = q (log2 units) | k | v
      a = attention_fwd(c, qkv.p, qkv.ld, qkv.p + C, qkv.ld, qkv.p + 2 * C, qkv.ld, B, s.heads, T, T, s.dh, scale,
                        x.H, x.W, /*q_log2=*/true);
    } else {
      Act qk = conv_fwd(c, *s.qk1, n1, nullptr, p0);  // [B*T][2C]
      const int Tpad = round_up(T, 64);
      bf16_t* vt = (bf16_t*)c.arena->alloc((size_t)B * C * Tpad * 2);
      if (Tpad != T) HIP_CHECK(hipMemsetAsync(vt, 0, (size_t)B * C * Tpad * 2, c.st));
      vt_gemm(c, *s.v1, n1.p, n1.ld, B, T, Tpad, vt);
      a = attention_vt_fwd(c, qk.p, qk.ld, qk.p + C, qk.ld, vt, B, s.heads, T, T, Tpad, s.dh, scale, x.H, x.W,
                           /*q_log2=*/true);
    }
    ConvOpts o; o.pad = 0; o.resid = &h; o.out = h.p; o.out_ld = h.ld;  // in-place residual update:
    conv_fwd(c, *s.o1, a, nullptr, o);  // each element is read then written by the same lane
    c.arena->release(m2);
  }
  if (dup) {  // from here on the repeated samples see their own contexts
    Act h2 = alloc_act(c, B + dup_tail, x_in.H, x_in.W, C);
    dup_rows(c, h.p, h2.p, (int64_t)B * T, (int64_t)dup_tail * T, C);
    h = h2;
    B += dup_tail;
  }
  {  // cross-attention over the cached context K / V
    const size_t m2 = c.arena->mark();
    Act q;
    if (s.q2_ln && conv_ln_fold_available(c, *s.q2_ln, h.rows())) {  // norm2 inside the projection kernel
      ConvOpts pl; pl.pad = 0; pl.ln_fold = true;
      q = conv_fwd(c, *s.q2_ln, h, nullptr, pl);
    } else {
      Act n2 = layernorm_fwd(c, s.ln2, h);
      q = conv_fwd(c, *s.q2, n2, nullptr, p0);
    }
    Act a = attention_fwd(c, q.p, q.ld, k2c, C, v2c, C, B, s.heads, T, ctx_L_, s.dh, scale, x.H, x.W,
                          /*q_log2=*/true);
    ConvOpts o; o.pad = 0; o.resid = &h; o.out = h.p; o.out_ld = h.ld;  // in-place residual update
    conv_fwd(c, *s.o2, a, nullptr, o);
    c.arena->release(m2);
  }
  {  // GEGLU feed-forward
    const size_t m2 = c.arena->mark();
    Act g;  // [B*T][4C]
    if (s.ff1_ln && conv_ln_fold_available(c, *s.ff1_ln, h.rows())) {  // norm3 inside the GEGLU projection kernel
      ConvOpts pl; pl.pad = 0; pl.ln_fold = true;
      g = conv_fwd(c, *s.ff1_ln, h, nullptr, pl);
    } else {
      Act n3 = layernorm_fwd(c, s.ln3, h);
      g = conv_fwd(c, *s.ff1, n3, nullptr, p0);
    }
    ConvOpts o; o.pad = 0; o.resid = &h; o.out = h.p; o.out_ld = h.ld;
    conv_fwd(c, *s.ff2, g, nullptr, o);
    c.arena->release(m2);
  }
  ConvOpts po; po.pad = 0; po.resid = &x; po.out = out.p; po.out_ld = out.ld; po.out_stats = out.stats_buf;
  conv_fwd(c, *s.proj_out, h, nullptr, po);
  c.arena->release(mk);
}

Act UNetOpenAI::ab_fwd(Ctx& c, const ABW& a, const Act& x) {
  const int B = x.B, T = x.H * x.W, C = a.C;
  Act out = alloc_act(c, B, x.H, x.W, C, /*with_stats=*/true);
  const size_t mk = c.arena->mark();
  ConvOpts p0; p0.pad = 0;
  Act n = groupnorm_fwd(c, a.norm, x, nullptr, false);
  Act qk = conv_fwd(c, *a.qk, n, nullptr, p0);
  if (c.f32) {  // fp32 path: V as ordinary tokens, its bias added to the output (rows of P sum to 1)
    Act v = conv_fwd(c, *a.v, n, nullptr, p0);
    Act o = attention_f32_fwd(c, qk, v, a.heads, a.dh, 1.0f / sqrtf((float)a.dh), a.vbias);
    ConvOpts po; po.pad = 0; po.resid = &x; po.out = out.p; po.out_ld = out.ld;
    conv_fwd(c, *a.proj, o, nullptr, po);
    c.arena->release(mk);
    return out;
  }
  const int Tpad = round_up(T, 64);
  bf16_t* vt = (bf16_t*)c.arena->alloc((size_t)B * C * Tpad * 2);
  if (Tpad != T) HIP_CHECK(hipMemsetAsync(vt, 0, (size_t)B * C * Tpad * 2, c.st));
  vt_gemm(c, *a.v, n.p, n.ld, B, T, Tpad, vt);
  Act o = alloc_act(c, B, x.H, x.W, C);
  AttnParams p;
  p.q = qk.p; p.k = qk.p + C; p.vt = vt; p.o = o.p;
  p.B = B; p.H = a.heads; p.Tq = T; p.Tk = T; p.D = a.dh;
  p.ldq = qk.ld; p.ldk = qk.ld; p.ldo = o.ld;
  p.q_bs = (int64_t)T * qk.ld; p.k_bs = (int64_t)T * qk.ld; p.o_bs = (int64_t)T * o.ld;
  p.vt_dpad = a.dh; p.vt_tpad = Tpad; p.scale = 1.0f / sqrtf((float)a.dh); p.obias = a.vbias;
  launch_attention(c.st, p);
  ConvOpts po; po.pad = 0; po.resid = &x; po.out = out.p; po.out_ld = out.ld; po.out_stats = out.stats_buf;
  conv_fwd(c, *a.proj, o, nullptr, po);
  out.stats = out.stats_buf;
  c.arena->release(mk);
  return out;
}

Act UNetOpenAI::run_block(Ctx& c, const Block& b, Act h, const Act* skip, const float* proj, int proj_ld,
                          bool t_shared) {
  bool first = true;
  for (const Layer& l : b.layers) {
    const Act* x2 = first ? skip : nullptr;
    switch (l.kind) {
      case Layer::CONV_IN: { ConvOpts o; o.want_stats = true; h = conv_fwd(c, *l.conv, h, nullptr, o); break; }
      case Layer::RES: h = res_fwd(c, res_[l.idx], h, x2, proj, proj_ld, t_shared); break;
      case Layer::ST: h = st_fwd(c, st_[l.idx], h); break;
      case Layer::AB: h = ab_fwd(c, ab_[l.idx], h); break;
      case Layer::DOWN_CONV: { ConvOpts o; o.stride = 2; o.want_stats = true; h = conv_fwd(c, *l.conv, h, nullptr, o); break; }
      case Layer::UP_CONV: { ConvOpts o; o.up = true; o.want_stats = true; h = conv_fwd(c, *l.conv, h, nullptr, o); break; }
    }
    first = false;
  }
  return h;
}

// W' = W diag(gamma), b' = b + W beta for the LayerNorm-folded layers (attention.py:211-215: x + attn2(norm2(x)),
// x + ff(norm3(x))), in both weight layouts; stream-ordered, a few microseconds, only after a parameter load
void UNetOpenAI::refresh_ln_folds(Ctx& c) {
  if (folded_version_ == params.version) return;
  for (STW& s : st_) {
    if (!s.q2_ln) continue;
    struct { const ConvW* src; ConvW* dst; const LNW* ln; } jobs[2] = {{s.q2, s.q2_ln, &s.ln2}, {s.ff1, s.ff1_ln, &s.ln3}};
    for (auto& j : jobs) {
      launch_fold_ln(c.st, j.src->w, j.src->Ktot(), j.ln->g, j.ln->b, j.src->b, j.dst->w, j.dst->b, j.src->Npad,
                     j.src->Cpad);
      launch_pack_wfrag(c.st, j.dst->w, j.dst->Ktot(), j.dst->wfrag, j.dst->Npad);
    }
  }
  folded_version_ = params.version;
}

void UNetOpenAI::forward(Ctx& c, const UNetIO& io) {
  const size_t mk0 = c.arena->mark();
  c.f32 = f32; c.x3 = x3;
  refresh_ln_folds(c);
  const int B = io.B, R = image_size;
  // ---- time embedding: sinusoid -> Linear -> SiLU -> Linear, then every ResBlock's
  //      emb_layers (SiLU -> Linear) in one launch (openaimodel.py:506-511,723-724,263)
  const int tB = io.t_shared ? 1 : B;
  float* sinu = (float*)c.arena->alloc((size_t)tB * mc_ * 4);
  float* e1 = (float*)c.arena->alloc((size_t)tB * hidden_ * 4);
  float* emb = (float*)c.arena->alloc((size_t)tB * hidden_ * 4);
  float* proj = (float*)c.arena->alloc((size_t)tB * te_.proj_total * 4);
  launch_timestep_embedding(c.st, io.tab, io.step_ptr, io.step, io.t_explicit, sinu, tB, mc_, 0);
  launch_vec_linear(c.st, sinu, mc_, te_.w0, te_.b0, e1, hidden_, tB, mc_, hidden_, 0, 1);
  launch_vec_linear(c.st, e1, hidden_, te_.w1, te_.b1, emb, hidden_, tB, hidden_, hidden_, 0, 0);
  launch_vec_linear(c.st, emb, hidden_, te_.proj_w, te_.proj_b, proj, te_.proj_total, tB, hidden_,
                    te_.proj_total, 1, 0);
  const int proj_ld = te_.proj_total;
  Act x; x.p = (bf16_t*)io.xin; x.B = B; x.H = R; x.W = R; x.C = in_cpad; x.ld = in_cpad; x.f32 = f32;
  if (x3) { x.split = true; x.ld = 2 * in_cpad; }  // the samplers hand CD_PREC_F32X3 networks their input as fp16 pairs
  std::vector<Act> hs;
  Act h = x;
  size_t first_full = 0;  // first input block that runs at the full batch
  // Classifier-free-guidance batch built from one x_t: conv_in, the first ResBlock and the first transformer block up
  // to its self-attention output see identical rows in both halves (the context enters at the cross-attention): they run
  // once on B / 2 rows (CYCLEDIFF_CFG_SHARE=0 turns this off for A/B runs). Per-row arithmetic is unchanged.
  static const bool cfg_share = [] { const char* e = getenv("CYCLEDIFF_CFG_SHARE"); return !(e && e[0] == '0'); }();
  CD_CHECK(!(io.cfg_dup && io.dup_tail) && io.dup_tail >= 0 && 2 * io.dup_tail <= B, "U-Net: repeated-row description");
  const int dtail = io.cfg_dup ? (B % 2 == 0 ? B / 2 : 0) : io.dup_tail;  // trailing samples that repeat the ones ahead of them
  if (dtail > 0 && cfg_share && !f32 && io.t_shared && in_blocks_.size() >= 2 &&
      in_blocks_[0].layers.size() == 1 && in_blocks_[0].layers[0].kind == Layer::CONV_IN &&
      in_blocks_[1].layers.size() == 2 && in_blocks_[1].layers[0].kind == Layer::RES &&
      in_blocks_[1].layers[1].kind == Layer::ST) {
    const int Bh = B - dtail;  // unique samples
    Act xh = x; xh.B = Bh;
    Act h0 = run_block(c, in_blocks_[0], xh, nullptr, proj, proj_ld, true);  // [Bh] conv_in output
    // the skip connection of the last output block needs it at full batch (with its GroupNorm statistics)
    Act h0f = alloc_act(c, B, h0.H, h0.W, h0.C, /*with_stats=*/h0.stats != nullptr);
    CD_CHECK(h0.ld == h0.C, "conv_in output: dense tensor expected");
    const int64_t rows_tail = (int64_t)dtail * h0.H * h0.W;
    dup_rows(c, h0.p, h0f.p, h0.rows(), rows_tail, h0.C);
    if (h0.stats) {  // [row block][2][C] fp32 (a 32-row block lies inside one sample), copied as 2 x 16 bit
      dup_rows(c, (const bf16_t*)h0.stats, (bf16_t*)h0f.stats_buf, (h0.rows() / 32) * 2, (rows_tail / 32) * 2, h0.C * 2);
      h0f.stats = h0f.stats_buf;
    }
    hs.push_back(h0f);
    const Layer& lr = in_blocks_[1].layers[0];
    const Layer& lt = in_blocks_[1].layers[1];
    Act hr = res_fwd(c, res_[lr.idx], h0, nullptr, proj, proj_ld, true);
    h = st_fwd(c, st_[lt.idx], hr, /*dup_tail=*/dtail);
    hs.push_back(h);
    first_full = 2;
  }
  for (size_t bi = first_full; bi < in_blocks_.size(); ++bi) {
    h = run_block(c, in_blocks_[bi], h, nullptr, proj, proj_ld, io.t_shared);
    hs.push_back(h);
  }
  h = run_block(c, mid_, h, nullptr, proj, proj_ld, io.t_shared);
  for (const Block& b : out_blocks_) {
    Act skip = hs.back(); hs.pop_back();  // th.cat([h, hs.pop()], dim=1) (openaimodel.py:736): dual-source K loop
    h = run_block(c, b, h, &skip, proj, proj_ld, io.t_shared);
  }
  Act hn = groupnorm_fwd(c, out_norm_, h, nullptr, true);
  ConvOpts oo; oo.out_f32 = true; oo.out = io.out; oo.out_ld = io.out_ld;
  conv_fwd(c, *out_conv_, hn, nullptr, oo);
  c.arena->release(mk0);
  c.f32 = false; c.x3 = false;
}

}  // namespace

std::unique_ptr<UNet> make_unet_openai(const cd_net_desc& d) { return std::unique_ptr<UNet>(new UNetOpenAI(d)); }

}  // namespace cd
