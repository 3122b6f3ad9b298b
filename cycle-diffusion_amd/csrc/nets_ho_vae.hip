// Executors for the pytorch_diffusion-style blocks shared by
//   * the KL-f8 autoencoder: Encoder / Decoder / AutoencoderKL.encode|decode
//     (ldm/modules/diffusionmodules/model.py:368-568, ldm/models/autoencoder.py:324-333)
//   * the Ho et al. DDPM U-Net (model/lib/ddpm_ddim/models/ddpm/diffusion.py:192-337)
// ResnetBlock: GN(eps 1e-6) -> swish -> conv3x3 (+ temb_proj(swish(temb))) -> GN -> swish -> conv3x3,
// 1x1 nin_shortcut when channels change; AttnBlock: single head over H*W tokens with d = C.
#include <limits.h>

#include <algorithm>

#include "engine.h"

namespace cd {

namespace {

GNW mk_gn(ParamStore& ps, const std::string& pfx, int C) {
  GNW g; g.C = C; g.eps = 1e-6f;
  g.g = ps.new_vec(C, 1.f); g.b = ps.new_vec(C, 0.f);
  ps.vec(pfx + ".weight", g.g, C);
  ps.vec(pfx + ".bias", g.b, C);
  return g;
}
ConvW* mk_conv(ParamStore& ps, const std::string& pfx, int N, int Cin, int k, bool bias = true) {
  ConvW* c = ps.new_conv(N, Cin, k, k, bias);
  ps.conv_weight(pfx + ".weight", c, 4);
  if (bias) ps.conv_bias(pfx + ".bias", c);
  return c;
}

struct RnW {
  int cin = 0, cout = 0;
  GNW n1, n2;
  ConvW *c1 = nullptr, *c2 = nullptr, *nin = nullptr;
  int emb_off = -1;  // offset into the fused temb projection (Ho-DDPM); -1 = no temb (VAE)
};
struct AtW {
  int C = 0;
  GNW norm;
  ConvW *qk = nullptr, *v = nullptr, *proj = nullptr;
  float* vbias = nullptr;
};

RnW mk_rn(ParamStore& ps, const std::string& pfx, int cin, int cout) {
  RnW r; r.cin = cin; r.cout = cout;
  r.n1 = mk_gn(ps, pfx + ".norm1", cin);
  r.c1 = mk_conv(ps, pfx + ".conv1", cout, cin, 3);
  r.n2 = mk_gn(ps, pfx + ".norm2", cout);
  r.c2 = mk_conv(ps, pfx + ".conv2", cout, cout, 3);
  if (cin != cout) r.nin = mk_conv(ps, pfx + ".nin_shortcut", cout, cin, 1);
  return r;
}
AtW mk_at(ParamStore& ps, const std::string& pfx, int C) {
  AtW a; a.C = C;
  a.norm = mk_gn(ps, pfx + ".norm", C);
  a.qk = ps.new_conv(2 * C, C, 1, 1, true);
  ps.conv_rows(pfx + ".q.weight", {C, C, 1, 1}, a.qk, 0, C, 0, C, 0);
  ps.conv_rows(pfx + ".k.weight", {C, C, 1, 1}, a.qk, C, C, 0, C, 0);
  ps.bias_rows(pfx + ".q.bias", C, a.qk->b, 0, C, 0, C, 0);
  ps.bias_rows(pfx + ".k.bias", C, a.qk->b, C, C, 0, C, 0);
  a.v = ps.new_conv(C, C, 1, 1, false);
  ps.conv_weight(pfx + ".v.weight", a.v, 4);
  a.vbias = ps.new_vec(C);
  ps.vec(pfx + ".v.bias", a.vbias, C);
  a.proj = mk_conv(ps, pfx + ".proj_out", C, C, 1);
  return a;
}

Act rn_fwd(Ctx& c, const RnW& r, const Act& x, const Act* x2, const float* proj, int proj_ld, bool t_shared) {
  Act out = alloc_act(c, x.B, x.H, x.W, r.cout, /*with_stats=*/true);
  const size_t mk = c.arena->mark();
  Act h = groupnorm_fwd(c, r.n1, x, x2, true);
  ConvOpts o1; o1.want_stats = true;
  if (r.emb_off >= 0) {
    o1.rowvec = proj + r.emb_off; o1.rowvec_ld = proj_ld;
    o1.rows_per_vec = t_shared ? INT_MAX : x.H * x.W;
  }
  Act h2 = conv_fwd(c, *r.c1, h, nullptr, o1);
  Act h3 = groupnorm_fwd(c, r.n2, h2, nullptr, true);
  Act skip;
  if (r.nin) { ConvOpts os; os.pad = 0; skip = conv_fwd(c, *r.nin, x, x2, os); }
  else { CD_CHECK(!x2, "identity shortcut with concat input"); skip = x; }
  ConvOpts o2; o2.resid = &skip; o2.out = out.p; o2.out_ld = out.ld; o2.out_stats = out.stats_buf;
  conv_fwd(c, *r.c2, h3, nullptr, o2);
  out.stats = out.stats_buf;
  c.arena->release(mk);
  return out;
}

void vt_gemm2(Ctx& c, const ConvW& wv, const bf16_t* x, int ldx, int B, int T, int Tpad, bf16_t* vt) {
  ConvGemmParams p;
  p.src0 = wv.w; p.C0 = wv.Cpad; p.ld0 = wv.Cpad;
  p.B = 1; p.Hs = wv.N; p.Ws = 1; p.Hin = wv.N; p.Win = 1; p.Hout = wv.N; p.Wout = 1;
  p.M = wv.N;
  p.wgt = x; p.Ktot = wv.Cpad; p.ldw = ldx; p.N = T;
  p.nbatch = B; p.a_bs = 0; p.w_bs = (int64_t)T * ldx; p.o_bs = (int64_t)wv.N * Tpad;
  p.out = vt; p.out_ld = Tpad; p.zeros = c.zeros;
  launch_conv_gemm(c.st, p);
}

Act at_fwd(Ctx& c, const AtW& a, const Act& x) {
  const int B = x.B, T = x.H * x.W, C = a.C;
  Act out = alloc_act(c, B, x.H, x.W, C, /*with_stats=*/true);
  const size_t mk = c.arena->mark();
  ConvOpts p0; p0.pad = 0;
  Act n = groupnorm_fwd(c, a.norm, x, nullptr, false);
  Act qk = conv_fwd(c, *a.qk, n, nullptr, p0);  // [B*T][2C]
  if (c.f32) {  // fp32 path (Ho-DDPM in CD_PREC_F32): single head of width C
    Act v = conv_fwd(c, *a.v, n, nullptr, p0);
    Act o = attention_f32_fwd(c, qk, v, 1, C, 1.0f / sqrtf((float)C), a.vbias);
    ConvOpts po; po.pad = 0; po.resid = &x; po.out = out.p; po.out_ld = out.ld;
    conv_fwd(c, *a.proj, o, nullptr, po);
    c.arena->release(mk);
    return out;
  }
  const int Tpad = round_up(T, 64);
  bf16_t* vt = (bf16_t*)c.arena->alloc((size_t)B * C * Tpad * 2);
  if (Tpad != T) HIP_CHECK(hipMemsetAsync(vt, 0, (size_t)B * C * Tpad * 2, c.st));
  vt_gemm2(c, *a.v, n.p, n.ld, B, T, Tpad, vt);
  const float scale = 1.0f / sqrtf((float)C);  // w_ * int(c)**(-0.5), model.py:192
  Act o = alloc_act(c, B, x.H, x.W, C);
  if (C <= 160) {
    AttnParams p;
    p.q = qk.p; p.k = qk.p + C; p.vt = vt; p.o = o.p;
    p.B = B; p.H = 1; p.Tq = T; p.Tk = T; p.D = C;
    p.ldq = qk.ld; p.ldk = qk.ld; p.ldo = o.ld;
    p.q_bs = (int64_t)T * qk.ld; p.k_bs = (int64_t)T * qk.ld; p.o_bs = (int64_t)T * o.ld;
    p.vt_dpad = C; p.vt_tpad = Tpad; p.scale = scale; p.obias = a.vbias;
    launch_attention(c.st, p);
  } else {
    // wide single head (VAE mid block: 4096 tokens x 512 ch): scores materialised once per image,
    // S = QK^T and O = PV on the MFMA GEMM kernel, fp32 row softmax in between
    CD_CHECK(T % 32 == 0, "AttnBlock: token count %d must be a multiple of 32", T);
    float* S = (float*)c.arena->alloc((size_t)B * T * T * 4);
    bf16_t* P = (bf16_t*)c.arena->alloc((size_t)B * T * T * 2);
    ConvGemmParams g;
    g.src0 = qk.p; g.C0 = C; g.ld0 = qk.ld; g.B = 1; g.Hs = T; g.Ws = 1; g.Hin = T; g.Win = 1;
    g.Hout = T; g.Wout = 1; g.M = T;
    g.wgt = qk.p + C; g.Ktot = C; g.ldw = qk.ld; g.N = T;
    g.nbatch = B; g.a_bs = (int64_t)T * qk.ld; g.w_bs = (int64_t)T * qk.ld; g.o_bs = (int64_t)T * T;
    g.alpha = scale; g.out = S; g.out_ld = T; g.out_f32 = 1; g.zeros = c.zeros;
    launch_conv_gemm(c.st, g);
    launch_softmax_rows(c.st, S, T, P, T, (int64_t)B * T, T);
    ConvGemmParams h;
    h.src0 = P; h.C0 = T; h.ld0 = T; h.B = 1; h.Hs = T; h.Ws = 1; h.Hin = T; h.Win = 1;
    h.Hout = T; h.Wout = 1; h.M = T;
    h.wgt = vt; h.Ktot = T; h.ldw = Tpad; h.N = C;
    h.nbatch = B; h.a_bs = (int64_t)T * T; h.w_bs = (int64_t)C * Tpad; h.o_bs = (int64_t)T * C;
    h.bias = a.vbias; h.out = o.p; h.out_ld = o.ld; h.zeros = c.zeros;
    launch_conv_gemm(c.st, h);
  }
  ConvOpts po; po.pad = 0; po.resid = &x; po.out = out.p; po.out_ld = out.ld; po.out_stats = out.stats_buf;
  conv_fwd(c, *a.proj, o, nullptr, po);
  out.stats = out.stats_buf;
  c.arena->release(mk);
  return out;
}

// ================================================================== KL autoencoder
class VAEKL : public VAE {
 public:
  explicit VAEKL(const cd_net_desc& d);
  void encode_moments(Ctx& c, const bf16_t* img, int B, int R, float* moments) override;
  void decode(Ctx& c, const bf16_t* z, int B, int h, float* img) override;
 private:
  int ch_, nres_, nlev_;
  std::vector<int> mult_;
  // encoder
  ConvW* e_in_ = nullptr; std::vector<std::vector<RnW>> e_blk_; std::vector<ConvW*> e_down_;
  RnW e_m1_, e_m2_; AtW e_at_; GNW e_no_; ConvW* e_out_ = nullptr; ConvW* quant_ = nullptr;
  // decoder
  ConvW* pq_ = nullptr; ConvW* d_in_ = nullptr; RnW d_m1_, d_m2_; AtW d_at_;
  std::vector<std::vector<RnW>> d_blk_; std::vector<ConvW*> d_up_; GNW d_no_; ConvW* d_out_ = nullptr;
  int in_ch_, out_ch_, embed_, moments_;
};

VAEKL::VAEKL(const cd_net_desc& d) {
  desc = d;
  // round 5: the first stage in the reference's arithmetic too (`precision = "full"` covers the VAE,
  // stable_diffusion_stochastic_text_wrapper.py:117, autoencoder.py:324-333): CD_PREC_F32 runs every layer on the fp32
  // path (f32_path.hip), CD_PREC_F32X3 runs the GroupNorm-fed convolutions - 97 % of the FLOPs - as three-term split-fp16
  // products and the rest (nin_shortcut, down / up-sampling convs on raw activations, the single-head attention) in fp32
  f32 = d.precision == CD_PREC_F32 || d.precision == CD_PREC_F32X3;
  x3 = d.precision == CD_PREC_F32X3;
  params.f32 = f32; params.x3 = x3;
  CD_CHECK(!(f32 && d.n_embed > 0), "the VQ first stage (codebook lookup on 16-bit rows) runs in the 16-bit format only");
  ch_ = d.model_channels; nres_ = d.num_res_blocks; nlev_ = d.n_mult;
  for (int i = 0; i < nlev_; ++i) mult_.push_back(d.channel_mult[i]);
  z_channels = d.z_channels; embed_ = d.embed_dim; in_ch_ = d.in_channels; out_ch_ = d.out_channels;
  factor = 1 << (nlev_ - 1);
  // AutoencoderKL: double_z, Gaussian posterior (autoencoder.py:285-333). VQModelInterface (n_embed > 0): the encoder
  // emits z_channels directly and decode() snaps to the codebook first (autoencoder.py:264-282)
  CD_CHECK((d.double_z != 0) != (d.n_embed > 0), "first stage must be either KL (double_z) or VQ (n_embed > 0)");
  CD_CHECK(d.n_attn == 0, "attn_resolutions inside the VAE levels are not used by the f4 / f8 configs");
  const int zmul = d.double_z ? 2 : 1;
  moments_ = zmul * z_channels;
  moments_channels = zmul * embed_;
  n_embed = d.n_embed;
  // ---- encoder (model.py:368-459)
  e_in_ = mk_conv(params, "encoder.conv_in", ch_, in_ch_, 3);
  int bin = ch_;
  e_blk_.resize(nlev_); e_down_.assign(nlev_, nullptr);
  for (int l = 0; l < nlev_; ++l) {
    const int bout = ch_ * mult_[l];
    for (int b = 0; b < nres_; ++b) {
      e_blk_[l].push_back(mk_rn(params, "encoder.down." + std::to_string(l) + ".block." + std::to_string(b), bin, bout));
      bin = bout;
    }
    if (l != nlev_ - 1) e_down_[l] = mk_conv(params, "encoder.down." + std::to_string(l) + ".downsample.conv", bin, bin, 3);
  }
  e_m1_ = mk_rn(params, "encoder.mid.block_1", bin, bin);
  e_at_ = mk_at(params, "encoder.mid.attn_1", bin);
  e_m2_ = mk_rn(params, "encoder.mid.block_2", bin, bin);
  e_no_ = mk_gn(params, "encoder.norm_out", bin);
  e_out_ = mk_conv(params, "encoder.conv_out", moments_, bin, 3);
  quant_ = mk_conv(params, "quant_conv", zmul * embed_, moments_, 1);
  if (n_embed > 0) {
    float* cb = params.new_vec(n_embed * embed_);
    params.mat_f32("quantize.embedding.weight", cb, n_embed, embed_);
    codebook = cb;
  }
  // ---- decoder (model.py:462-568)
  pq_ = mk_conv(params, "post_quant_conv", z_channels, embed_, 1);
  bin = ch_ * mult_[nlev_ - 1];
  d_in_ = mk_conv(params, "decoder.conv_in", bin, z_channels, 3);
  d_m1_ = mk_rn(params, "decoder.mid.block_1", bin, bin);
  d_at_ = mk_at(params, "decoder.mid.attn_1", bin);
  d_m2_ = mk_rn(params, "decoder.mid.block_2", bin, bin);
  d_blk_.resize(nlev_); d_up_.assign(nlev_, nullptr);
  for (int l = nlev_ - 1; l >= 0; --l) {
    const int bout = ch_ * mult_[l];
    for (int b = 0; b <= nres_; ++b) {
      d_blk_[l].push_back(mk_rn(params, "decoder.up." + std::to_string(l) + ".block." + std::to_string(b), bin, bout));
      bin = bout;
    }
    if (l != 0) d_up_[l] = mk_conv(params, "decoder.up." + std::to_string(l) + ".upsample.conv", bin, bin, 3);
  }
  d_no_ = mk_gn(params, "decoder.norm_out", bin);
  d_out_ = mk_conv(params, "decoder.conv_out", out_ch_, bin, 3);
}

void VAEKL::encode_moments(Ctx& c, const bf16_t* img, int B, int R, float* moments) {
  const size_t mk = c.arena->mark();
  c.f32 = f32; c.x3 = x3;
  const size_t esz = f32 ? 4 : 2;
  Act x; x.p = (bf16_t*)img; x.B = B; x.H = R; x.W = R; x.C = round_up(in_ch_, 32); x.ld = x.C; x.f32 = f32;
  if (x3) { x.split = true; x.ld = 2 * x.C; }  // the caller hands a split-mode network its input as fp16 pairs
  ConvOpts o3; o3.want_stats = true;
  Act h = conv_fwd(c, *e_in_, x, nullptr, o3);
  for (int l = 0; l < nlev_; ++l) {
    for (auto& r : e_blk_[l]) h = rn_fwd(c, r, h, nullptr, nullptr, 0, true);
    if (e_down_[l]) { ConvOpts od; od.stride = 2; od.asym = true; od.want_stats = true; h = conv_fwd(c, *e_down_[l], h, nullptr, od); }
  }
  h = rn_fwd(c, e_m1_, h, nullptr, nullptr, 0, true);
  h = at_fwd(c, e_at_, h);
  h = rn_fwd(c, e_m2_, h, nullptr, nullptr, 0, true);
  Act n = groupnorm_fwd(c, e_no_, h, nullptr, true);
  // conv_out -> bf16 padded to 32 channels so the 1x1 quant_conv can consume it
  const int cp = round_up(moments_, 32);
  Act mo = alloc_act(c, B, h.H, h.W, cp);
  HIP_CHECK(hipMemsetAsync(mo.p, 0, (size_t)mo.rows() * cp * esz, c.st));
  ConvOpts oo; oo.out = mo.p; oo.out_ld = cp;
  conv_fwd(c, *e_out_, n, nullptr, oo);
  ConvOpts oq; oq.pad = 0; oq.out_f32 = true; oq.out = moments; oq.out_ld = moments_channels;
  conv_fwd(c, *quant_, mo, nullptr, oq);
  c.arena->release(mk);
  c.f32 = false; c.x3 = false;
}

void VAEKL::decode(Ctx& c, const bf16_t* z, int B, int hl, float* img) {
  const size_t mk = c.arena->mark();
  c.f32 = f32; c.x3 = x3;
  Act x; x.p = (bf16_t*)z; x.B = B; x.H = hl; x.W = hl; x.C = round_up(embed_, 32); x.ld = x.C; x.f32 = f32;
  if (x3) { x.split = true; x.ld = 2 * x.C; }
  const int cp = round_up(z_channels, 32);
  Act zq = alloc_act(c, B, hl, hl, cp);
  HIP_CHECK(hipMemsetAsync(zq.p, 0, (size_t)zq.rows() * cp * (f32 ? 4 : 2), c.st));
  ConvOpts oq; oq.pad = 0; oq.out = zq.p; oq.out_ld = cp;
  conv_fwd(c, *pq_, x, nullptr, oq);
  ConvOpts o3; o3.want_stats = true;
  Act h = conv_fwd(c, *d_in_, zq, nullptr, o3);
  h = rn_fwd(c, d_m1_, h, nullptr, nullptr, 0, true);
  h = at_fwd(c, d_at_, h);
  h = rn_fwd(c, d_m2_, h, nullptr, nullptr, 0, true);
  for (int l = nlev_ - 1; l >= 0; --l) {
    for (auto& r : d_blk_[l]) h = rn_fwd(c, r, h, nullptr, nullptr, 0, true);
    if (d_up_[l]) { ConvOpts ou; ou.up = true; ou.want_stats = true; h = conv_fwd(c, *d_up_[l], h, nullptr, ou); }
  }
  Act n = groupnorm_fwd(c, d_no_, h, nullptr, true);
  ConvOpts oo; oo.out_f32 = true; oo.out = img; oo.out_ld = out_ch_;
  conv_fwd(c, *d_out_, n, nullptr, oo);
  c.arena->release(mk);
  c.f32 = false; c.x3 = false;
}

// ================================================================== Ho et al. DDPM U-Net
class UNetHo : public UNet {
 public:
  explicit UNetHo(const cd_net_desc& d);
  int kind() const override { return CD_NET_UNET_HO; }
  void forward(Ctx& c, const UNetIO& io) override;
  size_t workspace_hint(int B) const override {
    return (size_t)B * image_size * image_size * ch_ * 8 * 2 * 24 + (64u << 20);
  }
 private:
  int ch_, nres_, nlev_, temb_ch_;
  std::vector<int> mult_;
  TimeEmb te_;
  ConvW* cin_ = nullptr;
  struct Lvl { std::vector<RnW> blk; std::vector<AtW> att; ConvW* resamp = nullptr; };
  std::vector<Lvl> down_, up_;
  RnW m1_, m2_; AtW mat_;
  GNW no_; ConvW* cout_ = nullptr;
};

UNetHo::UNetHo(const cd_net_desc& d) {
  desc = d;
  f32 = d.precision == CD_PREC_F32 || d.precision == CD_PREC_F32X3;
  x3 = d.precision == CD_PREC_F32X3;
  params.f32 = f32; params.x3 = x3;
  ch_ = d.model_channels; nres_ = d.num_res_blocks; nlev_ = d.n_mult; temb_ch_ = 4 * ch_;
  image_size = d.image_size; out_channels = d.out_channels; in_cpad = round_up(d.in_channels, 32);
  CD_CHECK(ch_ % 32 == 0, "ch must be a multiple of 32");
  CD_CHECK(d.conv_resample, "resamp_with_conv=False is not used by the reference configs");
  for (int i = 0; i < nlev_; ++i) mult_.push_back(d.channel_mult[i]);
  auto has_attn = [&](int res) { for (int i = 0; i < d.n_attn; ++i) if (d.attn[i] == res) return true; return false; };
  std::vector<std::pair<std::string, RnW*>> temb_users;
  cin_ = mk_conv(params, "conv_in", ch_, d.in_channels, 3);
  int res = d.image_size, bin = ch_;
  std::vector<int> in_mult{1};
  for (int m : mult_) in_mult.push_back(m);
  down_.resize(nlev_);
  for (int l = 0; l < nlev_; ++l) {
    bin = ch_ * in_mult[l];
    const int bout = ch_ * mult_[l];
    down_[l].blk.reserve(nres_); down_[l].att.reserve(nres_);
    for (int b = 0; b < nres_; ++b) {
      const std::string pfx = "down." + std::to_string(l);
      down_[l].blk.push_back(mk_rn(params, pfx + ".block." + std::to_string(b), bin, bout));
      temb_users.push_back({pfx + ".block." + std::to_string(b), &down_[l].blk.back()});
      bin = bout;
      if (has_attn(res)) down_[l].att.push_back(mk_at(params, pfx + ".attn." + std::to_string(b), bin));
    }
    if (l != nlev_ - 1) {
      down_[l].resamp = mk_conv(params, "down." + std::to_string(l) + ".downsample.conv", bin, bin, 3);
      res /= 2;
    }
  }
  m1_ = mk_rn(params, "mid.block_1", bin, bin); temb_users.push_back({"mid.block_1", &m1_});
  mat_ = mk_at(params, "mid.attn_1", bin);
  m2_ = mk_rn(params, "mid.block_2", bin, bin); temb_users.push_back({"mid.block_2", &m2_});
  up_.resize(nlev_);
  for (int l = nlev_ - 1; l >= 0; --l) {
    const int bout = ch_ * mult_[l];
    int skip_in = ch_ * mult_[l];
    up_[l].blk.reserve(nres_ + 1); up_[l].att.reserve(nres_ + 1);
    for (int b = 0; b <= nres_; ++b) {
      if (b == nres_) skip_in = ch_ * in_mult[l];
      const std::string pfx = "up." + std::to_string(l);
      up_[l].blk.push_back(mk_rn(params, pfx + ".block." + std::to_string(b), bin + skip_in, bout));
      temb_users.push_back({pfx + ".block." + std::to_string(b), &up_[l].blk.back()});
      bin = bout;
      if (has_attn(res)) up_[l].att.push_back(mk_at(params, pfx + ".attn." + std::to_string(b), bin));
    }
    if (l != 0) {
      up_[l].resamp = mk_conv(params, "up." + std::to_string(l) + ".upsample.conv", bin, bin, 3);
      res *= 2;
    }
  }
  no_ = mk_gn(params, "norm_out", bin);
  cout_ = mk_conv(params, "conv_out", d.out_channels, bin, 3);
  // timestep embedding (diffusion.py:210-217, 294-297) + fused temb_proj of every ResnetBlock
  te_.mode = 1; te_.dim = ch_; te_.hidden = temb_ch_;
  te_.w0 = params.new_vec(temb_ch_ * ch_); te_.b0 = params.new_vec(temb_ch_);
  te_.w1 = params.new_vec(temb_ch_ * temb_ch_); te_.b1 = params.new_vec(temb_ch_);
  params.mat_f32("temb.dense.0.weight", te_.w0, temb_ch_, ch_);
  params.vec("temb.dense.0.bias", te_.b0, temb_ch_);
  params.mat_f32("temb.dense.1.weight", te_.w1, temb_ch_, temb_ch_);
  params.vec("temb.dense.1.bias", te_.b1, temb_ch_);
  int total = 0;
  for (auto& u : temb_users) { u.second->emb_off = total; total += u.second->cout; }
  te_.proj_total = total;
  te_.proj_w = params.new_vec(total * temb_ch_);
  te_.proj_b = params.new_vec(total);
  for (auto& u : temb_users) {
    params.mat_f32(u.first + ".temb_proj.weight", te_.proj_w, u.second->cout, temb_ch_, u.second->emb_off);
    params.bias_rows(u.first + ".temb_proj.bias", u.second->cout, te_.proj_b, u.second->emb_off, u.second->cout, 0,
                     u.second->cout, 0);
  }
}

void UNetHo::forward(Ctx& c, const UNetIO& io) {
  const size_t mk0 = c.arena->mark();
  c.f32 = f32; c.x3 = x3;
  const int B = io.B, R = image_size;
  const int tB = io.t_shared ? 1 : B;
  float* sinu = (float*)c.arena->alloc((size_t)tB * ch_ * 4);
  float* e1 = (float*)c.arena->alloc((size_t)tB * temb_ch_ * 4);
  float* emb = (float*)c.arena->alloc((size_t)tB * temb_ch_ * 4);
  float* proj = (float*)c.arena->alloc((size_t)tB * te_.proj_total * 4);
  launch_timestep_embedding(c.st, io.tab, io.step_ptr, io.step, io.t_explicit, sinu, tB, ch_, 1);
  launch_vec_linear(c.st, sinu, ch_, te_.w0, te_.b0, e1, temb_ch_, tB, ch_, temb_ch_, 0, 1);
  launch_vec_linear(c.st, e1, temb_ch_, te_.w1, te_.b1, emb, temb_ch_, tB, temb_ch_, temb_ch_, 0, 0);
  launch_vec_linear(c.st, emb, temb_ch_, te_.proj_w, te_.proj_b, proj, te_.proj_total, tB, temb_ch_,
                    te_.proj_total, 1, 0);
  const int pl = te_.proj_total;
  Act x; x.p = (bf16_t*)io.xin; x.B = B; x.H = R; x.W = R; x.C = in_cpad; x.ld = in_cpad; x.f32 = f32;
  if (x3) { x.split = true; x.ld = 2 * in_cpad; }  // the samplers hand CD_PREC_F32X3 networks their input as fp16 pairs
  ConvOpts o3; o3.want_stats = true;
  std::vector<Act> hs;
  hs.push_back(conv_fwd(c, *cin_, x, nullptr, o3));
  for (int l = 0; l < nlev_; ++l) {
    for (int b = 0; b < nres_; ++b) {
      Act h = rn_fwd(c, down_[l].blk[b], hs.back(), nullptr, proj, pl, io.t_shared);
      if (!down_[l].att.empty()) h = at_fwd(c, down_[l].att[b], h);
      hs.push_back(h);
    }
    if (down_[l].resamp) {
      ConvOpts od; od.stride = 2; od.asym = true; od.want_stats = true;
      hs.push_back(conv_fwd(c, *down_[l].resamp, hs.back(), nullptr, od));
    }
  }
  Act h = hs.back();
  h = rn_fwd(c, m1_, h, nullptr, proj, pl, io.t_shared);
  h = at_fwd(c, mat_, h);
  h = rn_fwd(c, m2_, h, nullptr, proj, pl, io.t_shared);
  for (int l = nlev_ - 1; l >= 0; --l) {
    for (int b = 0; b <= nres_; ++b) {
      Act skip = hs.back(); hs.pop_back();
      h = rn_fwd(c, up_[l].blk[b], h, &skip, proj, pl, io.t_shared);
      if (!up_[l].att.empty()) h = at_fwd(c, up_[l].att[b], h);
    }
    if (up_[l].resamp) { ConvOpts ou; ou.up = true; ou.want_stats = true; h = conv_fwd(c, *up_[l].resamp, h, nullptr, ou); }
  }
  Act n = groupnorm_fwd(c, no_, h, nullptr, true);
  ConvOpts oo; oo.out_f32 = true; oo.out = io.out; oo.out_ld = io.out_ld;
  conv_fwd(c, *cout_, n, nullptr, oo);
  c.arena->release(mk0);
  c.f32 = false; c.x3 = false;
}

}  // namespace

std::unique_ptr<VAE> make_vae_kl(const cd_net_desc& d) { return std::unique_ptr<VAE>(new VAEKL(d)); }
std::unique_ptr<UNet> make_unet_ho(const cd_net_desc& d) { return std::unique_ptr<UNet>(new UNetHo(d)); }

}  // namespace cd
