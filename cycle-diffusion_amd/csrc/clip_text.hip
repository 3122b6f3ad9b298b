// Text conditioning encoders: pre-LN token transformers on the engine's kernels.
//
// Variant CLIP (Stable Diffusion):
//
// Reference call site: FrozenCLIPEmbedder.forward (ldm/modules/encoders/modules.py:136-161) returns
// `CLIPTextModel(...).last_hidden_state` of HF transformers (pinned 4.19.2 by the reference's environment;
// not vendored) for "openai/clip-vit-large-patch14": 12 pre-LN layers, width 768, 12 heads, MLP 3072 with
// quick-GELU, causal attention over 77 positions, final LayerNorm. Algorithm restated from
// transformers/models/clip/modeling_clip.py (CLIPTextEmbeddings, CLIPAttention, CLIPMLP, CLIPEncoderLayer,
// CLIPTextTransformer); weights are keyed by that module's state_dict names (the names found under
// `cond_stage_model.transformer.` in an SD checkpoint).
//
// Variant BERT_XTR (LDM text2img-large): BERTEmbedder.transformer = TransformerWrapper(num_tokens 30522, 77
// positions, Encoder(dim 1280, depth 32)) of the vendored x-transformers copy
// (model/lib/latentdiff/ldm/modules/encoders/modules.py:75-98; ldm/modules/x_transformer.py: Attention :215-330
// - 8 heads x 64 whatever the width, bias-free q/k/v, no mask; FeedForward :194-211 - Linear, exact GELU, Linear;
// AttentionLayers pre-norm residual blocks :371-560; TransformerWrapper.forward :600-640 with return_embeddings);
// weights keyed by that module's state_dict names (`token_emb`, `pos_emb.emb`, `attn_layers.layers.{j}.{0,1}`, `norm`).
//
// Variants OCLIP_TEXT / OCLIP_VISION: the two towers of OpenAI CLIP ViT-B/32 that
// DirectionalCLIP ranks ensemble candidates with (model/energy/clean_clip.py:7-41 -> `clip.load("ViT-B/32")`, the
// un-vendored openai/CLIP package; algorithm restated from its clip/model.py: VisionTransformer.forward - conv1
// patch embedding, class token, positional embedding, ln_pre, transformer, ln_post(x[:,0]) @ proj; CLIP.encode_text -
// token + positional embedding, causal transformer, ln_final, x[arange, text.argmax(-1)] @ text_projection;
// ResidualAttentionBlock with nn.MultiheadAttention (fused in_proj) and QuickGELU). Weights are keyed by that
// package's state_dict names (`visual.*` for the image tower, unprefixed for the text tower).
//
// Every contraction runs on the implicit-GEMM kernel (conv_gemm.hip), attention on the flash kernel with its
// causal mask (attn.hip); residual adds, biases and quick-GELU live in GEMM epilogues.
#include "engine.h"

namespace cd {

namespace {

struct ClipLayer {
  LNW ln1, ln2;
  ConvW *qk = nullptr, *v = nullptr, *o = nullptr, *fc1 = nullptr, *fc2 = nullptr;
  float* vbias = nullptr;
};

LNW mk_ln(ParamStore& ps, const std::string& pfx, int C) {
  LNW g; g.C = C;
  g.g = ps.new_vec(C, 1.f); g.b = ps.new_vec(C, 0.f);
  ps.vec(pfx + ".weight", g.g, C);
  ps.vec(pfx + ".bias", g.b, C);
  return g;
}
ConvW* mk_linear(ParamStore& ps, const std::string& pfx, int N, int K) {
  ConvW* c = ps.new_conv(N, K, 1, 1, true);
  ps.conv_weight(pfx + ".weight", c, 2);
  ps.conv_bias(pfx + ".bias", c);
  return c;
}

// V^T[b] = Wv . X[b]^T : weights as the A operand, tokens as the B operand -> [B][C][Tpad]
void vt_gemm(Ctx& c, const ConvW& wv, const bf16_t* x, int ldx, int B, int T, int Tpad, bf16_t* vt) {
  ConvGemmParams p;
  p.src0 = wv.w; p.C0 = wv.Cpad; p.ld0 = wv.Cpad;
  p.B = 1; p.Hs = wv.N; p.Ws = 1; p.Hin = wv.N; p.Win = 1; p.Hout = wv.N; p.Wout = 1;
  p.M = wv.N;
  p.wgt = x; p.Ktot = wv.Cpad; p.ldw = ldx; p.N = T;
  p.nbatch = B; p.a_bs = 0; p.w_bs = (int64_t)T * ldx; p.o_bs = (int64_t)wv.N * Tpad;
  p.out = vt; p.out_ld = Tpad; p.zeros = c.zeros;
  launch_conv_gemm(c.st, p);
}

ClipLayer mk_layer_openai(ParamStore& ps, const std::string& lp, int D, int mlp) {
  ClipLayer L;
  L.ln1 = mk_ln(ps, lp + ".ln_1", D);
  L.ln2 = mk_ln(ps, lp + ".ln_2", D);
  // nn.MultiheadAttention: in_proj_weight [3D, D] = [Wq; Wk; Wv], in_proj_bias [3D]
  L.qk = ps.new_conv(2 * D, D, 1, 1, true);
  ps.conv_rows(lp + ".attn.in_proj_weight", {3 * D, D}, L.qk, 0, D, 0, D, 0);
  ps.conv_rows(lp + ".attn.in_proj_weight", {3 * D, D}, L.qk, D, D, D, D, 0);
  ps.bias_rows(lp + ".attn.in_proj_bias", 3 * D, L.qk->b, 0, D, 0, D, 0);
  ps.bias_rows(lp + ".attn.in_proj_bias", 3 * D, L.qk->b, D, D, D, D, 0);
  L.v = ps.new_conv(D, D, 1, 1, false);
  ps.conv_rows(lp + ".attn.in_proj_weight", {3 * D, D}, L.v, 0, D, 2 * D, D, 0);
  L.vbias = ps.new_vec(D);
  ps.bias_rows(lp + ".attn.in_proj_bias", 3 * D, L.vbias, 0, D, 2 * D, D, 0);
  L.o = mk_linear(ps, lp + ".attn.out_proj", D, D);
  L.fc1 = mk_linear(ps, lp + ".mlp.c_fc", mlp, D);
  L.fc2 = mk_linear(ps, lp + ".mlp.c_proj", D, mlp);
  return L;
}

class ClipText : public TextEncoder {
 public:
  explicit ClipText(const cd_net_desc& d) {
    desc = d;
    if (d.kind == CD_NET_OCLIP_TEXT || d.kind == CD_NET_OCLIP_VISION) { init_openai(d); return; }
    xtr_ = d.kind == CD_NET_BERT_XTR;
    width_ = d.model_channels; layers_n_ = d.num_res_blocks; heads_ = d.num_heads;
    mlp_ = d.context_dim; vocab_ = d.in_channels; maxpos_ = d.image_size;
    // x-transformers fixes heads x dim_head independently of the width (8 x 64 = 512 for a 1280-wide model)
    inner_ = xtr_ ? heads_ * d.num_head_channels : width_;
    causal_ = xtr_ ? 0 : 1;
    act_ = xtr_ ? ACT_GELU : ACT_QGELU;
    CD_CHECK(width_ > 0 && width_ % 64 == 0 && heads_ > 0 && inner_ % heads_ == 0 && inner_ % 64 == 0 &&
                 (inner_ / heads_) % 8 == 0 && inner_ / heads_ <= 160,
             "text encoder: width %d / heads %d / inner %d unsupported", width_, heads_, inner_);
    CD_CHECK(layers_n_ > 0 && mlp_ % 64 == 0 && vocab_ > 0 && maxpos_ > 0, "text encoder: bad descriptor");
    const int D = width_, I = inner_;
    tok_ = params.new_vec(vocab_ * width_);
    pos_ = params.new_vec(maxpos_ * width_);
    if (!xtr_) {
      const std::string root = "text_model.";
      params.mat_f32(root + "embeddings.token_embedding.weight", tok_, vocab_, D);
      params.mat_f32(root + "embeddings.position_embedding.weight", pos_, maxpos_, D);
      for (int i = 0; i < layers_n_; ++i) {
        const std::string lp = root + "encoder.layers." + std::to_string(i);
        ClipLayer L;
        L.ln1 = mk_ln(params, lp + ".layer_norm1", D);
        L.ln2 = mk_ln(params, lp + ".layer_norm2", D);
        L.qk = params.new_conv(2 * D, D, 1, 1, true);
        params.conv_rows(lp + ".self_attn.q_proj.weight", {D, D}, L.qk, 0, D, 0, D, 0);
        params.conv_rows(lp + ".self_attn.k_proj.weight", {D, D}, L.qk, D, D, 0, D, 0);
        params.bias_rows(lp + ".self_attn.q_proj.bias", D, L.qk->b, 0, D, 0, D, 0);
        params.bias_rows(lp + ".self_attn.k_proj.bias", D, L.qk->b, D, D, 0, D, 0);
        L.v = params.new_conv(D, D, 1, 1, false);
        params.conv_weight(lp + ".self_attn.v_proj.weight", L.v, 2);
        L.vbias = params.new_vec(D);  // added to the attention output: rows of softmax(.) sum to 1
        params.vec(lp + ".self_attn.v_proj.bias", L.vbias, D);
        L.o = mk_linear(params, lp + ".self_attn.out_proj", D, D);
        L.fc1 = mk_linear(params, lp + ".mlp.fc1", mlp_, D);
        L.fc2 = mk_linear(params, lp + ".mlp.fc2", D, mlp_);
        layers_.push_back(L);
      }
      final_ = mk_ln(params, root + "final_layer_norm", D);
    } else {
      params.mat_f32("token_emb.weight", tok_, vocab_, D);
      params.mat_f32("pos_emb.emb.weight", pos_, maxpos_, D);
      for (int i = 0; i < layers_n_; ++i) {
        const std::string la = "attn_layers.layers." + std::to_string(2 * i);      // (LayerNorm, Attention, Residual)
        const std::string lf = "attn_layers.layers." + std::to_string(2 * i + 1);  // (LayerNorm, FeedForward, Residual)
        ClipLayer L;
        L.ln1 = mk_ln(params, la + ".0", D);
        L.ln2 = mk_ln(params, lf + ".0", D);
        L.qk = params.new_conv(2 * I, D, 1, 1, false);
        params.conv_rows(la + ".1.to_q.weight", {I, D}, L.qk, 0, I, 0, I, 0);
        params.conv_rows(la + ".1.to_k.weight", {I, D}, L.qk, I, I, 0, I, 0);
        L.v = params.new_conv(I, D, 1, 1, false);
        params.conv_weight(la + ".1.to_v.weight", L.v, 2);
        L.o = mk_linear(params, la + ".1.to_out", D, I);
        L.fc1 = mk_linear(params, lf + ".1.net.0.0", mlp_, D);
        L.fc2 = mk_linear(params, lf + ".1.net.2", D, mlp_);
        layers_.push_back(L);
      }
      final_ = mk_ln(params, "norm", D);
    }
  }
  void init_openai(const cd_net_desc& d) {
    vision_ = d.kind == CD_NET_OCLIP_VISION;
    pooled_ = true;
    width_ = d.model_channels; layers_n_ = d.num_res_blocks; heads_ = d.num_heads; mlp_ = d.context_dim;
    embed_ = d.out_channels; inner_ = width_; act_ = ACT_QGELU; causal_ = vision_ ? 0 : 1;
    const int D = width_;
    CD_CHECK(D > 0 && D % 64 == 0 && heads_ > 0 && D % heads_ == 0 && (D / heads_) % 8 == 0 && D / heads_ <= 160 &&
                 layers_n_ > 0 && mlp_ % 64 == 0 && embed_ > 0,
             "clip tower: bad descriptor");
    const std::string root = vision_ ? "visual." : "";
    if (vision_) {
      res_ = d.image_size; patch_ = d.z_channels;
      CD_CHECK(patch_ > 0 && res_ % patch_ == 0 && d.in_channels == 3, "clip vision: resolution %d / patch %d", res_, patch_);
      maxpos_ = (res_ / patch_) * (res_ / patch_) + 1;
      conv1_ = params.new_conv(D, 3, patch_, patch_, false);
      params.conv_weight(root + "conv1.weight", conv1_, 4);
      cls_ = params.new_vec(D);
      params.vec(root + "class_embedding", cls_, D);
      pos_ = params.new_vec(maxpos_ * D);
      params.mat_f32(root + "positional_embedding", pos_, maxpos_, D);
      pre_ = mk_ln(params, root + "ln_pre", D);
      final_ = mk_ln(params, root + "ln_post", D);
      proj_ = params.new_conv(embed_, D, 1, 1, false);
      params.conv_weight_t(root + "proj", proj_);
    } else {
      vocab_ = d.in_channels; maxpos_ = d.image_size;
      tok_ = params.new_vec(vocab_ * D);
      pos_ = params.new_vec(maxpos_ * D);
      params.mat_f32("token_embedding.weight", tok_, vocab_, D);
      params.mat_f32("positional_embedding", pos_, maxpos_, D);
      final_ = mk_ln(params, "ln_final", D);
      proj_ = params.new_conv(embed_, D, 1, 1, false);
      params.conv_weight_t("text_projection", proj_);
    }
    for (int i = 0; i < layers_n_; ++i)
      layers_.push_back(mk_layer_openai(params, root + "transformer.resblocks." + std::to_string(i), D, mlp_));
  }
  int kind() const override { return desc.kind; }
  int width() const override { return width_; }
  int max_positions() const override { return maxpos_; }

  // ids [B][L] int32 (device) -> last_hidden_state fp32 [B][L][width]
  void encode(Ctx& c, const int* ids, int B, int L, float* out) override {
    CD_CHECK(L > 0 && L <= maxpos_, "clip text: sequence length %d exceeds %d positions", L, maxpos_);
    const size_t mk = c.arena->mark();
    CD_CHECK(!vision_, "an image tower takes images: cd_clip_image_features");
    const int D = width_;
    Act h = alloc_act(c, B, L, 1, D);
    launch_embed_tokens(c.st, ids, tok_, pos_, h.p, B, L, D, vocab_);
    run_layers(c, h, B, L);
    Act y = layernorm_fwd(c, final_, h);
    launch_nhwc_to_nchw(c.st, y.p, 0, y.ld, out, B * L, D, 1, 1.f, 0.f);  // 16-bit rows -> fp32 [B][L][D]
    c.arena->release(mk);
  }

  // CLIP.encode_text (clip/model.py): ids [B][L] -> features fp32 [B][embed]
  void text_features(Ctx& c, const int* ids, int B, int L, float* out) override {
    CD_CHECK(pooled_ && !vision_, "net is not an OpenAI-CLIP text tower");
    CD_CHECK(L > 0 && L <= maxpos_, "clip text: sequence length %d exceeds %d positions", L, maxpos_);
    const size_t mk = c.arena->mark();
    const int D = width_;
    Act h = alloc_act(c, B, L, 1, D);
    launch_embed_tokens(c.st, ids, tok_, pos_, h.p, B, L, D, vocab_);
    run_layers(c, h, B, L);
    Act e = alloc_act(c, B, 1, 1, D);
    launch_gather_eot(c.st, h.p, ids, e.p, B, L, D);
    project(c, e, B, out);
    c.arena->release(mk);
  }

  // VisionTransformer.forward (clip/model.py): img [B][3][R][R] fp32, already resized / cropped / normalised
  void image_features(Ctx& c, const float* img, int B, float* out) override {
    CD_CHECK(pooled_ && vision_, "net is not an OpenAI-CLIP image tower");
    const size_t mk = c.arena->mark();
    const int D = width_, g = res_ / patch_, T = g * g;
    Act x = alloc_act(c, B, res_, res_, 32);
    launch_nchw_to_nhwc(c.st, img, x.p, B, 3, res_ * res_, 32, 1.f, 0.f, 0);
    ConvOpts oc; oc.pad = 0; oc.stride = patch_;
    Act pe = conv_fwd(c, *conv1_, x, nullptr, oc);  // [B*T][D]
    Act h = alloc_act(c, B, T + 1, 1, D);
    launch_vit_tokens(c.st, pe.p, cls_, pos_, h.p, B, T, D);
    Act hp = layernorm_fwd(c, pre_, h);
    run_layers(c, hp, B, T + 1);
    // ln_post(x[:, 0]): one strided row per image
    Act e = alloc_act(c, B, 1, 1, D);
    launch_layernorm(c.st, hp.p, (T + 1) * D, e.p, D, B, D, final_.g, final_.b, 1e-5f);
    project_raw(c, e, B, out);
    c.arena->release(mk);
  }

 private:
  void project(Ctx& c, const Act& e, int B, float* out) {  // ln_final then @ text_projection
    Act n = layernorm_fwd(c, final_, e);
    project_raw(c, n, B, out);
  }
  void project_raw(Ctx& c, const Act& e, int B, float* out) {
    ConvOpts o; o.pad = 0; o.out_f32 = true; o.out = out; o.out_ld = embed_;
    conv_fwd(c, *proj_, e, nullptr, o);
  }

  // the pre-LN residual layers, in place on h [B][L][D]
  void run_layers(Ctx& c, Act& h, int B, int L) {
    const size_t mk = c.arena->mark();
    const int D = width_, I = inner_, dh = I / heads_;
    const int Tpad = round_up(L, 64);
    bf16_t* vt = (bf16_t*)c.arena->alloc((size_t)B * I * Tpad * 2);
    HIP_CHECK(hipMemsetAsync(vt, 0, (size_t)B * I * Tpad * 2, c.st));
    const float scale = 1.0f / sqrtf((float)dh);  // CLIPAttention: q * head_dim**-0.5; x-transformers: dim_head**-0.5
    ConvOpts p0; p0.pad = 0;
    for (const ClipLayer& Lw : layers_) {
      const size_t m2 = c.arena->mark();
      Act n1 = layernorm_fwd(c, Lw.ln1, h);
      Act qk = conv_fwd(c, *Lw.qk, n1, nullptr, p0);  // [B*L][2*inner]
      vt_gemm(c, *Lw.v, n1.p, n1.ld, B, L, Tpad, vt);
      Act a = alloc_act(c, B, L, 1, I);
      AttnParams ap;
      ap.q = qk.p; ap.k = qk.p + I; ap.vt = vt; ap.o = a.p;
      ap.B = B; ap.H = heads_; ap.Tq = L; ap.Tk = L; ap.D = dh;
      ap.ldq = qk.ld; ap.ldk = qk.ld; ap.ldo = a.ld;
      ap.q_bs = (int64_t)L * qk.ld; ap.k_bs = (int64_t)L * qk.ld; ap.o_bs = (int64_t)L * a.ld;
      ap.vt_dpad = dh; ap.vt_tpad = Tpad; ap.scale = scale; ap.obias = Lw.vbias; ap.causal = causal_;
      launch_attention(c.st, ap);
      ConvOpts o; o.pad = 0; o.resid = &h; o.out = h.p; o.out_ld = h.ld;  // h += out_proj(attn), in place
      conv_fwd(c, *Lw.o, a, nullptr, o);
      Act n2 = layernorm_fwd(c, Lw.ln2, h);
      ConvOpts f1; f1.pad = 0; f1.act = act_;
      Act g = conv_fwd(c, *Lw.fc1, n2, nullptr, f1);
      ConvOpts f2; f2.pad = 0; f2.resid = &h; f2.out = h.p; f2.out_ld = h.ld;  // h += fc2(quick_gelu(fc1))
      conv_fwd(c, *Lw.fc2, g, nullptr, f2);
      c.arena->release(m2);
    }
    c.arena->release(mk);
  }

  bool xtr_ = false, vision_ = false, pooled_ = false;
  int embed_ = 0, res_ = 0, patch_ = 0;
  ConvW *conv1_ = nullptr, *proj_ = nullptr;
  float* cls_ = nullptr;
  LNW pre_;
  int width_ = 0, layers_n_ = 0, heads_ = 0, mlp_ = 0, vocab_ = 0, maxpos_ = 0, inner_ = 0, causal_ = 1, act_ = ACT_QGELU;
  float *tok_ = nullptr, *pos_ = nullptr;
  std::vector<ClipLayer> layers_;
  LNW final_;
};

}  // namespace

std::unique_ptr<TextEncoder> make_clip_text(const cd_net_desc& d) { return std::unique_ptr<TextEncoder>(new ClipText(d)); }

}  // namespace cd
