// Arena, parameter store and the shared building blocks (conv / norm / attention wrappers).
#include <stdlib.h>

#include "engine.h"

#include <string.h>

namespace cd {

// ------------------------------------------------------------------ Arena
Arena::~Arena() { if (base_) (void)hipFree(base_); }
void Arena::init(size_t bytes) {
  if (base_) { (void)hipFree(base_); base_ = nullptr; }
  HIP_CHECK(hipMalloc((void**)&base_, bytes));
  cap_ = bytes; off_ = 0; high_ = 0;
}
void* Arena::alloc(size_t bytes) {
  size_t a = (off_ + 255) & ~(size_t)255;
  CD_CHECK(a + bytes <= cap_, "workspace arena exhausted: need %zu more bytes (capacity %zu); "
           "create the engine with a larger workspace", a + bytes - cap_, cap_);
  off_ = a + bytes;
  if (off_ > high_) high_ = off_;
  return base_ + a;
}

// ------------------------------------------------------------------ ParamStore
ParamStore::~ParamStore() {
  for (void* p : allocs_) (void)hipFree(p);
  if (staging_) (void)hipFree(staging_);
}
void* ParamStore::dmalloc(size_t bytes) {
  void* p = nullptr;
  HIP_CHECK(hipMalloc(&p, bytes < 256 ? 256 : bytes));
  HIP_CHECK(hipMemset(p, 0, bytes < 256 ? 256 : bytes));
  allocs_.push_back(p);
  dev_bytes_ += bytes;
  return p;
}
ConvW* ParamStore::new_conv(int N, int Cin, int KH, int KW, bool bias, bool geglu) {
  convs_.emplace_back(new ConvW());
  ConvW* c = convs_.back().get();
  c->N = N; c->Cin = Cin; c->KH = KH; c->KW = KW; c->geglu = geglu;
  c->Cpad = round_up(Cin, 32);
  c->Npad = round_up(N, 128);
  c->f32 = f32;
  c->w = (bf16_t*)dmalloc((size_t)c->Npad * c->Ktot() * (f32 ? sizeof(float) : sizeof(bf16_t)));
  if (!f32 && KH == 1 && KW == 1 && c->Cpad == 320)  // the streaming linear kernel reads fragment-major weights
    c->wfrag = (bf16_t*)dmalloc((size_t)c->Npad * c->Cpad * sizeof(bf16_t));
  if (f32 && x3) c->w3 = (bf16_t*)dmalloc((size_t)c->Npad * c->Ktot() * 3 * sizeof(bf16_t));
  if (bias) c->b = (float*)dmalloc((size_t)c->Npad * sizeof(float));
  return c;
}
float* ParamStore::new_vec(int n, float init) {
  float* p = (float*)dmalloc((size_t)n * sizeof(float));
  if (init != 0.f) {
    std::vector<float> h(n, init);
    HIP_CHECK(hipMemcpy(p, h.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  }
  return p;
}
ParamDecl& ParamStore::declare(const std::string& name, std::vector<int64_t> shape) {
  auto it = by_name_.find(name);
  if (it != by_name_.end()) {
    CD_CHECK(it->second->shape == shape, "parameter %s declared twice with different shapes", name.c_str());
    return *it->second;
  }
  decls_.emplace_back(new ParamDecl());
  ParamDecl* d = decls_.back().get();
  d->name = name; d->shape = std::move(shape);
  by_name_[name] = d;
  return *d;
}
void ParamStore::conv_weight(const std::string& name, ConvW* c, int ref_ndim, float scale) {
  std::vector<int64_t> shape;
  if (ref_ndim == 0) ref_ndim = (c->KH == 1 && c->KW == 1) ? 2 : 4;
  if (ref_ndim == 2) shape = {c->N, c->Cin};
  else if (ref_ndim == 3) shape = {c->N, c->Cin, 1};
  else shape = {c->N, c->Cin, c->KH, c->KW};
  ParamDecl& d = declare(name, shape);
  PackTarget t; t.kind = PackTarget::MATRIX_BF16; t.conv = c; t.dst_row0 = 0; t.rows = c->N;
  t.src_base = 0; t.grp = c->N; t.grp_stride = 0; t.geglu = c->geglu; t.scale = scale;
  d.targets.push_back(t);
}
void ParamStore::conv_weight_t(const std::string& name, ConvW* c) {
  CD_CHECK(c->KH == 1 && c->KW == 1, "transposed weights are linear layers");
  ParamDecl& d = declare(name, {c->Cin, c->N});
  d.transposed = true;
  PackTarget t; t.kind = PackTarget::MATRIX_BF16; t.conv = c; t.dst_row0 = 0; t.rows = c->N;
  t.src_base = 0; t.grp = c->N; t.grp_stride = 0; t.geglu = false;
  d.targets.push_back(t);
}
void ParamStore::conv_bias(const std::string& name, ConvW* c) {
  CD_CHECK(c->b, "conv %s has no bias storage", name.c_str());
  ParamDecl& d = declare(name, {c->N});
  PackTarget t; t.kind = PackTarget::VECTOR_F32; t.fdst = c->b; t.dst_off = 0; t.rows = c->N;
  t.src_base = 0; t.grp = c->N; t.grp_stride = 0; t.geglu = c->geglu; t.geglu_N = c->N;
  d.targets.push_back(t);
}
void ParamStore::conv_rows(const std::string& name, std::vector<int64_t> shape, ConvW* c, int dst_row0,
                           int rows, int src_base, int grp, int grp_stride, float scale) {
  ParamDecl& d = declare(name, std::move(shape));
  PackTarget t; t.kind = PackTarget::MATRIX_BF16; t.conv = c; t.dst_row0 = dst_row0; t.rows = rows;
  t.src_base = src_base; t.grp = grp; t.grp_stride = grp_stride; t.scale = scale;
  d.targets.push_back(t);
}
void ParamStore::bias_rows(const std::string& name, int64_t n_total, float* dst, int dst_off, int rows,
                           int src_base, int grp, int grp_stride) {
  ParamDecl& d = declare(name, {n_total});
  PackTarget t; t.kind = PackTarget::VECTOR_F32; t.fdst = dst; t.dst_off = dst_off; t.rows = rows;
  t.src_base = src_base; t.grp = grp; t.grp_stride = grp_stride;
  d.targets.push_back(t);
}
void ParamStore::vec(const std::string& name, float* dst, int n) {
  bias_rows(name, n, dst, 0, n, 0, n, 0);
}
void ParamStore::mat_f32(const std::string& name, float* dst, int N, int K, int dst_row0) {
  ParamDecl& d = declare(name, {N, K});
  PackTarget t; t.kind = PackTarget::MATRIX_F32; t.fdst = dst; t.dst_off = dst_row0; t.rows = N; t.K = K;
  t.src_base = 0; t.grp = N; t.grp_stride = 0;
  d.targets.push_back(t);
}

// repack with a row map: dst row j <- src row src_base + (j/grp)*grp_stride + j%grp
template <typename OutT>
__global__ void k_pack_rows(const float* __restrict__ w, OutT* __restrict__ out, int rows, int Cin,
                            int KH, int KW, int Cpad, int dst_row0, int src_base, int grp,
                            int grp_stride, int geglu, int Ntot, float scale) {
  const int64_t total = (int64_t)rows * KH * KW * Cpad;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    int64_t t = i / Cpad;
    const int s = (int)(t % KW); t /= KW;
    const int r = (int)(t % KH); t /= KH;
    const int j = (int)t;  // destination row (relative)
    int jj = j;
    if (geglu) {  // packed row j: blocks of 64 = [32 value | 32 gate]
      const int blk = j / 64, within = j % 64;
      jj = (within < 32) ? blk * 32 + within : Ntot / 2 + blk * 32 + (within - 32);
    }
    const int src = src_base + (jj / grp) * grp_stride + (jj % grp);
    float v = 0.f;
    if (c < Cin) v = w[(((int64_t)src * Cin + c) * KH + r) * KW + s] * scale;
    const int64_t o = ((int64_t)(dst_row0 + j) * KH * KW + (int64_t)r * KW + s) * Cpad + c;
    if constexpr (sizeof(OutT) == 4) out[o] = v;
    else out[o] = f2bf(v);
  }
}

void ParamStore::load(hipStream_t st, const std::string& name, const float* host, int ndim,
                      const int64_t* shape) {
  ++version;
  auto it = by_name_.find(name);
  CD_CHECK(it != by_name_.end(), "unknown parameter name '%s'", name.c_str());
  ParamDecl& d = *it->second;
  int64_t n = 1, nd = 1;
  for (int i = 0; i < ndim; ++i) n *= shape[i];
  for (auto s : d.shape) nd *= s;
  // accept [N,C] for [N,C,1,1] / [N,C,1] (conv1d / 1x1 conv stored as linear and vice versa)
  bool ok = (n == nd) && ndim >= 1 && shape[0] == d.shape[0];
  if (ok && d.shape.size() >= 2 && ndim >= 2) ok = (shape[1] == d.shape[1]);
  CD_CHECK(ok, "parameter %s: shape mismatch (got %lld elements, ndim %d; expected %lld)", name.c_str(),
           (long long)n, ndim, (long long)nd);
  const size_t bytes = (size_t)n * sizeof(float);
  if (bytes > staging_bytes_) {
    if (staging_) HIP_CHECK(hipFree(staging_));
    staging_bytes_ = bytes < (1u << 20) ? (1u << 20) : bytes;
    HIP_CHECK(hipMalloc((void**)&staging_, staging_bytes_));
  }
  HIP_CHECK(hipStreamSynchronize(st));
  if (d.transposed) {  // [K][N] -> [N][K] on the host (small projection matrices)
    const int64_t K = d.shape[0], N = d.shape[1];
    std::vector<float> tr((size_t)n);
    for (int64_t k = 0; k < K; ++k)
      for (int64_t j = 0; j < N; ++j) tr[(size_t)j * K + k] = host[(size_t)k * N + j];
    HIP_CHECK(hipMemcpy(staging_, tr.data(), bytes, hipMemcpyHostToDevice));
  } else {
    HIP_CHECK(hipMemcpy(staging_, host, bytes, hipMemcpyHostToDevice));
  }
  for (const PackTarget& t : d.targets) {
    if (t.kind == PackTarget::MATRIX_BF16) {
      const ConvW* c = t.conv;
      const int64_t total = (int64_t)t.rows * c->KH * c->KW * c->Cpad;
      int grid = (int)((total + 255) / 256);
      if (grid > 8192) grid = 8192;
      if (c->f32)
        hipLaunchKernelGGL(k_pack_rows<float>, dim3(grid), dim3(256), 0, st, staging_, (float*)c->w, t.rows, c->Cin,
                           c->KH, c->KW, c->Cpad, t.dst_row0, t.src_base, t.grp, t.grp_stride, t.geglu ? 1 : 0, c->N,
                           t.scale);
      else
        hipLaunchKernelGGL(k_pack_rows<bf16_t>, dim3(grid), dim3(256), 0, st, staging_, c->w, t.rows, c->Cin, c->KH,
                           c->KW, c->Cpad, t.dst_row0, t.src_base, t.grp, t.grp_stride, t.geglu ? 1 : 0, c->N, t.scale);
      if (c->wfrag) launch_pack_wfrag(st, c->w, c->Ktot(), c->wfrag, c->Npad);  // whole matrix: rows may come in parts
      if (c->w3) launch_pack_w3(st, (const float*)c->w, c->w3, c->Npad, c->KH * c->KW, c->Cpad, overflow);
    } else {
      // small: map on the host
      const int K = (t.kind == PackTarget::MATRIX_F32) ? t.K : 1;
      std::vector<float> tmp((size_t)t.rows * K);
      for (int j = 0; j < t.rows; ++j) {
        int jj = j;
        if (t.geglu) {
          const int blk = j / 64, within = j % 64;
          jj = (within < 32) ? blk * 32 + within : t.geglu_N / 2 + blk * 32 + (within - 32);
        }
        const int src = t.src_base + (jj / t.grp) * t.grp_stride + (jj % t.grp);
        memcpy(&tmp[(size_t)j * K], host + (size_t)src * K, (size_t)K * sizeof(float));
      }
      HIP_CHECK(hipMemcpy(t.fdst + (size_t)t.dst_off * K, tmp.data(), tmp.size() * sizeof(float),
                          hipMemcpyHostToDevice));
    }
  }
  HIP_CHECK(hipStreamSynchronize(st));
  d.loaded = true;
}

int ParamStore::missing(std::string* first) const {
  int n = 0;
  for (auto& d : decls_)
    if (!d->loaded) {
      if (n == 0 && first) *first = d->name;
      ++n;
    }
  return n;
}

// ------------------------------------------------------------------ building blocks
Act alloc_act(Ctx& c, int B, int H, int W, int C, bool with_stats) {
  Act a; a.B = B; a.H = H; a.W = W; a.C = C; a.ld = C; a.f32 = c.f32;
  a.p = (bf16_t*)c.arena->alloc((size_t)B * H * W * C * (c.f32 ? sizeof(float) : sizeof(bf16_t)));
  if ((!c.f32 || c.x3) && with_stats && ((H * W) % 32) == 0)  // fp32 path: only its split mode emits statistics
    a.stats_buf = (float*)c.arena->alloc((size_t)(B * H * W / 32) * 2 * C * sizeof(float));
  return a;
}

// CD_PREC_F32X3: x = [hi | lo] fp16 pairs (a GroupNorm output), weights [wh | wh | wl]: one 16-bit implicit GEMM
// over the channel list [hi | lo | hi], fp32 bias / time-embedding row / residual / output (kernels.h kX3ActScale)
static Act conv_split_fwd(Ctx& c, const ConvW& w, const Act& x, const ConvOpts& o) {
  CD_CHECK(c.f32 && c.x3 && w.w3 && (!w.geglu || o.raw_geglu), "conv: split input outside the split-fp16 mode");
  CD_CHECK(x.C == w.Cpad && x.ld == 2 * x.C, "conv: split input of %d channels (ld %d) against Cpad %d", x.C, x.ld, w.Cpad);
  CD_CHECK(!o.ln_fold && (!o.resid || (o.resid->f32 && !o.resid->split)), "conv: split-mode operands");
  ConvGemmParams p;
  p.src0 = x.p; p.C0 = 2 * x.C; p.ld0 = x.ld;
  p.src1 = x.p; p.C1 = x.C; p.ld1 = x.ld;
  p.B = x.B; p.Hs = x.H; p.Ws = x.W; p.up = o.up ? 1 : 0;
  p.Hin = o.up ? x.H * 2 : x.H; p.Win = o.up ? x.W * 2 : x.W;
  p.KH = w.KH; p.KW = w.KW; p.stride = o.stride;
  if (o.asym) { p.pad_t = 0; p.pad_l = 0; p.Hout = (p.Hin + 1 - w.KH) / o.stride + 1; p.Wout = (p.Win + 1 - w.KW) / o.stride + 1; }
  else {
    p.pad_t = o.pad; p.pad_l = o.pad;
    p.Hout = (p.Hin + 2 * o.pad - w.KH) / o.stride + 1;
    p.Wout = (p.Win + 2 * o.pad - w.KW) / o.stride + 1;
  }
  p.M = x.B * p.Hout * p.Wout;
  p.wgt = w.w3; p.Ktot = 3 * w.Ktot(); p.N = w.N;
  p.alpha = o.alpha * (1.0f / (kX3ActScale * kX3WgtScale)); p.bias = w.b;
  p.rowvec = o.rowvec; p.rowvec_ld = o.rowvec_ld; p.rows_per_vec = o.rows_per_vec;
  p.act = o.act;
  Act y; y.B = x.B; y.H = p.Hout; y.W = p.Wout; y.C = w.N; y.f32 = true;
  if (o.out) { y.p = (bf16_t*)o.out; y.ld = o.out_ld; }
  else { y.ld = w.N; y.p = (bf16_t*)c.arena->alloc((size_t)p.M * w.N * 4); }
  if (o.resid) {
    CD_CHECK(o.resid->rows() == p.M && o.resid->C == w.N, "conv: residual shape mismatch");
    p.resid = o.resid->p; p.resid_ld = o.resid->ld; p.resid_f32 = 1;
  }
  p.out = y.p; p.out_ld = y.ld; p.out_f32 = 1;
  p.zeros = c.zeros; p.tile = o.tile;
  p.prof_flop_scale = 1.0f / 3.0f;
  if (((p.Hout * p.Wout) % 32) == 0) {  // the consumer's GroupNorm statistics from the epilogue, as on the 16-bit path
    if (o.out && o.out_stats) y.stats_buf = o.out_stats;
    else if (!o.out && o.want_stats)
      y.stats_buf = (float*)c.arena->alloc((size_t)(p.M / 32) * 2 * w.N * sizeof(float));
    p.stats = y.stats_buf;
    y.stats = y.stats_buf;
  }
  launch_conv_gemm(c.st, p);
  return y;
}

Act conv_fwd(Ctx& c, const ConvW& w, const Act& x, const Act* x2, const ConvOpts& o) {
  if (x.split) {
    CD_CHECK(!x2, "conv: a split activation cannot be concatenated");
    return conv_split_fwd(c, w, x, o);
  }
  ConvGemmParams p;
  CD_CHECK(w.f32 == c.f32 && x.f32 == c.f32 && (!x2 || x2->f32 == c.f32) && (!o.resid || o.resid->f32 == c.f32),
           "conv: operand precision does not match the running network");
  CD_CHECK((!x2 || !x2->split) && (!o.resid || !o.resid->split), "conv: split activation as a second source / residual");
  p.src0 = x.p; p.C0 = round_up(x.C, 32); p.ld0 = x.ld;
  if (x2) {
    CD_CHECK(x2->B == x.B && x2->H == x.H && x2->W == x.W, "conv: concat sources differ in shape");
    p.src1 = x2->p; p.C1 = x2->C; p.ld1 = x2->ld;
    CD_CHECK(x.C % 32 == 0 && x2->C % 32 == 0, "conv: concat channels must be multiples of 32");
  }
  CD_CHECK(p.C0 + p.C1 == w.Cpad, "conv: input channels %d+%d do not match weight Cpad %d", p.C0, p.C1, w.Cpad);
  CD_CHECK(x.ld >= p.C0, "conv: source row shorter than padded channel count");
  p.B = x.B; p.Hs = x.H; p.Ws = x.W; p.up = o.up ? 1 : 0;
  p.Hin = o.up ? x.H * 2 : x.H; p.Win = o.up ? x.W * 2 : x.W;
  p.KH = w.KH; p.KW = w.KW; p.stride = o.stride;
  if (o.asym) { p.pad_t = 0; p.pad_l = 0; p.Hout = (p.Hin + 1 - w.KH) / o.stride + 1; p.Wout = (p.Win + 1 - w.KW) / o.stride + 1; }
  else {
    p.pad_t = o.pad; p.pad_l = o.pad;
    p.Hout = (p.Hin + 2 * o.pad - w.KH) / o.stride + 1;
    p.Wout = (p.Win + 2 * o.pad - w.KW) / o.stride + 1;
  }
  p.M = x.B * p.Hout * p.Wout;
  p.wgt = w.w; p.wgt_frag = w.wfrag; p.Ktot = w.Ktot(); p.N = w.N;
  p.alpha = o.alpha; p.bias = w.b;
  p.rowvec = o.rowvec; p.rowvec_ld = o.rowvec_ld; p.rows_per_vec = o.rows_per_vec;
  const bool fuse_geglu = w.geglu && !o.raw_geglu;
  p.act = fuse_geglu ? ACT_GEGLU : o.act;
  const int Nout = fuse_geglu ? w.N / 2 : w.N;
  Act y; y.B = x.B; y.H = p.Hout; y.W = p.Wout; y.C = Nout; y.f32 = c.f32;
  if (o.out) { y.p = (bf16_t*)o.out; y.ld = o.out_ld; }
  else {
    y.ld = Nout;
    y.p = (bf16_t*)c.arena->alloc((size_t)p.M * Nout * ((o.out_f32 || c.f32) ? 4 : 2));
  }
  if (o.resid) {
    CD_CHECK(o.resid->rows() == p.M && o.resid->C == Nout, "conv: residual shape mismatch");
    p.resid = o.resid->p; p.resid_ld = o.resid->ld;
  }
  p.out = y.p; p.out_ld = y.ld; p.out_f32 = o.out_f32 ? 1 : 0;
  p.zeros = c.zeros; p.tile = o.tile;
  if (o.ln_fold) { p.ln_fold = 1; p.ln_eps = o.ln_eps; p.tile = kLinStreamTile; }
  if (c.f32) {
    p.out_f32 = 1;
    launch_conv_gemm_f32(c.st, p);
    return y;
  }
  if (!w.geglu && !o.out_f32 && ((p.Hout * p.Wout) % 32) == 0) {
    if (o.out && o.out_stats) y.stats_buf = o.out_stats;
    else if (!o.out && o.want_stats)
      y.stats_buf = (float*)c.arena->alloc((size_t)(p.M / 32) * 2 * Nout * sizeof(float));
    p.stats = y.stats_buf;
    y.stats = y.stats_buf;
  }
  launch_conv_gemm(c.st, p);
  return y;
}

bool conv_ln_fold_available(const Ctx& c, const ConvW& w, int64_t rows) {
  static const bool on = [] { const char* e = getenv("CYCLEDIFF_LN_FOLD"); return !(e && e[0] == '0'); }();  // A/B runs
  // the streaming kernel owns 256-row strips, one persistent workgroup per CU: below ~3/4 of the CUs' worth of strips the
  // tile kernel + a LayerNorm launch is the better pair (CYCLEDIFF_LN_FOLD_MIN_ROWS overrides the threshold for A/B runs)
  static const int64_t min_rows = [] { const char* e = getenv("CYCLEDIFF_LN_FOLD_MIN_ROWS"); return e ? atoll(e) : 32768ll; }();
  return on && !c.f32 && w.wfrag && w.KH == 1 && w.KW == 1 && w.Cpad == 320 && rows >= min_rows && rows % 32 == 0 &&
         w.N % 64 == 0 && w.N <= 2560;
}

Act groupnorm_fwd(Ctx& c, const GNW& w, const Act& x, const Act* x2, bool silu, const float* film,
                  int film_ld) {
  GroupNormParams p;
  p.x = x.p; p.C0 = x.C; p.ld0 = x.ld;
  if (x2) { p.x1 = x2->p; p.C1 = x2->C; p.ld1 = x2->ld; }
  const int C = p.C0 + p.C1;
  CD_CHECK(C == w.C, "groupnorm: channels %d != weight %d", C, w.C);
  p.B = x.B; p.HW = x.H * x.W; p.G = 32; p.eps = w.eps; p.gamma = w.g; p.beta = w.b;
  p.film = film; p.film_ld = film_ld; p.silu = silu ? 1 : 0;
  Act y = alloc_act(c, x.B, x.H, x.W, C);
  p.y = y.p;
  if (c.f32) {
    CD_CHECK(x.f32 && (!x2 || x2->f32) && !x.split && (!x2 || !x2->split), "groupnorm: operand precision");
    if (c.x3) { p.split_out = 1; p.overflow = c.overflow; y.split = true; y.ld = 2 * C; }  // same bytes as the fp32 tensor
    if (x.stats && x.ld == x.C && (!x2 || (x2->stats && x2->ld == x2->C))) {  // written by the split convs' epilogues
      p.pre0 = x.stats;
      p.pre1 = x2 ? x2->stats : nullptr;
    }
    const size_t mk = c.arena->mark();
    void* ws = c.arena->alloc(groupnorm_f32_workspace(p.B, p.HW, C));
    launch_groupnorm_f32(c.st, p, ws);
    c.arena->release(mk);  // stream order keeps the scratch alive until the three kernels have run
    return y;
  }
  p.S = groupnorm_slabs(p.B, p.HW, C);
  const size_t part_floats = round_up((size_t)p.B * p.S * p.G * 2, (size_t)64);
  CD_CHECK(part_floats + (size_t)p.B * 2 * C <= c.gn_partial_floats, "groupnorm: workspace too small");
  p.partial = c.gn_partial;
  p.coef = c.gn_partial + part_floats;
  if (x.stats && x.ld == x.C && (!x2 || (x2->stats && x2->ld == x2->C))) {
    p.pre0 = x.stats;
    p.pre1 = x2 ? x2->stats : nullptr;
  }
  launch_groupnorm(c.st, p);
  return y;
}

Act avgpool2_fwd(Ctx& c, const Act& x) {
  if (x.split) {
    Act y = alloc_act(c, x.B, x.H / 2, x.W / 2, x.C);
    y.split = true; y.ld = 2 * x.C;
    launch_avgpool2_split(c.st, x.p, y.p, x.B, x.H, x.W, x.C, c.overflow);
    return y;
  }
  CD_CHECK(x.ld == x.C, "avgpool: dense input expected");
  Act y = alloc_act(c, x.B, x.H / 2, x.W / 2, x.C);
  if (c.f32) launch_avgpool2_f32(c.st, x.pf(), y.pf(), x.B, x.H, x.W, x.C);
  else launch_avgpool2(c.st, x.p, y.p, x.B, x.H, x.W, x.C);
  return y;
}
Act upsample2_fwd(Ctx& c, const Act& x) {
  CD_CHECK(x.ld == x.C && !x.split, "upsample: dense input expected");
  Act y = alloc_act(c, x.B, x.H * 2, x.W * 2, x.C);
  if (c.f32) launch_upsample2_f32(c.st, x.pf(), y.pf(), x.B, x.H, x.W, x.C);
  else launch_upsample2(c.st, x.p, y.p, x.B, x.H, x.W, x.C);
  return y;
}
Act attention_f32_fwd(Ctx& c, const Act& qk, const Act& v, int H, int D, float scale, const float* obias) {
  CD_CHECK(c.f32 && qk.f32 && v.f32 && !qk.split && !v.split && qk.C == 2 * H * D && v.C == H * D,
           "attention_f32: operands");
  Act o = alloc_act(c, qk.B, qk.H, qk.W, H * D);
  const int T = qk.H * qk.W;
  if (D <= 160 && (D % 4) == 0 && (qk.ld % 4) == 0 && (v.ld % 4) == 0) {
    // round 5: heads up to 160 wide (the improved-DDPM AttentionBlocks: 64) on the fp32 flash kernel of st_f32.hip - fp32
    // matrix instructions instead of one wave per query (3 % of a config-5 step); wider single heads (Ho-DDPM, the KL-f8
    // first stage: d = C) stay on k_attention_f32
    launch_flash_f32(c.st, qk.pf(), qk.ld, (int64_t)T * qk.ld, qk.pf() + H * D, qk.ld, (int64_t)T * qk.ld, v.pf(), v.ld,
                     (int64_t)T * v.ld, o.pf(), o.ld, (int64_t)T * o.ld, qk.B, H, T, T, D, scale * 1.44269504088896340736f, 0,
                     c.overflow, obias);
    return o;
  }
  launch_attention_f32(c.st, qk.pf(), qk.ld, qk.pf() + H * D, qk.ld, v.pf(), v.ld, o.pf(), o.ld, qk.B, H, T, D, scale, obias);
  return o;
}

Act layernorm_fwd(Ctx& c, const LNW& w, const Act& x) {
  CD_CHECK(x.C == w.C, "layernorm: channels");
  if (c.f32) {  // fp32 rows; in the split mode the output is the fp16 pair the next three-term GEMM consumes
    CD_CHECK(x.f32 && !x.split, "layernorm_f32: operand precision");
    Act y = alloc_act(c, x.B, x.H, x.W, x.C);
    if (c.x3) { y.split = true; y.ld = 2 * x.C; }  // same bytes as the fp32 tensor
    launch_layernorm_f32(c.st, x.pf(), x.ld, y.pf(), x.rows(), x.C, w.g, w.b, 1e-5f, c.x3 ? 1 : 0, c.overflow);
    return y;
  }
  Act y = alloc_act(c, x.B, x.H, x.W, x.C);
  launch_layernorm(c.st, x.p, x.ld, y.p, y.ld, (int)x.rows(), x.C, w.g, w.b, 1e-5f);
  return y;
}

Act attention_fwd(Ctx& c, const bf16_t* q, int ldq, const bf16_t* k, int ldk, const bf16_t* v, int ldv, int B,
                  int H, int Tq, int Tk, int D, float scale, int Himg, int Wimg, bool q_log2) {
  CD_CHECK(!c.f32, "attention: 16-bit kernel called on the fp32 path");
  Act o = alloc_act(c, B, Himg, Wimg, H * D);
  AttnParams p;
  p.q = q; p.k = k; p.v = v; p.o = o.p;
  p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.D = D;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = o.ld;
  p.q_bs = (int64_t)Tq * ldq; p.k_bs = (int64_t)Tk * ldk; p.v_bs = (int64_t)Tk * ldv; p.o_bs = (int64_t)Tq * o.ld;
  p.scale = scale; p.q_log2 = q_log2 ? 1 : 0;
  launch_attention(c.st, p);
  return o;
}

Act attention_vt_fwd(Ctx& c, const bf16_t* q, int ldq, const bf16_t* k, int ldk, const bf16_t* vt, int B,
                     int H, int Tq, int Tk, int Tpad, int D, float scale, int Himg, int Wimg, bool q_log2) {
  CD_CHECK(!c.f32, "attention: 16-bit kernel called on the fp32 path");
  Act o = alloc_act(c, B, Himg, Wimg, H * D);
  AttnParams p;
  p.q = q; p.k = k; p.vt = vt; p.o = o.p;
  p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.D = D;
  p.ldq = ldq; p.ldk = ldk; p.ldo = o.ld;
  p.q_bs = (int64_t)Tq * ldq; p.k_bs = (int64_t)Tk * ldk; p.o_bs = (int64_t)Tq * o.ld;
  p.vt_dpad = D; p.vt_tpad = Tpad;
  p.scale = scale; p.q_log2 = q_log2 ? 1 : 0;
  launch_attention(c.st, p);
  return o;
}

Act attention_flash_f32_fwd(Ctx& c, const float* q, int ldq, const float* k, int ldk, int64_t k_bs, const float* v, int ldv,
                            int64_t v_bs, int B, int H, int Tq, int Tk, int D, float scale, int Himg, int Wimg, bool q_log2) {
  CD_CHECK(c.f32, "attention_flash_f32: fp32 path only");
  Act o = alloc_act(c, B, Himg, Wimg, H * D);
  // split mode: the output feeds to_out as fp16 pairs (same bytes as the fp32 tensor; range guard as for the norms)
  launch_flash_f32(c.st, q, ldq, (int64_t)Tq * ldq, k, ldk, k_bs, v, ldv, v_bs, o.pf(), o.ld, (int64_t)Tq * o.ld, B, H, Tq,
                   Tk, D, q_log2 ? 1.0f : scale * 1.44269504088896340736f, c.x3 ? 1 : 0, c.overflow);
  if (c.x3) { o.split = true; o.ld = 2 * o.C; }
  return o;
}

Act geglu_f32_fwd(Ctx& c, const Act& h) {
  CD_CHECK(c.f32 && h.f32 && !h.split && h.ld == h.C && (h.C % 64) == 0, "geglu_f32: operand");
  Act y = alloc_act(c, h.B, h.H, h.W, h.C / 2);
  launch_geglu_f32(c.st, h.pf(), y.pf(), h.rows(), h.C / 2, c.x3 ? 1 : 0, c.overflow);
  if (c.x3) { y.split = true; y.ld = 2 * y.C; }
  return y;
}

Act split_rows_f32_fwd(Ctx& c, const Act& x, const Act* x2) {
  CD_CHECK(c.f32 && c.x3 && x.f32 && !x.split && (!x2 || (x2->f32 && !x2->split && x2->rows() == x.rows())),
           "split_rows_f32: operand");
  const int C = x.C + (x2 ? x2->C : 0);
  Act y = alloc_act(c, x.B, x.H, x.W, C);
  launch_split_rows_f32(c.st, x.pf(), x.ld, x.C, x2 ? x2->pf() : nullptr, x2 ? x2->ld : 4, x2 ? x2->C : 0, y.p, x.rows(),
                        c.overflow);
  y.split = true; y.ld = 2 * C;
  return y;
}

}  // namespace cd
