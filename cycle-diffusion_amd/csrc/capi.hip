// extern "C" boundary (include/cyclediff.h): engine lifetime, weight loading, the sampler loops
// (DPM-Encoder / coupled decode) and single-kernel entry points used by the parity tests.
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>

#include "engine.h"

using namespace cd;

// Host pacing (one process per GPU, one launching thread per engine): the sampler loops enqueue ~350 kernels per step
// and would otherwise run seconds ahead of the GPU, where the HIP runtime busy-waits for queue slots (round 2 measured
// 2.0 host cores per rank). After every sampler step an event is recorded; the host then SLEEPS (blocking-sync event)
// until the step kAhead steps back has finished - the GPU always has two whole steps (tens of ms) queued.
struct StepPacer {
  static constexpr int kAhead = 2, kRing = kAhead + 1;
  hipEvent_t ev[kRing] = {nullptr, nullptr, nullptr};
  long n = 0;
  bool enabled = true;
  void init() {
    const char* e = getenv("CYCLEDIFF_HOST_PACING");
    enabled = !(e && e[0] == '0');
    for (auto& x : ev) HIP_CHECK(hipEventCreateWithFlags(&x, hipEventBlockingSync | hipEventDisableTiming));
  }
  // hipEventSynchronize was measured to keep a core busy even on hipEventBlockingSync events (round 3, call 4: the
  // launching thread at 1.00 core with pacing on): the wait is a query + nanosleep loop instead - with two sampler
  // steps (tens of ms) queued, a 100 us wake-up granularity costs nothing
  static void sleep_until_done(hipEvent_t e) {
    for (;;) {
      const hipError_t q = hipEventQuery(e);
      if (q == hipSuccess) return;
      if (q != hipErrorNotReady) HIP_CHECK(q);
      usleep(100);
    }
  }
  void tick(hipStream_t st) {
    if (!enabled) return;
    HIP_CHECK(hipEventRecord(ev[n % kRing], st));
    ++n;
    if (n > kAhead) sleep_until_done(ev[(n - 1 - kAhead) % kRing]);
  }
  void wait_all(hipStream_t st) {  // sleeping equivalent of hipStreamSynchronize
    if (!ev[0]) { HIP_CHECK(hipStreamSynchronize(st)); return; }
    HIP_CHECK(hipEventRecord(ev[n % kRing], st));
    sleep_until_done(ev[n % kRing]);
    ++n;
  }
  void destroy() { for (auto& x : ev) if (x) { (void)hipEventDestroy(x); x = nullptr; } }
};

// Scheduler tables travel through a small ring of pinned host slots: the copy is asynchronous (no stream drain at the
// start of every sampler call) and the caller may free its table as soon as the entry point returns.
struct CoefStaging {
  static constexpr int kSlots = 4;
  static constexpr size_t kSlotBytes = 64 << 10;
  char* host = nullptr;
  hipEvent_t done[kSlots] = {nullptr, nullptr, nullptr, nullptr};
  int next = 0;
  void init() {
    HIP_CHECK(hipHostMalloc((void**)&host, kSlots * kSlotBytes, hipHostMallocDefault));
    for (auto& x : done) HIP_CHECK(hipEventCreateWithFlags(&x, hipEventBlockingSync | hipEventDisableTiming));
  }
  void destroy() {
    for (auto& x : done) if (x) { (void)hipEventDestroy(x); x = nullptr; }
    if (host) { (void)hipHostFree(host); host = nullptr; }
  }
};

struct cd_engine {
  hipStream_t st = nullptr;
  StepPacer pacer;
  CoefStaging coef_staging;
  Arena arena;
  bf16_t* zeros = nullptr;
  float* gn_partial = nullptr;
  size_t gn_partial_floats = 0;
  std::vector<std::unique_ptr<Net>> nets;
  std::vector<ParamStore*> op_stores;  // storage behind cd_op_pack_conv
  ParamStore op_params;
  std::unique_ptr<KernelProfiler> prof;
  SplitKWorkspace splitk;  // split-K partial tiles + arrival counters of THIS engine's stream (conv_gemm.hip)
  // CD_PREC_F32X3 range guard: one word of mapped host memory the split kernels set when a scaled activation / weight
  // leaves the fp16 range; read (no stream drain needed) at every API entry and after cd_engine_synchronize
  int* overflow_host = nullptr;
  int* overflow_dev = nullptr;
  void check_overflow() {
    if (overflow_host && *(volatile int*)overflow_host) {
      *(volatile int*)overflow_host = 0;
      CD_CHECK(false, "CD_PREC_F32X3: a GroupNorm output or weight left the fp16 range of the split representation "
                      "(|x| >= 4094 or |w| >= 255); results since the last synchronisation are invalid - run this "
                      "network with CD_PREC_F32");
    }
  }
  Ctx ctx() {
    Ctx c; c.st = st; c.arena = &arena; c.zeros = zeros; c.gn_partial = gn_partial;
    c.gn_partial_floats = gn_partial_floats; c.overflow = overflow_dev;
    return c;
  }
};

// Restores the engine's bump arena on every exit path of an entry point, including a CD_CHECK / HIP_CHECK that throws
// into CD_API_END (an un-restored arena would stay elevated - after an out-of-workspace error nearly full - and make
// later, smaller calls fail until the engine is destroyed).
struct ArenaScope {
  Arena& a;
  size_t m;
  explicit ArenaScope(Arena& arena) : a(arena), m(arena.mark()) {}
  ~ArenaScope() { a.release(m); }
  ArenaScope(const ArenaScope&) = delete;
  ArenaScope& operator=(const ArenaScope&) = delete;
};

static thread_local std::string g_err;

// Engines are independent (own stream, arena, workspaces): several may run concurrently from different host
// threads - that is how bench.py keeps two batches in flight per GPU. The calling thread binds the kernel
// layer's per-stream state to its engine at every API entry.
static void enter_engine(cd_engine* h) {
  if (!h) return;
  g_conv_splitk = h->splitk;
  g_conv_prof = (h->prof && h->prof->enabled) ? h->prof.get() : nullptr;
  h->check_overflow();
}

#define CD_API_BEGIN try {
#define CD_API_END                                  \
  }                                                 \
  catch (const std::exception& e) {                 \
    g_err = e.what();                               \
    return 1;                                       \
  }                                                 \
  catch (...) {                                     \
    g_err = "unknown C++ exception";                \
    return 1;                                       \
  }                                                 \
  return 0;

static UNet* get_unet(cd_handle h, int net) {
  CD_CHECK(h && net >= 0 && net < (int)h->nets.size(), "bad net id %d", net);
  Net* n = h->nets[net].get();
  CD_CHECK(n->kind() == CD_NET_UNET_OPENAI || n->kind() == CD_NET_UNET_HO, "net %d is not a U-Net", net);
  return static_cast<UNet*>(n);
}
static VAE* get_vae(cd_handle h, int net) {
  CD_CHECK(h && net >= 0 && net < (int)h->nets.size(), "bad net id %d", net);
  Net* n = h->nets[net].get();
  CD_CHECK(n->kind() == CD_NET_VAE_KL, "net %d is not a VAE", net);
  return static_cast<VAE*>(n);
}

extern "C" {

const char* cd_last_error(void) { return g_err.c_str(); }
int cd_version(void) { return 100; }
int cd_act_format(void) { return CD_ACT_FP16 ? 1 : 0; }

int cd_engine_create(void* hip_stream, size_t workspace_bytes, cd_handle* out) {
  CD_API_BEGIN
  CD_CHECK(out, "null out pointer");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  CD_CHECK(e == hipSuccess && ndev > 0, "no HIP device available (%s): the engine has no CPU fallback",
           hipGetErrorString(e));
  std::unique_ptr<cd_engine> h(new cd_engine());
  h->st = (hipStream_t)hip_stream;
  if (workspace_bytes < (256u << 20)) workspace_bytes = 256u << 20;
  h->arena.init(workspace_bytes);
  HIP_CHECK(hipMalloc((void**)&h->zeros, 4096));
  HIP_CHECK(hipMemset(h->zeros, 0, 4096));
  {  // implicit-GEMM autotuner (CYCLEDIFF_AUTOTUNE=0 turns it off)
    const char* e = getenv("CYCLEDIFF_AUTOTUNE");
    if (!(e && e[0] == '0') && !g_conv_tuner.scratch) {
      g_conv_tuner.scratch_bytes = (size_t)768 << 20;
      HIP_CHECK(hipMalloc(&g_conv_tuner.scratch, g_conv_tuner.scratch_bytes));
      g_conv_tuner.enabled = true;
    }
  }
  h->splitk.scratch_bytes = (size_t)256 << 20;
  h->splitk.nflags = 1 << 16;
  HIP_CHECK(hipMalloc((void**)&h->splitk.scratch, h->splitk.scratch_bytes));
  HIP_CHECK(hipMalloc((void**)&h->splitk.flags, h->splitk.nflags * sizeof(int)));
  HIP_CHECK(hipMemset(h->splitk.flags, 0, h->splitk.nflags * sizeof(int)));
  HIP_CHECK(hipDeviceSynchronize());
  h->gn_partial_floats = (size_t)1 << 23;  // 32 MB: statistics partials + coefficients of a B = 192 forward (coupled loop, fold 16) need 1.8 M floats
  HIP_CHECK(hipMalloc((void**)&h->gn_partial, h->gn_partial_floats * sizeof(float)));
  h->pacer.init();
  h->coef_staging.init();
  HIP_CHECK(hipHostMalloc((void**)&h->overflow_host, 64, hipHostMallocMapped));
  *h->overflow_host = 0;
  HIP_CHECK(hipHostGetDevicePointer((void**)&h->overflow_dev, h->overflow_host, 0));
  *out = h.release();
  CD_API_END
}

int cd_engine_destroy(cd_handle h) {
  CD_API_BEGIN
  if (h && h->overflow_host) *h->overflow_host = 0;  // nothing left to report to
  enter_engine(h);
  if (h) {
    (void)hipStreamSynchronize(h->st);
    if (g_conv_prof == h->prof.get()) g_conv_prof = nullptr;
    if (g_conv_splitk.scratch == h->splitk.scratch) g_conv_splitk = SplitKWorkspace();
    if (h->splitk.scratch) (void)hipFree(h->splitk.scratch);
    if (h->splitk.flags) (void)hipFree(h->splitk.flags);
    if (h->zeros) (void)hipFree(h->zeros);
    if (h->gn_partial) (void)hipFree(h->gn_partial);
    h->pacer.destroy();
    h->coef_staging.destroy();
    if (h->overflow_host) (void)hipHostFree(h->overflow_host);
    delete h;
  }
  CD_API_END
}

int cd_engine_synchronize(cd_handle h) {
  CD_API_BEGIN
  CD_CHECK(h, "null handle");
  h->pacer.wait_all(h->st);
  h->check_overflow();
  CD_API_END
}

int cd_prof_enable(cd_handle h, int on) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h, "null handle");
  if (!h->prof) h->prof.reset(new KernelProfiler());
  h->prof->enabled = on != 0;
  {
    const char* e = getenv("CYCLEDIFF_GEMM_LOG");
    h->prof->verbose = e && e[0] == '1';
  }
  g_conv_prof = on ? h->prof.get() : nullptr;
  CD_API_END
}

int cd_prof_collect(cd_handle h, int* launches, double* total_ms, double* total_flops) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && h->prof && launches && total_ms && total_flops, "profiler not enabled");
  HIP_CHECK(hipStreamSynchronize(h->st));
  h->prof->collect(launches, total_ms, total_flops);
  CD_API_END
}

int cd_engine_workspace_high_water(cd_handle h, size_t* bytes) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && bytes, "null argument");
  *bytes = h->arena.high_water();
  CD_API_END
}

int cd_net_create(cd_handle h, const cd_net_desc* d, int* net_id) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && d && net_id, "null argument");
  std::unique_ptr<Net> n;
  switch (d->kind) {
    case CD_NET_UNET_OPENAI: n = make_unet_openai(*d); break;
    case CD_NET_UNET_HO: n = make_unet_ho(*d); break;
    case CD_NET_VAE_KL: n = make_vae_kl(*d); break;
    case CD_NET_CLIP_TEXT:
    case CD_NET_BERT_XTR:
    case CD_NET_OCLIP_TEXT:
    case CD_NET_OCLIP_VISION: n = make_clip_text(*d); break;
    default: CD_CHECK(false, "unknown net kind %d", d->kind);
  }
  n->params.overflow = h->overflow_dev;
  h->nets.push_back(std::move(n));
  *net_id = (int)h->nets.size() - 1;
  CD_API_END
}

int cd_net_param_count(cd_handle h, int net, int* n) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && n && net >= 0 && net < (int)h->nets.size(), "bad argument");
  *n = (int)h->nets[net]->params.decls().size();
  CD_API_END
}

int cd_net_param_info(cd_handle h, int net, int index, char* name, int name_cap, int* ndim, int64_t shape[4]) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && net >= 0 && net < (int)h->nets.size(), "bad net id");
  auto& ds = h->nets[net]->params.decls();
  CD_CHECK(index >= 0 && index < (int)ds.size(), "bad parameter index");
  const ParamDecl& d = *ds[index];
  if (name && name_cap > 0) { strncpy(name, d.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (ndim) *ndim = (int)d.shape.size();
  if (shape) for (size_t i = 0; i < 4; ++i) shape[i] = i < d.shape.size() ? d.shape[i] : 1;
  CD_API_END
}

int cd_net_load_param(cd_handle h, int net, const char* name, const float* data_host, int ndim,
                      const int64_t* shape) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && name && data_host && shape && net >= 0 && net < (int)h->nets.size(), "bad argument");
  h->nets[net]->params.load(h->st, name, data_host, ndim, shape);
  CD_API_END
}

int cd_net_missing_params(cd_handle h, int net, int* n_missing, char* first_name, int name_cap) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && n_missing && net >= 0 && net < (int)h->nets.size(), "bad argument");
  std::string first;
  *n_missing = h->nets[net]->params.missing(&first);
  if (first_name && name_cap > 0) { strncpy(first_name, first.c_str(), name_cap - 1); first_name[name_cap - 1] = 0; }
  CD_API_END
}

}  // extern "C"

// ------------------------------------------------------------------ helpers for the forward paths
namespace {

// ctx fp32 [B][L][Dc] -> bf16, optionally [uncond | cond] stacking for classifier-free guidance
bf16_t* make_ctx(cd_engine* h, const float* a, const float* b, int B, int L, int Dc, bool f32 = false) {
  const int nb = b ? 2 * B : B;
  if (f32) {  // fp32 networks (CD_PREC_F32 / F32X3 with transformer blocks) take the context as it is
    float* out = (float*)h->arena.alloc((size_t)nb * L * Dc * 4);
    const size_t half = (size_t)B * L * Dc;
    HIP_CHECK(hipMemcpyAsync(out, a, half * 4, hipMemcpyDeviceToDevice, h->st));
    if (b) HIP_CHECK(hipMemcpyAsync(out + half, b, half * 4, hipMemcpyDeviceToDevice, h->st));
    return (bf16_t*)out;
  }
  bf16_t* out = (bf16_t*)h->arena.alloc((size_t)nb * L * Dc * 2);
  launch_nchw_to_nhwc(h->st, a, out, B * L, Dc, 1, Dc, 1.f, 0.f, 0);
  if (b) launch_nchw_to_nhwc(h->st, b, out + (size_t)B * L * Dc, B * L, Dc, 1, Dc, 1.f, 0.f, 0);
  return out;
}

struct Guidance {
  bool cfg = false;       // run the 2B batch [uncond | cond]
  const float* ctx_single = nullptr;
};
// ddim.py:550-559: scale==1 -> conditional only; scale==0 -> unconditional only; else CFG
Guidance resolve_guidance(const float* ctx_c, const float* ctx_uc, float g) {
  Guidance r;
  if (!ctx_c && !ctx_uc) return r;
  if (!ctx_uc || g == 1.0f) { r.ctx_single = ctx_c; return r; }
  if (g == 0.0f) { r.ctx_single = ctx_uc; return r; }
  r.cfg = true;
  return r;
}

StepCoef* upload_coef(cd_engine* h, const cd_step_coef* host, int n) {
  static_assert(sizeof(cd_step_coef) == sizeof(StepCoef), "coef ABI");
  const size_t bytes = (size_t)n * sizeof(StepCoef);
  StepCoef* d = (StepCoef*)h->arena.alloc(bytes);
  CoefStaging& cs = h->coef_staging;
  if (cs.host && bytes <= CoefStaging::kSlotBytes) {
    const int slot = cs.next;
    cs.next = (cs.next + 1) % CoefStaging::kSlots;
    StepPacer::sleep_until_done(cs.done[slot]);  // the copy that last used this slot (a never-recorded event is done)
    char* stage = cs.host + (size_t)slot * CoefStaging::kSlotBytes;
    memcpy(stage, host, bytes);
    HIP_CHECK(hipMemcpyAsync(d, stage, bytes, hipMemcpyHostToDevice, h->st));
    HIP_CHECK(hipEventRecord(cs.done[slot], h->st));
  } else {
    HIP_CHECK(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, h->st));
    HIP_CHECK(hipStreamSynchronize(h->st));  // host table may be freed by the caller right after return
  }
  return d;
}

struct SamplerState {
  UNet* u = nullptr;
  int B = 0, Bn = 0, C = 0, HW = 0, cpad = 0, out_ld = 0;
  bool cfg = false;
  bool f32 = false;  // CD_PREC_F32 network: its NHWC input is fp32 and is rebuilt from xt before every forward
  bf16_t* xin16() const { return f32 ? nullptr : xin; }
  float g = 1.f;
  float* xt = nullptr;
  bf16_t* xin = nullptr;
  float* eh = nullptr;
  StepCoef* tab = nullptr;
  EpsHat ehv;
};

SamplerState setup_sampler(cd_engine* h, int net, const float* ctx_c, const float* ctx_uc, int ctx_len,
                           float guidance, int B) {
  SamplerState s;
  s.u = get_unet(h, net);
  s.B = B; s.C = s.u->desc.in_channels; s.HW = s.u->image_size * s.u->image_size;
  s.cpad = s.u->in_cpad; s.out_ld = s.u->out_channels;
  Guidance gd = resolve_guidance(ctx_c, ctx_uc, guidance);
  s.cfg = gd.cfg; s.g = guidance; s.Bn = s.cfg ? 2 * B : B;
  Ctx c = h->ctx();
  if (ctx_c || ctx_uc) {
    const int Dc = s.u->desc.context_dim;
    CD_CHECK(Dc > 0 && ctx_len > 0, "network has no cross-attention but a context was given");
    bf16_t* cx = s.cfg ? make_ctx(h, ctx_uc, ctx_c, B, ctx_len, Dc, s.u->f32) : make_ctx(h, gd.ctx_single, nullptr, B, ctx_len, Dc, s.u->f32);
    s.u->set_context(c, cx, s.Bn, ctx_len);
  } else {
    CD_CHECK(s.u->desc.context_dim <= 0 || !s.u->desc.use_spatial_transformer,
             "network expects a cross-attention context");
  }
  s.f32 = s.u->f32;
  const size_t esz = s.f32 ? 4 : 2;
  s.xt = (float*)h->arena.alloc((size_t)B * s.C * s.HW * 4);
  s.xin = (bf16_t*)h->arena.alloc((size_t)s.Bn * s.HW * s.cpad * esz);
  HIP_CHECK(hipMemsetAsync(s.xin, 0, (size_t)s.Bn * s.HW * s.cpad * esz, h->st));
  s.eh = (float*)h->arena.alloc((size_t)s.Bn * s.HW * s.out_ld * 4);
  s.ehv.p = s.eh; s.ehv.sb = (int64_t)s.HW * s.out_ld; s.ehv.sc = 1; s.ehv.sp = s.out_ld;
  s.ehv.cfg = s.cfg ? 1 : 0; s.ehv.g = guidance;
  return s;
}

void run_unet(cd_engine* h, SamplerState& s, int step) {
  Ctx c = h->ctx();
  if (s.f32) {
    launch_nchw_to_nhwc_f32(h->st, s.xt, (float*)s.xin, s.B, s.C, s.HW, s.cpad, 1.f, 0.f, s.u->x3 ? 1 : 0, h->overflow_dev);
    if (s.cfg)  // the classifier-free-guidance batch [uncond | cond] is built from one x_t (ddim.py:553-559)
      launch_nchw_to_nhwc_f32(h->st, s.xt, (float*)s.xin + (size_t)s.B * s.HW * s.cpad, s.B, s.C, s.HW, s.cpad, 1.f, 0.f,
                              s.u->x3 ? 1 : 0, h->overflow_dev);
  }
  UNetIO io;
  io.xin = s.xin; io.B = s.Bn; io.tab = s.tab; io.step = step; io.t_shared = true;
  io.cfg_dup = s.cfg;
  io.out = s.eh; io.out_ld = s.out_ld;
  s.u->forward(c, io);
}

}  // namespace

extern "C" {

int cd_unet_forward(cd_handle h, int net, const float* x, const float* t, const float* ctx, int B,
                    int ctx_len, float* eps_out) {
  CD_API_BEGIN
  enter_engine(h);
  UNet* u = get_unet(h, net);
  CD_CHECK(x && t && eps_out && B > 0, "bad argument");
  ArenaScope arena_scope(h->arena);
  Ctx c = h->ctx();
  const int C = u->desc.in_channels, HW = u->image_size * u->image_size, Co = u->out_channels;
  if (ctx) {
    const int Dc = u->desc.context_dim;
    CD_CHECK(Dc > 0, "network has no cross-attention");
    bf16_t* cx = make_ctx(h, ctx, nullptr, B, ctx_len, Dc, u->f32);
    u->set_context(c, cx, B, ctx_len);
  }
  bf16_t* xin = (bf16_t*)h->arena.alloc((size_t)B * HW * u->in_cpad * (u->f32 ? 4 : 2));
  if (u->f32) launch_nchw_to_nhwc_f32(h->st, x, (float*)xin, B, C, HW, u->in_cpad, 1.f, 0.f, u->x3 ? 1 : 0, h->overflow_dev);
  else launch_nchw_to_nhwc(h->st, x, xin, B, C, HW, u->in_cpad, 1.f, 0.f, 0);
  float* eh = (float*)h->arena.alloc((size_t)B * HW * Co * 4);
  UNetIO io; io.xin = xin; io.B = B; io.t_explicit = t; io.t_shared = false; io.out = eh; io.out_ld = Co;
  u->forward(c, io);
  launch_nhwc_to_nchw(h->st, eh, 1, Co, eps_out, B, Co, HW, 1.f, 0.f);
  CD_API_END
}

int cd_text_encode(cd_handle h, int net, const int32_t* tokens, int B, int L, float* out) {
  CD_API_BEGIN
  CD_CHECK(h, "null handle");
  enter_engine(h);
  CD_CHECK(net >= 0 && net < (int)h->nets.size() &&
               (h->nets[net]->kind() == CD_NET_CLIP_TEXT || h->nets[net]->kind() == CD_NET_BERT_XTR),
           "net %d is not a text encoder", net);
  CD_CHECK(tokens && out && B > 0 && L > 0, "bad argument");
  ArenaScope arena_scope(h->arena);
  Ctx c = h->ctx();
  static_cast<TextEncoder*>(h->nets[net].get())->encode(c, (const int*)tokens, B, L, out);
  CD_API_END
}

static TextEncoder* get_tower(cd_handle h, int net, int kind) {
  CD_CHECK(h && net >= 0 && net < (int)h->nets.size() && h->nets[net]->kind() == kind, "net %d has the wrong kind", net);
  return static_cast<TextEncoder*>(h->nets[net].get());
}

int cd_clip_text_features(cd_handle h, int net, const int32_t* tokens, int B, int L, float* out) {
  CD_API_BEGIN
  CD_CHECK(h, "null handle");
  enter_engine(h);
  CD_CHECK(tokens && out && B > 0 && L > 0, "bad argument");
  ArenaScope arena_scope(h->arena);
  Ctx c = h->ctx();
  get_tower(h, net, CD_NET_OCLIP_TEXT)->text_features(c, (const int*)tokens, B, L, out);
  CD_API_END
}

int cd_clip_image_features(cd_handle h, int net, const float* img, int B, float* out) {
  CD_API_BEGIN
  CD_CHECK(h, "null handle");
  enter_engine(h);
  CD_CHECK(img && out && B > 0, "bad argument");
  ArenaScope arena_scope(h->arena);
  Ctx c = h->ctx();
  get_tower(h, net, CD_NET_OCLIP_VISION)->image_features(c, img, B, out);
  CD_API_END
}

int cd_vae_encode(cd_handle h, int net, const float* img, const float* noise, uint64_t seed, int B, int R,
                  int sample, float scale, float* z0) {
  CD_API_BEGIN
  enter_engine(h);
  VAE* v = get_vae(h, net);
  CD_CHECK(img && z0 && B > 0 && R % v->factor == 0, "bad argument");
  ArenaScope arena_scope(h->arena);
  Ctx c = h->ctx();
  const int cin = v->desc.in_channels, cp = round_up(cin, 32), hl = R / v->factor;
  // 16-bit rows, or - first stage in fp32 / in the split mode - fp32 rows / fp16 pairs (the same bytes)
  bf16_t* xin = (bf16_t*)h->arena.alloc((size_t)B * R * R * cp * (v->f32 ? 4 : 2));
  if (v->f32) launch_nchw_to_nhwc_f32(h->st, img, (float*)xin, B, cin, R * R, cp, 1.f, 0.f, v->x3 ? 1 : 0, h->overflow_dev);
  else launch_nchw_to_nhwc(h->st, img, xin, B, cin, R * R, cp, 1.f, 0.f, 0);
  const int mch = v->moments_channels;
  float* mom = (float*)h->arena.alloc((size_t)B * hl * hl * mch * 4);
  v->encode_moments(c, xin, B, R, mom);
  // VQ first stage: no distribution, the encoding is the quant_conv output itself (ddpm.py get_first_stage_encoding)
  const bool vq = v->codebook != nullptr;
  launch_posterior_sample(h->st, mom, mch, noise, seed, z0, B, v->desc.embed_dim, hl * hl, scale, (sample && !vq) ? 0 : 1);
  CD_API_END
}

int cd_vae_decode(cd_handle h, int net, const float* z0, int B, int hlat, float scale, float out_mul,
                  float out_add, float* img) {
  CD_API_BEGIN
  enter_engine(h);
  VAE* v = get_vae(h, net);
  CD_CHECK(z0 && img && B > 0 && hlat > 0, "bad argument");
  ArenaScope arena_scope(h->arena);
  Ctx c = h->ctx();
  const int zc = v->desc.embed_dim, cp = round_up(zc, 32), R = hlat * v->factor, co = v->desc.out_channels;
  bf16_t* zin = (bf16_t*)h->arena.alloc((size_t)B * hlat * hlat * cp * (v->f32 ? 4 : 2));
  if (v->codebook)  // VQModelInterface.decode: quantise to the codebook first (autoencoder.py:274-280)
    launch_vq_quantize(h->st, z0, 1.0f / scale, v->codebook, v->n_embed, zc, B, hlat * hlat, zin, cp);
  else if (v->f32)
    launch_nchw_to_nhwc_f32(h->st, z0, (float*)zin, B, zc, hlat * hlat, cp, 1.0f / scale, 0.f, v->x3 ? 1 : 0, h->overflow_dev);
  else
    launch_nchw_to_nhwc(h->st, z0, zin, B, zc, hlat * hlat, cp, 1.0f / scale, 0.f, 0);  // z = 1/scale * z (ddpm.py:705)
  float* o = (float*)h->arena.alloc((size_t)B * R * R * co * 4);
  v->decode(c, zin, B, hlat, o);
  launch_nhwc_to_nchw(h->st, o, 1, co, img, B, co, R * R, out_mul, out_add);
  CD_API_END
}

int cd_dpm_encode(cd_handle h, int net, int sched_kind, const float* x0, const float* ctx_c,
                  const float* ctx_uc, int ctx_len, float guidance, int B, int K,
                  const cd_step_coef* coef_host, const float* noise, uint64_t seed, int last_uses_x0,
                  float* z_out) {
  CD_API_BEGIN
  enter_engine(h);
  // K = 0: only x_T is drawn (white_box_steps = -1 of the text wrappers: every decode step then draws fresh noise)
  CD_CHECK(h && x0 && coef_host && z_out && B > 0 && K >= 0, "bad argument");
  ArenaScope arena_scope(h->arena);
  SamplerState s = setup_sampler(h, net, ctx_c, ctx_uc, ctx_len, guidance, B);
  // the 'ddpm' posterior kernels carry no classifier-free-guidance combine (the pixel DDPMs that use them are unconditional,
  // ddpm_ddim_wrapper.py:230-238): a guided call would silently run unguided
  CD_CHECK(sched_kind == CD_SCHED_DDIM || !s.cfg, "classifier-free guidance is only implemented for sched_kind = CD_SCHED_DDIM");
  s.tab = upload_coef(h, coef_host, K + 1);
  const int64_t chw = (int64_t)s.C * s.HW, n = (int64_t)B * chw;
  const int64_t zbs = (int64_t)(K + 1) * chw;
  launch_init_xt(h->st, x0, noise, seed, 0u, s.xt, z_out, zbs, B, s.C, s.HW, s.tab, K, s.xin16(), s.cpad,
                 s.cfg ? 1 : 0);
  for (int i = 0; i < K; ++i) {
    const int k = K - 1 - i;
    run_unet(h, s, k);
    const int is_last = (last_uses_x0 && k == 0) ? 1 : 0;
    const float* nz = (noise && !is_last) ? noise + (int64_t)(1 + i) * n : nullptr;
    launch_encode_step(h->st, sched_kind, x0, s.xt, s.ehv, nz, seed, (uint32_t)(1 + i), z_out + (1 + i) * chw,
                       zbs, B, s.C, s.HW, s.tab, nullptr, k, is_last, s.xin16(), s.cpad, s.cfg ? 1 : 0);
    h->pacer.tick(h->st);
  }
  CD_API_END
}

static void ddim_decode_impl(cd_handle h, int net, int sched_kind, const float* z, int z_slots, int n_eps,
                             const float* ctx_c, const float* ctx_uc, int ctx_len, float guidance, const float* gvec,
                             int B, int K, const cd_step_coef* coef_host, const float* noise_tail, uint64_t seed,
                             float* x_out) {
  CD_CHECK(h && z && coef_host && x_out && B > 0 && K > 0 && n_eps <= z_slots - 1, "bad argument");
  ArenaScope arena_scope(h->arena);
  SamplerState s = setup_sampler(h, net, ctx_c, ctx_uc, ctx_len, guidance, B);
  // the 'ddpm' posterior kernels carry no classifier-free-guidance combine (the pixel DDPMs that use them are unconditional,
  // ddpm_ddim_wrapper.py:230-238): a guided call would silently run unguided
  CD_CHECK(sched_kind == CD_SCHED_DDIM || !s.cfg, "classifier-free guidance is only implemented for sched_kind = CD_SCHED_DDIM");
  if (gvec) {
    CD_CHECK(s.cfg, "per-sample guidance needs the classifier-free-guidance batch (both contexts)");
    s.ehv.gvec = gvec;
  }
  s.tab = upload_coef(h, coef_host, K);
  const int64_t chw = (int64_t)s.C * s.HW, n = (int64_t)B * chw;
  const int64_t zbs = (int64_t)z_slots * chw;
  // x = z[:, 0]  (sd_wrapper:153; ddpm_ddim_wrapper.py:404)
  HIP_CHECK(hipMemcpy2DAsync(s.xt, chw * 4, z, zbs * 4, chw * 4, B, hipMemcpyDeviceToDevice, h->st));
  if (!s.f32) launch_nchw_to_nhwc(h->st, s.xt, s.xin, B, s.C, s.HW, s.cpad, 1.f, 0.f, s.cfg ? 1 : 0);
  for (int i = 0; i < K; ++i) {
    const int k = K - 1 - i;
    run_unet(h, s, k);
    const float* eps = (i < n_eps) ? z + (int64_t)(1 + i) * chw : nullptr;
    const float* nz = (!eps && noise_tail) ? noise_tail + (int64_t)(i - n_eps) * n : nullptr;
    launch_decode_step(h->st, sched_kind, s.xt, s.ehv, eps, zbs, nz, seed, (uint32_t)(0x1000 + i), B, s.C,
                       s.HW, s.tab, nullptr, k, s.xin16(), s.cpad, s.cfg ? 1 : 0, nullptr);
    h->pacer.tick(h->st);
  }
  HIP_CHECK(hipMemcpyAsync(x_out, s.xt, (size_t)n * 4, hipMemcpyDeviceToDevice, h->st));
}

int cd_ddim_decode(cd_handle h, int net, int sched_kind, const float* z, int z_slots, int n_eps,
                   const float* ctx_c, const float* ctx_uc, int ctx_len, float guidance, int B, int K,
                   const cd_step_coef* coef_host, const float* noise_tail, uint64_t seed, float* x_out) {
  CD_API_BEGIN
  enter_engine(h);
  ddim_decode_impl(h, net, sched_kind, z, z_slots, n_eps, ctx_c, ctx_uc, ctx_len, guidance, nullptr, B, K, coef_host,
                   noise_tail, seed, x_out);
  CD_API_END
}

int cd_ddim_decode_v(cd_handle h, int net, int sched_kind, const float* z, int z_slots, int n_eps,
                     const float* ctx_c, const float* ctx_uc, int ctx_len, const float* guidance_per_sample, int B,
                     int K, const cd_step_coef* coef_host, const float* noise_tail, uint64_t seed, float* x_out) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(guidance_per_sample && ctx_c && ctx_uc, "cd_ddim_decode_v: guidance vector and both contexts are required");
  // any scale other than 0 / 1 selects the [uncond | cond] batch (ddim.py:550-559); the vector supplies the values
  ddim_decode_impl(h, net, sched_kind, z, z_slots, n_eps, ctx_c, ctx_uc, ctx_len, 2.0f, guidance_per_sample, B, K,
                   coef_host, noise_tail, seed, x_out);
  CD_API_END
}

// The coupled source -> target loop (north_star): DPM-Encoder step k and decode step k evaluate the SAME network at the SAME
// timestep, and the decode step only needs eps_k AFTER its forward - so both ride in one U-Net batch
//   [ encoder rows (B, or uncond B | cond B under encoder guidance) | decoder rows (Bd = n_dec * B, or uncond Bd | cond Bd) ]
// followed by the encoder's step kernel (x_{k-1}, eps_k -> z) and then the decoder's (consumes eps_k). 99 forwards of
// B + 2 Bd rows instead of 99 of B and 99 of 2 Bd. Per-sample arithmetic is that of cd_dpm_encode followed by cd_ddim_decode(_v).
int cd_cycle_translate(cd_handle h, int net, int sched_kind, const float* x0, const float* enc_ctx_c,
                       const float* enc_ctx_uc, float enc_guidance, const float* dec_ctx_c, const float* dec_ctx_uc,
                       float dec_guidance, const float* dec_guidance_per_sample, int ctx_len, int B, int n_dec, int K,
                       const cd_step_coef* coef_enc_host, const cd_step_coef* coef_dec_host, const float* noise,
                       uint64_t seed, int last_uses_x0, float* z_out, float* x_out) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && x0 && coef_enc_host && coef_dec_host && z_out && x_out && B > 0 && n_dec > 0 && K > 0, "bad argument");
  ArenaScope arena_scope(h->arena);
  UNet* u = get_unet(h, net);
  const int Bd = B * n_dec;
  const int C = u->desc.in_channels, HW = u->image_size * u->image_size, cpad = u->in_cpad, out_ld = u->out_channels;
  const bool f32 = u->f32;
  Guidance ge = resolve_guidance(enc_ctx_c, enc_ctx_uc, enc_guidance);
  Guidance gd = resolve_guidance(dec_ctx_c, dec_ctx_uc, dec_guidance_per_sample ? 2.0f : dec_guidance);
  if (dec_guidance_per_sample) CD_CHECK(gd.cfg, "per-sample guidance needs the classifier-free-guidance batch (both contexts)");
  CD_CHECK(sched_kind == CD_SCHED_DDIM || (!ge.cfg && !gd.cfg), "classifier-free guidance is only implemented for sched_kind = CD_SCHED_DDIM");
  const bool has_ctx = enc_ctx_c || enc_ctx_uc;
  CD_CHECK(has_ctx == (dec_ctx_c || dec_ctx_uc), "cd_cycle_translate: contexts for both passes or for neither");
  const int Ben = ge.cfg ? 2 * B : B, Bdn = gd.cfg ? 2 * Bd : Bd, Bn = Ben + Bdn;
  Ctx c = h->ctx();
  if (has_ctx) {
    const int Dc = u->desc.context_dim;
    CD_CHECK(Dc > 0 && ctx_len > 0, "network has no cross-attention but a context was given");
    // one context tensor in batch-row order: [enc (uncond | cond) | dec (uncond | cond)]
    const size_t esz = f32 ? 4 : 2, per = (size_t)ctx_len * Dc;
    char* cx = (char*)h->arena.alloc((size_t)Bn * per * esz);
    auto put = [&](const float* src, int rows, size_t row0) {
      if (f32) HIP_CHECK(hipMemcpyAsync(cx + row0 * per * 4, src, (size_t)rows * per * 4, hipMemcpyDeviceToDevice, h->st));
      else launch_nchw_to_nhwc(h->st, src, (bf16_t*)cx + row0 * per, rows * ctx_len, Dc, 1, Dc, 1.f, 0.f, 0);
    };
    if (ge.cfg) { put(enc_ctx_uc, B, 0); put(enc_ctx_c, B, B); } else put(ge.ctx_single, B, 0);
    if (gd.cfg) { put(dec_ctx_uc, Bd, Ben); put(dec_ctx_c, Bd, Ben + Bd); } else put(gd.ctx_single, Bd, Ben);
    u->set_context(c, (const bf16_t*)cx, Bn, ctx_len);
  } else {
    CD_CHECK(u->desc.context_dim <= 0 || !u->desc.use_spatial_transformer, "network expects a cross-attention context");
  }
  const size_t esz = f32 ? 4 : 2;
  const int64_t chw = (int64_t)C * HW, n = (int64_t)B * chw;
  float* xt_e = (float*)h->arena.alloc((size_t)B * chw * 4);
  float* xt_d = (float*)h->arena.alloc((size_t)Bd * chw * 4);
  char* xin = (char*)h->arena.alloc((size_t)Bn * HW * cpad * esz);
  HIP_CHECK(hipMemsetAsync(xin, 0, (size_t)Bn * HW * cpad * esz, h->st));
  float* eh = (float*)h->arena.alloc((size_t)Bn * HW * out_ld * 4);
  const size_t row_in = (size_t)HW * cpad * esz;  // bytes of one sample of the network input
  bf16_t* xin_e = f32 ? nullptr : (bf16_t*)xin;
  bf16_t* xin_d = f32 ? nullptr : (bf16_t*)(xin + (size_t)Ben * row_in);
  EpsHat eh_e, eh_d;
  eh_e.p = eh; eh_e.sb = (int64_t)HW * out_ld; eh_e.sc = 1; eh_e.sp = out_ld; eh_e.cfg = ge.cfg ? 1 : 0; eh_e.g = enc_guidance;
  eh_d = eh_e; eh_d.p = eh + (int64_t)Ben * HW * out_ld; eh_d.cfg = gd.cfg ? 1 : 0; eh_d.g = dec_guidance;
  eh_d.gvec = dec_guidance_per_sample;
  StepCoef* tab_e = upload_coef(h, coef_enc_host, K + 1);
  StepCoef* tab_d = upload_coef(h, coef_dec_host, K);
  const int64_t zbs = (int64_t)(K + 1) * chw;
  // x_T (ddim.py:477-479), z[:, 0]; the decoder starts from the same tensor (sd_wrapper:153), once per decoder scale
  launch_init_xt(h->st, x0, noise, seed, 0u, xt_e, z_out, zbs, B, C, HW, tab_e, K, xin_e, cpad, ge.cfg ? 1 : 0);
  for (int j = 0; j < n_dec; ++j)
    HIP_CHECK(hipMemcpyAsync(xt_d + (int64_t)j * n, xt_e, (size_t)n * 4, hipMemcpyDeviceToDevice, h->st));
  if (!f32) launch_nchw_to_nhwc(h->st, xt_d, xin_d, Bd, C, HW, cpad, 1.f, 0.f, gd.cfg ? 1 : 0);
  // rows that repeat the rows just ahead of them (the decoder's cond half repeats its uncond half): the network computes
  // everything ahead of the first cross-attention once for them - only when the encoder half has no such pair of its own
  const int dup_tail = (gd.cfg && !ge.cfg) ? Bd : 0;
  for (int i = 0; i < K; ++i) {
    const int k = K - 1 - i;
    if (f32) {
      float* xf = (float*)xin;
      launch_nchw_to_nhwc_f32(h->st, xt_e, xf, B, C, HW, cpad, 1.f, 0.f, u->x3 ? 1 : 0, h->overflow_dev);
      if (ge.cfg) launch_nchw_to_nhwc_f32(h->st, xt_e, xf + (size_t)B * HW * cpad, B, C, HW, cpad, 1.f, 0.f, u->x3 ? 1 : 0, h->overflow_dev);
      float* xd = xf + (size_t)Ben * HW * cpad;
      launch_nchw_to_nhwc_f32(h->st, xt_d, xd, Bd, C, HW, cpad, 1.f, 0.f, u->x3 ? 1 : 0, h->overflow_dev);
      if (gd.cfg) launch_nchw_to_nhwc_f32(h->st, xt_d, xd + (size_t)Bd * HW * cpad, Bd, C, HW, cpad, 1.f, 0.f, u->x3 ? 1 : 0, h->overflow_dev);
    }
    UNetIO io;
    io.xin = (const bf16_t*)xin; io.B = Bn; io.tab = tab_d; io.step = k; io.t_shared = true;  // rows k of both tables carry the same t
    io.dup_tail = dup_tail;
    io.out = eh; io.out_ld = out_ld;
    u->forward(c, io);
    const int is_last = (last_uses_x0 && k == 0) ? 1 : 0;
    const float* nz = (noise && !is_last) ? noise + (int64_t)(1 + i) * n : nullptr;
    float* zslot = z_out + (1 + i) * chw;
    launch_encode_step(h->st, sched_kind, x0, xt_e, eh_e, nz, seed, (uint32_t)(1 + i), zslot, zbs, B, C, HW, tab_e, nullptr, k,
                       is_last, xin_e, cpad, ge.cfg ? 1 : 0);
    launch_decode_step(h->st, sched_kind, xt_d, eh_d, zslot, zbs, nullptr, seed, (uint32_t)(0x1000 + i), Bd, C, HW, tab_d,
                       nullptr, k, xin_d, cpad, gd.cfg ? 1 : 0, nullptr, /*eps_bmod=*/n_dec > 1 ? B : 0);
    h->pacer.tick(h->st);
  }
  HIP_CHECK(hipMemcpyAsync(x_out, xt_d, (size_t)Bd * chw * 4, hipMemcpyDeviceToDevice, h->st));
  CD_API_END
}

int cd_pix_refine(cd_handle h, int net, int sched_kind, float* x, int B, int R, const cd_step_coef* coef_host,
                  const float* noise, uint64_t seed) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && x && coef_host && B > 0 && R > 0, "bad argument");
  ArenaScope arena_scope(h->arena);
  SamplerState s = setup_sampler(h, net, nullptr, nullptr, 0, 1.f, B);
  s.tab = upload_coef(h, coef_host, R + 1);
  const int64_t n = (int64_t)B * s.C * s.HW;
  launch_init_xt(h->st, x, noise, seed, 0x2000u, s.xt, nullptr, 0, B, s.C, s.HW, s.tab, R, s.xin16(), s.cpad, 0);
  for (int i = 0; i < R; ++i) {
    const int k = R - 1 - i;
    run_unet(h, s, k);
    const float* nz = noise ? noise + (int64_t)(1 + i) * n : nullptr;
    launch_decode_step(h->st, sched_kind, s.xt, s.ehv, nullptr, 0, nz, seed, (uint32_t)(0x2001 + i), B, s.C,
                       s.HW, s.tab, nullptr, k, s.xin16(), s.cpad, 0, nullptr);
    h->pacer.tick(h->st);
  }
  HIP_CHECK(hipMemcpyAsync(x, s.xt, (size_t)n * 4, hipMemcpyDeviceToDevice, h->st));
  CD_API_END
}

// ------------------------------------------------------------------ single-kernel entry points
int cd_op_pack_conv_weight(cd_handle h, const float* w_host, int N, int Cin, int KH, int KW, int geglu,
                           void** packed_dev, int* Npad, int* Cpad) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && w_host && packed_dev, "bad argument");
  static int counter = 0;
  ConvW* c = h->op_params.new_conv(N, Cin, KH, KW, false, geglu != 0);
  const std::string name = "op." + std::to_string(counter++);
  h->op_params.conv_weight(name, c);
  int64_t shape[4] = {N, Cin, KH, KW};
  h->op_params.load(h->st, name, w_host, 4, shape);
  *packed_dev = c;
  if (Npad) *Npad = c->Npad;
  if (Cpad) *Cpad = c->Cpad;
  CD_API_END
}

int cd_op_free(cd_handle, void*) { return 0; }  // op weights live until the engine is destroyed

static void op_conv2d(cd_handle h, const float* x0, int C0, const float* x1, int C1, int B, int H, int W,
                      const void* packed_w, int N, int KH, int KW, int stride, int pad, int asym_pad, int up,
                      const float* bias, const float* rowvec, const float* resid, int act, int tile, bool out16,
                      float* y, float* stats) {
  CD_CHECK(h && x0 && packed_w && y, "bad argument");
  ArenaScope arena_scope(h->arena);
  Ctx c = h->ctx();
  ConvW w = *(const ConvW*)packed_w;
  CD_CHECK(w.N == N && w.KH == KH && w.KW == KW, "packed weight does not match the call");
  w.b = const_cast<float*>(bias);
  const int c0p = x1 ? C0 : round_up(C0, 32);
  Act a0 = alloc_act(c, B, H, W, c0p);
  launch_nchw_to_nhwc(h->st, x0, a0.p, B, C0, H * W, c0p, 1.f, 0.f, 0);
  Act a1;
  if (x1) {
    a1 = alloc_act(c, B, H, W, C1);
    launch_nchw_to_nhwc(h->st, x1, a1.p, B, C1, H * W, C1, 1.f, 0.f, 0);
  }
  ConvOpts o; o.stride = stride; o.pad = pad; o.asym = asym_pad != 0; o.up = up != 0; o.tile = tile;
  if (act & 0x400) {  // LayerNorm (no gain / bias) of the input rows inside the kernel: streaming linear kernel only
    CD_CHECK(out16 && conv_ln_fold_available(c, w, (int64_t)B * H * W) , "LayerNorm fold needs tile 30 shapes");
    o.ln_fold = true;
    act &= ~0x400;
  }
  o.act = act;
  o.out_f32 = !out16;
  o.want_stats = stats != nullptr;
  const int Hin = up ? 2 * H : H, Win = up ? 2 * W : W;
  const int Ho = asym_pad ? (Hin + 1 - KH) / stride + 1 : (Hin + 2 * pad - KH) / stride + 1;
  const int Wo = asym_pad ? (Win + 1 - KW) / stride + 1 : (Win + 2 * pad - KW) / stride + 1;
  const int Nout = w.geglu ? N / 2 : N;
  if (rowvec) { o.rowvec = rowvec; o.rowvec_ld = N; o.rows_per_vec = Ho * Wo; }
  Act r;
  if (resid) {
    r = alloc_act(c, B, Ho, Wo, Nout);
    launch_nchw_to_nhwc(h->st, resid, r.p, B, Nout, Ho * Wo, Nout, 1.f, 0.f, 0);
    o.resid = &r;
  }
  Act out = conv_fwd(c, w, a0, x1 ? &a1 : nullptr, o);
  launch_nhwc_to_nchw(h->st, out.p, out16 ? 0 : 1, out.ld, y, B, Nout, Ho * Wo, 1.f, 0.f);
  if (stats) {
    CD_CHECK(out.stats, "this convolution produces no GroupNorm statistics (GEGLU, or rows %% 32 != 0)");
    HIP_CHECK(hipMemcpyAsync(stats, out.stats, (size_t)(out.rows() / 32) * 2 * Nout * sizeof(float),
                             hipMemcpyDeviceToDevice, h->st));
  }
}

int cd_op_conv2d(cd_handle h, const float* x0, int C0, const float* x1, int C1, int B, int H, int W,
                 const void* packed_w, int N, int KH, int KW, int stride, int pad, int asym_pad, int up,
                 const float* bias, const float* rowvec, const float* resid, int act, int tile, float* y) {
  CD_API_BEGIN
  enter_engine(h);
  op_conv2d(h, x0, C0, x1, C1, B, H, W, packed_w, N, KH, KW, stride, pad, asym_pad, up, bias, rowvec, resid, act, tile,
            false, y, nullptr);
  CD_API_END
}

int cd_op_conv2d_16(cd_handle h, const float* x0, int C0, const float* x1, int C1, int B, int H, int W,
                    const void* packed_w, int N, int KH, int KW, int stride, int pad, int asym_pad, int up,
                    const float* bias, const float* rowvec, const float* resid, int act, int tile, float* y,
                    float* stats) {
  CD_API_BEGIN
  enter_engine(h);
  op_conv2d(h, x0, C0, x1, C1, B, H, W, packed_w, N, KH, KW, stride, pad, asym_pad, up, bias, rowvec, resid, act, tile,
            true, y, stats);
  CD_API_END
}

int cd_op_groupnorm(cd_handle h, const float* x, int B, int C, int H, int W, int G, float eps,
                    const float* gamma, const float* beta, const float* film, int silu, float* y) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && x && y && G == 32, "bad argument (G must be 32)");
  ArenaScope arena_scope(h->arena);
  Ctx c = h->ctx();
  Act a = alloc_act(c, B, H, W, C);
  launch_nchw_to_nhwc(h->st, x, a.p, B, C, H * W, C, 1.f, 0.f, 0);
  GNW w; w.g = const_cast<float*>(gamma); w.b = const_cast<float*>(beta); w.C = C; w.eps = eps;
  Act o = groupnorm_fwd(c, w, a, nullptr, silu != 0, film, film ? 2 * C : 0);
  launch_nhwc_to_nchw(h->st, o.p, 0, o.ld, y, B, C, H * W, 1.f, 0.f);
  CD_API_END
}

int cd_op_layernorm(cd_handle h, const float* x, int rows, int C, const float* gamma, const float* beta,
                    float eps, float* y) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && x && y, "bad argument");
  ArenaScope arena_scope(h->arena);
  bf16_t* a = (bf16_t*)h->arena.alloc((size_t)rows * C * 2);
  bf16_t* o = (bf16_t*)h->arena.alloc((size_t)rows * C * 2);
  launch_nchw_to_nhwc(h->st, x, a, rows, C, 1, C, 1.f, 0.f, 0);
  launch_layernorm(h->st, a, C, o, C, rows, C, gamma, beta, eps);
  launch_nhwc_to_nchw(h->st, o, 0, C, y, rows, C, 1, 1.f, 0.f);
  CD_API_END
}

int cd_op_attention(cd_handle h, const float* q, const float* k, const float* v, int B, int H, int Tq,
                    int Tk, int D, float scale, int use_transpose_kernel, float* o) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && q && k && v && o, "bad argument");
  ArenaScope arena_scope(h->arena);
  const int C = H * D, Tpad = round_up(Tk, 64);
  bf16_t* qb = (bf16_t*)h->arena.alloc((size_t)B * Tq * C * 2);
  bf16_t* kb = (bf16_t*)h->arena.alloc((size_t)B * Tk * C * 2);
  bf16_t* vb = (bf16_t*)h->arena.alloc((size_t)B * Tk * C * 2);
  bf16_t* ob = (bf16_t*)h->arena.alloc((size_t)B * Tq * C * 2);
  launch_nchw_to_nhwc(h->st, q, qb, B * Tq, C, 1, C, 1.f, 0.f, 0);
  launch_nchw_to_nhwc(h->st, k, kb, B * Tk, C, 1, C, 1.f, 0.f, 0);
  launch_nchw_to_nhwc(h->st, v, vb, B * Tk, C, 1, C, 1.f, 0.f, 0);
  AttnParams p;
  p.q = qb; p.k = kb; p.o = ob; p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.D = D;
  p.ldq = C; p.ldk = C; p.ldo = C; p.q_bs = (int64_t)Tq * C; p.k_bs = (int64_t)Tk * C; p.o_bs = (int64_t)Tq * C;
  p.scale = scale;
  if (use_transpose_kernel) {  // V^T [B][H][D][Tpad] layout (k_transpose_v), else token-major V (LDS transpose reads)
    bf16_t* vt = (bf16_t*)h->arena.alloc((size_t)B * C * Tpad * 2);
    launch_transpose_v(h->st, vb, C, (int64_t)Tk * C, vt, B, H, Tk, D, D, Tpad);
    p.vt = vt; p.vt_dpad = D; p.vt_tpad = Tpad;
  } else {
    p.v = vb; p.ldv = C; p.v_bs = (int64_t)Tk * C;
  }
  launch_attention(h->st, p);
  launch_nhwc_to_nchw(h->st, ob, 0, C, o, B * Tq, C, 1, 1.f, 0.f);
  CD_API_END
}

int cd_op_softmax_rows(cd_handle h, const float* s, int64_t rows, int cols, float* p) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && s && p, "bad argument");
  ArenaScope arena_scope(h->arena);
  bf16_t* pb = (bf16_t*)h->arena.alloc((size_t)rows * cols * 2);
  launch_softmax_rows(h->st, s, cols, pb, cols, rows, cols);
  launch_nhwc_to_nchw(h->st, pb, 0, cols, p, (int)rows, cols, 1, 1.f, 0.f);
  CD_API_END
}

int cd_op_timestep_embedding(cd_handle h, const float* t, int B, int dim, int mode, float* out) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && t && out, "bad argument");
  launch_timestep_embedding(h->st, nullptr, nullptr, 0, t, out, B, dim, mode);
  CD_API_END
}

int cd_op_sched_step(cd_handle h, int mode, int sched_kind, const cd_step_coef* coef_host, const float* x0,
                     float* xt, const float* eps_hat, int cfg, float guidance, const float* noise,
                     const float* eps_in, int is_last, int B, int C, int HW, float* z_slot) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && coef_host && xt, "bad argument");
  ArenaScope arena_scope(h->arena);
  StepCoef* tab = upload_coef(h, coef_host, 1);
  const int64_t chw = (int64_t)C * HW;
  EpsHat eh; eh.p = eps_hat; eh.sb = chw; eh.sc = HW; eh.sp = 1; eh.cfg = cfg; eh.g = guidance;  // NCHW view
  if (mode == 0) {
    launch_init_xt(h->st, x0, noise, 0, 0, xt, z_slot, chw, B, C, HW, tab, 0, nullptr, 0, 0);
  } else if (mode == 1) {
    launch_encode_step(h->st, sched_kind, x0, xt, eh, noise, 0, 0, z_slot, chw, B, C, HW, tab, nullptr, 0,
                       is_last, nullptr, 0, 0);
  } else {
    launch_decode_step(h->st, sched_kind, xt, eh, eps_in, chw, noise, 0, 0, B, C, HW, tab, nullptr, 0, nullptr,
                       0, 0, nullptr);
  }
  HIP_CHECK(hipStreamSynchronize(h->st));
  CD_API_END
}

}  // extern "C"

// ------------------------------------------------------------------ hardware layout probes
namespace {
__global__ void k_probe_mfma(float* out) {
  // which=0: D = A.B with A[i][0] = i+1 (else 0), B[0][j] = 1  -> D[i][j] = i+1 (row map)
  //          then A[i][0] = 1, B[0][j] = j+1                   -> D[i][j] = j+1 (col map)
  const int lane = threadIdx.x & 63;
  bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (lane < 32) { a[0] = (short)f2bf((float)(lane + 1)); b[0] = (short)f2bf(1.0f); }
  acc = CD_MFMA_32x32x16(a, b, acc);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (lane < 32) { a[0] = (short)f2bf(1.0f); b[0] = (short)f2bf((float)(lane + 1)); }
  acc = CD_MFMA_32x32x16(a, b, acc);
  for (int r = 0; r < 16; ++r) out[1024 + lane * 16 + r] = acc[r];
  // k-slot check: A[i][k] = 1 for all i, only k = kk set; B[kk][j] = kk+1 -> D = sum over matching k
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int j = 0; j < 8; ++j) { a[j] = (short)f2bf(1.0f); b[j] = (short)f2bf((float)(8 * (lane >> 5) + j + 1)); }
  acc = CD_MFMA_32x32x16(a, b, acc);
  for (int r = 0; r < 16; ++r) out[2048 + lane * 16 + r] = acc[r];  // expect 1+2+...+16 = 136
}
__global__ void k_probe_tr(float* out) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[256];
  const int lane = threadIdx.x & 63;
  for (int i = lane; i < 256; i += 64) lds[i] = f2bf((float)i);
  __syncthreads();
  uint2 v;
  const unsigned addr = (unsigned)(size_t)(lds) + lane * 8;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[lane * 4 + 0] = bf2f((bf16_t)(v.x & 0xffff));
  out[lane * 4 + 1] = bf2f((bf16_t)(v.x >> 16));
  out[lane * 4 + 2] = bf2f((bf16_t)(v.y & 0xffff));
  out[lane * 4 + 3] = bf2f((bf16_t)(v.y >> 16));
}
// The address pattern k_attention uses for its PV A operand (token-major V tile [64 keys][96] in LDS): 16 lanes fetch a
// 4-key x 16-d block, lane l must receive column d = dt*32 + (l & 31) of keys base + 4*(l >> 5) + 0..3.
__global__ void k_probe_tr_attn(float* out) {
  constexpr int VS = 96;
  __shared__ __attribute__((aligned(16))) bf16_t lds[64 * VS];
  const int lane = threadIdx.x & 63;
  for (int i = lane; i < 64 * VS; i += 64) lds[i] = (bf16_t)((i / VS) * 64 + (i % VS < 64 ? i % VS : 0));  // raw key*64+d
  __syncthreads();
  typedef __attribute__((address_space(3))) bf16x4* lds4_t;
  const int half = lane >> 5;
  const int vtr_off = (4 * half + ((lane & 15) >> 2)) * VS + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  int n = 0;
  for (int sel = 0; sel < 2; ++sel) {  // (kh, s2, dt) = (0, 0, 0) and (1, 1, 1)
    const bf16_t* vr = lds + vtr_off + (sel * 32 + 16 * sel) * VS + sel * 32;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)vr);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vr + 8 * VS));
    for (int j = 0; j < 4; ++j) out[lane * 16 + n++] = (float)(unsigned short)lo[j];
    for (int j = 0; j < 4; ++j) out[lane * 16 + n++] = (float)(unsigned short)hi[j];
  }
}
}  // namespace

namespace {
__global__ void k_fill_hash16(bf16_t* p, int64_t n, uint32_t seed, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = f2bf(((float)(x & 0xffff) / 32768.0f - 1.0f) * scale);  // uniform [-scale, scale): full-range data
  }
}
}  // namespace

// Micro-benchmark of one conv / GEMM shape on synthetic full-range data: `iters` back-to-back launches
// between two HIP events on the engine stream. ms_out = average per launch.
extern "C" int cd_op_bench_conv(cd_handle h, int B, int H, int W, int C0, int C1, int N, int k, int stride,
                                int up, int act, int tile, int iters, float* ms_out) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && ms_out && iters > 0, "bad argument");
  ArenaScope arena_scope(h->arena);
  Ctx c = h->ctx();
  ConvW w;
  w.N = N; w.Cin = C0 + C1; w.KH = k; w.KW = k; w.Cpad = round_up(C0 + C1, 32); w.Npad = round_up(N, 128);
  w.geglu = (act == ACT_GEGLU);
  const size_t wn = (size_t)w.Npad * w.Ktot();
  w.w = (bf16_t*)h->arena.alloc(wn * 2);
  w.b = (float*)h->arena.alloc((size_t)w.Npad * 4);
  HIP_CHECK(hipMemsetAsync(w.b, 0, (size_t)w.Npad * 4, h->st));
  hipLaunchKernelGGL(k_fill_hash16, dim3(1024), dim3(256), 0, h->st, w.w, (int64_t)wn, 17u,
                     1.0f / sqrtf((float)w.Ktot()));
  if (k == 1 && w.Cpad == 320) {
    w.wfrag = (bf16_t*)h->arena.alloc(wn * 2);
    launch_pack_wfrag(h->st, w.w, w.Ktot(), w.wfrag, w.Npad);
  }
  const bool inplace_resid = (act & 0x100) != 0, want_stats = (act & 0x200) != 0;
  act &= 0xff;
  w.geglu = (act == ACT_GEGLU);
  Act a0 = alloc_act(c, B, H, W, C1 ? C0 : round_up(C0, 32));
  hipLaunchKernelGGL(k_fill_hash16, dim3(1024), dim3(256), 0, h->st, a0.p, (int64_t)a0.rows() * a0.C, 3u, 1.0f);
  Act a1;
  if (C1) {
    a1 = alloc_act(c, B, H, W, C1);
    hipLaunchKernelGGL(k_fill_hash16, dim3(1024), dim3(256), 0, h->st, a1.p, (int64_t)a1.rows() * a1.C, 5u, 1.0f);
  }
  ConvOpts o; o.stride = stride; o.pad = k / 2; o.up = up != 0; o.act = (act == ACT_GEGLU) ? ACT_NONE : act; o.tile = tile;
  Act ro;
  if (inplace_resid || want_stats) {  // the in-place residual update of the to_out / proj_out projections
    const int ho = up ? 2 * H : (H + 2 * o.pad - k) / stride + 1;
    ro = alloc_act(c, B, ho, ho, N, /*with_stats=*/true);
    hipLaunchKernelGGL(k_fill_hash16, dim3(1024), dim3(256), 0, h->st, ro.p, (int64_t)ro.rows() * ro.C, 7u, 0.01f);
    o.out = ro.p; o.out_ld = ro.ld;
    if (inplace_resid) o.resid = &ro;
    if (want_stats) o.out_stats = ro.stats_buf;
  }
  const char* probe_path = getenv("CYCLEDIFF_PROBE_OUT");
  const size_t probe_words = (size_t)8192 * 8 * kProbeWords;  // up to 8192 workgroups of 8 waves
  unsigned long long* probe_buf = nullptr;
  if (probe_path) {
    probe_buf = (unsigned long long*)h->arena.alloc(probe_words * 8);
    HIP_CHECK(hipMemsetAsync(probe_buf, 0, probe_words * 8, h->st));
  }
  const size_t mk2 = h->arena.mark();
  conv_fwd(c, w, a0, C1 ? &a1 : nullptr, o);  // warm-up
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
  HIP_CHECK(hipEventRecord(e0, h->st));
  for (int i = 0; i < iters; ++i) { h->arena.release(mk2); conv_fwd(c, w, a0, C1 ? &a1 : nullptr, o); }
  HIP_CHECK(hipEventRecord(e1, h->st));
  HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0;
  HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  // phase-timing build (lib/libcyclediff_probe.so): one more launch whose waves leave their s_memtime stamps, dumped as
  // text header + raw 64-bit words to $CYCLEDIFF_PROBE_OUT (scripts/probe_report.py reads it)
  if (probe_buf) {
    const char* path = probe_path;
    const size_t words = probe_words;
    unsigned long long* buf = probe_buf;
    g_conv_probe = buf;
    h->arena.release(mk2);
    conv_fwd(c, w, a0, C1 ? &a1 : nullptr, o);
    g_conv_probe = nullptr;
    std::vector<unsigned long long> host(words);
    HIP_CHECK(hipMemcpyAsync(host.data(), buf, words * 8, hipMemcpyDeviceToHost, h->st));
    HIP_CHECK(hipStreamSynchronize(h->st));
    if (FILE* f = fopen(path, "wb")) {
      fprintf(f, "probe words=%d cfg=\"%s\" B=%d H=%d C0=%d C1=%d N=%d k=%d act=%d ms=%.4f\n", kProbeWords,
              conv_gemm_last_config(), B, H, C0, C1, N, k, act, ms / iters);
      size_t used = words;
      while (used > 0 && host[used - 1] == 0) --used;
      used = (used + kProbeWords - 1) / kProbeWords * kProbeWords;
      fwrite(host.data(), 8, used, f);
      fclose(f);
    }
  }
  CD_API_END
}

extern "C" int cd_op_bench_mfma_sustained(cd_handle h, int target_ms, float* tflops_out, float* ghz_out) {
  CD_API_BEGIN
  enter_engine(h);
  CD_CHECK(h && tflops_out && ghz_out && target_ms > 0 && target_ms <= 5000, "bad argument");
  launch_mfma_sustained(h->st, target_ms, tflops_out, ghz_out);
  CD_API_END
}

extern "C" int cd_op_probe(cd_handle h, int which, const void* in, void* out, size_t n) {
  CD_API_BEGIN
  enter_engine(h);
  (void)in;
  CD_CHECK(h && out, "bad argument");
  if (which == 0) {
    CD_CHECK(n >= 3072 * sizeof(float), "probe 0 needs 3072 floats");
    hipLaunchKernelGGL(k_probe_mfma, dim3(1), dim3(64), 0, h->st, (float*)out);
  } else if (which == 1) {
    CD_CHECK(n >= 256 * sizeof(float), "probe 1 needs 256 floats");
    hipLaunchKernelGGL(k_probe_tr, dim3(1), dim3(64), 0, h->st, (float*)out);
  } else if (which == 2) {
    CD_CHECK(n >= 1024 * sizeof(float), "probe 2 needs 1024 floats");
    hipLaunchKernelGGL(k_probe_tr_attn, dim3(1), dim3(64), 0, h->st, (float*)out);
  } else {
    CD_CHECK(false, "unknown probe %d", which);
  }
  CD_API_END
}
