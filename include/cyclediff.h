/* libcyclediff — C ABI of the MI355X-native CycleDiffusion engine.
 *
 * The reference (ChenWu98/cycle-diffusion) is pure Python/PyTorch and has no FFI; this header is
 * the seam a maintainer binds (ctypes) underneath the reference's plugin API:
 *   model/gan_wrapper/stable_diffusion_stochastic_text_wrapper.py:102-253  (SDStochasticTextWrapper)
 *   model/gan_wrapper/latentdiff_stochastic_text_wrapper.py               (LatentDiffStochasticTextWrapper)
 *   model/gan_wrapper/ddpm_ddim_wrapper.py:317-542                         (DDPMDDIMWrapper)
 * Each entry point names the reference function it replaces.
 *
 * Conventions: every pointer is a DEVICE pointer unless its name ends in _host; tensors at the
 * boundary are fp32, NCHW, contiguous (the reference's layout); all calls are asynchronous on the
 * engine's HIP stream; return value 0 = ok, non-zero = error with text in cd_last_error();
 * no exception crosses the ABI; one handle per rank / stream. A handle is not thread safe (one host
 * thread at a time), but DIFFERENT handles are independent - own stream, workspace, split-K scratch -
 * and may be driven concurrently from different host threads on the same GPU (several batches in
 * flight; tests/test_gpu_concurrency.py). The caller owns every buffer it passes, the engine owns
 * weights and workspace.
 */
#ifndef CYCLEDIFF_H
#define CYCLEDIFF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cd_engine* cd_handle;

enum { CD_NET_UNET_OPENAI = 1, CD_NET_UNET_HO = 2, CD_NET_VAE_KL = 3, CD_NET_CLIP_TEXT = 4, CD_NET_BERT_XTR = 5,
       CD_NET_OCLIP_TEXT = 6, CD_NET_OCLIP_VISION = 7 };
enum { CD_SCHED_DDIM = 0, CD_SCHED_DDPM = 1 };
enum { CD_PREC_16 = 0, CD_PREC_F32 = 1, CD_PREC_F32X3 = 2 };

/* Architecture descriptor (the hyper-parameters of the reference's YAML / dict configs):
 *   UNET_OPENAI : ldm/modules/diffusionmodules/openaimodel.py:413-470 (SD v1, LDM text2img) and
 *                 model/lib/ddpm_ddim/models/improved_ddpm/unet.py:401-470 (i_DDPM AFHQ)
 *   UNET_HO     : model/lib/ddpm_ddim/models/ddpm/diffusion.py:192-290
 *   VAE_KL      : ldm/models/autoencoder.py:285-333 + diffusionmodules/model.py:368-568; with n_embed > 0 the VQ-f4
 *                 first stage (VQModelInterface) of the unconditional LDMs
 *   CLIP_TEXT   : the HF `CLIPTextModel` behind FrozenCLIPEmbedder (ldm/modules/encoders/modules.py:136-161);
 *                 descriptor fields reused: model_channels = width (768), num_res_blocks = layers (12),
 *                 num_heads (12), context_dim = MLP width (3072), in_channels = vocabulary (49408),
 *                 image_size = positions (77); weights keyed by `text_model.*` (HF state_dict names)
 *   BERT_XTR    : BERTEmbedder.transformer, the x-transformers TransformerWrapper(Encoder(dim 1280, depth 32)) of
 *                 LDM text2img (model/lib/latentdiff/ldm/modules/encoders/modules.py:75-98); same fields plus
 *                 num_head_channels = dim_head (64; heads 8 -> inner 512); weights keyed by `token_emb`,
 *                 `pos_emb.emb`, `attn_layers.layers.*`, `norm`
 *   OCLIP_TEXT / OCLIP_VISION : the text and image towers of OpenAI CLIP (ViT-B/32 in the reference:
 *                 model/energy/clean_clip.py:10 `clip.load("ViT-B/32")`), weights keyed by the openai/CLIP package's
 *                 state_dict names; out_channels = embedding width (512); VISION: image_size = input resolution
 *                 (224), z_channels = patch size (32), in_channels = 3 */
typedef struct cd_net_desc {
  int kind;
  int image_size;          /* spatial size of the network input (latent 64, pixel 256, ...)      */
  int in_channels, out_channels;
  int model_channels;      /* `model_channels` / `ch`                                            */
  int num_res_blocks;
  int n_mult;  int channel_mult[8];
  int n_attn;  int attn[8];/* OPENAI: downsample rates with attention; HO/VAE: resolutions       */
  int num_heads;           /* -1 when num_head_channels is used                                   */
  int num_head_channels;   /* -1 when num_heads is used                                           */
  int use_spatial_transformer, context_dim, transformer_depth;
  int use_scale_shift_norm, resblock_updown, conv_resample;
  /* VAE */
  int z_channels, embed_dim, double_z;
  /* Storage / arithmetic of the network: CD_PREC_16 = 16-bit activations and weights, fp32 accumulate (default);
   * CD_PREC_F32 = fp32 activations, weights and matrix instructions - what the reference itself computes in
   * (`use_fp16=False`, improved_ddpm/script_util.py:15; `precision = "full"`,
   * stable_diffusion_stochastic_text_wrapper.py:117). U-Nets only: the pixel-space DDPMs of ddpm_ddim_wrapper.py, whose
   * 'ddim' chain needs eps_hat at fp32 resolution, and - round 4 - the text-conditioned SD / LDM U-Nets
   * (SpatialTransformer blocks with fp32 LayerNorm, fp32 flash attention and exact-erf GEGLU, csrc/st_f32.hip), for which
   * it restores the encode -> decode cycle the 16-bit engine only closes to 2e-2;
   * CD_PREC_F32X3 = the fp32 network with its GroupNorm- / LayerNorm-fed convolutions and projections evaluated as
   * three-term split-fp16 products on the 16-bit matrix cores (x = hi + lo, w = wh + wl; hi.wh + lo.wh + hi.wl, fp32
   * accumulate: 2^-22 per product instead of 2^-24; everything else as CD_PREC_F32). fp16 build of the library only. */
  int precision;
  /* VAE_KL only: > 0 selects the VQ first stage of the unconditional LDMs (VQModelInterface,
   * model/lib/latentdiff/ldm/models/autoencoder.py:264-282, `n_embed` codebook rows of width embed_dim): double_z = 0,
   * encode = encoder + quant_conv (no sampling), decode = nearest-codebook quantisation + post_quant_conv + decoder */
  int n_embed;
  int reserved[6];
} cd_net_desc;

/* Per-step scheduler coefficients, evaluated by the host in fp32 in the reference's operation
 * order (ddim.py:570-579 / ddpm_ddim_wrapper.py:291-302); see csrc/kernels.h StepCoef. */
typedef struct cd_step_coef {
  float sa, s1a, sap, dirc, sigma, r, t_mask;
  int32_t t;
} cd_step_coef;

const char* cd_last_error(void);
int cd_version(void);
/* 16-bit storage format of activations / packed weights inside the engine: 1 = IEEE fp16, 0 = bfloat16 */
int cd_act_format(void);

/* engine lifetime; `hip_stream` is a hipStream_t (0 = default stream) */
int cd_engine_create(void* hip_stream, size_t workspace_bytes, cd_handle* out);
int cd_engine_destroy(cd_handle h);
int cd_engine_workspace_high_water(cd_handle h, size_t* bytes);
/* wait for everything queued on the engine's stream, SLEEPING (blocking-sync event) instead of spinning: what a rank
 * calls where the reference's Trainer would call torch.cuda.synchronize() (trainer/trainer.py:1055-1062 times
 * evaluate() around it). The sampler entry points also pace themselves: the host never runs more than two sampler
 * steps ahead of the GPU (CYCLEDIFF_HOST_PACING=0 turns that off). */
int cd_engine_synchronize(cd_handle h);
/* per-launch timing of the implicit-GEMM kernel family with HIP events on the engine's stream
 * (bench.py roofline leg): enable, run, then collect launches / summed ms / summed 2*M*N*K flops */
int cd_prof_enable(cd_handle h, int on);
int cd_prof_collect(cd_handle h, int* launches, double* total_ms, double* total_flops);

/* networks: build from a descriptor, then load weights by the reference's state_dict names
 * (replaces load_model_from_config, model/lib/stable_diffusion/txt2img.py:25-42, and
 *  generator.load_state_dict, ddpm_ddim_wrapper.py:378-379) */
int cd_net_create(cd_handle h, const cd_net_desc* desc, int* net_id);
int cd_net_param_count(cd_handle h, int net, int* n);
int cd_net_param_info(cd_handle h, int net, int index, char* name, int name_cap, int* ndim, int64_t shape[4]);
int cd_net_load_param(cd_handle h, int net, const char* name, const float* data_host, int ndim,
                      const int64_t* shape);
int cd_net_missing_params(cd_handle h, int net, int* n_missing, char* first_name, int name_cap);

/* eps_hat = UNet(x, t, context)  — LatentDiffusion.apply_model -> DiffusionWrapper.forward ->
 * UNetModel.forward (ddpm.py:882-983,1392-1394; openaimodel.py:710-742) and DDPM.forward
 * (ddpm/diffusion.py:292-337). x [B,C,H,W]; t [B] float timesteps; ctx [B,L,Dc] or NULL;
 * eps_out [B,Cout,H,W]. */
int cd_unet_forward(cd_handle h, int net, const float* x, const float* t, const float* ctx, int B,
                    int ctx_len, float* eps_out);

/* c = FrozenCLIPEmbedder(text): last_hidden_state of the CLIP text transformer for already tokenised text
 * (modules.py:148-158: tokenizer(..., max_length=77, padding="max_length") then transformer(input_ids)).
 * tokens [B,L] int32 (device), out [B,L,width] fp32 - the `ctx` tensors of the sampler entry points.
 * For a BERT_XTR net: BERTEmbedder.forward (tokens -> transformer(tokens, return_embeddings=True)). */
int cd_text_encode(cd_handle h, int net, const int32_t* tokens, int B, int L, float* out);

/* DirectionalCLIP's feature extractors (model/energy/clean_clip.py:24-31): model.encode_text(tokens) and
 * model.encode_image(preprocessed) of OpenAI CLIP, un-normalised. tokens [B,L] int32; img [B,3,R,R] fp32 already
 * resized / centre-cropped / mean-std normalised; out [B,embed] fp32. */
int cd_clip_text_features(cd_handle h, int net, const int32_t* tokens, int B, int L, float* out);
int cd_clip_image_features(cd_handle h, int net, const float* img, int B, float* out);

/* z0 = scale * posterior(E(img)).sample() (or .mode() when sample==0) — encode_first_stage +
 * get_first_stage_encoding (ddpm.py:817-854, 536-543); img [B,3,R,R] in [-1,1]; noise [B,zc,R/8,R/8]
 * or NULL (then Philox(seed)); z0 [B,zc,R/8,R/8]. */
int cd_vae_encode(cd_handle h, int net, const float* img, const float* noise, uint64_t seed, int B, int R,
                  int sample, float scale, float* z0);
/* img = D(z0/scale)*out_mul + out_add — decode_first_stage (ddpm.py:698-755); the wrapper's
 * post_process (x+1)/2 (sd_wrapper:135-137) is out_mul=0.5, out_add=0.5. */
int cd_vae_decode(cd_handle h, int net, const float* z0, int B, int hlat, float scale, float out_mul,
                  float out_add, float* img);

/* DPM-Encoder: DDIMSampler.ddpm_ddim_encoding / _ddpm_ddim_encoding (ddim.py:230-286, 450-501)
 * and DDPMDDIMWrapper.encode's loop (ddpm_ddim_wrapper.py:483-520).
 *   x0 [B,C,H,W]; ctx_c / ctx_uc [B,L,Dc] or NULL (pixel DDPMs); guidance g as in ddim.py:550-559;
 *   coef_host: K+1 rows — row K initialises x_T (sa, s1a), rows K-1..0 are the loop steps;
 *   noise [K,B,C,H,W] (slot 0 = x_T draw, slots 1..K-1 = per-step draws in loop order) or NULL;
 *   z_out [B,K+1,C,H,W] = stack([x_T, eps_{K-1}, ..., eps_0], dim=1) (sd_wrapper:203).
 *   last_uses_x0: 1 = latent sampler (index 0 returns x0, no draw, ddim.py:583-584);
 *                 0 = pixel wrapper (K-1 ordinary steps; z has K entries).
 *   A white-box prefix shorter than the chain (`white_box_steps` of the text wrappers, ddim.py:486: the loop breaks after
 *   n < K steps) is the same call on the n + 1 table rows [K-n .. K-1, K] with last_uses_x0 = 0 and n + 1 noise slots - every
 *   row carries its own timestep; K = 0 draws x_T only (white_box_steps = -1). */
int cd_dpm_encode(cd_handle h, int net, int sched_kind, const float* x0, const float* ctx_c,
                  const float* ctx_uc, int ctx_len, float guidance, int B, int K,
                  const cd_step_coef* coef_host, const float* noise, uint64_t seed,
                  int last_uses_x0, float* z_out);

/* Decode with injected eps: DDIMSampler.sample_with_eps / ddim_sampling_with_eps /
 * p_sample_ddim_with_eps (ddim.py:170-228, 395-448, 603-646) and DDPMDDIMWrapper.generate
 * (ddpm_ddim_wrapper.py:392-455).
 *   z [B,T,C,H,W] with z[:,0] = x_T and z[:,1+i] the eps of loop step i; steps i >= n_eps draw fresh
 *   noise (`noise_tail` [K-n_eps,B,C,H,W] or Philox). coef_host: K rows in loop order index K-1..0
 *   stored at row index = k. x_out [B,C,H,W]. */
int cd_ddim_decode(cd_handle h, int net, int sched_kind, const float* z, int z_slots, int n_eps,
                   const float* ctx_c, const float* ctx_uc, int ctx_len, float guidance, int B, int K,
                   const cd_step_coef* coef_host, const float* noise_tail, uint64_t seed, float* x_out);

/* The same with one classifier-free-guidance scale PER SAMPLE (`guidance_per_sample`: device pointer, B floats, each
 * neither 0 nor 1): the ensemble loop of the text wrappers decodes every z at each of
 * `decoder_unconditional_guidance_scales` (stable_diffusion_stochastic_text_wrapper.py:155-166) - members that differ
 * only in that scale run as one batch. Per-sample arithmetic is that of cd_ddim_decode with the sample's scale. */
int cd_ddim_decode_v(cd_handle h, int net, int sched_kind, const float* z, int z_slots, int n_eps,
                     const float* ctx_c, const float* ctx_uc, int ctx_len, const float* guidance_per_sample, int B,
                     int K, const cd_step_coef* coef_host, const float* noise_tail, uint64_t seed, float* x_out);

/* The coupled source -> target loop in ONE call: what Model.forward composes from the wrapper's encode() and forward()
 * (model/text_unsupervised_translation.py:24-40: z = gan_wrapper.encode(image, encode_text); img = gan_wrapper(z, ...)) when
 * both run on the same network over the whole chain (white_box_steps = custom_steps + 1): the DPM-Encoder step and the decode
 * step of index k evaluate the network at the same timestep, and the decode step needs eps_k only after its forward, so each
 * of the K iterations runs ONE forward over [encoder rows | decoder rows], then the encoder's step kernel (ddim.py:582-601,
 * 545-580) and the decoder's (ddim.py:603-646), which reads the eps the encoder just wrote.
 *   x0 [B,C,H,W]; enc_ctx_* [B,L,Dc] and enc_guidance as cd_dpm_encode's; the decoder runs n_dec decodes per encoder sample
 *   (the wrapper's decoder_unconditional_guidance_scales of one kind, sd_wrapper:155-166): dec_ctx_* [n_dec*B,L,Dc], decoder
 *   row j*B + b decodes the z of encoder sample b; dec_guidance (scalar) or dec_guidance_per_sample (device, n_dec*B floats,
 *   each neither 0 nor 1) as cd_ddim_decode / cd_ddim_decode_v; coef_enc_host K+1 rows, coef_dec_host K rows (the same rows
 *   0..K-1); noise [K,B,C,H,W] or NULL and last_uses_x0 as cd_dpm_encode's.
 *   z_out [B,K+1,C,H,W] (what encode() returns), x_out [n_dec*B,C,H,W] (what the decode returns). */
int cd_cycle_translate(cd_handle h, int net, int sched_kind, const float* x0, const float* enc_ctx_c,
                       const float* enc_ctx_uc, float enc_guidance, const float* dec_ctx_c, const float* dec_ctx_uc,
                       float dec_guidance, const float* dec_guidance_per_sample, int ctx_len, int B, int n_dec, int K,
                       const cd_step_coef* coef_enc_host, const cd_step_coef* coef_dec_host, const float* noise,
                       uint64_t seed, int last_uses_x0, float* z_out, float* x_out);

/* Stochastic refinement (ddpm_ddim_wrapper.py:431-453): x_t = sa*x + s1a*n (row R of coef_host),
 * then R random-noise steps rows R-1..0. noise [R+1,B,C,H,W] or NULL. In/out x [B,C,H,W]. */
int cd_pix_refine(cd_handle h, int net, int sched_kind, float* x, int B, int R,
                  const cd_step_coef* coef_host, const float* noise, uint64_t seed);

/* ---- single-kernel entry points (parity tests call the HIP kernels through these) ----------- */
int cd_op_pack_conv_weight(cd_handle h, const float* w_host, int N, int Cin, int KH, int KW, int geglu,
                           void** packed_dev, int* Npad, int* Cpad);
int cd_op_free(cd_handle h, void* dev);
/* x: fp32 NCHW [B,C0(,+C1),H,W] (x1 optional second source); y: fp32 NCHW [B,N,Ho,Wo] */
int cd_op_conv2d(cd_handle h, const float* x0, int C0, const float* x1, int C1, int B, int H, int W,
                 const void* packed_w, int N, int KH, int KW, int stride, int pad, int asym_pad, int up,
                 const float* bias, const float* rowvec, const float* resid, int act, int tile, float* y);
/* the same convolution with the engine's 16-bit output (the product path's format; y still arrives as fp32 NCHW) and,
 * if `stats` is given, the fused GroupNorm statistics of the output: fp32 [B*Ho*Wo / 32][2][N] = per-channel sum |
 * sum of squares over each block of 32 rows. tile = 30 selects the streaming K = 320 linear kernel (lin_stream.hip);
 * act | 0x400 additionally LayerNorm-s the input rows inside that kernel (statistics only, eps 1e-5; >= 65536 rows). */
int cd_op_conv2d_16(cd_handle h, const float* x0, int C0, const float* x1, int C1, int B, int H, int W,
                    const void* packed_w, int N, int KH, int KW, int stride, int pad, int asym_pad, int up,
                    const float* bias, const float* rowvec, const float* resid, int act, int tile, float* y,
                    float* stats);
int cd_op_groupnorm(cd_handle h, const float* x, int B, int C, int H, int W, int G, float eps,
                    const float* gamma, const float* beta, const float* film, int silu, float* y);
int cd_op_layernorm(cd_handle h, const float* x, int rows, int C, const float* gamma, const float* beta,
                    float eps, float* y);
/* q [B,Tq,H*D], k,v [B,Tk,H*D] fp32 -> o [B,Tq,H*D]; use_transpose_kernel: 0 = V consumed token-major (the U-Net
   path: fused q|k|v projection, LDS transpose reads), 1 = V pre-transposed to [B,H,D,Tk_pad] first */
int cd_op_attention(cd_handle h, const float* q, const float* k, const float* v, int B, int H, int Tq,
                    int Tk, int D, float scale, int use_transpose_kernel, float* o);
int cd_op_softmax_rows(cd_handle h, const float* s, int64_t rows, int cols, float* p);
int cd_op_timestep_embedding(cd_handle h, const float* t, int B, int dim, int mode, float* out);
/* one scheduler step on explicit tensors (bit-exact checks): mode 0 init_xt, 1 encode, 2 decode */
int cd_op_sched_step(cd_handle h, int mode, int sched_kind, const cd_step_coef* coef_host, const float* x0,
                     float* xt, const float* eps_hat, int cfg, float guidance, const float* noise,
                     const float* eps_in, int is_last, int B, int C, int HW, float* z_slot);
/* micro-benchmark of one conv / GEMM shape on synthetic data (scripts/bench_gemm.py): average ms per launch.
 * act: low byte = activation; | 0x100 = in-place residual update of the output; | 0x200 = fused GroupNorm statistics */
int cd_op_bench_conv(cd_handle h, int B, int H, int W, int C0, int C1, int N, int k, int stride, int up,
                     int act, int tile, int iters, float* ms_out);
/* what the matrix cores of this device sustain on 16-bit operands under its power cap: a bare MFMA loop on every CU for
 * about target_ms (no reference counterpart: measurement support for bench.py's roofline object) */
int cd_op_bench_mfma_sustained(cd_handle h, int target_ms, float* tflops_out, float* ghz_out);
/* raw MFMA / LDS layout probe used by tests/test_gpu_ops.py */
int cd_op_probe(cd_handle h, int which, const void* in, void* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* CYCLEDIFF_H */
