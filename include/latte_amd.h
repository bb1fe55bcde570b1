/*
 * latte_amd.h — C-ABI of the MI355X-native Latte denoising engine (liblatte_amd.so).
 *
 * The reference (Vchitect/Latte) exposes its hot path through Python protocols, not a native
 * plugin ABI; each entry point below names the reference interface it stands
 * in for.  Conventions:
 *   - plain C symbols, plain pointers and sizes, no torch / C++ types;
 *   - tensors are CALLER-OWNED DEVICE pointers in the reference layouts (contiguous);
 *     the engine owns only its packed weights and workspace;
 *   - every compute call takes a hipStream_t (as void*), is stream-ordered, never synchronises;
 *   - return value 0 = ok, non-zero = error; latte_last_error() gives the thread-local message
 *     (the reference raises Python exceptions / asserts: latte.py:38, gaussian_diffusion.py:278,290);
 *   - one engine per (device, host thread); no internal threads.
 */
#ifndef LATTE_AMD_H_
#define LATTE_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LATTE_OK 0
#define LATTE_ERR_INVALID 1   /* bad argument / unsupported configuration */
#define LATTE_ERR_HIP 2       /* a HIP runtime call failed                 */
#define LATTE_ERR_STATE 3     /* e.g. weights missing, batch > max_batch   */

/* compute dtype of the MFMA operands (accumulation, residual stream, LN/softmax statistics and the
 * sampler update are always fp32) */
#define LATTE_DTYPE_BF16 0
#define LATTE_DTYPE_F16 1

const char* latte_last_error(void);
/* "latte_amd <version> gfx950 ..." build identification */
const char* latte_version(void);

/* ------------------------------------------------------------------ schedule (host, fp64)
 * Replaces diffusion/__init__.py:10-47 create_diffusion -> respace.py:12-62 space_timesteps,
 * respace.py:73-87 SpacedDiffusion.__init__, gaussian_diffusion.py:153-201 table construction.
 * Integer outputs are bit-exact with the reference; fp64 tables are computed in the same order
 * of operations. */
typedef struct latte_schedule latte_schedule_t;

int latte_schedule_create(int diffusion_steps, const char* timestep_respacing /* "", "250", "ddim50", "10,15,20" */,
                          const char* noise_schedule /* "linear" | "squaredcos_cap_v2" */,
                          latte_schedule_t** out);
/* What the model predicts, as create_diffusion derives it (diffusion/__init__.py:32-45): predict_xstart -> START_X
 * instead of EPSILON; learn_sigma -> LEARNED_RANGE (model output 2C channels), else FIXED_LARGE / FIXED_SMALL
 * (sigma_small) with C output channels (gaussian_diffusion.py:289-313,323-328).  Default: 0, 1, 0. */
int latte_schedule_set_model_types(latte_schedule_t* s, int predict_xstart, int learn_sigma, int sigma_small);
void latte_schedule_destroy(latte_schedule_t* s);
int latte_schedule_num_timesteps(const latte_schedule_t* s);
/* SpacedDiffusion.timestep_map (respace.py:75,86): respaced index -> original timestep */
int latte_schedule_timestep_map(const latte_schedule_t* s, int64_t* out, int n);
/* name in {betas, alphas_cumprod, alphas_cumprod_prev, sqrt_recip_alphas_cumprod,
 * sqrt_recipm1_alphas_cumprod, posterior_variance, posterior_log_variance_clipped,
 * posterior_mean_coef1, posterior_mean_coef2, log_betas, sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod} */
int latte_schedule_table(const latte_schedule_t* s, const char* name, double* out, int n);

/* ------------------------------------------------------------------ engine
 * Replaces models/latte.py:204-255 Latte.__init__ (one of the Latte_models presets, :464-506). */
typedef struct latte_engine latte_engine_t;

typedef struct latte_model_config {
  int input_size;    /* latent H = W (image_size / 8)          latte.py:210 */
  int patch_size;    /* 2 | 4 | 8                                latte.py:211 */
  int in_channels;   /* 4                                        latte.py:212 */
  int hidden_size;   /* multiple of 128                          latte.py:213 */
  int depth;         /* even: spatial/temporal alternate         latte.py:214,345 */
  int num_heads;     /* head_dim = hidden/heads in {64, 72}      latte.py:215 */
  int mlp_hidden;    /* int(hidden * mlp_ratio)                  latte.py:169 */
  int num_frames;    /*                                          latte.py:217 */
  int num_classes;   /* label table has num_classes+1 rows       latte.py:131 */
  int learn_sigma;   /* out_channels = 2*in_channels if set      latte.py:226 */
  int extras;        /* 1 = unconditional, 2 = class-conditional, 78 = text embedding (latte.py:221,235-242) */
  int compute_dtype; /* LATTE_DTYPE_* for the MFMA operands */
} latte_model_config_t;

int latte_engine_create(const latte_model_config_t* cfg, int max_batch, latte_engine_t** out);
void latte_engine_destroy(latte_engine_t* e);

/* Engine options: "gemm_variant" (0 auto; 1-3 simple kernel 128x128 / 256x128 / 256x256; 4-6 ping-pong kernel
 * 256x128 / 256x192 / 256x256 tiles; 7-9 the persistent ping-pong kernel, same tiles; 10 / 11 the 12-wave
 * producer / consumer kernel, 256x192, two-segment / rolling schedule; 12 / 13 the small-M kernel, 128x144 tile with a
 * four-stage ring, DMA by the twelve MFMA waves / by four extra waves -- what the gated GEMMs of a 4096-row batch run on),
 * "gemm_variant_qkv" / "_proj" / "_fc1" / "_fc2" (the same, for one of the four GEMMs of the block only; tuning hook),
 * "gated_split_k" (small batches: 0 = the rule -- a gated GEMM with >= 64 K tiles whose tiles fill at most half of the
 * CUs, and which the 128x144 tile does not take, runs as 2..4 partial products + one reduction into the residual stream; 1 = never; 2..4 = force that many),
 * "fuse_qkv_attn" (bit 0: spatial blocks, bit 1: temporal blocks run the QKV projection and the attention core of
 * latte.py:48-70 as ONE kernel with q / k / v held in LDS -- csrc/qkv_attn.hip -- wherever the shape allows it: 256 tokens per
 * frame / 16 frames, head_dim 64 | 72; default 3, 0 = the separate qkv GEMM + attention kernels; values 0..31: bits 2, 3, 4 switch
 * OFF one default schedule feature of the fused kernel each -- the next unit's first operand tile fetched under the attention
 * phase, the attention-phase issue priority of wave group 0, the four-heads-per-XCD unit order of 16-head models (A/B hooks);
 * every setting gives the same bits),
 * "guided_split" (bits; guided calls -- latte_forward_with_cfg and the guided sample loop -- only; default 20 on f16 engines, ignored
 * beyond bits 0 / 1 on bf16 ones: bit 0 = the attention output
 * that feeds the out-projection, bit 1 = the LayerNorm-modulate output that feeds fc1 are carried as SPLIT operand pairs [hi | lo] (two
 * halves per value) against weights stored [W | W], i.e. those two linears run on K' = 2 K without rounding their activation operand.
 * The guidance combination of latte.py:394-398 amplifies the operand rounding that differs between the two halves; with f16 operands
 * the XL/2 guided output at trained-scale gates sits AT 1e-3 of the fp32 reference (0.6 - 1.2e-3), with the split pairs at 0.4 - 0.8e-3
 * for +31 % of the guided step at XL/2.  Bits 2 / 3 (round 6) carry the same two operands as f16 + an FP8
 * remainder (e4m3 of lo * 2^12, one byte per value) whose product with an fp8 copy of the weight is collected by a block-scaled fp8
 * MFMA pass behind the f16 K loop of the SAME GEMM launch: the same parity margin at half the extra MFMA time and a quarter of the
 * extra operand bytes; a bit-2 / 3 setting wins over bit 0 / 1 for its operand.  Bit 4 (round 6) carries fc1's operand as f16 + an FP4
 * remainder (e2m1 codes, two per byte, ONE E8M0 scale per row) -- the block-scaled MFMA runs fp4 x fp4 at twice the fp8 rate; it wins
 * over bits 1 / 3.  12 = both remainders as fp8 (+15 % per guided forward), 20 = the default (+11 %), 3 = round 5's f16 pairs (+30 %), all
 * at the same parity.  0 = the plain f16 operands of the unguided path.
 * Operands whose shape has no split form stay plain: latte_engine_get_option("guided_split_active") reports the bits the last
 * guided forward really used),
 * "seed" (Philox seed of the engine's own noise stream, used by latte_sample_loop when no noise
 * pointer is supplied; the reference draws torch.randn_like, gaussian_diffusion.py:413,555). */
int latte_engine_set_option(latte_engine_t* e, const char* name, int64_t value);
/* Reads an option back (same names), or one of the read-only facts "guided_split_active" (bits of "guided_split" the last guided
 * forward ran with) and "guided_split_failed" (1 after an allocation for the split operands failed: guided calls then run plain). */
int latte_engine_get_option(const latte_engine_t* e, const char* name, int64_t* value);

/* Replaces nn.Module.load_state_dict (sample.py:62-64) for ONE tensor named by its reference
 * state_dict key (SURVEY.md §8(b) lists them: "blocks.3.attn.qkv.weight", "pos_embed", ...).
 * `data` is fp32, contiguous, reference shape; host pointer if on_device == 0, else device pointer.
 * The engine converts / packs into its own storage (caller may free `data` on return). */
int latte_engine_load_tensor(latte_engine_t* e, const char* key, const float* data, int64_t numel,
                             int on_device, void* stream);
/* Returns 0 when every tensor the configuration needs has been loaded; otherwise LATTE_ERR_STATE
 * and latte_last_error() names the first missing key (load_state_dict strict=True behaviour). */
int latte_engine_check_weights(latte_engine_t* e);
/* number of reference keys expected / name of the i-th one (for enumeration by the host shim) */
int latte_engine_num_keys(const latte_engine_t* e);
const char* latte_engine_key(const latte_engine_t* e, int i);

/* Timestep-embedding table of a schedule: out[i, :] = t_embedder(timestep_map[i]) for every respaced step i
 * (TimestepEmbedder, latte.py:84-123, through respace.py:125-130) -- [num_timesteps, hidden] fp32, device.  It depends on
 * nothing but the schedule and the weights, so the multi-GPU driver lets rank 0 compute it and broadcasts it over RCCL
 * (250 x 1152 x 4 B = 1.15 MB, the only payload collective of the sampling path); latte_engine_set_temb_table installs
 * a received table (copied) together with the schedule's timestep_map; latte_sample_loop uses it only for a schedule with
 * exactly that map (create_diffusion("250") and ("ddim250") both have 250 steps but different maps) and falls back to
 * computing its own otherwise.  Loading any t_embedder.* tensor uninstalls it.  table == NULL or s == NULL uninstalls. */
int latte_engine_temb_table(latte_engine_t* e, const latte_schedule_t* s, float* out, void* stream);
int latte_engine_set_temb_table(latte_engine_t* e, const latte_schedule_t* s, const float* table, void* stream);

/* Text-conditioned variant (extras == 78; latte.py:238-242,340-363): project a batch of text embeddings once,
 * text_embedding:[batch, 77*768] fp32 (device), Linear(SiLU(.)) -> [batch, D] kept inside the engine.  Every later
 * latte_forward / latte_forward_with_cfg / latte_sample_loop with the same batch conditions the 28 blocks on
 * t_emb + projected text and the final layer on t_emb alone (latte.py:372-373); y is ignored.  For guidance pass the
 * doubled batch [text, null text] (forward_with_cfg forwards text_embedding unchanged, latte.py:388). */
int latte_engine_set_text_embedding(latte_engine_t* e, const float* text_embedding, int batch, void* stream);

/* Latte.forward (latte.py:314-377).  x:[B,F,C,H,W] fp32, t: int64[B] ORIGINAL timesteps (device),
 * y: int64[B] labels (device) or NULL when extras == 1, out:[B,F,Cout,H,W] fp32. */
int latte_forward(latte_engine_t* e, const float* x, const int64_t* t, const int64_t* y, int batch,
                  float* out, void* stream);
/* Latte.forward_with_cfg (latte.py:379-398): x is the doubled batch [2b,...] whose first half is
 * duplicated internally; guidance on the first 4 channels; out:[2b,F,Cout,H,W]. */
int latte_forward_with_cfg(latte_engine_t* e, const float* x, const int64_t* t, const int64_t* y,
                           int batch /* = 2b */, float cfg_scale, float* out, void* stream);

/* ------------------------------------------------------------------ sampler update
 * Replaces gaussian_diffusion.py:254-336 p_mean_variance (EPSILON + LEARNED_RANGE, as built by
 * create_diffusion) followed by :380-421 p_sample (method 0 = "ddpm") or :517-564 ddim_sample
 * (method 1 = "ddim").  `index` is the respaced step i.  model_out:[B,F,2C,H,W]; noise may be NULL
 * when no noise term is needed (ddim eta == 0, or index == 0).  Outputs may alias x. */
#define LATTE_METHOD_DDPM 0
#define LATTE_METHOD_DDIM 1
int latte_sampler_step(const latte_schedule_t* s, int method, int index, float eta, int clip_denoised,
                       const float* x, const float* model_out, const float* noise,
                       int batch, int frames, int channels, int hw /* H*W */,
                       float* sample_out, float* pred_xstart_out /* may be NULL */, void* stream);
/* The same step with the two caller hooks of gaussian_diffusion.py:
 *   denoised_fn (process_xstart, :316-321): call once with predict_only = 1 to get the raw x_start prediction in
 *     pred_xstart_out, apply the function, and pass its result as pred_xstart_in (it replaces the prediction BEFORE the
 *     clamp);
 *   cond_fn (:345-375; the hook sees ORIGINAL timesteps, respace.py:100-104): pass cond_grad = cond_fn(x, t).  DDPM
 *     adds variance * gradient to the mean (condition_mean), DDIM shifts eps by sqrt(1 - alpha_bar) * gradient and
 *     re-derives pred_xstart from it (condition_score).
 * Either pointer may be NULL. */
int latte_sampler_step_ex(const latte_schedule_t* s, int method, int index, float eta, int clip_denoised,
                          const float* x, const float* model_out, const float* noise, const float* pred_xstart_in,
                          const float* cond_grad, int predict_only, int batch, int frames, int channels, int hw,
                          float* sample_out, float* pred_xstart_out, void* stream);

/* ------------------------------------------------------------------ fused sampling loop
 * Replaces gaussian_diffusion.py:423-515 p_sample_loop / :604-684 ddim_sample_loop driving the
 * model through respace.py:125-130 _WrappedModel (index -> original timestep), i.e. the body of
 * sample/sample.py:100-107.  Runs steps start_index .. end_index (inclusive, descending; the full
 * chain is num_timesteps-1 .. 0).  x:[B,F,C,H,W] fp32 updated in place.  cfg_scale > 1 selects
 * forward_with_cfg semantics (batch is then the doubled batch, sample.py:88-94).
 * noise: NULL, or [n_steps][B,F,C,H,W] with noise[k] consumed by the k-th executed step (parity
 * runs feed the reference's draws).  trail_sample / trail_x0: NULL or [n_steps][B,F,C,H,W] receiving
 * every step's {"sample","pred_xstart"} (the *_progressive generators, :468,:637). */
int latte_sample_loop(latte_engine_t* e, const latte_schedule_t* s, int method, float eta,
                      int clip_denoised, float cfg_scale, float* x, const int64_t* y, int batch,
                      int start_index, int end_index, const float* noise,
                      float* trail_sample, float* trail_x0, void* stream);
/* The same loop with the model callable named explicitly: guided != 0 drives Latte.forward_with_cfg (doubled batch, first
 * half duplicated, eps_u + cfg_scale * (eps_c - eps_u)) for ANY cfg_scale -- the reference's method has no threshold
 * (latte.py:379-398); sample.py:51 only decides which callable it passes.  latte_sample_loop = guided iff cfg_scale > 1. */
int latte_sample_loop_ex(latte_engine_t* e, const latte_schedule_t* s, int method, float eta, int clip_denoised, int guided,
                         float cfg_scale, float* x, const int64_t* y, int batch, int start_index, int end_index,
                         const float* noise, float* trail_sample, float* trail_x0, void* stream);

/* ------------------------------------------------------------------ training path, forward evaluation
 * SURVEY.md section 8(f) rank 3, first slice: the loss VALUES of GaussianDiffusion.training_losses
 * (gaussian_diffusion.py:719-795, through SpacedDiffusion.training_losses respace.py:95-98; called by train.py:224-226)
 * on device tensors, one timestep per sample -- no gradients yet (the backward kernels are the next slice).
 *   latte_q_sample          gaussian_diffusion.py:216-229   x_t = sqrt(ab[t]) x_0 + sqrt(1 - ab[t]) noise
 *   latte_training_losses   given x_0, x_t, noise and the model's output on (x_t, timestep_map[t]):
 *                           mse = mean_flat((target - prediction)^2) (target = noise | x_0, :776-785),
 *                           vb  = _vb_terms_bpd with the frozen mean (:686-717,:760-774; KL in bits, decoder NLL at t == 0),
 *                           loss by loss_type: 0 MSE (mse [+ vb when the variance is learned]), 1 RESCALED_MSE (vb *
 *                           num_timesteps / 1000), 2 KL (vb), 3 RESCALED_KL (vb * num_timesteps)  (diffusion/__init__.py:22-27)
 * t: int64[batch] RESPACED indices (device); tensors [batch, frames, channels, hw] fp32 (model_out with 2 * channels when the
 * schedule's variance is learned); mse_out / vb_out may be NULL; workspace: latte_training_workspace_floats() floats. */
int latte_q_sample(const latte_schedule_t* s, const float* x_start, const float* noise, const int64_t* t, int batch,
                   int64_t numel_per_sample, float* x_t, void* stream);
int64_t latte_training_workspace_floats(int batch, int64_t numel_per_sample);
int latte_training_losses(const latte_schedule_t* s, int loss_type, const float* x_start, const float* x_t, const float* noise,
                          const float* model_out, const int64_t* t, int batch, int frames, int channels, int hw, float* workspace,
                          int64_t workspace_floats, float* mse_out, float* vb_out, float* loss_out, void* stream);

/* ------------------------------------------------------------------ training step
 * Replaces the body of the reference's optimisation loop, train.py:197-236, for the class-conditional / unconditional Latte
 * (train.py:213-218 refuses text-to-video training):
 *     x_t = q_sample(x_0, t); terms = diffusion.training_losses(model, x_0, t, dict(y=y)); loss = terms["loss"].mean()
 *     loss.backward(); clip_grad_norm_(...); opt.step() [torch.optim.AdamW(lr, weight_decay=0), train.py:127]; update_ema(...)
 * on device buffers.  Parameters, gradients, AdamW moments and the EMA live in caller-owned FLAT fp32 device buffers of
 * latte_trainer_total_numel() floats; tensor i (reference named_parameters() order, latte.py:226-255, without the frozen
 * pos_embed / temp_embed tables, which latte_trainer_set_frozen supplies) occupies [offset_i, offset_i + numel_i) in reference
 * shape.  The data-parallel driver (train.py:125 DDP) all-reduces the gradient buffer between latte_trainer_forward_backward
 * and latte_trainer_optimizer_step; nothing here communicates.  MFMA operands are half copies (cfg->compute_dtype) of the fp32
 * masters, fp32 accumulation, fp32 residual stream / LayerNorm / softmax statistics / loss. */
typedef struct latte_trainer latte_trainer_t;
int latte_trainer_create(const latte_model_config_t* cfg, int max_batch, latte_trainer_t** out);
void latte_trainer_destroy(latte_trainer_t* e);
int latte_trainer_num_params(const latte_trainer_t* e);
const char* latte_trainer_param_key(const latte_trainer_t* e, int i);
int64_t latte_trainer_param_offset(const latte_trainer_t* e, int i);
int64_t latte_trainer_param_numel(const latte_trainer_t* e, int i);
int64_t latte_trainer_total_numel(const latte_trainer_t* e);
/* device pointers; ema may be NULL (no EMA update) */
int latte_trainer_bind(latte_trainer_t* e, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* ema);
int latte_trainer_set_frozen(latte_trainer_t* e, const float* pos_embed, const float* temp_embed, int on_device, void* stream);
/* re-packs the half operand copies from the fp32 masters (after the caller changed the parameter buffer, e.g. load_state_dict) */
int latte_trainer_sync_weights(latte_trainer_t* e, void* stream);
/* One micro-batch: forward with saved activations, loss terms, backward; the gradient of terms["loss"].mean() is ASSIGNED to the
 * bound gradient buffer (label-table rows are accumulated: the optimiser step leaves the buffer zeroed).  x_start / noise
 * [batch, F, C, H, W] fp32, t int64[batch] RESPACED indices, y int64[batch] labels AFTER the caller's label dropout
 * (LabelEmbedder.token_drop, latte.py:138-153: dropped labels = num_classes) or NULL, loss_type 0 MSE | 1 RESCALED_MSE,
 * terms_out device float [3][batch] = loss, mse, vb; model_out_copy optional [batch, F, C_out, H, W]. */
int latte_trainer_forward_backward(latte_trainer_t* e, const latte_schedule_t* s, int loss_type, const float* x_start, const float* noise,
                                   const int64_t* t, const int64_t* y, int batch, float* terms_out, float* model_out_copy, void* stream);
/* The same micro-batch in pieces, for a data-parallel driver that overlaps its gradient all-reduce with the backward (what
 * DistributedDataParallel's buckets do for train.py:125): latte_trainer_begin = forward + loss terms + d loss / d model_output;
 * then latte_trainer_backward_stage(k) for k = 0 .. latte_trainer_num_stages() - 1 IN ORDER (0 = final layer, 1 .. depth =
 * blocks depth-1 .. 0, depth + 1 = patch embed and the t / y embedders).  After stage k the contiguous slice
 * latte_trainer_stage_range(k) of the gradient buffer is final: the driver enqueues its collective on that slice (28 MB per
 * Latte-B/2 block, 96 MB per XL/2 block: large buckets, as the per-link-bound xGMI ring wants) while the next stages run.
 * `y` must stay alive until the last stage. */
int latte_trainer_begin(latte_trainer_t* e, const latte_schedule_t* s, int loss_type, const float* x_start, const float* noise,
                        const int64_t* t, const int64_t* y, int batch, float* terms_out, float* model_out_copy, void* stream);
int latte_trainer_num_stages(const latte_trainer_t* e);
int latte_trainer_stage_range(const latte_trainer_t* e, int stage, int64_t* offset, int64_t* numel);
int latte_trainer_backward_stage(latte_trainer_t* e, int stage, void* stream);
/* clip_grad_norm_ (utils.py:72-117: total 2-norm, g *= min(max_norm / (norm + 1e-6), 1) when clip != 0) + AdamW + update_ema
 * (utils.py:191-200) on the bound buffers.  `step` = 0 (what LatteTrainer passes): AdamW's bias correction uses the trainer's own count
 * of APPLIED updates (what torch.optim.AdamW does: a fresh optimiser state starts at 1 even when the training-step counter continues
 * from a checkpoint, train.py:195-196 -- the count that drives clipping and logging stays with the caller; a skipped update does not
 * advance it).  `step` >= 1: the caller supplies the bias-correction step -- valid only while NO update is ever skipped (the caller cannot
 * see a skip without latte_trainer_scaler_state; after one, a caller-side count runs ahead of the moments' age, which differs from
 * GradScaler + torch.optim.AdamW): with loss scaling on, pass 0.  A NON-FINITE gradient norm
 * (overflow of the loss-scaled f16 backward or of an f16 activation; impossible in the reference's fp32 range) skips the update:
 * parameters, moments and EMA stay untouched, the gradients are zeroed, norm_out reports the non-finite norm with coefficient 0.
 * norm_out: optional device float[2] = {norm, applied coefficient} */
/* "loss_scale" (a power of two in [1, 2^24]): d loss / d model_output is multiplied by it before the backward and every finished
 * gradient slice by its inverse, so that the half-precision gradient operands stay inside the operand type's range.  Default: 1 with
 * bf16 operands, 16384 with f16 operands -- f16's 10 mantissa bits are the precision class of the TF32 matmuls the reference trains
 * with (train.py:12-14 allow_tf32), bf16 has 7; the reference itself needs no scaling because it keeps fp32 ranges.
 * "dynamic_loss_scale" (0 / 1; default 1 with f16, 0 with bf16): a skipped update halves the scale (floor 1), and
 * "loss_scale_growth_interval" (default 2000) applied updates in a row double it (cap max(initial scale, 2^16)) -- all on the
 * device, no host round trip.  Setting "loss_scale" restarts from that value.
 * "fuse_gelu" (default 1): the MLP's GELU passes inside the fc1 forward / fc2 input-gradient GEMM epilogues.
 * "fuse_small" (default 1, round 6b): the step's tiny launches folded (csrc/train_fin.hip) -- one finalize launch per block
 * stage, the gated residual's backward on the LayerNorm backward's pass, gated add + next LayerNorm in one forward pass, fc1 / qkv
 * bias gradients on the weight-gradient launch, one weight-pack launch; 0 = the separate launches (A/B tests); refused while a
 * step is in flight (between latte_trainer_begin and the last backward stage).  Gradients agree with the separate path to 2e-4. */
int latte_trainer_set_option(latte_trainer_t* e, const char* name, double value);
/* out8 (host) = {loss scale, applied updates since it changed, applied updates in total, skipped updates, last call skipped (0/1),
 * dynamic (0/1), growth interval, largest scale}.  Synchronises the device: for logging and tests, not for the step path. */
int latte_trainer_scaler_state(latte_trainer_t* e, double* out8);
int latte_trainer_optimizer_step(latte_trainer_t* e, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                 float clip_max_norm, int clip, float ema_decay, float* norm_out, void* stream);

/* ------------------------------------------------------------------ VAE decoder
 * Replaces diffusers.AutoencoderKL (sample/sample.py:69 from_pretrained, :113-115 vae.decode(z / 0.18215).sample;
 * sample_ddp.py:90,165-168) for the stabilityai/sd-vae-ft-* architecture: latent_channels 4,
 * block_out_channels (128, 256, 512, 512), layers_per_block 2, norm_num_groups 32, SiLU.  diffusers is not vendored
 * in the reference: the restated algorithm and its parity status are in oracle/vae_oracle.py. */
typedef struct latte_vae latte_vae_t;
/* latent_size: H = W of the latent (multiple of 16); max_frames: largest N of one decode call; compute_dtype: LATTE_DTYPE_F16
 * only (MFMA operands f16 as in the reference's fp16 decode, residual stream fp32) */
int latte_vae_create(int latent_size, int max_frames, int compute_dtype, latte_vae_t** out);
/* diffusers.AutoencoderKLTemporalDecoder (sample/sample_t2x.py:31-32, pipeline_latte.py:779-798: vae.decode(z[i : i + 14],
 * num_frames=n).sample): the same handle type and entry points; every resnet is a SpatioTemporalResBlock (keys
 * "...resnets.N.spatial_res_block.*", "...temporal_res_block.*" with Conv3d weights [C, C, 3, 1, 1], "...time_mixer.mix_factor"),
 * there is no post_quant_conv, and "decoder.time_conv_out.*" follows conv_out.  ONE latte_vae_decode call decodes ONE chunk:
 * its n_frames frames are the temporal extent (max_frames >= the chunk length).  Restated from memory of diffusers 0.24.0:
 * parity unpinned (oracle/vae_temporal_oracle.py). */
int latte_vae_create_temporal(int latent_size, int max_frames, int compute_dtype, latte_vae_t** out);
void latte_vae_destroy(latte_vae_t* v);
/* load_state_dict for ONE tensor named by its diffusers key ("decoder.up_blocks.2.resnets.0.conv1.weight",
 * "post_quant_conv.bias", ...); fp32, reference shape; encoder.* / quant_conv.* keys are not accepted. */
int latte_vae_num_keys(const latte_vae_t* v);
const char* latte_vae_key(const latte_vae_t* v, int i);
int latte_vae_load_tensor(latte_vae_t* v, const char* key, const float* data, int64_t numel, int on_device,
                          void* stream);
int latte_vae_check_weights(latte_vae_t* v);
/* vae.decode(z * z_scale).sample: z [N,4,h,w] fp32 (device, the layout sample.py:112 produces), z_scale = 1/0.18215
 * folds the caller's division.  out_mode 0: fp32 [N,3,8h,8w] (the reference's `.sample`); out_mode 1: uint8
 * [N,8h,8w,3] = ((x*0.5+0.5)*255+0.5).clamp(0,255) of sample.py:122 fused into the last convolution. */
int latte_vae_decode(latte_vae_t* v, const float* z, int n_frames, float z_scale, int out_mode, void* out,
                     void* stream);
/* Measurement hook beside latte_profile_forward: ONE decode with a HIP event behind every launch.  ms_out[c] / launches_out[c]
 * (n >= 5 entries) per kernel class c: 0 = conv3x3 (implicit GEMM on the MFMA pipe), 1 = GroupNorm statistics, 2 = GroupNorm
 * apply (+ SiLU), 3 = mid-block attention and the 1x1 shortcut GEMMs, 4 = the small kernels (post_quant_conv, conv_in, conv_out,
 * fp32 -> half copies).  Synchronises the stream; bench.py's `vae_decode.roofline_table` is built from it. */
int latte_vae_profile_decode(latte_vae_t* v, const float* z, int n_frames, float z_scale, int out_mode, void* out, float* ms_out,
                             int* launches_out, int n, void* stream);

/* ------------------------------------------------------------------ LatteT2V denoiser (Latte-1 text-to-video)
 * LatteT2V.forward, /root/reference/models/latte_t2v.py:677-941 (constructor :475-672).
 * PixArt-alpha style blocks with norm_type "ada_norm_single", attention_bias = True, activation_fn "gelu-approximate",
 * norm_elementwise_affine = False, norm_eps 1e-6 -- the configuration Latte-1 ships; weights by the reference's
 * state-dict keys (to_q / to_k / to_v are packed into one fused GEMM on load).  The diffusers 0.24.0 leaves the reference
 * imports are restated, not vendored: parity is pinned against the reference's own file only (oracle/latte_t2v_oracle.py). */
typedef struct latte_t2v latte_t2v_t;
typedef struct {
  int num_attention_heads, attention_head_dim;   /* inner dim = heads * head_dim (16 x 72 = 1152) */
  int in_channels, out_channels;                 /* 4, 8 (learned sigma) */
  int num_layers;                                /* 28 spatial + 28 temporal blocks */
  int sample_size, patch_size;                   /* latent side (64 for 512 px), 2 */
  int cross_attention_dim, caption_channels;     /* 1152 (= inner dim), 4096 (T5-XXL features) */
  int video_length;                              /* frames, 16 */
  int max_text_tokens;                           /* workspace for the text tokens of one sample (120) */
  int compute_dtype;                             /* LATTE_DTYPE_BF16 | LATTE_DTYPE_F16 */
} latte_t2v_config_t;
int latte_t2v_create(const latte_t2v_config_t* cfg, int max_batch, latte_t2v_t** out);
void latte_t2v_destroy(latte_t2v_t* e);
int latte_t2v_num_keys(const latte_t2v_t* e);
const char* latte_t2v_key(const latte_t2v_t* e, int i);
int latte_t2v_load_tensor(latte_t2v_t* e, const char* key, const float* data, int64_t numel, int on_device, void* stream);
int latte_t2v_check_weights(latte_t2v_t* e);
/* "fuse_qkv_attn": as latte_engine_set_option (the to_q | to_k | to_v projection of attn1 and its attention core as one kernel
 * wherever the sequences are 256 tokens of a frame / 16 frames of a token; default 3; every setting gives the same bits). */
int latte_t2v_set_option(latte_t2v_t* e, const char* name, int64_t value);
/* x:[B,C,F,H,W] fp32 (channels before frames, latte_t2v.py:729), t: int64[B], encoder_hidden_states:[B,n_text,caption_channels]
 * fp32, encoder_attention_mask:[B,n_text] fp32 1 = keep / 0 = padded, or NULL; out:[B,out_channels,F,H,W] fp32.  All device
 * pointers.  enable_temporal_attentions = 0 skips the temporal blocks (latte_t2v.py:867). */
int latte_t2v_forward(latte_t2v_t* e, const float* x, const int64_t* t, const float* encoder_hidden_states,
                      const float* encoder_attention_mask, int batch, int n_text, int enable_temporal_attentions, float* out,
                      void* stream);

/* Chain-level text context: the caption projection, the cross-attention K|V of every spatial block and the mask bias depend on
 * the text only; latte_t2v_set_text computes them once ([batch, n_text, caption_channels] fp32, mask [batch, n_text] or NULL)
 * and latte_t2v_forward with encoder_hidden_states == NULL / latte_t2v_guided_ddim_loop reuse them for every step.  Loading
 * a tensor uninstalls the context; encoder_hidden_states == NULL here uninstalls it too. */
int latte_t2v_set_text(latte_t2v_t* e, const float* encoder_hidden_states, const float* encoder_attention_mask, int batch,
                       int n_text, void* stream);
/* The denoising loop of LattePipeline.__call__ (sample/pipeline_latte.py:700-760) for the classifier-free-guidance case with
 * a DDIM scheduler at eta = 0, entirely inside the engine: per step the transformer on the guidance pair (the latents are
 * duplicated inside, :725), noise_pred = uncond + s (text - uncond) (:748-749), the learned-sigma half dropped (:752-753),
 * and DDIMScheduler.step (x0 = (x - sqrt(1 - a_t) eps) / sqrt(a_t); x' = sqrt(a_prev) x0 + sqrt(1 - a_prev) eps).
 * x: [samples, C, F, H, W] fp32, updated in place; the installed text context must hold 2 * samples rows ordered
 * [negative prompts | prompts] (:647).  timesteps / alpha_t / alpha_prev: HOST arrays of n_steps entries (the scheduler's
 * timesteps and alphas_cumprod at t and at the previous timestep, final_alpha_cumprod for the last step). */
int latte_t2v_guided_ddim_loop(latte_t2v_t* e, float* x, int samples, int n_steps, const int64_t* timesteps,
                               const double* alpha_t, const double* alpha_prev, float guidance_scale,
                               int enable_temporal_attentions, void* stream);

/* ------------------------------------------------------------------ measurement hooks (bench.py)
 * Runs ONE denoiser forward eagerly with HIP events around every kernel launch on `stream`,
 * synchronises, and reports per-kernel-class totals.  classes (fixed order):
 *   0 gemm_qkv 1 gemm_proj 2 gemm_fc1 3 gemm_fc2 4 attn_spatial 5 attn_temporal 6 ln_modulate
 *   7 embed_cond 8 patch_embed 9 final_layer 10 qkv_attn_spatial 11 qkv_attn_temporal (the fused QKV projection +
 *   attention kernel of csrc/qkv_attn.hip, which replaces classes 0 + 4 / 0 + 5 in the blocks whose shape allows it)
 * ms_out / launches_out: arrays of n (>= 12). */
#define LATTE_NUM_KERNEL_CLASSES 12
int latte_profile_forward(latte_engine_t* e, const float* x, const int64_t* t, const int64_t* y, int batch,
                          float* out, float* ms_out, int* launches_out, int n, void* stream);
/* The same with guided != 0: the denoiser call of latte_forward_with_cfg (x = the first half of the batch, used for both halves;
 * split operands per the "guided_split" option), WITHOUT the guidance combination -- so the table describes the kernels a guided
 * sampling step really runs (bench.py: config3.roofline_table). */
int latte_profile_forward_ex(latte_engine_t* e, const float* x, const int64_t* t, const int64_t* y, int batch, int guided,
                             float* out, float* ms_out, int* launches_out, int n, void* stream);
/* Stand-alone timing of the dominant kernel: C[M,N] = A[M,K] * W[N,K]^T (+bias epilogue `epi`,
 * 0 = bf16 out, 1 = bias+GELU, 2 = gated residual) on synthetic operands already resident in HBM;
 * `iters` back-to-back launches between two HIP events on `stream`.  variant selects the tile
 * configuration (0 = engine default for the shape). */
int latte_bench_gemm(int M, int N, int K, int epi, int dtype, int variant, int iters, float* ms_per_launch,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LATTE_AMD_H_ */
