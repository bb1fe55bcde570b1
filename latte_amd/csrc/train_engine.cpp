// Training step on the engine (round 2): forward with saved activations, backward, AdamW + EMA.
//
//   train.py:197-236       x_t = q_sample(x_0, t), loss = training_losses(model, x_0, t)["loss"].mean(); loss.backward();
//                          clip_grad_norm_; opt.step(); update_ema
//   gaussian_diffusion.py:719-795   MSE on the mean head, variational bound on the variance head (mean detached)
//   latte.py:177-181,314-377        the block and the forward whose backward this is
//
// Layout: the same canonical token order as the inference engine (row = (b F + f) T + t); parameters, gradients and the
// optimiser state live in caller-owned FLAT fp32 buffers (reference named_parameters() order without the two frozen sin-cos
// tables; the host shim makes the nn.Parameters views of the parameter buffer and all-reduces the gradient buffer over RCCL).
// MFMA operands are half copies of the masters ([N, K] for the forward / weight-gradient GEMMs, [K, N] for the
// input-gradient GEMMs), re-packed after every optimiser step.  Saved per block: the two fp32 residual inputs of its LayerNorms
// and the half tensors xn1, qkv, attention output, y1, xn2, u (pre-GELU), h, y2.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "common.h"

using namespace latte;

namespace {
struct ParamInfo {
  std::string key;
  int64_t offset, numel;
};
struct BlockBuf {
  half_t *xn1, *qkv, *att, *y1, *xn2, *u, *h, *y2;
  half_t *qkv_w, *qkv_wt, *proj_w, *proj_wt, *fc1_w, *fc1_wt, *fc2_w, *fc2_wt;
};
}  // namespace

struct latte_trainer {
  latte_model_config_t cfg;
  int max_batch = 0;
  int D = 0, T = 0, F = 0, G = 0, Cin = 0, Cout = 0, H = 0, P = 0, KPE = 0, Hm = 0, hd = 0, nmod = 0, dt = 0;
  int64_t rows_max = 0, rows_pad = 0, ld = 0;
  // Loss scaling (f16 operands): d loss / d model_output is multiplied by loss_scale before the backward and every finished slice
  // of the gradient buffer by 1 / loss_scale -- the backward is linear in that seed, the scale is a power of two, so the result is
  // the unscaled gradient exactly unless something leaves f16's range.  Per-token gradients of this model are 1e-7 ... 1e-4
  // (Latte-B/2, batch 5): x 2^14 puts them at 1.6e-3 ... 1.6, inside f16's normal range (6e-5 ... 65504).  bf16: 1 (fp32's range).
  // Round 4: the scale lives in DEVICE memory (`scaler`, float[8], layout at gradnorm_finalize_kernel in train.hip) so that the
  // optimiser step can react to an overflow without a host round trip: a non-finite gradient norm SKIPS the update (parameters,
  // AdamW moments, EMA untouched) and, with dynamic scaling (the f16 default), halves the scale; 2000 applied updates in a row double
  // it again (cap 2^16 -- the largest of the three scales validated bit-exact in tests/test_training_step.py).  The same array
  // counts the APPLIED updates: AdamW's bias-correction step is that count (a fresh optimiser state starts at 1 whatever the
  // training-step counter of a continued run says; torch.optim.AdamW keeps its own count too).
  float loss_scale = 1.0f;       // initial / static value (host copy; the live value is scaler[0])
  int dynamic_scale = 0;
  int fuse_gelu = 1;        // the MLP's GELU passes inside the fc1 forward / fc2 input-gradient GEMMs (round 6; "fuse_gelu" option, A/B tests)
  // round 6b ("fuse_small" option, default 1; train_fin.hip): one finalize launch per block stage instead of ~25 tiny ones -- the
  // row-run / column partials of a stage stay in their own buffers until its end, the adaLN linear's input gradient is one batched
  // product at the last stage, the loss-scale pass of the block slices rides on the kernels that write them, one weight-pack launch
  int fuse_small = 1;
  float *pg1 = nullptr, *pl1 = nullptr, *pg2 = nullptr, *pg2b = nullptr, *pl2 = nullptr, *pc_fc1 = nullptr, *pc_qkv = nullptr, *dc_ws = nullptr, *no_ws = nullptr;
  PackDesc* pack_descs = nullptr;
  PackPlan pack_plan{};
  float growth_interval = 2000.0f;
  float* scaler = nullptr;
  std::vector<ParamInfo> params;
  std::map<std::string, int> index;
  int64_t total = 0;
  float *Pm = nullptr, *Gr = nullptr, *M1 = nullptr, *V2 = nullptr, *Ema = nullptr;   // bound flat buffers (caller-owned)
  float *pos = nullptr, *temp = nullptr, *pe_wt = nullptr, *fin_wt = nullptr;
  std::vector<float*> xs;        // 2 depth + 1 residual-stream snapshots, fp32 [rows_pad, D]
  std::vector<BlockBuf> blk;
  // conditioning
  float *tfreq = nullptr, *temb_pre = nullptr, *temb_act = nullptr, *cvec = nullptr, *csilu = nullptr, *mod = nullptr, *dmod = nullptr,
        *dc = nullptr, *dtmp = nullptr, *dtmp2 = nullptr;
  int64_t* t_orig = nullptr;
  int64_t* tmap_dev = nullptr;
  int tmap_n = 0;
  const latte_schedule_t* tmap_of = nullptr;
  // scratch
  float *x_t = nullptr, *model_out = nullptr, *dmodel_out = nullptr, *dtok = nullptr, *f32a = nullptr, *f32b = nullptr, *dx = nullptr,
        *pix = nullptr, *ones = nullptr, *zeros = nullptr, *part_rows = nullptr, *part_cols = nullptr, *wg_ws = nullptr, *ng_ws = nullptr,
        *attn_stats = nullptr, *loss_ws = nullptr, *stats = nullptr;
  double* sumsq = nullptr;
  half_t *dyD = nullptr, *dhH = nullptr, *dxnH = nullptr, *dqkvH = nullptr, *xnh = nullptr;
  int64_t wg_ws_floats = 0, ng_ws_floats = 0, loss_ws_floats = 0;
  bool weights_synced = false;
  int cur_batch = 0, next_stage = 1 << 30;   // step in flight: batch, labels (caller keeps them alive), next backward stage
  const int64_t* cur_y = nullptr;
  std::vector<void*> allocs;
};

namespace {

template <typename Tp>
int talloc(latte_trainer* e, Tp** p, size_t count, bool zero = true) {
  void* q = nullptr;
  const size_t bytes = std::max<size_t>(count * sizeof(Tp), 16);
  LATTE_HIP(hipMalloc(&q, bytes));
  if (zero) LATTE_HIP(hipMemset(q, 0, bytes));
  e->allocs.push_back(q);
  *p = (Tp*)q;
  return LATTE_OK;
}

void add_param(latte_trainer* e, const std::string& key, int64_t numel) {
  ParamInfo p{key, e->total, numel};
  e->index[key] = (int)e->params.size();
  e->params.push_back(p);
  e->total += (numel + 3) / 4 * 4;   // 16-byte aligned tensors inside the flat buffers
}

float* P_(latte_trainer* e, const std::string& key) { return e->Pm + e->params[e->index.at(key)].offset; }
float* G_(latte_trainer* e, const std::string& key) { return e->Gr + e->params[e->index.at(key)].offset; }

int gemm_half(latte_trainer* e, const half_t* A, const half_t* W, const float* bias, half_t* out, int M, int N, int K, hipStream_t st) {
  GemmArgs g{};
  g.A = A; g.W = W; g.bias = bias; g.out = out; g.M = M; g.N = N; g.K = K; g.rows_per_sample = M; g.gate_stride = 0;
  return launch_gemm(g, EPI_BIAS_H16, e->dt, 0, st);
}

// The rolling 12-wave kernel's training epilogues (gemm_pw.hip: EPI_BIAS_GELU_DUAL_H16 / EPI_DGELU_H16) take the shape
bool gelu_fusable(const latte_trainer* e, int M, int N, int K) {
  return e->fuse_gelu && N % 192 == 0 && K % 64 == 0 && K >= 128 && (uint64_t)((M + 255) / 256 * 256) * K * 2 < (1ull << 32) &&
         (uint64_t)N * K * 2 < (1ull << 32);
}
// out = A W^T + bias with the GELU pass fused: epi EPI_BIAS_GELU_DUAL_H16 (out = u, aux = gelu(u)) or EPI_DGELU_H16 (out = (A W^T) gelu'(aux))
int gemm_gelu(latte_trainer* e, int epi, const half_t* A, const half_t* W, const float* bias, half_t* out, half_t* aux, int M, int N, int K,
              hipStream_t st) {
  GemmArgs g{};
  g.A = A; g.W = W; g.bias = bias; g.out = out; g.aux = aux; g.M = M; g.N = N; g.K = K; g.rows_per_sample = M; g.gate_stride = 0;
  return launch_gemm_pw(g, epi, e->dt, 1, st);
}

// dW[N, K] = dY[M, N]^T X[M, K] on the transposed-operand GEMM (gemm_tn.hip: no transposed copies), the contraction split so
// that about four workgroups per CU are in flight; the partial products are reduced in a fixed order; result ASSIGNED to dW
// (unscale: the reduction also takes the result out of the loss-scaled domain, x 1 / scaler[0])
// cs_partial / cs_rows: the column sums of dY per split ([*cs_rows][N]: the bias gradient's partial rows) ride on the launch
int wgrad(latte_trainer* e, const half_t* dY, const half_t* X, int M, int N, int K, float* dW, hipStream_t st, bool unscale = false,
          float* cs_partial = nullptr, int* cs_rows = nullptr) {
  int rc, chunk = 0;
  const int splits = gemm_tn_plan(M, N, K, &chunk);
  if ((int64_t)splits * N * K > e->wg_ws_floats) return fail(LATTE_ERR_STATE, "wgrad: workspace too small");
  if (cs_rows) *cs_rows = splits;
  if ((rc = launch_gemm_tn(dY, X, e->wg_ws, M, N, K, chunk, e->dt, st, cs_partial))) return rc;
  return launch_split_reduce(e->wg_ws, splits, (size_t)N * K, (size_t)N * K, dW, 0, st, unscale ? e->scaler : nullptr);
}

// loss-scale state -> device.  what: 0 = everything incl. the counters (create), 1 = a new scale (restarts the growth count),
// 2 = the policy fields only (dynamic on / off, growth interval): the live scale and the counters stay.  Synchronous (an option
// call, not on the step path).
int upload_scaler(latte_trainer* e, int what) {
  float h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (what != 0) {
    LATTE_HIP(hipDeviceSynchronize());
    LATTE_HIP(hipMemcpy(h, e->scaler, sizeof(h), hipMemcpyDeviceToHost));
  }
  if (what != 2) {
    h[0] = e->loss_scale;
    h[1] = 0.0f;
  }
  h[5] = (float)e->dynamic_scale;
  h[6] = e->growth_interval;
  h[7] = std::max(std::max(e->loss_scale, h[0]), 65536.0f);
  LATTE_HIP(hipMemcpy(e->scaler, h, sizeof(h), hipMemcpyHostToDevice));
  return LATTE_OK;
}
bool scaling_active(const latte_trainer* e) { return e->loss_scale != 1.0f || e->dynamic_scale; }

}  // namespace

extern "C" {

int latte_trainer_create(const latte_model_config_t* cfg, int max_batch, latte_trainer_t** out) {
  if (!cfg || !out || max_batch <= 0) return fail(LATTE_ERR_INVALID, "trainer_create: bad arguments");
  const auto& c = *cfg;
  if (c.extras != 1 && c.extras != 2) return fail(LATTE_ERR_INVALID, "trainer: extras must be 1 or 2 (train.py:213-218 refuses text-to-video training)");
  if (c.hidden_size % 128 || c.hidden_size > 1280 || c.depth % 2 || c.hidden_size % c.num_heads)
    return fail(LATTE_ERR_INVALID, "trainer: hidden_size must be a multiple of 128 (<= 1280), depth even");
  const int hd = c.hidden_size / c.num_heads;
  if (hd != 64 && hd != 72) return fail(LATTE_ERR_INVALID, "trainer: head dim must be 64 or 72");
  if (c.mlp_hidden % 128) return fail(LATTE_ERR_INVALID, "trainer: mlp_hidden must be a multiple of 128");
  if (c.input_size % c.patch_size) return fail(LATTE_ERR_INVALID, "trainer: input_size % patch_size != 0");
  auto* e = new latte_trainer();
  e->cfg = c;
  e->max_batch = max_batch;
  e->D = c.hidden_size; e->F = c.num_frames; e->G = c.input_size / c.patch_size; e->T = e->G * e->G;
  e->Cin = c.in_channels; e->Cout = c.learn_sigma ? 2 * c.in_channels : c.in_channels; e->H = c.input_size;
  e->P = c.patch_size * c.patch_size * e->Cout; e->KPE = c.in_channels * c.patch_size * c.patch_size;
  e->Hm = c.mlp_hidden; e->hd = hd; e->dt = c.compute_dtype;
  e->loss_scale = c.compute_dtype == LATTE_DTYPE_F16 ? 16384.0f : 1.0f;
  e->dynamic_scale = c.compute_dtype == LATTE_DTYPE_F16 ? 1 : 0;
  e->nmod = c.depth * 6 * e->D + 2 * e->D;
  if ((e->F * e->T) % 64) { delete e; return fail(LATTE_ERR_INVALID, "trainer: frames * tokens per sample must be a multiple of 64"); }
  e->rows_max = (int64_t)max_batch * e->F * e->T;
  e->rows_pad = (e->rows_max + 255) / 256 * 256;
  e->ld = (e->rows_max + 63) / 64 * 64;
  const int D = e->D, Hm = e->Hm;
  // ---- parameter table: reference named_parameters() order (latte.py:226-255), frozen tables excluded
  add_param(e, "x_embedder.proj.weight", (int64_t)D * e->KPE);
  add_param(e, "x_embedder.proj.bias", D);
  add_param(e, "t_embedder.mlp.0.weight", (int64_t)D * 256);
  add_param(e, "t_embedder.mlp.0.bias", D);
  add_param(e, "t_embedder.mlp.2.weight", (int64_t)D * D);
  add_param(e, "t_embedder.mlp.2.bias", D);
  if (c.extras == 2) add_param(e, "y_embedder.embedding_table.weight", (int64_t)(c.num_classes + 1) * D);
  for (int i = 0; i < c.depth; ++i) {
    const std::string p = "blocks." + std::to_string(i) + ".";
    add_param(e, p + "attn.qkv.weight", (int64_t)3 * D * D);
    add_param(e, p + "attn.qkv.bias", 3 * D);
    add_param(e, p + "attn.proj.weight", (int64_t)D * D);
    add_param(e, p + "attn.proj.bias", D);
    add_param(e, p + "mlp.fc1.weight", (int64_t)Hm * D);
    add_param(e, p + "mlp.fc1.bias", Hm);
    add_param(e, p + "mlp.fc2.weight", (int64_t)D * Hm);
    add_param(e, p + "mlp.fc2.bias", D);
    add_param(e, p + "adaLN_modulation.1.weight", (int64_t)6 * D * D);
    add_param(e, p + "adaLN_modulation.1.bias", 6 * D);
  }
  add_param(e, "final_layer.linear.weight", (int64_t)e->P * D);
  add_param(e, "final_layer.linear.bias", e->P);
  add_param(e, "final_layer.adaLN_modulation.1.weight", (int64_t)2 * D * D);
  add_param(e, "final_layer.adaLN_modulation.1.bias", 2 * D);

  int rc = LATTE_OK;
  const size_t R = (size_t)e->rows_pad;
  auto A = [&](auto** p, size_t n) { if (!rc) rc = talloc(e, p, n); };
  A(&e->pos, (size_t)e->T * D); A(&e->temp, (size_t)e->F * D); A(&e->pe_wt, (size_t)e->KPE * D); A(&e->fin_wt, (size_t)D * e->P);
  e->xs.resize(2 * c.depth + 1);
  for (auto& x : e->xs) A(&x, R * D);
  e->blk.resize(c.depth);
  for (auto& b : e->blk) {
    A(&b.xn1, R * D); A(&b.qkv, R * 3 * D); A(&b.att, R * D); A(&b.y1, R * D); A(&b.xn2, R * D); A(&b.u, R * Hm); A(&b.h, R * Hm);
    A(&b.y2, R * D);
    A(&b.qkv_w, (size_t)3 * D * D); A(&b.qkv_wt, (size_t)3 * D * D); A(&b.proj_w, (size_t)D * D); A(&b.proj_wt, (size_t)D * D);
    A(&b.fc1_w, (size_t)Hm * D); A(&b.fc1_wt, (size_t)Hm * D); A(&b.fc2_w, (size_t)Hm * D); A(&b.fc2_wt, (size_t)Hm * D);
  }
  const size_t Bm = (size_t)max_batch;
  A(&e->tfreq, Bm * 256); A(&e->temb_pre, Bm * D); A(&e->temb_act, Bm * D); A(&e->cvec, Bm * D); A(&e->csilu, Bm * D);
  A(&e->mod, Bm * e->nmod); A(&e->dmod, Bm * e->nmod); A(&e->dc, Bm * D); A(&e->dtmp, Bm * D); A(&e->dtmp2, Bm * D);
  A(&e->t_orig, Bm);
  const size_t lat = Bm * e->F * e->H * e->H;
  A(&e->x_t, lat * e->Cin); A(&e->model_out, lat * e->Cout); A(&e->dmodel_out, lat * e->Cout);
  A(&e->dtok, R * e->P); A(&e->f32a, R * D); A(&e->f32b, R * D); A(&e->dx, R * D); A(&e->pix, R * e->KPE);
  A(&e->ones, R); A(&e->zeros, (size_t)std::max(Hm, 3 * D) + 256);
  const int Rr = train_rows_per_run(e->F * e->T);
  A(&e->part_rows, (size_t)(e->rows_max / Rr + 4) * 2 * D);
  A(&e->part_cols, (size_t)(colsum_chunks((int)e->rows_max) + 1) * std::max(Hm, 3 * D));
  {
    const size_t pr = (size_t)(e->rows_max / Rr / 4 + 1) * 2 * D;   // one block of 4 runs -> 2 partial rows
    A(&e->pg1, pr); A(&e->pl1, pr); A(&e->pg2, pr); A(&e->pg2b, pr); A(&e->pl2, pr);
    const size_t pcr = (size_t)std::max(colsum_chunks((int)e->rows_max) + 1, 260);   // column-sum chunks or weight-gradient splits (<= 256)
    A(&e->pc_fc1, pcr * Hm);
    A(&e->pc_qkv, pcr * 3 * D);
    A(&e->dc_ws, (size_t)adaln_dc_splits(e->nmod) * Bm * D);
    A(&e->no_ws, (size_t)narrow_blocks((int)e->rows_max) * ((size_t)32 * D + 32 + D));
    A(&e->pack_descs, (size_t)c.depth * 4);
    const int shp[4][2] = {{3 * D, D}, {D, D}, {Hm, D}, {D, Hm}};
    int t0 = 0;
    for (int j = 0; j < 4; ++j) {
      e->pack_plan.tile0[j] = t0;
      t0 += ((shp[j][0] + 31) / 32) * ((shp[j][1] + 31) / 32);
    }
    e->pack_plan.tiles_per_block = t0;
  }
  {
    // split-K partial products: splits * N * K with splits <= ceil(1024 / tiles) (+1), tiles = ceil(N / 128) (K / 128)
    int64_t worst = 0;
    const int shapes[4][2] = {{3 * D, D}, {D, D}, {Hm, D}, {D, Hm}};
    for (auto& s : shapes) {
      const int tiles = ((s[0] + 255) / 256) * (s[1] / 128);
      const int64_t splits = std::min<int64_t>(e->ld / 64 + 1, (768 + tiles - 1) / tiles) + 1;
      worst = std::max<int64_t>(worst, splits * s[0] * s[1]);
    }
    e->wg_ws_floats = worst;
    A(&e->wg_ws, (size_t)worst);
  }
  e->ng_ws_floats = 64LL * std::max<int64_t>((int64_t)e->P * D, (int64_t)max_batch * D) + 64LL * D * e->KPE + 64LL * D;
  A(&e->ng_ws, (size_t)e->ng_ws_floats);
  A(&e->attn_stats, (size_t)e->rows_max * c.num_heads * 3 + 16);
  e->loss_ws_floats = latte_training_workspace_floats(max_batch, (int64_t)e->F * e->Cin * e->H * e->H) + 3 * max_batch;
  A(&e->loss_ws, (size_t)e->loss_ws_floats);
  A(&e->stats, 4);
  A(&e->scaler, 8);
  A(&e->sumsq, (size_t)sumsq_blocks());
  A(&e->dyD, R * D); A(&e->dhH, R * Hm); A(&e->dxnH, R * D); A(&e->dqkvH, R * 3 * D); A(&e->xnh, R * D);
  if (!rc) rc = launch_fill_f32(e->ones, 1.0f, R, nullptr);
  if (!rc) rc = upload_scaler(e, 0);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(LATTE_ERR_HIP, "trainer_create: device error");
  if (rc) { latte_trainer_destroy(e); return rc; }
  *out = e;
  return LATTE_OK;
}

void latte_trainer_destroy(latte_trainer_t* e) {
  if (!e) return;
  for (void* p : e->allocs) (void)hipFree(p);
  delete e;
}

int latte_trainer_num_params(const latte_trainer_t* e) { return e ? (int)e->params.size() : 0; }
const char* latte_trainer_param_key(const latte_trainer_t* e, int i) {
  return (e && i >= 0 && i < (int)e->params.size()) ? e->params[i].key.c_str() : nullptr;
}
int64_t latte_trainer_param_offset(const latte_trainer_t* e, int i) { return (e && i >= 0 && i < (int)e->params.size()) ? e->params[i].offset : -1; }
int64_t latte_trainer_param_numel(const latte_trainer_t* e, int i) { return (e && i >= 0 && i < (int)e->params.size()) ? e->params[i].numel : -1; }
int64_t latte_trainer_total_numel(const latte_trainer_t* e) { return e ? e->total : 0; }

int latte_trainer_bind(latte_trainer_t* e, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* ema) {
  if (!e || !params || !grads || !exp_avg || !exp_avg_sq) return fail(LATTE_ERR_INVALID, "trainer_bind: null buffer");
  e->Pm = params; e->Gr = grads; e->M1 = exp_avg; e->V2 = exp_avg_sq; e->Ema = ema;
  e->weights_synced = false;
  {   // pointer table of the one-launch weight pack (train_fin.hip): [block][qkv, proj, fc1, fc2]
    const int D = e->D, Hm = e->Hm;
    std::vector<PackDesc> h((size_t)e->cfg.depth * 4);
    for (int i = 0; i < e->cfg.depth; ++i) {
      const std::string p = "blocks." + std::to_string(i) + ".";
      BlockBuf& b = e->blk[i];
      h[i * 4 + 0] = PackDesc{P_(e, p + "attn.qkv.weight"), b.qkv_w, b.qkv_wt, 3 * D, D};
      h[i * 4 + 1] = PackDesc{P_(e, p + "attn.proj.weight"), b.proj_w, b.proj_wt, D, D};
      h[i * 4 + 2] = PackDesc{P_(e, p + "mlp.fc1.weight"), b.fc1_w, b.fc1_wt, Hm, D};
      h[i * 4 + 3] = PackDesc{P_(e, p + "mlp.fc2.weight"), b.fc2_w, b.fc2_wt, D, Hm};
    }
    LATTE_HIP(hipDeviceSynchronize());
    LATTE_HIP(hipMemcpy(e->pack_descs, h.data(), h.size() * sizeof(PackDesc), hipMemcpyHostToDevice));
  }
  return LATTE_OK;
}

int latte_trainer_set_frozen(latte_trainer_t* e, const float* pos_embed, const float* temp_embed, int on_device, void* stream) {
  if (!e || !pos_embed || !temp_embed) return fail(LATTE_ERR_INVALID, "trainer_set_frozen: null");
  const hipMemcpyKind k = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  LATTE_HIP(hipMemcpyAsync(e->pos, pos_embed, sizeof(float) * e->T * e->D, k, (hipStream_t)stream));
  LATTE_HIP(hipMemcpyAsync(e->temp, temp_embed, sizeof(float) * e->F * e->D, k, (hipStream_t)stream));
  LATTE_HIP(hipStreamSynchronize((hipStream_t)stream));
  return LATTE_OK;
}

int latte_trainer_sync_weights(latte_trainer_t* e, void* stream) {
  if (!e || !e->Pm) return fail(LATTE_ERR_STATE, "trainer_sync_weights: bind the parameter buffers first");
  hipStream_t st = (hipStream_t)stream;
  const int D = e->D, Hm = e->Hm;
  int rc;
  if (e->fuse_small) {
    if ((rc = launch_pack_weights(e->pack_descs, e->cfg.depth, e->pack_plan, e->dt, st))) return rc;
  } else
  for (int i = 0; i < e->cfg.depth; ++i) {
    const std::string p = "blocks." + std::to_string(i) + ".";
    BlockBuf& b = e->blk[i];
    if ((rc = launch_pack_weight(P_(e, p + "attn.qkv.weight"), b.qkv_w, b.qkv_wt, 3 * D, D, e->dt, st))) return rc;
    if ((rc = launch_pack_weight(P_(e, p + "attn.proj.weight"), b.proj_w, b.proj_wt, D, D, e->dt, st))) return rc;
    if ((rc = launch_pack_weight(P_(e, p + "mlp.fc1.weight"), b.fc1_w, b.fc1_wt, Hm, D, e->dt, st))) return rc;
    if ((rc = launch_pack_weight(P_(e, p + "mlp.fc2.weight"), b.fc2_w, b.fc2_wt, D, Hm, e->dt, st))) return rc;
  }
  if ((rc = launch_transpose_f32(P_(e, "x_embedder.proj.weight"), e->pe_wt, D, e->KPE, st))) return rc;
  if ((rc = launch_transpose_f32(P_(e, "final_layer.linear.weight"), e->fin_wt, e->P, D, st))) return rc;
  e->weights_synced = true;
  return LATTE_OK;
}

// Forward with saved activations + loss terms + d loss / d model_output; the backward follows in stages (below).
// terms_out: device float [3][batch] = loss, mse, vb;  model_out_copy: optional device copy of the model output
int latte_trainer_begin(latte_trainer_t* e, const latte_schedule_t* s, int loss_type, const float* x_start, const float* noise,
                        const int64_t* t, const int64_t* y, int batch, float* terms_out, float* model_out_copy, void* stream) {
  if (!e || !s || !x_start || !noise || !t || !terms_out) return fail(LATTE_ERR_INVALID, "train step: null argument");
  if (!e->Pm) return fail(LATTE_ERR_STATE, "train step: bind the parameter buffers first");
  if (batch <= 0 || batch > e->max_batch) return fail(LATTE_ERR_STATE, "train step: batch exceeds max_batch");
  if (loss_type != 0 && loss_type != 1) return fail(LATTE_ERR_INVALID, "train step: loss_type must be 0 (MSE) or 1 (RESCALED_MSE); the KL losses train no mean head (gaussian_diffusion.py:741-752)");
  if (s->mean_type != 0) return fail(LATTE_ERR_INVALID, "train step: only the epsilon-prediction models of train.py:92 are supported");
  if ((s->var_type == 0) != (e->cfg.learn_sigma != 0)) return fail(LATTE_ERR_INVALID, "train step: schedule / model disagree on learn_sigma");
  const auto& c = e->cfg;
  if (c.extras == 2 && !y) return fail(LATTE_ERR_INVALID, "train step: class-conditional model needs y");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (!e->weights_synced && (rc = latte_trainer_sync_weights(e, stream))) return rc;
  const int D = e->D, T = e->T, F = e->F, Hm = e->Hm, B = batch, dt = e->dt, nmod = e->nmod;
  const int M = B * F * T, rps = F * T;
  const int hw = e->H * e->H;
  const float* tab = nullptr;
  if ((rc = schedule_device_tables(s, &tab, st))) return rc;
  // ---- x_t, original timesteps
  if ((rc = launch_q_sample(tab, s->num_timesteps, x_start, noise, t, B, (size_t)F * e->Cin * hw, e->x_t, st))) return rc;
  if (e->tmap_of != s || e->tmap_n != s->num_timesteps) {
    if (e->tmap_n < s->num_timesteps) { if ((rc = talloc(e, &e->tmap_dev, (size_t)s->num_timesteps, false))) return rc; }
    LATTE_HIP(hipMemcpyAsync(e->tmap_dev, s->timestep_map.data(), sizeof(int64_t) * s->num_timesteps, hipMemcpyHostToDevice, st));
    LATTE_HIP(hipStreamSynchronize(st));
    e->tmap_n = s->num_timesteps;
    e->tmap_of = s;
  }
  if ((rc = launch_gather_i64(e->tmap_dev, t, e->t_orig, B, st))) return rc;

  // ================================================================ forward
  // conditioning (latte.py:119-123,332-348): temb_pre = W0 freq(t) + b0; c = W2 SiLU(temb_pre) + b2 (+ y_emb); mod = adaLN(SiLU(c))
  if ((rc = launch_tfreq(e->t_orig, e->tfreq, B, st))) return rc;
  if ((rc = launch_small_linear(IN_PLAIN, e->tfreq, nullptr, P_(e, "t_embedder.mlp.0.weight"), P_(e, "t_embedder.mlp.0.bias"), nullptr,
                                nullptr, e->temb_pre, B, D, 256, D, st))) return rc;
  if ((rc = launch_silu_rows(e->temb_pre, e->temb_act, (size_t)B * D, st))) return rc;
  if ((rc = launch_small_linear(IN_PLAIN, e->temb_act, nullptr, P_(e, "t_embedder.mlp.2.weight"), P_(e, "t_embedder.mlp.2.bias"),
                                c.extras == 2 ? P_(e, "y_embedder.embedding_table.weight") : nullptr, c.extras == 2 ? y : nullptr, e->cvec, B,
                                D, D, D, st))) return rc;
  if ((rc = launch_silu_rows(e->cvec, e->csilu, (size_t)B * D, st))) return rc;
  for (int i = 0; i < c.depth; ++i) {
    const std::string p = "blocks." + std::to_string(i) + ".adaLN_modulation.1.";
    if ((rc = launch_small_linear(IN_PLAIN, e->csilu, nullptr, P_(e, p + "weight"), P_(e, p + "bias"), nullptr, nullptr,
                                  e->mod + (size_t)i * 6 * D, B, 6 * D, D, nmod, st))) return rc;
  }
  if ((rc = launch_small_linear(IN_PLAIN, e->csilu, nullptr, P_(e, "final_layer.adaLN_modulation.1.weight"),
                                P_(e, "final_layer.adaLN_modulation.1.bias"), nullptr, nullptr, e->mod + (size_t)c.depth * 6 * D, B, 2 * D, D,
                                nmod, st))) return rc;
  if ((rc = launch_patch_embed(e->x_t, e->pe_wt, P_(e, "x_embedder.proj.bias"), e->pos, e->xs[0], B * F, e->Cin, e->H, c.patch_size, D, st)))
    return rc;
  for (int i = 0; i < c.depth; ++i) {
    const bool spatial = (i % 2) == 0;
    const std::string p = "blocks." + std::to_string(i) + ".";
    BlockBuf& b = e->blk[i];
    const float* mb = e->mod + (size_t)i * 6 * D;
    float* x0 = e->xs[2 * i];
    float* x1 = e->xs[2 * i + 1];
    float* x2 = e->xs[2 * i + 2];
    // (fuse_small: blocks 1 .. depth-1 got xn1 -- and the temporal embedding -- from the previous block's closing gated add)
    if ((!e->fuse_small || i == 0) &&
        (rc = launch_ln_modulate(x0, x0, b.xn1, mb, mb + D, nmod, M, D, rps, i == 1 ? e->temp : nullptr, T, F, dt, st))) return rc;
    if ((rc = gemm_half(e, b.xn1, b.qkv_w, P_(e, p + "attn.qkv.bias"), b.qkv, M, 3 * D, D, st))) return rc;
    AttnArgs a{};
    a.qkv = b.qkv; a.out = b.att; a.heads = c.num_heads; a.hd = e->hd; a.D = D;
    a.sample_stride = rps; a.scale = 1.0f / std::sqrt((float)e->hd);
    if (spatial) { a.num_seq = B * F; a.L = T; a.U = F; a.seq_stride = T; a.row_stride = 1; }
    else         { a.num_seq = B * T; a.L = F; a.U = T; a.seq_stride = 1; a.row_stride = T; }
    if ((rc = launch_attention(a, dt, st))) return rc;
    if ((rc = gemm_half(e, b.att, b.proj_w, P_(e, p + "attn.proj.bias"), b.y1, M, D, D, st))) return rc;
    if (e->fuse_small) {   // x1 = x0 + g1 y1 and xn2 = LN-modulate(x1) in one pass over the rows
      if ((rc = launch_gated_add_ln(x0, b.y1, mb + 2 * D, nmod, x1, b.xn2, mb + 3 * D, mb + 4 * D, nmod, M, D, rps, nullptr, T, F, dt, st)))
        return rc;
    } else {
      if ((rc = launch_gated_add(x0, b.y1, mb + 2 * D, nmod, x1, M, D, rps, dt, st))) return rc;
      if ((rc = launch_ln_modulate(x1, x1, b.xn2, mb + 3 * D, mb + 4 * D, nmod, M, D, rps, nullptr, T, F, dt, st))) return rc;
    }
    if (gelu_fusable(e, M, Hm, D)) {   // u and h = gelu(u) out of one launch (bit-identical to the separate pass)
      if ((rc = gemm_gelu(e, EPI_BIAS_GELU_DUAL_H16, b.xn2, b.fc1_w, P_(e, p + "mlp.fc1.bias"), b.u, b.h, M, Hm, D, st))) return rc;
    } else {
      if ((rc = gemm_half(e, b.xn2, b.fc1_w, P_(e, p + "mlp.fc1.bias"), b.u, M, Hm, D, st))) return rc;
      if ((rc = launch_gelu_fwd(b.u, b.h, (size_t)M * Hm, dt, st))) return rc;
    }
    if ((rc = gemm_half(e, b.h, b.fc2_w, P_(e, p + "mlp.fc2.bias"), b.y2, M, D, Hm, st))) return rc;
    if (e->fuse_small && i + 1 < c.depth) {   // x2 = x1 + g2 y2 (+ temp_embed in front of block 1) and the NEXT block's xn1
      const float* nb = e->mod + (size_t)(i + 1) * 6 * D;
      if ((rc = launch_gated_add_ln(x1, b.y2, mb + 5 * D, nmod, x2, e->blk[i + 1].xn1, nb, nb + D, nmod, M, D, rps,
                                    i + 1 == 1 ? e->temp : nullptr, T, F, dt, st))) return rc;
    } else if ((rc = launch_gated_add(x1, b.y2, mb + 5 * D, nmod, x2, M, D, rps, dt, st))) return rc;
  }
  float* xl = e->xs[2 * c.depth];
  const float* fm = e->mod + (size_t)c.depth * 6 * D;
  if ((rc = launch_final_layer(xl, fm, fm + D, nmod, e->fin_wt, P_(e, "final_layer.linear.bias"), e->model_out, M, D, rps, T, c.patch_size,
                               e->Cout, e->H, st))) return rc;
  if (model_out_copy)
    LATTE_HIP(hipMemcpyAsync(model_out_copy, e->model_out, sizeof(float) * (size_t)B * F * e->Cout * hw, hipMemcpyDeviceToDevice, st));

  // ================================================================ loss values and d loss / d model_output
  float vb_scale = 1.0f;
  if (loss_type == 1) vb_scale = (float)(s->num_timesteps / 1000.0);
  if ((rc = latte_training_losses(s, loss_type, x_start, e->x_t, noise, e->model_out, t, B, F, e->Cin, hw, e->loss_ws,
                                  e->loss_ws_floats - 3 * e->max_batch, terms_out + B, terms_out + 2 * B, terms_out, stream))) return rc;
  if ((rc = launch_loss_grad(tab, s->num_timesteps, s->mean_type, s->var_type, x_start, e->x_t, noise, e->model_out, t, B, F, e->Cin, hw,
                             vb_scale, e->dmodel_out, st))) return rc;
  if (scaling_active(e) && (rc = launch_scale_f32_dev(e->dmodel_out, e->scaler, 0, (size_t)B * F * e->Cout * hw, st))) return rc;

  LATTE_HIP(hipMemsetAsync(e->dc, 0, sizeof(float) * (size_t)B * D, st));   // d SiLU(c), summed over the adaLN linears by the stages
  e->cur_batch = B;
  e->cur_y = y;
  e->next_stage = 0;
  return LATTE_OK;
}

// adaLN linear of block i (i == depth: the final layer's): bias / weight gradients from the finished dmod rows, and its
// contribution to d SiLU(c)
static int adaln_bwd(latte_trainer_t* e, int i, hipStream_t st) {
  const auto& c = e->cfg;
  const int D = e->D, B = e->cur_batch, nmod = e->nmod;
  const bool fin = i == c.depth;
  const std::string p = fin ? "final_layer.adaLN_modulation.1." : "blocks." + std::to_string(i) + ".adaLN_modulation.1.";
  const int N = fin ? 2 * D : 6 * D;
  const float* dm = e->dmod + (size_t)i * 6 * D;
  int rc;
  if ((rc = launch_rows_sum(dm, B, nmod, N, G_(e, p + "bias"), 0, st))) return rc;
  if ((rc = launch_naive_gemm(dm, 1, nmod, e->csilu, D, 1, G_(e, p + "weight"), D, 1, N, D, B, 1.0f, 0, st))) return rc;          // dW[n, k]
  if ((rc = launch_naive_gemm(dm, nmod, 1, P_(e, p + "weight"), D, 1, e->dtmp, D, 1, B, D, N, 1.0f, 0, st, 48, e->ng_ws))) return rc;  // d csilu
  return launch_add_rows(e->dc, e->dtmp, (size_t)B * D, st);
}

int latte_trainer_num_stages(const latte_trainer_t* e) { return e ? e->cfg.depth + 2 : 0; }

// the slice of the flat gradient buffer that stage `stage` finalises: 0 = final layer, 1 .. depth = blocks depth-1 .. 0,
// depth + 1 = patch embed + t / y embedders (the head of the buffer)
int latte_trainer_stage_range(const latte_trainer_t* e, int stage, int64_t* offset, int64_t* numel) {
  if (!e || !offset || !numel || stage < 0 || stage > e->cfg.depth + 1) return fail(LATTE_ERR_INVALID, "stage_range: bad stage");
  auto off = [&](const std::string& k) { return e->params[e->index.at(k)].offset; };
  const int depth = e->cfg.depth;
  if (stage == 0) { *offset = off("final_layer.linear.weight"); *numel = e->total - *offset; }
  else if (stage <= depth) {
    const int i = depth - stage;
    *offset = off("blocks." + std::to_string(i) + ".attn.qkv.weight");
    const int64_t end = i + 1 < depth ? off("blocks." + std::to_string(i + 1) + ".attn.qkv.weight") : off("final_layer.linear.weight");
    *numel = end - *offset;
  } else { *offset = 0; *numel = off("blocks.0.attn.qkv.weight"); }
  return LATTE_OK;
}

// Backward of one stage (in order 0, 1, ..., depth + 1 after latte_trainer_begin).  After stage k the gradient slice
// latte_trainer_stage_range(k) is final: the data-parallel driver starts its all-reduce while the next stages run.
static int backward_stage_impl(latte_trainer_t* e, int stage, void* stream);
int latte_trainer_backward_stage(latte_trainer_t* e, int stage, void* stream) {
  int rc = backward_stage_impl(e, stage, stream);
  // (fuse_small: the block stages' slices were unscaled by the kernels that wrote them)
  if (!rc && scaling_active(e) && !(e->fuse_small && stage >= 1 && stage <= e->cfg.depth)) {   // the slice this stage finalised leaves the loss-scaled domain
    int64_t off = 0, n = 0;
    if ((rc = latte_trainer_stage_range(e, stage, &off, &n))) return rc;
    rc = launch_scale_f32_dev(e->Gr + off, e->scaler, 1, (size_t)n, (hipStream_t)stream);
  }
  return rc;
}

int latte_trainer_set_option(latte_trainer_t* e, const char* name, double value) {
  if (!e || !name) return fail(LATTE_ERR_INVALID, "trainer_set_option: null argument");
  if (std::string(name) == "loss_scale") {
    int ex = 0;
    if (!(value >= 1.0) || value > 16777216.0 || std::frexp(value, &ex) != 0.5)
      return fail(LATTE_ERR_INVALID, "loss_scale must be a power of two in [1, 2^24]");
    e->loss_scale = (float)value;
    return upload_scaler(e, 1);
  }
  if (std::string(name) == "dynamic_loss_scale") {   // 0: the scale stays what "loss_scale" set (overflowing steps are still skipped)
    e->dynamic_scale = value != 0.0 ? 1 : 0;
    return upload_scaler(e, 2);
  }
  if (std::string(name) == "loss_scale_growth_interval") {
    if (!(value >= 1.0) || value > 1e7) return fail(LATTE_ERR_INVALID, "loss_scale_growth_interval must be in [1, 1e7]");
    e->growth_interval = (float)value;
    return upload_scaler(e, 2);
  }
  if (std::string(name) == "fuse_small") {   // 0: the separate finalize / column-sum / adaLN launches of rounds 2 - 6, 1: train_fin.hip (default)
    if (e->next_stage <= e->cfg.depth + 1) return fail(LATTE_ERR_STATE, "fuse_small: a step is in flight");
    e->fuse_small = value != 0.0 ? 1 : 0;
    return LATTE_OK;
  }
  if (std::string(name) == "fuse_gelu") {   // 0: separate GELU passes (rounds 2 - 5), 1: inside the GEMM epilogues (default)
    e->fuse_gelu = value != 0.0 ? 1 : 0;
    return LATTE_OK;
  }
  return fail(LATTE_ERR_INVALID, std::string("trainer_set_option: unknown option '") + name + "'");
}

static int backward_stage_impl(latte_trainer_t* e, int stage, void* stream) {
  if (!e || !e->Pm) return fail(LATTE_ERR_STATE, "backward_stage: no step in flight");
  if (stage != e->next_stage || stage > e->cfg.depth + 1) return fail(LATTE_ERR_STATE, "backward_stage: stages run in order after latte_trainer_begin");
  e->next_stage = stage + 1;
  const auto& c = e->cfg;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  const int D = e->D, T = e->T, F = e->F, Hm = e->Hm, B = e->cur_batch, dt = e->dt, nmod = e->nmod;
  const int M = B * F * T, rps = F * T;
  const int64_t* y = e->cur_y;
  if (stage == 0) {
  float* xl = e->xs[2 * c.depth];
  const float* fm = e->mod + (size_t)c.depth * 6 * D;
  // final layer: out = unpatchify(Linear(LN-mod(x)))
  if ((rc = launch_unpatchify_bwd(e->dmodel_out, e->dtok, B * F, e->G, c.patch_size, e->Cout, st))) return rc;
  if (e->fuse_small && e->P <= 32) {
    // the linear's weight / bias gradients (exact fp32 products of dtok with the half LN output, as below) in one launch + its
    // reductions, and its input gradient straight to half (train_fin.hip)
    if ((rc = launch_ln_modulate(xl, xl, e->xnh, fm, fm + D, nmod, M, D, rps, nullptr, T, F, dt, st))) return rc;
    if ((rc = launch_narrow_outer(e->dtok, e->P, e->xnh, 1, D, M, G_(e, "final_layer.linear.weight"), D, 1, G_(e, "final_layer.linear.bias"),
                                  nullptr, e->no_ws, dt, nullptr, st))) return rc;
    if ((rc = launch_narrow_dx(e->dtok, e->P, P_(e, "final_layer.linear.weight"), D, M, e->dxnH, dt, st))) return rc;
  } else {
  if ((rc = launch_naive_gemm(e->ones, 0, 1, e->dtok, e->P, 1, G_(e, "final_layer.linear.bias"), e->P, 1, 1, e->P, M, 1.0f, 0, st, 64, e->ng_ws)))
    return rc;
  if ((rc = launch_ln_modulate(xl, xl, e->xnh, fm, fm + D, nmod, M, D, rps, nullptr, T, F, dt, st))) return rc;
  if ((rc = launch_convert_h16_to_f32(e->xnh, e->f32a, (int64_t)M * D, dt, st))) return rc;
  if ((rc = launch_naive_gemm(e->dtok, 1, e->P, e->f32a, D, 1, G_(e, "final_layer.linear.weight"), D, 1, e->P, D, M, 1.0f, 0, st, 64, e->ng_ws)))
    return rc;
  if ((rc = launch_naive_gemm(e->dtok, e->P, 1, P_(e, "final_layer.linear.weight"), D, 1, e->f32b, D, 1, M, D, e->P, 1.0f, 0, st))) return rc;
  if ((rc = launch_convert_f32_to_h16(e->f32b, e->dxnH, (int64_t)M * D, dt, st))) return rc;
  }
  {
    float* dm = e->dmod + (size_t)c.depth * 6 * D;
    if (e->fuse_small) {   // shift / scale gradients of the final modulation and its adaLN linear's bias / weight gradients in one launch
      {   // ... and the gated residual's backward of the last block's MLP branch on the same pass over dx (its stage starts at the wgrad)
        const int il = c.depth - 1;
        if ((rc = launch_ln_bwd(e->dxnH, xl, fm + D, nmod, nullptr, e->dx, e->pl1, nullptr, nullptr, nmod, M, D, rps, dt, st, e->blk[il].y2,
                                e->mod + (size_t)il * 6 * D + 5 * D, nmod, e->dyD, (il & 1) ? e->pg2b : e->pg2))) return rc;
      }
      StageFinArgs a{};
      a.mod_src[0] = a.mod_src[1] = e->pl1; a.mod_nsum[0] = a.mod_nsum[1] = 2; a.mod_which[0] = 0; a.mod_which[1] = 1;
      a.n_mod = 2; a.rows_per_sample = rps / (4 * train_rows_per_run(rps)); a.B = B; a.D = D;
      a.dmod = dm; a.dmod_stride = nmod; a.csilu = e->csilu;
      a.dW = G_(e, "final_layer.adaLN_modulation.1.weight"); a.db = G_(e, "final_layer.adaLN_modulation.1.bias");
      a.n_bias = 0; a.scaler = nullptr;   // this stage's slice is unscaled by the pass behind the stage
      return launch_stage_finalize(a, st);
    }
    if ((rc = launch_ln_bwd(e->dxnH, xl, fm + D, nmod, nullptr, e->dx, e->part_rows, dm, dm + D, nmod, M, D, rps, dt, st))) return rc;
  }
  return adaln_bwd(e, c.depth, st);
  }
  if (stage <= c.depth) {
    const int i = c.depth - stage;
    const bool spatial = (i % 2) == 0;
    const std::string p = "blocks." + std::to_string(i) + ".";
    BlockBuf& b = e->blk[i];
    const float* mb = e->mod + (size_t)i * 6 * D;
    float* dm = e->dmod + (size_t)i * 6 * D;
    if (e->fuse_small) {
      const bool us = scaling_active(e);
      // ---- MLP branch: x2 = x1 + g2 * (fc2(gelu(fc1(xn2)))).  dy = g2 dx and the gate / bias partial rows were left by the LayerNorm
      // backward that produced dx (the previous stage's last pass); the partial rows wait for this stage's finalize launch
      float* pg2 = (i & 1) ? e->pg2b : e->pg2;
      if ((rc = wgrad(e, e->dyD, b.h, M, D, Hm, G_(e, p + "mlp.fc2.weight"), st, us))) return rc;
      if (gelu_fusable(e, M, Hm, D)) {
        if ((rc = gemm_gelu(e, EPI_DGELU_H16, e->dyD, b.fc2_wt, e->zeros, e->dhH, b.u, M, Hm, D, st))) return rc;
      } else {
        if ((rc = gemm_half(e, e->dyD, b.fc2_wt, e->zeros, e->dhH, M, Hm, D, st))) return rc;
        if ((rc = launch_gelu_bwd(b.u, e->dhH, e->dhH, (size_t)M * Hm, dt, st))) return rc;
      }
      // bias gradients of fc1 / qkv: column sums of dY on the weight-gradient launch where the 8-wave kernel takes the shape
      int fc1_rows = colsum_chunks(M), qkv_rows = colsum_chunks(M);
      const bool cs_fc1 = gemm_tn8_ok(M, Hm, D), cs_qkv = gemm_tn8_ok(M, 3 * D, D);
      if (!cs_fc1 && (rc = launch_colsum_half(e->dhH, M, Hm, e->pc_fc1, nullptr, 0, dt, st))) return rc;
      if ((rc = wgrad(e, e->dhH, b.xn2, M, Hm, D, G_(e, p + "mlp.fc1.weight"), st, us, cs_fc1 ? e->pc_fc1 : nullptr, cs_fc1 ? &fc1_rows : nullptr)))
        return rc;
      if ((rc = gemm_half(e, e->dhH, b.fc1_wt, e->zeros, e->dxnH, M, D, Hm, st))) return rc;
      // ---- attention branch: x1 = x0 + g1 * proj(attn(qkv(xn1))): its gate backward rides on LN2's backward
      if ((rc = launch_ln_bwd(e->dxnH, e->xs[2 * i + 1], mb + 4 * D, nmod, e->dx, e->dx, e->pl2, nullptr, nullptr, nmod, M, D, rps, dt, st,
                              b.y1, mb + 2 * D, nmod, e->dyD, e->pg1))) return rc;
      if ((rc = wgrad(e, e->dyD, b.att, M, D, D, G_(e, p + "attn.proj.weight"), st, us))) return rc;
      if ((rc = gemm_half(e, e->dyD, b.proj_wt, e->zeros, e->dxnH, M, D, D, st))) return rc;   // d(attention output)
      if (spatial) rc = launch_attention_bwd(b.qkv, b.att, e->dxnH, e->dqkvH, e->attn_stats, B * F, T, c.num_heads, e->hd, F, rps, T, 1, dt, st);
      else         rc = launch_attention_bwd(b.qkv, b.att, e->dxnH, e->dqkvH, e->attn_stats, B * T, F, c.num_heads, e->hd, T, rps, 1, T, dt, st);
      if (rc) return rc;
      if (!cs_qkv && (rc = launch_colsum_half(e->dqkvH, M, 3 * D, e->pc_qkv, nullptr, 0, dt, st))) return rc;
      if ((rc = wgrad(e, e->dqkvH, b.xn1, M, 3 * D, D, G_(e, p + "attn.qkv.weight"), st, us, cs_qkv ? e->pc_qkv : nullptr,
                      cs_qkv ? &qkv_rows : nullptr))) return rc;
      if ((rc = gemm_half(e, e->dqkvH, b.qkv_wt, e->zeros, e->dxnH, M, D, 3 * D, st))) return rc;
      if (i > 0) {   // LN1's backward + the gate backward of block i - 1's MLP branch (the next stage's first step)
        if ((rc = launch_ln_bwd(e->dxnH, e->xs[2 * i], mb + D, nmod, e->dx, e->dx, e->pl1, nullptr, nullptr, nmod, M, D, rps, dt, st,
                                e->blk[i - 1].y2, e->mod + (size_t)(i - 1) * 6 * D + 5 * D, nmod, e->dyD, ((i - 1) & 1) ? e->pg2b : e->pg2)))
          return rc;
      } else if ((rc = launch_ln_bwd(e->dxnH, e->xs[2 * i], mb + D, nmod, e->dx, e->dx, e->pl1, nullptr, nullptr, nmod, M, D, rps, dt, st)))
        return rc;
      // ---- one launch: the six modulation gradients, the adaLN linear's bias / weight gradients, the four linears' bias gradients
      StageFinArgs a{};
      const float* msrc[6] = {e->pl1, e->pl1, e->pg1, e->pl2, e->pl2, pg2};
      const int mwhich[6] = {0, 1, 0, 0, 1, 0};
      for (int k = 0; k < 6; ++k) { a.mod_src[k] = msrc[k]; a.mod_nsum[k] = 2; a.mod_which[k] = mwhich[k]; }
      const int rb = rps / (4 * train_rows_per_run(rps));   // partial rows (blocks of 4 runs) per sample
      a.n_mod = 6; a.rows_per_sample = rb; a.B = B; a.D = D;
      a.dmod = dm; a.dmod_stride = nmod; a.csilu = e->csilu;
      a.dW = G_(e, p + "adaLN_modulation.1.weight"); a.db = G_(e, p + "adaLN_modulation.1.bias");
      a.n_bias = 4;
      a.bias_src[0] = e->pc_qkv;   a.bias_rows[0] = qkv_rows;   a.bias_stride[0] = 3 * D; a.bias_cols[0] = 3 * D; a.bias_out[0] = G_(e, p + "attn.qkv.bias");
      a.bias_src[1] = e->pg1 + D;  a.bias_rows[1] = B * rb; a.bias_stride[1] = 2 * D; a.bias_cols[1] = D;     a.bias_out[1] = G_(e, p + "attn.proj.bias");
      a.bias_src[2] = e->pc_fc1;   a.bias_rows[2] = fc1_rows;   a.bias_stride[2] = Hm;    a.bias_cols[2] = Hm;    a.bias_out[2] = G_(e, p + "mlp.fc1.bias");
      a.bias_src[3] = pg2 + D;      a.bias_rows[3] = B * rb; a.bias_stride[3] = 2 * D; a.bias_cols[3] = D;     a.bias_out[3] = G_(e, p + "mlp.fc2.bias");
      a.scaler = us ? e->scaler : nullptr;
      return launch_stage_finalize(a, st);
    }
    // ---- MLP branch: x2 = x1 + g2 * (fc2(gelu(fc1(xn2))))
    if ((rc = launch_gate_bwd(e->dx, b.y2, mb + 5 * D, nmod, e->dyD, e->part_rows, dm + 5 * D, nmod, M, D, rps, dt, st))) return rc;
    if ((rc = launch_colsum_half(e->dyD, M, D, e->part_cols, G_(e, p + "mlp.fc2.bias"), 0, dt, st))) return rc;
    if ((rc = wgrad(e, e->dyD, b.h, M, D, Hm, G_(e, p + "mlp.fc2.weight"), st))) return rc;
    if (gelu_fusable(e, M, Hm, D)) {   // du = (dy W2) gelu'(u) in the GEMM's epilogue (the product is not rounded to half in between)
      if ((rc = gemm_gelu(e, EPI_DGELU_H16, e->dyD, b.fc2_wt, e->zeros, e->dhH, b.u, M, Hm, D, st))) return rc;
    } else {
      if ((rc = gemm_half(e, e->dyD, b.fc2_wt, e->zeros, e->dhH, M, Hm, D, st))) return rc;
      if ((rc = launch_gelu_bwd(b.u, e->dhH, e->dhH, (size_t)M * Hm, dt, st))) return rc;
    }
    if ((rc = launch_colsum_half(e->dhH, M, Hm, e->part_cols, G_(e, p + "mlp.fc1.bias"), 0, dt, st))) return rc;
    if ((rc = wgrad(e, e->dhH, b.xn2, M, Hm, D, G_(e, p + "mlp.fc1.weight"), st))) return rc;
    if ((rc = gemm_half(e, e->dhH, b.fc1_wt, e->zeros, e->dxnH, M, D, Hm, st))) return rc;
    if ((rc = launch_ln_bwd(e->dxnH, e->xs[2 * i + 1], mb + 4 * D, nmod, e->dx, e->dx, e->part_rows, dm + 3 * D, dm + 4 * D, nmod, M, D, rps, dt,
                            st))) return rc;
    // ---- attention branch: x1 = x0 + g1 * proj(attn(qkv(xn1)))
    if ((rc = launch_gate_bwd(e->dx, b.y1, mb + 2 * D, nmod, e->dyD, e->part_rows, dm + 2 * D, nmod, M, D, rps, dt, st))) return rc;
    if ((rc = launch_colsum_half(e->dyD, M, D, e->part_cols, G_(e, p + "attn.proj.bias"), 0, dt, st))) return rc;
    if ((rc = wgrad(e, e->dyD, b.att, M, D, D, G_(e, p + "attn.proj.weight"), st))) return rc;
    if ((rc = gemm_half(e, e->dyD, b.proj_wt, e->zeros, e->dxnH, M, D, D, st))) return rc;   // d(attention output)
    if (spatial) rc = launch_attention_bwd(b.qkv, b.att, e->dxnH, e->dqkvH, e->attn_stats, B * F, T, c.num_heads, e->hd, F, rps, T, 1, dt, st);
    else         rc = launch_attention_bwd(b.qkv, b.att, e->dxnH, e->dqkvH, e->attn_stats, B * T, F, c.num_heads, e->hd, T, rps, 1, T, dt, st);
    if (rc) return rc;
    if ((rc = launch_colsum_half(e->dqkvH, M, 3 * D, e->part_cols, G_(e, p + "attn.qkv.bias"), 0, dt, st))) return rc;
    if ((rc = wgrad(e, e->dqkvH, b.xn1, M, 3 * D, D, G_(e, p + "attn.qkv.weight"), st))) return rc;
    if ((rc = gemm_half(e, e->dqkvH, b.qkv_wt, e->zeros, e->dxnH, M, D, 3 * D, st))) return rc;
    if ((rc = launch_ln_bwd(e->dxnH, e->xs[2 * i], mb + D, nmod, e->dx, e->dx, e->part_rows, dm, dm + D, nmod, M, D, rps, dt, st))) return rc;
    return adaln_bwd(e, i, st);
  }
  // ---- patch embed (latte.py:233,330-331): tokens = pix W^T + b + pos
  if (e->fuse_small && e->KPE <= 32) {   // dW[k][j] = sum_m dx[m][k] pix[m][j], db[k] = sum_m dx[m][k]: one launch + reductions
    if ((rc = launch_im2col_patch(e->x_t, e->pix, B * F, e->G, c.patch_size, e->Cin, st))) return rc;
    if ((rc = launch_narrow_outer(e->pix, e->KPE, e->dx, 0, D, M, G_(e, "x_embedder.proj.weight"), 1, e->KPE, nullptr,
                                  G_(e, "x_embedder.proj.bias"), e->no_ws, dt, nullptr, st))) return rc;
  } else {
  if ((rc = launch_naive_gemm(e->ones, 0, 1, e->dx, D, 1, G_(e, "x_embedder.proj.bias"), D, 1, 1, D, M, 1.0f, 0, st, 64, e->ng_ws))) return rc;
  if ((rc = launch_im2col_patch(e->x_t, e->pix, B * F, e->G, c.patch_size, e->Cin, st))) return rc;
  if ((rc = launch_naive_gemm(e->dx, 1, D, e->pix, e->KPE, 1, G_(e, "x_embedder.proj.weight"), e->KPE, 1, D, e->KPE, M, 1.0f, 0, st, 64,
                              e->ng_ws))) return rc;
  }
  // ---- conditioning tail: d SiLU(c) (summed by the stages) -> c = temb (+ y_emb) -> t_embedder MLP
  if (e->fuse_small) {   // d SiLU(c) = dmod W over every adaLN linear of the model at once (the stages left their dmod rows)
    auto off = [&](const std::string& k) { return e->params[e->index.at(k)].offset; };
    const int64_t o0 = off("blocks.0.adaLN_modulation.1.weight");
    const int64_t stride = off("blocks.1.adaLN_modulation.1.weight") - o0;
    if ((rc = launch_adaln_dc(e->dmod, nmod, B, e->Pm + o0, (long)stride, c.depth, 6 * D, P_(e, "final_layer.adaLN_modulation.1.weight"), D,
                              e->dc_ws, e->dc, st))) return rc;
  }
  if ((rc = launch_silu_bwd(e->dc, e->cvec, e->dtmp, (size_t)B * D, 0, st))) return rc;     // dtmp = dc (gradient of c = temb + y_emb)
  if (c.extras == 2) {
    // the scatter accumulates (a label may repeat inside the batch): clear the slice first, so that forward_backward ASSIGNS this
    // gradient like every other one (two calls without an optimiser step in between must not double it)
    float* gy = G_(e, "y_embedder.embedding_table.weight");
    LATTE_HIP(hipMemsetAsync(gy, 0, (size_t)(c.num_classes + 1) * D * sizeof(float), st));
    if ((rc = launch_embedding_bwd(e->dtmp, y, gy, B, D, st))) return rc;
  }
  // temb = W2 SiLU(temb_pre) + b2
  if ((rc = launch_rows_sum(e->dtmp, B, D, D, G_(e, "t_embedder.mlp.2.bias"), 0, st))) return rc;
  if ((rc = launch_naive_gemm(e->dtmp, 1, D, e->temb_act, D, 1, G_(e, "t_embedder.mlp.2.weight"), D, 1, D, D, B, 1.0f, 0, st))) return rc;
  if ((rc = launch_naive_gemm(e->dtmp, D, 1, P_(e, "t_embedder.mlp.2.weight"), D, 1, e->dtmp2, D, 1, B, D, D, 1.0f, 0, st))) return rc;
  if ((rc = launch_silu_bwd(e->dtmp2, e->temb_pre, e->dtmp2, (size_t)B * D, 0, st))) return rc;
  // temb_pre = W0 freq(t) + b0
  if ((rc = launch_rows_sum(e->dtmp2, B, D, D, G_(e, "t_embedder.mlp.0.bias"), 0, st))) return rc;
  if ((rc = launch_naive_gemm(e->dtmp2, 1, D, e->tfreq, 256, 1, G_(e, "t_embedder.mlp.0.weight"), 256, 1, D, 256, B, 1.0f, 0, st))) return rc;
  return LATTE_OK;
}

// begin + every stage: one micro-batch, gradients of terms["loss"].mean() in the bound gradient buffer
int latte_trainer_forward_backward(latte_trainer_t* e, const latte_schedule_t* s, int loss_type, const float* x_start, const float* noise,
                                   const int64_t* t, const int64_t* y, int batch, float* terms_out, float* model_out_copy, void* stream) {
  int rc = latte_trainer_begin(e, s, loss_type, x_start, noise, t, y, batch, terms_out, model_out_copy, stream);
  for (int k = 0; !rc && k < latte_trainer_num_stages(e); ++k) rc = latte_trainer_backward_stage(e, k, stream);
  return rc;
}

// clip_grad_norm_ (utils.py:72-117) + AdamW (train.py:127) + update_ema (utils.py:191-200) on the bound flat buffers, then the
// half operand copies are re-packed.  norm_out: optional device float[2] = {total gradient norm, applied clip coefficient}.
int latte_trainer_optimizer_step(latte_trainer_t* e, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                 float clip_max_norm, int clip, float ema_decay, float* norm_out, void* stream) {
  if (!e || !e->Pm) return fail(LATTE_ERR_STATE, "optimizer_step: bind the parameter buffers first");
  if (step < 0) return fail(LATTE_ERR_INVALID, "optimizer_step: step counts from 1 (0: the trainer's own count of applied updates)");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if ((rc = launch_grad_norm(e->Gr, (size_t)e->total, e->sumsq, clip_max_norm, clip, e->stats, e->scaler, st))) return rc;
  if (norm_out) LATTE_HIP(hipMemcpyAsync(norm_out, e->stats, sizeof(float) * 2, hipMemcpyDeviceToDevice, st));
  if ((rc = launch_adamw_ema(e->Pm, e->Gr, e->M1, e->V2, e->Ema, (size_t)e->total, lr, beta1, beta2, eps, weight_decay, step, ema_decay,
                             e->stats, e->scaler + 2, st))) return rc;
  return latte_trainer_sync_weights(e, stream);
}

// {loss scale, applied steps since it changed, applied optimiser updates, skipped updates, last call skipped, dynamic, growth interval,
// largest scale} -> host doubles (synchronises the device: logging / tests, not the step path)
int latte_trainer_scaler_state(latte_trainer_t* e, double* out8) {
  if (!e || !out8) return fail(LATTE_ERR_INVALID, "trainer_scaler_state: null argument");
  float h[8];
  LATTE_HIP(hipDeviceSynchronize());
  LATTE_HIP(hipMemcpy(h, e->scaler, sizeof(h), hipMemcpyDeviceToHost));
  for (int i = 0; i < 8; ++i) out8[i] = h[i];
  return LATTE_OK;
}

}  // extern "C"
