// LatteT2V (Latte-1 text-to-video) denoiser forward on the MI355X kernels -- SURVEY.md section 8(f) rank 2.
//
//   LatteT2V.forward                 /root/reference/models/latte_t2v.py:677-941
//   BasicTransformerBlock_ (temporal) :126-396     FeedForward :69-124     AdaLayerNormSingle :398-428
//   spatial block / Attention / PatchEmbed / CaptionProjection / CombinedTimestepSizeEmbeddings: diffusers==0.24.0
//   (not vendored in the reference; restated in oracle/diffusers_standin.py -- parity unpinned for those leaves)
//
// Layout: the same canonical token order as the class-conditional engine (row = (b*F + f)*T + t, fp32 residual stream,
// half GEMM operands); the [B, C, F, H, W] input / output of the reference is permuted once at each end.  Per forward:
//   temb = MLP(sincos(t)); t6 = Linear(SiLU(temb)); mod = adaLN-single tables + t6 (one row per sample)
//   ctx  = caption projection of the T5 tokens (two GEMMs, GELU-tanh between), shared by the frames of a sample
//   spatial block : LN-modulate -> fused q|k|v GEMM -> self-attention over T tokens -> out GEMM (gated residual)
//                   -> q GEMM on the UN-normalised stream, k|v GEMM on ctx -> cross-attention (additive mask bias)
//                   -> out GEMM (residual, gate 1) -> LN-modulate -> fc1 GELU -> fc2 (gated residual)
//   temporal block: the same without cross-attention, attention over the F frames of a token (row stride T);
//                   temp_pos_embed is added in front of the first one
//   head          : LN-modulate with scale_shift_table + temb -> proj_out -> unpatchify
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/latte_amd.h"
#include "common.h"

using namespace latte;

namespace {

enum T2VKind { TK_F32, TK_F32_TRANSPOSE, TK_H16, TK_SKIP };
struct T2VSlot {
  std::string key;
  int64_t numel;
  T2VKind kind;
  void* dst;
  int rows, cols;
  bool loaded = false;
};

struct T2VBlock {
  half_t *qkv_w, *o_w, *q2_w, *kv2_w, *o2_w, *fc1_w, *fc2_w;   // q2 / kv2 / o2: spatial blocks only
  float *qkv_b, *o_b, *q2_b, *kv2_b, *o2_b, *fc1_b, *fc2_b;
};

void sincos_1d(int dim, const std::vector<double>& pos, std::vector<double>& out /* [n][dim] */) {
  const int half = dim / 2, n = (int)pos.size();
  out.assign((size_t)n * dim, 0.0);
  for (int m = 0; m < n; ++m)
    for (int d = 0; d < half; ++d) {
      const double omega = 1.0 / std::pow(10000.0, (double)d / (dim / 2.0));
      const double v = pos[m] * omega;
      out[(size_t)m * dim + d] = std::sin(v);
      out[(size_t)m * dim + half + d] = std::cos(v);
    }
}

}  // namespace

struct latte_t2v {
  latte_t2v_config_t cfg;
  int max_batch = 0, D = 0, T = 0, G = 0, F = 0, H = 0, P = 0, KPE = 0, Hm = 0, hd = 0, heads = 0, L = 0, Cc = 0, maxk = 0;
  int nblk = 0;            // 2 * num_layers: block 2i = spatial i, 2i + 1 = temporal i
  int fuse_qkv_attn = 3;   // as latte_engine: bit 0 spatial / bit 1 temporal blocks run to_q|k|v + attention as ONE kernel (qkv_attn.hip)
                           // where the shape allows it (256 tokens per frame / 16 frames); latte_t2v_set_option("fuse_qkv_attn", ...)
  int64_t rows_pad = 0, trows_pad = 0;
  std::vector<T2VBlock> blocks;
  float *tables = nullptr, *head_table = nullptr, *pos = nullptr, *temp = nullptr, *pe_wt = nullptr, *pe_b = nullptr,
        *t1_w = nullptr, *t1_b = nullptr, *t2_w = nullptr, *t2_b = nullptr, *ada_w = nullptr, *ada_b = nullptr, *fin_wt = nullptr,
        *fin_b = nullptr, *cap1_b = nullptr, *cap2_b = nullptr, *ones = nullptr;
  half_t *cap1_w = nullptr, *cap2_w = nullptr;
  float *xin = nullptr, *xres = nullptr, *temb0 = nullptr, *temb = nullptr, *t6 = nullptr, *mod = nullptr, *out_bf = nullptr,
        *kbias = nullptr, *stage = nullptr;
  half_t *xn = nullptr, *qkv = nullptr, *hbuf = nullptr, *ctx_in = nullptr, *ctx1 = nullptr, *ctx = nullptr, *kv = nullptr;
  // text context of a chain (latte_t2v_set_text): cross-attention K|V of EVERY spatial block + the additive mask bias,
  // functions of the text alone, computed once instead of once per denoising step
  half_t* kv_all = nullptr;         // [num_layers][trows_pad][2D]
  int txt_batch = 0, txt_nk = 0;    // > 0: a text context for that many samples / tokens is installed
  bool txt_has_mask = false;
  int64_t* tsteps = nullptr;        // device copy of a chain's timesteps (latte_t2v_guided_ddim_loop)
  int tsteps_cap = 0;
  int64_t stage_numel = 0;
  std::vector<T2VSlot> slots;
  std::map<std::string, int> slot_index;
  std::vector<void*> allocs;
};

namespace {

template <typename Tp>
int t2v_alloc(latte_t2v* e, Tp** p, size_t count) {
  void* q = nullptr;
  const size_t bytes = std::max<size_t>(count * sizeof(Tp), 16);
  LATTE_HIP(hipMalloc(&q, bytes));
  LATTE_HIP(hipMemset(q, 0, bytes));
  e->allocs.push_back(q);
  *p = (Tp*)q;
  return LATTE_OK;
}

void t2v_slot(latte_t2v* e, const std::string& key, int64_t numel, T2VKind kind, void* dst, int rows = 0, int cols = 0) {
  T2VSlot s{key, numel, kind, dst, rows, cols};
  e->slot_index[key] = (int)e->slots.size();
  e->slots.push_back(s);
  if (numel > e->stage_numel) e->stage_numel = numel;
}

}  // namespace

extern "C" {

int latte_t2v_create(const latte_t2v_config_t* cfg, int max_batch, latte_t2v_t** out) {
  if (!cfg || !out || max_batch <= 0) return fail(LATTE_ERR_INVALID, "t2v_create: bad arguments");
  const auto& c = *cfg;
  const int D = c.num_attention_heads * c.attention_head_dim;
  if (c.attention_head_dim != 64 && c.attention_head_dim != 72) return fail(LATTE_ERR_INVALID, "t2v_create: attention_head_dim must be 64 or 72");
  if (D % 128 != 0 || D > 1152) return fail(LATTE_ERR_INVALID, "t2v_create: heads * head_dim must be a multiple of 128, at most 1152");
  if (c.cross_attention_dim != D) return fail(LATTE_ERR_INVALID, "t2v_create: cross_attention_dim must equal the inner dimension (caption projection output)");
  if (c.caption_channels <= 0 || c.caption_channels % 64) return fail(LATTE_ERR_INVALID, "t2v_create: caption_channels must be a multiple of 64");
  if (c.patch_size <= 0 || c.sample_size % c.patch_size) return fail(LATTE_ERR_INVALID, "t2v_create: sample_size % patch_size != 0");
  if (c.in_channels != 4) return fail(LATTE_ERR_INVALID, "t2v_create: in_channels must be 4");
  if (c.num_layers <= 0 || c.video_length <= 0 || c.max_text_tokens <= 0) return fail(LATTE_ERR_INVALID, "t2v_create: bad sizes");
  // f16 operands only: the type the reference runs this transformer in (sample_t2x.py:29 `.to(device, dtype=torch.float16)`).  With
  // bf16 operands the guidance pair (scale 7.5, pipeline_latte.py:752-754) amplifies the 2^-9 operand roundoff past the 1e-3
  // parity bar on every stress case (measured < 3e-3), so the type is not offered.
  if (c.compute_dtype != LATTE_DTYPE_F16) return fail(LATTE_ERR_INVALID, "t2v_create: LatteT2V runs with f16 MFMA operands only (LATTE_DTYPE_F16)");

  auto* e = new latte_t2v();
  e->cfg = c;
  e->max_batch = max_batch;
  e->D = D;
  e->heads = c.num_attention_heads;
  e->hd = c.attention_head_dim;
  e->G = c.sample_size / c.patch_size;
  e->T = e->G * e->G;
  e->F = c.video_length;
  e->H = c.sample_size;
  e->P = c.patch_size * c.patch_size * c.out_channels;
  e->KPE = c.in_channels * c.patch_size * c.patch_size;
  e->Hm = 4 * D;
  e->L = c.num_layers;
  e->nblk = 2 * c.num_layers;
  e->Cc = c.caption_channels;
  e->maxk = c.max_text_tokens;
  e->rows_pad = ((int64_t)max_batch * e->F * e->T + 255) / 256 * 256;
  e->trows_pad = ((int64_t)max_batch * e->maxk + 255) / 256 * 256;
  int rc = LATTE_OK;
#define TRY(x) do { if ((rc = (x))) { latte_t2v_destroy(e); return rc; } } while (0)
  TRY(t2v_alloc(e, &e->tables, (size_t)e->nblk * 6 * D));
  TRY(t2v_alloc(e, &e->head_table, (size_t)2 * D));
  TRY(t2v_alloc(e, &e->pos, (size_t)e->T * D));
  TRY(t2v_alloc(e, &e->temp, (size_t)e->F * D));
  TRY(t2v_alloc(e, &e->pe_wt, (size_t)e->KPE * D));
  TRY(t2v_alloc(e, &e->pe_b, (size_t)D));
  TRY(t2v_alloc(e, &e->t1_w, (size_t)D * 256));
  TRY(t2v_alloc(e, &e->t1_b, (size_t)D));
  TRY(t2v_alloc(e, &e->t2_w, (size_t)D * D));
  TRY(t2v_alloc(e, &e->t2_b, (size_t)D));
  TRY(t2v_alloc(e, &e->ada_w, (size_t)6 * D * D));
  TRY(t2v_alloc(e, &e->ada_b, (size_t)6 * D));
  TRY(t2v_alloc(e, &e->fin_wt, (size_t)D * e->P));
  TRY(t2v_alloc(e, &e->fin_b, (size_t)e->P));
  TRY(t2v_alloc(e, &e->cap1_w, (size_t)D * e->Cc));
  TRY(t2v_alloc(e, &e->cap1_b, (size_t)D));
  TRY(t2v_alloc(e, &e->cap2_w, (size_t)D * D));
  TRY(t2v_alloc(e, &e->cap2_b, (size_t)D));
  TRY(t2v_alloc(e, &e->ones, (size_t)D));
  TRY(launch_fill_f32(e->ones, 1.0f, (size_t)D, nullptr));

  t2v_slot(e, "scale_shift_table", 2 * D, TK_F32, e->head_table);
  t2v_slot(e, "pos_embed.proj.weight", (int64_t)D * e->KPE, TK_F32_TRANSPOSE, e->pe_wt, D, e->KPE);
  t2v_slot(e, "pos_embed.proj.bias", D, TK_F32, e->pe_b);
  t2v_slot(e, "adaln_single.emb.timestep_embedder.linear_1.weight", (int64_t)D * 256, TK_F32, e->t1_w);
  t2v_slot(e, "adaln_single.emb.timestep_embedder.linear_1.bias", D, TK_F32, e->t1_b);
  t2v_slot(e, "adaln_single.emb.timestep_embedder.linear_2.weight", (int64_t)D * D, TK_F32, e->t2_w);
  t2v_slot(e, "adaln_single.emb.timestep_embedder.linear_2.bias", D, TK_F32, e->t2_b);
  t2v_slot(e, "adaln_single.linear.weight", (int64_t)6 * D * D, TK_F32, e->ada_w);
  t2v_slot(e, "adaln_single.linear.bias", 6 * D, TK_F32, e->ada_b);
  t2v_slot(e, "caption_projection.linear_1.weight", (int64_t)D * e->Cc, TK_H16, e->cap1_w);
  t2v_slot(e, "caption_projection.linear_1.bias", D, TK_F32, e->cap1_b);
  t2v_slot(e, "caption_projection.linear_2.weight", (int64_t)D * D, TK_H16, e->cap2_w);
  t2v_slot(e, "caption_projection.linear_2.bias", D, TK_F32, e->cap2_b);
  t2v_slot(e, "caption_projection.y_embedding", (int64_t)120 * e->Cc, TK_SKIP, nullptr);   // null-caption buffer, unused at inference
  t2v_slot(e, "proj_out.weight", (int64_t)e->P * D, TK_F32_TRANSPOSE, e->fin_wt, e->P, D);
  t2v_slot(e, "proj_out.bias", e->P, TK_F32, e->fin_b);

  e->blocks.resize(e->nblk);
  for (int i = 0; i < e->nblk; ++i) {
    T2VBlock& w = e->blocks[i];
    const bool spatial = (i % 2) == 0;
    const std::string p = std::string(spatial ? "transformer_blocks." : "temporal_transformer_blocks.") + std::to_string(i / 2) + ".";
    TRY(t2v_alloc(e, &w.qkv_w, (size_t)3 * D * D));
    TRY(t2v_alloc(e, &w.qkv_b, (size_t)3 * D));
    TRY(t2v_alloc(e, &w.o_w, (size_t)D * D));
    TRY(t2v_alloc(e, &w.o_b, (size_t)D));
    TRY(t2v_alloc(e, &w.fc1_w, (size_t)e->Hm * D));
    TRY(t2v_alloc(e, &w.fc1_b, (size_t)e->Hm));
    TRY(t2v_alloc(e, &w.fc2_w, (size_t)D * e->Hm));
    TRY(t2v_alloc(e, &w.fc2_b, (size_t)D));
    t2v_slot(e, p + "scale_shift_table", 6 * D, TK_F32, e->tables + (size_t)i * 6 * D);
    // to_q | to_k | to_v concatenated: one fused GEMM, columns ordered [3][heads][hd] like latte.py:50
    t2v_slot(e, p + "attn1.to_q.weight", (int64_t)D * D, TK_H16, w.qkv_w);
    t2v_slot(e, p + "attn1.to_k.weight", (int64_t)D * D, TK_H16, w.qkv_w + (size_t)D * D);
    t2v_slot(e, p + "attn1.to_v.weight", (int64_t)D * D, TK_H16, w.qkv_w + (size_t)2 * D * D);
    t2v_slot(e, p + "attn1.to_q.bias", D, TK_F32, w.qkv_b);
    t2v_slot(e, p + "attn1.to_k.bias", D, TK_F32, w.qkv_b + D);
    t2v_slot(e, p + "attn1.to_v.bias", D, TK_F32, w.qkv_b + 2 * D);
    t2v_slot(e, p + "attn1.to_out.0.weight", (int64_t)D * D, TK_H16, w.o_w);
    t2v_slot(e, p + "attn1.to_out.0.bias", D, TK_F32, w.o_b);
    if (spatial) {
      TRY(t2v_alloc(e, &w.q2_w, (size_t)D * D));
      TRY(t2v_alloc(e, &w.q2_b, (size_t)D));
      TRY(t2v_alloc(e, &w.kv2_w, (size_t)2 * D * D));
      TRY(t2v_alloc(e, &w.kv2_b, (size_t)2 * D));
      TRY(t2v_alloc(e, &w.o2_w, (size_t)D * D));
      TRY(t2v_alloc(e, &w.o2_b, (size_t)D));
      t2v_slot(e, p + "attn2.to_q.weight", (int64_t)D * D, TK_H16, w.q2_w);
      t2v_slot(e, p + "attn2.to_q.bias", D, TK_F32, w.q2_b);
      t2v_slot(e, p + "attn2.to_k.weight", (int64_t)D * D, TK_H16, w.kv2_w);
      t2v_slot(e, p + "attn2.to_v.weight", (int64_t)D * D, TK_H16, w.kv2_w + (size_t)D * D);
      t2v_slot(e, p + "attn2.to_k.bias", D, TK_F32, w.kv2_b);
      t2v_slot(e, p + "attn2.to_v.bias", D, TK_F32, w.kv2_b + D);
      t2v_slot(e, p + "attn2.to_out.0.weight", (int64_t)D * D, TK_H16, w.o2_w);
      t2v_slot(e, p + "attn2.to_out.0.bias", D, TK_F32, w.o2_b);
    }
    t2v_slot(e, p + "ff.net.0.proj.weight", (int64_t)e->Hm * D, TK_H16, w.fc1_w);
    t2v_slot(e, p + "ff.net.0.proj.bias", e->Hm, TK_F32, w.fc1_b);
    t2v_slot(e, p + "ff.net.2.weight", (int64_t)D * e->Hm, TK_H16, w.fc2_w);
    t2v_slot(e, p + "ff.net.2.bias", D, TK_F32, w.fc2_b);
  }

  // fixed tables (non-persistent buffers of the reference: PatchEmbed.pos_embed, temp_pos_embed, latte_t2v.py:571-581,670-671)
  {
    const int G = e->G;
    const double base_size = (double)(c.sample_size / c.patch_size), interp = (double)std::max(c.sample_size / 64, 1);
    std::vector<double> gw((size_t)G * G), gh((size_t)G * G), ew, eh;
    for (int i = 0; i < G; ++i)        // meshgrid(grid_w, grid_h): grid[0][i][j] = w_j, grid[1][i][j] = h_i (float32 positions)
      for (int j = 0; j < G; ++j) {
        gw[(size_t)i * G + j] = (double)((float)j / (float)((double)G / base_size) / (float)interp);
        gh[(size_t)i * G + j] = (double)((float)i / (float)((double)G / base_size) / (float)interp);
      }
    sincos_1d(D / 2, gw, ew);
    sincos_1d(D / 2, gh, eh);
    std::vector<float> pos((size_t)e->T * D);
    for (int t = 0; t < e->T; ++t)
      for (int d = 0; d < D / 2; ++d) {
        pos[(size_t)t * D + d] = (float)ew[(size_t)t * (D / 2) + d];
        pos[(size_t)t * D + D / 2 + d] = (float)eh[(size_t)t * (D / 2) + d];
      }
    LATTE_HIP(hipMemcpy(e->pos, pos.data(), pos.size() * sizeof(float), hipMemcpyHostToDevice));
    std::vector<double> fp(e->F), et;
    for (int f = 0; f < e->F; ++f) fp[f] = (double)f;
    sincos_1d(D, fp, et);
    std::vector<float> tf(et.begin(), et.end());
    LATTE_HIP(hipMemcpy(e->temp, tf.data(), tf.size() * sizeof(float), hipMemcpyHostToDevice));
  }

  const int64_t R = e->rows_pad, TR = e->trows_pad;
  TRY(t2v_alloc(e, &e->stage, (size_t)e->stage_numel));
  TRY(t2v_alloc(e, &e->xin, (size_t)max_batch * e->F * c.in_channels * e->H * e->H));
  TRY(t2v_alloc(e, &e->out_bf, (size_t)max_batch * e->F * c.out_channels * e->H * e->H));
  TRY(t2v_alloc(e, &e->xres, (size_t)R * D));
  TRY(t2v_alloc(e, &e->xn, (size_t)R * D));
  TRY(t2v_alloc(e, &e->qkv, (size_t)R * 3 * D));
  TRY(t2v_alloc(e, &e->hbuf, (size_t)R * e->Hm));
  TRY(t2v_alloc(e, &e->ctx_in, (size_t)TR * e->Cc));
  TRY(t2v_alloc(e, &e->ctx1, (size_t)TR * D));
  TRY(t2v_alloc(e, &e->ctx, (size_t)TR * D));
  TRY(t2v_alloc(e, &e->kv, (size_t)TR * 2 * D));
  TRY(t2v_alloc(e, &e->kv_all, (size_t)c.num_layers * TR * 2 * D));
  TRY(t2v_alloc(e, &e->tsteps, (size_t)1024));
  e->tsteps_cap = 1024;
  TRY(t2v_alloc(e, &e->kbias, (size_t)max_batch * e->maxk));
  TRY(t2v_alloc(e, &e->temb0, (size_t)max_batch * D));
  TRY(t2v_alloc(e, &e->temb, (size_t)max_batch * D));
  TRY(t2v_alloc(e, &e->t6, (size_t)max_batch * 6 * D));
  TRY(t2v_alloc(e, &e->mod, (size_t)max_batch * (6 * e->nblk + 2) * D));
#undef TRY
  LATTE_HIP(hipDeviceSynchronize());
  *out = e;
  return LATTE_OK;
}

void latte_t2v_destroy(latte_t2v_t* e) {
  if (!e) return;
  for (void* p : e->allocs) (void)hipFree(p);
  delete e;
}

int latte_t2v_num_keys(const latte_t2v_t* e) { return e ? (int)e->slots.size() : 0; }
const char* latte_t2v_key(const latte_t2v_t* e, int i) {
  if (!e || i < 0 || i >= (int)e->slots.size()) return nullptr;
  return e->slots[i].key.c_str();
}

int latte_t2v_load_tensor(latte_t2v_t* e, const char* key, const float* data, int64_t numel, int on_device, void* stream) {
  if (!e || !key || !data) return fail(LATTE_ERR_INVALID, "t2v_load_tensor: null argument");
  auto it = e->slot_index.find(key);
  if (it == e->slot_index.end()) return fail(LATTE_ERR_INVALID, std::string("t2v_load_tensor: unexpected key '") + key + "'");
  T2VSlot& s = e->slots[it->second];
  if (s.kind == TK_SKIP) { s.loaded = true; return LATTE_OK; }
  if (numel != s.numel)
    return fail(LATTE_ERR_INVALID, std::string("t2v_load_tensor: size mismatch for '") + key + "': got " + std::to_string(numel) +
                                       ", expected " + std::to_string(s.numel));
  hipStream_t st = (hipStream_t)stream;
  const float* src = data;
  if (!on_device) {
    LATTE_HIP(hipMemcpyAsync(e->stage, data, sizeof(float) * numel, hipMemcpyHostToDevice, st));
    src = e->stage;
  }
  int rc = LATTE_OK;
  switch (s.kind) {
    case TK_F32: LATTE_HIP(hipMemcpyAsync(s.dst, src, sizeof(float) * numel, hipMemcpyDeviceToDevice, st)); break;
    case TK_F32_TRANSPOSE: rc = launch_transpose_f32(src, (float*)s.dst, s.rows, s.cols, st); break;
    case TK_H16: rc = launch_convert_f32_to_h16(src, (half_t*)s.dst, numel, e->cfg.compute_dtype, st); break;
    default: break;
  }
  if (rc) return rc;
  if (!on_device) LATTE_HIP(hipStreamSynchronize(st));
  s.loaded = true;
  e->txt_batch = 0;   // an installed text context was projected with the old weights
  return LATTE_OK;
}

int latte_t2v_set_option(latte_t2v_t* e, const char* name, int64_t value) {
  if (!e || !name) return fail(LATTE_ERR_INVALID, "t2v_set_option: null argument");
  if (std::string(name) == "fuse_qkv_attn") {
    if (value < 0 || value > 31) return fail(LATTE_ERR_INVALID, "fuse_qkv_attn: bit 0 = spatial blocks, bit 1 = temporal blocks, bits 2-3 = schedule variant");
    e->fuse_qkv_attn = (int)value;
    return LATTE_OK;
  }
  return fail(LATTE_ERR_INVALID, std::string("t2v_set_option: unknown option '") + name + "'");
}

int latte_t2v_check_weights(latte_t2v_t* e) {
  if (!e) return fail(LATTE_ERR_INVALID, "t2v_check_weights: null engine");
  for (const auto& s : e->slots)
    if (!s.loaded && s.kind != TK_SKIP) return fail(LATTE_ERR_STATE, "Missing key(s) in state_dict: \"" + s.key + "\"");
  return LATTE_OK;
}

// Text context: caption projection (latte_t2v.py:781-793: Linear -> GELU(tanh) -> Linear on [B * Lk, caption_channels]),
// the k|v projection of every spatial block's cross-attention on it, and the additive score bias of the padded tokens
// ((1 - mask) * -10000, :746-749).  All of it depends on the text only.
static int t2v_text_context(latte_t2v* e, const float* enc, const float* mask, int B, int Lk, hipStream_t st) {
  const auto& c = e->cfg;
  const int D = e->D, dt = c.compute_dtype, MT = B * Lk;
  int rc;
  GemmArgs g{};
  g.rows_per_sample = MT; g.gate_stride = 0;
  if ((rc = launch_convert_f32_to_h16(enc, e->ctx_in, (int64_t)MT * e->Cc, dt, st))) return rc;
  g.M = MT; g.A = e->ctx_in; g.W = e->cap1_w; g.bias = e->cap1_b; g.out = e->ctx1; g.N = D; g.K = e->Cc;
  if ((rc = launch_gemm(g, EPI_BIAS_GELU_H16, dt, 0, st))) return rc;
  g.A = e->ctx1; g.W = e->cap2_w; g.bias = e->cap2_b; g.out = e->ctx; g.K = D;
  if ((rc = launch_gemm(g, EPI_BIAS_H16, dt, 0, st))) return rc;
  for (int l = 0; l < e->L; ++l) {
    const T2VBlock& w = e->blocks[2 * l];
    GemmArgs gk{};
    gk.M = MT; gk.rows_per_sample = MT; gk.A = e->ctx; gk.W = w.kv2_w; gk.bias = w.kv2_b;
    gk.out = e->kv_all + (size_t)l * e->trows_pad * 2 * D; gk.N = 2 * D; gk.K = D;
    if ((rc = launch_gemm(gk, EPI_BIAS_H16, dt, 0, st))) return rc;
  }
  e->txt_has_mask = mask != nullptr;
  if (mask && (rc = launch_mask_bias(mask, e->kbias, (size_t)MT, st))) return rc;
  e->txt_batch = B;
  e->txt_nk = Lk;
  return LATTE_OK;
}

// The denoiser on an installed text context.  x: [xb, C, F, H, W] with xb = B, or B / 2 when `dup` (guidance pair: both
// halves of the batch see the same latents, pipeline_latte.py:725); t: device int64, one per sample or ONE shared by all
// (t_shared: the adaLN-single rows are then computed once, row stride 0).  Result: e->out_bf, frame layout [B * F, Cout, H, W].
static int t2v_core(latte_t2v* e, const float* x, const int64_t* t, bool t_shared, int B, bool dup, int enable_temporal,
                    hipStream_t st) {
  const auto& c = e->cfg;
  const int D = e->D, T = e->T, F = e->F, dt = c.compute_dtype, Lk = e->txt_nk;
  const int M = B * F * T, rps = F * T;
  const int nt = t_shared ? 1 : B;
  const int mstride = t_shared ? 0 : (6 * e->nblk + 2) * D;
  const float* kbias = e->txt_has_mask ? e->kbias : nullptr;
  int rc;
  // ---- conditioning: embedded timestep, its 6D projection, all adaLN-single rows (latte_t2v.py:398-428,775-779)
  if ((rc = launch_small_linear(IN_TFREQ, nullptr, t, e->t1_w, e->t1_b, nullptr, nullptr, e->temb0, nt, D, 256, D, st))) return rc;
  if ((rc = launch_small_linear(IN_SILU, e->temb0, nullptr, e->t2_w, e->t2_b, nullptr, nullptr, e->temb, nt, D, D, D, st))) return rc;
  if ((rc = launch_small_linear(IN_SILU, e->temb, nullptr, e->ada_w, e->ada_b, nullptr, nullptr, e->t6, nt, 6 * D, D, 6 * D, st))) return rc;
  if ((rc = launch_adaln_single(e->tables, e->head_table, e->t6, e->temb, e->mod, nt, e->nblk, D, st))) return rc;
  // ---- [xb, C, F, H, W] -> frames, patch embed + positions (twice for the guidance pair)
  const int xb = dup ? B / 2 : B;
  if ((rc = launch_permute_cf(x, e->xin, xb, c.in_channels, F, e->H * e->H, 1, st))) return rc;
  if ((rc = launch_patch_embed(e->xin, e->pe_wt, e->pe_b, e->pos, e->xres, xb * F, c.in_channels, e->H, c.patch_size, D, st))) return rc;
  if (dup && (rc = launch_patch_embed(e->xin, e->pe_wt, e->pe_b, e->pos, e->xres + (size_t)xb * rps * D, xb * F, c.in_channels,
                                      e->H, c.patch_size, D, st))) return rc;

  GemmArgs g{};
  for (int i = 0; i < e->nblk; ++i) {
    const bool spatial = (i % 2) == 0;
    if (!spatial && !enable_temporal) continue;
    const T2VBlock& w = e->blocks[i];
    const float* mb = e->mod + (size_t)i * 6 * D;   // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
    const float* te = (i == 1 && F > 1) ? e->temp : nullptr;                  // latte_t2v.py:889-890
    if ((rc = launch_ln_modulate(e->xres, e->xres, e->xn, mb, mb + D, mstride, M, D, rps, te, T, F, dt, st))) return rc;
    g = GemmArgs{};
    g.M = M; g.rows_per_sample = rps; g.gate_stride = mstride;
    const half_t* attn_out = e->xn;
    const float attn_scale = 1.0f / std::sqrt((float)e->hd);
    if (((e->fuse_qkv_attn >> (spatial ? 0 : 1)) & 1) && qkv_attention_fusable(D, e->heads, e->hd, F, T, spatial ? 0 : 1, M)) {
      // to_q | to_k | to_v + the attention core in one kernel, q / k / v only in LDS (Latte-1: the temporal blocks, 16 frames per
      // token; the 1024-token spatial sequences keep the separate kernels); output to the idle qkv buffer viewed as [rows, D]
      QkvAttnArgs qa{};
      qa.xn = e->xn; qa.w = w.qkv_w; qa.bias = w.qkv_b; qa.out = e->qkv; qa.B = B; qa.F = F; qa.T = T; qa.D = D;
      qa.heads = e->heads; qa.hd = e->hd; qa.mode = spatial ? 0 : 1; qa.scale = attn_scale; qa.flags = ((e->fuse_qkv_attn >> 2) & 7) ^ 7;
      if ((rc = launch_qkv_attention(qa, dt, st))) return rc;
      attn_out = e->qkv;
    } else {
      g.A = e->xn; g.W = w.qkv_w; g.bias = w.qkv_b; g.out = e->qkv; g.N = 3 * D; g.K = D;
      if ((rc = launch_gemm(g, EPI_BIAS_H16, dt, 0, st))) return rc;
      AttnArgs a{};
      a.qkv = e->qkv; a.out = e->xn; a.heads = e->heads; a.hd = e->hd; a.D = D;
      a.sample_stride = rps; a.scale = attn_scale;
      if (spatial) { a.num_seq = B * F; a.L = T; a.U = F; a.seq_stride = T; a.row_stride = 1; }
      else         { a.num_seq = B * T; a.L = F; a.U = T; a.seq_stride = 1; a.row_stride = T; }
      if ((rc = launch_attention(a, dt, st))) return rc;
    }
    g.A = attn_out; g.W = w.o_w; g.bias = w.o_b; g.out = e->xres; g.gate = mb + 2 * D; g.N = D; g.K = D; g.tag = 0;
    if ((rc = launch_gemm(g, EPI_GATE_RES_F32, dt, 0, st))) return rc;
    if (spatial) {
      // cross-attention on the UN-normalised stream (PixArt: no norm before attn2, no gate after it)
      if ((rc = launch_convert_f32_to_h16(e->xres, e->xn, (int64_t)M * D, dt, st))) return rc;
      g.A = e->xn; g.W = w.q2_w; g.bias = w.q2_b; g.out = e->qkv; g.gate = nullptr; g.N = D; g.K = D;
      if ((rc = launch_gemm(g, EPI_BIAS_H16, dt, 0, st))) return rc;
      AttnArgs x2{};
      x2.qkv = e->qkv; x2.q_ld = D; x2.kv = e->kv_all + (size_t)(i / 2) * e->trows_pad * 2 * D; x2.kbias = kbias; x2.Lk = Lk;
      x2.out = e->xn;
      x2.heads = e->heads; x2.hd = e->hd; x2.D = D; x2.sample_stride = rps; x2.scale = attn_scale;
      x2.num_seq = B * F; x2.L = T; x2.U = F; x2.seq_stride = T; x2.row_stride = 1;
      if ((rc = launch_cross_attention(x2, dt, st))) return rc;
      g.A = e->xn; g.W = w.o2_w; g.bias = w.o2_b; g.out = e->xres; g.gate = e->ones; g.gate_stride = 0; g.N = D; g.K = D;
      if ((rc = launch_gemm(g, EPI_GATE_RES_F32, dt, 0, st))) return rc;
      g.gate_stride = mstride;
    }
    if ((rc = launch_ln_modulate(e->xres, e->xres, e->xn, mb + 3 * D, mb + 4 * D, mstride, M, D, rps, nullptr, T, F, dt, st))) return rc;
    g.A = e->xn; g.W = w.fc1_w; g.bias = w.fc1_b; g.out = e->hbuf; g.gate = nullptr; g.N = e->Hm; g.K = D;
    if ((rc = launch_gemm(g, EPI_BIAS_GELU_H16, dt, 0, st))) return rc;
    g.A = e->hbuf; g.W = w.fc2_w; g.bias = w.fc2_b; g.out = e->xres; g.gate = mb + 5 * D; g.N = D; g.K = e->Hm; g.tag = 1;
    if ((rc = launch_gemm(g, EPI_GATE_RES_F32, dt, 0, st))) return rc;
  }
  // ---- output head (latte_t2v.py:913-931), frame layout
  const float* hm = e->mod + (size_t)e->nblk * 6 * D;   // chunk(2): shift, scale
  return launch_final_layer(e->xres, hm, hm + D, mstride, e->fin_wt, e->fin_b, e->out_bf, M, D, rps, T, c.patch_size,
                            c.out_channels, e->H, st);
}

int latte_t2v_set_text(latte_t2v_t* e, const float* encoder_hidden_states, const float* encoder_attention_mask, int batch,
                       int n_text, void* stream) {
  if (!e) return fail(LATTE_ERR_INVALID, "t2v_set_text: null engine");
  if (!encoder_hidden_states) {   // uninstall
    e->txt_batch = 0;
    return LATTE_OK;
  }
  int rc = latte_t2v_check_weights(e);
  if (rc) return rc;
  if (batch <= 0 || batch > e->max_batch) return fail(LATTE_ERR_STATE, "t2v_set_text: batch exceeds max_batch of the engine");
  if (n_text <= 0 || n_text > e->maxk) return fail(LATTE_ERR_STATE, "t2v_set_text: more text tokens than max_text_tokens");
  return t2v_text_context(e, encoder_hidden_states, encoder_attention_mask, batch, n_text, (hipStream_t)stream);
}

int latte_t2v_forward(latte_t2v_t* e, const float* x, const int64_t* t, const float* encoder_hidden_states,
                      const float* encoder_attention_mask, int batch, int n_text, int enable_temporal_attentions, float* out,
                      void* stream) {
  if (!e || !x || !t || !out) return fail(LATTE_ERR_INVALID, "t2v_forward: null argument");
  int rc = latte_t2v_check_weights(e);
  if (rc) return rc;
  if (batch <= 0 || batch > e->max_batch) return fail(LATTE_ERR_STATE, "t2v_forward: batch exceeds max_batch of the engine");
  hipStream_t st = (hipStream_t)stream;
  if (encoder_hidden_states) {
    if (n_text <= 0 || n_text > e->maxk) return fail(LATTE_ERR_STATE, "t2v_forward: more text tokens than max_text_tokens");
    if ((rc = t2v_text_context(e, encoder_hidden_states, encoder_attention_mask, batch, n_text, st))) return rc;
  } else if (e->txt_batch != batch) {
    return fail(LATTE_ERR_STATE, "t2v_forward: no encoder_hidden_states and no installed text context for this batch (latte_t2v_set_text)");
  }
  if ((rc = t2v_core(e, x, t, false, batch, false, enable_temporal_attentions, st))) return rc;
  const auto& c = e->cfg;
  return launch_permute_cf(e->out_bf, out, batch, c.out_channels, e->F, e->H * e->H, 0, st);   // back to [B, C, F, H, W]
}

int latte_t2v_guided_ddim_loop(latte_t2v_t* e, float* x, int samples, int n_steps, const int64_t* timesteps,
                               const double* alpha_t, const double* alpha_prev, float guidance_scale,
                               int enable_temporal_attentions, void* stream) {
  if (!e || !x || !timesteps || !alpha_t || !alpha_prev || samples <= 0 || n_steps <= 0)
    return fail(LATTE_ERR_INVALID, "t2v_guided_ddim_loop: bad arguments");
  int rc = latte_t2v_check_weights(e);
  if (rc) return rc;
  const int B = 2 * samples;
  if (B > e->max_batch) return fail(LATTE_ERR_STATE, "t2v_guided_ddim_loop: the guidance pair exceeds max_batch of the engine");
  if (e->txt_batch != B)
    return fail(LATTE_ERR_STATE, "t2v_guided_ddim_loop: install the text context of the guidance pair first "
                                 "(latte_t2v_set_text with [negative | prompt] embeddings, 2 * samples rows)");
  const auto& c = e->cfg;
  if (c.out_channels != c.in_channels && c.out_channels != 2 * c.in_channels)
    return fail(LATTE_ERR_INVALID, "t2v_guided_ddim_loop: out_channels must be C or 2C (learned sigma)");
  hipStream_t st = (hipStream_t)stream;
  if (n_steps > e->tsteps_cap) return fail(LATTE_ERR_INVALID, "t2v_guided_ddim_loop: more than 1024 steps");
  LATTE_HIP(hipMemcpyAsync(e->tsteps, timesteps, sizeof(int64_t) * n_steps, hipMemcpyHostToDevice, st));
  LATTE_HIP(hipStreamSynchronize(st));   // the caller's host array may go away
  for (int k = 0; k < n_steps; ++k) {
    if ((rc = t2v_core(e, x, e->tsteps + k, true, B, true, enable_temporal_attentions, st))) return rc;
    // diffusers DDIMScheduler.step, eta = 0, no clipping: Python-float coefficients times fp32 tensors
    const double at = alpha_t[k], ap = alpha_prev[k];
    if (!(at > 0.0 && at < 1.0) || !(ap > 0.0 && ap <= 1.0)) return fail(LATTE_ERR_INVALID, "t2v_guided_ddim_loop: alpha out of range");
    if ((rc = launch_t2v_guided_ddim(x, e->out_bf, samples, c.in_channels, c.out_channels, e->F, e->H * e->H, guidance_scale,
                                     (float)std::sqrt(1.0 - at), (float)std::sqrt(at), (float)std::sqrt(ap),
                                     (float)std::sqrt(1.0 - ap), st))) return rc;
  }
  return LATTE_OK;
}

}  // extern "C"
