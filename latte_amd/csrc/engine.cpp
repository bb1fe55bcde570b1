// Engine host logic: weight ingestion, workspace, the denoiser forward and the fused sampling loop.
//
//   Latte.forward / forward_with_cfg      /root/reference/models/latte.py:314-398
//   p_sample_loop / ddim_sample_loop      /root/reference/diffusion/gaussian_diffusion.py:423-515,604-684
//   _WrappedModel (index -> timestep)     /root/reference/diffusion/respace.py:125-130
//
// Data layout in HBM (one canonical token order, no transposes between spatial and temporal blocks):
//   residual stream  xres  fp32 [B*F*T (padded to 256), D]   row = (b*F + f)*T + t
//   GEMM operands    xn    half [rows, D], qkv half [rows, 3D], h half [rows, mlp]
//   conditioning     mod   fp32 [B, depth*6D + 2D]  (all adaLN outputs of one forward, computed once per
//                    SAMPLE; the reference recomputes them on F or T repeated rows, latte.py:333-339)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace latte {

static thread_local std::string g_last_error;
static std::atomic<int> g_debug_choice[DBG_NUM_CHOICES];
int debug_choice(DebugChoice c) { return g_debug_choice[c].load(std::memory_order_relaxed); }
int set_debug_choice(const char* name, int value) {
  static const char* const names[DBG_NUM_CHOICES] = {"attn_variant", "xattn_flash", "tn_kernel", "tn_wn", "attn_bwd_tiles", "conv_kernel", "vae_split"};
  static const int allowed[DBG_NUM_CHOICES][4] = {{1, 5, 12, 13}, {1, 0, 0, 0}, {4, 0, 0, 0}, {4, 0, 0, 0}, {1, 2, 0, 0}, {1, 2, 3, 4}, {0, 0, 0, 0}};
  for (int i = 0; i < DBG_NUM_CHOICES; ++i) {
    if (name && std::string(name) == names[i]) {
      bool ok = value == 0;
      for (int k = 0; k < 4; ++k) ok = ok || (allowed[i][k] != 0 && value == allowed[i][k]);
      ok = ok || (i == DBG_VAE_SPLIT && (value >> 24) == 1);   // (1 << 24) | pass mask (vae_engine.cpp: vae_split_mask)
#ifdef LATTE_GEMM_ABLATE
      ok = ok || (i == DBG_ATTN_VARIANT && value >= 7 && value <= 19);   // measurement build: attn_stream ablations (results garbage)
#endif
      if (!ok) return -1;
      g_debug_choice[i].store(value, std::memory_order_relaxed);
      return 0;
    }
  }
  return -1;
}
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

enum PackKind { PK_F32, PK_F32_TRANSPOSE, PK_H16 };
struct TensorSlot {
  std::string key;
  int64_t numel;
  PackKind kind;
  void* dst;
  int rows, cols;  // for PK_F32_TRANSPOSE: source [rows][cols]
  bool loaded = false;
};

struct BlockW {
  half_t *qkv_w, *proj_w, *fc1_w, *fc2_w;
  float *qkv_b, *proj_b, *fc1_b, *fc2_b;
  half_t *proj_w2 = nullptr, *fc1_w2 = nullptr;   // [N, 2 K] = [W | W]: the weight of a split-operand linear (guided_split bits 0 / 1), built lazily
  unsigned char *fc1_w4 = nullptr, *fc1_w4s = nullptr;    // fp4 image of fc1's weight + its row scales (guided_split bit 4; GemmArgs::W4 / W4s)
  unsigned char *proj_w8 = nullptr, *fc1_w8 = nullptr;   // [N, K] e4m3(W 2^LO8_W_SHIFT): the weight of a GEMM's fp8 correction pass (bits 2 / 3)
};

struct Prof {
  std::vector<hipEvent_t> ev;
  std::vector<int> cls;
};

}  // namespace latte

using namespace latte;

struct latte_engine {
  latte_model_config_t cfg;
  int max_batch = 0;
  int D = 0, T = 0, F = 0, G = 0, Cin = 0, Cout = 0, H = 0, P = 0, KPE = 0, Hm = 0, hd = 0;
  int nmod = 0;           // depth*6D + 2D
  int64_t rows_max = 0, rows_pad = 0;
  int gemm_variant = 0;  // 0 = per-shape choice (gemm_auto_variant)
  int gemm_variant_of[4] = {0, 0, 0, 0};   // per-GEMM override (qkv, proj, fc1, fc2); 0 = gemm_variant
  int fuse_qkv_attn = 3;                   // bit 0: spatial blocks, bit 1: temporal blocks run qkv projection + attention as ONE kernel
                                           // (qkv_attn.hip) wherever the shape allows it (T == 256 / F == 16); 0 = the un-fused pair;
                                           // bits 2, 3: QkvAttnArgs::flags (schedule variants, same results)
  int gated_split_k = 0;                   // gated GEMMs of small batches: 0 = rule of gated_gemm, 1 = never split, 2..4 = force
  // Guided calls (forward_with_cfg, latte.py:379-398): eps_u + s (eps_c - eps_u) amplifies the part of the operand rounding that differs
  // between the two halves (s = 7: 7 c - 6 u).  At trained-scale gates the f16 forward of XL/2 then sits AT north_star's 1e-3 (0.6 - 1.2e-3
  // over gate_std x timestep x seed, two of twelve draws above: profiles/r5_gate_parity_guided.json); the per-rounding-point attribution
  // on CPU (oracle/emulate_operands.py) names the operands of the out-projection (attention output) and of fc1 / fc2.  guided_split:
  // bit 0 = the attention output, bit 1 = the LayerNorm-modulate output in front of fc1 are carried as SPLIT pairs [hi | lo] (two halves
  // per value, ~22 mantissa bits) against weights stored [W | W] -- the same GEMM kernels on K' = 2 K, no rounding of that operand.
  // Round 6: bit 2 / bit 3 = the same two operands with the remainder in FP8 (e4m3 of lo 2^12, one byte per value) and the product
  // lo . W8^T collected by a block-scaled fp8 MFMA pass behind the f16 K loop of the same GEMM launch (GemmArgs::A8, gemm_pw.hip):
  // half the MFMA time and a quarter of the operand bytes of the [hi | lo] . [W | W] form; the remainder term is 2^-12 of the product,
  // so its own fp8 rounding (2^-4 relative) is 2^-16 -- below the weight operand's f16 rounding.  A bit-2/3 setting wins over bit 0/1
  // for its operand.  Guided calls of f16 engines only (bf16 cannot reach 1e-3 with or without it: default 0 there); where a shape
  // Round 6, second form: bit 4 = fc1's operand with the remainder in FP4 (e2m1 codes, one E8M0 scale per row: ln_modulate's SPLIT4 output,
  // GemmArgs::A4) -- the block-scaled MFMA runs fp4 x fp4 at twice the fp8 rate and a code row is half the bytes; wins over bits 1 / 3.
  // has no split form (un-fused attention, N % 192, K % 128) that operand stays plain -- latte_engine_get_info("guided_split_active")
  // reports what the last guided forward really ran.
  int guided_split = 20;
  int guided_split_active = 0;             // what the last guided forward used (bits as above)
  bool split_failed = false;               // an allocation for the split operands failed once: guided calls run plain from then on
  bool split_w_ready = false;              // the derived weight copies (proj_w2 / fc1_w2 / proj_w8 / fc1_w8) hold the current weights
  hipEvent_t load_event = nullptr;         // recorded behind every weight conversion on ITS stream: the derived copies wait for it
  half_t* xn2 = nullptr;                   // [rows_pad, 2 D]: split LayerNorm-modulate output (lazily allocated)
  unsigned char* lo8 = nullptr;            // [rows_pad, D] bytes: the fp8 remainder of the attention output, then of fc1's operand
  unsigned char *lo4 = nullptr, *lo4s = nullptr;   // [rows_pad, lo4_pitch(D)] e2m1 codes + [rows_pad] row scales: the fp4 remainder of fc1's operand (bit 4)
  float *split_ws = nullptr, *zero_bias = nullptr;   // partial products of the split gated GEMMs, a zero bias row for them
  std::vector<BlockW> blocks;
  float *ada_w = nullptr, *ada_b = nullptr, *pos = nullptr, *temp = nullptr, *pe_wt = nullptr, *pe_b = nullptr,
        *t0_w = nullptr, *t0_b = nullptr, *t2_w = nullptr, *t2_b = nullptr, *ytab = nullptr, *fin_wt = nullptr,
        *fin_b = nullptr;
  float *xres = nullptr, *mod = nullptr, *temb0 = nullptr, *cvec = nullptr, *model_out = nullptr, *stage = nullptr,
        *noise_buf = nullptr;
  half_t *xn = nullptr, *qkv = nullptr, *hbuf = nullptr;
  int64_t* tmap_dev = nullptr;
  int64_t tmap_cap = 0;
  // extras == 78 (latte.py:238-242): text projection weights, the projected rows of the current text batch, identity
  // row index (the projected rows enter the conditioning exactly where the label-table rows do), t-only vector / rows
  // for the final layer, whose conditioning never includes the text (latte.py:372-373)
  float *txt_w = nullptr, *txt_b = nullptr, *txt_proj = nullptr, *cvec_t = nullptr, *cond_rows_t = nullptr;
  int64_t* iota = nullptr;
  int64_t cond_t_cap = 0;
  int txt_rows = 0;
  // conditioning of a whole chain (latte_sample_loop): timestep-embedding table (own or installed after the RCCL
  // broadcast), per-(step, sample) conditioning rows and all adaLN outputs
  float *temb_table = nullptr, *temb_own = nullptr, *temb_work = nullptr, *cond_rows = nullptr, *mod_all = nullptr;
  int temb_table_n = 0;        // > 0: an installed table of that many respaced steps
  std::vector<int64_t> temb_table_map;   // the timestep_map the installed table was computed for
  std::vector<int64_t> temb_own_map;     // the timestep_map temb_own currently holds (empty: none); a chain run in several
                                         // latte_sample_loop segments computes its table once
  int64_t temb_table_cap = 0, temb_own_cap = 0, temb_cap = 0, cond_cap = 0, mod_all_cap = 0;
  int64_t stage_numel = 0;
  std::vector<TensorSlot> slots;
  std::map<std::string, int> slot_index;
  std::vector<void*> allocs;
  uint64_t seed = 0, rng_offset = 0;
};

namespace {

template <typename Tp>
int dev_alloc(latte_engine* e, Tp** p, size_t count, bool zero = true) {
  void* q = nullptr;
  const size_t bytes = count * sizeof(Tp);
  LATTE_HIP(hipMalloc(&q, bytes ? bytes : 16));
  if (zero) LATTE_HIP(hipMemset(q, 0, bytes ? bytes : 16));
  e->allocs.push_back(q);
  *p = (Tp*)q;
  return LATTE_OK;
}

void add_slot(latte_engine* e, const std::string& key, int64_t numel, PackKind kind, void* dst, int rows = 0, int cols = 0) {
  TensorSlot s;
  s.key = key;
  s.numel = numel;
  s.kind = kind;
  s.dst = dst;
  s.rows = rows;
  s.cols = cols;
  e->slot_index[key] = (int)e->slots.size();
  e->slots.push_back(s);
  if (numel > e->stage_numel) e->stage_numel = numel;
}

struct Timer {  // optional per-launch HIP events (latte_profile_forward)
  Prof* p;
  hipStream_t st;
  void mark(int cls) {
    if (!p) return;
    hipEvent_t ev;
    (void)hipEventCreate(&ev);
    (void)hipEventRecord(ev, st);
    p->ev.push_back(ev);
    p->cls.push_back(cls);
  }
};
enum { C_QKV = 0, C_PROJ, C_FC1, C_FC2, C_ATTN_S, C_ATTN_T, C_LN, C_COND, C_PATCH, C_FINAL, C_QKVATTN_S, C_QKVATTN_T, C_NONE = -1 };

// Gated read-modify-write GEMM x += gate * (A W^T + bias) (attention out-projection, fc2; latte.py:179-180).  At small batches
// its 256 x 192 output tiles fill a fraction of the 256 CUs (B = 1 at XL/2: 96 tiles, fc2 78 us per launch against 33 us at the
// B = 8 rate): when at least two splits of a DEEP contraction fit on the chip, the GEMM runs as `s` partial products (tile
// variant 5, grid.y = s, fp32 slabs in split_ws) and gated_split_reduce adds them to the residual stream in slab order.
// Measured inside the XL/2 forward at B = 1 (same box): fc2 (K = 4608) 77.6 -> 68.2 us including the reduction, the out-projection
// (K = 1152) 30.6 -> 39.0 us -- the reduction pass costs more than half of a 18-K-tile GEMM, hence the >= 32 K tiles per split.
// (A stream-K decomposition of the 12-wave kernel with in-launch hand-over of the partial products was built and measured
// slower than this at every small-batch shape: DESIGN.md section 8.)
constexpr int64_t SPLIT_WS_FLOATS = 256ll * 256 * 192;   // s * tiles <= 256 tiles of 256 x 192
int gated_split_choice(const latte_engine* e, int M, int N, int K, int variant) {
  if (variant != 0 || e->gated_split_k == 1 || N % 192 || K % 64) return 1;
  if (e->gated_split_k == 0 && gemm_small_tile_ok(M, N, K)) return 1;   // one 128 x 144 tile per CU beats the split (launch_gemm)
  const int tiles = ((M + 255) / 256) * (N / 192), nk = K / 64;
  if (e->gated_split_k >= 2) {
    const int s = e->gated_split_k;
    return (nk % s == 0 && (int64_t)s * M * N <= SPLIT_WS_FLOATS) ? s : 1;
  }
  for (int s = 4; s >= 2; --s)
    if (tiles * s <= 256 && nk % s == 0 && nk / s >= 32 && (int64_t)s * M * N <= SPLIT_WS_FLOATS) return s;
  return 1;
}
int gated_gemm(latte_engine* e, const GemmArgs& g, int dt, int variant, hipStream_t st) {
  if (g.A8) return launch_gemm(g, EPI_GATE_RES_F32, dt, 0, st);   // fp8 correction pass: the rolling 12-wave kernel carries it
  const int s = gated_split_choice(e, g.M, g.N, g.K, variant);
  if (s == 1) return launch_gemm(g, EPI_GATE_RES_F32, dt, variant, st);
  GemmArgs p = g;
  p.bias = e->zero_bias;
  p.gate = nullptr;
  p.out = e->split_ws;
  p.k_chunk = g.K / s;
  p.split_stride = (long)g.M * g.N;
  if (int rc = launch_gemm(p, EPI_BIAS_F32, dt, 5, st)) return rc;
  return launch_gated_split_reduce((float*)g.out, e->split_ws, s, (size_t)g.M * g.N, g.bias, g.gate, g.gate_stride,
                                   g.rows_per_sample, g.M, g.N, st);
}

// Derived weights / operand buffers of the split operands of a guided call, for the bits of `need` only (guided_split), each pointer
// guarded on its own.  An allocation failure does not fail the forward: it is reported once on stderr, `split_failed` is set and this
// and every later guided call runs the plain f16 path.  The copies are built on the forward's stream BEHIND the last weight
// conversion (load_event), whatever stream that ran on.  Returns the bits that are usable.
int ensure_split_weights(latte_engine* e, int need, hipStream_t st) {
  const int D = e->D, Hm = e->Hm;
  if (e->split_failed) return 0;
  auto give_up = [&](const char* what) {
    fprintf(stderr, "latte_amd: guided_split: allocation of %s failed (%s); guided calls run with plain f16 operands\n", what, latte_last_error());
    e->split_failed = true;
    return 0;
  };
  bool fresh = false;
  if ((need & 2) && !e->xn2 && dev_alloc(e, &e->xn2, (size_t)e->rows_pad * 2 * D)) return give_up("the [rows, 2 D] operand buffer");
  if ((need & 12) && !e->lo8 && dev_alloc(e, &e->lo8, (size_t)e->rows_pad * D)) return give_up("the fp8 remainder buffer");
  if ((need & 16) && !e->lo4 && (dev_alloc(e, &e->lo4, (size_t)e->rows_pad * lo4_pitch(D)) || dev_alloc(e, &e->lo4s, (size_t)e->rows_pad)))
    return give_up("the fp4 remainder buffer");   // (dev_alloc zero-fills: the padding columns of a code row stay zero codes)
  for (auto& w : e->blocks) {
    if ((need & 1) && !w.proj_w2) { if (dev_alloc(e, &w.proj_w2, (size_t)D * 2 * D, false)) return give_up("[W | W] of the out-projection"); fresh = true; }
    if ((need & 2) && !w.fc1_w2) { if (dev_alloc(e, &w.fc1_w2, (size_t)Hm * 2 * D, false)) return give_up("[W | W] of fc1"); fresh = true; }
    if ((need & 4) && !w.proj_w8) { if (dev_alloc(e, &w.proj_w8, (size_t)D * D, false)) return give_up("W8 of the out-projection"); fresh = true; }
    if ((need & 8) && !w.fc1_w8) { if (dev_alloc(e, &w.fc1_w8, (size_t)Hm * D, false)) return give_up("W8 of fc1"); fresh = true; }
    if ((need & 16) && !w.fc1_w4) {
      if (dev_alloc(e, &w.fc1_w4, (size_t)Hm * lo4_pitch(D), false) || dev_alloc(e, &w.fc1_w4s, (size_t)Hm, false)) return give_up("W4 of fc1");
      fresh = true;
    }
  }
  if (e->split_w_ready && !fresh) return need;
  if (e->load_event && hipStreamWaitEvent(st, e->load_event, 0) != hipSuccess) return give_up("the wait for the weight conversion");
  const int dt = e->cfg.compute_dtype;
  for (auto& w : e->blocks) {
    for (int h = 0; h < 2; ++h) {
      if (w.proj_w2 && hipMemcpy2DAsync(w.proj_w2 + h * D, (size_t)2 * D * 2, w.proj_w, (size_t)D * 2, (size_t)D * 2, D, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return give_up("the [W | W] copy");
      if (w.fc1_w2 && hipMemcpy2DAsync(w.fc1_w2 + h * D, (size_t)2 * D * 2, w.fc1_w, (size_t)D * 2, (size_t)D * 2, Hm, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return give_up("the [W | W] copy");
    }
    if (w.proj_w8 && launch_pack_w8(w.proj_w, w.proj_w8, (int64_t)D * D, dt, st)) return give_up("the W8 pack");
    if (w.fc1_w8 && launch_pack_w8(w.fc1_w, w.fc1_w8, (int64_t)Hm * D, dt, st)) return give_up("the W8 pack");
    if (w.fc1_w4 && launch_pack_w4(w.fc1_w, w.fc1_w4, w.fc1_w4s, Hm, D, dt, st)) return give_up("the W4 pack");
  }
  e->split_w_ready = true;
  return need;
}

// mod_override != nullptr: the adaLN outputs of this step were precomputed ([B or 1 rows, nmod], row stride mod_stride;
// stride 0 = one row shared by every sample) and the conditioning launches are skipped.
int run_forward(latte_engine* e, const float* x, const int64_t* t, const int64_t* y, int B, bool cfg_dup, float* out,
                hipStream_t st, Prof* prof, const float* mod_override = nullptr, int mod_stride_override = 0) {
  const auto& c = e->cfg;
  if (B <= 0 || B > e->max_batch) return fail(LATTE_ERR_STATE, "forward: batch exceeds max_batch of the engine");
  if (c.extras == 2 && y == nullptr && !mod_override) return fail(LATTE_ERR_INVALID, "forward: class-conditional model needs y");
  if (c.extras == 78 && e->txt_rows != B && !mod_override)
    return fail(LATTE_ERR_STATE, "forward: text-conditioned model needs latte_engine_set_text_embedding with one row per sample first");
  if (cfg_dup && (B % 2)) return fail(LATTE_ERR_INVALID, "forward_with_cfg: batch must be even");
  const int D = e->D, T = e->T, F = e->F, dt = c.compute_dtype;
  const int M = B * F * T;
  const int rps = F * T;
  Timer tm{prof, st};
  int rc;
  tm.mark(C_NONE);
  // --- conditioning: t_emb = MLP(sincos(t)) (latte.py:119-123); c = t_emb (+ y_emb) (:337,:348); all adaLN at once
  const float* modp = e->mod;
  int mstride = e->nmod;
  if (mod_override) {
    modp = mod_override;
    mstride = mod_stride_override;
  } else {
    if ((rc = launch_small_linear(IN_TFREQ, nullptr, t, e->t0_w, e->t0_b, nullptr, nullptr, e->temb0, B, D, 256, D, st))) return rc;
    const float* add_tab = c.extras == 2 ? e->ytab : c.extras == 78 ? e->txt_proj : nullptr;
    const int64_t* add_idx = c.extras == 78 ? e->iota : y;
    if ((rc = launch_small_linear(IN_SILU, e->temb0, nullptr, e->t2_w, e->t2_b, add_tab, add_idx, e->cvec, B, D, D, D, st))) return rc;
    // adaLN_modulation = Linear(SiLU(c)): SiLU once per row here, not once per output feature inside the linear
    if ((rc = launch_silu_rows(e->cvec, e->cvec, (size_t)B * D, st))) return rc;
    if ((rc = launch_small_linear(IN_PLAIN, e->cvec, nullptr, e->ada_w, e->ada_b, nullptr, nullptr, e->mod, B, e->nmod, D,
                                  e->nmod, st))) return rc;
    if (c.extras == 78) {   // final layer: c = t only (latte.py:372-373) -> redo its 2D columns from the plain t_emb
      const size_t fo = (size_t)c.depth * 6 * D;
      if ((rc = launch_small_linear(IN_SILU, e->temb0, nullptr, e->t2_w, e->t2_b, nullptr, nullptr, e->cvec_t, B, D, D, D, st))) return rc;
      if ((rc = launch_silu_rows(e->cvec_t, e->cvec_t, (size_t)B * D, st))) return rc;
      if ((rc = launch_small_linear(IN_PLAIN, e->cvec_t, nullptr, e->ada_w + fo * D, e->ada_b + fo, nullptr, nullptr,
                                    e->mod + fo, B, 2 * D, D, e->nmod, st))) return rc;
    }
  }
  tm.mark(C_COND);
  // --- patch embed + pos_embed (latte.py:330-331)
  if (cfg_dup) {
    const int hb = B / 2;
    if ((rc = launch_patch_embed(x, e->pe_wt, e->pe_b, e->pos, e->xres, hb * F, e->Cin, e->H, c.patch_size, D, st))) return rc;
    if ((rc = launch_patch_embed(x, e->pe_wt, e->pe_b, e->pos, e->xres + (size_t)hb * rps * D, hb * F, e->Cin, e->H,
                                 c.patch_size, D, st))) return rc;
  } else {
    if ((rc = launch_patch_embed(x, e->pe_wt, e->pe_b, e->pos, e->xres, B * F, e->Cin, e->H, c.patch_size, D, st))) return rc;
  }
  tm.mark(C_PATCH);
  // split-operand linears of a guided call (latte_engine::guided_split); the K' = 2 K operands must stay inside the 32-bit buffer offsets
  int gsplit = cfg_dup ? e->guided_split : 0;
  if (dt != LATTE_DTYPE_F16) gsplit &= 3;                                          // the fp8 / fp4 remainders exist beside f16 only
  if ((gsplit & 16) && !gemm_lo4_ok(M, e->Hm, D)) gsplit = (gsplit & ~16) | 8;     // no fp4 form for the shape: the fp8 one
  if (gsplit & 16) gsplit &= ~10;                                                  // one form per operand: fp4 wins for fc1's
  if (gsplit & 4) gsplit &= ~1;                                                    // fp8 wins over the f16 pair
  if (gsplit & 8) gsplit &= ~2;
  if ((gsplit & 4) && !gemm_lo8_ok(M, D, D)) gsplit = (gsplit & ~4) | 1;          // no fp8 form for the shape (N % 192, K % 128): the f16 pair
  if ((gsplit & 8) && !gemm_lo8_ok(M, e->Hm, D)) gsplit = (gsplit & ~8) | 2;
  if ((gsplit & 3) && (uint64_t)e->rows_pad * 2 * D * 2 >= (1ull << 32)) gsplit &= ~3;
  if (gsplit) gsplit = ensure_split_weights(e, gsplit, st);
  int gactive = gsplit & 26;   // bits 1 / 3 / 4 always apply; bits 0 / 2 once a fused attention kernel has produced the pair

  for (int i = 0; i < c.depth; ++i) {
    const bool spatial = (i % 2) == 0;  // latte.py:345-346
    const BlockW& w = e->blocks[i];
    const float* mb = modp + (size_t)i * 6 * D;  // chunk(6): shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
    bool split_proj = false;   // this block's attention output leaves the fused kernel as a split pair
    // x + temp_embed once, after the first spatial block (latte.py:357-358)
    const float* te = (i == 1) ? e->temp : nullptr;
    if ((rc = launch_ln_modulate(e->xres, e->xres, e->xn, mb, mb + D, mstride, M, D, rps, te, T, F, dt, st))) return rc;
    tm.mark(C_LN);
    GemmArgs g{};
    g.M = M; g.rows_per_sample = rps; g.gate_stride = mstride;
    const half_t* attn_out = e->xn;
    if (((e->fuse_qkv_attn >> (spatial ? 0 : 1)) & 1) && qkv_attention_fusable(D, c.num_heads, e->hd, F, T, spatial ? 0 : 1, M)) {
      // qkv projection + attention core in one kernel (q / k / v of a head only in LDS); the output goes to the (otherwise idle)
      // qkv buffer viewed as [rows, D], because xn is still being read by other units of the same launch
      QkvAttnArgs qa{};
      qa.xn = e->xn; qa.w = w.qkv_w; qa.bias = w.qkv_b; qa.out = e->qkv; qa.B = B; qa.F = F; qa.T = T; qa.D = D;
      qa.heads = c.num_heads; qa.hd = e->hd; qa.mode = spatial ? 0 : 1; qa.scale = 1.0f / std::sqrt((float)e->hd);
      qa.flags = ((e->fuse_qkv_attn >> 2) & 7) ^ 7;   // option bits 2-4 switch the default schedule features OFF (A/B hook)
      split_proj = gsplit & 5;
      qa.out_split = (gsplit & 4) ? 2 : (gsplit & 1);   // 1: [rows, 2 D] = [hi | lo] inside the [rows, 3 D] qkv buffer; 2: + fp8 remainder
      qa.out8 = e->lo8;
      gactive |= gsplit & 5;
      if ((rc = launch_qkv_attention(qa, dt, st))) return rc;
      tm.mark(spatial ? C_QKVATTN_S : C_QKVATTN_T);
      attn_out = e->qkv;
    } else {
      g.A = e->xn; g.W = w.qkv_w; g.bias = w.qkv_b; g.out = e->qkv; g.N = 3 * D; g.K = D;
      if ((rc = launch_gemm(g, EPI_BIAS_H16, dt, e->gemm_variant_of[0] ? e->gemm_variant_of[0] : e->gemm_variant, st))) return rc;
      tm.mark(C_QKV);
      AttnArgs a{};
      a.qkv = e->qkv; a.out = e->xn; a.heads = c.num_heads; a.hd = e->hd; a.D = D;
      a.sample_stride = rps; a.scale = 1.0f / std::sqrt((float)e->hd);
      if (spatial) { a.num_seq = B * F; a.L = T; a.U = F; a.seq_stride = T; a.row_stride = 1; }
      else         { a.num_seq = B * T; a.L = F; a.U = T; a.seq_stride = 1; a.row_stride = T; }
      if (gsplit & 4) {   // the un-fused attention kernels carry the fp8 remainder too (round 6); the f16-pair form (bit 0) exists in the fused kernel only
        a.out8 = e->lo8;
        split_proj = true;
        gactive |= 4;
      }
      if ((rc = launch_attention(a, dt, st))) return rc;
      tm.mark(spatial ? C_ATTN_S : C_ATTN_T);
    }
    g.A = attn_out; g.W = w.proj_w; g.bias = w.proj_b; g.out = e->xres; g.gate = mb + 2 * D; g.N = D; g.K = D; g.tag = 0;
    if (split_proj && (gsplit & 4)) { g.A8 = e->lo8; g.W8 = w.proj_w8; }          // o_hi . W^T + o_lo8 . W8^T in one launch
    else if (split_proj) { g.W = w.proj_w2; g.K = 2 * D; }                          // [o_hi | o_lo] . [W | W]^T
    if ((rc = gated_gemm(e, g, dt, e->gemm_variant_of[1] ? e->gemm_variant_of[1] : e->gemm_variant, st))) return rc;
    g.A8 = nullptr; g.W8 = nullptr;
    tm.mark(C_PROJ);
    const int split_fc1 = (gsplit & 16) ? 3 : (gsplit & 8) ? 2 : (gsplit & 2) ? 1 : 0;
    if (split_fc1 == 3) rc = launch_ln_modulate_split4(e->xres, e->xn, e->lo4, e->lo4s, mb + 3 * D, mb + 4 * D, mstride, M, D, rps, dt, st);
    else rc = launch_ln_modulate(e->xres, e->xres, split_fc1 == 1 ? e->xn2 : e->xn, mb + 3 * D, mb + 4 * D, mstride, M, D, rps, nullptr, T, F, dt, st,
                                 split_fc1, e->lo8);
    if (rc) return rc;
    tm.mark(C_LN);
    g.A = e->xn; g.W = w.fc1_w; g.bias = w.fc1_b; g.out = e->hbuf; g.gate = nullptr; g.N = e->Hm; g.K = D;
    if (split_fc1 == 1) { g.A = e->xn2; g.W = w.fc1_w2; g.K = 2 * D; }
    if (split_fc1 == 2) { g.A8 = e->lo8; g.W8 = w.fc1_w8; }
    if (split_fc1 == 3) { g.A4 = e->lo4; g.A4s = e->lo4s; g.W4 = w.fc1_w4; g.W4s = w.fc1_w4s; }
    if ((rc = launch_gemm(g, EPI_BIAS_GELU_H16, dt, e->gemm_variant_of[2] ? e->gemm_variant_of[2] : e->gemm_variant, st))) return rc;
    g.A8 = nullptr; g.W8 = nullptr; g.A4 = nullptr; g.A4s = nullptr; g.W4 = nullptr; g.W4s = nullptr;
    tm.mark(C_FC1);
    g.A = e->hbuf; g.W = w.fc2_w; g.bias = w.fc2_b; g.out = e->xres; g.gate = mb + 5 * D; g.N = D; g.K = e->Hm; g.tag = 1;

    if ((rc = gated_gemm(e, g, dt, e->gemm_variant_of[3] ? e->gemm_variant_of[3] : e->gemm_variant, st))) return rc;
    tm.mark(C_FC2);
  }
  // --- final layer (latte.py:197-201) + unpatchify (:297-310)
  const float* fm = modp + (size_t)c.depth * 6 * D;  // chunk(2): shift, scale
  if ((rc = launch_final_layer(e->xres, fm, fm + D, mstride, e->fin_wt, e->fin_b, out, M, D, rps, T, c.patch_size,
                               e->Cout, e->H, st))) return rc;
  tm.mark(C_FINAL);
  if (cfg_dup) e->guided_split_active = gactive;
  return LATTE_OK;
}

// fp32 emulation of the reference's per-step tensor arithmetic (gaussian_diffusion.py:869-881: fp64 table ->
// .float()); compiled with -ffp-contract=off.
int grow(latte_engine* e, float** p, int64_t* cap, int64_t need) {
  if (*cap >= need) return LATTE_OK;
  if (*p) {   // release the smaller block (setup path: the implicit device synchronisation of hipFree is fine here)
    e->allocs.erase(std::remove(e->allocs.begin(), e->allocs.end(), (void*)*p), e->allocs.end());
    (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
  }
  int rc = dev_alloc(e, p, (size_t)need, false);
  if (!rc) *cap = need;
  return rc;
}

// Timestep-embedding table of a schedule: row i = t_embedder(timestep_map[i]) (latte.py:84-123, respace.py:125-130).
// This is the [steps, D] fp32 table rank 0 broadcasts over RCCL in the multi-GPU driver (1.15 MB for 250 x 1152).
int compute_temb_table(latte_engine* e, const latte_schedule_t* s, float* out, hipStream_t st) {
  const int n = s->num_timesteps, D = e->D;
  int rc;
  if (e->tmap_cap < n) {
    if ((rc = dev_alloc(e, &e->tmap_dev, (size_t)n, false))) return rc;
    e->tmap_cap = n;
  }
  LATTE_HIP(hipMemcpyAsync(e->tmap_dev, s->timestep_map.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice, st));
  LATTE_HIP(hipStreamSynchronize(st));   // the host vector may be freed by the caller
  if ((rc = grow(e, &e->temb_work, &e->temb_cap, (int64_t)n * D))) return rc;
  constexpr int CH = 64;   // rows per launch: the kernel keeps a weight row in registers and loops over the rows
  for (int r0 = 0; r0 < n; r0 += CH) {
    const int rows = std::min(CH, n - r0);
    if ((rc = launch_small_linear(IN_TFREQ, nullptr, e->tmap_dev + r0, e->t0_w, e->t0_b, nullptr, nullptr,
                                  e->temb_work + (size_t)r0 * D, rows, D, 256, D, st))) return rc;
    if ((rc = launch_small_linear(IN_SILU, e->temb_work + (size_t)r0 * D, nullptr, e->t2_w, e->t2_b, nullptr, nullptr,
                                  out + (size_t)r0 * D, rows, D, D, D, st))) return rc;
  }
  return LATTE_OK;
}

SamplerCoefs make_coefs(const latte_schedule_t* s, int method, int i, float eta, int clip) {
  SamplerCoefs c{};
  c.method = method;
  c.clip = clip;
  c.min_log = (float)s->posterior_log_variance_clipped[i];
  c.max_log = (float)s->log_betas[i];
  c.sqrt_recip = (float)s->sqrt_recip_alphas_cumprod[i];
  c.sqrt_recipm1 = (float)s->sqrt_recipm1_alphas_cumprod[i];
  c.coef1 = (float)s->posterior_mean_coef1[i];
  c.coef2 = (float)s->posterior_mean_coef2[i];
  const float ab = (float)s->alphas_cumprod[i], abp = (float)s->alphas_cumprod_prev[i];
  // gd:549-553  sigma = eta * sqrt((1-abp)/(1-ab)) * sqrt(1 - ab/abp)
  const float sigma = (eta * std::sqrt((1.0f - abp) / (1.0f - ab))) * std::sqrt(1.0f - ab / abp);
  c.sigma = sigma;
  c.sqrt_ab_prev = std::sqrt(abp);
  c.dir_coef = std::sqrt((1.0f - abp) - sigma * sigma);  // gd:558: sqrt(1 - abp - sigma**2)
  c.nonzero = i == 0 ? 0.0f : 1.0f;
  c.cfg_scale = 1.0f;
  c.sqrt_one_minus_ab = std::sqrt(1.0f - ab);   // gd:368 on the fp32 alpha_bar
  c.mean_type = s->mean_type;
  c.var_type = s->var_type;
  // gd:298-313: FIXED_LARGE = log(append(posterior_variance[1], betas[1:])), FIXED_SMALL = posterior_log_variance_clipped
  if (s->var_type == 1) c.fixed_log_var = (float)std::log(i == 0 ? s->posterior_variance[1] : s->betas[i]);
  else c.fixed_log_var = (float)s->posterior_log_variance_clipped[i];
  return c;
}

}  // namespace

extern "C" {

const char* latte_last_error(void) { return latte::g_last_error.c_str(); }
const char* latte_version(void) { return "latte_amd 0.1 (gfx950, HIP MFMA engine)"; }

int latte_engine_create(const latte_model_config_t* cfg, int max_batch, latte_engine_t** out) {
  if (!cfg || !out || max_batch <= 0) return fail(LATTE_ERR_INVALID, "engine_create: bad arguments");
  const auto& c = *cfg;
  if (c.hidden_size % 128 != 0) return fail(LATTE_ERR_INVALID, "engine_create: hidden_size must be a multiple of 128");
  if (c.depth <= 0 || c.depth % 2) return fail(LATTE_ERR_INVALID, "engine_create: depth must be even (spatial/temporal pairs)");
  if (c.num_heads <= 0 || c.hidden_size % c.num_heads) return fail(LATTE_ERR_INVALID, "dim should be divisible by num_heads");
  const int hd = c.hidden_size / c.num_heads;
  if (hd != 64 && hd != 72) return fail(LATTE_ERR_INVALID, "engine_create: head_dim must be 64 or 72");
  if (c.patch_size <= 0 || c.input_size % c.patch_size) return fail(LATTE_ERR_INVALID, "engine_create: input_size % patch_size != 0");
  if (c.mlp_hidden % 128 != 0) return fail(LATTE_ERR_INVALID, "engine_create: mlp_hidden must be a multiple of 128");
  if (c.extras != 1 && c.extras != 2 && c.extras != 78)
    return fail(LATTE_ERR_INVALID, "engine_create: extras must be 1 (uncond), 2 (class-cond) or 78 (text embedding)");
  if (c.in_channels != 4) return fail(LATTE_ERR_INVALID, "engine_create: in_channels must be 4 (guidance is hard-wired to 4 channels, latte.py:394)");
  if (c.compute_dtype != LATTE_DTYPE_BF16 && c.compute_dtype != LATTE_DTYPE_F16) return fail(LATTE_ERR_INVALID, "engine_create: bad compute dtype");

  auto* e = new latte_engine();
  e->cfg = c;
  if (c.compute_dtype != LATTE_DTYPE_F16) e->guided_split = 0;   // bf16 operands are a 3e-3 type at trained-scale gates, pair or not
  e->max_batch = max_batch;
  e->D = c.hidden_size;
  e->G = c.input_size / c.patch_size;
  e->T = e->G * e->G;
  e->F = c.num_frames;
  e->Cin = c.in_channels;
  e->Cout = c.learn_sigma ? 2 * c.in_channels : c.in_channels;
  e->H = c.input_size;
  e->P = c.patch_size * c.patch_size * e->Cout;
  e->KPE = c.in_channels * c.patch_size * c.patch_size;
  e->Hm = c.mlp_hidden;
  e->hd = hd;
  e->nmod = c.depth * 6 * e->D + 2 * e->D;
  e->rows_max = (int64_t)max_batch * e->F * e->T;
  e->rows_pad = (e->rows_max + 255) / 256 * 256;
  const int D = e->D;
  int rc = LATTE_OK;
#define TRY(x) do { if ((rc = (x))) { latte_engine_destroy(e); return rc; } } while (0)
  TRY(dev_alloc(e, &e->ada_w, (size_t)e->nmod * D));
  TRY(dev_alloc(e, &e->ada_b, (size_t)e->nmod));
  TRY(dev_alloc(e, &e->pos, (size_t)e->T * D));
  TRY(dev_alloc(e, &e->temp, (size_t)e->F * D));
  TRY(dev_alloc(e, &e->pe_wt, (size_t)e->KPE * D));
  TRY(dev_alloc(e, &e->pe_b, (size_t)D));
  TRY(dev_alloc(e, &e->t0_w, (size_t)D * 256));
  TRY(dev_alloc(e, &e->t0_b, (size_t)D));
  TRY(dev_alloc(e, &e->t2_w, (size_t)D * D));
  TRY(dev_alloc(e, &e->t2_b, (size_t)D));
  TRY(dev_alloc(e, &e->fin_wt, (size_t)D * e->P));
  TRY(dev_alloc(e, &e->fin_b, (size_t)e->P));
  add_slot(e, "pos_embed", (int64_t)e->T * D, PK_F32, e->pos);
  add_slot(e, "temp_embed", (int64_t)e->F * D, PK_F32, e->temp);
  add_slot(e, "x_embedder.proj.weight", (int64_t)D * e->KPE, PK_F32_TRANSPOSE, e->pe_wt, D, e->KPE);
  add_slot(e, "x_embedder.proj.bias", D, PK_F32, e->pe_b);
  add_slot(e, "t_embedder.mlp.0.weight", (int64_t)D * 256, PK_F32, e->t0_w);
  add_slot(e, "t_embedder.mlp.0.bias", D, PK_F32, e->t0_b);
  add_slot(e, "t_embedder.mlp.2.weight", (int64_t)D * D, PK_F32, e->t2_w);
  add_slot(e, "t_embedder.mlp.2.bias", D, PK_F32, e->t2_b);
  if (c.extras == 2) {
    TRY(dev_alloc(e, &e->ytab, (size_t)(c.num_classes + 1) * D));
    add_slot(e, "y_embedder.embedding_table.weight", (int64_t)(c.num_classes + 1) * D, PK_F32, e->ytab);
  }
  if (c.extras == 78) {
    constexpr int64_t TK = 77 * 768;   // latte.py:241
    TRY(dev_alloc(e, &e->txt_w, (size_t)D * TK));
    TRY(dev_alloc(e, &e->txt_b, (size_t)D));
    TRY(dev_alloc(e, &e->txt_proj, (size_t)max_batch * D));
    TRY(dev_alloc(e, &e->cvec_t, (size_t)max_batch * D));
    TRY(dev_alloc(e, &e->iota, (size_t)max_batch));
    TRY(launch_iota(e->iota, max_batch, nullptr));
    LATTE_HIP(hipStreamSynchronize(nullptr));
    add_slot(e, "text_embedding_projection.1.weight", (int64_t)D * TK, PK_F32, e->txt_w);
    add_slot(e, "text_embedding_projection.1.bias", D, PK_F32, e->txt_b);
  }
  e->blocks.resize(c.depth);
  for (int i = 0; i < c.depth; ++i) {
    BlockW& w = e->blocks[i];
    TRY(dev_alloc(e, &w.qkv_w, (size_t)3 * D * D));
    TRY(dev_alloc(e, &w.proj_w, (size_t)D * D));
    TRY(dev_alloc(e, &w.fc1_w, (size_t)e->Hm * D));
    TRY(dev_alloc(e, &w.fc2_w, (size_t)D * e->Hm));
    TRY(dev_alloc(e, &w.qkv_b, (size_t)3 * D));
    TRY(dev_alloc(e, &w.proj_b, (size_t)D));
    TRY(dev_alloc(e, &w.fc1_b, (size_t)e->Hm));
    TRY(dev_alloc(e, &w.fc2_b, (size_t)D));
    const std::string p = "blocks." + std::to_string(i) + ".";
    add_slot(e, p + "attn.qkv.weight", (int64_t)3 * D * D, PK_H16, w.qkv_w);
    add_slot(e, p + "attn.qkv.bias", 3 * D, PK_F32, w.qkv_b);
    add_slot(e, p + "attn.proj.weight", (int64_t)D * D, PK_H16, w.proj_w);
    add_slot(e, p + "attn.proj.bias", D, PK_F32, w.proj_b);
    add_slot(e, p + "mlp.fc1.weight", (int64_t)e->Hm * D, PK_H16, w.fc1_w);
    add_slot(e, p + "mlp.fc1.bias", e->Hm, PK_F32, w.fc1_b);
    add_slot(e, p + "mlp.fc2.weight", (int64_t)D * e->Hm, PK_H16, w.fc2_w);
    add_slot(e, p + "mlp.fc2.bias", D, PK_F32, w.fc2_b);
    add_slot(e, p + "adaLN_modulation.1.weight", (int64_t)6 * D * D, PK_F32, e->ada_w + (size_t)i * 6 * D * D);
    add_slot(e, p + "adaLN_modulation.1.bias", 6 * D, PK_F32, e->ada_b + (size_t)i * 6 * D);
  }
  add_slot(e, "final_layer.linear.weight", (int64_t)e->P * D, PK_F32_TRANSPOSE, e->fin_wt, e->P, D);
  add_slot(e, "final_layer.linear.bias", e->P, PK_F32, e->fin_b);
  add_slot(e, "final_layer.adaLN_modulation.1.weight", (int64_t)2 * D * D, PK_F32, e->ada_w + (size_t)c.depth * 6 * D * D);
  add_slot(e, "final_layer.adaLN_modulation.1.bias", 2 * D, PK_F32, e->ada_b + (size_t)c.depth * 6 * D);

  TRY(dev_alloc(e, &e->stage, (size_t)e->stage_numel, false));
  TRY(dev_alloc(e, &e->xres, (size_t)e->rows_pad * D));
  TRY(dev_alloc(e, &e->split_ws, (size_t)SPLIT_WS_FLOATS, false));
  TRY(dev_alloc(e, &e->zero_bias, (size_t)std::max(D, e->Hm)));
  TRY(dev_alloc(e, &e->xn, (size_t)e->rows_pad * D));
  TRY(dev_alloc(e, &e->qkv, (size_t)e->rows_pad * 3 * D));
  TRY(dev_alloc(e, &e->hbuf, (size_t)e->rows_pad * e->Hm));
  TRY(dev_alloc(e, &e->mod, (size_t)max_batch * e->nmod));
  TRY(dev_alloc(e, &e->temb0, (size_t)max_batch * D));
  TRY(dev_alloc(e, &e->cvec, (size_t)max_batch * D));
  TRY(dev_alloc(e, &e->model_out, (size_t)max_batch * e->F * e->Cout * e->H * e->H));
  TRY(dev_alloc(e, &e->noise_buf, (size_t)max_batch * e->F * e->Cin * e->H * e->H));
#undef TRY
  *out = e;
  return LATTE_OK;
}

void latte_engine_destroy(latte_engine_t* e) {
  if (!e) return;
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->load_event) (void)hipEventDestroy(e->load_event);
  delete e;
}

int latte_engine_num_keys(const latte_engine_t* e) { return e ? (int)e->slots.size() : 0; }
const char* latte_engine_key(const latte_engine_t* e, int i) {
  if (!e || i < 0 || i >= (int)e->slots.size()) return nullptr;
  return e->slots[i].key.c_str();
}

int latte_engine_set_option(latte_engine_t* e, const char* name, int64_t value) {
  if (!e || !name) return fail(LATTE_ERR_INVALID, "set_option: null argument");
  const std::string k = name;
  if (k == "gemm_variant") {
    if (value < 0 || (value > 13 && value != 18 && value != 19)) return fail(LATTE_ERR_INVALID, "gemm_variant must be 0..13, 18 or 19");
    const int bn = value >= 7 && value <= 9 ? gemm_tile_n((int)value) / 4 : gemm_tile_n((int)value);   // 7-9: whole wave widths
    if (value != 0 && ((3 * e->D) % bn || e->D % bn || e->Hm % bn))
      return fail(LATTE_ERR_INVALID, "gemm_variant: every N of the model must be a multiple of the tile width");
    e->gemm_variant = (int)value;
    return LATTE_OK;
  }
  for (int gi = 0; gi < 4; ++gi) {
    static const char* names[4] = {"gemm_variant_qkv", "gemm_variant_proj", "gemm_variant_fc1", "gemm_variant_fc2"};
    if (k == names[gi]) {
#ifdef LATTE_GEMM_ABLATE   // measurement build: 17 = the two-accumulator-set kernel of the gated GEMMs (gemm_pw.hip)
      if (value == 17 && (gi == 1 || gi == 3)) { e->gemm_variant_of[gi] = 17; return LATTE_OK; }
#endif
      if (value < 0 || (value > 13 && value != 18 && value != 19)) return fail(LATTE_ERR_INVALID, "gemm_variant_*: must be 0..13, 18 or 19");
      e->gemm_variant_of[gi] = (int)value;
      return LATTE_OK;
    }
  }
  if (k == "gated_split_k") {
    if (value < 0 || value > 4) return fail(LATTE_ERR_INVALID, "gated_split_k: 0 (rule), 1 (off) or 2..4 splits");
    e->gated_split_k = (int)value;
    return LATTE_OK;
  }
  if (k == "fuse_qkv_attn") {
    if (value < 0 || value > 31)
      return fail(LATTE_ERR_INVALID, "fuse_qkv_attn: bit 0 = spatial blocks, bit 1 = temporal blocks, bits 2-4 = schedule features off (0..31)");
    e->fuse_qkv_attn = (int)value;
    return LATTE_OK;
  }
  if (k == "guided_split") {
    if (value < 0 || value > 31)
      return fail(LATTE_ERR_INVALID, "guided_split: bit 0 / 1 = attention output / fc1 operand as f16 split pairs, bit 2 / 3 = the same with an fp8 remainder, bit 4 = fc1's operand with an fp4 remainder, in guided calls (0..31)");
    e->guided_split = (int)value;
    return LATTE_OK;
  }
  if (k == "seed") {
    e->seed = (uint64_t)value;
    e->rng_offset = 0;
    return LATTE_OK;
  }
  return fail(LATTE_ERR_INVALID, "set_option: unknown option '" + k + "'");
}

int latte_engine_get_option(const latte_engine_t* e, const char* name, int64_t* value) {
  if (!e || !name || !value) return fail(LATTE_ERR_INVALID, "get_option: null argument");
  const std::string k = name;
  if (k == "gemm_variant") *value = e->gemm_variant;
  else if (k == "gemm_variant_qkv") *value = e->gemm_variant_of[0];
  else if (k == "gemm_variant_proj") *value = e->gemm_variant_of[1];
  else if (k == "gemm_variant_fc1") *value = e->gemm_variant_of[2];
  else if (k == "gemm_variant_fc2") *value = e->gemm_variant_of[3];
  else if (k == "gated_split_k") *value = e->gated_split_k;
  else if (k == "fuse_qkv_attn") *value = e->fuse_qkv_attn;
  else if (k == "guided_split") *value = e->guided_split;
  else if (k == "guided_split_active") *value = e->guided_split_active;
  else if (k == "guided_split_failed") *value = e->split_failed ? 1 : 0;
  else if (k == "seed") *value = (int64_t)e->seed;
  else return fail(LATTE_ERR_INVALID, "get_option: unknown option '" + k + "'");
  return LATTE_OK;
}

int latte_engine_load_tensor(latte_engine_t* e, const char* key, const float* data, int64_t numel, int on_device,
                             void* stream) {
  if (!e || !key || !data) return fail(LATTE_ERR_INVALID, "load_tensor: null argument");
  auto it = e->slot_index.find(key);
  if (it == e->slot_index.end()) return fail(LATTE_ERR_INVALID, std::string("load_tensor: unexpected key '") + key + "'");
  TensorSlot& s = e->slots[it->second];
  if (numel != s.numel)
    return fail(LATTE_ERR_INVALID, std::string("load_tensor: size mismatch for '") + key + "': got " + std::to_string(numel) +
                                       ", expected " + std::to_string(s.numel));
  hipStream_t st = (hipStream_t)stream;
  const float* src = data;
  if (!on_device) {
    LATTE_HIP(hipMemcpyAsync(e->stage, data, sizeof(float) * numel, hipMemcpyHostToDevice, st));
    src = e->stage;
  }
  int rc = LATTE_OK;
  switch (s.kind) {
    case PK_F32:
      LATTE_HIP(hipMemcpyAsync(s.dst, src, sizeof(float) * numel, hipMemcpyDeviceToDevice, st));
      break;
    case PK_F32_TRANSPOSE:
      rc = launch_transpose_f32(src, (float*)s.dst, s.rows, s.cols, st);
      break;
    case PK_H16:
      rc = launch_convert_f32_to_h16(src, (half_t*)s.dst, numel, e->cfg.compute_dtype, st);
      break;
  }
  if (rc) return rc;
  if (!on_device) LATTE_HIP(hipStreamSynchronize(st));  // the staging buffer is reused by the next call
  s.loaded = true;
  if (s.kind == PK_H16) {   // a block weight changed: the derived copies ([W | W], W8) are stale and must be rebuilt BEHIND this conversion
    e->split_w_ready = false;
    if (!e->load_event) LATTE_HIP(hipEventCreateWithFlags(&e->load_event, hipEventDisableTiming));
    LATTE_HIP(hipEventRecord(e->load_event, st));
  }
  if (s.key.compare(0, 11, "t_embedder.") == 0) {   // an installed timestep-embedding table was computed from the old weights
    e->temb_table_n = 0;
    e->temb_table_map.clear();
    e->temb_own_map.clear();
  }
  return LATTE_OK;
}

int latte_engine_check_weights(latte_engine_t* e) {
  if (!e) return fail(LATTE_ERR_INVALID, "check_weights: null engine");
  for (const auto& s : e->slots)
    if (!s.loaded) return fail(LATTE_ERR_STATE, "Missing key(s) in state_dict: \"" + s.key + "\"");
  return LATTE_OK;
}

int latte_engine_temb_table(latte_engine_t* e, const latte_schedule_t* s, float* out, void* stream) {
  if (!e || !s || !out) return fail(LATTE_ERR_INVALID, "temb_table: null argument");
  int rc = latte_engine_check_weights(e);
  if (rc) return rc;
  return compute_temb_table(e, s, out, (hipStream_t)stream);
}

int latte_engine_set_temb_table(latte_engine_t* e, const latte_schedule_t* s, const float* table, void* stream) {
  if (!e) return fail(LATTE_ERR_INVALID, "set_temb_table: null engine");
  if (!table || !s) {   // uninstall: the engine computes its own table again
    e->temb_table_n = 0;
    e->temb_table_map.clear();
    return LATTE_OK;
  }
  const int num_timesteps = s->num_timesteps;
  int rc = grow(e, &e->temb_table, &e->temb_table_cap, (int64_t)num_timesteps * e->D);
  if (rc) return rc;
  LATTE_HIP(hipMemcpyAsync(e->temb_table, table, sizeof(float) * (size_t)num_timesteps * e->D, hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
  e->temb_table_n = num_timesteps;
  e->temb_table_map = s->timestep_map;   // row i is t_embedder(timestep_map[i]): valid for this map only
  return LATTE_OK;
}

int latte_engine_set_text_embedding(latte_engine_t* e, const float* text_embedding, int batch, void* stream) {
  if (!e || !text_embedding) return fail(LATTE_ERR_INVALID, "set_text_embedding: null argument");
  if (e->cfg.extras != 78) return fail(LATTE_ERR_STATE, "set_text_embedding: the model has no text_embedding_projection (extras != 78)");
  if (batch <= 0 || batch > e->max_batch) return fail(LATTE_ERR_STATE, "set_text_embedding: batch exceeds max_batch of the engine");
  int rc = latte_engine_check_weights(e);
  if (rc) return rc;
  if ((rc = launch_text_proj(text_embedding, e->txt_w, e->txt_b, e->txt_proj, batch, e->D, 77 * 768, (hipStream_t)stream))) return rc;
  e->txt_rows = batch;
  return LATTE_OK;
}

int latte_forward(latte_engine_t* e, const float* x, const int64_t* t, const int64_t* y, int batch, float* out,
                  void* stream) {
  if (!e || !x || !t || !out) return fail(LATTE_ERR_INVALID, "forward: null argument");
  int rc = latte_engine_check_weights(e);
  if (rc) return rc;
  return run_forward(e, x, t, y, batch, false, out, (hipStream_t)stream, nullptr);
}

int latte_forward_with_cfg(latte_engine_t* e, const float* x, const int64_t* t, const int64_t* y, int batch,
                           float cfg_scale, float* out, void* stream) {
  if (!e || !x || !t || !out) return fail(LATTE_ERR_INVALID, "forward_with_cfg: null argument");
  int rc = latte_engine_check_weights(e);
  if (rc) return rc;
  if ((rc = run_forward(e, x, t, y, batch, true, out, (hipStream_t)stream, nullptr))) return rc;
  return launch_cfg_combine(out, batch / 2, e->F, e->Cout, e->H * e->H, cfg_scale, (hipStream_t)stream);
}

int latte_sampler_step_ex(const latte_schedule_t* s, int method, int index, float eta, int clip_denoised, const float* x,
                          const float* model_out, const float* noise, const float* pred_xstart_in, const float* cond_grad,
                          int predict_only, int batch, int frames, int channels, int hw, float* sample_out,
                          float* pred_xstart_out, void* stream) {
  if (!s || !x || !model_out) return fail(LATTE_ERR_INVALID, "sampler_step: null argument");
  if (predict_only ? !pred_xstart_out : !sample_out) return fail(LATTE_ERR_INVALID, "sampler_step: null output");
  if (index < 0 || index >= s->num_timesteps) return fail(LATTE_ERR_INVALID, "sampler_step: index out of range");
  if (method != LATTE_METHOD_DDPM && method != LATTE_METHOD_DDIM) return fail(LATTE_ERR_INVALID, "sampler_step: bad method");
  if (s->num_timesteps < 2) return fail(LATTE_ERR_INVALID, "sampler_step: learned-range variance needs >= 2 timesteps");
  SamplerCoefs c = make_coefs(s, method, index, eta, clip_denoised);
  if (!predict_only && method == LATTE_METHOD_DDPM && index != 0 && noise == nullptr)
    return fail(LATTE_ERR_INVALID, "sampler_step: DDPM needs noise for index > 0");
  const bool need_noise = (method == LATTE_METHOD_DDPM) ? (index != 0) : (c.sigma != 0.0f && index != 0);
  return launch_sampler_update(c, x, model_out, need_noise ? noise : nullptr, batch, frames, channels, hw, 0, sample_out,
                               pred_xstart_out, (hipStream_t)stream, pred_xstart_in, cond_grad, predict_only);
}

int latte_sampler_step(const latte_schedule_t* s, int method, int index, float eta, int clip_denoised, const float* x,
                       const float* model_out, const float* noise, int batch, int frames, int channels, int hw,
                       float* sample_out, float* pred_xstart_out, void* stream) {
  return latte_sampler_step_ex(s, method, index, eta, clip_denoised, x, model_out, noise, nullptr, nullptr, 0, batch, frames,
                               channels, hw, sample_out, pred_xstart_out, stream);
}

int latte_sample_loop(latte_engine_t* e, const latte_schedule_t* s, int method, float eta, int clip_denoised,
                      float cfg_scale, float* x, const int64_t* y, int batch, int start_index, int end_index,
                      const float* noise, float* trail_sample, float* trail_x0, void* stream) {
  return latte_sample_loop_ex(e, s, method, eta, clip_denoised, cfg_scale > 1.0f ? 1 : 0 /* sample.py:51 */, cfg_scale, x, y,
                              batch, start_index, end_index, noise, trail_sample, trail_x0, stream);
}

int latte_sample_loop_ex(latte_engine_t* e, const latte_schedule_t* s, int method, float eta, int clip_denoised, int guided,
                         float cfg_scale, float* x, const int64_t* y, int batch, int start_index, int end_index,
                         const float* noise, float* trail_sample, float* trail_x0, void* stream) {
  if (!e || !s || !x) return fail(LATTE_ERR_INVALID, "sample_loop: null argument");
  int rc = latte_engine_check_weights(e);
  if (rc) return rc;
  const int n = s->num_timesteps;
  if (start_index >= n || end_index < 0 || start_index < end_index) return fail(LATTE_ERR_INVALID, "sample_loop: bad index range");
  if (n < 2) return fail(LATTE_ERR_INVALID, "sample_loop: learned-range variance needs >= 2 timesteps");
  if (batch <= 0 || batch > e->max_batch) return fail(LATTE_ERR_STATE, "sample_loop: batch exceeds max_batch");
  if ((s->var_type == 0) != (e->Cout == 2 * e->Cin))
    return fail(LATTE_ERR_INVALID, "sample_loop: the model's learn_sigma and the diffusion's learn_sigma disagree "
                                   "(model output channels vs ModelVarType, gd:290 / :338)");
  const bool use_cfg = guided != 0;   // forward_with_cfg semantics for ANY scale (latte.py:379-398 has no threshold)
  if (use_cfg && (batch % 2)) return fail(LATTE_ERR_INVALID, "sample_loop: guidance needs the doubled batch");
  hipStream_t st = (hipStream_t)stream;
  // ---- conditioning of the whole chain, once: it depends on (timestep, label) only, never on x.
  //   temb[i]      = t_embedder(timestep_map[i])           (own, or the table installed after the RCCL broadcast)
  //   c[i, b]      = SiLU(temb[i] (+ y_embedder(y[b])))     latte.py:337,348 (+ the SiLU of :173)
  //   mod[i, b, :] = all 28 adaLN_modulation + the final layer's, = Linear(SiLU(c))   latte.py:172-178,192-198
  // Unconditional models have one row per step shared by every sample (row stride 0).  The adaLN weights (0.9 GB
  // fp32) are then streamed once per 64 rows instead of once per denoising step.
  const int D = e->D;
  const int n_run = start_index - end_index + 1;
  const int ex = e->cfg.extras;
  const int bu = ex == 1 ? 1 : batch;
  if (ex == 2 && y == nullptr) return fail(LATTE_ERR_INVALID, "sample_loop: class-conditional model needs y");
  if (ex == 78 && e->txt_rows != batch)
    return fail(LATTE_ERR_STATE, "sample_loop: text-conditioned model needs latte_engine_set_text_embedding with one row per sample first");
  const float* temb = nullptr;
  if (e->temb_table_n == n && e->temb_table_map == s->timestep_map) {   // installed for exactly this timestep_map
    temb = e->temb_table;
  } else {
    if (e->temb_own_map != s->timestep_map) {
      e->temb_own_map.clear();
      if ((rc = grow(e, &e->temb_own, &e->temb_own_cap, (int64_t)n * D))) return rc;
      if ((rc = compute_temb_table(e, s, e->temb_own, st))) return rc;
      e->temb_own_map = s->timestep_map;
    }
    temb = e->temb_own;
  }
  // The conditioning rows of the chain are produced in chunks of <= 256 rows (= 256 / bu steps), right before the steps
  // that use them: the adaLN outputs are 783 KB per row at XL/2 (nmod = 195 840 fp32), so the whole chain at once would
  // be 3.1 GB for 250 guided steps at B = 16.  The adaLN weights are streamed once per 64 rows either way.
  const int chunk_steps = std::max(1, 256 / bu);
  const int64_t rows_chunk = (int64_t)std::min(chunk_steps, n_run) * bu;
  if ((rc = grow(e, &e->cond_rows, &e->cond_cap, rows_chunk * D))) return rc;
  if ((rc = grow(e, &e->mod_all, &e->mod_all_cap, rows_chunk * e->nmod))) return rc;
  if (ex == 78 && (rc = grow(e, &e->cond_rows_t, &e->cond_t_cap, rows_chunk * D))) return rc;
  const size_t fo = (size_t)e->cfg.depth * 6 * D;
  // conditioning of respaced steps [lo, lo + cnt): rows ordered by index ascending, row (i - lo) * bu + b
  auto chain_conditioning = [&](int lo, int cnt) -> int {
    const int64_t rows_all = (int64_t)cnt * bu;
    int r2;
    if ((r2 = launch_cond_rows(temb + (size_t)lo * D, ex == 2 ? e->ytab : ex == 78 ? e->txt_proj : nullptr,
                               ex == 78 ? e->iota : y, e->cond_rows, cnt, bu, D, st))) return r2;
    if (ex == 78 &&   // t-only rows for the final layer (latte.py:372-373)
        (r2 = launch_cond_rows(temb + (size_t)lo * D, nullptr, nullptr, e->cond_rows_t, cnt, bu, D, st))) return r2;
    for (int64_t r0 = 0; r0 < rows_all; r0 += 64) {
      const int rows = (int)std::min<int64_t>(64, rows_all - r0);
      if ((r2 = launch_small_linear(IN_PLAIN, e->cond_rows + (size_t)r0 * D, nullptr, e->ada_w, e->ada_b, nullptr, nullptr,
                                    e->mod_all + (size_t)r0 * e->nmod, rows, e->nmod, D, e->nmod, st))) return r2;
      if (ex == 78 &&
          (r2 = launch_small_linear(IN_PLAIN, e->cond_rows_t + (size_t)r0 * D, nullptr, e->ada_w + fo * D, e->ada_b + fo, nullptr,
                                    nullptr, e->mod_all + (size_t)r0 * e->nmod + fo, rows, 2 * D, D, e->nmod, st))) return r2;
    }
    return LATTE_OK;
  };
  const size_t numel = (size_t)batch * e->F * e->Cin * e->H * e->H;
  int k = 0;
  int chunk_lo = start_index + 1;   // first respaced index covered by the rows currently in mod_all
  for (int i = start_index; i >= end_index; --i, ++k) {
    if (i < chunk_lo) {   // next chunk: steps i, i - 1, ..., down to max(end_index, i - chunk_steps + 1)
      chunk_lo = std::max(end_index, i - chunk_steps + 1);
      if ((rc = chain_conditioning(chunk_lo, i - chunk_lo + 1))) return rc;
    }
    const float* mod_i = e->mod_all + (size_t)(i - chunk_lo) * bu * e->nmod;
    if ((rc = run_forward(e, x, nullptr, y, batch, use_cfg, e->model_out, st, nullptr, mod_i, bu == 1 ? 0 : e->nmod))) return rc;
    SamplerCoefs c = make_coefs(s, method, i, eta, clip_denoised);
    c.cfg_scale = cfg_scale;
    const bool need_noise = (method == LATTE_METHOD_DDPM) ? (i != 0) : (c.sigma != 0.0f && i != 0);
    const float* nz = nullptr;
    if (need_noise) {
      if (noise) {
        nz = noise + (size_t)k * numel;
      } else {
        if ((rc = launch_fill_normal(e->noise_buf, numel, e->seed, e->rng_offset, st))) return rc;
        e->rng_offset += numel;
        nz = e->noise_buf;
      }
    }
    float* x0 = trail_x0 ? trail_x0 + (size_t)k * numel : nullptr;
    if ((rc = launch_sampler_update(c, x, e->model_out, nz, batch, e->F, e->Cin, e->H * e->H, use_cfg ? 1 : 0, x, x0, st))) return rc;
    if (trail_sample)
      LATTE_HIP(hipMemcpyAsync(trail_sample + (size_t)k * numel, x, sizeof(float) * numel, hipMemcpyDeviceToDevice, st));
  }
  return LATTE_OK;
}

// fp32 device copies of the schedule tables for the batched-timestep kernels (one device per schedule object)
int schedule_device_tables(const latte_schedule_t* s, const float** out, hipStream_t st) {
  int dev = 0;
  LATTE_HIP(hipGetDevice(&dev));
  const int n = s->num_timesteps;
  if (s->dev_tables && s->dev_tables_device == dev) {
    *out = s->dev_tables;
    return LATTE_OK;
  }
  if (n < 2) return fail(LATTE_ERR_INVALID, "training: needs >= 2 timesteps (posterior_log_variance_clipped)");
  if (s->dev_tables) {
    (void)hipFree(s->dev_tables);
    s->dev_tables = nullptr;
  }
  std::vector<float> h((size_t)LATTE_NUM_DEV_TABLES * n);
  auto put = [&](int which, const std::vector<double>& v) {
    for (int i = 0; i < n; ++i) h[(size_t)which * n + i] = (float)v[i];
  };
  put(DT_SQRT_AC, s->sqrt_alphas_cumprod);
  put(DT_SQRT_1MAC, s->sqrt_one_minus_alphas_cumprod);
  put(DT_COEF1, s->posterior_mean_coef1);
  put(DT_COEF2, s->posterior_mean_coef2);
  put(DT_POST_LOGVAR, s->posterior_log_variance_clipped);
  put(DT_LOG_BETAS, s->log_betas);
  put(DT_SQRT_RECIP, s->sqrt_recip_alphas_cumprod);
  put(DT_SQRT_RECIPM1, s->sqrt_recipm1_alphas_cumprod);
  for (int i = 0; i < n; ++i)   // gd:298-313: FIXED_LARGE = log(append(posterior_variance[1], betas[1:])), else the clipped posterior
    h[(size_t)DT_FIXED_LOGVAR * n + i] = s->var_type == 1 ? (float)std::log(i == 0 ? s->posterior_variance[1] : s->betas[i])
                                                          : (float)s->posterior_log_variance_clipped[i];
  float* d = nullptr;
  LATTE_HIP(hipMalloc((void**)&d, h.size() * sizeof(float)));
  LATTE_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  (void)st;
  s->dev_tables = d;
  s->dev_tables_device = dev;
  *out = d;
  return LATTE_OK;
}

int latte_q_sample(const latte_schedule_t* s, const float* x_start, const float* noise, const int64_t* t, int batch,
                   int64_t numel_per_sample, float* x_t, void* stream) {
  if (!s || !x_start || !noise || !t || !x_t || batch <= 0 || numel_per_sample <= 0)
    return fail(LATTE_ERR_INVALID, "q_sample: bad arguments");
  const float* tab = nullptr;
  int rc = schedule_device_tables(s, &tab, (hipStream_t)stream);
  if (rc) return rc;
  return launch_q_sample(tab, s->num_timesteps, x_start, noise, t, batch, (size_t)numel_per_sample, x_t, (hipStream_t)stream);
}

int latte_training_losses(const latte_schedule_t* s, int loss_type, const float* x_start, const float* x_t, const float* noise,
                          const float* model_out, const int64_t* t, int batch, int frames, int channels, int hw, float* workspace,
                          int64_t workspace_floats, float* mse_out, float* vb_out, float* loss_out, void* stream) {
  if (!s || !x_start || !x_t || !noise || !model_out || !t || !workspace || !loss_out || batch <= 0)
    return fail(LATTE_ERR_INVALID, "training_losses: bad arguments");
  if (loss_type < 0 || loss_type > 3) return fail(LATTE_ERR_INVALID, "training_losses: loss_type must be 0 MSE, 1 RESCALED_MSE, 2 KL, 3 RESCALED_KL");
  const size_t per = (size_t)frames * channels * hw;
  const int blocks = training_terms_blocks(per);
  const int64_t need = (int64_t)batch * blocks * 2 + 2 * (int64_t)batch;
  if (workspace_floats < need)
    return fail(LATTE_ERR_INVALID, "training_losses: workspace too small (latte_training_workspace_floats)");
  const float* tab = nullptr;
  hipStream_t st = (hipStream_t)stream;
  int rc = schedule_device_tables(s, &tab, st);
  if (rc) return rc;
  float* partial = workspace;
  float* mse = workspace + (size_t)batch * blocks * 2;
  float* vb = mse + batch;
  if ((rc = launch_training_terms(tab, s->num_timesteps, s->mean_type, s->var_type, x_start, x_t, noise, model_out, t, batch, frames,
                                  channels, hw, partial, blocks, mse, vb, st))) return rc;
  // gd:741-793: which terms exist and how they are scaled
  const bool kl = loss_type >= 2;
  const bool has_vb = kl || s->var_type == 0;
  float vb_scale = 1.0f;
  if (loss_type == 3) vb_scale = (float)s->num_timesteps;                       // RESCALED_KL, gd:751-752
  if (loss_type == 1) vb_scale = (float)(s->num_timesteps / 1000.0);            // RESCALED_MSE, gd:771-774
  return launch_training_combine(mse, vb, has_vb ? 1 : 0, kl ? 1 : 0, vb_scale, batch, mse_out, vb_out, loss_out, st);
}

int64_t latte_training_workspace_floats(int batch, int64_t numel_per_sample) {
  if (batch <= 0 || numel_per_sample <= 0) return 0;
  return (int64_t)batch * training_terms_blocks((size_t)numel_per_sample) * 2 + 2 * (int64_t)batch;
}

int latte_profile_forward(latte_engine_t* e, const float* x, const int64_t* t, const int64_t* y, int batch, float* out,
                          float* ms_out, int* launches_out, int n, void* stream) {
  return latte_profile_forward_ex(e, x, t, y, batch, 0, out, ms_out, launches_out, n, stream);
}

int latte_profile_forward_ex(latte_engine_t* e, const float* x, const int64_t* t, const int64_t* y, int batch, int guided, float* out,
                             float* ms_out, int* launches_out, int n, void* stream) {
  if (!e || !ms_out || !launches_out || n < LATTE_NUM_KERNEL_CLASSES) return fail(LATTE_ERR_INVALID, "profile_forward: bad arguments");
  int rc = latte_engine_check_weights(e);
  if (rc) return rc;
  Prof prof;
  hipStream_t st = (hipStream_t)stream;
  rc = run_forward(e, x, t, y, batch, guided != 0, out, st, &prof);
  if (rc) return rc;
  LATTE_HIP(hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i) {
    ms_out[i] = 0.f;
    launches_out[i] = 0;
  }
  for (size_t i = 1; i < prof.ev.size(); ++i) {
    float ms = 0.f;
    LATTE_HIP(hipEventElapsedTime(&ms, prof.ev[i - 1], prof.ev[i]));
    const int c = prof.cls[i];
    if (c >= 0 && c < n) {
      ms_out[c] += ms;
      launches_out[c] += 1;
    }
  }
  for (auto ev : prof.ev) (void)hipEventDestroy(ev);
  return LATTE_OK;
}

int latte_bench_gemm(int M, int N, int K, int epi, int dtype, int variant, int iters, float* ms_per_launch, void* stream) {
  int stagger = 0, tag = 0;
  if (variant >= 1000) { tag = variant / 1000; variant %= 1000; }    // measurement: + 1000 * call-site tag (GemmArgs::tag)
  if (variant >= 100) { stagger = variant / 100; variant %= 100; }   // measurement: variant + 100 * cohorts
  if (M <= 0 || N <= 0 || K <= 0 || iters <= 0 || !ms_per_launch) return fail(LATTE_ERR_INVALID, "bench_gemm: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t Mp = ((int64_t)M + 255) / 256 * 256;
  half_t *A = nullptr, *W = nullptr;
  float *bias = nullptr, *gate = nullptr, *tmp = nullptr;
  void* out = nullptr;
  const size_t out_bytes = (size_t)Mp * N * 4;
  LATTE_HIP(hipMalloc((void**)&A, (size_t)Mp * K * 2));
  LATTE_HIP(hipMalloc((void**)&W, (size_t)N * K * 2));
  LATTE_HIP(hipMalloc((void**)&bias, (size_t)N * 4));
  LATTE_HIP(hipMalloc((void**)&gate, (size_t)N * 4));
  LATTE_HIP(hipMalloc(&out, out_bytes));
  const size_t big = (size_t)Mp * K > (size_t)N * K ? (size_t)Mp * K : (size_t)N * K;
  LATTE_HIP(hipMalloc((void**)&tmp, big * 4));
  LATTE_HIP(hipMemsetAsync(out, 0, out_bytes, st));
  int rc = LATTE_OK;
  // uniform-ish random operands (guide §5.4 rule 25: never bench MFMA kernels on zero-filled data)
  if (!rc) rc = launch_fill_normal(tmp, (size_t)Mp * K, 1234, 0, st);
  if (!rc) rc = launch_convert_f32_to_h16(tmp, A, (int64_t)Mp * K, dtype, st);
  if (!rc) rc = launch_fill_normal(tmp, (size_t)N * K, 99, 0, st);
  if (!rc) rc = launch_convert_f32_to_h16(tmp, W, (int64_t)N * K, dtype, st);
  if (!rc) rc = launch_fill_normal(bias, (size_t)N, 7, 0, st);
  if (!rc) rc = launch_fill_normal(gate, (size_t)N, 8, 0, st);
  GemmArgs g{};
  g.A = A; g.W = W; g.bias = bias; g.out = out; g.gate = gate; g.M = M; g.N = N; g.K = K;
  g.gate_stride = 0; g.rows_per_sample = M; g.stagger = stagger; g.tag = tag;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 3 && !rc; ++i) rc = launch_gemm(g, epi, dtype, variant, st);
  if (!rc) {
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < iters && !rc; ++i) rc = launch_gemm(g, epi, dtype, variant, st);
    (void)hipEventRecord(e1, st);
    hipError_t he = hipStreamSynchronize(st);
    if (he != hipSuccess) rc = fail(LATTE_ERR_HIP, std::string("bench_gemm: ") + hipGetErrorString(he));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *ms_per_launch = ms / iters;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(A); (void)hipFree(W); (void)hipFree(bias); (void)hipFree(gate); (void)hipFree(out); (void)hipFree(tmp);
  return rc;
}

}  // extern "C"
