// Host logic of the SD-VAE decoder: weight ingestion by diffusers state-dict key, workspace, decode().
//
//   AutoencoderKL.decode(z).sample      diffusers 0.24.0 (un-vendored; oracle/vae_oracle.py restates it)
//   called by the reference at          /root/reference/sample/sample.py:113-115, sample_ddp.py:165-168
//   uint8 video conversion              /root/reference/sample/sample.py:122
//
// Activation layout: NHWC.  The decoder's RESIDUAL STREAM (conv_in output, every ResnetBlock2D / attention / upsampler
// output) is fp32 -- two ping-pong buffers -- so the skip path is never rounded: with a half stream each of the ~20
// block outputs added one half rounding of the whole activation and the f16 decode ended at 1.35e-3 rel-L2 against the
// fp32 restatement (round 1).  conv1's output inside a ResnetBlock2D is fp32 as well (it only feeds GroupNorm 2).  Only
// the MFMA operands (GroupNorm+SiLU outputs, attention q/k/v/P, the half copies the shortcut / upsampler convs read) are
// half: three half scratch buffers, all sized for the largest map ([N, 8h, 8w, 256]).
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "common.h"

using namespace latte;

namespace {
struct KProf {
  std::vector<hipEvent_t> ev;
  std::vector<int> cls;
};
thread_local KProf* g_kprof = nullptr;
}  // namespace

namespace latte {
void kprof_mark(int cls, hipStream_t st) {
  if (!g_kprof) return;
  hipEvent_t ev;
  (void)hipEventCreate(&ev);
  (void)hipEventRecord(ev, st);
  g_kprof->ev.push_back(ev);
  g_kprof->cls.push_back(cls);
}
}  // namespace latte

namespace {

enum VPack { VP_F32, VP_CONV3, VP_LINEAR_H16, VP_SMALL_T, VP_SMALL, VP_CONVT };
struct VSlot {
  std::string key;
  int64_t numel;
  VPack kind;
  void* dst;
  int cout, cin;
  bool loaded = false;
  void* dst_lo = nullptr;   // VP_CONV3 / VP_LINEAR_H16: the f16 rounding residual of the packed weight (split-operand passes), or nullptr
};
struct Resnet {
  int cin, cout;
  float *n1w, *n1b, *n2w, *n2b, *c1b, *c2b, *scb = nullptr;
  half_t *c1w, *c2w, *scw = nullptr;
  half_t *c1w_lo = nullptr, *c2w_lo = nullptr, *scw_lo = nullptr;   // f16 rounding residuals of the weights (temporal decoder's split passes)
};
// TemporalResnetBlock + AlphaBlender of a SpatioTemporalResBlock (AutoencoderKLTemporalDecoder): Conv3d (3,1,1) weights as
// [C][3 C] half; conv2 and its bias are kept in fp32 too and folded with sigmoid(mix_factor) before the first decode
struct TResnet {
  int c = 0;
  float *n1w, *n1b, *n2w, *n2b, *c1b, *c2b, *c2b_eff, *c2w_f32, *mix;
  half_t *c1w, *c2w, *c1w_lo, *c2w_lo;   // _lo: the f16 rounding residual of the weights (split-operand convolution)
};

}  // namespace

struct latte_vae {
  int h = 0, max_frames = 0, dtype = 0;
  int ch[4] = {128, 256, 512, 512};   // block_out_channels
  std::vector<VSlot> slots;
  std::map<std::string, int> index;
  std::vector<void*> allocs;
  float *pq_w, *pq_b, *ci_wt, *ci_b, *co_w, *co_b, *no_w, *no_b;
  Resnet mid[2];
  Resnet up[4][3];
  half_t* upc_w[3];
  half_t* upc_w_lo[3] = {nullptr, nullptr, nullptr};   // temporal decoder: f16 rounding residual of the upsampler weights
  float* upc_b[3];
  float *agn_w, *agn_b, *aq_b, *ak_b, *av_b, *ao_b, *ao_b_eff, *zero_bias;
  float* ao_w_f32;     // to_out weight in fp32 (for the folded bias  Wo bv + bo)
  half_t *aq_w, *ak_w, *av_w, *ao_w;
  half_t* buf[3];      // half scratch (MFMA operands)
  float* sbuf[2];      // fp32 residual stream, ping-pong
  float* tbuf;         // fp32 conv1 output of a ResnetBlock2D (GroupNorm 2 normalises it before anything rounds it)
  float* ones;         // [512] gate vector of ones (attention out-projection through the gated fp32 residual epilogue)
  half_t* zeros;
  float *pq_out, *scores, *gn_partial, *gn_stats, *stage;
  int64_t stage_numel = 0;
  bool bias_folded = false;
  // AutoencoderKLTemporalDecoder mode (latte_vae_create_temporal): every resnet is a SpatioTemporalResBlock, no
  // post_quant_conv, time_conv_out after conv_out; one decode call = ONE video chunk of n_frames frames
  bool temporal = false;
  TResnet tmid[2], tup[4][3];
  float *tco_w = nullptr, *tco_b = nullptr;
};

namespace {

template <typename Tp>
int valloc(latte_vae* v, Tp** p, size_t count) {
  void* q = nullptr;
  const size_t bytes = count * sizeof(Tp);
  LATTE_HIP(hipMalloc(&q, bytes ? bytes : 16));
  LATTE_HIP(hipMemset(q, 0, bytes ? bytes : 16));
  v->allocs.push_back(q);
  *p = (Tp*)q;
  return LATTE_OK;
}

void vslot(latte_vae* v, const std::string& key, int64_t numel, VPack kind, void* dst, int cout = 0, int cin = 0) {
  VSlot s;
  s.key = key; s.numel = numel; s.kind = kind; s.dst = dst; s.cout = cout; s.cin = cin;
  v->index[key] = (int)v->slots.size();
  v->slots.push_back(s);
  if (numel > v->stage_numel) v->stage_numel = numel;
}

int make_resnet(latte_vae* v, Resnet& r, const std::string& p, int cin, int cout) {
  r.cin = cin; r.cout = cout;
  int rc;
  if ((rc = valloc(v, &r.n1w, cin)) || (rc = valloc(v, &r.n1b, cin)) || (rc = valloc(v, &r.n2w, cout)) ||
      (rc = valloc(v, &r.n2b, cout)) || (rc = valloc(v, &r.c1b, cout)) || (rc = valloc(v, &r.c2b, cout)) ||
      (rc = valloc(v, &r.c1w, (size_t)cout * cin * 9)) || (rc = valloc(v, &r.c2w, (size_t)cout * cout * 9)))
    return rc;
  vslot(v, p + "norm1.weight", cin, VP_F32, r.n1w);
  vslot(v, p + "norm1.bias", cin, VP_F32, r.n1b);
  vslot(v, p + "conv1.weight", (int64_t)cout * cin * 9, VP_CONV3, r.c1w, cout, cin);
  if ((rc = valloc(v, &r.c1w_lo, (size_t)cout * cin * 9)) || (rc = valloc(v, &r.c2w_lo, (size_t)cout * cout * 9))) return rc;
  v->slots.back().dst_lo = r.c1w_lo;
  vslot(v, p + "conv1.bias", cout, VP_F32, r.c1b);
  vslot(v, p + "norm2.weight", cout, VP_F32, r.n2w);
  vslot(v, p + "norm2.bias", cout, VP_F32, r.n2b);
  vslot(v, p + "conv2.weight", (int64_t)cout * cout * 9, VP_CONV3, r.c2w, cout, cout);
  v->slots.back().dst_lo = r.c2w_lo;
  vslot(v, p + "conv2.bias", cout, VP_F32, r.c2b);
  if (cin != cout) {
    if ((rc = valloc(v, &r.scw, (size_t)cout * cin)) || (rc = valloc(v, &r.scb, cout))) return rc;
    vslot(v, p + "conv_shortcut.weight", (int64_t)cout * cin, VP_LINEAR_H16, r.scw);
    if ((rc = valloc(v, &r.scw_lo, (size_t)cout * cin))) return rc;
    v->slots.back().dst_lo = r.scw_lo;
    vslot(v, p + "conv_shortcut.bias", cout, VP_F32, r.scb);
  }
  return LATTE_OK;
}

int make_tresnet(latte_vae* v, TResnet& t, const std::string& p, int c) {
  t.c = c;
  int rc;
  if ((rc = valloc(v, &t.n1w, c)) || (rc = valloc(v, &t.n1b, c)) || (rc = valloc(v, &t.n2w, c)) || (rc = valloc(v, &t.n2b, c)) ||
      (rc = valloc(v, &t.c1b, c)) || (rc = valloc(v, &t.c2b, c)) || (rc = valloc(v, &t.c2b_eff, c)) ||
      (rc = valloc(v, &t.c2w_f32, (size_t)c * c * 3)) || (rc = valloc(v, &t.mix, 4)) || (rc = valloc(v, &t.c1w, (size_t)c * c * 3)) ||
      (rc = valloc(v, &t.c2w, (size_t)c * c * 3)) || (rc = valloc(v, &t.c1w_lo, (size_t)c * c * 3)) ||
      (rc = valloc(v, &t.c2w_lo, (size_t)c * c * 3)))
    return rc;
  const std::string q = p + "temporal_res_block.";
  vslot(v, q + "norm1.weight", c, VP_F32, t.n1w);
  vslot(v, q + "norm1.bias", c, VP_F32, t.n1b);
  vslot(v, q + "conv1.weight", (int64_t)c * c * 3, VP_CONVT, &t, c, c);
  vslot(v, q + "conv1.bias", c, VP_F32, t.c1b);
  vslot(v, q + "norm2.weight", c, VP_F32, t.n2w);
  vslot(v, q + "norm2.bias", c, VP_F32, t.n2b);
  vslot(v, q + "conv2.weight", (int64_t)c * c * 3, VP_F32, t.c2w_f32);
  vslot(v, q + "conv2.bias", c, VP_F32, t.c2b);
  vslot(v, p + "time_mixer.mix_factor", 1, VP_F32, t.mix);
  return LATTE_OK;
}

// Which convolutions of the temporal decoder run as split-operand products (round 6).  Bits 0..4: the spatial resnets of {mid block, up block
// 0..3} add the pass on the activation's f16 rounding residual; bits 5..9: the temporal resnets of the same stages run hi*hi + lo*hi + hi*lo
// instead of one pass; bit 10: the 1x1 shortcuts' half copy of the stream as hi + lo (a second GEMM pass); bit 11: conv_out reads its input as
// hi + lo; bits 12..14: the upsampler convolution of up block 0..2 adds the pass on the residual of its half copy of the stream; bits 15..19: the
// spatial resnets of the five stages add the pass on the WEIGHTS' f16 rounding residual; bit 20: the shortcuts' weight residual; bits 21..23: the
// upsamplers' weight residual.
// latte_debug_set_choice("vae_split", (1 << 24) | mask) overrides the default (measurement / parity sweeps).
constexpr int VAE_SPLIT_DEFAULT_TEMPORAL = 0x319c03, VAE_SPLIT_DEFAULT_SPATIAL = 0x301c00;
int vae_split_mask(const latte_vae* v) {
  const int c = debug_choice(DBG_VAE_SPLIT);
  return (c >> 24) == 1 ? (c & 0xffffff) : (v->temporal ? VAE_SPLIT_DEFAULT_TEMPORAL : VAE_SPLIT_DEFAULT_SPATIAL);
}

int gemm_h16(const half_t* A, const half_t* W, const float* bias, void* out, const half_t* res, int M, int N, int K, int epi,
             int dtype, hipStream_t st) {
  GemmArgs g{};
  g.A = A; g.W = W; g.bias = bias; g.out = out; g.res = res; g.M = M; g.N = N; g.K = K; g.rows_per_sample = M;
  // plain 128 x 128 kernel: small, oddly shaped problems (round 6: the library's own tile choice for the 1x1 shortcuts of the large maps
  // measured the same decode time, profiles/r6_vae_split_sweep.log)
  const int rc_ = launch_gemm(g, epi, dtype, 1, st);
  kprof_mark(VC_ATTN, st);                              // (only the mid-block attention and the 1x1 shortcuts come through here)
  return rc_;
}

// ResnetBlock2D on the fp32 stream: x = *s -> *s (in place when cin == cout, else through *s2 and the two are swapped);
// b, c, d: half scratch.  [N, H, W, C]
int run_resnet(latte_vae* v, const Resnet& r, float** s, float** s2, half_t* b, half_t* c, half_t* d, int N, int H, int W,
               hipStream_t st, int stage) {
  int rc;
  const int HW = H * W, dt = v->dtype;
  float* x = *s;
  // temporal-decoder mode: the activation operand of both 3x3 convolutions is split hi + lo (b = the f16 rounding residual of the
  // GroupNorm output) and a second pass adds conv(lo): the decoder with twice as many blocks per stage stays under the 1e-3 bar
  // (round 6: per decoder stage -- split_mask bit `stage`, 0 = mid block, 1 + i = up block i; vae_split_mask())
  half_t* lo = ((vae_split_mask(v) >> stage) & 1) ? b : nullptr;
  const bool wlo = (vae_split_mask(v) >> (15 + stage)) & 1;   // + the pass hi * (weight residual)
  if (r.cin != r.cout) {   // conv_shortcut 1x1 = a GEMM over pixels on a half copy of the stream, fp32 result (first: b is free here)
    float* y = *s2;
    if ((vae_split_mask(v) >> 10) & 1) {   // the half copy as hi + lo: y = hi W^T + b, then y += lo W^T (gated-residual epilogue, gate = 1)
      if ((rc = launch_convert_f32_to_h16_split(x, d, b, (int64_t)N * HW * r.cin, dt, st))) return rc;
      kprof_mark(VC_SMALL, st);
      if ((rc = gemm_h16(d, r.scw, r.scb, y, nullptr, N * HW, r.cout, r.cin, EPI_BIAS_F32, dt, st))) return rc;
      GemmArgs g{};
      g.A = b; g.W = r.scw; g.bias = v->zero_bias; g.out = y; g.gate = v->ones; g.gate_stride = 0;
      g.M = N * HW; g.N = r.cout; g.K = r.cin; g.rows_per_sample = N * HW;
      const int sc_variant = 1;
      if ((rc = launch_gemm(g, EPI_GATE_RES_F32, dt, sc_variant, st))) return rc;
      kprof_mark(VC_ATTN, st);
      if (r.scw_lo && ((vae_split_mask(v) >> 20) & 1)) {   // + hi * (weight residual)
        g.A = d; g.W = r.scw_lo;
        if ((rc = launch_gemm(g, EPI_GATE_RES_F32, dt, sc_variant, st))) return rc;
        kprof_mark(VC_ATTN, st);
      }
    } else {
      if ((rc = launch_convert_f32_to_h16(x, d, (int64_t)N * HW * r.cin, dt, st))) return rc;
      kprof_mark(VC_SMALL, st);
      if ((rc = gemm_h16(d, r.scw, r.scb, y, nullptr, N * HW, r.cout, r.cin, EPI_BIAS_F32, dt, st))) return rc;
    }
  }
  if ((rc = launch_groupnorm(x, 1, c, r.n1w, r.n1b, v->gn_partial, v->gn_stats, N, HW, r.cin, 1, dt, st, 1e-6f, groupnorm_max_slabs(), lo))) return rc;
  if ((rc = launch_conv3x3(c, r.c1w, r.c1b, nullptr, nullptr, v->zeros, N, H, W, r.cin, r.cout, 0, dt, st, nullptr, v->tbuf))) return rc;
  if (lo && (rc = launch_conv3x3(lo, r.c1w, v->zero_bias, nullptr, nullptr, v->zeros, N, H, W, r.cin, r.cout, 0, dt, st, v->tbuf, v->tbuf))) return rc;
  if (wlo && (rc = launch_conv3x3(c, r.c1w_lo, v->zero_bias, nullptr, nullptr, v->zeros, N, H, W, r.cin, r.cout, 0, dt, st, v->tbuf, v->tbuf))) return rc;
  if ((rc = launch_groupnorm(v->tbuf, 1, c, r.n2w, r.n2b, v->gn_partial, v->gn_stats, N, HW, r.cout, 1, dt, st, 1e-6f, groupnorm_max_slabs(), lo))) return rc;
  if (r.cin != r.cout) {
    float* y = *s2;
    if ((rc = launch_conv3x3(c, r.c2w, r.c2b, nullptr, nullptr, v->zeros, N, H, W, r.cout, r.cout, 0, dt, st, y, y))) return rc;
    if (lo && (rc = launch_conv3x3(lo, r.c2w, v->zero_bias, nullptr, nullptr, v->zeros, N, H, W, r.cout, r.cout, 0, dt, st, y, y))) return rc;
    if (wlo && (rc = launch_conv3x3(c, r.c2w_lo, v->zero_bias, nullptr, nullptr, v->zeros, N, H, W, r.cout, r.cout, 0, dt, st, y, y))) return rc;
    std::swap(*s, *s2);
    return LATTE_OK;
  }
  if ((rc = launch_conv3x3(c, r.c2w, r.c2b, nullptr, nullptr, v->zeros, N, H, W, r.cout, r.cout, 0, dt, st, x, x))) return rc;
  if (lo && (rc = launch_conv3x3(lo, r.c2w, v->zero_bias, nullptr, nullptr, v->zeros, N, H, W, r.cout, r.cout, 0, dt, st, x, x))) return rc;
  if (wlo) return launch_conv3x3(c, r.c2w_lo, v->zero_bias, nullptr, nullptr, v->zeros, N, H, W, r.cout, r.cout, 0, dt, st, x, x);
  return LATTE_OK;
}

// TemporalResnetBlock + AlphaBlender on the fp32 stream of ONE video [T, H, W, C]: the frames are the rows of an "image"
// [T][H W], so the Conv3d (3,1,1) is the implicit-GEMM conv kernel with 3 taps along the rows and GroupNorm sees T H W pixels
int run_tresnet(latte_vae* v, const TResnet& t, float* x, half_t* c, half_t* clo, int T, int H, int W, hipStream_t st, int stage) {
  // The two Conv3d run as SPLIT-OPERAND convolutions: activation = hi + lo and weight = hi + lo in f16 (lo = the rounding
  // residual), three MFMA passes hi*hi + lo*hi + hi*lo accumulated in the fp32 output -- the temporal branch then adds ~1e-6 of
  // error instead of one more f16 roundoff per block (with single-pass convolutions the decode measured 1.04e-3 against the
  // fp32 restatement, above the 1e-3 bar; a 3-tap convolution costs a third of a 3x3 one, so three passes cost one)
  int rc;
  const int HW = H * W, dt = v->dtype, C = t.c;
  if (!((vae_split_mask(v) >> (5 + stage)) & 1)) {   // single-pass temporal convolutions at this stage (split_mask bit 5 + stage clear)
    if ((rc = launch_groupnorm(x, 1, c, t.n1w, t.n1b, v->gn_partial, v->gn_stats, 1, T * HW, C, 1, dt, st, 1e-5f, groupnorm_max_slabs() * T, nullptr))) return rc;
    if ((rc = launch_conv3x3(c, t.c1w, t.c1b, nullptr, nullptr, v->zeros, 1, T, HW, C, C, 0, dt, st, nullptr, v->tbuf, 1))) return rc;
    if ((rc = launch_groupnorm(v->tbuf, 1, c, t.n2w, t.n2b, v->gn_partial, v->gn_stats, 1, T * HW, C, 1, dt, st, 1e-5f, groupnorm_max_slabs() * T, nullptr))) return rc;
    return launch_conv3x3(c, t.c2w, t.c2b_eff, nullptr, nullptr, v->zeros, 1, T, HW, C, C, 0, dt, st, x, x, 1);
  }
  if ((rc = launch_groupnorm(x, 1, c, t.n1w, t.n1b, v->gn_partial, v->gn_stats, 1, T * HW, C, 1, dt, st, 1e-5f, groupnorm_max_slabs() * T, clo))) return rc;
  if ((rc = launch_conv3x3(c, t.c1w, t.c1b, nullptr, nullptr, v->zeros, 1, T, HW, C, C, 0, dt, st, nullptr, v->tbuf, 1))) return rc;
  if ((rc = launch_conv3x3(clo, t.c1w, v->zero_bias, nullptr, nullptr, v->zeros, 1, T, HW, C, C, 0, dt, st, v->tbuf, v->tbuf, 1))) return rc;
  if ((rc = launch_conv3x3(c, t.c1w_lo, v->zero_bias, nullptr, nullptr, v->zeros, 1, T, HW, C, C, 0, dt, st, v->tbuf, v->tbuf, 1))) return rc;
  if ((rc = launch_groupnorm(v->tbuf, 1, c, t.n2w, t.n2b, v->gn_partial, v->gn_stats, 1, T * HW, C, 1, dt, st, 1e-5f, groupnorm_max_slabs() * T, clo))) return rc;
  // out = x_spatial + sigmoid(mix) * (conv2 + bias): weights and bias were folded with the blend factor
  if ((rc = launch_conv3x3(c, t.c2w, t.c2b_eff, nullptr, nullptr, v->zeros, 1, T, HW, C, C, 0, dt, st, x, x, 1))) return rc;
  if ((rc = launch_conv3x3(clo, t.c2w, v->zero_bias, nullptr, nullptr, v->zeros, 1, T, HW, C, C, 0, dt, st, x, x, 1))) return rc;
  return launch_conv3x3(c, t.c2w_lo, v->zero_bias, nullptr, nullptr, v->zeros, 1, T, HW, C, C, 0, dt, st, x, x, 1);
}

int vae_create_impl(int latent_size, int max_frames, int compute_dtype, bool temporal, latte_vae_t** out);

}  // namespace

extern "C" {

int latte_vae_create(int latent_size, int max_frames, int compute_dtype, latte_vae_t** out) {
  return vae_create_impl(latent_size, max_frames, compute_dtype, false, out);
}
int latte_vae_create_temporal(int latent_size, int max_frames, int compute_dtype, latte_vae_t** out) {
  return vae_create_impl(latent_size, max_frames, compute_dtype, true, out);
}

}  // extern "C"

namespace {
int vae_create_impl(int latent_size, int max_frames, int compute_dtype, bool temporal, latte_vae_t** out) {
  if (!out || latent_size <= 0 || max_frames <= 0) return fail(LATTE_ERR_INVALID, "vae_create: bad arguments");
  if (compute_dtype != LATTE_DTYPE_F16)
    return fail(LATTE_ERR_INVALID, "vae_create: the decoder runs f16 MFMA operands only (the reference decodes in fp16, sample.py:74; "
                                   "bf16 operands measured 7e-3 against the fp32 restatement and are not offered)");
  if (latent_size % 16 != 0 || latent_size > 64)
    return fail(LATTE_ERR_INVALID, "vae_create: latent_size must be a multiple of 16, at most 64");
  auto* v = new latte_vae();
  v->h = latent_size; v->max_frames = max_frames; v->dtype = compute_dtype; v->temporal = temporal;
  const int top = v->ch[3];
  const std::string sp = temporal ? "spatial_res_block." : "";
  int rc = LATTE_OK;
#define TRY(x) do { if ((rc = (x))) { latte_vae_destroy(v); return rc; } } while (0)
  TRY(valloc(v, &v->pq_w, 16)); TRY(valloc(v, &v->pq_b, 4));
  TRY(valloc(v, &v->ci_wt, (size_t)36 * top)); TRY(valloc(v, &v->ci_b, top));
  TRY(valloc(v, &v->co_w, (size_t)27 * v->ch[0])); TRY(valloc(v, &v->co_b, 4));
  TRY(valloc(v, &v->no_w, v->ch[0])); TRY(valloc(v, &v->no_b, v->ch[0]));
  if (!temporal) {
    vslot(v, "post_quant_conv.weight", 16, VP_F32, v->pq_w);
    vslot(v, "post_quant_conv.bias", 4, VP_F32, v->pq_b);
  } else {   // AutoencoderKLTemporalDecoder has no post_quant_conv: the 1x1 kernel runs with the identity
    const float eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    LATTE_HIP(hipMemcpy(v->pq_w, eye, sizeof(eye), hipMemcpyHostToDevice));
  }
  vslot(v, "decoder.conv_in.weight", (int64_t)top * 36, VP_SMALL_T, v->ci_wt, top, 4);
  vslot(v, "decoder.conv_in.bias", top, VP_F32, v->ci_b);
  TRY(make_resnet(v, v->mid[0], "decoder.mid_block.resnets.0." + sp, top, top));
  if (temporal) TRY(make_tresnet(v, v->tmid[0], "decoder.mid_block.resnets.0.", top));
  {
    const std::string a = "decoder.mid_block.attentions.0.";
    TRY(valloc(v, &v->agn_w, top)); TRY(valloc(v, &v->agn_b, top));
    TRY(valloc(v, &v->aq_b, top)); TRY(valloc(v, &v->ak_b, top)); TRY(valloc(v, &v->av_b, top)); TRY(valloc(v, &v->ao_b, top));
    TRY(valloc(v, &v->ao_b_eff, top)); TRY(valloc(v, &v->zero_bias, 4096));
    TRY(valloc(v, &v->aq_w, (size_t)top * top)); TRY(valloc(v, &v->ak_w, (size_t)top * top));
    TRY(valloc(v, &v->av_w, (size_t)top * top)); TRY(valloc(v, &v->ao_w, (size_t)top * top));
    TRY(valloc(v, &v->ao_w_f32, (size_t)top * top));
    vslot(v, a + "group_norm.weight", top, VP_F32, v->agn_w);
    vslot(v, a + "group_norm.bias", top, VP_F32, v->agn_b);
    vslot(v, a + "to_q.weight", (int64_t)top * top, VP_LINEAR_H16, v->aq_w);
    vslot(v, a + "to_q.bias", top, VP_F32, v->aq_b);
    vslot(v, a + "to_k.weight", (int64_t)top * top, VP_LINEAR_H16, v->ak_w);
    vslot(v, a + "to_k.bias", top, VP_F32, v->ak_b);
    vslot(v, a + "to_v.weight", (int64_t)top * top, VP_LINEAR_H16, v->av_w);
    vslot(v, a + "to_v.bias", top, VP_F32, v->av_b);
    vslot(v, a + "to_out.0.weight", (int64_t)top * top, VP_LINEAR_H16, v->ao_w);
    vslot(v, a + "to_out.0.bias", top, VP_F32, v->ao_b);
  }
  TRY(make_resnet(v, v->mid[1], "decoder.mid_block.resnets.1." + sp, top, top));
  if (temporal) TRY(make_tresnet(v, v->tmid[1], "decoder.mid_block.resnets.1.", top));
  int prev = top;
  for (int i = 0; i < 4; ++i) {
    const int cout = v->ch[3 - i];
    for (int r = 0; r < 3; ++r) {
      const std::string rp = "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(r) + ".";
      TRY(make_resnet(v, v->up[i][r], rp + sp, r == 0 ? prev : cout, cout));
      if (temporal) TRY(make_tresnet(v, v->tup[i][r], rp, cout));
    }
    prev = cout;
    if (i < 3) {
      TRY(valloc(v, &v->upc_w[i], (size_t)cout * cout * 9));
      TRY(valloc(v, &v->upc_b[i], cout));
      const std::string p = "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv.";
      vslot(v, p + "weight", (int64_t)cout * cout * 9, VP_CONV3, v->upc_w[i], cout, cout);
      TRY(valloc(v, &v->upc_w_lo[i], (size_t)cout * cout * 9));
      v->slots.back().dst_lo = v->upc_w_lo[i];
      vslot(v, p + "bias", cout, VP_F32, v->upc_b[i]);
    }
  }
  vslot(v, "decoder.conv_norm_out.weight", v->ch[0], VP_F32, v->no_w);
  vslot(v, "decoder.conv_norm_out.bias", v->ch[0], VP_F32, v->no_b);
  vslot(v, "decoder.conv_out.weight", (int64_t)27 * v->ch[0], VP_SMALL, v->co_w, 3, v->ch[0]);
  vslot(v, "decoder.conv_out.bias", 3, VP_F32, v->co_b);
  if (temporal) {
    TRY(valloc(v, &v->tco_w, 27)); TRY(valloc(v, &v->tco_b, 4));
    vslot(v, "decoder.time_conv_out.weight", 27, VP_F32, v->tco_w);      // [3, 3, 3, 1, 1] = (co, ci, tap)
    vslot(v, "decoder.time_conv_out.bias", 3, VP_F32, v->tco_b);
  }

  // workspace: the largest NHWC map is [N, 8h, 8w, 256] (output of up_blocks.2's upsampler)
  const size_t big = (size_t)max_frames * (8 * latent_size) * (8 * latent_size) * 256;
  for (int i = 0; i < 3; ++i) TRY(valloc(v, &v->buf[i], big));
  for (int i = 0; i < 2; ++i) TRY(valloc(v, &v->sbuf[i], big));
  TRY(valloc(v, &v->tbuf, big));
  TRY(valloc(v, &v->ones, 512));
  {
    std::vector<float> one(512, 1.0f);
    LATTE_HIP(hipMemcpy(v->ones, one.data(), sizeof(float) * 512, hipMemcpyHostToDevice));
  }
  TRY(valloc(v, &v->zeros, 64));
  TRY(valloc(v, &v->pq_out, (size_t)max_frames * latent_size * latent_size * 4));
  const size_t L = (size_t)latent_size * latent_size;
  TRY(valloc(v, &v->scores, L * L));
  TRY(valloc(v, &v->gn_partial, (size_t)max_frames * groupnorm_max_slabs() * 64));
  TRY(valloc(v, &v->gn_stats, (size_t)max_frames * 64));
  TRY(valloc(v, &v->stage, (size_t)v->stage_numel));
#undef TRY
  *out = v;
  return LATTE_OK;
}
}  // namespace

extern "C" {

void latte_vae_destroy(latte_vae_t* v) {
  if (!v) return;
  for (void* p : v->allocs) (void)hipFree(p);
  delete v;
}

int latte_vae_num_keys(const latte_vae_t* v) { return v ? (int)v->slots.size() : 0; }
const char* latte_vae_key(const latte_vae_t* v, int i) {
  if (!v || i < 0 || i >= (int)v->slots.size()) return nullptr;
  return v->slots[i].key.c_str();
}

int latte_vae_load_tensor(latte_vae_t* v, const char* key, const float* data, int64_t numel, int on_device, void* stream) {
  if (!v || !key || !data) return fail(LATTE_ERR_INVALID, "vae_load_tensor: null argument");
  auto it = v->index.find(key);
  if (it == v->index.end()) return fail(LATTE_ERR_INVALID, std::string("vae_load_tensor: unexpected key '") + key + "'");
  VSlot& s = v->slots[it->second];
  if (numel != s.numel)
    return fail(LATTE_ERR_INVALID, std::string("vae_load_tensor: size mismatch for '") + key + "': got " +
                                       std::to_string(numel) + ", expected " + std::to_string(s.numel));
  hipStream_t st = (hipStream_t)stream;
  const float* src = data;
  if (!on_device) {
    LATTE_HIP(hipMemcpyAsync(v->stage, data, sizeof(float) * numel, hipMemcpyHostToDevice, st));
    src = v->stage;
  }
  int rc = LATTE_OK;
  switch (s.kind) {
    case VP_F32: LATTE_HIP(hipMemcpyAsync(s.dst, src, sizeof(float) * numel, hipMemcpyDeviceToDevice, st)); break;
    case VP_CONV3: rc = launch_pack_conv_w(src, (half_t*)s.dst, s.cout, s.cin, v->dtype, st, (half_t*)s.dst_lo); break;
    case VP_LINEAR_H16:
      rc = s.dst_lo ? launch_convert_f32_to_h16_split(src, (half_t*)s.dst, (half_t*)s.dst_lo, numel, v->dtype, st)
                    : launch_convert_f32_to_h16(src, (half_t*)s.dst, numel, v->dtype, st);
      break;
    case VP_SMALL_T: rc = launch_pack_small_w(src, (float*)s.dst, s.cout, s.cin, 1, st); break;
    case VP_SMALL: rc = launch_pack_small_w(src, (float*)s.dst, s.cout, s.cin, 0, st); break;
    case VP_CONVT: {
      TResnet* t = (TResnet*)s.dst;
      rc = launch_pack_conv_t(src, t->c1w, s.cout, s.cin, nullptr, v->dtype, st, t->c1w_lo);
      break;
    }
  }
  if (rc) return rc;
  if (s.key == "decoder.mid_block.attentions.0.to_out.0.weight")
    LATTE_HIP(hipMemcpyAsync(v->ao_w_f32, src, sizeof(float) * numel, hipMemcpyDeviceToDevice, st));
  if (!on_device) LATTE_HIP(hipStreamSynchronize(st));
  s.loaded = true;
  v->bias_folded = false;
  return LATTE_OK;
}

int latte_vae_check_weights(latte_vae_t* v) {
  if (!v) return fail(LATTE_ERR_INVALID, "vae_check_weights: null");
  for (const auto& s : v->slots)
    if (!s.loaded) return fail(LATTE_ERR_STATE, "Missing key(s) in state_dict: \"" + s.key + "\"");
  return LATTE_OK;
}

static int vae_decode_impl(latte_vae_t* v, const float* z, int n_frames, float z_scale, int out_mode, void* out, void* stream,
                           int stop_after, float* trace_out, int64_t* trace_numel, int* trace_dims);

// One decode with a HIP event behind every launch: ms_out[c] / launches_out[c] per VaeKernelClass (conv3x3, GroupNorm statistics,
// GroupNorm apply, mid-block attention + 1x1 shortcut GEMMs, small kernels).  Synchronises the stream.
int latte_vae_profile_decode(latte_vae_t* v, const float* z, int n_frames, float z_scale, int out_mode, void* out, float* ms_out,
                             int* launches_out, int n, void* stream) {
  if (!ms_out || !launches_out || n < VC_NUM_CLASSES) return fail(LATTE_ERR_INVALID, "vae_profile_decode: bad arguments");
  KProf prof;
  g_kprof = &prof;
  kprof_mark(VC_START, (hipStream_t)stream);
  int rc = vae_decode_impl(v, z, n_frames, z_scale, out_mode, out, stream, -1, nullptr, nullptr, nullptr);
  g_kprof = nullptr;
  if (!rc && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = fail(LATTE_ERR_HIP, "vae_profile_decode: device error");
  for (int i = 0; i < n; ++i) { ms_out[i] = 0.f; launches_out[i] = 0; }
  for (size_t i = 1; !rc && i < prof.ev.size(); ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, prof.ev[i - 1], prof.ev[i]) != hipSuccess) { rc = fail(LATTE_ERR_HIP, "vae_profile_decode: event"); break; }
    const int c = prof.cls[i];
    if (c >= 0 && c < n) { ms_out[c] += ms; launches_out[c] += 1; }
  }
  for (auto ev : prof.ev) (void)hipEventDestroy(ev);
  return rc;
}

int latte_vae_decode(latte_vae_t* v, const float* z, int n_frames, float z_scale, int out_mode, void* out, void* stream) {
  return vae_decode_impl(v, z, n_frames, z_scale, out_mode, out, stream, -1, nullptr, nullptr, nullptr);
}

/* test hook (include/latte_amd_debug.h): run the decoder up to and including stage `stop_after` and return that stage's
 * NHWC activation as fp32. */
int latte_debug_vae_trace(latte_vae_t* v, const float* z, int n_frames, float z_scale, int stop_after, float* trace_out,
                          int64_t* trace_numel, int* trace_dims, void* stream) {
  if (!trace_out || !trace_numel || !trace_dims || stop_after < 0) return fail(LATTE_ERR_INVALID, "vae_trace: bad arguments");
  return vae_decode_impl(v, z, n_frames, z_scale, 0, trace_out, stream, stop_after, trace_out, trace_numel, trace_dims);
}

static int vae_decode_impl(latte_vae_t* v, const float* z, int n_frames, float z_scale, int out_mode, void* out, void* stream,
                           int stop_after, float* trace_out, int64_t* trace_numel, int* trace_dims) {
  if (!v || !z || !out) return fail(LATTE_ERR_INVALID, "vae_decode: null argument");
  if (n_frames <= 0 || n_frames > v->max_frames) return fail(LATTE_ERR_STATE, "vae_decode: n_frames exceeds max_frames");
  if (out_mode != 0 && out_mode != 1) return fail(LATTE_ERR_INVALID, "vae_decode: out_mode must be 0 (fp32 NCHW) or 1 (uint8 NHWC)");
  int rc = latte_vae_check_weights(v);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int N = n_frames, dt = v->dtype, top = v->ch[3];
  int H = v->h, W = v->h;
  if (!v->bias_folded) {
    // softmax rows sum to 1, so  to_out(P (V0 + 1 bv^T)) = to_out(P V0) + (Wo bv + bo): the value bias is folded into
    // the output bias and V^T is produced directly by a GEMM (no transpose kernel)
    if ((rc = launch_small_linear(IN_PLAIN, v->av_b, nullptr, v->ao_w_f32, v->ao_b, nullptr, nullptr, v->ao_b_eff, 1, top, top, top, st))) return rc;
    if (v->temporal) {   // AlphaBlender folded into conv2 of every temporal resnet: out = x_spatial + sigmoid(mix) (conv2 + b)
      auto fold = [&](TResnet& t) -> int {
        int r2 = launch_pack_conv_t(t.c2w_f32, t.c2w, t.c, t.c, t.mix, dt, st, t.c2w_lo);
        return r2 ? r2 : launch_scale_by_sigmoid(t.c2b, t.c2b_eff, t.c, t.mix, st);
      };
      for (int k = 0; k < 2; ++k) if ((rc = fold(v->tmid[k]))) return rc;
      for (int i = 0; i < 4; ++i) for (int r = 0; r < 3; ++r) if ((rc = fold(v->tup[i][r]))) return rc;
    }
    v->bias_folded = true;
  }
  half_t *b = v->buf[0], *c = v->buf[1], *d = v->buf[2];
  float *a = v->sbuf[0], *a2 = v->sbuf[1];   // the fp32 residual stream and its ping-pong partner
  int stage_no = 0, cur_c = top;
  // stage numbering: 0 conv_in | 1 mid.resnet0 | 2 mid.attention | 3 mid.resnet1 | then per up block: 3 resnets (+ upsampler)
  auto traced = [&](int& rc_out) -> bool {
    if (stage_no++ != stop_after) return false;
    const int64_t n = (int64_t)N * H * W * cur_c;
    rc_out = LATTE_OK;
    if (hipMemcpyAsync(trace_out, a, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, st) != hipSuccess)
      rc_out = fail(LATTE_ERR_HIP, "vae_trace: copy failed");
    *trace_numel = n;
    trace_dims[0] = N; trace_dims[1] = H; trace_dims[2] = W; trace_dims[3] = cur_c;
    return true;
  };
  if ((rc = launch_post_quant(z, v->pq_w, v->pq_b, v->pq_out, N, H * W, z_scale, st))) return rc;
  if ((rc = launch_conv_in(v->pq_out, v->ci_wt, v->ci_b, a, N, H, W, top, st))) return rc;
  if (traced(rc)) return rc;
  if ((rc = run_resnet(v, v->mid[0], &a, &a2, b, c, d, N, H, W, st, 0))) return rc;
  if (v->temporal && (rc = run_tresnet(v, v->tmid[0], a, c, d, N, H, W, st, 0))) return rc;
  if (traced(rc)) return rc;
  {  // mid-block attention: 1 head, dim 512, tokens = H*W per frame
    const int L = H * W;
    if (L % 128 != 0) return fail(LATTE_ERR_INVALID, "vae_decode: H*W must be a multiple of 128 for the attention GEMMs");
    if ((rc = launch_groupnorm(a, 1, c, v->agn_w, v->agn_b, v->gn_partial, v->gn_stats, N, L, top, 0, dt, st))) return rc;
    if ((rc = gemm_h16(c, v->aq_w, v->aq_b, b, nullptr, N * L, top, top, EPI_BIAS_H16, dt, st))) return rc;   // q  [N L, 512]
    if ((rc = gemm_h16(c, v->ak_w, v->ak_b, d, nullptr, N * L, top, top, EPI_BIAS_H16, dt, st))) return rc;   // k  [N L, 512]
    half_t* vt = b + (size_t)N * L * top;   // V0^T per frame [512, L], behind q in buffer b
    half_t* pm = d + (size_t)N * L * top;   // P per frame [L, L], behind k in buffer d
    half_t* o = c + (size_t)N * L * top;    // attention output [N L, 512], behind the normed input in buffer c
    const float scale = 1.0f / std::sqrt((float)top);
    for (int f = 0; f < N; ++f) {
      const half_t* hf = c + (size_t)f * L * top;
      if ((rc = gemm_h16(v->av_w, hf, v->zero_bias, vt, nullptr, top, L, top, EPI_BIAS_H16, dt, st))) return rc;          // V0^T = Wv h^T
      if ((rc = gemm_h16(b + (size_t)f * L * top, d + (size_t)f * L * top, v->zero_bias, v->scores, nullptr, L, L, top,
                         EPI_BIAS_F32, dt, st))) return rc;                                                               // S = q k^T
      if ((rc = launch_softmax_rows(v->scores, pm, L, L, scale, dt, st))) return rc;
      if ((rc = gemm_h16(pm, vt, v->zero_bias, o + (size_t)f * L * top, nullptr, L, top, L, EPI_BIAS_H16, dt, st))) return rc;  // P V0
    }
    {  // to_out + residual straight into the fp32 stream: stream += 1 * (o Wo^T + b_eff)
      GemmArgs g{};
      g.A = o; g.W = v->ao_w; g.bias = v->ao_b_eff; g.out = a; g.gate = v->ones; g.gate_stride = 0;
      g.M = N * L; g.N = top; g.K = top; g.rows_per_sample = N * L;
      if ((rc = launch_gemm(g, EPI_GATE_RES_F32, dt, 1, st))) return rc;
      kprof_mark(VC_ATTN, st);
    }
  }
  if (traced(rc)) return rc;
  if ((rc = run_resnet(v, v->mid[1], &a, &a2, b, c, d, N, H, W, st, 0))) return rc;
  if (v->temporal && (rc = run_tresnet(v, v->tmid[1], a, c, d, N, H, W, st, 0))) return rc;
  if (traced(rc)) return rc;
  for (int i = 0; i < 4; ++i) {
    for (int r = 0; r < 3; ++r) {
      if ((rc = run_resnet(v, v->up[i][r], &a, &a2, b, c, d, N, H, W, st, 1 + i))) return rc;
      if (v->temporal && (rc = run_tresnet(v, v->tup[i][r], a, c, d, N, H, W, st, 1 + i))) return rc;
      cur_c = v->up[i][r].cout;
      if (traced(rc)) return rc;
    }
    if (i < 3) {  // Upsample2D: nearest x2 folded into the conv's gather (on a half copy of the stream), fp32 result
      const int cch = v->ch[3 - i];
      const bool ups_lo = (vae_split_mask(v) >> (12 + i)) & 1;   // the half copy as hi + lo, a second pass on lo
      if (ups_lo) rc = launch_convert_f32_to_h16_split(a, d, c, (int64_t)N * H * W * cch, dt, st);
      else rc = launch_convert_f32_to_h16(a, d, (int64_t)N * H * W * cch, dt, st);
      if (rc) return rc;
      kprof_mark(VC_SMALL, st);
      if ((rc = launch_conv3x3(d, v->upc_w[i], v->upc_b[i], nullptr, nullptr, v->zeros, N, H, W, cch, cch, 1, dt, st, nullptr, a2))) return rc;
      if (ups_lo && (rc = launch_conv3x3(c, v->upc_w[i], v->zero_bias, nullptr, nullptr, v->zeros, N, H, W, cch, cch, 1, dt, st, a2, a2))) return rc;
      if (v->upc_w_lo[i] && ((vae_split_mask(v) >> (21 + i)) & 1) &&
          (rc = launch_conv3x3(d, v->upc_w_lo[i], v->zero_bias, nullptr, nullptr, v->zeros, N, H, W, cch, cch, 1, dt, st, a2, a2))) return rc;
      std::swap(a, a2);
      H *= 2;
      W *= 2;
      if (traced(rc)) return rc;
    }
  }
  if (stop_after >= 0) return fail(LATTE_ERR_INVALID, "vae_trace: stage index beyond the last traced stage");
  half_t* co_lo = ((vae_split_mask(v) >> 11) & 1) ? b : nullptr;   // conv_out reads the GroupNorm output as hi + lo (its weights are fp32)
  if ((rc = launch_groupnorm(a, 1, c, v->no_w, v->no_b, v->gn_partial, v->gn_stats, N, H * W, v->ch[0], 1, dt, st, 1e-6f, groupnorm_max_slabs(), co_lo))) return rc;
  if (!v->temporal) return launch_conv_out(c, v->co_w, v->co_b, out, N, H, W, v->ch[0], out_mode, dt, st, co_lo);
  // conv_out to fp32 NCHW frames, then time_conv_out over the frames of the chunk
  if ((rc = launch_conv_out(c, v->co_w, v->co_b, v->tbuf, N, H, W, v->ch[0], 0, dt, st, co_lo))) return rc;
  return launch_time_conv_out(v->tbuf, v->tco_w, v->tco_b, out, N, H * W, out_mode, st);
}

}  // extern "C"
