// Internal declarations shared by the engine's translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/latte_amd.h"

// Host-side schedule object behind latte_schedule_t (built in schedule.cpp).
struct latte_schedule {
  int num_timesteps = 0;
  std::vector<int64_t> timestep_map;
  std::vector<double> betas, alphas_cumprod, alphas_cumprod_prev, sqrt_recip_alphas_cumprod,
      sqrt_recipm1_alphas_cumprod, posterior_variance, posterior_log_variance_clipped, posterior_mean_coef1,
      posterior_mean_coef2, log_betas, sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod;
  // what the model predicts (gd ModelMeanType / ModelVarType as create_diffusion sets them, diffusion/__init__.py:32-45)
  int mean_type = 0;   // 0 EPSILON, 1 START_X (predict_xstart=True)
  int var_type = 0;    // 0 LEARNED_RANGE (learn_sigma=True), 1 FIXED_LARGE, 2 FIXED_SMALL (learn_sigma=False [, sigma_small])
  // fp32 copies of the tables on ONE device for the batched-timestep kernels (q_sample / training losses), built lazily:
  // [LATTE_NUM_DEV_TABLES][num_timesteps] = from_numpy(arr).float() of gaussian_diffusion.py:869-881
  mutable float* dev_tables = nullptr;
  mutable int dev_tables_device = -1;
};
enum DevTable : int { DT_SQRT_AC = 0, DT_SQRT_1MAC, DT_COEF1, DT_COEF2, DT_POST_LOGVAR, DT_LOG_BETAS, DT_SQRT_RECIP, DT_SQRT_RECIPM1,
                      DT_FIXED_LOGVAR, LATTE_NUM_DEV_TABLES };

// fp32 device copies of a schedule's tables (engine.cpp), [LATTE_NUM_DEV_TABLES][num_timesteps]
extern "C" int schedule_device_tables(const latte_schedule_t* s, const float** out, hipStream_t st);

namespace latte {

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
#define LATTE_HIP(call)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return ::latte::fail(LATTE_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE function attribute: the opt-in is remembered per device
// id (bit mask, thread-safe), so an engine created on a second GPU of the same process gets its own.
inline int ensure_dynamic_lds(const void* fn, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  LATTE_HIP(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    LATTE_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.fetch_or(bit, std::memory_order_release);
  }
  return LATTE_OK;
}

// 16-bit storage element (bf16 or f16 bit pattern depending on the engine's compute dtype)
typedef uint16_t half_t;

// ---- per-class kernel timing of the VAE decoder (latte_vae_profile_decode, vae_engine.cpp): when a profile is armed on this
// thread, every launcher records a HIP event on its stream after its launch; the time since the previous event goes to `cls`
enum VaeKernelClass : int { VC_CONV3 = 0, VC_GN_STATS, VC_GN_APPLY, VC_ATTN, VC_SMALL, VC_NUM_CLASSES, VC_START = -1 };
void kprof_mark(int cls, hipStream_t st);

// ---- GEMM  C[M,N] = A[M,K] · W[N,K]^T, fused epilogues ----------------------------------------
enum GemmEpi : int {
  EPI_BIAS_H16 = 0,       // out(half) = acc + bias
  EPI_BIAS_GELU_H16 = 1,  // out(half) = gelu_tanh(acc + bias)
  EPI_GATE_RES_F32 = 2,   // res(fp32)[m,n] += gate[sample(m)][n] * (acc + bias)
  EPI_BIAS_F32 = 3,       // out(fp32) = acc + bias
  EPI_ABLATE_NOSTORE = 4, // measurement only: bias add, nothing written (persistent kernel only)
  EPI_BIAS_RES_H16 = 5,   // out(half) = acc + bias + res(half)[m,n]   (plain kernel only; VAE attention out-proj)
  // measurement only, persistent kernel only, nothing written: main-loop ablations (results are garbage)
  EPI_ABLATE_NODMA = 6,     // no operand DMA inside the K loop
  EPI_ABLATE_NOLDSREAD = 7, // fragments read from LDS once, then reused
  EPI_ABLATE_NOMFMA = 8,    // DMA + fragment reads, no MFMA
  EPI_ABLATE_HOTSRC = 9,    // every K tile's DMA reads K tile 0 again (cache-hot source)
  EPI_ABLATE_DMA_A = 10,    // only the A operand is DMA'd in the loop
  EPI_ABLATE_DMA_B = 11,    // only the B operand is DMA'd in the loop
  EPI_ABLATE_TRACE = 12,    // full loop, no stores; workgroup 0 writes per-wave phase times (s_memtime ticks) to `out`:
                            // int64 [8 waves][8] = {L, barrier-1 wait, C issue, vmcnt wait, barrier-2 wait, DMA issue, total, K tiles}
  // training step (round 6; rolling 12-wave kernel only, gemm_pw.hip -- launch_gemm_pw): the GELU passes of the MLP inside the GEMMs
  EPI_BIAS_GELU_DUAL_H16 = 13,  // out(half) = u = acc + bias AND aux(half) = gelu_tanh(u) (of the ROUNDED u: what a separate GELU pass reads)
  EPI_DGELU_H16 = 14,           // out(half) = (acc + bias) * gelu_tanh'(aux(half)[m,n])     (the fc2 input-gradient GEMM: du = dh gelu'(u))
};
struct GemmArgs {
  const half_t* A;    // [Mpad, K]   (rows >= M may hold anything finite or not; never read back)
  const half_t* W;    // [N, K]
  const float* bias;  // [N]
  void* out;          // half [Mpad, N] | fp32 [Mpad, N]
  const float* gate;  // EPI_GATE_RES: gate base, row stride gate_stride (per sample)
  const half_t* res;  // EPI_BIAS_RES_H16: residual [Mpad, N] (may alias out)
  int M, N, K;        // M = valid rows; grid covers ceil(M / BM) tiles
  int gate_stride;
  int rows_per_sample;
  int stagger;        // persistent kernel: number of start cohorts (0/1 = none); cohort c sleeps c/stagger of a tile time
  int group_m;        // persistent kernel: tile rows walked together by the grouped tile order (0 = 8)
  int rmw_mode;       // measurement build only: look-ahead depth + 16 * non-temporal loads of the read-modify-write epilogue
  int tag;            // call site of a gated-residual GEMM (0 = attention out-projection, 1 = fc2): separate kernel symbols
  int k_chunk;        // plain kernels (variants 1-3) only: > 0 splits the contraction, grid.y = ceil(K / k_chunk) partial products
  long split_stride;  // ... written to (float*)out + blockIdx.y * split_stride (use EPI_BIAS_F32 with a zero bias)
  // Low-precision CORRECTION pass (round 6; rolling 12-wave kernel only, gemm_pw.hip): when A8 != nullptr the accumulators of a tile
  // also collect  A8 . W8^T  -- fp8 (OCP e4m3) operands on the block-scaled MFMA v_mfma_scale_f32_16x16x128_f8f6f4 with the two
  // constant E8M0 scales below -- behind the half-precision K loop and in front of the epilogue: the `lo` half of a split operand
  // pair (mfma_util.h: split2) at a quarter of the operand bytes and half the MFMA time of the [hi | lo] . [W | W] form.
  const uint8_t* A8;  // [Mpad, K] bytes: e4m3(lo * 2^LO8_A_SHIFT)
  const uint8_t* W8;  // [N, K]    bytes: e4m3(W * 2^LO8_W_SHIFT)
  // FP4 correction pass (round 6, second form; rolling kernel, LO = 2): A4 / W4 hold OCP e2m1 codes, two per byte (element k in bits
  // 4 (k & 1) of byte k >> 1), rows of lo4_pitch(K) bytes (K rounded up to 256, zero padded), with ONE E8M0 scale byte per ROW
  // (A4s [Mpad], W4s [N]): value = code * 2^(scale - 127).  v_mfma_scale_f32_16x16x128_f8f6f4 with cbsz = blgp = 4 runs at twice the
  // fp8 rate and a K tile of 128 row bytes is K = 256 (mfma_util.h: quant4; tools/mx_probe.hip layout H0).
  const uint8_t* A4;
  const uint8_t* W4;
  const uint8_t* A4s;
  const uint8_t* W4s;
  half_t* aux;        // EPI_BIAS_GELU_DUAL_H16: second output [Mpad, N]; EPI_DGELU_H16: the pre-activation u [Mpad, N] (read only)
};
// constant block scales of the correction pass: A8 holds the rounding remainder of a half operand (|lo| <= 2^-11 |x|), W8 weights of
// |w| < 7; the MFMA multiplies the products back by 2^-(LO8_A_SHIFT + LO8_W_SHIFT) (E8M0 scale bytes 127 - shift)
constexpr int LO8_A_SHIFT = 12, LO8_W_SHIFT = 6;
inline int lo4_pitch(int K) { return (K + 255) / 256 * 128; }   // bytes per row of an fp4 correction operand
bool gemm_lo8_ok(int M, int N, int K);
bool gemm_lo4_ok(int M, int N, int K);   // the shape takes the fp4 correction pass (whole 192-wide tile columns, K % 64 == 0, 32-bit offsets)   // the shape takes the correction pass (whole 192-wide tile columns, K % 128 == 0, 32-bit offsets)
// variant: 0 = pick for the shape; simple double-buffered kernel: 1 = 128x128 tile, 2 = 256x128, 3 = 256x256;
// ping-pong kernel: 4 = 256x128, 5 = 256x192, 6 = 256x256; persistent ping-pong kernel: 7 = 256x128,
// 8 = 256x192, 9 = 256x256   (N % tileN == 0 required)
int launch_gemm(const GemmArgs& a, int epi, int dtype, int variant, hipStream_t st);
// variants 10 / 11 = producer-wave kernels, 256x192 tile, 12 waves (gemm_pw.hip): two-segment / rolling schedule;
// variants 12 / 13 = small-M kernel, 128x144 tile, four LDS stages: 12 waves / 12 MFMA + 4 DMA waves (gemm.hip)
int launch_gemm_pw(const GemmArgs& a, int epi, int dtype, int roll, hipStream_t st);
int gemm_tile_m(int variant);
int gemm_tile_n(int variant);
int gemm_auto_variant(int M, int N, int epi);
int gemm_resolve_variant(int M, int N, int K, int epi);   // the variant launch_gemm runs for variant == 0
bool gemm_small_tile_ok(int M, int N, int K);   // gated GEMM: the 128 x 144 tile (variant 13) takes it when no variant is forced

// ---- attention --------------------------------------------------------------------------------
struct AttnArgs {
  const half_t* qkv;  // [rows, 3*D]; columns ordered [3][heads][hd]   (latte.py:50)
  half_t* out;        // [rows, D];   column = head*hd + d              (latte.py:70)
  int num_seq;        // sequences
  int L;              // tokens per sequence
  int heads, hd, D;
  int U;              // sequences per sample
  int64_t sample_stride;  // rows per sample (F*T)
  int64_t seq_stride;     // row offset between consecutive sequences of a sample (T spatial, 1 temporal)
  int64_t row_stride;     // row offset between consecutive tokens of a sequence (1 spatial, T temporal)
  float scale;            // hd^-0.5
  int variant;            // 0 = pick by L; 1 = force the generic flash kernel for L > 16 (test hook)
  // cross-attention (LatteT2V attn2, launch_cross_attention): queries from `qkv` viewed as [rows, q_ld] (column head*hd),
  // keys / values from kv [num_samples * Lk, 2*D] ([K | V], column head*hd), shared by the U sequences of a sample;
  // kbias: additive score bias [num_samples, Lk] (the -10000 * (1 - mask) of latte_t2v.py:746-749) or NULL
  const half_t* kv;
  const float* kbias;
  int Lk, q_ld;
  unsigned char* out8;    // f16 only, may be nullptr: [rows, D] bytes receive the fp8 remainder of the output (mfma_util.h: split8_f16)
};
int launch_attention(const AttnArgs& a, int dtype, hipStream_t st);
int launch_cross_attention(const AttnArgs& a, int dtype, hipStream_t st);

// ---- fused QKV projection + attention (qkv_attn.hip): out[rows, D] = attention(xn W_qkv^T + b) per (sequence, head), Q / K / V
// of a head live only in LDS.  mode 0: spatial sequences (T == 256 tokens of one frame), mode 1: temporal sequences (F == 16
// frames of one token); rows in the canonical [B, F, T] order.
struct QkvAttnArgs {
  const half_t* xn;    // [B F T, D]  LN-modulated activations (must not alias out)
  const half_t* w;     // [3 D, D]    qkv weight, rows ordered [3][heads][hd]   (latte.py:50)
  const float* bias;   // [3 D]
  half_t* out;         // [B F T, D]  column = head * hd + d                    (latte.py:70)
  half_t* dbg_qkv;     // test hook: when set, [B F T, 3 D] receives the half q | k | v the kernel holds in LDS
  long long* dbg_trace;  // measurement hook: int64 [8 waves][4] = shader-clock ticks of workgroup 0 in {projection loop, image
                         // write, attention} summed over its units, and the unit count
  int B, F, T, D, heads, hd;
  int mode;
  float scale;         // hd^-0.5
  int flags;           // schedule variants, same results (qkv_attn.hip): bit 0 = next unit's first operand tile fetched under the
                       // attention phase, bit 1 = attention-phase issue priority for group 0, bit 2 = four heads per XCD (16 heads)
  int out_split;       // 1: out is [B F T, 2 D] = [hi | lo] -- the attention output as a split operand pair (mfma_util.h: split2)
                       // 2 (f16): out stays [B F T, D], out8 receives the fp8 remainder (mfma_util.h: split8_f16; GemmArgs::A8)
  unsigned char* out8; // [B F T, D] bytes
};
bool qkv_attention_fusable(int D, int heads, int hd, int F, int T, int mode, int64_t rows);
int launch_qkv_attention(const QkvAttnArgs& a, int dtype, hipStream_t st);

// ---- pointwise / small kernels ------------------------------------------------------------------
// y(half)[m, :] = LN(x[m, :]) * (1 + scale[s(m), :]) + shift[s(m), :], eps 1e-6, no affine.
// If temp_embed != nullptr: x[m,:] += temp_embed[frame(m), :] first and is written back (latte.py:357-358).
// LayerNorm-modulate with the f16 + FP4-remainder output (y4: [M, lo4_pitch(D)] e2m1 codes, y4s: [M] E8M0 row scales; GemmArgs::A4 / A4s)
int launch_ln_modulate_split4(const float* x_in, half_t* y, unsigned char* y4, unsigned char* y4s, const float* shift, const float* scale,
                              int mod_stride, int M, int D, int rows_per_sample, int dtype, hipStream_t st);
int launch_ln_modulate(const float* x_in, float* x_rw, half_t* y, const float* shift, const float* scale,
                       int mod_stride, int M, int D, int rows_per_sample, const float* temp_embed, int T,
                       int F, int dtype, hipStream_t st, int split = 0,   // split 1: y is [M, 2 D] = [hi | lo] (mfma_util.h: split2)
                       unsigned char* y8 = nullptr);   // split 2 (f16): y [M, D] + y8 [M, D] bytes = fp8 remainder (GemmArgs::A8)
int launch_pack_w8(const half_t* in, unsigned char* out, int64_t n, int dtype, hipStream_t st);   // GemmArgs::W8 of an f16 weight
enum SmallIn : int { IN_PLAIN = 0, IN_SILU = 1, IN_TFREQ = 2 };
// out[b, n] = bias[n] + sum_k in(b,k) * W[n, k]  (+ add_table[add_idx[b], n]); fp32 exact.
int launch_small_linear(int in_mode, const float* in, const int64_t* t, const float* W, const float* bias,
                        const float* add_table, const int64_t* add_idx, float* out, int B, int N, int K,
                        int out_stride, hipStream_t st);
int launch_patch_embed(const float* x, const float* Wt /* [C*p*p][D] */, const float* bias, const float* pos,
                       float* out, int BF, int C, int H, int p, int D, hipStream_t st);
int launch_final_layer(const float* x, const float* shift, const float* scale, int mod_stride,
                       const float* Wt /* [D][P] */, const float* bias, float* out, int M, int D,
                       int rows_per_sample, int T, int p, int Cout, int H, hipStream_t st);
int launch_cond_rows(const float* temb, const float* ytab, const int64_t* y, float* out, int n_steps, int bu, int D,
                     hipStream_t st);
// text_embedding_projection of the extras == 78 variant (latte.py:238-242): out[B,N] = Linear(SiLU(text[B,K]))
int launch_text_proj(const float* text, const float* W, const float* bias, float* out, int B, int N, int K, hipStream_t st);
int launch_iota(int64_t* p, int n, hipStream_t st);
// x[M, N] += gate[m / rows_per_sample, :] * (sum of `splits` fp32 partial products (slab stride `stride`) + bias): the reduction
// of a split-K gated GEMM (engine.cpp: gated_gemm)
int launch_gated_split_reduce(float* x, const float* ws, int splits, size_t stride, const float* bias, const float* gate,
                              int gate_stride, int rows_per_sample, int M, int N, hipStream_t st);
int launch_silu_rows(const float* in, float* out, size_t n, hipStream_t st);   // out = SiLU(in), may alias
// adaLN-single (latte_t2v.py:301-304,913-915): mod[b, j, :] = table[j, :] + t6[b, (j % 6) * D ...] for the 6 * nblk block
// rows, then the 2 head rows = head_table[r, :] + temb[b, :]; mod is [B, (6 * nblk + 2) * D].
int launch_adaln_single(const float* tables /* [nblk, 6, D] */, const float* head_table /* [2, D] */, const float* t6 /* [B, 6D] */,
                        const float* temb /* [B, D] */, float* mod, int B, int nblk, int D, hipStream_t st);
// out[b, f, c, :] = in[b, c, f, :] (to_bfc = 1) or out[b, c, f, :] = in[b, f, c, :] (to_bfc = 0); hw contiguous floats
int launch_permute_cf(const float* in, float* out, int B, int C, int F, int hw, int to_bfc, hipStream_t st);
int launch_fill_f32(float* p, float v, size_t n, hipStream_t st);
// Kernel-choice overrides of the A/B tests (latte_debug_set_choice, include/latte_amd_debug.h; engine.cpp).  Process-global, 0 = the
// library's own choice.  Every value selects another implementation of the SAME function; nothing here can change a result beyond
// rounding.  (Round 3 read environment variables at every launch instead, among them ablations with garbage results.)
enum DebugChoice { DBG_ATTN_VARIANT = 0, DBG_XATTN_FLASH, DBG_TN_KERNEL, DBG_TN_WN, DBG_ATTN_BWD_TILES, DBG_CONV_KERNEL, DBG_VAE_SPLIT, DBG_NUM_CHOICES };
int debug_choice(DebugChoice c);
int set_debug_choice(const char* name, int value);   // 0 = ok, -1 = unknown name / value not offered by this build

int launch_scale_f32(float* p, float s, size_t n, hipStream_t st);   // p[i] *= s
int launch_scale_f32_dev(float* p, const float* s_dev, int inverse, size_t n, hipStream_t st);   // p[i] *= *s_dev (or /=), factor in device memory
// One guided DDIM step of the text-to-video loop (pipeline_latte.py:747-758 + DDIMScheduler.step, eta = 0) on x [b, C, F, HW]
// in place: model_out is the denoiser output of the guidance pair in FRAME layout [(2b) F, Cout, HW] ([negative | prompt]):
// eps = u + s (c - u) on the first C channels (learned sigma dropped), x0 = (x - c1 eps) / c2, x' = c3 x0 + c4 eps.
int launch_t2v_guided_ddim(float* x, const float* model_out, int b, int C, int Cout, int F, int hw, float scale, float c1, float c2,
                           float c3, float c4, hipStream_t st);
int launch_mask_bias(const float* mask, float* bias, size_t n, hipStream_t st);   // bias = (1 - mask) * -10000
int launch_cfg_combine(float* out, int half_batch, int F, int Cout, int HW, float cfg_scale, hipStream_t st);
int launch_convert_f32_to_h16(const float* in, half_t* out, int64_t n, int dtype, hipStream_t st);
// W4 image of a [N, K] half weight: e2m1 codes [N, lo4_pitch(K)] + one E8M0 scale byte per row (GemmArgs::W4 / W4s)
int launch_pack_w4(const half_t* in, unsigned char* out4, unsigned char* out_scale, int N, int K, int dtype, hipStream_t st);
int launch_convert_f32_to_h16_split(const float* in, half_t* out, half_t* out_lo, int64_t n, int dtype, hipStream_t st);   // + the f16 rounding residual
int launch_convert_h16_to_f32(const half_t* in, float* out, int64_t n, int dtype, hipStream_t st);
int launch_transpose_f32(const float* in, float* out, int rows, int cols, hipStream_t st);
// Philox4x32-10 + Box-Muller standard normals; element i depends only on (seed, offset + i).
int launch_fill_normal(float* out, size_t n, uint64_t seed, uint64_t offset, hipStream_t st);

// ---- VAE decoder kernels (vae.hip) -----------------------------------------------------------------
// out32 != nullptr: fp32 output  out32 = conv + bias (+ res32)  (the decoder's residual stream is fp32); else half `out` (+ res)
int launch_conv3x3(const half_t* in, const half_t* w, const float* bias, const half_t* res, half_t* out,
                   const half_t* zeros, int N, int Hin, int Win, int Cin, int Cout, int ups, int dtype, hipStream_t st,
                   const float* res32 = nullptr, float* out32 = nullptr, int taps3 = 0);
// AutoencoderKLTemporalDecoder pieces: Conv3d (3,1,1) weight pack (optionally scaled by sigmoid(*mix)), bias scale, time_conv_out
int launch_pack_conv_t(const float* w, half_t* out, int Cout, int Cin, const float* mix, int dtype, hipStream_t st, half_t* out_lo = nullptr);
int launch_scale_by_sigmoid(const float* in, float* out, int n, const float* mix, hipStream_t st);
int launch_time_conv_out(const float* in, const float* w, const float* bias, void* out, int T, int HW, int out_mode, hipStream_t st);
// x: half [N, HW, C] or (x_is_f32) fp32; y: half
int launch_groupnorm(const void* x, int x_is_f32, half_t* y, const float* gamma, const float* beta, float* partial, float* stats,
                     int N, int HW, int C, int silu, int dtype, hipStream_t st, float eps = 1e-6f, int max_slabs = 256, half_t* y_lo = nullptr);   // max_slabs <= groupnorm_max_slabs() (x T frames for one T-frame sample)
int groupnorm_max_slabs();
int launch_post_quant(const float* z, const float* w, const float* b, float* out, int N, int hw, float z_scale, hipStream_t st);
int launch_conv_in(const float* x, const float* wt, const float* bias, float* out, int N, int H, int W, int Cout, hipStream_t st);
int launch_conv_out(const half_t* x, const float* wt, const float* bias, void* out, int N, int H, int W, int C, int out_mode,
                    int dtype, hipStream_t st, const half_t* x_lo = nullptr);   // x_lo: the f16 rounding residual of x (split input), or nullptr
int launch_softmax_rows(const float* s, half_t* p, int rows, int L, float scale, int dtype, hipStream_t st);
int launch_pack_conv_w(const float* w, half_t* out, int Cout, int Cin, int dtype, hipStream_t st, half_t* out_lo = nullptr);   // out_lo: the f16 rounding residual
int launch_pack_small_w(const float* w, float* out, int Cout, int Cin, int transpose, hipStream_t st);

// ---- training-step kernels (train.hip, train_attn.hip) --------------------------------------------------------------
int train_rows_per_run(int rps);
int launch_gated_add(const float* x_in, const half_t* y, const float* gate, int gate_stride, float* x_out, int M, int D, int rps,
                     int dtype, hipStream_t st);
int launch_gated_add_ln(const float* x_in, const half_t* y, const float* gate, int gate_stride, float* x_out, half_t* xn, const float* shift,
                        const float* scale, int mod_stride, int M, int D, int rps, const float* te, int T, int F, int dtype, hipStream_t st);
// dgate == nullptr: no finalize launch (the stage's finalize kernel reads the partial rows);  bias_partial != 0: partial is
// [M / (4 R)][2][D] and row 1 of a run holds sum_rows gate * dx (the bias gradient of the branch's output linear)
int launch_gate_bwd(const float* dx, const half_t* y, const float* gate, int gate_stride, half_t* dy, float* partial, float* dgate,
                    int out_stride, int M, int D, int rps, int dtype, hipStream_t st, int bias_partial = 0);
int launch_ln_bwd(const half_t* dy, const float* x, const float* scale, int mod_stride, const float* dx_in, float* dx_out, float* partial,
                  float* dshift, float* dscale, int out_stride, int M, int D, int rps, int dtype, hipStream_t st,
                  const half_t* y2 = nullptr, const float* gate2 = nullptr, int gate2_stride = 0, half_t* dy2 = nullptr,
                  float* gpartial = nullptr);   // y2: + the gated residual's backward of the branch below on the same pass
int launch_gelu_fwd(const half_t* u, half_t* h, size_t n, int dtype, hipStream_t st);
int launch_gelu_bwd(const half_t* u, const half_t* dh, half_t* du, size_t n, int dtype, hipStream_t st);
int colsum_chunks(int M);
int launch_colsum_half(const half_t* in, int M, int C, float* partial, float* out, int accumulate, int dtype, hipStream_t st);
// inv_scale_dev: optional device float; the sum is multiplied by 1 / *inv_scale_dev (the loss scale, a power of two) on the way out
int launch_split_reduce(const float* partial, int splits, size_t stride, size_t n, float* out, int accumulate, hipStream_t st,
                        const float* inv_scale_dev = nullptr);
int launch_pack_weight(const float* w, half_t* wn, half_t* wt, int N, int K, int dtype, hipStream_t st);
// ---- small-kernel consolidation of the training step (train_fin.hip)
struct StageFinArgs {
  const float* mod_src[6];   // per modulation chunk: row-run partials [B rows_per_sample][nsum][D]
  int mod_nsum[6], mod_which[6];
  int n_mod, rows_per_sample, B, D;
  float* dmod;               // [B][dmod_stride], this linear's first column (assigned, loss-scaled domain)
  int dmod_stride;
  const float* csilu;        // [B][D]
  float *dW, *db;            // adaLN linear gradients [n_mod D][D], [n_mod D]
  int n_bias;                // column sums: out[col] = sum_r src[r stride + col]
  const float* bias_src[4];
  int bias_rows[4], bias_stride[4], bias_cols[4];
  float* bias_out[4];
  int bias_blk[5];           // (filled by the launcher)
  const float* scaler;       // device loss scale or nullptr: dW, db and the bias sums leave the scaled domain
};
int launch_stage_finalize(const StageFinArgs& a, hipStream_t st);
int adaln_dc_splits(int nmod);
int launch_adaln_dc(const float* dmod, int nmod, int B, const float* w_blocks, long blk_stride, int depth, int rows6, const float* w_final,
                    int D, float* ws, float* dc, hipStream_t st);
int narrow_blocks(int M);
int launch_narrow_outer(const float* nar, int P, const void* wide, int wide_half, int D, int M, float* dW, long so_p, long so_k,
                        float* nsum_out, float* wsum_out, float* ws, int dtype, const float* inv_scale_dev, hipStream_t st);
int launch_narrow_dx(const float* nar, int P, const float* W, int D, int M, half_t* out, int dtype, hipStream_t st);
struct PackDesc { const float* w; half_t* wn; half_t* wt; int N, K; };
struct PackPlan { int tiles_per_block; int tile0[4]; };
int launch_pack_weights(const PackDesc* descs_dev, int blocks, const PackPlan& pl, int dtype, hipStream_t st);
int launch_naive_gemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long scm, long scn, int M, int N,
                      int K, float alpha, int accumulate, hipStream_t st, int splits = 1, float* ws = nullptr);
int launch_tfreq(const int64_t* t, float* out, int B, hipStream_t st);
int launch_gather_i64(const int64_t* table, const int64_t* idx, int64_t* out, int n, hipStream_t st);
int launch_unpatchify_bwd(const float* dout, float* dtok, int BF, int G, int p, int Cout, hipStream_t st);
int launch_im2col_patch(const float* x, float* pix, int BF, int G, int p, int C, hipStream_t st);
int launch_embedding_bwd(const float* dc, const int64_t* idx, float* dtable, int B, int D, hipStream_t st);
int launch_silu_bwd(const float* dout, const float* pre, float* din, size_t n, int accumulate, hipStream_t st);
int launch_add_rows(float* dst, const float* src, size_t n, hipStream_t st);
int launch_rows_sum(const float* in, int B, long stride, int N, float* out, int accumulate, hipStream_t st);
int launch_loss_grad(const float* tables, int n_steps, int mean_type, int var_type, const float* x_start, const float* x_t,
                     const float* noise, const float* model_out, const int64_t* t, int batch, int frames, int channels, int hw,
                     float vb_scale, float* dmodel_out, hipStream_t st);
int sumsq_blocks();
int launch_grad_norm(const float* g, size_t n, double* partial, float max_norm, int clip, float* stats, float* scaler, hipStream_t st);
int launch_adamw_ema(float* p, float* g, float* m, float* v, float* ema, size_t n, float lr, float b1, float b2, float eps, float wd,
                     int step, float ema_decay, const float* stats, const float* step_dev, hipStream_t st);
// dW[N, K] = dY[M, N]^T X[M, K] without transposed copies (gemm_tn.hip); partial: float [ceil(M / m_chunk)][N][K]
int gemm_tn_tile_n();   // rows of dW per workgroup tile (for the caller's split heuristic)
int gemm_tn_plan(int M, int N, int K, int* chunk);   // -> splits of the contraction, *chunk = rows per split
int launch_gemm_tn(const half_t* dY, const half_t* X, float* partial, int M, int N, int K, int m_chunk, int dtype, hipStream_t st,
                   float* colsum_partial = nullptr);   // colsum_partial: [splits][N] column sums of dY per split (8-wave kernel only)
bool gemm_tn8_ok(int M, int N, int K);
int launch_attention_bwd(const half_t* qkv, const half_t* o, const half_t* dout, half_t* dqkv, float* stats, int num_seq, int L, int heads,
                         int hd, int U, int64_t sample_stride, int64_t seq_stride, int64_t row_stride, int dtype, hipStream_t st);

struct SamplerCoefs {  // fp32 values of the fp64 tables at the step (gaussian_diffusion.py:869-881)
  float min_log, max_log, sqrt_recip, sqrt_recipm1, coef1, coef2;
  float sqrt_ab_prev, dir_coef, sigma;  // ddim: sqrt(ab_prev), sqrt(1-ab_prev-sigma^2), sigma
  float nonzero;                        // 0 when index == 0
  float cfg_scale;                      // > 1: model_out is a raw doubled batch, combine here
  float sqrt_one_minus_ab;              // condition_score: (1 - alpha_bar).sqrt() in fp32 (gd:368)
  float fixed_log_var;                  // var_type 1 / 2: the step's log variance (gd:298-313)
  int mean_type, var_type;              // latte_schedule::mean_type / var_type; var_type != 0: model_out has C channels
  int method, clip;
};
// x0_in: pred_xstart after the caller's denoised_fn (replaces the computed one BEFORE the clamp); grad: cond_fn(x, t);
// predict_only: write the raw (unclamped) x_start prediction to x0_out and nothing else.
int launch_sampler_update(const SamplerCoefs& c, const float* x, const float* model_out, const float* noise,
                          int batch, int frames, int channels, int hw, int raw_cfg, float* sample_out,
                          float* x0_out, hipStream_t st, const float* x0_in = nullptr, const float* grad = nullptr,
                          int predict_only = 0);

// ---- batched-timestep kernels of the training path (gaussian_diffusion.py:216-229 q_sample, :686-795 training_losses)
int launch_q_sample(const float* tables, int n_steps, const float* x_start, const float* noise, const int64_t* t, int batch,
                    size_t per_sample, float* x_t, hipStream_t st);
// per-sample mse = mean((target - pred)^2), vb = (t == 0 ? decoder NLL : KL) in bits; partial: [batch][blocks][2] scratch
int launch_training_terms(const float* tables, int n_steps, int mean_type, int var_type, const float* x_start, const float* x_t,
                          const float* noise, const float* model_out, const int64_t* t, int batch, int frames, int channels, int hw,
                          float* partial, int blocks_per_sample, float* mse, float* vb, hipStream_t st);
int training_terms_blocks(size_t per_sample);
// loss = kl_only ? scale * vb : mse (+ scale * vb when has_vb); mse_out / vb_out may be nullptr
int launch_training_combine(const float* mse, const float* vb, int has_vb, int kl_only, float vb_scale, int batch, float* mse_out,
                            float* vb_out, float* loss_out, hipStream_t st);

}  // namespace latte
