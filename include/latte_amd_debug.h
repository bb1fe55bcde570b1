/*
 * latte_amd_debug.h — per-kernel test hooks of liblatte_amd.so (C-ABI, same conventions as
 * latte_amd.h).  They exist so tests/ can check every HIP kernel against the PyTorch op sequence it
 * replaces (the reference has no tests at all).  All pointers are device pointers;
 * "half" buffers hold bf16 or f16 bit patterns according to `dtype` (LATTE_DTYPE_*).
 */
#ifndef LATTE_AMD_DEBUG_H_
#define LATTE_AMD_DEBUG_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* C[M,N] = A[M,K] W[N,K]^T with epilogue epi: 0 half=acc+bias | 1 half=gelu_tanh(acc+bias) |
 * 2 out_f32[m,n] += gate[(m / rows_per_sample) * gate_stride + n] * (acc+bias) | 3 out_f32 = acc+bias.
 * A must be allocated with rows padded to a multiple of 256.  (nn.Linear, latte.py:43,45,171)
 * variant: tile configuration as in csrc/common.h (0 = the engine's choice) + 1000 * call-site tag of a gated-residual
 * GEMM (0 attention out-projection, 1 fc2: separate kernel symbols for the profiler). */
int latte_debug_gemm(const void* A, const void* W, const float* bias, void* out, const float* gate, int M, int N,
                     int K, int gate_stride, int rows_per_sample, int epi, int dtype, int variant, void* stream);
/* The training step's GELU epilogues of the rolling 12-wave GEMM (csrc/gemm_pw.hip; needs N % 192 == 0, K % 64 == 0, K >= 128):
 * epi 13: out = u = A W^T + bias (half) and aux = gelu_tanh(u) of the rounded u (latte.py:170 through fc1, one launch);
 * epi 14: out = (A W^T + bias) * gelu_tanh'(aux) with aux = u read only (the fc2 input-gradient GEMM of train.py's backward). */
int latte_debug_gemm_gelu(const void* A, const void* W, const float* bias, void* out, void* aux, int M, int N, int K, int epi, int dtype,
                          void* stream);
/* The same product with the FP8 CORRECTION PASS of a split operand (round 6; engine option guided_split bits 2 / 3): the tile's
 * accumulators also collect  dec(A8) 2^-12 . (dec(W8) 2^-6)^T  -- A8 [Mpad, K] / W8 [N, K] bytes of OCP e4m3 codes -- on the block-
 * scaled MFMA v_mfma_scale_f32_16x16x128_f8f6f4 behind the half-precision K loop (csrc/gemm_pw.hip, rolling 12-wave kernel; f16,
 * N % 192 == 0, K % 128 == 0; epi 1 = bias + GELU -> half, epi 2 = gated fp32 read-modify-write). */
int latte_debug_gemm_lo8(const void* A, const void* W, const void* A8, const void* W8, const float* bias, void* out, const float* gate,
                         int M, int N, int K, int gate_stride, int rows_per_sample, int epi, int dtype, void* stream);
/* W8 of an f16 weight: out8[i] = e4m3(clamp(w[i] * 2^6, +-448)) (n % 4 == 0). */
int latte_debug_pack_w8(const void* w, void* out8, int64_t n, int dtype, void* stream);
/* FP4 form of the correction pass (round 6): OCP e2m1 codes, two per byte (element k in bits 4 (k & 1) of byte k >> 1), rows of
 * ((K + 255) / 256) * 128 bytes (zero padded) with ONE E8M0 scale byte per row (value = code * 2^(scale - 127)); the block-scaled MFMA runs
 * them at twice the fp8 rate.  latte_debug_pack_w4: [N, K] f16 weights -> codes + row scales; latte_debug_ln_modulate_split4: LayerNorm-
 * modulate with y [M, D] = the nearest f16 (the plain call's bits) + y4 / y4s = codes and row scales of the remainder; latte_debug_gemm_lo4:
 * the GEMM of latte_debug_gemm (epi 1 or 2) with  (A4 2^A4s) . (W4 2^W4s)^T  collected behind its f16 K loop. */
int latte_debug_pack_w4(const void* w, void* out4, void* out_scale, int N, int K, int dtype, void* stream);
int latte_debug_ln_modulate_split4(const float* x, void* y, void* y4, void* y4s, const float* shift, const float* scale, int mod_stride, int M,
                                   int D, int rows_per_sample, int dtype, void* stream);
int latte_debug_gemm_lo4(const void* A, const void* W, const void* A4, const void* A4s, const void* W4, const void* W4s, const float* bias,
                         void* out, const float* gate, int M, int N, int K, int gate_stride, int rows_per_sample, int epi, int dtype,
                         void* stream);
/* latte_debug_ln_modulate with the split output of the fp8 form: y [M, D] = the nearest f16 (bit for bit the plain kernel's output),
 * y8 [M, D] bytes = e4m3(clamp((value - y) * 2^12, +-448)). */
int latte_debug_ln_modulate_split8(const float* x, void* y, void* y8, const float* shift, const float* scale, int mod_stride, int M, int D,
                                   int rows_per_sample, int dtype, void* stream);
/* latte_debug_qkv_attention with the same split of the attention output: out [B F T, D] f16 + out8 [B F T, D] bytes. */
int latte_debug_qkv_attention_split8(const void* xn, const void* w, const float* bias, void* out, void* out8, int B, int F, int T, int D,
                                     int heads, int mode, int dtype, void* stream);
/* Attention core of latte.py:50-70 on a [rows, 3*D] qkv buffer (see AttnArgs in csrc/common.h). */
/* Host logic only (no GPU needed): the tile / kernel variant launch_gemm picks for C[M, N] = A[M, K] W[N, K]^T with epilogue
 * `epi` when none is forced (csrc/gemm.hip: gemm_resolve_variant; variants as for the "gemm_variant" engine option). */
int latte_debug_gemm_choice(int M, int N, int K, int epi);
/* Host logic only: 1 when the fused QKV projection + attention kernel takes the shape (csrc/qkv_attn.hip: head_dim 64 | 72,
 * spatial mode 0: 256 tokens per frame; temporal mode 1: 16 frames and a token count that is a multiple of 16; operands inside the
 * 4 GiB buffer-offset range), else the engine runs the separate qkv GEMM + attention kernels. */
int latte_debug_qkv_attention_fusable(int D, int heads, int F, int T, int mode, int64_t rows);
/* Host logic only: how the weight-gradient GEMM dW[N, K] = dY[M, N]^T X[M, K] splits its contraction (csrc/gemm_tn.hip):
 * returns the number of partial products, *rows_per_split (a multiple of 64) rows of M each. */
int latte_debug_gemm_tn_plan(int M, int N, int K, int* rows_per_split);
int latte_debug_attention(const void* qkv, void* out, int num_seq, int L, int heads, int hd, int U,
                          int64_t sample_stride, int64_t seq_stride, int64_t row_stride, int dtype, void* stream);
/* latte_debug_attention with the f16 + fp8-remainder output of guided calls: out [rows, D] (bit for bit the plain call's output) and
 * out8 [rows, D] bytes = e4m3(clamp((value - out) * 2^12, +-448)); f16 only, every kernel of the un-fused path (L <= 16, generic flash,
 * 128 < L <= 256, L > 256). */
int latte_debug_attention_split8(const void* qkv, void* out, void* out8, int num_seq, int L, int heads, int hd, int U, int64_t sample_stride,
                                 int64_t seq_stride, int64_t row_stride, int dtype, void* stream);
/* Fused QKV projection + attention core (csrc/qkv_attn.hip; latte.py:48-70 up to, not including, the output projection):
 * out[B F T, D] = attention(xn[B F T, D] W[3D, D]^T + bias[3D]) per (sequence, head), rows in the canonical [B, F, T] order.
 * mode 0: spatial sequences (needs T == 256), mode 1: temporal sequences (needs F == 16, T % 16 == 0); head_dim D / heads
 * must be 64 or 72.  dbg_qkv (may be NULL): half [B F T, 3D] receives the q | k | v values the kernel holds in LDS (what the
 * un-fused qkv GEMM would have written).  xn must not alias out.  flags = QkvAttnArgs::flags, schedule variants with identical
 * results: bit 0 = the next unit's first operand tile is fetched under the attention phase, bit 1 = attention-phase issue priority
 * for wave group 0 (spatial mode), bit 2 = four heads per XCD instead of all heads of every eighth sequence group (16 heads);
 * bit 8 = QkvAttnArgs::out_split: out is [B F T, 2 D] = [hi | lo], the attention output as a split operand pair (the half nearest to
 * each value and the half nearest to the remainder) -- what guided calls feed the out-projection (engine option guided_split). */
int latte_debug_qkv_attention(const void* xn, const void* w, const float* bias, void* out, void* dbg_qkv, int B, int F, int T,
                              int D, int heads, int mode, int flags, int dtype, void* stream);
/* The same launch with a phase trace (measurement): trace = int64 [8 waves][4] receives workgroup 0's shader-clock ticks in
 * {QKV projection loop, LDS image write, attention phase} summed over its units, and its unit count. */
int latte_debug_qkv_attention_trace(const void* xn, const void* w, const float* bias, void* out, void* dbg_qkv, long long* trace,
                                    int B, int F, int T, int D, int heads, int mode, int flags, int dtype, void* stream);
/* Backward of the attention core (autograd of latte.py:61-70 on the [rows, 3 D] qkv layout): dqkv [rows, 3 D] from qkv, the
 * forward output o [rows, D] and its gradient dout [rows, D]; stats: float scratch [num_seq * heads * L * 3].  Sequences are
 * addressed as in latte_debug_attention.  L <= 16 runs one wave per (sequence, head), larger L the two tile passes
 * (latte_debug_set_choice("attn_bwd_tiles", 1) forces the tile passes: test hook). */
int latte_debug_attention_bwd(const void* qkv, const void* o, const void* dout, void* dqkv, float* stats, int num_seq, int L,
                              int heads, int hd, int U, int64_t sample_stride, int64_t seq_stride, int64_t row_stride, int dtype,
                              void* stream);
/* Weight-gradient product of the training step: dW[N, K] = sum_m dY[m, n] X[m, k] (autograd of nn.Linear, latte.py:43-45) on the
 * transposed-operand GEMM of csrc/gemm_tn.hip + its fixed-order split reduction; dY half [M, N], X half [M, K], both row-major.
 * workspace: >= splits * N * K floats (256 * 256 * ceil(N/256) * ceil(K/256) * 256 is always enough). */
int latte_debug_gemm_tn(const void* dY, const void* X, float* dW, float* workspace, int64_t workspace_floats, int M, int N, int K,
                        int dtype, void* stream);
/* The same product with colsum[n] = sum_m dY[m, n] (the linear's bias gradient, autograd of nn.Linear) formed on the launch from the
 * dY fragments the 8-wave kernel holds in registers (round 6b); M % 64 == 0, N % 128 == 0, K % 128 == 0, else LATTE_ERR_INVALID.
 * workspace: >= splits * (N * K + N) floats. */
int latte_debug_gemm_tn_colsum(const void* dY, const void* X, float* dW, float* colsum, float* workspace, int64_t workspace_floats,
                               int M, int N, int K, int dtype, void* stream);
/* half y = LN(x) * (1 + scale[sample]) + shift[sample]; optional x += temp_embed[frame] first
 * (latte.py:28-29,166,179; :357-358). */
int latte_debug_ln_modulate(float* x, void* y, const float* shift, const float* scale, int mod_stride, int M, int D,
                            int rows_per_sample, const float* temp_embed, int T, int F, int dtype, void* stream);
int latte_debug_convert(const float* in, void* out, int64_t n, int dtype, void* stream);
int latte_debug_fill_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
/* Writes, for every lane l (0..63) and element j (0..3), the LDS element INDEX that
 * ds_read_b64_tr_b16 returned when lane l supplies byte address 8*l (LDS pre-filled with
 * lds[i] = i as 16-bit values): out[l*4 + j].  Documents the transpose-read lane map on this chip. */
int latte_debug_tr16_probe(uint16_t* out, void* stream);

/* 3x3 convolution, padding 1, NHWC half: out[N,H<<ups,W<<ups,Cout] = conv(nearest_upsample^ups(in[N,H,W,Cin]), w) + bias
 * (+ res).  w is the PyTorch weight [Cout,Cin,3,3] in fp32 (packed internally).  (F.conv2d / Upsample2D) */
int latte_debug_conv3x3(const void* in, const float* w, const float* bias, const void* res, void* out, int N, int H, int W,
                        int Cin, int Cout, int ups, int dtype, void* stream);
/* GroupNorm(32 groups, eps 1e-6, affine) [+ SiLU] on NHWC half [N, HW, C]. */
int latte_debug_groupnorm(const void* x, void* y, const float* gamma, const float* beta, int N, int HW, int C, int silu,
                          int dtype, void* stream);

/* The forms the decoder's fp32 residual stream uses: out32[N,Ho,Wo,Cout] = conv(in half) + bias (+ res32), nothing rounded to
 * half; GroupNorm [+ SiLU] of an fp32 NHWC input to a half output. */
/* The 3-tap form of the convolution kernels -- the temporal Conv3d (3,1,1) of AutoencoderKLTemporalDecoder on the "image"
 * [T frames][HW pixels] of one video chunk: in half [T, HW, Cin], w_packed half [Cout, 3 * Cin] (k = ky * Cin + ci), bias [Cout],
 * res32 (may be NULL) / out32 fp32 [T, HW, Cout].  HW may exceed 65535 (512 x 512 frames). */
int latte_debug_conv3rows_f32(const void* in, const void* w_packed, const float* bias, const float* res32, float* out32, int T, int HW,
                              int Cin, int Cout, int dtype, void* stream);
int latte_debug_conv3x3_f32(const void* in, const float* w, const float* bias, const float* res32, float* out32, int N, int H,
                            int W, int Cin, int Cout, int ups, int dtype, void* stream);
int latte_debug_groupnorm_f32(const float* x, void* y, const float* gamma, const float* beta, int N, int HW, int C, int silu,
                              int dtype, void* stream);

/* Runs the VAE decoder (include/latte_amd.h) up to and including stage `stop_after` and returns that stage's NHWC
 * activation (the fp32 residual stream) (trace_dims = N, H, W, C).  Stages: 0 conv_in, 1 mid.resnets.0, 2 mid.attentions.0,
 * 3 mid.resnets.1, then for up block i: three resnets and (i < 3) the upsampler -> 4..18.  Localises a divergence. */
struct latte_vae;
int latte_debug_vae_trace(struct latte_vae* v, const float* z, int n_frames, float z_scale, int stop_after, float* trace_out,
                          int64_t* trace_numel, int* trace_dims, void* stream);

/* Operand-path probe (measurement): 256 workgroups x `waves` waves, every wave issues `reps` bursts of 16 loads over a
 * cache-hot 16 KB window of `src` (>= 8 MiB readable).  mode 0: buffer_load_dwordx4 ... lds, 1: buffer_load_dword ... lds,
 * 2: global_load_dwordx4 into registers, 3: 2 + ds_write_b128.  out: int64 [8][2] = {issue ticks, landed ticks} of
 * workgroup 0 (s_memtime ticks, summed over the bursts). */
int latte_debug_dma_probe(const void* src, long long* out, int mode, int waves, int reps, void* stream);

/* Kernel-choice overrides for the A/B tests (process-global; value 0 restores the library's own choice).  Every offered value
 * selects another implementation of the SAME function (results equal up to rounding):
 *   "attn_variant"    1 = the generic flash kernel for every L > 16, 5 = the streaming kernel for 128 < L <= 256 too, 12 | 13 = the
 *                     round-6c forms of the streaming kernel on 32 x 32 x 16 MFMA tiles for head dim 72, L > 256 (4 waves x 64 queries on
 *                     one wave per SIMD with the softmax inside the P V stream | that pipeline on 8 waves; measured +14 % / -1 % against the default,
 *                     DESIGN.md section 4.2)
 *   "xattn_flash"     1 = the generic flash kernel for text cross-attention instead of the whole-panel kernel
 *   "tn_kernel"       4 = the 4-wave weight-gradient GEMM;   "tn_wn" 4 = its 256 x 128 tile
 *   "attn_bwd_tiles"  1 = the tiled attention-backward kernels for 16-token sequences too, 2 = also for 64 < L <= 256 (instead of the
 *                     resident-image kernels of round 6b)
 *   "conv_kernel"     1 = the plain 128 x 128 implicit-GEMM convolution everywhere, 2 | 3 = the ping-pong 256-pixel kernel everywhere,
 *                     4 = its persistent form (round 6: bit-identical, measured 6 - 13 % slower, not a default anywhere)
 *   "vae_split"       1024 + m: which stages of the temporal decoder run split-operand convolutions -- bits 0..4 of m = the spatial resnets of
 *                     {mid block, up block 0..3} add the pass on the activation's f16 rounding residual, bits 5..9 = the temporal resnets of the
 *                     same stages run three passes (hi*hi + lo*hi + hi*lo) instead of one (csrc/vae_engine.cpp: vae_split_mask)
 * Anything else is refused (LATTE_ERR_INVALID).  Replaces the LATTE_* environment variables round 3 read at every launch; the
 * measurement ablations whose results are garbage (attention variants 7-10, 16-19) exist only in a LATTE_DEBUG_BUILD=1 library. */
int latte_debug_set_choice(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif
