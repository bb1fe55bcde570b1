// Per-kernel test hooks (include/latte_amd_debug.h).
#include "../../include/latte_amd_debug.h"
#include "common.h"

namespace {
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__global__ void tr16_probe_kernel(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[512];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  auto p = (__attribute__((address_space(3))) bf16x4*)((__attribute__((address_space(3))) char*)lds + 8 * lane);
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p);
  typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
  u16x4 u = __builtin_bit_cast(u16x4, v);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = u[j];
}

// ---- operand-path probe: how fast can the waves of one CU move L2-resident data towards LDS / VGPRs?
// mode 0: buffer_load_dwordx4 ... lds (16 B per lane straight into LDS), 1: buffer_load_dword ... lds (4 B per lane),
// 2: global_load_dwordx4 into VGPRs (no LDS), 3: global_load_dwordx4 + ds_write_b128.
// Every wave issues `reps` bursts of 16 instructions over its own 16 KB source window (cache-hot) and reports the
// ticks (s_memtime) it spent ISSUING them and the ticks until they had all landed.
typedef __attribute__((address_space(3))) void lds_void_t;
__global__ void __launch_bounds__(512) dma_probe_kernel(const char* src, long long* out, int mode, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1u << 30, 0x00020000);
  char* my = smem + wave * 16384;
  const unsigned base = ((blockIdx.x & 63) * 8 + wave) * 16384u;
  long long t_issue = 0, t_all = 0;
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  u4 sink = {0, 0, 0, 0};
  __syncthreads();
  if (mode >= 7 && mode <= 9 && wave >= 4) {
    // companion waves (one per SIMD next to a DMA wave): back-to-back MFMAs like a GEMM compute segment.
    // mode 7: at s_setprio 1, mode 8: default priority, mode 9: default priority while the DMA waves run at priority 3
    typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
    typedef __attribute__((ext_vector_type(4))) float f4;
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
    f4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    if (mode == 7) __builtin_amdgcn_s_setprio(1);
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int r = 0; r < reps * 3; ++r) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    __builtin_amdgcn_s_setprio(0);
    f4 sum = acc[0];
    for (int i = 1; i < 16; ++i) sum += acc[i];
    if (lane == 0 && blockIdx.x == 0) {
      out[wave * 2] = t1 - t0;             // ticks for reps * 3 * 64 MFMAs
      out[wave * 2 + 1] = (long long)reps * 3 * 64;
    }
    if (sum[0] == 12345.f) out[101] = 1;
    return;
  }
  if (mode == 9) __builtin_amdgcn_s_setprio(3);
  const int dmode = (mode >= 7 && mode <= 9) ? 0 : mode;
  for (int r = 0; r < reps; ++r) {
    const long long t0 = (long long)__builtin_readcyclecounter();
    if (dmode == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(my + j * 1024), 16, lane * 16u, base + j * 1024u, 0, 0);
    } else if (dmode == 1) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(my + j * 256), 4, lane * 4u, base + j * 256u, 0, 0);
    } else if (dmode == 10 || dmode == 11) {
      // the GEMM's operand pattern: per instruction 8 rows x 128 B (row pitch 2304 B), K tile r of an own 256-row panel,
      // a new panel every 18 K tiles: never TCP-resident.  mode 11: all workgroups of an XCD share 4 panels (L2 reuse).
      const unsigned panel = (dmode == 10 ? blockIdx.x : (blockIdx.x & 7) * 4 + ((blockIdx.x >> 3) & 3)) + 256u * ((unsigned)r / 18u % 4u);
      const unsigned so = panel * 256u * 2304u + (unsigned)(r % 18) * 128u + (unsigned)(wave & 3) * 8u * 2304u;
      const unsigned vo = (unsigned)(lane >> 3) * 2304u + (unsigned)(lane & 7) * 16u;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(my + j * 1024), 16, vo, so + (unsigned)j * 32u * 2304u, 0, 0);
    } else if (dmode == 4) {   // one M0 value for the whole burst (same LDS destination), only the source moves
#pragma unroll
      for (int j = 0; j < 16; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)my, 16, lane * 16u, base + j * 1024u, 0, 0);
    } else if (dmode == 5) {   // one M0 value, LDS destination and source moved by the instruction's immediate offset
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(my + j * 4096), 16, lane * 16u, base + j * 4096u, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(my + j * 4096), 16, lane * 16u, base + j * 4096u, 1024, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(my + j * 4096), 16, lane * 16u, base + j * 4096u, 2048, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(my + j * 4096), 16, lane * 16u, base + j * 4096u, 3072, 0);
      }
    } else if (dmode == 6) {   // global_load ... lds (flat-global addressing) instead of the buffer form
#pragma unroll
      for (int j = 0; j < 16; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + base + j * 1024 + lane * 16),
                                         (lds_void_t*)(my + j * 1024), 16, 0, 0);
    } else {
      u4 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = *(const u4*)(src + base + j * 1024 + lane * 16);
      const long long t1 = (long long)__builtin_readcyclecounter();   // (forces the loads to have been ISSUED only)
      t_issue += t1 - t0;
      if (dmode == 3) {
#pragma unroll
        for (int j = 0; j < 16; ++j) *(u4*)(my + j * 1024 + lane * 16) = v[j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) sink += v[j];
      }
      t_all += (long long)__builtin_readcyclecounter() - t0;
      continue;
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t_issue += t1 - t0;
    t_all += (long long)__builtin_readcyclecounter() - t0;
  }
  if (lane == 0 && blockIdx.x == 0) {
    out[wave * 2] = t_issue;
    out[wave * 2 + 1] = t_all;
  }
  if (sink[0] == 0x12345678u && sink[1] == 1u) out[100] = sink[2];
}
}  // namespace

using namespace latte;
extern "C" {

// out: int64 [8 waves][2] = {ticks spent issuing, ticks until landed}, each summed over `reps` bursts of 16 instructions.
int latte_debug_dma_probe(const void* src_1gib_window, long long* out, int mode, int waves, int reps, void* stream) {
  if (waves < 1 || waves > 8) return fail(LATTE_ERR_INVALID, "dma_probe: waves must be 1..8");
  static std::atomic<uint64_t> attr_done{0};
  if (int rc_ = ensure_dynamic_lds((const void*)dma_probe_kernel, 8 * 16384, attr_done)) return rc_;
  hipLaunchKernelGGL(dma_probe_kernel, dim3(256), dim3(64 * waves), 8 * 16384, (hipStream_t)stream, (const char*)src_1gib_window,
                     out, mode, reps);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int latte_debug_gemm(const void* A, const void* W, const float* bias, void* out, const float* gate, int M, int N, int K,
                     int gate_stride, int rows_per_sample, int epi, int dtype, int variant, void* stream) {
  GemmArgs g{};
  g.A = (const half_t*)A; g.W = (const half_t*)W; g.bias = bias; g.out = out; g.gate = gate;
  g.M = M; g.N = N; g.K = K; g.gate_stride = gate_stride; g.rows_per_sample = rows_per_sample;
  if (epi == EPI_BIAS_RES_H16) g.res = (const half_t*)gate;   // epi 5: `gate` carries the half residual [Mpad, N]
  if (variant >= 1000) { g.tag = variant / 1000; variant %= 1000; }   // + 1000 * call-site tag (GemmArgs::tag)
  return launch_gemm(g, epi, dtype, variant, (hipStream_t)stream);
}

int latte_debug_gemm_gelu(const void* A, const void* W, const float* bias, void* out, void* aux, int M, int N, int K, int epi, int dtype,
                          void* stream) {
  if (epi != EPI_BIAS_GELU_DUAL_H16 && epi != EPI_DGELU_H16) return fail(LATTE_ERR_INVALID, "gemm_gelu: epi 13 (dual GELU) or 14 (dGELU)");
  GemmArgs g{};
  g.A = (const half_t*)A; g.W = (const half_t*)W; g.bias = bias; g.out = out; g.aux = (half_t*)aux;
  g.M = M; g.N = N; g.K = K; g.rows_per_sample = M;
  return launch_gemm_pw(g, epi, dtype, 1, (hipStream_t)stream);
}

int latte_debug_gemm_lo8(const void* A, const void* W, const void* A8, const void* W8, const float* bias, void* out, const float* gate,
                         int M, int N, int K, int gate_stride, int rows_per_sample, int epi, int dtype, void* stream) {
  if (!A8 || !W8) return fail(LATTE_ERR_INVALID, "gemm_lo8: null fp8 operand");
  GemmArgs g{};
  g.A = (const half_t*)A; g.W = (const half_t*)W; g.bias = bias; g.out = out; g.gate = gate;
  g.M = M; g.N = N; g.K = K; g.gate_stride = gate_stride; g.rows_per_sample = rows_per_sample;
  g.A8 = (const uint8_t*)A8; g.W8 = (const uint8_t*)W8;
  return launch_gemm(g, epi, dtype, 0, (hipStream_t)stream);
}

int latte_debug_gemm_lo4(const void* A, const void* W, const void* A4, const void* A4s, const void* W4, const void* W4s, const float* bias,
                         void* out, const float* gate, int M, int N, int K, int gate_stride, int rows_per_sample, int epi, int dtype,
                         void* stream) {
  if (!A4 || !A4s || !W4 || !W4s) return fail(LATTE_ERR_INVALID, "gemm_lo4: null fp4 operand");
  GemmArgs g{};
  g.A = (const half_t*)A; g.W = (const half_t*)W; g.bias = bias; g.out = out; g.gate = gate;
  g.M = M; g.N = N; g.K = K; g.gate_stride = gate_stride; g.rows_per_sample = rows_per_sample;
  g.A4 = (const uint8_t*)A4; g.A4s = (const uint8_t*)A4s; g.W4 = (const uint8_t*)W4; g.W4s = (const uint8_t*)W4s;
  return launch_gemm(g, epi, dtype, 0, (hipStream_t)stream);
}

int latte_debug_pack_w4(const void* w, void* out4, void* out_scale, int N, int K, int dtype, void* stream) {
  return launch_pack_w4((const half_t*)w, (unsigned char*)out4, (unsigned char*)out_scale, N, K, dtype, (hipStream_t)stream);
}

int latte_debug_ln_modulate_split4(const float* x, void* y, void* y4, void* y4s, const float* shift, const float* scale, int mod_stride, int M,
                                   int D, int rows_per_sample, int dtype, void* stream) {
  return launch_ln_modulate_split4(x, (half_t*)y, (unsigned char*)y4, (unsigned char*)y4s, shift, scale, mod_stride, M, D, rows_per_sample,
                                   dtype, (hipStream_t)stream);
}

int latte_debug_pack_w8(const void* w, void* out8, int64_t n, int dtype, void* stream) {
  return launch_pack_w8((const half_t*)w, (unsigned char*)out8, n, dtype, (hipStream_t)stream);
}

int latte_debug_ln_modulate_split8(const float* x, void* y, void* y8, const float* shift, const float* scale, int mod_stride, int M, int D,
                                   int rows_per_sample, int dtype, void* stream) {
  return launch_ln_modulate(x, nullptr, (half_t*)y, shift, scale, mod_stride, M, D, rows_per_sample, nullptr, 1, 1, dtype, (hipStream_t)stream, 2,
                            (unsigned char*)y8);
}

int latte_debug_qkv_attention_split8(const void* xn, const void* w, const float* bias, void* out, void* out8, int B, int F, int T, int D,
                                     int heads, int mode, int dtype, void* stream) {
  if (heads <= 0 || D % heads) return fail(LATTE_ERR_INVALID, "qkv_attention: D must be a multiple of heads");
  QkvAttnArgs a{};
  a.xn = (const half_t*)xn; a.w = (const half_t*)w; a.bias = bias; a.out = (half_t*)out; a.out8 = (unsigned char*)out8;
  a.B = B; a.F = F; a.T = T; a.D = D; a.heads = heads; a.hd = D / heads; a.mode = mode; a.flags = 7; a.out_split = 2;
  a.scale = 1.0f / sqrtf((float)a.hd);
  return launch_qkv_attention(a, dtype, (hipStream_t)stream);
}

int latte_debug_gemm_choice(int M, int N, int K, int epi) { return gemm_resolve_variant(M, N, K, epi); }

int latte_debug_qkv_attention_fusable(int D, int heads, int F, int T, int mode, int64_t rows) {
  return heads > 0 && D % heads == 0 && qkv_attention_fusable(D, heads, D / heads, F, T, mode, rows) ? 1 : 0;
}

int latte_debug_gemm_tn_plan(int M, int N, int K, int* rows_per_split) {
  int chunk = 0;
  const int splits = gemm_tn_plan(M, N, K, &chunk);
  if (rows_per_split) *rows_per_split = chunk;
  return splits;
}

int latte_debug_attention(const void* qkv, void* out, int num_seq, int L, int heads, int hd, int U, int64_t sample_stride,
                          int64_t seq_stride, int64_t row_stride, int dtype, void* stream) {
  AttnArgs a{};
  a.qkv = (const half_t*)qkv; a.out = (half_t*)out; a.num_seq = num_seq; a.L = L; a.heads = heads; a.hd = hd;
  a.D = heads * hd; a.U = U; a.sample_stride = sample_stride; a.seq_stride = seq_stride; a.row_stride = row_stride;
  a.scale = 1.0f / sqrtf((float)hd);
  return launch_attention(a, dtype, (hipStream_t)stream);
}

int latte_debug_attention_split8(const void* qkv, void* out, void* out8, int num_seq, int L, int heads, int hd, int U, int64_t sample_stride,
                                 int64_t seq_stride, int64_t row_stride, int dtype, void* stream) {
  AttnArgs a{};
  a.qkv = (const half_t*)qkv; a.out = (half_t*)out; a.out8 = (unsigned char*)out8; a.num_seq = num_seq; a.L = L; a.heads = heads; a.hd = hd;
  a.D = heads * hd; a.U = U; a.sample_stride = sample_stride; a.seq_stride = seq_stride; a.row_stride = row_stride;
  a.scale = 1.0f / sqrtf((float)hd);
  return launch_attention(a, dtype, (hipStream_t)stream);
}

int latte_debug_qkv_attention(const void* xn, const void* w, const float* bias, void* out, void* dbg_qkv, int B, int F, int T, int D,
                              int heads, int mode, int flags, int dtype, void* stream) {
  return latte_debug_qkv_attention_trace(xn, w, bias, out, dbg_qkv, nullptr, B, F, T, D, heads, mode, flags, dtype, stream);
}

int latte_debug_qkv_attention_trace(const void* xn, const void* w, const float* bias, void* out, void* dbg_qkv, long long* trace, int B,
                                    int F, int T, int D, int heads, int mode, int flags, int dtype, void* stream) {
  if (heads <= 0 || D % heads) return fail(LATTE_ERR_INVALID, "qkv_attention: D must be a multiple of heads");
  QkvAttnArgs a{};
  a.xn = (const half_t*)xn; a.w = (const half_t*)w; a.bias = bias; a.out = (half_t*)out; a.dbg_qkv = (half_t*)dbg_qkv;
  a.dbg_trace = trace;
  a.B = B; a.F = F; a.T = T; a.D = D; a.heads = heads; a.hd = D / heads; a.mode = mode; a.flags = flags & 255;
  a.out_split = (flags >> 8) & 1;   // flags bit 8: out is [rows, 2 D] = [hi | lo] (the split-pair output of guided calls)
  a.scale = 1.0f / sqrtf((float)a.hd);
  return launch_qkv_attention(a, dtype, (hipStream_t)stream);
}

int latte_debug_attention_bwd(const void* qkv, const void* o, const void* dout, void* dqkv, float* stats, int num_seq, int L, int heads,
                               int hd, int U, int64_t sample_stride, int64_t seq_stride, int64_t row_stride, int dtype, void* stream) {
  return launch_attention_bwd((const half_t*)qkv, (const half_t*)o, (const half_t*)dout, (half_t*)dqkv, stats, num_seq, L, heads, hd, U,
                              sample_stride, seq_stride, row_stride, dtype, (hipStream_t)stream);
}

int latte_debug_gemm_tn(const void* dY, const void* X, float* dW, float* workspace, int64_t workspace_floats, int M, int N, int K,
                        int dtype, void* stream) {
  int chunk = 0;
  const int splits = gemm_tn_plan(M, N, K, &chunk);
  if ((int64_t)splits * N * K > workspace_floats) return fail(LATTE_ERR_INVALID, "gemm_tn: workspace too small");
  if (int rc = launch_gemm_tn((const half_t*)dY, (const half_t*)X, workspace, M, N, K, chunk, dtype, (hipStream_t)stream)) return rc;
  return launch_split_reduce(workspace, splits, (size_t)N * K, (size_t)N * K, dW, 0, (hipStream_t)stream);
}

// the same product with the column sums of dY riding on the launch (gemm_tn8_kernel<DT, true>: the bias gradient of the linear);
// workspace: >= splits * (N * K + N) floats.  Shapes the 8-wave kernel does not take are refused.
int latte_debug_gemm_tn_colsum(const void* dY, const void* X, float* dW, float* colsum, float* workspace, int64_t workspace_floats, int M,
                               int N, int K, int dtype, void* stream) {
  int chunk = 0;
  const int splits = gemm_tn_plan(M, N, K, &chunk);
  if (!gemm_tn8_ok(M, N, K)) return fail(LATTE_ERR_INVALID, "gemm_tn_colsum: the column sums ride on the 8-wave kernel only (M % 64, N % 128, K % 128)");
  if ((int64_t)splits * ((int64_t)N * K + N) > workspace_floats) return fail(LATTE_ERR_INVALID, "gemm_tn_colsum: workspace too small");
  float* cs = workspace + (size_t)splits * N * K;
  if (int rc = launch_gemm_tn((const half_t*)dY, (const half_t*)X, workspace, M, N, K, chunk, dtype, (hipStream_t)stream, cs)) return rc;
  if (int rc = launch_split_reduce(workspace, splits, (size_t)N * K, (size_t)N * K, dW, 0, (hipStream_t)stream)) return rc;
  return launch_split_reduce(cs, splits, (size_t)N, (size_t)N, colsum, 0, (hipStream_t)stream);
}

int latte_debug_ln_modulate(float* x, void* y, const float* shift, const float* scale, int mod_stride, int M, int D,
                            int rows_per_sample, const float* temp_embed, int T, int F, int dtype, void* stream) {
  return launch_ln_modulate(x, x, (half_t*)y, shift, scale, mod_stride, M, D, rows_per_sample, temp_embed, T, F, dtype,
                            (hipStream_t)stream);
}

int latte_debug_convert(const float* in, void* out, int64_t n, int dtype, void* stream) {
  return launch_convert_f32_to_h16(in, (half_t*)out, n, dtype, (hipStream_t)stream);
}

int latte_debug_fill_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
  return launch_fill_normal(out, (size_t)n, seed, offset, (hipStream_t)stream);
}

int latte_debug_conv3x3(const void* in, const float* w, const float* bias, const void* res, void* out, int N, int H, int W,
                        int Cin, int Cout, int ups, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  half_t *wp = nullptr, *zeros = nullptr;
  LATTE_HIP(hipMalloc((void**)&wp, (size_t)Cout * Cin * 9 * 2));
  LATTE_HIP(hipMalloc((void**)&zeros, 64));
  LATTE_HIP(hipMemsetAsync(zeros, 0, 64, st));
  int rc = launch_pack_conv_w(w, wp, Cout, Cin, dtype, st);
  if (!rc) rc = launch_conv3x3((const half_t*)in, wp, bias, (const half_t*)res, (half_t*)out, zeros, N, H, W, Cin, Cout, ups, dtype, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(wp);
  (void)hipFree(zeros);
  return rc;
}

/* The decoder's fp32-stream forms: out32 = conv(in) + bias (+ res32), and GroupNorm of an fp32 input. */
// The 3-tap form of the convolution kernels (the temporal Conv3d (3,1,1) of AutoencoderKLTemporalDecoder on the "image" [T frames][HW
// pixels]): in half [T, HW, Cin], w_packed half [Cout, 3 * Cin] (k = ky * Cin + ci), fp32 residual / output [T, HW, Cout].
int latte_debug_conv3rows_f32(const void* in, const void* w_packed, const float* bias, const float* res32, float* out32, int T, int HW,
                              int Cin, int Cout, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  half_t* zeros = nullptr;
  LATTE_HIP(hipMalloc((void**)&zeros, 64));
  LATTE_HIP(hipMemsetAsync(zeros, 0, 64, st));
  int rc = launch_conv3x3((const half_t*)in, (const half_t*)w_packed, bias, nullptr, nullptr, zeros, 1, T, HW, Cin, Cout, 0, dtype, st, res32, out32, 1);
  (void)hipStreamSynchronize(st);
  (void)hipFree(zeros);
  return rc;
}

int latte_debug_conv3x3_f32(const void* in, const float* w, const float* bias, const float* res32, float* out32, int N, int H,
                            int W, int Cin, int Cout, int ups, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  half_t *wp = nullptr, *zeros = nullptr;
  LATTE_HIP(hipMalloc((void**)&wp, (size_t)Cout * Cin * 9 * 2));
  LATTE_HIP(hipMalloc((void**)&zeros, 64));
  LATTE_HIP(hipMemsetAsync(zeros, 0, 64, st));
  int rc = launch_pack_conv_w(w, wp, Cout, Cin, dtype, st);
  if (!rc) rc = launch_conv3x3((const half_t*)in, wp, bias, nullptr, nullptr, zeros, N, H, W, Cin, Cout, ups, dtype, st, res32, out32);
  (void)hipStreamSynchronize(st);
  (void)hipFree(wp);
  (void)hipFree(zeros);
  return rc;
}

int latte_debug_groupnorm_f32(const float* x, void* y, const float* gamma, const float* beta, int N, int HW, int C, int silu,
                              int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  float *partial = nullptr, *stats = nullptr;
  LATTE_HIP(hipMalloc((void**)&partial, (size_t)N * groupnorm_max_slabs() * 64 * 4));
  LATTE_HIP(hipMalloc((void**)&stats, (size_t)N * 64 * 4));
  int rc = launch_groupnorm(x, 1, (half_t*)y, gamma, beta, partial, stats, N, HW, C, silu, dtype, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(partial);
  (void)hipFree(stats);
  return rc;
}

int latte_debug_groupnorm(const void* x, void* y, const float* gamma, const float* beta, int N, int HW, int C, int silu,
                          int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  float *partial = nullptr, *stats = nullptr;
  LATTE_HIP(hipMalloc((void**)&partial, (size_t)N * groupnorm_max_slabs() * 64 * 4));
  LATTE_HIP(hipMalloc((void**)&stats, (size_t)N * 64 * 4));
  int rc = launch_groupnorm(x, 0, (half_t*)y, gamma, beta, partial, stats, N, HW, C, silu, dtype, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(partial);
  (void)hipFree(stats);
  return rc;
}

int latte_debug_set_choice(const char* name, int value) {
  if (set_debug_choice(name, value) != 0)
    return fail(LATTE_ERR_INVALID, std::string("latte_debug_set_choice: unknown name or value not offered by this build: ") + (name ? name : "(null)"));
  return LATTE_OK;
}

int latte_debug_tr16_probe(uint16_t* out, void* stream) {
  hipLaunchKernelGGL(tr16_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

}  // extern "C"
