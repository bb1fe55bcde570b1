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
}  // namespace

using namespace latte;
extern "C" {

int latte_debug_gemm(const void* A, const void* W, const float* bias, void* out, const float* gate, int M, int N, int K,
                     int gate_stride, int rows_per_sample, int epi, int dtype, int variant, void* stream) {
  GemmArgs g{};
  g.A = (const half_t*)A; g.W = (const half_t*)W; g.bias = bias; g.out = out; g.gate = gate;
  g.M = M; g.N = N; g.K = K; g.gate_stride = gate_stride; g.rows_per_sample = rows_per_sample;
  if (epi == EPI_BIAS_RES_H16) g.res = (const half_t*)gate;   // epi 5: `gate` carries the half residual [Mpad, N]
  return launch_gemm(g, epi, dtype, variant, (hipStream_t)stream);
}

int latte_debug_attention(const void* qkv, void* out, int num_seq, int L, int heads, int hd, int U, int64_t sample_stride,
                          int64_t seq_stride, int64_t row_stride, int dtype, void* stream) {
  AttnArgs a{};
  a.qkv = (const half_t*)qkv; a.out = (half_t*)out; a.num_seq = num_seq; a.L = L; a.heads = heads; a.hd = hd;
  a.D = heads * hd; a.U = U; a.sample_stride = sample_stride; a.seq_stride = seq_stride; a.row_stride = row_stride;
  a.scale = 1.0f / sqrtf((float)hd);
  return launch_attention(a, dtype, (hipStream_t)stream);
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

int latte_debug_groupnorm(const void* x, void* y, const float* gamma, const float* beta, int N, int HW, int C, int silu,
                          int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  float *partial = nullptr, *stats = nullptr;
  LATTE_HIP(hipMalloc((void**)&partial, (size_t)N * groupnorm_max_slabs() * 64 * 4));
  LATTE_HIP(hipMalloc((void**)&stats, (size_t)N * 64 * 4));
  int rc = launch_groupnorm((const half_t*)x, (half_t*)y, gamma, beta, partial, stats, N, HW, C, silu, dtype, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(partial);
  (void)hipFree(stats);
  return rc;
}

int latte_debug_tr16_probe(uint16_t* out, void* stream) {
  hipLaunchKernelGGL(tr16_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

}  // extern "C"
