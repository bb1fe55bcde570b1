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

int latte_debug_tr16_probe(uint16_t* out, void* stream) {
  hipLaunchKernelGGL(tr16_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

}  // extern "C"
