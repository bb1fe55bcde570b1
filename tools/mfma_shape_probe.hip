// Measurement only (not part of the library): the two f16 MFMA shapes of gfx950 on the SAME wave tile under the socket's power cap.
// One workgroup of 8 waves per CU (two per SIMD, as the GEMM kernels run); a wave owns a 64 x 96 tile = 96 fp32 accumulators and per
// "K step" of 64 multiplies 8 A fragments by 12 B fragments (16 bytes per lane each, random f16 ~ N(0, 1)):
//   shape 16: v_mfma_f32_16x16x32_f16, 4 x 6 tiles x 2 k-halves = 48 instructions of 16 clocks
//   shape 32: v_mfma_f32_32x32x16_f16, 2 x 3 tiles x 4 k-quarters = 24 instructions of 32 clocks
// ORDER (shape 16): 0 = A fragment outer, B inner (the GEMM kernels' order); 1 = the same, serpentine; 2 = B outer, A inner.
// DMA = n: every wave also issues n LDS-DMA pieces of 1 KB per K step out of a 1 MB pool (L2 hits): the GEMM's operand transport without
// its waits (7 = the 256 x 192 tile's 56 KB per K step and CU).
// LDS = 0: fragments stay in registers; LDS = 1: the 20 fragments are re-read from the LDS every K step (20 ds_read_b128 per wave, the
// fragment traffic of the 12-wave GEMM's consumer waves).  Prints TFLOP/s over launches long enough for the power cap to settle.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_probe.hip -o /tmp/mfma_shape_probe && /tmp/mfma_shape_probe [seconds per case]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SHAPE, int LDS, int ORDER = 0, int DMA = 0>
__global__ void __launch_bounds__(512) probe(const h8* __restrict__ src, float* __restrict__ out, int iters, const char* __restrict__ pool) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  h8 a[8], b[12];
#pragma unroll
  for (int f = 0; f < 8; ++f) a[f] = src[(f * 8 + wave) * 64 + lane];
#pragma unroll
  for (int f = 0; f < 12; ++f) b[f] = src[((8 + f) * 8 + wave) * 64 + lane];
  h8* mine = (h8*)smem + wave * 20 * 64 + lane;
  if (LDS) {
#pragma unroll
    for (int f = 0; f < 8; ++f) mine[f * 64] = a[f];
#pragma unroll
    for (int f = 0; f < 12; ++f) mine[(8 + f) * 64] = b[f];
    __syncthreads();
  }
  if constexpr (SHAPE == 16) {
    f4 acc[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      if constexpr (DMA > 0) {   // operand transport beside the MFMAs: DMA pieces of 1 KB per wave and K step, L2 -> LDS, nobody reads them
#pragma unroll
        for (int p = 0; p < DMA; ++p)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pool + (size_t)(((it * DMA + p) & 127) * 8 + wave) * 1024 + lane * 16),
                                           (__attribute__((address_space(3))) void*)(smem + (LDS ? 8 * 20 * 64 * 16 : 0) + (wave * 8 + (p & 7)) * 1024), 16, 0, 0);
      }
      if (LDS) {
#pragma unroll
        for (int f = 0; f < 8; ++f) a[f] = mine[f * 64];
#pragma unroll
        for (int f = 0; f < 12; ++f) b[f] = mine[(8 + f) * 64];
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (ORDER == 2) {   // B fragment outer, A fragment inner
#pragma unroll
          for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[ks * 6 + j], a[ks * 4 + i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
              const int j = (ORDER == 1 && (i & 1)) ? 5 - jj : jj;   // ORDER 1: serpentine, one operand changes per instruction
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[ks * 6 + j], a[ks * 4 + i], acc[i][j], 0, 0, 0);
            }
        }
        if constexpr (ORDER != 0) __builtin_amdgcn_sched_barrier(0);
      }
    }
    f4 s = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) s += acc[i][j];
    out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  } else {
    f16v acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if (LDS) {
#pragma unroll
        for (int f = 0; f < 8; ++f) a[f] = mine[f * 64];
#pragma unroll
        for (int f = 0; f < 12; ++f) b[f] = mine[(8 + f) * 64];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ks * 3 + j], a[ks * 2 + i], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

template <int SHAPE, int LDS, int ORDER = 0, int DMA = 0>
static void run(const h8* src, float* out, double seconds, int quiet, const char* pool = nullptr) {
  const int grid = 256, iters = 20000;
  const size_t lds = (LDS ? 8 * 20 * 64 * 16 : 0) + (DMA ? 64 * 1024 : 0);   // LDS and DMA together do not fit: 160 + 64 KB
  CK(hipFuncSetAttribute((const void*)probe<SHAPE, LDS, ORDER, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double flop = 2.0 * 64 * 96 * 64 * (double)iters * 8 * grid;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // settle: launches for `seconds`, the last third timed
  probe<SHAPE, LDS, ORDER, DMA><<<grid, 512, lds>>>(src, out, iters, pool);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  probe<SHAPE, LDS, ORDER, DMA><<<grid, 512, lds>>>(src, out, iters, pool);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms1;
  CK(hipEventElapsedTime(&ms1, e0, e1));
  const int n = (int)(seconds * 1000.0 / ms1) + 3, warm = 2 * n / 3;
  for (int i = 0; i < warm; ++i) probe<SHAPE, LDS, ORDER, DMA><<<grid, 512, lds>>>(src, out, iters, pool);
  CK(hipEventRecord(e0));
  for (int i = warm; i < n; ++i) probe<SHAPE, LDS, ORDER, DMA><<<grid, 512, lds>>>(src, out, iters, pool);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double tf = flop * (n - warm) / (ms * 1e-3) / 1e12;
  printf("shape %2d order %d DMA %2d KB/K step/CU  fragments %s  %s operands: first launch %.2f ms = %.0f TFLOP/s, settled (%d launches after %d) %.0f TFLOP/s = %.3f of 2500\n",
         SHAPE, ORDER, DMA * 8, LDS ? "re-read from LDS" : "in registers     ", quiet ? "all-zero" : "random  ", ms1, flop / (ms1 * 1e-3) / 1e12, n - warm,
         warm + 2, tf, tf / 2500.0);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  const size_t nfrag = 20 * 8 * 64;
  std::vector<_Float16> h(1 << 19);
  unsigned s = 12345u;
  for (auto& v : h) {   // sum of 4 uniforms: ~N(0, 1) after scaling
    float acc = 0.f;
    for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; acc += (float)(s >> 8) / 16777216.f - 0.5f; }
    v = (_Float16)(acc * 1.7320508f);
  }
  h8* src;
  float* out;
  CK(hipMalloc(&src, nfrag * 16));
  CK(hipMalloc(&out, 256 * 512 * 4));
  char* pool;   // 1 MB that every CU streams through: L2-resident operand panels
  CK(hipMalloc(&pool, 1 << 20));
  CK(hipMemcpy(pool, h.data(), 1 << 20, hipMemcpyHostToDevice));
  for (int quiet = 0; quiet < 2; ++quiet) {
    if (quiet) CK(hipMemset(src, 0, nfrag * 16)); else CK(hipMemcpy(src, h.data(), nfrag * 16, hipMemcpyHostToDevice));
    run<16, 0>(src, out, seconds, quiet);
    run<32, 0>(src, out, seconds, quiet);
    run<16, 1>(src, out, seconds, quiet);
    run<32, 1>(src, out, seconds, quiet);
    run<16, 0, 1>(src, out, seconds, quiet);
    run<16, 0, 2>(src, out, seconds, quiet);
    run<16, 0, 0, 7>(src, out, seconds, quiet, pool);    // the 256 x 192 tile's 56 KB per K step
    run<16, 0, 0, 14>(src, out, seconds, quiet, pool);
  }
  return 0;
}
