// Weight-gradient GEMM of the training step:  dW[N, K] = sum_m dY[m, n] X[m, k]  with BOTH operands row-major in the
// contraction index m (dY [M, N] and X [M, K] exactly as the backward / the forward left them) -- no transposed copies.
//
// The MFMA wants 8 consecutive contraction elements per lane; in these layouts they are a COLUMN of the tile.  The tiles are
// therefore staged row-major in LDS ([64 m][128 columns], pitch 288 B) and the fragments come out through the hardware transpose
// read (ds_read_b64_tr_b16: lane i of a 16-lane group supplies the address of 4 columns of row (i >> 2) and receives the 4 rows of
// column i), two reads per 16 x 32 fragment, the same pattern for both operands -- so both see the contraction index in the same
// (permuted) slot order and the product is exact.  Pitch 288 B = 32 x 9: the 8 rows x 32 B of a 32-lane service group tile the 64
// banks.  128 x 128 output tile (a 128 x 256 instantiation exists for measurements: see the launcher), 4 waves (2 x 2, wave tile
// 64 x 64), 64 rows of m per stage, two stages, register-staged global loads (rows of a tile are 256 B contiguous).  The contraction (M = 20 480 rows at Latte-B/2, batch 5) is split over grid.y;
// every split ASSIGNS its fp32 partial product to its own slab (fixed-order reduction afterwards: deterministic).
#include <cstdlib>

#include "common.h"
#include "mfma_util.h"

namespace latte {
namespace {

typedef __attribute__((__vector_size__(4 * sizeof(short)))) short i16v4t;
__device__ __forceinline__ u32x2 tr16t(const char* p) {
  i16v4t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16v4t*)p);
  return __builtin_bit_cast(u32x2, v);
}

constexpr int TN_PITCH = 288;

struct TnArgs {
  const half_t* dY;   // [M, N]
  const half_t* X;    // [M, K]
  float* out;         // [splits][N][K] partial products
  int M, N, K, m_chunk;
};

template <int DT, int KW>   // KW = 1 | 2: 128 KW columns of X (rows of... columns of dW) per tile
__global__ void __launch_bounds__(256) gemm_tn_kernel(TnArgs g) {
  constexpr int PITCH_B = KW == 1 ? TN_PITCH : 544;   // 544 = 32 x 17: conflict-free as 288 (odd multiple of 32 B)
  constexpr int IMG = 64 * TN_PITCH;            // the dY tile image
  constexpr int IMGB = 64 * PITCH_B;            // the X tile image
  constexpr int STG = IMG + IMGB;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x 2 operand images = 72 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;      // wave tile: n rows [64 wn, +64), k columns [64 wk, +64)
  const int fl = lane & 15, gq = lane >> 4;
  const int tiles_k = g.K / (128 * KW);
  const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
  const int n0 = tn * 128, k0 = tk * 128 * KW;
  const int m_begin = blockIdx.y * g.m_chunk;
  const int m_end = min(g.M, m_begin + g.m_chunk);
  float* outp = g.out + (size_t)blockIdx.y * g.N * g.K;

  // staging: a tile is 64 rows x 16 chunks of 16 B; thread handles chunks (row = tid / 16 + 16 j, chunk = tid % 16), j = 0..3
  const int srow = tid >> 4, sch = tid & 15;
  const bool n_ok = n0 + sch * 8 < g.N;         // partial last tile row of dW (N not a multiple of 128): zero columns
  auto load_tiles = [&](int m0, u32x4* ra, u32x4* rb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + srow + 16 * j;
      ra[j] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
      for (int w = 0; w < KW; ++w) rb[j * KW + w] = (u32x4){0u, 0u, 0u, 0u};
      if (m < m_end) {
        if (n_ok) ra[j] = *(const u32x4*)(g.dY + (size_t)m * g.N + n0 + sch * 8);
#pragma unroll
        for (int w = 0; w < KW; ++w) rb[j * KW + w] = *(const u32x4*)(g.X + (size_t)m * g.K + k0 + w * 128 + sch * 8);
      }
    }
  };
  auto store_tiles = [&](int buf, const u32x4* ra, const u32x4* rb) {
    char* a = smem + buf * STG;
    char* b = a + IMG;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *(u32x4*)(a + (srow + 16 * j) * TN_PITCH + sch * 16) = ra[j];
#pragma unroll
      for (int w = 0; w < KW; ++w) *(u32x4*)(b + (srow + 16 * j) * PITCH_B + w * 256 + sch * 16) = rb[j * KW + w];
    }
  };

  constexpr int NJ = 4 * KW;                    // 16-column fragments of X per wave
  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 ra[4], rb[4 * KW];
  load_tiles(m_begin, ra, rb);
  store_tiles(0, ra, rb);
  __syncthreads();
  int buf = 0;
  for (int m0 = m_begin; m0 < m_end; m0 += 64, buf ^= 1) {
    const bool more = m0 + 64 < m_end;
    if (more) load_tiles(m0 + 64, ra, rb);              // global loads in flight under the MFMAs
    const char* a = smem + buf * STG;
    const char* b = a + IMG;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                    // 32 rows of m per MFMA
      // fragment of column block c (16 columns) of an image: contraction rows {32 ks + 4 gq + 0..3} and {+16}
      const int rrow = 32 * ks + 4 * gq + (fl >> 2), rcol = (fl & 3) * 8;
      u32x4 nf[4], kf[NJ];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const char* pa = a + rrow * TN_PITCH + rcol + (wn * 64 + c * 16) * 2;
        const u32x2 alo = tr16t(pa), ahi = tr16t(pa + 16 * TN_PITCH);
        nf[c] = (u32x4){alo[0], alo[1], ahi[0], ahi[1]};
      }
#pragma unroll
      for (int c = 0; c < NJ; ++c) {
        const char* pb = b + rrow * PITCH_B + rcol + (wk * 64 * KW + c * 16) * 2;
        const u32x2 blo = tr16t(pb), bhi = tr16t(pb + 16 * PITCH_B);
        kf[c] = (u32x4){blo[0], blo[1], bhi[0], bhi[1]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16<DT>(kf[j], nf[i], acc[i][j]);   // D[k = 4 gq + r][n = fl]
    }
    if (more) store_tiles(buf ^ 1, ra, rb);
    __syncthreads();
  }
  // lane holds dW[n = n0 + 64 wn + 16 i + fl][k = k0 + 64 KW wk + 16 j + 4 gq + {0..3}]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + wn * 64 + i * 16 + fl;
    if (n >= g.N) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int k = k0 + wk * 64 * KW + j * 16 + gq * 4;
      *(float4*)(outp + (size_t)n * g.K + k) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

}  // namespace

// partial: float [splits][N][K] with splits = ceil(M / m_chunk) (m_chunk a multiple of 64); K % 128 == 0, N % 8 == 0
int launch_gemm_tn(const half_t* dY, const half_t* X, float* partial, int M, int N, int K, int m_chunk, int dtype, hipStream_t st) {
  if (K % 128 || N % 8 || m_chunk % 64 || m_chunk <= 0) return fail(LATTE_ERR_INVALID, "gemm_tn: need K % 128 == 0, N % 8 == 0, m_chunk % 64 == 0");
  TnArgs a{dY, X, partial, M, N, K, m_chunk};
  const int splits = (M + m_chunk - 1) / m_chunk;
  // 128 x 128 tiles.  The 128 x 256 tile (KW = 2; LATTE_TN_KW=2 selects it for measurements) has 25 % fewer LDS bytes per MFMA but
  // 312 registers and 106 KB of LDS: one workgroup per CU instead of two, and it measured SLOWER (training step 26.96 against
  // 24.97 ms at Latte-B/2, batch 5): the second resident workgroup hides the global-load latency and the barrier bubbles.
  int kw = 1;
  if (const char* e_ = getenv("LATTE_TN_KW")) { if (atoi(e_) == 2 && K % 256 == 0) kw = 2; }
  dim3 grid(((N + 127) / 128) * (K / (128 * kw)), splits), block(256);
  const int lds = 2 * (64 * TN_PITCH + 64 * (kw == 1 ? TN_PITCH : 544));
#define LATTE_TN_CASE(DT, KW)                                                                       \
  {                                                                                                 \
    static std::atomic<uint64_t> done{0};                                                           \
    if (int rc = ensure_dynamic_lds((const void*)gemm_tn_kernel<DT, KW>, lds, done)) return rc;     \
    hipLaunchKernelGGL((gemm_tn_kernel<DT, KW>), grid, block, lds, st, a);                          \
  }
  if (dtype == LATTE_DTYPE_BF16) { if (kw == 2) LATTE_TN_CASE(LATTE_DTYPE_BF16, 2) else LATTE_TN_CASE(LATTE_DTYPE_BF16, 1) }
  else if (dtype == LATTE_DTYPE_F16) { if (kw == 2) LATTE_TN_CASE(LATTE_DTYPE_F16, 2) else LATTE_TN_CASE(LATTE_DTYPE_F16, 1) }
  else return fail(LATTE_ERR_INVALID, "gemm_tn: unknown dtype");
#undef LATTE_TN_CASE
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

}  // namespace latte
