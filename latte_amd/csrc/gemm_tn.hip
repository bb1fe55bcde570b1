// Weight-gradient GEMM of the training step:  dW[N, K] = sum_m dY[m, n] X[m, k]  with BOTH operands row-major in the
// contraction index m (dY [M, N] and X [M, K] exactly as the backward / the forward left them) -- no transposed copies.
//
// The MFMA wants 8 consecutive contraction elements per lane; in these layouts they are a COLUMN of the tile.  The tiles are
// therefore staged row-major in LDS ([64 m][128 columns], pitch 288 B) and the fragments come out through the hardware transpose
// read (ds_read_b64_tr_b16: lane i of a 16-lane group supplies the address of 4 columns of row (i >> 2) and receives the 4 rows of
// column i), two reads per 16 x 32 fragment, the same pattern for both operands -- so both see the contraction index in the same
// (permuted) slot order and the product is exact.  Pitch 288 B = 32 x 9: the 8 rows x 32 B of a 32-lane service group tile the 64
// banks.  128 x 128 output tile on 4 waves (2 x 2, wave tile 64 x 64), two workgroups per CU (a 256 x 128 / 8-wave instantiation
// is kept for measurements: slower, see the launcher), 64 rows of m per stage, two stages, register-staged global loads (rows
// of a tile are 256 B contiguous).  The contraction (M = 20 480 rows at Latte-B/2, batch 5) is split over grid.y;
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

template <int DT, int WN>   // WN = 2 | 4 wave rows: output tile 64 WN (n) x 128 (k), 2 WN waves
__global__ void __launch_bounds__(WN * 128) gemm_tn_kernel(TnArgs g) {
  constexpr int NT = WN * 128;                          // threads
  constexpr int TILE_N = 64 * WN;
  constexpr int PITCH_A = TILE_N == 128 ? 288 : 544;    // odd multiples of 32 B: conflict-free transpose reads
  constexpr int IMGA = 64 * PITCH_A, IMGB = 64 * TN_PITCH, STG = IMGA + IMGB;
  constexpr int CHA = TILE_N / 8;                       // 16-byte chunks per dY tile row
  constexpr int JA = 64 * CHA / NT, JB = 64 * 16 / NT;  // chunks per thread and stage: 4 / 4 (WN = 2), 4 / 2 (WN = 4)
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (dY image + X image)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;              // wave tile: n rows [64 wn, +64), k columns [64 wk, +64)
  const int fl = lane & 15, gq = lane >> 4;
  const int tiles_k = g.K / 128;
  const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
  const int n0 = tn * TILE_N, k0 = tk * 128;
  const int m_begin = blockIdx.y * g.m_chunk;
  const int m_end = min(g.M, m_begin + g.m_chunk);
  float* outp = g.out + (size_t)blockIdx.y * g.N * g.K;

  // staging: thread handles chunk (tid % CH) of rows tid / CH + (NT / CH) j
  const int arow = tid / CHA, ach = tid % CHA, brow = tid >> 4, bch = tid & 15;
  const bool n_ok = n0 + ach * 8 < g.N;                 // partial last tile row of dW: zero columns
  auto load_tiles = [&](int m0, u32x4* ra, u32x4* rb) {
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      const int m = m0 + arow + (NT / CHA) * j;
      ra[j] = (u32x4){0u, 0u, 0u, 0u};
      if (m < m_end && n_ok) ra[j] = *(const u32x4*)(g.dY + (size_t)m * g.N + n0 + ach * 8);
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int m = m0 + brow + (NT / 16) * j;
      rb[j] = (u32x4){0u, 0u, 0u, 0u};
      if (m < m_end) rb[j] = *(const u32x4*)(g.X + (size_t)m * g.K + k0 + bch * 8);
    }
  };
  auto store_tiles = [&](int buf, const u32x4* ra, const u32x4* rb) {
    char* a = smem + buf * STG;
    char* b = a + IMGA;
#pragma unroll
    for (int j = 0; j < JA; ++j) *(u32x4*)(a + (arow + (NT / CHA) * j) * PITCH_A + ach * 16) = ra[j];
#pragma unroll
    for (int j = 0; j < JB; ++j) *(u32x4*)(b + (brow + (NT / 16) * j) * TN_PITCH + bch * 16) = rb[j];
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 ra[JA], rb[JB];
  load_tiles(m_begin, ra, rb);
  store_tiles(0, ra, rb);
  __syncthreads();
  int buf = 0;
  for (int m0 = m_begin; m0 < m_end; m0 += 64, buf ^= 1) {
    const bool more = m0 + 64 < m_end;
    if (more) load_tiles(m0 + 64, ra, rb);              // global loads in flight under the MFMAs
    const char* a = smem + buf * STG;
    const char* b = a + IMGA;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                    // 32 rows of m per MFMA
      // fragment of column block c (16 columns) of an image: contraction rows {32 ks + 4 gq + 0..3} and {+16}
      const int rrow = 32 * ks + 4 * gq + (fl >> 2), rcol = (fl & 3) * 8;
      u32x4 nf[4], kf[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const char* pa = a + rrow * PITCH_A + rcol + (wn * 64 + c * 16) * 2;
        const char* pb = b + rrow * TN_PITCH + rcol + (wk * 64 + c * 16) * 2;
        const u32x2 alo = tr16t(pa), ahi = tr16t(pa + 16 * PITCH_A);
        const u32x2 blo = tr16t(pb), bhi = tr16t(pb + 16 * TN_PITCH);
        nf[c] = (u32x4){alo[0], alo[1], ahi[0], ahi[1]};
        kf[c] = (u32x4){blo[0], blo[1], bhi[0], bhi[1]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<DT>(kf[j], nf[i], acc[i][j]);   // D[k = 4 gq + r][n = fl]
    }
    if (more) store_tiles(buf ^ 1, ra, rb);
    __syncthreads();
  }
  // lane holds dW[n = n0 + 64 wn + 16 i + fl][k = k0 + 64 wk + 16 j + 4 gq + {0..3}]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + wn * 64 + i * 16 + fl;
    if (n >= g.N) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + wk * 64 + j * 16 + gq * 4;
      *(float4*)(outp + (size_t)n * g.K + k) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

}  // namespace

// partial: float [splits][N][K] with splits = ceil(M / m_chunk) (m_chunk a multiple of 64); K % 128 == 0, N % 8 == 0
int gemm_tn_tile_n() {
  const char* e_ = getenv("LATTE_TN_WN");     // measurement hook: 4 = the 8-wave 256 x 128 tile
  return (e_ && atoi(e_) == 4) ? 256 : 128;
}
int launch_gemm_tn(const half_t* dY, const half_t* X, float* partial, int M, int N, int K, int m_chunk, int dtype, hipStream_t st) {
  if (K % 128 || N % 8 || m_chunk % 64 || m_chunk <= 0) return fail(LATTE_ERR_INVALID, "gemm_tn: need K % 128 == 0, N % 8 == 0, m_chunk % 64 == 0");
  TnArgs a{dY, X, partial, M, N, K, m_chunk};
  const int splits = (M + m_chunk - 1) / m_chunk;
  // 128 x 128 tile on 4 waves, two workgroups per CU.  Measured against it (training step, Latte-B/2, batch 5, same box): one
  // 256 x 128 / 8-wave workgroup per CU (a quarter fewer ds_write_b128 bytes per MFMA) 28.1 against 26.4 ms, one 128 x 256 /
  // 4-wave workgroup per CU 27.0 against 25.0 ms -- two independent workgroups de-phase (one's barrier bubble and global-load
  // wait sit under the other's MFMAs), one bigger workgroup marches in lock step.
  const int tile_n = gemm_tn_tile_n();
  dim3 grid(((N + tile_n - 1) / tile_n) * (K / 128), splits), block(tile_n * 2);
  const int lds = 2 * (64 * (tile_n == 128 ? 288 : 544) + 64 * TN_PITCH);
#define LATTE_TN_CASE(DT, WN)                                                                       \
  {                                                                                                 \
    static std::atomic<uint64_t> done{0};                                                           \
    if (int rc = ensure_dynamic_lds((const void*)gemm_tn_kernel<DT, WN>, lds, done)) return rc;     \
    hipLaunchKernelGGL((gemm_tn_kernel<DT, WN>), grid, block, lds, st, a);                          \
  }
  if (dtype == LATTE_DTYPE_BF16) { if (tile_n == 256) LATTE_TN_CASE(LATTE_DTYPE_BF16, 4) else LATTE_TN_CASE(LATTE_DTYPE_BF16, 2) }
  else if (dtype == LATTE_DTYPE_F16) { if (tile_n == 256) LATTE_TN_CASE(LATTE_DTYPE_F16, 4) else LATTE_TN_CASE(LATTE_DTYPE_F16, 2) }
  else return fail(LATTE_ERR_INVALID, "gemm_tn: unknown dtype");
#undef LATTE_TN_CASE
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

}  // namespace latte
