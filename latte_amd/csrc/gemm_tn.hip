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
#include <algorithm>
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
  float* colsum;      // gemm_tn8_kernel<DT, true>: [splits][N] column sums of dY over the split's rows (the bias gradient's partials)
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


// ------------------------------------------------------------------------------------------------
// gemm_tn8_kernel (round 3): the same product on the machinery of the forward GEMMs -- 256 (n) x 256 (k) output tile, 8 waves as two
// groups (n halves) x 4 waves (64 k columns each, wave tile 128 x 64 = 32 accumulators), 64 rows of the contraction per stage, two
// stages, the ping-pong schedule of gemm_pp_kernel (gemm.hip), and the tiles staged by LDS DMA instead of through registers (the
// 128 x 128 kernel above is bound by its ds_write_b128 tile stores: 550 TF/s).  A tile is kept as four half-images [64 m][128
// columns] (rows of 256 B = the 64 banks exactly): dY columns of group 0 / group 1, X columns 0-127 / 128-255.  The DMA image is
// lane-linear (one instruction = 4 rows x 256 B), so the swizzle that makes the transpose reads conflict-free lives in the per-lane
// SOURCE address: 32-byte block b of row m is stored at block b ^ (m & 7) -- the 8 rows x 32 B that a 32-lane service group of a
// transpose read touches then tile the 64 banks.  For a lane the block swizzle of instruction i differs from instruction 0's by
// 4 (i & 1): its source offset is voffset0 ^ ((i & 1) << 7) (row pitches are multiples of 256 B: N % 128 == 0, K % 128 == 0).
// Fragments as above (two ds_read_b64_tr_b16 per 16 x 32 fragment, the same slot permutation for both operands).
// Group g's waves stage their own dY half (4 instructions per wave and stage), group 0's also the X tile (8): the DMA split and
// every hand-off are those of gemm_pp_kernel.  Requires M % 64 == 0 (no ragged contraction rows), N % 128 == 0, K % 128 == 0;
// a half-image beyond N / K is not staged and its outputs are not stored.
// Transpose read as inline assembly for the DMA-staged kernel: hipcc puts `s_waitcnt vmcnt(0)` in front of the BUILTIN whenever an
// LDS DMA is in flight (it treats the pending DMA as a store the read may alias), which drained the next tiles' DMA at the top of
// every K step (5.4 k clocks per step instead of ~3 k).  The asm form is invisible to that bookkeeping; its completion is waited
// for by the kernel's own `s_waitcnt lgkmcnt(0)` + barrier before the MFMAs (cdna_hip_programming.md section 5.7, form (iii)).
template <int OFF>
__device__ __forceinline__ u32x2 tr16_asm(unsigned lds_addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "i"(OFF));
  return v;
}
typedef __attribute__((address_space(3))) void lds_void_tn;
__device__ __forceinline__ void dma16tn(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_tn*)lds_wave_base, 16, voff, soff, 0, 0);
}

// COLSUM (round 6b): the bias gradient db[n] = sum_m dY[m, n] rides on the launch -- the dY fragments of a K step are in registers
// anyway (lane (fl, gq) holds column 16 i + fl, rows 32 ks + 4 gq + {0..3, 16..19}), so wave 0 of each group in the k-tile-0
// workgroups adds its 8 halves per fragment with four v_dot2c_f32_f16 against (1, 1) (fp32 accumulate) between the MFMAs; 8 extra
// registers, no extra memory traffic.  The separate column-sum kernel read dY a second time: 0.69 ms per Latte-B/2 step.
template <int DT>
__device__ __forceinline__ float dot2_ones(unsigned int w, float acc) {
  if constexpr (DT == LATTE_DTYPE_BF16) {
    typedef __attribute__((ext_vector_type(2))) __bf16 b2t;
    const b2t one = {(__bf16)1.0f, (__bf16)1.0f};
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2t, w), one, acc, false);
  } else {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2t;
    const h2t one = {(_Float16)1.0f, (_Float16)1.0f};
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2t, w), one, acc, false);
  }
}
template <int DT, bool COLSUM>
__global__ void __launch_bounds__(512) gemm_tn8_kernel(TnArgs g) {
  constexpr int HALF = 64 * 256;             // one half-image: 64 rows x 256 B
  constexpr int STG = 4 * HALF;              // dY half 0 | dY half 1 | X half 0 | X half 1
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, wk = wave & 3;
  const int fl = lane & 15, gq = lane >> 4;
  // Workgroup -> (split, tile).  N and K are a few tiles only, so every panel of a split (dY columns of an n tile, X columns of
  // a k tile) is read by several tiles: un-mapped, the launch moved 566 MB for the 126 MB of a Latte-B/2 qkv weight gradient and
  // ran at HBM speed (6.1 TB/s).  Workgroups are dispatched round-robin over the 8 XCDs; give every XCD a contiguous range of
  // the (split-major) work list, so that the tiles of one split run on ONE XCD and share their panels in its L2.
  const int tiles_k = (g.K + 255) / 256;
  const int tiles = tiles_k * ((g.N + 255) / 256);
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int split = wg / tiles, tile = wg - split * tiles;
  const int tn = tile / tiles_k, tk = tile % tiles_k;
  const int n0 = tn * 256, k0 = tk * 256;
  const int m_begin = split * g.m_chunk;
  const int m_end = min(g.M, m_begin + g.m_chunk);
  const int nk = (m_end - m_begin) / 64;
  float* outp = g.out + (size_t)split * g.N * g.K;
  // which halves exist (N, K multiples of 128: a tile may own only its first half)
  const bool y_ok = n0 + grp * 128 < g.N;                       // this group's dY half / output rows
  const bool x1_ok = k0 + 128 < g.K;                            // X half 1
  const bool k_ok = k0 + (wk >> 1) * 128 < g.K;                 // this wave's output columns

  const unsigned ldy = (unsigned)g.N * 2u, ldx = (unsigned)g.K * 2u;
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)g.dY, 0, (unsigned)g.M * ldy, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)g.X, 0, (unsigned)g.M * ldx, 0x00020000);
  // DMA lane map: row (lane >> 4) of a 4-row group, physical 16-byte chunk (lane & 15) = physical 32-byte block (lane & 15) >> 1;
  // instruction i covers rows 4 i .. 4 i + 3, so (row & 7) = 4 (i & 1) + (lane >> 4): logical block = physical ^ that
  const unsigned col0 = (unsigned)(((((lane & 15) >> 1) ^ (lane >> 4)) << 5) + ((lane & 1) << 4));
  const unsigned voy = (unsigned)(lane >> 4) * ldy + col0, vox = (unsigned)(lane >> 4) * ldx + col0;
  // a wave's instructions of a half-image: i = wk + 4 j (j = 0..3): i & 1 = wk & 1 for every j
  const unsigned voy_w = voy ^ ((unsigned)(wk & 1) << 7), vox_w = vox ^ ((unsigned)(wk & 1) << 7);
  auto dma_y = [&](int kt, int stg) {       // own dY half of K tile kt
    if (!y_ok) return;
    char* dst = smem + stg * STG + grp * HALF + wk * 1024;
    const unsigned so = (unsigned)(m_begin + kt * 64 + wk * 4) * ldy + (unsigned)(n0 + grp * 128) * 2u;
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16tn(rsY, dst + j * 4096, voy_w, so + (unsigned)(16 * j) * ldy);
  };
  auto dma_x = [&](int kt, int stg) {       // both X halves of K tile kt (group 0's waves)
    char* dst = smem + stg * STG + 2 * HALF + wk * 1024;
    const unsigned so = (unsigned)(m_begin + kt * 64 + wk * 4) * ldx + (unsigned)k0 * 2u;
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16tn(rsX, dst + j * 4096, vox_w, so + (unsigned)(16 * j) * ldx);
    if (x1_ok) {
#pragma unroll
      for (int j = 0; j < 4; ++j) dma16tn(rsX, dst + HALF + j * 4096, vox_w, so + 256u + (unsigned)(16 * j) * ldx);
    }
  };

  // fragment read addresses: rows 32 ks + 4 gq + (fl >> 2) (+ 16), 8 bytes at column block c, in-block offset (fl & 3) * 8;
  // (row & 7) = 4 (gq & 1) + (fl >> 2) for every such row -> physical block c ^ that
  const int sblk = ((gq & 1) << 2) + (fl >> 2);
  const int lane_base = (4 * gq + (fl >> 2)) * 256 + (fl & 3) * 8 + (sblk << 5);
  const int yb = grp * HALF, xb = 2 * HALF + (wk >> 1) * HALF, xc0 = (wk & 1) * 4;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float cs[COLSUM ? 8 : 1];
  const bool do_cs = COLSUM && wk == 0 && tk == 0 && y_ok;   // wave-uniform
  if constexpr (COLSUM) {
#pragma unroll
    for (int i = 0; i < 8; ++i) cs[i] = 0.f;
  }

  dma_y(0, 0);
  if (grp == 0) dma_x(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (nk > 1) {
    dma_y(1, 1);
    if (grp == 0) dma_x(1, 1);
  }
  if (grp == 1) __builtin_amdgcn_s_barrier();   // stagger the two groups by one segment
  const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned sb = smem_base + (unsigned)((kt & 1) * STG);
    u32x4 nf[2][8], kf[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned pb = sb + (unsigned)(xb + (lane_base ^ ((xc0 + c) << 5)));
      const u32x2 l0 = tr16_asm<0>(pb), h0 = tr16_asm<4096>(pb), l1 = tr16_asm<8192>(pb), h1 = tr16_asm<12288>(pb);
      kf[0][c] = (u32x4){l0[0], l0[1], h0[0], h0[1]};
      kf[1][c] = (u32x4){l1[0], l1[1], h1[0], h1[1]};
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const unsigned pa = sb + (unsigned)(yb + (lane_base ^ (c << 5)));
      const u32x2 l0 = tr16_asm<0>(pa), h0 = tr16_asm<4096>(pa), l1 = tr16_asm<8192>(pa), h1 = tr16_asm<12288>(pa);
      nf[0][c] = (u32x4){l0[0], l0[1], h0[0], h0[1]};
      nf[1][c] = (u32x4){l1[0], l1[1], h1[0], h1[1]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);   // nothing that consumes the fragments may move above the wait (guide rule 18)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<DT>(kf[ks][j], nf[ks][i], acc[i][j]);   // D[k = 4 gq + r][n = fl]
    if constexpr (COLSUM) {
      if (do_cs) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float a = cs[i];
#pragma unroll
            for (int w = 0; w < 4; ++w) a = dot2_ones<DT>(nf[ks][i][w], a);
            cs[i] = a;
          }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) {
      dma_y(kt + 2, kt & 1);
      if (grp == 0) dma_x(kt + 2, kt & 1);
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();   // balance group 1's extra barrier
  if constexpr (COLSUM) {
    if (do_cs) {   // the four row quarters (gq) of a column, fixed order
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float v = cs[i];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (gq == 0) g.colsum[(size_t)split * g.N + n0 + grp * 128 + i * 16 + fl] = v;
      }
    }
  }
  // lane holds dW[n = n0 + 128 grp + 16 i + fl][k = k0 + 64 wk + 16 j + 4 gq + {0..3}]
  if (y_ok && k_ok) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = n0 + grp * 128 + i * 16 + fl;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + wk * 64 + j * 16 + gq * 4;
        *(float4*)(outp + (size_t)n * g.K + k) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
  }
}

}  // namespace

// The 8-wave LDS-DMA kernel takes a shape when the contraction has no ragged 64-row step and both output extents are whole
// 128-column half-tiles (every linear of the Latte family); latte_debug_set_choice("tn_kernel", 4) (A/B tests) forces the 4-wave kernel.
bool gemm_tn8_ok(int M, int N, int K) {
  if (debug_choice(DBG_TN_KERNEL) == 4) return false;
  return M % 64 == 0 && N % 128 == 0 && K % 128 == 0 && (uint64_t)M * N * 2 < (1ull << 32) && (uint64_t)M * K * 2 < (1ull << 32);
}
// split of the contraction: -> number of splits, *chunk = rows per split (a multiple of 64).  8-wave kernel: one workgroup per
// CU (128 KB of LDS), so about 256 workgroups; 4-wave kernel: about four workgroups per CU in flight.
int gemm_tn_plan(int M, int N, int K, int* chunk) {
  int splits;
  if (gemm_tn8_ok(M, N, K)) {
    const int tiles = ((N + 255) / 256) * ((K + 255) / 256);
    splits = std::max(1, std::min(M / 64, 256 / tiles));
  } else {
    const int tn = gemm_tn_tile_n();
    const int tiles = ((N + tn - 1) / tn) * (K / 128);
    splits = std::max(1, std::min(M / 64, ((tn == 256 ? 512 : 768) + tiles - 1) / tiles));
  }
  *chunk = ((M + splits - 1) / splits + 63) / 64 * 64;
  return (M + *chunk - 1) / *chunk;
}

// partial: float [splits][N][K] with splits = ceil(M / m_chunk) (m_chunk a multiple of 64); K % 128 == 0, N % 8 == 0
int gemm_tn_tile_n() {
  return debug_choice(DBG_TN_WN) == 4 ? 256 : 128;     // latte_debug_set_choice("tn_wn", 4): the 8-wave 256 x 128 tile (A/B)
}
// colsum_partial (optional, only where gemm_tn8_ok(M, N, K)): float [splits][N] receives the column sums of dY per split
int launch_gemm_tn(const half_t* dY, const half_t* X, float* partial, int M, int N, int K, int m_chunk, int dtype, hipStream_t st,
                   float* colsum_partial) {
  if (K % 128 || N % 8 || m_chunk % 64 || m_chunk <= 0) return fail(LATTE_ERR_INVALID, "gemm_tn: need K % 128 == 0, N % 8 == 0, m_chunk % 64 == 0");
  TnArgs a{dY, X, partial, M, N, K, m_chunk, colsum_partial};
  const int splits = (M + m_chunk - 1) / m_chunk;
  if (colsum_partial && !gemm_tn8_ok(M, N, K)) return fail(LATTE_ERR_INVALID, "gemm_tn: column sums ride on the 8-wave kernel only");
  if (gemm_tn8_ok(M, N, K)) {
    constexpr int LDS8 = 2 * 4 * 64 * 256;
    dim3 grid8(((N + 255) / 256) * ((K + 255) / 256) * splits), block8(512);   // 1-D: the kernel maps workgroups to (split, tile)
#define LATTE_TN8_CASE(DT, CS)                                                                      \
  {                                                                                                 \
    static std::atomic<uint64_t> done{0};                                                           \
    if (int rc = ensure_dynamic_lds((const void*)gemm_tn8_kernel<DT, CS>, LDS8, done)) return rc;   \
    hipLaunchKernelGGL((gemm_tn8_kernel<DT, CS>), grid8, block8, LDS8, st, a);                      \
  }
    if (dtype == LATTE_DTYPE_BF16) { if (colsum_partial) LATTE_TN8_CASE(LATTE_DTYPE_BF16, true) else LATTE_TN8_CASE(LATTE_DTYPE_BF16, false) }
    else if (dtype == LATTE_DTYPE_F16) { if (colsum_partial) LATTE_TN8_CASE(LATTE_DTYPE_F16, true) else LATTE_TN8_CASE(LATTE_DTYPE_F16, false) }
    else return fail(LATTE_ERR_INVALID, "gemm_tn: unknown dtype");
#undef LATTE_TN8_CASE
    LATTE_HIP(hipGetLastError());
    return LATTE_OK;
  }
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
