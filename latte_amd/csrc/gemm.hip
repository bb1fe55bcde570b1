// MFMA GEMM for the Latte linears:  C[M,N] = A[M,K] · W[N,K]^T  with fused epilogues.
//
// Replaces the nn.Linear calls of the reference block (latte.py:43-45,50,75 qkv/proj and the timm
// Mlp fc1/fc2 at latte.py:171) together with the elementwise ops that follow them (bias, GELU-tanh
// latte.py:170, gate * y + residual latte.py:179-180).  98 % of the denoiser's FLOPs run here.
//
// gfx950 design (guide: cdna_hip_programming.md §5, T1/T2/T3):
//  * both operands are K-contiguous, so every MFMA fragment is one ds_read_b128;
//  * tiles are staged HBM/L2 -> LDS by global_load_lds (16 B/lane, no VGPR round trip);
//    the LDS image is lane-linear, so the bank swizzle is applied to the per-lane SOURCE address
//    (chunk ^= (row>>1)&7 inside each 128-B row) and mirrored on the ds_read side;
//  * double-buffered LDS, one barrier per 64-deep K step, next tile's DMA in flight under the MFMAs;
//  * operands are passed to the MFMA swapped (W fragment as A, activation fragment as B), so each
//    lane ends up with 4 CONSECUTIVE output columns of one row -> 8-byte half / 16-byte fp32 stores
//    and float4 bias / gate loads in the epilogue;
//  * XCD-aware, grouped block->tile mapping so that co-resident tiles of one XCD share A/W panels in
//    that XCD's private L2.
#include "common.h"

namespace latte {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <int DT>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (DT == LATTE_DTYPE_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int DT>
__device__ __forceinline__ unsigned int pack2(float lo, float hi) {
  if constexpr (DT == LATTE_DTYPE_BF16) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned int, v);
  } else {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    f16x2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(unsigned int, v);
  }
}

// GELU(tanh approximation) = x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3)   (latte.py:170)
__device__ __forceinline__ float gelu_tanh(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return x / (1.0f + __expf(-2.0f * u));
}

__device__ __forceinline__ void glds16(const half_t* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware bijective remap (blocks are dispatched round-robin over the 8 XCDs): give every XCD a
// contiguous chunk of the tile sequence, then walk tiles in groups of GROUP_M tile-rows.
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int& tm, int& tn) {
  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n;
  const int group = wg / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int in_group = wg - group * per_group;
  tm = first_m + in_group % gsz;
  tn = in_group / gsz;
}

template <int BM, int BN, int WGM, int WGN, int EPI, int DT>
__global__ void __launch_bounds__(WGM* WGN * 64) gemm_kernel(GemmArgs g) {
  constexpr int NW = WGM * WGN;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int FM = WTM / 16, FN = WTN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "staging split");
  static_assert((NW * 4) % 8 == 0, "source swizzle must not depend on the instruction index");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  int tm, tn;
  tile_coords((g.M + BM - 1) / BM, g.N / BN, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int K = g.K;

  // ---- staging addresses: lane i of a wave-instruction fills LDS bytes [16 i, 16 i + 16) of an
  // 8-row group; it must therefore FETCH the chunk that belongs at that (row, chunk-position).
  const int lrow = lane >> 3, cpos = lane & 7;
  const int srow = wave * 8 + lrow;                       // tile row handled by instruction 0
  const int schunk = cpos ^ ((srow >> 1) & 7);            // same for every instruction (NW*8 % 16 == 0)
  const half_t* a_src = g.A + (size_t)(m0 + srow) * K + schunk * 8;
  const half_t* b_src = g.W + (size_t)(n0 + srow) * K + schunk * 8;
  const size_t jstride = (size_t)NW * 8 * K;

  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * STAGE + wave * 1024;
    char* sB = sA + A_BYTES;
    const int koff = kt * 64;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) glds16(a_src + j * jstride + koff, sA + j * NW * 1024);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) glds16(b_src + j * jstride + koff, sB + j * NW * 1024);
  };

  // ---- fragment read offsets (row = 16*f + (lane&15); chunk = (lane>>4) + 4*ks, swizzled)
  const int frow = lane & 15;
  const int sw = (lane >> 1) & 7;
  const int chunk0 = ((lane >> 4) ^ sw) * 16;
  const int a_off = (wm * WTM + frow) * 128 + chunk0;
  const int b_off = A_BYTES + (wn * WTN + frow) * 128 + chunk0;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = K / 64;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed (own DMA drained, then everybody's via the barrier); the barrier also
    // proves every wave finished reading the other buffer (tile kt-1), so it may be overwritten.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* sbuf = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *(const u32x4*)(sbuf + ((a_off + i * 2048) ^ (ks << 6)));
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = *(const u32x4*)(sbuf + ((b_off + j * 2048) ^ (ks << 6)));
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16<DT>(bf[j], af[i], acc[i][j]);
    }
  }

  // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + (lane>>4)*4 + {0,1,2,3}]
  const int ncol = n0 + wn * WTN + (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * WTM + i * 16 + frow;
    if (m >= g.M) continue;
    const float* gate_row = nullptr;
    if constexpr (EPI == EPI_GATE_RES_F32) gate_row = g.gate + (size_t)(m / g.rows_per_sample) * g.gate_stride;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = ncol + j * 16;
      const float4 b4 = *(const float4*)(g.bias + n);
      float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
      const size_t o = (size_t)m * g.N + n;
      if constexpr (EPI == EPI_BIAS_H16 || EPI == EPI_BIAS_GELU_H16) {
        if constexpr (EPI == EPI_BIAS_GELU_H16) {
          v0 = gelu_tanh(v0); v1 = gelu_tanh(v1); v2 = gelu_tanh(v2); v3 = gelu_tanh(v3);
        }
        u32x2 p = {pack2<DT>(v0, v1), pack2<DT>(v2, v3)};
        *(u32x2*)((half_t*)g.out + o) = p;
      } else if constexpr (EPI == EPI_GATE_RES_F32) {
        const float4 g4 = *(const float4*)(gate_row + n);
        float4* dst = (float4*)((float*)g.out + o);
        float4 r = *dst;
        r.x += g4.x * v0; r.y += g4.y * v1; r.z += g4.z * v2; r.w += g4.w * v3;
        *dst = r;
      } else {
        *(float4*)((float*)g.out + o) = make_float4(v0, v1, v2, v3);
      }
    }
  }
}

template <int BM, int BN, int WGM, int WGN, int DT>
int launch_cfg(const GemmArgs& a, int epi, hipStream_t st) {
  constexpr int LDS = 2 * (BM + BN) * 128;
  const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
  dim3 grid(tiles), block(WGM * WGN * 64);
#define LATTE_GEMM_CASE(E)                                                                           \
  case E: {                                                                                          \
    auto kern = gemm_kernel<BM, BN, WGM, WGN, E, DT>;                                                \
    static bool attr_done = false;                                                                   \
    if (!attr_done) {                                                                                \
      LATTE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
      attr_done = true;                                                                              \
    }                                                                                                \
    hipLaunchKernelGGL(kern, grid, block, LDS, st, a);                                               \
    break;                                                                                           \
  }
  switch (epi) {
    LATTE_GEMM_CASE(EPI_BIAS_H16)
    LATTE_GEMM_CASE(EPI_BIAS_GELU_H16)
    LATTE_GEMM_CASE(EPI_GATE_RES_F32)
    LATTE_GEMM_CASE(EPI_BIAS_F32)
    default:
      return fail(LATTE_ERR_INVALID, "gemm: unknown epilogue");
  }
#undef LATTE_GEMM_CASE
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

template <int DT>
int launch_dt(const GemmArgs& a, int epi, int variant, hipStream_t st) {
  if (variant == 0) variant = 1;
  switch (variant) {
    case 1: return launch_cfg<128, 128, 2, 2, DT>(a, epi, st);
    case 2: return launch_cfg<256, 128, 4, 2, DT>(a, epi, st);
    case 3: return launch_cfg<256, 256, 2, 4, DT>(a, epi, st);
    default: return fail(LATTE_ERR_INVALID, "gemm: unknown tile variant");
  }
}

}  // namespace

int gemm_tile_m(int variant) { return (variant == 2 || variant == 3) ? 256 : 128; }

int launch_gemm(const GemmArgs& a, int epi, int dtype, int variant, hipStream_t st) {
  const int bn = (variant == 3) ? 256 : 128;
  if (a.K % 64 != 0 || a.N % bn != 0 || a.M <= 0)
    return fail(LATTE_ERR_INVALID, "gemm: shape not tileable (need K % 64 == 0, N % tileN == 0)");
  if (dtype == LATTE_DTYPE_BF16) return launch_dt<LATTE_DTYPE_BF16>(a, epi, variant, st);
  if (dtype == LATTE_DTYPE_F16) return launch_dt<LATTE_DTYPE_F16>(a, epi, variant, st);
  return fail(LATTE_ERR_INVALID, "gemm: unknown dtype");
}

}  // namespace latte
