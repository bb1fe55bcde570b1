// Device helpers shared by the MFMA kernels (gemm.hip, vae.hip): fragment types, the 16x16x32 MFMA wrapper,
// half packing, LDS DMA and the XCD-aware tile order.  gfx950 only.
#pragma once
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

// Split-operand pair of two fp32 values: hi = the half nearest to v, lo = the half nearest to (v - hi) -- hi + lo carries ~22 mantissa
// bits of v.  The K-concatenated operand [hi | lo] against [W | W] is the product a single half operand would give, without the
// activation's rounding (engine option "guided_split", engine.cpp).
template <int DT>
__device__ __forceinline__ void split2(float v0, float v1, unsigned int& hi, unsigned int& lo) {
  if constexpr (DT == LATTE_DTYPE_BF16) {
    const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
    hi = pack2<DT>((float)h0, (float)h1);
    lo = pack2<DT>(v0 - (float)h0, v1 - (float)h1);
  } else {
    const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
    hi = pack2<DT>((float)h0, (float)h1);
    lo = pack2<DT>(v0 - (float)h0, v1 - (float)h1);
  }
}

// fp8 remainder of a split operand (round 6, common.h: LO8_A_SHIFT): four values -> their nearest f16 (hi01 / hi23) and ONE word of
// four OCP e4m3 codes of (v - hi) * 2^LO8_A_SHIFT, clamped to the format's +-448 (the conversion itself does not saturate): the operand
// of the GEMMs' block-scaled correction pass (gemm_pw.hip), whose constant E8M0 scale multiplies the 2^-LO8_A_SHIFT back.
__device__ __forceinline__ unsigned int split8_f16(float v0, float v1, float v2, float v3, unsigned int& hi01, unsigned int& hi23) {
  const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1, h2 = (_Float16)v2, h3 = (_Float16)v3;
  hi01 = pack2<LATTE_DTYPE_F16>((float)h0, (float)h1);
  hi23 = pack2<LATTE_DTYPE_F16>((float)h2, (float)h3);
  constexpr float S = (float)(1 << LO8_A_SHIFT);
  auto cl = [](float r) { return __builtin_fminf(__builtin_fmaxf(r, -448.f), 448.f); };
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(cl((v0 - (float)h0) * S), cl((v1 - (float)h1) * S), 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(cl((v2 - (float)h2) * S), cl((v3 - (float)h3) * S), w, true);
  return (unsigned int)w;
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

}  // namespace
}  // namespace latte
