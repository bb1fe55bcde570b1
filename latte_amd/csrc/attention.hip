// Factored spatial / temporal self-attention of the Latte block on gfx950 MFMA.
//
// Replaces Attention.forward, attention_mode='math' (latte.py:48-77): the [S,H,L,L] score tensor
// (67 MB / sample fp32 in the reference) is never materialised.  Both kernels read Q/K/V straight
// out of the QKV GEMM's row-major [rows, 3*D] output (column order [3][heads][hd], latte.py:50) in
// the canonical [B,F,T,D] token order: a "sequence" is addressed by (base row, row stride), so the
// reference's two physical transposes per block pair (latte.py:355,368) do not exist here.
//
// Common structure (cdna_hip_programming.md "swapped QK^T"):
//   S^T = K · Q^T   -> lane holds 4 keys x 1 query: softmax row statistics need only two
//                      cross-lane steps (xor 16, 32), P stays in the lane that needs it;
//   O^T = V^T · P^T -> P (packed to half in registers, k-slot order chosen to match) is the B
//                      operand with no cross-lane traffic; lane ends with 4 consecutive d of one
//                      query -> 8-byte output stores.
//  attn_flash : any L; 64 queries (4 waves x 16) per workgroup, 64-key tiles staged in LDS,
//               online softmax.  V is staged TRANSPOSED ([d][key], key slots permuted to the MFMA
//               k-slot order) so the V^T fragment is one ds_read_b128.
//  attn_small : L <= 16 (temporal attention over frames); one wave per (sequence, head), Q/K
//               fragments straight from global memory, V through a wave-private LDS patch.
#include "common.h"

namespace latte {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <int DT>
__device__ __forceinline__ f32x4 mfma_k32(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (DT == LATTE_DTYPE_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int DT>
__device__ __forceinline__ f32x4 mfma_k16(u32x2 a, u32x2 b, f32x4 c) {
  if constexpr (DT == LATTE_DTYPE_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
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

constexpr float NEG_BIG = -1.0e30f;
constexpr int PITCH = 144;  // LDS row pitch in bytes: 9 x 16-B chunks (odd -> b128 reads spread over banks)

__device__ __forceinline__ int64_t seq_base_row(const AttnArgs& a, int seq) {
  return (int64_t)(seq / a.U) * a.sample_stride + (int64_t)(seq % a.U) * a.seq_stride;
}

// ------------------------------------------------------------------------------------------------
template <int HD, int DT>
__global__ void __launch_bounds__(256) attn_flash_kernel(AttnArgs a) {
  constexpr int KS = (HD + 31) / 32;  // k-steps of the QK^T contraction (hd padded to 32)
  constexpr int DF = (HD + 15) / 16;  // 16-wide d fragments of the PV product
  constexpr int NCH = HD / 8;         // 16-byte chunks per head row
  __shared__ __attribute__((aligned(16))) char lds[64 * PITCH + DF * 16 * PITCH];
  char* const k_lds = lds;
  char* const vt_lds = lds + 64 * PITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fl = lane & 15, g = lane >> 4;
  const int q_tiles = (a.L + 63) >> 6;
  const int qt = blockIdx.x % q_tiles;
  const int head = (blockIdx.x / q_tiles) % a.heads;
  const int seq = blockIdx.x / (q_tiles * a.heads);
  const int64_t base = seq_base_row(a, seq);
  const size_t ld = (size_t)3 * a.D;
  const half_t* qkv_h = a.qkv + (size_t)head * HD;

  // Q fragments (B operand of S^T = K·Q^T): lane = (query fl, chunk g + 4 ks)
  const int q_idx = qt * 64 + wave * 16 + fl;
  const int q_ld = min(q_idx, a.L - 1);
  u32x4 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int ch = g + 4 * ks;
    qf[ks] = (u32x4){0u, 0u, 0u, 0u};
    if (ch < NCH) qf[ks] = *(const u32x4*)(qkv_h + (size_t)(base + (int64_t)q_ld * a.row_stride) * ld + ch * 8);
  }

  f32x4 o[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;
  const float c = a.scale * 1.4426950408889634f;  // softmax in the exp2 domain

  const int kv_tiles = (a.L + 63) >> 6;
  for (int kt = 0; kt < kv_tiles; ++kt) {
    __syncthreads();  // previous tile fully consumed
    for (int id = tid; id < 64 * NCH; id += 256) {
      const int key = id / NCH, ch = id % NCH;
      const int key_ld = min(kt * 64 + key, a.L - 1);
      const half_t* rowp = qkv_h + (size_t)(base + (int64_t)key_ld * a.row_stride) * ld + ch * 8;
      const u32x4 kv = *(const u32x4*)(rowp + a.D);
      const u32x4 vv = *(const u32x4*)(rowp + 2 * a.D);
      *(u32x4*)(k_lds + key * PITCH + ch * 16) = kv;
      // transposed V image: element (d, key) at vt[d][slot(key)], slot = MFMA k-slot order:
      // within each 32-key group  slot = g*8 + half*4 + r  for  key = half*16 + g*4 + r
      const int k5 = key & 31;
      const int slot = (key & 32) + ((k5 >> 2) & 3) * 8 + (k5 >> 4) * 4 + (k5 & 3);
      half_t* vcol = (half_t*)(vt_lds + (ch * 8) * PITCH) + slot;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vcol[(2 * e) * (PITCH / 2)] = (half_t)(vv[e] & 0xffffu);
        vcol[(2 * e + 1) * (PITCH / 2)] = (half_t)(vv[e] >> 16);
      }
    }
    __syncthreads();

    // S^T[key][q] for 64 keys x 16 queries
    f32x4 st[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      st[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int ch = g + 4 * ks;
        u32x4 kf = *(const u32x4*)(k_lds + (16 * j + fl) * PITCH + ch * 16);
        if (ch >= NCH) kf = (u32x4){0u, 0u, 0u, 0u};
        st[j] = mfma_k32<DT>(kf, qf[ks], st[j]);
      }
    }
    // online softmax over the key axis (rows of S^T): in-lane over 16 values, then lanes g = 0..3
    float mx = NEG_BIG;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 64 + 16 * j + 4 * g + r;
        const float z = key < a.L ? st[j][r] * c : NEG_BIG;
        st[j][r] = z;
        mx = fmaxf(mx, z);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float ls = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(st[j][r] - m_new);
        st[j][r] = p;
        ls += p;
      }
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * alpha + ls;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < DF; ++d) o[d] *= alpha;

    // O^T += V^T · P^T ; k-slot (8g + i) <-> key 32 ks2 + (i < 4 ? 4g + i : 16 + 4g + i - 4)
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      const u32x4 pb = {pack2<DT>(st[2 * ks2][0], st[2 * ks2][1]), pack2<DT>(st[2 * ks2][2], st[2 * ks2][3]),
                        pack2<DT>(st[2 * ks2 + 1][0], st[2 * ks2 + 1][1]), pack2<DT>(st[2 * ks2 + 1][2], st[2 * ks2 + 1][3])};
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const u32x4 vf = *(const u32x4*)(vt_lds + (16 * d + fl) * PITCH + ks2 * 64 + g * 16);
        o[d] = mfma_k32<DT>(vf, pb, o[d]);
      }
    }
  }

  if (q_idx < a.L) {
    const float inv = 1.0f / l_run;
    half_t* orow = a.out + (size_t)(base + (int64_t)q_idx * a.row_stride) * a.D + head * HD;
#pragma unroll
    for (int d = 0; d < DF; ++d) {
      const int dd = 16 * d + 4 * g;
      if (dd < HD) {
        u32x2 pk = {pack2<DT>(o[d][0] * inv, o[d][1] * inv), pack2<DT>(o[d][2] * inv, o[d][3] * inv)};
        *(u32x2*)(orow + dd) = pk;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <int HD, int DT>
__global__ void __launch_bounds__(256) attn_small_kernel(AttnArgs a) {
  constexpr int KS = (HD + 31) / 32;
  constexpr int DF = (HD + 15) / 16;
  constexpr int NCH = HD / 8;
  __shared__ __attribute__((aligned(16))) char lds[4 * 17 * PITCH];  // per wave: 16 V rows (+1 row of slack)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fl = lane & 15, g = lane >> 4;
  const int items = a.num_seq * a.heads;
  int item = blockIdx.x * 4 + wave;
  const bool active = item < items;
  item = min(item, items - 1);
  const int seq = item / a.heads, head = item % a.heads;
  const int64_t base = seq_base_row(a, seq);
  const size_t ld = (size_t)3 * a.D;
  const int tok = min(fl, a.L - 1);
  const half_t* rowp = a.qkv + (size_t)(base + (int64_t)tok * a.row_stride) * ld + (size_t)head * HD;
  char* v_lds = lds + wave * 17 * PITCH;

  u32x4 qf[KS], kf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int ch = g + 4 * ks;
    qf[ks] = (u32x4){0u, 0u, 0u, 0u};
    kf[ks] = (u32x4){0u, 0u, 0u, 0u};
    if (ch < NCH) {
      qf[ks] = *(const u32x4*)(rowp + ch * 8);
      kf[ks] = *(const u32x4*)(rowp + a.D + ch * 8);
      *(u32x4*)(v_lds + fl * PITCH + ch * 16) = *(const u32x4*)(rowp + 2 * a.D + ch * 8);
    }
  }
  f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) st = mfma_k32<DT>(kf[ks], qf[ks], st);  // S^T[key = 4g + r][q = fl]

  const float c = a.scale * 1.4426950408889634f;
  float mx = NEG_BIG;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    st[r] = (4 * g + r) < a.L ? st[r] * c : NEG_BIG;
    mx = fmaxf(mx, st[r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float ls = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    st[r] = __builtin_amdgcn_exp2f(st[r] - mx);
    ls += st[r];
  }
  ls += __shfl_xor(ls, 16, 64);
  ls += __shfl_xor(ls, 32, 64);
  const u32x2 pb = {pack2<DT>(st[0], st[1]), pack2<DT>(st[2], st[3])};  // P^T[key = 4g + i][q = fl]

  __syncthreads();  // V rows of this wave are in LDS (block-wide barrier keeps it simple)
  const float inv = 1.0f / ls;
  half_t* orow = a.out + (size_t)(base + (int64_t)tok * a.row_stride) * a.D + head * HD;
#pragma unroll
  for (int d = 0; d < DF; ++d) {
    // V^T fragment: lane = (d index 16 d + fl, keys 4g .. 4g+3): column gather from the row-major patch
    const half_t* vp = (const half_t*)(v_lds + (4 * g) * PITCH) + 16 * d + fl;
    const unsigned int e0 = vp[0], e1 = vp[PITCH / 2], e2 = vp[2 * (PITCH / 2)], e3 = vp[3 * (PITCH / 2)];
    const u32x2 vf = {e0 | (e1 << 16), e2 | (e3 << 16)};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = mfma_k16<DT>(vf, pb, acc);  // O^T[d = 16 d + 4g + r][q = fl]
    const int dd = 16 * d + 4 * g;
    if (active && fl < a.L && dd < HD) {
      u32x2 pk = {pack2<DT>(acc[0] * inv, acc[1] * inv), pack2<DT>(acc[2] * inv, acc[3] * inv)};
      *(u32x2*)(orow + dd) = pk;
    }
  }
}

}  // namespace

int launch_attention(const AttnArgs& a, int dtype, hipStream_t st) {
  if (a.hd != 64 && a.hd != 72) return fail(LATTE_ERR_INVALID, "attention: head_dim must be 64 or 72");
  if (a.L <= 0) return fail(LATTE_ERR_INVALID, "attention: empty sequence");
  const bool small = a.L <= 16;
  dim3 block(256);
  dim3 grid = small ? dim3((a.num_seq * a.heads + 3) / 4) : dim3(a.num_seq * a.heads * ((a.L + 63) / 64));
#define ATTN_LAUNCH(HD, DT)                                                       \
  do {                                                                            \
    if (small)                                                                    \
      hipLaunchKernelGGL((attn_small_kernel<HD, DT>), grid, block, 0, st, a);     \
    else                                                                          \
      hipLaunchKernelGGL((attn_flash_kernel<HD, DT>), grid, block, 0, st, a);     \
  } while (0)
  if (dtype == LATTE_DTYPE_BF16) {
    if (a.hd == 64) ATTN_LAUNCH(64, LATTE_DTYPE_BF16); else ATTN_LAUNCH(72, LATTE_DTYPE_BF16);
  } else if (dtype == LATTE_DTYPE_F16) {
    if (a.hd == 64) ATTN_LAUNCH(64, LATTE_DTYPE_F16); else ATTN_LAUNCH(72, LATTE_DTYPE_F16);
  } else {
    return fail(LATTE_ERR_INVALID, "attention: unknown dtype");
  }
#undef ATTN_LAUNCH
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

}  // namespace latte
