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
#include <cstdlib>
#include <type_traits>

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
// hardware transpose read: lane i of a 16-lane group supplies the address of 4 d-values of key (i >> 2) of a ROW-MAJOR image
// and receives the 4 keys of d-column i (see attn_full_kernel below)
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short i16v4;
template <int DT>
__device__ __forceinline__ u32x2 lds_tr16(const char* p) {   // 16-bit elements: the bit pattern is dtype-agnostic
  i16v4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16v4*)p);
  return __builtin_bit_cast(u32x2, v);
}

// CROSS = true (LatteT2V attn2): queries from a [rows, q_ld] buffer, Lk keys / values per SAMPLE from a.kv ([K | V], 2D
// columns), an optional additive score bias per (sample, key); everything else is the same kernel.
// (mfma_util.h: split8_f16, restated here: this file keeps its own fragment helpers) four values -> their nearest f16 and one word of four
// OCP e4m3 codes of (v - hi) * 2^LO8_A_SHIFT, clamped to +-448
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

// One output piece of an attention kernel: four consecutive head-dim values of a token -> 8 bytes of half at `dst` (a.out based), and,
// with LO8 (f16; guided calls, engine option guided_split bit 2), their fp8 remainder word at the same element offset of a.out8
// (mfma_util.h: split8_f16) -- the correction operand of the out-projection's GEMM when the fused QKV + attention kernel does not take
// the shape (round 6; round 5 dropped the split operand silently on this path).
template <int DT, bool LO8>
__device__ __forceinline__ void store_out4(const AttnArgs& a, half_t* dst, float v0, float v1, float v2, float v3) {
  if constexpr (LO8) {
    unsigned int h0_, h1_;
    const unsigned int l8 = split8_f16(v0, v1, v2, v3, h0_, h1_);
    *(u32x2*)dst = (u32x2){h0_, h1_};
    *(unsigned int*)(a.out8 + (dst - a.out)) = l8;
  } else {
    *(u32x2*)dst = (u32x2){pack2<DT>(v0, v1), pack2<DT>(v2, v3)};
  }
}

template <int HD, int DT, bool CROSS = false, bool LO8 = false>
__global__ void __launch_bounds__(256) attn_flash_kernel(AttnArgs a) {
  constexpr int KS = (HD + 31) / 32;  // k-steps of the QK^T contraction (hd padded to 32)
  constexpr int DF = (HD + 15) / 16;  // 16-wide d fragments of the PV product
  constexpr int NCH = HD / 8;         // 16-byte chunks per head row
  constexpr int VP = 160;             // row pitch of the row-major V image: conflict-free for the transpose reads
  __shared__ __attribute__((aligned(16))) char lds[64 * PITCH + 64 * VP];
  char* const k_lds = lds;
  char* const v_lds = lds + 64 * PITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fl = lane & 15, g = lane >> 4;
  const int q_tiles = (a.L + 63) >> 6;
  const int qt = blockIdx.x % q_tiles;
  const int head = (blockIdx.x / q_tiles) % a.heads;
  const int seq = blockIdx.x / (q_tiles * a.heads);
  const int64_t base = seq_base_row(a, seq);
  const size_t ld = CROSS ? (size_t)a.q_ld : (size_t)3 * a.D;
  const half_t* qkv_h = a.qkv + (size_t)head * HD;
  const int NK = CROSS ? a.Lk : a.L;                       // keys per sequence
  const int smp = seq / a.U;                               // CROSS: the sample whose text tokens are the keys
  const half_t* kv_h = CROSS ? a.kv + ((size_t)smp * a.Lk) * (2 * (size_t)a.D) + (size_t)head * HD : nullptr;

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

  const int kv_tiles = (NK + 63) >> 6;
  for (int kt = 0; kt < kv_tiles; ++kt) {
    __syncthreads();  // previous tile fully consumed
    for (int id = tid; id < 64 * NCH; id += 256) {
      const int key = id / NCH, ch = id % NCH;
      const int key_ld = min(kt * 64 + key, NK - 1);
      u32x4 kv, vv;
      if constexpr (CROSS) {
        const half_t* rowp = kv_h + (size_t)key_ld * (2 * (size_t)a.D) + ch * 8;
        kv = *(const u32x4*)rowp;
        vv = *(const u32x4*)(rowp + a.D);
      } else {
        const half_t* rowp = qkv_h + (size_t)(base + (int64_t)key_ld * a.row_stride) * ld + ch * 8;
        kv = *(const u32x4*)(rowp + a.D);
        vv = *(const u32x4*)(rowp + 2 * a.D);
      }
      *(u32x4*)(k_lds + key * PITCH + ch * 16) = kv;
      *(u32x4*)(v_lds + key * VP + ch * 16) = vv;     // row-major: V^T fragments come out of ds_read_b64_tr_b16
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
        float z = key < NK ? st[j][r] * c : NEG_BIG;
        if constexpr (CROSS) {
          if (a.kbias != nullptr && key < NK) z += a.kbias[(size_t)smp * a.Lk + key] * 1.4426950408889634f;
        }
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
        // pad d-columns (>= HD) read finite neighbouring data and only produce unused O rows
        const char* vb = v_lds + (32 * ks2 + 4 * g + (fl >> 2)) * VP + (fl & 3) * 8 + d * 32;
        const u32x2 lo = lds_tr16<DT>(vb);
        const u32x2 hi = lds_tr16<DT>(vb + 16 * VP);
        o[d] = mfma_k32<DT>((u32x4){lo[0], lo[1], hi[0], hi[1]}, pb, o[d]);
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
        store_out4<DT, LO8>(a, orow + dd, o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// attn_full : 128 < L <= 256 (the spatial attention of every 256-pixel Latte config: T = 256 tokens per frame).
// One workgroup per (sequence, head); ALL keys and values of that head are staged once, row-major, in LDS
// (2 x 40 KB: two workgroups per CU), so there is no key-tile loop, no barrier after the staging and no online
// rescale: a wave computes the complete S^T for 32 of its 64 queries (2 x 16 MFMA column groups sharing every K
// fragment read), takes an exact softmax over the 256 scores it holds in registers, and feeds P straight into
// the PV MFMAs.  V^T fragments come from the row-major V image through ds_read_b64_tr_b16 (the hardware
// transpose read: lane i of a 16-lane group supplies the address of 4 d-values of key (i >> 2) and receives the
// 4 keys of d-column i), so V is never transposed in memory.
// Row pitch 160 B (10 chunks) for both images: conflict-free for the b128 K reads (row 10r + chunk distinct mod 16
// inside every 16-lane service group) and for the tr reads (8 rows x 32 B tile the 64 banks).


template <int HD, int DT, bool LO8 = false>
__global__ void __launch_bounds__(256, 2) attn_full_kernel(AttnArgs a) {
  constexpr int KS = (HD + 31) / 32;
  constexpr int DF = (HD + 15) / 16;
  constexpr int NCH = HD / 8;
  constexpr int RP = 160;            // row pitch of the K and V images (bytes)
  constexpr int NKT = 16;            // 16-key tiles
  extern __shared__ __attribute__((aligned(16))) char smem_attn[];
  char* const k_lds = smem_attn;
  char* const v_lds = smem_attn + 256 * RP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fl = lane & 15, g = lane >> 4;
  // Block -> (sequence, head): blocks are dispatched round-robin over the 8 XCDs, and the 16 head slices of one
  // token row share 128-byte lines (144 B per head at hd = 72), so all heads of a sequence are kept on ONE XCD
  // (blocks b, b + 8, b + 16, ... walk the heads): the straddling lines are fetched from HBM once, not per XCD.
  int head, seq;
  if ((a.num_seq & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    head = slot % a.heads;
    seq = (slot / a.heads) * 8 + xcd;
  } else {
    head = blockIdx.x % a.heads;
    seq = blockIdx.x / a.heads;
  }
  const int64_t base = seq_base_row(a, seq);
  const size_t ld = (size_t)3 * a.D;
  const half_t* qkv_h = a.qkv + (size_t)head * HD;

  // ---- stage K and V by LDS DMA: 2 x 2560 16-byte chunks, 20 back-to-back instructions per wave, ONE wait.
  // The image is lane-linear (chunk index = 64 * instruction + lane -> row = idx / 10, chunk = idx % 10), the source
  // address is per lane.  Pad chunk 9 (d 72..79 at hd = 72; chunks 8, 9 at hd = 64) and rows >= L re-read a valid
  // chunk: they only ever meet a zero Q chunk, an unused O row or P = 0, and the data is finite.
  if (a.variant != 3) {   // (variant 2 / 3: measurement ablations -- staging only / compute only)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int inst = wave_u * 10 + j;                 // 40 instructions per image, 10 per wave
      const int idx = inst * 64 + lane;                 // chunk index 0..2559
      const int key = idx / 10, ch = idx - key * 10;
      const int key_ld = min(key, a.L - 1), ch_ld = min(ch, NCH - 1);
      const half_t* rowp = qkv_h + (size_t)(base + (int64_t)key_ld * a.row_stride) * ld + ch_ld * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rowp + a.D),
                                       (__attribute__((address_space(3))) void*)(k_lds + inst * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rowp + 2 * a.D),
                                       (__attribute__((address_space(3))) void*)(v_lds + inst * 1024), 16, 0, 0);
    }
  }
  // Q fragments of the two 16-query groups of a pass (B operand of S^T = K Q^T): lane = (query fl, chunk g + 4 ks).
  // Pass 0's are fetched under the staging DMA, pass 1's under pass 0's softmax / PV phase.
  u32x4 qf[2][KS];
  auto load_q = [&](int pass_, u32x4 (&dst)[2][KS]) {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      const int q_ld = min(wave * 64 + pass_ * 32 + gq * 16 + fl, a.L - 1);
      const half_t* qrow = qkv_h + (size_t)(base + (int64_t)q_ld * a.row_stride) * ld;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int ch = g + 4 * ks;
        dst[gq][ks] = (u32x4){0u, 0u, 0u, 0u};
        if (ch < NCH) dst[gq][ks] = *(const u32x4*)(qrow + ch * 8);
      }
    }
  };
  load_q(0, qf);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (a.variant == 2) {
    if (tid == 0) a.out[(size_t)base * a.D + head * HD] = *(const half_t*)(k_lds + 2 * (seq & 63));
    return;
  }

  const float c = a.scale * 1.4426950408889634f;  // softmax in the exp2 domain
  const char* kbase = k_lds + fl * RP + g * 16;
  // tr-read address of this lane: key row (fl >> 2) of a 4-key block, d-bytes (fl & 3) * 8 of a 16-d block
  const char* vbase = v_lds + (4 * g + (fl >> 2)) * RP + (fl & 3) * 8;

#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int q0 = wave * 64 + pass * 32;
    if (q0 >= a.L) break;
    // S^T[key][q] for all 256 keys x 2 x 16 queries.  K fragments are software-pipelined two key tiles ahead
    // through four rotating register sets; sched_barrier pins the order (the compiler's own schedule waited
    // lgkmcnt(0) after every few MFMAs and exposed the LDS latency).
    f32x4 st[2][NKT];
    u32x4 kf[4][KS];
    auto load_k = [&](int kt, u32x4 (&dst)[KS]) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        // chunks >= NCH (pad chunk 9, and 10 / 11 = the next row's first chunks) hold finite data and only ever
        // multiply the zero Q chunks: no select here, it would force a wait on the load just issued
        dst[ks] = *(const u32x4*)(kbase + kt * 16 * RP + ks * 64);
      }
    };
    load_k(0, kf[0]);
    load_k(1, kf[1]);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt + 2 < NKT) load_k(kt + 2, kf[(kt + 2) & 3]);
      __builtin_amdgcn_sched_barrier(0);
      st[0][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      st[1][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        st[0][kt] = mfma_k32<DT>(kf[kt & 3][ks], qf[0][ks], st[0][kt]);
        st[1][kt] = mfma_k32<DT>(kf[kt & 3][ks], qf[1][ks], st[1][kt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    u32x4 qn[2][KS];
    if (pass == 0) load_q(1, qn);
    // exact softmax over the key axis: in-lane over 64 values, then the 4 lanes g = 0..3 of a query.
    // max on the raw scores (c > 0), p = exp2(s * c - max * c): one max, one fma, one exp2, one add per element
    float inv[2];
    const bool ragged = a.L < 256;
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      if (ragged) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * kt + 4 * g + r >= a.L) st[gq][kt][r] = NEG_BIG;
      }
      float mx = NEG_BIG;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[gq][kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float nm = -mx * c;
      // two scores per VALU instruction where the ISA has packed fp32 forms (v_pk_fma_f32, v_pk_add_f32); the exp2 is scalar
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const f32x2 c2 = {c, c}, nm2 = {nm, nm};
      f32x2 ls2 = {0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 e = __builtin_elementwise_fma((f32x2){st[gq][kt][r], st[gq][kt][r + 1]}, c2, nm2);
          const f32x2 p = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
          st[gq][kt][r] = p.x;
          st[gq][kt][r + 1] = p.y;
          ls2 += p;
        }
      float ls = ls2.x + ls2.y;
      ls += __shfl_xor(ls, 16, 64);
      ls += __shfl_xor(ls, 32, 64);
      inv[gq] = 1.0f / ls;
    }
    // O^T += V^T P^T ; k-slot (8g + i) <-> key 32 ks2 + (i < 4 ? 4g + i : 16 + 4g + i - 4)
    f32x4 o[2][DF];
#pragma unroll
    for (int gq = 0; gq < 2; ++gq)
#pragma unroll
      for (int d = 0; d < DF; ++d) o[gq][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 vfr[2][DF];   // V^T fragments, one 32-key step ahead
    auto load_v = [&](int ks2, u32x4 (&dst)[DF]) {
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const u32x2 lo = lds_tr16<DT>(vbase + (32 * ks2) * RP + d * 32);
        const u32x2 hi = lds_tr16<DT>(vbase + (32 * ks2 + 16) * RP + d * 32);
        dst[d] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
      }
    };
    load_v(0, vfr[0]);
#pragma unroll
    for (int ks2 = 0; ks2 < NKT / 2; ++ks2) {
      if (ks2 + 1 < NKT / 2) load_v(ks2 + 1, vfr[(ks2 + 1) & 1]);
      u32x4 pb[2];
#pragma unroll
      for (int gq = 0; gq < 2; ++gq)
        pb[gq] = (u32x4){pack2<DT>(st[gq][2 * ks2][0], st[gq][2 * ks2][1]), pack2<DT>(st[gq][2 * ks2][2], st[gq][2 * ks2][3]),
                         pack2<DT>(st[gq][2 * ks2 + 1][0], st[gq][2 * ks2 + 1][1]),
                         pack2<DT>(st[gq][2 * ks2 + 1][2], st[gq][2 * ks2 + 1][3])};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        o[0][d] = mfma_k32<DT>(vfr[ks2 & 1][d], pb[0], o[0][d]);
        o[1][d] = mfma_k32<DT>(vfr[ks2 & 1][d], pb[1], o[1][d]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      const int q_idx = q0 + gq * 16 + fl;
      if (q_idx < a.L) {
        half_t* orow = a.out + (size_t)(base + (int64_t)q_idx * a.row_stride) * a.D + head * HD;
#pragma unroll
        for (int d = 0; d < DF; ++d) {
          const int dd = 16 * d + 4 * g;
          if (dd < HD) {
            store_out4<DT, LO8>(a, orow + dd, o[gq][d][0] * inv[gq], o[gq][d][1] * inv[gq], o[gq][d][2] * inv[gq], o[gq][d][3] * inv[gq]);
          }
        }
      }
    }
    if (pass == 0) {
#pragma unroll
      for (int gq = 0; gq < 2; ++gq)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[gq][ks] = qn[gq][ks];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// attn_stream (round 3) : L > 256 (the 1024-token spatial attention of Latte-1 / the 512-pixel class-conditional models,
// latte_t2v.py:804-907).  (The round-2 kernel for this shape, attn_blocks_kernel -- stage a 256-key block, wait, compute, restage,
// 128 queries per workgroup: 1.3 GB of L2 -> LDS traffic per launch at Latte-1, 278-300 us for 155 GFLOP -- was REMOVED in round 4:
// it was the only kernel of the library that used scratch memory, and its first launch behind a running streaming kernel ended
// twice in an unexplained GPU memory fault; a kernel that may fault depending on queue state has no place in
// the library, `git show 513ac9e:latte_amd/csrc/attention.hip` has it.)  Here one 8-wave
// workgroup per CU owns 256 queries (32 per wave, two 16-query MFMA column groups sharing every fragment read) and STREAMS the
// keys through a ring of three 128-key blocks (K image | V image, 40 KB each): the LDS DMA of blocks kb + 1 and kb + 2 is in
// flight while block kb is multiplied (counted vmcnt, one barrier per block), and each block is staged half as often.
// Online softmax across blocks: running max on the raw scores (the scale is positive), alpha = exp2((m_old - m_new) c).
// The V^T fragments come from the transpose read in its inline-assembly form: in front of the builtin hipcc drains every LDS
// DMA in flight (see gemm_tn.hip), which would serialise the ring again.
template <int OFF>
__device__ __forceinline__ u32x2 lds_tr16_asm(const char* p) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)(const __attribute__((address_space(3))) char*)p), "i"(OFF));
  return v;
}

template <int HD, int DT, int ABL = 0, bool LO8 = false>
__global__ void __launch_bounds__(512) attn_stream_kernel(AttnArgs a) {
  constexpr int KS = (HD + 31) / 32, DF = (HD + 15) / 16, NCH = HD / 8;
  constexpr int RP = 160, KB = 128, NKT = KB / 16;     // keys per block, 16-key tiles per block
  constexpr int IMG = KB * RP, BLK = 2 * IMG;          // K image | V image
  static_assert(DF <= 5, "the V fragment reads below are written out for up to five 16-wide d fragments");
  extern __shared__ __attribute__((aligned(16))) char smem_attn[];   // 3 x BLK

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fl = lane & 15, g = lane >> 4;
  const int qblocks = (a.L + 255) >> 8;
  int seq, head, qb;
  {
    int b = blockIdx.x;
    const int per_seq = a.heads * qblocks;
    if ((a.num_seq & 7) == 0) {   // the heads and query blocks of one sequence on ONE XCD (shared K / V panels, shared output lines)
      const int xcd = b & 7, slot = b >> 3;
      seq = (slot / per_seq) * 8 + xcd;
      b = slot % per_seq;
    } else {
      seq = b / per_seq;
      b = b % per_seq;
    }
    head = b / qblocks;
    qb = b % qblocks;
  }
  const int64_t base = seq_base_row(a, seq);
  const size_t ld = (size_t)3 * a.D;
  const half_t* qkv_h = a.qkv + (size_t)head * HD;
  const int q0 = qb * 256 + wave * 32;
  const int nkb = (a.L + KB - 1) / KB;

  // Q fragments first, and completed before any DMA is issued (hipcc would otherwise wait for them with vmcnt(0) at their first
  // use, i.e. in the middle of the ring)
  u32x4 qf[2][KS];
#pragma unroll
  for (int gq = 0; gq < 2; ++gq) {
    const int q_ld = min(q0 + gq * 16 + fl, a.L - 1);
    const half_t* qrow = qkv_h + (size_t)(base + (int64_t)q_ld * a.row_stride) * ld;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ch = g + 4 * ks;
      qf[gq][ks] = (u32x4){0u, 0u, 0u, 0u};
      if (ch < NCH) qf[gq][ks] = *(const u32x4*)(qrow + ch * 8);
    }
  }
#pragma unroll
  for (int gq = 0; gq < 2; ++gq)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[gq][ks]));

  // a block = 40 DMA instructions (20 per image: 1280 16-byte chunks); wave w issues 5 of them, all inside one image.  The
  // per-lane part of a source address (key row inside the block, chunk, K or V column offset) does not depend on the block:
  // it is formed once as a 32-bit byte offset (the launcher checks the range); a block adds a wave-uniform base.  Only a ragged
  // last block re-forms the offsets (its rows >= L re-read row L - 1: they meet P = 0 and finite data).
  unsigned voff[5];
  auto lane_offset = [&](int j, int kb) __attribute__((always_inline)) -> unsigned {
    const int inst = wave * 5 + j;                      // 0..19: K image, 20..39: V image
    const int ii = inst >= 20 ? inst - 20 : inst;
    const int idx = ii * 64 + lane;
    const int key = idx / 10, ch = idx - key * 10;
    const int key_ld = min(key, a.L - 1 - kb * KB), ch_ld = min(ch, NCH - 1);
    return (unsigned)(((int64_t)key_ld * a.row_stride * (int64_t)ld + ch_ld * 8 + (inst >= 20 ? 2 : 1) * a.D) * 2);
  };
#pragma unroll
  for (int j = 0; j < 5; ++j) voff[j] = lane_offset(j, 0);
  const bool ragged = (a.L % KB) != 0;
  auto stage = [&](int kb, int slot) __attribute__((always_inline)) {
    char* dst = smem_attn + slot * BLK;
    const char* blk = (const char*)(qkv_h + (size_t)(base + (int64_t)kb * KB * a.row_stride) * ld);   // wave-uniform
    if (ragged && kb == nkb - 1) {
#pragma unroll
      for (int j = 0; j < 5; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(blk + lane_offset(j, kb)),
                                         (__attribute__((address_space(3))) void*)(dst + (wave * 5 + j) * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 5; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(blk + voff[j]),
                                         (__attribute__((address_space(3))) void*)(dst + (wave * 5 + j) * 1024), 16, 0, 0);
    }
  };
  // (ABL 11, measurement build: the five pieces of block kb + 2 spread over the MFMA phases of block kb instead of issued together behind the
  // barrier -- both waves of a SIMD are then not held at their DMA issues at the same time)
  auto stage_piece = [&](int kb, int slot, int j) __attribute__((always_inline)) {
    char* dst = smem_attn + slot * BLK;
    const char* blk = (const char*)(qkv_h + (size_t)(base + (int64_t)kb * KB * a.row_stride) * ld);
    const unsigned off = (ragged && kb == nkb - 1) ? lane_offset(j, kb) : voff[j];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(blk + off),
                                     (__attribute__((address_space(3))) void*)(dst + (wave * 5 + j) * 1024), 16, 0, 0);
  };
  stage(0, 0);
  if (nkb > 1) stage(1, 1);

  const float c = a.scale * 1.4426950408889634f;
  f32x4 o[2][DF];
#pragma unroll
  for (int gq = 0; gq < 2; ++gq)
#pragma unroll
    for (int d = 0; d < DF; ++d) o[gq][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {NEG_BIG, NEG_BIG}, l_run[2] = {0.f, 0.f};
  const bool wave_active = q0 < a.L;
  f32x4 st[2][NKT];     // S^T of one block, then P: [16-query group][16-key tile]

  // S^T = K Q^T of block kb (MFMA phase 1): K fragments two key tiles ahead
  auto scores = [&](int kb) __attribute__((always_inline)) {
    const char* kbase = smem_attn + (kb % 3) * BLK + fl * RP + g * 16;
    u32x4 kf[4][KS];
    auto load_k = [&](int kt, u32x4 (&dst)[KS]) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) dst[ks] = *(const u32x4*)(kbase + kt * 16 * RP + ks * 64);
    };
    load_k(0, kf[0]);
    load_k(1, kf[1]);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt + 2 < NKT) load_k(kt + 2, kf[(kt + 2) & 3]);
      if constexpr (ABL == 11) {
        if ((kt & 1) == 1 && kb + 2 < nkb) stage_piece(kb + 2, (kb + 2) % 3, kt >> 1);   // pieces 0..3 behind key tiles 1, 3, 5, 7
      }
      __builtin_amdgcn_sched_barrier(0);
      st[0][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      st[1][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        st[0][kt] = mfma_k32<DT>(kf[kt & 3][ks], qf[0][ks], st[0][kt]);
        st[1][kt] = mfma_k32<DT>(kf[kt & 3][ks], qf[1][ks], st[1][kt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // online softmax of the block held in st (VALU phase): running max on the raw scores, unconditional rescale
  auto softmax = [&](int kb) __attribute__((always_inline)) {
    const int kleft = a.L - kb * KB;
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      if (kleft < KB) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * kt + 4 * g + r >= kleft) st[gq][kt][r] = NEG_BIG;
      }
      float mx = NEG_BIG;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[gq][kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[gq], mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run[gq] - m_new) * c);   // first block: exp2(-huge) = 0 on o = l = 0
      const float nm = -m_new * c;
      // two scores per VALU instruction where the ISA has packed fp32 forms (v_pk_fma_f32, v_pk_add_f32); the exp2 is scalar
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const f32x2 c2 = {c, c}, nm2 = {nm, nm};
      f32x2 ls2 = {0.f, 0.f};
      float ls;
      if constexpr (ABL == 10) {   // single-issue fp32 forms only (no v_pk_*_f32 beside the partner wave's MFMAs)
        float l0 = 0.f, l1 = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            float e0, e1, p0, p1;
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(st[gq][kt][r]), "v"(c), "v"(nm));
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(st[gq][kt][r + 1]), "v"(c), "v"(nm));
            p0 = __builtin_amdgcn_exp2f(e0);
            p1 = __builtin_amdgcn_exp2f(e1);
            st[gq][kt][r] = p0;
            st[gq][kt][r + 1] = p1;
            l0 += p0;
            l0 += p1;
          }
        ls = l0 + l1;
      } else {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 e = __builtin_elementwise_fma((f32x2){st[gq][kt][r], st[gq][kt][r + 1]}, c2, nm2);
          const f32x2 p = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
          st[gq][kt][r] = p.x;
          st[gq][kt][r + 1] = p.y;
          ls2 += p;
        }
      ls = ls2.x + ls2.y;
      }
      ls += __shfl_xor(ls, 16, 64);
      ls += __shfl_xor(ls, 32, 64);
      l_run[gq] = l_run[gq] * alpha + ls;
      // no query of this wave raised its maximum (the usual case after the first blocks): alpha is exactly 1 in every lane and
      // the rescale of the accumulators is the identity -- skipped, bit-identical
      const bool raised = __builtin_amdgcn_ballot_w64(m_new != m_run[gq]) != 0;
      m_run[gq] = m_new;
      if (raised) {
#pragma unroll
        for (int d = 0; d < DF; ++d) o[gq][d] *= alpha;
      }
    }
  };
  auto softmax_sel = [&](int kb) __attribute__((always_inline)) { if constexpr (ABL != 8) softmax(kb); };   // (ABL: measurement ablations, results garbage)
  // O^T += V^T P^T of block kb (MFMA phase 2); k-slot (8g + i) <-> key 32 ks2 + (i < 4 ? 4g + i : 16 + 4g + i - 4).  V^T
  // fragments one 32-key step ahead; the reads are inline assembly, so their completion is counted here: 2 DF reads per step
  auto weighted_sum = [&](int kb) __attribute__((always_inline)) {
    const char* vbase = smem_attn + (kb % 3) * BLK + IMG + (4 * g + (fl >> 2)) * RP + (fl & 3) * 8;
    u32x2 vlo[2][5], vhi[2][5];
    auto load_v = [&](int ks2, u32x2 (&lo)[5], u32x2 (&hi)[5]) {
      const char* pv = vbase + (32 * ks2) * RP;
      lo[0] = lds_tr16_asm<0>(pv); hi[0] = lds_tr16_asm<16 * RP>(pv);
      lo[1] = lds_tr16_asm<32>(pv); hi[1] = lds_tr16_asm<16 * RP + 32>(pv);
      lo[2] = lds_tr16_asm<64>(pv); hi[2] = lds_tr16_asm<16 * RP + 64>(pv);
      lo[3] = lds_tr16_asm<96>(pv); hi[3] = lds_tr16_asm<16 * RP + 96>(pv);
      if constexpr (DF == 5) { lo[4] = lds_tr16_asm<128>(pv); hi[4] = lds_tr16_asm<16 * RP + 128>(pv); }
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the compiler's own LDS traffic of the softmax shuffles)
    load_v(0, vlo[0], vhi[0]);
#pragma unroll
    for (int ks2 = 0; ks2 < NKT / 2; ++ks2) {
      if (ks2 + 1 < NKT / 2) {
        load_v(ks2 + 1, vlo[(ks2 + 1) & 1], vhi[(ks2 + 1) & 1]);
        if constexpr (DF == 5) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if constexpr (ABL == 11) {
        if (ks2 == 1 && kb + 2 < nkb) stage_piece(kb + 2, (kb + 2) % 3, 4);
      }
      __builtin_amdgcn_sched_barrier(0);
      u32x4 pb[2];
#pragma unroll
      for (int gq = 0; gq < 2; ++gq)
        pb[gq] = (u32x4){pack2<DT>(st[gq][2 * ks2][0], st[gq][2 * ks2][1]), pack2<DT>(st[gq][2 * ks2][2], st[gq][2 * ks2][3]),
                         pack2<DT>(st[gq][2 * ks2 + 1][0], st[gq][2 * ks2 + 1][1]),
                         pack2<DT>(st[gq][2 * ks2 + 1][2], st[gq][2 * ks2 + 1][3])};
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const u32x4 vfrag = {vlo[ks2 & 1][d][0], vlo[ks2 & 1][d][1], vhi[ks2 & 1][d][0], vhi[ks2 & 1][d][1]};
        o[0][d] = mfma_k32<DT>(vfrag, pb[0], o[0][d]);
        o[1][d] = mfma_k32<DT>(vfrag, pb[1], o[1][d]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Interval kb: own DMA of block kb landed (block kb + 1 may stay in flight), then everybody's; the barrier also retires every
  // read of block kb - 1, whose slot takes block kb + 2.  (Running waves 4-7 one phase behind waves 0-3 -- softmax of block
  // kb - 1 under the other wave's score MFMAs -- was built and measured: 250 against 242 us, no gain; ABL: measurement ablations.)
  for (int kb = 0; kb < nkb; ++kb) {
    if constexpr (ABL != 9) {
      if (kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if constexpr (ABL != 7 && ABL != 11) {
      if (kb + 2 < nkb) stage(kb + 2, (kb + 2) % 3);
    }
    if constexpr (ABL == 11) {   // (a wave without queries still has to bring its pieces)
      if (!wave_active && kb + 2 < nkb) stage(kb + 2, (kb + 2) % 3);
    }
    if (wave_active) {
      scores(kb);
      softmax_sel(kb);
      weighted_sum(kb);
    }
  }
  if (!wave_active) return;
#pragma unroll
  for (int gq = 0; gq < 2; ++gq) {
    const int q_idx = q0 + gq * 16 + fl;
    if (q_idx < a.L) {
      const float inv = 1.0f / l_run[gq];
      half_t* orow = a.out + (size_t)(base + (int64_t)q_idx * a.row_stride) * a.D + head * HD;
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const int dd = 16 * d + 4 * g;
        if (dd < HD) {
          store_out4<DT, LO8>(a, orow + dd, o[gq][d][0] * inv, o[gq][d][1] * inv, o[gq][d][2] * inv, o[gq][d][3] * inv);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Round 6c: the L > 256 attention on the 32 x 32 x 16 MFMA shape (attn_stream64_kernel below; the first form, attn_stream32_kernel -- the
// 8-wave kernel above with these tiles and nothing else changed: 251.8 against 243.2 us -- is `git show 9a6a033:latte_amd/csrc/attention.hip`).
//   S^T = K Q^T as 32-key x 32-query tiles: a lane holds, for ITS query (lane & 31), 16 keys of every tile (key = (r & 3) + 8 (r >> 2) +
//   4 (lane >> 5)), so the row maximum and the row sum are in-lane chains with one half-wave exchange (v_permlane32_swap) per block for the
//   maximum and one per launch for the sum -- no ds_bpermute in the loop;
//   O^T = V^T P^T: registers 8 s .. 8 s + 7 of a tile ARE the B operand of k-step s (k-slot 8 hi + j <-> key 16 s + 4 hi + (j & 3) +
//   8 (j >> 2)) -- P is packed to half where it is produced and never moves; the V^T operand comes from two transpose reads per d tile.
// K rows sit at a pitch of 144 bytes (9 chunks: conflict-free ds_read_b128 for 32-row fragments).
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int DT>
__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
  if constexpr (DT == LATTE_DTYPE_BF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float fma1(float a, float b, float c) {   // one v_fma_f32 the SLP vectoriser cannot pair
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// exchange with the other half of the wave: after v_permlane32_swap a = (a.lo, b.lo), b = (a.hi, b.hi) (rows of 32 lanes), so with a = b = x
// every lane holds its own and its counterpart's value.  Inline assembly: through the builtin hipcc folded max(r[0], r[1]) into r[0]
// (the exchange result was dropped; measured on the GPU as a per-half maximum); the s_nops are the wait states the compiler puts around it.
__device__ __forceinline__ void half_swap(float x, float& r0, float& r1) {
  r0 = x;
  r1 = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(r0), "+v"(r1));
}
__device__ __forceinline__ float half_swap_max(float x) {
  float r0, r1;
  half_swap(x, r0, r1);
  return fmaxf(r0, r1);
}
__device__ __forceinline__ float half_swap_sum(float x) {
  float r0, r1;
  half_swap(x, r0, r1);
  return r0 + r1;
}

// ------------------------------------------------------------------------------------------------
// attn_stream64 (round 6c): L > 256, head dim 72, on ONE wave per SIMD.  A workgroup is 4 waves = 256 queries of a (sequence, head);
// a wave owns 64 queries (two 32-query column groups) and the whole 512-register file, so every K fragment and every V^T fragment
// read from LDS feeds TWO MFMAs, and the softmax is software-pipelined against the matrix work inside the wave:
//   iteration i:  phase 1   S(i+1) = K(i+1) Q^T, both groups            (40 MFMAs, 20 ds_read_b128)
//                 phase 2   O += V(i)^T P(i)^T, both groups             (48 MFMAs, 48 transpose reads)
//                           with the softmax of block i+1 -- S(i+1) -> P(i+1) -- in the same instruction stream: the exponentials of
//                           key tile t are written over P(i)'s tile t after the MFMAs that read it have been issued.
// 64-key blocks; K and V have their own rings of four (K(i+1) and V(i) are read in the same iteration); one barrier per iteration.
// The V image has a pitch of 160 bytes (the tenth chunk is padding) with the keys of every 8-key group stored as key 8 b + 4 h + i -> row
// 8 b + 2 i + h: the four key rows a 32-lane group of a transpose read touches are then 16 banks apart.
// Measured (profiles/r6c_attn_stream32_stream64_probe.log): 278 us (NG = 2) / 248 us (NG = 1) against the 8-wave
// 16 x 16 kernel's 243 - 251 us on the Latte-1 shape -- NOT the default; kept as the worked example of the shape and of what it costs.
constexpr int S64_KP = 144, S64_VP = 160, S64_KB = 64, S64_NS = 4, S64_KIMG = S64_KB * S64_KP, S64_VIMG = S64_KB * S64_VP;
constexpr int STREAM64_LDS = S64_NS * (S64_KIMG + S64_VIMG) + 256;

// NG = 32-query groups per wave: 2 = the one-wave-per-SIMD form described above (4 waves); 1 = the SAME pipeline on 8 waves x 32 queries, two
// waves per SIMD (everything fits the architectural registers: the S MFMAs are the compiler's) -- `attn_variant` 13.
template <int HD, int DT, bool LO8 = false, int ABL = 0, int NG = 2>
__global__ void __launch_bounds__(512 / NG) attn_stream64_kernel(AttnArgs a) {
  constexpr int NWV = 8 / NG;                    // waves
  constexpr int NPC = 19, NPW = (NPC + NWV - 1) / NWV;   // DMA pieces per issue group (9 K + 10 V), per wave
  constexpr int KS = (HD + 15) / 16, DTL = (HD + 31) / 32, NCH = HD / 8;
  constexpr int KP = S64_KP, VP = S64_VP, KB = S64_KB, NT = KB / 32, KIMG = S64_KIMG, VIMG = S64_VIMG, NS = S64_NS;
  static_assert(HD == 72, "written for 9-chunk rows");
  extern __shared__ __attribute__((aligned(16))) char smem_attn[];
  char* const kring = smem_attn;
  char* const vring = smem_attn + NS * KIMG;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ql = lane & 31, hi = lane >> 5;
  const int qblocks = (a.L + 255) >> 8;
  int seq, head, qb;
  {
    int b = blockIdx.x;
    const int per_seq = a.heads * qblocks;
    if ((a.num_seq & 7) == 0) {   // the heads and query blocks of one sequence on ONE XCD
      const int xcd = b & 7, slot = b >> 3;
      seq = (slot / per_seq) * 8 + xcd;
      b = slot % per_seq;
    } else {
      seq = b / per_seq;
      b = b % per_seq;
    }
    head = b / qblocks;
    qb = b % qblocks;
  }
  const int64_t base = seq_base_row(a, seq);
  const size_t ld = (size_t)3 * a.D;
  const half_t* qkv_h = a.qkv + (size_t)head * HD;
  const int q0 = qb * 256 + wave * 32 * NG;
  const int nkb = (a.L + KB - 1) / KB;

  u32x4 qf[NG][KS];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int q_ld = min(q0 + 32 * g + ql, a.L - 1);
    const half_t* qrow = qkv_h + (size_t)(base + (int64_t)q_ld * a.row_stride) * ld;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ch = 2 * ks + hi;
      qf[g][ks] = (u32x4){0u, 0u, 0u, 0u};
      if (ch < NCH) qf[g][ks] = *(const u32x4*)(qrow + ch * 8);
    }
  }
  // (the loads complete here; "+a": the Q fragments live in the accumulator half of the register file for the whole kernel -- MFMA
  // reads its B operand from there directly -- which leaves the architectural half to the scores and the softmax)
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if constexpr (NG == 2) asm volatile("" : "+a"(qf[g][ks]));
      else asm volatile("" : "+v"(qf[g][ks]));
    }

  // DMA.  K image of a 64-key block: 9 instructions (9 chunks x 64 rows); V image: 10 (10 chunks x 64 rows, LDS row r holds key
  // (r & ~7) | ((r & 1) << 2) | ((r >> 1) & 3)).  An issue group G(j) = [K(j + 1), V(j)] is 19 pieces; EVERY wave issues NPW of them --
  // piece j of wave w is min(w + NWV j, 18): the last V piece is written more than once with the same bytes -- and a block past the end
  // re-reads the last block into the (free) slot it would have taken: no branch around a DMA piece, one vmcnt value for every wave
  // and iteration.
  auto piece_offset = [&](int j, int kb) __attribute__((always_inline)) -> unsigned {
    const int pc = min(wave + NWV * j, NPC - 1);
    const bool isv = pc >= 9;
    const int idx = (isv ? pc - 9 : pc) * 64 + lane;
    const int row = isv ? idx / 10 : idx / 9, ch = idx - row * (isv ? 10 : 9);
    const int key = isv ? ((row & ~7) | ((row & 1) << 2) | ((row >> 1) & 3)) : row;
    const int key_ld = min(key, a.L - 1 - kb * KB), ch_ld = min(ch, NCH - 1);
    return (unsigned)(((int64_t)key_ld * a.row_stride * (int64_t)ld + ch_ld * 8 + (isv ? 2 : 1) * a.D) * 2);
  };
  unsigned poff[NPW], poff_l[NPW];   // whole blocks / the last block (rows >= L re-read row L - 1: they meet P = 0)
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    poff[j] = piece_offset(j, 0);
    poff_l[j] = piece_offset(j, nkb - 1);
  }
  // the wave-uniform part of an issue group's addresses, formed ONCE per group (as scalar arithmetic per piece it was ~65 SALU
  // instructions per iteration and wave: 64-bit products of the block index)
  struct Group { const char* gk; const char* gv; char* dk; char* dv; bool lk, lv; };
  const char* const blk0 = (const char*)(qkv_h + (size_t)base * ld);
  const int64_t blk_step = (int64_t)KB * a.row_stride * (int64_t)ld * 2;
  auto make_group = [&](int kbk, int kbv) __attribute__((always_inline)) -> Group {
    const int kk = min(kbk, nkb - 1), kv = min(kbv, nkb - 1);
    return Group{blk0 + kk * blk_step, blk0 + kv * blk_step, kring + (kbk % NS) * KIMG, vring + (kbv % NS) * VIMG, kk == nkb - 1, kv == nkb - 1};
  };
  auto stage_piece = [&](const Group& gr, int j) __attribute__((always_inline)) {
    const int pc = min(wave + NWV * j, NPC - 1);
    const bool isv = pc >= 9;
    char* dst = isv ? gr.dv + (pc - 9) * 1024 : gr.dk + pc * 1024;
    const char* blk = isv ? gr.gv : gr.gk;
    const unsigned off = (isv ? gr.lv : gr.lk) ? poff_l[j] : poff[j];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(blk + off),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto stage_group = [&](int kbk, int kbv) __attribute__((always_inline)) {
    const Group gr = make_group(kbk, kbv);
#pragma unroll
    for (int j = 0; j < NPW; ++j) stage_piece(gr, j);
  };
  // [K(0)] (its pieces only), then G(0..2) here, G(i + 3) in iteration i
#pragma unroll
  for (int j = 0; j < NPW; ++j)
    if (wave + NWV * j < 9) stage_piece(make_group(0, 0), j);
  stage_group(1, 0);
  stage_group(2, 1);
  stage_group(3, 2);

  const float c = a.scale * 1.4426950408889634f;
  f32x16 o[NG][DTL];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int d = 0; d < DTL; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[g][d][r] = 0.f;
  float m_run[NG], l_run[NG], alpha[NG], nm[NG], ls[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) { m_run[g] = NEG_BIG; l_run[g] = 0.f; alpha[g] = 1.f; nm[g] = 0.f; ls[g] = 0.f; }
  bool raised = false;
  const bool wave_active = q0 < a.L;
  f32x16 st[NG][NT];
  unsigned pk[NG][NT][8];

  const int kread = ql * KP + hi * 16;
  const int vread = (2 * ((lane & 15) >> 2) + hi) * VP + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

  // phase 1: S(kb), both groups, every K fragment read once.  The MFMAs are inline assembly with architectural destination registers:
  // hipcc puts every MFMA result of a kernel this size into the accumulator half and the softmax then pays one v_accvgpr_read per score
  // and pass (128 per block, measured in the listing).  What the compiler does for its own MFMAs is done by hand here: the chain of a
  // (group, tile) accumulates in place (back-to-back srcC = vDst needs no wait state), and its last MFMA carries the 20 wait states
  // a VALU read of a 16-pass result needs -- hidden, the matrix pipe is busy for 32 cycles anyway.
  auto scores = [&](int kb, const Group* dma) __attribute__((always_inline)) {
    const char* kbase = kring + (kb % NS) * KIMG + kread;
    u32x4 kf[2][KS];
    auto load_k = [&](int t, u32x4 (&dst)[KS]) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) dst[ks] = *(const u32x4*)(kbase + t * 32 * KP + ks * 32);
    };
    static_assert(NT == 2, "both key tiles' fragments are held at once");
    load_k(0, kf[0]);
    load_k(1, kf[1]);
    // k-step outermost: the NT NG accumulator chains advance in turn (a chain's next MFMA waits for its previous result: 16 passes)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (ABL != 7 && dma != nullptr && g == 0) {   // (the prologue call passes nullptr: compile-time after inlining)
            if constexpr (NG == 2) {
              if (ks == 1 && t == 0) stage_piece(*dma, 0);   // pieces 0, 1 of the group behind 4 / 12 MFMAs of the phase
              if (ks == 3 && t == 0) stage_piece(*dma, 1);
            } else {
              if (ks == 2 && t == 0) stage_piece(*dma, 0);
            }
          }
          if constexpr (NG == 1) {
            if (ks == 0) {
#pragma unroll
              for (int r = 0; r < 16; ++r) st[g][t][r] = 0.f;
            }
            st[g][t] = mfma32<DT>(kf[t & 1][ks], qf[g][ks], st[g][t]);
          } else
          if constexpr (DT == LATTE_DTYPE_BF16) {
            if (ks == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(st[g][t]) : "v"(kf[t & 1][ks]), "a"(qf[g][ks]));
            else if (ks + 1 < KS) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(st[g][t]) : "v"(kf[t & 1][ks]), "a"(qf[g][ks]));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(st[g][t]) : "v"(kf[t & 1][ks]), "a"(qf[g][ks]));
          } else {
            if (ks == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(st[g][t]) : "v"(kf[t & 1][ks]), "a"(qf[g][ks]));
            else if (ks + 1 < KS) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(st[g][t]) : "v"(kf[t & 1][ks]), "a"(qf[g][ks]));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(st[g][t]) : "v"(kf[t & 1][ks]), "a"(qf[g][ks]));
          }
        }
    }
  };
  // softmax of block kb, part 1: masks (MASKED: the block may be ragged), row maxima, the rescale factor of the accumulators
  auto softmax_max = [&](int kb, auto masked) __attribute__((always_inline)) {
    raised = false;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if constexpr (decltype(masked)::value) {
        const int kleft = a.L - kb * KB;
        if (kleft < KB) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi >= kleft) st[g][t][r] = NEG_BIG;
        }
      }
      float mx = NEG_BIG;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[g][t][r]);
      mx = half_swap_max(mx);
      // Lazy reference maximum: the running value is only raised when some query of the group found a score more than LAZY (in
      // exp2 units) above it -- on every block it is raised for all lanes that exceed theirs at all, the block's P then is <= 1; in
      // between P <= 2^LAZY, which half holds with the same relative rounding, and the sums are fp32.  With 64 queries per wave a
      // raise by ANY amount happens in nearly every block (the accumulators sit in the accumulator registers: 5 instructions per
      // pair to rescale them, 240 per block); a raise by 2^8 only in the first block or two of a sequence.
      constexpr float LAZY = 8.0f;
      const bool up = __builtin_amdgcn_ballot_w64((mx - m_run[g]) * c > LAZY) != 0;
      const float m_new = up ? fmaxf(m_run[g], mx) : m_run[g];
      alpha[g] = up ? __builtin_amdgcn_exp2f((m_run[g] - m_new) * c) : 1.0f;
      nm[g] = -m_new * c;
      raised = raised || up;
      m_run[g] = m_new;
      ls[g] = 0.f;
    }
  };
  // part 2, one key tile: exponentials -> packed P (in place), row-sum chain
  auto softmax_exp = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(fma1(st[g][t][r], c, nm[g]));
        const float p1 = __builtin_amdgcn_exp2f(fma1(st[g][t][r + 1], c, nm[g]));
        pk[g][t][r >> 1] = pack2<DT>(p0, p1);
        ls[g] += p0;
        ls[g] += p1;
      }
  };
  auto softmax_end = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < NG; ++g) l_run[g] = l_run[g] * alpha[g] + ls[g];
  };

  // V^T fragments of k-step `step` of the block at vbase (two transpose reads per d tile); completion is counted by hand
  auto load_v = [&](const char* vbase, int step, u32x2 (&lo)[DTL], u32x2 (&hi2)[DTL]) __attribute__((always_inline)) {
    const char* pv = vbase + 16 * step * VP;
    lo[0] = lds_tr16_asm<0>(pv); hi2[0] = lds_tr16_asm<8 * VP>(pv);
    lo[1] = lds_tr16_asm<64>(pv); hi2[1] = lds_tr16_asm<8 * VP + 64>(pv);
    lo[2] = lds_tr16_asm<128>(pv); hi2[2] = lds_tr16_asm<8 * VP + 128>(pv);
  };

  // prologue: K(0) landed (G(0..2) may stay in flight: 3 x (nk + nv) instructions of this wave) -> S(0), softmax(0) with nothing beside it
  if constexpr (NG == 2) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");   // K(0): everything but G(0..2), NPW pieces each
  else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  typedef std::integral_constant<bool, true> yes_t;
  typedef std::integral_constant<bool, false> no_t;
  if (wave_active) {
    scores(0, nullptr);
    softmax_max(0, yes_t{});
#pragma unroll
    for (int t = 0; t < NT; ++t) softmax_exp(t);
    softmax_end();
  }
  // one iteration: G(i) landed; [rescale]; S(i+1); max(i+1); P(i) with exp(i+1) in the same stream.  MORE = block i+1 exists,
  // MASKED = block i+1 may be ragged: the steady-state instance (MORE, not MASKED) has no branch between its MFMAs.
  auto iteration = [&](int i, auto more, auto masked) __attribute__((always_inline)) {
    constexpr bool MORE = decltype(more)::value;
    if constexpr (ABL != 9) {   // G(i) landed; G(i+1), G(i+2) may stay in flight
      if constexpr (NG == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    if constexpr (ABL != 9) __builtin_amdgcn_s_barrier();
    if (!wave_active) {
      if constexpr (ABL != 7) stage_group(i + 4, i + 3);
      return;
    }
    const Group gr = make_group(i + 4, i + 3);
    // V^T fragments of P(i)'s first two k-steps are requested before anything else (V(i) has landed): three buffers, two steps ahead
    const char* vbase = vring + (i % NS) * VIMG + vread;
    u32x2 vlo[3][DTL], vhi[3][DTL];
    load_v(vbase, 0, vlo[0], vhi[0]);
    load_v(vbase, 1, vlo[1], vhi[1]);
    if (raised) {   // the accumulators meet P(i): rescale by the factor softmax(i) found
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int d = 0; d < DTL; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[g][d][r] *= alpha[g];
    }
    if constexpr (MORE) {
      scores(i + 1, &gr);
      if constexpr (ABL != 8) softmax_max(i + 1, masked);
    } else {
      if constexpr (ABL != 7) {
        stage_piece(gr, 0);
        if constexpr (NG == 2) stage_piece(gr, 1);
      }
    }
#pragma unroll
    for (int step = 0; step < 2 * NT; ++step) {
      const int cur = step % 3;
      if (step + 2 < 2 * NT) {
        load_v(vbase, step + 2, vlo[(step + 2) % 3], vhi[(step + 2) % 3]);
        asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(vlo[cur][0]), "+v"(vlo[cur][1]), "+v"(vlo[cur][2]), "+v"(vhi[cur][0]), "+v"(vhi[cur][1]), "+v"(vhi[cur][2]) :: "memory");
      } else if (step + 1 < 2 * NT) {
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(vlo[cur][0]), "+v"(vlo[cur][1]), "+v"(vlo[cur][2]), "+v"(vhi[cur][0]), "+v"(vhi[cur][1]), "+v"(vhi[cur][2]) :: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[cur][0]), "+v"(vlo[cur][1]), "+v"(vlo[cur][2]), "+v"(vhi[cur][0]), "+v"(vhi[cur][1]), "+v"(vhi[cur][2]) :: "memory");
      }
      if constexpr (ABL != 7) {   // the group's remaining pieces between the k-steps
        if constexpr (NG == 2) {
          if (step < 3) stage_piece(gr, 2 + step);
        } else {
          if (step == 0 || step == 2) stage_piece(gr, 1 + step / 2);
        }
      }
      const int t = step >> 1, s2 = step & 1;
#pragma unroll
      for (int d = 0; d < DTL; ++d) {
        const u32x4 vfrag = {vlo[cur][d][0], vlo[cur][d][1], vhi[cur][d][0], vhi[cur][d][1]};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const u32x4 pb = {pk[g][t][4 * s2], pk[g][t][4 * s2 + 1], pk[g][t][4 * s2 + 2], pk[g][t][4 * s2 + 3]};
          o[g][d] = mfma32<DT>(vfrag, pb, o[g][d]);
        }
      }
      if constexpr (MORE && ABL != 8) {
        if (s2 == 1) softmax_exp(t);   // P(i)'s tile t has been read: P(i+1)'s tile t takes its registers
      }
    }
    if constexpr (MORE) softmax_end();
  };
  for (int i = 0; i + 2 < nkb; ++i) iteration(i, yes_t{}, no_t{});
  if (nkb >= 2) iteration(nkb - 2, yes_t{}, yes_t{});
  iteration(nkb - 1, no_t{}, no_t{});
  if (!wave_active) return;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int q_idx = q0 + 32 * g + ql;
    const float inv = 1.0f / half_swap_sum(l_run[g]);
    if (q_idx < a.L) {
      half_t* orow = a.out + (size_t)(base + (int64_t)q_idx * a.row_stride) * a.D + head * HD;
#pragma unroll
      for (int d = 0; d < DTL; ++d)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int dd = 32 * d + 8 * r4 + 4 * hi;
          if (dd < HD)
            store_out4<DT, LO8>(a, orow + dd, o[g][d][4 * r4] * inv, o[g][d][4 * r4 + 1] * inv, o[g][d][4 * r4 + 2] * inv,
                                o[g][d][4 * r4 + 3] * inv);
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// attn_cross (round 3): the text cross-attention of LatteT2V (attn2, latte_t2v.py:740-760) for Lk <= 128 keys (Latte-1: 120 T5
// tokens).  The generic flash kernel gave every 64 queries their own workgroup, each staging the sample's K / V through registers
// in two 64-key tiles with an online softmax between them: 102 us per launch at the Latte-1 shape for 18 GFLOP and 150 MB.
// Here one 8-wave workgroup owns 256 queries of a (frame, head) as in attn_stream_kernel, ALL keys and values of the head are
// staged ONCE by LDS DMA (two 128-row images, rows >= Lk re-read row Lk - 1), the score bias of the caption mask sits in LDS,
// and the softmax over the <= 128 scores a query meets is exact in registers (no running maximum, no rescale).
// z = s * scale * log2(e) + bias * log2(e) as in the flash kernel; keys >= Lk get -1e30.
template <int HD, int DT>
__global__ void __launch_bounds__(512) attn_cross_kernel(AttnArgs a) {
  constexpr int KS = (HD + 31) / 32, DF = (HD + 15) / 16, NCH = HD / 8;
  constexpr int RP = 160, KB = 128, NKT = KB / 16;
  constexpr int IMG = KB * RP;                          // K image | V image | bias[128]
  static_assert(DF <= 5, "the V fragment reads below are written out for up to five 16-wide d fragments");
  extern __shared__ __attribute__((aligned(16))) char smem_attn[];
  float* const bias_lds = (float*)(smem_attn + 2 * IMG);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fl = lane & 15, g = lane >> 4;
  const int qblocks = (a.L + 255) >> 8;
  int seq, head, qb;
  {
    int b = blockIdx.x;
    const int per_seq = a.heads * qblocks;
    if ((a.num_seq & 7) == 0) {   // the heads and query blocks of one sequence on ONE XCD (shared output lines, one K / V panel)
      const int xcd = b & 7, slot = b >> 3;
      seq = (slot / per_seq) * 8 + xcd;
      b = slot % per_seq;
    } else {
      seq = b / per_seq;
      b = b % per_seq;
    }
    head = b / qblocks;
    qb = b % qblocks;
  }
  const int64_t base = seq_base_row(a, seq);
  const size_t ld = (size_t)a.q_ld;
  const half_t* q_h = a.qkv + (size_t)head * HD;
  const int NK = a.Lk, smp = seq / a.U;
  const half_t* kv_h = a.kv + ((size_t)smp * a.Lk) * (2 * (size_t)a.D) + (size_t)head * HD;
  const int q0 = qb * 256 + wave * 32;

  u32x4 qf[2][KS];
#pragma unroll
  for (int gq = 0; gq < 2; ++gq) {
    const int q_ld = min(q0 + gq * 16 + fl, a.L - 1);
    const half_t* qrow = q_h + (size_t)(base + (int64_t)q_ld * a.row_stride) * ld;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ch = g + 4 * ks;
      qf[gq][ks] = (u32x4){0u, 0u, 0u, 0u};
      if (ch < NCH) qf[gq][ks] = *(const u32x4*)(qrow + ch * 8);
    }
  }
  // 40 DMA instructions (20 per image: 1280 16-byte chunks), 5 per wave, all of a wave inside one image
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int inst = wave * 5 + j;                      // 0..19: K image, 20..39: V image
    const int ii = inst >= 20 ? inst - 20 : inst;
    const int idx = ii * 64 + lane;
    const int key = idx / 10, ch = idx - key * 10;
    const int key_ld = min(key, NK - 1), ch_ld = min(ch, NCH - 1);
    const half_t* rowp = kv_h + (size_t)key_ld * (2 * (size_t)a.D) + ch_ld * 8 + (inst >= 20 ? a.D : 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)rowp,
                                     (__attribute__((address_space(3))) void*)(smem_attn + inst * 1024), 16, 0, 0);
  }
  if (threadIdx.x < KB) {
    const int k = threadIdx.x;
    float b = NEG_BIG;
    if (k < NK) b = a.kbias != nullptr ? a.kbias[(size_t)smp * a.Lk + k] * 1.4426950408889634f : 0.f;
    bias_lds[k] = b;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (q0 >= a.L) return;

  const float c = a.scale * 1.4426950408889634f;
  f32x4 st[2][NKT];
  {  // S^T = K Q^T, K fragments two key tiles ahead
    const char* kbase = smem_attn + fl * RP + g * 16;
    u32x4 kf[4][KS];
    auto load_k = [&](int kt, u32x4 (&dst)[KS]) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) dst[ks] = *(const u32x4*)(kbase + kt * 16 * RP + ks * 64);
    };
    load_k(0, kf[0]);
    load_k(1, kf[1]);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt + 2 < NKT) load_k(kt + 2, kf[(kt + 2) & 3]);
      __builtin_amdgcn_sched_barrier(0);
      st[0][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      st[1][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        st[0][kt] = mfma_k32<DT>(kf[kt & 3][ks], qf[0][ks], st[0][kt]);
        st[1][kt] = mfma_k32<DT>(kf[kt & 3][ks], qf[1][ks], st[1][kt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float inv[2];
#pragma unroll
  for (int gq = 0; gq < 2; ++gq) {
    float mx = NEG_BIG;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const float4 b4 = *(const float4*)(bias_lds + 16 * kt + 4 * g);     // keys 16 kt + 4 g + r
      st[gq][kt][0] = fmaf(st[gq][kt][0], c, b4.x);
      st[gq][kt][1] = fmaf(st[gq][kt][1], c, b4.y);
      st[gq][kt][2] = fmaf(st[gq][kt][2], c, b4.z);
      st[gq][kt][3] = fmaf(st[gq][kt][3], c, b4.w);
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[gq][kt][r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float ls = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(st[gq][kt][r] - mx);
        st[gq][kt][r] = p;
        ls += p;
      }
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    inv[gq] = 1.0f / ls;
  }
  // O^T = V^T P^T ; k-slot (8g + i) <-> key 32 ks2 + (i < 4 ? 4g + i : 16 + 4g + i - 4)
  f32x4 o[2][DF];
#pragma unroll
  for (int gq = 0; gq < 2; ++gq)
#pragma unroll
    for (int d = 0; d < DF; ++d) o[gq][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const char* vbase = smem_attn + IMG + (4 * g + (fl >> 2)) * RP + (fl & 3) * 8;
#pragma unroll
    for (int ks2 = 0; ks2 < NKT / 2; ++ks2) {
      u32x4 pb[2];
#pragma unroll
      for (int gq = 0; gq < 2; ++gq)
        pb[gq] = (u32x4){pack2<DT>(st[gq][2 * ks2][0], st[gq][2 * ks2][1]), pack2<DT>(st[gq][2 * ks2][2], st[gq][2 * ks2][3]),
                         pack2<DT>(st[gq][2 * ks2 + 1][0], st[gq][2 * ks2 + 1][1]),
                         pack2<DT>(st[gq][2 * ks2 + 1][2], st[gq][2 * ks2 + 1][3])};
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const char* vb = vbase + (32 * ks2) * RP + d * 32;
        const u32x2 lo = lds_tr16<DT>(vb);
        const u32x2 hi = lds_tr16<DT>(vb + 16 * RP);
        const u32x4 vfrag = {lo[0], lo[1], hi[0], hi[1]};
        o[0][d] = mfma_k32<DT>(vfrag, pb[0], o[0][d]);
        o[1][d] = mfma_k32<DT>(vfrag, pb[1], o[1][d]);
      }
    }
  }
#pragma unroll
  for (int gq = 0; gq < 2; ++gq) {
    const int q_idx = q0 + gq * 16 + fl;
    if (q_idx < a.L) {
      half_t* orow = a.out + (size_t)(base + (int64_t)q_idx * a.row_stride) * a.D + head * HD;
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const int dd = 16 * d + 4 * g;
        if (dd < HD) {
          u32x2 pk = {pack2<DT>(o[gq][d][0] * inv[gq], o[gq][d][1] * inv[gq]), pack2<DT>(o[gq][d][2] * inv[gq], o[gq][d][3] * inv[gq])};
          *(u32x2*)(orow + dd) = pk;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <int HD, int DT, bool LO8 = false>
__global__ void __launch_bounds__(256) attn_small_kernel(AttnArgs a) {
  constexpr int KS = (HD + 31) / 32;
  constexpr int DF = (HD + 15) / 16;
  constexpr int NCH = HD / 8;
  constexpr int VP = 160;   // row pitch of the V patch: conflict-free for the transpose reads (as in attn_full_kernel)
  __shared__ __attribute__((aligned(16))) char lds[4 * 17 * VP];  // per wave: 16 V rows (+1 row of slack)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fl = lane & 15, g = lane >> 4;
  const int items = a.num_seq * a.heads;
  // Block -> 4 (sequence, head) items.  As in attn_full_kernel the head slices of a token row share 128-byte lines, so the
  // blocks that cover the heads of one sequence are kept on ONE XCD (blocks b, b + 8, ... walk the head groups).
  int blk = blockIdx.x;
  const int hgroups = a.heads >> 2;
  if ((a.heads & 3) == 0 && (a.num_seq & 7) == 0) {
    const int xcd = blk & 7, slot = blk >> 3;
    blk = ((slot / hgroups) * 8 + xcd) * hgroups + slot % hgroups;
  }
  int item = blk * 4 + wave;
  const bool active = item < items;
  item = min(item, items - 1);
  const int seq = item / a.heads, head = item % a.heads;
  const int64_t base = seq_base_row(a, seq);
  const size_t ld = (size_t)3 * a.D;
  const int tok = min(fl, a.L - 1);
  const half_t* rowp = a.qkv + (size_t)(base + (int64_t)tok * a.row_stride) * ld + (size_t)head * HD;
  char* v_lds = lds + wave * 17 * VP;

  u32x4 qf[KS], kf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int ch = g + 4 * ks;
    qf[ks] = (u32x4){0u, 0u, 0u, 0u};
    kf[ks] = (u32x4){0u, 0u, 0u, 0u};
    if (ch < NCH) {
      qf[ks] = *(const u32x4*)(rowp + ch * 8);
      kf[ks] = *(const u32x4*)(rowp + a.D + ch * 8);
      *(u32x4*)(v_lds + fl * VP + ch * 16) = *(const u32x4*)(rowp + 2 * a.D + ch * 8);
    }
  }
  f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) st = mfma_k32<DT>(kf[ks], qf[ks], st);  // S^T[key = 4g + r][q = fl]

  // softmax in the exp2 domain, max on the RAW scores (c > 0), p = exp2(fma(s, c, -max c)): every operation is explicit (one
  // multiply, one fma, no contraction left to the compiler), so the fused kernel of qkv_attn.hip -- which repeats exactly this
  // sequence on the same half q / k / v -- gives the same bits
  const float c = a.scale * 1.4426950408889634f;
  float mx = NEG_BIG;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if ((4 * g + r) >= a.L) st[r] = NEG_BIG;
    mx = fmaxf(mx, st[r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float nm = -mx * c;
  float ls = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    st[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c, nm));
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
    // V^T fragment: lane = (d index 16 d + fl, keys 4g .. 4g+3) out of the row-major patch through the hardware transpose
    // read: lane i of a 16-lane group supplies the address of 4 d-values of key (i >> 2) and receives the 4 keys of
    // d-column i (one ds_read_b64_tr_b16 instead of four 2-byte reads + packing).  Pad columns (>= HD) of the last
    // fragment read the next row's first bytes / the slack row: finite, and only feed unused O rows.
    const u32x2 vf = lds_tr16<DT>(v_lds + (4 * g + (fl >> 2)) * VP + (fl & 3) * 8 + d * 32);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = mfma_k16<DT>(vf, pb, acc);  // O^T[d = 16 d + 4g + r][q = fl]
    const int dd = 16 * d + 4 * g;
    if (active && fl < a.L && dd < HD) {
      store_out4<DT, LO8>(a, orow + dd, acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
    }
  }
}

}  // namespace

int launch_attention(const AttnArgs& a_in, int dtype, hipStream_t st) {
  const AttnArgs& a0 = a_in;
  if (a0.hd != 64 && a0.hd != 72) return fail(LATTE_ERR_INVALID, "attention: head_dim must be 64 or 72");
  if (a0.L <= 0) return fail(LATTE_ERR_INVALID, "attention: empty sequence");
  const bool small = a0.L <= 16;
  AttnArgs a = a_in;
  // kernel-choice override of the A/B tests (latte_debug_set_choice("attn_variant", v), include/latte_amd_debug.h): 1 = the generic
  // flash kernel for every L > 16, 5 = the streaming kernel for 128 < L <= 256 as well.  Every choice computes the same function;
  // the measurement ablations that do not (7-9: no softmax / no waits / no staging) exist in the LATTE_DEBUG_BUILD library only.
  if (const int ov = debug_choice(DBG_ATTN_VARIANT)) a.variant = ov;
  const bool full = a.L > 128 && a.L <= 256 && a.variant != 1 && a.variant != 5;
  // L > 256: the streaming kernel (8 waves, 256 queries, ring of 128-key blocks); its per-lane source offsets are 32-bit: 128 rows
  // of a block must span less than 2 GiB, otherwise the generic flash kernel runs
  const bool stream_ok = (int64_t)a.row_stride * 3 * a.D * 2 * 128 < (1ll << 31);
  const bool stream = stream_ok && ((a.L > 256 && a.variant != 1) || (a.L > 128 && a.variant == 5));
  constexpr int FULL_LDS = 2 * 256 * 160, STREAM_LDS = 3 * 2 * 128 * 160;
  dim3 block(stream ? 512 : 256);
  dim3 grid = small ? dim3((a.num_seq * a.heads + 3) / 4)
                    : (full && !stream ? dim3(a.num_seq * a.heads)
                            : (stream ? dim3(a.num_seq * a.heads * ((a.L + 255) / 256)) : dim3(a.num_seq * a.heads * ((a.L + 63) / 64))));
#ifdef LATTE_GEMM_ABLATE
#define ATTN_STREAM_ABLATIONS(HD, DT)                                                                         \
      if (((a.variant >= 7 && a.variant <= 10) || a.variant == 16) && HD == 72 && DT == LATTE_DTYPE_F16) {                            \
        static std::atomic<uint64_t> attr_done_a[5];                                                          \
        const void* fn_ = a.variant == 7 ? (const void*)attn_stream_kernel<72, LATTE_DTYPE_F16, 7>            \
                          : a.variant == 8 ? (const void*)attn_stream_kernel<72, LATTE_DTYPE_F16, 8>          \
                          : a.variant == 9 ? (const void*)attn_stream_kernel<72, LATTE_DTYPE_F16, 9>          \
                          : a.variant == 10 ? (const void*)attn_stream_kernel<72, LATTE_DTYPE_F16, 10>        \
                                           : (const void*)attn_stream_kernel<72, LATTE_DTYPE_F16, 11>;        \
        if (int rc_ = ensure_dynamic_lds(fn_, STREAM_LDS, attr_done_a[a.variant == 16 ? 4 : a.variant - 7])) return rc_;            \
        void* args_[] = {(void*)&a};                                                                          \
        LATTE_HIP(hipLaunchKernel(fn_, grid, block, args_, STREAM_LDS, st));                                  \
      } else
#define ATTN_STREAM64_ABLATIONS(DT)                                                                           \
      if (a.variant >= 17 && a.variant <= 19 && DT == LATTE_DTYPE_F16) {                                      \
        static std::atomic<uint64_t> attr_done_b[3];                                                          \
        const void* fn_ = a.variant == 17 ? (const void*)attn_stream64_kernel<72, LATTE_DTYPE_F16, false, 7>  \
                          : a.variant == 18 ? (const void*)attn_stream64_kernel<72, LATTE_DTYPE_F16, false, 8>\
                                            : (const void*)attn_stream64_kernel<72, LATTE_DTYPE_F16, false, 9>;\
        if (int rc_ = ensure_dynamic_lds(fn_, STREAM64_LDS, attr_done_b[a.variant - 17])) return rc_;         \
        void* args_[] = {(void*)&a};                                                                          \
        LATTE_HIP(hipLaunchKernel(fn_, grid, dim3(256), args_, STREAM64_LDS, st));                            \
      } else
#else
#define ATTN_STREAM_ABLATIONS(HD, DT)
#define ATTN_STREAM64_ABLATIONS(DT)
#endif
#define ATTN_LAUNCH(HD, DT)                                                                                   \
  do {                                                                                                        \
    if (small)                                                                                                \
      hipLaunchKernelGGL((attn_small_kernel<HD, DT>), grid, block, 0, st, a);                                 \
    else if (stream) {                                                                                        \
      static std::atomic<uint64_t> attr_done_s{0};                                                            \
      if (int rc_ = ensure_dynamic_lds((const void*)attn_stream_kernel<HD, DT>, STREAM_LDS, attr_done_s)) return rc_; \
      ATTN_STREAM_ABLATIONS(HD, DT)                                                                           \
      ATTN_STREAM64_ABLATIONS(DT)                                                                             \
      if (HD == 72 && a.variant == 13) {                                                                      \
        static std::atomic<uint64_t> attr_done_s64b{0};                                                       \
        if (int rc_ = ensure_dynamic_lds((const void*)attn_stream64_kernel<72, DT, false, 0, 1>, STREAM64_LDS, attr_done_s64b)) return rc_; \
        hipLaunchKernelGGL((attn_stream64_kernel<72, DT, false, 0, 1>), grid, dim3(512), STREAM64_LDS, st, a); \
      } else if (HD == 72 && a.variant == 12) {                                                               \
        static std::atomic<uint64_t> attr_done_s64{0};                                                        \
        if (int rc_ = ensure_dynamic_lds((const void*)attn_stream64_kernel<72, DT>, STREAM64_LDS, attr_done_s64)) return rc_; \
        hipLaunchKernelGGL((attn_stream64_kernel<72, DT>), grid, dim3(256), STREAM64_LDS, st, a);             \
      } else                                                                                                  \
      hipLaunchKernelGGL((attn_stream_kernel<HD, DT>), grid, block, STREAM_LDS, st, a);                       \
    } else if (full) {                                                                                          \
      static std::atomic<uint64_t> attr_done{0};                                                              \
      if (int rc_ = ensure_dynamic_lds((const void*)attn_full_kernel<HD, DT>, FULL_LDS, attr_done)) return rc_; \
      hipLaunchKernelGGL((attn_full_kernel<HD, DT>), grid, block, FULL_LDS, st, a);                           \
    } else                                                                                                    \
      hipLaunchKernelGGL((attn_flash_kernel<HD, DT>), grid, block, 0, st, a);                                 \
  } while (0)
  if (a.out8) {   // f16 + fp8 remainder output (guided calls; AttnArgs::out8)
    if (dtype != LATTE_DTYPE_F16) return fail(LATTE_ERR_INVALID, "attention: the fp8-remainder output is f16 only");
#define ATTN_LAUNCH8(HD)                                                                                                        \
  do {                                                                                                                        \
    if (small)                                                                                                                \
      hipLaunchKernelGGL((attn_small_kernel<HD, LATTE_DTYPE_F16, true>), grid, block, 0, st, a);                              \
    else if (stream) {                                                                                                        \
      static std::atomic<uint64_t> attr_done_s8{0};                                                                           \
      if (int rc_ = ensure_dynamic_lds((const void*)attn_stream_kernel<HD, LATTE_DTYPE_F16, 0, true>, STREAM_LDS, attr_done_s8)) return rc_; \
      hipLaunchKernelGGL((attn_stream_kernel<HD, LATTE_DTYPE_F16, 0, true>), grid, block, STREAM_LDS, st, a);                 \
    } else if (full) {                                                                                                        \
      static std::atomic<uint64_t> attr_done8{0};                                                                             \
      if (int rc_ = ensure_dynamic_lds((const void*)attn_full_kernel<HD, LATTE_DTYPE_F16, true>, FULL_LDS, attr_done8)) return rc_; \
      hipLaunchKernelGGL((attn_full_kernel<HD, LATTE_DTYPE_F16, true>), grid, block, FULL_LDS, st, a);                        \
    } else                                                                                                                    \
      hipLaunchKernelGGL((attn_flash_kernel<HD, LATTE_DTYPE_F16, false, true>), grid, block, 0, st, a);                       \
  } while (0)
    if (a.hd == 64) ATTN_LAUNCH8(64); else ATTN_LAUNCH8(72);
#undef ATTN_LAUNCH8
    LATTE_HIP(hipGetLastError());
    return LATTE_OK;
  }
  if (dtype == LATTE_DTYPE_BF16) {
    if (a.hd == 64) ATTN_LAUNCH(64, LATTE_DTYPE_BF16); else ATTN_LAUNCH(72, LATTE_DTYPE_BF16);
  } else if (dtype == LATTE_DTYPE_F16) {
    if (a.hd == 64) ATTN_LAUNCH(64, LATTE_DTYPE_F16); else ATTN_LAUNCH(72, LATTE_DTYPE_F16);
  } else {
    return fail(LATTE_ERR_INVALID, "attention: unknown dtype");
  }
#undef ATTN_LAUNCH
#undef ATTN_STREAM_ABLATIONS
#undef ATTN_STREAM64_ABLATIONS
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_cross_attention(const AttnArgs& a, int dtype, hipStream_t st) {
  if (a.hd != 64 && a.hd != 72) return fail(LATTE_ERR_INVALID, "cross attention: head_dim must be 64 or 72");
  if (a.L <= 0 || a.Lk <= 0 || !a.kv || a.q_ld < a.D) return fail(LATTE_ERR_INVALID, "cross attention: bad arguments");
  // Lk <= 128 keys (Latte-1: 120) and at least half a 256-query block per sequence: the whole-panel kernel; latte_debug_set_choice("xattn_flash", 1)
  // keeps the generic flash kernel (tests, A/B)
  const bool panel = a.Lk <= 128 && a.L >= 128 && debug_choice(DBG_XATTN_FLASH) != 1;
  constexpr int CROSS_LDS = 2 * 128 * 160 + 128 * 4;
  dim3 block(panel ? 512 : 256), grid(panel ? a.num_seq * a.heads * ((a.L + 255) / 256) : a.num_seq * a.heads * ((a.L + 63) / 64));
#define XATTN_LAUNCH(HD, DT)                                                                                  \
  do {                                                                                                        \
    if (panel) {                                                                                              \
      static std::atomic<uint64_t> attr_done_x{0};                                                            \
      if (int rc_ = ensure_dynamic_lds((const void*)attn_cross_kernel<HD, DT>, CROSS_LDS, attr_done_x)) return rc_; \
      hipLaunchKernelGGL((attn_cross_kernel<HD, DT>), grid, block, CROSS_LDS, st, a);                         \
    } else                                                                                                    \
      hipLaunchKernelGGL((attn_flash_kernel<HD, DT, true>), grid, block, 0, st, a);                           \
  } while (0)
  if (dtype == LATTE_DTYPE_BF16) {
    if (a.hd == 64) XATTN_LAUNCH(64, LATTE_DTYPE_BF16); else XATTN_LAUNCH(72, LATTE_DTYPE_BF16);
  } else if (dtype == LATTE_DTYPE_F16) {
    if (a.hd == 64) XATTN_LAUNCH(64, LATTE_DTYPE_F16); else XATTN_LAUNCH(72, LATTE_DTYPE_F16);
  } else {
    return fail(LATTE_ERR_INVALID, "cross attention: unknown dtype");
  }
#undef XATTN_LAUNCH
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

}  // namespace latte
