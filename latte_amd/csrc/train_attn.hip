// Attention backward on gfx950 MFMA (training step, round 2): dQ, dK, dV of  O = softmax(Q K^T * hd^-0.5) V  (latte.py:48-77)
// for the factored spatial / temporal attention, straight on the row-major [rows, 3 D] qkv layout of the forward.
//
// Recompute design (nothing but O is kept from the forward; the score matrix is rebuilt tile by tile):
//   D_q   = sum_d dO[q, d] O[q, d]
//   P     = exp2(c S - m_q) / l_q                    (m_q, l_q: row max / sum of the exp2-domain scores, recomputed in pass A)
//   dP    = dO V^T,   dS = P (dP - D_q) * scale
//   dQ    = dS K,     dK = dS^T Q,     dV = P^T dO
// Two MFMA patterns, the two of the forward flash kernel (attention.hip):
//   "score" product  T^T[tile row][own row] = Y_tile . X_own^T : tile rows from a row-major LDS image (b128 reads), own rows as
//                    register fragments loaded from global memory; a lane ends with 4 tile rows x 1 own row;
//   "value" product  Out^T[d][own row] += Y_tile^T . W^T       : W = the lane's packed 4 x 1 values, Y^T fragments through the
//                    hardware transpose read of the same row-major image.
// Pass A (own rows = 16 queries per wave, tiles = 64 keys): sweep 1 row statistics, sweep 2  S, dP -> dS -> dQ; writes
//   stats[(seq, head, q)] = {m, 1 / l, D}.
// Pass B (own rows = 16 keys per wave, tiles = 64 queries): S^T and dP^T by the same score products with the roles swapped
//   (images Q and dO, fragments K and V), P and dS from the per-query statistics of pass A (staged in LDS) -> dV, dK.
#include <cstdlib>

#include "common.h"

namespace latte {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8b;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8b;
typedef __attribute__((ext_vector_type(4))) float f32x4b;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4b;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2b;

template <int DT>
__device__ __forceinline__ f32x4b mfma32(u32x4b a, u32x4b b, f32x4b c) {
  if constexpr (DT == LATTE_DTYPE_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8b, a), __builtin_bit_cast(bf16x8b, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8b, a), __builtin_bit_cast(f16x8b, b), c, 0, 0, 0);
}
template <int DT>
__device__ __forceinline__ unsigned int pk2(float lo, float hi) {
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
template <int DT>
__device__ __forceinline__ float hf(unsigned short h) {
  if constexpr (DT == LATTE_DTYPE_BF16) return __builtin_bit_cast(float, (unsigned int)h << 16);
  else return (float)__builtin_bit_cast(_Float16, h);
}
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short i16v4b;
__device__ __forceinline__ u32x2b tr16(const char* p) {
  i16v4b v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16v4b*)p);
  return __builtin_bit_cast(u32x2b, v);
}

constexpr float NEG_BIG_B = -1.0e30f;
constexpr int RPB = 160;   // row pitch of every LDS image (bytes): conflict-free for the b128 row reads and the transpose reads

struct AttnBwdArgs {
  const half_t* qkv;   // [rows, 3 D]
  const half_t* o;     // [rows, D]   forward output
  const half_t* dout;  // [rows, D]   gradient of the forward output
  half_t* dqkv;        // [rows, 3 D] result
  float* stats;        // [num_seq * heads * L][3] = {m, 1 / l, D}
  int num_seq, L, heads, hd, D, U;
  int64_t sample_stride, seq_stride, row_stride;
  float scale;
  int force_tiles;     // test / measurement hook: 1 = the two tile kernels also for L <= 16, 2 = also for 64 < L <= 256 (no resident images)
};

__device__ __forceinline__ int64_t seq_base(const AttnBwdArgs& a, int seq) {
  return (int64_t)(seq / a.U) * a.sample_stride + (int64_t)(seq % a.U) * a.seq_stride;
}

// stage 64 rows (tile `tile` of a sequence) of a [rows, ld] matrix, columns [col0, col0 + HD), into a row-major LDS image
template <int HD, int NT = 256>
__device__ __forceinline__ void stage_tile(char* img, const half_t* src, size_t ld, int col0, int64_t base, int64_t row_stride, int tile,
                                           int L, int tid) {
  constexpr int NCH = HD / 8;
  for (int id = tid; id < 64 * NCH; id += NT) {
    const int r = id / NCH, ch = id % NCH;
    const int rl = min(tile * 64 + r, L - 1);
    *(u32x4b*)(img + r * RPB + ch * 16) = *(const u32x4b*)(src + (size_t)(base + (int64_t)rl * row_stride) * ld + col0 + ch * 8);
  }
}

// all (<= 4) tiles of TWO images at once: every global load of the workgroup is issued before the first LDS store (a tile-by-tile
// walk waits for one load latency per tile and image -- 8 in a row with nothing else resident on the CU)
template <int HD, int NT>
__device__ __forceinline__ void stage_pair_all(char* img_a, const half_t* src_a, size_t ld_a, int col_a, char* img_b, const half_t* src_b,
                                               size_t ld_b, int col_b, int64_t base, int64_t row_stride, int tiles, int L, int tid) {
  constexpr int NCH = HD / 8, IT = (64 * NCH + NT - 1) / NT, TIMG = 64 * RPB;
  u32x4b ra[4][IT], rb[4][IT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int id = tid + it * NT;
      if (t < tiles && id < 64 * NCH) {
        const int r = id / NCH, ch = id % NCH;
        const int64_t row = base + (int64_t)min(t * 64 + r, L - 1) * row_stride;
        ra[t][it] = *(const u32x4b*)(src_a + (size_t)row * ld_a + col_a + ch * 8);
        rb[t][it] = *(const u32x4b*)(src_b + (size_t)row * ld_b + col_b + ch * 8);
      }
    }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int id = tid + it * NT;
      if (t < tiles && id < 64 * NCH) {
        const int r = id / NCH, ch = id % NCH;
        *(u32x4b*)(img_a + t * TIMG + r * RPB + ch * 16) = ra[t][it];
        *(u32x4b*)(img_b + t * TIMG + r * RPB + ch * 16) = rb[t][it];
      }
    }
}

// T^T[tile row 16 j + 4 g + r][own row fl] for the 64 tile rows: st[j][r]
template <int HD, int DT>
__device__ __forceinline__ void score_product(const char* img, const u32x4b* own, f32x4b* st, int fl, int g) {
  constexpr int KS = (HD + 31) / 32, NCH = HD / 8;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    st[j] = (f32x4b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ch = g + 4 * ks;
      u32x4b f = *(const u32x4b*)(img + (16 * j + fl) * RPB + ch * 16);
      if (ch >= NCH) f = (u32x4b){0u, 0u, 0u, 0u};
      st[j] = mfma32<DT>(f, own[ks], st[j]);
    }
  }
}
// Out^T[d][own row] += sum over the 64 tile rows of img[row][d] * w[row][own row];  w = st-layout values
template <int HD, int DT>
__device__ __forceinline__ void value_product(const char* img, const f32x4b* w, f32x4b* out, int fl, int g) {
  constexpr int DF = (HD + 15) / 16;
#pragma unroll
  for (int ks2 = 0; ks2 < 2; ++ks2) {
    const u32x4b pb = {pk2<DT>(w[2 * ks2][0], w[2 * ks2][1]), pk2<DT>(w[2 * ks2][2], w[2 * ks2][3]),
                       pk2<DT>(w[2 * ks2 + 1][0], w[2 * ks2 + 1][1]), pk2<DT>(w[2 * ks2 + 1][2], w[2 * ks2 + 1][3])};
#pragma unroll
    for (int d = 0; d < DF; ++d) {
      const char* vb = img + (32 * ks2 + 4 * g + (fl >> 2)) * RPB + (fl & 3) * 8 + d * 32;
      const u32x2b lo = tr16(vb);
      const u32x2b hi = tr16(vb + 16 * RPB);
      out[d] = mfma32<DT>((u32x4b){lo[0], lo[1], hi[0], hi[1]}, pb, out[d]);
    }
  }
}
// own-row fragments of a [rows, ld] matrix: lane = (row fl, chunk g + 4 ks)
template <int HD>
__device__ __forceinline__ void load_own(u32x4b* f, const half_t* rowp, int g) {
  constexpr int KS = (HD + 31) / 32, NCH = HD / 8;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int ch = g + 4 * ks;
    f[ks] = (u32x4b){0u, 0u, 0u, 0u};
    if (ch < NCH) f[ks] = *(const u32x4b*)(rowp + ch * 8);
  }
}
template <int HD, int DT>
__device__ __forceinline__ void store_own(half_t* rowp, const f32x4b* o, float mul, int g) {
  constexpr int DF = (HD + 15) / 16;
#pragma unroll
  for (int d = 0; d < DF; ++d) {
    const int dd = 16 * d + 4 * g;
    if (dd < HD) {
      const u32x2b p = {pk2<DT>(o[d][0] * mul, o[d][1] * mul), pk2<DT>(o[d][2] * mul, o[d][3] * mul)};
      *(u32x2b*)(rowp + dd) = p;
    }
  }
}

// ------------------------------------------------------------------------------------------------ pass A: statistics and dQ
template <int HD, int DT>
__global__ void __launch_bounds__(256) attn_bwd_q_kernel(AttnBwdArgs a) {
  constexpr int KS = (HD + 31) / 32, DF = (HD + 15) / 16, NCH = HD / 8;
  __shared__ __attribute__((aligned(16))) char lds[2 * 64 * RPB + 16 * RPB];   // (+ slack rows: pad d-columns read past row 63)
  char* const k_img = lds;
  char* const v_img = lds + 64 * RPB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fl = lane & 15, g = lane >> 4;
  const int tiles = (a.L + 63) >> 6;
  const int qt = blockIdx.x % tiles;
  const int head = (blockIdx.x / tiles) % a.heads;
  const int seq = blockIdx.x / (tiles * a.heads);
  const int64_t base = seq_base(a, seq);
  const size_t ld3 = (size_t)3 * a.D;
  const int q_idx = qt * 64 + wave * 16 + fl;
  const int q_ld = min(q_idx, a.L - 1);
  const int64_t q_row = base + (int64_t)q_ld * a.row_stride;
  u32x4b qf[KS], dof[KS];
  load_own<HD>(qf, a.qkv + (size_t)q_row * ld3 + head * HD, g);
  load_own<HD>(dof, a.dout + (size_t)q_row * a.D + head * HD, g);
  // D_q = dO . O over the head's columns: this lane's chunks, then the 4 lanes of the query
  float dq_dot = 0.f;
  {
    const half_t* orow = a.o + (size_t)q_row * a.D + head * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ch = g + 4 * ks;
      if (ch < NCH) {
        const u32x4b ov = *(const u32x4b*)(orow + ch * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dq_dot += hf<DT>((unsigned short)(ov[e] & 0xffffu)) * hf<DT>((unsigned short)(dof[ks][e] & 0xffffu));
          dq_dot += hf<DT>((unsigned short)(ov[e] >> 16)) * hf<DT>((unsigned short)(dof[ks][e] >> 16));
        }
      }
    }
    dq_dot += __shfl_xor(dq_dot, 16, 64);
    dq_dot += __shfl_xor(dq_dot, 32, 64);
  }
  const float c = a.scale * 1.4426950408889634f;
  // ---- sweep 1: row maximum and sum (exp2 domain)
  float m_run = NEG_BIG_B, l_run = 0.f;
  for (int kt = 0; kt < tiles; ++kt) {
    __syncthreads();
    stage_tile<HD>(k_img, a.qkv, ld3, a.D + head * HD, base, a.row_stride, kt, a.L, tid);
    __syncthreads();
    f32x4b st[4];
    score_product<HD, DT>(k_img, qf, st, fl, g);
    float mx = NEG_BIG_B;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 64 + 16 * j + 4 * g + r;
        const float z = key < a.L ? st[j][r] * c : NEG_BIG_B;
        st[j][r] = z;
        mx = fmaxf(mx, z);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    float ls = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) ls += __builtin_amdgcn_exp2f(st[j][r] - m_new);
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * __builtin_amdgcn_exp2f(m_run - m_new) + ls;
    m_run = m_new;
  }
  const float inv_l = 1.0f / l_run;
  if (q_idx < a.L && g == 0) {
    float* sp = a.stats + ((size_t)(seq * a.heads + head) * a.L + q_idx) * 3;
    sp[0] = m_run; sp[1] = inv_l; sp[2] = dq_dot;
  }
  // ---- sweep 2: dS and dQ
  f32x4b acc[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) acc[d] = (f32x4b){0.f, 0.f, 0.f, 0.f};
  for (int kt = 0; kt < tiles; ++kt) {
    __syncthreads();
    stage_tile<HD>(k_img, a.qkv, ld3, a.D + head * HD, base, a.row_stride, kt, a.L, tid);
    stage_tile<HD>(v_img, a.qkv, ld3, 2 * a.D + head * HD, base, a.row_stride, kt, a.L, tid);
    __syncthreads();
    f32x4b st[4], dp[4];
    score_product<HD, DT>(k_img, qf, st, fl, g);
    score_product<HD, DT>(v_img, dof, dp, fl, g);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 64 + 16 * j + 4 * g + r;
        const float p = key < a.L ? __builtin_amdgcn_exp2f(st[j][r] * c - m_run) * inv_l : 0.f;
        st[j][r] = p * (dp[j][r] - dq_dot) * a.scale;           // dS
      }
    value_product<HD, DT>(k_img, st, acc, fl, g);                // dQ^T += K^T dS^T
  }
  if (q_idx < a.L) store_own<HD, DT>(a.dqkv + (size_t)(base + (int64_t)q_idx * a.row_stride) * ld3 + head * HD, acc, 1.0f, g);
}

// ------------------------------------------------------------------------------------------------ pass B: dK and dV
template <int HD, int DT>
__global__ void __launch_bounds__(256) attn_bwd_kv_kernel(AttnBwdArgs a) {
  constexpr int KS = (HD + 31) / 32, DF = (HD + 15) / 16;
  __shared__ __attribute__((aligned(16))) char lds[2 * 64 * RPB + 16 * RPB];
  __shared__ float qstat[64][3];
  char* const q_img = lds;
  char* const do_img = lds + 64 * RPB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fl = lane & 15, g = lane >> 4;
  const int tiles = (a.L + 63) >> 6;
  const int kt = blockIdx.x % tiles;
  const int head = (blockIdx.x / tiles) % a.heads;
  const int seq = blockIdx.x / (tiles * a.heads);
  const int64_t base = seq_base(a, seq);
  const size_t ld3 = (size_t)3 * a.D;
  const int k_idx = kt * 64 + wave * 16 + fl;
  const int k_ld = min(k_idx, a.L - 1);
  const int64_t k_row = base + (int64_t)k_ld * a.row_stride;
  u32x4b kf[KS], vf[KS];
  load_own<HD>(kf, a.qkv + (size_t)k_row * ld3 + a.D + head * HD, g);
  load_own<HD>(vf, a.qkv + (size_t)k_row * ld3 + 2 * a.D + head * HD, g);
  const float c = a.scale * 1.4426950408889634f;
  f32x4b dv[DF], dk[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) {
    dv[d] = (f32x4b){0.f, 0.f, 0.f, 0.f};
    dk[d] = (f32x4b){0.f, 0.f, 0.f, 0.f};
  }
  const float* sbase = a.stats + (size_t)(seq * a.heads + head) * a.L * 3;
  for (int qt = 0; qt < tiles; ++qt) {
    __syncthreads();
    stage_tile<HD>(q_img, a.qkv, ld3, head * HD, base, a.row_stride, qt, a.L, tid);
    stage_tile<HD>(do_img, a.dout, (size_t)a.D, head * HD, base, a.row_stride, qt, a.L, tid);
    if (tid < 64) {
      const int q = qt * 64 + tid;
      const bool ok = q < a.L;
      qstat[tid][0] = ok ? sbase[(size_t)q * 3 + 0] : 0.f;
      qstat[tid][1] = ok ? sbase[(size_t)q * 3 + 1] : 0.f;     // 1 / l = 0 masks the query
      qstat[tid][2] = ok ? sbase[(size_t)q * 3 + 2] : 0.f;
    }
    __syncthreads();
    f32x4b st[4], dp[4];
    score_product<HD, DT>(q_img, kf, st, fl, g);      // S[q][k]: tile row = query, own row = key
    score_product<HD, DT>(do_img, vf, dp, fl, g);     // dP[q][k] = dO[q] . V[k]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = 16 * j + 4 * g + r;
        const float il = qstat[ql][1];
        const float p = il != 0.f ? __builtin_amdgcn_exp2f(st[j][r] * c - qstat[ql][0]) * il : 0.f;
        st[j][r] = p;
        dp[j][r] = p * (dp[j][r] - qstat[ql][2]) * a.scale;     // dS
      }
    value_product<HD, DT>(do_img, st, dv, fl, g);     // dV^T += dO^T P
    value_product<HD, DT>(q_img, dp, dk, fl, g);      // dK^T += Q^T dS
  }
  if (k_idx < a.L) {
    half_t* rowp = a.dqkv + (size_t)(base + (int64_t)k_idx * a.row_stride) * ld3 + head * HD;
    store_own<HD, DT>(rowp + a.D, dk, 1.0f, g);
    store_own<HD, DT>(rowp + 2 * a.D, dv, 1.0f, g);
  }
}


// ------------------------------------------------------------------------------------------------ 64 < L <= 256: resident images
// Round 6b.  The two tile passes above stage one 64-row tile, synchronise, multiply, synchronise -- per tile and sweep, through
// registers, nothing in flight under the MFMAs: at the spatial attention of the training step (L = 256, 960 (sequence, head)
// problems) they ran at 0.11 of the MFMA peak, 230 us per block.  A (sequence, head) of L <= 256 rows is 4 tiles: here ALL tiles of
// both images (K | V in pass A, Q | dO in pass B: 80 KB at pitch 160) are staged ONCE per workgroup, the workgroup is 8 waves = 128
// own rows (the launcher uses 16 waves: the whole sequence, 4 waves per SIMD), and the sweeps run over the resident tiles without a
// barrier.  Same MFMA products in the same order per own row; the elementwise chain between them is a shorter form of the same
// arithmetic (comment in pass A): equal to the tile passes to rounding (tests/test_gpu_kernels.py::test_attention_backward checks
// both against torch autograd and against each other).
template <int HD, int DT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) attn_bwd_q_res_kernel(AttnBwdArgs a) {
  constexpr int KS = (HD + 31) / 32, DF = (HD + 15) / 16, NCH = HD / 8, NT = WAVES * 64, QB = WAVES * 16, TIMG = 64 * RPB;
  extern __shared__ __attribute__((aligned(16))) char lds_res[];   // K tiles | V tiles | 16 slack rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fl = lane & 15, g = lane >> 4;
  const int tiles = (a.L + 63) >> 6;
  const int qblocks = (a.L + QB - 1) / QB;
  char* const k_img = lds_res;
  char* const v_img = lds_res + tiles * TIMG;
  const int qb = blockIdx.x % qblocks;
  const int head = (blockIdx.x / qblocks) % a.heads;
  const int seq = blockIdx.x / (qblocks * a.heads);
  const int64_t base = seq_base(a, seq);
  const size_t ld3 = (size_t)3 * a.D;
  stage_pair_all<HD, NT>(k_img, a.qkv, ld3, a.D + head * HD, v_img, a.qkv, ld3, 2 * a.D + head * HD, base, a.row_stride, tiles, a.L, tid);
  const int q_idx = qb * QB + wave * 16 + fl;
  const int q_ld = min(q_idx, a.L - 1);
  const int64_t q_row = base + (int64_t)q_ld * a.row_stride;
  u32x4b qf[KS], dof[KS];
  load_own<HD>(qf, a.qkv + (size_t)q_row * ld3 + head * HD, g);
  load_own<HD>(dof, a.dout + (size_t)q_row * a.D + head * HD, g);
  float dq_dot = 0.f;
  {
    const half_t* orow = a.o + (size_t)q_row * a.D + head * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ch = g + 4 * ks;
      if (ch < NCH) {
        const u32x4b ov = *(const u32x4b*)(orow + ch * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dq_dot += hf<DT>((unsigned short)(ov[e] & 0xffffu)) * hf<DT>((unsigned short)(dof[ks][e] & 0xffffu));
          dq_dot += hf<DT>((unsigned short)(ov[e] >> 16)) * hf<DT>((unsigned short)(dof[ks][e] >> 16));
        }
      }
    }
    dq_dot += __shfl_xor(dq_dot, 16, 64);
    dq_dot += __shfl_xor(dq_dot, 32, 64);
  }
  __syncthreads();
  if (qb * QB + wave * 16 >= a.L) return;   // (wave-uniform; no barrier below)
  // The passes are VALU-bound (16 own rows per wave: 16 scores per lane and tile against 8 - 24 MFMAs), so the elementwise chain is
  // kept short: the running maximum is taken on the RAW scores (the scale is positive), exp2 takes one FMA as its argument, the
  // normaliser and the softmax scale enter as log2 terms of that argument, key masks only exist on a ragged last tile.
  const float c = a.scale * 1.4426950408889634f;
  float m_raw = NEG_BIG_B, l_run = 0.f;   // running maximum of the raw scores; row sum in the exp2 domain relative to it
  for (int kt = 0; kt < tiles; ++kt) {
    f32x4b st[4];
    score_product<HD, DT>(k_img + kt * TIMG, qf, st, fl, g);
    if (kt * 64 + 64 > a.L) {   // ragged last tile (wave-uniform)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kt * 64 + 16 * j + 4 * g + r >= a.L) st[j][r] = NEG_BIG_B;
    }
    float mx = NEG_BIG_B;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[j][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_raw, mx);
    const float nm = -m_new * c;
    float ls = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) ls += __builtin_amdgcn_exp2f(__builtin_fmaf(st[j][r], c, nm));   // masked keys: exp2(-huge) = 0
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * __builtin_amdgcn_exp2f((m_raw - m_new) * c) + ls;
    m_raw = m_new;
  }
  const float m_run = m_raw * c;            // the statistics pass B reads: exp2-domain maximum, 1 / l, D
  const float inv_l = 1.0f / l_run;
  if (q_idx < a.L && g == 0) {
    float* sp = a.stats + ((size_t)(seq * a.heads + head) * a.L + q_idx) * 3;
    sp[0] = m_run; sp[1] = inv_l; sp[2] = dq_dot;
  }
  const float nm2 = __builtin_amdgcn_logf(inv_l * a.scale) - m_run;   // (v_log_f32 = log2) dS = exp2(S c + nm2) (dP - D)
  f32x4b acc[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) acc[d] = (f32x4b){0.f, 0.f, 0.f, 0.f};
  for (int kt = 0; kt < tiles; ++kt) {
    f32x4b st[4], dp[4];
    score_product<HD, DT>(k_img + kt * TIMG, qf, st, fl, g);
    score_product<HD, DT>(v_img + kt * TIMG, dof, dp, fl, g);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[j][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[j][r], c, nm2)) * (dp[j][r] - dq_dot);   // dS
    if (kt * 64 + 64 > a.L) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kt * 64 + 16 * j + 4 * g + r >= a.L) st[j][r] = 0.f;
    }
    value_product<HD, DT>(k_img + kt * TIMG, st, acc, fl, g);    // dQ^T += K^T dS^T
  }
  if (q_idx < a.L) store_own<HD, DT>(a.dqkv + (size_t)(base + (int64_t)q_idx * a.row_stride) * ld3 + head * HD, acc, 1.0f, g);
}

template <int HD, int DT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) attn_bwd_kv_res_kernel(AttnBwdArgs a) {
  constexpr int KS = (HD + 31) / 32, DF = (HD + 15) / 16, NT = WAVES * 64, QB = WAVES * 16, TIMG = 64 * RPB;
  extern __shared__ __attribute__((aligned(16))) char lds_res[];   // Q tiles | dO tiles | 16 slack rows | per-query statistics [64 tiles][3]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fl = lane & 15, g = lane >> 4;
  const int tiles = (a.L + 63) >> 6;
  const int kblocks = (a.L + QB - 1) / QB;
  char* const q_img = lds_res;
  char* const do_img = lds_res + tiles * TIMG;
  // per-query terms, one array each so that a lane's four consecutive queries are one 16-byte read:
  //   qnm[q] = log2(1 / l_q) - m_q  (-inf for an absent query: P = exp2(-inf) = 0),   qds[q] = D_q * scale
  float* const qnm = (float*)(lds_res + 2 * tiles * TIMG + 16 * RPB);
  float* const qds = qnm + tiles * 64;
  const int kb = blockIdx.x % kblocks;
  const int head = (blockIdx.x / kblocks) % a.heads;
  const int seq = blockIdx.x / (kblocks * a.heads);
  const int64_t base = seq_base(a, seq);
  const size_t ld3 = (size_t)3 * a.D;
  const float* sbase = a.stats + (size_t)(seq * a.heads + head) * a.L * 3;
  stage_pair_all<HD, NT>(q_img, a.qkv, ld3, head * HD, do_img, a.dout, (size_t)a.D, head * HD, base, a.row_stride, tiles, a.L, tid);
  for (int q = tid; q < tiles * 64; q += NT) {
    const bool ok = q < a.L;
    const float il = ok ? sbase[(size_t)q * 3 + 1] : 0.f;   // 1 / l = 0 masks the query
    qnm[q] = __builtin_amdgcn_logf(il) - (ok ? sbase[(size_t)q * 3 + 0] : 0.f);
    qds[q] = ok ? sbase[(size_t)q * 3 + 2] * a.scale : 0.f;
  }
  const int k_idx = kb * QB + wave * 16 + fl;
  const int k_ld = min(k_idx, a.L - 1);
  const int64_t k_row = base + (int64_t)k_ld * a.row_stride;
  u32x4b kf[KS], vf[KS];
  load_own<HD>(kf, a.qkv + (size_t)k_row * ld3 + a.D + head * HD, g);
  load_own<HD>(vf, a.qkv + (size_t)k_row * ld3 + 2 * a.D + head * HD, g);
  __syncthreads();
  if (kb * QB + wave * 16 >= a.L) return;   // (wave-uniform; no barrier below)
  const float c = a.scale * 1.4426950408889634f;
  f32x4b dv[DF], dk[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) {
    dv[d] = (f32x4b){0.f, 0.f, 0.f, 0.f};
    dk[d] = (f32x4b){0.f, 0.f, 0.f, 0.f};
  }
  for (int qt = 0; qt < tiles; ++qt) {
    f32x4b st[4], dp[4];
    score_product<HD, DT>(q_img + qt * TIMG, kf, st, fl, g);      // S[q][k]: tile row = query, own row = key
    score_product<HD, DT>(do_img + qt * TIMG, vf, dp, fl, g);     // dP[q][k] = dO[q] . V[k]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 nm4 = *(const float4*)(qnm + qt * 64 + 16 * j + 4 * g);
      const float4 ds4 = *(const float4*)(qds + qt * 64 + 16 * j + 4 * g);
      const float nmr[4] = {nm4.x, nm4.y, nm4.z, nm4.w}, dsr[4] = {ds4.x, ds4.y, ds4.z, ds4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[j][r], c, nmr[r]));   // P (0 for an absent query)
        st[j][r] = p;
        dp[j][r] = p * __builtin_fmaf(dp[j][r], a.scale, -dsr[r]);                      // dS = P (dP - D) scale
      }
    }
    value_product<HD, DT>(do_img + qt * TIMG, st, dv, fl, g);     // dV^T += dO^T P
    value_product<HD, DT>(q_img + qt * TIMG, dp, dk, fl, g);      // dK^T += Q^T dS
  }
  if (k_idx < a.L) {
    half_t* rowp = a.dqkv + (size_t)(base + (int64_t)k_idx * a.row_stride) * ld3 + head * HD;
    store_own<HD, DT>(rowp + a.D, dk, 1.0f, g);
    store_own<HD, DT>(rowp + 2 * a.D, dv, 1.0f, g);
  }
}

// ------------------------------------------------------------------------------------------------ L <= 16: one wave per (sequence, head)
// The temporal attention of the training step (16 frames per token; latte.py:355-368) gave the two tile kernels above one
// quarter-filled 64-row tile per workgroup -- 110 us per pass for 0.4 GFLOP.  Here a wave owns a whole (sequence, head) problem:
// every score-type product is one MFMA chain on row fragments loaded straight from global memory (both orientations: S^T for dQ,
// S for dK / dV, so that the softmax statistics -- per query -- are lane-local in the first and fetched by three lane shuffles per
// query in the second), the three value-type products (dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS) contract over the 16 tokens
// with the 16 x 16 x 16 MFMA on hardware-transposed reads of wave-private row-major LDS images of K, dO and Q.
template <int DT>
__device__ __forceinline__ f32x4b mfma16k(u32x2b a, u32x2b b, f32x4b c) {
  typedef __attribute__((ext_vector_type(4))) short s16x4b;
  typedef __attribute__((ext_vector_type(4))) _Float16 f16x4b;
  if constexpr (DT == LATTE_DTYPE_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4b, a), __builtin_bit_cast(s16x4b, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4b, a), __builtin_bit_cast(f16x4b, b), c, 0, 0, 0);
}

template <int HD, int DT>
__global__ void __launch_bounds__(256) attn_bwd_small_kernel(AttnBwdArgs a) {
  constexpr int KS = (HD + 31) / 32, DF = (HD + 15) / 16, NCH = HD / 8;
  constexpr int IMG = 17 * RPB;                                  // 16 rows + one row of slack (pad d-columns of the last fragment)
  __shared__ __attribute__((aligned(16))) char lds[4 * 3 * IMG];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fl = lane & 15, g = lane >> 4;
  const int items = a.num_seq * a.heads;
  int item = blockIdx.x * 4 + wave;
  const bool active = item < items;
  item = min(item, items - 1);
  const int seq = item / a.heads, head = item % a.heads;
  const int64_t base = seq_base(a, seq);
  const size_t ld3 = (size_t)3 * a.D;
  const int tok = min(fl, a.L - 1);
  const int64_t row = base + (int64_t)tok * a.row_stride;
  char* const k_img = lds + wave * 3 * IMG;
  char* const do_img = k_img + IMG;
  char* const q_img = k_img + 2 * IMG;

  // own-row fragments: lane = (token fl, chunk g + 4 ks); the same chunks go into the row-major images
  u32x4b qf[KS], kf[KS], vf[KS], dof[KS];
  const half_t* qrow = a.qkv + (size_t)row * ld3 + head * HD;
  load_own<HD>(qf, qrow, g);
  load_own<HD>(kf, qrow + a.D, g);
  load_own<HD>(vf, qrow + 2 * a.D, g);
  load_own<HD>(dof, a.dout + (size_t)row * a.D + head * HD, g);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int ch = g + 4 * ks;
    if (ch < NCH) {
      *(u32x4b*)(k_img + fl * RPB + ch * 16) = kf[ks];
      *(u32x4b*)(do_img + fl * RPB + ch * 16) = dof[ks];
      *(u32x4b*)(q_img + fl * RPB + ch * 16) = qf[ks];
    }
  }
  // D_q = dO[q] . O[q] for q = fl (this lane's chunks, then the 4 lanes of the token)
  float dsum = 0.f;
  {
    const half_t* orow = a.o + (size_t)row * a.D + head * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ch = g + 4 * ks;
      if (ch < NCH) {
        const u32x4b ov = *(const u32x4b*)(orow + ch * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dsum += hf<DT>((unsigned short)(ov[e] & 0xffffu)) * hf<DT>((unsigned short)(dof[ks][e] & 0xffffu));
          dsum += hf<DT>((unsigned short)(ov[e] >> 16)) * hf<DT>((unsigned short)(dof[ks][e] >> 16));
        }
      }
    }
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
  }
  const float c = a.scale * 1.4426950408889634f;
  // ---- orientation 1: S^T[key = 4 g + r][q = fl], dP^T likewise -> statistics of query fl, dS^T -> dQ
  f32x4b st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    st = mfma32<DT>(kf[ks], qf[ks], st);
    dpt = mfma32<DT>(vf[ks], dof[ks], dpt);
  }
  float mx = NEG_BIG_B;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (4 * g + r >= a.L) st[r] = NEG_BIG_B;
    mx = fmaxf(mx, st[r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float nm = -mx * c;
  float ls = 0.f;
  float pt[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    pt[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c, nm));   // masked keys: exp2(-huge) = 0
    ls += pt[r];
  }
  ls += __shfl_xor(ls, 16, 64);
  ls += __shfl_xor(ls, 32, 64);
  const float inv_l = 1.0f / ls;
  f32x4b dst;
#pragma unroll
  for (int r = 0; r < 4; ++r) dst[r] = pt[r] * inv_l * (dpt[r] - dsum) * a.scale;       // dS^T[key][q]
  __syncthreads();   // the images of this wave are complete (block-wide barrier keeps it simple)
  const u32x2b dsb = {pk2<DT>(dst[0], dst[1]), pk2<DT>(dst[2], dst[3])};                  // 4 keys x query fl
  const int tro = (4 * g + (fl >> 2)) * RPB + (fl & 3) * 8;                                // transpose-read lane offset inside an image
  const bool tok_ok = active && fl < a.L;
  half_t* drow = a.dqkv + (size_t)row * ld3 + head * HD;
#pragma unroll
  for (int d = 0; d < DF; ++d) {
    f32x4b o = {0.f, 0.f, 0.f, 0.f};
    o = mfma16k<DT>(tr16(k_img + tro + d * 32), dsb, o);                                   // dQ^T[d][q = fl]
    const int dd = 16 * d + 4 * g;
    if (tok_ok && dd < HD) *(u32x2b*)(drow + dd) = (u32x2b){pk2<DT>(o[0], o[1]), pk2<DT>(o[2], o[3])};
  }
  // ---- orientation 2: S[q = 4 g + r][key = fl], dP likewise; statistics of query 4 g + r from the lanes that own it
  f32x4b s2 = {0.f, 0.f, 0.f, 0.f}, dp2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    s2 = mfma32<DT>(qf[ks], kf[ks], s2);
    dp2 = mfma32<DT>(dof[ks], vf[ks], dp2);
  }
  f32x4b p2, ds2;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = 4 * g + r;                                   // lane q (g = 0) holds the statistics of query q
    const float nm_q = __shfl(nm, q, 64), il_q = __shfl(inv_l, q, 64), d_q = __shfl(dsum, q, 64);
    const bool ok = q < a.L && fl < a.L;                       // masked key columns / absent query rows contribute nothing
    const float p = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s2[r], c, nm_q)) * il_q : 0.f;
    p2[r] = p;
    ds2[r] = p * (dp2[r] - d_q) * a.scale;
  }
  const u32x2b pb = {pk2<DT>(p2[0], p2[1]), pk2<DT>(p2[2], p2[3])};                        // 4 queries x key fl
  const u32x2b dsb2 = {pk2<DT>(ds2[0], ds2[1]), pk2<DT>(ds2[2], ds2[3])};
#pragma unroll
  for (int d = 0; d < DF; ++d) {
    f32x4b ov = {0.f, 0.f, 0.f, 0.f}, ok_ = {0.f, 0.f, 0.f, 0.f};
    ov = mfma16k<DT>(tr16(do_img + tro + d * 32), pb, ov);                                 // dV^T[d][key = fl]
    ok_ = mfma16k<DT>(tr16(q_img + tro + d * 32), dsb2, ok_);                              // dK^T[d][key = fl]
    const int dd = 16 * d + 4 * g;
    if (tok_ok && dd < HD) {
      *(u32x2b*)(drow + a.D + dd) = (u32x2b){pk2<DT>(ok_[0], ok_[1]), pk2<DT>(ok_[2], ok_[3])};
      *(u32x2b*)(drow + 2 * a.D + dd) = (u32x2b){pk2<DT>(ov[0], ov[1]), pk2<DT>(ov[2], ov[3])};
    }
  }
}

template <int HD, int DT>
int launch_hd_dt(const AttnBwdArgs& a, hipStream_t st) {
  if (a.L <= 16 && a.force_tiles != 1) {
    hipLaunchKernelGGL((attn_bwd_small_kernel<HD, DT>), dim3((a.num_seq * a.heads + 3) / 4), dim3(256), 0, st, a);
    LATTE_HIP(hipGetLastError());
    return LATTE_OK;
  }
  const int tiles = (a.L + 63) / 64;
  if (a.L > 64 && a.L <= 256 && a.force_tiles != 2) {   // resident images, 8 waves = 128 own rows per workgroup (round 6b)
    constexpr int W = 16;
    const int blocks = (a.L + 16 * W - 1) / (16 * W);
    const int lds = 2 * tiles * 64 * RPB + 16 * RPB + tiles * 64 * 3 * (int)sizeof(float);
    static std::atomic<uint64_t> done_q{0}, done_kv{0};
    constexpr int lds_max = 2 * 4 * 64 * RPB + 16 * RPB + 4 * 64 * 3 * (int)sizeof(float);   // 4 tiles: the per-device opt-in covers every L
    if (int rc = ensure_dynamic_lds((const void*)attn_bwd_q_res_kernel<HD, DT, W>, lds_max, done_q)) return rc;
    if (int rc = ensure_dynamic_lds((const void*)attn_bwd_kv_res_kernel<HD, DT, W>, lds_max, done_kv)) return rc;
    dim3 gridr(a.num_seq * a.heads * blocks), blockr(W * 64);
    hipLaunchKernelGGL((attn_bwd_q_res_kernel<HD, DT, W>), gridr, blockr, lds, st, a);
    hipLaunchKernelGGL((attn_bwd_kv_res_kernel<HD, DT, W>), gridr, blockr, lds, st, a);
    LATTE_HIP(hipGetLastError());
    return LATTE_OK;
  }
  dim3 grid(a.num_seq * a.heads * tiles), block(256);
  hipLaunchKernelGGL((attn_bwd_q_kernel<HD, DT>), grid, block, 0, st, a);
  hipLaunchKernelGGL((attn_bwd_kv_kernel<HD, DT>), grid, block, 0, st, a);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

}  // namespace

// stats: float [num_seq * heads * L * 3] scratch
int launch_attention_bwd(const half_t* qkv, const half_t* o, const half_t* dout, half_t* dqkv, float* stats, int num_seq, int L, int heads,
                         int hd, int U, int64_t sample_stride, int64_t seq_stride, int64_t row_stride, int dtype, hipStream_t st) {
  AttnBwdArgs a{};
  a.qkv = qkv; a.o = o; a.dout = dout; a.dqkv = dqkv; a.stats = stats;
  a.num_seq = num_seq; a.L = L; a.heads = heads; a.hd = hd; a.D = heads * hd; a.U = U;
  a.sample_stride = sample_stride; a.seq_stride = seq_stride; a.row_stride = row_stride;
  a.scale = 1.0f / sqrtf((float)hd);
  // latte_debug_set_choice("attn_bwd_tiles", v) (A/B tests): 1 = the tile passes also for L <= 16, 2 = also for 64 < L <= 256
  a.force_tiles = debug_choice(DBG_ATTN_BWD_TILES);
  if (dtype != LATTE_DTYPE_BF16 && dtype != LATTE_DTYPE_F16) return fail(LATTE_ERR_INVALID, "attention_bwd: unknown dtype");
#define CASE(HD)                                                                                         \
  case HD:                                                                                               \
    return dtype == LATTE_DTYPE_BF16 ? launch_hd_dt<HD, LATTE_DTYPE_BF16>(a, st) : launch_hd_dt<HD, LATTE_DTYPE_F16>(a, st);
  switch (hd) {
    CASE(64)
    CASE(72)
    default:
      return fail(LATTE_ERR_INVALID, "attention_bwd: head dim must be 64 or 72");
  }
#undef CASE
}

}  // namespace latte
