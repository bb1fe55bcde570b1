// Fused QKV projection + attention core of one Latte block on gfx950:  out = softmax(q k^T * hd^-0.5) v  with
// [q | k | v] = xn W_qkv^T + b computed INSIDE the kernel, per (sequence, head).
//
// Replaces Attention.forward up to (not including) the output projection (latte.py:48-77: `qkv = self.qkv(x)` at :50, the
// reshape / permute copy at :50-51, `attn = (q @ k.transpose(-2, -1)) * self.scale` :67, softmax :68, `attn @ v` :70) for the
// two sequence shapes of every 256-pixel, 16-frame Latte config: spatial blocks (256 tokens of one frame) and temporal blocks
// (16 frames of one token, latte.py:355-368).  Un-fused, the qkv tensor makes a round trip through HBM (226 MB written and
// read again per block at XL/2, B = 8) and the stand-alone attention kernel is bound by exactly that traffic (arithmetic
// intensity 128 FLOP/B); here Q, K and V of a head exist only in LDS.
//
// One work unit = (sequence group, head):
//   spatial : the 256 tokens of one frame  -> rows  sg * 256 + r                       of the canonical [B, F, T, D] order
//   temporal: 16 neighbouring tokens x 16 frames -> tile row r = p * 16 + f  <->  row  b F T + f T + 16 tq + p
//             (16 sequences of 16 tokens; the rows of a sequence are the 16 consecutive tile rows 16 p .. 16 p + 15)
// and runs three phases on one 8-wave workgroup (one workgroup per CU, persistent over its units):
//   G  [256 x NPAD] = xn_tile[256 x D] W_h^T, NPAD = 3 hd rounded up to 32 (224 at hd = 72): the ping-pong MFMA schedule of
//      gemm.hip (two groups of 4 waves one barrier apart, operands staged by LDS DMA with the bank swizzle on the source
//      address, BK = 64, two stages), wave tile 64 x NPAD/2.  The W tile's rows are gathered from the three row blocks
//      q | k | v of the head (8-row DMA groups never straddle a block: hd % 8 == 0); the pad rows re-read rows of the q block
//      and their output columns are dropped.
//   E  accumulators + bias (staged in LDS one unit ahead) -> half -> three row-major LDS images Q, K, V [256][160 B] (they
//      overlay the operand stages).
//   A  spatial: every wave owns 32 queries against all 256 keys -- the register-resident exact softmax of attn_full_kernel
//      (attention.hip), V^T through ds_read_b64_tr_b16; temporal: every wave owns two 16-token sequences (attn_small_kernel's
//      arithmetic).  Outputs are written as in the un-fused kernels (row = token, column = head * hd + d).
// The half values of Q / K / V and every later operation are the same as on the un-fused path (same K order, same rounding
// points), so the two paths agree bit for bit (tests/test_gpu_kernels.py::test_fused_qkv_attention_*).
#include <type_traits>

#include "common.h"
#include "mfma_util.h"

namespace latte {
namespace {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short i16v4_t;
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr float NEG_BIG_F = -1.0e30f;

template <int DT>
__device__ __forceinline__ f32x4 mfma_k16h(u32x2 a, u32x2 b, f32x4 c) {
  if constexpr (DT == LATTE_DTYPE_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a), __builtin_bit_cast(s16x4_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, a), __builtin_bit_cast(f16x4_t, b), c, 0, 0, 0);
}
// hardware transpose read (see attention.hip): lane i of a 16-lane group supplies the address of 4 d-values of key (i >> 2) of a
// ROW-MAJOR image and receives the 4 keys of d-column i
__device__ __forceinline__ u32x2 tr16(const char* p) {
  i16v4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16v4_t*)p);
  return __builtin_bit_cast(u32x2, v);
}
// The same read as inline assembly: in front of the BUILTIN hipcc waits (s_waitcnt vmcnt) for any LDS DMA in flight, which in the
// temporal attention phase is the next unit's operand fill issued a few instructions earlier.  Completion: the caller's own
// s_waitcnt lgkmcnt(0) + sched_barrier (cdna_hip_programming.md section 5.7, form (iii)).
template <int OFF>
__device__ __forceinline__ u32x2 tr16_asm(const char* p) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)(const __attribute__((address_space(3))) char*)p), "i"(OFF));
  return v;
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)lds_wave_base, 16, voff, soff, 0, 0);
}

// Row pitches of the images: Q and K 144 B (9 x 16 B: b128 fragment reads of 16 consecutive rows spread over all banks, and the
// 8-byte image writes of the 16 rows of an accumulator fragment are 2-way instead of 4-way conflicted), V 160 B (the pitch at which
// the transpose reads of 8 rows x 32 B tile the 64 banks).  At hd = 72 a Q / K row has no pad: chunk 9 of the padded contraction
// is the next row's first chunk (real, finite data; it meets a zero Q chunk).
constexpr int RPQ = 144, RPV = 160;
[[maybe_unused]] constexpr int Q_OFF = 0;
constexpr int K_OFF = 256 * RPQ, V_OFF = 2 * 256 * RPQ, IMG_END = V_OFF + 256 * RPV;   // 0, 36864, 73728, 114688
// LDS map (bytes).  The images overlay the operand stages; stage 0's W tile sits BEHIND the images, so that all of stage 0 is
// dead memory as soon as every wave has fetched its Q fragments (the A tile of stage 0 lies inside the Q image):
//   [0, 36864) Q image        [36864, 73728) K image       [73728, 114688) V image
//   [0, 32768) A tile stage 0                               [61440, 94208) A tile stage 1   [94208, 122880) W tile stage 1
//   [122880, 151552) W tile stage 0     [151552, 152704) bias of the head (3 hd floats)
constexpr int A0_OFF = 0, A1_OFF = 61440, B1_OFF = 94208, B0_OFF = 122880, BIAS_OFF = B0_OFF + 224 * 128;
constexpr int FUSED_LDS = BIAS_OFF + 288 * 4;

// 4 x 4 transpose of one word per lane across the four 16-lane rows of a wave (gfx950: v_permlane32_swap exchanges the upper half of its
// first operand with the lower half of its second, v_permlane16_swap the odd rows of the first with the even rows of the second): on
// entry register d of lane row g holds word (d, g); on exit register k of lane row g holds word (g, k) -- the four words of ONE d-group
// side by side in one lane.  The fp8 remainder of the attention output (SPLIT == 2) uses it: a lane's four bytes per d-group become
// sixteen contiguous bytes per lane, one 16-byte store instead of four 4-byte ones on 16-byte row pieces (config 3: the attention
// kernels' out8 stores cost 60 - 80 us per launch as dword stores, profiles/r6_bench_steps20_after_fp8_remainder.json).
__device__ __forceinline__ void rows_transpose4(unsigned int& x0, unsigned int& x1, unsigned int& x2, unsigned int& x3) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u2;
  const u2 a = __builtin_amdgcn_permlane32_swap(x0, x2, false, false);
  const u2 b = __builtin_amdgcn_permlane32_swap(x1, x3, false, false);
  const u2 c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
  const u2 d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
  x0 = c[0]; x1 = c[1]; x2 = d[0]; x3 = d[1];
}

// FLAGS bit 0 (EARLY): the next unit's K tile 0 is DMA'd into stage 0 right after the attention phase has fetched its Q
//   fragments (one barrier, taken while the waves are still aligned), so the fill latency of the next unit's operand pipeline
//   hides under the attention phase.
// FLAGS bit 1 (PRIO): during the attention phase the waves of group 0 (one per SIMD) run at a higher issue priority than their
//   SIMD partners of group 1: the two fall out of phase (group 1's MFMA segments run under group 0's softmax VALU work and vice
//   versa) instead of competing for the same pipe in lock step.
// SPLIT: the output leaves as [hi | lo] pairs, row pitch 2 D (QkvAttnArgs::out_split) -- a template parameter, not a run-time branch: a
// branch on the kernel argument inside the attention phase changed the f16 instantiation's results (hipcc, ROCm 7.2; the bf16 one was
// unaffected), so the plain kernel stays byte for byte what rounds 3 / 4 measured and verified.
template <int HD, int DT, int MODE, int FLAGS, int SPLIT = 0>
__global__ void __launch_bounds__(512) qkv_attn_kernel(QkvAttnArgs a) {
  constexpr bool EARLY = (FLAGS & 1) != 0, PRIO = (FLAGS & 2) != 0;
  constexpr int NQ = 3 * HD;
  constexpr int NPAD = (NQ + 31) / 32 * 32;      // 224 | 192
  constexpr int FN = NPAD / 32;                  // 16-column fragments per wave: 7 | 6
  constexpr int B_MAIN = NPAD / 8 / 4;           // W-tile DMA groups (8 rows) per wave of group 0: 7 | 6
  constexpr int GPM = HD / 8;                    // DMA groups per q / k / v row block: 9 | 8
  constexpr int KS = (HD + 31) / 32, DF = (HD + 15) / 16, NCH = HD / 8;
  static_assert(NPAD <= 224 && NQ <= 288 && HD % 8 == 0 && A1_OFF >= 256 * 128 && B1_OFF - A1_OFF == 256 * 128 && B1_OFF + NPAD * 128 <= B0_OFF && IMG_END <= B0_OFF && K_OFF >= 256 * 128,
                "LDS plan");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
  const int D = a.D, T = a.T, F = a.F;
  const int out_ld = SPLIT == 1 ? 2 * D : D;   // row pitch of the output in halves
  const unsigned row_bytes = (unsigned)D * 2u;
  const int nk = D / 64;

  // ---- this workgroup's units: the heads of a sequence group stay on ONE XCD (shared xn panel in that L2; the head slices of
  // an output row share 128-byte lines)
  const int S = MODE == 0 ? a.B * F : a.B * (T >> 4);
  const bool xmap = (S & 7) == 0;
  // a.flags bit 2 (16 heads, even S): an XCD owns FOUR heads and every second sequence group instead of all heads of every
  // eighth one -- its 32 workgroups then share 4 W slices (2 MB, resident in the 4 MB L2 across rounds) and 8 xn panels per
  // round instead of 16 W slices (8 MB, re-streamed from the Infinity Cache every round) and 2 panels
  const bool hmap = (a.flags & 4) != 0 && a.heads == 16 && xmap;
  const int xcd = blockIdx.x & 7;
  int it = xmap ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int it_step = xmap ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int it_end = xmap ? (S >> 3) * a.heads : S * a.heads;

  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.xn, 0, (unsigned)((size_t)a.B * F * T) * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)(3 * D) * row_bytes, 0x00020000);

  // per-lane DMA offsets: lane = (row lrow of an 8-row group, 16-byte chunk cpos); the LDS image is lane-linear, so the bank
  // swizzle chunk ^= (tile_row >> 1) & 7 is applied to the SOURCE chunk and mirrored by the fragment reads.  A wave's groups
  // are gi = base + w4 + 4 j: (tile_row >> 1) & 7 = ((w4 & 1) * 4 + (lrow >> 1)) & 7 for every j.
  const int lrow = lane >> 3, cpos = lane & 7;
  const unsigned swz16 = (unsigned)((cpos ^ ((((w4 & 1) << 2) + (lrow >> 1)) & 7)) << 4);
  const unsigned voff_b = (unsigned)lrow * row_bytes + swz16;
  const unsigned voff_a = MODE == 0 ? voff_b : (unsigned)lrow * (unsigned)T * row_bytes + swz16;   // temporal: a group's rows are 8 frames
  const unsigned a_jstep = (MODE == 0 ? 32u : 2u) * row_bytes;

  // fragment read offsets inside a stage (row = 16 f + (lane & 15), chunk = (lane >> 4) + 4 ks, swizzled)
  const int fr = lane & 15, g = lane >> 4;
  const int chunkb = (g ^ ((lane >> 1) & 7)) << 4;
  const int a_off = (grp * 128 + wm * 64 + fr) * 128 + chunkb;
  const int b_off = (wn * (NPAD / 2) + fr) * 128 + chunkb;

  struct Unit {
    unsigned a_so0;            // A-row DMA offset of this wave's first group
    unsigned b_so[B_MAIN];     // W-row DMA offsets of this wave's groups (group 0's waves)
    int row_base, head;        // spatial: first row; temporal: b F T + 16 tq
  };
  auto decode = [&](int it_) -> Unit {
    Unit u;
    u.head = hmap ? ((xcd & 3) << 2) + (it_ & 3) : it_ % a.heads;
    const int sg = hmap ? ((it_ >> 2) << 1) + (xcd >> 2) : xmap ? (it_ / a.heads) * 8 + xcd : it_ / a.heads;
    if constexpr (MODE == 0) {
      u.row_base = sg * 256;
      u.a_so0 = (unsigned)(u.row_base + (grp * 16 + w4) * 8) * row_bytes;          // groups grp * 16 + w4 + 4 j
    } else {
      const int tqn = T >> 4;
      u.row_base = (sg / tqn) * F * T + (sg % tqn) * 16;
      // group gi = grp * 16 + w4 + 4 j: p = gi >> 1 = grp * 8 + (w4 >> 1) + 2 j, first frame (w4 & 1) * 8
      u.a_so0 = (unsigned)(u.row_base + (w4 & 1) * 8 * T + grp * 8 + (w4 >> 1)) * row_bytes;
    }
#pragma unroll
    for (int j = 0; j < B_MAIN; ++j) {   // group gi = w4 + 4 j -> rows of block gi / GPM
      const int gi = w4 + 4 * j, mat = gi / GPM, wi = gi - mat * GPM;
      // pad groups (columns >= 3 hd, hd = 72 only) re-read rows of the q block: their accumulator columns are never stored, so
      // any finite-or-not data will do, but the address must be valid (an SGPR offset is not part of the buffer bounds check)
      u.b_so[j] = (unsigned)((mat < 3 ? mat : 0) * D + u.head * HD + wi * 8) * row_bytes;
    }
    return u;
  };
  auto dma_a_half = [&](const Unit& u, int kt, int stg) {
    char* sA = smem + (stg ? A1_OFF : A0_OFF) + (grp * 16 + w4) * 1024;
    const unsigned so = u.a_so0 + (unsigned)kt * 128u;
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(rsA, sA + j * 4096, voff_a, so + (unsigned)j * a_jstep);
  };
  auto dma_b_all = [&](const Unit& u, int kt, int stg) {
    char* sB = smem + (stg ? B1_OFF : B0_OFF) + w4 * 1024;
#pragma unroll
    for (int j = 0; j < B_MAIN; ++j) dma16(rsB, sB + j * 4096, voff_b, u.b_so[j] + (unsigned)kt * 128u);
  };
  // K tile 0 of a unit into stage 0.  Nothing of the ping-pong stagger applies here (the fill is drained by every wave before
  // the barrier that opens the unit), so the W tile's groups are spread over all 8 waves: gi = wave + 8 j (same swizzle parity
  // as w4 + 4 j: wave & 1 == w4 & 1)
  auto fill_first = [&](const Unit& u) {
    dma_a_half(u, 0, 0);
    char* sB = smem + B0_OFF + wave * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gi = wave + 8 * j;
      if (gi < NPAD / 8) {
        const int mat = gi / GPM, wi = gi - mat * GPM;
        dma16(rsB, sB + j * 8192, voff_b, (unsigned)((mat < 3 ? mat : 0) * D + u.head * HD + wi * 8) * row_bytes);
      }
    }
  };

  // bias of a head (q | k | v: 3 hd floats) -> LDS, fetched one unit ahead so that the image-write phase never waits for HBM:
  // thread i < 3 hd / 4 carries floats 4 i .. 4 i + 3 of the NEXT unit's head in a register from the fetch point to the next
  // unit's first barrier, where the operand DMA is drained anyway
  const int bias_i = min((int)threadIdx.x, NQ / 4 - 1) * 4, bias_mat = bias_i / HD, bias_d = bias_i - bias_mat * HD;
  auto fetch_side = [&](const Unit& un) -> float4 { return *(const float4*)(a.bias + bias_mat * D + un.head * HD + bias_d); };

  if (it >= it_end) return;
  Unit cur = decode(it);
  fill_first(cur);
  float4 side = fetch_side(cur);
  // measurement hook (tools/fused_probe.py --trace): workgroup 0 sums shader-clock ticks per phase over its units
  const bool trace = a.dbg_trace != nullptr && blockIdx.x == 0;
  long long tacc[4] = {0, 0, 0, 0}, tprev = 0;
  if (trace) tprev = (long long)__builtin_readcyclecounter();
#define LATTE_PHASE(IDX)                                               \
  if (trace) {                                                         \
    const long long now_ = (long long)__builtin_readcyclecounter();    \
    tacc[IDX] += now_ - tprev;                                         \
    tprev = now_;                                                      \
  }
  for (;;) {
    const int head = cur.head, row_base = cur.row_base;
    const int it_next = it + it_step;
    const bool has_next = it_next < it_end;
    Unit nxt = cur;
    if (has_next) nxt = decode(it_next);
    bool filled_next = false;
    // hook of the attention phase, called right after a wave has fetched its Q fragments: once every wave has, stage 0 (A tile
    // inside the Q image, W tile behind the images) is dead memory and takes the next unit's K tile 0
    auto early_fill = [&]() {
      if constexpr (EARLY) {
        if (has_next) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          fill_first(nxt);
          side = fetch_side(nxt);
          filled_next = true;
        }
      }
    };
    // ================================================================ G: [256 x NPAD] = xn_tile W_h^T
    f32x4 acc[4][FN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x < NQ / 4) *(float4*)(smem + BIAS_OFF + threadIdx.x * 16) = side;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // K tile 0 has landed for everybody; the head's bias is in LDS
    dma_a_half(cur, 1, 1);
    if (grp == 0) dma_b_all(cur, 1, 1);
    if (grp == 1) __builtin_amdgcn_s_barrier();         // stagger the two groups by one segment
    // Per K tile u (stage u & 1): L(u) fragment reads, barrier, C(u) MFMAs, own DMA of u + 1 confirmed, barrier, then the
    // DMA of u + 2 into the stage just consumed (group 0 issues after the barrier that ends its C(u): by then group 1 has
    // finished L(u); group 1 only overwrites its own A rows).  Hand-offs as in gemm_pp_kernel / gemm_pps_kernel.
    // (Measured and removed, round 3: group 1 taking 3 of the 7 W-tile DMA groups at the START of its compute segment, to shorten
    //  group 0's [fragment reads + 11 DMA issues] segment -- 264 -> 284 us per launch: an LDS-DMA issue in front of a compute
    //  segment delays that segment's MFMAs by more than it saves the other group.)
    for (int kt = 0; kt < nk; ++kt) {
      const char* sbuf_a = smem + ((kt & 1) ? A1_OFF : A0_OFF);
      const char* sbuf_b = smem + ((kt & 1) ? B1_OFF : B0_OFF);
      u32x4 bf[2][FN], af[2][4];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[ks][j] = *(const u32x4*)(sbuf_b + ((b_off + j * 2048) ^ (ks << 6)));
#pragma unroll
        for (int i = 0; i < 4; ++i) af[ks][i] = *(const u32x4*)(sbuf_a + ((a_off + i * 2048) ^ (ks << 6)));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = mfma16<DT>(bf[ks][j], af[ks][i], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < nk) {
        dma_a_half(cur, kt + 2, kt & 1);
        if (grp == 0) dma_b_all(cur, kt + 2, kt & 1);
      }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();         // balance group 1's extra barrier: all stage reads are done
    LATTE_PHASE(0)

    // ================================================================ E: accumulators + bias -> half -> Q | K | V images
    {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n0 = wn * (NPAD / 2) + j * 16 + g * 4;     // 4 consecutive columns, never straddling a block (hd % 4 == 0)
        if (n0 < NQ) {
          const int mat = n0 >= 2 * HD ? 2 : (n0 >= HD ? 1 : 0);
          const int d = n0 - mat * HD;
          const float4 b4 = *(const float4*)(smem + BIAS_OFF + n0 * 4);     // q | k | v bias of the head, column n0 .. n0 + 3
          char* dst = smem + (mat == 2 ? V_OFF + (grp * 128 + wm * 64 + fr) * RPV : mat * K_OFF + (grp * 128 + wm * 64 + fr) * RPQ) + d * 2;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const u32x2 pk = {pack2<DT>(acc[i][j][0] + b4.x, acc[i][j][1] + b4.y), pack2<DT>(acc[i][j][2] + b4.z, acc[i][j][3] + b4.w)};
            *(u32x2*)(dst + i * 16 * (mat == 2 ? RPV : RPQ)) = pk;
          }
        }
      }
    }
    __syncthreads();
    LATTE_PHASE(1)

    if (a.dbg_qkv != nullptr) {   // test hook: the images as the [rows, 3 D] tensor the un-fused GEMM writes
      for (int idx = threadIdx.x; idx < 256 * 3 * NCH; idx += 512) {
        const int r = idx / (3 * NCH), rem = idx - r * (3 * NCH), mat = rem / NCH, ch = rem - mat * NCH;
        const int grow = MODE == 0 ? row_base + r : row_base + (r & 15) * T + (r >> 4);
        *(u32x4*)(a.dbg_qkv + (size_t)grow * (3 * D) + mat * D + head * HD + ch * 8) = *(const u32x4*)(smem + (mat == 2 ? V_OFF + r * RPV : mat * K_OFF + r * RPQ) + ch * 16);
      }
    }

    // ================================================================ A: attention on the LDS images
    const char* q_img = smem;
    const char* k_img = smem + K_OFF;
    const char* v_img = smem + V_OFF;
    const float c = a.scale * 1.4426950408889634f;   // softmax in the exp2 domain
    if constexpr (PRIO) {
      if (grp == 0) __builtin_amdgcn_s_setprio(2);
    }
    if constexpr (MODE == 0) {
      // every wave: 32 queries (two 16-query MFMA column groups sharing every K / V fragment read) x 256 keys;
      // attn_full_kernel's pass (attention.hip)
      constexpr int NKT = 16;
      const int q0 = wave * 32;
      const char* kbase = k_img + fr * RPQ + g * 16;
      const char* vbase = v_img + (4 * g + (fr >> 2)) * RPV + (fr & 3) * 8;
      constexpr int NG = 2;   // query groups of a wave, processed together
      {
        u32x4 qf[NG][KS];
#pragma unroll
        for (int gq = 0; gq < NG; ++gq)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const int ch = g + 4 * ks;
            qf[gq][ks] = *(const u32x4*)(q_img + (q0 + gq * 16 + fr) * RPQ + ch * 16);
            if (ch >= NCH) qf[gq][ks] = (u32x4){0u, 0u, 0u, 0u};
          }
        early_fill();   // the Q image is dead for this wave from here on
        // S^T[key][q] for all 256 keys; K fragments software-pipelined two key tiles ahead through four rotating register sets
        f32x4 st[NG][NKT];
        u32x4 kf[4][KS];
        auto load_k = [&](int kt, u32x4 (&dst)[KS]) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) dst[ks] = *(const u32x4*)(kbase + kt * 16 * RPQ + ks * 64);
        };
        load_k(0, kf[0]);
        load_k(1, kf[1]);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          if (kt + 2 < NKT) load_k(kt + 2, kf[(kt + 2) & 3]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int gq = 0; gq < NG; ++gq) st[gq][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) st[gq][kt] = mfma16<DT>(kf[kt & 3][ks], qf[gq][ks], st[gq][kt]);
          __builtin_amdgcn_sched_barrier(0);
        }
        // exact softmax over the key axis: in-lane over 64 values, then the 4 lanes g = 0..3 of a query.
        // max on the raw scores (c > 0), p = exp2(s * c - max * c): one max, one fma, one exp2, one add per element
        float inv[NG];
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
          float mx = NEG_BIG_F;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[gq][kt][r]);
          mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          const float nm = -mx * c;
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          const f32x2 c2 = {c, c}, nm2 = {nm, nm};
          f32x2 ls2 = {0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              const f32x2 e = __builtin_elementwise_fma((f32x2){st[gq][kt][r], st[gq][kt][r + 1]}, c2, nm2);
              const f32x2 pp = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
              st[gq][kt][r] = pp.x;
              st[gq][kt][r + 1] = pp.y;
              ls2 += pp;
            }
          float ls = ls2.x + ls2.y;
          ls += __shfl_xor(ls, 16, 64);
          ls += __shfl_xor(ls, 32, 64);
          inv[gq] = 1.0f / ls;
        }
        // O^T += V^T P^T ; k-slot (8g + i) <-> key 32 ks2 + (i < 4 ? 4g + i : 16 + 4g + i - 4)
        f32x4 o[NG][DF];
#pragma unroll
        for (int gq = 0; gq < NG; ++gq)
#pragma unroll
          for (int d = 0; d < DF; ++d) o[gq][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // (Round 4: the V^T reads of this phase were moved to the inline-assembly transpose read with hand-counted lgkmcnt, as in the
        //  temporal mode, so that hipcc's s_waitcnt vmcnt in front of the BUILTIN no longer waits for the next unit's early operand
        //  fill.  Double-buffered whole steps did not fit 256 VGPRs at hd = 72 (an asm read cannot target half of a 128-bit MFMA
        //  operand: 40 fragment registers + the copies); the single-buffered in-place form fitted and measured no gain -- spatial
        //  kernel 3.88 ms per forward against 3.78-3.84 with the builtin on the boxes of the same day -- so the builtin stays.)
        u32x4 vfr[2][DF];
        auto load_v = [&](int ks2, u32x4 (&dst)[DF]) {
#pragma unroll
          for (int d = 0; d < DF; ++d) {
            const u32x2 lo = tr16(vbase + (32 * ks2) * RPV + d * 32);
            const u32x2 hi = tr16(vbase + (32 * ks2 + 16) * RPV + d * 32);
            dst[d] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
          }
        };
        load_v(0, vfr[0]);
#pragma unroll
        for (int ks2 = 0; ks2 < NKT / 2; ++ks2) {
          if (ks2 + 1 < NKT / 2) load_v(ks2 + 1, vfr[(ks2 + 1) & 1]);
          u32x4 pb[NG];
#pragma unroll
          for (int gq = 0; gq < NG; ++gq)
            pb[gq] = (u32x4){pack2<DT>(st[gq][2 * ks2][0], st[gq][2 * ks2][1]), pack2<DT>(st[gq][2 * ks2][2], st[gq][2 * ks2][3]),
                             pack2<DT>(st[gq][2 * ks2 + 1][0], st[gq][2 * ks2 + 1][1]),
                             pack2<DT>(st[gq][2 * ks2 + 1][2], st[gq][2 * ks2 + 1][3])};
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int d = 0; d < DF; ++d)
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) o[gq][d] = mfma16<DT>(vfr[ks2 & 1][d], pb[gq], o[gq][d]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
          half_t* orow = a.out + (size_t)(row_base + q0 + gq * 16 + fr) * out_ld + head * HD;
          unsigned int l8w[DF];   // SPLIT == 2: the fp8 remainder words of this row, one per d-group
#pragma unroll
          for (int d = 0; d < DF; ++d) {
            const int dd = 16 * d + 4 * g;
            l8w[d] = 0u;
            if (dd < HD) {
              if constexpr (SPLIT == 1) {   // [hi | lo]: the out-projection's K-concatenated operand
                unsigned int h0_, l0_, h1_, l1_;
                split2<DT>(o[gq][d][0] * inv[gq], o[gq][d][1] * inv[gq], h0_, l0_);
                split2<DT>(o[gq][d][2] * inv[gq], o[gq][d][3] * inv[gq], h1_, l1_);
                const u32x2 hi = {h0_, h1_}, lo = {l0_, l1_};
                *(u32x2*)(orow + dd) = hi;
                *(u32x2*)(orow + D + dd) = lo;
              } else if constexpr (SPLIT == 2) {   // f16 + fp8 remainder: the out-projection's correction operand (GemmArgs::A8)
                unsigned int h0_, h1_;
                l8w[d] = split8_f16(o[gq][d][0] * inv[gq], o[gq][d][1] * inv[gq], o[gq][d][2] * inv[gq], o[gq][d][3] * inv[gq], h0_, h1_);
                *(u32x2*)(orow + dd) = (u32x2){h0_, h1_};
              } else {
                const u32x2 pk = {pack2<DT>(o[gq][d][0] * inv[gq], o[gq][d][1] * inv[gq]),
                                  pack2<DT>(o[gq][d][2] * inv[gq], o[gq][d][3] * inv[gq])};
                *(u32x2*)(orow + dd) = pk;
              }
            }
          }
          if constexpr (SPLIT == 2) {   // d-groups 0-3: sixteen contiguous bytes per lane (lane row g stores d-group g); d-group 4 (hd = 72) as words
            unsigned char* o8 = a.out8 + (size_t)(row_base + q0 + gq * 16 + fr) * D + head * HD;
            rows_transpose4(l8w[0], l8w[1], l8w[2], l8w[3]);
            *(u32x4*)(o8 + 16 * g) = (u32x4){l8w[0], l8w[1], l8w[2], l8w[3]};
            if constexpr (DF > 4) { if (64 + 4 * g < HD) *(unsigned int*)(o8 + 64 + 4 * g) = l8w[4]; }
          }
        }
      }
    } else {
      // every wave: two sequences of 16 tokens (tile rows 16 p .. 16 p + 15); attn_small_kernel's arithmetic (attention.hip)
      f32x4 st2[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int p = wave * 2 + s2;
        u32x4 qf[KS], kf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int ch = g + 4 * ks;
          qf[ks] = *(const u32x4*)(q_img + (16 * p + fr) * RPQ + ch * 16);
          kf[ks] = *(const u32x4*)(k_img + (16 * p + fr) * RPQ + ch * 16);
          if (ch >= NCH) {
            qf[ks] = (u32x4){0u, 0u, 0u, 0u};
            kf[ks] = (u32x4){0u, 0u, 0u, 0u};
          }
        }
        st2[s2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) st2[s2] = mfma16<DT>(kf[ks], qf[ks], st2[s2]);   // S^T[key = 4g + r][q = fr]
      }
      early_fill();
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int p = wave * 2 + s2;
        f32x4 st = st2[s2];
        float mx = NEG_BIG_F;
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float nm = -mx * c;
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c, nm));   // attn_small_kernel's explicit sequence
          ls += st[r];
        }
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        const u32x2 pb = {pack2<DT>(st[0], st[1]), pack2<DT>(st[2], st[3])};   // P^T[key = 4g + i][q = fr]
        const float inv = 1.0f / ls;
        half_t* orow = a.out + (size_t)(row_base + fr * T + p) * out_ld + head * HD;    // token fr (frame) of sequence p
        u32x2 vf[5];
        {
          const char* vp = v_img + (16 * p + 4 * g + (fr >> 2)) * RPV + (fr & 3) * 8;
          vf[0] = tr16_asm<0>(vp); vf[1] = tr16_asm<32>(vp); vf[2] = tr16_asm<64>(vp); vf[3] = tr16_asm<96>(vp); vf[4] = tr16_asm<128>(vp);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
        unsigned int l8w[DF];   // SPLIT == 2: the fp8 remainder words of this token, one per d-group
#pragma unroll
        for (int d = 0; d < DF; ++d) {
          f32x4 oacc = {0.f, 0.f, 0.f, 0.f};
          oacc = mfma_k16h<DT>(vf[d], pb, oacc);                               // O^T[d = 16 d + 4g + r][q = fr]
          const int dd = 16 * d + 4 * g;
          l8w[d] = 0u;
          if (dd < HD) {
            if constexpr (SPLIT == 1) {
              unsigned int h0_, l0_, h1_, l1_;
              split2<DT>(oacc[0] * inv, oacc[1] * inv, h0_, l0_);
              split2<DT>(oacc[2] * inv, oacc[3] * inv, h1_, l1_);
              const u32x2 hi = {h0_, h1_}, lo = {l0_, l1_};
              *(u32x2*)(orow + dd) = hi;
              *(u32x2*)(orow + D + dd) = lo;
            } else if constexpr (SPLIT == 2) {
              unsigned int h0_, h1_;
              l8w[d] = split8_f16(oacc[0] * inv, oacc[1] * inv, oacc[2] * inv, oacc[3] * inv, h0_, h1_);
              *(u32x2*)(orow + dd) = (u32x2){h0_, h1_};
            } else {
              const u32x2 pk = {pack2<DT>(oacc[0] * inv, oacc[1] * inv), pack2<DT>(oacc[2] * inv, oacc[3] * inv)};
              *(u32x2*)(orow + dd) = pk;
            }
          }
        }
        if constexpr (SPLIT == 2) {
          unsigned char* o8 = a.out8 + (size_t)(row_base + fr * T + p) * D + head * HD;
          rows_transpose4(l8w[0], l8w[1], l8w[2], l8w[3]);
          *(u32x4*)(o8 + 16 * g) = (u32x4){l8w[0], l8w[1], l8w[2], l8w[3]};
          if constexpr (DF > 4) { if (64 + 4 * g < HD) *(unsigned int*)(o8 + 64 + 4 * g) = l8w[4]; }
        }
      }
    }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave is done with the images: the next unit's DMA may overwrite all of them
    LATTE_PHASE(2)
    tacc[3] += 1;
    if (!has_next) break;
    if (!filled_next) {
      fill_first(nxt);
      side = fetch_side(nxt);
    }
    cur = nxt;
    it = it_next;
  }
#undef LATTE_PHASE
  if (trace && lane == 0) {
    long long* o = a.dbg_trace + wave * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = tacc[i];
  }
}

template <int HD, int DT, int MODE, int FLAGS, int SPLIT = 0>
int launch_one(const QkvAttnArgs& a, dim3 grid, hipStream_t st) {
  auto kern = qkv_attn_kernel<HD, DT, MODE, FLAGS, SPLIT>;
  static std::atomic<uint64_t> done{0};
  if (int rc = ensure_dynamic_lds((const void*)kern, FUSED_LDS, done)) return rc;
  hipLaunchKernelGGL(kern, grid, dim3(512), FUSED_LDS, st, a);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

template <int HD, int DT>
int launch_mode(const QkvAttnArgs& a, hipStream_t st) {
  const int S = a.mode == 0 ? a.B * a.F : a.B * (a.T >> 4);
  const int units = S * a.heads;
  const dim3 grid(units >= 256 ? 256 : (units + 7) / 8 * 8);   // one workgroup per CU, a multiple of the 8 XCDs
  if (a.out_split == 2) {   // f16 + fp8 remainder (f16 only)
    if constexpr (DT == LATTE_DTYPE_F16) {
      if (!a.out8) return fail(LATTE_ERR_INVALID, "fused qkv + attention: out_split 2 needs out8");
      return a.mode == 0 ? launch_one<HD, DT, 0, 3, 2>(a, grid, st) : launch_one<HD, DT, 1, 1, 2>(a, grid, st);
    } else {
      return fail(LATTE_ERR_INVALID, "fused qkv + attention: the fp8-remainder output is f16 only");
    }
  }
  if (a.out_split)   // the split-pair output exists for the default schedule of each mode
    return a.mode == 0 ? launch_one<HD, DT, 0, 3, 1>(a, grid, st) : launch_one<HD, DT, 1, 1, 1>(a, grid, st);
  if (a.mode == 0) {
    switch (a.flags & 3) {
      case 0: return launch_one<HD, DT, 0, 0>(a, grid, st);
      case 1: return launch_one<HD, DT, 0, 1>(a, grid, st);
      case 2: return launch_one<HD, DT, 0, 2>(a, grid, st);
      default: return launch_one<HD, DT, 0, 3>(a, grid, st);
    }
  }
  return (a.flags & 1) ? launch_one<HD, DT, 1, 1>(a, grid, st) : launch_one<HD, DT, 1, 0>(a, grid, st);   // (bit 2 is a runtime flag)
}

}  // namespace

bool qkv_attention_fusable(int D, int heads, int hd, int F, int T, int mode, int64_t rows) {
  if (heads * hd != D || (hd != 64 && hd != 72) || D % 64 != 0 || D < 128) return false;
  if ((uint64_t)rows * D * 2 >= (1ull << 32) || (uint64_t)3 * D * D * 2 >= (1ull << 32)) return false;   // 32-bit buffer offsets
  if (mode == 0) return T == 256;                 // one frame = one 256-row tile
  return F == 16 && T % 16 == 0;                  // 16 tokens x 16 frames = one 256-row tile
}

int launch_qkv_attention(const QkvAttnArgs& a, int dtype, hipStream_t st) {
  if (!qkv_attention_fusable(a.D, a.heads, a.hd, a.F, a.T, a.mode, (int64_t)a.B * a.F * a.T))
    return fail(LATTE_ERR_INVALID, "fused qkv + attention: shape not supported (need T == 256 spatial / F == 16 temporal, hd 64 | 72)");
  if (dtype == LATTE_DTYPE_BF16) return a.hd == 64 ? launch_mode<64, LATTE_DTYPE_BF16>(a, st) : launch_mode<72, LATTE_DTYPE_BF16>(a, st);
  if (dtype == LATTE_DTYPE_F16) return a.hd == 64 ? launch_mode<64, LATTE_DTYPE_F16>(a, st) : launch_mode<72, LATTE_DTYPE_F16>(a, st);
  return fail(LATTE_ERR_INVALID, "fused qkv + attention: unknown dtype");
}

}  // namespace latte
