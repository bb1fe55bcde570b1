// Producer-wave GEMM (round 2):  C[M,N] = A[M,K] · W[N,K]^T, 256 x 192 tile, TWELVE waves per workgroup.
//
// Same operand layout, LDS image (source-side swizzle), persistent XCD-chunked tile walk and segment schedule as
// gemm_pps_kernel (gemm.hip), with ONE change of structure: the operand DMA is not issued by the MFMA waves.
// What round 1 measured: a `buffer_load ... lds` holds the wave that issues it for 26-105 shader
// clocks, so in the 8-wave kernel every K tile serialises [64 MFMAs | 12 DMA issues | 24 fragment reads] inside group 0's
// waves and the matrix pipe idles for a third of the period; a DMA wave NEXT TO MFMA waves does not slow them at all
// (`latte_debug_dma_probe` modes 7-9: 16.4 clocks per MFMA beside 4 streaming DMA waves).  So:
//   * waves 0-7  = consumers, 2 groups x 4 (output rows 0-127 / 128-255, 48 columns each): fragment reads + MFMAs + epilogue;
//   * waves 8-11 = producers, one per SIMD (a workgroup's waves are placed on the SIMDs cyclically, so waves w, w + 4, w + 8
//     share one): they only issue the LDS DMA of K tile u + 1 while the consumers work on K tile u, confirm it with counted
//     `vmcnt` and take part in the workgroup barriers.
// Three waves per SIMD means 168 VGPRs per wave (512 / 3, one allocation for the whole kernel), which is why the tile is
// 256 x 192 (96 accumulators) and the fragments of the second half of a K tile are read INSIDE the compute segment, each
// A fragment into the registers its first-half twin has just left: 96 + 24 (B, both halves) + 32 (A) = 152.
//
// Barrier-delimited intervals (barrier b ends interval I(b); P = the barrier after the pipeline fill):
//   group 0:   L(u) in I(2u),   C(u) in I(2u+1)          L(u): B(u) both halves + A(u) first half  -> registers
//   group 1:   L(u) in I(2u+1), C(u) in I(2u+2)          C(u): 48 MFMAs, the 8 second-half A reads between them
//   producers (LDS rings: THREE A stages, two B stages -- with two A stages the producers of the K = 4608 GEMM waited 745 clocks
//   per K tile for an A operand that comes from HBM / the Infinity Cache with one K tile of look-ahead):
//              I(2v)   issue B(v+1) and A rows 0-127 of K tile v+2, then vmcnt(18): everything of K tile v has landed;
//              I(2v+1) issue A rows 128-255 of K tile v+2, then vmcnt(8): B(v+1) (and the older A(v+1)) have landed.
// The K-tile counter runs across tile boundaries, so the producers never drain; a consumer group runs its epilogue
// around the barrier that ends its last compute segment (group 1 before, group 0 after it), as in gemm_pps_kernel.
#include <type_traits>

#include "common.h"
#include "mfma_util.h"

namespace latte {
namespace {

template <int DT>
__device__ __forceinline__ void unpack2pw(unsigned int u, float& a, float& b) {   // two halves of a word -> fp32
  if constexpr (DT == LATTE_DTYPE_BF16) {
    a = __builtin_bit_cast(float, u << 16);
    b = __builtin_bit_cast(float, u & 0xffff0000u);
  } else {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_;
    const f16x2_ h = __builtin_bit_cast(f16x2_, u);
    a = (float)h[0];
    b = (float)h[1];
  }
}


typedef __attribute__((address_space(3))) void lds_void_pw;
__device__ __forceinline__ void pw_bload_lds16(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_pw*)lds_wave_base, 16, voff, soff, 0, 0);
}

// In-place accumulate, pinned by inline assembly: with the fragments carried around the loop (rolling kernel) the register
// allocator otherwise moves accumulators into fragment registers that have just been freed and ends up spilling three of
// them per K tile.  The compiler's hazard recogniser does not look inside: no VALU / memory instruction may read an
// accumulator within 11 wait states of the MFMA that wrote it (the kernel issues two `s_nop 15` before its epilogue), and a
// fragment register is only ever overwritten by an LDS read, which has no hazard against an MFMA reading it as SrcA / SrcB.
template <int DT>
__device__ __forceinline__ void mfma16_ip(f32x4& c, const u32x4& a, const u32x4& b) {
  if constexpr (DT == LATTE_DTYPE_BF16)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// first product of an accumulator: C = 0 (inline constant) instead of a zeroed register set
template <int DT>
__device__ __forceinline__ void mfma16_first(f32x4& c, const u32x4& a, const u32x4& b) {
  if constexpr (DT == LATTE_DTYPE_BF16)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
  else
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
}

// Correction pass of the rolling kernel (GemmArgs::A8 / W8): one block-scaled fp8 MFMA = K = 128 of the contraction, in place like
// mfma16_ip.  Operand maps established on the GPU against a CPU emulation by tools/mx_probe.hip (profiles/r6_mx_probe.log): lane l
// supplies row (l & 15) and the 32 consecutive bytes k = 32 (l >> 4) ... + 31 of its operand in 8 VGPRs, byte op_sel (0 here) of the
// lane's scale register is the E8M0 scale of those 32 elements, C / D as every 16 x 16 MFMA.  The kernel feeds it the two 16-byte
// fragment reads of the half-precision K loop (chunk (l >> 4) and chunk 4 + (l >> 4) of a 128-byte LDS row) as ONE 32-byte operand:
// both operands then hold the same 32 bytes-of-K per lane group, so the instruction contracts over a permutation of K -- the sum is
// the same.  16 passes: the accumulator must not be read by a VALU within 19 wait states (the epilogue's two s_nop 15 cover it).
typedef __attribute__((ext_vector_type(8))) unsigned int u32x8;
__device__ __forceinline__ u32x8 cat8(const u32x4& lo, const u32x4& hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ void mfma8_ip(f32x4& c, const u32x8& a, const u32x8& b, unsigned sa, unsigned sb) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int EPI, int DT, int TAG>
__global__ void __launch_bounds__(768) gemm_pw_kernel(GemmArgs g) {
  constexpr int BM = 256, BN = 192, FN = 3, WTN = 48;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int NA = 3;                         // LDS rings: 3 A stages (96 KB) at offset 0, 2 B stages (48 KB) behind them
  constexpr int B_BASE = NA * A_BYTES;
  constexpr int AH_INSTR = 4, BG_INSTR = 6;   // per producer wave: 128 A rows / (4 waves x 8 rows), 192 W rows / (4 x 8)
  constexpr int GROUP_M = 8;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = g.K;
  const unsigned row_bytes = (unsigned)K * 2u;

  // ---- this workgroup's tile sequence (identical to gemm_pps_kernel)
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = g.N / BN, nwg = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int chunk0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int cnt = q + (xcd < r ? 1 : 0);
  if (slot >= cnt) return;
  const int group_m = g.group_m > 0 ? g.group_m : GROUP_M;
  auto decode = [&](int wg, int& tm, int& tn) {
    const int per_group = group_m * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * group_m;
    const int gsz = min(tiles_m - first_m, group_m);
    const int in_group = wg - group * per_group;
    tm = first_m + in_group % gsz;
    tn = in_group / gsz;
  };
  const int nk = K / 64;
  const int ntile = (cnt - slot + per - 1) / per;   // tiles this workgroup walks
  // measurement build only: per-wave phase times of workgroup 0 (s_memtime ticks), no output written
  constexpr bool TRACE = EPI == EPI_ABLATE_TRACE;
  long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0, tstart = 0;
  if constexpr (TRACE) tstart = tprev = (long long)__builtin_readcyclecounter();
#define LATTE_TS(IDX)                                                \
  if constexpr (TRACE) {                                             \
    const long long now_ = (long long)__builtin_readcyclecounter();  \
    tacc[IDX] += now_ - tprev;                                       \
    tprev = now_;                                                    \
  }
  auto trace_out = [&](int n_it) {
    if constexpr (TRACE) {
      if (blockIdx.x == 0 && lane == 0) {
        long long* o = (long long*)g.out + wave * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) o[i] = tacc[i];
        o[6] = (long long)__builtin_readcyclecounter() - tstart;
        o[7] = n_it;
      }
    }
  };

  if (wave >= 8) {
    // ================================ producer ================================
    const int pw = wave - 8;
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (unsigned)tiles_m * BM * row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (unsigned)g.N * row_bytes, 0x00020000);
    const int lrow = lane >> 3, cpos = lane & 7;
    // row (lane >> 3) of an 8-row group, 16-byte chunk (lane & 7) ^ swizzle(row); rows pw*8 + 32 j: 32 j does not move the swizzle
    const unsigned voff = (unsigned)lrow * row_bytes + (unsigned)((cpos ^ (((pw * 8 + lrow) >> 1) & 7)) * 16);
    unsigned step32 = 32u * row_bytes;
    asm volatile("" : "+s"(step32));
    auto dma_a = [&](int half, int tm_, int kt, int stg) {
      char* sA = smem + stg * A_BYTES + half * 128 * 128 + pw * 1024;
      const unsigned so = (unsigned)(tm_ * BM + half * 128 + pw * 8) * row_bytes + (unsigned)kt * 128u;
#pragma unroll
      for (int j = 0; j < AH_INSTR; ++j) pw_bload_lds16(rsA, sA + j * 4 * 1024, voff, so + (unsigned)j * step32);
    };
    auto dma_b = [&](int tn_, int kt, int stg) {
      char* sB = smem + B_BASE + stg * B_BYTES + pw * 1024;
      const unsigned so = (unsigned)(tn_ * BN + pw * 8) * row_bytes + (unsigned)kt * 128u;
#pragma unroll
      for (int j = 0; j < BG_INSTR; ++j) pw_bload_lds16(rsB, sB + j * 4 * 1024, voff, so + (unsigned)j * step32);
    };
    // Two walkers over the K tiles of this workgroup's tile sequence: `b` (for B) is one K tile ahead of the consumers,
    // `a` (for A, the operand that streams from HBM in the long-K GEMMs) two.
    struct Walk { int pos, tm, tn, kt; };
    auto advance = [&](Walk& w) {
      if (++w.kt == nk) {
        w.kt = 0;
        w.pos += per;
        if (w.pos < cnt) decode(chunk0 + w.pos, w.tm, w.tn);
      }
    };
    Walk wb{slot, 0, 0, 0};
    decode(chunk0 + slot, wb.tm, wb.tn);
    Walk wa = wb;
    const int U = ntile * nk;
    // pipeline fill: K tile 0 complete, A of K tile 1
    dma_b(wb.tn, 0, 0);
    dma_a(0, wa.tm, 0, 0);
    dma_a(1, wa.tm, 0, 0);
    advance(wa);
    advance(wb);                       // wb -> K tile 1
    if (U > 1) { dma_a(0, wa.tm, wa.kt, 1); dma_a(1, wa.tm, wa.kt, 1); }
    advance(wa);                       // wa -> K tile 2
    if (U > 1) wait_vm<2 * AH_INSTR>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();   // P
    int sa = 2, sb = 1;             // ring slots of K tile v + 2 (A) and v + 1 (B)
    for (int v = 0; v < U; ++v) {
      const bool more_b = v + 1 < U, more_a = v + 2 < U;
      // I(2v): B(v+1), A rows 0-127 of K tile v+2; then everything of K tile v must have landed
      LATTE_TS(5)
      if (more_b) dma_b(wb.tn, wb.kt, sb);
      if (more_a) dma_a(0, wa.tm, wa.kt, sa);
      LATTE_TS(0)
      // newer than K tile v: A(v+1) 8, B(v+1) 6, A0(v+2) 4
      if (more_a) wait_vm<2 * AH_INSTR + BG_INSTR + AH_INSTR>(); else wait_vm<0>();
      LATTE_TS(1)
      __builtin_amdgcn_s_barrier();
      LATTE_TS(2)
      // I(2v+1): A rows 128-255 of K tile v+2; then B(v+1) (and with it A(v+1), which is older) must have landed
      if (more_a) dma_a(1, wa.tm, wa.kt, sa);
      LATTE_TS(0)
      if (more_a) wait_vm<2 * AH_INSTR>(); else wait_vm<0>();
      LATTE_TS(3)
      __builtin_amdgcn_s_barrier();
      LATTE_TS(4)
      advance(wa);
      advance(wb);
      sa = sa == NA - 1 ? 0 : sa + 1;
      sb ^= 1;
    }
    __builtin_amdgcn_s_barrier();   // barrier 2U
    trace_out(U);
    return;
  }

  // ================================ consumer ================================
  const int grp = wave >> 2, wn = wave & 3;
  const int sw = (lane >> 1) & 7;
  const int chunkb = ((lane >> 4) ^ sw) * 16;
  const int a_off = (grp * 128 + (lane & 15)) * 128 + chunkb;
  const int b_off = B_BASE + (wn * WTN + (lane & 15)) * 128 + chunkb;

  f32x4 acc[8][FN];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Epilogue of tile (tm_, tn_) for this wave's 128 x 48 sub-tile, then clear the accumulators.
  auto epilogue = [&](int tm_, int tn_) {
    int le = lane;
    asm volatile("" : "+v"(le));   // opaque: keeps every lane-derived epilogue index out of the K loop's live set
    const int fr = le & 15;
    const int ncol = tn_ * BN + wn * WTN + (le >> 4) * 4;
    const int mbase = tm_ * BM + grp * 128 + fr;
    if constexpr (TRACE) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) { asm volatile("" ::"v"(acc[i][j])); acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      return;
    }
    if constexpr (EPI == EPI_GATE_RES_F32) {
      // res[m, n] += gate[sample, n] * (acc + bias); residual loads run two fragments ahead of the stores (gemm.hip)
      if ((g.rows_per_sample % BM) == 0) {
        float* const outp = (float*)g.out;
        const float* gr = g.gate + (size_t)((tm_ * BM) / g.rows_per_sample) * g.gate_stride + ncol;
        float4 b4[FN], g1[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          b4[j] = *(const float4*)(g.bias + ncol + j * 16);
          g1[j] = *(const float4*)(gr + j * 16);
        }
        constexpr int NF = 8 * FN, AHEAD = 2;
        auto frag_ptr = [&](int f) -> float* {
          const int mc = min(mbase + (f / FN) * 16, g.M - 1);
          return outp + (size_t)mc * g.N + ncol + (f % FN) * 16;
        };
        float4 qa[AHEAD];
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) qa[a] = *(const float4*)frag_ptr(a);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          float4 rr = qa[f % AHEAD];
          if (f + AHEAD < NF) qa[f % AHEAD] = *(const float4*)frag_ptr(f + AHEAD);
          asm volatile("" ::: "memory");
          const int i = f / FN, j = f % FN;
          rr.x += g1[j].x * (acc[i][j][0] + b4[j].x);
          rr.y += g1[j].y * (acc[i][j][1] + b4[j].y);
          rr.z += g1[j].z * (acc[i][j][2] + b4[j].z);
          rr.w += g1[j].w * (acc[i][j][3] + b4[j].w);
          if (mbase + i * 16 < g.M) *(float4*)(outp + (size_t)(mbase + i * 16) * g.N + ncol + j * 16) = rr;
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = mbase + i * 16;
      if (m < g.M) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int n = ncol + j * 16;
          const float4 b4 = *(const float4*)(g.bias + n);
          float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
          const size_t o = (size_t)m * g.N + n;
          if constexpr (EPI == EPI_GATE_RES_F32) {
            const float4 g4 = *(const float4*)(g.gate + (size_t)(m / g.rows_per_sample) * g.gate_stride + n);
            float4* dst = (float4*)((float*)g.out + o);
            float4 rr = *dst;
            rr.x += g4.x * v0; rr.y += g4.y * v1; rr.z += g4.z * v2; rr.w += g4.w * v3;
            *dst = rr;
          } else if constexpr (EPI == EPI_BIAS_F32) {
            *(float4*)((float*)g.out + o) = make_float4(v0, v1, v2, v3);
          } else {
            if constexpr (EPI == EPI_BIAS_GELU_H16) {
              auto gelu = [](float x) {
                const float p = __builtin_fmaf(x * x, -0.10294324f, -2.3022082f);
                return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p * x));
              };
              v0 = gelu(v0); v1 = gelu(v1); v2 = gelu(v2); v3 = gelu(v3);
            }
            const u32x2 p = {pack2<DT>(v0, v1), pack2<DT>(v2, v3)};
            *(u32x2*)((half_t*)g.out + o) = p;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };

  int pos = slot, tm, tn;
  decode(chunk0 + pos, tm, tn);
  __builtin_amdgcn_s_barrier();                 // P: K tile 0 has landed
  if (grp == 1) __builtin_amdgcn_s_barrier();   // barrier 0: group 1 runs one segment behind

  int it = 0, ia = 0;   // K-tile counter of the walk, its A ring slot (it % 3)
  for (;;) {
    for (int kt = 0; kt < nk; ++kt, ++it) {
      const bool last = kt + 1 == nk;
      const char* sbufA = smem + ia * A_BYTES;
      const char* sbufB = smem + (it & 1) * B_BYTES;
      ia = ia == NA - 1 ? 0 : ia + 1;
      u32x4 bf[2][FN], af[8];
      // ---- L(u)
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[0][j] = *(const u32x4*)(sbufB + (b_off + j * 2048));
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = *(const u32x4*)(sbufA + (a_off + i * 2048));
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[1][j] = *(const u32x4*)(sbufB + ((b_off + j * 2048) ^ 64));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      LATTE_TS(0)
      __builtin_amdgcn_s_barrier();
      LATTE_TS(1)
      // ---- C(u)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16<DT>(bf[0][j], af[i], acc[i][j]);
        // second-half fragment of the same rows into the registers the MFMAs above have just read
        af[i] = *(const u32x4*)(sbufA + ((a_off + i * 2048) ^ 64));
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16<DT>(bf[1][j], af[i], acc[i][j]);
      }
      __builtin_amdgcn_s_setprio(0);
      LATTE_TS(2)
      if (grp == 1 && last) epilogue(tm, tn);
      LATTE_TS(3)
      __builtin_amdgcn_s_barrier();
      LATTE_TS(4)
      if (grp == 0 && last) epilogue(tm, tn);
      LATTE_TS(5)
    }
    pos += per;
    if (pos >= cnt) break;
    decode(chunk0 + pos, tm, tn);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();   // barrier 2U (group 1 spent it up front)
  trace_out(it);
#undef LATTE_TS
}


// ------------------------------------------------------------------------------------------------
// Rolling variant (variant 11): the same 12-wave workgroup, but NO load segment at all.  The trace of variant 10 shows what is
// left once the MFMA waves stop issuing DMA: the two consumer groups still hand the matrix pipe to each other through two
// barriers per K tile (~190 clocks each, 20 % of the period) because a group's fragment reads sit in their own segment.  Here
// every consumer wave keeps its MFMA stream going and reads the fragments of the NEXT half K tile (32 deep) between the MFMAs
// of the current one, each A fragment into the registers the MFMAs just issued have left (B fragments are double-buffered:
// 24 registers) -- a software pipeline one half-step deep that costs no register beyond variant 10's 56.  Both groups run in
// lock step (two waves per SIMD share the matrix pipe; a read is consumed ~24 own MFMAs after it was issued), and ONE barrier
// per K tile -- placed 6 MFMAs into the second half-step, when every read of stage u has been issued long ago -- tells the
// producers that stage u may be overwritten and the consumers that stage u + 1 has landed.
//   consumer, K tile u:  h0: MFMAs on first-half fragments | second-half A fragments of u roll in
//                        h1: 6 MFMAs, lgkmcnt(0), barrier B_u, B first half of u+1, 18 MFMAs | first-half A fragments of u+1
//                            roll in (lagging two fragment rows), then B second half of u+1
//   producer:            ... vmcnt(8) [stage u+1 landed], barrier B_u, issue B(u+2) and A(u+3)      (3 A stages, 2 B stages)
// At a tile boundary the pipeline is drained (no look-ahead reads in the last half-step: the epilogue needs the registers)
// and refilled from stage u + 1 after the epilogue.
// LO (round 6): behind the K / 64 half-precision K tiles of an output tile the walk continues over K / 128 "virtual" K tiles of the
// fp8 correction operands A8 / W8 (GemmArgs) -- a 128-byte row piece of an fp8 operand is K = 128, so a correction K tile has the
// byte geometry, the LDS image, the swizzle and the DMA pieces of a half-precision one; the producers only switch descriptor and row
// pitch, the consumers run lo_tile() (24 MFMAs of 16 passes on the same fragment reads) instead of ktile() (48 of 8 passes).
// ABL (measurement build only, round 6 -- the probe-to-kernel LADDER of tools/ladder_probe.py, results are garbage): bits taken away from
// the production kernel one at a time, top down:  1 = no epilogue (accumulators dropped at a tile boundary), 2 = no tile boundaries (the
// K walk never drains / refills), 4 = operands from an L2-resident pool (tile 0, K tiles 0-3) instead of their real addresses, 8 = no
// vmcnt / lgkmcnt waits and no barrier inside the K walk, 16 = fragments read from the LDS once and reused, 32 = no operand DMA.
template <int EPI, int DT, int TAG, int LO = 0, int ABL = 0>
__global__ void __launch_bounds__(768) gemm_pwr_kernel(GemmArgs g) {
  constexpr int BM = 256, BN = 192, FN = 3, WTN = 48;
  constexpr bool A_NOEPI = (ABL & 1) != 0, A_ONE = (ABL & 2) != 0, A_HOT = (ABL & 4) != 0, A_NOWAIT = (ABL & 8) != 0,
                 A_NOLDS = (ABL & 16) != 0, A_NODMA = (ABL & 32) != 0;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int NA = 3;
  constexpr int B_BASE = NA * A_BYTES;
  constexpr int AH_INSTR = 4, BG_INSTR = 6;
  constexpr int GROUP_M = 8;
  constexpr int PATCH_BYTES = 16 * 112;   // wave-private epilogue patch (half-precision outputs): 16 rows, pitch 112 B

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = g.K;
  const unsigned row_bytes = (unsigned)K * 2u;

  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = g.N / BN, nwg = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int chunk0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int cnt = q + (xcd < r ? 1 : 0);
  if (slot >= cnt) return;
  const int group_m = g.group_m > 0 ? g.group_m : GROUP_M;
  auto decode = [&](int wg, int& tm, int& tn) {
    const int per_group = group_m * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * group_m;
    const int gsz = min(tiles_m - first_m, group_m);
    const int in_group = wg - group * per_group;
    tm = first_m + in_group % gsz;
    tn = in_group / gsz;
  };
  const int nk = K / 64;
  const int nk8 = LO >= 2 ? (K + 255) / 256 : LO ? K / 128 : 0;   // correction K tiles behind the nk half-precision ones (fp4: K = 256 per 128 row bytes)
  const int nkt = nk + nk8;                       // K tiles of the walk per output tile
  const int ntile = (cnt - slot + per - 1) / per;
  constexpr bool TRACE = EPI == EPI_ABLATE_TRACE;
  long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0, tstart = 0;
  if constexpr (TRACE) tstart = tprev = (long long)__builtin_readcyclecounter();
#define LATTE_TS(IDX)                                                \
  if constexpr (TRACE) {                                             \
    const long long now_ = (long long)__builtin_readcyclecounter();  \
    tacc[IDX] += now_ - tprev;                                       \
    tprev = now_;                                                    \
  }
  auto trace_out = [&](int n_it) {
    if constexpr (TRACE) {
      if (blockIdx.x == 0 && lane == 0) {
        long long* o = (long long*)g.out + wave * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) o[i] = tacc[i];
        o[6] = (long long)__builtin_readcyclecounter() - tstart;
        o[7] = n_it;
      }
    }
  };

  if (wave >= 8) {
    // ================================ producer ================================
    const int pw = wave - 8;
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (unsigned)tiles_m * BM * row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (unsigned)g.N * row_bytes, 0x00020000);
    const int lrow = lane >> 3, cpos = lane & 7;
    const unsigned voff = (unsigned)lrow * row_bytes + (unsigned)((cpos ^ (((pw * 8 + lrow) >> 1) & 7)) * 16);
    unsigned step32 = 32u * row_bytes;
    asm volatile("" : "+s"(step32));
    // correction operands: rows of K bytes
    const unsigned row_bytes8 = LO >= 2 ? (unsigned)((K + 255) / 256 * 128) : (unsigned)K;   // fp4: two codes per byte, rows padded to K % 256 == 0
    const __amdgpu_buffer_rsrc_t rsA8 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(LO >= 2 ? (const void*)g.A4 : LO ? (const void*)g.A8 : (const void*)g.A), 0, (unsigned)tiles_m * BM * row_bytes8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB8 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(LO >= 2 ? (const void*)g.W4 : LO ? (const void*)g.W8 : (const void*)g.W), 0, (unsigned)g.N * row_bytes8, 0x00020000);
    const unsigned voff8 = (unsigned)lrow * row_bytes8 + (unsigned)((cpos ^ (((pw * 8 + lrow) >> 1) & 7)) * 16);
    unsigned step32_8 = 32u * row_bytes8;
    asm volatile("" : "+s"(step32_8));
    auto dma_a = [&](int tm_, int kt, int stg) {   // all 256 rows: row-groups pw + 4 j, j = 0..7
      char* sA = smem + stg * A_BYTES + pw * 1024;
      if constexpr (A_NODMA) return;
      if constexpr (A_HOT) { tm_ = 0; kt &= 3; }
      if (LO && kt >= nk) {
        const unsigned so = (unsigned)(tm_ * BM + pw * 8) * row_bytes8 + (unsigned)(kt - nk) * 128u;
#pragma unroll
        for (int j = 0; j < 2 * AH_INSTR; ++j) pw_bload_lds16(rsA8, sA + j * 4 * 1024, voff8, so + (unsigned)j * step32_8);
        return;
      }
      const unsigned so = (unsigned)(tm_ * BM + pw * 8) * row_bytes + (unsigned)kt * 128u;
#pragma unroll
      for (int j = 0; j < 2 * AH_INSTR; ++j) pw_bload_lds16(rsA, sA + j * 4 * 1024, voff, so + (unsigned)j * step32);
    };
    auto dma_b = [&](int tn_, int kt, int stg) {
      char* sB = smem + B_BASE + stg * B_BYTES + pw * 1024;
      if constexpr (A_NODMA) return;
      if constexpr (A_HOT) { tn_ = 0; kt &= 3; }
      if (LO && kt >= nk) {
        const unsigned so = (unsigned)(tn_ * BN + pw * 8) * row_bytes8 + (unsigned)(kt - nk) * 128u;
#pragma unroll
        for (int j = 0; j < BG_INSTR; ++j) pw_bload_lds16(rsB8, sB + j * 4 * 1024, voff8, so + (unsigned)j * step32_8);
        return;
      }
      const unsigned so = (unsigned)(tn_ * BN + pw * 8) * row_bytes + (unsigned)kt * 128u;
#pragma unroll
      for (int j = 0; j < BG_INSTR; ++j) pw_bload_lds16(rsB, sB + j * 4 * 1024, voff, so + (unsigned)j * step32);
    };
    struct Walk { int pos, tm, tn, kt; };
    auto advance = [&](Walk& w) {
      if (++w.kt == nkt) {
        w.kt = 0;
        w.pos += per;
        if (w.pos < cnt) decode(chunk0 + w.pos, w.tm, w.tn);
      }
    };
    const int U = ntile * nkt;
    Walk wb{slot, 0, 0, 0};
    decode(chunk0 + slot, wb.tm, wb.tn);
    Walk wa = wb;
    // pipeline fill, issue order  B0 A0 | B1 A1 | A2   (U >= 2: K >= 128)
    dma_b(wb.tn, wb.kt, 0); advance(wb);
    dma_a(wa.tm, wa.kt, 0); advance(wa);
    dma_b(wb.tn, wb.kt, 1); advance(wb);
    dma_a(wa.tm, wa.kt, 1); advance(wa);
    if (U > 2) { dma_a(wa.tm, wa.kt, 2); wait_vm<BG_INSTR + 4 * AH_INSTR>(); } else { wait_vm<BG_INSTR + 2 * AH_INSTR>(); }
    advance(wa);                    // wa -> K tile 3, wb -> K tile 2
    __builtin_amdgcn_s_barrier();   // P: K tile 0 has landed
    int sa = 0, sb = 0;             // ring slots of K tile u + 3 (A: (u + 3) % 3 = u % 3) and u + 2 (B: u & 1)
    for (int u = 0; u < U; ++u) {
      LATTE_TS(5)
      // stage u + 1 must have landed; the only younger DMA is A(u+2)
      if constexpr (!A_NOWAIT) { if (u + 2 < U) wait_vm<2 * AH_INSTR>(); else wait_vm<0>(); }
      LATTE_TS(1)
      if constexpr (!A_NOWAIT) __builtin_amdgcn_s_barrier();   // B_u: stage u is free
      LATTE_TS(2)
      if (u + 2 < U) dma_b(wb.tn, wb.kt, sb);
      if (u + 3 < U) dma_a(wa.tm, wa.kt, sa);
      LATTE_TS(0)
      advance(wa);
      advance(wb);
      sa = sa == NA - 1 ? 0 : sa + 1;
      sb ^= 1;
    }
    trace_out(U);
    return;
  }

  // ================================ consumer ================================
  const int grp = wave >> 2, wn = wave & 3;
  const int sw = (lane >> 1) & 7;
  const int chunkb = ((lane >> 4) ^ sw) * 16;
  const int a_off = (grp * 128 + (lane & 15)) * 128 + chunkb;
  const int b_off = B_BASE + (wn * WTN + (lane & 15)) * 128 + chunkb;

  f32x4 acc[8][FN];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto epilogue = [&](int tm_, int tn_) {
    int le = lane;
    asm volatile("" : "+v"(le));
    const int fr = le & 15;
    const int ncol = tn_ * BN + wn * WTN + (le >> 4) * 4;
    const int mbase = tm_ * BM + grp * 128 + fr;
    if constexpr (TRACE || A_NOEPI) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) { asm volatile("" ::"v"(acc[i][j])); acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      return;
    }
    if constexpr (EPI == EPI_BIAS_H16 || EPI == EPI_BIAS_GELU_H16 || EPI == EPI_BIAS_GELU_DUAL_H16 || EPI == EPI_DGELU_H16) {
      // Half-precision outputs: a fragment row (16 rows x 48 columns = 96 B per row) goes through a wave-private LDS patch
      // (pitch 112 B, behind the operand rings) so that the global stores are 16 B per lane on contiguous 96-byte row
      // segments: 1.5 store instructions per fragment row instead of 3 with 8 B per lane on 32-byte pieces.
      // Training forms (round 6): EPI_BIAS_GELU_DUAL_H16 stores u and, through the same patch a second time, gelu(u) to g.aux;
      // EPI_DGELU_H16 multiplies by gelu'(u) with u read from g.aux (8 B per lane: the fragment's own elements).
      char* patch = smem + B_BASE + 2 * B_BYTES + wave * PATCH_BYTES;
      const int ncol0 = tn_ * BN + wn * WTN;
      const int gq = le >> 4;
      float4 b4[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) b4[j] = *(const float4*)(g.bias + ncol0 + j * 16 + gq * 4);
      const int r0 = le / 6, p0 = le - r0 * 6;               // piece le      -> (row, 16-byte piece) of the 16 x 6 grid
      const int r1 = (le + 64) / 6, p1 = (le + 64) - r1 * 6; // piece le + 64 (lanes 0-31)
      half_t* const outp = (half_t*)g.out;
      const int mrow0 = tm_ * BM + grp * 128;
      auto gelu = [](float x) {
        const float p = __builtin_fmaf(x * x, -0.10294324f, -2.3022082f);
        return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p * x));
      };
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        u32x2 side[FN];   // DUAL: gelu(u) of this fragment row, DGELU: its u
        if constexpr (EPI == EPI_DGELU_H16) {
          const int mu = min(mrow0 + i * 16 + fr, g.M - 1);
#pragma unroll
          for (int j = 0; j < FN; ++j) side[j] = *(const u32x2*)(g.aux + (size_t)mu * g.N + ncol0 + j * 16 + gq * 4);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          float v0 = acc[i][j][0] + b4[j].x, v1 = acc[i][j][1] + b4[j].y, v2 = acc[i][j][2] + b4[j].z, v3 = acc[i][j][3] + b4[j].w;
          if constexpr (EPI == EPI_BIAS_GELU_H16) {
            v0 = gelu(v0); v1 = gelu(v1); v2 = gelu(v2); v3 = gelu(v3);
          }
          if constexpr (EPI == EPI_DGELU_H16) {   // train.hip: gelu_kernel<BWD>, the same arithmetic on the unrounded product
            float x[4];
            unpack2pw<DT>(side[j][0], x[0], x[1]);
            unpack2pw<DT>(side[j][1], x[2], x[3]);
            auto dgelu = [](float xx) {
              const float p = __builtin_fmaf(xx * xx, -0.10294324f, -2.3022082f);
              const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p * xx));
              const float du = 0.7978845608f * (1.0f + 0.134145f * xx * xx);
              return sg + xx * sg * (1.0f - sg) * 2.0f * du;
            };
            v0 *= dgelu(x[0]); v1 *= dgelu(x[1]); v2 *= dgelu(x[2]); v3 *= dgelu(x[3]);
          }
          const u32x2 pk = {pack2<DT>(v0, v1), pack2<DT>(v2, v3)};
          *(u32x2*)(patch + fr * 112 + j * 32 + gq * 8) = pk;
          if constexpr (EPI == EPI_BIAS_GELU_DUAL_H16) {   // the GELU of the ROUNDED pre-activation (what gelu_kernel<FWD> computes from u)
            float x[4];
            unpack2pw<DT>(pk[0], x[0], x[1]);
            unpack2pw<DT>(pk[1], x[2], x[3]);
            side[j] = (u32x2){pack2<DT>(gelu(x[0]), gelu(x[1])), pack2<DT>(gelu(x[2]), gelu(x[3]))};
          }
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const int m0_ = mrow0 + i * 16 + r0, m1_ = mrow0 + i * 16 + r1;
        {
          const u32x4 w0 = *(const u32x4*)(patch + r0 * 112 + p0 * 16);
          if (m0_ < g.M) *(u32x4*)(outp + (size_t)m0_ * g.N + ncol0 + p0 * 8) = w0;
          if (le < 32) {
            const u32x4 w1 = *(const u32x4*)(patch + r1 * 112 + p1 * 16);
            if (m1_ < g.M) *(u32x4*)(outp + (size_t)m1_ * g.N + ncol0 + p1 * 8) = w1;
          }
        }
        if constexpr (EPI == EPI_BIAS_GELU_DUAL_H16) {   // second trip through the patch (LDS operations of a wave execute in order)
#pragma unroll
          for (int j = 0; j < FN; ++j) *(u32x2*)(patch + fr * 112 + j * 32 + gq * 8) = side[j];
          const u32x4 w0 = *(const u32x4*)(patch + r0 * 112 + p0 * 16);
          if (m0_ < g.M) *(u32x4*)(g.aux + (size_t)m0_ * g.N + ncol0 + p0 * 8) = w0;
          if (le < 32) {
            const u32x4 w1 = *(const u32x4*)(patch + r1 * 112 + p1 * 16);
            if (m1_ < g.M) *(u32x4*)(g.aux + (size_t)m1_ * g.N + ncol0 + p1 * 8) = w1;
          }
        }
      }
      return;
    }
    if constexpr (EPI == EPI_GATE_RES_F32) {
      if ((g.rows_per_sample % BM) == 0) {
        float* const outp = (float*)g.out;
        const float* gr = g.gate + (size_t)((tm_ * BM) / g.rows_per_sample) * g.gate_stride + ncol;
        float4 b4[FN], g1[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          b4[j] = *(const float4*)(g.bias + ncol + j * 16);
          g1[j] = *(const float4*)(gr + j * 16);
        }
        if constexpr ((ABL & 64) != 0) {   // ladder experiment: the read-modify-write as fire-and-forget fp32 atomic adds (L2-side add)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const bool ok = mbase + i * 16 < g.M;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
              float* dst = outp + (size_t)min(mbase + i * 16, g.M - 1) * g.N + ncol + j * 16;
              const float v0 = g1[j].x * (acc[i][j][0] + b4[j].x), v1 = g1[j].y * (acc[i][j][1] + b4[j].y);
              const float v2 = g1[j].z * (acc[i][j][2] + b4[j].z), v3 = g1[j].w * (acc[i][j][3] + b4[j].w);
              if (ok) {
                typedef __attribute__((address_space(1))) float gfloat;
                __builtin_amdgcn_global_atomic_fadd_f32((gfloat*)dst, v0);
                __builtin_amdgcn_global_atomic_fadd_f32((gfloat*)(dst + 1), v1);
                __builtin_amdgcn_global_atomic_fadd_f32((gfloat*)(dst + 2), v2);
                __builtin_amdgcn_global_atomic_fadd_f32((gfloat*)(dst + 3), v3);
              }
              acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
          }
          return;
        }
        constexpr int NF = 8 * FN, AHEAD = (ABL & 128) ? 6 : (ABL & 256) ? 12 : 2;   // (ladder experiment: deeper residual look-ahead)
        auto frag_ptr = [&](int f) -> float* {
          const int mc = min(mbase + (f / FN) * 16, g.M - 1);
          return outp + (size_t)mc * g.N + ncol + (f % FN) * 16;
        };
        float4 qa[AHEAD];
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) qa[a] = *(const float4*)frag_ptr(a);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          float4 rr = qa[f % AHEAD];
          if (f + AHEAD < NF) qa[f % AHEAD] = *(const float4*)frag_ptr(f + AHEAD);
          asm volatile("" ::: "memory");
          const int i = f / FN, j = f % FN;
          rr.x += g1[j].x * (acc[i][j][0] + b4[j].x);
          rr.y += g1[j].y * (acc[i][j][1] + b4[j].y);
          rr.z += g1[j].z * (acc[i][j][2] + b4[j].z);
          rr.w += g1[j].w * (acc[i][j][3] + b4[j].w);
          if (mbase + i * 16 < g.M) *(float4*)(outp + (size_t)(mbase + i * 16) * g.N + ncol + j * 16) = rr;
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = mbase + i * 16;
      if (m < g.M) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int n = ncol + j * 16;
          const float4 b4 = *(const float4*)(g.bias + n);
          float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
          const size_t o = (size_t)m * g.N + n;
          if constexpr (EPI == EPI_GATE_RES_F32) {
            const float4 g4 = *(const float4*)(g.gate + (size_t)(m / g.rows_per_sample) * g.gate_stride + n);
            float4* dst = (float4*)((float*)g.out + o);
            float4 rr = *dst;
            rr.x += g4.x * v0; rr.y += g4.y * v1; rr.z += g4.z * v2; rr.w += g4.w * v3;
            *dst = rr;
          } else if constexpr (EPI == EPI_BIAS_F32) {
            *(float4*)((float*)g.out + o) = make_float4(v0, v1, v2, v3);
          } else {
            if constexpr (EPI == EPI_BIAS_GELU_H16) {
              auto gelu = [](float x) {
                const float p = __builtin_fmaf(x * x, -0.10294324f, -2.3022082f);
                return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p * x));
              };
              v0 = gelu(v0); v1 = gelu(v1); v2 = gelu(v2); v3 = gelu(v3);
            }
            const u32x2 p = {pack2<DT>(v0, v1), pack2<DT>(v2, v3)};
            *(u32x2*)((half_t*)g.out + o) = p;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };

  u32x4 bf[2][FN], af[8];
  auto fill = [&](const char* sA, const char* sB) {   // the complete first half-step and the B fragments of the second
#pragma unroll
    for (int j = 0; j < FN; ++j) bf[0][j] = *(const u32x4*)(sB + (b_off + j * 2048));
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = *(const u32x4*)(sA + (a_off + i * 2048));
#pragma unroll
    for (int j = 0; j < FN; ++j) bf[1][j] = *(const u32x4*)(sB + ((b_off + j * 2048) ^ 64));
  };

  int pos = slot, tm, tn;
  decode(chunk0 + pos, tm, tn);
  __builtin_amdgcn_s_barrier();   // P: K tile 0 has landed
  fill(smem, smem);

  // One K tile of the pipeline; LOOK = read the fragments of stage u + 1 while the second half-step computes.
  auto ktile = [&](const char* sA, const char* sAn, const char* sBn, auto look) {
    constexpr bool LOOK = decltype(look)::value;
    // ---- h0(u): first-half MFMAs, second-half A fragments roll in
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) mfma16_ip<DT>(acc[i][j], bf[0][j], af[i]);
      if constexpr (!A_NOLDS) af[i] = *(const u32x4*)(sA + ((a_off + i * 2048) ^ 64));
      __builtin_amdgcn_sched_barrier(0);
    }
    LATTE_TS(0)
    // ---- h1(u), fragment rows 0-1; then every read of stage u has completed: B_u
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) mfma16_ip<DT>(acc[i][j], bf[1][j], af[i]);
    }
    if constexpr (!A_NOWAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LATTE_TS(1)
    if constexpr (!A_NOWAIT) __builtin_amdgcn_s_barrier();
    LATTE_TS(2)
    if constexpr (LOOK && !A_NOLDS) {
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[0][j] = *(const u32x4*)(sBn + (b_off + j * 2048));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 2; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) mfma16_ip<DT>(acc[i][j], bf[1][j], af[i]);
      if constexpr (LOOK && !A_NOLDS) af[i - 2] = *(const u32x4*)(sAn + (a_off + (i - 2) * 2048));
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (LOOK && !A_NOLDS) {
      af[6] = *(const u32x4*)(sAn + (a_off + 6 * 2048));
      af[7] = *(const u32x4*)(sAn + (a_off + 7 * 2048));
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[1][j] = *(const u32x4*)(sBn + ((b_off + j * 2048) ^ 64));
    }
    LATTE_TS(3)
  };

  // ---- correction K tiles (LO): the B pairs of the tile in b8, the A pairs in a ring of four (a8[i & 3] = fragment row i); same
  // one-barrier protocol as ktile(): every read of stage u has returned in front of B_u, stage u + 1 is read behind it.
  //   rows 0-3: 3 MFMAs each, the pair of row i + 4 rolls into the ring slot just read | rows 4-5 | lgkmcnt(0), B_u |
  //   rows 6-7, with LOOK the pairs of stage u + 1 roll in behind the MFMAs that free their registers (A rows 0-3, then B)
  u32x8 b8[FN], a8[4];
  unsigned sc_w = 0, sc_a = 0;
  if constexpr (LO == 1) {
    sc_w = 127u - LO8_W_SHIFT;
    sc_a = 127u - LO8_A_SHIFT;
    asm volatile("" : "+v"(sc_w), "+v"(sc_a));
  }
  auto pair_a = [&](const char* sA, int i) {
    return cat8(*(const u32x4*)(sA + (a_off + i * 2048)), *(const u32x4*)(sA + ((a_off + i * 2048) ^ 64)));
  };
  auto pair_b = [&](const char* sB, int j) {
    return cat8(*(const u32x4*)(sB + (b_off + j * 2048)), *(const u32x4*)(sB + ((b_off + j * 2048) ^ 64)));
  };
  auto lo_fill = [&](const char* sA, const char* sB) {
#pragma unroll
    for (int j = 0; j < FN; ++j) b8[j] = pair_b(sB, j);
#pragma unroll
    for (int i = 0; i < 4; ++i) a8[i] = pair_a(sA, i);
  };
  auto lo_tile = [&](const char* sA, const char* sAn, const char* sBn, auto look) {
    constexpr bool LOOK = decltype(look)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) mfma8_ip(acc[i][j], b8[j], a8[i], sc_w, sc_a);
      a8[i] = pair_a(sA, i + 4);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 4; i < 6; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) mfma8_ip(acc[i][j], b8[j], a8[i & 3], sc_w, sc_a);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (LOOK) {
      a8[0] = pair_a(sAn, 0);
      a8[1] = pair_a(sAn, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < FN; ++j) mfma8_ip(acc[6][j], b8[j], a8[2], sc_w, sc_a);
    if constexpr (LOOK) a8[2] = pair_a(sAn, 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      mfma8_ip(acc[7][j], b8[j], a8[3], sc_w, sc_a);
      if constexpr (LOOK) b8[j] = pair_b(sBn, j);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (LOOK) a8[3] = pair_a(sAn, 3);
  };

  // ---- FP4 correction K tiles (LO == 2, round 6): a K tile is 128 row bytes = K 256; the two 16-byte reads of a fragment row are two
  // operands of v_mfma_scale_f32_16x16x128_f8f6f4 (cbsz = blgp = 4: 4 passes, layout H0 of tools/mx_probe.hip -- lane (row, g) holds the 32
  // consecutive K of chunk g), so a stage is walked as two HALVES (chunks 0-3, then 4-7) with half-size register sets; the scale operand
  // of a lane is the E8M0 byte of ITS row (GemmArgs::A4s / W4s), eight + three per tile, packed four to a register and shifted into byte 0.
  u32x4 b4[FN], a4[4];
  unsigned sa_pk[2] = {0u, 0u}, sw_pk = 0u;
  auto rd_a4 = [&](const char* sA, int i, int h) { return *(const u32x4*)(sA + ((a_off + i * 2048) ^ (h << 6))); };
  auto rd_b4 = [&](const char* sB, int j, int h) { return *(const u32x4*)(sB + ((b_off + j * 2048) ^ (h << 6))); };
  auto mfma4_ip = [](f32x4& c, const u32x4& a, const u32x4& b, unsigned sa_, unsigned sb_) {
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                 : "+v"(c) : "v"(a), "v"(b), "v"(sa_), "v"(sb_));
  };
  auto lo4_scales = [&](int tm_, int tn_) {
    if constexpr (LO >= 2) {
      const int r = lane & 15;
      const unsigned char* as = g.A4s + (size_t)tm_ * BM + grp * 128 + r;
      const unsigned char* ws = g.W4s + (size_t)tn_ * BN + wn * WTN + r;
      sa_pk[0] = (unsigned)as[0] | ((unsigned)as[16] << 8) | ((unsigned)as[32] << 16) | ((unsigned)as[48] << 24);
      sa_pk[1] = (unsigned)as[64] | ((unsigned)as[80] << 8) | ((unsigned)as[96] << 16) | ((unsigned)as[112] << 24);
      sw_pk = (unsigned)ws[0] | ((unsigned)ws[16] << 8) | ((unsigned)ws[32] << 16);
    }
  };
  auto lo4_fill = [&](const char* sA, const char* sB) {
#pragma unroll
    for (int j = 0; j < FN; ++j) b4[j] = rd_b4(sB, j, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) a4[i] = rd_a4(sA, i, 0);
  };
  // LO == 3: the last fp4 tile of a row whose codes end inside its first half (0 < K % 256 <= 128: K = 1152) walks half 0 only -- a
  // COMPILE-TIME choice (the launcher picks the instantiation): as a run-time branch it cost 532 bytes of scratch per lane (behind the
  // branch the register allocator no longer keeps the in-place accumulators where they are) and +340 us per fc1 launch.
  auto lo4_tail = [&](const char* sA) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) mfma4_ip(acc[i][j], b4[j], a4[i], sw_pk >> (8 * j), sa_pk[0] >> (8 * i));
      a4[i] = rd_a4(sA, i + 4, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 4; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) mfma4_ip(acc[i][j], b4[j], a4[i & 3], sw_pk >> (8 * j), sa_pk[1] >> (8 * (i & 3)));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  auto lo4_tile = [&](const char* sA, const char* sB, const char* sAn, const char* sBn, auto look) {
    constexpr bool LOOK = decltype(look)::value;
    // half 0: rows 0-3 (rows 4-7 of the half roll in), rows 4-7 (rows 0-3 of half 1 roll in; B of half 1 behind row 7's MFMAs)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) mfma4_ip(acc[i][j], b4[j], a4[i], sw_pk >> (8 * j), sa_pk[0] >> (8 * i));
      a4[i] = rd_a4(sA, i + 4, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 4; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        mfma4_ip(acc[i][j], b4[j], a4[i & 3], sw_pk >> (8 * j), sa_pk[1] >> (8 * (i & 3)));
        if (i == 7) b4[j] = rd_b4(sB, j, 1);
      }
      a4[i & 3] = rd_a4(sA, i - 4, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // half 1: as the fp8 tile -- rows 0-3 | rows 4-5 | lgkmcnt(0), B_u | rows 6-7 with the look-ahead of stage u + 1 (half 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) mfma4_ip(acc[i][j], b4[j], a4[i], sw_pk >> (8 * j), sa_pk[0] >> (8 * i));
      a4[i] = rd_a4(sA, i + 4, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 4; i < 6; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) mfma4_ip(acc[i][j], b4[j], a4[i & 3], sw_pk >> (8 * j), sa_pk[1] >> (8 * (i & 3)));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (LOOK) {
      a4[0] = rd_a4(sAn, 0, 0);
      a4[1] = rd_a4(sAn, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < FN; ++j) mfma4_ip(acc[6][j], b4[j], a4[2], sw_pk >> (8 * j), sa_pk[1] >> 16);
    if constexpr (LOOK) a4[2] = rd_a4(sAn, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      mfma4_ip(acc[7][j], b4[j], a4[3], sw_pk >> (8 * j), sa_pk[1] >> 24);
      if constexpr (LOOK) b4[j] = rd_b4(sBn, j, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (LOOK) a4[3] = rd_a4(sAn, 3, 0);
  };

  int it = 0, ia = 0;
  if constexpr (A_ONE) {   // ladder rung: the whole walk as ONE K loop (no drain / epilogue / refill between output tiles)
    const int total = ntile * nk;
    for (; it + 1 < total; ++it) {
      const char* sA = smem + ia * A_BYTES;
      ia = ia == NA - 1 ? 0 : ia + 1;
      ktile(sA, smem + ia * A_BYTES, smem + ((it + 1) & 1) * B_BYTES, std::true_type{});
    }
    ktile(smem + ia * A_BYTES, nullptr, nullptr, std::false_type{});
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    epilogue(tm, tn);
    return;
  }
  for (;;) {
    for (int kt = 0; kt + 1 < nk; ++kt, ++it) {
      const char* sA = smem + ia * A_BYTES;
      ia = ia == NA - 1 ? 0 : ia + 1;
      ktile(sA, smem + ia * A_BYTES, smem + ((it + 1) & 1) * B_BYTES, std::true_type{});
    }
    {  // last K tile of the output tile: drain, epilogue, refill from stage u + 1 (landed at B_u)
      const char* sA = smem + ia * A_BYTES;
      ia = ia == NA - 1 ? 0 : ia + 1;
      ktile(sA, nullptr, nullptr, std::false_type{});
      ++it;
      if constexpr (LO >= 2) {   // the fp4 correction K tiles (see lo4_tile)
        lo4_scales(tm, tn);
        lo4_fill(smem + ia * A_BYTES, smem + (it & 1) * B_BYTES);
        for (int k8 = 0; k8 + 1 < nk8; ++k8, ++it) {
          const char* sA8 = smem + ia * A_BYTES;
          ia = ia == NA - 1 ? 0 : ia + 1;
          lo4_tile(sA8, smem + (it & 1) * B_BYTES, smem + ia * A_BYTES, smem + ((it + 1) & 1) * B_BYTES, std::true_type{});
        }
        const char* sA8 = smem + ia * A_BYTES;
        ia = ia == NA - 1 ? 0 : ia + 1;
        if constexpr (LO == 3) lo4_tail(sA8);
        else lo4_tile(sA8, smem + (it & 1) * B_BYTES, nullptr, nullptr, std::false_type{});
        ++it;
      } else if constexpr (LO) {   // the correction K tiles: stage `it` landed at the barrier of the tile above
        lo_fill(smem + ia * A_BYTES, smem + (it & 1) * B_BYTES);
        for (int k8 = 0; k8 + 1 < nk8; ++k8, ++it) {
          const char* sA8 = smem + ia * A_BYTES;
          ia = ia == NA - 1 ? 0 : ia + 1;
          lo_tile(sA8, smem + ia * A_BYTES, smem + ((it + 1) & 1) * B_BYTES, std::true_type{});
        }
        const char* sA8 = smem + ia * A_BYTES;
        ia = ia == NA - 1 ? 0 : ia + 1;
        lo_tile(sA8, nullptr, nullptr, std::false_type{});
        ++it;
      }
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> VALU reads (see mfma16_ip)
      epilogue(tm, tn);
      LATTE_TS(4)
    }
    pos += per;
    if (pos >= cnt) break;
    decode(chunk0 + pos, tm, tn);
    fill(smem + ia * A_BYTES, smem + (it & 1) * B_BYTES);
    LATTE_TS(5)
  }
  trace_out(it);
#undef LATTE_TS
}


// ------------------------------------------------------------------------------------------------
#ifdef LATTE_GEMM_ABLATE
// Two-accumulator-set variant (variant 17, round 5; MEASUREMENT BUILD ONLY -- correct, bit-identical, and slower than variant 11).
// The gated read-modify-write GEMMs pay main loop + epilogue burst: all 256 CUs reach a tile boundary together and idle their
// matrix pipes while the memory system moves every workgroup's 393 KB fp32 patch.  This kernel hides the
// burst: a 256 x 192 tile is two PASSES of 128 rows, all eight consumer waves compute every pass (waves 2 x 4, wave tile 64 x 48 =
// 48 accumulators) and each wave owns TWO accumulator sets -- while set X collects pass p, set Y still holds pass p - 1 and leaves
// one fragment per K step: the first sixteen K steps of a pass are unrolled, step s carries slice s of the drain (residual load
// of fragment s issued, fragment s - RING read-modify-written), so every index is static, every wait a counted vmcnt the compiler
// derives, and a residual load is consumed RING K steps (about 1.5 us) after it was issued.  The sets swap roles by name (the pass
// loop is unrolled by two), an accumulator starts from C = 0 in the first K step of its pass (nothing is zeroed or copied), bias
// and gate of a pass are staged by the producers (4-byte LDS-DMA pieces into a slice per pass parity and wave column).  Rings of
// NA = 5 A stages (128 rows) and NB = 3 B stages, one workgroup barrier per K step, K order and epilogue arithmetic of variants
// 8 / 10 / 11: same bits.
// MEASURED (tools/alt_probe.py, profiles/r5_gated_overlap_*.log; XL/2 B = 8, f16): the drain is hidden, and the launch is slower --
// out-projection 107 us against 104 stand-alone and 135 against 123 in the forward, fc2 374 against 316 / 377 against 307.  A first
// form of the idea (the two consumer groups of variant 11 strictly alternating, one group's K loop beside the other's sixteen-slice
// epilogue, i.e. one MFMA wave per SIMD; built, bit-identical, traced in profiles/r5_gated_overlap_v16_trace.log, not kept) landed on
// the SAME times, so it is not the wave arrangement: a pass
// fetches the W tile of a K step once per 128 rows instead of once per 256, 40 KB of LDS DMA per 48 MFMAs per SIMD against 56 KB per
// 96 -- 1.43 x the operand bytes per MFMA through a path that delivers about 39 B / clock / CU, and that
// path, not the epilogue, is what bounds the 192-wide kernels (56 KB / 39 = 1436 clocks per K step against 1536 of MFMA work).
// Any scheme that needs a second accumulator set halves the tile a workgroup can hold and pays this.
template <int EPI, int DT, int TAG>
__global__ void __launch_bounds__(768) gemm_dacc_kernel(GemmArgs g) {
  constexpr int BM = 256, HM = 128, BN = 192, FN = 3, FM = 4, WTN = 48;
  constexpr int A_BYTES = HM * 128, B_BYTES = BN * 128;
  constexpr int NA = 5, NB = 3;
  constexpr int B_BASE = NA * A_BYTES;
  constexpr int SIDE_BASE = B_BASE + NB * B_BYTES;   // [pass parity][wave column][64 bias | 64 gate] floats: 2 x 4 x 512 B, staged by the producers
  constexpr int AH_INSTR = 4, BG_INSTR = 6;
  constexpr int GROUP_M = 8;
  constexpr int S = 16;                         // unrolled K steps at the head of a pass (K / 64 >= S + 2)
  constexpr int RING = 3;                       // residual fragments in flight per wave
  constexpr int NF = FM * FN;                   // 12 fragments per wave and pass

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = g.K;
  const unsigned row_bytes = (unsigned)K * 2u;

  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = g.N / BN, nwg = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int chunk0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int cnt = q + (xcd < r ? 1 : 0);
  if (slot >= cnt) return;
  const int group_m = g.group_m > 0 ? g.group_m : GROUP_M;
  auto decode = [&](int wg, int& tm, int& tn) {
    const int per_group = group_m * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * group_m;
    const int gsz = min(tiles_m - first_m, group_m);
    const int in_group = wg - group * per_group;
    tm = first_m + in_group % gsz;
    tn = in_group / gsz;
  };
  const int nk = K / 64;
  const int ntile = (cnt - slot + per - 1) / per;
  const int NP = 2 * ntile;

  if (wave >= 8) {
    // ================================ producer ================================
    const int pw = wave - 8;
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (unsigned)tiles_m * BM * row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (unsigned)g.N * row_bytes, 0x00020000);
    const int lrow = lane >> 3, cpos = lane & 7;
    const unsigned voff = (unsigned)lrow * row_bytes + (unsigned)((cpos ^ (((pw * 8 + lrow) >> 1) & 7)) * 16);
    unsigned step32 = 32u * row_bytes;
    asm volatile("" : "+s"(step32));
    struct Walk { int pos, tm, tn, kt, half; };
    auto dma_a = [&](const Walk& w, int stg) {
      char* sA = smem + stg * A_BYTES + pw * 1024;
      const unsigned so = (unsigned)(w.tm * BM + w.half * HM + pw * 8) * row_bytes + (unsigned)w.kt * 128u;
#pragma unroll
      for (int j = 0; j < AH_INSTR; ++j) pw_bload_lds16(rsA, sA + j * 4 * 1024, voff, so + (unsigned)j * step32);
    };
    auto dma_b = [&](const Walk& w, int stg) {
      char* sB = smem + B_BASE + stg * B_BYTES + pw * 1024;
      const unsigned so = (unsigned)(w.tn * BN + pw * 8) * row_bytes + (unsigned)w.kt * 128u;
#pragma unroll
      for (int j = 0; j < BG_INSTR; ++j) pw_bload_lds16(rsB, sB + j * 4 * 1024, voff, so + (unsigned)j * step32);
    };
    auto advance = [&](Walk& w) {
      if (++w.kt == nk) {
        w.kt = 0;
        w.half ^= 1;
        if (w.half == 0) {
          w.pos += per;
          if (w.pos < cnt) decode(chunk0 + w.pos, w.tm, w.tn);
        }
      }
    };
    // bias / gate of a pass for the wave column wn = pw: two 4-byte LDS-DMA pieces (lanes 48-63 land in the padding) into the
    // slice of the pass's parity, issued when the pass starts and read by the consumers one pass later
    const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)g.bias, 0, (unsigned)g.N * 4u, 0x00020000);
    const int n_samples = (g.M - 1) / g.rows_per_sample + 1;
    const __amdgpu_buffer_rsrc_t rsGate =
        __builtin_amdgcn_make_buffer_rsrc((void*)g.gate, 0, ((unsigned)(n_samples - 1) * (unsigned)g.gate_stride + (unsigned)g.N) * 4u, 0x00020000);
    auto dma_side = [&](int pp) {
      int tm_, tn_;
      decode(chunk0 + slot + (pp >> 1) * per, tm_, tn_);
      const int row0 = min(tm_ * BM + (pp & 1) * HM, g.M - 1);
      char* dst = smem + SIDE_BASE + (pp & 1) * 2048 + pw * 512;
      const unsigned col = (unsigned)(tn_ * BN + pw * WTN) * 4u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBias, (lds_void_pw*)dst, 4, (unsigned)lane * 4u, col, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsGate, (lds_void_pw*)(dst + 256), 4, (unsigned)lane * 4u,
                                               (unsigned)(row0 / g.rows_per_sample) * (unsigned)g.gate_stride * 4u + col, 0, 0);
    };
    const int U = NP * nk;
    Walk wb{slot, 0, 0, 0, 0};
    decode(chunk0 + slot, wb.tm, wb.tn);
    Walk wa = wb;
    int sa = 0, sb = 0;
    dma_side(0);
#pragma unroll
    for (int i = -NA; i < 0; ++i) {
      if (i + NB >= 0) { dma_b(wb, sb); advance(wb); sb = sb == NB - 1 ? 0 : sb + 1; }
      dma_a(wa, sa); advance(wa); sa = sa == NA - 1 ? 0 : sa + 1;
    }
    constexpr int YOUNGER = (NB - 2) * (AH_INSTR + BG_INSTR) + AH_INSTR;
    wait_vm<YOUNGER>();
    __builtin_amdgcn_s_barrier();   // P
    // (the two side pieces of a pass are issued behind the A stage of an iteration: wherever they are younger than the stage a
    //  wait confirms, the count below is merely two too strict)
    int pp = 0, kk = 0;   // pass / K step of iteration u
    for (int u = 0; u < U; ++u) {
      if (u + NA <= U) wait_vm<YOUNGER>(); else wait_vm<0>();
      __builtin_amdgcn_s_barrier();   // B_u
      if (u + NB < U) { dma_b(wb, sb); advance(wb); sb = sb == NB - 1 ? 0 : sb + 1; }
      if (u + NA < U) { dma_a(wa, sa); advance(wa); sa = sa == NA - 1 ? 0 : sa + 1; }
      if (++kk == nk) {
        kk = 0;
        if (++pp < NP) dma_side(pp);   // B_u with u the last step of pass pp - 1: the drain of pass pp - 2 (same parity) ended long ago
      }
    }
    return;
  }

  // ================================ consumer ================================
  const int wm = wave >> 2, wn = wave & 3;
  const int sw = (lane >> 1) & 7;
  const int chunkb = ((lane >> 4) ^ sw) * 16;
  const int a_off = (wm * 64 + (lane & 15)) * 128 + chunkb;
  const int b_off = B_BASE + (wn * WTN + (lane & 15)) * 128 + chunkb;
  const float* d_side = nullptr;   // [64 bias | 64 gate] slice of the pass being drained

  f32x4 accX[FM][FN], accY[FM][FN];   // never zeroed: the first K step of a pass starts every accumulator from C = 0

  // drain state of the pass that left last (scalars) and the per-lane part of a fragment address
  const unsigned n4 = (unsigned)g.N * 4u;
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (unsigned)g.M * n4, 0x00020000);
  const unsigned srow16 = 16u * n4;
  unsigned d_sbase = 0;
  u32x4 qr[RING];
  float4 d_b4, d_g4;   // bias / gate of the fragment the next slice writes: read from the LDS slice inside the K step, in front of the
                       // lgkmcnt(0) that precedes its barrier anyway (a read behind the step's fragment look-ahead would wait for all of it)
  auto side_read = [&](auto tt) {
    constexpr int t = decltype(tt)::value;
    if constexpr (t >= RING && t - RING < NF) {
      constexpr int j = (t - RING) % FN;
      int le = lane;
      asm volatile("" : "+v"(le));
      d_b4 = *(const float4*)(d_side + j * 16 + (le >> 4) * 4);
      d_g4 = *(const float4*)(d_side + 64 + j * 16 + (le >> 4) * 4);
    }
  };
#pragma unroll
  for (int k = 0; k < RING; ++k) asm volatile("" : "=v"(qr[k]));   // "defined": no initial value to keep alive across the first pass
  // slice t of the drain of `prv`: fragment t - RING is read-modify-written, the residual of fragment t is requested
  auto drain_slice = [&](f32x4 (&prv)[FM][FN], auto tt) {
    constexpr int t = decltype(tt)::value;
    int le = lane;
    asm volatile("" : "+v"(le));
    const unsigned voff = (unsigned)(le & 15) * n4 + (unsigned)(le >> 4) * 16u;
    auto soff = [&](int f) -> unsigned { return d_sbase + (unsigned)(f / FN) * srow16 + (unsigned)((f % FN) * 64); };
    if constexpr (t >= RING && t - RING < NF) {
      constexpr int f = t - RING, i = f / FN, j = f % FN;
      const float4 b4 = d_b4, g4 = d_g4;
      f32x4 rr = __builtin_bit_cast(f32x4, qr[f % RING]);
      rr[0] += g4.x * (prv[i][j][0] + b4.x);
      rr[1] += g4.y * (prv[i][j][1] + b4.y);
      rr[2] += g4.z * (prv[i][j][2] + b4.z);
      rr[3] += g4.w * (prv[i][j][3] + b4.w);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rr), rsO, voff, soff(f), 0);
    }
    if constexpr (t < NF) qr[t % RING] = __builtin_amdgcn_raw_buffer_load_b128(rsO, voff, soff(t), 0);
  };

  auto fill = [&](u32x4 (&bf)[2][FN], u32x4 (&af)[FM], const char* sA, const char* sB) {
#pragma unroll
    for (int j = 0; j < FN; ++j) bf[0][j] = *(const u32x4*)(sB + (b_off + j * 2048));
#pragma unroll
    for (int i = 0; i < FM; ++i) af[i] = *(const u32x4*)(sA + (a_off + i * 2048));
#pragma unroll
    for (int j = 0; j < FN; ++j) bf[1][j] = *(const u32x4*)(sB + ((b_off + j * 2048) ^ 64));
  };
  // One K step of the rolling pipeline on four fragment rows; LOOK = the fragments of stage u + 1 roll in behind the barrier.
  auto kstep = [&](f32x4 (&acc)[FM][FN], u32x4 (&bf)[2][FN], u32x4 (&af)[FM], const char* sA, const char* sAn, const char* sBn, auto look, auto first, auto slice) {
    constexpr bool LOOK = decltype(look)::value, FIRST = decltype(first)::value;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if constexpr (FIRST) mfma16_first<DT>(acc[i][j], bf[0][j], af[i]);
        else mfma16_ip<DT>(acc[i][j], bf[0][j], af[i]);
      }
      af[i] = *(const u32x4*)(sA + ((a_off + i * 2048) ^ 64));
      __builtin_amdgcn_sched_barrier(0);
    }
    side_read(slice);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) mfma16_ip<DT>(acc[i][j], bf[1][j], af[i]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (LOOK) {
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[0][j] = *(const u32x4*)(sBn + (b_off + j * 2048));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 2; i < FM; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) mfma16_ip<DT>(acc[i][j], bf[1][j], af[i]);
      if constexpr (LOOK) af[i - 2] = *(const u32x4*)(sAn + (a_off + (i - 2) * 2048));
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (LOOK) {
      af[2] = *(const u32x4*)(sAn + (a_off + 2 * 2048));
      af[3] = *(const u32x4*)(sAn + (a_off + 3 * 2048));
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[1][j] = *(const u32x4*)(sBn + ((b_off + j * 2048) ^ 64));
    }
  };

  // Pass p of the workgroup's stream: accumulate into `cur`; DRAIN: the previous pass leaves `prv` meanwhile (compile-time, so that
  // every slice is unconditional straight-line code and the compiler's vmcnt waits stay counted).  Then stage the bias / gate slice
  // and the address of `cur` for ITS drain.
  auto pass = [&](f32x4 (&cur)[FM][FN], f32x4 (&prv)[FM][FN], int p, auto drain) {
    constexpr bool DRAIN = decltype(drain)::value;
    int tm, tn;
    decode(chunk0 + slot + (p >> 1) * per, tm, tn);
    const int half = p & 1;
    const int u0 = p * nk;
    int ia = u0 % NA, ib = u0 % NB;
    u32x4 bf[2][FN], af[FM];   // fragment registers: live inside a pass only
    fill(bf, af, smem + ia * A_BYTES, smem + ib * B_BYTES);
    auto step = [&](auto first, auto slice) {   // slice: the drain slice that follows this K step (-1: none)
      const char* sA = smem + ia * A_BYTES;
      ia = ia == NA - 1 ? 0 : ia + 1;
      ib = ib == NB - 1 ? 0 : ib + 1;
      kstep(cur, bf, af, sA, smem + ia * A_BYTES, smem + ib * B_BYTES, std::true_type{}, first, slice);
    };
    constexpr std::integral_constant<int, -1> NONE{};
    if constexpr (DRAIN) {
      // the first S K steps carry the drain, one slice each
#define LATTE_DACC_STEP(T)                                                   \
      step(std::integral_constant<bool, T == 0>{}, std::integral_constant<int, T>{});   \
      drain_slice(prv, std::integral_constant<int, T>{});
      LATTE_DACC_STEP(0) LATTE_DACC_STEP(1) LATTE_DACC_STEP(2) LATTE_DACC_STEP(3)
      LATTE_DACC_STEP(4) LATTE_DACC_STEP(5) LATTE_DACC_STEP(6) LATTE_DACC_STEP(7)
      LATTE_DACC_STEP(8) LATTE_DACC_STEP(9) LATTE_DACC_STEP(10) LATTE_DACC_STEP(11)
      LATTE_DACC_STEP(12) LATTE_DACC_STEP(13) LATTE_DACC_STEP(14) LATTE_DACC_STEP(15)
#undef LATTE_DACC_STEP
      static_assert(S == 16 && NF + RING <= S, "drain slices must fit the unrolled head of a pass");
      for (int kt = S; kt + 1 < nk; ++kt) step(std::false_type{}, NONE);
    } else {
      step(std::true_type{}, NONE);
      for (int kt = 1; kt + 1 < nk; ++kt) step(std::false_type{}, NONE);
    }
    kstep(cur, bf, af, smem + ia * A_BYTES, nullptr, nullptr, std::false_type{}, std::false_type{}, NONE);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> VALU reads (see mfma16_ip)
    // what the drain of `cur` needs: rows >= M lie beyond the descriptor (loads return zeros, stores are dropped)
    d_sbase = (unsigned)(tm * BM + half * HM + wm * 64) * n4 + (unsigned)(tn * BN + wn * WTN) * 4u;
    d_side = (const float*)(smem + SIDE_BASE + half * 2048 + wn * 512);
  };

  __builtin_amdgcn_s_barrier();   // P: stage 0 has landed
  pass(accX, accY, 0, std::false_type{});
  for (int p = 1; p < NP; p += 2) {
    pass(accY, accX, p, std::true_type{});
    if (p + 1 < NP) pass(accX, accY, p + 1, std::true_type{});
  }
  // the last pass (in accY) leaves without a K loop beside it
  {
#define LATTE_DACC_TAIL(T) side_read(std::integral_constant<int, T>{}); drain_slice(accY, std::integral_constant<int, T>{});
    LATTE_DACC_TAIL(0) LATTE_DACC_TAIL(1) LATTE_DACC_TAIL(2) LATTE_DACC_TAIL(3) LATTE_DACC_TAIL(4) LATTE_DACC_TAIL(5)
    LATTE_DACC_TAIL(6) LATTE_DACC_TAIL(7) LATTE_DACC_TAIL(8) LATTE_DACC_TAIL(9) LATTE_DACC_TAIL(10) LATTE_DACC_TAIL(11)
    LATTE_DACC_TAIL(12) LATTE_DACC_TAIL(13) LATTE_DACC_TAIL(14)
#undef LATTE_DACC_TAIL
  }
}

#endif   // LATTE_GEMM_ABLATE

template <int DT>
int launch_pw_dt(const GemmArgs& a, int epi, int roll, hipStream_t st) {
  constexpr int LDS = 3 * 256 * 128 + 2 * 192 * 128 + 8 * 16 * 112;   // A ring + B ring + epilogue patches (rolling kernel)
  const int tiles = ((a.M + 255) / 256) * (a.N / 192);
  const int nblk = tiles >= 256 ? 256 : (tiles + 7) / 8 * 8;
  dim3 grid(nblk), block(768);
#define LATTE_PW_CASE(E, T)                                                                          \
  {                                                                                                  \
    if (roll) {                                                                                      \
      auto kern = gemm_pwr_kernel<E, DT, T>;                                                         \
      static std::atomic<uint64_t> attr_done{0};                                                     \
      if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;               \
      hipLaunchKernelGGL(kern, grid, block, LDS, st, a);                                             \
    } else {                                                                                         \
      auto kern = gemm_pw_kernel<E, DT, T>;                                                          \
      static std::atomic<uint64_t> attr_done{0};                                                     \
      if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;               \
      hipLaunchKernelGGL(kern, grid, block, LDS, st, a);                                             \
    }                                                                                                \
  }
#ifdef LATTE_GEMM_ABLATE
  if (roll == 1 && a.rmw_mode > 0 && epi == EPI_GATE_RES_F32 && !a.A8) {   // the ladder rungs (GemmArgs::rmw_mode = ABL bits; LATTE_PWR_ABL)
#define LATTE_ABL_CASE(MASK)                                                                        \
    case MASK: {                                                                                    \
      auto kern = gemm_pwr_kernel<EPI_GATE_RES_F32, DT, 1, false, MASK>;                            \
      static std::atomic<uint64_t> attr_done{0};                                                    \
      if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;              \
      hipLaunchKernelGGL(kern, grid, block, LDS, st, a);                                            \
      break;                                                                                        \
    }
    switch (a.rmw_mode) {
      LATTE_ABL_CASE(1) LATTE_ABL_CASE(3) LATTE_ABL_CASE(7) LATTE_ABL_CASE(15) LATTE_ABL_CASE(31) LATTE_ABL_CASE(63)
      LATTE_ABL_CASE(4) LATTE_ABL_CASE(16) LATTE_ABL_CASE(32) LATTE_ABL_CASE(5) LATTE_ABL_CASE(35) LATTE_ABL_CASE(64) LATTE_ABL_CASE(128) LATTE_ABL_CASE(256)
      default: return fail(LATTE_ERR_INVALID, "gemm (ladder): ablation mask not instantiated");
    }
#undef LATTE_ABL_CASE
    LATTE_HIP(hipGetLastError());
    return LATTE_OK;
  }
  if (roll == 3) {
    constexpr int LDS_DACC = 5 * 128 * 128 + 3 * 192 * 128 + 2 * 4 * 512;   // 5 A stages of 128 rows, 3 B stages, bias / gate slices
    if (a.K < 1152 || a.rows_per_sample % 128 != 0 || (uint64_t)a.M * a.N * 4 >= (1ull << 32))
      return fail(LATTE_ERR_INVALID, "gemm (two-accumulator-set kernel): need K >= 1152, rows_per_sample % 128 == 0, M N 4 < 4 GiB");
    if (epi != EPI_GATE_RES_F32) return fail(LATTE_ERR_INVALID, "gemm (two-accumulator-set kernel): gated read-modify-write epilogue only");
    if (a.tag == 1) {
      auto kern = gemm_dacc_kernel<EPI_GATE_RES_F32, DT, 1>;
      static std::atomic<uint64_t> attr_done{0};
      if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS_DACC, attr_done)) return rc_;
      hipLaunchKernelGGL(kern, grid, block, LDS_DACC, st, a);
    } else {
      auto kern = gemm_dacc_kernel<EPI_GATE_RES_F32, DT, 0>;
      static std::atomic<uint64_t> attr_done{0};
      if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS_DACC, attr_done)) return rc_;
      hipLaunchKernelGGL(kern, grid, block, LDS_DACC, st, a);
    }
    LATTE_HIP(hipGetLastError());
    return LATTE_OK;
  }
#endif
  if (a.A4) {   // fp4 correction pass (GemmArgs::A4 / W4 + row scales): the rolling kernel's LO = 2 instantiations
    if (!roll || !a.W4 || !a.A4s || !a.W4s || !(epi == EPI_GATE_RES_F32 || epi == EPI_BIAS_GELU_H16) || DT != LATTE_DTYPE_F16)
      return fail(LATTE_ERR_INVALID, "gemm: the fp4 correction pass needs the rolling kernel, f16 operands, both code images with their row scales, and the gated / GELU epilogue");
    if constexpr (DT == LATTE_DTYPE_F16) {
      const bool half_tail = (a.K & 255) != 0 && (a.K & 255) <= 128;   // the last code tile ends inside its first half: LO = 3
#define LATTE_PW_LO4(E, L)                                                                   \
  {                                                                                          \
    auto kern = gemm_pwr_kernel<E, DT, 0, L>;                                                \
    static std::atomic<uint64_t> attr_done{0};                                               \
    if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;         \
    hipLaunchKernelGGL(kern, grid, block, LDS, st, a);                                       \
  }
      if (epi == EPI_GATE_RES_F32) { if (half_tail) LATTE_PW_LO4(EPI_GATE_RES_F32, 3) else LATTE_PW_LO4(EPI_GATE_RES_F32, 2) }
      else { if (half_tail) LATTE_PW_LO4(EPI_BIAS_GELU_H16, 3) else LATTE_PW_LO4(EPI_BIAS_GELU_H16, 2) }
#undef LATTE_PW_LO4
    }
    LATTE_HIP(hipGetLastError());
    return LATTE_OK;
  }
  if (a.A8) {   // correction pass (GemmArgs::A8 / W8): the rolling kernel's LO instantiations, out-projection and fc1 of guided calls
    if (!roll || !a.W8 || a.K % 128 != 0 || !(epi == EPI_GATE_RES_F32 || epi == EPI_BIAS_GELU_H16) || DT != LATTE_DTYPE_F16)
      return fail(LATTE_ERR_INVALID, "gemm (correction pass): rolling kernel, f16 operands, K % 128 == 0, gated-residual or GELU epilogue only");
    if constexpr (DT == LATTE_DTYPE_F16) {
      if (epi == EPI_GATE_RES_F32) {
        auto kern = gemm_pwr_kernel<EPI_GATE_RES_F32, DT, 0, 1>;
        static std::atomic<uint64_t> attr_done{0};
        if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;
        hipLaunchKernelGGL(kern, grid, block, LDS, st, a);
      } else {
        auto kern = gemm_pwr_kernel<EPI_BIAS_GELU_H16, DT, 0, 1>;
        static std::atomic<uint64_t> attr_done{0};
        if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;
        hipLaunchKernelGGL(kern, grid, block, LDS, st, a);
      }
    }
    LATTE_HIP(hipGetLastError());
    return LATTE_OK;
  }
  if (epi == EPI_GATE_RES_F32 && a.tag == 1) LATTE_PW_CASE(EPI_GATE_RES_F32, 1)
  else if (epi == EPI_GATE_RES_F32) LATTE_PW_CASE(EPI_GATE_RES_F32, 0)
  else if (epi == EPI_BIAS_F32) LATTE_PW_CASE(EPI_BIAS_F32, 0)
  else if (epi == EPI_BIAS_H16) LATTE_PW_CASE(EPI_BIAS_H16, 0)
  else if (epi == EPI_BIAS_GELU_H16) LATTE_PW_CASE(EPI_BIAS_GELU_H16, 0)
  else if (epi == EPI_BIAS_GELU_DUAL_H16 || epi == EPI_DGELU_H16) {   // training epilogues: rolling kernel only
    if (!roll || !a.aux) return fail(LATTE_ERR_INVALID, "gemm (producer-wave kernel): the GELU training epilogues need the rolling kernel and GemmArgs::aux");
    if (epi == EPI_BIAS_GELU_DUAL_H16) {
      auto kern = gemm_pwr_kernel<EPI_BIAS_GELU_DUAL_H16, DT, 0>;
      static std::atomic<uint64_t> attr_done{0};
      if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;
      hipLaunchKernelGGL(kern, grid, block, LDS, st, a);
    } else {
      auto kern = gemm_pwr_kernel<EPI_DGELU_H16, DT, 0>;
      static std::atomic<uint64_t> attr_done{0};
      if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;
      hipLaunchKernelGGL(kern, grid, block, LDS, st, a);
    }
  }
#ifdef LATTE_GEMM_ABLATE
  else if (epi == EPI_ABLATE_TRACE) LATTE_PW_CASE(EPI_ABLATE_TRACE, 0)
#endif
  else return fail(LATTE_ERR_INVALID, "gemm (producer-wave kernel): unknown epilogue");
#undef LATTE_PW_CASE
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

}  // namespace

bool gemm_lo4_ok(int M, int N, int K) {
  return M > 0 && N % 192 == 0 && K % 64 == 0 && K >= 128 && (uint64_t)((M + 255) / 256 * 256) * K * 2 < (1ull << 32) &&
         (uint64_t)N * K * 2 < (1ull << 32);
}
bool gemm_lo8_ok(int M, int N, int K) {
  return M > 0 && N % 192 == 0 && K % 128 == 0 && K >= 128 && (uint64_t)((M + 255) / 256 * 256) * K * 2 < (1ull << 32) &&
         (uint64_t)N * K * 2 < (1ull << 32);
}

// variant 10: limits of the 32-bit buffer offsets and K >= 128 as for the persistent kernel; whole 192-wide tile columns
int launch_gemm_pw(const GemmArgs& a, int epi, int dtype, int roll, hipStream_t st) {
  if (a.N % 192 != 0 || a.K % 64 != 0 || a.K < 128 || a.M <= 0)
    return fail(LATTE_ERR_INVALID, "gemm (producer-wave kernel): need N % 192 == 0, K % 64 == 0, K >= 128");
  if ((uint64_t)((a.M + 255) / 256 * 256) * a.K * 2 >= (1ull << 32) || (uint64_t)a.N * a.K * 2 >= (1ull << 32))
    return fail(LATTE_ERR_INVALID, "gemm (producer-wave kernel): operand exceeds the 4 GiB buffer-offset range");
  if (dtype == LATTE_DTYPE_BF16) return launch_pw_dt<LATTE_DTYPE_BF16>(a, epi, roll, st);
  if (dtype == LATTE_DTYPE_F16) return launch_pw_dt<LATTE_DTYPE_F16>(a, epi, roll, st);
  return fail(LATTE_ERR_INVALID, "gemm: unknown dtype");
}

}  // namespace latte
