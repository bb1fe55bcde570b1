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
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "mfma_util.h"

namespace latte {
namespace {

// GELU(tanh approximation) = x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3)   (latte.py:170)
// = x / (1 + exp2(x (a + b x^2))) with a = -2 log2(e) sqrt(2/pi), b = 0.044715 a: 3 mul + 1 fma + 1 add + exp2 + rcp
__device__ __forceinline__ float gelu_tanh(float x) {
  const float p = __builtin_fmaf(x * x, -0.10294324f, -2.3022082f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p * x));
}

// One accumulator fragment -> memory.  Lane holds 4 consecutive columns n..n+3 of row m.
template <int EPI, int DT>
__device__ __forceinline__ void epilogue_store(const GemmArgs& g, const f32x4& a, int m, int n, const float* gate_row) {
  const float4 b4 = *(const float4*)(g.bias + n);
  float v0 = a[0] + b4.x, v1 = a[1] + b4.y, v2 = a[2] + b4.z, v3 = a[3] + b4.w;
  const size_t o = (size_t)m * g.N + n;
  if constexpr (EPI == EPI_ABLATE_NOSTORE || EPI >= EPI_ABLATE_NODMA) {
    asm volatile("" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3));
  } else if constexpr (EPI == EPI_BIAS_RES_H16) {
    const u32x2 r2 = *(const u32x2*)(g.res + o);
    if constexpr (DT == LATTE_DTYPE_BF16) {
      v0 += __builtin_bit_cast(float, r2[0] << 16); v1 += __builtin_bit_cast(float, r2[0] & 0xffff0000u);
      v2 += __builtin_bit_cast(float, r2[1] << 16); v3 += __builtin_bit_cast(float, r2[1] & 0xffff0000u);
    } else {
      // (scalar copies first: bit-casting the vector elements directly was miscompiled into re-using element 0)
      const unsigned int lo = r2[0], hi = r2[1];
      v0 += (float)__builtin_bit_cast(_Float16, (unsigned short)(lo & 0xffffu));
      v1 += (float)__builtin_bit_cast(_Float16, (unsigned short)(lo >> 16));
      v2 += (float)__builtin_bit_cast(_Float16, (unsigned short)(hi & 0xffffu));
      v3 += (float)__builtin_bit_cast(_Float16, (unsigned short)(hi >> 16));
    }
    u32x2 p = {pack2<DT>(v0, v1), pack2<DT>(v2, v3)};
    *(u32x2*)((half_t*)g.out + o) = p;
  } else if constexpr (EPI == EPI_BIAS_H16 || EPI == EPI_BIAS_GELU_H16) {
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
  // split K (weight-gradient GEMMs: few output tiles, a very long contraction): blockIdx.y owns K range
  // [y k_chunk, (y + 1) k_chunk) and writes its partial product to out + y * split_stride (EPI_BIAS_F32, bias = 0)
  int k_begin = 0, k_len = K;
  if (g.k_chunk > 0) {
    k_begin = blockIdx.y * g.k_chunk;
    k_len = min(g.k_chunk, K - k_begin);
    g.out = (void*)((float*)g.out + (size_t)blockIdx.y * (size_t)g.split_stride);
  }

  // ---- staging addresses: lane i of a wave-instruction fills LDS bytes [16 i, 16 i + 16) of an
  // 8-row group; it must therefore FETCH the chunk that belongs at that (row, chunk-position).
  const int lrow = lane >> 3, cpos = lane & 7;
  const int srow = wave * 8 + lrow;                       // tile row handled by instruction 0
  const int schunk = cpos ^ ((srow >> 1) & 7);            // same for every instruction (NW*8 % 16 == 0)
  const half_t* a_src = g.A + (size_t)(m0 + srow) * K + schunk * 8 + k_begin;
  const half_t* b_src = g.W + (size_t)(n0 + srow) * K + schunk * 8 + k_begin;
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

  const int nk = k_len / 64;
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
    for (int j = 0; j < FN; ++j) epilogue_store<EPI, DT>(g, acc[i][j], m, ncol + j * 16, gate_row);
  }
}

// ------------------------------------------------------------------------------------------------
// Small-M kernel (variants 12 / 13): 128 x 144 tile, twelve MFMA waves (4 along M x 3 along N, wave tile 32 x 48), four LDS
// stages.  At B = 1 (M = 4096 rows) the gated GEMMs with N = 1152 have 96 tiles of 256 x 192 for 256 CUs; 128 x 144 gives
// exactly 32 x 8 = 256 tiles, one per CU, and N = 1152 = 8 x 144 -- the tile shape that minimises the operand bytes a CU has to
// pull (BM + BN at BM x BN = M N / 256), which is what bounds these launches: a CU keeps three K tiles (104 KB) in flight and
// gets 47 GB/s out of them whatever issues the DMA (measured: fc2 57 / 55 / 54 us with the DMA after the barrier, between the
// MFMA groups, in producer waves; warming the L2 eight K tiles ahead with one-dword loads made it 68).  (The same idea for fc1
// at B = 1 -- a 256 x 288 tile, 16 x 16 = 256 tiles, eight waves, K tiles of 32 in a four-stage ring, 64-byte LDS rows with the
// chunk position kg ^ (row & 8 ? 3 : 0), conflict-free by PMC -- was built, passed the GEMM tests and TIED the 256 x 192 12-wave
// kernel inside the forward: 54.1 against 54.3 us; removed.)  One tile per workgroup
// means nothing hides the pipeline fill of a persistent walk, so the ring is deep instead: 4 stages of (128 + 144) x 128 B =
// 34 KB, the DMA of K tiles kt + 1 .. kt + 3 in flight while K tile kt is multiplied (counted vmcnt, one barrier per K tile).
// A K tile is 34 one-KB DMA pieces (16 A row-groups, then 18 W row-groups: the LDS image of a stage is contiguous in that
// order).  PRODUCER = 0 (variant 12, 768 threads): wave w issues pieces w, w + 12, w + 24.  PRODUCER = 1 (variant 13, 1024
// threads, the default for the shapes it takes): waves 12-15, one per SIMD, issue everything -- pieces pw + 4 j, nine per wave
// -- and the MFMA waves issue none.  A wave whose last piece index runs past 33 repeats its previous piece (same bytes to the
// same place), so that every issuing wave has the same number of pieces per K tile in flight and one vmcnt constant serves all;
// 12 and 4 are even, so the source-side swizzle of a wave's pieces is the same.  24 accumulators per lane.
// BM = 256 (variants 18 / 19, round 6): the same kernel on a 256 x 144 tile -- wave tile 64 x 48 (48 accumulators), THREE stages of
// (256 + 144) x 128 B = 50 KB, two K tiles in flight.  For the shapes whose 256 x 192 tiling leaves the last round of the chip half
// empty: fc1 at B = 1 (M = 4096, N = 4608: 384 tiles of 256 x 192 = 1.5 rounds -> 512 of 256 x 144 = 2 full rounds of 3/4 the work) and
// the gated GEMMs at B = 2 (M = 8192, N = 1152: 192 tiles on 256 CUs -> 256).  With the four producer waves the register budget is 128:
// no room for the residual prefetch, so the gated epilogue of variant 19 is a plain load / add / store behind the K loop; variant 18
// (768 threads, 170 registers) keeps the prefetch.
template <int EPI, int DT, int PRODUCER, int BM = 128>
__global__ void __launch_bounds__(PRODUCER ? 1024 : 768) gemm_n144_kernel(GemmArgs g) {
  constexpr int BN = 144, NS = BM == 128 ? 4 : 3, LOOK = NS - 1;
  constexpr int FM = BM / 64, FN = 3, WTM = BM / 4;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128, NPIECE = (BM + BN) / 8, APIECE = BM / 8;
  constexpr int NISSUE = PRODUCER ? 4 : 12, PER_WAVE = (NPIECE + NISSUE - 1) / NISSUE;   // issuing waves, pieces per wave
  constexpr bool PREFETCH = EPI == EPI_GATE_RES_F32 && (BM == 128 || !PRODUCER);          // residual / gate / bias ahead of the K loop
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int tm, tn;
  tile_coords((g.M + BM - 1) / BM, g.N / BN, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int K = g.K, nk = K / 64;

  // ---- DMA: piece p < APIECE = A rows 8 p .. 8 p + 7, else W rows 8 (p - APIECE) ..; lane -> (row lane >> 3, chunk swizzled)
  const bool issuer = PRODUCER ? wave >= 12 : true;
  const half_t* src[PER_WAVE];
  int dst_off[PER_WAVE];
  if (issuer) {
    const int iw = PRODUCER ? wave - 12 : wave;
    const int lrow = lane >> 3, cpos = lane & 7;
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
      int p = iw + NISSUE * j;
      if (p >= NPIECE) p -= NISSUE;
      const int grp8 = p < APIECE ? p : p - APIECE;
      const int row = grp8 * 8 + lrow;
      const int schunk = cpos ^ ((row >> 1) & 7);
      // A rows beyond M exist (row-padded operand); W rows are all inside N (N % 144 == 0)
      src[j] = (p < APIECE ? g.A + (size_t)(m0 + row) * K : g.W + (size_t)(n0 + row) * K) + schunk * 8;
      dst_off[j] = p * 1024;
    }
  }
  auto stage = [&](int kt, int slot) __attribute__((always_inline)) {
    char* s = smem + slot * STAGE;
    const int koff = kt * 64;
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) glds16(src[j] + koff, s + dst_off[j]);
  };
  // own pieces of K tile kt landed (those of the LOOK - 1 newer K tiles may stay in flight)
  auto wait_tile = [&](int kt) __attribute__((always_inline)) {
    if (kt + 2 < nk && LOOK >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_WAVE) : "memory");
    else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto next_slot = [](int s_) { return s_ == NS - 1 ? 0 : s_ + 1; };

  if constexpr (PRODUCER) {
    if (wave >= 12) {
      int islot = 0;
#pragma unroll
      for (int t = 0; t < LOOK; ++t)
        if (nk > t) { stage(t, islot); islot = next_slot(islot); }
      for (int kt = 0; kt < nk; ++kt) {
        wait_tile(kt);
        __builtin_amdgcn_s_barrier();   // K tile kt is complete for the MFMA waves; every read of K tile kt - 1 has retired
        if (kt + LOOK < nk) { stage(kt + LOOK, islot); islot = next_slot(islot); }
      }
      return;
    }
  }

  const int wm = wave / 3, wn = wave - wm * 3;
  const int frow = lane & 15;
  const int sw = (lane >> 1) & 7;
  const int chunk0 = ((lane >> 4) ^ sw) * 16;
  const int a_off = (wm * WTM + frow) * 128 + chunk0;
  const int b_off = A_BYTES + (wn * 48 + frow) * 128 + chunk0;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Gated read-modify-write epilogue: every residual element belongs to exactly one lane of one workgroup, so its old value,
  // the gate and the bias are fetched NOW, ahead of the first DMA: vmcnt retires in order, so the first counted wait of the K
  // loop also covers them and their latency disappears under the main loop (a load / wait / store chain per fragment after it
  // cost six dependent round trips).  Rows >= M read row M - 1 and are not stored.
  const int ncol = n0 + wn * 48 + (lane >> 4) * 4;
  constexpr int PF = PREFETCH ? FM : 1, PG = PREFETCH && BM == 128 ? FM : 1;   // BM = 256: rows_per_sample % 256 == 0 (launcher), one gate row per tile
  float4 rres[PF][FN], g4[PG][FN], b4[FN];
  if constexpr (PREFETCH) {
#pragma unroll
    for (int j = 0; j < FN; ++j) b4[j] = *(const float4*)(g.bias + ncol + j * 16);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int mc = min(m0 + wm * WTM + i * 16 + frow, g.M - 1);
      const float* gate_row = g.gate + (size_t)(mc / g.rows_per_sample) * g.gate_stride;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        rres[i][j] = *(const float4*)((const float*)g.out + (size_t)mc * g.N + ncol + j * 16);
        if (PG == FM || i == 0) g4[PG == FM ? i : 0][j] = *(const float4*)(gate_row + ncol + j * 16);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  int islot = 0, cslot = 0;
  if constexpr (!PRODUCER) {
#pragma unroll
    for (int t = 0; t < LOOK; ++t)
      if (nk > t) { stage(t, islot); islot = next_slot(islot); }
  }
  for (int kt = 0; kt < nk; ++kt) {
    // K tile kt has landed for everybody; the barrier also retires every read of K tile kt - 1, whose stage takes K tile kt + LOOK
    if constexpr (!PRODUCER) wait_tile(kt);
    __builtin_amdgcn_s_barrier();
    if constexpr (!PRODUCER) {
      if (kt + LOOK < nk) { stage(kt + LOOK, islot); islot = next_slot(islot); }
    }
    const char* sbuf = smem + cslot * STAGE;
    cslot = next_slot(cslot);
    if constexpr (BM == 128) {
      u32x4 af[2][FM], bf[2][FN];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < FM; ++i) af[ks][i] = *(const u32x4*)(sbuf + ((a_off + i * 2048) ^ (ks << 6)));
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[ks][j] = *(const u32x4*)(sbuf + ((b_off + j * 2048) ^ (ks << 6)));
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = mfma16<DT>(bf[ks][j], af[ks][i], acc[i][j]);
    } else {   // 256-row tile: one k-half of fragments at a time (28 instead of 56 fragment registers)
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
  }

#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * WTM + i * 16 + frow;
    if (m >= g.M) continue;
    const float* gate_row = nullptr;
    if constexpr (EPI == EPI_GATE_RES_F32 && !PREFETCH) gate_row = g.gate + (size_t)(m / g.rows_per_sample) * g.gate_stride;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      if constexpr (PREFETCH) {
        float4 r = rres[i][j];
        const float4 gg = g4[PG == FM ? i : 0][j];
        r.x += gg.x * (acc[i][j][0] + b4[j].x);
        r.y += gg.y * (acc[i][j][1] + b4[j].y);
        r.z += gg.z * (acc[i][j][2] + b4[j].z);
        r.w += gg.w * (acc[i][j][3] + b4[j].w);
        *(float4*)((float*)g.out + (size_t)m * g.N + ncol + j * 16) = r;
      } else {
        epilogue_store<EPI, DT>(g, acc[i][j], m, ncol + j * 16, gate_row);
      }
    }
  }
}

template <int DT, int PRODUCER, int BM = 128>
int launch_n144(const GemmArgs& a, int epi, hipStream_t st) {
  constexpr int LDS = (BM == 128 ? 4 : 3) * (BM + 144) * 128;
  if (a.N % 144 != 0 || a.K % 64 != 0 || a.k_chunk != 0)
    return fail(LATTE_ERR_INVALID, "gemm (144-wide tile): need N % 144 == 0, K % 64 == 0, no K split");
  dim3 grid(((a.M + BM - 1) / BM) * (a.N / 144)), block(PRODUCER ? 1024 : 768);
#define LATTE_GEMM_CASE(E)                                                                           \
  case E: {                                                                                          \
    auto kern = gemm_n144_kernel<E, DT, PRODUCER, BM>;                                               \
    static std::atomic<uint64_t> attr_done{0};                                                       \
    if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;                 \
    hipLaunchKernelGGL(kern, grid, block, LDS, st, a);                                               \
    break;                                                                                           \
  }
  switch (epi) {
    LATTE_GEMM_CASE(EPI_BIAS_H16)
    LATTE_GEMM_CASE(EPI_BIAS_GELU_H16)
    LATTE_GEMM_CASE(EPI_GATE_RES_F32)
    LATTE_GEMM_CASE(EPI_BIAS_F32)
    default:
      return fail(LATTE_ERR_INVALID, "gemm (144-wide tile): unknown epilogue");
  }
#undef LATTE_GEMM_CASE
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

template <int BM, int BN, int WGM, int WGN, int DT>
int launch_cfg(const GemmArgs& a, int epi, hipStream_t st) {
  constexpr int LDS = 2 * (BM + BN) * 128;
  const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
  const int splits = a.k_chunk > 0 ? (a.K + a.k_chunk - 1) / a.k_chunk : 1;
  if (a.k_chunk % 64) return fail(LATTE_ERR_INVALID, "gemm: k_chunk must be a multiple of 64");
  dim3 grid(tiles, splits), block(WGM * WGN * 64);
#define LATTE_GEMM_CASE(E)                                                                           \
  case E: {                                                                                          \
    auto kern = gemm_kernel<BM, BN, WGM, WGN, E, DT>;                                                \
    static std::atomic<uint64_t> attr_done{0};                                                       \
    if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;                 \
    hipLaunchKernelGGL(kern, grid, block, LDS, st, a);                                               \
    break;                                                                                           \
  }
  switch (epi) {
    LATTE_GEMM_CASE(EPI_BIAS_H16)
    LATTE_GEMM_CASE(EPI_BIAS_GELU_H16)
    LATTE_GEMM_CASE(EPI_GATE_RES_F32)
    LATTE_GEMM_CASE(EPI_BIAS_F32)
    LATTE_GEMM_CASE(EPI_BIAS_RES_H16)
    default:
      return fail(LATTE_ERR_INVALID, "gemm: unknown epilogue");
  }
#undef LATTE_GEMM_CASE
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}


// ------------------------------------------------------------------------------------------------
// Ping-pong kernel: 256 x BN tile, 8 waves = 2 groups (output rows 0-127 / 128-255) x 4 waves along
// N (wave tile 128 x BN/4), BK = 64, two LDS stages.  Per K tile every wave runs two
// barrier-delimited segments
//     L(t): 8*2 A + FN*2 B fragment reads (ds_read_b128)          ~550 cycles
//     C(t): all 16*FN MFMAs of the K tile                          ~16*FN*16 cycles
// and group 1 executes one extra barrier up front, so it is always one segment behind group 0: each
// SIMD hosts one wave of each group, and while one is in C the other is in L (guide: 8-phase /
// ping-pong, T3+T4+T5).  Intervals: 2t = {G0 L(t), G1 C(t-1)}, 2t+1 = {G0 C(t), G1 L(t)}.
//
// DMA (global_load_lds) placement: group 0's waves load the B tile and A rows 0-127 of K tile t+1 at
// the end of L(t) [interval 2t] -- what group 0 itself reads first, in interval 2t+2; group 1's waves
// load only A rows 128-255 (read by group 1 alone, in 2t+3) at the end of their L(t) [2t+1].  Every
// wave drains its own DMA (vmcnt(0)) at the end of its C(t), i.e. one barrier before the first reader.
// WAR: readers retire their ds_reads (lgkmcnt(0)) before the barrier ending their L segment; the DMA
// that overwrites that stage is issued after it.
// Measured alternatives (DESIGN.md "GEMM experiments"): four segments per K tile (all DMA in one
// 16-read segment) -15 %; moving half of the B DMA into group 1's compute segment with counted vmcnt
// (prefetch distance 2) -5 %; ablations on 8192x4096x4096: no DMA 208 us, no ds_read 241 us, neither
// 200 us vs 244-251 us for the full kernel -> the schedule runs at ~80 % of its load-free ceiling.
template <int BN, int EPI, int DT>
__global__ void __launch_bounds__(512) gemm_pp_kernel(GemmArgs g) {
  constexpr int BM = 256, NW = 8;
  constexpr int WTN = BN / 4, FN = WTN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;   // prologue: all 8 waves cooperate
  constexpr int AH_INSTR = 128 / 8 / 4;                          // a group's 4 waves load its 128 A rows: 4 each
  constexpr int BG_INSTR = BN / 8 / 4;                           // B row-groups per wave of ONE group: 4 / 6 / 8

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;

  int tm, tn;
  tile_coords((g.M + BM - 1) / BM, g.N / BN, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int K = g.K;
  // split K (see gemm_kernel): blockIdx.y owns [y k_chunk, (y + 1) k_chunk) of the contraction, partial product to its own slab
  int k_begin = 0, k_len = K;
  if (g.k_chunk > 0) {
    k_begin = blockIdx.y * g.k_chunk;
    k_len = min(g.k_chunk, K - k_begin);
    g.out = (void*)((float*)g.out + (size_t)blockIdx.y * (size_t)g.split_stride);
  }

  // DMA addressing: ONE 32-bit per-lane byte offset (row-in-group, swizzled chunk); tile base, row-group
  // stride and K offset stay in scalar registers (SGPR base + VGPR offset form: no 64-bit pointer VGPRs).
  const int lrow = lane >> 3, cpos = lane & 7;
  // lane part is identical for the 8-wave prologue mapping (row-group = wave + 8 j) and the 4-wave
  // main-loop mapping (row-group = wn + 4 j): the swizzle only sees ((row >> 1) & 7) and both
  // 8*wave and 8*wn contribute (wave or wn)*4 -> add that parity term per mapping below.
  const unsigned lane_row_off = (unsigned)lrow * (unsigned)K * 2u;
  auto lane_off = [&](int group_row0) -> unsigned {   // group_row0 = first row of the wave's row-group (multiple of 8)
    const int row = group_row0 + lrow;
    return lane_row_off + (unsigned)((cpos ^ ((row >> 1) & 7)) * 16);
  };
  const unsigned off_pro = lane_off(wave * 8);          // prologue: rows wave*8 + 64 j  (64 j does not move the swizzle)
  const unsigned off_main = lane_off(wn * 8);           // main loop: rows wn*8 + 32 j
  const char* a_tile = (const char*)(g.A + (size_t)m0 * K + k_begin);
  const char* b_tile = (const char*)(g.W + (size_t)n0 * K + k_begin);
  const size_t row_bytes = (size_t)K * 2;

  auto dma_a_half = [&](int kt) {   // own 128 A rows of K tile kt: row-groups wn + 4 j
    char* sA = smem + (kt & 1) * STAGE + grp * 128 * 128 + wn * 1024;
    const char* src = a_tile + ((size_t)(grp * 128 + wn * 8)) * row_bytes + (size_t)kt * 128;
#pragma unroll
    for (int j = 0; j < AH_INSTR; ++j)
      glds16((const half_t*)(src + (size_t)(32 * j) * row_bytes + off_main), sA + j * 4 * 1024);
  };
  auto dma_b_all = [&](int kt) {   // whole B tile by ONE group's 4 waves: row-groups wn + 4 j
    char* sB = smem + (kt & 1) * STAGE + A_BYTES + wn * 1024;
    const char* src = b_tile + ((size_t)(wn * 8)) * row_bytes + (size_t)kt * 128;
#pragma unroll
    for (int j = 0; j < BG_INSTR; ++j) glds16((const half_t*)(src + (size_t)(32 * j) * row_bytes + off_main), sB + j * 4 * 1024);
  };

  const int frow = lane & 15;
  const int sw = (lane >> 1) & 7;
  const int chunk0 = ((lane >> 4) ^ sw) * 16;
  const int a_off = (grp * 128 + frow) * 128 + chunk0;
  const int b_off = A_BYTES + (wn * WTN + frow) * 128 + chunk0;

  f32x4 acc[8][FN];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = k_len / 64;
  {  // prologue: K tile 0 by all waves; G1 also places its share of B(1) (its "C(-1)" slot)
    char* sA = smem + wave * 1024;
    char* sB = sA + A_BYTES;
    const char* srcA = a_tile + (size_t)(wave * 8) * row_bytes;
    const char* srcB = b_tile + (size_t)(wave * 8) * row_bytes;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) glds16((const half_t*)(srcA + (size_t)(64 * j) * row_bytes + off_pro), sA + j * NW * 1024);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) glds16((const half_t*)(srcB + (size_t)(64 * j) * row_bytes + off_pro), sB + j * NW * 1024);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger the two groups by one segment

  for (int kt = 0; kt < nk; ++kt) {
    const char* sbuf = smem + (kt & 1) * STAGE;
    u32x4 bf[2][FN], af[2][8];
    // ---- L(kt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[ks][j] = *(const u32x4*)(sbuf + ((b_off + j * 2048) ^ (ks << 6)));
#pragma unroll
      for (int i = 0; i < 8; ++i) af[ks][i] = *(const u32x4*)(sbuf + ((a_off + i * 2048) ^ (ks << 6)));
    }
    if (kt + 1 < nk) {
      dma_a_half(kt + 1);
      if (grp == 0) dma_b_all(kt + 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // stage kt fully consumed by this wave
    __builtin_amdgcn_s_barrier();
    // ---- C(kt)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma16<DT>(bf[ks][j], af[ks][i], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // own DMA of K tile kt+1 landed
    __builtin_amdgcn_s_barrier();
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();  // balance group 1's extra barrier

  const int ncol = n0 + wn * WTN + (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + grp * 128 + i * 16 + frow;
    if (m >= g.M) continue;
    const float* gate_row = nullptr;
    if constexpr (EPI == EPI_GATE_RES_F32) gate_row = g.gate + (size_t)(m / g.rows_per_sample) * g.gate_stride;
#pragma unroll
    for (int j = 0; j < FN; ++j) epilogue_store<EPI, DT>(g, acc[i][j], m, ncol + j * 16, gate_row);
  }
}

// ------------------------------------------------------------------------------------------------
// Persistent ping-pong kernel: the same 256 x BN segment schedule as gemm_pp_kernel, but ONE workgroup
// per CU walks a sequence of output tiles and the K loop is continued across tile boundaries:
//   * the DMA of the next tile's K tile 0 is issued in the last L segment of the current tile, so only
//     the first tile of a workgroup pays the HBM/L2 latency of an empty pipeline;
//   * the epilogue's stores are asynchronous and drain under the next tile's main loop instead of all
//     256 CUs bursting to HBM at the end of a lock-step round;
//   * at a tile boundary group 1 (one segment behind) runs its epilogue BEFORE the barrier that ends its
//     last compute segment and group 0 AFTER it, so the two epilogues overlap each other and the matrix
//     pipe idles for about one epilogue, not two.
// Register budget (2 waves / SIMD -> 256 VGPRs): 128 accumulators + 96 fragment registers leave ~30 for
// everything else, so the DMA uses buffer_load ... lds (ONE 32-bit per-lane offset; tile / row-group /
// K offsets are scalar soffsets against a descriptor of the whole matrix) and the epilogue re-derives
// its lane-dependent indices from an opaque copy of the lane id (nothing of it stays live in the K loop).
// Tile sequence of a workgroup: XCD x (= blockIdx & 7) owns a contiguous chunk of the grouped tile
// order (tile_coords' order); slot s (= blockIdx >> 3) takes positions s, s + G/8, ... of that chunk, so
// the 32 workgroups of one XCD work on a compact patch of tiles at any time (shared A / W panels in L2).
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void bload_lds16(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// RMW_AHEAD / RMW_NT: the gated fp32 read-modify-write epilogue keeps RMW_AHEAD residual fragments (1 KB per wave each)
// in flight ahead of the stores; RMW_NT loads them with the non-temporal policy (they are read exactly once).
// TAG only separates the instantiations of the two gated-residual call sites (0 = attention out-projection, 1 = fc2), which
// share every other template argument, so that a kernel trace lists them as two kernels.
template <int BN, int EPI, int DT, int RMW_AHEAD = 2, int RMW_NT = 0, int TAG = 0>
__global__ void __launch_bounds__(512) gemm_pps_kernel(GemmArgs g) {
  constexpr int BM = 256, NW = 8;
  constexpr int WTN = BN / 4, FN = WTN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
  constexpr int AH_INSTR = 128 / 8 / 4;
  constexpr int BG_INSTR = BN / 8 / 4;
  constexpr int GROUP_M = 8;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  const int K = g.K;
  const unsigned row_bytes = (unsigned)K * 2u;

  // ---- this workgroup's tile sequence
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN, nwg = tiles_m * tiles_n;   // N % WTN == 0 (launcher)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int chunk0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int cnt = q + (xcd < r ? 1 : 0);
  if (slot >= cnt) return;
  if (g.stagger > 1) {
    // De-synchronise the CUs: all workgroups otherwise reach their epilogue together and the store bursts (32 MB per
    // round) cost their full HBM time because an in-order vmcnt cannot confirm later DMA past them.  Cohort c of the
    // workgroups of an XCD starts c / stagger of a tile time late (only used when a workgroup walks many tiles).
    const int cohort = slot % g.stagger;
    const long long wait = (long long)cohort * (K / 64) * 2200 / g.stagger;   // ~2200 cycles per K tile (two segments)
    const long long t0 = __builtin_readcyclecounter();
    while ((long long)__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
  const int group_m = g.group_m > 0 ? g.group_m : GROUP_M;   // tile rows walked together (L2 reuse of the W panels)
  auto decode = [&](int wg, int& tm, int& tn) {
    const int per_group = group_m * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * group_m;
    const int gsz = min(tiles_m - first_m, group_m);
    const int in_group = wg - group * per_group;
    tm = first_m + in_group % gsz;
    tn = in_group / gsz;
  };

  // descriptors of the whole (row-padded) activation matrix and of the weight matrix (< 4 GiB: launcher)
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (unsigned)tiles_m * BM * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (unsigned)g.N * row_bytes, 0x00020000);

  // per-lane DMA offset: row (lane >> 3) of an 8-row group, 16-byte chunk (lane & 7) ^ swizzle(row)
  const int lrow = lane >> 3, cpos = lane & 7;
  auto lane_off = [&](int group_row0) -> unsigned {
    const int row = group_row0 + lrow;
    return (unsigned)lrow * row_bytes + (unsigned)((cpos ^ ((row >> 1) & 7)) * 16);
  };
  const unsigned off_main = lane_off(wn * 8);   // rows wn*8 + 32 j (32 j does not move the swizzle)

  unsigned step32 = 32u * row_bytes;   // 32 rows further: one instruction step of a 4-wave DMA
  asm volatile("" : "+s"(step32));     // opaque: keeps the per-instruction offsets as one s_add each (not s_mul)
  // own 128 A rows of K tile kt of tile-row tm_: row-groups wn + 4 j
  auto dma_a_half = [&](int tm_, int kt, int stg) {
    char* sA = smem + stg * STAGE + grp * 128 * 128 + wn * 1024;
    const unsigned so = (unsigned)(tm_ * BM + grp * 128 + wn * 8) * row_bytes + (unsigned)kt * 128u;
#pragma unroll
    for (int j = 0; j < AH_INSTR; ++j) bload_lds16(rsA, sA + j * 4 * 1024, off_main, so + (unsigned)j * step32);
  };
  // whole B tile by ONE group's 4 waves
  auto dma_b_all = [&](int tn_, int kt, int stg) {
    char* sB = smem + stg * STAGE + A_BYTES + wn * 1024;
    const unsigned so = (unsigned)(tn_ * BN + wn * 8) * row_bytes + (unsigned)kt * 128u;
#pragma unroll
    for (int j = 0; j < BG_INSTR; ++j) bload_lds16(rsB, sB + j * 4 * 1024, off_main, so + (unsigned)j * step32);
  };

  const int sw = (lane >> 1) & 7;
  const int chunkb = ((lane >> 4) ^ sw) * 16;
  const int a_off = (grp * 128 + (lane & 15)) * 128 + chunkb;
  const int b_off = A_BYTES + (wn * WTN + (lane & 15)) * 128 + chunkb;

  f32x4 acc[8][FN];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int pos = slot, tm, tn;
  decode(chunk0 + pos, tm, tn);
  const int nk = K / 64;   // >= 2 (launcher)
  {  // pipeline fill: K tile 0 of the first tile by all 8 waves (row-groups wave + 8 j)
    const unsigned off_pro = lane_off(wave * 8);
    char* sA = smem + wave * 1024;
    char* sB = sA + A_BYTES;
    const unsigned soA = (unsigned)(tm * BM + wave * 8) * row_bytes;
    const unsigned soB = (unsigned)(tn * BN + wave * 8) * row_bytes;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) bload_lds16(rsA, sA + j * NW * 1024, off_pro, soA + (unsigned)(64 * j) * row_bytes);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) bload_lds16(rsB, sB + j * NW * 1024, off_pro, soB + (unsigned)(64 * j) * row_bytes);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // K tile 1 (stage 1 is untouched so far): each group its own A rows, group 0 the B tile
  dma_a_half(tm, 1, 1);
  if (grp == 0) dma_b_all(tn, 1, 1);
  if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger the two groups by one segment

  // Epilogue of tile (tm_, tn_) for this wave's 128 x WTN sub-tile, then clear the accumulators.
  // Half-precision outputs of the 256-wide tile go through a wave-private 4 KB LDS patch (the 32 KB the two
  // stages leave free) so that every global store instruction writes 8 complete 128-byte rows instead of
  // 16 rows x 32 bytes: the direct form is store-ISSUE bound (measured: 25-30 % of the fc1 / qkv launch).
  constexpr bool LDS_EPI = (EPI == EPI_BIAS_H16 || EPI == EPI_BIAS_GELU_H16) && BN == 256;
  auto epilogue = [&](int tm_, int tn_) {
    int le = lane;
    asm volatile("" : "+v"(le));   // opaque: keeps every lane-derived epilogue index out of the K loop's live set
    const int fr = le & 15;
    const int ncol0 = tn_ * BN + wn * WTN;          // first column of this wave's sub-tile
    if (ncol0 >= g.N) {                             // N edge: this wave's columns do not exist (wave-uniform)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      return;
    }
    if constexpr (LDS_EPI) {
      // patch image: [32 rows][128 B]; 16-byte piece p of row r lives at piece p ^ ((r >> 1) & 7)
      char* patch = smem + 2 * STAGE + wave * 4096;
      const int gq = le >> 4;
      float4 b4[FN];
      const int mrow0 = tm_ * BM + grp * 128;
#pragma unroll
      for (int j = 0; j < FN; ++j) b4[j] = *(const float4*)(g.bias + ncol0 + j * 16 + gq * 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int i = 2 * c + ii;
          const int row = ii * 16 + fr;
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            float v0 = acc[i][j][0] + b4[j].x, v1 = acc[i][j][1] + b4[j].y, v2 = acc[i][j][2] + b4[j].z, v3 = acc[i][j][3] + b4[j].w;
            if constexpr (EPI == EPI_BIAS_GELU_H16) {
              v0 = gelu_tanh(v0); v1 = gelu_tanh(v1); v2 = gelu_tanh(v2); v3 = gelu_tanh(v3);
            }
            const u32x2 pk = {pack2<DT>(v0, v1), pack2<DT>(v2, v3)};
            const int piece = j * 2 + (gq >> 1);
            *(u32x2*)(patch + row * 128 + ((piece ^ ((row >> 1) & 7)) << 4) + ((gq & 1) << 3)) = pk;
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
          const int row = kq * 8 + (le >> 3), piece = le & 7;
          const u32x4 v = *(const u32x4*)(patch + row * 128 + ((piece ^ ((row >> 1) & 7)) << 4));
          const int m = mrow0 + c * 32 + row;
          if (m < g.M) *(u32x4*)((half_t*)g.out + (size_t)m * g.N + ncol0 + piece * 8) = v;
        }
      }
    } else {
      const int ncol = ncol0 + (le >> 4) * 4;
      const int mbase = tm_ * BM + grp * 128 + fr;
      if constexpr (EPI == EPI_GATE_RES_F32 && FN <= 3) {
        // res[m, n] += gate[sample, n] * (acc + bias), fast path for tiles inside one sample (rows_per_sample % 256 == 0:
        // every Latte config at F*T >= 256).  vmcnt retires in issue order and counts stores, so in the plain
        // load / wait / store sequence every load also waits for the previous store: a chain of 24 load + store round
        // trips per tile.  Here the residual loads run TWO fragments ahead of the stores, so a wait only ever covers a
        // load issued before the last stores.  (Issuing ALL 24 loads first was measured 24 % slower in the model --
        // fc2 337 -> 417 us: 6 MB of loaded lines per XCD overflow the 4 MB L2 before the stores arrive.)
        if ((g.rows_per_sample % BM) == 0) {
          float* const outp = (float*)g.out;
          const float* gr = g.gate + (size_t)((tm_ * BM) / g.rows_per_sample) * g.gate_stride + ncol;
          float4 b4[FN], g1[FN];
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            b4[j] = *(const float4*)(g.bias + ncol + j * 16);
            g1[j] = *(const float4*)(gr + j * 16);
          }
          // fragments in (i, j) order, residual loads running RMW_AHEAD fragments ahead of the stores
          constexpr int NF = 8 * FN;
          auto frag_ptr = [&](int f) -> float* {
            const int mc = min(mbase + (f / FN) * 16, g.M - 1);      // clamped row: the load is unconditional
            return outp + (size_t)mc * g.N + ncol + (f % FN) * 16;
          };
          auto load_res = [&](int f) -> float4 {
            if constexpr (RMW_NT) {
              const f32x4 v = __builtin_nontemporal_load((const f32x4*)frag_ptr(f));
              return make_float4(v[0], v[1], v[2], v[3]);
            } else {
              return *(const float4*)frag_ptr(f);
            }
          };
          float4 qa[RMW_AHEAD];
#pragma unroll
          for (int a = 0; a < RMW_AHEAD; ++a) qa[a] = load_res(a);
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            float4 r = qa[f % RMW_AHEAD];
            if (f + RMW_AHEAD < NF) qa[f % RMW_AHEAD] = load_res(f + RMW_AHEAD);
            asm volatile("" ::: "memory");                           // keep the prefetch ahead of this fragment's store
            const int i = f / FN, j = f % FN;
            r.x += g1[j].x * (acc[i][j][0] + b4[j].x);
            r.y += g1[j].y * (acc[i][j][1] + b4[j].y);
            r.z += g1[j].z * (acc[i][j][2] + b4[j].z);
            r.w += g1[j].w * (acc[i][j][3] + b4[j].w);
            if (mbase + i * 16 < g.M) *(float4*)(outp + (size_t)(mbase + i * 16) * g.N + ncol + j * 16) = r;
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
          return;
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = mbase + i * 16;
        if (m < g.M) {
          const float* gate_row = nullptr;
          if constexpr (EPI == EPI_GATE_RES_F32) gate_row = g.gate + (size_t)(m / g.rows_per_sample) * g.gate_stride;
#pragma unroll
          for (int j = 0; j < FN; ++j) epilogue_store<EPI, DT>(g, acc[i][j], m, ncol + j * 16, gate_row);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  // Schedule per K tile u (global counter `it`), stage(u) = u & 1:
  //   L(u)  : fragment reads of stage(u), lgkmcnt(0), barrier
  //   C(u)  : MFMAs; wait for DMA(u+1)
  //   then  : issue DMA(u+2) into stage(u) -- group 1 BEFORE the barrier ending its C(u) (it only writes its
  //           own A rows, which nobody else reads), group 0 AFTER the barrier ending its C(u) (by then group 1
  //           has finished L(u), the last reader of stage(u)); at a tile boundary the epilogue follows.
  // vmcnt retires in issue order and counts stores, so the DMA is issued BEFORE the epilogue: the epilogue's own
  // first use of a loaded value (bias / residual) then implies that DMA has landed, and the C segment that follows
  // a boundary needs no wait at all while the stores keep draining under the next tile's loop.
  // (Tiles with rows >= M, and waves whose columns lie beyond N, take the plain vmcnt(0) path.)
  int it = 0;
  bool counted = false;   // the DMA needed by this iteration was issued before an epilogue that loaded and used data
  // main-loop ablations (measurement only)
  constexpr bool NO_DMA = EPI == EPI_ABLATE_NODMA, NO_LDSR = EPI == EPI_ABLATE_NOLDSREAD, NO_MFMA = EPI == EPI_ABLATE_NOMFMA;
  u32x4 hold_b[NO_LDSR ? 2 : 1][NO_LDSR ? FN : 1], hold_a[NO_LDSR ? 2 : 1][NO_LDSR ? 8 : 1];
  constexpr bool DO_A = EPI != EPI_ABLATE_DMA_B, DO_B = EPI != EPI_ABLATE_DMA_A;
  constexpr bool TRACE = EPI == EPI_ABLATE_TRACE;
  long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0, tstart = 0;
  if constexpr (TRACE) tstart = tprev = (long long)__builtin_readcyclecounter();
#define LATTE_TS(IDX)                                                \
  if constexpr (TRACE) {                                             \
    const long long now_ = (long long)__builtin_readcyclecounter();  \
    tacc[IDX] += now_ - tprev;                                       \
    tprev = now_;                                                    \
  }
  for (;;) {
    const int npos = pos + per;
    const bool has_next = npos < cnt;
    int ntm = tm, ntn = tn;
    if (has_next) decode(chunk0 + npos, ntm, ntn);
    const bool full_rows = (tm + 1) * BM <= g.M;

    for (int kt = 0; kt < nk; ++kt, ++it) {
      // (scalar bookkeeping of this iteration first: it then runs under the fragment reads, not between the last MFMA
      //  and the barrier that releases the other group)
      const bool last = kt + 1 == nk;
      const bool in_tile = kt + 2 < nk;
      const bool do_dma = !NO_DMA && (in_tile || has_next);
      const int stm = in_tile ? tm : ntm, stn = in_tile ? tn : ntn;
      const int skt = EPI == EPI_ABLATE_HOTSRC ? 0 : in_tile ? kt + 2 : kt + 2 - nk;
      const char* sbuf = smem + (it & 1) * STAGE;
      u32x4 bf[2][FN], af[2][8];
      if (!NO_LDSR || it == 0) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int j = 0; j < FN; ++j) bf[ks][j] = *(const u32x4*)(sbuf + ((b_off + j * 2048) ^ (ks << 6)));
#pragma unroll
          for (int i = 0; i < 8; ++i) af[ks][i] = *(const u32x4*)(sbuf + ((a_off + i * 2048) ^ (ks << 6)));
        }
      }

      if constexpr (NO_LDSR) {   // ablation: keep the first K tile's fragments
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int j = 0; j < FN; ++j) { if (it == 0) hold_b[ks][j] = bf[ks][j]; else bf[ks][j] = hold_b[ks][j]; }
#pragma unroll
          for (int i = 0; i < 8; ++i) { if (it == 0) hold_a[ks][i] = af[ks][i]; else af[ks][i] = hold_a[ks][i]; }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      LATTE_TS(0)
      __builtin_amdgcn_s_barrier();
      LATTE_TS(1)
      __builtin_amdgcn_s_setprio(1);
      if constexpr (NO_MFMA) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int j = 0; j < FN; ++j) asm volatile("" ::"v"(bf[ks][j]));
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(af[ks][i]));
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = mfma16<DT>(bf[ks][j], af[ks][i], acc[i][j]);
      }
      __builtin_amdgcn_s_setprio(0);
      LATTE_TS(2)
      // DMA(u+1) must have landed.  After a tile boundary it provably has: the epilogue began with global loads
      // (bias / residual) that were issued AFTER DMA(u) and DMA(u+1) and were consumed before its first store, and
      // vmcnt retires in issue order -- so no wait here, and the epilogue's stores keep draining under this tile.
      if (!counted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      counted = false;
      LATTE_TS(3)
      // K tile u+2 (same tile, or the next tile's K tile kt + 2 - nk) into the stage just consumed.
      // Group 1's own A rows may be overwritten as soon as its L(u) is done, but the issue (4 buffer loads per wave,
      // which stall when the address unit's queue is full) must not sit between its last MFMA and the barrier that
      // releases group 0's compute segment.  So it follows the barrier -- except at a tile boundary, where it has to
      // precede the epilogue (in-order vmcnt argument above).
      if (grp == 1 && last) {
        if (do_dma && DO_A) dma_a_half(stm, skt, it & 1);
        asm volatile("" ::: "memory");   // the epilogue's loads must stay BEHIND the DMA issue
        epilogue(tm, tn);
      }
      __builtin_amdgcn_s_barrier();
      LATTE_TS(4)
      if (grp == 1 && !last) {
        if (do_dma && DO_A) dma_a_half(stm, skt, it & 1);
      }
      if (grp == 0) {
        if (do_dma) {
          if (DO_A) dma_a_half(stm, skt, it & 1);
          if (DO_B) dma_b_all(stn, skt, it & 1);
        }
        asm volatile("" ::: "memory");
        if (last) epilogue(tm, tn);
      }
      if (last) counted = do_dma && full_rows && (tn * BN + wn * WTN < g.N);
      LATTE_TS(5)
    }
    if (!has_next) break;
    pos = npos; tm = ntm; tn = ntn;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();  // balance group 1's extra barrier
#undef LATTE_TS
  if constexpr (TRACE) {   // phase times of workgroup 0 (tools/gpu_first_light.py gemm_trace)
    if (blockIdx.x == 0 && lane == 0) {
      long long* o = (long long*)g.out + wave * 8;
#pragma unroll
      for (int i = 0; i < 6; ++i) o[i] = tacc[i];
      o[6] = (long long)__builtin_readcyclecounter() - tstart;
      o[7] = it;
    }
  }
}


template <int BN, int DT>
int launch_pp(const GemmArgs& a, int epi, hipStream_t st);

template <int BN, int DT>
int launch_pps(const GemmArgs& a, int epi, hipStream_t st) {
  constexpr int LDS = 2 * (256 + BN) * 128 + (BN == 256 ? 8 * 4096 : 0);   // + wave-private epilogue patches
  const int tiles = ((a.M + 255) / 256) * ((a.N + BN - 1) / BN);
  if ((uint64_t)((a.M + 255) / 256 * 256) * a.K * 2 >= (1ull << 32) || (uint64_t)a.N * a.K * 2 >= (1ull << 32) || a.K < 128)
  {  // 32-bit buffer offsets would overflow / a single K tile: non-persistent kernel (whole tile columns only)
    if (a.N % BN) return fail(LATTE_ERR_INVALID, "gemm: shape needs the persistent kernel but exceeds its 4 GiB / K >= 128 limits");
    return launch_pp<BN, DT>(a, epi, st);
  }
  const int nblk = tiles >= 256 ? 256 : (tiles + 7) / 8 * 8;   // one workgroup per CU, multiple of the 8 XCDs
  dim3 grid(nblk), block(512);
#define LATTE_GEMM_CASE(E)                                                                           \
  case E: {                                                                                          \
    auto kern = gemm_pps_kernel<BN, E, DT>;                                                          \
    static std::atomic<uint64_t> attr_done{0};                                                       \
    if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;                 \
    hipLaunchKernelGGL(kern, grid, block, LDS, st, a);                                               \
    break;                                                                                           \
  }
#ifdef LATTE_GEMM_ABLATE
  // measurement build: look-ahead depth / cache policy of the read-modify-write epilogue (GemmArgs::rmw_mode = depth + 16 * nt)
  if constexpr (BN == 192 && DT == LATTE_DTYPE_BF16) {
    if (epi == EPI_GATE_RES_F32 && a.rmw_mode) {
#define LATTE_RMW_CASE(AH, NT)                                                                        \
  case AH + 16 * NT: {                                                                                \
    auto kern = gemm_pps_kernel<BN, EPI_GATE_RES_F32, DT, AH, NT>;                                    \
    static std::atomic<uint64_t> attr_done{0};                                                        \
    if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;                  \
    hipLaunchKernelGGL(kern, grid, block, LDS, st, a);                                                \
    LATTE_HIP(hipGetLastError());                                                                     \
    return LATTE_OK;                                                                                  \
  }
      switch (a.rmw_mode) {
        LATTE_RMW_CASE(3, 0) LATTE_RMW_CASE(4, 0) LATTE_RMW_CASE(6, 0) LATTE_RMW_CASE(8, 0) LATTE_RMW_CASE(12, 0)
        LATTE_RMW_CASE(2, 1) LATTE_RMW_CASE(4, 1) LATTE_RMW_CASE(6, 1) LATTE_RMW_CASE(8, 1)
        default: break;
      }
#undef LATTE_RMW_CASE
    }
  }
#endif
  if (epi == EPI_GATE_RES_F32 && a.tag == 1) {   // fc2: its own kernel symbol
    auto kern = gemm_pps_kernel<BN, EPI_GATE_RES_F32, DT, 2, 0, 1>;
    static std::atomic<uint64_t> attr_done{0};
    if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;
    hipLaunchKernelGGL(kern, grid, block, LDS, st, a);
    LATTE_HIP(hipGetLastError());
    return LATTE_OK;
  }
  switch (epi) {
    LATTE_GEMM_CASE(EPI_BIAS_H16)
    LATTE_GEMM_CASE(EPI_BIAS_GELU_H16)
    LATTE_GEMM_CASE(EPI_GATE_RES_F32)
    LATTE_GEMM_CASE(EPI_BIAS_F32)
#ifdef LATTE_GEMM_ABLATE   // measurement build only (LATTE_DEBUG_BUILD=1 python -m latte_amd.build): main-loop ablations
    LATTE_GEMM_CASE(EPI_ABLATE_NOSTORE)
    LATTE_GEMM_CASE(EPI_ABLATE_NODMA)
    LATTE_GEMM_CASE(EPI_ABLATE_NOLDSREAD)
    LATTE_GEMM_CASE(EPI_ABLATE_NOMFMA)
    LATTE_GEMM_CASE(EPI_ABLATE_HOTSRC)
    LATTE_GEMM_CASE(EPI_ABLATE_DMA_A)
    LATTE_GEMM_CASE(EPI_ABLATE_DMA_B)
    LATTE_GEMM_CASE(EPI_ABLATE_TRACE)
#endif
    default:
      return fail(LATTE_ERR_INVALID, "gemm: unknown epilogue");
  }
#undef LATTE_GEMM_CASE
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

template <int BN, int DT>
int launch_pp(const GemmArgs& a, int epi, hipStream_t st) {
  constexpr int LDS = 2 * (256 + BN) * 128;
  const int tiles = ((a.M + 255) / 256) * (a.N / BN);
  const int splits = a.k_chunk > 0 ? (a.K + a.k_chunk - 1) / a.k_chunk : 1;
  if (a.k_chunk % 64) return fail(LATTE_ERR_INVALID, "gemm: k_chunk must be a multiple of 64");
  dim3 grid(tiles, splits), block(512);
#define LATTE_GEMM_CASE(E)                                                                           \
  case E: {                                                                                          \
    auto kern = gemm_pp_kernel<BN, E, DT>;                                                           \
    static std::atomic<uint64_t> attr_done{0};                                                       \
    if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;                 \
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

#ifdef LATTE_GEMM_ABLATE
// ------------------------------------------------------------------------------------------------
// Measurement build only (variant 14): ONE MFMA wave per SIMD.  256 x 256 tile, four waves (2 x 2, wave tile 128 x 128:
// 64 accumulator fragments = the 256 AGPRs, two fragment sets of 16 ds_read_b128 = 128 VGPRs), persistent over the same
// XCD-chunked tile order as the ping-pong kernel, K tiles of 64 in two 64 KB LDS stages.  Per K tile and wave:
//     A: MFMAs of the first half (fragment set 0)   ||  ds_reads of set 1
//     lgkmcnt(0), own DMA of K tile t + 1 landed, ONE barrier (everybody done reading stage t & 1, everybody's t + 1 landed)
//     B: MFMAs of the second half (set 1)           ||  DMA of K tile t + 2 into stage t & 1  ||  ds_reads of set 0 of K tile t + 1
// The (tile, K tile) sequence is flat: the first two K tiles of the next output tile are in flight / landed when a tile's
// epilogue starts, and the epilogue's stores are issued AFTER that DMA, so the first barrier of the next tile waits with a
// counted vmcnt (the stores) instead of draining them.  DESIGN.md section 8: the question is whether a lone wave with a full
// register file covers its own LDS latency, which the 168-register waves of the 12-wave kernel do not.
template <int EPI, int DT>
__global__ void __launch_bounds__(256) gemm_w4_kernel(GemmArgs g) {
  constexpr int BM = 256, BN = 256, NW = 4, FM = 8, FN = 8;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr bool LDS_EPI = EPI == EPI_BIAS_H16 || EPI == EPI_BIAS_GELU_H16;
  constexpr int EPI_STORES = LDS_EPI ? 32 : 0;   // global store instructions of one epilogue (LDS_EPI form)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int K = g.K, nk = K / 64;
  const unsigned row_bytes = (unsigned)K * 2u;

  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = g.N / BN, nwg = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int chunk0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int cnt = q + (xcd < r ? 1 : 0);
  if (slot >= cnt) return;
  const int group_m = g.group_m > 0 ? g.group_m : 8;
  auto decode = [&](int wg, int& tm, int& tn) {
    const int per_group = group_m * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * group_m;
    const int gsz = min(tiles_m - first_m, group_m);
    const int in_group = wg - group * per_group;
    tm = first_m + in_group % gsz;
    tn = in_group / gsz;
  };
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (unsigned)tiles_m * BM * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (unsigned)g.N * row_bytes, 0x00020000);
  const int lrow = lane >> 3, cpos = lane & 7;
  const unsigned off_lane = (unsigned)lrow * row_bytes + (unsigned)((cpos ^ (((wave * 8 + lrow) >> 1) & 7)) * 16);
  unsigned step32 = 32u * row_bytes;
  asm volatile("" : "+s"(step32));
  // K tile kt of output tile (tm_, tn_) into stage stg: row-groups wave + 4 j of A and of W, 16 one-KB pieces per wave
  auto dma = [&](int tm_, int tn_, int kt, int stg) __attribute__((always_inline)) {
    char* sA = smem + stg * STAGE + wave * 1024;
    const unsigned soA = (unsigned)(tm_ * BM + wave * 8) * row_bytes + (unsigned)kt * 128u;
    const unsigned soB = (unsigned)(tn_ * BN + wave * 8) * row_bytes + (unsigned)kt * 128u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bload_lds16(rsA, sA + j * NW * 1024, off_lane, soA + (unsigned)j * step32);
      bload_lds16(rsB, sA + A_BYTES + j * NW * 1024, off_lane, soB + (unsigned)j * step32);
    }
  };
  const int sw = (lane >> 1) & 7;
  const int chunkb = ((lane >> 4) ^ sw) * 16;
  const int a_off = (wm * 128 + (lane & 15)) * 128 + chunkb;
  const int b_off = A_BYTES + (wn * 128 + (lane & 15)) * 128 + chunkb;
  u32x4 fa[2][FM], fb[2][FN];
  auto read_set = [&](int set, int stg, int ks) __attribute__((always_inline)) {
    const char* sbuf = smem + stg * STAGE;
#pragma unroll
    for (int i = 0; i < FM; ++i) fa[set][i] = *(const u32x4*)(sbuf + ((a_off + i * 2048) ^ (ks << 6)));
#pragma unroll
    for (int j = 0; j < FN; ++j) fb[set][j] = *(const u32x4*)(sbuf + ((b_off + j * 2048) ^ (ks << 6)));
  };
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mma_set = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = mfma16<DT>(fb[set][j], fa[set][i], acc[i][j]);
  };

  // the bias of the wave's 128 columns goes global -> LDS by one DMA piece (1 KB: the 128 floats + the next 128, unused) into a
  // wave-private slot, double-buffered by tile parity, issued BEFORE the operand DMA of the tile's last K step: no register
  // holds it through the K loop and no load sits behind the DMA in the in-order vmcnt queue when the epilogue starts
  const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)g.bias, 0, (unsigned)g.N * 4u, 0x00020000);
  char* const bias_lds = smem + 2 * STAGE + NW * 4096 + wave * 2048;
  auto dma_bias = [&](int tn_, int par) __attribute__((always_inline)) {
    bload_lds16(rsBias, bias_lds + par * 1024, (unsigned)lane * 16u, (unsigned)(tn_ * BN + wn * 128) * 4u);
  };
  int bias_par = 0;
  auto epilogue = [&](int tm_, int tn_) __attribute__((always_inline)) {
    int le = lane;
    asm volatile("" : "+v"(le));
    const int fr = le & 15, gq = le >> 4;
    const int ncol0 = tn_ * BN + wn * 128, mrow0 = tm_ * BM + wm * 128;
    if constexpr (LDS_EPI) {
      // wave-private 4 KB patch [32 rows][128 B] behind the two stages: every global store writes 8 complete 128-byte row pieces
      char* patch = smem + 2 * STAGE + wave * 4096;
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        float4 b4[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) b4[jj] = *(const float4*)(bias_lds + bias_par * 1024 + (jh * 64 + jj * 16 + gq * 4) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * c + ii, row = ii * 16 + fr;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const int j = jh * 4 + jj;
              float v0 = acc[i][j][0] + b4[jj].x, v1 = acc[i][j][1] + b4[jj].y, v2 = acc[i][j][2] + b4[jj].z, v3 = acc[i][j][3] + b4[jj].w;
              if constexpr (EPI == EPI_BIAS_GELU_H16) {
                v0 = gelu_tanh(v0); v1 = gelu_tanh(v1); v2 = gelu_tanh(v2); v3 = gelu_tanh(v3);
              }
              const u32x2 pk = {pack2<DT>(v0, v1), pack2<DT>(v2, v3)};
              const int piece = jj * 2 + (gq >> 1);
              *(u32x2*)(patch + row * 128 + ((piece ^ ((row >> 1) & 7)) << 4) + ((gq & 1) << 3)) = pk;
              acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
          }
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) {
            const int row = kq * 8 + (le >> 3), piece = le & 7;
            const u32x4 v = *(const u32x4*)(patch + row * 128 + ((piece ^ ((row >> 1) & 7)) << 4));
            const int m = mrow0 + c * 32 + row;
            if (m < g.M) *(u32x4*)((half_t*)g.out + (size_t)m * g.N + ncol0 + jh * 64 + piece * 8) = v;
          }
        }
      }
    } else {
      const int ncol = ncol0 + gq * 4;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int m = mrow0 + i * 16 + fr;
        const float* gate_row = nullptr;
        if constexpr (EPI == EPI_GATE_RES_F32) gate_row = g.gate + (size_t)(min(m, g.M - 1) / g.rows_per_sample) * g.gate_stride;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if (m < g.M) epilogue_store<EPI, DT>(g, acc[i][j], m, ncol + j * 16, gate_row);
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  };

  // flat walk over (tile, K tile) steps: the K loop below multiplies step (tile, kt); n1 = the step after it (landing / landed),
  // n2 = the one after that (issued in this step's second half); both cross into the workgroup's next tile on their own
  int pos = slot, tm, tn;
  decode(chunk0 + pos, tm, tn);
  // next step; when the walk ends p = -1 and the coordinates stay where they were (a valid K tile: the DMA / reads of the last two
  // steps run unconditionally -- a branch would split the scheduling region the interleave below is written for -- and are unused)
  auto advance = [&](int& p, int& a, int& b, int& k) __attribute__((always_inline)) {
    if (k + 1 < nk) { ++k; return; }
    if (p + per >= cnt) { p = -1; return; }
    k = 0;
    p += per;
    decode(chunk0 + p, a, b);
  };
  int pos1 = pos, tm1 = tm, tn1 = tn, kt1 = 0;
  if constexpr (LDS_EPI) dma_bias(tn, 0);
  dma(tm, tn, 0, 0);
  advance(pos1, tm1, tn1, kt1);                       // step 1 (nk >= 2: same tile, kt 1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  dma(tm1, tn1, kt1, 1);
  read_set(0, 0, 0);
  int pos2 = pos1, tm2 = tm1, tn2 = tn1, kt2 = kt1;
  advance(pos2, tm2, tn2, kt2);                       // step 2
  int stg = 0;
  bool after_stores = false;
  for (;;) {
    int posN = -1, tmN = 0, tnN = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const bool opens_next = kt == nk - 1 && pos1 >= 0;     // n1 is K tile 0 of the workgroup's next tile
      if (LDS_EPI && opens_next) dma_bias(tn1, bias_par ^ 1);
      // ---- A: first half of the K tile (set 0), fragment set 1 arriving under it: one read per two MFMAs, then MFMAs alone
      __builtin_amdgcn_sched_barrier(0);
      read_set(1, stg, 1);
      mma_set(0);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (after_stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EPI_STORES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      after_stores = false;
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- B: second half (set 1), the DMA of step + 2 into the stage just released, set 0 of step + 1
      dma(tm2, tn2, kt2, stg);
      read_set(0, stg ^ 1, 0);
      mma_set(1);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (opens_next) { posN = pos1; tmN = tm1; tnN = tn1; }
      pos1 = pos2; tm1 = tm2; tn1 = tn2; kt1 = kt2;
      if (pos2 >= 0) advance(pos2, tm2, tn2, kt2);
      stg ^= 1;
    }
    epilogue(tm, tn);
    after_stores = EPI_STORES > 0;
    bias_par ^= 1;
    if (posN < 0) break;
    pos = posN; tm = tmN; tn = tnN;
  }
}

template <int DT>
int launch_w4(const GemmArgs& a, int epi, hipStream_t st) {
  constexpr int LDS = 2 * 512 * 128 + 4 * 4096 + 4 * 2048;
  if (a.N % 256 || a.K % 64 || a.K < 128 || a.k_chunk) return fail(LATTE_ERR_INVALID, "gemm w4: needs N % 256 == 0, K % 64 == 0, K >= 128, no split K");
  const int tiles = ((a.M + 255) / 256) * (a.N / 256);
  const int nblk = tiles >= 256 ? 256 : (tiles + 7) / 8 * 8;
  dim3 grid(nblk), block(256);
#define LATTE_GEMM_CASE(E)                                                                           \
  case E: {                                                                                          \
    auto kern = gemm_w4_kernel<E, DT>;                                                               \
    static std::atomic<uint64_t> attr_done{0};                                                       \
    if (int rc_ = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_;                 \
    hipLaunchKernelGGL(kern, grid, block, LDS, st, a);                                               \
    break;                                                                                           \
  }
  switch (epi) {
    LATTE_GEMM_CASE(EPI_BIAS_H16)
    LATTE_GEMM_CASE(EPI_BIAS_GELU_H16)
    LATTE_GEMM_CASE(EPI_GATE_RES_F32)
    LATTE_GEMM_CASE(EPI_BIAS_F32)
    LATTE_GEMM_CASE(EPI_ABLATE_NOSTORE)
    default:
      return fail(LATTE_ERR_INVALID, "gemm w4: unknown epilogue");
  }
#undef LATTE_GEMM_CASE
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}
#endif   // LATTE_GEMM_ABLATE

template <int DT>
int launch_dt(const GemmArgs& a, int epi, int variant, hipStream_t st) {
  switch (variant) {
    case 1: return launch_cfg<128, 128, 2, 2, DT>(a, epi, st);
    case 2: return launch_cfg<256, 128, 4, 2, DT>(a, epi, st);
    case 3: return launch_cfg<256, 256, 2, 4, DT>(a, epi, st);
    case 4: return launch_pp<128, DT>(a, epi, st);
    case 5: return launch_pp<192, DT>(a, epi, st);
    case 6: return launch_pp<256, DT>(a, epi, st);
    case 7: return launch_pps<128, DT>(a, epi, st);
    case 8: return launch_pps<192, DT>(a, epi, st);
    case 9: return launch_pps<256, DT>(a, epi, st);
    case 12: return launch_n144<DT, 0>(a, epi, st);
    case 13: return launch_n144<DT, 1>(a, epi, st);
    case 18:   // the prefetching form keeps ONE gate row per 256-row tile: samples must be whole tiles, else the producer-wave form
      if (epi == EPI_GATE_RES_F32 && a.rows_per_sample % 256 != 0) return launch_n144<DT, 1, 256>(a, epi, st);
      return launch_n144<DT, 0, 256>(a, epi, st);
    case 19: return launch_n144<DT, 1, 256>(a, epi, st);
#ifdef LATTE_GEMM_ABLATE   // measurement build: one consumer wave per SIMD (4 waves x 512 registers, wave tile 128 x 128) on the plain template
    case 14: return launch_w4<DT>(a, epi, st);
    case 15: return launch_cfg<256, 256, 2, 2, DT>(a, epi, st);   // the same wave layout on the plain two-stage template
#endif
    default: return fail(LATTE_ERR_INVALID, "gemm: unknown tile variant");
  }
}

}  // namespace

int gemm_tile_m(int variant) { return variant == 1 || variant == 12 || variant == 13 ? 128 : 256; }   // (18 / 19: 256 x 144)

int gemm_tile_n(int variant) {
  switch (variant) {
    case 3: case 6: case 9: case 14: case 15: return 256;
    case 5: case 8: case 10: case 11: case 17: return 192;
    case 12: case 13: case 18: case 19: return 144;
    default: return 128;
  }
}

// variant 0: pick the tile by a wave-quantisation model.  score = (tiles / (rounds * slots)) * rate where
// slots = resident workgroups on 256 CUs (ping-pong: 1 per CU; 128x128: 2 per CU) and rate is the kernel's
// relative throughput at full occupancy (microbenchmarks).
int gemm_auto_variant(int M, int N, int epi) {
  struct Cand { int variant, bm, bn, slots; float rate; };
  // relative throughput at full occupancy (M = 32768 microbenchmarks): the 256-wide persistent tile has the
  // full-line LDS-transposed epilogue for half-precision outputs (qkv: 220 us vs 250-266 us at 192), for the fp32
  // read-modify-write epilogue the two are within 3 %
  const bool half_out = epi == EPI_BIAS_H16 || epi == EPI_BIAS_GELU_H16;
  // (11 = the rolling 12-wave kernel with the LDS-patch epilogue: for half outputs it ties the 256-wide tile at M = 32768 and wins
  //  wherever 192-wide tiles quantise better -- fc1 at B = 1, 2: 384 / 768 tiles; it needs whole 192-wide tile columns)
  const Cand cands[] = {{9, 256, 256, 256, 1.00f}, {11, 256, 192, 256, half_out ? 0.96f : 0.0f}, {8, 256, 192, 256, half_out ? 0.86f : 0.97f},
                        {7, 256, 128, 256, 0.80f}, {1, 128, 128, 512, 0.78f}};
  int best = 1;
  float best_score = -1.f;
  for (const Cand& c : cands) {
    // the persistent kernels take a partial last tile column as long as every wave's WTN = bn / 4 columns
    // are all inside or all outside N
    if (c.variant == 11 ? (N % 192) != 0 : c.variant >= 7 ? (N % (c.bn / 4)) != 0 : (N % c.bn) != 0) continue;
    const long tiles_n = (N + c.bn - 1) / c.bn;
    const long tiles = (long)((M + c.bm - 1) / c.bm) * tiles_n;
    const long rounds = (tiles + c.slots - 1) / c.slots;
    const float score = (float)tiles / (float)(rounds * c.slots) * c.rate * ((float)N / (float)(tiles_n * c.bn));
    if (score > best_score) { best_score = score; best = c.variant; }
  }
  return best;
}

// The 128 x 144 tile (variant 13) for a gated read-modify-write GEMM: whole 144-wide tile columns and at most one tile per CU
// -- B = 1 at XL/2 (M = 4096, N = 1152: 256 tiles): out-projection 31 -> 23 us, fc2 70 (two-way split K + reduction) -> 54 us
// per launch inside the forward; at B = 2 (512 tiles) it loses to the 256 x 192 producer-wave kernel (45 against 38 us, 113
// against 88).
bool gemm_small_tile_ok(int M, int N, int K) {
  return N % 144 == 0 && K % 64 == 0 && (long)((M + 127) / 128) * (N / 144) <= 256;
}

// (Round 6: the 256 x 144 tile -- variants 18 / 19, gemm_n144_kernel<..., 256> -- was built for the shapes whose 192-wide tiling wastes
//  the last round of the chip (fc1 at B = 1: 384 tiles = 1.5 rounds; the gated GEMMs at B = 2: 192 tiles on 256 CUs) and measured inside the
//  XL/2 forward, profiles/r6_gemm_tile_256x144_inmodel_B1_B2.log: fc1 at B = 1 54.7 - 55.6 us against 52.7 for the 12-wave 256 x 192 kernel,
//  proj / fc2 at B = 2 35.8 / 85.5 against 36.9 / 86.5 -- ties.  The small-batch GEMMs are bound by the bytes a CU can pull through its
//  L2 -> LDS path (~ 50 GB/s per CU): one tile per CU of M N / 256 outputs needs (BM + BN) K 2 bytes whatever its shape, fc2 at B = 1
//  2.5 MB per CU = 50 us.  The variants stay selectable ("gemm_variant_*" = 18 | 19) and tested; no shape rule picks them.)
// What launch_gemm runs when no variant is forced (pure host logic; latte_debug_gemm_choice exposes it to the CPU tests).
int gemm_resolve_variant(int M, int N, int K, int epi) {
  if (epi == EPI_GATE_RES_F32 && gemm_small_tile_ok(M, N, K)) return 13;
  int variant = gemm_auto_variant(M, N, epi);
  // the gated read-modify-write GEMMs on 192-wide tiles run on the 12-wave producer / consumer kernel (gemm_pw.hip, rolling
  // schedule): proj 127 -> 119 us, fc2 316 -> 289 us per launch in the XL/2 forward at B = 8 (same box, round-2 sweep)
  const bool pw_ok = N % 192 == 0 && K % 64 == 0 && K >= 128 && (uint64_t)((M + 255) / 256 * 256) * K * 2 < (1ull << 32) &&
                     (uint64_t)N * K * 2 < (1ull << 32);
  if (variant == 8 && epi == EPI_GATE_RES_F32 && pw_ok) variant = 11;
  if (variant == 11 && !pw_ok) variant = 8;
  // (start cohorts -- GemmArgs::stagger -- stay off: +7 % on the stand-alone fc1 launch, where A streams from HBM,
  //  but -9 % inside the model, where A was just written by the LN kernel and is Infinity-Cache resident)
  return variant;
}

int launch_gemm(const GemmArgs& a_in, int epi, int dtype, int variant, hipStream_t st) {
  GemmArgs a = a_in;
  // grouped tile order of the persistent kernel: the gated-residual GEMMs (192-wide tiles: an A K-tile is 32 KB, a W K-tile
  // 24 KB) walk 4 tile rows together instead of 8 (fc2 in the XL/2 forward at B = 8: 341 -> 333 us, round-2 sweep)
  if (a.group_m == 0 && epi == EPI_GATE_RES_F32) a.group_m = 4;
#ifdef LATTE_GEMM_ABLATE
  if (const char* m = getenv("LATTE_RMW_MODE")) a.rmw_mode = atoi(m);
  if (const char* m = getenv("LATTE_PWR_ABL")) { if (epi == EPI_GATE_RES_F32) a.rmw_mode = atoi(m); }   // ladder rungs of the rolling kernel (gemm_pw.hip)
  if (const char* m = getenv("LATTE_RMW_VARIANT")) { if (epi == EPI_GATE_RES_F32 && variant == 0) variant = atoi(m); }
  if (const char* m = getenv("LATTE_GROUP_M")) {   // "epi:value[,epi:value]" e.g. "2:5" = gated-residual GEMMs walk 5 tile rows together
    for (const char* q = m; q && *q;) {
      const int e_ = atoi(q);
      const char* c_ = strchr(q, ':');
      if (!c_) break;
      if (e_ == epi) a.group_m = atoi(c_ + 1);
      q = strchr(c_, ',');
      if (q) ++q;
    }
  }
#endif
  if (a.A4) {   // fp4 correction pass (GemmArgs::A4 / W4 + row scales): rolling 12-wave kernel, any K % 64 == 0 (rows are padded to K % 256 == 0)
    if (!gemm_lo4_ok(a.M, a.N, a.K)) return fail(LATTE_ERR_INVALID, "gemm: the fp4 correction pass needs N % 192 == 0, K % 64 == 0, K >= 128");
    return launch_gemm_pw(a, epi, dtype, 1, st);
  }
  if (a.A8) {   // correction pass of a split operand (GemmArgs::A8 / W8): the rolling 12-wave kernel is the one that has it
    if (!gemm_lo8_ok(a.M, a.N, a.K)) return fail(LATTE_ERR_INVALID, "gemm: the fp8 correction pass needs N % 192 == 0 and K % 128 == 0");
    return launch_gemm_pw(a, epi, dtype, 1, st);
  }
  if (variant == 0) variant = gemm_resolve_variant(a.M, a.N, a.K, epi);
  if (variant == 10 || variant == 11) return launch_gemm_pw(a, epi, dtype, variant == 11, st);
#ifdef LATTE_GEMM_ABLATE
  if (variant == 17) return launch_gemm_pw(a, epi, dtype, 3, st);   // two-accumulator-set kernel (gemm_pw.hip; measured: loses)
#endif
  const int bn = gemm_tile_n(variant);
  const int nq = variant >= 7 && variant <= 9 ? bn / 4 : bn;   // persistent kernels: partial last tile column in whole wave widths
  if (a.K % 64 != 0 || a.N % nq != 0 || a.M <= 0)
    return fail(LATTE_ERR_INVALID, "gemm: shape not tileable (need K % 64 == 0, N % tileN == 0)");
  if (dtype == LATTE_DTYPE_BF16) return launch_dt<LATTE_DTYPE_BF16>(a, epi, variant, st);
  if (dtype == LATTE_DTYPE_F16) return launch_dt<LATTE_DTYPE_F16>(a, epi, variant, st);
  return fail(LATTE_ERR_INVALID, "gemm: unknown dtype");
}

}  // namespace latte
