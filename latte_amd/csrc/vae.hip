// SD-VAE decoder kernels (gfx950):  AutoencoderKL.decode of diffusers 0.24.0, called by the reference at
// /root/reference/sample/sample.py:113-115 and sample_ddp.py:165-168 (class not vendored: see oracle/vae_oracle.py).
//
// Activations are NHWC half [N, H, W, C] (a pixel's channels are contiguous), so
//   conv3x3     = implicit GEMM  out[p, co] = sum_{tap, ci} in[p + tap, ci] * Wp[co, tap * Cin + ci]
//                 on the MFMA tile machinery of gemm.hip: 128 pixels x 128 output channels per workgroup, BK = 64
//                 (one tap, 64 input channels), both operands K-contiguous, staged by global_load_lds with the
//                 bank swizzle on the SOURCE address.  The zero padding, and the nearest-2x upsample of
//                 Upsample2D, live in the per-lane source address (a padded tap reads a zero page; an upsampled
//                 tap reads pixel (y >> 1, x >> 1)) -- the upsampled tensor is never materialised;
//   GroupNorm   = partial sums per (image, slab, group) -> finalize -> apply (+ SiLU), fp32 statistics;
//   conv_in / post_quant_conv / conv_out (4 -> 4, 4 -> 512, 128 -> 3 channels) = small direct kernels;
//   attention   = 1 head of 512 over H*W tokens: plain GEMMs (gemm.hip) + a row softmax.
#include "mfma_util.h"

namespace latte {
namespace {

__device__ __forceinline__ float h2f_bf16(unsigned int lo16) { return __builtin_bit_cast(float, lo16 << 16); }
template <int DT>
__device__ __forceinline__ void unpack2(unsigned int u, float& a, float& b) {
  if constexpr (DT == LATTE_DTYPE_BF16) {
    a = __builtin_bit_cast(float, u << 16);
    b = __builtin_bit_cast(float, u & 0xffff0000u);
  } else {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    const f16x2 h = __builtin_bit_cast(f16x2, u);
    a = (float)h[0];
    b = (float)h[1];
  }
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

struct ConvArgs {
  const half_t* in;     // [N, Hin, Win, Cin]
  const half_t* w;      // [Cout, 9 * Cin], k = (ky * 3 + kx) * Cin + ci
  const float* bias;    // [Cout]
  const half_t* res;    // nullptr or [N, Hout, Wout, Cout] residual (may alias out)
  half_t* out;          // [N, Hout, Wout, Cout]  (ignored when out32 is set)
  const float* res32;   // fp32 residual (the decoder's residual stream) or nullptr; may alias out32
  float* out32;         // fp32 output: out32 = acc + bias (+ res32), nothing is rounded to half
  const half_t* zeros;  // >= 16 bytes of zeros (padded taps)
  int N, Hin, Win, Cin, Cout, ups;   // Hout = Hin << ups
  int taps3;   // 1: a 3-tap convolution along the image ROWS (w = [Cout, 3 * Cin], k = ky * Cin + ci; the temporal Conv3d (3,1,1) of
               // AutoencoderKLTemporalDecoder on the "image" [frames][h * w] of one video)
};

// 128 x 128 tile, 4 waves (2 x 2), wave tile 64 x 64 = 4 x 4 MFMA 16x16x32 accumulators.
template <int DT>
__global__ void __launch_bounds__(256) conv3x3_kernel(ConvArgs g) {
  constexpr int BM = 128, BN = 128, NW = 4;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int INSTR = BM / 8 / NW;   // 4 row-group instructions per operand per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int Hout = g.Hin << g.ups, Wout = g.Win << g.ups;
  const int M = g.N * Hout * Wout;
  const int K = (g.taps3 ? 3 : 9) * g.Cin;
  int tm, tn;
  tile_coords((M + BM - 1) / BM, g.Cout / BN, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // staging: lane fills bytes [16 * lane, +16) of an 8-row group -> row lrow, chunk position cpos of the image;
  // it fetches logical chunk cpos ^ swizzle(row)
  const int lrow = lane >> 3, cpos = lane & 7;
  const int srow = wave * 8 + lrow;
  const int schunk = cpos ^ ((srow >> 1) & 7);
  int py[INSTR], px[INSTR], pbase[INSTR];   // output pixel (y, x) and image base pixel index of row srow + 32 j
#pragma unroll
  for (int j = 0; j < INSTR; ++j) {
    const int m = m0 + srow + 32 * j;
    if (m < M) {
      const int img = m / (Hout * Wout), rem = m - img * (Hout * Wout);
      py[j] = rem / Wout;
      px[j] = rem - py[j] * Wout;
      pbase[j] = img * g.Hin * g.Win;
    } else {
      py[j] = -4;   // every tap out of range
      px[j] = 0;
      pbase[j] = 0;
    }
  }
  const half_t* b_src = g.w + (size_t)(n0 + srow) * K + schunk * 8;
  const size_t bstride = (size_t)NW * 8 * K;
  const int cpt = g.Cin >> 6;   // K tiles per tap

  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * STAGE + wave * 1024;
    char* sB = sA + A_BYTES;
    const int tap = kt / cpt, c0 = (kt - tap * cpt) << 6;
    const int dy = g.taps3 ? tap - 1 : tap / 3 - 1, dx = g.taps3 ? 0 : tap - (tap / 3) * 3 - 1;
#pragma unroll
    for (int j = 0; j < INSTR; ++j) {
      const int yy = py[j] + dy, xx = px[j] + dx;
      const bool ok = yy >= 0 && yy < Hout && xx >= 0 && xx < Wout;
      const int sy = yy >> g.ups, sx = xx >> g.ups;
      const half_t* src = ok ? g.in + ((size_t)(pbase[j] + sy * g.Win + sx) * g.Cin + c0 + schunk * 8) : g.zeros;
      glds16(src, sA + j * NW * 1024);
    }
#pragma unroll
    for (int j = 0; j < INSTR; ++j) glds16(b_src + j * bstride + kt * 64, sB + j * NW * 1024);
  };

  const int frow = lane & 15;
  const int sw = (lane >> 1) & 7;
  const int chunk0 = ((lane >> 4) ^ sw) * 16;
  const int a_off = (wm * 64 + frow) * 128 + chunk0;
  const int b_off = A_BYTES + (wn * 64 + frow) * 128 + chunk0;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = K / 64;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* sbuf = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const u32x4*)(sbuf + ((a_off + i * 2048) ^ (ks << 6)));
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *(const u32x4*)(sbuf + ((b_off + j * 2048) ^ (ks << 6)));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<DT>(bf[j], af[i], acc[i][j]);
    }
  }

  // lane holds out[pixel m = .. + (lane & 15)][co = .. + (lane >> 4) * 4 + {0..3}]
  const int ncol = n0 + wn * 64 + (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + frow;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = ncol + j * 16;
      const float4 b4 = *(const float4*)(g.bias + n);
      float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
      const size_t o = (size_t)m * g.Cout + n;
      if (g.out32 != nullptr) {   // fp32 stream: the skip path is never rounded (wave-uniform branch)
        if (g.res32 != nullptr) {
          const float4 r4 = *(const float4*)(g.res32 + o);
          v0 += r4.x; v1 += r4.y; v2 += r4.z; v3 += r4.w;
        }
        *(float4*)(g.out32 + o) = make_float4(v0, v1, v2, v3);
        continue;
      }
      if (g.res != nullptr) {
        const u32x2 r2 = *(const u32x2*)(g.res + o);
        float r0, r1, r2f, r3;
        unpack2<DT>(r2[0], r0, r1);
        unpack2<DT>(r2[1], r2f, r3);
        v0 += r0; v1 += r1; v2 += r2f; v3 += r3;
      }
      const u32x2 p = {pack2<DT>(v0, v1), pack2<DT>(v2, v3)};
      *(u32x2*)(g.out + o) = p;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Round 4: the same implicit GEMM on the PING-PONG schedule of gemm_pp_kernel (gemm.hip): 256 pixels x BN output channels per
// workgroup, eight waves = two groups of four (wave tile 128 x BN/4), BK = 64, two LDS stages; per K tile every wave runs a
// fragment-read segment L and a compute segment C, group 1 one segment behind group 0, so on every SIMD one wave multiplies
// while its partner reads.  The plain 128 x 128 kernel above has every wave issue 8 LDS-DMA instructions per 32 MFMAs -- the
// issue holds the wave (gemm_pw.hip's header) -- and ran the decoder's convolutions at 0.31 of the MFMA peak; here it is 4 (A:
// a group's waves gather its own 128 pixel rows) + BN/32 (B, group 0 only) per 16 BN/16 MFMAs.  The gather (zero padding,
// nearest-2x upsample, 3-tap mode) is the per-lane source address as before; each lane keeps the (y, x, image) of its four
// pixel rows in registers for the whole K loop.
template <int BN, int DT>
__global__ void __launch_bounds__(512) conv3x3_pp_kernel(ConvArgs g) {
  constexpr int BM = 256;
  constexpr int WTN = BN / 4, FN = WTN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int AH_INSTR = 4;            // a group's 4 waves gather its 128 pixel rows: 8-row groups wn + 4 j
  constexpr int BG_INSTR = BN / 8 / 4;   // W rows per wave of group 0
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  const int Hout = g.Hin << g.ups, Wout = g.Win << g.ups;
  const int M = g.N * Hout * Wout;
  const int K = (g.taps3 ? 3 : 9) * g.Cin;
  int tm, tn;
  tile_coords((M + BM - 1) / BM, g.Cout / BN, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  const int lrow = lane >> 3, cpos = lane & 7;
  const int srow = wn * 8 + lrow;                       // row inside the group's 32-row step (swizzle: 32 j and 128 grp do not move it)
  const int schunk = cpos ^ ((srow >> 1) & 7);
  // Per pixel row of this lane (row grp*128 + srow + 32 j): without upsampling, the byte offset of the pixel's own channel chunk and
  // a bit mask of the taps that stay inside the image -- a tap is then ONE add of a scalar delta and a select; with the nearest-2x
  // upsample the source pixel of a tap depends on the parities, so (y, x, image) are kept and the address is formed per tap.
  const bool fast = g.ups == 0;
  int pyx[AH_INSTR], pbase[AH_INSTR];                   // ups: (y << 16 | x), image base pixel;  fast: tap mask, centre byte offset
  const int ntap = g.taps3 ? 3 : 9;
#pragma unroll
  for (int j = 0; j < AH_INSTR; ++j) {
    const int m = m0 + grp * 128 + srow + 32 * j;
    if (m < M) {
      const int img = m / (Hout * Wout), rem = m - img * (Hout * Wout);
      const int y = rem / Wout, x = rem - y * Wout;
      if (fast) {
        int mask = 0;
        for (int tap = 0; tap < ntap; ++tap) {
          const int yy = y + (g.taps3 ? tap - 1 : tap / 3 - 1), xx = x + (g.taps3 ? 0 : tap % 3 - 1);
          if (yy >= 0 && yy < Hout && xx >= 0 && xx < Wout) mask |= 1 << tap;
        }
        pyx[j] = mask;
        pbase[j] = (int)(((unsigned)(img * g.Hin * g.Win + y * g.Win + x) * (unsigned)g.Cin + (unsigned)(schunk * 8)) * 2u);
      } else {
        pyx[j] = (y << 16) | x;
        pbase[j] = img * g.Hin * g.Win;
      }
    } else {
      pyx[j] = fast ? 0 : (0x7ff0 << 16);   // every tap out of range
      pbase[j] = 0;
    }
  }
  // buffer addressing (one 32-bit per-lane offset per DMA instruction, scalar offsets for the rest; 64-bit pointers per
  // instruction spilled at BN = 256): a padded tap gets an offset beyond the descriptor's range, which reads as zeros
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.in, 0, (unsigned)((size_t)g.N * g.Hin * g.Win * g.Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.w, 0, (unsigned)((size_t)g.Cout * K * 2), 0x00020000);
  const unsigned voff_b = ((unsigned)srow * (unsigned)K + (unsigned)(schunk * 8)) * 2u;
  const unsigned b_step = 32u * (unsigned)K * 2u;
  const int cpt = g.Cin >> 6;   // K tiles per tap
  typedef __attribute__((address_space(3))) void lds_void_c;

  auto dma_a_half = [&](int kt) {
    char* sA = smem + (kt & 1) * STAGE + grp * 128 * 128 + wn * 1024;
    const int tap = kt / cpt, c0 = (kt - tap * cpt) << 6;
    const int dy = g.taps3 ? tap - 1 : tap / 3 - 1, dx = g.taps3 ? 0 : tap - (tap / 3) * 3 - 1;
    if (fast) {
      const int delta = ((dy * g.Win + dx) * g.Cin + c0) * 2;   // wave-uniform byte step of this tap / channel tile
#pragma unroll
      for (int j = 0; j < AH_INSTR; ++j) {
        const unsigned voff = ((pyx[j] >> tap) & 1) ? (unsigned)(pbase[j] + delta) : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_c*)(sA + j * 4 * 1024), 16, voff, 0u, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < AH_INSTR; ++j) {
      const int yy = (pyx[j] >> 16) + dy, xx = (pyx[j] & 0xffff) + dx;
      const bool ok = yy >= 0 && yy < Hout && xx >= 0 && xx < Wout;
      const int sy = yy >> g.ups, sx = xx >> g.ups;
      const unsigned voff = ok ? ((unsigned)(pbase[j] + sy * g.Win + sx) * (unsigned)g.Cin + (unsigned)(c0 + schunk * 8)) * 2u : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_c*)(sA + j * 4 * 1024), 16, voff, 0u, 0, 0);
    }
  };
  auto dma_b_all = [&](int kt) {
    char* sB = smem + (kt & 1) * STAGE + A_BYTES + wn * 1024;
    const unsigned so = ((unsigned)n0 * (unsigned)K + (unsigned)kt * 64u) * 2u;
#pragma unroll
    for (int j = 0; j < BG_INSTR; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_c*)(sB + j * 4 * 1024), 16, voff_b, so + (unsigned)j * b_step, 0, 0);
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

  const int nk = K / 64;
  dma_a_half(0);                       // prologue: K tile 0 in the main loop's own mapping
  if (grp == 0) dma_b_all(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (nk > 1) {                        // K tile 1 (stage 1 is untouched so far): each group its own pixel rows, group 0 the W tile
    dma_a_half(1);
    if (grp == 0) dma_b_all(1);
  }
  if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger the two groups by one segment

  // Round 5: the DMA runs TWO K tiles ahead, as in gemm_pps_kernel (gemm.hip): K tile kt + 2 is issued into the stage just consumed,
  // behind the barrier that ends C(kt) -- for group 0 that barrier is the one group 1 passes after ITS L(kt), so the shared W tile
  // and group 0's pixel rows of the stage are free; group 1 only ever overwrites its own rows -- and the vmcnt(0) behind C(kt)
  // confirms K tile kt + 1, which was issued a whole iteration earlier (round 4 issued kt + 1 inside L(kt) and waited for it one
  // compute segment later: the gathered rows come out of the L2 / Infinity Cache with less slack than that).
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // own DMA of K tile kt + 1 landed
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) {
      dma_a_half(kt + 2);
      if (grp == 0) dma_b_all(kt + 2);
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();  // balance group 1's extra barrier

  // lane holds out[pixel m = .. + (lane & 15)][co = .. + (lane >> 4) * 4 + {0..3}]  (the plain kernel's epilogue)
  const int ncol = n0 + wn * WTN + (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + grp * 128 + i * 16 + frow;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = ncol + j * 16;
      const float4 b4 = *(const float4*)(g.bias + n);
      float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
      const size_t o = (size_t)m * g.Cout + n;
      if (g.out32 != nullptr) {
        if (g.res32 != nullptr) {
          const float4 r4 = *(const float4*)(g.res32 + o);
          v0 += r4.x; v1 += r4.y; v2 += r4.z; v3 += r4.w;
        }
        *(float4*)(g.out32 + o) = make_float4(v0, v1, v2, v3);
        continue;
      }
      if (g.res != nullptr) {
        const u32x2 r2 = *(const u32x2*)(g.res + o);
        float r0, r1, r2f, r3;
        unpack2<DT>(r2[0], r0, r1);
        unpack2<DT>(r2[1], r2f, r3);
        v0 += r0; v1 += r1; v2 += r2f; v3 += r3;
      }
      const u32x2 p = {pack2<DT>(v0, v1), pack2<DT>(v2, v3)};
      *(u32x2*)(g.out + o) = p;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Round 6: the ping-pong kernel above as a PERSISTENT kernel -- one workgroup per CU walks a sequence of output tiles (XCD-chunked
// grouped order, as gemm_pps_kernel), and the K-tile counter runs ACROSS tiles: the gather / W DMA stays two K tiles ahead of the
// multiplies through a tile boundary, so the first two K tiles of the next output tile are in flight under the epilogue of the current
// one (the non-persistent kernel paid a serial two-K-tile fill, the epilogue and the workgroup launch per 18 ... 72 K tiles; DESIGN.md
// section 4.3 named this as what separates the convolutions' 0.36 from the GEMMs' 0.45).  Two walkers: the DMA walker (tile + K tile of
// the NEXT issue, with the per-lane gather state of ITS tile) and the compute walker (tile of the epilogue).  vmcnt retires in order
// and counts stores: the wait behind C(kt) would also wait for the previous tile's epilogue stores, so the first K tile of a tile
// that follows an epilogue skips it -- its DMA (K tile 1 of the tile) was issued BEFORE that epilogue, whose own bias load has been
// consumed since, which proves it landed.  Same products, same K order, same epilogue arithmetic as conv3x3_pp_kernel: same bits.
template <int BN, int DT>
__global__ void __launch_bounds__(512) conv3x3_pps_kernel(ConvArgs g) {
  constexpr int BM = 256;
  constexpr int WTN = BN / 4, FN = WTN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int AH_INSTR = 4;
  constexpr int BG_INSTR = BN / 8 / 4;
  constexpr int GROUP_M = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  const int Hout = g.Hin << g.ups, Wout = g.Win << g.ups;
  const int M = g.N * Hout * Wout;
  const int K = (g.taps3 ? 3 : 9) * g.Cin;
  const int nk = K / 64;

  // ---- this workgroup's tile sequence
  const int tiles_m = (M + BM - 1) / BM, tiles_n = g.Cout / BN, nwg = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int chunk0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int cnt = q + (xcd < r ? 1 : 0);
  if (slot >= cnt) return;
  auto decode = [&](int wg, int& tm, int& tn) {
    const int per_group = GROUP_M * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_group = wg - group * per_group;
    tm = first_m + in_group % gsz;
    tn = in_group / gsz;
  };
  const int ntile = (cnt - slot + per - 1) / per;
  const int U = ntile * nk;

  const int lrow = lane >> 3, cpos = lane & 7;
  const int srow = wn * 8 + lrow;
  const int schunk = cpos ^ ((srow >> 1) & 7);
  const bool fast = g.ups == 0;
  const int ntap = g.taps3 ? 3 : 9;
  int pyx[AH_INSTR], pbase[AH_INSTR];   // gather state of the DMA walker's tile (conv3x3_pp_kernel)
  auto gather_state = [&](int m0) {
#pragma unroll
    for (int j = 0; j < AH_INSTR; ++j) {
      const int m = m0 + grp * 128 + srow + 32 * j;
      if (m < M) {
        const int img = m / (Hout * Wout), rem = m - img * (Hout * Wout);
        const int y = rem / Wout, x = rem - y * Wout;
        if (fast) {
          int mask = 0;
          for (int tap = 0; tap < ntap; ++tap) {
            const int yy = y + (g.taps3 ? tap - 1 : tap / 3 - 1), xx = x + (g.taps3 ? 0 : tap % 3 - 1);
            if (yy >= 0 && yy < Hout && xx >= 0 && xx < Wout) mask |= 1 << tap;
          }
          pyx[j] = mask;
          pbase[j] = (int)(((unsigned)(img * g.Hin * g.Win + y * g.Win + x) * (unsigned)g.Cin + (unsigned)(schunk * 8)) * 2u);
        } else {
          pyx[j] = (y << 16) | x;
          pbase[j] = img * g.Hin * g.Win;
        }
      } else {
        pyx[j] = fast ? 0 : (0x7ff0 << 16);
        pbase[j] = 0;
      }
    }
  };
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.in, 0, (unsigned)((size_t)g.N * g.Hin * g.Win * g.Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.w, 0, (unsigned)((size_t)g.Cout * K * 2), 0x00020000);
  const unsigned voff_b = ((unsigned)srow * (unsigned)K + (unsigned)(schunk * 8)) * 2u;
  const unsigned b_step = 32u * (unsigned)K * 2u;
  const int cpt = g.Cin >> 6;
  typedef __attribute__((address_space(3))) void lds_void_c;

  auto dma_a_half = [&](int kt, int stg) {
    char* sA = smem + stg * STAGE + grp * 128 * 128 + wn * 1024;
    const int tap = kt / cpt, c0 = (kt - tap * cpt) << 6;
    const int dy = g.taps3 ? tap - 1 : tap / 3 - 1, dx = g.taps3 ? 0 : tap - (tap / 3) * 3 - 1;
    if (fast) {
      const int delta = ((dy * g.Win + dx) * g.Cin + c0) * 2;
#pragma unroll
      for (int j = 0; j < AH_INSTR; ++j) {
        const unsigned voff = ((pyx[j] >> tap) & 1) ? (unsigned)(pbase[j] + delta) : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_c*)(sA + j * 4 * 1024), 16, voff, 0u, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < AH_INSTR; ++j) {
      const int yy = (pyx[j] >> 16) + dy, xx = (pyx[j] & 0xffff) + dx;
      const bool ok = yy >= 0 && yy < Hout && xx >= 0 && xx < Wout;
      const int sy = yy >> g.ups, sx = xx >> g.ups;
      const unsigned voff = ok ? ((unsigned)(pbase[j] + sy * g.Win + sx) * (unsigned)g.Cin + (unsigned)(c0 + schunk * 8)) * 2u : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_c*)(sA + j * 4 * 1024), 16, voff, 0u, 0, 0);
    }
  };
  auto dma_b_all = [&](int n0_, int kt, int stg) {
    char* sB = smem + stg * STAGE + A_BYTES + wn * 1024;
    const unsigned so = ((unsigned)n0_ * (unsigned)K + (unsigned)kt * 64u) * 2u;
#pragma unroll
    for (int j = 0; j < BG_INSTR; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_c*)(sB + j * 4 * 1024), 16, voff_b, so + (unsigned)j * b_step, 0, 0);
  };

  // ---- DMA walker: issues global K tile v (stage v & 1), then advances; entering a tile recomputes the gather state
  int d_pos = slot, d_kt = 0, d_tm, d_tn, v = 0;
  decode(chunk0 + d_pos, d_tm, d_tn);
  gather_state(d_tm * BM);
  auto issue_next = [&]() {
    dma_a_half(d_kt, v & 1);
    if (grp == 0) dma_b_all(d_tn * BN, d_kt, v & 1);
    ++v;
    if (++d_kt == nk) {
      d_kt = 0;
      d_pos += per;
      if (d_pos < cnt) {
        decode(chunk0 + d_pos, d_tm, d_tn);
        gather_state(d_tm * BM);
      }
    }
  };

  const int frow = lane & 15;
  const int sw = (lane >> 1) & 7;
  const int chunk0b = ((lane >> 4) ^ sw) * 16;
  const int a_off = (grp * 128 + frow) * 128 + chunk0b;
  const int b_off = A_BYTES + (wn * WTN + frow) * 128 + chunk0b;

  f32x4 acc[8][FN];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  issue_next();                        // global K tile 0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (U > 1) issue_next();             // global K tile 1 (stage 1 is untouched so far)
  if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger the two groups by one segment

  int pos = slot, tm, tn, u = 0;
  decode(chunk0 + pos, tm, tn);
  bool after_epilogue = false;
  for (;;) {
    for (int kt = 0; kt < nk; ++kt, ++u) {
      const char* sbuf = smem + (u & 1) * STAGE;
      u32x4 bf[2][FN], af[2][8];
      // ---- L(u)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[ks][j] = *(const u32x4*)(sbuf + ((b_off + j * 2048) ^ (ks << 6)));
#pragma unroll
        for (int i = 0; i < 8; ++i) af[ks][i] = *(const u32x4*)(sbuf + ((a_off + i * 2048) ^ (ks << 6)));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // ---- C(u)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = mfma16<DT>(bf[ks][j], af[ks][i], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      // own DMA of global K tile u + 1 landed.  Behind an epilogue the wait would include that epilogue's stores; the DMA in question
      // was issued before the epilogue and the epilogue has consumed a load issued after it: it has landed (in-order retirement).
      if (!(kt == 0 && after_epilogue)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (v < U) issue_next();         // global K tile u + 2 into the stage just consumed
    }
    // ---- epilogue of tile (tm, tn) (conv3x3_pp_kernel's), accumulators cleared
    {
      const int m0 = tm * BM, n0 = tn * BN;
      const int ncol = n0 + wn * WTN + (lane >> 4) * 4;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = m0 + grp * 128 + i * 16 + frow;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int n = ncol + j * 16;
          const float4 b4 = *(const float4*)(g.bias + n);
          float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (m >= M) continue;
          const size_t o = (size_t)m * g.Cout + n;
          if (g.out32 != nullptr) {
            if (g.res32 != nullptr) {
              const float4 r4 = *(const float4*)(g.res32 + o);
              v0 += r4.x; v1 += r4.y; v2 += r4.z; v3 += r4.w;
            }
            *(float4*)(g.out32 + o) = make_float4(v0, v1, v2, v3);
            continue;
          }
          if (g.res != nullptr) {
            const u32x2 r2 = *(const u32x2*)(g.res + o);
            float r0, r1, r2f, r3;
            unpack2<DT>(r2[0], r0, r1);
            unpack2<DT>(r2[1], r2f, r3);
            v0 += r0; v1 += r1; v2 += r2f; v3 += r3;
          }
          const u32x2 p = {pack2<DT>(v0, v1), pack2<DT>(v2, v3)};
          *(u32x2*)(g.out + o) = p;
        }
      }
    }
    after_epilogue = true;
    pos += per;
    if (pos >= cnt) break;
    decode(chunk0 + pos, tm, tn);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();  // balance group 1's extra barrier
}

// ------------------------------------------------------------------------------------------------ GroupNorm
// partial[n][slab][g] = (sum, sumsq) over the slab's pixels and the group's channels.  One thread owns 8 consecutive
// channels of a pixel (16-byte loads); the per-thread sums are combined in a FIXED order (deterministic).
// 8 consecutive channels of a pixel: from the half activation (16-byte load) or from the fp32 residual stream
template <int DT, bool IN32>
__device__ __forceinline__ void load_oct(const void* x, size_t oct_index, float (&f)[8]) {
  if constexpr (IN32) {
    const float4 a = *(const float4*)((const float*)x + oct_index * 8), b = *(const float4*)((const float*)x + oct_index * 8 + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else {
    const u32x4 v = *(const u32x4*)((const half_t*)x + oct_index * 8);
    unpack2<DT>(v[0], f[0], f[1]); unpack2<DT>(v[1], f[2], f[3]);
    unpack2<DT>(v[2], f[4], f[5]); unpack2<DT>(v[3], f[6], f[7]);
  }
}

// Round 4: the pass streams (four independent 16 / 32-byte loads in flight per thread) and its workgroup reduction is two short
// fixed-order steps -- (pixel lanes -> one sum per half-octet column, then columns -> groups) -- instead of 32 threads each walking
// all 256 per-thread partials with a division per step: that serial tail and one load in flight per thread held the pass at
// 2.2 TB/s on the 128 x 256 x 256 map (537 MB fp32 per 16 frames; profiles/r3_kernel_stats_forward_B8_plus_vae_decode.csv).
template <int DT, bool IN32>
__global__ void __launch_bounds__(256) gn_partial_kernel(const void* __restrict__ x, float* __restrict__ partial, int HW,
                                                         int C, int slabs) {
  __shared__ float red[256 * 4];
  __shared__ float col[128 * 2];
  const int n = blockIdx.y, slab = blockIdx.x;
  const int oct_per_px = C >> 3;               // threads per pixel
  const int px_per_it = 256 / oct_per_px;      // C in {128, 256, 512} -> 16, 8, 4 pixels per iteration
  const int oct = threadIdx.x % oct_per_px, pl = threadIdx.x / oct_per_px;
  const int per = (HW + slabs - 1) / slabs;
  const int p0 = slab * per, p1 = min(HW, p0 + per);
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;   // channels [8 oct, +4) and [8 oct + 4, +4)
  auto add = [&](const float (&f)[8]) {
    s0 += f[0] + f[1]; q0 += f[0] * f[0] + f[1] * f[1];
    s0 += f[2] + f[3]; q0 += f[2] * f[2] + f[3] * f[3];
    s1 += f[4] + f[5]; q1 += f[4] * f[4] + f[5] * f[5];
    s1 += f[6] + f[7]; q1 += f[6] * f[6] + f[7] * f[7];
  };
  const size_t base = (size_t)n * HW;
  int p = p0 + pl;
  for (; p + 7 * px_per_it < p1; p += 8 * px_per_it) {   // same pixel order per thread as a one-by-one walk: same sums
    float f0[8], f1[8], f2[8], f3[8], f4[8], f5[8], f6[8], f7[8];
    load_oct<DT, IN32>(x, (base + p) * oct_per_px + oct, f0);
    load_oct<DT, IN32>(x, (base + p + px_per_it) * oct_per_px + oct, f1);
    load_oct<DT, IN32>(x, (base + p + 2 * px_per_it) * oct_per_px + oct, f2);
    load_oct<DT, IN32>(x, (base + p + 3 * px_per_it) * oct_per_px + oct, f3);
    load_oct<DT, IN32>(x, (base + p + 4 * px_per_it) * oct_per_px + oct, f4);
    load_oct<DT, IN32>(x, (base + p + 5 * px_per_it) * oct_per_px + oct, f5);
    load_oct<DT, IN32>(x, (base + p + 6 * px_per_it) * oct_per_px + oct, f6);
    load_oct<DT, IN32>(x, (base + p + 7 * px_per_it) * oct_per_px + oct, f7);
    add(f0); add(f1); add(f2); add(f3); add(f4); add(f5); add(f6); add(f7);
  }
  for (; p + 3 * px_per_it < p1; p += 4 * px_per_it) {
    float f0[8], f1[8], f2[8], f3[8];
    load_oct<DT, IN32>(x, (base + p) * oct_per_px + oct, f0);
    load_oct<DT, IN32>(x, (base + p + px_per_it) * oct_per_px + oct, f1);
    load_oct<DT, IN32>(x, (base + p + 2 * px_per_it) * oct_per_px + oct, f2);
    load_oct<DT, IN32>(x, (base + p + 3 * px_per_it) * oct_per_px + oct, f3);
    add(f0); add(f1); add(f2); add(f3);
  }
  for (; p < p1; p += px_per_it) {
    float f[8];
    load_oct<DT, IN32>(x, (base + p) * oct_per_px + oct, f);
    add(f);
  }
  red[threadIdx.x * 4 + 0] = s0; red[threadIdx.x * 4 + 1] = q0;
  red[threadIdx.x * 4 + 2] = s1; red[threadIdx.x * 4 + 3] = q1;
  __syncthreads();
  // column j = 2 oct + h (4 consecutive channels): sum over the pixel lanes, in lane order
  const int ncol = 2 * oct_per_px;              // 32, 64, 128
  if ((int)threadIdx.x < ncol) {
    const int o = threadIdx.x >> 1, h = threadIdx.x & 1;
    float s = 0.f, q = 0.f;
    for (int l = 0; l < px_per_it; ++l) {
      s += red[(l * oct_per_px + o) * 4 + 2 * h];
      q += red[(l * oct_per_px + o) * 4 + 2 * h + 1];
    }
    col[threadIdx.x * 2] = s;
    col[threadIdx.x * 2 + 1] = q;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int gi = threadIdx.x, cpc = ncol >> 5;   // columns per group: 1, 2, 4
    float s = 0.f, q = 0.f;
    for (int k = 0; k < cpc; ++k) {
      s += col[(gi * cpc + k) * 2];
      q += col[(gi * cpc + k) * 2 + 1];
    }
    float* out = partial + (((size_t)n * slabs + slab) * 32 + gi) * 2;
    out[0] = s;
    out[1] = q;
  }
}

// 256 threads per sample: thread (part = t >> 5, group = t & 31) adds slabs part, part + 8, ... in fp64, then the eight parts of a
// group are added in part order -- a fixed order whatever the slab count (round 3: one thread per group walked all slabs, 8 us of
// dependent loads per launch, 30 launches per decode)
__global__ void __launch_bounds__(256) gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats, int slabs, float count,
                                                          float eps) {
  __shared__ double red[256 * 2];
  const int n = blockIdx.x, gi = threadIdx.x & 31, part = threadIdx.x >> 5;
  // (round 6: four independent chains per thread -- the temporal decoder's GroupNorm is ONE sample of up to 3584 slabs, 448 dependent
  //  load + fp64 add steps per thread were 25 us per launch, 116 launches per video; the order stays fixed for a given slab count)
  double s4[4] = {0.0, 0.0, 0.0, 0.0}, q4[4] = {0.0, 0.0, 0.0, 0.0};
  int k = part;
  for (; k + 24 < slabs; k += 32) {
    float2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const float2*)(partial + (((size_t)n * slabs + k + 8 * u) * 32 + gi) * 2);
#pragma unroll
    for (int u = 0; u < 4; ++u) { s4[u] += v[u].x; q4[u] += v[u].y; }
  }
  for (int u = 0; k < slabs; k += 8, ++u) {
    const float* p = partial + (((size_t)n * slabs + k) * 32 + gi) * 2;
    s4[u & 3] += p[0];
    q4[u & 3] += p[1];
  }
  double s = (s4[0] + s4[1]) + (s4[2] + s4[3]), q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
  red[threadIdx.x * 2] = s;
  red[threadIdx.x * 2 + 1] = q;
  __syncthreads();
  if (threadIdx.x < 32) {
    s = 0.0;
    q = 0.0;
    for (int k = 0; k < 8; ++k) {
      s += red[(k * 32 + gi) * 2];
      q += red[(k * 32 + gi) * 2 + 1];
    }
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[(n * 32 + gi) * 2] = (float)mean;
    stats[(n * 32 + gi) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// Round 4: a thread's channel octet never changes along its grid-stride walk (the stride is a multiple of the octets per pixel), so
// gamma / beta live in registers and the octet spans at most two groups (channels per group >= 4): per element of the walk one
// 32-byte load, two 8-byte statistics loads and the stores -- round 3 looked the statistics up per channel (16 loads) and gamma /
// beta per iteration (4 more) for every 32 bytes of payload.
template <int DT, bool SILU, bool IN32>
__global__ void __launch_bounds__(256) gn_apply_kernel(const void* __restrict__ x, half_t* __restrict__ y,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int HW, int C, size_t total_oct,
                                                       half_t* __restrict__ y_lo) {
  const int cpg = C >> 5, oct_per_px = C >> 3;
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;   // step % oct_per_px == 0
  const int oct = (int)(i0 % oct_per_px), c0 = oct * 8;
  const int g0 = c0 / cpg, g1 = (c0 + 4) / cpg;
  const float4 ga = *(const float4*)(gamma + c0), gb = *(const float4*)(gamma + c0 + 4);
  const float4 ba = *(const float4*)(beta + c0), bb = *(const float4*)(beta + c0 + 4);
  const float gam[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
  const float bet[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
  for (size_t i = i0; i < total_oct; i += step) {
    const int n = (int)((i / oct_per_px) / HW);
    const float2 s0 = *(const float2*)(stats + (n * 32 + g0) * 2), s1 = *(const float2*)(stats + (n * 32 + g1) * 2);   // (mean, rstd)
    float f[8];
    load_oct<DT, IN32>(x, i, f);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float mean = e < 4 ? s0.x : s1.x, rstd = e < 4 ? s0.y : s1.y;
      float t = (f[e] - mean) * rstd * gam[e] + bet[e];
      if constexpr (SILU) t = silu_f(t);
      o[e] = t;
    }
    const u32x4 w = {pack2<DT>(o[0], o[1]), pack2<DT>(o[2], o[3]), pack2<DT>(o[4], o[5]), pack2<DT>(o[6], o[7])};
    *(u32x4*)(y + i * 8) = w;
    if (y_lo != nullptr) {   // the rounding residual as a second half tensor (split-operand convolutions)
      float h[8];
      unpack2<DT>(w[0], h[0], h[1]); unpack2<DT>(w[1], h[2], h[3]);
      unpack2<DT>(w[2], h[4], h[5]); unpack2<DT>(w[3], h[6], h[7]);
      const u32x4 l = {pack2<DT>(o[0] - h[0], o[1] - h[1]), pack2<DT>(o[2] - h[2], o[3] - h[3]),
                       pack2<DT>(o[4] - h[4], o[5] - h[5]), pack2<DT>(o[6] - h[6], o[7] - h[7])};
      *(u32x4*)(y_lo + i * 8) = l;
    }
  }
}

// ------------------------------------------------------------------------------------------------ small convs
// z [N, 4, h, w] fp32 (NCHW, reference layout) * z_scale -> post_quant_conv (1x1, 4 -> 4) -> [N, h, w, 4] fp32
__global__ void post_quant_kernel(const float* __restrict__ z, const float* __restrict__ w, const float* __restrict__ b,
                                  float* __restrict__ out, int N, int hw, float z_scale) {
  const int total = N * hw;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int n = i / hw, p = i - n * hw;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = z[((size_t)n * 4 + c) * hw + p] * z_scale;
    float4 o;
    o.x = b[0] + w[0] * v[0] + w[1] * v[1] + w[2] * v[2] + w[3] * v[3];
    o.y = b[1] + w[4] * v[0] + w[5] * v[1] + w[6] * v[2] + w[7] * v[3];
    o.z = b[2] + w[8] * v[0] + w[9] * v[1] + w[10] * v[2] + w[11] * v[3];
    o.w = b[3] + w[12] * v[0] + w[13] * v[1] + w[14] * v[2] + w[15] * v[3];
    *(float4*)(out + (size_t)i * 4) = o;
  }
}

// conv_in: [N, h, w, 4] fp32 -> [N, h, w, Cout] fp32 (the residual stream), 3x3 pad 1.  wt = [36][Cout] fp32 (k = (ky*3+kx)*4 + ci).
__global__ void __launch_bounds__(256) conv_in_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                      const float* __restrict__ bias, float* __restrict__ out, int N, int H,
                                                      int W, int Cout) {
  __shared__ float patch[36];
  const int p = blockIdx.x;   // output pixel
  const int n = p / (H * W), rem = p - n * H * W, y = rem / W, xw = rem - y * W;
  if (threadIdx.x < 36) {
    const int tap = threadIdx.x >> 2, ci = threadIdx.x & 3;
    const int yy = y + tap / 3 - 1, xx = xw + tap % 3 - 1;
    patch[threadIdx.x] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[(((size_t)n * H + yy) * W + xx) * 4 + ci] : 0.f;
  }
  __syncthreads();
  for (int co = threadIdx.x * 2; co < Cout; co += 512) {
    float a0 = bias[co], a1 = bias[co + 1];
#pragma unroll
    for (int k = 0; k < 36; ++k) {
      const float2 w2 = *(const float2*)(wt + (size_t)k * Cout + co);
      a0 = fmaf(patch[k], w2.x, a0);
      a1 = fmaf(patch[k], w2.y, a1);
    }
    *(float2*)(out + (size_t)p * Cout + co) = make_float2(a0, a1);
  }
}

// conv_out: [N, H, W, C] half -> 3 channels, 3x3 pad 1.  wt = [3][9 * C] fp32.  One wave per output pixel row
// segment: lanes over pixels, channels in the inner loop (16-byte loads).
// out_mode 0: fp32 NCHW [N, 3, H, W] (the reference's .sample); 1: uint8 NHWC [N, H, W, 3] = sample.py:122
template <int DT>
__global__ void __launch_bounds__(256) conv_out_kernel(const half_t* __restrict__ x, const float* __restrict__ wt,
                                                       const float* __restrict__ bias, void* __restrict__ out, int N, int H,
                                                       int W, int C, int out_mode, const half_t* __restrict__ x_lo) {
  extern __shared__ float wl[];   // [3][9 * C]
  for (int i = threadIdx.x; i < 27 * C; i += 256) wl[i] = wt[i];
  __syncthreads();
  const int total = N * H * W;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= total) return;
  const int n = p / (H * W), rem = p - n * H * W, y = rem / W, xw = rem - y * W;
  float a0 = bias[0], a1 = bias[1], a2 = bias[2];
  for (int tap = 0; tap < 9; ++tap) {
    const int yy = y + tap / 3 - 1, xx = xw + tap % 3 - 1;
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
    const half_t* px = x + (((size_t)n * H + yy) * W + xx) * C;
    const float* w0 = wl + tap * C;
    const float* w1 = wl + 9 * C + tap * C;
    const float* w2 = wl + 18 * C + tap * C;
    for (int c = 0; c < C; c += 8) {
      const u32x4 v = *(const u32x4*)(px + c);
      float f[8];
      unpack2<DT>(v[0], f[0], f[1]); unpack2<DT>(v[1], f[2], f[3]);
      unpack2<DT>(v[2], f[4], f[5]); unpack2<DT>(v[3], f[6], f[7]);
      if (x_lo != nullptr) {   // split input: the f16 rounding residual of the GroupNorm output (the weights here are fp32 already)
        const u32x4 l = *(const u32x4*)(x_lo + (px - x) + c);
        float g[8];
        unpack2<DT>(l[0], g[0], g[1]); unpack2<DT>(l[1], g[2], g[3]);
        unpack2<DT>(l[2], g[4], g[5]); unpack2<DT>(l[3], g[6], g[7]);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += g[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a0 = fmaf(f[e], w0[c + e], a0);
        a1 = fmaf(f[e], w1[c + e], a1);
        a2 = fmaf(f[e], w2[c + e], a2);
      }
    }
  }
  if (out_mode == 0) {
    float* o = (float*)out;
    const size_t hw = (size_t)H * W;
    o[((size_t)n * 3 + 0) * hw + rem] = a0;
    o[((size_t)n * 3 + 1) * hw + rem] = a1;
    o[((size_t)n * 3 + 2) * hw + rem] = a2;
  } else {
#pragma clang fp contract(off)
    unsigned char* o = (unsigned char*)out + (size_t)p * 3;
    const float v[3] = {a0, a1, a2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float t = (v[c] * 0.5f + 0.5f) * 255.0f + 0.5f;     // sample.py:122
      t = fminf(fmaxf(t, 0.0f), 255.0f);
      o[c] = (unsigned char)t;                             // .to(torch.uint8) truncates
    }
  }
}

// conv_out for C == 128 and W % 16 == 0 (round 4).  The kernel above gives every thread one pixel: a wave's 16-byte loads then hit
// 64 different 256-byte pixel rows per instruction and the launch ran at 0.86 ms per 16-frame video for 268 MB of input (0.3 TB/s).
// Here 16 lanes share a pixel (lane = (pixel lane >> 4, 8-channel chunk lane & 15)), a wave covers 16 consecutive pixels of a row in
// four groups of four, so every load instruction reads 1 KB of contiguous channels; the 24 weights of a (tap, chunk) are read from
// LDS once per tap and reused for the four pixel groups; the three outputs of a pixel are reduced over its 16 lanes at the end.
template <int DT>
__global__ void __launch_bounds__(256) conv_out_c128_kernel(const half_t* __restrict__ x, const float* __restrict__ wt,
                                                            const float* __restrict__ bias, void* __restrict__ out, int N, int H,
                                                            int W, int out_mode, const half_t* __restrict__ x_lo) {
  constexpr int C = 128;
  extern __shared__ float wl[];   // [3][9 * C]
  for (int i = threadIdx.x; i < 27 * C; i += 256) wl[i] = wt[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, chunk = lane & 15, pix = lane >> 4;
  const long total = (long)N * H * W;
  const long base = ((long)blockIdx.x * 4 + wave) * 16;      // 16 consecutive pixels of one row (W % 16 == 0)
  if (base >= total) return;
  const int n = (int)(base / ((long)H * W));
  const int rem = (int)(base - (long)n * H * W), y = rem / W, x0 = rem - y * W;
  float acc[4][3];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g][0] = acc[g][1] = acc[g][2] = 0.f;
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1, yy = y + dy;
    if (yy < 0 || yy >= H) continue;                          // wave-uniform
    float w[3][8];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      const float4 wa = *(const float4*)(wl + o * 9 * C + tap * C + chunk * 8), wb = *(const float4*)(wl + o * 9 * C + tap * C + chunk * 8 + 4);
      w[o][0] = wa.x; w[o][1] = wa.y; w[o][2] = wa.z; w[o][3] = wa.w; w[o][4] = wb.x; w[o][5] = wb.y; w[o][6] = wb.z; w[o][7] = wb.w;
    }
    const half_t* rowp = x + ((size_t)n * H + yy) * W * C + chunk * 8;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int xx = x0 + g * 4 + pix + dx;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (xx >= 0 && xx < W) v = *(const u32x4*)(rowp + (size_t)xx * C);
      float f[8];
      unpack2<DT>(v[0], f[0], f[1]); unpack2<DT>(v[1], f[2], f[3]);
      unpack2<DT>(v[2], f[4], f[5]); unpack2<DT>(v[3], f[6], f[7]);
      if (x_lo != nullptr && xx >= 0 && xx < W) {   // split input (see conv_out_kernel)
        const u32x4 l = *(const u32x4*)(x_lo + (rowp - x) + (size_t)xx * C);
        float q[8];
        unpack2<DT>(l[0], q[0], q[1]); unpack2<DT>(l[1], q[2], q[3]);
        unpack2<DT>(l[2], q[4], q[5]); unpack2<DT>(l[3], q[6], q[7]);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += q[e];
      }
#pragma unroll
      for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][o] = fmaf(f[e], w[o][e], acc[g][o]);
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float a = acc[g][o];
      a += __shfl_xor(a, 1, 64);
      a += __shfl_xor(a, 2, 64);
      a += __shfl_xor(a, 4, 64);
      a += __shfl_xor(a, 8, 64);
      acc[g][o] = a + bias[o];
    }
  if (chunk == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long p = base + g * 4 + pix;
      if (out_mode == 0) {
        float* o = (float*)out;
        const size_t hw = (size_t)H * W;
        const size_t r = (size_t)(rem + g * 4 + pix);
        o[((size_t)n * 3 + 0) * hw + r] = acc[g][0];
        o[((size_t)n * 3 + 1) * hw + r] = acc[g][1];
        o[((size_t)n * 3 + 2) * hw + r] = acc[g][2];
      } else {
#pragma clang fp contract(off)
        unsigned char* o = (unsigned char*)out + (size_t)p * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float t = (acc[g][c] * 0.5f + 0.5f) * 255.0f + 0.5f;     // sample.py:122
          t = fminf(fmaxf(t, 0.0f), 255.0f);
          o[c] = (unsigned char)t;                                 // .to(torch.uint8) truncates
        }
      }
    }
  }
}

// P[row, :] = softmax(scale * S[row, :]) -> half; one wave per row (L % 64 == 0, L <= 4096)
template <int DT>
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* __restrict__ s, half_t* __restrict__ p, int rows, int L,
                                                           float scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sr = s + (size_t)row * L;
  float mx = -1e30f;
  for (int c = lane; c < L; c += 64) mx = fmaxf(mx, sr[c] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int c = lane; c < L; c += 64) sum += __expf(sr[c] * scale - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float inv = 1.0f / sum;
  half_t* pr = p + (size_t)row * L;
  for (int c = lane; c < L; c += 64) {
    const float v = __expf(sr[c] * scale - mx) * inv;
    if constexpr (DT == LATTE_DTYPE_BF16) {
      const __bf16 h = (__bf16)v;
      pr[c] = __builtin_bit_cast(half_t, h);
    } else {
      const _Float16 h = (_Float16)v;
      pr[c] = __builtin_bit_cast(half_t, h);
    }
  }
}

// [Cout, Cin, 3, 3] fp32 -> [Cout, 9 * Cin] half, k = (ky * 3 + kx) * Cin + ci
template <int DT>
__global__ void pack_conv_w_kernel(const float* __restrict__ w, half_t* __restrict__ out, int Cout, int Cin, half_t* __restrict__ out_lo) {
  const size_t total = (size_t)Cout * Cin * 9;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i / ((size_t)Cin * 9));
    const int r = (int)(i - (size_t)co * Cin * 9);
    const int tap = r / Cin, ci = r - tap * Cin;
    const float v = w[((size_t)co * Cin + ci) * 9 + tap];
    if constexpr (DT == LATTE_DTYPE_BF16) {
      const __bf16 h = (__bf16)v;
      out[i] = __builtin_bit_cast(half_t, h);
    } else {
      const _Float16 h = (_Float16)v;
      out[i] = __builtin_bit_cast(half_t, h);
      if (out_lo != nullptr) out_lo[i] = __builtin_bit_cast(half_t, (_Float16)(v - (float)h));   // the weight's f16 rounding residual
    }
  }
}

// [Cout, Cin, 3, 3] fp32 -> [9 * Cin][Cout] fp32 (conv_in) or [Cout][9 * Cin] fp32 (conv_out: tr = 0)
__global__ void pack_small_w_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int tr) {
  const int total = Cout * Cin * 9;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int co = i / (Cin * 9), r = i - co * Cin * 9;
    const int ci = r / 9, tap = r - ci * 9;
    const int k = tap * Cin + ci;
    if (tr) out[(size_t)k * Cout + co] = w[i];
    else out[(size_t)co * 9 * Cin + k] = w[i];
  }
}

// Conv3d weight [Cout, Cin, 3, 1, 1] -> half [Cout][3 * Cin] (k = tap * Cin + ci), scaled by sigmoid(*mix) when mix != nullptr
// (AlphaBlender with switch_spatial_to_temporal_mix: out = x_spatial + sigmoid(mix_factor) * temporal branch)
template <int DT>
__global__ void pack_conv_t_kernel(const float* __restrict__ w, half_t* __restrict__ out, int Cout, int Cin, const float* __restrict__ mix,
                                   half_t* __restrict__ out_lo) {
  const float sc = mix ? 1.0f / (1.0f + __expf(-mix[0])) : 1.0f;
  const size_t total = (size_t)Cout * Cin * 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i / ((size_t)Cin * 3));
    const int r = (int)(i - (size_t)co * Cin * 3);
    const int tap = r / Cin, ci = r - tap * Cin;
    const float v = w[((size_t)co * Cin + ci) * 3 + tap] * sc;
    if constexpr (DT == LATTE_DTYPE_BF16) {
      const __bf16 h = (__bf16)v;
      out[i] = __builtin_bit_cast(half_t, h);
      if (out_lo) out_lo[i] = __builtin_bit_cast(half_t, (__bf16)(v - (float)h));
    } else {
      const _Float16 h = (_Float16)v;
      out[i] = __builtin_bit_cast(half_t, h);
      if (out_lo) out_lo[i] = __builtin_bit_cast(half_t, (_Float16)(v - (float)h));
    }
  }
}
__global__ void scale_by_sigmoid_kernel(const float* __restrict__ in, float* __restrict__ out, int n, const float* __restrict__ mix) {
  const float sc = 1.0f / (1.0f + __expf(-mix[0]));
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * sc;
}
// time_conv_out: Conv3d(3, 3, (3, 1, 1), padding (1, 0, 0)) over the T frames of one video, fp32 NCHW in;
// out_mode 0: fp32 [T, 3, H, W];  1: uint8 [T, H, W, 3] = ((v * 0.5 + 0.5) * 255 + 0.5).clamp(0, 255)  (sample.py:122)
__global__ void time_conv_out_kernel(const float* __restrict__ in, const float* __restrict__ w /* [3][3][3] = co, ci, tap */,
                                     const float* __restrict__ bias, void* __restrict__ out, int T, int HW, int out_mode) {
  const size_t total = (size_t)T * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int f = (int)(i / HW);
    const size_t p = i - (size_t)f * HW;
    float o[3] = {bias[0], bias[1], bias[2]};
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      const int ff = f + tap - 1;
      if (ff < 0 || ff >= T) continue;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float v = in[((size_t)ff * 3 + ci) * HW + p];
#pragma unroll
        for (int co = 0; co < 3; ++co) o[co] += w[(co * 3 + ci) * 3 + tap] * v;
      }
    }
    if (out_mode == 0) {
#pragma unroll
      for (int co = 0; co < 3; ++co) ((float*)out)[((size_t)f * 3 + co) * HW + p] = o[co];
    } else {
#pragma unroll
      for (int co = 0; co < 3; ++co) {
        const float q = fminf(fmaxf((o[co] * 0.5f + 0.5f) * 255.0f + 0.5f, 0.0f), 255.0f);
        ((unsigned char*)out)[i * 3 + co] = (unsigned char)q;
      }
    }
  }
}

inline int grid_for(size_t n, int block) {
  size_t g = (n + block - 1) / block;
  return (int)(g > 8192 ? 8192 : (g == 0 ? 1 : g));
}

}  // namespace

int launch_conv3x3(const half_t* in, const half_t* w, const float* bias, const half_t* res, half_t* out,
                   const half_t* zeros, int N, int Hin, int Win, int Cin, int Cout, int ups, int dtype, hipStream_t st,
                   const float* res32, float* out32, int taps3) {
  if (Cin % 64 != 0 || Cout % 128 != 0) return fail(LATTE_ERR_INVALID, "conv3x3: need Cin % 64 == 0 and Cout % 128 == 0");
  if (!out && !out32) return fail(LATTE_ERR_INVALID, "conv3x3: no output");
  if (taps3 && ups) return fail(LATTE_ERR_INVALID, "conv3x3: the 3-tap form has no upsampling");
  ConvArgs a{in, w, bias, res, out, res32, out32, zeros, N, Hin, Win, Cin, Cout, ups, taps3};
  const int M = N * (Hin << ups) * (Win << ups);
  if (dtype != LATTE_DTYPE_F16) return fail(LATTE_ERR_INVALID, "conv3x3: the VAE kernels are built for f16 operands only");
  // the ping-pong kernel (256 pixels x 128 | 256 channels) wherever its tiles fill the chip; small maps keep the 128 x 128 tile
  // (latte_debug_set_choice("conv_kernel", 1) forces the plain kernel: A/B tests)
  const int bn = Cout % 256 == 0 ? 256 : 128;
  const int pp_tiles = ((M + 255) / 256) * (Cout / bn);
  // its gather addresses the input through 32-bit buffer offsets and, with the upsample, packs (y, x) into 16 bits each
  const bool pp_ok = (uint64_t)N * Hin * Win * Cin * 2 < (1ull << 32) && (uint64_t)Cout * 9 * Cin * 2 < (1ull << 32) &&
                     (!ups || ((Hin << ups) < 32768 && (Win << ups) < 65536));
  // round 6: the persistent form of the ping-pong kernel (conv3x3_pps_kernel) is bit-identical and SLOWER (SD-VAE decode: convolutions
  // 11.5 -> 12.2 ms; temporal decoder chunk 117 -> 137 ms, profiles/r6_vae_persistent_conv_ab.log): it runs only when conv_kernel 4 asks for it
  if (pp_ok && debug_choice(DBG_CONV_KERNEL) == 4) {
    const int nblk = pp_tiles >= 256 ? 256 : (pp_tiles + 7) / 8 * 8;
    if (bn == 256) {
      constexpr int LDS_PP = 2 * (256 + 256) * 128;
      static std::atomic<uint64_t> attr_a{0};
      if (int rc_ = ensure_dynamic_lds((const void*)conv3x3_pps_kernel<256, LATTE_DTYPE_F16>, LDS_PP, attr_a)) return rc_;
      hipLaunchKernelGGL((conv3x3_pps_kernel<256, LATTE_DTYPE_F16>), dim3(nblk), dim3(512), LDS_PP, st, a);
    } else {
      constexpr int LDS_PP = 2 * (256 + 128) * 128;
      static std::atomic<uint64_t> attr_b{0};
      if (int rc_ = ensure_dynamic_lds((const void*)conv3x3_pps_kernel<128, LATTE_DTYPE_F16>, LDS_PP, attr_b)) return rc_;
      hipLaunchKernelGGL((conv3x3_pps_kernel<128, LATTE_DTYPE_F16>), dim3(nblk), dim3(512), LDS_PP, st, a);
    }
    kprof_mark(VC_CONV3, st);
    LATTE_HIP(hipGetLastError());
    return LATTE_OK;
  }
  if (pp_ok && debug_choice(DBG_CONV_KERNEL) != 1 && (pp_tiles >= 192 || debug_choice(DBG_CONV_KERNEL) >= 2)) {
    if (bn == 256) {
      constexpr int LDS_PP = 2 * (256 + 256) * 128;
      static std::atomic<uint64_t> attr_a{0};
      if (int rc_ = ensure_dynamic_lds((const void*)conv3x3_pp_kernel<256, LATTE_DTYPE_F16>, LDS_PP, attr_a)) return rc_;
      hipLaunchKernelGGL((conv3x3_pp_kernel<256, LATTE_DTYPE_F16>), dim3(pp_tiles), dim3(512), LDS_PP, st, a);
    } else {
      constexpr int LDS_PP = 2 * (256 + 128) * 128;
      static std::atomic<uint64_t> attr_b{0};
      if (int rc_ = ensure_dynamic_lds((const void*)conv3x3_pp_kernel<128, LATTE_DTYPE_F16>, LDS_PP, attr_b)) return rc_;
      hipLaunchKernelGGL((conv3x3_pp_kernel<128, LATTE_DTYPE_F16>), dim3(pp_tiles), dim3(512), LDS_PP, st, a);
    }
    kprof_mark(VC_CONV3, st);
    LATTE_HIP(hipGetLastError());
    return LATTE_OK;
  }
  const int tiles = ((M + 127) / 128) * (Cout / 128);
  constexpr int LDS = 2 * 256 * 128;
  static std::atomic<uint64_t> attr_f16{0};
  if (int rc_ = ensure_dynamic_lds((const void*)conv3x3_kernel<LATTE_DTYPE_F16>, LDS, attr_f16)) return rc_;
  hipLaunchKernelGGL(conv3x3_kernel<LATTE_DTYPE_F16>, dim3(tiles), dim3(256), LDS, st, a);
  kprof_mark(VC_CONV3, st);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_groupnorm(const void* x, int x_is_f32, half_t* y, const float* gamma, const float* beta, float* partial, float* stats,
                     int N, int HW, int C, int silu, int dtype, hipStream_t st, float eps, int max_slabs, half_t* y_lo) {
  if (C != 128 && C != 256 && C != 512) return fail(LATTE_ERR_INVALID, "groupnorm: C must be 128, 256 or 512");
  int slabs = HW / 256;   // >= 256 pixels per slab (round 3: 1024 -- the 64 x 64 and 128 x 128 maps then had 64 / 256 workgroups for 256 CUs)
  if (slabs < 1) slabs = 1;
  if (slabs > max_slabs) slabs = max_slabs;
  const size_t total_oct = (size_t)N * HW * C / 8;
  if (dtype != LATTE_DTYPE_F16) return fail(LATTE_ERR_INVALID, "groupnorm: the VAE kernels are built for f16 operands only");
  if (x_is_f32) hipLaunchKernelGGL((gn_partial_kernel<LATTE_DTYPE_F16, true>), dim3(slabs, N), dim3(256), 0, st, x, partial, HW, C, slabs);
  else hipLaunchKernelGGL((gn_partial_kernel<LATTE_DTYPE_F16, false>), dim3(slabs, N), dim3(256), 0, st, x, partial, HW, C, slabs);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(N), dim3(256), 0, st, partial, stats, slabs, (float)HW * (float)(C / 32), eps);
  kprof_mark(VC_GN_STATS, st);
  const dim3 grid(grid_for(total_oct, 256));
#define GN_APPLY(S, I) hipLaunchKernelGGL((gn_apply_kernel<LATTE_DTYPE_F16, S, I>), grid, dim3(256), 0, st, x, y, stats, gamma, beta, HW, C, total_oct, y_lo)
  if (silu) { if (x_is_f32) GN_APPLY(true, true); else GN_APPLY(true, false); }
  else      { if (x_is_f32) GN_APPLY(false, true); else GN_APPLY(false, false); }
#undef GN_APPLY
  kprof_mark(VC_GN_APPLY, st);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}
int groupnorm_max_slabs() { return 256; }   // (round 3: 64 -- 1024 workgroups on the largest map, 64 iterations of one load each per thread)

int launch_post_quant(const float* z, const float* w, const float* b, float* out, int N, int hw, float z_scale, hipStream_t st) {
  hipLaunchKernelGGL(post_quant_kernel, dim3(grid_for((size_t)N * hw, 256)), dim3(256), 0, st, z, w, b, out, N, hw, z_scale);
  kprof_mark(VC_SMALL, st);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_conv_in(const float* x, const float* wt, const float* bias, float* out, int N, int H, int W, int Cout, hipStream_t st) {
  hipLaunchKernelGGL(conv_in_kernel, dim3(N * H * W), dim3(256), 0, st, x, wt, bias, out, N, H, W, Cout);
  kprof_mark(VC_SMALL, st);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_conv_out(const half_t* x, const float* wt, const float* bias, void* out, int N, int H, int W, int C, int out_mode,
                    int dtype, hipStream_t st, const half_t* x_lo) {
  const int total = N * H * W;
  const size_t lds = (size_t)27 * C * sizeof(float);
  if (dtype != LATTE_DTYPE_F16) return fail(LATTE_ERR_INVALID, "conv_out: the VAE kernels are built for f16 operands only");
  if (C == 128 && W % 16 == 0)   // 64 pixels per workgroup (4 waves x 16)
    hipLaunchKernelGGL(conv_out_c128_kernel<LATTE_DTYPE_F16>, dim3((total + 63) / 64), dim3(256), lds, st, x, wt, bias, out, N, H, W, out_mode, x_lo);
  else
    hipLaunchKernelGGL(conv_out_kernel<LATTE_DTYPE_F16>, dim3((total + 255) / 256), dim3(256), lds, st, x, wt, bias, out, N, H, W, C, out_mode, x_lo);
  kprof_mark(VC_SMALL, st);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_softmax_rows(const float* s, half_t* p, int rows, int L, float scale, int dtype, hipStream_t st) {
  if (dtype != LATTE_DTYPE_F16) return fail(LATTE_ERR_INVALID, "softmax_rows: the VAE kernels are built for f16 operands only");
  hipLaunchKernelGGL(softmax_rows_kernel<LATTE_DTYPE_F16>, dim3((rows + 3) / 4), dim3(256), 0, st, s, p, rows, L, scale);
  kprof_mark(VC_ATTN, st);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_pack_conv_w(const float* w, half_t* out, int Cout, int Cin, int dtype, hipStream_t st, half_t* out_lo) {
  const size_t n = (size_t)Cout * Cin * 9;
  if (dtype != LATTE_DTYPE_F16) return fail(LATTE_ERR_INVALID, "pack_conv_w: the VAE kernels are built for f16 operands only");
  hipLaunchKernelGGL(pack_conv_w_kernel<LATTE_DTYPE_F16>, dim3(grid_for(n, 256)), dim3(256), 0, st, w, out, Cout, Cin, out_lo);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_pack_conv_t(const float* w, half_t* out, int Cout, int Cin, const float* mix, int dtype, hipStream_t st, half_t* out_lo) {
  if (dtype != LATTE_DTYPE_F16) return fail(LATTE_ERR_INVALID, "pack_conv_t: the VAE kernels are built for f16 operands only");
  hipLaunchKernelGGL(pack_conv_t_kernel<LATTE_DTYPE_F16>, dim3(grid_for((size_t)Cout * Cin * 3, 256)), dim3(256), 0, st, w, out, Cout, Cin, mix,
                     out_lo);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}
int launch_scale_by_sigmoid(const float* in, float* out, int n, const float* mix, hipStream_t st) {
  hipLaunchKernelGGL(scale_by_sigmoid_kernel, dim3((n + 255) / 256), dim3(256), 0, st, in, out, n, mix);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}
int launch_time_conv_out(const float* in, const float* w, const float* bias, void* out, int T, int HW, int out_mode, hipStream_t st) {
  hipLaunchKernelGGL(time_conv_out_kernel, dim3(grid_for((size_t)T * HW, 256)), dim3(256), 0, st, in, w, bias, out, T, HW, out_mode);
  kprof_mark(VC_SMALL, st);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_pack_small_w(const float* w, float* out, int Cout, int Cin, int transpose, hipStream_t st) {
  hipLaunchKernelGGL(pack_small_w_kernel, dim3(grid_for((size_t)Cout * Cin * 9, 256)), dim3(256), 0, st, w, out, Cout, Cin, transpose);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

}  // namespace latte
