// Small-kernel consolidation of the training step (round 6b).  The backward of a block used to end in a chain of ~25 tiny
// launches -- six per-sample finalize kernels, four column-sum reductions, the adaLN linear's bias / weight / input gradients as
// strided fp32 GEMMs with K = batch, their split reductions, the loss-scale pass -- each a few microseconds of work behind
// ~8 - 10 us of launch-to-launch latency on the stream (rocprofv3: 320 launches below 12 us per step = 2.6 of 21.4 ms).  Here:
//
//   stage_finalize_kernel   ONE launch per block stage: per-sample sums of the row-run partials -> the six modulation gradients
//                           (latte.py:178-180: shift / scale / gate of both branches), the adaLN linear's bias and weight
//                           gradients from them (dW[n, k] = sum_b dmod[b, n] silu(c)[b, k]: K = batch, an outer product -- written
//                           once at HBM speed), and the four bias gradients of the block's linears from the column partials
//                           their producers left (the gate backward for proj / fc2; for qkv / fc1 the weight-gradient launch, gemm_tn.hip, or colsum_half).  Gradients leave the
//                           loss-scaled domain here (x 1 / scale, a power of two: exact).
//   adaln_dc_kernel         d silu(c)[b, k] = sum over ALL adaLN linears of dmod[b, n] W[n, k], once per step (the stages only
//                           need it at the very end, train_engine.cpp) instead of one split GEMM + reduce + add per block
//   pack_weights_kernel     the half operand copies of every block weight ([N, K] and [K, N]) in one launch from a device table
//
// Every reduction runs in a fixed order (deterministic); nothing here is on the inference path.
#include <cmath>

#include "common.h"

namespace latte {
namespace {

template <int DT>
__device__ __forceinline__ unsigned short f2h_t(float v) {
  if constexpr (DT == LATTE_DTYPE_BF16) {
    const __bf16 h = (__bf16)v;
    return __builtin_bit_cast(unsigned short, h);
  } else {
    const _Float16 h = (_Float16)v;
    return __builtin_bit_cast(unsigned short, h);
  }
}

constexpr int FIN_SB = 8;        // samples per pass of the finalize / dc kernels
constexpr int FIN_DMAX = 1280;   // hidden size bound of the trainer

// blocks [0, n_mod D / 64): modulation chunk (bid / (D / 64)), 64 columns;  then the bias column sums, 64 columns per block
__global__ void __launch_bounds__(1024) stage_finalize_kernel(StageFinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float fin_sm[];   // red [FIN_SB][16][64] | dm_s [FIN_SB][64] | cs_s [FIN_SB][D]
  float (*red)[16][64] = (float (*)[16][64])fin_sm;
  float (*dm_s)[64] = (float (*)[64])(fin_sm + FIN_SB * 16 * 64);
  float* cs_s = fin_sm + FIN_SB * 16 * 64 + FIN_SB * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int D = a.D, cgs = D >> 6;
  const float inv = a.scaler ? 1.0f / a.scaler[0] : 1.0f;
  int bid = blockIdx.x;
  if (bid < a.n_mod * cgs) {
    const int chunk = bid / cgs, cg = bid - chunk * cgs;
    const int col = cg * 64 + tx;
    const float* src = a.mod_src[chunk];
    const int nsum = a.mod_nsum[chunk], which = a.mod_which[chunk], rows = a.rows_per_sample;
    for (int bs = 0; bs < a.B; bs += FIN_SB) {
      const int nb = min(FIN_SB, a.B - bs);
      float acc[FIN_SB];
#pragma unroll
      for (int b = 0; b < FIN_SB; ++b) acc[b] = 0.f;
      {   // the samples' partial rows side by side: FIN_SB independent loads per step
        const float* p = src + (((size_t)bs * rows + ty) * nsum + which) * D + col;
        const size_t step = (size_t)16 * nsum * D, sample = (size_t)rows * nsum * D;
        for (int r = ty; r < rows; r += 16, p += step) {
#pragma unroll
          for (int b = 0; b < FIN_SB; ++b)
            if (b < nb) acc[b] += p[b * sample];
        }
      }
#pragma unroll
      for (int b = 0; b < FIN_SB; ++b) red[b][ty][tx] = acc[b];
      for (int i = threadIdx.x; i < nb * (D >> 2); i += 1024) {
        const int b = i / (D >> 2), q = i - b * (D >> 2);
        ((float4*)(cs_s + (size_t)b * D))[q] = ((const float4*)(a.csilu + (size_t)(bs + b) * D))[q];
      }
      __syncthreads();
      if (ty < nb) {   // wave ty owns sample bs + ty
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[ty][k][tx];
        dm_s[ty][tx] = s;
        a.dmod[(size_t)(bs + ty) * a.dmod_stride + chunk * D + col] = s;   // stays in the loss-scaled domain (feeds d silu(c))
      }
      __syncthreads();
      if (ty == 0) {
        float s = 0.f;
        for (int b = 0; b < nb; ++b) s += dm_s[b][tx];
        float* o = a.db + chunk * D + col;
        *o = bs ? *o + s * inv : s * inv;
      }
      const int nq = D >> 2;
      for (int i = threadIdx.x; i < 64 * nq; i += 1024) {
        const int j = i / nq, q = i - j * nq;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = 0; b < nb; ++b) {
          const float d = dm_s[b][j];
          const float4 c = ((const float4*)(cs_s + (size_t)b * D))[q];
          o.x += d * c.x; o.y += d * c.y; o.z += d * c.z; o.w += d * c.w;
        }
        float4* w = (float4*)(a.dW + ((size_t)chunk * D + cg * 64 + j) * D) + q;
        o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
        if (bs) {
          const float4 p = *w;
          o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        *w = o;
      }
      __syncthreads();
    }
    return;
  }
  bid -= a.n_mod * cgs;
  int s = 0;
  while (s + 1 < a.n_bias && bid >= a.bias_blk[s + 1]) ++s;
  const int col = (bid - a.bias_blk[s]) * 64 + tx;
  float acc = 0.f;
  if (col < a.bias_cols[s]) {   // thread row ty adds rows ty, ty + 16, ...: four independent chains, combined in a fixed order
    const float* p = a.bias_src[s] + (size_t)ty * a.bias_stride[s] + col;
    const size_t step = (size_t)16 * a.bias_stride[s];
    const int rows = a.bias_rows[s];
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    int r = ty;
    for (; r + 48 < rows; r += 64, p += 4 * step) {
      c0 += p[0]; c1 += p[step]; c2 += p[2 * step]; c3 += p[3 * step];
    }
    for (; r < rows; r += 16, p += step) c0 += *p;
    acc = (c0 + c1) + (c2 + c3);
  }
  red[0][ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && col < a.bias_cols[s]) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[0][k][tx];
    a.bias_out[s][col] = t * inv;
  }
}

// partial[split][b][k] = sum over the split's rows n of dmod[b][n] W(n)[k];  W(n): row n of the concatenated adaLN weights
// (depth block linears of rows6 rows at a constant stride in the flat parameter buffer, then the final layer's)
constexpr int DC_ROWS = 1024;
__global__ void __launch_bounds__(256) adaln_dc_kernel(const float* __restrict__ dmod, int nmod, int B, const float* __restrict__ w_blocks,
                                                       long blk_stride, int depth, int rows6, const float* __restrict__ w_final, int D,
                                                       float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float sm[FIN_SB * DC_ROWS];   // dmod rows of the split, then the 16 x 8 x 64 reduction image
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int col = blockIdx.x * 64 + tx * 4;
  const int n0 = blockIdx.y * DC_ROWS, n1 = min(nmod, n0 + DC_ROWS);
  for (int bs = 0; bs < B; bs += FIN_SB) {
    const int nb = min(FIN_SB, B - bs);
    for (int i = threadIdx.x; i < nb * DC_ROWS; i += 256) {
      const int b = i / DC_ROWS, r = i - b * DC_ROWS;
      sm[i] = n0 + r < n1 ? dmod[(size_t)(bs + b) * nmod + n0 + r] : 0.f;
    }
    __syncthreads();
    float4 acc[FIN_SB];
#pragma unroll
    for (int b = 0; b < FIN_SB; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = n0 + ty; n < n1; n += 16) {
      const int blk = n / rows6;
      const float* wr = blk < depth ? w_blocks + (size_t)blk * blk_stride + (size_t)(n - blk * rows6) * D
                                    : w_final + (size_t)(n - depth * rows6) * D;
      const float4 w = *(const float4*)(wr + col);
#pragma unroll
      for (int b = 0; b < FIN_SB; ++b) {
        const float d = sm[b * DC_ROWS + (n - n0)];
        acc[b].x += d * w.x; acc[b].y += d * w.y; acc[b].z += d * w.z; acc[b].w += d * w.w;
      }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < FIN_SB; ++b) ((float4*)sm)[(ty * FIN_SB + b) * 16 + tx] = acc[b];
    __syncthreads();
    if (ty < nb) {   // thread row ty adds sample bs + ty over the 16 row lanes, fixed order
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float4 v = ((const float4*)sm)[(k * FIN_SB + ty) * 16 + tx];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *(float4*)(partial + ((size_t)blockIdx.y * B + bs + ty) * D + col) = s;
    }
    __syncthreads();
  }
}

// fp32 [N, K] master weights -> half [N, K] and half [K, N], every weight of every block in one launch (32 x 32 tiles): the
// blocks' four linears have the same shapes, so workgroup -> (block, linear, tile) is arithmetic on by-value tile offsets and
// one table read (descs[block * 4 + linear]: the pointers)
template <int DT>
__global__ void __launch_bounds__(256) pack_weights_kernel(const PackDesc* __restrict__ descs, PackPlan pl) {
  __shared__ float tile[32][33];
  const int blk = blockIdx.x / pl.tiles_per_block;
  int t = blockIdx.x - blk * pl.tiles_per_block;
  const int j = (t >= pl.tile0[1]) + (t >= pl.tile0[2]) + (t >= pl.tile0[3]);
  t -= pl.tile0[j];
  const PackDesc d = descs[blk * 4 + j];
  const int tk = (d.K + 31) >> 5;
  const int n0 = (t / tk) * 32, k0 = (t % tk) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    float v = 0.f;
    if (n0 + r < d.N && k0 + tx < d.K) {
      v = d.w[(size_t)(n0 + r) * d.K + k0 + tx];
      d.wn[(size_t)(n0 + r) * d.K + k0 + tx] = f2h_t<DT>(v);
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (k0 + r < d.K && n0 + tx < d.N) d.wt[(size_t)(k0 + r) * d.N + n0 + tx] = f2h_t<DT>(tile[tx][r]);
}

// ---------------------------------------------------------------------------------------------- narrow-operand products, exact fp32
// The final layer's linear (D -> P = p p C_out <= 32 columns, latte.py:196-207) and the patch embed (K = p p C_in <= 32, :233) have one
// operand with a handful of columns: no MFMA shape fits, and the strided fp32 GEMM ran them at a few percent of anything.
//   narrow_outer_kernel   partial[blk][p so_p + k so_k] = sum over the block's rows m of nar[m][p] wide[m][k]   (weight gradients;
//                         + the column sums of either operand = the bias gradients), thread = one float4 of wide columns x all P
//   narrow_dx_kernel      out[m][k] = half(sum_p nar[m][p] W[p][k])   (input gradient of the final linear), W column in registers
constexpr int NO_ROWS = 16;     // rows per LDS stage
constexpr int NO_PMAX = 32;
template <typename WT, int DT>
__global__ void __launch_bounds__(320) narrow_outer_kernel(const float* __restrict__ nar, int P, const WT* __restrict__ wide, int D, int M,
                                                           int rows_per_block, float* __restrict__ partial, long so_p, long so_k,
                                                           float* __restrict__ part_nsum, float* __restrict__ part_wsum) {
  extern __shared__ __attribute__((aligned(16))) float no_sm[];   // wide [NO_ROWS][D] | narrow [NO_ROWS][NO_PMAX]
  float* ws = no_sm;
  float* ns = no_sm + (size_t)NO_ROWS * D;
  const int tid = threadIdx.x, nq = D >> 2;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  const bool own = tid < nq;
  float4 acc[NO_PMAX];
#pragma unroll
  for (int p = 0; p < NO_PMAX; ++p) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 wsum = make_float4(0.f, 0.f, 0.f, 0.f);
  float nsum = 0.f;
  for (int r0 = m0; r0 < m1; r0 += NO_ROWS) {
    const int nr = min(NO_ROWS, m1 - r0);
    for (int i = tid; i < nr * nq; i += blockDim.x) {
      const int r = i / nq, q = i - r * nq;
      float4 v;
      if constexpr (sizeof(WT) == 4) v = *(const float4*)((const float*)wide + (size_t)(r0 + r) * D + q * 4);
      else {
        const uint2 h = *(const uint2*)((const unsigned short*)wide + (size_t)(r0 + r) * D + q * 4);
        if constexpr (DT == LATTE_DTYPE_BF16) {
          v = make_float4(__builtin_bit_cast(float, h.x << 16), __builtin_bit_cast(float, h.x & 0xffff0000u),
                          __builtin_bit_cast(float, h.y << 16), __builtin_bit_cast(float, h.y & 0xffff0000u));
        } else {
          v = make_float4((float)__builtin_bit_cast(_Float16, (unsigned short)(h.x & 0xffffu)), (float)__builtin_bit_cast(_Float16, (unsigned short)(h.x >> 16)),
                          (float)__builtin_bit_cast(_Float16, (unsigned short)(h.y & 0xffffu)), (float)__builtin_bit_cast(_Float16, (unsigned short)(h.y >> 16)));
        }
      }
      ((float4*)ws)[r * nq + q] = v;
    }
    for (int i = tid; i < nr * NO_PMAX; i += blockDim.x) {
      const int r = i / NO_PMAX, p = i - r * NO_PMAX;
      ns[i] = p < P ? nar[(size_t)(r0 + r) * P + p] : 0.f;
    }
    __syncthreads();
    if (own) {
      for (int r = 0; r < nr; ++r) {
        const float4 w = ((const float4*)ws)[r * nq + tid];
        wsum.x += w.x; wsum.y += w.y; wsum.z += w.z; wsum.w += w.w;
#pragma unroll
        for (int p4 = 0; p4 < NO_PMAX / 4; ++p4) {
          const float4 n = ((const float4*)ns)[r * (NO_PMAX / 4) + p4];
          acc[p4 * 4 + 0].x += n.x * w.x; acc[p4 * 4 + 0].y += n.x * w.y; acc[p4 * 4 + 0].z += n.x * w.z; acc[p4 * 4 + 0].w += n.x * w.w;
          acc[p4 * 4 + 1].x += n.y * w.x; acc[p4 * 4 + 1].y += n.y * w.y; acc[p4 * 4 + 1].z += n.y * w.z; acc[p4 * 4 + 1].w += n.y * w.w;
          acc[p4 * 4 + 2].x += n.z * w.x; acc[p4 * 4 + 2].y += n.z * w.y; acc[p4 * 4 + 2].z += n.z * w.z; acc[p4 * 4 + 2].w += n.z * w.w;
          acc[p4 * 4 + 3].x += n.w * w.x; acc[p4 * 4 + 3].y += n.w * w.y; acc[p4 * 4 + 3].z += n.w * w.z; acc[p4 * 4 + 3].w += n.w * w.w;
        }
      }
    }
    if (tid < NO_PMAX)
      for (int r = 0; r < nr; ++r) nsum += ns[r * NO_PMAX + tid];
    __syncthreads();
  }
  float* out = partial + (size_t)blockIdx.x * P * D;
  if (own) {
#pragma unroll
    for (int p = 0; p < NO_PMAX; ++p) {
      if (p < P) {
        float* o = out + p * so_p + (size_t)(tid * 4) * so_k;
        o[0] = acc[p].x; o[so_k] = acc[p].y; o[2 * so_k] = acc[p].z; o[3 * so_k] = acc[p].w;
      }
    }
    if (part_wsum) ((float4*)(part_wsum + (size_t)blockIdx.x * D))[tid] = wsum;
  }
  if (part_nsum && tid < P) part_nsum[(size_t)blockIdx.x * P + tid] = nsum;
}

template <int DT>
__global__ void __launch_bounds__(320) narrow_dx_kernel(const float* __restrict__ nar, int P, const float* __restrict__ W, int D, int M,
                                                        int rows_per_block, half_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float ns[NO_ROWS * NO_PMAX];
  const int tid = threadIdx.x, nq = D >> 2;
  const bool own = tid < nq;
  float4 w[NO_PMAX];
#pragma unroll
  for (int p = 0; p < NO_PMAX; ++p) w[p] = (own && p < P) ? ((const float4*)(W + (size_t)p * D))[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  for (int r0 = m0; r0 < m1; r0 += NO_ROWS) {
    const int nr = min(NO_ROWS, m1 - r0);
    for (int i = tid; i < nr * NO_PMAX; i += blockDim.x) {
      const int r = i / NO_PMAX, p = i - r * NO_PMAX;
      ns[i] = p < P ? nar[(size_t)(r0 + r) * P + p] : 0.f;
    }
    __syncthreads();
    if (own) {
      for (int r = 0; r < nr; ++r) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p4 = 0; p4 < NO_PMAX / 4; ++p4) {
          const float4 n = ((const float4*)ns)[r * (NO_PMAX / 4) + p4];
          a.x += n.x * w[p4 * 4].x; a.y += n.x * w[p4 * 4].y; a.z += n.x * w[p4 * 4].z; a.w += n.x * w[p4 * 4].w;
          a.x += n.y * w[p4 * 4 + 1].x; a.y += n.y * w[p4 * 4 + 1].y; a.z += n.y * w[p4 * 4 + 1].z; a.w += n.y * w[p4 * 4 + 1].w;
          a.x += n.z * w[p4 * 4 + 2].x; a.y += n.z * w[p4 * 4 + 2].y; a.z += n.z * w[p4 * 4 + 2].z; a.w += n.z * w[p4 * 4 + 2].w;
          a.x += n.w * w[p4 * 4 + 3].x; a.y += n.w * w[p4 * 4 + 3].y; a.z += n.w * w[p4 * 4 + 3].z; a.w += n.w * w[p4 * 4 + 3].w;
        }
        uint2 o;
        o.x = (unsigned int)f2h_t<DT>(a.x) | ((unsigned int)f2h_t<DT>(a.y) << 16);
        o.y = (unsigned int)f2h_t<DT>(a.z) | ((unsigned int)f2h_t<DT>(a.w) << 16);
        ((uint2*)(out + (size_t)(r0 + r) * D))[tid] = o;
      }
    }
    __syncthreads();
  }
}

}  // namespace

int launch_stage_finalize(const StageFinArgs& a, hipStream_t st) {
  if (a.D % 64 || a.D > FIN_DMAX || a.n_mod < 0 || a.n_mod > 6 || a.n_bias < 0 || a.n_bias > 4)
    return fail(LATTE_ERR_INVALID, "stage_finalize: need D % 64 == 0, D <= 1280, <= 6 modulation chunks, <= 4 bias sums");
  StageFinArgs b = a;
  b.bias_blk[0] = 0;
  for (int i = 0; i < a.n_bias; ++i) b.bias_blk[i + 1] = b.bias_blk[i] + (a.bias_cols[i] + 63) / 64;
  const int blocks = a.n_mod * (a.D / 64) + b.bias_blk[a.n_bias];
  if (blocks <= 0) return LATTE_OK;
  const int lds = (FIN_SB * 16 * 64 + FIN_SB * 64 + FIN_SB * a.D) * (int)sizeof(float);
  static std::atomic<uint64_t> done{0};
  // (the opt-in is remembered per device: ask for the largest image the kernel can need, not this launch's)
  constexpr int LDS_MAX = (FIN_SB * 16 * 64 + FIN_SB * 64 + FIN_SB * FIN_DMAX) * (int)sizeof(float);
  if (int rc = ensure_dynamic_lds((const void*)stage_finalize_kernel, LDS_MAX, done)) return rc;
  hipLaunchKernelGGL(stage_finalize_kernel, dim3(blocks), dim3(1024), lds, st, b);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

constexpr int NO_ROWS_PER_BLOCK = 128;
int narrow_blocks(int M) { return (M + NO_ROWS_PER_BLOCK - 1) / NO_ROWS_PER_BLOCK; }
// dW[p so_p + k so_k] = sum_m nar[m][p] wide[m][k] (assigned), optional bias gradients nsum_out[p] = sum_m nar[m][p],
// wsum_out[k] = sum_m wide[m][k];  wide: fp32 (wide_half == 0) or the engine's half type;  ws: float [narrow_blocks(M)][P D + P + D];
// inv_scale_dev: optional device loss scale (results x 1 / scale)
int launch_narrow_outer(const float* nar, int P, const void* wide, int wide_half, int D, int M, float* dW, long so_p, long so_k,
                        float* nsum_out, float* wsum_out, float* ws, int dtype, const float* inv_scale_dev, hipStream_t st) {
  if (P < 1 || P > NO_PMAX || D % 4 || D > FIN_DMAX) return fail(LATTE_ERR_INVALID, "narrow_outer: need 1 <= P <= 32, D % 4 == 0, D <= 1280");
  const int nb = narrow_blocks(M), threads = (D / 4 + 63) / 64 * 64;
  float* part = ws;
  float* pn = ws + (size_t)nb * P * D;
  float* pw = pn + (size_t)nb * P;
  const int lds = (NO_ROWS * D + NO_ROWS * NO_PMAX) * (int)sizeof(float);
  constexpr int lds_max = (NO_ROWS * FIN_DMAX + NO_ROWS * NO_PMAX) * (int)sizeof(float);   // the per-device opt-in covers every D
#define NO_CASE(WT, DT)                                                                                                          \
  {                                                                                                                              \
    static std::atomic<uint64_t> done{0};                                                                                        \
    if (int rc = ensure_dynamic_lds((const void*)narrow_outer_kernel<WT, DT>, lds_max, done)) return rc;                          \
    hipLaunchKernelGGL((narrow_outer_kernel<WT, DT>), dim3(nb), dim3(threads), lds, st, nar, P, (const WT*)wide, D, M,            \
                       NO_ROWS_PER_BLOCK, part, so_p, so_k, nsum_out ? pn : nullptr, wsum_out ? pw : nullptr);                   \
  }
  if (!wide_half) NO_CASE(float, LATTE_DTYPE_F16)
  else if (dtype == LATTE_DTYPE_BF16) NO_CASE(half_t, LATTE_DTYPE_BF16)
  else if (dtype == LATTE_DTYPE_F16) NO_CASE(half_t, LATTE_DTYPE_F16)
  else return fail(LATTE_ERR_INVALID, "narrow_outer: unknown dtype");
#undef NO_CASE
  LATTE_HIP(hipGetLastError());
  int rc;
  if ((rc = launch_split_reduce(part, nb, (size_t)P * D, (size_t)P * D, dW, 0, st, inv_scale_dev))) return rc;
  if (nsum_out && (rc = launch_split_reduce(pn, nb, (size_t)P, (size_t)P, nsum_out, 0, st, inv_scale_dev))) return rc;
  if (wsum_out && (rc = launch_split_reduce(pw, nb, (size_t)D, (size_t)D, wsum_out, 0, st, inv_scale_dev))) return rc;
  return LATTE_OK;
}
// out[m][k] = half(sum_p nar[m][p] W[p][k])
int launch_narrow_dx(const float* nar, int P, const float* W, int D, int M, half_t* out, int dtype, hipStream_t st) {
  if (P < 1 || P > NO_PMAX || D % 4 || D > FIN_DMAX) return fail(LATTE_ERR_INVALID, "narrow_dx: need 1 <= P <= 32, D % 4 == 0, D <= 1280");
  const int threads = (D / 4 + 63) / 64 * 64;
  if (dtype == LATTE_DTYPE_BF16)
    hipLaunchKernelGGL(narrow_dx_kernel<LATTE_DTYPE_BF16>, dim3(narrow_blocks(M)), dim3(threads), 0, st, nar, P, W, D, M, NO_ROWS_PER_BLOCK, out);
  else if (dtype == LATTE_DTYPE_F16)
    hipLaunchKernelGGL(narrow_dx_kernel<LATTE_DTYPE_F16>, dim3(narrow_blocks(M)), dim3(threads), 0, st, nar, P, W, D, M, NO_ROWS_PER_BLOCK, out);
  else return fail(LATTE_ERR_INVALID, "narrow_dx: unknown dtype");
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int adaln_dc_splits(int nmod) { return (nmod + DC_ROWS - 1) / DC_ROWS; }
// dc[b][k] = sum_n dmod[b][n] W(n)[k] (assigned);  ws: float [adaln_dc_splits(nmod)][B][D]
int launch_adaln_dc(const float* dmod, int nmod, int B, const float* w_blocks, long blk_stride, int depth, int rows6, const float* w_final,
                    int D, float* ws, float* dc, hipStream_t st) {
  if (D % 64) return fail(LATTE_ERR_INVALID, "adaln_dc: D % 64 != 0");
  const int splits = adaln_dc_splits(nmod);
  hipLaunchKernelGGL(adaln_dc_kernel, dim3(D / 64, splits), dim3(256), 0, st, dmod, nmod, B, w_blocks, blk_stride, depth, rows6, w_final, D, ws);
  LATTE_HIP(hipGetLastError());
  return launch_split_reduce(ws, splits, (size_t)B * D, (size_t)B * D, dc, 0, st);
}

// descs_dev: PackDesc [blocks][4];  pl: tiles of the four linears of one block
int launch_pack_weights(const PackDesc* descs_dev, int blocks, const PackPlan& pl, int dtype, hipStream_t st) {
  if (blocks <= 0 || pl.tiles_per_block <= 0) return LATTE_OK;
  const dim3 grid((unsigned)blocks * pl.tiles_per_block);
  if (dtype == LATTE_DTYPE_BF16) hipLaunchKernelGGL(pack_weights_kernel<LATTE_DTYPE_BF16>, grid, dim3(256), 0, st, descs_dev, pl);
  else if (dtype == LATTE_DTYPE_F16) hipLaunchKernelGGL(pack_weights_kernel<LATTE_DTYPE_F16>, grid, dim3(256), 0, st, descs_dev, pl);
  else return fail(LATTE_ERR_INVALID, "pack_weights: unknown dtype");
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

}  // namespace latte
