// Kernels of the training step (round 2): everything of forward-with-saved-activations,
// backward and the optimiser that is NOT a big GEMM or the attention backward (train_attn.hip).
//
//   loss gradient      d mean(terms["loss"]) / d model_output                      gaussian_diffusion.py:719-795
//   gated residual     x' = x + gate * y  (training forward keeps y)               latte.py:179-180
//   its backward       dy = gate * dx (half), dgate[s] = sum_rows dx * y
//   LN + modulate bwd  dx, dshift[s], dscale[s]                                    latte.py:28-29,166,168
//   GELU(tanh)         h = gelu(u);  du = dh * gelu'(u)                            latte.py:170
//   column sums        bias gradients  db[n] += sum_m dy[m, n]
//   split reduce       dW (+)= sum of the split-K partial products
//   small GEMMs        strided fp32 (adaLN / embedder / final-layer linears with <= 64 rows or <= 32 columns)
//   unpatchify^-1, im2col of the patch embed, label-table scatter, SiLU backward
//   AdamW + EMA        torch.optim.AdamW(lr, weight_decay = 0) + update_ema        train.py:127,233-236, utils.py:191-200
//   grad norm / clip   clip_grad_norm_                                              utils.py:72-117
//
// Row-wise kernels use one wave per token row (lane owns the float4 chunks {lane + 64 c} of the row, as ln_modulate_kernel);
// per-sample column reductions (dshift / dscale / dgate) are two-stage and deterministic: a wave walks a run of consecutive
// rows of ONE sample and writes one partial row, a finalize kernel adds the partial rows in a fixed order.
#include <cmath>

#include "common.h"

namespace latte {
namespace {

__device__ __forceinline__ float wave_sum_t(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int DT>
__device__ __forceinline__ unsigned int pack2t(float lo, float hi) {
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
__device__ __forceinline__ float h2f(unsigned short h) {
  if constexpr (DT == LATTE_DTYPE_BF16) return __builtin_bit_cast(float, (unsigned int)h << 16);
  else return (float)__builtin_bit_cast(_Float16, h);
}
template <int DT>
__device__ __forceinline__ void unpack4(const uint2 p, float& a, float& b, float& c, float& d) {
  a = h2f<DT>((unsigned short)(p.x & 0xffffu)); b = h2f<DT>((unsigned short)(p.x >> 16));
  c = h2f<DT>((unsigned short)(p.y & 0xffffu)); d = h2f<DT>((unsigned short)(p.y >> 16));
}

constexpr int NQ_MAX = 5;   // float4 chunk groups per lane: D <= 1280

// ---------------------------------------------------------------------------------------------- gated residual, forward
template <int DT>
__global__ void __launch_bounds__(256) gated_add_kernel(const float* __restrict__ x_in, const half_t* __restrict__ y,
                                                        const float* __restrict__ gate, int gate_stride, float* __restrict__ x_out,
                                                        int M, int D, int rps) {
  const int nt = D >> 2;
  const size_t total = (size_t)M * nt;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / nt), c = (int)(i % nt);
    const float4 xv = ((const float4*)x_in)[i];
    const float4 g = ((const float4*)(gate + (size_t)(row / rps) * gate_stride))[c];
    float a, b, cc, d;
    unpack4<DT>(((const uint2*)y)[i], a, b, cc, d);
    ((float4*)x_out)[i] = make_float4(xv.x + g.x * a, xv.y + g.y * b, xv.z + g.z * cc, xv.w + g.w * d);
  }
}

// ---------------------------------------------------------------------------------------------- gated residual + the next LayerNorm
// x' = x + gate * y (+ temp_embed row, latte.py:355-358) written once, and xn = LN(x') (1 + scale) + shift of the NEXT LayerNorm in
// the same pass over the row (round 6b): the separate pair moved 16 bytes per element, this 12, in one launch.  One wave per row,
// the arithmetic of gated_add_kernel followed by ln_modulate_kernel (pointwise.hip) expression for expression; equal to the pair up
// to the compiler's FMA contraction choices inside the fp32 statistics (measured: model output within 9e-6 absolute,
// tests/test_training_step.py::test_small_kernel_consolidation_matches_the_separate_launches).
template <int DT, bool ADD_TE>
__global__ void __launch_bounds__(256) gated_add_ln_kernel(const float* __restrict__ x_in, const half_t* __restrict__ y,
                                                           const float* __restrict__ gate, int gate_stride, float* __restrict__ x_out,
                                                           half_t* __restrict__ xn, const float* __restrict__ shift,
                                                           const float* __restrict__ scale, int mod_stride, int M, int D, int rps,
                                                           const float* __restrict__ te, int T, int F) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nt = D >> 2;
  const int smp = row / rps;
  const size_t ro = (size_t)row * nt;
  const float4* g4 = (const float4*)(gate + (size_t)smp * gate_stride);
  const float4* sh = (const float4*)(shift + (size_t)smp * mod_stride);
  const float4* sc = (const float4*)(scale + (size_t)smp * mod_stride);
  float4 v[NQ_MAX], sha[NQ_MAX], sca[NQ_MAX];
#pragma unroll
  for (int c = 0; c < NQ_MAX; ++c) {
    const int ch = c * 64 + lane;
    v[c] = sha[c] = sca[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ch < nt) {
      const float4 xv = ((const float4*)x_in)[ro + ch];
      const float4 g = g4[ch];
      float a, b, cc, d;
      unpack4<DT>(((const uint2*)y)[ro + ch], a, b, cc, d);
      sha[c] = sh[ch];
      sca[c] = sc[ch];
      v[c] = make_float4(xv.x + g.x * a, xv.y + g.y * b, xv.z + g.z * cc, xv.w + g.w * d);
      if constexpr (ADD_TE) {
        const float4 e = ((const float4*)(te + (size_t)((row / T) % F) * D))[ch];
        v[c].x += e.x; v[c].y += e.y; v[c].z += e.z; v[c].w += e.w;
      }
      ((float4*)x_out)[ro + ch] = v[c];
    }
  }
  const float invD = 1.0f / (float)D;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NQ_MAX; ++c) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);     // absent chunks hold zeros
  const float mean = wave_sum_t(s) * invD;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NQ_MAX; ++c) {
    if (c * 64 + lane < nt) {
      const float a = v[c].x - mean, b = v[c].y - mean, d = v[c].z - mean, e = v[c].w - mean;
      q += (a * a + b * b) + (d * d + e * e);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum_t(q) * invD + 1e-6f);
#pragma unroll
  for (int c = 0; c < NQ_MAX; ++c) {
    const int ch = c * 64 + lane;
    if (ch < nt) {
      const float4 a = sha[c], b = sca[c];
      const float o0 = (v[c].x - mean) * rstd * (1.0f + b.x) + a.x;
      const float o1 = (v[c].y - mean) * rstd * (1.0f + b.y) + a.y;
      const float o2 = (v[c].z - mean) * rstd * (1.0f + b.z) + a.z;
      const float o3 = (v[c].w - mean) * rstd * (1.0f + b.w) + a.w;
      uint2 o;
      o.x = pack2t<DT>(o0, o1);
      o.y = pack2t<DT>(o2, o3);
      ((uint2*)xn)[ro + ch] = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------- gated residual, backward
// one wave per run of R consecutive rows of one sample: dy = gate * dx (half), partial[run][col] = sum_rows dx * y
template <int DT, bool BIAS>
__global__ void __launch_bounds__(256) gate_bwd_kernel(const float* __restrict__ dx, const half_t* __restrict__ y,
                                                       const float* __restrict__ gate, int gate_stride, half_t* __restrict__ dy,
                                                       float* __restrict__ partial, int M, int D, int rps, int R) {
  constexpr int NS = BIAS ? 2 : 1;   // partial rows per block: sum dx * y [, sum gate * dx = the output linear's bias gradient]
  const int lane = threadIdx.x & 63;
  const int run = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int row0 = run * R;          // (launcher: M / R is a multiple of 4, so no wave of a block is out of range)
  const int nt = D >> 2;
  const float4* g4 = (const float4*)(gate + (size_t)(row0 / rps) * gate_stride);
  float4 g[NQ_MAX], acc[NQ_MAX], accb[BIAS ? NQ_MAX : 1];
#pragma unroll
  for (int c = 0; c < NQ_MAX; ++c) {
    const int ch = c * 64 + lane;
    g[c] = ch < nt ? g4[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
    acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (BIAS) accb[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int r = 0; r < R; ++r) {
    const size_t ro = (size_t)(row0 + r) * nt;
#pragma unroll
    for (int c = 0; c < NQ_MAX; ++c) {
      const int ch = c * 64 + lane;
      if (ch < nt) {
        const float4 d = ((const float4*)dx)[ro + ch];
        float a, b, cc, e;
        unpack4<DT>(((const uint2*)y)[ro + ch], a, b, cc, e);
        acc[c].x += d.x * a; acc[c].y += d.y * b; acc[c].z += d.z * cc; acc[c].w += d.w * e;
        const float4 gd = make_float4(g[c].x * d.x, g[c].y * d.y, g[c].z * d.z, g[c].w * d.w);
        if constexpr (BIAS) { accb[c].x += gd.x; accb[c].y += gd.y; accb[c].z += gd.z; accb[c].w += gd.w; }
        uint2 o;
        o.x = pack2t<DT>(gd.x, gd.y);
        o.y = pack2t<DT>(gd.z, gd.w);
        ((uint2*)dy)[ro + ch] = o;
      }
    }
  }
  // the block's 4 waves (4 consecutive runs of one sample) are added through LDS: NS partial rows per block
  extern __shared__ __attribute__((aligned(16))) float red_t[];   // [3 waves][NS][D]
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < NQ_MAX; ++c) {
    const int ch = c * 64 + lane;
    if (ch < nt && wv > 0) {
      ((float4*)(red_t + ((size_t)(wv - 1) * NS + 0) * D))[ch] = acc[c];
      if constexpr (BIAS) ((float4*)(red_t + ((size_t)(wv - 1) * NS + 1) * D))[ch] = accb[c];
    }
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int c = 0; c < NQ_MAX; ++c) {
      const int ch = c * 64 + lane;
      if (ch < nt) {
        float4 a = acc[c];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const float4 o = ((const float4*)(red_t + ((size_t)w * NS + 0) * D))[ch];
          a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        ((float4*)(partial + ((size_t)blockIdx.x * NS + 0) * D))[ch] = a;
        if constexpr (BIAS) {
          float4 b = accb[c];
#pragma unroll
          for (int w = 0; w < 3; ++w) {
            const float4 o = ((const float4*)(red_t + ((size_t)w * NS + 1) * D))[ch];
            b.x += o.x; b.y += o.y; b.z += o.z; b.w += o.w;
          }
          ((float4*)(partial + ((size_t)blockIdx.x * NS + 1) * D))[ch] = b;
        }
      }
    }
  }
}


// out[s * out_stride + col] = sum over the runs of sample s of partial[run][which][col]   (nsum partial rows per run).
// One block per (sample, 64 columns): 4 thread rows walk the runs with stride 4, fixed-order LDS reduction (deterministic).
__global__ void __launch_bounds__(256) sample_colsum_finalize_kernel(const float* __restrict__ partial, int runs_per_sample, int nsum,
                                                                     int which, int D, float* __restrict__ out, int out_stride) {
  __shared__ float red[4][64];
  const int s = blockIdx.y;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  float a = 0.f;
  if (col < D)
    for (int r = ty; r < runs_per_sample; r += 4) a += partial[(((size_t)s * runs_per_sample + r) * nsum + which) * D + col];
  red[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && col < D) out[(size_t)s * out_stride + col] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

// ---------------------------------------------------------------------------------------------- LN + modulate, backward
// y = LN(x) (1 + scale) + shift.  Given dy (half):  dshift = sum dy, dscale = sum dy * xhat (per sample),
// dxhat = dy (1 + scale),  dx = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)),  dx_out = dx_in + dx.
// GATE (round 6b): the gated residual's backward of the branch BELOW this LayerNorm rides on the same pass -- the fresh dx row is
// still in registers when gate_bwd_kernel would read it back: dy2 = gate2 * dx (half), gpartial[block][0] = sum dx * y2,
// gpartial[block][1] = sum gate2 * dx (the branch's output-linear bias gradient).  NQ: float4 chunk groups per lane (3: D <= 768).
template <int DT, int NQ, bool GATE>
__global__ void __launch_bounds__(256, NQ <= 3 ? 3 : 2) ln_bwd_kernel(const half_t* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ scale, int mod_stride, const float* dx_in,
                                                     float* dx_out, float* __restrict__ partial, int M, int D, int rps, int R,
                                                     const half_t* __restrict__ y2, const float* __restrict__ gate2, int gate2_stride,
                                                     half_t* __restrict__ dy2, float* __restrict__ gpartial) {
  const int lane = threadIdx.x & 63;
  const int run = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int row0 = run * R;          // (launcher: M / R is a multiple of 4)
  const int nt = D >> 2;
  const float invD = 1.0f / (float)D;
  const float4* sc4 = (const float4*)(scale + (size_t)(row0 / rps) * mod_stride);
  float4 sc[NQ], a_sh[NQ], a_sc[NQ];
  float4 g2[GATE ? NQ : 1], a_g[GATE ? NQ : 1], a_b[GATE ? NQ : 1];
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    const int ch = c * 64 + lane;
    sc[c] = ch < nt ? sc4[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
    a_sh[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    a_sc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (GATE) {
      g2[c] = ch < nt ? ((const float4*)(gate2 + (size_t)(row0 / rps) * gate2_stride))[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
      a_g[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      a_b[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // the next row's x and dy are requested before this row's three dependent wave reductions (round 6b: a block holds ~2.5 waves per
  // SIMD at the training shapes, so the row chain's load latency was exposed: 55 -> see profiles/r6b_*)
  float4 nx[NQ];
  uint2 nd[NQ];
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    const int ch = c * 64 + lane;
    nx[c] = ch < nt ? ((const float4*)x)[(size_t)row0 * nt + ch] : make_float4(0.f, 0.f, 0.f, 0.f);
    nd[c] = ch < nt ? ((const uint2*)dy)[(size_t)row0 * nt + ch] : make_uint2(0u, 0u);
  }
  for (int r = 0; r < R; ++r) {
    const size_t ro = (size_t)(row0 + r) * nt;
    float4 v[NQ], g[NQ];
    uint2 dyr[NQ];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      v[c] = nx[c];
      dyr[c] = nd[c];
      s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
    }
    if (r + 1 < R) {
#pragma unroll
      for (int c = 0; c < NQ; ++c) {
        const int ch = c * 64 + lane;
        if (ch < nt) {
          nx[c] = ((const float4*)x)[ro + nt + ch];
          nd[c] = ((const uint2*)dy)[ro + nt + ch];
        }
      }
    }
    const float mean = wave_sum_t(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      if (c * 64 + lane < nt) {
        v[c].x -= mean; v[c].y -= mean; v[c].z -= mean; v[c].w -= mean;
        q += (v[c].x * v[c].x + v[c].y * v[c].y) + (v[c].z * v[c].z + v[c].w * v[c].w);
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum_t(q) * invD + 1e-6f);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int ch = c * 64 + lane;
      g[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ch < nt) {
        v[c].x *= rstd; v[c].y *= rstd; v[c].z *= rstd; v[c].w *= rstd;       // xhat
        float a, b, cc, e;
        unpack4<DT>(dyr[c], a, b, cc, e);
        a_sh[c].x += a; a_sh[c].y += b; a_sh[c].z += cc; a_sh[c].w += e;
        a_sc[c].x += a * v[c].x; a_sc[c].y += b * v[c].y; a_sc[c].z += cc * v[c].z; a_sc[c].w += e * v[c].w;
        g[c] = make_float4(a * (1.0f + sc[c].x), b * (1.0f + sc[c].y), cc * (1.0f + sc[c].z), e * (1.0f + sc[c].w));   // dxhat
        m1 += (g[c].x + g[c].y) + (g[c].z + g[c].w);
        m2 += (g[c].x * v[c].x + g[c].y * v[c].y) + (g[c].z * v[c].z + g[c].w * v[c].w);
      }
    }
    m1 = wave_sum_t(m1) * invD;
    m2 = wave_sum_t(m2) * invD;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int ch = c * 64 + lane;
      if (ch < nt) {
        float4 o = make_float4(rstd * (g[c].x - m1 - v[c].x * m2), rstd * (g[c].y - m1 - v[c].y * m2),
                               rstd * (g[c].z - m1 - v[c].z * m2), rstd * (g[c].w - m1 - v[c].w * m2));
        if (dx_in) {
          const float4 p = ((const float4*)dx_in)[ro + ch];
          o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        ((float4*)dx_out)[ro + ch] = o;
        if constexpr (GATE) {   // gate_bwd_kernel<DT, true>'s row body on the row just written
          float a, b, cc, e;
          unpack4<DT>(((const uint2*)y2)[ro + ch], a, b, cc, e);
          a_g[c].x += o.x * a; a_g[c].y += o.y * b; a_g[c].z += o.z * cc; a_g[c].w += o.w * e;
          const float4 gd = make_float4(g2[c].x * o.x, g2[c].y * o.y, g2[c].z * o.z, g2[c].w * o.w);
          a_b[c].x += gd.x; a_b[c].y += gd.y; a_b[c].z += gd.z; a_b[c].w += gd.w;
          uint2 w;
          w.x = pack2t<DT>(gd.x, gd.y);
          w.y = pack2t<DT>(gd.z, gd.w);
          ((uint2*)dy2)[ro + ch] = w;
        }
      }
    }
  }
  extern __shared__ __attribute__((aligned(16))) float red_t[];   // [3 waves][2 or 4][D]
  constexpr int NS = GATE ? 4 : 2;
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    const int ch = c * 64 + lane;
    if (ch < nt && wv > 0) {
      ((float4*)(red_t + ((size_t)(wv - 1) * NS + 0) * D))[ch] = a_sh[c];
      ((float4*)(red_t + ((size_t)(wv - 1) * NS + 1) * D))[ch] = a_sc[c];
      if constexpr (GATE) {
        ((float4*)(red_t + ((size_t)(wv - 1) * NS + 2) * D))[ch] = a_g[c];
        ((float4*)(red_t + ((size_t)(wv - 1) * NS + 3) * D))[ch] = a_b[c];
      }
    }
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int ch = c * 64 + lane;
      if (ch < nt) {
        float4 a = a_sh[c], b = a_sc[c];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const float4 o = ((const float4*)(red_t + ((size_t)w * NS + 0) * D))[ch];
          const float4 q = ((const float4*)(red_t + ((size_t)w * NS + 1) * D))[ch];
          a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
          b.x += q.x; b.y += q.y; b.z += q.z; b.w += q.w;
        }
        ((float4*)(partial + ((size_t)blockIdx.x * 2 + 0) * D))[ch] = a;
        ((float4*)(partial + ((size_t)blockIdx.x * 2 + 1) * D))[ch] = b;
        if constexpr (GATE) {
          float4 e = a_g[c], f = a_b[c];
#pragma unroll
          for (int w = 0; w < 3; ++w) {
            const float4 o = ((const float4*)(red_t + ((size_t)w * NS + 2) * D))[ch];
            const float4 q = ((const float4*)(red_t + ((size_t)w * NS + 3) * D))[ch];
            e.x += o.x; e.y += o.y; e.z += o.z; e.w += o.w;
            f.x += q.x; f.y += q.y; f.z += q.z; f.w += q.w;
          }
          ((float4*)(gpartial + ((size_t)blockIdx.x * 2 + 0) * D))[ch] = e;
          ((float4*)(gpartial + ((size_t)blockIdx.x * 2 + 1) * D))[ch] = f;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- GELU(tanh)
// gelu(x) = x s(x), s = sigmoid(2u), u = sqrt(2/pi)(x + 0.044715 x^3);  gelu'(x) = s + x s (1 - s) 2 u',  u' = sqrt(2/pi)(1 + 3*0.044715 x^2)
__device__ __forceinline__ float gelu_sig(float x) {
  const float p = __builtin_fmaf(x * x, -0.10294324f, -2.3022082f);     // -2 log2(e) sqrt(2/pi) (1 + 0.044715 x^2)
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(p * x));
}
template <int DT, bool BWD>
__global__ void __launch_bounds__(256) gelu_kernel(const half_t* __restrict__ u, const half_t* __restrict__ dh, half_t* __restrict__ out,
                                                   size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float x[4];
    unpack4<DT>(((const uint2*)u)[i], x[0], x[1], x[2], x[3]);
    float o[4];
    if constexpr (BWD) {
      float d[4];
      unpack4<DT>(((const uint2*)dh)[i], d[0], d[1], d[2], d[3]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float s = gelu_sig(x[k]);
        const float du = 0.7978845608f * (1.0f + 0.134145f * x[k] * x[k]);
        o[k] = d[k] * (s + x[k] * s * (1.0f - s) * 2.0f * du);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = x[k] * gelu_sig(x[k]);
    }
    uint2 p;
    p.x = pack2t<DT>(o[0], o[1]);
    p.y = pack2t<DT>(o[2], o[3]);
    ((uint2*)out)[i] = p;
  }
}

// ---------------------------------------------------------------------------------------------- column sums of a half matrix
// partial[chunk][col] = sum over the chunk's rows of in[row][col]; tile = 128 columns x CS_ROWS rows per block
constexpr int CS_ROWS = 512;
template <int DT>
__global__ void __launch_bounds__(256) colsum_half_kernel(const half_t* __restrict__ in, int M, int C, float* __restrict__ partial) {
  __shared__ float red[16][128];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int col0 = blockIdx.x * 128 + tx * 8;
  const int r0 = blockIdx.y * CS_ROWS;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col0 < C) {
    const int rend = min(r0 + CS_ROWS, M);
#pragma unroll 8
    for (int r = r0 + ty; r < rend; r += 16) {
      const uint4 p = *(const uint4*)(in + (size_t)r * C + col0);
      float v[8];
      unpack4<DT>(make_uint2(p.x, p.y), v[0], v[1], v[2], v[3]);
      unpack4<DT>(make_uint2(p.z, p.w), v[4], v[5], v[6], v[7]);
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[ty][tx * 8 + k] = a[k];
  __syncthreads();
  if (threadIdx.x < 128) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][threadIdx.x];
    const int col = blockIdx.x * 128 + threadIdx.x;
    if (col < C) partial[(size_t)blockIdx.y * C + col] = s;
  }
}
// out[i] (+)= sum_s partial[s * stride + i]
// (inv_scale: optional device loss scale; the SUM -- not the accumulated-into value -- leaves the scaled domain: x 1 / scale, exact)
__global__ void split_reduce_kernel(const float* __restrict__ partial, int splits, size_t stride, size_t n, float* __restrict__ out,
                                    int accumulate, const float* __restrict__ inv_scale) {
  const float f = inv_scale ? 1.0f / inv_scale[0] : 1.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float a = (accumulate && !inv_scale) ? out[i] : 0.f;
    for (int s = 0; s < splits; ++s) a += partial[(size_t)s * stride + i];
    if (inv_scale) a = accumulate ? out[i] + a * f : a * f;
    out[i] = a;
  }
}
// the same sum in the same (slab) order on 16-byte accesses, four slabs in flight per step (n, stride multiples of 4, 16-byte
// aligned buffers: every weight matrix): the reductions of the weight-gradient partial products are bandwidth-bound streams
__global__ void __launch_bounds__(256) split_reduce4_kernel(const float4* __restrict__ partial, int splits, size_t stride4, size_t n4,
                                                            float4* __restrict__ out, int accumulate,
                                                            const float* __restrict__ inv_scale) {
  const float f = inv_scale ? 1.0f / inv_scale[0] : 1.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 a = (accumulate && !inv_scale) ? out[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 4 <= splits; s += 4) {
      const float4 p0 = partial[(size_t)s * stride4 + i], p1 = partial[(size_t)(s + 1) * stride4 + i];
      const float4 p2 = partial[(size_t)(s + 2) * stride4 + i], p3 = partial[(size_t)(s + 3) * stride4 + i];
      a.x = (((a.x + p0.x) + p1.x) + p2.x) + p3.x; a.y = (((a.y + p0.y) + p1.y) + p2.y) + p3.y;
      a.z = (((a.z + p0.z) + p1.z) + p2.z) + p3.z; a.w = (((a.w + p0.w) + p1.w) + p2.w) + p3.w;
    }
    for (; s < splits; ++s) {
      const float4 p = partial[(size_t)s * stride4 + i];
      a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    }
    if (inv_scale) {   // the sum leaves the loss-scaled domain (power of two: exact)
      a.x *= f; a.y *= f; a.z *= f; a.w *= f;
      if (accumulate) {
        const float4 o = out[i];
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
      }
    }
    out[i] = a;
  }
}

// ---------------------------------------------------------------------------------------------- weight packing
// fp32 [N, K] master weight -> half [N, K] and half [K, N] (the operand of the input-gradient GEMM dX = dY W)
template <int DT>
__global__ void __launch_bounds__(256) pack_weight_kernel(const float* __restrict__ w, half_t* __restrict__ wn, half_t* __restrict__ wt,
                                                          int N, int K) {
  __shared__ float tile[32][33];
  const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    float v = 0.f;
    if (n0 + r < N && k0 + tx < K) v = w[(size_t)(n0 + r) * K + k0 + tx];
    tile[r][tx] = v;
    if (wn && n0 + r < N && k0 + tx < K) {
      const unsigned int p = pack2t<DT>(v, 0.f);
      wn[(size_t)(n0 + r) * K + k0 + tx] = (half_t)(p & 0xffffu);
    }
  }
  __syncthreads();
  if (wt) {
    for (int r = ty; r < 32; r += 8) {
      if (k0 + r < K && n0 + tx < N) {
        const unsigned int p = pack2t<DT>(tile[tx][r], 0.f);
        wt[(size_t)(k0 + r) * N + n0 + tx] = (half_t)(p & 0xffffu);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- strided fp32 GEMM (small shapes)
// C[m scm + n scn] (+)= alpha * sum_k A[m sam + k sak] * B[k sbk + n sbn];  threads run along n.  k_chunk > 0: blockIdx.z owns
// the K range [z k_chunk, (z + 1) k_chunk) and ASSIGNS its partial product to C + z * c_split_stride (reduce afterwards).
__global__ void __launch_bounds__(256) naive_gemm_kernel(const float* __restrict__ A, long sam, long sak, const float* __restrict__ B,
                                                         long sbk, long sbn, float* __restrict__ C, long scm, long scn, int M, int N,
                                                         int K, float alpha, int accumulate, int k_chunk, long c_split_stride) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (n >= N || m >= M) return;
  int k0 = 0, k1 = K;
  if (k_chunk > 0) {
    k0 = blockIdx.z * k_chunk;
    k1 = min(K, k0 + k_chunk);
    C += (size_t)blockIdx.z * c_split_stride;
  }
  const float* a = A + (size_t)m * sam;
  const float* b = B + (size_t)n * sbn;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = k0;
  for (; k + 4 <= k1; k += 4) {
    s0 += a[(size_t)k * sak] * b[(size_t)k * sbk];
    s1 += a[(size_t)(k + 1) * sak] * b[(size_t)(k + 1) * sbk];
    s2 += a[(size_t)(k + 2) * sak] * b[(size_t)(k + 2) * sbk];
    s3 += a[(size_t)(k + 3) * sak] * b[(size_t)(k + 3) * sbk];
  }
  for (; k < k1; ++k) s0 += a[(size_t)k * sak] * b[(size_t)k * sbk];
  const float r = alpha * ((s0 + s1) + (s2 + s3));
  float* c = C + (size_t)m * scm + (size_t)n * scn;
  *c = (accumulate && k_chunk == 0) ? *c + r : r;
}

__global__ void tfreq_kernel(const int64_t* __restrict__ t, float* __restrict__ out, int B) {
  // timestep_embedding (latte.py:96-117): [cos(t f_i), sin(t f_i)], f_i = exp(-ln(10000) i / 128), 256 columns
  const int b = blockIdx.x, i = threadIdx.x;   // 128 threads
  const float f = expf(-9.210340371976184f * (float)i / 128.0f);
  const float a = (float)t[b] * f;
  out[(size_t)b * 256 + i] = cosf(a);
  out[(size_t)b * 256 + 128 + i] = sinf(a);
}
__global__ void gather_i64_kernel(const int64_t* __restrict__ table, const int64_t* __restrict__ idx, int64_t* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = table[idx[i]];
}

// ---------------------------------------------------------------------------------------------- token <-> latent layout helpers
// dtok[m][(p q c)] = dout[bf][c][gh p_ + p][gw p_ + q]   (inverse of unpatchify, latte.py:297-310), m = bf * T + gh * G + gw
__global__ void unpatchify_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dtok, int BF, int G, int p, int Cout) {
  const int P = p * p * Cout, H = G * p;
  const size_t total = (size_t)BF * G * G * P;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int j = (int)(i % P);
    const size_t m = i / P;
    const int gw = (int)(m % G), gh = (int)((m / G) % G);
    const size_t bf = m / ((size_t)G * G);
    const int c = j % Cout, q = (j / Cout) % p, pp = j / (Cout * p);
    dtok[i] = dout[((bf * Cout + c) * H + gh * p + pp) * H + gw * p + q];
  }
}
// pix[m][(c p q)] = x[bf][c][gh p_ + p][gw p_ + q]  (the Conv2d(k = s = p) patch of token m, x_embedder.proj.weight order)
__global__ void im2col_patch_kernel(const float* __restrict__ x, float* __restrict__ pix, int BF, int G, int p, int C) {
  const int P = C * p * p, H = G * p;
  const size_t total = (size_t)BF * G * G * P;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int j = (int)(i % P);
    const size_t m = i / P;
    const int gw = (int)(m % G), gh = (int)((m / G) % G);
    const size_t bf = m / ((size_t)G * G);
    const int q = j % p, pp = (j / p) % p, c = j / (p * p);
    pix[i] = x[((bf * C + c) * H + gh * p + pp) * H + gw * p + q];
  }
}
// dtable[idx[b]][:] += dc[b][:]  (serial over b: a label may repeat)
__global__ void embedding_bwd_kernel(const float* __restrict__ dc, const int64_t* __restrict__ idx, float* __restrict__ dtable, int B,
                                     int D) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= D) return;
  for (int b = 0; b < B; ++b) dtable[(size_t)idx[b] * D + col] += dc[(size_t)b * D + col];
}
// din[i] = dout[i] * silu'(pre[i]),  silu'(x) = s (1 + x (1 - s)),  s = sigmoid(x)
__global__ void silu_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ pre, float* __restrict__ din, size_t n,
                                int accumulate) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float x = pre[i];
    const float s = 1.0f / (1.0f + __expf(-x));
    const float v = dout[i] * (s * (1.0f + x * (1.0f - s)));
    din[i] = accumulate ? din[i] + v : v;
  }
}
__global__ void add_rows_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] += src[i];
}
// out[col] (+)= sum_b in[b * stride + col]
__global__ void rows_sum_kernel(const float* __restrict__ in, int B, long stride, int N, float* __restrict__ out, int accumulate) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= N) return;
  float a = accumulate ? out[col] : 0.f;
  for (int b = 0; b < B; ++b) a += in[(size_t)b * stride + col];
  out[col] = a;
}

// ---------------------------------------------------------------------------------------------- loss gradient
// d/d model_output of  mean_b( mse_b + vb_scale * vb_b ),  mse_b = mean_flat((target - eps)^2),
// vb_b = mean_flat(t == 0 ? decoder NLL : KL) / ln 2 evaluated on cat([eps.detach(), v])  (gaussian_diffusion.py:743-787):
// the eps channels get the MSE gradient only, the variance channels the bound's.
__device__ __forceinline__ float cdf_approx(float x) {
  return 0.5f * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float cdf_approx_grad(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  const float th = tanhf(u);
  return 0.5f * (1.0f - th * th) * 0.7978845608028654f * (1.0f + 0.134145f * x * x);
}
__global__ void __launch_bounds__(256) loss_grad_kernel(const float* __restrict__ tab, int n_steps, int mean_type, int var_type,
                                                        const float* __restrict__ x0, const float* __restrict__ xt,
                                                        const float* __restrict__ noise, const float* __restrict__ mo,
                                                        const int64_t* __restrict__ t, int batch, int frames, int C, int hw,
                                                        float vb_scale, float* __restrict__ dmo) {
  const int b = blockIdx.y;
  const int ti = (int)t[b];
  const float coef1 = tab[DT_COEF1 * n_steps + ti], coef2 = tab[DT_COEF2 * n_steps + ti];
  const float plv = tab[DT_POST_LOGVAR * n_steps + ti], lb = tab[DT_LOG_BETAS * n_steps + ti];
  const float srec = tab[DT_SQRT_RECIP * n_steps + ti], srecm1 = tab[DT_SQRT_RECIPM1 * n_steps + ti];
  const size_t chw = (size_t)C * hw, per = (size_t)frames * chw;
  const int Cm = var_type == 0 ? 2 * C : C;
  const float wmse = 2.0f / ((float)per * (float)batch);
  const float wvb = vb_scale / ((float)per * (float)batch * 0.6931471805599453f);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < per; e += (size_t)gridDim.x * 256) {
    const size_t f = e / chw, r = e % chw;
    const size_t i = (size_t)b * per + e;
    const size_t o = (((size_t)b * frames + f) * Cm) * hw + r;
    const float pred = mo[o];
    const float xs = x0[i], xv = xt[i];
    const float target = mean_type == 1 ? xs : noise[i];
    dmo[o] = wmse * (pred - target);
    if (var_type == 0) {
      const float v = mo[o + chw];
      const float frac = (v + 1.0f) / 2.0f;
      const float lv = frac * lb + (1.0f - frac) * plv;
      const float x0p = mean_type == 1 ? pred : srec * xv - srecm1 * pred;
      const float mean = coef1 * x0p + coef2 * xv;
      float dlv;
      if (ti != 0) {
        const float dm = (coef1 * xs + coef2 * xv) - mean;
        dlv = 0.5f * (1.0f - expf(plv - lv) - dm * dm * expf(-lv));
      } else {
        const float centered = xs - mean;
        const float inv_stdv = expf(-(0.5f * lv));
        const float pin = inv_stdv * (centered + 0.00392156862745098f), min_ = inv_stdv * (centered - 0.00392156862745098f);
        const float cp = cdf_approx(pin), cm = cdf_approx(min_);
        // d/d log_scale of the selected log-probability (d pin / d ls = -pin, d min / d ls = -min)
        float dls;
        if (xs < -0.999f) dls = cp > 1e-12f ? cdf_approx_grad(pin) * (-pin) / cp : 0.f;
        else if (xs > 0.999f) dls = (1.0f - cm) > 1e-12f ? -cdf_approx_grad(min_) * (-min_) / (1.0f - cm) : 0.f;
        else dls = (cp - cm) > 1e-12f ? (cdf_approx_grad(pin) * (-pin) - cdf_approx_grad(min_) * (-min_)) / (cp - cm) : 0.f;
        dlv = -0.5f * dls;                                         // term = -log p, log_scale = lv / 2
      }
      dmo[o + chw] = wvb * dlv * 0.5f * (lb - plv);                 // d lv / d v = (max_log - min_log) / 2
    }
  }
}

// ---------------------------------------------------------------------------------------------- optimiser
// sum of squares of the flat gradient: partial per block, then a single-block finalize -> norm, clip coefficient
__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ g, size_t n, double* __restrict__ partial) {
  // round 6: 16-byte loads and four independent fp64 chains per thread (2.4 -> the copy rate of the other passes); fixed order per launch shape
  double a4[4] = {0.0, 0.0, 0.0, 0.0};
  const bool vec = ((uintptr_t)g & 15) == 0;
  const size_t n4 = vec ? n / 4 : 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 q = ((const float4*)g)[i];
    a4[0] += (double)q.x * (double)q.x;
    a4[1] += (double)q.y * (double)q.y;
    a4[2] += (double)q.z * (double)q.z;
    a4[3] += (double)q.w * (double)q.w;
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a4[0] += (double)g[i] * (double)g[i];
  const double a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  __shared__ double red[256];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// stats[0] = total 2-norm, stats[1] = coefficient the optimiser multiplies every gradient with (utils.py:108-114), stats[2] = 1 when
// the update must be SKIPPED (non-finite gradient norm: an overflow of the loss-scaled half-precision backward or of an f16 forward
// activation; the reference trains in fp32 range and cannot produce one).  scaler (device float[8], optional) is the trainer's
// loss-scaling state, advanced here so that no host round trip is needed:
//   [0] loss scale (power of two)  [1] applied steps since the scale last changed  [2] applied optimiser updates (AdamW's `step`)
//   [3] skipped updates in total   [4] 1 if this call skipped                      [5] dynamic scaling on / off
//   [6] growth interval            [7] largest scale
// On a skip with dynamic scaling the scale halves (floor 1); after [6] consecutive applied updates it doubles (cap [7]).
__global__ void gradnorm_finalize_kernel(const double* __restrict__ partial, int blocks, float max_norm, int clip, float* __restrict__ stats,
                                         float* __restrict__ scaler) {
  // round 6: 256 threads add the partials (thread t: t, t + 256, ...; then a fixed tree) -- one thread walking all of them was 55 us
  __shared__ double fred[256];
  double a = 0.0;
  for (int k = threadIdx.x; k < blocks; k += 256) a += partial[k];
  fred[threadIdx.x] = a;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) fred[threadIdx.x] += fred[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  a = fred[0];
  const float norm = (float)sqrt(a);
  const bool ok = isfinite(norm);
  stats[0] = norm;
  stats[1] = !ok ? 0.0f : clip ? fminf(max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
  stats[2] = ok ? 0.0f : 1.0f;
  if (scaler) {
    if (ok) {
      scaler[2] += 1.0f;
      scaler[1] += 1.0f;
      scaler[4] = 0.0f;
      if (scaler[5] != 0.0f && scaler[1] >= scaler[6]) {
        scaler[0] = fminf(scaler[0] * 2.0f, scaler[7]);
        scaler[1] = 0.0f;
      }
    } else {
      scaler[3] += 1.0f;
      scaler[4] = 1.0f;
      scaler[1] = 0.0f;
      if (scaler[5] != 0.0f) scaler[0] = fmaxf(scaler[0] * 0.5f, 1.0f);
    }
  }
}
// torch.optim.AdamW single-tensor update + update_ema; g is multiplied by stats[1] first (clipping) and zeroed afterwards.
// stats[2] != 0: the update is skipped (parameters, moments and EMA untouched, gradients still zeroed).  step_dev != nullptr: the
// bias corrections come from the trainer's own count of APPLIED updates (scaler[2], already advanced for this call) instead of the
// host's step -- a fresh optimiser state starts at 1 whatever the training-step counter says (torch.optim.AdamW counts its own steps)
__global__ void __launch_bounds__(256) adamw_ema_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ ema, size_t n, float lr, float b1,
                                                        float b2, float eps, float wd, float bc1, float sqrt_bc2, float ema_decay,
                                                        const float* __restrict__ stats, const float* __restrict__ step_dev) {
#pragma clang fp contract(off)
  const float coef = stats ? stats[1] : 1.0f;
  const bool skip = stats && stats[2] != 0.0f;
  if (step_dev) {
    const double st = (double)step_dev[0];
    bc1 = 1.0f - (float)pow((double)b1, st);
    sqrt_bc2 = (float)sqrt(1.0 - pow((double)b2, st));
  }
  // one element of the update, the arithmetic of torch.optim.AdamW's single-tensor path in its order (no contraction, see above)
  auto upd = [&](float& pi, float& gi_io, float& mi_io, float& vi_io, float& ei) {
    const float gi = gi_io * coef;
    pi = pi * (1.0f - lr * wd);
    const float mi = mi_io * b1 + gi * (1.0f - b1);
    const float vi = vi_io * b2 + (gi * gi) * (1.0f - b2);
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    pi = pi - (lr / bc1) * (mi / denom);
    mi_io = mi;
    vi_io = vi;
    ei = ei * ema_decay + pi * (1.0f - ema_decay);
    gi_io = 0.f;
  };
  // round 6: 16-byte accesses (the scalar form of rounds 2 - 5 ran the 2.6 GB of state at 2.9 TB/s); the five buffers are slices of
  // caller-owned flat buffers at the same element offset -- all 16-byte aligned or none (then the scalar tail loop takes everything)
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)(ema ? ema : p)) & 15) == 0);
  const size_t n4 = vec ? n / 4 : 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    if (skip) {
      ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    float4 P = ((float4*)p)[i], G = ((float4*)g)[i], Mo = ((float4*)m)[i], V = ((float4*)v)[i];
    float4 E = ema ? ((float4*)ema)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    upd(P.x, G.x, Mo.x, V.x, E.x);
    upd(P.y, G.y, Mo.y, V.y, E.y);
    upd(P.z, G.z, Mo.z, V.z, E.z);
    upd(P.w, G.w, Mo.w, V.w, E.w);
    ((float4*)p)[i] = P;
    ((float4*)m)[i] = Mo;
    ((float4*)v)[i] = V;
    if (ema) ((float4*)ema)[i] = E;
    ((float4*)g)[i] = G;
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (skip) {
      g[i] = 0.f;
      continue;
    }
    float pi = p[i], gi = g[i], mi = m[i], vi = v[i], ei = ema ? ema[i] : 0.f;
    upd(pi, gi, mi, vi, ei);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (ema) ema[i] = ei;
    g[i] = gi;
  }
}

inline int blocks_for(size_t n, int cap = 4096) {
  const size_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > (size_t)cap ? (size_t)cap : b));
}

}  // namespace

#define LATTE_DT_SWITCH(dtype, CALL)                                    \
  do {                                                                  \
    if ((dtype) == LATTE_DTYPE_BF16) { CALL(LATTE_DTYPE_BF16); }        \
    else if ((dtype) == LATTE_DTYPE_F16) { CALL(LATTE_DTYPE_F16); }     \
    else return fail(LATTE_ERR_INVALID, "train kernel: unknown dtype"); \
  } while (0)

// rows one wave walks: short runs keep >= 10 waves per CU busy at the training batch sizes (64-row runs: 320 waves for
// M = 20480 -- the first version ran the LN backward at 0.8 TB/s); the partial rows cost 2 D floats per run
int train_rows_per_run(int rps) {
  for (int r = 8; r > 1; r >>= 1)
    if (rps % r == 0) return r;
  return 1;
}

int launch_gated_add(const float* x_in, const half_t* y, const float* gate, int gate_stride, float* x_out, int M, int D, int rps,
                     int dtype, hipStream_t st) {
  if (D % 4) return fail(LATTE_ERR_INVALID, "gated_add: D % 4 != 0");
#define CALL(DT) hipLaunchKernelGGL(gated_add_kernel<DT>, dim3(blocks_for((size_t)M * D / 4)), dim3(256), 0, st, x_in, y, gate, gate_stride, x_out, M, D, rps)
  LATTE_DT_SWITCH(dtype, CALL);
#undef CALL
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

// x_out = x_in + gate * y (+ temp_embed when te != nullptr: added to x_out as ln_modulate's x_rw form does), xn = LN-modulate(x_out)
int launch_gated_add_ln(const float* x_in, const half_t* y, const float* gate, int gate_stride, float* x_out, half_t* xn, const float* shift,
                        const float* scale, int mod_stride, int M, int D, int rps, const float* te, int T, int F, int dtype, hipStream_t st) {
  if (D % 4 || D > NQ_MAX * 256) return fail(LATTE_ERR_INVALID, "gated_add_ln: need D % 4 == 0 and D <= 1280");
#define CALL(DT)                                                                                                                     \
  if (te) hipLaunchKernelGGL((gated_add_ln_kernel<DT, true>), dim3((M + 3) / 4), dim3(256), 0, st, x_in, y, gate, gate_stride, x_out, xn, \
                             shift, scale, mod_stride, M, D, rps, te, T, F);                                                         \
  else hipLaunchKernelGGL((gated_add_ln_kernel<DT, false>), dim3((M + 3) / 4), dim3(256), 0, st, x_in, y, gate, gate_stride, x_out, xn,  \
                          shift, scale, mod_stride, M, D, rps, te, T, F)
  LATTE_DT_SWITCH(dtype, CALL);
#undef CALL
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

// partial: float [M / (4 R)][1 or 2][D];  dgate: [B][out_stride] (assigned) or nullptr (no finalize launch)
int launch_gate_bwd(const float* dx, const half_t* y, const float* gate, int gate_stride, half_t* dy, float* partial, float* dgate,
                    int out_stride, int M, int D, int rps, int dtype, hipStream_t st, int bias_partial) {
  if (D % 4 || D > NQ_MAX * 256) return fail(LATTE_ERR_INVALID, "gate_bwd: need D % 4 == 0 and D <= 1280");
  const int R = train_rows_per_run(rps), runs = M / R;
  if (rps % (4 * R)) return fail(LATTE_ERR_INVALID, "gate_bwd: rows per sample must be a multiple of 4 runs");
  const int ns = bias_partial ? 2 : 1;
#define CALL(DT)                                                                                                                          \
  if (bias_partial) hipLaunchKernelGGL((gate_bwd_kernel<DT, true>), dim3(runs / 4), dim3(256), 3 * ns * D * sizeof(float), st, dx, y, gate, \
                                       gate_stride, dy, partial, M, D, rps, R);                                                           \
  else hipLaunchKernelGGL((gate_bwd_kernel<DT, false>), dim3(runs / 4), dim3(256), 3 * ns * D * sizeof(float), st, dx, y, gate, gate_stride, \
                          dy, partial, M, D, rps, R)
  LATTE_DT_SWITCH(dtype, CALL);
#undef CALL
  if (dgate)
    hipLaunchKernelGGL(sample_colsum_finalize_kernel, dim3((D + 63) / 64, M / rps), dim3(256), 0, st, partial, rps / (4 * R), ns, 0, D, dgate,
                       out_stride);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

// partial: float [M / (4 R)][2][D];  dshift / dscale: [B][out_stride] (assigned) or nullptr (no finalize launches).
// y2 != nullptr: the gated residual's backward of the branch below rides on the pass (dy2 = gate2 * dx_out in half,
// gpartial [M / (4 R)][2][D] = {sum dx * y2, sum gate2 * dx} per block of 4 runs) -- gate_bwd_kernel<DT, true> without re-reading dx
int launch_ln_bwd(const half_t* dy, const float* x, const float* scale, int mod_stride, const float* dx_in, float* dx_out, float* partial,
                  float* dshift, float* dscale, int out_stride, int M, int D, int rps, int dtype, hipStream_t st, const half_t* y2,
                  const float* gate2, int gate2_stride, half_t* dy2, float* gpartial) {
  if (D % 4 || D > NQ_MAX * 256) return fail(LATTE_ERR_INVALID, "ln_bwd: need D % 4 == 0 and D <= 1280");
  const int R = train_rows_per_run(rps), runs = M / R;
  if (rps % (4 * R)) return fail(LATTE_ERR_INVALID, "ln_bwd: rows per sample must be a multiple of 4 runs");
  if (y2 && (!gate2 || !dy2 || !gpartial)) return fail(LATTE_ERR_INVALID, "ln_bwd: the gated form needs gate2, dy2 and gpartial");
  const int ns = y2 ? 4 : 2;
#define CALLQ(DT, NQ)                                                                                                                  \
  if (y2) hipLaunchKernelGGL((ln_bwd_kernel<DT, NQ, true>), dim3(runs / 4), dim3(256), 3 * ns * D * sizeof(float), st, dy, x, scale,   \
                             mod_stride, dx_in, dx_out, partial, M, D, rps, R, y2, gate2, gate2_stride, dy2, gpartial);                \
  else hipLaunchKernelGGL((ln_bwd_kernel<DT, NQ, false>), dim3(runs / 4), dim3(256), 3 * ns * D * sizeof(float), st, dy, x, scale,     \
                          mod_stride, dx_in, dx_out, partial, M, D, rps, R, y2, gate2, gate2_stride, dy2, gpartial)
#define CALL(DT)                       \
  if (D <= 768) { CALLQ(DT, 3); }      \
  else { CALLQ(DT, NQ_MAX); }
  LATTE_DT_SWITCH(dtype, CALL);
#undef CALL
#undef CALLQ
  if (dshift) {   // nullptr: the stage's finalize kernel reads the partial rows
    hipLaunchKernelGGL(sample_colsum_finalize_kernel, dim3((D + 63) / 64, M / rps), dim3(256), 0, st, partial, rps / (4 * R), 2, 0, D, dshift,
                       out_stride);
    hipLaunchKernelGGL(sample_colsum_finalize_kernel, dim3((D + 63) / 64, M / rps), dim3(256), 0, st, partial, rps / (4 * R), 2, 1, D, dscale,
                       out_stride);
  }
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_gelu_fwd(const half_t* u, half_t* h, size_t n, int dtype, hipStream_t st) {
  if (n % 4) return fail(LATTE_ERR_INVALID, "gelu: n % 4 != 0");
#define CALL(DT) hipLaunchKernelGGL((gelu_kernel<DT, false>), dim3(blocks_for(n / 4, 16384)), dim3(256), 0, st, u, (const half_t*)nullptr, h, n / 4)
  LATTE_DT_SWITCH(dtype, CALL);
#undef CALL
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}
int launch_gelu_bwd(const half_t* u, const half_t* dh, half_t* du, size_t n, int dtype, hipStream_t st) {
  if (n % 4) return fail(LATTE_ERR_INVALID, "gelu: n % 4 != 0");
#define CALL(DT) hipLaunchKernelGGL((gelu_kernel<DT, true>), dim3(blocks_for(n / 4, 16384)), dim3(256), 0, st, u, dh, du, n / 4)
  LATTE_DT_SWITCH(dtype, CALL);
#undef CALL
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int colsum_chunks(int M) { return (M + CS_ROWS - 1) / CS_ROWS; }
// out[col] (+)= sum_m in[m][col];  partial: float [colsum_chunks(M)][C]
int launch_colsum_half(const half_t* in, int M, int C, float* partial, float* out, int accumulate, int dtype, hipStream_t st) {
  if (C % 8) return fail(LATTE_ERR_INVALID, "colsum: C % 8 != 0");
  const int chunks = colsum_chunks(M);
#define CALL(DT) hipLaunchKernelGGL(colsum_half_kernel<DT>, dim3((C + 127) / 128, chunks), dim3(256), 0, st, in, M, C, partial)
  LATTE_DT_SWITCH(dtype, CALL);
#undef CALL
  if (out)   // nullptr: the chunk partials [colsum_chunks(M)][C] are reduced by the stage's finalize kernel
    hipLaunchKernelGGL(split_reduce_kernel, dim3(blocks_for((size_t)C)), dim3(256), 0, st, partial, chunks, (size_t)C, (size_t)C, out, accumulate,
                       (const float*)nullptr);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}
int launch_split_reduce(const float* partial, int splits, size_t stride, size_t n, float* out, int accumulate, hipStream_t st,
                        const float* inv_scale_dev) {
  if (n % 4 == 0 && stride % 4 == 0 && ((uintptr_t)partial & 15) == 0 && ((uintptr_t)out & 15) == 0 && n >= 4096) {
    hipLaunchKernelGGL(split_reduce4_kernel, dim3(blocks_for(n / 4, 8192)), dim3(256), 0, st, (const float4*)partial, splits, stride / 4, n / 4,
                       (float4*)out, accumulate, inv_scale_dev);
    LATTE_HIP(hipGetLastError());
    return LATTE_OK;
  }
  hipLaunchKernelGGL(split_reduce_kernel, dim3(blocks_for(n)), dim3(256), 0, st, partial, splits, stride, n, out, accumulate, inv_scale_dev);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_pack_weight(const float* w, half_t* wn, half_t* wt, int N, int K, int dtype, hipStream_t st) {
#define CALL(DT) hipLaunchKernelGGL(pack_weight_kernel<DT>, dim3((K + 31) / 32, (N + 31) / 32), dim3(256), 0, st, w, wn, wt, N, K)
  LATTE_DT_SWITCH(dtype, CALL);
#undef CALL
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

// splits > 1: the contraction is cut into `splits` ranges, partial products go to `ws` (splits * M * N floats, dense [M][N]) and are
// reduced into C (which must then be dense row-major: scm = N, scn = 1)
int launch_naive_gemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long scm, long scn, int M, int N,
                      int K, float alpha, int accumulate, hipStream_t st, int splits, float* ws) {
  if (splits > 1) {
    if (!ws || scm != N || scn != 1) return fail(LATTE_ERR_INVALID, "naive_gemm: split needs a workspace and a dense output");
    const int chunk = (K + splits - 1) / splits;
    const int ns = (K + chunk - 1) / chunk;
    hipLaunchKernelGGL(naive_gemm_kernel, dim3((N + 63) / 64, (M + 3) / 4, ns), dim3(256), 0, st, A, sam, sak, B, sbk, sbn, ws, (long)N, 1L, M,
                       N, K, alpha, 0, chunk, (long)M * N);
    hipLaunchKernelGGL(split_reduce_kernel, dim3(blocks_for((size_t)M * N)), dim3(256), 0, st, ws, ns, (size_t)M * N, (size_t)M * N, C,
                       accumulate, (const float*)nullptr);
  } else {
    hipLaunchKernelGGL(naive_gemm_kernel, dim3((N + 63) / 64, (M + 3) / 4, 1), dim3(256), 0, st, A, sam, sak, B, sbk, sbn, C, scm, scn, M, N,
                       K, alpha, accumulate, 0, 0L);
  }
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}
int launch_tfreq(const int64_t* t, float* out, int B, hipStream_t st) {
  hipLaunchKernelGGL(tfreq_kernel, dim3(B), dim3(128), 0, st, t, out, B);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}
int launch_gather_i64(const int64_t* table, const int64_t* idx, int64_t* out, int n, hipStream_t st) {
  hipLaunchKernelGGL(gather_i64_kernel, dim3((n + 255) / 256), dim3(256), 0, st, table, idx, out, n);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_unpatchify_bwd(const float* dout, float* dtok, int BF, int G, int p, int Cout, hipStream_t st) {
  hipLaunchKernelGGL(unpatchify_bwd_kernel, dim3(blocks_for((size_t)BF * G * G * p * p * Cout)), dim3(256), 0, st, dout, dtok, BF, G, p, Cout);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}
int launch_im2col_patch(const float* x, float* pix, int BF, int G, int p, int C, hipStream_t st) {
  hipLaunchKernelGGL(im2col_patch_kernel, dim3(blocks_for((size_t)BF * G * G * p * p * C)), dim3(256), 0, st, x, pix, BF, G, p, C);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}
int launch_embedding_bwd(const float* dc, const int64_t* idx, float* dtable, int B, int D, hipStream_t st) {
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3((D + 255) / 256), dim3(256), 0, st, dc, idx, dtable, B, D);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}
int launch_silu_bwd(const float* dout, const float* pre, float* din, size_t n, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, st, dout, pre, din, n, accumulate);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}
int launch_add_rows(float* dst, const float* src, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(add_rows_kernel, dim3(blocks_for(n)), dim3(256), 0, st, dst, src, n);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}
int launch_rows_sum(const float* in, int B, long stride, int N, float* out, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(rows_sum_kernel, dim3((N + 255) / 256), dim3(256), 0, st, in, B, stride, N, out, accumulate);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_loss_grad(const float* tables, int n_steps, int mean_type, int var_type, const float* x_start, const float* x_t,
                     const float* noise, const float* model_out, const int64_t* t, int batch, int frames, int channels, int hw,
                     float vb_scale, float* dmodel_out, hipStream_t st) {
  const size_t per = (size_t)frames * channels * hw;
  hipLaunchKernelGGL(loss_grad_kernel, dim3(blocks_for(per, 256), batch), dim3(256), 0, st, tables, n_steps, mean_type, var_type, x_start,
                     x_t, noise, model_out, t, batch, frames, channels, hw, vb_scale, dmodel_out);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

constexpr int SUMSQ_BLOCKS = 1024;
int sumsq_blocks() { return SUMSQ_BLOCKS; }
// stats: float[4] = {norm, clip coefficient, skip flag, -};  partial: double [sumsq_blocks()];  scaler: device float[8] or nullptr
int launch_grad_norm(const float* g, size_t n, double* partial, float max_norm, int clip, float* stats, float* scaler, hipStream_t st) {
  hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, st, g, n, partial);
  hipLaunchKernelGGL(gradnorm_finalize_kernel, dim3(1), dim3(256), 0, st, partial, SUMSQ_BLOCKS, max_norm, clip, stats, scaler);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}
int launch_adamw_ema(float* p, float* g, float* m, float* v, float* ema, size_t n, float lr, float b1, float b2, float eps, float wd,
                     int step, float ema_decay, const float* stats, const float* step_dev, hipStream_t st) {
  // step >= 1: the caller's AdamW step; step == 0: the device-side count of applied updates (step_dev)
  const float bc1 = step >= 1 ? 1.0f - (float)std::pow((double)b1, (double)step) : 1.0f;
  const float sqrt_bc2 = step >= 1 ? (float)std::sqrt(1.0 - std::pow((double)b2, (double)step)) : 1.0f;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(blocks_for(n, 8192)), dim3(256), 0, st, p, g, m, v, ema, n, lr, b1, b2, eps, wd, bc1, sqrt_bc2,
                     ema_decay, stats, step >= 1 ? nullptr : step_dev);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

}  // namespace latte
