// HBM-bound and small kernels of the Latte denoiser + the sampler update (gfx950).
//
//  ln_modulate   : LayerNorm(eps 1e-6, no affine) + adaLN modulate -> half      latte.py:28-29,166,168,179-180
//  small_linear  : exact-fp32 row-vector linears (t-embedder MLP, all adaLN)     latte.py:90-94,119-123,172-178,192-198
//  patch_embed   : Conv2d(k=s=p) as a [C p p]-deep dot + bias + pos_embed        latte.py:233,330-331
//  final_layer   : LN + modulate + Linear(D, p*p*Cout) + unpatchify              latte.py:197-201,297-310,374-376
//  cfg_combine   : classifier-free guidance on the first 4 channels              latte.py:394-398
//  sampler_update: p_mean_variance + p_sample / ddim_sample                      gaussian_diffusion.py:254-336,380-421,517-564
#include "common.h"

namespace latte {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
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

// split-operand pair (mfma_util.h: split2): hi = nearest half, lo = nearest half of the remainder
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
// four OCP e4m3 codes of (v - hi) * 2^LO8_A_SHIFT, clamped to the format's +-448 (the conversion itself does not saturate) -- the
// operand of the GEMMs' block-scaled correction pass (gemm_pw.hip), whose constant E8M0 scale multiplies the 2^-LO8_A_SHIFT back.
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

__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// FP4 (e2m1) block quantisation with a power-of-two scale (round 6): the E8M0 exponent for a block whose largest magnitude is `amax`
// -- ONE BELOW the smallest e with amax / 2^e <= 6 (the format's largest value): the top binade of the block saturates at 6 and everything
// else gains a bit (remainders of N(0, 1)-like rows keep 1.7 - 2.2 % of their variance instead of 3 - 4.6 %, heavy-tailed rows 5.6 % instead
// of 12 %: simulation in DESIGN.md section 2), clamped to the scale byte's range -- and four values -> one
// half-word of four codes (element j in bits 4 j) by the hardware convert (round to nearest even, saturating at +-6).
__device__ __forceinline__ int quant4_exponent(float amax) {
  if (!(amax > 0.f)) return -127;
  int ex;
  const float m = __builtin_frexpf(amax * (1.0f / 6.0f), &ex);   // amax / 6 = m 2^ex, m in [0.5, 1)
  const int e = (m == 0.5f ? ex - 1 : ex) - 1;
  return e < -127 ? -127 : e > 127 ? 127 : e;
}
__device__ __forceinline__ unsigned int quant4_pk4(float v0, float v1, float v2, float v3, float scale_pow2) {
  unsigned int w = 0;
  w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, v0, v1, scale_pow2, 0);
  w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, v2, v3, scale_pow2, 1);
  return w & 0xffffu;
}



// ------------------------------------------------------------------------------------------------
// One wave per token row; lane owns the 16-byte chunks {lane + 64 c} of the row (float4 loads, 8-byte half stores: the
// widest accesses the row length allows -- D / 4 chunks, the last group of 64 only half populated when D / 128 is odd).
// Two-pass statistics in registers.
// SPLIT: y is [M, 2 D] -- columns [0, D) the half nearest to the value, [D, 2 D) the half nearest to the remainder (mfma_util.h:
// split2): the K-concatenated operand of a linear whose weight is stored [W | W].
// SPLIT == 2 (f16 only): y stays [M, D] (the nearest f16) and y8 [M, D] bytes receives the fp8 remainder (split8_f16) -- the operand
// pair of a GEMM with the fp8 correction pass (GemmArgs::A8).
template <int NCH, int DT, bool ADD_TE, int SPLIT = 0>
__global__ void __launch_bounds__(256) ln_modulate_kernel(const float* __restrict__ x_in, float* x_rw,
                                                          half_t* __restrict__ y, const float* __restrict__ shift,
                                                          const float* __restrict__ scale, int mod_stride, int M,
                                                          int rows_per_sample, const float* __restrict__ te, int T,
                                                          int F, unsigned char* __restrict__ y8 = nullptr,
                                                          unsigned char* __restrict__ y4s = nullptr) {
  constexpr int D = NCH * 128;
  constexpr int NT = NCH * 32;            // float4 chunks per row
  constexpr int NQ = (NT + 63) / 64;      // chunk groups per lane
  const int lane = threadIdx.x & 63;
  // (Round 4 measured other row walks -- descending, descending inside every sample -- against the cache recency of what this pass
  //  reads, together with reversed tile walks of the GEMMs around it: every combination within +-0.5 % of this one,
  //  profiles/r4_row_walk_probe_B8.log; removed again.)
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float4* xr = (const float4*)(x_in + (size_t)row * D);
  auto has = [&](int c) -> bool { return (c + 1) * 64 <= NT || c * 64 + lane < NT; };
  float4 v[NQ];
#pragma unroll
  for (int c = 0; c < NQ; ++c) v[c] = has(c) ? xr[c * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  // the modulation vectors of the row's sample, requested together with the row: their latency sits under the two reductions
  // (round 5: the LayerNorm class 42.2 -> 40.9 us per launch in the XL/2 forward at B = 8, same bits)
  const int smp = row / rows_per_sample;
  const float4* sh = (const float4*)(shift + (size_t)smp * mod_stride);
  const float4* sc = (const float4*)(scale + (size_t)smp * mod_stride);
  float4 sha[NQ], sca[NQ];
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    sha[c] = has(c) ? sh[c * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    sca[c] = has(c) ? sc[c * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if constexpr (ADD_TE) {
    const int f = (row / T) % F;
    const float4* tr = (const float4*)(te + (size_t)f * D);
    float4* xw = (float4*)(x_rw + (size_t)row * D);
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      if (has(c)) {
        const float4 e = tr[c * 64 + lane];
        v[c].x += e.x; v[c].y += e.y; v[c].z += e.z; v[c].w += e.w;
        xw[c * 64 + lane] = v[c];
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NQ; ++c) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);     // absent chunks hold zeros
  const float mean = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    if (has(c)) {
      const float a = v[c].x - mean, b = v[c].y - mean, d = v[c].z - mean, e = v[c].w - mean;
      q += (a * a + b * b) + (d * d + e * e);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + 1e-6f);
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
  u32x2_t* yr = (u32x2_t*)(y + (size_t)row * (SPLIT == 1 ? 2 * D : D));
  if constexpr (SPLIT == 3) {
    // f16 + FP4 remainder with one E8M0 scale per ROW (GemmArgs::A4 / A4s): y8 = [M, lo4_pitch(D)] bytes of e2m1 codes, y4s = [M] scale bytes
    float lo[NQ][4];
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      if (has(c)) {
        const float4 a = sha[c], b = sca[c];
        const float o0 = (v[c].x - mean) * rstd * (1.0f + b.x) + a.x;
        const float o1 = (v[c].y - mean) * rstd * (1.0f + b.y) + a.y;
        const float o2 = (v[c].z - mean) * rstd * (1.0f + b.z) + a.z;
        const float o3 = (v[c].w - mean) * rstd * (1.0f + b.w) + a.w;
        const _Float16 h0 = (_Float16)o0, h1 = (_Float16)o1, h2 = (_Float16)o2, h3 = (_Float16)o3;
        yr[c * 64 + lane] = (u32x2_t){pack2<LATTE_DTYPE_F16>((float)h0, (float)h1), pack2<LATTE_DTYPE_F16>((float)h2, (float)h3)};
        lo[c][0] = o0 - (float)h0; lo[c][1] = o1 - (float)h1; lo[c][2] = o2 - (float)h2; lo[c][3] = o3 - (float)h3;
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(lo[c][0]), fabsf(lo[c][1])), fmaxf(fabsf(lo[c][2]), fabsf(lo[c][3]))));
      } else {
        lo[c][0] = lo[c][1] = lo[c][2] = lo[c][3] = 0.f;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const int e = quant4_exponent(amax);
    const float sc2 = __builtin_ldexpf(1.0f, e);
    unsigned short* y4r = (unsigned short*)(y8 + (size_t)row * (size_t)((D + 255) / 256 * 128));
#pragma unroll
    for (int c = 0; c < NQ; ++c)
      if (has(c)) y4r[c * 64 + lane] = (unsigned short)quant4_pk4(lo[c][0], lo[c][1], lo[c][2], lo[c][3], sc2);
    if (lane == 0) y4s[row] = (unsigned char)(e + 127);
    return;
  }
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    if (has(c)) {
      const float4 a = sha[c], b = sca[c];
      const float o0 = (v[c].x - mean) * rstd * (1.0f + b.x) + a.x;
      const float o1 = (v[c].y - mean) * rstd * (1.0f + b.y) + a.y;
      const float o2 = (v[c].z - mean) * rstd * (1.0f + b.z) + a.z;
      const float o3 = (v[c].w - mean) * rstd * (1.0f + b.w) + a.w;
      if constexpr (SPLIT == 1) {
        unsigned int h0_, l0_, h1_, l1_;
        split2<DT>(o0, o1, h0_, l0_);
        split2<DT>(o2, o3, h1_, l1_);
        const u32x2_t hi = {h0_, h1_}, lo = {l0_, l1_};
        yr[c * 64 + lane] = hi;
        yr[NT + c * 64 + lane] = lo;     // + D halves = NT 8-byte chunks
      } else if constexpr (SPLIT == 2) {
        unsigned int h0_, h1_;
        const unsigned int l8 = split8_f16(o0, o1, o2, o3, h0_, h1_);
        yr[c * 64 + lane] = (u32x2_t){h0_, h1_};
        ((unsigned int*)(y8 + (size_t)row * D))[c * 64 + lane] = l8;
      } else {
        yr[c * 64 + lane] = (u32x2_t){pack2<DT>(o0, o1), pack2<DT>(o2, o3)};
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// out[b, n] = bias[n] + sum_k f(in[b, k]) W[n, k]   (one wave per output feature n, fp32 FMA chain).
// The weight row is read ONCE into registers (K <= 1152) and reused for every batch row: the adaLN
// table is 0.9 GB of fp32 weights per forward, so this kernel is HBM-bound on W alone.
constexpr int SL_MAXCH = 9;  // float2 chunks per lane: K / 128
template <int IN_MODE>
__global__ void __launch_bounds__(256) small_linear_kernel(const float* __restrict__ in, const int64_t* __restrict__ t,
                                                           const float* __restrict__ W, const float* __restrict__ bias,
                                                           const float* __restrict__ add_table,
                                                           const int64_t* __restrict__ add_idx, float* __restrict__ out,
                                                           int B, int N, int K, int out_stride) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float2* wr = (const float2*)(W + (size_t)n * K);
  const int nch = K >> 7;  // K % 128 == 0, nch <= SL_MAXCH (checked by the launcher)
  float2 w[SL_MAXCH];
#pragma unroll
  for (int c = 0; c < SL_MAXCH; ++c) w[c] = c < nch ? wr[c * 64 + lane] : make_float2(0.f, 0.f);
  const float bn = bias[n];
  for (int b = 0; b < B; ++b) {
    float acc = 0.f;
    if constexpr (IN_MODE == IN_TFREQ) {
      // latte.py:97-117: freqs = exp(-ln(1e4) * arange(half) / half) in fp32; emb = [cos | sin]
      const float tv = (float)t[b];
      const int half = K >> 1;
#pragma unroll
      for (int c = 0; c < SL_MAXCH; ++c) {
        if (c < nch) {
          float e[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int k = 2 * (c * 64 + lane) + u;
            const int fi = k < half ? k : k - half;
            const float freq = expf((-9.210340371976184f * (float)fi) / (float)half);
            const float arg = tv * freq;
            e[u] = k < half ? cosf(arg) : sinf(arg);
          }
          acc = fmaf(e[0], w[c].x, acc);
          acc = fmaf(e[1], w[c].y, acc);
        }
      }
    } else {
      const float2* ir = (const float2*)(in + (size_t)b * K);
#pragma unroll
      for (int c = 0; c < SL_MAXCH; ++c) {
        if (c < nch) {
          float2 a = ir[c * 64 + lane];
          if constexpr (IN_MODE == IN_SILU) {
            a.x = silu(a.x);
            a.y = silu(a.y);
          }
          acc = fmaf(a.x, w[c].x, acc);
          acc = fmaf(a.y, w[c].y, acc);
        }
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      float r = acc + bn;
      if (add_table != nullptr) r += add_table[(size_t)add_idx[b] * N + n];
      out[(size_t)b * out_stride + n] = r;
    }
  }
}

// ------------------------------------------------------------------------------------------------
constexpr int PE_TOK = 16;
// block = 16 tokens x 128 output features: thread (token group tg = tid >> 5, feature quad q = tid & 31) computes 4 tokens x 4
// consecutive features, so weights / positions / outputs move as 16-byte accesses (512 B per token row and instruction)
__global__ void __launch_bounds__(128) patch_embed_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                          const float* __restrict__ bias, const float* __restrict__ pos,
                                                          float* __restrict__ out, int ntok, int C, int H, int p, int D) {
  extern __shared__ float xs[];  // [PE_TOK][K]
  const int K = C * p * p;
  const int G = H / p, T = G * G;
  const int tok0 = blockIdx.x * PE_TOK;
  for (int i = threadIdx.x; i < PE_TOK * K; i += 128) {
    const int tk = i / K, k = i % K;
    const int n = tok0 + tk;
    float v = 0.f;
    if (n < ntok) {
      const int bf = n / T, tt = n % T;
      const int hp = tt / G, wp = tt % G;
      const int c = k / (p * p), ii = (k / p) % p, jj = k % p;
      v = x[(((size_t)bf * C + c) * H + hp * p + ii) * H + wp * p + jj];
    }
    xs[i] = v;
  }
  __syncthreads();
  const int tg = threadIdx.x >> 5, q = threadIdx.x & 31;
  const int d = blockIdx.y * 128 + q * 4;
  float4 acc[4];
#pragma unroll
  for (int tk = 0; tk < 4; ++tk) acc[tk] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < K; ++k) {
    const float4 w = *(const float4*)(Wt + (size_t)k * D + d);
#pragma unroll
    for (int tk = 0; tk < 4; ++tk) {
      const float xv = xs[(tg * 4 + tk) * K + k];
      acc[tk].x = fmaf(xv, w.x, acc[tk].x);
      acc[tk].y = fmaf(xv, w.y, acc[tk].y);
      acc[tk].z = fmaf(xv, w.z, acc[tk].z);
      acc[tk].w = fmaf(xv, w.w, acc[tk].w);
    }
  }
  const float4 bv = *(const float4*)(bias + d);
#pragma unroll
  for (int tk = 0; tk < 4; ++tk) {
    const int n = tok0 + tg * 4 + tk;
    if (n < ntok) {
      const float4 pv = *(const float4*)(pos + (size_t)(n % T) * D + d);
      *(float4*)(out + (size_t)n * D + d) = make_float4(acc[tk].x + bv.x + pv.x, acc[tk].y + bv.y + pv.y, acc[tk].z + bv.z + pv.z,
                                                        acc[tk].w + bv.w + pv.w);
    }
  }
}

// ------------------------------------------------------------------------------------------------
constexpr int FL_ROWS = 8;
template <int NCH>
__global__ void __launch_bounds__(256) final_layer_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                                          const float* __restrict__ scale, int mod_stride,
                                                          const float* __restrict__ Wt, const float* __restrict__ bias,
                                                          float* __restrict__ out, int M, int rows_per_sample, int T,
                                                          int p, int Cout, int H) {
  constexpr int D = NCH * 128;
  extern __shared__ float xs[];  // [FL_ROWS][D] normalised + modulated rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * FL_ROWS;
  for (int rr = wave; rr < FL_ROWS; rr += 4) {
    const int row = row0 + rr;
    float2* xo = (float2*)(xs + rr * D);
    if (row >= M) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) xo[c * 64 + lane] = make_float2(0.f, 0.f);
      continue;
    }
    const float2* xr = (const float2*)(x + (size_t)row * D);
    float2 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      v[c] = xr[c * 64 + lane];
      s += v[c].x + v[c].y;
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float a = v[c].x - mean, b = v[c].y - mean;
      q += a * a + b * b;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + 1e-6f);
    const int smp = row / rows_per_sample;
    const float2* sh = (const float2*)(shift + (size_t)smp * mod_stride);
    const float2* sc = (const float2*)(scale + (size_t)smp * mod_stride);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float2 a = sh[c * 64 + lane], b = sc[c * 64 + lane];
      xo[c * 64 + lane] = make_float2((v[c].x - mean) * rstd * (1.0f + b.x) + a.x,
                                      (v[c].y - mean) * rstd * (1.0f + b.y) + a.y);
    }
  }
  __syncthreads();
  const int P = p * p * Cout;
  const int G = H / p;
  if (P == 32) {
    // patch 2 x 2 x 8 channels (every "/2" model with learned sigma): thread = (output feature j, one of 8 K slices); a weight
    // element is loaded ONCE per block and used for all FL_ROWS rows (the generic path below loads it once per row), the
    // row values are LDS broadcasts; partial sums meet through a half-wave shuffle and a [4 waves][rows][32] LDS patch.
    const int j = lane & 31, ksl = wave * 2 + (lane >> 5);
    constexpr int KSL = D / 8;
    float acc[FL_ROWS];
#pragma unroll
    for (int r = 0; r < FL_ROWS; ++r) acc[r] = 0.f;
    for (int k = ksl * KSL; k < (ksl + 1) * KSL; k += 4) {   // KSL = D / 8 is a multiple of 16
      const float w0 = Wt[(size_t)k * 32 + j], w1 = Wt[(size_t)(k + 1) * 32 + j], w2 = Wt[(size_t)(k + 2) * 32 + j],
                  w3 = Wt[(size_t)(k + 3) * 32 + j];
#pragma unroll
      for (int r = 0; r < FL_ROWS; ++r) {
        const float4 xv = *(const float4*)(xs + r * D + k);     // one 16-byte LDS broadcast per row and 4 k
        acc[r] = fmaf(xv.w, w3, fmaf(xv.z, w2, fmaf(xv.y, w1, fmaf(xv.x, w0, acc[r]))));
      }
    }
#pragma unroll
    for (int r = 0; r < FL_ROWS; ++r) acc[r] += __shfl_xor(acc[r], 32, 64);
    __syncthreads();                       // every wave is done reading the rows: reuse the LDS for the partial sums
    if (lane < 32) {
#pragma unroll
      for (int r = 0; r < FL_ROWS; ++r) xs[(wave * FL_ROWS + r) * 32 + j] = acc[r];
    }
    __syncthreads();
    const int rr = threadIdx.x >> 5, jj = threadIdx.x & 31;
    const int row = row0 + rr;
    if (row < M) {
      const float r = ((xs[(0 * FL_ROWS + rr) * 32 + jj] + xs[(1 * FL_ROWS + rr) * 32 + jj]) +
                       (xs[(2 * FL_ROWS + rr) * 32 + jj] + xs[(3 * FL_ROWS + rr) * 32 + jj])) + bias[jj];
      const int c = jj % Cout, qi = (jj / Cout) % p, pi = jj / (Cout * p);
      const int bf = row / T, tt = row % T;
      const int hp = tt / G, wp = tt % G;
      out[(((size_t)bf * Cout + c) * H + hp * p + pi) * H + wp * p + qi] = r;
    }
    return;
  }
  for (int idx = threadIdx.x; idx < FL_ROWS * P; idx += 256) {
    const int rr = idx / P, j = idx % P;
    const int row = row0 + rr;
    if (row >= M) continue;
    const float* xr = xs + rr * D;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int k = 0; k < D; k += 4) {
      a0 = fmaf(xr[k + 0], Wt[(size_t)(k + 0) * P + j], a0);
      a1 = fmaf(xr[k + 1], Wt[(size_t)(k + 1) * P + j], a1);
      a2 = fmaf(xr[k + 2], Wt[(size_t)(k + 2) * P + j], a2);
      a3 = fmaf(xr[k + 3], Wt[(size_t)(k + 3) * P + j], a3);
    }
    const float r = (a0 + a1) + (a2 + a3) + bias[j];
    // unpatchify 'nhwpqc->nchpwq' (latte.py:308): j = (pi*p + qi)*Cout + c
    const int c = j % Cout, qi = (j / Cout) % p, pi = j / (Cout * p);
    const int bf = row / T, tt = row % T;
    const int hp = tt / G, wp = tt % G;
    out[(((size_t)bf * Cout + c) * H + hp * p + pi) * H + wp * p + qi] = r;
  }
}

// ------------------------------------------------------------------------------------------------
// c[(i, b), :] = temb[i, :] (+ ytab[y[b], :]): the conditioning vector of step i, sample b (latte.py:337,348)
// (the rows only ever feed adaLN_modulation = Linear(SiLU(c)), latte.py:172-175: SiLU is applied here, once per row,
//  instead of once per output feature inside the linear)
__global__ void cond_rows_kernel(const float* __restrict__ temb, const float* __restrict__ ytab, const int64_t* __restrict__ y,
                                 float* __restrict__ out, int n_steps, int bu, int D) {
  const size_t total = (size_t)n_steps * bu * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const size_t r = i / D;
    const int b = (int)(r % bu), step = (int)(r / bu);
    float v = temb[(size_t)step * D + d];
    if (ytab != nullptr) v += ytab[(size_t)y[b] * D + d];
    out[i] = silu(v);
  }
}

// text_embedding_projection (latte.py:238-242,341): out[b, n] = bias[n] + sum_k silu(text[b, k]) * W[n, k], K = 77*768.
// Once per chain.  One wave per output feature streams its 236 KB weight row once per group of TP_B samples (the rows
// of W are the HBM traffic: 272 MB at D = 1152; the [B, K] input stays cache resident).
constexpr int TP_B = 8;
__global__ void __launch_bounds__(256) text_proj_kernel(const float* __restrict__ text, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ out, int B,
                                                        int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float2* wr = (const float2*)(W + (size_t)n * K);
  const int nch = K >> 7;   // K % 128 == 0 (launcher)
  for (int b0 = 0; b0 < B; b0 += TP_B) {
    float acc[TP_B];
#pragma unroll
    for (int u = 0; u < TP_B; ++u) acc[u] = 0.f;
    for (int c = 0; c < nch; ++c) {
      const float2 w = wr[c * 64 + lane];
#pragma unroll
      for (int u = 0; u < TP_B; ++u) {
        if (b0 + u < B) {
          const float2 a = ((const float2*)(text + (size_t)(b0 + u) * K))[c * 64 + lane];
          acc[u] = fmaf(silu(a.x), w.x, acc[u]);
          acc[u] = fmaf(silu(a.y), w.y, acc[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < TP_B; ++u) {
      const float r = wave_sum(acc[u]);
      if (lane == 0 && b0 + u < B) out[(size_t)(b0 + u) * N + n] = r + bias[n];
    }
  }
}

// Second half of the split-K gated GEMM of small batches (engine.cpp: gated_gemm): the GEMM wrote `splits` fp32 partial products
// [M, N] (slab stride `stride`); x += gate * ((p0 + p1 + ...) + bias), the partials summed in slab order (deterministic), the
// expression of the gated read-modify-write epilogue (latte.py:179-180).  16 bytes per lane, N % 4 == 0.
__global__ void __launch_bounds__(256) gated_split_reduce_kernel(float* __restrict__ x, const float* __restrict__ ws, int splits,
                                                                 size_t stride, const float* __restrict__ bias,
                                                                 const float* __restrict__ gate, int gate_stride, int rps, int M,
                                                                 int N) {
  const int n4 = N >> 2;
  const size_t total = (size_t)M * n4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / n4), n = (int)(i - (size_t)m * n4) * 4;
    const size_t o = (size_t)m * N + n;
    float4 a = *(const float4*)(ws + o);
    for (int k = 1; k < splits; ++k) {
      const float4 b = *(const float4*)(ws + (size_t)k * stride + o);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const float4 b4 = *(const float4*)(bias + n);
    const float4 g4 = *(const float4*)(gate + (size_t)(m / rps) * gate_stride + n);
    float4 r = *(const float4*)(x + o);
    r.x += g4.x * (a.x + b4.x); r.y += g4.y * (a.y + b4.y); r.z += g4.z * (a.z + b4.z); r.w += g4.w * (a.w + b4.w);
    *(float4*)(x + o) = r;
  }
}

__global__ void silu_rows_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = silu(in[i]);
}

__global__ void adaln_single_kernel(const float* __restrict__ tables, const float* __restrict__ head_table,
                                    const float* __restrict__ t6, const float* __restrict__ temb, float* __restrict__ mod, int B,
                                    int nblk, int D) {
  const size_t per = (size_t)(6 * nblk + 2) * D, total = (size_t)B * per;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / per, r = i % per;
    const int row = (int)(r / D), d = (int)(r % D);
    float v;
    if (row < 6 * nblk) v = tables[(size_t)row * D + d] + t6[b * 6 * D + (size_t)(row % 6) * D + d];
    else v = head_table[(size_t)(row - 6 * nblk) * D + d] + temb[b * D + d];
    mod[i] = v;
  }
}

__global__ void permute_cf_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int F, int hw, int to_bfc) {
  const size_t total = (size_t)B * C * F * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t q = i % hw, r = i / hw;
    size_t b, c, f;
    if (to_bfc) { c = r % C; f = (r / C) % F; b = r / ((size_t)C * F); out[i] = in[((b * C + c) * F + f) * hw + q]; }
    else        { f = r % F; c = (r / F) % C; b = r / ((size_t)C * F); out[i] = in[((b * F + f) * C + c) * hw + q]; }
  }
}

__global__ void mask_bias_kernel(const float* __restrict__ mask, float* __restrict__ bias, size_t n) {
  // latte_t2v.py:746-747: (1 - mask) * -10000.0, added to the cross-attention scores
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    bias[i] = (1.0f - mask[i]) * -10000.0f;
}

__global__ void fill_f32_kernel(float* p, float v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void iota_kernel(int64_t* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

__global__ void cfg_combine_kernel(float* out, int half_batch, int F, int Cout, int HW, float s) {
  // eps channels are [0, 4) (hard-coded 4 in the reference, latte.py:394)
  const size_t per_sample = (size_t)F * 4 * HW;
  const size_t total = (size_t)half_batch * per_sample;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / per_sample, r = i % per_sample;
    const size_t f = r / (4 * (size_t)HW), q = r % (4 * (size_t)HW);
    const size_t oc = ((b * F + f) * Cout) * HW + q;
    const size_t ou = (((b + half_batch) * F + f) * Cout) * HW + q;
    const float cond = out[oc], unc = out[ou];
    const float h = unc + s * (cond - unc);
    out[oc] = h;
    out[ou] = h;
  }
}

__global__ void sampler_update_kernel(SamplerCoefs c, const float* __restrict__ x, const float* __restrict__ mo,
                                      const float* __restrict__ noise, int batch, int frames, int C, int hw,
                                      int raw_cfg, float* sample_out, float* x0_out, const float* __restrict__ x0_in,
                                      const float* __restrict__ grad, int predict_only) {
#pragma clang fp contract(off)
  // op-for-op the fp32 tensor arithmetic of gaussian_diffusion.py (no FMA contraction)
  const size_t chw = (size_t)C * hw;
  const size_t total = (size_t)batch * frames * chw;
  const int hb = batch >> 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t bf = i / chw, r = i % chw;
    const size_t b = bf / frames, f = bf % frames;
    const int Cm = c.var_type == 0 ? 2 * C : C;     // channels of the model output: eps | v, or eps / x_start alone
    const size_t o = ((b * frames + f) * Cm) * hw + r;
    float eps;
    if (raw_cfg && r < (size_t)4 * hw) {
      const size_t bc = b % hb;
      const float cond = mo[((bc * frames + f) * Cm) * hw + r];
      const float unc = mo[(((bc + hb) * frames + f) * Cm) * hw + r];
      eps = unc + c.cfg_scale * (cond - unc);
    } else {
      eps = mo[o];
    }
    const float v = c.var_type == 0 ? mo[o + chw] : 0.0f;
    const float xv = x[i];
    float x0 = c.sqrt_recip * xv - c.sqrt_recipm1 * eps;               // gd:338-343
    if (c.mean_type == 1) x0 = eps;                                    // START_X: the model output is x_start (gd:323-324)
    if (predict_only) {                                                // the caller applies its denoised_fn to this
      x0_out[i] = x0;
      continue;
    }
    if (x0_in != nullptr) x0 = x0_in[i];                               // process_xstart: denoised_fn, then the clamp (gd:316-321)
    if (c.clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    float s;
    if (c.method == LATTE_METHOD_DDPM) {
      const float frac = (v + 1.0f) / 2.0f;                            // gd:295
      float log_var = frac * c.max_log + (1.0f - frac) * c.min_log;
      if (c.var_type != 0) log_var = c.fixed_log_var;                  // gd:298-313
      float mean = c.coef1 * x0 + c.coef2 * xv;                        // gd:232-241
      if (grad != nullptr) mean = mean + expf(log_var) * grad[i];      // condition_mean, gd:354-355 (variance = exp(log_var), gd:297)
      const float nz = noise != nullptr ? noise[i] : 0.0f;
      s = mean + (c.nonzero * expf(0.5f * log_var)) * nz;              // gd:420
    } else {
      if (grad != nullptr) {                                           // condition_score, gd:366-373
        float e = (c.sqrt_recip * xv - x0) / c.sqrt_recipm1;
        e = e - c.sqrt_one_minus_ab * grad[i];
        x0 = c.sqrt_recip * xv - c.sqrt_recipm1 * e;
      }
      const float e2 = (c.sqrt_recip * xv - x0) / c.sqrt_recipm1;      // gd:545
      const float mean_pred = x0 * c.sqrt_ab_prev + c.dir_coef * e2;   // gd:556-559
      s = mean_pred;
      if (noise != nullptr) s = mean_pred + (c.nonzero * c.sigma) * noise[i];
    }
    sample_out[i] = s;
    if (x0_out != nullptr) x0_out[i] = x0;
  }
}


// ------------------------------------------------------------------------------------------------
// Training path, forward evaluation (gaussian_diffusion.py:216-229 q_sample; :686-717 _vb_terms_bpd; :719-795
// training_losses; diffusion_utils.py:10-88).  One timestep PER SAMPLE: coefficients are gathered from the fp32 device
// copies of the fp64 tables (= _extract_into_tensor's from_numpy(arr)[t].float()), the arithmetic is the reference's fp32
// tensor arithmetic op for op (no FMA contraction).
__global__ void q_sample_kernel(const float* __restrict__ tab, int n_steps, const float* __restrict__ x0,
                                const float* __restrict__ noise, const int64_t* __restrict__ t, size_t per_sample,
                                size_t total, float* __restrict__ xt) {
#pragma clang fp contract(off)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ti = (int)t[i / per_sample];
    const float a = tab[DT_SQRT_AC * n_steps + ti], b = tab[DT_SQRT_1MAC * n_steps + ti];
    xt[i] = a * x0[i] + b * noise[i];                                  // gd:226-229
  }
}

__device__ __forceinline__ float approx_std_normal_cdf(float x) {      // diffusion_utils.py:39-44
#pragma clang fp contract(off)
  const float c = 0.7978845608028654f;                                 // np.sqrt(2.0 / np.pi) as fp32
  return 0.5f * (1.0f + tanhf(c * (x + 0.044715f * (x * x * x))));
}

constexpr int TT_THREADS = 256;
__global__ void __launch_bounds__(TT_THREADS) training_terms_kernel(const float* __restrict__ tab, int n_steps, int mean_type,
                                                                    int var_type, const float* __restrict__ x0,
                                                                    const float* __restrict__ xt, const float* __restrict__ noise,
                                                                    const float* __restrict__ mo, const int64_t* __restrict__ t,
                                                                    int frames, int C, int hw, float* __restrict__ partial) {
#pragma clang fp contract(off)
  const int b = blockIdx.y;
  const int ti = (int)t[b];
  const float coef1 = tab[DT_COEF1 * n_steps + ti], coef2 = tab[DT_COEF2 * n_steps + ti];
  const float plv = tab[DT_POST_LOGVAR * n_steps + ti], lb = tab[DT_LOG_BETAS * n_steps + ti];
  const float srec = tab[DT_SQRT_RECIP * n_steps + ti], srecm1 = tab[DT_SQRT_RECIPM1 * n_steps + ti];
  const float flv = tab[DT_FIXED_LOGVAR * n_steps + ti];
  const size_t chw = (size_t)C * hw, per = (size_t)frames * chw;
  const int Cm = var_type == 0 ? 2 * C : C;
  float s_mse = 0.f, s_vb = 0.f;
  for (size_t e = (size_t)blockIdx.x * TT_THREADS + threadIdx.x; e < per; e += (size_t)gridDim.x * TT_THREADS) {
    const size_t f = e / chw, r = e % chw;
    const size_t i = (size_t)b * per + e;
    const size_t o = (((size_t)b * frames + f) * Cm) * hw + r;
    const float pred = mo[o];
    const float xs = x0[i], xv = xt[i];
    const float target = mean_type == 1 ? xs : noise[i];               // gd:776-782
    const float d = target - pred;
    s_mse += d * d;
    // q(x_{t-1} | x_t, x_0), gd:232-241
    const float true_mean = coef1 * xs + coef2 * xv;
    // p_mean_variance with clip_denoised = False, gd:289-336
    float lv;
    if (var_type == 0) {
      const float v = mo[o + chw];
      const float frac = (v + 1.0f) / 2.0f;
      lv = frac * lb + (1.0f - frac) * plv;
    } else {
      lv = flv;
    }
    const float x0p = mean_type == 1 ? pred : srec * xv - srecm1 * pred;
    const float mean = coef1 * x0p + coef2 * xv;
    float term;
    if (ti != 0) {                                                     // KL(q || p), diffusion_utils.py:29-36
      const float dm = true_mean - mean;
      term = 0.5f * (-1.0f + lv - plv + expf(plv - lv) + (dm * dm) * expf(-lv));
    } else {                                                           // decoder NLL, diffusion_utils.py:62-88
      const float centered = xs - mean;
      const float inv_stdv = expf(-(0.5f * lv));
      const float cdf_plus = approx_std_normal_cdf(inv_stdv * (centered + 0.00392156862745098f));
      const float cdf_min = approx_std_normal_cdf(inv_stdv * (centered - 0.00392156862745098f));
      const float log_cdf_plus = logf(fmaxf(cdf_plus, 1e-12f));
      const float log_one_minus = logf(fmaxf(1.0f - cdf_min, 1e-12f));
      const float log_delta = logf(fmaxf(cdf_plus - cdf_min, 1e-12f));
      const float lp = xs < -0.999f ? log_cdf_plus : (xs > 0.999f ? log_one_minus : log_delta);
      term = -lp;
    }
    s_vb += term;
  }
  // deterministic block reduction (fixed tree), then one slot per (sample, block)
  __shared__ float red[2][TT_THREADS];
  red[0][threadIdx.x] = s_mse;
  red[1][threadIdx.x] = s_vb;
  __syncthreads();
  for (int w = TT_THREADS / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[((size_t)b * gridDim.x + blockIdx.x) * 2] = red[0][0];
    partial[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1] = red[1][0];
  }
}

__global__ void training_finalize_kernel(const float* __restrict__ partial, int blocks, double per_sample, float* __restrict__ mse,
                                         float* __restrict__ vb) {
  const int b = blockIdx.x;
  double a = 0.0, c = 0.0;
  for (int k = 0; k < blocks; ++k) {
    a += (double)partial[((size_t)b * blocks + k) * 2];
    c += (double)partial[((size_t)b * blocks + k) * 2 + 1];
  }
  mse[b] = (float)(a / per_sample);                                    // mean_flat
  vb[b] = (float)(c / per_sample) / 0.6931471805599453f;               // mean_flat(.) / np.log(2.0), fp32 division
}

__global__ void t2v_guided_ddim_kernel(float* __restrict__ x, const float* __restrict__ mo, int b, int C, int Cout, int F, int hw,
                                       float scale, float c1, float c2, float c3, float c4) {
#pragma clang fp contract(off)
  const size_t total = (size_t)b * C * F * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i % hw, f = (i / hw) % F, c = (i / ((size_t)hw * F)) % C, bb = i / ((size_t)hw * F * C);
    const float un = mo[(((bb * F + f) * Cout) + c) * hw + r];                 // negative-prompt half
    const float tx = mo[((((bb + b) * F + f) * Cout) + c) * hw + r];           // prompt half
    const float eps = un + scale * (tx - un);                                  // pipeline_latte.py:748-749
    const float xv = x[i];
    const float x0 = (xv - c1 * eps) / c2;
    x[i] = c3 * x0 + c4 * eps;
  }
}

__global__ void training_combine_kernel(const float* __restrict__ mse, const float* __restrict__ vb, int has_vb, int kl_only,
                                        float vb_scale, int batch, float* mse_out, float* vb_out, float* loss_out) {
#pragma clang fp contract(off)
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const float v = has_vb ? vb[b] * vb_scale : 0.0f;                    // gd:751-752 / :771-774 (no scaling = * 1)
  if (kl_only) {
    loss_out[b] = v;
    return;
  }
  if (mse_out) mse_out[b] = mse[b];
  if (vb_out && has_vb) vb_out[b] = v;
  loss_out[b] = has_vb ? mse[b] + v : mse[b];                          // gd:788-791
}

template <int DT>
__global__ void convert_kernel(const float* __restrict__ in, half_t* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if constexpr (DT == LATTE_DTYPE_BF16) {
      const __bf16 h = (__bf16)in[i];
      out[i] = __builtin_bit_cast(half_t, h);
    } else {
      const _Float16 h = (_Float16)in[i];
      out[i] = __builtin_bit_cast(half_t, h);
    }
  }
}

// fp32 -> nearest f16 and the f16 of the rounding residual (the split operand of the VAE's shortcut / upsampler products)
__global__ void convert_split_kernel(const float* __restrict__ in, half_t* __restrict__ out, half_t* __restrict__ out_lo, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = in[i];
    const _Float16 h = (_Float16)v;
    out[i] = __builtin_bit_cast(half_t, h);
    out_lo[i] = __builtin_bit_cast(half_t, (_Float16)(v - (float)h));
  }
}

template <int DT>
__global__ void widen_kernel(const half_t* __restrict__ in, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if constexpr (DT == LATTE_DTYPE_BF16) out[i] = __builtin_bit_cast(float, (unsigned int)in[i] << 16);
    else out[i] = (float)__builtin_bit_cast(_Float16, in[i]);
  }
}

// W4 of the GEMMs' fp4 correction pass (common.h: GemmArgs::W4): one wave per weight row -- row maximum -> E8M0 scale, e2m1 codes
__global__ void __launch_bounds__(256) pack_w4_kernel(const half_t* __restrict__ in, unsigned char* __restrict__ out4,
                                                      unsigned char* __restrict__ out_scale, int N, int K, int pitch) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const half_t* r = in + (size_t)row * K;
  float amax = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const uint2 q = *(const uint2*)(r + k);
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_;
    const f16x4_ hv = __builtin_bit_cast(f16x4_, q);
    const float a = (float)hv[0], b = (float)hv[1], c = (float)hv[2], d = (float)hv[3];
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const int e = quant4_exponent(amax);
  const float sc2 = __builtin_ldexpf(1.0f, e);
  unsigned short* o4 = (unsigned short*)(out4 + (size_t)row * pitch);
  for (int k = lane * 4; k < pitch * 2; k += 256) {
    unsigned short code = 0;
    if (k < K) {
      const uint2 q = *(const uint2*)(r + k);
      typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_;
      const f16x4_ hv = __builtin_bit_cast(f16x4_, q);
      const float a = (float)hv[0], b = (float)hv[1], c = (float)hv[2], d = (float)hv[3];
      code = (unsigned short)quant4_pk4(a, b, c, d, sc2);
    }
    o4[k >> 2] = code;
  }
  if (lane == 0) out_scale[row] = (unsigned char)(e + 127);
}

// W8 of the GEMMs' fp8 correction pass (common.h: LO8_W_SHIFT): four f16 weights -> four OCP e4m3 codes of w * 2^LO8_W_SHIFT, clamped
__global__ void pack_w8_kernel(const half_t* __restrict__ in, unsigned char* __restrict__ out, size_t n4) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
  constexpr float S = (float)(1 << LO8_W_SHIFT);
  auto cl = [](float r) { return __builtin_fminf(__builtin_fmaxf(r, -448.f), 448.f); };
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const u32x2_t v = ((const u32x2_t*)in)[i];
    // (scalar copies first: bit-casting the vector elements directly is miscompiled into re-using element 0, gemm.hip)
    const unsigned int lo = v[0], hi = v[1];
    const float w0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(lo & 0xffffu)), w1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(lo >> 16));
    const float w2 = (float)__builtin_bit_cast(_Float16, (unsigned short)(hi & 0xffffu)), w3 = (float)__builtin_bit_cast(_Float16, (unsigned short)(hi >> 16));
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(w0 * S), cl(w1 * S), 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(w2 * S), cl(w3 * S), w, true);
    ((unsigned int*)out)[i] = (unsigned int)w;
  }
}

__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  const size_t total = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / cols, c = i % cols;
    out[c * rows + r] = in[i];
  }
}

// Philox4x32-10 (Salmon et al. 2011): counter = element-quad index, key = seed.
__device__ __forceinline__ void philox_round(unsigned int (&c)[4], unsigned int k0, unsigned int k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
  const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k0;
  const unsigned int n1 = (unsigned int)p1;
  const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k1;
  const unsigned int n3 = (unsigned int)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ void fill_normal_kernel(float* __restrict__ out, size_t n, unsigned long long seed, unsigned long long offset) {
  const size_t quads = (n + 3) / 4;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
    // quad q covers elements [4q, 4q+4) of THIS call; the counter is the global element-quad index
    const unsigned long long ctr = (offset >> 2) + q;
    unsigned int c[4] = {(unsigned int)ctr, (unsigned int)(ctr >> 32), (unsigned int)(offset & 3), 0u};
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float r = sqrtf(-2.0f * logf(u1));
      float sn, cs;
      sincosf(6.283185307179586f * u2, &sn, &cs);
      z[2 * h] = r * cs;
      z[2 * h + 1] = r * sn;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * q + j < n) out[4 * q + j] = z[j];
  }
}

inline int grid_for(size_t n, int block) {
  size_t g = (n + block - 1) / block;
  return (int)(g > 4096 ? 4096 : (g == 0 ? 1 : g));
}

}  // namespace

#define LATTE_NCH_SWITCH(D, MACRO)                                                      \
  switch ((D) / 128) {                                                                  \
    case 1: MACRO(1); break;                                                            \
    case 2: MACRO(2); break;                                                            \
    case 3: MACRO(3); break;                                                            \
    case 4: MACRO(4); break;                                                            \
    case 6: MACRO(6); break;                                                            \
    case 8: MACRO(8); break;                                                            \
    case 9: MACRO(9); break;                                                            \
    default: return fail(LATTE_ERR_INVALID, "hidden_size must be 128*{1,2,3,4,6,8,9}"); \
  }

int launch_q_sample(const float* tables, int n_steps, const float* x_start, const float* noise, const int64_t* t, int batch,
                    size_t per_sample, float* x_t, hipStream_t st) {
  const size_t total = (size_t)batch * per_sample;
  hipLaunchKernelGGL(q_sample_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, tables, n_steps, x_start, noise, t, per_sample,
                     total, x_t);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_t2v_guided_ddim(float* x, const float* model_out, int b, int C, int Cout, int F, int hw, float scale, float c1, float c2,
                           float c3, float c4, hipStream_t st) {
  const size_t total = (size_t)b * C * F * hw;
  hipLaunchKernelGGL(t2v_guided_ddim_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, x, model_out, b, C, Cout, F, hw, scale, c1,
                     c2, c3, c4);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_training_combine(const float* mse, const float* vb, int has_vb, int kl_only, float vb_scale, int batch, float* mse_out,
                            float* vb_out, float* loss_out, hipStream_t st) {
  hipLaunchKernelGGL(training_combine_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, mse, vb, has_vb, kl_only, vb_scale, batch,
                     mse_out, vb_out, loss_out);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int training_terms_blocks(size_t per_sample) {
  size_t b = (per_sample + (size_t)TT_THREADS * 8 - 1) / ((size_t)TT_THREADS * 8);
  return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

int launch_training_terms(const float* tables, int n_steps, int mean_type, int var_type, const float* x_start, const float* x_t,
                          const float* noise, const float* model_out, const int64_t* t, int batch, int frames, int channels, int hw,
                          float* partial, int blocks_per_sample, float* mse, float* vb, hipStream_t st) {
  hipLaunchKernelGGL(training_terms_kernel, dim3(blocks_per_sample, batch), dim3(TT_THREADS), 0, st, tables, n_steps, mean_type,
                     var_type, x_start, x_t, noise, model_out, t, frames, channels, hw, partial);
  hipLaunchKernelGGL(training_finalize_kernel, dim3(batch), dim3(1), 0, st, partial, blocks_per_sample,
                     (double)frames * channels * hw, mse, vb);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_ln_modulate_split4(const float* x_in, half_t* y, unsigned char* y4, unsigned char* y4s, const float* shift, const float* scale,
                              int mod_stride, int M, int D, int rows_per_sample, int dtype, hipStream_t st) {
  if (D % 128 != 0 || !y4 || !y4s || dtype != LATTE_DTYPE_F16)
    return fail(LATTE_ERR_INVALID, "ln_modulate: the fp4-remainder output is f16 only, D % 128 == 0, and needs its two outputs");
  dim3 grid((M + 3) / 4), block(256);
#define LN_LAUNCH_SPLIT4(NCH)                                                                                          \
  hipLaunchKernelGGL((ln_modulate_kernel<NCH, LATTE_DTYPE_F16, false, 3>), grid, block, 0, st, x_in, (float*)nullptr, y, shift, scale, \
                     mod_stride, M, rows_per_sample, (const float*)nullptr, 1, 1, y4, y4s)
  LATTE_NCH_SWITCH(D, LN_LAUNCH_SPLIT4)
#undef LN_LAUNCH_SPLIT4
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_ln_modulate(const float* x_in, float* x_rw, half_t* y, const float* shift, const float* scale,
                       int mod_stride, int M, int D, int rows_per_sample, const float* temp_embed, int T, int F,
                       int dtype, hipStream_t st, int split, unsigned char* y8) {
  if (D % 128 != 0) return fail(LATTE_ERR_INVALID, "ln_modulate: D % 128 != 0");
  dim3 grid((M + 3) / 4), block(256);
  if (split == 3) {   // [M, D] f16 + [M, lo4_pitch(D)] fp4 remainder + [M] row scales (y8 = codes, the scales behind them: y8s)
    return fail(LATTE_ERR_INVALID, "ln_modulate: the fp4-remainder output goes through launch_ln_modulate_split4");
  }
  if (split == 2) {   // [M, D] f16 + [M, D] fp8 remainder (the fp8 correction operand of the GEMM behind it)
    if (temp_embed || !y8 || dtype != LATTE_DTYPE_F16)
      return fail(LATTE_ERR_INVALID, "ln_modulate: the fp8-remainder output is f16 only, needs y8 and has no temp_embed form");
#define LN_LAUNCH_SPLIT8(NCH)                                                                                          \
  hipLaunchKernelGGL((ln_modulate_kernel<NCH, LATTE_DTYPE_F16, false, 2>), grid, block, 0, st, x_in, x_rw, y, shift, scale, \
                     mod_stride, M, rows_per_sample, temp_embed, T, F, y8)
    LATTE_NCH_SWITCH(D, LN_LAUNCH_SPLIT8)
#undef LN_LAUNCH_SPLIT8
    LATTE_HIP(hipGetLastError());
    return LATTE_OK;
  }
  if (split) {   // [M, 2 D] split-operand output (no temp_embed form: the LayerNorm in front of fc1 never adds it)
    if (temp_embed) return fail(LATTE_ERR_INVALID, "ln_modulate: the split-operand output has no temp_embed form");
#define LN_LAUNCH_SPLIT(NCH)                                                                                   \
  do {                                                                                                         \
    if (dtype == LATTE_DTYPE_BF16)                                                                             \
      hipLaunchKernelGGL((ln_modulate_kernel<NCH, LATTE_DTYPE_BF16, false, 1>), grid, block, 0, st, x_in, x_rw, y, \
                         shift, scale, mod_stride, M, rows_per_sample, temp_embed, T, F);                      \
    else                                                                                                       \
      hipLaunchKernelGGL((ln_modulate_kernel<NCH, LATTE_DTYPE_F16, false, 1>), grid, block, 0, st, x_in, x_rw, y, \
                         shift, scale, mod_stride, M, rows_per_sample, temp_embed, T, F);                      \
  } while (0)
    LATTE_NCH_SWITCH(D, LN_LAUNCH_SPLIT)
#undef LN_LAUNCH_SPLIT
    LATTE_HIP(hipGetLastError());
    return LATTE_OK;
  }
#define LN_LAUNCH(NCH)                                                                                         \
  do {                                                                                                         \
    if (dtype == LATTE_DTYPE_BF16) {                                                                           \
      if (temp_embed)                                                                                          \
        hipLaunchKernelGGL((ln_modulate_kernel<NCH, LATTE_DTYPE_BF16, true>), grid, block, 0, st, x_in, x_rw, y, \
                           shift, scale, mod_stride, M, rows_per_sample, temp_embed, T, F);                    \
      else                                                                                                     \
        hipLaunchKernelGGL((ln_modulate_kernel<NCH, LATTE_DTYPE_BF16, false>), grid, block, 0, st, x_in, x_rw, y, \
                           shift, scale, mod_stride, M, rows_per_sample, temp_embed, T, F);                    \
    } else {                                                                                                   \
      if (temp_embed)                                                                                          \
        hipLaunchKernelGGL((ln_modulate_kernel<NCH, LATTE_DTYPE_F16, true>), grid, block, 0, st, x_in, x_rw, y, \
                           shift, scale, mod_stride, M, rows_per_sample, temp_embed, T, F);                    \
      else                                                                                                     \
        hipLaunchKernelGGL((ln_modulate_kernel<NCH, LATTE_DTYPE_F16, false>), grid, block, 0, st, x_in, x_rw, y, \
                           shift, scale, mod_stride, M, rows_per_sample, temp_embed, T, F);                    \
    }                                                                                                          \
  } while (0)
  LATTE_NCH_SWITCH(D, LN_LAUNCH)
#undef LN_LAUNCH
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_small_linear(int in_mode, const float* in, const int64_t* t, const float* W, const float* bias,
                        const float* add_table, const int64_t* add_idx, float* out, int B, int N, int K,
                        int out_stride, hipStream_t st) {
  if (K % 128 != 0 || K > 128 * SL_MAXCH) return fail(LATTE_ERR_INVALID, "small_linear: need K % 128 == 0 and K <= 1152");
  dim3 grid((N + 3) / 4), block(256);
  switch (in_mode) {
    case IN_PLAIN:
      hipLaunchKernelGGL(small_linear_kernel<IN_PLAIN>, grid, block, 0, st, in, t, W, bias, add_table, add_idx, out, B, N, K, out_stride);
      break;
    case IN_SILU:
      hipLaunchKernelGGL(small_linear_kernel<IN_SILU>, grid, block, 0, st, in, t, W, bias, add_table, add_idx, out, B, N, K, out_stride);
      break;
    case IN_TFREQ:
      hipLaunchKernelGGL(small_linear_kernel<IN_TFREQ>, grid, block, 0, st, in, t, W, bias, add_table, add_idx, out, B, N, K, out_stride);
      break;
    default:
      return fail(LATTE_ERR_INVALID, "small_linear: bad input mode");
  }
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_patch_embed(const float* x, const float* Wt, const float* bias, const float* pos, float* out, int BF,
                       int C, int H, int p, int D, hipStream_t st) {
  const int G = H / p, ntok = BF * G * G, K = C * p * p;
  dim3 grid((ntok + PE_TOK - 1) / PE_TOK, D / 128), block(128);
  hipLaunchKernelGGL(patch_embed_kernel, grid, block, PE_TOK * K * sizeof(float), st, x, Wt, bias, pos, out, ntok, C, H, p, D);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_final_layer(const float* x, const float* shift, const float* scale, int mod_stride, const float* Wt,
                       const float* bias, float* out, int M, int D, int rows_per_sample, int T, int p, int Cout,
                       int H, hipStream_t st) {
  dim3 grid((M + FL_ROWS - 1) / FL_ROWS), block(256);
  const size_t lds = (size_t)FL_ROWS * D * sizeof(float);
#define FL_LAUNCH(NCH)                                                                                       \
  hipLaunchKernelGGL(final_layer_kernel<NCH>, grid, block, lds, st, x, shift, scale, mod_stride, Wt, bias, out, M, \
                     rows_per_sample, T, p, Cout, H)
  LATTE_NCH_SWITCH(D, FL_LAUNCH)
#undef FL_LAUNCH
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_cond_rows(const float* temb, const float* ytab, const int64_t* y, float* out, int n_steps, int bu, int D,
                     hipStream_t st) {
  const size_t n = (size_t)n_steps * bu * D;
  hipLaunchKernelGGL(cond_rows_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, temb, ytab, y, out, n_steps, bu, D);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_text_proj(const float* text, const float* W, const float* bias, float* out, int B, int N, int K,
                     hipStream_t st) {
  if (K % 128) return fail(LATTE_ERR_INVALID, "text_proj: K must be a multiple of 128");
  hipLaunchKernelGGL(text_proj_kernel, dim3((N + 3) / 4), dim3(256), 0, st, text, W, bias, out, B, N, K);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_gated_split_reduce(float* x, const float* ws, int splits, size_t stride, const float* bias, const float* gate,
                              int gate_stride, int rows_per_sample, int M, int N, hipStream_t st) {
  if (N % 4 || splits < 1) return fail(LATTE_ERR_INVALID, "gated_split_reduce: need N % 4 == 0, splits >= 1");
  const size_t n = (size_t)M * (N / 4);
  hipLaunchKernelGGL(gated_split_reduce_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, x, ws, splits, stride, bias, gate,
                     gate_stride, rows_per_sample, M, N);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_silu_rows(const float* in, float* out, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(silu_rows_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, in, out, n);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_adaln_single(const float* tables, const float* head_table, const float* t6, const float* temb, float* mod, int B,
                        int nblk, int D, hipStream_t st) {
  const size_t n = (size_t)B * (6 * nblk + 2) * D;
  hipLaunchKernelGGL(adaln_single_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, tables, head_table, t6, temb, mod, B, nblk, D);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_permute_cf(const float* in, float* out, int B, int C, int F, int hw, int to_bfc, hipStream_t st) {
  const size_t n = (size_t)B * C * F * hw;
  hipLaunchKernelGGL(permute_cf_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, in, out, B, C, F, hw, to_bfc);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_mask_bias(const float* mask, float* bias, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(mask_bias_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, mask, bias, n);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

__global__ void scale_f32_kernel(float* __restrict__ p, float s, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] *= s;
}

int launch_scale_f32(float* p, float s, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(scale_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, p, s, n);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

// p[i] *= *s or p[i] /= *s with the factor read from DEVICE memory (the trainer's loss scale, which its optimiser step may halve
// or double without a host round trip); *s is a power of two, so the division is exact
__global__ void scale_f32_dev_kernel(float* __restrict__ p, const float* __restrict__ s, int inverse, size_t n) {
  const float f = inverse ? 1.0f / s[0] : s[0];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] *= f;
}

int launch_scale_f32_dev(float* p, const float* s_dev, int inverse, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(scale_f32_dev_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, p, s_dev, inverse, n);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_fill_f32(float* p, float v, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, p, v, n);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_iota(int64_t* p, int n, hipStream_t st) {
  hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p, n);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_cfg_combine(float* out, int half_batch, int F, int Cout, int HW, float cfg_scale, hipStream_t st) {
  const size_t n = (size_t)half_batch * F * 4 * HW;
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, out, half_batch, F, Cout, HW, cfg_scale);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_sampler_update(const SamplerCoefs& c, const float* x, const float* model_out, const float* noise,
                          int batch, int frames, int channels, int hw, int raw_cfg, float* sample_out, float* x0_out,
                          hipStream_t st, const float* x0_in, const float* grad, int predict_only) {
  const size_t n = (size_t)batch * frames * channels * hw;
  hipLaunchKernelGGL(sampler_update_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, c, x, model_out, noise, batch,
                     frames, channels, hw, raw_cfg, sample_out, x0_out, x0_in, grad, predict_only);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_convert_f32_to_h16(const float* in, half_t* out, int64_t n, int dtype, hipStream_t st) {
  if (dtype == LATTE_DTYPE_BF16)
    hipLaunchKernelGGL(convert_kernel<LATTE_DTYPE_BF16>, dim3(grid_for(n, 256)), dim3(256), 0, st, in, out, (size_t)n);
  else
    hipLaunchKernelGGL(convert_kernel<LATTE_DTYPE_F16>, dim3(grid_for(n, 256)), dim3(256), 0, st, in, out, (size_t)n);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_convert_f32_to_h16_split(const float* in, half_t* out, half_t* out_lo, int64_t n, int dtype, hipStream_t st) {
  if (dtype != LATTE_DTYPE_F16) return fail(LATTE_ERR_INVALID, "convert_split: f16 only");
  hipLaunchKernelGGL(convert_split_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, in, out, out_lo, (size_t)n);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_convert_h16_to_f32(const half_t* in, float* out, int64_t n, int dtype, hipStream_t st) {
  if (dtype == LATTE_DTYPE_BF16)
    hipLaunchKernelGGL(widen_kernel<LATTE_DTYPE_BF16>, dim3(grid_for(n, 256)), dim3(256), 0, st, in, out, (size_t)n);
  else
    hipLaunchKernelGGL(widen_kernel<LATTE_DTYPE_F16>, dim3(grid_for(n, 256)), dim3(256), 0, st, in, out, (size_t)n);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_pack_w4(const half_t* in, unsigned char* out4, unsigned char* out_scale, int N, int K, int dtype, hipStream_t st) {
  if (dtype != LATTE_DTYPE_F16 || K % 4) return fail(LATTE_ERR_INVALID, "pack_w4: f16 weights, K % 4 == 0");
  hipLaunchKernelGGL(pack_w4_kernel, dim3((N + 3) / 4), dim3(256), 0, st, in, out4, out_scale, N, K, lo4_pitch(K));
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_pack_w8(const half_t* in, unsigned char* out, int64_t n, int dtype, hipStream_t st) {
  if (dtype != LATTE_DTYPE_F16 || n % 4) return fail(LATTE_ERR_INVALID, "pack_w8: f16 weights, n % 4 == 0");
  hipLaunchKernelGGL(pack_w8_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, st, in, out, (size_t)(n / 4));
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_fill_normal(float* out, size_t n, uint64_t seed, uint64_t offset, hipStream_t st) {
  hipLaunchKernelGGL(fill_normal_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, st, out, n,
                     (unsigned long long)seed, (unsigned long long)offset);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

int launch_transpose_f32(const float* in, float* out, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3(grid_for((size_t)rows * cols, 256)), dim3(256), 0, st, in, out, rows, cols);
  LATTE_HIP(hipGetLastError());
  return LATTE_OK;
}

}  // namespace latte
