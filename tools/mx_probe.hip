// Measurement / documentation tool (not part of the library): the operand and scale layouts of gfx950's block-scaled MFMA
//   v_mfma_scale_f32_16x16x128_f8f6f4   D[16x16] += (A[16x128] * 2^(sa - 127)) . (B[16x128] * 2^(sb - 127))^T
// established ON THE GPU against a CPU emulation of the OCP MX formats (this file is the documentation the round-5 review asked
// for: the material at hand describes neither the lane -> element map of the 8-VGPR operands nor the scale operand).
//
// Part 1 (layout, one wave): for every format the instruction takes (cbsz / blgp: 0 = fp8 e4m3fn, 1 = bf8 e5m2, 2 = fp6 e2m3,
// 3 = bf6 e3m2, 4 = fp4 e2m1) random codes and random E8M0 scale bytes (one 32-bit scale register per lane, byte chosen by the
// instruction's op_sel) are multiplied on the GPU and on the CPU under three hypotheses (H0 / H1 / H2 at layout_case below), common part:
//   * lane l supplies row (l & 15) of its operand, 32 elements of it,
//   * element e of the lane's 32 sits at bit e * W of the lane's little-endian register string (W = 8 / 6 / 4 bits: fp8 uses all
//     8 VGPRs, fp6 the first 6, fp4 the first 4),
//   * a scale register's byte op_sel is an E8M0 scale 2^(byte - 127),
//   * D: lane l holds column (l & 15), rows 4 (l >> 4) ... + 3 of D = A . B^T  (the map of every 16x16 MFMA);
// they differ in WHICH 32 k a lane holds and WHOSE scale register scales an element (result: fp6 / fp4 = H0, fp8 / bf8 = H1).
// Codes are drawn so that every product and every partial sum is exact in fp32 (exponents in a narrow band, scales 2^0 .. 2^3):
// the GPU result must then equal the CPU's BIT FOR BIT whatever the hardware's summation order; a second draw with the full code
// range (subnormals, extreme exponents, no NaN / Inf codes) is compared against an fp64 sum at 2e-6 relative to sum |products|.
//
// Part 2 (rate under the socket's power cap, all CUs): what a LOW-PRECISION CORRECTION PASS costs beside an f16 GEMM main loop.
// 8 waves per CU, wave tile 128 x 48 (the 12-wave GEMM's consumer tile: 8 x 3 accumulators), per "K = 128" of the contraction:
//   hi      : 2 K-tiles of f16            = 96 v_mfma_f32_16x16x32_f16               (the plain operand)
//   hi+lo16 : 4 K-tiles of f16            = 192 of them                              (round 5's [hi | lo] . [W | W])
//   hi+lo8  : 2 K-tiles of f16 + 1 of fp8 = 96 + 24 v_mfma_scale_f32_16x16x128 fp8   (this round's candidate)
//   hi+lo4  : 2 K-tiles of f16 + 1 of fp4 = 96 + 24 ... fp4
// on random operands (f16 ~ N(0, 1); fp8 / fp4 codes uniform over the finite codes), fragments in registers, no memory traffic.
//   hipcc -O3 --offload-arch=gfx950 tools/mx_probe.hip -o /tmp/mx_probe && /tmp/mx_probe [seconds per case]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---------------------------------------------------------------- CPU side: OCP MX element formats
static double decode(int fmt, unsigned code) {
  int eb, mb, bias;
  switch (fmt) {
    case 0: eb = 4; mb = 3; bias = 7; break;    // fp8 e4m3fn
    case 1: eb = 5; mb = 2; bias = 15; break;   // bf8 e5m2
    case 2: eb = 2; mb = 3; bias = 1; break;    // fp6 e2m3
    case 3: eb = 3; mb = 2; bias = 3; break;    // bf6 e3m2
    default: eb = 2; mb = 1; bias = 1; break;   // fp4 e2m1
  }
  const unsigned m = code & ((1u << mb) - 1), e = (code >> mb) & ((1u << eb) - 1), s = (code >> (mb + eb)) & 1;
  double v = e == 0 ? std::ldexp((double)m, 1 - bias - mb) : std::ldexp((double)((1u << mb) | m), (int)e - bias - mb);
  return s ? -v : v;
}
static int width(int fmt) { return fmt <= 1 ? 8 : fmt <= 3 ? 6 : 4; }
static bool finite_code(int fmt, unsigned code) {
  if (fmt == 0) return (code & 0x7f) != 0x7f;          // e4m3fn: S.1111.111 = NaN, no infinities
  if (fmt == 1) return ((code >> 2) & 0x1f) != 0x1f;   // e5m2: exponent 31 = Inf / NaN
  return true;                                         // fp6 / fp4: every code is a number
}
static void put_bits(uint32_t* regs, int e, int w, unsigned code) {   // element e at bit e * w of the little-endian register string
  const int bit = e * w;
  regs[bit >> 5] |= code << (bit & 31);
  if ((bit & 31) + w > 32) regs[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
}

// ---------------------------------------------------------------- Part 1 kernel: one instruction, formats / op_sel as template values
template <int FA, int FB, int OA, int OB>
__global__ void one_mfma(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f4* out) {
  const int l = threadIdx.x;
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, FA, FB, OA, sa[l], OB, sb[l]);
  out[l] = c;
}

static uint32_t rng_state = 0x1234567u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

// Hypotheses for (lane, element e of the lane's 32) -> k and for which scale register scales a product term:
//   H0: k = 32 g + e (g = lane >> 4: one MX block per lane), the lane's own scale                       [holds for fp6 / fp4]
//   H1: k = 64 (e / 16) + 16 g + e % 16 (VGPRs 0-3 = the lane group's 16 bytes of K 0..63, VGPRs 4-7 = of K 64..127: the K = 64
//       instruction twice), MX block b = k / 32 of a row scaled by the scale register of lane (row, g = b)
//   H2: the k map of H1, every element scaled by its own lane's scale register
static int kmap(int hyp, int g, int e) { return hyp == 0 ? 32 * g + e : 64 * (e / 16) + 16 * g + e % 16; }

template <int FA, int FB, int OA, int OB>
static int layout_case(bool exact) {
  // register contents first (codes per lane and element, one scale byte per lane), the logical matrices follow from the hypothesis
  std::vector<unsigned> ca(64 * 32), cb(64 * 32), sca(64), scb(64);
  auto draw = [&](int fmt) -> unsigned {
    const int w = width(fmt);
    for (;;) {
      unsigned c = rnd() & ((1u << w) - 1);
      if (!finite_code(fmt, c)) continue;
      if (exact) {   // |value| in [1, 4) or zero: 2 - 4 significant bits, products exact, sums of 128 exact in fp32
        const double v = std::fabs(decode(fmt, c));
        if (!(v == 0.0 || (v >= 1.0 && v < 4.0))) continue;
      }
      return c;
    }
  };
  for (auto& c : ca) c = draw(FA);
  for (auto& c : cb) c = draw(FB);
  for (auto& s : sca) s = exact ? 127 + (rnd() & 3) : 127 - 20 + (rnd() % 41);
  for (auto& s : scb) s = exact ? 127 + (rnd() & 3) : 127 - 20 + (rnd() % 41);
  std::vector<uint32_t> ra(64 * 8, 0), rb(64 * 8, 0), rsa(64), rsb(64);
  for (int l = 0; l < 64; ++l) {
    for (int e = 0; e < 32; ++e) {
      put_bits(&ra[l * 8], e, width(FA), ca[l * 32 + e]);
      put_bits(&rb[l * 8], e, width(FB), cb[l * 32 + e]);
    }
    // the selected byte carries the scale, the other three bytes garbage that must be ignored
    uint32_t ga = rnd() | (rnd() << 24), gb = rnd() | (rnd() << 24);
    rsa[l] = (ga & ~(0xffu << (8 * OA))) | (sca[l] << (8 * OA));
    rsb[l] = (gb & ~(0xffu << (8 * OB))) | (scb[l] << (8 * OB));
  }
  i32x8 *da, *db;
  int *dsa, *dsb;
  f4* dout;
  CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dout, 64 * 16));
  CK(hipMemcpy(da, ra.data(), 64 * 32, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, rb.data(), 64 * 32, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsa, rsa.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsb, rsb.data(), 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL((one_mfma<FA, FB, OA, OB>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
  CK(hipDeviceSynchronize());
  std::vector<float> got(64 * 4);
  CK(hipMemcpy(got.data(), dout, 64 * 16, hipMemcpyDeviceToHost));
  CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dsa)); CK(hipFree(dsb)); CK(hipFree(dout));
  int matched = -1;
  printf("  A fmt %d (op_sel %d) x B fmt %d (op_sel %d), %s:", FA, OA, FB, OB, exact ? "exact codes, bit compare" : "full code range, vs fp64 ");
  for (int hyp = 0; hyp < 3; ++hyp) {
    // effective (scaled) logical operands under the hypothesis
    std::vector<double> ea(16 * 128), eb(16 * 128);
    for (int l = 0; l < 64; ++l) {
      const int row = l & 15, g = l >> 4;
      for (int e = 0; e < 32; ++e) {
        const int k = kmap(hyp, g, e);
        const int sl = hyp == 1 ? ((k / 32) << 4 | row) : l;   // the lane whose scale register scales this element
        ea[row * 128 + k] = decode(FA, ca[l * 32 + e]) * std::ldexp(1.0, (int)sca[sl] - 127);
        eb[row * 128 + k] = decode(FB, cb[l * 32 + e]) * std::ldexp(1.0, (int)scb[sl] - 127);
      }
    }
    int bad = 0;
    double worst = 0.0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * (l >> 4) + r, n = l & 15;   // lane l: column n = l & 15, rows 4 (l >> 4) + r
        double s = 0.0, sabs = 0.0;
        for (int k = 0; k < 128; ++k) {
          const double p = ea[m * 128 + k] * eb[n * 128 + k];
          s += p;
          sabs += std::fabs(p);
        }
        const float g_ = got[l * 4 + r];
        if (exact) {
          const float want = (float)s;
          if (std::memcmp(&g_, &want, 4) != 0) ++bad;
        } else {
          const double rel = std::fabs((double)g_ - s) / (sabs > 0 ? sabs : 1.0);
          if (rel > worst) worst = rel;
          if (rel > 2e-6) ++bad;
        }
      }
    printf("  H%d %s", hyp, bad ? "no" : "MATCH");
    if (!exact) printf(" (%.1e)", worst);
    if (!bad && matched < 0) matched = hyp;
  }
  printf("\n");
  return matched;
}

// ---------------------------------------------------------------- Part 2: rate under the cap
// MODE 0 hi, 1 hi + lo16, 2 hi + lo8, 3 hi + lo4, 4 lo8 only, 5 lo4 only
template <int MODE>
__global__ void __launch_bounds__(512) rate_probe(const h8* __restrict__ src, const i32x8* __restrict__ src8, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  h8 a[8], b[2][3];
  i32x8 a8[4], b8[3];
#pragma unroll
  for (int f = 0; f < 8; ++f) a[f] = src[(f * 8 + wave) * 64 + lane];
#pragma unroll
  for (int f = 0; f < 6; ++f) b[f / 3][f % 3] = src[((8 + f) * 8 + wave) * 64 + lane];
#pragma unroll
  for (int f = 0; f < 4; ++f) a8[f] = src8[(f * 8 + wave) * 64 + lane];
#pragma unroll
  for (int f = 0; f < 3; ++f) b8[f] = src8[((4 + f) * 8 + wave) * 64 + lane];
  f4 acc[8][3];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
  const int sc = 0x73737373 + (lane & 1);   // 2^-12, 2^-11
  for (int it = 0; it < iters; ++it) {
    constexpr int NT16 = MODE == 1 ? 4 : MODE >= 4 ? 0 : 2;
#pragma unroll
    for (int t = 0; t < NT16; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[ks][j], a[(i + t) & 7], acc[i][j], 0, 0, 0);
    if constexpr (MODE == 2 || MODE == 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8[j], a8[i & 3], acc[i][j], 0, 0, 0, sc, 0, sc);
    }
    if constexpr (MODE == 3 || MODE == 5) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8[j], a8[i & 3], acc[i][j], 4, 4, 0, sc, 0, sc);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
static void run_rate(const h8* src, const i32x8* src8, float* out, double seconds, const char* what, bool quiet) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((rate_probe<MODE>), dim3(256), dim3(512), 0, 0, src, src8, out, iters);
  CK(hipDeviceSynchronize());
  // settle under the power cap first, then time the second half of the interval
  double total_ms = 0.0, ms_per = 0.0;
  int launches = 0;
  while (total_ms < seconds * 1000.0) {
    CK(hipEventRecord(e0));
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL((rate_probe<MODE>), dim3(256), dim3(512), 0, 0, src, src8, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    total_ms += ms;
    ms_per = ms / 5.0;   // the last batch = the settled state
    ++launches;
  }
  // per iteration: K = 128 of a 128 x 48 wave tile = 2 * 128 * 48 * 128 FLOP of the plain product, 8 waves x 256 CUs
  const double flop_plain = 2.0 * 128 * 48 * 128 * 8 * 256 * (double)iters;
  const double us_per_k128 = ms_per * 1000.0 / iters;
  printf("  %-34s %s operands: %8.3f us per K = 128 step of all CUs  = %7.1f TFLOP/s of the PLAIN product  (%d batches)\n", what,
         quiet ? "all-zero" : "random  ", us_per_k128, flop_plain / (ms_per * 1e-3) / 1e12, launches);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  printf("== part 1: layout of v_mfma_scale_f32_16x16x128_f8f6f4 (hypothesis in the header) ==\n");
  bool ok = true;
  ok &= layout_case<0, 0, 0, 0>(true) >= 0;
  ok &= layout_case<0, 0, 1, 2>(true) >= 0;
  ok &= layout_case<0, 0, 3, 1>(true) >= 0;
  ok &= layout_case<1, 1, 0, 0>(true) >= 0;
  ok &= layout_case<0, 1, 2, 3>(true) >= 0;
  ok &= layout_case<2, 2, 0, 0>(true) >= 0;
  ok &= layout_case<3, 3, 1, 1>(true) >= 0;
  ok &= layout_case<4, 4, 0, 0>(true) >= 0;
  ok &= layout_case<4, 3, 3, 0>(true) >= 0;
  ok &= layout_case<0, 0, 0, 0>(false) >= 0;
  ok &= layout_case<1, 0, 1, 0>(false) >= 0;
  ok &= layout_case<3, 2, 0, 0>(false) >= 0;
  ok &= layout_case<4, 4, 2, 2>(false) >= 0;
  printf("  (mixed 8-bit x 4/6-bit operands, for the record:)\n");
  layout_case<0, 4, 0, 2>(true);
  printf("part 1: %s\n", ok ? "every same-width case matches one of the hypotheses" : "SOME CASE MATCHES NONE of the hypotheses");

  printf("== part 2: MFMA streams under the power cap, %.1f s per case ==\n", seconds);
  const size_t nfrag = 14 * 8 * 64;
  std::vector<_Float16> h(nfrag * 8);
  for (auto& v : h) {
    float acc = 0.f;
    for (int k = 0; k < 4; ++k) acc += (float)rnd() / 16777216.f - 0.5f;
    v = (_Float16)(acc * 1.7320508f);
  }
  std::vector<uint32_t> h8v(7 * 8 * 64 * 8);
  for (auto& v : h8v) {   // four fp8 codes per word, no NaN codes (0x7f / 0xff); as fp4 every nibble is a number
    uint32_t w = 0;
    for (int k = 0; k < 4; ++k) { uint32_t c = rnd() & 0xff; if ((c & 0x7f) == 0x7f) c ^= 1; w |= c << (8 * k); }
    v = w;
  }
  h8* src;
  i32x8* src8;
  float* out;
  CK(hipMalloc(&src, nfrag * 16));
  CK(hipMalloc(&src8, h8v.size() * 4));
  CK(hipMalloc(&out, 256 * 512 * 4));
  for (int quiet = 0; quiet < 2; ++quiet) {
    if (quiet) { CK(hipMemset(src, 0, nfrag * 16)); CK(hipMemset(src8, 0, h8v.size() * 4)); }
    else { CK(hipMemcpy(src, h.data(), nfrag * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(src8, h8v.data(), h8v.size() * 4, hipMemcpyHostToDevice)); }
    run_rate<0>(src, src8, out, seconds, "hi (96 f16 MFMAs)", quiet);
    run_rate<1>(src, src8, out, seconds, "hi + lo16 (192 f16 MFMAs)", quiet);
    run_rate<2>(src, src8, out, seconds, "hi + lo8 (96 f16 + 24 fp8 K=128)", quiet);
    run_rate<3>(src, src8, out, seconds, "hi + lo4 (96 f16 + 24 fp4 K=128)", quiet);
    run_rate<4>(src, src8, out, seconds, "lo8 alone (24 fp8 K=128)", quiet);
    run_rate<5>(src, src8, out, seconds, "lo4 alone (24 fp4 K=128)", quiet);
  }
  return ok ? 0 : 1;
}
