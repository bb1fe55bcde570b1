// Layout probe (run on the GPU): v_mfma_f32_32x32x16_f16 operand / result lanes, v_permlane32_swap, ds_read_b64_tr_b16.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short i16v4;
__global__ void k(float* out, unsigned* sw, unsigned short* tr) {
  const int l = threadIdx.x, i = l & 31, hi = l >> 5;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    const int kk = 8 * hi + j;
    a[j] = (_Float16)((kk == (i & 15)) ? 1.0f : 0.0f);      // A[i][k] = delta(k, i % 16)
    b[j] = (_Float16)(float)(kk * 32 + i);                    // B[k][n] = 32 k + n  (n = l & 31)
  }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];       // expect D[row][col] = 32 (row % 16) + col
  unsigned x = 100 + l, y = 200 + l;
  auto r2 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  sw[2 * l] = r2[0]; sw[2 * l + 1] = r2[1];
  __shared__ unsigned short img[16 * 64];
  for (int e = l; e < 16 * 64; e += 64) img[e] = (unsigned short)e;   // row-major [16 rows][64 cols], value = 64 row + col
  __syncthreads();
  const int fl = l & 15, g = l >> 4;
  // group g: rows 4 g .. 4 g + 3, cols 0..15: lane fl -> row 4 g + (fl >> 2), cols 4 (fl & 3)
  i16v4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16v4*)(img + (4 * g + (fl >> 2)) * 64 + 4 * (fl & 3)));
  for (int j = 0; j < 4; ++j) tr[4 * l + j] = (unsigned short)v[j];
}
int main() {
  float* out; unsigned* sw; unsigned short* tr;
  hipMalloc(&out, 64 * 16 * 4); hipMalloc(&sw, 128 * 4); hipMalloc(&tr, 256 * 2);
  k<<<1, 64>>>(out, sw, tr);
  float h[1024]; unsigned hs[128]; unsigned short ht[256];
  hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hs, sw, sizeof hs, hipMemcpyDeviceToHost); hipMemcpy(ht, tr, sizeof ht, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    if (h[l * 16 + r] != 32.0f * (row % 16) + col) ++bad;
  }
  printf("mfma 32x32x16 C layout (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)): %d mismatches\n", bad);
  if (bad) for (int l = 0; l < 64; l += 9) { printf("lane %d:", l); for (int r = 0; r < 16; ++r) printf(" %g", h[l * 16 + r]); printf("\n"); }
  printf("permlane32_swap(x = 100 + l, y = 200 + l): lane 0 -> (%u, %u), lane 5 -> (%u, %u), lane 32 -> (%u, %u), lane 37 -> (%u, %u)\n", hs[0], hs[1], hs[10], hs[11], hs[64], hs[65], hs[74], hs[75]);
  printf("tr16: lane 0 -> %u %u %u %u | lane 1 -> %u %u %u %u | lane 5 -> %u %u %u %u | lane 17 -> %u %u %u %u\n", ht[0], ht[1], ht[2], ht[3], ht[4], ht[5], ht[6], ht[7], ht[20], ht[21], ht[22], ht[23], ht[68], ht[69], ht[70], ht[71]);
  return 0;
}
