// Operand / accumulator layout of v_mfma_f32_16x16x16_bf16 (the _1k builtin), read off the hardware (diagnostic).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_layout_bf16.hip -o tools/mfma_layout_bf16.bin
// Hypothesis: A[m = l & 15][k = 4 (l >> 4) + i], B[k = 4 (l >> 4) + i][n = l & 15] (i = element of the lane's 4-vector),
// D[m = 4 (l >> 4) + r][n = l & 15].  Test: A[m][k] = (m + 1) if k == K0 else 0; B[k][n] = (n + 1) if k == K0 else 0
// -> D[m][n] = (m + 1)(n + 1) for every K0 in 0..15 (small integers: exact in bf16).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ short bf16_of(float x) { return (short)(__float_as_uint(x) >> 16); }
__global__ void k(float *out) {
  const int l = threadIdx.x;
  for (int K0 = 0; K0 < 16; ++K0) {
    s4 a, b;
    for (int i = 0; i < 4; ++i) {
      const int kk = 4 * (l >> 4) + i;
      a[i] = kk == K0 ? bf16_of((float)((l & 15) + 1)) : (short)0;
      b[i] = kk == K0 ? bf16_of((float)((l & 15) + 1)) : (short)0;
    }
    f4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(K0 * 64 + l) * 4 + r] = acc[r];
  }
}
int main() {
  float *d;
  (void)hipMalloc(&d, 16 * 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  static float h[16 * 256];
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int K0 = 0; K0 < 16; ++K0)
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * (l >> 4) + r, n = l & 15;
        if (h[(K0 * 64 + l) * 4 + r] != (float)((m + 1) * (n + 1))) ok = 0;
      }
  printf("bf16 16x16x16: A[m = l & 15][k = 4 (l >> 4) + i], B[k = 4 (l >> 4) + i][n = l & 15], D[m = 4 (l >> 4) + r][n = l & 15]: %s\n",
         ok ? "yes" : "NO");
  printf("K0 = 5, lane 17: %g %g %g %g\n", h[(5 * 64 + 17) * 4], h[(5 * 64 + 17) * 4 + 1], h[(5 * 64 + 17) * 4 + 2], h[(5 * 64 + 17) * 4 + 3]);
  return 0;
}
