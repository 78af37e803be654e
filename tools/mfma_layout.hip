// Operand / accumulator layout of v_mfma_f32_16x16x4_f32 and v_mfma_f64_16x16x4_f64, read off the hardware (diagnostic):
// A[m][k] = 1 + m + 100 k at lane (m = l & 15, k = l >> 4), B[k][n] = [k == kk] picks one k: D[m][n] = 1 + m + 100 kk.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_layout.hip -o tools/mfma_layout.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(float *o32, double *o64) {
  const int l = threadIdx.x;
  const float a = 1.f + (l & 15) + 100.f * (l >> 4);
  const float b = ((l >> 4) == 2) ? (float)(1000 * (l & 15)) + 1.f : 0.f;       // k = 2 only; value encodes n
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) o32[l * 4 + r] = acc[r];
  d4 acd = {0, 0, 0, 0};
  acd = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a, (double)b, acd, 0, 0, 0);
  for (int r = 0; r < 4; ++r) o64[l * 4 + r] = acd[r];
}
int main() {
  float *d32; double *d64;
  hipMalloc(&d32, 256 * 4); hipMalloc(&d64, 256 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d32, d64);
  float h32[256]; double h64[256];
  hipMemcpy(h32, d32, sizeof(h32), hipMemcpyDeviceToHost);
  hipMemcpy(h64, d64, sizeof(h64), hipMemcpyDeviceToHost);
  // D[m][n] = A[m][2] * B[2][n] = (1 + m + 200) * (1000 n + 1)
  int ok32a = 1, ok32b = 1, ok64a = 1, ok64b = 1;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int n = l & 15;
      const int ma = 4 * (l >> 4) + r, mb = 4 * r + (l >> 4);
      const double ea = (201.0 + ma) * (1000.0 * n + 1.0), eb = (201.0 + mb) * (1000.0 * n + 1.0);
      ok32a &= (double)h32[l * 4 + r] == (double)(float)ea; ok32b &= (double)h32[l * 4 + r] == (double)(float)eb;
      ok64a &= h64[l * 4 + r] == ea; ok64b &= h64[l * 4 + r] == eb;
    }
  printf("f32 16x16x4: D[m = 4 (l >> 4) + r][n = l & 15]: %s;  D[m = 4 r + (l >> 4)][n = l & 15]: %s\n", ok32a ? "yes" : "no", ok32b ? "yes" : "no");
  printf("f64 16x16x4: D[m = 4 (l >> 4) + r][n = l & 15]: %s;  D[m = 4 r + (l >> 4)][n = l & 15]: %s\n", ok64a ? "yes" : "no", ok64b ? "yes" : "no");
  printf("lane 17: f32 %g %g %g %g   f64 %g %g %g %g\n", h32[68], h32[69], h32[70], h32[71], h64[68], h64[69], h64[70], h64[71]);
  return 0;
}
