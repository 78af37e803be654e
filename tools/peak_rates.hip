// Whole-chip issue rates of the instructions the throughput kernels are made of (diagnostic, not product): what
// `roofline` figures in DESIGN.md are priced against when the bound is arithmetic.
//   hipcc --offload-arch=gfx950 -O3 tools/peak_rates.hip -o tools/peak_rates.bin && tools/peak_rates.bin
// Every kernel: `waves` waves per SIMD on all 1024 SIMDs, each issuing ITER x 16 independent instructions of one
// kind; rate = instructions x flops / HIP-event time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define ITER 4096

// MODE 0: v_mfma_f64_16x16x4_f64 (2048 flop)   1: v_fma_f64 (128 flop per wave instruction)
// MODE 2: v_mfma_f32_32x32x2_f32 (4096 flop)   3: v_mfma_f32_16x16x4_f32 (2048 flop)   4: v_add_f64 + v_max_f64 pairs
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(double *out, double seed) {
  const double a = seed + threadIdx.x * 1e-9, b = 1.0 - 1e-9 * threadIdx.x;
  if (MODE == 0) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
  } else if (MODE == 1) {
    double x[16];
    for (int i = 0; i < 16; ++i) x[i] = a + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], b, a);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 12345.678) out[0] = s;
  } else if (MODE == 2) {
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
      for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    const float fa = (float)a, fb = (float)b;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
      for (int k = 0; k < 16; ++k) s += acc[i][k];
    if (s == 12345.678f) out[0] = s;
  } else if (MODE == 3) {
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f4){0, 0, 0, 0};
    const float fa = (float)a, fb = (float)b;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
  } else {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        double t = x[i] + b;
        asm volatile("v_max_f64 %0, %1, %2" : "=v"(x[i]) : "v"(t), "v"(a));
      }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678) out[0] = s;
  }
}

template <int MODE>
static void run(const char *name, double flop_per_instr, int waves_per_simd, double *d_out) {
  const int blocks = 256 * waves_per_simd;      // 256 CUs, one 4-wave block per CU and wave slot
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)ITER * 16.0;
  const double waves = (double)blocks * 4.0;
  const double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
  printf("%-28s %d wave(s)/SIMD: %8.3f ms  %7.2f TFLOP/s  %6.1f ns per instruction and SIMD\n", name, waves_per_simd, ms,
         instr_per_wave * waves * flop_per_instr / (ms * 1e-3) / 1e12, ns_per_instr_per_simd);
}

int main() {
  double *d_out;
  hipMalloc(&d_out, 64);
  for (int w : {1, 2, 4}) {
    run<0>("v_mfma_f64_16x16x4_f64", 2048.0, w, d_out);
    run<1>("v_fma_f64", 128.0, w, d_out);
    run<4>("v_add_f64 + v_max_f64", 64.0, w, d_out);
    run<2>("v_mfma_f32_32x32x2_f32", 4096.0, w, d_out);
    run<3>("v_mfma_f32_16x16x4_f32", 2048.0, w, d_out);
  }
  hipFree(d_out);
  return 0;
}
