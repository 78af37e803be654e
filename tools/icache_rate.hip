// Does a LONG straight-line stream of 8-byte fp64 VALU instructions (scalar operand each) issue at the rate of a short loop?
// (diagnostic)  hipcc --offload-arch=gfx950 -O3 tools/icache_rate.hip -o tools/icache_rate.bin
// BODY instructions (add/max pairs, as the quantised Viterbi pass) per loop iteration, straight line; waves/SIMD 1..3.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"

#define P4(a) a a a a
#define P16(a) P4(P4(a))
#define P64(a) P4(P16(a))
#define P256(a) P4(P64(a))
#define PAIR                                                                     \
  asm volatile("v_add_f64 %0, %1, %2\n v_max_f64 %3, %3, %0" : "=&v"(t), "+s"(s0) : "v"(w0), "v"(x0)); \
  asm volatile("v_add_f64 %0, %1, %2\n v_max_f64 %3, %3, %0" : "=&v"(t), "+s"(s1) : "v"(w1), "v"(x1)); \
  asm volatile("v_add_f64 %0, %1, %2\n v_max_f64 %3, %3, %0" : "=&v"(t), "+s"(s2) : "v"(w2), "v"(x2)); \
  asm volatile("v_add_f64 %0, %1, %2\n v_max_f64 %3, %3, %0" : "=&v"(t), "+s"(s3) : "v"(w3), "v"(x3));

template <int BODY>   // BODY = 8 * REP instructions
__global__ __launch_bounds__(256) void k_stream(double *out, double s0, double s1, double s2, double s3, int iters) {
  double w0 = threadIdx.x, w1 = w0 + 1, w2 = w0 + 2, w3 = w0 + 3;
  double x0 = 0, x1 = 0, x2 = 0, x3 = 0, t;
  for (int it = 0; it < iters; ++it) {
    if (BODY == 16) { P4(PAIR) P4(PAIR) }   // 2 * 4 * 8 / 4 ... see count below
    if (BODY == 256) { P16(PAIR) P16(PAIR) }
    if (BODY == 1024) { P64(PAIR) P64(PAIR) }
    if (BODY == 4096) { P256(PAIR) P256(PAIR) }
  }
  if (x0 + x1 + x2 + x3 == 12345.678) out[0] = x0;
}
// instructions per iteration: PAIR = 8 instructions; BODY==16 -> 8 PAIR = 64; 256 -> 32 PAIR = 256; 1024 -> 128 PAIR = 1024; 4096 -> 512 PAIR = 4096
template <int BODY>
static void run(int wps, double *d) {
  const int per_iter = BODY == 16 ? 64 : BODY;
  const int iters = (1 << 22) / per_iter;
  const int blocks = 256 * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_stream<BODY>, dim3(blocks), dim3(256), 0, 0, d, 1.0, 2.0, 3.0, 4.0, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_stream<BODY>, dim3(blocks), dim3(256), 0, 0, d, 1.0, 2.0, 3.0, 4.0, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * per_iter;
  printf("straight-line body %5d instr (%6d B), %d wave(s)/SIMD: %7.3f ms  %5.2f ns per instruction and SIMD\n", per_iter, per_iter * 8,
         wps, ms, ms * 1e6 / (n * wps));
}
int main() {
  double *d; hipMalloc(&d, 64);
  for (int w : {1, 2, 3}) { run<16>(w, d); run<256>(w, d); run<1024>(w, d); run<4096>(w, d); }
  return 0;
}
