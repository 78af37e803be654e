// Two waves of one workgroup (different SIMDs) exchanging a 36-double partial vector every step
// through LDS with a sequence word (no s_barrier): cost of the hand-shake vs the work per step.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
template <int WORK>
__global__ __launch_bounds__(128) void k_pp(double *out, unsigned long long *cyc, int iters, int *fail) {
  __shared__ double part[2][2][64];     // [parity][wave][lane]
  __shared__ volatile int seq[2];       // per wave: last step published
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < 2) seq[threadIdx.x] = -1;
  __syncthreads();
  double v = 1.0 + lane * 1e-3, c = out[0] + 1e-9;
  unsigned long long t0 = now();
  for (int it = 0; it < iters; ++it) {
    double a0 = v, a1 = v, a2 = v, a3 = v;
#pragma unroll
    for (int k = 0; k < WORK / 4; ++k) { a0 = fma(a0, c, c); a1 = fma(a1, c, c); a2 = fma(a2, c, c); a3 = fma(a3, c, c); }
    const double mine = (a0 + a1) + (a2 + a3);
    part[it & 1][w][lane] = mine;
    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): data landed before the sequence word
    if (lane == 0) seq[w] = it;
    // wait for the partner
    int spins = 0;
    while (seq[w ^ 1] < it) {
      if (++spins > 100000) { *fail = 1; break; }
    }
    const double other = part[it & 1][w ^ 1][lane];
    v = fmax(mine, other) * 0.5 + 0.25;
  }
  unsigned long long t1 = now();
  out[1 + threadIdx.x] = v;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double *out; unsigned long long *cyc; int *fail;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64); hipMalloc(&fail, 4);
  hipMemset(out, 0, 4096); hipMemset(fail, 0, 4);
  const int iters = 20000;
  unsigned long long h; int hf;
#define RUN(W) hipLaunchKernelGGL((k_pp<W>), dim3(1), dim3(128), 0, 0, out, cyc, iters, fail); hipDeviceSynchronize(); \
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost); \
  printf("work %3d fma per step: %.1f cycles per step (work alone ~%.0f) fail=%d\n", W, (double)h / iters, W * 4.7, hf);
  RUN(8) RUN(40) RUN(72) RUN(104)
  return 0;
}
