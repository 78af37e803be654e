// Microbenchmarks of the primitives the chain kernels are built from (diagnostic, not product).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

#define REP 64
template <int MODE>
__global__ void k_valu(double *out, unsigned long long *cyc, int iters) {
  double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,
         a7 = a0 + 7;
  float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
  const double c = out[0];
  const float cf = (float)c;
  unsigned long long t0 = now();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (MODE == 0) {   // 8 independent f64 add chains
        a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c;
      } else if (MODE == 1) {   // one dependent f64 add chain
        a0 += c; a0 += c; a0 += c; a0 += c; a0 += c; a0 += c; a0 += c; a0 += c;
      } else if (MODE == 2) {   // 8 independent f64 max (asm so that it is not folded)
#define VMAX(x) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(c))
        VMAX(a0); VMAX(a1); VMAX(a2); VMAX(a3); VMAX(a4); VMAX(a5); VMAX(a6); VMAX(a7);
      } else if (MODE == 3) {   // 8 independent f64 fma
        a0 = fma(a0, c, c); a1 = fma(a1, c, c); a2 = fma(a2, c, c); a3 = fma(a3, c, c);
        a4 = fma(a4, c, c); a5 = fma(a5, c, c); a6 = fma(a6, c, c); a7 = fma(a7, c, c);
      } else if (MODE == 4) {   // 8 independent f32 add
        f0 += cf; f1 += cf; f2 += cf; f3 += cf; f4 += cf; f5 += cf; f6 += cf; f7 += cf;
      } else if (MODE == 5) {   // dependent f32 add chain
        f0 += cf; f0 += cf; f0 += cf; f0 += cf; f0 += cf; f0 += cf; f0 += cf; f0 += cf;
      } else if (MODE == 6) {   // readlane pairs feeding independent f64 adds
        a0 += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a7), 1), __builtin_amdgcn_readlane(__double2loint(a7), 1));
        a1 += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a7), 2), __builtin_amdgcn_readlane(__double2loint(a7), 2));
        a2 += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a7), 3), __builtin_amdgcn_readlane(__double2loint(a7), 3));
        a3 += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a7), 4), __builtin_amdgcn_readlane(__double2loint(a7), 4));
        a4 += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a7), 5), __builtin_amdgcn_readlane(__double2loint(a7), 5));
        a5 += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a7), 6), __builtin_amdgcn_readlane(__double2loint(a7), 6));
        a6 += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a7), 7), __builtin_amdgcn_readlane(__double2loint(a7), 7));
        a7 += c;
      } else if (MODE == 7) {   // f64 compare (to vcc) + 32-bit select
#define VCS(x, f) asm volatile("v_cmp_eq_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(f) : "v"(x), "v"(c), "v"(cf) : "vcc")
        VCS(a0, f0); VCS(a1, f1); VCS(a2, f2); VCS(a3, f3); VCS(a4, f4); VCS(a5, f5); VCS(a6, f6); VCS(a7, f7);
      } else if (MODE == 8) {   // f64 compare alone
#define VC(x) asm volatile("v_cmp_eq_f64 vcc, %0, %1" : : "v"(x), "v"(c) : "vcc")
        VC(a0); VC(a1); VC(a2); VC(a3); VC(a4); VC(a5); VC(a6); VC(a7);
      } else if (MODE == 9) {   // v_ldexp_f64
#define VL(x) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x) : "v"(it))
        VL(a0); VL(a1); VL(a2); VL(a3); VL(a4); VL(a5); VL(a6); VL(a7);
      } else if (MODE == 10) {  // v_mul_f64
        a0 *= c; a1 *= c; a2 *= c; a3 *= c; a4 *= c; a5 *= c; a6 *= c; a7 *= c;
      } else if (MODE == 11) {  // v_mov_b32 (32-bit op)
#define VM(f) asm volatile("v_mov_b32 %0, %1" : "=v"(f) : "v"(cf))
        VM(f0); VM(f1); VM(f2); VM(f3); VM(f4); VM(f5); VM(f6); VM(f7);
      }
    }
  }
  unsigned long long t1 = now();
  out[1 + threadIdx.x + blockIdx.x * blockDim.x] =
      a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// LDS write -> broadcast read round trip (dependent), one wave
__global__ void k_lds_rt(double *out, unsigned long long *cyc, int iters) {
  __shared__ double buf[128];
  double v = threadIdx.x;
  buf[threadIdx.x] = v;
  unsigned long long t0 = now();
  for (int it = 0; it < iters; ++it) {
    buf[threadIdx.x & 63] = v;
    __builtin_amdgcn_wave_barrier();
    v = buf[(it & 31)] + 1.0;     // broadcast read of a value just written by another lane
    __builtin_amdgcn_wave_barrier();
  }
  unsigned long long t1 = now();
  out[1 + threadIdx.x] = v;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// 18 broadcast ds_read_b128 issued together, then all consumed (issue + return bandwidth).
// MODE 0: reads + 36 adds; MODE 1: reads + 1 add per read pair (light use); MODE 2: ds_read_b64 x36
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const d2 lds_cd2;
typedef __attribute__((address_space(3))) const double lds_cd;
template <int MODE>
__global__ void k_lds_bcast(double *out, unsigned long long *cyc, int iters) {
  __shared__ __attribute__((aligned(16))) double buf[128];
  buf[threadIdx.x & 63] = threadIdx.x;
  buf[64 + (threadIdx.x & 63)] = threadIdx.x;
  __syncthreads();
  double acc = 0, acc2 = 0;
  unsigned long long t0 = now();
  for (int it = 0; it < iters; ++it) {
    unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const double *)(buf + (it & 1) * 64);
    asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(a));
    if (MODE == 2) {
      lds_cd *p = (lds_cd *)(size_t)a;
      double v[36];
#pragma unroll
      for (int i = 0; i < 36; ++i) v[i] = p[i];
#pragma unroll
      for (int i = 0; i < 36; i += 2) { acc += v[i]; acc2 += v[i + 1]; }
    } else {
      lds_cd2 *p = (lds_cd2 *)(size_t)a;
      d2 v[18];
#pragma unroll
      for (int i = 0; i < 18; ++i) v[i] = p[i];
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 18; ++i) { acc += v[i].x; acc2 += v[i].y; }
      } else {
#pragma unroll
        for (int i = 0; i < 18; i += 6) acc += v[i].x + v[i + 5].y;
      }
    }
  }
  unsigned long long t1 = now();
  out[1 + threadIdx.x] = acc + acc2;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// s_barrier cost: blockDim waves hit the barrier `iters` times with a little work in between
__global__ void k_barrier(double *out, unsigned long long *cyc, int iters, int skew) {
  double v = threadIdx.x;
  const double c = out[0];
  unsigned long long t0 = now();
  for (int it = 0; it < iters; ++it) {
    if (skew && (threadIdx.x >> 6) == (it & 3)) { v += c; v += c; v += c; v += c; }
    __syncthreads();
  }
  unsigned long long t1 = now();
  out[1 + threadIdx.x] = v;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// LDS exchange between waves through a barrier: write partial, barrier, read 4 partials
__global__ void k_exchange(double *out, unsigned long long *cyc, int iters) {
  __shared__ double part[2][4][64];
  double v = threadIdx.x;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long t0 = now();
  for (int it = 0; it < iters; ++it) {
    part[it & 1][w][lane] = v;
    __syncthreads();
    v = fmax(fmax(part[it & 1][0][lane], part[it & 1][1][lane]), fmax(part[it & 1][2][lane], part[it & 1][3][lane])) + 1.0;
  }
  unsigned long long t1 = now();
  out[1 + threadIdx.x] = v;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  double *out;
  unsigned long long *cyc;
  hipMalloc(&out, 1 << 20);
  hipMalloc(&cyc, 4096);
  hipMemset(out, 0, 1 << 20);
  std::vector<unsigned long long> h(64);
  const int iters = 2000;
  const char *names[] = {"f64 add indep", "f64 add dep", "f64 max indep", "f64 fma indep", "f32 add indep",
                         "f32 add dep", "readlane x2 + f64 add", "f64 cmp + cndmask", "f64 cmp", "f64 ldexp",
                         "f64 mul", "v_mov_b32"};
  for (int waves = 1; waves <= 8; waves *= 8) {
    printf("--- %d wave(s) per workgroup (one workgroup), cycles per wave-instruction group\n", waves);
#define RUN(M)                                                                                  \
  hipLaunchKernelGGL((k_valu<M>), dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);            \
  hipDeviceSynchronize();                                                                       \
  hipMemcpy(h.data(), cyc, 8, hipMemcpyDeviceToHost);                                           \
  printf("  %-24s %7.2f cycles per op\n", names[M], (double)h[0] / ((double)iters * REP));
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
  }
#define RUNL(M, label)                                                                          \
  hipLaunchKernelGGL((k_lds_bcast<M>), dim3(1), dim3(64), 0, 0, out, cyc, iters);               \
  hipDeviceSynchronize();                                                                       \
  hipMemcpy(h.data(), cyc, 8, hipMemcpyDeviceToHost);                                           \
  printf("%s: %.1f cycles per iteration\n", label, (double)h[0] / iters);
  RUNL(0, "18 x ds_read_b128 broadcast + 36 adds (2 chains)")
  RUNL(1, "18 x ds_read_b128 broadcast + 6 adds")
  RUNL(2, "36 x ds_read_b64 broadcast + 36 adds (2 chains)")
  hipLaunchKernelGGL(k_lds_rt, dim3(1), dim3(64), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), cyc, 8, hipMemcpyDeviceToHost);
  printf("LDS write->broadcast read->use round trip (1 wave): %.1f cycles\n", (double)h[0] / iters);
  for (int waves = 1; waves <= 8; waves *= 2)
    for (int skew = 0; skew < 2; ++skew) {
      hipLaunchKernelGGL(k_barrier, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters, skew);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), cyc, 8, hipMemcpyDeviceToHost);
      printf("s_barrier, %d waves, skew=%d: %.1f cycles per iteration\n", waves, skew, (double)h[0] / iters);
    }
  hipLaunchKernelGGL(k_exchange, dim3(1), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), cyc, 8, hipMemcpyDeviceToHost);
  printf("4-wave LDS exchange (write, barrier, 4 reads, 3 max, add): %.1f cycles per iteration\n",
         (double)h[0] / iters);
  return 0;
}
