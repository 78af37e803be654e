// Checks the semantics and cost of the DPP row_newbcast / permlane swap primitives (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
template <int N_>
__device__ __forceinline__ double bcast(double v) {   // value of lane N_ of the lane's own row
  return __longlong_as_double(__builtin_amdgcn_mov_dpp(__double_as_longlong(v), 0x150 + N_, 0xf, 0xf, true));
}
// replicate row R (0..3) of v into all four rows
template <int R>
__device__ __forceinline__ double rep_row(double v) {
  unsigned lo = __double2loint(v), hi = __double2hiint(v);
  auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);   // a[0] rows (0,1,0,1), a[1] rows (2,3,2,3)
  auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  unsigned sl = R < 2 ? a[0] : a[1], sh = R < 2 ? b[0] : b[1];
  auto c = __builtin_amdgcn_permlane16_swap(sl, sl, false, false);   // c[0] even-row content, c[1] odd-row content
  auto d = __builtin_amdgcn_permlane16_swap(sh, sh, false, false);
  return __hiloint2double((int)((R & 1) ? d[1] : d[0]), (int)((R & 1) ? c[1] : c[0]));
}
__global__ void k_sem(double *out) {
  const int lane = threadIdx.x;
  double v = 100.0 + lane;
  out[lane] = bcast<5>(v);                 // expect 100 + 16*row + 5
  out[64 + lane] = rep_row<0>(v);          // expect 100 + (lane & 15)
  out[128 + lane] = rep_row<1>(v);         // expect 116 + (lane & 15)
  out[192 + lane] = rep_row<2>(v);         // expect 132 + (lane & 15)
  double s = 1.0;
  double a = 2.0;
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(s) : "v"(v), "v"(a));
  out[256 + lane] = s;                     // expect 1 + 2*(100 + 16*row + 3)
}
#define REPS 36
template <int MODE>
__global__ void k_time(double *out, unsigned long long *cyc, int iters) {
  const int lane = threadIdx.x;
  double v = 1.0 + lane * 1e-3;
  double c[REPS];
#pragma unroll
  for (int i = 0; i < REPS; ++i) c[i] = out[i] + lane;
  double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  unsigned long long t0 = now();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // 36 x (mov_dpp + add), then fold
#define STEP(i) { double x = bcast<(i) & 15>(v) + c[i]; if ((i) & 1) acc1 = fmax(acc1, x); else acc0 = fmax(acc0, x); }
      STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11)
      STEP(12) STEP(13) STEP(14) STEP(15) STEP(16) STEP(17) STEP(18) STEP(19) STEP(20) STEP(21) STEP(22) STEP(23)
      STEP(24) STEP(25) STEP(26) STEP(27) STEP(28) STEP(29) STEP(30) STEP(31) STEP(32) STEP(33) STEP(34) STEP(35)
      v = fmax(acc0, acc1) * 0.5;
    } else if (MODE == 1) {   // 36 x fmac_dpp into 4 accumulators
#define FM(i, acc) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #i " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(c[i]));
      FM(0, acc0) FM(1, acc1) FM(2, acc2) FM(3, acc3) FM(4, acc0) FM(5, acc1) FM(6, acc2) FM(7, acc3)
      FM(8, acc0) FM(9, acc1) FM(10, acc2) FM(11, acc3) FM(12, acc0) FM(13, acc1) FM(14, acc2) FM(15, acc3)
      FM(0, acc0) FM(1, acc1) FM(2, acc2) FM(3, acc3) FM(4, acc0) FM(5, acc1) FM(6, acc2) FM(7, acc3)
      FM(8, acc0) FM(9, acc1) FM(10, acc2) FM(11, acc3) FM(12, acc0) FM(13, acc1) FM(14, acc2) FM(15, acc3)
      FM(0, acc0) FM(1, acc1) FM(2, acc2) FM(3, acc3)
      v = ((acc0 + acc1) + (acc2 + acc3)) * 1e-3;
      acc0 = acc1 = acc2 = acc3 = 0;
    } else if (MODE == 2) {   // replicate three rows (permlane swaps), dependent on v
      double ra = rep_row<0>(v), rb = rep_row<1>(v), rc = rep_row<2>(v);
      v = (ra + rb) + rc;
    }
  }
  unsigned long long t1 = now();
  out[64 + lane] = v + acc0 + acc1;
  if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
  double *out; unsigned long long *cyc;
  hipMalloc(&out, 1 << 16); hipMalloc(&cyc, 64);
  hipMemset(out, 0, 1 << 16);
  hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, out);
  std::vector<double> h(320);
  hipMemcpy(h.data(), out, 320 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    int row = l >> 4;
    if (h[l] != 100 + 16 * row + 5) bad++;
    if (h[64 + l] != 100 + (l & 15)) bad++;
    if (h[128 + l] != 116 + (l & 15)) bad++;
    if (h[192 + l] != 132 + (l & 15)) bad++;
    if (h[256 + l] != 1 + 2.0 * (100 + 16 * row + 3)) bad++;
  }
  printf("semantics: %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
  if (bad) { for (int l = 0; l < 64; l += 9) printf(" lane %d: bcast %.0f rep0 %.0f rep1 %.0f rep2 %.0f fmac %.0f\n", l, h[l], h[64+l], h[128+l], h[192+l], h[256+l]); }
  unsigned long long hc;
  const int iters = 2000;
  const char *nm[] = {"36 x (v_mov_b64_dpp + add + max)", "36 x v_fmac_f64_dpp", "replicate 3 rows (permlane swaps)"};
#define RUN(M) hipLaunchKernelGGL((k_time<M>), dim3(1), dim3(64), 0, 0, out, cyc, iters); hipDeviceSynchronize(); \
  hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); printf("%-40s %.1f cycles per iteration\n", nm[M], (double)hc / iters);
  RUN(0) RUN(1) RUN(2)
  return 0;
}
