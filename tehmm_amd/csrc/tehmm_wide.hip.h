// tehmm_wide.hip.h -- chunk-parallel forward / backward / posterior for 64 <= N <= 128 states (round 3).
//
// Reference: BaseHMM.score_samples (basehmm.py:238-273) over _hmm._forward / _backward (_hmm.pyx:120-198) for the
// model sizes of BASELINE configs[4] (100 states).  Round 2 ran these as ONE four-wave workgroup per interval
// (k_forward_wide / k_backward_wide: 1.5 us per position and interval, 300 ms for 2 Mb in 20 intervals).  Here the
// scaled sum-product recurrence runs item-parallel on the fp64 matrix cores, in the transposed form of
// tehmm_fused.hip.h (lane = (item of a 16-item tile, state quarter kq), a lane holds the states kq + 4 k; the
// accumulator layout of v_mfma_f64_16x16x4 is the next step's B operand), with what does not fit a register file
// at this size moved out:
//   * the N x N transition matrix lives in LDS as matrix-core A fragments [row tile][k step][lane] (100 KB at 112
//     padded states, 131 KB at 128): one conflict-free ds_read_b64 per matrix instruction, shared by the four waves
//     of a workgroup -- LDS bandwidth is 4 % of the matrix time;
//   * the emission rows are not fused: k_wide_emis (lane = state pair, coalesced table rows, reference summation
//     order) leaves exp(row - max) and the max per position in HBM, both passes read them back (2.7 KB per position,
//     HBM time << matrix time at this N);
//   * there is no sequential chain: an item either starts exactly (the first item of an interval forward, the last
//     one backward) or warms up over Wu positions from a uniform vector, and k_wide_links CHECKS every link (Hilbert
//     distance <= TEHMM_FB_TOL between the vector an item arrives with and the one its neighbour left).  One failed
//     link -- or an emission row no state can emit, whose semantics (quirk Q9, NaN lattices) belong to the sequential
//     kernels -- and the host retries with twice the warm-up, then falls back to the sequential kernels.
// Padded state counts: 80, 96, 112, 128 (row tiles of 16); pads carry zero probability.
// Tolerance: posteriors to 1e-6 (alpha' rows are kept as floats between the passes, as in the fused passes);
// the forward log-likelihood is summed in fp64 from the items' scale records and the links' ratios.
#pragma once
#include "tehmm_fused.hip.h"

namespace tehmm {

template <int NPW>
struct WideGeom {
  static constexpr int KS = NPW / 4;        // states per lane
  static constexpr int RT = NPW / 16;       // row tiles
  static constexpr size_t FRAG_BYTES = (size_t)RT * KS * 64 * sizeof(double);
};

// ---- emission rows: E [row][NPW] = exp(x - max) (pads 0), ms [row] = max; row = user row (out0 + t) ------------
// one wave per item, lane = state pair (j, j + 64); flags[0] counts rows no state can emit
__global__ __launch_bounds__(256) void k_wide_emis(IntervalTab iv, EmisTab em, LaneGeom lg, int N, int NPW, double *E,
                                                   double *ms, int *flags) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= lg.n_items) return;
  const int id = lg.item_iv[item];
  const int64_t t0 = lg.item_t0[item], T = iv.len[id], p0 = iv.pos0[id], r0 = iv.out0[id];
  const int len = (int)min((int64_t)lg.L, T - t0);
  bool bad = false;
  for (int s = 0; s < len; ++s) {
    double x[2];
    emis_log_wide(em, em.tab, p0 + t0 + s, lane, N, x);        // (no LDS copy of the small tracks: ldsbase < 0 everywhere)
    const double m = row_max<2>(x, lane, N);
    const bool good = m > -1e20;
    bad = bad | !good;
    double *dst = E + (r0 + t0 + s) * (int64_t)NPW;
    dst[lane] = (good && lane < N) ? exp_nonpos(x[0] - m) : 0.0;
    if (lane + 64 < NPW) dst[lane + 64] = (good && lane + 64 < N) ? exp_nonpos(x[1] - m) : 0.0;
    if (lane == 0) ms[r0 + t0 + s] = good ? m : 0.0;
  }
  if (bad && lane == 0) atomicAdd(&flags[0], 1);
}

// A fragments of the workgroup: fwd: frag[rt][s][l] = A[4 s + (l >> 4)][16 rt + (l & 15)]  (new = A^T old)
//                                bwd: frag[rt][s][l] = A[16 rt + (l & 15)][4 s + (l >> 4)]  (beta = A w)
template <int NPW, int DIR>
__device__ __forceinline__ void wide_stage_frags(double *frag, const double *__restrict__ A, int NP) {
  using G = WideGeom<NPW>;
  for (int i = threadIdx.x; i < G::RT * G::KS * 64; i += blockDim.x) {
    const int l = i & 63, s = (i >> 6) % G::KS, rt = (i >> 6) / G::KS;
    const int a = 4 * s + (l >> 4), bcol = 16 * rt + (l & 15);
    const int r = DIR == 0 ? a : bcol, c = DIR == 0 ? bcol : a;
    frag[i] = (r < NP && c < NP) ? A[(size_t)r * NP + c] : 0.0;
  }
}

// acc[rt] += sum_k frag[rt][k] x v[k]: the A fragments of step k + 1 are requested from LDS while the matrix
// instructions of step k run (two RT-register buffers, pinned: left alone the scheduler hoists all RT KS reads and
// the kernel spills)
template <int NPW>
__device__ __forceinline__ void wide_product(const double *frag, int lane, const double (&v)[WideGeom<NPW>::KS],
                                             lane_d4 (&acc)[WideGeom<NPW>::RT]) {
  constexpr int KS = WideGeom<NPW>::KS, RT = WideGeom<NPW>::RT;
  double f[2][RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) f[0][rt] = frag[(rt * KS) * 64 + lane];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    if (k + 1 < KS) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) f[(k + 1) & 1][rt] = frag[(rt * KS + k + 1) * 64 + lane];
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[k & 1][rt], v[k], acc[rt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// AL element (tile, s, k, lane): float alpha' of state (lane >> 4) + 4 k of item 16 tile + (lane & 15)
template <int NPW>
__device__ __forceinline__ int64_t wide_al_index(int64_t tile, int L, int s, int k, int lane) {
  return (((tile * L + s) * WideGeom<NPW>::KS + k) << 6) + lane;
}

// ------------------------------------------------------------------------------------------
// Forward pass.  block = 256 (four 16-item tiles), grid = ceil(tiles / 4).  Per item: pre (the vector it arrives
// with at its first position, after the warm-up), end (the vector at its last position), SL (log-scale gained over
// its official positions).  Items that start an interval start exactly from pi * b_0.
// ------------------------------------------------------------------------------------------
template <int NPW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_wide_fwd(IntervalTab iv, LaneGeom lg, int N, int NP, int Wu, const double *__restrict__ A,
                const double *__restrict__ pi, const double *__restrict__ E, const double *__restrict__ ms, float *AL,
                double *pre, double *end, double *SL) {
  using G = WideGeom<NPW>;
  constexpr int KS = G::KS, RT = G::RT;
  extern __shared__ double wide_lds[];
  wide_stage_frags<NPW, 0>(wide_lds, A, NP);
  __syncthreads();
  const int lane = threadIdx.x & 63, kq = lane >> 4;
  const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile * 16 >= lg.n_items) return;
  const int64_t item = tile * 16 + (lane & 15);
  const bool valid = item < lg.n_items;
  const int id = valid ? lg.item_iv[item] : 0;
  const int64_t t0 = valid ? lg.item_t0[item] : 0, T = iv.len[id], r0 = iv.out0[id] + t0;
  const int L = lg.L;
  const int len = valid ? (int)min((int64_t)L, T - t0) : 0;
  // warm-up: Wu positions before the item, or -- where the interval starts within that reach -- everything from
  // position 0, started EXACTLY from pi * b_0 (Wu may exceed the item length)
  const int wu = (int)min((int64_t)Wu, t0);
  const bool exact = t0 <= (int64_t)Wu;
  double v[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) v[k] = kq + 4 * k < N ? 1.0 / (double)N : 0.0;
  double slog = 0.0;
  auto vec_out = [&](double *dst) {
#pragma unroll
    for (int k = 0; k < KS; ++k) dst[item * NPW + kq + 4 * k] = v[k];
  };
  for (int s = -Wu; s < L; ++s) {
    const bool act = valid && s < len && s >= -wu;
    if (s == 0 && valid && t0 > 0 && len > 0) vec_out(pre);
    const double *er = E + (r0 + (act ? s : 0)) * (int64_t)NPW + kq;
    const double msv = act ? ms[r0 + s] : 0.0;
    lane_d4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (lane_d4){0.0, 0.0, 0.0, 0.0};
    wide_product<NPW>(wide_lds, lane, v, acc);
    if (__any(exact && s == -wu)) {
      // position 0 of an interval: alpha_0 = pi * b_0 instead of the product (the accumulators are free here)
#pragma unroll
      for (int k = 0; k < KS; ++k)
        if (exact && s == -wu) acc[k >> 2][k & 3] = kq + 4 * k < N ? exp(pi[kq + 4 * k]) : 0.0;
    }
    // the emission row is read twice (sum, then scaled values) instead of being held in KS more registers
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < KS; ++k) t += acc[k >> 2][k & 3] * (act ? er[4 * k] : 0.0);
    t = item_sum4(t);
    const int e = ((__double2hiint(t) >> 20) & 0x7ff) - 1022;
    const double scale = __hiloint2double((1023 - e) << 20, 0);
    if (act) {
#pragma unroll
      for (int k = 0; k < KS; ++k) v[k] = (acc[k >> 2][k & 3] * er[4 * k]) * scale;
    }
    if (act && s >= 0) {
      slog += (double)e * 0.6931471805599453 + msv;
#pragma unroll
      for (int k = 0; k < KS; ++k) AL[wide_al_index<NPW>(tile, L, s, k, lane)] = (float)v[k];
    }
  }
  if (valid && len > 0) {
    vec_out(end);
    if (kq == 0) SL[item] = slog;
  }
}

// ------------------------------------------------------------------------------------------
// Backward pass + posterior.  v = w_{t+1} = b_{t+1} * beta_{t+1};  beta_t = normalise(A v);  posterior row
// normalise(alpha'_t * beta_t) (+ the float32-eps quirk of score_samples, basehmm.py:271-272) straight to
// post [T][N].  pre = w at the item's last position + 1 as the warm-up left it, end = w at its first position.
// Items that end an interval start exactly from beta_{T-1} = 1.
// ------------------------------------------------------------------------------------------
template <int NPW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_wide_bwd(IntervalTab iv, LaneGeom lg, int N, int NP, int Wu, const double *__restrict__ A,
                const double *__restrict__ E, const float *__restrict__ AL, double *post, double *pre, double *end) {
  using G = WideGeom<NPW>;
  constexpr int KS = G::KS, RT = G::RT;
  extern __shared__ double wide_lds[];
  wide_stage_frags<NPW, 1>(wide_lds, A, NP);
  __syncthreads();
  const int lane = threadIdx.x & 63, kq = lane >> 4;
  const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile * 16 >= lg.n_items) return;
  const int64_t item = tile * 16 + (lane & 15);
  const bool valid = item < lg.n_items;
  const int id = valid ? lg.item_iv[item] : 0;
  const int64_t t0 = valid ? lg.item_t0[item] : 0, T = iv.len[id], r0 = iv.out0[id] + t0;
  const int L = lg.L;
  const int len = valid ? (int)min((int64_t)L, T - t0) : 0;
  const bool last = t0 + len >= T;
  const int wu = (!valid || last) ? 0 : (int)min((int64_t)Wu, T - (t0 + L));   // warm-up positions behind the item
  const int top = last ? len - 1 : L + wu - 1;                                  // where this item's recurrence starts
  double v[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) v[k] = 0.0;
  const double eps = 1.1920928955078125e-07;
  const double inv_epsden = 1.0 / (1.0 + (double)N * eps);
  auto vec_out = [&](double *dst) {
#pragma unroll
    for (int k = 0; k < KS; ++k) dst[item * NPW + kq + 4 * k] = v[k];
  };
  for (int s = L + Wu - 1; s >= 0; --s) {
    const bool act = valid && len > 0 && s <= top;
    if (s == L - 1 && valid && !last) vec_out(pre);              // v = w_{t0 + L} as the warm-up left it
    const double *er = E + (r0 + (act ? s : 0)) * (int64_t)NPW + kq;
    const bool official = act && s < len;
    lane_d4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (lane_d4){0.0, 0.0, 0.0, 0.0};
    wide_product<NPW>(wide_lds, lane, v, acc);
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < KS; ++k) t += acc[k >> 2][k & 3];
    t = item_sum4(t);
    const int e = ((__double2hiint(t) >> 20) & 0x7ff) - 1022;
    const double scale = __hiloint2double((1023 - e) << 20, 0);
    // beta_t in the accumulators (beta_{T-1} = 1 / the uniform start of a warm-up at s == top)
#pragma unroll
    for (int k = 0; k < KS; ++k)
      acc[k >> 2][k & 3] = s == top ? (kq + 4 * k < N ? 1.0 : 0.0) : acc[k >> 2][k & 3] * scale;
    if (s < L && official) {
      // the alpha' row is read twice (row sum, then the values) instead of being held in registers
      const float *ar = AL + wide_al_index<NPW>(tile, L, s, 0, lane);
      double gt = 0.0;
#pragma unroll
      for (int k = 0; k < KS; ++k) gt += (double)ar[(int64_t)k << 6] * acc[k >> 2][k & 3];
      gt = item_sum4(gt);
      const double inv = 1.0 / gt;
      double *pr = post + (r0 + s) * (int64_t)N + kq;
#pragma unroll
      for (int k = 0; k < KS; ++k)
        if (kq + 4 * k < N) pr[4 * k] = (((double)ar[(int64_t)k << 6] * acc[k >> 2][k & 3]) * inv + eps) * inv_epsden;
    }                                 // (the four lanes of an item take the branch together: item_sum4 is safe)
    if (act) {
#pragma unroll
      for (int k = 0; k < KS; ++k) v[k] = er[4 * k] * acc[k >> 2][k & 3];
    }
  }
  if (valid && len > 0) vec_out(end);
}

// ---- links and log-likelihood --------------------------------------------------------------------------------
// one thread per item: forward link (item vs item - 1) and backward link (item vs item + 1) inside an interval;
// flags[1] counts failed links; lr [item] = log(rho) of the forward link
__global__ __launch_bounds__(256) void k_wide_links(IntervalTab iv, LaneGeom lg, int N, int NPW, const double *pre_f,
                                                    const double *end_f, const double *pre_b, const double *end_b,
                                                    double *lr, int *flags) {
  const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= lg.n_items) return;
  const int id = lg.item_iv[item];
  const int64_t t0 = lg.item_t0[item], T = iv.len[id];
  int fails = 0;
  lr[item] = 0.0;
  if (t0 > 0) {                                              // forward: pre_f[item] against end_f[item - 1]
    const double *a = pre_f + item * NPW, *b = end_f + (item - 1) * NPW;
    bool bad = false;
    double rmax = -1.0, rmin = INFINITY;
    for (int j = 0; j < N; ++j) link_accum(a[j], b[j], bad, rmax, rmin);
    const bool ok = !bad && rmax > 0.0 && rmin < INFINITY && rmax / rmin - 1.0 <= TEHMM_FB_TOL;
    fails += !ok;
    lr[item] = ok ? log(rmax) : 0.0;
  }
  if (t0 + lg.L < T) {                                       // backward: pre_b[item] against end_b[item + 1]
    const double *a = pre_b + item * NPW, *b = end_b + (item + 1) * NPW;
    bool bad = false;
    double rmax = -1.0, rmin = INFINITY;
    for (int j = 0; j < N; ++j) link_accum(a[j], b[j], bad, rmax, rmin);
    const bool ok = !bad && rmax > 0.0 && rmin < INFINITY && rmax / rmin - 1.0 <= TEHMM_FB_TOL;
    fails += !ok;
  }
  if (fails) atomicAdd(&flags[1], fails);
}

// one thread per interval: log P = sum of the items' scale gains - the links' log(rho) + log(sum of the last vector)
__global__ __launch_bounds__(64) void k_wide_loglik(IntervalTab iv, LaneGeom lg, int N, int NPW, const double *end_f,
                                                    const double *SL, const double *lr, double *fwd_logprob) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= iv.n) return;
  const int64_t T = iv.len[id];
  if (T <= 0) return;
  const int64_t i0 = lg.ifirst[id], i1 = lg.ifirst[id + 1];
  double s = 0.0;
  for (int64_t it = i0; it < i1; ++it) s += SL[it] - lr[it];
  double tot = 0.0;
  const double *e = end_f + (i1 - 1) * NPW;
  for (int j = 0; j < N; ++j) tot += e[j];
  fwd_logprob[id] = s + log(tot);
}

}  // namespace tehmm
