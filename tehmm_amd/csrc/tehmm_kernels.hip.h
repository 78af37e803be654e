// tehmm_kernels.hip.h -- device code of libtehmm_hip.so (gfx950 / CDNA4, wave64): shared types,
// the array-level kernels (1:1 with the reference's Cython functions), the traceback kernels and
// the generic one-wave-per-interval fused kernels used for 64 <= N <= 128.  The cooperative
// kernels of the main path (N < 64) are in tehmm_coop.hip.h.
//
// Layout conventions (device side)
//   * generic kernels: one wavefront owns one interval; lane j (and j+64 when SPL==2) owns state
//     j; the N x N transition table lives in LDS as [from][NP] rows so that "lane = to" reads are
//     conflict-free ds_read_b64 and the previous state vector is an LDS broadcast read;
//   * observations are repacked to [position][KP] bytes (KP = K rounded up to 4) so that one row
//     is a wave-uniform (scalar) load; emission tables are packed [row][NP] with one row per
//     (track, symbol) so that a row read is one coalesced 8*N byte load;
//   * every interval starts at an internal position base that is a multiple of 64.
//
// Arithmetic: fp64 everywhere; the file must be compiled with -ffp-contract=off because the
// Viterbi recurrence has to reproduce the reference's separate multiply / add roundings
// (SURVEY Appendix B, Q6).  No MFMA: the recurrences are max-plus / scaled sum-product
// vector-matrix steps on the VALU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TEHMM_WAVE 64
// issue priority of the latency kernels (fix-up chains, traceback walks) next to co-resident throughput waves
#ifndef TEHMM_PRIO_HI
#define TEHMM_PRIO_HI 3
#define TEHMM_PRIO_LO 2
#endif
#define TEHMM_MAX_TRACKS 128
#define TEHMM_PB 16          // positions per emission phase (LDS ring depth)
#define TEHMM_TB_CHUNK 256   // traceback chunk length

namespace tehmm {

// ---- diagnostic cycle stamps (only in the -DTEHMM_STAMPS build; never in the product library)
#ifdef TEHMM_STAMPS
__device__ unsigned long long g_stamps[2 * 4096 * 16];
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define ST_DECL unsigned long long st_a = 0, st_b = 0, st_c = 0, st_t0 = 0, st_t1 = 0
#define ST_BEGIN st_t0 = stamp_now()
#define ST_ADD(acc) do { st_t1 = stamp_now(); acc += st_t1 - st_t0; st_t0 = st_t1; } while (0)
#define ST_FLUSH(w_)                                                                        \
  do {                                                                                      \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) {                                     \
      g_stamps[(blockIdx.x * 4 + (w_)) * 4 + 0] = st_a;                                     \
      g_stamps[(blockIdx.x * 4 + (w_)) * 4 + 1] = st_b;                                     \
      g_stamps[(blockIdx.x * 4 + (w_)) * 4 + 2] = st_c;                                     \
    }                                                                                       \
  } while (0)
#else
#define ST_DECL
#define ST_BEGIN
#define ST_ADD(acc)
#define ST_FLUSH(w_)
#endif

struct EmisTab {
  const uint32_t *obs32;   // [pos][KPW] packed observation rows
  const double *tab;       // [rows][NP] emission log-prob rows
  const double *ratios;    // per internal position, may be null (emission ratios)
  double normalize;
  int K, KPW, NP;
  int rowbase[TEHMM_MAX_TRACKS];
  int rowcnt[TEHMM_MAX_TRACKS];
  // cooperative kernels: tracks whose rows are staged in LDS (ldsbase[k] >= 0 = first LDS row)
  int ldsbase[TEHMM_MAX_TRACKS];
  int lds_rows;
  const double *ltab_src;  // [lds_rows][NP] the LDS-resident rows, packed
  // A symbol beyond a track's last one reads the reference's zero padding (logProbs is
  // [K][N][1 + max symbols], emission.py:136-138): row zero_row of tab / row lds_zero of the LDS
  // copy hold zeros for it (lds_zero is only meaningful while lds_rows > 0).
  int zero_row, lds_zero;
};

struct IntervalTab {
  const int *order;        // launch slot -> interval id (longest first)
  const int64_t *pos0;     // internal (64-aligned) position base per interval
  const int64_t *len;      // T per interval
  const int64_t *out0;     // user-facing concatenated row offset per interval
  int n;
};

__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}
// frexp-style exponent of a non-negative double (0 -> -1022)
__device__ __forceinline__ int exp_of(double a) {
  return ((__double2hiint(a) >> 20) & 0x7ff) - 1022;
}

// Emission log-likelihood of one position for the states this lane owns.
// Same operation order as _emission.pyx:65-72: x = 0; x += table[k][j][obs[k]] (k ascending);
// x *= normalize; x *= ratio.
template <int SPL>
__device__ __forceinline__ void emis_log(const EmisTab &e, int64_t gpos, int lane, int N,
                                         double (&x)[SPL]) {
  const uint32_t *row = e.obs32 + gpos * e.KPW;
#pragma unroll
  for (int s = 0; s < SPL; ++s) x[s] = 0.0;
  for (int k = 0; k < e.K; ++k) {
    uint32_t w = row[k >> 2];
    int sym = (int)((w >> ((k & 3) * 8)) & 0xffu);
    const int trow = sym < e.rowcnt[k] ? e.rowbase[k] + sym : e.zero_row;
    const double *tr = e.tab + (int64_t)trow * e.NP;
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
      int j = lane + TEHMM_WAVE * s;
      if (j < N) x[s] += tr[j];
    }
  }
#pragma unroll
  for (int s = 0; s < SPL; ++s) x[s] *= e.normalize;
  if (e.ratios) {
    double r = e.ratios[gpos];
#pragma unroll
    for (int s = 0; s < SPL; ++s) x[s] *= r;
  }
}

// max over the valid states of this wave
template <int SPL>
__device__ __forceinline__ double row_max(const double (&x)[SPL], int lane, int N) {
  double m = -INFINITY;
#pragma unroll
  for (int s = 0; s < SPL; ++s)
    if (lane + TEHMM_WAVE * s < N) m = fmax(m, x[s]);
  return wave_max_f64(m);
}

// ------------------------------------------------------------------------------------------
// Observation repack: user [total][K] bytes -> internal [pos][KP] rows at 64-aligned bases.
// ------------------------------------------------------------------------------------------
__global__ void k_repack_obs(int n, const int64_t *out0, const int64_t *pos0, const int64_t *len,
                             int K, int KP, const uint8_t *src, uint8_t *dst,
                             const double *rsrc, double *rdst) {
  int iv = blockIdx.y;
  if (iv >= n) return;
  int64_t T = len[iv];
  const uint8_t *s = src + out0[iv] * K;
  uint8_t *d = dst + pos0[iv] * KP;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < T * KP;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = idx / KP;
    int k = (int)(idx - t * KP);
    d[idx] = k < K ? s[t * K + k] : (uint8_t)0;
  }
  if (rsrc) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < T;
         t += (int64_t)gridDim.x * blockDim.x)
      rdst[pos0[iv] + t] = rsrc[out0[iv] + t];
  }
}

// ------------------------------------------------------------------------------------------
// Array-level emission (fastAllLogProbs, _emission.pyx:20-144)
// ------------------------------------------------------------------------------------------
template <typename ObsT>
__global__ void k_emission(int64_t T, int K, int N, int S, const ObsT *obs, const double *lp,
                           double normalize, const double *ratios, double *out,
                           unsigned long long *first_good) {
  int64_t total = T * N;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = idx / N;
    int j = (int)(idx - t * N);
    double x = 0.0;
    for (int k = 0; k < K; ++k) {
      int64_t sym = (int64_t)obs[t * K + k];
      // (the reference indexes unchecked, quirk Q10; a symbol outside the table contributes nothing here)
      if (sym >= 0 && sym < S) x += lp[((int64_t)k * N + j) * S + sym];
    }
    x *= normalize;
    if (ratios) x *= ratios[t];
    out[idx] = x;
    if (x > -1e20) atomicMin(first_good, (unsigned long long)t);
  }
}
// rows before the first emittable row are zeroed (_emission.pyx:73-80, quirk Q9)
__global__ void k_emission_fix(int64_t T, int N, double *out, const unsigned long long *first_good) {
  unsigned long long fg = *first_good;
  int64_t rows = fg > (unsigned long long)T ? T : (int64_t)fg;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < rows * N;
       idx += (int64_t)gridDim.x * blockDim.x)
    out[idx] = 0.0;
}

// ------------------------------------------------------------------------------------------
// Array-level forward / backward in log space, exact reference operation order
// (_hmm.pyx:120-198).  One wave; lane j owns states j, j+64, ...; any N.
// LDS: 2*N doubles.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_forward_log(int64_t T, int N, const double *pi,
                                                    const double *lt, const double *frame,
                                                    const double *ratios, double *fwd) {
  extern __shared__ double sm[];
  double *prev = sm, *next = sm + N;
  const int lane = threadIdx.x;
  const bool hasr = ratios != nullptr;
  {
    double r0 = hasr ? ratios[0] : 0.0;
    for (int j = lane; j < N; j += TEHMM_WAVE) {
      double a = pi[j] + frame[j];
      if (hasr && r0 > 1.) a += lt[(int64_t)j * N + j] * (r0 - 1.);
      fwd[j] = a;
      prev[j] = a;
    }
  }
  __syncthreads();
  for (int64_t t = 1; t < T; ++t) {
    double rt = hasr ? ratios[t] : 0.0;
    bool rr = hasr && rt > 1.;
    for (int j = lane; j < N; j += TEHMM_WAVE) {
      double add = rr ? lt[(int64_t)j * N + j] * (rt - 1.) : 0.0;
      double vmax = -INFINITY;
      for (int i = 0; i < N; ++i) {
        double w = prev[i] + lt[(int64_t)i * N + j];
        if (rr) w += add;
        if (w > vmax) vmax = w;
      }
      double ps = 0.0;
      for (int i = 0; i < N; ++i) {
        double w = prev[i] + lt[(int64_t)i * N + j];
        if (rr) w += add;
        ps += exp(w - vmax);
      }
      double v = log(ps) + vmax + frame[t * N + j];
      if (v <= -1e200) v = -INFINITY;
      fwd[t * N + j] = v;
      next[j] = v;
    }
    __syncthreads();
    double *tmp = prev; prev = next; next = tmp;
  }
}

__global__ __launch_bounds__(64) void k_backward_log(int64_t T, int N, const double *lt,
                                                     const double *frame, const double *ratios,
                                                     double *bwd) {
  extern __shared__ double sm[];
  double *prev = sm, *next = sm + N;   // prev holds frame[t+1][j] + bwd[t+1][j] pieces separately
  double *fb = sm + 2 * N;             // frame row t+1
  const int lane = threadIdx.x;
  const bool hasr = ratios != nullptr;
  {
    double v = log(1. / (double)N);
    for (int j = lane; j < N; j += TEHMM_WAVE) {
      bwd[(T - 1) * N + j] = v;
      prev[j] = v;
    }
  }
  __syncthreads();
  for (int64_t t = T - 2; t >= 0; --t) {
    double rt = hasr ? ratios[t + 1] : 0.0;
    bool rr = hasr && rt > 1.;
    for (int j = lane; j < N; j += TEHMM_WAVE) fb[j] = frame[(t + 1) * N + j];
    __syncthreads();
    for (int i = lane; i < N; i += TEHMM_WAVE) {
      double vmax = -INFINITY;
      for (int j = 0; j < N; ++j) {
        double w = lt[(int64_t)i * N + j] + fb[j] + prev[j];
        if (rr) w += lt[(int64_t)j * N + j] * (rt - 1.);
        if (w > vmax) vmax = w;
      }
      double ps = 0.0;
      for (int j = 0; j < N; ++j) {
        double w = lt[(int64_t)i * N + j] + fb[j] + prev[j];
        if (rr) w += lt[(int64_t)j * N + j] * (rt - 1.);
        ps += exp(w - vmax);
      }
      double v = log(ps) + vmax;
      if (v <= -1e200) v = -INFINITY;
      bwd[t * N + i] = v;
      next[i] = v;
    }
    __syncthreads();
    double *tmp = prev; prev = next; next = tmp;
  }
}

// ------------------------------------------------------------------------------------------
// Array-level xi log-sum (_hmm.pyx:62-117): one thread per (i, j) pair, sequential over t so the
// two-pass max / sum keeps the reference's accumulation order.
// ------------------------------------------------------------------------------------------
__global__ void k_xi_logsum(int64_t T, int N, const double *fwd, const double *lt,
                            const double *bwd, const double *frame, double logprob,
                            const double *ratios, double *out) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * N) return;
  int i = idx / N, j = idx - i * N;
  const bool hasr = ratios != nullptr;
  double lij = lt[idx], ljj = lt[(int64_t)j * N + j];
  double mx = -INFINITY;
  for (int64_t t = 0; t < T - 1; ++t) {
    double x = fwd[t * N + i] + lij + frame[(t + 1) * N + j] + bwd[(t + 1) * N + j] - logprob;
    if (hasr && ratios[t + 1] > 1.) {
      x += ljj * (ratios[t + 1] - 1.);
      if (i == j) {
        double y = fwd[(t + 1) * N + i] + bwd[(t + 1) * N + j] + log(ratios[t + 1] - 1.) - logprob;
        if (y > mx) mx = y;
      }
    }
    if (x > mx) mx = x;
  }
  double acc = out[idx];
  for (int64_t t = 0; t < T - 1; ++t) {
    double x = fwd[t * N + i] + lij + frame[(t + 1) * N + j] + bwd[(t + 1) * N + j] - logprob;
    if (hasr && ratios[t + 1] > 1.) {
      x += ljj * (ratios[t + 1] - 1.);
      if (i == j) {
        double y = fwd[(t + 1) * N + i] + bwd[(t + 1) * N + j] + log(ratios[t + 1] - 1.) - logprob;
        acc += exp(y - mx);
      }
    }
    acc += exp(x - mx);
  }
  out[idx] = log(acc) + mx;
}

// ------------------------------------------------------------------------------------------
// Array-level emission statistics (_emission.pyx:165-234): one thread per (track, state) row of
// obsStats, sequential over t (same accumulation order as the reference, no atomics).
// ------------------------------------------------------------------------------------------
template <typename ObsT>
__global__ void k_accumulate_obs(int64_t T, int K, int N, int S, const ObsT *obs,
                                 double *obsStats, const double *post, const double *ratios) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= K * N) return;
  int k = idx / N, j = idx - k * N;
  double *row = obsStats + (int64_t)idx * S;
  for (int64_t t = 0; t < T; ++t) {
    double p = post[t * N + j];
    if (ratios) p *= ratios[t];
    const int64_t sym = (int64_t)obs[t * K + k];
    if (sym >= 0 && sym < S) row[sym] += p;
  }
}

// ------------------------------------------------------------------------------------------
// Supervised emission counts (fastUpdateCounts -> _fastUpdateCountsU8/U16/32, _emission.pyx:236-332):
//   for pos in [start, end): obsStats[track][state][obs[pos][track]] += (ratios ? ratios[pos] : 1.0)
// for MANY labelled intervals at once (the reference calls it once per BED interval,
// emission.py:307-322).  Every cell of obsStats is owned by one thread, which walks the intervals in
// the given order and their positions ascending -- the reference's accumulation order, so the
// segment-ratio sums round identically.  grid = (K, N), threads = symbol values (strided).
//   iv_start / iv_end / iv_state: [n_iv] table-relative ranges and integer state labels
// ------------------------------------------------------------------------------------------
template <typename ObsT>
__global__ void k_update_counts(int n_iv, const int64_t *iv_start, const int64_t *iv_end,
                                const int32_t *iv_state, int K, int N, int S, const ObsT *obs,
                                double *obsStats, const double *ratios) {
  const int k = blockIdx.x, state = blockIdx.y;
  for (int sym = threadIdx.x; sym < S; sym += blockDim.x) {
    double acc = obsStats[((int64_t)k * N + state) * S + sym];
    bool touched = false;
    for (int i = 0; i < n_iv; ++i) {
      if (iv_state[i] != state) continue;
      for (int64_t pos = iv_start[i]; pos < iv_end[i]; ++pos) {
        if ((int64_t)obs[pos * K + k] == (int64_t)sym) {
          acc += ratios ? ratios[pos] : 1.0;
          touched = true;
        }
      }
    }
    if (touched) obsStats[((int64_t)k * N + state) * S + sym] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// Output reductions of teHmmEval (bin/teHmmEval.py:238-275).
//   k_post_masksum: out[r] = sum_j post[r][j] * mask[j]   (the --pd / --pdStates column, :270-272);
//                   one wave per row, lane = state.
//   k_bed_coords  : BED coordinates of every row of a (possibly segmented, possibly masked) table
//                   (:243-266): start = tableStart + d_i + maskOff[d_i], end = start + len_i with
//                   d_i = segOffsets[i] - segOffsets[0] (i for unsegmented tables) and
//                   len_i = TrackTable.getSegmentLength(i) (track.py:497-502).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_post_masksum(int64_t rows, int N, const double *post,
                                                      const double *mask, double *out) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double mk[2] = {lane < N ? mask[lane] : 0.0, lane + 64 < N ? mask[lane + 64] : 0.0};
  for (int64_t r = wid; r < rows; r += nw) {
    double g = lane < N ? post[r * N + lane] * mk[0] : 0.0;
    if (lane + 64 < N) g += post[r * N + lane + 64] * mk[1];
    g = wave_sum_f64(g);
    if (lane == 0) out[r] = g;
  }
}

__global__ void k_bed_coords(int64_t n_rows, int64_t table_start, int64_t table_end,
                             const int64_t *segOffsets, const int32_t *maskOffsets, int64_t *starts,
                             int64_t *ends) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t d = i, len = 1;
    if (segOffsets) {
      d = segOffsets[i] - segOffsets[0];
      len = i + 1 < n_rows ? segOffsets[i + 1] - segOffsets[i] : table_end - (table_start + segOffsets[i]);
    }
    int64_t st = table_start + d;
    if (maskOffsets) st += maskOffsets[d];
    starts[i] = st;
    ends[i] = st + len;
  }
}

// ------------------------------------------------------------------------------------------
// Viterbi (fused emission or frame input).  _hmm.pyx:201-259 operation order:
//   from = 0 : c = (V[t-1][0] + lt[0][to]) + b[t][to]; if ratios: c += lt[to][to]*r[t],
//              and if to == 0: c -= lt[0][0]                                  (quirk Q4)
//   from >= 1: c = (V[t-1][from] + lt[from][to]) + b[t][to]; if ratios and r[t] > 1:
//              c += lt[to][to]*(r[t]-1);  strict '>' keeps the lowest index on ties (Q5).
// Traceback pointers: one byte per (t, state): tb[(pos0 + t) * TBW + state].
// LDS: lt [N][NP] | ring [PB][64*SPL] | v [2][64*SPL]
// ------------------------------------------------------------------------------------------
template <int SPL, bool RATIO, bool FRAME>
__global__ __launch_bounds__(64) void k_viterbi(IntervalTab iv, EmisTab em, int N, int NP,
                                                const double *g_lt, const double *g_pi,
                                                const double *tratios, const double *frame,
                                                int TBW, uint8_t *tb, int *last_state,
                                                double *logprob) {
  extern __shared__ double sm[];
  constexpr int W = TEHMM_WAVE * SPL;
  double *LT = sm;
  double *ring = LT + (size_t)N * NP;
  double *vb = ring + TEHMM_PB * W;
  const int lane = threadIdx.x;
  const int id = iv.order[blockIdx.x];
  const int64_t T = iv.len[id];
  const int64_t p0 = iv.pos0[id];
  if (T <= 0) return;

  for (int idx = lane; idx < N * NP; idx += TEHMM_WAVE) LT[idx] = g_lt[idx];
  double ltd[SPL];
#pragma unroll
  for (int s = 0; s < SPL; ++s) {
    int j = lane + TEHMM_WAVE * s;
    ltd[s] = j < N ? g_lt[(size_t)j * NP + j] : 0.0;
  }
  const double lt00 = g_lt[0];
  __syncthreads();

  bool seen = false;
  int cur = 0;

  for (int64_t t0 = 0; t0 < T; t0 += TEHMM_PB) {
    const int np = (int)min((int64_t)TEHMM_PB, T - t0);
    // ---- phase 1: emission rows of this block into the LDS ring (independent of the chain)
    for (int p = 0; p < np; ++p) {
      double x[SPL];
      if (FRAME) {
#pragma unroll
        for (int s = 0; s < SPL; ++s) {
          int j = lane + TEHMM_WAVE * s;
          x[s] = j < N ? frame[(t0 + p) * N + j] : -INFINITY;
        }
      } else {
        emis_log<SPL>(em, p0 + t0 + p, lane, N, x);
        if (!seen) {   // _emission.pyx:73-80: only rows before the first emittable row
          double m = row_max<SPL>(x, lane, N);
          if (m > -1e20) seen = true;
          else {
#pragma unroll
            for (int s = 0; s < SPL; ++s) x[s] = 0.0;
          }
        }
      }
#pragma unroll
      for (int s = 0; s < SPL; ++s) ring[p * W + lane + TEHMM_WAVE * s] = x[s];
    }
    __syncthreads();
    // ---- phase 2: the sequential max-plus chain
    for (int p = 0; p < np; ++p) {
      const int64_t t = t0 + p;
      double b[SPL];
#pragma unroll
      for (int s = 0; s < SPL; ++s) b[s] = ring[p * W + lane + TEHMM_WAVE * s];
      double r = 0.0;
      if (RATIO) r = tratios[p0 + t];
      double *vprev = vb + cur * W;
      double *vnext = vb + (cur ^ 1) * W;
      if (t == 0) {
#pragma unroll
        for (int s = 0; s < SPL; ++s) {
          int j = lane + TEHMM_WAVE * s;
          if (j < N) {
            double v = g_pi[j] + b[s];
            if (RATIO && r > 1.) v += ltd[s] * (r - 1.);
            vnext[j] = v;
          }
        }
      } else {
        const bool rg = RATIO && r > 1.;
        const double rm1 = r - 1.;
#pragma unroll
        for (int s = 0; s < SPL; ++s) {
          int j = lane + TEHMM_WAVE * s;
          int jj = j < N ? j : 0;
          double best = (vprev[0] + LT[jj]) + b[s];
          if (RATIO) {
            best += ltd[s] * r;
            if (j == 0) best -= lt00;
          }
          int arg = 0;
          const double addr = ltd[s] * rm1;
          if (rg) {
#pragma unroll 4
            for (int from = 1; from < N; ++from) {
              double c = (vprev[from] + LT[from * NP + jj]) + b[s];
              c += addr;
              if (c > best) { best = c; arg = from; }
            }
          } else {
#pragma unroll 4
            for (int from = 1; from < N; ++from) {
              double c = (vprev[from] + LT[from * NP + jj]) + b[s];
              if (c > best) { best = c; arg = from; }
            }
          }
          if (j < N) vnext[j] = best;
          if (j < N) tb[(p0 + t) * TBW + j] = (uint8_t)arg;
        }
      }
      cur ^= 1;
      __syncthreads();
    }
  }
  // np.argmax over V[T-1] (first maximum; a NaN wins as soon as it is met)
  if (lane == 0) {
    const double *v = vb + cur * W;
    int last = 0;
    double m = v[0];
    if (m == m) {
      for (int j = 1; j < N; ++j) {
        double x = v[j];
        if (x != x) { last = j; break; }
        if (x > m) { m = x; last = j; }
      }
    }
    last_state[id] = last;
    logprob[id] = v[last];
  }
}

__device__ __forceinline__ int tb_get(const uint8_t *tb, int TBW, int64_t gpos, int s) {
  return (int)tb[gpos * TBW + s];
}

// Chunk c of an interval covers pointers t in (lo, hi], lo = c*C, hi = min((c+1)*C, T-1), and maps
// the state at hi to the state at lo.  compose: G[chunk][s_hi] = s_lo for every s_hi.
// Both walkers stage the chunk's pointer bytes in LDS first (one coalesced read of <= 256 x TBW bytes per
// stage), so that the dependent byte lookups of the walk are LDS reads (~64 cycles) instead of global ones
// (~500+ next to the throughput kernels): 5 ms -> well under 1 ms per 100 Mb for compose, likewise for fill.
// blockDim = 256: one chunk per wave.  Dynamic LDS: 4 x `stage` bytes, stage = tb_stage_bytes(TBW) (a whole
// chunk when it fits 16 KB: 9 KB at 36 states, so four workgroups share a CU).
#define TEHMM_TB_STAGE 16384       // upper bound of the bytes of pointer rows per wave and stage
__host__ __device__ inline int tb_stage_bytes(int TBW) {
  const int whole = TEHMM_TB_CHUNK * TBW;
  return whole < TEHMM_TB_STAGE ? whole : (TEHMM_TB_STAGE / TBW) * TBW;
}

__device__ __forceinline__ void tb_stage(const uint8_t *tb, int TBW, int64_t row0, int nrows, uint8_t *dst, int lane) {
  // rows row0 .. row0 + nrows - 1, contiguous in memory; TBW and the row base are multiples of 4
  const uint32_t *src = (const uint32_t *)(tb + row0 * TBW);
  uint32_t *d32 = (uint32_t *)dst;
  const int nd = nrows * TBW / 4;
  for (int i = lane; i < nd; i += TEHMM_WAVE) d32[i] = src[i];
}

__global__ __launch_bounds__(256) void k_tb_compose(IntervalTab iv, const int *chunk_iv,
                                                    const int64_t *chunk0, int n_chunks, int N, int NP, int TBW,
                                                    const uint8_t *tb, uint8_t *G) {
  extern __shared__ uint8_t tb_lds[];
  __builtin_amdgcn_s_setprio(TEHMM_PRIO_HI);      // latency kernel: issue ahead of co-resident throughput waves
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 4 + w;
  if (c >= n_chunks) return;
  const int stage = tb_stage_bytes(TBW);
  uint8_t *rows = tb_lds + w * stage;
  const int id = chunk_iv[c];
  const int64_t T = iv.len[id], p0 = iv.pos0[id];
  const int64_t cl = c - chunk0[id];
  const int64_t lo = cl * TEHMM_TB_CHUNK;
  const int64_t hi = min(lo + TEHMM_TB_CHUNK, T - 1);
  const int rps = stage / TBW;                          // rows per stage
  int s0 = lane, s1 = lane + TEHMM_WAVE;                // up to two states per lane (N <= 128)
  for (int64_t top = hi; top > lo; top -= rps) {
    const int64_t bot = max(lo, top - rps);             // pointers t in (bot, top]
    const int n = (int)(top - bot);
    __builtin_amdgcn_wave_barrier();
    tb_stage(tb, TBW, p0 + bot + 1, n, rows, lane);
    __builtin_amdgcn_wave_barrier();
    for (int r = n - 1; r >= 0; --r) {
      if (s0 < N) s0 = rows[r * TBW + s0];
      if (s1 < N) s1 = rows[r * TBW + s1];
    }
  }
  if (lane < N) G[(int64_t)c * NP + lane] = (uint8_t)s0;
  if (lane + TEHMM_WAVE < N) G[(int64_t)c * NP + lane + TEHMM_WAVE] = (uint8_t)s1;
}
// scan: sequential over the chunks of one interval (T/C dependent byte lookups)
// One wave per interval.  The chain over the interval's chunks (state at the chunk end -> state at its
// start) is a dependent walk; 64 chunk maps at a time are fetched cooperatively into LDS so that the walk
// itself only touches LDS (it used to pay one global-memory latency per chunk, ~16 ms next to an
// HBM-bound kernel for a 2 Mb interval).
__global__ __launch_bounds__(64) void k_tb_scan(IntervalTab iv, const int64_t *chunk0, int NP, const uint8_t *G,
                                                const int *last_state, uint8_t *bstate, int64_t *paths) {
  __builtin_amdgcn_s_setprio(TEHMM_PRIO_HI);      // latency kernel: issue ahead of co-resident throughput waves
  __shared__ uint8_t rows[64 * 136];   // NP <= 132
  __shared__ uint8_t bst[64];
  const int id = blockIdx.x;
  const int lane = threadIdx.x;
  if (id >= iv.n) return;
  const int64_t T = iv.len[id];
  if (T <= 0) return;
  int s = last_state[id];
  const int64_t c0 = chunk0[id], nc = chunk0[id + 1] - c0;
  if (nc == 0) {
    if (lane == 0) paths[iv.out0[id]] = s;
    return;
  }
  for (int64_t hi = nc; hi > 0; hi -= 64) {
    const int64_t lo = hi > 64 ? hi - 64 : 0;
    const int n = (int)(hi - lo);
    const uint8_t *src = G + (c0 + lo) * NP;
    for (int i = lane; i < n * NP; i += 64) rows[i] = src[i];
    __syncthreads();
    if (lane == 0) {
      for (int c = n - 1; c >= 0; --c) {
        bst[c] = (uint8_t)s;
        s = rows[c * NP + s];
      }
    }
    __syncthreads();
    s = __shfl(s, 0);
    if (lane < n) bstate[c0 + lo + lane] = bst[lane];
    __syncthreads();
  }
}
// Two-level form of the scan for long intervals (a 10 Mb interval has 39 k chunk maps: 1.7 ms of dependent LDS
// lookups in one wave).  Tiles of 64 chunk maps, counted from the END of the interval (tile ty covers the chunks
// [nc - 64 ty - 64, nc - 64 ty)), are composed in parallel (k_tb_group: lane = state at the tile's upper end), one wave
// per interval walks the tile maps (k_tb_scan_top), then every tile walks its own chunks from the state it was handed
// (k_tb_scan_tiles).  Tile storage of interval id starts at chunk0[id] / 64 + id (never overlapping).
__global__ __launch_bounds__(64) void k_tb_group(IntervalTab iv, const int64_t *chunk0, int NP, const uint8_t *G,
                                                 uint8_t *Gg) {
  __shared__ uint8_t rows[64 * 136];
  const int id = blockIdx.x, ty = blockIdx.y, lane = threadIdx.x;
  if (id >= iv.n || iv.len[id] <= 0) return;
  const int64_t c0 = chunk0[id], nc = chunk0[id + 1] - c0;
  const int64_t hi = nc - 64 * (int64_t)ty;
  if (hi <= 0) return;
  const int64_t lo = hi > 64 ? hi - 64 : 0;
  const int n = (int)(hi - lo);
  const uint8_t *src = G + (c0 + lo) * NP;
  for (int i = lane; i < n * NP; i += 64) rows[i] = src[i];
  __syncthreads();
  const int64_t tile = c0 / 64 + id + ty;
  for (int s0 = lane; s0 < NP; s0 += 64) {
    int s = s0;
    for (int c = n - 1; c >= 0; --c) s = rows[c * NP + s];
    Gg[tile * NP + s0] = (uint8_t)s;
  }
}
__global__ __launch_bounds__(64) void k_tb_scan_top(IntervalTab iv, const int64_t *chunk0, int NP, const uint8_t *Gg,
                                                    const int *last_state, uint8_t *tstate, int64_t *paths) {
  __builtin_amdgcn_s_setprio(TEHMM_PRIO_HI);
  __shared__ uint8_t rows[64 * 136];
  __shared__ uint8_t bst[64];
  const int id = blockIdx.x, lane = threadIdx.x;
  if (id >= iv.n || iv.len[id] <= 0) return;
  int s = last_state[id];
  const int64_t c0 = chunk0[id], nc = chunk0[id + 1] - c0;
  if (nc == 0) {
    if (lane == 0) paths[iv.out0[id]] = s;
    return;
  }
  const int64_t nt = (nc + 63) / 64, t0 = c0 / 64 + id;
  for (int64_t lo = 0; lo < nt; lo += 64) {            // tiles ascend = positions descend
    const int n = (int)min((int64_t)64, nt - lo);
    const uint8_t *src = Gg + (t0 + lo) * NP;
    for (int i = lane; i < n * NP; i += 64) rows[i] = src[i];
    __syncthreads();
    if (lane == 0) {
      for (int c = 0; c < n; ++c) {
        bst[c] = (uint8_t)s;
        s = rows[c * NP + s];
      }
    }
    __syncthreads();
    s = __shfl(s, 0);
    if (lane < n) tstate[t0 + lo + lane] = bst[lane];
    __syncthreads();
  }
}
__global__ __launch_bounds__(64) void k_tb_scan_tiles(IntervalTab iv, const int64_t *chunk0, int NP, const uint8_t *G,
                                                      const uint8_t *tstate, uint8_t *bstate) {
  __shared__ uint8_t rows[64 * 136];
  __shared__ uint8_t bst[64];
  const int id = blockIdx.x, ty = blockIdx.y, lane = threadIdx.x;
  if (id >= iv.n || iv.len[id] <= 0) return;
  const int64_t c0 = chunk0[id], nc = chunk0[id + 1] - c0;
  const int64_t hi = nc - 64 * (int64_t)ty;
  if (hi <= 0) return;
  const int64_t lo = hi > 64 ? hi - 64 : 0;
  const int n = (int)(hi - lo);
  const uint8_t *src = G + (c0 + lo) * NP;
  for (int i = lane; i < n * NP; i += 64) rows[i] = src[i];
  __syncthreads();
  if (lane == 0) {
    int s = tstate[c0 / 64 + id + ty];
    for (int c = n - 1; c >= 0; --c) {
      bst[c] = (uint8_t)s;
      s = rows[c * NP + s];
    }
  }
  __syncthreads();
  if (lane < n) bstate[c0 + lo + lane] = bst[lane];
}
// fill: one wave per chunk: pointer rows staged in LDS, lane 0 walks them and leaves the states in LDS, the
// wave writes the int64 path coalesced
__global__ __launch_bounds__(256) void k_tb_fill(IntervalTab iv, int n_chunks, const int *chunk_iv, const int64_t *chunk0,
                                                 int TBW, const uint8_t *tb, const uint8_t *bstate, int64_t *paths) {
  extern __shared__ uint8_t tb_lds[];
  __shared__ uint8_t walked[4][TEHMM_TB_CHUNK];
  __builtin_amdgcn_s_setprio(TEHMM_PRIO_HI);      // latency kernel: issue ahead of co-resident throughput waves
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 4 + w;
  if (c >= n_chunks) return;
  const int stage = tb_stage_bytes(TBW);
  uint8_t *rows = tb_lds + w * stage;
  const int id = chunk_iv[c];
  const int64_t T = iv.len[id], p0 = iv.pos0[id];
  const int64_t cl = c - chunk0[id];
  const int64_t lo = cl * TEHMM_TB_CHUNK;
  const int64_t hi = min(lo + TEHMM_TB_CHUNK, T - 1);
  int64_t *out = paths + iv.out0[id];
  int s = bstate[c];
  if (lane == 0) out[hi] = s;
  const int rps = stage / TBW;
  for (int64_t top = hi; top > lo; top -= rps) {
    const int64_t bot = max(lo, top - rps);             // pointers t in (bot, top] give the states at bot .. top - 1
    const int n = (int)(top - bot);
    __builtin_amdgcn_wave_barrier();
    tb_stage(tb, TBW, p0 + bot + 1, n, rows, lane);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      for (int r = n - 1; r >= 0; --r) {
        s = rows[r * TBW + s];
        walked[w][r] = (uint8_t)s;                      // state at position bot + r
      }
    }
    __builtin_amdgcn_wave_barrier();
    s = __shfl(s, 0);
    for (int r = lane; r < n; r += TEHMM_WAVE) out[bot + r] = (int64_t)walked[w][r];
  }
}

// ------------------------------------------------------------------------------------------
// Forward pass in the scaled linear domain (sum-product form of _hmm.pyx:120-158).
//   a_t[j] = bh_t[j] * sum_i a_{t-1}[i] * A[i][j],  bh_t = exp(b_t - max_j b_t)
// Scaling is by exact powers of two: a_t is multiplied by 2^-e where e is the binary exponent of
// max_j a_{t-1} (known before step t ends, so it is off the critical path); the exponents and the
// emission shifts are summed and give log P(obs) = log(sum_j a_{T-1}) + ln2*E + M.
// The scaled a_t rows are written to the posterior buffer; the backward kernel turns them into
// posteriors in place.  TRATIO: transition segment-ratio factor exp(lt[j][j]*(r_t-1)) for r_t > 1.
// LDS: A [N][NP] | ring [PB][W] | v [2][W] | ms [PB]
// ------------------------------------------------------------------------------------------
template <int SPL, bool TRATIO>
__global__ __launch_bounds__(64) void k_forward_lin(IntervalTab iv, EmisTab em, int N, int NP,
                                                    const double *g_A, const double *g_lt,
                                                    const double *g_pi, const double *tratios,
                                                    double *post, double *fwd_logprob,
                                                    int64_t *first_good) {
  extern __shared__ double sm[];
  constexpr int W = TEHMM_WAVE * SPL;
  double *A = sm;
  double *ring = A + (size_t)N * NP;
  double *vb = ring + TEHMM_PB * W;
  double *ms = vb + 2 * W;
  const int lane = threadIdx.x;
  const int id = iv.order[blockIdx.x];
  const int64_t T = iv.len[id];
  const int64_t p0 = iv.pos0[id];
  if (T <= 0) return;
  double *out = post + iv.out0[id] * N;

  for (int idx = lane; idx < N * NP; idx += TEHMM_WAVE) A[idx] = g_A[idx];
  double ltd[SPL];
#pragma unroll
  for (int s = 0; s < SPL; ++s) {
    int j = lane + TEHMM_WAVE * s;
    ltd[s] = j < N ? g_lt[(size_t)j * NP + j] : 0.0;
  }
  __syncthreads();

  bool seen = false;
  int64_t fg = T;
  int cur = 0;
  int eprev = 0;
  double Ecum = 0.0, Mcum = 0.0;
  double a[SPL];

  for (int64_t t0 = 0; t0 < T; t0 += TEHMM_PB) {
    const int np = (int)min((int64_t)TEHMM_PB, T - t0);
    for (int p = 0; p < np; ++p) {
      double x[SPL];
      emis_log<SPL>(em, p0 + t0 + p, lane, N, x);
      if (!seen) {
        double m0 = row_max<SPL>(x, lane, N);
        if (m0 > -1e20) { seen = true; fg = t0 + p; }
        else {
#pragma unroll
          for (int s = 0; s < SPL; ++s) x[s] = 0.0;
        }
      }
      if (TRATIO) {
        double r = tratios[p0 + t0 + p];
        if (r > 1.) {
#pragma unroll
          for (int s = 0; s < SPL; ++s) x[s] += ltd[s] * (r - 1.);
        }
      }
      double m = row_max<SPL>(x, lane, N);
#pragma unroll
      for (int s = 0; s < SPL; ++s) ring[p * W + lane + TEHMM_WAVE * s] = exp(x[s] - m);
      if (lane == 0) ms[p] = m;
    }
    __syncthreads();
    for (int p = 0; p < np; ++p) {
      const int64_t t = t0 + p;
      double *vprev = vb + cur * W;
      double *vnext = vb + (cur ^ 1) * W;
      Mcum += ms[p];
      if (t == 0) {
#pragma unroll
        for (int s = 0; s < SPL; ++s) {
          int j = lane + TEHMM_WAVE * s;
          a[s] = j < N ? exp(g_pi[j]) * ring[p * W + j] : 0.0;
        }
      } else {
#pragma unroll
        for (int s = 0; s < SPL; ++s) {
          int j = lane + TEHMM_WAVE * s;
          int jj = j < N ? j : 0;
          double acc0 = 0.0, acc1 = 0.0;
          int i = 0;
          for (; i + 1 < N; i += 2) {
            acc0 = fma(vprev[i], A[i * NP + jj], acc0);
            acc1 = fma(vprev[i + 1], A[(i + 1) * NP + jj], acc1);
          }
          if (i < N) acc0 = fma(vprev[i], A[i * NP + jj], acc0);
          double v = (acc0 + acc1) * ring[p * W + lane + TEHMM_WAVE * s];
          a[s] = j < N ? ldexp(v, -eprev) : 0.0;
        }
        Ecum += (double)eprev;
      }
      int e = -1022;
#pragma unroll
      for (int s = 0; s < SPL; ++s) {
        int j = lane + TEHMM_WAVE * s;
        if (j < N) {
          vnext[j] = a[s];
          out[t * N + j] = a[s];
          e = max(e, exp_of(a[s]));
        }
      }
      eprev = wave_max_i32(e);
      cur ^= 1;
      __syncthreads();
    }
  }
  double tot = 0.0;
#pragma unroll
  for (int s = 0; s < SPL; ++s) tot += a[s];
  tot = wave_sum_f64(tot);
  if (lane == 0) {
    fwd_logprob[id] = log(tot) + Ecum * 0.6931471805599453 + Mcum;
    first_good[id] = fg;
  }
}

// ------------------------------------------------------------------------------------------
// Backward pass in the scaled linear domain + posterior (sum-product form of _hmm.pyx:160-198
// and basehmm.py:265-272 / 516-517):
//   w_{t+1}[j] = bh_{t+1}[j] * beta_{t+1}[j];  beta_t[i] = sum_j A[i][j] * w_{t+1}[j]
//   post_t = normalise(a_t * beta_t);  EPS: += float32 eps, renormalise (score_samples only).
// AT is A transposed ([to][NP] rows indexed by from-lane).  The terminal beta is a constant
// (log(1/N) in the reference), which the per-row normalisation removes.
// LDS: AT [N][NP] | ring [PB][W] | w [2][W]
// ------------------------------------------------------------------------------------------
template <int SPL, bool TRATIO, bool EPS>
__global__ __launch_bounds__(64) void k_backward_lin(IntervalTab iv, EmisTab em, int N, int NP,
                                                     const double *g_AT, const double *g_lt,
                                                     const double *tratios, double *post,
                                                     const int64_t *first_good) {
  extern __shared__ double sm[];
  constexpr int W = TEHMM_WAVE * SPL;
  double *AT = sm;
  double *ring = AT + (size_t)N * NP;
  double *wb = ring + TEHMM_PB * W;
  const int lane = threadIdx.x;
  const int id = iv.order[blockIdx.x];
  const int64_t T = iv.len[id];
  const int64_t p0 = iv.pos0[id];
  if (T <= 0) return;
  double *out = post + iv.out0[id] * N;
  const int64_t fg = first_good[id];

  for (int idx = lane; idx < N * NP; idx += TEHMM_WAVE) AT[idx] = g_AT[idx];
  double ltd[SPL];
#pragma unroll
  for (int s = 0; s < SPL; ++s) {
    int j = lane + TEHMM_WAVE * s;
    ltd[s] = j < N ? g_lt[(size_t)j * NP + j] : 0.0;
  }
  __syncthreads();

  const double eps = 1.1920928955078125e-07;
  const double epsden = 1.0 + (double)N * eps;
  double beta[SPL];
  int cur = 0;
  int eprev = 0;

  // position T-1: beta = 1
  {
    double g[SPL], tot = 0.0;
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
      int j = lane + TEHMM_WAVE * s;
      beta[s] = j < N ? 1.0 : 0.0;
      g[s] = j < N ? out[(T - 1) * N + j] : 0.0;
      tot += g[s];
    }
    tot = wave_sum_f64(tot);
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
      int j = lane + TEHMM_WAVE * s;
      double pr = g[s] / tot;
      if (EPS) pr = (pr + eps) / epsden;
      if (j < N) out[(T - 1) * N + j] = pr;
    }
    eprev = 1;   // exponent of 1.0 in the frexp convention
  }

  // blocks of positions t = T-2 ... 0, each step needs bh'[t+1]
  for (int64_t thi = T - 2; thi >= 0; thi -= TEHMM_PB) {
    const int np = (int)min((int64_t)TEHMM_PB, thi + 1);
    for (int p = 0; p < np; ++p) {   // ring[p] <- bh'[(thi - p) + 1]
      const int64_t u = thi - p + 1;
      double x[SPL];
      emis_log<SPL>(em, p0 + u, lane, N, x);
      if (u < fg) {
#pragma unroll
        for (int s = 0; s < SPL; ++s) x[s] = 0.0;
      }
      if (TRATIO) {
        double r = tratios[p0 + u];
        if (r > 1.) {
#pragma unroll
          for (int s = 0; s < SPL; ++s) x[s] += ltd[s] * (r - 1.);
        }
      }
      double m = row_max<SPL>(x, lane, N);
#pragma unroll
      for (int s = 0; s < SPL; ++s) ring[p * W + lane + TEHMM_WAVE * s] = exp(x[s] - m);
    }
    __syncthreads();
    for (int p = 0; p < np; ++p) {
      const int64_t t = thi - p;
      double *wv = wb + cur * W;
      double av[SPL];
#pragma unroll
      for (int s = 0; s < SPL; ++s) {
        int j = lane + TEHMM_WAVE * s;
        av[s] = j < N ? out[t * N + j] : 0.0;          // scaled alpha row (independent load)
        if (j < N) wv[j] = ring[p * W + j] * beta[s];
      }
      __syncthreads();
      int e = -1022;
      double g[SPL], tot = 0.0;
#pragma unroll
      for (int s = 0; s < SPL; ++s) {
        int i = lane + TEHMM_WAVE * s;
        int ii = i < N ? i : 0;
        double acc0 = 0.0, acc1 = 0.0;
        int j = 0;
        for (; j + 1 < N; j += 2) {
          acc0 = fma(AT[j * NP + ii], wv[j], acc0);
          acc1 = fma(AT[(j + 1) * NP + ii], wv[j + 1], acc1);
        }
        if (j < N) acc0 = fma(AT[j * NP + ii], wv[j], acc0);
        beta[s] = i < N ? ldexp(acc0 + acc1, -eprev) : 0.0;
        e = max(e, exp_of(beta[s]));
        g[s] = av[s] * beta[s];
        tot += g[s];
      }
      eprev = wave_max_i32(e);
      tot = wave_sum_f64(tot);
#pragma unroll
      for (int s = 0; s < SPL; ++s) {
        int i = lane + TEHMM_WAVE * s;
        double pr = g[s] / tot;
        if (EPS) pr = (pr + eps) / epsden;
        if (i < N) out[t * N + i] = pr;
      }
      cur ^= 1;
    }
    __syncthreads();
  }
}


// ==========================================================================================
// "Wide" sequential kernels for 64 <= N <= 128 (BASELINE config 5): one workgroup of four waves per
// interval instead of one wave.  Wave w owns the from-states [w Q, (w + 1) Q), Q = NP / 4, for ALL to-states
// (two per lane) and keeps its quarter of the transition matrix in registers (2 x 33 doubles per lane: the LDS
// copy of the one-wave kernels cost two LDS reads per candidate); the state vector is read from LDS as a
// broadcast.  Per step: partial results -> LDS -> ONE workgroup barrier -> every wave combines the four
// partials (same arithmetic, so the four private copies of the new vector agree) -> next step.  The emission
// rows of a block of TEHMM_PB positions are computed by the four waves side by side.
// Arithmetic: Viterbi candidates are formed exactly as in k_viterbi and compared in ascending from-order (inside a
// wave, then wave 0..3 with a strict '>'), so path and score are bit-identical; forward / backward sum the four
// partial dot products pairwise (tolerance 1e-6, like every scaled-linear kernel here).
// LDS (doubles): ring [PB][128] | vown [4][2][128] | pval [2][4][128] | parg (int) [2][4][128] | ms [PB] (+8) |
//                ltab [lds_rows][NP]
// ==========================================================================================
// wave-wide max of an int / sum of a double without LDS shuffles (six dependent ds_bpermute round trips cost ~700
// cycles per step in the sequential kernels): xor butterfly inside the 16-lane rows by DPP, the rows by permlane swaps
__device__ __forceinline__ int wave_max_i32_dpp(int v) {
  v = max(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
  v = max(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
  v = max(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true));    // row_half_mirror
  v = max(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true));    // row_mirror
  const auto a = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  v = max((int)a[0], (int)a[1]);
  const auto c = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  return max((int)c[0], (int)c[1]);
}
__device__ __forceinline__ double wave_sum_f64_dpp(double v) {
#define TEHMM_DPP_ADD(CTRL)                                                                                  \
  do {                                                                                                       \
    const int lo_ = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);                       \
    const int hi_ = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);                       \
    v += __hiloint2double(hi_, lo_);                                                                         \
  } while (0)
  TEHMM_DPP_ADD(0xB1);
  TEHMM_DPP_ADD(0x4E);
  TEHMM_DPP_ADD(0x141);
  TEHMM_DPP_ADD(0x140);
#undef TEHMM_DPP_ADD
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const double t1 = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
  const unsigned lo1 = __double2loint(t1), hi1 = __double2hiint(t1);
  const auto c = __builtin_amdgcn_permlane16_swap(lo1, lo1, false, false);
  const auto d = __builtin_amdgcn_permlane16_swap(hi1, hi1, false, false);
  return __hiloint2double((int)d[0], (int)c[0]) + __hiloint2double((int)d[1], (int)c[1]);
}

#define TEHMM_WIDE_QM 33
#define TEHMM_WIDE_W 128
#define TEHMM_WIDE_FIXED (TEHMM_PB * TEHMM_WIDE_W + 8 * TEHMM_WIDE_W + 8 * TEHMM_WIDE_W + 4 * TEHMM_WIDE_W + TEHMM_PB + 8)
__host__ __device__ inline size_t wide_lds_bytes(int lds_rows, int NP) {
  return ((size_t)TEHMM_WIDE_FIXED + (size_t)lds_rows * NP) * sizeof(double);
}
__device__ __forceinline__ double *wide_stage_table(const EmisTab &em, double *sm) {
  double *ltab = sm + TEHMM_WIDE_FIXED;
  for (int i = threadIdx.x; i < em.lds_rows * em.NP; i += blockDim.x) ltab[i] = em.ltab_src[i];
  return ltab;
}

// Emission row of one position for the two states of a lane, small tracks from the LDS copy of their rows
// (ltab [lds_rows][NP], staged once per workgroup), the others from the global table (L2).  The rows of four
// tracks are requested together and then added in track order (the reference's operation order,
// _emission.pyx:65-72); the LDS / global choice is a pointer select (flat loads), not a branch.
__device__ __forceinline__ void emis_log_wide(const EmisTab &e, const double *ltab, int64_t gpos, int lane, int N,
                                              double (&x)[2]) {
  const uint32_t *row = e.obs32 + gpos * e.KPW;
  x[0] = 0.0;
  x[1] = 0.0;
  const int j0 = lane < N ? lane : 0, j1 = lane + 64 < N ? lane + 64 : 0;
  for (int k0 = 0; k0 < e.K; k0 += 4) {
    double v[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = min(k0 + u, e.K - 1);
      const uint32_t wd = row[k >> 2];
      const int sym = (int)((wd >> ((k & 3) * 8)) & 0xffu);
      const bool inr = sym < e.rowcnt[k];
      const int lb = e.ldsbase[k];
      const double *tr = lb >= 0 ? ltab + (size_t)(inr ? lb + sym : e.lds_zero) * e.NP
                                 : e.tab + (int64_t)(inr ? e.rowbase[k] + sym : e.zero_row) * e.NP;
      v[u][0] = tr[j0];
      v[u][1] = tr[j1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (k0 + u < e.K) {
        x[0] += v[u][0];
        x[1] += v[u][1];
      }
  }
  x[0] *= e.normalize;
  x[1] *= e.normalize;
  if (e.ratios) {
    const double r = e.ratios[gpos];
    x[0] *= r;
    x[1] *= r;
  }
}

// emission rows of one block: every wave computes the positions p = w (mod 4); before the first emittable row of
// the interval (`seen` false at the block start) every wave walks ALL positions in order so that the leading-rows
// quirk (_emission.pyx:73-80) stays sequential -- that only ever happens in the first blocks of an interval.
// MODE 0: log rows (Viterbi); MODE 1: exp(row - max) rows + ms (forward); fg = first emittable position.
template <int MODE, bool TRATIO>
__device__ __forceinline__ void wide_emission_block(const EmisTab &em, int64_t gpos0, int64_t t0, int np, int lane, int w,
                                                    int N, const double (&ltd)[2], const double *tratios, bool &seen,
                                                    int64_t &fg, double *ring, double *ms, const double *ltab) {
  const bool slow = !seen;
  for (int p = slow ? 0 : w; p < np; p += slow ? 1 : 4) {
    double x[2];
    emis_log_wide(em, ltab, gpos0 + p, lane, N, x);
    if (!seen) {
      const double m0 = row_max<2>(x, lane, N);
      if (m0 > -1e20) { seen = true; fg = t0 + p; }
      else { x[0] = 0.0; x[1] = 0.0; }
    }
    if (slow && (p & 3) != w) continue;
    if (MODE == 1) {
      if (TRATIO) {
        const double r = tratios[gpos0 + p];
        if (r > 1.) { x[0] += ltd[0] * (r - 1.); x[1] += ltd[1] * (r - 1.); }
      }
      const double m = row_max<2>(x, lane, N);
      ring[p * TEHMM_WIDE_W + lane] = exp(x[0] - m);
      ring[p * TEHMM_WIDE_W + lane + 64] = exp(x[1] - m);
      if (lane == 0) ms[p] = m;
    } else {
      ring[p * TEHMM_WIDE_W + lane] = x[0];
      ring[p * TEHMM_WIDE_W + lane + 64] = x[1];
    }
  }
}

template <bool RATIO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_viterbi_wide(IntervalTab iv, EmisTab em, int N, int NP, const double *g_lt, const double *g_pi,
                    const double *tratios, int TBW, uint8_t *tb, int *last_state, double *logprob) {
  extern __shared__ double sm[];
  constexpr int W = TEHMM_WIDE_W, QM = TEHMM_WIDE_QM;
  double *ring = sm;
  double *vown = ring + TEHMM_PB * W;
  double *pval = vown + 8 * W;
  int *parg = (int *)(pval + 8 * W);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int id = iv.order[blockIdx.x];
  const int64_t T = iv.len[id];
  const int64_t p0 = iv.pos0[id];
  if (T <= 0) return;
  const int Q = NP / 4, f0 = w * Q;
  double ltq[2][QM], ltd[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int j = lane + 64 * s;
    ltd[s] = j < N ? g_lt[(size_t)j * NP + j] : 0.0;
#pragma unroll
    for (int i = 0; i < QM; ++i) {
      const int f = f0 + i;
      ltq[s][i] = (i < Q && f < N && j < N) ? g_lt[(size_t)f * NP + j] : -INFINITY;
    }
  }
  const double lt00 = g_lt[0];
  double *vmine = vown + w * 2 * W;
  vmine[lane] = vmine[lane + 64] = vmine[W + lane] = vmine[W + lane + 64] = -INFINITY;
  const double *ltab = wide_stage_table(em, sm);
  __syncthreads();
  bool seen = false;
  int64_t fg = T;
  int cur = 0;
  for (int64_t t0 = 0; t0 < T; t0 += TEHMM_PB) {
    const int np = (int)min((int64_t)TEHMM_PB, T - t0);
    wide_emission_block<0, false>(em, p0 + t0, t0, np, lane, w, N, ltd, nullptr, seen, fg, ring, nullptr, ltab);
    __syncthreads();
    for (int p = 0; p < np; ++p) {
      const int64_t t = t0 + p;
      const double b[2] = {ring[p * W + lane], ring[p * W + lane + 64]};
      double r = 0.0;
      if (RATIO) r = tratios[p0 + t];
      const double *vp = vmine + cur * W;
      double *vn = vmine + (cur ^ 1) * W;
      if (t == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int j = lane + 64 * s;
          double v = (j < N ? g_pi[j] : -INFINITY) + b[s];
          if (RATIO && r > 1.) v += ltd[s] * (r - 1.);
          vn[j] = j < N ? v : -INFINITY;
        }
      } else {
        const bool rg = RATIO && r > 1.;
        const double rm1 = r - 1.;
        double best[2] = {-INFINITY, -INFINITY};
        int arg[2] = {0, 0};
        if (w == 0) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            double c = (vp[0] + ltq[s][0]) + b[s];
            if (RATIO) {
              c += ltd[s] * r;
              if (lane + 64 * s == 0) c -= lt00;
            }
            best[s] = c;
          }
        }
#pragma unroll
        for (int i = 0; i < QM; ++i) {
          if (i == 0 && w == 0) continue;                    // (uniform per wave)
          const double vf = vp[min(f0 + i, W - 1)];
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            double c = (vf + ltq[s][i]) + b[s];
            if (rg) c += ltd[s] * rm1;
            if (c > best[s]) { best[s] = c; arg[s] = f0 + i; }
          }
        }
        const int par = (int)(t & 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          pval[(par * 4 + w) * W + lane + 64 * s] = best[s];
          parg[(par * 4 + w) * W + lane + 64 * s] = arg[s];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int j = lane + 64 * s;
          double fin = pval[(par * 4 + 0) * W + j];
          int fa = parg[(par * 4 + 0) * W + j];
#pragma unroll
          for (int ww = 1; ww < 4; ++ww) {
            const double c = pval[(par * 4 + ww) * W + j];
            if (c > fin) { fin = c; fa = parg[(par * 4 + ww) * W + j]; }
          }
          vn[j] = j < N ? fin : -INFINITY;
          if (w == 0 && j < N) tb[(p0 + t) * TBW + j] = (uint8_t)fa;
        }
      }
      cur ^= 1;
    }
    __syncthreads();
  }
  // np.argmax over V[T-1] (first maximum; a NaN wins as soon as it is met)
  if (threadIdx.x == 0) {
    const double *v = vmine + cur * W;
    int last = 0;
    double m = v[0];
    if (m == m) {
      for (int j = 1; j < N; ++j) {
        const double x = v[j];
        if (x != x) { last = j; break; }
        if (x > m) { m = x; last = j; }
      }
    }
    last_state[id] = last;
    logprob[id] = v[last];
  }
}

template <bool TRATIO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_forward_wide(IntervalTab iv, EmisTab em, int N, int NP, const double *g_A, const double *g_lt,
                    const double *g_pi, const double *tratios, double *post, double *fwd_logprob,
                    int64_t *first_good) {
  extern __shared__ double sm[];
  constexpr int W = TEHMM_WIDE_W, QM = TEHMM_WIDE_QM;
  double *ring = sm;
  double *vown = ring + TEHMM_PB * W;
  double *pval = vown + 8 * W;
  double *ms = pval + 8 * W + 4 * W;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int id = iv.order[blockIdx.x];
  const int64_t T = iv.len[id];
  const int64_t p0 = iv.pos0[id];
  if (T <= 0) return;
  double *out = post + iv.out0[id] * N;
  const int Q = NP / 4, f0 = w * Q;
  double aq[2][QM], ltd[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int j = lane + 64 * s;
    ltd[s] = j < N ? g_lt[(size_t)j * NP + j] : 0.0;
#pragma unroll
    for (int i = 0; i < QM; ++i) {
      const int f = f0 + i;
      aq[s][i] = (i < Q && f < N && j < N) ? g_A[(size_t)f * NP + j] : 0.0;
    }
  }
  double *vmine = vown + w * 2 * W;
  vmine[lane] = vmine[lane + 64] = vmine[W + lane] = vmine[W + lane + 64] = 0.0;
  const double *ltab = wide_stage_table(em, sm);
  __syncthreads();
  bool seen = false;
  int64_t fg = T;
  int cur = 0, eprev = 0;
  double Ecum = 0.0, Mcum = 0.0;
  double a[2] = {0.0, 0.0};
  for (int64_t t0 = 0; t0 < T; t0 += TEHMM_PB) {
    const int np = (int)min((int64_t)TEHMM_PB, T - t0);
    wide_emission_block<1, TRATIO>(em, p0 + t0, t0, np, lane, w, N, ltd, tratios, seen, fg, ring, ms, ltab);
    __syncthreads();
    for (int p = 0; p < np; ++p) {
      const int64_t t = t0 + p;
      const double *vp = vmine + cur * W;
      double *vn = vmine + (cur ^ 1) * W;
      Mcum += ms[p];
      if (t == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int j = lane + 64 * s;
          a[s] = j < N ? exp(g_pi[j]) * ring[p * W + j] : 0.0;
        }
      } else {
        double acc[2] = {0.0, 0.0};
#pragma unroll
        for (int i = 0; i < QM; ++i) {
          const double vf = vp[min(f0 + i, W - 1)];
          acc[0] = fma(vf, aq[0][i], acc[0]);
          acc[1] = fma(vf, aq[1][i], acc[1]);
        }
        const int par = (int)(t & 1);
        pval[(par * 4 + w) * W + lane] = acc[0];
        pval[(par * 4 + w) * W + lane + 64] = acc[1];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int j = lane + 64 * s;
          const double sum = (pval[(par * 4 + 0) * W + j] + pval[(par * 4 + 1) * W + j]) +
                             (pval[(par * 4 + 2) * W + j] + pval[(par * 4 + 3) * W + j]);
          a[s] = j < N ? ldexp(sum * ring[p * W + j], -eprev) : 0.0;
        }
        Ecum += (double)eprev;
      }
      int e = -1022;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int j = lane + 64 * s;
        vn[j] = a[s];
        if (j < N) {
          if (w == 0) out[t * N + j] = a[s];
          e = max(e, exp_of(a[s]));
        }
      }
      eprev = wave_max_i32_dpp(e);
      cur ^= 1;
    }
    __syncthreads();
  }
  double tot = wave_sum_f64(a[0] + a[1]);
  if (threadIdx.x == 0) {
    fwd_logprob[id] = log(tot) + Ecum * 0.6931471805599453 + Mcum;
    first_good[id] = fg;
  }
}

template <bool TRATIO, bool EPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_backward_wide(IntervalTab iv, EmisTab em, int N, int NP, const double *g_AT, const double *g_lt,
                     const double *tratios, double *post, const int64_t *first_good) {
  extern __shared__ double sm[];
  constexpr int W = TEHMM_WIDE_W, QM = TEHMM_WIDE_QM;
  double *ring = sm;
  double *vown = ring + TEHMM_PB * W;
  double *pval = vown + 8 * W;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int id = iv.order[blockIdx.x];
  const int64_t T = iv.len[id];
  const int64_t p0 = iv.pos0[id];
  if (T <= 0) return;
  double *out = post + iv.out0[id] * N;
  const int64_t fg = first_good[id];
  const int Q = NP / 4, f0 = w * Q;
  double atq[2][QM], ltd[2];                  // atq[s][i] = A[state of slot s][f0 + i]
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int j = lane + 64 * s;
    ltd[s] = j < N ? g_lt[(size_t)j * NP + j] : 0.0;
#pragma unroll
    for (int i = 0; i < QM; ++i) {
      const int f = f0 + i;
      atq[s][i] = (i < Q && f < N && j < N) ? g_AT[(size_t)f * NP + j] : 0.0;
    }
  }
  double *wmine = vown + w * 2 * W;
  wmine[lane] = wmine[lane + 64] = 0.0;
  const double *ltab = wide_stage_table(em, sm);
  __syncthreads();
  const double eps = 1.1920928955078125e-07;
  const double epsden = 1.0 + (double)N * eps;
  double beta[2];
  int eprev = 0;
  {   // position T-1: beta = 1
    double g[2], tot = 0.0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int j = lane + 64 * s;
      beta[s] = j < N ? 1.0 : 0.0;
      g[s] = j < N ? out[(T - 1) * N + j] : 0.0;
      tot += g[s];
    }
    tot = wave_sum_f64(tot);
    __syncthreads();                          // every wave has read the alpha row before wave 0 overwrites it
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int j = lane + 64 * s;
      double pr = g[s] / tot;
      if (EPS) pr = (pr + eps) / epsden;
      if (w == 0 && j < N) out[(T - 1) * N + j] = pr;
    }
    eprev = 1;
  }
  for (int64_t thi = T - 2; thi >= 0; thi -= TEHMM_PB) {
    const int np = (int)min((int64_t)TEHMM_PB, thi + 1);
    for (int p = w; p < np; p += 4) {   // ring[p] <- bh'[(thi - p) + 1]
      const int64_t u = thi - p + 1;
      double x[2];
      emis_log_wide(em, ltab, p0 + u, lane, N, x);
      if (u < fg) { x[0] = 0.0; x[1] = 0.0; }
      if (TRATIO) {
        const double r = tratios[p0 + u];
        if (r > 1.) { x[0] += ltd[0] * (r - 1.); x[1] += ltd[1] * (r - 1.); }
      }
      const double m = row_max<2>(x, lane, N);
      ring[p * W + lane] = exp(x[0] - m);
      ring[p * W + lane + 64] = exp(x[1] - m);
    }
    __syncthreads();
    for (int p = 0; p < np; ++p) {
      const int64_t t = thi - p;
      double av[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int j = lane + 64 * s;
        av[s] = j < N ? out[t * N + j] : 0.0;          // scaled alpha row (independent load)
        wmine[j] = j < N ? ring[p * W + j] * beta[s] : 0.0;
      }
      double acc[2] = {0.0, 0.0};
#pragma unroll
      for (int i = 0; i < QM; ++i) {
        const double wf = wmine[min(f0 + i, W - 1)];
        acc[0] = fma(atq[0][i], wf, acc[0]);
        acc[1] = fma(atq[1][i], wf, acc[1]);
      }
      const int par = (int)(t & 1);
      pval[(par * 4 + w) * W + lane] = acc[0];
      pval[(par * 4 + w) * W + lane + 64] = acc[1];
      __syncthreads();
      int e = -1022;
      double g[2], tot = 0.0;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int j = lane + 64 * s;
        const double sum = (pval[(par * 4 + 0) * W + j] + pval[(par * 4 + 1) * W + j]) +
                           (pval[(par * 4 + 2) * W + j] + pval[(par * 4 + 3) * W + j]);
        beta[s] = j < N ? ldexp(sum, -eprev) : 0.0;
        e = max(e, exp_of(beta[s]));
        g[s] = av[s] * beta[s];
        tot += g[s];
      }
      eprev = wave_max_i32_dpp(e);
      tot = wave_sum_f64_dpp(tot);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int j = lane + 64 * s;
        double pr = g[s] / tot;
        if (EPS) pr = (pr + eps) / epsden;
        if (w == 0 && j < N) out[t * N + j] = pr;
      }
    }
    __syncthreads();
  }
}

}  // namespace tehmm
