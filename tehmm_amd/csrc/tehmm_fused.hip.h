// tehmm_fused.hip.h -- forward / backward lane passes with the emission rows AND the posterior combine
// fused in (round 2): the chunk-parallel posterior pipeline without its HBM intermediates.
//
// Round 1 streamed every position through HBM seven times for the posterior alone: k_emis_lane wrote the
// linear emission rows (288 B), k_fb_lane read them twice and wrote alpha' and beta' (2 x 288 B),
// k_combine_lane read both back to write the 280-byte posterior row.  Here
//   * the emission row of a position is recomputed where it is needed, as a PRODUCT of per-track table
//     rows in the linear domain:  q_t[j] = prod_k P[k][obs_t[k]][j],  P[k][s][j] = exp(lp[k][j][s] - c[k][s]),
//     c[k][s] = max_j lp[k][j][s]  (one multiply per track and state instead of an add and, per state, a
//     22-instruction exp); the per-position log-scale is sum_k c[k][obs_t[k]].  Small tracks' rows sit in
//     LDS, the 250-bin tracks' rows come from L2; the 10..12-byte observation row is the only stream read;
//   * the backward pass multiplies beta_t with the stored alpha'_t, normalises in registers and writes the
//     posterior row itself (+ the float32-eps quirk of score_samples, basehmm.py:271-272): beta' never
//     exists in memory, only one row in 64 is kept for the fix-up chain's direction check.
// Per position: forward 12 B in + 288 B out, backward 12 + 288 B in + 280 B out (was 2 938 B).
//
// Both passes run on the fp64 matrix cores in the transposed form of k_fb_mfma (tehmm_lane.hip.h):
// lane l owns item (l & 15) of its 16-item tile and the states (l >> 4) + 4 s, the accumulator layout of
// v_mfma_f64_16x16x4_f64 is the B-operand layout of the next step.  Tables are stored so that a lane's
// states are contiguous: row r, quarter kq -> KSP doubles [P[kq], P[kq + 4], ..., c] (the spare slot of the
// 16-byte padding carries c[k][s], so the log-scale costs no extra load).
//
// LOGDOM = true keeps the emission in the log domain (sum, * normalize, - row max, exp): needed when the
// model's normalizeFac != 1 (--emFac), where the product form does not apply.
#pragma once
#include "tehmm_lane.hip.h"

namespace tehmm {

template <int NT>
struct FusedGeom {
  static constexpr int KS = NT / 4;                       // states per lane
  static constexpr int KSP = ((KS + 1) + 1) & ~1;         // slice length in doubles (>= KS + 1, even)
  static constexpr int RT = (NT + 15) / 16;               // row tiles of the product
  static constexpr int SLICE_B = KSP * 8;                 // bytes per (row, quarter)
  static constexpr int ROW_D = 4 * KSP;                   // doubles per table row
};

struct FusedTab {
  const double *ptab;       // [(R + 1)][4][KSP] linear (or log, LOGDOM) rows + scale slot, global
  const double *ptab_lds;   // [lds_rows][4][KSP] the LDS-staged rows, packed like EmisTab::ltab_src
};

// the slice of table row `row` for this lane's quarter: KSP doubles into x[]
template <int NT>
__device__ __forceinline__ void fused_load_lds(const double *lds, int row, int kq, double (&x)[FusedGeom<NT>::KSP]) {
  using G = FusedGeom<NT>;
  lds_cd2 *p = (lds_cd2 *)(size_t)((unsigned)(size_t)(__attribute__((address_space(3))) const double *)lds +
                                   (unsigned)((row * 4 + kq) * G::SLICE_B));
#pragma unroll
  for (int i = 0; i < G::KSP / 2; ++i) {
    const d2v v = p[i];
    x[2 * i] = v.x;
    x[2 * i + 1] = v.y;
  }
}
template <int NT>
__device__ __forceinline__ void fused_load_glb(const double *tab, int row, int kq, double (&x)[FusedGeom<NT>::KSP]) {
  using G = FusedGeom<NT>;
  const double2 *p = (const double2 *)(tab + ((int64_t)row * 4 + kq) * G::KSP);
#pragma unroll
  for (int i = 0; i < G::KSP / 2; ++i) {
    const double2 v = p[i];
    x[2 * i] = v.x;
    x[2 * i + 1] = v.y;
  }
}

// Emission row of position gpos for this lane's KS states.  Linear form: q = product of the track rows,
// ms = sum of their scale slots.  LOGDOM: q = exp(normalize * sum - rowmax), ms = rowmax.
// A row no state can emit comes out as all zeros (the callers turn that into NaN).
template <int NT, bool LOGDOM>
__device__ __forceinline__ void fused_emission(const EmisTab &e, const FusedTab &ft, const double *lds, int64_t gpos,
                                               int kq, int N, double (&q)[FusedGeom<NT>::KS], double &ms) {
  using G = FusedGeom<NT>;
  const uint32_t *row = e.obs32 + gpos * e.KPW;
#pragma unroll
  for (int s = 0; s < G::KS; ++s) q[s] = LOGDOM ? 0.0 : 1.0;
  double sc = 0.0;
  uint32_t wn = row[0];
  for (int d = 0; d < e.KPW; ++d) {
    const uint32_t w = wn;
    if (d + 1 < e.KPW) wn = row[d + 1];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      const int k = 4 * d + bb;
      if (k < e.K) {
        const int sym = (int)((w >> (8 * bb)) & 0xffu);
        const bool inr = sym < e.rowcnt[k];
        const int lb = e.ldsbase[k];
        double x[G::KSP];
        if (lb >= 0) fused_load_lds<NT>(lds, inr ? lb + sym : e.lds_zero, kq, x);
        else fused_load_glb<NT>(ft.ptab, inr ? e.rowbase[k] + sym : e.zero_row, kq, x);
#pragma unroll
        for (int s = 0; s < G::KS; ++s) q[s] = LOGDOM ? q[s] + x[s] : q[s] * x[s];
        if (!LOGDOM) sc += x[G::KS];
      }
    }
  }
  if (LOGDOM) {
    double m = -INFINITY;
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
      q[s] *= e.normalize;
      m = fmax(m, kq + 4 * s < N ? q[s] : -INFINITY);
    }
    m = fmax(m, __shfl_xor(m, 16));
    m = fmax(m, __shfl_xor(m, 32));
    const bool good = m > -1e20;
#pragma unroll
    for (int s = 0; s < G::KS; ++s) q[s] = (good && kq + 4 * s < N) ? exp_nonpos(q[s] - m) : 0.0;
    ms = good ? m : 0.0;
  } else {
    ms = sc;
  }
}

// stage the LDS-resident table rows
template <int NT>
__device__ __forceinline__ void fused_stage(const EmisTab &e, const FusedTab &ft, double *lds) {
  const int n = e.lds_rows * FusedGeom<NT>::ROW_D;
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = ft.ptab_lds[i];
}

// ------------------------------------------------------------------------------------------
// Forward pass.  Same contract as k_fb_lane<NT, 0> / k_fb_mfma<NT, 0>: alpha' rows of the official range
// (item-interleaved), pre / end vectors, cumulative log-scale records; the emission rows are computed here.
// ------------------------------------------------------------------------------------------
template <int NT, bool LOGDOM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_fused_fwd(IntervalTab iv, EmisTab em, FusedTab ft, LaneGeom lg, int N, int CS, int Wu,
                 const double *__restrict__ tab /* A, [NT][NT] row-major */, double *rows, double *pre,
                 double *end, double *slog32) {
  using G = FusedGeom<NT>;
  constexpr int KS = G::KS, RT = G::RT;
  extern __shared__ double fused_lds[];
  fused_stage<NT>(em, ft, fused_lds);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= lg.n_groups * 4) return;
  const int L = lg.L;
  const int kq = lane >> 4;
  const int64_t item = (int64_t)tile * 16 + (lane & 15);
  const bool valid = item < lg.n_items;
  const int id = valid ? lg.item_iv[item] : 0;
  const int64_t t0 = valid ? lg.item_t0[item] : 0;
  const int64_t T = iv.len[id], p0 = iv.pos0[id];
  const int64_t ct0 = (t0 / CS) * CS;
  const bool run = valid && ct0 + CS <= T && ct0 > 0;
  if (!__any(run)) return;
  double tf[RT][KS];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int j = 16 * rt + (lane & 15), k = 4 * s + kq;
      tf[rt][s] = (j < NT) ? tab[k * NT + j] : 0.0;
    }
  }
  double v[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) v[s] = (kq + 4 * s < N) ? 1.0 / (double)N : 0.0;
  double slog = 0.0;
  const int64_t soff = (int64_t)kq << 6;
  const int64_t gbase = p0 + (run ? t0 : 0);       // lanes that do not run read some valid row
  auto vec_out = [&](double *dst) {
    const int64_t po = ((((item >> 6) * NT) + kq) << 6) + (item & 63);
#pragma unroll
    for (int s = 0; s < KS; ++s) dst[po + ((int64_t)(4 * s) << 6)] = v[s];
  };
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  double q[KS], ms;
  fused_emission<NT, LOGDOM>(em, ft, fused_lds, gbase + (run ? -Wu : 0), kq, N, q, ms);
  for (int s = -Wu; s < L; ++s) {
    if (s == 0) {
      if (run) vec_out(pre);
      slog = 0.0;
    }
    // the product goes to the matrix cores first; the next position's emission row is gathered meanwhile
    lane_d4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (lane_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < KS; ++k) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(tf[rt][k], v[k], acc[rt], 0, 0, 0);
    }
    double qn[KS], msn = 0.0;
    if (s + 1 < L) fused_emission<NT, LOGDOM>(em, ft, fused_lds, gbase + (run ? s + 1 : 0), kq, N, qn, msn);
    double a[KS];
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      a[k] = acc[k >> 2][k & 3] * q[k];
      t += a[k];
    }
    t += __shfl_xor(t, 16);
    t += __shfl_xor(t, 32);
    const int e = ((__double2hiint(t) >> 20) & 0x7ff) - 1022;
    // nothing can emit here / the product underflowed: poison the item (its links fail, the exact chain walks)
    const bool okrow = t > 1e-280 && t < INFINITY;
#pragma unroll
    for (int k = 0; k < KS; ++k) v[k] = okrow ? ldexp(a[k], -e) : qnan;
    slog += (double)e * 0.6931471805599453 + ms;
    if (s >= 0 && run) {
      const int64_t o = lane_row(lg, NT, item, s) + soff;
#pragma unroll
      for (int k = 0; k < KS; ++k) rows[o + ((int64_t)(4 * k) << 6)] = v[k];
      if ((s & 31) == 31 && kq == 0) slog32[item * (L / 32) + (s >> 5)] = slog;
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) q[k] = qn[k];
    ms = msn;
  }
  if (run) vec_out(end);
}

// ------------------------------------------------------------------------------------------
// Backward pass + posterior.  v = w_{t+1} = bh'_{t+1} * beta_{t+1} -> beta_t = normalise(A v); the posterior
// row normalise(alpha'_t * beta_t) (+ eps quirk) goes straight to post [T][N]; chk [item][L / 64][NT]
// keeps beta_t at the positions == 31 (mod 64), where k_fb_fix checks its direction.
// ------------------------------------------------------------------------------------------
template <int NT, bool LOGDOM, bool EPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_fused_bwd(IntervalTab iv, EmisTab em, FusedTab ft, LaneGeom lg, int N, int CS, int Wu,
                 const double *__restrict__ tab /* A, [NT][NT] row-major */, const double *__restrict__ alpha,
                 double *post, double *pre, double *end, double *chk) {
  using G = FusedGeom<NT>;
  constexpr int KS = G::KS, RT = G::RT;
  extern __shared__ double fused_lds[];
  fused_stage<NT>(em, ft, fused_lds);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= lg.n_groups * 4) return;
  const int L = lg.L;
  const int kq = lane >> 4;
  const int64_t item = (int64_t)tile * 16 + (lane & 15);
  const bool valid = item < lg.n_items;
  const int id = valid ? lg.item_iv[item] : 0;
  const int64_t t0 = valid ? lg.item_t0[item] : 0;
  const int64_t T = iv.len[id], p0 = iv.pos0[id];
  const int64_t ct0 = (t0 / CS) * CS;
  const bool run = valid && ct0 + CS <= T && ct0 + CS < T;
  if (!__any(run)) return;
  const int wu = !run ? 0 : (int)min((int64_t)Wu, T - (t0 + L));
  double tf[RT][KS];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int j = 16 * rt + (lane & 15), k = 4 * s + kq;
      tf[rt][s] = (j < NT) ? tab[j * NT + k] : 0.0;
    }
  }
  double v[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) v[s] = 0.0;
  const int64_t soff = (int64_t)kq << 6;
  const int64_t gbase = p0 + (run ? t0 : 0);
  const int64_t prow0 = (iv.out0[id] + t0) * N;       // posterior row of the item's first position
  auto vec_out = [&](double *dst) {
    const int64_t po = ((((item >> 6) * NT) + kq) << 6) + (item & 63);
#pragma unroll
    for (int s = 0; s < KS; ++s) dst[po + ((int64_t)(4 * s) << 6)] = v[s];
  };
  const double eps = 1.1920928955078125e-07;
  const double inv_epsden = 1.0 / (1.0 + (double)N * eps);
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  const int top = L + wu - 1;                          // first (highest) warm-up position of this item
  auto pos_of = [&](int s) { return s >= L ? max(min(s, top), L) : s; };
  double q[KS], ms;
  fused_emission<NT, LOGDOM>(em, ft, fused_lds, gbase + (run ? pos_of(L + Wu - 1) : 0), kq, N, q, ms);
  for (int s = L + Wu - 1; s >= 0; --s) {
    if (s == L - 1 && run) vec_out(pre);               // v = w_{t0+L} after the warm-up
    lane_d4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (lane_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < KS; ++k) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(tf[rt][k], v[k], acc[rt], 0, 0, 0);
    }
    double qn[KS], msn = 0.0;
    if (s > 0) fused_emission<NT, LOGDOM>(em, ft, fused_lds, gbase + (run ? pos_of(s - 1) : 0), kq, N, qn, msn);
    // alpha' row of this position (official range only), requested before the reduction that needs it
    double al[KS];
    const int64_t o = lane_row(lg, NT, item, s < L ? s : 0) + soff;
    if (s < L) {
#pragma unroll
      for (int k = 0; k < KS; ++k) al[k] = run ? alpha[o + ((int64_t)(4 * k) << 6)] : 0.0;
    }
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < KS; ++k) t += acc[k >> 2][k & 3];
    t += __shfl_xor(t, 16);
    t += __shfl_xor(t, 32);
    const int e = ((__double2hiint(t) >> 20) & 0x7ff) - 1022;
    double bt[KS];
    // an impossible emission row (all zeros) must poison the item: q enters v below, a zero v gives t = 0
    const bool okrow = (s >= L && s >= top) || (t > 0.0 && t < INFINITY);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      bt[k] = okrow ? ldexp(acc[k >> 2][k & 3], -e) : qnan;                // beta_t
      if (s >= L && s == top) bt[k] = kq + 4 * k < N ? 1.0 : 0.0;           // uniform start
    }
    if (s < L) {
      // posterior row: normalise(alpha' * beta) in registers, straight to post [T][N]
      double g[KS], gt = 0.0;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        g[k] = al[k] * bt[k];
        gt += g[k];
      }
      gt += __shfl_xor(gt, 16);
      gt += __shfl_xor(gt, 32);
      const double inv = 1.0 / gt;
      if (run) {
        double *pr = post + prow0 + (int64_t)s * N + kq;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          double p = g[k] * inv;
          if (EPS) p = (p + eps) * inv_epsden;
          if (kq + 4 * k < N) pr[4 * k] = p;
        }
        if ((s & 63) == 31) {
          double *cr = chk + (item * (L / 64) + (s >> 6)) * NT + kq;
#pragma unroll
          for (int k = 0; k < KS; ++k) cr[4 * k] = bt[k];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) v[k] = (s >= L && s > top) ? 0.0 : q[k] * bt[k];
#pragma unroll
    for (int k = 0; k < KS; ++k) q[k] = qn[k];
    ms = msn;
  }
  (void)ms;
  if (run) vec_out(end);
}

}  // namespace tehmm
