// tehmm_coop.hip.h -- cooperative ("one workgroup per interval") kernels: the latency-bound regime.
//
// A teHmmEval batch is a few dozen to a few hundred long intervals, and every DP over one interval
// is a strictly sequential chain (the Viterbi chain must even keep the reference's fp64 rounding
// order), so the time of a batch is (longest interval) x (cycles per chain step).  These kernels
// give every interval a whole workgroup of 4 wavefronts (one per SIMD of a CU) and strip the chain
// wave down to the bare recurrence:
//   * the chain wave runs only the value recurrence: lane = destination state, its column of the
//     transition table in registers; the previous state vector reaches every lane by DPP row
//     broadcasts out of registers (no LDS, no readlane on the recurrence: measured 16 cycles per
//     ds_read_b128 and 7 per v_readlane for a lone wave, against 4.7 for a DPP move);
//   * helper waves compute, one 64-position block ahead, the emission rows (lane = position, so
//     the table gathers of 64 positions are all in flight together; small tracks' tables are
//     staged in LDS) and, one block behind, whatever is parallel over positions: the Viterbi
//     arg-max / traceback pointers, recomputed from the stored value vectors in exactly the
//     reference's operation order.
// Blocks are handed over through LDS rings with one workgroup barrier per 64 positions.
// NT = number of states padded to a multiple of 4 (compile time); padding states have -inf
// transitions / zero probability.
#pragma once
#include "tehmm_kernels.hip.h"

namespace tehmm {

// Reading a whole LDS row with unrolled ds_read_b128: the row address is laundered through an
// opaque v_mov so that it stays in ONE VGPR and the reads become `ds_read_b128 v, vbase offset:imm`
// (one instruction each).
typedef double d2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const d2v lds_cd2;
__device__ __forceinline__ lds_cd2 *lds_row(const double *row) {
  unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const double *)row;
  asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(a));
  return (lds_cd2 *)(size_t)a;
}

// ---- state-vector broadcast without LDS -------------------------------------------------------
// The chain wave keeps the state vector with lane = state.  Every lane needs every element each
// step.  rep_rows() makes, for each 16-lane row R of the wave, a register in which ALL four rows
// hold row R's 16 values (two v_permlane32_swap + v_permlane16_swap per dword); bcast<F>() then
// reads element F with a single DPP move (row_newbcast: lane F&15 of the lane's own row), and the
// forward / backward products fold the broadcast into the FMA itself (v_fmac_f64_dpp).
template <int NT>
__device__ __forceinline__ void rep_rows(double v, double (&r)[(NT + 15) / 16]) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);   // [0]: rows 0,1,0,1  [1]: 2,3,2,3
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const auto c = __builtin_amdgcn_permlane16_swap(a[0], a[0], false, false); // [0]: row 0 x4   [1]: row 1 x4
  const auto d = __builtin_amdgcn_permlane16_swap(b[0], b[0], false, false);
  r[0] = __hiloint2double((int)d[0], (int)c[0]);
  if constexpr (NT > 16) r[1] = __hiloint2double((int)d[1], (int)c[1]);
  if constexpr (NT > 32) {
    const auto e = __builtin_amdgcn_permlane16_swap(a[1], a[1], false, false);   // row 2 x4, row 3 x4
    const auto f = __builtin_amdgcn_permlane16_swap(b[1], b[1], false, false);
    r[2] = __hiloint2double((int)f[0], (int)e[0]);
    if constexpr (NT > 48) r[3] = __hiloint2double((int)f[1], (int)e[1]);
  }
}
template <int L>
__device__ __forceinline__ double bcast(double v) {   // lane L (0..15) of the lane's own row
  return __longlong_as_double(
      __builtin_amdgcn_mov_dpp(__double_as_longlong(v), 0x150 + L, 0xf, 0xf, true));
}
// x[F] = V[F] + ltc[F] for F = 0..NT-1 (compile-time recursion: the DPP control is an immediate)
template <int F, int NT>
struct BcastAdd {
  static __device__ __forceinline__ void run(const double (&r)[(NT + 15) / 16], const double (&ltc)[NT],
                                             double (&x)[NT]) {
    x[F] = bcast<F & 15>(r[F >> 4]) + ltc[F];
    BcastAdd<F + 1, NT>::run(r, ltc, x);
  }
};
template <int NT>
struct BcastAdd<NT, NT> {
  static __device__ __forceinline__ void run(const double (&)[(NT + 15) / 16], const double (&)[NT],
                                             double (&)[NT]) {}
};
// s[F & 3] += V[F] * ac[F] with the broadcast folded into the FMA
template <int F, int NT>
struct BcastFma {
  static __device__ __forceinline__ void run(const double (&r)[(NT + 15) / 16], const double (&ac)[NT],
                                             double (&s)[4]) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(s[F & 3])
                 : "v"(r[F >> 4]), "v"(ac[F]), "n"(F & 15));
    BcastFma<F + 1, NT>::run(r, ac, s);
  }
};
template <int NT>
struct BcastFma<NT, NT> {
  static __device__ __forceinline__ void run(const double (&)[(NT + 15) / 16], const double (&)[NT],
                                             double (&)[4]) {}
};

// exp(y) for y <= 0 (the only range the scaled emission needs): k = rint(y/ln2),
// r = y - k*ln2 (two-piece ln2), degree-13 Taylor polynomial (|r| <= 0.347: truncation 4e-18),
// ldexp.  About 1 ulp; ~22 VALU instructions instead of the generic libm path.
__device__ __forceinline__ double exp_nonpos(double y) {
  const double k = rint(y * 1.4426950408889634);
  double r = fma(k, -6.93147180369123816490e-01, y);
  r = fma(k, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;            // 1/13!
  p = fma(p, r, 2.08767569878681e-09);          // 1/12!
  p = fma(p, r, 2.505210838544172e-08);         // 1/11!
  p = fma(p, r, 2.755731922398589e-07);         // 1/10!
  p = fma(p, r, 2.7557319223985893e-06);        // 1/9!
  p = fma(p, r, 2.48015873015873e-05);          // 1/8!
  p = fma(p, r, 1.984126984126984e-04);         // 1/7!
  p = fma(p, r, 1.388888888888889e-03);         // 1/6!
  p = fma(p, r, 8.333333333333333e-03);         // 1/5!
  p = fma(p, r, 4.1666666666666664e-02);        // 1/4!
  p = fma(p, r, 1.6666666666666666e-01);        // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const double res = ldexp(p, (int)k);
  return y < -1100.0 ? 0.0 : res;               // also maps -inf to 0; NaN stays NaN
}

// Emission log-likelihood of ONE position per lane, all NT (padded) states in registers.
// Operation order per state as in _emission.pyx:65-72.  Table rows are [NT] doubles (pads 0).
template <int NT>
__device__ __forceinline__ void emis_rows(const EmisTab &e, const double *ltab, int64_t gpos,
                                          double (&x)[NT]) {
  const uint32_t *row = e.obs32 + gpos * e.KPW;
#pragma unroll
  for (int j = 0; j < NT; ++j) x[j] = 0.0;
  uint32_t wn = row[0];
  for (int d = 0; d < e.KPW; ++d) {
    const uint32_t w = wn;
    if (d + 1 < e.KPW) wn = row[d + 1];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      const int k = 4 * d + bb;
      if (k < e.K) {
        const int sym = (int)((w >> (8 * bb)) & 0xffu);
        const bool inr = sym < e.rowcnt[k];       // beyond the track's symbols: the zero padding
        const int lb = e.ldsbase[k];
        if (lb >= 0) {
          lds_cd2 *tr = (lds_cd2 *)(size_t)((unsigned)(size_t)(
              __attribute__((address_space(3))) const double *)ltab +
              (unsigned)((inr ? lb + sym : e.lds_zero) * NT * 8));
#pragma unroll
          for (int jj = 0; jj < NT / 2; ++jj) {
            const d2v v = tr[jj];
            x[2 * jj] += v.x;
            x[2 * jj + 1] += v.y;
          }
        } else {
          const double2 *tr = (const double2 *)(e.tab + (int64_t)(inr ? e.rowbase[k] + sym : e.zero_row) * NT);
#pragma unroll
          for (int jj = 0; jj < NT / 2; ++jj) {
            const double2 v = tr[jj];
            x[2 * jj] += v.x;
            x[2 * jj + 1] += v.y;
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) x[j] *= e.normalize;
  if (e.ratios) {
    const double r = e.ratios[gpos];
#pragma unroll
    for (int j = 0; j < NT; ++j) x[j] *= r;
  }
}

// One block of up to 64 emission rows (lane = position) into an LDS ring, row stride RS doubles.
//   LINEAR=false: log rows (Viterbi).   LINEAR=true: exp(row - rowmax) and rowmax to ms[].
//   ASCENDING=true: blocks arrive in increasing t and the leading-rows quirk (_emission.pyx:73-80)
//   is tracked in `seen`; a later impossible row sets *dead (the reference's lattices turn NaN).
//   ASCENDING=false: every impossible row is zeroed (correct unless *dead gets set by the
//   ascending pass of the same interval).
template <int NT, bool LINEAR, bool ASCENDING, bool TRATIO>
__device__ __forceinline__ void emis_block(const EmisTab &e, const double *ltab, int64_t gbase,
                                           int np, int lane, int N, bool &seen, int *dead,
                                           const double *ltdiag, const double *tratios,
                                           double *ring, int RS, double *ms) {
  const bool act = lane < np;
  const int64_t gpos = gbase + (act ? lane : np - 1);
  double x[NT];
  emis_rows<NT>(e, ltab, gpos, x);
  // pads (j >= N) hold 0.0 (table pads are zero); mask them out of the row maximum once
  double m = x[0];
#pragma unroll
  for (int j = 1; j < NT; ++j) m = fmax(m, j < N ? x[j] : -INFINITY);
  const bool good = m > -1e20;
  bool zero;
  if (ASCENDING) {
    const unsigned long long gm = __ballot(good && act);
    const int first = gm ? __ffsll((long long)gm) - 1 : 64;
    zero = !seen && lane < first;
    if (!good && !zero && act && dead) *dead = 1;
    if (gm) seen = true;
  } else {
    zero = !good;
  }
  if (zero) {
#pragma unroll
    for (int j = 0; j < NT; ++j) x[j] = 0.0;
    m = 0.0;
  }
  if (TRATIO) {
    const double r = tratios[gpos];
    if (r > 1.) {
      m = -INFINITY;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        x[j] += ltdiag[j] * (r - 1.);
        m = fmax(m, j < N ? x[j] : -INFINITY);
      }
    }
  }
  if (act) {
    double *dst = ring + lane * RS;
    if (LINEAR) {
#pragma unroll
      for (int j = 0; j < NT; ++j) dst[j] = j < N ? exp_nonpos(x[j] - m) : 0.0;
      ms[lane] = m;
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j) dst[j] = x[j];
    }
  }
}

// copies the LDS-resident part of the emission table (rows [0, lds_rows) of e.ltab_src)
__device__ __forceinline__ void stage_emis_table(const EmisTab &e, double *ltab, int NT) {
  const int n = e.lds_rows * NT;
  for (int i = threadIdx.x; i < n; i += blockDim.x) ltab[i] = e.ltab_src[i];
}

// ------------------------------------------------------------------------------------------
// Cooperative Viterbi.  blockDim = 256.  Byte traceback table tb[(pos0 + t) * NT + state].
//   wave 0      : value chain.  V_t[to] = max( c0 , max_{from>=1} x_from (+b, +ratio term) ) where
//                 x_from = V_{t-1}[from] + lt[from][to]; hoisting "+ b" out of the max is exact
//                 because fp64 rounding is monotone (the ARG-max is not hoisted: see below).
//   wave 1      : emission rows of block it;
//   waves 2, 3  : the exact arg-max of block it-2 (lane = position, destination states split):
//                 c_from recomputed in the reference's order (V+lt)+b [+ratio term],
//                 arg = lowest from with c_from == V_t[to] (== the reference's strict '>' scan).
// LDS (doubles): bring [3][CPB][RS] | Vring [2][CPB+1][VS] | ltab [lds_rows][NT]
// ------------------------------------------------------------------------------------------
template <int NT, int CPB, bool RATIO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_vit_coop(IntervalTab iv, EmisTab em, int N, const double *g_lt, const double *g_ltT,
                const double *g_pi, const double *tratios, uint8_t *tb, int *last_state, double *logprob) {
  extern __shared__ double sm[];
  constexpr int RS = NT + 1;
  constexpr int VS = NT + 2;   // V row stride: conflict-free per-lane ds_read_b128 in the arg pass
  double *bring = sm;
  double *Vring = bring + 3 * CPB * RS;
  double *ltab = Vring + 2 * (CPB + 1) * VS;
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int id = iv.order[blockIdx.x];
  const int64_t T = iv.len[id];
  const int64_t p0 = iv.pos0[id];
  if (T <= 0) return;
  stage_emis_table(em, ltab, NT);
  const int jl = min(lane, NT - 1);
  const bool live = lane < N;
  ST_DECL;
  __syncthreads();
  const int64_t nb = (T + CPB - 1) / CPB;
  // Every role runs its own loop over the block iterations (so that its registers are not live in
  // the other roles' code); all of them execute the same nb + 2 workgroup barriers.
  if (w == 0) {
    // ============================================================== value chain (block it-1)
    double ltc[NT];   // column `lane` of lt; pad lanes: -inf so their value stays -inf unmasked
#pragma unroll
    for (int f = 0; f < NT; ++f) ltc[f] = live ? g_lt[f * NT + jl] : -INFINITY;
    const double ltd = live ? g_lt[jl * NT + jl] : 0.0;
    const double lt00 = g_lt[0];
    const double pij = live ? g_pi[jl] : -INFINITY;
    const int vslot = lane < NT ? lane : NT + 1;
    double vcur = -INFINITY;
    for (int64_t it = 0; it < nb + 2; ++it) {
      ST_BEGIN;
      const int64_t bn = it - 1;
      if (bn >= 0 && bn < nb) {
        const int np = (int)min((int64_t)CPB, T - bn * CPB);
        double *Vr = Vring + (bn & 1) * (CPB + 1) * VS;
        const double *br = bring + (bn % 3) * CPB * RS;
        if (bn > 0) Vr[vslot] = vcur;
        for (int p = 0; p < np; ++p) {
          const int64_t t = bn * CPB + p;
          const double b = br[p * RS + jl];
          double r = 0.0;
          if (RATIO) r = tratios[p0 + t];
          double v;
          if (t == 0) {
            v = pij + b;
            if (RATIO && r > 1.) v += ltd * (r - 1.);
          } else {
            // x_from = V[from] + lt[from][to]: V[from] by DPP row broadcast (no LDS on the chain),
            // NT independent adds, then a max tree
            double rr[(NT + 15) / 16];
            rep_rows<NT>(vcur, rr);
            double x[NT];
            BcastAdd<0, NT>::run(rr, ltc, x);
            double c0 = x[0] + b;
            x[0] = -INFINITY;
#pragma unroll
            for (int n = NT; n > 1; n = (n + 1) / 2) {
#pragma unroll
              for (int i = 0; i < n / 2; ++i) x[i] = fmax(x[i], x[i + (n + 1) / 2]);
            }
            if (RATIO) {
              c0 += ltd * r;
              if (lane == 0) c0 -= lt00;
            }
            double c1 = x[0] + b;
            if (RATIO && r > 1.) c1 += ltd * (r - 1.);
            v = c1 > c0 ? c1 : c0;      // c0 NaN stays (reference: nothing is '>' a NaN best)
          }
          vcur = v;
          Vr[(p + 1) * VS + vslot] = v;     // lanes >= NT hold -inf and share the pad slot
        }
      }
      ST_ADD(st_a);
      __syncthreads();
      ST_ADD(st_c);
    }
    ST_FLUSH(w);
    double *scratch = Vring;
    if (lane < NT) scratch[lane] = vcur;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      int last = 0;
      double m = scratch[0];
      if (m == m) {
        for (int j = 1; j < N; ++j) {
          double x = scratch[j];
          if (x != x) { last = j; break; }
          if (x > m) { m = x; last = j; }
        }
      }
      last_state[id] = last;
      logprob[id] = scratch[last];
    }
  } else if (w == 1) {
    // ============================================================== emission rows (block it)
    bool seen = false;
    for (int64_t it = 0; it < nb + 2; ++it) {
      ST_BEGIN;
#ifdef TEHMM_DBG_NOEMIS
      if (it < 3) {
#else
      if (it < nb) {
#endif
        const int np = (int)min((int64_t)CPB, T - it * CPB);
        emis_block<NT, false, true, false>(em, ltab, p0 + it * CPB, np, lane, N, seen, nullptr,
                                           nullptr, nullptr, bring + (it % 3) * CPB * RS, RS,
                                           nullptr);
      }
      ST_ADD(st_a);
      __syncthreads();
      ST_ADD(st_c);
    }
    ST_FLUSH(w);
  } else {
    // ============================================================== exact arg-max (block it-2)
    // lane = position; the two waves split the destination states.  For every (position, to):
    // c_from recomputed in the reference's order (V+lt)+b [+ratio term] with the lt column as
    // scalar operands; arg = lowest from with c_from == V_t[to] (== the reference's strict '>'
    // ascending scan, because V_t[to] is the maximum of the c_from).
    const int half = (N + 1) / 2;
    const int to_lo = (w - 2) * half, to_hi = min(N, to_lo + half);
    const double lt00 = g_lt[0];
    for (int64_t it = 0; it < nb + 2; ++it) {
      ST_BEGIN;
      const int64_t ba = it - 2;
#ifdef TEHMM_DBG_NOARGS
      if (ba >= 0 && ba < 1) {
#else
      if (ba >= 0 && ba < nb) {
#endif
        const int np = (int)min((int64_t)CPB, T - ba * CPB);
        const double *Vr = Vring + (ba & 1) * (CPB + 1) * VS;
        const double *br = bring + (ba % 3) * CPB * RS;
        const int pl = lane < np ? lane : 0;
        const int64_t t = ba * CPB + pl;
        const bool actp = lane < np && t > 0;
        lds_cd2 *vp = lds_row(Vr + pl * VS);          // V[t-1][*] of this lane's position
        d2v pv[NT / 2];
#pragma unroll
        for (int f2 = 0; f2 < NT / 2; ++f2) pv[f2] = vp[f2];
        const double *vnext = Vr + (pl + 1) * VS;     // V[t][*]
        const double *brow = br + pl * RS;
        double r = 0.0;
        if (RATIO) r = tratios[p0 + t];
        const bool rg = RATIO && r > 1.;
        for (int to = to_lo; to < to_hi; ++to) {
          const double *lc = g_ltT + to * NT;           // lt[*][to], wave-uniform
          const double vt = vnext[to];
          const double b = brow[to];
          double c[NT];
#pragma unroll
          for (int f2 = 0; f2 < NT / 2; ++f2) {
            c[2 * f2] = (pv[f2].x + lc[2 * f2]) + b;
            c[2 * f2 + 1] = (pv[f2].y + lc[2 * f2 + 1]) + b;
          }
          if (RATIO) {
            const double ld = lc[to];
            c[0] += ld * r;
            if (to == 0) c[0] -= lt00;
            const double addr = rg ? ld * (r - 1.) : 0.0;   // x + 0.0 == x for the equality below
#pragma unroll
            for (int f = 1; f < NT; ++f) c[f] += addr;
          }
          int arg = 0;
#pragma unroll
          for (int f = NT - 1; f >= 1; --f) arg = c[f] == vt ? f : arg;
          arg = c[0] == vt ? 0 : arg;
          if (actp) tb[(p0 + t) * NT + to] = (uint8_t)arg;
        }
      }
      ST_ADD(st_b);
      __syncthreads();
      ST_ADD(st_c);
    }
    ST_FLUSH(w);
  }
}

// ------------------------------------------------------------------------------------------
// Cooperative forward + backward (scaled linear domain, see k_forward_lin / k_backward_lin).
// blockDim = 256: wave 0 forward chain, wave 1 backward chain, wave 2 / 3 their emission rows.
// alpha rows go to `alpha` ([total][N], the posterior buffer), beta rows to `beta`.
// Lane NT-1 is always a padding lane (N < NT): it carries a column of ones, so the same FMA loop
// leaves in it the SUM of the previous vector, whose binary exponent is the power-of-two scale of
// the step -- no cross-lane reduction on the chain.
// LDS (doubles): ringF [2][CPB][RS] | ringB [2][CPB][RS] | msF [2][CPB] | msB [2][CPB] |
//                xF [2][NT] | xB [2][NT] | ltd [NT] | ltab [lds_rows][NT]
// ------------------------------------------------------------------------------------------
template <int NT, int CPB, bool TRATIO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_fb_coop(IntervalTab iv, EmisTab em, int N, const double *g_A, const double *g_lt,
               const double *g_pi, const double *tratios, double *alpha, double *beta,
               double *fwd_logprob, int *dead_flag, double *wrows, int *escale) {
  extern __shared__ double sm[];
  constexpr int RS = NT + 1;
  double *ringF = sm;
  double *ringB = ringF + 2 * CPB * RS;
  double *msF = ringB + 2 * CPB * RS;
  double *msB = msF + 2 * CPB;
  double *xF = msB + 2 * CPB;
  double *xB = xF + 2 * NT;
  double *ltdv = xB + 2 * NT;
  double *ltab = ltdv + NT;
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int id = iv.order[blockIdx.x];
  const int64_t T = iv.len[id];
  const int64_t p0 = iv.pos0[id];
  if (T <= 0) return;
  const int jl = min(lane, NT - 1);
  const bool live = lane < N;
  if (threadIdx.x < NT) ltdv[threadIdx.x] = g_lt[threadIdx.x * NT + threadIdx.x];
  stage_emis_table(em, ltab, NT);
  ST_DECL;
  __syncthreads();
  const int64_t nb = (T + CPB - 1) / CPB;
  if (w == 0) {
    // ============================================================== forward chain (block it-1)
    double ac[NT];   // column jl of A (A[i][jl]); pad lane NT-1: ones
#pragma unroll
    for (int i = 0; i < NT; ++i) ac[i] = live ? g_A[i * NT + jl] : (lane == NT - 1 ? 1.0 : 0.0);
    double *al = alpha + iv.out0[id] * N;
    int *es = escale ? escale + iv.out0[id] : nullptr;     // E-step: exponent applied at each step
    double Ecum = 0.0, Mcum = 0.0, a = 0.0;
    for (int64_t it = 0; it < nb + 1; ++it) {
      ST_BEGIN;
      const int64_t bn = it - 1;
      if (bn >= 0) {
        const int np = (int)min((int64_t)CPB, T - bn * CPB);
        const double *br = ringF + (bn & 1) * CPB * RS;
        const double *mr = msF + (bn & 1) * CPB;
        for (int p = 0; p < np; ++p) {
          const int64_t t = bn * CPB + p;
          const double bh = br[p * RS + jl];
          Mcum += mr[p];
          if (t == 0) {
            a = live ? exp(g_pi[jl]) * bh : 0.0;
          } else {
            double rr[(NT + 15) / 16];
            rep_rows<NT>(a, rr);
            double sacc[4] = {0.0, 0.0, 0.0, 0.0};
            asm volatile("s_nop 1" ::: "memory");   // VALU write -> DPP read wait states
            BcastFma<0, NT>::run(rr, ac, sacc);
            const double ssum = (sacc[0] + sacc[1]) + (sacc[2] + sacc[3]);
            // lane NT-1 holds sum_i a_{t-1}[i]: its exponent scales this step
            const int e = ((__builtin_amdgcn_readlane(__double2hiint(ssum), NT - 1) >> 20) & 0x7ff) - 1022;
            a = live ? ldexp(ssum * bh, -e) : 0.0;
            Ecum += (double)e;
            if (es && lane == 0) es[t] = e;
          }
          if (live) al[t * N + lane] = a;
        }
      }
      ST_ADD(st_a);
      __syncthreads();
      ST_ADD(st_c);
    }
    ST_FLUSH(w);
    double tot = wave_sum_f64(live ? a : 0.0);
    if (lane == 0) fwd_logprob[id] = log(tot) + Ecum * 0.6931471805599453 + Mcum;
  } else if (w == 1) {
    // ============================================================== backward chain (block it-1)
    double ac[NT];   // row jl of A (A[jl][j]); pad lane NT-1: ones
#pragma unroll
    for (int j = 0; j < NT; ++j) ac[j] = live ? g_A[jl * NT + j] : (lane == NT - 1 ? 1.0 : 0.0);
    double *be = beta + iv.out0[id] * N;
    double *wr = wrows ? wrows + iv.out0[id] * N : nullptr;   // E-step: w_u = bh'_u * beta_u rows
    double bt = 0.0, wv = 0.0;
    for (int64_t it = 0; it < nb + 1; ++it) {
      ST_BEGIN;
      const int64_t bn = it - 1;
      if (bn >= 0) {
        const int64_t uhi = T - bn * CPB;
        const int64_t ulo = max((int64_t)0, uhi - CPB);
        const double *br = ringB + (bn & 1) * CPB * RS;
        for (int64_t u = uhi - 1; u >= ulo; --u) {
          // w[u+1] = bh'[u+1] * beta[u+1] (lane = state) feeds beta[u]; beta[T-1] = 1
          const double bh = br[(int)(u - ulo) * RS + jl];
          if (u == T - 1) {
            bt = live ? 1.0 : 0.0;
          } else {
            double rr[(NT + 15) / 16];
            rep_rows<NT>(wv, rr);
            double sacc[4] = {0.0, 0.0, 0.0, 0.0};
            asm volatile("s_nop 1" ::: "memory");
            BcastFma<0, NT>::run(rr, ac, sacc);
            const double ssum = (sacc[0] + sacc[1]) + (sacc[2] + sacc[3]);
            const int e = ((__builtin_amdgcn_readlane(__double2hiint(ssum), NT - 1) >> 20) & 0x7ff) - 1022;
            bt = live ? ldexp(ssum, -e) : 0.0;
          }
          if (live) be[u * N + lane] = bt;
          wv = live ? bh * bt : 0.0;
          if (wr && live) wr[u * N + lane] = wv;
        }
      }
      ST_ADD(st_a);
      __syncthreads();
      ST_ADD(st_c);
    }
    ST_FLUSH(w);
  } else if (w == 2) {
    // ============================================================== forward emission (block it)
    bool seen = false;
    for (int64_t it = 0; it < nb + 1; ++it) {
      ST_BEGIN;
      if (it < nb) {
        const int np = (int)min((int64_t)CPB, T - it * CPB);
        emis_block<NT, true, true, TRATIO>(em, ltab, p0 + it * CPB, np, lane, N, seen, dead_flag + id,
                                           ltdv, tratios, ringF + (it & 1) * CPB * RS, RS,
                                           msF + (it & 1) * CPB);
      }
      ST_ADD(st_a);
      __syncthreads();
      ST_ADD(st_c);
    }
    ST_FLUSH(w);
  } else {
    // ============================================================== backward emission (block it)
    // block it covers positions u in [T - (it+1)*CPB, T - it*CPB) clipped at 0, stored ascending
    // inside the ring (slot = u - ulo)
    for (int64_t it = 0; it < nb + 1; ++it) {
      ST_BEGIN;
      if (it < nb) {
        const int64_t uhi = T - it * CPB;
        const int64_t ulo = max((int64_t)0, uhi - CPB);
        bool dummy = true;
        emis_block<NT, true, false, TRATIO>(em, ltab, p0 + ulo, (int)(uhi - ulo), lane, N, dummy,
                                            nullptr, ltdv, tratios, ringB + (it & 1) * CPB * RS, RS,
                                            msB + (it & 1) * CPB);
      }
      ST_ADD(st_a);
      __syncthreads();
      ST_ADD(st_c);
    }
    ST_FLUSH(w);
  }
}

// posterior rows from alpha / beta rows: post = normalise(alpha * beta) [+ float32 eps quirk of
// score_samples, basehmm.py:271-272].  One wave per row, lane = state; in place into alpha.
template <bool EPS>
__global__ __launch_bounds__(256) void k_combine(int64_t rows, int N, double *alpha,
                                                 const double *beta) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const double eps = 1.1920928955078125e-07;
  const double epsden = 1.0 + (double)N * eps;
  for (int64_t r0 = wid * 4; r0 < rows; r0 += nw * 4) {
    double g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t r = r0 + k;
      g[k] = 0.0;
      if (r < rows && lane < N) g[k] = alpha[r * N + lane] * beta[r * N + lane];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t r = r0 + k;
      const double tot = wave_sum_f64(g[k]);
      double pr = g[k] / tot;
      if (EPS) pr = (pr + eps) / epsden;
      if (r < rows && lane < N) alpha[r * N + lane] = pr;
    }
  }
}
// ------------------------------------------------------------------------------------------
// Baum-Welch E-step accumulation (basehmm.py:516-522 + hmm.py:545-574 + _hmm.pyx:62-117 +
// _emission.pyx:165-190 in the scaled linear domain), parallel over position chunks.
// With a_t, beta_t the scaled rows of k_fb_coop, w_t = bh'_t * beta_t, e_t the power-of-two exponent
// applied at forward step t and G_t = sum_j a_t[j] beta_t[j]:
//   gamma_t = a_t * beta_t / G_t                                  (posterior, no eps: fit)
//   xi_t(i,j) = a_t[i] A[i][j] w_{t+1}[j] / (2^{e_{t+1}} G_{t+1})   (sums to 1 over i,j)
//   trans    += (1/N) * ( A o sum_t a_t (x) w_{t+1}/Z_t  +  diag( sum_{r_{t+1}>1} (r_{t+1}-1) gamma_{t+1} ) )
//               (the 1/N is the reference's log(1/N) backward terminal, quirk Q3; the diagonal
//                term is its extra `y` term for segment ratios, _hmm.pyx:94-99)
//   obs[k][j][sym] += gamma_t[j] * r_t ;  start += gamma_0.
// One wave per chunk of CH positions, lane = state j, C[i][lane] in registers, a_{t-1}[i] as
// scalar operands.  Small tracks' histograms are privatised in LDS, the others use fp64 global
// atomics.  C, D, start, the histograms are summed into global accumulators with atomics.
// ------------------------------------------------------------------------------------------
// sum over the 64 lanes, every lane gets it, without LDS: xor-butterfly inside the 16-lane rows by DPP on the two
// halves of the double (quad_perm, row_half_mirror, row_mirror), then the four rows by permlane swaps (item_sum4)
__device__ __forceinline__ double wave_sum_f64_all(double v) {
#define TEHMM_DPP_ADD(CTRL)                                                                                  \
  do {                                                                                                       \
    const int lo_ = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);                       \
    const int hi_ = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);                       \
    v += __hiloint2double(hi_, lo_);                                                                         \
  } while (0)
  TEHMM_DPP_ADD(0xB1);     // quad_perm [1,0,3,2]
  TEHMM_DPP_ADD(0x4E);     // quad_perm [2,3,0,1]
  TEHMM_DPP_ADD(0x141);    // row_half_mirror
  TEHMM_DPP_ADD(0x140);    // row_mirror
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
// C[F] += V[F] * wz for F = 0..NT-1, the broadcast of V[F] folded into the FMA (r = rep_rows(V))
template <int F, int NT>
struct BcastOuter {
  static __device__ __forceinline__ void run(const double (&r)[(NT + 15) / 16], double wz, double (&C)[NT]) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(C[F])
                 : "v"(r[F >> 4]), "v"(wz), "n"(F & 15));
    BcastOuter<F + 1, NT>::run(r, wz, C);
  }
};
template <int NT>
struct BcastOuter<NT, NT> {
  static __device__ __forceinline__ void run(const double (&)[(NT + 15) / 16], double, double (&)[NT]) {}
};

// One wave per chunk of CH positions, lane = state j, C[i][lane] in registers; a_{t-1}[i] reaches the FMA as a DPP
// row broadcast of the previous position's register (the first version read it back from memory: 36 dependent
// uniform loads per position, 8.6 us per position and wave).  The rows of position t + 1 are requested before
// position t is worked on.  Small tracks' histograms are privatised in LDS, the others use fp64 global atomics.
// C, D, start, the histograms are summed into global accumulators with atomics.
template <int NT, bool RATIO>
__global__ __launch_bounds__(256) void k_estep_accum(IntervalTab iv, EmisTab em, int N, int chunk_len,
                                                     const int *chunk_iv, const int64_t *chunk_t0,
                                                     int n_chunks, const double *__restrict__ alpha,
                                                     const double *__restrict__ beta, const double *__restrict__ wrows,
                                                     const int *__restrict__ escale, double *gC, double *gD,
                                                     double *gstart, double *gstat) {
  extern __shared__ double sm[];
  double *lstat = sm;                       // [lds_rows][NT] privatised histogram of the small tracks
  int *tinfo = (int *)(sm + (size_t)em.lds_rows * NT);     // [3][K]: ldsbase, rowcnt, rowbase (LDS: no per-use scalar loads)
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < em.lds_rows * NT; i += 256) lstat[i] = 0.0;
  for (int k = threadIdx.x; k < em.K; k += 256) {
    tinfo[k] = em.ldsbase[k];
    tinfo[em.K + k] = em.rowcnt[k];
    tinfo[2 * em.K + k] = em.rowbase[k];
  }
  __syncthreads();
  const bool live = lane < N;
  const int jl = min(lane, N - 1);
  const int c = blockIdx.x * 4 + w;
  if (c < n_chunks) {
    const int id = chunk_iv[c];
    const int64_t T = iv.len[id];
    const int64_t p0 = iv.pos0[id];
    const int64_t r0 = iv.out0[id];
    const int64_t tlo = chunk_t0[c];
    const int64_t thi = min(T, tlo + chunk_len);
    const int K = em.K, KPW = em.KPW;
    // track k's table info in lane k (read back with v_readlane: no memory access in the track loop); tracks
    // beyond 64 use the LDS copy
    const int ti_lb = lane < K ? tinfo[lane] : -1, ti_cnt = lane < K ? tinfo[K + lane] : 0,
              ti_rb = lane < K ? tinfo[2 * K + lane] : 0;
    double C[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) C[i] = 0.0;
    double D = 0.0;
    double aprev = (tlo > 0 && live) ? alpha[(r0 + tlo - 1) * N + lane] : 0.0;
    // the rows of the position being worked on (requested one iteration ahead)
    double a_n = 0.0, b_n = 0.0, w_n = 0.0, r_n = 1.0;
    int e_n = 0;
    uint32_t o_n[8];                          // observation words of the first 32 tracks (more: re-read below)
    auto request = [&](int64_t t) {
      const int64_t row = r0 + t;
      a_n = alpha[row * N + jl];
      b_n = beta[row * N + jl];
      w_n = wrows[row * N + jl];
      e_n = escale[row];
      if (RATIO) r_n = em.ratios[p0 + t];
      const uint32_t *orow = em.obs32 + (p0 + t) * KPW;
#pragma unroll
      for (int d = 0; d < 8; ++d) o_n[d] = d < KPW ? orow[d] : 0u;
    };
    if (tlo < thi) request(tlo);
    for (int64_t t = tlo; t < thi; ++t) {
      const double a = live ? a_n : 0.0, b = live ? b_n : 0.0, wv = live ? w_n : 0.0, r = r_n;
      const int es = e_n;
      uint32_t ow[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) ow[d] = o_n[d];
      if (t + 1 < thi) request(t + 1);
      const double g = a * b;
      const double G = wave_sum_f64_all(g);
      const double invG = 1.0 / G;
      const double gam = g * invG;
      if (t == 0 && live) atomicAdd(&gstart[lane], gam);
      // ---- transition statistics for the pair (t-1, t)
      if (t > 0) {
        const double wz = wv * ldexp(invG, -es);
        double rr[(NT + 15) / 16];
        rep_rows<NT>(aprev, rr);
        asm volatile("s_nop 1" ::: "memory");                 // VALU write -> DPP read wait states
        BcastOuter<0, NT>::run(rr, wz, C);
        if (RATIO && r > 1.) D += (r - 1.) * gam;
      }
      aprev = a;
      // ---- emission statistics
      const double gr = RATIO ? gam * r : gam;
      for (int k = 0; k < K; ++k) {
        uint32_t word = ow[0];
#pragma unroll
        for (int d = 1; d < 8; ++d) word = (k >> 2) == d ? ow[d] : word;
        if ((k >> 2) >= 8) word = em.obs32[(p0 + t) * KPW + (k >> 2)];
        const int sym = (int)((word >> ((k & 3) * 8)) & 0xffu);
        const int lb = k < 64 ? __builtin_amdgcn_readlane(ti_lb, k) : tinfo[k];
        const int cnt = k < 64 ? __builtin_amdgcn_readlane(ti_cnt, k) : tinfo[K + k];
        // (a symbol beyond the track's last one lands in the reference's padding cells, which
        //  emission.maximize never reads: not booked here)
        if (live && sym < cnt) {
          if (lb >= 0) {
            atomicAdd(&lstat[(lb + sym) * NT + lane], gr);
          } else {
            const int rb = k < 64 ? __builtin_amdgcn_readlane(ti_rb, k) : tinfo[2 * K + k];
            atomicAdd(&gstat[(int64_t)(rb + sym) * NT + lane], gr);
          }
        }
      }
    }
    if (live) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
        if (i < N) atomicAdd(&gC[i * NT + lane], C[i]);
      if (RATIO) atomicAdd(&gD[lane], D);
    }
  }
  __syncthreads();
  // flush the privatised histogram: LDS row (ldsbase[k] + s) -> global row (rowbase[k] + s)
  for (int k = 0; k < em.K; ++k) {
    const int lb = em.ldsbase[k];
    if (lb < 0) continue;
    for (int i = threadIdx.x; i < em.rowcnt[k] * NT; i += 256) {
      const double v = lstat[lb * NT + i];
      if (v != 0.0) atomicAdd(&gstat[(int64_t)em.rowbase[k] * NT + i], v);
    }
  }
}

// intervals whose forward pass met an impossible row after the first emittable one: the
// reference's lattices are NaN from there on; make the outputs say so.
__global__ void k_poison_dead(IntervalTab iv, const int *dead_flag, int N, double *post,
                              double *fwd_logprob) {
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  for (int id = blockIdx.y; id < iv.n; id += gridDim.y) {
    if (!dead_flag[id]) continue;
    const int64_t T = iv.len[id];
    double *o = post + iv.out0[id] * N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < T * N;
         i += (int64_t)gridDim.x * blockDim.x)
      o[i] = nan;
    if (blockIdx.x == 0 && threadIdx.x == 0) fwd_logprob[id] = nan;
  }
}

}  // namespace tehmm
