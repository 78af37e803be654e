// tehmm_lane.hip.h -- the throughput half of the chunk-parallel DPs: lane = sub-chunk.
//
// The speculative passes of tehmm_spec.hip.h keep "lane = destination state": 35 of 64 lanes work
// and every step pays for broadcasting the state vector (about 140 VALU instructions per position
// and pass).  Once a batch has tens of thousands of independent chunks there is a better map:
//   * every chunk of CS positions is cut into sub-chunks of L positions ("items", L | CS, 64 | L);
//   * a wavefront owns 64 consecutive items, ONE ITEM PER LANE: the lane keeps the whole state vector
//     of its item in registers, the transition table arrives as SCALAR operands (s_load, the same
//     for all lanes), so one step is NT*NT full-width v_fma_f64 (forward / backward) or
//     v_add_f64 + v_max_f64 pairs (Viterbi) and nothing else: ~20-40 instructions per position;
//   * an item starts Wu positions early (warm-up) from a uniform / zero vector; whether it has
//     forgotten that start by its first official position is CHECKED afterwards against the end
//     vector of the previous item ("link", k_*_stitch) -- a chunk whose links hold looks to the
//     sequential fix-up chain exactly like a chunk of the lane = state speculative pass;
//   * emission rows are computed once (k_emis_lane) into an item-interleaved layout
//         X[group][t_rel][state][lane]        (group = item / 64, lane = item % 64)
//     so that every load and store of the lane kernels is one coalesced 512-byte access; alpha and
//     beta rows stay in this layout (the fix-up chains address it too) and k_combine_lane turns them
//     into the [T][N] posterior with an LDS transpose.
#pragma once
#include "tehmm_spec.hip.h"

namespace tehmm {

// constant address space: uniform loads from it become scalar loads (s_load), their results SGPR operands
typedef __attribute__((address_space(4))) const double const_f64;
// Tables read as scalar operands are stored output-group-major: G[o / 4][f][o % 4] (NT * 4 contiguous
// doubles per group of four outputs).  A step then finishes four outputs at a time -- four independent
// accumulator chains fed by one contiguous scalar stream -- so only the state vector, its successor
// and a handful of temporaries are ever live (v[NT] + out[NT] doubles = 4 * NT VGPRs).
#define TEHMM_TG(f, o, NT) ((((o) >> 2) * (NT) + (f)) * 4 + ((o) & 3))



// ------------------------------------------------------------------------------------------
// Emission rows of every position, item-interleaved.  B = log rows (the reference's operation
// order, _emission.pyx:65-72), B32 = the same rounded to float, BH = exp(B - rowmax), MS = rowmax.  A row no state can emit is
// written as NaN: it poisons the lane passes, their links fail and the exact chain -- which owns the
// reference's semantics for such rows (leading-rows quirk, dead lattices) -- walks through it.
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k_emis_lane(IntervalTab iv, EmisTab em, LaneGeom lg, int N,
                                                   double *B, double *BH, double *MS, float *B32) {
  extern __shared__ double emis_ltab[];           // the small tracks' table rows (em.lds_rows of them)
  stage_emis_table(em, emis_ltab, NT);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= lg.n_groups) return;
  const int64_t item = (int64_t)g * 64 + lane;
  const bool valid = item < lg.n_items;
  const int id = valid ? lg.item_iv[item] : 0;
  const int64_t t0 = valid ? lg.item_t0[item] : 0;
  const int64_t T = iv.len[id], p0 = iv.pos0[id];
  const int len = valid ? (int)min((int64_t)lg.L, T - t0) : 0;
  const int maxlen = wave_max_i32(len);
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  for (int s = 0; s < maxlen; ++s) {
    const bool act = s < len;
    const int64_t gpos = p0 + t0 + (act ? s : 0);
    double x[NT];
    emis_rows<NT>(em, emis_ltab, gpos, x);
    double m = x[0];
#pragma unroll
    for (int j = 1; j < NT; ++j) m = fmax(m, j < N ? x[j] : -INFINITY);
    const bool good = m > -1e20;
    if (act) {
      const int64_t o = ((((int64_t)g * lg.L + s) * NT) << 6) + lane;
      if (B) {
#pragma unroll
        for (int j = 0; j < NT; ++j) B[o + ((int64_t)j << 6)] = good ? x[j] : qnan;
      }
      if (B32) {                        // log rows rounded to float: enough for the binade-placement pass
#pragma unroll
        for (int j = 0; j < NT; ++j) B32[o + ((int64_t)j << 6)] = good ? (float)x[j] : (float)qnan;
      }
      if (BH) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
          BH[o + ((int64_t)j << 6)] = good ? (j < N ? exp_nonpos(x[j] - m) : 0.0) : qnan;
        MS[(((int64_t)g * lg.L + s) << 6) + lane] = m;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Forward (DIR 0) / backward (DIR 1) lane pass.  tab = A (forward) or A^T (backward), [NT][NT],
// so that a step is  out[o] = sum_f in[f] * tab[f][o]  with tab[f][o] a scalar operand.
// An item runs when its chunk is full and is not the interval's first (forward) / last (backward)
// chunk -- those belong to the exact chain anyway.
//   rows  : alpha' / beta' rows of the official range, item-interleaved
//   pre   : the chain vector after the warm-up (forward: a_{t0-1}; backward: w_{t0+L}), [g][NT][64]
//   end   : the chain vector after the last official step (forward: a_{t0+L-1}; backward: w_{t0})
//   slog32: forward only, [item][L/32] cumulative log-scale since `pre` at positions t0+32k+31
// ------------------------------------------------------------------------------------------
template <int NT, int DIR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_fb_lane(IntervalTab iv, LaneGeom lg, int N, int CS, int Wu, const double *__restrict__ tab,
               const double *__restrict__ BH, const double *__restrict__ MS, double *rows, double *pre,
               double *end, double *slog32) {
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= lg.n_groups) return;
  const int L = lg.L;
  const int64_t item = (int64_t)g * 64 + lane;
  const bool valid = item < lg.n_items;
  const int id = valid ? lg.item_iv[item] : 0;
  const int64_t t0 = valid ? lg.item_t0[item] : 0;
  const int64_t T = iv.len[id];
  const int64_t ct0 = (t0 / CS) * CS;
  const bool run = valid && ct0 + CS <= T && (DIR == 0 ? ct0 > 0 : ct0 + CS < T);
  if (!__any(run)) return;
  // the neighbour item whose emission rows cover the warm-up positions (same interval by construction)
  const int64_t nb = run ? (DIR == 0 ? item - 1 : item + 1) : item;
  const int wu = !run ? 0 : (DIR == 0 ? Wu : (int)min((int64_t)Wu, T - (t0 + L)));
  double v[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) v[j] = DIR == 0 ? (j < N ? 1.0 / (double)N : 0.0) : 0.0;
  double slog = 0.0;

  // out[o] = sum_f v[f] * tab[f][o], as a pinned software pipeline over the scalar stream.  SMEM answers
  // out of order, so a wait is always for ALL outstanding loads: block b + 1 (4 table rows x 4 outputs
  // = 16 doubles = two s_load_dwordx16) is therefore requested only AFTER the first FMA of block b has
  // waited for block b's operands (the empty asm makes the request depend on that FMA), and it
  // arrives while the other 15 FMAs run.  The scheduling barrier keeps the blocks apart; the pointer
  // is laundered every step so that nothing is hoisted out of the position loop.
  auto matvec = [&](double (&acc)[NT]) {
    const_f64 *tp = (const_f64 *)(size_t)tab;
    asm volatile("" : "+s"(tp));
    constexpr int BLK = 16, NB = NT * NT / BLK;
    double t[2][BLK];
#pragma unroll
    for (int i = 0; i < BLK; ++i) t[0][i] = tp[i];
    double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma clang loop unroll(full)
    for (int b = 0; b < NB; ++b) {
      const int og = (b * 4) / NT, f0 = (b * 4) % NT;
      const double *tc = t[b & 1];
      double *tn = t[(b + 1) & 1];
      a[0] = fma(v[f0], tc[0], a[0]);
      const_f64 *tq = tp;
      asm volatile("" : "+s"(tq) : "v"(a[0]));
      if (b + 1 < NB) {
#pragma unroll
        for (int i = 0; i < BLK; ++i) tn[i] = tq[(b + 1) * BLK + i];
      }
#pragma unroll
      for (int i = 1; i < BLK; ++i) a[i & 3] = fma(v[f0 + (i >> 2)], tc[i], a[i & 3]);
      if (f0 + 4 == NT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[4 * og + q] = a[q];
          a[q] = 0.0;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto expo = [&](const double (&a)[NT]) {
    double s0 = a[0], s1 = a[1], s2 = a[2], s3 = a[3];
#pragma unroll
    for (int j = 4; j < NT; j += 4) {
      s0 += a[j];
      s1 += a[j + 1];
      s2 += a[j + 2];
      s3 += a[j + 3];
    }
    const double s = (s0 + s1) + (s2 + s3);
    return ((__double2hiint(s) >> 20) & 0x7ff) - 1022;
  };

  if (DIR == 0) {
    // ---- forward: v = a_{t-1} -> a_t = normalise((v A) * bh_t)
    auto step = [&](const double *bp, double ms) {
      double bh[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bh[j] = bp[(int64_t)j << 6];
      double acc[NT];
      matvec(acc);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] *= bh[j];
      const int e = expo(acc);
#pragma unroll
      for (int j = 0; j < NT; ++j) v[j] = ldexp(acc[j], -e);
      slog += (double)e * 0.6931471805599453 + ms;
    };
    for (int s = -Wu; s < 0; ++s) {
      const int64_t o = lane_row(lg, NT, nb, L + s);
      step(BH + o, 0.0);
    }
    if (run) {
      const int64_t po = (((int64_t)g * NT) << 6) + lane;
#pragma unroll
      for (int j = 0; j < NT; ++j) pre[po + ((int64_t)j << 6)] = v[j];
    }
    slog = 0.0;
    for (int s = 0; s < L; ++s) {
      const int64_t o = lane_row(lg, NT, item, s);
      step(BH + o, MS[(((int64_t)g * L + s) << 6) + lane]);
      if (run) {
#pragma unroll
        for (int j = 0; j < NT; ++j) rows[o + ((int64_t)j << 6)] = v[j];
        if ((s & 31) == 31) slog32[item * (L / 32) + (s >> 5)] = slog;
      }
    }
    if (run) {
      const int64_t po = (((int64_t)g * NT) << 6) + lane;
#pragma unroll
      for (int j = 0; j < NT; ++j) end[po + ((int64_t)j << 6)] = v[j];
    }
  } else {
    // ---- backward: v = w_{t+1} = bh'_{t+1} * beta_{t+1} -> beta_t = normalise(A v), w_t = bh'_t * beta_t
    for (int s = L + Wu - 1; s >= L; --s) {
      const int top = L + wu - 1;                       // first (highest) warm-up position of this lane
      const int64_t o = lane_row(lg, NT, nb, min(s, top) - L < 0 ? 0 : min(s, top) - L);
      double bh[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bh[j] = BH[o + ((int64_t)j << 6)];
      double acc[NT];
      matvec(acc);
      const int e = expo(acc);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        double bt = ldexp(acc[j], -e);
        if (s == top) bt = j < N ? 1.0 : 0.0;           // uniform start
        v[j] = s > top ? 0.0 : bh[j] * bt;
      }
    }
    if (run) {
      const int64_t po = (((int64_t)g * NT) << 6) + lane;
#pragma unroll
      for (int j = 0; j < NT; ++j) pre[po + ((int64_t)j << 6)] = v[j];
    }
    for (int s = L - 1; s >= 0; --s) {
      const int64_t o = lane_row(lg, NT, item, s);
      double bh[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bh[j] = BH[o + ((int64_t)j << 6)];
      double acc[NT];
      matvec(acc);
      const int e = expo(acc);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = ldexp(acc[j], -e);          // beta_t
      if (run) {
#pragma unroll
        for (int j = 0; j < NT; ++j) rows[o + ((int64_t)j << 6)] = acc[j];
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) v[j] = bh[j] * acc[j];
    }
    if (run) {
      const int64_t po = (((int64_t)g * NT) << 6) + lane;
#pragma unroll
      for (int j = 0; j < NT; ++j) end[po + ((int64_t)j << 6)] = v[j];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Forward / backward lane pass on the fp64 MATRIX cores (same interface and results as k_fb_lane).
// The VALU form is instruction-fetch bound (one 4..8-byte instruction per 64 FMAs); one
// v_mfma_f64_16x16x4_f64 carries 1 024 FMAs and needs no scalar operand stream.  Per step and tile of
// 16 items the product is computed TRANSPOSED,
//     D'[state j][item i] = sum_k  T[j][k] * V[k][i],     T[j][k] = A[k][j] (forward) or A[j][k] (backward),
// because then the accumulator layout of the instruction (lane l, register r: row (l >> 4) + 4 r of a
// 16-row tile, column l & 15) IS the B-operand layout of the next step (lane l: k = 4 s + (l >> 4),
// column l & 15) with s = r + 4 * (row tile): the recurrence never leaves the registers and needs no
// cross-lane movement.  Lane l owns item (l & 15) of its tile and the states (l >> 4) + 4 s; the
// table lives in NT/4 * ceil(NT/16) constant fragments (27 doubles per lane for NT = 36).
// ------------------------------------------------------------------------------------------
typedef double lane_d4 __attribute__((ext_vector_type(4)));

template <int NT, int DIR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_fb_mfma(IntervalTab iv, LaneGeom lg, int N, int CS, int Wu,
                                                 const double *__restrict__ tab /* A, [NT][NT] row-major */,
                                                 const double *__restrict__ BH, const double *__restrict__ MS,
                                                 double *rows, double *pre, double *end, double *slog32) {
  constexpr int KS = NT / 4;             // k-steps of 4 states
  constexpr int RT = (NT + 15) / 16;     // row tiles of 16 states
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= lg.n_groups * 4) return;
  const int L = lg.L;
  const int kq = lane >> 4;
  const int64_t item = (int64_t)tile * 16 + (lane & 15);
  const bool valid = item < lg.n_items;
  const int id = valid ? lg.item_iv[item] : 0;
  const int64_t t0 = valid ? lg.item_t0[item] : 0;
  const int64_t T = iv.len[id];
  const int64_t ct0 = (t0 / CS) * CS;
  const bool run = valid && ct0 + CS <= T && (DIR == 0 ? ct0 > 0 : ct0 + CS < T);
  if (!__any(run)) return;
  const int64_t nb = run ? (DIR == 0 ? item - 1 : item + 1) : item;
  const int wu = !run ? 0 : (DIR == 0 ? Wu : (int)min((int64_t)Wu, T - (t0 + L)));
  // table fragments: A-operand lane map = T[16 rt + (l & 15)][4 s + (l >> 4)]
  double tf[RT][KS];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int j = 16 * rt + (lane & 15), k = 4 * s + kq;
      tf[rt][s] = (j < NT) ? (DIR == 0 ? tab[k * NT + j] : tab[j * NT + k]) : 0.0;
    }
  }
  // this lane's states: kq + 4 s
  double v[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) v[s] = (DIR == 0 && kq + 4 * s < N) ? 1.0 / (double)N : 0.0;
  double slog = 0.0;
  const int64_t soff = (int64_t)kq << 6;          // state kq of a row; further states are 4 * 64 doubles apart

  auto product = [&](double (&out)[KS]) {
    lane_d4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (lane_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(tf[rt][s], v[s], acc[rt], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) out[s] = acc[s >> 2][s & 3];
  };
  auto item_sum = [&](const double (&a)[KS]) {        // over the item's states: in-lane, then its 4 lanes
    double t = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) t += a[s];
    t += __shfl_xor(t, 16);
    t += __shfl_xor(t, 32);
    return t;
  };
  auto vec_out = [&](double *dst) {
    const int64_t po = ((((item >> 6) * NT) + kq) << 6) + (item & 63);
#pragma unroll
    for (int s = 0; s < KS; ++s) dst[po + ((int64_t)(4 * s) << 6)] = v[s];
  };

  if (DIR == 0) {
    for (int s = -Wu; s < L; ++s) {
      if (s == 0) {
        if (run) vec_out(pre);
        slog = 0.0;
      }
      const int64_t o = (s < 0 ? lane_row(lg, NT, nb, L + s) : lane_row(lg, NT, item, s)) + soff;
      double bh[KS];
#pragma unroll
      for (int q = 0; q < KS; ++q) bh[q] = BH[o + ((int64_t)(4 * q) << 6)];
      const double ms = s < 0 ? 0.0 : MS[(((item >> 6) * L + s) << 6) + (item & 63)];
      double acc[KS];
      product(acc);
#pragma unroll
      for (int q = 0; q < KS; ++q) acc[q] *= bh[q];
      const double tot = item_sum(acc);
      const int e = ((__double2hiint(tot) >> 20) & 0x7ff) - 1022;
#pragma unroll
      for (int q = 0; q < KS; ++q) v[q] = ldexp(acc[q], -e);
      slog += (double)e * 0.6931471805599453 + ms;
      if (s >= 0 && run) {
#pragma unroll
        for (int q = 0; q < KS; ++q) rows[o + ((int64_t)(4 * q) << 6)] = v[q];
        if ((s & 31) == 31 && kq == 0) slog32[item * (L / 32) + (s >> 5)] = slog;
      }
    }
    if (run) vec_out(end);
  } else {
    for (int s = L + Wu - 1; s >= 0; --s) {
      const int top = L + wu - 1;                        // first (highest) warm-up position of this item
      if (s == L - 1 && run) vec_out(pre);               // v = w_{t0+L} after the warm-up
      const int sc = s >= L ? max(min(s, top) - L, 0) : s;
      const int64_t o = (s >= L ? lane_row(lg, NT, nb, sc) : lane_row(lg, NT, item, sc)) + soff;
      double bh[KS];
#pragma unroll
      for (int q = 0; q < KS; ++q) bh[q] = BH[o + ((int64_t)(4 * q) << 6)];
      double acc[KS];
      product(acc);
      const double tot = item_sum(acc);
      const int e = ((__double2hiint(tot) >> 20) & 0x7ff) - 1022;
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        double bt = ldexp(acc[q], -e);                   // beta_t
        if (s >= L && s == top) bt = kq + 4 * q < N ? 1.0 : 0.0;     // uniform start
        if (s < L && run) rows[o + ((int64_t)(4 * q) << 6)] = bt;
        v[q] = (s >= L && s > top) ? 0.0 : bh[q] * bt;
      }
    }
    if (run) vec_out(end);
  }
}

// ------------------------------------------------------------------------------------------
// Links of the forward / backward lane pass.  Whether an item had forgotten its (uniform) start by its
// first official position is CHECKED: its vector after the warm-up against the end vector of its
// neighbour at the same position (Hilbert distance <= TEHMM_FB_TOL).
//   k_fb_itemlinks: one THREAD per item, loop over the states -- with lane = item every load of the
//     item-interleaved vectors is coalesced and the max / min of the ratios need no cross-lane step;
//     dl_f / lr_f: distance and log(rho) of item i against item i - 1 (forward), dl_b: item i against i + 1.
//   k_fb_stitch: one thread per chunk: a chunk is usable by the fix-up chain when all links inside it
//     hold; the items' log-scale records are laid end to end (scale[c][CS/32], frame of the chunk's first
//     item); link_f / glog_f / link_b connect consecutive chunks for k_fb_runs.
// mode: 1 = forward tables only, 2 = backward only, 3 = both (the forward half runs, with the forward
// fix-up chain behind it, while the backward lane pass is still at work).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void link_accum(double a, double b, bool &bad, double &rmax, double &rmin) {
  const bool both0 = a == 0.0 && b == 0.0;
  const bool okl = both0 || (a > 0.0 && b > 0.0 && a < INFINITY && b < INFINITY);
  bad = bad | !okl;
  if (okl && !both0) {
    const double r = a / b;
    rmax = fmax(rmax, r);
    rmin = fmin(rmin, r);
  }
}

template <int NT>
__global__ __launch_bounds__(256) void k_fb_itemlinks(LaneGeom lg, int N, const double *pre_f, const double *end_f,
                                                      const double *pre_b, const double *end_b, double *dl_f,
                                                      double *lr_f, double *dl_b, int mode) {
  const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= lg.n_items) return;
  auto vecp = [&](const double *p, int64_t it) { return p + (((it >> 6) * NT) << 6) + (it & 63); };
  if ((mode & 1) && item >= 1) {
    const double *a = vecp(pre_f, item), *b = vecp(end_f, item - 1);
    bool bad = false;
    double rmax = -1.0, rmin = INFINITY;
    for (int j = 0; j < N; ++j) link_accum(a[(int64_t)j << 6], b[(int64_t)j << 6], bad, rmax, rmin);
    const bool ok = !bad && rmax > 0.0 && rmin < INFINITY;
    dl_f[item] = ok ? rmax / rmin - 1.0 : INFINITY;
    lr_f[item] = ok ? log(rmax) : 0.0;
  }
  if ((mode & 2) && item + 1 < lg.n_items) {
    const double *a = vecp(pre_b, item), *b = vecp(end_b, item + 1);
    bool bad = false;
    double rmax = -1.0, rmin = INFINITY;
    for (int j = 0; j < N; ++j) link_accum(a[(int64_t)j << 6], b[(int64_t)j << 6], bad, rmax, rmin);
    const bool ok = !bad && rmax > 0.0 && rmin < INFINITY;
    dl_b[item] = ok ? rmax / rmin - 1.0 : INFINITY;
  }
}

template <int NT>
__global__ __launch_bounds__(256) void k_fb_stitch(IntervalTab iv, LaneGeom lg, FbChunks fc, int N,
                                                   const double *slog32, const double *end_b, const double *dl_f,
                                                   const double *lr_f, const double *dl_b, int *ok_f, int *ok_b,
                                                   int mode) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= fc.n) return;
  const int id = fc.iv[c];
  const int64_t T = iv.len[id];
  const int64_t ct0 = fc.t0[c];
  const int L = lg.L, SUB = fc.CS / L, R = L / 32;
  const bool full = ct0 + fc.CS <= T;
  const int64_t item0 = lg.ifirst[id] + ct0 / L;
  if (mode & 1) {
    bool okf = full && c != fc.first[id];
    int lf = 0;
    double gl = 0.0;
    if (okf) {
      double off = 0.0;
      for (int k = 0; k < SUB; ++k) {
        const int64_t item = item0 + k;
        if (k > 0) {
          okf = okf && dl_f[item] <= TEHMM_FB_TOL;
          off += slog32[(item - 1) * R + R - 1] - lr_f[item];
        }
        for (int m = 0; m < R; ++m) fc.scale[(int64_t)c * (fc.CS / 32) + k * R + m] = slog32[item * R + m] + off;
      }
      // link to the previous chunk (if the lanes ran it); glog = log-scale gained over this chunk, expressed
      // in the previous chunk's frame
      if (okf && c - 1 != fc.first[id]) {
        lf = dl_f[item0] <= TEHMM_FB_TOL ? 1 : 0;
        gl = off + slog32[(item0 + SUB - 1) * R + R - 1] - lr_f[item0];
      }
    }
    fc.link_f[c] = lf;
    fc.glog_f[c] = gl;
    ok_f[c] = okf ? 1 : 0;
  }
  if (mode & 2) {
    bool okb = full && ct0 + fc.CS < T;
    int lb = 0;
    if (okb) {
      for (int k = SUB - 2; k >= 0; --k) okb = okb && dl_b[item0 + k] <= TEHMM_FB_TOL;
      const double *w = end_b + (((item0 >> 6) * NT) << 6) + (item0 & 63);
      for (int j = 0; j < NT; ++j) fc.wstart[(int64_t)c * NT + j] = j < N ? w[(int64_t)j << 6] : 0.0;
      // link to the next chunk (if the lanes ran it for the backward pass)
      if (okb && ct0 + 2 * (int64_t)fc.CS < T) lb = dl_b[item0 + SUB - 1] <= TEHMM_FB_TOL ? 1 : 0;
    }
    fc.link_b[c] = lb;
    ok_b[c] = okb ? 1 : 0;
  }
}

// Runs of linked chunks, one thread per interval: a verified jump of the fix-up chain carries on over
// every following chunk whose links hold (forward: towards the interval end; backward: towards its
// start), so a long interval costs its first chunk plus one verification, not one per chunk.
__global__ __launch_bounds__(64) void k_fb_runs(IntervalTab iv, FbChunks fc, const int *ok_f, const int *ok_b,
                                                int extend, int mode) {
  // one wave per interval, 64 chunks per round: run ends / starts from ballots of the "run breaks here"
  // flags, the log-scale prefix sums from a wave scan
  const int id = blockIdx.x;
  const int lane = threadIdx.x;
  if (id >= iv.n) return;
  const int64_t c0 = fc.first[id], c1 = fc.first[id + 1];
  // forward prefix sums of glog_f over the chunks that continue their predecessor
  double carry = 0.0;
  for (int64_t b0 = c0; (mode & 1) && b0 < c1; b0 += 64) {
    const int64_t c = b0 + lane;
    double x = (c < c1 && fc.link_f[c]) ? fc.glog_f[c] : 0.0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    x += carry;
    if (c < c1) fc.pre_f[c] = x;
    carry = __shfl(x, 63);
  }
  // runend_f[c] = first chunk >= c after which the forward run breaks (descending over the rounds)
  int run_carry = (int)(c1 - 1);
  for (int64_t b0 = c0 + ((c1 - c0 - 1) / 64) * 64; (mode & 1) && b0 >= c0; b0 -= 64) {
    const int64_t c = b0 + lane;
    const bool brk = c >= c1 - 1 || !(extend && ok_f[c + 1] != 0 && fc.link_f[c + 1] != 0);
    const unsigned long long m = __ballot(brk && c < c1) & (~0ull << lane);
    const int r = m ? (int)(b0 + __ffsll((long long)m) - 1) : run_carry;
    if (c < c1) fc.runend_f[c] = r;
    run_carry = __shfl(r, 0);
  }
  // runstart_b[c] = last chunk <= c before which the backward run breaks (ascending)
  run_carry = (int)c0;
  for (int64_t b0 = c0; (mode & 2) && b0 < c1; b0 += 64) {
    const int64_t c = b0 + lane;
    const bool brk = c <= c0 || !(extend && c < c1 && ok_b[c - 1] != 0 && fc.link_b[c - 1] != 0);
    const unsigned long long m = __ballot(brk && c < c1) & (lane == 63 ? ~0ull : ((2ull << lane) - 1));
    const int r = m ? (int)(b0 + 63 - __clzll((long long)m)) : run_carry;
    if (c < c1) fc.runstart_b[c] = r;
    const int last = (int)min((int64_t)63, c1 - 1 - b0);
    run_carry = __shfl(r, last);
  }
}

// ------------------------------------------------------------------------------------------
// posterior rows [T][N] from the item-interleaved alpha' / beta' rows (+ the float32 eps quirk of
// score_samples, basehmm.py:271-272).  One workgroup per (group, 8 positions): coalesced tile loads,
// LDS transpose, per-row normalisation, 280-byte contiguous row stores.
// ------------------------------------------------------------------------------------------
template <int NT, bool EPS>
__global__ __launch_bounds__(256) void k_combine_lane(IntervalTab iv, LaneGeom lg, int N, const double *al,
                                                      const double *be, double *post) {
  // two positions per round: tiles [2][64 items][NT + 1], row scales [2][64]
  __shared__ double tile[2 * 64 * (NT + 1)];
  __shared__ double rs[2 * 64];
  __shared__ int64_t pb[64];
  __shared__ int rlen[64];
  const int tid = threadIdx.x;
  const int L = lg.L;
  const int per = L / 8;
  const int g = blockIdx.x / per;
  const int s0 = (blockIdx.x % per) * 8;
  if (tid < 64) {
    const int64_t item = (int64_t)g * 64 + tid;
    const bool valid = item < lg.n_items;
    const int id = valid ? lg.item_iv[item] : 0;
    const int64_t t0 = valid ? lg.item_t0[item] : 0;
    pb[tid] = iv.out0[id] + t0;
    rlen[tid] = valid ? (int)min((int64_t)L, iv.len[id] - t0) : 0;
  }
  const double eps = 1.1920928955078125e-07;
  const double inv_epsden = 1.0 / (1.0 + (double)N * eps);
  __syncthreads();
  for (int s = s0; s < s0 + 8; s += 2) {
    // coalesced tile loads of both positions (NT * 64 doubles each), transposed into LDS
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t o = (((int64_t)g * L + s + h) * NT) << 6;
      double *tl = tile + h * 64 * (NT + 1);
      for (int idx = tid; idx < NT * 64; idx += 256) {
        const int j = idx >> 6, ln = idx & 63;
        double p = 0.0;
        if (s + h < rlen[ln]) p = al[o + idx] * be[o + idx];
        tl[ln * (NT + 1) + j] = p;
      }
    }
    __syncthreads();
    if (tid < 128) {                         // 1 / row sum of each of the 2 x 64 rows
      const double *tr = tile + (tid >> 6) * 64 * (NT + 1) + (tid & 63) * (NT + 1);
      double t = 0.0;
      for (int j = 0; j < N; ++j) t += tr[j];
      rs[tid] = 1.0 / t;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const double *tl = tile + h * 64 * (NT + 1);
      for (int idx = tid; idx < 64 * N; idx += 256) {
        const int r = idx / N, col = idx - r * N;
        if (s + h < rlen[r]) {
          double pr = tl[r * (NT + 1) + col] * rs[h * 64 + r];
          if (EPS) pr = (pr + eps) * inv_epsden;
          post[(pb[r] + s + h) * N + col] = pr;
        }
      }
    }
    __syncthreads();
  }
}


// ==========================================================================================
// Exact quantised Viterbi pass (P2 of tehmm_spec.hip.h) as a lane = item pass: the binade's exact
// integer max-plus recurrence, one item per lane.  It needs the quantised transition table of the item's
// binade as scalar operands, so a wave is launched per (group, binade) pair and lanes of another binade
// sit idle (binades change ~8 times per interval); waves are sorted by binade so that co-resident waves
// share one table in the scalar cache.  Outputs: packed traceback dwords straight into
// tb[(pos0 + t) * NT + state], W rows at every 16th position straight into vc.rows (lane frame), pre /
// end vectors, ties, per-piece minima; k_vit_stitch then links the items of a chunk into segments for
// k_vit_fix.  (QUANT = false is the same recurrence in plain fp64; P0 uses k_vit_gain_lane instead.)
//
// What the compiler needs to be told here (each cost a factor when it was missing):
//   * the block loop must be FULLY unrolled (`#pragma clang loop unroll(full)`): a partial unroll
//     indexes W / t dynamically and every step goes through scratch;
//   * no per-element booleans in the unrolled body: each v_cmp pins a scalar mask, 36 of them spill
//     thousands of SGPRs -- ties are found by a running minimum and ONE compare, dead states by maxNum;
//   * the rarely used chunk / item tables come through pointers re-read in the rare paths;
//   * 340 registers: one wave per SIMD.  (Parking the successor vector in LDS for a second wave loses:
//     LDS traffic shares lgkmcnt with the scalar operand stream.)
// ==========================================================================================
#define TEHMM_LANE_MAXTI 8
struct VitItems {
  double *pre, *end;        // [group][NT][64]  lane-frame vectors of positions t0-1 and t0+L-1
  double *gain;             // P0: [item]
  int *bad;                 // P2: [item] the item cannot be used (dead vector, impossible row, > MAXTI ties)
  int *ntie;                // P2: [item] ties inside the official range
  int *ties;                // P2: [item][MAXTI] tie positions relative to t0, ascending
  double *tierows;          // P2: [item][MAXTI][NT] lane-frame row of the position before each tie
  double *piecemin;         // P2: [item][MAXTI + 1] lowest live value of each piece between ties
};

#ifndef TEHMM_P2_WAVES
#define TEHMM_P2_WAVES 1
#endif
// RATIO (round 2): segment ratios on the transitions (decode on a segmented TrackTable, _hmm.pyx:229-247).  Inside a
// binade every separately rounded addend of the reference's sums is just another grid-rounded term:
//   from >= 1:  ((V + lt) + b) + lt[to][to] * (r - 1)  [if r > 1]     =  V + R(lt) + R(b) + R(z1)
//   from == 0:  ((V + lt) + b) + lt[to][to] * r  [- lt[0][0] if to == 0] = V + R(lt) + R(b) + R(z0) [- R(lt00)]
// (the from-state-0 quirk Q4).  So the candidates from >= 1 share the offset R(z1) and candidate 0 carries the
// difference d0 = R(z0) - R(z1) - [to == 0] R(lt00), a multiple of u: the table stream gets one header block per
// output group with lt[to][to] of its four outputs (and R(lt00)), the products are formed and rounded per
// position, and a product that lands exactly between two grid points is a rounding tie like one of b.
template <int NT, bool QUANT, bool RATIO = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TEHMM_P2_WAVES, TEHMM_P2_WAVES)))
void k_vit_lane(IntervalTab iv, LaneGeom lg, const VitChunks *vcp, const VitItems *vip, int N, int Wu, const int *wk_g,
                const int *wk_e, int n_work, const double *__restrict__ tabs, int e0,
                const double *__restrict__ B, uint8_t *tb, const double *__restrict__ ratios = nullptr,
                const int *__restrict__ wk_items = nullptr, const int *__restrict__ n_work_dev = nullptr) {
  const int lane = threadIdx.x & 63;
  const int wk = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));   // wave-uniform
  if (wk >= (n_work_dev ? *n_work_dev : n_work)) return;      // (device-side placement: the grid is an upper bound)
  // work unit: a whole group of 64 consecutive items (wk_g >= 0: all its chunks of binade e run), or -- groups
  // that mix binades (interval heads, binade crossings) -- a list of up to 64 items of ONE binade collected
  // over all such groups (wk_g = -(1 + slot), items wk_items[64 slot ..], -1 = empty lane)
  const int gq = QUANT ? wk_g[wk] : wk;
  const int e = QUANT ? wk_e[wk] : 0;
  // the chunk / item tables are reached through pointers that are re-read where they are needed (rare
  // paths): as by-value arguments their 22 pointers stay in SGPRs across the unrolled step and spill
  auto VC = [&]() { const VitChunks *q = vcp; asm volatile("" : "+s"(q)); return *q; };
  auto VI = [&]() { const VitItems *q = vip; asm volatile("" : "+s"(q)); return *q; };
  const int L = lg.L, CS = vcp->CS;
  int64_t item = (int64_t)gq * 64 + lane;
  bool valid = item < lg.n_items;
  if (QUANT && gq < 0) {
    const int li = wk_items[(int64_t)(-1 - gq) * 64 + lane];
    valid = li >= 0;
    item = valid ? li : 0;
  }
  const int id = valid ? lg.item_iv[item] : 0;
  const int64_t t0 = valid ? lg.item_t0[item] : 0;
  const int64_t T = iv.len[id], p0 = iv.pos0[id];
  const int64_t ct0 = (t0 / CS) * CS;
  const int64_t c = vcp->first[id] + t0 / CS;
  bool run = valid && ct0 > 0 && ct0 + CS <= T;      // first chunk and ragged tail: exact chain only
  if (QUANT) run = run && vcp->e[c] == e;
  if (!__any(run)) return;
  const int64_t nb = run ? item - 1 : item;
  constexpr int TABSZ = RATIO ? NT * NT + (NT / 4) * 16 : NT * NT;      // RATIO: + one header block per output group
  const_f64 *tab0 = (const_f64 *)(size_t)tabs + (QUANT ? (int64_t)(e - e0) * TABSZ : 0);
  const double u = QUANT ? ldexp(1.0, e - 52) : 0.0;
  const double M = QUANT ? ldexp(1.5, e) : 0.0;          // fl(z + M) - M rounds z to the grid u
  const double half_u = 0.5 * u;
  // the arg-max extraction: xs + CM has exponent e for -2^e + 128 u < xs <= 63 u, i.e. its unit in the last place is
  // u and the six low mantissa bits ARE xs / u mod 64 = 63 - (first arg-max); CM is a multiple of 64 u
  const double CM = QUANT ? ldexp(1.0, e + 1) - ldexp(1.0, e - 45) : 0.0;          // 2^(e+1) - 128 u
  const double wlim = QUANT ? -(ldexp(1.0, e) - ldexp(1.5, e - 45)) : -INFINITY;   // -(2^e - 192 u)
  const double zlim = QUANT ? ldexp(1.0, e - 1) : INFINITY;
  double W[NT];                                          // QUANT: 64 x (value - base); plain: value
#pragma unroll
  for (int j = 0; j < NT; ++j) W[j] = j < N ? 0.0 : -INFINITY;
  double base = 0.0, pmin = INFINITY;
  int nt = 0;
  bool bad = false;
  uint32_t *tb32 = (uint32_t *)tb;

  auto restart = [&]() {
#pragma unroll
    for (int j = 0; j < NT; ++j) W[j] = j < N ? 0.0 : -INFINITY;
    base = 0.0;
    pmin = INFINITY;
  };
  // One position: s = position relative to t0 (negative: warm-up), bp = its emission row, bpn = the row
  // of the next position.  Outputs are finished four at a time; the four emission values of a group are
  // requested one group ahead (bnx: the first group of a step during the last group of the step
  // before), so the row is read once and its latency hides behind a group of max-plus work.  The old
  // vector stays intact until the end of the step, which is when a rounding tie found on the way is
  // recorded.
  double bnx[4];
#ifndef TEHMM_P2_PF
#define TEHMM_P2_PF 2      // emission values are requested this many output groups ahead
#endif
  // (a row of one group only -- NT = 4 -- has no "group after next" in the same or the following row)
  constexpr int P2_PF = (TEHMM_P2_PF == 2 && NT / 4 >= 2) ? 2 : 1;
  double bny[4];
  bool pending = false;
  auto prefetch = [&](const double *row, int og) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bnx[q] = row[(int64_t)(4 * og + q) << 6];
  };
  auto prefetch2 = [&](const double *row, int og) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bny[q] = row[(int64_t)(4 * og + q) << 6];
  };
  auto step = [&](const double *bp, const double *bpn, int s, double rt) {
    const bool official = s >= 0;
    double tmin = 1.0;
    // the max-plus recurrence as a pinned software pipeline over the scalar stream (see k_fb_lane)
    const_f64 *tp = tab0;
    asm volatile("" : "+s"(tp));
    constexpr int BLK = 16, GB = NT / 4 + (RATIO ? 1 : 0), NB = (NT / 4) * GB;   // blocks per output group, in all
    double t[2][BLK];
#pragma unroll
    for (int i = 0; i < BLK; ++i) t[0][i] = tp[i];
    double Wn[NT];
    double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
    double bc[4];
    double d0[4] = {0.0, 0.0, 0.0, 0.0}, zq1[4] = {0.0, 0.0, 0.0, 0.0};      // RATIO: per output of the group
    const bool rgt = RATIO && rt > 1.0;
    const double rm1 = rt - 1.0;
#pragma clang loop unroll(full)
    for (int b = 0; b < NB; ++b) {
      const int og = b / GB, f0 = RATIO ? ((b % GB) - 1) * 4 : (b % GB) * 4;
      const double *tc = t[b & 1];
      double *tn = t[(b + 1) & 1];
      if (RATIO && f0 < 0) {
        // header block of the group: tc[q] = lt[o][o] of its outputs o = 4 og + q, tc[4] = R_u(lt[0][0])
        const_f64 *tq = tp;
        asm volatile("" : "+s"(tq) : "v"(x0));
        if (b + 1 < NB) {
#pragma unroll
          for (int i = 0; i < BLK; ++i) tn[i] = tq[(b + 1) * BLK + i];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double z0 = tc[q] * rt;
          const double q0 = (z0 + M) - M;
          bad = bad | !(fabs(z0) < zlim);                      // fl(z + M) - M needs |z| < 2^(e-1)
          tmin = fmin(tmin, fabs(fabs(z0 - q0) - half_u));
          const double z1 = tc[q] * rm1;
          const double q1r = (z1 + M) - M;
          // (r <= 1: the term is absent; its tie test is neutralised with the distance 1.0)
          tmin = fmin(tmin, rgt ? fabs(fabs(z1 - q1r) - half_u) : 1.0);
          const double q1 = rgt ? q1r : 0.0;
          zq1[q] = q1;
          d0[q] = (q0 - q1) - ((4 * og + q == 0) ? tc[4] : 0.0);
        }
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      if (f0 == 0) {
        // x = max_f W[f] + tab[f][o]   (tab scalar; QUANT: it carries the from-index in its low bits)
#pragma unroll
        for (int q = 0; q < 4; ++q) bc[q] = bnx[q];
        if (P2_PF == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) bnx[q] = bny[q];
          if (og + 2 < NT / 4) prefetch2(bp, og + 2);
          else prefetch2(bpn, og + 2 - NT / 4);
        } else {
          if (og + 1 < NT / 4) prefetch(bp, og + 1);
          else prefetch(bpn, 0);
        }
        x0 = W[0] + tc[0];
        if (RATIO) x0 += 64.0 * d0[0];
      } else {
        x0 = fmax(x0, W[f0] + tc[0]);
      }
      const_f64 *tq = tp;
      asm volatile("" : "+s"(tq) : "v"(x0));
      if (b + 1 < NB) {
#pragma unroll
        for (int i = 0; i < BLK; ++i) tn[i] = tq[(b + 1) * BLK + i];
      }
      if (f0 == 0) {
        x1 = W[0] + tc[1];
        x2 = W[0] + tc[2];
        x3 = W[0] + tc[3];
        if (RATIO) {
          x1 += 64.0 * d0[1];
          x2 += 64.0 * d0[2];
          x3 += 64.0 * d0[3];
        }
      } else {
        x1 = fmax(x1, W[f0] + tc[1]);
        x2 = fmax(x2, W[f0] + tc[2]);
        x3 = fmax(x3, W[f0] + tc[3]);
      }
#pragma unroll
      for (int r = 1; r < 4; ++r) {
        x0 = fmax(x0, W[f0 + r] + tc[4 * r + 0]);
        x1 = fmax(x1, W[f0 + r] + tc[4 * r + 1]);
        x2 = fmax(x2, W[f0 + r] + tc[4 * r + 2]);
        x3 = fmax(x3, W[f0 + r] + tc[4 * r + 3]);
      }
      if (f0 + 4 < NT) {
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      const double xs[4] = {x0, x1, x2, x3};
      uint32_t pw = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int o = 4 * og + q;
        const double bo = bc[q];
        if (o == 0) bad = bad | (bo != bo);                   // (bitwise: no control flow in this body)
        if (QUANT) {
          // xs = 64 * (best value) + (63 - first arg-max) * u  (exact)
          const double bq = (bo + M) - M;
          // tie <=> |b - R_u(b)| == u/2 for some state: tracked as a running minimum of the distance to
          // u/2 (one compare per step; 36 compares would each occupy a scalar mask).  Pads hold b = 0.
          tmin = fmin(tmin, fabs(fabs(bo - bq) - half_u));
          // (a dead state: xs = -inf stays -inf through all of this; a live state below the range makes the item
          //  `bad` at the end of the step -- the new value is never above xs)
          const double y = xs[q] + CM;                         // exact
          const unsigned ylo = (unsigned)__double2loint(y);
          const double m6 = __hiloint2double(__double2hiint(y), (int)(ylo & ~63u)) - CM;   // xs with the index bits cleared
          Wn[o] = m6 + 64.0 * (RATIO ? bq + zq1[q] : bq);
          pw |= (~ylo & 63u) << (8 * q);                        // 63 - (xs / u mod 64)
        } else {
          Wn[o] = xs[q] + bo;
        }
      }
      // traceback bytes of this group (a tie position gets garbage: the exact chain rewrites it)
      if (QUANT && run && official && tb32) tb32[(p0 + t0 + s) * (NT / 4) + og] = pw;
      // pin the tie reduction to its group: left to itself the scheduler sinks all 36 of them to the end
      // of the step and keeps the whole emission row alive for it
      if (QUANT) asm volatile("" : "+v"(tmin));
      __builtin_amdgcn_sched_barrier(0);
    }
    const bool tie = QUANT && tmin == 0.0;
    if (QUANT) {
      // a rounding tie at this position: the vector is NOT advanced (W stays W_{t-1}); the next step
      // records it, closes the piece and restarts from zeros (see `pending` below)
#pragma unroll
      for (int j = 0; j < NT; ++j) W[j] = tie ? W[j] : Wn[j];
      pending = tie;
      double wl = INFINITY;
#pragma unroll
      for (int j = 0; j < NT; ++j) wl = fmin(wl, (j < NT - 8 || j < N) ? W[j] : INFINITY);
      if (!tie) pmin = fmin(pmin, wl * 0.015625 + base);
      // W = 64 (value - base) + index bits must stay inside the range of the arg-max extraction: |W| < 2^e - 192 u
      // (half of what exact representability alone would allow).  A state that falls further behind within a
      // re-basing window (or dies) makes the item unusable.
      bad = bad | (wl <= wlim);
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j) W[j] = Wn[j];
    }
  };
  // the tie found at position s - 1: close the piece (record W_{s-2}... i.e. the row of the position before
  // the tie), restart from zeros behind it; the exact chain handles the tie position itself
  auto close_piece = [&](int s) {
    if (!QUANT) return;
    if (__any(pending)) {
      if (pending) {
        if (run && s - 1 >= 0) {
          if (nt < TEHMM_LANE_MAXTI) {
            const VitItems vi = VI();
            vi.ties[item * TEHMM_LANE_MAXTI + nt] = s - 1;
            double *tr = vi.tierows + (item * TEHMM_LANE_MAXTI + nt) * NT;
#pragma unroll
            for (int j = 0; j < NT; ++j) tr[j] = W[j] * 0.015625 + base;
            vi.piecemin[item * (TEHMM_LANE_MAXTI + 1) + nt] = pmin;
          }
          ++nt;
        }
        restart();
      }
      pending = false;
    }
  };
  // every 16th position: QUANT records the row of an official position; every 32nd it first re-bases so
  // that the index bits keep fitting
  auto rebase = [&](int s, bool do_rebase) {
    if (!QUANT) return;
    if (do_rebase) {
      double mx = W[0];
#pragma unroll
      for (int j = 1; j < NT; ++j) mx = fmax(mx, W[j]);
      if (mx > -INFINITY) {
#pragma unroll
        for (int j = 0; j < NT; ++j) W[j] -= mx;
        base += mx * 0.015625;
      } else {
        bad = true;
      }
    }
    if (s >= 0 && run) {
      double *row = VC().rows + ((int64_t)c * (CS / TEHMM_VROW) + (t0 - ct0 + s) / TEHMM_VROW) * NT;
#pragma unroll
      for (int j = 0; j < NT; ++j) row[j] = W[j] * 0.015625 + base;
    }
  };
  auto vec_out = [&](double *dst) {
    const int64_t po = (((item >> 6) * NT) << 6) + (item & 63);
#pragma unroll
    for (int j = 0; j < NT; ++j) dst[po + ((int64_t)j << 6)] = QUANT ? W[j] * 0.015625 + base : W[j];
  };
  auto vec_max = [&]() {
    double mx = W[0];
#pragma unroll
    for (int j = 1; j < NT; ++j) mx = fmax(mx, W[j]);
    return mx;
  };

  double g0 = 0.0, rnx = 1.0;
  for (int s = -Wu; s < L; ++s) {                       // one loop: the unrolled step exists once
    close_piece(s);
    if (s == 0) {
      if (run) vec_out(VI().pre);
      pmin = INFINITY;
      if (!QUANT) g0 = vec_max();
    }
    const double *bp = B + (s < 0 ? lane_row(lg, NT, nb, L + s) : lane_row(lg, NT, item, s));
    const int sn = min(s + 1, L - 1);
    const double *bpn = B + (sn < 0 ? lane_row(lg, NT, nb, L + sn) : lane_row(lg, NT, item, sn));
    if (s == -Wu) {
      prefetch(bp, 0);
      if (P2_PF == 2) prefetch2(bp, 1);
    }
    double rt = 1.0;
    if (RATIO) {
      // (positions t0 + s of the interval: the warm-up reads the previous item's; one step ahead)
      if (s == -Wu) rnx = run ? ratios[p0 + t0 + s] : 1.0;
      rt = rnx;
      rnx = (run && s + 1 < L) ? ratios[p0 + t0 + s + 1] : 1.0;
    }
    step(bp, bpn, s, rt);
    if (((s + Wu) & (TEHMM_VROW - 1)) == TEHMM_VROW - 1) rebase(s, ((s + Wu) & 31) == 31);
  }
  close_piece(L);
  if (run) {
    const VitItems vi = VI();
    vec_out(vi.end);
    if (QUANT) {
      vi.ntie[item] = nt;
      vi.bad[item] = (bad || nt > TEHMM_LANE_MAXTI) ? 1 : 0;
      vi.piecemin[item * (TEHMM_LANE_MAXTI + 1) + min(nt, TEHMM_LANE_MAXTI)] = pmin;
    } else {
      const double g1 = vec_max();
      vi.gain[item] = bad ? __longlong_as_double(0x7ff8000000000000LL) : g1 - g0;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Links of the Viterbi lane pass, one wave per chunk (lane = state): items whose pre vector equals
// the previous item's end vector up to ONE constant (exactly) continue its segment -- their rows,
// tie rows and minima are shifted into the segment's frame; a failed link, like a rounding tie,
// starts a new segment (recorded as a tie at the item's first position).  Output: the chunk's
// segment list in the form k_vit_fix expects (ties, tierows, rows, segmin, ok).
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k_vit_stitch(IntervalTab iv, LaneGeom lg, VitChunks vc, VitItems vi,
                                                    int N, int soft = 0) {
  // soft != 0: the ties the items recorded are SOFT (the pass kept its frame through them, tehmm_lane3.hip.h): rows,
  // tie rows and minima behind them stay in the segment's frame; only failed item links break it
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= vc.n) return;
  if (vc.e[c] == TEHMM_SPEC_NONE) {
    if (lane == 0) { vc.ok[c] = 0; vc.ntie[c] = 0; vc.offend[c] = 0.0; if (vc.tsoft) { vc.tsoft[c] = 0u; vc.tpar[c] = 0u; } }
    return;
  }
  unsigned softmask = 0u, parmask = 0u;
  const double uinv = ldexp(1.0, 52 - vc.e[c]);           // 1 / u: offsets are multiples of the grid unit
  auto odd_mult = [&](double x) { const double q = x * uinv * 0.5; return q != trunc(q); };
  const int id = vc.iv[c];
  const int64_t ct0 = vc.t0[c];
  const int L = lg.L, SUB = vc.CS / L, R = L / TEHMM_VROW;
  const int64_t item0 = lg.ifirst[id] + ct0 / L;
  const bool live = lane < N;
  const int jl = min(lane, NT - 1);
  auto at = [&](const double *p, int64_t item) { return p[((((item >> 6) * NT) + jl) << 6) + (item & 63)]; };
  double off = 0.0, off_end = 0.0, segmin = INFINITY;
  int nT = 0;
  bool okc = true;
  for (int k = 0; k < SUB; ++k) {
    const int64_t item = item0 + k;
    okc = okc && vi.bad[item] == 0;
    off = off_end;
    if (k > 0) {
      const double a = at(vi.pre, item), bp = at(vi.end, item - 1);
      const bool both_dead = a == -INFINITY && bp == -INFINITY;
      const double d = bp - a;
      const double d0 = wave_max_live((both_dead || !live) ? -INFINITY : d, live);
      const bool same = __all(!live || both_dead || d == d0) && d0 == d0 && d0 > -INFINITY && d0 < INFINITY;
      if (same) {
        off = off_end + d0;
      } else {
        if (nT < TEHMM_SPEC_MAXT) {
          if (lane == 0) {
            vc.ties[(int64_t)c * TEHMM_SPEC_MAXT + nT] = k * L;
            vc.segmin[(int64_t)c * (TEHMM_SPEC_MAXT + 1) + nT] = segmin;
          }
          if (lane < NT) vc.tierows[((int64_t)c * TEHMM_SPEC_MAXT + nT) * NT + lane] = bp + off_end;
        }
        ++nT;
        off = 0.0;
        segmin = INFINITY;
      }
    }
    const int nti = min(vi.ntie[item], TEHMM_LANE_MAXTI);
    const int first_tie = nti > 0 ? vi.ties[item * TEHMM_LANE_MAXTI] : L;
    if (off != 0.0 && lane < NT) {
      for (int m = 0; m < R && (soft || TEHMM_VROW * m + TEHMM_VROW - 1 < first_tie); ++m)
        vc.rows[((int64_t)c * (vc.CS / TEHMM_VROW) + k * R + m) * NT + lane] += off;
    }
    segmin = fmin(segmin, vi.piecemin[item * (TEHMM_LANE_MAXTI + 1)] + off);
    for (int i = 0; i < nti; ++i) {
      if (nT < TEHMM_SPEC_MAXT) {
        if (lane == 0) {
          vc.ties[(int64_t)c * TEHMM_SPEC_MAXT + nT] = k * L + vi.ties[item * TEHMM_LANE_MAXTI + i];
          vc.segmin[(int64_t)c * (TEHMM_SPEC_MAXT + 1) + nT] = segmin;
        }
        if (lane < NT)
          vc.tierows[((int64_t)c * TEHMM_SPEC_MAXT + nT) * NT + lane] =
              vi.tierows[(item * TEHMM_LANE_MAXTI + i) * NT + lane] + ((i == 0 || soft) ? off : 0.0);
        if (soft) {
          softmask |= 1u << nT;
          if (odd_mult(off)) parmask |= 1u << nT;        // passable for a chain delta of this parity
        }
      }
      ++nT;
      segmin = vi.piecemin[item * (TEHMM_LANE_MAXTI + 1) + i + 1] + (soft ? off : 0.0);
    }
    off_end = (nti > 0 && !soft) ? 0.0 : off;
  }
  if (lane == 0) {
    vc.segmin[(int64_t)c * (TEHMM_SPEC_MAXT + 1) + min(nT, TEHMM_SPEC_MAXT)] = segmin;
    vc.ntie[c] = nT;
    vc.ok[c] = (okc && nT <= TEHMM_SPEC_MAXT) ? 1 : 0;
    vc.offend[c] = off_end;
    if (vc.tsoft) { vc.tsoft[c] = softmask; vc.tpar[c] = parmask; }
  }
}

// Links BETWEEN chunks (after k_vit_stitch, one wave per chunk): chunk c continues chunk c - 1 when both
// are usable, live in the same binade and the vector before c's first position is the previous chunk's
// end vector plus ONE constant (exactly).  k_vit_fix then lets a verified jump run on through the linked
// chunks up to the next tie: one verification per run of chunks instead of one per chunk.
template <int NT>
__global__ __launch_bounds__(256) void k_vit_links(IntervalTab iv, LaneGeom lg, VitChunks vc, VitItems vi, int N) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= vc.n) return;
  const int id = vc.iv[c];
  bool cand = c > vc.first[id] && vc.e[c] != TEHMM_SPEC_NONE && vc.e[c - 1] == vc.e[c] && vc.ok[c] != 0 &&
              vc.ok[c - 1] != 0;
  int link = 0;
  double lk = 0.0;
  if (cand) {
    const int L = lg.L;
    const int64_t item0 = lg.ifirst[id] + vc.t0[c] / L;
    const bool live = lane < N;
    const int jl = min(lane, NT - 1);
    auto at = [&](const double *p, int64_t item) { return p[((((item >> 6) * NT) + jl) << 6) + (item & 63)]; };
    const double a = at(vi.pre, item0);                              // frame of c's first segment
    const double bp = at(vi.end, item0 - 1) + vc.offend[c - 1];      // frame of c - 1's last segment
    const bool both_dead = a == -INFINITY && bp == -INFINITY;
    const double d = bp - a;
    const double d0 = wave_max_live((both_dead || !live) ? -INFINITY : d, live);
    const bool same = __all(!live || both_dead || d == d0) && d0 == d0 && d0 > -INFINITY && d0 < INFINITY;
    // (an item of c - 1 that ended on a pending tie restarted from zeros: then its end vector is the zero
    //  vector of a segment that begins at c's first position, which k_vit_stitch has recorded as a tie at
    //  the chunk end -- such chunks do not link: their last "tie" is at CS)
    if (same && vi.bad[item0 - 1] == 0 && vi.bad[item0] == 0) {
      link = 1;
      lk = d0;
    }
  }
  if (lane == 0) {
    vc.clink[c] = link;
    vc.clk[c] = lk;
  }
}


// Runs of linked chunks (after k_vit_links, one wave per interval): for every chunk, where a jump that
// enters its first segment lands -- the chunk's first tie, else on through the linked chunks behind it to
// their first tie or the end of the run -- with the frame offset and the lowest live value collected on the
// way.  A suffix scan over the interval's chunks (tiles of 64 from the back) of the maps
//   STOP(target, sel, mn)                    chunk with a tie, or the last of its run
//   PASS(k, s): (target, sel, acc, mn) -> (target, sel, k + acc, min(s, k + mn))     k = clk of the next chunk
// whose composition is associative and exact (every term is a multiple of the binade's grid unit).
// Round 4 (soft ties): the scan is kept per parity h of the chain's verified delta on entering the chunk's first
// segment.  Entry k of a chunk's tie list lets a chain of parity h pass iff it is soft and bit k of tpar equals h;
// the first entry that does not stops it.  Passing on into the next linked chunk changes the parity by that of the
// link constant (delta_next = delta + clk).  A chunk is therefore a pair of maps with an outgoing parity each:
struct VitRunMap {
  int stop;
  int64_t target, sel;
  double acc, mn;
};
struct VitRunMap2 {
  VitRunMap m[2];
  int hout[2];
};
__device__ __forceinline__ VitRunMap2 vit_run_compose(const VitRunMap2 &a, const VitRunMap2 &b) {   // a, then b
  VitRunMap2 r;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (a.m[h].stop) {
      r.m[h] = a.m[h];
      r.hout[h] = 0;
    } else {
      const int h2 = a.hout[h];
      const VitRunMap &bb = h2 ? b.m[1] : b.m[0];
      r.m[h].stop = bb.stop;
      r.m[h].target = bb.target;
      r.m[h].sel = bb.sel;
      r.m[h].acc = a.m[h].acc + bb.acc;
      r.m[h].mn = fmin(a.m[h].mn, a.m[h].acc + bb.mn);
      r.hout[h] = h2 ? b.hout[1] : b.hout[0];
    }
  }
  return r;
}
__global__ __launch_bounds__(64) void k_vit_runs(VitChunks vc, int n_iv) {
  const int lane = threadIdx.x;
  const int id = blockIdx.x;
  if (id >= n_iv) return;
  const int64_t c0 = vc.first[id], c1 = vc.first[id + 1];
  const int64_t nc = vc.n;
  VitRunMap2 carry;
  carry.m[0] = carry.m[1] = VitRunMap{0, 0, 0, 0.0, INFINITY};   // identity (the interval's last chunk always stops)
  carry.hout[0] = 0;
  carry.hout[1] = 1;
  for (int64_t hi = c1; hi > c0; hi -= 64) {
    const int64_t c = hi - 64 + lane;                        // lanes ascending in chunk order; tile = [hi - 64, hi)
    const bool in = c >= c0;
    VitRunMap2 m;
    m.m[0] = m.m[1] = VitRunMap{0, 0, 0, 0.0, INFINITY};
    m.hout[0] = 0;
    m.hout[1] = 1;
    if (in) {
      const double *sm = vc.segmin + c * (TEHMM_SPEC_MAXT + 1);
      const bool linked_next = c + 1 < c1 && vc.clink[c + 1] != 0;
      const int nt = min(vc.ntie[c], TEHMM_SPEC_MAXT);
      const unsigned soft = vc.tsoft ? vc.tsoft[c] : 0u, par = vc.tsoft ? vc.tpar[c] : 0u;
      int kodd = 0;
      if (linked_next) {
        const double q = vc.clk[c + 1] * ldexp(1.0, 52 - vc.e[c]) * 0.5;
        kodd = q != trunc(q);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int kh = -1;
        double mn = sm[0];
        for (int k = 0; k < nt; ++k) {
          if (!((soft >> k) & 1u) || (int)((par >> k) & 1u) != h) { kh = k; break; }
          mn = fmin(mn, sm[k + 1]);
        }
        if (kh >= 0) m.m[h] = VitRunMap{1, vc.t0[c] + vc.ties[c * TEHMM_SPEC_MAXT + kh], 2 * (64 * c + kh), 0.0, mn};
        else if (linked_next) m.m[h] = VitRunMap{0, 0, 0, vc.clk[c + 1], mn};
        else m.m[h] = VitRunMap{1, vc.t0[c] + vc.CS, 2 * (64 * c) + 1, 0.0, mn};
        m.hout[h] = h ^ kodd;
      }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      VitRunMap2 o;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        o.m[h].stop = __shfl_down(m.m[h].stop, d);
        o.m[h].target = __shfl_down(m.m[h].target, d);
        o.m[h].sel = __shfl_down(m.m[h].sel, d);
        o.m[h].acc = __shfl_down(m.m[h].acc, d);
        o.m[h].mn = __shfl_down(m.m[h].mn, d);
        o.hout[h] = __shfl_down(m.hout[h], d);
      }
      if (lane + d < 64) m = vit_run_compose(m, o);
    }
    m = vit_run_compose(m, carry);
    if (in) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        vc.rtarget[h * nc + c] = m.m[h].target;
        vc.rsel[h * nc + c] = m.m[h].sel;
        vc.racc[h * nc + c] = m.m[h].acc;
        vc.rmn[h * nc + c] = m.m[h].mn;
      }
    }
    // carry = the map of the tile's first chunk (lane 0 is the lowest chunk index of the tile, or out of range
    // in the interval's first tile, after which the loop ends)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      carry.m[h].stop = __shfl(m.m[h].stop, 0);
      carry.m[h].target = __shfl(m.m[h].target, 0);
      carry.m[h].sel = __shfl(m.m[h].sel, 0);
      carry.m[h].acc = __shfl(m.m[h].acc, 0);
      carry.m[h].mn = __shfl(m.m[h].mn, 0);
      carry.hout[h] = __shfl(m.hout[h], 0);
    }
  }
}


// ------------------------------------------------------------------------------------------
// P0 of the exact chunk-parallel Viterbi as a lane = item pass in PACKED FLOAT arithmetic.  P0 only
// has to place every chunk in its fp64 binade (the exact kernels re-verify whatever they use), so the
// score gained over an item needs ~1e-4 relative accuracy: float max-plus over 512 + 64 positions
// from a zero start is far inside that.  Two outputs share a register pair: v_pk_add_f32 adds the
// broadcast W[f] to a scalar pair of table entries, v_max3_f32 folds two candidates per half.
// gain[item] = max W_end - max W_pre (the converged vector's decrease over the official range).
// tabf: [o / 2][f][o % 2] floats (NT * NT), pads -inf.
// ------------------------------------------------------------------------------------------
typedef float lane_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(4))) const lane_f2 const_f2;

template <int NT>
__global__ __launch_bounds__(256) void k_vit_gain_lane(IntervalTab iv, LaneGeom lg, int N, int CS, int Wu,
                                                       const float *__restrict__ tabf,
                                                       const float *__restrict__ B32, double *gain) {
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= lg.n_groups) return;
  const int L = lg.L;
  const int64_t item = (int64_t)g * 64 + lane;
  const bool valid = item < lg.n_items;
  const int id = valid ? lg.item_iv[item] : 0;
  const int64_t t0 = valid ? lg.item_t0[item] : 0;
  const int64_t T = iv.len[id];
  const int64_t ct0 = (t0 / CS) * CS;
  const bool run = valid && ct0 > 0 && ct0 + CS <= T;
  if (!__any(run)) return;
  const int64_t nb = run ? item - 1 : item;
  lane_f2 W[NT / 2];
#pragma unroll
  for (int j = 0; j < NT / 2; ++j)
    W[j] = (lane_f2){2 * j < N ? 0.f : -INFINITY, 2 * j + 1 < N ? 0.f : -INFINITY};
  bool bad = false;
  auto vec_max = [&]() {
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NT / 2; ++j) mx = fmaxf(mx, fmaxf(W[j].x, W[j].y));
    return mx;
  };
  float g0 = 0.f;
  for (int s = -Wu; s < L; ++s) {
    if (s == 0) g0 = vec_max();
    const float *bp = B32 + (s < 0 ? lane_row(lg, NT, nb, L + s) : lane_row(lg, NT, item, s));
    float b[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = bp[(int64_t)j << 6];
    bad = bad | (b[0] != b[0]);
    int z = 0;
    asm volatile("" : "+s"(z));
    const_f2 *tp = (const_f2 *)(size_t)tabf + z;
    lane_f2 x[NT / 2];
#pragma unroll
    for (int op = 0; op < NT / 2; ++op) {
      lane_f2 acc = (lane_f2){W[0].x, W[0].x} + tp[op * NT];
#pragma unroll
      for (int f = 1; f < NT; ++f) {
        const float wf = (f & 1) ? W[f >> 1].y : W[f >> 1].x;
        acc = __builtin_elementwise_max(acc, (lane_f2){wf, wf} + tp[op * NT + f]);
      }
      x[op] = acc;
    }
#pragma unroll
    for (int j = 0; j < NT / 2; ++j) W[j] = x[j] + (lane_f2){b[2 * j], b[2 * j + 1]};
  }
  if (run) gain[item] = bad ? __longlong_as_double(0x7ff8000000000000LL) : (double)vec_max() - (double)g0;
}

// ------------------------------------------------------------------------------------------
// Emission rows + P0 in ONE pass (round 2).  k_emis_lane wrote the log rows twice (fp64 for the exact pass,
// rounded to float for P0) and k_vit_gain_lane read the float copy back: 297 bytes per position of HBM
// traffic for a pass whose arithmetic fits under the row kernel's own stores.  Here the lane that computes
// the rows of its item (reference operation order, _emission.pyx:65-72) feeds them, rounded to float, straight
// into the packed-float max-plus recurrence of P0 -- starting Wu positions early for the warm-up (those rows
// are recomputed, not stored: + Wu / L work).  Outputs: B (fp64 log rows, item-interleaved; NaN rows where no
// state can emit) and gain[item] as k_vit_gain_lane.
// ------------------------------------------------------------------------------------------
template <int NT, bool RATIO = false>
#ifndef TEHMM_EMIS_WAVES
#define TEHMM_EMIS_WAVES 2
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TEHMM_EMIS_WAVES, TEHMM_EMIS_WAVES))) void k_emis_gain_lane(IntervalTab iv, EmisTab em, LaneGeom lg, int N, int CS, int Wu,
                                                        const float *__restrict__ tabf, double *B, double *gain,
                                                        const double *__restrict__ ratios = nullptr) {
  extern __shared__ double emis_ltab[];
  stage_emis_table(em, emis_ltab, NT);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= lg.n_groups) return;
  const int L = lg.L;
  const int64_t item = (int64_t)g * 64 + lane;
  const bool valid = item < lg.n_items;
  const int id = valid ? lg.item_iv[item] : 0;
  const int64_t t0 = valid ? lg.item_t0[item] : 0;
  const int64_t T = iv.len[id], p0 = iv.pos0[id];
  const int len = valid ? (int)min((int64_t)L, T - t0) : 0;
  const int maxlen = wave_max_i32(len);
  const int64_t ct0 = (t0 / CS) * CS;
  const bool run = valid && ct0 > 0 && ct0 + CS <= T;       // P0 runs on full chunks but an interval's first
  const int s_first = __any(run) ? -Wu : 0;
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  lane_f2 W[NT / 2];
#pragma unroll
  for (int j = 0; j < NT / 2; ++j)
    W[j] = (lane_f2){2 * j < N ? 0.f : -INFINITY, 2 * j + 1 < N ? 0.f : -INFINITY};
  bool bad = false;
  auto vec_max = [&]() {
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NT / 2; ++j) mx = fmaxf(mx, fmaxf(W[j].x, W[j].y));
    return mx;
  };
  float g0 = 0.f;
  for (int s = s_first; s < maxlen; ++s) {
    const bool act = s >= 0 ? s < len : run;
    const int64_t gpos = p0 + t0 + (act ? s : 0);
    double x[NT];
    emis_rows<NT>(em, emis_ltab, gpos, x);
    double m = x[0];
#pragma unroll
    for (int j = 1; j < NT; ++j) m = fmax(m, j < N ? x[j] : -INFINITY);
    const bool good = m > -1e20;
    if (B && s >= 0 && act) {
      const int64_t o = ((((int64_t)g * lg.L + s) * NT) << 6) + lane;
#pragma unroll
      for (int j = 0; j < NT; ++j) B[o + ((int64_t)j << 6)] = good ? x[j] : qnan;
    }
    // ---- P0 step on the row rounded to float (see k_vit_gain_lane)
    if (s == 0) g0 = vec_max();
    bad = bad | (act && !good);
    int z = 0;
    asm volatile("" : "+s"(z));
    const_f2 *tp = (const_f2 *)(size_t)tabf + z;
    lane_f2 xn[NT / 2];
    // RATIO: the self-transition terms of the segment ratio (see k_vit_lane); tabf carries lt[o][o] behind the table
    float rf = 1.f, rm1 = 0.f;
    if (RATIO) {
      rf = (float)ratios[gpos];
      rm1 = rf > 1.f ? rf - 1.f : 0.f;
    }
#pragma unroll
    for (int op = 0; op < NT / 2; ++op) {
      lane_f2 acc = (lane_f2){W[0].x, W[0].x} + tp[op * NT];
      if (RATIO) {
        const lane_f2 ltd = tp[NT * NT / 2 + op];
        acc = acc + ltd * (lane_f2){rf - rm1, rf - rm1};
        if (op == 0) acc.x -= ((const float *)(size_t)tabf)[z + NT * NT + NT];
      }
#pragma unroll
      for (int f = 1; f < NT; ++f) {
        const float wf = (f & 1) ? W[f >> 1].y : W[f >> 1].x;
        acc = __builtin_elementwise_max(acc, (lane_f2){wf, wf} + tp[op * NT + f]);
      }
      xn[op] = acc;
    }
#pragma unroll
    for (int j = 0; j < NT / 2; ++j) {
      W[j] = xn[j] + (lane_f2){(float)x[2 * j], (float)x[2 * j + 1]};
      if (RATIO) W[j] = W[j] + tp[NT * NT / 2 + j] * (lane_f2){rm1, rm1};
    }
  }
  if (run) gain[item] = (bad || len < L) ? qnan : (double)vec_max() - (double)g0;
}

}  // namespace tehmm
