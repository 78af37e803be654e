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



// ------------------------------------------------------------------------------------------
// Emission rows of every position, item-interleaved.  B = log rows (the reference's operation
// order, _emission.pyx:65-72), BH = exp(B - rowmax), MS = rowmax.  A row no state can emit is
// written as NaN: it poisons the lane passes, their links fail and the exact chain -- which owns the
// reference's semantics for such rows (leading-rows quirk, dead lattices) -- walks through it.
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k_emis_lane(IntervalTab iv, EmisTab em, LaneGeom lg, int N,
                                                   double *B, double *BH, double *MS) {
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
    emis_rows<NT>(em, nullptr, gpos, x);
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

  // out[o] = sum_f v[f] * tab[f][o]
  // (the table pointer is laundered every step: hoisting 1296 loop-invariant scalar loads out of the
  // position loop would spill thousands of SGPRs; re-issued s_loads hit the scalar cache)
  auto matvec = [&](double (&acc)[NT]) {
    int z = 0;
    asm volatile("" : "+s"(z));
    const_f64 *tp = (const_f64 *)(size_t)tab + z;
#pragma unroll
    for (int o = 0; o < NT; ++o) acc[o] = 0.0;
#pragma unroll
    for (int f = 0; f < NT; ++f) {
#pragma unroll
      for (int o = 0; o < NT; ++o) acc[o] = fma(v[f], tp[f * NT + o], acc[o]);
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
// Links of the forward / backward lane pass, one wave per chunk (lane = state).  A chunk is usable
// by the fix-up chain when the vectors of consecutive items agree in direction (Hilbert distance
// <= TEHMM_FB_TOL) at every item boundary inside the chunk.  Forward also lays the items' log-scale
// records end to end: scale[c][CS/32] in the frame of the chunk's first item.
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k_fb_stitch(IntervalTab iv, LaneGeom lg, FbChunks fc, int N,
                                                   const double *pre_f, const double *end_f,
                                                   const double *slog32, const double *pre_b,
                                                   const double *end_b, int *ok_f, int *ok_b) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= fc.n) return;
  const int id = fc.iv[c];
  const int64_t T = iv.len[id];
  const int64_t ct0 = fc.t0[c];
  const int L = lg.L, SUB = fc.CS / L, R = L / 32;
  const bool full = ct0 + fc.CS <= T;
  const int64_t item0 = lg.ifirst[id] + ct0 / L;
  const bool live = lane < N;
  const int jl = min(lane, NT - 1);
  auto at = [&](const double *p, int64_t item) {
    return live ? p[((((item >> 6) * NT) + jl) << 6) + (item & 63)] : 0.0;
  };
  bool okf = full && c != fc.first[id];
  if (okf) {
    double off = 0.0;
    for (int k = 0; k < SUB; ++k) {
      const int64_t item = item0 + k;
      if (k > 0) {
        double rho;
        const double d = proj_dist(at(pre_f, item), at(end_f, item - 1), live, rho);
        okf = okf && d <= TEHMM_FB_TOL;
        off += slog32[(item - 1) * R + R - 1] - log(rho);
      }
      for (int m = lane; m < R; m += 64) fc.scale[(int64_t)c * (fc.CS / 32) + k * R + m] = slog32[item * R + m] + off;
    }
  }
  bool okb = full && ct0 + fc.CS < T;
  if (okb) {
    for (int k = SUB - 2; k >= 0; --k) {
      double rho;
      const double d = proj_dist(at(pre_b, item0 + k), at(end_b, item0 + k + 1), live, rho);
      okb = okb && d <= TEHMM_FB_TOL;
    }
    if (lane < NT) fc.wstart[(int64_t)c * NT + lane] = at(end_b, item0);
  }
  if (lane == 0) {
    ok_f[c] = okf ? 1 : 0;
    ok_b[c] = okb ? 1 : 0;
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
  __shared__ double tile[64 * (NT + 1)];
  __shared__ double rs[64];
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
  const double epsden = 1.0 + (double)N * eps;
  __syncthreads();
  for (int s = s0; s < s0 + 8; ++s) {
    const int64_t o = (((int64_t)g * L + s) * NT) << 6;
    for (int idx = tid; idx < NT * 64; idx += 256) {
      const int j = idx >> 6, ln = idx & 63;
      double p = 0.0;
      if (s < rlen[ln]) p = al[o + idx] * be[o + idx];
      tile[ln * (NT + 1) + j] = p;
    }
    __syncthreads();
    if (tid < 64) {
      double t = 0.0;
      for (int j = 0; j < N; ++j) t += tile[tid * (NT + 1) + j];
      rs[tid] = t;
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * N; idx += 256) {
      const int r = idx / N, col = idx - r * N;
      if (s < rlen[r]) {
        double pr = tile[r * (NT + 1) + col] / rs[r];
        if (EPS) pr = (pr + eps) / epsden;
        post[(pb[r] + s) * N + col] = pr;
      }
    }
    __syncthreads();
  }
}

}  // namespace tehmm
