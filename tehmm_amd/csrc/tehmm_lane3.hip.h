// tehmm_lane3.hip.h -- the exact quantised Viterbi pass (P2) of tehmm_lane.hip.h with the OUTPUT states of a
// step split over the NW waves of a workgroup (round 4).
//
// k_vit_lane keeps the vector W[NT] and its successor Wn[NT] of one item per lane: 4 * NT VGPRs before anything
// else, 282 registers at 36 states = ONE wave per SIMD, which then waits 30 % of its time for its own scalar
// operand stream and emission loads with nobody to fill the gaps, and carries 72 v_cndmask per step for "a lane
// that met a rounding tie does not advance".  A full bipartite step (every output needs every input) cannot live
// in fewer than 2 * NT doubles per lane -- so the step is split over waves instead:
//   * a work unit (64 items of one binade) is one workgroup of NW waves; wave w owns the output groups
//     [w * GW, (w + 1) * GW) of every step (GW = ceil(NT / 4 / NW)): W[NT] + Wn[4 GW] doubles per lane,
//     ~150 registers at 36 states and NW = 3 -> three waves per SIMD (64 states: two instead of one with spills);
//   * the new vector is exchanged through LDS: two buffers [NT / 2][64] double2 (lane-contiguous 16-byte slots:
//     conflict-free ds_write_b128 / ds_read_b128), a step reads buffer s & 1 and writes the other: ONE workgroup
//     barrier per step, behind which the waves of two other workgroups keep the SIMD's fp64 pipe busy;
//   * a rounding tie is a property of the emission row (and the ratio products), found by the wave that owns the
//     output: the waves exchange their tie ballots with the vector, and a lane that met a tie simply does NOT read
//     the new vector (an EXEC-masked load: its registers still hold W_{t-1}) -- no select per state;
//   * every wave runs the SAME code (a third of k_vit_lane's body: 10 KB of instructions instead of 29 KB in the
//     instruction cache two CUs share), the wave index only moves the table / row / traceback pointers.
// Outputs are those of k_vit_lane bit for bit (rows, pre / end vectors, ties, tie rows, piece minima, traceback
// bytes): the arithmetic is exact integer max-plus, so k_vit_stitch and the exact chain see no difference.
#pragma once
#include "tehmm_lane.hip.h"

namespace tehmm {

#ifndef TEHMM_P2_NW
#define TEHMM_P2_NW 3
#endif
// timing experiments only (results are wrong with any bit set): 1 no barrier, 2 no table stream, 4 no LDS vector read,
// 8 no emission loads, 16 emission loads from one row (cache hits), 32 no traceback stores
#ifndef TEHMM_L3_EXP
#define TEHMM_L3_EXP 0
#endif
typedef double lane3_d2 __attribute__((ext_vector_type(2)));
// Pointers read out of the chunk / item tables are generic to the compiler, and ONE flat store inside the step loop makes
// its waitcnt pass drain vmcnt to zero at every use of a prefetched emission value (flat operations may return out of
// order): 5 of the pass's 17 ms.  They are global pointers; say so.
typedef __attribute__((address_space(1))) double lane3_gf64;
typedef __attribute__((address_space(1))) int lane3_gi32;
#define LANE3_G(p) ((lane3_gf64 *)(p))
#define LANE3_GI(p) ((lane3_gi32 *)(p))
template <typename T>
__device__ __forceinline__ T lane3_gload(T const *p) {
  return *(const __attribute__((address_space(1))) T *)p;
}

#ifdef TEHMM_STAMPS
#define L3ST(i) do { const unsigned long long n_ = stamp_now(); l3st[i] += n_ - l3tt; l3tt = n_; } while (0)
#else
#define L3ST(i) do { } while (0)
#endif

template <int NT, int NW>
struct Lane3Geom {
  static constexpr int G = NT / 4;                       // output groups of a step
  static constexpr int GW = (G + NW - 1) / NW;           // ... per wave
  static constexpr size_t VEC_BYTES = (size_t)2 * (NT / 2) * 64 * sizeof(lane3_d2);
  static constexpr size_t LDS_BYTES = VEC_BYTES + 2 * 8 * sizeof(unsigned long long) + (size_t)NW * 64 * sizeof(double);
};

// waves per SIMD the register allocator is held to: three up to 36 states (<= 168 registers), two above
template <int NT> constexpr int lane3_waves() { return NT <= 36 ? 3 : 2; }

template <int NT, int NW, bool RATIO>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(lane3_waves<NT>(), lane3_waves<NT>())))
void k_vit_lane3(IntervalTab iv, LaneGeom lg, const VitChunks *vcp, const VitItems *vip, int N, int Wu, const int *wk_g,
                 const int *wk_e, int n_work, const double *__restrict__ tabs, int e0,
                 const double *__restrict__ B, uint8_t *tb, const double *__restrict__ ratios,
                 const int *__restrict__ wk_items, const int *__restrict__ n_work_dev = nullptr, int soft_ties = 0) {
  using GEO = Lane3Geom<NT, NW>;
  // Soft ties (round 4, without segment ratios): a rounding tie does not end the speculation.  The true value of a
  // state is its frame value plus ONE unknown constant delta (a multiple of the grid unit u); fl(v + b) for a b exactly
  // between two grid points rounds to the EVEN neighbour, i.e. depends on the parity of v / u.  The pass goes through
  // the tie assuming delta / u EVEN -- every candidate of a tied output is rounded on its own frame parity -- and
  // keeps its frame.  The exact chain knows delta when it has verified a chunk: if it is even, the pass was exact
  // through every tie of the run and the chain jumps over them; if it is odd, the chain lands on the next tie as
  // before (the tie and the row before it are still recorded), walks it exactly and verifies again behind it.
  const bool soft = !RATIO && soft_ties != 0;
  int pbase = 0;                                           // parity of base / u
  constexpr int G = GEO::G, GW = GEO::GW;
  constexpr bool EVEN = G % NW == 0;
  extern __shared__ lane3_d2 lane3_lds[];
  lane3_d2 *Wl = lane3_lds;                                                    // [2][NT / 2][64]
  unsigned long long *tieb = (unsigned long long *)(lane3_lds + 2 * (NT / 2) * 64);   // [2][8]
  double *xch = (double *)(tieb + 16);                                         // [NW][64]
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wk = blockIdx.x;
  if (wk >= (n_work_dev ? *n_work_dev : n_work)) return;      // (device-side placement: the grid is an upper bound)
  const int gq = wk_g[wk];
  const int e = wk_e[wk];
  // (the chunk / item tables are reached through pointers re-read in the rare paths, field by field, as GLOBAL loads)
  auto VC = [&]() { const VitChunks *q = vcp; asm volatile("" : "+s"(q)); return q; };
  auto VI = [&]() { const VitItems *q = vip; asm volatile("" : "+s"(q)); return q; };
  const int L = lg.L, CS = vcp->CS;
  int64_t item = (int64_t)gq * 64 + lane;
  bool valid = item < lg.n_items;
  if (gq < 0) {
    const int li = wk_items[(int64_t)(-1 - gq) * 64 + lane];
    valid = li >= 0;
    item = valid ? li : 0;
  }
  const int id = valid ? lg.item_iv[item] : 0;
  const int64_t t0 = valid ? lg.item_t0[item] : 0;
  const int64_t T = iv.len[id], p0 = iv.pos0[id];
  const int64_t ct0 = (t0 / CS) * CS;
  const int64_t c = vcp->first[id] + t0 / CS;
  const bool run = valid && ct0 > 0 && ct0 + CS <= T && vcp->e[c] == e;
  if (!__any(run)) return;                               // (the same lanes in every wave: uniform over the workgroup)
  const int64_t nb = run ? item - 1 : item;
#ifdef TEHMM_L3_ROT
  const int wr = (w + TEHMM_L3_ROT) % NW;                // (experiment: which wave takes which third)
#else
  const int wr = w;
#endif
  const int og0 = wr * GW;                               // this wave's output groups [og0, og0 + ng)
  const int ng = min(GW, G - og0);
  constexpr int BLK = 16, GB = NT / 4 + (RATIO ? 1 : 0);
  constexpr int TABSZ = RATIO ? NT * NT + (NT / 4) * 16 : NT * NT;
  const_f64 *tab0 = (const_f64 *)(size_t)tabs + (int64_t)(e - e0) * TABSZ + (int64_t)og0 * GB * BLK;
  const double u = ldexp(1.0, e - 52);
  const double M = ldexp(1.5, e);
  const double half_u = 0.5 * u;
  const double CM = ldexp(1.0, e + 1) - ldexp(1.0, e - 45);
  const double zlim = ldexp(1.0, e - 1);
#ifdef TEHMM_STAMPS
  unsigned long long l3st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, l3tt = stamp_now();
#endif
  double W[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) W[j] = j < N ? 0.0 : -INFINITY;
  double base = 0.0, pmin = INFINITY, wl_prev = INFINITY;
  int nt = 0;
  bool bad = false;
  uint32_t *tb32 = (uint32_t *)tb;

  // the emission values of this wave's outputs (requested at the top of the step)
  double bnx[4 * GW];
  // the emission row of a position is NT * 64 doubles behind the previous one (item-interleaved layout); the warm-up
  // reads the end of the previous item.  One 64-bit pointer per lane, advanced by a constant: no 64-bit multiplies in
  // the step loop.  (It points at this wave's first output.)
  const lane3_gf64 *bptr = (const lane3_gf64 *)B + lane_row(lg, NT, nb, L - Wu) + ((int64_t)(4 * og0) << 6);
  const int64_t bjump = (lane_row(lg, NT, item, 0) - lane_row(lg, NT, nb, L)) ;      // added when the warm-up ends
  auto prefetch_group = [&](const lane3_gf64 *row, int g) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bnx[4 * g + q] = (TEHMM_L3_EXP & 8) ? -1.0 - q : row[(4 * g + q) << 6];
  };
  // traceback dwords of this wave's groups: (p0 + t0 + s) * (NT / 4) + og0 + g
  uint32_t *tbptr = tb32 ? tb32 + (p0 + t0) * (NT / 4) + og0 : nullptr;

  // One position for this wave's outputs; the new values go to the LDS buffer `nxt`, the tie ballot to tieb.
  auto step = [&](int s, double rt, int nxt, bool &tie_out) {
    const bool official = s >= 0;
    double tmin = 1.0;
    double wl = INFINITY;
    const_f64 *tp = tab0;
    asm volatile("" : "+s"(tp));
    double t[2][BLK];
#pragma unroll
    for (int i = 0; i < BLK; ++i) t[0][i] = tp[i];
    double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
    double d0[4] = {0.0, 0.0, 0.0, 0.0}, zq1[4] = {0.0, 0.0, 0.0, 0.0};
    const bool rgt = RATIO && rt > 1.0;
    const double rm1 = rt - 1.0;
    constexpr int NB = GW * GB;
#pragma clang loop unroll(full)
    for (int b = 0; b < NB; ++b) {
      const int g = b / GB, f0 = RATIO ? ((b % GB) - 1) * 4 : (b % GB) * 4;
      // (the last wave may own fewer than GW groups when NW does not divide G: it skips the tail -- a uniform branch;
      //  the block prefetch of its last group reads one block past its range: the table buffer is padded for it)
      if (!EVEN && g >= ng) continue;
      const double *tc = t[b & 1];
      double *tn = t[(b + 1) & 1];
      if (RATIO && f0 < 0) {
        const_f64 *tq = tp;
        asm volatile("" : "+s"(tq) : "v"(x0));
        if (b + 1 < NB) {
#pragma unroll
          for (int i = 0; i < BLK; ++i) tn[i] = (TEHMM_L3_EXP & 2) ? tc[i] : tq[(b + 1) * BLK + i];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double z0 = tc[q] * rt;
          const double q0 = (z0 + M) - M;
          bad = bad | !(fabs(z0) < zlim);
          tmin = fmin(tmin, fabs(fabs(z0 - q0) - half_u));
          const double z1 = tc[q] * rm1;
          const double q1r = (z1 + M) - M;
          tmin = fmin(tmin, rgt ? fabs(fabs(z1 - q1r) - half_u) : 1.0);
          const double q1 = rgt ? q1r : 0.0;
          zq1[q] = q1;
          d0[q] = (q0 - q1) - ((og0 + g == 0 && q == 0) ? tc[4] : 0.0);
        }
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      if (f0 == 0) {
        x0 = W[0] + tc[0];
        if (RATIO) x0 += 64.0 * d0[0];
      } else {
        x0 = fmax(x0, W[f0] + tc[0]);
      }
      const_f64 *tq = tp;
      asm volatile("" : "+s"(tq) : "v"(x0));
      if (b + 1 < NB) {
#pragma unroll
        for (int i = 0; i < BLK; ++i) tn[i] = (TEHMM_L3_EXP & 2) ? tc[i] : tq[(b + 1) * BLK + i];
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
      // ---- the four outputs of group og0 + g are complete
      const double xs[4] = {x0, x1, x2, x3};
      double bc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bc[q] = bnx[4 * g + q];
      constexpr bool mine = true;                                   // (tail groups never get here)
      uint32_t pw = 0;
      double wn[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double bo = bc[q];
        if (g == 0 && q == 0) bad = bad | ((og0 == 0) & (bo != bo));
        const double bq = (bo + M) - M;
        tmin = fmin(tmin, mine ? fabs(fabs(bo - bq) - half_u) : 1.0);
        const double y = xs[q] + CM;
        const unsigned ylo = (unsigned)__double2loint(y);
        const double m6 = __hiloint2double(__double2hiint(y), (int)(ylo & ~63u)) - CM;
        wn[q] = m6 + 64.0 * (RATIO ? bq + zq1[q] : bq);
        pw |= (~ylo & 63u) << (8 * q);
      }
      if (!RATIO && soft) {
        // (rare) tied outputs of this group: the max again with every candidate rounded on its own parity.  Candidate
        // Y = W[f] + tab[f][o] = 64 (X - base) + index bits; bit 6 of Y / u is the parity of (X - base) / u; a candidate
        // of odd parity takes the OTHER neighbour of b: bq' = bq + 2 (b - bq), i.e. 128 (b - bq) more in this frame.
        double tg = 1.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) tg = fmin(tg, fabs(fabs(bc[q] - ((bc[q] + M) - M)) - half_u));
        if (__any(tg == 0.0)) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double bo = bc[q];
            const double bq = (bo + M) - M;
            const bool tq = fabs(fabs(bo - bq) - half_u) == 0.0;
            if (__any(tq)) {
              if (tq) {
                const double dq = 128.0 * (bo - bq);
                const_f64 *tcol = tab0 + (g * NT) * 4 + q;             // tab[f][o] of this output, f = 0 .. NT - 1
                double xm = -INFINITY;
#pragma unroll
                for (int f = 0; f < NT; ++f) {             // (FULLY unrolled: W[] must stay in registers)
                  const double Y = W[f] + tcol[f * 4];
                  const unsigned lo = (unsigned)__double2loint(Y + CM);
                  const bool odd = (((lo >> 6) & 1u) ^ (unsigned)pbase) != 0u;
                  xm = fmax(xm, odd ? Y + dq : Y);
                }
                const double y = xm + CM;
                const unsigned ylo = (unsigned)__double2loint(y);
                const double m6 = __hiloint2double(__double2hiint(y), (int)(ylo & ~63u)) - CM;
                wn[q] = m6 + 64.0 * bq;
                pw = (pw & ~(0xffu << (8 * q))) | ((~ylo & 63u) << (8 * q));
              }
            }
          }
        }
      }
      if (mine) {
        const int og = og0 + g;
        if (run && official && tb32 && !(TEHMM_L3_EXP & 32)) ((__attribute__((address_space(1))) uint32_t *)tbptr)[g] = pw;
        // lowest live value of the new vector (pads, which sit in the last two groups, hold -inf)
        if (og >= G - 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) wl = fmin(wl, 4 * og + q < N ? wn[q] : INFINITY);
        } else {
          wl = fmin(wl, fmin(fmin(wn[0], wn[1]), fmin(wn[2], wn[3])));
        }
        lane3_d2 *dst = Wl + ((size_t)nxt * (NT / 2) + 2 * og) * 64 + lane;
        dst[0] = (lane3_d2){wn[0], wn[1]};
        dst[64] = (lane3_d2){wn[2], wn[3]};
      }
      asm volatile("" : "+v"(tmin));
      __builtin_amdgcn_sched_barrier(0);
      L3ST(2 + (g < 3 ? g : 3));
    }
    tie_out = tmin == 0.0;
    wl_prev = wl;
  };

  auto restart = [&]() {
#pragma unroll
    for (int j = 0; j < NT; ++j) W[j] = j < N ? 0.0 : -INFINITY;
    base = 0.0;
    pbase = 0;
    pmin = INFINITY;
  };
  // minimum of a per-wave quantity over the workgroup's waves (rare paths; every wave calls it)
  auto wg_min = [&](double v) {
    xch[w * 64 + lane] = v;
    __syncthreads();
    double r = xch[lane];
#pragma unroll
    for (int k = 1; k < NW; ++k) r = fmin(r, xch[k * 64 + lane]);
    __syncthreads();
    return r;
  };

  // ---- finish step s - 1 (k = its successor's step counter): who met a tie?  everybody else takes the new vector
  //      from LDS; deferred bookkeeping of the step; every 16th position the row, every 32nd the re-basing; pieces
  bool pending = false;
  auto finish = [&](int s, int k) {
    L3ST(6);
    if (!(TEHMM_L3_EXP & 1)) __syncthreads();
    L3ST(0);
    const unsigned long long *tbq = tieb + ((k - 1) & 1) * 8;
    unsigned long long tmv = tbq[0];
#pragma unroll
    for (int q = 1; q < NW; ++q) tmv |= tbq[q];
    // (the same value in every lane of every wave: make it scalar so that the branches on it are uniform)
    const unsigned long long tm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(tmv >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)tmv);
    const bool tie_prev = (tm >> lane) & 1ull;
    if (!tie_prev && !(TEHMM_L3_EXP & 4)) {
      const lane3_d2 *src = Wl + (size_t)(k & 1) * (NT / 2) * 64 + lane;
#pragma unroll
      for (int jp = 0; jp < NT / 2; ++jp) {
        const lane3_d2 v = src[jp * 64];
        W[2 * jp] = v.x;
        W[2 * jp + 1] = v.y;
      }
      pmin = fmin(pmin, wl_prev * 0.015625 + base);
      // W = 64 (value - base) + index bits must stay above -(2^e - 192 u) (the range of the arg-max extraction)
      bad = bad | (wl_prev <= -(ldexp(1.0, e) - ldexp(1.5, e - 45)));
    }
    pending = tie_prev;
    const int sp = s - 1;
    auto read_vector = [&]() {
      const lane3_d2 *src = Wl + (size_t)(k & 1) * (NT / 2) * 64 + lane;
#pragma unroll
      for (int jp = 0; jp < NT / 2; ++jp) {
        const lane3_d2 v = src[jp * 64];
        W[2 * jp] = v.x;
        W[2 * jp + 1] = v.y;
      }
    };
    auto record_tie = [&](double pm) {
      if (run && s - 1 >= 0) {
        if (nt < TEHMM_LANE_MAXTI && w == 0) {
          const VitItems *vi = VI();
          LANE3_GI(lane3_gload(&vi->ties))[item * TEHMM_LANE_MAXTI + nt] = s - 1;
          lane3_gf64 *tr = LANE3_G(lane3_gload(&vi->tierows)) + (item * TEHMM_LANE_MAXTI + nt) * NT;
#pragma unroll
          for (int j = 0; j < NT; ++j) tr[j] = W[j] * 0.015625 + base;
          LANE3_G(lane3_gload(&vi->piecemin))[item * (TEHMM_LANE_MAXTI + 1) + nt] = pm;
        }
        ++nt;
      }
    };
    // soft ties: the tie and the row BEFORE it (still in the registers of the lanes that met it) are recorded, then
    // those lanes take the new vector too -- computed through the tie on the even-delta hypothesis -- and a new piece
    // of the SAME frame begins with it.  (Before the re-basing below: the new vector is relative to the old base.)
    if (soft && tm != 0ull) {                          // (uniform over the workgroup)
      const double pm = wg_min(pmin);
      if (pending) {
        record_tie(pm);
        read_vector();
        pmin = wl_prev * 0.015625 + base;
        bad = bad | (wl_prev <= -(ldexp(1.0, e) - ldexp(1.5, e - 45)));
      }
      pending = false;
    }
    if (((sp + Wu) & (TEHMM_VROW - 1)) == TEHMM_VROW - 1) {
      if (((sp + Wu) & 31) == 31) {
        double mx = W[0];
#pragma unroll
        for (int j = 1; j < NT; ++j) mx = fmax(mx, W[j]);
        if (mx > -INFINITY) {
#pragma unroll
          for (int j = 0; j < NT; ++j) W[j] -= mx;
          base += mx * 0.015625;
          // parity of base / u: mx is a multiple of 64 u, bit 6 of (mx + CM) / u is the parity of mx / (64 u)
          pbase ^= (int)(((unsigned)__double2loint(mx + CM) >> 6) & 1u);
        } else {
          bad = true;
        }
      }
      const int rec = (sp + Wu) / TEHMM_VROW;
      if (sp >= 0 && run && (rec % NW) == w) {
        lane3_gf64 *row = LANE3_G(lane3_gload(&VC()->rows)) + ((int64_t)c * (CS / TEHMM_VROW) + (t0 - ct0 + sp) / TEHMM_VROW) * NT;
#pragma unroll
        for (int j = 0; j < NT; ++j) row[j] = W[j] * 0.015625 + base;
      }
    }
    // hard ties (segment ratios, or soft ties switched off): close the piece with the row of the position before the
    // tie, restart from zeros behind it
    if (!soft && tm != 0ull) {                         // (uniform over the workgroup)
      const double pm = wg_min(pmin);
      if (pending) {
        record_tie(pm);
        restart();
      }
      pending = false;
    }
  };

  const int s_first = -Wu;
  for (int s = s_first; s < L; ++s) {
    const int k = s - s_first;                          // step counter: the LDS buffers alternate on its parity
    // this step's emission values first: they have the barrier, the vector read and a group of max-plus work to arrive
    // (nothing of them is carried around the loop: loop-carried load results cost a vmcnt(0) and 12 copies per step)
    {
      if (s == 0) bptr += bjump;
#pragma unroll
      for (int g = 0; g < GW; ++g)
        if (EVEN || g < ng) prefetch_group(bptr, g);
      if (!(TEHMM_L3_EXP & 16)) bptr += NT * 64;
    }
    double rt = 1.0;
    if (RATIO) rt = run ? (double)((const lane3_gf64 *)ratios)[p0 + t0 + s] : 1.0;
    if (k > 0) finish(s, k);
    if (s == 0) {
      if (run && w == 0) {
        lane3_gf64 *dst = LANE3_G(lane3_gload(&VI()->pre));
        const int64_t po = (((item >> 6) * NT) << 6) + (item & 63);
#pragma unroll
        for (int j = 0; j < NT; ++j) dst[po + ((int64_t)j << 6)] = W[j] * 0.015625 + base;
      }
      pmin = INFINITY;
    }
    bool tie = false;
    L3ST(1);
    step(s, rt, (k + 1) & 1, tie);
    const unsigned long long bal = __ballot(tie);
    if (lane == 0) tieb[(k & 1) * 8 + w] = bal;
    if (s >= 0 && tbptr) tbptr += NT / 4;
  }
  finish(L, L - s_first);
#ifdef TEHMM_STAMPS
  if (lane == 0 && blockIdx.x < 1024)
    {
      l3st[7] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
      for (int i = 0; i < 8; ++i) g_stamps[(blockIdx.x * 4 + w) * 8 + i] = l3st[i];
    }
#endif
  // ---- epilogue (W = the vector of position L - 1, pieces closed)
  {
    const double pm = wg_min(pmin);
    const bool anybad = bad || nt > TEHMM_LANE_MAXTI;
    if (run) {
      const VitItems *vi = VI();
      if (w == 0) {
        const int64_t po = (((item >> 6) * NT) << 6) + (item & 63);
        lane3_gf64 *dst = LANE3_G(lane3_gload(&vi->end));
#pragma unroll
        for (int j = 0; j < NT; ++j) dst[po + ((int64_t)j << 6)] = W[j] * 0.015625 + base;
        LANE3_GI(lane3_gload(&vi->ntie))[item] = nt;
        LANE3_G(lane3_gload(&vi->piecemin))[item * (TEHMM_LANE_MAXTI + 1) + min(nt, TEHMM_LANE_MAXTI)] = pm;
      }
      if (anybad) LANE3_GI(lane3_gload(&vi->bad))[item] = 1;                      // (zeroed before the launch; any wave may raise it)
    }
  }
}

}  // namespace tehmm

namespace tehmm {

// ==========================================================================================
// Emission rows + P0 (k_emis_gain_lane) with the STATES of a position split over the NW waves of a unit (round 4).
//
// k_emis_gain_lane keeps a whole fp64 emission row (72 registers), the packed-float vector and its successor per
// lane: 236 registers, two waves per SIMD, and a dependent gather (observation word -> table row -> 18 16-byte
// reads per track) whose latency those two waves cannot cover: the vector pipe is active 36 % of the time.  Here a
// unit = 64 items x NW waves; wave w gathers and sums only ITS states of the row (a third of every table row), runs
// the packed-float max-plus step for ITS outputs, and the new float vector (plus the waves' row maxima, which
// decide whether anything can emit the row at all) goes round through LDS, one workgroup barrier per position.
// ~130 registers -> three waves per SIMD with a third of the gather each.  Two units share one workgroup and one
// LDS copy of the small tracks' table rows.  Outputs as k_emis_gain_lane: B (fp64 log rows, item-interleaved,
// NaN rows where no state can emit) and gain[item].
// ==========================================================================================
template <int NT, int NW, int NU>
struct Emis3Geom {
  using L3 = Lane3Geom<NT, NW>;
  static constexpr size_t UNIT_BYTES = (size_t)2 * (NT / 2) * 64 * sizeof(float) * 2 + (size_t)2 * NW * 64 * sizeof(double);
  static size_t lds_bytes(int lds_rows) { return (size_t)lds_rows * NT * sizeof(double) + NU * UNIT_BYTES; }
};

// the states [j0, j0 + NS) of the emission log-likelihood row of one position (operation order per state as in
// _emission.pyx:65-72; emis_rows of tehmm_coop.hip.h restricted to a slice; j0 even)
template <int NT, int NS>
__device__ __forceinline__ void emis_rows_slice(const EmisTab &e, const double *ltab, int64_t gpos, int j0, double (&x)[NS]) {
  const uint32_t *row = e.obs32 + gpos * e.KPW;
#pragma unroll
  for (int j = 0; j < NS; ++j) x[j] = 0.0;
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
        if (lb >= 0) {
          lds_cd2 *tr = (lds_cd2 *)(size_t)((unsigned)(size_t)(__attribute__((address_space(3))) const double *)ltab +
                                            (unsigned)(((inr ? lb + sym : e.lds_zero) * NT + j0) * 8));
#pragma unroll
          for (int jj = 0; jj < NS / 2; ++jj) {
            const d2v v = tr[jj];
            x[2 * jj] += v.x;
            x[2 * jj + 1] += v.y;
          }
        } else {
          const double2 *tr = (const double2 *)(e.tab + (int64_t)(inr ? e.rowbase[k] + sym : e.zero_row) * NT + j0);
#pragma unroll
          for (int jj = 0; jj < NS / 2; ++jj) {
            const double2 v = tr[jj];
            x[2 * jj] += v.x;
            x[2 * jj + 1] += v.y;
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NS; ++j) x[j] *= e.normalize;
  if (e.ratios) {
    const double r = e.ratios[gpos];
#pragma unroll
    for (int j = 0; j < NS; ++j) x[j] *= r;
  }
}

template <int NT, int NW, int NU, bool RATIO>
__global__ __launch_bounds__(64 * NW * NU)
void k_emis_gain_lane3(IntervalTab iv, EmisTab em, LaneGeom lg, int N, int CS, int Wu, const float *__restrict__ tabf,
                       double *B, double *gain, const double *__restrict__ ratios) {
  using GEO = Lane3Geom<NT, NW>;
  using EG = Emis3Geom<NT, NW, NU>;
  constexpr int G = GEO::G, GW = GEO::GW, OW = 4 * GW, PW = 2 * GW;
  constexpr bool EVEN = G % NW == 0;
  extern __shared__ double e3_lds[];
  double *ltab = e3_lds;
  stage_emis_table(em, ltab, NT);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int u = wave / NW, w = wave - u * NW;
  char *ubase = (char *)(ltab + em.lds_rows * NT) + (size_t)u * EG::UNIT_BYTES;
  lane_f2 *Wx = (lane_f2 *)ubase;                                        // [2][NT / 2][64]
  double *rmx = (double *)(ubase + (size_t)2 * (NT / 2) * 64 * sizeof(lane_f2));   // [2][NW][64]
  const int g = blockIdx.x * NU + u;
  const bool glive = g < lg.n_groups;                                   // (uniform per unit; every wave keeps the barriers)
  const int L = lg.L;
  const int64_t item = (int64_t)(glive ? g : 0) * 64 + lane;
  const bool valid = glive && item < lg.n_items;
  const int id = valid ? lg.item_iv[item] : 0;
  const int64_t t0 = valid ? lg.item_t0[item] : 0;
  const int64_t T = iv.len[id], p0 = iv.pos0[id];
  const int len = valid ? (int)min((int64_t)L, T - t0) : 0;
  const int64_t ct0 = (t0 / CS) * CS;
  const bool run = valid && ct0 > 0 && ct0 + CS <= T;                   // P0 runs on full chunks but an interval's first
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  const int og0 = w * GW;                                               // this wave's output groups / states 4 og0 ..
  const int ng = min(GW, G - og0);
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
  double xp[OW];                                                        // this wave's states of the previous position
#pragma unroll
  for (int j = 0; j < OW; ++j) xp[j] = 0.0;
  bool act_prev = false;
  // finish position sp (step counter k = its successor's): the new vector and the waves' row maxima from LDS; the row
  // goes to B now that it is known whether ANY state can emit it
  auto finish = [&](int sp, int k) {
    __syncthreads();
    const lane_f2 *src = Wx + (size_t)(k & 1) * (NT / 2) * 64 + lane;
#pragma unroll
    for (int jp = 0; jp < NT / 2; ++jp) W[jp] = src[jp * 64];
    const double *rm = rmx + (size_t)((k - 1) & 1) * NW * 64 + lane;
    double m = rm[0];
#pragma unroll
    for (int q = 1; q < NW; ++q) m = fmax(m, rm[q * 64]);
    const bool good = m > -1e20;
    if (B && sp >= 0 && act_prev) {
      lane3_gf64 *dst = (lane3_gf64 *)B + ((((int64_t)g * L + sp) * NT + 4 * og0) << 6) + lane;
#pragma unroll
      for (int j = 0; j < OW; ++j)
        if (EVEN || j < 4 * ng) dst[(int64_t)j << 6] = good ? xp[j] : qnan;
    }
    bad = bad | (act_prev && !good);
  };
  int k = 0;
  for (int s = -Wu; s < L; ++s, ++k) {
    if (k > 0) finish(s - 1, k);
    if (s == 0) g0 = vec_max();
    const bool act = s >= 0 ? s < len : run;
    const int64_t gpos = p0 + t0 + (act ? s : 0);
    double x[OW];
    emis_rows_slice<NT, OW>(em, ltab, gpos, 4 * og0, x);
    // (the last wave of an uneven split reads past the row's NT states -- the next row, the table's slack or the LDS
    //  behind the staged rows: readable, and what lies outside its own states is not used)
    double pm = -INFINITY;
#pragma unroll
    for (int j = 0; j < OW; ++j) pm = fmax(pm, (4 * og0 + j < N && (EVEN || j < 4 * ng)) ? x[j] : -INFINITY);
    // ---- P0 step of this wave's outputs on the row rounded to float (see k_vit_gain_lane)
    int z = 0;
    asm volatile("" : "+s"(z));
    const_f2 *tp = (const_f2 *)(size_t)tabf + z;
    float rf = 1.f, rm1 = 0.f;
    if (RATIO) {
      rf = (float)ratios[gpos];
      rm1 = rf > 1.f ? rf - 1.f : 0.f;
    }
    lane_f2 *dstw = Wx + ((size_t)((k + 1) & 1) * (NT / 2) + 2 * og0) * 64 + lane;
#pragma unroll
    for (int op = 0; op < PW; ++op) {
      if (!EVEN && op >= 2 * ng) continue;
      const int opg = 2 * og0 + op;                                     // (wave-uniform)
      lane_f2 acc = (lane_f2){W[0].x, W[0].x} + tp[opg * NT];
      if (RATIO) {
        const lane_f2 ltd = tp[NT * NT / 2 + opg];
        acc = acc + ltd * (lane_f2){rf - rm1, rf - rm1};
        if (op == 0 && og0 == 0) acc.x -= ((const float *)(size_t)tabf)[z + NT * NT + NT];
      }
#pragma unroll
      for (int f = 1; f < NT; ++f) {
        const float wf = (f & 1) ? W[f >> 1].y : W[f >> 1].x;
        acc = __builtin_elementwise_max(acc, (lane_f2){wf, wf} + tp[opg * NT + f]);
      }
      lane_f2 wn = acc + (lane_f2){(float)x[2 * op], (float)x[2 * op + 1]};
      if (RATIO) wn = wn + tp[NT * NT / 2 + opg] * (lane_f2){rm1, rm1};
      dstw[op * 64] = wn;
    }
    rmx[(size_t)(k & 1) * NW * 64 + w * 64 + lane] = pm;
#pragma unroll
    for (int j = 0; j < OW; ++j) xp[j] = x[j];
    act_prev = act;
  }
  finish(L - 1, k);
  if (run && w == 0) ((lane3_gf64 *)gain)[item] = (bad || len < L) ? qnan : (double)vec_max() - (double)g0;
}

}  // namespace tehmm
