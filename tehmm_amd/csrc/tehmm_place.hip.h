// tehmm_place.hip.h -- binade placement of the chunk-parallel exact Viterbi ON THE DEVICE (round 4).
//
// Between the gain pass P0 and the quantised pass P2 (tehmm_spec.hip.h, tehmm_lane.hip.h) every chunk has to be
// placed in its binade and the work list of P2 -- one unit per (group of 64 items, binade), the items of mixed
// groups pooled per binade -- has to be built.  Rounds 1..3 did that on the host: item gains D2H, a stream
// synchronisation, prefix sums and list building, five H2D copies and four memsets.  Alone that was ~2 ms of an
// idle GPU between two throughput kernels; with the posterior pipeline running on the other stream every one of
// those nine small transfers queued behind running kernels for a wave slot: 8.8 ms from the end of P0 to the start
// of P2 (rocprofv3 kernel trace, 100 Mb).  Here four small kernels do the same work in stream order -- no host
// round trip inside tehmm_eval_batch -- and the quantised tables of ALL binades a score can reach are built once per
// model version (tehmm_hip.hip: ensure_qtabs) instead of per call.
//
// The rules are those of spec_assign_binades / chunk_gains (tehmm_hip.hip), restated: which chunk is speculated in
// which binade only decides what the exact chain may adopt after verifying it bit for bit, never a result.
#pragma once
#include "tehmm_lane.hip.h"

namespace tehmm {

#define TEHMM_PLACE_EMAX 46                              // binades TEHMM_SPEC_MIN_E .. TEHMM_PLACE_EMAX have a table
#define TEHMM_PLACE_NE (TEHMM_PLACE_EMAX - TEHMM_SPEC_MIN_E + 1)

struct PlaceCounts {                                     // one per batch, zeroed before every placement
  int units[TEHMM_PLACE_NE];                             // groups whose speculated items share binade e
  int pooled[TEHMM_PLACE_NE];                            // items of mixed groups, per binade
  int unit_base[TEHMM_PLACE_NE];                         // first work unit of binade e (uniform groups, then pooled slots)
  int slot_base[TEHMM_PLACE_NE];                         // first 64-item slot of binade e in wk_items
  int cur_units[TEHMM_PLACE_NE];
  int cur_pooled[TEHMM_PLACE_NE];
  int n_work;
};

__device__ __forceinline__ double place_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// inclusive prefix sum over the wave's lanes (lanes beyond n contribute their own v; the caller zeroes them)
__device__ __forceinline__ double place_wave_scan(double v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double u = __shfl_up(v, o);
    if (lane >= o) v += u;
  }
  return v;
}

// One wave per interval: item gains -> chunk gains (a chunk the lanes did not run, i.e. the interval's first and its
// ragged tail, gets the interval's mean gain per position; a full chunk whose item failed stays NaN and ends the
// placement of everything behind it), prefix sums, binade per chunk.  qok[e - MIN_E] = 0: the quantised table of that
// binade has an entry exactly between two grid points (unusable).
__global__ __launch_bounds__(64) void k_vit_place_chunks(IntervalTab iv, const int64_t *__restrict__ cfirst,
                                                         const int64_t *__restrict__ ct0, const int64_t *__restrict__ ifirst,
                                                         int n_iv, int CS, int LS, const double *__restrict__ igain,
                                                         const int *__restrict__ qok, double rel, int cross, int *e_out,
                                                         double *cgain_out) {
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i >= n_iv) return;
  const int64_t c0 = cfirst[i], c1 = cfirst[i + 1], T = iv.len[i];
  const int SUB = CS / LS;
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  // pass 1: chunk gains of the chunks the lanes ran, and their mean
  double sum = 0.0, cnt = 0.0;
  for (int64_t cb = c0; cb < c1; cb += 64) {
    const int64_t c = cb + lane;
    if (c < c1) {
      const int64_t t0 = ct0[c];
      const bool full = t0 + CS <= T;
      double g = qnan;
      if (full && c != c0) {
        const int64_t it0 = ifirst[i] + t0 / LS;
        g = 0.0;
        for (int k = 0; k < SUB; ++k) g += igain[it0 + k];
      }
      cgain_out[c] = g;
      if (g == g) { sum += g; cnt += 1.0; }
    }
  }
  sum = place_wave_sum(sum);
  cnt = place_wave_sum(cnt);
  const double mean = cnt > 0.0 ? sum / cnt : qnan;
  // pass 2: fill in, prefix sums, binades
  double carry = 0.0;
  for (int64_t cb = c0; cb < c1; cb += 64) {
    const int64_t c = cb + lane;
    const bool in = c < c1;
    double g = 0.0;
    bool full = false;
    if (in) {
      const int64_t t0 = ct0[c];
      full = t0 + CS <= T;
      g = cgain_out[c];
      if (!(g == g)) {
        const int64_t clen = min((int64_t)CS, T - t0);
        g = (c == c0 || clen < CS) ? mean * (double)clen / (double)CS : qnan;
        cgain_out[c] = g;
      }
    }
    const double incl = place_wave_scan(in ? g : 0.0, lane);
    const double ve = carry + incl, vs = ve - g;         // (NaN once a NaN gain has entered: nothing behind it is placed)
    carry += __shfl(incl, 63);
    if (!in) continue;
    int be = TEHMM_SPEC_NONE;
    if (c != c0 && full && g == g && g < 0.0) {
      const double margin = 512.0 + rel * fabs(ve);
      const double lo = fabs(vs) - margin, hi = fabs(ve) + margin;
      if (lo > 0.0 && hi < INFINITY) {
        int ex = 0, exh = 0;
        (void)frexp(lo, &ex);
        (void)frexp(hi, &exh);
        if (exh == ex || (cross && exh == ex + 1)) {
          const int b = exh - 1;
          if (b >= TEHMM_SPEC_MIN_E && b <= TEHMM_PLACE_EMAX && qok[b - TEHMM_SPEC_MIN_E]) be = b;
        }
      }
    }
    e_out[c] = be;
  }
}

// the binade of an item's chunk
__device__ __forceinline__ int place_item_e(const LaneGeom &lg, const int64_t *cfirst, const int *e, int CS, int64_t item) {
  return e[cfirst[lg.item_iv[item]] + lg.item_t0[item] / CS];
}

// Work list, pass A: one thread per group of 64 items: uniform (one unit) or mixed (its items are pooled per binade)
__global__ __launch_bounds__(256) void k_vit_place_count(LaneGeom lg, const int64_t *__restrict__ cfirst,
                                                         const int *__restrict__ e, int CS, PlaceCounts *pc, int *gclass) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= lg.n_groups) return;
  const int nl = min(64, lg.n_items - g * 64);
  int e1 = TEHMM_SPEC_NONE;
  bool mixed = false;
  for (int ln = 0; ln < nl; ++ln) {
    const int ei = place_item_e(lg, cfirst, e, CS, (int64_t)g * 64 + ln);
    if (ei == TEHMM_SPEC_NONE) continue;
    if (e1 == TEHMM_SPEC_NONE) e1 = ei;
    else if (ei != e1) mixed = true;
  }
  if (e1 == TEHMM_SPEC_NONE) { gclass[g] = TEHMM_SPEC_NONE; return; }
  if (!mixed) {
    gclass[g] = e1;
    atomicAdd(&pc->units[e1 - TEHMM_SPEC_MIN_E], 1);
    return;
  }
  gclass[g] = TEHMM_SPEC_NONE + 1;                       // mixed
  for (int ln = 0; ln < nl; ++ln) {
    const int ei = place_item_e(lg, cfirst, e, CS, (int64_t)g * 64 + ln);
    if (ei != TEHMM_SPEC_NONE) atomicAdd(&pc->pooled[ei - TEHMM_SPEC_MIN_E], 1);
  }
}

// offsets: the units of one binade next to each other (they share one quantised table in the scalar cache), binades
// ascending; within a binade the uniform groups first, then the pooled slots
__global__ void k_vit_place_scan(PlaceCounts *pc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int run = 0, slots = 0;
  for (int k = 0; k < TEHMM_PLACE_NE; ++k) {
    pc->unit_base[k] = run;
    pc->slot_base[k] = slots;
    const int ns = (pc->pooled[k] + 63) / 64;
    run += pc->units[k] + ns;
    slots += ns;
  }
  pc->n_work = run;
}

// pass B: fill wk_g / wk_e / wk_items (wk_items pre-filled with -1)
__global__ __launch_bounds__(256) void k_vit_place_fill(LaneGeom lg, const int64_t *__restrict__ cfirst,
                                                        const int *__restrict__ e, int CS, PlaceCounts *pc,
                                                        const int *__restrict__ gclass, int *wk_g, int *wk_e, int *wk_items) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= lg.n_groups) return;
  const int cls = gclass[g];
  if (cls == TEHMM_SPEC_NONE) return;
  if (cls != TEHMM_SPEC_NONE + 1) {
    const int k = cls - TEHMM_SPEC_MIN_E;
    const int pos = pc->unit_base[k] + atomicAdd(&pc->cur_units[k], 1);
    wk_g[pos] = g;
    wk_e[pos] = cls;
    return;
  }
  // the items of a mixed group go to their binades' pools RUN BY RUN (consecutive items of one binade): one
  // reservation per run keeps neighbours in neighbouring lanes of a pooled unit, whose row loads then coalesce as far
  // as a pooled unit's can (items scattered one by one cost the quantised pass 9 of 16 ms: every load of such a unit
  // touched 64 cache lines)
  const int nl = min(64, lg.n_items - g * 64);
  int ln = 0;
  while (ln < nl) {
    const int ei = place_item_e(lg, cfirst, e, CS, (int64_t)g * 64 + ln);
    int len = 1;
    while (ln + len < nl && place_item_e(lg, cfirst, e, CS, (int64_t)g * 64 + ln + len) == ei) ++len;
    if (ei != TEHMM_SPEC_NONE) {
      const int k = ei - TEHMM_SPEC_MIN_E;
      const int idx0 = atomicAdd(&pc->cur_pooled[k], len);
      for (int j = 0; j < len; ++j) {
        const int idx = idx0 + j;
        const int slot = pc->slot_base[k] + idx / 64;
        wk_items[(int64_t)slot * 64 + (idx & 63)] = g * 64 + ln + j;
        if ((idx & 63) == 0) {
          const int pos = pc->unit_base[k] + pc->units[k] + idx / 64;
          wk_g[pos] = -(1 + slot);
          wk_e[pos] = ei;
        }
      }
    }
    ln += len;
  }
}

}  // namespace tehmm
