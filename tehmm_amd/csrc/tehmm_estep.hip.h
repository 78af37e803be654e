// tehmm_estep.hip.h -- the Baum-Welch E-step reductions on top of the chunk-parallel fused passes (round 3).
//
// Reference: BaseHMM.fit's per-sequence loop (basehmm.py:504-523) + MultitrackHmm._accumulate_sufficient_statistics
// (hmm.py:545-574) + _hmm._log_sum_lneta (_hmm.pyx:62-117) + _emission.fastAccumulateStats (_emission.pyx:146-190).
//
// Round 2 ran the E-step as ONE sequential forward / backward chain per interval (k_fb_coop, fp64 alpha, beta
// and w rows through HBM: 840 bytes per position) and a lane = state accumulation kernel that issued 36 DPP
// FMAs and up to 70 fp64 global atomics per position.  Here the posterior pipeline of tehmm_fused.hip.h does the
// forward / backward work chunk-parallel (k_fused_fwd, k_fused_bwd<ESTEP>, the exact chains k_fb_fix for what
// the speculation cannot vouch for), leaving three FLOAT rows per position in the alpha' layout (al32_index):
//     alpha'_t                      (k_fused_fwd / forward chain)
//     gamma_t = alpha'_t beta_t / G_t                         -> start, emission histograms
//     wz_t    = w_{t+1} scale_t / G_t                         -> xi_t(i, j) = alpha'_t[i] A[i][j] wz_t[j]
// Rows are WRITTEN (the exact chain overwrites what the speculation left), never accumulated in place, so every
// position is counted once whatever the chains decided; k_estep_reduce then reads each row once per track group:
//   * C[i][j] += sum_t alpha'_t[i] wz_t[j] is a GEMM over the positions (M = N = states, K = positions) and runs
//     on the fp64 matrix cores: v_mfma_f64_16x16x4 contracts four ITEMS of a 16-item tile per instruction, the
//     16 rows / columns of an operand are states.  The alpha' layout keeps, for a state pair-quad, 16 items x
//     (state, state + 4) as consecutive float2, so operand (row tile, 4 items) is one 8-byte load per lane and
//     row tiles are the state sets {8 p + kq} and {8 p + 4 + kq} (p = 4 q .. 4 q + 3, kq = 0 .. 3) -- any
//     bijection rows <-> states is legal as long as the accumulators are written back through it;
//   * the emission histograms obs[k][sym][j] += gamma_t[j]: tracks with FEW symbols are one more product on the
//     matrix cores (one-hot rows x gamma rows, k_estep_hist_mfma); tracks with many symbols (the 250-bin gaussian
//     tracks) are privatised in LDS in fp64 (ds_add_f64, lane = (item, state quarter), k_estep_hist_lds), grouped
//     so that a group's rows fit the 160 KB of a CU; one flush per workgroup instead of per-position global
//     atomics (round 2: 23 of 62 ms).  First version: everything through LDS atomics, 36 of its 40 ms per 50 Mb
//     were ds_add_f64 at ~1 lane per cycle;
//   * start += gamma_0 of every interval.
// Tolerance: the rows are floats (relative 6e-8 each), the sums fp64: statistics agree with the reference to
// ~1e-7 (bar 1e-6, asserted in tests/test_gpu_r3.py).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "tehmm_fused.hip.h"

// the one-hot product's instruction: bf16 16x16x16 (default), fp32 16x16x4 (-DTEHMM_ESTEP_HIST_F32) or fp64 16x16x4
// (-DTEHMM_ESTEP_HIST_F64, the first version)
#if !defined(TEHMM_ESTEP_HIST_F64) && !defined(TEHMM_ESTEP_HIST_F32) && !defined(TEHMM_ESTEP_HIST_BF16)
#define TEHMM_ESTEP_HIST_BF16 1
#endif

namespace tehmm {

// Track partition of the reduction (built on the host, estep_build_groups):
//   * LDS groups: tracks with many symbols, histograms privatised in LDS (ds_add_f64 costs ~55 cycles per wave
//     instruction whatever the addresses -- measured: 2.3 ms per track and 50 Mb --, i.e. ~500 LDS cycles per track,
//     tile and step), at most `cap` rows per group;
//   * row-tile groups: tracks with few symbols go through the matrix cores instead (16 (track, symbol) rows per
//     tile, 12 matrix instructions = 192 CU cycles per row tile, tile and step: cheaper below ~2.6 row tiles per
//     track), at most TEHMM_ESTEP_RTG row tiles per group.
#define TEHMM_ESTEP_MAXRT 64          // row tiles of the one-hot reduction (16 rows each)
struct EstepGroups {
  int n_lds;                           // LDS groups (grid.y of k_estep_hist_lds)
  int first[TEHMM_MAX_TRACKS + 1];     // LDS group g owns the slots [first[g], first[g + 1])
  int rows[TEHMM_MAX_TRACKS];          // LDS group -> rows of its histogram
  int info[TEHMM_MAX_TRACKS];          // slot -> observation column | rows of the track << 7 | first LDS row << 16
  int gbase[TEHMM_MAX_TRACKS];         // slot -> first row of the track in the global statistics table
  int n_rt;                            // row tiles of the one-hot reduction
  int rt_info[TEHMM_ESTEP_MAXRT * 16]; // row -> observation column | symbol << 8, or -1 (padding row)
  int rt_grow[TEHMM_ESTEP_MAXRT * 16]; // row -> row of the global statistics table
};

template <int NT>
struct EstepGeom {
  static constexpr int KS = NT / 4;
  static constexpr int P = al32_pairs(NT);               // float2 per lane and row
  static constexpr int PQ = (P + 3) / 4;                 // pair quads
  // state tiles (q, h): states 8 (4 q + pp) + 4 h + kq; a tile exists when its first state does
  static constexpr int NTILE = 2 * PQ - ((32 * (PQ - 1) + 4 < NT) ? 0 : 1);
  static constexpr int RTG = 8;                          // row tiles per workgroup of the one-hot reduction (<= 2 per wave)
  static __host__ __device__ constexpr int state(int tile, int m) {
    return 8 * (4 * (tile >> 1) + (m >> 2)) + 4 * (tile & 1) + (m & 3);
  }
};

// Reproducible sums (round 4).  Rounds 1..3 flushed every persistent workgroup's accumulators into the statistics with
// floating-point atomics: the order of those additions -- and with it the last bits of every statistic, of the
// M-step's parameters and of the --maxProb / convergence decisions taken on them -- changed from run to run.  Now
// every writer (a wave of k_estep_xi, a workgroup of the two histogram kernels) owns one SLOT of a partial buffer
// and writes its sums there; k_estep_fold_* add the slots in ascending order.  The work of a writer is a fixed
// function of the batch geometry, so the same input gives the same bits.  The LDS histograms of k_estep_hist_lds are
// accumulated by eight waves in whatever order they arrive: they are kept as 64-bit FIXED-POINT integers
// (ds_add_u64 is exact, hence order-free; the scale leaves 62 bits for the largest sum a workgroup can reach).

// What every reduction kernel knows about the 16-item tile a wave works on.
struct EstepTile {
  int ns, nsmax;            // positions of this lane's item / of the longest item of the tile
  int64_t t0;               // first position of this lane's item
  const uint8_t *orow;      // observation row of (this lane's item, position 0)
  int64_t tb2;              // float2 index of (position 0, pair 0) of the tile; one position further = 256 P float2
};
template <int NT>
__device__ __forceinline__ EstepTile estep_tile(const IntervalTab &iv, const LaneGeom &lg, int64_t tile, int i16, int KP,
                                                const uint8_t *obs) {
  EstepTile tc;
  const int64_t item = tile * 16 + i16;
  const bool valid = item < lg.n_items;
  const int id = valid ? lg.item_iv[item] : 0;
  tc.t0 = valid ? lg.item_t0[item] : 0;
  const int64_t T = iv.len[id];
  tc.ns = valid ? (int)min((int64_t)lg.L, T - tc.t0) : 0;
  int nsmax = tc.ns;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) nsmax = max(nsmax, __shfl_xor(nsmax, o));
  tc.nsmax = __builtin_amdgcn_readfirstlane(nsmax);
  tc.orow = obs + (iv.pos0[id] + tc.t0) * KP;
  tc.tb2 = (((tile >> 2) * lg.L) * 4 + (tile & 3)) * (int64_t)(al32_pairs(NT) * 64);
  return tc;
}

// ------------------------------------------------------------------------------------------
// xi:  C[i][j] += sum over positions alpha'_t[i] wz_t[j]  and  start += gamma_0.
// grid = persistent workgroups over the tiles, block = 256 (one tile per wave at a time), no LDS.
// GEMM role of a lane: operand row m = lane & 15 -> (pp = m >> 2, kq = m & 3), contraction index k = lane >> 4
// -> item 4 kk + k of the tile.
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k_estep_xi(IntervalTab iv, LaneGeom lg, int N, const float *__restrict__ al32,
                                                  const float *__restrict__ gam32, const float *__restrict__ wz32,
                                                  double *part /* [4 gridDim.x][NT * NT + NT] */) {
  using G = EstepGeom<NT>;
  constexpr int P = G::P, PQ = G::PQ, NTILE = G::NTILE;
  double sacc[2 * P];                                              // start statistics of this lane's (item, state quarter)
#pragma unroll
  for (int p = 0; p < 2 * P; ++p) sacc[p] = 0.0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kq = lane >> 4, i16 = lane & 15;
  const int g_pp = (lane & 15) >> 2, g_kq = lane & 3, g_k = lane >> 4;
  lane_d4 acc[NTILE][NTILE];
#pragma unroll
  for (int a = 0; a < NTILE; ++a)
#pragma unroll
    for (int b = 0; b < NTILE; ++b) acc[a][b] = (lane_d4){0.0, 0.0, 0.0, 0.0};
  const int64_t n_tiles = (int64_t)lg.n_groups * 4;
  const float2 *gam2 = (const float2 *)gam32, *al2 = (const float2 *)al32, *wz2 = (const float2 *)wz32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const EstepTile tc = estep_tile<NT>(iv, lg, tile, i16, 0, nullptr);
    if (tc.nsmax <= 0) continue;
    // start statistics: gamma_0 of every interval (lane = (item, state quarter), as the passes stored it)
    if (tc.ns > 0 && tc.t0 == 0) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float2 g = gam2[tc.tb2 + kq * 16 + i16 + p * 64];
        sacc[2 * p] += (double)g.x;
        sacc[2 * p + 1] += (double)g.y;
      }
    }
    const int64_t gb2 = tc.tb2 + g_kq * 16 + g_k;                  // + (4 q + pp) * 64 + 4 kk, + 256 P per position
    float2 xa[PQ][4], xw[PQ][4];                                   // operands of the step being requested
    auto request = [&](int s) {
#pragma unroll
      for (int q = 0; q < PQ; ++q)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          // (unconditional loads from a clamped pair: a predicated load costs an exec-mask region and a branch, a select
          //  right behind the load a wait for it)
          const int64_t ix = gb2 + (int64_t)s * (256 * P) + min(4 * q + g_pp, P - 1) * 64 + 4 * kk;
          xa[q][kk] = al2[ix];                      // (raw: a pair beyond the last one is dropped when consumed)
          xw[q][kk] = wz2[ix];
        }
    };
    request(0);
    for (int s = 0; s < tc.nsmax; ++s) {
      double a[4][NTILE], w[4][NTILE];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
          const bool have = 4 * (t >> 1) + g_pp < P;
          a[kk][t] = have ? (double)((t & 1) ? xa[t >> 1][kk].y : xa[t >> 1][kk].x) : 0.0;
          w[kk][t] = have ? (double)((t & 1) ? xw[t >> 1][kk].y : xw[t >> 1][kk].x) : 0.0;
        }
      if (s + 1 < tc.nsmax) request(s + 1);          // (positions beyond an item's end hold zeros: never written)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int ta = 0; ta < NTILE; ++ta)
#pragma unroll
          for (int tw = 0; tw < NTILE; ++tw)
            acc[ta][tw] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk][ta], w[kk][tw], acc[ta][tw], 0, 0, 0);
    }
  }
  // accumulator (lane, register r) of tile pair (ta, tw) is C[state(ta, 4 r + (lane >> 4))][state(tw, lane & 15)]
  double *slot = part + (size_t)(blockIdx.x * 4 + wv) * (NT * NT + NT);
#pragma unroll
  for (int ta = 0; ta < NTILE; ++ta)
#pragma unroll
    for (int tw = 0; tw < NTILE; ++tw)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = G::state(ta, 4 * r + (lane >> 4)), j = G::state(tw, lane & 15);
        const double v = acc[ta][tw][r];
        if (i < NT && j < NT) slot[i * NT + j] = (i < N && j < N) ? v : 0.0;       // (every cell, zeros included)
      }
  // start: the 16 items of the tile lanes, fixed tree; states kq + 8 p (+ 4)
#pragma unroll
  for (int p = 0; p < 2 * P; ++p) {
    double v = sacc[p];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
    const int st = kq + 8 * (p >> 1) + 4 * (p & 1);
    if (i16 == 0 && st < NT) slot[NT * NT + st] = st < N ? v : 0.0;
  }
}

// ------------------------------------------------------------------------------------------
// Emission histograms of the tracks with few symbols, on the matrix cores:
//   stat[(track, symbol)][j] += sum over positions [obs_t[track] == symbol] gamma_t[j]
// = (one-hot rows) x (gamma rows), contracted over the items of the tile exactly like the xi product: operand A
// of row tile rt is 1.0 where the item's symbol of the row's track equals the row's symbol.
// grid (x = persistent workgroups over the tiles, y = group of RTG row tiles), block = 256, no LDS.
// ------------------------------------------------------------------------------------------
// (RT = the row tiles this workgroup role really has: the schedule is branch-free and without padding tiles)
template <int NT, int RT>
__device__ __forceinline__ void estep_hist_mfma_run(const IntervalTab &iv, const LaneGeom &lg, const EstepGroups *__restrict__ egp,
                                                    int N, int KP, const uint8_t *__restrict__ obs,
                                                    const float *__restrict__ gam32, double *part, int rt0) {
  using G = EstepGeom<NT>;
  constexpr int P = G::P, PQ = G::PQ, NTILE = G::NTILE;
  const int lane = threadIdx.x & 63;
  const int i16 = lane & 15;
  const int g_pp = (lane & 15) >> 2, g_kq = lane & 3, g_k = lane >> 4;
  // this lane's row of every row tile: observation column and symbol (padding rows: a symbol no observation
  // byte of a small track can take, so their products are zero)
  int rcol[RT], rsym[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int inf = egp->rt_info[(rt0 + r) * 16 + (lane & 15)];
    rcol[r] = inf < 0 ? 0 : (inf & 255);
    rsym[r] = inf < 0 ? 0x100 : (inf >> 8);
  }
  lane_d4 acc[RT][NTILE];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < NTILE; ++b) acc[a][b] = (lane_d4){0.0, 0.0, 0.0, 0.0};
#ifndef TEHMM_ESTEP_HIST_F64
  // The products are exact in single precision (1.0 x a float gamma) and v_mfma_f32_16x16x4 issues twice as fast
  // as the fp64 form (13.5 against 28 ns per SIMD, profiles/r03_peak_rates.txt): the matrix cores sum eight steps
  // (<= 128 terms <= 1 per cell) in floats, then the partial sums go to the fp64 accumulators -- a relative 1e-7 per
  // partial sum at worst, averaged over thousands of them per cell (the rows themselves are floats, 6e-8 each).
  typedef float estep_f4 __attribute__((ext_vector_type(4)));
  estep_f4 acc32[RT][NTILE];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < NTILE; ++b) acc32[a][b] = (estep_f4){0.f, 0.f, 0.f, 0.f};
  auto flush32 = [&]() {
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int b = 0; b < NTILE; ++b) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[a][b][q] += (double)acc32[a][b][q];
        acc32[a][b] = (estep_f4){0.f, 0.f, 0.f, 0.f};
      }
  };
#endif
  const int64_t n_tiles = (int64_t)lg.n_groups * 4;
  const float2 *gam2 = (const float2 *)gam32;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const EstepTile tc = estep_tile<NT>(iv, lg, tile, i16, KP, obs);
    if (tc.nsmax <= 0) continue;
    // the four items this lane contracts over -- item 4 kk + g_k of the tile for the 16x16x4 instructions (k = lane >> 4),
    // items 4 g_k + kk for the 16x16x16 bf16 one (a lane holds four consecutive k) --: observation rows and lengths
    int64_t oo[4];
    int nsk[4];
    const int64_t myoff = tc.orow - obs;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#ifdef TEHMM_ESTEP_HIST_BF16
      const int src = 4 * g_k + kk;
#else
      const int src = 4 * kk + g_k;                                 // lane `src` holds that item's data
#endif
      nsk[kk] = __shfl(tc.ns, src);
      const unsigned lo = (unsigned)__shfl((int)(unsigned)(unsigned long long)myoff, src);
      const unsigned hi = (unsigned)__shfl((int)(unsigned)((unsigned long long)myoff >> 32), src);
      oo[kk] = (int64_t)(((unsigned long long)hi << 32) | lo);
    }
#ifdef TEHMM_ESTEP_HIST_BF16
    const int64_t gb2 = tc.tb2 + g_kq * 16 + 4 * g_k;              // + kk: the lane's four items are consecutive float2
#else
    const int64_t gb2 = tc.tb2 + g_kq * 16 + g_k;
#endif
    // Two steps are on their way at any time (buffers 0 / 1): the wave is alone on its SIMD (its accumulators take
    // the register file), so nothing else hides the latency of the gamma rows and observation bytes -- with ONE step
    // in flight a tile step took 4.8 us for 1.1 us of matrix instructions
    float2 xg[2][PQ][4];
    int sy[2][RT][4];
    auto request = [&](int s, auto bsel) {
      constexpr int B = decltype(bsel)::value;
      // Every load is UNCONDITIONAL, from a clamped address inside the tile's own rows, and a select drops what lies
      // beyond the item's end (a predicated load compiles to an exec-mask region with a branch: 36 of them per step
      // were what this kernel spent its time on -- 4.8 us per tile step whatever the matrix instructions cost)
#pragma unroll
      for (int q = 0; q < PQ; ++q)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int pc = min(4 * q + g_pp, P - 1);
#ifdef TEHMM_ESTEP_HIST_BF16
          xg[B][q][kk] = gam2[gb2 + (int64_t)s * (256 * P) + pc * 64 + kk];
#else
          xg[B][q][kk] = gam2[gb2 + (int64_t)s * (256 * P) + pc * 64 + 4 * kk];      // (raw: selected when consumed)
#endif
        }
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int sc = max(0, min(s, nsk[kk] - 1));
          sy[B][r][kk] = (int)obs[oo[kk] + (int64_t)sc * KP + rcol[r]];
        }
    };
    auto step = [&](int s, auto bsel) {
      constexpr int B = decltype(bsel)::value;
#ifdef TEHMM_ESTEP_HIST_BF16
      // ONE matrix instruction contracts all 16 items of the tile (v_mfma_f32_16x16x16_bf16: a lane holds four
      // consecutive k).  The one-hot operand is exact in bf16; a float gamma is the exact sum of three bf16 pieces (top
      // 8 bits of the value, of the remainder, the rest), so three instructions per (row tile, state tile) give the same
      // single-precision sums as the 16x16x4 form with a third of the operand building per matrix instruction.
      typedef short estep_s4 __attribute__((ext_vector_type(4)));
      typedef unsigned estep_u2 __attribute__((ext_vector_type(2)));
      estep_s4 a[RT], w[NTILE][3];
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        unsigned h[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) h[kk] = ((s < nsk[kk]) & (sy[B][r][kk] == rsym[r])) ? 0x3F80u : 0u;      // bf16 1.0
        const estep_u2 pk = {h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        a[r] = __builtin_bit_cast(estep_s4, pk);
      }
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        unsigned p1[4], p2[4], p3[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const float g = (t & 1) ? xg[B][t >> 1][kk].y : xg[B][t >> 1][kk].x;
          p1[kk] = __float_as_uint(g) & 0xFFFF0000u;
          const float r1 = g - __uint_as_float(p1[kk]);                 // exact
          p2[kk] = __float_as_uint(r1) & 0xFFFF0000u;
          const float r2 = r1 - __uint_as_float(p2[kk]);                // exact, <= 8 significant bits
          p3[kk] = __float_as_uint(r2);
        }
        // the upper halves of two dwords side by side: v_perm_b32 (bytes 2, 3 of the second operand, then of the first)
        const estep_u2 q1 = {__builtin_amdgcn_perm(p1[1], p1[0], 0x07060302u), __builtin_amdgcn_perm(p1[3], p1[2], 0x07060302u)};
        const estep_u2 q2 = {__builtin_amdgcn_perm(p2[1], p2[0], 0x07060302u), __builtin_amdgcn_perm(p2[3], p2[2], 0x07060302u)};
        const estep_u2 q3 = {__builtin_amdgcn_perm(p3[1], p3[0], 0x07060302u), __builtin_amdgcn_perm(p3[3], p3[2], 0x07060302u)};
        w[t][0] = __builtin_bit_cast(estep_s4, q1);
        w[t][1] = __builtin_bit_cast(estep_s4, q2);
        w[t][2] = __builtin_bit_cast(estep_s4, q3);
      }
      if (s + 2 < tc.nsmax) request(s + 2, bsel);
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int tw = 0; tw < NTILE; ++tw)
            acc32[r][tw] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[r], w[tw][pc], acc32[r][tw], 0, 0, 0);
      if ((s & 7) == 7) flush32();
#elif defined(TEHMM_ESTEP_HIST_F64)
      double a[4][RT], w[4][NTILE];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bool live = s < nsk[kk];
#pragma unroll
        for (int r = 0; r < RT; ++r) a[kk][r] = __hiloint2double((live && sy[B][r][kk] == rsym[r]) ? 0x3ff00000 : 0, 0);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
          const bool have = live && 4 * (t >> 1) + g_pp < P;
          w[kk][t] = have ? (double)((t & 1) ? xg[B][t >> 1][kk].y : xg[B][t >> 1][kk].x) : 0.0;
        }
      }
      if (s + 2 < tc.nsmax) request(s + 2, bsel);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int tw = 0; tw < NTILE; ++tw)
            acc[r][tw] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk][r], w[kk][tw], acc[r][tw], 0, 0, 0);
#else
      float a[4][RT], w[4][NTILE];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bool live = s < nsk[kk];                   // beyond the item's end: no row matches, gamma counts as zero
#pragma unroll
        for (int r = 0; r < RT; ++r) a[kk][r] = (live && sy[B][r][kk] == rsym[r]) ? 1.f : 0.f;
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
          const bool have = live && 4 * (t >> 1) + g_pp < P;
          const float v = (t & 1) ? xg[B][t >> 1][kk].y : xg[B][t >> 1][kk].x;
          w[kk][t] = have ? v : 0.f;
        }
      }
      if (s + 2 < tc.nsmax) request(s + 2, bsel);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int tw = 0; tw < NTILE; ++tw)
            acc32[r][tw] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk][r], w[kk][tw], acc32[r][tw], 0, 0, 0);
      if ((s & 7) == 7) flush32();
#endif
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    request(0, B0{});
    if (tc.nsmax > 1) request(1, B1{});
    for (int s = 0; s < tc.nsmax; s += 2) {
      step(s, B0{});
      if (s + 1 < tc.nsmax) step(s + 1, B1{});
    }
#ifndef TEHMM_ESTEP_HIST_F64
    flush32();
#endif
  }
  // accumulator (lane, register q) of (row tile r, state tile tw): state(tw, lane & 15) and row 4 q + (lane >> 4)
  // for the fp64 instruction, 4 (lane >> 4) + q for the fp32 one (tools/mfma_layout.hip)
#pragma unroll
  for (int r = 0; r < RT; ++r) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#ifdef TEHMM_ESTEP_HIST_F64
      const int row = (rt0 + r) * 16 + 4 * q + (lane >> 4);
#else
      const int row = (rt0 + r) * 16 + 4 * (lane >> 4) + q;
#endif
      // this workgroup's slot: [n_rt * 16 rows][NT]; every cell of the wave's row tiles is written (k_estep_fold_rows
      // maps the rows to the statistics table and drops the padding rows)
      double *srow = part + ((size_t)blockIdx.x * (egp->n_rt * 16) + row) * NT;
#pragma unroll
      for (int tw = 0; tw < NTILE; ++tw) {
        const int j = G::state(tw, lane & 15);
        if (j < NT) srow[j] = j < N ? acc[r][tw][q] : 0.0;
      }
    }
  }
}

#ifndef TEHMM_ESTEP_HW
#define TEHMM_ESTEP_HW 4            // waves per workgroup of k_estep_hist_mfma
#endif
// The waves of a workgroup take the SAME tile and split the group's row tiles among them (<= 2 each): a wave
// then holds 24 accumulator registers per row tile instead of the whole group's 96 doubles, three workgroups fit a CU
// and one wave's matrix instructions run under another's loads and operand building (with all row tiles in one wave --
// one wave per SIMD -- a tile step took 4.8 us for 1.1 us of matrix instructions).  The gamma rows are read by all
// four (L1 / L2 hits after the first).
template <int NT>
__global__ __launch_bounds__(TEHMM_ESTEP_HW * 64) void k_estep_hist_mfma(IntervalTab iv, LaneGeom lg, const EstepGroups *__restrict__ egp,
                                                                       int N, int KP, const uint8_t *__restrict__ obs,
                                                                       const float *__restrict__ gam32, double *part) {
  constexpr int RTG = EstepGeom<NT>::RTG;
  const int rtg0 = blockIdx.y * RTG;
  const int nrt = min(RTG, egp->n_rt - rtg0);
  const int rtw = (nrt + TEHMM_ESTEP_HW - 1) / TEHMM_ESTEP_HW;      // row tiles per wave
  const int wv = threadIdx.x >> 6;
  const int rt0 = rtg0 + wv * rtw;
  const int mine = max(0, min(rtw, nrt - wv * rtw));
  if (mine == 2) estep_hist_mfma_run<NT, 2>(iv, lg, egp, N, KP, obs, gam32, part, rt0);
  else if (mine == 1) estep_hist_mfma_run<NT, 1>(iv, lg, egp, N, KP, obs, gam32, part, rt0);
#if TEHMM_ESTEP_HW < 4
  else if (mine == 3) estep_hist_mfma_run<NT, 3>(iv, lg, egp, N, KP, obs, gam32, part, rt0);
  else if (mine == 4) estep_hist_mfma_run<NT, 4>(iv, lg, egp, N, KP, obs, gam32, part, rt0);
#endif
}

// ------------------------------------------------------------------------------------------
// Emission histograms of the tracks with many symbols: privatised in LDS (fp64), one flush per workgroup.
// grid (x = persistent workgroups over the tiles, y = LDS group), block = 512 (8 waves, one tile each).
// LDS: hist [rows][NT] doubles | info [slots] ints.  Lane = (item = lane & 15, state quarter kq = lane >> 4).
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(512) void k_estep_hist_lds(IntervalTab iv, LaneGeom lg, const EstepGroups *__restrict__ egp,
                                                        int N, int KP, const uint8_t *__restrict__ obs,
                                                        const float *__restrict__ gam32, double *part, int shift) {
  using G = EstepGeom<NT>;
  constexpr int KS = G::KS, P = G::P;
  extern __shared__ double estep_lds[];
  const int grp = blockIdx.y;
  const int s0 = egp->first[grp], nslot = egp->first[grp + 1] - s0, rows = egp->rows[grp];
  // fixed point: value * 2^shift as a 64-bit integer (gamma <= 1; 2^(62 - shift) bounds what one workgroup can add up)
  unsigned long long *hist = (unsigned long long *)estep_lds;
  const double fscale = ldexp(1.0, shift), finv = ldexp(1.0, -shift);
  int *tinfo = (int *)(hist + (size_t)rows * NT);
  for (int i = threadIdx.x; i < rows * NT; i += blockDim.x) hist[i] = 0ull;
  for (int i = threadIdx.x; i < nslot; i += blockDim.x) tinfo[i] = egp->info[s0 + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kq = lane >> 4, i16 = lane & 15;
  const int64_t n_tiles = (int64_t)lg.n_groups * 4;
  const float2 *gam2 = (const float2 *)gam32;
  for (int64_t tile = (int64_t)blockIdx.x * 8 + wv; tile < n_tiles; tile += (int64_t)gridDim.x * 8) {
    const EstepTile tc = estep_tile<NT>(iv, lg, tile, i16, KP, obs);
    if (tc.nsmax <= 0) continue;
    const int64_t hb2 = tc.tb2 + kq * 16 + i16;                                  // + p * 64
    float2 gn[P];
    uint32_t on[4];
    auto request = [&](int s) {
      const int sc = max(0, min(s, tc.ns - 1));             // unconditional loads from inside the item
#pragma unroll
      for (int p = 0; p < P; ++p) gn[p] = gam2[hb2 + (int64_t)sc * (256 * P) + p * 64];      // (raw: used only when live)
      const uint32_t *ow = (const uint32_t *)(tc.orow + (int64_t)sc * KP);
#pragma unroll
      for (int d = 0; d < 4; ++d) on[d] = ow[min(d, (KP >> 2) - 1)];
    };
    request(0);
    for (int s = 0; s < tc.nsmax; ++s) {
      float2 gc[P];
      uint32_t oc[4];
#pragma unroll
      for (int p = 0; p < P; ++p) gc[p] = gn[p];
#pragma unroll
      for (int d = 0; d < 4; ++d) oc[d] = on[d];
      if (s + 1 < tc.nsmax) request(s + 1);
      const bool live = s < tc.ns;
      for (int j = 0; j < nslot; ++j) {
        const int inf = tinfo[j];
        const int col = inf & 127, cnt = (inf >> 7) & 511, lb = (int)((unsigned)inf >> 16);
        uint32_t word = oc[0];
#pragma unroll
        for (int d = 1; d < 4; ++d) word = (col >> 2) == d ? oc[d] : word;
        int sym = (int)((word >> ((col & 3) * 8)) & 0xffu);
        if (col >= 16) sym = live ? (int)tc.orow[(int64_t)s * KP + col] : 0;
        // (a symbol beyond the track's last one lands in the reference's padding cells, which
        //  emission.maximize never reads: not booked)
        if (live && sym < cnt) {
          unsigned long long *hr = hist + (size_t)(lb + sym) * NT + kq;
#pragma unroll
          for (int k = 0; k < KS; ++k)
            if (kq + 4 * k < N)
              atomicAdd(hr + 4 * k, (unsigned long long)__double2ll_rn((double)((k & 1) ? gc[k >> 1].y : gc[k >> 1].x) * fscale));
        }
      }
    }
  }
  __syncthreads();
  // flush: this workgroup's slot of the group's partial buffer, [gridDim.x][rows of the group][NT] behind the groups
  // before it; k_estep_fold_lds maps LDS rows to rows of the statistics table
  size_t goff = 0;
  for (int g = 0; g < grp; ++g) goff += (size_t)gridDim.x * egp->rows[g] * NT;
  double *slot = part + goff + (size_t)blockIdx.x * rows * NT;
  for (int i = threadIdx.x; i < rows * NT; i += blockDim.x) slot[i] = (double)(long long)hist[i] * finv;
}

// ---- the ordered sums over the writers' slots (one thread per cell) ------------------------------------------------
// Eight interleaved running sums (slots s = u mod 8) combined in a fixed tree: one order of additions per cell -- the
// same bits run to run -- with eight loads in flight instead of one (k_estep_fold_xi walked 2 048 slots one dependent
// load at a time: 1.6 ms per E-step).
__device__ __forceinline__ double estep_fold8(const double *src, size_t stride, int nslot) {
  double ps[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int s = 0;
  for (; s + 8 <= nslot; s += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) ps[u] += src[(size_t)(s + u) * stride];
  }
  for (int u = 0; s + u < nslot; ++u) ps[u] += src[(size_t)(s + u) * stride];
  return ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
}
__global__ __launch_bounds__(256) void k_estep_fold_xi(const double *__restrict__ part, int nslot, int N, int NT, double *gC,
                                                       double *gstart) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int cells = NT * NT + NT;
  if (idx >= cells) return;
  const double sum = estep_fold8(part + idx, (size_t)cells, nslot);
  if (idx < NT * NT) {
    if (idx / NT < N && idx % NT < N) gC[idx] += sum;
  } else if (idx - NT * NT < N) {
    gstart[idx - NT * NT] += sum;
  }
}
__global__ __launch_bounds__(256) void k_estep_fold_rows(const double *__restrict__ part, int nslot, int N, int NT,
                                                         const EstepGroups *__restrict__ egp, double *gstat) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nrow = egp->n_rt * 16;
  if (idx >= nrow * NT) return;
  const int row = idx / NT, j = idx - row * NT;
  if (egp->rt_info[row] < 0 || j >= N) return;
  const double sum = estep_fold8(part + idx, (size_t)nrow * NT, nslot);
  gstat[(int64_t)egp->rt_grow[row] * NT + j] += sum;
}
// grid (x over the cells of the largest group, y = LDS group)
__global__ __launch_bounds__(256) void k_estep_fold_lds(const double *__restrict__ part, int nslot, int N, int NT,
                                                        const EstepGroups *__restrict__ egp, double *gstat) {
  const int grp = blockIdx.y;
  const int rows = egp->rows[grp];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * NT) return;
  const int lrow = idx / NT, j = idx - lrow * NT;
  if (j >= N) return;
  size_t goff = 0;
  for (int g = 0; g < grp; ++g) goff += (size_t)nslot * egp->rows[g] * NT;
  // which track of the group owns LDS row lrow
  int grow = -1;
  for (int sl = egp->first[grp]; sl < egp->first[grp + 1]; ++sl) {
    const int inf = egp->info[sl];
    const int cnt = (inf >> 7) & 511, lb = (int)((unsigned)inf >> 16);
    if (lrow >= lb && lrow < lb + cnt) grow = egp->gbase[sl] + (lrow - lb);
  }
  if (grow < 0) return;
  const double sum = estep_fold8(part + goff + idx, (size_t)rows * NT, nslot);
  gstat[(int64_t)grow * NT + j] += sum;
}

}  // namespace tehmm
