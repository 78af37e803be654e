// tehmm_wide_estep.hip.h -- the Baum-Welch E-step on the item-parallel passes of tehmm_wide.hip.h (round 4).
//
// Reference: BaseHMM.fit's per-sequence loop (basehmm.py:504-523) + MultitrackHmm._accumulate_sufficient_statistics
// (hmm.py:545-574) + _hmm._log_sum_lneta (_hmm.pyx:62-117) + _emission.fastAccumulateStats (_emission.pyx:146-190),
// for the two cases the fused passes of tehmm_estep.hip.h do not serve:
//   * 64 <= N <= 128 states (BASELINE configs[4]: the 100-state model has to be TRAINED before it is evaluated);
//     rounds 2-3 ran BaseHMM._do_estep over the array-level entry points, [T][N] fp64 lattices across PCIe five
//     times per sequence;
//   * segment ratios at any N <= 128 (fit on a segmented table, basehmm.py:510-512).  The passes of tehmm_wide.hip.h
//     read their emission rows from HBM as exp(x - max), so the ratio is two more terms of x before the exponential
//     (k_wide_emis_tile, modes 2 / 3):
//         x_t[j] = r_t * normalize * sum_k logb_k[j]  +  [r_t > 1] lt[j][j] (r_t - 1)      (_hmm.pyx:131-140, 178-181)
//     and the recurrences themselves do not change.  Rounds 2-3: ONE sequential chain per interval (k_fb_coop<TRATIO>).
//
// k_wide_fwd leaves alpha'_t as floats; k_wide_bwd<ESTEP> leaves, in the same layout,
//     gamma_t = alpha'_t beta_t / G_t                 -> start, emission histograms, the diagonal ratio term
//     wz_t    = w_{t+1} scale_t / G_t                 -> xi_t(i, j) = alpha'_t[i] A[i][j] wz_t[j]
// The layout (wide_al_index) is, per (16-item tile, position s), a dense [state][item] float matrix: element
// (state j, item i) at ((tile L + s) NPW + j) 16 + i.  Both reductions are GEMMs contracted over the items, on the fp64
// matrix cores (v_mfma_f64_16x16x4: products of floats are exact in fp64, the sums are fp64):
//   * k_wide_estep_xi:    C[i][j] += sum_{items, s} alpha'[i][item] wz[j][item]
//   * k_wide_estep_rows:  out[row][j] += sum_{items, s} a[row][item] gamma[j][item], where a row is
//         (track, symbol):  a = r_t [obs_t[track] == symbol]          (fastAccumulateStats: posterior * segRatio)
//         START:            a = [t == 0]                              (stats['start'] += posteriors[0])
//         DIAG:             a = [t > 0 and r_t > 1] (r_t - 1)         (the `y` term of _log_sum_lneta, _hmm.pyx:94-99:
//                                                                      exp(fwd + bwd + log(r - 1) - logprob) = gamma (r - 1))
//     so every statistic that is linear in gamma comes out of ONE product.
// An operand lane holds FOUR CONSECUTIVE ITEMS of its row (one 16-byte load): the contraction index k of the four
// matrix instructions of a step is item 4 (lane >> 4) + kk -- any bijection items <-> (lane >> 4, kk) is legal as
// long as both operands use it.
// Slots no pass ever writes (positions beyond an item's end, items beyond the last) are zero from the allocation on,
// so the reductions need no masks.  Sums are reproducible: every writer owns a slot of a partial buffer, the fold
// kernels add the slots in ascending order (as tehmm_estep.hip.h).
// There is no exact chain here: an attempt whose links do not verify is repeated with twice the warm-up, then the
// caller falls back (sequential kernels below 64 states, TEHMM_ERR_UNSUPPORTED -> array-level host loop above).
// Tolerance: rows are floats, sums fp64: statistics to ~1e-7 of the reference (bar 1e-6, tests/test_gpu_r4.py).
#pragma once
#include "tehmm_wide.hip.h"
#include "tehmm_estep.hip.h"

#define TEHMM_WIDE_ROW_START 0x40000000
#define TEHMM_WIDE_ROW_DIAG 0x40000001

namespace tehmm {

struct WideRows {
  int n_rt;                              // row tiles (16 rows each)
  int info[TEHMM_ESTEP_MAXRT * 16];      // row -> observation column | symbol << 8, a TEHMM_WIDE_ROW_* code, or -1 (padding)
  int grow[TEHMM_ESTEP_MAXRT * 16];      // (track, symbol) row -> row of the global statistics table
};

// Tracks with MANY symbols (the 250-bin gaussian tracks) do not go through the one-hot product -- 16 row tiles of
// which a 16-item tile touches a few, 448 matrix instructions per position tile and track at 112 states, 5.7 of the
// 6.3 ms of the first version at 100 states -- but are privatised in LDS as 64-bit FIXED-POINT integers (ds_add_u64 is
// exact, hence order-free: reproducible whatever order the waves arrive in), one (track, state range) per workgroup
// role so that [rows][states of the range] fits the 160 KB of a CU.
struct WideLds {
  int n_trk;                             // big tracks
  int col[TEHMM_MAX_TRACKS];             // observation column
  int rows[TEHMM_MAX_TRACKS];            // symbols (rows of the histogram)
  int gbase[TEHMM_MAX_TRACKS];           // first row of the track in the global statistics table
};

// largest segment ratio of a batch (positive doubles order like their bit patterns)
__global__ __launch_bounds__(256) void k_ratio_max(const double *__restrict__ r, int64_t n, unsigned long long *out) {
  double m = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmax(m, r[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

template <int NPW>
struct WideEstepGeom {
  static constexpr int RT = NPW / 16;                        // state tiles
  static constexpr int RSPLIT = RT > 4 ? 4 : 1;              // waves of a workgroup that share one item tile (xi)
  static constexpr int RPW = (RT + RSPLIT - 1) / RSPLIT;     // row tiles of the xi product per wave
  static constexpr int TPW = 4 / RSPLIT;                     // item tiles a workgroup works on at a time
};

// longest item of a 16-item tile (wave-uniform)
__device__ __forceinline__ int wide_tile_nsmax(const IntervalTab &iv, const LaneGeom &lg, int64_t tile, int lane) {
  const int64_t item = tile * 16 + (lane & 15);
  const bool valid = item < lg.n_items;
  const int id = valid ? lg.item_iv[item] : 0;
  const int64_t t0 = valid ? lg.item_t0[item] : 0;
  int ns = valid ? (int)min((int64_t)lg.L, iv.len[id] - t0) : 0;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) ns = max(ns, __shfl_xor(ns, o));
  return __builtin_amdgcn_readfirstlane(ns);
}

// ------------------------------------------------------------------------------------------
// xi:  C[i][j] += sum over (item, position) alpha'[i] wz[j].  grid = persistent workgroups, block = 256.
// NPW > 64: the four waves share an item tile and split the row tiles of C (wave w: row tiles w, w + 4);
// NPW <= 64: every wave takes its own item tile and the whole of C.
// part: [gridDim.x * TPW slots][NPW][NPW]; every cell of every slot is written.
// ------------------------------------------------------------------------------------------
template <int NPW>
__global__ __launch_bounds__(256) void k_wide_estep_xi(IntervalTab iv, LaneGeom lg, const float *__restrict__ AL,
                                                       const float *__restrict__ WZ, double *part, int SQ) {
  using G = WideEstepGeom<NPW>;
  constexpr int RT = G::RT, RSPLIT = G::RSPLIT, RPW = G::RPW, TPW = G::TPW;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int sub = wv % RSPLIT, tsel = wv / RSPLIT;
  const int m = lane & 15, k4 = lane >> 4;
  lane_d4 acc[RPW][RT];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int tb = 0; tb < RT; ++tb) acc[r][tb] = (lane_d4){0.0, 0.0, 0.0, 0.0};
  const int64_t n_tiles = ((int64_t)lg.n_items + 15) / 16;
  const float4 *al4 = (const float4 *)AL, *wz4 = (const float4 *)WZ;
  // work unit = (item tile, one of SQ position ranges of L / SQ): a 2 Mb batch has fewer item tiles than the GPU has SIMDs
  const int LQ = lg.L / SQ;
  for (int64_t unit = (int64_t)blockIdx.x * TPW + tsel; unit < n_tiles * SQ; unit += (int64_t)gridDim.x * TPW) {
    const int64_t tile = unit / SQ;
    const int s_lo = (int)(unit - tile * SQ) * LQ;
    const int nsmax = min(wide_tile_nsmax(iv, lg, tile, lane), s_lo + LQ);
    if (nsmax <= s_lo) continue;
    // float4 (tile, s, state, k4) = ((tile L + s) NPW + state) 4 + k4
    const int64_t b4 = tile * lg.L * (int64_t)(NPW * 4) + m * 4 + k4;
    float4 xa[RPW], xw[RT];
    auto request = [&](int s) {
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const int ta = min(sub + RSPLIT * r, RT - 1);
        xa[r] = al4[b4 + ((int64_t)s * NPW + 16 * ta) * 4];
      }
#pragma unroll
      for (int tb = 0; tb < RT; ++tb) xw[tb] = wz4[b4 + ((int64_t)s * NPW + 16 * tb) * 4];
    };
    request(s_lo);
    for (int s = s_lo; s < nsmax; ++s) {
      double a[RPW][4], w[RT][4];
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        a[r][0] = (double)xa[r].x; a[r][1] = (double)xa[r].y; a[r][2] = (double)xa[r].z; a[r][3] = (double)xa[r].w;
      }
#pragma unroll
      for (int tb = 0; tb < RT; ++tb) {
        w[tb][0] = (double)xw[tb].x; w[tb][1] = (double)xw[tb].y; w[tb][2] = (double)xw[tb].z; w[tb][3] = (double)xw[tb].w;
      }
      request(min(s + 1, nsmax - 1));            // (unconditional: count-static waits, see k_wide_estep_rows)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          if (sub + RSPLIT * r < RT) {
#pragma unroll
            for (int tb = 0; tb < RT; ++tb)
              acc[r][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r][kk], w[tb][kk], acc[r][tb], 0, 0, 0);
          }
        }
    }
  }
  // accumulator (lane, register q) of (row tile ta, column tile tb) is C[16 ta + 4 q + (lane >> 4)][16 tb + (lane & 15)]
  double *slot = part + ((size_t)blockIdx.x * TPW + tsel) * (size_t)(NPW * NPW);
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int ta = sub + RSPLIT * r;
    if (ta < RT) {
#pragma unroll
      for (int tb = 0; tb < RT; ++tb)
#pragma unroll
        for (int q = 0; q < 4; ++q) slot[(size_t)(16 * ta + 4 * q + k4) * NPW + 16 * tb + m] = acc[r][tb][q];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Everything linear in gamma: out[row][j] += sum over (item, position) a[row][item] gamma[j][item].
// grid (x = persistent workgroups over the units, y = groups of RTW row tiles), block = 256: every wave takes its own
// unit and the RTW row tiles of its group (the first version gave a wave ONE row tile: 7 row loads and 28 conversions
// per 28 matrix instructions, 1.0 ms per 2 Mb for 0.27 ms of matrix pipe).  part: [4 gridDim.x slots][n_rt * 16 rows][NPW].
// ------------------------------------------------------------------------------------------
template <int NPW, bool RATIO, int RTW>
__global__ __launch_bounds__(256) void k_wide_estep_rows(IntervalTab iv, LaneGeom lg, const WideRows *__restrict__ wr, int KP,
                                                         const uint8_t *__restrict__ obs, const double *__restrict__ ratios,
                                                         const float *__restrict__ GAM, double *part, int SQ) {
  constexpr int RT = NPW / 16;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int m = lane & 15, k4 = lane >> 4;
  const int nrt = wr->n_rt;
  // this wave's RTW row tiles (rows beyond the table: padding rows, a = 0)
  int kind[RTW], col[RTW], rsym[RTW];
#pragma unroll
  for (int r = 0; r < RTW; ++r) {
    const int rt = blockIdx.y * RTW + r;
    const int inf = rt < nrt ? wr->info[rt * 16 + m] : -1;
    kind[r] = inf < 0 ? 3 : inf == TEHMM_WIDE_ROW_START ? 1 : inf == TEHMM_WIDE_ROW_DIAG ? 2 : 0;
    col[r] = kind[r] == 0 ? (inf & 255) : 0;
    rsym[r] = kind[r] == 0 ? (inf >> 8) : -1;
  }
  lane_d4 acc[RTW][RT];
#pragma unroll
  for (int r = 0; r < RTW; ++r)
#pragma unroll
    for (int tb = 0; tb < RT; ++tb) acc[r][tb] = (lane_d4){0.0, 0.0, 0.0, 0.0};
  const int64_t n_tiles = ((int64_t)lg.n_items + 15) / 16;
  const float4 *gam4 = (const float4 *)GAM;
  const int LQ = lg.L / SQ;
  for (int64_t unit = (int64_t)blockIdx.x * 4 + wv; unit < n_tiles * SQ; unit += (int64_t)gridDim.x * 4) {
    const int64_t tile = unit / SQ;
    const int s_lo = (int)(unit - tile * SQ) * LQ;
    // the lane's four items 4 k4 + kk: observation rows, ratios, lengths
    int64_t oo[4], rp[4], t0k[4];
    int nsk[4], nsmax = 0;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int64_t item = tile * 16 + 4 * k4 + kk;
      const bool valid = item < lg.n_items;
      const int64_t itc = valid ? item : (int64_t)lg.n_items - 1;
      const int id = lg.item_iv[itc];
      const int64_t t0 = lg.item_t0[itc];
      nsk[kk] = valid ? (int)min((int64_t)lg.L, iv.len[id] - t0) : 0;
      rp[kk] = iv.pos0[id] + t0;
      oo[kk] = rp[kk] * KP;
      t0k[kk] = t0;
      nsmax = max(nsmax, nsk[kk]);
    }
    nsmax = max(nsmax, __shfl_xor(nsmax, 16));
    nsmax = max(nsmax, __shfl_xor(nsmax, 32));
    nsmax = min(__builtin_amdgcn_readfirstlane(nsmax), s_lo + LQ);
    if (nsmax <= s_lo) continue;
    const int64_t b4 = tile * lg.L * (int64_t)(NPW * 4) + m * 4 + k4;
    // TWO steps are on their way at any time (buffers 0 / 1): the matrix instructions of one step do not cover the
    // latency of the next step's rows
    float4 xg[2][RT];
    int sy[2][RTW][4];
    double rr[2][4];
    auto request = [&](int s, auto bsel) {
      constexpr int B = decltype(bsel)::value;
#pragma unroll
      for (int tb = 0; tb < RT; ++tb) xg[B][tb] = gam4[b4 + ((int64_t)s * NPW + 16 * tb) * 4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int sc = max(0, min(s, nsk[kk] - 1));           // unconditional loads from inside the item
#pragma unroll
        for (int r = 0; r < RTW; ++r) sy[B][r][kk] = (int)obs[oo[kk] + (int64_t)sc * KP + col[r]];
        rr[B][kk] = RATIO ? ratios[rp[kk] + sc] : 1.0;
      }
    };
    auto step = [&](int s, auto bsel) {
      constexpr int B = decltype(bsel)::value;
      double a[RTW][4], g[RT][4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double r = rr[B][kk];
        const bool live = s < min(nsk[kk], nsmax);
        const double vstart = (s == 0 && t0k[kk] == 0) ? 1.0 : 0.0;
        const double vdiag = (RATIO && r > 1. && t0k[kk] + s > 0) ? r - 1. : 0.0;
#pragma unroll
        for (int q = 0; q < RTW; ++q) {
          double v = (sy[B][q][kk] == rsym[q]) ? r : 0.0;                             // kind 0 (rsym = -1 otherwise)
          v = kind[q] == 1 ? vstart : v;
          v = kind[q] == 2 ? vdiag : v;
          a[q][kk] = live ? v : 0.0;
        }
      }
#pragma unroll
      for (int tb = 0; tb < RT; ++tb) {
        g[tb][0] = (double)xg[B][tb].x; g[tb][1] = (double)xg[B][tb].y; g[tb][2] = (double)xg[B][tb].z; g[tb][3] = (double)xg[B][tb].w;
      }
      // (every step issues the same loads -- the last ones a second time --: behind a branch that may or may not have
      //  issued them the compiler's wait for the OLDER buffer becomes vmcnt(0), and the latency of every step shows)
      request(min(s + 2, nsmax - 1), bsel);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int q = 0; q < RTW; ++q)
#pragma unroll
          for (int tb = 0; tb < RT; ++tb) acc[q][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][kk], g[tb][kk], acc[q][tb], 0, 0, 0);
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    request(s_lo, B0{});
    request(min(s_lo + 1, nsmax - 1), B1{});
    for (int s = s_lo; s < nsmax; s += 2) {
      step(s, B0{});
      step(s + 1, B1{});                         // (s + 1 == nsmax: nothing is live, zeros are added)
    }
  }
  double *slot = part + ((size_t)blockIdx.x * 4 + wv) * (size_t)(nrt * 16) * NPW;
#pragma unroll
  for (int r = 0; r < RTW; ++r) {
    const int rt = blockIdx.y * RTW + r;
    if (rt < nrt) {
#pragma unroll
      for (int tb = 0; tb < RT; ++tb)
#pragma unroll
        for (int q = 0; q < 4; ++q) slot[(size_t)(rt * 16 + 4 * q + k4) * NPW + 16 * tb + m] = acc[r][tb][q];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Emission histograms of the tracks with many symbols: hist [rows][HS] fixed-point in LDS, HS = NPW / NSPLIT states.
// grid (x = persistent workgroups over the item tiles, y = track * NSPLIT + state range), block = 512 (8 waves, one
// item tile each).  Lane = (state sub-index lane >> 4, item lane & 15): one load covers four states of the 16 items.
// part: per y, [gridDim.x slots][rows][HS] behind the y before it.
// ------------------------------------------------------------------------------------------
template <int NPW, int NSPLIT, bool RATIO>
__global__ __launch_bounds__(512) void k_wide_estep_hist_lds(IntervalTab iv, LaneGeom lg, const WideLds *__restrict__ wl, int N,
                                                             int KP, const uint8_t *__restrict__ obs,
                                                             const double *__restrict__ ratios, const float *__restrict__ GAM,
                                                             double *part, int shift, int SQ) {
  constexpr int HS = NPW / NSPLIT, HQ = HS / 4;
  extern __shared__ unsigned long long wide_hist[];
  const int trk = blockIdx.y / NSPLIT, j0 = (blockIdx.y % NSPLIT) * HS;
  const int col = wl->col[trk], rows = wl->rows[trk];
  for (int i = threadIdx.x; i < rows * HS; i += blockDim.x) wide_hist[i] = 0ull;
  __syncthreads();
  const double fscale = ldexp(1.0, shift), finv = ldexp(1.0, -shift);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kq = lane >> 4, i16 = lane & 15;
  const int64_t n_tiles = ((int64_t)lg.n_items + 15) / 16;
  const int LQ = lg.L / SQ;
  for (int64_t unit = (int64_t)blockIdx.x * 8 + wv; unit < n_tiles * SQ; unit += (int64_t)gridDim.x * 8) {
    const int64_t tile = unit / SQ;
    const int s_lo = (int)(unit - tile * SQ) * LQ;
    const int64_t item = tile * 16 + i16;
    const bool valid = item < lg.n_items;
    const int64_t itc = valid ? item : (int64_t)lg.n_items - 1;
    const int id = lg.item_iv[itc];
    const int64_t t0 = lg.item_t0[itc];
    const int ns = valid ? (int)min((int64_t)lg.L, iv.len[id] - t0) : 0;
    int nsmax = ns;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) nsmax = max(nsmax, __shfl_xor(nsmax, o));
    nsmax = min(__builtin_amdgcn_readfirstlane(nsmax), s_lo + LQ);
    if (nsmax <= s_lo) continue;
    const int64_t pos = iv.pos0[id] + t0;
    const uint8_t *op = obs + pos * KP + col;
    const float *gp = GAM + (tile * lg.L * (int64_t)NPW + j0 + kq) * 16 + i16;       // + (s NPW + 4 i) 16
    float gn[HQ];
    int syn;
    double rn;
    auto request = [&](int s) {
      const int sc = max(0, min(s, ns - 1));                 // unconditional loads from inside the item
#pragma unroll
      for (int i = 0; i < HQ; ++i) gn[i] = gp[((int64_t)s * NPW + 4 * i) * 16];
      syn = (int)op[(int64_t)sc * KP];
      rn = RATIO ? ratios[pos + sc] : 1.0;
    };
    request(s_lo);
    for (int s = s_lo; s < nsmax; ++s) {
      float gc[HQ];
#pragma unroll
      for (int i = 0; i < HQ; ++i) gc[i] = gn[i];
      const int sym = syn;
      const double f = rn * fscale;
      request(min(s + 1, nsmax - 1));
      // (a symbol beyond the track's last one lands in the reference's padding cells, which emission.maximize never
      //  reads: not booked)
      if (s < ns && sym < rows) {
        unsigned long long *hr = wide_hist + (size_t)sym * HS + kq;
#pragma unroll
        for (int i = 0; i < HQ; ++i)
        {
          // (posteriors are sparse: most states of a position hold nothing at the histogram's resolution, and adding an
          //  exact zero is no addition)
          const long long q = __double2ll_rn((double)gc[i] * f);
          if (q != 0 && j0 + 4 * i + kq < N) atomicAdd(hr + 4 * i, (unsigned long long)q);
        }
      }
    }
  }
  __syncthreads();
  size_t goff = 0;
  for (int y = 0; y < (int)blockIdx.y; ++y) goff += (size_t)gridDim.x * wl->rows[y / NSPLIT] * HS;
  double *slot = part + goff + (size_t)blockIdx.x * rows * HS;
  for (int i = threadIdx.x; i < rows * HS; i += blockDim.x) slot[i] = (double)(long long)wide_hist[i] * finv;
}

// grid (x over the cells of the largest track, y = track * nsplit + state range)
__global__ __launch_bounds__(256) void k_wide_fold_lds(const double *__restrict__ part, int nslot, int N, int NP, int HS, int nsplit,
                                                       const WideLds *__restrict__ wl, double *gstat) {
  const int trk = blockIdx.y / nsplit, j0 = (blockIdx.y % nsplit) * HS;
  const int rows = wl->rows[trk];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * HS) return;
  const int lrow = idx / HS, j = j0 + idx - lrow * HS;
  if (j >= N) return;
  size_t goff = 0;
  for (int y = 0; y < (int)blockIdx.y; ++y) goff += (size_t)nslot * wl->rows[y / nsplit] * HS;
  double sum = 0.0;
  for (int s = 0; s < nslot; ++s) sum += part[goff + (size_t)s * rows * HS + idx];
  gstat[(int64_t)(wl->gbase[trk] + lrow) * NP + j] += sum;
}

// ---- the ordered sums over the writers' slots (one thread per cell, slots ascending) ----------------------------
__global__ __launch_bounds__(256) void k_wide_fold_xi(const double *__restrict__ part, int nslot, int N, int NPW, int NP, double *gC) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * N) return;
  const int i = idx / N, j = idx - i * N;
  // eight interleaved running sums (slots s = u mod 8), combined in a fixed order: still one order of additions per
  // cell, with eight loads in flight instead of one
  double ps[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const double *src = part + (size_t)i * NPW + j;
  int s = 0;
  for (; s + 8 <= nslot; s += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) ps[u] += src[(size_t)(s + u) * NPW * NPW];
  }
  for (int u = 0; s + u < nslot; ++u) ps[u] += src[(size_t)(s + u) * NPW * NPW];
  gC[(size_t)i * NP + j] += ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
}
__global__ __launch_bounds__(256) void k_wide_fold_rows(const double *__restrict__ part, int nslot, int N, int NPW, int NP,
                                                        const WideRows *__restrict__ wr, double *gstat, double *gstart, double *gD) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nrow = wr->n_rt * 16;
  if (idx >= nrow * N) return;
  const int row = idx / N, j = idx - row * N;
  const int inf = wr->info[row];
  if (inf < 0) return;
  double ps[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // (eight interleaved running sums, fixed order: see k_wide_fold_xi)
  const double *src = part + (size_t)row * NPW + j;
  const size_t stride = (size_t)nrow * NPW;
  int s = 0;
  for (; s + 8 <= nslot; s += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) ps[u] += src[(size_t)(s + u) * stride];
  }
  for (int u = 0; s + u < nslot; ++u) ps[u] += src[(size_t)(s + u) * stride];
  const double sum = ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
  if (inf == TEHMM_WIDE_ROW_START) gstart[j] += sum;
  else if (inf == TEHMM_WIDE_ROW_DIAG) gD[j] += sum;
  else gstat[(int64_t)wr->grow[row] * NP + j] += sum;
}

}  // namespace tehmm
