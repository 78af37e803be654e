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
//   * the emission rows are not fused: k_wide_emis_tile (round 4: in the lane mapping of the passes, see below) leaves
//     exp(row - max) and the max per position in HBM in the layout of the alpha' rows, both passes read them back
//     (2.7 KB per position, HBM time << matrix time at this N);
//   * there is no sequential chain: an item either starts exactly (the first item of an interval forward, the last
//     one backward) or warms up over Wu positions from a uniform vector, and k_wide_links CHECKS every link (Hilbert
//     distance <= 1e-8 (round 4; 1e-10 before) between the vector an item arrives with and the one its neighbour left).  One failed
//     link -- or an emission row no state can emit, whose semantics (quirk Q9, NaN lattices) belong to the sequential
//     kernels -- and the host retries with twice the warm-up, then falls back to the sequential kernels.
// Padded state counts: 80, 96, 112, 128 (row tiles of 16); pads carry zero probability.
// Tolerance: posteriors to 1e-6 (alpha' rows are kept as floats between the passes, as in the fused passes);
// the forward log-likelihood is summed in fp64 from the items' scale records and the links' ratios.
#pragma once
#include <type_traits>
#include "tehmm_fused.hip.h"

#define TEHMM_WIDE_S 128            // row stride of the wide log-row buffer and of the LDS table

namespace tehmm {

template <int NPW>
struct WideGeom {
  static constexpr int KS = NPW / 4;        // states per lane
  static constexpr int RT = NPW / 16;       // row tiles
  static constexpr size_t FRAG_BYTES = (size_t)RT * KS * 64 * sizeof(double);
};

// ---- emission rows in the TILE layout ----------------------------------------------------------------------------
// E element (item, position s of the item, state kq + 4 k) at wide_e_row(item, L, s) + k 64 + kq 16: per (16-item
// tile, position) a dense [state][item] matrix of doubles, the layout of the alpha' rows (wide_al_index) -- the passes
// read a row with one coalesced 512-byte load per k.  ms [(tile L + s) 16 + item & 15] = the row's maximum.
// (Rounds 3 / 4a kept E as [user row][NPW], written by a lane = state kernel that walked an item position by position
//  behind ~480 mostly scalar instructions per position -- symbol decode and table-row addresses are per POSITION work --
//  1.7 ms per 2 Mb at 100 states, 3.4 ms per 5 Mb at 35; the passes then read it 32 bytes per lane group.)
template <int NPW>
__device__ __forceinline__ int64_t wide_e_row(int64_t item, int L, int s) {
  return ((((item >> 4) * L + s) * WideGeom<NPW>::KS) << 6) + (item & 15);
}
// position s (relative to `item`, may lie before or behind it inside the same interval) = row s2 of item it2
__device__ __forceinline__ void wide_e_start(int64_t item, int L, int s, int64_t &it2, int &s2) {
  const int d = s >= 0 ? s / L : -((-s + L - 1) / L);
  it2 = item + d;
  s2 = s - d * L;
}
// row stride (doubles) of k_wide_emis_tile's redistribution buffer: the smallest value >= NPW that is 4 mod 32, so that the
// 64 lanes (item, kq) of a read-back hit 32 different 8-byte banks twice
__host__ __device__ constexpr int wide_emis_tstride(int NPW) { return NPW <= 36 ? 36 : NPW <= 68 ? 68 : NPW <= 100 ? 100 : 132; }

// The emission kernel in the lane mapping of the passes: lane = (item of a 16-item tile, state quarter kq), a lane
// holds the states kq + 4 k of ITS item's position.  Per track the symbol is per-lane data (one byte extract), the
// table row one base address, and the row's KS values KS loads at immediate offsets + KS additions: ~80 instructions
// per position instead of ~480.  Small tracks' rows and the diagonal of the transition matrix are staged in LDS.
//   MODE 0: tables (posterior: no ratios);  MODE 1: from the log rows BL [internal position][128] the exact Viterbi of
//   the same evaluation has written;  MODE 2: fit without ratios;  MODE 3: fit with segment ratios (emission rows
//   scaled by r_t, emission.py:195-196; + lt[j][j] (r_t - 1) where r_t > 1, _hmm.pyx:131-140).
// grid = persistent workgroups over units of (tile, 16 positions); block = 256 (one unit per wave at a time).
// LDS: tbuf [4 waves][8][TS] | ltab [lds_rows][NP] | ltd [NPW] | tinfo [3][K].  flags[0] counts units with a row no state
// can emit.
template <int NPW, int MODE>
__global__ __launch_bounds__(256) void k_wide_emis_tile(IntervalTab iv, EmisTab em, LaneGeom lg, int N, int NP,
                                                        const double *__restrict__ g_lt, const double *__restrict__ tratios,
                                                        const double *__restrict__ BL, double *E, double *ms, int *flags) {
  constexpr int KS = WideGeom<NPW>::KS;
  constexpr bool FROM_LOG = MODE == 1, TRATIO = MODE == 3;
  extern __shared__ double emis_lds[];
  constexpr int TS = wide_emis_tstride(NPW);          // row stride of the per-wave redistribution buffer (== 4 mod 32)
  double *tbuf = emis_lds + (size_t)(threadIdx.x >> 6) * (8 * TS);      // [4 waves][8 items][TS]
  double *ltab = emis_lds + (FROM_LOG ? 0 : 4 * 8 * TS);
  double *ltd = ltab + (size_t)(FROM_LOG ? 0 : em.lds_rows) * NP;
  int *tinfo = (int *)(ltd + NPW);
  const int K = em.K, KPW = em.KPW;
  if (!FROM_LOG) {
    // (em.ldsbase is this kernel's own assignment, launch_wide_emis: more tracks than the 32 KB the cooperative kernels
    //  stage -- a row gathered from L2 in this lane mapping touches 16 cache lines per load; rows come from the table itself)
    for (int k = 0; k < K; ++k) {
      const int lb = em.ldsbase[k];
      if (lb < 0) continue;
      const double *src = em.tab + (size_t)em.rowbase[k] * NP;
      for (int i = threadIdx.x; i < em.rowcnt[k] * NP; i += blockDim.x) ltab[(size_t)lb * NP + i] = src[i];
    }
    for (int i = threadIdx.x; i < NP; i += blockDim.x) ltab[(size_t)em.lds_zero * NP + i] = 0.0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
      tinfo[k] = em.ldsbase[k];
      tinfo[K + k] = em.rowcnt[k];
      tinfo[2 * K + k] = em.rowbase[k];
    }
  }
  for (int j = threadIdx.x; j < NPW; j += blockDim.x) ltd[j] = (TRATIO && j < N) ? g_lt[(size_t)j * NP + j] : 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kq = lane >> 4, i16 = lane & 15;
  const int L = lg.L;
  const int SB = 16, nsb = (L + SB - 1) / SB;
  const int64_t n_tiles = ((int64_t)lg.n_items + 15) / 16;
  const bool eratio = em.ratios != nullptr;
  int nbad = 0;
  for (int64_t unit = (int64_t)blockIdx.x * 4 + wv; unit < n_tiles * nsb; unit += (int64_t)gridDim.x * 4) {
    const int64_t tile = unit / nsb;
    const int s_lo = (int)(unit - tile * nsb) * SB;
    const int64_t item = tile * 16 + i16;
    const bool valid = item < lg.n_items;
    const int64_t itc = valid ? item : (int64_t)lg.n_items - 1;
    const int id = lg.item_iv[itc];
    const int64_t t0 = lg.item_t0[itc];
    const int len = valid ? (int)min((int64_t)L, iv.len[id] - t0) : 0;
    const int64_t g0 = iv.pos0[id] + t0;
    const int s_hi = min(L, s_lo + SB);
    // the observation words of the NEXT position are requested while this one is worked on (first four words: 16 tracks)
    uint32_t wn[4] = {0u, 0u, 0u, 0u};
    auto request = [&](int s) {
      const uint32_t *orow = em.obs32 + (g0 + (s < len ? s : 0)) * KPW;
#pragma unroll
      for (int d = 0; d < 4; ++d)
        if (d < KPW) wn[d] = orow[d];
    };
    if (!FROM_LOG) request(s_lo);
    for (int s = s_lo; s < s_hi; ++s) {
      const bool act = s < len;
      const int sc = act ? s : 0;                                  // (loads from inside the item whatever happens)
      double x[KS];
      if (FROM_LOG) {
        const double *src = BL + (g0 + sc) * TEHMM_WIDE_S + kq;
#pragma unroll
        for (int k = 0; k < KS; ++k) x[k] = src[4 * k];
      } else {
#pragma unroll
        for (int k = 0; k < KS; ++k) x[k] = 0.0;
        uint32_t wc[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) wc[d] = wn[d];
        if (s + 1 < s_hi) request(s + 1);
        auto track = [&](int kk, uint32_t wd) {
          const int sym = (int)((wd >> ((kk & 3) * 8)) & 0xffu);
          const int lb = tinfo[kk];
          const bool inr = sym < tinfo[K + kk];
          // (a row is read up to state NPW - 1: beyond the model's NP padded states that is the next row or the slack
          //  behind the table -- those states are dropped below)
          // (two loops, not one pointer selected between LDS and global memory: that would be a FLAT pointer, and flat
          //  loads of LDS addresses take the slow path -- the first version spent 78 % of its time waiting on them)
          if (lb >= 0) {
            const double *tr = ltab + (size_t)(inr ? lb + sym : em.lds_zero) * NP + kq;
#pragma unroll
            for (int k = 0; k < KS; ++k) x[k] += tr[4 * k];
          } else {
            // A track whose rows stay in global memory (the 250-bin tracks).  Gathered in this lane mapping -- 32 bytes
            // per item and load -- a row costs 16 cache-line reads per load, 28 line reads per position and track at 112
            // states: 1.1e8 per 2 Mb, 7 TB/s of L2 traffic, 1.9 ms with the vector pipe 7 % busy.  Instead the WAVE reads
            // the row of one item at a time, coalesced (two 512-byte loads), into a per-wave LDS buffer [8 items][TS]
            // (TS == 4 mod 32: the read-back below is conflict-free), and every lane takes its states from there.
            const int rb = tinfo[2 * K + kk];
            const int rown = inr ? rb + sym : em.zero_row;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              double r0v[8], r1v[8];
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const int ri = __builtin_amdgcn_readlane(rown, half * 8 + i);      // lane i16 (kq = 0) holds item i16's row
                const double *row = em.tab + (int64_t)ri * NP;
                r0v[i] = row[min(lane, NP - 1)];
                r1v[i] = NPW > 64 ? row[min(lane + 64, NP - 1)] : 0.0;
              }
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                if (lane < TS) tbuf[i * TS + lane] = r0v[i];
                if (NPW > 64 && lane + 64 < TS) tbuf[i * TS + lane + 64] = r1v[i];
              }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              if ((i16 >> 3) == half) {
                const double *tr = tbuf + (i16 & 7) * TS + kq;
#pragma unroll
                for (int k = 0; k < KS; ++k) x[k] += tr[4 * k];
              }
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
              __builtin_amdgcn_wave_barrier();
            }
          }
        };
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          if (4 * d < K) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (4 * d + u < K) track(4 * d + u, wc[d]);
          }
        }
        for (int kk = 16; kk < K; ++kk) track(kk, em.obs32[(g0 + sc) * KPW + (kk >> 2)]);
        double r = 1.0;
        if (eratio || TRATIO) r = (TRATIO ? tratios : em.ratios)[g0 + sc];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          x[k] *= em.normalize;
          if (eratio) x[k] *= r;
          if (TRATIO && r > 1.) x[k] += ltd[kq + 4 * k] * (r - 1.);
        }
      }
      double m = -INFINITY;
#pragma unroll
      for (int k = 0; k < KS; ++k)
        if (kq + 4 * k < N) m = fmax(m, x[k]);
      m = item_max4(m);
      const bool good = m > -1e20;
      nbad += (act && !good) ? 1 : 0;
      if (act) {
        double *dst = E + wide_e_row<NPW>(item, L, s) + kq * 16;
#pragma unroll
        for (int k = 0; k < KS; ++k) dst[(int64_t)k << 6] = (good && kq + 4 * k < N) ? exp_nonpos(x[k] - m) : 0.0;
        if (kq == 0) ms[(tile * L + s) * 16 + i16] = good ? m : 0.0;
      }
    }
  }
  if (__any(nbad > 0) && lane == 0) atomicAdd(&flags[0], 1);
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
  const int64_t t0 = valid ? lg.item_t0[item] : 0, T = iv.len[id];
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
  // Emission rows in the TILE layout (wide_e_row: the layout of the alpha' rows, one coalesced 512-byte load per k):
  // position s of this lane's item is row s2 of item it2 -- another item of the same interval while s runs through the
  // warm-up (it2, s2 advance with s)
  const int64_t itc = valid ? item : (int64_t)lg.n_items - 1;
  int64_t it2;
  int s2;
  wide_e_start(itc, L, -Wu, it2, s2);
  const int64_t own0 = wide_e_row<NPW>(itc, L, 0) + kq * 16;
  for (int s = -Wu; s < L; ++s) {
    const bool act = valid && s < len && s >= -wu;
    if (s == 0 && valid && t0 > 0 && len > 0) vec_out(pre);
    const double *er = E + (act ? wide_e_row<NPW>(it2, L, s2) + kq * 16 : own0);
    const double msv = act ? ms[((it2 >> 4) * L + s2) * 16 + (it2 & 15)] : 0.0;
    if (++s2 == L) { s2 = 0; ++it2; }
    // the emission row is requested BEFORE the product (one wave per SIMD: nothing else hides its latency; round 3 read
    // it behind the product, twice)
    double ev[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) ev[k] = er[(int64_t)k << 6];
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
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      ev[k] = acc[k >> 2][k & 3] * ev[k];
      t += act ? ev[k] : 0.0;
    }
    t = item_sum4(t);
    const int e = ((__double2hiint(t) >> 20) & 0x7ff) - 1022;
    const double scale = __hiloint2double((1023 - e) << 20, 0);
    if (act) {
#pragma unroll
      for (int k = 0; k < KS; ++k) v[k] = ev[k] * scale;
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
// ESTEP (tehmm_wide_estep.hip.h): instead of the posterior row, the float rows gamma_t = alpha'_t beta_t / G_t and
// wz_t = w_{t+1} scale_t / G_t (zero at an interval's last position: no transition leaves it) in the alpha' layout.
template <int NPW, bool ESTEP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_wide_bwd(IntervalTab iv, LaneGeom lg, int N, int NP, int Wu, const double *__restrict__ A,
                const double *__restrict__ E, const float *__restrict__ AL, double *post, double *pre, double *end,
                float *GAM = nullptr, float *WZ = nullptr) {
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
  const int64_t itc = valid ? item : (int64_t)lg.n_items - 1;
  int64_t it2;
  int s2;
  wide_e_start(itc, L, L + Wu - 1, it2, s2);                     // (emission rows in the tile layout: see k_wide_fwd)
  const int64_t own0 = wide_e_row<NPW>(itc, L, 0) + kq * 16;
  for (int s = L + Wu - 1; s >= 0; --s) {
    const bool act = valid && len > 0 && s <= top;
    if (s == L - 1 && valid && !last) vec_out(pre);              // v = w_{t0 + L} as the warm-up left it
    const double *er = E + (act ? wide_e_row<NPW>(it2, L, s2) + kq * 16 : own0);
    if (--s2 < 0) { s2 = L - 1; --it2; }
    const bool official = act && s < len;
    // emission and alpha' rows of this step are requested BEFORE the product (one wave per SIMD: nothing else hides
    // their latency); the alpha' row from a clamped position where the step has none
    // (128 padded states, posterior form: the two rows do not fit the 512 registers -- they are read behind the product)
    constexpr bool PRE_AL = ESTEP || NPW < 128;
    double ev[KS];
    float av[KS];
    const float *ar = AL + wide_al_index<NPW>(tile, L, min(s, L - 1), 0, lane);
    if (PRE_AL) {
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        ev[k] = er[(int64_t)k << 6];
        av[k] = ar[(int64_t)k << 6];
      }
    }
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
      if (!PRE_AL) {
#pragma unroll
        for (int k = 0; k < KS; ++k) av[k] = ar[(int64_t)k << 6];
      }
      double gt = 0.0;                          // (the products are formed twice rather than held in KS more registers)
#pragma unroll
      for (int k = 0; k < KS; ++k) gt += (double)av[k] * acc[k >> 2][k & 3];
      gt = item_sum4(gt);
      const double inv = 1.0 / gt;
      if (ESTEP) {
        const int64_t ix = wide_al_index<NPW>(tile, L, s, 0, lane);
        const double wzs = scale * inv;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          GAM[ix + ((int64_t)k << 6)] = (float)(((double)av[k] * acc[k >> 2][k & 3]) * inv);
          WZ[ix + ((int64_t)k << 6)] = s == top ? 0.f : (float)(v[k] * wzs);
        }
      } else {
        double *pr = post + (r0 + s) * (int64_t)N + kq;
#pragma unroll
        for (int k = 0; k < KS; ++k)
          if (kq + 4 * k < N) pr[4 * k] = (((double)av[k] * acc[k >> 2][k & 3]) * inv + eps) * inv_epsden;
      }
    }                                 // (the four lanes of an item take the branch together: item_sum4 is safe)
    if (act) {
#pragma unroll
      for (int k = 0; k < KS; ++k) v[k] = (PRE_AL ? ev[k] : er[(int64_t)k << 6]) * acc[k >> 2][k & 3];
    }
  }
  if (valid && len > 0) vec_out(end);
}

// ---- links and log-likelihood --------------------------------------------------------------------------------
// one thread per item: forward link (item vs item - 1) and backward link (item vs item + 1) inside an interval;
// flags[1] counts failed links; lr [item] = log(rho) of the forward link
__global__ __launch_bounds__(256) void k_wide_links(IntervalTab iv, LaneGeom lg, int N, int NPW, const double *pre_f,
                                                    const double *end_f, const double *pre_b, const double *end_b,
                                                    double *lr, int *flags, double tol = TEHMM_FB_TOL) {
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
    const bool ok = !bad && rmax > 0.0 && rmin < INFINITY && rmax / rmin - 1.0 <= tol;
    fails += !ok;
    lr[item] = ok ? 0.5 * (log(rmax) + log(rmin)) : 0.0;     // (geometric mean of the extreme ratios: no bias of one sign)
  }
  if (t0 + lg.L < T) {                                       // backward: pre_b[item] against end_b[item + 1]
    const double *a = pre_b + item * NPW, *b = end_b + (item + 1) * NPW;
    bool bad = false;
    double rmax = -1.0, rmin = INFINITY;
    for (int j = 0; j < N; ++j) link_accum(a[j], b[j], bad, rmax, rmin);
    const bool ok = !bad && rmax > 0.0 && rmin < INFINITY && rmax / rmin - 1.0 <= tol;
    fails += !ok;
  }
  if (fails) atomicAdd(&flags[1], fails);
}

// one WAVE per interval (round 3: one thread, 0.2 ms for 20 intervals of 780 items): log P = sum of the items' scale
// gains - the links' log(rho) + log(sum of the last vector); lane l adds the items l, l + 64, ... in order, the lanes
// are folded in a fixed tree -- the same bits run to run
__global__ __launch_bounds__(256) void k_wide_loglik(IntervalTab iv, LaneGeom lg, int N, int NPW, const double *end_f,
                                                    const double *SL, const double *lr, double *fwd_logprob) {
  const int lane = threadIdx.x & 63;
  const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (id >= iv.n) return;
  const int64_t T = iv.len[id];
  if (T <= 0) return;
  const int64_t i0 = lg.ifirst[id], i1 = lg.ifirst[id + 1];
  double s = 0.0;
  for (int64_t it = i0 + lane; it < i1; it += 64) s += SL[it] - lr[it];
  double tot = 0.0;
  const double *e = end_f + (i1 - 1) * NPW;
  for (int j = lane; j < N; j += 64) tot += e[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    tot += __shfl_xor(tot, o);
  }
  if (lane == 0) fwd_logprob[id] = s + log(tot);
}


// ==========================================================================================================
// Chunk-parallel EXACT Viterbi for 64 <= N <= 128 (decode on BASELINE configs[4]: 100 states, segment ratios).
// The scheme of tehmm_spec.hip.h -- P0 plain chunk gains -> host binade placement -> P2 quantised chunk pass (exact
// integer max-plus inside a binade, the from-index in the low mantissa bits of the quantised table) -> exact chain
// with verified jumps -> traceback -- with the lane = state mapping at two states per lane:
//   * k_vit_wide_spec: ONE WAVE PER CHUNK, lane = to-states (lane, lane + 64).  The table of the workgroup's binade
//     lives in LDS ([from][128] doubles, built per workgroup: eight chunks of one binade share it), the vector of
//     the previous position reaches the lanes as scalar operands (v_readlane), so a step is N x (2 readlane,
//     2 ds_read_b64, 2 add, 2 max) and nothing crosses lanes but the rare reductions.  Seven index bits:
//     W = 128 (value - base) + (127 - from) u, re-based every 32 steps.
//   * segment ratios (_hmm.pyx:229-247, quirk Q4) as in k_vit_lane: every separately rounded addend of the
//     reference's sums is one more grid-rounded term; candidates from >= 1 share R(lt[j][j] (r - 1)) [r > 1],
//     candidate 0 carries d0 = R(lt[j][j] r) - R(lt[j][j] (r - 1)) - [j == 0] R(lt[0][0]), a multiple of u.
//   * k_vit_wide_fix: the exact chain = the four-wave step of k_viterbi_wide over a dynamic sequence of
//     32-position blocks; inside a usable chunk it compares its vector with the recorded row at the first recorded
//     position >= 12 steps into the block (V - W one constant over the live states, every value inside the binade
//     now and -- by the chunk's P0 gain -- up to the landing position) and jumps to the next rounding tie or the
//     chunk end with V = W + delta.  Every wave holds the whole vector, so the four waves take the decision
//     redundantly and no extra barrier is needed.
// Rounding ties (b or a ratio product exactly between two grid points) end a speculative segment as in
// tehmm_spec.hip.h; the chain lands exactly on them.
// ==========================================================================================================
#define TEHMM_WIDE_MAXTT 8          // transition-table entries per binade that may be rounding ties

__device__ __forceinline__ double wide_readlane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// log emission rows of every position, [internal position][128] (pads 0); one wave per chunk
__global__ __launch_bounds__(256) void k_wide_logrows(IntervalTab iv, EmisTab em, VitChunks vc, int N, double *BL) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= vc.n) return;
  const int id = vc.iv[c];
  const int64_t T = iv.len[id], p0 = iv.pos0[id], t0 = vc.t0[c];
  const int len = (int)min((int64_t)vc.CS, T - t0);
  for (int s = 0; s < len; ++s) {
    double x[2];
    emis_log_wide(em, em.tab, p0 + t0 + s, lane, N, x);
    double *dst = BL + (p0 + t0 + s) * TEHMM_WIDE_S;
    dst[lane] = lane < N ? x[0] : 0.0;
    dst[lane + 64] = lane + 64 < N ? x[1] : 0.0;
  }
}

// block = 512: eight chunks of one binade (wk_c [8 per workgroup], -1 = none; wk_e [workgroup]); QUANT = false:
// plain fp64 gains of every listed chunk (P0).  LDS: table [N][128] | flag
//
// Rounding ties.  Where an addend of the reference's sum (b, lt[j][j] r, lt[j][j] (r - 1)) lies exactly between two
// grid points, fl(v + z) depends on the PARITY of v / u -- candidate by candidate, because the reference adds b and
// the ratio term inside the max (_hmm.pyx:232-244).  At 100 states and with segment ratios such positions are
// frequent (a tie is a property of the VALUE: one in ~2^(e-5) per state and position for b, and the few hundred
// distinct products lt[j][j] r recur all along the data), so ending a segment at each of them -- as the narrow
// kernels do -- leaves the chain walking most of a chunk.  Instead the pass carries BOTH hypotheses about the
// constant delta between its frame and the true values, delta / u even (h = 0) and odd (h = 1): under a hypothesis
// the parity of a true value is known from the speculative one, every candidate's sum is rounded half-to-even on
// its own parity, and the recurrence stays exact THROUGH the tie.  Two vectors, two sets of recorded rows and of
// traceback bytes (the second set written from the chunk's first tie on: before it the hypotheses agree); the chain
// learns delta when it verifies the chunk and adopts the set whose parity matches.
template <bool QUANT, bool RATIO>
__global__ __launch_bounds__(512) void k_vit_wide_spec(IntervalTab iv, VitChunks vc, int N, int NP,
                                                       const double *__restrict__ g_lt, const double *__restrict__ BL,
                                                       const double *__restrict__ tratios, const int *__restrict__ wk_c,
                                                       const int *__restrict__ wk_e, int TBW, uint8_t *tb, uint8_t *tb2,
                                                       double *rows2, int WU, double *pre, int *ready = nullptr) {
  extern __shared__ double wsm[];
  double *tq = wsm;
  volatile int *tflag = (volatile int *)(wsm + (size_t)N * TEHMM_WIDE_S);
  const int e = QUANT ? wk_e[blockIdx.x] : 0;
  const double u = QUANT ? ldexp(1.0, e - 52) : 0.0;
  const double M = QUANT ? ldexp(1.5, e) : 0.0;                 // fl(z + M) - M rounds z to the grid u
  const double half_u = 0.5 * u;
  const double CM = QUANT ? ldexp(1.0, e + 1) - ldexp(1.0, e - 44) : 0.0;          // 2^(e+1) - 256 u
  const double wlim = QUANT ? -(ldexp(1.0, e) - ldexp(1.5, e - 44)) : -INFINITY;   // -(2^e - 384 u)
  const double zlim = QUANT ? ldexp(1.0, e - 1) : INFINITY;
  const double i2u = QUANT ? ldexp(1.0, 51 - e) : 0.0;                             // 1 / (2 u)
  // LDS behind the table: flag | number of tie entries | tie entries (from | to << 8 | parity of the lower neighbour << 16)
  volatile int *ntt = tflag + 1;
  volatile int *ttab = tflag + 2;
  int *hbig = (int *)(tflag + 2 + TEHMM_WIDE_MAXTT);             // [4]: states with a "never" transition into them
  if (threadIdx.x == 0) { *tflag = 0; *ntt = 0; hbig[0] = hbig[1] = hbig[2] = hbig[3] = 0; }
  __syncthreads();
  const double zhuge = QUANT ? -ldexp(1.0, e + 1) : -INFINITY;
  for (int i = threadIdx.x; i < N * TEHMM_WIDE_S; i += blockDim.x) {
    const int f = i >> 7, j = i & 127;
    double z = j < N ? g_lt[(size_t)f * NP + j] : -INFINITY;
    if (QUANT && z > -INFINITY && z <= zhuge) {
      // a "never" transition (the reference's log(0) = -1e100): V[f] + z lies below everything a candidate with an
      // ordinary transition reaches from inside the binade (V <= -2^e, z <= -2^(e+1), ordinary |z'| < 2^(e-1)), so it
      // counts as -inf here -- as long as the state has such a candidate from a live state: a state that comes out
      // at -inf although it has "never" transitions spoils the chunk (finish)
      atomicOr(&hbig[j >> 5], 1 << (j & 31));
      z = -INFINITY;
    }
    double val = z;
    if (QUANT) {
      double q = (z + M) - M;
      if (z > -INFINITY && !(fabs(z) < zlim)) *tflag = 1;       // out of the range of the rounding trick
      if (z > -INFINITY && fabs(z - q) == half_u) {
        // a transition exactly between two grid points: fl(V[f] + z) depends on the parity of V[f] / u.  The table
        // holds the LOWER neighbour; the step adds u where round-half-even goes up (k_vit_wide_spec, tie_adj)
        const double qo = 2.0 * z - q;                          // the odd neighbour (q / u is even)
        const int k = atomicAdd((int *)ntt, 1);
        if (k < TEHMM_WIDE_MAXTT) ttab[k] = f | (j << 8) | ((qo < q ? 1 : 0) << 16);
        else *tflag = 1;
        q = fmin(q, qo);
      }
      val = 128.0 * q + (double)(127 - f) * u;                  // -inf stays -inf
    }
    tq[i] = val;
  }
  __syncthreads();
  const int n_tt = QUANT ? min((int)*ntt, TEHMM_WIDE_MAXTT) : 0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = wk_c[blockIdx.x * 8 + w];
  if (c < 0) return;
  const int id = vc.iv[c];
  const int64_t p0 = iv.pos0[id], t0 = vc.t0[c];
  const int CS = vc.CS;
  const int len = (int)min((int64_t)CS, iv.len[id] - t0);     // (P0 also runs ragged tails; P2 only full chunks)
  const bool live0 = lane < N, live1 = lane + 64 < N;
  const bool big0 = QUANT && ((hbig[lane >> 5] >> (lane & 31)) & 1) != 0;
  const bool big1 = QUANT && ((hbig[2 + (lane >> 5)] >> (lane & 31)) & 1) != 0;
  constexpr int H = QUANT ? 2 : 1;
  double W0[H], W1[H], base[H];          // QUANT: 128 x (value - base) of the lane's two states, per hypothesis
  int pb[H];                             // parity of base / u, XOR the hypothesis
#pragma unroll
  for (int h = 0; h < H; ++h) {
    W0[h] = live0 ? 0.0 : -INFINITY;
    W1[h] = live1 ? 0.0 : -INFINITY;
    base[h] = 0.0;
    pb[h] = h;
  }
  // a tie has been met: the two hypotheses are computed separately from here (from the start when the binade's
  // transition table itself has tie entries: their rounding follows the hypothesis at every step)
  int nt = 0, first_tie = n_tt > 0 ? 0 : CS;
  bool diverged = n_tt > 0;
  bool bad = QUANT && *tflag != 0;
  const double ltd0 = live0 ? g_lt[(size_t)lane * NP + lane] : 0.0;
  const double ltd1 = live1 ? g_lt[(size_t)(lane + 64) * NP + lane + 64] : 0.0;
  const double lt00 = g_lt[0];
  const double lt00q = QUANT ? (lt00 + M) - M : lt00;
  if (QUANT && RATIO && (fabs(lt00 - lt00q) == half_u || !(fabs(lt00) < zlim))) bad = true;
  // QUANT: the pass starts WU positions BEFORE its chunk (a speculated chunk is never an interval's first), so that
  // by the chunk's first position it has forgotten its zero start: `pre` [hypothesis][chunk][NP] keeps the vector
  // at position t0 - 1, against which the chain can verify the chunk without walking a single step of it
  const int s_first = QUANT ? -WU : 0;
  const double *bp = BL + (p0 + t0) * TEHMM_WIDE_S;
  double bn0 = bp[(int64_t)s_first * TEHMM_WIDE_S + lane], bn1 = bp[(int64_t)s_first * TEHMM_WIDE_S + lane + 64];
  const int n0 = min(N, 64);
  for (int s = s_first; s < len; ++s) {
    const int64_t t = t0 + s;
    const double b0 = bn0, b1 = bn1;
    if (s + 1 < len) {
      bn0 = bp[(int64_t)(s + 1) * TEHMM_WIDE_S + lane];
      bn1 = bp[(int64_t)(s + 1) * TEHMM_WIDE_S + lane + 64];
    }
    // ---- segment-ratio terms of this position
    double add0 = 0.0, add1 = 0.0;       // what every candidate from >= 1 gets (QUANT: grid-rounded, not yet x 128)
    double dz0 = 0.0, dz1 = 0.0;         // what candidate 0 gets on top of that
    double za0 = 0.0, za1 = 0.0, zb0 = 0.0, zb1 = 0.0, qa0 = 0.0, qa1 = 0.0, qb0 = 0.0, qb1 = 0.0;
    bool tza0 = false, tza1 = false, tzb0 = false, tzb1 = false, rg = false;
    if (RATIO) {
      const double r = tratios[p0 + t];
      rg = r > 1.0;
      za0 = ltd0 * r; za1 = ltd1 * r;                          // from == 0: lt[j][j] * r, always
      zb0 = ltd0 * (r - 1.0); zb1 = ltd1 * (r - 1.0);          // from >= 1: lt[j][j] * (r - 1) if r > 1
      if (QUANT) {
        qa0 = (za0 + M) - M; qa1 = (za1 + M) - M;
        qb0 = rg ? (zb0 + M) - M : 0.0; qb1 = rg ? (zb1 + M) - M : 0.0;
        tza0 = live0 && fabs(za0 - qa0) == half_u;
        tza1 = live1 && fabs(za1 - qa1) == half_u;
        tzb0 = live0 && rg && fabs(zb0 - qb0) == half_u;
        tzb1 = live1 && rg && fabs(zb1 - qb1) == half_u;
        bad = bad || (live0 && !(fabs(za0) < zlim)) || (live1 && !(fabs(za1) < zlim));
        add0 = qb0; add1 = qb1;
        dz0 = (qa0 - qb0) - (lane == 0 ? lt00q : 0.0);
        dz1 = qa1 - qb1;
      } else {
        add0 = rg ? zb0 : 0.0; add1 = rg ? zb1 : 0.0;
        dz0 = (za0 - add0) - (lane == 0 ? lt00 : 0.0);
        dz1 = za1 - add1;
      }
    }
    const double bq0 = QUANT ? (b0 + M) - M : 0.0, bq1 = QUANT ? (b1 + M) - M : 0.0;
    const bool tb0 = QUANT && live0 && fabs(b0 - bq0) == half_u, tb1 = QUANT && live1 && fabs(b1 - bq1) == half_u;
    const bool anytie = QUANT && __any(tb0 || tb1 || tza0 || tza1 || tzb0 || tzb1);
    if (QUANT) bad = bad | (b0 != b0);
    // ---- transitions that are rounding ties in this binade (rare: a handful per binade at most): the table holds
    // the lower neighbour of fl(V[f] + lt); round-half-even takes the upper one when the lower one is odd, i.e. when
    // parity(V[f] / u) differs from the parity of the entry's lower neighbour.  +128 u for that (from, to), else 0.
    auto tie_adj = [&](auto hc, int k, double &a0, double &a1) {
      constexpr int h = decltype(hc)::value;
      const int ent = ttab[k];
      const int f = ent & 255, j = (ent >> 8) & 255, pz = (ent >> 16) & 1;
      const double wf = f < 64 ? wide_readlane(W0[h], f) : wide_readlane(W1[h], f - 64);
      // V[f] = W[f] / 128 + base: W[f] is a multiple of 128 u, its parity bit is bit 7 of W[f] / u
      const int pv = (int)(((unsigned)__double2loint(wf + CM) >> 7) & 1u) ^ pb[h];
      const bool up = wf > -INFINITY && (pv ^ pz) != 0;
      a0 = (up && j == lane) ? 128.0 * u : 0.0;
      a1 = (up && j == lane + 64) ? 128.0 * u : 0.0;
      return f;
    };
    // ---- one hypothesis, no tie at this position: x[j] = max_f W[f] + table[f][j] (the table carries 127 - f in its
    // low bits: the first arg-max wins), then the grid-rounded addends
    auto fast = [&](auto hc) {
      constexpr int h = decltype(hc)::value;
      const double s0 = wide_readlane(W0[h], 0);
      double x0 = s0 + tq[lane], x1 = s0 + tq[lane + 64];
      if (RATIO) {
        x0 += QUANT ? 128.0 * dz0 : dz0;
        x1 += QUANT ? 128.0 * dz1 : dz1;
      }
      // eight from-states at a time: their sixteen table entries are requested before the first is used (left as
      // a rolled loop every iteration waits for its own two LDS reads: 11 us per step)
      auto oct = [&](const double wreg, int fb, int lb) {
        double ta[8], tc[8], sv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          ta[i] = tq[(fb + i) * TEHMM_WIDE_S + lane];
          tc[i] = tq[(fb + i) * TEHMM_WIDE_S + lane + 64];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) sv[i] = wide_readlane(wreg, lb + i);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          x0 = fmax(x0, sv[i] + ta[i]);
          x1 = fmax(x1, sv[i] + tc[i]);
        }
      };
      int f = 1;
      for (; f + 8 <= n0; f += 8) oct(W0[h], f, f);
      for (; f < n0; ++f) {
        const double sv = wide_readlane(W0[h], f);
        x0 = fmax(x0, sv + tq[f * TEHMM_WIDE_S + lane]);
        x1 = fmax(x1, sv + tq[f * TEHMM_WIDE_S + lane + 64]);
      }
      f = 64;
      for (; f + 8 <= N; f += 8) oct(W1[h], f, f - 64);
      for (; f < N; ++f) {
        const double sv = wide_readlane(W1[h], f - 64);
        x0 = fmax(x0, sv + tq[f * TEHMM_WIDE_S + lane]);
        x1 = fmax(x1, sv + tq[f * TEHMM_WIDE_S + lane + 64]);
      }
      if (QUANT) {
        for (int k = 0; k < n_tt; ++k) {      // (the loop above already holds these candidates at the lower neighbour)
          double a0, a1;
          const int f = tie_adj(hc, k, a0, a1);
          const double sv = f < 64 ? wide_readlane(W0[h], f) : wide_readlane(W1[h], f - 64);
          double c0 = sv + tq[f * TEHMM_WIDE_S + lane] + a0, c1 = sv + tq[f * TEHMM_WIDE_S + lane + 64] + a1;
          if (RATIO && f == 0) {
            c0 += 128.0 * dz0;
            c1 += 128.0 * dz1;
          }
          x0 = fmax(x0, c0);
          x1 = fmax(x1, c1);
        }
        W0[h] = x0 + 128.0 * (bq0 + add0);    // (multiples of 128 u: the index bits ride along)
        W1[h] = x1 + 128.0 * (bq1 + add1);
      } else {
        W0[h] = (x0 + b0) + add0;
        W1[h] = (x1 + b1) + add1;
      }
    };
    // ---- both hypotheses, no tie at this position: one pass over the table for the two vectors (the pass is bound
    // by LDS bandwidth: 100 KB of table per step and wave)
    auto fast_both = [&]() {
      const double sa = wide_readlane(W0[0], 0), sb = wide_readlane(W0[H - 1], 0);
      const double t00 = tq[lane], t01 = tq[lane + 64];
      double xa0 = sa + t00, xa1 = sa + t01, xb0 = sb + t00, xb1 = sb + t01;
      if (RATIO) {
        xa0 += 128.0 * dz0; xa1 += 128.0 * dz1;
        xb0 += 128.0 * dz0; xb1 += 128.0 * dz1;
      }
      auto oct = [&](const double wa, const double wb, int fb, int lb) {
        double ta[8], tc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          ta[i] = tq[(fb + i) * TEHMM_WIDE_S + lane];
          tc[i] = tq[(fb + i) * TEHMM_WIDE_S + lane + 64];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const double va = wide_readlane(wa, lb + i), vb = wide_readlane(wb, lb + i);
          xa0 = fmax(xa0, va + ta[i]);
          xa1 = fmax(xa1, va + tc[i]);
          xb0 = fmax(xb0, vb + ta[i]);
          xb1 = fmax(xb1, vb + tc[i]);
        }
      };
      auto one = [&](const double wa, const double wb, int f, int l) {
        const double ta = tq[f * TEHMM_WIDE_S + lane], tc = tq[f * TEHMM_WIDE_S + lane + 64];
        const double va = wide_readlane(wa, l), vb = wide_readlane(wb, l);
        xa0 = fmax(xa0, va + ta);
        xa1 = fmax(xa1, va + tc);
        xb0 = fmax(xb0, vb + ta);
        xb1 = fmax(xb1, vb + tc);
      };
      int f = 1;
      for (; f + 8 <= n0; f += 8) oct(W0[0], W0[H - 1], f, f);
      for (; f < n0; ++f) one(W0[0], W0[H - 1], f, f);
      f = 64;
      for (; f + 8 <= N; f += 8) oct(W1[0], W1[H - 1], f, f - 64);
      for (; f < N; ++f) one(W1[0], W1[H - 1], f, f - 64);
      W0[0] = xa0 + 128.0 * (bq0 + add0);
      W1[0] = xa1 + 128.0 * (bq1 + add1);
      W0[H - 1] = xb0 + 128.0 * (bq0 + add0);
      W1[H - 1] = xb1 + 128.0 * (bq1 + add1);
    };
    // ---- one hypothesis at a tie position: every candidate rounded half-to-even on its own parity
    auto slow = [&](auto hc) {
      constexpr int h = decltype(hc)::value;
      const int pbq0 = __builtin_amdgcn_fract(bq0 * i2u) != 0.0, pbq1 = __builtin_amdgcn_fract(bq1 * i2u) != 0.0;
      const double bo0 = 2.0 * b0 - bq0, bo1 = 2.0 * b1 - bq1;         // the other neighbour of a tied addend
      const double qao0 = 2.0 * za0 - qa0, qao1 = 2.0 * za1 - qa1, qbo0 = 2.0 * zb0 - qb0, qbo1 = 2.0 * zb1 - qb1;
      const int ph = pb[h];
      double x0 = -INFINITY, x1 = -INFINITY;
      // (the addends selected below are captured BY VALUE: selected through by-reference captures the compiler kept
      //  pointers to them in scratch and loaded the chosen one through a flat pointer -- 328 bytes of scratch per lane)
      auto cand = [&, bo0, bo1, bq0, bq1, qa0, qa1, qao0, qao1, qb0, qb1, qbo0, qbo1, lt00q, rg, ph, pbq0, pbq1,
                   tb0, tb1, tza0, tza1, tzb0, tzb1](double sv, int f) {
        double c0 = sv + tq[f * TEHMM_WIDE_S + lane], c1 = sv + tq[f * TEHMM_WIDE_S + lane + 64];
        for (int k = 0; k < n_tt; ++k)
          if ((ttab[k] & 255) == f) {
            double a0, a1;
            (void)tie_adj(hc, k, a0, a1);
            c0 += a0;
            c1 += a1;
          }
        // parity of (V[f] + lt) / u: bit 7 of the candidate's integer (its low seven bits are the index)
        const int p0a = (int)(((unsigned)__double2loint(c0 + CM) >> 7) & 1u) ^ ph;
        const int p1a = (int)(((unsigned)__double2loint(c1 + CM) >> 7) & 1u) ^ ph;
        const double be0 = (tb0 && p0a) ? bo0 : bq0, be1 = (tb1 && p1a) ? bo1 : bq1;
        const int p0b = tb0 ? 0 : (p0a ^ pbq0), p1b = tb1 ? 0 : (p1a ^ pbq1);     // a tied sum comes out even
        double z0 = 0.0, z1 = 0.0;
        if (RATIO) {
          if (f == 0) {
            z0 = ((tza0 && p0b) ? qao0 : qa0) - (lane == 0 ? lt00q : 0.0);
            z1 = (tza1 && p1b) ? qao1 : qa1;
          } else {
            z0 = rg ? ((tzb0 && p0b) ? qbo0 : qb0) : 0.0;
            z1 = rg ? ((tzb1 && p1b) ? qbo1 : qb1) : 0.0;
          }
        }
        x0 = fmax(x0, c0 + 128.0 * (be0 + z0));
        x1 = fmax(x1, c1 + 128.0 * (be1 + z1));
      };
      for (int f = 0; f < n0; ++f) cand(wide_readlane(W0[h], f), f);
      for (int f = 64; f < N; ++f) cand(wide_readlane(W1[h], f - 64), f);
      W0[h] = x0;
      W1[h] = x1;
    };
    // ---- index extraction, traceback bytes, range check of one hypothesis
    auto finish = [&](auto hc, uint8_t *tbo) {
      constexpr int h = decltype(hc)::value;
      const bool wr = s >= 0;
      // W = 128 (new value - base) + (127 - first arg-max) u, exact;  W + CM has exponent e, its low seven mantissa
      // bits are W / u mod 128
      const double x0 = W0[h], x1 = W1[h];
      const double y0 = x0 + CM, y1 = x1 + CM;
      const unsigned l0 = (unsigned)__double2loint(y0), l1 = (unsigned)__double2loint(y1);
      W0[h] = x0 > -INFINITY ? __hiloint2double(__double2hiint(y0), (int)(l0 & ~127u)) - CM : -INFINITY;
      W1[h] = x1 > -INFINITY ? __hiloint2double(__double2hiint(y1), (int)(l1 & ~127u)) - CM : -INFINITY;
      if (wr && live0) tbo[(p0 + t) * TBW + lane] = x0 > -INFINITY ? (uint8_t)(~l0 & 127u) : (uint8_t)0;
      if (wr && live1) tbo[(p0 + t) * TBW + lane + 64] = x1 > -INFINITY ? (uint8_t)(~l1 & 127u) : (uint8_t)0;
      bad = bad | (live0 && W0[h] <= wlim && W0[h] > -INFINITY) | (live1 && W1[h] <= wlim && W1[h] > -INFINITY);
      bad = bad | (big0 && !(x0 > -INFINITY)) | (big1 && !(x1 > -INFINITY));
    };
    // re-base (every 8 steps: at 128 x the range holds 2^(e-7) log units, a position costs ~25) and record
    auto rebase = [&](auto hc) {
      constexpr int h = decltype(hc)::value;
      {
        const double mx = wave_max_f64(fmax(live0 ? W0[h] : -INFINITY, live1 ? W1[h] : -INFINITY));
        if (mx > -INFINITY) {
          W0[h] -= mx;
          W1[h] -= mx;
          base[h] += mx * 0.0078125;
          pb[h] ^= __builtin_amdgcn_fract(mx * ldexp(1.0, 44 - e)) != 0.0;     // parity of (mx / 128) / u
        } else {
          bad = true;                                   // the whole vector died
        }
      }
    };
    auto record = [&](auto hc, double *rows) {
      constexpr int h = decltype(hc)::value;
      double *row = rows + ((int64_t)c * (CS / TEHMM_VROW) + s / TEHMM_VROW) * NP;
      row[lane] = live0 ? W0[h] * 0.0078125 + base[h] : -INFINITY;
      if (lane + 64 < NP) row[lane + 64] = live1 ? W1[h] * 0.0078125 + base[h] : -INFINITY;
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, H - 1>;
    if (!QUANT) {
      fast(H0{});
      continue;
    }
    if (anytie) {
      first_tie = min(first_tie, max(s, 0));
      nt += s >= 0;
      slow(H0{});
      slow(H1{});
      diverged = true;
    } else if (diverged && n_tt == 0) {
      fast_both();
    } else {
      fast(H0{});
      if (diverged) fast(H1{});
    }
    finish(H0{}, tb);
    if (diverged) finish(H1{}, tb2);
    if ((s & 7) == 7) {
      rebase(H0{});
      if (diverged) rebase(H1{});
    }
    if (s >= 0 && (s & (TEHMM_VROW - 1)) == TEHMM_VROW - 1) {
      record(H0{}, vc.rows);
      if (diverged) record(H1{}, rows2);
    }
    if (!diverged) {                      // the hypotheses still agree: h = 1 is h = 0 with the other parity
      W0[H - 1] = W0[0];
      W1[H - 1] = W1[0];
      base[H - 1] = base[0];
      pb[H - 1] = pb[0] ^ 1;
    }
    if (s == -1) {                        // the vector the chunk is entered with, both hypotheses
#pragma unroll
      for (int h = 0; h < H; ++h) {
        double *row = pre + ((int64_t)h * vc.n + c) * NP;
        row[lane] = live0 ? W0[h] * 0.0078125 + base[h] : -INFINITY;
        if (lane + 64 < NP) row[lane + 64] = live1 ? W1[h] * 0.0078125 + base[h] : -INFINITY;
      }
    }
    if (!diverged) {
      if (s >= 0 && (s & (TEHMM_VROW - 1)) == TEHMM_VROW - 1) {
        double *row = rows2 + ((int64_t)c * (CS / TEHMM_VROW) + s / TEHMM_VROW) * NP;
        row[lane] = live0 ? W0[0] * 0.0078125 + base[0] : -INFINITY;
        if (lane + 64 < NP) row[lane + 64] = live1 ? W1[0] * 0.0078125 + base[0] : -INFINITY;
      }
    }
  }
  if (QUANT) {
    const unsigned long long anybad = __ballot(bad);
    if (lane == 0) {
      vc.ntie[c] = nt;
      vc.ties[(int64_t)c * TEHMM_SPEC_MAXT] = first_tie;      // where the second set of traceback bytes starts
      vc.ok[c] = anybad ? 0 : 1;
    }
    if (ready) {
      // the exact chain may be waiting for this chunk (k_vit_wide_fix, `ready`): everything the wave wrote -- rows, pre,
      // traceback bytes, ok -- is visible before the flag is
      __threadfence();
      if (lane == 0) __hip_atomic_store(&ready[c], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    const double g = wave_max_f64(fmax(live0 ? W0[0] : -INFINITY, live1 ? W1[0] : -INFINITY));
    if (lane == 0) vc.gain[c] = g;
  }
}

// ------------------------------------------------------------------------------------------
// P0 in packed floats: the gain of every chunk (max of the vector after the chunk, started from zeros) only places the
// chunk in its binade -- spec_assign_binades keeps 512 + 2e-5 |score| of margin -- so single precision does, and the
// lane's two states share one v_pk_add_f32 / v_pk_max_f32 per from-state.  The table sits in LDS as float2
// [from][lane] = (to lane, to lane + 64): 51 KB at 100 states, two workgroups per CU.  Every 32 positions the vector
// is re-based on its maximum (accumulated in fp64), so the floats stay below ~10^3.  One wave per chunk, ragged
// tails included; a NaN row (nothing can emit) makes the gain NaN.
// ------------------------------------------------------------------------------------------
template <bool RATIO>
__global__ __launch_bounds__(512) void k_vit_wide_gain(IntervalTab iv, VitChunks vc, int N, int NP,
                                                       const double *__restrict__ g_lt, const double *__restrict__ BL,
                                                       const double *__restrict__ tratios) {
  extern __shared__ double wsm[];
  float2 *tqf = (float2 *)wsm;
  for (int i = threadIdx.x; i < N * 64; i += blockDim.x) {
    const int f = i >> 6, j = i & 63;
    tqf[i] = make_float2(j < N ? (float)g_lt[(size_t)f * NP + j] : -INFINITY,
                         j + 64 < N ? (float)g_lt[(size_t)f * NP + j + 64] : -INFINITY);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 8 + w;
  if (c >= vc.n) return;
  const int id = vc.iv[c];
  const int64_t p0 = iv.pos0[id], t0 = vc.t0[c];
  const int len = (int)min((int64_t)vc.CS, iv.len[id] - t0);
  const bool live0 = lane < N, live1 = lane + 64 < N;
  lane_f2 W = (lane_f2){live0 ? 0.f : -INFINITY, live1 ? 0.f : -INFINITY};
  const float ltd0 = live0 ? (float)g_lt[(size_t)lane * NP + lane] : 0.f;
  const float ltd1 = live1 ? (float)g_lt[(size_t)(lane + 64) * NP + lane + 64] : 0.f;
  const float lt00 = (float)g_lt[0];
  const int n0 = min(N, 64);
  const double *bp = BL + (p0 + t0) * TEHMM_WIDE_S;
  double bn0 = bp[lane], bn1 = bp[lane + 64];
  double acc = 0.0;
  bool bad = false;
  const lane_f2 *tp = (const lane_f2 *)tqf + lane;
  for (int s = 0; s < len; ++s) {
    const double b0 = bn0, b1 = bn1;
    if (s + 1 < len) {
      bn0 = bp[(int64_t)(s + 1) * TEHMM_WIDE_S + lane];
      bn1 = bp[(int64_t)(s + 1) * TEHMM_WIDE_S + lane + 64];
    }
    bad = bad | (live0 && b0 != b0) | (live1 && b1 != b1);
    lane_f2 add = (lane_f2){0.f, 0.f}, dz = (lane_f2){0.f, 0.f};
    if (RATIO) {
      // _hmm.pyx:201-259 with segment ratios: the candidate from state 0 gets lt[j][j] r, the others lt[j][j] (r - 1)
      // when r > 1 (and state 0's own column loses lt[0][0]): see k_vit_wide_spec
      const float r = (float)tratios[p0 + t0 + s];
      const lane_f2 za = (lane_f2){ltd0, ltd1} * (lane_f2){r, r};
      if (r > 1.f) add = (lane_f2){ltd0, ltd1} * (lane_f2){r - 1.f, r - 1.f};
      dz = za - add;
      if (lane == 0) dz.x -= lt00;
    }
    const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, W.x), 0));
    lane_f2 x = (lane_f2){s0, s0} + tp[0] + dz;
    auto oct = [&](const float wreg, int fb, int lb) {
      lane_f2 ta[8];
      float sv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) ta[i] = tp[(fb + i) * 64];
#pragma unroll
      for (int i = 0; i < 8; ++i) sv[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wreg), lb + i));
#pragma unroll
      for (int i = 0; i < 8; ++i) x = __builtin_elementwise_max(x, (lane_f2){sv[i], sv[i]} + ta[i]);
    };
    int f = 1;
    for (; f + 8 <= n0; f += 8) oct(W.x, f, f);
    for (; f < n0; ++f) {
      const float sv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, W.x), f));
      x = __builtin_elementwise_max(x, (lane_f2){sv, sv} + tp[f * 64]);
    }
    f = 64;
    for (; f + 8 <= N; f += 8) oct(W.y, f, f - 64);
    for (; f < N; ++f) {
      const float sv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, W.y), f - 64));
      x = __builtin_elementwise_max(x, (lane_f2){sv, sv} + tp[f * 64]);
    }
    W = x + (lane_f2){(float)b0, (float)b1} + add;
    if (!live0) W.x = -INFINITY;
    if (!live1) W.y = -INFINITY;
    if ((s & 31) == 31) {
      const float mx = (float)wave_max_f64((double)fmaxf(W.x, W.y));
      if (mx > -INFINITY) {
        W = W - (lane_f2){mx, mx};
        acc += (double)mx;
      }
    }
  }
  const double g = acc + wave_max_f64((double)fmaxf(W.x, W.y));
  const bool anybad = __ballot(bad) != 0ull;
  if (lane == 0) vc.gain[c] = anybad ? __longlong_as_double(0x7ff8000000000000LL) : g;
}

// ------------------------------------------------------------------------------------------
// Exact chain with verified jumps.  One four-wave workgroup per interval; the step is k_viterbi_wide's (wave w owns
// the from-states [w NP / 4, (w + 1) NP / 4), partial maxima meet in LDS behind one barrier per step, every wave
// combines them itself), the blocks are 16 positions (TEHMM_PB) and their sequence is dynamic.  Once the interval's
// first emittable row is behind it (the leading-rows quirk, _emission.pyx:73-80, is the block kernel's business) the
// chain takes its emission rows from the log-row buffer of k_wide_logrows -- recomputing them cost half of a step.
// A verified chunk is adopted to its end with the hypothesis (see k_vit_wide_spec) whose parity matches delta:
// sel_hyp [chunk] = that hypothesis, sel_from [chunk] = the first adopted position (k_wide_tb_select then moves the
// second set of traceback bytes in where h = 1 was taken).
// ------------------------------------------------------------------------------------------
template <bool RATIO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_vit_wide_fix(IntervalTab iv, EmisTab em, VitChunks vc, int N, int NP, const double *g_lt, const double *g_pi,
                    const double *tratios, int TBW, uint8_t *tb, int *last_state, double *logprob, int *stats,
                    const double *rows2, const double *__restrict__ BL, int64_t *sel_from, int *sel_hyp,
                    const double *pre, int phase, const int64_t *__restrict__ hstop, double *hvec, int *hflag,
                    const int *ready = nullptr, int spin_limit = 1 << 19) {
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
  const int64_t cfirst = vc.first[id];
  const int CS = vc.CS;
  int n_block = 0, n_jump = 0;
  double vfin[2] = {-INFINITY, -INFINITY};          // this lane's two states of the current vector
  int64_t t0 = 0, since = 0;                        // block start; where the exact run (re)started
  // phase 1 = the interval's head only: positions [0, hstop) -- everything before its first speculated chunk, which no
  // quantised pass helps with -- walked while that pass is still running (another stream); the vector at hstop - 1
  // goes to hvec.  phase 2 resumes there.  phase 0: the whole interval in one launch.
  // ready != nullptr: the quantised pass may still be RUNNING (it was launched before this kernel, on another stream,
  // and sets ready[c] behind everything else it writes for chunk c).  Before the chain looks at a speculated chunk it
  // waits for that flag: thread 0 polls, the workgroup meets at a barrier, every thread then fences (acquire, agent
  // scope) so that no stale cache line answers for the chunk's rows.  The pass does not depend on the chain and the
  // host launches this form only when the chain's workgroups cannot fill the GPU, so the wait always ends; the spin
  // budget (the host's estimate of the pass, tenfold) is a safety net behind which the chunk is simply walked exactly -- its quantised traceback bytes
  // live in their own buffers (k_wide_tb_select), so a late writer cannot disturb the exact ones.
  __shared__ int s_rdy;
  int64_t lastready = -1;
  int spin_budget = spin_limit;      // (thread 0) polls left for the whole interval, ~2.7 us each: the host allows about ten
                                     // times what the pass should need (under a profiler that serialises kernels the pass
                                     // cannot run beside the chain at all: the budget is what that costs)
  auto wait_chunk = [&](int64_t c) -> bool {
    if (!ready || lastready == c) return true;
    __syncthreads();
    if (threadIdx.x == 0) {
      int ok = 0;
      for (;;) {
        ok = __hip_atomic_load(&ready[c], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (ok || spin_budget <= 0) break;
        --spin_budget;
        __builtin_amdgcn_s_sleep(64);
      }
      s_rdy = ok;
    }
    __syncthreads();
    const int ok = s_rdy;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (ok) lastready = c;
    return ok != 0;
  };
  const int64_t hs = phase != 0 ? min(T, hstop[id]) : 0;
  const int64_t Tend = phase == 1 ? hs : T;
  if (phase == 2 && hs > 0 && hflag[id] != 0) {
    double *vj = vmine + cur * W;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int j = lane + 64 * s;
      vfin[s] = j < N ? hvec[(size_t)id * NP + j] : -INFINITY;
      vj[j] = vfin[s];
    }
    seen = true;
    t0 = since = hs;
  }
  double bnx[2] = {0.0, 0.0};                       // log row of position tnx, requested a step ahead
  int64_t tnx = -1;
  // Does the current vector (vfin, position `at`) equal a recorded row of chunk c's quantised pass up to ONE constant
  // -- for the hypothesis whose parity that constant has --, is every value in the chunk's binade and (by the chunk's
  // P0 gain) does it stay there up to the chunk end?  Then the chunk is adopted to its end: V = last row + constant
  // (exact: multiples of u inside one binade).  Every wave holds the whole vector: all four decide alike.
  auto adopt = [&](int64_t c, int e, const double (&wrow)[2][2], const double (&wend)[2][2], double span, int64_t from) {
    double vmn = INFINITY;
    bool inb = true;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int j = lane + 64 * s;
      if (j < N && vfin[s] > -INFINITY) {
        vmn = fmin(vmn, vfin[s]);
        inb = inb && exp_of(-vfin[s]) == e + 1;            // 2^e <= |v| < 2^(e+1)
      }
    }
    const double vlow = wave_min_f64(vmn) - span;
    const bool common = __all(inb) && vlow > -INFINITY && exp_of(-vlow) == e + 1;
    int take = -1;
    double dtake = 0.0;
    const double i2u = ldexp(1.0, 51 - e);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double dmx = -INFINITY;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int j = lane + 64 * s;
        const bool both_dead = vfin[s] == -INFINITY && wrow[h][s] == -INFINITY;
        if (j < N && !both_dead) dmx = fmax(dmx, vfin[s] - wrow[h][s]);
      }
      const double d0 = wave_max_f64(dmx);
      bool okl = true;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int j = lane + 64 * s;
        const bool both_dead = vfin[s] == -INFINITY && wrow[h][s] == -INFINITY;
        if (j < N && !both_dead) okl = okl && (vfin[s] - wrow[h][s]) == d0;
      }
      const bool fin = d0 == d0 && d0 > -INFINITY && d0 < INFINITY;
      const bool odd = fin && __builtin_amdgcn_fract(d0 * i2u) != 0.0;
      if (take < 0 && fin && __all(okl) && (int)odd == h) {
        take = h;
        dtake = d0;
      }
    }
    if (!(common && take >= 0)) {
#ifdef TEHMM_WIDE_DEBUG
      if (threadIdx.x == 0 && stats) {
        atomicAdd(&stats[2], 1);
        if (!__all(inb)) atomicAdd(&stats[3], 1);
        else if (!common) atomicAdd(&stats[4], 1);
        if (take < 0) atomicAdd(&stats[5], 1);
      }
#endif
      return false;
    }
    double *vj = vmine + cur * W;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int j = lane + 64 * s;
      vfin[s] = j < N ? (take ? wend[1][s] : wend[0][s]) + dtake : -INFINITY;
      vj[j] = vfin[s];
    }
    if (threadIdx.x == 0) {
      sel_hyp[c] = take;
      sel_from[c] = from;
    }
    ++n_jump;
    return true;
  };
  while (t0 < Tend) {
    // ---- at a chunk's first position: the quantised pass warmed up before the chunk, its vector at t0 - 1 (`pre`)
    // can be compared with the chain's right away -- a verified chunk costs the chain no step at all
    if (phase != 1) {
      const int64_t c = cfirst + t0 / CS;
      const int64_t ct0 = vc.t0[c];
      const int e = vc.e[c];
      if (t0 == ct0 && t0 > 0 && e != TEHMM_SPEC_NONE && seen && wait_chunk(c) && vc.ok[c] != 0) {
        double wrow[2][2], wend[2][2];
        const int64_t eo = ((c * (CS / TEHMM_VROW)) + (CS / TEHMM_VROW - 1)) * NP;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int j = lane + 64 * s;
          wrow[0][s] = j < N ? pre[c * NP + j] : -INFINITY;
          wrow[1][s] = j < N ? pre[((int64_t)vc.n + c) * NP + j] : -INFINITY;
          wend[0][s] = j < N ? vc.rows[eo + j] : -INFINITY;
          wend[1][s] = j < N ? rows2[eo + j] : -INFINITY;
        }
        if (adopt(c, e, wrow, wend, fabs(vc.gain[c]) * 1.01 + 256.0, ct0)) {
          t0 = ct0 + CS;
          since = t0;
          continue;
        }
      }
    }
    ++n_block;
    const int np = (int)min((int64_t)TEHMM_PB, Tend - t0);
    const bool use_ring = !seen;                    // (uniform: every wave walks the leading rows alike)
    if (use_ring) {
      wide_emission_block<0, false>(em, p0 + t0, t0, np, lane, w, N, ltd, nullptr, seen, fg, ring, nullptr, ltab);
      __syncthreads();
    }
    // chunk of this block and whether it can be verified against the quantised pass.  A block of 16 positions holds
    // exactly one recorded row (positions = 15 mod 16 of the chunk): the chain checks there once it has run
    // TEHMM_FIX_MINSTEP exact steps since it (re)started (rank convergence takes 3-16 steps; a check that comes too
    // early simply fails and the next block tries again)
    const int64_t c = cfirst + t0 / CS;
    const int64_t ct0 = vc.t0[c];
    const int e = vc.e[c];
    const bool spec = phase != 1 && e != TEHMM_SPEC_NONE && np == TEHMM_PB && seen && wait_chunk(c) && vc.ok[c] != 0;
    const int64_t g = t0 + (TEHMM_VROW - 1 - ((t0 - ct0) & (TEHMM_VROW - 1)));
    const int64_t target = ct0 + CS;
    const bool do_check = spec && g < t0 + np && g + 1 < target && g >= since + TEHMM_FIX_MINSTEP;
    const int pg = (int)(g - t0);
    double wrow[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, wend[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    double span = 0.0;
    if (do_check) {
      const int64_t ro = ((c * (CS / TEHMM_VROW)) + (g - ct0) / TEHMM_VROW) * NP;
      const int64_t eo = ((c * (CS / TEHMM_VROW)) + (CS / TEHMM_VROW - 1)) * NP;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int j = lane + 64 * s;
        wrow[0][s] = j < N ? vc.rows[ro + j] : -INFINITY;
        wrow[1][s] = j < N ? rows2[ro + j] : -INFINITY;
        wend[0][s] = j < N ? vc.rows[eo + j] : -INFINITY;
        wend[1][s] = j < N ? rows2[eo + j] : -INFINITY;
      }
      span = fabs(vc.gain[c]) * 1.01 + 256.0;
    }
    bool jumped = false;
    for (int p = 0; p < np; ++p) {
      const int64_t t = t0 + p;
      double b[2];
      if (use_ring) {
        b[0] = ring[p * W + lane];
        b[1] = ring[p * W + lane + 64];
      } else {
        if (tnx != t) {
          bnx[0] = BL[(p0 + t) * TEHMM_WIDE_S + lane];
          bnx[1] = BL[(p0 + t) * TEHMM_WIDE_S + lane + 64];
        }
        b[0] = bnx[0];
        b[1] = bnx[1];
        if (t + 1 < T) {
          bnx[0] = BL[(p0 + t + 1) * TEHMM_WIDE_S + lane];
          bnx[1] = BL[(p0 + t + 1) * TEHMM_WIDE_S + lane + 64];
          tnx = t + 1;
        }
      }
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
          vfin[s] = j < N ? v : -INFINITY;
          vn[j] = vfin[s];
        }
      } else {
        const bool rg = RATIO && r > 1.;
        const double rm1 = r - 1.;
        double best[2] = {-INFINITY, -INFINITY};
        int arg[2] = {0, 0};
        if (w == 0) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            double cc = (vp[0] + ltq[s][0]) + b[s];
            if (RATIO) {
              cc += ltd[s] * r;
              if (lane + 64 * s == 0) cc -= lt00;
            }
            best[s] = cc;
          }
        }
#pragma unroll
        for (int i = 0; i < QM; ++i) {
          if (i == 0 && w == 0) continue;                    // (uniform per wave)
          const double vf = vp[min(f0 + i, W - 1)];
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            double cc = (vf + ltq[s][i]) + b[s];
            if (rg) cc += ltd[s] * rm1;
            if (cc > best[s]) { best[s] = cc; arg[s] = f0 + i; }
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
            const double cc = pval[(par * 4 + ww) * W + j];
            if (cc > fin) { fin = cc; fa = parg[(par * 4 + ww) * W + j]; }
          }
          vfin[s] = j < N ? fin : -INFINITY;
          vn[j] = vfin[s];
          if (w == 0 && j < N) tb[(p0 + t) * TBW + j] = (uint8_t)fa;
        }
      }
      cur ^= 1;
      if (do_check && p == pg) {
        if (adopt(c, e, wrow, wend, span, g + 1)) {
          jumped = true;
          t0 = target;
          since = target;
          break;
        }
      }
    }
    if (!jumped) t0 += np;
    __syncthreads();
  }
  if (phase == 1) {
    if (w == 0) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
        if (lane + 64 * s < NP) hvec[(size_t)id * NP + lane + 64 * s] = vfin[s];
    }
    if (threadIdx.x == 0) {
      hflag[id] = (seen && hs > 0) ? 1 : 0;
      if (stats) atomicAdd(&stats[0], n_block);
    }
    return;
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
    if (stats) { atomicAdd(&stats[0], n_block); atomicAdd(&stats[1], n_jump); }
  }
}

// Traceback bytes of the adopted chunks: the quantised pass keeps them in its own two buffers (hypothesis 0 for the
// whole chunk, hypothesis 1 from the chunk's first tie on -- before it the hypotheses agree), the chain says which
// hypothesis it took and from where; the exact chain's own bytes, written straight to tb, are never touched by a
// pass that is still running.  One workgroup per chunk.
__global__ __launch_bounds__(256) void k_wide_tb_select(IntervalTab iv, VitChunks vc, int N, int TBW, uint8_t *tb,
                                                        const uint8_t *__restrict__ tb1, const uint8_t *__restrict__ tb2,
                                                        const int64_t *sel_from, const int *sel_hyp) {
  const int c = blockIdx.x;
  const int hyp = sel_hyp[c];
  if (hyp < 0) return;
  const int id = vc.iv[c];
  const int64_t p0 = iv.pos0[id], ct0 = vc.t0[c];
  const int64_t from = sel_from[c], end = ct0 + vc.CS;
  const int64_t split = hyp == 1 ? min(end, max(from, ct0 + (int64_t)vc.ties[(int64_t)c * TEHMM_SPEC_MAXT])) : end;
  const int64_t o = (p0 + from) * (int64_t)TBW;
  const int64_t n1 = (split - from) * (int64_t)TBW, n = (end - from) * (int64_t)TBW;
  // (TBW is a multiple of 4: dwords)
  const uint32_t *s1 = (const uint32_t *)(tb1 + o), *s2 = (const uint32_t *)(tb2 + o);
  uint32_t *d = (uint32_t *)(tb + o);
  for (int64_t i = threadIdx.x; i < n / 4; i += blockDim.x) d[i] = i < n1 / 4 ? s1[i] : s2[i];
}

}  // namespace tehmm
