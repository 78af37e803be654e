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
  static constexpr int ROW_D = 4 * KSP;                   // doubles per table row (global copy)
  // LDS copy: rows are 2 doubles further apart, so that consecutive rows start 4 banks x an odd number
  // apart (8 KSP + 4 dwords = 4 mod 8): 16 lanes gathering 16 different rows spread over all 64 banks
  // instead of piling up on 4 (a 320-byte stride is 16 banks mod 64)
  static constexpr int ROW_D_LDS = 4 * KSP + 2;
};

// What the fused passes read besides the transition table:
//   * rixx: the observation rows ALREADY TRANSLATED to table-row indices (k_fused_rowindex, once per batch,
//     model and lane geometry), one uint16 per track in processing order -- the first n_glb entries of a row
//     are rows of the global table ptab (tracks too large for LDS, served from L2), the rest rows of the LDS
//     copy; a symbol beyond its track's range points at the all-ones (log: zeros) row.  With the indices
//     precomputed a step needs no per-track metadata (the first version read rowcnt / rowbase / ldsbase of
//     every track from the kernel-argument segment: 20 scalar loads and waits per position).
//     Layout: per 16-item tile and per block of SB steps one contiguous 3 KB record
//         [tile][block][step in block][word][item in tile]   (uint64 words of four indices)
//     over the item's EXTENDED range of L + 2 Wu steps (position t0 - Wu + e, clamped to the interval), so
//     that both passes find their warm-up positions in their own record.  A wave copies the record of its
//     next block into LDS with three coalesced 16-byte loads per lane, one block (>= 3 steps) ahead: the
//     index words are then LDS reads, and no wait of the step loop is ever for a stream load (vector-memory
//     waits complete in issue order: a wait for a fresh 8-byte read 12 KB away from its neighbour's, behind the
//     previous step's stores, cost ~2 000 cycles three times per step in the version that read them directly).
//   * ptab / the LDS copy: rows [4][KSP] as described above.
#ifndef TEHMM_FUSED_2W
#define TEHMM_FUSED_2W 36        // two waves per SIMD (256 registers each) up to this many padded states, one above
#endif
#define TEHMM_FUSED_BLKW 24      // index words per (item, block): SB = 24 / FKW steps per block
#ifndef TEHMM_FUSED_NSLOT_CAP
#define TEHMM_FUSED_NSLOT_CAP 64
#endif

struct FusedTab {
  const unsigned long long *rixx;   // [tiles][NB][SB][FKW][16]
  const double *ptab;       // [(R + 1)][4][KSP] linear (or log, LOGDOM) rows + scale slot, global
  const double *ptab_lds;   // [lds_rows][ROW_D_LDS] the LDS-staged rows
  int FKW;                  // index words per position: ceil(K / 4) rounded up to a divisor of 24
  int SB, NB;               // steps per block, blocks per item (extended range)
  int K, n_glb;             // entries of an index record (tracks + identity padding, see EmisStream); the first
                            // n_glb (>= EmisStream::NGS) come from ptab, then >= NSLOT from the LDS copy
  int lds_rows;
  double normalize;         // LOGDOM only
};

// the slice of table row `row` for this lane's quarter: KSP doubles into x[]
template <int NT>
__device__ __forceinline__ void fused_load_lds(const double *lds, int row, int kq, double (&x)[FusedGeom<NT>::KSP]) {
  using G = FusedGeom<NT>;
  lds_cd2 *p = (lds_cd2 *)(size_t)((unsigned)(size_t)(__attribute__((address_space(3))) const double *)lds +
                                   (unsigned)(row * (G::ROW_D_LDS * 8) + kq * G::SLICE_B));
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

// sum over the four lanes of an item (lanes l, l + 16, l + 32, l + 48) without going through LDS:
// v_permlane32_swap folds the wave's halves, v_permlane16_swap the odd / even rows (see rep_rows)
__device__ __forceinline__ double item_sum4(double t) {
  const unsigned lo = __double2loint(t), hi = __double2hiint(t);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const double t1 = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
  const unsigned lo1 = __double2loint(t1), hi1 = __double2hiint(t1);
  const auto c = __builtin_amdgcn_permlane16_swap(lo1, lo1, false, false);
  const auto d = __builtin_amdgcn_permlane16_swap(hi1, hi1, false, false);
  return __hiloint2double((int)d[0], (int)c[0]) + __hiloint2double((int)d[1], (int)c[1]);
}
__device__ __forceinline__ double item_max4(double t) {
  const unsigned lo = __double2loint(t), hi = __double2hiint(t);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const double t1 = fmax(__hiloint2double((int)b[0], (int)a[0]), __hiloint2double((int)b[1], (int)a[1]));
  const unsigned lo1 = __double2loint(t1), hi1 = __double2hiint(t1);
  const auto c = __builtin_amdgcn_permlane16_swap(lo1, lo1, false, false);
  const auto d = __builtin_amdgcn_permlane16_swap(hi1, hi1, false, false);
  return fmax(__hiloint2double((int)d[0], (int)c[0]), __hiloint2double((int)d[1], (int)c[1]));
}

// The wave's window on its index records: two 3 KB LDS buffers, the record of the next block on its way
// through three registers per lane.  dir = +1 (forward pass: blocks ascend) or -1.
typedef unsigned int fused_u4 __attribute__((ext_vector_type(4)));
struct IndexWindow {
  const fused_u4 *src;       // first record of this tile
  unsigned lds_base;         // LDS byte address of the wave's two buffers
  int lane, dir, NB;
  int cur = -1;              // block currently readable in buffer (cur & 1)
  fused_u4 r0, r1, r2;       // record of block cur + dir (valid when inflight)
  bool inflight = false;

  __device__ __forceinline__ void request(int blk) {
    if (blk < 0 || blk >= NB) { inflight = false; return; }
    const fused_u4 *p = src + (int64_t)blk * 192 + lane * 3;
    r0 = p[0];
    r1 = p[1];
    r2 = p[2];
    inflight = true;
  }
  __device__ __forceinline__ void commit(int blk) {
    typedef __attribute__((address_space(3))) fused_u4 lds_u4;
    lds_u4 *d = (lds_u4 *)(size_t)(lds_base + (unsigned)((blk & 1) * 3072 + lane * 48));
    d[0] = r0;
    d[1] = r1;
    d[2] = r2;
    cur = blk;
  }
  __device__ __forceinline__ void start(int blk) {
    request(blk);
    commit(blk);
    request(blk + dir);
  }
  // make block `blk` readable (it is cur or cur + dir)
  __device__ __forceinline__ void need(int blk) {
    if (blk != cur) {
      commit(blk);
      request(blk + dir);
    }
  }
  __device__ __forceinline__ unsigned word_addr(int blk, int r, int FKW, int w) const {
    return lds_base + (unsigned)((blk & 1) * 3072 + ((r * FKW + w) * 16 + (lane & 15)) * 8);
  }
};
__device__ __forceinline__ unsigned long long lds_read_u64(unsigned addr) {
  typedef __attribute__((address_space(3))) const unsigned long long lds_cu64;
  return *(lds_cu64 *)(size_t)addr;
}

// Emission row of ONE position for this lane's KS states, computed in slots that the passes interleave
// with their matrix-core instructions (an MFMA occupies its pipe for 64 cycles during which the wave can
// issue about a dozen independent vector instructions: two waves that run the same program in lock step
// gain nothing from each other, so the overlap has to happen inside the wave).
//   begin(e) ; slot(0) .. slot(NSLOT - 1) ; finish(q, ms)
// Linear form: q = product of the track rows, ms = sum of their scale slots.  LOGDOM: q = exp(normalize *
// sum - rowmax), ms = rowmax.  A row no state can emit comes out as all zeros (the callers turn it into NaN).
// The schedule is FIXED at compile time AND free of branches: the host pads the track list of the index records
// with identity rows (all ones; log: zeros) to NGS tracks served from the global table followed by NSLOT
// LDS-staged ones, so every slot does the same thing whatever the model (a run-time state machine cost ~100
// scalar instructions per slot; per-slot tests of the track counts cost ~35 scalar branches per step, each a
// refetch through the instruction cache the kernel is bound by).  Slot k folds the LDS track whose rows slot
// k - 1 requested, requests the rows of LDS track k and reads the index word of track k + 1; the first
// global-table (L2) track is requested in begin() and folded in slot GF, the second requested there and folded
// in finish().  Whatever does not fit (more than NSLOT LDS tracks, more than NGS global ones) is done in
// finish(), one track after the other.
template <int NT, bool LOGDOM>
struct EmisStream {
  using G = FusedGeom<NT>;
  static constexpr int NSLOT = G::KS < TEHMM_FUSED_NSLOT_CAP ? G::KS : TEHMM_FUSED_NSLOT_CAP;
  static constexpr int NGS = NSLOT > 1 ? 2 : 1;  // global tracks of the fixed schedule (FusedTab::n_glb >= NGS)
  static constexpr int GF = NSLOT >= 8 ? 4 : NSLOT / 2;
  const FusedTab &ft;
  const double *lds;
  IndexWindow &win;
  const int kq;
  unsigned wbase = 0;                            // LDS address of word 0 of the position being gathered
  double q[G::KS];
  double sc = 0.0;
  double xa[G::KSP];                             // global-table (L2) track in flight
  double xc[G::KSP];                             // LDS track in flight (requested in one slot, folded in the next)
  unsigned long long wcur = 0;                   // index word of the next LDS track

  __device__ __forceinline__ EmisStream(const FusedTab &ft_, const double *lds_, IndexWindow &win_, int kq_)
      : ft(ft_), lds(lds_), win(win_), kq(kq_) {}
  static __device__ __forceinline__ int entry(unsigned long long w, int i) { return (int)((w >> (16 * (i & 3))) & 0xffffull); }
  __device__ __forceinline__ unsigned long long word(int w) const { return lds_read_u64(wbase + (unsigned)w * 128u); }
  __device__ __forceinline__ void fold(const double (&x)[G::KSP]) {
#pragma unroll
    for (int s = 0; s < G::KS; ++s) q[s] = LOGDOM ? q[s] + x[s] : q[s] * x[s];
    if (!LOGDOM) sc += x[G::KS];
  }
  __device__ __forceinline__ void glb_request(int i) { fused_load_glb<NT>(ft.ptab, entry(word(i >> 2), i), kq, xa); }
  // e = extended step of the position to gather (the window must hold its block)
  __device__ __forceinline__ void begin(int e) {
    const int blk = e / ft.SB, r = e - blk * ft.SB;
    win.need(blk);
    wbase = win.word_addr(blk, r, ft.FKW, 0);
#pragma unroll
    for (int s = 0; s < G::KS; ++s) q[s] = LOGDOM ? 0.0 : 1.0;
    sc = 0.0;
    glb_request(0);
    wcur = word(ft.n_glb >> 2);
  }
  __device__ __forceinline__ void slot(int k) {
    if (k >= NSLOT) return;
    const int il = ft.n_glb + k;                                // this slot's LDS track
    if (k > 0) fold(xc);                                        // requested by the previous slot
    if (k == GF && NSLOT > 1) {
      fold(xa);
      glb_request(1);
    }
    fused_load_lds<NT>(lds, entry(wcur, il), kq, xc);
    if (k + 1 < NSLOT) wcur = word((il + 1) >> 2);
  }
  __device__ __forceinline__ void finish(int N, double (&qo)[G::KS], double &ms) {
    fold(xc);                                                   // the last slot's track
    fold(xa);                                                   // the last global track of the schedule
    for (int i = NGS; i < ft.n_glb; ++i) {                      // further global tracks, one after the other
      glb_request(i);
      fold(xa);
    }
    for (int il = ft.n_glb + NSLOT; il < ft.K; ++il) {          // LDS tracks beyond the slots
      fused_load_lds<NT>(lds, entry(word(il >> 2), il), kq, xc);
      fold(xc);
    }
    if (LOGDOM) {
      double m = -INFINITY;
#pragma unroll
      for (int s = 0; s < G::KS; ++s) {
        q[s] *= ft.normalize;
        m = fmax(m, kq + 4 * s < N ? q[s] : -INFINITY);
      }
      m = item_max4(m);
      const bool good = m > -1e20;
#pragma unroll
      for (int s = 0; s < G::KS; ++s) qo[s] = (good && kq + 4 * s < N) ? exp_nonpos(q[s] - m) : 0.0;
      ms = good ? m : 0.0;
    } else {
#pragma unroll
      for (int s = 0; s < G::KS; ++s) qo[s] = q[s];
      ms = sc;
    }
  }
};

// LDS of a fused pass: table rows [lds_rows][ROW_D_LDS] doubles | per wave two index buffers of 3 KB | (backward
// pass) per wave a [16][NT] tile of posterior values + 16 row bases for the transposing store
template <int NT>
__host__ __device__ inline size_t fused_lds_bytes(int lds_rows, bool post_tiles) {
  return (size_t)lds_rows * FusedGeom<NT>::ROW_D_LDS * 8 + 4 * 2 * 3072 + (post_tiles ? 4 * (16 * NT * 8 + 128) : 0);
}
template <int NT>
__device__ __forceinline__ void fused_stage(const FusedTab &ft, double *lds) {
  const int n = ft.lds_rows * FusedGeom<NT>::ROW_D_LDS;
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = ft.ptab_lds[i];
}

// Observation rows -> index records (see FusedTab).  order[i] = the track processed i-th.
struct FusedOrder {
  int K, n_glb, FKW, KP, SB, NB, Wu;
  int zero_glb, zero_lds;
  unsigned char order[TEHMM_MAX_TRACKS];
  int base[TEHMM_MAX_TRACKS];        // by processing slot: row base in the table the slot reads
  int cnt[TEHMM_MAX_TRACKS];         // by processing slot: rows of the track
};
// one thread per (tile, block, item in tile, step in block): FKW words.  Consecutive threads take consecutive STEPS of
// one item -- SB observation rows, one contiguous run of the interval -- and the threads of the next item follow: the
// reads are runs of SB rows and the writes runs of eight words per step (thread order (tile, step, item) read one
// 12-byte row per 128-byte line: 4.2 ms per 100 Mb, ten times the bytes)
__global__ __launch_bounds__(256) void k_fused_rowindex(IntervalTab iv, LaneGeom lg, FusedOrder fo, const uint8_t *obs,
                                                        unsigned long long *rixx) {
  const int E = fo.NB * fo.SB;
  const int64_t n = (int64_t)lg.n_groups * 4 * E * 16;
  const int per_blk = 16 * fo.SB;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tb = idx / per_blk;                  // (tile, block)
    const int rem = (int)(idx - tb * per_blk);
    const int i16 = rem / fo.SB;
    const int e = (int)(tb % fo.NB) * fo.SB + (rem - i16 * fo.SB);
    const int64_t tile = tb / fo.NB;
    const int64_t item = tile * 16 + i16;
    const bool valid = item < lg.n_items;
    const int id = valid ? lg.item_iv[item] : 0;
    const int64_t T = iv.len[id];
    int64_t t = (valid ? lg.item_t0[item] : 0) - fo.Wu + e;
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);
    const uint8_t *orow = obs + (iv.pos0[id] + (T > 0 ? t : 0)) * fo.KP;
    const int blk = e / fo.SB, r = e - blk * fo.SB;
    unsigned long long *dst = rixx + ((tile * fo.NB + blk) * TEHMM_FUSED_BLKW + (int64_t)r * fo.FKW) * 16 + i16;
    for (int w = 0; w < fo.FKW; ++w) {
      unsigned long long word = 0;
      for (int b = 0; b < 4; ++b) {
        const int i = 4 * w + b;
        int v = 0;
        if (i < fo.K) {
          const int sym = orow[fo.order[i]];
          v = sym < fo.cnt[i] ? fo.base[i] + sym : (i < fo.n_glb ? fo.zero_glb : fo.zero_lds);
        }
        word |= (unsigned long long)(v & 0xffff) << (16 * b);
      }
      dst[(int64_t)w * 16] = word;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Forward pass.  Same contract as k_fb_lane<NT, 0> / k_fb_mfma<NT, 0>: alpha' rows of the official range
// (item-interleaved), pre / end vectors, cumulative log-scale records; the emission rows are computed here.
// Extended step e <-> position t0 - Wu + e; the pass covers e = 0 .. L + Wu - 1.
// ------------------------------------------------------------------------------------------
template <int NT, bool LOGDOM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT <= TEHMM_FUSED_2W ? 2 : 1, NT <= TEHMM_FUSED_2W ? 2 : 1)))
void k_fused_fwd(IntervalTab iv, FusedTab ft, LaneGeom lg, int N, int CS, int Wu,
                 const double *__restrict__ tab /* A, [NT][NT] row-major */, float *al32, double *chkf, double *pre,
                 double *end, double *slog32) {
  using G = FusedGeom<NT>;
  constexpr int KS = G::KS, RT = G::RT;
  extern __shared__ double fused_lds[];
  fused_stage<NT>(ft, fused_lds);
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
  const int64_t T = iv.len[id];
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
  auto vec_out = [&](double *dst) {
    const int64_t po = ((((item >> 6) * NT) + kq) << 6) + (item & 63);
#pragma unroll
    for (int s = 0; s < KS; ++s) dst[po + ((int64_t)(4 * s) << 6)] = v[s];
  };
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  IndexWindow win;
  win.src = (const fused_u4 *)ft.rixx + (int64_t)tile * ft.NB * 192;
  win.lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) const double *)fused_lds +
                 (unsigned)(ft.lds_rows * G::ROW_D_LDS * 8 + (threadIdx.x >> 6) * 2 * 3072);
  win.lane = lane;
  win.dir = 1;
  win.NB = ft.NB;
  win.start(0);
  double q[KS], ms;
  EmisStream<NT, LOGDOM> es(ft, fused_lds, win, kq);
  es.begin(0);
#pragma unroll
  for (int k = 0; k < KS; ++k) es.slot(k);
  es.finish(N, q, ms);
  const int E1 = ft.NB * ft.SB - 1;                  // last extended step the records hold
  es.begin(min(1, E1));
  {
    // (the same five stores once BEHIND the first begin(): every path into slot GF's wait now has at least five stores
    //  younger than the table loads it waits for, so the compiler can make it vmcnt(5) -- see the stores in the loop)
    float2 *ar = (float2 *)al32 + al32_index<NT>(lg, item, 0, kq) / 2;
#pragma unroll
    for (int p = 0; p < (KS + 1) / 2; ++p) ar[(int64_t)p * 64] = make_float2(0.f, 0.f);
  }
#ifdef TEHMM_STAMPS
  unsigned long long stq[4] = {0, 0, 0, 0}, stt = stamp_now();
#define FST(i) do { const unsigned long long n_ = stamp_now(); stq[i] += n_ - stt; stt = n_; } while (0)
#else
#define FST(i)
#endif
  for (int s = -Wu; s < L; ++s) {
    if (s == 0) {
      if (run) vec_out(pre);
      slog = 0.0;
    }
    FST(3);
    // the product on the matrix cores, the next position's emission row gathered between its instructions
    lane_d4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (lane_d4){0.0, 0.0, 0.0, 0.0};
    // (the gather of position s + 1 was begun before the previous step's stores: vector-memory waits complete
    //  in issue order, and the first table load must not queue up behind them)
#pragma unroll
    for (int k = 0; k < KS; ++k) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(tf[rt][k], v[k], acc[rt], 0, 0, 0);
      es.slot(k);
      __builtin_amdgcn_sched_barrier(0);
    }
    FST(0);
    double qn[KS], msn = 0.0;
    es.finish(N, qn, msn);
    FST(1);
    double a[KS];
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      a[k] = acc[k >> 2][k & 3] * q[k];
      t += a[k];
    }
    t = item_sum4(t);
    const int e = ((__double2hiint(t) >> 20) & 0x7ff) - 1022;
    // scale by 2^-e; nothing can emit here / the product underflowed: scale by NaN instead, which poisons the
    // item (its links fail, the exact chain walks)
    const bool okrow = t > 1e-280 && t < INFINITY;
    const double scale = okrow ? __hiloint2double((1023 - e) << 20, 0) : qnan;
#pragma unroll
    for (int k = 0; k < KS; ++k) v[k] = a[k] * scale;
    slog += (double)e * 0.6931471805599453 + ms;
#pragma unroll
    for (int k = 0; k < KS; ++k) q[k] = qn[k];
    const double ms_now = ms;
    ms = msn;
    (void)ms_now;
    es.begin(min(s + 2 + Wu, E1));        // next step's gather: its first loads go out ahead of the stores below
    FST(2);
    {
      // alpha' row as floats: five float2 per lane (states kq + 4 (2p), kq + 4 (2p + 1)), 512 bytes per wave store.
      // UNCONDITIONAL (round 4): the warm-up steps write their vectors to the row of position 0, which step 0
      // overwrites, and lanes whose item is not run write rows that the exact chain rewrites (it walks those items
      // whole) or nobody reads.  With the stores behind a branch the wait for the table rows requested ahead of them
      // (slot GF of the next step) has to be vmcnt(0) -- on the path without stores nothing younger is outstanding --
      // and then waits for the stores' acknowledgement as well: ~1 us in the middle of every step's MFMA loop.
      float2 *ar = (float2 *)al32 + al32_index<NT>(lg, item, s < 0 ? 0 : s, kq) / 2;
#pragma unroll
      for (int p = 0; p < (KS + 1) / 2; ++p)
        ar[(int64_t)p * 64] = run ? make_float2((float)v[2 * p], 2 * p + 1 < KS ? (float)v[2 * p + 1] : 0.f)
                                  : make_float2(0.f, 0.f);           // (zeros, not this lane's meaningless vector)
    }
    if (s >= 0 && run) {
      if ((s & 31) == 31) {
        if (kq == 0) slog32[item * (L / 32) + (s >> 5)] = slog;
        if ((s & 63) == 31) {                              // fp64 row where the exact chain checks its direction
          double *cr = chkf + (item * (L / 64) + (s >> 6)) * NT + kq;
#pragma unroll
          for (int k = 0; k < KS; ++k) cr[4 * k] = v[k];
        }
      }
    }
  }
  if (run) vec_out(end);
#ifdef TEHMM_STAMPS
  if (lane == 0 && blockIdx.x < 4096)
    for (int i = 0; i < 4; ++i) g_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + i] = stq[i];
#endif
#undef FST
}

// ------------------------------------------------------------------------------------------
// Backward pass + posterior.  v = w_{t+1} = bh'_{t+1} * beta_{t+1} -> beta_t = normalise(A v); the posterior
// row normalise(alpha'_t * beta_t) (+ eps quirk) goes straight to post [T][N]; chk [item][L / 64][NT]
// keeps beta_t at the positions == 31 (mod 64), where k_fb_fix checks its direction.
// Item-relative step s (L + Wu - 1 down to 0) <-> extended step s + Wu.
// ------------------------------------------------------------------------------------------
// ESTEP (Baum-Welch E-step, tehmm_estep.hip.h): instead of the posterior rows the pass leaves, as floats in the
// alpha' layout (al32_index), gamma_t = normalise(alpha'_t * beta_t) (no eps: fit, basehmm.py:516-517) and
// wz_t = w_{t+1} * scale_t / G_t, the row that makes xi_t(i, j) = alpha'_t[i] A[i][j] wz_t[j] sum to one
// (_hmm.pyx:62-117 in the scaled linear domain; G_t = sum_j alpha'_t[j] beta_t[j], scale_t the power of two
// that normalised beta_t): k_estep_reduce turns the three float rows of a position into the statistics.
template <int NT, bool LOGDOM, bool EPS, bool ESTEP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT <= TEHMM_FUSED_2W ? 2 : 1, NT <= TEHMM_FUSED_2W ? 2 : 1)))
void k_fused_bwd(IntervalTab iv, FusedTab ft, LaneGeom lg, int N, int CS, int Wu,
                 const double *__restrict__ tab /* A, [NT][NT] row-major */, const float *__restrict__ al32,
                 double *post, double *pre, double *end, double *chk, float *gam32 = nullptr, float *wz32 = nullptr,
                 double *sink = nullptr /* [tiles][64]: where the masked elements of the posterior stores go */) {
  using G = FusedGeom<NT>;
  constexpr int KS = G::KS, RT = G::RT;
  extern __shared__ double fused_lds[];
  fused_stage<NT>(ft, fused_lds);
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
  const int64_t T = iv.len[id];
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
  // (the emission row is gathered at step s itself also above `top`: positions beyond the interval end
  //  were clamped when the records were built and v is forced to zero there)
  IndexWindow win;
  win.src = (const fused_u4 *)ft.rixx + (int64_t)tile * ft.NB * 192;
  win.lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) const double *)fused_lds +
                 (unsigned)(ft.lds_rows * G::ROW_D_LDS * 8 + (threadIdx.x >> 6) * 2 * 3072);
  win.lane = lane;
  win.dir = -1;
  win.NB = ft.NB;
  win.start((L + 2 * Wu - 1) / ft.SB);
  // Transposing store of the posterior rows.  A lane holds 9 scattered states of one item: stored directly,
  // every store instruction is 16 pieces of 32 bytes in 16 different rows, and the wave then spends ~40 % of its
  // time waiting for them to retire.  Instead the 16 x N values of a step go through an LDS tile and leave as 9
  // stores of 64 consecutive doubles of the [16][NT] tile: runs of up to 280 contiguous bytes per row.
  typedef __attribute__((address_space(3))) double lds_f64;
  typedef __attribute__((address_space(3))) long long lds_i64;
  const unsigned ptile = (unsigned)(size_t)(__attribute__((address_space(3))) const double *)fused_lds +
                         (unsigned)(ft.lds_rows * G::ROW_D_LDS * 8 + 4 * 2 * 3072 + (threadIdx.x >> 6) * (16 * NT * 8 + 128));
  const unsigned pinfo = ptile + 16 * NT * 8;
  if (!ESTEP && kq == 0) *(lds_i64 *)(size_t)(pinfo + (lane & 15) * 8) = run ? (long long)prow0 : -1ll;
  double q[KS], ms;
  EmisStream<NT, LOGDOM> es(ft, fused_lds, win, kq);
  es.begin(L + 2 * Wu - 1);
#pragma unroll
  for (int k = 0; k < KS; ++k) es.slot(k);
  es.finish(N, q, ms);
  es.begin(max(L + 2 * Wu - 2, 0));
#ifdef TEHMM_STAMPS
  unsigned long long stq[4] = {0, 0, 0, 0}, stt = stamp_now();
#define FST(i) do { const unsigned long long n_ = stamp_now(); stq[i] += n_ - stt; stt = n_; } while (0)
#else
#define FST(i)
#endif
  for (int s = L + Wu - 1; s >= 0; --s) {
    if (s == L - 1 && run) vec_out(pre);               // v = w_{t0+L} after the warm-up
    FST(3);
    lane_d4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (lane_d4){0.0, 0.0, 0.0, 0.0};
    // (the gather of position s - 1 was begun before the previous step's stores, see k_fused_fwd)
    // alpha' row of this position (official range only; floats, five float2 per lane)
    // (loaded at EVERY step, the warm-up steps read the row of position L - 1 for nothing: a fixed number of
    //  vector-memory operations per step lets the compiler count its waits instead of draining -- see the stores below)
    float2 alp[(KS + 1) / 2];
    {
      const float2 *ar = (const float2 *)al32 + al32_index<NT>(lg, item, s < L ? s : L - 1, kq) / 2;
      // (all five requested together, whatever `run` says -- every lane's row exists: left alone the compiler sinks each
      //  load behind a test of `run` and waits for it on the spot, five dependent round trips to L2 / HBM per step)
#pragma unroll
      for (int p = 0; p < (KS + 1) / 2; ++p) alp[p] = ar[(int64_t)p * 64];
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(tf[rt][k], v[k], acc[rt], 0, 0, 0);
      es.slot(k);
      __builtin_amdgcn_sched_barrier(0);
    }
    FST(0);
    // (the alpha' row is USED here, unconditionally as far as the compiler can tell: its wait sits behind the MFMA loop)
#pragma unroll
    for (int p = 0; p < (KS + 1) / 2; ++p) asm volatile("" : "+v"(alp[p].x), "+v"(alp[p].y));
#pragma unroll
    for (int p = 0; p < (KS + 1) / 2; ++p) alp[p] = run ? alp[p] : make_float2(0.f, 0.f);
    double qn[KS], msn = 0.0;
    es.finish(N, qn, msn);
    FST(1);
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < KS; ++k) t += acc[k >> 2][k & 3];
    t = item_sum4(t);
    const int e = ((__double2hiint(t) >> 20) & 0x7ff) - 1022;
    double bt[KS];
    // an impossible emission row (all zeros) must poison the item: q enters v below, a zero v gives t = 0
    const bool okrow = (s >= L && s >= top) || (t > 0.0 && t < INFINITY);
    const double scale = okrow ? __hiloint2double((1023 - e) << 20, 0) : qnan;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      bt[k] = acc[k >> 2][k & 3] * scale;                                  // beta_t
      if (s >= L && s == top) bt[k] = kq + 4 * k < N ? 1.0 : 0.0;           // uniform start
    }
    double g[KS];
    float wzf[ESTEP ? KS : 1];
#pragma unroll
    for (int k = 0; k < KS; ++k) g[k] = 0.0;
#pragma unroll
    for (int k = 0; k < (ESTEP ? KS : 1); ++k) wzf[k] = 0.f;
    if (s < L) {
      // posterior row: normalise(alpha' * beta) in registers, straight to post [T][N]
      double gt = 0.0;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        g[k] = (double)((k & 1) ? alp[k >> 1].y : alp[k >> 1].x) * bt[k];
        gt += g[k];
      }
      gt = item_sum4(gt);
      const double inv = 1.0 / gt;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        g[k] *= inv;
        if (EPS) g[k] = (g[k] + eps) * inv_epsden;
      }
      if constexpr (ESTEP) {
        const double wzs = scale * inv;                    // v is still w_{t+1} here
#pragma unroll
        for (int k = 0; k < KS; ++k) wzf[k] = (float)(v[k] * wzs);
      }
#pragma unroll
      for (int k = 0; k < KS; ++k) v[k] = q[k] * bt[k];
#pragma unroll
      for (int k = 0; k < KS; ++k) q[k] = qn[k];
    }
    if (s >= L) {                          // warm-up steps: no posterior
#pragma unroll
      for (int k = 0; k < KS; ++k) v[k] = s > top ? 0.0 : q[k] * bt[k];
#pragma unroll
      for (int k = 0; k < KS; ++k) q[k] = qn[k];
    }
    es.begin(max(s - 2 + Wu, 0));         // next step's gather: its first loads go out ahead of the stores below
    // The stores of a step are UNCONDITIONAL and always the same number (round 4): masked elements -- padding columns,
    // items that are not run, the warm-up steps -- go to a per-wave sink (ESTEP: to the row of position L - 1, which the
    // first official step overwrites; rows of items that are not run are rewritten by the exact chain).  Behind a
    // branch their number depends on the path, the waits for the table rows requested ahead of them become vmcnt(0),
    // and every step's MFMA loop then waits for the previous step's stores to be acknowledged.
    {
      const bool off = s < L;
      if constexpr (ESTEP) {
        const int64_t ai = al32_index<NT>(lg, item, off ? s : L - 1, kq) / 2;
        float2 *gr = (float2 *)gam32 + ai, *wr = (float2 *)wz32 + ai;
#pragma unroll
        for (int p = 0; p < (KS + 1) / 2; ++p) {
          // (lanes whose item is not run write zeros: the reductions read slots beyond an interval's end, which the
          //  exact chain never rewrites)
          gr[(int64_t)p * 64] = run ? make_float2((float)g[2 * p], 2 * p + 1 < KS ? (float)g[2 * p + 1] : 0.f)
                                    : make_float2(0.f, 0.f);
          wr[(int64_t)p * 64] = run ? make_float2(wzf[2 * p], 2 * p + 1 < KS ? wzf[2 * p + 1] : 0.f)
                                    : make_float2(0.f, 0.f);
        }
      } else {
#pragma unroll
        for (int k = 0; k < KS; ++k) *(lds_f64 *)(size_t)(ptile + (unsigned)(((lane & 15) * NT + kq + 4 * k) * 8)) = g[k];
        int lane_v = lane;                                   // (opaque: keeps the per-store index arithmetic inside
        asm volatile("" : "+v"(lane_v));                     //  the step instead of 27 hoisted registers)
        double *const snk = sink + ((int64_t)tile << 6) + lane;
        int offv = off ? 1 : 0;                              // (opaque and per lane: as a uniform condition the compiler
        asm volatile("" : "+v"(offv));                       //  splits the nine stores into two branchy versions)
        // all LDS reads of the step first, then the stores (read - wait - store nine times over cost nine LDS round
        // trips in a row: 2.7 of the 5.6 thousand cycles this tail took)
        constexpr int NJ = (16 * NT + 63) / 64;
        double val[NJ];
        long long rbase[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int e = 64 * j + lane_v;                     // element of the [16][NT] tile, row-major
          const int ec = (16 * NT % 64 == 0 || e < 16 * NT) ? e : 0;
          const int r = (ec * (65536 / NT + 1)) >> 16;       // e / NT (e < 16 NT <= 1024)
          val[j] = *(lds_f64 *)(size_t)(ptile + (unsigned)ec * 8u);
          rbase[j] = *(lds_i64 *)(size_t)(pinfo + (unsigned)r * 8u);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int e = 64 * j + lane_v;
          const int ec = (16 * NT % 64 == 0 || e < 16 * NT) ? e : 0;
          const int r = (ec * (65536 / NT + 1)) >> 16, c = ec - r * NT;
          const bool real = (offv != 0) & (rbase[j] >= 0) & (c < N) & (16 * NT % 64 == 0 || e < 16 * NT);
          double *dst = real ? post + (rbase[j] + (int64_t)s * N + c) : snk;
          asm volatile("" : "+v"(dst));                      // (ONE store through the selected pointer, no branches;
          *(__attribute__((address_space(1))) double *)dst = val[j];   //  a GLOBAL store: flat ones drain vmcnt)
        }
      }
      if (off && run) {
        if ((s & 63) == 31) {
          double *cr = chk + (item * (L / 64) + (s >> 6)) * NT + kq;
#pragma unroll
          for (int k = 0; k < KS; ++k) cr[4 * k] = bt[k];
        }
      }
    }
    FST(2);
    ms = msn;
  }
  (void)ms;
  if (run) vec_out(end);
#ifdef TEHMM_STAMPS
  if (lane == 0 && blockIdx.x < 4096)
    for (int i = 0; i < 4; ++i) g_stamps[4096 * 16 + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + i] = stq[i];
#endif
#undef FST
}

}  // namespace tehmm
