// tehmm_spec.hip.h -- chunk-parallel EXACT Viterbi: speculative chunk pass + sequential fix-up.
//
// Why this is exact.  The reference computes V_t[j] = fl( max_i fl(V_{t-1}[i] + lt[i][j]) + b_t[j] )
// in fp64 (_hmm.pyx:229-247).  While every finite V value sits in one binade [2^e, 2^(e+1)) they
// are all multiples of u = 2^(e-52), and adding an arbitrary double z to such a value rounds z to the
// grid:  fl(v + z) = v + R_u(z)  (R_u = nearest multiple of u; exact ties are detected and excluded,
// magnitudes only grow because every term is <= 0).  Inside a binade the recurrence is therefore an
// EXACT integer max-plus recurrence with the quantised tables ltq = R_u(lt), bq_t = R_u(b_t): it is
// translation invariant (V -> V + c gives the same arg-max decisions) and tropical matrix products
// converge to rank one, so a chunk started from an ARBITRARY vector W makes, after a few steps, the
// same decisions as the true chain and V_t - W_t becomes one constant delta (Maleki et al., "Parallelizing
// dynamic programming through rank convergence", PPoPP 2014 -- here applied binade by binade so that
// the fp64 rounding of the reference is reproduced bit for bit).
//
//   P0  k_vit_spec<QUANT=false>  every chunk from a zero vector in plain fp64: chunk score gains ->
//       host prefix sum -> which binade each chunk lives in (chunks near a binade boundary, the first
//       chunk of an interval and ragged tails are simply left to the sequential chain);
//   P2  k_vit_spec<QUANT=true>   every speculated chunk from a zero vector with the quantised tables of
//       its binade, exact arithmetic: traceback bytes for the whole chunk, the W rows at every 32nd
//       position, the chunk minimum;
//   FIX k_vit_fix                the exact sequential chain (same arithmetic as k_vit_coop) runs the
//       first block of every chunk; at its 32nd step it compares V with the stored W row: if
//       V - W is one constant, the binade matches and stays matched to the chunk end, the chain
//       jumps to the next chunk with V = W_end + delta; otherwise it simply keeps going.
// Exact rounding ties (b_t[j] an odd multiple of u/2: about one element in 2^(e-5)) are the one case
// where fl(v + z) depends on the parity of v.  P2 does not model them: a position with a tie ends the
// current speculative segment (its W row is recorded), the vector restarts from zeros behind it, and
// the exact chain -- which always runs 64 exact positions after every jump -- lands exactly ON that
// position.  A chunk is therefore a list of segments [tie, next tie).
// The arg-max of P2 comes for free: the from-index is folded into six spare low bits of the
// quantised transition table (values are re-based every 32 steps so they stay small).
#pragma once
#include "tehmm_coop.hip.h"

namespace tehmm {

#define TEHMM_SPEC_NONE (-2147483647 - 1)
#ifndef TEHMM_SPEC_MIN_E
// speculate only where |V| >= 2^15: the quantised pass keeps 64 x (offset from the best state) + 6 index bits
// below 2^e, so a state may fall 2^(e-6) = 512 behind inside a re-basing window at e = 15 (items that exceed it
// are left to the exact chain), and a chunk of 1024 positions still fits inside two binades
#define TEHMM_SPEC_MIN_E 15
#endif

struct VitChunks {
  const int *iv;          // chunk -> interval id
  const int64_t *t0;      // chunk -> first position (multiple of CS)
  const int64_t *first;   // interval id -> index of its first chunk
  int n;                  // number of chunks
  int CS;                 // positions per chunk (multiple of 64)
  const int *e;           // binade exponent the chunk is speculated in, or TEHMM_SPEC_NONE
  double *gain;           // P0: max_j W_end of the chunk started from zeros
  int *ok;                // P2: 1 = the speculative results of the chunk are usable
  double *wmin;           // P2: most negative W (relative to the zero start) seen in the chunk
  double *rows;           // P2: [chunk][CS/16][NT] W rows at positions t0 + 16k + 15
  int *ntie;              // P2: number of tie positions in the chunk (> TEHMM_SPEC_MAXT: unusable)
  int *ties;              // P2: [chunk][MAXT] tie positions relative to t0, ascending
  double *tierows;        // P2: [chunk][MAXT][NT] W row of the position just before each tie
  double *segmin;         // lane passes: [chunk][MAXT + 1] lowest live W of each segment (its frame)
  double *offend;         // lane passes: [chunk] segment-frame offset of the chunk's last item (k_vit_stitch)
  int *clink;             // lane passes: [chunk] 1 = the chunk's first segment continues the previous
                          //   chunk's last one (same binade, exact constant difference)
  double *clk;            // lane passes: [chunk] that constant: frame(previous) = frame(this) + clk
  // k_vit_runs: where a verified jump that enters the chunk's FIRST segment ends -- its first tie, or on
  // through the linked chunks behind it (a suffix scan, so that the chain does not walk the run itself)
  int64_t *rtarget;       // [2][chunk] position the jump lands on (by the parity of the chain's delta, see below)
  int64_t *rsel;          // [2][chunk] which recorded row holds the vector before that position (see below)
  double *racc;           // [2][chunk] sum of clk over the chunks walked: frame(this) = frame(chunk') + racc
  double *rmn;            // [2][chunk] lowest live W up to the landing position, in the chunk's frame
  // soft ties (round 4, tehmm_lane3.hip.h): bit k of tsoft[chunk] = entry k of the chunk's tie list is a ROUNDING tie
  // the quantised pass went through keeping its frame; it did so assuming that the constant between ITS item's frame
  // and the truth is an even multiple of the grid unit, so the tie is passable for a chain whose verified delta
  // (relative to the segment frame) has the parity bit k of tpar[chunk] (the parity of the item's offset in the
  // segment frame).  The other entries are frame breaks (failed item links): every chain stops there.
  // The run scan (k_vit_runs) is kept per entry parity h: rtarget / rsel / racc / rmn hold [2][chunks] (h * n + chunk);
  // rsel = 2 * (64 * chunk' + tie index) + (0: the row before that tie, 1: the chunk's last row).
  unsigned *tsoft, *tpar;
};
#define TEHMM_SPEC_MAXT 32
#ifndef TEHMM_VROW
#define TEHMM_VROW 16             // spacing of the recorded W rows
#endif

// wave-wide max / min over the live lanes (fp64, shuffle based: used once per 32 steps)
__device__ __forceinline__ double wave_max_live(double v, bool live) {
  return wave_max_f64(live ? v : -INFINITY);
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
  return v;
}

// ------------------------------------------------------------------------------------------
// Speculative chunk pass.  blockDim = 256, every wave owns one chunk at a time (grid-stride).
// Phases alternate inside the wave: 64 emission rows (lane = position) into its LDS ring, then 64
// chain steps (lane = destination state, DPP row broadcast of the vector).
// LDS (doubles): ring [4 waves][64][RS]
// ------------------------------------------------------------------------------------------
template <int NT, bool QUANT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_vit_spec(IntervalTab iv, EmisTab em, VitChunks vc, int N,
                                                  const double *g_lt, uint8_t *tb) {
  extern __shared__ double sm[];
  constexpr int RS = NT + 1;
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  double *ring = sm + w * 64 * RS;
  volatile int *tieflag = (volatile int *)(sm + 4 * 64 * RS) + w * 64;   // per position of the block
  const int jl = min(lane, NT - 1);
  const bool live = lane < N;
  for (int c = blockIdx.x * 4 + w; c < vc.n; c += gridDim.x * 4) {
    const int e = QUANT ? vc.e[c] : 0;
    if (QUANT && e == TEHMM_SPEC_NONE) continue;
    const int id = vc.iv[c];
    const int64_t T = iv.len[id];
    const int64_t p0 = iv.pos0[id];
    const int64_t t0 = vc.t0[c];
    const int64_t t1 = min(T, t0 + vc.CS);
    const double u = QUANT ? ldexp(1.0, e - 52) : 0.0;
    const double M = QUANT ? ldexp(1.5, e) : 0.0;        // fl(z + M) - M rounds z to the grid u
    const double half_u = 0.5 * u;
    const double wlim = QUANT ? -ldexp(1.0, e + 1) : -INFINITY;
    bool bad = false;
    // column `lane` of the transition table, quantised and carrying the from-index in its low bits
    double ltc[NT];
#pragma unroll
    for (int f = 0; f < NT; ++f) {
      const double z = live ? g_lt[f * NT + jl] : -INFINITY;
      if (QUANT) {
        const double q = (z + M) - M;
        if (fabs(z - q) == half_u) bad = true;            // exact tie: rounding would depend on V
        ltc[f] = 64.0 * q + (double)(63 - f) * u;         // -inf stays -inf
      } else {
        ltc[f] = z;
      }
    }
    double W = live ? 0.0 : -INFINITY;     // QUANT: 64 x (value - base); plain: value
    double base = 0.0;                     // QUANT: multiple of u, sum of the re-basings
    int nt = 0;                            // QUANT: tie positions found so far
    for (int64_t tb0 = t0; tb0 < t1; tb0 += 64) {
      const int np = (int)min((int64_t)64, t1 - tb0);
      // ---- emission rows of this block (lane = position); the interval's first row is never in a
      // speculated chunk, so the leading-rows quirk does not apply (seen = true)
      {
        const bool act = lane < np;
        const int64_t gpos = p0 + tb0 + (act ? lane : np - 1);
        double x[NT];
        emis_rows<NT>(em, nullptr, gpos, x);
        bool tie = false;
        if (act) {
          double *dst = ring + lane * RS;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            double b = x[j];
            if (QUANT) {
              const double q = (b + M) - M;
              if (j < N && fabs(b - q) == half_u) tie = true;
              b = 64.0 * q;
            }
            dst[j] = b;
          }
        }
        if (QUANT) tieflag[lane] = tie ? 1 : 0;
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);
      // ---- chain steps
      for (int p = 0; p < np; ++p) {
        const int64_t t = tb0 + p;
        const double b = ring[p * RS + jl];
        double rr[(NT + 15) / 16];
        rep_rows<NT>(W, rr);
        double x[NT];
        BcastAdd<0, NT>::run(rr, ltc, x);
#pragma unroll
        for (int n = NT; n > 1; n = (n + 1) / 2) {
#pragma unroll
          for (int i = 0; i < n / 2; ++i) x[i] = fmax(x[i], x[i + (n + 1) / 2]);
        }
        const double m = x[0];
        if (QUANT) {
          if (tieflag[p]) {
            // a rounding tie at this position: close the segment (record W_{t-1}) and restart from
            // zeros; the exact chain handles position t itself
            if (nt < TEHMM_SPEC_MAXT) {
              if (lane == 0) vc.ties[(int64_t)c * TEHMM_SPEC_MAXT + nt] = (int)(t - t0);
              if (lane < NT)
                vc.tierows[((int64_t)c * TEHMM_SPEC_MAXT + nt) * NT + lane] =
                    live ? W * 0.015625 + base : -INFINITY;
            }
            ++nt;
            W = live ? 0.0 : -INFINITY;
            base = 0.0;
          } else {
            // m = 64 * (best value) + (63 - first arg-max) * u  (exact)
            const double k = ldexp(m, 52 - e);                 // m / u, an integer
            const double r = k - 64.0 * floor(k * 0.015625);   // k mod 64 in [0, 63]
            const bool fin = m > -INFINITY;
            const int arg = fin ? 63 - (int)r : 0;
            W = fin ? (m - r * u) + b : -INFINITY;
            // W = 64 (value - base) + index bits must stay exactly representable: |W| < 2^53 u = 2^(e+1)
            if (live && W <= wlim && W > -INFINITY) bad = true;
            if (live) tb[(p0 + t) * NT + lane] = (uint8_t)arg;
          }
          if ((p & (TEHMM_VROW - 1)) == TEHMM_VROW - 1) {
            if ((p & 31) == 31) {
              // re-base so that the index bits keep fitting
              const double mx = wave_max_live(W, live);
              if (mx > -INFINITY) {
                W -= mx;
                base += mx * 0.015625;
              } else {
                bad = true;                                  // the whole vector died
              }
            }
            // record the row
            if (lane < NT)
              vc.rows[((int64_t)c * (vc.CS / TEHMM_VROW) + (t - t0) / TEHMM_VROW) * NT + lane] =
                  live ? W * 0.015625 + base : -INFINITY;
          }
        } else {
          W = m + b;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (QUANT) {
      const unsigned long long anybad = __ballot(bad);
      if (lane == 0) {
        vc.ntie[c] = nt;
        vc.ok[c] = (anybad || nt > TEHMM_SPEC_MAXT) ? 0 : 1;
      }
    } else {
      const double g = wave_max_live(W, live);
      if (lane == 0) vc.gain[c] = g;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Exact sequential chain with jumps over verified chunks.  Same roles, rings and arithmetic as
// k_vit_coop<NT, 64, false> (no segment ratios); the sequence of 64-position blocks is dynamic:
//   seqpos[it & 3] = first position of the block of iteration `it` (>= T: finished), published by
//   the chain wave at the check step of the previous block (gen = it) so that the emission wave
//   can prepare exactly that block.
// LDS (doubles): bring [3][64][RS] | Vring [2][65][VS] | ltab [lds_rows][NT] | seq (4 x i64, gen)
// ------------------------------------------------------------------------------------------
// SEGMIN: the chunk data come from the lane passes (tehmm_lane.hip.h): "stays in the binade up to the
// segment end" is decided from the segment's recorded minimum instead of the P0 gain estimate.
// RATIO: segment ratios on the transitions, in the reference's operation order incl. its from-state-0 quirk
// (_hmm.pyx:229-247, quirk Q4; same arithmetic as k_vit_coop<NT, CPB, true>).
template <int NT, bool SEGMIN, bool RATIO = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_vit_fix(IntervalTab iv, EmisTab em, VitChunks vc, int N, const double *g_lt,
               const double *g_ltT, const double *g_pi, uint8_t *tb, int *last_state,
               double *logprob, int *stats, const double *tratios = nullptr) {
  extern __shared__ double sm[];
  constexpr int RS = NT + 1;
  constexpr int VS = NT + 2;
  constexpr int CPB = 32;      // short blocks: a tie or a verification costs 32 exact steps, not 64
  double *bring = sm;
  double *Vring = bring + 3 * CPB * RS;
  double *ltab = Vring + 2 * (CPB + 1) * VS;
  volatile int64_t *seqpos = (volatile int64_t *)(ltab + em.lds_rows * NT);
  volatile int *gen = (volatile int *)(seqpos + 4);
  volatile int *blen = gen + 2;         // [4] positions the chain really ran in the block of iteration it & 3
  // the position a block WOULD jump to, published when the block starts (the decision falls 12..27 steps later):
  // the emission wave prepares that block meanwhile, so a verified jump does not wait for its rows
  volatile int64_t *tent = seqpos + 8;  // [4]
  volatile int *tgen = (volatile int *)(tent + 4);
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int id = iv.order[blockIdx.x];
  const int64_t T = iv.len[id];
  const int64_t p0 = iv.pos0[id];
  if (T <= 0) return;
  stage_emis_table(em, ltab, NT);
  if (threadIdx.x == 0) {
    seqpos[0] = 0;
    seqpos[1] = seqpos[2] = seqpos[3] = T;
    *gen = 0;
    blen[0] = blen[1] = blen[2] = blen[3] = 0;
    tent[0] = tent[1] = tent[2] = tent[3] = T;
    *tgen = 0;
  }
  const int jl = min(lane, NT - 1);
  const bool live = lane < N;
  const int64_t cfirst = vc.first[id];
  // latency kernel next to throughput kernels of the other pipeline: ask the SIMD arbiter to issue this
  // workgroup's waves first (the chain wave above its helpers)
  if (w == 0) __builtin_amdgcn_s_setprio(TEHMM_PRIO_HI);
  else __builtin_amdgcn_s_setprio(TEHMM_PRIO_LO);
  __syncthreads();
  bool seen = false;     // leading-rows quirk state of the emission wave (_emission.pyx:73-80)
  if (w == 1)            // prologue: emission rows of block 0
    emis_block<NT, false, true, false>(em, ltab, p0, (int)min((int64_t)CPB, T), lane, N, seen, nullptr,
                                       nullptr, nullptr, bring, RS, nullptr);
  __syncthreads();
  if (w == 0) {
    // ============================================================== value chain
    double ltc[NT];
#pragma unroll
    for (int f = 0; f < NT; ++f) ltc[f] = live ? g_lt[f * NT + jl] : -INFINITY;
    const double ltd = live ? g_lt[jl * NT + jl] : 0.0;
    const double lt00 = g_lt[0];
    const double pij = live ? g_pi[jl] : -INFINITY;
    const int vslot = lane < NT ? lane : NT + 1;
    double vcur = -INFINITY;
    int n_jump = 0, n_block = 0;
#ifdef TEHMM_CHAIN_PROF
    unsigned long long prof[4] = {0, 0, 0, 0}, pt = __builtin_amdgcn_s_memrealtime();
#define CPROF(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); prof[i] += n_ - pt; pt = n_; } while (0)
#else
#define CPROF(i)
#endif
    for (int it = 0;; ++it) {
      const int64_t cur = seqpos[it & 3];
      const int64_t prev = it > 0 ? seqpos[(it - 1) & 3] : T;
      if (cur >= T && prev >= T) break;
      if (cur < T) {
        ++n_block;
        const int np = (int)min((int64_t)CPB, T - cur);
        double *Vr = Vring + (it & 1) * (CPB + 1) * VS;
        const double *br = bring + (it % 3) * CPB * RS;
        if (cur > 0) Vr[vslot] = vcur;
        // chunk of this block and whether it can be verified against the speculative pass:
        // check row g = first recorded row (positions t0 + 32k + 31) at least 16 steps into the block;
        // jump target = next tie position behind g (its predecessor's W row was recorded) or the
        // chunk end.
        const int64_t c = cfirst + cur / vc.CS;
        const int64_t ct0 = vc.t0[c];
        const int e = vc.e[c];
        bool spec = e != TEHMM_SPEC_NONE && np == CPB && vc.ok[c] != 0;
        // smallest recorded row >= cur + TEHMM_FIX_MINSTEP (rank convergence takes 3-16 steps; a check that comes too
        // early simply fails and the next block tries again)
#ifndef TEHMM_FIX_MINSTEP
#define TEHMM_FIX_MINSTEP 12
#endif
        const int64_t g = ct0 + ((cur + TEHMM_FIX_MINSTEP - ct0) / TEHMM_VROW) * TEHMM_VROW + (TEHMM_VROW - 1);
        const int pg = (int)(g - cur);                          // in-block step of the check, 12..27
        // Where a verified jump lands depends (round 4, soft ties) on the parity h of the verified delta in units of
        // the grid: entry k of the chunk's tie list is passed iff it is a soft tie of that parity (tsoft / tpar), the
        // first entry that is not stops the jump; past the chunk end k_vit_runs has walked the linked run for either
        // parity.  Both plans are laid out before the block's steps; without soft ties they coincide.
        int64_t target = ct0 + vc.CS, target1 = ct0 + vc.CS;
        double smin = 0.0, smin1 = 0.0, lkacc = 0.0, lkacc1 = 0.0;
        const double *trow = vc.rows + ((c * (vc.CS / TEHMM_VROW)) + (vc.CS / TEHMM_VROW - 1)) * NT;
        const double *trow1 = trow;
        bool spec0 = false, spec1 = false;
        if (spec) {
          const int ntie = vc.ntie[c];
          const int *tl = vc.ties + c * TEHMM_SPEC_MAXT;
          int kseg = ntie;
          for (int k = 0; k < ntie; ++k)
            if (ct0 + tl[k] > g) { kseg = k; break; }
          const unsigned soft = (SEGMIN && vc.tsoft) ? vc.tsoft[c] : 0u, par = (SEGMIN && vc.tsoft) ? vc.tpar[c] : 0u;
          const int64_t ncn = vc.n;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            int64_t tg = ct0 + vc.CS;
            const double *tr = vc.rows + ((c * (vc.CS / TEHMM_VROW)) + (vc.CS / TEHMM_VROW - 1)) * NT;
            double sm = SEGMIN ? vc.segmin[c * (TEHMM_SPEC_MAXT + 1) + kseg] : 0.0, la = 0.0;
            int kh = -1;
            for (int k = kseg; k < ntie; ++k) {
              if (!((soft >> k) & 1u) || (int)((par >> k) & 1u) != h) { kh = k; break; }
              if (SEGMIN) sm = fmin(sm, vc.segmin[c * (TEHMM_SPEC_MAXT + 1) + k + 1]);
            }
            if (kh >= 0) {
              tg = ct0 + tl[kh];
              tr = vc.tierows + (c * TEHMM_SPEC_MAXT + kh) * NT;
            } else if (SEGMIN && c + 1 < vc.first[id + 1] && vc.clink[c + 1] != 0) {
              // the segment runs on through the following chunks (the lane passes link them exactly when they share
              // the binade): k_vit_runs has walked that run for the parity the chain arrives with
              const int64_t cn = c + 1;
              const double k1 = vc.clk[cn];
              const double qq = k1 * ldexp(1.0, 52 - e) * 0.5;
              const int h2 = h ^ (qq != trunc(qq) ? 1 : 0);
              const int64_t sel = vc.rsel[h2 * ncn + cn];
              la = k1 + vc.racc[h2 * ncn + cn];
              sm = fmin(sm, k1 + vc.rmn[h2 * ncn + cn]);
              tg = vc.rtarget[h2 * ncn + cn];
              const int64_t sc = (sel >> 1) >> 6, sk = (sel >> 1) & 63;
              tr = (sel & 1) ? vc.rows + ((sc * (vc.CS / TEHMM_VROW)) + (vc.CS / TEHMM_VROW - 1)) * NT
                             : vc.tierows + (sc * TEHMM_SPEC_MAXT + sk) * NT;
            }
            if (h == 0) { target = tg; trow = tr; smin = sm; lkacc = la; }
            else { target1 = tg; trow1 = tr; smin1 = sm; lkacc1 = la; }
          }
          const bool posok = !(pg >= CPB || g >= ct0 + vc.CS);
          spec0 = posok && target > cur + CPB;
          spec1 = posok && target1 > cur + CPB;
          spec = spec0 || spec1;
        }
        double wrow = 0.0, wend = 0.0, wend1 = 0.0, delta = 0.0;
        bool jump = false;
        int64_t jtarget = target;
        if (spec) {
          // (the emission wave prepares the farther landing position; if the parity decides otherwise it prepares again)
          if (lane == 0) { tent[(it + 1) & 3] = spec0 && (!spec1 || target >= target1) ? target : target1; *tgen = it + 1; }
          wrow = vc.rows[((c * (vc.CS / TEHMM_VROW)) + (g - ct0) / TEHMM_VROW) * NT + jl];
          wend = trow[jl] + lkacc;        // exact: multiples of u inside one binade
          wend1 = trow1[jl] + lkacc1;
        } else {
          if (lane == 0) { seqpos[(it + 1) & 3] = cur + np; *gen = it + 1; }
        }
        const double span = (spec && !SEGMIN) ? fabs(vc.gain[c]) * 1.01 + 256.0 : 0.0;
        int nlen = np;
#ifdef TEHMM_CHAIN_PROF
        asm volatile("" :: "v"(wrow), "v"(wend), "v"(smin));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        CPROF(0);
        for (int p = 0; p < np; ++p) {
          const int64_t t = cur + p;
          const double b = br[p * RS + jl];
          double r = 0.0;
          if (RATIO) r = tratios[p0 + t];
          double v;
          if (t == 0) {
            v = pij + b;
            if (RATIO && r > 1.) v += ltd * (r - 1.);
          } else {
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
            v = c1 > c0 ? c1 : c0;
          }
          vcur = v;
          Vr[(p + 1) * VS + vslot] = v;
          if (spec && p == pg) {
            // verified iff V - W is one constant over the live states, every V is in the binade the
            // chunk was quantised for, and stays in it up to the chunk end
            const double d = v - wrow;
            const bool both_dead = v == -INFINITY && wrow == -INFINITY;
            const double d0 = wave_max_live(both_dead ? -INFINITY : d, live);
            const bool same = !live || both_dead || d == d0;
            const bool inb = !live || both_dead || exp_of(-v) == e + 1;   // 2^e <= |v| < 2^(e+1)
            // parity of the constant in grid units decides which plan applies
            const double qh = d0 * ldexp(1.0, 52 - e) * 0.5;
            const bool hodd = d0 == d0 && d0 > -INFINITY && d0 < INFINITY && qh != trunc(qh);
            const bool planok = hodd ? spec1 : spec0;
            const double sminh = hodd ? smin1 : smin;
            jtarget = hodd ? target1 : target;
            const double vlow = SEGMIN ? sminh + d0 : wave_min_f64(live && !both_dead ? v : 0.0) - span;
            const bool endok = d0 == d0 && d0 > -INFINITY && exp_of(-vlow) == e + 1;
            jump = __all(same && inb) && endok && planok;
            delta = d0;
            if (hodd) wend = wend1;
            if (lane == 0) { seqpos[(it + 1) & 3] = jump ? jtarget : cur + np; *gen = it + 1; }
            // verified: the rest of the block belongs to the speculative pass too (its pointer bytes stand)
            if (jump) { nlen = p + 1; break; }
          }
        }
        if (lane == 0) blen[it & 3] = nlen;
        if (jump) {
          vcur = wend + delta;          // V at position target - 1 (exact: both multiples of u)
          ++n_jump;
        }
        CPROF(1);
      } else {
        if (lane == 0) { seqpos[(it + 1) & 3] = T; *gen = it + 1; }
      }
      __syncthreads();
      CPROF(2);
    }
#ifdef TEHMM_CHAIN_PROF
    if (lane == 0 && stats)
      for (int i = 0; i < 4; ++i) atomicAdd(&stats[8 + i], (int)(prof[i] / 10));      // 100 MHz ticks -> 0.1 us
#endif
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
      if (stats) { atomicAdd(&stats[0], n_block); atomicAdd(&stats[1], n_jump); }
    }
  } else if (w == 1) {
    // ============================================================== emission rows of the NEXT block
    for (int it = 0;; ++it) {
      const int64_t cur = seqpos[it & 3];
      const int64_t prev = it > 0 ? seqpos[(it - 1) & 3] : T;
      if (cur >= T && prev >= T) break;
      int spins = 0;
      while (*gen < it + 1 && *tgen < it + 1) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1 << 26)) break;      // never expected; bounded so that the grid drains
      }
      int64_t done = -1;                     // position whose rows already sit in the ring slot
      if (*gen < it + 1 && *tgen >= it + 1 && seen) {
        // only the tentative landing position is known yet: prepare it (every row up to it is emittable -- the
        // jump would not be offered otherwise -- so the leading-rows state `seen` is not disturbed)
        const int64_t tp = tent[(it + 1) & 3];
        if (tp < T) {
          const int np = (int)min((int64_t)CPB, T - tp);
          emis_block<NT, false, true, false>(em, ltab, p0 + tp, np, lane, N, seen, nullptr, nullptr,
                                             nullptr, bring + ((it + 1) % 3) * CPB * RS, RS, nullptr);
          done = tp;
        }
      }
      while (*gen < it + 1) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1 << 26)) break;
      }
      const int64_t nx = seqpos[(it + 1) & 3];
      if (nx < T && nx != done) {
        const int np = (int)min((int64_t)CPB, T - nx);
        emis_block<NT, false, true, false>(em, ltab, p0 + nx, np, lane, N, seen, nullptr, nullptr,
                                           nullptr, bring + ((it + 1) % 3) * CPB * RS, RS, nullptr);
      }
      __syncthreads();
    }
  } else {
    // ============================================================== exact arg-max of the PREVIOUS block
    const int half = (N + 1) / 2;
    const int to_lo = (w - 2) * half, to_hi = min(N, to_lo + half);
    const double lt00 = g_lt[0];
    for (int it = 0;; ++it) {
      const int64_t cur = seqpos[it & 3];
      const int64_t prev = it > 0 ? seqpos[(it - 1) & 3] : T;
      if (cur >= T && prev >= T) break;
      if (it > 0 && prev < T) {
        const int np = blen[(it - 1) & 3];
        const double *Vr = Vring + ((it - 1) & 1) * (CPB + 1) * VS;
        const double *br = bring + ((it - 1) % 3) * CPB * RS;
        const int pl = lane < np ? lane : 0;
        const int64_t t = prev + pl;
        const bool actp = lane < np && t > 0;
        lds_cd2 *vp = lds_row(Vr + pl * VS);
        d2v pv[NT / 2];
#pragma unroll
        for (int f2 = 0; f2 < NT / 2; ++f2) pv[f2] = vp[f2];
        const double *vnext = Vr + (pl + 1) * VS;
        const double *brow = br + pl * RS;
        double r = 0.0;
        if (RATIO) r = tratios[p0 + t];
        const bool rg = RATIO && r > 1.;
        for (int to = to_lo; to < to_hi; ++to) {
          const double *lc = g_ltT + to * NT;
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
      __syncthreads();
    }
  }
}


// ==========================================================================================
// Chunk-parallel forward / backward (scaled linear domain, tolerance 1e-6 on posteriors).
//
// The normalised forward (backward) vector forgets its starting point: a positive linear map never
// increases Hilbert's projective distance and these chains contract it by ~0.4 per step (measured:
// 1e-12 within 20-50 steps on the bench model).  So every chunk is first run from a UNIFORM vector
// (k_fb_spec, all chunks in parallel, rows written straight into the alpha / beta buffers), then
// the sequential chain (k_fb_fix) runs 64-position blocks exactly (overwriting those rows) and, at
// the 32nd step of each block, measures the projective distance between its vector and the
// speculative row: below 1e-10 it adopts the speculative row at the chunk end and jumps -- every
// later row of the chunk is then at least that close in direction, and posteriors only depend on
// directions (k_combine normalises each row).  The forward log-likelihood is carried across a jump
// by the speculative pass's own cumulative log-scale: S_end = S_g + log(rho) + (s'_end - s'_g).
// ==========================================================================================
struct FbChunks {
  const int *iv;          // chunk -> interval id
  const int64_t *t0;      // chunk -> first position
  const int64_t *first;   // interval id -> index of its first chunk
  int n, CS;
  double *scale;          // forward spec: [chunk][CS/32] cumulative log-scale at positions t0+32k+31
  double *wstart;         // backward spec: [chunk][NT] w row (bh' * beta) at the chunk's first position
  // lane passes only: runs of consecutive chunks whose item links all hold (k_fb_stitch / k_fb_runs)
  int *link_f;            // [chunk] 1 = the chunk's first item continues the previous chunk's last item
  double *glog_f;         // [chunk] log-scale gained over the chunk, in the frame of the previous chunk
  int *link_b;            // [chunk] 1 = the chunk's last item continues the next chunk's first item
  int *runend_f;          // [chunk] last chunk of the forward run that contains the chunk
  double *pre_f;          // [chunk] prefix sums of glog_f along the interval
  int *runstart_b;        // [chunk] first chunk of the backward run that contains the chunk
};
#define TEHMM_FB_TOL 1e-10

// item (sub-chunk) geometry of the lane = item passes, see tehmm_lane.hip.h
struct LaneGeom {
  const int *item_iv;       // item -> interval id
  const int64_t *item_t0;   // item -> first position inside the interval (multiple of L)
  const int64_t *ifirst;    // interval id -> its first item
  int n_items, n_groups, L;
};
// element (t_rel, state 0) of item `item` in an item-interleaved buffer; states are 64 doubles apart
__device__ __forceinline__ int64_t lane_row(const LaneGeom &lg, int NT, int64_t item, int64_t trel) {
  return ((((item >> 6) * lg.L + trel) * NT) << 6) + (item & 63);
}

// Fused passes (tehmm_fused.hip.h): the alpha' rows are kept in FLOAT (a posterior only needs their direction
// to 1e-6) in a layout where the nine states of a matrix-core lane sit in five float2:
//   element (item, t, state)  ->  [group][t][tile in group][state pair p = (state >> 2) >> 1][state & 3][item & 15][(state >> 2) & 1]
// i.e. one 512-byte row per wave load / store; 512 * P floats per (group, position), P = al32_pairs(NT)
// float2 per lane (5 at 36 states: 2560 floats).
__host__ __device__ constexpr int al32_pairs(int NT) { return (NT / 4 + 1) / 2; }
template <int NT>
__device__ __forceinline__ int64_t al32_index(const LaneGeom &lg, int64_t item, int64_t trel, int state) {
  constexpr int P = al32_pairs(NT);
  const int64_t gt = (item >> 6) * lg.L + trel;
  const int tg = (int)((item >> 4) & 3), s = state >> 2;
  return ((((gt * 4 + tg) * P + (s >> 1)) * 4 + (state & 3)) * 16 + (item & 15)) * 2 + (s & 1);
}

// Hilbert projective distance (as max/min ratio - 1) between two non-negative vectors over the live
// lanes; returns a huge value when their supports differ or anything is not finite.  rho = the ratio
// a / b at the lane where it is largest.
__device__ __forceinline__ double proj_dist(double a, double b, bool live, double &rho) {
  const bool both0 = a == 0.0 && b == 0.0;
  const bool okl = !live || both0 || (a > 0.0 && b > 0.0 && a < INFINITY && b < INFINITY);
  const double r = (live && !both0 && okl) ? a / b : -1.0;
  const double rmax = wave_max_f64(r);
  const double rmin = wave_min_f64(r < 0.0 ? INFINITY : r);
  rho = rmax;
  if (!__all(okl) || !(rmax > 0.0) || !(rmin < INFINITY)) return INFINITY;
  return rmax / rmin - 1.0;
}

// ------------------------------------------------------------------------------------------
// Speculative pass.  DIR = 0 forward (alpha rows + scale records), DIR = 1 backward (beta rows +
// the w row at the chunk start).  blockDim = 256, one chunk per wave at a time.
// Forward speculates chunks 1.. (chunk 0 starts from the true start vector in the fix-up chain),
// backward speculates every chunk but the interval's last one; ragged tails are not speculated.
// LDS (doubles): ring [4][64][RS] | ms [4][64]
// ------------------------------------------------------------------------------------------
template <int NT, int DIR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_fb_spec(IntervalTab iv, EmisTab em, FbChunks fc, int N,
                                                 const double *g_A, double *rows_out) {
  extern __shared__ double sm[];
  constexpr int RS = NT + 1;
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  double *ring = sm + w * 64 * RS;
  double *ms = sm + 4 * 64 * RS + w * 64;
  const int jl = min(lane, NT - 1);
  const bool live = lane < N;
  double ac[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
    ac[i] = live ? (DIR == 0 ? g_A[i * NT + jl] : g_A[jl * NT + i]) : (lane == NT - 1 ? 1.0 : 0.0);
  for (int c = blockIdx.x * 4 + w; c < fc.n; c += gridDim.x * 4) {
    const int id = fc.iv[c];
    const int64_t T = iv.len[id];
    const int64_t p0 = iv.pos0[id];
    const int64_t t0 = fc.t0[c];
    const int64_t t1 = t0 + fc.CS;
    if (t1 > T) continue;                                   // ragged tail: sequential
    if (DIR == 0 && c == fc.first[id]) continue;            // forward: first chunk is exact anyway
    if (DIR == 1 && t1 >= T) continue;                      // backward: last chunk is exact anyway
    double *out = rows_out + iv.out0[id] * N;
    double v = live ? 1.0 / (double)N : 0.0;                // forward: a_{t0-1}; backward: w_{t1}
    double s = 0.0;                                         // forward cumulative log-scale
    bool seen = true;
    if (DIR == 1) {
      // w_{t1} = bh'_{t1} * 1: one emission row (position t1) through the ring
      emis_block<NT, true, false, false>(em, nullptr, p0 + t1, 1, lane, N, seen, nullptr, nullptr, nullptr,
                                         ring, RS, ms);
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);
      v = live ? ring[jl] : 0.0;
      __builtin_amdgcn_wave_barrier();
    }
    for (int blk = 0; blk < fc.CS / 64; ++blk) {
      const int64_t lo = DIR == 0 ? t0 + 64 * blk : t1 - 64 * (blk + 1);
      if (DIR == 0)
        emis_block<NT, true, true, false>(em, nullptr, p0 + lo, 64, lane, N, seen, nullptr, nullptr, nullptr,
                                          ring, RS, ms);
      else
        emis_block<NT, true, false, false>(em, nullptr, p0 + lo, 64, lane, N, seen, nullptr, nullptr,
                                           nullptr, ring, RS, ms);
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);
      for (int q = 0; q < 64; ++q) {
        const int p = DIR == 0 ? q : 63 - q;
        const int64_t t = lo + p;
        const double bh = ring[p * RS + jl];
        double rr[(NT + 15) / 16];
        rep_rows<NT>(v, rr);
        double sacc[4] = {0.0, 0.0, 0.0, 0.0};
        asm volatile("s_nop 1" ::: "memory");
        BcastFma<0, NT>::run(rr, ac, sacc);
        const double ssum = (sacc[0] + sacc[1]) + (sacc[2] + sacc[3]);
        const int e = ((__builtin_amdgcn_readlane(__double2hiint(ssum), NT - 1) >> 20) & 0x7ff) - 1022;
        if (DIR == 0) {
          v = live ? ldexp(ssum * bh, -e) : 0.0;            // a_t
          s += (double)e * 0.6931471805599453 + ms[p];
          if (live) out[t * N + lane] = v;
          if ((p & 31) == 31 && lane == 0) fc.scale[(int64_t)c * (fc.CS / 32) + (t - t0) / 32] = s;
        } else {
          const double bt = live ? ldexp(ssum, -e) : 0.0;   // beta_t
          if (live) out[t * N + lane] = bt;
          v = live ? bh * bt : 0.0;                         // w_t
          if (t == t0 && lane < NT) fc.wstart[(int64_t)c * NT + lane] = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ------------------------------------------------------------------------------------------
// Warm-up probe (round 3).  How many steps the normalised forward (DIR = 0) / backward (DIR = 1) vector needs to
// forget its start is a property of the model AND the data: 20-50 steps on the dense bench model, several
// hundred with sparse (-1e100) or sticky (0.995) transitions -- where a fixed 64-step warm-up left thousands of
// item links unverified and the exact chain walking them (backward_chain 20.8 of 55 ms per 30 Mb, round 2).
// One wave per probe window: two chains over the SAME observations, one from the uniform vector the lane passes
// start from, one from a steeply skewed vector (ratio e^-j), until their Hilbert distance is below
// TEHMM_FB_TOL / 10; steps[w] = the number of steps that took (WMAX if it never got there).
// blockDim = 64.  LDS (doubles): ring [64][RS] | ms [64]
// ------------------------------------------------------------------------------------------
template <int NT, int DIR>
__global__ __launch_bounds__(64) void k_fb_probe(IntervalTab iv, EmisTab em, int N, const double *g_A, const int *p_iv,
                                                 const int64_t *p_t0, int WMAX, int *steps) {
  extern __shared__ double sm[];
  constexpr int RS = NT + 1;
  const int lane = threadIdx.x & 63;
  double *ring = sm;
  double *ms = sm + 64 * RS;
  const int jl = min(lane, NT - 1);
  const bool live = lane < N;
  double ac[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
    ac[i] = live ? (DIR == 0 ? g_A[i * NT + jl] : g_A[jl * NT + i]) : (lane == NT - 1 ? 1.0 : 0.0);
  const int id = p_iv[blockIdx.x];
  const int64_t p0 = iv.pos0[id];
  const int64_t t0 = p_t0[blockIdx.x];          // DIR 0: first position of the window; DIR 1: one past its last
  double u = live ? 1.0 / (double)N : 0.0;
  double v = live ? exp(-(double)lane) : 0.0;
  bool seen = true;
  int found = WMAX;
  for (int blk = 0; blk < WMAX / 64 && found == WMAX; ++blk) {
    const int64_t lo = DIR == 0 ? t0 + 64 * blk : t0 - 64 * (blk + 1);
    if (DIR == 0)
      emis_block<NT, true, true, false>(em, nullptr, p0 + lo, 64, lane, N, seen, nullptr, nullptr, nullptr, ring, RS, ms);
    else
      emis_block<NT, true, false, false>(em, nullptr, p0 + lo, 64, lane, N, seen, nullptr, nullptr, nullptr, ring, RS, ms);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int q = 0; q < 64; ++q) {
      const int p = DIR == 0 ? q : 63 - q;
      const double bh = ring[p * RS + jl];
      double x2[2] = {u, v};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        // forward: a_t = (A^T a_{t-1}) * bh;  backward: w_t = bh * (A w_{t+1})  -- same contraction either way
        double rr[(NT + 15) / 16];
        rep_rows<NT>(x2[c], rr);
        double sacc[4] = {0.0, 0.0, 0.0, 0.0};
        asm volatile("s_nop 1" ::: "memory");
        BcastFma<0, NT>::run(rr, ac, sacc);
        const double ssum = (sacc[0] + sacc[1]) + (sacc[2] + sacc[3]);
        const int e = ((__builtin_amdgcn_readlane(__double2hiint(ssum), NT - 1) >> 20) & 0x7ff) - 1022;
        x2[c] = live ? ldexp(ssum * bh, -e) : 0.0;
      }
      u = x2[0];
      v = x2[1];
      double rho;
      const double d = proj_dist(u, v, live, rho);
      if (d <= 0.1 * TEHMM_FB_TOL || !(d == d)) {        // converged (or an impossible row: nothing to learn here)
        found = 64 * blk + q + 1;
        break;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) steps[blockIdx.x] = found;
}

// ------------------------------------------------------------------------------------------
// Sequential chain with verified jumps.  blockDim = 128: wave 0 chain, wave 1 emission rows of the
// next block (same publish / wait protocol as k_vit_fix).  DIR = 0 forward: seqpos = first position
// of a block (ascending); DIR = 1 backward: seqpos = one past the LAST position of a block
// (descending; <= 0: finished).
// LDS (doubles): ring [2][64][RS] | ms [2][64] | ltd [NT] | ltab [lds_rows][NT] | seq
// ------------------------------------------------------------------------------------------
// LANE: `rows` is item-interleaved (tehmm_lane.hip.h) and okc[c] tells whether the chunk's item links
// hold; otherwise rows is [T][N] as written by k_fb_spec.
// FUSED (with the fused lane passes of tehmm_fused.hip.h): the alpha' rows live in al32 (float, al32_index);
// speculative rows in fp64 exist only at the check positions (chk [item][L / 64][NT]) and at the item ends
// (endv, the lane pass's end vectors).  Forward: exact rows are written to al32, a jump adopts endv.
// Backward: al32 is read only and the chain writes the posterior rows normalise(alpha' * beta) of its exact
// blocks (+ eps quirk) to post.
// ESTEP (backward chain of the fused E-step): the exact blocks leave gamma and wz rows as floats in the alpha'
// layout (gam32 / wz32, see k_fused_bwd) instead of posterior rows; LDS grows by one [64][RS] block (the
// unnormalised wz rows of the block being walked).
template <int NT, int DIR, bool TRATIO, bool LANE, bool FUSED = false, bool ESTEP = false>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_fb_fix(IntervalTab iv, EmisTab em, FbChunks fc, int N, const double *g_A, const double *g_lt,
              const double *g_pi, const double *tratios, double *rows, double *fwd_logprob,
              int *dead_flag, double *wrows, int *escale, int allow_jump, int *stats, LaneGeom lg,
              const int *okc, const double *chk = nullptr, double *post = nullptr, float *al32 = nullptr,
              const double *endv = nullptr, float *gam32 = nullptr, float *wz32 = nullptr) {
  extern __shared__ double sm[];
  constexpr int RS = NT + 1;
  constexpr int CPB = 64;
  double *ring = sm;
  double *msr = ring + 2 * CPB * RS;
  double *ltdv = msr + 2 * CPB;
  double *ltab = ltdv + NT;
  volatile int64_t *seqpos = (volatile int64_t *)(ltab + em.lds_rows * NT);
  volatile int *gen = (volatile int *)(seqpos + 4);
  double *wzu = (double *)(seqpos + 8);          // ESTEP: [64][RS]
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
  // a block is identified by `key`: DIR 0 -> its first position, DIR 1 -> one past its last position
  const int64_t END = DIR == 0 ? T : 0;
  if (threadIdx.x == 0) {
    seqpos[0] = DIR == 0 ? 0 : T;
    seqpos[1] = seqpos[2] = seqpos[3] = END;
    *gen = 0;
  }
  const int64_t cfirst = fc.first[id];
  if (w == 0) __builtin_amdgcn_s_setprio(TEHMM_PRIO_HI);       // see k_vit_fix
  else __builtin_amdgcn_s_setprio(TEHMM_PRIO_LO);
  __syncthreads();
  bool seen = false;
  // blocks are aligned to multiples of 64 in both directions (the backward chain starts with the
  // ragged piece [64*floor((T-1)/64), T)), so every block lies inside one chunk
  auto block_lo = [&](int64_t key) { return DIR == 0 ? key : ((key - 1) / CPB) * CPB; };
  auto block_np = [&](int64_t key) {
    return (int)(DIR == 0 ? min((int64_t)CPB, T - key) : key - ((key - 1) / CPB) * CPB);
  };
  auto done = [&](int64_t key) { return DIR == 0 ? key >= T : key <= 0; };
  auto emit = [&](int64_t key, int slot) {
    const int64_t lo = block_lo(key);
    const int np = block_np(key);
    if (DIR == 0)
      emis_block<NT, true, true, TRATIO>(em, ltab, p0 + lo, np, lane, N, seen, dead_flag + id, ltdv, tratios,
                                         ring + slot * CPB * RS, RS, msr + slot * CPB);
    else {
      bool dummy = true;
      emis_block<NT, true, false, TRATIO>(em, ltab, p0 + lo, np, lane, N, dummy, nullptr, ltdv, tratios,
                                          ring + slot * CPB * RS, RS, msr + slot * CPB);
    }
  };
  if (w == 1) emit(seqpos[0], 0);      // prologue: rows of the first block
  __syncthreads();
  if (w == 0) {
    double ac[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
      ac[i] = live ? (DIR == 0 ? g_A[i * NT + jl] : g_A[jl * NT + i]) : (lane == NT - 1 ? 1.0 : 0.0);
    double *out = LANE ? rows : rows + iv.out0[id] * N;
    const int64_t ifirst = LANE ? lg.ifirst[id] : 0;
    // element of this lane in the row of position t
    auto rowp = [&](int64_t t) -> double * {
      if (LANE) return out + lane_row(lg, NT, ifirst + t / lg.L, t % lg.L) + ((int64_t)jl << 6);
      return out + t * N + lane;
    };
    int *es = (DIR == 0 && escale) ? escale + iv.out0[id] : nullptr;
    double *wr = (DIR == 1 && wrows) ? wrows + iv.out0[id] * N : nullptr;
    double Ecum = 0.0, Mcum = 0.0;
    double v = 0.0;          // forward: a_t;  backward: w_t = bh'_t * beta_t
    int n_block = 0, n_jump = 0;
    for (int it = 0;; ++it) {
      const int64_t cur = seqpos[it & 3];
      if (done(cur)) break;
      if (it > (1 << 22)) { if (stats && lane == 0) atomicAdd(&stats[6], 1); break; }   // never expected: the grid must drain
      ++n_block;
      const int64_t lo = block_lo(cur);
      const int np = block_np(cur);
      const double *br = ring + (it & 1) * CPB * RS;
      const double *mr = msr + (it & 1) * CPB;
      // chunk of this block; every full block inside a speculated chunk (but the chunk's last one
      // in chain direction) verifies at its 32nd step and may jump to the chunk end
      const int64_t c = cfirst + lo / fc.CS;
      const int64_t ct0 = fc.t0[c];
      const bool cfull = ct0 + fc.CS <= T;
      const bool spec = allow_jump && np == CPB && cfull && (!LANE || okc[c] != 0) &&
                        (DIR == 0 ? (c != cfirst && lo + CPB < ct0 + fc.CS) : (ct0 + fc.CS < T && lo > ct0));
      double *brow = rowp(lo);                                  // row of position lo (blocks never straddle items)
      const int64_t rstride = LANE ? (int64_t)NT << 6 : (int64_t)N;
      const int pg = DIR == 0 ? 31 : 32;                        // check step inside the block
      const int64_t tg = DIR == 0 ? lo + pg : cur - 1 - pg;     // its position
      // LANE: a jump runs to the end of the whole run of linked chunks, not just of this chunk
      int64_t cj = c;
      if (LANE) {                                   // (clamped: a jump can only move in chain direction)
        cj = DIR == 0 ? max(c, (int64_t)fc.runend_f[c]) : min(c, (int64_t)fc.runstart_b[c]);
        cj = DIR == 0 ? min(cj, (int64_t)fc.first[id + 1] - 1) : max(cj, cfirst);
      }
      const int64_t target = DIR == 0 ? fc.t0[cj] + fc.CS : fc.t0[cj];      // key of the block after a jump
      double srow = 0.0;
      if (spec) {                                               // speculative row, before overwriting
        if (FUSED)
          srow = live ? chk[((ifirst + tg / lg.L) * (lg.L / 64) + (tg % lg.L) / 64) * NT + jl] : 0.0;
        else
          srow = live ? brow[(tg - lo) * rstride] : 0.0;
      }
      else if (lane == 0) {
        seqpos[(it + 1) & 3] = DIR == 0 ? cur + np : lo;
        *gen = it + 1;
      }
      bool jump = false;
      double Sg = 0.0, rho = 1.0;
      for (int q = 0; q < np; ++q) {
        const int p = DIR == 0 ? q : np - 1 - q;
        const int64_t t = lo + p;
        const double bh = br[p * RS + jl];
        if (DIR == 0) {
          Mcum += mr[p];
          if (t == 0) {
            v = live ? exp(g_pi[jl]) * bh : 0.0;
          } else {
            double rr[(NT + 15) / 16];
            rep_rows<NT>(v, rr);
            double sacc[4] = {0.0, 0.0, 0.0, 0.0};
            asm volatile("s_nop 1" ::: "memory");
            BcastFma<0, NT>::run(rr, ac, sacc);
            const double ssum = (sacc[0] + sacc[1]) + (sacc[2] + sacc[3]);
            const int e = ((__builtin_amdgcn_readlane(__double2hiint(ssum), NT - 1) >> 20) & 0x7ff) - 1022;
            v = live ? ldexp(ssum * bh, -e) : 0.0;
            Ecum += (double)e;
            if (es && lane == 0) es[t] = e;
          }
          if (FUSED) {
            // (the padding states too: the backward lane pass multiplies them with its zero beta, and a stale
            //  NaN / Inf bit pattern in a slot nobody wrote would turn the row's normalisation into NaN)
            if (lane < NT) al32[al32_index<NT>(lg, ifirst + t / lg.L, t % lg.L, lane)] = live ? (float)v : 0.f;
          } else if (live) {
            brow[p * rstride] = v;
          }
        } else {
          double bt;
          if (t == T - 1) {
            bt = live ? 1.0 : 0.0;
            if (ESTEP && lane < NT) wzu[p * RS + lane] = 0.0;                 // no transition leaves the last position
          } else {
            double rr[(NT + 15) / 16];
            rep_rows<NT>(v, rr);
            double sacc[4] = {0.0, 0.0, 0.0, 0.0};
            asm volatile("s_nop 1" ::: "memory");
            BcastFma<0, NT>::run(rr, ac, sacc);
            const double ssum = (sacc[0] + sacc[1]) + (sacc[2] + sacc[3]);
            const int e = ((__builtin_amdgcn_readlane(__double2hiint(ssum), NT - 1) >> 20) & 0x7ff) - 1022;
            bt = live ? ldexp(ssum, -e) : 0.0;
            if (ESTEP && lane < NT) wzu[p * RS + lane] = live ? ldexp(v, -e) : 0.0;   // w_{t+1} * scale_t
          }
          if (FUSED) {
            if (lane < NT) const_cast<double *>(br)[p * RS + lane] = bt;      // (the bh slot of this step is free)
          } else if (live) {
            brow[p * rstride] = bt;
          }
          if (spec && q == pg) {
            const double d = proj_dist(bt, srow, live, rho);
            jump = d <= TEHMM_FB_TOL;
            if (lane == 0) { seqpos[(it + 1) & 3] = jump ? target : lo; *gen = it + 1; }
          }
          v = live ? bh * bt : 0.0;
          if (wr && live) wr[t * N + lane] = v;
        }
        if (DIR == 0 && spec && q == pg) {
          const double d = proj_dist(v, srow, live, rho);
          jump = d <= TEHMM_FB_TOL;
          Sg = Ecum * 0.6931471805599453 + Mcum;
          if (lane == 0) { seqpos[(it + 1) & 3] = jump ? target : cur + np; *gen = it + 1; }
        }
      }
      if (FUSED && DIR == 1 && ESTEP) {
        // gamma and wz rows of this exact block (floats, alpha' layout; the padding states get zeros)
        for (int p = 0; p < np; ++p) {
          const int64_t tp = lo + p;
          const int64_t ai = al32_index<NT>(lg, ifirst + tp / lg.L, tp % lg.L, jl);
          const double a = live ? (double)al32[ai] : 0.0;
          const double g = a * (live ? br[p * RS + lane] : 0.0);
          const double tot = wave_sum_f64(g);
          if (lane < NT) {
            gam32[ai] = live ? (float)(g / tot) : 0.f;
            wz32[ai] = live ? (float)(wzu[p * RS + lane] / tot) : 0.f;
          }
        }
      } else if (FUSED && DIR == 1) {
        // posterior rows of this exact block: lane = state, one wave reduction per row
        const double eps = 1.1920928955078125e-07;
        const double inv_epsden = 1.0 / (1.0 + (double)N * eps);
        double *po = post + (iv.out0[id] + lo) * N;
        for (int p = 0; p < np; ++p) {
          const int64_t tp = lo + p;
          const double a = live ? (double)al32[al32_index<NT>(lg, ifirst + tp / lg.L, tp % lg.L, jl)] : 0.0;
          const double g = a * (live ? br[p * RS + lane] : 0.0);
          const double tot = wave_sum_f64(g);
          if (live) po[(int64_t)p * N + lane] = (g / tot + eps) * inv_epsden;
        }
      }
      if (jump) {
        ++n_jump;
        if (DIR == 0) {
          // adopt the speculative row at the chunk end; carry the log-likelihood over the jump
          const int64_t tl = target - 1;
          if (FUSED) {
            const int64_t it_end = ifirst + tl / lg.L;              // tl is the last position of its item
            v = live ? endv[((((it_end >> 6) * NT) + jl) << 6) + (it_end & 63)] : 0.0;
          } else {
            v = live ? *rowp(tl) : 0.0;
          }
          const double *sc = fc.scale + c * (fc.CS / 32);
          Mcum = Sg + log(rho) + (sc[fc.CS / 32 - 1] - sc[(tg - ct0) / 32]);
          if (LANE) Mcum += fc.pre_f[cj] - fc.pre_f[c];
          Ecum = 0.0;
        } else {
          v = live ? fc.wstart[cj * NT + jl] : 0.0;
        }
      }
      __syncthreads();
    }
    if (DIR == 0) {
      const double tot = wave_sum_f64(live ? v : 0.0);
      if (lane == 0) fwd_logprob[id] = log(tot) + Ecum * 0.6931471805599453 + Mcum;
    }
    if (stats && lane == 0) { atomicAdd(&stats[2 + 2 * DIR], n_block); atomicAdd(&stats[3 + 2 * DIR], n_jump); }
  } else {
    for (int it = 0;; ++it) {
      const int64_t cur = seqpos[it & 3];
      if (done(cur)) break;
      if (it > (1 << 22)) break;
      int spins = 0;
      while (*gen < it + 1) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1 << 26)) break;
      }
      const int64_t nx = seqpos[(it + 1) & 3];
      if (!done(nx)) emit(nx, (it + 1) & 1);
      __syncthreads();
    }
  }
}

}  // namespace tehmm
