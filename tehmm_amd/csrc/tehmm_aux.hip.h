// tehmm_aux.hip.h -- the steps either side of the DP path, as device kernels:
//   * segment compression of a track table (TrackTable.segment: interpolateSegments + compressSegments,
//     track.py:449-533, 594-620) and mask compaction (IntegerTrackTable.setMaskTable /
//     getMaskRunningOffsets, track.py:622-662, with _track.runSum, _track.pyx:13-25);
//   * the Baum-Welch M-step on device-resident statistics (MultitrackHmm._do_mstep, hmm.py:576-616;
//     emission.maximize, emission.py:243-267; the gaussian refit, emission.py:502-593) and the rebuild of
//     every table layout the DP kernels read.
// All of it is byte / small-table work: coalesced streaming, LDS histograms, nothing for the matrix cores.
#pragma once
#include "tehmm_kernels.hip.h"

namespace tehmm {

// ------------------------------------------------------------------------------------------
// Segment compression.  Segment i covers table rows [off[i], off[i+1]) (the last one runs to T).
//   categorical tracks: out[i][k] = mode of data[off[i] : end, k] (lowest symbol among the most frequent,
//                       scipy.stats.mode as called at track.py:619);
//   gaussian tracks   : mean[i][k] = (sum of mapback[data[x][k]], x ascending) / length -- the caller
//                       maps the mean back to a symbol (CategoryMap.getMap(mean, update=True) may create
//                       a new one, track.py:612-616); out[i][k] keeps the segment's first value.
// One wave per segment; a 256-bin LDS histogram per wave serves every categorical track in turn.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_segment_table(int64_t T, int K, const uint8_t *data, int64_t n_seg,
                                                       const int64_t *off, const uint8_t *is_gauss,
                                                       const double *mapback, uint8_t *out, double *means) {
  __shared__ unsigned hist[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t i = (int64_t)blockIdx.x * 4 + w; i < n_seg; i += nw) {
    const int64_t a = off[i], b = i + 1 < n_seg ? off[i + 1] : T;
    for (int k = 0; k < K; ++k) {
      if (is_gauss[k]) {
        if (lane == 0) {
          const double *mb = mapback + (int64_t)k * 256;
          double total = 0.0;
          for (int64_t x = a; x < b; ++x) total += mb[data[x * K + k]];
          means[i * K + k] = total / (double)(b - a);
          out[i * K + k] = data[a * K + k];
        }
        continue;
      }
      unsigned *h = hist[w];
#pragma unroll
      for (int q = 0; q < 4; ++q) h[lane * 4 + q] = 0;
      __builtin_amdgcn_wave_barrier();
      for (int64_t x = a + lane; x < b; x += 64) atomicAdd(&h[data[x * K + k]], 1u);
      __builtin_amdgcn_wave_barrier();
      // best = highest count, lowest symbol among equals: key = count * 256 + (255 - symbol)
      unsigned long long best = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int sym = lane * 4 + q;
        const unsigned long long key = (unsigned long long)h[sym] * 256ull + (unsigned long long)(255 - sym);
        best = key > best ? key : best;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o);
        best = other > best ? other : best;
      }
      if (lane == 0) {
        out[i * K + k] = (uint8_t)(255 - (int)(best & 255ull));
        means[i * K + k] = 0.0;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ------------------------------------------------------------------------------------------
// Mask compaction.  keep[i] = (sum of the mask tracks at i) <= KM  (BinaryMap: False = 1, True = 2, so any
// covering mask track makes the sum exceed KM, track.py:629-634); run_full[i] = number of cut positions
// before i (runSum, _track.pyx:13-25); kept rows are packed to out_data / run_masked[i - run_full[i]].
// Three passes: flags + per-block counts, scan of the block counts (one workgroup), scan + scatter.
// ------------------------------------------------------------------------------------------
#define TEHMM_SCAN_BLOCK 2048      // positions per workgroup (256 threads x 8)

__global__ __launch_bounds__(256) void k_mask_flags(int64_t T, int KM, const uint8_t *mask, uint8_t *keep,
                                                    unsigned *block_cut) {
  __shared__ unsigned wsum[4];
  const int64_t base = (int64_t)blockIdx.x * TEHMM_SCAN_BLOCK;
  unsigned cut = 0;
  for (int q = 0; q < 8; ++q) {
    const int64_t i = base + q * 256 + threadIdx.x;
    if (i < T) {
      int s = 0;
      for (int m = 0; m < KM; ++m) s += mask[i * KM + m];
      const bool kp = s <= KM;
      keep[i] = kp ? 1 : 0;
      cut += kp ? 0u : 1u;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cut += __shfl_xor(cut, o);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cut;
  __syncthreads();
  if (threadIdx.x == 0) block_cut[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the block counts in place (one workgroup; 64-bit running total in *total_cut)
__global__ __launch_bounds__(256) void k_scan_blocks(int64_t n_blocks, unsigned *block_cut, int64_t *total_cut) {
  __shared__ unsigned sh[256];
  __shared__ unsigned carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t b0 = 0; b0 < n_blocks; b0 += 256) {
    const int64_t i = b0 + threadIdx.x;
    const unsigned v = i < n_blocks ? block_cut[i] : 0u;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const unsigned add = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0u;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    const unsigned incl = sh[threadIdx.x], carry = carry_s;
    if (i < n_blocks) block_cut[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_cut = (int64_t)carry_s;
}

__global__ __launch_bounds__(256) void k_mask_scatter(int64_t T, int K, const uint8_t *data, const uint8_t *keep,
                                                      const unsigned *block_cut, int32_t *run_full,
                                                      uint8_t *out_data, int32_t *run_masked) {
  __shared__ unsigned wtot[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * TEHMM_SCAN_BLOCK;
  unsigned carry = block_cut[blockIdx.x];
  // thread t owns the 8 consecutive positions base + 8 t .. base + 8 t + 7
  const int64_t i0 = base + (int64_t)threadIdx.x * 8;
  unsigned c[8], mine = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    c[q] = (i0 + q < T && keep[i0 + q] == 0) ? 1u : 0u;
    mine += c[q];
  }
  unsigned incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) wtot[w] = incl;
  __syncthreads();
  unsigned before = carry + incl - mine;
  for (int q = 0; q < w; ++q) before += wtot[q];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int64_t i = i0 + q;
    if (i < T) {
      run_full[i] = (int32_t)before;
      if (!c[q]) {
        const int64_t o = i - (int64_t)before;
        run_masked[o] = (int32_t)before;
        for (int k = 0; k < K; ++k) out_data[o * K + k] = data[i * K + k];
      }
      before += c[q];
    }
  }
}

// ------------------------------------------------------------------------------------------
// M-step on device-resident statistics.
// Raw statistics buffer (doubles), summed over batches / ranks before the M-step:
//   [0] forward log-likelihood sum   [1] number of sequences
//   start[NP] | C[NP][NP] | D[NP] | stat[(R + 1)][NP]        (offsets: stats_off_*)
// with trans statistics = (A o C + diag(D)) / N (k_estep_accum, tehmm_coop.hip.h).
// ------------------------------------------------------------------------------------------
__host__ __device__ inline int64_t stats_off_start() { return 2; }
__host__ __device__ inline int64_t stats_off_C(int NP) { return 2 + NP; }
__host__ __device__ inline int64_t stats_off_D(int NP) { return 2 + NP + (int64_t)NP * NP; }
__host__ __device__ inline int64_t stats_off_stat(int NP) { return 2 + 2 * (int64_t)NP + (int64_t)NP * NP; }
__host__ __device__ inline int64_t stats_size(int NP, int R) { return stats_off_stat(NP) + (int64_t)(R + 1) * NP; }

#define TEHMM_DBL_EPS 2.220446049250313e-16      // common.py:25 EPSILON = np.finfo(float).eps

__device__ __forceinline__ double my_log(double x, double log_zero) {      // common.py:27-30
  return fabs(x) < TEHMM_DBL_EPS ? log_zero : log(x);
}

// transitions + start: one thread per state (row).  lt [NP][NP] in / out, pi [NP] in / out.
__global__ void k_mstep_trans(int N, int NP, const double *stats, int do_start, int do_trans,
                              double start_prior, double trans_prior, double *lt, double *pi) {
  const int i = threadIdx.x;
  __shared__ double sh[128];
  if (do_start) {
    // normalize(np.maximum(prior - 1 + start, 1e-20)): += eps, / sum  (hmm.py:585-587, common normalize)
    double v = 0.0;
    if (i < N) v = fmax(start_prior - 1.0 + stats[stats_off_start() + i], 1e-20) + TEHMM_DBL_EPS;
    sh[i] = i < N ? v : 0.0;
    __syncthreads();
    double tot = 0.0;
    for (int j = 0; j < N; ++j) tot += sh[j];
    if (i < N) pi[i] = my_log(v / tot, -1e100);
    __syncthreads();
  }
  if (do_trans && i < N) {
    const double *C = stats + stats_off_C(NP), *D = stats + stats_off_D(NP);
    const double invN = 1.0 / (double)N;
    double rowsum = 0.0;
    for (int j = 0; j < N; ++j) {
      double v = exp(lt[i * NP + j]) * C[i * NP + j];
      if (i == j) v += D[i];
      rowsum += trans_prior - 1.0 + v * invN;
    }
    if (!(rowsum < TEHMM_DBL_EPS)) {           // an orphaned state keeps last iteration's row (hmm.py:594-597)
      for (int j = 0; j < N; ++j) {
        double v = exp(lt[i * NP + j]) * C[i * NP + j];
        if (i == j) v += D[i];
        lt[i * NP + j] = my_log((trans_prior - 1.0 + v * invN) / rowsum, -1e100);
      }
    }
  }
}

// emission.maximize (emission.py:243-267): one thread per (track, state).  tab rows [(rowbase[k] + s)][NP];
// symbols 1 .. rowcnt[k] - 1 are the track's symbols (0 = missing data, untouched).
struct MstepTracks {
  int K;
  int rowbase[TEHMM_MAX_TRACKS];
  int rowcnt[TEHMM_MAX_TRACKS];
};

__global__ void k_mstep_emis(MstepTracks tr, int N, int NP, const double *stats, double fudge, double *tab) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= tr.K * N) return;
  const int k = idx / N, j = idx - k * N;
  const double *st = stats + stats_off_stat(NP);
  const int rb = tr.rowbase[k], cnt = tr.rowcnt[k];
  double total = 0.0;
  for (int s = 1; s < cnt; ++s) total += fudge + st[(int64_t)(rb + s) * NP + j];
  const double denom = fmax(fudge, total);
  double track_sum = 0.0;
  for (int s = 1; s < cnt; ++s) {
    const double p = denom != 0.0 ? (fudge + st[(int64_t)(rb + s) * NP + j]) / denom : 0.0;
    track_sum += p;
  }
  if (track_sum < TEHMM_DBL_EPS) return;        // orphaned state / track: the row stays as it was
  for (int s = 1; s < cnt; ++s) {
    const double p = denom != 0.0 ? (fudge + st[(int64_t)(rb + s) * NP + j]) / denom : 0.0;
    tab[(int64_t)(rb + s) * NP + j] = my_log(p, -1e6);
  }
}

// gaussian refit (emission.py:502-584): mu / sigma from the multinomial row, table = uniform mix + pdf,
// renormalised.  One thread per (gaussian track, state).  values [n_gauss][S]: the real value of each
// symbol (CategoryMap.getMapBack); gparams out [n_gauss][N][2].
__global__ void k_mstep_gauss(MstepTracks tr, int n_gauss, const int *gtracks, const double *values, int S,
                              int N, int NP, double uniform_mix, double *tab, double *gparams) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_gauss * N) return;
  const int g = idx / N, j = idx - g * N;
  const int k = gtracks[g];
  const int rb = tr.rowbase[k], cnt = tr.rowcnt[k];
  const double *val = values + (int64_t)g * S;
  double mu = 0.0;
  for (int s = 1; s < cnt; ++s) mu += val[s] * exp(tab[(int64_t)(rb + s) * NP + j]);
  double var = 0.0;
  for (int s = 1; s < cnt; ++s) {
    const double d = val[s] - mu;
    var += (d * d) * exp(tab[(int64_t)(rb + s) * NP + j]);
  }
  const double sigma = fmax(sqrt(var), TEHMM_DBL_EPS);
  gparams[((int64_t)g * N + j) * 2] = mu;
  gparams[((int64_t)g * N + j) * 2 + 1] = sigma;
  const double uniform = 1.0 / (double)(cnt - 1) * uniform_mix;
  const double inv_norm = 1.0 / (2.5066282746310002 * sigma);       // sqrt(2 pi)
  double tot = 0.0;
  for (int s = 1; s < cnt; ++s) {
    const double z = (val[s] - mu) / sigma;
    const double p = uniform + (1.0 - uniform_mix) * (exp(-0.5 * z * z) * inv_norm);
    const double lp = my_log(p, -1e100);
    tab[(int64_t)(rb + s) * NP + j] = lp;
    tot += exp(lp);
  }
  for (int s = 1; s < cnt; ++s)
    tab[(int64_t)(rb + s) * NP + j] = my_log(exp(tab[(int64_t)(rb + s) * NP + j]) / tot, -1e100);
}

// every other layout of the transition table from lt [NP][NP] (pads -inf): transposes, linear copies,
// output-group-major copies for the lane = item kernels, float pairs for the packed P0 pass
__global__ void k_model_layouts(int NP, const double *lt, double *ltT, double *A, double *AT, double *ltG,
                                double *AG, double *ATG, float *ltP) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NP * NP) return;
  const int f = idx / NP, o = idx - f * NP;
  const double l = lt[idx];
  const double a = exp(l);                       // pads: exp(-inf) = 0
  ltT[o * NP + f] = l;
  A[idx] = a;
  AT[o * NP + f] = a;
  const int tg = (((o >> 2) * NP + f) * 4) + (o & 3);
  const int tgT = (((f >> 2) * NP + o) * 4) + (f & 3);
  ltG[tg] = l;
  AG[tg] = a;
  ATG[tgT] = a;                                  // group-major copy of A^T: entry [from = o][out = f]
  ltP[((o >> 1) * NP + f) * 2 + (o & 1)] = (float)l;
  // behind the table: lt[o][o] per output (pads 0) and lt[0][0], for the segment-ratio terms of P0
  if (f == o) ltP[NP * NP + o] = l > -INFINITY ? (float)l : 0.f;
  if (idx == 0) ltP[NP * NP + NP] = (float)l;
}

// the LDS-staged copy of the small tracks' rows
__global__ void k_model_ltab(MstepTracks tr, const int *ldsbase, int NP, const double *tab, double *ltab) {
  const int k = blockIdx.x;
  if (k >= tr.K || ldsbase[k] < 0) return;
  for (int i = threadIdx.x; i < tr.rowcnt[k] * NP; i += blockDim.x)
    ltab[(int64_t)ldsbase[k] * NP + i] = tab[(int64_t)tr.rowbase[k] * NP + i];
}

// Table rows of the fused forward / backward passes (tehmm_fused.hip.h): row r of tab [NP] (log) ->
// ptab [r][4][KSP]: quarter kq holds the states kq, kq + 4, ... and, in its last slot, the row's scale
// c = max_j tab[r][j];  entries are exp(tab - c) (log_domain: the log values themselves, slot unused).
// ldsrow[r] >= 0: the row is also written to the LDS-staged copy at that index.
__global__ void k_build_ptab(int R1, int N, int NP, int KSP, const double *tab, int log_domain, double *ptab) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R1) return;
  const double *row = tab + (int64_t)r * NP;
  double c = -INFINITY;
  for (int j = 0; j < N; ++j) c = fmax(c, row[j]);
  const bool dead = !(c > -INFINITY);
  double *o = ptab + (int64_t)r * 4 * KSP;
  const int KS = NP / 4;
  for (int kq = 0; kq < 4; ++kq) {
    for (int s = 0; s < KSP; ++s) {
      const int j = kq + 4 * s;
      double v = 0.0;
      if (s < KS && j < N) v = log_domain ? row[j] : (dead ? 0.0 : exp(row[j] - c));
      else if (s < KS && log_domain) v = -INFINITY;
      else if (s == KS) v = (log_domain || dead) ? 0.0 : c;
      o[kq * KSP + s] = v;
    }
  }
}
// (rows of the LDS copy are ROW_L >= ROW_D doubles apart: bank spreading, see FusedGeom)
__global__ void k_pack_ptab_lds(int K, const int *rowbase, const int *rowcnt, const int *ldsbase, int lds_zero,
                                int zero_row, int ROW_D, int ROW_L, const double *ptab, double *ptab_lds) {
  const int k = blockIdx.x;          // block K: the zero row
  int src, dst, cnt;
  if (k == K) { src = zero_row; dst = lds_zero; cnt = 1; }
  else { if (ldsbase[k] < 0) return; src = rowbase[k]; dst = ldsbase[k]; cnt = rowcnt[k]; }
  for (int i = threadIdx.x; i < cnt * ROW_L; i += blockDim.x) {
    const int r = i / ROW_L, c = i - r * ROW_L;
    ptab_lds[(int64_t)(dst + r) * ROW_L + c] = c < ROW_D ? ptab[(int64_t)(src + r) * ROW_D + c] : 0.0;
  }
}

// sum of the per-interval forward log-likelihoods (interval order) into stats[0], count into stats[1]
__global__ void k_stats_logprob(int n, const int *ids, const double *fwd_lp, double *stats) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += fwd_lp[ids[i]];
    stats[0] += s;
    stats[1] += (double)n;
  }
}

}  // namespace tehmm
