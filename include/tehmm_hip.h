/*
 * tehmm_hip.h -- C ABI of libtehmm_hip.so, the MI355X (gfx950) implementation of teHmm's hot path.
 *
 * The library is a drop-in for the seven Cython functions the reference's Python model API calls
 * (hmm.py:57 `from . import _hmm`, emission.py:19 `from ._emission import ...`) plus fused,
 * device-resident entry points for the three drivers built on them (BaseHMM.decode,
 * BaseHMM.score_samples and the Baum-Welch E-step of BaseHMM.fit).
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types; every function returns 0 on success or a
 *     negative TEHMM_ERR_* code (tehmm_last_error() gives the message for the calling thread);
 *   - all floating point is IEEE fp64, arrays are C-contiguous, little-endian;
 *   - "host" pointers are ordinary process memory owned by the caller (like the NumPy buffers the
 *     Cython functions receive); the callee never keeps them after returning;
 *   - segRatios may be NULL ("None" in the reference);
 *   - handles are opaque, tied to the HIP device that was current when they were created, and are
 *     safe to use from one thread at a time.
 *
 * Citations are file:line in glennhickey/teHmm.
 */
#ifndef TEHMM_HIP_H
#define TEHMM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEHMM_OK 0
#define TEHMM_ERR_ARG (-1)         /* bad argument (NULL, negative size, N out of range ...) */
#define TEHMM_ERR_HIP (-2)         /* HIP runtime error (no device, out of memory, launch failure) */
#define TEHMM_ERR_UNSUPPORTED (-3) /* shape outside what the kernels support (see tehmm_max_states) */

/* ---- library / device ---------------------------------------------------------------------- */
int tehmm_abi_version(void);                 /* bumps when a signature changes or is added (3) */
const char *tehmm_last_error(void);          /* thread-local message of the last failing call */
int tehmm_device_count(int *count);          /* hipGetDeviceCount */
int tehmm_set_device(int device);            /* hipSetDevice; one process per GPU calls this once */
int tehmm_max_states(void);                  /* largest N the fused kernels accept (128) */

/* ---- array-level entry points: 1:1 replacements of the Cython module functions --------------
 * Caller owns every buffer (host memory); results are written in place exactly where the Cython
 * functions write them. */

/* _emission.fastAllLogProbs -> _fastAllLogProbsU8/U16/32 (_emission.pyx:20-144).
 * obs [T][K], logProbs [K][N][S], outProbs [T][N]; includes the leading-rows quirk (pyx:73-80). */
int tehmm_emission_u8(int64_t T, int K, int N, int S, const uint8_t *obs, const double *logProbs,
                      double normalize, const double *segRatios, double *outProbs);
int tehmm_emission_u16(int64_t T, int K, int N, int S, const uint16_t *obs, const double *logProbs,
                       double normalize, const double *segRatios, double *outProbs);
int tehmm_emission_i32(int64_t T, int K, int N, int S, const int32_t *obs, const double *logProbs,
                       double normalize, const double *segRatios, double *outProbs);

/* _hmm._forward (_hmm.pyx:120-158): fwdlattice [T][N] out. */
int tehmm_forward(int64_t T, int N, const double *log_startprob, const double *log_transmat,
                  const double *framelogprob, const double *segRatios, double *fwdlattice);

/* _hmm._backward (_hmm.pyx:160-198): bwdlattice [T][N] out. */
int tehmm_backward(int64_t T, int N, const double *log_startprob, const double *log_transmat,
                   const double *framelogprob, const double *segRatios, double *bwdlattice);

/* _hmm._viterbi (_hmm.pyx:201-259): state_sequence int64 [T] out, *logprob out. */
int tehmm_viterbi(int64_t T, int N, const double *log_startprob, const double *log_transmat,
                  const double *segRatios, const double *framelogprob, int64_t *state_sequence,
                  double *logprob);

/* _hmm._log_sum_lneta (_hmm.pyx:62-117): logsum_lneta [N][N] in/out (caller zero-fills it,
 * hmm.py:557). */
int tehmm_xi_logsum(int64_t T, int N, const double *fwdlattice, const double *log_transmat,
                    const double *bwdlattice, const double *framelogprob, double logprob,
                    const double *segRatios, double *logsum_lneta);

/* _emission.fastAccumulateStats -> _fastAccumulateStatsU8/U16/32 (_emission.pyx:146-234):
 * obsStats [K][N][S] += ... in place, in the reference's accumulation order. */
int tehmm_accumulate_obs_u8(int64_t T, int K, int N, int S, const uint8_t *obs, double *obsStats,
                            const double *posteriors, const double *segRatios);
int tehmm_accumulate_obs_u16(int64_t T, int K, int N, int S, const uint16_t *obs, double *obsStats,
                             const double *posteriors, const double *segRatios);
int tehmm_accumulate_obs_i32(int64_t T, int K, int N, int S, const int32_t *obs, double *obsStats,
                             const double *posteriors, const double *segRatios);

/* _emission.fastUpdateCounts -> _fastUpdateCountsU8/U16/32 (_emission.pyx:236-332), batched over the
 * labelled intervals of ONE table (the reference calls it once per overlap, emission.py:307-322):
 *   for i in 0..n_intervals-1, pos in [starts[i], ends[i]):      (table-relative coordinates)
 *     obsStats[track][states[i]][obs[pos][track]] += segRatios ? segRatios[pos] : 1.0
 * in that order (so ratio sums round as in the reference).  obs [T][K], obsStats [K][N][S] in place. */
int tehmm_update_counts_u8(int64_t T, int K, int N, int S, const uint8_t *obs, int n_intervals,
                           const int64_t *starts, const int64_t *ends, const int32_t *states,
                           const double *segRatios, double *obsStats);
int tehmm_update_counts_u16(int64_t T, int K, int N, int S, const uint16_t *obs, int n_intervals,
                            const int64_t *starts, const int64_t *ends, const int32_t *states,
                            const double *segRatios, double *obsStats);
int tehmm_update_counts_i32(int64_t T, int K, int N, int S, const int32_t *obs, int n_intervals,
                            const int64_t *starts, const int64_t *ends, const int32_t *states,
                            const double *segRatios, double *obsStats);

/* ---- fused, device-resident entry points ------------------------------------------------------
 * A model handle keeps the N x N log-transition matrix, start vector and emission tables on the
 * device; a batch handle keeps the observation columns (and segment ratios) of many independent
 * intervals (TrackTables) on the device, plus the result buffers.  One call then runs a whole
 * driver of the reference over every interval of the batch. */
typedef struct tehmm_model tehmm_model_t;
typedef struct tehmm_batch tehmm_batch_t;

/* Model state the reference keeps in MultitrackHmm._log_transmat / _log_startprob (hmm.py:625-666)
 * and emissionModel.logProbs [K][N][S] + normalizeFac (emission.py:44,58-60).
 * symbolsPerTrack (may be NULL) = emissionModel.numSymbolsPerTrack: lets the library pack the
 * table raggedly (symbols 0..symbolsPerTrack[k] of track k); with NULL all S columns are kept. */
int tehmm_model_create(int N, int K, int S, const double *log_transmat, const double *log_startprob,
                       const double *logProbs, double normalize, const int32_t *symbolsPerTrack,
                       tehmm_model_t **out);
int tehmm_model_destroy(tehmm_model_t *model);

/* Observations of n_intervals TrackTables, concatenated: interval i is rows
 * offsets[i] .. offsets[i+1]-1 of obs [total][K] (uint8, IntegerTrackTable.data, track.py:555) and
 * of segRatios [total] (may be NULL).  obs_on_device != 0: obs/segRatios are device pointers on
 * the current device (the data stays where it is and is repacked by a kernel on the batch's own stream, after
 * the work queued on the default stream; a producer on any other stream must have finished). */
int tehmm_batch_create(int n_intervals, const int64_t *offsets, int K, const uint8_t *obs,
                       const double *segRatios, int obs_on_device, tehmm_batch_t **out);
/* The same from the other two observation types of the reference (IntegerTrackTable with uint16 / int32 data:
 * _fastAllLogProbsU16 / 32, _fastAccumulateStatsU16 / 32, _emission.pyx:82-144, 192-234), host arrays only.  The fused
 * kernels keep one byte per track and position: a symbol outside 0..255 returns TEHMM_ERR_UNSUPPORTED (such tables go
 * through the array-level entry points, which take all three types). */
int tehmm_batch_create_u16(int n_intervals, const int64_t *offsets, int K, const uint16_t *obs,
                           const double *segRatios, tehmm_batch_t **out);
int tehmm_batch_create_i32(int n_intervals, const int64_t *offsets, int K, const int32_t *obs,
                           const double *segRatios, tehmm_batch_t **out);
int tehmm_batch_destroy(tehmm_batch_t *batch);
int64_t tehmm_batch_total(const tehmm_batch_t *batch);
/* Forgets what earlier evaluations derived from the batch's observations for a model (the table-row index
 * records of the fused posterior passes): the next call pays for them again, as the first evaluation of a
 * fresh batch does.  bench.py calls it before every timed step -- teHmmEval evaluates a batch once. */
int tehmm_batch_reset_cache(tehmm_batch_t *batch);

/* Flags for tehmm_eval_batch. */
#define TEHMM_EVAL_VITERBI 1   /* BaseHMM.decode -> _decode_viterbi (basehmm.py:301-330, 361-396) */
#define TEHMM_EVAL_POSTERIOR 2 /* BaseHMM.score_samples (basehmm.py:238-273) */
#define TEHMM_EVAL_USE_RATIOS 4 /* the batch's segRatios apply the way the reference applies them:
                                   decode: transitions only (Q11); score_samples: never (Q12) */

/* Runs decode and/or score_samples over every interval.  Results stay on the device (fetch with
 * the tehmm_batch_get_* calls); viterbi_logprob / forward_logprob [n_intervals] are host arrays
 * (may be NULL). */
int tehmm_eval_batch(tehmm_model_t *model, tehmm_batch_t *batch, int flags,
                     double *viterbi_logprob, double *forward_logprob);

/* Copy results of the last tehmm_eval_batch to host memory: paths int64 [total] (the concatenated
 * state sequences), posteriors [total][N]; row range [row0,row1) of the concatenation. */
int tehmm_batch_get_paths(tehmm_batch_t *batch, int64_t row0, int64_t row1, int64_t *paths);
int tehmm_batch_get_posteriors(tehmm_batch_t *batch, int64_t row0, int64_t row1, double *post);
/* Pinned host memory for the destinations above (one DMA at link speed instead of the runtime's bounce buffers;
 * any other destination is staged in 32 MB pieces through two pinned buffers).  Freed blocks are cached in the
 * library -- pinning is the slow part -- up to 24 GB. */
int tehmm_host_alloc(size_t bytes, void **out);
int tehmm_host_free(void *ptr);
/* The library also keeps the DEVICE blocks of destroyed batches for the next batch (a fresh batch per call allocates the
 * same workspaces again; hipMalloc of memory the process has just returned can cost hundreds of milliseconds): up to
 * TEHMM_DEVICE_POOL_GB (environment, default 48, 0 = no caching).  tehmm_trim_pools returns every cached device and
 * pinned host block to the system (call it before handing the GPU to another allocator that needs the room). */
int tehmm_trim_pools(void);
/* Device pointers of the same buffers (valid until the batch is destroyed or re-evaluated). */
int tehmm_batch_device_ptrs(tehmm_batch_t *batch, void **paths_i64, void **posteriors_f64);

/* ---- output reductions: what teHmmEval writes per row (bin/teHmmEval.py:238-275) ----------------
 * Posterior column of --pd / --pdStates (:270-272): out[r - row0] = sum_j posteriors[r][j] * mask[j]
 * for rows [row0, row1) of the last evaluation's device-resident posteriors (8 instead of 8 N bytes
 * per row cross PCIe).  mask [N] and out are host arrays.  (The reference's off-by-one row, quirk
 * Q15, is the caller's: tehmm_amd/output.py.) */
int tehmm_batch_posterior_masksum(tehmm_batch_t *batch, const double *mask, int64_t row0, int64_t row1,
                                  double *out);
/* BED coordinates of every row of a table (:243-266): segOffsets [n_rows] (NULL: unsegmented),
 * maskOffsets [n_mask] = TrackTable.getMaskRunningOffsets() (NULL: no mask); starts / ends [n_rows]. */
int tehmm_bed_coords(int64_t n_rows, int64_t table_start, int64_t table_end, const int64_t *segOffsets,
                     const int32_t *maskOffsets, int64_t n_mask, int64_t *starts, int64_t *ends);
/* Host-side writer of the per-row lines "chrom\tstart\tend\tX\n" (:266-275).  X = names[states[i]]
 * (names NULL: the integer) or, with values != NULL, values[i] as Python 2 prints a float64. */
int tehmm_write_bed(const char *path, int append, const char *chrom, int64_t n, const int64_t *starts,
                    const int64_t *ends, const int64_t *states, int n_names, const char *const *names,
                    const double *values);

/* Baum-Welch E-step over every interval of the batch (basehmm.py:504-523 with
 * MultitrackHmm._accumulate_sufficient_statistics, hmm.py:545-574): accumulates INTO the host
 * arrays start[N], trans[N][N], obsStats[K][N][S] (the caller initialises them, e.g. with
 * emission.initStats' fudge) and returns the summed forward log-likelihood.  use_ratios: apply
 * the batch's segRatios everywhere, as fit does for segmented TrackTables (emission rows scaled,
 * emission.py:195-196; lt[j][j] (r - 1) in the recurrences, _hmm.pyx:131-140; ratio-weighted posteriors in
 * the histograms, _emission.pyx:183-190; the diagonal term of _log_sum_lneta, _hmm.pyx:94-99).
 * N <= 128.  Below 64 states without ratios: the chunk-parallel fused passes with exact chains; with ratios,
 * and at 64..128 states: the item-parallel passes (warm-up + verified links); what those do not take (rows no
 * state can emit, links that never verify) runs on sequential kernels below 64 states and returns
 * TEHMM_ERR_UNSUPPORTED at 64 and above (the array-level entry points then serve the reference's own loop). */
int tehmm_estep_batch(tehmm_model_t *model, tehmm_batch_t *batch, int use_ratios, double *start,
                      double *trans, double *obsStats, double *logprob_sum);

/* ---- device-resident Baum-Welch ----------------------------------------------
 * The E-step's raw sufficient statistics stay on the device in ONE flat fp64 buffer
 * (tehmm_model_stats_size doubles: [logprob sum, sequence count, start, transition accumulators,
 * emission histograms]); buffers of several batches / ranks simply add (one all-reduce per EM
 * iteration over this buffer, basehmm.py:507-522 summed across shards), and tehmm_model_mstep turns the
 * sum into the next parameters without leaving the device (MultitrackHmm._do_mstep, hmm.py:576-616;
 * emission.maximize, emission.py:243-267; gaussian refit, emission.py:502-593, quirk Q19). */
int64_t tehmm_model_stats_size(const tehmm_model_t *model);
int tehmm_stats_alloc(const tehmm_model_t *model, double **dev_stats);   /* zero-filled device buffer */
int tehmm_stats_zero(const tehmm_model_t *model, double *dev_stats);
int tehmm_stats_free(double *dev_stats);
int tehmm_stats_head(const double *dev_stats, double *logprob_sum, double *n_sequences);
/* Copies the whole buffer device -> host (to_device = 0) or host -> device (to_device = 1): how a process group
 * whose backend cannot see device memory (gloo) sums the statistics of its ranks. */
int tehmm_stats_copy(const tehmm_model_t *model, double *dev_stats, double *host_buf, int to_device);
/* tehmm_estep_batch with the statistics ADDED into dev_stats (a device pointer: from tehmm_stats_alloc
 * or e.g. a torch tensor that RCCL will all-reduce). */
int tehmm_estep_batch_device(tehmm_model_t *model, tehmm_batch_t *batch, int use_ratios, double *dev_stats,
                             double *logprob_sum);
/* M-step: updates the model handle in place from the (summed) statistics.  update_*: the 's', 't', 'e'
 * of the reference's `params`; priors default to 1.0 there; fudge = emission model's fudge
 * (initStats + maximize); gaussian tracks: indices, the real value of every symbol
 * (gauss_values [n_gauss][S], CategoryMap.getMapBack) and the uniform mix (0.1); gauss_params
 * [n_gauss][N][2] (mu, sigma) out, may be NULL. */
int tehmm_model_mstep(tehmm_model_t *model, const double *dev_stats, int update_start, int update_trans,
                      int update_emission, double startprob_prior, double transmat_prior, double fudge,
                      int n_gauss, const int32_t *gauss_tracks, const double *gauss_values, double uniform_mix,
                      double *gauss_params);
/* Current parameters of the handle: log_transmat [N][N], log_startprob [N], logProbs [K][N][S] (only
 * the cells of real symbols are written); any pointer may be NULL. */
int tehmm_model_get_params(tehmm_model_t *model, double *log_transmat, double *log_startprob, double *logProbs);

/* ---- the step before the path: segment compression and mask compaction --------
 * TrackTable.segment = interpolateSegments + compressSegments (track.py:449-533, 594-620) for a uint8
 * table data [T][K] and ascending table-relative segOffsets [n_seg]: out [n_seg][K] holds the mode of
 * every categorical track over the segment; for gaussian tracks (is_gaussian[k] != 0) means [n_seg][K]
 * holds the mean of mapback[k][symbol] (mapback [K][256], CategoryMap.getMapBackTable) and out keeps
 * the segment's first symbol -- the caller maps the mean to a symbol (CategoryMap.getMap(update=True)
 * may create one, track.py:612-616). */
int tehmm_segment_table_u8(int64_t T, int K, const uint8_t *data, int64_t n_seg, const int64_t *segOffsets,
                           const uint8_t *is_gaussian, const double *mapback, uint8_t *out, double *means);
/* IntegerTrackTable.setMaskTable + getMaskRunningOffsets (track.py:622-662; _track.runSum,
 * _track.pyx:13-25): keep [T] = 1 where no mask track covers the position, run_full [T] = number of cut
 * positions before i, out_data [n_keep][K] / run_masked [n_keep] = the kept rows and their offsets. */
int tehmm_mask_table_u8(int64_t T, int K, const uint8_t *data, int KM, const uint8_t *maskdata, uint8_t *keep,
                        int32_t *run_full, uint8_t *out_data, int32_t *run_masked, int64_t *n_keep);

/* Forward log-likelihood of every interval from the last tehmm_estep_batch or posterior evaluation
 * (the per-sequence `lpr` of basehmm.py:513, which MultitrackHmm's best-iteration bookkeeping,
 * hmm.py:690-711, consumes sequence by sequence); out [n_intervals] host. */
int tehmm_batch_get_interval_logprobs(tehmm_batch_t *batch, double *out);

/* Per-kernel device time of the last fused call on this batch, measured with HIP events on the
 * streams the kernels ran on.  names[i] is a static string; returns the number of entries
 * written (<= max_entries). */
int tehmm_batch_last_timing(tehmm_batch_t *batch, int max_entries, const char **names,
                            double *milliseconds);

/* Diagnostic builds only (-DTEHMM_STAMPS): per-wave cycle stamps of the last cooperative kernel,
 * [workgroup][wave][4] counters.  The product library returns TEHMM_ERR_UNSUPPORTED. */
int tehmm_debug_read_stamps(unsigned long long *out, int n);

#ifdef __cplusplus
}
#endif
#endif /* TEHMM_HIP_H */
