/*
 * tehmm_oracle.c -- CPU restatement of teHmm's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the MI355X build: a plain-C, single-threaded
 * re-expression of the reference's Cython loops with the SAME fp64 operation order,
 * so that integer results (Viterbi paths) and emission frames are bit-identical and
 * floating-point lattices agree to the last ulp when linked against the same libm.
 * It is pinned against golden vectors produced by the real reference (see
 * tests/golden/make_golden.py and tests/test_oracle_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (tehmm_amd/) never links, imports or calls anything in oracle/.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -pthread (see oracle/Makefile).
 *
 * Reference citations are relative to /root/reference.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_ZEROLOGPROB (-1e200) /* _hmm.pyx:60 */
#define ORACLE_MINDBL (-1e20)       /* _emission.pyx:12 */
#define ORACLE_F32_EPS 1.1920928955078125e-07 /* np.finfo(np.float32).eps, basehmm.py:271 */

/* ---------------------------------------------------------------------------------
 * Emission: _emission.pyx:50-80 (_fastAllLogProbsU8).  maxProb is initialised once per
 * call and never reset (quirk Q9): only rows before the first emittable row are zeroed.
 * logProbs is [K][N][S] C-contiguous, obs is [T][K] uint8, out is [T][N].
 * ------------------------------------------------------------------------------- */
int oracle_emission_u8(int64_t T, int K, int N, int S, const uint8_t *obs,
                       const double *logProbs, double normalize, const double *ratios,
                       double *out) {
  double maxProb = ORACLE_MINDBL;
  for (int64_t i = 0; i < T; ++i) {
    for (int j = 0; j < N; ++j) {
      double x = 0.0;
      for (int k = 0; k < K; ++k)
        x += logProbs[((size_t)k * N + j) * S + obs[i * K + k]];
      x *= normalize;
      if (ratios) x *= ratios[i];
      out[i * N + j] = x;
      if (x > maxProb) maxProb = x;
    }
    if (maxProb == ORACLE_MINDBL)
      for (int j = 0; j < N; ++j) out[i * N + j] = 0.0;
  }
  return 0;
}

/* Same loop for uint16 / int32 observations (_emission.pyx:82-144). */
int oracle_emission_u16(int64_t T, int K, int N, int S, const uint16_t *obs,
                        const double *logProbs, double normalize, const double *ratios,
                        double *out) {
  double maxProb = ORACLE_MINDBL;
  for (int64_t i = 0; i < T; ++i) {
    for (int j = 0; j < N; ++j) {
      double x = 0.0;
      for (int k = 0; k < K; ++k)
        x += logProbs[((size_t)k * N + j) * S + obs[i * K + k]];
      x *= normalize;
      if (ratios) x *= ratios[i];
      out[i * N + j] = x;
      if (x > maxProb) maxProb = x;
    }
    if (maxProb == ORACLE_MINDBL)
      for (int j = 0; j < N; ++j) out[i * N + j] = 0.0;
  }
  return 0;
}

int oracle_emission_i32(int64_t T, int K, int N, int S, const int32_t *obs,
                        const double *logProbs, double normalize, const double *ratios,
                        double *out) {
  double maxProb = ORACLE_MINDBL;
  for (int64_t i = 0; i < T; ++i) {
    for (int j = 0; j < N; ++j) {
      double x = 0.0;
      for (int k = 0; k < K; ++k)
        x += logProbs[((size_t)k * N + j) * S + obs[i * K + k]];
      x *= normalize;
      if (ratios) x *= ratios[i];
      out[i * N + j] = x;
      if (x > maxProb) maxProb = x;
    }
    if (maxProb == ORACLE_MINDBL)
      for (int j = 0; j < N; ++j) out[i * N + j] = 0.0;
  }
  return 0;
}

/* ---------------------------------------------------------------------------------
 * Forward: _hmm.pyx:120-158.
 * ------------------------------------------------------------------------------- */
int oracle_forward(int64_t T, int N, const double *pi, const double *lt,
                   const double *frame, const double *ratios, double *fwd) {
  double *w = (double *)malloc(sizeof(double) * (size_t)N);
  if (!w) return -1;
  for (int i = 0; i < N; ++i) {
    fwd[i] = pi[i] + frame[i];
    if (ratios && ratios[0] > 1.) fwd[i] += lt[i * N + i] * (ratios[0] - 1.);
  }
  for (int64_t t = 1; t < T; ++t) {
    for (int j = 0; j < N; ++j) {
      double vmax = -INFINITY;
      for (int i = 0; i < N; ++i) {
        w[i] = fwd[(t - 1) * N + i] + lt[i * N + j];
        if (ratios && ratios[t] > 1.) w[i] += lt[j * N + j] * (ratios[t] - 1.);
        if (w[i] > vmax) vmax = w[i];
      }
      double ps = 0.0;
      for (int i = 0; i < N; ++i) ps += exp(w[i] - vmax);
      double v = log(ps) + vmax + frame[t * N + j];
      if (v <= ORACLE_ZEROLOGPROB) v = -INFINITY;
      fwd[t * N + j] = v;
    }
  }
  free(w);
  return 0;
}

/* ---------------------------------------------------------------------------------
 * Backward: _hmm.pyx:160-198.  Terminal row is log(1/N) (quirk Q3).
 * ------------------------------------------------------------------------------- */
int oracle_backward(int64_t T, int N, const double *pi, const double *lt,
                    const double *frame, const double *ratios, double *bwd) {
  (void)pi;
  double *w = (double *)malloc(sizeof(double) * (size_t)N);
  if (!w) return -1;
  for (int i = 0; i < N; ++i) bwd[(T - 1) * N + i] = log(1. / (double)N);
  for (int64_t t = T - 2; t >= 0; --t) {
    for (int i = 0; i < N; ++i) {
      double vmax = -INFINITY;
      for (int j = 0; j < N; ++j) {
        w[j] = lt[i * N + j] + frame[(t + 1) * N + j] + bwd[(t + 1) * N + j];
        if (ratios && ratios[t + 1] > 1.) w[j] += lt[j * N + j] * (ratios[t + 1] - 1.);
        if (w[j] > vmax) vmax = w[j];
      }
      double ps = 0.0;
      for (int j = 0; j < N; ++j) ps += exp(w[j] - vmax);
      double v = log(ps) + vmax;
      if (v <= ORACLE_ZEROLOGPROB) v = -INFINITY;
      bwd[t * N + i] = v;
    }
  }
  free(w);
  return 0;
}

/* ---------------------------------------------------------------------------------
 * Viterbi: _hmm.pyx:201-259.  Note the from-state-0 segment-ratio quirk (Q4), strict
 * '>' ascending scan (lowest index wins ties, Q5) and emission inside the max (Q6).
 * tb16 (optional, may be NULL) receives the int16 traceback table [T][N].
 * ------------------------------------------------------------------------------- */
int oracle_viterbi(int64_t T, int N, const double *pi, const double *lt,
                   const double *ratios, const double *frame, int64_t *path,
                   double *logprob) {
  double *V = (double *)malloc(sizeof(double) * (size_t)T * N);
  int16_t *tb = (int16_t *)malloc(sizeof(int16_t) * (size_t)T * N);
  if (!V || !tb) {
    free(V);
    free(tb);
    return -1;
  }
  for (int j = 0; j < N; ++j) V[j] = pi[j] + frame[j];
  if (ratios && ratios[0] > 1.)
    for (int j = 0; j < N; ++j) V[j] += lt[j * N + j] * (ratios[0] - 1.);
  for (int64_t t = 1; t < T; ++t) {
    for (int to = 0; to < N; ++to) {
      double maxprob = V[(t - 1) * N + 0] + lt[0 * N + to] + frame[t * N + to];
      if (ratios) {
        maxprob += lt[to * N + to] * ratios[t];
        if (to == 0) maxprob -= lt[0 * N + to];
      }
      int16_t maxState = 0;
      for (int from = 1; from < N; ++from) {
        double cur = V[(t - 1) * N + from] + lt[from * N + to] + frame[t * N + to];
        if (ratios && ratios[t] > 1.) cur += lt[to * N + to] * (ratios[t] - 1.);
        if (cur > maxprob) {
          maxprob = cur;
          maxState = (int16_t)from;
        }
      }
      V[t * N + to] = maxprob;
      tb[t * N + to] = maxState;
    }
  }
  /* np.argmax: first maximum; a NaN is returned as soon as it is met. */
  int last = 0;
  {
    const double *row = V + (T - 1) * N;
    double m = row[0];
    if (m == m) {
      for (int j = 1; j < N; ++j) {
        if (row[j] != row[j]) {
          last = j;
          break;
        }
        if (row[j] > m) {
          m = row[j];
          last = j;
        }
      }
    }
  }
  path[T - 1] = last;
  *logprob = V[(T - 1) * N + last];
  for (int64_t t = T - 1; t > 0; --t) path[t - 1] = tb[t * N + path[t]];
  free(V);
  free(tb);
  return 0;
}

/* ---------------------------------------------------------------------------------
 * logsumexp: basehmm.py:70-93 for a 1-D vector (vmax = arr.max(); sequential sum).
 * ------------------------------------------------------------------------------- */
double oracle_logsumexp(int n, const double *x) {
  double vmax = x[0];
  for (int i = 1; i < n && vmax == vmax; ++i) { /* np.max propagates NaN */
    if (x[i] != x[i] || x[i] > vmax) vmax = x[i];
  }
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += exp(x[i] - vmax);
  return log(s) + vmax;
}

/* ---------------------------------------------------------------------------------
 * Xi log-sum: _hmm.pyx:62-117 (two passes: max, then sum of exp).  `out` must be
 * zero-initialised by the caller exactly like the reference (hmm.py:557).
 * ------------------------------------------------------------------------------- */
int oracle_xi_logsum(int64_t T, int N, const double *fwd, const double *lt,
                     const double *bwd, const double *frame, double logprob,
                     const double *ratios, double *out) {
  double *mx = (double *)malloc(sizeof(double) * (size_t)N * N);
  if (!mx) return -1;
  for (int i = 0; i < N * N; ++i) mx[i] = -INFINITY;
  for (int64_t t = 0; t < T - 1; ++t)
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        double x = fwd[t * N + i] + lt[i * N + j] + frame[(t + 1) * N + j] +
                   bwd[(t + 1) * N + j] - logprob;
        if (ratios && ratios[t + 1] > 1.) {
          x += lt[j * N + j] * (ratios[t + 1] - 1.);
          if (i == j) {
            double y = fwd[(t + 1) * N + i] + bwd[(t + 1) * N + j] +
                       log(ratios[t + 1] - 1.) - logprob;
            if (y > mx[i * N + j]) mx[i * N + j] = y;
          }
        }
        if (x > mx[i * N + j]) mx[i * N + j] = x;
      }
  for (int64_t t = 0; t < T - 1; ++t)
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        double x = fwd[t * N + i] + lt[i * N + j] + frame[(t + 1) * N + j] +
                   bwd[(t + 1) * N + j] - logprob;
        if (ratios && ratios[t + 1] > 1.) {
          x += lt[j * N + j] * (ratios[t + 1] - 1.);
          if (i == j) {
            double y = fwd[(t + 1) * N + i] + bwd[(t + 1) * N + j] +
                       log(ratios[t + 1] - 1.) - logprob;
            out[i * N + j] += exp(y - mx[i * N + j]);
          }
        }
        out[i * N + j] += exp(x - mx[i * N + j]);
      }
  for (int i = 0; i < N * N; ++i) out[i] = log(out[i]) + mx[i];
  free(mx);
  return 0;
}

/* ---------------------------------------------------------------------------------
 * Emission statistics: _emission.pyx:165-190.  obsStats is [K][N][S].
 * ------------------------------------------------------------------------------- */
int oracle_accumulate_obs_u8(int64_t T, int K, int N, int S, const uint8_t *obs,
                             double *obsStats, const double *post, const double *ratios) {
  for (int64_t i = 0; i < T; ++i)
    for (int k = 0; k < K; ++k) {
      int v = obs[i * K + k];
      for (int j = 0; j < N; ++j) {
        double p = post[i * N + j];
        if (ratios) p *= ratios[i];
        obsStats[((size_t)k * N + j) * S + v] += p;
      }
    }
  return 0;
}

/* ---------------------------------------------------------------------------------
 * Posteriors from lattices.  add_eps=1: basehmm.py:265-272 (score_samples: exp(gamma -
 * logsumexp_row), += float32 eps, row renormalise).  add_eps=0: basehmm.py:516-517 (fit).
 * NumPy's row sum uses pairwise/unrolled summation; the plain sequential sum here differs
 * by at most a few ulp, covered by the 1e-12 tolerance the tests use for this function.
 * ------------------------------------------------------------------------------- */
int oracle_posteriors(int64_t T, int N, const double *fwd, const double *bwd, int add_eps,
                      double *post) {
  for (int64_t t = 0; t < T; ++t) {
    double vmax = -INFINITY;
    for (int j = 0; j < N; ++j) {
      double g = fwd[t * N + j] + bwd[t * N + j];
      post[t * N + j] = g;
      if (g > vmax) vmax = g;
    }
    double s = 0.0;
    for (int j = 0; j < N; ++j) s += exp(post[t * N + j] - vmax);
    double lse = log(s) + vmax;
    double tot = 0.0;
    for (int j = 0; j < N; ++j) {
      double p = exp(post[t * N + j] - lse);
      if (add_eps) p += ORACLE_F32_EPS;
      post[t * N + j] = p;
      tot += p;
    }
    if (add_eps)
      for (int j = 0; j < N; ++j) post[t * N + j] /= tot;
  }
  return 0;
}

/* ---------------------------------------------------------------------------------
 * Drivers that mirror the reference call stacks (used by bench.py's cpu_baseline leg
 * and by parity tests of the fused GPU entry points).
 * ------------------------------------------------------------------------------- */

/* BaseHMM.decode -> _decode_viterbi (basehmm.py:301-330): emission WITHOUT ratios
 * (np.asarray strips the table, Q11), Viterbi WITH ratios. */
int oracle_decode(int64_t T, int K, int N, int S, const uint8_t *obs, const double *logProbs,
                  double normalize, const double *pi, const double *lt, const double *ratios,
                  int64_t *path, double *logprob) {
  double *frame = (double *)malloc(sizeof(double) * (size_t)T * N);
  if (!frame) return -1;
  oracle_emission_u8(T, K, N, S, obs, logProbs, normalize, NULL, frame);
  int rc = oracle_viterbi(T, N, pi, lt, ratios, frame, path, logprob);
  free(frame);
  return rc;
}

/* BaseHMM.score_samples (basehmm.py:238-273): no ratios anywhere (Q12). */
int oracle_score_samples(int64_t T, int K, int N, int S, const uint8_t *obs,
                         const double *logProbs, double normalize, const double *pi,
                         const double *lt, double *logprob, double *post) {
  size_t n = (size_t)T * N;
  double *frame = (double *)malloc(sizeof(double) * n);
  double *fwd = (double *)malloc(sizeof(double) * n);
  double *bwd = (double *)malloc(sizeof(double) * n);
  if (!frame || !fwd || !bwd) {
    free(frame);
    free(fwd);
    free(bwd);
    return -1;
  }
  oracle_emission_u8(T, K, N, S, obs, logProbs, normalize, NULL, frame);
  oracle_forward(T, N, pi, lt, frame, NULL, fwd);
  *logprob = oracle_logsumexp(N, fwd + (size_t)(T - 1) * N);
  oracle_backward(T, N, pi, lt, frame, NULL, bwd);
  oracle_posteriors(T, N, fwd, bwd, 1, post);
  free(frame);
  free(fwd);
  free(bwd);
  return 0;
}

/* One sequence of the Baum-Welch E-step: basehmm.py:509-522 + hmm.py:545-574.
 * Ratios (if any) are applied everywhere.  Accumulates into start[N], trans[N][N],
 * obsStats[K][N][S] and *logprob_sum (curr_logprob += lpr). */
int oracle_estep_seq(int64_t T, int K, int N, int S, const uint8_t *obs, const double *logProbs,
                     double normalize, const double *pi, const double *lt, const double *ratios,
                     double *start, double *trans, double *obsStats, double *logprob_sum) {
  size_t n = (size_t)T * N;
  double *frame = (double *)malloc(sizeof(double) * n);
  double *fwd = (double *)malloc(sizeof(double) * n);
  double *bwd = (double *)malloc(sizeof(double) * n);
  double *post = (double *)malloc(sizeof(double) * n);
  double *xi = (double *)calloc((size_t)N * N, sizeof(double));
  if (!frame || !fwd || !bwd || !post || !xi) {
    free(frame); free(fwd); free(bwd); free(post); free(xi);
    return -1;
  }
  oracle_emission_u8(T, K, N, S, obs, logProbs, normalize, ratios, frame);
  oracle_forward(T, N, pi, lt, frame, ratios, fwd);
  double lpr = oracle_logsumexp(N, fwd + (size_t)(T - 1) * N);
  oracle_backward(T, N, pi, lt, frame, ratios, bwd);
  oracle_posteriors(T, N, fwd, bwd, 0, post);
  *logprob_sum += lpr;
  for (int j = 0; j < N; ++j) start[j] += post[j];
  if (T > 1) {
    oracle_xi_logsum(T, N, fwd, lt, bwd, frame, lpr, ratios, xi);
    for (int i = 0; i < N * N; ++i) trans[i] += exp(xi[i]);
  }
  oracle_accumulate_obs_u8(T, K, N, S, obs, obsStats, post, ratios);
  free(frame); free(fwd); free(bwd); free(post); free(xi);
  return 0;
}

/* ---------------------------------------------------------------------------------
 * Multi-threaded eval baseline: the reference's own scaling mechanism is independent
 * worker processes over intervals (bin/teHmmEval.py:312-383); here: threads pulling
 * intervals from a shared counter.  Per interval the reference teHmmEval flow runs
 * posteriorDistribution (score_samples) and viterbi (decode), each recomputing the
 * emission frame.
 * ------------------------------------------------------------------------------- */
typedef struct {
  int n_intervals;
  const int64_t *offsets; /* n_intervals+1 prefix offsets into obs rows */
  int K, N, S;
  const uint8_t *obs;
  const double *logProbs;
  double normalize;
  const double *pi, *lt;
  const double *ratios; /* may be NULL; concatenated like obs */
  int64_t *paths;       /* [total T] */
  double *vit_logprob;  /* [n_intervals] */
  double *fwd_logprob;  /* [n_intervals] */
  double *post;         /* [total T][N] or NULL to skip posteriors */
  int next;
  pthread_mutex_t mu;
  int status;
} eval_job_t;

static void *eval_worker(void *arg) {
  eval_job_t *job = (eval_job_t *)arg;
  for (;;) {
    pthread_mutex_lock(&job->mu);
    int idx = job->next++;
    pthread_mutex_unlock(&job->mu);
    if (idx >= job->n_intervals) break;
    int64_t t0 = job->offsets[idx], T = job->offsets[idx + 1] - t0;
    if (T <= 0) continue;
    const uint8_t *o = job->obs + (size_t)t0 * job->K;
    const double *r = job->ratios ? job->ratios + t0 : NULL;
    int rc = 0;
    if (job->post)
      rc |= oracle_score_samples(T, job->K, job->N, job->S, o, job->logProbs, job->normalize,
                                 job->pi, job->lt, &job->fwd_logprob[idx],
                                 job->post + (size_t)t0 * job->N);
    rc |= oracle_decode(T, job->K, job->N, job->S, o, job->logProbs, job->normalize, job->pi,
                        job->lt, r, job->paths + t0, &job->vit_logprob[idx]);
    if (rc) job->status = rc;
  }
  return NULL;
}

int oracle_eval_batch(int n_intervals, const int64_t *offsets, int K, int N, int S,
                      const uint8_t *obs, const double *logProbs, double normalize,
                      const double *pi, const double *lt, const double *ratios,
                      int64_t *paths, double *vit_logprob, double *fwd_logprob, double *post,
                      int n_threads) {
  eval_job_t job;
  memset(&job, 0, sizeof(job));
  job.n_intervals = n_intervals;
  job.offsets = offsets;
  job.K = K; job.N = N; job.S = S;
  job.obs = obs; job.logProbs = logProbs; job.normalize = normalize;
  job.pi = pi; job.lt = lt; job.ratios = ratios;
  job.paths = paths; job.vit_logprob = vit_logprob; job.fwd_logprob = fwd_logprob;
  job.post = post;
  pthread_mutex_init(&job.mu, NULL);
  if (n_threads < 1) n_threads = 1;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
  for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, eval_worker, &job);
  for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
  free(th);
  pthread_mutex_destroy(&job.mu);
  return job.status;
}
