"""Seeded synthetic models / observations shaped like teHmm workloads.

This is the generator SURVEY.md section 8(d) describes: sticky random transition matrix,
uniform start, per-track multinomial rows (symbol 0 = "missing", log-prob 0 as in
emission.py:155-159) and 250-bin gaussian tracks baked into the same table
(emission.py:552-584).  Used by the tests, by tests/golden/make_golden.py and by bench.py.
Nothing here touches the GPU or the CPU checker.
"""
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from .common import myLog

# config 2/3 of BASELINE.json: 8 multinomial + 2 gaussian tracks
CONFIG2_SYMBOLS = (2, 2, 3, 4, 5, 8, 12, 30, 250, 250)
CONFIG2_GAUSSIAN = (8, 9)
# config 3b: the shape of data/mustang_alyrata_tracks_clean.xml -- 15 multinomial + 14 gaussian
# (250 bins) + 3 binary tracks
CONFIG3B_SYMBOLS = (2, 3, 3, 4, 4, 5, 6, 8, 8, 10, 12, 16, 20, 24, 30) + (250,) * 14 + (2, 2, 2)
CONFIG3B_GAUSSIAN = tuple(range(15, 29))
# config 4: 10 multinomial + 2 gaussian tracks
CONFIG4_SYMBOLS = (2, 2, 3, 3, 4, 5, 8, 12, 20, 30, 250, 250)
CONFIG4_GAUSSIAN = (10, 11)


@dataclass
class SynthModel:
    n_states: int
    symbols_per_track: List[int]
    log_startprob: np.ndarray      # f64 [N]
    log_transmat: np.ndarray       # f64 [N, N]
    log_probs: np.ndarray          # f64 [K, N, S], S = 1 + max(symbols_per_track)
    transmat: np.ndarray           # f64 [N, N] (linear)
    probs: np.ndarray              # f64 [K, N, S] (linear, symbol 0 -> 1.0)

    @property
    def n_tracks(self):
        return len(self.symbols_per_track)


def make_model(n_states: int, symbols_per_track: Sequence[int] = CONFIG2_SYMBOLS,
               gaussian_tracks: Sequence[int] = CONFIG2_GAUSSIAN, seed: int = 0,
               sparse: float = 0.0, log_zero_emission: Optional[float] = None,
               stay: Optional[float] = None) -> SynthModel:
    """stay: self-transition probability of every state (trained TE models sit at 0.99+); default: the
    generator's A[i, i] += N, i.e. about 0.67 at N = 35."""
    rs = np.random.RandomState(seed)
    N = n_states
    K = len(symbols_per_track)
    A = rs.rand(N, N)
    A[np.arange(N), np.arange(N)] += N
    if sparse > 0.0:
        mask = rs.rand(N, N) < sparse
        mask[np.arange(N), np.arange(N)] = False
        A[mask] = 0.0
    if stay is not None and N > 1:
        off = A.copy()
        off[np.arange(N), np.arange(N)] = 0.0
        rowsum = off.sum(axis=1, keepdims=True)
        off = np.where(rowsum > 0, off / np.where(rowsum > 0, rowsum, 1.0) * (1.0 - stay), 0.0)
        A = off
        A[np.arange(N), np.arange(N)] = np.where(rowsum[:, 0] > 0, stay, 1.0)
    A /= A.sum(axis=1, keepdims=True)
    lt = np.asarray(myLog(A), dtype=np.float64)
    pi = np.full(N, 1.0 / N)
    lpi = np.asarray(myLog(pi), dtype=np.float64)
    S = 1 + max(symbols_per_track)
    P = np.zeros((K, N, S))
    P[:, :, 0] = 1.0
    for k, sk in enumerate(symbols_per_track):
        for j in range(N):
            if k in gaussian_tracks:
                mu = rs.uniform(0, sk)
                sigma = rs.uniform(5, 40)
                x = np.arange(sk, dtype=np.float64)
                pdf = np.exp(-0.5 * ((x - mu) / sigma) ** 2) / (sigma * np.sqrt(2 * np.pi))
                row = 0.1 / sk + 0.9 * pdf
            else:
                row = 0.2 + 0.6 * rs.rand(sk)
            row = row / row.sum()
            P[k, j, 1:1 + sk] = row
    with np.errstate(divide="ignore"):
        logP = np.log(np.where(P > 0, P, 1.0))
    # unused (padding) symbols keep log-prob 0.0 like np.zeros-initialised logProbs
    if log_zero_emission is not None:
        logP = np.where((P == 0) & (np.arange(S)[None, None, :] > 0)
                        & (np.arange(S)[None, None, :] <= np.asarray(symbols_per_track)[:, None, None]),
                        log_zero_emission, logP)
    return SynthModel(N, list(symbols_per_track), lpi, lt, np.ascontiguousarray(logP), A,
                      np.ascontiguousarray(P))


def sample_obs(model: SynthModel, T: int, seed: int = 0, missing: float = 0.0) -> np.ndarray:
    """Sample uint8 [T, K] observations from the HMM itself (exact sequential sampler;
    use for test-sized T).  ``missing`` = fraction of entries replaced by symbol 0."""
    rs = np.random.RandomState(seed)
    N, K = model.n_states, model.n_tracks
    tcdf = np.cumsum(model.transmat, axis=1)
    states = np.empty(T, dtype=np.int64)
    s = rs.randint(N)
    u = rs.rand(T)
    for t in range(T):
        if t > 0:
            s = min(int(np.searchsorted(tcdf[s], u[t], side="right")), N - 1)
        states[t] = s
    obs = np.empty((T, K), dtype=np.uint8)
    for k, sk in enumerate(model.symbols_per_track):
        cdf = np.cumsum(model.probs[k, :, 1:1 + sk], axis=1)          # [N, sk]
        uu = rs.rand(T)
        flat = (cdf + np.arange(N)[:, None]).ravel()
        idx = np.searchsorted(flat, uu * 0.999999999 + states, side="right") - states * sk
        obs[:, k] = (np.clip(idx, 0, sk - 1) + 1).astype(np.uint8)
    if missing > 0.0:
        obs[rs.rand(T, K) < missing] = 0
    return obs


def random_obs(model: SynthModel, T: int, seed: int = 0, missing: float = 0.02) -> np.ndarray:
    """Uniform random symbols in 1..S_k (plus a sprinkle of 0 = missing)."""
    rs = np.random.RandomState(seed)
    obs = np.empty((T, model.n_tracks), dtype=np.uint8)
    for k, sk in enumerate(model.symbols_per_track):
        obs[:, k] = rs.randint(1, sk + 1, size=T)
    if missing > 0.0:
        obs[rs.rand(T, model.n_tracks) < missing] = 0
    return obs


def random_ratios(T: int, seed: int = 0, eff_len: float = 20.0, max_len: int = 100) -> np.ndarray:
    """Segment-length ratios: lengths ~ 1 + Geometric(1/20) capped at ``max_len``,
    ratio = length / effective length (track.py:504-513)."""
    rs = np.random.RandomState(seed)
    lens = np.minimum(1 + rs.geometric(1.0 / 20.0, size=T), max_len).astype(np.float64)
    return lens / float(eff_len)


def interval_lengths(total: int, lo: int, hi: int, seed: int = 0) -> np.ndarray:
    """Cut ``total`` positions into intervals with lengths ~ U(lo, hi) (config-3 geometry:
    scaffolds cut at synthetic mask gaps)."""
    rs = np.random.RandomState(seed)
    out = []
    left = total
    while left > 0:
        L = int(rs.randint(lo, hi + 1))
        L = min(L, left)
        if left - L < lo // 2:
            L = left
        out.append(L)
        left -= L
    return np.asarray(out, dtype=np.int64)
