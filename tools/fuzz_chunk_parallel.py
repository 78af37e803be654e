"""Randomised parity fuzz of the chunk-parallel evaluation against the CPU oracle (test infrastructure:
run it from the repo root on a GPU box: python tools/fuzz_chunk_parallel.py [n_cases] [seed0] [long];
tests/test_gpu_r3.py runs a fixed-seed slice of it inside `pytest -m gpu`)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KEYS = ("TEHMM_SPEC_CHUNK", "TEHMM_LANE_SUB", "TEHMM_LANE_WARMUP", "TEHMM_LANE_WARMUP_VIT", "TEHMM_LANE_VIT",
        "TEHMM_LANE_P0", "TEHMM_LANE_MFMA", "TEHMM_FB_RUNS", "TEHMM_VIT_RUNS", "TEHMM_FUSED")


def run_case(case, seed0=0, long_mode=False, bign=False, verbose=True):
    """One seeded case; returns None when the library agrees with the oracle, else a description."""
    from numpy.testing import assert_array_equal, assert_allclose
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    rs = np.random.RandomState(seed0 + case)
    saved = {k: os.environ.get(k) for k in KEYS}
    for k in KEYS:
        os.environ.pop(k, None)
    cs = int(rs.choice([512, 1024, 2048, 4096] if long_mode else [128, 256, 512, 1024]))
    env = {"TEHMM_SPEC_CHUNK": str(cs)}
    sub = int(rs.choice([0, 64, 128, 256, 512]))
    if sub and sub <= cs:
        env["TEHMM_LANE_SUB"] = str(sub)
    elif sub == 0:
        env["TEHMM_LANE_SUB"] = "0"
    if rs.rand() < 0.3:
        env["TEHMM_LANE_VIT"] = "0"
    if rs.rand() < 0.2:
        env["TEHMM_LANE_MFMA"] = "1"
    if rs.rand() < 0.2:
        env["TEHMM_LANE_WARMUP"] = str(int(rs.choice([8, 24, 48])))
    if rs.rand() < 0.2:
        env["TEHMM_VIT_RUNS"] = "0"
    if rs.rand() < 0.2:
        env["TEHMM_FB_RUNS"] = "0"
    if rs.rand() < 0.15:
        env["TEHMM_FUSED"] = "0"
    os.environ.update(env)
    N = int(rs.choice([2, 3, 5, 8, 13, 20, 27, 35, 36, 41, 50, 63]))
    big_n = bign and rs.rand() < 0.7      # 64 <= N <= 128
    if big_n:
        N = int(rs.choice([64, 65, 77, 100, 127, 128]))
    K = int(rs.randint(1, 13))
    syms = [int(rs.choice([1, 2, 3, 5, 17, 100, 255])) for _ in range(K)]
    gauss = [k for k in range(K) if syms[k] >= 100 and rs.rand() < 0.5]
    model = synth.make_model(N, syms, gauss, seed=seed0 + case, sparse=float(rs.choice([0.0, 0.0, 0.3, 0.7])))
    normalize = float(rs.choice([1.0, 1.0, 3.0 / K]))
    lens = [int(x) for x in rs.choice([1, 7, 64, 300, 1500, 4097, 9000, 20000, 33000], size=int(rs.randint(1, 6)))]
    if rs.rand() < 0.5:
        lens.append(int(rs.randint(30000, 70000)))
    if big_n:
        lens = [int(x) for x in rs.choice([1, 7, 64, 300, 1500, 4097, 9000], size=int(rs.randint(1, 5)))]
    if long_mode:
        lens = [int(rs.randint(100000, 500000)) for _ in range(int(rs.randint(1, 4)))]
        N = int(rs.choice([5, 8, 20, 35, 36]))
        model = synth.make_model(N, syms, gauss, seed=seed0 + case, sparse=float(rs.choice([0.0, 0.0, 0.3])))
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=case, missing=float(rs.choice([0.0, 0.05, 0.3])))
    with_ratio = bool(rs.rand() < 0.4)
    ratios = synth.random_ratios(int(offs[-1]), seed=case) if with_ratio else None
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, normalize, model.symbols_per_track)
    hb = HipBatch(obs, offs, ratios)
    try:
        res = hm.eval(hb, viterbi=True, posterior=True)
        p_o, vlp_o, flp_o, post_o = oracle.eval_batch(obs, offs, model.log_probs, model.log_startprob,
                                                      model.log_transmat, normalize, ratios, n_threads=8)
        status = None
        try:
            assert_array_equal(hb.paths(), p_o)
            assert_array_equal(res["viterbi_logprob"], vlp_o)
            # (one-symbol models have log P = 0 exactly: compare those absolutely, 5e-13 per position -- every
            #  verified jump carries ~1e-11 of log-scale rounding)
            assert_allclose(res["forward_logprob"], flp_o, rtol=1e-6, atol=1e-9 + 5e-13 * float(offs[-1]))
            assert_allclose(hb.posteriors(), post_o, rtol=1e-6, atol=1e-15)
        except AssertionError as e:
            status = "MISMATCH " + " | ".join(x.strip() for x in str(e).splitlines()[:8])[:400]
        t = hb.timing()
        if verbose:
            print(case, status or "ok", "N", N, "K", K, "T", int(offs[-1]), env, "ratio" if with_ratio else "",
                  {k.split(":")[1]: int(v) for k, v in t.items() if k.startswith("count:")}, flush=True)
        if status:
            status = "case %d seed0 %d N %d K %d T %d %s%s: %s" % (case, seed0, N, K, int(offs[-1]), env,
                                                                   " ratio" if with_ratio else "", status)
        return status
    finally:
        hb.close()
        hm.close()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    long_mode = len(sys.argv) > 3 and sys.argv[3] == "long"     # few long intervals: several binades per interval
    bad = 0
    t_start = time.time()
    for case in range(int(os.environ.get("FUZZ_START", "0")), n_cases):
        bad += run_case(case, seed0, long_mode, bool(os.environ.get("FUZZ_BIGN"))) is not None
    print("cases", n_cases, "mismatches", bad, "%.0f s" % (time.time() - t_start))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
