"""Counters of the chunk-parallel paths on a small workload (how much of tests' workload jumps)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tehmm_amd import synth
from tehmm_amd.engine import HipBatch, HipModel
for cfg in ({"TEHMM_SPEC_CHUNK": "128", "TEHMM_LANE_SUB": "0"}, {"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "64"},
            {"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "64", "TEHMM_LANE_VIT": "1"}):
    for k in ("TEHMM_SPEC_CHUNK", "TEHMM_LANE_SUB", "TEHMM_LANE_WARMUP", "TEHMM_LANE_VIT"):
        os.environ.pop(k, None)
    os.environ.update(cfg)
    for N in (35, 7):
        model = synth.make_model(N, seed=3 + N)
        lens = [1, 63, 300, 1024, 2500, 4097, 6000, 9000, 20000, 60000]
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        obs = synth.sample_obs(model, int(offs[-1]), seed=1, missing=0.03)
        hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
        hb = HipBatch(obs, offs)
        res = hm.eval(hb, viterbi=True, posterior=True)
        print(cfg, N, {k: v for k, v in hb.timing().items() if k.startswith("count:")}, res["viterbi_logprob"][-1])
