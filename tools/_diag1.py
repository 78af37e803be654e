import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from tehmm_amd import synth
from tehmm_amd.engine import HipBatch, HipModel
N=35
model = synth.make_model(N, (3, 5, 4, 30), (), seed=12 + N)
rs = np.random.RandomState(3 + N)
lens = [int(x) for x in rs.randint(20000, 45000, size=5)] + [1, 70, 1500, 1024, 2048 + 64, 4097]
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
obs = synth.sample_obs(model, int(offs[-1]), seed=5, missing=0.02)
K, _, S = model.log_probs.shape
for normalize in (1.0, 0.5):
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, normalize, model.symbols_per_track)
    hb = HipBatch(obs, offs)
    hm.eval(hb, viterbi=False, posterior=True)
    print("eval normalize", normalize, {k: v for k, v in hb.timing().items() if k.startswith("count")})
    start, trans, st = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
    hm.estep(hb, False, start, trans, st)
    print("estep normalize", normalize, {k: v for k, v in hb.timing().items() if k.startswith("count")})
    hb.close()
