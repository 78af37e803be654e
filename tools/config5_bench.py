"""BASELINE configs[4] shape alone: 100 states, 10 tracks, segment ratios, 2 Mb in 20 intervals.
usage: python tools/config5_bench.py [n_intervals]"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tehmm_amd import synth
from tehmm_amd.engine import HipBatch, HipModel

n_iv = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
m5 = synth.make_model(100, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=0)
l5 = np.full(n_iv, 100_000, dtype=np.int64)
o5 = np.concatenate([[0], np.cumsum(l5)]).astype(np.int64)
ob = bench.gen_obs_torch(m5, l5, seed=33, device=dev)
r5 = torch.full((int(o5[-1]),), 0.2, dtype=torch.float64, device=dev)
hm5 = HipModel(m5.log_transmat, m5.log_startprob, m5.log_probs, symbols_per_track=m5.symbols_per_track)
hb5 = HipBatch(ob.data_ptr(), o5, ratios=r5.data_ptr(), device_ptrs=True, K=m5.n_tracks)
for kw in (dict(viterbi=True, posterior=False), dict(viterbi=False, posterior=True), dict(viterbi=True, posterior=True)):
    hm5.eval(hb5, use_ratios=True, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hm5.eval(hb5, use_ratios=True, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(kw, "%.1f ms  %.3g positions/s" % (dt * 1e3, int(o5[-1]) / dt), json.dumps({k: round(v, 1) for k, v in hb5.timing().items()}), flush=True)
hb5.close(); hm5.close()
