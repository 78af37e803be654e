"""Stage timings of the fused eval in isolation: posterior only, Viterbi only, both (bench workload).
usage: [STAGES=viterbi] [SINGLE=1] [TRACKS=8,30,12,250g,250g] python tools/stage_bench.py [Mb]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tehmm_amd import synth
from tehmm_amd.engine import HipBatch, HipModel

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
dev = torch.device("cuda", 0)
syms, gauss = synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN
if os.environ.get("TRACKS"):              # e.g. TRACKS=8,30,12,250g,250g  (g = gaussian bins)
    ent = os.environ["TRACKS"].split(",")
    syms = tuple(int(e.rstrip("g")) for e in ent)
    gauss = tuple(i for i, e in enumerate(ent) if e.endswith("g"))
model = synth.make_model(int(os.environ.get("STATES", bench.N_STATES)), syms, gauss, seed=0)
total = int(mb * 1e6)
lens = synth.interval_lengths(total, 200_000, 2_000_000, seed=1000)
if os.environ.get("SINGLE"):              # one interval of `Mb` (BASELINE configs[1] geometry)
    lens = np.asarray([total], dtype=np.int64)
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
obs = bench.gen_obs_torch(model, lens, seed=17, device=dev)
torch.cuda.synchronize()
hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, symbols_per_track=model.symbols_per_track)
hb = HipBatch(obs.data_ptr(), offs, device_ptrs=True, K=model.n_tracks)
del obs
torch.cuda.empty_cache()
import time
only = os.environ.get("STAGES")          # e.g. STAGES=posterior
for name, kw in (("posterior", dict(viterbi=False, posterior=True)), ("viterbi", dict(viterbi=True, posterior=False)),
                 ("both", dict(viterbi=True, posterior=True))):
    if only and name not in only.split(","):
        continue
    hm.eval(hb, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hm.eval(hb, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print(name, "%.1f ms" % dt, json.dumps({k: round(v, 2) for k, v in hb.timing().items()}), flush=True)
hb.close()
hm.close()
