"""ms per evaluation (Viterbi + posterior, every step a first evaluation) by batch size and TEHMM_DEFER mode.
usage: python tools/defer_sweep.py [Mb ...]   (SINGLE=1: one interval)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tehmm_amd import synth, _lib
from tehmm_amd.engine import HipBatch, HipModel

sizes = [float(a) for a in sys.argv[1:]] or [10, 20, 30, 50, 70, 100]
dev = torch.device("cuda", 0)
model = synth.make_model(bench.N_STATES, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=0)
hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, symbols_per_track=model.symbols_per_track)
lib = _lib.load()
for mb in sizes:
    total = int(mb * 1e6)
    lens = synth.interval_lengths(total, 200_000, 2_000_000, seed=1000)
    if os.environ.get("SINGLE"):
        lens = np.asarray([total], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = bench.gen_obs_torch(model, lens, seed=17, device=dev)
    torch.cuda.synchronize()
    hb = HipBatch(obs.data_ptr(), offs, device_ptrs=True, K=model.n_tracks)
    del obs
    torch.cuda.empty_cache()
    row = []
    for mode in ("0", "1", "3", "5"):
        os.environ["TEHMM_DEFER"] = mode
        hm.eval(hb, viterbi=True, posterior=True)
        best = 1e9
        for _ in range(3):
            lib.tehmm_batch_reset_cache(hb._h)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hm.eval(hb, viterbi=True, posterior=True)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        row.append(best)
    print("%6.0f Mb  (%3d intervals)  mode0 %.2f  mode1 %.2f  mode3 %.2f  mode5 %.2f" % (mb, len(lens), *row), flush=True)
    hb.close()
    lib.tehmm_trim_pools()
hm.close()
