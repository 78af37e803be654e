#!/usr/bin/env python3
"""Model variants (sparse / sticky transitions) on the bench geometry, 30 Mb: stage times and exact-block counts for
a list of forward / backward warm-up lengths (TEHMM_LANE_WARMUP; 0 = the library's own choice).
python tools/variant_bench.py [warmups...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    warmups = [int(x) for x in sys.argv[1:]] or [0]
    dev = torch.device("cuda", 0)
    lens = synth.interval_lengths(30_000_000, 200_000, 2_000_000, seed=1000)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for tag, kw in (("dense", {}), ("sparse_p0.5", dict(sparse=0.5)), ("sticky_0.995", dict(stay=0.995)),
                    ("sparse_sticky", dict(sparse=0.5, stay=0.995))):
        mdl = synth.make_model(35, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=0, **kw)
        ob = bench.gen_obs_torch(mdl, lens, seed=31, device=dev)
        hm = HipModel(mdl.log_transmat, mdl.log_startprob, mdl.log_probs, symbols_per_track=mdl.symbols_per_track)
        for wu in warmups:
            if wu:
                os.environ["TEHMM_LANE_WARMUP"] = str(wu)
            else:
                os.environ.pop("TEHMM_LANE_WARMUP", None)
            hb = HipBatch(ob.data_ptr(), offs, device_ptrs=True, K=mdl.n_tracks)
            d = bench.time_eval(hm, hb, torch, viterbi=True, posterior=True)
            t = hb.timing()
            print("%-14s warmup %3d: %.1f ms  " % (tag, wu, d * 1e3),
                  {k: round(v, 1) for k, v in t.items() if not k.startswith("count:")},
                  {k.split(":")[1]: int(v) for k, v in t.items() if k.startswith("count:")}, flush=True)
            hb.close()
        hm.close()
        del ob


if __name__ == "__main__":
    main()
