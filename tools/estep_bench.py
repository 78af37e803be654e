#!/usr/bin/env python3
"""Secondary measurement (not the headline metric): fused Baum-Welch E-step throughput on the
config-4 shape -- 35 states, 12 tracks (10 multinomial + 2 gaussian), 100 kb training chunks."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
    syms, gauss = (2, 2, 3, 4, 5, 8, 12, 30, 6, 10, 250, 250), (10, 11)
    if os.environ.get("TRACKS"):              # e.g. TRACKS=2,2,3,4,5,8,12,30,6,10  (g suffix = gaussian bins)
        ent = os.environ["TRACKS"].split(",")
        syms = tuple(int(e.rstrip("g")) for e in ent)
        gauss = tuple(i for i, e in enumerate(ent) if e.endswith("g"))
    model = synth.make_model(35, syms, gauss, seed=0)
    total = int(mb * 1e6)
    lens = np.full(total // 100000, 100000, dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    dev = torch.device("cuda", 0)
    obs = bench.gen_obs_torch(model, lens, 5, dev)
    torch.cuda.synchronize()
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs,
                  symbols_per_track=model.symbols_per_track)
    hb = HipBatch(obs.data_ptr(), offs, device_ptrs=True, K=model.n_tracks)
    K, N, S = model.log_probs.shape
    for it in range(3):
        start, trans, st = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
        t0 = time.perf_counter()
        lp = hm.estep(hb, False, start, trans, st)
        dt = time.perf_counter() - t0
        print("E-step %d: %.1f Mb in %d chunks: %.3f s -> %.3e positions/s  (logprob %.6e, sum(trans)=%.3f, "
              "sum(obs)=%.1f)" % (it, mb, len(lens), dt, total / dt, lp, trans.sum(), st.sum()))
        print("   stages (ms):", {k: round(v, 2) for k, v in hb.timing().items()})


if __name__ == "__main__":
    main()
