#!/usr/bin/env python3
"""PCIe-inclusive timing of a fresh 20 Mb batch, repeated: H2D of the observations, workspace allocation, evaluation,
D2H of the paths and the masked posterior sums (or the full posteriors).  python tools/e2e_check.py [repeats]"""
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
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda", 0)
    model = synth.make_model(35, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=0)
    lens = synth.interval_lengths(20_000_000, 200_000, 2_000_000, seed=1000)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    host_obs = bench.gen_obs_torch(model, lens, seed=17, device=dev).cpu().numpy()
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, symbols_per_track=model.symbols_per_track)
    mask = (np.arange(model.n_states) % 3 == 0).astype(np.float64)
    for full in (False, True):
        for r in range(reps):
            t0 = time.perf_counter()
            hb = HipBatch(host_obs, offs)
            t1 = time.perf_counter()
            hm.eval(hb, viterbi=True, posterior=True)
            t2 = time.perf_counter()
            p = hb.paths()
            q = hb.posteriors() if full else hb.posterior_masksum(mask)
            t3 = time.perf_counter()
            print("%s rep %d: batch create (H2D) %.1f ms, eval %.1f ms, D2H %.1f ms, total %.1f ms" % (
                "full posteriors" if full else "masked sums", r, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3,
                (t3 - t0) * 1e3), flush=True)
            hb.close()
            del p, q


if __name__ == "__main__":
    main()
