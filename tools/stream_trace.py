"""Where the time of engine.eval_stream goes: per-group timestamps of the producer (create, eval) and the consumer
(fetch paths, fetch posteriors, close).  python tools/stream_trace.py [Mb] [group_rows]"""
import os, sys, time, threading, queue
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tehmm_amd import synth
from tehmm_amd.engine import HipBatch, HipModel

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
group_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
model = synth.make_model(35, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=0)
lens = synth.interval_lengths(int(mb * 1e6), 200_000, 2_000_000, seed=1000)
offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
obs = bench.gen_obs_torch(model, lens, seed=17, device=torch.device("cuda", 0)).cpu().numpy()
if os.environ.get("PINNED_OBS"):
    from tehmm_amd import _lib
    po = _lib.pinned_empty(obs.shape, np.uint8)
    po[...] = obs
    obs = po
hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, symbols_per_track=model.symbols_per_track)
n = len(offsets) - 1
groups, g0 = [], 0
while g0 < n:
    g1 = g0 + 1
    while g1 < n and offsets[g1 + 1] - offsets[g0] <= group_rows:
        g1 += 1
    groups.append((g0, g1))
    g0 = g1
for rep in range(3):
    T0 = time.perf_counter()
    log = []
    ready = queue.Queue(maxsize=2)

    def producer():
        for (a, b) in groups:
            t1 = time.perf_counter()
            r0, r1 = int(offsets[a]), int(offsets[b])
            hb = HipBatch(obs[r0:r1], offsets[a:b + 1] - offsets[a])
            t2 = time.perf_counter()
            res = hm.eval(hb, viterbi=True, posterior=True)
            t3 = time.perf_counter()
            log.append(("P", a, round((t1 - T0) * 1e3, 1), round((t2 - T0) * 1e3, 1), round((t3 - T0) * 1e3, 1)))
            ready.put((a, b, hb, res))
    th = threading.Thread(target=producer, daemon=True)
    th.start()
    keep = []
    for _ in groups:
        a, b, hb, res = ready.get()
        t1 = time.perf_counter()
        p = hb.paths()
        t2 = time.perf_counter()
        q = hb.posteriors()
        t3 = time.perf_counter()
        hb.close()
        t4 = time.perf_counter()
        keep.append((p, q))
        log.append(("C", a, round((t1 - T0) * 1e3, 1), round((t2 - T0) * 1e3, 1), round((t3 - T0) * 1e3, 1), round((t4 - T0) * 1e3, 1)))
    th.join()
    print("rep", rep, "total %.1f ms" % ((time.perf_counter() - T0) * 1e3))
    for e in sorted(log, key=lambda e: e[2]):
        print("   ", e)
    del keep
