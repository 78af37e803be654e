#!/usr/bin/env python3
"""Diagnostic (not product): where a wave of k_vit_lane3 spends its cycles per step (needs a -DTEHMM_STAMPS library,
tools/devbuild.sh stamps 36 -DTEHMM_STAMPS; select it with TEHMM_HIP_LIB).  python tools/stamps_lane3.py [Mb]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from tehmm_amd import _lib, synth
from tehmm_amd.engine import HipBatch, HipModel
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
dev = torch.device("cuda", 0)
model = synth.make_model(35, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=0)
lens = synth.interval_lengths(int(mb * 1e6), 200_000, 2_000_000, seed=1000)
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
obs = bench.gen_obs_torch(model, lens, seed=17, device=dev)
hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, symbols_per_track=model.symbols_per_track)
hb = HipBatch(obs.data_ptr(), offs, device_ptrs=True, K=model.n_tracks)
for _ in range(2):
    hm.eval(hb, viterbi=True, posterior=False)
print({k: round(v, 2) for k, v in hb.timing().items()})
n = 2 * 4096 * 16
buf = (ctypes.c_uint64 * n)()
_lib.check(_lib.load().tehmm_debug_read_stamps(buf, n), "stamps")
a = np.frombuffer(buf, dtype=np.uint64)[:1024 * 4 * 8].reshape(1024, 4, 8).astype(np.float64)[:, :3, :]
a = a[a.sum(axis=(1, 2)) > 0]
hw = a[:, :, 7].astype(np.int64)
simd = (hw >> 4) & 3
print("SIMD of wave 0/1/2, first 12 workgroups:", simd[:12].tolist())
print("waves per SIMD id over the sampled workgroups:", np.bincount(simd.ravel(), minlength=4).tolist())
a[:, :, 7] = 0
steps = 512 + 32
names = ("barrier", "finish(read W..)", "group0", "group1", "group2", "group3+", "ballot..barrier", "-")
print("workgroups", a.shape[0], "cycles per step and wave:")
for w in range(3):
    print("  wave %d: " % w + "  ".join("%s %.0f" % (names[i], a[:, w, i].mean() / steps) for i in range(7)) +
          "  total %.0f" % (a[:, w, :].sum(axis=1).mean() / steps))
hb.close(); hm.close()
