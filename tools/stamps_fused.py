#!/usr/bin/env python3
"""Diagnostic (not part of the product): where a wave of k_fused_fwd spends its cycles per step.
  python tools/stamps_fused.py --build     (libtehmm_hip_diag.so with -DTEHMM_STAMPS -DTEHMM_DEV_NT=36)
  python tools/stamps_fused.py [Mb]"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DIAG = os.path.join(ROOT, "tehmm_amd", "libtehmm_hip_diag.so")
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared",
                           "-std=c++17", "-DTEHMM_STAMPS", "-DTEHMM_DEV_NT=36"] + [a for a in sys.argv if a.startswith("-D")] + ["-o", DIAG,
                           os.path.join(ROOT, "tehmm_amd", "csrc", "tehmm_hip.hip")])
    sys.exit(0)
os.environ.setdefault("TEHMM_HIP_LIB", DIAG)
import torch, bench
from tehmm_amd import _lib, synth
from tehmm_amd.engine import HipBatch, HipModel
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda", 0)
model = synth.make_model(35, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=0)
lens = synth.interval_lengths(int(mb * 1e6), 200_000, 2_000_000, seed=1000)
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
obs = bench.gen_obs_torch(model, lens, seed=17, device=dev)
hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, symbols_per_track=model.symbols_per_track)
hb = HipBatch(obs.data_ptr(), offs, device_ptrs=True, K=model.n_tracks)
for _ in range(2):
    hm.eval(hb, viterbi=False, posterior=True)
print({k: round(v, 2) for k, v in hb.timing().items()})
n = 2 * 4096 * 16
buf = (ctypes.c_uint64 * n)()
_lib.check(_lib.load().tehmm_debug_read_stamps(buf, n), "stamps")
steps = 512 + 64
for name, a in zip(("forward", "backward"), np.frombuffer(buf, dtype=np.uint64).reshape(2, 4096, 4, 4).astype(np.float64)):
    a = a[a.sum(axis=(1, 2)) > 0]
    print(name, "waves", a.shape[0] * 4, "cycles per step: mfma+slots %.0f  finish %.0f  tail %.0f  stores+loop %.0f  total %.0f"
          % tuple(list(a.mean(axis=(0, 1)) / steps) + [a.sum(axis=2).mean() / steps]))
hb.close(); hm.close()
