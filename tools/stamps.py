#!/usr/bin/env python3
"""Diagnostic (not part of the product): builds libtehmm_hip_diag.so with -DTEHMM_STAMPS, runs one
Viterbi-only and one posterior-only evaluation of a few long intervals and prints where each wave
of the cooperative kernels spends its cycles (work vs. waiting at the block barrier)."""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DIAG = os.path.join(ROOT, "tehmm_amd", "libtehmm_hip_diag.so")


def build(extra=()):
    src = os.path.join(ROOT, "tehmm_amd", "csrc", "tehmm_hip.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off",
                           "-fPIC", "-shared", "-std=c++17", "-DTEHMM_STAMPS"] + list(extra) +
                          ["-o", DIAG, src])


def main():
    if "--build" in sys.argv:
        build([a for a in sys.argv[1:] if a.startswith("-D")])
        return
    os.environ["TEHMM_HIP_LIB"] = DIAG
    from tehmm_amd import _lib, synth
    from tehmm_amd.engine import HipBatch, HipModel
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    T = int(float(args[0])) if args else 200000
    n = 4
    model = synth.make_model(35, seed=0)
    offs = np.arange(n + 1, dtype=np.int64) * T
    obs = synth.random_obs(model, n * T, seed=1)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs,
                  symbols_per_track=model.symbols_per_track)
    hb = HipBatch(obs, offs)
    buf = (ctypes.c_uint64 * (n * 16))()
    for what, kw, names in (("viterbi", dict(viterbi=True, posterior=False),
                             ["chain", "emis+args", "args", "args"]),
                            ("forward_backward", dict(viterbi=False, posterior=True),
                             ["fwd chain", "bwd chain", "fwd emis", "bwd emis"])):
        hm.eval(hb, **kw)
        hm.eval(hb, **kw)
        ms = hb.timing()
        _lib.check(_lib.load().tehmm_debug_read_stamps(buf, n * 16), "stamps")
        a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4, 4).astype(np.float64)
        print("== %s: T=%d  kernel ms=%s" % (what, T, {k: round(v, 2) for k, v in ms.items()}))
        for w in range(4):
            wk = a[:, w, 0].mean() + a[:, w, 1].mean()
            print("  wave %d (%-10s): work %8.1f cyc/pos (a=%.1f b=%.1f)  barrier wait %8.1f cyc/pos"
                  % (w, names[w], wk / T, a[:, w, 0].mean() / T, a[:, w, 1].mean() / T,
                     a[:, w, 2].mean() / T))


if __name__ == "__main__":
    main()
