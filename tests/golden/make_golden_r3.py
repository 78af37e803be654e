#!/usr/bin/env python3
"""Round-3 golden vectors, by RUNNING THE REAL REFERENCE in the build container (recipe: make_golden.py).

  bic     MultitrackHmm.getNumFreeParameters (hmm.py:490-520) for every fix{Trans,Start,Emission}
          combination of a 3-track model with one gaussian track, and the --bic file bin/teHmmEval.py
          writes (teHmmEval.py:216-234): that block of the script is cut out of the reference at generation
          time, 2to3-converted in the scratch dir (the text never enters the repository) and executed
          with the reference model object.
Re-run:  python tests/golden/make_golden_r3.py
"""
import os
import subprocess
import sys
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402

REF = mg.REF


def bic_block(root):
    """The `if args.bic is not None:` block of the reference's teHmmEval.main as a py3 code object."""
    lines = open(os.path.join(REF, "bin", "teHmmEval.py")).read().splitlines(True)
    start = next(i for i, l in enumerate(lines) if l.strip() == "if args.bic is not None:")
    end = next(i for i in range(start, len(lines)) if lines[i].strip() == "bicFile.close()") + 1
    tmp = os.path.join(root, "_bic_block.py")
    open(tmp, "w").write(textwrap.dedent("".join(lines[start:end])))
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", tmp],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return compile(open(tmp).read(), tmp, "exec")


class Args(object):
    bic = None


def main():
    root = mg.build_reference()
    from teHmm.hmm import MultitrackHmm
    from teHmm.emission import IndependentMultinomialAndGaussianEmissionModel
    from teHmm.track import Track
    code = bic_block(root)

    def make_tracks():
        t0, t1, t2 = Track(number=0), Track(number=1), Track(number=2)
        t0.name, t1.name, t2.name = "cat", "gauss", "cat2"
        t1.dist, t1.defaultVal, t1.scale = "gaussian", "0", 0.5
        t1._init()
        gmap = t1.getValueMap()
        for v in range(0, 24, 2):
            gmap.getMap(v, update=True)
        gmap.sort()
        return [t0, t1, t2]

    N = 4
    symbols = [5, 12, 3]
    out = {}
    combos = []
    for fixTrans in (False, True):
        for fixStart in (False, True):
            for fixEmission in (False, True):
                tracks = make_tracks()
                rs = np.random.RandomState(3)
                em = IndependentMultinomialAndGaussianEmissionModel(N, symbols, tracks, random_state=rs)
                h = MultitrackHmm(em, fixTrans=fixTrans, fixStart=fixStart, fixEmission=fixEmission)
                h.trackList = tracks
                k = h.getNumFreeParameters()
                combos.append((int(fixTrans), int(fixStart), int(fixEmission), int(k)))
                if (fixTrans, fixStart, fixEmission) == (False, True, False):
                    args = Args()
                    args.bic = os.path.join(root, "out.bic")
                    ns = {"args": args, "model": h, "totalScore": -123456.789012, "totalDatapoints": 3 * 41000,
                          "np": np}
                    exec(code, ns)
                    out["bic_text"] = np.asarray(open(args.bic).read())
                    out["bic_score"] = -123456.789012
                    out["bic_datapoints"] = 3 * 41000
    out["combos"] = np.asarray(combos, dtype=np.int64)
    out["n_states"] = N
    out["symbols"] = np.asarray(symbols, dtype=np.int64)
    mg.save("bic", **out)


if __name__ == "__main__":
    main()
