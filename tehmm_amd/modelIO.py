"""Model (de)serialisation without pickle (the reference pickles the whole model object,
modelIO.py:17-32; unpickling a stranger's file executes code).  A model is its arrays: one .npz."""
import numpy as np


def saveModel(path, hmm):
    em = hmm.getEmissionModel()
    gp = getattr(em, "gaussParams", None)
    with open(path, "wb") as f:
        np.savez(f, format=np.asarray("tehmm_amd.model.v1"), log_transmat=hmm._log_transmat,
                 log_startprob=hmm._log_startprob, log_probs=em.getLogProbs(),
                 symbols=np.asarray(em.getNumSymbolsPerTrack(), dtype=np.int64),
                 normalize_fac=float(em.normalizeFac), fudge=float(em.fudge),
                 eff_seg_len=np.asarray(-1.0 if em.effectiveSegmentLength is None else em.effectiveSegmentLength),
                 gauss_params=np.zeros(0) if gp is None else gp,
                 iteration=np.asarray(-1 if hmm.current_iteration is None else hmm.current_iteration))


def loadModel(path):
    from .emission import IndependentMultinomialEmissionModel
    from .hmm import MultitrackHmm
    with np.load(path, allow_pickle=False) as z:
        assert str(z["format"]) == "tehmm_amd.model.v1"
        lp = z["log_probs"]
        eff = float(z["eff_seg_len"])
        em = IndependentMultinomialEmissionModel(lp.shape[1], [int(x) for x in z["symbols"]],
                                                 fudge=float(z["fudge"]),
                                                 effectiveSegmentLength=None if eff < 0 else eff)
        em.logProbs = lp.copy()
        em.normalizeFac = float(z["normalize_fac"])
        if z["gauss_params"].size:
            em.gaussParams = z["gauss_params"].copy()
        hmm = MultitrackHmm(em)
        hmm._log_transmat = z["log_transmat"].copy()
        hmm._log_startprob = z["log_startprob"].copy()
        it = int(z["iteration"])
        hmm.current_iteration = None if it < 0 else it
    return hmm
