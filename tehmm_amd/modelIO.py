"""Model (de)serialisation without pickle (the reference pickles the whole model object,
modelIO.py:17-32; unpickling a stranger's file executes code).  A model is its arrays plus a JSON
description of everything else teHmmEval / teHmmTrain read back from it: the track list with its value
maps (teHmmEval.py:149 re-encodes the evaluation data with the TRAINING symbols), the state-name map
(teHmmEval.py:297), the emission-model class with its gaussian parameters and the hmm-level flags.
One .npz, loaded with allow_pickle=False."""
import json

import numpy as np

FORMAT = "tehmm_amd.model.v2"
FORMAT_V1 = "tehmm_amd.model.v1"       # round-2 files: arrays only (no track list, value maps, flags)


def _load_v1(z):
    """Files written by the round-2 format: the arrays with default metadata (a multinomial emission model, no track
    list -- evaluation data must already be encoded with the training symbols)."""
    from .emission import IndependentMultinomialEmissionModel
    from .hmm import MultitrackHmm
    lp = z["log_probs"]
    eff = float(z["eff_seg_len"])
    em = IndependentMultinomialEmissionModel(lp.shape[1], [int(x) for x in z["symbols"]], fudge=float(z["fudge"]),
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


def _map_to_json(vm):
    from .track import CategoryMap, IdentityValueMap
    if vm is None:
        return None
    if isinstance(vm, IdentityValueMap):
        return {"type": "identity", "scale": vm.scale, "shift": vm.shift}
    if isinstance(vm, CategoryMap):
        def enc(key):                       # numeric tracks key on strings, categorical ones on the raw value
            return ["s", key] if isinstance(key, str) else ["r", repr(key)]
        return {"type": "category", "reserved": vm.reserved, "scale": vm.scale, "logBase": vm.logBase,
                "shift": vm.shift, "defaultVal": None if vm.defaultVal is None else enc(vm.defaultVal),
                "missingVal": int(vm.missingVal),
                "entries": [[enc(k), int(s)] for k, s in sorted(vm.fwd.items(), key=lambda kv: kv[1])]}
    raise TypeError("saveModel: value map of type %s cannot be serialised" % type(vm).__name__)


def _map_from_json(d):
    import ast
    from .track import CategoryMap, IdentityValueMap
    if d is None:
        return None
    if d["type"] == "identity":
        return IdentityValueMap(d["scale"], d["shift"])

    def dec(e):
        return e[1] if e[0] == "s" else ast.literal_eval(e[1])
    vm = CategoryMap(reserved=d["reserved"])
    vm.scale, vm.logBase, vm.shift = d["scale"], d["logBase"], d["shift"]
    vm.defaultVal = None if d["defaultVal"] is None else dec(d["defaultVal"])
    vm.missingVal = d["missingVal"]
    vm.fwd = {dec(k): int(s) for k, s in d["entries"]}
    vm.back = {s: k for k, s in vm.fwd.items()}
    return vm


def saveModel(path, hmm):
    from .emission import IndependentMultinomialAndGaussianEmissionModel
    hmm.validate()
    em = hmm.getEmissionModel()
    gp = getattr(em, "gaussParams", None)
    tl = hmm.getTrackList()
    meta = {
        "emission_class": "gaussian" if isinstance(em, IndependentMultinomialAndGaussianEmissionModel) else "multinomial",
        "zeroAsMissingData": bool(em.zeroAsMissingData),
        "uniformMixProb": float(em.uniformMixProb),
        "randRange": list(em.randRange),
        "tracks": None if tl is None else [{"name": t.getName(), "number": int(t.getNumber()), "dist": t.getDist(),
                                            "valueMap": _map_to_json(t.getValueMap())} for t in tl],
        "stateNameMap": _map_to_json(hmm.getStateNameMap()),
        "hmm": {"fudge": float(hmm.fudge), "fixTrans": bool(hmm.fixTrans), "fixEmission": bool(hmm.fixEmission),
                "fixStart": bool(hmm.fixStart), "maxProb": bool(hmm.maxProb),
                "maxProbCut": hmm.maxProbCut, "transMatEpsilons": bool(hmm.transMatEpsilons),
                "algorithm": hmm._algorithm, "n_iter": int(hmm.n_iter), "thresh": float(hmm.thresh),
                "numZeroInitEdges": int(hmm.numZeroInitEdges), "numZeroInitStarts": int(hmm.numZeroInitStarts),
                "last_forward_log_prob": hmm.last_forward_log_prob,
                "last_forward_log_prob_it": int(hmm.last_forward_log_prob_it)},
    }
    with open(path, "wb") as f:
        np.savez(f, format=np.asarray(FORMAT), log_transmat=hmm._log_transmat,
                 log_startprob=hmm._log_startprob,
                 log_probs=em.getLogProbs(),
                 symbols=np.asarray(em.getNumSymbolsPerTrack(), dtype=np.int64),
                 normalize_fac=float(em.normalizeFac), fudge=float(em.fudge),
                 eff_seg_len=np.asarray(-1.0 if em.effectiveSegmentLength is None else em.effectiveSegmentLength),
                 gauss_params=np.zeros(0) if gp is None else gp,
                 iteration=np.asarray(-1 if hmm.current_iteration is None else hmm.current_iteration),
                 meta=np.asarray(json.dumps(meta)))


def loadModel(path):
    from .emission import (IndependentMultinomialAndGaussianEmissionModel,
                           IndependentMultinomialEmissionModel)
    from .hmm import MultitrackHmm
    from .track import Track, TrackList
    with np.load(path, allow_pickle=False) as z:
        if str(z["format"]) == FORMAT_V1:
            return _load_v1(z)
        if str(z["format"]) != FORMAT:
            raise ValueError("loadModel: %s is not a %s file" % (path, FORMAT))
        meta = json.loads(str(z["meta"]))
        lp = z["log_probs"]
        eff = float(z["eff_seg_len"])
        symbols = [int(x) for x in z["symbols"]]
        tl = None
        if meta["tracks"] is not None:
            tl = TrackList(Track(t["name"], t["number"], t["dist"], _map_from_json(t["valueMap"]))
                           for t in meta["tracks"])
        kw = dict(zeroAsMissingData=meta["zeroAsMissingData"], fudge=float(z["fudge"]),
                  effectiveSegmentLength=None if eff < 0 else eff, randRange=tuple(meta["randRange"]))
        if meta["emission_class"] == "gaussian":
            if tl is None:
                raise ValueError("loadModel: a gaussian emission model needs its track list")
            em = IndependentMultinomialAndGaussianEmissionModel(lp.shape[1], symbols, tl, **kw)
            em.gaussParams = z["gauss_params"].copy()
        else:
            em = IndependentMultinomialEmissionModel(lp.shape[1], symbols, **kw)
        em.uniformMixProb = float(meta["uniformMixProb"])
        em.logProbs = lp.copy()
        em.normalizeFac = float(z["normalize_fac"])
        h = meta["hmm"]
        hmm = MultitrackHmm(em, algorithm=h["algorithm"], n_iter=h["n_iter"], thresh=h["thresh"],
                            state_name_map=_map_from_json(meta["stateNameMap"]), fudge=h["fudge"],
                            fixTrans=h["fixTrans"], fixEmission=h["fixEmission"], fixStart=h["fixStart"],
                            transMatEpsilons=h["transMatEpsilons"], maxProb=h["maxProb"], maxProbCut=h["maxProbCut"])
        hmm.trackList = tl
        # (transmat_ / startprob_ are views of the log tables, hmm.py:622-666: zeros stay -1e100)
        hmm._log_transmat = z["log_transmat"].copy()
        hmm._log_startprob = z["log_startprob"].copy()
        hmm.numZeroInitEdges = h["numZeroInitEdges"]
        hmm.numZeroInitStarts = h["numZeroInitStarts"]
        hmm.last_forward_log_prob = h["last_forward_log_prob"]
        hmm.last_forward_log_prob_it = h["last_forward_log_prob_it"]
        it = int(z["iteration"])
        hmm.current_iteration = None if it < 0 else it
    hmm.validate()
    return hmm
