#!/usr/bin/env python3
"""Round-2 golden vectors, again by RUNNING THE REAL REFERENCE in the build container (see
make_golden.py for the recipe: scratch copy outside the repo, 2to3 + Cython-3 patches, import).

Covers the SURVEY section 8(f) rows and the a7 dtype variants:
  counts_*        _emission.fastUpdateCounts U8/U16/32 (+ segment ratios), fastAccumulateStats U16/32
  supervised_*    MultitrackHmm.supervisedTrain on two tables (unsegmented / segmented with ratios)
  segment_*       IntegerTrackTable.segment (mode / gaussian-mean interpolation, compression),
                  setMaskTable + getMaskRunningOffsets (_track.runSum), segment on a masked table
  states_to_bed   bin/teHmmEval.py statesToBed lines for a segmented + masked table
  mstep_gauss     one EM iteration with a gaussian track (emission.maximize + makeGaussian, Q19)

Runtime shims (documented, none touches arithmetic): integer division in common.binSearch (a
Python-2 `/`), scipy.stats.mode(keepdims=True) (old scipy returned arrays).
Re-run:  python tests/golden/make_golden_r2.py
"""
import ast
import io
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402  (build_reference, save)

REF = mg.REF


def py3_function_from(path, name, root):
    """Source of ONE function of a reference script, 2to3-converted in the scratch dir (the text
    never enters the repository), compiled into a function object."""
    tmp = os.path.join(root, "_fn_" + name + ".py")
    src = open(path).read()
    tree = ast.parse(src) if False else None      # (py2 source: cannot be parsed by py3's ast)
    del tree
    lines = src.splitlines(True)
    start = next(i for i, l in enumerate(lines) if l.startswith("def %s(" % name))
    end = next((i for i in range(start + 1, len(lines))
                if lines[i][:1] not in (" ", "\t", "\n", "#", "")), len(lines))
    open(tmp, "w").write("".join(lines[start:end]))
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", tmp],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ns = {"np": np}
    exec(compile(open(tmp).read(), tmp, "exec"), ns)
    return ns[name]


class DuckTrackData(object):
    def __init__(self, tables, trackList=None):
        self.tables = tables
        self.trackList = trackList

    def getTrackTableList(self):
        return self.tables

    def getTrackList(self):
        return self.trackList


def main():
    root = mg.build_reference()
    import teHmm.track as rtrack
    import teHmm.common as rcommon
    from teHmm import _emission
    from teHmm.hmm import MultitrackHmm
    from teHmm.emission import (IndependentMultinomialEmissionModel,
                                IndependentMultinomialAndGaussianEmissionModel)
    from teHmm.track import IntegerTrackTable, Track, CategoryMap
    from scipy import stats as sstats
    from tehmm_amd import synth

    # ---- shims for Python-2 semantics the 2to3 pass cannot see
    def bin_search(items, val, idx=[0, 1], first=None, last=None):
        if first is None:
            first = 0
        if last is None:
            last = len(items) - 1
        pivot = (first + last) // 2
        pv = tuple([items[pivot][i] for i in idx])
        if first == last:
            return first if pv == val else None
        if pv == val:
            return pivot
        elif pv > val:
            return bin_search(items, val, idx, first, pivot)
        return bin_search(items, val, idx, pivot + 1, last)
    rtrack.binSearch = bin_search
    rcommon.binSearch = bin_search
    rtrack.mode = lambda a: sstats.mode(a, keepdims=True)

    def table(K, chrom, start, data, seg_lens=None, dtype=np.uint8):
        """Reference IntegerTrackTable holding `data`; seg_lens: already-compressed segmented table."""
        total = int(np.sum(seg_lens)) if seg_lens is not None else data.shape[0]
        tab = IntegerTrackTable(K, chrom, start, start + total, dtype=dtype)
        tab.data = np.ascontiguousarray(data, dtype=dtype)
        if seg_lens is not None:
            tab.segOffsets = np.concatenate([[0], np.cumsum(seg_lens)[:-1]]).astype(np.int64)
        tab.shape = tab.data.shape
        return tab

    # ------------------------------------------------------------------ 1. counts
    rs = np.random.RandomState(5)
    N, T = 5, 700
    for dtype, tag, syms in ((np.uint8, "u8", (4, 6, 200)), (np.uint16, "u16", (4, 6, 600)),
                             (np.int32, "i32", (4, 6, 600))):
        K = len(syms)
        S = max(syms) + 1
        obs = np.stack([rs.randint(0, s + 1, size=T) for s in syms], axis=1).astype(dtype)
        for with_ratio in (0, 1):
            seg_lens = np.minimum(1 + rs.geometric(1 / 20.0, size=T), 100) if with_ratio else None
            tab = table(K, "chr1", 1000, obs, seg_lens, dtype)
            ratios = tab.getSegmentLengthsAsRatio(20.0) if with_ratio else None
            # labelled intervals in TABLE coordinates (what getOverlapInTableCoords hands over)
            cuts = np.sort(rs.choice(np.arange(1, T), size=24, replace=False))
            ivs = [("chr1", int(a), int(b), int(rs.randint(0, N))) for a, b in zip(cuts[:-1:2], cuts[1::2])]
            stats = 0.25 + np.zeros((K, N, S))
            for iv in ivs:
                _emission.fastUpdateCounts(iv, tab, stats, ratios)
            post = rs.dirichlet(np.ones(N), size=T)
            acc = np.zeros((K, N, S))
            _emission.fastAccumulateStats(obs, acc, post, ratios)
            mg.save("counts_%s_r%d" % (tag, with_ratio), obs=obs,
                    ratios=(ratios if ratios is not None else np.zeros(0)),
                    iv_start=np.asarray([i[1] for i in ivs], dtype=np.int64),
                    iv_end=np.asarray([i[2] for i in ivs], dtype=np.int64),
                    iv_state=np.asarray([i[3] for i in ivs], dtype=np.int32),
                    stats_init=0.25, stats=stats, post=post, acc=acc, n_states=N)

    # ------------------------------------------------------------------ 2. supervisedTrain
    for with_ratio in (0, 1):
        model = synth.make_model(4, (3, 5, 4), (), seed=61)
        tabs, obs_l, lens_l, starts = [], [], [], [500, 4000]
        for i, T in enumerate((300, 220)):
            o = synth.sample_obs(model, T, seed=70 + i, missing=0.05)
            ln = np.minimum(1 + np.random.RandomState(80 + i).geometric(1 / 8.0, size=T), 40) if with_ratio else None
            tabs.append(table(3, "chrA", starts[i], o, ln))
            obs_l.append(o)
            lens_l.append(ln if ln is not None else np.zeros(0, dtype=np.int64))
        # sorted, non-overlapping labelled intervals in GENOME coordinates; some abut (transition
        # counts), some straddle a table end, one lies between the tables
        rs = np.random.RandomState(90 + with_ratio)
        beds = []
        for i, tb in enumerate(tabs):
            lo, hi = tb.getStart() - 30, tb.getEnd() + 30
            cuts = np.sort(rs.choice(np.arange(lo, hi), size=14, replace=False))
            for a, b in zip(cuts[:-1], cuts[1:]):
                if rs.rand() < 0.75:
                    beds.append(("chrA", int(a), int(b), int(rs.randint(0, 4))))
        em = IndependentMultinomialEmissionModel(4, model.symbols_per_track, fudge=0.1,
                                                 effectiveSegmentLength=(8 if with_ratio else None))
        h = MultitrackHmm(em, fudge=0.1)
        h.supervisedTrain(DuckTrackData(tabs), beds)
        mg.save("supervised_r%d" % with_ratio, obs0=obs_l[0], obs1=obs_l[1], seg_lens0=lens_l[0],
                seg_lens1=lens_l[1], starts=np.asarray(starts), eff_len=8, fudge=0.1,
                symbols=np.asarray(model.symbols_per_track),
                bed_start=np.asarray([b[1] for b in beds], dtype=np.int64),
                bed_end=np.asarray([b[2] for b in beds], dtype=np.int64),
                bed_state=np.asarray([b[3] for b in beds], dtype=np.int64),
                transmat=h.transmat_, log_transmat=h._log_transmat, startprob=h.startprob_,
                log_probs=h.emissionModel.logProbs)

    # ------------------------------------------------------------------ 3. segmentation / masking
    def make_tracks():
        t0, t1, t2 = Track(number=0), Track(number=1), Track(number=2)
        t0.name, t1.name, t2.name = "cat", "gauss", "cat2"
        t1.dist, t1.defaultVal, t1.scale = "gaussian", "0", 0.5
        t1._init()
        return [t0, t1, t2]

    rs = np.random.RandomState(17)
    T = 900
    tracks = make_tracks()
    gmap = tracks[1].getValueMap()
    raw = np.round(np.abs(rs.normal(40, 25, size=T))).astype(int)        # raw values of the gaussian track
    for v in sorted(set(raw.tolist())):
        gmap.getMap(v, update=True)
    gmap.sort()
    data = np.zeros((T, 3), dtype=np.uint8)
    data[:, 0] = np.repeat(rs.randint(1, 6, size=T // 6 + 1), 6)[:T]     # runs, so modes are non-trivial
    data[:, 1] = [gmap.getMap(v) for v in raw]
    data[:, 2] = rs.randint(1, 4, size=T)
    seg_lens = []
    while sum(seg_lens) < T:
        seg_lens.append(min(int(1 + rs.geometric(1 / 7.0)), T - sum(seg_lens)))
    seg_lens = np.asarray(seg_lens, dtype=np.int64)
    seg_start = 2000 + np.concatenate([[0], np.cumsum(seg_lens)[:-1]])
    segIntervals = [("chrS", int(a), int(a + l)) for a, l in zip(seg_start, seg_lens)]
    mapback_before = gmap.getMapBackTable(np.uint8).copy()
    n_before = len(gmap)
    tab = IntegerTrackTable(3, "chrS", 2000, 2000 + T)
    tab.data = data.copy()
    tab.segment(segIntervals, tracks, interpolate=True)
    mg.save("segment_plain", data=data, seg_lens=seg_lens, start=2000, gauss_track=1,
            mapback=np.where(mapback_before > 1e300, np.nan, mapback_before), gauss_scale=0.5,
            gauss_default=0.0, gauss_reserved=1, n_symbols_before=n_before, n_symbols_after=len(gmap),
            mapback_after=np.where(gmap.getMapBackTable(np.uint8) > 1e300, np.nan, gmap.getMapBackTable(np.uint8)),
            out_data=tab.data, out_offsets=np.asarray(tab.segOffsets, dtype=np.int64),
            ratios=tab.getSegmentLengthsAsRatio(10.0))

    # masked: binary mask tracks (1 = False, 2 = True; covered positions are cut out), mask runs are
    # aligned with segment boundaries as the reference requires (track.py:469-473)
    tracks = make_tracks()
    gmap = tracks[1].getValueMap()
    for v in sorted(set(raw.tolist())):
        gmap.getMap(v, update=True)
    gmap.sort()
    mask = np.ones((T, 2), dtype=np.uint8)
    pos = 0
    for i, l in enumerate(seg_lens):
        if i % 5 == 3:
            mask[pos:pos + l, i % 2] = 2
        pos += l
    mtab = IntegerTrackTable(2, "chrS", 2000, 2000 + T)
    mtab.data = mask.copy()
    tab = IntegerTrackTable(3, "chrS", 2000, 2000 + T)
    tab.data = data.copy()
    tab.setMaskTable(mtab)
    keep = tab.maskArray.copy()
    run_masked = tab.getMaskRunningOffsets()
    run_full = tab.getMaskRunningOffsets(reverseTransform=True)
    data_masked = tab.data.copy()
    tab.segment(segIntervals, tracks, interpolate=True)
    # statesToBed of the real script on this segmented + masked table
    states_to_bed = py3_function_from(os.path.join(REF, "bin", "teHmmEval.py"), "statesToBed", root)
    n_rows = len(tab)
    states = rs.randint(0, 4, size=n_rows)
    post = rs.dirichlet(np.ones(4), size=n_rows)
    pmask = np.asarray([1, 0, 1, 0])
    bed_io, post_io = io.StringIO(), io.StringIO()
    states_to_bed(tab, states, bed_io, post, pmask, post_io, None, None, None)
    mg.save("segment_masked", data=data, mask=mask, seg_lens=seg_lens, start=2000, gauss_track=1,
            keep=keep.astype(np.uint8), run_masked=np.asarray(run_masked, dtype=np.int32),
            run_full=np.asarray(run_full, dtype=np.int32), data_masked=data_masked,
            mapback=np.where(mapback_before > 1e300, np.nan, mapback_before), gauss_scale=0.5,
            out_data=tab.data, out_offsets=np.asarray(tab.segOffsets, dtype=np.int64),
            table_end=tab.getEnd(), states=states, post=post, post_mask=pmask,
            bed_text=np.frombuffer(bed_io.getvalue().encode(), dtype=np.uint8),
            post_text=np.frombuffer(post_io.getvalue().encode(), dtype=np.uint8))

    # ------------------------------------------------------------------ 4. M-step with a gaussian track
    tracks = make_tracks()
    gmap = tracks[1].getValueMap()
    for v in range(0, 60, 2):
        gmap.getMap(v, update=True)
    gmap.sort()
    nsym = [5, len(gmap), 3]
    rs = np.random.RandomState(23)
    Nst = 4
    emg = IndependentMultinomialAndGaussianEmissionModel(Nst, nsym, tracks, fudge=0.0, randomize=True,
                                                         random_state=np.random.RandomState(3))
    lp0 = emg.logProbs.copy()
    A = rs.rand(Nst, Nst) + np.eye(Nst) * 3
    A /= A.sum(axis=1, keepdims=True)
    hg = MultitrackHmm(emg, n_iter=2, thresh=0.0, fixStart=False)
    hg.transmat_ = A
    hg.startprob_ = np.full(Nst, 1.0 / Nst)
    hg.init_params = ""
    hg.trackList = tracks
    seqs = []
    for i, T in enumerate((260, 180)):
        o = np.stack([rs.randint(1, n + 1, size=T) for n in nsym], axis=1).astype(np.uint8)
        o[rs.rand(T) < 0.03, 0] = 0
        seqs.append(o)
    captured = {}
    orig = hg._do_mstep

    def spy(stats, params, _o=orig, _c=captured):
        if "obs" not in _c:
            _c.update(start=stats["start"].copy(), trans=stats["trans"].copy(), obs=stats["obs"].copy())
        return _o(stats, params)
    hg._do_mstep = spy
    hg.fit(seqs)
    mback = np.asarray([float(gmap.getMapBack(s)) if gmap.getMapBack(s) is not None else np.nan
                        for s in range(nsym[1] + 1)])
    mg.save("mstep_gauss", obs0=seqs[0], obs1=seqs[1], symbols=np.asarray(nsym), log_probs=lp0, transmat=A,
            gauss_track=1, gauss_values=mback, uniform_mix=0.1,
            stats_start=captured["start"], stats_trans=captured["trans"], stats_obs=captured["obs"],
            transmat_after=hg.transmat_, startprob_after=hg.startprob_, log_probs_after=hg.emissionModel.logProbs,
            gauss_params_after=hg.emissionModel.gaussParams, last_logprob=hg.last_forward_log_prob)

    shutil.rmtree(root, ignore_errors=True)
    print("done; reference scratch build removed:", root)


if __name__ == "__main__":
    main()
