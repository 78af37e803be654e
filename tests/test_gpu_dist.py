"""Two ranks sharing the one GPU of the test box (gloo for the collectives, HIP for the compute):
sharded evaluation and the sharded E-step with its packed all-reduce agree with a single-process
run -- bit-equal Viterbi paths, statistics within 1e-6."""
import os
import socket

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make():
    from tehmm_amd import synth
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    model = synth.make_model(12, (3, 5, 4, 30), (), seed=5, sparse=0.2)
    lens = [900, 1, 420, 77, 1310, 64, 333, 2000]
    tables = [synth.sample_obs(model, L, seed=70 + i, missing=0.02) for i, L in enumerate(lens)]
    em = IndependentMultinomialEmissionModel(12, model.symbols_per_track)
    em.logProbs = model.log_probs.copy()
    h = MultitrackHmm(em, fixStart=False)
    h.transmat_ = model.transmat.copy()
    h.startprob_ = np.exp(model.log_startprob)
    h.current_iteration = 1
    return h, tables


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tehmm_amd.dist import ShardedEvaluator, sharded_estep
        h, tables = _make()
        mine, res = ShardedEvaluator(lambda sub: h._eval_tables(sub, True, False)).run(tables)
        stats, logprob = sharded_estep(tables, lambda sub, st: h._do_estep(sub, st) if sub else 0.0,
                                       h._initialize_sufficient_statistics)
        q.put((rank, list(mine), res["viterbi_logprob"], [np.asarray(p) for p in res["paths"]],
               stats, logprob))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_match_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=500) for _ in range(2)], key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    h, tables = _make()
    single = h._eval_tables(tables, True, False)
    st = h._initialize_sufficient_statistics()
    lp = h._do_estep(tables, st)
    assert sorted(got[0][1] + got[1][1]) == list(range(len(tables)))
    for rank, mine, vlp, paths, stats, logprob in got:
        assert_array_equal(vlp, single["viterbi_logprob"])
        for a, b in zip(paths, single["paths"]):
            assert_array_equal(a, b)
        assert stats["nobs"] == len(tables)
        assert_allclose(logprob, lp, rtol=1e-9)
        for k in ("start", "trans", "obs"):
            assert_allclose(stats[k], st[k], rtol=1e-6, atol=1e-12)
