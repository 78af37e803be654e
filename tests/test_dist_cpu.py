"""world_size-2 gloo tests of the multi-GPU layer (sharding, gathers, the packed all-reduce of
E-step statistics).  The per-rank compute is the CPU oracle here (no GPU in this container); on
the GPU box the same layer wraps the HIP path (tests/test_gpu_dist.py)."""
import os
import socket

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal


def test_lpt_shard_balance_and_cover():
    from tehmm_amd.dist import lpt_shard
    rs = np.random.RandomState(0)
    lens = rs.randint(200_000, 2_000_000, size=97)
    for world in (1, 2, 4, 8):
        sh = lpt_shard(lens, world)
        allidx = np.sort(np.concatenate(sh))
        assert_array_equal(allidx, np.arange(97))
        loads = np.asarray([lens[s].sum() for s in sh])
        assert loads.max() - loads.min() <= lens.max()
    assert [list(x) for x in lpt_shard([5, 1, 1], 4)] == [[0], [1], [2], []]


def test_pack_unpack_roundtrip():
    from tehmm_amd.dist import pack_stats, unpack_stats
    rs = np.random.RandomState(1)
    st = {"nobs": 7, "start": rs.rand(5), "trans": rs.rand(5, 5), "obs": rs.rand(3, 5, 9)}
    out, lp = unpack_stats(pack_stats(st, -12.5), st)
    assert out["nobs"] == 7 and lp == -12.5
    for k in ("start", "trans", "obs"):
        assert_array_equal(out[k], st[k])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from tehmm_amd import synth
        from tehmm_amd.dist import ShardedEvaluator, sharded_estep
        model = synth.make_model(6, (3, 5, 4), (), seed=21)
        lens = [300, 1, 120, 77, 510, 64, 33]
        tables = [synth.sample_obs(model, L, seed=40 + i, missing=0.02) for i, L in enumerate(lens)]
        pi, lt, lp = model.log_startprob, model.log_transmat, model.log_probs

        def compute(sub):
            out = {"viterbi_logprob": [], "paths": [], "forward_logprob": []}
            for t in sub:
                v, p = oracle.decode(t, lp, pi, lt)
                f, _ = oracle.score_samples(t, lp, pi, lt)
                out["viterbi_logprob"].append(v)
                out["paths"].append(p)
                out["forward_logprob"].append(f)
            return out
        mine, res = ShardedEvaluator(compute).run(tables)

        def empty():
            K, N, S = lp.shape
            return {"nobs": 0, "start": np.zeros(N), "trans": np.zeros((N, N)),
                    "obs": np.full((K, N, S), 0.25)}      # non-zero initial value (fudge)

        def estep_fn(sub, stats):
            st = oracle.estep(sub, lp, pi, lt)
            stats["nobs"] += st["nobs"]
            for k in ("start", "trans", "obs"):
                stats[k] += st[k]
            return st["logprob"]
        stats, logprob = sharded_estep(tables, estep_fn, empty)
        q.put((rank, list(mine), res, stats, logprob))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_sharded_eval_and_estep_match_serial():
    import torch.multiprocessing as mp
    from oracle import oracle
    from tehmm_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = synth.make_model(6, (3, 5, 4), (), seed=21)
    lens = [300, 1, 120, 77, 510, 64, 33]
    tables = [synth.sample_obs(model, L, seed=40 + i, missing=0.02) for i, L in enumerate(lens)]
    pi, lt, lp = model.log_startprob, model.log_transmat, model.log_probs
    serial = oracle.estep(tables, lp, pi, lt)
    shards = sorted(got, key=lambda g: g[0])
    assert sorted(shards[0][1] + shards[1][1]) == list(range(7))
    for rank, mine, res, stats, logprob in shards:
        for i, t in enumerate(tables):                    # every rank holds the gathered results
            v, p = oracle.decode(t, lp, pi, lt)
            assert res["viterbi_logprob"][i] == v
            assert_array_equal(res["paths"][i], p)
        assert stats["nobs"] == 7
        assert_allclose(logprob, serial["logprob"], rtol=1e-12)
        assert_allclose(stats["start"], serial["start"], rtol=1e-12)
        assert_allclose(stats["trans"], serial["trans"], rtol=1e-12)
        assert_allclose(stats["obs"], serial["obs"] + 0.25, rtol=1e-12)


def _worker_rows(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tehmm_amd.dist import ShardedEvaluator, check_same_lengths, gather_rows
        lens = [40, 1, 17, 0, 23]
        rs = np.random.RandomState(7)
        full = [rs.rand(L, 6) for L in lens]                 # "posterior rows" [T, N]
        msum = [f[:, :2].sum(axis=1) for f in full]          # the --pd column: masked sum per position

        def compute(sub_idx_tables):
            # (tables here are just index arrays of the right length: the compute looks its rows up)
            ids = [int(t[0]) if len(t) else 3 for t in sub_idx_tables]
            return {"posteriors": [full[i] for i in ids], "posterior_masksum": [msum[i] for i in ids],
                    "viterbi_logprob": [float(i) for i in ids]}
        tables = [np.full(L, i, dtype=np.int64) for i, L in enumerate(lens)]
        mine, res = ShardedEvaluator(compute).run(tables)
        # an empty shard: rank 1 of 2 holds nothing -> it learns the row width from rank 0
        only0 = gather_rows([0, 2] if rank == 0 else [], [full[0], full[2]] if rank == 0 else [], lens)
        # different lists on the ranks are refused on EVERY rank
        try:
            check_same_lengths(lens if rank == 0 else lens[:-1])
            refused = False
        except ValueError:
            refused = True
        # a bad block on ONE rank raises on both (no rank left in a collective)
        try:
            gather_rows([1], [np.zeros((2, 6))] if rank == 0 else [np.zeros((1, 6))], lens)
            together = False
        except ValueError:
            together = True
        q.put((rank, res, only0, refused, together))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_posterior_gather_and_collective_checks():
    """north_star's "gather of per-interval posteriors": masked sums (8 B per row) and full rows through
    ShardedEvaluator; list-consistency check; validation errors raise on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rows, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lens = [40, 1, 17, 0, 23]
    rs = np.random.RandomState(7)
    full = [rs.rand(L, 6) for L in lens]
    for rank, res, only0, refused, together in got:
        assert refused and together
        for i, L in enumerate(lens):
            assert_array_equal(res["posteriors"][i], full[i])
            assert_array_equal(res["posterior_masksum"][i], full[i][:, :2].sum(axis=1))
        assert_array_equal(only0[0], full[0])
        assert_array_equal(only0[2], full[2])
        assert only0[1] is None and only0[4] is None
