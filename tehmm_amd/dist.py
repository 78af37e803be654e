"""Multi-GPU layer: one process per GPU, intervals (TrackTables) sharded across ranks.

The reference's only parallelism is independent per-chromosome worker processes whose outputs are
concatenated (bin/teHmmEval.py:312-383), and a serial "+=" of sufficient statistics over sequences
(basehmm.py:507-522, hmm.py:545-574).  Here:
  * evaluation (Viterbi / posterior): EVERY rank passes the same global interval list (checked: check_same_lengths),
    the list is LPT-sharded by length here, every rank evaluates its own shard on its own GPU -- no collective on
    the data path; an optional gather brings the per-interval log-probabilities, the uint8 paths and the
    per-position posterior results (masked sums or full rows: gather_rows) to every rank;
  * training: each rank's E-step statistics are packed into ONE fp64 buffer
    [nobs, logprob, start[N], trans[N,N], obs[K,N,S]] and summed with a single all-reduce per EM
    iteration (RCCL over xGMI with the "nccl" backend; "gloo" on CPU for tests).  The summation order
    differs from the serial reference, hence <= 1e-6 relative agreement instead of bit equality.
torch.distributed is plumbing only.
"""
import numpy as np


def lpt_shard(lengths, world_size):
    """Greedy longest-processing-time assignment of intervals to ranks (cost = length).
    Returns a list of index arrays, one per rank, each sorted ascending (stable output order)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    buckets = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(load))
        buckets[r].append(int(i))
        load[r] += lengths[i]
    return [np.asarray(sorted(b), dtype=np.int64) for b in buckets]


def _dist():
    import torch.distributed as dist
    return dist


def world_rank():
    """(world size, rank) of the default process group; (1, 0) without one."""
    try:
        dist = _dist()
    except ImportError:
        return 1, 0
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(), dist.get_rank()


def _device_for_backend():
    import torch
    dist = _dist()
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def pack_stats(stats, logprob):
    """One flat fp64 buffer for the all-reduce (~0.7 MB at N=35, K=10)."""
    return np.concatenate([[float(stats["nobs"]), float(logprob)], stats["start"].ravel(),
                           stats["trans"].ravel(), stats["obs"].ravel()]).astype(np.float64)


def unpack_stats(buf, like):
    n = like["start"].size
    nn = like["trans"].size
    out = dict(like)
    out["nobs"] = int(round(buf[0]))
    logprob = float(buf[1])
    out["start"] = buf[2:2 + n].reshape(like["start"].shape).copy()
    out["trans"] = buf[2 + n:2 + n + nn].reshape(like["trans"].shape).copy()
    out["obs"] = buf[2 + n + nn:].reshape(like["obs"].shape).copy()
    return out, logprob


def allreduce_stats(stats, logprob):
    """Sum the E-step sufficient statistics and the log-likelihood over all ranks."""
    import torch
    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return stats, logprob
    t = torch.from_numpy(pack_stats(stats, logprob)).to(_device_for_backend())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return unpack_stats(t.cpu().numpy(), stats)


def allreduce_device_stats(stats):
    """The one collective of an EM iteration, in place on the device buffer of the raw statistics
    (engine.DeviceStats).  nccl backend: RCCL all-reduce of the buffer's torch tensor over xGMI;
    any other backend (gloo: the two-rank tests) cannot see device memory, so the buffer is staged
    through the host (0.2 MB at N = 35, K = 10)."""
    import torch
    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    if stats.tensor is not None and dist.get_backend() == "nccl":
        dist.all_reduce(stats.tensor, op=dist.ReduceOp.SUM)
        return
    t = torch.from_numpy(stats.to_host())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    stats.from_host(t.numpy())


def allgather_values(local_idx, local_values, n_total):
    """gather_interval_scalars under a name the EM driver reads better with."""
    return gather_interval_scalars(local_idx, local_values, n_total)


def all_agree(flag):
    """True iff `flag` is true on EVERY rank (one all-reduce; trivially `flag` without a process group)."""
    import torch
    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=_device_for_backend())
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def check_same_lengths(lengths, what="table list"):
    """Every rank must pass the SAME global interval list to the sharding entry points (they cut it with lpt_shard
    themselves); a caller that passes per-rank shards would silently train on 1/world of each.  One all-reduce (MIN and
    MAX of the count and of a hash of the length vector, as MAX of v and of -v); raises ValueError on EVERY rank."""
    import torch
    dist = _dist()
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    lengths = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64))
    import zlib
    h = zlib.crc32(lengths.tobytes()) & 0x7fffffff
    v = np.asarray([len(lengths), h, int(lengths.sum()) & 0x7fffffffffff], dtype=np.int64)
    t = torch.from_numpy(np.concatenate([v, -v])).to(_device_for_backend())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t = t.cpu().numpy()
    if not np.array_equal(t[:3], -t[3:]):
        raise ValueError("tehmm_amd.dist: the ranks passed different %ss (count / lengths differ); every rank passes the "
                         "full global list and the library shards it" % what)


def _raise_together(msg):
    """Collective-safe validation: `msg` is this rank's error (or None); if ANY rank has one, every rank raises, so
    that nobody is left waiting in the next collective."""
    ok = all_agree(msg is None)
    if not ok:
        raise ValueError(msg if msg is not None else "tehmm_amd.dist: another rank failed its argument check")


def gather_interval_scalars(local_idx, local_values, n_total, lengths=None):
    """All ranks get the full per-interval array (e.g. Viterbi log-probs) of a sharded batch: one
    all-gather of (index, value) pairs padded to the largest shard."""
    import torch
    dist = _dist()
    full = np.zeros(n_total, dtype=np.float64)
    local_idx = np.asarray(local_idx, dtype=np.int64)
    full[local_idx] = np.asarray(local_values, dtype=np.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        cap = (n_total + world - 1) // world + 1
        cnt = torch.tensor([len(local_idx)], dtype=torch.int64, device=_device_for_backend())
        cmax = cnt.clone()
        dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
        cap = max(cap, int(cmax.item()))
        buf = np.full((cap, 2), -1.0, dtype=np.float64)
        buf[:len(local_idx), 0] = local_idx
        buf[:len(local_idx), 1] = np.asarray(local_values, dtype=np.float64)
        send = torch.from_numpy(buf).to(_device_for_backend())
        recv = torch.empty((world * cap, 2), dtype=torch.float64, device=send.device)
        dist.all_gather_into_tensor(recv, send)
        recv = recv.cpu().numpy()
        ok = recv[:, 0] >= 0
        full[recv[ok, 0].astype(np.int64)] = recv[ok, 1]
    return full


def gather_paths(local_idx, local_paths, lengths):
    """Gather of the variable-length per-interval Viterbi paths to every rank: ONE all-gather of the
    ranks' concatenated paths as uint8 (states < 256: 1 byte per position on the wire instead of the
    reference's int64), padded to the largest shard; returned in global interval order as int64."""
    import torch
    dist = _dist()
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        out = [None] * n
        for i, p in zip(local_idx, local_paths):
            out[int(i)] = np.asarray(p, dtype=np.int64)
        return out
    world = dist.get_world_size()
    dev = _device_for_backend()
    local_idx = np.asarray(local_idx, dtype=np.int64)
    err = None
    cat = np.zeros(0, dtype=np.int64)
    if len(local_idx) != len(local_paths):
        err = "gather_paths: %d indices for %d paths" % (len(local_idx), len(local_paths))
    else:
        for i, p in zip(local_idx, local_paths):
            if len(p) != int(lengths[i]):
                err = ("gather_paths: path of interval %d has %d states, the interval %d rows"
                       % (int(i), len(p), int(lengths[i])))
                break
    if err is None and len(local_paths):
        cat = np.concatenate([np.asarray(p) for p in local_paths]) if len(local_paths) > 1 else np.asarray(local_paths[0])
        if cat.size and (int(cat.max()) >= 256 or int(cat.min()) < 0):
            err = "gather_paths: states must fit a byte"
    _raise_together(err)              # (every rank raises or none: no rank is left alone in the collectives below)
    # which intervals every rank holds, in the order it sends them (any sharding, not only lpt_shard's)
    cnt = torch.tensor([len(local_idx)], dtype=torch.int64, device=dev)
    cmax = cnt.clone()
    dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
    icap = max(int(cmax.item()), 1)
    ibuf = np.full(icap, -1, dtype=np.int64)
    ibuf[:len(local_idx)] = local_idx
    isend = torch.from_numpy(ibuf).to(dev)
    irecv = torch.empty(world * icap, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(irecv, isend)
    held = irecv.cpu().numpy().reshape(world, icap)
    sizes = [int(lengths[h[h >= 0]].sum()) for h in held]
    cap = max(max(sizes), 1)
    mine = np.zeros(cap, dtype=np.uint8)
    mine[:cat.size] = cat.astype(np.uint8)
    send = torch.from_numpy(mine).to(dev)
    recv = torch.empty(world * cap, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send)
    recv = recv.cpu().numpy().reshape(world, cap)
    out = [None] * n
    for r in range(world):
        o = 0
        for i in held[r][held[r] >= 0]:
            if out[int(i)] is not None:     # (every rank sees the same `held`: all of them raise)
                raise ValueError("gather_paths: interval %d is held by more than one rank" % int(i))
            out[int(i)] = recv[r, o:o + int(lengths[i])].astype(np.int64)
            o += int(lengths[i])
    return out


def gather_rows(local_idx, local_rows, lengths, width=None):
    """Gather of per-POSITION float64 results of the sharded intervals to every rank, in global interval order:
    `local_rows[j]` is an array [lengths[local_idx[j]]] or [lengths[..], width] -- the masked posterior sums teHmmEval
    writes with --pd (8 bytes per position: the natural payload, tehmm_batch_posterior_masksum) or the full posterior
    rows [T, N] (bin/teHmmEval.py:270-272; its per-chromosome workers' outputs are concatenated, :312-383).  ONE
    all-gather of the index lists and ONE of the concatenated rows padded to the largest shard."""
    import torch
    dist = _dist()
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    local_idx = np.asarray(local_idx, dtype=np.int64)
    rows = [np.asarray(r, dtype=np.float64) for r in local_rows]
    if width is None and rows:
        width = 0 if rows[0].ndim == 1 else int(rows[0].shape[1])
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        out = [None] * n
        for i, r in zip(local_idx, rows):
            out[int(i)] = r
        return out
    world = dist.get_world_size()
    dev = _device_for_backend()
    # the row width must be the same everywhere; a rank with an empty shard learns it from the others
    NONE = -(1 << 40)
    wt = torch.tensor([width, -width] if width is not None else [NONE, NONE], dtype=torch.int64, device=dev)
    dist.all_reduce(wt, op=dist.ReduceOp.MAX)
    wmax, wmin = int(wt[0].item()), -int(wt[1].item())
    err = None
    if width is None:
        width = max(wmax, 0)
    elif wmax != width or wmin != width:
        err = "gather_rows: the ranks hold row blocks of different widths (%d here)" % width
    if err is None and len(local_idx) != len(rows):
        err = "gather_rows: %d indices for %d row blocks" % (len(local_idx), len(rows))
    if err is None:
        for i, r in zip(local_idx, rows):
            if r.shape[0] != int(lengths[i]) or (width and (r.ndim != 2 or r.shape[1] != width)) or (not width and r.ndim != 1):
                err = "gather_rows: block of interval %d has shape %s, expected %d rows x %d" % (
                    int(i), r.shape, int(lengths[i]), max(width, 1))
                break
    _raise_together(err)
    cnt = torch.tensor([len(local_idx)], dtype=torch.int64, device=dev)
    cmax = cnt.clone()
    dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
    icap = max(int(cmax.item()), 1)
    ibuf = np.full(icap, -1, dtype=np.int64)
    ibuf[:len(local_idx)] = local_idx
    irecv = torch.empty(world * icap, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(irecv, torch.from_numpy(ibuf).to(dev))
    held = irecv.cpu().numpy().reshape(world, icap)
    per = max(width, 1)
    sizes = [int(lengths[h[h >= 0]].sum()) * per for h in held]
    cap = max(max(sizes), 1)
    mine = np.zeros(cap, dtype=np.float64)
    if rows:
        cat = np.concatenate([r.reshape(-1) for r in rows])
        mine[:cat.size] = cat
    recv = torch.empty(world * cap, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(recv, torch.from_numpy(mine).to(dev))
    recv = recv.cpu().numpy().reshape(world, cap)
    out = [None] * n
    for r in range(world):
        o = 0
        for i in held[r][held[r] >= 0]:
            if out[int(i)] is not None:
                raise ValueError("gather_rows: interval %d is held by more than one rank" % int(i))
            m = int(lengths[i]) * per
            blk = recv[r, o:o + m].copy()
            out[int(i)] = blk.reshape(int(lengths[i]), width) if width else blk
            o += m
    return out


class ShardedEvaluator(object):
    """Evaluates a list of tables with the intervals sharded over the ranks of the default process
    group.  `compute(tables_subset) -> dict` is the per-rank work (on a GPU box:
    MultitrackHmm._eval_tables; in CPU tests: anything with the same result keys)."""

    def __init__(self, compute):
        self.compute = compute

    def run(self, tables, gather=True):
        dist = _dist()
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        lengths = [len(t) for t in tables]
        check_same_lengths(lengths, "interval list")
        mine = lpt_shard(lengths, world)[rank]
        res = self.compute([tables[i] for i in mine])
        if not gather or world == 1:
            return mine, res
        out = {}
        # per-position posterior results (north_star: "gather of per-interval posteriors and Viterbi paths"): the masked
        # sums of the --pd column (8 B per row) and / or the full rows, whatever `compute` returned
        if res.get("posterior_masksum") is not None:
            out["posterior_masksum"] = gather_rows(mine, res["posterior_masksum"], lengths, width=0)
        if res.get("posteriors") is not None:
            out["posteriors"] = gather_rows(mine, res["posteriors"], lengths)
        if res.get("viterbi_logprob") is not None:
            out["viterbi_logprob"] = gather_interval_scalars(mine, res["viterbi_logprob"], len(tables))
        if res.get("forward_logprob") is not None:
            out["forward_logprob"] = gather_interval_scalars(mine, res["forward_logprob"], len(tables))
        if res.get("paths") is not None:
            out["paths"] = gather_paths(mine, res["paths"], lengths)
        return mine, out


def sharded_estep(tables, estep_fn, empty_stats):
    """One EM E-step with the sequences sharded over ranks and ONE all-reduce of the packed
    statistics.  `estep_fn(tables_subset, stats) -> logprob` accumulates into `stats`."""
    dist = _dist()
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    check_same_lengths([len(t) for t in tables], "training table list")
    mine = lpt_shard([len(t) for t in tables], world)[rank]
    stats = empty_stats()
    if rank != 0:
        # the caller's initial values (fudge etc.) must be counted once, not world_size times
        base = empty_stats()
        for k in ("start", "trans", "obs"):
            stats[k] = stats[k] - base[k]
    logprob = estep_fn([tables[i] for i in mine], stats)
    return allreduce_stats(stats, logprob)
