"""Multi-GPU layer: one process per GPU, intervals (TrackTables) sharded across ranks.

The reference's only parallelism is independent per-chromosome worker processes whose outputs are
concatenated (bin/teHmmEval.py:312-383), and a serial "+=" of sufficient statistics over sequences
(basehmm.py:507-522, hmm.py:545-574).  Here:
  * evaluation (Viterbi / posterior): intervals are LPT-sharded by length, every rank evaluates its
    own shard on its own GPU -- no collective on the data path; an optional gather brings the
    per-interval log-probabilities / paths to every rank;
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
    if len(local_idx) != len(local_paths):
        raise ValueError("gather_paths: %d indices for %d paths" % (len(local_idx), len(local_paths)))
    for i, p in zip(local_idx, local_paths):
        if len(p) != int(lengths[i]):
            raise ValueError("gather_paths: path of interval %d has %d states, the interval %d rows"
                             % (int(i), len(p), int(lengths[i])))
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
    if len(local_paths):
        cat = np.concatenate([np.asarray(p) for p in local_paths]) if len(local_paths) > 1 else np.asarray(local_paths[0])
        if cat.size and (int(cat.max()) >= 256 or int(cat.min()) < 0):
            raise ValueError("gather_paths: states must fit a byte")
        mine[:cat.size] = cat.astype(np.uint8)
    send = torch.from_numpy(mine).to(dev)
    recv = torch.empty(world * cap, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send)
    recv = recv.cpu().numpy().reshape(world, cap)
    out = [None] * n
    for r in range(world):
        o = 0
        for i in held[r][held[r] >= 0]:
            if out[int(i)] is not None:
                raise ValueError("gather_paths: interval %d is held by more than one rank" % int(i))
            out[int(i)] = recv[r, o:o + int(lengths[i])].astype(np.int64)
            o += int(lengths[i])
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
        mine = lpt_shard(lengths, world)[rank]
        res = self.compute([tables[i] for i in mine])
        if not gather or world == 1:
            return mine, res
        out = {}
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
    mine = lpt_shard([len(t) for t in tables], world)[rank]
    stats = empty_stats()
    if rank != 0:
        # the caller's initial values (fudge etc.) must be counted once, not world_size times
        base = empty_stats()
        for k in ("start", "trans", "obs"):
            stats[k] = stats[k] - base[k]
    logprob = estep_fn([tables[i] for i in mine], stats)
    return allreduce_stats(stats, logprob)
