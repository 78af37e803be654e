#!/usr/bin/env python3
"""Headline benchmark: genome positions/sec for Viterbi + posterior (teHmmEval's hot path),
35 states x 10 tracks, on N MI355X (one process per GPU, intervals sharded, no data-path
collective: "weak" scaling, every rank evaluates its own 100 Mb shard).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one tehmm_eval_batch(VITERBI | POSTERIOR) over the rank's whole batch of intervals,
observations already resident in HBM, results left in HBM (paths int64 + posteriors f64).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_STATES = 35
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def gen_obs_torch(model, lens, seed, device):
    """Synthetic observations on the GPU (torch is plumbing here): state path made of geometric
    runs (sticky chain like the model's diagonal), symbols drawn from each state's per-track
    distribution by inverse-CDF.  Returns a uint8 [total, K] CUDA tensor."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    total = int(np.sum(lens))
    N, K = model.n_states, model.n_tracks
    stay = float(np.mean(np.diag(model.transmat)))
    n_runs = int(total * (1.0 - stay) * 1.3) + 1024
    u = torch.rand(n_runs, generator=g, device=device, dtype=torch.float64)
    run_len = (torch.log1p(-u) / np.log(stay)).floor().to(torch.int64) + 1
    run_state = torch.randint(0, N, (n_runs,), generator=g, device=device)
    states = torch.repeat_interleave(run_state, run_len)[:total]
    if states.numel() < total:
        states = torch.cat([states, states.new_zeros(total - states.numel())])
    del u, run_len, run_state
    obs = torch.empty((total, K), dtype=torch.uint8, device=device)
    for k, sk in enumerate(model.symbols_per_track):
        cdf = np.cumsum(model.probs[k, :, 1:1 + sk], axis=1)
        cdf[:, -1] = 1.0
        flat = torch.tensor((cdf + np.arange(N)[:, None]).ravel(), device=device, dtype=torch.float64)
        step = 1 << 24
        for a in range(0, total, step):
            st = states[a:a + step]
            uu = torch.rand(st.numel(), generator=g, device=device, dtype=torch.float64) * 0.999999
            idx = torch.searchsorted(flat, uu + st.to(torch.float64), right=True) - st * sk
            obs[a:a + step, k] = (idx.clamp_(0, sk - 1) + 1).to(torch.uint8)
    return obs


def cpu_baseline(model, n_threads, per_interval, seed=123):
    """Times the CPU oracle (a port of the reference's Cython loops, oracle/tehmm_oracle.c) on a
    bounded sample of the same workload: the teHmmEval flow (score_samples + decode) per interval,
    one interval per worker thread at a time."""
    from oracle import oracle
    from tehmm_amd import synth
    n_iv = 4 * n_threads
    lens = np.full(n_iv, per_interval, dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=seed)
    t0 = time.perf_counter()
    oracle.eval_batch(obs, offs, model.log_probs, model.log_startprob, model.log_transmat,
                      1.0, None, want_post=True, n_threads=n_threads)
    dt = time.perf_counter() - t0
    return {"value": float(offs[-1]) / dt, "unit": "positions/s", "cores": int(n_threads),
            "kind": "port",
            "sample": "%d intervals x %d positions, Viterbi + posterior, %d threads, %.1f s"
                      % (n_iv, per_interval, n_threads, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mb", type=float, default=100.0, help="Mb of genome per GPU")
    ap.add_argument("--min-kb", type=int, default=200, help="shortest interval (kb)")
    ap.add_argument("--max-kb", type=int, default=2000, help="longest interval (kb)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the untimed extra measurement (BASELINE configs[1]: one 10 Mb interval)")
    ap.add_argument("--cpu-sample-kb", type=int, default=250,
                    help="positions per CPU-baseline interval (kb); 4 intervals per thread")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    from tehmm_amd import _lib, synth
    from tehmm_amd.engine import HipBatch, HipModel

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    _lib.check(_lib.load().tehmm_set_device(local_rank), "tehmm_set_device")
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ      # torchrun (even with one rank) -> RCCL path
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    model = synth.make_model(N_STATES, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=0)
    total = int(args.mb * 1e6)
    lens = synth.interval_lengths(total, args.min_kb * 1000, args.max_kb * 1000, seed=1000 + rank)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = gen_obs_torch(model, lens, seed=17 + rank, device=device)
    torch.cuda.synchronize()

    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs,
                  symbols_per_track=model.symbols_per_track)
    hb = HipBatch(obs.data_ptr(), offs, device_ptrs=True, K=model.n_tracks)
    del obs
    torch.cuda.empty_cache()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        hm.eval(hb, viterbi=True, posterior=True)
    kt = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hm.eval(hb, viterbi=True, posterior=True)
        for name, ms in hb.timing().items():      # HIP events on the library's own streams
            if not name.startswith("count:"):
                kt.setdefault(name, []).append(ms)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        K, N = model.n_tracks, model.n_states
        pos_per_step = float(total) * world
        value = pos_per_step * args.steps / dt
        kavg = {k: float(np.mean(v)) for k, v in kt.items()}
        dom = max(kavg, key=kavg.get)
        # algorithmic bytes per position of each stage (API-level I/O only, the
        # intermediate lattices / emission rows / traceback tables do not count):
        #   viterbi*: obs in (K) [+ int64 path out (8) where the stage produces it];  forward*: obs in
        #   (K);  posterior_combine / backward_posterior: posterior row out (8N);  traceback: path out (8)
        alg = {"viterbi": K + 8, "viterbi_speculate": K, "traceback": 8, "emission_rows": K,
               "forward_backward_speculate": K, "forward_backward": K, "forward": K,
               "posterior_combine": 8 * N, "backward_posterior": K + 8 * N}
        # fp64 VALU operations per position of each stage (the resource that actually binds):
        #   max-plus pass: N*N (add + max);  forward or backward pass: N*N fma = 2 N*N flop
        flop = {"viterbi_speculate": 2 * 2 * N * N, "viterbi": 2 * N * N, "forward_backward_speculate": 2 * 2 * N * N,
                "forward_backward": 2 * 2 * N * N, "forward": 2 * N * N, "backward_posterior": 2 * N * N}
        achieved = alg[dom] * float(total) / (kavg[dom] * 1e-3) / 1e9
        traffic = None
        stage_hbm = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):     # HBM bytes per position measured with rocprofv3 --pmc (see file)
            tj = json.load(open(tpath)).get("hbm_bytes_per_position", {})
            per_pos = tj.get(dom)
            if per_pos is not None:
                traffic = per_pos * float(total)
            # measured HBM bytes (PMC) over the live stage durations: which stages sit on the HBM roof
            stage_hbm = {k: tj[k] * float(total) / (kavg[k] * 1e-3) / 1e9 for k in kavg if k in tj}
        out = {
            "metric": "genome positions/sec (Viterbi+posterior), 35 states x 10 tracks",
            "value": value, "unit": "positions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "teHmmEval Viterbi+posterior, 35 states, 10 tracks (8 multinomial "
                                   "+ 2 gaussian/250 bins), %.0f Mb per GPU in %d intervals of "
                                   "%d-%d kb (config-3 geometry), obs resident in HBM"
                                   % (args.mb, len(lens), args.min_kb, args.max_kb),
                       "positions_per_gpu": total, "intervals_per_gpu": int(len(lens)),
                       "parallelism": "intervals sharded over %d GPU(s), no collective" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "alg_bytes_per_position": alg[dom],
                         "whole_path_GBps": (K + 8 + 8 * N) * value / world / 1e9,
                         "stage_hbm_traffic_GBps": stage_hbm,
                         # the stage that sits highest on the HBM roof (measured bytes / live duration)
                         "busiest_hbm_stage": (lambda k: {"kernel": k, "achieved": stage_hbm[k], "unit": "GB/s",
                                                          "frac": stage_hbm[k] / HBM_PEAK_GBPS})(
                             max(stage_hbm, key=stage_hbm.get)) if stage_hbm else None,
                         # what binds instead of HBM: fp64 vector issue (78.6 TFLOP/s peak on MI355X)
                         "valu_f64": {"achieved": flop.get(dom, 0) * float(total) / (kavg[dom] * 1e-3) / 1e12,
                                      "peak": 78.6, "unit": "TFLOP/s",
                                      "frac": flop.get(dom, 0) * float(total) / (kavg[dom] * 1e-3) / 1e12 / 78.6}},
            "kernel_ms": kavg,
        }
        if world == 1 and not args.no_extra:
            # BASELINE.json configs[1] for orientation (NOT the headline): the same model on ONE
            # 10 Mb interval -- a single dependent chain, i.e. pure per-step latency.
            hb.close()
            one = np.asarray([10_000_000], dtype=np.int64)
            obs1 = gen_obs_torch(model, one, seed=99, device=device)
            hb1 = HipBatch(obs1.data_ptr(), np.asarray([0, one[0]], dtype=np.int64), device_ptrs=True,
                           K=model.n_tracks)
            hm.eval(hb1, viterbi=True, posterior=True)      # warm-up (allocates the result buffers)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            hm.eval(hb1, viterbi=True, posterior=True)
            d1 = time.perf_counter() - t1
            out["extra"] = {"config2_single_10Mb_interval": {
                "value": float(one[0]) / d1, "unit": "positions/s", "ms": d1 * 1e3,
                "kernel_ms": hb1.timing()}}
            hb1.close()
            del obs1
        if world == 1 and not args.no_cpu_baseline:
            nthr = max(1, min(16, os.cpu_count() or 1))
            out["cpu_baseline"] = cpu_baseline(model, nthr, args.cpu_sample_kb * 1000)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
