#!/usr/bin/env python3
"""bench.py -- curve-pairs/s of modified band depth (J=2) on MI355X.

Workload (BASELINE.json configs[1] at N=1): every rank owns n_loc = 10 000 synthetic
random-walk curves x T = 1 000 timepoints (fp64, time-major); the data set is the
union of all ranks' curves (n = N * n_loc), each rank computes the exact MBD
containment totals of its own curves against the full set.  One step = one pass of
the hot path: (N > 1: RCCL all-gather of the curve blocks) + sd_mbd_counts on the
resident matrix.  value = ordered (target, other) curve pairs evaluated over all T
timepoints per second, whole job.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md)


def pmc_traffic(kernel, n, T, J):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc summary (profiles/), collected in
    separate FETCH_SIZE / WRITE_SIZE passes and corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on
    gfx950, calibrated on a kernel of known traffic).  Only valid for the workload it was measured on."""
    if (n, T, J) != (10000, 1000, 2):
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        ks = json.load(f)["kernels"]
    for name, d in ks.items():
        if kernel in name and "hbm_bytes_per_launch_corrected" in d:
            return d["hbm_bytes_per_launch_corrected"]
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n-loc", type=int, default=10000)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--J", type=int, default=2)
    ap.add_argument("--algo", default="auto", choices=["auto", "pairwise", "rank"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-targets", type=int, default=10000)
    ap.add_argument("--variant", default="walk", choices=["walk", "ties"],
                    help="walk: continuous random walks (headline); ties: the same rounded to 1 decimal with 1 %% "
                         "duplicated curves (SURVEY.md 8(d) config 2 variant)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the torch.distributed path even at world size 1 (exercises RCCL on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from statdepth_amd import _native, engine
    from statdepth_amd._native import ALGOS, check

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    N = world
    assert args.gpus == N, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = _native.require_device()

    T, n_loc, J = args.T, args.n_loc, args.J
    n = n_loc * N
    # synthetic curves: random walks (SURVEY.md 8(d) config 2 recipe), one block per rank
    rng = np.random.default_rng(1234 + rank)
    X_host = rng.normal(size=(T, n_loc)).cumsum(axis=0)
    if args.variant == "ties":
        X_host = np.round(X_host, 1)
        dup = rng.choice(n_loc, size=max(1, n_loc // 100), replace=False)
        X_host[:, dup] = X_host[:, (dup + 1) % n_loc]
    X_loc = torch.from_numpy(X_host).to(dev)                       # [T, n_loc] time-major, resident in HBM
    X_all = X_loc
    out = torch.empty((n_loc, J - 1), dtype=torch.int64, device=dev)
    algo = ALGOS[args.algo]
    wsb = 0 if use_dist else lib.sd_mbd_workspace_bytes(T, n, n, 1, n_loc, J, algo)
    ws = torch.empty(max(int(wsb), 8), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)

    from statdepth_amd.distributed import sharded_mbd_counts
    sizes = [n_loc] * N

    def step():
        if use_dist:
            # product multi-GPU path (statdepth_amd/distributed.py): time-sharded for the rank kernels
            # (RCCL all-to-all of curve blocks -> per-curve partial totals -> reduce-scatter), target-sharded
            # (all-gather) for --algo pairwise
            out.copy_(sharded_mbd_counts(X_loc, J=J, algo=args.algo, sizes=sizes))
        else:
            check(lib.sd_mbd_counts_range(X_all.data_ptr(), T, n, n, 1, rank * n_loc, n_loc, J, algo,
                                          out.data_ptr(), ws.data_ptr(), wsb, stream.cuda_stream))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    barrier()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) / args.steps          # device time per step on the launch stream
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    pairs_per_step = float(n_loc) * N * (n - 1)          # ordered (target, other) pairs, whole job
    value = pairs_per_step / (dt / args.steps)

    result_sum = int(out.sum().item())
    if rank == 0:
        used = "rank" if (args.algo == "rank" or (args.algo == "auto" and J <= 3)) else "pairwise"
        # dominant kernel of the step: the bucket rank kernel (n <= 16384, J <= 3), the value-bucket sort of the large-n
        # route, or the pairwise kernel
        kern = "mbd_pairwise_kernel" if used == "pairwise" else ("rank_bucket_kernel" if n <= 16384 else "bucket_rank_kernel")
        bytes_alg = 8.0 * T * (n + n_loc) + 8.0 * n_loc * (J - 1)   # per GPU per call
        achieved = bytes_alg / (dev_ms * 1e-3)
        line = {
            "metric": "curve-pairs/sec (MBD)", "value": value, "unit": "curve-pairs/s",
            "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"MBD J={J}: {n} curves x {T} timepoints (n_loc={n_loc} targets per GPU), fp64 "
                                   f"random walks{' rounded to 0.1 + 1% duplicates' if args.variant == 'ties' else ''}, time-major", "algorithm": used,
                       "parallelism": (f"single GPU" if N == 1 else
                                       (f"targets sharded x{N}, RCCL all-gather of curve blocks" if used == "pairwise" else
                                        f"curves owned x{N}, timepoints sharded for the sort: RCCL all-to-all + reduce-scatter"))},
            "pair_timepoints_per_s": value * T,
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": pmc_traffic(kern, n, T, J) if N == 1 else None,
                         "kernel": kern,
                         "kernel_ms": dev_ms, "algorithmic_bytes": bytes_alg},
            "checksum": result_sum,
        }
        if not args.no_cpu_baseline and N == 1:
            import oracle
            oracle.build()
            if "OMP_NUM_THREADS" not in os.environ:
                oracle.set_num_threads(min(16, os.cpu_count() or 1))     # the box's CPU share for one GPU
            m_cpu = min(args.cpu_targets, n_loc)
            tg = np.arange(m_cpu, dtype=np.int64)
            t1 = time.perf_counter()
            want = oracle.mbd_counts(X_host, tg, J)
            cpu_dt = time.perf_counter() - t1
            got = out[:m_cpu].cpu().numpy()
            assert (got == want).all(), "HIP counts differ from the CPU oracle on the baseline sample"
            line["cpu_baseline"] = {
                "value": m_cpu * (n - 1) / cpu_dt, "unit": "curve-pairs/s", "cores": oracle.num_threads(),
                "kind": "port",
                "sample": f"oracle_mbd_counts (C, OpenMP) on the first {m_cpu} of {n_loc} targets x all {n} curves x "
                          f"{T} timepoints, {cpu_dt:.2f} s; work is linear in #targets",
            }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
