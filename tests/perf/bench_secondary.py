#!/usr/bin/env python3
"""Throughput of the secondary kernels of the path (K3 strict depth, K4 simplex, K5 L1, chunked K1+K2,
external targets), each next to the CPU oracle timed on a stated sample.  One JSON object per line.
Run on the GPU box: python tests/perf/bench_secondary.py > gpurun_out/secondary.jsonl
"""
import json
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import oracle  # noqa: E402
from statdepth_amd import engine  # noqa: E402


def tm(f, reps=3):
    f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def cpu(f):
    t = time.perf_counter()
    r = f()
    return time.perf_counter() - t, r


def emit(**kw):
    print(json.dumps(kw), flush=True)


def main():
    oracle.build()
    oracle.set_num_threads(min(16, os.cpu_count() or 1))
    cores = oracle.num_threads()
    rng = np.random.default_rng(0)

    # K3 strict band depth, config-2-shaped but n = 2000 (pairs grow as n^2 per target)
    n, T = 2000, 1000
    X = np.round(np.sort(rng.normal(size=n))[None, :] * 3 + rng.normal(size=(T, n)) * 0.3, 1)
    Xd = engine.to_device_matrix(X)
    t = tm(lambda: engine.bd_strict_counts(Xd))
    ct, want = cpu(lambda: oracle.bd_strict_counts(X, np.arange(64)))
    assert (engine.bd_strict_counts(Xd, np.arange(64))[:, 0] == want).all()
    emit(kernel="K3 bd_strict (J=2)", workload=f"{n} curves x {T} timepoints, all targets", seconds=t,
         unit="pair-tests/s", value=n * math.comb(n - 1, 2) / t,
         cpu_value=64 * math.comb(n - 1, 2) / ct, cpu_cores=cores, cpu_sample="64 targets")

    # K4 simplex: exhaustive point cloud, d = 2 and 3
    for (n, d) in ((200, 2), (60, 3)):
        P = rng.normal(size=(n, d))
        t = tm(lambda: engine.pointcloud_simplex_counts(P))
        ct, want = cpu(lambda: oracle.pointcloud_simplex_counts(P, np.arange(8)))
        assert (engine.pointcloud_simplex_counts(P, np.arange(8)) == want).all()
        emit(kernel="K4 simplex (pointcloud, exhaustive)", workload=f"n={n}, d={d}", seconds=t,
             unit="simplex-tests/s", value=n * math.comb(n - 1, d + 1) / t,
             cpu_value=8 * math.comb(n - 1, d + 1) / ct, cpu_cores=cores, cpu_sample="8 targets")
    # K4 sampled estimators (configs 4 / 5 shapes, reduced)
    P = rng.normal(size=(100000, 3))
    t = tm(lambda: engine.pointcloud_simplex_counts(P, samples=256, seed=1), 1)
    ct, want = cpu(lambda: oracle.simplex_sampled(P, np.arange(2000), samples=256, seed=1))
    assert (engine.pointcloud_simplex_counts(P, np.arange(2000), samples=256, seed=1) == want).all()
    emit(kernel="K4 simplex (pointcloud, sampled)", workload="n=1e5 points in R^3, 256 tetrahedra per point",
         seconds=t, unit="simplex-tests/s", value=1e5 * 256 / t, cpu_value=2000 * 256 / ct, cpu_cores=cores,
         cpu_sample="2000 targets")
    C = rng.normal(size=(500, 50, 8)).cumsum(axis=1)
    t = tm(lambda: engine.multi_simplex_counts(C, samples=256, seed=1), 1)
    ct, want = cpu(lambda: oracle.simplex_sampled(C, np.arange(16), samples=256, seed=1))
    assert (engine.multi_simplex_counts(C, np.arange(16), samples=256, seed=1) == want).all()
    emit(kernel="K4 simplex (multivariate d=8, sampled)", workload="500 curves x 50 timepoints x 8 features, 256 subsets",
         seconds=t, unit="simplex-tests/s", value=500 * 256 * 50 / t, cpu_value=16 * 256 * 50 / ct, cpu_cores=cores,
         cpu_sample="16 targets")

    # K5 L1 depth
    P = rng.normal(size=(100000, 3))
    t = tm(lambda: engine.l1_depth(P), 1)
    ct, want = cpu(lambda: oracle.l1_depth(P, np.arange(2000)))
    assert np.max(np.abs(engine.l1_depth(P, np.arange(2000)) - want)) <= 1e-12
    emit(kernel="K5 l1_depth", workload="n=1e5 points in R^3", seconds=t, unit="point-pairs/s", value=1e10 / t,
         cpu_value=2000 * 1e5 / ct, cpu_cores=cores, cpu_sample="2000 targets")

    # large-n K1+K2 (value buckets): config 3 on one GPU
    n, T = 100000, 256
    Xd = torch.randn(T, n, dtype=torch.float64, device="cuda").cumsum(0)
    t = tm(lambda: engine.mbd_counts(Xd, None, 2, algo="rank", return_tensor=True), 3)
    emit(kernel="K1+K2 large-n rank (value buckets)", workload=f"{n} curves x {T} timepoints (config 3), 1 GPU",
         seconds=t, unit="curve-pairs/s", value=n * (n - 1) / t,
         algorithmic_bytes=8.0 * T * 2 * n + 8.0 * n, roofline_frac=(8.0 * T * 2 * n + 8.0 * n) / t / 8e12)
    # external targets (homogeneity P3 shape)
    F = rng.normal(size=(1000, 10000)).cumsum(axis=0)
    G = rng.normal(size=(1000, 2000)).cumsum(axis=0)
    t = tm(lambda: engine.mbd_external_counts(F, G), 1)
    emit(kernel="K1+K2 external targets (bucket look-up)", workload="2000 external curves vs 10000 curves x 1000 timepoints (incl. H2D)",
         seconds=t, unit="curve-pairs/s", value=2000 * 10000 / t)


if __name__ == "__main__":
    main()
