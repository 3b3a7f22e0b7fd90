#!/usr/bin/env python3
"""bench.py -- curve-pairs/s of modified band depth (J=2) on MI355X.

Headline workload (BASELINE.json configs[1] at N=1): every rank owns n_loc = 10 000 synthetic random-walk curves x
T = 1 000 timepoints (fp64, time-major); the data set is the union of all ranks' curves (n = N * n_loc), each rank
computes the exact MBD containment totals of its own curves against the full set.  One step = one pass of the hot
path: (N > 1: the RCCL exchange of the curve blocks) + the rank kernels on the resident matrix.  value = ordered
(target, other) curve pairs evaluated over all T timepoints per second, whole job.

N = 1: the timed steps ROTATE over --rotate distinct matrices (default 4 x 80 MB > the 256 MiB Infinity Cache), so every
step streams its matrix from HBM; the same loop replayed on ONE matrix (cache-resident, what round 1 reported) is
printed beside it as `replay`.  `extras` holds the secondary workloads (config 3 on one GPU, the 10^5 x 10^3 stretch
case, the tie-heavy variant, strict depth, L1, sampled simplex), each checked against the oracle on a sample in this
run.  N > 1: `extras` holds config 3 (10^5 curves x 256 timepoints in all, strong scaling, both decompositions) and
config 5 (10^6 points in R^3 through sharded_pointcloud).

Prints ONE JSON line (rank 0).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md)
# lane-instructions/s: 256 CUs x 4 SIMD-32 x 32 lanes x 2.4 GHz -- a wave64 instruction issues over 2 cycles when its SIMD has
# several waves to pick from (MI355X_MICROARCH.md, "Each CU has 4 SIMD-32 units"; one wave alone: 4).  Rounds 1-2 priced against
# 16 lanes (4 cycles): half of this.  Measured per instruction (profiles/r03m_valu_rate.txt): two-operand integer / fp32 reach it,
# three-operand, compare, carry and fp64 instructions take 4.1 cycles -- a kernel made of those saturates at frac ~ 0.5
VALU_PEAK = 256 * 4 * 32 * 2.4e9


def issue_roofline(key, units_per_s):
    """Roofline object of a VALU-issue-bound kernel: lane-instructions per unit from the committed SQ counters
    (profiles/*issue_roofline.json: SQ_INSTS_VALU x 64 / units) x the unit rate measured in this run, against the chip's vector
    issue peak.  None when no committed counter file names the kernel."""
    for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "*issue_roofline.json")))):
        with open(path) as f:
            doc = json.load(f)
        if key in doc and "valu_lane_instructions_per_unit" in doc[key]:
            per = float(doc[key]["valu_lane_instructions_per_unit"])
            ach = per * units_per_s
            return {"bound": "valu", "achieved": ach / 1e12, "peak": VALU_PEAK / 1e12, "unit": "T lane-instructions/s",
                    "frac": ach / VALU_PEAK, "traffic": None, "kernel": doc[key].get("kernel"),
                    "lane_instructions_per_unit": per, "counter_file": os.path.relpath(path, ROOT)}
    return None


def pmc_traffic(kernels, n, T, J):
    """HBM bytes per step from the committed rocprofv3 --pmc summary (profiles/), collected in separate FETCH_SIZE /
    WRITE_SIZE passes and corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950, calibrated on a kernel of
    known traffic).  Only valid for the workload it was measured on; returns (bytes, source file) or (None, None)."""
    if (n, T, J) != (10000, 1000, 2):
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")))
    for path in reversed(files):
        with open(path) as f:
            doc = json.load(f)
        ks = doc.get("kernels", {})
        if doc.get("workload", "10000x1000") != "10000x1000":
            continue
        got = [d["hbm_bytes_per_launch_corrected"] for name, d in ks.items()
               if any(k in name for k in kernels) and "hbm_bytes_per_launch_corrected" in d]
        if len(got) >= len(kernels):
            return float(sum(got)), os.path.relpath(path, ROOT)
    return None, None


def profiled_kernel_us(kernel, n, T, J):
    """Average duration (us) of `kernel` in the newest committed rocprofv3 --stats summary of this workload
    (profiles/*bench_kernel_stats.csv), with its file name; (None, None) for any other workload."""
    if (n, T, J) != (10000, 1000, 2):
        return None, None
    import csv
    for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "*bench_kernel_stats.csv")))):
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel in row.get("Name", ""):
                    return float(row["AverageNs"]) / 1e3, os.path.relpath(path, ROOT)
    return None, None


def strict_roofline():
    """Issue roofline of the strict band depth's dominant kernel (strict_masks_rank_kernel: all 32-bit integer VALU) from
    the committed SQ counters of the 10 000 x 1 000 workload (profiles/*issue_strict.json): wave-instructions per second
    against the chip's 256 x 4 SIMDs x 2.4 GHz / 2 cycles (VALU_PEAK)."""
    for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "*issue_strict.json")))):
        with open(path) as f:
            doc = json.load(f)
        k = doc.get("sd::strict_masks_rank_kernel")
        if k and "valu_wave_instr_per_us" in k:
            ach = float(k["valu_wave_instr_per_us"]) * 1e6
            peak = VALU_PEAK / 64.0
            return {"bound": "valu", "achieved": ach / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s", "frac": ach / peak,
                    "traffic": None, "kernel": "strict_masks_rank_kernel", "counter_file": os.path.relpath(path, ROOT),
                    "note": "dominant kernel of the leg, profiled (committed counters of the same workload); the leg's ms is measured in this run"}
    return None


def timed(fn, steps, warmup, stream, torch, barrier=None, clock_warm=0):
    """(wall seconds, device ms per step) of `steps` calls of fn(i) behind `warmup` untimed ones.  clock_warm: further untimed
    calls in front of those -- a GPU that has just been idle (a fresh process, a few microseconds of work per step) has not
    reached its sustained clocks after 5 warm-up steps (measured: 0.0535 against 0.0518 ms per step over 20 timed steps);
    reported in the line as clock_warmup_steps."""
    for i in range(clock_warm):
        fn(i)
        if i % 100 == 99:
            torch.cuda.synchronize()                       # keep the launch queue short: the timed region must not inherit a backlog
    if clock_warm:
        torch.cuda.synchronize()
    for i in range(warmup):
        fn(i)
    (barrier or torch.cuda.synchronize)()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(steps):
        fn(i)
    ev1.record(stream)
    (barrier or torch.cuda.synchronize)()
    dt = time.perf_counter() - t0
    return dt, ev0.elapsed_time(ev1) / steps


def walks(torch, T, n, seed, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.randn(T, n, dtype=torch.float64, device=dev, generator=g).cumsum(0)


def extras_single_gpu(torch, dev, stream, lib, check, ALGOS, oracle, reps=10):
    """Secondary workloads on one GPU, each verified against the oracle on a sample of targets."""
    from statdepth_amd import engine
    out = {}

    def mbd_ms(X, J=2, algo="auto", reps=reps, warmup=2):
        T, n = X.shape
        res = torch.empty((n, J - 1), dtype=torch.int64, device=dev)
        a = ALGOS[algo]
        wsb = lib.sd_mbd_workspace_bytes(T, n, n, 1, n, J, a)
        ws = torch.empty(max(int(wsb), 8), dtype=torch.uint8, device=dev)

        def step(_):
            check(lib.sd_mbd_counts(X.data_ptr(), T, n, n, 1, 0, n, J, a, res.data_ptr(), ws.data_ptr(), wsb,
                                    stream.cuda_stream))
        _, ms = timed(step, reps, warmup, stream, torch)
        return ms, res

    def rate(T, n, ms):
        return float(n) * (n - 1) / (ms * 1e-3)

    # config 3 on one GPU, the stretch case, the tie-heavy variant of config 2
    for key, (T, n, seed, nt) in {"config3_one_gpu": (256, 100000, 1235, 5), "stretch_1e5x1e3": (1000, 100000, 1238, 3)}.items():
        X = walks(torch, T, n, seed, dev)
        # 10 untimed + 30 timed calls: behind the generation of the matrix (and the CPU check of the leg before) the GPU needs a
        # few milliseconds of work to be back at its sustained clocks (2 + 5 calls read 4 - 8 % slower than tools/time_rank.py)
        ms, res = mbd_ms(X, reps=30, warmup=10)
        tg = np.linspace(0, n - 1, nt).astype(np.int64)
        want = oracle.mbd_counts(X.cpu().numpy(), tg, 2)
        assert (res[torch.from_numpy(tg).to(dev)].cpu().numpy() == want).all(), key
        balg = 8.0 * T * 2 * n + 8.0 * n
        out[key] = {"workload": f"MBD J=2, {n} curves x {T} timepoints, 1 GPU", "ms": ms, "curve_pairs_per_s": rate(T, n, ms),
                    "roofline_frac": balg / (ms * 1e-3) / HBM_PEAK, "checked_targets": int(nt), "timed_calls": 30,
                    "untimed_calls": 10,
                    "roofline": {"bound": "hbm", "achieved": balg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                 "frac": balg / (ms * 1e-3) / HBM_PEAK, "traffic": None, "algorithmic_bytes": balg,
                                 "kernels_of_step": ["bucket_setup_kernel", "bucket_partition3_kernel", "bucket_rank32_kernel",
                                                     "big_fallback_kernel", "rank_accumulate_h8_kernel"]}}
        del X, res
    Xh = np.round(np.random.default_rng(1234).normal(size=(1000, 10000)).cumsum(axis=0), 1)
    dup = np.random.default_rng(7).choice(10000, size=100, replace=False)
    Xh[:, dup] = Xh[:, (dup + 1) % 10000]
    ms, res = mbd_ms(torch.from_numpy(Xh).to(dev))
    assert (res.cpu().numpy() == oracle.mbd_counts_ranksort(Xh, 2)).all(), "ties variant"
    out["config2_ties"] = {"workload": "config 2 rounded to 0.1 + 1 % duplicated curves", "ms": ms,
                           "curve_pairs_per_s": rate(1000, 10000, ms), "checked_targets": 10000}
    # heavy tails at every timepoint (Cauchy rows): the three-piece key map's tail codes (round 4; every row went to the fp64
    # sort before: 0.24 - 0.27 ms)
    Xh = np.random.default_rng(5).standard_cauchy(size=(1000, 10000))
    ms, res = mbd_ms(torch.from_numpy(Xh).to(dev))
    assert (res.cpu().numpy() == oracle.mbd_counts_ranksort(Xh, 2)).all(), "cauchy variant"
    out["config2_cauchy"] = {"workload": "config 2's shape, standard Cauchy entries (heavy tails at every timepoint)", "ms": ms,
                             "curve_pairs_per_s": rate(1000, 10000, ms), "checked_targets": 10000}
    # the same config-2 call from a HOST array through the Python engine: H2D of the 80 MB matrix + workspace + kernels + D2H
    # (PCIe-inclusive; never the headline value, which starts with the inputs resident in HBM)
    Xw = np.random.default_rng(1234).normal(size=(1000, 10000)).cumsum(axis=0)
    engine.mbd_counts(Xw, None, 2)
    t1 = time.perf_counter()
    for _ in range(3):
        engine.mbd_counts(Xw, None, 2)
    host_ms = (time.perf_counter() - t1) / 3 * 1e3
    out["config2_from_host_array"] = {"workload": "config 2 from a host ndarray through statdepth_amd.engine (pageable H2D of 80 MB, "
                                      "allocation, kernels, D2H, synchronisation)", "ms": host_ms,
                                      "curve_pairs_per_s": rate(1000, 10000, host_ms)}
    # strict band depth (relax=False), 2 000 banded curves x 1 000 timepoints
    rng = np.random.default_rng(11)
    Xs = np.sort(rng.normal(size=2000))[None, :] * 3.0 + rng.normal(size=(1000, 2000)) * 0.3
    Xd = torch.from_numpy(Xs).to(dev)
    T, n = Xs.shape
    res = torch.empty((n, 1), dtype=torch.int64, device=dev)
    wsb = lib.sd_bd_strict_workspace_bytes(T, n, n, 1, n)
    ws = torch.empty(int(wsb), dtype=torch.uint8, device=dev)
    _, ms = timed(lambda _: check(lib.sd_bd_strict_counts(Xd.data_ptr(), T, n, n, 1, 0, n, res.data_ptr(), ws.data_ptr(), wsb,
                                                         stream.cuda_stream)), 3, 1, stream, torch)
    tg = np.arange(0, n, 250)
    assert (res.cpu().numpy()[tg, 0] == oracle.bd_strict_counts(Xs, tg)).all(), "strict"
    out["strict_2000x1000"] = {"workload": "strict band depth J=2, 2000 banded curves x 1000 timepoints", "ms": ms,
                               "pair_tests_per_s": n * (n - 1) * (n - 2) / 2 / (ms * 1e-3), "checked_targets": len(tg)}
    # ... and at config 2's size: 10 000 random walks x 1 000 timepoints, every target
    Xw = np.random.default_rng(12).normal(size=(1000, 10000)).cumsum(axis=0)
    Xd = torch.from_numpy(Xw).to(dev)
    T, n = Xw.shape
    res = torch.empty((n, 1), dtype=torch.int64, device=dev)
    wsb = lib.sd_bd_strict_workspace_bytes(T, n, n, 1, n)
    ws = torch.empty(int(wsb), dtype=torch.uint8, device=dev)
    _, ms = timed(lambda _: check(lib.sd_bd_strict_counts(Xd.data_ptr(), T, n, n, 1, 0, n, res.data_ptr(), ws.data_ptr(), wsb,
                                                         stream.cuda_stream)), 2, 1, stream, torch)
    tg = np.array([0, 2500, 5000, 9999])
    assert (res.cpu().numpy()[tg, 0] == oracle.bd_strict_counts(Xw, tg)).all(), "strict 10000"
    out["strict_10000x1000"] = {"workload": "strict band depth J=2, 10000 random walks x 1000 timepoints, every target", "ms": ms,
                                "pair_tests_per_s": n * (n - 1) * (n - 2) / 2 / (ms * 1e-3), "checked_targets": len(tg),
                                "roofline": strict_roofline()}
    del Xd, res, ws
    # config 5 (i), the reference's default relax=False: L-infinity (box) depth of 10^6 points in R^3, every point a target
    P6 = np.random.default_rng(1237).normal(size=(1000000, 3))
    X6 = np.ascontiguousarray(P6.T)
    Xd = torch.from_numpy(X6).to(dev)
    T, n = X6.shape
    res = torch.empty((n, 1), dtype=torch.int64, device=dev)
    wsb = lib.sd_bd_strict_workspace_bytes(T, n, n, 1, n)
    ws = torch.empty(int(wsb), dtype=torch.uint8, device=dev)
    _, ms = timed(lambda _: check(lib.sd_bd_strict_counts(Xd.data_ptr(), T, n, n, 1, 0, n, res.data_ptr(), ws.data_ptr(), wsb,
                                                         stream.cuda_stream)), 1, 1, stream, torch)
    tg = np.array([0, 123456, 999999])
    assert (res.cpu().numpy()[tg, 0] == oracle.bd_strict_counts_by_states(X6, tg)).all(), "strict linf 1e6"
    out["config5_linf_strict_1e6x3"] = {"workload": "L-infinity (box) depth, relax=False, 10^6 points in R^3, every point", "ms": ms,
                                        "route": "grid of cells",
                                        "point_pairs_per_s": float(n) * (n - 1) / (ms * 1e-3), "checked_targets": len(tg)}
    del Xd, res, ws
    # the same depth in R^4 (grid of cells, like config 5) and R^6, R^8 (10^5 points, every point a target: the workgroup form of
    # the state-class kernel, priced against the vector issue peak)
    for d in (4, 6, 8):
        Xh = np.ascontiguousarray(np.random.default_rng(1237 + d).normal(size=(100000, d)).T)
        Xd = torch.from_numpy(Xh).to(dev)
        T, n = Xh.shape
        res = torch.empty((n, 1), dtype=torch.int64, device=dev)
        wsb = lib.sd_bd_strict_workspace_bytes(T, n, n, 1, n)
        ws = torch.empty(int(wsb), dtype=torch.uint8, device=dev)
        _, ms = timed(lambda _: check(lib.sd_bd_strict_counts(Xd.data_ptr(), T, n, n, 1, 0, n, res.data_ptr(), ws.data_ptr(), wsb,
                                                             stream.cuda_stream)), 3, 1, stream, torch)
        tg = np.array([0, 12345, 99999])
        assert (res.cpu().numpy()[tg, 0] == oracle.bd_strict_counts_by_states(Xh, tg)).all(), f"strict linf 1e5 x {d}"
        out[f"linf_strict_1e5x{d}"] = {"workload": f"L-infinity (box) depth, relax=False, 10^5 points in R^{d}, every point", "ms": ms,
                                       "point_pairs_per_s": float(n) * (n - 1) / (ms * 1e-3), "checked_targets": len(tg),
                                       "route": "grid of cells" if d <= 4 else "state classes, all pairs",
                                       "roofline": issue_roofline(f"class{d}", float(n) * n / (ms * 1e-3)) if d > 4 else None}
        del Xd, res, ws
    # L1 depth and sampled simplicial depth
    P = np.random.default_rng(1237).normal(size=(100000, 3))
    Pd = torch.from_numpy(P).to(dev)
    o = torch.empty(100000, dtype=torch.float64, device=dev)
    _, ms = timed(lambda _: check(lib.sd_l1_depth(Pd.data_ptr(), 100000, 3, 0, 100000, o.data_ptr(), stream.cuda_stream)),
                  3, 1, stream, torch)
    tg = np.arange(0, 100000, 3125)
    assert np.max(np.abs(o.cpu().numpy()[tg] - oracle.l1_depth(P, tg))) <= 1e-12, "l1"
    out["l1_1e5x3"] = {"workload": "L1 depth, 10^5 points in R^3", "ms": ms, "point_pairs_per_s": 1e10 / (ms * 1e-3),
                       "checked_targets": len(tg)}
    oc = torch.empty(100000, dtype=torch.int64, device=dev)
    wss = int(lib.sd_simplex_sampled_workspace_bytes(100000, 0, 3, 256))
    wsx = torch.empty(max(wss, 8), dtype=torch.uint8, device=dev)
    _, ms = timed(lambda _: check(lib.sd_pointcloud_simplex_sampled(Pd.data_ptr(), 100000, 3, 0, 100000, 1e-7, 256, 1237,
                                                                   oc.data_ptr(), wsx.data_ptr(), wss, stream.cuda_stream)), 3, 1, stream, torch)
    tg = np.arange(0, 100000, 500)
    assert (oc.cpu().numpy()[tg] == oracle.simplex_sampled(P, tg, samples=256, seed=1237)).all(), "simplex d=3"
    out["simplex_sampled_d3"] = {"workload": "10^5 points in R^3, 256 tetrahedra per point", "ms": ms,
                                 "simplex_tests_per_s": 1e5 * 256 / (ms * 1e-3), "checked_targets": len(tg)}
    C = np.random.default_rng(1236).normal(size=(500, 50, 8)).cumsum(axis=1)
    C[:8] *= 0.02
    Cd = torch.from_numpy(C).to(dev)
    oc = torch.empty(500, dtype=torch.int64, device=dev)
    wss = int(lib.sd_simplex_sampled_workspace_bytes(500, 50, 8, 256))
    wsx = torch.empty(max(wss, 8), dtype=torch.uint8, device=dev)
    _, ms = timed(lambda _: check(lib.sd_multi_simplex_sampled(Cd.data_ptr(), 500, 50, 8, 0, 500, 1, 1e-7, 256, 1236,
                                                              oc.data_ptr(), wsx.data_ptr(), wss, stream.cuda_stream)), 3, 1, stream, torch)
    del wsx
    tg = np.arange(0, 500, 31)
    assert (oc.cpu().numpy()[tg] == oracle.simplex_sampled(C, tg, relax=True, samples=256, seed=1236)).all(), "simplex d=8"
    out["simplex_sampled_d8"] = {"workload": "500 curves x 50 timepoints x 8 features, 256 subsets per target (config 4 shape)",
                                 "ms": ms, "simplex_tests_per_s": 500 * 256 * 50 / (ms * 1e-3), "checked_targets": len(tg)}
    # config 4 in its componentwise-band form ('r2_enum'): exact, every target
    n4, T4, d4 = 5000, 500, 8
    P4 = walks(torch, T4, n4 * d4, 1236, dev).view(T4, n4, d4).permute(1, 0, 2).contiguous()
    o4 = torch.empty(n4, dtype=torch.int64, device=dev)
    wsb4 = int(lib.sd_multi_band_workspace_bytes(n4, T4, d4))
    ws4 = torch.empty(wsb4, dtype=torch.uint8, device=dev)
    _, ms = timed(lambda _: check(lib.sd_multi_band_counts(P4.data_ptr(), n4, T4, d4, 0, n4, o4.data_ptr(), ws4.data_ptr(), wsb4,
                                                          stream.cuda_stream)), 2, 1, stream, torch)
    sub = P4[:, :2, :].contiguous()                                  # the oracle's literal pair enumeration on 2 timepoints
    o2 = torch.empty(n4, dtype=torch.int64, device=dev)
    check(lib.sd_multi_band_counts(sub.data_ptr(), n4, 2, d4, 0, n4, o2.data_ptr(), ws4.data_ptr(), wsb4, stream.cuda_stream))
    tg = np.array([0, 1234, 4999])
    assert (o2.cpu().numpy()[tg] == oracle.multi_band_enum(sub.cpu().numpy(), tg, 2, True)[:, 0]).all(), "r2_enum"
    out["config4_componentwise_band"] = {"workload": "5000 curves x 500 timepoints x 8 features, componentwise band containment "
                                         "(r2_enum), exact pair counts, all targets", "ms": ms,
                                         "curve_pairs_per_s": float(n4) * (n4 - 1) / (ms * 1e-3), "checked_targets": len(tg)}
    del P4, ws4
    out.update(extras_full_size_configs(torch, dev, stream, lib, check, oracle))
    return out


def extras_full_size_configs(torch, dev, stream, lib, check, oracle):
    """BASELINE.json configs 1, 4 and 5 at their stated sizes, each checked against the oracle on a
    sample and carrying its own roofline object (VALU-issue based: these kernels are fp64-compute bound, not HBM bound)."""
    out = {}
    # ---- config 1: 50 curves x 100 timepoints through the DataFrame -> Series API, both relax modes, beside the
    # reference's own time on this recipe (39.9 s / 42.5 s on one core: tests/golden g7)
    import pandas as pd
    from statdepth_amd import FunctionalDepth
    X1 = np.random.default_rng(0).normal(size=(100, 50))
    df1 = pd.DataFrame(X1)
    for relax, ref_s in ((True, 39.9), (False, 42.5)):
        FunctionalDepth([df1], J=2, relax=relax)                               # first call: allocations, module load
        t1 = time.perf_counter()
        for _ in range(5):
            d = FunctionalDepth([df1], J=2, relax=relax)
        ms = (time.perf_counter() - t1) / 5 * 1e3
        want = oracle.univariate_depths(X1, None, J=2, relax=relax)
        assert np.max(np.abs(d.to_numpy() - want)) <= 1e-12, "config 1"
        out[f"config1_api_relax_{relax}"] = {
            "workload": f"config 1: FunctionalDepth([50 curves x 100 timepoints DataFrame], J=2, relax={relax}) -> Series, "
                        "host API end to end (validation, H2D, kernels, D2H, normalisation)",
            "ms": ms, "curve_pairs_per_s": 50 * 49 / (ms * 1e-3), "reference_seconds_recorded": ref_s,
            "speedup_vs_reference_recorded": ref_s / (ms * 1e-3), "checked_targets": 50,
            "roofline": {"bound": "latency", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                         "note": "40 KB of data: the call is host / launch latency, no device roof applies"}}
    # ---- config 4 (i): 5 000 curves x 500 timepoints x 8 features, 4 096 sampled 9-point simplices per target
    n4, T4, d4, S4 = 5000, 500, 8, 4096
    g = torch.Generator(device=dev).manual_seed(1236)
    C4 = torch.randn(n4, T4, d4, dtype=torch.float64, device=dev, generator=g).cumsum(1)
    o4 = torch.empty(n4, dtype=torch.int64, device=dev)
    wss = int(lib.sd_simplex_sampled_workspace_bytes(n4, T4, d4, S4))
    wsx = torch.empty(max(wss, 8), dtype=torch.uint8, device=dev)
    _, ms = timed(lambda _: check(lib.sd_multi_simplex_sampled(C4.data_ptr(), n4, T4, d4, 0, n4, 1, 1e-7, S4, 1236, o4.data_ptr(),
                                                              wsx.data_ptr(), wss, stream.cuda_stream)), 1, 1, stream, torch)
    tg = np.array([0, 2500, 4999])
    assert (o4.cpu().numpy()[tg] == oracle.simplex_sampled(C4.cpu().numpy(), tg, relax=True, samples=S4, seed=1236)).all(), "config 4"
    units = float(n4) * S4 * T4
    out["config4_sampled_simplex_full"] = {
        "workload": "config 4 (i): 5000 curves x 500 timepoints x 8 features, 4096 sampled subsets per target (seed 1236), relax=True",
        "ms": ms, "simplex_tests_per_s": units / (ms * 1e-3), "checked_targets": len(tg),
        "roofline": issue_roofline("simplex8", units / (ms * 1e-3))}
    del C4, o4, wsx
    # ---- config 5 (ii) L1 depth and (iii) 4 096 sampled tetrahedra per point, 10^6 points in R^3
    n5 = 1000000
    P5 = np.random.default_rng(1237).normal(size=(n5, 3))
    Pd = torch.from_numpy(P5).to(dev)
    o5 = torch.empty(n5, dtype=torch.float64, device=dev)
    _, ms = timed(lambda _: check(lib.sd_l1_depth(Pd.data_ptr(), n5, 3, 0, n5, o5.data_ptr(), stream.cuda_stream)), 1, 0, stream, torch)
    tg = np.arange(0, n5, 62500)
    assert np.max(np.abs(o5.cpu().numpy()[tg] - oracle.l1_depth(P5, tg))) <= 1e-12, "config 5 l1"
    units = float(n5) * n5
    out["config5_l1_full"] = {"workload": "config 5 (ii): L1 depth, 10^6 points in R^3, every point", "ms": ms,
                              "point_pairs_per_s": units / (ms * 1e-3), "checked_targets": len(tg),
                              "roofline": issue_roofline("l1", units / (ms * 1e-3))}
    oc = torch.empty(n5, dtype=torch.int64, device=dev)
    wss = int(lib.sd_simplex_sampled_workspace_bytes(n5, 0, 3, 4096))
    wsx = torch.empty(max(wss, 8), dtype=torch.uint8, device=dev)
    _, ms = timed(lambda _: check(lib.sd_pointcloud_simplex_sampled(Pd.data_ptr(), n5, 3, 0, n5, 1e-7, 4096, 1237, oc.data_ptr(),
                                                                   wsx.data_ptr(), wss, stream.cuda_stream)), 2, 1, stream, torch)
    tg = np.arange(0, n5, 10000)
    assert (oc.cpu().numpy()[tg] == oracle.simplex_sampled(P5, tg, samples=4096, seed=1237)).all(), "config 5 simplex"
    units = float(n5) * 4096
    out["config5_sampled_simplex_full"] = {
        "workload": "config 5 (iii): 10^6 points in R^3, 4096 sampled tetrahedra per point (seed 1237)", "ms": ms,
        "simplex_tests_per_s": units / (ms * 1e-3), "checked_targets": len(tg),
        "roofline": issue_roofline("simplex3", units / (ms * 1e-3))}
    return out


def extras_multi_gpu(torch, dist, dev, rank, N, reps=5):
    """Configs 3 and 5 of BASELINE.json across the N ranks (strong scaling: the data set is fixed, the ranks share it)."""
    from statdepth_amd.distributed import sharded_mbd_counts, sharded_pointcloud
    stream = torch.cuda.current_stream(dev)
    out = {}

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    def tmax(dt):
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    n, T = 100000, 256
    sizes = [n // N + (1 if r < n % N else 0) for r in range(N)]
    X_loc = walks(torch, T, sizes[rank], 1235 + rank, dev)
    for mode in ("time", "targets"):
        dt, _ = timed(lambda _: sharded_mbd_counts(X_loc, J=2, sizes=sizes, mode=mode), reps, 2, stream, torch, barrier)
        dt = tmax(dt)
        out[f"config3_{mode}"] = {"workload": f"MBD J=2, {n} curves x {T} timepoints over {N} GPUs ({mode}-sharded)",
                                  "scaling": "strong", "ms": dt / reps * 1e3, "curve_pairs_per_s": float(n) * (n - 1) / (dt / reps)}
    # strict band depth (the reference's default relax=False) of the same curves: targets sharded, one all-gather
    from statdepth_amd.distributed import sharded_bd_strict_counts
    dt, _ = timed(lambda _: sharded_bd_strict_counts(X_loc, J=2, sizes=sizes), 1, 1, stream, torch, barrier)
    dt = tmax(dt)
    out["config3_strict_targets"] = {"workload": f"strict band depth J=2, {n} curves x {T} timepoints over {N} GPUs (target-sharded)",
                                     "scaling": "strong", "ms": dt * 1e3, "targets_per_s": n / dt}
    npts = 1000000
    P_loc = torch.from_numpy(np.random.default_rng(1237 + rank).normal(size=(npts // N, 3))).to(dev)
    # config 5 (i): L-infinity (box) depth, the reference's default relax=False, targets sharded
    Xp = P_loc.t().contiguous()
    psz = [npts // N] * N
    dt, _ = timed(lambda _: sharded_bd_strict_counts(Xp, J=2, sizes=psz), 1, 1, stream, torch, barrier)
    dt = tmax(dt)
    out["config5_linf_strict"] = {"workload": f"L-infinity depth (relax=False), 10^6 points in R^3 over {N} GPUs (target-sharded)",
                                  "scaling": "strong", "ms": dt * 1e3, "units_per_s": float(npts) * npts / dt}
    for cont, kw, units in (("simplex", {"samples": 4096, "seed": 1237}, npts * 4096.0), ("l1", {}, float(npts) * npts)):
        dt, _ = timed(lambda _: sharded_pointcloud(P_loc, cont, **kw), 2, 1, stream, torch, barrier)
        dt = tmax(dt)
        out[f"config5_{cont}"] = {"workload": f"10^6 points in R^3 over {N} GPUs, {cont}" + (" (4096 sampled tetrahedra per point)" if kw else ""),
                                  "scaling": "strong", "ms": dt / 2 * 1e3, "units_per_s": units / (dt / 2)}
    return out


def launch_plan(gpus, argv, env):
    """How `bench.py --gpus N` gets its ranks.  Under a launcher (WORLD_SIZE set: the driver's `python -m torch.distributed.run
    ...`) or at N = 1 this process IS a rank.  With N > 1 and no launcher the process becomes a parent that starts N ranks of
    itself through torch.distributed.run on 127.0.0.1 and relays rank 0's JSON line -- it never imports torch or touches the
    GPU, so nothing that has initialised HIP is ever exec'ed or re-exec'ed."""
    world = env.get("WORLD_SIZE")
    child_args = [a for a in argv if a != "--print-launch"]
    if gpus <= 1 or world is not None:
        return {"self_launch": False, "world_size": int(world or 1), "argv": None,
                "reason": "launched under torch.distributed.run" if world is not None else "single GPU: this process is rank 0"}
    port = int(env.get("MASTER_PORT", "0")) or (29600 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + child_args
    return {"self_launch": True, "world_size": gpus, "argv": cmd,
            "reason": f"--gpus {gpus} without WORLD_SIZE: parent starts {gpus} ranks as child processes (one per GPU, RCCL)"}


def self_launch(plan):
    """Parent side of a self-launched multi-GPU run: children inherit stdout (rank 0 prints the one JSON line), the parent
    exits with their code.  No GPU call happens in this process."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    sys.stdout.flush()
    return subprocess.run(plan["argv"], env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n-loc", type=int, default=10000)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--J", type=int, default=2)
    ap.add_argument("--algo", default="auto", choices=["auto", "pairwise", "rank"])
    ap.add_argument("--rotate", type=int, default=4,
                    help="distinct matrices the timed steps cycle through at N = 1 (4 x 80 MB > 256 MiB Infinity Cache); "
                         "1 = replay one matrix")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--cpu-targets", type=int, default=10000)
    ap.add_argument("--variant", default="walk", choices=["walk", "ties"],
                    help="walk: continuous random walks (headline); ties: the same rounded to 1 decimal with 1 %% "
                         "duplicated curves (SURVEY.md 8(d) config 2 variant)")
    ap.add_argument("--mode", default="auto", choices=["auto", "time", "targets"], help="multi-GPU decomposition")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the torch.distributed path even at world size 1 (exercises RCCL on one GPU)")
    ap.add_argument("--extras-multi", action="store_true",
                    help="with --force-dist: also run the multi-GPU extras (configs 3 and 5 through the sharded paths) at world size 1")
    ap.add_argument("--print-launch", action="store_true",
                    help="print the launch plan (the child command for --gpus N > 1 without a launcher) as JSON and exit; "
                         "touches no GPU")
    args = ap.parse_args()

    plan = launch_plan(args.gpus, sys.argv[1:], os.environ)
    if args.print_launch:
        print(json.dumps(plan))
        return 0
    if plan["self_launch"]:
        return self_launch(plan)

    import torch
    import torch.distributed as dist
    from statdepth_amd import _native
    from statdepth_amd._native import ALGOS, check

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: rank {rank}: no HIP device {local} on this host ({torch.cuda.device_count()} visible); "
                         "the hot path has no CPU fallback")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on STDOUT when its first communicator comes up; the contract is ONE JSON line there,
        # so stdout points at stderr until the communicator exists
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            torch.cuda.set_device(local)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
    N = world
    if args.gpus != N:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or "
                         f"unset WORLD_SIZE and let bench.py start its own ranks")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = _native.require_device()

    T, n_loc, J = args.T, args.n_loc, args.J
    n = n_loc * N
    K = 1 if use_dist else max(1, args.rotate)

    def make_host(seed):
        # synthetic curves: random walks (SURVEY.md 8(d) config 2 recipe), one block per rank
        rng = np.random.default_rng(seed)
        X = rng.normal(size=(T, n_loc)).cumsum(axis=0)
        if args.variant == "ties":
            X = np.round(X, 1)
            dup = rng.choice(n_loc, size=max(1, n_loc // 100), replace=False)
            X[:, dup] = X[:, (dup + 1) % n_loc]
        return X

    X_host = make_host(1234 + rank)
    mats = [torch.from_numpy(X_host).to(dev)]                     # [T, n_loc] time-major, resident in HBM
    for i in range(1, K):
        mats.append(torch.from_numpy(make_host(4321 + i)).to(dev))
    outs = [torch.empty((n_loc, J - 1), dtype=torch.int64, device=dev) for _ in range(K)]
    algo = ALGOS[args.algo]
    wsb = 0 if use_dist else lib.sd_mbd_workspace_bytes(T, n, n, 1, n_loc, J, algo)
    ws = torch.empty(max(int(wsb), 8), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)

    from statdepth_amd.distributed import sharded_mbd_counts
    sizes = [n_loc] * N

    def step_on(k):
        if use_dist:
            # product multi-GPU path (statdepth_amd/distributed.py): time-sharded for the rank kernels (RCCL all-to-all of
            # curve blocks -> per-curve partial totals -> reduce-scatter), target-sharded (all-gather) for --algo pairwise
            outs[0].copy_(sharded_mbd_counts(mats[0], J=J, algo=args.algo, sizes=sizes, mode=args.mode))
        else:
            check(lib.sd_mbd_counts_range(mats[k].data_ptr(), T, n, n, 1, rank * n_loc, n_loc, J, algo,
                                          outs[k].data_ptr(), ws.data_ptr(), wsb, stream.cuda_stream))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    clock_warm = max(0, 400 - args.warmup)                  # ~20 ms of the same step at N = 1
    dt, dev_ms = timed(lambda i: step_on(i % K), args.steps, args.warmup, stream, torch, barrier, clock_warm)
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    pairs_per_step = float(n_loc) * N * (n - 1)          # ordered (target, other) pairs, whole job
    value = pairs_per_step / (dt / args.steps)
    replay = None
    if K > 1:                                            # the same loop on ONE matrix: the working set stays in the Infinity Cache
        rdt, rdev = timed(lambda i: step_on(0), args.steps, args.warmup, stream, torch, barrier)
        replay = {"ms_per_step": rdt / args.steps * 1e3, "kernel_ms": rdev,
                  "note": "one 80 MB matrix replayed back to back (Infinity-Cache resident)"}

    result_sum = int(outs[0].sum().item())
    extras = None
    extras_error = None
    if not args.no_extras and args.variant == "walk" and (n_loc, T, J) == (10000, 1000, 2):
        if N > 1 or (use_dist and args.extras_multi):
            try:                                         # a failing secondary leg must not cost the headline its line
                extras = extras_multi_gpu(torch, dist, dev, rank, N)
            except Exception as e:                       # noqa: BLE001
                extras_error = f"{type(e).__name__}: {e}"
    if rank == 0:
        used = "rank" if (args.algo == "rank" or (args.algo == "auto" and J <= 3)) else "pairwise"
        if used == "pairwise":
            kerns = ["mbd_pairwise_kernel"]
        elif 3072 < n <= 11264 and J == 2 and n_loc == n and 256 <= T <= 4096:
            # two launches: 32-bit key images, two workgroups per CU; then rank_bucket_kernel's SEL form (finalize + flagged rows)
            kerns = ["rank_bucket32_kernel", "rank_bucket_kernel"]
        elif n <= 16384:
            kerns = ["rank_bucket_kernel", "rank_finalize"]
        else:
            kerns = ["bucket_rank_kernel"]
        bytes_alg = 8.0 * T * (n + n_loc) + 8.0 * n_loc * (J - 1)   # per GPU per call
        achieved = bytes_alg / (dev_ms * 1e-3)
        traffic, source = pmc_traffic(kerns, n, T, J) if N == 1 else (None, None)
        kus, ksrc = profiled_kernel_us(kerns[0], n, T, J) if N == 1 else (None, None)
        from statdepth_amd.distributed import mode_cost_model
        model = mode_cost_model(T, n_loc, N, J, args.algo)
        mode = args.mode if args.mode != "auto" else model["choice"]
        line = {
            "metric": "curve-pairs/sec (MBD)", "value": value, "unit": "curve-pairs/s",
            "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "untimed_steps": clock_warm + args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"MBD J={J}: {n} curves x {T} timepoints (n_loc={n_loc} targets per GPU), fp64 "
                                   f"random walks{' rounded to 0.1 + 1% duplicates' if args.variant == 'ties' else ''}, time-major"
                                   + (f"; {K} distinct matrices in rotation ({K * T * n_loc * 8 / 2**20:.0f} MiB > Infinity Cache)" if K > 1 else ""),
                       "algorithm": used,
                       "parallelism": (f"single GPU" if N == 1 else
                                       (f"targets sharded x{N}, RCCL all-gather of curve blocks" if mode == "targets" else
                                        f"curves owned x{N}, timepoints sharded for the ranking: RCCL all-to-all + reduce-scatter"))},
            "pair_timepoints_per_s": value * T,
            "clock_warmup_steps": clock_warm,
            "mode": {"used": mode, "cost_model_seconds": {"time": model["time"], "targets": model["targets"]},
                     "predicted_step_ms": (model[mode] * 1e3 if mode in model else None), "measured_step_ms": ms_per_step,
                     "note": "statdepth_amd.distributed.mode_cost_model: exchange over one xGMI link per peer + the measured "
                             "single-GPU ranking rates; mode='auto' takes the cheaper decomposition"},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": source,
                         "hbm_rate": (traffic / (dev_ms * 1e-3) / 1e9) if traffic else None,
                         "kernel": kerns[0], "kernels_of_step": kerns,
                         "kernel_ms": dev_ms, "algorithmic_bytes": bytes_alg,
                         "dominant_kernel_us_profiled": kus, "dominant_kernel_source": ksrc,
                         "frac_dominant_kernel_profiled": (bytes_alg / (kus * 1e-6) / HBM_PEAK) if kus else None,
                         "note": "kernel_ms = device time of one whole step (all kernels of the step) by HIP events on the launch "
                                 "stream, measured in this run; achieved = algorithmic_bytes / kernel_ms, a lower bound for the "
                                 "dominant kernel, whose own average duration (rocprofv3 --stats of the same command, committed "
                                 "file named beside it) gives frac_dominant_kernel_profiled"},
            "checksum": result_sum,
        }
        if replay:
            replay["roofline_frac"] = bytes_alg / (replay["kernel_ms"] * 1e-3) / HBM_PEAK
            line["replay"] = replay
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
            got = outs[0][:m_cpu].cpu().numpy()
            assert (got == want).all(), "HIP counts differ from the CPU oracle on the baseline sample"
            line["cpu_baseline"] = {
                "value": m_cpu * (n - 1) / cpu_dt, "unit": "curve-pairs/s", "cores": oracle.num_threads(),
                "kind": "port",
                "sample": f"oracle_mbd_counts (C, OpenMP; the reference's O(n^2 T) enumeration in closed form) on the first {m_cpu} "
                          f"of {n_loc} targets x all {n} curves x {T} timepoints, {cpu_dt:.2f} s; work is linear in #targets",
            }
            t1 = time.perf_counter()
            want2 = oracle.mbd_counts_ranksort(X_host, J)
            cpu_dt2 = time.perf_counter() - t1
            assert (outs[0].cpu().numpy() == want2).all(), "HIP counts differ from the CPU rank-sort oracle"
            line["cpu_baseline_rank"] = {
                "value": n_loc * (n - 1) / cpu_dt2, "unit": "curve-pairs/s", "cores": oracle.num_threads(), "kind": "port",
                "sample": f"oracle_mbd_counts_ranksort (C, OpenMP): the GPU's own O(n T log n) rank formulation, all {n_loc} "
                          f"targets, {cpu_dt2:.2f} s -- the like-for-like CPU number",
            }
            if not args.no_extras and not use_dist and args.variant == "walk" and (n_loc, T, J) == (10000, 1000, 2):
                del mats[1:], outs[1:]
                try:
                    extras = extras_single_gpu(torch, dev, stream, lib, check, ALGOS, oracle)
                except Exception as e:                   # noqa: BLE001
                    extras_error = f"{type(e).__name__}: {e}"
        if extras:
            line["extras"] = extras
        if extras_error:
            line["extras_error"] = extras_error
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
