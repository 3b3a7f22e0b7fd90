"""world_size=2 (and 3, ragged) gloo tests of the target-sharded path on CPU.

The exchange logic (ragged all-gather of curve blocks, target slices, result gather,
global normaliser) is what is under test; the compute hook is the oracle here because
there is no GPU in this container -- on a GPU box the default hook is the HIP engine
(tests/test_hip_parity.py covers that kernel).
"""
import os
import socket

import numpy as np
import pandas as pd
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_compute(X_all, targets, J, algo):
    import oracle
    return oracle.mbd_counts(X_all.cpu().numpy(), targets, J)


def _oracle_compute_all(X_rows, J, algo):
    import oracle
    return oracle.mbd_counts(X_rows.cpu().numpy(), None, J)


def _oracle_compute_strict(X_all, targets, J):
    import oracle
    X = X_all.cpu().numpy()
    if J == 2:
        return oracle.bd_strict_counts(X, targets)[:, None]
    return oracle.band_enum(X, targets, J, relax=False)


def _worker(rank, world, port, splits, J, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from statdepth_amd.distributed import sharded_functional_depth, sharded_mbd_counts
        rng = np.random.default_rng(42)
        T, n = 23, splits[-1]
        X = np.round(rng.normal(size=(T, n)).cumsum(axis=0), 1)
        lo, hi = splits[rank], splits[rank + 1]
        X_loc = torch.from_numpy(np.ascontiguousarray(X[:, lo:hi]))
        hooks = dict(_compute=_oracle_compute, _compute_all=_oracle_compute_all)
        loc = sharded_mbd_counts(X_loc, J=J, mode="targets", **hooks)
        full = sharded_mbd_counts(X_loc, J=J, mode="targets", gather_result=True, **hooks)
        # time-sharded exchange (all-to-all + reduction): same integers
        loc_t = sharded_mbd_counts(X_loc, J=J, mode="time", **hooks)
        full_t = sharded_mbd_counts(X_loc, J=J, mode="auto", gather_result=True, **hooks)
        assert (loc_t == loc).all() and (full_t == full).all()
        # the exchange pipelined in 1, 3 and (more sub-slices than rows per rank) 50 pieces: same integers
        # (and the all-gather of the targets mode in row chunks, each computed on as it arrives: totals add over timepoints)
        for k in (1, 3, 50):
            assert (sharded_mbd_counts(X_loc, J=J, mode="time", chunks=k, **hooks) == loc).all(), k
            assert (sharded_mbd_counts(X_loc, J=J, mode="targets", chunks=k, **hooks) == loc).all(), k
        # a row range of the blocks, asynchronously: what the chunked targets mode is made of
        from statdepth_amd.distributed import gather_curve_blocks
        (work, finish), offs = gather_curve_blocks(X_loc, rows=(3, 11), async_op=True)
        assert (finish().numpy() == X[3:11]).all() and list(offs) == list(splits)
        df = pd.DataFrame(X[:, lo:hi], columns=[f"c{i}" for i in range(lo, hi)])
        ser = sharded_functional_depth(df, J=J, relax=True, **hooks)
        # strict depth: targets split, all-gather of the blocks
        from statdepth_amd.distributed import sharded_bd_strict_counts
        strict = sharded_bd_strict_counts(X_loc, J=J, _compute=_oracle_compute_strict)
        ser_s = sharded_functional_depth(df, J=J, relax=False, _compute_strict=_oracle_compute_strict)
        # ranks that disagree on T must get an error, not a hang inside the collective
        bad = X_loc[: T - 1] if rank == 0 else X_loc
        try:
            sharded_mbd_counts(bad, J=J, mode="time", **hooks)
            mismatch = "no error"
        except ValueError as e:
            mismatch = "ValueError" if "disagree" in str(e) else repr(e)
        q.put((rank, loc.numpy(), full.numpy(), ser, strict.numpy(), ser_s, mismatch))
    except Exception as e:   # surface the failure instead of letting the parent time out
        q.put((rank, repr(e), None, None, None, None, None))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("splits,J", [([0, 20, 40], 2), ([0, 7, 30], 3), ([0, 5, 6, 31], 2)])
def test_sharded_equals_single(splits, J):
    import oracle
    world = len(splits) - 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, splits, J, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(42)
    T, n = 23, splits[-1]
    X = np.round(rng.normal(size=(T, n)).cumsum(axis=0), 1)
    want = oracle.mbd_counts(X, None, J)
    wantd = oracle.univariate_depths(X, None, J=J, relax=True)
    want_s = oracle.bd_strict_counts(X)[:, None] if J == 2 else oracle.band_enum(X, None, J, relax=False)
    wantd_s = oracle.univariate_depths(X, None, J=J, relax=False)
    for rank, loc, full, ser, strict, ser_s, mismatch in res:
        assert not isinstance(loc, str), loc
        lo, hi = splits[rank], splits[rank + 1]
        assert (loc == want[lo:hi]).all()            # integer totals do not depend on the sharding
        assert (full == want).all()
        assert list(ser.index) == [f"c{i}" for i in range(lo, hi)]
        assert np.max(np.abs(ser.to_numpy() - wantd[lo:hi])) <= 1e-12
        assert (strict == want_s[lo:hi]).all()       # strict depth, targets split
        assert np.max(np.abs(ser_s.to_numpy() - wantd_s[lo:hi])) <= 1e-12
        assert mismatch == "ValueError", mismatch    # unequal T across ranks is refused before any exchange


def _single_rank_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from statdepth_amd.distributed import sharded_mbd_counts
        X = np.round(np.random.default_rng(3).normal(size=(11, 40)).cumsum(axis=0), 1)
        X_loc = torch.from_numpy(X)
        calls = []

        def counting(X_rows, J, algo):
            calls.append(tuple(X_rows.shape))
            return _oracle_compute_all(X_rows, J, algo)
        a = sharded_mbd_counts(X_loc, J=2, mode="time", _compute_all=counting)
        n_direct = len(calls)
        b = sharded_mbd_counts(X_loc, J=2, mode="time", _compute_all=counting, _force_exchange=True)
        q.put((a.numpy(), b.numpy(), n_direct, len(calls) - n_direct, X))
    finally:
        dist.destroy_process_group()


def test_single_rank_needs_no_exchange():
    """World size 1: one compute call on the block as it lies (no all-to-all, no reduction); the forced exchange gives
    the same integers through the collectives."""
    import oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_rank_worker, args=(_free_port(), q))
    p.start()
    a, b, n_direct, n_forced, X = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    want = oracle.mbd_counts(X, None, 2)
    assert (a == want).all() and (b == want).all()
    assert n_direct == 1 and n_forced >= 1


def _pc_oracle(P_all, targets, containment, samples, seed):
    import oracle
    P = P_all.cpu().numpy()
    if containment == "l1":
        return oracle.l1_depth(P, targets)
    if samples is None:
        return oracle.pointcloud_simplex_counts(P, targets)
    return oracle.simplex_sampled(P, targets, samples=samples, seed=seed)


def _pc_worker(rank, world, port, splits, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from statdepth_amd.distributed import sharded_pointcloud
        rng = np.random.default_rng(7)
        P = rng.normal(size=(splits[-1], 2))
        lo, hi = splits[rank], splits[rank + 1]
        P_loc = torch.from_numpy(np.ascontiguousarray(P[lo:hi]))
        ex, n1 = sharded_pointcloud(P_loc, "simplex", _compute=_pc_oracle)
        sm, n2 = sharded_pointcloud(P_loc, "simplex", samples=50, seed=3, _compute=_pc_oracle)
        l1, n3 = sharded_pointcloud(P_loc, "l1", _compute=_pc_oracle)
        assert n1 == n2 == n3 == splits[-1]
        q.put((rank, np.asarray(ex), np.asarray(sm), np.asarray(l1)))
    except Exception as e:
        q.put((rank, repr(e), None, None))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("splits", [[0, 9, 18], [0, 4, 11, 19]])
def test_sharded_pointcloud_equals_single(splits):
    import oracle
    world = len(splits) - 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pc_worker, args=(r, world, port, splits, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P = np.random.default_rng(7).normal(size=(splits[-1], 2))
    want_ex = oracle.pointcloud_simplex_counts(P)
    want_sm = oracle.simplex_sampled(P, np.arange(splits[-1]), samples=50, seed=3)
    want_l1 = oracle.l1_depth(P)
    for rank, ex, sm, l1 in res:
        assert not isinstance(ex, str), ex
        lo, hi = splits[rank], splits[rank + 1]
        assert (ex == want_ex[lo:hi]).all() and (sm == want_sm[lo:hi]).all()
        assert np.max(np.abs(l1 - want_l1[lo:hi])) <= 1e-12


def test_mode_cost_model_choices_are_pinned():
    """mode="auto" picks the collective pattern from mode_cost_model: the choice for representative shapes, and that it is a
    function of what all ranks share (T, the largest block, the world size, J, algo) -- never of the calling rank."""
    from statdepth_amd.distributed import mode_cost_model
    # one GPU: no exchange either way, the time split never loses
    assert mode_cost_model(1000, 10000, 1)["choice"] == "time"
    # the rank kernels (J <= 3): splitting the timepoints divides the ranking, the all-gather split does all of it on every rank
    for world in (2, 4, 8):
        m = mode_cost_model(1000, 10000, world)
        assert m["choice"] == "time" and m["time"] < m["targets"]
        m = mode_cost_model(256, 12500, world)
        assert m["choice"] == "time"
    # fewer timepoints than ranks: nothing to split
    assert mode_cost_model(3, 125000, 8)["choice"] == "targets"
    # the pairwise kernel (asked for, or J >= 4): per-target work, the targets are what is split
    assert mode_cost_model(1000, 10000, 8, algo="pairwise")["choice"] == "targets"
    assert mode_cost_model(1000, 10000, 8, J=4)["choice"] == "targets"
    # same arguments, same answer: the model holds no state and reads no rank
    a, b = mode_cost_model(777, 4321, 4, 3), mode_cost_model(777, 4321, 4, 3)
    assert a == b and set(a) >= {"time", "targets", "choice"}
