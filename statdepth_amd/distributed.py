"""Multi-GPU band depth: one process per GPU, targets sharded, one exchange step.

Targets are independent (the reference's outer loops carry no state, _functional.py:74,
_pointcloud.py:45), so rank r owns a block of curves, contributes it to an all-gather
(RCCL over xGMI when the backend is "nccl"; gloo in CPU tests) and computes the totals of
its own curves against the gathered set.  Integer totals do not depend on the number of
ranks.  The only other communication is the optional gather of the (tiny) results.
"""
import numpy as np

from . import engine


def _dist():
    import torch.distributed as dist
    return dist


def _default_compute(X_all, targets, J, algo):
    """HIP path: counts of `targets` within the gathered matrix (device tensor in, device tensor out)."""
    return engine.mbd_counts(X_all, targets, J=J, algo=algo, return_tensor=True)


def gather_curve_blocks(X_loc, group=None, sizes=None):
    """All-gather of per-rank curve blocks.

    X_loc: [T, n_loc] time-major tensor (n_loc may differ between ranks).
    sizes: per-rank block sizes when the caller already knows the partition (skips the size exchange).
    Returns (X_all [T, n] with rank blocks side by side, offsets [world+1]).
    """
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    T, n_loc = X_loc.shape
    if sizes is None:
        szt = torch.zeros(world, dtype=torch.int64, device=X_loc.device)
        mine = torch.tensor([n_loc], dtype=torch.int64, device=X_loc.device)
        dist.all_gather_into_tensor(szt, mine, group=group)
        sizes = szt.cpu().tolist()
    sizes = [int(v) for v in sizes]
    assert len(sizes) == world and sizes[dist.get_rank(group)] == n_loc
    nmax = max(sizes)
    # blocks travel curve-major ([n_loc, T] rows are whole curves) so a ragged tail is plain padding
    send = torch.zeros((nmax, T), dtype=X_loc.dtype, device=X_loc.device)
    send[:n_loc].copy_(X_loc.t())
    recv = torch.empty((world * nmax, T), dtype=X_loc.dtype, device=X_loc.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, nmax, T)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    X_all = torch.empty((T, int(offsets[-1])), dtype=X_loc.dtype, device=X_loc.device)
    for r in range(world):
        X_all[:, offsets[r]:offsets[r + 1]].copy_(recv[r, :sizes[r]].t())
    return X_all, offsets


def sharded_mbd_counts(X_loc, J=2, algo="auto", group=None, gather_result=False, sizes=None, _compute=None):
    """MBD containment totals of this rank's curves against the union of all ranks' curves.

    Returns int64 [n_loc, J-1] (device of X_loc), or with gather_result=True the full
    [n, J-1] array on every rank in global curve order.
    `_compute` is a test hook (CPU/gloo tests inject a checker); the product path is the HIP engine.
    """
    import torch
    dist = _dist()
    rank = dist.get_rank(group)
    X_all, offsets = gather_curve_blocks(X_loc, group, sizes)
    targets = np.arange(offsets[rank], offsets[rank + 1], dtype=np.int64)
    compute = _compute or _default_compute
    local = compute(X_all, targets, J, algo)
    if not isinstance(local, torch.Tensor):
        local = torch.as_tensor(np.asarray(local), device=X_loc.device)
    if not gather_result:
        return local
    world = dist.get_world_size(group)
    sizes = np.diff(offsets)
    nmax = int(sizes.max())
    send = torch.zeros((nmax, J - 1), dtype=torch.int64, device=X_loc.device)
    send[:local.shape[0]].copy_(local)
    recv = torch.empty((world * nmax, J - 1), dtype=torch.int64, device=X_loc.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, nmax, J - 1)
    return torch.cat([recv[r, :int(sizes[r])] for r in range(world)], dim=0)


def sharded_functional_depth(df_local, J=2, relax=True, algo="auto", group=None, _compute=None):
    """FunctionalDepth over curves sharded by column blocks: returns this rank's depth Series.

    Normalisation as the reference (_functional.py:229,253): / T / C(n, j) with n the GLOBAL number of curves.
    """
    import pandas as pd
    import torch
    from scipy.special import binom
    if not relax:
        raise NotImplementedError("sharded path covers relax=True (modified band depth)")
    dist = _dist()
    X = np.ascontiguousarray(df_local.to_numpy(dtype=np.float64))
    dev = engine._device() if _compute is None else torch.device("cpu")
    X_loc = torch.from_numpy(X).to(dev)
    counts = sharded_mbd_counts(X_loc, J=J, algo=algo, group=group, _compute=_compute).cpu().numpy()
    n_tot = torch.tensor([X.shape[1]], dtype=torch.int64, device=dev)
    dist.all_reduce(n_tot, group=group)
    n, T = int(n_tot.item()), X.shape[0]
    depth = np.zeros(X.shape[1])
    for j in range(2, J + 1):
        depth += counts[:, j - 2].astype(np.float64) / T / binom(n, j)
    return pd.Series(index=df_local.columns, data=depth)
