"""Multi-GPU band depth: one process per GPU, curves owned by blocks, one exchange step.

Targets are independent (the reference's outer loops carry no state, _functional.py:74,
_pointcloud.py:45), so rank r owns a block of curves, contributes it to an all-gather
(RCCL over xGMI when the backend is "nccl"; gloo in CPU tests) and computes the totals of
its own curves against the gathered set.  Integer totals do not depend on the number of
ranks.  The only other communication is the optional gather of the (tiny) results.
"""
import numpy as np

from . import engine

# Pieces the exchanges are cut into so that one piece travels (RCCL's stream) while the one before it is computed on (the
# current stream): sub-slices of every rank's time slice in mode "time", row chunks of the all-gather in mode "targets".
# A keyword of the sharded_* functions (`chunks=`); the same on every rank.
DEFAULT_CHUNKS = 2


def _dist():
    import torch.distributed as dist
    return dist


_DTYPE_CODES = {"torch.float64": 1, "torch.float32": 2, "torch.float16": 3, "torch.bfloat16": 4}


def block_sizes(X_loc, group=None, sizes=None):
    """Per-rank curve-block sizes of a column-sharded data set, with the consistency every exchange below relies on.

    One small all-gather of (n_loc, T, dtype): ranks that disagree on the number of timepoints or the dtype would make
    the all-to-all / all-gather split sizes disagree, which hangs or fails deep inside RCCL -- here it is a ValueError on
    every rank.  Callers that know the partition pass `sizes` and skip the exchange (and its host synchronisation).
    """
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    if sizes is not None:
        sizes = [int(v) for v in sizes]
        if len(sizes) != world or sizes[dist.get_rank(group)] != X_loc.shape[1]:
            raise ValueError(f"sizes={sizes} does not describe this rank's block of {X_loc.shape[1]} curves")
        return sizes
    mine = torch.tensor([X_loc.shape[1], X_loc.shape[0], _DTYPE_CODES.get(str(X_loc.dtype), 0)], dtype=torch.int64,
                        device=X_loc.device)
    allv = torch.zeros(world * 3, dtype=torch.int64, device=X_loc.device)
    dist.all_gather_into_tensor(allv, mine, group=group)
    allv = allv.view(world, 3).cpu().tolist()
    if len({(int(t), int(c)) for _, t, c in allv}) != 1:
        raise ValueError("ranks disagree on the number of timepoints or the dtype of their curve blocks: "
                         + ", ".join(f"rank {r}: T={int(t)} dtype code {int(c)}" for r, (_, t, c) in enumerate(allv)))
    return [int(v[0]) for v in allv]


def _default_compute(X_all, targets, J, algo):
    """HIP path: totals of the contiguous target block within the gathered matrix (device tensors)."""
    return engine.mbd_counts_range(X_all, int(targets[0]), len(targets), J=J, algo=algo, return_tensor=True)


def gather_curve_blocks(X_loc, group=None, sizes=None, rows=None, async_op=False):
    """All-gather of per-rank curve blocks.

    X_loc: [T, n_loc] time-major tensor (n_loc may differ between ranks).
    sizes: per-rank block sizes when the caller already knows the partition (skips the size exchange).
    rows:  (lo, hi): only these timepoints of every block travel (the caller pipelines row chunks against the compute:
           band totals are sums over timepoints, so each chunk is computed on as it arrives).
    Returns (X_all [T or hi - lo, n] with rank blocks side by side, offsets [world+1]); with async_op=True
    ((work, finish), offsets): `finish()` waits for the exchange (the current stream waits) and lays the rows out.
    """
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    sizes = block_sizes(X_loc, group, sizes)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    if rows is not None:
        X_loc = X_loc[int(rows[0]):int(rows[1])]
    T, n_loc = X_loc.shape
    if min(sizes) == max(sizes):
        # equal blocks: gather the time-major blocks as they lie, then one strided copy lays the rows out
        recv = torch.empty((world * T, n_loc), dtype=X_loc.dtype, device=X_loc.device)
        send = X_loc.contiguous()
        work = dist.all_gather_into_tensor(recv, send, group=group, async_op=async_op)

        def finish():
            if work is not None:
                work.wait()
            X_all = torch.empty((T, world * n_loc), dtype=X_loc.dtype, device=X_loc.device)
            X_all.view(T, world, n_loc).copy_(recv.view(world, T, n_loc).permute(1, 0, 2))
            _ = send                                         # the send buffer lives as long as this closure
            return X_all
    else:
        nmax = max(sizes)
        # blocks travel curve-major ([n_loc, T] rows are whole curves) so a ragged tail is plain padding
        send = torch.zeros((nmax, T), dtype=X_loc.dtype, device=X_loc.device)
        send[:n_loc].copy_(X_loc.t())
        recv = torch.empty((world * nmax, T), dtype=X_loc.dtype, device=X_loc.device)
        work = dist.all_gather_into_tensor(recv, send, group=group, async_op=async_op)

        def finish():
            if work is not None:
                work.wait()
            rv = recv.view(world, nmax, T)
            X_all = torch.empty((T, int(offsets[-1])), dtype=X_loc.dtype, device=X_loc.device)
            for r in range(world):
                X_all[:, offsets[r]:offsets[r + 1]].copy_(rv[r, :sizes[r]].t())
            _ = send
            return X_all
    if async_op:
        return (work, finish), offsets
    return finish(), offsets


def sharded_mbd_counts_targets(X_loc, J=2, algo="auto", group=None, sizes=None, _compute=None, chunks=None):
    """Target-sharded totals (north_star's wording: all-gather of curve blocks, every rank computes its own targets), with the
    all-gather cut into `chunks` row chunks: band totals are sums over timepoints, so chunk k is computed on (current stream)
    while chunk k + 1 travels (RCCL's stream), and the partial totals add up.  Returns (int64 [n_loc, J-1], offsets)."""
    import torch
    dist = _dist()
    rank = dist.get_rank(group)
    sizes = block_sizes(X_loc, group, sizes)
    T = int(X_loc.shape[0])
    K = max(1, min(int(chunks if chunks is not None else DEFAULT_CHUNKS), max(T, 1)))
    bounds = _chunk_bounds(T, K)
    compute = _compute or _default_compute
    total, offsets = None, None
    pending, offsets = gather_curve_blocks(X_loc, group, sizes, rows=(bounds[0], bounds[1]), async_op=True)
    for k in range(K):
        nxt = gather_curve_blocks(X_loc, group, sizes, rows=(bounds[k + 1], bounds[k + 2]), async_op=True)[0] if k + 1 < K else None
        X_all = pending[1]()
        pending = nxt
        if X_all.shape[0] == 0:
            continue
        targets = np.arange(offsets[rank], offsets[rank + 1], dtype=np.int64)
        part = compute(X_all, targets, J, algo)
        if not isinstance(part, torch.Tensor):
            part = torch.as_tensor(np.asarray(part), device=X_loc.device)
        total = part if total is None else total + part
    if total is None:
        total = torch.zeros((int(sizes[rank]), J - 1), dtype=torch.int64, device=X_loc.device)
    return total, offsets


def _time_slices(T, world):
    """Contiguous split of the T timepoints over the ranks (first T % world ranks get one more)."""
    base, extra = divmod(T, world)
    cnt = [base + (1 if r < extra else 0) for r in range(world)]
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    return cnt, off


def _chunk_bounds(cnt, K):
    """Row range [lo, hi) of sub-slice k inside a time slice of cnt rows, for k = 0..K-1 (as even as possible)."""
    return [(cnt * k) // K for k in range(K + 1)]


def exchange_to_time_slices(X_loc, sizes, group=None, chunk=0, chunks=1, async_op=False):
    """All-to-all that turns curve blocks into time slices.

    In: this rank's curves X_loc [T, n_loc] (time-major).  Out: X_rows [T_me, n] = ALL curves at this
    rank's timepoints (rank blocks side by side), plus the column offsets.  Each rank sends only
    (world-1)/world of its block and receives the same amount: 1/world of an all-gather's traffic.

    chunks > 1: only sub-slice `chunk` of every rank's time slice travels (the caller pipelines the sub-slices
    against the compute).  async_op=True returns (work, finish) instead: `finish()` waits for the exchange and
    lays the rows out.
    """
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    T, n_loc = X_loc.shape
    cnt, toff = _time_slices(T, world)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    bounds = [_chunk_bounds(cnt[s], chunks) for s in range(world)]
    rows_to = [bounds[s][chunk + 1] - bounds[s][chunk] for s in range(world)]     # rows this rank sends to s
    t_me = rows_to[rank]
    in_split = [rows_to[s] * n_loc for s in range(world)]
    out_split = [t_me * int(sizes[r]) for r in range(world)]
    recv = torch.empty(int(sum(out_split)), dtype=X_loc.dtype, device=X_loc.device)
    X_loc = X_loc.contiguous()
    if chunks > 1 and dist.get_backend(group) == "nccl":
        # the rows for destination s are one contiguous range of this rank's (time-major) block: RCCL takes them where
        # they lie, one send per destination, no staging copy
        send = [X_loc[toff[s] + bounds[s][chunk]: toff[s] + bounds[s][chunk + 1]].reshape(-1) for s in range(world)]
        outs = list(torch.split(recv, out_split))
        work = dist.all_to_all(outs, send, group=group, async_op=async_op)
    else:
        if chunks == 1:
            send = X_loc.view(-1)                            # rows of destination s are contiguous
        else:                                                # gloo: stage the sub-slices of all destinations back to back
            send = torch.cat([X_loc[toff[s] + bounds[s][chunk]: toff[s] + bounds[s][chunk + 1]] for s in range(world)],
                             dim=0).contiguous().view(-1)
        work = dist.all_to_all_single(recv, send, output_split_sizes=out_split, input_split_sizes=in_split, group=group,
                                      async_op=async_op)
    n = int(offsets[-1])

    def finish():
        if work is not None:
            work.wait()                                      # the current stream waits for the exchange
        X_rows = torch.empty((t_me, n), dtype=X_loc.dtype, device=X_loc.device)
        if min(sizes) == max(sizes):
            X_rows.view(t_me, world, n_loc).copy_(recv.view(world, t_me, n_loc).permute(1, 0, 2))
        else:
            pos = 0
            for r in range(world):
                blk = recv[pos:pos + out_split[r]].view(t_me, int(sizes[r]))
                X_rows[:, offsets[r]:offsets[r + 1]].copy_(blk)
                pos += out_split[r]
        _ = send                                             # the staging buffer lives as long as this closure
        return X_rows

    if async_op:
        return work, finish
    return finish(), offsets


def _default_compute_all(X_rows, J, algo):
    """HIP path: totals of every curve over the given rows (device tensors)."""
    return engine.mbd_counts(X_rows, None, J=J, algo=algo, return_tensor=True)


def sharded_mbd_counts_time(X_loc, J=2, algo="auto", group=None, sizes=None, _compute_all=None, chunks=None,
                            _force_exchange=False):
    """Time-sharded form of the same totals (the right decomposition for the rank kernels).

    The rank formulation sorts whole rows, so splitting the TARGETS would make every GPU sort every
    row.  Splitting the TIMEPOINTS divides the sort work: an all-to-all hands rank s all curves at its
    timepoints, it computes per-curve partial totals over those rows, and a reduce-scatter (int64 sum)
    returns to every rank the totals of its own curves.  Integer sums commute: results are identical
    to the target-sharded and single-GPU paths.
    """
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    T, n_loc = X_loc.shape
    compute = _compute_all or _default_compute_all
    if world == 1 and not _force_exchange:
        # a single rank owns every curve at every timepoint: there is nothing to exchange and nothing to reduce
        # (_force_exchange: tests run the collectives on one rank)
        pk = compute(X_loc.contiguous(), J, algo)
        return pk if isinstance(pk, torch.Tensor) else torch.as_tensor(np.asarray(pk), device=X_loc.device)
    sizes = block_sizes(X_loc, group, sizes)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offsets[-1])
    # The exchange is cut into K sub-slices of every rank's time slice; sub-slice k+1 travels (RCCL's stream) while
    # the rows of sub-slice k are ranked (current stream).  K is the same on every rank: it depends on T and the
    # world size only.
    cnt, _ = _time_slices(T, world)
    K = int(chunks) if chunks is not None else DEFAULT_CHUNKS
    K = max(1, min(K, min(cnt)))
    part = None
    pending = exchange_to_time_slices(X_loc, sizes, group, 0, K, async_op=True)
    for k in range(K):
        nxt = exchange_to_time_slices(X_loc, sizes, group, k + 1, K, async_op=True) if k + 1 < K else None
        X_rows = pending[1]()
        pending = nxt
        if X_rows.shape[0] > 0:
            pk = compute(X_rows, J, algo)
            if not isinstance(pk, torch.Tensor):
                pk = torch.as_tensor(np.asarray(pk), device=X_loc.device)
            part = pk if part is None else part + pk
    if part is None:
        part = torch.zeros((n, J - 1), dtype=torch.int64, device=X_loc.device)
    part = part.contiguous()
    if min(sizes) == max(sizes) and dist.get_backend(group) == "nccl":
        out = torch.empty((n_loc, J - 1), dtype=torch.int64, device=X_loc.device)
        dist.reduce_scatter_tensor(out, part, op=dist.ReduceOp.SUM, group=group)
        return out
    dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)     # ragged blocks / gloo: reduce, then slice
    return part[offsets[rank]:offsets[rank + 1]].clone()


# The three constants of mode_cost_model, in one place.  UNVALIDATED BEYOND ONE GPU: the ranking rates are measured single-GPU
# rates (keys ranked per second, MI355X) and so is the pairwise kernel's (target x curve x timepoint
# triples per second); the link rate is an assumption (one xGMI link per peer, ~50 GB/s per direction sustained) that no run
# on more than one GPU has met yet -- bench.py prints the model's predicted step beside the measured one (`mode` object of
# its line) so that the first multi-GPU run calibrates them.
_RANK_KEYS_PER_S = ((16384, 2.1e11), (32768, 1.0e11), (1 << 62, 9.4e10))
_PAIR_TRIPLES_PER_S = 8.3e12
_LINK_BYTES_PER_S = 50e9


def mode_cost_model(T, n_loc, world, J=2, algo="auto"):
    """Estimated seconds per call of the two decompositions on `world` GPUs of one node, exchange included.

    time:    all-to-all (every rank sends (world-1)/world of its block, one link per peer) + ranking T/world rows of all n
             curves + reduce-scatter of the int64 totals.
    targets: all-gather of the curve blocks (every rank receives world-1 blocks, one link per peer) + either ranking ALL T
             rows (the rank kernels rank whole rows whoever the targets are) or the pairwise kernel on its own targets.
    Returns {"time": s, "targets": s, "choice": ...}; the choice is what mode="auto" takes."""
    n = n_loc * world
    rank_ok = J <= 3 and algo != "pairwise"

    def rank_s(rows):
        rate = next(r for cap, r in _RANK_KEYS_PER_S if n <= cap)
        return rows * n / rate
    peers = max(1, world - 1)
    t_time = None
    if rank_ok and T >= world:
        xch = 0.0 if world == 1 else (8.0 * T * n_loc / world) / _LINK_BYTES_PER_S      # per link, links in parallel
        red = 0.0 if world == 1 else (8.0 * n * (J - 1) / world) / _LINK_BYTES_PER_S
        t_time = xch + rank_s((T + world - 1) // world) + red
    gat = 0.0 if world == 1 else (8.0 * T * n_loc) / _LINK_BYTES_PER_S                  # one block per link
    t_tg = gat + (rank_s(T) if rank_ok else T * float(n) * n_loc / _PAIR_TRIPLES_PER_S)
    choice = "time" if (t_time is not None and t_time <= t_tg) else "targets"
    return {"time": t_time, "targets": t_tg, "choice": choice, "peers": peers}


def sharded_mbd_counts(X_loc, J=2, algo="auto", group=None, gather_result=False, sizes=None, mode="auto",
                       _compute=None, _compute_all=None, _force_exchange=False, chunks=None):
    """MBD containment totals of this rank's curves against the union of all ranks' curves.

    mode: "targets" (all-gather of curve blocks, each rank computes its own targets: the pairwise
    kernel's natural split), "time" (all-to-all + reduce-scatter: the rank kernels' natural split, see
    sharded_mbd_counts_time) or "auto" (the cheaper one by mode_cost_model; it needs the block sizes of all ranks: one small
    all-gather with a host synchronisation per call unless the caller passes `sizes`).
    chunks: pieces the exchange is cut into and pipelined against the compute (default DEFAULT_CHUNKS; the same on every rank).

    Returns int64 [n_loc, J-1] (device of X_loc), or with gather_result=True the full
    [n, J-1] array on every rank in global curve order.
    `_compute` is a test hook (CPU/gloo tests inject a checker); the product path is the HIP engine.
    """
    import torch
    dist = _dist()
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    if mode == "auto":
        # Every rank must take the same branch (the two modes run different collectives): the model sees only what all ranks
        # share -- T, the world size and the LARGEST block (one small all-gather of the sizes, also the consistency check;
        # skipped when the caller passes `sizes`) -- never this rank's own block size.
        sizes = block_sizes(X_loc, group, sizes)
        mode = mode_cost_model(int(X_loc.shape[0]), int(max(sizes)), world, J, algo)["choice"]
    if mode == "time":
        sizes = block_sizes(X_loc, group, sizes)
        local = sharded_mbd_counts_time(X_loc, J=J, algo=algo, group=group, sizes=sizes, _compute_all=_compute_all,
                                        _force_exchange=_force_exchange, chunks=chunks)
        offsets = np.concatenate([[0], np.cumsum([int(v) for v in sizes])]).astype(np.int64)
    else:
        local, offsets = sharded_mbd_counts_targets(X_loc, J=J, algo=algo, group=group, sizes=sizes, _compute=_compute, chunks=chunks)
    if not gather_result:
        return local
    sizes = np.diff(offsets)
    nmax = int(sizes.max())
    send = torch.zeros((nmax, J - 1), dtype=torch.int64, device=X_loc.device)
    send[:local.shape[0]].copy_(local)
    recv = torch.empty((world * nmax, J - 1), dtype=torch.int64, device=X_loc.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, nmax, J - 1)
    return torch.cat([recv[r, :int(sizes[r])] for r in range(world)], dim=0)


def _default_compute_strict(X_all, targets, J):
    """HIP path: strict band depth totals of the given targets within the gathered matrix."""
    import torch
    return torch.as_tensor(engine.bd_strict_counts(X_all, targets, J=J), device=X_all.device)


def sharded_bd_strict_counts(X_loc, J=2, group=None, sizes=None, _compute=None):
    """Strict band depth (relax=False) totals of this rank's curves against the union of all ranks' curves.

    The strict kernels test every pair of other curves against one target over all T timepoints (K3, O(m n^2 T / 32)):
    the work is per TARGET, so the targets are what is split -- the reference's outer loop (_functional.py:74-75) carries
    no state.  One all-gather of the curve blocks, then every rank runs sd_bd_strict_j_counts for its own block of
    targets.  Returns int64 [n_loc, J-1]."""
    import torch
    dist = _dist()
    rank = dist.get_rank(group)
    X_all, offsets = gather_curve_blocks(X_loc, group, sizes)
    targets = np.arange(offsets[rank], offsets[rank + 1], dtype=np.int64)
    local = (_compute or _default_compute_strict)(X_all, targets, J)
    if not isinstance(local, torch.Tensor):
        local = torch.as_tensor(np.asarray(local), device=X_loc.device)
    return local.reshape(len(targets), J - 1)


def gather_point_blocks(P_loc, group=None):
    """All-gather of per-rank point blocks [n_loc, d] (ragged n_loc allowed).  Returns (P_all [n, d], offsets)."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    n_loc, d = P_loc.shape
    szt = torch.zeros(world, dtype=torch.int64, device=P_loc.device)
    dist.all_gather_into_tensor(szt, torch.tensor([n_loc], dtype=torch.int64, device=P_loc.device), group=group)
    sizes = [int(v) for v in szt.cpu().tolist()]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    nmax = max(sizes)
    send = torch.zeros((nmax, d), dtype=P_loc.dtype, device=P_loc.device)
    send[:n_loc].copy_(P_loc)
    recv = torch.empty((world * nmax, d), dtype=P_loc.dtype, device=P_loc.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    recv = recv.view(world, nmax, d)
    P_all = torch.cat([recv[r, :sizes[r]] for r in range(world)], dim=0).contiguous()
    return P_all, offsets


def _default_pointcloud_compute(P_all, targets, containment, samples, seed):
    """HIP path (device tensors in, numpy out)."""
    if containment == "l1":
        return engine.l1_depth(P_all, targets)
    return engine.pointcloud_simplex_counts(P_all, targets, samples=samples, seed=seed)


def sharded_pointcloud(P_loc, containment="simplex", samples=None, seed=0, group=None, _compute=None):
    """Point-cloud depth quantities of this rank's points against the union of all ranks' points.

    The reference's target loop (_pointcloud.py:45) carries no state: rank r owns a block of points, the blocks are
    all-gathered once (RCCL over xGMI; a 10^6 x 3 cloud is 24 MB) and every rank runs the kernel for its own targets.
    containment "simplex": containment counts (int64; exhaustive, or `samples` seeded subsets per point -- the subset
    draws are keyed by the GLOBAL point index, so the result does not depend on the sharding); "l1": L1 depths (fp64).
    Returns (values of the local points, n_total).  `_compute` is a test hook (CPU/gloo tests inject the oracle).
    """
    dist = _dist()
    rank = dist.get_rank(group)
    P_all, offsets = gather_point_blocks(P_loc, group)
    targets = np.arange(offsets[rank], offsets[rank + 1], dtype=np.int64)
    compute = _compute or _default_pointcloud_compute
    return compute(P_all, targets, containment, samples, seed), int(offsets[-1])


def sharded_functional_depth(df_local, J=2, relax=True, algo="auto", group=None, mode="auto", _compute=None,
                             _compute_all=None, _compute_strict=None):
    """FunctionalDepth over curves sharded by column blocks: returns this rank's depth Series.

    Normalisation as the reference (_functional.py:229,253): relax -> / T / C(n, j), strict -> / C(n, j), with n the
    GLOBAL number of curves.  relax=True: sharded_mbd_counts; relax=False: sharded_bd_strict_counts (targets split).
    """
    import pandas as pd
    import torch
    from scipy.special import binom
    X = np.ascontiguousarray(df_local.to_numpy(dtype=np.float64))
    hooks = (_compute, _compute_all, _compute_strict)
    dev = engine._device() if all(h is None for h in hooks) else torch.device("cpu")
    X_loc = torch.from_numpy(X).to(dev)
    sizes = block_sizes(X_loc, group)
    n, T = int(sum(sizes)), X.shape[0]
    if relax:
        counts = sharded_mbd_counts(X_loc, J=J, algo=algo, group=group, mode=mode, sizes=sizes, _compute=_compute,
                                    _compute_all=_compute_all).cpu().numpy().astype(np.float64) / T
    else:
        counts = sharded_bd_strict_counts(X_loc, J=J, group=group, sizes=sizes,
                                          _compute=_compute_strict).cpu().numpy().astype(np.float64)
    depth = np.zeros(X.shape[1])
    for j in range(2, J + 1):
        depth += counts[:, j - 2] / binom(n, j)
    return pd.Series(index=df_local.columns, data=depth)
