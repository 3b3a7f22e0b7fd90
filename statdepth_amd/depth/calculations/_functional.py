"""Functional (band) depth drivers on the HIP engine.

Mirrors statdepth/depth/calculations/_functional.py: `_functionaldepth` (:17-97),
`_samplefunctionaldepth` (:99-196).  The per-target subset enumeration of
`_univariate_band_depth` (:198-255) and `_simplex_depth` (:257-286) is replaced by
calls into libstatdepth_hip (statdepth_amd.engine); the float normalisers stay here,
in fp64, written as the reference writes them.
"""
import math
from itertools import combinations
from typing import List, Union

import numpy as np
import pandas as pd
from scipy.special import binom

from ... import engine
from ..._native import SD_ERR_OVERFLOW, StatdepthHipError
from ._containment import _select_containment
from ._helper import DepthDegeneracy, _handle_depth_errors

__all__ = ['_functionaldepth', '_samplefunctionaldepth']


# tests switch this off to compare the one-launch strict estimator with the block-by-block evaluation
_BATCH_STRICT_BLOCKS = True

def _positions(df: pd.DataFrame, labels) -> np.ndarray:
    """Column positions of `labels` (the reference addresses targets by label, :232)."""
    pos = df.columns.get_indexer(list(labels))
    if (pos < 0).any():
        missing = [l for l, p in zip(labels, pos) if p < 0]
        raise KeyError(f'{missing} not in columns')
    return pos.astype(np.int64)


def _require_unique_labels(df: pd.DataFrame) -> None:
    """Curves are addressed by column label (:232,248).  With duplicated labels the reference's label tuples collapse
    through a set (_helper.py:32) and `.loc` returns every column carrying a label, so bands silently merge curves, and
    any target whose own label is duplicated -- hence the default `to_compute=None` -- dies with a TypeError inside
    `_r2_containment` (tests/golden/g11_duplabels.json records this).  That accident is not reproduced: duplicated
    labels are refused up front."""
    if not df.columns.is_unique:
        dup = list(df.columns[df.columns.duplicated()].unique())
        raise ValueError(f'column labels must be unique to address curves; duplicated: {dup}')


def _univariate_depths(df: pd.DataFrame, cols, J: int, relax: bool, device=None, algo='auto') -> np.ndarray:
    """Band depth of columns `cols` of `df` (rows = timepoints): sum_j S_nj / binom(n, j).

    n = number of columns INCLUDING the target (:229) while bands come from the other
    n-1 (:235,243); S_nj = (sum_t contained j-bands)/T for relax (_containment.py:80
    `c/T`) or the number of j-bands containing the curve at every t (`c//T`).
    """
    X = df.to_numpy(dtype=np.float64, copy=False)      # keeps the frame's memory layout
    T, n = X.shape
    tg = _positions(df, cols)
    if relax:
        try:
            counts = engine.mbd_counts(X, tg, J=J, algo=algo, device=device).astype(np.float64) / T
        except StatdepthHipError as e:
            if e.code != SD_ERR_OVERFLOW:
                raise
            # T * C(n-1, J) beyond int64 (J >= 4 with ~10^5 curves): two-limb totals, exact big-integer normalisation
            wide = engine.mbd_counts_wide(X, tg, J=J, algo=algo, device=device)
            depth = np.zeros(len(tg), dtype=np.float64)
            for j in range(2, J + 1):
                den = T * math.comb(n, j)
                depth += np.array([int(v) / den for v in wide[:, j - 2]], dtype=np.float64)     # (:253), exact quotient
            return depth
    else:
        if J > 4:
            raise NotImplementedError('strict band depth (relax=False) is implemented for J <= 4')
        counts = engine.bd_strict_counts(X, tg, J=J, device=device).astype(np.float64)
    depth = np.zeros(len(tg), dtype=np.float64)
    for j in range(2, J + 1):
        depth += counts[:, j - 2] / binom(n, j)        # (:253)
    return depth


def _callable_band_depth(data: pd.DataFrame, curve, relax: bool, containment, J: int) -> float:
    """Generic enumerator for a user-supplied containment callable (:228-255).

    Not the hot path: arbitrary Python cannot run on the GPU, so the plug-in protocol
    of docs/index.md:124-148 is honoured on the host, one call per subset.
    """
    band_depth = 0
    n = data.shape[1]
    curvedata = data.loc[:, curve]
    data = data.drop(curve, axis=1)
    for j in range(2, J + 1):
        S_nj = 0
        for sequence in combinations(list(data.columns), j):
            S_nj += containment(data=data.loc[:, list(sequence)], curve=curvedata, relax=relax)
        band_depth += S_nj / binom(n, j)
    return band_depth


def _curves_tensor(data: List[pd.DataFrame]) -> np.ndarray:
    """list of n frames (T x d) -> (n, T, d) fp64."""
    return np.stack([np.asarray(df.to_numpy(dtype=np.float64)) for df in data])


def _componentwise_band_depth(data: List[pd.DataFrame], to_compute, J: int, relax: bool, device=None) -> pd.Series:
    """Multivariate band depth with the 'r2_enum' containment the reference declares and leaves unimplemented
    (_containment.py:83-103: every component treated as a real-valued function, contained iff all components are).

    Built as `_univariate_band_depth` (:238-253) with that predicate: bands from j-subsets of the n-1 other curves,
    sum_j S_nj / binom(n, j) with n INCLUDING the target (:229,253) -- so one feature (d = 1) gives the univariate depth.
    relax=True, J = 2: one launch (sd_multi_band_counts, pairs counted through the 3^d state classes).  relax=False
    (contained at every timepoint in every component) is the strict univariate depth of the T*d component series.
    """
    f = [i for i in range(len(data))] if to_compute is None else to_compute
    P = _curves_tensor(data)
    n, T, d = P.shape
    tg = np.asarray(list(f), dtype=np.int64)
    depth = np.zeros(len(tg), dtype=np.float64)
    if relax:
        if J != 2:
            raise NotImplementedError("'r2_enum' with relax=True is implemented for J = 2")
        depth += engine.multi_band_counts(P, tg, device=device).astype(np.float64) / T / binom(n, 2)
    else:
        if J > 4:
            raise NotImplementedError('strict band depth (relax=False) is implemented for J <= 4')
        X = P.reshape(n, T * d).T                        # rows = (timepoint, feature), columns = curves; no copy
        counts = engine.bd_strict_counts(X, tg, J=J, device=device).astype(np.float64)
        for j in range(2, J + 1):
            depth += counts[:, j - 2] / binom(n, j)
    return pd.Series(index=f, data=depth)


def _functionaldepth(data: List[pd.DataFrame], to_compute: Union[list, pd.Index] = None, J=2, containment='r2',
                     relax=False, deep_check=False, quiet=True, device=None, algo='auto') -> pd.Series:
    _handle_depth_errors(data=data, J=J, containment=containment, relax=relax, deep_check=deep_check)
    cdef = _select_containment(containment=containment)

    if len(data) == 1:                                   # real-valued case by assumption (:60-61)
        if cdef == 'simplex':
            cdef = 'r2'                                  # (:62-63)
        df = data[0]
        _require_unique_labels(df)
        cols = df.columns if to_compute is None else to_compute     # (:68-71)
        if cdef == 'r2':
            depths = _univariate_depths(df, cols, J, relax, device=device, algo=algo)
        elif cdef == 'r2_enum':
            raise NotImplementedError                    # _containment.py:103
        else:
            depths = [_callable_band_depth(df, col, relax, cdef, J) for col in cols]
        return pd.Series(index=cols, data=depths)        # (:78)

    # multivariate case (:79-95)
    if cdef == 'simplex':
        f = [i for i in range(len(data))] if to_compute is None else to_compute
        P = _curves_tensor(data)
        n, T, d = P.shape
        counts = engine.multi_simplex_counts(P, np.asarray(list(f), dtype=np.int64), relax=relax,
                                             device=device).astype(np.float64)
        if relax:
            counts = counts / T                          # _containment.py:136
        depths = counts / binom(n - 1, d + 1)            # (:278,286): n there = number of OTHERS
        return pd.Series(index=f, data=depths)
    if cdef == 'r2_enum':
        return _componentwise_band_depth(data, to_compute, J, relax, device)
    raise NotImplementedError('custom containment callables are only supported for univariate data')


def _samplefunctionaldepth(data: List[pd.DataFrame], K: int, to_compute: Union[list, pd.Index] = None, J=2,
                           containment='r2', relax=False, deep_check=False, quiet=True, device=None,
                           algo='auto') -> pd.Series:
    """K-block sampled band depth (:99-196).

    Block selection is host logic and consumes the global numpy RNG exactly as the
    reference does (`df.sample(n=ss, axis=1)`, :176), so `np.random.seed` reproduces the
    reference's blocks; each block's depth is one device call.
    """
    samples = []
    _handle_depth_errors(data=data, J=J, containment=containment, relax=relax, deep_check=deep_check)
    cdef = _select_containment(containment=containment)
    if len(data) != 1:
        return pd.Series(dtype=np.float64)               # reference stub returns an empty list (:187-196)

    df = data[0]
    _require_unique_labels(df)
    cols = df.columns if to_compute is None else to_compute
    orig = df.loc[:, cols]                               # (:161)
    ss = df.shape[1] // K                                # (:162)
    if ss == 0:
        raise DepthDegeneracy(f'Block size {K} is too large, not enough functions to sample.')
    if cdef == 'simplex':
        cdef = 'r2'
    if cdef == 'r2_enum':
        raise NotImplementedError

    # every (target, block) pair in one launch: relax=True always; the reference's default relax=False for J = 2 (the
    # blocks' masks in LDS, or in the workspace when n / K runs into the thousands)
    batched = cdef == 'r2' and (relax or (J == 2 and _BATCH_STRICT_BLOCKS))
    blocks, block_targets = [], []           # column positions (in data[0]) of every block, in draw order
    full = data[0]
    for col in orig.columns:
        depths = []
        for _ in range(K):
            t = df.sample(n=ss, axis=1)                  # (:176) global numpy RNG
            df = df.drop(t.columns, axis=1)              # (:177) without replacement across blocks
            members = list(t.columns)
            if col not in members:
                members.append(col)                      # (:178) force the target into the block
            if batched:
                blocks.append(full.columns.get_indexer(members))
                block_targets.append(full.columns.get_loc(col))
                continue
            t = t.copy()
            t.loc[:, col] = orig.loc[:, col]
            if cdef == 'r2':
                depths.append(_univariate_depths(t, [col], J, relax, device=device, algo='pairwise')[0])
            else:
                depths.append(_callable_band_depth(t, col, relax, cdef, J))
        if not batched:
            samples.append(np.mean(depths))              # (:182)
        df = orig.copy()                                 # (:183) -- the pool shrinks to `cols`, as in the reference
    if batched:
        # every (target, block) pair of the estimator in ONE launch
        width = max(len(b) for b in blocks)
        mem = np.full((len(blocks), width), -1, dtype=np.int32)
        for i, b in enumerate(blocks):
            mem[i, :len(b)] = b
        X = full.to_numpy(dtype=np.float64)
        T = X.shape[0]
        if relax:
            counts = engine.mbd_subset_counts(X, mem, np.asarray(block_targets, dtype=np.int32), J=J,
                                              device=device).astype(np.float64) / T
        else:
            counts = engine.bd_strict_subset_counts(X, mem, np.asarray(block_targets, dtype=np.int32),
                                                    device=device).astype(np.float64)[:, None]
        sizes = np.array([len(b) for b in blocks], dtype=np.float64)
        depth = np.zeros(len(blocks))
        for j in range(2, J + 1):
            depth += counts[:, j - 2] / binom(sizes, j)  # (:253) n = block size including the target
        samples = [np.mean(depth[i * K:(i + 1) * K]) for i in range(len(orig.columns))]
    return pd.Series(index=df.columns, data=samples)     # (:186)
