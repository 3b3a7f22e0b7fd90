"""Point-cloud depth drivers on the HIP engine.

Mirrors statdepth/depth/calculations/_pointcloud.py: `_pointwisedepth` (:14-66),
`_samplepointwisedepth` (:68-123), `_L1_depth` (:125-150).  Mahalanobis and Oja depth
(:152-205) are outside the containment/count hot path and are not provided.
"""
from typing import Union

import numpy as np
import pandas as pd
from scipy.special import binom

from ... import engine

__all__ = ['_pointwisedepth', '_samplepointwisedepth']


def _row_positions(data: pd.DataFrame, labels) -> np.ndarray:
    pos = data.index.get_indexer(list(labels))
    if (pos < 0).any():
        missing = [l for l, p in zip(labels, pos) if p < 0]
        raise KeyError(f'{missing} not in index')
    return pos.astype(np.int64)


def _pointwisedepth(data: pd.DataFrame, to_compute: Union[list, pd.Index] = None, containment='simplex',
                    quiet=True, device=None) -> pd.Series:
    n, d = data.shape
    if to_compute is None:
        to_compute = data.index                          # (:41-42)
    if containment == 'simplex':
        P = data.to_numpy(dtype=np.float64)
        counts = engine.pointcloud_simplex_counts(P, _row_positions(data, to_compute), device=device)
        depths = counts.astype(np.float64) / binom(n, d + 1)     # n INCLUDES the point (:38,56)
        return pd.Series(index=to_compute, data=depths)
    elif containment == 'l1':
        P = data.to_numpy(dtype=np.float64)
        depths = engine.l1_depth(P, _row_positions(data, to_compute), device=device)
        return pd.Series(index=to_compute, data=depths)          # (:150)
    elif containment in ('mahalanobis', 'oja'):
        raise NotImplementedError(f'{containment} depth is outside the band-depth hot path this engine covers')
    else:
        raise ValueError(f'{containment} is not a valid containment measure. ')   # (:63-64)


def _samplepointwisedepth(data: pd.DataFrame, to_compute: pd.Index = None, K=2, containment='simplex',
                          quiet=True, device=None) -> pd.Series:
    """K-block sampled point-cloud depth (:68-123).

    Same sampling rule and RNG consumption as the reference: `ss = n // K` (:107) and,
    per point, `ss` repetitions (:113 -- the loop bound is ss, not K) of a `data.sample(n=ss)`
    draw (:114) with the point appended when missing (:117-118; the reference's
    `DataFrame.append` no longer exists in pandas >= 2, `pd.concat` is its definition).
    """
    if K == 1:
        return _pointwisedepth(data=data, to_compute=to_compute, containment=containment, device=device)
    n, d = data.shape
    depths = []
    if to_compute is None:
        to_compute = data.index
    ss = n // K
    for time in to_compute:
        cd = []
        for _ in range(ss):
            sdata = data.sample(n=ss, axis=0)
            if time not in sdata.index:
                sdata = pd.concat([sdata, data.loc[[time], :]])
            cd.append(_pointwisedepth(data=sdata, to_compute=[time], containment=containment, device=device))
        depths.append(np.mean(cd))
    return pd.Series(index=to_compute, data=depths)
