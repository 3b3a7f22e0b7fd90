"""Argument validation and the degeneracy exception of the depth factories.

Mirrors statdepth/depth/calculations/_helper.py: `DepthDegeneracy` (:13-14) and the
checks of `_handle_depth_errors` (:34-107), in the same order with the same
exception types, so callers' try/except blocks keep working.
"""
from typing import Callable

import numpy as np


class DepthDegeneracy(Exception):
    """Depth is not well defined for this input (degenerate simplices, empty blocks ...)."""


def _handle_depth_errors(data, J, containment, relax, deep_check) -> None:
    # type checks (_helper.py:59-72)
    if not isinstance(data, list):
        raise ValueError('data must be passed as a list.')
    if not isinstance(J, int):
        raise ValueError('J must be an integer.')
    if not (isinstance(containment, str) or isinstance(containment, Callable)):
        raise ValueError('containment must be of type str or Callable.')
    if not isinstance(deep_check, bool):
        raise ValueError('deep_check must be of type bool.')
    if not isinstance(relax, bool):
        raise ValueError('relax must be of type bool')
    # J = 0, 1 make no sense (:75-76)
    if J < 2:
        raise ValueError('Parameter J must be greater than or equal to 2.')
    if len(data) == 0:
        raise ValueError('No data passed.')
    # NB the univariate branch compares J with the number of ROWS (timepoints), as the reference does (:83)
    if len(data) == 1 and J >= len(data[0]) or len(data) > 1 and J >= len(data):
        raise ValueError('Parameter J must be less than the number of observations.')
    if len(data) > 1 and containment == 'r2':
        raise ValueError('containment argument \'r2\' is invalid for multivariate data. '
                         'Use one of [\'r2_enum\', \'simplex \'] or a passed containment method. ')
    # fewer than d + 2 functions: every simplex is degenerate (:92-93).  The reference means to raise
    # DepthDegeneracy here (its message formatting itself fails with TypeError); we raise what it means.
    if len(data) < data[0].shape[1] + 2 and containment == 'simplex':
        raise DepthDegeneracy(f'Error: Need at least {data[0].shape[1] + 2} functions to form non-degenerate '
                              f'simplices in {data[0].shape[1]} dimensional space. Only have {len(data)}.')
    if deep_check:   # (:95-107)
        indices = []
        for df in data:
            indices.append(df.index)
            df = df.infer_objects()
            for col in df:
                if not np.issubdtype(df[col].dtype, np.number):
                    raise ValueError('DataFrame must only contain numeric dtypes.')
        if not all([all(indices[0] == i) for i in indices]):
            raise ValueError('DataFrames indices must be the same')
