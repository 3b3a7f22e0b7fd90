"""Argument validation and the degeneracy exception of the depth factories.

Behavioural mirror of statdepth/depth/calculations/_helper.py: `DepthDegeneracy` (:13-14) and the
checks `_handle_depth_errors` performs (:59-107) -- same order, same exception types, same messages,
so callers' try/except blocks and message matching keep working.  The checks are kept as a table of
(condition, exception, message) rows evaluated top to bottom.
"""
from typing import Callable

import numpy as np


class DepthDegeneracy(Exception):
    """Depth is not well defined for this input (degenerate simplices, empty sample blocks ...)."""


def _n_observations_too_few(data, J):
    # NB the univariate branch compares J with the number of ROWS (timepoints) of the single frame,
    # exactly as the reference does (_helper.py:83)
    if len(data) == 1:
        return J >= len(data[0])
    return J >= len(data)


def _validation_table(data, J, containment, relax, deep_check):
    """Rows are lazily evaluated: each condition is a thunk so later rows may assume earlier ones passed."""
    return (
        (lambda: not isinstance(data, list), ValueError, 'data must be passed as a list.'),
        (lambda: not isinstance(J, int), ValueError, 'J must be an integer.'),
        (lambda: not isinstance(containment, (str, Callable)), ValueError,
         'containment must be of type str or Callable.'),
        (lambda: not isinstance(deep_check, bool), ValueError, 'deep_check must be of type bool.'),
        (lambda: not isinstance(relax, bool), ValueError, 'relax must be of type bool'),
        (lambda: J < 2, ValueError, 'Parameter J must be greater than or equal to 2.'),
        (lambda: len(data) == 0, ValueError, 'No data passed.'),
        (lambda: _n_observations_too_few(data, J), ValueError,
         'Parameter J must be less than the number of observations.'),
        (lambda: len(data) > 1 and containment == 'r2', ValueError,
         'containment argument \'r2\' is invalid for multivariate data. '
         'Use one of [\'r2_enum\', \'simplex \'] or a passed containment method. '),
    )


def _handle_depth_errors(data, J, containment, relax, deep_check) -> None:
    for failed, exc, message in _validation_table(data, J, containment, relax, deep_check):
        if failed():
            raise exc(message)
    # Fewer than d + 2 functions: every simplex is degenerate (_helper.py:92-93).  The reference means to
    # raise DepthDegeneracy here (formatting its message fails with TypeError first); we raise what it means.
    if containment == 'simplex':
        dim = data[0].shape[1]
        if len(data) < dim + 2:
            raise DepthDegeneracy(f'Error: Need at least {dim + 2} functions to form non-degenerate '
                                  f'simplices in {dim} dimensional space. Only have {len(data)}.')
    if deep_check:
        _deep_check(data)


def _deep_check(data) -> None:
    """Numeric dtypes everywhere and one common index (_helper.py:95-107)."""
    reference_index = data[0].index
    for frame in data:
        inferred = frame.infer_objects()
        if any(not np.issubdtype(inferred[c].dtype, np.number) for c in inferred):
            raise ValueError('DataFrame must only contain numeric dtypes.')
    for frame in data:
        if not all(reference_index == frame.index):
            raise ValueError('DataFrames indices must be the same')
