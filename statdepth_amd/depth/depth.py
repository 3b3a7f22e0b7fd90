"""Public factories and result objects (reference: statdepth/depth/depth.py).

`FunctionalDepth` (:362-402) and `PointcloudDepth` (:347-359) keep the reference's
signatures; `device=` and `algo=` are keyword-only additions.  Result classes keep
the reference's method names and conventions (:14-65,178-185,337-341), including the
plotly views (:91-175,193-335; `_plotting.py`, host-side visualisation outside the hot path).
"""
from typing import List

import pandas as pd

from abc import ABC, abstractmethod
from .calculations._functional import _functionaldepth, _samplefunctionaldepth
from .calculations._helper import DepthDegeneracy   # noqa: F401
from .calculations._pointcloud import _pointwisedepth, _samplepointwisedepth
from . import _plotting

__all__ = ['FunctionalDepth', 'PointcloudDepth']


class AbstractDepth(ABC):
    """What every depth result can do (reference: statdepth/depth/abstract.py:3-18)."""

    @abstractmethod
    def ordered(self, ascending=False): ...

    @abstractmethod
    def deepest(self, n=1): ...

    @abstractmethod
    def outlying(self, n=1): ...


class _FunctionalDepthSeries(AbstractDepth, pd.Series):
    """The depth values as a Series that also remembers the data they were computed from (:14-65).

    Ordering conventions of the reference are kept: `ordered()` caches the FIRST ordering it is asked for
    (:25-27), `deepest`/`outlying` read the head/tail of a descending ordering (:29-46), `quartile` ignores
    its argument and returns the lower half (:55-56).  Ties are ordered by `Series.sort_values`'s default
    sort, like the reference.
    """

    def __init__(self, df: pd.DataFrame, depths: pd.Series):
        super().__init__(data=depths)
        # plain attributes (bypassing pandas' attribute machinery); a reference to the frame, not a copy (:19)
        for name, value in (('_orig_data', df), ('_depths', depths), ('_ordered_depths', None)):
            object.__setattr__(self, name, value)

    def _ranking(self, ascending=False) -> pd.Series:
        cached = self._ordered_depths
        if cached is None:
            cached = self._depths.sort_values(ascending=ascending)
            object.__setattr__(self, '_ordered_depths', cached)
        return cached

    @staticmethod
    def _slice(ranked: pd.Series, sel) -> pd.Series:
        return pd.Series(index=list(ranked.index[sel]), data=list(ranked.values[sel]))

    def ordered(self, ascending=False) -> pd.Series:
        return self._ranking(ascending)

    def deepest(self, n=1) -> pd.Series:
        return self._slice(self._ranking(False), slice(0, n))

    def outlying(self, n=1) -> pd.Series:
        return self._slice(self._ranking(False), slice(-n, None))

    # aliases (:48-65)
    def sorted(self, ascending=False):
        return self.ordered(ascending=ascending)

    def median(self):
        return self.deepest(n=1)

    def quartile(self, ratio=0.5):
        return self._depths.sort_values().head(int(self._depths.shape[0] * 0.5))

    def get_depths(self):
        return self._depths

    def depths(self):
        return self._depths

    def get_data(self):
        return self._orig_data


class _FunctionalDepthUnivariate(_FunctionalDepthSeries):
    '''Univariate curves are the COLUMNS of the frame (:178-185).'''

    def drop_outlying_data(self, n=1) -> pd.DataFrame:
        return self._orig_data.drop(self.outlying(n=n).index, axis=1)

    def get_deepest_data(self, n=1) -> pd.DataFrame:
        return self._orig_data.loc[:, self.deepest(n=n).index]

    def get_outlying_data(self, n=1) -> pd.DataFrame:
        return self._orig_data.loc[:, self.outlying(n=n).index]

    def plot_deepest(self, n=1, title=None, xaxis_title=None, yaxis_title=None, return_plot=False, showlegend=False):
        '''All curves, the n deepest in red (:141-157).'''
        return _plotting.curves_figure(self._orig_data, self.deepest(n=n).index, title, xaxis_title, yaxis_title,
                                       return_plot, showlegend)

    def plot_outlying(self, n=1, title=None, xaxis_title=None, yaxis_title=None, return_plot=False, showlegend=False):
        '''All curves, the n most outlying in red (:159-175).'''
        return _plotting.curves_figure(self._orig_data, self.outlying(n=n).index, title, xaxis_title, yaxis_title,
                                       return_plot, showlegend)


class _PointwiseDepth(_FunctionalDepthSeries):
    '''Points are the ROWS of the frame (:337-341).'''

    def drop_outlying_data(self, n=1) -> pd.DataFrame:
        return self._orig_data.drop(self.outlying(n=n).index, axis=0)

    def get_deepest_data(self, n=1) -> pd.DataFrame:
        return self._orig_data.loc[self.deepest(n=n).index, :]

    def plot_depths(self, invert_colors=False, marker=None, return_plot=False, title='', xaxis_title=None,
                    yaxis_title=None):
        '''Points coloured by depth (:193-243).'''
        return _plotting.depth_coloured_figure(self._orig_data, self._depths, invert_colors, marker, return_plot, title,
                                               xaxis_title, yaxis_title)

    def plot_distribution(self, invert_colors=False, marker=None):
        '''Alias of plot_depths, shown at once (:343-344).'''
        return self.plot_depths(invert_colors=invert_colors, marker=marker)

    def plot_deepest(self, n=1, return_plot=False, title='', xaxis_title=None, yaxis_title=None):
        '''All points in blue, the n deepest in red (:309-321).'''
        return _plotting.points_figure(self._orig_data, self.deepest(n=n).index, return_plot, title, xaxis_title,
                                       yaxis_title)

    def plot_outlying(self, n=1, return_plot=False, title='', xaxis_title=None, yaxis_title=None):
        '''All points in blue, the n most outlying in red (:323-335).'''
        return _plotting.points_figure(self._orig_data, self.outlying(n=n).index, return_plot, title, xaxis_title,
                                       yaxis_title)


def PointcloudDepth(data: pd.DataFrame, to_compute: pd.Index = None, K=None, containment='simplex', quiet=True,
                    *, device=None) -> _PointwiseDepth:
    if K is not None:
        depth = _samplepointwisedepth(data=data, to_compute=to_compute, K=K, containment=containment,
                                      device=device)
    else:
        depth = _pointwisedepth(data=data, to_compute=to_compute, containment=containment, device=device)
    return _PointwiseDepth(df=data, depths=depth)


def FunctionalDepth(data: List[pd.DataFrame], to_compute=None, K=None, J=2, containment='r2', relax=False,
                    deep_check=False, quiet=True, *, device=None, algo='auto'):
    if K is not None:
        depth = _samplefunctionaldepth(data=data, to_compute=to_compute, K=K, J=J, containment=containment,
                                       relax=relax, deep_check=deep_check, quiet=quiet, device=device, algo=algo)
    else:
        depth = _functionaldepth(data=data, to_compute=to_compute, J=J, containment=containment, relax=relax,
                                 deep_check=deep_check, quiet=quiet, device=device, algo=algo)
    if len(data) == 1:                                   # univariate by assumption (:399-400)
        return _FunctionalDepthUnivariate(df=data[0], depths=depth)
    return _FunctionalDepthSeries(df=data[0], depths=depth)   # multivariate (:401-402)
