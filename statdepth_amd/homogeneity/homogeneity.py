"""Homogeneity coefficients between two samples -- the heaviest CALLER of the depth hot path.

Mirrors statdepth/homogeneity/homogeneity.py: `FunctionalHomogeneity` (:9-35),
`PointcloudHomogeneity` (:37-60), `_functionalhomogeneity` (:65-153), `_pointcloudhomogeneity` (:155-200),
`P1_homogeneity` / `P2_homogeneity` (:214-306).  Same values, same conventions; unlike the reference
the caller's frames are not mutated.

Where the reference issues |G| separate `FunctionalDepth(F + [g], to_compute=[g])` calls (P3, :125-128)
-- each one enumerating C(n,2) pairs in Python -- this module makes ONE device launch for all of G
(`sd_mbd_external_counts`: every curve of G against the curves of F).  With K (block sampling) or a custom
containment the reference's call-by-call structure is kept, because those paths consume the global RNG /
user code per call.
"""
from typing import List

import numpy as np
import pandas as pd
from scipy.special import binom

from .. import engine
from ..depth.depth import FunctionalDepth, PointcloudDepth


class FunctionalHomogeneity:
    def __init__(self, F, G, method='p1', K=None, J=2, containment='r2', relax=False, deep_check=False, quiet=False):
        self._orig_F = F
        self._orig_G = G
        self._hom = _functionalhomogeneity(F=F, G=G, K=K, J=J, containment=containment, method=method, relax=relax,
                                           deep_check=deep_check, quiet=quiet)

    def homogeneity(self):
        return self._hom

    def __str__(self):
        return str(self.homogeneity())

    def __repr__(self):
        return str(self.homogeneity())


class PointcloudHomogeneity:
    def __init__(self, F, G, method='p1', K=None, J=None, containment='simplex', relax=False, deep_check=False):
        self._orig_F = F
        self._orig_G = G
        self._F_depths, self._G_depths, self._hom = _pointcloudhomogeneity(F=F, G=G, K=K, containment=containment,
                                                                           method=method)

    def F_depths(self):
        return self._F_depths

    def G_depths(self):
        return self._G_depths

    def homogeneity(self):
        return self._hom

    def __str__(self):
        return str(self.homogeneity())


def _handle_errors(F, G, method='p1'):
    if len(F) != len(G):                                   # (:203-204)
        raise ValueError('F and G must have data of the same length')
    if len(F) == 1:
        if F[0].shape[0] != G[0].shape[0]:                 # (:206-208)
            raise ValueError('Univariate data must have same number of time indices to check containment.')


def _depths_of_external(F: pd.DataFrame, Gcols: pd.DataFrame, J: int, relax: bool) -> np.ndarray:
    """Depth of every column g of `Gcols` inside F u {g}, one launch (built-in 'r2', relax=True).

    The reference builds F with g appended (n_F + 1 columns) and divides by binom(n_F + 1, j)
    (_functional.py:229,253); bands come from the n_F curves of F (:235).
    """
    Fx = F.to_numpy(dtype=np.float64)
    Gx = Gcols.to_numpy(dtype=np.float64)
    T, nF = Fx.shape
    counts = engine.mbd_external_counts(Fx, Gx, J=J).astype(np.float64) / T
    depth = np.zeros(Gx.shape[1])
    for j in range(2, J + 1):
        depth += counts[:, j - 2] / binom(nF + 1, j)
    return depth


def _functionalhomogeneity(F: List[pd.DataFrame], G: List[pd.DataFrame], K=None, J=2, containment='r2', method='p1',
                           relax=False, deep_check=False, quiet=False):
    _handle_errors(F, G, method)
    kw = dict(K=K, J=J, containment=containment, relax=relax, deep_check=deep_check, quiet=quiet)
    G_depths = FunctionalDepth(data=G, **kw)               # (:79-87)

    if len(F) == 1:                                        # univariate (:90-138)
        Fd, Gd = F[0], G[0]
        if 'g_deepest' in Fd.columns:
            Fd = Fd.drop('g_deepest', axis=1)
        G_deepest = G_depths.get_deepest_data(n=1)          # (:95)
        batched = (K is None and containment == 'r2' and relax)
        if batched:
            G_deep_in_F = pd.Series(index=['g_deepest'], data=_depths_of_external(Fd, G_deepest, J, relax))
        else:
            Fg = Fd.copy()
            Fg.loc[:, 'g_deepest'] = G_deepest.iloc[:, 0].to_numpy()     # (:101)
            G_deep_in_F = FunctionalDepth([Fg], to_compute=['g_deepest'], **kw)
        if method == 'p1':
            return G_deep_in_F
        elif method == 'p2':
            F_depths = FunctionalDepth([Fd], **kw)
            return np.abs(G_deep_in_F - F_depths.median().iloc[0])       # (:120)
        elif method == 'p3':
            if batched:
                t = _depths_of_external(Fd, Gd, J, relax)
            else:
                t = []
                for col in Gd.columns:                      # (:125-128)
                    Fg = Fd.copy()
                    Fg.loc[:, col] = Gd.loc[:, col].to_numpy()
                    t.append(FunctionalDepth([Fg], to_compute=[col], K=K, J=J, containment=containment, relax=relax,
                                             deep_check=deep_check).loc[col])
            depths_G_in_F = pd.Series(index=list(Gd.columns), data=t).sort_values(ascending=False)
            return depths_G_in_F.iloc[0] / G_depths.median().iloc[0]      # (:133)
        elif method == 'p4':
            raise NotImplementedError()
        else:
            raise ValueError(f'{method} is not a valid depth method for the given data. '
                             f'Use one of [\'p1\', \'p2\', \'p3\', \'p4\']')
    else:                                                   # multivariate (:139-153)
        G_deepest = G[G_depths.index[0]]                    # the reference takes the FIRST index, not the deepest (:140)
        Fp = list(F) + [G_deepest]
        G_deep_in_F = FunctionalDepth(Fp, to_compute=[len(Fp) - 1], K=K, J=J, containment=containment, relax=relax,
                                      deep_check=deep_check).ordered().iloc[0]
        if method == 'p1':
            return G_deep_in_F / G_depths.median().iloc[0]
        elif method == 'p2':
            F_depths = FunctionalDepth(F, None, K, J, containment, relax, deep_check)
            return 1 - np.abs(G_deep_in_F - F_depths.median().iloc[0])
        elif method == 'p3':
            return None                                      # `pass` in the reference (:150-151)
        else:
            raise ValueError(f'{method} is not a valid depth method for the given data. '
                             f'Use one of [\'p1\', \'p2\', \'p3\', \'p4\']')


def _pointcloudhomogeneity(F: pd.DataFrame, G: pd.DataFrame, K=None, containment='simplex', method='p1'):
    _handle_errors(F, G, method)
    G_depths = PointcloudDepth(data=G, K=K, containment=containment)
    F_depths = PointcloudDepth(data=F, K=K, containment=containment)
    hom = 0
    G_deepest = G_depths.get_deepest_data(n=1).copy()
    G_deepest.index = ['g_deepest']
    # the reference uses DataFrame.append (:173), gone in pandas >= 2; pd.concat is its definition
    Fg = pd.concat([F, G_deepest])
    G_deep_in_F = PointcloudDepth(Fg, to_compute=['g_deepest'], K=K, containment=containment).ordered().loc['g_deepest']
    if method == 'p1':
        hom = G_deep_in_F / F_depths.median().iloc[0]
    elif method == 'p2':
        hom = 1 - np.abs(G_deep_in_F - F_depths.median().iloc[0])
    elif method == 'p3':
        t = []
        for point in G.index:                               # (:183-186)
            Fp = F.copy()
            Fp.loc[point, :] = G.loc[point, :]
            t.append(PointcloudDepth(Fp, to_compute=[point], K=K, containment=containment).loc[point])
        depths_G_in_F = pd.Series(index=list(G.index), data=t).sort_values(ascending=False)
        hom = depths_G_in_F.iloc[0] / G_depths.median().iloc[0]
    elif method == 'p4':
        t1 = np.abs(_pointcloudhomogeneity(F, G, K, containment, 'p3')[2] - _pointcloudhomogeneity(F, F, K, containment, 'p1')[2])
        t2 = np.abs(_pointcloudhomogeneity(F, G, K, containment, 'p3')[2] - _pointcloudhomogeneity(G, G, K, containment, 'p1')[2])
        hom = t1 * t2
    else:
        raise ValueError(f'{method} is not a valid depth method for the given data. '
                         f'Use one of [\'p1\', \'p2\', \'p3\', \'p4\']')
    return F_depths, G_depths, hom


def P1_homogeneity(F: pd.DataFrame, G: pd.DataFrame, K=None, J=2, containment='r2', relax=False, quiet=False) -> float:
    '''P1 coefficient (:214-260): depth of G's deepest curve inside F.'''
    G_depth = FunctionalDepth(data=[G], K=K, J=J, containment=containment, relax=relax, quiet=quiet)
    G_deepest = G_depth.get_deepest_data()
    Fg = F.copy()
    Fg.loc[:, 'G_deepest'] = G_deepest.iloc[:, 0].to_numpy()
    G_deep_in_F = FunctionalDepth([Fg], to_compute=['G_deepest'], K=K, J=J, containment=containment, relax=relax,
                                  quiet=quiet)
    return G_deep_in_F.iloc[0]


def P2_homogeneity(F: pd.DataFrame, G: pd.DataFrame, K=None, J=2, containment='r2', relax=False, quiet=False) -> float:
    '''P2 coefficient (:262-306): |P1(F,G) - depth of F's own deepest curve|.'''
    P1_F_G = P1_homogeneity(F=F, G=G, K=K, J=J, containment=containment, relax=relax, quiet=quiet)
    P1_F_F = FunctionalDepth(data=[F], K=K, J=J, containment=containment, relax=relax, quiet=quiet).deepest().iloc[0]
    return np.abs(P1_F_G - P1_F_F)
