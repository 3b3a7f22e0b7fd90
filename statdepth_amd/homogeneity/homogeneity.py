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
    """Depth of every column g of `Gcols` inside F u {g}, one launch (built-in 'r2'; relax=True, or the
    reference's default relax=False with J = 2).

    The reference builds F with g appended (n_F + 1 columns) and divides by binom(n_F + 1, j)
    (_functional.py:229,253); bands come from the n_F curves of F (:235).
    """
    Fx = F.to_numpy(dtype=np.float64)
    Gx = Gcols.to_numpy(dtype=np.float64)
    T, nF = Fx.shape
    if relax:
        counts = engine.mbd_external_counts(Fx, Gx, J=J).astype(np.float64) / T
    else:
        counts = engine.bd_strict_external_counts(Fx, Gx).astype(np.float64)[:, None]      # `c // T`: every timepoint
    depth = np.zeros(Gx.shape[1])
    for j in range(2, J + 1):
        depth += counts[:, j - 2] / binom(nF + 1, j)
    return depth


def _fresh_label(frame_labels, stem):
    """A label no column / row of the frame carries: the temporary name g travels under inside F."""
    taken = set(frame_labels)
    label, k = stem, 0
    while label in taken:
        k += 1
        label = f'{stem}_{k}'
    return label


def _depth_of_one_external(Fd: pd.DataFrame, g: np.ndarray, kw) -> float:
    """FunctionalDepth(F u {g}, to_compute=[g]) with g under a label of its own (never one of F's: the reference's
    `F.loc[:, col] = G.loc[:, col]` (:126) overwrites F's curve when the samples share labels)."""
    lab = _fresh_label(Fd.columns, 'g_external')
    Fg = Fd.copy()
    Fg.loc[:, lab] = g
    return FunctionalDepth([Fg], to_compute=[lab], **kw).loc[lab]


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
        # every g is evaluated inside the intact F u {g} (n_F + 1 curves), batched or not: one semantics for both
        batched = K is None and containment == 'r2' and (relax or J == 2)
        if batched:
            deep = _depths_of_external(Fd, G_deepest, J, relax)
        else:
            deep = [_depth_of_one_external(Fd, G_deepest.iloc[:, 0].to_numpy(), kw)]
        G_deep_in_F = pd.Series(index=['g_deepest'], data=deep)
        if method == 'p1':
            return G_deep_in_F
        elif method == 'p2':
            F_depths = FunctionalDepth([Fd], **kw)
            return np.abs(G_deep_in_F - F_depths.median().iloc[0])       # (:120)
        elif method == 'p3':
            if batched:
                t = _depths_of_external(Fd, Gd, J, relax)    # (:125-128) for all of G in one launch
            else:
                t = [_depth_of_one_external(Fd, Gd.iloc[:, c].to_numpy(), kw) for c in range(Gd.shape[1])]
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


# ---- point clouds (:155-200) -------------------------------------------------------------------------------------
# The reference appends one point of G to F, asks for that point's depth, and drops it again -- once for G's deepest
# point and, for P3, once per point of G.  Here the points of G are EXTERNAL targets of one launch against F
# (sd_pointcloud_simplex_external_counts / sd_l1_external_depth); the sample the depth refers to is F u {g}
# (n_F + 1 points), as in the reference's temporary frame.  With K (block sampling) the estimator draws from the
# global RNG per call, so the calls stay separate -- each is itself one launch (_samplepointwisedepth).
def _external_point_depths(F: pd.DataFrame, pts: pd.DataFrame, K, containment) -> np.ndarray:
    if K is not None:
        out = []
        for r in range(pts.shape[0]):
            lab = _fresh_label(F.index, 'g_external')
            Fg = pd.concat([F, pts.iloc[[r], :].set_axis([lab], axis=0)])
            out.append(PointcloudDepth(Fg, to_compute=[lab], K=K, containment=containment).loc[lab])
        return np.asarray(out, dtype=np.float64)
    Fx, Qx = F.to_numpy(dtype=np.float64), pts.to_numpy(dtype=np.float64)
    n, d = Fx.shape
    if containment == 'simplex':
        return engine.pointcloud_simplex_external_counts(Fx, Qx).astype(np.float64) / binom(n + 1, d + 1)
    if containment == 'l1':
        return engine.l1_external_depth(Fx, Qx)
    if containment in ('mahalanobis', 'oja'):
        raise NotImplementedError(f'{containment} depth is outside the band-depth hot path this engine covers')
    raise ValueError(f'{containment} is not a valid containment measure. ')


def _pointcloudhomogeneity(F: pd.DataFrame, G: pd.DataFrame, K=None, containment='simplex', method='p1'):
    if method not in ('p1', 'p2', 'p3', 'p4'):
        raise ValueError(f'{method} is not a valid depth method for the given data. '
                         f'Use one of [\'p1\', \'p2\', \'p3\', \'p4\']')
    _handle_errors(F, G, method)
    depth_of = lambda sample: PointcloudDepth(data=sample, K=K, containment=containment)      # noqa: E731
    F_depths, G_depths = depth_of(F), depth_of(G)

    def deepest_in(sample_depths, host):                   # depth of a sample's deepest point inside host u {point}
        return _external_point_depths(host, sample_depths.get_deepest_data(n=1), K, containment)[0]

    def p3():                                              # best depth any point of G reaches inside F, over G's median
        return _external_point_depths(F, G, K, containment).max() / G_depths.median().iloc[0]

    if method == 'p1':
        hom = deepest_in(G_depths, F) / F_depths.median().iloc[0]
    elif method == 'p2':
        hom = 1 - np.abs(deepest_in(G_depths, F) - F_depths.median().iloc[0])
    elif method == 'p3':
        hom = p3()
    else:                                                  # p4: |p3 - p1(F,F)| * |p3 - p1(G,G)| (:189-192)
        v = p3()
        p1_FF = deepest_in(F_depths, F) / F_depths.median().iloc[0]
        p1_GG = deepest_in(G_depths, G) / G_depths.median().iloc[0]
        hom = np.abs(v - p1_FF) * np.abs(v - p1_GG)
    return F_depths, G_depths, hom


def _with_g_deepest(F: pd.DataFrame, G: pd.DataFrame, kw) -> pd.DataFrame:
    G_deepest = FunctionalDepth(data=[G], **kw).get_deepest_data()
    Fg = F.copy()
    Fg.loc[:, 'G_deepest'] = G_deepest.iloc[:, 0].to_numpy()          # (:244)
    return Fg


def P1_homogeneity(F: pd.DataFrame, G: pd.DataFrame, K=None, J=2, containment='r2', relax=False, quiet=False) -> float:
    '''P1 coefficient (:214-260): depth of G's deepest curve inside F u {that curve}.'''
    kw = dict(K=K, J=J, containment=containment, relax=relax, quiet=quiet)
    return FunctionalDepth([_with_g_deepest(F, G, kw)], to_compute=['G_deepest'], **kw).iloc[0]


def P2_homogeneity(F: pd.DataFrame, G: pd.DataFrame, K=None, J=2, containment='r2', relax=False, quiet=False) -> float:
    '''P2 coefficient (:262-306): |P1(F,G) - deepest depth of F|.

    The reference's P1 leaves its 'G_deepest' column in the caller's F (:244), so the second term is the deepest
    depth of F WITH that column (:297-304).  The value is reproduced; the caller's frame is left alone.'''
    kw = dict(K=K, J=J, containment=containment, relax=relax, quiet=quiet)
    Fg = _with_g_deepest(F, G, kw)
    P1_F_G = FunctionalDepth([Fg], to_compute=['G_deepest'], **kw).iloc[0]
    P1_F_F = FunctionalDepth(data=[Fg], **kw).deepest().iloc[0]
    return np.abs(P1_F_G - P1_F_F)
