"""Plotly views of depth results (reference: statdepth/depth/depth.py:91-175 curves, :193-335 point clouds).

Pure visualisation on the host, outside the hot path: same method names, arguments and conventions as the reference
(everything drawn in a light colour, the n deepest / most outlying items in red; `return_plot=True` hands the figure
back instead of showing it).  plotly is imported on first use.
"""
import pandas as pd

_PLAIN = dict(color='#6ea8ff', width=.5)
_MARKED = dict(color='Red', width=1)


def _go():
    try:
        import plotly.graph_objects as go
    except ImportError as e:                      # pragma: no cover - plotly is an optional dependency here
        raise ImportError('plotting needs plotly (pip install plotly)') from e
    return go


def _finish(fig, return_plot, showlegend=False):
    fig.update_layout(showlegend=showlegend)
    if return_plot:
        return fig
    fig.show()
    return None


def curves_figure(frame: pd.DataFrame, marked, title=None, xaxis_title=None, yaxis_title=None, return_plot=False,
                  showlegend=False):
    """Every column of `frame` as a thin line, the columns listed in `marked` on top in red (:91-139)."""
    go = _go()
    marked = list(marked)
    plain = [c for c in frame.columns if c not in set(marked)]
    traces = [go.Scatter(x=frame.index, y=frame[c], mode='lines', name=str(c), line=style)
              for group, style in ((plain, _PLAIN), (marked, _MARKED)) for c in group]
    layout = go.Layout(title=dict(text=title, y=0.9, x=0.5, xanchor='center', yanchor='top'),
                       xaxis=dict(title=xaxis_title), yaxis=dict(title=yaxis_title))
    return _finish(go.Figure(data=traces, layout=layout), return_plot, showlegend)


def _scatter(go, frame: pd.DataFrame, **marker_kw):
    """2-D or 3-D marker trace of a point cloud; other dimensions have no picture in the reference either (:215-216,307-308)."""
    cols = list(frame.columns)
    if len(cols) == 2:
        return go.Scatter(x=frame[cols[0]], y=frame[cols[1]], mode='markers', **marker_kw)
    if len(cols) == 3:
        return go.Scatter3d(x=frame[cols[0]], y=frame[cols[1]], z=frame[cols[2]], mode='markers', **marker_kw)
    if len(cols) < 2:
        raise ValueError(f'Error: Dimensionality of data must be >=2. Value found is {len(cols)}')
    raise NotImplementedError('point clouds of more than 3 dimensions have no plot (parallel axes are a stub upstream)')


def points_figure(frame: pd.DataFrame, marked_index, return_plot=False, title='', xaxis_title=None, yaxis_title=None):
    """All points in blue, the points listed in `marked_index` in red (:248-307)."""
    go = _go()
    traces = [_scatter(go, frame, marker_color='blue', name=''),
              _scatter(go, frame.loc[list(marked_index), :], marker_color='red', name='')]
    fig = go.Figure(data=traces, layout=go.Layout(title=title, xaxis_title=xaxis_title, yaxis_title=yaxis_title))
    return _finish(fig, return_plot)


def depth_coloured_figure(frame: pd.DataFrame, depths: pd.Series, invert_colors=False, marker=None, return_plot=False,
                          title='', xaxis_title=None, yaxis_title=None):
    """Points coloured by their depth (:193-243)."""
    go = _go()
    shade = 1 - depths if invert_colors else depths
    if marker is None:
        marker = dict(color=shade, colorscale='viridis', size=7)
    fig = go.Figure(data=[_scatter(go, frame, marker=marker)],
                    layout=go.Layout(title=title, xaxis_title=xaxis_title, yaxis_title=yaxis_title))
    return _finish(fig, return_plot)
