"""Oracle: spatial axes of the py_driver_2d model (test infrastructure only).

Follows reference `nk_ooc/spatial_axis.py:14-45` (derived axis metrics) and
`nk_ooc/spatial_axis.py:253-290` (`_edges_from_defn_dict`, the polynomially
stretched layer thicknesses).  Operation order is kept identical so that the
generated edges are bit-for-bit those of the reference.
"""

import numpy as np


def stretched_edges(nlevs, edge_start, edge_end, delta_ratio_max):
    """layer edges with thickness ratio `delta_ratio_max` between last and first
    layer (reference `spatial_axis.py:253-290`)"""
    if delta_ratio_max <= 0.0:
        raise ValueError("delta_ratio_max must be > 0.0 to ensure delta > 0.0")
    coord = np.linspace(-1.0, 1.0, nlevs)
    shape_fcn = 0.125 * coord * (15 + coord * coord * (3 * coord * coord - 10))
    delta_avg = (1.0 / nlevs) * (edge_end - edge_start)
    amp = delta_avg * (delta_ratio_max - 1) / (delta_ratio_max + 1)
    delta = delta_avg + amp * shape_fcn
    edges = np.empty(1 + nlevs)
    edges[0] = edge_start
    edges[1:] = edge_start + delta.cumsum()
    return edges


class Axis:
    """axis metrics derived from edges (reference `spatial_axis.py:35-39`)"""

    def __init__(self, name, edges):
        self.name = name
        self.edges = np.asarray(edges, dtype=np.float64)
        self.mid = 0.5 * (self.edges[:-1] + self.edges[1:])
        self.delta = self.edges[1:] - self.edges[:-1]
        self.delta_r = 1.0 / self.delta
        self.delta_mid = self.mid[1:] - self.mid[:-1]
        self.delta_mid_r = 1.0 / self.delta_mid

    def __len__(self):
        return len(self.mid)


def default_axes(nz, ny):
    """axes of `input/py_driver_2d/model_params.cfg:6-23` at (nz, ny) levels:
    depth 0..4000 m with delta_ratio_max 19, ypos 0..5e6 m uniform"""
    depth = Axis("depth", stretched_edges(nz, 0.0, 4000.0, 19.0))
    ypos = Axis("ypos", stretched_edges(ny, 0.0, 50.0e5, 1.0))
    return depth, ypos
