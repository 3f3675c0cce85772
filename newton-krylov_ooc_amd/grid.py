"""Host-side grid set-up for the py_driver_2d path: axes and time-invariant fields.

These are the small, one-off arrays the device context is created from (they are
uploaded once and stay in HBM).  They follow the reference so that the generated
inputs are bit-for-bit the reference's:
  SpatialAxis metrics            nk_ooc/spatial_axis.py:14-45
  stretched edges                nk_ooc/spatial_axis.py:253-290
  stream function / velocities   nk_ooc/py_driver_2d/advection.py:23-49
  horizontal mixing coefficient  nk_ooc/py_driver_2d/horiz_mix.py:25-48
  boundary-layer depth profile   nk_ooc/py_driver_2d/vert_mix.py:89-101
"""

import numpy as np

YEAR = 365.0 * 86400.0


class SpatialAxis:
    """layer edges and the quantities derived from them"""

    def __init__(self, axisname, edges, units=None, defn_dict_values=None):
        self.axisname = axisname
        self.edges = np.asarray(edges, dtype=np.float64)
        self.units = "m" if units is None else units
        self.defn_dict_values = defn_dict_values
        self.mid = 0.5 * (self.edges[:-1] + self.edges[1:])
        self.delta = self.edges[1:] - self.edges[:-1]
        self.delta_r = 1.0 / self.delta
        self.delta_mid = self.mid[1:] - self.mid[:-1]
        self.delta_mid_r = 1.0 / self.delta_mid
        self.dump_names = {
            "bounds": f"{axisname}_bounds",
            "edges": f"{axisname}_edges",
            "delta": f"{axisname}_delta",
        }

    def __len__(self):
        return len(self.mid)

    def dump_dimensions(self):
        return {self.axisname: len(self), "nbnds": 2, self.dump_names["edges"]: len(self) + 1}

    def dump_vars_metadata(self):
        name = self.axisname
        return {
            name: {
                "dimensions": (name,),
                "attrs": {"long_name": f"{name} layer midpoints", "units": self.units,
                          "bounds": self.dump_names["bounds"]},
            },
            self.dump_names["bounds"]: {
                "dimensions": (name, "nbnds"),
                "attrs": {"long_name": f"{name} layer bounds"},
            },
            self.dump_names["edges"]: {
                "dimensions": (self.dump_names["edges"],),
                "attrs": {"long_name": f"{name} layer edges", "units": self.units},
            },
            self.dump_names["delta"]: {
                "dimensions": (name,),
                "attrs": {"long_name": f"{name} layer thickness", "units": self.units},
            },
        }

    def dump_vals_dict(self):
        return {
            self.axisname: self.mid,
            self.dump_names["bounds"]: np.stack((self.edges[:-1], self.edges[1:]), axis=1),
            self.dump_names["edges"]: self.edges,
            self.dump_names["delta"]: self.delta,
        }


def edges_from_defn(nlevs, edge_start, edge_end, delta_ratio_max):
    """polynomially stretched layer edges (first/last thickness ratio given)"""
    if delta_ratio_max <= 0.0:
        raise ValueError("delta_ratio_max must be > 0.0 to ensure delta > 0.0")
    coord = np.linspace(-1.0, 1.0, nlevs)
    stretch_fcn = 0.125 * coord * (15 + coord * coord * (3 * coord * coord - 10))
    delta_avg = (1.0 / nlevs) * (edge_end - edge_start)
    stretch_factor = delta_avg * (delta_ratio_max - 1) / (delta_ratio_max + 1)
    delta = delta_avg + stretch_factor * stretch_fcn
    edges = np.empty(1 + nlevs)
    edges[0] = edge_start
    edges[1:] = edge_start + delta.cumsum()
    return edges


def axis_from_modelinfo(axisname, modelinfo):
    """axis from the `<axisname>_*` keys of the [modelinfo] cfg section
    (py_driver_2d/setup_solver.py:185-198 with input/py_driver_2d/model_params.cfg)"""
    nlevs = int(modelinfo[f"{axisname}_nlevs"])
    e0 = float(modelinfo[f"{axisname}_edge_start"])
    e1 = float(modelinfo[f"{axisname}_edge_end"])
    ratio = float(modelinfo[f"{axisname}_delta_ratio_max"])
    units = modelinfo.get(f"{axisname}_units", "m")
    defn = {"axisname": axisname, "units": units, "nlevs": nlevs, "edge_start": e0,
            "edge_end": e1, "delta_ratio_max": ratio, "delta_start": None}
    defn_str = "\n".join(f"{key}={value}" for key, value in defn.items())
    return SpatialAxis(axisname, edges_from_defn(nlevs, e0, e1, ratio), units, defn_str)


def gen_vel_field(depth, ypos, max_abs_vvel):
    """stream function and the (vvel, wvel) derived from it"""
    depth_norm = (depth.edges - depth.edges.min()) / (depth.edges.max() - depth.edges.min())
    stretch = 2.0
    depth_norm = stretch * depth_norm / (1 + (stretch - 1) * depth_norm)
    depth_fcn = (27.0 / 4.0) * depth_norm * (1.0 - depth_norm) ** 2
    ypos_norm = (ypos.edges - ypos.edges.min()) / (ypos.edges.max() - ypos.edges.min())
    ypos_fcn = 4.0 * ypos_norm * (1.0 - ypos_norm)
    stream = np.outer(depth_fcn, ypos_fcn)
    vvel = (stream[1:, :] - stream[:-1, :]) * depth.delta_r[:, np.newaxis]
    stream = stream * max_abs_vvel / abs(vvel).max()
    vvel = (stream[1:, :] - stream[:-1, :]) * depth.delta_r[:, np.newaxis]
    wvel = (stream[:, 1:] - stream[:, :-1]) * ypos.delta_r
    return stream, vvel, wvel


def gen_hmix_coeff(depth, ypos, vvel, horiz_mix_coeff):
    """lateral mixing coefficient / dy at interior ypos faces, Peclet number <= 2"""
    if horiz_mix_coeff > 0.0:
        res = np.full((len(depth), len(ypos) - 1), horiz_mix_coeff)
        peclet_p5 = (0.5 / horiz_mix_coeff) * ypos.delta_mid[:] * abs(vvel[:, 1:-1])
        res *= np.where(peclet_p5 > 1.0, peclet_p5, 1.0)
        res *= ypos.delta_mid_r
    else:
        res = 0.5 * abs(vvel[:, 1:-1])
    return res


BLDEPTH_MIN = 35.0
BLDEPTH_YPOS = [0.4e6, 0.8e6, 1.0e6, 1.2e6, 1.4e6, 1.5e6]
BLDEPTH_VALS = [3000.0, 800.0, 415.0, 325.0, 280.0, BLDEPTH_MIN]


def gen_bldepth_max(ypos):
    return np.interp(ypos.mid, BLDEPTH_YPOS, BLDEPTH_VALS)


def bldepth_time_knots():
    return YEAR * np.array([0.25, 0.35, 0.65, 0.75]), np.array([0.0, 1.0, 1.0, 0.0])


class Grid2d:
    """everything time invariant the device context needs for one (depth, ypos) grid"""

    def __init__(self, depth, ypos, max_abs_vvel, horiz_mix_coeff):
        self.depth = depth
        self.ypos = ypos
        self.max_abs_vvel = float(max_abs_vvel)
        self.horiz_mix_coeff = float(horiz_mix_coeff)
        self.stream, self.vvel, self.wvel = gen_vel_field(depth, ypos, self.max_abs_vvel)
        self.hmix_coeff = gen_hmix_coeff(depth, ypos, self.vvel, self.horiz_mix_coeff)
        self.bldepth_max = gen_bldepth_max(ypos)

    @classmethod
    def from_modelinfo(cls, modelinfo):
        depth = axis_from_modelinfo(modelinfo.get("depth_axisname", "depth"), modelinfo)
        ypos = axis_from_modelinfo(modelinfo.get("ypos_axisname", "ypos"), modelinfo)
        return cls(depth, ypos, float(modelinfo["max_abs_vvel"]),
                   float(modelinfo["horiz_mix_coeff"]))

    @classmethod
    def default(cls, nz, ny, max_abs_vvel=0.1, horiz_mix_coeff=1000.0):
        """the grid of input/py_driver_2d/model_params.cfg at (nz, ny) levels"""
        depth = SpatialAxis("depth", edges_from_defn(nz, 0.0, 4000.0, 19.0))
        ypos = SpatialAxis("ypos", edges_from_defn(ny, 0.0, 50.0e5, 1.0))
        return cls(depth, ypos, max_abs_vvel, horiz_mix_coeff)
