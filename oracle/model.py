"""Oracle: py_driver_2d processes, tendencies, Jacobian, preconditioner.

TEST INFRASTRUCTURE ONLY (see `oracle/__init__.py`).

State layout everywhere: C-order `(tracer, depth, ypos)` flattened, i.e. flat
index `tr*nz*ny + k*ny + j` (reference `py_driver_2d/model_state.py:97`,
`advection.py:129`).

Reference lines restated here
  velocity field           nk_ooc/py_driver_2d/advection.py:23-49
  advection tendency       nk_ooc/py_driver_2d/advection.py:51-76
  advection Jacobian       nk_ooc/py_driver_2d/advection.py:111-179
  horiz mixing coeff       nk_ooc/py_driver_2d/horiz_mix.py:25-48
  horiz mixing tendency    nk_ooc/py_driver_2d/horiz_mix.py:50-71
  horiz mixing Jacobian    nk_ooc/py_driver_2d/horiz_mix.py:100-149
  vert mixing coeff        nk_ooc/py_driver_2d/vert_mix.py:44-101
  conservative remap       nk_ooc/spatial_axis.py:136-187
  vert mixing tendency     nk_ooc/py_driver_2d/vert_mix.py:24-42
  vert mixing Jacobian     nk_ooc/py_driver_2d/vert_mix.py:140-188
  sum over processes       nk_ooc/py_driver_2d/tracer_module_state.py:98-108,262-270
  iage sources/Jacobian    nk_ooc/py_driver_2d/iage.py:17-64
  iage preconditioner      nk_ooc/py_driver_2d/iage.py:66-93
"""

import numpy as np
from scipy import interpolate, sparse
from scipy.sparse import linalg as sp_linalg

YEAR = 365.0 * 86400.0


def _interp2(x, x0, x1, y0, y1):
    """np.interp(x, [x0, x1], [y0, y1]) spelled out (two-knot clamp-linear ramp);
    x, x0, x1 broadcast.  Same arithmetic as numpy's compiled interp:
    slope*(x - x0) + y0 inside, end values outside / at the knots."""
    slope = (y1 - y0) / (x1 - x0)
    inner = slope * (x - x0) + y0
    res = np.where(x > x1, y1, np.where(x < x0, y0, inner))
    res = np.where(x == x1, y1, res)
    res = np.where(x == x0, y0, res)
    return res


def remap_two_knots(edges, delta_r, xv, yv):
    """literal restatement of `SpatialAxis.remap_linear_interpolant` (spatial_axis.py:136-187) for an
    interpolant with two knots: layer averages over [edges[k], edges[k+1]] of the piecewise linear
    function through (xv, yv), constant outside the knots.  Pinned by the cases of the reference's
    tests/test_spatial_axis.py:156-185 (tests/test_oracle_static.py)."""
    edges = np.asarray(edges, dtype=np.float64)
    ye = np.interp(edges, xv, yv)
    res = 0.5 * (ye[:-1] + ye[1:])
    lay = 0
    iv = 0
    while iv < 2:
        if xv[iv] < edges[0]:
            iv += 1
            continue
        if xv[iv] >= edges[-1]:
            break
        while xv[iv] >= edges[lay + 1]:
            lay += 1
        acc = (xv[iv] - edges[lay]) * (0.5 * (ye[lay] + yv[iv]))
        while iv < 2 and xv[iv] < edges[lay + 1]:
            if iv + 1 < 2 and xv[iv + 1] < edges[lay + 1]:
                acc += (xv[iv + 1] - xv[iv]) * (0.5 * (yv[iv] + yv[iv + 1]))
            else:
                acc += (edges[lay + 1] - xv[iv]) * (0.5 * (yv[iv] + ye[lay + 1]))
            iv += 1
        res[lay] = acc * delta_r[lay]
    return res


class Py2dModel:
    """time-invariant fields of the py_driver_2d processes on a (depth, ypos) grid"""

    def __init__(self, depth, ypos, max_abs_vvel=0.1, horiz_mix_coeff=1000.0):
        self.depth = depth
        self.ypos = ypos
        self.nz = len(depth)
        self.ny = len(ypos)
        self.max_abs_vvel = float(max_abs_vvel)
        self.horiz_mix_coeff = float(horiz_mix_coeff)
        self._gen_vel_field()
        self._gen_hmix_coeff()
        self._vmix_cache_t = None
        self._vmix_cache = None
        self._jac_static = None

    # ---- advection.py:23-49 -------------------------------------------------
    def _gen_vel_field(self):
        ze = self.depth.edges
        ye = self.ypos.edges
        dn = (ze - ze.min()) / (ze.max() - ze.min())
        stretch = 2.0
        dn = stretch * dn / (1 + (stretch - 1) * dn)
        fd = (27.0 / 4.0) * dn * (1.0 - dn) ** 2
        yn = (ye - ye.min()) / (ye.max() - ye.min())
        fy = 4.0 * yn * (1.0 - yn)
        stream = np.outer(fd, fy)
        vvel = (stream[1:, :] - stream[:-1, :]) * self.depth.delta_r[:, np.newaxis]
        stream = stream * self.max_abs_vvel / abs(vvel).max()
        self.stream = stream
        self.vvel = (stream[1:, :] - stream[:-1, :]) * self.depth.delta_r[:, np.newaxis]
        self.wvel = (stream[:, 1:] - stream[:, :-1]) * self.ypos.delta_r

    # ---- horiz_mix.py:25-48 -------------------------------------------------
    def _gen_hmix_coeff(self):
        kcoef = self.horiz_mix_coeff
        if kcoef > 0.0:
            res = np.full((self.nz, self.ny - 1), kcoef)
            peclet_p5 = (0.5 / kcoef) * self.ypos.delta_mid[:] * abs(self.vvel[:, 1:-1])
            res *= np.where(peclet_p5 > 1.0, peclet_p5, 1.0)
            res *= self.ypos.delta_mid_r
        else:
            res = 0.5 * abs(self.vvel[:, 1:-1])
        self.hmix_coeff = res

    # ---- vert_mix.py:89-101 -------------------------------------------------
    def bldepth(self, time):
        bld_min = 35.0
        bld_max = np.interp(
            self.ypos.mid,
            [0.4e6, 0.8e6, 1.0e6, 1.2e6, 1.4e6, 1.5e6],
            [3000.0, 800.0, 415.0, 325.0, 280.0, bld_min],
        )
        tvals = YEAR * np.array([0.25, 0.35, 0.65, 0.75])
        frac = np.interp(time, tvals, [0.0, 1.0, 1.0, 0.0])
        return bld_min + (bld_max - bld_min) * frac

    # ---- spatial_axis.py:136-187 specialised to two knots ---------------------
    def _remap_ramp(self, bld):
        """layer averages over [z_mid[k], z_mid[k+1]] of the ramp
        ln(10) @ bld-20 -> ln(5e-4) @ bld+20, for all columns at once.
        Elementwise the floating-point operations are those of the reference loop
        (`remap_ramp_loop` below is the literal loop form used to check this)."""
        y0 = np.log(1.0e1)
        y1 = np.log(5.0e-4)
        e = self.depth.mid[:, np.newaxis]  # edges of the remap target axis
        e0, e1 = e[:-1], e[1:]
        dr = self.depth.delta_mid_r[:, np.newaxis]
        x0 = (bld - 20.0)[np.newaxis, :]
        x1 = (bld + 20.0)[np.newaxis, :]
        ye = _interp2(e, x0, x1, y0, y1)
        ye0, ye1 = ye[:-1], ye[1:]
        res = 0.5 * (ye0 + ye1)
        in0 = (e0 <= x0) & (x0 < e1)
        in1 = (e0 <= x1) & (x1 < e1)
        # x0 in layer (x1 possibly too)
        s0 = (x0 - e0) * (0.5 * (ye0 + y0))
        s_both = s0 + (x1 - x0) * (0.5 * (y0 + y1))
        s_both = s_both + (e1 - x1) * (0.5 * (y1 + ye1))
        s_only0 = s0 + (e1 - x0) * (0.5 * (y0 + ye1))
        # only x1 in layer
        s1 = (x1 - e0) * (0.5 * (ye0 + y1))
        s1 = s1 + (e1 - x1) * (0.5 * (y1 + ye1))
        res = np.where(in0 & in1, s_both * dr, res)
        res = np.where(in0 & ~in1, s_only0 * dr, res)
        res = np.where(~in0 & in1, s1 * dr, res)
        return res

    def remap_ramp_loop(self, bld_val):
        """the reference's remap loop for ONE column of the vertical mixing ramp
        (spatial_axis.py:136-187), for checking `_remap_ramp` on small cases"""
        return remap_two_knots(self.depth.mid, self.depth.delta_mid_r, [bld_val - 20.0, bld_val + 20.0],
                               [np.log(1.0e1), np.log(5.0e-4)])

    # ---- vert_mix.py:44-87 --------------------------------------------------
    def vmix_coeff(self, time):
        """vertical mixing coeff / distance between layer mids, (nz-1, ny), m/s"""
        if self._vmix_cache_t is not None and time == self._vmix_cache_t:
            return self._vmix_cache
        kv = np.exp(self._remap_ramp(self.bldepth(time)))
        peclet_p5 = (
            0.5 * self.depth.delta_mid[:, np.newaxis] * abs(self.wvel[1:-1, :]) / kv
        )
        kv = kv * np.where(peclet_p5 > 1.0, peclet_p5, 1.0)
        kv = kv * self.depth.delta_mid_r[:, np.newaxis]
        self._vmix_cache_t = time
        self._vmix_cache = kv
        return kv

    # ---- tendencies ---------------------------------------------------------
    def tend_processes(self, time, c):
        """adv + hmix + vmix tendency for c of shape (tc, nz, ny)
        (tracer_module_state.py:98-108 with advection.py:51-76, horiz_mix.py:50-71,
        vert_mix.py:24-42); summed in the reference's process order"""
        nz, ny = self.nz, self.ny
        dyr = self.ypos.delta_r
        dzr = self.depth.delta_r[:, np.newaxis]
        kv = self.vmix_coeff(time)
        out = np.zeros_like(c)
        fy = np.zeros((nz, ny + 1))
        fz = np.zeros((nz + 1, ny))
        for tr in range(c.shape[0]):
            ct = c[tr]
            # advection
            fy[:, 1:-1] = 0.5 * (ct[:, 1:] + ct[:, :-1])
            fy *= self.vvel
            adv = dyr * (fy[:, :-1] - fy[:, 1:])
            fz[1:-1, :] = 0.5 * (ct[1:, :] + ct[:-1, :])
            fz *= self.wvel
            adv += dzr * (fz[1:, :] - fz[:-1, :])
            out[tr] += adv
            # horizontal mixing
            fy[:, 1:-1] = self.hmix_coeff * (ct[:, 1:] - ct[:, :-1])
            fy[:, 0] = 0.0
            fy[:, -1] = 0.0
            out[tr] += dyr * (fy[:, 1:] - fy[:, :-1])
            # vertical mixing
            fz[1:-1, :] = kv * (ct[1:, :] - ct[:-1, :])
            fz[0, :] = 0.0
            fz[-1, :] = 0.0
            out[tr] += dzr * (fz[1:, :] - fz[:-1, :])
        return out

    # ---- Jacobian as five diagonals per tracer ---------------------------------
    def _static_jac_parts(self):
        """advection + horizontal mixing pieces (time invariant)"""
        if self._jac_static is not None:
            return self._jac_static
        nz, ny = self.nz, self.ny
        dzr = self.depth.delta_r[:, np.newaxis]
        dyr = self.ypos.delta_r[np.newaxis, :]
        z = np.zeros((nz, ny))
        a_up, a_dn, a_s, a_n = z.copy(), z.copy(), z.copy(), z.copy()
        a_up[1:, :] = (-0.5 * self.wvel[1:-1, :]) * dzr[1:, :]
        a_s[:, 1:] = (0.5 * self.vvel[:, 1:-1]) * dyr[:, 1:]
        a_n[:, :-1] = (-0.5 * self.vvel[:, 1:-1]) * dyr[:, :-1]
        a_dn[:-1, :] = (0.5 * self.wvel[1:-1, :]) * dzr[:-1, :]
        # tmp_sum accumulates in the order up, south, north, down
        a_c = ((a_up + a_s) + a_n) + a_dn
        h_s, h_n = z.copy(), z.copy()
        h_s[:, 1:] = self.hmix_coeff * dyr[:, 1:]
        h_n[:, :-1] = self.hmix_coeff * dyr[:, :-1]
        h_c = -(h_s + h_n)
        self._jac_static = dict(a_up=a_up, a_dn=a_dn, a_s=a_s, a_n=a_n, a_c=a_c,
                                h_s=h_s, h_n=h_n, h_c=h_c)
        return self._jac_static

    def jac_diags(self, time):
        """(up, south, center, north, down) each (nz, ny): d tend[k,j] / d c[k-1,j],
        c[k,j-1], c[k,j], c[k,j+1], c[k+1,j]; entries for missing neighbours are 0.
        Process sums are taken in the reference order adv + hmix + vmix."""
        st = self._static_jac_parts()
        nz, ny = self.nz, self.ny
        dzr = self.depth.delta_r[:, np.newaxis]
        kv = self.vmix_coeff(time)
        v_up = np.zeros((nz, ny))
        v_dn = np.zeros((nz, ny))
        v_up[1:, :] = kv * dzr[1:, :]
        v_dn[:-1, :] = kv * dzr[:-1, :]
        v_c = -(v_up + v_dn)
        up = st["a_up"] + v_up
        dn = st["a_dn"] + v_dn
        south = st["a_s"] + st["h_s"]
        north = st["a_n"] + st["h_n"]
        center = (st["a_c"] + st["h_c"]) + v_c
        return up, south, center, north, dn

    def diags_to_csr(self, up, south, center, north, dn):
        """single-tracer CSR matrix from the five diagonals"""
        nz, ny = self.nz, self.ny
        P = nz * ny
        mat = sparse.diags(
            [up.reshape(-1)[ny:], south.reshape(-1)[1:], center.reshape(-1),
             north.reshape(-1)[:-1], dn.reshape(-1)[:-ny]],
            [-ny, -1, 0, 1, ny], shape=(P, P), format="csr")
        return mat


class TracerModule:
    """base of the oracle tracer modules: tend = processes + module sources"""

    tc = 1
    tracer_names = ()

    def __init__(self, model):
        self.model = model

    def comp_tend(self, time, y):
        m = self.model
        c = y.reshape(self.tc, m.nz, m.ny)
        return self._add_sources(time, c, m.tend_processes(time, c)).reshape(-1)

    def diag_extra(self, tr):
        """(nz, ny) array added to the Jacobian diagonal of tracer tr"""
        return np.zeros((self.model.nz, self.model.ny))

    def comp_jacobian(self, time, y=None):
        m = self.model
        up, south, center, north, dn = m.jac_diags(time)
        blocks = [m.diags_to_csr(up, south, center + self.diag_extra(tr), north, dn)
                  for tr in range(self.tc)]
        return sparse.block_diag(blocks, "csr")


class Iage(TracerModule):
    """iage module: ideal age with fast and 100x slower surface restoring
    (reference `py_driver_2d/iage.py:17-64`)"""

    tc = 2
    tracer_names = ("iage", "iage_slow_rest")

    def __init__(self, model):
        super().__init__(model)
        self.surf_restore_rate = 24.0 / 86400.0 * 10.0 / model.depth.delta[0]
        self.surf_slow_factor = 0.01

    def _add_sources(self, time, c, tend):
        tend[0, 0, :] -= self.surf_restore_rate * c[0, 0, :]
        tend[1, 0, :] -= self.surf_slow_factor * self.surf_restore_rate * c[1, 0, :]
        tend += 1.0 / (365.0 * 86400.0)
        return tend

    def diag_extra(self, tr):
        ex = np.zeros((self.model.nz, self.model.ny))
        rate = -self.surf_restore_rate
        ex[0, :] = rate if tr == 0 else self.surf_slow_factor * rate
        return ex

    def precond_matrix(self, time_range=(0.0, YEAR)):
        """I - prod_k (I - dt J(t_k)), dt = T/3 (iage.py:78-90), built with the same
        scipy.sparse expressions as the reference (the result is extremely sensitive to
        rounding, see `apply_precond_stable`)"""
        n = self.tc * self.model.nz * self.model.ny
        time_n = 3
        time_delta = (time_range[1] - time_range[0]) / time_n
        mat_id = sparse.identity(n)
        mat = sparse.identity(n)
        for time_ind in range(time_n):
            time = time_range[0] + (time_ind + 0.5) * time_delta
            mat_tmp = time_delta * self.comp_jacobian(time)
            mat = mat * (mat_id - mat_tmp)
        return mat_id - mat

    def apply_precond(self, v, time_range=(0.0, YEAR)):
        """M^-1 v = spsolve(I - prod(I - dt J_k), v) - v (iage.py:91-93)"""
        res = sp_linalg.spsolve(self.precond_matrix(time_range), v)
        return res - v


class Forced(TracerModule):
    """forced_{suff} module, one tracer (reference `py_driver_2d/forced.py:57-202`): surface
    restoring to a constant or to a field read from a file (or none); source-minus-sink
    constant / first-order decay / field read from a file (or none), the file source scaled
    down where it is a sink and the tracer is below `sink_thres`.  File fields are given as
    `(times, values)` already on the model axes (`forcing_on_model_axes`); they are interpolated
    in time as `utils.gen_forcing_fcn` does (utils.py:529-531: scipy's linear interp1d with
    extrapolation)."""

    tc = 1

    def __init__(self, model, surf_restore_opt="none", surf_restore_const=0.0, sms_opt="decay",
                 sms_decay_rate=0.0, sms_const=0.0, surf_restore_rate_10m=24.0 / 86400.0,
                 surf_restore_series=None, sms_series=None, sink_thres=None):
        super().__init__(model)
        if surf_restore_opt not in ("none", "const", "file") or sms_opt not in ("none", "const", "decay", "file"):
            raise ValueError("unknown forced option")
        if surf_restore_opt == "none" and sms_opt != "decay":
            raise ValueError("forced_sms_opt must be decay if forced_surf_restore_opt == none")   # forced.py:32-38
        self.surf_restore_opt = surf_restore_opt
        self.sms_opt = sms_opt
        self.surf_restore_rate = 10.0 / model.depth.delta[0] * surf_restore_rate_10m
        self.surf_restore_const = surf_restore_const
        self.sms_decay_rate = sms_decay_rate
        self.sms_const = sms_const
        self.sink_thres = sink_thres if sms_opt == "file" else None
        self.surf_restore_fcn = self.sms_fcn = None
        if surf_restore_opt == "file":
            self.surf_restore_fcn = interpolate.interp1d(
                surf_restore_series[0], surf_restore_series[1], axis=0, fill_value="extrapolate", assume_sorted=True)
        if sms_opt == "file":
            self.sms_fcn = interpolate.interp1d(
                sms_series[0], sms_series[1], axis=0, fill_value="extrapolate", assume_sorted=True)

    @property
    def state_dependent(self):
        return self.sink_thres is not None

    def _add_sources(self, time, c, tend):
        if self.surf_restore_opt != "none":
            target = self.surf_restore_const if self.surf_restore_opt == "const" else self.surf_restore_fcn(time)
            tend[0, 0, :] += self.surf_restore_rate * (target - c[0, 0, :])
        if self.sms_opt == "const":
            tend[0, :] += self.sms_const
        if self.sms_opt == "decay":
            tend[0, :] += -self.sms_decay_rate * c[0, :]
        if self.sms_opt == "file":
            sms = np.array(self.sms_fcn(time))
            if self.sink_thres is not None:
                tmp = (1.0 / self.sink_thres) * c[0, :]
                sms *= np.where((sms < 0.0) & (tmp > 0.0) & (tmp < 1.0), tmp, 1.0)   # forced.py:141-152
            tend[0, :] += sms
        return tend

    def diag_extra(self, tr, time=None, y=None):
        ex = np.zeros((self.model.nz, self.model.ny))
        if self.surf_restore_opt != "none":
            ex[0, :] += -self.surf_restore_rate
        if self.sms_opt == "decay":
            ex += -self.sms_decay_rate
        if self.sink_thres is not None:
            # forced.py:188-202
            sms = self.sms_fcn(time)
            sink_thres_r = 1.0 / self.sink_thres
            tmp = sink_thres_r * np.asarray(y).reshape(self.model.nz, self.model.ny)
            ex += np.where((sms < 0.0) & (tmp > 0.0) & (tmp < 1.0), sink_thres_r * sms, 0.0)
        return ex

    def comp_jacobian(self, time, y=None):
        m = self.model
        up, south, center, north, dn = m.jac_diags(time)
        return m.diags_to_csr(up, south, center + self.diag_extra(0, time, y), north, dn)

    # same three-step product formula as iage (forced.py:204-241); with a sink threshold the Jacobian
    # of time level k is evaluated at `states[k]`, the tracer at the end of that third of the year
    def precond_matrix(self, time_range=(0.0, YEAR), states=None):
        n = self.model.nz * self.model.ny
        time_n = 3
        time_delta = (time_range[1] - time_range[0]) / time_n
        mat_id = sparse.identity(n)
        mat = sparse.identity(n)
        for time_ind in range(time_n):
            time = time_range[0] + (time_ind + 0.5) * time_delta
            state = states[time_ind] if states is not None else np.zeros(n)
            mat_tmp = time_delta * self.comp_jacobian(time, state)
            mat = mat * (mat_id - mat_tmp)
        return mat_id - mat

    def apply_precond(self, v, time_range=(0.0, YEAR), states=None):
        return sp_linalg.spsolve(self.precond_matrix(time_range, states), v) - v


def forcing_on_model_axes(data, dims_in, dims_out, scalef=1.0):
    """the spatial part of `utils.gen_forcing_fcn` (utils.py:511-527): scale the field
    [time, dims...] and interpolate it linearly (with extrapolation) along every non-time axis
    whose coordinate differs from the model's"""
    data = scalef * np.asarray(data, dtype=np.float64)
    for axis in range(1, data.ndim):
        dim_in, dim_out = np.asarray(dims_in[axis - 1]), np.asarray(dims_out[axis - 1])
        if len(dim_in) != len(dim_out) or (dim_in != dim_out).any():
            data = interpolate.interp1d(dim_in, data, axis=axis, fill_value="extrapolate", assume_sorted=True)(dim_out)
    return data


class Phosphorus(TracerModule):
    """phosphorus module: po4, dop, pop coupled per cell (reference
    `py_driver_2d/phosphorus.py:17-172`): light- and po4-limited uptake (Michaelis-Menten),
    remineralisation of dop and pop back to po4, sinking of pop"""

    tc = 3
    tracer_names = ("po4", "dop", "pop")

    def __init__(self, model, **params):
        super().__init__(model)
        self.light_lim = np.outer(
            np.exp((-1.0 / 25.0) * model.depth.mid),
            np.exp(-1.0 * ((model.ypos.mid - 2.5e6) / 1.5e6) ** 2),
        )
        self.params = {
            "po4_halfsat": 0.5,
            "max_uptake_rate": 1.0 / (3.0 * 86400.0),
            "sigma": 0.67,
            "dop_remin_rate": 1.0 / (0.5 * 365.0 * 86400.0),
            "pop_remin_rate": 1.0 / (0.5 * 365.0 * 86400.0),
            "pop_sink_vel": 2.0 / 86400.0,
        }
        self.params.update(params)

    def po4_uptake(self, po4):
        lim = po4 / (po4 + self.params["po4_halfsat"])
        return self.params["max_uptake_rate"] * self.light_lim * lim

    def po4_uptake_deriv(self, po4):
        lim_d = self.params["po4_halfsat"] / (po4 + self.params["po4_halfsat"]) ** 2
        return self.params["max_uptake_rate"] * self.light_lim * lim_d

    def _add_sources(self, time, c, tend):
        prm = self.params
        uptake = self.po4_uptake(c[0])
        tend[0] -= uptake
        tend[1] += prm["sigma"] * uptake
        tend[2] += (1.0 - prm["sigma"]) * uptake
        dop_remin = prm["dop_remin_rate"] * c[1]
        pop_remin = prm["pop_remin_rate"] * c[2]
        tend[0] += dop_remin + pop_remin
        tend[1] -= dop_remin
        tend[2] -= pop_remin
        sink = np.zeros((self.model.nz + 1, self.model.ny))
        sink[1:-1, :] = prm["pop_sink_vel"] * c[2, :-1]
        tend[2] += self.model.depth.delta_r[:, np.newaxis] * (sink[:-1, :] - sink[1:, :])
        return tend

    def comp_jacobian(self, time, y):
        m = self.model
        prm = self.params
        P = m.nz * m.ny
        c = np.asarray(y).reshape(self.tc, m.nz, m.ny)
        up, south, center, north, dn = m.jac_diags(time)
        base = m.diags_to_csr(up, south, center, north, dn)
        jac = sparse.block_diag([base] * 3, "csr")
        upd = sparse.diags(self.po4_uptake_deriv(c[0]).reshape(-1))
        zero = sparse.csr_matrix((P, P))
        ident = sparse.identity(P)
        jac = jac + sparse.bmat([[-upd, zero, zero], [prm["sigma"] * upd, zero, zero],
                                 [(1.0 - prm["sigma"]) * upd, zero, zero]])
        rd = prm["dop_remin_rate"] * ident
        rp = prm["pop_remin_rate"] * ident
        jac = jac + sparse.bmat([[zero, rd, rp], [zero, -rd, zero], [zero, zero, -rp]])
        d0 = np.empty((m.nz, m.ny))
        d0[:] = -prm["pop_sink_vel"] * m.depth.delta_r[:, np.newaxis]
        d0[-1, :] = 0.0
        dm1 = np.empty((m.nz - 1, m.ny))
        dm1[:] = prm["pop_sink_vel"] * m.depth.delta_r[1:, np.newaxis]
        sink = sparse.diags((d0.reshape(-1), dm1.reshape(-1)), (0, -m.ny))
        jac = jac + sparse.bmat([[zero, zero, zero], [zero, zero, zero], [zero, zero, sink]])
        return jac.tocsr()


def apply_precond_stable(module, v, time_range=(0.0, YEAR), time_n=3, states=None):
    """The SAME operator as `Iage.apply_precond`, M^-1 = (I - A_0 A_1 A_2)^-1 - I with
    A_k = I - dt J(t_k), evaluated without forming the triple product: with
    u_1 = A_0^-1 u_0, u_2 = A_1^-1 u_1, u_3 = A_2^-1 u_2 and u_0 - u_3 = v one gets the
    time-periodic block system

        [ A_0   0  -I ] [u_1]   [v]
        [ -I  A_1   0 ] [u_2] = [0] ,     M^-1 v = -(u_3 + v)
        [  0  -I  A_2 ] [u_3]   [0]

    (exact-arithmetic identity: (I - A_0A_1A_2)^-1 v - v = -(I - G)^-1 v with
    G = A_2^-1 A_1^-1 A_0^-1).  The block matrix is a diagonally dominant M-matrix, so
    this form is backward stable, whereas the explicit product of the reference is
    roundoff dominated beyond toy grids (tests/test_oracle_precond.py measures both).
    The HIP preconditioner implements this form and is checked against it."""
    n = v.size
    dt = (time_range[1] - time_range[0]) / time_n
    eye = sparse.identity(n, format="csr")
    A = [eye - dt * (module.comp_jacobian(time_range[0] + (k + 0.5) * dt) if states is None else
                     module.comp_jacobian(time_range[0] + (k + 0.5) * dt, states[k]))
         for k in range(time_n)]
    rows = []
    for k in range(time_n):
        row = [None] * time_n
        row[k] = A[k]
        row[(k - 1) % time_n] = -eye
        rows.append(row)
    big = sparse.bmat(rows, format="csc")
    rhs = np.zeros(time_n * n)
    rhs[:n] = v
    u = sp_linalg.splu(big).solve(rhs)
    return -(u[(time_n - 1) * n:] + v)


def gen_init_iterate(model, knots_z=(55.0, 200.0), knots_v=(0.0, 2.0), tc=2):
    """`gen_init_iterate` pseudo-file (py_driver_2d/tracer_module_state.py:41-68 with
    the iage metadata of input/py_driver_2d/tracer_module_defs.yaml:9-16)"""
    col = np.interp(model.depth.mid, knots_z, knots_v)
    one = np.broadcast_to(col[:, np.newaxis], (model.nz, model.ny))
    return np.stack([one] * tc).copy()


def phosphorus_precond_matrix(module, po4, time_range=(0.0, YEAR)):
    """mat of `phosphorus.apply_precond_jacobian` (phosphorus.py:208-230) for time_n = 1:
    mat = I - (I - T J(T/2, po4)) with only po4 non-zero in the linearisation state"""
    m = module.model
    vals = np.zeros((module.tc, m.nz, m.ny))
    vals[0] = po4
    n = vals.size
    time_delta = time_range[1] - time_range[0]
    time_mid = time_range[0] + 0.5 * time_delta
    mat_id = sparse.identity(n)
    mat = mat_id * (mat_id - time_delta * module.comp_jacobian(time_mid, vals.reshape(-1)))
    return (mat_id - mat).tocsc()


def phosphorus_small_eigs(mat, sigma, k=5, v0=None):
    """`sp_linalg.eigs(mat, k=5, sigma=...)` sorted by magnitude.  The reference passes
    sigma=0.0 (phosphorus.py:239), i.e. shift-invert about an eigenvalue of the singular mat;
    a small positive sigma keeps the inner LU well conditioned."""
    e_vals, e_vects = sp_linalg.eigs(mat, k=k, sigma=sigma, v0=v0)
    order = np.argsort(np.abs(e_vals))
    return e_vals[order], e_vects[:, order]


def apply_precond_phosphorus(module, regions, po4, v, time_range=(0.0, YEAR), eig_sigma=0.02, shift=None):
    """restatement of `phosphorus.apply_precond_jacobian` (phosphorus.py:197-274): shifted
    double solve extrapolated to zero shift, then removal of the null-space component so that
    the (region weighted, summed over tracers) mean of the solution is zero.  `eig_sigma=0.0`
    reproduces the reference's eigs call, the default evaluates the same eigen-pair with a
    non-singular shift-invert.  Returns (result, e_vals, shift)."""
    mat = phosphorus_precond_matrix(module, po4, time_range)
    n = mat.shape[0]
    mat_id = sparse.identity(n, format="csc")
    e_vals, e_vects = phosphorus_small_eigs(mat, eig_sigma)
    null_comp = e_vects[:, 0]
    if max(abs(null_comp.imag)) > 1.0e-10 * max(abs(null_comp.real)):
        raise RuntimeError("1st eigenvector has non-trivial imaginary part")
    null_vect = null_comp.real
    if shift is None:
        shift = 0.5 * e_vals[1].real
    solve_tmp = sp_linalg.spsolve(mat - shift * mat_id, v)
    solve_vals = sp_linalg.spsolve(mat - (0.5 * shift) * mat_id, v)
    solve_vals = 2.0 * solve_vals - solve_tmp

    def mean(x):  # TracerModuleStateBase.mean: summed over the module's tracers
        return sum(regions.mean_of(plane) for plane in x.reshape(module.tc, -1))

    mask = regions.mask.reshape(-1)
    e_vect = null_vect.reshape(module.tc, -1) / regions.bcast(mean(null_vect)).reshape(-1)
    sol = solve_vals.reshape(module.tc, -1) - regions.bcast(mean(solve_vals)).reshape(-1) * e_vect
    sol = np.where(mask != 0, sol, solve_vals.reshape(module.tc, -1))
    return sol.reshape(-1) - v, e_vals, shift
