"""Oracle: region-weighted state algebra, finite-difference JVP and the GMRES loop.

TEST INFRASTRUCTURE ONLY (see `oracle/__init__.py`).

A model state is a list with one flat fp64 vector per tracer module, each in
C order (tracer, depth, ypos).  Per-(module, region) scalars are arrays
`[ntm, nreg]` exactly as in the reference.

Reference lines restated here
  region mean matrix / dot_prod      nk_ooc/model_config.py:272-315,
                                     nk_ooc/tracer_module_state_base.py:379-388
  region broadcast (fill 1.0)        nk_ooc/tracer_module_state_base.py:502-515
  division = multiply by reciprocal  nk_ooc/model_state_base.py:290-303
  mod_gram_schmidt                   nk_ooc/model_state_base.py:365-377
  lin_comb                           nk_ooc/model_state_base.py:619-624
  comp_jacobian_fcn_state_prod       nk_ooc/model_state_base.py:492-527
  KrylovSolver._solve0 / solve       nk_ooc/krylov_solver.py:85-165
  _comp_krylov_basis_coeffs          nk_ooc/krylov_solver.py:168-181
"""

import numpy as np


class Regions:
    """region_mask / grid_weight handling of model_config.gen_grid_vars"""

    def __init__(self, mask, weight):
        mask = np.array(mask, dtype=np.int32)
        weight = np.array(weight, dtype=np.float64)
        mask[:] = np.where(weight == 0.0, 0, mask)
        weight[:] = np.where(mask == 0, 0.0, weight)
        self.mask = mask
        self.weight = weight
        self.nreg = int(mask.max())
        flat_m = mask.reshape(-1)
        flat_w = weight.reshape(-1)
        # one CSR row per region: w / sum_region(w), column indices ascending (model_config.py:292-315)
        import scipy.sparse

        indices, indptr, data = [], [0], []
        for r in range(self.nreg):
            idx = np.nonzero(flat_m == r + 1)[0]
            raw = flat_w[idx]
            sum_r = 1.0 / sum(raw)
            indices.extend(idx)
            indptr.append(len(indices))
            data.extend([sum_r * val for val in raw])
        self.mean_matrix = scipy.sparse.csr_array((data, indices, indptr), shape=(self.nreg, flat_w.size))

    def mean_of(self, plane_flat):
        """region_comp_mean_matrix.dot(plane): the CSR mat-vec sums every row in index order, one term
        after the other -- the association the reference's inner products have"""
        return self.mean_matrix.dot(plane_flat)

    def bcast(self, vals, fill=1.0):
        res = np.full(self.mask.shape, fill)
        for r, val in enumerate(vals):
            res = np.where(self.mask == r + 1, val, res)
        return res


class OracleModule:
    """one tracer module of the oracle: tendencies + regions"""

    def __init__(self, tm, regions, precond="reference"):
        self.tm = tm
        self.reg = regions
        self.tc = tm.tc
        self.nz = tm.model.nz
        self.ny = tm.model.ny
        self.precond_kind = precond
        self.replay = None  # optional schedule consumed by comp_fcn

    def planes(self, x):
        return x.reshape(self.tc, self.nz * self.ny)

    def dot(self, a, b):
        res = np.zeros(self.reg.nreg)
        for pa, pb in zip(self.planes(a), self.planes(b)):
            res += self.reg.mean_of(pa * pb)
        return np.array(res)

    def scale(self, x, vals):
        """x * ndarray[nreg] (region broadcast, fill 1.0)"""
        fac = self.reg.bcast(vals).reshape(-1)
        return (self.planes(x) * fac).reshape(-1)

    def mask_out(self, x):
        keep = (self.reg.mask != 0).reshape(-1)
        return np.where(keep, self.planes(x), 0.0).reshape(-1)

    def comp_fcn(self, x):
        from . import radau

        return self.mask_out(radau.comp_fcn(self.tm, x, replay=self.replay))

    def apply_precond(self, v):
        if self.precond_kind == "reference":
            return self.tm.apply_precond(v)
        if self.precond_kind == "phosphorus":
            # self.precond_po4: po4 at the end of the iterate's forward year (set by the caller)
            from .model import apply_precond_phosphorus

            return apply_precond_phosphorus(self.tm, self.reg, self.precond_po4, v)[0]
        from .model import apply_precond_stable

        # self.precond_states: tracer at the end of each third of the year, for modules whose Jacobian
        # depends on the state (forced module with a sink threshold; set by the caller)
        return apply_precond_stable(self.tm, v, states=getattr(self, "precond_states", None))


# ---- model-state level helpers (lists over modules, scalars [ntm, nreg]) -----------
def dot_prod(mods, a, b):
    return np.array([m.dot(x, y) for m, x, y in zip(mods, a, b)])


def norm(mods, a):
    return np.sqrt(dot_prod(mods, a, a))


def mul(mods, a, vals):
    return [m.scale(x, v) for m, x, v in zip(mods, a, vals)]


def div(mods, a, vals):
    return [m.scale(x, 1.0 / v) for m, x, v in zip(mods, a, vals)]


def lin_comb(mods, coeff, vec_list):
    """coeff [ntm, n, nreg]; vec_list[i] is a model state"""
    res = mul(mods, vec_list[0], coeff[:, 0, :])
    for j in range(1, coeff.shape[1]):
        term = mul(mods, vec_list[j], coeff[:, j, :])
        res = [r + t for r, t in zip(res, term)]
    return res


def mod_gram_schmidt(mods, w, basis):
    h = np.empty((len(mods), len(basis), mods[0].reg.nreg))
    for i, v in enumerate(basis):
        h[:, i, :] = dot_prod(mods, w, v)
        proj = mul(mods, v, h[:, i, :])
        w = [x - p for x, p in zip(w, proj)]
    return h, w


def jvp(mods, x, fcn, direction, comp_fcn=None):
    """finite-difference Jacobian-vector product; returns (w_raw, perturb_fcn, sigma)"""
    sigma = 1.0e-4 * norm(mods, x)
    sigma = np.where(sigma == 0.0, 1.0, sigma)
    step = mul(mods, direction, sigma)
    perturb = [a + b for a, b in zip(x, step)]
    if comp_fcn is None:
        pf = [m.comp_fcn(p) for m, p in zip(mods, perturb)]
    else:
        pf = comp_fcn(perturb)
    diff = [a - b for a, b in zip(pf, fcn)]
    return div(mods, diff, sigma), pf, sigma


def basis_coeffs(beta, h_mat):
    """argmin || beta e_1 - H c ||_2 per (module, region), np.linalg.lstsq as the reference"""
    shape = h_mat.shape
    coeff = np.zeros((shape[0], shape[2], shape[3]))
    rhs = np.zeros(shape[1])
    for im in range(shape[0]):
        for ir in range(shape[3]):
            rhs[0] = beta[im, ir]
            coeff[im, :, ir] = np.linalg.lstsq(h_mat[im, :, :, ir], rhs, rcond=None)[0]
    return coeff


def krylov_solve(mods, x, fcn, rel_tol=0.01, min_iter=0, max_iter=50, comp_fcn=None):
    """left-preconditioned GMRES on  J dx = -fcn  (x0 = 0, no restart).
    Returns (increment, trace); trace holds every per-iteration quantity the
    reference logs or checkpoints."""
    ntm, nreg = len(mods), mods[0].reg.nreg
    precond_fcn = [m.apply_precond(f) for m, f in zip(mods, fcn)]
    beta = norm(mods, precond_fcn)
    basis = [div(mods, [-p for p in precond_fcn], beta)]
    w_files = []
    trace = {"beta": beta, "h_mat": [], "coeff": [], "resid_norm": [], "w_raw": [],
             "precond_fcn": precond_fcn, "basis": basis, "w": w_files, "krylov_res": [],
             "sigma": [], "perturb_fcn": []}
    h_prev = None
    it = 0
    while True:
        j = it
        h_mat = np.zeros((ntm, j + 2, j + 1, nreg))
        if j > 0:
            h_mat[:, :-1, :-1, :] = h_prev
        w_raw, pf, sigma = jvp(mods, x, fcn, basis[j], comp_fcn)
        w_j = [m.apply_precond(v) for m, v in zip(mods, w_raw)]
        w_files.append([v.copy() for v in w_j])
        h_col, w_j = mod_gram_schmidt(mods, w_j, basis[: j + 1])
        h_mat[:, :-1, -1, :] = h_col
        h_mat[:, -1, -1, :] = norm(mods, w_j)
        w_j = div(mods, w_j, h_mat[:, -1, -1, :])
        h_prev = h_mat
        coeff = basis_coeffs(beta, h_mat)
        res = lin_comb(mods, coeff, basis[: j + 1])
        resid = lin_comb(mods, coeff, w_files)
        resid = [r + p for r, p in zip(resid, precond_fcn)]
        resid_norm = norm(mods, resid)
        trace["h_mat"].append(h_mat)
        trace["coeff"].append(coeff)
        trace["resid_norm"].append(resid_norm)
        trace["w_raw"].append(w_raw)
        trace["krylov_res"].append(res)
        trace["sigma"].append(sigma)
        trace["perturb_fcn"].append(pf)
        it += 1
        if ((it >= min_iter) & (resid_norm < rel_tol * beta)).all() or it >= max_iter:
            break
        basis.append(w_j)
    trace["iterations"] = it
    return res, trace
