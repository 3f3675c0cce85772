"""ctypes binding of libnk2d.so (C ABI in include/nk2d.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C csrc`.  There
is deliberately no fallback: if the shared object is missing or a symbol is not
exported, loading raises.
"""

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("NK2D_LIB_PATH") or os.path.join(CSRC, "libnk2d.so")   # (the override: A/B builds of tools/)

MAX_TRACERS = 4
SCHED_WIDTH = 8

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_int64_p = ctypes.POINTER(ctypes.c_int64)


class Desc(ctypes.Structure):
    _fields_ = [
        ("nz", ctypes.c_int32),
        ("ny", ctypes.c_int32),
        ("tc", ctypes.c_int32),
        ("device_id", ctypes.c_int32),
        ("depth_edges", c_double_p),
        ("ypos_edges", c_double_p),
        ("vvel", c_double_p),
        ("wvel", c_double_p),
        ("hmix_coeff", c_double_p),
        ("bldepth_max", c_double_p),
        ("bldepth_min", ctypes.c_double),
        ("bld_tvals", ctypes.c_double * 4),
        ("bld_fvals", ctypes.c_double * 4),
        ("vmix_log_shallow", ctypes.c_double),
        ("vmix_log_deep", ctypes.c_double),
        ("vmix_half_width", ctypes.c_double),
        ("surf_rate", ctypes.c_double * MAX_TRACERS),
        ("surf_target", ctypes.c_double * MAX_TRACERS),
        ("decay_rate", ctypes.c_double * MAX_TRACERS),
        ("const_src", ctypes.c_double),
        ("t0", ctypes.c_double),
        ("t1", ctypes.c_double),
        ("rtol", ctypes.c_double),
        ("atol", ctypes.c_double),
        ("max_step_frac", ctypes.c_double),
        ("lin_tol", ctypes.c_double),
        ("module_kind", ctypes.c_int32),
        ("reserved0", ctypes.c_int32),
        ("phos_params", ctypes.c_double * 6),
        ("light_lim", c_double_p),
        ("restore_nrec", ctypes.c_int32),
        ("sms_nrec", ctypes.c_int32),
        ("restore_times", c_double_p),
        ("restore_vals", c_double_p),
        ("sms_times", c_double_p),
        ("sms_vals", c_double_p),
        ("sink_thres", ctypes.c_double),
    ]


class Stats(ctypes.Structure):
    _fields_ = [
        ("nfev", ctypes.c_int64),
        ("njev", ctypes.c_int64),
        ("nlu", ctypes.c_int64),
        ("nsteps", ctypes.c_int64),
        ("nrejected", ctypes.c_int64),
        ("nnewton", ctypes.c_int64),
        ("nsolve", ctypes.c_int64),
        ("nsweeps", ctypes.c_int64),
        ("nlaunch", ctypes.c_int64),
        ("seconds", ctypes.c_double),
        ("nresumed", ctypes.c_int64),
        ("nerr_checked", ctypes.c_int64),
        ("max_err", ctypes.c_double),
        ("nbarrier_timeouts", ctypes.c_int64),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


_vp = ctypes.c_void_p
_ci = ctypes.c_int
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_d = ctypes.c_double

# name -> (restype, argtypes); every symbol include/nk2d.h declares
SIGNATURES = {
    "nk2d_create": (_ci, [ctypes.POINTER(Desc), ctypes.POINTER(_vp)]),
    "nk2d_destroy": (None, [_vp]),
    "nk2d_last_error": (ctypes.c_char_p, [_vp]),
    "nk2d_version": (ctypes.c_char_p, []),
    "nk2d_set_region": (_ci, [_vp, c_int32_p, c_double_p, _i32]),
    "nk2d_vec_alloc": (_ci, [_vp, ctypes.POINTER(_vp)]),
    "nk2d_vec_free": (_ci, [_vp, _vp]),
    "nk2d_vec_upload": (_ci, [_vp, _vp, c_double_p]),
    "nk2d_vec_download": (_ci, [_vp, _vp, c_double_p]),
    "nk2d_vec_download_begin": (_ci, [_vp, _vp, ctypes.POINTER(ctypes.c_void_p)]),
    "nk2d_vec_download_end": (_ci, [_vp, _vp, c_double_p]),
    "nk2d_vec_copy": (_ci, [_vp, _vp, _vp]),
    "nk2d_vec_zero": (_ci, [_vp, _vp]),
    "nk2d_tend": (_ci, [_vp, _d, _vp, _vp]),
    "nk2d_vmix_coeff": (_ci, [_vp, _d, c_double_p]),
    "nk2d_jacobian_diags": (_ci, [_vp, _d, c_double_p]),
    "nk2d_set_lin_state": (_ci, [_vp, _vp]),
    "nk2d_shift_factor": (_ci, [_vp, _d, _d, ctypes.c_int32, c_double_p]),
    "nk2d_shift_solve": (_ci, [_vp, ctypes.c_int32, _vp, _vp]),
    "nk2d_jacobian_apply": (_ci, [_vp, _d, _vp, _vp]),
    "nk2d_shifted_solve": (_ci, [_vp, _d, _d, _d, _d, _vp, _vp, _vp, _vp, c_int32_p]),
    "nk2d_comp_fcn": (_ci, [_vp, _vp, _vp, ctypes.POINTER(Stats), c_double_p, _i64,
                            c_double_p, _i64, c_int64_p]),
    "nk2d_comp_fcn_frozen": (_ci, [_vp, _vp, _vp, ctypes.POINTER(Stats), c_double_p, _i64]),
    "nk2d_set_frozen_schedule": (_ci, [_vp, c_double_p, _i64]),
    "nk2d_last_schedule": (_ci, [_vp, c_double_p, _i64, c_int64_p]),
    "nk2d_frozen_fallbacks": (_ci, [_vp, c_int64_p]),
    "nk2d_frozen_resumes": (_ci, [_vp, c_int64_p]),
    "nk2d_get_counter": (_ci, [_vp, ctypes.c_char_p, c_int64_p]),
    "nk2d_schedule_fingerprint": (_ci, [_vp, c_double_p]),
    "nk2d_comp_fcn_hist": (_ci, [_vp, _vp, _vp, ctypes.POINTER(Stats), _i32, c_double_p, c_double_p]),
    "nk2d_precond_setup": (_ci, [_vp]),
    "nk2d_precond_setup_states": (_ci, [_vp, ctypes.POINTER(_vp)]),
    "nk2d_precond_apply": (_ci, [_vp, _vp, _vp]),
    "nk2d_dot": (_ci, [_vp, _vp, _vp, c_double_p]),
    "nk2d_axpby": (_ci, [_vp, _vp, c_double_p, _vp, c_double_p, _vp]),
    "nk2d_scale": (_ci, [_vp, _vp, _vp, c_double_p]),
    "nk2d_diff_scale": (_ci, [_vp, _vp, _vp, _vp, c_double_p]),
    "nk2d_lin_comb": (_ci, [_vp, _vp, _i32, ctypes.POINTER(_vp), c_double_p]),
    "nk2d_mgs": (_ci, [_vp, _vp, _i32, ctypes.POINTER(_vp), c_double_p]),
    "nk2d_apply_region_mask": (_ci, [_vp, _vp]),
    "nk2d_profile_reset": (_ci, [_vp, _i32]),
    "nk2d_profile_read": (_ci, [_vp, c_double_p, c_int64_p, c_int64_p, c_double_p, c_double_p, c_int64_p]),
    "nk2d_profile_totals": (_ci, [_vp, c_int64_p, c_double_p]),
    "nk2d_profile_shapes": (_ci, [_vp, c_int64_p, c_double_p]),
    "nk2d_profile_replay": (_ci, [_vp, _i32, _i32, c_double_p, c_double_p]),
    "nk2d_timer_begin": (_ci, [_vp]),
    "nk2d_timer_end": (_ci, [_vp, c_double_p]),
    "nk2d_jvp": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp, c_double_p, ctypes.POINTER(Stats)]),
    "nk2d_gmres_solve": (_ci, [_vp, _vp, _vp, _d, _i32, _i32, _vp, c_double_p, c_double_p, c_double_p,
                               c_double_p, c_int32_p]),
    "nk2d_multi_dot": (_ci, [_vp, _vp, _i32, ctypes.POINTER(_vp), c_double_p]),
    "nk2d_multi_axpy": (_ci, [_vp, _vp, _i32, ctypes.POINTER(_vp), c_double_p, _d]),
    "nk2d_set_norm_hook": (_ci, [_vp, _vp, _vp, _d]),
    "nk2d_set_norm_hook_vec": (_ci, [_vp, _vp, _vp, _d]),
    "nk2d_set_option": (_ci, [_vp, ctypes.c_char_p, _d]),
    "nk2d_sync": (_ci, [_vp]),
    "nk2d_stream": (_vp, [_vp]),
}

# double (*nk2d_norm_hook_fn)(void* user, double local_sum_of_squares)
NORM_HOOK = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_void_p, ctypes.c_double)
# void (*nk2d_norm_hook_vec_fn)(void* user, double* sums, int32_t n)
NORM_HOOK_VEC = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int32)

_lib = None


def build(force=False):
    """compile csrc/*.hip into csrc/libnk2d.so for gfx950 (hipcc cross-compiles)"""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    res = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"building libnk2d.so failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


def _one_hip_runtime():
    """PyTorch-ROCm ships its own HIP/HSA runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7, which its libraries
    ask for by FILE name).  Loaded after torch, libnk2d.so resolves `libamdhip64.so.7` to that copy and the process has one
    runtime; loaded BEFORE torch it would pull in /opt/rocm's copy, torch would add its own beside it, and the second
    runtime of the process finds no GPU ("No HIP GPUs are available" from a later torch.cuda call -- dist.ShardComm's
    device buffers, ModuleEngine.vec_tensor).  So: where torch is installed and not imported yet, its copy is loaded
    first, without importing torch, and both orders end with the same single runtime."""
    import importlib.util
    import sys

    if "torch" in sys.modules or os.environ.get("NK2D_HIP_RUNTIME", "") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def load():
    """load libnk2d.so and bind every declared symbol; raises when unavailable"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built (run "
            "`python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback"
        )
    _one_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
