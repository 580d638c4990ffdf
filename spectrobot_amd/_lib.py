"""ctypes binding of libspectrobot_hip.so (include/spectrobot_hip.h).

There is no fallback: if the HIP library is missing or does not load, importing
this module raises.  torch is imported first so that the process holds ONE HIP
runtime (torch bundles libamdhip64.so.7; loading ours afterwards resolves to the
same SONAME) and device pointers / streams can be shared with torch tensors.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below, see docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPECTROBOT_HIP_LIB: a variant library built by build.py --out (tuning sweeps); default = the in-tree build
LIB_PATH = os.environ.get("SPECTROBOT_HIP_LIB") or os.path.join(_HERE, "lib", "libspectrobot_hip.so")

SR_OK = 0
SR_ERR_ARG, SR_ERR_LIMIT, SR_ERR_HIP, SR_ERR_NODEVICE, SR_ERR_UNSUPPORTED, SR_ERR_TABLE = -1, -2, -3, -4, -5, -6
IMXSIG = 13010

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)


class SpectRobotHipError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = lib.sr_strerror(status).decode()
        if status == SR_ERR_HIP or status == SR_ERR_UNSUPPORTED or status == SR_ERR_NODEVICE:
            extra = lib.sr_last_error().decode()
            if extra:
                msg += " (" + extra + ")"
        super().__init__("%s: %s [%d]" % (where, msg, status))


class LinesDesc(C.Structure):
    _fields_ = [("n_lines", C.c_int64), ("freq", dp), ("a_coeff", dp), ("e_lower", dp), ("g_up", dp),
                ("g_lo", dp), ("air_broad", dp), ("t_dep_broad", dp), ("lev_up", ip), ("lev_lo", ip)]


class IsoMolecDesc(C.Structure):
    _fields_ = [("mol", C.c_int), ("iso", C.c_int), ("mm", C.c_double), ("n_levels", C.c_int),
                ("level_energy", dp)]


class GridDesc(C.Structure):
    _fields_ = [("w0", C.c_double), ("step", C.c_double), ("n_grid", C.c_int64)]


class LayersDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("temps", dp), ("press", dp), ("tvib", dp), ("q_part", dp)]


class LosDesc(C.Structure):
    _fields_ = [("n_rays", C.c_int), ("n_gas", C.c_int), ("seg_off", ip), ("seg_layer", ip), ("pt_off", ip),
                ("x", dp), ("nd", dp), ("vmr", dp), ("col_scale", dp), ("los_order", C.c_int),
                ("solo_absorption", C.c_int), ("init_mode", C.c_int), ("t_init", C.c_double), ("w0", C.c_double),
                ("step", C.c_double), ("g_lo", C.c_int64)]


class OeDesc(C.Structure):
    _fields_ = [("n_obs", C.c_int32), ("obs", dp), ("noise", dp), ("mask", C.POINTER(C.c_uint8)), ("sa_inv", dp),
                ("x_apriori", dp), ("lambda_lm", C.c_double)]


class LoopDesc(C.Structure):
    _fields_ = [("max_it", C.c_int32), ("chi_threshold", C.c_double), ("positive", C.POINTER(C.c_uint8)),
                ("n_dof_par", C.c_int32)]


# every symbol include/spectrobot_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "sr_strerror": (C.c_char_p, [C.c_int]),
    "sr_last_error": (C.c_char_p, []),
    "sr_abi_version": (C.c_int, []),
    "sr_set_device": (C.c_int, [C.c_int]),
    "sr_device_info": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), dp]),
    "sr_recommended_hw_queues": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sr_humliv_bb": (C.c_int, [dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, dp]),
    "sr_sum_all_lines": (C.c_int, [dp, C.c_int64, dp, ip, ip, C.c_int, C.c_int]),
    "sr_bd_tips_2003": (C.c_int, [C.c_int, C.c_int, dp, dp, dp]),
    "sr_calc_partition_sum": (C.c_int, [C.c_int, C.c_int, dp, C.c_int, dp]),
    "sr_curgod": (C.c_int, [C.c_int, dp, dp, dp, dp, ip, C.c_int, dp]),
    "sr_lineset_create": (C.c_int, [C.POINTER(LinesDesc), C.POINTER(IsoMolecDesc), C.POINTER(GridDesc),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "sr_lineset_destroy": (C.c_int, [C.c_void_p]),
    "sr_abscoeff_layers_dev": (C.c_int, [C.c_void_p, C.POINTER(LayersDesc), C.c_int64, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "sr_abscoeff_layers": (C.c_int, [C.c_void_p, C.POINTER(LayersDesc), C.c_int64, C.c_int64, dp, dp]),
    "sr_glevel_pairs_dev": (C.c_int, [C.c_void_p, C.POINTER(LayersDesc), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "sr_gcoeff_levels_dev": (C.c_int, [C.c_void_p, C.POINTER(LayersDesc), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "sr_set_level_route": (C.c_int, [C.c_int]),
    "sr_last_level_tables_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "sr_glevel_combine_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, ip, dp, dp,
                                        C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sr_gcoeff_layers_dev": (C.c_int, [C.c_void_p, C.POINTER(LayersDesc), C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                                       C.c_void_p]),
    "sr_abscoeff_level_dev": (C.c_int, [C.c_void_p, C.POINTER(LayersDesc), C.c_int, C.c_int64, C.c_int64,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "sr_lut_interp_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, ip, dp, dp, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "sr_los_columns": (C.c_int, [C.POINTER(LosDesc), dp]),
    "sr_limb_rays_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.POINTER(LosDesc), C.c_void_p,
                                   C.c_void_p]),
    "sr_los_create": (C.c_int, [C.POINTER(LosDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "sr_los_destroy": (C.c_int, [C.c_void_p]),
    "sr_los_create_par": (C.c_int, [C.POINTER(LosDesc), C.c_int, C.c_int, ip, dp, C.POINTER(C.c_void_p)]),
    "sr_los_set_vmr": (C.c_int, [C.c_void_p, dp, C.c_void_p]),
    "sr_los_refresh_columns": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sr_limb_rays_jac_los_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "sr_limb_rays_los_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                       C.c_void_p]),
    "sr_retrieval_forward_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64, dp, C.c_double,
                                           C.c_double, dp, dp, C.c_int, C.c_double, C.c_int, dp, C.c_void_p, dp, C.c_void_p]),
    "sr_retrieval_step_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64, dp, C.c_double,
                                        C.c_double, dp, dp, C.c_int, C.c_double, C.c_int, dp, C.c_void_p, dp, C.POINTER(OeDesc), dp,
                                        C.POINTER(C.c_int32), dp, dp, dp, C.c_void_p]),
    "sr_retrieval_loop_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64, dp, C.c_double,
                                        C.c_double, dp, dp, C.c_int, C.c_double, C.c_int, dp, C.c_void_p, dp, C.POINTER(OeDesc),
                                        C.POINTER(LoopDesc), dp, dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp, dp, C.c_void_p]),
    "sr_limb_step_dev": (C.c_int, [C.c_void_p, C.POINTER(LayersDesc), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "sr_limb_rays_jac_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.POINTER(LosDesc), C.c_int, ip, dp,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "sr_limb_rays_jacobians_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                             C.POINTER(LosDesc), ip, C.c_int, C.c_int, ip, dp, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "sr_limb_rays_jac_layer_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                             C.POINTER(LosDesc), C.c_void_p, C.c_void_p]),
    "sr_radiance_rays_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, ip, ip, dp, C.c_int,
                                       C.c_void_p, C.c_void_p]),
    "sr_radiance_jac_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, ip, ip, dp, dp, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "sr_radiance_jac_layer_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                            C.c_int, ip, ip, dp, C.c_void_p, C.c_void_p]),
    "sr_hires_to_lowres_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_double, dp, dp, C.c_int,
                                         C.c_double, C.c_int, dp, C.c_void_p]),
    "sr_hires_to_lowres_shard_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_double, C.c_double, dp, dp,
                                               C.c_int, C.c_double, C.c_int, dp, C.c_void_p]),
    "sr_set_points_per_lane": (C.c_int, [C.c_int]),
    "sr_set_band_fusion": (C.c_int, [C.c_int]),
    "sr_far_field_truncation_bound": (C.c_double, []),
    "sr_los_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "sr_set_jac_layer_mode": (C.c_int, [C.c_int]),
    "sr_last_limb_route": (C.c_int, []),
    "sr_set_far_field": (C.c_int, [C.c_int]),
    "sr_set_overlap": (C.c_int, [C.c_int]),
    "sr_set_kernel_repeat": (C.c_int, [C.c_int, C.c_int]),
    "sr_set_table_budget": (C.c_int, [C.c_int64]),
    "sr_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "sr_set_counting": (C.c_int, [C.c_int]),
    "sr_set_timing": (C.c_int, [C.c_int]),
    "sr_lineset_set_bounds_temps": (C.c_int, [C.c_void_p, dp, C.c_int]),
    "sr_lineset_set_linear_weights": (C.c_int, [C.c_void_p, C.c_int]),
    "sr_last_eval_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError("libspectrobot_hip.so is not built (%s missing): run `python -c \"import "
                      "__graft_entry__ as g; g.build()\"` or `python spectrobot_amd/build.py`; there is no "
                      "CPU fallback" % LIB_PATH)
lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SYMBOLS.items():
    _f = getattr(lib, _name)  # AttributeError here = header and library out of sync
    _f.restype = _res
    _f.argtypes = _args


def check(status, where):
    if status != SR_OK:
        raise SpectRobotHipError(status, where)
