"""Loader and ctypes prototypes for ludwig_amd/liblbmi.so (include/lbmi.h)."""

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# LBMI_LIB selects an alternative build (ablation variants, tools/)
LIB_PATH = os.environ.get("LBMI_LIB", os.path.join(_HERE, "liblbmi.so"))
UNIQUE_ID_BYTES = 128


class LbmiError(RuntimeError):
    pass


class Options(ctypes.Structure):
    _fields_ = [
        ("nvel", ctypes.c_int),
        ("ndist", ctypes.c_int),
        ("nlocal", ctypes.c_int * 3),
        ("nhalo", ctypes.c_int),
        ("device", ctypes.c_int),
        ("mode", ctypes.c_int),
        ("halo_scheme", ctypes.c_int),
        ("cartsz", ctypes.c_int),
        ("cartrank", ctypes.c_int),
        ("cartdim", ctypes.c_int),
        ("cartgrid", ctypes.c_int * 3),
        ("cartcoords", ctypes.c_int * 3),
    ]


class HydroPtrs(ctypes.Structure):
    _fields_ = [
        ("force", ctypes.c_void_p),
        ("status", ctypes.c_void_p),
        ("rho", ctypes.c_void_p),
        ("u", ctypes.c_void_p),
        ("eta", ctypes.c_void_p),
        ("nsite", ctypes.c_longlong),
    ]


class XOp(ctypes.Structure):
    """lbmi_xop_t: one point-to-point operation of the X exchange"""
    _fields_ = [
        ("kind", ctypes.c_int),
        ("peer", ctypes.c_int),
        ("buffer", ctypes.c_int),
        ("offset", ctypes.c_longlong),
        ("count", ctypes.c_longlong),
    ]


class FeSymm(ctypes.Structure):
    """lbmi_fe_symm_t"""
    _fields_ = [
        ("a", ctypes.c_double),
        ("b", ctypes.c_double),
        ("kappa", ctypes.c_double),
        ("mobility", ctypes.c_double),
        ("phi", ctypes.c_void_p),
        ("grad", ctypes.c_void_p),
        ("delsq", ctypes.c_void_p),
        ("nsite", ctypes.c_longlong),
    ]


# Every symbol declared in include/lbmi.h: (name, restype, argtypes)
_vp = ctypes.c_void_p
_i = ctypes.c_int
_d = ctypes.c_double
_pd = ctypes.POINTER(ctypes.c_double)
class IoFile(ctypes.Structure):
    """lbmi_io_file_t: one file of the i/o grid {nfile, 1, 1}."""
    _fields_ = [("nfile", ctypes.c_int), ("index", ctypes.c_int),
                ("file_nx", ctypes.c_int), ("file_x0", ctypes.c_int),
                ("periodic", ctypes.c_int * 3)]


SYMBOLS = [
    ("lbmi_options_default", _i, [ctypes.POINTER(Options)]),
    ("lbmi_create", _i, [ctypes.POINTER(Options), ctypes.POINTER(_vp)]),
    ("lbmi_free", _i, [_vp]),
    ("lbmi_last_error", ctypes.c_char_p, []),
    ("lbmi_nsite", _i, [_vp, ctypes.POINTER(ctypes.c_size_t)]),
    ("lbmi_nall", _i, [_vp, ctypes.POINTER(_i)]),
    ("lbmi_model", _i, [_i, _vp, _vp, _vp, _vp]),
    ("lbmi_set_relaxation", _i, [_vp, _i, _d, _d, _d]),
    ("lbmi_set_body_force", _i, [_vp, _pd]),
    ("lbmi_relaxation_rates", _i, [_vp, _pd]),
    ("lbmi_collide", _i, [_vp, _vp, ctypes.POINTER(HydroPtrs)]),
    ("lbmi_halo", _i, [_vp, _vp, _i]),
    ("lbmi_halo_x_count", _i, [_vp, _i, ctypes.POINTER(ctypes.c_size_t),
                               ctypes.POINTER(ctypes.c_size_t)]),
    ("lbmi_halo_x_pack", _i, [_vp, _vp, _i, _vp, _vp]),
    ("lbmi_halo_x_unpack", _i, [_vp, _vp, _i, _vp, _vp]),
    ("lbmi_halo_yz", _i, [_vp, _vp, _i]),
    ("lbmi_propagate", _i, [_vp, _vp, _vp]),
    ("lbmi_propagate_collide", _i, [_vp, _vp, _vp, ctypes.POINTER(HydroPtrs), _i]),
    ("lbmi_field_halo", _i, [_vp, _i, _vp]),
    ("lbmi_moments", _i, [_vp, _vp, _vp, _pd]),
    ("lbmi_lb_bind", _i, [_vp, _vp, _vp]),
    ("lbmi_lb_pointers", _i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    ("lbmi_lb_pointers_store", _i, [_vp, _vp, _vp]),
    ("lbmi_lb_collide", _i, [_vp, ctypes.POINTER(HydroPtrs)]),
    ("lbmi_lb_halo", _i, [_vp]),
    ("lbmi_lb_propagation", _i, [_vp]),
    ("lbmi_lb_flush", _i, [_vp]),
    ("lbmi_lb_mode_set", _i, [_vp, _i]),
    ("lbmi_lb_run", _i, [_vp, ctypes.POINTER(HydroPtrs), _i]),
    ("lbmi_lb_state", _i, [_vp, ctypes.POINTER(_i)]),
    ("lbmi_wall_map", _i, [_vp, ctypes.POINTER(_i), _vp]),
    ("lbmi_wall_links_build", _i, [_vp, _vp, ctypes.POINTER(_i),
                                   ctypes.POINTER(_i)]),
    ("lbmi_wall_links", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("lbmi_wall_velocity_set", _i, [_vp, _pd, _pd]),
    ("lbmi_wall_bbl", _i, [_vp]),
    ("lbmi_wall_bbl_arrays", _i, [_vp, _i, _vp, _vp, _vp, _vp, _pd, _pd, _vp]),
    ("lbmi_wall_momentum", _i, [_vp, _pd]),
    ("lbmi_wall_links_set", _i, [_vp, _i, _vp, _vp, _vp, _vp]),
    ("lbmi_wall_fnet_bind", _i, [_vp, _vp]),
    ("lbmi_wall_slip_links_set", _i, [_vp, _vp, _vp, _vp, _pd]),
    ("lbmi_wall_status_set", _i, [_vp, _vp]),
    ("lbmi_wall_slip_set", _i, [_vp, _vp, _pd, _pd]),
    ("lbmi_wall_slip_links", _i, [_vp, _vp, _vp, _vp]),
    ("lbmi_wall_bbl_slip_arrays", _i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _pd, _vp]),
    ("lbmi_lb_phi_to_field", _i, [_vp, _vp]),
    ("lbmi_lb_collide_binary", _i, [_vp, ctypes.POINTER(HydroPtrs),
                                    ctypes.POINTER(FeSymm)]),
    ("lbmi_lb_collide_fe", _i, [_vp, ctypes.POINTER(HydroPtrs),
                                ctypes.POINTER(FeSymm)]),
    ("lbmi_lb_memcpy_h2d", _i, [_vp, _vp]),
    ("lbmi_lb_memcpy_d2h", _i, [_vp, _vp]),
    ("lbmi_lb_dirty", _i, [_vp]),
    ("lbmi_lb_moments", _i, [_vp, _vp, _pd]),
    ("lbmi_lb_density", _i, [_vp, _vp]),
    ("lbmi_field_stats", _i, [_vp, _vp, _vp, _pd]),
    ("lbmi_lb_hydro_sync", _i, [_vp]),
    ("lbmi_noise_set", _i, [_vp, _vp, ctypes.c_longlong, _d, _i]),
    ("lbmi_hydro_field_dirty", _i, [_vp, _vp]),
    ("lbmi_hydro_field_set", _i, [_vp, _vp, _i, _pd]),
    ("lbmi_field_halo_n", _i, [_vp, _i, _i, _vp]),
    ("lbmi_fe_scheme_set", _i, [_vp, _i, _i]),
    ("lbmi_field_interior_copy", _i, [_vp, _i, _vp, _vp]),
    ("lbmi_field_grad_7pt", _i, [_vp, _vp, _vp, _vp]),
    ("lbmi_field_grad_27pt", _i, [_vp, _vp, _vp, _vp]),
    ("lbmi_field_grad", _i, [_vp, _vp, _vp, _vp]),
    ("lbmi_symmetric_force", _i, [_vp, _d, _d, _d, _vp, _vp, _vp, _vp]),
    ("lbmi_cahn_hilliard", _i, [_vp, _d, _d, _d, _d, _vp, _vp, _vp, _vp]),
    ("lbmi_symmetric_step", _i, [_vp, _d, _d, _d, _d, _vp, _vp, _vp, _vp, _i]),
    ("lbmi_symmetric_step_periodic", _i, [_vp, _d, _d, _d, _d, _vp, _vp, _vp,
                                          _vp, _i]),
    ("lbmi_symmetric_step_grad", _i, [_vp, _d, _d, _d, _d, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _i]),
    ("lbmi_symmetric_lb_step", _i, [_vp, ctypes.POINTER(HydroPtrs), _vp, _d, _d,
                                    _d, _d, _vp, _vp]),
    ("lbmi_symmetric_lb_collide", _i, [_vp, ctypes.POINTER(HydroPtrs), _vp, _d, _d,
                                       _d, _d, _vp, _vp]),
    ("lbmi_lb_records_pack", _i, [_vp, _vp]),
    ("lbmi_lb_records_unpack", _i, [_vp, _vp]),
    ("lbmi_lb_io_write", _i, [_vp, ctypes.c_char_p, _i, _i, _i]),
    ("lbmi_io_format_set", _i, [_vp, _i]),
    ("lbmi_io_metadata_write_fmt", _i, [ctypes.c_char_p, ctypes.c_char_p, _i, _i,
                                        ctypes.POINTER(_i), _i]),
    ("lbmi_lb_io_read", _i, [_vp, ctypes.c_char_p, _i, _i, _i]),
    ("lbmi_io_metadata_write", _i, [ctypes.c_char_p, ctypes.c_char_p, _i,
                                    ctypes.POINTER(_i)]),
    ("lbmi_io_filename", _i, [ctypes.c_char_p, ctypes.c_char_p, _i,
                              ctypes.c_char_p, ctypes.c_size_t]),
    ("lbmi_io_file_set", _i, [_vp, ctypes.POINTER(IoFile)]),
    ("lbmi_io_metadata_write_file", _i, [ctypes.c_char_p, ctypes.c_char_p, _i, _i,
                                         ctypes.POINTER(_i), _i, ctypes.POINTER(IoFile)]),
    ("lbmi_io_filename_fmt", _i, [ctypes.c_char_p, ctypes.c_char_p, _i, _i,
                                  ctypes.c_char_p, ctypes.c_size_t]),
    ("lbmi_io_single_metadata_write", _i, [ctypes.c_char_p, ctypes.c_char_p, _i, _i,
                                           ctypes.POINTER(_i), _i, _i,
                                           ctypes.POINTER(_i)]),
    ("lbmi_synchronize", _i, [_vp]),
    ("lbmi_stream", _i, [_vp, ctypes.POINTER(_vp)]),
    ("lbmi_set_stream", _i, [_vp, _vp]),
    ("lbmi_timing", _i, [_vp, _i]),
    ("lbmi_timing_read", _i, [_vp, _pd, ctypes.POINTER(_i)]),
    ("lbmi_timing_read_detail", _i, [_vp, _pd, ctypes.POINTER(_i)]),
    ("lbmi_tune", _i, [_vp, ctypes.c_char_p, _i]),
    ("lbmi_comm_unique_id", _i, [_vp]),
    ("lbmi_comm_init", _i, [_vp, _vp]),
    ("lbmi_comm_free", _i, [_vp]),
    ("lbmi_comm_info", _i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i),
                            ctypes.POINTER(_i)]),
    ("lbmi_x_schedule", _i, [ctypes.POINTER(Options), _i, _i, _vp, _i,
                             ctypes.POINTER(_i)]),
    ("lbmi_x_schedule_dim", _i, [ctypes.POINTER(Options), _i, _i, _i, _vp, _i,
                                 ctypes.POINTER(_i)]),
    ("lbmi_ring_create", _i, [_i, ctypes.POINTER(_vp)]),
    ("lbmi_comm_init_ring", _i, [_vp, _vp]),
    ("lbmi_ring_free", _i, [_vp]),
    ("lbmi_ring_abort", _i, [_vp]),
]


def build(force=False):
    """Compile liblbmi.so for gfx950 with hipcc + gcc (in-tree)."""
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.run(args + ["clean"], check=True)
    subprocess.run(args, check=True)
    if not os.path.exists(LIB_PATH):
        raise LbmiError("build did not produce " + LIB_PATH)


_lib = None


def library():
    """Load liblbmi.so; fail loudly if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LbmiError(
            "%s not found: build it with `python -c 'import __graft_entry__ "
            "as g; g.build()'` or `make -C ludwig_amd/csrc`. There is no CPU "
            "fallback." % LIB_PATH)
    # torch ships the HIP runtime and RCCL under the same sonames as
    # /opt/rocm; importing it first makes the process use ONE runtime.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        msg = library().lbmi_last_error().decode("utf-8", "replace")
        raise LbmiError("liblbmi error %d: %s" % (rc, msg))
