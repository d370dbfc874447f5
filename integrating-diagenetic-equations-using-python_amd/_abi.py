"""ctypes binding of libmarl_hip.so (C ABI: include/marl_hip.h, include/marl_params.h).

This is the whole "thin ctypes shim" between the host Python layer and the HIP kernels.  There
is no fallback: if the shared library has not been built (``python -c "import __graft_entry__ as g;
g.build()"`` or ``make -C integrating-diagenetic-equations-using-python_amd/csrc``) loading raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MARL_HIP_LIBRARY: another BUILD of the same library - kernel-lab A/B runs on the GPU box; never a different implementation)
LIB_PATH = os.environ.get("MARL_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libmarl_hip.so")



def lab_options():
    """[(name, value)] from MARL_HIP_OPTIONS="name=value,..." - library options (marl_set_option) applied to every context the
    Python layer creates.  For kernel-lab A/B runs on one box; bench.py records the variable in its output line."""
    out = []
    for item in os.environ.get("MARL_HIP_OPTIONS", "").split(","):
        if item.strip():
            name, _, value = item.partition("=")
            out.append((name.strip(), int(value)))
    return out


def library_fingerprint():
    """(path, first 16 hex digits of the sha256) of the library in use."""
    import hashlib
    with open(LIB_PATH, "rb") as f:
        return LIB_PATH, hashlib.sha256(f.read()).hexdigest()[:16]


NFIELDS = 5
NEVENTS = 7
LAYOUT_FIELD_MAJOR = 0
LAYOUT_TILED = 1

# field order == struct marl_params (include/marl_params.h) == ctor kwargs of the reference's
# LMAHeureuxPorosityDiff (marlpde/LHeureux_model.py:12-16) + grid/mask numbers
PARAM_DOUBLES = (
    "CA0", "CC0", "cCa0", "cCO30", "Phi0", "sedimentationrate", "Xstar", "Tstar",
    "k1", "k2", "k3", "k4", "m1", "m2", "n1", "n2", "b", "beta", "rhos", "rhow", "rhos0",
    "KA", "KC", "muA", "D0Ca", "PhiNR", "PhiInfty", "PhiIni", "DCa", "DCO3",
    "length", "shallow_limit", "deep_limit",
)


class MarlParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in PARAM_DOUBLES] + [("FV_switch", C.c_int32), ("dPhi_variable", C.c_int32)]


class MarlStats(C.Structure):
    _fields_ = [
        ("nfev", C.c_int64), ("n_accepted", C.c_int64), ("n_rejected", C.c_int64),
        ("status", C.c_int32), ("reserved", C.c_int32),
        ("t", C.c_double), ("h_next", C.c_double),
        ("event_value", C.c_double * NEVENTS), ("n_events", C.c_int64 * NEVENTS),
        ("njev", C.c_int64), ("nlu", C.c_int64),
    ]


# name -> (restype, argtypes); every symbol include/marl_hip.h declares
_P = C.c_void_p
_D = C.c_double
_I = C.c_int
_L = C.c_int64
PROTOTYPES = {
    "marl_ctx_create": (_I, [C.POINTER(MarlParams), _L, _L, _I, C.POINTER(_P)]),
    "marl_ctx_set_params": (_I, [_P, C.POINTER(MarlParams), _L]),
    "marl_ctx_destroy": (None, [_P]),
    "marl_last_error": (C.c_char_p, [_P]),
    "marl_set_stream": (_I, [_P, _P]),
    "marl_synchronize": (_I, [_P]),
    "marl_set_option": (_I, [_P, C.c_char_p, _L]),
    "marl_get_constants": (_I, [_P, _L, C.POINTER(_D)]),
    "marl_state_doubles": (_L, [_P, _I]),
    "marl_rhs": (_I, [_P, _D, _P, _P]),
    "marl_rhs_dev": (_I, [_P, _D, _P, _P, _I]),
    "marl_events": (_I, [_P, _P, _P]),
    "marl_events_dev": (_I, [_P, _P, _I, _P]),
    "marl_convert_layout_dev": (_I, [_P, _P, _P, _I, _I]),
    "marl_debug_math": (_I, [_P, _I, _P, _P, _L, _D]),
    "marl_integrate_rk4": (_I, [_P, _P, _D, _L]),
    "marl_integrate_rk4_dev": (_I, [_P, _P, _I, _D, _L]),
    "marl_sweep_rk4_dev": (_I, [_P, _P, _P, _L]),
    "marl_integrate_rk45": (_I, [_P, _P, _D, _D, _D, _D, _D, _P, _L, _P, _P, _L, _L, C.POINTER(MarlStats)]),
    "marl_integrate_rk45_dev": (_I, [_P, _P, _I, _D, _D, _D, _D, _D, _L, C.POINTER(MarlStats)]),
    "marl_sweep_rk45_dev": (_I, [_P, _P, _D, _D, _D, _D, _D, _L, C.POINTER(MarlStats)]),
    "marl_integrate_radau": (_I, [_P, _P, _D, _D, _D, _D, _D, _P, _P, _L, _P, _P, _L, _L, C.POINTER(MarlStats)]),
    "marl_integrate_bdf": (_I, [_P, _P, _D, _D, _D, _D, _D, _P, _P, _L, _P, _P, _L, _L, C.POINTER(MarlStats)]),
    "marl_sweep_radau_dev": (_I, [_P, _P, _D, _D, _D, _D, _D, _P, _L, C.POINTER(MarlStats)]),
    "marl_sweep_radau_events_dev": (_I, [_P, _P, _D, _D, _D, _D, _D, _P, _L, _P, _L, C.POINTER(MarlStats)]),
    "marl_ctx_create_slab": (_I, [C.POINTER(MarlParams), _L, _L, _L, _L, _I, C.POINTER(_P)]),
    "marl_slab_load": (_I, [_P, _P]),
    "marl_slab_store": (_I, [_P, _P]),
    "marl_slab_pack": (_I, [_P, _I, _P, _P]),
    "marl_slab_unpack": (_I, [_P, _I, _P, _P]),
    "marl_slab_rhs0": (_I, [_P]),
    "marl_slab_monitors": (_I, [_P, _P]),
    "marl_slab_init_control": (_I, [_P, _P, _L, _D, _D, _D, _D, _D, _L]),
    "marl_slab_attempt": (_I, [_P, _P]),
    "marl_slab_control": (_I, [_P, _P, _L]),
    "marl_slab_status": (_I, [_P, C.POINTER(MarlStats)]),
    "marl_slab_comm_probe": (_I, [C.c_char_p]),
    "marl_slab_comm_id": (_I, [C.c_char_p, C.c_char_p]),
    "marl_slab_comm_init": (_I, [_P, C.c_char_p, C.c_char_p, _I, _I]),
    "marl_slab_exchange": (_I, [_P, _I]),
    "marl_slab_run": (_I, [_P, C.POINTER(MarlStats)]),
}

_lib = None


class MarlError(RuntimeError):
    pass


def load():
    """Load libmarl_hip.so (once).  Raises if the HIP extension is missing - there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MarlError(f"{LIB_PATH} not found: build the HIP extension first (__graft_entry__.build()); "
                            "this package has no CPU fallback")
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
        # libamdhip64.so.7, the same as /opt/rocm's).  Loaded first, it satisfies this library's
        # DT_NEEDED by SONAME; loaded second, it would come up as a SECOND runtime beside the system
        # one and torch.cuda would find no device.  So torch (the device-buffer provider) goes first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(ctx, rc, what):
    """Raise MarlError with the library's message for a negative return code."""
    if rc < 0:
        msg = load().marl_last_error(ctx)
        raise MarlError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
    return rc
