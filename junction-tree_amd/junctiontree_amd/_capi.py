"""ctypes binding of libjtprop.so (include/jtprop.h).  No torch, no pybind: a plain C ABI.

The library is required: if it cannot be loaded, or no MI355X is visible when a plan needs
the device, every entry point raises - there is no CPU fallback in this package.
"""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libjtprop.so")
LIB_PATH = os.environ.get("JTPROP_LIB", LIB_PATH)      # developer aid: A/B runs of two builds

JTP_OK, JTP_EINVAL, JTP_EHIP, JTP_ECOMM, JTP_ENOMEM, JTP_EUNSUPPORTED = 0, -1, -2, -3, -4, -5
JTP_F32, JTP_F64 = 0, 1
JTP_PLAN_ONLY = 1
JTP_SPLIT_VARIANTS = 2
JTP_KEEP_ROOT = 4
JTP_LEVEL_LAUNCHES = 8
JTP_SHARE_POTENTIALS = 32
JTP_FLOW_TICKETS = 16
JTP_MULTISET = 64
JTP_NO_COMPACT = 128
N_VARIANTS = 23
MAX_VARS = 32            # variables per node the C ABI takes (JT_MAX_VARS); engine.Plan keeps one-state variables beyond that on the host


class TreeDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("n_vars", C.c_int32),
        ("var_card", C.POINTER(C.c_int32)),
        ("n_cliques", C.c_int32),
        ("n_nodes", C.c_int32),
        ("node_var_off", C.POINTER(C.c_int32)),
        ("node_var_ids", C.POINTER(C.c_int32)),
        ("parent_clique", C.POINTER(C.c_int32)),
        ("parent_sep", C.POINTER(C.c_int32)),
        ("dtype", C.c_int32),
        ("device", C.c_int32),
        ("n_batch", C.c_int32),
        ("n_ranks", C.c_int32),
        ("rank", C.c_int32),
        ("clique_owner", C.POINTER(C.c_int32)),
        ("flags", C.c_uint32),
        ("lds_budget", C.c_int32),
        ("block_log2", C.c_int32),
        ("layout_policy", C.c_int32),
        ("cover_off", C.POINTER(C.c_int32)),
        ("cover_ids", C.POINTER(C.c_int32)),
        ("fold_n", C.c_int32),
        ("fold_pad", C.c_int32),
        ("fold_cliques", C.POINTER(C.c_int32)),
        ("fold_var_off", C.POINTER(C.c_int32)),
        ("fold_var_ids", C.POINTER(C.c_int32)),
    ]


class Factor(C.Structure):
    _fields_ = [
        ("host", C.c_void_p),
        ("n_vars", C.c_int32),
        ("dtype", C.c_int32),
        ("var_ids", C.POINTER(C.c_int32)),
        ("shape", C.POINTER(C.c_int64)),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("n_launches", C.c_int32),
        ("n_messages", C.c_int32),
        ("n_tasks", C.c_int32),
        ("algorithmic_bytes", C.c_double),
        ("collect_ms", C.c_double),
        ("distribute_ms", C.c_double),
        ("kernel_ms", C.c_double * 32),
        ("kernel_bytes", C.c_double * 32),
        ("kernel_launches", C.c_int32 * 32),
        ("flow_fallbacks", C.c_int32),
        ("launch_mode", C.c_int32),
        ("tickets_used", C.c_int32),
        ("flow_propagates", C.c_int32),
        ("device_bytes", C.c_double),
        ("storage_dtype", C.c_int32),
        ("foreign_seen", C.c_int32),
        ("f64_flops", C.c_double),
        ("f64_insts", C.c_double),
        ("algorithmic_bytes_full", C.c_double),
        ("fixed_bytes", C.c_double),
        ("n_unit_cliques", C.c_int32),
        ("n_static_tables", C.c_int32),
        ("flight_board", C.c_int32),
        ("lean_refused", C.c_int32),
    ]


class JtpError(RuntimeError):
    """HIP / RCCL / allocation failure reported by libjtprop."""


class UnsupportedStructure(ValueError):
    """JTP_EUNSUPPORTED: a well-formed structure outside the engine's limits (a table beyond 2^31 entries, sub-boxes that
    no layout fits into the LDS of a CU, ...).  A ValueError, as every structural refusal is."""


_lib = None

# name -> (restype, argtypes); every symbol include/jtprop.h declares
SYMBOLS = {
    "jtp_plan_create": (C.c_int, [C.POINTER(TreeDesc), C.POINTER(C.c_void_p)]),
    "jtp_plan_destroy": (None, [C.c_void_p]),
    "jtp_plan_describe": (C.c_char_p, [C.c_void_p]),
    "jtp_set_potential": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                    C.POINTER(C.c_int64), C.c_int32]),
    "jtp_set_potential_product": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Factor)]),
    "jtp_set_potential_products": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jtp_fill_synthetic": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.POINTER(C.c_double)]),
    "jtp_set_evidence": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "jtp_propagate": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "jtp_sync": (C.c_int, [C.c_void_p]),
    "jtp_get_belief": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    "jtp_get_marginal": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                   C.c_int32, C.POINTER(C.c_double)]),
    "jtp_get_marginals": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "jtp_get_z": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]),
    "jtp_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "jtp_set_profiling_granularity": (C.c_int, [C.c_void_p, C.c_int32]),
    "jtp_set_profiling_stride": (C.c_int, [C.c_void_p, C.c_int32]),
    "jtp_region_begin": (C.c_int, [C.c_void_p]),
    "jtp_region_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "jtp_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "jtp_get_launch_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int32]),
    "jtp_debug_read_msg": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_double)]),
    "jtp_debug_set": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "jtp_kernel_name": (C.c_char_p, [C.c_int32]),
    "jtp_comm_unique_id": (C.c_int, [C.c_void_p]),
    "jtp_comm_init": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    "jtp_comm_destroy": (C.c_int, []),
    "jtp_comm_info": (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "jtp_comm_selftest": (C.c_int, [C.c_int32]),
    "jtp_device_count": (C.c_int, [C.POINTER(C.c_int32)]),
    "jtp_device_memory": (C.c_int, [C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "jtp_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "jtp_host_free": (C.c_int, [C.c_void_p]),
    "jtp_last_error": (C.c_char_p, []),
    "jtp_version": (C.c_char_p, []),
}


def lib():
    """Load libjtprop.so once.  Raises ImportError with build instructions if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libjtprop.so not found at %s: build it with `python junction-tree_amd/build.py` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    handle = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            if "JTPROP_LIB" in os.environ:      # (A/B against an older build of the library: it may lack newer entry points)
                continue
            raise
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return _lib


def check(rc):
    """Map a libjtprop status to the exception the reference's users would expect:
    structural problems -> ValueError (numpy raises ValueError for shape mismatches in the
    reference), everything else -> JtpError (a RuntimeError)."""
    if rc == JTP_OK:
        return
    msg = lib().jtp_last_error().decode("utf-8", "replace")
    if rc == JTP_EUNSUPPORTED:
        raise UnsupportedStructure(msg)
    if rc == JTP_EINVAL:
        raise ValueError(msg)
    if rc == JTP_ENOMEM:
        raise MemoryError(msg)
    raise JtpError(msg)


def device_count():
    n = C.c_int32(0)
    rc = lib().jtp_device_count(C.byref(n))
    return n.value if rc == JTP_OK else 0
