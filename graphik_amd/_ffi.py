"""ctypes binding of the C ABI declared in include/graphik_amd.h.

There is NO fallback: if the HIP library is missing or fails to load, importing the symbols
raises.  (The CPU restatement under oracle/ is test infrastructure and is never used here.)
"""
import ctypes as C
import os

# PyTorch-ROCm ships its own libamdhip64.so.7; it must be the HIP runtime this process uses (the
# device pointers we are handed come from it).  Importing torch first makes the dynamic loader
# resolve our library's libamdhip64.so.7 dependency to the copy torch already loaded.
import torch  # noqa: F401  (ordering matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GIK_LIB_PATH") or os.path.join(_HERE, "lib", "libgraphik_amd.so")

TERM_EQ, TERM_LOWER, TERM_UPPER = 1, 2, 3
ABI_VERSION = 6


class TemplateDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("N", C.c_int32), ("k", C.c_int32), ("n_terms", C.c_int32),
        ("term_i", C.POINTER(C.c_int32)), ("term_j", C.POINTER(C.c_int32)),
        ("term_kind", C.POINTER(C.c_int32)),
        ("mingradnorm", C.c_double), ("maxiter", C.c_int32), ("maxinner", C.c_int32),
        ("mininner", C.c_int32), ("theta", C.c_double), ("kappa", C.c_double),
        ("rho_prime", C.c_double), ("rho_regularization", C.c_double),
        ("planar_proj_exact", C.c_int32), ("force_block_path", C.c_int32),
        ("waves_per_cu", C.c_int32), ("slice_outer_its", C.c_int32), ("debug_flags", C.c_int32),
        ("solver", C.c_int32), ("cg_minstepsize", C.c_double), ("cg_orth_value", C.c_double),
        ("cg_beta_type", C.c_int32), ("clique_closed_form", C.c_int32), ("hessian_form", C.c_int32),
    ]


CLIQUE_AUTO, CLIQUE_OFF, CLIQUE_DENSE = 0, 1, 2
HESS_COLUMN, HESS_PER_EDGE, HESS_AUTO = 0, 1, 2


class TemplateInfo(C.Structure):
    _fields_ = [("is_block", C.c_int32), ("max_terms_per_node", C.c_int32), ("n_clique", C.c_int32),
                ("n_slot_terms", C.c_int32), ("slots_per_thread", C.c_int32), ("waves_per_cu", C.c_int32),
                ("n_cu", C.c_int32), ("lds_bytes", C.c_int32), ("clique_closed_form", C.c_int32),
                ("anchored", C.c_int32), ("has_pipeline", C.c_int32), ("prepare_is_block", C.c_int32),
                ("node_per_lane", C.c_int32), ("problems_per_wave", C.c_int32), ("goals_per_wave", C.c_int32),
                ("hessian_form", C.c_int32)]


SOLVER_TRUST_REGIONS, SOLVER_CONJUGATE_GRADIENT = 0, 1


class Stats(C.Structure):
    """gik_stats (48 bytes per problem); engine.py sizes and decodes the stats buffer from this."""
    _fields_ = [("f", C.c_double), ("gradnorm", C.c_double), ("iterations", C.c_int32),
                ("inner_total", C.c_int32), ("stop", C.c_int32), ("n_accept", C.c_int32),
                ("inner_executed", C.c_int32), ("flags", C.c_int32), ("stepsize", C.c_double)]


STATS_BYTES = C.sizeof(Stats)
# column of each int32 field when the [B, STATS_BYTES / 8] fp64 stats buffer is viewed as int32
STATS_I32 = {name: getattr(Stats, name).offset // 4 for name, ct in Stats._fields_ if ct is C.c_int32}
STATS_F64 = {name: getattr(Stats, name).offset // 8 for name, ct in Stats._fields_ if ct is C.c_double}


class Trace(C.Structure):
    _fields_ = [("cap", C.c_int32), ("d_Delta", C.c_void_p), ("d_numit", C.c_void_p),
                ("d_stop", C.c_void_p), ("d_f_before", C.c_void_p),
                ("d_gradnorm_after", C.c_void_p), ("d_accept", C.c_void_p)]


class PipelineDesc(C.Structure):
    _fields_ = [
        ("n_joints", C.c_int32), ("T0", C.POINTER(C.c_double)),
        ("p_index", C.POINTER(C.c_int32)), ("q_index", C.POINTER(C.c_int32)),
        ("x_index", C.c_int32), ("y_index", C.c_int32), ("axis_length", C.c_double),
        ("goal_node0", C.c_int32), ("goal_node1", C.c_int32), ("goal_len", C.c_double),
        ("base_lower", C.POINTER(C.c_double)), ("base_upper", C.POINTER(C.c_double)),
        ("n_anchor", C.c_int32), ("anchor_index", C.POINTER(C.c_int32)),
        ("anchor_pos", C.POINTER(C.c_double)),
        ("n_pairs", C.c_int32), ("pair_i", C.POINTER(C.c_int32)), ("pair_j", C.POINTER(C.c_int32)),
        ("term_src", C.POINTER(C.c_int32)), ("term_static", C.POINTER(C.c_double)),
        ("last_link_along_z", C.c_int32), ("jacobi_sweeps", C.c_int32),
        ("force_block_prepare", C.c_int32), ("n_ee", C.c_int32),
        ("ee_goal_nodes", C.POINTER(C.c_int32)), ("ee_path", C.POINTER(C.c_int32)),
        ("n_goal_pairs", C.c_int32), ("reserved1", C.c_int32),
        ("goal_pair_a", C.POINTER(C.c_int32)), ("goal_pair_b", C.POINTER(C.c_int32)),
        ("ee_goal_len", C.POINTER(C.c_double)),
    ]


class AnchoredDesc(C.Structure):
    _fields_ = [
        ("n_anchor", C.c_int32), ("n_goal_anchor", C.c_int32), ("anchor_pos", C.POINTER(C.c_double)),
        ("term_target", C.POINTER(C.c_double)), ("n_pin", C.c_int32),
        ("pin_node", C.POINTER(C.c_int32)), ("pin_anchor", C.POINTER(C.c_int32)),
        ("pin_kind", C.POINTER(C.c_int32)), ("pin_target", C.POINTER(C.c_double)),
        ("n_obs", C.c_int32), ("reserved0", C.c_int32), ("obs", C.POINTER(C.c_double)),
        ("obs_node_mask", C.POINTER(C.c_int32)), ("full_N", C.c_int32), ("reserved1", C.c_int32),
        ("free_full_index", C.POINTER(C.c_int32)), ("anchor_full_index", C.POINTER(C.c_int32)),
        ("axis_length", C.c_double),
    ]


class PrepareDiag(C.Structure):
    _fields_ = [("d_lb", C.c_void_p), ("d_ub", C.c_void_p), ("d_eig", C.c_void_p)]


# every symbol include/graphik_amd.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "gik_last_error": (C.c_char_p, []),
    "gik_abi_version": (C.c_int, []),
    "gik_device_count": (C.c_int, []),
    "gik_default_params": (None, [C.POINTER(TemplateDesc)]),
    "gik_default_cg_params": (None, [C.POINTER(TemplateDesc)]),
    "gik_template_create": (C.c_int, [C.POINTER(TemplateDesc), C.POINTER(C.c_void_p)]),
    "gik_template_create_anchored": (C.c_int, [C.POINTER(TemplateDesc), C.POINTER(AnchoredDesc),
                                               C.POINTER(C.c_void_p)]),
    "gik_anchored_ws_doubles": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_int]),
    "gik_anchored_ik_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gik_anchored_last_solve_ms": (C.c_double, [C.c_void_p]),
    "gik_template_destroy": (None, [C.c_void_p]),
    "gik_template_get_info": (C.c_int, [C.c_void_p, C.POINTER(TemplateInfo)]),
    "gik_cost": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gik_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gik_cost_and_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "gik_hess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                           C.c_void_p]),
    "gik_proj": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gik_solve_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_void_p, C.POINTER(Trace), C.c_void_p]),
    "gik_pipeline_attach": (C.c_int, [C.c_void_p, C.POINTER(PipelineDesc)]),
    "gik_prepare_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "gik_prepare_batch_debug": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.POINTER(PrepareDiag), C.c_void_p]),
    "gik_recover_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "gik_ik_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


class GikError(RuntimeError):
    pass


def lib():
    """Load libgraphik_amd.so (raises if it has not been built: python -m graphik_amd.build)."""
    global _lib
    if _lib is None:
        if "GIK_LIB_PATH" not in os.environ:
            # never run against a binary that is older than its sources: rebuild when the toolchain
            # is here (one process builds under a file lock, the others wait), refuse otherwise.  A
            # prebuilt library whose .digest stamp was lost in packaging cannot be checked: with no
            # toolchain it is used as it is, with a warning.
            from . import build as _build
            if os.path.exists(LIB_PATH) and _build._stale(LIB_PATH):
                if _build.have_toolchain():
                    try:
                        _build.build()
                    except (OSError, RuntimeError, __import__("subprocess").CalledProcessError) as e:
                        raise GikError(f"rebuilding {LIB_PATH} failed: {e}") from e
                elif not os.path.exists(LIB_PATH + ".digest"):
                    import warnings
                    warnings.warn(f"{LIB_PATH}.digest is missing and there is no hipcc: cannot check "
                                  "that the library matches the sources in this tree")
                else:
                    raise GikError(f"{LIB_PATH} was built from other sources than the ones in this tree "
                                   "and there is no hipcc to rebuild it")
        if not os.path.exists(LIB_PATH):
            raise GikError(
                f"HIP extension not built: {LIB_PATH} is missing. Run `python -m graphik_amd.build` "
                "(or __graft_entry__.build()). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.gik_abi_version() != ABI_VERSION:
            raise GikError("libgraphik_amd.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise GikError(lib().gik_last_error().decode("utf-8", "replace"))
