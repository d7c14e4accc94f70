"""ctypes binding of include/ls_amd.h (libls_amd.so, built in-tree by
constraint-solver_amd/build.py with hipcc for gfx950).

There is no CPU fallback: if the shared library is missing, or no HIP device is
usable, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
SO_PATH = os.environ.get("LS_AMD_LIB") or os.path.join(_PKG, "lib", "libls_amd.so")   # LS_AMD_LIB: a diagnostic build (tools/table_stamps.py)
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "ls_amd.h")

LS_OK = 0
LS_ERR_INVALID, LS_ERR_NO_DEVICE, LS_ERR_HIP, LS_ERR_UNSUPPORTED, LS_ERR_NOMEM, LS_ERR_STATE = 1, 2, 3, 4, 5, 6
LS_NQ_ALGO_SCAN, LS_NQ_ALGO_TABLE, LS_NQ_ALGO_AUTO = 0, 1, 2
LS_NQ_MAX_N = 32768
LS_SCHED_MAX_DAYS = 64
LS_N_COUNTERS = 7
COUNTER_NAMES = ("server_requests", "server_launches", "server_waits_on_drained_launch", "server_boards_uploaded",
                 "completion_word_fallbacks", "resident_board_reuploads", "fibres_taken_over")
INT64_MIN = -(2 ** 63)

_ERR_NAMES = {1: "LS_ERR_INVALID", 2: "LS_ERR_NO_DEVICE", 3: "LS_ERR_HIP", 4: "LS_ERR_UNSUPPORTED",
              5: "LS_ERR_NOMEM", 6: "LS_ERR_STATE"}


class LsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{_ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class NqResult(C.Structure):
    _fields_ = [("n_scored", C.c_int64), ("best_index", C.c_int64), ("best_score", C.c_int64),
                ("best_col", C.c_int32), ("best_value", C.c_int32)]


class SchedMove(C.Structure):
    _fields_ = [("type", C.c_int32), ("a", C.c_int32), ("b", C.c_int32)]


class SolverParams(C.Structure):
    _fields_ = [("local_search_max_iterations", C.c_uint64), ("window_size", C.c_uint64),
                ("best_solutions_capacity", C.c_uint64), ("all_solutions_capacity", C.c_uint64),
                ("all_solution_iteration_expiry", C.c_uint64), ("iterated_local_search_max_iterations", C.c_uint64),
                ("max_allow_no_improvement_for", C.c_uint64)]


class SchedResult(C.Structure):
    _fields_ = [("n_scored", C.c_int64), ("best_index", C.c_int64), ("best_hard", C.c_int64),
                ("best_soft", C.c_int64)]


_vp = C.c_void_p
_i32, _i64 = C.c_int32, C.c_int64

# name -> (restype, argtypes); one entry per function declared in include/ls_amd.h
SIGNATURES = {
    "ls_abi_version": (C.c_int, []),
    "ls_device_count": (C.c_int, []),
    "ls_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "ls_ctx_destroy": (C.c_int, [_vp]),
    "ls_last_error": (C.c_char_p, [_vp]),
    "ls_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "ls_ctx_synchronize": (C.c_int, [_vp]),
    "ls_ctx_set_nq_algo": (C.c_int, [_vp, C.c_int]),
    "ls_ctx_get_nq_algo": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "ls_ctx_get_counters": (C.c_int, [_vp, _vp]),
    "ls_ctx_device_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.c_char_p, C.c_int]),
    "ls_nqueens_eval": (C.c_int, [_vp, _i32, _vp, _i64, _vp, _i32, _i64, _vp, _vp]),
    "ls_nqueens_eval_tabu": (C.c_int, [_vp, _i32, _vp, _i64, _vp, _i32, _i64, _vp, _i32, _vp, _vp]),
    "ls_nqueens_eval_batch": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp]),
    "ls_nqueens_eval_batch_dev": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp]),
    "ls_nqueens_eval_batch_next": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp]),
    "ls_nqueens_eval_batch_next_dev": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp]),
    "ls_nqueens_col_scores": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "ls_nqueens_col_scores_dev": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "ls_sched_instance_upload": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp]),
    "ls_sched_score": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "ls_sched_eval": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "ls_sched_moves_upload": (C.c_int, [_vp, _vp, _i64]),
    "ls_sched_exhaustive_moves": (C.c_int, [_vp, _i32, _vp, _vp]),
    "ls_sched_exhaustive_moves_dev": (C.c_int, [_vp, _i32, _vp, _vp]),
    "ls_sched_eval_exhaustive": (C.c_int, [_vp, _i32, _vp, _i64, _vp]),
    "ls_sched_eval_resident": (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "ls_sched_eval_batch": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "ls_sched_eval_batch_dev": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "ls_solver_create_nqueens": (C.c_int, [_vp, C.c_uint64, C.c_char_p, _vp, C.POINTER(_vp)]),
    "ls_solver_create_sched": (C.c_int, [_vp, _i32, _i32, _i32, _vp, C.c_char_p, _vp, C.POINTER(_vp)]),
    "ls_solver_execute_round": (C.c_int, [_vp]),
    "ls_solver_is_finished": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "ls_solver_get_iteration_info": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ls_solver_get_best_solution": (C.c_int, [_vp, _vp, _vp, _i32, C.POINTER(_i32), C.POINTER(C.c_int)]),
    "ls_solver_get_current_solution": (C.c_int, [_vp, _vp, _vp, _i32, C.POINTER(_i32)]),
    "ls_solver_get_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ls_solver_free": (C.c_int, [_vp]),
    "ls_multi_solver_create_nqueens": (C.c_int, [_vp, C.c_uint64, _i32, _vp, _vp, C.POINTER(_vp)]),
    "ls_multi_solver_create_sched": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, C.POINTER(_vp)]),
    "ls_multi_solver_configure": (C.c_int, [_vp, _i32, _i32]),
    "ls_multi_solver_run": (C.c_int, [_vp, C.c_uint64, C.c_double]),
    "ls_multi_solver_get_best": (C.c_int, [_vp, _i32, _vp, _vp, _i32, C.POINTER(_i32), C.POINTER(C.c_int)]),
    "ls_multi_solver_get_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64)]),
    "ls_multi_solver_get_restart_stats": (C.c_int, [_vp, _i32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ls_multi_solver_get_counters": (C.c_int, [_vp, _vp]),
    "ls_multi_solver_free": (C.c_int, [_vp]),
    "ls_host_hash_seed": (C.c_int, [C.c_char_p, _vp]),
    "ls_host_chacha20_block": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "ls_host_rng_script": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, C.POINTER(_i64)]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, the
    same as /opt/rocm's).  Two HIP runtimes in one process cannot both own the GPU, so
    when torch is installed its copy is loaded first and libls_amd.so binds to it by
    SONAME; a later `import torch` then finds its runtime already mapped.  torch itself
    is NOT imported here.  LS_AMD_SYSTEM_HIP=1 keeps the system runtime instead."""
    if os.environ.get("LS_AMD_SYSTEM_HIP") == "1":
        return None
    import importlib.util
    import sys

    origin = None
    if "torch" in sys.modules:
        origin = getattr(sys.modules["torch"], "__file__", None)
    else:
        try:
            spec = importlib.util.find_spec("torch")
            origin = spec.origin if spec else None
        except (ImportError, ValueError):
            origin = None
    if not origin:
        return None
    cand = os.path.join(os.path.dirname(origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
        return cand
    return None


HIP_RUNTIME = None


def load() -> C.CDLL:
    """Load libls_amd.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: build it with `python constraint-solver_amd/build.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        global HIP_RUNTIME
        HIP_RUNTIME = _share_hip_runtime_with_torch() or "system"
        lib = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
