"""GPU: error behaviour of the C ABI — bad arguments must come back as status codes with a message
(the reference panics; include/ls_amd.h promises nothing aborts across the boundary)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def test_raw_calls_return_codes_not_crashes():
    import ls_amd
    from ls_amd import _lib
    lib = _lib.load()
    # no context
    assert lib.ls_ctx_synchronize(None) == _lib.LS_ERR_INVALID
    assert lib.ls_ctx_set_nq_algo(None, 0) == _lib.LS_ERR_INVALID
    assert lib.ls_ctx_destroy(None) == _lib.LS_OK                      # free(NULL) semantics
    h = C.c_void_p()
    assert lib.ls_ctx_create(10_000, C.byref(h)) == _lib.LS_ERR_NO_DEVICE
    assert b"out of range" in lib.ls_last_error(None)
    assert lib.ls_ctx_create(0, None) == _lib.LS_ERR_INVALID
    ctx = ls_amd.Context(0)
    hh = ctx._h
    rows = np.arange(8, dtype=np.int32)
    cols = np.array([1, 2], dtype=np.int32)
    res = _lib.NqResult()
    # null pointers / bad sizes
    assert lib.ls_nqueens_eval(hh, 8, None, 0, _vp(cols), 2, 10, C.byref(res), None) == _lib.LS_ERR_INVALID
    assert lib.ls_nqueens_eval(hh, 8, _vp(rows), 0, None, 2, 10, C.byref(res), None) == _lib.LS_ERR_INVALID
    assert lib.ls_nqueens_eval(hh, 8, _vp(rows), 0, _vp(cols), 2, 10, None, None) == _lib.LS_ERR_INVALID
    assert lib.ls_nqueens_eval(hh, 0, _vp(rows), 0, _vp(cols), 2, 10, C.byref(res), None) == _lib.LS_ERR_INVALID
    assert lib.ls_nqueens_eval(hh, 8, _vp(rows), 0, _vp(cols), -1, 10, C.byref(res), None) == _lib.LS_ERR_INVALID
    assert lib.ls_nqueens_eval(hh, 40_000, _vp(rows), 0, _vp(cols), 2, 10, C.byref(res), None) == _lib.LS_ERR_UNSUPPORTED
    assert len(lib.ls_last_error(hh)) > 0
    assert lib.ls_ctx_set_nq_algo(hh, 7) == _lib.LS_ERR_INVALID
    assert lib.ls_nqueens_col_scores(hh, 1, 8, None, None, None) == _lib.LS_ERR_INVALID
    assert lib.ls_nqueens_eval_batch_next(hh, 1, 8, _vp(rows), _vp(np.zeros(1, np.int64)), _vp(cols),
                                          _vp(np.array([2], np.int32)), 2, 10, C.byref(res), None) == _lib.LS_ERR_INVALID
    # scheduling: call order and argument checks
    sres = _lib.SchedResult()
    st = np.zeros(32, dtype=np.int32)
    assert lib.ls_sched_eval_resident(hh, _vp(st), 0, 10, C.byref(sres)) == _lib.LS_ERR_STATE
    assert lib.ls_sched_moves_upload(hh, None, 0) == _lib.LS_ERR_STATE
    assert lib.ls_sched_eval_exhaustive(hh, 1, _vp(st), 10, C.byref(sres)) == _lib.LS_ERR_STATE
    assert lib.ls_sched_exhaustive_moves(hh, 1, _vp(st), _vp(np.zeros(32 * 7 * 3, np.int32))) == _lib.LS_ERR_STATE
    assert lib.ls_sched_instance_upload(hh, 31, 32, 9, 7, _vp(np.zeros(7, np.uint64))) == _lib.LS_ERR_INVALID   # weekday 9
    assert lib.ls_sched_instance_upload(hh, 31, 30, 0, 7, _vp(np.zeros(7, np.uint64))) == _lib.LS_ERR_INVALID   # stored < scored
    assert lib.ls_sched_instance_upload(hh, 31, 32, 0, 7, None) == _lib.LS_ERR_INVALID
    assert lib.ls_sched_instance_upload(hh, 31, 32, 0, 7, _vp(np.zeros(7, np.uint64))) == _lib.LS_OK
    assert lib.ls_sched_eval(hh, _vp(st), None, 5, 10, C.byref(sres), None, None) == _lib.LS_ERR_INVALID
    assert lib.ls_sched_eval_exhaustive(hh, 1, None, 10, C.byref(sres)) == _lib.LS_ERR_INVALID
    assert lib.ls_sched_eval_exhaustive(hh, 1, _vp(st), -1, C.byref(sres)) == _lib.LS_ERR_INVALID
    assert lib.ls_sched_eval_exhaustive(hh, 0, _vp(st), 10, C.byref(sres)) == _lib.LS_ERR_INVALID
    assert lib.ls_sched_exhaustive_moves(hh, 1, _vp(st), None) == _lib.LS_ERR_INVALID
    assert lib.ls_sched_eval_exhaustive(hh, 1, _vp(np.full(32, 7, np.int32)), 10, C.byref(sres)) == _lib.LS_ERR_INVALID  # employee 7 of 0..6
    assert lib.ls_sched_eval(hh, _vp(st), None, 0, 10, C.byref(sres), None, None) == _lib.LS_OK       # empty move list is fine
    assert sres.n_scored == 0 and sres.best_index == -1
    assert lib.ls_sched_eval_resident(hh, _vp(st), 5, 10, C.byref(sres)) == _lib.LS_ERR_INVALID       # nothing resident yet
    # solver-level
    p = _lib.SolverParams(10, 10, 1, 10, 10, 10, 1)
    s = C.c_void_p()
    assert lib.ls_solver_create_nqueens(hh, 0, b"42", C.byref(p), C.byref(s)) == _lib.LS_ERR_UNSUPPORTED
    assert lib.ls_solver_create_nqueens(hh, 8, None, C.byref(p), C.byref(s)) == _lib.LS_ERR_INVALID
    assert lib.ls_solver_create_nqueens(hh, 8, b"42", None, C.byref(s)) == _lib.LS_ERR_INVALID
    assert lib.ls_solver_execute_round(None) == _lib.LS_ERR_INVALID
    assert lib.ls_solver_free(None) == _lib.LS_OK
    # the context is still healthy afterwards
    got, _ = ctx.nqueens_eval([1, 3, 0, 2], 0, [0], 10)
    assert got["n_scored"] == 3
    ctx.close()
