"""ctypes loader for the CPU oracle (oracle/*.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (constraint-solver_amd/) never
imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
SKIPPED = -(2 ** 63)

_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/*.c with gcc (a few seconds)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


_i64 = C.c_int64
_p64 = C.POINTER(C.c_int64)


class SchedInstance(C.Structure):
    _fields_ = [("n_scored_days", _i64), ("start_weekday", _i64),
                ("hol_emp", _p64), ("hol_day", _p64), ("n_holidays", _i64)]


class SchedMove(C.Structure):
    _fields_ = [("type", _i64), ("a", _i64), ("b", _i64)]


def _declare(l):
    l.orc_nq_get_col_scores.argtypes = [_p64, _i64, _p64]
    l.orc_nq_get_col_scores.restype = None
    l.orc_nq_col_scores_table.argtypes = [_p64, _i64, _p64]
    l.orc_nq_col_scores_table.restype = None
    l.orc_nq_score.argtypes = [_p64, _i64]
    l.orc_nq_score.restype = _i64
    for name in ("orc_nq_neighbourhood_literal", "orc_nq_neighbourhood_delta",
                 "orc_nq_neighbourhood_table"):
        f = getattr(l, name)
        f.argtypes = [_p64, _i64, _p64, _i64, _i64, _p64, _p64, _p64, _p64]
        f.restype = C.c_int
    l.orc_nq_neighbourhood_literal_tabu.argtypes = [_p64, _i64, _p64, _i64, _i64, _p64, _i64, _p64, _p64, _p64, _p64]
    l.orc_nq_neighbourhood_literal_tabu.restype = C.c_int
    l.orc_sched_neighbourhood_tabu.argtypes = [
        C.POINTER(SchedInstance), _p64, _i64, C.POINTER(SchedMove), _i64, _i64, _p64, _i64,
        _p64, _p64, _p64, _p64, _p64, _p64]
    l.orc_sched_neighbourhood_tabu.restype = C.c_int
    l.orc_nq_baseline_run.argtypes = [_p64, _i64, _p64, _i64, _i64, _p64]
    l.orc_nq_baseline_run.restype = _i64
    l.orc_weekday_from_monday.argtypes = [_i64, _i64, _i64]
    l.orc_weekday_from_monday.restype = _i64
    l.orc_days_between.argtypes = [_i64] * 6
    l.orc_days_between.restype = _i64
    l.orc_sched_score.argtypes = [C.POINTER(SchedInstance), _p64, _p64, _p64, _p64]
    l.orc_sched_score.restype = C.c_int
    l.orc_sched_neighbourhood.argtypes = [
        C.POINTER(SchedInstance), _p64, _i64, C.POINTER(SchedMove), _i64, _i64,
        _p64, _p64, _p64, _p64, _p64, _p64]
    l.orc_sched_neighbourhood.restype = C.c_int
    l.orc_sched_exhaustive_moves.argtypes = [_p64, _i64, _p64, _i64, C.POINTER(SchedMove)]
    l.orc_sched_exhaustive_moves.restype = _i64


def _a64(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.int64))


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_p64)


# ----------------------------------------------------------------- N-queens
def nq_col_scores(rows) -> np.ndarray:
    r = _a64(rows)
    out = np.zeros(len(r), dtype=np.int64)
    lib().orc_nq_get_col_scores(_ptr(r), len(r), _ptr(out))
    return out


def nq_col_scores_table(rows) -> np.ndarray:
    r = _a64(rows)
    out = np.zeros(len(r), dtype=np.int64)
    lib().orc_nq_col_scores_table(_ptr(r), len(r), _ptr(out))
    return out


def nq_score(rows) -> int:
    r = _a64(rows)
    return int(lib().orc_nq_score(_ptr(r), len(r)))


def nq_neighbourhood(rows, cols, window, formulation="literal", want_scores=True):
    """Returns dict(n_scored, best_index, best_score, scores[k] or None)."""
    r = _a64(rows)
    c = _a64(cols)
    n = len(r)
    scores = np.empty(len(c) * n, dtype=np.int64) if want_scores else None
    ns, bi, bs = _i64(0), _i64(0), _i64(0)
    f = getattr(lib(), "orc_nq_neighbourhood_" + formulation)
    rc = f(_ptr(r), n, _ptr(c), len(c), int(window),
           _ptr(scores) if want_scores else None,
           C.byref(ns), C.byref(bi), C.byref(bs))
    if rc != 0:
        raise RuntimeError(f"oracle nq neighbourhood rc={rc}")
    return dict(n_scored=ns.value, best_index=bi.value, best_score=bs.value, scores=scores)


def nq_neighbourhood_tabu(rows, cols, window, tabu_boards, want_scores=True):
    """lines :315-325 with a GENERAL tabu set (an iterable of boards; the current board is not added implicitly)."""
    r = _a64(rows)
    c = _a64(cols)
    n = len(r)
    t = _a64([list(b) for b in tabu_boards]).reshape(-1, n) if len(tabu_boards) else np.zeros((0, n), dtype=np.int64)
    scores = np.empty(len(c) * n, dtype=np.int64) if want_scores else None
    ns, bi, bs = _i64(0), _i64(0), _i64(0)
    rc = lib().orc_nq_neighbourhood_literal_tabu(_ptr(r), n, _ptr(c), len(c), int(window), _ptr(t) if len(t) else None, len(t),
                                                 _ptr(scores) if want_scores else None, C.byref(ns), C.byref(bi), C.byref(bs))
    if rc != 0:
        raise RuntimeError(f"oracle nq neighbourhood (tabu) rc={rc}")
    return dict(n_scored=ns.value, best_index=bi.value, best_score=bs.value, scores=scores)


def nq_baseline_run(rows, cols, max_candidates) -> tuple[int, int]:
    r = _a64(rows)
    c = _a64(cols)
    bs = _i64(0)
    ns = lib().orc_nq_baseline_run(_ptr(r), len(r), _ptr(c), len(c), int(max_candidates), C.byref(bs))
    return int(ns), int(bs.value)


# --------------------------------------------------------------- scheduling
class Sched:
    """Instance wrapper: start/end dates as (y, m, d) like the reference's
    NaiveDate pair; holidays as {employee id: iterable of day indices}."""

    def __init__(self, start, end, holidays=None):
        l = lib()
        self.start, self.end = tuple(start), tuple(end)
        self.n_scored_days = int(l.orc_days_between(*self.start, *self.end)) + 1
        self.start_weekday = int(l.orc_weekday_from_monday(*self.start))
        pairs = sorted({(int(e), int(d)) for e, ds in (holidays or {}).items() for d in ds})
        self._he = _a64([p[0] for p in pairs])
        self._hd = _a64([p[1] for p in pairs])
        self.inst = SchedInstance(self.n_scored_days, self.start_weekday,
                                  _ptr(self._he), _ptr(self._hd), len(pairs))

    def score(self, e, terms=False):
        a = _a64(e)
        assert len(a) >= self.n_scored_days
        h, s = _i64(0), _i64(0)
        t = np.zeros(8, dtype=np.int64)
        rc = lib().orc_sched_score(C.byref(self.inst), _ptr(a), C.byref(h), C.byref(s), _ptr(t))
        if rc != 0:
            raise RuntimeError(f"oracle sched score rc={rc}")
        return (h.value, s.value, t) if terms else (h.value, s.value)

    def neighbourhood(self, e, moves, window, want_scores=True):
        """moves: array-like [n_moves, 3] of (type, a, b) (an int64 array is passed to C as it is)."""
        a = _a64(e)
        m = _a64(moves).reshape(-1, 3)
        mv = C.cast(m.ctypes.data, C.POINTER(SchedMove))  # SchedMove = three int64: the array IS the move list
        hard = np.empty(len(m), dtype=np.int64) if want_scores else None
        soft = np.empty(len(m), dtype=np.int64) if want_scores else None
        ns, bi, bh, bs = _i64(0), _i64(0), _i64(0), _i64(0)
        rc = lib().orc_sched_neighbourhood(
            C.byref(self.inst), _ptr(a), len(a), mv, len(m), int(window),
            _ptr(hard) if want_scores else None, _ptr(soft) if want_scores else None,
            C.byref(ns), C.byref(bi), C.byref(bh), C.byref(bs))
        if rc != 0:
            raise RuntimeError(f"oracle sched neighbourhood rc={rc}")
        return dict(n_scored=ns.value, best_index=bi.value, best_hard=bh.value,
                    best_soft=bs.value, hard=hard, soft=soft)

    def neighbourhood_tabu(self, e, moves, window, tabu_states):
        """Like neighbourhood() with a GENERAL tabu set (iterable of stored vectors; `e` itself is not added implicitly)."""
        a = _a64(e)
        m = _a64(moves).reshape(-1, 3)
        mv = (SchedMove * len(m))()
        for i, (t, x, y) in enumerate(m):
            mv[i].type, mv[i].a, mv[i].b = int(t), int(x), int(y)
        tb = _a64([list(s) for s in tabu_states]).reshape(-1, len(a)) if len(tabu_states) else np.zeros((0, len(a)), dtype=np.int64)
        hard = np.empty(len(m), dtype=np.int64)
        soft = np.empty(len(m), dtype=np.int64)
        ns, bi, bh, bs = _i64(0), _i64(0), _i64(0), _i64(0)
        rc = lib().orc_sched_neighbourhood_tabu(
            C.byref(self.inst), _ptr(a), len(a), mv, len(m), int(window), _ptr(tb) if len(tb) else None, len(tb),
            _ptr(hard), _ptr(soft), C.byref(ns), C.byref(bi), C.byref(bh), C.byref(bs))
        if rc != 0:
            raise RuntimeError(f"oracle sched neighbourhood (tabu) rc={rc}")
        return dict(n_scored=ns.value, best_index=bi.value, best_hard=bh.value, best_soft=bs.value, hard=hard, soft=soft)

    @staticmethod
    def exhaustive_moves(e, employees) -> np.ndarray:
        a = _a64(e)
        emp = _a64(sorted(employees))
        mv = (SchedMove * (len(a) * len(emp)))()
        cnt = lib().orc_sched_exhaustive_moves(_ptr(a), len(a), _ptr(emp), len(emp), mv)
        if cnt < 0:
            raise RuntimeError(f"oracle exhaustive moves rc={cnt}")
        return np.array([(mv[i].type, mv[i].a, mv[i].b) for i in range(cnt)], dtype=np.int64).reshape(-1, 3)
