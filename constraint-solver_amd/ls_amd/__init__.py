"""ls_amd — Python view of the MI355X local-search neighbourhood evaluator.

Thin wrappers over the C ABI (include/ls_amd.h); all compute happens in the HIP
kernels of constraint-solver_amd/csrc.  numpy arrays in / numpy arrays out for
the host-pointer entry points, raw device addresses (e.g. torch
`tensor.data_ptr()`) for the *_dev ones.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import (INT64_MIN, LS_NQ_ALGO_AUTO, LS_NQ_ALGO_SCAN, LS_NQ_ALGO_TABLE, LS_NQ_MAX_N, LS_SCHED_MAX_DAYS,  # noqa: F401
                   LsError, NqResult, SchedMove, SchedResult, SolverParams)

ALGOS = {"scan": LS_NQ_ALGO_SCAN, "table": LS_NQ_ALGO_TABLE, "auto": LS_NQ_ALGO_AUTO}


def device_count() -> int:
    return int(_lib.load().ls_device_count())


def _c(a, dtype):
    return np.ascontiguousarray(np.asarray(a), dtype=dtype)


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU's context (ls_ctx).  Not thread-safe; one per device."""

    def __init__(self, device: int = 0, algo: str = "auto"):
        self._l = _lib.load()
        h = C.c_void_p()
        rc = self._l.ls_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise LsError(rc, self._l.ls_last_error(None).decode())
        self._h = h
        self.device = int(device)
        self.set_algo(algo)

    # -- plumbing
    def _check(self, rc: int):
        if rc != 0:
            raise LsError(rc, self._l.ls_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._l.ls_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, hip_stream: int | None):
        self._check(self._l.ls_ctx_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def synchronize(self):
        self._check(self._l.ls_ctx_synchronize(self._h))

    def set_algo(self, algo: str | int):
        a = ALGOS[algo] if isinstance(algo, str) else int(algo)
        self._check(self._l.ls_ctx_set_nq_algo(self._h, a))

    def counters(self) -> dict:
        """ls_ctx_get_counters: the slow branches of the latency-critical path (see include/ls_amd.h)."""
        out = np.zeros(_lib.LS_N_COUNTERS, dtype=np.uint64)
        self._check(self._l.ls_ctx_get_counters(self._h, _vp(out)))
        return {k: int(v) for k, v in zip(_lib.COUNTER_NAMES, out)}

    def device_info(self) -> dict:
        cus, ws, khz = C.c_int(), C.c_int(), C.c_int()
        name = C.create_string_buffer(256)
        self._check(self._l.ls_ctx_device_info(self._h, C.byref(cus), C.byref(ws), C.byref(khz), name, 256))
        return dict(n_cus=cus.value, wave_size=ws.value, clock_khz=khz.value, name=name.value.decode())

    # -- N-queens
    def nqueens_eval_batch(self, rows, cur_scores, cols, n_cols, window, want_scores=False):
        """rows [R, n] int32; cur_scores [R]; cols [R, max_cols]; n_cols [R].
        Returns (list of dict per restart, scores [R, max_cols*n] or None)."""
        rows = _c(rows, np.int32)
        assert rows.ndim == 2
        R, n = rows.shape
        cur = _c(cur_scores, np.int64).reshape(R)
        cols = _c(cols, np.int32).reshape(R, -1)
        max_cols = cols.shape[1]
        ncols = _c(n_cols, np.int32).reshape(R)
        res = (NqResult * R)()
        scores = np.empty((R, max_cols * n), dtype=np.int64) if want_scores else None
        self._check(self._l.ls_nqueens_eval_batch(self._h, R, n, _vp(rows), _vp(cur), _vp(cols), _vp(ncols),
                                                  max_cols, int(window), C.cast(res, C.c_void_p), _vp(scores)))
        out = [dict(n_scored=r.n_scored, best_index=r.best_index, best_score=r.best_score,
                    best_col=r.best_col, best_value=r.best_value) for r in res]
        return out, scores

    def nqueens_eval_batch_next(self, rows, cur_scores, cols, n_cols, window):
        """Like nqueens_eval_batch, plus get_col_scores of every restart's winner board [R, n]."""
        rows = _c(rows, np.int32)
        R, n = rows.shape
        cur = _c(cur_scores, np.int64).reshape(R)
        cols = _c(cols, np.int32).reshape(R, -1)
        max_cols = cols.shape[1]
        ncols = _c(n_cols, np.int32).reshape(R)
        res = (NqResult * R)()
        nxt = np.empty((R, n), dtype=np.int64)
        self._check(self._l.ls_nqueens_eval_batch_next(self._h, R, n, _vp(rows), _vp(cur), _vp(cols), _vp(ncols), max_cols,
                                                       int(window), C.cast(res, C.c_void_p), _vp(nxt)))
        out = [dict(n_scored=r.n_scored, best_index=r.best_index, best_score=r.best_score,
                    best_col=r.best_col, best_value=r.best_value) for r in res]
        return out, nxt

    def nqueens_eval(self, rows, cur_score, cols, window, want_scores=False):
        rows = _c(rows, np.int32).reshape(1, -1)
        cols = _c(cols, np.int32).reshape(1, -1)
        out, scores = self.nqueens_eval_batch(rows, [cur_score], cols, [cols.shape[1]], window, want_scores)
        return out[0], (scores[0] if scores is not None else None)

    def nqueens_eval_tabu(self, rows, cur_score, cols, window, tabu_positions, want_scores=False):
        """ls_nqueens_eval_tabu: tabu_positions = sorted unfiltered positions of candidates equal to a tabu board other
        than `rows` (general History)."""
        rows = _c(rows, np.int32).reshape(-1)
        cols = _c(cols, np.int32).reshape(-1)
        tp = _c(tabu_positions, np.int64).reshape(-1)
        n = len(rows)
        res = NqResult()
        scores = np.empty(len(cols) * n, dtype=np.int64) if want_scores else None
        self._check(self._l.ls_nqueens_eval_tabu(self._h, n, _vp(rows), int(cur_score), _vp(cols), len(cols), int(window),
                                                 _vp(tp), len(tp), C.byref(res), _vp(scores)))
        return dict(n_scored=res.n_scored, best_index=res.best_index, best_score=res.best_score, best_col=res.best_col,
                    best_value=res.best_value), scores

    def nqueens_eval_batch_dev(self, n_restarts, n, rows_d, cur_scores_d, cols_d, n_cols_d, max_cols, window,
                               results_d, scores_out_d=None):
        """All arguments are device addresses (ints); asynchronous."""
        self._check(self._l.ls_nqueens_eval_batch_dev(
            self._h, int(n_restarts), int(n), C.c_void_p(rows_d), C.c_void_p(cur_scores_d), C.c_void_p(cols_d),
            C.c_void_p(n_cols_d), int(max_cols), int(window), C.c_void_p(results_d),
            C.c_void_p(scores_out_d or 0)))

    def nqueens_eval_batch_next_dev(self, n_restarts, n, rows_d, cur_scores_d, cols_d, n_cols_d, max_cols, window,
                                    results_d, next_rows_d, next_col_scores_d, next_scores_d=None):
        """Device-pointer LS step: result + winner board + its column scores (+ its score); asynchronous."""
        self._check(self._l.ls_nqueens_eval_batch_next_dev(
            self._h, int(n_restarts), int(n), C.c_void_p(rows_d), C.c_void_p(cur_scores_d), C.c_void_p(cols_d),
            C.c_void_p(n_cols_d), int(max_cols), int(window), C.c_void_p(results_d), C.c_void_p(next_rows_d),
            C.c_void_p(next_col_scores_d), C.c_void_p(next_scores_d or 0)))

    def nqueens_col_scores(self, rows):
        """rows [R, n] -> (col_scores [R, n] int64, scores [R] int64)."""
        rows = _c(rows, np.int32)
        if rows.ndim == 1:
            rows = rows.reshape(1, -1)
        R, n = rows.shape
        cs = np.empty((R, n), dtype=np.int64)
        tot = np.empty(R, dtype=np.int64)
        self._check(self._l.ls_nqueens_col_scores(self._h, R, n, _vp(rows), _vp(cs), _vp(tot)))
        return cs, tot

    def nqueens_col_scores_dev(self, n_restarts, n, rows_d, col_scores_d, scores_d):
        self._check(self._l.ls_nqueens_col_scores_dev(self._h, int(n_restarts), int(n), C.c_void_p(rows_d),
                                                      C.c_void_p(col_scores_d or 0), C.c_void_p(scores_d or 0)))

    # -- scheduling
    def sched_instance_upload(self, n_days_scored, n_days_stored, start_weekday, n_employees, holiday_bits=None):
        hb = np.zeros(n_employees, dtype=np.uint64) if holiday_bits is None else _c(holiday_bits, np.uint64)
        assert hb.shape == (n_employees,)
        self._check(self._l.ls_sched_instance_upload(self._h, int(n_days_scored), int(n_days_stored),
                                                     int(start_weekday), int(n_employees), _vp(hb)))
        self._sched_dims = (int(n_days_scored), int(n_days_stored), int(n_employees))

    def sched_score(self, states, want_terms=False):
        states = _c(states, np.int32)
        if states.ndim == 1:
            states = states.reshape(1, -1)
        ns = states.shape[0]
        hard = np.empty(ns, dtype=np.int64)
        soft = np.empty(ns, dtype=np.int64)
        terms = np.empty((ns, 8), dtype=np.int64) if want_terms else None
        self._check(self._l.ls_sched_score(self._h, ns, _vp(states), _vp(hard), _vp(soft), _vp(terms)))
        return (hard, soft, terms) if want_terms else (hard, soft)

    @staticmethod
    def pack_moves(moves) -> np.ndarray:
        """[..., 3] (type, a, b) -> contiguous int32 array laid out as ls_sched_move."""
        return _c(moves, np.int32).reshape(-1, 3)

    def sched_eval_batch(self, states, moves, n_moves, window, want_scores=False):
        """states [R, S]; moves [R, max_moves, 3]; n_moves [R]."""
        states = _c(states, np.int32)
        R = states.shape[0]
        mv = _c(moves, np.int32).reshape(R, -1, 3)
        max_moves = mv.shape[1]
        nm = _c(n_moves, np.int64).reshape(R)
        res = (SchedResult * R)()
        hard = np.empty((R, max_moves), dtype=np.int64) if want_scores else None
        soft = np.empty((R, max_moves), dtype=np.int64) if want_scores else None
        self._check(self._l.ls_sched_eval_batch(self._h, R, _vp(states), _vp(mv), _vp(nm), max_moves, int(window),
                                                C.cast(res, C.c_void_p), _vp(hard), _vp(soft)))
        out = [dict(n_scored=r.n_scored, best_index=r.best_index, best_hard=r.best_hard, best_soft=r.best_soft)
               for r in res]
        return out, hard, soft

    def sched_eval(self, state, moves, window, want_scores=False):
        mv = _c(moves, np.int32).reshape(1, -1, 3)
        out, hard, soft = self.sched_eval_batch(_c(state, np.int32).reshape(1, -1), mv, [mv.shape[1]], window,
                                                want_scores)
        return out[0], (hard[0] if hard is not None else None), (soft[0] if soft is not None else None)

    def sched_moves_upload(self, moves):
        mv = _c(moves, np.int32).reshape(-1, 3)
        self._check(self._l.ls_sched_moves_upload(self._h, _vp(mv), len(mv)))

    def sched_exhaustive_moves(self, states):
        """ScheduleMoveProposer order (lib.rs:493-559) for states [R, S] -> moves [R, S * E, 3] (type, a, b)."""
        st = _c(states, np.int32)
        if st.ndim == 1:
            st = st.reshape(1, -1)
        R, S = st.shape
        E = self._sched_dims[2]
        mv = np.empty((R, S * E, 3), dtype=np.int32)
        self._check(self._l.ls_sched_exhaustive_moves(self._h, R, _vp(st), _vp(mv)))
        return mv

    def sched_eval_exhaustive(self, states, window):
        """Exhaustive neighbourhood of every state, generated and scored on the device."""
        st = _c(states, np.int32)
        if st.ndim == 1:
            st = st.reshape(1, -1)
        R = st.shape[0]
        res = (SchedResult * R)()
        self._check(self._l.ls_sched_eval_exhaustive(self._h, R, _vp(st), int(window), C.cast(res, C.c_void_p)))
        return [dict(n_scored=r.n_scored, best_index=r.best_index, best_hard=r.best_hard, best_soft=r.best_soft) for r in res]

    def sched_eval_resident(self, state, n_moves, window):
        st = _c(state, np.int32)
        res = SchedResult()
        self._check(self._l.ls_sched_eval_resident(self._h, _vp(st), int(n_moves), int(window), C.byref(res)))
        return dict(n_scored=res.n_scored, best_index=res.best_index, best_hard=res.best_hard, best_soft=res.best_soft)

    def sched_eval_batch_dev(self, n_restarts, states_d, moves_d, n_moves_d, max_moves, window, results_d,
                             hard_out_d=None, soft_out_d=None):
        self._check(self._l.ls_sched_eval_batch_dev(
            self._h, int(n_restarts), C.c_void_p(states_d), C.c_void_p(moves_d), C.c_void_p(n_moves_d),
            int(max_moves), int(window), C.c_void_p(results_d), C.c_void_p(hard_out_d or 0),
            C.c_void_p(soft_out_d or 0)))


# ------------------------------------------------------------------ whole solver
_PARAM_FIELDS = [f[0] for f in SolverParams._fields_]


def _params(d: dict) -> SolverParams:
    return SolverParams(*[int(d[k]) for k in _PARAM_FIELDS])


class Solver:
    """ls_solver: the reference's IteratedLocalSearch driver on the host, every score and
    neighbourhood on the GPU of `ctx` (mirrors the wasm SolverContext API)."""

    def __init__(self, ctx: Context, handle, kind: str, sol_len: int):
        self._ctx, self._h, self.kind, self._len = ctx, handle, kind, sol_len
        self._l = ctx._l

    @classmethod
    def nqueens(cls, ctx: Context, board_size: int, seed: str, params: dict) -> "Solver":
        h = C.c_void_p()
        p = _params(params)
        ctx._check(ctx._l.ls_solver_create_nqueens(ctx._h, int(board_size), seed.encode(), C.byref(p), C.byref(h)))
        return cls(ctx, h, "nqueens", int(board_size))

    @classmethod
    def scheduling(cls, ctx: Context, n_days_scored: int, start_weekday: int, n_employees: int, seed: str,
                   params: dict, holiday_bits=None) -> "Solver":
        hb = np.zeros(n_employees, dtype=np.uint64) if holiday_bits is None else _c(holiday_bits, np.uint64)
        h = C.c_void_p()
        p = _params(params)
        ctx._check(ctx._l.ls_solver_create_sched(ctx._h, int(n_days_scored), int(start_weekday), int(n_employees),
                                                 _vp(hb), seed.encode(), C.byref(p), C.byref(h)))
        return cls(ctx, h, "sched", int(n_days_scored) + 1)

    def close(self):
        if getattr(self, "_h", None):
            self._l.ls_solver_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def execute_round(self):
        self._ctx._check(self._l.ls_solver_execute_round(self._h))

    def is_finished(self) -> bool:
        f = C.c_int()
        self._ctx._check(self._l.ls_solver_is_finished(self._h, C.byref(f)))
        return bool(f.value)

    def iteration_info(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._ctx._check(self._l.ls_solver_get_iteration_info(self._h, C.byref(a), C.byref(b)))
        return dict(current=a.value, total=b.value)

    def _score(self, sc):
        return int(sc[0]) if self.kind == "nqueens" else (int(sc[0]), int(sc[1]))

    def best(self):
        sc = np.zeros(2, dtype=np.int64)
        sol = np.zeros(self._len, dtype=np.int64)
        n, have = C.c_int32(), C.c_int()
        self._ctx._check(self._l.ls_solver_get_best_solution(self._h, _vp(sc), _vp(sol), self._len, C.byref(n),
                                                             C.byref(have)))
        if not have.value:
            return None
        return self._score(sc), tuple(int(x) for x in sol[:n.value])

    def current(self):
        sc = np.zeros(2, dtype=np.int64)
        sol = np.zeros(self._len, dtype=np.int64)
        n = C.c_int32()
        self._ctx._check(self._l.ls_solver_get_current_solution(self._h, _vp(sc), _vp(sol), self._len, C.byref(n)))
        return self._score(sc), tuple(int(x) for x in sol[:n.value])

    def stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._ctx._check(self._l.ls_solver_get_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(ls_iterations=a.value, neighbours_scored=b.value, gpu_evals=c.value)


class MultiSolver:
    """ls_multi_solver: many independent restarts on one GPU, each restart's result equal to running it alone.
    MultiSolver(ctx, board_size, seeds, params) = N-queens; MultiSolver.scheduling(...) = the scheduling plug-in."""

    MODES = {"lockstep": 0, "free": 1}

    def __init__(self, ctx: Context, board_size: int, seeds, params: dict, n_slots: int | None = None, mode: str | None = None,
                 _sched=None):
        self._ctx, self._l, self.n, self.seeds = ctx, ctx._l, int(board_size), [str(s) for s in seeds]
        self.kind = "nqueens" if _sched is None else "sched"
        arr = (C.c_char_p * len(self.seeds))(*[s.encode() for s in self.seeds])
        h = C.c_void_p()
        p = _params(params)
        if _sched is None:
            ctx._check(self._l.ls_multi_solver_create_nqueens(ctx._h, self.n, len(self.seeds), C.cast(arr, C.c_void_p),
                                                              C.byref(p), C.byref(h)))
        else:
            n_days_scored, start_weekday, n_employees, hb = _sched
            ctx._check(self._l.ls_multi_solver_create_sched(ctx._h, int(n_days_scored), int(start_weekday), int(n_employees), _vp(hb),
                                                            len(self.seeds), C.cast(arr, C.c_void_p), C.byref(p), C.byref(h)))
        self._h = h
        if _sched is not None or n_slots is not None or mode is not None:  # (scheduling: configure also makes the slots' shared launch object)
            ctx._check(self._l.ls_multi_solver_configure(h, int(n_slots or 0), self.MODES[mode or "free"]))

    @classmethod
    def scheduling(cls, ctx: Context, n_days_scored: int, start_weekday: int, n_employees: int, seeds, params: dict,
                   holiday_bits=None, n_slots: int | None = None) -> "MultiSolver":
        """ls_multi_solver_create_sched: the restarts are the slots of ONE resident launch (step server or sub-grids)."""
        hb = np.zeros(n_employees, dtype=np.uint64) if holiday_bits is None else _c(holiday_bits, np.uint64)
        return cls(ctx, int(n_days_scored) + 1, seeds, params, n_slots=n_slots, _sched=(n_days_scored, start_weekday, n_employees, hb))

    def run(self, max_rounds_per_restart=2 ** 62, max_seconds=0.0):
        self._ctx._check(self._l.ls_multi_solver_run(self._h, int(max_rounds_per_restart), float(max_seconds)))

    def best(self, restart: int):
        sc = np.zeros(2, dtype=np.int64)
        sol = np.zeros(self.n, dtype=np.int64)
        n, have = C.c_int32(), C.c_int()
        self._ctx._check(self._l.ls_multi_solver_get_best(self._h, int(restart), _vp(sc), _vp(sol), self.n, C.byref(n),
                                                          C.byref(have)))
        if not have.value:
            return None
        score = int(sc[0]) if self.kind == "nqueens" else (int(sc[0]), int(sc[1]))
        return score, tuple(int(x) for x in sol[:n.value])

    def stats(self):
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._ctx._check(self._l.ls_multi_solver_get_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(ls_iterations=a.value, neighbours_scored=b.value, gpu_launches=c.value, rounds=d.value)

    def restart_stats(self, restart: int) -> dict:
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._ctx._check(self._l.ls_multi_solver_get_restart_stats(self._h, int(restart), C.byref(a), C.byref(b), C.byref(c)))
        return dict(ls_iterations=a.value, neighbours_scored=b.value, rounds=c.value)

    def counters(self) -> dict:
        out = np.zeros(_lib.LS_N_COUNTERS, dtype=np.uint64)
        self._ctx._check(self._l.ls_multi_solver_get_counters(self._h, _vp(out)))
        return {k: int(v) for k, v in zip(_lib.COUNTER_NAMES, out)}

    def close(self):
        if getattr(self, "_h", None):
            self._l.ls_multi_solver_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# host RNG primitives (no GPU needed)
def host_hash_seed(seed: str) -> bytes:
    out = (C.c_uint8 * 32)()
    rc = _lib.load().ls_host_hash_seed(seed.encode(), C.cast(out, C.c_void_p))
    if rc != 0:
        raise LsError(rc, "ls_host_hash_seed")
    return bytes(out)


def host_chacha20_block(key_words, w12, w13, w14, w15):
    key = (C.c_uint32 * 8)(*[int(k) for k in key_words])
    out = (C.c_uint32 * 16)()
    rc = _lib.load().ls_host_chacha20_block(C.cast(key, C.c_void_p), w12, w13, w14, w15, C.cast(out, C.c_void_p))
    if rc != 0:
        raise LsError(rc, "ls_host_chacha20_block")
    return [int(x) for x in out]


def host_rng_script(seed32: bytes, ops, weights=None, cap=1 << 16):
    seed = (C.c_uint8 * 32)(*seed32)
    o = _c(ops, np.int64).reshape(-1, 3)
    w = None if weights is None else _c(weights, np.float64)
    out = np.zeros(cap, dtype=np.uint64)
    n = C.c_int64()
    rc = _lib.load().ls_host_rng_script(C.cast(seed, C.c_void_p), _vp(o), len(o), _vp(w), _vp(out), cap, C.byref(n))
    if rc != 0:
        raise LsError(rc, "ls_host_rng_script")
    return [int(x) for x in out[:n.value]]
