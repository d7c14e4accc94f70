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


def _dev_eval(ctx, torch, rows, cur, cols, ncols, window, want_scores=True, algo="table"):
    dev = torch.device("cuda:0")
    R, n = rows.shape
    mc = cols.shape[1]
    t = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (rows.astype(np.int32), cur.astype(np.int64),
                                                                     cols.astype(np.int32), ncols.astype(np.int32))]
    res = torch.zeros((R, 4), dtype=torch.int64, device=dev)
    sc = torch.zeros((R, mc * n), dtype=torch.int64, device=dev) if want_scores else None
    ctx.set_algo(algo)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.nqueens_eval_batch_dev(R, n, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), mc, window,
                                   res.data_ptr(), sc.data_ptr() if want_scores else None)
        torch.cuda.synchronize()
    finally:
        ctx.set_stream(None)
    return res.cpu().numpy(), (sc.cpu().numpy() if want_scores else None)


@pytest.mark.parametrize("algo", ["scan", "table"])
@pytest.mark.parametrize("n", [96, 5000])           # register / LDS-table paths and the LDS scan / global-table paths
def test_unvalidated_device_entry_counts_what_it_scores(algo, n):
    # ls_nqueens_eval_batch_dev does not validate cols or rows (include/ls_amd.h): a slot whose column, or whose
    # rows[column], is outside [0, n) is skipped but keeps its n - 1 window positions; n_scored must count exactly
    # the candidates that were scored (the eval kernels and the finalize kernel must agree)
    torch = pytest.importorskip("torch")
    import ls_amd
    import oracle as orc
    rng = np.random.default_rng(n)
    with ls_amd.Context(0) as ctx:
        rows = rng.integers(0, n, size=(1, n))
        cur = np.array([orc.nq_score(rows[0]) if n < 1000 else int(orc.nq_col_scores_table(rows[0]).sum())])
        good = rng.permutation(n)[:5]
        cols = np.array([[good[0], n + 3, good[1], -1, good[2], good[3], good[4]]])
        for window in (10 ** 9, 3 * (n - 1) + 10, 1 * (n - 1) + 5, 5 * (n - 1) + 7):
            res, sc = _dev_eval(ctx, torch, rows, cur, cols, np.array([7]), window, algo=algo)
            scored = sc[0] != -(2 ** 63)
            assert res[0, 0] == scored.sum(), (window, res[0], scored.sum())
            per_slot = scored.reshape(7, n).sum(axis=1)
            assert per_slot[1] == 0 and per_slot[3] == 0                       # the bad columns contribute nothing ...
            exp = [max(0, min(n - 1, window - s * (n - 1))) if s not in (1, 3) else 0 for s in range(7)]
            assert per_slot.tolist() == exp                                    # ... but keep their window positions
            if res[0, 0] > 0:
                k = int(res[0, 1])
                assert scored[k] and sc[0][k] == sc[0][scored].min() == res[0, 2]
        # the fused LS-step kernel (next_dev entry) applies the same rule
        dev = torch.device("cuda:0")
        ctx.set_algo("auto")
        if n <= 8192:
            t = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (rows.astype(np.int32), cur.astype(np.int64),
                                                                             cols.astype(np.int32), np.array([7], np.int32))]
            res = torch.zeros((1, 4), dtype=torch.int64, device=dev)
            nxt, ncs = torch.zeros((1, n), dtype=torch.int32, device=dev), torch.zeros((1, n), dtype=torch.int64, device=dev)
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            try:
                ctx.nqueens_eval_batch_next_dev(1, n, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), 7,
                                                3 * (n - 1) + 10, res.data_ptr(), nxt.data_ptr(), ncs.data_ptr())
                torch.cuda.synchronize()
            finally:
                ctx.set_stream(None)
            assert res.cpu().numpy()[0, 0] == 2 * (n - 1)         # slots 0 and 2 in full; slots 1 and 3 (the window's last 10) are skipped
        # a row value outside [0, n) at a proposed column: that slot is skipped as well
        rows2 = rows.copy()
        rows2[0, good[1]] = n + 17
        res, sc = _dev_eval(ctx, torch, rows2, cur, np.array([[good[0], good[1], good[2]]]), np.array([3]), 10 ** 9, algo=algo)
        scored = sc[0] != -(2 ** 63)
        assert res[0, 0] == scored.sum() == 2 * (n - 1)
        assert scored.reshape(3, n)[1].sum() == 0


def test_workspaces_grow_under_queued_work_and_streams_switch():
    # DevBuf::reserve must not free a buffer that launches queued on the stream still use, and switching streams must not
    # let a new launch overtake the previous stream's finalize (which restores the argmin keys): asynchronous device-pointer
    # calls with growing batch sizes, alternating between two torch streams, never synchronised in between
    torch = pytest.importorskip("torch")
    import ls_amd
    import oracle as orc
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    n, mc = 700, 6
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    with ls_amd.Context(0) as ctx:
        for algo in ("table", "scan"):
            ctx.set_algo(algo)
            keep, expect = [], []
            for i, R in enumerate([1, 2, 3, 5, 9, 17, 40, 90, 200]):
                rows = rng.integers(0, n, size=(R, n)).astype(np.int32)
                cols = np.stack([rng.permutation(n)[:mc] for _ in range(R)]).astype(np.int32)
                cur = np.array([orc.nq_score(rows[r]) for r in range(R)], dtype=np.int64)
                st = streams[(i // 3) % 2]          # three growing launches back to back on one stream, then switch
                with torch.cuda.stream(st):
                    t = [torch.from_numpy(x).to(dev, non_blocking=True) for x in (rows, cur, cols, np.full(R, mc, np.int32))]
                    res = torch.zeros((R, 4), dtype=torch.int64, device=dev)
                    ctx.set_stream(st.cuda_stream)
                    ctx.nqueens_eval_batch_dev(R, n, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), mc,
                                               5 * n, res.data_ptr())
                keep.append((t, res))
                expect.append([orc.nq_neighbourhood(rows[r], cols[r], 5 * n, "table", want_scores=False) for r in range(R)])
            torch.cuda.synchronize()
            ctx.set_stream(None)
            for (t, res), exp in zip(keep, expect):
                got = res.cpu().numpy()
                for r, e in enumerate(exp):
                    assert (got[r, 0], got[r, 1], got[r, 2]) == (e["n_scored"], e["best_index"], e["best_score"]), (algo, len(exp), r)


def test_two_contexts_on_two_threads():
    # include/ls_amd.h: "different contexts may be driven concurrently from different threads" (the reference's
    # Solution/Score are Send, nothing is Sync: local_search.rs:16-23).  Two host threads, one context + stream each
    # on device 0, run different workloads at the same time -- including each kernel family's FIRST launch, which sets
    # function attributes -- and must reproduce the single-context results.
    import threading
    import ls_amd
    import oracle as orc

    def workload(seed, algo, n, R, mc, window, sched_emp):
        rng = np.random.default_rng(seed)
        rows = rng.integers(0, n, size=(R, n))
        cols = np.stack([rng.permutation(n)[:mc] for _ in range(R)])
        cur = [int(orc.nq_col_scores_table(rows[r]).sum()) for r in range(R)]
        state = rng.integers(0, sched_emp, size=32)
        moves = np.array([(1, int(a), int(b)) if t else (0, int(a), int(rng.integers(0, sched_emp)))
                          for t, (a, b) in zip(rng.integers(0, 5, size=300) > 0,
                                               (rng.choice(31, size=2, replace=False) for _ in range(300)))])
        return dict(algo=algo, rows=rows, cols=cols, cur=cur, window=window, mc=mc, state=state, moves=moves, emp=sched_emp)

    def run(ctx, w, reps):
        outs = []
        ctx.sched_instance_upload(31, 32, 0, w["emp"])
        for _ in range(reps):
            ctx.set_algo(w["algo"])
            o, _ = ctx.nqueens_eval_batch(w["rows"], w["cur"], w["cols"], np.full(len(w["rows"]), w["mc"]), w["window"])
            ctx.set_algo("auto")
            o2, nxt = ctx.nqueens_eval_batch_next(w["rows"], w["cur"], w["cols"], np.full(len(w["rows"]), w["mc"]), w["window"])
            cs, tot = ctx.nqueens_col_scores(w["rows"])
            s, _, _ = ctx.sched_eval(w["state"], w["moves"], 100)
            outs.append((o, o2, nxt.tobytes(), cs.tobytes(), tot.tobytes(), s))
        return outs

    wa = workload(1, "scan", 3000, 3, 40, 5 * 3000, 7)
    wb = workload(2, "table", 4096, 5, 204, 20480, 300)
    results = {}
    errors = []

    def thread(name, w):
        try:
            with ls_amd.Context(0) as c:
                results[name] = run(c, w, 25)
        except BaseException as e:  # noqa: BLE001
            errors.append((name, repr(e)))

    ts = [threading.Thread(target=thread, args=("a", wa)), threading.Thread(target=thread, args=("b", wb))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    with ls_amd.Context(0) as c:
        ref_a, ref_b = run(c, wa, 1)[0], run(c, wb, 1)[0]
    assert all(r == ref_a for r in results["a"])
    assert all(r == ref_b for r in results["b"])
    for w, ref in ((wa, ref_a), (wb, ref_b)):                                   # and the reference itself against the oracle
        for r in range(len(w["rows"])):
            e = orc.nq_neighbourhood(w["rows"][r], w["cols"][r], w["window"], "table", want_scores=False)
            assert (ref[0][r]["n_scored"], ref[0][r]["best_index"], ref[0][r]["best_score"]) == \
                   (e["n_scored"], e["best_index"], e["best_score"])
