"""GPU parity tests of the resident LS-step server (csrc/nq_server.hip): ONE never-relaunched kernel whose workgroups are
the slots; a one-restart host call is a 64-byte mailbox write and a 32-byte answer, the slot's board and count tables
stay in its workgroup's LDS.  Everything here goes through the C ABI and is compared with the CPU oracle, with the
launch-per-call route (LS_NQ_STEP=3) and with single-solver runs.

The multi-solver tests are the instantiation `bench.py`'s config-4 end-to-end leg times (free-running slots as fibres,
n = 4096 and 1024): VERDICT r2 "the end-to-end path that is TIMED is not the one that is parity-TESTED"."""
import time

import numpy as np
import pytest

import ls_oracle as lo
import oracle as orc

pytestmark = pytest.mark.gpu


def _board(rng, n, kind):
    return rng.permutation(n) if kind == "perm" else rng.integers(0, n, size=n)


def _expect(rows, cols, window):
    exp = orc.nq_neighbourhood(rows, cols, window, "table", want_scores=False) if len(cols) else \
        dict(n_scored=0, best_index=-1, best_score=0)
    b = np.array(rows).copy()
    if exp["best_index"] >= 0:
        b[cols[exp["best_index"] // len(b)]] = exp["best_index"] % len(b)
    return exp, b, orc.nq_col_scores(b)


def _step(c, rows, cols, window, with_next=True):
    cur = int(orc.nq_col_scores_table(rows).sum())
    exp, board, cs = _expect(rows, cols, window)
    cc = np.asarray(cols, dtype=np.int64).reshape(1, -1) if len(cols) else np.zeros((1, 1), dtype=np.int64)
    if with_next:
        out, nxt = c.nqueens_eval_batch_next(rows.reshape(1, -1), [cur], cc, [len(cols)], window)
        assert np.array_equal(nxt[0], cs)
    else:
        out, _ = c.nqueens_eval_batch(rows.reshape(1, -1), [cur], cc, [len(cols)], window)
    o = out[0]
    assert (o["n_scored"], o["best_index"]) == (exp["n_scored"], exp["best_index"])
    if exp["best_index"] >= 0:
        n = len(rows)
        assert o["best_score"] == exp["best_score"] == cs.sum()
        assert o["best_col"] == cols[exp["best_index"] // n] and o["best_value"] == exp["best_index"] % n
    else:
        assert (o["best_score"], o["best_col"], o["best_value"]) == (0, -1, -1)
    return board


def test_server_follows_any_call_sequence(monkeypatch):
    # Whatever the caller does next -- continue from the winner (local_search.rs:335: nothing is uploaded, the workgroup's
    # LDS board and tables carry the move), jump to an unrelated board (perturbation, iterated_local_search.rs:192-194),
    # repeat a call, ask for an empty neighbourhood, change the board size -- each call equals a stateless evaluation.
    import ls_amd
    rng = np.random.default_rng(4242)
    with ls_amd.Context(0) as c:
        # (n, columns, window): odd n = one candidate per lane, even n >= 128 = paired wavefront teams; windows that cut a
        # column, fill exactly, exceed the list; more than ten reachable columns = the column list travels out of line
        shapes = ((2, 2, 10), (3, 3, 4), (8, 1, 40), (33, 4, 100), (64, 3, 64), (127, 5, 5 * 127), (128, 6, 5 * 128 + 3),
                  (300, 9, 1500), (300, 40, 15 * 299 + 7), (1024, 51, 5120), (1025, 11, 10 ** 9), (4096, 204, 20480),
                  (4096, 12, 12 * 4095), (8192, 12, 40960), (8192, 7, 65535), (500, 3, 1))
        expected_requests = 0
        for n, mc, window in shapes:
            rows = _board(rng, n, "uniform")
            for it in range(10):
                cols = rng.permutation(n)[:mc] if mc <= n else rng.integers(0, n, size=mc)
                if mc >= 3 and it % 3 == 0:
                    cols[2] = cols[0]                         # a duplicated column: the earlier slot must win ties
                k = int(rng.integers(1, mc + 1))
                nxt = _step(c, rows, cols[:k], window, with_next=it % 5 != 4)
                what = it % 4
                if what in (0, 1):
                    rows = nxt                                # continue from the winner
                elif what == 2:
                    rows = _board(rng, n, "perm")             # perturbed: the slot reads the new board from its pinned shadow
                # what == 3: the same board again (the slot holds the WINNER now: it must notice)
            _step(c, rows, np.zeros(0, dtype=np.int64), window)          # empty neighbourhood: the slot's board is unchanged
            rows = _step(c, rows, rng.permutation(n)[:mc] if mc <= n else rng.integers(0, n, size=mc), window)
            for it in range(30):                              # a long walk: LDS tables and host-kept column scores must not drift
                rows = _step(c, rows, rng.permutation(n)[:min(mc, n)][:int(rng.integers(1, min(mc, n) + 1))], window)
            cs, tot = c.nqueens_col_scores(rows)              # one board: served by the slot as well
            assert np.array_equal(cs[0], orc.nq_col_scores(rows)) and tot[0] == cs[0].sum()
            cnt = c.counters()                                # (a context's own server is rebuilt when the board size changes:
            expected_requests += 43                           #  the context keeps the totals of the servers it retired)
            assert cnt["server_requests"] == expected_requests   # the server is what ran: every call above was one request
            assert cnt["completion_word_fallbacks"] == 0 and cnt["resident_board_reuploads"] == 0


def test_server_leaves_when_idle_and_comes_back(monkeypatch):
    # The launch must never outlive its use: with no request for LS_NQ_SERVER_IDLE_US every workgroup leaves; the next call
    # starts it again and the slot, which has lost its LDS state, reads the board from the pinned shadow (which the host
    # keeps up to date with every winner).  Same for the age limit (LS_NQ_SERVER_LIMIT_MS) under a busy caller.
    import ls_amd
    monkeypatch.setenv("LS_NQ_SERVER_IDLE_US", "300")
    monkeypatch.setenv("LS_NQ_SERVER_LIMIT_MS", "8")
    rng = np.random.default_rng(7)
    n, mc, window = 1024, 51, 5120
    with ls_amd.Context(0) as c:
        rows = _board(rng, n, "uniform")
        for it in range(6):
            rows = _step(c, rows, rng.permutation(n)[:mc], window)
            time.sleep(0.02)                                   # far beyond the idle time-out
        launches = c.counters()["server_launches"]
        assert launches >= 5
        t0 = time.time()
        while time.time() - t0 < 0.1:                          # a busy caller crosses the 8 ms age limit several times
            rows = _step(c, rows, rng.permutation(n)[:mc], window)
        assert c.counters()["server_launches"] >= launches + 2


def test_server_equals_launch_per_call_route(monkeypatch):
    # the same call sequence through the resident server (LS_NQ_STEP=4, default) and through one launch per call (=3)
    import ls_amd
    rng = np.random.default_rng(99)
    seqs = []
    for n, mc in ((4096, 204), (8192, 30), (1000, 50), (999, 50)):
        rows = _board(rng, n, "uniform")
        seqs.append((n, rows, [(rng.permutation(n)[:int(rng.integers(1, mc + 1))], it % 7 == 6) for it in range(25)]))

    def run(mode):
        monkeypatch.setenv("LS_NQ_STEP", mode)
        prng = np.random.default_rng(12345)                    # the perturbations: the same in both passes
        got = []
        with ls_amd.Context(0) as c:
            for n, rows0, calls in seqs:
                rows = rows0.copy()
                for cols, perturb in calls:
                    cur = int(orc.nq_col_scores_table(rows).sum())
                    out, nxt = c.nqueens_eval_batch_next(rows.reshape(1, -1), [cur], cols.reshape(1, -1), [len(cols)], 5 * n)
                    got.append((out[0], nxt[0].copy()))
                    rows = rows.copy()
                    if out[0]["best_index"] >= 0:
                        rows[out[0]["best_col"]] = out[0]["best_value"]
                    if perturb:
                        rows[prng.integers(0, n, size=n // 20)] = 0
            assert (c.counters()["server_requests"] > 0) == (mode == "4")
        return got

    g4, g3 = run("4"), run("3")
    assert len(g4) == len(g3) == 100
    for (o4, c4), (o3, c3) in zip(g4, g3):
        assert o4 == o3 and np.array_equal(c4, c3)


def test_two_contexts_two_servers_interleaved():
    import ls_amd
    rng = np.random.default_rng(5)
    with ls_amd.Context(0) as a, ls_amd.Context(0) as b:
        ra, rb = _board(rng, 512, "uniform"), _board(rng, 2048, "perm")
        for it in range(20):
            ra = _step(a, ra, rng.permutation(512)[:6], 5 * 512)
            rb = _step(b, rb, rng.permutation(2048)[:8], 5 * 2048)
        assert a.counters()["server_requests"] >= 20 and b.counters()["server_requests"] >= 20


def test_server_route_argument_errors_leave_the_slot_usable():
    # the board of a one-restart call is range-checked where it is compared with the slot's shadow; a bad call must fail with the
    # reference's own failure mode translated (an error code, not a panic) and leave the slot consistent for the next call
    import ls_amd
    rng = np.random.default_rng(11)
    n = 600
    with ls_amd.Context(0) as c:
        rows = _board(rng, n, "uniform")
        rows = _step(c, rows, rng.permutation(n)[:6], 5 * n)
        bad = rows.copy()
        bad[17] = n                                              # outside [0, n)
        with pytest.raises(ls_amd.LsError) as ei:
            c.nqueens_eval_batch_next(bad.reshape(1, -1), [0], np.arange(3).reshape(1, -1), [3], 5 * n)
        assert ei.value.code == ls_amd._lib.LS_ERR_INVALID and "rows[17]" in str(ei.value)
        with pytest.raises(ls_amd.LsError):
            c.nqueens_eval_batch_next(rows.reshape(1, -1), [0], np.array([[1, n]]), [2], 5 * n)   # a column outside [0, n)
        for _ in range(5):                                       # the slot still holds the last good board
            rows = _step(c, rows, rng.permutation(n)[:6], 5 * n)
        assert c.counters()["server_boards_uploaded"] == 1       # the rejected board never reached the shadow


def test_batched_step_on_server_slots_and_its_error_path():
    # A batched call takes one server slot per restart (all requests in flight at once).  Walks of four restarts continuing
    # from their winners, one restart perturbed in between; then a batch whose THIRD board is out of range: the call fails, the
    # two requests already posted are answered and dropped, and the slots serve the next (valid) batch correctly.
    import ls_amd
    rng = np.random.default_rng(23)
    n, R, mc, window = 1024, 4, 9, 5 * 1024
    with ls_amd.Context(0) as c:
        rows = np.stack([_board(rng, n, "uniform") for _ in range(R)])

        def batch(rows):
            cols = np.stack([rng.permutation(n)[:mc] for _ in range(R)])
            cur = [int(orc.nq_col_scores_table(rows[r]).sum()) for r in range(R)]
            out, nxt = c.nqueens_eval_batch_next(rows, cur, cols, [mc] * R, window)
            boards = []
            for r in range(R):
                exp, board, cs = _expect(rows[r], cols[r], window)
                assert (out[r]["n_scored"], out[r]["best_index"]) == (exp["n_scored"], exp["best_index"]), r
                assert np.array_equal(nxt[r], cs), r
                boards.append(board)
            return np.stack(boards)

        for it in range(12):
            rows = batch(rows)
            if it == 5:
                rows[2] = _board(rng, n, "perm")
        before = c.counters()["server_requests"]
        assert before >= 12 * R                                  # the server is what ran
        bad = rows.copy()
        bad[2, 100] = -1
        with pytest.raises(ls_amd.LsError) as ei:
            c.nqueens_eval_batch_next(bad, [0] * R, np.tile(np.arange(mc), (R, 1)), [mc] * R, window)
        assert ei.value.code == ls_amd._lib.LS_ERR_INVALID and "rows[100]" in str(ei.value)
        for it in range(4):                                      # (restarts 0 and 1 moved on the device: their boards are re-read)
            rows = batch(rows)
        assert c.counters()["completion_word_fallbacks"] == 0


def test_servers_of_two_contexts_on_two_threads():
    import threading
    import ls_amd
    errs = []

    def work(seed, n):
        try:
            rng = np.random.default_rng(seed)
            with ls_amd.Context(0) as c:
                rows = _board(rng, n, "uniform")
                for it in range(60):
                    rows = _step(c, rows, rng.permutation(n)[:8][:int(rng.integers(1, 9))], 5 * n)
                    if it % 17 == 16:
                        rows = _board(rng, n, "perm")
                assert c.counters()["server_requests"] == 60
        except BaseException as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(s, n)) for s, n in ((1, 1024), (2, 4096), (3, 333))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_multi_solver_more_slots_than_one_server_launch_holds():
    # n_slots beyond what one resident launch holds (240): the slots fall back to one launch per call; results unchanged
    import ls_amd
    n, rounds = 24, 3
    seeds = [str(s) for s in range(1000, 1000 + 250)]
    p = _nq_params(n)
    with ls_amd.Context(0) as c:
        singles, iters = _single_runs(c, n, seeds[:12], p, rounds)
        for slots in (250, 240):
            m = ls_amd.MultiSolver(c, n, seeds, p, n_slots=slots, mode="free")
            m.run(max_rounds_per_restart=rounds)
            assert [m.best(r) for r in range(12)] == singles
            assert [m.restart_stats(r)["ls_iterations"] for r in range(12)] == iters
            assert (m.counters()["server_requests"] > 0) == (slots <= 240)
            m.close()


def _nq_params(n, expiry=10_000):
    p = dict(lo.NQUEENS_MAIN_PARAMS)
    p["window_size"] = 5 * n
    p["all_solution_iteration_expiry"] = expiry
    return p


def _single_runs(ctx, n, seeds, p, rounds):
    import ls_amd
    singles, iters = [], []
    for seed in seeds:
        s = ls_amd.Solver.nqueens(ctx, n, seed, p)
        for _ in range(rounds):
            s.execute_round()
            b = s.best()
            if b is not None and b[0] == 0:
                break
        singles.append(s.best())
        iters.append(s.stats()["ls_iterations"])
        s.close()
    return singles, iters


@pytest.mark.parametrize("n,n_seeds,rounds,n_slots", [(4096, 20, 2, 20), (1024, 40, 3, 40), (4096, 9, 2, 4)])
def test_multi_solver_server_slots_equal_single_launched_runs(monkeypatch, n, n_seeds, rounds, n_slots):
    # config 4's end-to-end instantiation: free-running slots (more slots than the 16 cores a GPU box gives a process =>
    # cooperative fibres), every slot a workgroup of the resident server, n = 4096 / 1024.  Every restart must equal its
    # single-solver run on the launch-per-call route (LS_NQ_STEP=3: nq_step_kernel<true>, a different code path), result
    # and LS-iteration count.  (n_slots < n_seeds: a finished restart's slot takes the next seed.)
    import ls_amd
    seeds = [str(s) for s in range(42, 42 + n_seeds)]        # examples/nqueens/src/main.rs:167 seed convention
    p = _nq_params(n)
    monkeypatch.setenv("LS_NQ_STEP", "3")
    with ls_amd.Context(0) as c3:
        singles, iters = _single_runs(c3, n, seeds, p, rounds)
        assert c3.counters()["server_requests"] == 0
    monkeypatch.setenv("LS_NQ_STEP", "4")
    with ls_amd.Context(0) as c4:
        m = ls_amd.MultiSolver(c4, n, seeds, p, n_slots=n_slots, mode="free")
        m.run(max_rounds_per_restart=rounds)
        assert [m.best(r) for r in range(n_seeds)] == singles
        st, cnt = m.stats(), m.counters()
        assert st["ls_iterations"] == sum(iters)
        assert cnt["server_requests"] >= st["ls_iterations"]  # every LS iteration was one mailbox request (+ the score() calls)
        assert cnt["completion_word_fallbacks"] == 0
        m.close()
        # the single solver on the server route as well (its own one-slot server)
        s4, i4 = _single_runs(c4, n, seeds[:3], p, rounds)
        assert s4 == singles[:3] and i4 == iters[:3]


def test_multi_solver_fibres_survive_server_relaunches(monkeypatch):
    # A launch of the server that ages out (or idles out) while requests are posted is started again by whoever waits on it
    # (SrvHost::wait).  Under the fibre pool a waiter is only resumed early when its answer HAS arrived -- so a worker whose
    # fibres all wait must resume them anyway after a while, or nobody ever relaunches (this hung config 4 on 4 host cores).
    # 4 cores, 16 slots, an age limit of 3 ms: dozens of relaunches under load; results as on the launch-per-call route.
    import os
    import ls_amd
    n, rounds = 1024, 2
    seeds = [str(s) for s in range(42, 42 + 16)]
    p = _nq_params(n)
    monkeypatch.setenv("LS_NQ_STEP", "3")
    with ls_amd.Context(0) as c3:
        singles, iters = _single_runs(c3, n, seeds, p, rounds)
    monkeypatch.setenv("LS_NQ_STEP", "4")
    monkeypatch.setenv("LS_NQ_SERVER_LIMIT_MS", "3")
    monkeypatch.setenv("LS_NQ_SERVER_IDLE_US", "100")
    before = os.sched_getaffinity(0)
    os.sched_setaffinity(0, set(sorted(before)[:4]))
    try:
        with ls_amd.Context(0) as c:
            m = ls_amd.MultiSolver(c, n, seeds, p, n_slots=16, mode="free")
            m.run(max_rounds_per_restart=rounds)
            assert [m.best(r) for r in range(len(seeds))] == singles
            assert m.stats()["ls_iterations"] == sum(iters)
            assert m.counters()["server_launches"] >= 5
            m.close()
    finally:
        os.sched_setaffinity(0, before)


def test_multi_solver_server_n1024_follows_the_oracle_driver():
    # one restart of the multi-solver (server slots, fibres) against the oracle's literal Python driver, round for round
    import ls_amd
    n, rounds = 1024, 2
    p = _nq_params(n)
    seeds = [str(s) for s in range(42, 42 + 24)]
    o = lo.IteratedLocalSearch(lo.NQueens(n), p, seeds[5])
    for _ in range(rounds):
        o.execute_round()
        b = o.get_best_solution()
        if b is not None and b[0] == 0:
            break
    with ls_amd.Context(0) as c:
        m = ls_amd.MultiSolver(c, n, seeds, p, n_slots=24, mode="free")
        m.run(max_rounds_per_restart=rounds)
        assert m.best(5) == o.get_best_solution()
        m.close()


def test_multi_solver_lockstep_large_n_equals_single_runs():
    import ls_amd
    n, rounds = 1024, 2
    p = _nq_params(n)
    seeds = [str(s) for s in range(42, 50)]
    with ls_amd.Context(0) as c:
        singles, iters = _single_runs(c, n, seeds, p, rounds)
        m = ls_amd.MultiSolver(c, n, seeds, p, n_slots=8, mode="lockstep")
        m.run(max_rounds_per_restart=rounds)
        assert [m.best(r) for r in range(len(seeds))] == singles
        assert m.stats()["ls_iterations"] == sum(iters)
        m.close()


# ------------------------------------------------------------------------------------------------- scheduling
def _random_moves(rng, n, D, E):
    mv = np.zeros((n, 3), dtype=np.int64)
    for i in range(n):
        if rng.integers(0, 5) == 0:
            mv[i] = (0, rng.integers(0, D), rng.integers(0, E))       # ChangeDay (lib.rs:466-470)
        else:
            a, b = rng.choice(D, size=2, replace=False)
            mv[i] = (1, a, b)                                          # SwapDays (:471-478)
    return mv


@pytest.mark.parametrize("dates,E", [(((2022, 5, 9), (2022, 6, 8)), 7),        # the example instance: 31 scored / 32 stored days
                                     (((2022, 5, 9), (2022, 7, 10)), 40),       # 63 / 64 days: the 64-bit day sets
                                     (((2022, 5, 11), (2022, 5, 24)), 3),       # 14 / 15 days, starts on a Wednesday
                                     (((2022, 5, 9), (2022, 6, 8)), 8_000),     # the day-set table fits the slot's LDS, the holiday words do not
                                     (((2022, 5, 9), (2022, 6, 8)), 12_000)])   # too many employees for the LDS look-up table: day-by-day scorers
def test_sched_server_follows_any_call_sequence(monkeypatch, dates, E):
    # ls_sched_eval_resident / ls_sched_score through the resident scheduling workgroup (no launch per call): continue from
    # the winner (the workgroup applies it to its LDS state), perturb, repeat, windows that cut / exceed the list, a new
    # resident list and a new instance under a live server, idle time-out; every call against the oracle and, for the same
    # sequence, against the launch-per-call route.
    import ls_amd
    monkeypatch.setenv("LS_NQ_SERVER_IDLE_US", "500")
    rng = np.random.default_rng(77)
    hol = {e: list(rng.choice(10, size=int(rng.integers(0, 3)), replace=False)) for e in range(E)}
    inst = orc.Sched(dates[0], dates[1], hol)
    D, S = inst.n_scored_days, inst.n_scored_days + 1
    bits = np.zeros(E, dtype=np.uint64)
    for e, ds in hol.items():
        for d in ds:
            bits[e] |= np.uint64(1) << np.uint64(int(d))
    results = {}
    for server in ("1", "0"):   # resident | launch per call
        monkeypatch.setenv("LS_SCHED_SERVER", server)
        prng = np.random.default_rng(5)
        got = []
        with ls_amd.Context(0) as c:
            c.sched_instance_upload(D, S, inst.start_weekday, E, bits)
            moves = _random_moves(prng, 400, D, E)
            c.sched_moves_upload(moves)
            state = prng.integers(0, E, size=S)
            for it in range(60):
                k = int(prng.integers(1, len(moves) + 1))
                w = int(prng.choice([1, 7, 100, 299, 399, 400, 10 ** 6]))
                r = c.sched_eval_resident(state, k, w)
                exp = inst.neighbourhood(state, moves[:k], w)
                assert (r["n_scored"], r["best_index"]) == (exp["n_scored"], exp["best_index"]), (it, k, w)
                if exp["best_index"] >= 0:
                    assert (r["best_hard"], r["best_soft"]) == (exp["best_hard"], exp["best_soft"])
                got.append(r)
                what = it % 5
                if what in (0, 1, 2) and r["best_index"] >= 0:           # continue from the winner (local_search.rs:335)
                    t, a, b = moves[r["best_index"]]
                    state = state.copy()
                    if t == 0:
                        state[a] = b
                    else:
                        state[a], state[b] = state[b], state[a]
                elif what == 3:
                    state = prng.integers(0, E, size=S)                  # perturbed
                    h, s_ = c.sched_score(state)                         # score of a fresh state: the same workgroup
                    assert (int(h[0]), int(s_[0])) == inst.score(state)
                if it == 20:
                    moves = _random_moves(prng, 300, D, E)               # a new resident list under a live server
                    c.sched_moves_upload(moves)
                if it == 35:
                    time.sleep(0.01)                                     # the launch leaves (idle) and comes back
                if it == 45:
                    c.sched_instance_upload(D, S, inst.start_weekday, E, bits)   # the instance again: the launch is restarted
            cnt = c.counters()
            assert (cnt["server_requests"] > 60) == (server != "0")
            if server != "0":
                assert cnt["server_launches"] >= 3
        results[server] = got
    assert results["1"] == results["0"]


def test_sched_grid_server_long_lists(monkeypatch):
    # ls_sched_eval_resident for lists beyond one workgroup: the resident GRID (one workgroup per 512-move chunk, a device-side
    # go record, the last workgroup finishes and scores the cut chunk again).  Windows on, just before and just after chunk
    # boundaries' cumulative candidate counts, beyond the list, 1; short requests on a long resident list; idle relaunch;
    # a new list under a live grid; against the oracle and against the two-launch route.
    import ls_amd
    monkeypatch.setenv("LS_NQ_SERVER_IDLE_US", "500")
    rng = np.random.default_rng(2024)
    E = 23
    hol = {e: list(rng.choice(20, size=int(rng.integers(0, 3)), replace=False)) for e in range(E)}
    inst = orc.Sched((2022, 5, 9), (2022, 6, 8), hol)
    D, S = inst.n_scored_days, inst.n_scored_days + 1
    bits = np.zeros(E, dtype=np.uint64)
    for e, ds in hol.items():
        for d in ds:
            bits[e] |= np.uint64(1) << np.uint64(int(d))
    results = {}
    for server in ("1", "0"):   # resident | launch per call
        monkeypatch.setenv("LS_SCHED_SERVER", server)
        prng = np.random.default_rng(9)
        got = []
        with ls_amd.Context(0) as c:
            c.sched_instance_upload(D, S, inst.start_weekday, E, bits)
            moves = _random_moves(prng, 2600, D, E)                     # 6 chunks, the last one partial
            c.sched_moves_upload(moves)
            state = prng.integers(0, E, size=S)
            for it in range(40):
                k = int(prng.choice([len(moves), len(moves) - 1, 513, 1025, int(prng.integers(513, len(moves) + 1))]))
                exp_all = inst.neighbourhood(state, moves[:k], 10 ** 9)
                real = exp_all["n_scored"]                               # real candidates among the first k moves
                w = int(prng.choice([1, 2, real - 1, real, real + 1, 10 ** 9, int(prng.integers(1, real + 1)), 400, 420, 840]))
                w = max(1, w)
                r = c.sched_eval_resident(state, k, w)
                exp = inst.neighbourhood(state, moves[:k], w)
                assert (r["n_scored"], r["best_index"]) == (exp["n_scored"], exp["best_index"]), (it, k, w)
                if exp["best_index"] >= 0:
                    assert (r["best_hard"], r["best_soft"]) == (exp["best_hard"], exp["best_soft"])
                got.append(r)
                if it % 3 != 2 and r["best_index"] >= 0:
                    t, a, b = moves[r["best_index"]]
                    state = state.copy()
                    if t == 0:
                        state[a] = b
                    else:
                        state[a], state[b] = state[b], state[a]
                else:
                    state = prng.integers(0, E, size=S)
                if it == 15:
                    time.sleep(0.01)                                     # the grid leaves (idle) and comes back
                if it == 25:
                    moves = _random_moves(prng, 1500, D, E)              # a new, shorter list: another grid size
                    c.sched_moves_upload(moves)
            cnt = c.counters()
            assert (cnt["server_requests"] >= 40) == (server != "0")
        results[server] = got
    assert results["1"] == results["0"]


def test_sched_exhaustive_grid(monkeypatch):
    # ls_sched_eval_exhaustive for ONE state on the resident grid (sched_grid_server_kernel<GEN>): the S x E candidates of the
    # exhaustive proposer (lib.rs:493-559) are generated in place, a workgroup takes up to four consecutive 512-candidate
    # chunks, one record per workgroup, the finisher is the workgroup the window cut last time.  Windows that cut inside the
    # first / a middle / the last chunk of a workgroup, on workgroup seams, beyond everything, 1; states with few distinct
    # employees (many no-ops per chunk); alternating with the resident-STREAM grid (one grid at a time: each switch relaunches);
    # a new instance under a live grid.  Against the oracle and against the launched pipeline (LS_SCHED_SERVER=0).
    import ls_amd
    monkeypatch.setenv("LS_NQ_SERVER_IDLE_US", "500")
    rng = np.random.default_rng(4242)
    results = {}
    for server in ("1", "0"):
        monkeypatch.setenv("LS_SCHED_SERVER", server)
        prng = np.random.default_rng(17)
        got = []
        with ls_amd.Context(0) as c:
            for E, span in ((9000, ((2022, 5, 9), (2022, 6, 8))), (700, ((2022, 5, 9), (2022, 6, 8))), (300, ((2022, 5, 9), (2022, 6, 20)))):
                hol = {int(e): list(prng.choice(20, size=2, replace=False)) for e in prng.choice(E, size=E // 3, replace=False)}
                inst = orc.Sched(span[0], span[1], hol)
                D, S = inst.n_scored_days, inst.n_scored_days + 1
                bits = np.zeros(E, dtype=np.uint64)
                for e, ds in hol.items():
                    for d in ds:
                        bits[e] |= np.uint64(1) << np.uint64(int(d))
                c.sched_instance_upload(D, S, inst.start_weekday, E, bits)
                per = S * E
                stream = _random_moves(prng, 3000, D, E)
                c.sched_moves_upload(stream)
                state = prng.integers(0, E, size=S)
                for it in range(10 if E == 9000 else 14):
                    if it % 4 == 3:
                        state = prng.integers(0, 5, size=S)                # few distinct employees
                    real = S * (E - 1)
                    w = int(prng.choice([10 ** 9, real, real - 1, 1, 0, 511, 512, 513, 1536, 1537, per // 2, int(prng.integers(1, real)),
                                         int(prng.integers(1, real))]))
                    r = c.sched_eval_exhaustive(state[None, :], w)[0]
                    exp = inst.neighbourhood(state, orc.Sched.exhaustive_moves(state, range(E)), w)
                    assert (r["n_scored"], r["best_index"]) == (exp["n_scored"], exp["best_index"]), (server, E, it, w)
                    if exp["best_index"] >= 0:
                        assert (r["best_hard"], r["best_soft"]) == (exp["best_hard"], exp["best_soft"])
                    got.append(r)
                    if it % 3 == 1:                                        # the stream grid in between: the other kind of request
                        rs = c.sched_eval_resident(state, len(stream), 2500)
                        es = inst.neighbourhood(state, stream, 2500)
                        assert (rs["n_scored"], rs["best_index"]) == (es["n_scored"], es["best_index"])
                        got.append(rs)
                    if r["best_index"] >= 0 and it % 2 == 0:               # continue from the winner
                        d, i = divmod(r["best_index"], E)
                        state = state.copy()
                        state[d] = (state[d] + 1 + i) % E
                    if it == 6:
                        time.sleep(0.01)                                   # the launch leaves (idle) and comes back
            cnt = c.counters()
            assert (cnt["server_requests"] > 30) == (server != "0")
        results[server] = got
    assert results["1"] == results["0"]


@pytest.mark.parametrize("E,chunks_per_workgroup", [(5000, 2), (16000, 4)])
def test_sched_exhaustive_grid_chunks_per_workgroup(E, chunks_per_workgroup):
    # the exhaustive grid with two and with four consecutive chunks per workgroup (E = 9000 in the test above gives three):
    # windows beyond everything, inside a workgroup's second / last chunk (first request: the finisher is workgroup 0 and has
    # to re-score another workgroup's chunks; repeated: the cut workgroup is the finisher and uses the keys it holds), 1
    import ls_amd
    rng = np.random.default_rng(E)
    hol = {int(e): [int(rng.integers(0, 31))] for e in rng.choice(E, size=E // 4, replace=False)}
    inst = orc.Sched((2022, 5, 9), (2022, 6, 8), hol)
    D, S = inst.n_scored_days, inst.n_scored_days + 1
    bits = np.zeros(E, dtype=np.uint64)
    for e, ds in hol.items():
        for d in ds:
            bits[e] |= np.uint64(1) << np.uint64(int(d))
    per = S * E
    assert (((per + 511) // 512) + 255) // 256 == chunks_per_workgroup
    with ls_amd.Context(0) as c:
        c.sched_instance_upload(D, S, inst.start_weekday, E, bits)
        state = rng.integers(0, E, size=S)
        moves = orc.Sched.exhaustive_moves(state, range(E))
        span = 512 * chunks_per_workgroup
        for w in (10 ** 9, 7 * span + 512 + 77, 7 * span + 512 + 77, 100 * span + span - 3, 100 * span + span - 3, per // 2, 1):
            r = c.sched_eval_exhaustive(state[None, :], w)[0]
            exp = inst.neighbourhood(state, moves, w)
            assert (r["n_scored"], r["best_index"]) == (exp["n_scored"], exp["best_index"]), (E, w)
            assert (r["best_hard"], r["best_soft"]) == (exp["best_hard"], exp["best_soft"])
        assert c.counters()["server_requests"] >= 7


def test_two_scheduling_grids_that_cannot_both_be_resident():
    # The grid's workgroups wait for each other, one per CU: two contexts with 70 000-move lists (137 workgroups each) do not
    # fit a 256-CU device together.  Taking turns, each finds the CUs free after the other's idle time-out (a relaunch per turn);
    # with one context BUSY on another thread, the second one's launches leave unanswered (its workgroups are never all
    # resident), it gives the grid up after two of them and answers through the launched pipeline: same results, no error, no
    # hang (the first version of this test hung: an aborting launch served the pending request).
    import threading
    import ls_amd
    rng = np.random.default_rng(606)
    E = 500
    hol = {int(e): [int(rng.integers(0, 31))] for e in rng.choice(E, size=100, replace=False)}
    inst = orc.Sched((2022, 5, 9), (2022, 6, 8), hol)
    D, S = inst.n_scored_days, inst.n_scored_days + 1
    bits = np.zeros(E, dtype=np.uint64)
    for e, ds in hol.items():
        for d in ds:
            bits[e] |= np.uint64(1) << np.uint64(int(d))
    moves = _random_moves(rng, 70_000, D, E)
    state = rng.integers(0, E, size=S)
    exp = {w: inst.neighbourhood(state, moves, w) for w in (65_536, 10 ** 9)}

    def check(c, tag):
        for w, e in exp.items():
            r = c.sched_eval_resident(state, len(moves), w)
            assert (r["n_scored"], r["best_index"], r["best_hard"], r["best_soft"]) == \
                   (e["n_scored"], e["best_index"], e["best_hard"], e["best_soft"]), (tag, w)

    with ls_amd.Context(0) as a, ls_amd.Context(0) as b:
        if a.device_info()["n_cus"] >= 2 * ((len(moves) + 511) // 512):
            pytest.skip("both grids fit this device")
        for c in (a, b):
            c.sched_instance_upload(D, S, inst.start_weekday, E, bits)
            c.sched_moves_upload(moves)
        for it in range(3):                                            # taking turns
            check(a, ("turns a", it))
            check(b, ("turns b", it))
        assert a.counters()["server_requests"] >= 6 and b.counters()["server_requests"] >= 6
        stop, errors = threading.Event(), []

        def busy():
            try:
                while not stop.is_set():
                    check(a, "busy a")
            except Exception as e:  # noqa: BLE001 - reported by the main thread
                errors.append(e)

        t = threading.Thread(target=busy)
        t.start()
        try:
            time.sleep(0.02)                                           # a's grid is resident and stays busy
            before = b.counters()
            for it in range(4):
                check(b, ("b beside a busy a", it))
            after = b.counters()
        finally:
            stop.set()
            t.join()
        assert not errors, errors
        # b's launches left unanswered at least twice, then it stopped asking its grid (4 x 2 calls, at most a few requests)
        assert after["server_waits_on_drained_launch"] - before["server_waits_on_drained_launch"] >= 2
        assert after["server_requests"] - before["server_requests"] <= 2
        assert a.counters()["completion_word_fallbacks"] == 0 and after["completion_word_fallbacks"] == 0


# ------------------------------------------------------------------------------------ many scheduling restarts per GPU
SCHED_EXAMPLE = dict(local_search_max_iterations=1_000, window_size=100, best_solutions_capacity=64, all_solutions_capacity=100_000,
                     all_solution_iteration_expiry=1_000, iterated_local_search_max_iterations=250, max_allow_no_improvement_for=20)


def _single_sched_run(c, D, wd, E, seed, params, bits, rounds):
    import ls_amd
    s = ls_amd.Solver.scheduling(c, D, wd, E, seed, params, holiday_bits=bits)
    done = 0
    while not s.is_finished() and done < rounds:
        s.execute_round()
        done += 1
    out = (s.best(), s.stats()["ls_iterations"], s.stats()["neighbours_scored"], done)
    s.close()
    return out


@pytest.mark.parametrize("n_seeds,n_slots,rounds", [(64, 64, 12), (9, 4, 20)])
def test_multi_sched_step_server_slots_equal_single_runs(monkeypatch, n_seeds, n_slots, rounds):
    # ls_multi_solver_create_sched on the example instance (config 3: window 100): the restarts are the workgroups of ONE resident
    # scheduling LS-step launch (64 slots as fibres over the cores; 4 slots for 9 seeds: a finished restart's slot takes the next
    # seed, whose move stream the workgroup then loads under a new id).  Every restart's best solution, LS-iteration count and
    # candidate count equal its single ls_solver run on the launch-per-call route (LS_SCHED_SERVER=0: another code path).
    import ls_amd
    seeds = [str(42 + r) for r in range(n_seeds)]
    monkeypatch.setenv("LS_SCHED_SERVER", "0")
    with ls_amd.Context(0) as c0:
        single = [_single_sched_run(c0, 31, 0, 7, sd, SCHED_EXAMPLE, None, rounds) for sd in seeds]
        assert c0.counters()["server_requests"] == 0
    monkeypatch.setenv("LS_SCHED_SERVER", "1")
    with ls_amd.Context(0) as c:
        m = ls_amd.MultiSolver.scheduling(c, 31, 0, 7, seeds, SCHED_EXAMPLE, n_slots=n_slots)
        m.run(max_rounds_per_restart=rounds)
        cnt = m.counters()
        for r in range(n_seeds):
            st = m.restart_stats(r)
            assert (m.best(r), st["ls_iterations"], st["neighbours_scored"], st["rounds"]) == single[r], (r, seeds[r])
        tot = m.stats()
        assert tot["ls_iterations"] == sum(s[1] for s in single)
        # one request per LS iteration + one per score() (initial solution, each round's perturbed start); ONE launch unless it idled out
        assert cnt["server_requests"] >= tot["ls_iterations"] + n_seeds * rounds
        assert cnt["completion_word_fallbacks"] == 0
        m.close()


def test_multi_sched_one_restart_follows_the_oracle_driver():
    # one restart of a 16-slot run against oracle/ls_oracle.py (the reference's driver restated with the literal pipeline), round for round
    import ls_amd
    import ls_oracle as lo
    seeds = [str(100 + r) for r in range(16)]
    rounds = 5
    with ls_amd.Context(0) as c:
        m = ls_amd.MultiSolver.scheduling(c, 31, 0, 7, seeds, SCHED_EXAMPLE, n_slots=16)
        m.run(max_rounds_per_restart=rounds)
        for r in (0, 7, 15):
            o = lo.IteratedLocalSearch(lo.Scheduling((2022, 5, 9), (2022, 6, 8), 7), dict(SCHED_EXAMPLE), seeds[r])
            for _ in range(rounds):
                o.execute_round()
            assert m.best(r) == o.get_best_solution(), seeds[r]
            assert m.restart_stats(r)["ls_iterations"] == o.local_search.ls_iterations
        m.close()


def test_multi_sched_sub_grids_equal_single_runs(monkeypatch):
    # config 5's shape at test size: 2 000 employees with holidays, window 4 096 (a stream prefix of 5 136 moves: 11 chunks) --
    # beyond one workgroup, so the restarts are the SUB-GRIDS of one resident grid launch (8 restarts side by side; 3 slots for 5
    # seeds: a sub-grid takes the next seed's stream under a new id).  Against single ls_solver runs on the launched pipeline.
    import ls_amd
    rng = np.random.default_rng(55)
    E = 2000
    bits = np.zeros(E, dtype=np.uint64)
    for e in range(E):
        for d in rng.choice(31, size=int(rng.integers(0, 4)), replace=False):
            bits[e] |= np.uint64(1) << np.uint64(int(d))
    p = dict(SCHED_EXAMPLE, window_size=4096)
    rounds = 3
    for n_seeds, n_slots in ((8, 8), (5, 3)):
        seeds = [str(7 + r) for r in range(n_seeds)]
        monkeypatch.setenv("LS_SCHED_SERVER", "0")
        with ls_amd.Context(0) as c0:
            single = [_single_sched_run(c0, 31, 0, E, sd, p, bits, rounds) for sd in seeds]
        monkeypatch.setenv("LS_SCHED_SERVER", "1")
        with ls_amd.Context(0) as c:
            m = ls_amd.MultiSolver.scheduling(c, 31, 0, E, seeds, p, holiday_bits=bits, n_slots=n_slots)
            m.run(max_rounds_per_restart=rounds)
            cnt = m.counters()
            for r in range(n_seeds):
                st = m.restart_stats(r)
                assert (m.best(r), st["ls_iterations"], st["neighbours_scored"], st["rounds"]) == single[r], (n_slots, seeds[r])
            assert cnt["server_requests"] >= m.stats()["ls_iterations"] and cnt["completion_word_fallbacks"] == 0
            m.close()


@pytest.mark.parametrize("E", [26_000, 20_000])
def test_sched_exhaustive_grid_many_chunks_per_workgroup_day_by_day(E):
    # 26 000 employees: 832 000 exhaustive candidates = 1 625 chunks, seven per workgroup (the eight-chunk instantiation), and
    # too many employees for the LDS look-up table (the scorers compare day by day).  Windows inside the first workgroups.
    # 20 000 employees: the day-set table fits the LDS, the copy of the holiday words behind it does not (global gathers).
    import ls_amd
    rng = np.random.default_rng(88)
    hol = {int(e): [int(rng.integers(0, 31))] for e in rng.choice(E, size=3000, replace=False)}
    inst = orc.Sched((2022, 5, 9), (2022, 6, 8), hol)
    D, S = inst.n_scored_days, inst.n_scored_days + 1
    bits = np.zeros(E, dtype=np.uint64)
    for e, ds in hol.items():
        for d in ds:
            bits[e] |= np.uint64(1) << np.uint64(int(d))
    with ls_amd.Context(0) as c:
        c.sched_instance_upload(D, S, inst.start_weekday, E, bits)
        state = rng.integers(0, E, size=S)
        moves = c.sched_exhaustive_moves(state[None, :])[0]
        for w in (3000, 3584 + 1, 7 * 512 * 3 + 17, 40_000):
            r = c.sched_eval_exhaustive(state[None, :], w)[0]
            exp = inst.neighbourhood(state, moves[:w + 64], w)
            assert (r["n_scored"], r["best_index"], r["best_hard"], r["best_soft"]) == \
                   (exp["n_scored"], exp["best_index"], exp["best_hard"], exp["best_soft"]), w
        assert c.counters()["server_requests"] >= 4


def test_more_slots_than_resident_workgroups_fall_back_to_launches(monkeypatch):
    # every workgroup of the N-queens server must be resident for the life of the launch (1024 threads, 147 KB of LDS at
    # n = 8192: one per CU), so its slot count is bounded by what the CUs hold (nq_server_max_slots: 240 of 256).  256 slots do not
    # fit: the run takes one fused launch per LS iteration instead (no starved slot, no 10 s error), with the same results.
    import ls_amd
    n, rounds = 8192, 1
    seeds = [str(900 + r) for r in range(256)]
    p = dict(local_search_max_iterations=10_000, window_size=5 * n, best_solutions_capacity=32, all_solutions_capacity=100_000,
             all_solution_iteration_expiry=10_000, iterated_local_search_max_iterations=10_000, max_allow_no_improvement_for=5)
    with ls_amd.Context(0) as c:
        if c.device_info()["n_cus"] > 272:
            pytest.skip("this device holds 256 resident workgroups of the server")
        m = ls_amd.MultiSolver(c, n, seeds, p, n_slots=256, mode="free")
        m.run(max_rounds_per_restart=rounds)
        cnt = m.counters()
        # (a completion-word fallback = a launch that took > 2 ms and was waited for on the stream instead: benign, and with 256 slots'
        # launches queued on one device and 16 host cores a few happen on a busy host)
        assert cnt["server_requests"] == 0 and cnt["completion_word_fallbacks"] < 16
        got = {r: (m.best(r), m.restart_stats(r)["ls_iterations"]) for r in (0, 100, 255)}
        m.close()
        for r, g in got.items():
            s = ls_amd.Solver.nqueens(c, n, seeds[r], p)
            for _ in range(rounds):
                s.execute_round()
            assert (s.best(), s.stats()["ls_iterations"]) == g, seeds[r]
            s.close()


def test_multi_sched_config5_size_sub_grids_equal_single_runs(monkeypatch):
    # BASELINE config 5 at its real size on ONE GPU: 10 000 employees x 31 days, holidays U{0..3}, window 65 536 (a stream prefix of
    # 73 744 moves = 145 chunks), all 8 restarts side by side as the sub-grids of one resident launch (8 x 29 workgroups x 5 chunks).
    # Every restart's best solution, LS-iteration and candidate counts equal its single ls_solver run on the launched pipeline.
    import ls_amd
    rng = np.random.default_rng(5)
    E = 10_000
    bits = np.zeros(E, dtype=np.uint64)
    for e in range(E):
        for d in rng.choice(31, size=int(rng.integers(0, 4)), replace=False):
            bits[e] |= np.uint64(1) << np.uint64(int(d))
    p = dict(SCHED_EXAMPLE, window_size=65_536)
    rounds = 12
    seeds = [str(42 + r) for r in range(8)]
    monkeypatch.setenv("LS_SCHED_SERVER", "0")
    with ls_amd.Context(0) as c0:
        single = [_single_sched_run(c0, 31, 0, E, sd, p, bits, rounds) for sd in seeds]
        assert c0.counters()["server_requests"] == 0
    monkeypatch.setenv("LS_SCHED_SERVER", "1")
    with ls_amd.Context(0) as c:
        m = ls_amd.MultiSolver.scheduling(c, 31, 0, E, seeds, p, holiday_bits=bits, n_slots=8)
        m.run(max_rounds_per_restart=rounds)
        cnt = m.counters()
        for r in range(8):
            st = m.restart_stats(r)
            assert (m.best(r), st["ls_iterations"], st["neighbours_scored"], st["rounds"]) == single[r], seeds[r]
        assert cnt["server_requests"] >= m.stats()["ls_iterations"] and cnt["completion_word_fallbacks"] == 0
        # a second run of the same object reuses the launch object and the slot contexts
        m.run(max_rounds_per_restart=rounds)
        assert all((m.best(r), m.restart_stats(r)["ls_iterations"]) == (single[r][0], single[r][1]) for r in range(8))
        m.close()
