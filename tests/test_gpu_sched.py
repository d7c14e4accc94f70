"""GPU parity tests (through the C ABI) for the employee-scheduling path: HIP
kernels vs the CPU oracle, bit-exact on (hard, soft) per candidate, on the eight
constraint terms, on n_scored and on the selected candidate.  The reference
holds no fixture for these scores (parity unpinned by the reference); the oracle
is cross-checked separately in test_oracle_sched.py."""
import datetime as dt
import json
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


# every test runs on the default DELTA scorer (one lane per candidate; long move lists in two launches) and on the
# wavefront-per-candidate kernels
@pytest.fixture(scope="module", params=["delta", "wavefront"])
def ctx(request):
    import ls_amd
    knobs = {"LS_SCHED_DELTA": "0" if request.param == "wavefront" else "1"}
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        c = ls_amd.Context(0)   # the knobs are read when the context is created
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    yield c
    c.close()


def _upload(ctx, start, end, n_emp, hol=None, n_stored=None):
    inst = orc.Sched(start, end, hol or {})
    D = inst.n_scored_days
    S = D + 1 if n_stored is None else n_stored
    bits = np.zeros(n_emp, dtype=np.uint64)
    for e, days in (hol or {}).items():
        for d in days:
            bits[e] |= np.uint64(1) << np.uint64(d)
    ctx.sched_instance_upload(D, S, inst.start_weekday, n_emp, bits)
    return inst, D, S


def _random_moves(rng, n, D, n_emp, state):
    mv = np.zeros((n, 3), dtype=np.int64)
    for i in range(n):
        if rng.integers(0, 5) == 0 or D < 2:              # ChangeDay : SwapDays = 1 : 4 (lib.rs:435)
            mv[i] = (0, rng.integers(0, D), rng.integers(0, n_emp))
        else:
            a, b = rng.choice(D, size=2, replace=False)
            mv[i] = (1, a, b)
    return mv


def test_survey_kats(ctx, golden_dir):
    with open(os.path.join(golden_dir, "sched_kats.json")) as f:
        g = json.load(f)
    for case in g["cases"]:
        hol = {int(k): v for k, v in case["holidays"].items()}
        _upload(ctx, g["start"], g["end"], 7, hol, g["n_stored"])
        d = np.arange(g["n_stored"])
        e = {"const0": d * 0, "mod7": d % 7, "mod2": d % 2, "mod3": d % 3}[case["pattern"]]
        hard, soft, terms = ctx.sched_score(e, want_terms=True)
        assert (hard[0], soft[0]) == (case["hard"], case["soft"]), case
        assert terms[0].tolist() == case["terms"], case


@pytest.mark.parametrize("start,end,n_emp", [
    ((2022, 5, 9), (2022, 6, 8), 7),       # config 3: the reference example (main.rs:11-21)
    ((2022, 5, 9), (2022, 6, 8), 2),
    ((2022, 5, 9), (2022, 6, 8), 10000),   # config 5 employee count
    ((2023, 12, 30), (2024, 2, 29), 5),    # 62 scored / 63 stored days, Saturday start
    ((2024, 1, 6), (2024, 3, 9), 4),       # 64 scored days (the maximum), stored == scored
    ((2024, 1, 3), (2024, 1, 10), 3),      # 8 days: no 9-/14-window
    ((2024, 1, 3), (2024, 1, 3), 3),       # one day
])
def test_score_vs_oracle(ctx, start, end, n_emp):
    rng = np.random.default_rng(abs(hash((start, end, n_emp))) % 2 ** 32)
    d = (dt.date(*end) - dt.date(*start)).days + 1
    hol = {}
    for emp in range(min(n_emp, 50)):
        k = int(rng.integers(0, 4))
        if k:
            hol[emp] = rng.choice(d, size=min(k, d), replace=False).tolist()
    inst, D, S = _upload(ctx, start, end, n_emp, hol, n_stored=min(d + 1, 64))
    pool = min(n_emp, 50) if n_emp > 100 else n_emp      # force repeats even with 10k employees
    states = rng.integers(0, pool, size=(300, S))
    states[:50] = rng.integers(0, min(3, n_emp), size=(50, S))
    hard, soft, terms = ctx.sched_score(states, want_terms=True)
    for i in range(len(states)):
        h, s, t = inst.score(states[i], terms=True)
        assert terms[i].tolist() == t.tolist(), (i, states[i])
        assert (hard[i], soft[i]) == (h, s)


@pytest.mark.parametrize("n_emp,window", [(7, 100), (7, 10 ** 9), (2, 37), (10000, 4096)])
def test_neighbourhood_random_stream(ctx, n_emp, window):
    start, end = (2022, 5, 9), (2022, 6, 8)
    rng = np.random.default_rng(n_emp + window % 1000)
    hol = {e: rng.choice(31, size=2, replace=False).tolist() for e in range(0, min(n_emp, 40), 3)}
    inst, D, S = _upload(ctx, start, end, n_emp, hol)
    for trial in range(3):
        state = rng.integers(0, min(n_emp, 12), size=S)
        moves = _random_moves(rng, 700, D, min(n_emp, 12), state)
        exp = inst.neighbourhood(state, moves, window)
        got, hard, soft = ctx.sched_eval(state, moves, window, want_scores=True)
        assert got["n_scored"] == exp["n_scored"]
        assert got["best_index"] == exp["best_index"]
        assert (got["best_hard"], got["best_soft"]) == (exp["best_hard"], exp["best_soft"])
        assert np.array_equal(hard, exp["hard"]) and np.array_equal(soft, exp["soft"])


def test_neighbourhood_exhaustive_proposer(ctx):
    # ScheduleMoveProposer order (lib.rs:526-550): 32 stored days x E employees, one no-op per day
    start, end, n_emp = (2022, 5, 9), (2022, 6, 8), 40
    rng = np.random.default_rng(3)
    inst, D, S = _upload(ctx, start, end, n_emp, {1: [4], 7: [0, 30]})
    state = rng.integers(0, n_emp, size=S)
    moves = orc.Sched.exhaustive_moves(state, range(n_emp))
    assert len(moves) == S * n_emp
    exp = inst.neighbourhood(state, moves, 10 ** 9)
    got, hard, soft = ctx.sched_eval(state, moves, 10 ** 9, want_scores=True)
    assert got["n_scored"] == exp["n_scored"] == S * (n_emp - 1)
    assert got["best_index"] == exp["best_index"]
    assert np.array_equal(hard, exp["hard"]) and np.array_equal(soft, exp["soft"])


def test_batched_restarts_ragged(ctx):
    start, end, n_emp = (2022, 5, 9), (2022, 6, 8), 7
    rng = np.random.default_rng(17)
    inst, D, S = _upload(ctx, start, end, n_emp)
    R, max_moves = 6, 250
    states = rng.integers(0, n_emp, size=(R, S))
    moves = np.stack([_random_moves(rng, max_moves, D, n_emp, states[r]) for r in range(R)])
    n_moves = np.array([0, 1, 100, 250, 17, 249])
    out, hard, soft = ctx.sched_eval_batch(states, moves, n_moves, 100, want_scores=True)
    for r in range(R):
        exp = inst.neighbourhood(states[r], moves[r, :n_moves[r]], 100)
        assert out[r]["n_scored"] == exp["n_scored"]
        assert out[r]["best_index"] == exp["best_index"]
        if exp["best_index"] >= 0:
            assert (out[r]["best_hard"], out[r]["best_soft"]) == (exp["best_hard"], exp["best_soft"])
        assert np.array_equal(hard[r, :n_moves[r]], exp["hard"])
        assert (hard[r, n_moves[r]:] == orc.SKIPPED).all()


def test_tie_break_and_first_of_equals(ctx):
    # identical moves proposed twice: first position wins; swap(a,b) == swap(b,a)
    inst, D, S = _upload(ctx, (2022, 5, 9), (2022, 6, 8), 7)
    state = (np.arange(S) % 7).astype(np.int64)
    moves = [(1, 3, 9), (1, 9, 3), (0, 5, 0), (0, 5, 0), (1, 3, 9), (0, 31, 2), (0, 31, 2)]
    exp = inst.neighbourhood(state, moves, 100)
    got, hard, soft = ctx.sched_eval(state, moves, 100, want_scores=True)
    assert got["best_index"] == exp["best_index"]
    assert np.array_equal(hard, exp["hard"]) and np.array_equal(soft, exp["soft"])
    # stored-but-unscored tail entry (index 31) changes the vector order, not the score
    assert hard[5] == hard[6] and exp["hard"][5] != orc.SKIPPED


def test_errors(ctx):
    import ls_amd
    c2 = ls_amd.Context(0)
    with pytest.raises(ls_amd.LsError) as ei:
        c2.sched_score(np.zeros(32, dtype=np.int32))
    assert ei.value.code == ls_amd._lib.LS_ERR_STATE
    with pytest.raises(ls_amd.LsError) as ei:
        c2.sched_instance_upload(70, 71, 0, 3)
    assert ei.value.code == ls_amd._lib.LS_ERR_UNSUPPORTED
    c2.sched_instance_upload(31, 32, 0, 3)
    with pytest.raises(ls_amd.LsError):
        c2.sched_score(np.full(32, 3, dtype=np.int32))       # employee index out of range
    with pytest.raises(ls_amd.LsError):
        c2.sched_eval(np.zeros(32, dtype=np.int32), [(3, 0, 0)], 10)   # unknown move type (2 = LS_SCHED_MOVE_SKIP is valid)
    with pytest.raises(ls_amd.LsError):
        c2.sched_instance_upload(31, 32, 0, 1, np.array([1 << 40], dtype=np.uint64))  # holiday outside range
    c2.close()


def test_resident_move_list_matches_host_list(ctx):
    inst, D, S = _upload(ctx, (2022, 5, 9), (2022, 6, 8), 7)
    rng = np.random.default_rng(8)
    state = rng.integers(0, 7, size=S)
    moves = _random_moves(rng, 900, D, 7, state)
    ctx.sched_moves_upload(moves)
    for n_moves, window in ((900, 100), (137, 10 ** 9), (0, 5), (900, 0)):
        a = ctx.sched_eval_resident(state, n_moves, window)
        b, _, _ = ctx.sched_eval(state, moves[:n_moves], window) if n_moves else (dict(n_scored=0, best_index=-1, best_hard=0, best_soft=0), None, None)
        assert a == b
    import ls_amd
    with pytest.raises(ls_amd.LsError):
        ctx.sched_eval_resident(state, 901, 10)


@pytest.mark.parametrize("server", ["1", "0"])
@pytest.mark.parametrize("delta", ["1", "0"])
def test_small_launch_fused_kernel(monkeypatch, delta, server):
    # move lists of <= 512 entries run locate + eval + reduce as ONE block per restart; the resident entry goes to the
    # resident workgroup (LS_SCHED_SERVER=1, default) or hands the launched kernels its inputs in the kernel arguments (=0)
    import ls_amd
    monkeypatch.setenv("LS_SCHED_DELTA", delta)
    monkeypatch.setenv("LS_SCHED_SERVER", server)
    rng = np.random.default_rng(40 + int(delta) * 2 + int(server))
    with ls_amd.Context(0) as c:
        for start, end, n_emp in (((2022, 5, 9), (2022, 6, 8), 7),      # half-wavefront kernel (32 stored days)
                                  ((2024, 1, 3), (2024, 1, 10), 3),     # 8 days
                                  ((2023, 12, 30), (2024, 2, 29), 5)):  # 63 stored days: full-wavefront kernel
            hol = {0: [0, 1], 2: [3]}
            inst, D, S = _upload(c, start, end, n_emp, hol)
            for n_moves, window in ((150, 100), (512, 100), (512, 10 ** 9), (64, 64), (65, 1), (7, 0), (300, 299), (513, 100)):
                state = rng.integers(0, n_emp, size=S)
                moves = _random_moves(rng, n_moves, D, n_emp, state)
                exp = inst.neighbourhood(state, moves, window)
                got, hard, soft = c.sched_eval(state, moves, window, want_scores=True)
                assert (got["n_scored"], got["best_index"]) == (exp["n_scored"], exp["best_index"]), (n_moves, window)
                if exp["best_index"] >= 0:
                    assert (got["best_hard"], got["best_soft"]) == (exp["best_hard"], exp["best_soft"])
                assert np.array_equal(hard, exp["hard"]) and np.array_equal(soft, exp["soft"])
                c.sched_moves_upload(moves)
                res = c.sched_eval_resident(state, n_moves, window)
                assert res == {k: got[k] for k in res}
            # batched, ragged
            R, mm = 5, 200
            states = rng.integers(0, n_emp, size=(R, S))
            moves = np.stack([_random_moves(rng, mm, D, n_emp, states[r]) for r in range(R)])
            n_moves = np.array([0, 1, 200, 63, 129])
            out, _, _ = c.sched_eval_batch(states, moves, n_moves, 50)
            for r in range(R):
                exp = inst.neighbourhood(states[r], moves[r, :n_moves[r]], 50)
                assert (out[r]["n_scored"], out[r]["best_index"]) == (exp["n_scored"], exp["best_index"])


def test_exhaustive_proposer_entries(ctx):
    # ScheduleMoveProposer (lib.rs:493-559) generated on the device: same list as the oracle's restatement,
    # and the generate + score + argmin entry agrees with evaluating that list
    rng = np.random.default_rng(91)
    for start, end, n_emp, hol in (((2022, 5, 9), (2022, 6, 8), 7, {0: [0, 1], 3: [3]}),
                                   ((2022, 5, 9), (2022, 6, 8), 1, {}),           # one employee: every move is a no-op
                                   ((2024, 1, 3), (2024, 1, 10), 40, {1: [4]}),
                                   ((2023, 12, 30), (2024, 2, 29), 300, {})):
        inst, D, S = _upload(ctx, start, end, n_emp, hol)
        R = 3
        states = rng.integers(0, n_emp, size=(R, S))
        mv = ctx.sched_exhaustive_moves(states)
        assert mv.shape == (R, S * n_emp, 3)
        for window in (10 ** 9, 37, 0):
            got = ctx.sched_eval_exhaustive(states, window)
            for r in range(R):
                exp_moves = orc.Sched.exhaustive_moves(states[r], range(n_emp))
                assert np.array_equal(mv[r], exp_moves)
                exp = inst.neighbourhood(states[r], exp_moves, window)
                assert (got[r]["n_scored"], got[r]["best_index"]) == (exp["n_scored"], exp["best_index"]), (n_emp, window, r)
                if exp["best_index"] >= 0:
                    assert (got[r]["best_hard"], got[r]["best_soft"]) == (exp["best_hard"], exp["best_soft"])
                if window == 10 ** 9:
                    assert exp["n_scored"] == S * (n_emp - 1)


# ---------------------------------------------------------------------------------------------
# Config 5 at its launch size: 10 000 employees x 31 scored / 32 stored days, holidays U{0..3} per
# employee.  The 320 000-move exhaustive list is the only workload that takes the second trip of the 256-chunk
# prefix loop in sched_locate_kernel (313 chunks of 1024 moves).

def _config5(ctx, seed=2022):
    rng = np.random.default_rng(seed)
    E, D = 10_000, 31
    hol = {e: rng.choice(D, size=int(rng.integers(0, 4)), replace=False).tolist() for e in range(E)}
    inst, D2, S = _upload(ctx, (2022, 5, 9), (2022, 6, 8), E, hol)
    assert (D2, S) == (31, 32)
    return rng, inst, E, D, S


@pytest.mark.parametrize("window", [10 ** 9, 300_000, 270_000, 262_144 + 1, 1024 * 256, 1])
def test_config5_exhaustive_at_launch_size(ctx, window):
    # window 270 000 stops inside chunk 263 (second trip of the prefix loop), 262 145 in its first chunk,
    # 262 144 = 256 * 1024 at the seam (one tabu move per day shifts the cut a few positions past the chunk edge)
    rng, inst, E, D, S = _config5(ctx)
    states = np.stack([rng.integers(0, E, size=S), rng.integers(0, 50, size=S)])
    got = ctx.sched_eval_exhaustive(states, window)
    for r in range(2):
        moves = orc.Sched.exhaustive_moves(states[r], range(E))
        assert len(moves) == 320_000
        exp = inst.neighbourhood(states[r], moves, window)
        assert got[r]["n_scored"] == exp["n_scored"] == min(window, S * (E - 1))
        assert got[r]["best_index"] == exp["best_index"]
        assert (got[r]["best_hard"], got[r]["best_soft"]) == (exp["best_hard"], exp["best_soft"])
        if r == 1 and window in (10 ** 9, 270_000):
            # the same list through the caller-supplied-moves entry (ls_sched_eval) must agree, incl. every score
            g1, hard, soft = ctx.sched_eval(states[1], moves, window, want_scores=True)
            assert (g1["n_scored"], g1["best_index"]) == (exp["n_scored"], exp["best_index"])
            assert np.array_equal(hard, exp["hard"]) and np.array_equal(soft, exp["soft"])


@pytest.mark.parametrize("window", [65_536, 10 ** 9])
def test_config5_random_stream_at_launch_size(ctx, window):
    # the 70 000-move random stream of bench.py's sched10k_random_stream_window_65536 line
    rng, inst, E, D, S = _config5(ctx, seed=77)
    for trial in range(2):
        state = rng.integers(0, E if trial == 0 else 30, size=S)
        moves = _random_moves(rng, 70_000, D, E if trial == 0 else 30, state)
        exp = inst.neighbourhood(state, moves, window)
        got, hard, soft = ctx.sched_eval(state, moves, window, want_scores=True)
        assert got["n_scored"] == exp["n_scored"]
        assert got["best_index"] == exp["best_index"]
        assert (got["best_hard"], got["best_soft"]) == (exp["best_hard"], exp["best_soft"])
        assert np.array_equal(hard, exp["hard"]) and np.array_equal(soft, exp["soft"])
        # production shape (no score dump, resident list) agrees
        ctx.sched_moves_upload(moves)
        res = ctx.sched_eval_resident(state, len(moves), window)
        assert (res["n_scored"], res["best_index"], res["best_hard"], res["best_soft"]) == \
               (exp["n_scored"], exp["best_index"], exp["best_hard"], exp["best_soft"])


@pytest.mark.parametrize("E", [700, 13_000])
def test_two_launch_pipeline_with_and_without_employee_table(monkeypatch, E):
    # sched_eval_all_kernel with the employee -> scored-days look-up table in LDS (700 employees: 2.8 KB per block) and with the
    # day-by-day compare loops it falls back to when the table does not fit beside three blocks per CU (13 000 employees: 52 KB);
    # batches of 64 candidates per wavefront chosen from the grid size.  E x 32 exhaustive candidates.
    import ls_amd
    monkeypatch.setenv("LS_SCHED_SERVER", "0")                        # the launched pipeline, not the resident grid
    rng = np.random.default_rng(31)
    hol = {int(e): list(rng.choice(31, size=2, replace=False)) for e in rng.choice(E, size=200, replace=False)}
    with ls_amd.Context(0) as c:
        inst, D, S = _upload(c, (2022, 5, 9), (2022, 6, 8), E, hol)
        for trial in range(2 if E < 1000 else 1):
            states = np.stack([rng.integers(0, E, size=S), rng.integers(0, 9, size=S), rng.integers(0, E, size=S)])
            for window in (10 ** 9, 22_000, 1537, 1536, 511, 1) if E < 1000 else (10 ** 9, 1537):
                got = c.sched_eval_exhaustive(states, window)
                for r in range(len(states)):
                    moves = orc.Sched.exhaustive_moves(states[r], range(E))
                    exp = inst.neighbourhood(states[r], moves, window)
                    assert (got[r]["n_scored"], got[r]["best_index"]) == (exp["n_scored"], exp["best_index"]), (trial, window, r)
                    assert (got[r]["best_hard"], got[r]["best_soft"]) == (exp["best_hard"], exp["best_soft"])
            state = states[1]
            moves = _random_moves(rng, 5000, D, 12, state)            # a caller's stream with many SwapDays and no-ops
            c.sched_moves_upload(moves)
            for window in (10 ** 9, 3000, 1025):
                exp = inst.neighbourhood(state, moves, window)
                res = c.sched_eval_resident(state, len(moves), window)
                assert (res["n_scored"], res["best_index"], res["best_hard"], res["best_soft"]) == \
                       (exp["n_scored"], exp["best_index"], exp["best_hard"], exp["best_soft"])
        assert c.counters()["server_requests"] == 0


def test_skip_moves_for_a_general_tabu_set(ctx):
    # LS_SCHED_MOVE_SKIP: how a caller whose History keeps more than the current solution tabu
    # (all_solution_iteration_expiry == 0, local_search.rs:155-199) tells the GPU that a move leads to a tabu solution:
    # neither scored nor counted.  Oracle: the literal pipeline with a whole-vector tabu set.
    start, end, n_emp = (2022, 5, 9), (2022, 6, 8), 7
    rng = np.random.default_rng(8)
    inst, D, S = _upload(ctx, start, end, n_emp, {2: [5, 6]})
    for trial in range(5):
        state = rng.integers(0, n_emp, size=S)
        moves = _random_moves(rng, 400, D, n_emp, state)
        cands = []
        for t, a, b in moves:
            c = list(state)
            if t == 0:
                c[a] = b
            else:
                c[a], c[b] = state[b], state[a]
            cands.append(tuple(int(x) for x in c))
        tabu = [tuple(int(x) for x in state)] + [cands[i] for i in rng.choice(len(cands), size=25, replace=False)]
        marked = moves.copy()
        for i, c in enumerate(cands):
            if c != tabu[0] and c in set(tabu[1:]):
                marked[i] = (2, 0, 0)
        for window in (100, 10 ** 9, 7):
            exp = inst.neighbourhood_tabu(state, moves, window, tabu)
            got, hard, soft = ctx.sched_eval(state, marked, window, want_scores=True)
            assert (got["n_scored"], got["best_index"]) == (exp["n_scored"], exp["best_index"]), (trial, window)
            assert (got["best_hard"], got["best_soft"]) == (exp["best_hard"], exp["best_soft"])
            assert np.array_equal(hard, exp["hard"]) and np.array_equal(soft, exp["soft"])


@pytest.mark.parametrize("days,n_emp", [(31, 7), (31, 2), (63, 5)])
def test_window_cuts_of_the_two_launch_pipeline(ctx, days, n_emp):
    # production shape (no score dump), move lists beyond one block: sched_eval_all_kernel scores 512-move chunks,
    # sched_finish_kernel finds the chunk in which take(window) stops and re-scores it up to the window.  Windows
    # are placed on, just before and just after every chunk's cumulative count of real candidates.
    start = (2022, 5, 9)
    end_d = dt.date(*start) + dt.timedelta(days=days - 1)
    rng = np.random.default_rng(days * 100 + n_emp)
    inst, D, S = _upload(ctx, start, (end_d.year, end_d.month, end_d.day), n_emp, {0: [0, 3], 1: [5]})
    R, max_moves = 3, 1700
    states = rng.integers(0, n_emp, size=(R, S))
    moves = np.stack([_random_moves(rng, max_moves, D, n_emp, states[r]) for r in range(R)])
    n_moves = np.array([1700, 1025, 513])
    full = [inst.neighbourhood(states[r], moves[r, :n_moves[r]], 10 ** 9) for r in range(R)]
    real = [(full[r]["hard"] != orc.SKIPPED) for r in range(R)]
    windows = {0, 1, 2, 10 ** 9}
    for r in range(R):
        cum = np.cumsum(real[r])
        for edge in range(512, n_moves[r] + 512, 512):
            c = int(cum[min(edge, n_moves[r]) - 1])
            windows.update({c - 1, c, c + 1})
        windows.update({int(cum[-1]) - 1, int(cum[-1]), int(cum[-1]) + 1})
    for window in sorted(w for w in windows if w >= 0):
        out, _, _ = ctx.sched_eval_batch(states, moves, n_moves, window)
        for r in range(R):
            exp = inst.neighbourhood(states[r], moves[r, :n_moves[r]], window)
            assert (out[r]["n_scored"], out[r]["best_index"]) == (exp["n_scored"], exp["best_index"]), (window, r)
            if exp["best_index"] >= 0:
                assert (out[r]["best_hard"], out[r]["best_soft"]) == (exp["best_hard"], exp["best_soft"]), (window, r)


@pytest.mark.parametrize("start,days,stored,n_emp", [
    ((2022, 5, 9), 1, 2, 3), ((2022, 5, 9), 2, 3, 1), ((2022, 5, 9), 6, 7, 2), ((2022, 5, 9), 7, 8, 3),
    ((2022, 5, 14), 8, 9, 2), ((2022, 5, 15), 13, 14, 3), ((2022, 5, 9), 14, 15, 4), ((2022, 5, 13), 15, 16, 2),
    ((2022, 5, 9), 32, 32, 3), ((2022, 5, 10), 32, 33, 5), ((2022, 5, 14), 33, 34, 40), ((2022, 5, 11), 64, 64, 6),
])
def test_neighbourhood_odd_shapes(ctx, start, days, stored, n_emp):
    # short and long date ranges, stored == scored and stored == scored + 1, one employee, more employees than days,
    # moves that touch the stored-but-unscored last entry: per-candidate scores (score-dump chain), the production
    # pipelines (fused block / two launches, no dump) and the exhaustive proposer, all against the oracle
    end_d = dt.date(*start) + dt.timedelta(days=days - 1)
    rng = np.random.default_rng(days * 1000 + stored * 10 + n_emp)
    hol = {e: [int(d) for d in range(days) if rng.random() < 0.15] for e in range(min(n_emp, 8))}
    inst, D, S = _upload(ctx, start, (end_d.year, end_d.month, end_d.day), n_emp, hol, n_stored=stored)
    assert (D, S) == (days, stored)
    for trial, n_moves in enumerate((37, 512, 700, 1500)):
        state = rng.integers(0, n_emp if trial % 2 == 0 else min(n_emp, 2), size=S)
        mv = np.zeros((n_moves, 3), dtype=np.int64)
        for i in range(n_moves):
            if rng.integers(0, 2) == 0 or S < 2:
                mv[i] = (0, rng.integers(0, S), rng.integers(0, n_emp))        # may edit the unscored entry
            else:
                a, b = rng.choice(S, size=2, replace=False)
                mv[i] = (1, a, b)
        for window in (10 ** 9, int(rng.integers(1, n_moves + 1))):
            exp = inst.neighbourhood(state, mv, window)
            got, hard, soft = ctx.sched_eval(state, mv, window, want_scores=True)
            assert np.array_equal(hard, exp["hard"]) and np.array_equal(soft, exp["soft"]), (n_moves, window)
            assert (got["n_scored"], got["best_index"]) == (exp["n_scored"], exp["best_index"])
            got2, _, _ = ctx.sched_eval(state, mv, window)
            assert got2 == got, (n_moves, window)
        ex = orc.Sched.exhaustive_moves(state, range(n_emp))
        for window in (10 ** 9, max(1, len(ex) // 3)):
            exp = inst.neighbourhood(state, ex, window)
            got = ctx.sched_eval_exhaustive(state, window)[0]
            assert (got["n_scored"], got["best_index"]) == (exp["n_scored"], exp["best_index"]), window
            if exp["best_index"] >= 0:
                assert (got["best_hard"], got["best_soft"]) == (exp["best_hard"], exp["best_soft"])
