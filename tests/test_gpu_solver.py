"""GPU end-to-end: the product's host driver (C++ mirror of IteratedLocalSearch / LocalSearch /
plug-ins, every score and neighbourhood on the GPU) against the oracle's independent Python
restatement with the literal CPU neighbourhood pipeline: same seed => identical trajectory
(current and best score after every round) and identical final solutions."""
import os

import numpy as np
import pytest

import ls_oracle as lo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import ls_amd
    c = ls_amd.Context(0)
    yield c
    c.close()


def _nq_params(n, expiry=10_000):
    p = dict(lo.NQUEENS_MAIN_PARAMS)
    p["window_size"] = 5 * n                       # examples/nqueens/src/main.rs:130
    p["all_solution_iteration_expiry"] = expiry
    return p


def _trace_product(solver, rounds):
    tr = []
    for _ in range(rounds):
        if solver.is_finished():
            break
        solver.execute_round()
        b = solver.best()
        tr.append((solver.current()[0], None if b is None else b[0]))
    return tr


def _trace_oracle(ils, rounds):
    tr = []
    for _ in range(rounds):
        if ils.is_finished():
            break
        ils.execute_round()
        b = ils.get_best_solution()
        tr.append((ils.current[0], None if b is None else b[0]))
    return tr


@pytest.mark.parametrize("algo", ["scan", "table"])
def test_nq8_config1_matches_oracle_and_solves(ctx, algo):
    # config 1 (examples/nqueens defaults, n = 8) incl. the reference's `repeatable` expectations
    import ls_amd
    ctx.set_algo(algo)
    for seed in map(str, range(42, 50)):
        p = _nq_params(8, expiry=1_000)
        s = ls_amd.Solver.nqueens(ctx, 8, seed, p)
        o = lo.IteratedLocalSearch(lo.NQueens(8), p, seed)
        assert s.current() == o.current                               # same initial permutation and score
        t1, t2 = _trace_product(s, 40), _trace_oracle(o, 40)
        assert t1 == t2, seed
        best = s.best()
        assert best == o.get_best_solution()
        assert best[0] == 0, seed                                     # main.rs:194-198
        s.close()
    ctx.set_algo("scan")


@pytest.mark.parametrize("n,rounds,seed", [(20, 12, "42"), (64, 6, "7"), (200, 3, "abc"), (1024, 2, "42")])
def test_nqueens_trajectory_matches_oracle(ctx, n, rounds, seed):
    import ls_amd
    p = _nq_params(n)
    s = ls_amd.Solver.nqueens(ctx, n, seed, p)
    o = lo.IteratedLocalSearch(lo.NQueens(n), p, seed)
    assert s.current() == o.current
    assert _trace_product(s, rounds) == _trace_oracle(o, rounds)
    assert s.current() == o.current
    assert s.best() == o.get_best_solution()
    assert s.stats()["ls_iterations"] == o.local_search.ls_iterations
    s.close()


def test_sched_example_trajectory_matches_oracle(ctx):
    # config 3: examples/employee-scheduling/src/main.rs:11-31
    import ls_amd
    p = dict(lo.SCHED_MAIN_PARAMS)
    o = lo.IteratedLocalSearch(lo.Scheduling((2022, 5, 9), (2022, 6, 8), 7), p, "42")
    s = ls_amd.Solver.scheduling(ctx, 31, 0, 7, "42", p)
    assert s.current() == o.current
    assert _trace_product(s, 4) == _trace_oracle(o, 4)
    assert s.current() == o.current
    assert s.best() == o.get_best_solution()
    st = s.stats()
    assert st["ls_iterations"] == o.local_search.ls_iterations
    assert st["neighbours_scored"] == 100 * st["ls_iterations"]      # window always fills (infinite stream)
    s.close()


def test_sched_with_holidays_trajectory(ctx):
    import ls_amd
    hol = {0: [0, 1, 14], 3: [3], 6: [30]}
    bits = np.zeros(7, dtype=np.uint64)
    for e, ds in hol.items():
        for d in ds:
            bits[e] |= np.uint64(1) << np.uint64(d)
    p = dict(lo.SCHED_MAIN_PARAMS)
    p["window_size"] = 37
    o = lo.IteratedLocalSearch(lo.Scheduling((2022, 5, 9), (2022, 6, 8), 7, hol), p, "holiday seed")
    s = ls_amd.Solver.scheduling(ctx, 31, 0, 7, "holiday seed", p, holiday_bits=bits)
    assert _trace_product(s, 3) == _trace_oracle(o, 3)
    assert s.best() == o.get_best_solution()
    s.close()


def test_solver_api_edges(ctx):
    import ls_amd
    p = _nq_params(8)
    s = ls_amd.Solver.nqueens(ctx, 8, "42", p)
    assert s.best() is None                                          # best set empty before the first round
    assert s.iteration_info() == dict(current=0, total=10_000)
    assert not s.is_finished()
    s.execute_round()
    assert s.iteration_info()["current"] == 1
    s.close()
    bad = dict(p, best_solutions_capacity=0)                         # the reference would panic (best set of capacity 0)
    with pytest.raises(ls_amd.LsError) as ei:
        ls_amd.Solver.nqueens(ctx, 8, "42", bad)
    assert ei.value.code == ls_amd._lib.LS_ERR_INVALID
    m = ls_amd.MultiSolver(ctx, 8, ["42", "43"], dict(p, all_solution_iteration_expiry=0), n_slots=2, mode="lockstep")
    with pytest.raises(ls_amd.LsError):                              # a growing tabu set needs one-restart launches
        m.run(max_rounds_per_restart=2)
    m.close()
    m = ls_amd.MultiSolver(ctx, 8, ["42", "43"], dict(p, all_solution_iteration_expiry=0), n_slots=2, mode="free")
    m.run(max_rounds_per_restart=3)
    m.close()
    short = dict(p, iterated_local_search_max_iterations=2)
    s = ls_amd.Solver.nqueens(ctx, 30, "1", short)
    for _ in range(2):
        s.execute_round()
    assert s.is_finished()
    s.close()


def test_multi_restart_lockstep_equals_single_runs(ctx):
    # restarts are independent; batching their launches must not change any result
    import ls_amd
    n, rounds = 64, 5
    seeds = [str(s) for s in range(42, 48)]
    p = _nq_params(n)
    singles = []
    for seed in seeds:
        s = ls_amd.Solver.nqueens(ctx, n, seed, p)
        for _ in range(rounds):
            s.execute_round()
            b = s.best()
            if b is not None and b[0] == 0:
                break
        singles.append(s.best())
        s.close()
    m = ls_amd.MultiSolver(ctx, n, seeds, p, n_slots=len(seeds), mode="lockstep")
    m.run(max_rounds_per_restart=rounds)
    assert [m.best(r) for r in range(len(seeds))] == singles
    st = m.stats()
    assert st["gpu_launches"] < 2 * st["ls_iterations"]            # launches are shared between restarts
    m.close()
    m = ls_amd.MultiSolver(ctx, n, seeds, p)                        # the default: free-running slots
    m.run(max_rounds_per_restart=rounds)
    assert [m.best(r) for r in range(len(seeds))] == singles
    m.close()
    # oracle cross-check of one of them
    o = lo.IteratedLocalSearch(lo.NQueens(n), p, seeds[1])
    for _ in range(rounds):
        o.execute_round()
        b = o.get_best_solution()
        if b is not None and b[0] == 0:
            break
    assert o.get_best_solution() == singles[1]


@pytest.mark.parametrize("algo", ["auto", "scan"])      # fused one-launch LS step / multi-launch chain (argmin keys + finalize)
@pytest.mark.parametrize("mode,n_slots", [("lockstep", 4), ("free", 4), ("free", 16), ("lockstep", 1)])
def test_seeds_stream_through_slots(ctx, mode, n_slots, algo):
    # config 4's shape in small: more seeds than slots -- a finished restart's slot takes the next seed
    # (examples/nqueens/src/main.rs:167 seed convention) -- every restart equal to its single-solver run
    import ls_amd
    ctx.set_algo(algo)
    n, rounds = 48, 6
    seeds = [str(s) for s in range(42, 42 + 13)]
    p = _nq_params(n)
    singles, iters = [], 0
    for seed in seeds:
        s = ls_amd.Solver.nqueens(ctx, n, seed, p)
        for _ in range(rounds):
            s.execute_round()
            b = s.best()
            if b is not None and b[0] == 0:
                break
        singles.append(s.best())
        iters += s.stats()["ls_iterations"]
        s.close()
    m = ls_amd.MultiSolver(ctx, n, seeds, p, n_slots=n_slots, mode=mode)
    m.run(max_rounds_per_restart=rounds)
    assert [m.best(r) for r in range(len(seeds))] == singles
    st = m.stats()
    assert st["ls_iterations"] == iters
    m.close()


@pytest.mark.parametrize("algo", ["auto", "scan"])
def test_more_slots_than_cores_share_threads_as_fibres(ctx, algo):
    # free-running slots beyond the host's cores run as cooperative fibres, several restarts per thread, switching in the
    # completion-word wait: every restart must still equal its single-solver run
    import ls_amd
    ctx.set_algo(algo)
    n, rounds = 40, 4
    n_seeds = 53                                           # the GPU box gives a process 16 cores
    seeds = [str(s) for s in range(500, 500 + n_seeds)]
    p = _nq_params(n)
    singles, iters = [], 0
    for seed in seeds:
        s = ls_amd.Solver.nqueens(ctx, n, seed, p)
        for _ in range(rounds):
            s.execute_round()
            b = s.best()
            if b is not None and b[0] == 0:
                break
        singles.append(s.best())
        iters += s.stats()["ls_iterations"]
        s.close()
    m = ls_amd.MultiSolver(ctx, n, seeds, p, n_slots=n_seeds, mode="free")
    m.run(max_rounds_per_restart=rounds)
    assert [m.best(r) for r in range(len(seeds))] == singles
    assert m.stats()["ls_iterations"] == iters
    m.close()


@pytest.mark.parametrize("n,rounds,seed", [(8, 25, "42"), (12, 12, "7"), (30, 5, "abc")])
def test_general_history_expiry_zero_nqueens(ctx, n, rounds, seed):
    # all_solution_iteration_expiry == 0: the reference's tabu set keeps every solution LocalSearch visits
    # (local_search.rs:155-199); no shipped configuration does this, but it is reference behaviour.  The product driver
    # hands the GPU the tabu boards at Hamming distance 1 as an exclusion list; trajectory = the oracle's literal driver.
    import ls_amd
    ctx.set_algo("auto")
    p = _nq_params(n, expiry=0)
    p["all_solutions_capacity"] = 50                      # small enough for the size eviction (:173-180) to kick in
    s = ls_amd.Solver.nqueens(ctx, n, seed, p)
    o = lo.IteratedLocalSearch(lo.NQueens(n, "literal"), p, seed)
    assert s.current() == o.current
    t1, t2 = _trace_product(s, rounds), _trace_oracle(o, rounds)
    assert t1 == t2
    assert s.best() == o.get_best_solution()
    s.close()


def test_general_history_expiry_zero_scheduling(ctx):
    import ls_amd
    p = dict(lo.SCHED_MAIN_PARAMS)
    p["all_solution_iteration_expiry"] = 0
    p["all_solutions_capacity"] = 40
    p["iterated_local_search_max_iterations"] = 12
    s = ls_amd.Solver.scheduling(ctx, 31, 0, 7, "42", p)
    o = lo.IteratedLocalSearch(lo.Scheduling((2022, 5, 9), (2022, 6, 8), 7), p, "42")
    assert s.current() == o.current
    t1, t2 = _trace_product(s, 12), _trace_oracle(o, 12)
    assert t1 == t2
    assert s.best() == o.get_best_solution()
    s.close()


def test_c_example_program_matches_python_solver(ctx, tmp_path):
    # examples/nqueens_c.c: the reference's nqueens binary in plain C on the C ABI (no Python, no torch in the process)
    import subprocess
    import ls_amd
    from test_abi import _build_c_example
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe, "8", "42"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rows = [int(x) for x in r.stdout.split("[")[1].split("]")[0].split(",")]
    assert "result.score: 0" in r.stdout
    ctx.set_algo("auto")
    s = ls_amd.Solver.nqueens(ctx, 8, "42", _nq_params(8))
    while not s.is_finished():
        s.execute_round()
        b = s.best()
        if b is not None and b[0] == 0:
            break
    assert list(s.best()[1]) == rows
    s.close()


def test_c_scheduling_example_matches_python_solver(ctx, tmp_path):
    import subprocess
    import ls_amd
    from test_abi import _build_c_example
    exe = _build_c_example(tmp_path, "employee_scheduling_c")
    r = subprocess.run([exe, "42"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    sol = [int(x) for x in r.stdout.split("[")[1].split("]")[0].split(",")]
    s = ls_amd.Solver.scheduling(ctx, 31, 0, 7, "42", dict(lo.SCHED_MAIN_PARAMS))
    while not s.is_finished():
        s.execute_round()
    score, best = s.best()
    s.close()
    assert list(best) == sol
    assert f"result.score: hard {score[0]}, soft {score[1]}" in r.stdout
