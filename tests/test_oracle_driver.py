"""CPU-only: the oracle's restatement of the reference's host driver reproduces the reference's
own end-to-end test — examples/nqueens/src/main.rs:156-200 `repeatable`: n = 8, seeds
"42".."49", identical result on every run and final score 0 (expiry 1000 there)."""
import pytest

import ls_oracle as lo


def _params():
    p = dict(lo.NQUEENS_MAIN_PARAMS)
    p["window_size"] = 8 * 5
    p["all_solution_iteration_expiry"] = 1_000
    return p


def test_repeatable_nq8():
    for seed in map(str, range(42, 50)):
        results = []
        for _ in range(2):
            ils = lo.IteratedLocalSearch(lo.NQueens(8), _params(), seed)
            results.append(lo.run(ils))
        assert results[0] == results[1], seed
        assert results[0][0] == 0, seed                      # satisfiable
        assert len(set(results[0][1])) == 8                 # a real 8-queens placement


def test_sched_example_instance_runs_and_improves():
    # examples/employee-scheduling/src/main.rs:11-31 (7 employees, 31 scored days), a few rounds
    p = dict(lo.SCHED_MAIN_PARAMS)
    ils = lo.IteratedLocalSearch(lo.Scheduling((2022, 5, 9), (2022, 6, 8), 7), p, "42")
    first = ils.current[0]
    trace = []
    lo.run(ils, max_rounds=3, trace=trace)
    best = ils.get_best_solution()
    assert best is not None and best[0] < first
    assert len(best[1]) == 32


# ---- the reference's own tests of the GENERIC loop (they use a third plug-in, the f64 Ackley function) ----

def test_ackley_function_kats():
    # math-util/src/ackley.rs:60-104 (values from the published MATLAB implementation, epsilon 1e-12)
    assert abs(lo.ackley_function([0.0, 0.0]) - 0.0) <= 1e-12
    assert abs(lo.ackley_function([1.0, 1.0]) - 3.625384938440363) <= 1e-12
    assert abs(lo.ackley_function([float(i % 10) for i in range(20)]) - 13.12408690638194) <= 1e-12


def _ackley_local_search(seed=42):
    # local_search.rs:364-391: dimensions 2, move size 1e-6..0.1, 100 000 iterations, window 256, capacities 16 / 10 000 / 10 000
    return lo.LocalSearch(lo.Ackley(2, 1e-6, 0.1), 100_000, 256, 16, 10_000, 10_000, lo.ChaCha20Rng(lo.seed_from_u64(seed)))


def test_ackley_local_minima_found():
    # local_search.rs:363-418: from a random start the search ends strictly better and somewhere else
    ls = _ackley_local_search()
    start = ls.p.generate_initial_solution(lo.ChaCha20Rng(lo.seed_from_u64(42)))
    end = ls.execute(start, 1)
    assert end[0] < ls.p.score(start)
    assert end[1] != start


def test_ackley_when_starting_from_global_minima_does_not_move():
    # local_search.rs:420-470
    ls = _ackley_local_search()
    start = (0.0, 0.0)
    end = ls.execute(start, 1)
    assert abs(end[0] - ls.p.score(start)) <= 1e-12
    assert end[1] == start


@pytest.mark.parametrize("dimensions", [2, 10, 20])
def test_ackley_iterated_local_search_reaches_the_global_minimum(dimensions):
    # iterated_local_search.rs:221-321 (`_ackley` for 2, 10 and 20 dimensions, seed 0): move size 1e-3..0.5, window 500,
    # LS max 100 000, ILS max 10 000 rounds, no-improvement limit 5; the best solution scores 0 within 1e-2
    params = dict(local_search_max_iterations=100_000, window_size=500, best_solutions_capacity=16,
                  all_solutions_capacity=10_000, all_solution_iteration_expiry=10_000,
                  iterated_local_search_max_iterations=10_000, max_allow_no_improvement_for=5)
    ils = lo.IteratedLocalSearch(lo.Ackley(dimensions, 1e-3, 0.5), params, lo.seed_from_u64(0))
    while not ils.is_finished():
        ils.execute_round()
        b = ils.get_best_solution()
        if b is not None and lo.Ackley.is_best(b[0]):
            break                                            # the reference keeps spinning through no-op rounds
    assert abs(ils.get_best_solution()[0]) <= 1e-2


def _literal_pipeline(candidates, tabu, score, window):
    """local_search.rs:315-325, literally: filter(!tabu) -> map(score) -> take(window) -> collect -> sort -> first."""
    nb = []
    for sol in candidates:
        if sol in tabu:
            continue
        nb.append((score(sol), sol))
        if len(nb) == window:
            break
    nb.sort()
    return nb[0] if nb else None


@pytest.mark.parametrize("n", [4, 8, 13, 20])
def test_winner_only_proposer_is_equivalent_under_the_unchanged_pipeline(n):
    # The Rust drop-in (bindings/rust/examples/*/gpu.rs) leaves LocalSearch::execute untouched
    # and plugs in a MoveProposer that yields ONLY the neighbourhood's winner.  With the tabu set the reference's History
    # produces for its shipped configurations ({current}), feeding that one-element iterator through the literal
    # filter/map/take/sort/first pipeline must select exactly what the full candidate stream selects.
    import numpy as np
    rng = np.random.default_rng(n)
    for trial in range(60):
        rows = tuple(int(x) for x in (rng.permutation(n) if trial % 2 else rng.integers(0, n, size=n)))
        cols = [int(c) for c in rng.integers(0, n, size=int(rng.integers(1, 4)))]
        window = int(rng.integers(1, len(cols) * n + 2))
        tabu = {rows}

        def full():                               # MoveIterator, examples/nqueens/src/lib.rs:214-234
            for c in cols:
                for v in range(n):
                    b = list(rows); b[c] = v
                    yield tuple(b)

        expect = _literal_pipeline(full(), tabu, lambda b: lo.orc.nq_score(b), window)
        res = lo.orc.nq_neighbourhood(rows, cols, window, "literal", want_scores=False)   # what the GPU call returns
        if res["best_index"] < 0:
            assert expect is None
            continue
        w = list(rows); w[cols[res["best_index"] // n]] = res["best_index"] % n
        got = _literal_pipeline(iter([tuple(w)]), tabu, lambda b: lo.orc.nq_score(b), window)
        assert got == expect == (res["best_score"], tuple(w))


def test_general_history_expiry_zero_runs_on_the_oracle_driver():
    # all_solution_iteration_expiry == 0: nothing is evicted by age, the tabu set grows up to all_solutions_capacity + 1
    # (local_search.rs:155-199) and the literal HashSet filter is in force; the driver must still solve n = 8
    p = dict(lo.NQUEENS_MAIN_PARAMS)
    p.update(window_size=40, all_solution_iteration_expiry=0, all_solutions_capacity=50)
    o = lo.IteratedLocalSearch(lo.NQueens(8, "literal"), p, "42")
    biggest = 0
    for _ in range(25):
        o.execute_round()
        biggest = max(biggest, len(o.local_search.history.lookup))
    assert biggest == 51
    assert o.get_best_solution()[0] == 0
