"""CPU-only: the oracle's restatement of the reference's host driver reproduces the reference's
own end-to-end test — examples/nqueens/src/main.rs:156-200 `repeatable`: n = 8, seeds
"42".."49", identical result on every run and final score 0 (expiry 1000 there)."""
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
