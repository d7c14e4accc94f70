"""CPU-only: pins the N-queens oracle against the reference's own known-answer
tests (examples/nqueens/src/lib.rs:93-119) and cross-checks its three
formulations (literal / delta / count-table) against each other."""
import json
import os

import numpy as np
import pytest

import oracle as orc


def _kats(golden_dir):
    with open(os.path.join(golden_dir, "nqueens_kats.json")) as f:
        return json.load(f)


def test_reference_kats(golden_dir):
    for group in ("reference", "derived"):
        for kat in _kats(golden_dir)[group]:
            got = orc.nq_col_scores(kat["rows"])
            assert got.tolist() == kat["col_scores"], kat
            assert orc.nq_score(kat["rows"]) == kat["score"], kat
            assert orc.nq_col_scores_table(kat["rows"]).tolist() == kat["col_scores"], kat


@pytest.mark.parametrize("n", [1, 2, 3, 8, 33, 100, 257])
def test_score_identities(n):
    rng = np.random.default_rng(n)
    for perm in (False, True):
        rows = rng.permutation(n) if perm else rng.integers(0, n, size=n)
        cs = orc.nq_col_scores(rows)
        assert cs.sum() == orc.nq_score(rows)
        assert (cs == orc.nq_col_scores_table(rows)).all()
        # score = sum over lines of k(k-1)  
        tot = 0
        for keys in (rows, rows - np.arange(n), rows + np.arange(n)):
            _, cnt = np.unique(keys, return_counts=True)
            tot += int((cnt * (cnt - 1)).sum())
        assert tot == cs.sum()


@pytest.mark.parametrize("n,n_cols,window", [(8, 1, 40), (8, 3, 10), (33, 5, 10**9), (100, 5, 333), (64, 64, 10**9)])
def test_formulations_agree(n, n_cols, window):
    rng = np.random.default_rng(1000 * n + n_cols)
    for trial in range(3):
        rows = rng.integers(0, n, size=n) if trial else rng.permutation(n)
        cols = rng.permutation(n)[:n_cols]
        lit = orc.nq_neighbourhood(rows, cols, window, "literal")
        for form in ("delta", "table"):
            other = orc.nq_neighbourhood(rows, cols, window, form)
            assert other["n_scored"] == lit["n_scored"]
            assert other["best_index"] == lit["best_index"]
            assert other["best_score"] == lit["best_score"]
            assert (other["scores"] == lit["scores"]).all()


def test_window_counts_only_non_tabu():
    # .filter(tabu).map(score).take(window) — no-ops do not consume window
    rows = np.array([1, 3, 0, 2], dtype=np.int64)
    res = orc.nq_neighbourhood(rows, [0, 1], 4, "literal")
    sc = res["scores"]
    assert res["n_scored"] == 4
    assert sc[1] == orc.SKIPPED            # (col 0, value 1) is the current solution
    assert (sc[[0, 2, 3, 4]] != orc.SKIPPED).all() and (sc[5:] == orc.SKIPPED).all()


def test_tie_break_is_lexicographic_solution():
    # all-same-row board: many equal-score candidates; winner = smallest vector
    rows = np.zeros(6, dtype=np.int64)
    res = orc.nq_neighbourhood(rows, [4, 1], 10**9, "literal")
    sc = res["scores"]
    best = sc[sc != orc.SKIPPED].min()
    ties = [k for k in range(len(sc)) if sc[k] == best]
    boards = []
    for k in ties:
        b = rows.copy()
        b[[4, 1][k // 6]] = k % 6
        boards.append((b.tolist(), k))
    assert min(boards)[1] == res["best_index"]


def test_empty_neighbourhood():
    res = orc.nq_neighbourhood([0, 2], [], 10, "literal")
    assert res["n_scored"] == 0 and res["best_index"] == -1
    res = orc.nq_neighbourhood([0], [0], 10, "literal")   # only candidate is the no-op
    assert res["n_scored"] == 0 and res["best_index"] == -1
