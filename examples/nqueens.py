#!/usr/bin/env python3
"""N-queens example on the GPU path — the counterpart of the reference's `examples/nqueens` binary
(examples/nqueens/src/main.rs:95-150): same flags (--seed/-s string, --board-size/-b), same
hard-coded hyper-parameters (:129-135), same output shape (board drawing + score).

    python examples/nqueens.py --board-size 8 --seed 42 [--algo auto|scan|table]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
import ls_amd  # noqa: E402


def draw(rows) -> str:
    """The reference's Debug drawing of a board (examples/nqueens/src/lib.rs:26-60)."""
    n = len(rows)
    lines = []
    for r in range(2 * n + 1):
        if r % 2 == 0:
            lines.append("-" * (4 * n + 1))
            continue
        row = (r - 1) // 2
        lines.append("".join("| Q " if rows[c] == row else "|   " for c in range(n)) + "|")
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser(description="Local Search N-Queens Example (MI355X)")
    ap.add_argument("-s", "--seed", default="42", help="Random seed, any string")
    ap.add_argument("-b", "--board-size", type=int, default=8)
    ap.add_argument("--algo", choices=["auto", "scan", "table"], default="auto")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    n = args.board_size
    params = dict(local_search_max_iterations=10_000, window_size=5 * n, best_solutions_capacity=32,
                  all_solutions_capacity=100_000, all_solution_iteration_expiry=10_000,
                  iterated_local_search_max_iterations=10_000, max_allow_no_improvement_for=5)
    print("local search n-queens example")
    with ls_amd.Context(args.device, algo=args.algo) as ctx:
        s = ls_amd.Solver.nqueens(ctx, n, args.seed, params)
        t0 = time.perf_counter()
        while not s.is_finished():
            s.execute_round()
            b = s.best()
            if b is not None and b[0] == 0:
                break                      # the reference keeps looping through no-op rounds
        dt = time.perf_counter() - t0
        score, rows = s.best()
        st = s.stats()
        s.close()
    if n <= 64:
        print("result.solution:\n" + draw(rows))
    else:
        print("result.solution: rows =", list(rows[:16]), "...")
    print(f"result.score: NQueensScore({score})")
    print(f"[{dt:.3f} s, {st['ls_iterations']} local-search iterations, {st['neighbours_scored']} candidates scored on the GPU]")


if __name__ == "__main__":
    main()
