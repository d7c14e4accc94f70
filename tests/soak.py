#!/usr/bin/env python3
"""Randomised parity soak (test infrastructure; not collected by pytest): random N-queens and scheduling
neighbourhoods through every kernel path vs the CPU oracle.  Usage: python tests/soak.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "constraint-solver_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402

import ls_amd  # noqa: E402
import oracle as orc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end = time.time() + budget
stats = dict(nq=0, nq_candidates=0, sched=0, sched_candidates=0, sched_scores=0)


def fail(msg, **kw):
    print("MISMATCH:", msg)
    for k, v in kw.items():
        print(" ", k, "=", v if not hasattr(v, "tolist") else v.tolist())
    sys.exit(1)


def nq_case(ctx):
    n = int(rng.choice([rng.integers(1, 70), rng.integers(60, 600), rng.integers(500, 4200), rng.integers(4097, 9000)],
                       p=[0.35, 0.35, 0.25, 0.05]))
    kind = rng.integers(0, 4)
    if kind == 0:
        rows = rng.permutation(n)
    elif kind == 1:
        rows = rng.integers(0, n, size=n)
    elif kind == 2:                                   # few distinct rows: massive ties and conflicts
        rows = rng.integers(0, max(1, n // 50 + 1), size=n)
    else:                                             # near-diagonal
        rows = np.clip(np.arange(n) + rng.integers(-2, 3, size=n), 0, n - 1)
    n_cols = int(rng.integers(0, min(n, 8) + 1))
    cols = rng.integers(0, n, size=n_cols) if rng.integers(0, 4) == 0 else rng.permutation(n)[:n_cols]
    window = int(rng.choice([0, 1, rng.integers(1, 3 * n + 2), n_cols * (n - 1), 10 ** 12]))
    form = "literal" if n <= 48 else ("delta" if n <= 400 else "table")
    exp = orc.nq_neighbourhood(rows, cols, window, form)
    cur = orc.nq_score(rows)
    for algo in ("scan", "table"):
        ctx.set_algo(algo)
        got, scores = ctx.nqueens_eval(rows, cur, cols, window, want_scores=True)
        if (got["n_scored"], got["best_index"]) != (exp["n_scored"], exp["best_index"]) or \
                (exp["best_index"] >= 0 and got["best_score"] != exp["best_score"]) or not np.array_equal(scores, exp["scores"]):
            fail(f"nqueens {algo}", n=n, rows=rows, cols=cols, window=window, got=got,
                 exp={k: exp[k] for k in ("n_scored", "best_index", "best_score")})
        got2, _ = ctx.nqueens_eval(rows, cur, cols, window)          # production path (no score output; "table": the resident LS-step server)
        if got2 != got:
            fail(f"nqueens {algo}: debug and production paths differ", n=n, rows=rows, cols=cols, window=window)
    stats["nq"] += 1
    stats["nq_candidates"] += exp["n_scored"]
    # the solver's per-iteration call: result + winner board's column scores (fused step kernel under "auto"
    # whenever the launch fits it, multi-launch otherwise), here as a ragged 3-restart batch
    if n >= 2:
        R = 3
        brows = np.stack([rows, rng.integers(0, n, size=n), rng.permutation(n)])
        mc = max(1, n_cols)
        bcols = np.stack([np.resize(cols, mc) if n_cols else np.zeros(mc, dtype=np.int64), rng.integers(0, n, size=mc),
                          rng.permutation(n)[:mc] if mc <= n else rng.integers(0, n, size=mc)])
        bn = np.array([n_cols, mc, int(rng.integers(0, mc + 1))])
        bcur = [orc.nq_score(b) for b in brows]
        for algo in ("auto", "scan"):
            ctx.set_algo(algo)
            out, nxt = ctx.nqueens_eval_batch_next(brows, bcur, bcols, bn, window)
            for r in range(R):
                c = bcols[r][:bn[r]]
                e = orc.nq_neighbourhood(brows[r], c, window, "table", want_scores=False) if len(c) else dict(n_scored=0, best_index=-1)
                b = brows[r].copy()
                if e["best_index"] >= 0:
                    b[c[e["best_index"] // n]] = e["best_index"] % n
                if (out[r]["n_scored"], out[r]["best_index"]) != (e["n_scored"], e["best_index"]) or \
                        (e["best_index"] >= 0 and out[r]["best_score"] != e["best_score"]) or \
                        not np.array_equal(nxt[r], orc.nq_col_scores(b)):
                    fail(f"nqueens step {algo}", n=n, rows=brows[r], cols=c, window=window, got=out[r])
        stats["nq_steps"] = stats.get("nq_steps", 0) + 2 * R


def nq_two_column_case(ctx):
    """Batches whose wavefront shares hold several whole columns (n > 512; beyond 4096 the LDS kernel): the scan kernel's two-column walk,
    its hand-over to the one-column walk at cut columns, odd leftovers and the window's end."""
    n = int(rng.integers(513, 4097)) if rng.integers(0, 6) else int(rng.integers(4097, 9001))   # the LDS kernel now and then
    R = int(rng.choice([32, 64, 128, 256])) if n <= 4096 else 256
    n_cols = int(rng.integers(2, 16)) * ((512 if n > 2048 else 1024) // R) + int(rng.integers(0, 5))
    if n > 4096:
        n_cols = int(rng.integers(2, 4)) * 32 + int(rng.integers(0, 20))     # 8192 / R = 32 wavefronts per restart
    rows = np.stack([rng.integers(0, n, size=n) if rng.integers(0, 3) else rng.permutation(n) for _ in range(R)])
    cols = rng.integers(0, n, size=(R, n_cols))
    dup = rng.integers(0, 2, size=(R, n_cols - 1)).astype(bool) & (rng.integers(0, 8, size=(R, n_cols - 1)) == 0)
    cols[:, 1:][dup] = cols[:, :-1][dup]                              # list neighbours that are the same column
    nc = rng.integers(0, n_cols + 1, size=R)
    nc[rng.integers(0, R, size=R // 2)] = n_cols
    k = int(rng.integers(0, n_cols + 1))
    window = int(rng.choice([10 ** 12, k * (n - 1), k * (n - 1) + int(rng.integers(1, n - 1)), max(1, k * (n - 1) - 1)]))
    want = bool(rng.integers(0, 2)) and R * n_cols * n * 8 < (1 << 30)
    cur = [int(orc.nq_col_scores_table(rows[r]).sum()) for r in range(R)]
    ctx.set_algo("scan")
    out, scores = ctx.nqueens_eval_batch(rows, cur, cols, nc, window, want_scores=want)
    for r in range(R):
        e = orc.nq_neighbourhood(rows[r], cols[r, :nc[r]], window, "table", want_scores=want)
        if (out[r]["n_scored"], out[r]["best_index"]) != (e["n_scored"], e["best_index"]) or \
                (e["best_index"] >= 0 and out[r]["best_score"] != e["best_score"]) or \
                (want and not np.array_equal(scores[r, :nc[r] * n], e["scores"])):
            fail("nqueens scan, several columns per wavefront", n=n, R=R, r=r, rows=rows[r], cols=cols[r, :nc[r]], window=window,
                 got=out[r], exp={kk: e[kk] for kk in ("n_scored", "best_index", "best_score")})
        stats["nq_candidates"] += e["n_scored"]
    stats["nq_two_column_batches"] = stats.get("nq_two_column_batches", 0) + 1


def sched_case(ctx):
    D = int(rng.choice([rng.integers(1, 32), 31, rng.integers(32, 65)], p=[0.4, 0.3, 0.3]))
    S = int(min(64, D + rng.integers(0, 2)))
    E = int(rng.choice([1, 2, 3, 7, 40, 1000]))
    start_wd = int(rng.integers(0, 7))
    hol = {}
    bits = np.zeros(E, dtype=np.uint64)
    for e in range(min(E, 30)):
        for d in rng.choice(D, size=int(min(D, rng.integers(0, 4))), replace=False):
            hol.setdefault(e, []).append(int(d))
            bits[e] |= np.uint64(1) << np.uint64(int(d))
    # oracle instance with the same start weekday: 2024-01-01 is a Monday
    import datetime as dt
    start = dt.date(2024, 1, 1) + dt.timedelta(days=start_wd)
    end = start + dt.timedelta(days=D - 1)
    inst = orc.Sched((start.year, start.month, start.day), (end.year, end.month, end.day), hol)
    assert inst.n_scored_days == D and inst.start_weekday == start_wd
    ctx.sched_instance_upload(D, S, start_wd, E, bits)
    pool = min(E, int(rng.choice([1, 2, 4, 12, 1000])))
    states = rng.integers(0, pool, size=(16, S))
    hard, soft, terms = ctx.sched_score(states, want_terms=True)
    for i in range(len(states)):
        h, s, t = inst.score(states[i], terms=True)
        if (hard[i], soft[i]) != (h, s) or terms[i].tolist() != t.tolist():
            fail("sched score", D=D, S=S, E=E, wd=start_wd, hol=hol, state=states[i], got=terms[i], exp=t)
    stats["sched_scores"] += len(states)
    state = states[0]
    n_moves = int(rng.choice([0, 1, rng.integers(1, 200), rng.integers(200, 3000)]))
    moves = np.zeros((n_moves, 3), dtype=np.int64)
    for i in range(n_moves):
        if rng.integers(0, 5) == 0 or S < 2:
            moves[i] = (0, rng.integers(0, S), rng.integers(0, pool))
        else:
            moves[i] = (1, rng.integers(0, S), rng.integers(0, S))
    window = int(rng.choice([0, 1, rng.integers(1, n_moves + 2), 10 ** 12]))
    if n_moves == 0:
        return
    exp = inst.neighbourhood(state, moves, window)
    got, gh, gs = ctx.sched_eval(state, moves, window, want_scores=True)
    if (got["n_scored"], got["best_index"]) != (exp["n_scored"], exp["best_index"]) or \
            (exp["best_index"] >= 0 and (got["best_hard"], got["best_soft"]) != (exp["best_hard"], exp["best_soft"])) or \
            not np.array_equal(gh, exp["hard"]) or not np.array_equal(gs, exp["soft"]):
        fail("sched neighbourhood", D=D, S=S, E=E, wd=start_wd, hol=hol, state=state, moves=moves, window=window, got=got,
             exp={k: exp[k] for k in ("n_scored", "best_index", "best_hard", "best_soft")})
    got2, _, _ = ctx.sched_eval(state, moves, window)
    if got2 != got:
        fail("sched: debug and production paths differ", D=D, S=S, E=E, state=state, moves=moves, window=window)
    # resident move list (the solver's per-iteration entry) and, for small employee counts, the exhaustive proposer
    ctx.sched_moves_upload(moves)
    got3 = ctx.sched_eval_resident(state, n_moves, window)
    if got3 != {k: got[k] for k in got3}:
        fail("sched: resident entry differs", D=D, S=S, E=E, state=state, moves=moves, window=window, got=got3)
    if E <= 40 or rng.integers(0, 4) == 0:    # (E = 1000: 64 000 candidates on the resident exhaustive grid, now and then)
        ex = orc.Sched.exhaustive_moves(state, range(E))
        wx = int(rng.choice([1, rng.integers(1, len(ex) + 2), 10 ** 12]))
        expx = inst.neighbourhood(state, ex, wx)
        gotx = ctx.sched_eval_exhaustive(state, wx)[0]
        if (gotx["n_scored"], gotx["best_index"]) != (expx["n_scored"], expx["best_index"]) or \
                (expx["best_index"] >= 0 and (gotx["best_hard"], gotx["best_soft"]) != (expx["best_hard"], expx["best_soft"])):
            fail("sched exhaustive", D=D, S=S, E=E, wd=start_wd, hol=hol, state=state, window=wx, got=gotx,
                 exp={k: expx[k] for k in ("n_scored", "best_index", "best_hard", "best_soft")})
        stats["sched_candidates"] += expx["n_scored"]
    stats["sched"] += 1
    stats["sched_candidates"] += exp["n_scored"]


def server_walk(ctx):
    """A short LS walk through the resident LS-step server (one-restart calls continuing from the winner, a perturbation in
    between): the slot's LDS board / tables and the host-kept column scores against the oracle at every step."""
    n = int(rng.choice([rng.integers(2, 200), rng.integers(128, 1100) & ~1, 4096]))
    rows = rng.integers(0, n, size=n)
    ctx.set_algo("auto")
    for it in range(12):
        mc = int(rng.integers(1, min(n, 16) + 1))
        cols = rng.permutation(n)[:mc]
        window = int(rng.choice([1, 5 * n, rng.integers(1, mc * (n - 1) + 2)]))
        exp = orc.nq_neighbourhood(rows, cols, window, "table", want_scores=False)
        out, nxt = ctx.nqueens_eval_batch_next(rows.reshape(1, -1), [orc.nq_score(rows)], cols.reshape(1, -1), [mc], window)
        b = rows.copy()
        if exp["best_index"] >= 0:
            b[cols[exp["best_index"] // n]] = exp["best_index"] % n
        if (out[0]["n_scored"], out[0]["best_index"]) != (exp["n_scored"], exp["best_index"]) or \
                (exp["best_index"] >= 0 and out[0]["best_score"] != exp["best_score"]) or not np.array_equal(nxt[0], orc.nq_col_scores(b)):
            fail("nqueens server walk", n=n, rows=rows, cols=cols, window=window, got=out[0])
        rows = rng.integers(0, n, size=n) if it == 6 else b
    stats["server_steps"] = stats.get("server_steps", 0) + 12


def multi_sched_case(ctx):
    """Many scheduling restarts on one GPU (ls_multi_solver_create_sched): the slots of one shared resident launch -- step-server
    workgroups (small windows) or sub-grids (windows beyond 512 moves) -- against each restart's own ls_solver run."""
    E = int(rng.choice([3, 7, 40, 600]))
    D = int(rng.choice([14, 31, 45]))
    window = int(rng.choice([20, 100, 700, 1500]))
    bits = np.zeros(E, dtype=np.uint64)
    for e in range(min(E, 20)):
        for d in rng.choice(D, size=int(rng.integers(0, 3)), replace=False):
            bits[e] |= np.uint64(1) << np.uint64(int(d))
    p = dict(local_search_max_iterations=200, window_size=window, best_solutions_capacity=8, all_solutions_capacity=1000,
             all_solution_iteration_expiry=100, iterated_local_search_max_iterations=3, max_allow_no_improvement_for=5)
    seeds = [str(int(rng.integers(0, 10 ** 6))) for _ in range(int(rng.integers(2, 7)))]
    m = ls_amd.MultiSolver.scheduling(ctx, D, int(rng.integers(0, 7)) if False else 0, E, seeds, p, holiday_bits=bits, n_slots=int(rng.integers(1, len(seeds) + 1)))
    m.run(max_rounds_per_restart=3)
    got = [(m.best(r), m.restart_stats(r)["ls_iterations"]) for r in range(len(seeds))]
    m.close()
    for r, sd in enumerate(seeds):
        s1 = ls_amd.Solver.scheduling(ctx, D, 0, E, sd, p, holiday_bits=bits)
        while not s1.is_finished():
            s1.execute_round()
        one = (s1.best(), s1.stats()["ls_iterations"])
        s1.close()
        if one != got[r]:
            fail("multi-restart scheduling run differs from the single run", E=E, D=D, window=window, seed=sd, got=got[r], exp=one)
    stats["multi_sched_restarts"] = stats.get("multi_sched_restarts", 0) + len(seeds)


os.environ["LS_NQ_STEP"] = "3"          # a second context on the launch-per-call routes (the knob is read when a context is created)
ctx3 = ls_amd.Context(0)
os.environ.pop("LS_NQ_STEP")
with ls_amd.Context(0) as ctx:
    last = time.time()
    turn = 0
    while time.time() < t_end:
        nq_case(ctx if turn % 2 == 0 else ctx3)   # batched LS steps: resident server slots | fused launches
        sched_case(ctx)
        server_walk(ctx)
        if turn % 8 == 0:
            multi_sched_case(ctx)
        if turn % 8 == 4:
            nq_two_column_case(ctx)
        turn += 1
        if time.time() - last > 30:
            print("progress", stats, flush=True)
            last = time.time()
    cnt = ctx.counters()
    cnt3 = ctx3.counters()
    ctx3.close()
    assert cnt3["server_requests"] == 0 and cnt3["completion_word_fallbacks"] == 0, cnt3
# the slow branches of the latency-critical path must not have been taken silently
assert cnt["completion_word_fallbacks"] == 0, cnt
print("SOAK OK", stats, "seed", seed, "counters", cnt)
