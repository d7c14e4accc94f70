"""CPU-only: the scheduling oracle against (a) an independent Python/Counter
restatement of examples/employee-scheduling/src/lib.rs:261-375 and (b) the
hand-derivable cases of SURVEY.md 8(c) (tests/golden/sched_kats.json).  The
reference itself holds no test for these scores: PARITY UNPINNED by the
reference."""
import datetime as dt
import json
import os
from collections import Counter

import numpy as np
import pytest

import oracle as orc


def py_score(start, end, e, holidays):
    """Independent formulation with datetime + Counter (mirrors the reference's
    HashMap/itertools structure rather than the oracle's loops)."""
    d0, d1 = dt.date(*start), dt.date(*end)
    days = []
    cur = d0
    idx = 0
    while True:
        days.append((cur, int(e[idx])))
        if cur >= d1:
            break
        cur += dt.timedelta(days=1)
        idx += 1
    wk = lambda d: d.weekday() >= 5
    hard = 0
    for emp, hs in holidays.items():
        for h in set(hs):
            if days[h][1] == int(emp):
                hard += 1
    h1 = hard
    h2 = sum(1 for a, b in zip(days, days[1:]) if a[1] == b[1])
    h3 = 0
    for i in range(len(days) - 8):
        w = days[i:i + 9]
        if not (wk(w[0][0]) and wk(w[1][0])):
            continue
        h3 += (w[0][1] == w[7][1]) + (w[0][1] == w[8][1]) + (w[1][1] == w[7][1]) + (w[1][1] == w[8][1])
    h4 = sum(sum(1 for c in Counter(x[1] for x in days[i:i + 14]).values() if c > 3)
             for i in range(len(days) - 13))
    s1 = sum(sum(1 for c in Counter(x[1] for x in days[i:i + 7]).values() if c > 2)
             for i in range(len(days) - 6))
    s2 = 0
    by_wd = {}
    for d, emp in days:
        if wk(d):
            continue
        by_wd.setdefault(d.weekday(), Counter())[emp] += 1
    for cnt in by_wd.values():
        if len(cnt) <= 1:
            continue
        s2 += min(cnt.values())
    tot = Counter(emp for _, emp in days)
    s3 = (max(tot.values()) - min(tot.values())) if len(tot) >= 2 else 0
    wtot = {emp: 0 for emp in tot}
    for d, emp in days:
        if wk(d):
            wtot[emp] += 1
    s4 = (max(wtot.values()) - min(wtot.values())) if len(wtot) >= 2 else 0
    return h1 + h2 + h3 + h4, s1 + s2 + s3 + s4, [h1, h2, h3, h4, s1, s2, s3, s4]


def _pattern(name, n):
    d = np.arange(n)
    return {"const0": d * 0, "mod7": d % 7, "mod2": d % 2, "mod3": d % 3}[name].astype(np.int64)


def test_survey_kats(golden_dir):
    with open(os.path.join(golden_dir, "sched_kats.json")) as f:
        g = json.load(f)
    for case in g["cases"]:
        hol = {int(k): v for k, v in case["holidays"].items()}
        inst = orc.Sched(g["start"], g["end"], hol)
        assert inst.n_scored_days == 31 and inst.start_weekday == 0
        e = _pattern(case["pattern"], g["n_stored"])
        h, s, t = inst.score(e, terms=True)
        assert (h, s) == (case["hard"], case["soft"]), case
        assert t.tolist() == case["terms"], case
        assert py_score(g["start"], g["end"], e, hol)[2] == case["terms"]


@pytest.mark.parametrize("start,end,n_emp", [
    ((2022, 5, 9), (2022, 6, 8), 7),      # the reference example instance (main.rs:11-21)
    ((2022, 5, 9), (2022, 6, 8), 2),
    ((2022, 5, 9), (2022, 6, 8), 40),
    ((2023, 12, 30), (2024, 2, 29), 5),   # 62 days, starts on a Saturday, crosses a leap day
    ((2024, 1, 3), (2024, 1, 10), 3),     # 8 days: no 9- or 14-window
    ((2024, 1, 3), (2024, 1, 3), 3),      # single day
])
def test_random_vs_python(start, end, n_emp):
    rng = np.random.default_rng(hash((start, end, n_emp)) % 2**32)
    d = (dt.date(*end) - dt.date(*start)).days + 1
    for trial in range(25):
        hol = {}
        if trial % 2:
            for emp in range(n_emp):
                k = int(rng.integers(0, 4))
                if k:
                    hol[emp] = rng.choice(d, size=min(k, d), replace=False).tolist()
        inst = orc.Sched(start, end, hol)
        assert inst.n_scored_days == d
        assert inst.start_weekday == dt.date(*start).weekday()
        e = rng.integers(0, n_emp, size=d + 1)
        h, s, t = inst.score(e, terms=True)
        ph, ps, pt = py_score(start, end, e, hol)
        assert t.tolist() == pt
        assert (h, s) == (ph, ps)


def test_stored_tail_is_not_scored_but_orders_and_tabus():
    # entry 31 is never scored but is part of Eq/Ord.
    inst = orc.Sched((2022, 5, 9), (2022, 6, 8))
    e = (np.arange(32) % 7).astype(np.int64)
    e2 = e.copy(); e2[31] = 5
    assert inst.score(e) == inst.score(e2)
    # a ChangeDay on a scored day to the same employee is a no-op (tabu), one to another is not
    res = inst.neighbourhood(e, [(0, 3, e[3]), (0, 3, (e[3] + 1) % 7), (1, 0, 7), (1, 0, 1)], 100)
    assert res["hard"][0] == orc.SKIPPED and res["hard"][2] == orc.SKIPPED  # e[0]==e[7]: swap is a no-op
    assert res["n_scored"] == 2


def test_neighbourhood_argmin_matches_bruteforce():
    inst = orc.Sched((2022, 5, 9), (2022, 6, 8))
    rng = np.random.default_rng(7)
    e = rng.integers(0, 7, size=32)
    moves = []
    for _ in range(300):
        if rng.integers(0, 5) == 0:
            moves.append((0, int(rng.integers(0, 31)), int(rng.integers(0, 7))))
        else:
            a, b = rng.choice(31, size=2, replace=False)
            moves.append((1, int(a), int(b)))
    res = inst.neighbourhood(e, moves, 100)
    cands = []
    for k, (t, a, b) in enumerate(moves):
        c = e.copy()
        if t == 0:
            c[a] = b
        else:
            c[a], c[b] = e[b], e[a]
        if (c == e).all():
            continue
        if len(cands) == 100:
            break
        h, s, _ = py_score((2022, 5, 9), (2022, 6, 8), c, {})
        cands.append(((h, s, c.tolist()), k))
        assert (res["hard"][k], res["soft"][k]) == (h, s)
    assert res["n_scored"] == len(cands)
    assert res["best_index"] == min(cands)[1]


def test_exhaustive_moves_order():
    # lib.rs:526-550: ring walk per stored day, last candidate of a day is the base
    e = np.array([2, 0, 1], dtype=np.int64)
    mv = orc.Sched.exhaustive_moves(e, [0, 1, 2])
    assert mv.tolist() == [[0, 0, 0], [0, 0, 1], [0, 0, 2],
                           [0, 1, 1], [0, 1, 2], [0, 1, 0],
                           [0, 2, 2], [0, 2, 0], [0, 2, 1]]
