"""The arithmetic of the scheduling DELTA scorer (sched_device.hpp: sched_delta_score), restated with Python
integers and checked against the CPU oracle's full rescoring (examples/employee-scheduling/src/lib.rs:261-375):

  * every term of the score is a function of the per-employee day sets, and a ChangeDay / SwapDays candidate
    changes exactly two of them;
  * H4 / S1 = number of windows in which a day set holds more than 3 of 14 / 2 of 7 days, by bit-sliced sliding sums;
  * S2 / S3 / S4 follow from the SETS of per-employee counts held by at least 1 / 2 / 3 employees.

No GPU: this pins the formulas the kernel implements (tests/test_gpu_sched.py pins the kernel itself)."""
import datetime as dt

import numpy as np
import pytest

import oracle as orc


def window_counts(m, valid7, valid14):
    a1 = m >> 1
    b0, b1 = m ^ a1, m & a1
    c0, c1 = b0 >> 2, b1 >> 2
    t0, k0, x1 = b0 ^ c0, b0 & c0, b1 ^ c1
    t1, t2 = x1 ^ k0, (b1 & c1) | (k0 & x1)
    e0, e1, m6 = b0 >> 4, b1 >> 4, m >> 6
    d0, d1 = e0 ^ m6, e1 | (e0 & m6)
    u0, k, y1 = t0 ^ d0, t0 & d0, t1 ^ d1
    u1, u2 = y1 ^ k, t2 | (t1 & d1) | (k & y1)
    f7 = bin((u2 | (u1 & u0)) & valid7).count("1")
    v0, v1, v2 = u0 >> 7, u1 >> 7, u2 >> 7
    j0, z1 = u0 & v0, u1 ^ v1
    j1, z2 = (u1 & v1) | (j0 & z1), u2 ^ v2
    x2, x3 = z2 ^ j1, (u2 & v2) | (j1 & z2)
    f14 = bin((x2 | x3) & valid14).count("1")
    return f7, f14


def popc(x):
    return bin(x).count("1")


def lsb(x):
    return (x & -x).bit_length() - 1


def msb(x):
    return x.bit_length() - 1


class Model:
    def __init__(self, D, S, start_weekday, holiday_bits, state):
        self.D, self.S = D, S
        weekend = lambda q: (start_weekday + q) % 7 >= 5
        self.scored = (1 << D) - 1
        self.weekend = sum(1 << q for q in range(D) if weekend(q))
        up = []
        for d in range(64):
            m = 0
            if d < D:
                if d + 1 < D:
                    m |= 1 << (d + 1)
                if d + 8 < D and weekend(d) and weekend(d + 1):
                    m |= (1 << (d + 7)) | (1 << (d + 8))
                if d >= 1 and d + 7 < D and weekend(d - 1) and weekend(d):
                    m |= (1 << (d + 6)) | (1 << (d + 7))
            up.append(m)
        self.up = up
        self.adj = [up[d] | sum(1 << q for q in range(64) if up[q] >> d & 1) for d in range(64)]
        every7 = sum(1 << i for i in range(0, 64, 7))
        self.wd = [((every7 << (d % 7)) & self.scored) if d < D and (start_weekday + d) % 7 < 5 else 0 for d in range(64)]
        self.valid7 = self.scored >> 6 if D >= 7 else 0
        self.valid14 = self.scored >> 13 if D >= 14 else 0
        self.hb = holiday_bits
        self.base(state)

    def hol(self, e, d):
        return (self.hb[e] >> d) & 1 if 0 <= e < len(self.hb) and d < self.D else 0

    def base(self, state):
        D, S = self.D, self.S
        self.e = [int(state[d]) if d < S else -1 - d for d in range(64)]
        self.eq = [sum(1 << q for q in range(D) if self.e[q] == self.e[d]) for d in range(64)]
        self.g = [window_counts(self.eq[d], self.valid7, self.valid14) for d in range(64)]
        self.holb = [self.hol(self.e[d], d) if d < D else 0 for d in range(64)]
        # which values are held by at least 1 / 2 / 3 employees (the kernel: prefix-OR scans over one-hot words)
        self.tot_b, self.wk_b = [0, 0, 0], [0, 0, 0]
        self.grp_b = [[0, 0, 0] for _ in range(7)]
        self.grp_size, self.grp_v = [0] * 7, [0] * 7
        hard = soft = 0
        for d in range(64):
            eq, low = self.eq[d], (1 << d) - 1
            if d < D:
                hard += self.holb[d] + popc(eq & self.up[d])
                if eq & low == 0:
                    hard += self.g[d][1]
                    soft += self.g[d][0]
                    c, wc = popc(eq), popc(eq & self.weekend)
                    self.add_value(self.tot_b, c - 1)
                    self.add_value(self.wk_b, wc)
            if self.wd[d] and eq & self.wd[d] & low == 0:
                self.add_value(self.grp_b[d % 7], popc(eq & self.wd[d]) - 1)
            if d < 7 and self.wd[d]:
                self.grp_size[d] = popc(self.wd[d])
        for r in range(7):
            if self.grp_size[r]:
                mn = lsb(self.grp_b[r][0]) + 1
                self.grp_v[r] = mn if mn < self.grp_size[r] else 0
        self.hard, self.soft_part = hard, soft + sum(self.grp_v)

    @staticmethod
    def add_value(b, v):
        one = 1 << v
        b[2] |= b[1] & one
        b[1] |= b[0] & one
        b[0] |= one

    @staticmethod
    def replace(b, hXo, vXo, hYo, vYo, hXn, vXn, hYn, vYn):
        bits = b[0]
        both = hXo and hYo and vXo == vYo
        others = b[2] if both else b[1]
        if hXo and not (others >> vXo) & 1:
            bits &= ~(1 << vXo)
        if hYo and not (others >> vYo) & 1:
            bits &= ~(1 << vYo)
        if hXn:
            bits |= 1 << vXn
        if hYn:
            bits |= 1 << vYn
        return bits

    def score(self, mtype, a, b):
        """(hard, soft) of the candidate, or None when it equals the current vector (tabu)."""
        D = self.D
        swap = mtype == 1
        p1, p2 = a, (b if swap else a)
        ea, eb = self.e[p1], self.e[p2]
        n1, n2 = (eb, ea) if swap else (b, b)
        if n1 == ea and n2 == eb:
            return None
        bit1 = 1 << p1 if p1 < D else 0
        bit2 = 1 << p2 if (swap and p2 < D) else 0
        mX, gX = self.eq[p1], self.g[p1]
        if swap:
            mY, gY = self.eq[p2], self.g[p2]
        else:
            mY = sum(1 << d for d in range(D) if self.e[d] == n1)
            gY = self.g[lsb(mY)] if mY else (0, 0)
        mXn, mYn = (mX & ~bit1) | bit2, (mY & ~bit2) | bit1
        A1, A2 = self.adj[p1], (self.adj[p2] if swap else 0)
        f7X, f14X = window_counts(mXn, self.valid7, self.valid14)
        f7Y, f14Y = window_counts(mYn, self.valid7, self.valid14)
        hard = self.hard + self.hol(n1, p1) - self.holb[p1]
        if swap:
            hard += self.hol(n2, p2) - self.holb[p2]
        hard += popc(A1 & mYn) - popc(A1 & mX) + popc(A2 & mXn) - popc(A2 & mY)
        hard += f14X + f14Y - gX[1] - gY[1]
        soft = self.soft_part + f7X + f7Y - gX[0] - gY[0]
        cX, cY, cXn, cYn = popc(mX), popc(mY), popc(mXn), popc(mYn)
        hXo, hYo, hXn, hYn = cX > 0, cY > 0, cXn > 0, cYn > 0
        bts = self.replace(self.tot_b, hXo, cX - 1, hYo, cY - 1, hXn, cXn - 1, hYn, cYn - 1)
        soft += msb(bts) - lsb(bts)
        wk = self.weekend
        bts = self.replace(self.wk_b, hXo, popc(mX & wk), hYo, popc(mY & wk), hXn, popc(mXn & wk), hYn, popc(mYn & wk))
        soft += msb(bts) - lsb(bts)
        G1, G2 = self.wd[p1], (self.wd[p2] if swap else 0)
        for i, G in enumerate((G1, G2)):
            if G == 0 or (i == 1 and G == G1):
                continue
            r = (p1 if i == 0 else p2) % 7
            x, y, xn, yn = popc(mX & G), popc(mY & G), popc(mXn & G), popc(mYn & G)
            bts = self.replace(self.grp_b[r], x > 0, x - 1, y > 0, y - 1, xn > 0, xn - 1, yn > 0, yn - 1)
            mn = lsb(bts) + 1
            soft += (mn if mn < self.grp_size[r] else 0) - self.grp_v[r]
        return hard, soft


def test_window_counts_exhaustive_small():
    # every 16-bit day set, D = 16: against the definition
    D = 16
    valid7, valid14 = (1 << (D - 6)) - 1, (1 << (D - 13)) - 1
    for m in range(1 << D):
        f7 = sum(popc((m >> w) & 0x7F) > 2 for w in range(D - 6))
        f14 = sum(popc((m >> w) & 0x3FFF) > 3 for w in range(D - 13))
        assert window_counts(m, valid7, valid14) == (f7, f14), m


def test_window_counts_random_64():
    rng = np.random.default_rng(1)
    for D in (7, 13, 14, 31, 32, 33, 63, 64):
        valid7 = ((1 << D) - 1) >> 6
        valid14 = ((1 << D) - 1) >> 13 if D >= 14 else 0
        for _ in range(2000):
            dens = rng.choice([0.1, 0.3, 0.6, 0.9])
            m = sum(1 << q for q in range(D) if rng.random() < dens)
            f7 = sum(popc((m >> w) & 0x7F) > 2 for w in range(D - 6))
            f14 = sum(popc((m >> w) & 0x3FFF) > 3 for w in range(max(D - 13, 0)))
            assert window_counts(m, valid7, valid14) == (f7, f14), (D, m)


CASES = [
    # (start, n scored days, stored days, employees, holiday density, candidates per state)
    ((2022, 5, 9), 31, 32, 7, 0.1, 400),      # the reference's example (lib.rs main.rs:21-24)
    ((2022, 5, 9), 31, 32, 3, 0.0, 300),
    ((2022, 5, 9), 31, 32, 2, 0.2, 300),
    ((2022, 5, 14), 31, 31, 5, 0.1, 300),     # starts on a Saturday; stored == scored
    ((2022, 5, 15), 30, 31, 40, 0.05, 300),   # starts on a Sunday; more employees than days
    ((2022, 5, 11), 64, 64, 9, 0.1, 300),
    ((2022, 5, 13), 63, 64, 4, 0.3, 300),
    ((2022, 5, 9), 14, 15, 3, 0.1, 200),
    ((2022, 5, 9), 13, 14, 3, 0.1, 200),
    ((2022, 5, 9), 7, 8, 2, 0.1, 200),
    ((2022, 5, 9), 6, 7, 3, 0.1, 100),
    ((2022, 5, 9), 2, 3, 3, 0.5, 60),
    ((2022, 5, 9), 1, 2, 3, 0.5, 30),
]


@pytest.mark.parametrize("start,D,S,E,hd,n_cand", CASES)
def test_delta_model_matches_full_rescoring(start, D, S, E, hd, n_cand):
    rng = np.random.default_rng(D * 131 + S * 7 + E)
    end = dt.date(*start) + dt.timedelta(days=D - 1)
    hol = {e: [d for d in range(D) if rng.random() < hd] for e in range(E)}
    inst = orc.Sched(start, (end.year, end.month, end.day), hol)
    assert inst.n_scored_days == D
    bits = [sum(1 << d for d in hol[e]) for e in range(E)]
    for trial in range(4):
        # clustered states make the range terms and weekday groups degenerate now and then
        state = rng.integers(0, max(1, E if trial % 2 == 0 else min(E, 2)), size=S)
        model = Model(D, S, inst.start_weekday, bits, state)
        assert (model.hard, None) == (inst.score(state)[0], None)
        h, s, t = inst.score(state, terms=True)
        assert model.soft_part == t[4] + t[5]
        for _ in range(n_cand):
            if rng.integers(0, 2) == 0 or S < 2:
                mv = (0, int(rng.integers(0, S)), int(rng.integers(0, E)))
            else:
                a, b = rng.choice(S, size=2, replace=False)
                mv = (1, int(a), int(b))
            cand = state.copy()
            if mv[0] == 0:
                cand[mv[1]] = mv[2]
            else:
                cand[mv[1]], cand[mv[2]] = state[mv[2]], state[mv[1]]
            got = model.score(*mv)
            if np.array_equal(cand, state):
                assert got is None
            else:
                assert got == inst.score(cand), (mv, state.tolist())
