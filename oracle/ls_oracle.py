"""oracle/ls_oracle.py — CPU ORACLE of the reference's host driver, TEST INFRASTRUCTURE ONLY.

An independent (Python) restatement of the reference's solver core and plug-in drivers, used
to check the product's C++ host mirror (constraint-solver_amd/csrc/host/) end to end:

  History, LocalSearch::execute          local-search/src/local_search.rs:105-250, 301-342
  IteratedLocalSearch, AcceptanceCriterion  local-search/src/iterated_local_search.rs:51-71, 130-202
  N-queens generator / proposer / perturbation   examples/nqueens/src/lib.rs:152-161, 177-255, 285-320
  scheduling generator / proposer / perturbation examples/employee-scheduling/src/lib.rs:400-420, 444-490, 582-612
  seed hashing + generators                examples/nqueens/src/main.rs:28-39, 66

Here the neighbourhood pipeline (local_search.rs:315-325) is the LITERAL one: enumerate, drop
tabu via the History's set, score every candidate with the C oracle, take(window), pick the
minimum by (score, solution) with first-of-equals.

The rand 0.8.5 / rand_chacha 0.3.1 / blake2 0.10.4 crates are not in /root/reference; their
algorithms are restated from the published sources.  PARITY UNPINNED
against the Rust binaries' random streams (no Rust toolchain); Blake2b is hashlib's and the
ChaCha20 block function is checked against RFC 7539 in the tests.  The reference's own
`repeatable` test (examples/nqueens/src/main.rs:156-200: n = 8, seeds "42".."49" reach score 0,
deterministically) is reproduced on this oracle.
"""
from __future__ import annotations

import hashlib
import math
import struct
from collections import deque

import numpy as np

import oracle as orc

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


def hash_str(seed: str) -> bytes:
    """main.rs:28-33: Blake2b with a 32-byte digest."""
    return hashlib.blake2b(seed.encode(), digest_size=32).digest()


def chacha20_block(key_words, w12, w13, w14, w15):
    """One ChaCha20 block (djb layout: words 12-13 counter, 14-15 stream id) -> 16 u32."""
    def rotl(v, n):
        return ((v << n) & M32) | (v >> (32 - n))
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [w12, w13, w14, w15]
    x = list(s)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32; x[b] = rotl(x[b] ^ x[c], 7)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(x[i] + s[i]) & M32 for i in range(16)]


class ChaCha20Rng:
    """rand_chacha::ChaCha20Rng::from_seed over BlockRng (64-word buffer = 4 blocks)."""

    def __init__(self, seed32: bytes):
        self.key = [int.from_bytes(seed32[4 * i:4 * i + 4], "little") for i in range(8)]
        self.counter = 0
        self.buf = [0] * 64
        self.index = 64

    def clone(self):
        c = ChaCha20Rng.__new__(ChaCha20Rng)
        c.key, c.counter, c.buf, c.index = list(self.key), self.counter, list(self.buf), self.index
        return c

    def _generate(self, idx):
        # vectorised over the four blocks of one refill
        k = np.array(self.key, dtype=np.uint32)
        ctr = np.arange(self.counter, self.counter + 4, dtype=np.uint64)
        s = np.zeros((16, 4), dtype=np.uint32)
        s[0], s[1], s[2], s[3] = 0x61707865, 0x3320646E, 0x79622D32, 0x6B206574
        for i in range(8):
            s[4 + i] = k[i]
        s[12] = (ctr & np.uint64(M32)).astype(np.uint32)
        s[13] = (ctr >> np.uint64(32)).astype(np.uint32)
        x = s.copy()

        def rotl(v, n):
            return (v << np.uint32(n)) | (v >> np.uint32(32 - n))

        def qr(a, b, c, d):
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16)
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12)
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8)
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7)
        with np.errstate(over="ignore"):
            for _ in range(10):
                qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
                qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
            x += s
        self.buf = [int(v) for v in x.T.reshape(-1)]      # block-major
        self.counter += 4
        self.index = idx

    def next_u32(self):
        if self.index >= 64:
            self._generate(0)
        v = self.buf[self.index]
        self.index += 1
        return v

    def next_u64(self):
        if self.index < 63:
            lo, hi = self.buf[self.index], self.buf[self.index + 1]
            self.index += 2
            return (hi << 32) | lo
        if self.index >= 64:
            self._generate(2)
            return (self.buf[1] << 32) | self.buf[0]
        x = self.buf[63]
        self._generate(1)
        return (self.buf[0] << 32) | x

    def gen_f64(self):
        return (self.next_u64() >> 11) * (1.0 / (1 << 53))


def _lz(v, bits):
    return bits - v.bit_length()


def gen_range_incl(rng, low, high, bits):
    """UniformInt::sample_single_inclusive for a `bits`-wide unsigned draw."""
    mask = (1 << bits) - 1
    rng_range = (high - low + 1) & mask
    draw = rng.next_u32 if bits == 32 else rng.next_u64
    if rng_range == 0:
        return draw()
    zone = ((rng_range << _lz(rng_range, bits)) & mask) - 1
    while True:
        m = draw() * rng_range
        if (m & mask) <= zone:
            return low + (m >> bits)


def uniform_below(rng, high, bits):
    """Uniform::new(0, high).sample — the WeightedIndex draw (different zone)."""
    mask = (1 << bits) - 1
    draw = rng.next_u32 if bits == 32 else rng.next_u64
    ints_to_reject = (mask - high + 1) % high
    zone = mask - ints_to_reject
    while True:
        m = draw() * high
        if (m & mask) <= zone:
            return m >> bits


def gen_index(rng, ubound):
    return gen_range_incl(rng, 0, ubound - 1, 32 if ubound <= M32 else 64)


def shuffle(rng, v):
    for i in range(len(v) - 1, 0, -1):
        j = gen_index(rng, i + 1)
        v[i], v[j] = v[j], v[i]


def choose_weighted_index(rng, weights, bits):
    cumulative, total = [], 0
    for i, w in enumerate(weights):
        if i + 1 < len(weights):
            cumulative.append(total + w)
        total += w
    chosen = uniform_below(rng, total, bits)
    idx = 0
    while idx < len(cumulative) and cumulative[idx] <= chosen:
        idx += 1
    return idx


def index_sample(rng, length, amount):
    assert amount <= length

    def floyd():
        fs = amount < 50
        idx = []
        for j in range(length - amount, length):
            t = gen_range_incl(rng, 0, j, 32)
            if fs:
                if t in idx:
                    idx.insert(idx.index(t), j)
                    continue
            elif t in idx:
                idx.append(j)
                continue
            idx.append(t)
        if not fs:
            for i in range(amount - 1, 0, -1):
                k = gen_range_incl(rng, 0, i, 32)
                idx[i], idx[k] = idx[k], idx[i]
        return idx

    def inplace():
        idx = list(range(length))
        for i in range(amount):
            j = gen_range_incl(rng, i, length - 1, 32)
            idx[i], idx[j] = idx[j], idx[i]
        return idx[:amount]

    def rejection():
        cache, idx = set(), []
        for _ in range(amount):
            pos = uniform_below(rng, length, 32)
            while pos in cache:
                pos = uniform_below(rng, length, 32)
            cache.add(pos)
            idx.append(pos)
        return idx

    f32 = np.float32
    j = 0 if length < 500_000 else 1
    if amount < 163:
        c = [[f32(1.6), f32(8.0) / f32(45.0)], [f32(10.0), f32(70.0) / f32(9.0)]]
        amount_fp = f32(amount)
        m4 = c[0][j] * amount_fp
        if amount > 11 and f32(length) < (c[1][j] + m4) * amount_fp:
            return inplace()
        return floyd()
    c2 = [f32(270.0), f32(330.0) / f32(9.0)]
    if f32(length) < c2[j] * f32(amount):
        return inplace()
    return rejection()


def index_sample_weighted(rng, weights, amount):
    """Efraimidis-Spirakis: `amount` largest keys u^(1/w), descending."""
    if amount == 0:
        return []
    keyed = []
    for i, w in enumerate(weights):
        assert w >= 0.0
        keyed.append((math.pow(rng.gen_f64(), 1.0 / w), i))
    # BinaryHeap pops the maximum key; ties (probability ~0) are not modelled
    keyed.sort(key=lambda t: -t[0])
    return [i for _, i in keyed[:amount]]


# ------------------------------------------------------------------ solver core
class History:
    def __init__(self, best_cap, all_cap, expiry):
        self.best = []            # sorted list of (score, solution) — BTreeSet
        self.best_cap, self.all_cap, self.expiry = best_cap, all_cap, expiry
        self.all = deque()        # front = newest
        self.lookup = set()
        self.iteration_count = 0

    def seen_solution(self, ss):
        self.iteration_count += 1
        while self.all:                                             # _pop_solution_for_age
            sol, it = self.all[-1]
            if it + self.expiry >= self.iteration_count:
                self.lookup.discard(sol)
                self.all.pop()
                continue
            break
        if ss[1] in self.lookup:
            return
        while len(self.all) > self.all_cap:                         # _pop_solution_for_size
            sol, _ = self.all.pop()
            self.lookup.discard(sol)
        self.all.appendleft((ss[1], self.iteration_count))
        self.lookup.add(ss[1])

    def is_solution_tabu(self, sol):
        return sol in self.lookup

    def is_best_solution(self, ss):
        return ss in self.best

    def local_search_chose_solution(self, ss):
        if len(self.best) < self.best_cap:
            if ss not in self.best:
                self.best.append(ss); self.best.sort()
            return
        worst = self.best[-1]
        if ss[0] <= worst[0]:
            self.best.remove(worst)
            if ss not in self.best:
                self.best.append(ss); self.best.sort()

    def get_random_best_solution(self, rng):
        if not self.best:
            return None
        return self.best[gen_index(rng, len(self.best))]

    def get_best(self):
        return self.best[0] if self.best else None


class LocalSearch:
    def __init__(self, problem, max_iterations, window, best_cap, all_cap, expiry, rng):
        self.p, self.max_iterations, self.window = problem, max_iterations, window
        self.history = History(best_cap, all_cap, expiry)
        self.rng = rng
        self.ls_iterations = 0

    def execute(self, start, allow_no_improvement_for):
        current = (self.p.score(start), start)
        best = current
        no_imp = 0
        for _ in range(self.max_iterations):
            self.history.seen_solution(current)
            if self.p.is_best(current[0]):
                return current
            self.ls_iterations += 1
            nb = self.p.neighbourhood_first(current, self.rng, self.window, self.history)
            if nb is not None:
                if nb[0] < current[0]:
                    best = nb
                    no_imp = 0
                else:
                    no_imp += 1
                    if no_imp >= allow_no_improvement_for:
                        break
                current = nb
            else:
                break
        return best


class IteratedLocalSearch:
    def __init__(self, problem, params, seed):
        self.p = problem
        seed32 = seed if isinstance(seed, (bytes, bytearray)) else hash_str(seed)   # 32 raw bytes, or the examples' seed string
        self.local_search = LocalSearch(problem, params["local_search_max_iterations"], params["window_size"],
                                        params["best_solutions_capacity"], params["all_solutions_capacity"],
                                        params["all_solution_iteration_expiry"], ChaCha20Rng(seed32))
        self.history = History(params["best_solutions_capacity"], params["all_solutions_capacity"],
                               params["all_solution_iteration_expiry"])
        self.rng = ChaCha20Rng(seed32)
        self.iteration = 0
        self.max_iterations = params["iterated_local_search_max_iterations"]
        self.max_allow = params["max_allow_no_improvement_for"]
        s = problem.generate_initial_solution(self.rng)
        self.current = (problem.score(s), s)

    def is_finished(self):
        return self.iteration >= self.max_iterations

    def get_best_solution(self):
        return self.history.get_best()

    def execute_round(self):
        self.iteration += 1
        best = self.history.get_best()
        if best is not None and self.p.is_best(best[0]):
            return
        if self.iteration > 0 and self.iteration % 50 == 0:
            s = self.p.generate_initial_solution(self.rng)
            self.current = (self.p.score(s), s)
        perturbed = self.p.propose_new_starting_solution(self.current, self.history, self.rng)
        new = self.local_search.execute(perturbed, self.max_allow)
        self.history.local_search_chose_solution(new)
        # AcceptanceCriterion::choose
        rb = self.history.get_random_best_solution(self.rng)
        choices = [self.current, new] + ([rb] if rb is not None else [])
        weights = [1, 5] + ([1] if rb is not None else [])
        self.current = choices[choose_weighted_index(self.rng, weights, 32)]


# -------------------------------------------------------------------- plug-ins
class NQueens:
    def __init__(self, n, formulation=None):
        self.n = n
        self.formulation = formulation or ("literal" if n <= 64 else "delta")

    @staticmethod
    def is_best(score):
        return score == 0

    def score(self, sol):
        return orc.nq_score(sol)

    def generate_initial_solution(self, rng):
        rows = list(range(self.n))
        shuffle(rng, rows)
        return tuple(rows)

    def propose_columns(self, start, rng):
        cs = orc.nq_col_scores(start)
        conflict = [(c, int(s)) for c, s in enumerate(cs) if s != 0]
        if not conflict:
            return None
        amount = min(max(len(start) // 20, 1), len(conflict))
        picked = index_sample_weighted(rng, [s + 1e-4 for _, s in conflict], amount)
        cols = [conflict[i][0] for i in picked]
        num_cols = gen_range_incl(rng, 1, len(cols), 64)
        return [cols[i] for i in index_sample(rng, len(cols), num_cols)]

    def neighbourhood_first(self, current, rng, window, history):
        cols = self.propose_columns(current[1], rng)
        if cols is None:
            return None
        if history.lookup == {current[1]}:      # every shipped configuration: the tabu set is {current}
            res = orc.nq_neighbourhood(current[1], cols, window, self.formulation, want_scores=False)
        else:                                   # general History (all_solution_iteration_expiry == 0): literal HashSet filter
            assert current[1] in history.lookup
            res = orc.nq_neighbourhood_tabu(current[1], cols, window, sorted(history.lookup), want_scores=False)
        if res["best_index"] < 0:
            return None
        b = list(current[1])
        b[cols[res["best_index"] // self.n]] = res["best_index"] % self.n
        return (res["best_score"], tuple(b))

    def propose_new_starting_solution(self, current, history, rng):
        strategy = choose_weighted_index(rng, [100, 10], 64)
        new = list(current[1])
        if strategy == 1:
            return tuple(new)
        n = len(new)
        rows = list(range(n))
        shuffle(rng, rows)
        clamp = lambda x: min(max(x, 1), n)
        if history.is_best_solution(current):
            k = gen_range_incl(rng, 1, clamp(n // 20), 64)
        else:
            k = gen_range_incl(rng, 1, clamp(n // 2), 64)
        for i in rows[:k]:
            new[i] = gen_range_incl(rng, 0, n - 1, 64)
        return tuple(new)


class Scheduling:
    def __init__(self, start, end, n_employees, holidays=None):
        self.inst = orc.Sched(start, end, holidays or {})
        self.D = self.inst.n_scored_days
        self.S = self.D + 1
        self.E = n_employees

    @staticmethod
    def is_best(score):
        return score == (0, 0)

    def score(self, sol):
        return tuple(self.inst.score(sol))

    def generate_initial_solution(self, rng):
        return tuple(gen_index(rng, self.E) for _ in range(self.S))

    def neighbourhood_first(self, current, rng, window, history):
        assert current[1] in history.lookup        # seen_solution always leaves the current solution in the set
        stream = rng.clone()                       # lib.rs:488 — the proposer never advances LocalSearch's rng
        e = current[1]
        best = None
        scored = 0
        while scored < window:
            t = choose_weighted_index(stream, [1, 4], 64)
            cand = list(e)
            if t == 0:
                day = gen_index(stream, self.D)
                cand[day] = gen_index(stream, self.E)
            else:
                a, b = index_sample(stream, self.D, 2)
                cand[a], cand[b] = e[b], e[a]
            cand = tuple(cand)
            if cand in history.lookup:
                continue
            ss = (tuple(self.inst.score(cand)), cand)
            scored += 1
            if best is None or ss < best:
                best = ss
        return best

    def propose_new_starting_solution(self, current, history, rng):
        strategy = choose_weighted_index(rng, [10, 100], 64)
        new = list(current[1])
        if strategy == 0:
            return tuple(new)
        total = len(new)
        clamp = lambda x: min(max(x, 1), total)
        if history.is_best_solution(current):
            k = gen_range_incl(rng, 1, clamp(total // 20), 64)
        else:
            k = gen_range_incl(rng, 1, clamp(total // 2), 64)
        idx = list(range(total))
        shuffle(rng, idx)
        for i in idx[:k]:
            new[i] = gen_index(rng, self.E)
        return tuple(new)


NQUEENS_MAIN_PARAMS = dict(  # examples/nqueens/src/main.rs:129-135 (window = 5 * board_size)
    local_search_max_iterations=10_000, best_solutions_capacity=32, all_solutions_capacity=100_000,
    all_solution_iteration_expiry=10_000, iterated_local_search_max_iterations=10_000,
    max_allow_no_improvement_for=5)

SCHED_MAIN_PARAMS = dict(    # examples/employee-scheduling/src/main.rs:24-31
    local_search_max_iterations=1_000, window_size=100, best_solutions_capacity=64, all_solutions_capacity=100_000,
    all_solution_iteration_expiry=1_000, iterated_local_search_max_iterations=250, max_allow_no_improvement_for=20)


# ------------------------------------------------------------------ Ackley
# The reference's OWN tests of the generic loop (local-search/src/local_search.rs:351-440,
# iterated_local_search.rs:205-321) drive LocalSearch / IteratedLocalSearch with a third plug-in, the f64 Ackley
# function (local-search/src/ackley.rs, math-util/src/ackley.rs).  It is restated here ONLY so that those tests'
# assertions can be run against this oracle's History / LocalSearch / IteratedLocalSearch (tests/test_oracle_driver.py);
# the product does not build it.  The RNG draws (rand_core seed_from_u64, UniformFloat, Normal)
# follow the published algorithms from memory -- stream equality with the Rust binaries is unverified, and the
# reference's assertions (improves / does not move / reaches 1e-2) do not depend on it.
def seed_from_u64(state: int) -> bytes:
    """rand_core 0.6 SeedableRng::seed_from_u64: PCG32 output words fill the 32-byte seed."""
    out = b""
    for _ in range(8):
        state = (state * 6364136223846793005 + 11634580027462260723) & M64
        xorshifted = (((state >> 18) ^ state) >> 27) & M32
        rot = state >> 59
        x = ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & M32
        out += x.to_bytes(4, "little")
    return out


def gen_range_f64(rng, low, high):
    """rand 0.8.5 UniformFloat<f64>::sample_single for low..high."""
    scale = high - low
    while True:
        bits = (rng.next_u64() >> 12) | (1023 << 52)          # a float in [1, 2)
        value1_2 = struct.unpack("<d", struct.pack("<Q", bits))[0]
        res = value1_2 * scale + (low - scale)
        if res < high:
            return res


def ackley_function(xs, a=20.0, b=0.2, c=2.0 * math.pi):
    """math-util/src/ackley.rs:19-33."""
    d = float(len(xs))
    square_sum = 0.0
    cosine_sum = 0.0
    for x in xs:
        square_sum += x * x
        cosine_sum += math.cos(c * x)
    fx = 0.0
    fx += -a * math.exp(-b * math.sqrt(square_sum / d))
    fx -= math.exp(cosine_sum / d)
    fx += a + math.e
    return fx


class Ackley:
    """AckleySolution / Score / ScoreCalculator / InitialSolutionGenerator / MoveProposer / Perturbation
    (local-search/src/ackley.rs:21-262) with the LITERAL neighbourhood pipeline of local_search.rs:315-325."""
    X_MIN, X_MAX = -32.768, 32.768

    def __init__(self, dimensions, min_move_size, max_move_size):
        self.d, self.lo, self.hi = dimensions, min_move_size, max_move_size

    @staticmethod
    def is_best(score):                                         # ackley.rs:36-41
        return abs(score - 0.0) <= 1e-2

    def score(self, sol):
        return ackley_function(sol)

    def generate_initial_solution(self, rng):                   # :96-105
        return tuple(gen_range_f64(rng, self.X_MIN, self.X_MAX) for _ in range(self.d))

    def neighbourhood_first(self, current, rng, window, history):
        dims = list(range(self.d))                              # iter_local_moves, :187-197
        shuffle(rng, dims)
        move = gen_range_f64(rng, self.lo, self.hi)
        scored = []
        for dim in dims:                                        # MoveIterator::next, :156-178: up, then down
            for delta in (move, -move):
                c = list(current[1])
                c[dim] = c[dim] + delta
                c = tuple(c)
                if history.is_solution_tabu(c):                 # local_search.rs:319
                    continue
                scored.append((self.score(c), c))               # :320
                if len(scored) == window:                       # :321
                    break
            if len(scored) == window:
                break
        if not scored:
            return None
        scored.sort()                                           # :323
        return scored[0]                                        # :325

    def propose_new_starting_solution(self, current, history, rng):   # ackley.rs:231-261
        strategy = choose_weighted_index(rng, [100, 10], 64)
        if strategy == 1:
            return current[1]
        new = list(current[1])
        dims = list(range(len(new)))
        shuffle(rng, dims)
        k = gen_range_incl(rng, 0, len(dims) - 1, 64)           # gen_range(0..len): may be 0
        for i in dims[:k]:
            # rand_distr::Normal (ziggurat) is not restated: Box-Muller on two uniform draws (stream differs)
            u1, u2 = max(rng.gen_f64(), 1e-300), rng.gen_f64()
            z = math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * math.pi * u2)
            new[i] = min(max(new[i] + z, self.X_MIN), self.X_MAX)
        return tuple(new)


def run(ils: IteratedLocalSearch, max_rounds=None, trace=None):
    rounds = 0
    while not ils.is_finished() and (max_rounds is None or rounds < max_rounds):
        ils.execute_round()
        rounds += 1
        if trace is not None:
            b = ils.get_best_solution()
            trace.append((ils.current[0], None if b is None else b[0]))
        b = ils.get_best_solution()
        if max_rounds is None and b is not None and ils.p.is_best(b[0]):
            break       # the reference keeps spinning through no-op rounds; nothing changes any more
    return ils.get_best_solution()
