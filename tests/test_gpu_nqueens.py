"""GPU parity tests (through the C ABI) for the N-queens neighbourhood
pipeline: HIP path vs the CPU oracle on the same seeded inputs — bit-exact on
every per-candidate score, on n_scored and on the selected candidate."""
import json
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu

ALGOS = ["scan", "table"]   # "auto" (the default) resolves to the TABLE family


@pytest.fixture(scope="module")
def ctx():
    import ls_amd
    c = ls_amd.Context(0)
    yield c
    c.close()


def _board(rng, n, kind):
    return rng.permutation(n) if kind == "perm" else rng.integers(0, n, size=n)


def _check(ctx, algo, rows, cols, window, formulation):
    ctx.set_algo(algo)
    exp = orc.nq_neighbourhood(rows, cols, window, formulation)
    cur = orc.nq_score(rows)
    got, scores = ctx.nqueens_eval(rows, cur, cols, window, want_scores=True)
    assert got["n_scored"] == exp["n_scored"]
    assert got["best_index"] == exp["best_index"]
    if exp["best_index"] >= 0:
        assert got["best_score"] == exp["best_score"]
        n = len(rows)
        assert got["best_col"] == cols[exp["best_index"] // n]
        assert got["best_value"] == exp["best_index"] % n
    assert np.array_equal(scores, exp["scores"])
    return got


def test_reference_kats_col_scores(ctx, golden_dir):
    with open(os.path.join(golden_dir, "nqueens_kats.json")) as f:
        g = json.load(f)
    for group in ("reference", "derived"):
        for kat in g[group]:
            cs, tot = ctx.nqueens_col_scores(kat["rows"])
            assert cs[0].tolist() == kat["col_scores"]
            assert tot[0] == kat["score"]


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("n", [1, 2, 3, 4, 8, 33, 63, 64, 65, 100])
def test_small_vs_literal(ctx, algo, n):
    rng = np.random.default_rng(n)
    for kind in ("perm", "uniform"):
        rows = _board(rng, n, kind)
        for n_cols in sorted({1, min(n, 3), n}):
            cols = rng.permutation(n)[:n_cols]
            for window in (1, n - 1, 5 * n, 10 ** 12):
                if window < 1:
                    continue
                _check(ctx, algo, rows, cols, window, "literal")


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("n", [127, 128, 129, 255, 256, 257, 500, 512, 513, 1000, 1024, 1025, 2047, 2048, 2049])
def test_register_tiers_vs_delta(ctx, algo, n):
    # every register tier of the scan kernel (n <= 64*NR) incl. its edges
    rng = np.random.default_rng(n)
    rows = _board(rng, n, "uniform" if n % 2 else "perm")
    cols = rng.permutation(n)[:7]
    _check(ctx, algo, rows, cols, 5 * n, "delta")
    _check(ctx, algo, rows, cols[:2], 10 ** 9, "table")


@pytest.mark.parametrize("algo", ALGOS)
def test_nq8_example_shape(ctx, algo):
    # config 1 shape: n = 8, window = 40, one column (amount = clamp(8/20, 1, L) = 1)
    rng = np.random.default_rng(42)
    for _ in range(20):
        rows = _board(rng, 8, "perm")
        _check(ctx, algo, rows, [int(rng.integers(0, 8))], 40, "literal")


@pytest.mark.parametrize("algo", ALGOS)
def test_nq1024_reference_shaped_launch(ctx, algo):
    # config 2 shape: n = 1024, window = 5n = 5120, up to 51 proposed columns
    rng = np.random.default_rng(1024)
    rows = _board(rng, 1024, "perm")
    cols = rng.permutation(1024)[:51]
    got = _check(ctx, algo, rows, cols, 5120, "delta")
    assert got["n_scored"] == 5120


@pytest.mark.parametrize("algo", ALGOS)
def test_nq4096_reference_shaped_launch(ctx, algo):
    # config 4 per-iteration shape: n = 4096, window = 20480, up to 204 columns
    rng = np.random.default_rng(4096)
    rows = _board(rng, 4096, "uniform")
    cols = rng.permutation(4096)[:204]
    got = _check(ctx, algo, rows, cols, 20480, "table")
    assert got["n_scored"] == 20480
    # independent check of the winner with the literal O(n^2) scorer
    b = rows.copy()
    b[got["best_col"]] = got["best_value"]
    assert orc.nq_score(b) == got["best_score"]


@pytest.mark.parametrize("algo", ALGOS)
def test_nq4096_full_neighbourhood(ctx, algo):
    # BASELINE.json full size, kernel-throughput shape: all n columns, window = n(n-1)
    n = 4096
    rng = np.random.default_rng(7)
    rows = _board(rng, n, "uniform")
    cols = np.arange(n)
    ctx.set_algo(algo)
    exp = orc.nq_neighbourhood(rows, cols, n * (n - 1), "table")
    got, scores = ctx.nqueens_eval(rows, orc.nq_score(rows), cols, n * (n - 1), want_scores=True)
    assert got["n_scored"] == n * (n - 1) == exp["n_scored"]
    assert got["best_index"] == exp["best_index"] and got["best_score"] == exp["best_score"]
    assert np.array_equal(scores, exp["scores"])
    # size-independent properties
    valid = scores != orc.SKIPPED
    assert valid.sum() == n * (n - 1)
    assert scores[valid].min() == got["best_score"]
    assert (scores[valid] % 2 == 0).all()          # every attacking pair counts twice
    b = rows.copy()
    b[got["best_col"]] = got["best_value"]
    assert orc.nq_score(b) == got["best_score"]


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("n", [4097, 5000, 8192, 20000, 32768])
def test_large_n_lds_path(ctx, algo, n):
    rng = np.random.default_rng(n)
    rows = _board(rng, n, "uniform")
    cols = rng.permutation(n)[:3]
    _check(ctx, algo, rows, cols, 2 * n + 17, "table")


@pytest.mark.parametrize("algo", ALGOS)
def test_batched_restarts_ragged(ctx, algo):
    ctx.set_algo(algo)
    rng = np.random.default_rng(99)
    n, R, max_cols = 200, 11, 9
    rows = np.stack([_board(rng, n, "uniform" if r % 2 else "perm") for r in range(R)])
    n_cols = rng.integers(0, max_cols + 1, size=R)
    n_cols[0], n_cols[1] = 0, max_cols                      # an empty and a full one
    cols = np.stack([rng.permutation(n)[:max_cols] for _ in range(R)])
    cur = [orc.nq_score(rows[r]) for r in range(R)]
    window = 3 * (n - 1) + 50                               # truncates inside the 4th column
    out, scores = ctx.nqueens_eval_batch(rows, cur, cols, n_cols, window, want_scores=True)
    for r in range(R):
        exp = orc.nq_neighbourhood(rows[r], cols[r, :n_cols[r]], window, "delta")
        assert out[r]["n_scored"] == exp["n_scored"]
        assert out[r]["best_index"] == exp["best_index"]
        if exp["best_index"] >= 0:
            assert out[r]["best_score"] == exp["best_score"]
        k = n_cols[r] * n
        assert np.array_equal(scores[r, :k], exp["scores"])
        assert (scores[r, k:] == orc.SKIPPED).all()


@pytest.mark.parametrize("algo", ALGOS)
def test_ties_and_duplicates(ctx, algo):
    # all queens on one row: massive score ties -> the lexicographic tie-break decides
    n = 96
    rows = np.full(n, 5)
    _check(ctx, algo, rows, [90, 3, 47], 10 ** 9, "literal")
    # duplicated proposal column: identical candidates, first position wins
    _check(ctx, algo, np.arange(n)[::-1].copy(), [10, 10, 4], 10 ** 9, "literal")


@pytest.mark.parametrize("algo", ALGOS)
def test_empty_neighbourhoods(ctx, algo):
    ctx.set_algo(algo)
    got, _ = ctx.nqueens_eval([0, 2, 1], 2, [], 10)
    assert got["n_scored"] == 0 and got["best_index"] == -1
    got, _ = ctx.nqueens_eval([0], 0, [0], 10)              # the only candidate is the no-op
    assert got["n_scored"] == 0 and got["best_index"] == -1
    got, _ = ctx.nqueens_eval([0, 2, 1], 2, [1], 0)         # window 0
    assert got["n_scored"] == 0 and got["best_index"] == -1


def test_col_scores_match_reference_loop(ctx):
    rng = np.random.default_rng(5)
    for n in (1, 7, 64, 1000, 4096):
        rows = np.stack([_board(rng, n, "perm"), _board(rng, n, "uniform")])
        cs, tot = ctx.nqueens_col_scores(rows)
        for r in range(2):
            exp = orc.nq_col_scores(rows[r])                 # literal lib.rs:74-87
            assert np.array_equal(cs[r], exp)
            assert tot[r] == exp.sum()


def test_argument_errors(ctx):
    import ls_amd
    with pytest.raises(ls_amd.LsError) as ei:
        ctx.nqueens_eval([0, 5], 0, [0], 10)                 # row value outside [0, n)
    assert ei.value.code == ls_amd._lib.LS_ERR_INVALID
    with pytest.raises(ls_amd.LsError):
        ctx.nqueens_eval([0, 1], 0, [2], 10)                 # column outside [0, n)
    with pytest.raises(ls_amd.LsError):
        ctx.nqueens_eval([0, 1], 0, [0], -1)                 # negative window
    with pytest.raises(ls_amd.LsError) as ei:
        ctx.nqueens_eval(np.zeros(40000, dtype=np.int32), 0, [0], 1)
    assert ei.value.code == ls_amd._lib.LS_ERR_UNSUPPORTED


def test_device_pointer_entry_matches_host_entry(ctx):
    torch = pytest.importorskip("torch")
    import ctypes as C
    import ls_amd
    rng = np.random.default_rng(11)
    n, R, mc = 300, 4, 5
    rows = np.stack([_board(rng, n, "uniform") for _ in range(R)]).astype(np.int32)
    cols = np.stack([rng.permutation(n)[:mc] for _ in range(R)]).astype(np.int32)
    ncols = np.full(R, mc, dtype=np.int32)
    cur = np.array([orc.nq_score(rows[r]) for r in range(R)], dtype=np.int64)
    ctx.set_algo("scan")
    host, _ = ctx.nqueens_eval_batch(rows, cur, cols, ncols, 777)
    dev = torch.device("cuda:0")
    t = [torch.from_numpy(x).to(dev) for x in (rows, cur, cols, ncols)]
    res = torch.zeros(R * 4, dtype=torch.int64, device=dev)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.nqueens_eval_batch_dev(R, n, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), mc,
                                   777, res.data_ptr())
        torch.cuda.synchronize()
    finally:
        ctx.set_stream(None)
    raw = res.cpu().numpy().reshape(R, 4)
    for r in range(R):
        assert raw[r, 0] == host[r]["n_scored"]
        assert raw[r, 1] == host[r]["best_index"]
        assert raw[r, 2] == host[r]["best_score"]


@pytest.mark.parametrize("variant", ["1"])
def test_scan_formulations_agree(monkeypatch, variant):
    # the register-scan kernel has two inner-loop formulations (fp32 product test = default,
    # packed-u16 min3); both must be bit-identical to the oracle
    import ls_amd
    monkeypatch.setenv("LS_NQ_SCAN_VARIANT", variant)
    with ls_amd.Context(0, algo="scan") as c:
        rng = np.random.default_rng(int(variant))
        for n, n_cols in ((9, 9), (200, 7), (1000, 5), (4096, 6), (6000, 2)):
            rows = _board(rng, n, "uniform")
            cols = rng.permutation(n)[:n_cols]
            _check(c, "scan", rows, cols, 5 * n, "table")


def test_device_entry_is_graph_capturable(ctx):
    # the *_dev entry point does no allocation / synchronisation once its workspaces exist, so an LS
    # step can be captured into a hipGraph and replayed (north star: streams and graphs, no tracer)
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(21)
    n, R, mc = 512, 4, 6
    dev = torch.device("cuda:0")
    rows_np = np.stack([_board(rng, n, "uniform") for _ in range(R)]).astype(np.int32)
    cols_np = np.stack([rng.permutation(n)[:mc] for _ in range(R)]).astype(np.int32)
    cur_np = np.array([orc.nq_score(rows_np[r]) for r in range(R)], dtype=np.int64)
    rows, cols = torch.from_numpy(rows_np).to(dev), torch.from_numpy(cols_np).to(dev)
    cur, ncols = torch.from_numpy(cur_np).to(dev), torch.full((R,), mc, dtype=torch.int32, device=dev)
    res = torch.zeros((R, 4), dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_algo("scan")
    ctx.set_stream(stream.cuda_stream)
    try:
        def launch():
            ctx.nqueens_eval_batch_dev(R, n, rows.data_ptr(), cur.data_ptr(), cols.data_ptr(), ncols.data_ptr(), mc,
                                       5 * n, res.data_ptr())
        with torch.cuda.stream(stream):
            launch()                                   # warm-up: allocates the context's workspaces
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            launch()
        for trial in range(3):                         # replay on changing inputs
            rows_np = np.stack([_board(rng, n, "perm") for _ in range(R)]).astype(np.int32)
            cur_np = np.array([orc.nq_score(rows_np[r]) for r in range(R)], dtype=np.int64)
            rows.copy_(torch.from_numpy(rows_np)); cur.copy_(torch.from_numpy(cur_np))
            res.zero_()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            out = res.cpu().numpy()
            for r in range(R):
                exp = orc.nq_neighbourhood(rows_np[r], cols_np[r], 5 * n, "table", want_scores=False)
                assert out[r, 0] == exp["n_scored"] and out[r, 1] == exp["best_index"] and out[r, 2] == exp["best_score"]
    finally:
        ctx.set_stream(None)


def test_eval_next_returns_winner_column_scores(ctx):
    # one round trip: neighbourhood result + get_col_scores of the winner board (the proposer's next input)
    rng = np.random.default_rng(31)
    n, R, mc = 300, 5, 4
    rows = np.stack([_board(rng, n, "uniform") for _ in range(R)])
    cols = np.stack([rng.permutation(n)[:mc] for _ in range(R)])
    n_cols = np.array([mc, mc, 0, 1, mc])
    cur = [orc.nq_score(rows[r]) for r in range(R)]
    for algo in ALGOS:
        ctx.set_algo(algo)
        out, nxt = ctx.nqueens_eval_batch_next(rows, cur, cols, n_cols, 2 * n)
        ref, _ = ctx.nqueens_eval_batch(rows, cur, cols, n_cols, 2 * n)
        assert out == ref
        for r in range(R):
            b = rows[r].copy()
            if out[r]["best_index"] >= 0:
                b[out[r]["best_col"]] = out[r]["best_value"]
            exp = orc.nq_col_scores(b)                      # literal lib.rs:74-87
            assert np.array_equal(nxt[r], exp)
            if out[r]["best_index"] >= 0:
                assert exp.sum() == out[r]["best_score"]
    ctx.set_algo("scan")


def _expect_step(rows, cols, n_cols, window):
    """Oracle view of one LS step of every restart: neighbourhood result, winner board, its column scores."""
    out = []
    for r in range(len(rows)):
        c = cols[r][:n_cols[r]]
        exp = orc.nq_neighbourhood(rows[r], c, window, "table", want_scores=False) if len(c) else \
            dict(n_scored=0, best_index=-1, best_score=0)
        b = np.array(rows[r]).copy()
        if exp["best_index"] >= 0:
            b[c[exp["best_index"] // len(b)]] = exp["best_index"] % len(b)
        out.append((exp, b, orc.nq_col_scores(b)))
    return out


@pytest.mark.parametrize("mode", ["4", "3"])
def test_fused_step_kernel_matches_oracle(monkeypatch, mode):
    # the solver's per-iteration launch: one block per restart does tables + argmin + move + column scores.
    # LS_NQ_STEP: 4 = no launch: one slot of the context's resident server per restart, all requests in flight at once (default),
    # 3 = zero-copy + resident boards + 16-bit column scores + polled completion word,
    # 2 = zero-copy on pinned host memory + stream sync, 1 = through staging copies, 0 = multi-launch path
    import ls_amd
    monkeypatch.setenv("LS_NQ_STEP", mode)
    rng = np.random.default_rng(100 + int(mode))
    with ls_amd.Context(0) as c:                      # default algo: auto
        for n, R, mc, window in ((2, 3, 2, 10), (3, 2, 3, 4), (8, 4, 1, 40), (64, 5, 3, 64), (300, 6, 9, 1500),
                                 (1024, 8, 51, 5120), (64, 70, 3, 64), (4096, 8, 204, 20480), (4096, 2, 3, 10 ** 9), (8192, 2, 12, 40960),
                                 (8192, 1, 9, 65535), (500, 3, 4, 1)):
            kind = "uniform" if n % 2 == 0 else "perm"
            rows = np.stack([_board(rng, n, kind) for _ in range(R)])
            cols = np.stack([rng.permutation(n)[:mc] if mc <= n else rng.integers(0, n, size=mc) for _ in range(R)])
            if mc >= 3:
                cols[0, 2] = cols[0, 0]               # a duplicated column: the earlier slot must win ties
            n_cols = np.full(R, mc)
            if R >= 3:
                n_cols[1], n_cols[2] = 0, 1           # ragged: an empty and a one-column restart
            cur = [orc.nq_score(rows[r]) for r in range(R)]
            out, nxt = c.nqueens_eval_batch_next(rows, cur, cols, n_cols, window)
            for r, (exp, board, cs) in enumerate(_expect_step(rows, cols, n_cols, window)):
                assert out[r]["n_scored"] == exp["n_scored"], (n, r)
                assert out[r]["best_index"] == exp["best_index"], (n, r)
                if exp["best_index"] >= 0:
                    assert out[r]["best_score"] == exp["best_score"] == cs.sum()
                    assert out[r]["best_col"] == cols[r][exp["best_index"] // n]
                    assert out[r]["best_value"] == exp["best_index"] % n
                assert np.array_equal(nxt[r], cs), (n, r)


def test_fused_step_resident_boards_follow_any_call_sequence(monkeypatch):
    # LS_NQ_STEP=3 keeps every restart's winner board on the device and uploads a board only when the caller's differs
    # from it.  Whatever the caller does next -- continue from the winner (local_search.rs:335), jump to an unrelated
    # board (perturbation, iterated_local_search.rs:192-194), change the batch size or the board size, repeat a call --
    # each call must equal a stateless evaluation.  LS_NQ_HOST_CS=1 (default): while a board stays resident the HOST
    # keeps its column scores (updated from each move in O(n)) and the kernel skips that pass; 0: the kernel writes
    # them every call.  LS_NQ_RESIDENT_TABS=1 (default): the count tables of a resident board stay on the device too and
    # every step patches them with its move; 0: rebuilt from the board every call.
    import ls_amd
    rng = np.random.default_rng(333)
    with ls_amd.Context(0) as c:
        def call(rows, cols, n_cols, window):
            cur = [int(orc.nq_col_scores_table(rows[r]).sum()) for r in range(len(rows))]
            out, nxt = c.nqueens_eval_batch_next(rows, cur, cols, n_cols, window)
            boards = []
            for r, (exp, board, cs) in enumerate(_expect_step(rows, cols, n_cols, window)):
                assert (out[r]["n_scored"], out[r]["best_index"]) == (exp["n_scored"], exp["best_index"]), r
                if exp["best_index"] >= 0:
                    assert out[r]["best_score"] == exp["best_score"]
                assert np.array_equal(nxt[r], cs), r
                boards.append(board)
            return np.stack(boards)

        for n, R, mc in ((300, 3, 5), (4096, 2, 204), (300, 5, 5), (64, 1, 3)):
            window = 5 * n
            rows = np.stack([_board(rng, n, "uniform") for _ in range(R)])
            for it in range(12):
                cols = np.stack([rng.permutation(n)[:mc] for _ in range(R)])
                n_cols = rng.integers(1, mc + 1, size=R)
                nxt = call(rows, cols, n_cols, window)
                what = it % 4
                if what == 0 or what == 1:
                    rows = nxt                                    # continue from the winners: nothing is uploaded
                elif what == 2:
                    rows = nxt.copy()
                    rows[rng.integers(0, R)] = _board(rng, n, "perm")   # one restart perturbed, the others continue
                else:
                    pass                                          # the same boards again (the device holds the winners!)
            # an empty neighbourhood leaves the resident board as it is
            call(rows, np.zeros((R, 1), dtype=np.int64), np.zeros(R, dtype=np.int64), window)
            call(rows, np.stack([rng.permutation(n)[:mc] for _ in range(R)]), np.full(R, mc), window)
            # a long run of continued calls: the host-kept column scores must not drift
            for it in range(40):
                rows = call(rows, np.stack([rng.permutation(n)[:mc] for _ in range(R)]), rng.integers(0, mc + 1, size=R), window)


def test_fused_step_device_entry(ctx):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(77)
    dev = torch.device("cuda:0")
    ctx.set_algo("auto")
    for n, R, mc, window in ((1024, 4, 51, 5120), (4096, 3, 204, 20480), (200, 2, 200, 10 ** 9)):
        rows = np.stack([_board(rng, n, "uniform") for _ in range(R)]).astype(np.int32)
        cols = np.stack([rng.permutation(n)[:mc] for _ in range(R)]).astype(np.int32)
        n_cols = np.full(R, mc, dtype=np.int32)
        cur = np.array([orc.nq_score(rows[r]) for r in range(R)], dtype=np.int64)
        t = [torch.from_numpy(x).to(dev) for x in (rows, cur, cols, n_cols)]
        res = torch.zeros((R, 4), dtype=torch.int64, device=dev)
        nrows = torch.zeros((R, n), dtype=torch.int32, device=dev)
        ncs = torch.zeros((R, n), dtype=torch.int64, device=dev)
        ntot = torch.zeros(R, dtype=torch.int64, device=dev)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        try:
            ctx.nqueens_eval_batch_next_dev(R, n, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), mc,
                                            window, res.data_ptr(), nrows.data_ptr(), ncs.data_ptr(), ntot.data_ptr())
            torch.cuda.synchronize()
        finally:
            ctx.set_stream(None)
        raw, nrows, ncs, ntot = res.cpu().numpy(), nrows.cpu().numpy(), ncs.cpu().numpy(), ntot.cpu().numpy()
        for r, (exp, board, cs) in enumerate(_expect_step(rows, cols, n_cols, window)):
            assert (raw[r, 0], raw[r, 1]) == (exp["n_scored"], exp["best_index"])
            assert raw[r, 2] == exp["best_score"]
            assert np.array_equal(nrows[r], board) and np.array_equal(ncs[r], cs) and ntot[r] == cs.sum()


def test_fused_step_device_entry_is_graph_capturable(ctx):
    # a whole LS step (argmin + move + next board + its column scores) is one kernel with no allocation or
    # synchronisation once warmed up: captured into a hipGraph, replayed as a chain board -> next board -> ...
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(88)
    n, R, mc = 1024, 3, 51
    dev = torch.device("cuda:0")
    rows_np = np.stack([_board(rng, n, "uniform") for _ in range(R)]).astype(np.int32)
    cols_np = np.stack([rng.permutation(n)[:mc] for _ in range(R)]).astype(np.int32)
    n_cols = np.full(R, mc, dtype=np.int32)
    cur_np = np.array([orc.nq_score(rows_np[r]) for r in range(R)], dtype=np.int64)
    rows, cols = torch.from_numpy(rows_np).to(dev), torch.from_numpy(cols_np).to(dev)
    cur, ncols = torch.from_numpy(cur_np).to(dev), torch.from_numpy(n_cols).to(dev)
    res = torch.zeros((R, 4), dtype=torch.int64, device=dev)
    nrows = torch.zeros((R, n), dtype=torch.int32, device=dev)
    ncs = torch.zeros((R, n), dtype=torch.int64, device=dev)
    ntot = torch.zeros(R, dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_algo("auto")
    ctx.set_stream(stream.cuda_stream)
    try:
        def launch():
            ctx.nqueens_eval_batch_next_dev(R, n, rows.data_ptr(), cur.data_ptr(), cols.data_ptr(), ncols.data_ptr(), mc,
                                            5 * n, res.data_ptr(), nrows.data_ptr(), ncs.data_ptr(), ntot.data_ptr())
        with torch.cuda.stream(stream):
            launch()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            launch()
        board, score = rows_np.copy(), cur_np.copy()
        for step in range(4):                       # feed every step's winner board back in
            rows.copy_(torch.from_numpy(board)); cur.copy_(torch.from_numpy(score))
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            out, nb, cs, tot = res.cpu().numpy(), nrows.cpu().numpy(), ncs.cpu().numpy(), ntot.cpu().numpy()
            for r, (exp, b, ecs) in enumerate(_expect_step(board, cols_np, n_cols, 5 * n)):
                assert (out[r, 0], out[r, 1], out[r, 2]) == (exp["n_scored"], exp["best_index"], exp["best_score"])
                assert np.array_equal(nb[r], b) and np.array_equal(cs[r], ecs) and tot[r] == ecs.sum() == exp["best_score"]
            board, score = nb.astype(np.int32).copy(), tot.copy()
    finally:
        ctx.set_stream(None)


# ---------------------------------------------------------------------------------------------
# The PRODUCTION (no score dump) instantiations at BASELINE size.  bench.py times
# nq_scan_regs_kernel<BoardF32<64>, false>; with want_scores=True the <.., true> sibling runs, which always
# fetches conf_old in a separate pass.  These cases run the no-score kernels and compare n_scored, best_index and
# best_score with the oracle's count-table formulation (local_search.rs:315-325).

def _check_batch_noscores(ctx, algo, rows, cols, n_cols, window):
    ctx.set_algo(algo)
    R, n = rows.shape
    cur = [orc.nq_score(rows[r]) if n <= 512 else int(orc.nq_col_scores_table(rows[r]).sum()) for r in range(R)]
    out, scores = ctx.nqueens_eval_batch(rows, cur, cols, n_cols, window, want_scores=False)
    assert scores is None
    for r in range(R):
        exp = orc.nq_neighbourhood(rows[r], cols[r, :n_cols[r]], window, "table", want_scores=False)
        got = out[r]
        assert got["n_scored"] == exp["n_scored"], (r, got, exp)
        assert got["best_index"] == exp["best_index"], (r, got, exp)
        if exp["best_index"] >= 0:
            assert got["best_score"] == exp["best_score"], (r, got, exp)
            assert got["best_col"] == cols[r, exp["best_index"] // n]
            assert got["best_value"] == exp["best_index"] % n
    return out


@pytest.mark.parametrize("algo", ALGOS)
def test_bench_workload_production_path(ctx, algo):
    # exactly bench.py's step: 8 restarts, uniform boards seeded by restart id, all n columns in order,
    # window = n(n-1), no score dump
    n, R = 4096, 8
    rows = np.stack([np.random.default_rng(r).integers(0, n, size=n) for r in range(R)])
    cols = np.tile(np.arange(n), (R, 1))
    out = _check_batch_noscores(ctx, algo, rows, cols, np.full(R, n), n * (n - 1))
    assert all(o["n_scored"] == n * (n - 1) for o in out)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("n,R,n_cols,window", [
    (4096, 1, 204, 20480),            # the reference-shaped launch of config 4 (window 5n cuts the 6th column)
    (4096, 1, 1000, 10 ** 12),        # wavefront shares of 2000 positions: most ranges start/stop inside a column
    (4096, 3, 333, 333 * 4095 - 7),   # grid not a multiple of the restart count; window cuts the last column
    (4095, 5, 77, 50000),             # padded lanes (n not a multiple of 64), window cut inside a column
    (3001, 2, 64, 10 ** 12),
    (2049, 7, 9, 5 * 2049),
])
def test_scan_production_path_large_n(ctx, algo, n, R, n_cols, window):
    rng = np.random.default_rng(n * 31 + R)
    rows = np.stack([_board(rng, n, "uniform" if r % 2 == 0 else "perm") for r in range(R)])
    cols = np.stack([rng.integers(0, n, size=n_cols) for _ in range(R)])     # duplicates allowed
    _check_batch_noscores(ctx, algo, rows, cols, np.full(R, n_cols), window)


@pytest.mark.parametrize("algo", ALGOS)
def test_scan_production_path_own_square_at_share_boundaries(ctx, algo):
    # rows[c] placed at the first / last value of a 64-candidate pass and of the board, and next to the positions
    # where a wavefront's share of the candidate list starts or stops: both ways nq_scan_column_regs obtains
    # conf_old (own square inside the range -> readlane; outside -> the extra one-step pass) must agree
    n = 4096
    rng = np.random.default_rng(2024)
    rows = rng.integers(0, n, size=n)
    special = [0, 1, 62, 63, 64, 65, 127, 128, 2047, 2048, 4031, 4032, 4094, 4095]
    cols = rng.permutation(n)[:len(special) * 3]
    for i, c in enumerate(cols):
        rows[c] = special[i % len(special)]
    for n_cols, window in ((len(cols), 10 ** 12), (len(cols), 20480), (5, 4 * 4095 + 63), (5, 4 * 4095 + 64),
                           (5, 4 * 4095 + 65), (17, 16 * 4095 + 1), (2, 4095 + 4094), (1, 1), (1, 4095)):
        _check_batch_noscores(ctx, algo, rows[None], cols[None, :n_cols], np.array([n_cols]), window)


@pytest.mark.parametrize("want_scores", [False, True])
@pytest.mark.parametrize("n,R,n_cols,window", [
    (2049, 128, 100, 10 ** 12),          # 16 wavefronts per restart, 6.25 columns each: a cut column, pairs, an odd one left
    (2049, 128, 100, 61 * 2048 + 17),    # the window ends inside a column in the middle of a wavefront's share
    (2049, 128, 100, 62 * 2048),         # ... and exactly at the end of a pair's first / second column
    (2049, 128, 100, 63 * 2048),
    (3001, 64, 99, 10 ** 12),            # padded lanes and a partial last pass (3001 = 46 * 64 + 57)
    (4095, 64, 64, 10 ** 12),            # whole columns per wavefront: 2 each, n not a multiple of 64
    (4096, 32, 67, 10 ** 12),
    (2048, 64, 40, 10 ** 12),            # the 32-register tier (1024 < n <= 2048)
    (1500, 128, 37, 30 * 1499 + 700),
    (1024, 128, 66, 10 ** 12),           # the 16-register tier (512 < n <= 1024)
    (700, 256, 50, 41 * 699),
    (5000, 256, 80, 10 ** 12),           # the LDS kernel (n > 4096): 32 wavefronts per restart, 2.5 columns each
    (4097, 256, 77, 70 * 4096 + 9),
    (8192, 256, 70, 10 ** 12),
])
def test_scan_two_column_walk(ctx, n, R, n_cols, window, want_scores):
    # nq_scan_column_pair_regs (512 < n <= 4096) / nq_scan_pair_lds (n > 4096): two whole list columns per pass wherever a wavefront's share and the
    # window hold both; everything else goes through the one-column walk.  Duplicated columns, queens sharing rows.
    ctx.set_algo("scan")
    rng = np.random.default_rng(n * 7 + R + (window % 1000))
    rows = np.stack([_board(rng, n, "uniform" if r % 3 else "perm") for r in range(R)])
    cols = np.stack([rng.integers(0, n, size=n_cols) for _ in range(R)])
    for r in range(0, R, 5):
        cols[r, 1::7] = cols[r, 0:-1:7][: len(cols[r, 1::7])]     # neighbours in the list that are the SAME column
    cur = [int(orc.nq_col_scores_table(rows[r]).sum()) for r in range(R)]
    nc = np.full(R, n_cols)
    nc[3] = 1
    nc[4] = 2
    nc[5] = 0
    out, scores = ctx.nqueens_eval_batch(rows, cur, cols, nc, window, want_scores=want_scores)
    for r in range(R):
        exp = orc.nq_neighbourhood(rows[r], cols[r, :nc[r]], window, "table", want_scores=want_scores)
        got = out[r]
        assert (got["n_scored"], got["best_index"]) == (exp["n_scored"], exp["best_index"]), (r, got, exp)
        if exp["best_index"] >= 0:
            assert got["best_score"] == exp["best_score"], (r, got, exp)
        if want_scores:
            k = nc[r] * n
            assert np.array_equal(scores[r, :k], exp["scores"]), r
            assert (scores[r, k:] == orc.SKIPPED).all()


@pytest.mark.parametrize("n,n_cols", [(8, 2), (33, 4), (200, 6), (1000, 9)])
def test_general_tabu_set_exclusion_list(ctx, n, n_cols):
    # History configurations whose tabu set holds more than the current board (all_solution_iteration_expiry == 0,
    # local_search.rs:155-199): ls_nqueens_eval_tabu takes the candidates that equal another tabu board as positions;
    # they are neither scored nor counted (later candidates move up in the window).  Oracle: the literal pipeline with a
    # whole-board HashSet filter.
    rng = np.random.default_rng(n + n_cols)
    for trial in range(6):
        rows = _board(rng, n, "uniform" if trial % 2 else "perm")
        cols = rng.permutation(n)[:n_cols]
        if trial >= 3 and n_cols >= 3:
            cols[2] = cols[0]                                       # duplicated column: both of its slots are affected
        tabu = [tuple(int(x) for x in rows)]
        for _ in range(int(rng.integers(1, 12))):                   # visited boards: most at distance 1 in a proposed column
            b = list(tabu[0])
            kind = rng.integers(0, 4)
            c = int(cols[rng.integers(0, n_cols)]) if kind < 3 else int(rng.integers(0, n))
            b[c] = int(rng.integers(0, n))
            if kind == 2:
                b[int(rng.integers(0, n))] = int(rng.integers(0, n))   # distance 2: can never equal a candidate
            tabu.append(tuple(b))
        base = tabu[0]
        positions = sorted({s * n + t[c] for t in tabu[1:] for s, c in enumerate(cols)
                            if sum(1 for j in range(n) if t[j] != base[j]) == 1 and t[c] != base[c]
                            and all(t[j] == base[j] for j in range(n) if j != c)})
        cur = orc.nq_score(rows)
        for window in (10 ** 9, n_cols * (n - 1) - 3, 2 * (n - 1) + 1, 5):
            exp = orc.nq_neighbourhood_tabu(rows, cols, window, tabu)
            got, scores = ctx.nqueens_eval_tabu(rows, cur, cols, window, positions, want_scores=True)
            assert got["n_scored"] == exp["n_scored"], (trial, window)
            assert got["best_index"] == exp["best_index"], (trial, window)
            if exp["best_index"] >= 0:
                assert got["best_score"] == exp["best_score"]
            assert np.array_equal(scores, exp["scores"]), (trial, window)
    import ls_amd
    with pytest.raises(ls_amd.LsError):
        ctx.nqueens_eval_tabu([1, 3, 0, 2], 0, [0], 10, [1])            # (col 0, value 1) is the no-op
    with pytest.raises(ls_amd.LsError):
        ctx.nqueens_eval_tabu([1, 3, 0, 2], 0, [0], 10, [2, 0])         # not ascending


@pytest.mark.parametrize("n", [512, 1024, 4096, 12288])
def test_table_quads_kernel_and_its_u16_fallback(ctx, n):
    # TABLE without a score dump at n = 0 (mod 256): four candidates per lane and look-up on 8-BIT counts (nq_table_quads_kernel).
    # Boards whose rows / diagonals hold at most 50 queens stay on it -- incl. exactly 50 on one row, every column residue mod 4
    # (shifted table copies), windows that cut a column, duplicated columns, a queen on the board's last rows (sentinel squares) --;
    # 51 on a row, a constant board and a full diagonal trip the per-block fallback to the 16-bit pairs.  Ragged batches mix both.
    rng = np.random.default_rng(n)
    def with_row(k):
        b = rng.integers(0, n, size=n)
        b[b == 7] = 8
        b[rng.permutation(n)[:k]] = 7          # exactly k queens on row 7
        return b
    boards = [rng.integers(0, n, size=n), rng.permutation(n), with_row(50), with_row(51), np.zeros(n, dtype=np.int64),
              np.arange(n), np.full(n, n - 1), (np.arange(n) * 5 + 3) % n]
    R = len(boards)
    rows = np.stack(boards)
    mc = 9
    cols = np.stack([np.concatenate([[0, 1, 2, 3], rng.permutation(n)[:mc - 4]]) for _ in range(R)])
    cols[1, 5] = cols[1, 4]                    # a duplicated column: the earlier slot wins
    cols[0, 4:8] = [n - 4, n - 3, n - 2, n - 1]
    for window in (10 ** 12, 4 * (n - 1) + n // 2, 3, (n - 1) * mc - 1):
        _check_batch_noscores(ctx, "table", rows, cols, np.array([mc, mc, mc, mc, mc, 5, 1, 0]), window)
    # the full neighbourhood of one board per launch (many blocks per restart, parts > 1)
    if n <= 4096:
        full = np.tile(np.arange(n), (2, 1))
        _check_batch_noscores(ctx, "table", rows[[0, 3]], full, np.full(2, n), n * (n - 1))
