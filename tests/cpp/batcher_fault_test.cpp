// CPU-side unit test of the lockstep batcher's failure path (host logic only; no GPU, no HIP).
// The product header csrc/host/plugins.hpp is compiled against a FAKE C ABI defined here: ls_nqueens_col_scores and
// ls_nqueens_eval_batch_next are brute-force CPU stand-ins (literal get_col_scores, examples/nqueens/src/lib.rs:74-87)
// that can be told to fail on the k-th call.  The test drives several restarts on threads exactly like
// ls_multi_solver_run does and checks, under AddressSanitizer / ThreadSanitizer, that
//   * without a fault every restart finishes and equals its single (unbatched) run,
//   * with a fault injected into the ABI, or thrown by one restart's own thread, every thread unwinds, the error is
//     reported, and nobody writes through a parked request of an unwound stack (plugins.hpp:69).
// Build + run: tests/test_host_batcher.py.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../constraint-solver_amd/csrc/host/plugins.hpp"

struct ls_ctx { std::string err; };
static std::atomic<long> g_calls{0};
static std::atomic<long> g_fail_at{-1};

static void col_scores(const int32_t *rows, int n, int64_t *out)
{
    for (int i = 0; i < n; i++) out[i] = 0;
    for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++) {
            const long d = (long)rows[b] - rows[a];
            if (d == 0 || (d < 0 ? -d : d) == b - a) { out[a]++; out[b]++; }
        }
}

extern "C" {
const char *ls_last_error(const ls_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

int ls_nqueens_col_scores(ls_ctx *ctx, int32_t R, int32_t n, const int32_t *rows, int64_t *cs, int64_t *tot)
{
    if (g_calls.fetch_add(1) == g_fail_at.load()) { ctx->err = "injected fault (col_scores)"; return LS_ERR_HIP; }
    std::vector<int64_t> tmp((size_t)n);
    for (int r = 0; r < R; r++) {
        col_scores(rows + (size_t)r * n, n, tmp.data());
        int64_t t = 0;
        for (int i = 0; i < n; i++) { if (cs) cs[(size_t)r * n + i] = tmp[i]; t += tmp[i]; }
        if (tot) tot[r] = t;
    }
    return LS_OK;
}

int ls_nqueens_eval_batch_next(ls_ctx *ctx, int32_t R, int32_t n, const int32_t *rows, const int64_t *cur, const int32_t *cols,
                               const int32_t *n_cols, int32_t max_cols, int64_t window, ls_nq_result *res, int64_t *next_cs)
{
    (void)cur;
    if (g_calls.fetch_add(1) == g_fail_at.load()) { ctx->err = "injected fault (eval)"; return LS_ERR_HIP; }
    std::vector<int32_t> b((size_t)n), best_b;
    std::vector<int64_t> cs((size_t)n);
    for (int r = 0; r < R; r++) {
        const int32_t *row = rows + (size_t)r * n;
        ls_nq_result out{0, -1, 0, -1, -1};
        for (int s = 0; s < n_cols[r] && out.n_scored < window; s++) {
            const int c = cols[(size_t)r * max_cols + s];
            for (int v = 0; v < n && out.n_scored < window; v++) {
                if (v == row[c]) continue; // tabu
                b.assign(row, row + n);
                b[(size_t)c] = v;
                col_scores(b.data(), n, cs.data());
                int64_t sc = 0;
                for (int i = 0; i < n; i++) sc += cs[(size_t)i];
                out.n_scored++;
                if (out.best_index < 0 || sc < out.best_score || (sc == out.best_score && b < best_b)) {
                    out.best_index = (int64_t)s * n + v; out.best_score = sc; out.best_col = c; out.best_value = v; best_b = b;
                }
            }
        }
        res[r] = out;
        if (out.best_index >= 0) col_scores(best_b.data(), n, next_cs + (size_t)r * n);
        else col_scores(row, n, next_cs + (size_t)r * n);
    }
    return LS_OK;
}
// general-History entries (plugins.hpp references them; this test never builds a tabu set beyond {current})
int ls_nqueens_eval_tabu(ls_ctx *ctx, int32_t, const int32_t *, int64_t, const int32_t *, int32_t, int64_t, const int64_t *, int32_t,
                         ls_nq_result *, int64_t *) { ctx->err = "not in this fake"; return LS_ERR_UNSUPPORTED; }
int ls_sched_eval(ls_ctx *ctx, const int32_t *, const ls_sched_move *, int64_t, int64_t, ls_sched_result *, int64_t *, int64_t *)
{ ctx->err = "not in this fake"; return LS_ERR_UNSUPPORTED; }
int ls_sched_score(ls_ctx *ctx, int32_t, const int32_t *, int64_t *, int64_t *, int64_t *) { ctx->err = "not in this fake"; return LS_ERR_UNSUPPORTED; }
int ls_sched_moves_upload(ls_ctx *ctx, const ls_sched_move *, int64_t) { ctx->err = "not in this fake"; return LS_ERR_UNSUPPORTED; }
int ls_sched_eval_resident(ls_ctx *ctx, const int32_t *, int64_t, int64_t, ls_sched_result *) { ctx->err = "not in this fake"; return LS_ERR_UNSUPPORTED; }
} // extern "C"

using namespace lshost;

struct Outcome { std::vector<ScoredSolution<NQueensProblem>> best; std::vector<std::string> errors; };

static Outcome run(int R, int n, int rounds, bool batched, int throwing_restart)
{
    ls_ctx ctx;
    Outcome o;
    o.best.resize((size_t)R);
    o.errors.resize((size_t)R);
    NqBatcher batcher(&ctx, n, R);
    auto worker = [&](int r) {
        try {
            NQueensProblem problem(&ctx, (size_t)n, batched ? &batcher : nullptr);
            const std::string seed = std::to_string(42 + r);
            LocalSearch<NQueensProblem> ls(&problem, 1000, (size_t)(5 * n), 8, 1000, 1000, ChaCha20Rng::from_seed_string(seed.c_str()));
            History<NQueensProblem> h(8, 1000, 1000);
            IteratedLocalSearch<NQueensProblem> ils(&problem, std::move(ls), std::move(h), 1000, 3, ChaCha20Rng::from_seed_string(seed.c_str()));
            for (int i = 0; i < rounds; i++) {
                if (r == throwing_restart && i == 1) throw std::runtime_error("restart-local failure");
                ils.execute_round();
            }
            ils.get_best_solution(o.best[(size_t)r]);
        } catch (const std::exception &e) {
            o.errors[(size_t)r] = e.what();
            batcher.fail(e.what());
        }
        batcher.thread_done();
    };
    std::vector<std::thread> th;
    for (int r = 0; r < R; r++) th.emplace_back(worker, r);
    for (auto &t : th) t.join();
    return o;
}

int main()
{
    const int R = 6, n = 12, rounds = 4;
    // 1. no fault: batched == unbatched
    g_fail_at = -1;
    const Outcome solo = run(R, n, rounds, false, -1);
    const Outcome lock = run(R, n, rounds, true, -1);
    for (int r = 0; r < R; r++) {
        if (!solo.errors[(size_t)r].empty() || !lock.errors[(size_t)r].empty()) { printf("FAIL unexpected error\n"); return 1; }
        if (!(solo.best[(size_t)r] == lock.best[(size_t)r])) { printf("FAIL restart %d differs\n", r); return 1; }
    }
    // 2. the ABI fails on its k-th call, for many k (first step, mid-run, while restarts are finishing)
    for (long k : {0L, 1L, 2L, 5L, 17L, 40L, 90L}) {
        g_calls = 0;
        g_fail_at = k;
        const Outcome f = run(R, n, rounds, true, -1);
        int failed = 0;
        for (const std::string &e : f.errors) failed += !e.empty();
        if (g_calls.load() > k && failed == 0) { printf("FAIL fault at call %ld was swallowed\n", k); return 1; }
        for (const std::string &e : f.errors)
            if (!e.empty() && e.find("injected fault") == std::string::npos) { printf("FAIL wrong message: %s\n", e.c_str()); return 1; }
    }
    // 3. one restart throws outside the batcher while the others are parked in submit()
    g_fail_at = -1;
    for (int victim = 0; victim < R; victim += 2) {
        const Outcome f = run(R, n, rounds, true, victim);
        if (f.errors[(size_t)victim].find("restart-local failure") == std::string::npos) { printf("FAIL victim not reported\n"); return 1; }
    }
    printf("OK\n");
    return 0;
}
