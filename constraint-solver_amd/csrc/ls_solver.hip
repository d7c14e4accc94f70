// ls_solver.hip — C entry points for the whole-solver view (include/ls_amd.h, "whole solver"):
// the host driver of csrc/host/ wired to the GPU path of this library.  Plain host C++; it is a
// .hip file only so that the one hipcc invocation style builds everything.
#include <sched.h>

#include <atomic>
#include <functional>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <thread>

#include "host/fibers.hpp"
#include "host/plugins.hpp"
#include "ls_common.hpp"

using namespace lshost;

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
asm(LS_FIBER_SWITCH_ASM); // host/fibers.hpp: the fibre switch, emitted once (this file has a device pass too: not there)
#endif

struct ls_solver {
    ls_ctx *ctx = nullptr;
    int kind = 0; // 0 N-queens, 1 scheduling
    std::unique_ptr<NQueensProblem> nq;
    std::unique_ptr<IteratedLocalSearch<NQueensProblem>> nq_ils;
    std::unique_ptr<SchedProblem> sc;
    std::unique_ptr<IteratedLocalSearch<SchedProblem>> sc_ils;
};

struct ls_multi_solver {
    ls_ctx *ctx = nullptr;
    int kind = 0; // 0 N-queens, 1 scheduling
    ls_solver_params params{};
    size_t board_size = 0;
    // scheduling: the instance every restart works on (examples/employee-scheduling/src/lib.rs:57-117)
    int32_t sc_days_scored = 0, sc_start_weekday = 0, sc_employees = 0;
    std::vector<uint64_t> sc_holidays;
    std::vector<ScoredSolution<SchedProblem>> sc_best;
    // every restart's proposer stream, drawn at creation like ls_solver_create_sched's (set-up, not part of a run: the stream is a
    // clone of the restart's LocalSearch generator in its seed state, lib.rs:488 -- the same list whenever it is drawn), and the
    // generator behind it, should a run ask for a longer prefix.  Empty: the restart draws its own when it starts.
    std::vector<std::vector<ls_sched_move>> sc_streams;
    std::vector<ChaCha20Rng> sc_stream_rng;
    // the resident launch the slots share and the slots' contexts: made by the first run, kept for later runs (pinned memory, a
    // stream on its own hardware queue), released by ls_multi_solver_free; the launch itself leaves at the end of every run
    lsamd::SchedStepServer *sc_step = nullptr;
    lsamd::SchedGrid *sc_grid = nullptr;
    std::vector<ls_ctx *> sc_slot_ctx;
    int sc_slots_made = 0;
    void sc_release()
    {
        if (sc_step) lsamd::sched_step_destroy(sc_step);
        if (sc_grid) lsamd::sched_grid_destroy(sc_grid);
        sc_step = nullptr; sc_grid = nullptr;
        for (ls_ctx *c : sc_slot_ctx) if (c) (void)ls_ctx_destroy(c); // after the launches have left (hipHostFree waits for resident kernels)
        sc_slot_ctx.clear();
        sc_slots_made = 0;
    }
    ~ls_multi_solver() { sc_release(); }
    std::vector<std::string> seeds;
    std::vector<ScoredSolution<NQueensProblem>> best;
    std::vector<int> have;
    int n_slots = 16;                 // restarts in flight at once (create: 4 per usable core); further seeds queue behind them
    int mode = LS_MULTI_FREE_RUNNING; // LS_MULTI_FREE_RUNNING: a context per slot; LS_MULTI_LOCKSTEP: one batched launch per step
    uint64_t ls_iterations = 0, neighbours_scored = 0, gpu_launches = 0, rounds = 0;
    uint64_t counters[LS_N_COUNTERS] = {0}; // ls_multi_solver_get_counters: the last run's
    uint64_t fibre_steals = 0;
    std::vector<uint64_t> restart_ls_iterations, restart_scored, restart_rounds; // per restart, of the last run
};

namespace {

int check_params(ls_ctx *ctx, const ls_solver_params *p)
{
    if (!p) return lsamd::set_err(ctx, LS_ERR_INVALID, "params is NULL");
    if (p->best_solutions_capacity == 0)
        return lsamd::set_err(ctx, LS_ERR_INVALID, "best_solutions_capacity must be >= 1 (the reference would panic)");
    return LS_OK;
}

// wait hook of a slot that is a fibre: say what the restart waits for and how far it has come (restart lengths are heavy-
// tailed: the one that has run longest goes first), then give the thread up
static void fibre_wait_hook(void *arg)
{
    const ls_ctx *c = static_cast<const ls_ctx *>(arg);
    FiberPool::set_priority(c->steps_done);
    if (!c->wait_words) { FiberPool::yield(); return; }
    FiberWait w;
    w.words = c->wait_words; w.n_words = c->n_wait_words; w.tag = c->wait_tag;
    FiberPool::yield(&w);
}

static void solver_set_idle_work(ls_ctx *c, bool (*fn)(void *), void *arg)
{
    c->idle_hook = fn;
    c->idle_arg = arg;
}

// Cores this process may really use: the affinity mask, further limited by the cgroup CPU quota when there is one
// (hardware_concurrency() reports the whole machine inside a container).
unsigned usable_cores()
{
    unsigned cores = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int c = CPU_COUNT(&set);
        if (c > 0) cores = (unsigned)c;
    }
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long long period = 0;
        if (fscanf(f, "%31s %lld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const long long q = atoll(quota) / period;
            if (q >= 1 && (unsigned)q < cores) cores = (unsigned)q;
        }
        fclose(f);
    }
    return cores < 1 ? 1 : cores;
}

template <class F> int guarded(ls_ctx *ctx, F &&f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return lsamd::set_err(ctx, LS_ERR_NOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return lsamd::set_err(ctx, LS_ERR_STATE, "%s", e.what());
    } catch (...) { // nothing crosses the C boundary
        return lsamd::set_err(ctx, LS_ERR_STATE, "unknown C++ exception");
    }
}

template <class V> int copy_solution(const V &v, int64_t *solution, int32_t cap, int32_t *len)
{
    if (len) *len = (int32_t)v.size();
    if (solution) {
        if ((size_t)cap < v.size()) return LS_ERR_INVALID;
        for (size_t i = 0; i < v.size(); i++) solution[i] = (int64_t)v[i];
    }
    return LS_OK;
}

} // namespace

extern "C" {

int ls_solver_create_nqueens(ls_ctx *ctx, uint64_t board_size, const char *seed, const ls_solver_params *params,
                             ls_solver **out)
{
    if (!ctx || !out || !seed) return LS_ERR_INVALID;
    *out = nullptr;
    int rc = check_params(ctx, params);
    if (rc != LS_OK) return rc;
    if (board_size < 1 || board_size > LS_NQ_MAX_N)
        return lsamd::set_err(ctx, LS_ERR_UNSUPPORTED, "board_size %llu outside [1, %d]", (unsigned long long)board_size,
                              LS_NQ_MAX_N);
    return guarded(ctx, [&]() {
        std::unique_ptr<ls_solver> s(new ls_solver());
        s->ctx = ctx;
        s->kind = 0;
        check_rc(ctx, lsamd::nq_ctx_prepare_server(ctx, (int)board_size), "LS-step server");
        s->nq.reset(new NQueensProblem(ctx, (size_t)board_size));
        s->nq->set_idle_work = &solver_set_idle_work;
        // examples/nqueens/src/main.rs:36-87: one seed, two identically seeded generators
        const ChaCha20Rng solver_rng = ChaCha20Rng::from_seed_string(seed);
        const ChaCha20Rng ils_rng = ChaCha20Rng::from_seed_string(seed);
        LocalSearch<NQueensProblem> ls(s->nq.get(), params->local_search_max_iterations, (size_t)params->window_size,
                                       (size_t)params->best_solutions_capacity, (size_t)params->all_solutions_capacity,
                                       params->all_solution_iteration_expiry, solver_rng);
        History<NQueensProblem> history((size_t)params->best_solutions_capacity, (size_t)params->all_solutions_capacity,
                                        params->all_solution_iteration_expiry);
        s->nq_ils.reset(new IteratedLocalSearch<NQueensProblem>(s->nq.get(), std::move(ls), std::move(history),
                                                                params->iterated_local_search_max_iterations,
                                                                params->max_allow_no_improvement_for, ils_rng));
        *out = s.release();
        return LS_OK;
    });
}

int ls_solver_create_sched(ls_ctx *ctx, int32_t n_days_scored, int32_t start_weekday, int32_t n_employees,
                           const uint64_t *holiday_bits, const char *seed, const ls_solver_params *params,
                           ls_solver **out)
{
    if (!ctx || !out || !seed) return LS_ERR_INVALID;
    *out = nullptr;
    int rc = check_params(ctx, params);
    if (rc != LS_OK) return rc;
    rc = ls_sched_instance_upload(ctx, n_days_scored, n_days_scored + 1, start_weekday, n_employees, holiday_bits);
    if (rc != LS_OK) return rc;
    return guarded(ctx, [&]() {
        std::unique_ptr<ls_solver> s(new ls_solver());
        s->ctx = ctx;
        s->kind = 1;
        s->sc.reset(new SchedProblem(ctx, n_days_scored, n_days_scored + 1, n_employees));
        const ChaCha20Rng solver_rng = ChaCha20Rng::from_seed_string(seed); // lib.rs:58-62
        const ChaCha20Rng ils_rng = ChaCha20Rng::from_seed_string(seed);    // lib.rs:94
        {   // set-up, not part of a round: the proposer's stream (a clone of the LocalSearch generator that nothing ever advances,
            // lib.rs:488 -- the same list whenever it is drawn) and the resident launch the LS iterations will ask
            const long long want = SchedProblem::list_prefix(params->window_size, n_employees);
            if (params->all_solution_iteration_expiry >= 1 && want > 0 && want < (1ll << 24)) {
                s->sc->extend_stream(solver_rng, (size_t)(want + want / 16));
                (void)lsamd::sched_ctx_prepare(ctx, want);
            }
        }
        LocalSearch<SchedProblem> ls(s->sc.get(), params->local_search_max_iterations, (size_t)params->window_size,
                                     (size_t)params->best_solutions_capacity, (size_t)params->all_solutions_capacity,
                                     params->all_solution_iteration_expiry, solver_rng);
        History<SchedProblem> history((size_t)params->best_solutions_capacity, (size_t)params->all_solutions_capacity,
                                      params->all_solution_iteration_expiry);
        s->sc_ils.reset(new IteratedLocalSearch<SchedProblem>(s->sc.get(), std::move(ls), std::move(history),
                                                              params->iterated_local_search_max_iterations,
                                                              params->max_allow_no_improvement_for, ils_rng));
        *out = s.release();
        return LS_OK;
    });
}

int ls_solver_execute_round(ls_solver *s)
{
    if (!s) return LS_ERR_INVALID;
    return guarded(s->ctx, [&]() {
        if (s->kind == 0) s->nq_ils->execute_round();
        else s->sc_ils->execute_round();
        return LS_OK;
    });
}

int ls_solver_is_finished(const ls_solver *s, int *finished)
{
    if (!s || !finished) return LS_ERR_INVALID;
    *finished = (s->kind == 0 ? s->nq_ils->is_finished() : s->sc_ils->is_finished()) ? 1 : 0;
    return LS_OK;
}

int ls_solver_get_iteration_info(const ls_solver *s, uint64_t *current, uint64_t *total)
{
    if (!s) return LS_ERR_INVALID;
    const IterationInfo i = s->kind == 0 ? s->nq_ils->get_iteration_info() : s->sc_ils->get_iteration_info();
    if (current) *current = i.current;
    if (total) *total = i.total;
    return LS_OK;
}

int ls_solver_get_best_solution(const ls_solver *s, int64_t score[2], int64_t *solution, int32_t cap, int32_t *len,
                                int *have)
{
    if (!s || !score) return LS_ERR_INVALID;
    if (s->kind == 0) {
        ScoredSolution<NQueensProblem> b;
        const bool ok = s->nq_ils->get_best_solution(b);
        if (have) *have = ok ? 1 : 0;
        if (!ok) { if (len) *len = 0; return LS_OK; }
        score[0] = b.score; score[1] = 0;
        return copy_solution(b.solution, solution, cap, len);
    }
    ScoredSolution<SchedProblem> b;
    const bool ok = s->sc_ils->get_best_solution(b);
    if (have) *have = ok ? 1 : 0;
    if (!ok) { if (len) *len = 0; return LS_OK; }
    score[0] = b.score.first; score[1] = b.score.second;
    return copy_solution(b.solution, solution, cap, len);
}

int ls_solver_get_current_solution(const ls_solver *s, int64_t score[2], int64_t *solution, int32_t cap, int32_t *len)
{
    if (!s || !score) return LS_ERR_INVALID;
    if (s->kind == 0) {
        const auto &c = s->nq_ils->current();
        score[0] = c.score; score[1] = 0;
        return copy_solution(c.solution, solution, cap, len);
    }
    const auto &c = s->sc_ils->current();
    score[0] = c.score.first; score[1] = c.score.second;
    return copy_solution(c.solution, solution, cap, len);
}

int ls_solver_get_stats(const ls_solver *s, uint64_t *ls_iterations, uint64_t *neighbours_scored, uint64_t *gpu_evals)
{
    if (!s) return LS_ERR_INVALID;
    if (s->kind == 0) {
        if (ls_iterations) *ls_iterations = s->nq_ils->ls_iterations();
        if (neighbours_scored) *neighbours_scored = s->nq->neighbours_scored;
        if (gpu_evals) *gpu_evals = s->nq->gpu_evals;
    } else {
        if (ls_iterations) *ls_iterations = s->sc_ils->ls_iterations();
        if (neighbours_scored) *neighbours_scored = s->sc->neighbours_scored;
        if (gpu_evals) *gpu_evals = s->sc->gpu_evals;
    }
    return LS_OK;
}

int ls_solver_free(ls_solver *s)
{
    delete s;
    return LS_OK;
}

int ls_multi_solver_create_nqueens(ls_ctx *ctx, uint64_t board_size, int32_t n_restarts, const char *const *seeds,
                                   const ls_solver_params *params, ls_multi_solver **out)
{
    if (!ctx || !out || !seeds || n_restarts < 1) return LS_ERR_INVALID;
    *out = nullptr;
    int rc = check_params(ctx, params);
    if (rc != LS_OK) return rc;
    if (board_size < 1 || board_size > LS_NQ_MAX_N)
        return lsamd::set_err(ctx, LS_ERR_UNSUPPORTED, "board_size %llu outside [1, %d]", (unsigned long long)board_size,
                              LS_NQ_MAX_N);
    return guarded(ctx, [&]() {
        std::unique_ptr<ls_multi_solver> m(new ls_multi_solver());
        m->ctx = ctx;
        m->params = *params;
        m->board_size = (size_t)board_size;
        for (int32_t r = 0; r < n_restarts; r++) {
            if (!seeds[r]) throw std::runtime_error("seed is NULL");
            m->seeds.emplace_back(seeds[r]);
        }
        m->best.resize((size_t)n_restarts);
        m->have.assign((size_t)n_restarts, 0);
        // default: four restarts in flight per usable core -- free-running slots beyond the cores share threads as
        // cooperative fibres, so a slot's host work runs while another slot's launch is in flight
        const unsigned hw = usable_cores();
        m->n_slots = (int)(4 * hw > 256 ? 256 : (4 * hw < 1 ? 1 : 4 * hw));
        *out = m.release();
        return LS_OK;
    });
}

int ls_multi_solver_create_sched(ls_ctx *ctx, int32_t n_days_scored, int32_t start_weekday, int32_t n_employees,
                                 const uint64_t *holiday_bits, int32_t n_restarts, const char *const *seeds,
                                 const ls_solver_params *params, ls_multi_solver **out)
{
    if (!ctx || !out || !seeds || n_restarts < 1) return LS_ERR_INVALID;
    *out = nullptr;
    int rc = check_params(ctx, params);
    if (rc != LS_OK) return rc;
    rc = ls_sched_instance_upload(ctx, n_days_scored, n_days_scored + 1, start_weekday, n_employees, holiday_bits); // validates; the run's shared launches carry it
    if (rc != LS_OK) return rc;
    return guarded(ctx, [&]() {
        std::unique_ptr<ls_multi_solver> m(new ls_multi_solver());
        m->ctx = ctx;
        m->kind = 1;
        m->params = *params;
        m->sc_days_scored = n_days_scored; m->sc_start_weekday = start_weekday; m->sc_employees = n_employees;
        m->sc_holidays.assign(holiday_bits, holiday_bits + n_employees);
        for (int32_t r = 0; r < n_restarts; r++) {
            if (!seeds[r]) throw std::runtime_error("seed is NULL");
            m->seeds.emplace_back(seeds[r]);
        }
        m->sc_best.resize((size_t)n_restarts);
        m->have.assign((size_t)n_restarts, 0);
        // eight restarts in flight per usable core (at most what one resident launch holds, multi_sched_setup): a scheduling LS iteration
        // is ~1 us of host work around a ~4.5 us round trip, and the fibres of a core hide the one behind the other.  Measured on the
        // example instance, 16 cores, 256 seeds: 64 / 96 / 128 / 256 slots -> 0.93 / 1.19 / 1.45 / 1.45 x 10^9 neighbours/s (64 seeds
        // in 64 slots: 9.2 x 10^8).  (Before every slot's tag and request count had a cache line of their own -- srv_common.hpp --
        // the same sweep peaked at 32 slots and 3.9 x 10^8.)
        const unsigned hw = usable_cores();
        m->n_slots = (int)(8 * hw > 256 ? 256 : (8 * hw < 1 ? 1 : 8 * hw));
        {   // the restarts' streams (at most 256 MiB of them), by as many threads as there are cores
            const long long want = SchedProblem::list_prefix(params->window_size, n_employees);
            const long long each = want + want / 16;
            if (params->all_solution_iteration_expiry >= 1 && want > 0 && want < (1ll << 24) &&
                each * (long long)sizeof(ls_sched_move) * n_restarts <= (256ll << 20)) {
                m->sc_streams.resize((size_t)n_restarts);
                m->sc_stream_rng.resize((size_t)n_restarts);
                std::atomic<int> next{0};
                auto draw = [&]() {
                    for (int r = next.fetch_add(1); r < n_restarts; r = next.fetch_add(1)) {
                        SchedProblem pr(ctx, n_days_scored, n_days_scored + 1, n_employees);
                        pr.extend_stream(ChaCha20Rng::from_seed_string(m->seeds[(size_t)r]), (size_t)each);
                        m->sc_streams[(size_t)r] = std::move(pr.stream);
                        m->sc_stream_rng[(size_t)r] = pr.stream_rng;
                    }
                };
                const int n_thr = (int)(hw < (unsigned)n_restarts ? hw : (unsigned)n_restarts);
                if (n_thr <= 1 || each < 4096) draw();
                else {
                    std::vector<std::thread> th;
                    for (int i = 0; i < n_thr; i++) th.emplace_back(draw);
                    for (auto &t : th) t.join();
                }
            }
        }
        *out = m.release();
        return LS_OK;
    });
}

static int multi_sched_setup(ls_multi_solver *m);

int ls_multi_solver_configure(ls_multi_solver *m, int32_t n_slots, int32_t mode)
{
    if (!m) return LS_ERR_INVALID;
    if (n_slots < 0 || n_slots > 256 || (mode != LS_MULTI_LOCKSTEP && mode != LS_MULTI_FREE_RUNNING))
        return lsamd::set_err(m->ctx, LS_ERR_INVALID, "ls_multi_solver_configure: n_slots %d (0..256), mode %d", n_slots, mode);
    if (n_slots == 0) { // the mode's default: lockstep slots are host threads, free-running slots may share them
        const unsigned hw = usable_cores();
        const unsigned want = mode == LS_MULTI_LOCKSTEP ? hw : (m->kind == 1 ? 8 * hw : 4 * hw);
        n_slots = (int)(want > 256 ? 256 : (want < 1 ? 1 : want));
    }
    m->n_slots = n_slots;
    m->mode = mode;
    if (m->kind == 1) return guarded(m->ctx, [&]() { (void)multi_sched_setup(m); return LS_OK; }); // set-up, not part of a run
    return LS_OK;
}

static int multi_run_sched(ls_multi_solver *m, uint64_t max_rounds_per_restart, double max_seconds);

int ls_multi_solver_run(ls_multi_solver *m, uint64_t max_rounds_per_restart, double max_seconds)
{
    if (!m) return LS_ERR_INVALID;
    if (m->kind == 1) return guarded(m->ctx, [&]() { return multi_run_sched(m, max_rounds_per_restart, max_seconds); });
    return guarded(m->ctx, [&]() {
        const int R = (int)m->seeds.size();
        const int T = m->n_slots < R ? m->n_slots : R; // slots = host threads; a finished restart's slot takes the next seed
        const bool lockstep = m->mode == LS_MULTI_LOCKSTEP;
        if (lockstep && m->params.all_solution_iteration_expiry == 0)
            throw std::runtime_error("a History that keeps more than the current solution tabu (all_solution_iteration_expiry == 0) "
                                     "issues one-restart launches: use LS_MULTI_FREE_RUNNING");
        // Free-running slots are the workgroups of ONE resident LS-step server launch (nq_server.hip): slot t's context sends its
        // one-restart steps to workgroup t, no launch per LS iteration.  (LS_NQ_STEP <= 3, LS_NQ_ALGO_SCAN, n > 8192: a launch per call.)
        struct SharedServer {
            lsamd::NqServer *s = nullptr;
            std::vector<ls_ctx *> owned; // per-slot contexts: destroyed after the server has left (hipHostFree waits for resident kernels)
            ~SharedServer()
            {
                if (s) lsamd::nq_server_destroy(s);
                for (ls_ctx *c : owned) if (c) (void)ls_ctx_destroy(c);
            }
        } shared;
        shared.owned.assign((size_t)T, nullptr);
        if (!lockstep && m->ctx->nq_step_mode >= 4 && lsamd::nq_server_supports(m->ctx, (int)m->board_size) &&
            T <= lsamd::nq_server_max_slots(m->ctx, (int)m->board_size)) // (more slots than the device holds resident: a launch per call)
            check_rc(m->ctx, lsamd::nq_server_create(m->ctx, (int)m->board_size, T, &shared.s), "LS-step server");
        std::vector<uint64_t> fallbacks((size_t)T, 0), reuploads((size_t)T, 0);
        NqBatcher batcher(m->ctx, (int32_t)m->board_size, T);
        std::vector<std::string> errors((size_t)T);
        std::vector<uint64_t> ls_it((size_t)R, 0), scored((size_t)R, 0), rounds((size_t)R, 0), launches((size_t)T, 0);
        std::atomic<int> next_restart{0};
        std::atomic<bool> stop{false};
        const auto t0 = std::chrono::steady_clock::now();
        auto worker = [&](int t, bool as_fibre) {
            ls_ctx *own = nullptr;
            try {
                ls_ctx *ctx = m->ctx;
                if (!lockstep && t > 0) { // free-running: one context (stream, staging block) per slot; slot 0 uses the caller's
                    check_rc(m->ctx, lsamd::ctx_clone_light(m->ctx, &own), "context for a slot");
                    shared.owned[(size_t)t] = own;
                    ctx = own;
                }
                struct Attach { // this slot's workgroup of the shared server (slot 0: the caller's context lends itself for the run)
                    ls_ctx *c; lsamd::NqServer *s0; int slot0; bool owned0; bool on;
                    ~Attach() { if (on) { c->server = s0; c->server_slot = slot0; c->server_owned = owned0; } }
                } attach{ctx, ctx->server, ctx->server_slot, ctx->server_owned, shared.s != nullptr};
                if (shared.s) { ctx->server = shared.s; ctx->server_slot = t; ctx->server_owned = false; }
                // no shared launch (more slots than the device holds resident, or a board the server does not take): the slots launch
                // per call -- they must not start a resident server EACH (that many launches could not all be resident either)
                struct StepMode { ls_ctx *c; int m0; ~StepMode() { c->nq_step_mode = m0; } } step_mode{ctx, ctx->nq_step_mode};
                if (!shared.s && !lockstep && ctx->nq_step_mode >= 4) ctx->nq_step_mode = 3;
                const uint64_t fb0 = ctx->n_done_fallbacks, ru0 = ctx->n_resident_uploads;
                struct Count { ls_ctx *c; uint64_t *fb, *ru; uint64_t fb0, ru0; ~Count() { *fb = c->n_done_fallbacks - fb0; *ru = c->n_resident_uploads - ru0; } }
                    count{ctx, &fallbacks[(size_t)t], &reuploads[(size_t)t], fb0, ru0};
                // more slots than host threads: a waiting slot hands its thread to a slot that has host work to do
                struct Restore {
                    ls_ctx *c; void (*h)(void *); void *a;
                    ~Restore() { c->wait_hook = h; c->wait_hook_arg = a; }
                } restore{ctx, ctx->wait_hook, ctx->wait_hook_arg};
                if (as_fibre) { ctx->wait_hook = &fibre_wait_hook; ctx->wait_hook_arg = ctx; }
                const ls_solver_params &p = m->params;
                for (;;) {
                    const int r = next_restart.fetch_add(1);
                    if (r >= R || stop.load()) break;
                    ctx->steps_done = 0; // the new restart's scheduling priority starts over
                    NQueensProblem problem(ctx, m->board_size, lockstep ? &batcher : nullptr);
                    problem.set_idle_work = &solver_set_idle_work;
                    const ChaCha20Rng solver_rng = ChaCha20Rng::from_seed_string(m->seeds[(size_t)r]);
                    const ChaCha20Rng ils_rng = ChaCha20Rng::from_seed_string(m->seeds[(size_t)r]);
                    LocalSearch<NQueensProblem> ls(&problem, p.local_search_max_iterations, (size_t)p.window_size,
                                                   (size_t)p.best_solutions_capacity, (size_t)p.all_solutions_capacity,
                                                   p.all_solution_iteration_expiry, solver_rng);
                    History<NQueensProblem> history((size_t)p.best_solutions_capacity, (size_t)p.all_solutions_capacity,
                                                    p.all_solution_iteration_expiry);
                    IteratedLocalSearch<NQueensProblem> ils(&problem, std::move(ls), std::move(history),
                                                            p.iterated_local_search_max_iterations,
                                                            p.max_allow_no_improvement_for, ils_rng);
                    while (!ils.is_finished() && rounds[(size_t)r] < max_rounds_per_restart) {
                        ils.execute_round();
                        rounds[(size_t)r]++;
                        ScoredSolution<NQueensProblem> b;
                        if (ils.get_best_solution(b) && NQueensProblem::is_best(b.score)) break; // later rounds are no-ops
                        if (max_seconds > 0 &&
                            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > max_seconds) {
                            stop.store(true); // no further seeds are started
                            break;
                        }
                    }
                    ScoredSolution<NQueensProblem> b;
                    if (ils.get_best_solution(b)) { m->best[(size_t)r] = b; m->have[(size_t)r] = 1; }
                    ls_it[(size_t)r] = ils.ls_iterations();
                    scored[(size_t)r] = problem.neighbours_scored;
                    launches[(size_t)t] += problem.gpu_evals + problem.gpu_col_score_calls;
                }
            } catch (const std::exception &e) {
                errors[(size_t)t] = e.what();
                batcher.fail(e.what());
            } catch (...) { // nothing may escape a std::thread (std::terminate)
                errors[(size_t)t] = "unknown exception in a restart";
                batcher.fail(errors[(size_t)t]);
            }
            batcher.thread_done();
        };
        // Free-running slots beyond the host's cores are cooperative fibres of ONE pool over the cores (slot t starts on worker
        // t mod n_threads; a worker that runs dry takes suspended fibres from a busy one, host/fibers.hpp).
        const int hwc = (int)usable_cores();
        const int n_threads = (!lockstep && hwc >= 1 && T > hwc) ? hwc : T;
        m->fibre_steals = 0;
        if (n_threads == T) {
            std::vector<std::thread> threads;
            for (int th = 0; th < n_threads; th++) threads.emplace_back([&, th]() { worker(th, false); });
            for (auto &t : threads) t.join();
        } else {
            FiberPool pool(n_threads);
            for (int t = 0; t < T; t++) pool.add([&, t]() { worker(t, true); }, t);
            pool.run();
            m->fibre_steals = pool.steals();
        }
        for (const std::string &e : errors)
            if (!e.empty()) throw std::runtime_error(e);
        m->restart_ls_iterations = ls_it; m->restart_scored = scored; m->restart_rounds = rounds;
        m->ls_iterations = m->neighbours_scored = m->rounds = m->gpu_launches = 0;
        for (int r = 0; r < R; r++) {
            m->ls_iterations += ls_it[(size_t)r];
            m->neighbours_scored += scored[(size_t)r];
            m->rounds += rounds[(size_t)r];
        }
        if (lockstep) m->gpu_launches = batcher.launches;
        else for (int t = 0; t < T; t++) m->gpu_launches += launches[(size_t)t];
        for (uint64_t &c : m->counters) c = 0;
        if (shared.s) {
            unsigned long long c4[4];
            lsamd::nq_server_counters(shared.s, c4);
            for (int i = 0; i < 4; i++) m->counters[i] = c4[i];
        }
        for (int t = 0; t < T; t++) { m->counters[4] += fallbacks[(size_t)t]; m->counters[5] += reuploads[(size_t)t]; }
        m->counters[6] = m->fibre_steals;
        return LS_OK;
    });
}

int ls_multi_solver_get_best(const ls_multi_solver *m, int32_t restart, int64_t score[2], int64_t *solution, int32_t cap,
                             int32_t *len, int *have)
{
    if (!m || !score || restart < 0 || (size_t)restart >= m->seeds.size()) return LS_ERR_INVALID;
    if (have) *have = m->have[(size_t)restart];
    if (!m->have[(size_t)restart]) { if (len) *len = 0; return LS_OK; }
    if (m->kind == 1) {
        score[0] = m->sc_best[(size_t)restart].score.first; score[1] = m->sc_best[(size_t)restart].score.second;
        return copy_solution(m->sc_best[(size_t)restart].solution, solution, cap, len);
    }
    score[0] = m->best[(size_t)restart].score; score[1] = 0;
    return copy_solution(m->best[(size_t)restart].solution, solution, cap, len);
}

int ls_multi_solver_get_stats(const ls_multi_solver *m, uint64_t *ls_iterations, uint64_t *neighbours_scored,
                              uint64_t *gpu_launches, uint64_t *rounds)
{
    if (!m) return LS_ERR_INVALID;
    if (ls_iterations) *ls_iterations = m->ls_iterations;
    if (neighbours_scored) *neighbours_scored = m->neighbours_scored;
    if (gpu_launches) *gpu_launches = m->gpu_launches;
    if (rounds) *rounds = m->rounds;
    return LS_OK;
}

int ls_multi_solver_get_restart_stats(const ls_multi_solver *m, int32_t restart, uint64_t *ls_iterations, uint64_t *neighbours_scored,
                                      uint64_t *rounds)
{
    if (!m || restart < 0 || (size_t)restart >= m->seeds.size()) return LS_ERR_INVALID;
    const size_t r = (size_t)restart;
    if (ls_iterations) *ls_iterations = r < m->restart_ls_iterations.size() ? m->restart_ls_iterations[r] : 0;
    if (neighbours_scored) *neighbours_scored = r < m->restart_scored.size() ? m->restart_scored[r] : 0;
    if (rounds) *rounds = r < m->restart_rounds.size() ? m->restart_rounds[r] : 0;
    return LS_OK;
}

int ls_multi_solver_get_counters(const ls_multi_solver *m, uint64_t out[LS_N_COUNTERS])
{
    if (!m || !out) return LS_ERR_INVALID;
    for (int i = 0; i < LS_N_COUNTERS; i++) out[i] = m->counters[i];
    return LS_OK;
}

int ls_multi_solver_free(ls_multi_solver *m)
{
    delete m;
    return LS_OK;
}

int ls_host_hash_seed(const char *seed, uint8_t out32[32])
{
    if (!seed || !out32) return LS_ERR_INVALID;
    blake2b_256(seed, strlen(seed), out32);
    return LS_OK;
}

int ls_host_chacha20_block(const uint32_t key[8], uint32_t w12, uint32_t w13, uint32_t w14, uint32_t w15,
                           uint32_t out16[16])
{
    if (!key || !out16) return LS_ERR_INVALID;
    chacha20_block(key, w12, w13, w14, w15, out16);
    return LS_OK;
}

int ls_host_rng_script(const uint8_t seed32[32], const int64_t *ops, int64_t n_ops, const double *weights,
                       uint64_t *out, int64_t cap, int64_t *n_out)
{
    if (!seed32 || !ops || !out || !n_out) return LS_ERR_INVALID;
    try {
        ChaCha20Rng rng(seed32);
        int64_t k = 0;
        auto push = [&](uint64_t v) {
            if (k >= cap) throw std::runtime_error("output buffer too small");
            out[k++] = v;
        };
        for (int64_t i = 0; i < n_ops; i++) {
            const int64_t op = ops[3 * i], a = ops[3 * i + 1], b = ops[3 * i + 2];
            switch (op) {
            case 0: push(rng.next_u32()); break;
            case 1: push(rng.next_u64()); break;
            case 2: push(gen_range_u64_incl(rng, (uint64_t)a, (uint64_t)b)); break;
            case 3: push(gen_range_u32_incl(rng, (uint32_t)a, (uint32_t)b)); break;
            case 4: push(uniform_u64_below(rng, (uint64_t)a)); break;
            case 5: push(uniform_u32_below(rng, (uint32_t)a)); break;
            case 6: {
                std::vector<uint64_t> v((size_t)a);
                for (size_t j = 0; j < v.size(); j++) v[j] = j;
                shuffle(rng, v);
                for (uint64_t x : v) push(x);
                break;
            }
            case 7:
                for (uint32_t x : index_sample(rng, (uint32_t)a, (uint32_t)b)) push(x);
                break;
            case 8: {
                const double d = rng.gen_f64();
                uint64_t bits;
                memcpy(&bits, &d, 8);
                push(bits);
                break;
            }
            case 9: {
                if (!weights) return LS_ERR_INVALID;
                std::vector<double> w(weights, weights + a);
                for (uint32_t x : index_sample_weighted(rng, w, (uint32_t)b)) push(x);
                break;
            }
            default: return LS_ERR_INVALID;
            }
        }
        *n_out = k;
        return LS_OK;
    } catch (const std::exception &) {
        return LS_ERR_INVALID;
    }
}

} // extern "C"

// Many scheduling restarts on one GPU: restart = the unchanged driver of examples/employee-scheduling/src/lib.rs:57-117
// on a host thread or fibre, seeds by the convention of examples/nqueens/src/main.rs:167; the slots' GPU requests go to ONE resident
// launch shared by all of them -- slot t is workgroup t of the LS-step server (sched_server.hip: lists of at most 512 moves, the
// reference example's window 100), or sub-grid t of the grid (sched_grid.hip: longer lists, config 5's window 65 536) -- so no
// slot ever waits for another and nothing is launched per LS iteration.  When neither fits (too many slots for the device, a
// window between the two) every slot's context takes its own routes.
// The slots of a scheduling run and what they share: how many restarts are in flight (the device may hold fewer workgroups / sub-grids
// side by side than n_slots asks for), the resident launch's host object, one light context per slot.  Made once per slot count --
// by ls_multi_solver_configure, or by the first run -- and kept; the launch itself starts with a run's first request.
static int multi_sched_setup(ls_multi_solver *m)
{
    const int R = (int)m->seeds.size();
    int T = m->n_slots < R ? m->n_slots : R;
    const ls_solver_params &p = m->params;
    ls_ctx *base = m->ctx;
    const long long want = SchedProblem::list_prefix(p.window_size, m->sc_employees); // the list prefix an LS iteration asks for (SchedProblem::best_of_neighbourhood)
    const bool resident = base->sched_server_on && base->sched_delta && m->sc_employees <= 32768 && p.all_solution_iteration_expiry >= 1;
    if (resident && want <= 512) {
        const int cap = lsamd::sched_step_max_slots(base);
        if (T > cap) T = cap;
    } else if (resident) {
        while (T > 1 && !lsamd::sched_grid_fits(base, T, false, want)) T--; // as many sub-grids as the device holds side by side
    }
    if (m->sc_slots_made != T) { // (first time, or another slot count: the shared launch is sized by it)
        m->sc_release();
        if (resident && want <= 512) check_rc(base, lsamd::sched_step_create(base, T, &m->sc_step), "scheduling LS-step server");
        else if (resident && lsamd::sched_grid_fits(base, T, false, want)) check_rc(base, lsamd::sched_grid_create(base, T, false, want, &m->sc_grid), "scheduling grid");
        m->sc_slot_ctx.assign((size_t)T, nullptr);
        for (int t = 1; t < T; t++) check_rc(base, lsamd::ctx_clone_light(base, &m->sc_slot_ctx[(size_t)t]), "context for a slot"); // (the instance is base's: create_sched uploaded it)
        m->sc_slots_made = T;
    }
    return T;
}

static int multi_run_sched(ls_multi_solver *m, uint64_t max_rounds_per_restart, double max_seconds)
{
    const int R = (int)m->seeds.size();
    int T = 0;
    const ls_solver_params &p = m->params;
    ls_ctx *base = m->ctx;
    static const bool trace = getenv("LS_AMD_TRACE") != nullptr;
    const auto T_begin = std::chrono::steady_clock::now();
    struct Phase { // LS_AMD_TRACE: where a run's wall time goes besides the restarts themselves
        std::chrono::steady_clock::time_point t0; bool on; double setup = 0, run = 0;
        ~Phase() { if (on) fprintf(stderr, "multi-restart scheduling run: set-up %.2f ms, restarts %.2f ms, tear-down %.2f ms\n", setup * 1e3, run * 1e3,
                                   (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() - setup - run) * 1e3); }
    } phase{T_begin, trace};
    // the caller's own resident launches (earlier single-solver calls on this context) leave now instead of 2 ms from now: the
    // shared launch may need every CU
    if (base->sched_step && base->sched_step_owned) lsamd::sched_step_pause(base->sched_step);
    if (base->sched_grid && base->sched_grid_owned) lsamd::sched_grid_pause(base->sched_grid);
    T = multi_sched_setup(m); // (made by ls_multi_solver_configure already, or by an earlier run: then this is a comparison)
    struct Shared { lsamd::SchedStepServer *step; lsamd::SchedGrid *grid; } shared{m->sc_step, m->sc_grid};
    struct LeaveAtEnd { // the resident launch leaves with the run (it would idle out by itself 2 ms later)
        Shared &s;
        ~LeaveAtEnd() { if (s.step) lsamd::sched_step_pause(s.step); if (s.grid) lsamd::sched_grid_pause(s.grid); }
    } leave_at_end{shared};
    std::vector<std::string> errors((size_t)T);
    std::vector<uint64_t> ls_it((size_t)R, 0), scored((size_t)R, 0), rounds((size_t)R, 0), launches((size_t)T, 0);
    std::atomic<int> next_restart{0};
    std::atomic<bool> stop{false};
    const auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int t, bool as_fibre) {
        try {
            ls_ctx *ctx = base;
            if (t > 0) ctx = m->sc_slot_ctx[(size_t)t]; // one context per slot (the drivers are independent); slot 0 uses the caller's
            struct Attach { // this slot's workgroup / sub-grid of the shared launch (slot 0: the caller's context lends itself for the run)
                ls_ctx *c; lsamd::SchedStepServer *s0; int slot0; bool owned0; lsamd::SchedGrid *g0; int sub0; bool gowned0; bool un0;
                ~Attach() { c->sched_step = s0; c->sched_step_slot = slot0; c->sched_step_owned = owned0; c->sched_grid = g0; c->sched_grid_sub = sub0; c->sched_grid_owned = gowned0; c->sched_grid_unavailable = un0; }
            } attach{ctx, ctx->sched_step, ctx->sched_step_slot, ctx->sched_step_owned, ctx->sched_grid, ctx->sched_grid_sub, ctx->sched_grid_owned, ctx->sched_grid_unavailable};
            if (shared.step) { ctx->sched_step = shared.step; ctx->sched_step_slot = t; ctx->sched_step_owned = false; }
            if (shared.grid) { ctx->sched_step = nullptr; ctx->sched_step_owned = false; ctx->sched_grid = shared.grid; ctx->sched_grid_sub = t; ctx->sched_grid_owned = false; ctx->sched_grid_unavailable = false; }
            struct Restore {
                ls_ctx *c; void (*h)(void *); void *a;
                ~Restore() { c->wait_hook = h; c->wait_hook_arg = a; }
            } restore{ctx, ctx->wait_hook, ctx->wait_hook_arg};
            if (as_fibre) { ctx->wait_hook = &fibre_wait_hook; ctx->wait_hook_arg = ctx; }
            for (;;) {
                const int r = next_restart.fetch_add(1);
                if (r >= R || stop.load()) break;
                ctx->steps_done = 0; // the new restart's scheduling priority starts over
                SchedProblem problem(ctx, m->sc_days_scored, m->sc_days_scored + 1, m->sc_employees);
                if ((size_t)r < m->sc_streams.size() && !m->sc_streams[(size_t)r].empty()) { // drawn at creation (a copy: the solver may run again)
                    problem.stream = m->sc_streams[(size_t)r];
                    problem.stream_rng = m->sc_stream_rng[(size_t)r];
                    problem.stream_started = true;
                }
                const ChaCha20Rng solver_rng = ChaCha20Rng::from_seed_string(m->seeds[(size_t)r]); // lib.rs:58-62
                const ChaCha20Rng ils_rng = ChaCha20Rng::from_seed_string(m->seeds[(size_t)r]);    // lib.rs:94
                LocalSearch<SchedProblem> ls(&problem, p.local_search_max_iterations, (size_t)p.window_size, (size_t)p.best_solutions_capacity,
                                             (size_t)p.all_solutions_capacity, p.all_solution_iteration_expiry, solver_rng);
                History<SchedProblem> history((size_t)p.best_solutions_capacity, (size_t)p.all_solutions_capacity, p.all_solution_iteration_expiry);
                IteratedLocalSearch<SchedProblem> ils(&problem, std::move(ls), std::move(history), p.iterated_local_search_max_iterations,
                                                      p.max_allow_no_improvement_for, ils_rng);
                while (!ils.is_finished() && rounds[(size_t)r] < max_rounds_per_restart) {
                    ils.execute_round();
                    rounds[(size_t)r]++;
                    ScoredSolution<SchedProblem> b;
                    if (ils.get_best_solution(b) && SchedProblem::is_best(b.score)) break; // later rounds are no-ops
                    if (max_seconds > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > max_seconds) {
                        stop.store(true); // no further seeds are started
                        break;
                    }
                }
                ScoredSolution<SchedProblem> b;
                if (ils.get_best_solution(b)) { m->sc_best[(size_t)r] = b; m->have[(size_t)r] = 1; }
                ls_it[(size_t)r] = ils.ls_iterations();
                scored[(size_t)r] = problem.neighbours_scored;
                launches[(size_t)t] += problem.gpu_evals;
            }
        } catch (const std::exception &e) {
            errors[(size_t)t] = e.what();
        } catch (...) { // nothing may escape a std::thread (std::terminate)
            errors[(size_t)t] = "unknown exception in a restart";
        }
    };
    const int hwc = (int)usable_cores();
    const int n_threads = (hwc >= 1 && T > hwc) ? hwc : T;
    m->fibre_steals = 0;
    phase.setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - T_begin).count();
    const auto T_run = std::chrono::steady_clock::now();
    if (n_threads == T) {
        std::vector<std::thread> threads;
        for (int th = 0; th < n_threads; th++) threads.emplace_back([&, th]() { worker(th, false); });
        for (auto &t : threads) t.join();
    } else {
        FiberPool pool(n_threads);
        for (int t = 0; t < T; t++) pool.add([&, t]() { worker(t, true); }, t);
        pool.run();
        m->fibre_steals = pool.steals();
    }
    phase.run = std::chrono::duration<double>(std::chrono::steady_clock::now() - T_run).count();
    for (const std::string &e : errors)
        if (!e.empty()) throw std::runtime_error(e);
    m->restart_ls_iterations = ls_it; m->restart_scored = scored; m->restart_rounds = rounds;
    m->ls_iterations = m->neighbours_scored = m->rounds = m->gpu_launches = 0;
    for (int r = 0; r < R; r++) {
        m->ls_iterations += ls_it[(size_t)r];
        m->neighbours_scored += scored[(size_t)r];
        m->rounds += rounds[(size_t)r];
    }
    for (int t = 0; t < T; t++) m->gpu_launches += launches[(size_t)t];
    for (uint64_t &c : m->counters) c = 0;
    unsigned long long c3[3] = {0, 0, 0};
    if (shared.step) lsamd::sched_step_counters(shared.step, c3);
    if (shared.grid) lsamd::sched_grid_counters(shared.grid, c3);
    for (int i = 0; i < 3; i++) m->counters[i] = c3[i];
    m->counters[6] = m->fibre_steals;
    return LS_OK;
}
