// ls_api.hip — the extern "C" boundary declared in include/ls_amd.h: context
// life-cycle, argument validation, host<->device staging.  Kernels live in
// nq_kernels.hip / nq_server.hip / sched_*.hip.  No CPU fallback exists anywhere in this
// library: without a HIP device ls_ctx_create fails.
#include <immintrin.h>
#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <new>

#include "ls_common.hpp"
#include "sched_common.hpp"

namespace lsamd {

static thread_local std::string g_create_err;

int set_err(const ls_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_err = buf;
    return code;
}

// nq_kernels.hip
int nq_eval_batch_dev(ls_ctx *ctx, int n_restarts, int n, const int *rows_d, const long long *cur_d,
                      const int *cols_d, const int *n_cols_d, int max_cols, long long window,
                      ls_nq_result *results_d, long long *scores_out_d, int *next_rows_d);
int nq_col_scores_dev(ls_ctx *ctx, int n_restarts, int n, const int *rows_d, long long *col_scores_d,
                      long long *scores_d);
int nq_eval_excl_dev(ls_ctx *ctx, int n, const int *rows_d, const long long *cur_d, const int *cols_d, const int *n_cols_d,
                     int max_cols, long long window, const long long *excl_d, int n_excl, ls_nq_result *results_d,
                     long long *scores_out_d);
bool nq_step_eligible(const ls_ctx *ctx, int n, int max_cols, long long window);
int nq_step_dev(ls_ctx *ctx, int n_restarts, int n, const int *rows, const long long *cur, const int *cols,
                const int *n_cols, int max_cols, long long window, ls_nq_result *results, int *next_rows,
                long long *next_col_scores, long long *next_totals, const NqStepExtra *extra, const NqStepInline *inl = nullptr);
// sched_wave.hip / sched_delta.hip
int sched_score_dev(ls_ctx *ctx, int n_states, const int *states_d, long long *hard_d, long long *soft_d,
                    long long *terms_d);
int sched_eval_batch_dev(ls_ctx *ctx, int n_restarts, const int *states_d, const ls_sched_move *moves_d,
                         const long long *n_moves_d, long long max_moves, long long window,
                         ls_sched_result *results_d, long long *hard_out_d, long long *soft_out_d,
                         unsigned long long *done_flag = nullptr, unsigned long long done_seq = 0,
                         const int *inline_state_h = nullptr, long long inline_n_moves = 0);
bool sched_is_fused_launch(const ls_ctx *ctx, long long max_moves);
int sched_exhaustive_moves_dev(ls_ctx *ctx, int n_restarts, const int *states_d, ls_sched_move *moves_d, long long *n_moves_d);
int sched_eval_exhaustive_dev(ls_ctx *ctx, int n_restarts, const int *states_d, long long window, ls_sched_result *results_d,
                              bool *handled, unsigned long long *done_flag = nullptr, unsigned long long done_seq = 0,
                              const int *inline_state_h = nullptr);

static void for_each_buf(ls_ctx *ctx, void (*f)(DevBuf &))
{
    DevBuf *bufs[] = {&ctx->nq_rows, &ctx->nq_cur, &ctx->nq_cols, &ctx->nq_ncols, &ctx->nq_results,
                      &ctx->nq_scores, &ctx->nq_keys, &ctx->nq_tabs, &ctx->nq_colscores, &ctx->nq_totals, &ctx->nq_next_rows,
                      &ctx->sc_states, &ctx->sc_moves, &ctx->sc_nmoves, &ctx->sc_results, &ctx->sc_hard,
                      &ctx->sc_soft, &ctx->sc_keys, &ctx->sc_terms, &ctx->sc_cand_keys, &ctx->sched.holiday_bits, &ctx->sched.lane_tab,
                      &ctx->sc_resident_moves};
    for (DevBuf *b : bufs) f(*b);
}

// Block until the context's stream has drained, then hand back the workspaces that grew out of use meanwhile
// (DevBuf::retired): nothing queued can still be reading them.
int sync_and_collect(ls_ctx *ctx)
{
    LS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for_each_buf(ctx, [](DevBuf &b) { if (!b.retired.empty()) b.collect(); });
    return LS_OK;
}

static int bind(ls_ctx *ctx)
{
    if (!ctx) return LS_ERR_INVALID;
    LS_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->stream) { // a light clone (ctx_clone_light): its stream is made when a launched route first needs one
        LS_HIP(ctx, hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
        ctx->stream = ctx->own_stream;
    }
    return LS_OK;
}

// A context for one slot of a launch shared with `base` (ls_multi_solver): same device, same kernel-family choices, the
// scheduling instance BORROWED from base (which outlives it), no stream, no allocation -- the steady-state path of such a
// slot makes no runtime call, so creating and destroying dozens of full contexts per run was most of a short run's wall time.
int ctx_clone_light(const ls_ctx *base, ls_ctx **out)
{
    *out = nullptr;
    ls_ctx *c = new (std::nothrow) ls_ctx();
    if (!c) return set_err(base, LS_ERR_NOMEM, "out of host memory");
    c->device = base->device;
    c->nq_algo = base->nq_algo; c->nq_step_mode = base->nq_step_mode; c->nq_scan_variant = base->nq_scan_variant;
    c->sched_delta = base->sched_delta; c->sched_server_on = base->sched_server_on;
    c->n_cus = base->n_cus; c->wave_size = base->wave_size; c->clock_khz = base->clock_khz;
    memcpy(c->dev_name, base->dev_name, sizeof c->dev_name);
    if (base->sched.valid) {
        c->sched.valid = true;
        c->sched.n_days_scored = base->sched.n_days_scored; c->sched.n_days_stored = base->sched.n_days_stored;
        c->sched.start_weekday = base->sched.start_weekday; c->sched.n_employees = base->sched.n_employees;
        c->sched.holiday_bits.borrow(base->sched.holiday_bits);
        c->sched.lane_tab.borrow(base->sched.lane_tab);
    }
    *out = c;
    return LS_OK;
}

// The server a one-restart LS step of this context goes to: the attached slot of a shared server (ls_multi_solver), or the
// context's own one-slot server, started on demand (and restarted when the board size changes).  *out stays null when the
// step has to take the launch-per-call route (a shared server built for another board size).
int nq_ctx_server(ls_ctx *ctx, int n, NqServer **out, int slots)
{
    *out = nullptr;
    if (ctx->server && nq_server_n(ctx->server) == n && nq_server_slots(ctx->server) >= slots) { *out = ctx->server; return LS_OK; }
    if (ctx->server && !ctx->server_owned) return LS_OK;
    if (ctx->server && nq_server_n(ctx->server) == n && nq_server_slots(ctx->server) > slots) slots = nq_server_slots(ctx->server); // never shrinks
    if (ctx->server) { // another n: the context's own server is replaced; its counters stay in the context's totals
        unsigned long long c4[4];
        nq_server_counters(ctx->server, c4);
        for (int i = 0; i < 4; i++) ctx->server_retired[i] += c4[i];
        nq_server_destroy(ctx->server);
        ctx->server = nullptr;
    }
    NqServer *s = nullptr;
    const int rc = nq_server_create(ctx, n, slots, &s);
    if (rc == LS_ERR_UNSUPPORTED) return LS_OK; // more slots than the device holds resident: the caller launches per call
    if (rc != LS_OK) return rc;
    ctx->server = s; ctx->server_slot = 0; ctx->server_owned = true;
    *out = s;
    return LS_OK;
}

// ls_solver_create_nqueens: start the context's own server before the driver's first get_col_scores call, so that the
// initial board is uploaded once and stays with the slot.
int nq_ctx_prepare_server(ls_ctx *ctx, int n)
{
    if (ctx->nq_step_mode < 4 || !nq_server_supports(ctx, n)) return LS_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return LS_OK;
    NqServer *s = nullptr;
    return nq_ctx_server(ctx, n, &s);
}

// ------------------------------------------------------------------------------------------- scheduling: resident routes
static bool sched_resident_ok(const ls_ctx *ctx)
{
    return ctx->sched_server_on && ctx->sched_delta && ctx->sched.valid && ctx->sched.n_employees <= 32768;
}

static void sched_retire_counters(ls_ctx *ctx, const unsigned long long c3[3])
{
    for (int i = 0; i < 3; i++) ctx->sched_retired[i] += c3[i];
}

// this context's own resident launches leave and are released (their arguments -- the instance -- are about to change, or the
// context goes away); an attached slot of a shared launch is only detached
void sched_ctx_drop_servers(ls_ctx *ctx, bool detach_shared)
{
    (void)hipSetDevice(ctx->device);
    if (ctx->sched_step && ctx->sched_step_owned) {
        unsigned long long c3[3];
        sched_step_counters(ctx->sched_step, c3);
        sched_retire_counters(ctx, c3);
        sched_step_destroy(ctx->sched_step);
        ctx->sched_step = nullptr;
    } else if (detach_shared) ctx->sched_step = nullptr;
    if (ctx->sched_grid && ctx->sched_grid_owned) {
        unsigned long long c3[3];
        sched_grid_counters(ctx->sched_grid, c3);
        sched_retire_counters(ctx, c3);
        sched_grid_destroy(ctx->sched_grid);
        ctx->sched_grid = nullptr;
    } else if (detach_shared) ctx->sched_grid = nullptr;
}

// the step server a call of this context goes to: the attached slot of a shared launch, or its own one-slot server (on demand)
static int sched_ctx_step(ls_ctx *ctx, SchedStepServer **out)
{
    if (!ctx->sched_step) {
        SchedStepServer *s = nullptr;
        const int rc = sched_step_create(ctx, 1, &s);
        if (rc != LS_OK) return rc;
        ctx->sched_step = s; ctx->sched_step_slot = 0; ctx->sched_step_owned = true;
    }
    *out = ctx->sched_step;
    return LS_OK;
}

// the grid a call of this context goes to, or null (the launched pipeline serves it): the attached sub-grid of a shared launch
// when it is of the right kind and size, or the context's own one-sub-grid launch, replaced when the kind or the list length changes
static int sched_ctx_grid(ls_ctx *ctx, bool gen, long long n_list, SchedGrid **out)
{
    *out = nullptr;
    if (ctx->sched_grid_unavailable || !sched_resident_ok(ctx)) return LS_OK;
    if (ctx->sched_grid) {
        SchedGrid *g = ctx->sched_grid;
        const bool fits = sched_grid_is_gen(g) == gen && (gen || n_list <= sched_grid_capacity(g));
        if (fits) { *out = g; return LS_OK; }
        if (!ctx->sched_grid_owned) return LS_OK;
        unsigned long long c3[3];
        sched_grid_counters(g, c3);
        sched_retire_counters(ctx, c3);
        sched_grid_destroy(g);
        ctx->sched_grid = nullptr;
    }
    if (!sched_grid_fits(ctx, 1, gen, n_list)) return LS_OK;
    SchedGrid *g = nullptr;
    const int rc = sched_grid_create(ctx, 1, gen, n_list, &g);
    if (rc != LS_OK) return rc;
    ctx->sched_grid = g; ctx->sched_grid_sub = 0; ctx->sched_grid_owned = true;
    *out = g;
    return LS_OK;
}

// ls_solver_create_sched: start the resident launch this solver's LS iterations will use (lists of `want` moves) before the first
// round, like nq_ctx_prepare_server -- pinned blocks, a stream on its own hardware queue and the first launch are set-up, not
// part of a round.  Failure to do so is not an error: the calls create what they need on demand.
int sched_ctx_prepare(ls_ctx *ctx, long long want)
{
    if (!sched_resident_ok(ctx) || want <= 0) return LS_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return LS_OK;
    if (want <= 512) {
        SchedStepServer *sv = nullptr;
        (void)sched_ctx_step(ctx, &sv);
    } else {
        SchedGrid *g = nullptr;
        (void)sched_ctx_grid(ctx, false, want, &g);
    }
    return LS_OK;
}

// the device copy of the resident list, for the launched pipelines (made when one first needs the current list)
static int sched_resident_dev(ls_ctx *ctx)
{
    if (ctx->sc_resident_dev_gen == ctx->sc_resident_gen) return LS_OK;
    {
        const int rc0 = bind(ctx);
        if (rc0 != LS_OK) return rc0;
    }
    const size_t bytes = ctx->sc_resident_host.size() * sizeof(ls_sched_move);
    int rc = sync_and_collect(ctx); // the stream may still read the previous list
    if (rc != LS_OK) return rc;
    LS_RESERVE(ctx, ctx->sc_resident_moves, bytes ? bytes : 1);
    if (bytes) LS_HIP(ctx, hipMemcpyAsync(ctx->sc_resident_moves.p, ctx->sc_resident_host.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = sync_and_collect(ctx);
    if (rc != LS_OK) return rc;
    ctx->sc_resident_dev_gen = ctx->sc_resident_gen;
    return LS_OK;
}

} // namespace lsamd

using namespace lsamd;

#define LS_BIND(ctx)                         \
    do {                                     \
        int rc__ = lsamd::bind(ctx);         \
        if (rc__ != LS_OK) return rc__;      \
    } while (0)

#define LS_SYNC(ctx)                                 \
    do {                                             \
        int rc__ = lsamd::sync_and_collect(ctx);     \
        if (rc__ != LS_OK) return rc__;              \
    } while (0)

#define LS_REQUIRE(ctx, cond, ...)                                           \
    do {                                                                     \
        if (!(cond)) return lsamd::set_err((ctx), LS_ERR_INVALID, __VA_ARGS__); \
    } while (0)

extern "C" {

int ls_abi_version(void) { return LS_ABI_VERSION; }

int ls_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ls_ctx_create(int device_id, ls_ctx **out)
{
    if (!out) return set_err(nullptr, LS_ERR_INVALID, "ls_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_err(nullptr, LS_ERR_NO_DEVICE,
                       "no HIP device available (%s); this library has no CPU fallback",
                       e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        return set_err(nullptr, LS_ERR_NO_DEVICE, "device id %d out of range [0, %d)", device_id, n);
    ls_ctx *ctx = new (std::nothrow) ls_ctx();
    if (!ctx) return set_err(nullptr, LS_ERR_NOMEM, "out of host memory");
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        set_err(nullptr, LS_ERR_HIP, "device %d initialisation failed: %s", device_id, hipGetErrorString(e));
        delete ctx;
        return LS_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    if (const char *v = getenv("LS_NQ_SCAN_VARIANT")) ctx->nq_scan_variant = atoi(v);
    if (const char *v = getenv("LS_NQ_STEP")) ctx->nq_step_mode = atoi(v) >= 4 ? 4 : 3; // 4 (default): resident server; 3: one fused launch per call
    if (const char *v = getenv("LS_SCHED_DELTA")) ctx->sched_delta = atoi(v) != 0;
    if (const char *v = getenv("LS_SCHED_SERVER")) ctx->sched_server_on = atoi(v) != 0;
    ctx->n_cus = prop.multiProcessorCount;
    ctx->wave_size = prop.warpSize;
    ctx->clock_khz = prop.clockRate;
    snprintf(ctx->dev_name, sizeof ctx->dev_name, "%s (%s)", prop.name, prop.gcnArchName);
    if (ctx->wave_size != 64) {
        set_err(nullptr, LS_ERR_UNSUPPORTED, "device %d has wavefront size %d; the kernels are written for wave64 (gfx950)",
                device_id, ctx->wave_size);
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return LS_ERR_UNSUPPORTED;
    }
    *out = ctx;
    return LS_OK;
}

int ls_ctx_destroy(ls_ctx *ctx)
{
    if (!ctx) return LS_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->server && ctx->server_owned) nq_server_destroy(ctx->server); // the resident launches leave before anything is freed
    ctx->server = nullptr;
    sched_ctx_drop_servers(ctx, true);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for_each_buf(ctx, [](DevBuf &b) { b.release(); });
    ctx->stage_in.release();
    ctx->stage_out.release();
    if (ctx->nq_res.calls && getenv("LS_AMD_TRACE"))
        fprintf(stderr, "ls step (resident): %llu calls, per call: prepare %.2f us, launch %.2f us, wait %.2f us, unpack %.2f us\n", ctx->nq_res.calls,
                ctx->nq_res.t_pre / ctx->nq_res.calls * 1e6, ctx->nq_res.t_launch / ctx->nq_res.calls * 1e6, ctx->nq_res.t_wait / ctx->nq_res.calls * 1e6,
                ctx->nq_res.t_post / ctx->nq_res.calls * 1e6);
    ctx->nq_res.dev_rows.release();
    ctx->nq_res.dev_tabs.release();
    ctx->nq_res.counter.release();
    if (ctx->nq_res.flag) (void)hipHostFree(ctx->nq_res.flag);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return LS_OK;
}

const char *ls_last_error(const ls_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int ls_ctx_set_stream(ls_ctx *ctx, void *hip_stream)
{
    if (!ctx) return LS_ERR_INVALID;
    hipStream_t next = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    if (next != ctx->stream) {
        // Work queued on the old stream (e.g. a finalize kernel restoring the argmin keys, or kernels reading
        // the context's workspaces) is not ordered before launches on the new one: drain it first -- unless the
        // old stream is being captured, where nothing runs and a synchronisation would break the capture.
        LS_BIND(ctx);
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(ctx->stream, &st) != hipSuccess) { (void)hipGetLastError(); st = hipStreamCaptureStatusNone; }
        if (st == hipStreamCaptureStatusNone) LS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->stream = next;
    }
    return LS_OK;
}

int ls_ctx_synchronize(ls_ctx *ctx)
{
    LS_BIND(ctx);
    return sync_and_collect(ctx);
}

int ls_ctx_set_nq_algo(ls_ctx *ctx, int algo)
{
    if (!ctx) return LS_ERR_INVALID;
    LS_REQUIRE(ctx, algo == LS_NQ_ALGO_SCAN || algo == LS_NQ_ALGO_TABLE || algo == LS_NQ_ALGO_AUTO,
               "unknown N-queens algo %d", algo);
    ctx->nq_algo = algo;
    return LS_OK;
}

int ls_ctx_get_nq_algo(const ls_ctx *ctx, int *algo)
{
    if (!ctx || !algo) return LS_ERR_INVALID;
    *algo = ctx->nq_algo;
    return LS_OK;
}

int ls_ctx_get_counters(const ls_ctx *ctx, uint64_t out[LS_N_COUNTERS])
{
    if (!ctx || !out) return LS_ERR_INVALID;
    for (int i = 0; i < LS_N_COUNTERS; i++) out[i] = 0;
    for (int i = 0; i < 4; i++) out[i] = ctx->server_retired[i]; // servers of earlier board sizes
    if (ctx->server && ctx->server_owned) {
        unsigned long long c4[4];
        nq_server_counters(ctx->server, c4);
        for (int i = 0; i < 4; i++) out[i] += c4[i];
    } else if (ctx->server) {
        unsigned long long c4[4];
        nq_server_counters(ctx->server, c4);
        for (int i = 0; i < 4; i++) out[i] = c4[i];
    }
    {
        // the resident scheduling launches of this context count into the same three
        unsigned long long c3[3];
        for (int i = 0; i < 3; i++) out[i] += ctx->sched_retired[i];
        if (ctx->sched_step && ctx->sched_step_owned) { sched_step_counters(ctx->sched_step, c3); for (int i = 0; i < 3; i++) out[i] += c3[i]; }
        if (ctx->sched_grid && ctx->sched_grid_owned) { sched_grid_counters(ctx->sched_grid, c3); for (int i = 0; i < 3; i++) out[i] += c3[i]; }
    }
    out[4] = ctx->n_done_fallbacks;
    out[5] = ctx->n_resident_uploads;
    return LS_OK;
}

int ls_ctx_device_info(const ls_ctx *ctx, int *n_cus, int *wave_size, int *clock_khz, char *name, int name_cap)
{
    if (!ctx) return LS_ERR_INVALID;
    if (n_cus) *n_cus = ctx->n_cus;
    if (wave_size) *wave_size = ctx->wave_size;
    if (clock_khz) *clock_khz = ctx->clock_khz;
    if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s", ctx->dev_name);
    return LS_OK;
}

/* ------------------------------------------------------------- N-queens */

static int nq_check_dims(ls_ctx *ctx, int32_t n_restarts, int32_t n, int32_t max_cols, int64_t window)
{
    LS_REQUIRE(ctx, n_restarts >= 1, "n_restarts must be >= 1 (got %d)", n_restarts);
    LS_REQUIRE(ctx, n >= 1, "n must be >= 1 (got %d)", n);
    if (n > LS_NQ_MAX_N) return set_err(ctx, LS_ERR_UNSUPPORTED, "n = %d exceeds LS_NQ_MAX_N = %d", n, LS_NQ_MAX_N);
    LS_REQUIRE(ctx, max_cols >= 0, "max_cols must be >= 0 (got %d)", max_cols);
    if (max_cols > LS_NQ_MAX_N)
        return set_err(ctx, LS_ERR_UNSUPPORTED, "max_cols = %d exceeds %d (15-bit slot field of the argmin key)", max_cols,
                       LS_NQ_MAX_N);
    LS_REQUIRE(ctx, window >= 0, "window must be >= 0 (got %lld)", (long long)window);
    return LS_OK;
}

// Range check of a large int32 array on the per-iteration path: one branch-free (vectorisable) pass for the common
// all-valid case; the element-by-element walk that names the offender only runs when it found something.
static bool all_in_range(const int32_t *p, long long count, int32_t hi_excl)
{
    uint32_t worst = 0; // (uint32_t)x < hi_excl  <=>  0 <= x < hi_excl
    for (long long i = 0; i < count; i++) worst = (uint32_t)p[i] > worst ? (uint32_t)p[i] : worst;
    return count == 0 || worst < (uint32_t)hi_excl;
}

static long long nq_clamp_window(int64_t window, int32_t max_cols, int32_t n)
{
    const long long cap = (long long)max_cols * n + 1;
    return window > cap ? cap : window;
}

int ls_nqueens_eval_batch_dev(ls_ctx *ctx, int32_t n_restarts, int32_t n, const int32_t *rows_d,
                              const int64_t *cur_scores_d, const int32_t *cols_d, const int32_t *n_cols_d,
                              int32_t max_cols, int64_t window, ls_nq_result *results_d, int64_t *scores_out_d)
{
    LS_BIND(ctx);
    int rc = nq_check_dims(ctx, n_restarts, n, max_cols, window);
    if (rc != LS_OK) return rc;
    LS_REQUIRE(ctx, rows_d && cur_scores_d && n_cols_d && results_d && (cols_d || max_cols == 0),
               "null device pointer");
    return nq_eval_batch_dev(ctx, n_restarts, n, rows_d, reinterpret_cast<const long long *>(cur_scores_d), cols_d,
                             n_cols_d, max_cols, nq_clamp_window(window, max_cols, n), results_d,
                             reinterpret_cast<long long *>(scores_out_d), nullptr);
}

// 16-bit column scores (as the LS-step kernel writes them) -> the ABI's int64 array; the AVX2 form takes 0.3 us for n = 4096
// where the scalar loop takes 2 us of a 19 us call.
__attribute__((target("avx2"))) static void widen_u16_avx2(const unsigned short *src, int64_t *dst, int n)
{
    int j = 0;
    for (; j + 4 <= n; j += 4) {
        const __m128i v = _mm_loadl_epi64(reinterpret_cast<const __m128i *>(src + j));
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(dst + j), _mm256_cvtepu16_epi64(v));
    }
    for (; j < n; j++) dst[j] = src[j];
}

static void widen_u16(const unsigned short *src, int64_t *dst, int n)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { widen_u16_avx2(src, dst, n); return; }
    for (int j = 0; j < n; j++) dst[j] = src[j];
}

// Column scores of the board after ONE queen moved (column mc: row mq -> mv), from the scores before the move
// (get_col_scores, examples/nqueens/src/lib.rs:74-87): a column j != mc loses the old queen's attack and gains the new
// queen's; column mc gets the number of columns attacking its new square.  `rows` is the board BEFORE the move.
// O(n) on the host instead of 2n bytes over PCIe and a pass of the step kernel.
__attribute__((target("avx2"))) static void nq_cs_after_move_avx2(const int32_t *rows, int32_t *cs, int n, int mc, int mv, int mq)
{
    const __m256i vmc = _mm256_set1_epi32(mc), vmv = _mm256_set1_epi32(mv), vmq = _mm256_set1_epi32(mq);
    __m256i idx = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7), cnt = _mm256_setzero_si256();
    const __m256i eight = _mm256_set1_epi32(8);
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        const __m256i q = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(rows + j));
        const __m256i dc = _mm256_abs_epi32(_mm256_sub_epi32(idx, vmc));
        const __m256i dn = _mm256_abs_epi32(_mm256_sub_epi32(q, vmv)), dq = _mm256_abs_epi32(_mm256_sub_epi32(q, vmq));
        // all-ones (-1) where the square is attacked
        const __m256i hn = _mm256_or_si256(_mm256_cmpeq_epi32(dn, _mm256_setzero_si256()), _mm256_cmpeq_epi32(dn, dc));
        const __m256i hq = _mm256_or_si256(_mm256_cmpeq_epi32(dq, _mm256_setzero_si256()), _mm256_cmpeq_epi32(dq, dc));
        __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(cs + j));
        c = _mm256_add_epi32(_mm256_sub_epi32(c, hn), hq); // c + [hn] - [hq]
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(cs + j), c);
        cnt = _mm256_sub_epi32(cnt, hn);
        idx = _mm256_add_epi32(idx, eight);
    }
    int32_t lanes[8];
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(lanes), cnt);
    int total = 0;
    for (int i = 0; i < 8; i++) total += lanes[i];
    for (; j < n; j++) {
        const int q = rows[j], dc = j > mc ? j - mc : mc - j;
        const int dn = q > mv ? q - mv : mv - q, dq = q > mq ? q - mq : mq - q;
        const int hn = (dn == 0) | (dn == dc), hq = (dq == 0) | (dq == dc);
        cs[j] += hn - hq;
        total += hn;
    }
    // column mc itself: its old queen (q = mq, dc = 0) counted hn = [mq == mv] = 0 and hq = 1 above; replace the entry
    cs[mc] = total;
}

} // extern "C"
namespace lsamd {
void nq_cs_after_move(const int32_t *rows, int32_t *cs, int n, int mc, int mv, int mq)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { nq_cs_after_move_avx2(rows, cs, n, mc, mv, mq); return; }
    int total = 0;
    for (int j = 0; j < n; j++) {
        const int q = rows[j], dc = j > mc ? j - mc : mc - j;
        const int dn = q > mv ? q - mv : mv - q, dq = q > mq ? q - mq : mq - q;
        const int hn = (dn == 0) | (dn == dc), hq = (dq == 0) | (dq == dc);
        cs[j] += hn - hq;
        total += hn;
    }
    cs[mc] = total;
}

} // namespace lsamd
extern "C" {

__attribute__((target("avx2"))) static void widen_i32_avx2(const int32_t *src, int64_t *dst, int n)
{
    int j = 0;
    for (; j + 4 <= n; j += 4)
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(dst + j), _mm256_cvtepi32_epi64(_mm_loadu_si128(reinterpret_cast<const __m128i *>(src + j))));
    for (; j < n; j++) dst[j] = src[j];
}

// nq_cs_after_move and widen_i32 in one pass (sixteen columns per step): the scores are read, corrected, stored and widened once
__attribute__((target("avx512f"))) static void nq_cs_after_move_widen_avx512(const int32_t *rows, int32_t *cs, int64_t *out, int n, int mc, int mv, int mq)
{
    const __m512i vmc = _mm512_set1_epi32(mc), vmv = _mm512_set1_epi32(mv), vmq = _mm512_set1_epi32(mq), one = _mm512_set1_epi32(1);
    __m512i idx = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15), cnt = _mm512_setzero_si512();
    const __m512i sixteen = _mm512_set1_epi32(16), zero = _mm512_setzero_si512();
    int j = 0;
    for (; j + 16 <= n; j += 16) {
        const __m512i q = _mm512_loadu_si512(rows + j);
        const __m512i dc = _mm512_abs_epi32(_mm512_sub_epi32(idx, vmc));
        const __m512i dn = _mm512_abs_epi32(_mm512_sub_epi32(q, vmv)), dq = _mm512_abs_epi32(_mm512_sub_epi32(q, vmq));
        const __mmask16 hn = _mm512_cmpeq_epi32_mask(dn, zero) | _mm512_cmpeq_epi32_mask(dn, dc); // attacked from the new square
        const __mmask16 hq = _mm512_cmpeq_epi32_mask(dq, zero) | _mm512_cmpeq_epi32_mask(dq, dc); // ... from the old one
        __m512i c = _mm512_loadu_si512(cs + j);
        c = _mm512_mask_add_epi32(c, hn, c, one);
        c = _mm512_mask_sub_epi32(c, hq, c, one);
        _mm512_storeu_si512(cs + j, c);
        _mm512_storeu_si512(out + j, _mm512_cvtepi32_epi64(_mm512_castsi512_si256(c)));
        _mm512_storeu_si512(out + j + 8, _mm512_cvtepi32_epi64(_mm512_extracti64x4_epi64(c, 1)));
        cnt = _mm512_mask_add_epi32(cnt, hn, cnt, one);
        idx = _mm512_add_epi32(idx, sixteen);
    }
    int total = _mm512_reduce_add_epi32(cnt);
    for (; j < n; j++) {
        const int q = rows[j], dc = j > mc ? j - mc : mc - j;
        const int dn = q > mv ? q - mv : mv - q, dq = q > mq ? q - mq : mq - q;
        const int hn = (dn == 0) | (dn == dc), hq = (dq == 0) | (dq == dc);
        cs[j] += hn - hq;
        out[j] = cs[j];
        total += hn;
    }
    cs[mc] = total; // column mc itself, as in nq_cs_after_move
    out[mc] = total;
}

} // extern "C"
namespace lsamd {
void widen_i32(const int32_t *src, int64_t *dst, int n);
void nq_cs_after_move_widen(const int32_t *rows, int32_t *cs, int64_t *out, int n, int mc, int mv, int mq)
{
    static const bool wide = __builtin_cpu_supports("avx512f") && getenv("LS_HOST_NO_AVX512") == nullptr;
    if (wide) { nq_cs_after_move_widen_avx512(rows, cs, out, n, mc, mv, mq); return; }
    nq_cs_after_move(rows, cs, n, mc, mv, mq);
    widen_i32(cs, out, n);
}
void widen_i32(const int32_t *src, int64_t *dst, int n)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { widen_i32_avx2(src, dst, n); return; }
    for (int j = 0; j < n; j++) dst[j] = src[j];
}

} // namespace lsamd
extern "C" {

// Wait for a kernel's completion word (coherent host memory) instead of synchronising the stream: the poll sees the
// word a PCIe write after the kernel's last store, a stream synchronisation costs a runtime round trip on top.
// Falls back to the stream (and reports its error) if the word does not arrive.
static int wait_done_word(ls_ctx *ctx, const unsigned long long *flag, unsigned long long seq)
{
    const auto t0 = std::chrono::steady_clock::now();
    bool idle_work = ctx->idle_hook != nullptr;
    for (unsigned spins = 0;; spins++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return LS_OK;
        if (idle_work) { idle_work = ctx->idle_hook(ctx->idle_arg); continue; } // a bounded piece of the caller's own work, then poll again
        if (ctx->wait_hook) ctx->wait_hook(ctx->wait_hook_arg); // another restart of this thread gets the core meanwhile
#if defined(__x86_64__)
        else __builtin_ia32_pause();
#endif
        // A launch takes >= 10 us: after a short spin give the core to another runnable thread (ls_multi_solver runs
        // more slots than cores: one slot's host proposer then runs while the other slot's launch is in flight)
        if ((spins & 0xfff) == 0xfff) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt > 0.002) { // far beyond any LS step: let the runtime tell whether the launch failed
                ctx->n_done_fallbacks++;
                LS_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return LS_OK;
                return set_err(ctx, LS_ERR_HIP, "the LS-step kernel finished without publishing its completion word");
            }
        }
    }
}


// The context's completion word (coherent host memory) and the block counter of multi-block publishers.  The counter is
// allocated and zeroed (by a kernel, in stream order: DESIGN 3 -- no memset node in any launch path) BEFORE the word is
// published in rs.flag, so a failed allocation cannot leave a launch with a null counter.
static int ensure_done_word(ls_ctx *ctx)
{
    NqResident &rs = ctx->nq_res;
    if (rs.flag) return LS_OK;
    LS_RESERVE(ctx, rs.counter, 256);
    hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(64), 0, ctx->stream, rs.counter.as<int>(), 0, 64ll);
    LS_HIP(ctx, hipGetLastError());
    unsigned long long *flag = nullptr;
    LS_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&flag), 64, hipHostMallocCoherent));
    *flag = 0;
    rs.flag = flag;
    return LS_OK;
}

// LS_NQ_STEP=3: the fused LS step with boards resident on the device.  Inputs that cross PCIe per call: the current
// scores, the proposed columns and -- only for a restart whose board is not the one the device already holds -- the
// board; outputs: the result records and the winners' column scores as 16-bit values.  One launch, no stream sync.
static int nq_host_step_resident(ls_ctx *ctx, int32_t R, int32_t n, const int32_t *rows, const int64_t *cur_scores,
                                 const int32_t *cols, const int32_t *n_cols, int32_t max_cols, long long win,
                                 ls_nq_result *results, int64_t *next_col_scores)
{
    NqResident &rs = ctx->nq_res;
    ctx->steps_done++;
    static const bool trace = getenv("LS_AMD_TRACE") != nullptr;
    const auto T0 = trace ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    {
        const int rc0 = ensure_done_word(ctx);
        if (rc0 != LS_OK) return rc0;
    }
    if (rs.n != n || (int)rs.valid.size() < R) { // geometry changed: nothing on the device can be reused
        // the step kernel's table layout (nq_step_dev picks the paired layout for even n >= 128): 6n + 4 or 5n + 1 u16
        const bool pairs = (n & 1) == 0 && n >= 128;
        const int tab_words = pairs ? ((((6 * n + 4) >> 1) + 3) & ~3) : ((((5 * n + 1) >> 1) + 3) & ~3);
        rs.n = 0; // nothing is committed before both reservations have succeeded (a failure leaves "no resident geometry")
        rs.valid.clear();
        LS_RESERVE(ctx, rs.dev_rows, (size_t)R * n * 4);
        LS_RESERVE(ctx, rs.dev_tabs, (size_t)R * tab_words * 4);
        rs.tab_words = tab_words;
        rs.valid.assign((size_t)R, 0);
        rs.cs_valid.assign((size_t)R, 0);
        rs.shadow.resize((size_t)R * n);
        rs.cs.resize((size_t)R * n);
        rs.n = n;
    }
    const size_t rows_b = (size_t)R * n * 4, cur_b = (size_t)R * 8, cols_b = (size_t)R * max_cols * 4, nc_b = (size_t)R * 4,
                 res_b = (size_t)R * sizeof(ls_nq_result), cs_b = next_col_scores ? (size_t)R * n * 2 : 0;
    const size_t o_cur = 0, o_cols = align256(o_cur + cur_b), o_nc = align256(o_cols + cols_b), o_rows = align256(o_nc + nc_b),
                 in_b = align256(o_rows + rows_b);
    const size_t o_res = 0, o_cs = align256(o_res + res_b), out_b = align256(o_cs + cs_b);
    {
        hipError_t e = ctx->stage_in.reserve(in_b);
        if (e == hipSuccess) e = ctx->stage_out.reserve(out_b);
        if (e != hipSuccess) return set_err(ctx, LS_ERR_NOMEM, "staging allocation failed: %s", hipGetErrorString(e));
    }
    Staging &si = ctx->stage_in, &so = ctx->stage_out;
    // A board that is not the one left by the previous call is uploaded; its column scores then come from the kernel.
    // Otherwise the host keeps them itself: cs(winner board) follows from cs(board) and the move in O(n).
    unsigned long long fresh = 0, want_cs = 0;
    for (int r = 0; r < R; r++) {
        const int32_t *src = rows + (size_t)r * n;
        int32_t *sh = rs.shadow.data() + (size_t)r * n;
        if (r >= 64 || !rs.valid[(size_t)r] || memcmp(src, sh, (size_t)n * 4) != 0) {
            memcpy(si.h<char>(o_rows) + (size_t)r * n * 4, src, (size_t)n * 4);
            memcpy(sh, src, (size_t)n * 4);
            rs.valid[(size_t)r] = 1;
            rs.cs_valid[(size_t)r] = 0;
            ctx->n_resident_uploads++;
            if (r < 64) fresh |= 1ull << r;
        }
        if (!cs_b) rs.cs_valid[(size_t)r] = 0; // nobody keeps the scores across this call
        if (cs_b && r < 64 && !rs.cs_valid[(size_t)r]) want_cs |= 1ull << r;
    }
    memcpy(si.h<char>(o_cur), cur_scores, cur_b);
    if (cols_b) memcpy(si.h<char>(o_cols), cols, cols_b);
    memcpy(si.h<char>(o_nc), n_cols, nc_b);
    NqStepExtra x;
    x.fresh_mask = fresh;
    x.resident_rows = rs.dev_rows.as<int>();
    x.resident_tabs = rs.dev_tabs.as<unsigned>();
    x.tab_words = rs.tab_words;
    x.next_cs16 = cs_b ? so.h<unsigned short>(o_cs) : nullptr;
    x.want_cs_mask = want_cs;
    x.done_flag = rs.flag;
    x.done_seq = ++rs.seq;
    x.block_counter = rs.counter.as<unsigned>();
    NqStepInline inl;
    inl.valid = 0;
    if (R == 1 && max_cols <= NQ_STEP_INLINE_COLS) { // the small inputs ride in the kernel arguments: no PCIe read in the kernel
        inl.valid = 1;
        inl.cur = cur_scores[0];
        inl.n_cols = n_cols[0];
        for (int s = 0; s < max_cols; s++) inl.cols[s] = cols[s];
    }
    const auto T1 = trace ? std::chrono::steady_clock::now() : T0;
    int rc = nq_step_dev(ctx, R, n, si.h<int>(o_rows), si.h<long long>(o_cur), si.h<int>(o_cols), si.h<int>(o_nc), max_cols, win,
                         so.h<ls_nq_result>(o_res), nullptr, nullptr, nullptr, &x, &inl);
    if (rc != LS_OK) { std::fill(rs.valid.begin(), rs.valid.end(), 0); return rc; }
    const auto T2 = trace ? std::chrono::steady_clock::now() : T0;
    rc = wait_done_word(ctx, rs.flag, rs.seq);
    if (rc != LS_OK) { std::fill(rs.valid.begin(), rs.valid.end(), 0); return rc; }
    const auto T3 = trace ? std::chrono::steady_clock::now() : T0;
    memcpy(results, so.h<char>(o_res), res_b);
    for (int r = 0; r < R; r++) {
        int32_t *sh = rs.shadow.data() + (size_t)r * n, *cs = rs.cs.data() + (size_t)r * n;
        const bool moved = results[r].best_index >= 0;
        const int mc = moved ? results[r].best_col : 0, mv = moved ? results[r].best_value : 0;
        if (cs_b) {
            if (r >= 64 || ((want_cs >> r) & 1ull)) { // from the kernel; remembered for the next call
                const unsigned short *k16 = so.h<unsigned short>(o_cs) + (size_t)r * n;
                widen_u16(k16, next_col_scores + (size_t)r * n, n);
                for (int j = 0; j < n; j++) cs[j] = k16[j];
                rs.cs_valid[(size_t)r] = r < 64;
            } else {
                if (moved) nq_cs_after_move_widen(sh, cs, next_col_scores + (size_t)r * n, n, mc, mv, sh[mc]);
                else widen_i32(cs, next_col_scores + (size_t)r * n, n);
            }
        }
        if (moved) sh[mc] = mv; // the device applied it
    }
    if (trace) {
        const auto T4 = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        rs.t_pre += us(T0, T1); rs.t_launch += us(T1, T2); rs.t_wait += us(T2, T3); rs.t_post += us(T3, T4);
        rs.calls++;
    }
    return LS_OK;
}

// Shared host-pointer path: validate, pack the inputs into the pinned staging block (one H2D copy), run the
// neighbourhood (+ optionally the winner's column scores), fetch results (+ column scores) with one D2H copy.
// Only the parity-mode per-candidate score array (up to GBs) is copied separately.
static int nq_host_call(ls_ctx *ctx, int32_t n_restarts, int32_t n, const int32_t *rows, const int64_t *cur_scores,
                        const int32_t *cols, const int32_t *n_cols, int32_t max_cols, int64_t window,
                        ls_nq_result *results, int64_t *scores_out, int64_t *next_col_scores)
{
    if (!ctx) return LS_ERR_INVALID;
    int rc = nq_check_dims(ctx, n_restarts, n, max_cols, window);
    if (rc != LS_OK) return rc;
    LS_REQUIRE(ctx, rows && cur_scores && n_cols && results && (cols || max_cols == 0), "null pointer");
    for (int r = 0; r < n_restarts; r++) {
        LS_REQUIRE(ctx, n_cols[r] >= 0 && n_cols[r] <= max_cols, "n_cols[%d] = %d outside [0, max_cols = %d]", r,
                   n_cols[r], max_cols);
        for (int s = 0; s < n_cols[r]; s++) {
            const int c = cols[(long long)r * max_cols + s];
            LS_REQUIRE(ctx, c >= 0 && c < n, "cols[%d][%d] = %d outside [0, %d)", r, s, c, n);
        }
    }
    if (!scores_out && n_restarts == 1 && ctx->nq_step_mode >= 4 && nq_step_eligible(ctx, n, max_cols, nq_clamp_window(window, max_cols, n))) {
        // One restart, no score dump: the resident LS-step server (no launch, no copy, no runtime call in the steady state; the
        // board stays in the slot's LDS).  The board is range-checked there, and only when it is not the one the slot holds.
        NqServer *srv = ctx->server && nq_server_n(ctx->server) == n ? ctx->server : nullptr;
        if (!srv) {
            LS_BIND(ctx);
            rc = nq_ctx_server(ctx, n, &srv);
            if (rc != LS_OK) return rc;
        }
        if (srv)
            return nq_server_step(ctx, srv, ctx->server_slot, rows, cur_scores[0], cols, n_cols[0], nq_clamp_window(window, max_cols, n), results,
                                  next_col_scores);
    }
    if (!scores_out && n_restarts > 1 && n_restarts <= 64 && ctx->nq_step_mode >= 4 && (!ctx->server || ctx->server_owned) &&
        nq_step_eligible(ctx, n, max_cols, nq_clamp_window(window, max_cols, n))) {
        // A batch of restarts (the reference-shaped "8 restarts per launch" of config 4, ls_multi_solver's lockstep mode): one slot of
        // the context's own server per restart, all requests in flight before the first answer is awaited -- still no launch.
        NqServer *srv = ctx->server && nq_server_n(ctx->server) == n && nq_server_slots(ctx->server) >= n_restarts ? ctx->server : nullptr;
        if (!srv) {
            LS_BIND(ctx);
            rc = nq_ctx_server(ctx, n, &srv, n_restarts);
            if (rc != LS_OK) return rc;
        }
        if (srv) {
            const long long win = nq_clamp_window(window, max_cols, n);
            for (int r = 0; r < n_restarts; r++) {
                rc = nq_server_post(ctx, srv, r, rows + (size_t)r * n, cols + (size_t)r * max_cols, n_cols[r], win, next_col_scores != nullptr);
                if (rc != LS_OK) { // rows out of range: the requests already posted are answered and dropped, the error stands
                    for (int q = 0; q < r; q++) {
                        ls_nq_result dropped;
                        (void)nq_server_finish(ctx, srv, q, cur_scores[q], cols + (size_t)q * max_cols, &dropped, nullptr);
                    }
                    return rc;
                }
            }
            int first_rc = LS_OK;
            for (int r = 0; r < n_restarts; r++) {
                rc = nq_server_finish(ctx, srv, r, cur_scores[r], cols + (size_t)r * max_cols, results + r,
                                      next_col_scores ? next_col_scores + (size_t)r * n : nullptr);
                if (rc != LS_OK && first_rc == LS_OK) first_rc = rc;
            }
            return first_rc;
        }
    }
    LS_BIND(ctx);
    if (!all_in_range(rows, (long long)n_restarts * n, n))
        for (long long i = 0; i < (long long)n_restarts * n; i++)
            LS_REQUIRE(ctx, rows[i] >= 0 && rows[i] < n, "rows[%lld] = %d outside [0, %d)", i, rows[i], n);
    const size_t rows_b = (size_t)n_restarts * n * 4, cur_b = (size_t)n_restarts * 8,
                 cols_b = (size_t)n_restarts * max_cols * 4, nc_b = (size_t)n_restarts * 4,
                 res_b = (size_t)n_restarts * sizeof(ls_nq_result), cs_b = next_col_scores ? (size_t)n_restarts * n * 8 : 0,
                 sc_b = scores_out ? (size_t)n_restarts * max_cols * n * 8 : 0;
    // inputs: cur | rows | cols | n_cols ;  outputs: results | next_col_scores
    const size_t o_cur = 0, o_rows = align256(o_cur + cur_b), o_cols = align256(o_rows + rows_b),
                 o_nc = align256(o_cols + cols_b), in_b = align256(o_nc + nc_b);
    const size_t o_res = 0, o_cs = align256(o_res + res_b), out_b = align256(o_cs + cs_b);
    {
        hipError_t e = ctx->stage_in.reserve(in_b);
        if (e == hipSuccess) e = ctx->stage_out.reserve(out_b);
        if (e != hipSuccess) return set_err(ctx, LS_ERR_NOMEM, "staging allocation failed: %s", hipGetErrorString(e));
    }
    const long long win = nq_clamp_window(window, max_cols, n);
    if (!sc_b && n_restarts <= 64 && nq_step_eligible(ctx, n, max_cols, win)) // packs its own (smaller) blocks
        return nq_host_step_resident(ctx, n_restarts, n, rows, cur_scores, cols, n_cols, max_cols, win, results, next_col_scores);
    if (cs_b) LS_RESERVE(ctx, ctx->nq_next_rows, rows_b);
    if (sc_b) LS_RESERVE(ctx, ctx->nq_scores, sc_b);
    Staging &si = ctx->stage_in, &so = ctx->stage_out;
    memcpy(si.h<char>(o_cur), cur_scores, cur_b);
    memcpy(si.h<char>(o_rows), rows, rows_b);
    if (cols_b) memcpy(si.h<char>(o_cols), cols, cols_b);
    memcpy(si.h<char>(o_nc), n_cols, nc_b);
    LS_HIP(ctx, hipMemcpyAsync(si.dev, si.host, in_b, hipMemcpyHostToDevice, ctx->stream));
    rc = nq_eval_batch_dev(ctx, n_restarts, n, si.d<int>(o_rows), si.d<long long>(o_cur), si.d<int>(o_cols), si.d<int>(o_nc),
                           max_cols, win, so.d<ls_nq_result>(o_res),
                           sc_b ? ctx->nq_scores.as<long long>() : nullptr, cs_b ? ctx->nq_next_rows.as<int>() : nullptr);
    if (rc != LS_OK) return rc;
    if (cs_b) {
        rc = nq_col_scores_dev(ctx, n_restarts, n, ctx->nq_next_rows.as<int>(), so.d<long long>(o_cs), nullptr);
        if (rc != LS_OK) return rc;
    }
    LS_HIP(ctx, hipMemcpyAsync(so.host, so.dev, out_b, hipMemcpyDeviceToHost, ctx->stream));
    if (sc_b) LS_HIP(ctx, hipMemcpyAsync(scores_out, ctx->nq_scores.p, sc_b, hipMemcpyDeviceToHost, ctx->stream));
    LS_SYNC(ctx);
    memcpy(results, so.h<char>(o_res), res_b);
    if (cs_b) memcpy(next_col_scores, so.h<char>(o_cs), cs_b);
    return LS_OK;
}

int ls_nqueens_eval_batch(ls_ctx *ctx, int32_t n_restarts, int32_t n, const int32_t *rows,
                          const int64_t *cur_scores, const int32_t *cols, const int32_t *n_cols, int32_t max_cols,
                          int64_t window, ls_nq_result *results, int64_t *scores_out)
{
    return nq_host_call(ctx, n_restarts, n, rows, cur_scores, cols, n_cols, max_cols, window, results, scores_out, nullptr);
}

int ls_nqueens_eval_batch_next_dev(ls_ctx *ctx, int32_t n_restarts, int32_t n, const int32_t *rows_d,
                                   const int64_t *cur_scores_d, const int32_t *cols_d, const int32_t *n_cols_d,
                                   int32_t max_cols, int64_t window, ls_nq_result *results_d, int32_t *next_rows_d,
                                   int64_t *next_col_scores_d, int64_t *next_scores_d)
{
    LS_BIND(ctx);
    int rc = nq_check_dims(ctx, n_restarts, n, max_cols, window);
    if (rc != LS_OK) return rc;
    LS_REQUIRE(ctx, rows_d && cur_scores_d && n_cols_d && results_d && (cols_d || max_cols == 0) && next_rows_d &&
                        next_col_scores_d, "null device pointer");
    const long long win = nq_clamp_window(window, max_cols, n);
    if (nq_step_eligible(ctx, n, max_cols, win))
        return nq_step_dev(ctx, n_restarts, n, rows_d, reinterpret_cast<const long long *>(cur_scores_d), cols_d, n_cols_d,
                           max_cols, win, results_d, next_rows_d, reinterpret_cast<long long *>(next_col_scores_d),
                           reinterpret_cast<long long *>(next_scores_d), nullptr);
    rc = nq_eval_batch_dev(ctx, n_restarts, n, rows_d, reinterpret_cast<const long long *>(cur_scores_d), cols_d, n_cols_d,
                           max_cols, win, results_d, nullptr, next_rows_d);
    if (rc != LS_OK) return rc;
    return nq_col_scores_dev(ctx, n_restarts, n, next_rows_d, reinterpret_cast<long long *>(next_col_scores_d),
                             reinterpret_cast<long long *>(next_scores_d));
}

int ls_nqueens_eval_batch_next(ls_ctx *ctx, int32_t n_restarts, int32_t n, const int32_t *rows,
                               const int64_t *cur_scores, const int32_t *cols, const int32_t *n_cols, int32_t max_cols,
                               int64_t window, ls_nq_result *results, int64_t *next_col_scores)
{
    if (ctx && !next_col_scores) return set_err(ctx, LS_ERR_INVALID, "next_col_scores is NULL");
    return nq_host_call(ctx, n_restarts, n, rows, cur_scores, cols, n_cols, max_cols, window, results, nullptr,
                        next_col_scores);
}

int ls_nqueens_eval(ls_ctx *ctx, int32_t n, const int32_t *rows, int64_t cur_score, const int32_t *cols,
                    int32_t n_cols, int64_t window, ls_nq_result *out, int64_t *scores_out)
{
    return ls_nqueens_eval_batch(ctx, 1, n, rows, &cur_score, cols, &n_cols, n_cols, window, out, scores_out);
}

int ls_nqueens_eval_tabu(ls_ctx *ctx, int32_t n, const int32_t *rows, int64_t cur_score, const int32_t *cols, int32_t n_cols,
                         int64_t window, const int64_t *tabu_positions, int32_t n_tabu, ls_nq_result *out, int64_t *scores_out)
{
    if (ctx && n_tabu == 0) return ls_nqueens_eval(ctx, n, rows, cur_score, cols, n_cols, window, out, scores_out);
    LS_BIND(ctx);
    int rc = nq_check_dims(ctx, 1, n, n_cols, window);
    if (rc != LS_OK) return rc;
    LS_REQUIRE(ctx, rows && out && (cols || n_cols == 0) && n_tabu >= 0 && tabu_positions, "null pointer / negative count");
    for (int i = 0; i < n; i++) LS_REQUIRE(ctx, rows[i] >= 0 && rows[i] < n, "rows[%d] = %d outside [0, %d)", i, rows[i], n);
    for (int s = 0; s < n_cols; s++) LS_REQUIRE(ctx, cols[s] >= 0 && cols[s] < n, "cols[%d] = %d outside [0, %d)", s, cols[s], n);
    for (int i = 0; i < n_tabu; i++) {
        const int64_t k = tabu_positions[i];
        LS_REQUIRE(ctx, k >= 0 && k < (int64_t)n_cols * n, "tabu_positions[%d] = %lld outside the candidate list", i, (long long)k);
        LS_REQUIRE(ctx, i == 0 || tabu_positions[i - 1] < k, "tabu_positions must be strictly ascending (entry %d)", i);
        LS_REQUIRE(ctx, (int)(k % n) != rows[cols[k / n]], "tabu_positions[%d] is a no-op (always tabu; do not list it)", i);
    }
    const size_t rows_b = (size_t)n * 4, cols_b = (size_t)n_cols * 4, ex_b = (size_t)n_tabu * 8,
                 sc_b = scores_out ? (size_t)n_cols * n * 8 : 0;
    // inputs: cur | n_cols | excl | rows | cols ;  output: result
    const size_t o_cur = 0, o_nc = 8, o_ex = 256, o_rows = align256(o_ex + ex_b), o_cols = align256(o_rows + rows_b),
                 in_b = align256(o_cols + cols_b);
    {
        hipError_t e = ctx->stage_in.reserve(in_b);
        if (e == hipSuccess) e = ctx->stage_out.reserve(256);
        if (e != hipSuccess) return set_err(ctx, LS_ERR_NOMEM, "staging allocation failed: %s", hipGetErrorString(e));
    }
    if (sc_b) LS_RESERVE(ctx, ctx->nq_scores, sc_b);
    Staging &si = ctx->stage_in, &so = ctx->stage_out;
    *si.h<long long>(o_cur) = cur_score;
    *si.h<int>(o_nc) = n_cols;
    memcpy(si.h<char>(o_ex), tabu_positions, ex_b);
    memcpy(si.h<char>(o_rows), rows, rows_b);
    if (cols_b) memcpy(si.h<char>(o_cols), cols, cols_b);
    LS_HIP(ctx, hipMemcpyAsync(si.dev, si.host, in_b, hipMemcpyHostToDevice, ctx->stream));
    rc = nq_eval_excl_dev(ctx, n, si.d<int>(o_rows), si.d<long long>(o_cur), si.d<int>(o_cols), si.d<int>(o_nc), n_cols,
                          nq_clamp_window(window, n_cols, n), si.d<long long>(o_ex), n_tabu, so.d<ls_nq_result>(0),
                          sc_b ? ctx->nq_scores.as<long long>() : nullptr);
    if (rc != LS_OK) return rc;
    LS_HIP(ctx, hipMemcpyAsync(so.host, so.dev, 256, hipMemcpyDeviceToHost, ctx->stream));
    if (sc_b) LS_HIP(ctx, hipMemcpyAsync(scores_out, ctx->nq_scores.p, sc_b, hipMemcpyDeviceToHost, ctx->stream));
    LS_SYNC(ctx);
    *out = *so.h<ls_nq_result>(0);
    return LS_OK;
}

int ls_nqueens_col_scores_dev(ls_ctx *ctx, int32_t n_restarts, int32_t n, const int32_t *rows_d,
                              int64_t *col_scores_d, int64_t *scores_d)
{
    LS_BIND(ctx);
    int rc = nq_check_dims(ctx, n_restarts, n, 0, 0);
    if (rc != LS_OK) return rc;
    LS_REQUIRE(ctx, rows_d != nullptr, "null device pointer");
    return nq_col_scores_dev(ctx, n_restarts, n, rows_d, reinterpret_cast<long long *>(col_scores_d),
                             reinterpret_cast<long long *>(scores_d));
}

int ls_nqueens_col_scores(ls_ctx *ctx, int32_t n_restarts, int32_t n, const int32_t *rows, int64_t *col_scores,
                          int64_t *scores)
{
    LS_BIND(ctx);
    int rc = nq_check_dims(ctx, n_restarts, n, 0, 0);
    if (rc != LS_OK) return rc;
    LS_REQUIRE(ctx, rows != nullptr, "null pointer");
    if (!all_in_range(rows, (long long)n_restarts * n, n))
        for (long long i = 0; i < (long long)n_restarts * n; i++)
            LS_REQUIRE(ctx, rows[i] >= 0 && rows[i] < n, "rows[%lld] = %d outside [0, %d)", i, rows[i], n);
    if (n_restarts == 1 && ctx->nq_step_mode >= 4 && ctx->server && nq_server_n(ctx->server) == n && nq_server_supports(ctx, n))
        return nq_server_col_scores(ctx, ctx->server, ctx->server_slot, rows, col_scores, scores); // the board stays with the slot
    const size_t rows_b = (size_t)n_restarts * n * 4, cs_b = (size_t)n_restarts * n * 8, t_b = (size_t)n_restarts * 8;
    LS_RESERVE(ctx, ctx->nq_rows, rows_b);
    LS_RESERVE(ctx, ctx->nq_colscores, cs_b);
    LS_RESERVE(ctx, ctx->nq_totals, t_b);
    LS_HIP(ctx, hipMemcpyAsync(ctx->nq_rows.p, rows, rows_b, hipMemcpyHostToDevice, ctx->stream));
    rc = nq_col_scores_dev(ctx, n_restarts, n, ctx->nq_rows.as<int>(), ctx->nq_colscores.as<long long>(),
                           ctx->nq_totals.as<long long>());
    if (rc != LS_OK) return rc;
    if (col_scores)
        LS_HIP(ctx, hipMemcpyAsync(col_scores, ctx->nq_colscores.p, cs_b, hipMemcpyDeviceToHost, ctx->stream));
    if (scores) LS_HIP(ctx, hipMemcpyAsync(scores, ctx->nq_totals.p, t_b, hipMemcpyDeviceToHost, ctx->stream));
    LS_SYNC(ctx);
    return LS_OK;
}

/* ------------------------------------------------------------ scheduling */

int ls_sched_instance_upload(ls_ctx *ctx, int32_t n_days_scored, int32_t n_days_stored, int32_t start_weekday,
                             int32_t n_employees, const uint64_t *holiday_bits)
{
    LS_BIND(ctx);
    LS_REQUIRE(ctx, n_days_scored >= 1, "n_days_scored must be >= 1 (got %d)", n_days_scored);
    if (n_days_stored > LS_SCHED_MAX_DAYS)
        return set_err(ctx, LS_ERR_UNSUPPORTED, "n_days_stored = %d exceeds LS_SCHED_MAX_DAYS = %d", n_days_stored,
                       LS_SCHED_MAX_DAYS);
    LS_REQUIRE(ctx, n_days_stored >= n_days_scored, "n_days_stored (%d) < n_days_scored (%d)", n_days_stored,
               n_days_scored);
    LS_REQUIRE(ctx, start_weekday >= 0 && start_weekday <= 6, "start_weekday %d outside [0, 6]", start_weekday);
    {   // The argmin key holds hard in 10 bits and soft in 8 (sched_kernels.hip); both are bounded by the instance length D:
        //   hard <= D (H1) + D-1 (H2) + 4*ceil(D/7) (H3: a 9-window counts only when it starts on a Saturday)
        //           + 3*(D-13) (H4: at most three employees can hold > 3 of 14 days)
        //   soft <= 2*(D-6) (S1: at most two employees hold > 2 of 7 days) + 5*floor(ceil(D/7)/2) (S2: minimum over >= 2
        //           employees of one weekday's ceil(D/7) days) + D-2 (S3) + 2*ceil(D/7) (S4: weekend days)
        // D = 64 gives 320 and 223.  Checked here so that raising LS_SCHED_MAX_DAYS cannot silently overflow the fields.
        const int D = n_days_scored, wk = (D + 6) / 7;
        const int hard_max = D + (D - 1) + 4 * wk + 3 * (D > 13 ? D - 13 : 0);
        const int soft_max = 2 * (D > 6 ? D - 6 : 0) + 5 * (wk / 2) + (D > 2 ? D - 2 : 0) + 2 * wk;
        if (hard_max > 1023 || soft_max > 255)
            return set_err(ctx, LS_ERR_UNSUPPORTED, "%d scored days: score bounds (hard %d, soft %d) exceed the argmin key fields",
                           D, hard_max, soft_max);
    }
    LS_REQUIRE(ctx, n_employees >= 1, "n_employees must be >= 1 (got %d)", n_employees);
    LS_REQUIRE(ctx, holiday_bits != nullptr, "holiday_bits is NULL");
    const uint64_t scored_mask = n_days_scored >= 64 ? ~0ull : ((1ull << n_days_scored) - 1);
    for (int e = 0; e < n_employees; e++)
        LS_REQUIRE(ctx, (holiday_bits[e] & ~scored_mask) == 0,
                   "employee %d has a holiday outside the scored range (the reference panics here: "
                   "examples/employee-scheduling/src/lib.rs:275)", e);
    // the stream (and the resident scheduling step, whose launch arguments are this instance) may still be reading the previous one
    sched_ctx_drop_servers(ctx, false);
    LS_SYNC(ctx);
    LS_RESERVE(ctx, ctx->sched.holiday_bits, (size_t)n_employees * 8);
    LS_HIP(ctx, hipMemcpyAsync(ctx->sched.holiday_bits.p, holiday_bits, (size_t)n_employees * 8,
                               hipMemcpyHostToDevice, ctx->stream));
    lsamd::SchedLaneConst tab[64];
    lsamd::sched_build_lane_table(n_days_scored, start_weekday, tab);
    LS_RESERVE(ctx, ctx->sched.lane_tab, sizeof tab);
    LS_HIP(ctx, hipMemcpyAsync(ctx->sched.lane_tab.p, tab, sizeof tab, hipMemcpyHostToDevice, ctx->stream));
    LS_SYNC(ctx);
    ctx->sched.n_days_scored = n_days_scored;
    ctx->sched.n_days_stored = n_days_stored;
    ctx->sched.start_weekday = start_weekday;
    ctx->sched.n_employees = n_employees;
    ctx->sched.valid = true;
    return LS_OK;
}

static int sched_check_state(ls_ctx *ctx, const int32_t *states, long long count)
{
    if (all_in_range(states, count, ctx->sched.n_employees)) return LS_OK;
    for (long long i = 0; i < count; i++)
        LS_REQUIRE(ctx, states[i] >= 0 && states[i] < ctx->sched.n_employees,
                   "state entry %lld = %d is not an employee index in [0, %d)", i, states[i], ctx->sched.n_employees);
    return LS_OK;
}

int ls_sched_score(ls_ctx *ctx, int32_t n_states, const int32_t *states, int64_t *hard, int64_t *soft,
                   int64_t *terms)
{
    if (!ctx) return LS_ERR_INVALID;
    if (!ctx->sched.valid) return set_err(ctx, LS_ERR_STATE, "ls_sched_instance_upload has not been called");
    LS_REQUIRE(ctx, n_states >= 1 && states && hard && soft, "bad arguments");
    if (n_states == 1 && !terms && sched_resident_ok(ctx)) { // no launch when a resident launch is there: the slot's workgroup, or the sub-grid's master
        if (ctx->sched_step) return sched_step_score(ctx, ctx->sched_step, ctx->sched_step_slot, states, hard, soft);
        if (ctx->sched_grid && !ctx->sched_grid_unavailable) {
            const int rc_g = sched_grid_score(ctx, ctx->sched_grid, ctx->sched_grid_sub, states, hard, soft);
            if (rc_g != SRV_RC_GIVE_UP) return rc_g;
            ctx->sched_grid_unavailable = true;
        }
    }
    LS_BIND(ctx);
    const int S = ctx->sched.n_days_stored;
    int rc = sched_check_state(ctx, states, (long long)n_states * S);
    if (rc != LS_OK) return rc;
    const size_t st_b = (size_t)n_states * S * 4, sc_b = (size_t)n_states * 8, tm_b = (size_t)n_states * 64;
    LS_RESERVE(ctx, ctx->sc_states, st_b);
    LS_RESERVE(ctx, ctx->sc_hard, sc_b);
    LS_RESERVE(ctx, ctx->sc_soft, sc_b);
    LS_RESERVE(ctx, ctx->sc_terms, tm_b);
    LS_HIP(ctx, hipMemcpyAsync(ctx->sc_states.p, states, st_b, hipMemcpyHostToDevice, ctx->stream));
    rc = sched_score_dev(ctx, n_states, ctx->sc_states.as<int>(), ctx->sc_hard.as<long long>(),
                         ctx->sc_soft.as<long long>(), ctx->sc_terms.as<long long>());
    if (rc != LS_OK) return rc;
    LS_HIP(ctx, hipMemcpyAsync(hard, ctx->sc_hard.p, sc_b, hipMemcpyDeviceToHost, ctx->stream));
    LS_HIP(ctx, hipMemcpyAsync(soft, ctx->sc_soft.p, sc_b, hipMemcpyDeviceToHost, ctx->stream));
    if (terms) LS_HIP(ctx, hipMemcpyAsync(terms, ctx->sc_terms.p, tm_b, hipMemcpyDeviceToHost, ctx->stream));
    LS_SYNC(ctx);
    return LS_OK;
}

int ls_sched_eval_batch_dev(ls_ctx *ctx, int32_t n_restarts, const int32_t *states_d, const ls_sched_move *moves_d,
                            const int64_t *n_moves_d, int64_t max_moves, int64_t window, ls_sched_result *results_d,
                            int64_t *hard_out_d, int64_t *soft_out_d)
{
    LS_BIND(ctx);
    if (!ctx->sched.valid) return set_err(ctx, LS_ERR_STATE, "ls_sched_instance_upload has not been called");
    LS_REQUIRE(ctx, n_restarts >= 1 && max_moves >= 0 && window >= 0, "bad sizes");
    LS_REQUIRE(ctx, states_d && n_moves_d && results_d && (moves_d || max_moves == 0), "null device pointer");
    return sched_eval_batch_dev(ctx, n_restarts, states_d, moves_d, reinterpret_cast<const long long *>(n_moves_d),
                                max_moves, window, results_d, reinterpret_cast<long long *>(hard_out_d),
                                reinterpret_cast<long long *>(soft_out_d));
}

int ls_sched_eval_batch(ls_ctx *ctx, int32_t n_restarts, const int32_t *states, const ls_sched_move *moves,
                        const int64_t *n_moves, int64_t max_moves, int64_t window, ls_sched_result *results,
                        int64_t *hard_out, int64_t *soft_out)
{
    LS_BIND(ctx);
    if (!ctx->sched.valid) return set_err(ctx, LS_ERR_STATE, "ls_sched_instance_upload has not been called");
    LS_REQUIRE(ctx, n_restarts >= 1 && max_moves >= 0 && window >= 0, "bad sizes");
    LS_REQUIRE(ctx, states && n_moves && results && (moves || max_moves == 0), "null pointer");
    const int S = ctx->sched.n_days_stored, D = ctx->sched.n_days_scored, E = ctx->sched.n_employees;
    int rc = sched_check_state(ctx, states, (long long)n_restarts * S);
    if (rc != LS_OK) return rc;
    for (int r = 0; r < n_restarts; r++) {
        LS_REQUIRE(ctx, n_moves[r] >= 0 && n_moves[r] <= max_moves, "n_moves[%d] = %lld outside [0, %lld]", r,
                   (long long)n_moves[r], (long long)max_moves);
        for (long long k = 0; k < n_moves[r]; k++) {
            const ls_sched_move &m = moves[(long long)r * max_moves + k];
            // ChangeDay may address any STORED day (the exhaustive proposer does, lib.rs:530);
            // the random proposer only addresses scored days (lib.rs:486)
            if (m.type == 0)
                LS_REQUIRE(ctx, m.a >= 0 && m.a < S && m.b >= 0 && m.b < E, "move %lld of restart %d: bad ChangeDay(%d, %d)",
                           k, r, m.a, m.b);
            else if (m.type == 1)
                LS_REQUIRE(ctx, m.a >= 0 && m.a < S && m.b >= 0 && m.b < S, "move %lld of restart %d: bad SwapDays(%d, %d)",
                           k, r, m.a, m.b);
            else if (m.type != LS_SCHED_MOVE_SKIP)
                return set_err(ctx, LS_ERR_INVALID, "move %lld of restart %d: unknown type %d", k, r, m.type);
        }
    }
    (void)D;
    const size_t st_b = (size_t)n_restarts * S * 4, mv_b = (size_t)n_restarts * max_moves * sizeof(ls_sched_move),
                 nm_b = (size_t)n_restarts * 8, res_b = (size_t)n_restarts * sizeof(ls_sched_result),
                 out_b = (size_t)n_restarts * max_moves * 8;
    if (!hard_out && !soft_out && st_b + mv_b + nm_b <= (size_t)4 << 20) {
        // Production shape (no per-candidate score dump): everything goes through the pinned staging blocks -- one
        // copy each way, or none at all for the small move lists the fused one-block launch serves (<= 512 moves:
        // the reference example's window-100 iteration), whose kernel reads its inputs in place.
        const size_t o_nm = 0, o_st = align256(o_nm + nm_b), o_mv = align256(o_st + st_b), in_b = align256(o_mv + mv_b);
        const size_t res_pad = align256(res_b);
        {
            hipError_t e = ctx->stage_in.reserve(in_b);
            if (e == hipSuccess) e = ctx->stage_out.reserve(res_pad);
            if (e != hipSuccess) return set_err(ctx, LS_ERR_NOMEM, "staging allocation failed: %s", hipGetErrorString(e));
        }
        Staging &si = ctx->stage_in, &so = ctx->stage_out;
        memcpy(si.h<char>(o_nm), n_moves, nm_b);
        memcpy(si.h<char>(o_st), states, st_b);
        if (mv_b) memcpy(si.h<char>(o_mv), moves, mv_b);
        const bool zc = max_moves <= 512;
        if (!zc) LS_HIP(ctx, hipMemcpyAsync(si.dev, si.host, in_b, hipMemcpyHostToDevice, ctx->stream));
        char *ib = static_cast<char *>(zc ? si.host : si.dev), *ob = static_cast<char *>(zc ? so.host : so.dev);
        rc = sched_eval_batch_dev(ctx, n_restarts, reinterpret_cast<int *>(ib + o_st), reinterpret_cast<ls_sched_move *>(ib + o_mv),
                                  reinterpret_cast<long long *>(ib + o_nm), max_moves, window,
                                  reinterpret_cast<ls_sched_result *>(ob), nullptr, nullptr);
        if (rc != LS_OK) return rc;
        if (!zc) LS_HIP(ctx, hipMemcpyAsync(so.host, so.dev, res_pad, hipMemcpyDeviceToHost, ctx->stream));
        LS_SYNC(ctx);
        memcpy(results, so.h<char>(0), res_b);
        return LS_OK;
    }
    LS_RESERVE(ctx, ctx->sc_states, st_b);
    LS_RESERVE(ctx, ctx->sc_moves, mv_b);
    LS_RESERVE(ctx, ctx->sc_nmoves, nm_b);
    LS_RESERVE(ctx, ctx->sc_results, res_b);
    if (hard_out) LS_RESERVE(ctx, ctx->sc_hard, out_b);
    if (soft_out) LS_RESERVE(ctx, ctx->sc_soft, out_b);
    LS_HIP(ctx, hipMemcpyAsync(ctx->sc_states.p, states, st_b, hipMemcpyHostToDevice, ctx->stream));
    if (mv_b) LS_HIP(ctx, hipMemcpyAsync(ctx->sc_moves.p, moves, mv_b, hipMemcpyHostToDevice, ctx->stream));
    LS_HIP(ctx, hipMemcpyAsync(ctx->sc_nmoves.p, n_moves, nm_b, hipMemcpyHostToDevice, ctx->stream));
    rc = sched_eval_batch_dev(ctx, n_restarts, ctx->sc_states.as<int>(), ctx->sc_moves.as<ls_sched_move>(),
                              ctx->sc_nmoves.as<long long>(), max_moves, window,
                              ctx->sc_results.as<ls_sched_result>(),
                              hard_out ? ctx->sc_hard.as<long long>() : nullptr,
                              soft_out ? ctx->sc_soft.as<long long>() : nullptr);
    if (rc != LS_OK) return rc;
    LS_HIP(ctx, hipMemcpyAsync(results, ctx->sc_results.p, res_b, hipMemcpyDeviceToHost, ctx->stream));
    if (hard_out && out_b)
        LS_HIP(ctx, hipMemcpyAsync(hard_out, ctx->sc_hard.p, out_b, hipMemcpyDeviceToHost, ctx->stream));
    if (soft_out && out_b)
        LS_HIP(ctx, hipMemcpyAsync(soft_out, ctx->sc_soft.p, out_b, hipMemcpyDeviceToHost, ctx->stream));
    LS_SYNC(ctx);
    return LS_OK;
}

static int sched_check_moves(ls_ctx *ctx, const ls_sched_move *moves, long long n, int r)
{
    const int S = ctx->sched.n_days_stored, E = ctx->sched.n_employees;
    for (long long k = 0; k < n; k++) {
        const ls_sched_move &m = moves[k];
        if (m.type == 0)
            LS_REQUIRE(ctx, m.a >= 0 && m.a < S && m.b >= 0 && m.b < E, "move %lld of restart %d: bad ChangeDay(%d, %d)", k, r,
                       m.a, m.b);
        else if (m.type == 1)
            LS_REQUIRE(ctx, m.a >= 0 && m.a < S && m.b >= 0 && m.b < S, "move %lld of restart %d: bad SwapDays(%d, %d)", k, r,
                       m.a, m.b);
        else if (m.type != LS_SCHED_MOVE_SKIP)
            return set_err(ctx, LS_ERR_INVALID, "move %lld of restart %d: unknown type %d", k, r, m.type);
    }
    return LS_OK;
}

int ls_sched_moves_upload(ls_ctx *ctx, const ls_sched_move *moves, int64_t n_moves)
{
    if (!ctx) return LS_ERR_INVALID;
    if (!ctx->sched.valid) return set_err(ctx, LS_ERR_STATE, "ls_sched_instance_upload has not been called");
    LS_REQUIRE(ctx, n_moves >= 0 && (moves || n_moves == 0), "bad arguments");
    int rc = sched_check_moves(ctx, moves, n_moves, 0);
    if (rc != LS_OK) return rc;
    // The list stays on the host: a resident workgroup reads its moves from the slot's pinned block once per list, a launched
    // pipeline gets a device copy when it first needs one (sched_resident_dev).  No runtime call, no launch paused.
    ctx->sc_resident_host.assign(moves, moves + n_moves);
    ctx->sc_resident_n = n_moves;
    ctx->sc_resident_gen++;
    return LS_OK;
}

int ls_sched_eval_resident(ls_ctx *ctx, const int32_t *state, int64_t n_moves, int64_t window, ls_sched_result *out)
{
    if (!ctx) return LS_ERR_INVALID;
    if (!ctx->sched.valid) return set_err(ctx, LS_ERR_STATE, "ls_sched_instance_upload has not been called");
    LS_REQUIRE(ctx, state && out && window >= 0, "bad arguments");
    LS_REQUIRE(ctx, n_moves >= 0 && n_moves <= ctx->sc_resident_n, "n_moves = %lld exceeds the resident list (%lld)",
               (long long)n_moves, ctx->sc_resident_n);
    const int S = ctx->sched.n_days_stored;
    // no launch and, in the steady state, no runtime call: the slot's resident workgroup (<= 512 moves) or its resident sub-grid
    if (sched_resident_ok(ctx) && ctx->sc_resident_n > 0) {
        if (n_moves <= 512) {
            SchedStepServer *sv = nullptr;
            const int rc_s = sched_ctx_step(ctx, &sv);
            if (rc_s == LS_OK) return sched_step_eval(ctx, sv, ctx->sched_step_slot, state, n_moves, window, out);
            if (rc_s != LS_ERR_UNSUPPORTED) return rc_s;
        } else {
            SchedGrid *g = nullptr;
            const int rc0 = sched_ctx_grid(ctx, false, n_moves, &g); // sized by the prefix the caller asks for (the driver keeps a longer list resident than it uses)
            if (rc0 != LS_OK) return rc0;
            if (g) {
                const int rc_g = sched_grid_eval(ctx, g, ctx->sched_grid_sub, state, n_moves, window, out);
                if (rc_g != SRV_RC_GIVE_UP) return rc_g;
                ctx->sched_grid_unavailable = true; // the grid cannot be resident now (another launch holds the CUs): the launched pipeline, from now on
            }
        }
    }
    LS_BIND(ctx);
    int rc = sched_check_state(ctx, state, S);
    if (rc != LS_OK) return rc;
    rc = sched_resident_dev(ctx);
    if (rc != LS_OK) return rc;
    {
        hipError_t e = ctx->stage_out.reserve(256);
        if (e != hipSuccess) return set_err(ctx, LS_ERR_NOMEM, "staging allocation failed: %s", hipGetErrorString(e));
    }
    // The per-iteration inputs are 136 bytes and the output 32: the state and n_moves travel in the kernel arguments (no copy,
    // no PCIe read by the kernels), the result record goes straight to pinned host memory and its writer publishes a
    // completion word, which the caller polls instead of synchronising the stream.
    NqResident &rs = ctx->nq_res; // the context's completion word is shared with the N-queens LS step (one call at a time per context)
    {
        const int rc0 = ensure_done_word(ctx);
        if (rc0 != LS_OK) return rc0;
    }
    rc = sched_eval_batch_dev(ctx, 1, nullptr, ctx->sc_resident_moves.as<ls_sched_move>(), nullptr, n_moves, window,
                              ctx->stage_out.h<ls_sched_result>(0), nullptr, nullptr, rs.flag, ++rs.seq, state, n_moves);
    if (rc != LS_OK) return rc;
    rc = wait_done_word(ctx, rs.flag, rs.seq);
    if (rc != LS_OK) return rc;
    *out = *ctx->stage_out.h<ls_sched_result>(0);
    return LS_OK;
}

int ls_sched_exhaustive_moves_dev(ls_ctx *ctx, int32_t n_restarts, const int32_t *states_d, ls_sched_move *moves_d)
{
    LS_BIND(ctx);
    if (!ctx->sched.valid) return set_err(ctx, LS_ERR_STATE, "ls_sched_instance_upload has not been called");
    LS_REQUIRE(ctx, n_restarts >= 1 && states_d && moves_d, "bad arguments");
    return sched_exhaustive_moves_dev(ctx, n_restarts, states_d, moves_d, nullptr);
}

int ls_sched_exhaustive_moves(ls_ctx *ctx, int32_t n_restarts, const int32_t *states, ls_sched_move *moves_out)
{
    LS_BIND(ctx);
    if (!ctx->sched.valid) return set_err(ctx, LS_ERR_STATE, "ls_sched_instance_upload has not been called");
    LS_REQUIRE(ctx, n_restarts >= 1 && states && moves_out, "bad arguments");
    const int S = ctx->sched.n_days_stored;
    int rc = sched_check_state(ctx, states, (long long)n_restarts * S);
    if (rc != LS_OK) return rc;
    const size_t st_b = (size_t)n_restarts * S * 4, mv_b = (size_t)n_restarts * S * ctx->sched.n_employees * sizeof(ls_sched_move);
    LS_RESERVE(ctx, ctx->sc_states, st_b);
    LS_RESERVE(ctx, ctx->sc_moves, mv_b);
    LS_HIP(ctx, hipMemcpyAsync(ctx->sc_states.p, states, st_b, hipMemcpyHostToDevice, ctx->stream));
    rc = sched_exhaustive_moves_dev(ctx, n_restarts, ctx->sc_states.as<int>(), ctx->sc_moves.as<ls_sched_move>(), nullptr);
    if (rc != LS_OK) return rc;
    LS_HIP(ctx, hipMemcpyAsync(moves_out, ctx->sc_moves.p, mv_b, hipMemcpyDeviceToHost, ctx->stream));
    LS_SYNC(ctx);
    return LS_OK;
}

int ls_sched_eval_exhaustive(ls_ctx *ctx, int32_t n_restarts, const int32_t *states, int64_t window, ls_sched_result *results)
{
    LS_BIND(ctx);
    if (!ctx->sched.valid) return set_err(ctx, LS_ERR_STATE, "ls_sched_instance_upload has not been called");
    LS_REQUIRE(ctx, n_restarts >= 1 && states && results && window >= 0, "bad arguments");
    const int S = ctx->sched.n_days_stored;
    int rc = sched_check_state(ctx, states, (long long)n_restarts * S);
    if (rc != LS_OK) return rc;
    const long long per = (long long)S * ctx->sched.n_employees;
    // one state: the resident grid generates and scores the S x E candidates without a launch (sched_grid_server_kernel<GEN>)
    if (n_restarts == 1 && sched_resident_ok(ctx)) {
        SchedGrid *g = nullptr;
        const int rc0 = sched_ctx_grid(ctx, true, per, &g);
        if (rc0 != LS_OK) return rc0;
        if (g) {
            const int rc_g = sched_grid_eval(ctx, g, ctx->sched_grid_sub, states, per, window, results);
            if (rc_g != SRV_RC_GIVE_UP) return rc_g;
            ctx->sched_grid_unavailable = true; // the grid cannot be resident now -- the launched pipeline below, from now on
        }
    }
    const size_t st_b = (size_t)n_restarts * S * 4, mv_b = (size_t)n_restarts * per * sizeof(ls_sched_move),
                 nm_b = (size_t)n_restarts * 8, res_b = (size_t)n_restarts * sizeof(ls_sched_result);
    bool handled = false;
    if (n_restarts == 1 && ctx->sched_delta) {
        // one restart (config 5's per-GPU shape): the state travels in the kernel arguments, the result record goes
        // straight to pinned host memory and its writer publishes the completion word the caller polls
        hipError_t e = ctx->stage_out.reserve(256);
        if (e != hipSuccess) return set_err(ctx, LS_ERR_NOMEM, "staging allocation failed: %s", hipGetErrorString(e));
        NqResident &rs = ctx->nq_res;
        {
            const int rc0 = ensure_done_word(ctx);
            if (rc0 != LS_OK) return rc0;
        }
        rc = sched_eval_exhaustive_dev(ctx, 1, nullptr, window, ctx->stage_out.h<ls_sched_result>(0), &handled, rs.flag, ++rs.seq, states);
        if (rc != LS_OK) return rc;
        rc = wait_done_word(ctx, rs.flag, rs.seq);
        if (rc != LS_OK) return rc;
        *results = *ctx->stage_out.h<ls_sched_result>(0);
        return LS_OK;
    }
    LS_RESERVE(ctx, ctx->sc_states, st_b);
    LS_RESERVE(ctx, ctx->sc_results, res_b);
    LS_HIP(ctx, hipMemcpyAsync(ctx->sc_states.p, states, st_b, hipMemcpyHostToDevice, ctx->stream));
    // nothing but the states crosses PCIe.  Delta scorer: the candidates are generated inside the scoring kernels ...
    rc = sched_eval_exhaustive_dev(ctx, n_restarts, ctx->sc_states.as<int>(), window, ctx->sc_results.as<ls_sched_result>(), &handled);
    if (rc != LS_OK) return rc;
    if (handled) {
        LS_HIP(ctx, hipMemcpyAsync(results, ctx->sc_results.p, res_b, hipMemcpyDeviceToHost, ctx->stream));
        LS_SYNC(ctx);
        return LS_OK;
    }
    // ... wavefront-per-candidate kernels (LS_SCHED_DELTA=0): the move list is written on the device first
    LS_RESERVE(ctx, ctx->sc_moves, mv_b);
    LS_RESERVE(ctx, ctx->sc_nmoves, nm_b);
    rc = sched_exhaustive_moves_dev(ctx, n_restarts, ctx->sc_states.as<int>(), ctx->sc_moves.as<ls_sched_move>(),
                                    ctx->sc_nmoves.as<long long>());
    if (rc != LS_OK) return rc;
    rc = sched_eval_batch_dev(ctx, n_restarts, ctx->sc_states.as<int>(), ctx->sc_moves.as<ls_sched_move>(),
                              ctx->sc_nmoves.as<long long>(), per, window, ctx->sc_results.as<ls_sched_result>(), nullptr, nullptr);
    if (rc != LS_OK) return rc;
    LS_HIP(ctx, hipMemcpyAsync(results, ctx->sc_results.p, res_b, hipMemcpyDeviceToHost, ctx->stream));
    LS_SYNC(ctx);
    return LS_OK;
}

int ls_sched_eval(ls_ctx *ctx, const int32_t *state, const ls_sched_move *moves, int64_t n_moves, int64_t window,
                  ls_sched_result *out, int64_t *hard_out, int64_t *soft_out)
{
    return ls_sched_eval_batch(ctx, 1, state, moves, &n_moves, n_moves, window, out, hard_out, soft_out);
}

} // extern "C"
