// sched_server.hip — the resident scheduling LS step (srv_common.hpp: mailbox protocol, leaving rules): ONE launch that is not
// relaunched, ONE workgroup of eight wavefronts PER SLOT.  A slot is one restart's neighbourhood pipeline
// (local-search/src/local_search.rs:315-325, plug-in examples/employee-scheduling/src/lib.rs) for move lists of at most 512
// entries -- the reference example's window-100 iteration (examples/employee-scheduling/src/main.rs:26) -- and its single-state
// scores (get_scored_solution, lib.rs:261-375).
//
// A context's own server has one slot (ls_sched_eval_resident / ls_sched_score of one state); ls_multi_solver_create_sched runs
// MANY restarts per GPU as the slots of one shared launch ("the restarts on a GPU are batched"; the seed convention
// of examples/nqueens/src/main.rs:167; per restart the driver of examples/employee-scheduling/src/lib.rs:57-117).
//
// What a workgroup keeps between requests: the slot's state vector in LDS and, one move per lane in REGISTERS, the slot's move
// stream -- the random proposer's stream never changes during a run (its iterator owns a clone of the generator, lib.rs:488),
// so it is read once per restart from the slot's pinned block (the request names the stream's id; a new restart on the slot is a
// new id).  A request is {flags, n_moves, window, state id, stream id}; the winner is applied to the LDS state by the workgroup
// and to the pinned shadow by the host (local_search.rs:335 always moves to the winner), so a steady-state LS iteration moves 64
// bytes down and 32 bytes up.  A state the workgroup does not hold (perturbation, relaunch) is read from the shadow.
// Body = sched_eval_small_kernel (sched_delta.hip): ballots place take(window), three wavefronts tabulate the base state, one
// candidate per lane, block minimum.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>

#include "sched_device.hpp"
#include "srv_common.hpp"

namespace lsamd {

constexpr unsigned SCS_F_EVAL = 1u, SCS_F_SCORE = 2u; // neighbourhood of the state | score of the state (bit 2 is SRV_F_QUIT)

struct SchedSrvArgs {
    SrvLink link;
    SchedInst in;
    const ls_sched_move *streams; // pinned host: [n_slots][SCS_CHUNK] the slots' move streams
    const int *shadow;            // pinned host: [n_slots][64] the state of each slot's current request
    int emp_entries;              // entries of the employee -> scored-days table in dynamic LDS (0: none)
    int hol_entries;              // entries of the copy of the holiday words behind it (0: gathered from global memory)
};

template <typename M, int NW>
__global__ __launch_bounds__(64 * NW) void sched_server_kernel(SchedSrvArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int slot = (int)blockIdx.x;
    const int S = a.in.S, D = a.in.D, E = a.in.n_employees;
    const unsigned *req = a.link.req + (size_t)slot * 16;
    unsigned *resp = a.link.resp + (size_t)slot * 16;
    const int *shadow = a.shadow + (size_t)slot * 64;
    const ls_sched_move *stream = a.streams + (size_t)slot * (64 * NW);
    __shared__ SchedBaseTab<M> T;
    __shared__ u64 s_key[NW];
    __shared__ unsigned s_idx[NW];
    __shared__ int s_wcnt[NW];
    __shared__ unsigned s_req[16];
    __shared__ int s_state[64];
    __shared__ int s_leave;
    __shared__ int s_win[3]; // the winner's move, for the state update
    __shared__ __align__(16) unsigned s_mv[3 * 64 * NW]; // staging of the slot's stream when it is fetched (once per restart)
    // employee -> scored days of the current state (sched_device.hpp): zero between requests
    extern __shared__ __align__(16) unsigned char s_dyn[];
    M *const emp = a.emp_entries > 0 ? reinterpret_cast<M *>(s_dyn) : nullptr;
    for (int i = tid; i < a.emp_entries; i += 64 * NW) emp[i] = 0;
    M *const hol = a.hol_entries > 0 ? reinterpret_cast<M *>(s_dyn) + a.emp_entries : nullptr; // (sched_grid.hip: the holiday words beside it)
    for (int i = tid; i < a.hol_entries; i += 64 * NW) hol[i] = (M)a.in.holiday_bits[i];
    if (a.hol_entries > 0) __syncthreads(); // (a request's first use may come before any other barrier)
    SchedLaneConst L{};
    static_assert(NW >= SCHED_TAB_PARTS, "the base table is built by SCHED_TAB_PARTS wavefronts side by side");
    if (wib < SCHED_TAB_PARTS) L = a.in.lane_tab[lane];
    int mtype = -1, mma = 0, mmb = 0; // this lane's move of the slot's stream
    unsigned lds_state = 0, my_stream = 0; // ids of the state in s_state / of the stream in the lanes (0: none); block-uniform
    SrvPoll poll{0, 0, 0};
    if (wib == 0) srv_poll_begin(poll, a.link, resp, lane);
#ifdef LS_NQ_STAMPS
    unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0}; // wave 0: request seen, state + stream in, table built, candidates scored, block minimum, answer
#define SCS_STAMP(I) do { stamp[I] = wall_clock64(); } while (0)
#else
#define SCS_STAMP(I) do { } while (0)
#endif
    for (;;) {
        if (wib == 0) {
            const int leave = srv_poll_wait(poll, a.link, req, s_req, lane);
            if (lane == 0) s_leave = leave;
            SCS_STAMP(0);
        }
        __syncthreads(); // B1: the request is in s_req, the previous winner is in s_state
        if (s_leave) break;
        const long long nm = (long long)s_req[1] < 64 * NW ? (long long)s_req[1] : 64 * NW;
        const long long window = (long long)s_req[2];
        const unsigned state_id = s_req[3] | (s_req[4] << 16);
        const unsigned stream_id = s_req[5];
        const bool eval = (s_req[0] & SCS_F_EVAL) != 0;
        bool fresh = false;
        if (eval && stream_id != my_stream) { // block-uniform: a new restart on this slot (or a relaunch): this lane's move, once
            static_assert(NW == 8, "srv_fetch_moves_512 moves one 512-move chunk");
            srv_fetch_moves_512(stream, s_mv, tid);
            mtype = (int)s_mv[3 * tid]; mma = (int)s_mv[3 * tid + 1]; mmb = (int)s_mv[3 * tid + 2];
            my_stream = stream_id;
        }
        if (state_id != lds_state) { // block-uniform: not the state this workgroup holds
            if (tid < 64) s_state[tid] = tid < S ? srv_sys_load(shadow + tid) : 0;
            lds_state = state_id;
            fresh = true;
        }
        if (fresh) __syncthreads();
        const int e_lane = lane < S ? s_state[lane] : -1 - lane;
        const int hol_base = wib == 0 ? sched_holiday_bit_of<M>(a.in, e_lane, lane, hol) : 0;
        if (!eval) { // block-uniform: the state's own score, by the wavefront-per-state scorer (sched_score_kernel, sched_wave.hip)
            if (wib == 0) {
                const u64 eq = sched_build_eq(D, e_lane) & L.scored_mask;
                const SchedScore r = sched_terms<false>(D, L, lane, eq, hol_base);
                unsigned pay = 0;
                if (lane == 3) pay = (unsigned)r.hard;
                else if (lane == 4) pay = (unsigned)r.soft;
                if (lane < 8) __hip_atomic_store(resp + lane, (poll.last_tag << 16) | (pay & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                srv_served(poll, a.link, lane);
            }
            continue;
        }
        SCS_STAMP(1);
        const long long k = tid;
        const SchedEdit ed = sched_edit_of_lds(s_state, S, E, k < nm ? mtype : -1, mma, mmb);
        const u64 bal = __ballot(ed.active);
        if (lane == 0) s_wcnt[wib] = __popcll(bal);
        const int hbits = sched_hbits_of<M>(a.in, ed, sched_hol_word<M>(a.in, ed.new1, hol), sched_hol_word<M>(a.in, ed.new2, hol));
        if (wib < SCHED_TAB_PARTS) sched_build_base_tab<M>(a.in, wib, e_lane, hol_base, L, lane, T, emp);
        __syncthreads();
        SCS_STAMP(2);
        int before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            if (i < wib) before += s_wcnt[i];
            total += s_wcnt[i];
        }
        const long long rank = before + __popcll(bal & ((2ull << lane) - 1)); // 1-based among the real candidates
        u64 best_key = SC_KEY_EMPTY;
        unsigned best_idx = ~0u; // position in the list (< 512)
        if (ed.active && rank <= window) {
            int hard, soft;
            best_key = sched_delta_key<M>(T, D, e_lane, ed, hbits, hard, soft, emp);
            best_idx = (unsigned)k;
        }
        SCS_STAMP(3);
        // block minimum (key, first position): the wavefronts' pairs through LDS, one barrier, then EVERY wavefront reduces the NW
        // pairs itself; the winner's lane leaves its move where wavefront 0 finds it
        {
            const u64 wk = wave_min_u64(best_key);
            const unsigned wi = wave_min_u32(best_key == wk && wk != SC_KEY_EMPTY ? best_idx : ~0u);
            if (lane == 0) { s_key[wib] = wk; s_idx[wib] = wi; }
        }
        __syncthreads();
        {
            static_assert(NW == 8, "wave_min8: one value per wavefront in lanes 0..7");
            const u64 k8 = lane < NW ? s_key[lane] : SC_KEY_EMPTY;
            best_key = wave_min8_u64(k8);
            best_idx = wave_min8_u32(lane < NW && k8 == best_key ? s_idx[lane] : ~0u);
        }
        if (best_idx == (unsigned)k && best_key != SC_KEY_EMPTY) { s_win[0] = mtype; s_win[1] = mma; s_win[2] = mmb; }
        SCS_STAMP(4);
        if (wib == 0) {
            if (emp && lane < D && e_lane >= 0) emp[e_lane] = 0; // every scorer is past the barrier above: the table is all zero again
            const bool has = best_key != SC_KEY_EMPTY;
            const long long n_scored = window <= 0 || !has ? 0 : (total < window ? total : window);
            unsigned pay = 0;
            if (lane == 0) pay = has ? 1u : 0u;
            else if (lane == 1) pay = (unsigned)n_scored;
            else if (lane == 2) pay = has ? (unsigned)best_idx : 0u;
            else if (lane == 3) pay = has ? (unsigned)(best_key >> 54) : 0u;          // hard
            else if (lane == 4) pay = has ? (unsigned)((best_key >> 46) & 0xffu) : 0u; // soft
#ifdef LS_NQ_STAMPS
            stamp[5] = wall_clock64();
            if (lane >= 8 && lane < 13) { const unsigned long long dt = stamp[lane - 7] - stamp[0]; pay = dt > 0xffffu ? 0xffffu : (unsigned)dt; } // d8..d12: phase ends, 10 ns ticks
            if (lane >= 8 && lane < 13) __hip_atomic_store(resp + lane, (poll.last_tag << 16) | (pay & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
            if (lane < 8) __hip_atomic_store(resp + lane, (poll.last_tag << 16) | (pay & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            srv_served(poll, a.link, lane);
        }
        __syncthreads(); // s_win is written
        if (best_key != SC_KEY_EMPTY && tid == 0) { // the loop continues from the winner (local_search.rs:335)
            const int ty = s_win[0], ma = s_win[1], mb = s_win[2];
            if (ty == 0) s_state[ma] = mb;
            else { const int ea = s_state[ma], eb = s_state[mb]; s_state[ma] = eb; s_state[mb] = ea; }
        }
    }
    if (tid == 0) srv_block_leaves(a.link);
}

// ------------------------------------------------------------------------------------------------------------ host side
struct SchedStepServer {
    SrvHost h;
    ls_ctx *owner = nullptr;      // the context whose instance (holiday words, lane table) the launch carries
    int n_slots = 0;
    struct alignas(64) Slot { // (a slot is one host thread's: no line shared with a neighbour)
        unsigned state_id = 0, stream_id = 0;
        bool valid = false;
        const ls_ctx *stream_ctx = nullptr;   // whose list, and which generation of it, the slot's pinned block holds
        unsigned long long stream_gen = 0;
        double t_wait = 0, t_call = 0, t_score = 0; // LS_AMD_TRACE
        unsigned long long calls = 0, score_calls = 0, dev_ticks[5] = {0, 0, 0, 0, 0};
    };
    std::vector<Slot> slots;
    int *shadow(int s) const { return reinterpret_cast<int *>(h.extra()) + (size_t)s * 64; }
    ls_sched_move *stream(int s) const { return reinterpret_cast<ls_sched_move *>(h.extra() + (size_t)n_slots * 256) + (size_t)s * SCS_CHUNK; }
};

static const void *step_fn(bool m32)
{
    return m32 ? reinterpret_cast<const void *>(&sched_server_kernel<unsigned, SCS_NW>)
               : reinterpret_cast<const void *>(&sched_server_kernel<u64, SCS_NW>);
}

// How many slots one resident launch can hold on this device: every workgroup must be resident for as long as the launch
// lives (a workgroup that never starts never answers), so the count is bounded by what the CUs hold at once.
int sched_step_max_slots(ls_ctx *ctx)
{
    const bool m32 = sched_m32(ctx);
    int hol_entries = 0;
    const size_t dyn = sched_lds_tables_bytes(ctx, step_fn(m32), m32, (size_t)40 << 10, (size_t)40 << 10, &hol_entries);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, step_fn(m32), 64 * SCS_NW, dyn) != hipSuccess) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    const long long cap = (long long)(per_cu < 1 ? 1 : per_cu) * ctx->n_cus - 8; // headroom: other contexts' servers, copy kernels
    return (int)(cap < 1 ? 1 : (cap > 1024 ? 1024 : cap));
}

int sched_step_create(ls_ctx *ctx, int n_slots, SchedStepServer **out)
{
    *out = nullptr;
    if (!ctx->sched.valid) return set_err(ctx, LS_ERR_STATE, "ls_sched_instance_upload has not been called");
    LS_HIP(ctx, hipSetDevice(ctx->device));
    if (n_slots < 1 || n_slots > sched_step_max_slots(ctx))
        return set_err(ctx, LS_ERR_UNSUPPORTED, "scheduling LS-step server: %d slots, the device holds at most %d resident workgroups of it",
                       n_slots, sched_step_max_slots(ctx));
    SchedStepServer *s = new (std::nothrow) SchedStepServer();
    if (!s) return set_err(ctx, LS_ERR_NOMEM, "out of host memory");
    s->owner = ctx;
    s->n_slots = n_slots;
    const int rc = s->h.init(ctx, "scheduling LS-step server", n_slots, (size_t)n_slots * (256 + SCS_CHUNK * sizeof(ls_sched_move)));
    if (rc != LS_OK) { delete s; return rc; }
    s->slots.resize((size_t)n_slots);
    s->h.launch = [s](const SrvLink &L) {
        ls_ctx *ctx = s->owner;
        SchedSrvArgs a;
        a.link = L;
        a.in = make_inst(ctx);
        a.streams = s->stream(0);
        a.shadow = s->shadow(0);
        const bool m32 = sched_m32(ctx);
        const size_t dyn = sched_lds_tables_bytes(ctx, step_fn(m32), m32, (size_t)40 << 10, (size_t)40 << 10, &a.hol_entries);
        a.emp_entries = dyn ? ctx->sched.n_employees : 0;
        void *args[] = {&a};
        return hipLaunchKernel(step_fn(m32), dim3((unsigned)s->n_slots), dim3(64 * SCS_NW), args, dyn, s->h.stream);
    };
    *out = s;
    return LS_OK;
}

void sched_step_destroy(SchedStepServer *s)
{
    if (!s) return;
    if (getenv("LS_AMD_TRACE")) {
        double wait = 0, call = 0, score = 0;
        unsigned long long calls = 0, scores = 0, ticks[5] = {0, 0, 0, 0, 0};
        for (const auto &sl : s->slots) {
            wait += sl.t_wait; call += sl.t_call; calls += sl.calls; score += sl.t_score; scores += sl.score_calls;
            for (int d = 0; d < 5; d++) ticks[d] += sl.dev_ticks[d];
        }
        if (scores) fprintf(stderr, "scheduling step server (%d slots): %llu single-state scores, %.2f us per call\n", s->n_slots, scores, score / scores * 1e6);
        if (calls) {
            fprintf(stderr, "scheduling step server (%d slots): %llu steps, answer wait %.2f us per step (whole call %.2f us), %llu launches\n",
                    s->n_slots, calls, wait / calls * 1e6, call / calls * 1e6, s->h.relaunches.load());
            if (ticks[4])
                fprintf(stderr, "  workgroup, after the request was seen: state + stream in %.2f us, table built %.2f, candidates scored %.2f, block minimum %.2f, "
                                "answer written %.2f\n", ticks[0] * 0.01 / calls, ticks[1] * 0.01 / calls, ticks[2] * 0.01 / calls, ticks[3] * 0.01 / calls,
                        ticks[4] * 0.01 / calls);
        }
    }
    s->h.shutdown();
    delete s;
}

int sched_step_slots(const SchedStepServer *s) { return s->n_slots; }
void sched_step_pause(SchedStepServer *s) { for (auto &sl : s->slots) sl.valid = false; s->h.pause(); }
void sched_step_counters(const SchedStepServer *s, unsigned long long out[3])
{
    out[0] = s->h.requests_total(); out[1] = s->h.relaunches.load(); out[2] = s->h.slow_waits.load();
}

// the state of this request versus what the slot's workgroup holds: a new one is range-checked, copied to the pinned shadow, gets a new id
static int step_sync_state(ls_ctx *ctx, SchedStepServer *s, int slot, const int32_t *state)
{
    const int S = ctx->sched.n_days_stored, E = ctx->sched.n_employees;
    SchedStepServer::Slot &sl = s->slots[(size_t)slot];
    int *sh = s->shadow(slot);
    if (!sl.valid || memcmp(sh, state, (size_t)S * 4) != 0) {
        for (int d = 0; d < S; d++)
            if (state[d] < 0 || state[d] >= E)
                return set_err(ctx, LS_ERR_INVALID, "state entry %d = %d is not an employee index in [0, %d)", d, state[d], E);
        memcpy(sh, state, (size_t)S * 4);
        sl.valid = true;
        sl.state_id = sl.state_id == 0xffffffffu ? 1u : sl.state_id + 1u;
    }
    return LS_OK;
}

static int step_check_slot(ls_ctx *ctx, const SchedStepServer *s, int slot)
{
    if (slot < 0 || slot >= s->n_slots) return set_err(ctx, LS_ERR_INVALID, "scheduling LS-step server: slot %d outside [0, %d)", slot, s->n_slots);
    return LS_OK;
}

// ls_sched_score for one state through the slot's workgroup (the LS walk that follows then starts from a state it holds)
int sched_step_score(ls_ctx *ctx, SchedStepServer *s, int slot, const int32_t *state, int64_t *hard, int64_t *soft)
{
    static const bool trace = getenv("LS_AMD_TRACE") != nullptr;
    const auto T0 = trace ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    int rc = step_check_slot(ctx, s, slot);
    if (rc != LS_OK) return rc;
    rc = step_sync_state(ctx, s, slot, state);
    if (rc != LS_OK) return rc;
    const SchedStepServer::Slot &sl = s->slots[(size_t)slot];
    unsigned pay[16] = {0};
    pay[0] = SCS_F_SCORE;
    pay[3] = sl.state_id & 0xffffu; pay[4] = sl.state_id >> 16;
    rc = s->h.ensure_running(ctx);
    if (rc != LS_OK) return rc;
    s->h.post(slot, pay);
    unsigned r[8];
    rc = s->h.wait(ctx, slot, r);
    if (rc != LS_OK) { s->slots[(size_t)slot].valid = false; return rc; }
    *hard = r[3];
    *soft = r[4];
    if (trace) {
        s->slots[(size_t)slot].t_score += std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
        s->slots[(size_t)slot].score_calls++;
    }
    return LS_OK;
}

// ls_sched_eval_resident through the slot's workgroup: the first n_moves (<= 512) entries of the context's resident list
// against `state` (range-checked here when it is not the one the workgroup holds)
int sched_step_eval(ls_ctx *ctx, SchedStepServer *s, int slot, const int32_t *state, long long n_moves, long long window, ls_sched_result *out)
{
    static const bool trace = getenv("LS_AMD_TRACE") != nullptr;
    const auto T0 = trace ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    int rc = step_check_slot(ctx, s, slot);
    if (rc != LS_OK) return rc;
    SchedStepServer::Slot &sl = s->slots[(size_t)slot];
    if (sl.stream_ctx != ctx || sl.stream_gen != ctx->sc_resident_gen) { // a new list (a new restart on this slot): into the slot's pinned block, under a new id
        const size_t n = ctx->sc_resident_host.size() < (size_t)SCS_CHUNK ? ctx->sc_resident_host.size() : (size_t)SCS_CHUNK;
        ls_sched_move *dst = s->stream(slot);
        if (n) memcpy(dst, ctx->sc_resident_host.data(), n * sizeof(ls_sched_move));
        for (size_t i = n; i < (size_t)SCS_CHUNK; i++) { dst[i].type = -1; dst[i].a = 0; dst[i].b = 0; }
        sl.stream_ctx = ctx;
        sl.stream_gen = ctx->sc_resident_gen;
        sl.stream_id = sl.stream_id >= 0xfffeu ? 1u : sl.stream_id + 1u;
    }
    rc = step_sync_state(ctx, s, slot, state);
    if (rc != LS_OK) return rc;
    ctx->steps_done++;
    int *sh = s->shadow(slot);
    unsigned pay[16] = {0};
    pay[0] = SCS_F_EVAL;
    pay[1] = (unsigned)n_moves;
    pay[2] = (unsigned)(window > 0xffff ? 0xffff : window); // n_moves <= 512: a larger window changes nothing
    pay[3] = sl.state_id & 0xffffu; pay[4] = sl.state_id >> 16;
    pay[5] = sl.stream_id;
    rc = s->h.ensure_running(ctx);
    if (rc != LS_OK) return rc;
    s->h.post(slot, pay);
    const auto T1 = trace ? std::chrono::steady_clock::now() : T0;
    unsigned r[8];
    rc = s->h.wait(ctx, slot, r);
    if (rc != LS_OK) { sl.valid = false; return rc; }
    if (!(r[0] & 1u)) {
        out->n_scored = 0; out->best_index = -1; out->best_hard = 0; out->best_soft = 0;
    } else {
        out->n_scored = r[1]; out->best_index = r[2]; out->best_hard = r[3]; out->best_soft = r[4];
        const ls_sched_move &m = ctx->sc_resident_host[(size_t)r[2]]; // the workgroup applied it to its state
        if (m.type == 0) sh[m.a] = m.b;
        else { const int ea = sh[m.a], eb = sh[m.b]; sh[m.a] = eb; sh[m.b] = ea; }
    }
    if (trace) {
        const auto T2 = std::chrono::steady_clock::now();
        sl.t_wait += std::chrono::duration<double>(T2 - T1).count();
        sl.t_call += std::chrono::duration<double>(T2 - T0).count();
        sl.calls++;
        const volatile unsigned *rl = s->h.resp(slot);
        for (int d = 0; d < 5; d++) sl.dev_ticks[d] += (rl[8 + d] >> 16) == s->h.per[(size_t)slot].tag ? rl[8 + d] & 0xffffu : 0u;
    }
    return LS_OK;
}

} // namespace lsamd
