// nq_server.hip — the resident LS-step server: ONE launch per device whose workgroups are the slots.
//
// What it replaces: one kernel launch + dispatch + completion poll per LS iteration (local_search.rs:309-341 calls the
// neighbourhood pipeline :315-325 once per iteration; at n = 4096 that was a 6 us kernel inside a 14.4 us round trip,
// DESIGN 3.3b).  Here workgroup s of a never-relaunched kernel owns slot s: it keeps the slot's board and count tables in
// ITS LDS between requests, polls a 64-byte mailbox line in coherent pinned host memory, runs nq_step_kernel's body on a
// request (score the candidates take(window) can reach, (score, board) minimum, apply the move to the LDS board and
// tables) and answers with one 32-byte line.  A steady-state LS iteration moves 64 bytes down and 32 bytes up and
// nothing else; a board the slot does not hold (after a perturbation, iterated_local_search.rs:192-194, or after a
// relaunch) is read once from the slot's pinned shadow copy.
//
// The mailbox protocol, the rules by which the launch leaves (QUIT, 2 ms without any request, 2 s age limit) and comes back
// on demand, and its private hardware queue are shared with the scheduling server: srv_common.hpp.
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <sched.h>

#include "nq_device.hpp"
#include "srv_common.hpp"

namespace lsamd {

constexpr int SRV_INLINE_COLS = 10;
constexpr unsigned SRV_F_STEP = 1u, SRV_F_WANT_CS = 2u, SRV_F_EXT_COLS = 8u; // + SRV_F_QUIT = 4 (srv_common.hpp)

struct NqSrvArgs {
    SrvLink link;                 // request / answer lines, control block, launch generation and leaving rules
    const int *cols_ext;          // [slots][NQ_STEP_MAX_SLOTS] pinned host
    const int *shadow;            // [slots][n_pad] pinned host: the board of the slot's current request
    unsigned *cs16;               // [slots][n_pad / 2] pinned host: column scores, two u16 per word
    int n, n_pad;
};

template <class T> __device__ __forceinline__ T sys_load(const T *p) { return srv_sys_load(p); }

// One workgroup = one slot.  Request line: d0 flags, d1 whole columns take(window) reaches, d2 candidates taken from the next
// column, d3 parts per column | steps per part << 8 (the split of the paired walk), d4-5 board id, d6-15 the first ten columns.
// Response line: d0 bit 0 "has a move", d1-2 n_scored, d3 winner slot, d4 winner value, d5-6 delta + bias (STEP) or the
// board's score (no STEP).  -DLS_NQ_STAMPS builds (tools/server_stamps.py) add d7-d11: when, in 10 ns ticks after the request
// was seen, wave 0 passed B1, B2, finished scoring, passed B3 and had the answer ready.
template <bool PAIRS>
__global__ __launch_bounds__(1024) void nq_server_kernel(NqSrvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned s_dyn[]; // tables | rows[n] | slot records[NQ_STEP_MAX_SLOTS]
    __shared__ unsigned s_req[16];
    __shared__ unsigned long long s_bestmin; // the block's minimum key: one LDS atomic per wavefront (NQ_KEY_EMPTY between requests)
    __shared__ int s_sum[16], s_cnt[16];
    __shared__ int s_leave;
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    const int n = a.n;
    const int words = PAIRS ? (((6 * n + 4) >> 1) + 3) & ~3 : (((5 * n + 1) >> 1) + 3) & ~3;
    unsigned *s_t16 = s_dyn;
    int *s_rows = reinterpret_cast<int *>(s_dyn + words);
    int4 *s_slot = reinterpret_cast<int4 *>(s_rows + ((n + 3) & ~3));
    const int oR1 = n, oA = PAIRS ? 2 * n : n, oB = PAIRS ? 4 * n + 2 : 3 * n;
    const unsigned *req = a.link.req + (size_t)blockIdx.x * 16;
    unsigned *resp = a.link.resp + (size_t)blockIdx.x * 16;
    const int *shadow = a.shadow + (size_t)blockIdx.x * a.n_pad;
    const int *cols_ext = a.cols_ext + (size_t)blockIdx.x * NQ_STEP_MAX_SLOTS;
    unsigned *cs16 = a.cs16 + (size_t)blockIdx.x * (a.n_pad >> 1);

    const unsigned magic = 0xffffffffu / (unsigned)n + 1u; // ceil(2^32 / n): k / n == umulhi(k, magic) for k < 2^17, n < 2^14
    unsigned lds_board = 0;      // id of the board this workgroup holds in LDS (0: none); block-uniform
    SrvPoll poll{0, 0, 0}; // wave 0's
#ifdef LS_NQ_STAMPS
    unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0}; // wave 0: request seen, B1 passed, slot records done (B2), scored, B3 passed, answer ready
    unsigned long long cyc0 = 0;                       // shader-clock counter when the request was seen
#define SRV_STAMP(I) do { if (wib == 0) { stamp[I] = wall_clock64(); if ((I) == 0) cyc0 = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define SRV_STAMP(I) do { } while (0)
#endif
    if (wib == 0) srv_poll_begin(poll, a.link, resp, lane);
    if (tid == 0) s_bestmin = NQ_KEY_EMPTY; // published by the first B1
    for (;;) {
        if (wib == 0) {
            const int leave = srv_poll_wait(poll, a.link, req, s_req, lane);
            SRV_STAMP(0);
            if (lane == 0) s_leave = leave;
        }
        __syncthreads(); // B1: the request is in s_req; wave 0's LDS patches of the previous step are visible
        SRV_STAMP(1);
        if (s_leave) break;
        // The host does the window arithmetic (64-bit divisions cost ~600 cycles each here): d1 = whole columns take(window)
        // reaches, d2 = candidates it takes from the next column (0: none), d3 = parts a column is split into | steps per part << 8.
        const unsigned flags = s_req[0];
        const int n_whole = (int)s_req[1], rem = (int)s_req[2];
        const int parts = (int)(s_req[3] & 0xffu), per_part = (int)(s_req[3] >> 8);
        const unsigned board = s_req[4] | (s_req[5] << 16);
        const bool step = (flags & SRV_F_STEP) != 0, want_cs = (flags & SRV_F_WANT_CS) != 0;
        const int n_slots = step ? n_whole + (rem > 0 ? 1 : 0) : 0; // <= NQ_STEP_MAX_SLOTS (the host checked)
        const int window = n_whole * (n - 1) + rem;                  // < 2^24
        int cv = -1;
        if (tid < n_slots) {
            if (flags & SRV_F_EXT_COLS) cv = sys_load(cols_ext + tid);
            else if (tid < SRV_INLINE_COLS) cv = (int)s_req[6 + tid];
        }
        if (board != lds_board) { // block-uniform: the slot's board is not the one in LDS -- read it from the pinned shadow, rebuild the tables
            int rv[NQ_STEP_MAX_N / 1024];
#pragma unroll
            for (int i = 0; i < NQ_STEP_MAX_N / 1024; i++) {
                const int j = tid + 1024 * i;
                rv[i] = j < n ? sys_load(shadow + j) : -1;
            }
            for (int i = tid; i < words; i += 1024) s_t16[i] = 0;
#pragma unroll
            for (int i = 0; i < NQ_STEP_MAX_N / 1024; i++) {
                const int j = tid + 1024 * i;
                if (j < n) s_rows[j] = rv[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NQ_STEP_MAX_N / 1024; i++) {
                const int j = tid + 1024 * i, q = rv[i];
                if (j < n && (unsigned)q < (unsigned)n) {
                    tab16_inc(s_t16, q);
                    if (PAIRS && q > 0) tab16_inc(s_t16, oR1 + q - 1);
                    tab16_inc(s_t16, oA + (q - j + n));
                    tab16_inc(s_t16, oB + (q + j));
                }
            }
            if (PAIRS && tid == 0) atomicAdd(&s_t16[(oR1 + n - 1) >> 1], NQ_PAIR_SENTINEL << (16 * ((oR1 + n - 1) & 1)));
            lds_board = board;
            __syncthreads();
        }
        // Inline columns on the paired layout (every reference-shaped request: window 5n reaches six columns): no slot-record phase
        // and no barrier for it -- a wavefront derives the record of the column it walks itself (two dependent LDS round trips),
        // and n_scored is the window: the host has range-checked the columns and the board, no slot is ever skipped here.
        const bool direct = PAIRS && !(flags & SRV_F_EXT_COLS);
        auto slot_record = [&](int slot, int &c, int &rc, int &conf_old) {
            if (direct) {
                c = (int)s_req[6 + slot];
                rc = s_rows[c];
                conf_old = tab16_get(s_t16, rc) + tab16_get(s_t16, oA + rc - c + n) + tab16_get(s_t16, oB + rc + c) - 3;
            } else {
                const int4 si = s_slot[slot];
                c = si.x; rc = si.y; conf_old = si.z;
            }
        };
        if (!direct) { // block-uniform
            int my_scored = 0;
            if (tid < n_slots) {
                const int c = cv;
                const int rc = (unsigned)c < (unsigned)n ? s_rows[c] : -1;
                int conf_old = 0;
                if ((unsigned)rc < (unsigned)n)
                    conf_old = tab16_get(s_t16, rc) + tab16_get(s_t16, oA + rc - c + n) + tab16_get(s_t16, oB + rc + c) - 3;
                s_slot[tid] = make_int4(c, (unsigned)rc < (unsigned)n ? rc : -1, conf_old, 0);
                if ((unsigned)rc < (unsigned)n) my_scored = tid < n_whole ? n - 1 : rem; // a skipped slot keeps its window positions (nq_slot_scored)
            }
            if (n_slots > 64 * wib) my_scored = wave_sum_i32(my_scored); // wave-uniform
            if (lane == 0) s_cnt[wib] = my_scored;
            __syncthreads(); // B2
        }
        SRV_STAMP(2);
        unsigned long long best = NQ_KEY_EMPTY;
        if (step) { // block-uniform
            if constexpr (PAIRS) {
                const int team = __builtin_amdgcn_readfirstlane(wib);
                const int units = n_whole * parts;
                for (int u = team; u < units; u += 16) {
                    const int slot = u / parts, part = u - slot * parts;
                    int c0, rc0, co0;
                    slot_record(slot, c0, rc0, co0);
                    const int c = __builtin_amdgcn_readfirstlane(c0), rc = __builtin_amdgcn_readfirstlane(rc0),
                              conf_old = __builtin_amdgcn_readfirstlane(co0);
                    if (rc < 0) continue;
                    nq_pairs_unit<64, false>(s_t16, n, lane, c, rc, conf_old, slot, part * per_part, per_part, nullptr, 0, best);
                }
                // the slot the window cuts: per-range loops.  A short remainder (window 5n leaves FIVE candidates for the sixth
                // column) is one wavefront's job -- the one the unit round-robin reaches next, idle or least loaded -- instead of
                // three dependent LDS round trips in all sixteen; a long one is walked by all lanes.
                const bool rem_one_wave = rem <= 256;
                if (rem > 0 && (!rem_one_wave || team == (units & 15))) { // wave-uniform
                    const int slot = n_whole;
                    int c, rc, conf_old;
                    slot_record(slot, c, rc, conf_old);
                    if (rc >= 0) {
                        const unsigned short *t16 = reinterpret_cast<const unsigned short *>(s_t16);
                        const NqRanges rg = nq_ranges(0, n, rc, 0, rem); // `rem` candidates are left for this column
                        const int first = rem_one_wave ? lane : tid, stride = rem_one_wave ? 64 : 1024;
                        unsigned m = NQ_KEY32_EMPTY;
                        for (int v = rg.d0 + first; v < rg.d1; v += stride) {
                            const unsigned kk = (((unsigned)t16[v] + (unsigned)t16[oA + v - c + n] + (unsigned)t16[oB + v + c]) << 15) | (unsigned)v;
                            m = kk < m ? kk : m;
                        }
                        for (int v = rg.u0 + first; v < rg.u1; v += stride) {
                            const unsigned kk = (((unsigned)t16[v] + (unsigned)t16[oA + v - c + n] + (unsigned)t16[oB + v + c]) << 15) | (unsigned)v;
                            m = kk < m ? kk : m;
                        }
                        if (m != NQ_KEY32_EMPTY) {
                            const int v = (int)(m & 0x7fffu);
                            best = u64min(best, nq_key((int)(m >> 15) - conf_old, nq_key_hi(v > rc ? 1 : 0, c, n, slot), v));
                        }
                    }
                }
            } else {
                const int kmax = n_whole * n + (rem > 0 ? rem + 1 : 0);   // unfiltered positions take(window) can reach (nq_k_limit)
#pragma unroll 4
                for (int k = tid; k < kmax; k += 1024) {
                    const int slot = (int)__umulhi((unsigned)k, magic);
                    const int v = k - slot * n;
                    const int4 si = s_slot[slot];
                    const int c = si.x, rc = si.y;
                    const bool up = v > rc;
                    const int rank = slot * (n - 1) + v - (up ? 1 : 0);
                    if (rc >= 0 && v != rc && rank < window) {
                        const int delta = tab16_get(s_t16, v) + tab16_get(s_t16, oA + v - c + n) + tab16_get(s_t16, oB + v + c) - si.z;
                        best = u64min(best, nq_key(delta, nq_key_hi(up ? 1 : 0, c, n, slot), v));
                    }
                }
            }
        }
        best = wave_min_u64(best);
        if (lane == 0 && best != NQ_KEY_EMPTY) __hip_atomic_fetch_min(&s_bestmin, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        SRV_STAMP(3);
        __syncthreads(); // B3
        SRV_STAMP(4);
        if (!want_cs && wib != 0) continue; // the other wavefronts are done: wave 0 answers, then patches the LDS state
        const unsigned long long key = s_bestmin;
        int n_scored = direct ? window : 0;
        if (!direct) {
#pragma unroll
            for (int i = 0; i < 16; i++) n_scored += s_cnt[i];
        }
        const bool has_move = key != NQ_KEY_EMPTY && n_scored != 0;
        int mc = -1, mv = -1, mq = -1;
        if (has_move) {
            mv = (int)((key >> 15) & 0x7fffu);
            const int colkey = (int)((key >> 30) & 0x7fffu);
            mc = ((key >> 45) & 1u) ? n - 1 - colkey : colkey;
            mq = s_rows[mc];
        }
        long long total = 0;
        if (want_cs) { // block-uniform: column scores of the board AFTER the move, from the old tables plus the move's two squares
            int local = 0;
            for (int p = tid; 2 * p < n; p += 1024) {
                unsigned packed = 0;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int j = 2 * p + h;
                    int sc = 0;
                    if (j < n) {
                        const int q = s_rows[j];
                        if (j == mc) {
                            sc = tab16_get(s_t16, mv) + tab16_get(s_t16, oA + mv - mc + n) + tab16_get(s_t16, oB + mv + mc);
                        } else if ((unsigned)q < (unsigned)n) {
                            sc = tab16_get(s_t16, q) + tab16_get(s_t16, oA + q - j + n) + tab16_get(s_t16, oB + q + j) - 3;
                            if (has_move) {
                                const int dc = j > mc ? j - mc : mc - j;
                                const int dn = q > mv ? q - mv : mv - q, dq = q > mq ? q - mq : mq - q;
                                sc += ((dn == 0) | (dn == dc)) - ((dq == 0) | (dq == dc));
                            }
                        }
                    }
                    packed |= (unsigned)(sc & 0xffff) << (16 * h);
                    local += sc;
                }
                __hip_atomic_store(cs16 + p, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            const int ws = wave_sum_i32(local);
            if (lane == 0) s_sum[wib] = ws;
            __threadfence_system(); // the column scores are in host memory before the line that announces them
            __syncthreads();
            for (int i = 0; i < 16; i++) total += s_sum[i];
            if (wib != 0) continue;
        }
        // wave 0: the response line, then the move goes into the LDS board and tables
        {
            if (lane == 0) s_bestmin = NQ_KEY_EMPTY; // every reader of this request's minimum has passed (B3, and B4 with column scores)
            const unsigned dl = step ? (unsigned)((int)(key >> 46)) : (unsigned)total; // delta + bias | the board's score
            unsigned pay = 0;
            if (lane == 0) pay = has_move ? 1u : 0u;
            else if (lane == 1) pay = (unsigned)n_scored & 0xffffu;
            else if (lane == 2) pay = (unsigned)n_scored >> 16;
            else if (lane == 3) pay = has_move ? (unsigned)(key & 0x7fffu) : 0u;
            else if (lane == 4) pay = has_move ? (unsigned)mv : 0u;
            else if (lane == 5) pay = (has_move || !step) ? dl & 0xffffu : 0u;
            else if (lane == 6) pay = (has_move || !step) ? dl >> 16 : 0u;
#ifdef LS_NQ_STAMPS
            stamp[5] = wall_clock64();
            if (lane >= 7 && lane < 12) { const unsigned long long dt = stamp[lane - 6] - stamp[0]; pay = dt > 0xffffu ? 0xffffu : (unsigned)dt; } // d7..d11: phase ends, 10 ns ticks
            if (lane == 12) { const unsigned long long dc = (__builtin_amdgcn_s_memtime() - cyc0) >> 2; pay = dc > 0xffffu ? 0xffffu : (unsigned)dc; } // d12: shader cycles / 4 of the service
            if (lane == 13) pay = poll.n_polls > 0xffffu ? 0xffffu : poll.n_polls;                                                                   // d13: polls this request took
            if (lane == 14) { const unsigned long long g = stamp[0] - poll.t_served; pay = g > 0xffffu ? 0xffffu : (unsigned)g; }                    // d14: idle gap before it, 10 ns ticks
            if (lane >= 8 && lane < 15) __hip_atomic_store(resp + lane, (poll.last_tag << 16) | (pay & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
            if (lane < 8) __hip_atomic_store(resp + lane, (poll.last_tag << 16) | (pay & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            srv_served(poll, a.link, lane);
            if (has_move && lane < 8) { // the old queen leaves (lanes 0..3), the new one arrives (4..7): row, shifted row copy, two diagonals
                const int q = lane < 4 ? mq : mv, what = lane & 3;
                int idx = -1;
                if (what == 0) idx = q;
                else if (what == 1) idx = (PAIRS && q > 0) ? oR1 + q - 1 : -1;
                else if (what == 2) idx = oA + (q - mc + n);
                else idx = oB + (q + mc);
                if (idx >= 0) {
                    if (lane < 4) tab16_dec(s_t16, idx); // the count includes the old queen: no borrow
                    else tab16_inc(s_t16, idx);
                }
                if (lane == 0) s_rows[mc] = mv;
            }
        }
    }
    if (tid == 0) srv_block_leaves(a.link);
}

// ------------------------------------------------------------------------------------------------------------ host side
struct NqServer {
    SrvHost h;                   // stream, lines, control block, (re)launch, post / wait
    int n = 0, n_pad = 0, n_slots = 0;
    bool pairs = false;
    size_t o_cols = 0, o_shadow = 0, o_cs = 0; // in h.extra(): cols_ext | shadow | cs16
    size_t lds = 0;
    std::atomic<unsigned long long> fresh_boards{0};
    struct alignas(64) Slot { // (a slot is one host thread's: no line shared with a neighbour)
        double t_pre = 0, t_wait = 0, t_post = 0; // LS_AMD_TRACE: where a call spends its time (host before / answer wait / host after)
        unsigned long long calls = 0, dev_ticks[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // stamps builds: phase ends of the workgroup's service, shader cycles / 4
        unsigned board_id = 0;
        bool valid = false, cs_valid = false;
        bool want_cs = false; // the request in flight asked the workgroup for the column scores (post -> finish)
        std::vector<int32_t> cs;
    };
    std::vector<Slot> slots;
    int *cols_ext(int s) const { return reinterpret_cast<int *>(h.extra() + o_cols) + (size_t)s * NQ_STEP_MAX_SLOTS; }
    int *shadow(int s) const { return reinterpret_cast<int *>(h.extra() + o_shadow) + (size_t)s * n_pad; }
    unsigned short *cs16(int s) const { return reinterpret_cast<unsigned short *>(h.extra() + o_cs) + (size_t)s * n_pad; }
};

bool nq_server_supports(const ls_ctx *ctx, int n)
{
    return ctx->nq_algo != LS_NQ_ALGO_SCAN && n >= 2 && n <= NQ_STEP_MAX_N;
}

int nq_server_n(const NqServer *s) { return s->n; }
int nq_server_slots(const NqServer *s) { return s->n_slots; }

void nq_server_counters(const NqServer *s, unsigned long long out[4])
{
    out[0] = s->h.requests_total(); out[1] = s->h.relaunches.load(); out[2] = s->h.slow_waits.load(); out[3] = s->fresh_boards.load();
}

static const void *server_fn(bool pairs)
{
    return pairs ? reinterpret_cast<const void *>(&nq_server_kernel<true>) : reinterpret_cast<const void *>(&nq_server_kernel<false>);
}

static size_t server_lds(int n, bool pairs)
{
    const size_t tab_words = pairs ? ((((size_t)(6 * n + 4) >> 1) + 3) & ~(size_t)3) : ((((size_t)(5 * n + 1) / 2) + 3) & ~(size_t)3);
    return (tab_words + (((size_t)n + 3) & ~(size_t)3) + 4 * (size_t)NQ_STEP_MAX_SLOTS) * 4;
}

static int server_set_lds_attr(ls_ctx *ctx)
{
    static std::atomic<bool> attr_done_dev[64];
    if (attr_done_dev[ctx->device & 63].load(std::memory_order_acquire)) return LS_OK;
    const int max_lds = (((6 * NQ_STEP_MAX_N + 4) / 2 + 4) + NQ_STEP_MAX_N + 4 * NQ_STEP_MAX_SLOTS) * 4;
    hipError_t e;
    if ((e = hipFuncSetAttribute(server_fn(false), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess ||
        (e = hipFuncSetAttribute(server_fn(true), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)) != hipSuccess)
        return set_err(ctx, LS_ERR_HIP, "LS-step server: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    attr_done_dev[ctx->device & 63].store(true, std::memory_order_release);
    return LS_OK;
}

// How many slots one resident launch can hold on this device for boards of size n: every workgroup (1024 threads, the
// board's tables in LDS) must be resident for as long as the launch lives -- a workgroup that is never scheduled never
// answers, and the served slots keep the launch from idling out -- so the count is bounded by what the CUs hold at once,
// less some headroom for whatever else must run (another context's server, the scheduling grid, copy kernels).
int nq_server_max_slots(ls_ctx *ctx, int n)
{
    if (server_set_lds_attr(ctx) != LS_OK) return 0;
    const bool pairs = (n & 1) == 0 && n >= 128;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, server_fn(pairs), 1024, server_lds(n, pairs)) != hipSuccess) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    const long long cap = (long long)(per_cu < 1 ? 1 : per_cu) * ctx->n_cus - (ctx->n_cus >= 64 ? 16 : 0);
    return (int)(cap < 1 ? 1 : (cap > 1024 ? 1024 : cap));
}

int nq_server_create(ls_ctx *ctx, int n, int n_slots, NqServer **out)
{
    *out = nullptr;
    if (n < 2 || n > NQ_STEP_MAX_N || n_slots < 1)
        return set_err(ctx, LS_ERR_UNSUPPORTED, "LS-step server: n = %d, %d slots outside what one resident launch holds", n, n_slots);
    {
        const int cap = nq_server_max_slots(ctx, n);
        if (n_slots > cap)
            return set_err(ctx, LS_ERR_UNSUPPORTED, "LS-step server: %d slots at n = %d, the device holds at most %d resident workgroups of it", n_slots, n, cap);
    }
    NqServer *s = new (std::nothrow) NqServer();
    if (!s) return set_err(ctx, LS_ERR_NOMEM, "out of host memory");
    s->n = n;
    s->n_pad = (n + 63) & ~63;
    s->n_slots = n_slots;
    s->pairs = (n & 1) == 0 && n >= 128;
    s->lds = server_lds(n, s->pairs);
    size_t off = 0;
    s->o_cols = off; off += (size_t)n_slots * NQ_STEP_MAX_SLOTS * 4;
    s->o_shadow = off; off += (size_t)n_slots * s->n_pad * 4;
    s->o_cs = off; off += (size_t)n_slots * s->n_pad * 2;
    int rc = s->h.init(ctx, "LS-step server", n_slots, off);
    if (rc != LS_OK) { delete s; return rc; }
    s->slots.resize((size_t)n_slots);
    for (auto &sl : s->slots) sl.cs.resize((size_t)n);
    s->h.launch = [s](const SrvLink &L) {
        NqSrvArgs a;
        a.link = L;
        a.cols_ext = s->cols_ext(0); a.shadow = s->shadow(0); a.cs16 = reinterpret_cast<unsigned *>(s->cs16(0));
        a.n = s->n; a.n_pad = s->n_pad;
        void *args[] = {&a};
        return hipLaunchKernel(server_fn(s->pairs), dim3((unsigned)s->n_slots), dim3(1024), args, s->lds, s->h.stream);
    };
    *out = s;
    return LS_OK;
}

void nq_server_destroy(NqServer *s)
{
    if (!s) return;
    if (getenv("LS_AMD_TRACE")) {
        double pre = 0, wait = 0, post = 0;
        unsigned long long calls = 0, ticks[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (const auto &sl : s->slots) {
            pre += sl.t_pre; wait += sl.t_wait; post += sl.t_post; calls += sl.calls;
            for (int d = 0; d < 8; d++) ticks[d] += sl.dev_ticks[d];
        }
        if (calls) {
            fprintf(stderr, "ls step server (n = %d, %d slots): %llu steps, per step: host before %.2f us, answer wait %.2f us, host after %.2f us; %llu launches, "
                            "%llu boards uploaded\n", s->n, s->n_slots, calls, pre / calls * 1e6, wait / calls * 1e6, post / calls * 1e6, s->h.relaunches.load(), s->fresh_boards.load());
            if (ticks[4])
                fprintf(stderr, "  workgroup, after the request was seen: request published to the block %.2f us, slot records %.2f, candidates scored %.2f, "
                                "block minimum %.2f, answer written %.2f; %.0f shader cycles (%.2f GHz)\n", ticks[0] * 0.01 / calls, ticks[1] * 0.01 / calls, ticks[2] * 0.01 / calls,
                        ticks[3] * 0.01 / calls, ticks[4] * 0.01 / calls, ticks[5] * 4.0 / calls, ticks[5] * 4.0 / (ticks[4] * 10.0));
            if (ticks[6])
                fprintf(stderr, "  the poll before a request: %.2f polls over an idle gap of %.2f us = one poll per %.2f us\n", (double)ticks[6] / calls, ticks[7] * 0.01 / calls,
                        ticks[7] * 0.01 / (double)ticks[6]);
        }
    }
    s->h.shutdown();
    delete s;
}

// The board of this request versus what the slot holds: an unchanged board costs a 4n-byte compare; a new one is range-
// checked, copied into the pinned shadow and gets a new id (the workgroup then reads it from there).  Returns the index of
// an entry outside [0, n), or -1.
static long long sync_board(NqServer *s, int slot, const int32_t *rows)
{
    NqServer::Slot &sl = s->slots[(size_t)slot];
    int *sh = s->shadow(slot);
    const int n = s->n;
    if (!sl.valid || memcmp(rows, sh, (size_t)n * 4) != 0) {
        uint32_t worst = 0; // (uint32_t)x < n  <=>  0 <= x < n
        for (int i = 0; i < n; i++) worst = (uint32_t)rows[i] > worst ? (uint32_t)rows[i] : worst;
        if (worst >= (uint32_t)n)
            for (int i = 0; i < n; i++)
                if ((uint32_t)rows[i] >= (uint32_t)n) return i;
        memcpy(sh, rows, (size_t)n * 4);
        sl.valid = true;
        sl.cs_valid = false;
        sl.board_id = sl.board_id == 0xffffffffu ? 1u : sl.board_id + 1u;
        s->fresh_boards.fetch_add(1, std::memory_order_relaxed);
    }
    return -1;
}

void nq_cs_after_move(const int32_t *rows, int32_t *cs, int n, int mc, int mv, int mq); // ls_api.hip
void widen_i32(const int32_t *src, int64_t *dst, int n);                                // ls_api.hip
void nq_cs_after_move_widen(const int32_t *rows, int32_t *cs, int64_t *out, int n, int mc, int mv, int mq); // ls_api.hip

// One LS iteration of one restart on slot `slot` (what ls_nqueens_eval_batch[_next] computes for n_restarts == 1).
// rows / cols are validated by the caller; window is already clamped.
// One LS step of a slot in two halves, so that a batched call can have the requests of all its restarts in flight at once:
// nq_server_post writes the request, nq_server_finish waits for the answer and decodes it.  nq_server_step = both.
int nq_server_post(ls_ctx *ctx, NqServer *s, int slot, const int32_t *rows, const int32_t *cols, int32_t n_cols, long long window,
                   bool has_next_col_scores)
{
    if (slot < 0 || slot >= s->n_slots) return set_err(ctx, LS_ERR_INVALID, "LS-step server: slot %d outside [0, %d)", slot, s->n_slots);
    const int n = s->n;
    NqServer::Slot &sl = s->slots[(size_t)slot];
    static const bool trace = getenv("LS_AMD_TRACE") != nullptr;
    const auto T0 = trace ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    {
        const long long bad = sync_board(s, slot, rows);
        if (bad >= 0) return set_err(ctx, LS_ERR_INVALID, "rows[%lld] = %d outside [0, %d)", bad, rows[bad], n);
    }
    ctx->steps_done++;
    if (!has_next_col_scores) sl.cs_valid = false; // nobody keeps the scores across this call
    const bool want_cs = has_next_col_scores && !sl.cs_valid;
    sl.want_cs = want_cs;
    // the window arithmetic of the step (nq_k_limit, nq_step_kernel) is done here: 64-bit divisions are slow on the device
    const long long full = window / (n - 1);
    const int n_whole = (int)(full < n_cols ? full : n_cols);
    const int rem = n_whole < n_cols ? (int)(window - (long long)n_whole * (n - 1)) : 0;
    const int reach = n_whole + (rem > 0 ? 1 : 0); // slots take(window) can reach (<= NQ_STEP_MAX_SLOTS: nq_step_eligible)
    const int steps = (n + 127) >> 7;
    int per_part = n_whole > 0 ? (int)(((long long)n_whole * steps + 15) / 16) : steps; // the sixteen wavefronts' fair share of the paired walk,
    per_part = (per_part + 3) & ~3;                                                      // rounded up to its unroll factor of four
    per_part = per_part > steps ? steps : (per_part < 4 ? 4 : per_part);
    const int parts = (steps + per_part - 1) / per_part;
    unsigned pay[16] = {0};
    pay[0] = SRV_F_STEP | (want_cs ? SRV_F_WANT_CS : 0u);
    pay[1] = (unsigned)n_whole;
    pay[2] = (unsigned)rem;
    pay[3] = (unsigned)parts | ((unsigned)per_part << 8);
    pay[4] = sl.board_id & 0xffffu; pay[5] = sl.board_id >> 16;
    if (reach > SRV_INLINE_COLS) {
        pay[0] |= SRV_F_EXT_COLS;
        memcpy(s->cols_ext(slot), cols, (size_t)reach * 4);
    } else {
        for (int i = 0; i < (int)reach; i++) pay[6 + i] = (unsigned)cols[i];
    }
    int rc = s->h.ensure_running(ctx);
    if (rc != LS_OK) return rc;
    s->h.post(slot, pay);
    if (trace) sl.t_pre += std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
    return LS_OK;
}

int nq_server_finish(ls_ctx *ctx, NqServer *s, int slot, int64_t cur_score, const int32_t *cols, ls_nq_result *res, int64_t *next_col_scores)
{
    const int n = s->n;
    NqServer::Slot &sl = s->slots[(size_t)slot];
    static const bool trace = getenv("LS_AMD_TRACE") != nullptr;
    const bool want_cs = sl.want_cs;
    const auto T0 = trace ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    int rc;
    const auto T1 = trace ? std::chrono::steady_clock::now() : T0;
    unsigned r[8];
    rc = s->h.wait(ctx, slot, r);
    if (rc != LS_OK) { sl.valid = false; return rc; }
    const auto T2 = trace ? std::chrono::steady_clock::now() : T0;
    const bool moved = (r[0] & 1u) != 0;
    if (!moved) {
        res->n_scored = 0; res->best_index = -1; res->best_score = 0; res->best_col = -1; res->best_value = -1;
    } else {
        const int bslot = (int)r[3], v = (int)r[4];
        res->n_scored = (long long)(r[1] | (r[2] << 16));
        res->best_index = (long long)bslot * n + v;
        res->best_score = cur_score + 2ll * ((int)(r[5] | (r[6] << 16)) - NQ_DELTA_BIAS);
        res->best_col = cols[bslot];
        res->best_value = v;
    }
    int *sh = s->shadow(slot);
    if (next_col_scores) {
        if (want_cs) { // from the kernel; remembered for the next call
            const unsigned short *k16 = s->cs16(slot);
            for (int j = 0; j < n; j++) { sl.cs[(size_t)j] = k16[j]; next_col_scores[j] = k16[j]; }
            sl.cs_valid = true;
        } else {
            if (moved) nq_cs_after_move_widen(sh, sl.cs.data(), next_col_scores, n, res->best_col, res->best_value, sh[res->best_col]);
            else widen_i32(sl.cs.data(), next_col_scores, n);
        }
    }
    else if (moved) sl.cs_valid = false; // nobody asked for the scores of the board this call leaves behind: the cached ones are the previous board's
    if (moved) sh[res->best_col] = res->best_value; // the workgroup applied it to its LDS board
    if (trace) {
        const auto T3 = std::chrono::steady_clock::now();
        auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        sl.t_wait += sec(T1, T2); sl.t_post += sec(T2, T3);
        sl.calls++;
        const volatile unsigned *rl = s->h.resp(slot);
        for (int d = 0; d < 8; d++) sl.dev_ticks[d] += (rl[7 + d] >> 16) == s->h.per[(size_t)slot].tag ? rl[7 + d] & 0xffffu : 0u;
    }
    return LS_OK;
}

int nq_server_step(ls_ctx *ctx, NqServer *s, int slot, const int32_t *rows, int64_t cur_score, const int32_t *cols, int32_t n_cols,
                   long long window, ls_nq_result *res, int64_t *next_col_scores)
{
    const int rc = nq_server_post(ctx, s, slot, rows, cols, n_cols, window, next_col_scores != nullptr);
    if (rc != LS_OK) return rc;
    return nq_server_finish(ctx, s, slot, cur_score, cols, res, next_col_scores);
}

// get_col_scores + score of one board (examples/nqueens/src/lib.rs:74-87, 126-140) on the slot; the board stays with the
// slot, so the LS walk that starts from it (local_search.rs:306-309) uploads nothing.
int nq_server_col_scores(ls_ctx *ctx, NqServer *s, int slot, const int32_t *rows, int64_t *col_scores, int64_t *total)
{
    if (slot < 0 || slot >= s->n_slots) return set_err(ctx, LS_ERR_INVALID, "LS-step server: slot %d outside [0, %d)", slot, s->n_slots);
    const int n = s->n;
    NqServer::Slot &sl = s->slots[(size_t)slot];
    {
        const long long bad = sync_board(s, slot, rows);
        if (bad >= 0) return set_err(ctx, LS_ERR_INVALID, "rows[%lld] = %d outside [0, %d)", bad, rows[bad], n);
    }
    unsigned pay[16] = {0};
    pay[0] = SRV_F_WANT_CS;
    pay[4] = sl.board_id & 0xffffu; pay[5] = sl.board_id >> 16;
    int rc = s->h.ensure_running(ctx);
    if (rc != LS_OK) return rc;
    s->h.post(slot, pay);
    unsigned r[8];
    rc = s->h.wait(ctx, slot, r);
    if (rc != LS_OK) { sl.valid = false; return rc; }
    const unsigned short *k16 = s->cs16(slot);
    for (int j = 0; j < n; j++) sl.cs[(size_t)j] = k16[j];
    sl.cs_valid = true;
    if (col_scores) for (int j = 0; j < n; j++) col_scores[j] = k16[j];
    if (total) *total = (int64_t)(r[5] | (r[6] << 16));
    return LS_OK;
}

} // namespace lsamd
