// sched_kernels.hip — gfx950 kernels for the employee-scheduling neighbourhood
// pipeline: local-search/src/local_search.rs:315-325 with the plug-in of
// examples/employee-scheduling/src/lib.rs (scorer :261-375, helpers :149-222,
// moves :464-479 / :526-550).
//
// Two families.  The DEFAULT is the delta scorer further down (sched_eval_small_kernel, sched_eval_all_kernel,
// sched_finish_kernel, sched_eval_delta_kernel for score dumps: one LANE per candidate, terms as functions of per-employee day sets).  The first half of the
// file is the wavefront-per-candidate family (LS_SCHED_DELTA=0; ls_sched_score and the count / locate passes of score
// dumps always use it):
//
// One wavefront scores one candidate at a time; lane d holds day d (<= 64 stored days).
// Every constraint term is a function of the day-equality matrix
//   eq[d] = bitmask over scored days q with employee(q) == employee(d)
//   H1 holidays            bit d of holiday_bits[employee(d)]               lib.rs:273-280
//   H2 consecutive days    bit d+1 of eq[d]                                 lib.rs:286-292
//   H3 consecutive w/ends  popcount(eq[d] & h3mask[d])  (static per-lane)   lib.rs:295-315
//   H4 >3 in 14 days       closed form over the first occurrence p of an
//                          employee in a window: windows w with
//                          max(prev(p)+1, p-13, 0, q4(p)-13) <= w <= min(p, D-14)
//                          where q4 = position of the 4th occurrence from p  lib.rs:318-327
//   S1 >2 in 7 days        same with 7 / 3rd occurrence                     lib.rs:330-339
//   S2 weekday consistency min over a weekday's lanes of popcount(eq & same-weekday),
//                          counted when the weekday has >= 2 employees      lib.rs:194-218
//   S3 / S4                max - min of popcount(eq) / popcount(eq & weekend) over the
//                          scored lanes, from ONE OR-reduction of one-hot counts   lib.rs:345-365
// All terms are small integers (the reference holds them in f64; exact).
//
// A candidate differs from the restart's base vector in at most two days, so its
// equality matrix is a delta update of the base matrix (built once per wavefront):
// clear the edited days' bits, set them where the lane's employee equals the new one,
// and the edited lanes take the ballots of those two comparisons.
//
// A wavefront handles candidates in batches of 64: each lane loads one move (coalesced),
// derives its edit (positions, new employees, tabu flag), gathers the two holiday bits it
// will need and its half of the argmin key; then the wavefront walks the non-tabu lanes of
// the batch, broadcasting one edit at a time.  Memory latency is paid once per 64 candidates.
//
// Argmin by (hard, soft, date_to_employee vector), first of equals (local_search.rs:29-37).
// The vector order of <= 2-edit candidates of one base is a closed form over the edits:
// per edit (class: lowers < none < raises, position key, new value).  Packed with the scores
// into one 64-bit key per candidate; the last kernel takes the minimum (key, index).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "ls_common.hpp"
#include "sched_common.hpp"
#include "srv_common.hpp"

namespace lsamd {

typedef unsigned long long u64;

#ifdef LS_NQ_STAMPS
// Diagnostic build only (tools/table_stamps.py build; tools/sched_stamps.py): wall-clock stamps (100 MHz) of the delta
// kernels' phases, written to a buffer no other code reads.  The product build has none of this.
__device__ unsigned long long g_sc_stamps[1024 * 16];
#define LS_SC_STAMP(P) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_sc_stamps[blockIdx.x * 16 + (P)] = wall_clock64(); } while (0)
#else
#define LS_SC_STAMP(P) do { } while (0)
#endif

constexpr u64 SC_KEY_EMPTY = ~0ull;
constexpr int SC_BIG = 1 << 20;
constexpr int SC_COUNT_CHUNK = 1024; // moves per block in the window-count pass
constexpr int SC_FUSED_WAVES = 16;   // wavefronts of the fused small-launch block (one block per restart)

struct SchedInst {
    int D;             // scored days
    int S;             // stored days
    int n_employees;
    const u64 *holiday_bits;
    const SchedLaneConst *lane_tab;
};

__device__ __forceinline__ unsigned wave_or_u32(unsigned x)
{
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_QUAD_1032, 0xf, 0xf, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_QUAD_2301, 0xf, 0xf, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_ROW_HALF_MIRROR, 0xf, 0xf, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_ROW_MIRROR, 0xf, 0xf, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_ROW_BCAST15, 0xa, 0xf, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_ROW_BCAST31, 0xc, 0xf, false);
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

// day-equality matrix from scratch: one readlane + compare per scored day
__device__ __forceinline__ u64 sched_build_eq(int D, int e)
{
    u64 eq = 0;
#pragma unroll
    for (int q = 0; q < 64; q++) {
        if (q < D) {
            const int eq_q = __builtin_amdgcn_readlane(e, q);
            if (e == eq_q) eq |= 1ull << q;
        }
    }
    return eq;
}

struct SchedScore {
    int hard, soft;
    int t[8];
};

// All eight terms from the equality matrix row `eq` of this lane's day and its holiday bit.
template <bool WANT_TERMS>
__device__ __forceinline__ SchedScore sched_terms(int D, const SchedLaneConst &L, int p, u64 eq, int holiday_bit)
{
    const bool scored = L.flags & 1;
    int h1 = 0, h2 = 0, h3 = 0, h4 = 0, s1 = 0, s2 = 0;
    u64 tot_onehot = 0;
    unsigned wk_onehot = 0;
    if (scored) {
        h1 = holiday_bit;
        h2 = (eq & L.mask2) ? 1 : 0;
        h3 = __popcll(eq & L.h3mask);
        const u64 below = eq & ((1ull << p) - 1);
        const int prev = below ? 63 - __clzll((long long)below) : -1;
        u64 t = eq >> p; // bit 0 = day p itself
        t &= t - 1;
        t &= t - 1;
        const int q3 = t ? p + __ffsll((long long)t) - 1 : SC_BIG;
        t &= t - 1;
        const int q4 = t ? p + __ffsll((long long)t) - 1 : SC_BIG;
        {
            int lo = prev + 1;
            lo = lo > p - 6 ? lo : p - 6;
            lo = lo > 0 ? lo : 0;
            lo = lo > q3 - 6 ? lo : q3 - 6;
            const int hi = p < D - 7 ? p : D - 7;
            s1 = hi - lo + 1 > 0 ? hi - lo + 1 : 0;
        }
        {
            int lo = prev + 1;
            lo = lo > p - 13 ? lo : p - 13;
            lo = lo > 0 ? lo : 0;
            lo = lo > q4 - 13 ? lo : q4 - 13;
            const int hi = p < D - 14 ? p : D - 14;
            h4 = hi - lo + 1 > 0 ? hi - lo + 1 : 0;
        }
        tot_onehot = 1ull << (__popcll(eq) - 1);                   // 1 <= days of this employee <= 64
        wk_onehot = 1u << __popcll(eq & L.weekend_mask);           // 0 <= weekend days <= 20
    }
    // S2: min over the lanes of one weekday (stride 7)
    int cnt = L.wdmask ? __popcll(eq & L.wdmask) : SC_BIG;
    {
        int o;
        o = __shfl_down(cnt, 7, 64);  cnt = o < cnt ? o : cnt;
        o = __shfl_down(cnt, 14, 64); cnt = o < cnt ? o : cnt;
        o = __shfl_down(cnt, 28, 64); cnt = o < cnt ? o : cnt;
        o = __shfl_down(cnt, 56, 64); cnt = o < cnt ? o : cnt;
    }
    if ((L.flags & 2) && cnt < L.gsize) s2 = cnt;

    SchedScore r;
    // hard <= 316, soft (sum part) <= 141: 16-bit fields
    const unsigned hs = (unsigned)wave_sum_i32((h1 + h2 + h3 + h4) | ((s1 + s2) << 16));
    // max - min of the per-employee counts = distance between the highest and lowest set bit of the
    // OR over all scored lanes of the one-hot counts
    const unsigned tot_lo = wave_or_u32((unsigned)tot_onehot), tot_hi = wave_or_u32((unsigned)(tot_onehot >> 32));
    const unsigned wk_or = wave_or_u32(wk_onehot);
    const u64 tot_or = ((u64)tot_hi << 32) | tot_lo;
    const int s3 = (63 - __clzll((long long)tot_or)) - (__ffsll((long long)tot_or) - 1);
    const int s4 = (31 - __clz((int)wk_or)) - (__ffs((int)wk_or) - 1);
    r.hard = (int)(hs & 0xffffu);
    r.soft = (int)(hs >> 16) + s3 + s4;
    if (WANT_TERMS) {
        r.t[0] = wave_sum_i32(h1); r.t[1] = wave_sum_i32(h2); r.t[2] = wave_sum_i32(h3);
        r.t[3] = wave_sum_i32(h4); r.t[4] = wave_sum_i32(s1); r.t[5] = wave_sum_i32(s2);
        r.t[6] = s3; r.t[7] = s4;
    }
    return r;
}

__device__ __forceinline__ int sched_holiday_bit(const SchedInst &in, int employee, int day)
{
    if ((unsigned)employee >= (unsigned)in.n_employees || (unsigned)day >= (unsigned)in.D) return 0;
    return (int)((in.holiday_bits[employee] >> day) & 1ull);
}

__global__ __launch_bounds__(256) void sched_score_kernel(SchedInst in, int n_states, const int *states,
                                                          long long *hard, long long *soft, long long *terms)
{
    const int lane = threadIdx.x & 63;
    const int gw = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
    const int n_waves = (gridDim.x * 256) >> 6;
    const SchedLaneConst L = in.lane_tab[lane];
    for (int s = gw; s < n_states; s += n_waves) {
        const int e = lane < in.S ? states[(long long)s * in.S + lane] : -1 - lane;
        const u64 eq = sched_build_eq(in.D, e) & L.scored_mask;
        const SchedScore r = sched_terms<true>(in.D, L, lane, eq, sched_holiday_bit(in, e, lane));
        if (lane == 0) {
            hard[s] = r.hard;
            soft[s] = r.soft;
            if (terms)
                for (int i = 0; i < 8; i++) terms[(long long)s * 8 + i] = r.t[i];
        }
    }
}

// --------------------------------------------------------------- neighbourhood
struct SchedEvalArgs {
    int emp_entries = 0;   // entries of a block's employee -> scored-days table in dynamic LDS (sched_eval_all_kernel; 0: none)
    SchedInst in;
    int n_restarts;
    const int *states;          // [n_restarts][S]
    const ls_sched_move *moves; // [n_restarts][max_moves]
    const long long *n_moves;   // [n_restarts]
    long long max_moves;
    long long window;
    int n_chunks;        // count-pass blocks per restart (delta pipeline: chunks of chunk_moves moves)
    long long chunk_moves; // delta pipeline: moves per block of sched_eval_all_kernel
    int *chunk_counts;   // [n_restarts][n_chunks] non-tabu moves per chunk
    long long *k_end;    // [n_restarts]  first index not scored
    long long *n_scored; // [n_restarts]
    int blocks_per_restart;
    int split;           // wavefronts sharing one 64-candidate batch (1 or 4): small launches are latency-bound
    u64 *part_key;       // [n_restarts][blocks_per_restart] per-block minimum key
    long long *part_idx; // [n_restarts][blocks_per_restart] and its move index
    long long *hard_out, *soft_out; // nullable [n_restarts][max_moves]
    ls_sched_result *results;       // written by pass 2 itself when blocks_per_restart == 1
    unsigned long long *done_flag;  // nullable; one-restart launches only: host word <- done_seq after the result record
    unsigned long long done_seq;
    unsigned *ticket;               // delta pipeline, one launch: [n_restarts] blocks done so far (zero between launches)
    u64 *cand_keys;                 // delta pipeline: [n_restarts][max_moves] argmin key per candidate (SC_KEY_EMPTY: not a candidate)
    unsigned *resp_line;            // nullable; resident grid: the result goes out as a tagged answer line (srv_common.hpp) instead
    unsigned resp_tag;
    // one-restart launches may carry their per-call inputs in the kernel arguments (no copy, no PCIe read):
    int inline_valid;
    long long inline_n_moves;
    int inline_state[64];
};

__device__ __forceinline__ int sched_state_at(const SchedEvalArgs &a, int r, int d)
{
    return a.inline_valid ? a.inline_state[d & 63] : a.states[(long long)r * a.in.S + d];
}

__device__ __forceinline__ bool sched_move_ok(int type, int a, int b, int S, int E)
{
    if (type == 0) return (unsigned)a < (unsigned)S && (unsigned)b < (unsigned)E;
    if (type == 1) return (unsigned)a < (unsigned)S && (unsigned)b < (unsigned)S;
    return false;
}

__device__ __forceinline__ long long sched_nm(const SchedEvalArgs &a, int r)
{
    const long long nm = a.inline_valid ? a.inline_n_moves : a.n_moves[r];
    return nm < 0 ? 0 : (nm > a.max_moves ? a.max_moves : nm);
}

// is move k a real (non-tabu, well-formed) candidate of the state held in s_state?
__device__ __forceinline__ bool sched_counts(const SchedEvalArgs &a, const ls_sched_move *mv, long long k, long long nm,
                                             const int *s_state)
{
    if (k >= nm) return false;
    const ls_sched_move m = mv[k];
    if (!sched_move_ok(m.type, m.a, m.b, a.in.S, a.in.n_employees)) return false;
    return m.type == 0 ? (s_state[m.a] != m.b) : (s_state[m.a] != s_state[m.b]);
}

// Pass 1a: non-tabu moves per chunk of SC_COUNT_CHUNK (tabu candidates do not consume window,
// local_search.rs:319-321).  grid = (n_chunks, n_restarts).
__global__ __launch_bounds__(256) void sched_count_kernel(SchedEvalArgs a)
{
    __shared__ int s_state[64];
    __shared__ int s_cnt[4];
    const int r = blockIdx.y;
    if (threadIdx.x < 64) s_state[threadIdx.x] = (int)threadIdx.x < a.in.S ? sched_state_at(a, r, (int)threadIdx.x) : -1;
    __syncthreads();
    const long long nm = sched_nm(a, r);
    const ls_sched_move *mv = a.moves + (long long)r * a.max_moves;
    const long long k0 = (long long)blockIdx.x * SC_COUNT_CHUNK;
    int c = 0;
#pragma unroll
    for (int i = 0; i < SC_COUNT_CHUNK / 256; i++) c += sched_counts(a, mv, k0 + i * 256 + threadIdx.x, nm, s_state) ? 1 : 0;
    c = wave_sum_i32(c);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) a.chunk_counts[(long long)r * a.n_chunks + blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// Pass 1b: where does take(window) stop?  One block per restart: a block-wide prefix scan over
// the chunk counts finds the chunk in which the running count reaches the window, then the exact
// position inside that chunk is resolved with ballots.
__global__ __launch_bounds__(256) void sched_locate_kernel(SchedEvalArgs a)
{
    __shared__ int s_state[64];
    __shared__ int s_wcnt[4];
    __shared__ int s_chunk;
    __shared__ long long s_run, s_found;
    const int r = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < 64) s_state[threadIdx.x] = (int)threadIdx.x < a.in.S ? sched_state_at(a, r, (int)threadIdx.x) : -1;
    if (threadIdx.x == 0) { s_chunk = -1; s_run = 0; s_found = -1; }
    __syncthreads();
    const long long nm = sched_nm(a, r);
    const ls_sched_move *mv = a.moves + (long long)r * a.max_moves;
    const int *counts = a.chunk_counts + (long long)r * a.n_chunks;
    long long run = 0, k_end = nm;
    if (a.window <= 0) {
        k_end = 0;
    } else {
        if (a.n_chunks == 1 && threadIdx.x == 0) { s_chunk = 0; s_run = 0; } // no count pass was run
        __syncthreads();
        for (int t0 = 0; t0 < a.n_chunks && a.n_chunks > 1; t0 += 256) {
            const int b = t0 + (int)threadIdx.x;
            const int c = b < a.n_chunks ? counts[b] : 0;
            int incl = c; // inclusive scan: wavefront, then across the four wavefronts
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            if (lane == 63) s_wcnt[w] = incl;
            __syncthreads();
            int before = 0, total = 0;
            for (int i = 0; i < 4; i++) {
                if (i < w) before += s_wcnt[i];
                total += s_wcnt[i];
            }
            incl += before;
            // exactly one chunk has (count before it) < window <= (count through it)
            if (b < a.n_chunks && run + incl - c < a.window && run + incl >= a.window) {
                s_chunk = b;
                s_run = run + incl - c;
            }
            __syncthreads();
            if (s_chunk >= 0) break;
            run += total;
        }
        const int chunk = s_chunk;
        if (chunk >= 0) {
            run = s_run;
            const long long base = (long long)chunk * SC_COUNT_CHUNK;
            for (int t = 0; t < SC_COUNT_CHUNK; t += 256) {
                const long long k = base + t + threadIdx.x;
                const bool flag = sched_counts(a, mv, k, nm, s_state);
                const u64 bal = __ballot(flag);
                __syncthreads(); // s_wcnt is reused
                if (lane == 0) s_wcnt[w] = __popcll(bal);
                __syncthreads();
                int before = 0, total = 0;
                for (int i = 0; i < 4; i++) {
                    if (i < w) before += s_wcnt[i];
                    total += s_wcnt[i];
                }
                const int incl = before + __popcll(bal & ((2ull << lane) - 1));
                if (flag && run + incl == a.window) s_found = k + 1;
                __syncthreads();
                if (s_found >= 0) break;
                run += total;
            }
            if (s_found >= 0) { k_end = s_found; run = a.window; }
        }
    }
    if (threadIdx.x == 0) {
        a.k_end[r] = k_end;
        a.n_scored[r] = a.window <= 0 ? 0 : run;
    }
}

__device__ __forceinline__ u64 sc_edit_key(int pos, int newv, int basev)
{
    const int up = newv > basev;
    const unsigned cls = up ? 2u : 0u;
    const unsigned poskey = up ? (unsigned)(63 - pos) : (unsigned)pos;
    return ((u64)cls << 21) | ((u64)poskey << 15) | (unsigned)newv;
}

__device__ __forceinline__ void sched_write_result_n(const SchedEvalArgs &a, int r, u64 bk, long long bi, long long n_scored)
{
    ls_sched_result out;
    out.n_scored = n_scored;
    if (bk == SC_KEY_EMPTY) { out.best_index = -1; out.best_hard = 0; out.best_soft = 0; out.n_scored = 0; }
    else {
        out.best_index = bi;
        out.best_hard = (long long)(bk >> 54);
        out.best_soft = (long long)((bk >> 46) & 0xffu);
    }
    if (a.resp_line) {
        // resident grid: d0 has-move, d1-2 n_scored, d3-4 best_index, d5 hard, d6 soft -- every dword tagged, one store each
        const unsigned has = bk != SC_KEY_EMPTY ? 1u : 0u;
        const unsigned pay[8] = {has, (unsigned)out.n_scored & 0xffffu, (unsigned)(out.n_scored >> 16) & 0xffffu,
                                 has ? (unsigned)out.best_index & 0xffffu : 0u, has ? (unsigned)(out.best_index >> 16) & 0xffffu : 0u,
                                 (unsigned)out.best_hard, (unsigned)out.best_soft, 0u};
        for (int d = 0; d < 8; d++) __hip_atomic_store(a.resp_line + d, (a.resp_tag << 16) | (pay[d] & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (a.done_flag) {
        // one-restart launch from a resident entry: the record is the kernel's only output and lives in pinned host
        // memory; the caller polls the completion word instead of synchronising the stream.  Write-through
        // (system-scope) stores, their acknowledgement, then the word: no system-scope release (an L2 write-back).
        long long *dst = reinterpret_cast<long long *>(a.results + r);
        __hip_atomic_store(dst + 0, out.n_scored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + 1, out.best_index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + 2, out.best_hard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + 3, out.best_soft, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(a.done_flag, a.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        a.results[r] = out;
    }
}

__device__ __forceinline__ void sched_write_result(const SchedEvalArgs &a, int r, u64 bk, long long bi)
{
    sched_write_result_n(a, r, bk, bi, a.n_scored[r]);
}

// Fused small launch (one block of NW wavefronts per restart, max_moves <= SC_COUNT_CHUNK): the block first finds
// where take(window) stops (pass 1b's in-chunk ballots, on the whole move list), then scores, then writes the
// result itself -- locate + eval + reduce in ONE launch.  The state is read once into LDS, so it may live in pinned
// host memory.
template <bool FUSED> struct SchedFusedShared {
    int state[64];
    int wcnt[16];
    long long found;
    ls_sched_move moves[FUSED ? 64 * SC_FUSED_WAVES : 1]; // the (<= 512-entry) move list, read from memory once
};

template <int NW>
__device__ __forceinline__ void sched_locate_in_block(const SchedEvalArgs &a, int r, SchedFusedShared<true> &sh, long long &k_end,
                                                      long long &n_scored)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // every global read of this phase is issued before the first wait: the move (device memory, speculative: it does
    // not depend on the state or on n_moves), the state and n_moves (possibly pinned host memory, one PCIe round trip)
    const ls_sched_move *mv = a.moves + (long long)r * a.max_moves;
    ls_sched_move m0;
    m0.type = -1; m0.a = 0; m0.b = 0;
    if ((long long)threadIdx.x < a.max_moves) m0 = mv[threadIdx.x];
    const long long nm = sched_nm(a, r);
    if (threadIdx.x < 64) sh.state[threadIdx.x] = (int)threadIdx.x < a.in.S ? sched_state_at(a, r, (int)threadIdx.x) : -1;
    if (threadIdx.x == 0) sh.found = -1;
    sh.moves[threadIdx.x] = m0;
    __syncthreads();
    long long run = 0;
    k_end = nm;
    if (a.window <= 0) { k_end = 0; n_scored = 0; return; }
    {   // the fused launch serves max_moves <= 64 * NW: one pass over the list
        const long long k = threadIdx.x;
        const bool flag = k < nm && sched_move_ok(m0.type, m0.a, m0.b, a.in.S, a.in.n_employees) &&
                          (m0.type == 0 ? (sh.state[m0.a] != m0.b) : (sh.state[m0.a] != sh.state[m0.b]));
        const u64 bal = __ballot(flag);
        if (lane == 0) sh.wcnt[w] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
        for (int i = 0; i < NW; i++) {
            if (i < w) before += sh.wcnt[i];
            total += sh.wcnt[i];
        }
        const int incl = before + __popcll(bal & ((2ull << lane) - 1));
        if (flag && run + incl == a.window) sh.found = k + 1;
        __syncthreads();
        if (sh.found < 0) run += total;
    }
    if (sh.found >= 0) { k_end = sh.found; run = a.window; }
    n_scored = run;
}

// Pass 2: score every non-tabu candidate k < k_end.
template <int NW, bool FUSED>
__global__ __launch_bounds__(64 * NW) void sched_eval_kernel(SchedEvalArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = FUSED ? (int)blockIdx.x : (int)(blockIdx.x % a.n_restarts);
    const int wave = FUSED ? wib : (int)(blockIdx.x / a.n_restarts) * NW + wib;
    const int n_waves = FUSED ? NW : (int)(gridDim.x / a.n_restarts) * NW;
    const int S = a.in.S, D = a.in.D, E = a.in.n_employees;
    __shared__ SchedFusedShared<FUSED> s_fused;
    long long k_end, n_scored_f = 0;
    if constexpr (FUSED) sched_locate_in_block<NW>(a, r, s_fused, k_end, n_scored_f);
    else k_end = a.k_end[r];
    const long long n_batches = ((k_end + 63) >> 6) * a.split; // work units: (batch, part)
    __shared__ u64 s_key[NW];
    __shared__ long long s_idx[NW];
    u64 best_key = SC_KEY_EMPTY; // this lane's running minimum (key, index)
    long long best_idx = -1;
    if (wave < n_batches) {
    const SchedLaneConst L = a.in.lane_tab[lane];
    const int base = lane < S ? (FUSED ? s_fused.state[lane] : sched_state_at(a, r, lane)) : -1 - lane;
    const u64 eq_base = sched_build_eq(D, base) & L.scored_mask;
    const int hb_base = sched_holiday_bit(a.in, base, lane);
    const ls_sched_move *mv = a.moves + (long long)r * a.max_moves;

    for (long long unit = wave; unit < n_batches; unit += n_waves) {
        const long long batch = unit / a.split;
        const int part = (int)(unit % a.split);
        const long long k = (batch << 6) + lane;
        // ---- per-lane: this lane's move, its edit, its holiday bits, its half of the key
        int type = -1, ma = 0, mb = 0;
        if (k < k_end) { const ls_sched_move m = FUSED ? s_fused.moves[FUSED ? k : 0] : mv[k]; type = m.type; ma = m.a; mb = m.b; }
        const bool ok = sched_move_ok(type, ma, mb, S, E);
        const int ia = ok ? ma : 0, ib = (ok && type == 1) ? mb : ia;
        const int ea = __shfl(base, ia, 64), eb = __shfl(base, ib, 64);
        // ChangeDay: day a <- employee b (lib.rs:466-470);  SwapDays: a <- e[b], b <- e[a] (:471-478)
        const int pos1 = ia, new1 = type == 0 ? mb : eb;
        const int pos2 = ib, new2 = type == 0 ? mb : ea;
        // candidate != current (tabu, local_search.rs:319); with split > 1 this wavefront takes every split-th lane
        const bool active = ok && !(new1 == ea && new2 == eb) && (lane & (a.split - 1)) == part;
        int hbits = 0;
        u64 lowkey = 0;
        if (active) {
            hbits = sched_holiday_bit(a.in, new1, pos1) | (sched_holiday_bit(a.in, new2, pos2) << 1);
            u64 e1, e2 = (u64)1 << 21; // "no second edit" sorts between lowering and raising edits
            if (type == 0) e1 = sc_edit_key(pos1, new1, ea);
            else {
                const bool a_first = pos1 < pos2;
                e1 = a_first ? sc_edit_key(pos1, new1, ea) : sc_edit_key(pos2, new2, eb);
                e2 = a_first ? sc_edit_key(pos2, new2, eb) : sc_edit_key(pos1, new1, ea);
            }
            lowkey = (e1 << 23) | e2;
        }
        const int packed = pos1 | (pos2 << 8) | (hbits << 16);
        int my_hs = 0;
        // ---- wave: one non-tabu candidate of the batch at a time
        u64 todo = __ballot(active);
        while (todo) {
            const int j = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int pk = __builtin_amdgcn_readlane(packed, j);
            const int n1 = __builtin_amdgcn_readlane(new1, j), n2 = __builtin_amdgcn_readlane(new2, j);
            const int p1 = pk & 0xff, p2 = (pk >> 8) & 0xff;
            int e = base;
            if (lane == p1) e = n1;
            if (lane == p2) e = n2;
            const bool c1 = e == n1, c2 = e == n2;
            const u64 m1 = __ballot(c1) & L.scored_mask, m2 = __ballot(c2) & L.scored_mask;
            const u64 bit1 = p1 < D ? 1ull << p1 : 0, bit2 = p2 < D ? 1ull << p2 : 0;
            u64 eq = eq_base & ~(bit1 | bit2);
            if (c1) eq |= bit1;
            if (c2) eq |= bit2;
            if (lane == p1) eq = m1;
            if (lane == p2) eq = m2;
            int hb = hb_base;
            if (lane == p1) hb = (pk >> 16) & 1;
            if (lane == p2) hb = (pk >> 17) & 1;
            const SchedScore sc = sched_terms<false>(D, L, lane, eq, hb);
            if (lane == j) my_hs = (sc.hard << 8) | sc.soft;
        }
        // ---- per-lane: assemble and store (coalesced)
        if (active) {
            const int hard = my_hs >> 8, soft = my_hs & 0xff;
            const u64 key = ((u64)(unsigned)hard << 54) | ((u64)(unsigned)soft << 46) | lowkey;
            if (key < best_key) { best_key = key; best_idx = k; } // ascending k per lane: first of equals kept
            if (a.hard_out) a.hard_out[(long long)r * a.max_moves + k] = hard;
            if (a.soft_out) a.soft_out[(long long)r * a.max_moves + k] = soft;
        }
    }
    }
    // minimum (key, index) of the wavefront, then of the block
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned klo = __shfl_xor((unsigned)best_key, m, 64), khi = __shfl_xor((unsigned)(best_key >> 32), m, 64);
        const unsigned ilo = __shfl_xor((unsigned)best_idx, m, 64), ihi = __shfl_xor((unsigned)((u64)best_idx >> 32), m, 64);
        const u64 ok = ((u64)khi << 32) | klo;
        const long long oi = (long long)(((u64)ihi << 32) | ilo);
        if (ok < best_key || (ok == best_key && ok != SC_KEY_EMPTY && oi < best_idx)) { best_key = ok; best_idx = oi; }
    }
    if (lane == 0) { s_key[wib] = best_key; s_idx[wib] = best_idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < NW; i++)
            if (s_key[i] < best_key || (s_key[i] == best_key && best_key != SC_KEY_EMPTY && s_idx[i] < best_idx)) {
                best_key = s_key[i]; best_idx = s_idx[i];
            }
        if (FUSED) {
            sched_write_result_n(a, r, best_key, best_idx, n_scored_f);
        } else if (a.blocks_per_restart == 1) {
            sched_write_result(a, r, best_key, best_idx);
        } else {
            const long long slot = (long long)r * a.blocks_per_restart + blockIdx.x / a.n_restarts;
            a.part_key[slot] = best_key;
            a.part_idx[slot] = best_idx;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Pass 2 for schedules of at most 32 stored days (both configurations the reference ships): a
// candidate occupies HALF a wavefront (lane & 31 = day), so two candidates are scored per pass and
// every mask is 32-bit.  A 64-bit ballot is naturally the pair (candidate A, candidate B).
// Reductions stay inside each half: four DPP row steps + row_bcast:15 leave the half totals in
// lanes 31 and 63.

__device__ __forceinline__ int half_sum_i32(int x) // lanes 16-31 / 48-63 end up holding their half's total
{
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_QUAD_1032, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_QUAD_2301, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_ROW_HALF_MIRROR, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_ROW_MIRROR, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_ROW_BCAST15, 0xa, 0xf, false);
    return x;
}

__device__ __forceinline__ unsigned half_or_u32(unsigned x)
{
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_QUAD_1032, 0xf, 0xf, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_QUAD_2301, 0xf, 0xf, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_ROW_HALF_MIRROR, 0xf, 0xf, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_ROW_MIRROR, 0xf, 0xf, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_ROW_BCAST15, 0xa, 0xf, false);
    return x;
}

struct SchedLane32 {
    unsigned scored_mask, weekend_mask, mask2, h3mask, wdmask;
    int gsize, flags;
};

// per-lane partial terms of one candidate row; the three values are reduced per half by the caller
struct SchedPartial32 {
    int hs;            // (h1+h2+h3+h4) | (s1+s2) << 16
    unsigned tot_onehot, wk_onehot;
};

__device__ __forceinline__ SchedPartial32 sched_terms32(int D, const SchedLane32 &L, int p, int lane, unsigned eq, int holiday_bit)
{
    SchedPartial32 r;
    int h = 0, sf = 0;
    r.tot_onehot = 0;
    r.wk_onehot = 0;
    if (L.flags & 1) {
        h = holiday_bit + ((eq & L.mask2) ? 1 : 0) + __popc(eq & L.h3mask);
        const unsigned below = eq & ((1u << p) - 1);
        const int prev = below ? 31 - __clz((int)below) : -1;
        unsigned t = eq >> p; // bit 0 = day p itself
        t &= t - 1;
        t &= t - 1;
        const int q3 = t ? p + __ffs((int)t) - 1 : SC_BIG;
        t &= t - 1;
        const int q4 = t ? p + __ffs((int)t) - 1 : SC_BIG;
        {
            int lo = prev + 1;
            lo = lo > p - 6 ? lo : p - 6;
            lo = lo > 0 ? lo : 0;
            lo = lo > q3 - 6 ? lo : q3 - 6;
            const int hi = p < D - 7 ? p : D - 7;
            sf = hi - lo + 1 > 0 ? hi - lo + 1 : 0;
        }
        {
            int lo = prev + 1;
            lo = lo > p - 13 ? lo : p - 13;
            lo = lo > 0 ? lo : 0;
            lo = lo > q4 - 13 ? lo : q4 - 13;
            const int hi = p < D - 14 ? p : D - 14;
            h += hi - lo + 1 > 0 ? hi - lo + 1 : 0;
        }
        r.tot_onehot = 1u << (__popc(eq) - 1);
        r.wk_onehot = 1u << __popc(eq & L.weekend_mask);
    }
    // S2: min over the lanes of one weekday (stride 7) without leaving the half
    int cnt = L.wdmask ? __popc(eq & L.wdmask) : SC_BIG;
#pragma unroll
    for (int off = 7; off <= 28; off <<= 1) {
        const int o = __shfl(cnt, p + off < 32 ? lane + off : lane, 64);
        cnt = o < cnt ? o : cnt;
    }
    if ((L.flags & 2) && cnt < L.gsize) sf += cnt;
    r.hs = h | (sf << 16);
    return r;
}

template <int NW, bool FUSED>
__global__ __launch_bounds__(64 * NW) void sched_eval32_kernel(SchedEvalArgs a)
{
    const int lane = threadIdx.x & 63;
    const int li = lane & 31;      // day
    const bool upper = lane >= 32; // which of the two candidates of a pass this lane works for
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = FUSED ? (int)blockIdx.x : (int)(blockIdx.x % a.n_restarts);
    const int wave = FUSED ? wib : (int)(blockIdx.x / a.n_restarts) * NW + wib;
    const int n_waves = FUSED ? NW : (int)(gridDim.x / a.n_restarts) * NW;
    const int S = a.in.S, D = a.in.D, E = a.in.n_employees;
    __shared__ SchedFusedShared<FUSED> s_fused;
    long long k_end, n_scored_f = 0;
    if constexpr (FUSED) sched_locate_in_block<NW>(a, r, s_fused, k_end, n_scored_f);
    else k_end = a.k_end[r];
    const long long n_units = ((k_end + 63) >> 6) * a.split; // work units: (batch, part)
    __shared__ u64 s_key[NW];
    __shared__ long long s_idx[NW];
    u64 best_key = SC_KEY_EMPTY;
    long long best_idx = -1;
    if (wave < n_units) {
        const SchedLaneConst L64 = a.in.lane_tab[li];
        SchedLane32 L;
        L.scored_mask = (unsigned)L64.scored_mask; L.weekend_mask = (unsigned)L64.weekend_mask;
        L.mask2 = (unsigned)L64.mask2; L.h3mask = (unsigned)L64.h3mask; L.wdmask = (unsigned)L64.wdmask;
        L.gsize = L64.gsize; L.flags = L64.flags;
        const int base = li < S ? (FUSED ? s_fused.state[li] : sched_state_at(a, r, li)) : -1 - li; // both halves hold the base vector
        unsigned eq_base = 0;
#pragma unroll
        for (int q = 0; q < 32; q++)
            if (q < D && base == __builtin_amdgcn_readlane(base, q)) eq_base |= 1u << q;
        eq_base &= L.scored_mask;
        const int hb_base = sched_holiday_bit(a.in, base, li);
        const ls_sched_move *mv = a.moves + (long long)r * a.max_moves;

        for (long long unit = wave; unit < n_units; unit += n_waves) {
            const long long batch = unit / a.split;
            const int part = (int)(unit % a.split);
            const long long k = (batch << 6) + lane;
            // ---- per-lane: this lane's move, its edit, its holiday bits, its half of the key
            int type = -1, ma = 0, mb = 0;
            if (k < k_end) { const ls_sched_move m = FUSED ? s_fused.moves[FUSED ? k : 0] : mv[k]; type = m.type; ma = m.a; mb = m.b; }
            const bool ok = sched_move_ok(type, ma, mb, S, E);
            const int ia = ok ? ma : 0, ib = (ok && type == 1) ? mb : ia;
            const int ea = __shfl(base, ia, 64), eb = __shfl(base, ib, 64);
            const int pos1 = ia, new1 = type == 0 ? mb : eb;
            const int pos2 = ib, new2 = type == 0 ? mb : ea;
            const bool active = ok && !(new1 == ea && new2 == eb) && (lane & (a.split - 1)) == part;
            int hbits = 0;
            u64 lowkey = 0;
            if (active) {
                hbits = sched_holiday_bit(a.in, new1, pos1) | (sched_holiday_bit(a.in, new2, pos2) << 1);
                u64 e1, e2 = (u64)1 << 21;
                if (type == 0) e1 = sc_edit_key(pos1, new1, ea);
                else {
                    const bool a_first = pos1 < pos2;
                    e1 = a_first ? sc_edit_key(pos1, new1, ea) : sc_edit_key(pos2, new2, eb);
                    e2 = a_first ? sc_edit_key(pos2, new2, eb) : sc_edit_key(pos1, new1, ea);
                }
                lowkey = (e1 << 23) | e2;
            }
            const int packed = pos1 | (pos2 << 8) | (hbits << 16);
            int my_hs = 0;
            // ---- wave: two non-tabu candidates of the batch per pass (lower half / upper half)
            u64 todo = __ballot(active);
            while (todo) {
                const int jA = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const bool haveB = todo != 0;
                const int jB = haveB ? __ffsll((long long)todo) - 1 : jA;
                todo &= todo - 1;
                const int pkA = __builtin_amdgcn_readlane(packed, jA), pkB = __builtin_amdgcn_readlane(packed, jB);
                const int n1A = __builtin_amdgcn_readlane(new1, jA), n1B = __builtin_amdgcn_readlane(new1, jB);
                const int n2A = __builtin_amdgcn_readlane(new2, jA), n2B = __builtin_amdgcn_readlane(new2, jB);
                const int pk = upper ? pkB : pkA, n1 = upper ? n1B : n1A, n2 = upper ? n2B : n2A;
                const int p1 = pk & 0xff, p2 = (pk >> 8) & 0xff;
                int e = base;
                if (li == p1) e = n1;
                if (li == p2) e = n2;
                const bool c1 = e == n1, c2 = e == n2;
                const u64 b1 = __ballot(c1), b2 = __ballot(c2);
                const unsigned m1 = (upper ? (unsigned)(b1 >> 32) : (unsigned)b1) & L.scored_mask;
                const unsigned m2 = (upper ? (unsigned)(b2 >> 32) : (unsigned)b2) & L.scored_mask;
                const unsigned bit1 = p1 < D ? 1u << p1 : 0, bit2 = p2 < D ? 1u << p2 : 0;
                unsigned eq = eq_base & ~(bit1 | bit2);
                if (c1) eq |= bit1;
                if (c2) eq |= bit2;
                if (li == p1) eq = m1;
                if (li == p2) eq = m2;
                int hb = hb_base;
                if (li == p1) hb = (pk >> 16) & 1;
                if (li == p2) hb = (pk >> 17) & 1;
                const SchedPartial32 pt = sched_terms32(D, L, li, lane, eq, hb);
                const int hs = half_sum_i32(pt.hs);
                const unsigned tor = half_or_u32(pt.tot_onehot), wor = half_or_u32(pt.wk_onehot);
#define LS_FINISH(LN, J)                                                                              \
    {                                                                                                 \
        const unsigned hsx = (unsigned)__builtin_amdgcn_readlane(hs, LN);                             \
        const unsigned t_or = (unsigned)__builtin_amdgcn_readlane((int)tor, LN);                      \
        const unsigned w_or = (unsigned)__builtin_amdgcn_readlane((int)wor, LN);                      \
        const int s3 = (31 - __clz((int)t_or)) - (__ffs((int)t_or) - 1);                              \
        const int s4 = (31 - __clz((int)w_or)) - (__ffs((int)w_or) - 1);                              \
        const int hard = (int)(hsx & 0xffffu), soft = (int)(hsx >> 16) + s3 + s4;                     \
        if (lane == (J)) my_hs = (hard << 8) | soft;                                                  \
    }
                LS_FINISH(31, jA)
                if (haveB) LS_FINISH(63, jB)
#undef LS_FINISH
            }
            // ---- per-lane: assemble (key, index); optional per-candidate outputs
            if (active) {
                const int hard = my_hs >> 8, soft = my_hs & 0xff;
                const u64 key = ((u64)(unsigned)hard << 54) | ((u64)(unsigned)soft << 46) | lowkey;
                if (key < best_key) { best_key = key; best_idx = k; }
                if (a.hard_out) a.hard_out[(long long)r * a.max_moves + k] = hard;
                if (a.soft_out) a.soft_out[(long long)r * a.max_moves + k] = soft;
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned klo = __shfl_xor((unsigned)best_key, m, 64), khi = __shfl_xor((unsigned)(best_key >> 32), m, 64);
        const unsigned ilo = __shfl_xor((unsigned)best_idx, m, 64), ihi = __shfl_xor((unsigned)((u64)best_idx >> 32), m, 64);
        const u64 ok = ((u64)khi << 32) | klo;
        const long long oi = (long long)(((u64)ihi << 32) | ilo);
        if (ok < best_key || (ok == best_key && ok != SC_KEY_EMPTY && oi < best_idx)) { best_key = ok; best_idx = oi; }
    }
    if (lane == 0) { s_key[wib] = best_key; s_idx[wib] = best_idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < NW; i++)
            if (s_key[i] < best_key || (s_key[i] == best_key && best_key != SC_KEY_EMPTY && s_idx[i] < best_idx)) {
                best_key = s_key[i]; best_idx = s_idx[i];
            }
        if (FUSED) {
            sched_write_result_n(a, r, best_key, best_idx, n_scored_f);
        } else if (a.blocks_per_restart == 1) {
            sched_write_result(a, r, best_key, best_idx);
        } else {
            const long long slot = (long long)r * a.blocks_per_restart + blockIdx.x / a.n_restarts;
            a.part_key[slot] = best_key;
            a.part_idx[slot] = best_idx;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Pass 2, DELTA formulation (the default): one LANE per candidate.  Every term of the score is a sum over
// employees of a function of the employee's day set (a bitmask over the scored days), or a range over such
// functions; a candidate changes the day sets of exactly two employees X (the employee of the first edited
// day) and Y (the new employee of that day):
//     ChangeDay(p1 <- Y):  X' = X - p1,        Y' = Y + p1        (lib.rs:466-470)
//     SwapDays(p1, p2):    X' = X - p1 + p2,   Y' = Y - p2 + p1   (lib.rs:471-478; Y = employee of p2)
// (bits of unscored stored days are never in a set).  Per block, wavefront 0 tabulates the base state once
// (lane = day): the day set of each day's employee, its H4 / S1 values, holiday bits, and for the three
// "range" terms the set of values present plus how many employees hold each value.  A lane then computes
//     H1      holiday bits of the edited days, new - old                                     lib.rs:273-280
//     H2+H3   popcount(adj[p] & set) for the sets that lose / gain day p (adj = the static pair relation) :286-315
//     H4, S1  F14 / F7 of X', Y' minus the tabulated values of X, Y; F = number of windows in which the set
//             holds more than 3 of 14 / 2 of 7 days, by bit-sliced sliding sums over the mask       :318-339
//     S2      per edited Mon-Fri weekday: the set of per-employee day counts on that weekday with X, Y
//             replaced; min if it is below the group size                                           :194-218
//     S3, S4  the sets of day counts / weekend-day counts with X, Y replaced; highest - lowest      :345-365
// -- about 350 instructions per LANE for <= 32-day schedules, where the wavefront-per-candidate kernels above
// spend 66 (two candidates per pass) or 135 wavefront instructions per CANDIDATE.

template <typename M> struct SchedMaskOps;
template <> struct SchedMaskOps<unsigned> {
    static __device__ __forceinline__ int popc(unsigned x) { return __popc(x); }
    static __device__ __forceinline__ int lsb(unsigned x) { return __ffs((int)x) - 1; }
    static __device__ __forceinline__ int msb(unsigned x) { return 31 - __clz((int)x); }
};
template <> struct SchedMaskOps<u64> {
    static __device__ __forceinline__ int popc(u64 x) { return __popcll(x); }
    static __device__ __forceinline__ int lsb(u64 x) { return __ffsll((long long)x) - 1; }
    static __device__ __forceinline__ int msb(u64 x) { return 63 - __clzll((long long)x); }
};

// the base state of one restart, tabulated in LDS
template <typename M> struct SchedBaseTab {
    int e[64];             // employee of stored day d (a distinct negative id beyond the stored days)
    M eq[64];              // scored days worked by the employee of day d
    M adj[64];             // scored days paired with day d by H2 / H3 (0 for an unscored day)
    M wd[64];              // scored days on the weekday of day d (0: weekend or unscored)
    unsigned short g[64];  // F14 | F7 << 8 of eq[d]
    unsigned char hol[64]; // holiday bit of (employee of day d, day d)
    // the three range terms: which values are held by at least 1 / 2 / 3 employees
    u64 tot_b[3];          // bit i: scored-day count i + 1
    unsigned wk_b[3];      // bit i: i weekend days (among the employees present)
    unsigned grp_b[7][3];  // weekday residue r, bit i: i + 1 of its days
    int grp_size[7];       // days of the weekday group (0: weekend residue)
    int grp_v[7];          // S2 value of the group in the base state
    M valid7, valid14, weekend;
    int hard, soft_s1, soft_s2; // base hard score; S1 and S2 of the base state
};

// number of 7-day / 14-day windows (starting at the bits of valid7 / valid14) in which mask m holds more than
// 2 / more than 3 days: bit-sliced sliding sums 1 -> 2 -> 4, 3 -> 7 -> 14
template <typename M>
__device__ __forceinline__ void sched_window_counts(M m, M valid7, M valid14, int &f7, int &f14)
{
    typedef SchedMaskOps<M> O;
    const M a1 = m >> 1;
    const M b0 = m ^ a1, b1 = m & a1;                                  // two days from w
    const M c0 = b0 >> 2, c1 = b1 >> 2;
    const M t0 = b0 ^ c0, k0 = b0 & c0, x1 = b1 ^ c1;
    const M t1 = x1 ^ k0, t2 = (b1 & c1) | (k0 & x1);                  // four days from w (0..4)
    const M e0 = b0 >> 4, e1 = b1 >> 4, m6 = m >> 6;
    const M d0 = e0 ^ m6, d1 = e1 | (e0 & m6);                         // three days from w + 4 (0..3)
    const M u0 = t0 ^ d0, k = t0 & d0, y1 = t1 ^ d1;
    const M u1 = y1 ^ k, u2 = t2 | (t1 & d1) | (k & y1);               // seven days from w (0..7)
    f7 = O::popc((u2 | (u1 & u0)) & valid7);                           // >= 3
    const M v0 = u0 >> 7, v1 = u1 >> 7, v2 = u2 >> 7;
    const M j0 = u0 & v0, z1 = u1 ^ v1;
    const M j1 = (u1 & v1) | (j0 & z1), z2 = u2 ^ v2;
    const M x2 = z2 ^ j1, x3 = (u2 & v2) | (j1 & z2);                  // fourteen days from w, bits 2 and 3
    f14 = O::popc((x2 | x3) & valid14);                                // >= 4
}

#define LS_DPP_ROW_SHR(n) (0x110 + (n))
#define LS_DPP_WAVE_SHR1 0x138

// inclusive prefix OR over the lanes of the wavefront
__device__ __forceinline__ unsigned wave_prefix_or_u32(unsigned x)
{
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_ROW_SHR(1), 0xf, 0xf, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_ROW_SHR(2), 0xf, 0xf, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_ROW_SHR(4), 0xf, 0xf, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_ROW_SHR(8), 0xf, 0xf, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_ROW_BCAST15, 0xa, 0xf, false);
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, LS_DPP_ROW_BCAST31, 0xc, 0xf, false);
    return x;
}

// every lane contributes one value as a one-hot word (or 0): which values are contributed by >= 1, >= 2, >= 3 lanes.
// A lane's value is a repeat if an earlier lane has it (exclusive prefix OR), a second repeat if an earlier lane was
// already a repeat of it -- no atomics, no serialisation on popular values.
__device__ __forceinline__ void wave_value_sets_u32(unsigned onehot, unsigned &b1, unsigned &b2, unsigned &b3)
{
    const unsigned in1 = wave_prefix_or_u32(onehot);
    const unsigned ex1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)in1, LS_DPP_WAVE_SHR1, 0xf, 0xf, false);
    const unsigned dup = onehot & ex1;
    const unsigned in2 = wave_prefix_or_u32(dup);
    const unsigned ex2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)in2, LS_DPP_WAVE_SHR1, 0xf, 0xf, false);
    b1 = (unsigned)__builtin_amdgcn_readlane((int)in1, 63);
    b2 = (unsigned)__builtin_amdgcn_readlane((int)in2, 63);
    b3 = wave_or_u32(onehot & ex2);
}

// The table is filled by wavefronts 0, 1 and 2 of the block side by side (lane = day; each derives the day-equality
// row of its lane first); the caller synchronises the block after.
//   part 0: per-day rows (employee, day set, pair / weekday masks, H4 | S1 values, holiday bit), base hard score, S1
//   part 1: S3 / S4 value sets (day counts, weekend-day counts)
//   part 2: S2 value sets per weekday, group sizes and base values
template <typename M>
__device__ __forceinline__ void sched_build_base_tab(const SchedInst &in, int part, int e /* employee of day `lane`; -1 - lane beyond S */,
                                                     int hol /* holiday bit of (e, lane), gathered by the caller (part 0) */,
                                                     const SchedLaneConst &L, int lane, SchedBaseTab<M> &T, M *emp = nullptr)
{
    typedef SchedMaskOps<M> O;
    const int D = in.D;
    M eq = 0;
    if (emp) { // the employee -> scored-days table (all zero between requests): every calling wavefront ORs the same bits in, then reads its own back
        if (lane < D && e >= 0) __hip_atomic_fetch_or(&emp[e], (M)1 << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (e >= 0) eq = __hip_atomic_load(&emp[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
#pragma unroll
        for (int q = 0; q < (int)sizeof(M) * 8; q++)
            if (q < D && e == __builtin_amdgcn_readlane(e, q)) eq |= (M)1 << q;
    }
    LS_SC_STAMP(8);
    const bool scored = L.flags & 1;
    const M low = ((M)1 << (lane & ((int)sizeof(M) * 8 - 1))) - 1;
    const M weekend = (M)L.weekend_mask, wdm = (M)L.wdmask;
    const bool rep = scored && (eq & low) == 0; // first scored day of its employee
    if (part == 0) {
        const M valid7 = D >= 7 ? (M)(L.scored_mask >> 6) : 0, valid14 = D >= 14 ? (M)(L.scored_mask >> 13) : 0;
        int f7, f14;
        sched_window_counts<M>(eq, valid7, valid14, f7, f14);
        T.e[lane] = e;
        T.eq[lane] = eq;
        T.adj[lane] = (M)L.adj;
        T.wd[lane] = wdm;
        T.g[lane] = (unsigned short)(f14 | (f7 << 8));
        T.hol[lane] = (unsigned char)hol;
        int hard_l = 0, soft_l = 0;
        if (scored) hard_l = hol + O::popc(eq & (M)(L.mask2 | L.h3mask));
        if (rep) { hard_l += f14; soft_l = f7; }
        const int hs = wave_sum_i32(hard_l | (soft_l << 16)); // hard <= 316, S1 <= 58 * 64: 16-bit fields
        if (lane == 0) {
            T.valid7 = valid7; T.valid14 = valid14; T.weekend = weekend;
            T.hard = hs & 0xffff;
            T.soft_s1 = hs >> 16;
        }
    } else if (part == 1) {
        u64 tot_onehot = 0;
        unsigned wk_onehot = 0;
        if (rep) {
            tot_onehot = 1ull << (O::popc(eq) - 1);
            wk_onehot = 1u << O::popc(eq & weekend);
        }
        unsigned tl[3], th[3] = {0, 0, 0}, wk[3];
        wave_value_sets_u32((unsigned)tot_onehot, tl[0], tl[1], tl[2]);
        if (sizeof(M) > 4) wave_value_sets_u32((unsigned)(tot_onehot >> 32), th[0], th[1], th[2]);
        wave_value_sets_u32(wk_onehot, wk[0], wk[1], wk[2]);
        if (lane < 3) {
            const unsigned lo = lane == 0 ? tl[0] : (lane == 1 ? tl[1] : tl[2]), hi = lane == 0 ? th[0] : (lane == 1 ? th[1] : th[2]);
            T.tot_b[lane] = ((u64)hi << 32) | lo;
            T.wk_b[lane] = lane == 0 ? wk[0] : (lane == 1 ? wk[1] : wk[2]);
        }
    } else {
        // weekday groups: residue r = lane % 7 occupies bits 10 * (r % 3) .. + 9 of word r / 3
        const int gr = lane % 7;
        unsigned g_onehot = 0;
        if (wdm != 0 && (eq & wdm & low) == 0) g_onehot = 1u << (10 * (gr % 3) + O::popc(eq & wdm) - 1); // first day of its employee on this weekday
        unsigned g0[3], g1[3], g2[3];
        wave_value_sets_u32(gr < 3 ? g_onehot : 0, g0[0], g0[1], g0[2]);
        wave_value_sets_u32(gr >= 3 && gr < 6 ? g_onehot : 0, g1[0], g1[1], g1[2]);
        wave_value_sets_u32(gr == 6 ? g_onehot : 0, g2[0], g2[1], g2[2]);
        int v = 0;
        if (lane < 7) {
            const int sh = 10 * (lane % 3);
            unsigned gb[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                gb[i] = ((lane < 3 ? g0[i] : (lane < 6 ? g1[i] : g2[i])) >> sh) & 0x3ffu;
                T.grp_b[lane][i] = gb[i];
            }
            const int gsize = (L.flags & 2) ? L.gsize : 0; // lanes 0..6 are the first days of their residues
            if (gsize > 0) {
                const int mn = __ffs((int)gb[0]); // smallest per-employee count on this weekday
                if (mn < gsize) v = mn;
            }
            T.grp_size[lane] = gsize;
            T.grp_v[lane] = v;
        }
        const int s2 = wave_sum_i32(v);
        if (lane == 0) T.soft_s2 = s2;
    }
    LS_SC_STAMP(11);
}

// the set of values held by the employees after X and Y trade their old values for new ones; b[0..2] = values held by
// at least 1 / 2 / 3 employees in the base state
template <typename B>
__device__ __forceinline__ B sched_replace_values(const B *b, bool hXo, int vXo, bool hYo, int vYo, bool hXn, int vXn, bool hYn, int vYn)
{
    B bits = b[0];
    const bool both = hXo && hYo && vXo == vYo;
    const B others = both ? b[2] : b[1]; // values that some employee other than X (and Y) holds as well
    if (hXo && !((others >> vXo) & 1)) bits &= ~((B)1 << vXo);
    if (hYo && !((others >> vYo) & 1)) bits &= ~((B)1 << vYo);
    if (hXn) bits |= (B)1 << vXn;
    if (hYn) bits |= (B)1 << vYn;
    return bits;
}

// (hard, soft) of one candidate; o1 / o2 = base employees of p1 / p2, n1 = new employee of p1, hbits = holiday bits
// of (n1, p1) | (n2, p2) << 1.  Only called for well-formed, non-tabu candidates.
template <typename M>
__device__ __forceinline__ void sched_delta_score(const SchedBaseTab<M> &T, int e_lane, int D, bool swap, int p1, int p2, int n1,
                                                  int hbits, int &hard_out, int &soft_out, const M *emp = nullptr)
{
    typedef SchedMaskOps<M> O;
    const M bit1 = p1 < D ? (M)1 << p1 : 0, bit2 = (swap && p2 < D) ? (M)1 << p2 : 0;
    const M mX = T.eq[p1];
    const int gX = T.g[p1];
    M mY;
    int gY;
    if (swap) {
        mY = T.eq[p2];
        gY = T.g[p2];
    } else if (emp) {
        mY = emp[n1]; // the day set of an arbitrary employee, from the block's employee -> scored-days table
        gY = mY ? T.g[O::lsb(mY)] : 0;
    } else {
        mY = 0; // ... or by comparing with every scored day (e_lane: lane d holds day d)
#pragma unroll
        for (int d = 0; d < (int)sizeof(M) * 8; d++)
            if (d < D && __builtin_amdgcn_readlane(e_lane, d) == n1) mY |= (M)1 << d;
        gY = mY ? T.g[O::lsb(mY)] : 0;
    }
    const M mXn = (mX & ~bit1) | bit2, mYn = (mY & ~bit2) | bit1;
    const M A1 = T.adj[p1], A2 = swap ? T.adj[p2] : 0;
    int f7X, f14X, f7Y, f14Y;
    sched_window_counts<M>(mXn, T.valid7, T.valid14, f7X, f14X);
    sched_window_counts<M>(mYn, T.valid7, T.valid14, f7Y, f14Y);
    int hard = T.hard + (hbits & 1) - T.hol[p1];
    if (swap) hard += ((hbits >> 1) & 1) - T.hol[p2];
    hard += O::popc(A1 & mYn) - O::popc(A1 & mX) + O::popc(A2 & mXn) - O::popc(A2 & mY);
    hard += f14X + f14Y - (gX & 0xff) - (gY & 0xff);
    int soft = T.soft_s1 + T.soft_s2 + f7X + f7Y - (gX >> 8) - (gY >> 8);
    const int cX = O::popc(mX), cY = O::popc(mY), cXn = O::popc(mXn), cYn = O::popc(mYn);
    const bool hXo = cX > 0, hYo = cY > 0, hXn = cXn > 0, hYn = cYn > 0;
    {   // S3: day counts (value index = count - 1)
        const u64 b = sched_replace_values<u64>(T.tot_b, hXo, cX - 1, hYo, cY - 1, hXn, cXn - 1, hYn, cYn - 1);
        soft += (63 - __clzll((long long)b)) - (__ffsll((long long)b) - 1);
    }
    {   // S4: weekend-day counts of the employees present
        const M wk = T.weekend;
        const unsigned b = sched_replace_values<unsigned>(T.wk_b, hXo, O::popc(mX & wk), hYo, O::popc(mY & wk), hXn,
                                                          O::popc(mXn & wk), hYn, O::popc(mYn & wk));
        soft += (31 - __clz((int)b)) - (__ffs((int)b) - 1);
    }
    // S2: the weekday groups of the edited days
    const M G1 = T.wd[p1], G2 = swap ? T.wd[p2] : 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const M G = i == 0 ? G1 : G2;
        if (G == 0 || (i == 1 && G == G1)) continue;
        const int r = (i == 0 ? p1 : p2) % 7;
        const int a = O::popc(mX & G), b = O::popc(mY & G), an = O::popc(mXn & G), bn = O::popc(mYn & G);
        const unsigned bits = sched_replace_values<unsigned>(T.grp_b[r], a > 0, a - 1, b > 0, b - 1, an > 0, an - 1,
                                                             bn > 0, bn - 1);
        const int mn = __ffs((int)bits);
        soft += (mn < T.grp_size[r] ? mn : 0) - T.grp_v[r];
    }
    hard_out = hard;
    soft_out = soft;
}

// one candidate of a delta kernel: its edit, and whether it is a real (well-formed, non-tabu) candidate.
// Needs only the wavefront's copy of the state (e_lane: lane d holds day d), not the table.
struct SchedEdit {
    bool active;
    bool swap;
    int p1, p2, ea, eb, new1, new2;
};

__device__ __forceinline__ SchedEdit sched_edit_of(int e_lane, int S, int E, int type, int ma, int mb)
{
    SchedEdit ed;
    const bool ok = sched_move_ok(type, ma, mb, S, E);
    ed.swap = type == 1;
    ed.p1 = ok ? ma : 0;
    ed.p2 = (ok && type == 1) ? mb : ed.p1;
    ed.ea = __shfl(e_lane, ed.p1, 64);
    ed.eb = __shfl(e_lane, ed.p2, 64);
    // ChangeDay: day a <- employee b (lib.rs:466-470);  SwapDays: a <- e[b], b <- e[a] (:471-478)
    ed.new1 = type == 0 ? mb : ed.eb;
    ed.new2 = type == 0 ? mb : ed.ea;
    ed.active = ok && !(ed.new1 == ed.ea && ed.new2 == ed.eb); // candidate != current (tabu, local_search.rs:319)
    return ed;
}

// holiday bits of the edited days under their new employees: a gather from global memory that depends on the move
// and the state only, so it is issued before the block waits for the table
__device__ __forceinline__ int sched_edit_hbits(const SchedInst &in, const SchedEdit &ed)
{
    if (!ed.active) return 0;
    return sched_holiday_bit(in, ed.new1, ed.p1) | (sched_holiday_bit(in, ed.new2, ed.p2) << 1);
}

// the argmin key of an active candidate (and its scores)
template <typename M>
__device__ __forceinline__ u64 sched_delta_key(const SchedBaseTab<M> &T, int D, int e_lane, const SchedEdit &ed, int hbits, int &hard,
                                               int &soft, const M *emp = nullptr)
{
    u64 e1, e2 = (u64)1 << 21; // "no second edit" sorts between lowering and raising edits
    if (!ed.swap) e1 = sc_edit_key(ed.p1, ed.new1, ed.ea);
    else {
        const bool a_first = ed.p1 < ed.p2;
        e1 = a_first ? sc_edit_key(ed.p1, ed.new1, ed.ea) : sc_edit_key(ed.p2, ed.new2, ed.eb);
        e2 = a_first ? sc_edit_key(ed.p2, ed.new2, ed.eb) : sc_edit_key(ed.p1, ed.new1, ed.ea);
    }
    sched_delta_score<M>(T, e_lane, D, ed.swap, ed.p1, ed.p2, ed.new1, hbits, hard, soft, emp);
    return ((u64)(unsigned)hard << 54) | ((u64)(unsigned)soft << 46) | (e1 << 23) | e2;
}

// candidate k of the exhaustive ScheduleMoveProposer (lib.rs:493-559), generated in place: day d = k / E,
// employee (state[d] + 1 + i) mod E for i = k mod E (the day's current employee last)
__device__ __forceinline__ void sched_gen_exhaustive(int e_lane, int E, long long k, int &type, int &ma, int &mb)
{
    const unsigned d = (unsigned)k / (unsigned)E, i = (unsigned)k - d * (unsigned)E;
    type = 0;
    ma = (int)d;
    mb = (int)(((unsigned)__shfl(e_lane, (int)(d & 63), 64) + 1u + i) % (unsigned)E);
}

__device__ __forceinline__ void sched_min_key(u64 &bk, long long &bi, u64 ok, long long oi)
{
    if (ok < bk || (ok == bk && ok != SC_KEY_EMPTY && oi < bi)) { bk = ok; bi = oi; }
}

// every lane ends up with the wavefront's minimum (key, index), first index of equal keys: two DPP reductions
__device__ __forceinline__ void sched_wave_min_key(u64 &best_key, long long &best_idx)
{
    const u64 mk = wave_min_u64(best_key);
    const u64 mi = wave_min_u64(best_key == mk ? (u64)best_idx : ~0ull);
    best_key = mk;
    best_idx = mk == SC_KEY_EMPTY ? -1 : (long long)mi;
}

// block minimum: every wavefront's minimum goes through LDS, wavefront 0 reduces the NW records in its lanes
// (one LDS read per lane instead of a serial walk by one thread); valid in wavefront 0 afterwards
template <int NW>
__device__ __forceinline__ void sched_block_min_key(u64 &key, long long &idx, u64 *s_key, long long *s_idx, int lane, int wib)
{
    sched_wave_min_key(key, idx);
    if (lane == 0) { s_key[wib] = key; s_idx[wib] = idx; }
    __syncthreads();
    if (wib == 0) {
        key = lane < NW ? s_key[lane] : SC_KEY_EMPTY;
        idx = lane < NW ? s_idx[lane] : -1;
        sched_wave_min_key(key, idx);
    }
}

// Pass 2 after count + locate: the chain that serves per-candidate score dumps of long move lists.
template <typename M, int NW>
__global__ __launch_bounds__(64 * NW) void sched_eval_delta_kernel(SchedEvalArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = (int)(blockIdx.x % a.n_restarts);
    const int wave = (int)(blockIdx.x / a.n_restarts) * NW + wib;
    const int n_waves = (int)(gridDim.x / a.n_restarts) * NW;
    const int S = a.in.S, D = a.in.D, E = a.in.n_employees;
    __shared__ SchedBaseTab<M> T;
    __shared__ u64 s_key[NW];
    __shared__ long long s_idx[NW];
    SchedLaneConst L{};
    if (wib < 3) L = a.in.lane_tab[lane];
    const long long k_end = a.k_end[r];
    const int e_lane = lane < S ? sched_state_at(a, r, lane) : -1 - lane;
    const int hol_base = wib == 0 ? sched_holiday_bit(a.in, e_lane, lane) : 0; // in flight until the table build needs it
    const long long n_batches = (k_end + 63) >> 6;
    const ls_sched_move *mv = a.moves + (long long)r * a.max_moves;
    // the first batch's moves, edits and holiday gathers are in flight while wavefronts 0..2 tabulate the base state
    long long batch = wave;
    int type = -1, ma = 0, mb = 0;
    {
        const long long k = (batch << 6) + lane;
        if (k < k_end) { const ls_sched_move m = mv[k]; type = m.type; ma = m.a; mb = m.b; }
    }
    SchedEdit ed = sched_edit_of(e_lane, S, E, type, ma, mb);
    int hbits = sched_edit_hbits(a.in, ed);
    if (wib < 3) sched_build_base_tab<M>(a.in, wib, e_lane, hol_base, L, lane, T);
    __syncthreads();
    u64 best_key = SC_KEY_EMPTY;
    long long best_idx = -1;
    while (batch < n_batches) {
        const long long k = (batch << 6) + lane;
        if (ed.active) {
            int hard, soft;
            const u64 key = sched_delta_key<M>(T, D, e_lane, ed, hbits, hard, soft);
            if (key < best_key) { best_key = key; best_idx = k; } // ascending k per lane: first of equals kept
            if (a.hard_out) a.hard_out[(long long)r * a.max_moves + k] = hard;
            if (a.soft_out) a.soft_out[(long long)r * a.max_moves + k] = soft;
        }
        batch += n_waves;
        if (batch < n_batches) {
            const long long k2 = (batch << 6) + lane;
            type = -1;
            if (k2 < k_end) { const ls_sched_move m = mv[k2]; type = m.type; ma = m.a; mb = m.b; }
            ed = sched_edit_of(e_lane, S, E, type, ma, mb);
            hbits = sched_edit_hbits(a.in, ed);
        }
    }
    sched_block_min_key<NW>(best_key, best_idx, s_key, s_idx, lane, wib);
    if (threadIdx.x == 0) {
        if (a.blocks_per_restart == 1) {
            sched_write_result(a, r, best_key, best_idx);
        } else {
            const long long slot = (long long)r * a.blocks_per_restart + blockIdx.x / a.n_restarts;
            a.part_key[slot] = best_key;
            a.part_idx[slot] = best_idx;
        }
    }
}

// Move lists of at most 64 * NW entries (the reference example's window-100 iteration): ONE block per restart does the
// whole pipeline.  Each lane loads its own move; whether it is a real candidate follows from the move and the state
// alone, so the ballots that place take(window) (local_search.rs:319-321) are written before the block waits for the
// table, and the same barrier publishes both: two barriers in all (table + counts, block minimum).
template <typename M, int NW>
__global__ __launch_bounds__(64 * NW) void sched_eval_small_kernel(SchedEvalArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = (int)blockIdx.x;
    const int S = a.in.S, D = a.in.D, E = a.in.n_employees;
    __shared__ SchedBaseTab<M> T;
    __shared__ u64 s_key[NW];
    __shared__ long long s_idx[NW];
    __shared__ int s_wcnt[NW];
    LS_SC_STAMP(0);
    SchedLaneConst L{};
    if (wib < 3) L = a.in.lane_tab[lane];
    const int e_lane = lane < S ? sched_state_at(a, r, lane) : -1 - lane;
    const int hol_base = wib == 0 ? sched_holiday_bit(a.in, e_lane, lane) : 0; // in flight until the table build needs it
    const long long nm = sched_nm(a, r);
    const long long k = threadIdx.x;
    int type = -1, ma = 0, mb = 0;
    if (k < nm) { const ls_sched_move m = a.moves[(long long)r * a.max_moves + k]; type = m.type; ma = m.a; mb = m.b; }
    LS_SC_STAMP(1);
    const SchedEdit ed = sched_edit_of(e_lane, S, E, type, ma, mb);
    const u64 bal = __ballot(ed.active);
    if (lane == 0) s_wcnt[wib] = __popcll(bal);
    const int hbits = sched_edit_hbits(a.in, ed);
    LS_SC_STAMP(2);
    if (wib < 3) sched_build_base_tab<M>(a.in, wib, e_lane, hol_base, L, lane, T);
    LS_SC_STAMP(3);
    __syncthreads();
    LS_SC_STAMP(4);
    int before = 0, total = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        if (i < wib) before += s_wcnt[i];
        total += s_wcnt[i];
    }
    const long long rank = before + __popcll(bal & ((2ull << lane) - 1)); // 1-based among the real candidates
    u64 best_key = SC_KEY_EMPTY;
    long long best_idx = -1;
    if (ed.active && rank <= a.window) {
        int hard, soft;
        best_key = sched_delta_key<M>(T, D, e_lane, ed, hbits, hard, soft);
        best_idx = k;
        if (a.hard_out) a.hard_out[(long long)r * a.max_moves + k] = hard;
        if (a.soft_out) a.soft_out[(long long)r * a.max_moves + k] = soft;
    }
    LS_SC_STAMP(5);
    sched_block_min_key<NW>(best_key, best_idx, s_key, s_idx, lane, wib);
    LS_SC_STAMP(6);
    if (threadIdx.x == 0) {
        const long long n_scored = a.window <= 0 ? 0 : (total < a.window ? total : a.window);
        sched_write_result_n(a, r, best_key, best_idx, n_scored);
        LS_SC_STAMP(7);
    }
}

// ---------------------------------------------------------------------------------------------
// The resident scheduling LS step (srv_common.hpp: mailbox protocol, leaving rules).  One workgroup that is never
// relaunched serves ls_sched_eval_resident for move lists of at most 64 * NW entries -- the reference example's window-100
// iteration (employee-scheduling/src/main.rs:26): the launch + dispatch + completion poll around a 7 us kernel were more
// than half of the 17.8 us LS iteration.  The workgroup keeps the slot's state vector in LDS and each lane its move of the
// resident stream (the stream never changes during a run: the reference's iterator clones the generator, lib.rs:488); a
// request names how many moves of the stream are candidates, the window and the state's id, the answer is one line.  The
// winner is applied to the LDS state (the host applies it to its copy: local_search.rs:335 always moves to the winner), so
// a steady-state LS iteration moves 64 bytes down and 32 bytes up; a state the workgroup does not hold (perturbation,
// relaunch) is read from the pinned shadow copy.  Body = sched_eval_small_kernel.
constexpr unsigned SCS_F_EVAL = 1u, SCS_F_SCORE = 2u; // neighbourhood of the state | score of the state (get_scored_solution, lib.rs:261-375)

struct SchedSrvArgs {
    SrvLink link;
    SchedInst in;
    const ls_sched_move *moves; // the resident stream (device memory), fixed while the launch lives
    long long n_resident;
    const int *shadow;          // pinned host: the state of the current request [64]
    int emp_entries;            // entries of the employee -> scored-days table in dynamic LDS (0: none)
};

template <typename M, int NW>
__global__ __launch_bounds__(64 * NW) void sched_server_kernel(SchedSrvArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int S = a.in.S, D = a.in.D, E = a.in.n_employees;
    __shared__ SchedBaseTab<M> T;
    __shared__ u64 s_key[NW];
    __shared__ long long s_idx[NW];
    __shared__ int s_wcnt[NW];
    __shared__ unsigned s_req[16];
    __shared__ int s_state[64];
    __shared__ int s_leave;
    // employee -> scored days of the current state (see sched_grid_server_kernel): zero between requests
    extern __shared__ __align__(16) unsigned char s_dyn[];
    M *const emp = a.emp_entries > 0 ? reinterpret_cast<M *>(s_dyn) : nullptr;
    for (int i = tid; i < a.emp_entries; i += 64 * NW) emp[i] = 0;
    SchedLaneConst L{};
    if (wib < 3) L = a.in.lane_tab[lane];
    // this lane's move of the resident stream: loaded once, the stream is fixed while the launch lives
    int mtype = -1, mma = 0, mmb = 0;
    if (tid < a.n_resident) { const ls_sched_move m = a.moves[tid]; mtype = m.type; mma = m.a; mmb = m.b; }
    unsigned lds_state = 0; // id of the state in s_state (0: none); block-uniform
    SrvPoll poll{0, 0, 0};
    if (wib == 0) srv_poll_begin(poll, a.link, a.link.resp, lane);
    for (;;) {
        if (wib == 0) {
            const int leave = srv_poll_wait(poll, a.link, a.link.req, s_req, lane);
            if (lane == 0) s_leave = leave;
        }
        __syncthreads(); // B1: the request is in s_req, the previous winner is in s_state
        if (s_leave) break;
        const long long nm = (long long)s_req[1] < a.n_resident ? (long long)s_req[1] : a.n_resident;
        const long long window = (long long)s_req[2];
        const unsigned state_id = s_req[3] | (s_req[4] << 16);
        if (state_id != lds_state) { // block-uniform: not the state this workgroup holds
            if (tid < 64) s_state[tid] = tid < S ? srv_sys_load(a.shadow + tid) : 0;
            lds_state = state_id;
            __syncthreads();
        }
        const int e_lane = lane < S ? s_state[lane] : -1 - lane;
        const int hol_base = wib == 0 ? sched_holiday_bit(a.in, e_lane, lane) : 0;
        if (s_req[0] & SCS_F_SCORE) { // block-uniform: the state's own score, by the wavefront-per-state scorer (sched_score_kernel)
            if (wib == 0) {
                const u64 eq = sched_build_eq(D, e_lane) & L.scored_mask;
                const SchedScore r = sched_terms<false>(D, L, lane, eq, hol_base);
                unsigned pay = 0;
                if (lane == 3) pay = (unsigned)r.hard;
                else if (lane == 4) pay = (unsigned)r.soft;
                if (lane < 8) __hip_atomic_store(a.link.resp + lane, (poll.last_tag << 16) | (pay & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                srv_served(poll, a.link, lane);
            }
            continue;
        }
        const long long k = tid;
        const SchedEdit ed = sched_edit_of(e_lane, S, E, k < nm ? mtype : -1, mma, mmb);
        const u64 bal = __ballot(ed.active);
        if (lane == 0) s_wcnt[wib] = __popcll(bal);
        const int hbits = sched_edit_hbits(a.in, ed);
        if (wib < 3) sched_build_base_tab<M>(a.in, wib, e_lane, hol_base, L, lane, T, emp);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            if (i < wib) before += s_wcnt[i];
            total += s_wcnt[i];
        }
        const long long rank = before + __popcll(bal & ((2ull << lane) - 1)); // 1-based among the real candidates
        u64 best_key = SC_KEY_EMPTY;
        long long best_idx = -1;
        if (ed.active && rank <= window) {
            int hard, soft;
            best_key = sched_delta_key<M>(T, D, e_lane, ed, hbits, hard, soft, emp);
            best_idx = k;
        }
        sched_block_min_key<NW>(best_key, best_idx, s_key, s_idx, lane, wib); // valid in wavefront 0 afterwards
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
            if (lane < 8) __hip_atomic_store(a.link.resp + lane, (poll.last_tag << 16) | (pay & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            srv_served(poll, a.link, lane);
            if (has && lane == 0) { // the loop continues from the winner (local_search.rs:335)
                const ls_sched_move m = a.moves[best_idx];
                if (m.type == 0) s_state[m.a] = m.b;
                else { const int ea = s_state[m.a], eb = s_state[m.b]; s_state[m.a] = eb; s_state[m.b] = ea; }
            }
        }
    }
    if (tid == 0) srv_block_leaves(a.link);
}

// ---------------------------------------------------------------------------------------------
// The production pipeline for move lists beyond one block (no per-candidate score dump): TWO launches.
//   sched_eval_all_kernel   every chunk of a.chunk_moves moves is scored by one block, which records each
//                           candidate's key, the chunk's number of real (non-tabu) candidates and its minimum
//                           (key, index) -- where the window ends is not known yet, and scoring a candidate costs
//                           little more than testing it;
//   sched_finish_kernel     one block per restart: a prefix scan over the chunk counts finds the chunk in which
//                           take(window) stops (local_search.rs:319-321); chunks before it contribute their minima,
//                           the cut chunk its recorded keys up to the window, and the block writes the result.
// GEN: the candidates are those of the exhaustive proposer, generated in place (no move list in memory).
template <int NW> struct SchedFinishShared;
template <int NW, bool COHERENT = false>
__device__ __forceinline__ void sched_finish_block(const SchedEvalArgs &a, bool gen, int r, SchedFinishShared<NW> &sh);

template <typename M, int NW, bool GEN>
__global__ __launch_bounds__(64 * NW) void sched_eval_all_kernel(SchedEvalArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = (int)(blockIdx.x % a.n_restarts), c = (int)(blockIdx.x / a.n_restarts);
    const int S = a.in.S, D = a.in.D, E = a.in.n_employees;
    __shared__ SchedBaseTab<M> T;
    __shared__ u64 s_key[NW];
    __shared__ long long s_idx[NW];
    __shared__ int s_cnt[NW];
    LS_SC_STAMP(0);
    SchedLaneConst L{};
    if (wib < 3) L = a.in.lane_tab[lane];
    const int e_lane = lane < S ? sched_state_at(a, r, lane) : -1 - lane;
    const int hol_base = wib == 0 ? sched_holiday_bit(a.in, e_lane, lane) : 0; // in flight until the table build needs it
    const long long nm = GEN ? a.max_moves : sched_nm(a, r);
    const long long c0 = (long long)c * a.chunk_moves;
    const long long c1 = c0 + a.chunk_moves < nm ? c0 + a.chunk_moves : nm;
    const ls_sched_move *mv = GEN ? nullptr : a.moves + (long long)r * a.max_moves;
    u64 *keys = a.cand_keys + (long long)r * a.max_moves;
    // the first batch's moves, edits and holiday gathers are in flight while wavefront 0 tabulates the base state
    long long base = c0 + wib * 64;
    int type = -1, ma = 0, mb = 0;
    if constexpr (GEN) {
        sched_gen_exhaustive(e_lane, E, base + lane, type, ma, mb); // every lane takes part in the shuffle
        if (base + lane >= c1) type = -1;
    } else if (base + lane < c1) { const ls_sched_move m = mv[base + lane]; type = m.type; ma = m.a; mb = m.b; }
    SchedEdit ed = sched_edit_of(e_lane, S, E, type, ma, mb);
    LS_SC_STAMP(1);
    int hbits = sched_edit_hbits(a.in, ed);
    LS_SC_STAMP(2);
    // employee -> scored days of the base state (see sched_grid_server_kernel), zeroed by the whole block while its loads fly
    extern __shared__ __align__(16) unsigned char s_dyn[];
    // (a device-resident state is not validated by the host entry: with an entry outside [0, E) the block compares day by day,
    // as without the table -- the same answer for the same garbage, and no LDS access out of range; every wavefront decides alike)
    const bool e_bad = lane < S && (unsigned)e_lane >= (unsigned)E;
    M *const emp = a.emp_entries > 0 && __builtin_amdgcn_ballot_w64(e_bad) == 0 ? reinterpret_cast<M *>(s_dyn) : nullptr;
    if (emp) {
        const int n16 = (int)(((size_t)a.emp_entries * sizeof(M) + 15) / 16); // the launcher rounds the allocation up to 16 bytes
        for (int i = (int)threadIdx.x; i < n16; i += 64 * NW) reinterpret_cast<uint4 *>(s_dyn)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    if (wib < 3) sched_build_base_tab<M>(a.in, wib, e_lane, hol_base, L, lane, T, emp);
    LS_SC_STAMP(3);
    __syncthreads();
    LS_SC_STAMP(4);
    u64 best_key = SC_KEY_EMPTY;
    long long best_idx = -1;
    int cnt = 0;
    u64 last_key = SC_KEY_EMPTY;
    long long last_k = c1;
    while (base < c1) {
        const long long k = base + lane;
        u64 key = SC_KEY_EMPTY;
        if (ed.active) {
            int hard, soft;
            key = sched_delta_key<M>(T, D, e_lane, ed, hbits, hard, soft, emp);
            if (key < best_key) { best_key = key; best_idx = k; }
            cnt++;
        }
        base += 64 * NW;
        if (base >= c1) { last_key = key; last_k = k; break; } // stored after the block's record: the barrier below would wait for it
        if (k < c1) keys[k] = key;
        {
            type = -1;
            if constexpr (GEN) {
                sched_gen_exhaustive(e_lane, E, base + lane, type, ma, mb);
                if (base + lane >= c1) type = -1;
            } else if (base + lane < c1) { const ls_sched_move m = mv[base + lane]; type = m.type; ma = m.a; mb = m.b; }
            ed = sched_edit_of(e_lane, S, E, type, ma, mb);
            hbits = sched_edit_hbits(a.in, ed);
        }
    }
    LS_SC_STAMP(5);
    cnt = wave_sum_i32(cnt);
    if (lane == 0) s_cnt[wib] = cnt;
    LS_SC_STAMP(12);
    sched_block_min_key<NW>(best_key, best_idx, s_key, s_idx, lane, wib); // synchronises the block
    LS_SC_STAMP(13);
    if (wib == 0) cnt = wave_sum_i32(lane < NW ? s_cnt[lane] : 0);
    LS_SC_STAMP(14);
    if (threadIdx.x == 0) {
        const long long slot = (long long)r * a.n_chunks + c;
        a.part_key[slot] = best_key;
        a.part_idx[slot] = best_idx;
        a.chunk_counts[slot] = cnt;
    }
    if (last_k < c1) keys[last_k] = last_key;
    LS_SC_STAMP(6);
    if (a.ticket) {
        // ONE launch: the block that finishes last (a ticket per restart) does the finish step itself.  Release: the
        // barrier waits for every wavefront's stores, thread 0's agent-scope fence writes them back before the ticket.
        __shared__ SchedFinishShared<NW> s_fin;
        __shared__ int s_last;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(a.ticket + r, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = t == (unsigned)a.n_chunks - 1;
            if (s_last) a.ticket[r] = 0; // for the next launch
        }
        __syncthreads();
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            sched_finish_block<NW>(a, GEN, r, s_fin);
        }
        LS_SC_STAMP(7);
    }
}

template <int NW> struct SchedFinishShared {
    u64 key[NW];
    long long idx[NW];
    int wcnt[NW];
    int cut;
    long long before;
};

// the finish step of restart r by one block of NW wavefronts (see above).  COHERENT: the records were written by other
// workgroups of the SAME launch with agent-scope (write-through) stores and are read with agent-scope loads -- no cache
// write-back / invalidate between the XCDs' L2s is needed (the resident grid).
template <int NW, bool COHERENT>
__device__ __forceinline__ void sched_finish_block(const SchedEvalArgs &a, bool gen, int r, SchedFinishShared<NW> &sh)
{
    auto ld_i = [](const int *p) { return COHERENT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
    auto ld_k = [](const u64 *p) { return COHERENT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
    auto ld_l = [](const long long *p) { return COHERENT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    u64 *s_key = sh.key;
    long long *s_idx = sh.idx;
    int *s_wcnt = sh.wcnt;
    int &s_cut = sh.cut;
    long long &s_before = sh.before;
    const int nch = a.n_chunks;
    const int *counts = a.chunk_counts + (long long)r * nch;
    const u64 *pk = a.part_key + (long long)r * nch;
    const long long *pi = a.part_idx + (long long)r * nch;
    if (threadIdx.x == 0) { s_cut = -1; s_before = 0; }
    __syncthreads();
    u64 best_key = SC_KEY_EMPTY;
    long long best_idx = -1;
    long long run = 0; // real candidates in the chunks seen so far
    if (a.window > 0) {
        for (int t0 = 0; t0 < nch; t0 += 64 * NW) {
            const int b = t0 + (int)threadIdx.x;
            const int cb = b < nch ? ld_i(counts + b) : 0;
            int incl = cb;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            if (lane == 63) s_wcnt[wib] = incl;
            __syncthreads();
            int before = 0, total = 0;
            for (int i = 0; i < NW; i++) {
                if (i < wib) before += s_wcnt[i];
                total += s_wcnt[i];
            }
            const long long through = run + before + incl; // real candidates up to and including chunk b
            if (b < nch) {
                if (through <= a.window) sched_min_key(best_key, best_idx, ld_k(pk + b), ld_l(pi + b)); // the whole chunk is inside the window
                else if (through - cb < a.window) { s_cut = b; s_before = through - cb; }  // the window ends inside this chunk
            }
            __syncthreads();
            run += total;
            if (run >= a.window) break;
        }
    }
    const long long n_scored = a.window <= 0 ? 0 : (run < a.window ? run : a.window);
    const int cut = s_cut;
    if (cut >= 0) { // the cut chunk: its recorded keys, up to the window
        const long long need = a.window - s_before; // 1 <= need < real candidates of the chunk
        const long long nm = gen ? a.max_moves : sched_nm(a, r);
        const long long c0 = (long long)cut * a.chunk_moves;
        const long long c1 = c0 + a.chunk_moves < nm ? c0 + a.chunk_moves : nm;
        const u64 *keys = a.cand_keys + (long long)r * a.max_moves;
        long long taken = 0;
        for (long long base = c0; base < c1 && taken < need; base += 64 * NW) {
            const long long k = base + threadIdx.x;
            const u64 key = k < c1 ? ld_k(keys + k) : SC_KEY_EMPTY;
            const u64 bal = __ballot(key != SC_KEY_EMPTY);
            __syncthreads(); // s_wcnt is reused
            if (lane == 0) s_wcnt[wib] = __popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
            for (int i = 0; i < NW; i++) {
                if (i < wib) before += s_wcnt[i];
                total += s_wcnt[i];
            }
            const long long rank = taken + before + __popcll(bal & ((2ull << lane) - 1)); // 1-based among the real candidates
            if (key != SC_KEY_EMPTY && rank <= need && key < best_key) { best_key = key; best_idx = k; }
            taken += total;
        }
    }
    sched_block_min_key<NW>(best_key, best_idx, s_key, s_idx, lane, wib);
    if (threadIdx.x == 0) sched_write_result_n(a, r, best_key, best_idx, n_scored);
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void sched_finish_kernel(SchedEvalArgs a, bool gen)
{
    __shared__ SchedFinishShared<NW> sh;
    sched_finish_block<NW>(a, gen, (int)blockIdx.x, sh);
}

// Pass 3: minimum (key, index) over the per-block minima of pass 2.
__global__ __launch_bounds__(256) void sched_reduce_kernel(SchedEvalArgs a)
{
    __shared__ u64 s_key[4];
    __shared__ long long s_idx[4];
    const int r = blockIdx.x;
    const u64 *pk = a.part_key + (long long)r * a.blocks_per_restart;
    const long long *pi = a.part_idx + (long long)r * a.blocks_per_restart;
    u64 bk = SC_KEY_EMPTY;
    long long bi = -1;
    for (int p = threadIdx.x; p < a.blocks_per_restart; p += 256) {
        const u64 key = pk[p];
        const long long idx = pi[p];
        if (key < bk || (key == bk && key != SC_KEY_EMPTY && idx < bi)) { bk = key; bi = idx; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned klo = __shfl_xor((unsigned)bk, m, 64), khi = __shfl_xor((unsigned)(bk >> 32), m, 64);
        const unsigned ilo = __shfl_xor((unsigned)bi, m, 64), ihi = __shfl_xor((unsigned)((u64)bi >> 32), m, 64);
        const u64 ok = ((u64)khi << 32) | klo;
        const long long oi = (long long)(((u64)ihi << 32) | ilo);
        if (ok < bk || (ok == bk && ok != SC_KEY_EMPTY && oi < bi)) { bk = ok; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_key[threadIdx.x >> 6] = bk; s_idx[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++)
            if (s_key[i] < bk || (s_key[i] == bk && bk != SC_KEY_EMPTY && s_idx[i] < bi)) { bk = s_key[i]; bi = s_idx[i]; }
        sched_write_result(a, r, bk, bi);
    }
}

// ScheduleMoveProposer::iter_local_moves (examples/employee-scheduling/src/lib.rs:493-559) as a move list: for every
// STORED day d, ChangeDay(d, e) for e = next(cur), next(next(cur)), ... in id order with wrap-around, the day's
// current employee last (:541-543; a no-op the tabu filter drops).  Employees are the indices 0..E-1 (sorted ids,
// BTreeSet :83), so next(e) = (e + 1) mod E.   moves[r][d * E + i] = ChangeDay(d, (state[r][d] + 1 + i) mod E).
__global__ __launch_bounds__(256) void sched_exhaustive_moves_kernel(const int *states, int S, int E, int n_restarts,
                                                                     ls_sched_move *moves, long long *n_moves)
{
    const long long per = (long long)S * E, total = per * n_restarts;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int r = (int)(t / per);
        const long long k = t - (long long)r * per;
        const int d = (int)(k / E), i = (int)(k - (long long)d * E);
        const int cur = states[(long long)r * S + d];
        ls_sched_move m;
        m.type = 0; m.a = d; m.b = (int)(((long long)cur + 1 + i) % E);
        moves[t] = m;
        if (k == 0 && n_moves) n_moves[r] = per;
    }
}

// ---------------------------------------------------------------------------------------------
// The resident scheduling GRID: ls_sched_eval_resident for lists beyond one workgroup (config 5: window 65 536 over a stream
// of ~82 000 moves) without the two launches per LS iteration of the pipeline above.  One never-relaunched kernel, one
// workgroup per 512-move chunk of the resident stream (each lane keeps its move in registers).  A request is ONE message of
// four tagged lines (srv_common.hpp): header (flags, n_moves, window) + the state, one entry per dword -- the workgroups keep
// no state, so there is nothing to apply or re-upload.  Workgroup 0 owns the host mailbox: it copies the four lines, tags and
// all, to a device-side "go" record, which every workgroup polls in ONE 64-lane load and accepts when all tags agree (the same
// torn-snapshot rule as over PCIe: no fence, no cache write-back between the XCDs' L2s -- the first version released and
// acquired at agent scope and an LS iteration took 60 us).  Every workgroup scores its chunk (sched_eval_all_kernel's body) and
// writes ONE record -- eight dwords, each tagged with the request's tag: count, minimum key, its index -- with a single
// write-through store instruction; workgroup 0, the finisher, polls the records until all carry the tag (no acknowledgement
// wait and no ticket in the workers), takes the prefix over the counts and the minima of the chunks inside the window, and
// scores the chunk the window cuts AGAIN up to the window (cheaper than 512 key stores in every workgroup).
// All workgroups must be resident at once: the launcher asks for at most ONE per CU (the kernel needs 180-200 VGPRs).
constexpr int SCG_STATE_AT = 16; // dword of the message that holds state[0]: the state fits lines 1..3 (S <= 48)

struct SchedGridCtl {        // device memory
    unsigned go[64];         // the current request, as the host wrote it (tagged dwords)
    unsigned pad[2];
    unsigned long long ready; // workgroups that have started, over all launches of this server
};

struct SchedGridArgs {
    SrvLink link;            // lines 0..3 of the mailbox = one message
    SchedInst in;
    const ls_sched_move *moves;
    long long n_resident;
    SchedGridCtl *g;
    unsigned long long ready_target; // g->ready once every workgroup of THIS launch has started (the host adds the grid size per launch)
    unsigned *rec;           // [n_chunks][8] every workgroup's record of the current request (tagged dwords)
    int emp_entries;         // entries of the workgroup's employee -> scored-days table in dynamic LDS (0: none, the scorers compare day by day)
    int poll_delay;          // the finisher sleeps this many x 64 cycles before its first look at the records; < 0: adapts (starts at 6)
    int cpw;                 // 512-candidate chunks per workgroup (consecutive ones): 1 for a resident stream, up to four for the exhaustive proposer
};

// GEN: the candidates are those of the exhaustive proposer (ScheduleMoveProposer::iter_local_moves, lib.rs:493-559), generated in
// place from the request's state -- no move list; a workgroup then takes a.cpw <= MAXC CONSECUTIVE chunks (32 x 10 000 candidates
// are 625 chunks: three per workgroup on 209 CUs), still one record per workgroup.
template <typename M, int NW, bool GEN, int MAXC>
__global__ __launch_bounds__(64 * NW) void sched_grid_server_kernel(SchedGridArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = (int)blockIdx.x, nch = (int)gridDim.x;
    const int S = a.in.S, D = a.in.D, E = a.in.n_employees;
    __shared__ SchedBaseTab<M> T;
    __shared__ u64 s_key[NW];
    __shared__ long long s_idx[NW];
    __shared__ int s_cnt[NW];
    __shared__ int s_cntc[MAXC][NW]; // real candidates per chunk and wavefront
    __shared__ int s_quit;           // the finisher: the launch is leaving, the records will not come
    __shared__ unsigned s_req[64];
    __shared__ int s_ctl[2]; // [0] leave, [1] the request's tag
    __shared__ int s_cut;
    __shared__ long long s_before;
    // Which scored days an employee works in the current state, for every employee: the table's day-equality rows and the day
    // set of a ChangeDay's new employee are one LDS read each instead of a 64-step compare loop (a third of the scorer's
    // instructions).  Zero between requests: the <= 64 entries a state touches are set before and cleared after its scoring.
    extern __shared__ __align__(16) unsigned char s_dyn[];
    M *const emp = a.emp_entries > 0 ? reinterpret_cast<M *>(s_dyn) : nullptr;
    for (int i = tid; i < a.emp_entries; i += 64 * NW) emp[i] = 0;
    SchedLaneConst L{};
    if (wib < 3) L = a.in.lane_tab[lane];
    // (Keeping the instance's holiday words in LDS -- 80 KB at 10 000 employees -- instead of gathering them from L2 per request was
    // tried: no change, 5.5 us from request to record either way; the workgroup's time is the delta scorer's dependent chain.)
    const SchedInst &in = a.in;
    const int cpw = a.cpw;
    const long long k0 = (long long)c * cpw * (64 * NW) + tid; // this lane's candidates: k0 + 512 j, j < cpw
    int mtype[MAXC], mma[MAXC], mmb[MAXC]; // resident stream: this lane's moves, loaded once
#pragma unroll
    for (int j = 0; j < MAXC; j++) {
        mtype[j] = -1; mma[j] = 0; mmb[j] = 0;
        if constexpr (!GEN) {
            const long long k = k0 + (long long)j * (64 * NW);
            if (j < cpw && k < a.n_resident) { const ls_sched_move m = a.moves[k]; mtype[j] = m.type; mma[j] = m.a; mmb[j] = m.b; }
        }
    }
    unsigned last_seq = 0;
    int pd = a.poll_delay < 0 ? 6 : a.poll_delay, calm = 0; // the finisher's wait before its first look at the records (x 64 cycles), requests without a second pass
    bool abort_launch = false; // workgroup 0: the workgroups of this launch are not all resident -- nothing is served, everybody leaves
    SrvPoll poll{0, 0, 0};
#ifdef LS_NQ_STAMPS
    unsigned long long gst[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // workgroup 0: request seen, go record out, edit + holiday bits in, table built, chunk scored, own chunk recorded, all records in, cut chunk scored, answer
#define SCG_STAMP(I) do { gst[I] = wall_clock64(); } while (0)
#else
#define SCG_STAMP(I) do { } while (0)
#endif
    if (wib == 0) {
        last_seq = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&a.g->go[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 16);
        if (c == 0) srv_poll_begin(poll, a.link, a.link.resp, lane);
        if (lane == 0) __hip_atomic_fetch_add(&a.g->ready, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c == 0) { // the master serves only once every workgroup of THIS launch polls the go record
            const unsigned long long t0 = wall_clock64();
            bool all = false;
            for (;;) {
                all = __hip_atomic_load(&a.g->ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.ready_target;
                if (all || wall_clock64() - t0 > a.link.idle_ticks) break;
                __builtin_amdgcn_s_sleep(8);
            }
            // they cannot all be resident: everybody leaves -- the others when their poll sees the quit word, this workgroup
            // WITHOUT looking at the mailbox (srv_poll_wait hands out a pending request before it looks at the quit word, and a
            // request served without the other workgroups would never see their records)
            abort_launch = !all;
            if (!all && lane == 0)
                __hip_atomic_store(&a.link.ctl->quit_gen, a.link.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    for (;;) {
        if (wib == 0) {
            int leave = 0;
            if (c == 0) { // the master: host mailbox -> go record, verbatim
                unsigned raw = 0;
                leave = abort_launch ? 1 : srv_poll_wait<64>(poll, a.link, a.link.req, s_req, lane, &raw);
                SCG_STAMP(0);
                if (!leave) {
                    __hip_atomic_store(&a.g->go[lane], raw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    SCG_STAMP(1);
                } else if (lane == 0) __hip_atomic_store(&a.link.ctl->quit_gen, a.link.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // tell the others
                last_seq = poll.last_tag;
            } else {
                for (;;) { // one 64-lane snapshot of the go record per poll; valid when all tags agree and the tag is new
                    const unsigned v = __hip_atomic_load(&a.g->go[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned t0 = __builtin_amdgcn_readfirstlane(v >> 16);
                    if (t0 != last_seq && t0 != 0 && __builtin_amdgcn_ballot_w64((v >> 16) != t0) == 0) {
                        s_req[lane] = v & 0xffffu;
                        last_seq = t0;
                        SCG_STAMP(0);
                        break;
                    }
                    unsigned long long q = 0;
                    if (lane == 0) q = __hip_atomic_load(&a.link.ctl->quit_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__builtin_amdgcn_readfirstlane((int)(q == a.link.gen))) { leave = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (lane == 0) { s_ctl[0] = leave; s_ctl[1] = (int)last_seq; }
        }
        __syncthreads();
        if (s_ctl[0]) break;
        const unsigned last_seq_all = (unsigned)s_ctl[1]; // the request's tag, in every wavefront
        long long nm = (long long)(s_req[1] | (s_req[2] << 16));
        nm = nm < a.n_resident ? nm : a.n_resident;
        const long long window = (long long)(s_req[3] | (s_req[4] << 16));
        const int fin = (int)s_req[5] < nch ? (int)s_req[5] : 0; // the finisher of THIS request: the chunk the host expects the window to cut
        const int e_lane = lane < S ? (int)s_req[SCG_STATE_AT + lane] : -1 - lane;
        const int hol_base = wib == 0 ? sched_holiday_bit(in, e_lane, lane) : 0;
        // ---- this workgroup's chunks (sched_eval_all_kernel with one batch per wavefront and chunk, keys not recorded)
        SchedEdit ed[MAXC];
        int hbits[MAXC];
#pragma unroll
        for (int j = 0; j < MAXC; j++) {
            const long long k = k0 + (long long)j * (64 * NW);
            int ty = -1, xa = 0, xb = 0;
            if (j < cpw) { // block-uniform
                if constexpr (GEN) sched_gen_exhaustive(e_lane, E, k, ty, xa, xb); // every lane takes part in the shuffle
                else { ty = mtype[j]; xa = mma[j]; xb = mmb[j]; }
                if (k >= nm) ty = -1;
            }
            ed[j] = sched_edit_of(e_lane, S, E, ty, xa, xb);
            hbits[j] = sched_edit_hbits(in, ed[j]);
        }
#ifdef LS_NQ_STAMPS
        { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); gst[2] = wall_clock64(); }
#endif
        if (wib < 3) sched_build_base_tab<M>(in, wib, e_lane, hol_base, L, lane, T, emp);
        __syncthreads();
        SCG_STAMP(3);
        u64 best_key = SC_KEY_EMPTY;
        long long best_idx = -1;
        u64 my_key[MAXC], my_bal[MAXC]; // this lane's own candidates: what the finisher needs if the window cuts ITS chunks
#pragma unroll
        for (int j = 0; j < MAXC; j++) {
            my_key[j] = SC_KEY_EMPTY;
            if (ed[j].active) {
                int hard, soft;
                my_key[j] = sched_delta_key<M>(T, D, e_lane, ed[j], hbits[j], hard, soft, emp);
                if (my_key[j] < best_key) { best_key = my_key[j]; best_idx = k0 + (long long)j * (64 * NW); }
            }
            my_bal[j] = __ballot(ed[j].active);
            if (lane == 0) s_cntc[j][wib] = __popcll(my_bal[j]);
        }
        SCG_STAMP(4);
        sched_block_min_key<NW>(best_key, best_idx, s_key, s_idx, lane, wib); // synchronises the block; valid in wavefront 0
        int my_rank[MAXC], total = 0; // 1-based among the workgroup's real candidates, in candidate order (chunk, wavefront, lane)
#pragma unroll
        for (int j = 0; j < MAXC; j++) {
            my_rank[j] = total + __popcll(my_bal[j] & ((2ull << lane) - 1));
            for (int i = 0; i < NW; i++) {
                if (i < wib) my_rank[j] += s_cntc[j][i];
                total += s_cntc[j][i];
            }
        }
        // ---- this chunk's record: eight tagged dwords (count | key in four 16-bit pieces | index in two), ONE store by eight lanes,
        // no acknowledgement wait and no ticket: the request's finisher polls the records until all carry the tag
        if (wib == 0) {
            unsigned pay = 0;
            if (lane == 0) pay = (unsigned)total; // <= 512 MAXC
            else if (lane <= 4) pay = (unsigned)(best_key >> (16 * (lane - 1))) & 0xffffu;
            else if (lane == 5) pay = (unsigned)best_idx & 0xffffu;
            else if (lane == 6) pay = (unsigned)(best_idx >> 16) & 0xffffu;
            if (lane < 8) __hip_atomic_store(a.rec + (size_t)c * 8 + lane, (last_seq << 16) | pay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) { s_cut = -1; s_before = 0; }
        __syncthreads();
        SCG_STAMP(5);
        if (c == fin) { // ---- the finish step: prefix over the counts, minima of the chunks inside the window, the cut chunk up to the window
            u64 fkey = SC_KEY_EMPTY;
            long long fidx = -1, run = 0;
            // every workgroup's record of THIS request (nch <= 64 NW: one record per thread)
            unsigned rv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            bool given_up = false;
            // A pass over the records is an L2-bypassing read of ~1.2 us; one that starts a moment too early misses the last records
            // and a second full pass follows, so the answer wait depends on WHEN the first pass starts: measured on config 5 with fixed
            // waits of 0 / 512 / 1024 cycles before it: 11.0 / 9.85 / 10.7 us on one build, 10.4 / 10.3 us on the next (a few
            // instructions moved the edge).  The finisher keeps its wait just above the point where second passes appear:
            // +2 x 64 cycles after a request that needed one, -1 after 64 requests that did not.
            for (int w = 0; w < pd; w++) __builtin_amdgcn_s_sleep(1);
            int passes = 0;
            for (;;) {
                passes++;
                bool ok = true;
                if (tid < nch) {
#pragma unroll
                    for (int d = 0; d < 8; d++) rv[d] = __hip_atomic_load(a.rec + (size_t)tid * 8 + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int d = 0; d < 7; d++) ok = ok && (rv[d] >> 16) == last_seq_all;
                }
                // the exit every poll loop of a resident kernel needs: records that never come (the launch is leaving)
                if (tid == 0) s_quit = __hip_atomic_load(&a.link.ctl->quit_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.link.gen;
                if (__syncthreads_and(ok ? 1 : 0)) break;
                if (s_quit) { given_up = true; break; } // block-uniform: written before the barrier above
                __syncthreads(); // ... and read by everybody before thread 0 writes it again
            }
            if (given_up) continue; // no answer: the host's wait finds the launch drained
            if (a.poll_delay < 0) { // block-uniform
                if (passes > 1) { pd = pd + 2 < 32 ? pd + 2 : 32; calm = 0; }
                else if (++calm >= 64) { pd = pd > 0 ? pd - 1 : 0; calm = 0; }
            }
            SCG_STAMP(6);
            if (window > 0) {
                const int b = tid;
                const int cb = b < nch ? (int)(rv[0] & 0xffffu) : 0;
                const u64 pkb = b < nch ? ((u64)(rv[1] & 0xffffu) | ((u64)(rv[2] & 0xffffu) << 16) | ((u64)(rv[3] & 0xffffu) << 32) | ((u64)(rv[4] & 0xffffu) << 48)) : SC_KEY_EMPTY;
                const long long pib = b < nch ? (long long)((rv[5] & 0xffffu) | ((rv[6] & 0xffffu) << 16)) : -1;
                int incl = cb;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += o;
                }
                if (lane == 63) s_cnt[wib] = incl;
                __syncthreads();
                int before = 0, total = 0;
                for (int i = 0; i < NW; i++) {
                    if (i < wib) before += s_cnt[i];
                    total += s_cnt[i];
                }
                const long long through = before + incl; // real candidates up to and including chunk b
                if (b < nch && cb > 0) {
                    if (through <= window) sched_min_key(fkey, fidx, pkb, pib);
                    else if (through - cb < window) { s_cut = b; s_before = through - cb; }
                }
                __syncthreads();
                run = total;
            }
            const long long n_scored = window <= 0 ? 0 : (run < window ? run : window);
            const int cut = s_cut;
            if (cut == c) { // the window cuts the finisher's own chunks, as the host expected: the keys are still in the lanes
                const long long need = window - s_before;
#pragma unroll
                for (int j = 0; j < MAXC; j++)
                    if (ed[j].active && my_rank[j] <= need && my_key[j] < fkey) { fkey = my_key[j]; fidx = k0 + (long long)j * (64 * NW); }
            } else if (cut >= 0) { // another workgroup's: scored again here (same state, same table), up to the window
                const long long need = window - s_before;
                int done = 0; // real candidates of the chunks already walked
                for (int j = 0; j < cpw && done < need; j++) { // block-uniform
                    const long long kk = ((long long)cut * cpw + j) * (64 * NW) + tid;
                    int ty = -1, xa = 0, xb = 0;
                    if constexpr (GEN) sched_gen_exhaustive(e_lane, E, kk, ty, xa, xb);
                    else if (kk < nm) { const ls_sched_move m = a.moves[kk]; ty = m.type; xa = m.a; xb = m.b; }
                    if (kk >= nm) ty = -1;
                    const SchedEdit e2 = sched_edit_of(e_lane, S, E, ty, xa, xb);
                    const int hb2 = sched_edit_hbits(in, e2);
                    const u64 bal = __ballot(e2.active);
                    __syncthreads(); // s_cnt is reused
                    if (lane == 0) s_cnt[wib] = __popcll(bal);
                    __syncthreads();
                    int before = done;
                    for (int i = 0; i < NW; i++) {
                        if (i < wib) before += s_cnt[i];
                        done += s_cnt[i];
                    }
                    const long long rank = before + __popcll(bal & ((2ull << lane) - 1)); // 1-based among the workgroup's real candidates
                    if (e2.active && rank <= need) {
                        int hard, soft;
                        const u64 key2 = sched_delta_key<M>(T, D, e_lane, e2, hb2, hard, soft, emp);
                        if (key2 < fkey) { fkey = key2; fidx = kk; }
                    }
                }
            }
            SCG_STAMP(7);
            sched_block_min_key<NW>(fkey, fidx, s_key, s_idx, lane, wib);
            if (wib == 0) { // the answer: d0 has-move, d1-2 n_scored, d3-4 best_index, d5 hard, d6 soft, d7 the cut chunk + 1 (0: none)
                const bool has = fkey != SC_KEY_EMPTY;
                unsigned pay = 0;
                if (lane == 0) pay = has ? 1u : 0u;
                else if (lane == 1) pay = has ? (unsigned)n_scored & 0xffffu : 0u;
                else if (lane == 2) pay = has ? (unsigned)(n_scored >> 16) : 0u;
                else if (lane == 3) pay = has ? (unsigned)fidx & 0xffffu : 0u;
                else if (lane == 4) pay = has ? (unsigned)(fidx >> 16) : 0u;
                else if (lane == 5) pay = has ? (unsigned)(fkey >> 54) : 0u;
                else if (lane == 6) pay = has ? (unsigned)((fkey >> 46) & 0xffu) : 0u;
                else if (lane == 7) pay = (unsigned)(cut + 1);
#ifdef LS_NQ_STAMPS
                gst[8] = wall_clock64();
                if (lane >= 8 && lane < 16) { const unsigned long long dt = gst[lane - 7] - gst[0]; pay = dt > 0xffffu ? 0xffffu : (unsigned)dt; }
                if (lane >= 8 && lane < 16) __hip_atomic_store(a.link.resp + lane, (last_seq << 16) | (pay & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
                if (lane < 8) __hip_atomic_store(a.link.resp + lane, (last_seq << 16) | (pay & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (lane == 0) __hip_atomic_store(&a.link.ctl->last_activity, wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (emp && wib == 0 && lane < D && e_lane >= 0) emp[e_lane] = 0; // the table is all zero again (every scorer of this request is past a barrier)
        if (c == 0 && wib == 0) poll.t_served = wall_clock64();
    }
    if (tid == 0) srv_block_leaves(a.link);
}

// ------------------------------------------------------------------ launchers
static SchedInst make_inst(const ls_ctx *ctx)
{
    SchedInst in;
    in.D = ctx->sched.n_days_scored;
    in.S = ctx->sched.n_days_stored;
    in.n_employees = ctx->sched.n_employees;
    in.holiday_bits = ctx->sched.holiday_bits.as<u64>();
    in.lane_tab = ctx->sched.lane_tab.as<SchedLaneConst>();
    return in;
}

int sched_score_dev(ls_ctx *ctx, int n_states, const int *states_d, long long *hard_d, long long *soft_d,
                    long long *terms_d)
{
    long long blocks = ((long long)n_states + 3) / 4;
    const long long cap = (long long)ctx->n_cus * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sched_score_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, make_inst(ctx),
                       n_states, states_d, hard_d, soft_d, terms_d);
    LS_HIP(ctx, hipGetLastError());
    return LS_OK;
}

int sched_exhaustive_moves_dev(ls_ctx *ctx, int n_restarts, const int *states_d, ls_sched_move *moves_d, long long *n_moves_d)
{
    const long long total = (long long)n_restarts * ctx->sched.n_days_stored * ctx->sched.n_employees;
    long long g = (total + 255) / 256;
    const long long cap = (long long)ctx->n_cus * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(sched_exhaustive_moves_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, states_d,
                       ctx->sched.n_days_stored, ctx->sched.n_employees, n_restarts, moves_d, n_moves_d);
    LS_HIP(ctx, hipGetLastError());
    return LS_OK;
}

bool sched_is_fused_launch(const ls_ctx *ctx, long long max_moves)
{
    const long long n_chunks = max_moves > 0 ? (max_moves + SC_COUNT_CHUNK - 1) / SC_COUNT_CHUNK : 1;
    const long long batches = max_moves > 0 ? (max_moves + 63) / 64 : 1;
    return n_chunks == 1 && batches * 4 <= 2 * SC_FUSED_WAVES && ctx->sched_fused;
}

// the two-launch delta pipeline (sched_eval_all_kernel + sched_finish_kernel); a holds everything but the workspace
static int sched_launch_delta_pipeline(ls_ctx *ctx, SchedEvalArgs a, bool gen)
{
    constexpr int DW = 8;
    // Batches of 64 candidates per wavefront: one while every CU gets at most two blocks (a block lives ~5 us, mostly latency);
    // beyond that the dispatcher's ~2 ns per wavefront is the limit and each wavefront takes 2 or 3 batches behind one base
    // table (exhaustive 320 k candidates: 16.8 / 14.5 / 14.1 us for 1 / 2 / 3; window 65 536 of 82 k: 10.2 / 12.4 / 16.3).
    // LS_SCHED_DELTA_BPW pins it (measurement knob).
    const char *bpw_s = getenv("LS_SCHED_DELTA_BPW"); // read per launch
    const int bpw_env = bpw_s ? atoi(bpw_s) : 0;
    int bpw = bpw_env;
    if (bpw < 1)
        for (bpw = 1; bpw < 3; bpw++) {
            const long long cm = 64LL * DW * bpw;
            if ((a.max_moves + cm - 1) / cm * a.n_restarts <= 2LL * ctx->n_cus) break;
        }
    a.chunk_moves = 64LL * DW * bpw;
    const long long nch = a.max_moves > 0 ? (a.max_moves + a.chunk_moves - 1) / a.chunk_moves : 1;
    if (nch * a.n_restarts > 0x7fffffffLL) return set_err(ctx, LS_ERR_UNSUPPORTED, "max_moves = %lld is too large", a.max_moves);
    const size_t parts = (size_t)a.n_restarts * (size_t)nch;
    LS_RESERVE(ctx, ctx->sc_keys, parts * 20);
    LS_RESERVE(ctx, ctx->sc_cand_keys, (size_t)a.n_restarts * (size_t)(a.max_moves > 0 ? a.max_moves : 1) * 8);
    a.n_chunks = (int)nch;
    a.blocks_per_restart = (int)nch;
    a.split = 1;
    a.part_key = ctx->sc_keys.as<u64>();
    a.part_idx = reinterpret_cast<long long *>(a.part_key + parts);
    a.chunk_counts = reinterpret_cast<int *>(a.part_idx + parts);
    a.cand_keys = ctx->sc_cand_keys.as<u64>();
    a.k_end = nullptr;
    a.n_scored = nullptr;
    const bool m32 = ctx->sched.n_days_stored <= 32 && !ctx->sched_force_64;
    const dim3 g1((unsigned)(nch * a.n_restarts)), g2((unsigned)a.n_restarts), b(64 * DW);
    // the employee -> scored-days table of a block: when three blocks of a CU can still hold one each
    size_t dyn = ((size_t)ctx->sched.n_employees * (m32 ? 4 : 8) + 15) & ~(size_t)15;
    if (dyn > ((size_t)48 << 10) || getenv("LS_SCHED_NO_EMP_TABLE")) dyn = 0;
    a.emp_entries = dyn ? ctx->sched.n_employees : 0;
#define LS_LAUNCH2(MT, GEN)                                                                             \
    {                                                                                                   \
        hipLaunchKernelGGL((sched_eval_all_kernel<MT, DW, GEN>), g1, b, dyn, ctx->stream, a);           \
        LS_HIP(ctx, hipGetLastError());                                                                 \
    }
    const bool one_launch = ctx->sched_one_launch;
    a.ticket = nullptr;
    if (one_launch) {
        const size_t before = ctx->sc_ticket.cap;
        LS_RESERVE(ctx, ctx->sc_ticket, (size_t)a.n_restarts * 4);
        if (ctx->sc_ticket.cap != before) { // a fresh buffer: zero it in stream order; every launch leaves it zero again
            const long long words = (long long)(ctx->sc_ticket.cap / 4);
            hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->sc_ticket.as<int>(), 0, words);
            LS_HIP(ctx, hipGetLastError());
        }
        a.ticket = ctx->sc_ticket.as<unsigned>();
    }
    if (m32) { if (gen) LS_LAUNCH2(unsigned, true) else LS_LAUNCH2(unsigned, false) }
    else { if (gen) LS_LAUNCH2(u64, true) else LS_LAUNCH2(u64, false) }
    if (!one_launch) {
        hipLaunchKernelGGL((sched_finish_kernel<DW>), g2, b, 0, ctx->stream, a, gen);
        LS_HIP(ctx, hipGetLastError());
    }
#undef LS_LAUNCH2
    return LS_OK;
}

// the exhaustive proposer's neighbourhood (lib.rs:493-559) of n_restarts states, candidates generated in place
static void sched_set_inline(const ls_ctx *ctx, SchedEvalArgs &a, const int *inline_state_h, long long inline_n_moves)
{
    a.inline_valid = 0;
    a.inline_n_moves = 0;
    if (inline_state_h && a.n_restarts == 1) { // one restart: the state (<= 256 bytes) and n_moves travel in the kernel arguments
        a.inline_valid = 1;
        a.inline_n_moves = inline_n_moves;
        for (int d = 0; d < 64; d++) a.inline_state[d] = d < ctx->sched.n_days_stored ? inline_state_h[d] : 0;
    }
}

int sched_eval_exhaustive_dev(ls_ctx *ctx, int n_restarts, const int *states_d, long long window, ls_sched_result *results_d,
                              bool *handled, unsigned long long *done_flag, unsigned long long done_seq, const int *inline_state_h)
{
    *handled = false;
    if (!ctx->sched_delta) return LS_OK; // the caller builds the move list and runs the list pipeline
    if (ctx->sched.n_employees > 32768)
        return set_err(ctx, LS_ERR_UNSUPPORTED, "n_employees = %d exceeds the 15-bit argmin key field (32768)",
                       ctx->sched.n_employees);
    SchedEvalArgs a{};
    a.in = make_inst(ctx);
    a.n_restarts = n_restarts;
    a.states = states_d;
    a.moves = nullptr;
    a.n_moves = nullptr;
    a.max_moves = (long long)ctx->sched.n_days_stored * ctx->sched.n_employees;
    a.window = window;
    a.hard_out = a.soft_out = nullptr;
    a.results = results_d;
    a.done_flag = n_restarts == 1 ? done_flag : nullptr;
    a.done_seq = done_seq;
    sched_set_inline(ctx, a, inline_state_h, a.max_moves);
    *handled = true;
    return sched_launch_delta_pipeline(ctx, a, true);
}

int sched_eval_batch_dev(ls_ctx *ctx, int n_restarts, const int *states_d, const ls_sched_move *moves_d,
                         const long long *n_moves_d, long long max_moves, long long window,
                         ls_sched_result *results_d, long long *hard_out_d, long long *soft_out_d,
                         unsigned long long *done_flag, unsigned long long done_seq, const int *inline_state_h,
                         long long inline_n_moves)
{
    if (ctx->sched.n_employees > 32768)
        return set_err(ctx, LS_ERR_UNSUPPORTED, "n_employees = %d exceeds the 15-bit argmin key field (32768)",
                       ctx->sched.n_employees);
    const long long n_chunks = max_moves > 0 ? (max_moves + SC_COUNT_CHUNK - 1) / SC_COUNT_CHUNK : 1;
    if (n_chunks > 65535 * 16)
        return set_err(ctx, LS_ERR_UNSUPPORTED, "max_moves = %lld is too large", max_moves);
    // one wavefront per batch of 64 candidates, never more blocks than the chip holds at once
    const long long batches = max_moves > 0 ? (max_moves + 63) / 64 : 1;
    const bool delta = ctx->sched_delta;
    const bool m32 = ctx->sched.n_days_stored <= 32 && !ctx->sched_force_64;
    constexpr int DW = 8; // wavefronts per block of the delta kernel
    const int split = delta ? 1 : (batches * n_restarts >= 4096 ? 1 : 4); // few batches: spread each over four wavefronts
    static const int delta_batches_per_wave = getenv("LS_SCHED_DELTA_BPW") ? atoi(getenv("LS_SCHED_DELTA_BPW")) : 1; // measurement knob
    long long bpr = delta ? (batches + (long long)DW * delta_batches_per_wave - 1) / ((long long)DW * delta_batches_per_wave)
                          : (batches * split + 3) / 4;
    static const int blocks_per_cu = getenv("LS_SCHED_BLOCKS_PER_CU") ? atoi(getenv("LS_SCHED_BLOCKS_PER_CU")) : 8; // measurement knob
    const long long cap = ((long long)ctx->n_cus * blocks_per_cu) / n_restarts;
    if (bpr > cap) bpr = cap;
    if (bpr < 1) bpr = 1;
    // workspace: k_end[R] | n_scored[R] | part_key[R][bpr] | part_idx[R][bpr] | chunk_counts[R][n_chunks]
    const size_t parts = (size_t)n_restarts * (size_t)bpr;
    LS_RESERVE(ctx, ctx->sc_keys, (size_t)n_restarts * 16 + parts * 16 + (size_t)n_restarts * n_chunks * 4);
    SchedEvalArgs a{};
    a.in = make_inst(ctx);
    a.n_restarts = n_restarts;
    a.states = states_d; a.moves = moves_d; a.n_moves = n_moves_d;
    a.max_moves = max_moves; a.window = window;
    a.k_end = ctx->sc_keys.as<long long>();
    a.n_scored = a.k_end + n_restarts;
    a.blocks_per_restart = (int)bpr;
    a.split = split;
    a.part_key = reinterpret_cast<u64 *>(a.n_scored + n_restarts);
    a.part_idx = reinterpret_cast<long long *>(a.part_key + parts);
    a.n_chunks = (int)n_chunks;
    a.chunk_counts = reinterpret_cast<int *>(a.part_idx + parts);
    a.hard_out = hard_out_d; a.soft_out = soft_out_d;
    a.results = results_d;
    a.done_flag = n_restarts == 1 ? done_flag : nullptr; // exactly one thread of one kernel writes the record of a one-restart launch
    a.done_seq = done_seq;
    a.cand_keys = nullptr;
    a.chunk_moves = 0;
    sched_set_inline(ctx, a, inline_state_h, inline_n_moves);
    if (delta && !hard_out_d && !soft_out_d && !(n_chunks == 1 && batches * 4 <= 2 * SC_FUSED_WAVES && ctx->sched_fused))
        return sched_launch_delta_pipeline(ctx, a, false);
    const long long total = (long long)n_restarts * max_moves;
    const unsigned fill_grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (hard_out_d && total > 0) {
        hipLaunchKernelGGL(fill_i64_kernel, dim3(fill_grid), dim3(256), 0, ctx->stream, hard_out_d, (long long)INT64_MIN, total);
        LS_HIP(ctx, hipGetLastError());
    }
    if (soft_out_d && total > 0) {
        hipLaunchKernelGGL(fill_i64_kernel, dim3(fill_grid), dim3(256), 0, ctx->stream, soft_out_d, (long long)INT64_MIN, total);
        LS_HIP(ctx, hipGetLastError());
    }
    if (n_chunks == 1 && batches * 4 <= 2 * SC_FUSED_WAVES && ctx->sched_fused) {
        // small launch (the reference's example: window 100): locate + eval + reduce in one block per restart
        a.split = 4;
        a.blocks_per_restart = 1;
        if (delta) {
            if (m32) hipLaunchKernelGGL((sched_eval_small_kernel<unsigned, DW>), dim3(n_restarts), dim3(64 * DW), 0, ctx->stream, a);
            else hipLaunchKernelGGL((sched_eval_small_kernel<u64, DW>), dim3(n_restarts), dim3(64 * DW), 0, ctx->stream, a);
        } else if (ctx->sched.n_days_stored <= 32 && !ctx->sched_force_64)
            hipLaunchKernelGGL((sched_eval32_kernel<SC_FUSED_WAVES, true>), dim3(n_restarts), dim3(64 * SC_FUSED_WAVES), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL((sched_eval_kernel<SC_FUSED_WAVES, true>), dim3(n_restarts), dim3(64 * SC_FUSED_WAVES), 0, ctx->stream, a);
        LS_HIP(ctx, hipGetLastError());
        return LS_OK;
    }
    if (n_chunks > 1) { // a single chunk is counted by the locate pass itself
        hipLaunchKernelGGL(sched_count_kernel, dim3((unsigned)n_chunks, (unsigned)n_restarts), dim3(256), 0, ctx->stream, a);
        LS_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(sched_locate_kernel, dim3(n_restarts), dim3(256), 0, ctx->stream, a);
    LS_HIP(ctx, hipGetLastError());
    if (delta) {
        if (m32) hipLaunchKernelGGL((sched_eval_delta_kernel<unsigned, DW>), dim3((unsigned)(bpr * n_restarts)), dim3(64 * DW), 0, ctx->stream, a);
        else hipLaunchKernelGGL((sched_eval_delta_kernel<u64, DW>), dim3((unsigned)(bpr * n_restarts)), dim3(64 * DW), 0, ctx->stream, a);
    } else if (ctx->sched.n_days_stored <= 32 && !ctx->sched_force_64)
        hipLaunchKernelGGL((sched_eval32_kernel<4, false>), dim3((unsigned)(bpr * n_restarts)), dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL((sched_eval_kernel<4, false>), dim3((unsigned)(bpr * n_restarts)), dim3(256), 0, ctx->stream, a);
    LS_HIP(ctx, hipGetLastError());
    if (bpr > 1) { // with one block per restart pass 2 has already written the result
        hipLaunchKernelGGL(sched_reduce_kernel, dim3(n_restarts), dim3(256), 0, ctx->stream, a);
        LS_HIP(ctx, hipGetLastError());
    }
    return LS_OK;
}

// ------------------------------------------------------------------------------------------ resident scheduling LS step
// Dynamic LDS for a resident workgroup's employee -> scored-days table (one mask per employee), or 0 when it does not fit
// beside the static LDS of the kernel (160 KB per CU on gfx950) or is switched off (LS_SCHED_NO_EMP_TABLE=1: the scorers then
// compare day by day, as the launched kernels do).
static size_t sched_emp_table_bytes(const ls_ctx *ctx, const void *fn, bool m32)
{
    const bool no_table = getenv("LS_SCHED_NO_EMP_TABLE") != nullptr; // read per launch
    const size_t dyn = (size_t)ctx->sched.n_employees * (m32 ? 4 : 8);
    if (no_table || dyn == 0 || dyn > ((size_t)96 << 10)) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return dyn;
}

constexpr int SCS_NW = 8; // wavefronts of the server workgroup: move lists of at most 512 entries

struct SchedServer {
    SrvHost h;               // the one-workgroup server (lists of <= 512 moves, single-state scores)
    unsigned state_id = 0;
    bool valid = false;
    int *shadow() const { return reinterpret_cast<int *>(h.extra()); }
    // the resident grid (longer lists): created on demand; mailbox = 5 lines (header + 4 state lines)
    SrvHost grid;
    bool grid_ready = false;
    SchedGridCtl *g = nullptr;
    DevBuf ws;               // the chunk records [n_chunks][8]
    unsigned long long ready_total = 0;
    unsigned long long grid_requests = 0;
    double t_wait = 0, t_call = 0;               // LS_AMD_TRACE
    unsigned long long dev_ticks[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // stamps builds: the finisher's timeline
    bool grid_unavailable = false;        // the grid's workgroups could not all be resident (another context's grid holds the CUs): not tried again
    bool grid_gen = false;                // what the resident grid serves: the resident stream | the exhaustive proposer (one at a time)
    int grid_fin = 0;                     // the chunk the window cut in the last answer: the next request's finisher
    unsigned long long grid_fin_hits = 0; // requests whose window cut the finisher's own chunk
};

bool sched_server_eligible(const ls_ctx *ctx, long long n_moves)
{
    return ctx->sched_server_on && ctx->sched_delta && ctx->sched.valid && n_moves <= 64 * SCS_NW && ctx->sched.n_employees <= 32768 &&
           ctx->sc_resident_n > 0 && (long long)ctx->sc_resident_host.size() == ctx->sc_resident_n;
}

void sched_server_counters(const ls_ctx *ctx, unsigned long long out[3])
{
    out[0] = out[1] = out[2] = 0;
    if (const SchedServer *s = ctx->sched_server) {
        out[0] = s->h.requests.load() + s->grid_requests;
        out[1] = s->h.relaunches.load() + s->grid.relaunches.load();
        out[2] = s->h.slow_waits.load() + s->grid.slow_waits.load();
    }
}

// the launch leaves (its arguments -- instance, resident stream -- are about to change, or memory is about to be released)
void sched_server_pause(ls_ctx *ctx)
{
    SchedServer *s = ctx->sched_server;
    if (!s) return;
    s->valid = false; // whatever changes next (instance, stream) may change what the state means
    (void)hipSetDevice(ctx->device);
    for (SrvHost *h : {&s->h, &s->grid}) {
        if (!h->stream) continue;
        if (!h->drained()) {
            const unsigned tag = h->next_tag(0);
            volatile unsigned *dst = h->req(0);
            for (int d = 0; d < 16; d++) dst[d] = (tag << 16) | (d == 0 ? SRV_F_QUIT : 0u);
            h->flush_lines();
        }
        (void)hipStreamSynchronize(h->stream);
    }
}

void sched_server_destroy(ls_ctx *ctx)
{
    SchedServer *s = ctx->sched_server;
    if (!s) return;
    if (getenv("LS_AMD_TRACE") && s->grid_requests) {
        fprintf(stderr, "scheduling grid: %llu requests, answer wait %.2f us per request (whole call %.2f us), %llu launches, the window cut the finisher's chunk in %llu\n",
                s->grid_requests, s->t_wait / s->grid_requests * 1e6, s->t_call / s->grid_requests * 1e6, s->grid.relaunches.load(), s->grid_fin_hits);
        if (s->dev_ticks[4])
            fprintf(stderr, "  the finisher, after it saw the request: edit + holiday bits in %.2f us, table built %.2f, chunk scored %.2f, "
                            "own chunk recorded %.2f, all records in %.2f, cut chunk scored %.2f, answer %.2f\n",
                    s->dev_ticks[1] * 0.01 / s->grid_requests, s->dev_ticks[2] * 0.01 / s->grid_requests, s->dev_ticks[3] * 0.01 / s->grid_requests,
                    s->dev_ticks[4] * 0.01 / s->grid_requests, s->dev_ticks[5] * 0.01 / s->grid_requests, s->dev_ticks[6] * 0.01 / s->grid_requests,
                    s->dev_ticks[7] * 0.01 / s->grid_requests);
    }
    s->h.shutdown();
    s->grid.shutdown();
    if (s->g) (void)hipFree(s->g);
    s->ws.release();
    delete s;
    ctx->sched_server = nullptr;
}

// the state of this request versus what the workgroup holds: a new one is range-checked, copied to the pinned shadow, gets a new id
static int sched_server_sync_state(ls_ctx *ctx, SchedServer *s, const int32_t *state)
{
    const int S = ctx->sched.n_days_stored, E = ctx->sched.n_employees;
    int *sh = s->shadow();
    if (!s->valid || memcmp(sh, state, (size_t)S * 4) != 0) {
        for (int d = 0; d < S; d++)
            if (state[d] < 0 || state[d] >= E)
                return set_err(ctx, LS_ERR_INVALID, "state entry %d = %d is not an employee index in [0, %d)", d, state[d], E);
        memcpy(sh, state, (size_t)S * 4);
        s->valid = true;
        s->state_id = s->state_id == 0xffffffffu ? 1u : s->state_id + 1u;
    }
    return LS_OK;
}

// ls_sched_score for one state through the resident workgroup (only once it exists: the LS walk that follows then starts
// from a state the workgroup already holds)
bool sched_server_can_score(const ls_ctx *ctx) { return sched_server_eligible(ctx, 0); } // (needs the resident list: it is a launch argument)

static int sched_small_server(ls_ctx *ctx, SchedServer **out);

int sched_server_score(ls_ctx *ctx, const int32_t *state, int64_t *hard, int64_t *soft)
{
    SchedServer *s = nullptr;
    int rc = sched_small_server(ctx, &s);
    if (rc != LS_OK) return rc;
    rc = sched_server_sync_state(ctx, s, state);
    if (rc != LS_OK) return rc;
    unsigned pay[16] = {0};
    pay[0] = SCS_F_SCORE;
    pay[3] = s->state_id & 0xffffu; pay[4] = s->state_id >> 16;
    rc = s->h.ensure_running(ctx);
    if (rc != LS_OK) return rc;
    s->h.post(0, pay);
    unsigned r[8];
    rc = s->h.wait(ctx, 0, r);
    if (rc != LS_OK) { s->valid = false; return rc; }
    *hard = r[3];
    *soft = r[4];
    return LS_OK;
}

// the context's one-workgroup scheduling server, created on first use
static int sched_small_server(ls_ctx *ctx, SchedServer **out)
{
    SchedServer *s = ctx->sched_server;
    if (!s) {
        s = new (std::nothrow) SchedServer();
        if (!s) return set_err(ctx, LS_ERR_NOMEM, "out of host memory");
        ctx->sched_server = s;
    }
    if (!s->h.stream) {
        LS_HIP(ctx, hipSetDevice(ctx->device));
        const int rc = s->h.init(ctx, "scheduling LS-step server", 1, 256);
        if (rc != LS_OK) return rc;
        s->h.launch = [s, ctx](const SrvLink &L) {
            SchedSrvArgs a;
            a.link = L;
            a.in = make_inst(ctx);
            a.moves = ctx->sc_resident_moves.as<ls_sched_move>();
            a.n_resident = ctx->sc_resident_n;
            a.shadow = s->shadow();
            void *args[] = {&a};
            const bool m32 = ctx->sched.n_days_stored <= 32 && !ctx->sched_force_64;
            const void *fn = m32 ? reinterpret_cast<const void *>(&sched_server_kernel<unsigned, SCS_NW>)
                                 : reinterpret_cast<const void *>(&sched_server_kernel<u64, SCS_NW>);
            const size_t dyn = sched_emp_table_bytes(ctx, fn, m32);
            a.emp_entries = dyn ? ctx->sched.n_employees : 0;
            return hipLaunchKernel(fn, dim3(1), dim3(64 * SCS_NW), args, dyn, s->h.stream);
        };
    }
    *out = s;
    return LS_OK;
}

// ls_sched_eval_resident through the resident workgroup; `state` is range-checked here when it is not the one the
// workgroup holds.  n_moves <= 512 (sched_server_eligible).
int sched_server_eval(ls_ctx *ctx, const int32_t *state, long long n_moves, long long window, ls_sched_result *out)
{
    SchedServer *s = nullptr;
    {
        const int rc0 = sched_small_server(ctx, &s);
        if (rc0 != LS_OK) return rc0;
    }
    int rc = sched_server_sync_state(ctx, s, state);
    if (rc != LS_OK) return rc;
    int *sh = s->shadow();
    unsigned pay[16] = {0};
    pay[0] = SCS_F_EVAL;
    pay[1] = (unsigned)n_moves;
    pay[2] = (unsigned)(window > 0xffff ? 0xffff : window); // n_moves <= 512: a larger window changes nothing
    pay[3] = s->state_id & 0xffffu; pay[4] = s->state_id >> 16;
    rc = s->h.ensure_running(ctx);
    if (rc != LS_OK) return rc;
    s->h.post(0, pay);
    unsigned r[8];
    rc = s->h.wait(ctx, 0, r);
    if (rc != LS_OK) { s->valid = false; return rc; }
    if (!(r[0] & 1u)) {
        out->n_scored = 0; out->best_index = -1; out->best_hard = 0; out->best_soft = 0;
    } else {
        out->n_scored = r[1]; out->best_index = r[2]; out->best_hard = r[3]; out->best_soft = r[4];
        const ls_sched_move &m = ctx->sc_resident_host[(size_t)r[2]]; // the workgroup applied it to its state
        if (m.type == 0) sh[m.a] = m.b;
        else { const int ea = sh[m.a], eb = sh[m.b]; sh[m.a] = eb; sh[m.b] = ea; }
    }
    return LS_OK;
}

// ---- the resident grid behind ls_sched_eval_resident for lists beyond 512 moves
constexpr int SCG_MAXC = 4; // chunks per workgroup of the exhaustive grid, at most

// workgroups of the grid and the chunks each takes.  Stream: one per 512 moves of the resident list.  Exhaustive proposer
// (gen): S x E candidates, consecutive chunks per workgroup -- as few as leave one workgroup per CU.  (All workgroups must be
// resident at once and the kernel needs 180-200 VGPRs: a CU holds ONE 512-thread workgroup of it, whatever the LDS says.)
static long long sched_grid_workgroups(const ls_ctx *ctx, bool gen, int *cpw_out)
{
    const long long total = gen ? (long long)ctx->sched.n_days_stored * ctx->sched.n_employees : ctx->sc_resident_n;
    const long long chunks = (total + 64 * SCS_NW - 1) / (64 * SCS_NW);
    int cpw = 1;
    if (gen) {
        cpw = (int)((chunks + ctx->n_cus - 1) / ctx->n_cus);
        if (cpw < 1) cpw = 1;
    }
    if (cpw_out) *cpw_out = cpw;
    return (chunks + cpw - 1) / cpw;
}
static long long sched_grid_chunks(const ls_ctx *ctx) { return sched_grid_workgroups(ctx, false, nullptr); }

// ls_sched_eval_exhaustive for one state on the resident grid: S x E candidates in at most 4 chunks for each of <= 2 x CUs workgroups
bool sched_xgrid_eligible(const ls_ctx *ctx)
{
    if (ctx->sched_server && ctx->sched_server->grid_unavailable) return false;
    int cpw = 1;
    const long long per = (long long)ctx->sched.n_days_stored * ctx->sched.n_employees;
    if (!(ctx->sched_server_on && ctx->sched_delta && ctx->sched.valid && per > 64 * SCS_NW && per < (1ll << 31) &&
          ctx->sched.n_employees <= 32768 && ctx->sched.n_days_stored <= 64 - SCG_STATE_AT))
        return false;
    const long long wgs = sched_grid_workgroups(ctx, true, &cpw);
    return cpw <= SCG_MAXC && wgs <= ctx->n_cus && wgs <= 64 * SCS_NW;
}

bool sched_grid_eligible(const ls_ctx *ctx, long long n_moves)
{
    if (ctx->sched_server && ctx->sched_server->grid_unavailable) return false;
    // every workgroup must be resident at once: one 512-thread workgroup per CU
    return ctx->sched_server_on && ctx->sched_delta && ctx->sched.valid && n_moves > 64 * SCS_NW && n_moves <= ctx->sc_resident_n &&
           ctx->sched.n_employees <= 32768 && ctx->sched.n_days_stored <= 64 - SCG_STATE_AT && n_moves < (1ll << 31) &&
           sched_grid_chunks(ctx) <= ctx->n_cus &&        // co-residency: one workgroup per CU (180-200 VGPRs x 512 threads)
           sched_grid_chunks(ctx) <= 64 * SCS_NW;          // one record per finisher thread
}

int sched_grid_eval(ls_ctx *ctx, const int32_t *state, long long n_moves, long long window, ls_sched_result *out, bool gen)
{
    static const bool trace_call = getenv("LS_AMD_TRACE") != nullptr;
    const auto TC0 = trace_call ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    struct CallTimer {
        ls_ctx *c; std::chrono::steady_clock::time_point t0; bool on;
        ~CallTimer() { if (on && c->sched_server) c->sched_server->t_call += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    } call_timer{ctx, TC0, trace_call};
    SchedServer *s = ctx->sched_server;
    if (!s) {
        s = new (std::nothrow) SchedServer();
        if (!s) return set_err(ctx, LS_ERR_NOMEM, "out of host memory");
        ctx->sched_server = s;
    }
    const int S = ctx->sched.n_days_stored, E = ctx->sched.n_employees;
    for (int d = 0; d < S; d++)
        if (state[d] < 0 || state[d] >= E)
            return set_err(ctx, LS_ERR_INVALID, "state entry %d = %d is not an employee index in [0, %d)", d, state[d], E);
    if (s->grid.stream && s->grid_gen != gen) { // the grid serves the other kind of request: it leaves, the launch below is of this kind
        LS_HIP(ctx, hipSetDevice(ctx->device));
        if (!s->grid.drained()) {
            const unsigned tag = s->grid.next_tag(0);
            volatile unsigned *dst = s->grid.req(0);
            for (int d = 0; d < 16; d++) dst[d] = (tag << 16) | (d == 0 ? SRV_F_QUIT : 0u);
            s->grid.flush_lines();
        }
        LS_HIP(ctx, hipStreamSynchronize(s->grid.stream));
        s->grid_fin = 0;
    }
    s->grid_gen = gen;
    const long long nch = sched_grid_workgroups(ctx, gen, nullptr);
    if (!s->grid.stream) {
        LS_HIP(ctx, hipSetDevice(ctx->device));
        const int rc = s->grid.init(ctx, "scheduling grid server", 4, 0); // four lines = one message
        if (rc != LS_OK) return rc;
        s->grid.relaunch_limit = 2;
        LS_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&s->g), 512));
        hipLaunchKernelGGL(fill_i64_kernel, dim3(1), dim3(64), 0, s->grid.stream, reinterpret_cast<long long *>(s->g), 0ll, 64ll);
        LS_HIP(ctx, hipGetLastError());
        LS_HIP(ctx, hipStreamSynchronize(s->grid.stream));
        s->grid.launch = [s, ctx](const SrvLink &L) {
            const bool gen = s->grid_gen;
            int cpw = 1;
            const long long n = sched_grid_workgroups(ctx, gen, &cpw);
            SchedGridArgs a;
            a.link = L;
            a.in = make_inst(ctx);
            a.moves = gen ? nullptr : ctx->sc_resident_moves.as<ls_sched_move>();
            a.n_resident = gen ? (long long)ctx->sched.n_days_stored * ctx->sched.n_employees : ctx->sc_resident_n;
            a.cpw = cpw;
            a.poll_delay = getenv("LS_SCHED_GRID_POLL_DELAY") ? atoi(getenv("LS_SCHED_GRID_POLL_DELAY")) : -1; // measurement knob; default: adaptive
            a.g = s->g;
            s->ready_total += (unsigned long long)n;
            a.ready_target = s->ready_total;
            a.rec = s->ws.as<unsigned>();
            void *args[] = {&a};
            const bool m32 = ctx->sched.n_days_stored <= 32 && !ctx->sched_force_64;
            const void *fn = gen ? (m32 ? reinterpret_cast<const void *>(&sched_grid_server_kernel<unsigned, SCS_NW, true, SCG_MAXC>)
                                        : reinterpret_cast<const void *>(&sched_grid_server_kernel<u64, SCS_NW, true, SCG_MAXC>))
                                 : (m32 ? reinterpret_cast<const void *>(&sched_grid_server_kernel<unsigned, SCS_NW, false, 1>)
                                        : reinterpret_cast<const void *>(&sched_grid_server_kernel<u64, SCS_NW, false, 1>));
            const size_t dyn = sched_emp_table_bytes(ctx, fn, m32); // the employee -> scored-days table (one workgroup per CU: it has the LDS to itself)
            a.emp_entries = dyn ? ctx->sched.n_employees : 0;
            return hipLaunchKernel(fn, dim3((unsigned)n), dim3(64 * SCS_NW), args, dyn, s->grid.stream);
        };
    }
    {   // workspaces for the current resident list (they only change while the launch is paused: sched_server_pause)
        const size_t need = align256((size_t)nch * 32);
        if (need > s->ws.cap) {
            sched_server_pause(ctx); // nothing resident may still use the old block
            LS_HIP(ctx, hipSetDevice(ctx->device));
            LS_RESERVE(ctx, s->ws, need);
            s->ws.collect();
            // tag 0 everywhere: an uninitialised record must not look like an answer to the current request
            hipLaunchKernelGGL(fill_i32_kernel, dim3(8), dim3(256), 0, s->grid.stream, s->ws.as<int>(), 0, (long long)(s->ws.cap / 4));
            LS_HIP(ctx, hipGetLastError());
            LS_HIP(ctx, hipStreamSynchronize(s->grid.stream));
        }
    }
    // the request: ONE message of four lines -- header dwords 0..15, the state from dword 16 on -- every dword with the same
    // new tag; the master accepts a snapshot only when all 64 tags agree, so the order of these stores does not matter
    {
        int rc = s->grid.ensure_running(ctx);
        if (rc != LS_OK) return rc;
        const unsigned tag = s->grid.next_tag(0);
        const unsigned w = (unsigned)(window > 0x7fffffffll ? 0x7fffffffll : window);
        unsigned pay[64] = {0};
        pay[0] = SCS_F_EVAL; pay[1] = (unsigned)n_moves & 0xffffu; pay[2] = (unsigned)n_moves >> 16; pay[3] = w & 0xffffu; pay[4] = w >> 16;
        pay[5] = (unsigned)s->grid_fin; // the finisher: the chunk the window cut last time (its lanes still hold the keys the cut needs)
        for (int d = 0; d < S; d++) pay[SCG_STATE_AT + d] = (unsigned)state[d] & 0xffffu;
        alignas(64) unsigned msg[64];
        for (int d = 0; d < 64; d++) msg[d] = (tag << 16) | pay[d];
        volatile unsigned *dst = s->grid.req(0);
        s->grid.flush_lines();
        for (int l = 3; l >= 0; l--) srv_store_line(dst + 16 * l, msg + 16 * l);
        s->grid.flush_lines();
        s->grid_requests++;
    }
    static const bool trace = getenv("LS_AMD_TRACE") != nullptr;
    const auto T1 = trace ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    unsigned r[8];
    const int rc = s->grid.wait(ctx, 0, r);
    if (rc == SRV_RC_GIVE_UP) { // the workgroups could not all be resident (twice): this context launches per call from now on
        s->grid_unavailable = true;
        return SRV_RC_GIVE_UP;
    }
    if (rc != LS_OK) return rc;
    if (trace) {
        s->t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - T1).count();
        const volatile unsigned *rl = s->grid.resp(0);
        for (int d = 0; d < 8; d++) s->dev_ticks[d] += (rl[8 + d] >> 16) == s->grid.tags[0] ? rl[8 + d] & 0xffffu : 0u;
    }
    if (r[7] != 0) {
        if ((int)r[7] - 1 == s->grid_fin) s->grid_fin_hits++;
        s->grid_fin = (int)r[7] - 1;
    }
    if (!(r[0] & 1u)) { out->n_scored = 0; out->best_index = -1; out->best_hard = 0; out->best_soft = 0; }
    else {
        out->n_scored = (long long)(r[1] | (r[2] << 16));
        out->best_index = (long long)(r[3] | (r[4] << 16));
        out->best_hard = r[5];
        out->best_soft = r[6];
    }
    return LS_OK;
}

} // namespace lsamd

#ifdef LS_NQ_STAMPS
extern "C" int ls_debug_sched_stamps(unsigned long long *out, int count)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lsamd::g_sc_stamps), (size_t)count * 8);
}
#endif
