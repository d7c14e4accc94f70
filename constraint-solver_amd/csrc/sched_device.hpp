// sched_device.hpp -- what the employee-scheduling kernels share (device side + the small host helpers of their launchers):
// the instance view, the scorer of examples/employee-scheduling/src/lib.rs:261-375 in its two forms -- per day, one wavefront
// per candidate (sched_terms), and per employee day set, one LANE per candidate (the delta scorer: SchedBaseTab,
// sched_delta_score) --, the argmin key of (hard, soft, date_to_employee vector) (local_search.rs:29-37), and the
// launch arguments.  Kernels live in sched_wave.hip (wavefront per candidate; count / locate / reduce passes, score dumps),
// sched_delta.hip (the launched delta pipeline), sched_server.hip (resident LS step: one workgroup per slot) and
// sched_grid.hip (resident grid for lists beyond one workgroup).
#pragma once

#include <cstdlib>

#include "ls_common.hpp"
#include "sched_common.hpp"

namespace lsamd {

typedef unsigned long long u64;

#ifndef LS_SC_STAMP // (diagnostic builds of sched_delta.hip define it before this header: wall-clock stamps of the delta kernels' phases)
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
    u64 *cand_keys;                 // delta pipeline: [n_restarts][max_moves] argmin key per candidate (SC_KEY_EMPTY: not a candidate)
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

__device__ __forceinline__ unsigned sc_edit_key32(int pos, int newv, int basev) // the same value (it has 23 bits)
{
    const bool up = newv > basev;
    return (up ? 2u << 21 : 0u) | ((unsigned)(up ? 63 - pos : pos) << 15) | (unsigned)newv;
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
    if (a.done_flag) {
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

// the base state of one restart, tabulated in LDS.  One ROW per stored day -- everything the scorer needs of an edited day in two
// (three: 64-day masks) 16-byte reads whose addresses follow from the edit alone, so a candidate issues all its reads but one
// (the H4 | S1 values of an arbitrary employee's first day) before it waits for any.
template <typename M> struct SchedDayRow;
template <> struct alignas(16) SchedDayRow<unsigned> {
    unsigned eq;   // scored days worked by the employee of day d
    unsigned adj;  // scored days paired with day d by H2 / H3 (0 for an unscored day)
    unsigned wd;   // scored days on the weekday of day d (0: weekend or unscored)
    unsigned gh;   // F14 | F7 << 8 of eq | holiday bit of (employee of day d, day d) << 16
    unsigned gb[3]; // the weekday group of day d: bit i: i + 1 of its days held by at least 1 / 2 / 3 employees
    unsigned gsv;  // days of the group | its S2 value in the base state << 16
};
template <> struct alignas(16) SchedDayRow<u64> {
    u64 eq, adj;
    u64 wd;
    unsigned gh, pad;
    unsigned gb[3];
    unsigned gsv;
};
template <typename M> struct SchedBaseTab {
    SchedDayRow<M> row[64];
    // the two range terms over all employees: which values are held by at least 1 / 2 / 3 employees
    M tot_b[3];            // bit i: scored-day count i + 1 (a count is at most the number of scored days: it fits the day mask's type)
    unsigned wk_b[3];      // bit i: i weekend days (among the employees present)
    M valid7, valid14, weekend;
    int hard, soft_s1, soft_s2[2]; // base hard score; S1 and S2 (weekday residues 0-2, 3-6) of the base state
};
// the table's wavefront-uniform part, in scalar registers: read once per request, not once per candidate
template <typename M> struct SchedBaseUni {
    M tot_b[3];
    unsigned wk_b[3];
    M valid7, valid14, weekend;
    int hard, soft12;
};
__device__ __forceinline__ unsigned sc_uniform(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ u64 sc_uniform(u64 x) { return ((u64)sc_uniform((unsigned)(x >> 32)) << 32) | sc_uniform((unsigned)x); }
template <typename M> __device__ __forceinline__ SchedBaseUni<M> sched_base_uniforms(const SchedBaseTab<M> &T)
{
    SchedBaseUni<M> U;
#pragma unroll
    for (int i = 0; i < 3; i++) { U.tot_b[i] = sc_uniform(T.tot_b[i]); U.wk_b[i] = sc_uniform(T.wk_b[i]); }
    U.valid7 = sc_uniform(T.valid7); U.valid14 = sc_uniform(T.valid14); U.weekend = sc_uniform(T.weekend);
    U.hard = (int)sc_uniform((unsigned)T.hard);
    U.soft12 = (int)sc_uniform((unsigned)(T.soft_s1 + T.soft_s2[0] + T.soft_s2[1]));
    return U;
}

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

// inclusive prefix sum over the lanes of the wavefront (six DPP adds; a __shfl_up scan costs six dependent LDS-crossbar round trips)
__device__ __forceinline__ int wave_prefix_sum_i32(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_ROW_SHR(1), 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_ROW_SHR(2), 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_ROW_SHR(4), 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_ROW_SHR(8), 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_ROW_BCAST15, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_ROW_BCAST31, 0xc, 0xf, false);
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

// The table is filled by the first SCHED_TAB_PARTS wavefronts of the block side by side -- one per SIMD: six wavefronts, two to a
// SIMD, were measured slower in the grid (the LDS atomics of the day sets and the DPP chains queue behind each other) -- (lane = day;
// each derives the day-equality row of its lane first), two DPP value-set chains per wavefront at most; the caller synchronises the
// block after.
//   part 0: per-day rows (day set, pair / weekday masks, H4 | S1 values, holiday bit), base hard score, S1
//   part 1: S3 and S4 value sets (day counts, weekend-day counts)
//   part 2: S2 value sets of the weekday residues 0-2, group sizes and base values into the rows of those days
//   part 3: the same for residues 3-5 and 6
constexpr int SCHED_TAB_PARTS = 4;
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
        T.row[lane].eq = eq;
        T.row[lane].adj = (M)L.adj;
        T.row[lane].wd = wdm;
        T.row[lane].gh = (unsigned)(f14 | (f7 << 8) | (hol << 16)); // F14 <= 51, F7 <= 58
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
        M tot_onehot = 0;
        unsigned wk_onehot = 0;
        if (rep) {
            tot_onehot = (M)1 << (O::popc(eq) - 1);
            wk_onehot = 1u << O::popc(eq & weekend);
        }
        unsigned tl[3], th[3] = {0, 0, 0}, wk[3];
        wave_value_sets_u32((unsigned)tot_onehot, tl[0], tl[1], tl[2]);
        if (sizeof(M) > 4) wave_value_sets_u32((unsigned)((u64)tot_onehot >> 32), th[0], th[1], th[2]);
        wave_value_sets_u32(wk_onehot, wk[0], wk[1], wk[2]);
        if (lane < 3) {
            const unsigned lo = lane == 0 ? tl[0] : (lane == 1 ? tl[1] : tl[2]), hi = lane == 0 ? th[0] : (lane == 1 ? th[1] : th[2]);
            T.tot_b[lane] = (M)(((u64)hi << 32) | lo);
            T.wk_b[lane] = lane == 0 ? wk[0] : (lane == 1 ? wk[1] : wk[2]);
        }
    } else {
        // weekday groups: residue r = lane % 7 occupies bits 10 * (r % 3) .. + 9 of word r / 3; part 2 has word 0, part 3 words 1 and 2
        const int gr = lane % 7;
        const bool mine = part == 2 ? gr < 3 : gr >= 3;
        unsigned g_onehot = 0;
        if (mine && wdm != 0 && (eq & wdm & low) == 0) g_onehot = 1u << (10 * (gr % 3) + O::popc(eq & wdm) - 1); // first day of its employee on this weekday
        unsigned g[3], g6[3] = {0, 0, 0};
        wave_value_sets_u32(gr < 6 ? g_onehot : 0, g[0], g[1], g[2]);
        if (part == 3 && __builtin_amdgcn_ballot_w64(gr == 6 && g_onehot != 0) != 0) // (residue 6 is a weekend day for four start weekdays of seven: no chain then)
            wave_value_sets_u32(gr == 6 ? g_onehot : 0, g6[0], g6[1], g6[2]);
        // every lane of these residues writes the group of ITS day's weekday into its row (lanes 0..6 are the first days of their residues:
        // the S2 sum is theirs)
        int v = 0;
        if (mine) {
            const int sh = 10 * (gr % 3);
            unsigned gb[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                gb[i] = ((gr == 6 ? g6[i] : g[i]) >> sh) & 0x3ffu;
                T.row[lane].gb[i] = gb[i];
            }
            const int gsize = L.gsize; // popcount of the lane's weekday mask: 0 on weekend and unscored lanes
            if (gsize > 0) {
                const int mn = __ffs((int)gb[0]); // smallest per-employee count on this weekday
                if (mn < gsize) v = mn;
            }
            T.row[lane].gsv = (unsigned)gsize | ((unsigned)v << 16);
            if (lane >= 7) v = 0;
        }
        const int s2 = wave_sum_i32(v);
        if (lane == 0) T.soft_s2[part - 2] = s2;
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
// of (n1, p1) | (n2, p2) << 1.  Only called for well-formed, non-tabu candidates (p2 == p1 for a ChangeDay).
template <typename M>
__device__ __forceinline__ void sched_delta_score(const SchedBaseTab<M> &T, const SchedBaseUni<M> &U, int e_lane, int D, bool swap, int p1, int p2,
                                                  int n1, int hbits, int &hard_out, int &soft_out, const M *emp = nullptr)
{
    typedef SchedMaskOps<M> O;
    // every read whose address the edit gives, before anything waits
    const SchedDayRow<M> r1 = T.row[p1];
    const SchedDayRow<M> r2 = T.row[p2];
    M mYe = 0;
    if (emp) mYe = emp[n1]; // the day set of an arbitrary employee, from the block's employee -> scored-days table
    else {                  // ... or by comparing with every scored day (e_lane: lane d holds day d)
#pragma unroll
        for (int d = 0; d < (int)sizeof(M) * 8; d++)
            if (d < D && __builtin_amdgcn_readlane(e_lane, d) == n1) mYe |= (M)1 << d;
    }
    const M bit1 = p1 < D ? (M)1 << p1 : 0, bit2 = (swap && p2 < D) ? (M)1 << p2 : 0;
    const M mX = r1.eq;
    const M mY = swap ? r2.eq : mYe;
    // the one dependent read: H4 | S1 of Y's day set, kept with its first day
    const int iy = swap ? p2 : (mY ? O::lsb(mY) : 0);
    const unsigned ghY = T.row[iy].gh;
    const int gX = (int)(r1.gh & 0xffffu);
    const M mXn = (mX & ~bit1) | bit2, mYn = (mY & ~bit2) | bit1;
    const M A1 = r1.adj, A2 = swap ? r2.adj : 0;
    int f7X, f14X, f7Y, f14Y;
    sched_window_counts<M>(mXn, U.valid7, U.valid14, f7X, f14X);
    sched_window_counts<M>(mYn, U.valid7, U.valid14, f7Y, f14Y);
    int hard = U.hard + (hbits & 1) - (int)((r1.gh >> 16) & 1u);
    if (swap) hard += ((hbits >> 1) & 1) - (int)((r2.gh >> 16) & 1u);
    hard += O::popc(A1 & mYn) - O::popc(A1 & mX) + O::popc(A2 & mXn) - O::popc(A2 & mY);
    int soft = U.soft12 + f7X + f7Y - (gX >> 8);
    hard += f14X + f14Y - (gX & 0xff);
    const int cX = O::popc(mX), cY = O::popc(mY), cXn = O::popc(mXn), cYn = O::popc(mYn);
    const bool hXo = cX > 0, hYo = cY > 0, hXn = cXn > 0, hYn = cYn > 0;
    {   // S3: day counts (value index = count - 1)
        const M b = sched_replace_values<M>(U.tot_b, hXo, cX - 1, hYo, cY - 1, hXn, cXn - 1, hYn, cYn - 1);
        soft += O::msb(b) - O::lsb(b);
    }
    {   // S4: weekend-day counts of the employees present
        const M wk = U.weekend;
        const unsigned b = sched_replace_values<unsigned>(U.wk_b, hXo, O::popc(mX & wk), hYo, O::popc(mY & wk), hXn,
                                                          O::popc(mXn & wk), hYn, O::popc(mYn & wk));
        soft += (31 - __clz((int)b)) - (__ffs((int)b) - 1);
    }
    // S2: the weekday groups of the edited days
    const M G1 = r1.wd, G2 = swap ? r2.wd : 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const M G = i == 0 ? G1 : G2;
        if (G == 0 || (i == 1 && G == G1)) continue;
        const SchedDayRow<M> &r = i == 0 ? r1 : r2;
        const int a = O::popc(mX & G), b = O::popc(mY & G), an = O::popc(mXn & G), bn = O::popc(mYn & G);
        const unsigned bits = sched_replace_values<unsigned>(r.gb, a > 0, a - 1, b > 0, b - 1, an > 0, an - 1, bn > 0, bn - 1);
        const int mn = __ffs((int)bits);
        soft += (mn < (int)(r.gsv & 0xffffu) ? mn : 0) - (int)(r.gsv >> 16);
    }
    // (the late read last)
    const int gY = (swap || mY) ? (int)(ghY & 0xffffu) : 0;
    hard_out = hard - (gY & 0xff);
    soft_out = soft - (gY >> 8);
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

// The same with the state read from LDS (e[d] = employee of stored day d, d < S) instead of exchanged between the lanes: a
// ds_bpermute costs the LDS crossbar a whole pass per wavefront -- two per candidate were 0.3 us per 512-candidate chunk in a
// resident workgroup -- a ds_read_b32 of at most S consecutive dwords is conflict-free and ten times cheaper.
template <typename T> __device__ __forceinline__ SchedEdit sched_edit_of_lds(const T *e, int S, int E, int type, int ma, int mb)
{
    SchedEdit ed;
    const bool ok = sched_move_ok(type, ma, mb, S, E);
    ed.swap = type == 1;
    ed.p1 = ok ? ma : 0;
    ed.p2 = (ok && type == 1) ? mb : ed.p1;
    ed.ea = (int)e[ed.p1];
    ed.eb = (int)e[ed.p2];
    ed.new1 = type == 0 ? mb : ed.eb;
    ed.new2 = type == 0 ? mb : ed.ea;
    ed.active = ok && !(ed.new1 == ed.ea && ed.new2 == ed.eb);
    return ed;
}
template <typename T> __device__ __forceinline__ void sched_gen_exhaustive_lds(const T *e, int S, int E, long long k, int &type, int &ma, int &mb)
{
    const unsigned d = (unsigned)k / (unsigned)E, i = (unsigned)k - d * (unsigned)E;
    type = 0;
    ma = (int)d;
    mb = (int)(((unsigned)(d < (unsigned)S ? e[d] : 0) + 1u + i) % (unsigned)E);
}

// holiday bits of the edited days under their new employees: a gather from global memory that depends on the move
// and the state only, so it is issued before the block waits for the table
__device__ __forceinline__ int sched_edit_hbits(const SchedInst &in, const SchedEdit &ed)
{
    if (!ed.active) return 0;
    return sched_holiday_bit(in, ed.new1, ed.p1) | (sched_holiday_bit(in, ed.new2, ed.p2) << 1);
}

// The same in two steps, for kernels that have several candidates per lane in flight: the holiday WORDS of the two new employees
// are fetched first (the loads of all candidates are issued before anything waits for one), the bits are taken when they are
// needed.  M = unsigned (<= 32 days): the low half of the 64-bit word is all there is to read.
// (hol: a resident workgroup's copy of the words in LDS, or nullptr -- the gather then goes to global memory)
template <typename M> __device__ __forceinline__ M sched_hol_word(const SchedInst &in, int employee, const M *hol = nullptr)
{
    if ((unsigned)employee >= (unsigned)in.n_employees) return 0;
    return hol ? hol[employee] : reinterpret_cast<const M *>(in.holiday_bits + employee)[0];
}
template <typename M> __device__ __forceinline__ int sched_holiday_bit_of(const SchedInst &in, int employee, int day, const M *hol)
{
    if ((unsigned)day >= (unsigned)in.D) return 0;
    return (int)((sched_hol_word<M>(in, employee, hol) >> day) & 1);
}
template <typename M> __device__ __forceinline__ int sched_hbits_of(const SchedInst &in, const SchedEdit &ed, M w1, M w2)
{
    if (!ed.active) return 0;
    const int b1 = (unsigned)ed.p1 < (unsigned)in.D ? (int)((w1 >> ed.p1) & 1) : 0;
    const int b2 = (unsigned)ed.p2 < (unsigned)in.D ? (int)((w2 >> ed.p2) & 1) : 0;
    return b1 | (b2 << 1);
}

// the argmin key of an active candidate (and its scores)
template <typename M>
__device__ __forceinline__ u64 sched_delta_key(const SchedBaseTab<M> &T, const SchedBaseUni<M> &U, int D, int e_lane, const SchedEdit &ed, int hbits,
                                               int &hard, int &soft, const M *emp = nullptr)
{
    // the edits' keys (sc_edit_key: 23 bits each) and the whole key in 32-bit halves, by selects: no 64-bit shifts, no divergent branch
    const unsigned k1 = sc_edit_key32(ed.p1, ed.new1, ed.ea), k2 = sc_edit_key32(ed.p2, ed.new2, ed.eb);
    const bool a_first = ed.p1 < ed.p2;
    const unsigned e1 = (!ed.swap || a_first) ? k1 : k2;
    const unsigned e2 = !ed.swap ? 1u << 21 : (a_first ? k2 : k1); // "no second edit" sorts between lowering and raising edits
    sched_delta_score<M>(T, U, e_lane, D, ed.swap, ed.p1, ed.p2, ed.new1, hbits, hard, soft, emp);
    const unsigned hi = ((unsigned)hard << 22) | ((unsigned)soft << 14) | (e1 >> 9), lo = (e1 << 23) | e2; // hard << 54 | soft << 46 | e1 << 23 | e2
    return ((u64)hi << 32) | lo;
}
// (one candidate per lane and request: the uniforms are read on the way)
template <typename M>
__device__ __forceinline__ u64 sched_delta_key(const SchedBaseTab<M> &T, int D, int e_lane, const SchedEdit &ed, int hbits, int &hard, int &soft,
                                               const M *emp = nullptr)
{
    const SchedBaseUni<M> U = sched_base_uniforms<M>(T);
    return sched_delta_key<M>(T, U, D, e_lane, ed, hbits, hard, soft, emp);
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

// ------------------------------------------------------------------ launchers
// ------------------------------------------------------------------------------------------------------------ host side
static inline SchedInst make_inst(const ls_ctx *ctx)
{
    SchedInst in;
    in.D = ctx->sched.n_days_scored;
    in.S = ctx->sched.n_days_stored;
    in.n_employees = ctx->sched.n_employees;
    in.holiday_bits = ctx->sched.holiday_bits.as<u64>();
    in.lane_tab = ctx->sched.lane_tab.as<SchedLaneConst>();
    return in;
}

// day sets fit 32-bit masks (both configurations the reference ships: 32 stored days)
static inline bool sched_m32(const ls_ctx *ctx) { return ctx->sched.n_days_stored <= 32; }

constexpr int SCS_NW = 8;                 // wavefronts of a resident workgroup: 512 candidates per pass
constexpr int SCS_CHUNK = 64 * SCS_NW;

// Dynamic LDS for a resident workgroup's employee -> scored-days table (one mask per employee), or 0 when it does not fit
// beside the static LDS of the kernel (160 KB per CU on gfx950): the scorers then compare day by day.
// The same with a copy of the instance's holiday words behind it (one M per employee: the gathers of a request's holiday bits
// become LDS reads), when both fit under `cap`: *hol_entries = n_employees then, else 0 and the table alone as below.
static inline size_t sched_emp_table_bytes(const ls_ctx *ctx, const void *fn, bool m32, size_t cap = (size_t)96 << 10);
static inline size_t sched_lds_tables_bytes(const ls_ctx *ctx, const void *fn, bool m32, size_t cap_both, size_t cap_emp, int *hol_entries)
{
    *hol_entries = 0;
    const size_t one = (size_t)ctx->sched.n_employees * (m32 ? 4 : 8);
    if (one != 0 && 2 * one <= cap_both) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * one)) == hipSuccess) {
            *hol_entries = ctx->sched.n_employees;
            return 2 * one;
        }
        (void)hipGetLastError();
    }
    return sched_emp_table_bytes(ctx, fn, m32, cap_emp);
}
static inline size_t sched_emp_table_bytes(const ls_ctx *ctx, const void *fn, bool m32, size_t cap)
{
    const size_t dyn = (size_t)ctx->sched.n_employees * (m32 ? 4 : 8);
    if (dyn == 0 || dyn > cap) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return dyn;
}

// sched_wave.hip: the passes of the wavefront-per-candidate family, as launchers (a kernel is launched from the file that defines it)
int sched_wave_locate(ls_ctx *ctx, const SchedEvalArgs &a);                      // count (lists beyond one chunk) + locate: k_end, n_scored
int sched_wave_eval(ls_ctx *ctx, const SchedEvalArgs &a, long long bpr, bool fused); // the scoring launch (fused: locate + eval + reduce, one block per restart)
int sched_wave_reduce(ls_ctx *ctx, const SchedEvalArgs &a);                      // minimum over the per-block minima

} // namespace lsamd
