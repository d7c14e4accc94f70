// sched_wave.hip — gfx950 kernels for the employee-scheduling neighbourhood
// pipeline: local-search/src/local_search.rs:315-325 with the plug-in of
// examples/employee-scheduling/src/lib.rs (scorer :261-375, helpers :149-222,
// moves :464-479 / :526-550).
//
// This file: the WAVEFRONT-PER-CANDIDATE family -- the north star's shape (LS_SCHED_DELTA=0 selects it for every neighbourhood
// launch; ls_sched_score and the count / locate / reduce passes of per-candidate score dumps always use it).  The default
// scorer is the one-lane-per-candidate delta formulation: sched_device.hpp, sched_delta.hip, sched_server.hip, sched_grid.hip.
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
#include "sched_device.hpp"

namespace lsamd {

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

int sched_wave_locate(ls_ctx *ctx, const SchedEvalArgs &a)
{
    if (a.n_chunks > 1) { // a single chunk is counted by the locate pass itself
        hipLaunchKernelGGL(sched_count_kernel, dim3((unsigned)a.n_chunks, (unsigned)a.n_restarts), dim3(256), 0, ctx->stream, a);
        LS_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(sched_locate_kernel, dim3(a.n_restarts), dim3(256), 0, ctx->stream, a);
    LS_HIP(ctx, hipGetLastError());
    return LS_OK;
}

int sched_wave_eval(ls_ctx *ctx, const SchedEvalArgs &a, long long bpr, bool fused)
{
    const bool half = sched_m32(ctx); // <= 32 stored days: two candidates per wavefront pass
    if (fused) {
        if (half) hipLaunchKernelGGL((sched_eval32_kernel<SC_FUSED_WAVES, true>), dim3(a.n_restarts), dim3(64 * SC_FUSED_WAVES), 0, ctx->stream, a);
        else hipLaunchKernelGGL((sched_eval_kernel<SC_FUSED_WAVES, true>), dim3(a.n_restarts), dim3(64 * SC_FUSED_WAVES), 0, ctx->stream, a);
    } else {
        if (half) hipLaunchKernelGGL((sched_eval32_kernel<4, false>), dim3((unsigned)(bpr * a.n_restarts)), dim3(256), 0, ctx->stream, a);
        else hipLaunchKernelGGL((sched_eval_kernel<4, false>), dim3((unsigned)(bpr * a.n_restarts)), dim3(256), 0, ctx->stream, a);
    }
    LS_HIP(ctx, hipGetLastError());
    return LS_OK;
}

int sched_wave_reduce(ls_ctx *ctx, const SchedEvalArgs &a)
{
    hipLaunchKernelGGL(sched_reduce_kernel, dim3(a.n_restarts), dim3(256), 0, ctx->stream, a);
    LS_HIP(ctx, hipGetLastError());
    return LS_OK;
}

} // namespace lsamd
