// sched_delta.hip — the LAUNCHED pipelines of the delta scorer (sched_device.hpp: one LANE per candidate, every term a function of
// per-employee day sets) for the employee-scheduling neighbourhood, local-search/src/local_search.rs:315-325 with the plug-in of
// examples/employee-scheduling/src/lib.rs (:261-375 scorer, :464-479 / :526-550 moves):
//   sched_eval_small_kernel                          lists of <= 512 moves: one block per restart does everything
//   sched_eval_all_kernel -> sched_finish_kernel     longer lists (and the exhaustive proposer, generated in place): TWO launches
//   count -> locate -> sched_eval_delta_kernel -> reduce   per-candidate score dumps of long lists (passes of sched_wave.hip)
// and the dispatcher behind ls_sched_eval_batch[_dev] / ls_sched_eval_exhaustive.  These are what a batched (multi-restart)
// call runs, and the fallback of the resident routes (sched_server.hip, sched_grid.hip; LS_SCHED_SERVER=0 forces it).
#include <cstdio>
#include <cstdlib>

#ifdef LS_NQ_STAMPS
// Diagnostic build only (tools/sched_stamps.py): wall-clock stamps (100 MHz) of the delta kernels' phases, written to a
// buffer no other code reads.  The product build has none of this.
namespace lsamd { __device__ unsigned long long g_sc_stamps[1024 * 16]; }
#define LS_SC_STAMP(P) do { if (threadIdx.x == 0 && blockIdx.x < 1024) lsamd::g_sc_stamps[blockIdx.x * 16 + (P)] = wall_clock64(); } while (0)
#endif
#include "sched_device.hpp"

namespace lsamd {

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
    static_assert(NW >= SCHED_TAB_PARTS, "the base table is built by SCHED_TAB_PARTS wavefronts side by side");
    if (wib < SCHED_TAB_PARTS) L = a.in.lane_tab[lane];
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
    if (wib < SCHED_TAB_PARTS) sched_build_base_tab<M>(a.in, wib, e_lane, hol_base, L, lane, T);
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
    static_assert(NW >= SCHED_TAB_PARTS, "the base table is built by SCHED_TAB_PARTS wavefronts side by side");
    if (wib < SCHED_TAB_PARTS) L = a.in.lane_tab[lane];
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
    if (wib < SCHED_TAB_PARTS) sched_build_base_tab<M>(a.in, wib, e_lane, hol_base, L, lane, T);
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
    static_assert(NW >= SCHED_TAB_PARTS, "the base table is built by SCHED_TAB_PARTS wavefronts side by side");
    if (wib < SCHED_TAB_PARTS) L = a.in.lane_tab[lane];
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
    if (wib < SCHED_TAB_PARTS) sched_build_base_tab<M>(a.in, wib, e_lane, hol_base, L, lane, T, emp);
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

bool sched_is_fused_launch(const ls_ctx *ctx, long long max_moves)
{
    const long long n_chunks = max_moves > 0 ? (max_moves + SC_COUNT_CHUNK - 1) / SC_COUNT_CHUNK : 1;
    const long long batches = max_moves > 0 ? (max_moves + 63) / 64 : 1;
    return n_chunks == 1 && batches * 4 <= 2 * SC_FUSED_WAVES;
}

// the two-launch delta pipeline (sched_eval_all_kernel + sched_finish_kernel); a holds everything but the workspace
static int sched_launch_delta_pipeline(ls_ctx *ctx, SchedEvalArgs a, bool gen)
{
    constexpr int DW = 8;
    // Batches of 64 candidates per wavefront: one while every CU gets at most two blocks (a block lives ~5 us, mostly latency);
    // beyond that the dispatcher's ~2 ns per wavefront is the limit and each wavefront takes 2 or 3 batches behind one base
    // table (exhaustive 320 k candidates: 16.8 / 14.5 / 14.1 us for 1 / 2 / 3; window 65 536 of 82 k: 10.2 / 12.4 / 16.3).
    int bpw = 1;
    for (; bpw < 3; bpw++) {
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
    const bool m32 = sched_m32(ctx);
    const dim3 g1((unsigned)(nch * a.n_restarts)), g2((unsigned)a.n_restarts), b(64 * DW);
    // the employee -> scored-days table of a block: when three blocks of a CU can still hold one each
    size_t dyn = ((size_t)ctx->sched.n_employees * (m32 ? 4 : 8) + 15) & ~(size_t)15;
    if (dyn > ((size_t)48 << 10)) dyn = 0;
    a.emp_entries = dyn ? ctx->sched.n_employees : 0;
#define LS_LAUNCH2(MT, GEN)                                                                             \
    {                                                                                                   \
        hipLaunchKernelGGL((sched_eval_all_kernel<MT, DW, GEN>), g1, b, dyn, ctx->stream, a);           \
        LS_HIP(ctx, hipGetLastError());                                                                 \
    }
    if (m32) { if (gen) LS_LAUNCH2(unsigned, true) else LS_LAUNCH2(unsigned, false) }
    else { if (gen) LS_LAUNCH2(u64, true) else LS_LAUNCH2(u64, false) }
#undef LS_LAUNCH2
    hipLaunchKernelGGL((sched_finish_kernel<DW>), g2, b, 0, ctx->stream, a, gen);
    LS_HIP(ctx, hipGetLastError());
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
    const bool m32 = sched_m32(ctx);
    constexpr int DW = 8; // wavefronts per block of the delta kernel
    const int split = delta ? 1 : (batches * n_restarts >= 4096 ? 1 : 4); // few batches: spread each over four wavefronts
    long long bpr = delta ? (batches + DW - 1) / DW : (batches * split + 3) / 4;
    const long long cap = ((long long)ctx->n_cus * 8) / n_restarts;
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
    const bool small = sched_is_fused_launch(ctx, max_moves); // one block per restart does the whole pipeline
    if (delta && !hard_out_d && !soft_out_d && !small) return sched_launch_delta_pipeline(ctx, a, false);
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
    if (small) { // the reference example's shape (window 100): locate + eval + reduce in one block per restart
        a.split = 4;
        a.blocks_per_restart = 1;
        if (!delta) return sched_wave_eval(ctx, a, 1, true);
        if (m32) hipLaunchKernelGGL((sched_eval_small_kernel<unsigned, DW>), dim3(n_restarts), dim3(64 * DW), 0, ctx->stream, a);
        else hipLaunchKernelGGL((sched_eval_small_kernel<u64, DW>), dim3(n_restarts), dim3(64 * DW), 0, ctx->stream, a);
        LS_HIP(ctx, hipGetLastError());
        return LS_OK;
    }
    int rc = sched_wave_locate(ctx, a);
    if (rc != LS_OK) return rc;
    if (!delta) rc = sched_wave_eval(ctx, a, bpr, false);
    else {
        if (m32) hipLaunchKernelGGL((sched_eval_delta_kernel<unsigned, DW>), dim3((unsigned)(bpr * n_restarts)), dim3(64 * DW), 0, ctx->stream, a);
        else hipLaunchKernelGGL((sched_eval_delta_kernel<u64, DW>), dim3((unsigned)(bpr * n_restarts)), dim3(64 * DW), 0, ctx->stream, a);
        LS_HIP(ctx, hipGetLastError());
    }
    if (rc != LS_OK) return rc;
    if (bpr > 1) return sched_wave_reduce(ctx, a); // with one block per restart pass 2 has already written the result
    return LS_OK;
}

} // namespace lsamd

#ifdef LS_NQ_STAMPS
extern "C" int ls_debug_sched_stamps(unsigned long long *out, int count)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lsamd::g_sc_stamps), (size_t)count * 8);
}
#endif
