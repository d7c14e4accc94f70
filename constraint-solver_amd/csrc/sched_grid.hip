// sched_grid.hip — the resident scheduling GRID: the neighbourhood pipeline (local-search/src/local_search.rs:315-325, plug-in
// examples/employee-scheduling/src/lib.rs) for candidate lists beyond one workgroup -- config 5: 10 000 employees, window 65 536
// over a stream of ~82 000 moves, or the exhaustive proposer's 32 x 10 000 candidates (lib.rs:493-559) -- without a launch per
// LS iteration.  ONE launch that is not relaunched holds n_sub SUB-GRIDS of G workgroups; a sub-grid serves one restart:
// a context's own grid has one (ls_sched_eval_resident beyond 512 moves, ls_sched_eval_exhaustive of one state), and
// ls_multi_solver_create_sched runs several restarts per GPU as the sub-grids of one shared launch.
//
// Workgroup c of a sub-grid owns `cpw` consecutive 512-candidate chunks of the restart's list; every lane keeps its moves in
// registers (the random proposer's stream never changes during a run, lib.rs:488: it is read once per restart from the
// sub-grid's pinned block, the request names its id; the exhaustive proposer's candidates are generated from the request's
// state).  A request is ONE message of four tagged lines (srv_common.hpp): header + the whole state, one entry per dword -- the
// workgroups keep no state, nothing to apply or re-upload.  Workgroup 0 of the sub-grid (its master) owns the host mailbox and
// copies the four lines verbatim, tags and all, to a device-side "go" record, which every workgroup polls with ONE 64-lane load
// and accepts when all tags agree: the torn-snapshot rule of the PCIe mailbox reused inside the device -- no fence, no ordering,
// no cache write-back / invalidate between the XCDs' L2s (agent-scope acquire / release pairs cost an LS iteration 60 us).
// Every workgroup scores its chunks (the delta scorer, sched_device.hpp) and writes ONE 16-byte record -- four dwords, each
// tagged: count, minimum key, its position -- with a single store instruction; the request's finisher (named in the header: the
// workgroup whose chunks the window cut last time) polls the records with one 16-byte read per lane until all carry the tag,
// takes the prefix over the counts and the minima of the workgroups inside the window, applies the window to the cut workgroup
// -- itself, with the keys still in its lanes, or another, whose chunks it scores AGAIN up to the window (same state, same base
// table) -- and answers the host with a tagged line.  A single-state score request (get_scored_solution, lib.rs:261-375) is
// answered by the master alone.
// All workgroups of the launch must be resident at once (they wait for each other): one 512-thread workgroup per CU.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>

#include "sched_device.hpp"
#include "srv_common.hpp"

namespace lsamd {

constexpr unsigned SCG_F_EVAL = 1u, SCG_F_SCORE = 2u; // (bit 2 is SRV_F_QUIT)
constexpr int SCG_STATE_AT = 16; // dword of the message that holds state[0]: the state fits lines 1..3 (S <= 48)
constexpr int SCG_MAXC = 8;      // chunks per workgroup, at most

struct SchedGridCtl {            // device memory
    unsigned long long ready;    // workgroups that have started, over all launches of this server
    unsigned long long pad[7];
};

struct SchedGridArgs {
    SrvLink link;                // lines 4 s .. 4 s + 3 = sub-grid s's message; its answer goes to line 4 s
    SchedInst in;
    const ls_sched_move *streams; // pinned host: [n_sub][stream_cap] (unused by the exhaustive grid)
    long long stream_cap;        // = G * cpw * 512
    long long gen_total;         // exhaustive grid: S x E candidates
    SchedGridCtl *g;
    unsigned *go;                // [n_sub][64] the current request of every sub-grid, as the host wrote it (tagged dwords): device memory
                                 // the masters copy it to -- or, DIRECT (go == link.req), fine-grained device memory the host itself writes
    unsigned long long ready_target; // g->ready once every workgroup of THIS launch has started (the host adds the grid size per launch)
    unsigned *rec;               // [n_sub][G][4] every workgroup's record of its sub-grid's current request (tagged dwords)
    int G;                       // workgroups per sub-grid
    int cpw;                     // chunks per workgroup (consecutive ones)
    int emp_entries;             // entries of the workgroup's employee -> scored-days table in dynamic LDS (0: the scorers compare day by day)
    int hol_entries;             // entries of the workgroup's copy of the holiday words behind it (0: the holiday bits are gathered from global memory)
};

__device__ __forceinline__ unsigned scg_tag8(unsigned tag16) { return tag16 % 255u + 1u; } // 1..255: never the 0 of a cleared record

// GEN: the candidates are those of the exhaustive proposer (ScheduleMoveProposer::iter_local_moves, lib.rs:493-559), generated in
// place from the request's state -- no move list.
template <typename M, int NW, bool GEN, int MAXC>
__global__ __launch_bounds__(64 * NW) void sched_grid_server_kernel(SchedGridArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int G = a.G;
    const int sub = (int)blockIdx.x / G, c = (int)blockIdx.x - sub * G;
    const int S = a.in.S, D = a.in.D, E = a.in.n_employees;
    const unsigned *req = a.link.req + (size_t)sub * 64;
    unsigned *resp = a.link.resp + (size_t)sub * 64;
    unsigned *go = a.go + (size_t)sub * 64;
    const bool direct = a.go == a.link.req; // the host writes the go records itself (through the PCIe BAR): no relay by the master
    unsigned *rec = a.rec + (size_t)sub * G * 4;
    const ls_sched_move *stream = GEN ? nullptr : a.streams + (size_t)sub * a.stream_cap;
    __shared__ SchedBaseTab<M> T;
    __shared__ u64 s_wkey[NW];      // per wavefront: its minimum key and that key's first position
    __shared__ unsigned s_wrel[NW];
    __shared__ int s_cnt[NW];
    __shared__ int s_cntc[MAXC][NW]; // real candidates per chunk and wavefront
    __shared__ int s_quit;           // the finisher: the launch is leaving, the records will not come
    __shared__ unsigned s_req[64];
    __shared__ int s_ctl[2]; // [0] leave, [1] the request's tag
    __shared__ int s_cut;
    __shared__ int s_before;
    __shared__ __align__(16) unsigned s_mv[3 * 64 * NW]; // staging of a chunk of the sub-grid's stream when it is fetched (once per restart)
    // Which scored days an employee works in the current state, for every employee: the table's day-equality rows and the day
    // set of a ChangeDay's new employee are one LDS read each instead of a 64-step compare loop (a third of the scorer's
    // instructions).  Zero between requests: the <= 64 entries a state touches are set before and cleared after its scoring.
    extern __shared__ __align__(16) unsigned char s_dyn[];
    M *const emp = a.emp_entries > 0 ? reinterpret_cast<M *>(s_dyn) : nullptr;
    for (int i = tid; i < a.emp_entries; i += 64 * NW) emp[i] = 0;
    // ... and the instance's holiday words (the low 32 days of each when M is 32 bits wide): a request's gathers of holiday bits --
    // the state's own, every candidate's two -- are LDS reads instead of global loads the table build and the scorers wait for
    M *const hol = a.hol_entries > 0 ? reinterpret_cast<M *>(s_dyn) + a.emp_entries : nullptr;
    for (int i = tid; i < a.hol_entries; i += 64 * NW) hol[i] = (M)a.in.holiday_bits[i];
    if (tid == 0) s_quit = 0;
    SchedLaneConst L{};
    static_assert(NW >= SCHED_TAB_PARTS, "the base table is built by SCHED_TAB_PARTS wavefronts side by side");
    if (wib < SCHED_TAB_PARTS) L = a.in.lane_tab[lane];
    const SchedInst &in = a.in;
    const int cpw = a.cpw;
    const int k0 = c * cpw * (64 * NW) + tid; // this lane's candidates: k0 + 512 j, j < cpw
    int mtype[MAXC], mma[MAXC], mmb[MAXC];    // resident stream: this lane's moves
#pragma unroll
    for (int j = 0; j < MAXC; j++) { mtype[j] = -1; mma[j] = 0; mmb[j] = 0; }
    unsigned my_stream = 0; // id of the stream in the lanes (0: none); block-uniform
    unsigned last_seq = 0;
    bool abort_launch = false; // a master: the workgroups of this launch are not all resident -- nothing is served, everybody leaves
    SrvPoll poll{0, 0, 0};
#ifdef LS_NQ_STAMPS
    unsigned long long gst[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // the finisher: request seen, edit + holiday bits in, table built, chunks scored, own record out, all records in, window placed, answer
#define SCG_STAMP(I) do { gst[I] = wall_clock64(); } while (0)
#else
#define SCG_STAMP(I) do { } while (0)
#endif
#if defined(LS_NQ_STAMPS) && defined(SCG_FINE) // experiment: stamps 1, 2 and 5 moved into the record phase
#define SCG_FSTAMP(I) do { gst[I] = wall_clock64(); } while (0)
#else
#define SCG_FSTAMP(I) do { } while (0)
#endif
    if (wib == 0) {
        // the last request this sub-grid ANSWERED (the answer line's tag, as srv_poll_begin reads it): a request that is still in the go
        // record from before this launch is served if, and only if, it has not been answered
        last_seq = __builtin_amdgcn_readfirstlane(srv_sys_load(resp) >> 16);
        if (c == 0) srv_poll_begin(poll, a.link, resp, lane);
        if (lane == 0) __hip_atomic_fetch_add(&a.g->ready, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c == 0) { // a master serves only once every workgroup of THIS launch polls its go record
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
            if (c == 0) { // the master: host mailbox -> go record, verbatim; a score request is answered here and now
                for (;;) {
                    unsigned raw = 0;
                    leave = abort_launch ? 1 : srv_poll_wait<64>(poll, a.link, req, s_req, lane, &raw);
                    if (leave) {
                        if (lane == 0) __hip_atomic_store(&a.link.ctl->quit_gen, a.link.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // tell the others
                        break;
                    }
                    if (!(s_req[0] & SCG_F_SCORE)) { // wave-uniform (LDS written by this wavefront, in order)
                        if (!direct) __hip_atomic_store(&go[lane], raw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    // get_scored_solution of the message's state, by the wavefront-per-state scorer (sched_wave.hip's sched_score_kernel)
                    const int e1 = lane < S ? (int)s_req[SCG_STATE_AT + lane] : -1 - lane;
                    const u64 eq = sched_build_eq(D, e1) & L.scored_mask;
                    const SchedScore r = sched_terms<false>(D, L, lane, eq, sched_holiday_bit(in, e1, lane)); // (before the first barrier: not from the LDS copy)
                    unsigned pay = 0;
                    if (lane == 5) pay = (unsigned)r.hard;
                    else if (lane == 6) pay = (unsigned)r.soft;
                    if (lane < 8) __hip_atomic_store(resp + lane, (poll.last_tag << 16) | (pay & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    srv_served(poll, a.link, lane);
                }
                last_seq = poll.last_tag;
                SCG_STAMP(0);
            } else {
                for (;;) { // one 64-lane snapshot of the go record per poll; valid when all tags agree and the tag is new
                    const unsigned v = __hip_atomic_load(&go[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned t0 = __builtin_amdgcn_readfirstlane(v >> 16);
                    if (t0 != last_seq && t0 != 0 && __builtin_amdgcn_ballot_w64((v >> 16) != t0) == 0) {
                        last_seq = t0;
                        const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffu));
                        if (flags & SRV_F_QUIT) { leave = 1; break; }   // (DIRECT only: the host's messages arrive here unfiltered)
                        if (flags & SCG_F_EVAL) {
                            s_req[lane] = v & 0xffffu;
                            SCG_STAMP(0);
                            break;
                        }
                        continue; // a score request: the master's alone
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
        const unsigned tag8 = scg_tag8((unsigned)s_ctl[1]); // the request's tag as the records carry it, in every wavefront
        int nm = (int)(s_req[1] | (s_req[2] << 16));
        {
            const long long lim = GEN ? a.gen_total : a.stream_cap;
            nm = nm < lim ? nm : (int)lim;
        }
        const int window = (int)(s_req[3] | (s_req[4] << 16));
        const int fin = (int)s_req[5] < G ? (int)s_req[5] : 0; // the finisher of THIS request: the workgroup the host expects the window to cut
        const unsigned stream_id = s_req[6];
        const int e_lane = lane < S ? (int)s_req[SCG_STATE_AT + lane] : -1 - lane;
        const int hol_base = wib == 0 ? sched_holiday_bit_of<M>(in, e_lane, lane, hol) : 0;
        if constexpr (!GEN) {
            if (stream_id != my_stream) { // block-uniform: a new restart on this sub-grid (or a relaunch): this lane's moves, once
                static_assert(NW == 8, "srv_fetch_moves_512 moves one 512-move chunk");
#pragma unroll
                for (int j = 0; j < MAXC; j++) {
                    mtype[j] = -1;
                    if (j < cpw) { // (the whole chunk, whatever this request's n_moves: a later request may ask for more of the same list)
                        srv_fetch_moves_512(stream + (size_t)(c * cpw + j) * (64 * NW), s_mv, tid);
                        mtype[j] = (int)s_mv[3 * tid]; mma[j] = (int)s_mv[3 * tid + 1]; mmb[j] = (int)s_mv[3 * tid + 2];
                    }
                }
                my_stream = stream_id;
            }
        }
        // ---- this workgroup's chunks: the delta scorer, one candidate per lane and chunk
        SchedEdit ed[MAXC];
        M hw1[MAXC], hw2[MAXC]; // the holiday words of the candidates' new employees: all fetched before anything waits for one
        // Straight-line over all MAXC chunks of the instantiation (a chunk beyond cpw has no moves: type -1, an inactive edit of day 0):
        // with a branch per chunk every chunk's two lane exchanges and two holiday reads were a round trip of their own -- 0.4 us per
        // chunk; now the exchanges of all chunks are in flight together, then the reads.
#pragma unroll
        for (int j = 0; j < MAXC; j++) {
            const int k = k0 + j * (64 * NW);
            int ty = -1, xa = 0, xb = 0;
            if constexpr (GEN) sched_gen_exhaustive_lds(s_req + SCG_STATE_AT, S, E, k, ty, xa, xb);
            else { ty = mtype[j]; xa = mma[j]; xb = mmb[j]; }
            if (k >= nm || j >= cpw) ty = -1;
            ed[j] = sched_edit_of_lds(s_req + SCG_STATE_AT, S, E, ty, xa, xb); // (the request's state, as every wavefront reads it for e_lane)
        }
#pragma unroll
        for (int j = 0; j < MAXC; j++) {
            hw1[j] = sched_hol_word<M>(in, ed[j].new1, hol);
            hw2[j] = sched_hol_word<M>(in, ed[j].new2, hol);
        }
#if defined(LS_NQ_STAMPS) && !defined(SCG_FINE)
        { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); gst[1] = wall_clock64(); }
#endif
        if (tid == 0) { s_cut = -1; s_before = 0; }
        if (wib < SCHED_TAB_PARTS) sched_build_base_tab<M>(in, wib, e_lane, hol_base, L, lane, T, emp);
        __syncthreads();
#ifndef SCG_FINE
        SCG_STAMP(2);
#endif
        const SchedBaseUni<M> U = sched_base_uniforms<M>(T); // the table's uniform part, once per request
        u64 best_key = SC_KEY_EMPTY;
        int best_rel = 0; // position of the lane's best candidate, relative to the workgroup's first
        u64 my_key[MAXC], my_bal[MAXC]; // this lane's own candidates: what the finisher needs if the window cuts ITS chunks
#pragma unroll
        for (int j = 0; j < MAXC; j++) {
            my_key[j] = SC_KEY_EMPTY;
            my_bal[j] = 0;
            if (j < cpw) { // block-uniform
                if (ed[j].active) {
                    int hard, soft;
                    my_key[j] = sched_delta_key<M>(T, U, D, e_lane, ed[j], sched_hbits_of<M>(in, ed[j], hw1[j], hw2[j]), hard, soft, emp);
                    if (my_key[j] < best_key) { best_key = my_key[j]; best_rel = tid + j * (64 * NW); }
                }
                my_bal[j] = __ballot(ed[j].active);
            }
            if (lane == 0) s_cntc[j][wib] = __popcll(my_bal[j]);
        }
        SCG_STAMP(3);
        // the workgroup's minimum (key, first position): every wavefront's pair through LDS, ONE barrier, then every wavefront
        // reduces the NW pairs itself (no second barrier, no serial walk)
        {
            const u64 wk = wave_min_u64(best_key);
            const unsigned wr = wave_min_u32(best_key == wk && wk != SC_KEY_EMPTY ? (unsigned)best_rel : ~0u);
            if (lane == 0) { s_wkey[wib] = wk; s_wrel[wib] = wr; }
        }
        SCG_FSTAMP(1);
        __syncthreads(); // the wavefronts' minima and the per-chunk counts are in LDS
        SCG_FSTAMP(2);
        u64 wg_key;
        unsigned wg_rel;
        {
            const u64 k8 = lane < NW ? s_wkey[lane] : SC_KEY_EMPTY;
            static_assert(NW == 8, "wave_min8: one value per wavefront in lanes 0..7");
            wg_key = wave_min8_u64(k8);
            wg_rel = wave_min8_u32(lane < NW && k8 == wg_key ? s_wrel[lane] : ~0u);
        }
        // real candidates per (chunk, wavefront) in candidate order, one per lane: the inclusive prefix gives the workgroup's total
        // and, for the finisher, the rank of every wavefront's first candidate (one LDS read and six DPP adds instead of MAXC x NW reads)
        static_assert(MAXC * NW <= 64, "one (chunk, wavefront) count per lane");
        const int cnt_l = lane < MAXC * NW ? (&s_cntc[0][0])[lane] : 0;
        const int cnt_incl = wave_prefix_sum_i32(cnt_l);
        const int total = __builtin_amdgcn_readlane(cnt_incl, 63);
        SCG_FSTAMP(5);
        // ---- this workgroup's record: four tagged dwords (count | key in 24 + 24 + 16 bits | position in 8 + 4 bits), ONE store by
        // four lanes, no acknowledgement wait and no ticket: the request's finisher polls the records until all carry the tag
        if (wib == 0) {
            const unsigned rel = wg_rel & 0xfffu; // < 512 MAXC = 4096
            unsigned pay = 0;
            if (lane == 0) pay = (unsigned)total | ((rel >> 8) << 13); // total <= 4096
            else if (lane == 1) pay = (unsigned)wg_key & 0xffffffu;
            else if (lane == 2) pay = (unsigned)(wg_key >> 24) & 0xffffffu;
            else if (lane == 3) pay = (unsigned)(wg_key >> 48) | ((rel & 0xffu) << 16);
            if (lane < 4) __hip_atomic_store(rec + (size_t)c * 4 + lane, (tag8 << 24) | pay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        SCG_STAMP(4);
        if (c == fin) { // ---- the finish step: prefix over the counts, minima of the workgroups inside the window, the cut workgroup up to the window
            int my_rank[MAXC]; // 1-based among the workgroup's real candidates, in candidate order (chunk, wavefront, lane): for a cut through THIS workgroup
            {
                const int cnt_excl = cnt_incl - cnt_l;
#pragma unroll
                for (int j = 0; j < MAXC; j++)
                    my_rank[j] = __builtin_amdgcn_readlane(cnt_excl, j * NW + wib) + __popcll(my_bal[j] & ((2ull << lane) - 1));
            }
            // the records of THIS request: one 16-byte read per lane and pass (two 8-byte loads that go to the coherence point,
            // past the L2), by as many wavefronts as there are 64 records; no barrier inside the poll
            const int npw = (G + 63) >> 6;
            u64 ckey = SC_KEY_EMPTY; // the minimum of the workgroup whose record this lane holds
            int cb = 0, crel = 0, incl = 0;
            if (wib < npw) {
                u64 lo = 0, hi = 0;
                for (unsigned pass = 0;; pass++) {
                    bool ok = true;
                    if (tid < G) {
                        const u64 *p = reinterpret_cast<const u64 *>(rec + (size_t)tid * 4);
                        lo = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        hi = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = (unsigned)(lo >> 24 & 0xffu) == tag8 && (unsigned)(lo >> 56) == tag8 && (unsigned)(hi >> 24 & 0xffu) == tag8 && (unsigned)(hi >> 56) == tag8;
                    }
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                    // the exit every poll loop of a resident kernel needs: records that never come (the launch is leaving)
                    if ((pass & 3u) == 3u) {
                        unsigned long long q = 0;
                        if (lane == 0) q = __hip_atomic_load(&a.link.ctl->quit_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__builtin_amdgcn_readfirstlane((int)(q == a.link.gen))) { if (lane == 0) s_quit = 1; break; }
                    }
                }
                if (tid < G) {
                    const unsigned d0 = (unsigned)lo & 0xffffffu, d1 = (unsigned)(lo >> 32) & 0xffffffu, d2 = (unsigned)hi & 0xffffffu, d3 = (unsigned)(hi >> 32) & 0xffffffu;
                    cb = (int)(d0 & 0x1fffu);
                    crel = (int)(((d0 >> 13) << 8) | (d3 >> 16));
                    ckey = (u64)d1 | ((u64)d2 << 24) | ((u64)(d3 & 0xffffu) << 48);
                }
                incl = wave_prefix_sum_i32(cb);
                if (lane == 63) s_cnt[wib] = incl;
            }
            __syncthreads(); // the wavefront totals of the counts (and s_quit) are in LDS
#ifndef SCG_FINE
            SCG_STAMP(5);
#endif
            if (s_quit) continue; // block-uniform; no answer: the host's wait finds the launch drained
            int run = 0;
            u64 fkey = SC_KEY_EMPTY; // this lane's candidate for the answer (key, position in the list)
            int fidx = 0x7fffffff;
            {
                int before = 0;
                for (int i = 0; i < npw; i++) {
                    if (i < wib) before += s_cnt[i];
                    run += s_cnt[i];
                }
                if (window > 0 && wib < npw && tid < G && cb > 0) {
                    const int through = before + incl; // real candidates up to and including workgroup tid
                    if (through <= window) { fkey = ckey; fidx = tid * cpw * (64 * NW) + crel; }
                    else if (through - cb < window) { s_cut = tid; s_before = through - cb; }
                }
            }
            __syncthreads(); // the cut workgroup is known
            const int n_scored = window <= 0 ? 0 : (run < window ? run : window);
            const int cut = s_cut;
            if (cut == c) { // the window cuts the finisher's own chunks, as the host expected: the keys are still in the lanes
                const int need = window - s_before;
#pragma unroll
                for (int j = 0; j < MAXC; j++)
                    if (ed[j].active && my_rank[j] <= need && (my_key[j] < fkey || (my_key[j] == fkey && k0 + j * (64 * NW) < fidx))) { fkey = my_key[j]; fidx = k0 + j * (64 * NW); }
            } else if (cut >= 0) { // another workgroup's: scored again here (same state, same table), up to the window
                const int need = window - s_before;
                int done = 0; // real candidates of the chunks already walked
                for (int j = 0; j < cpw && done < need; j++) { // block-uniform
                    const int kk = (cut * cpw + j) * (64 * NW) + tid;
                    int ty = -1, xa = 0, xb = 0;
                    if constexpr (GEN) sched_gen_exhaustive_lds(s_req + SCG_STATE_AT, S, E, kk, ty, xa, xb);
                    else if (kk < nm) { ty = srv_sys_load(&stream[kk].type); xa = srv_sys_load(&stream[kk].a); xb = srv_sys_load(&stream[kk].b); }
                    if (kk >= nm) ty = -1;
                    const SchedEdit e2 = sched_edit_of_lds(s_req + SCG_STATE_AT, S, E, ty, xa, xb);
                    const int hb2 = sched_hbits_of<M>(in, e2, sched_hol_word<M>(in, e2.new1, hol), sched_hol_word<M>(in, e2.new2, hol));
                    const u64 bal = __ballot(e2.active);
                    __syncthreads(); // s_cnt is reused
                    if (lane == 0) s_cnt[wib] = __popcll(bal);
                    __syncthreads();
                    int before = done;
                    for (int i = 0; i < NW; i++) {
                        if (i < wib) before += s_cnt[i];
                        done += s_cnt[i];
                    }
                    const int rank = before + __popcll(bal & ((2ull << lane) - 1)); // 1-based among the workgroup's real candidates
                    if (e2.active && rank <= need) {
                        int hard, soft;
                        const u64 key2 = sched_delta_key<M>(T, U, D, e_lane, e2, hb2, hard, soft, emp);
                        if (key2 < fkey || (key2 == fkey && kk < fidx)) { fkey = key2; fidx = kk; }
                    }
                }
            }
            // the minimum (key, first position) over the lanes' candidates: the wavefronts' pairs through LDS, one barrier
            {
                const u64 wk = wave_min_u64(fkey);
                const unsigned wr = wave_min_u32(fkey == wk && wk != SC_KEY_EMPTY ? (unsigned)fidx : ~0u);
                if (lane == 0) { s_wkey[wib] = wk; s_wrel[wib] = wr; }
            }
            __syncthreads();
            SCG_STAMP(6);
            if (wib == 0) { // the answer: d0 has-move, d1-2 n_scored, d3-4 best_index, d5 hard, d6 soft, d7 the cut workgroup + 1 (0: none)
                const u64 k8 = lane < NW ? s_wkey[lane] : SC_KEY_EMPTY;
                const u64 akey = wave_min8_u64(k8);
                const unsigned aidx = wave_min8_u32(lane < NW && k8 == akey ? s_wrel[lane] : ~0u);
                const bool has = akey != SC_KEY_EMPTY;
                unsigned pay = 0;
                if (lane == 0) pay = has ? 1u : 0u;
                else if (lane == 1) pay = has ? (unsigned)n_scored & 0xffffu : 0u;
                else if (lane == 2) pay = has ? (unsigned)n_scored >> 16 : 0u;
                else if (lane == 3) pay = has ? aidx & 0xffffu : 0u;
                else if (lane == 4) pay = has ? aidx >> 16 : 0u;
                else if (lane == 5) pay = has ? (unsigned)(akey >> 54) : 0u;
                else if (lane == 6) pay = has ? (unsigned)((akey >> 46) & 0xffu) : 0u;
                else if (lane == 7) pay = (unsigned)(cut + 1);
#ifdef LS_NQ_STAMPS
                gst[7] = wall_clock64();
                if (lane >= 8 && lane < 15) { const unsigned long long dt = gst[lane - 7] - gst[0]; pay = dt > 0xffffu ? 0xffffu : (unsigned)dt; }
                if (lane >= 8 && lane < 15) __hip_atomic_store(resp + lane, ((unsigned)s_ctl[1] << 16) | (pay & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
                if (lane < 8) __hip_atomic_store(resp + lane, ((unsigned)s_ctl[1] << 16) | (pay & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (lane == 0) __hip_atomic_store(&a.link.ctl->last_activity, wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (emp && wib == 0 && lane < D && e_lane >= 0) emp[e_lane] = 0; // the table is all zero again (every scorer of this request is past a barrier)
        if (c == 0 && wib == 0) poll.t_served = wall_clock64();
    }
    if (tid == 0) srv_block_leaves(a.link);
}

// ------------------------------------------------------------------------------------------------------------ host side
struct SchedGrid {
    SrvHost h;                   // 4 lines per sub-grid
    ls_ctx *owner = nullptr;     // the context whose instance the launch carries
    bool gen = false;            // the exhaustive proposer's candidates, generated in the kernel
    int n_sub = 0, G = 0, cpw = 0, maxc = 0;
    long long stream_cap = 0;    // moves per sub-grid = G * cpw * 512
    SchedGridCtl *g = nullptr;
    unsigned *go = nullptr;      // [n_sub][64]: fine-grained device memory the host writes through the BAR (go_direct), else plain device memory
    bool go_direct = false;
    DevBuf rec;
    unsigned long long ready_total = 0;
    struct alignas(64) Sub { // (a sub-grid is one host thread's: no line shared with a neighbour)
        unsigned stream_id = 0;
        const ls_ctx *stream_ctx = nullptr;
        unsigned long long stream_gen = 0;
        int fin = 0;                      // the workgroup the window cut in the last answer: the next request's finisher
        unsigned long long requests = 0, fin_hits = 0;
        double t_wait = 0, t_call = 0;    // LS_AMD_TRACE
        unsigned long long dev_ticks[7] = {0, 0, 0, 0, 0, 0, 0};
    };
    std::vector<Sub> subs;
    ls_sched_move *stream(int s) const { return reinterpret_cast<ls_sched_move *>(h.extra()) + (size_t)s * stream_cap; }
};

template <int MAXC> static const void *grid_fn_c(bool m32, bool gen)
{
    return gen ? (m32 ? reinterpret_cast<const void *>(&sched_grid_server_kernel<unsigned, SCS_NW, true, MAXC>)
                      : reinterpret_cast<const void *>(&sched_grid_server_kernel<u64, SCS_NW, true, MAXC>))
               : (m32 ? reinterpret_cast<const void *>(&sched_grid_server_kernel<unsigned, SCS_NW, false, MAXC>)
                      : reinterpret_cast<const void *>(&sched_grid_server_kernel<u64, SCS_NW, false, MAXC>));
}
static const void *grid_fn(bool m32, bool gen, int maxc)
{
    return maxc <= 1 ? grid_fn_c<1>(m32, gen) : (maxc <= 2 ? grid_fn_c<2>(m32, gen) : (maxc <= 4 ? grid_fn_c<4>(m32, gen) : (maxc <= 6 ? grid_fn_c<6>(m32, gen) : grid_fn_c<SCG_MAXC>(m32, gen))));
}

// Geometry of a grid of n_sub sub-grids for lists of `total` candidates: as few chunks per workgroup as leave every workgroup
// of the launch its own CU (they wait for each other, and the kernel's registers allow one 512-thread workgroup per CU), a few
// CUs left to whatever else must run (another context's one-workgroup server, copy kernels).  false: it does not fit.
static bool grid_geometry(const ls_ctx *ctx, int n_sub, long long total, int *G_out, int *cpw_out)
{
    const long long chunks = (total + SCS_CHUNK - 1) / SCS_CHUNK;
    if (chunks < 1 || n_sub < 1) return false;
    const int spare = ctx->n_cus >= 64 ? 4 : 0;
    const long long g_max = (ctx->n_cus - spare) / n_sub;
    if (g_max < 1) return false;
    long long cpw = (chunks + g_max - 1) / g_max;
    if (cpw < 1) cpw = 1;
    if (cpw > SCG_MAXC) return false;
    const long long G = (chunks + cpw - 1) / cpw;
    if (G > 4 * 64) return false; // one record per lane of at most four polling wavefronts
    *G_out = (int)G; *cpw_out = (int)cpw;
    return true;
}

bool sched_grid_fits(const ls_ctx *ctx, int n_sub, bool gen, long long n_moves)
{
    if (!(ctx->sched_server_on && ctx->sched_delta && ctx->sched.valid && ctx->sched.n_employees <= 32768 &&
          ctx->sched.n_days_stored <= 64 - SCG_STATE_AT))
        return false;
    const long long total = gen ? (long long)ctx->sched.n_days_stored * ctx->sched.n_employees : n_moves;
    int G = 0, cpw = 0;
    return total > SCS_CHUNK && total < (1ll << 24) && grid_geometry(ctx, n_sub, total, &G, &cpw);
}

void sched_grid_destroy(SchedGrid *s)
{
    if (!s) return;
    if (getenv("LS_AMD_TRACE")) {
        for (size_t i = 0; i < s->subs.size(); i++) {
            const SchedGrid::Sub &b = s->subs[i];
            if (!b.requests) continue;
            fprintf(stderr, "scheduling grid (%s, sub-grid %zu of %d: %d workgroups x %d chunks): %llu requests, answer wait %.2f us per request (whole call %.2f us), "
                            "%llu launches, the window cut the finisher's chunks in %llu\n", s->gen ? "exhaustive proposer" : "resident stream", i, s->n_sub, s->G, s->cpw,
                    b.requests, b.t_wait / b.requests * 1e6, b.t_call / b.requests * 1e6, s->h.relaunches.load(), b.fin_hits);
            if (b.dev_ticks[6])
                fprintf(stderr, "  the finisher, after it saw the request: edit + holiday bits in %.2f us, table built %.2f, chunks scored %.2f, "
                                "own record out %.2f, all records in %.2f, window placed %.2f, answer %.2f\n",
                        b.dev_ticks[0] * 0.01 / b.requests, b.dev_ticks[1] * 0.01 / b.requests, b.dev_ticks[2] * 0.01 / b.requests,
                        b.dev_ticks[3] * 0.01 / b.requests, b.dev_ticks[4] * 0.01 / b.requests, b.dev_ticks[5] * 0.01 / b.requests,
                        b.dev_ticks[6] * 0.01 / b.requests);
        }
    }
    s->h.shutdown();
    if (s->g) (void)hipFree(s->g);
    if (s->go) (void)hipFree(s->go);
    s->rec.release();
    delete s;
}

// A grid of n_sub sub-grids for lists of up to `capacity` moves each (gen: the exhaustive proposer's S x E candidates).
int sched_grid_create(ls_ctx *ctx, int n_sub, bool gen, long long capacity, SchedGrid **out)
{
    *out = nullptr;
    if (!ctx->sched.valid) return set_err(ctx, LS_ERR_STATE, "ls_sched_instance_upload has not been called");
    const long long total = gen ? (long long)ctx->sched.n_days_stored * ctx->sched.n_employees : capacity;
    int G = 0, cpw = 0;
    if (!sched_grid_fits(ctx, n_sub, gen, capacity) || !grid_geometry(ctx, n_sub, total, &G, &cpw))
        return set_err(ctx, LS_ERR_UNSUPPORTED, "scheduling grid: %d sub-grids of %lld candidates do not fit the device's %d CUs", n_sub, total, ctx->n_cus);
    LS_HIP(ctx, hipSetDevice(ctx->device));
    SchedGrid *s = new (std::nothrow) SchedGrid();
    if (!s) return set_err(ctx, LS_ERR_NOMEM, "out of host memory");
    s->owner = ctx; s->gen = gen; s->n_sub = n_sub; s->G = G; s->cpw = cpw;
    s->maxc = cpw <= 1 ? 1 : (cpw <= 2 ? 2 : (cpw <= 4 ? 4 : (cpw <= 6 ? 6 : SCG_MAXC)));
    s->stream_cap = (long long)G * cpw * SCS_CHUNK;
    s->subs.resize((size_t)n_sub);
    int rc = s->h.init(ctx, "scheduling grid server", 4 * n_sub, gen ? 0 : (size_t)n_sub * (size_t)s->stream_cap * sizeof(ls_sched_move));
    if (rc != LS_OK) { delete s; return rc; }
    s->h.lines_per_msg = 4;
    s->h.relaunch_limit = 2;
    // everything the launch needs exists before the first launch can be asked for; a failure leaves nothing half-made behind
    const size_t go_bytes = (size_t)n_sub * 256;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&s->g), sizeof(SchedGridCtl));
    if (e == hipSuccess) {
        // The go records in fine-grained device memory when the host can write there (large-BAR platform): every workgroup then
        // polls the request itself -- no PCIe read by a master, no relay.  Else plain device memory, filled by the masters.
        int large_bar = 0;
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, ctx->device) == hipSuccess && large_bar &&
            hipExtMallocWithFlags(reinterpret_cast<void **>(&s->go), go_bytes, hipDeviceMallocFinegrained) == hipSuccess) {
            s->go_direct = true;
        } else {
            (void)hipGetLastError();
            s->go = nullptr;
            e = hipMalloc(reinterpret_cast<void **>(&s->go), go_bytes);
        }
    }
    if (e == hipSuccess) e = s->rec.reserve(align256((size_t)n_sub * G * 16));
    if (e == hipSuccess) {
        // tag 0 everywhere: an uninitialised record or go line must not look like a message of the current request
        hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(64), 0, s->h.stream, reinterpret_cast<int *>(s->g), 0, (long long)(sizeof(SchedGridCtl) / 4));
        hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(256), 0, s->h.stream, reinterpret_cast<int *>(s->go), 0, (long long)(go_bytes / 4));
        hipLaunchKernelGGL(fill_i32_kernel, dim3(8), dim3(256), 0, s->h.stream, s->rec.as<int>(), 0, (long long)(s->rec.cap / 4));
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s->h.stream);
    }
    if (e != hipSuccess) {
        rc = set_err(ctx, LS_ERR_HIP, "scheduling grid: set-up failed: %s", hipGetErrorString(e));
        if (s->g) (void)hipFree(s->g);
        if (s->go) (void)hipFree(s->go);
        s->rec.release();
        s->h.release();
        delete s;
        return rc;
    }
    if (s->go_direct) s->h.req_dev = s->go; // the host's request lines ARE the go records
    s->h.launch = [s](const SrvLink &L) {
        ls_ctx *ctx = s->owner;
        SchedGridArgs a;
        a.link = L;
        a.in = make_inst(ctx);
        a.streams = s->gen ? nullptr : s->stream(0);
        a.stream_cap = s->stream_cap;
        a.gen_total = (long long)ctx->sched.n_days_stored * ctx->sched.n_employees;
        a.g = s->g;
        a.go = s->go;
        s->ready_total += (unsigned long long)s->n_sub * s->G;
        a.ready_target = s->ready_total;
        a.rec = s->rec.as<unsigned>();
        a.G = s->G; a.cpw = s->cpw;
        const bool m32 = sched_m32(ctx);
        const void *fn = grid_fn(m32, s->gen, s->maxc);
        // the employee -> scored-days table and the holiday words (one workgroup per CU: it has the LDS to itself)
        const size_t dyn = sched_lds_tables_bytes(ctx, fn, m32, (size_t)144 << 10, (size_t)96 << 10, &a.hol_entries);
        a.emp_entries = dyn ? ctx->sched.n_employees : 0;
        void *args[] = {&a};
        return hipLaunchKernel(fn, dim3((unsigned)(s->n_sub * s->G)), dim3(64 * SCS_NW), args, dyn, s->h.stream);
    };
    *out = s;
    return LS_OK;
}

int sched_grid_subs(const SchedGrid *s) { return s->n_sub; }
bool sched_grid_is_gen(const SchedGrid *s) { return s->gen; }
long long sched_grid_capacity(const SchedGrid *s) { return s->gen ? 0 : s->stream_cap; }
void sched_grid_pause(SchedGrid *s) { s->h.pause(); }
void sched_grid_counters(const SchedGrid *s, unsigned long long out[3])
{
    out[0] = s->h.requests_total(); out[1] = s->h.relaunches.load(); out[2] = s->h.slow_waits.load();
}

static int grid_check_state(ls_ctx *ctx, const int32_t *state)
{
    const int S = ctx->sched.n_days_stored, E = ctx->sched.n_employees;
    for (int d = 0; d < S; d++)
        if (state[d] < 0 || state[d] >= E)
            return set_err(ctx, LS_ERR_INVALID, "state entry %d = %d is not an employee index in [0, %d)", d, state[d], E);
    return LS_OK;
}

// ls_sched_score of one state by the sub-grid's master workgroup
int sched_grid_score(ls_ctx *ctx, SchedGrid *s, int sub, const int32_t *state, int64_t *hard, int64_t *soft)
{
    if (sub < 0 || sub >= s->n_sub) return set_err(ctx, LS_ERR_INVALID, "scheduling grid: sub-grid %d outside [0, %d)", sub, s->n_sub);
    int rc = grid_check_state(ctx, state);
    if (rc != LS_OK) return rc;
    const int S = ctx->sched.n_days_stored;
    unsigned pay[64] = {0};
    pay[0] = SCG_F_SCORE;
    for (int d = 0; d < S; d++) pay[SCG_STATE_AT + d] = (unsigned)state[d] & 0xffffu;
    rc = s->h.ensure_running(ctx);
    if (rc != LS_OK) return rc;
    s->h.post_lines(4 * sub, pay, 4);
    unsigned r[8];
    rc = s->h.wait(ctx, 4 * sub, r);
    if (rc != LS_OK) return rc; // (SRV_RC_GIVE_UP: the caller takes the launched route)
    *hard = r[5];
    *soft = r[6];
    return LS_OK;
}

// lines :315-325 for one state on sub-grid `sub`: the first n_moves entries of the context's resident list (gen: the exhaustive
// proposer's S x E candidates).  SRV_RC_GIVE_UP: the launch cannot be resident now -- the caller takes the launched pipeline.
int sched_grid_eval(ls_ctx *ctx, SchedGrid *s, int sub, const int32_t *state, long long n_moves, long long window, ls_sched_result *out)
{
    static const bool trace = getenv("LS_AMD_TRACE") != nullptr;
    const auto T0 = trace ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    if (sub < 0 || sub >= s->n_sub) return set_err(ctx, LS_ERR_INVALID, "scheduling grid: sub-grid %d outside [0, %d)", sub, s->n_sub);
    int rc = grid_check_state(ctx, state);
    if (rc != LS_OK) return rc;
    SchedGrid::Sub &b = s->subs[(size_t)sub];
    const int S = ctx->sched.n_days_stored;
    if (!s->gen && n_moves > s->stream_cap) return set_err(ctx, LS_ERR_UNSUPPORTED, "scheduling grid: %lld moves exceed the sub-grid's %lld", n_moves, s->stream_cap);
    if (!s->gen && (b.stream_ctx != ctx || b.stream_gen != ctx->sc_resident_gen)) { // a new list (a new restart on this sub-grid): into its pinned block, under a new id
        // (the prefix the sub-grid can hold: the caller's list may be longer than what its requests use)
        const size_t n = ctx->sc_resident_host.size() < (size_t)s->stream_cap ? ctx->sc_resident_host.size() : (size_t)s->stream_cap;
        if (n) memcpy(s->stream(sub), ctx->sc_resident_host.data(), n * sizeof(ls_sched_move));
        b.stream_ctx = ctx;
        b.stream_gen = ctx->sc_resident_gen;
        b.stream_id = b.stream_id >= 0xfffeu ? 1u : b.stream_id + 1u;
        b.fin = 0;
    }
    ctx->steps_done++;
    // the request: ONE message of four lines -- header dwords 0..15, the state from dword 16 on -- every dword with the same
    // new tag; the master accepts a snapshot only when all 64 tags agree, so the order of these stores does not matter
    rc = s->h.ensure_running(ctx);
    if (rc != LS_OK) return rc;
    {
        const unsigned w = (unsigned)(window > 0x7fffffffll ? 0x7fffffffll : window);
        unsigned pay[64] = {0};
        pay[0] = SCG_F_EVAL; pay[1] = (unsigned)n_moves & 0xffffu; pay[2] = (unsigned)n_moves >> 16; pay[3] = w & 0xffffu; pay[4] = w >> 16;
        pay[5] = (unsigned)b.fin; // the finisher: the workgroup the window cut last time (its lanes still hold the keys the cut needs)
        pay[6] = b.stream_id;
        for (int d = 0; d < S; d++) pay[SCG_STATE_AT + d] = (unsigned)state[d] & 0xffffu;
        s->h.post_lines(4 * sub, pay, 4);
        b.requests++;
    }
    const auto T1 = trace ? std::chrono::steady_clock::now() : T0;
    unsigned r[8];
    rc = s->h.wait(ctx, 4 * sub, r);
    if (rc != LS_OK) return rc; // SRV_RC_GIVE_UP: the workgroups could not all be resident (twice)
    if (trace) {
        const auto T2 = std::chrono::steady_clock::now();
        b.t_wait += std::chrono::duration<double>(T2 - T1).count();
        b.t_call += std::chrono::duration<double>(T2 - T0).count();
        const volatile unsigned *rl = s->h.resp(4 * sub);
        for (int d = 0; d < 7; d++) b.dev_ticks[d] += (rl[8 + d] >> 16) == s->h.per[(size_t)(4 * sub)].tag ? rl[8 + d] & 0xffffu : 0u;
    }
    if (r[7] != 0) {
        if ((int)r[7] - 1 == b.fin) b.fin_hits++;
        b.fin = (int)r[7] - 1;
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
