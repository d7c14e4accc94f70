// sched_kernels.hip — gfx950 kernels for the employee-scheduling neighbourhood
// pipeline: local-search/src/local_search.rs:315-325 with the plug-in of
// examples/employee-scheduling/src/lib.rs (scorer :261-375, helpers :149-222,
// moves :464-479 / :526-550).
//
// One wavefront scores one candidate; lane d holds day d (<= 64 stored days).
// Every constraint term is a function of the day-equality matrix
//   eq[d] = bitmask over scored days q with employee(q) == employee(d)
// which a wavefront builds with one readlane + compare per day:
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
//   S3 / S4                max - min of popcount(eq) / popcount(eq & weekend)
//                          over the scored lanes                            lib.rs:345-365
// All terms are small integers (the reference holds them in f64; exact).
//
// Argmin by (hard, soft, date_to_employee vector), first of equals
// (local_search.rs:29-37; stable sort + first()).  Candidates are <= 2-position
// edits of one base vector, so the vector order is a closed form over the
// edits: per edit (class: lowers/none/raises, position key, new value).  Packed
// with the scores into one 64-bit key per candidate; a second kernel takes the
// minimum (key, index).
#include "ls_common.hpp"

namespace lsamd {

typedef unsigned long long u64;

constexpr u64 SC_KEY_EMPTY = ~0ull;
constexpr int SC_BIG = 1 << 20;

struct SchedInst {
    int D;             // scored days
    int S;             // stored days
    int start_weekday; // Monday = 0
    int n_employees;
    const u64 *holiday_bits;
};

// per-lane constants of the instance
struct SchedLane {
    int d;
    bool scored;
    u64 scored_mask, weekend_mask;
    u64 mask2;  // bit d+1 if d+1 < D
    u64 h3mask; // bits compared with day d inside a weekend-started 9-window
    u64 wdmask; // scored days on day d's weekday, 0 on weekends / unscored lanes
    int gsize;
    bool s2_lead; // first lane of a Mon-Fri weekday group
};

__device__ __forceinline__ bool sc_weekend(int start_weekday, int q) { return (start_weekday + q) % 7 >= 5; }

__device__ __forceinline__ SchedLane sched_lane_init(const SchedInst &in, int lane)
{
    SchedLane L;
    const int D = in.D;
    L.d = lane;
    L.scored = lane < D;
    L.scored_mask = D >= 64 ? ~0ull : ((1ull << D) - 1);
    u64 wm = 0;
    for (int q = 0; q < 64; q++)
        if (q < D && sc_weekend(in.start_weekday, q)) wm |= 1ull << q;
    L.weekend_mask = wm;
    L.mask2 = (L.scored && lane + 1 < D) ? (1ull << (lane + 1)) : 0;
    u64 h3 = 0;
    if (L.scored) {
        // role date1: windows(9) starting at d with d, d+1 both weekend days; pairs (d,d+7), (d,d+8)
        if (lane + 8 < D && sc_weekend(in.start_weekday, lane) && sc_weekend(in.start_weekday, lane + 1))
            h3 |= (1ull << (lane + 7)) | (1ull << (lane + 8));
        // role date2: window starts at d-1; pairs (d,d+6), (d,d+7)
        if (lane >= 1 && lane - 1 + 8 < D && sc_weekend(in.start_weekday, lane - 1) &&
            sc_weekend(in.start_weekday, lane))
            h3 |= (1ull << (lane + 6)) | (1ull << (lane + 7));
    }
    L.h3mask = h3;
    const bool weekday = (in.start_weekday + lane) % 7 < 5;
    const u64 every7 = 0x8102040810204081ull; // bits 0, 7, 14, ..., 63
    L.wdmask = (L.scored && weekday) ? ((every7 << (lane % 7)) & L.scored_mask) : 0;
    L.gsize = __popcll(L.wdmask);
    L.s2_lead = L.scored && weekday && lane < 7;
    return L;
}

__device__ __forceinline__ int wave_max_dpp(int x)
{
#define LS_STEP(ctrl, rm)                                                   \
    {                                                                       \
        int o = __builtin_amdgcn_update_dpp(x, x, ctrl, rm, 0xf, false);    \
        x = o > x ? o : x;                                                  \
    }
    LS_STEP(LS_DPP_QUAD_1032, 0xf)
    LS_STEP(LS_DPP_QUAD_2301, 0xf)
    LS_STEP(LS_DPP_ROW_HALF_MIRROR, 0xf)
    LS_STEP(LS_DPP_ROW_MIRROR, 0xf)
    LS_STEP(LS_DPP_ROW_BCAST15, 0xa)
    LS_STEP(LS_DPP_ROW_BCAST31, 0xc)
#undef LS_STEP
    return __builtin_amdgcn_readlane(x, 63);
}

struct SchedScore {
    int hard, soft;
    int t[8];
};

// Scores the vector held one entry per lane in `e` (lanes >= S hold distinct negative sentinels).
template <bool WANT_TERMS>
__device__ __forceinline__ SchedScore sched_score_wave(const SchedInst &in, const SchedLane &L, int e)
{
    const int D = in.D;
    // day-equality matrix, one row per lane
    u64 eq = 0;
#pragma unroll
    for (int q = 0; q < 64; q++) {
        if (q < D) {
            const int eq_q = __builtin_amdgcn_readlane(e, q);
            if (e == eq_q) eq |= 1ull << q;
        }
    }
    eq &= L.scored_mask;
    const int p = L.d;
    int h1 = 0, h2 = 0, h3 = 0, h4 = 0, s1 = 0, s2 = 0;
    int tot_max = 0, tot_min = 0, wk_max = 0, wk_min = 0; // as "max of x" and "max of 64 - x"
    if (L.scored) {
        h1 = (unsigned)e < (unsigned)in.n_employees ? (int)((in.holiday_bits[e] >> p) & 1ull) : 0;
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
        const int tot = __popcll(eq);
        const int wk = __popcll(eq & L.weekend_mask);
        tot_max = tot; tot_min = 64 - tot;
        wk_max = wk; wk_min = 64 - wk;
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
    if (L.s2_lead && cnt < L.gsize) s2 = cnt;

    SchedScore r;
    // hard <= 316, soft (sum part) <= 141: 16-bit fields
    const unsigned hs = (unsigned)wave_sum_i32((h1 + h2 + h3 + h4) | ((s1 + s2) << 16));
    // lanes outside the scored range contribute 0 to every max
    const int tmax = wave_max_dpp(tot_max), tmin = 64 - wave_max_dpp(tot_min);
    const int wmax = wave_max_dpp(wk_max), wmin = 64 - wave_max_dpp(wk_min);
    const int s3 = tmax - tmin, s4 = wmax - wmin;
    r.hard = (int)(hs & 0xffffu);
    r.soft = (int)(hs >> 16) + s3 + s4;
    if (WANT_TERMS) {
        r.t[0] = wave_sum_i32(h1); r.t[1] = wave_sum_i32(h2); r.t[2] = wave_sum_i32(h3);
        r.t[3] = wave_sum_i32(h4); r.t[4] = wave_sum_i32(s1); r.t[5] = wave_sum_i32(s2);
        r.t[6] = s3; r.t[7] = s4;
    }
    return r;
}

__global__ __launch_bounds__(256) void sched_score_kernel(SchedInst in, int n_states, const int *states,
                                                          long long *hard, long long *soft, long long *terms)
{
    const int lane = threadIdx.x & 63;
    const int gw = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
    const int n_waves = (gridDim.x * 256) >> 6;
    const SchedLane L = sched_lane_init(in, lane);
    for (int s = gw; s < n_states; s += n_waves) {
        const int e = lane < in.S ? states[(long long)s * in.S + lane] : -1 - lane;
        const SchedScore r = sched_score_wave<true>(in, L, e);
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
    SchedInst in;
    int n_restarts;
    const int *states;          // [n_restarts][S]
    const ls_sched_move *moves; // [n_restarts][max_moves]
    const long long *n_moves;   // [n_restarts]
    long long max_moves;
    long long window;
    long long *k_end;    // [n_restarts]  first index not scored
    long long *n_scored; // [n_restarts]
    u64 *keys;           // [n_restarts][max_moves]
    long long *hard_out, *soft_out; // nullable [n_restarts][max_moves]
};

__device__ __forceinline__ bool sched_move_ok(const ls_sched_move &m, int S, int E)
{
    if (m.type == 0) return (unsigned)m.a < (unsigned)S && (unsigned)m.b < (unsigned)E;
    if (m.type == 1) return (unsigned)m.a < (unsigned)S && (unsigned)m.b < (unsigned)S;
    return false;
}

// Pass 1: where does take(window) stop?  Tabu (no-op) candidates do not count
// (local_search.rs:319-321).  One block per restart.
__global__ __launch_bounds__(256) void sched_window_kernel(SchedEvalArgs a)
{
    __shared__ int s_state[64];
    __shared__ int s_wcnt[4];
    __shared__ long long s_found;
    const int r = blockIdx.x;
    const int S = a.in.S;
    if (threadIdx.x < 64) s_state[threadIdx.x] = threadIdx.x < S ? a.states[(long long)r * S + threadIdx.x] : -1;
    if (threadIdx.x == 0) s_found = -1;
    __syncthreads();
    long long nm = a.n_moves[r];
    nm = nm < 0 ? 0 : (nm > a.max_moves ? a.max_moves : nm);
    const ls_sched_move *mv = a.moves + (long long)r * a.max_moves;
    long long run = 0;
    long long k_end = nm;
    if (a.window <= 0) { k_end = 0; }
    else {
        for (long long t = 0; t < nm; t += 256) {
            const long long k = t + threadIdx.x;
            bool flag = false;
            if (k < nm) {
                const ls_sched_move m = mv[k];
                if (sched_move_ok(m, S, a.in.n_employees))
                    flag = m.type == 0 ? (s_state[m.a] != m.b) : (s_state[m.a] != s_state[m.b]);
            }
            const u64 bal = __ballot(flag);
            const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
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

// Pass 2: one wavefront per candidate.
__global__ __launch_bounds__(256) void sched_eval_kernel(SchedEvalArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = blockIdx.x % a.n_restarts;
    const int wchunk = (blockIdx.x / a.n_restarts) * 4 + wib;
    const int n_wchunks = (gridDim.x / a.n_restarts) * 4;
    const int S = a.in.S;
    const SchedLane L = sched_lane_init(a.in, lane);
    const int base = lane < S ? a.states[(long long)r * S + lane] : -1 - lane;
    const long long k_end = a.k_end[r];
    const ls_sched_move *mv = a.moves + (long long)r * a.max_moves;
    u64 *keys = a.keys + (long long)r * a.max_moves;
    const long long k0 = a.max_moves * wchunk / n_wchunks;
    const long long k1 = a.max_moves * (wchunk + 1) / n_wchunks;
    for (long long k = k0; k < k1; k++) {
        u64 key = SC_KEY_EMPTY;
        if (k < k_end) {
            ls_sched_move m = mv[k];
            m.type = __builtin_amdgcn_readfirstlane(m.type);
            m.a = __builtin_amdgcn_readfirstlane(m.a);
            m.b = __builtin_amdgcn_readfirstlane(m.b);
            if (sched_move_ok(m, S, a.in.n_employees)) {
                const int ea = __builtin_amdgcn_readlane(base, m.a);
                int na, nb, pb;
                if (m.type == 0) { na = m.b; pb = m.a; nb = m.b; }
                else { const int eb = __builtin_amdgcn_readlane(base, m.b); na = eb; pb = m.b; nb = ea; }
                const int eb0 = __builtin_amdgcn_readlane(base, pb);
                if (na != ea || nb != eb0) { // not tabu
                    int e = base;
                    if (lane == m.a) e = na;
                    if (lane == pb) e = nb;
                    const SchedScore sc = sched_score_wave<false>(a.in, L, e);
                    u64 e1, e2 = (u64)1 << 21; // "no second edit"
                    if (m.type == 0) e1 = sc_edit_key(m.a, na, ea);
                    else {
                        const int p1 = m.a < m.b ? m.a : m.b, p2 = m.a < m.b ? m.b : m.a;
                        const int v1 = m.a < m.b ? na : nb, b1 = m.a < m.b ? ea : eb0;
                        const int v2 = m.a < m.b ? nb : na, b2 = m.a < m.b ? eb0 : ea;
                        e1 = sc_edit_key(p1, v1, b1);
                        e2 = sc_edit_key(p2, v2, b2);
                    }
                    key = ((u64)(unsigned)sc.hard << 54) | ((u64)(unsigned)sc.soft << 46) | (e1 << 23) | e2;
                    if (lane == 0) {
                        if (a.hard_out) a.hard_out[(long long)r * a.max_moves + k] = sc.hard;
                        if (a.soft_out) a.soft_out[(long long)r * a.max_moves + k] = sc.soft;
                    }
                }
            }
        }
        if (lane == 0) keys[k] = key;
    }
}

// Pass 3: minimum (key, index) per restart.
__global__ __launch_bounds__(256) void sched_reduce_kernel(SchedEvalArgs a, ls_sched_result *results)
{
    __shared__ u64 s_key[4];
    __shared__ long long s_idx[4];
    const int r = blockIdx.x;
    long long nm = a.n_moves[r];
    nm = nm < 0 ? 0 : (nm > a.max_moves ? a.max_moves : nm);
    const u64 *keys = a.keys + (long long)r * a.max_moves;
    u64 bk = SC_KEY_EMPTY;
    long long bi = -1;
    for (long long k = threadIdx.x; k < nm; k += 256) {
        const u64 key = keys[k];
        if (key < bk) { bk = key; bi = k; } // ascending k per thread: first of equals kept
    }
    // wave reduce on (key, idx)
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
        ls_sched_result out;
        out.n_scored = a.n_scored[r];
        if (bk == SC_KEY_EMPTY) { out.best_index = -1; out.best_hard = 0; out.best_soft = 0; out.n_scored = 0; }
        else {
            out.best_index = bi;
            out.best_hard = (long long)(bk >> 54);
            out.best_soft = (long long)((bk >> 46) & 0xffu);
        }
        results[r] = out;
    }
}

// ------------------------------------------------------------------ launchers
static SchedInst make_inst(const ls_ctx *ctx)
{
    SchedInst in;
    in.D = ctx->sched.n_days_scored;
    in.S = ctx->sched.n_days_stored;
    in.start_weekday = ctx->sched.start_weekday;
    in.n_employees = ctx->sched.n_employees;
    in.holiday_bits = ctx->sched.holiday_bits.as<u64>();
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

int sched_eval_batch_dev(ls_ctx *ctx, int n_restarts, const int *states_d, const ls_sched_move *moves_d,
                         const long long *n_moves_d, long long max_moves, long long window,
                         ls_sched_result *results_d, long long *hard_out_d, long long *soft_out_d)
{
    if (ctx->sched.n_employees > 32768)
        return set_err(ctx, LS_ERR_UNSUPPORTED, "n_employees = %d exceeds the 15-bit argmin key field (32768)",
                       ctx->sched.n_employees);
    LS_RESERVE(ctx, ctx->sc_keys, (size_t)n_restarts * (max_moves > 0 ? max_moves : 1) * 8 + (size_t)n_restarts * 16);
    SchedEvalArgs a;
    a.in = make_inst(ctx);
    a.n_restarts = n_restarts;
    a.states = states_d; a.moves = moves_d; a.n_moves = n_moves_d;
    a.max_moves = max_moves; a.window = window;
    // layout of sc_keys: k_end[n_restarts] | n_scored[n_restarts] | keys[n_restarts][max_moves]
    a.k_end = ctx->sc_keys.as<long long>();
    a.n_scored = a.k_end + n_restarts;
    a.keys = reinterpret_cast<u64 *>(a.n_scored + n_restarts);
    a.hard_out = hard_out_d; a.soft_out = soft_out_d;
    const long long total = (long long)n_restarts * max_moves;
    if (hard_out_d && total > 0) {
        hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)),
                           dim3(256), 0, ctx->stream, hard_out_d, (long long)INT64_MIN, total);
        LS_HIP(ctx, hipGetLastError());
    }
    if (soft_out_d && total > 0) {
        hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)),
                           dim3(256), 0, ctx->stream, soft_out_d, (long long)INT64_MIN, total);
        LS_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(sched_window_kernel, dim3(n_restarts), dim3(256), 0, ctx->stream, a);
    LS_HIP(ctx, hipGetLastError());
    if (max_moves > 0) {
        long long want_waves = (max_moves + 3) / 4; // >= 4 candidates per wavefront
        long long bpr = (want_waves + 3) / 4;
        const long long cap = ((long long)ctx->n_cus * 8) / n_restarts;
        if (bpr > cap) bpr = cap;
        if (bpr < 1) bpr = 1;
        hipLaunchKernelGGL(sched_eval_kernel, dim3((unsigned)(bpr * n_restarts)), dim3(256), 0, ctx->stream, a);
        LS_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(sched_reduce_kernel, dim3(n_restarts), dim3(256), 0, ctx->stream, a, results_d);
    LS_HIP(ctx, hipGetLastError());
    return LS_OK;
}

} // namespace lsamd
