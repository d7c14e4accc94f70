// nq_device.hpp — device helpers shared by the N-queens kernels (nq_kernels.hip: launched per call) and the resident LS-step
// server (nq_server.hip): the 64-bit argmin key, the window arithmetic, the packed-u16 occupancy tables and the paired
// wavefront-team column walk.  See nq_kernels.hip for the scoring identity and the closed form of the vector order.
#pragma once

#include "ls_common.hpp"

namespace lsamd {

constexpr int NQ_DELTA_BIAS = 1 << 17; // |delta| <= n + 2 <= 32770
constexpr unsigned long long NQ_KEY_EMPTY = ~0ull;

// key = delta+bias (18 bits) | up (1) | colkey (15) | v (15) | slot (15).  The slot (position of
// the column in the proposer's list) is last: it only separates identical candidates of a
// duplicated column, where the earlier one must win (stable sort + first()).
__device__ __forceinline__ unsigned long long nq_key_hi(int is_up, int c, int n, int slot)
{
    const unsigned colkey = is_up ? (unsigned)(n - 1 - c) : (unsigned)c;
    return ((unsigned long long)(unsigned)is_up << 45) | ((unsigned long long)colkey << 30) | (unsigned)slot;
}

__device__ __forceinline__ unsigned long long nq_key(int delta, unsigned long long hi, int v)
{
    return ((unsigned long long)(unsigned)(delta + NQ_DELTA_BIAS) << 46) | hi | ((unsigned long long)(unsigned)v << 15);
}

// Within one (column, direction) range only delta and v vary: a 32-bit running minimum of
// (delta + 2^16) << 15 | v  (|delta| <= n + 2 <= 32770, v < 2^15) orders exactly like the full key and is
// widened once per range instead of once per candidate.
constexpr unsigned NQ_KEY32_EMPTY = 0xffffffffu;
__device__ __forceinline__ unsigned nq_key32(int delta, int v) { return ((unsigned)(delta + 65536) << 15) | (unsigned)v; }
__device__ __forceinline__ unsigned long long nq_key_from32(unsigned k32, unsigned long long hi)
{
    return nq_key((int)(k32 >> 15) - 65536, hi, (int)(k32 & 0x7fffu));
}

// Unfiltered positions [0, nq_k_limit) are the only ones take(window) can reach: each column
// holds n - 1 real candidates and one no-op.
__device__ __forceinline__ long long nq_k_limit(long long K, int n, long long window)
{
    if (n <= 1 || window <= 0) return 0;
    const long long full = window / (n - 1), rem = window % (n - 1);
    const long long lim = full * n + (rem > 0 ? rem + 1 : 0);
    return lim < K ? lim : K;
}

__device__ __forceinline__ unsigned long long u64min(unsigned long long a, unsigned long long b)
{
    return b < a ? b : a;
}

// Sub-ranges of values v in [vs, ve) of one column that are real candidates:
// not the no-op v == rc, and inside the window.  rank(v) = base + v - (v > rc)
// must be < window  (tabu candidates do not consume window, local_search.rs:319-321).
struct NqRanges {
    int d0, d1; // "down" moves  v in [d0, d1),  v < rc
    int u0, u1; // "up" moves    v in [u0, u1),  v > rc
};

__device__ __forceinline__ NqRanges nq_ranges(int vs, int ve, int rc, long long base, long long window)
{
    long long wrem = window - base; // candidates still allowed from this column on
    NqRanges r;
    if (wrem <= 0) { r.d0 = r.d1 = r.u0 = r.u1 = 0; return r; }
    const long long lim_d = wrem;     // v < wrem           for v < rc
    const long long lim_u = wrem + 1; // v - 1 < wrem       for v > rc
    r.d0 = vs;
    long long d1 = ve < rc ? ve : rc;
    if (d1 > lim_d) d1 = lim_d;
    r.d1 = (int)d1;
    r.u0 = vs > rc + 1 ? vs : rc + 1;
    long long u1 = ve;
    if (u1 > lim_u) u1 = lim_u;
    r.u1 = (int)u1;
    if (r.d1 < r.d0) r.d1 = r.d0;
    if (r.u1 < r.u0) r.u1 = r.u0;
    return r;
}

// packed u16 occupancy counters, two per LDS word
__device__ __forceinline__ void tab16_inc(unsigned *t, int idx) { atomicAdd(&t[idx >> 1], 1u << (16 * (idx & 1))); }
__device__ __forceinline__ int tab16_get(const unsigned *t, int idx)
{
    return (int)reinterpret_cast<const unsigned short *>(t)[idx];
}
__device__ __forceinline__ void tab16_dec(unsigned *t, int idx) { atomicSub(&t[idx >> 1], 1u << (16 * (idx & 1))); }

// Candidates one restart's launch scores: slot s holds the n - 1 real candidates of column cols[s] at window ranks
// [s(n-1), (s+1)(n-1)); a slot whose column (or, for unvalidated device-pointer callers, whose rows[column]) is
// outside [0, n) is skipped by every eval kernel but keeps its window positions, so
//   n_scored = sum over valid slots of clamp(window - s(n-1), 0, n-1).
__device__ __forceinline__ int nq_slot_scored(int s, int c, int rc, int n, long long window)
{
    if ((unsigned)c >= (unsigned)n || (unsigned)rc >= (unsigned)n) return 0;
    const long long left = window - (long long)s * (n - 1);
    return left <= 0 ? 0 : (left < n - 1 ? (int)left : n - 1);
}

constexpr int NQ_STEP_MAX_N = 8192;         // 10n (tables) + 4n (rows) bytes of LDS
constexpr long long NQ_STEP_MAX_K = 65536;  // unfiltered positions per restart one block is asked to walk
constexpr int NQ_STEP_MAX_SLOTS = 1024;

// TABLE, column teams, TWO candidates per lane and LDS look-up (even n, 512 <= n <= NQ_TABLE_LDS_MAX_N): the team kernel
// above is bound by the LDS array (three 2-byte look-ups per candidate, 2 LDS cycles per wave-instruction whatever the
// width).  Here a lane owns the PAIR (v, v + 1): one ds_read_b32 per table returns both candidates' counts, which are
// added as packed 16-bit fields (a count is <= n + 2 < 2^15: no carry between the halves).  The three pairs are 4-byte
// aligned when v = c (mod 2): then v - c + n and v + c are even (n even), and the row table is read at v for even c and
// through a copy shifted by one entry (R1[i] = R[i + 1]) for odd c.  For odd c the pairs are (1,2) .. (n-1, n): square
// v = n does not exist -- R1[n - 1] holds a sentinel count that can never win -- and v = 0 is scored separately by lane 0.
constexpr unsigned NQ_PAIR_SENTINEL = 0x4000u; // > any real count (<= n + 2 <= 12290), and sentinel + 2n < 2^16: no carry

// One (slot, part) unit of the paired column walk: lane tv of a TL-lane team scores the pairs (v, v + 1),
// v = 2 tv + (c & 1) + 2 TL i for the steps i in [i0, i0 + per_part), and folds the unit's (count, v) minimum into `best`.
// Table layout (u16 indices): R at 0, R1 at n, A at 2n, B at 4n + 2 (nq_table_pairs_kernel).
template <int TL, bool WANT_SCORES>
__device__ __forceinline__ void nq_pairs_unit(const unsigned *s_t16, int n, int tv, int c, int rc, int conf_old, int slot, int i0,
                                              int per_part, long long *srow, long long cur, unsigned long long &best)
{
    constexpr int SH = TL == 64 ? 6 : 8;
    const unsigned short *t16 = reinterpret_cast<const unsigned short *>(s_t16);
    const int oR1 = n, oA = 2 * n, oB = 4 * n + 2;
    const int steps = (n + 2 * TL - 1) >> (SH + 1);
    const int par = c & 1;
    int i1 = i0 + per_part;
    i1 = i1 > steps ? steps : i1;
    const int full = n >> (SH + 1); // steps in which every lane's pair starts below n
    const int i1f = i1 < full ? i1 : full;
    // dword pointers: lane tv owns the pair starting at v = 2 tv + par (+ 512 i)
    const unsigned *pR = s_t16 + ((par ? oR1 : 0) >> 1) + tv;                 // R[v] / R1[v - 1]
    const unsigned *pA = s_t16 + ((oA + par - c + n) >> 1) + tv;             // A[v - c + n]
    const unsigned *pB = s_t16 + ((oB + par + c) >> 1) + tv;                 // B[v + c]
    const int vbase = 2 * tv + par;
    unsigned m = NQ_KEY32_EMPTY;
    int i = i0;
    // The per-lane running minimum orders by (count, position in the lane's walk): the position 2 i + {0, 1} is wave-uniform,
    // so it enters the key as a scalar operand (no per-candidate VGPR arithmetic); v is rebuilt from it once per unit.
#define LS_PAIR(J)                                                                                     \
    const unsigned s##J = pR[(i + J) << SH] + pA[(i + J) << SH] + pB[(i + J) << SH];                   \
    const unsigned lo##J = ((s##J << 15) & 0x7fffffffu) | (unsigned)(2 * (i + J));                     \
    const unsigned hi##J = ((s##J >> 16) << 15) | (unsigned)(2 * (i + J) + 1);
#define LS_PAIR_SCORES(J)                                                                              \
    if (WANT_SCORES) {                                                                                 \
        const int v##J = vbase + ((i + J) << (SH + 1));                                                \
        if (v##J != rc) srow[v##J] = cur + 2ll * ((int)(lo##J >> 15) - conf_old);                      \
        if (v##J + 1 != rc && v##J + 1 < n) srow[v##J + 1] = cur + 2ll * ((int)(hi##J >> 15) - conf_old); \
    }
    // (Issuing the NEXT four steps' twelve look-ups before reducing the current four -- a hand-pipelined loop -- was measured
    // slower, also where the launch is not dispatch-bound: 189 against 164 us for 64 restarts x 4096 columns; the register
    // copies cost more than the overlap buys.)
    for (; i + 4 <= i1f; i += 4) {
        LS_PAIR(0) LS_PAIR(1) LS_PAIR(2) LS_PAIR(3)
        unsigned m0 = lo0 < hi0 ? lo0 : hi0, m1 = lo1 < hi1 ? lo1 : hi1, m2 = lo2 < hi2 ? lo2 : hi2,
                 m3 = lo3 < hi3 ? lo3 : hi3;
        m0 = m0 < m1 ? m0 : m1;
        m2 = m2 < m3 ? m2 : m3;
        m0 = m0 < m2 ? m0 : m2;
        m = m0 < m ? m0 : m;
        LS_PAIR_SCORES(0) LS_PAIR_SCORES(1) LS_PAIR_SCORES(2) LS_PAIR_SCORES(3)
    }
    for (; i < i1; i++) {
        if (vbase + (i << (SH + 1)) < n) {
            LS_PAIR(0)
            const unsigned m0 = lo0 < hi0 ? lo0 : hi0;
            m = m0 < m ? m0 : m;
            LS_PAIR_SCORES(0)
        }
    }
    // position -> v:  v = vbase + 2 TL (pos >> 1) + (pos & 1)
    auto v_of = [&](unsigned key) { return vbase + (int)(((key & 0x7fffu) >> 1) << (SH + 1)) + (int)(key & 1u); };
    unsigned k_zero = NQ_KEY32_EMPTY; // odd column: square v = 0 belongs to no pair (lane 0 of the column's first part scores it)
    if (par && i0 == 0 && tv == 0) {
        k_zero = ((unsigned)t16[0] + (unsigned)t16[oA - c + n] + (unsigned)t16[oB + c]) << 15;
        if (WANT_SCORES && rc != 0) srow[0] = cur + 2ll * ((int)(k_zero >> 15) - conf_old);
    }
    const unsigned noop_cnt = (unsigned)(conf_old + 3);
    if (m != NQ_KEY32_EMPTY && (m >> 15) == noop_cnt && v_of(m) == rc) { // this lane's minimum is the tabu square itself: walk again without it
        m = NQ_KEY32_EMPTY;
        for (int i2 = i0; i2 < i1; i2++) {
            i = i2;
            const int v0 = vbase + (i << (SH + 1));
            if (v0 < n) {
                LS_PAIR(0)
                if (v0 != rc) m = lo0 < m ? lo0 : m;
                if (v0 + 1 != rc && v0 + 1 < n) m = hi0 < m ? hi0 : m;
            }
        }
    }
#undef LS_PAIR
#undef LS_PAIR_SCORES
    int cnt_best = -1, v_best = 0;
    if (m != NQ_KEY32_EMPTY) { cnt_best = (int)(m >> 15); v_best = v_of(m); }
    if (k_zero != NQ_KEY32_EMPTY && rc != 0 && (cnt_best < 0 || (int)(k_zero >> 15) <= cnt_best)) { // v = 0 precedes every other v
        cnt_best = (int)(k_zero >> 15);
        v_best = 0;
    }
    if (cnt_best >= 0)
        best = u64min(best, nq_key(cnt_best - conf_old, nq_key_hi(v_best > rc ? 1 : 0, c, n, slot), v_best));
}

} // namespace lsamd
