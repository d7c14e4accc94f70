// nq_kernels.hip — gfx950 kernels for the N-queens neighbourhood pipeline.
//
// Replaces, per launch and for a batch of independent restarts, the reference's
//   iter_local_moves -> filter(!tabu) -> map(score) -> take(window) -> sort -> first
// (local-search/src/local_search.rs:315-325) with the N-queens plug-in
// (examples/nqueens/src/lib.rs:74-87 pair test, :126-140 score, :214-234 order).
//
// Scoring identity used by every kernel here (checked against the literal
// O(n^2) restatement in the oracle tests):
//   score(board with rows[c] = v) = score(board) - 2*conf(c, rows[c]) + 2*conf(c, v)
//   conf(c, r) = #{ j != c : rows[j] == r  or  |rows[j] - r| == |j - c| }
// The pair test splits into three mutually exclusive equalities
//   rows[j] == r,   rows[j] - j == r - c,   rows[j] + j == r + c
// (used by the packed-u16 board and the count tables) or, with d = |rows[j] - r| and
// e = |j - c|, into the single product test  d * (d - e) == 0  (used by the fp32 boards).
//
// Winner selection: minimum by (score, board vector) (local_search.rs:29-37).
// All candidates of one launch are single-position edits of one board, so the
// vector order has a closed form: edits that lower rows[c] ("down") sort before
// all others by (c asc, v asc); edits that raise it ("up") sort after, by
// (c desc, v asc).  A 64-bit key (delta | up | colkey | v) makes the argmin an
// integer min (slot = position of the column in the proposer's list, last, so that the earlier of two
// identical candidates wins).  delta = conf(c,v) - conf(c,rows[c]) orders like the score.
//
// Kernel families
//   SCAN  (default, north-star shape): one wavefront per candidate; the packed state lives in
//         the wavefront's registers (n <= 4096) or in LDS (n <= 32768) and is scanned in full
//         for every candidate; per-lane counts are combined through a wave-private LDS tile
//         (register path: lane q then owns candidate q) or DPP adds (LDS path); wavefronts
//         keep running key minima, blocks combine through LDS, one 64-bit atomicMin per block.
//         n > 512: two list columns share a pass wherever both are whole (rows[j] - v is common
//         to them): 3.5 instead of 4 instructions per (candidate, board column) in the register
//         kernel, 3.75 instead of 4.5 in the LDS kernel.
//   TABLE occupancy histograms (rows / two diagonals) built once per board;
//         a candidate is three table reads.  One lane per candidate.
#include <atomic>

#include "ls_common.hpp"
#include "nq_device.hpp"

namespace lsamd {

#ifdef LS_NQ_STAMPS
// Diagnostic build only (tools/table_stamps.py builds its own copy of the library with -DLS_NQ_STAMPS): wall-clock stamps
// (100 MHz) of the TABLE kernel's phases, written to a buffer no other code reads.  The product build has none of this.
__device__ unsigned long long g_nq_stamps[1024 * 8];
#define LS_STAMP(P) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_nq_stamps[blockIdx.x * 8 + (P)] = wall_clock64(); } while (0)
#else
#define LS_STAMP(P) do { } while (0)
#endif


struct NqArgs {
    const int *rows;             // [n_restarts][n]
    const long long *cur_scores; // [n_restarts]
    const int *cols;             // [n_restarts][max_cols]
    const int *n_cols;           // [n_restarts]
    int n;
    int n_restarts;
    int max_cols;
    long long window;
    unsigned long long *best_keys; // [n_restarts], pre-set to NQ_KEY_EMPTY
    long long *scores_out;         // nullable [n_restarts][max_cols*n], pre-filled with INT64_MIN
    const int *tabs;               // TABLE only: [n_restarts][tab_stride]
    int tab_stride;
};


// ===================================================================== SCAN
// Register-resident state.  A "board" policy holds the three equality arrays of
// one restart in the wavefront's registers and counts, for four consecutive
// values v..v+3 of one column c, the matches of (v, v-c, v+c) over all columns.
// Variants differ only in how a match is turned into a count (measured on
// MI355X); all produce identical counts.
//   SELF = what the column's own queen adds when its own square is counted.
// (A third formulation — int32 compares into scalar masks, 3 v_cmp + 2 s_or + 1 v_addc per column —
// was the first version and ran at 1.28e9 neighbours/s; it is gone: v_cmp->sgpr is half rate and the
// scalar ORs serialise behind it.)

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// Waves per SIMD to run a kernel needing `vgprs` registers per lane at (512-entry file), rounded
// DOWN to a power of two: measured with tools/valu_rates.hip, the scan kernel's instruction mix
// issues at 0.92 / 1.16 / 0.89 ns per instruction with 2 / 3 / 4 waves per SIMD — an odd count
// costs 25 %.
constexpr int nq_waves_for(int vgprs)
{
    const int w = 512 / vgprs;
    return w >= 8 ? 8 : (w >= 4 ? 4 : (w >= 2 ? 2 : 1));
}

__device__ __forceinline__ us2 as_us2(unsigned x) { return __builtin_bit_cast(us2, x); }
__device__ __forceinline__ unsigned as_u32(us2 x) { return __builtin_bit_cast(unsigned, x); }

// min(x, y) per 16-bit half as ONE instruction.  Written as asm because hipcc
// rewrites min(x, 1) into two scalar-half compares + selects + a permute.
__device__ __forceinline__ us2 pk_min_u16_raw(us2 x, us2 y)
{
    us2 r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// Packed 16-bit lanes: register i of lane l holds columns l + 64*(2i) (low half)
// and l + 64*(2i+1) (high half)  (n <= 128*NRP <= 8192).  A is stored biased by
// +n so every entry is a u16.  A match zeroes the XOR with the replicated
// comparand; min over the three XORs is zero iff the column attacks; pure VALU,
// 3.5 instructions per column pair and candidate, no scalar mask traffic.
template <int NRP>
struct BoardPk16 {
    static constexpr int CAP = 128 * NRP;
    static constexpr int SELF = 1;
    static constexpr int WAVES = nq_waves_for(3 * NRP + 32);
    static constexpr bool PAIR = false; // one column per pass
    static constexpr int TILES = 1;
    us2 R[NRP], A[NRP], B[NRP];
    int n;
    __device__ __forceinline__ void load(const int *__restrict__ rows, int n_, int lane)
    {
        n = n_;
#pragma unroll
        for (int i = 0; i < NRP; i++) {
            unsigned pr = 0, pa = 0, pb = 0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j = lane + 64 * (2 * i + h);
                const bool in = j < n_;
                const int r = in ? rows[j] : 0;
                const unsigned vr = in ? (unsigned)r : 0xffffu;
                const unsigned va = in ? (unsigned)(r - j + n_) : 0xffffu;
                const unsigned vb = in ? (unsigned)(r + j) : 0xffffu;
                pr |= vr << (16 * h); pa |= va << (16 * h); pb |= vb << (16 * h);
            }
            R[i] = as_us2(pr); A[i] = as_us2(pa); B[i] = as_us2(pb);
        }
    }
    static constexpr bool MISMATCH = true; // lanes4 counts mismatching halves (<= 2*NRP <= 128)
    static constexpr int READ_UNROLL = 2;
    static constexpr int LANE_TOTAL = 2 * NRP;
    __device__ __forceinline__ void prepare(int, int) {}
    __device__ __forceinline__ unsigned lanes4(int v, int c) const
    {
        const unsigned rep = 0x00010001u;
        const unsigned Vr = (unsigned)v * rep, Va = (unsigned)(v - c + n) * rep, Vb = (unsigned)(v + c) * rep;
        const us2 one = as_us2(rep);
        us2 m0 = as_us2(0u), m1 = m0, m2 = m0, m3 = m0; // mismatching halves per candidate
#pragma unroll
        for (int i = 0; i < NRP; i++) {
            const unsigned r = as_u32(R[i]), a = as_u32(A[i]), b = as_u32(B[i]);
#define LS_PK_STEP(acc, u)                                                                          \
    {                                                                                               \
        const us2 x = __builtin_elementwise_min(                                                    \
            __builtin_elementwise_min(as_us2(r ^ (Vr + (u) * rep)), as_us2(a ^ (Va + (u) * rep))),  \
            as_us2(b ^ (Vb + (u) * rep)));                                                          \
        acc += pk_min_u16_raw(x, one);                                                              \
    }
            LS_PK_STEP(m0, 0u)
            LS_PK_STEP(m1, 1u)
            LS_PK_STEP(m2, 2u)
            LS_PK_STEP(m3, 3u)
#undef LS_PK_STEP
        }
        return (unsigned)(m0.x + m0.y) | ((unsigned)(m1.x + m1.y) << 8) | ((unsigned)(m2.x + m2.y) << 16) |
               ((unsigned)(m3.x + m3.y) << 24);
    }
};

// clamp(|x| * |y|, 0, 1) as ONE instruction (source abs + output clamp modifiers).  hipcc
// emits v_mul + v_max(clamp) for the equivalent C expression, so it is spelled out.
__device__ __forceinline__ float mul_abs_clamp(float x, float y)
{
    float r;
    asm("v_mul_f32_e64 %0, |%1|, |%2| clamp" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// FP32 lanes: lane l holds rows[j] as a float for j = l + 64*i, plus, for the column c
// being scanned, E[i] = |j - c| (recomputed when the column changes).  With d = |rows[j] - v|
// column j attacks square (c, v) iff d == 0 or d == |j - c|, i.e. iff d * (d - E) == 0
// (also true for the own column j == c where E == 0).  Both factors are integers of
// magnitude < 2^15, so the product is zero exactly when a factor is and otherwise >= 1 in
// magnitude; the clamp output modifier turns it into a 0/1 mismatch flag.  4 FP32
// instructions per column and candidate (sub, sub, mul+abs+clamp, add), which issue at
// twice the rate of the integer VALU ops on gfx950.
template <int NR>
struct BoardF32 {
    static constexpr int CAP = 64 * NR;
    static constexpr int SELF = 1;
    // NR >= 16 (512 < n <= 4096): TWO list columns per pass (lanes4x2).  d = rows[j] - v does not depend on the
    // candidate's column, so two columns that walk the same values share it: 3.5 instructions per (candidate, column)
    // instead of 4.  A third register array holds the second column's |j - c|: NR = 64 and 16 keep their 2 and 4 waves per
    // SIMD, NR = 32 goes from 4 to 2 (the same ns per instruction, nq_waves_for); below, the second tile would cost
    // residency for launches that rarely hold two whole columns per wavefront.
    static constexpr bool PAIR = NR >= 16;
    static constexpr int TILES = PAIR ? 2 : 1;
    static constexpr int WAVES = nq_waves_for((PAIR ? 3 : 2) * NR + 40);
    float R[NR], E[NR], E2[PAIR ? NR : 1];
    __device__ __forceinline__ void load(const int *__restrict__ rows, int n, int lane)
    {
#pragma unroll
        for (int i = 0; i < NR; i++) {
            const int j = lane + 64 * i;
            R[i] = j < n ? (float)rows[j] : 1.0e9f; // padding never matches
            E[i] = 0.0f;
        }
    }
    __device__ __forceinline__ void prepare(int c, int lane)
    {
#pragma unroll
        for (int i = 0; i < NR; i++) E[i] = __builtin_fabsf((float)(lane + 64 * i - c));
    }
    __device__ __forceinline__ void prepare2(int c2, int lane)
    {
        if constexpr (PAIR) {
#pragma unroll
            for (int i = 0; i < NR; i++) E2[i] = __builtin_fabsf((float)(lane + 64 * i - c2));
        }
    }
    static constexpr bool MISMATCH = true; // lanes4 counts mismatching columns (<= NR <= 64)
    static constexpr int READ_UNROLL = 4;
    static constexpr int LANE_TOTAL = NR;
    __device__ __forceinline__ unsigned lanes4(int v, int c) const
    {
        (void)c;
        const float f0 = (float)v, f1 = (float)(v + 1), f2 = (float)(v + 2), f3 = (float)(v + 3);
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f; // mismatching columns per candidate
        // Four candidates are interleaved so that neighbouring instructions are independent, and
        // four columns share one asm statement: hipcc pads every asm statement with an s_nop.
#define LS_F32_COL(r, e)                                                                            \
    "v_sub_f32_e32 %4, " r ", %20\n\tv_sub_f32_e32 %5, " r ", %21\n\t"                              \
    "v_sub_f32_e32 %6, " r ", %22\n\tv_sub_f32_e32 %7, " r ", %23\n\t"                              \
    "v_sub_f32_e64 %8, |%4|, " e "\n\tv_sub_f32_e64 %9, |%5|, " e "\n\t"                            \
    "v_sub_f32_e64 %10, |%6|, " e "\n\tv_sub_f32_e64 %11, |%7|, " e "\n\t"                          \
    "v_mul_f32_e64 %4, |%4|, |%8| clamp\n\tv_mul_f32_e64 %5, |%5|, |%9| clamp\n\t"                  \
    "v_mul_f32_e64 %6, |%6|, |%10| clamp\n\tv_mul_f32_e64 %7, |%7|, |%11| clamp\n\t"                \
    "v_add_f32_e32 %0, %0, %4\n\tv_add_f32_e32 %1, %1, %5\n\t"                                      \
    "v_add_f32_e32 %2, %2, %6\n\tv_add_f32_e32 %3, %3, %7\n\t"
        constexpr int G = NR >= 4 ? 4 : NR;
#pragma unroll
        for (int i = 0; i < NR; i += G) {
            float d0, d1, d2, d3, u0, u1, u2, u3;
            if constexpr (G == 4) {
                asm(LS_F32_COL("%12", "%16") LS_F32_COL("%13", "%17") LS_F32_COL("%14", "%18") LS_F32_COL("%15", "%19")
                    : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3),
                      "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3)
                    : "v"(R[i]), "v"(R[i + 1]), "v"(R[i + 2]), "v"(R[i + 3]), "v"(E[i]), "v"(E[i + 1]), "v"(E[i + 2]),
                      "v"(E[i + 3]), "v"(f0), "v"(f1), "v"(f2), "v"(f3));
            } else {
#pragma unroll
                for (int g = 0; g < G; g++) {
                    // same operand numbering as above with a single column (NR < 4: n <= 128)
                    asm(LS_F32_COL("%12", "%16")
                        : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3),
                          "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3)
                        : "v"(R[i + g]), "v"(R[i + g]), "v"(R[i + g]), "v"(R[i + g]), "v"(E[i + g]), "v"(E[i + g]),
                          "v"(E[i + g]), "v"(E[i + g]), "v"(f0), "v"(f1), "v"(f2), "v"(f3));
                }
            }
        }
#undef LS_F32_COL
        return (unsigned)(int)m0 | ((unsigned)(int)m1 << 8) | ((unsigned)(int)m2 << 16) | ((unsigned)(int)m3 << 24);
    }
    // Values v .. v+3 of the column of prepare() (-> wa) and of the column of prepare2() (-> wb): per board column the
    // four differences once, then sub / mul+abs+clamp / add per candidate: 28 instructions for 8 candidates.
    __device__ __forceinline__ void lanes4x2(int v, unsigned &wa, unsigned &wb) const
    {
        const float f0 = (float)v, f1 = (float)(v + 1), f2 = (float)(v + 2), f3 = (float)(v + 3);
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#define LS_F32_COL_E(e, a0, a1, a2, a3)                                                             \
    "v_sub_f32_e64 %12, |%8|, " e "\n\tv_sub_f32_e64 %13, |%9|, " e "\n\t"                          \
    "v_sub_f32_e64 %14, |%10|, " e "\n\tv_sub_f32_e64 %15, |%11|, " e "\n\t"                        \
    "v_mul_f32_e64 %12, |%8|, |%12| clamp\n\tv_mul_f32_e64 %13, |%9|, |%13| clamp\n\t"              \
    "v_mul_f32_e64 %14, |%10|, |%14| clamp\n\tv_mul_f32_e64 %15, |%11|, |%15| clamp\n\t"            \
    "v_add_f32_e32 " a0 ", " a0 ", %12\n\tv_add_f32_e32 " a1 ", " a1 ", %13\n\t"                    \
    "v_add_f32_e32 " a2 ", " a2 ", %14\n\tv_add_f32_e32 " a3 ", " a3 ", %15\n\t"
#define LS_F32_COL2(r, ea, eb)                                                                      \
    "v_sub_f32_e32 %8, " r ", %22\n\tv_sub_f32_e32 %9, " r ", %23\n\t"                              \
    "v_sub_f32_e32 %10, " r ", %24\n\tv_sub_f32_e32 %11, " r ", %25\n\t"                            \
    LS_F32_COL_E(ea, "%0", "%1", "%2", "%3") LS_F32_COL_E(eb, "%4", "%5", "%6", "%7")
        if constexpr (PAIR) {
#pragma unroll
            for (int i = 0; i < NR; i += 2) {
                float d0, d1, d2, d3, u0, u1, u2, u3;
                asm(LS_F32_COL2("%16", "%18", "%20") LS_F32_COL2("%17", "%19", "%21")
                    : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "=&v"(d0), "=&v"(d1),
                      "=&v"(d2), "=&v"(d3), "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3)
                    : "v"(R[i]), "v"(R[i + 1]), "v"(E[i]), "v"(E[i + 1]), "v"(E2[i]), "v"(E2[i + 1]), "v"(f0), "v"(f1), "v"(f2),
                      "v"(f3));
            }
        }
#undef LS_F32_COL2
#undef LS_F32_COL_E
        wa = (unsigned)(int)m0 | ((unsigned)(int)m1 << 8) | ((unsigned)(int)m2 << 16) | ((unsigned)(int)m3 << 24);
        wb = (unsigned)(int)p0 | ((unsigned)(int)p1 << 8) | ((unsigned)(int)p2 << 16) | ((unsigned)(int)p3 << 24);
    }
};

// 64 candidates per pass.  Sixteen steps of four candidates each leave per-lane counts (one byte
// per candidate) in a wave-private LDS tile, row = step, column = lane; then lane q OWNS candidate q:
// it sums byte (q & 3) of row (q >> 2) over the 64 lanes, forms delta and key, and keeps its own
// running minimum.  One tile transpose per 64 candidates replaces a 6-step DPP reduction, the
// conversions and the scalar key work per candidate; the debug score store becomes coalesced.
// Rows are 68 dwords apart so that the sixteen row groups of a b128 read hit different banks.
constexpr int NQ_TILE_STRIDE = 68;
constexpr int NQ_TILE_DWORDS = 16 * NQ_TILE_STRIDE;

template <bool WANT_SCORES, class Board>
__device__ __forceinline__ void nq_scan_column_regs(const Board &bd, unsigned *tile, const NqRanges &rg, int rc, int c, int n,
                                                    int slot, unsigned long long &best, long long *scores_row, long long cur,
                                                    int lane)
{
    // ONE range [lo, hi) per column through ONE copy of the (fully unrolled, ~8 KB) column scan: the "down" moves, the
    // column's own square and the "up" moves in a single sequence of 64-candidate passes.  The own square is not a
    // candidate (tabu) but its count is conf_old + SELF, so when it lies inside the range it costs nothing extra;
    // otherwise (the range was cut by the window or by the wavefront's share) a one-step pass fetches it first.
    // Inside the column only the count and v vary, so each lane keeps two 32-bit minima of (count << 15 | v)
    // (down / up) and widens them to 64-bit keys once, when conf_old is known.
    const int lo = rg.d1 > rg.d0 ? rg.d0 : rg.u0;
    const int hi = rg.u1 > rg.u0 ? rg.u1 : rg.d1;
    const bool own_inside = rc >= lo && rc < hi;
    const bool conf_first = WANT_SCORES || !own_inside; // the parity-mode score store needs conf_old up front
    int conf_old = 0, own_cnt = 0;
    unsigned md = NQ_KEY32_EMPTY, mu = NQ_KEY32_EMPTY;
#pragma unroll 1
    for (int seg = conf_first ? 0 : 1; seg < 2; seg++) {
        const int vbeg = seg == 0 ? rc : lo;
        const int vend = seg == 0 ? rc + 1 : hi;
#pragma unroll 1
        for (int vb = vbeg; vb < vend; vb += 64) {
            const int left = vend - vb;
            const int steps = left >= 64 ? 16 : (left + 3) >> 2;
            __builtin_amdgcn_wave_barrier(); // the previous pass's reads are program-ordered before these writes
#pragma unroll 1 // one copy of the (fully unrolled) column scan; unrolling this loop spills the board
            for (int st = 0; st < steps; st++) tile[st * NQ_TILE_STRIDE + lane] = bd.lanes4(vb + 4 * st, c);
            // The tile is private to this wavefront and LDS operations of one wavefront complete in order, so no
            // s_barrier is needed; the fences only stop the compiler from moving the reads above other lanes' writes.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned *row = tile + (lane >> 2) * NQ_TILE_STRIDE;
            const int sh = (lane & 3) * 8;
            int sum = 0;
#pragma clang loop unroll_count(Board::READ_UNROLL) // b128 reads in flight; more would spill the register-resident board
            for (int l4 = 0; l4 < 16; l4++) {
                const uint4 x = *reinterpret_cast<const uint4 *>(row + 4 * l4);
                sum += (int)((x.x >> sh) & 0xffu) + (int)((x.y >> sh) & 0xffu) + (int)((x.z >> sh) & 0xffu) +
                       (int)((x.w >> sh) & 0xffu);
            }
            const int cnt = Board::MISMATCH ? 64 * Board::LANE_TOTAL - sum : sum;
            if (seg == 0) {
                conf_old = __builtin_amdgcn_readfirstlane(cnt) - Board::SELF;
            } else {
                const int v = vb + lane;
                const unsigned k32 = ((unsigned)cnt << 15) | (unsigned)v;
                if (v < rg.d1) md = k32 < md ? k32 : md;                    // v in [d0, d1): lo == d0 whenever this is non-empty
                else if (v >= rg.u0 && v < rg.u1) mu = k32 < mu ? k32 : mu;
                else if (v == rc) own_cnt = cnt;
                if (WANT_SCORES && (v < rg.d1 || (v >= rg.u0 && v < rg.u1))) scores_row[v] = cur + 2ll * (cnt - conf_old);
            }
        }
    }
    if (!conf_first) conf_old = __builtin_amdgcn_readlane(own_cnt, (rc - lo) & 63) - Board::SELF;
    if (md != NQ_KEY32_EMPTY)
        best = u64min(best, nq_key((int)(md >> 15) - conf_old, nq_key_hi(0, c, n, slot), (int)(md & 0x7fffu)));
    if (mu != NQ_KEY32_EMPTY)
        best = u64min(best, nq_key((int)(mu >> 15) - conf_old, nq_key_hi(1, c, n, slot), (int)(mu & 0x7fffu)));
}

// lane q's candidate of a 64-candidate pass: byte (q & 3) of row (q >> 2), summed over the 64 lanes' words
template <class Board>
__device__ __forceinline__ int nq_tile_count(const unsigned *tile, int lane)
{
    const unsigned *row = tile + (lane >> 2) * NQ_TILE_STRIDE;
    const int sh = (lane & 3) * 8;
    int sum = 0;
#pragma clang loop unroll_count(Board::READ_UNROLL)
    for (int l4 = 0; l4 < 16; l4++) {
        const uint4 x = *reinterpret_cast<const uint4 *>(row + 4 * l4);
        sum += (int)((x.x >> sh) & 0xffu) + (int)((x.y >> sh) & 0xffu) + (int)((x.z >> sh) & 0xffu) + (int)((x.w >> sh) & 0xffu);
    }
    return Board::MISMATCH ? 64 * Board::LANE_TOTAL - sum : sum;
}

// TWO list columns (slots `slot` and `slot + 1`: columns ca, cb with queens on rows ra, rb) whose n - 1 candidates are all
// inside the wavefront's share and inside the window: both walk v = 0 .. n-1, so one pass of Board::lanes4x2 serves both
// (two tiles).  Same minima, same keys as two calls of nq_scan_column_regs; the own squares' counts (conf_old + SELF) come
// out of the walk, or, when the parity-mode score store needs them up front, out of a one-step pass per column.
template <bool WANT_SCORES, class Board>
__device__ __forceinline__ void nq_scan_column_pair_regs(const Board &bd, unsigned *tile, int ra, int rb, int ca, int cb, int n,
                                                         int slot, unsigned long long &best, long long *srow_a, long long *srow_b,
                                                         long long cur, int lane)
{
    unsigned *tile_b = tile + NQ_TILE_DWORDS;
    int conf_a = 0, conf_b = 0, own_a = 0, own_b = 0;
    unsigned mda = NQ_KEY32_EMPTY, mua = NQ_KEY32_EMPTY, mdb = NQ_KEY32_EMPTY, mub = NQ_KEY32_EMPTY;
    if (WANT_SCORES) {
        unsigned wa, wb, skip;
        __builtin_amdgcn_wave_barrier();
        bd.lanes4x2(ra, wa, skip);
        bd.lanes4x2(rb, skip, wb);
        tile[lane] = wa;
        tile_b[lane] = wb;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        conf_a = __builtin_amdgcn_readfirstlane(nq_tile_count<Board>(tile, lane)) - Board::SELF;
        conf_b = __builtin_amdgcn_readfirstlane(nq_tile_count<Board>(tile_b, lane)) - Board::SELF;
    }
#pragma unroll 1
    for (int vb = 0; vb < n; vb += 64) {
        const int left = n - vb;
        const int steps = left >= 64 ? 16 : (left + 3) >> 2;
        __builtin_amdgcn_wave_barrier(); // the previous pass's reads are program-ordered before these writes
#pragma unroll 1 // one copy of the (fully unrolled) two-column scan
        for (int st = 0; st < steps; st++) {
            unsigned wa, wb;
            bd.lanes4x2(vb + 4 * st, wa, wb);
            tile[st * NQ_TILE_STRIDE + lane] = wa;
            tile_b[st * NQ_TILE_STRIDE + lane] = wb;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int cnt_a = nq_tile_count<Board>(tile, lane);
        const int cnt_b = nq_tile_count<Board>(tile_b, lane);
        const int v = vb + lane;
        if (v < n) {
            const unsigned ka = ((unsigned)cnt_a << 15) | (unsigned)v, kb = ((unsigned)cnt_b << 15) | (unsigned)v;
            if (v < ra) mda = ka < mda ? ka : mda;
            else if (v > ra) mua = ka < mua ? ka : mua;
            else own_a = cnt_a;
            if (v < rb) mdb = kb < mdb ? kb : mdb;
            else if (v > rb) mub = kb < mub ? kb : mub;
            else own_b = cnt_b;
            if (WANT_SCORES) {
                if (v != ra) srow_a[v] = cur + 2ll * (cnt_a - conf_a);
                if (v != rb) srow_b[v] = cur + 2ll * (cnt_b - conf_b);
            }
        }
    }
    if (!WANT_SCORES) {
        conf_a = __builtin_amdgcn_readlane(own_a, ra & 63) - Board::SELF;
        conf_b = __builtin_amdgcn_readlane(own_b, rb & 63) - Board::SELF;
    }
    if (mda != NQ_KEY32_EMPTY)
        best = u64min(best, nq_key((int)(mda >> 15) - conf_a, nq_key_hi(0, ca, n, slot), (int)(mda & 0x7fffu)));
    if (mua != NQ_KEY32_EMPTY)
        best = u64min(best, nq_key((int)(mua >> 15) - conf_a, nq_key_hi(1, ca, n, slot), (int)(mua & 0x7fffu)));
    if (mdb != NQ_KEY32_EMPTY)
        best = u64min(best, nq_key((int)(mdb >> 15) - conf_b, nq_key_hi(0, cb, n, slot + 1), (int)(mdb & 0x7fffu)));
    if (mub != NQ_KEY32_EMPTY)
        best = u64min(best, nq_key((int)(mub >> 15) - conf_b, nq_key_hi(1, cb, n, slot + 1), (int)(mub & 0x7fffu)));
}

template <class Board, bool WANT_SCORES>
__global__ __launch_bounds__(256, Board::WAVES) void nq_scan_regs_kernel(NqArgs a)
{
    __shared__ unsigned long long s_best[4];
    __shared__ __attribute__((aligned(16))) unsigned s_tile[4 * Board::TILES * NQ_TILE_DWORDS];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int restart = blockIdx.x % a.n_restarts;
    const int wchunk = (blockIdx.x / a.n_restarts) * 4 + wib;
    const int n_wchunks = (gridDim.x / a.n_restarts) * 4;
    const int n = a.n;
    const int *__restrict__ rows = a.rows + (long long)restart * n;
    const int *__restrict__ cols = a.cols + (long long)restart * a.max_cols;
    int ncols = a.n_cols[restart];
    ncols = ncols < 0 ? 0 : (ncols > a.max_cols ? a.max_cols : ncols);
    const long long K = nq_k_limit((long long)ncols * n, n, a.window);
    const long long k0 = K * wchunk / n_wchunks;
    const long long k1 = K * (wchunk + 1) / n_wchunks;
    unsigned long long best = NQ_KEY_EMPTY; // per-lane running minimum
    unsigned *tile = s_tile + wib * Board::TILES * NQ_TILE_DWORDS;

    if (k0 < k1) {
        Board bd;
        bd.load(rows, n, lane);
        const long long cur = a.cur_scores[restart];
        int slot = (int)(k0 / n);
        int v0 = (int)(k0 - (long long)slot * n);
        long long k = k0;
        while (k < k1) {
            const long long left = k1 - k;
            const int ve = (left < (long long)(n - v0)) ? (int)(v0 + left) : n;
            const int c = cols[slot];
            const int rc = (unsigned)c < (unsigned)n ? rows[c] : -1;
            if constexpr (Board::PAIR) {
                // this column and the next one whole, both inside the share and the window: one two-column walk
                if (v0 == 0 && left >= 2ll * n && (unsigned)rc < (unsigned)n &&
                    a.window - (long long)slot * (n - 1) >= 2ll * (n - 1)) {
                    const int c2 = cols[slot + 1];
                    const int rc2 = (unsigned)c2 < (unsigned)n ? rows[c2] : -1;
                    if ((unsigned)rc2 < (unsigned)n) {
                        bd.prepare(c, lane);
                        bd.prepare2(c2, lane);
                        long long *srow = WANT_SCORES ? a.scores_out + ((long long)restart * a.max_cols + slot) * n : nullptr;
                        nq_scan_column_pair_regs<WANT_SCORES>(bd, tile, rc, rc2, c, c2, n, slot, best, srow,
                                                              WANT_SCORES ? srow + n : nullptr, cur, lane);
                        k += 2ll * n;
                        slot += 2;
                        continue;
                    }
                }
            }
            if ((unsigned)rc < (unsigned)n) { // device-pointer callers are not validated: such a slot is skipped (nq_slot_scored)
                const long long base = (long long)slot * (n - 1);
                const NqRanges rg = nq_ranges(v0, ve, rc, base, a.window);
                if (rg.d1 > rg.d0 || rg.u1 > rg.u0) {
                    bd.prepare(c, lane);
                    long long *srow = WANT_SCORES ? a.scores_out + ((long long)restart * a.max_cols + slot) * n : nullptr;
                    nq_scan_column_regs<WANT_SCORES>(bd, tile, rg, rc, c, n, slot, best, srow, cur, lane);
                }
            }
            k += ve - v0;
            slot++;
            v0 = 0;
        }
    }
    best = wave_min_u64(best);
    if (lane == 0) s_best[wib] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = u64min(u64min(s_best[0], s_best[1]), u64min(s_best[2], s_best[3]));
        if (b != NQ_KEY_EMPTY) atomicMin(&a.best_keys[restart], b);
    }
}

// LDS-resident state for n > 4096 (any n <= LS_NQ_MAX_N): the board as floats, 4*ceil64(n) bytes,
// shared by the 16 wavefronts of a 1024-thread block (so that even n = 32768, whose 128 KB allow one
// block per CU, runs 4 waves per SIMD).  Same fp32 product test as BoardF32; |j - c| is formed on the
// fly (two extra instructions per column visit, shared by the four interleaved candidates).
#define LS_F32_LDS_STEP                                                                              \
    "v_sub_f32_e32 %4, %12, %14\n\tv_sub_f32_e32 %5, %12, %15\n\t"                                   \
    "v_sub_f32_e32 %6, %12, %16\n\tv_sub_f32_e32 %7, %12, %17\n\t"                                   \
    "v_sub_f32_e64 %8, |%4|, |%13|\n\tv_sub_f32_e64 %9, |%5|, |%13|\n\t"                             \
    "v_sub_f32_e64 %10, |%6|, |%13|\n\tv_sub_f32_e64 %11, |%7|, |%13|\n\t"                           \
    "v_mul_f32_e64 %4, |%4|, |%8| clamp\n\tv_mul_f32_e64 %5, |%5|, |%9| clamp\n\t"                   \
    "v_mul_f32_e64 %6, |%6|, |%10| clamp\n\tv_mul_f32_e64 %7, |%7|, |%11| clamp\n\t"                 \
    "v_add_f32_e32 %0, %0, %4\n\tv_add_f32_e32 %1, %1, %5\n\t"                                       \
    "v_add_f32_e32 %2, %2, %6\n\tv_add_f32_e32 %3, %3, %7"

// attack counts (wave-uniform) of (c, v) .. (c, v+3): columns scanned minus mismatches
__device__ __forceinline__ void nq_count4_lds(const float *sR, int n_pad, int lane, int v, int c, int (&cnt)[4])
{
    const float f0 = (float)v, f1 = (float)(v + 1), f2 = (float)(v + 2), f3 = (float)(v + 3);
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    // one ds_read_b128 per visit: the lane owns four CONSECUTIVE columns 4*lane + 256*i .. +3
    float jc0 = (float)(4 * lane - c), jc1 = jc0 + 1.0f, jc2 = jc0 + 2.0f, jc3 = jc0 + 3.0f; // j - c per owned column
    const float4 *p4 = reinterpret_cast<const float4 *>(sR) + lane;
    const int visits8 = n_pad >> 9; // wave-uniform trip count, two b128 reads in flight per trip (n_pad % 512 == 0)
    for (int i = 0; i < visits8; i++) {
        const float4 r = p4[128 * i], t = p4[128 * i + 64];
        float d0, d1, d2, d3, u0, u1, u2, u3;
#define LS_LDS_COL(R, JC)                                                                                             \
        asm(LS_F32_LDS_STEP                                                                                           \
            : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(u0), "=&v"(u1), \
              "=&v"(u2), "=&v"(u3)                                                                                    \
            : "v"(R), "v"(JC), "v"(f0), "v"(f1), "v"(f2), "v"(f3));
        LS_LDS_COL(r.x, jc0) LS_LDS_COL(r.y, jc1) LS_LDS_COL(r.z, jc2) LS_LDS_COL(r.w, jc3)
        jc0 += 256.0f; jc1 += 256.0f; jc2 += 256.0f; jc3 += 256.0f;
        LS_LDS_COL(t.x, jc0) LS_LDS_COL(t.y, jc1) LS_LDS_COL(t.z, jc2) LS_LDS_COL(t.w, jc3)
        jc0 += 256.0f; jc1 += 256.0f; jc2 += 256.0f; jc3 += 256.0f;
#undef LS_LDS_COL
    }
    // per-lane mismatches <= n_pad/64 <= 512; wave totals of matches <= n + 2 < 2^16: two per register
    const int visits = n_pad >> 6;
    const unsigned t01 = (unsigned)wave_sum_i32((visits - (int)m0) | ((visits - (int)m1) << 16));
    const unsigned t23 = (unsigned)wave_sum_i32((visits - (int)m2) | ((visits - (int)m3) << 16));
    cnt[0] = (int)(t01 & 0xffffu); cnt[1] = (int)(t01 >> 16);
    cnt[2] = (int)(t23 & 0xffffu); cnt[3] = (int)(t23 >> 16);
}

// The same for TWO list columns ca, cb over the same four values (their rows[j] - v are shared): 28 instructions per board
// column for 8 candidates instead of 2 x 16.
#define LS_F32_LDS_HALF(JC, A0, A1, A2, A3)                                                          \
    "v_sub_f32_e64 %12, |%8|, |" JC "|\n\tv_sub_f32_e64 %13, |%9|, |" JC "|\n\t"                      \
    "v_sub_f32_e64 %14, |%10|, |" JC "|\n\tv_sub_f32_e64 %15, |%11|, |" JC "|\n\t"                    \
    "v_mul_f32_e64 %12, |%8|, |%12| clamp\n\tv_mul_f32_e64 %13, |%9|, |%13| clamp\n\t"                \
    "v_mul_f32_e64 %14, |%10|, |%14| clamp\n\tv_mul_f32_e64 %15, |%11|, |%15| clamp\n\t"              \
    "v_add_f32_e32 " A0 ", " A0 ", %12\n\tv_add_f32_e32 " A1 ", " A1 ", %13\n\t"                      \
    "v_add_f32_e32 " A2 ", " A2 ", %14\n\tv_add_f32_e32 " A3 ", " A3 ", %15\n\t"
#define LS_F32_LDS_STEP2                                                                             \
    "v_sub_f32_e32 %8, %16, %19\n\tv_sub_f32_e32 %9, %16, %20\n\t"                                   \
    "v_sub_f32_e32 %10, %16, %21\n\tv_sub_f32_e32 %11, %16, %22\n\t"                                 \
    LS_F32_LDS_HALF("%17", "%0", "%1", "%2", "%3") LS_F32_LDS_HALF("%18", "%4", "%5", "%6", "%7")

__device__ __forceinline__ void nq_count4x2_lds(const float *sR, int n_pad, int lane, int v, int ca, int cb, int (&cnt_a)[4],
                                                int (&cnt_b)[4])
{
    const float f0 = (float)v, f1 = (float)(v + 1), f2 = (float)(v + 2), f3 = (float)(v + 3);
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    float ja0 = (float)(4 * lane - ca), ja1 = ja0 + 1.0f, ja2 = ja0 + 2.0f, ja3 = ja0 + 3.0f; // j - ca per owned column
    float jb0 = (float)(4 * lane - cb), jb1 = jb0 + 1.0f, jb2 = jb0 + 2.0f, jb3 = jb0 + 3.0f; // j - cb
    const float4 *p4 = reinterpret_cast<const float4 *>(sR) + lane;
    const int visits8 = n_pad >> 9;
    for (int i = 0; i < visits8; i++) {
        const float4 r = p4[128 * i], t = p4[128 * i + 64];
        float d0, d1, d2, d3, u0, u1, u2, u3;
#define LS_LDS_COL2(R, JA, JB)                                                                                        \
        asm(LS_F32_LDS_STEP2                                                                                          \
            : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "=&v"(d0), "=&v"(d1),      \
              "=&v"(d2), "=&v"(d3), "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3)                                          \
            : "v"(R), "v"(JA), "v"(JB), "v"(f0), "v"(f1), "v"(f2), "v"(f3));
        LS_LDS_COL2(r.x, ja0, jb0) LS_LDS_COL2(r.y, ja1, jb1) LS_LDS_COL2(r.z, ja2, jb2) LS_LDS_COL2(r.w, ja3, jb3)
        ja0 += 256.0f; ja1 += 256.0f; ja2 += 256.0f; ja3 += 256.0f;
        jb0 += 256.0f; jb1 += 256.0f; jb2 += 256.0f; jb3 += 256.0f;
        LS_LDS_COL2(t.x, ja0, jb0) LS_LDS_COL2(t.y, ja1, jb1) LS_LDS_COL2(t.z, ja2, jb2) LS_LDS_COL2(t.w, ja3, jb3)
        ja0 += 256.0f; ja1 += 256.0f; ja2 += 256.0f; ja3 += 256.0f;
        jb0 += 256.0f; jb1 += 256.0f; jb2 += 256.0f; jb3 += 256.0f;
#undef LS_LDS_COL2
    }
    const int visits = n_pad >> 6;
    const unsigned a01 = (unsigned)wave_sum_i32((visits - (int)m0) | ((visits - (int)m1) << 16));
    const unsigned a23 = (unsigned)wave_sum_i32((visits - (int)m2) | ((visits - (int)m3) << 16));
    const unsigned b01 = (unsigned)wave_sum_i32((visits - (int)p0) | ((visits - (int)p1) << 16));
    const unsigned b23 = (unsigned)wave_sum_i32((visits - (int)p2) | ((visits - (int)p3) << 16));
    cnt_a[0] = (int)(a01 & 0xffffu); cnt_a[1] = (int)(a01 >> 16); cnt_a[2] = (int)(a23 & 0xffffu); cnt_a[3] = (int)(a23 >> 16);
    cnt_b[0] = (int)(b01 & 0xffffu); cnt_b[1] = (int)(b01 >> 16); cnt_b[2] = (int)(b23 & 0xffffu); cnt_b[3] = (int)(b23 >> 16);
}
#undef LS_F32_LDS_STEP2
#undef LS_F32_LDS_HALF

// two WHOLE list columns (slots `slot`, `slot + 1`), every value 0 .. n-1 except the own squares ra, rb
__device__ __forceinline__ void nq_scan_pair_lds(const float *sR, int n_pad, int n, int ca, int cb, int ra, int rb, int conf_a,
                                                 int conf_b, int slot, unsigned long long &best, long long *srow_a,
                                                 long long *srow_b, long long cur, int lane)
{
    const unsigned long long hda = nq_key_hi(0, ca, n, slot), hua = nq_key_hi(1, ca, n, slot);
    const unsigned long long hdb = nq_key_hi(0, cb, n, slot + 1), hub = nq_key_hi(1, cb, n, slot + 1);
    for (int v = 0; v < n; v += 4) {
        int cnt_a[4], cnt_b[4];
        nq_count4x2_lds(sR, n_pad, lane, v, ca, cb, cnt_a, cnt_b);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int w = v + u;
            if (w < n && w != ra) {
                const int delta = cnt_a[u] - conf_a;
                best = u64min(best, nq_key(delta, w > ra ? hua : hda, w));
                if (srow_a != nullptr && lane == 0) srow_a[w] = cur + 2ll * delta;
            }
            if (w < n && w != rb) {
                const int delta = cnt_b[u] - conf_b;
                best = u64min(best, nq_key(delta, w > rb ? hub : hdb, w));
                if (srow_b != nullptr && lane == 0) srow_b[w] = cur + 2ll * delta;
            }
        }
    }
}

__device__ __forceinline__ void nq_scan_range_lds(const float *sR, int n_pad, int vbeg, int vend, int c,
                                                  int conf_old, unsigned long long khi,
                                                  unsigned long long &best, long long *scores_row,
                                                  long long cur, int lane)
{
    for (int v = vbeg; v < vend; v += 4) {
        int cnt[4];
        nq_count4_lds(sR, n_pad, lane, v, c, cnt);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (v + u < vend) {
                const int delta = cnt[u] - conf_old;
                best = u64min(best, nq_key(delta, khi, v + u));
                if (scores_row != nullptr && lane == 0) scores_row[v + u] = cur + 2ll * delta;
            }
        }
    }
}

constexpr int NQ_LDS_BLOCK = 1024;

__global__ __launch_bounds__(NQ_LDS_BLOCK) void nq_scan_lds_kernel(NqArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float s_dynf[]; // [n_pad] state as floats, then 16 x u64 best (n_pad % 512 == 0)
    constexpr int WPB = NQ_LDS_BLOCK / 64;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int restart = blockIdx.x % a.n_restarts;
    const int wchunk = (blockIdx.x / a.n_restarts) * WPB + wib;
    const int n_wchunks = (gridDim.x / a.n_restarts) * WPB;
    const int n = a.n;
    const int n_pad = (n + 511) & ~511;
    float *sR = s_dynf;
    unsigned long long *s_best = reinterpret_cast<unsigned long long *>(s_dynf + n_pad);
    const int *__restrict__ rows = a.rows + (long long)restart * n;
    const int *__restrict__ cols = a.cols + (long long)restart * a.max_cols;
    for (int j = threadIdx.x; j < n_pad; j += NQ_LDS_BLOCK) sR[j] = j < n ? (float)rows[j] : 1.0e9f; // padding never matches
    __syncthreads();
    int ncols = a.n_cols[restart];
    ncols = ncols < 0 ? 0 : (ncols > a.max_cols ? a.max_cols : ncols);
    const long long K = nq_k_limit((long long)ncols * n, n, a.window);
    const long long k0 = K * wchunk / n_wchunks;
    const long long k1 = K * (wchunk + 1) / n_wchunks;
    unsigned long long best = NQ_KEY_EMPTY;
    if (k0 < k1) {
        const long long cur = a.cur_scores[restart];
        int slot = (int)(k0 / n);
        int v0 = (int)(k0 - (long long)slot * n);
        long long k = k0;
        while (k < k1) {
            const long long left = k1 - k;
            const int ve = (left < (long long)(n - v0)) ? (int)(v0 + left) : n;
            const int c = cols[slot];
            const int rc = (unsigned)c < (unsigned)n ? rows[c] : -1;
            // this column and the next one whole, both inside the share and the window: one two-column walk
            if (v0 == 0 && left >= 2ll * n && (unsigned)rc < (unsigned)n && a.window - (long long)slot * (n - 1) >= 2ll * (n - 1)) {
                const int c2 = cols[slot + 1];
                const int rc2 = (unsigned)c2 < (unsigned)n ? rows[c2] : -1;
                if ((unsigned)rc2 < (unsigned)n) {
                    int cnt[4], cnt2[4];
                    nq_count4_lds(sR, n_pad, lane, rc, c, cnt);
                    nq_count4_lds(sR, n_pad, lane, rc2, c2, cnt2);
                    long long *srow = a.scores_out ? a.scores_out + ((long long)restart * a.max_cols + slot) * n : nullptr;
                    nq_scan_pair_lds(sR, n_pad, n, c, c2, rc, rc2, cnt[0] - 1, cnt2[0] - 1, slot, best, srow,
                                     srow ? srow + n : nullptr, cur, lane);
                    k += 2ll * n;
                    slot += 2;
                    continue;
                }
            }
            if ((unsigned)rc < (unsigned)n) {
                const long long base = (long long)slot * (n - 1);
                const NqRanges rg = nq_ranges(v0, ve, rc, base, a.window);
                if (rg.d1 > rg.d0 || rg.u1 > rg.u0) {
                    int cnt[4];
                    nq_count4_lds(sR, n_pad, lane, rc, c, cnt);
                    const int conf_old = cnt[0] - 1; // the own queen satisfies the product test once
                    long long *srow = a.scores_out
                                          ? a.scores_out + ((long long)restart * a.max_cols + slot) * n
                                          : nullptr;
                    nq_scan_range_lds(sR, n_pad, rg.d0, rg.d1, c, conf_old, nq_key_hi(0, c, n, slot), best,
                                      srow, cur, lane);
                    nq_scan_range_lds(sR, n_pad, rg.u0, rg.u1, c, conf_old, nq_key_hi(1, c, n, slot), best,
                                      srow, cur, lane);
                }
            }
            k += ve - v0;
            slot++;
            v0 = 0;
        }
    }
    if (lane == 0) s_best[wib] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = s_best[0];
        for (int i = 1; i < WPB; i++) b = u64min(b, s_best[i]);
        if (b != NQ_KEY_EMPTY) atomicMin(&a.best_keys[restart], b);
    }
}

// ==================================================================== TABLE
// tabs[restart] layout (int32): cntR[0..n) | cntA[0..2n) index r-j+n | cntB[0..2n) index r+j

__global__ __launch_bounds__(256) void nq_tables_build_kernel(const int *rows_all, int n, int *tabs,
                                                              int tab_stride)
{
    const int restart = blockIdx.y;
    const int *rows = rows_all + (long long)restart * n;
    int *t = tabs + (long long)restart * tab_stride;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const int r = rows[j];
        if ((unsigned)r < (unsigned)n) {
            atomicAdd(&t[r], 1);
            atomicAdd(&t[n + (r - j + n)], 1);
            atomicAdd(&t[3 * n + (r + j)], 1);
        }
    }
}

__global__ __launch_bounds__(256) void nq_table_eval_kernel(NqArgs a)
{
    __shared__ unsigned long long s_best[4];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int restart = blockIdx.x % a.n_restarts;
    const int chunk = blockIdx.x / a.n_restarts;
    const int n_chunks = gridDim.x / a.n_restarts;
    const int n = a.n;
    const int *__restrict__ rows = a.rows + (long long)restart * n;
    const int *__restrict__ cols = a.cols + (long long)restart * a.max_cols;
    const int *__restrict__ tR = a.tabs + (long long)restart * a.tab_stride;
    const int *__restrict__ tA = tR + n;
    const int *__restrict__ tB = tR + 3 * n;
    int ncols = a.n_cols[restart];
    ncols = ncols < 0 ? 0 : (ncols > a.max_cols ? a.max_cols : ncols);
    const long long K = nq_k_limit((long long)ncols * n, n, a.window);
    const long long k0 = K * chunk / n_chunks;
    const long long k1 = K * (chunk + 1) / n_chunks;
    unsigned long long best = NQ_KEY_EMPTY;
    if (k0 < k1) {
        const long long cur = a.cur_scores[restart];
        int slot = (int)(k0 / n);
        int v0 = (int)(k0 - (long long)slot * n);
        long long k = k0;
        while (k < k1) {
            const long long left = k1 - k;
            const int ve = (left < (long long)(n - v0)) ? (int)(v0 + left) : n;
            const int c = cols[slot];
            const int rc = (unsigned)c < (unsigned)n ? rows[c] : -1;
            if ((unsigned)rc < (unsigned)n) { // device-pointer callers are not validated: never index a table out of range
                const long long base = (long long)slot * (n - 1);
                const NqRanges rg = nq_ranges(v0, ve, rc, base, a.window);
                const int conf_old = tR[rc] + tA[rc - c + n] + tB[rc + c] - 3;
                long long *srow =
                    a.scores_out ? a.scores_out + ((long long)restart * a.max_cols + slot) * n : nullptr;
                const unsigned long long khd = nq_key_hi(0, c, n, slot), khu = nq_key_hi(1, c, n, slot);
                unsigned md = NQ_KEY32_EMPTY, mu = NQ_KEY32_EMPTY;
                for (int v = rg.d0 + (int)threadIdx.x; v < rg.d1; v += 256) {
                    const int delta = tR[v] + tA[v - c + n] + tB[v + c] - conf_old;
                    const unsigned k32 = nq_key32(delta, v);
                    md = k32 < md ? k32 : md;
                    if (srow) srow[v] = cur + 2ll * delta;
                }
                for (int v = rg.u0 + (int)threadIdx.x; v < rg.u1; v += 256) {
                    const int delta = tR[v] + tA[v - c + n] + tB[v + c] - conf_old;
                    const unsigned k32 = nq_key32(delta, v);
                    mu = k32 < mu ? k32 : mu;
                    if (srow) srow[v] = cur + 2ll * delta;
                }
                if (md != NQ_KEY32_EMPTY) best = u64min(best, nq_key_from32(md, khd));
                if (mu != NQ_KEY32_EMPTY) best = u64min(best, nq_key_from32(mu, khu));
            }
            k += ve - v0;
            slot++;
            v0 = 0;
        }
    }
    best = wave_min_u64(best);
    if (lane == 0) s_best[wib] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = u64min(u64min(s_best[0], s_best[1]), u64min(s_best[2], s_best[3]));
        if (b != NQ_KEY_EMPTY) atomicMin(&a.best_keys[restart], b);
    }
}

// TABLE with the histograms in LDS (n <= NQ_TABLE_LDS_MAX_N): every block builds its restart's three
// tables itself (3n LDS atomics on packed u16 counters - counts are <= n < 65536), then evaluates its
// share of the candidates from LDS.  One launch instead of clear + build + eval, and the 10n bytes of
// tables never leave the CU.
constexpr int NQ_TABLE_LDS_MAX_N = 12288; // 5n u16 = 120 KB


__global__ __launch_bounds__(256) void nq_table_lds_kernel(NqArgs a)
{
    extern __shared__ unsigned s_t16[]; // packed u16: cntR[0..n) | cntA[0..2n) | cntB[0..2n)
    __shared__ unsigned long long s_best[4];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int restart = blockIdx.x % a.n_restarts;
    const int chunk = blockIdx.x / a.n_restarts;
    const int n_chunks = gridDim.x / a.n_restarts;
    const int n = a.n;
    const int *__restrict__ rows = a.rows + (long long)restart * n;
    const int *__restrict__ cols = a.cols + (long long)restart * a.max_cols;
    int ncols = a.n_cols[restart];
    ncols = ncols < 0 ? 0 : (ncols > a.max_cols ? a.max_cols : ncols);
    const long long K = nq_k_limit((long long)ncols * n, n, a.window);
    const long long k0 = K * chunk / n_chunks;
    const long long k1 = K * (chunk + 1) / n_chunks;
    unsigned long long best = NQ_KEY_EMPTY;
    if (k0 < k1) { // block-uniform
        const int words = (5 * n + 1) >> 1;
        for (int i = threadIdx.x; i < words; i += 256) s_t16[i] = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += 256) {
            const int q = rows[j];
            if ((unsigned)q < (unsigned)n) {
                tab16_inc(s_t16, q);
                tab16_inc(s_t16, n + (q - j + n));
                tab16_inc(s_t16, 3 * n + (q + j));
            }
        }
        __syncthreads();
        const int oA = n, oB = 3 * n;
        const long long cur = a.cur_scores[restart];
        int slot = (int)(k0 / n);
        int v0 = (int)(k0 - (long long)slot * n);
        long long k = k0;
        while (k < k1) {
            const long long left = k1 - k;
            const int ve = (left < (long long)(n - v0)) ? (int)(v0 + left) : n;
            const int c = cols[slot];
            const int rc = (unsigned)c < (unsigned)n ? rows[c] : -1;
            if ((unsigned)rc < (unsigned)n) {
                const long long base = (long long)slot * (n - 1);
                const NqRanges rg = nq_ranges(v0, ve, rc, base, a.window);
                const int conf_old = tab16_get(s_t16, rc) + tab16_get(s_t16, oA + rc - c + n) + tab16_get(s_t16, oB + rc + c) - 3;
                long long *srow =
                    a.scores_out ? a.scores_out + ((long long)restart * a.max_cols + slot) * n : nullptr;
                const unsigned long long khd = nq_key_hi(0, c, n, slot), khu = nq_key_hi(1, c, n, slot);
                unsigned md = NQ_KEY32_EMPTY, mu = NQ_KEY32_EMPTY;
                for (int v = rg.d0 + (int)threadIdx.x; v < rg.d1; v += 256) {
                    const int delta = tab16_get(s_t16, v) + tab16_get(s_t16, oA + v - c + n) + tab16_get(s_t16, oB + v + c) - conf_old;
                    const unsigned k32 = nq_key32(delta, v);
                    md = k32 < md ? k32 : md;
                    if (srow) srow[v] = cur + 2ll * delta;
                }
                for (int v = rg.u0 + (int)threadIdx.x; v < rg.u1; v += 256) {
                    const int delta = tab16_get(s_t16, v) + tab16_get(s_t16, oA + v - c + n) + tab16_get(s_t16, oB + v + c) - conf_old;
                    const unsigned k32 = nq_key32(delta, v);
                    mu = k32 < mu ? k32 : mu;
                    if (srow) srow[v] = cur + 2ll * delta;
                }
                if (md != NQ_KEY32_EMPTY) best = u64min(best, nq_key_from32(md, khd));
                if (mu != NQ_KEY32_EMPTY) best = u64min(best, nq_key_from32(mu, khu));
            }
            k += ve - v0;
            slot++;
            v0 = 0;
        }
    }
    best = wave_min_u64(best);
    if (lane == 0) s_best[wib] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = u64min(u64min(s_best[0], s_best[1]), u64min(s_best[2], s_best[3]));
        if (b != NQ_KEY_EMPTY) atomicMin(&a.best_keys[restart], b);
    }
}

// TABLE, column-team formulation (256 <= n <= NQ_TABLE_LDS_MAX_N): the default for every launch that holds whole columns.
// Why the per-range loops above run at 0.3 of their issue bound (rocprofv3 counters, profiles/): a candidate is three
// DEPENDENT-on-nothing LDS look-ups, but the loops have data-dependent trip counts (the column is split at rows[c]), so
// only one candidate per lane is in flight; and every column starts with two dependent global loads (cols[slot],
// rows[c]).  Here instead
//  * the block first turns its slot range into LDS records {column, its row, conf_old} (one pass, loads in flight together);
//  * a TEAM of 256 lanes owns one (slot, part) unit at a time; lane t walks v = t + 256 i.  Within one slot the (score, board)
//    order is (count, v) -- every "down" value is below every "up" value -- so ONE 32-bit running minimum of (count << 15 | v)
//    per unit is exact; the direction bit is recovered from v when the minimum is widened to the 64-bit key;
//  * the walk is unrolled by four with immediate LDS offsets: 12 look-ups in flight, v_add3 + v_lshl_or + half a v_min3 per
//    candidate;
//  * the no-op (v == rows[c], tabu) is not tested per candidate: its key value is known ((conf_old + 3) << 15 | rows[c]), keys of
//    one unit are distinct, so a lane re-walks its values with the test only if its minimum IS that value.
// A slot cut by the window (at most one per restart) is scored by the last block with the per-range loops.
constexpr int NQ_REC_CAP = 256; // slot records per pass

template <int TEAMS, bool WANT_SCORES>
__global__ __launch_bounds__(256 * TEAMS) void nq_table_teams_kernel(NqArgs a, int parts)
{
    extern __shared__ unsigned s_t16[]; // packed u16: cntR[0..n) | cntA[0..2n) | cntB[0..2n), then int4 records [NQ_REC_CAP]
    __shared__ unsigned long long s_best[4 * TEAMS];
    constexpr int NT = 256 * TEAMS;
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    const int team = tid >> 8, tv = tid & 255;
    const int restart = blockIdx.x % a.n_restarts;
    const int chunk = blockIdx.x / a.n_restarts;
    const int n_chunks = gridDim.x / a.n_restarts;
    const int n = a.n;
    const int *__restrict__ rows = a.rows + (long long)restart * n;
    const int *__restrict__ cols = a.cols + (long long)restart * a.max_cols;
    int ncols = a.n_cols[restart];
    ncols = ncols < 0 ? 0 : (ncols > a.max_cols ? a.max_cols : ncols);
    // slots [0, n_whole) lie entirely inside the window; slot n_whole (if it exists) holds the window's last `rem` candidates
    long long n_whole = a.window / (n - 1);
    if (n_whole > ncols) n_whole = ncols;
    const long long rem = n_whole < ncols ? a.window - n_whole * (n - 1) : 0;
    const long long units = n_whole * parts;
    const long long u0 = units * chunk / n_chunks, u1 = units * (chunk + 1) / n_chunks;
    const bool do_partial = rem > 0 && chunk == n_chunks - 1;
    unsigned long long best = NQ_KEY_EMPTY;
    if (u0 < u1 || do_partial) { // block-uniform
        const int words = (((5 * n + 1) >> 1) + 3) & ~3;
        int4 *s_rec = reinterpret_cast<int4 *>(s_t16 + words);
        for (int i = tid; i < words; i += NT) s_t16[i] = 0;
        __syncthreads();
        for (int j = tid; j < n; j += NT) {
            const int q = rows[j];
            if ((unsigned)q < (unsigned)n) {
                tab16_inc(s_t16, q);
                tab16_inc(s_t16, n + (q - j + n));
                tab16_inc(s_t16, 3 * n + (q + j));
            }
        }
        __syncthreads();
        const unsigned short *t16 = reinterpret_cast<const unsigned short *>(s_t16);
        const int oA = n, oB = 3 * n;
        const long long cur = WANT_SCORES ? a.cur_scores[restart] : 0;
        const int per_part = (((n + 255) >> 8) + parts - 1) / parts; // 256-value steps per part
        const int slot_lo = (int)(u0 / parts);
        const int slot_hi = u0 < u1 ? (int)((u1 - 1) / parts) : slot_lo - 1; // inclusive
        for (int pass_lo = slot_lo; pass_lo <= slot_hi; pass_lo += NQ_REC_CAP) {
            const int pass_hi = pass_lo + NQ_REC_CAP - 1 < slot_hi ? pass_lo + NQ_REC_CAP - 1 : slot_hi;
            __syncthreads(); // the previous pass's records are no longer read
            for (int sl = pass_lo + tid; sl <= pass_hi; sl += NT) {
                const int c = cols[sl];
                const int rc = (unsigned)c < (unsigned)n ? rows[c] : -1;
                int conf_old = 0;
                if ((unsigned)rc < (unsigned)n) conf_old = (int)t16[rc] + (int)t16[oA + rc - c + n] + (int)t16[oB + rc + c] - 3;
                s_rec[sl - pass_lo] = make_int4(c, (unsigned)rc < (unsigned)n ? rc : -1, conf_old, 0);
            }
            __syncthreads();
            long long ua = (long long)pass_lo * parts, ub = (long long)(pass_hi + 1) * parts;
            ua = ua < u0 ? u0 : ua;
            ub = ub > u1 ? u1 : ub;
            for (long long u = ua + team; u < ub; u += TEAMS) { // team-uniform
                const int slot = (int)(u / parts), part = (int)(u - (long long)slot * parts);
                const int4 rec = s_rec[slot - pass_lo];
                const int c = rec.x, rc = rec.y, conf_old = rec.z;
                if (rc < 0) continue; // unvalidated device-pointer input: the slot is skipped (nq_slot_scored)
                const int i0 = part * per_part;
                int i1 = i0 + per_part;
                const int steps = (n + 255) >> 8;
                i1 = i1 > steps ? steps : i1;
                const int full = n >> 8; // steps in which every lane's v is < n
                const int i1f = i1 < full ? i1 : full;
                const unsigned short *pR = t16 + tv, *pA = t16 + oA + tv - c + n, *pB = t16 + oB + tv + c;
                long long *srow = WANT_SCORES ? a.scores_out + ((long long)restart * a.max_cols + slot) * n : nullptr;
                unsigned m = NQ_KEY32_EMPTY;
                int i = i0;
#define LS_CAND(J)                                                                                              \
    const int v##J = tv + ((i + J) << 8);                                                                       \
    const unsigned k##J = (((unsigned)pR[(i + J) << 8] + (unsigned)pA[(i + J) << 8] + (unsigned)pB[(i + J) << 8]) << 15) | (unsigned)v##J;
                for (; i + 4 <= i1f; i += 4) {
                    LS_CAND(0) LS_CAND(1) LS_CAND(2) LS_CAND(3)
                    unsigned m01 = k0 < k1 ? k0 : k1, m23 = k2 < k3 ? k2 : k3;
                    m01 = m01 < m23 ? m01 : m23;
                    m = m01 < m ? m01 : m;
                    if (WANT_SCORES) {
                        if (v0 != rc) srow[v0] = cur + 2ll * ((int)(k0 >> 15) - conf_old);
                        if (v1 != rc) srow[v1] = cur + 2ll * ((int)(k1 >> 15) - conf_old);
                        if (v2 != rc) srow[v2] = cur + 2ll * ((int)(k2 >> 15) - conf_old);
                        if (v3 != rc) srow[v3] = cur + 2ll * ((int)(k3 >> 15) - conf_old);
                    }
                }
                for (; i < i1; i++) {
                    const int v0 = tv + (i << 8);
                    if (v0 < n) {
                        const unsigned k0 = (((unsigned)pR[i << 8] + (unsigned)pA[i << 8] + (unsigned)pB[i << 8]) << 15) | (unsigned)v0;
                        m = k0 < m ? k0 : m;
                        if (WANT_SCORES && v0 != rc) srow[v0] = cur + 2ll * ((int)(k0 >> 15) - conf_old);
                    }
                }
#undef LS_CAND
                const unsigned noop = ((unsigned)(conf_old + 3) << 15) | (unsigned)rc;
                if (m == noop) { // this lane's minimum is the tabu square itself: walk again without it
                    m = NQ_KEY32_EMPTY;
                    for (int i2 = i0; i2 < i1; i2++) {
                        const int v0 = tv + (i2 << 8);
                        if (v0 < n && v0 != rc) {
                            const unsigned k0 = (((unsigned)pR[i2 << 8] + (unsigned)pA[i2 << 8] + (unsigned)pB[i2 << 8]) << 15) | (unsigned)v0;
                            m = k0 < m ? k0 : m;
                        }
                    }
                }
                if (m != NQ_KEY32_EMPTY) {
                    const int v = (int)(m & 0x7fffu);
                    best = u64min(best, nq_key((int)(m >> 15) - conf_old, nq_key_hi(v > rc ? 1 : 0, c, n, slot), v));
                }
            }
        }
        if (do_partial) { // the slot the window cuts: per-range loops, all lanes of the block
            const int slot = (int)n_whole;
            const int c = cols[slot];
            const int rc = (unsigned)c < (unsigned)n ? rows[c] : -1;
            if ((unsigned)rc < (unsigned)n) {
                const NqRanges rg = nq_ranges(0, n, rc, (long long)slot * (n - 1), a.window);
                const int conf_old = (int)t16[rc] + (int)t16[oA + rc - c + n] + (int)t16[oB + rc + c] - 3;
                long long *srow = WANT_SCORES ? a.scores_out + ((long long)restart * a.max_cols + slot) * n : nullptr;
                unsigned m = NQ_KEY32_EMPTY;
                for (int v = rg.d0 + tid; v < rg.d1; v += NT) {
                    const unsigned k = (((unsigned)t16[v] + (unsigned)t16[oA + v - c + n] + (unsigned)t16[oB + v + c]) << 15) | (unsigned)v;
                    m = k < m ? k : m;
                    if (WANT_SCORES) srow[v] = cur + 2ll * ((int)(k >> 15) - conf_old);
                }
                for (int v = rg.u0 + tid; v < rg.u1; v += NT) {
                    const unsigned k = (((unsigned)t16[v] + (unsigned)t16[oA + v - c + n] + (unsigned)t16[oB + v + c]) << 15) | (unsigned)v;
                    m = k < m ? k : m;
                    if (WANT_SCORES) srow[v] = cur + 2ll * ((int)(k >> 15) - conf_old);
                }
                if (m != NQ_KEY32_EMPTY) {
                    const int v = (int)(m & 0x7fffu);
                    best = u64min(best, nq_key((int)(m >> 15) - conf_old, nq_key_hi(v > rc ? 1 : 0, c, n, slot), v));
                }
            }
        }
    }
    best = wave_min_u64(best);
    if (lane == 0) s_best[wib] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned long long b = s_best[0];
        for (int i = 1; i < 4 * TEAMS; i++) b = u64min(b, s_best[i]);
        if (b != NQ_KEY_EMPTY) atomicMin(&a.best_keys[restart], b);
    }
}


// the whole block's work of nq_table_pairs_kernel (also the fall-back of nq_table_quads_kernel for boards its 8-bit counts cannot hold)
template <int NT, int TL, bool WANT_SCORES>
__device__ __forceinline__ void nq_pairs_block(const NqArgs &a, int parts, int whole_max, unsigned *s_t16, unsigned long long *s_best)
{
    // s_t16: packed u16: R[n] | R1[n] | A[2n+2] | B[2n+2], then int4 records [NQ_REC_CAP]
    constexpr int TEAMS = NT / TL;            // a team = TL lanes (one wavefront or four) walking one column
    constexpr int SH = TL == 64 ? 6 : 8;      // log2(TL)
    static_assert(TL == 64 || TL == 256, "team width");
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    const int team = __builtin_amdgcn_readfirstlane(tid >> SH), tv = tid & (TL - 1);
    const int restart = blockIdx.x % a.n_restarts;
    const int chunk = blockIdx.x / a.n_restarts;
    const int n_chunks = gridDim.x / a.n_restarts;
    const int n = a.n;
    const int *__restrict__ rows = a.rows + (long long)restart * n;
    const int *__restrict__ cols = a.cols + (long long)restart * a.max_cols;
    int ncols = a.n_cols[restart];
    ncols = ncols < 0 ? 0 : (ncols > a.max_cols ? a.max_cols : ncols);
    // (32-bit arithmetic: units <= 12288 columns x 24 parts and at most 1024 chunks -- a 64-bit division costs ~150 scalar instructions)
    const int n_whole = whole_max < ncols ? whole_max : ncols;
    const long long rem = n_whole < ncols ? a.window - (long long)n_whole * (n - 1) : 0;
    const unsigned units = (unsigned)n_whole * (unsigned)parts;
    const long long u0 = units * (unsigned)chunk / (unsigned)n_chunks, u1 = units * (unsigned)(chunk + 1) / (unsigned)n_chunks;
    const bool do_partial = rem > 0 && chunk == n_chunks - 1;
    unsigned long long best = NQ_KEY_EMPTY;
    LS_STAMP(0);
    if (u0 < u1 || do_partial) { // block-uniform
        const int oR1 = n, oA = 2 * n, oB = 4 * n + 2; // u16 indices (all even)
        const int words = (((6 * n + 4) >> 1) + 3) & ~3;
        int4 *s_rec = reinterpret_cast<int4 *>(s_t16 + words);
        // every global read of the set-up is issued before the first LDS store: the board (NT == 1024: into registers) and
        // the first pass's slot columns with their rows -- one memory latency instead of three in a row
        const int slot_lo0 = (int)((unsigned)u0 / (unsigned)parts);
        const int slot_hi0 = u0 < u1 ? (int)((unsigned)(u1 - 1) / (unsigned)parts) : slot_lo0 - 1;
        int pre_c = -1, pre_rc = -1;
        if (tid < NQ_REC_CAP && slot_lo0 + tid <= slot_hi0) {
            pre_c = cols[slot_lo0 + tid];
            pre_rc = (unsigned)pre_c < (unsigned)n ? rows[pre_c] : -1;
        }
        constexpr int QV = NT == 1024 ? NQ_TABLE_LDS_MAX_N / 1024 : 1;
        int qv[QV];
        if constexpr (NT == 1024) {
#pragma unroll
            for (int kk = 0; kk < QV; kk++) {
                const int j = tid + 1024 * kk;
                qv[kk] = j < n ? rows[j] : -1;
            }
        }
        LS_STAMP(1);
        for (int i = tid; i < words; i += NT) s_t16[i] = 0;
        __syncthreads();
        LS_STAMP(2);
        auto count_queen = [&](int j, int q) {
            if ((unsigned)q < (unsigned)n) {
                tab16_inc(s_t16, q);
                if (q > 0) tab16_inc(s_t16, oR1 + q - 1);
                tab16_inc(s_t16, oA + (q - j + n));
                tab16_inc(s_t16, oB + (q + j));
            }
        };
        if constexpr (NT == 1024) {
#pragma unroll
            for (int kk = 0; kk < QV; kk++) count_queen(tid + 1024 * kk, qv[kk]);
        } else {
            for (int j = tid; j < n; j += NT) count_queen(j, rows[j]);
        }
        if (tid == 0) atomicAdd(&s_t16[(oR1 + n - 1) >> 1], NQ_PAIR_SENTINEL << (16 * ((oR1 + n - 1) & 1)));
        __syncthreads();
        LS_STAMP(3);
        const unsigned short *t16 = reinterpret_cast<const unsigned short *>(s_t16);
        const long long cur = WANT_SCORES ? a.cur_scores[restart] : 0;
        const int steps = (n + 2 * TL - 1) >> (SH + 1);    // 2 TL values (TL pairs) per step
        const int per_part = (steps + parts - 1) / parts;
        const int slot_lo = slot_lo0, slot_hi = slot_hi0; // inclusive
        for (int pass_lo = slot_lo; pass_lo <= slot_hi; pass_lo += NQ_REC_CAP) {
            const int pass_hi = pass_lo + NQ_REC_CAP - 1 < slot_hi ? pass_lo + NQ_REC_CAP - 1 : slot_hi;
            __syncthreads(); // the previous pass's records are no longer read
            for (int sl = pass_lo + tid; sl <= pass_hi; sl += NT) {
                const bool pre = pass_lo == slot_lo0; // first pass: loaded up front (NQ_REC_CAP <= NT: one slot per lane)
                const int c = pre ? pre_c : cols[sl];
                const int rc = pre ? pre_rc : ((unsigned)c < (unsigned)n ? rows[c] : -1);
                int conf_old = 0;
                if ((unsigned)rc < (unsigned)n) conf_old = (int)t16[rc] + (int)t16[oA + rc - c + n] + (int)t16[oB + rc + c] - 3;
                s_rec[sl - pass_lo] = make_int4(c, (unsigned)rc < (unsigned)n ? rc : -1, conf_old, 0);
            }
            __syncthreads();
            LS_STAMP(4);
            long long ua = (long long)pass_lo * parts, ub = (long long)(pass_hi + 1) * parts;
            ua = ua < u0 ? u0 : ua;
            ub = ub > u1 ? u1 : ub;
            // team-uniform (team is an SGPR).  (slot, part) of unit u are carried along instead of divided out per unit: a
            // 64-bit division is ~150 scalar instructions, once per 32-step column walk it was a sixth of a wavefront's time
            const unsigned first = (unsigned)(ua + team);
            int slot = (int)(first / (unsigned)parts), part = (int)(first - (unsigned)slot * (unsigned)parts);
            const int dq = TEAMS / parts, dr = TEAMS - dq * parts;
            for (long long u = ua + team; u < ub; u += TEAMS, slot += dq, part += dr) {
                if (part >= parts) { part -= parts; slot++; }
                const int4 rec = s_rec[slot - pass_lo];
                const int c = __builtin_amdgcn_readfirstlane(rec.x), rc = __builtin_amdgcn_readfirstlane(rec.y),
                          conf_old = __builtin_amdgcn_readfirstlane(rec.z);
                if (rc < 0) continue; // unvalidated device-pointer input: the slot is skipped (nq_slot_scored)
                long long *srow = WANT_SCORES ? a.scores_out + ((long long)restart * a.max_cols + slot) * n : nullptr;
                nq_pairs_unit<TL, WANT_SCORES>(s_t16, n, tv, c, rc, conf_old, slot, part * per_part, per_part, srow, cur, best);
            }
        }
        if (do_partial) { // the slot the window cuts: per-range loops, all lanes of the block
            const int slot = (int)n_whole;
            const int c = cols[slot];
            const int rc = (unsigned)c < (unsigned)n ? rows[c] : -1;
            if ((unsigned)rc < (unsigned)n) {
                const NqRanges rg = nq_ranges(0, n, rc, (long long)slot * (n - 1), a.window);
                const int conf_old = (int)t16[rc] + (int)t16[oA + rc - c + n] + (int)t16[oB + rc + c] - 3;
                long long *srow = WANT_SCORES ? a.scores_out + ((long long)restart * a.max_cols + slot) * n : nullptr;
                unsigned m = NQ_KEY32_EMPTY;
                for (int v = rg.d0 + tid; v < rg.d1; v += NT) {
                    const unsigned k = (((unsigned)t16[v] + (unsigned)t16[oA + v - c + n] + (unsigned)t16[oB + v + c]) << 15) | (unsigned)v;
                    m = k < m ? k : m;
                    if (WANT_SCORES) srow[v] = cur + 2ll * ((int)(k >> 15) - conf_old);
                }
                for (int v = rg.u0 + tid; v < rg.u1; v += NT) {
                    const unsigned k = (((unsigned)t16[v] + (unsigned)t16[oA + v - c + n] + (unsigned)t16[oB + v + c]) << 15) | (unsigned)v;
                    m = k < m ? k : m;
                    if (WANT_SCORES) srow[v] = cur + 2ll * ((int)(k >> 15) - conf_old);
                }
                if (m != NQ_KEY32_EMPTY) {
                    const int v = (int)(m & 0x7fffu);
                    best = u64min(best, nq_key((int)(m >> 15) - conf_old, nq_key_hi(v > rc ? 1 : 0, c, n, slot), v));
                }
            }
        }
    }
    LS_STAMP(5);
    best = wave_min_u64(best);
    if (lane == 0) s_best[wib] = best;
    __syncthreads();
    LS_STAMP(6);
    if (tid == 0) {
        unsigned long long b = s_best[0];
        for (int i = 1; i < NT / 64; i++) b = u64min(b, s_best[i]);
        if (b != NQ_KEY_EMPTY) atomicMin(&a.best_keys[restart], b);
    }
}

template <int NT, int TL, bool WANT_SCORES>
__global__ __launch_bounds__(NT) void nq_table_pairs_kernel(NqArgs a, int parts, int whole_max /* min(window / (n - 1), max_cols): divided on the host */)
{
    extern __shared__ unsigned s_t16[];
    __shared__ unsigned long long s_best[NT / 64];
    nq_pairs_block<NT, TL, WANT_SCORES>(a, parts, whole_max, s_t16, s_best);
}

// TABLE, FOUR candidates per lane and LDS look-up (n a multiple of 256, 512 <= n <= NQ_TABLE_LDS_MAX_N, no score dump): the pairs
// kernel above is bound by the LDS array and by half-rate VALU work in equal parts (three 4-byte look-ups and ~8 instructions
// per two candidates).  Here the counts are 8-BIT: a lane owns the quad (v .. v + 3), one ds_read_b32 per table returns all
// four candidates' counts and ONE v_add3_u32 adds them as packed bytes (a count is at most NQ_QUAD_MAX = 50 here, so a sum
// stays below 256 also beside the sentinel: no carry between the bytes).  The per-lane running minimum does not unpack them
// either: two v_perm_b32 spread the four sums over two registers as 16-bit fields (sum << 8 | step) and two v_pk_min_u16 keep,
// per byte position, the smallest (sum, first step) -- five instructions per FOUR candidates; positions are rebuilt once per
// column.  The three quads are 4-byte aligned when v = c (mod 4) (then v - c + n is a multiple of 4), so the row table is kept
// in four copies shifted by 0..3 entries and the B diagonal table in two (v + c = 2c mod 4 is 0 or 2): 10n bytes of LDS against
// 12n for the u16 pairs.  Squares v >= n of a lane's last quad read a sentinel count that can never win; squares v < c mod 4
// belong to no quad and are scored by the first lanes of the column's first part.
// A board with more than NQ_QUAD_MAX queens on one row or diagonal does not fit 8-bit sums: every block finds that out while it
// builds its tables (the old value of each increment) and takes nq_pairs_block instead -- block-uniform, same results.
constexpr unsigned NQ_QUAD_MAX = 50;        // largest count the quads formulation accepts
constexpr unsigned NQ_QUAD_SENTINEL = 151;  // > 3 NQ_QUAD_MAX (never the minimum), and sentinel + 2 NQ_QUAD_MAX < 256 (no carry)

__device__ __forceinline__ unsigned tab8_inc(unsigned *t, int idx) // returns the byte's old value
{
    const unsigned old = atomicAdd(&t[idx >> 2], 1u << (8 * (idx & 3)));
    return (old >> (8 * (idx & 3))) & 0xffu;
}

template <int NT>
__global__ __launch_bounds__(NT) void nq_table_quads_kernel(NqArgs a, int parts, int whole_max)
{
    // bytes: R0[n] | R1[n] | R2[n] | R3[n] (Rs[i] = cntR[i + s]) | A[2n] | B0[2n] | B2[2n] (B2[i] = cntB[i + 2]), then int4 records
    extern __shared__ unsigned s_t8[];
    __shared__ unsigned long long s_best[NT / 64];
    __shared__ int s_overflow;
    constexpr int TEAMS = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    const int team = __builtin_amdgcn_readfirstlane(wib);
    const int restart = blockIdx.x % a.n_restarts;
    const int chunk = blockIdx.x / a.n_restarts;
    const int n_chunks = gridDim.x / a.n_restarts;
    const int n = a.n;
    const int *__restrict__ rows = a.rows + (long long)restart * n;
    const int *__restrict__ cols = a.cols + (long long)restart * a.max_cols;
    int ncols = a.n_cols[restart];
    ncols = ncols < 0 ? 0 : (ncols > a.max_cols ? a.max_cols : ncols);
    const int n_whole = whole_max < ncols ? whole_max : ncols;
    const long long rem = n_whole < ncols ? a.window - (long long)n_whole * (n - 1) : 0;
    const unsigned units = (unsigned)n_whole * (unsigned)parts;
    const long long u0 = units * (unsigned)chunk / (unsigned)n_chunks, u1 = units * (unsigned)(chunk + 1) / (unsigned)n_chunks;
    const bool do_partial = rem > 0 && chunk == n_chunks - 1;
    unsigned long long best = NQ_KEY_EMPTY;
    if (!(u0 < u1 || do_partial)) return; // block-uniform: nothing to do (and nothing to contribute to the minimum)
    const int oA = 4 * n, oB0 = 6 * n, oB2 = 8 * n; // byte offsets (all multiples of 4)
    const int words = (10 * n) >> 2;
    int4 *s_rec = reinterpret_cast<int4 *>(s_t8 + words);
    const int slot_lo = (int)((unsigned)u0 / (unsigned)parts);
    const int slot_hi = u0 < u1 ? (int)((unsigned)(u1 - 1) / (unsigned)parts) : slot_lo - 1; // inclusive
    // every global read of the set-up is issued before the first LDS store: the board into registers, the first pass's columns
    int pre_c = -1, pre_rc = -1;
    if (tid < NQ_REC_CAP && slot_lo + tid <= slot_hi) {
        pre_c = cols[slot_lo + tid];
        pre_rc = (unsigned)pre_c < (unsigned)n ? rows[pre_c] : -1;
    }
    constexpr int QV = NQ_TABLE_LDS_MAX_N / NT;
    int qv[QV];
#pragma unroll
    for (int kk = 0; kk < QV; kk++) {
        const int j = tid + NT * kk;
        qv[kk] = j < n ? rows[j] : -1;
    }
    for (int i = tid; i < words; i += NT) s_t8[i] = 0;
    if (tid == 0) s_overflow = 0;
    __syncthreads();
    {
        unsigned worst = 0;
#pragma unroll
        for (int kk = 0; kk < QV; kk++) {
            const int j = tid + NT * kk, q = qv[kk];
            if ((unsigned)q < (unsigned)n) {
#pragma unroll
                for (int sft = 0; sft < 4; sft++)
                    if (q >= sft) { const unsigned o = tab8_inc(s_t8, sft * n + q - sft); worst = o > worst ? o : worst; }
                unsigned o = tab8_inc(s_t8, oA + (q - j + n));
                worst = o > worst ? o : worst;
                o = tab8_inc(s_t8, oB0 + (q + j));
                worst = o > worst ? o : worst;
                if (q + j >= 2) tab8_inc(s_t8, oB2 + (q + j - 2));
            }
        }
        if (worst >= NQ_QUAD_MAX) s_overflow = 1; // (old value >= MAX: the count reached MAX + 1)
        // squares beyond the board, seen by the shifted row copies: Rs[i] for i + s >= n
        if (tid < 6) {
            const int sft = tid < 1 ? 1 : (tid < 3 ? 2 : 3), k = tid < 1 ? 0 : (tid < 3 ? tid - 1 : tid - 3);
            const int idx = sft * n + (n - sft) + k;
            atomicAdd(&s_t8[idx >> 2], NQ_QUAD_SENTINEL << (8 * (idx & 3)));
        }
    }
    __syncthreads();
    if (s_overflow) { // block-uniform: this board needs 16-bit counts
        __syncthreads();
        nq_pairs_block<NT, 64, false>(a, parts, whole_max, s_t8, s_best);
        return;
    }
    const unsigned char *t8 = reinterpret_cast<const unsigned char *>(s_t8);
    const int steps = n >> 8;                    // 64 quads = 256 values per step
    const int per_part = (steps + parts - 1) / parts;
    for (int pass_lo = slot_lo; pass_lo <= slot_hi; pass_lo += NQ_REC_CAP) {
        const int pass_hi = pass_lo + NQ_REC_CAP - 1 < slot_hi ? pass_lo + NQ_REC_CAP - 1 : slot_hi;
        __syncthreads(); // the previous pass's records are no longer read
        for (int sl = pass_lo + tid; sl <= pass_hi; sl += NT) {
            const bool pre = pass_lo == slot_lo;
            const int c = pre ? pre_c : cols[sl];
            const int rc = pre ? pre_rc : ((unsigned)c < (unsigned)n ? rows[c] : -1);
            int conf_old = 0;
            if ((unsigned)rc < (unsigned)n) conf_old = (int)t8[rc] + (int)t8[oA + rc - c + n] + (int)t8[oB0 + rc + c] - 3;
            s_rec[sl - pass_lo] = make_int4(c, (unsigned)rc < (unsigned)n ? rc : -1, conf_old, 0);
        }
        __syncthreads();
        long long ua = (long long)pass_lo * parts, ub = (long long)(pass_hi + 1) * parts;
        ua = ua < u0 ? u0 : ua;
        ub = ub > u1 ? u1 : ub;
        const unsigned first = (unsigned)(ua + team);
        int slot = (int)(first / (unsigned)parts), part = (int)(first - (unsigned)slot * (unsigned)parts);
        const int dq = TEAMS / parts, dr = TEAMS - dq * parts;
        for (long long u = ua + team; u < ub; u += TEAMS, slot += dq, part += dr) { // wavefront-uniform
            if (part >= parts) { part -= parts; slot++; }
            const int4 rec = s_rec[slot - pass_lo];
            const int c = __builtin_amdgcn_readfirstlane(rec.x), rc = __builtin_amdgcn_readfirstlane(rec.y),
                      conf_old = __builtin_amdgcn_readfirstlane(rec.z);
            if (rc < 0) continue; // unvalidated device-pointer input: the slot is skipped (nq_slot_scored)
            const int sft = c & 3, bsh = (2 * c) & 3; // v = sft (mod 4); B copy shifted by bsh (0 or 2)
            const int i0 = part * per_part;
            int i1 = i0 + per_part;
            i1 = i1 > steps ? steps : i1;
            // dword pointers: lane `lane` owns the quad starting at v = sft + 4 lane (+ 256 i)
            const unsigned *pR = s_t8 + ((sft * n) >> 2) + lane;                                   // R[v ..] = Rs[v - sft ..]
            const unsigned *pA = s_t8 + ((oA + sft - c + n) >> 2) + lane;                          // A[v - c + n ..]
            const unsigned *pB = s_t8 + (((bsh ? oB2 : oB0) + sft + c - bsh) >> 2) + lane;         // B[v + c ..]
            us2 me = as_us2(0xffffffffu), mo = as_us2(0xffffffffu); // per byte position: the smallest (sum << 8 | step)
            int i = i0;
#define LS_QUAD(J)                                                                                                 \
    {                                                                                                              \
        const unsigned sum = pR[(i + J) << 6] + pA[(i + J) << 6] + pB[(i + J) << 6];                               \
        me = pk_min_u16_raw(me, as_us2(__builtin_amdgcn_perm(sum, (unsigned)(i + J), 0x06000400u)));               \
        mo = pk_min_u16_raw(mo, as_us2(__builtin_amdgcn_perm(sum, (unsigned)(i + J), 0x07000500u)));               \
    }
            for (; i + 4 <= i1; i += 4) { LS_QUAD(0) LS_QUAD(1) LS_QUAD(2) LS_QUAD(3) }
            for (; i < i1; i++) LS_QUAD(0)
#undef LS_QUAD
            // the lane's minimum (count, v): per byte position p the field is (sum << 8 | first step with that sum); v = sft + 4 lane + 256 step + p
            const unsigned f0 = as_u32(me) & 0xffffu, f2 = as_u32(me) >> 16, f1 = as_u32(mo) & 0xffffu, f3 = as_u32(mo) >> 16;
            const int vb = sft + 4 * lane;
            auto key_of = [&](unsigned f, int p) { return ((f >> 8) << 15) | (unsigned)(vb + (int)((f & 0xffu) << 8) + p); };
            unsigned m = NQ_KEY32_EMPTY;
            if (i0 < i1) {
                const unsigned k0 = key_of(f0, 0), k1 = key_of(f1, 1), k2 = key_of(f2, 2), k3 = key_of(f3, 3);
                const unsigned m01 = k0 < k1 ? k0 : k1, m23 = k2 < k3 ? k2 : k3;
                m = m01 < m23 ? m01 : m23;
            }
            const unsigned noop = ((unsigned)(conf_old + 3) << 15) | (unsigned)rc;
            if (m == noop) { // this lane's minimum is the tabu square itself: walk again without it
                m = NQ_KEY32_EMPTY;
                for (int i2 = i0; i2 < i1; i2++) {
                    const unsigned sum = pR[i2 << 6] + pA[i2 << 6] + pB[i2 << 6];
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const int v = vb + (i2 << 8) + p;
                        const unsigned k = (((sum >> (8 * p)) & 0xffu) << 15) | (unsigned)v;
                        if (v != rc) m = k < m ? k : m;
                    }
                }
            }
            // squares v < sft belong to no quad: the first lanes of the column's first part score them (v = lane)
            if (i0 == 0 && lane < sft && lane != rc) {
                const unsigned k = (((unsigned)t8[lane] + (unsigned)t8[oA + lane - c + n] + (unsigned)t8[oB0 + lane + c]) << 15) | (unsigned)lane;
                m = k < m ? k : m;
            }
            if (m != NQ_KEY32_EMPTY && (m & 0x7fffu) < (unsigned)n) { // (a lane whose only squares lie beyond the board holds a sentinel sum)
                const int v = (int)(m & 0x7fffu);
                best = u64min(best, nq_key((int)(m >> 15) - conf_old, nq_key_hi(v > rc ? 1 : 0, c, n, slot), v));
            }
        }
    }
    if (do_partial) { // the slot the window cuts: per-range loops, all lanes of the block
        const int slot = (int)n_whole;
        const int c = cols[slot];
        const int rc = (unsigned)c < (unsigned)n ? rows[c] : -1;
        if ((unsigned)rc < (unsigned)n) {
            const NqRanges rg = nq_ranges(0, n, rc, (long long)slot * (n - 1), a.window);
            const int conf_old = (int)t8[rc] + (int)t8[oA + rc - c + n] + (int)t8[oB0 + rc + c] - 3;
            unsigned m = NQ_KEY32_EMPTY;
            for (int v = rg.d0 + tid; v < rg.d1; v += NT) {
                const unsigned k = (((unsigned)t8[v] + (unsigned)t8[oA + v - c + n] + (unsigned)t8[oB0 + v + c]) << 15) | (unsigned)v;
                m = k < m ? k : m;
            }
            for (int v = rg.u0 + tid; v < rg.u1; v += NT) {
                const unsigned k = (((unsigned)t8[v] + (unsigned)t8[oA + v - c + n] + (unsigned)t8[oB0 + v + c]) << 15) | (unsigned)v;
                m = k < m ? k : m;
            }
            if (m != NQ_KEY32_EMPTY) {
                const int v = (int)(m & 0x7fffu);
                best = u64min(best, nq_key((int)(m >> 15) - conf_old, nq_key_hi(v > rc ? 1 : 0, c, n, slot), v));
            }
        }
    }
    best = wave_min_u64(best);
    if (lane == 0) s_best[wib] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned long long b = s_best[0];
        for (int i = 1; i < NT / 64; i++) b = u64min(b, s_best[i]);
        if (b != NQ_KEY_EMPTY) atomicMin(&a.best_keys[restart], b);
    }
}

// get_col_scores + sum by tables (examples/nqueens/src/lib.rs:74-87,130-139).
__global__ __launch_bounds__(256) void nq_col_scores_kernel(const int *rows_all, int n, const int *tabs,
                                                            int tab_stride, long long *col_scores,
                                                            long long *totals)
{
    __shared__ long long s_sum[4];
    const int restart = blockIdx.y;
    const int *rows = rows_all + (long long)restart * n;
    const int *tR = tabs + (long long)restart * tab_stride;
    const int *tA = tR + n, *tB = tR + 3 * n;
    long long local = 0;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const int r = rows[j];
        long long s = 0;
        if ((unsigned)r < (unsigned)n) s = (long long)tR[r] + tA[r - j + n] + tB[r + j] - 3;
        if (col_scores) col_scores[(long long)restart * n + j] = s;
        local += s;
    }
    // 64-bit wave sum via two 32-bit halves is overkill: per-thread partials are < 2^31
    int lo = wave_sum_i32((int)local);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = lo;
    __syncthreads();
    if (threadIdx.x == 0 && totals) {
        unsigned long long t = (unsigned long long)(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
        atomicAdd(reinterpret_cast<unsigned long long *>(&totals[restart]), t);
    }
}

// ================================================================= finalize
// One block per restart: decode the winning key into the result record, put the key back to EMPTY for
// the next call (so no separate clearing launch is needed), and - when asked - write the winner board
// (rows with the selected move applied: MoveIterator, lib.rs:226-228) for the fused column-score pass.

__global__ __launch_bounds__(256) void nq_finalize_kernel(NqArgs a, ls_nq_result *results, int *next_rows)
{
    __shared__ ls_nq_result s_out;
    __shared__ int s_cnt[4];
    const int r = blockIdx.x;
    const int n = a.n;
    {   // n_scored: what the eval kernels really scored (they skip out-of-range columns)
        int ncols = a.n_cols[r];
        ncols = ncols < 0 ? 0 : (ncols > a.max_cols ? a.max_cols : ncols);
        long long slots = ncols;
        if (n > 1 && a.window / (n - 1) < slots) slots = a.window / (n - 1) + 1;
        if (n <= 1 || a.window <= 0) slots = 0;
        int cnt = 0; // <= 32768 * 32767 < 2^31
        for (int sl = threadIdx.x; sl < (int)slots; sl += 256) {
            const int c = a.cols[(long long)r * a.max_cols + sl];
            const int rc = (unsigned)c < (unsigned)n ? a.rows[(long long)r * n + c] : -1;
            cnt += nq_slot_scored(sl, c, rc, n, a.window);
        }
        cnt = wave_sum_i32(cnt);
        if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ls_nq_result out;
        out.n_scored = (long long)s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        const unsigned long long key = a.best_keys[r];
        a.best_keys[r] = NQ_KEY_EMPTY;
        if (key == NQ_KEY_EMPTY || out.n_scored == 0) {
            out.n_scored = 0;
            out.best_index = -1;
            out.best_score = 0;
            out.best_col = -1;
            out.best_value = -1;
        } else {
            const int slot = (int)(key & 0x7fffu);
            const int v = (int)((key >> 15) & 0x7fffu);
            const int colkey = (int)((key >> 30) & 0x7fffu);
            const int is_up = (int)((key >> 45) & 1u);
            const int c = is_up ? n - 1 - colkey : colkey;
            const int delta = (int)(key >> 46) - NQ_DELTA_BIAS;
            out.best_index = (long long)slot * n + v;
            out.best_score = a.cur_scores[r] + 2ll * delta;
            out.best_col = c;
            out.best_value = v;
        }
        results[r] = out;
        s_out = out;
    }
    if (next_rows == nullptr) return;
    __syncthreads();
    const ls_nq_result res = s_out;
    for (int j = threadIdx.x; j < n; j += 256) {
        int v = a.rows[(long long)r * n + j];
        if (res.best_index >= 0 && j == res.best_col) v = res.best_value;
        next_rows[(long long)r * n + j] = v;
    }
}

// get_col_scores + sum of one board per block with the occupancy tables in LDS (n <= NQ_CS_LDS_MAX_N):
// one launch instead of clear + build + clear + score.
constexpr int NQ_CS_LDS_MAX_N = 6400; // 5n ints <= 125 KB
__global__ __launch_bounds__(1024) void nq_col_scores_lds_kernel(const int *rows_all, int n, long long *col_scores,
                                                                 long long *totals)
{
    extern __shared__ int s_tab[]; // cntR[n] | cntA[2n] | cntB[2n]
    __shared__ long long s_sum[16];
    const int r = blockIdx.x;
    const int *rows = rows_all + (long long)r * n;
    int *tR = s_tab, *tA = s_tab + n, *tB = s_tab + 3 * n;
    for (int i = threadIdx.x; i < 5 * n; i += 1024) s_tab[i] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += 1024) {
        const int q = rows[j];
        if ((unsigned)q < (unsigned)n) {
            atomicAdd(&tR[q], 1);
            atomicAdd(&tA[q - j + n], 1);
            atomicAdd(&tB[q + j], 1);
        }
    }
    __syncthreads();
    long long local = 0;
    for (int j = threadIdx.x; j < n; j += 1024) {
        const int q = rows[j];
        long long sc = 0;
        if ((unsigned)q < (unsigned)n) sc = (long long)tR[q] + tA[q - j + n] + tB[q + j] - 3;
        if (col_scores) col_scores[(long long)r * n + j] = sc;
        local += sc;
    }
    const int lo = wave_sum_i32((int)local);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = lo;
    __syncthreads();
    if (threadIdx.x == 0 && totals) {
        long long t = 0;
        for (int i = 0; i < 16; i++) t += s_sum[i];
        totals[r] = t;
    }
}

// ======================================================= general tabu sets
// History configurations whose tabu set is more than {current} (all_solution_iteration_expiry == 0: the reference's set
// grows, local_search.rs:173-195).  The caller passes the unfiltered positions k whose boards are tabu besides the no-ops
// (sorted, each a real candidate); they are neither scored nor counted, so every later candidate's window rank drops by
// the number of excluded positions before it.  One lane per position, count tables in global memory; this path serves a
// configuration no reference example uses and is written for clarity, not speed.  One restart per launch.
__global__ __launch_bounds__(256) void nq_table_excl_kernel(NqArgs a, const long long *excl, int n_excl, unsigned long long *scored_count)
{
    __shared__ unsigned long long s_best[4];
    __shared__ int s_cnt[4];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int n = a.n;
    const int *__restrict__ rows = a.rows, *__restrict__ cols = a.cols;
    const int *__restrict__ tR = a.tabs, *__restrict__ tA = tR + n, *__restrict__ tB = tR + 3 * n;
    int ncols = a.n_cols[0];
    ncols = ncols < 0 ? 0 : (ncols > a.max_cols ? a.max_cols : ncols);
    const long long K = nq_k_limit((long long)ncols * n, n, a.window + n_excl); // positions a rank below the window can have
    const long long cur = a.cur_scores[0];
    unsigned long long best = NQ_KEY_EMPTY;
    int scored = 0;
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < K; k += (long long)gridDim.x * 256) {
        const int slot = (int)(k / n), v = (int)(k - (long long)slot * n);
        const int c = cols[slot];
        const int rc = (unsigned)c < (unsigned)n ? rows[c] : -1;
        if ((unsigned)rc >= (unsigned)n || v == rc) continue;
        int lo = 0, hi = n_excl; // excluded positions before k
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (excl[mid] < k) lo = mid + 1; else hi = mid;
        }
        if (lo < n_excl && excl[lo] == k) continue; // tabu: not scored, not counted
        const long long rank = (long long)slot * (n - 1) + v - (v > rc ? 1 : 0) - lo;
        if (rank >= a.window) continue;
        const int conf_old = tR[rc] + tA[rc - c + n] + tB[rc + c] - 3;
        const int delta = tR[v] + tA[v - c + n] + tB[v + c] - conf_old;
        best = u64min(best, nq_key(delta, nq_key_hi(v > rc ? 1 : 0, c, n, slot), v));
        if (a.scores_out) a.scores_out[k] = cur + 2ll * delta;
        scored++;
    }
    best = wave_min_u64(best);
    scored = wave_sum_i32(scored);
    if (lane == 0) { s_best[wib] = best; s_cnt[wib] = scored; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long b = u64min(u64min(s_best[0], s_best[1]), u64min(s_best[2], s_best[3]));
        if (b != NQ_KEY_EMPTY) atomicMin(&a.best_keys[0], b);
        const int t = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (t) atomicAdd(scored_count, (unsigned long long)t);
    }
}

__global__ void nq_finalize_excl_kernel(NqArgs a, ls_nq_result *results, unsigned long long *scored_count)
{
    const int n = a.n;
    ls_nq_result out;
    out.n_scored = (long long)*scored_count;
    *scored_count = 0;
    const unsigned long long key = a.best_keys[0];
    a.best_keys[0] = NQ_KEY_EMPTY;
    if (key == NQ_KEY_EMPTY || out.n_scored == 0) {
        out.n_scored = 0; out.best_index = -1; out.best_score = 0; out.best_col = -1; out.best_value = -1;
    } else {
        const int slot = (int)(key & 0x7fffu);
        const int v = (int)((key >> 15) & 0x7fffu);
        const int colkey = (int)((key >> 30) & 0x7fffu);
        const int is_up = (int)((key >> 45) & 1u);
        const int delta = (int)(key >> 46) - NQ_DELTA_BIAS;
        out.best_index = (long long)slot * n + v;
        out.best_score = a.cur_scores[0] + 2ll * delta;
        out.best_col = is_up ? n - 1 - colkey : colkey;
        out.best_value = v;
    }
    results[0] = out;
}

// ============================================================= fused LS step
// One LS iteration of one restart in ONE launch, one 1024-thread block per restart (the solver's
// per-iteration shape: window = 5n => ~5 columns, ~20k candidates, where launch latency, not
// throughput, is the cost).  The block reads rows / cols ONCE into LDS (so the inputs may live in
// pinned host memory), builds the three packed-u16 occupancy tables, scores the candidates that
// take(window) can reach, picks the (score, solution) minimum, applies the move to its LDS board and
// tables, and emits the winner's column scores (the next iteration's proposer input, lib.rs:182) --
// what the multi-launch path does with table_lds + finalize + col_scores_lds.


// PAIRS (even n >= 128): the tables carry the shifted row copy of nq_table_pairs_kernel and the candidates are scored by
// wavefront teams, two per lane and look-up (nq_pairs_unit); otherwise a flat one-candidate-per-lane walk.
template <bool PAIRS>
__global__ __launch_bounds__(1024) void nq_step_kernel(NqStepInline in, NqArgs a, ls_nq_result *results, int *next_rows,
                                                       long long *next_col_scores, long long *next_totals, NqStepExtra x)
{
    extern __shared__ __attribute__((aligned(16))) unsigned s_dyn[]; // tables (packed u16: R[n] | A[2n] | B[2n], or R | R1 | A[2n+2] | B[2n+2]) | rows[n] | slot records[n_slots]
    __shared__ unsigned long long s_best[16];
    __shared__ int s_sum[16], s_cnt[16];
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    const int r = blockIdx.x, n = a.n;
    const int words = PAIRS ? (((6 * n + 4) >> 1) + 3) & ~3 : (((5 * n + 1) >> 1) + 3) & ~3; // keeps the slot records 16-byte aligned
    unsigned *s_t16 = s_dyn;
    int *s_rows = reinterpret_cast<int *>(s_dyn + words);
    int4 *s_slot = reinterpret_cast<int4 *>(s_rows + ((n + 3) & ~3)); // {column, its row, conf(column, row), -}
    const bool from_resident = x.resident_rows != nullptr && r < 64 && !((x.fresh_mask >> r) & 1ull);
    const int *__restrict__ rows = from_resident ? x.resident_rows + (long long)r * n : a.rows + (long long)r * n;
    const int *__restrict__ cols = a.cols + (long long)r * a.max_cols;
    // every global (possibly PCIe) read of the kernel is issued here, back to back: one latency, not three
    int rv[NQ_STEP_MAX_N / 1024];
#pragma unroll
    for (int i = 0; i < NQ_STEP_MAX_N / 1024; i++) {
        const int j = tid + 1024 * i;
        rv[i] = j < n ? rows[j] : -1;
    }
    // The count tables of a resident board live next to it on the device (every step patches them with its move's
    // eight entries): they are loaded with the board instead of being rebuilt (zero + 4 LDS atomics per column + a barrier).
    const bool tabs_resident = from_resident && x.resident_tabs != nullptr && x.tab_words == words;
    constexpr int TAB_VEC = (((6 * NQ_STEP_MAX_N + 4) >> 1) + 3) / 4 / 1024 + 1; // uint4 loads per thread at most
    uint4 tv[TAB_VEC];
    if (tabs_resident) {
        const uint4 *gt = reinterpret_cast<const uint4 *>(x.resident_tabs + (long long)r * x.tab_words);
#pragma unroll
        for (int i = 0; i < TAB_VEC; i++) {
            const int w4 = tid + 1024 * i;
            tv[i] = w4 < (words >> 2) ? gt[w4] : make_uint4(0, 0, 0, 0);
        }
    }
    LS_STAMP(0);
    // the launcher guarantees n_slots <= 1024; `in.valid` (one restart, <= NQ_STEP_INLINE_COLS columns): from the kernel arguments
    const int cv = tid < a.max_cols ? (in.valid ? in.cols[tid < NQ_STEP_INLINE_COLS ? tid : 0] : cols[tid]) : -1;
    int ncols = in.valid ? in.n_cols : a.n_cols[r];
    const long long cur = tid == 0 ? (in.valid ? in.cur : a.cur_scores[r]) : 0;
    if (tabs_resident) {
        uint4 *st = reinterpret_cast<uint4 *>(s_t16);
#pragma unroll
        for (int i = 0; i < TAB_VEC; i++) {
            const int w4 = tid + 1024 * i;
            if (w4 < (words >> 2)) st[w4] = tv[i];
        }
    } else {
        for (int i = tid; i < words; i += 1024) s_t16[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < NQ_STEP_MAX_N / 1024; i++) {
        const int j = tid + 1024 * i;
        if (j < n) s_rows[j] = rv[i];
    }
    LS_STAMP(1);
    ncols = ncols < 0 ? 0 : (ncols > a.max_cols ? a.max_cols : ncols);
    const long long K = nq_k_limit((long long)ncols * n, n, a.window);
    const int n_slots = (int)((K + n - 1) / n);
    __syncthreads();
    const int oR1 = n, oA = PAIRS ? 2 * n : n, oB = PAIRS ? 4 * n + 2 : 3 * n;
    if (!tabs_resident) { // block-uniform
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
        __syncthreads();
    }
    LS_STAMP(2);
    int my_scored = 0; // this slot's share of n_scored (nq_slot_scored: skipped slots keep their window positions)
    if (tid < n_slots) {
        const int c = cv;
        const int rc = (unsigned)c < (unsigned)n ? s_rows[c] : -1;
        int conf_old = 0;
        if ((unsigned)rc < (unsigned)n)
            conf_old = tab16_get(s_t16, rc) + tab16_get(s_t16, oA + rc - c + n) + tab16_get(s_t16, oB + rc + c) - 3;
        s_slot[tid] = make_int4(c, (unsigned)rc < (unsigned)n ? rc : -1, conf_old, 0);
        my_scored = nq_slot_scored(tid, c, rc, n, a.window);
    }
    my_scored = wave_sum_i32(my_scored);
    if (lane == 0) s_cnt[wib] = my_scored;
    __syncthreads();
    LS_STAMP(3);
    unsigned long long best = NQ_KEY_EMPTY;
    if constexpr (PAIRS) {
        // whole slots by wavefront teams (sixteen per block), the slot the window cuts by every lane with per-range loops
        long long n_whole = a.window / (n - 1);
        if (n_whole > ncols) n_whole = ncols;
        const long long rem = n_whole < ncols ? a.window - n_whole * (n - 1) : 0;
        const int steps = (n + 127) >> 7;
        // steps per unit: the sixteen wavefronts' fair share, rounded UP to the walk's unroll factor of four (a remainder
        // step costs a full LDS round trip of its own)
        int per_part = n_whole > 0 ? (int)((n_whole * steps + 15) / 16) : steps;
        per_part = (per_part + 3) & ~3;
        per_part = per_part > steps ? steps : (per_part < 4 ? 4 : per_part);
        const int parts = (steps + per_part - 1) / per_part;
        const int team = __builtin_amdgcn_readfirstlane(wib);
        const int units = (int)n_whole * parts;
        for (int u = team; u < units; u += 16) {
            const int slot = u / parts, part = u - slot * parts;
            const int4 si = s_slot[slot];
            const int c = __builtin_amdgcn_readfirstlane(si.x), rc = __builtin_amdgcn_readfirstlane(si.y),
                      conf_old = __builtin_amdgcn_readfirstlane(si.z);
            if (rc < 0) continue;
            nq_pairs_unit<64, false>(s_t16, n, lane, c, rc, conf_old, slot, part * per_part, per_part, nullptr, 0, best);
        }
        if (rem > 0) {
            const int slot = (int)n_whole;
            const int4 si = s_slot[slot];
            const int c = si.x, rc = si.y, conf_old = si.z;
            if (rc >= 0) {
                const unsigned short *t16 = reinterpret_cast<const unsigned short *>(s_t16);
                const NqRanges rg = nq_ranges(0, n, rc, (long long)slot * (n - 1), a.window);
                unsigned m = NQ_KEY32_EMPTY;
                for (int v = rg.d0 + tid; v < rg.d1; v += 1024) {
                    const unsigned kk = (((unsigned)t16[v] + (unsigned)t16[oA + v - c + n] + (unsigned)t16[oB + v + c]) << 15) | (unsigned)v;
                    m = kk < m ? kk : m;
                }
                for (int v = rg.u0 + tid; v < rg.u1; v += 1024) {
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
        // Flat walk over the unfiltered positions k = slot * n + v that take(window) can reach; iterations are
        // independent, so the LDS look-ups of several candidates are in flight together.  k / n by an exact
        // multiply-high (k < 2^17, n < 2^14).
        const unsigned magic = (unsigned)(0x100000000ull / (unsigned)n) + 1u;
        const int kmax = (int)K;
#pragma unroll 4
        for (int k = tid; k < kmax; k += 1024) {
            const int slot = (int)__umulhi((unsigned)k, magic);
            const int v = k - slot * n;
            const int4 si = s_slot[slot];
            const int c = si.x, rc = si.y;
            const bool up = v > rc;
            const long long rank = (long long)slot * (n - 1) + v - (up ? 1 : 0); // tabu candidates do not consume window
            if (rc >= 0 && v != rc && rank < a.window) {
                const int delta = tab16_get(s_t16, v) + tab16_get(s_t16, oA + v - c + n) + tab16_get(s_t16, oB + v + c) - si.z;
                best = u64min(best, nq_key(delta, nq_key_hi(up ? 1 : 0, c, n, slot), v));
            }
        }
    }
    best = wave_min_u64(best);
    if (lane == 0) s_best[wib] = best;
    __syncthreads();
    LS_STAMP(4);
    // Every lane decodes the winner itself (sixteen broadcast LDS reads) instead of waiting for lane 0 to do it and to
    // patch the tables: the winner board's column scores follow from the OLD tables plus the move's own two squares --
    // a column j != c loses the old queen's attack and gains the new queen's, column c gets conf(c, v) (its old queen
    // never counts towards another square of its own column).
    unsigned long long key = s_best[0];
#pragma unroll
    for (int i = 1; i < 16; i++) key = u64min(key, s_best[i]);
    long long n_scored = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) n_scored += s_cnt[i];
    const bool has_move = key != NQ_KEY_EMPTY && n_scored != 0;
    int mc = -1, mv = -1, mq = -1; // the move: column, new row, old row (MoveIterator, lib.rs:226-228)
    if (has_move) {
        mv = (int)((key >> 15) & 0x7fffu);
        const int colkey = (int)((key >> 30) & 0x7fffu);
        mc = ((key >> 45) & 1u) ? n - 1 - colkey : colkey;
        mq = s_rows[mc];
    }
    // One-restart launches that publish a completion word and write nothing else to host memory (the resident route
    // with host-kept column scores): thread 0 writes the record with write-through (system-scope) stores, waits for
    // their acknowledgement and then stores the word -- no block barrier and no system-scope release (an L2 write-back).
    const bool lean_publish = x.done_flag != nullptr && gridDim.x == 1 && !next_col_scores && !next_rows && !next_totals &&
                              !(x.next_cs16 != nullptr && (r >= 64 || ((x.want_cs_mask >> r) & 1ull)));
    if (tid == 0) {
        ls_nq_result out;
        if (!has_move) {
            out.n_scored = 0; out.best_index = -1; out.best_score = 0; out.best_col = -1; out.best_value = -1;
        } else {
            out.n_scored = n_scored;
            out.best_index = (long long)(key & 0x7fffu) * n + mv;
            out.best_score = cur + 2ll * ((int)(key >> 46) - NQ_DELTA_BIAS);
            out.best_col = mc;
            out.best_value = mv;
        }
        if (lean_publish) {
            long long *dst = reinterpret_cast<long long *>(results + r);
            __hip_atomic_store(dst + 0, out.n_scored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(dst + 1, out.best_index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(dst + 2, out.best_score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(dst + 3, (long long)(((unsigned long long)(unsigned)out.best_value << 32) | (unsigned)out.best_col),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            results[r] = out;
        }
    }
    if (x.resident_tabs != nullptr && r < 64 && x.tab_words == words) { // block-uniform
        unsigned *gtab = x.resident_tabs + (long long)r * x.tab_words;
        if (!tabs_resident) { // a fresh board: its tables (before the move) go to the device copy first
            uint4 *gt = reinterpret_cast<uint4 *>(gtab);
            const uint4 *st = reinterpret_cast<const uint4 *>(s_t16);
            for (int w4 = tid; w4 < (words >> 2); w4 += 1024) gt[w4] = st[w4];
            __syncthreads(); // the stores are acknowledged before any lane patches the same words
        }
        if (has_move && tid < 8) { // the old queen leaves (tid 0..3), the new one arrives (4..7): row, shifted row copy, two diagonals
            const int q = tid < 4 ? mq : mv, what = tid & 3;
            int idx = -1;
            if (what == 0) idx = q;
            else if (what == 1) idx = (PAIRS && q > 0) ? oR1 + q - 1 : -1;
            else if (what == 2) idx = oA + (q - mc + n);
            else idx = oB + (q + mc);
            if (idx >= 0) {
                const unsigned one = 1u << (16 * (idx & 1));
                if (tid < 4) atomicSub(&gtab[idx >> 1], one); // the count includes the old queen: no borrow
                else atomicAdd(&gtab[idx >> 1], one);
            }
        }
    }
    LS_STAMP(5);
    int local = 0;
    // The full pass over the columns (winner board + its column scores) is skipped when the caller keeps the column
    // scores itself (resident route: the host updates its copy from the move, O(n) there) and the board is already
    // resident -- then only the moved queen is written.
    const bool want_cs16 = x.next_cs16 != nullptr && (r >= 64 || ((x.want_cs_mask >> r) & 1ull));
    const bool full_pass = next_col_scores || next_rows || next_totals || want_cs16 || (x.resident_rows && !from_resident);
    if (!full_pass && x.resident_rows && has_move && tid == 0) x.resident_rows[(long long)r * n + mc] = mv;
    for (int j = tid; j < (full_pass ? n : 0); j += 1024) {
        int q = s_rows[j];
        int sc = 0;
        if (j == mc) {
            q = mv;
            sc = tab16_get(s_t16, mv) + tab16_get(s_t16, oA + mv - mc + n) + tab16_get(s_t16, oB + mv + mc); // conf(c, v)
        } else if ((unsigned)q < (unsigned)n) {
            sc = tab16_get(s_t16, q) + tab16_get(s_t16, oA + q - j + n) + tab16_get(s_t16, oB + q + j) - 3;
            if (has_move) {
                const int dc = j > mc ? j - mc : mc - j;
                const int dn = q > mv ? q - mv : mv - q, dq = q > mq ? q - mq : mq - q;
                sc += ((dn == 0) | (dn == dc)) - ((dq == 0) | (dq == dc));
            }
        }
        if (next_col_scores) next_col_scores[(long long)r * n + j] = sc;
        if (want_cs16) x.next_cs16[(long long)r * n + j] = (unsigned short)sc;
        if (next_rows) next_rows[(long long)r * n + j] = q;
        if (x.resident_rows) x.resident_rows[(long long)r * n + j] = q;
        local += sc;
    }
    if (next_totals) { // block-uniform
        const int ws = wave_sum_i32(local);
        if (lane == 0) s_sum[wib] = ws;
        __syncthreads();
        if (tid == 0) {
            long long t = 0;
            for (int i = 0; i < 16; i++) t += s_sum[i];
            next_totals[r] = t;
        }
    }
    LS_STAMP(6);
    if (lean_publish) { // block-uniform
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the record (and the moved queen) have been acknowledged
            __hip_atomic_store(x.done_flag, x.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            LS_STAMP(7);
        }
    } else if (x.done_flag) { // block-uniform
        // Completion word: every block makes its outputs (host memory) visible at system scope, then counts itself;
        // the block that counts last publishes the sequence number.  __syncthreads orders the other wavefronts'
        // stores before thread 0's release fence.
        __syncthreads();
        if (tid == 0) {
            __threadfence_system();
            if (gridDim.x == 1) { // one restart: nobody else to wait for
                __hip_atomic_store(x.done_flag, x.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            } else if (__hip_atomic_fetch_add(x.block_counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
                __hip_atomic_store(x.block_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(x.done_flag, x.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            LS_STAMP(7);
        }
    }
}

// ================================================================ launchers
static long long nq_clamp_k(long long k_all, int n, int max_cols, long long window)
{
    if (n > 1 && window / (n - 1) < max_cols) return (window / (n - 1) + 1) * n; // reachable by take(window)
    return k_all;
}

static int nq_blocks_per_cu(const void *fn, size_t dyn_lds)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, dyn_lds) != hipSuccess || nb < 1) nb = 1;
    return nb > 8 ? 8 : nb;
}

template <class Board>
static int launch_scan_board(ls_ctx *ctx, const NqArgs &a, long long bpr_cap)
{
    // Exactly Board::WAVES 256-thread blocks per CU: the register budget alone may admit an odd
    // number of waves per SIMD, so a dynamic-LDS reservation (never touched) caps residency.
    constexpr int bpc = Board::WAVES;
    constexpr size_t lds_cap = bpc >= 8 ? 0 : (size_t)(160 * 1024 / bpc - 2048 - 4 * Board::TILES * NQ_TILE_DWORDS * 4);
    // per device: function attributes belong to the device's code object.  Contexts may run on different threads:
    // the flag is atomic and setting the (same) attribute twice is harmless.
    static std::atomic<bool> attr_done_dev[64];
    std::atomic<bool> &attr_done = attr_done_dev[ctx->device & 63];
    if (!attr_done.load(std::memory_order_acquire) && lds_cap > 0) {
        LS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nq_scan_regs_kernel<Board, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
        LS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nq_scan_regs_kernel<Board, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
        attr_done.store(true, std::memory_order_release);
    }
    long long total = (long long)ctx->n_cus * bpc;
    long long bpr = total / a.n_restarts;
    if (bpr < 1) bpr = 1;
    if (bpr > bpr_cap) bpr = bpr_cap;
    const dim3 grid((unsigned)(bpr * a.n_restarts));
    if (a.scores_out) hipLaunchKernelGGL((nq_scan_regs_kernel<Board, true>), grid, dim3(256), lds_cap, ctx->stream, a);
    else hipLaunchKernelGGL((nq_scan_regs_kernel<Board, false>), grid, dim3(256), lds_cap, ctx->stream, a);
    LS_HIP(ctx, hipGetLastError());
    return LS_OK;
}

static int launch_scan_pk16(ls_ctx *ctx, const NqArgs &a, long long cap)
{
    const int n = a.n;
    if (n <= 128) return launch_scan_board<BoardPk16<1>>(ctx, a, cap);
    if (n <= 256) return launch_scan_board<BoardPk16<2>>(ctx, a, cap);
    if (n <= 512) return launch_scan_board<BoardPk16<4>>(ctx, a, cap);
    if (n <= 1024) return launch_scan_board<BoardPk16<8>>(ctx, a, cap);
    if (n <= 2048) return launch_scan_board<BoardPk16<16>>(ctx, a, cap);
    if (n <= 4096) return launch_scan_board<BoardPk16<32>>(ctx, a, cap);
    return launch_scan_board<BoardPk16<64>>(ctx, a, cap);
}

static int launch_scan_f32(ls_ctx *ctx, const NqArgs &a, long long cap)
{
    const int n = a.n;
    if (n <= 64) return launch_scan_board<BoardF32<1>>(ctx, a, cap);
    if (n <= 128) return launch_scan_board<BoardF32<2>>(ctx, a, cap);
    if (n <= 256) return launch_scan_board<BoardF32<4>>(ctx, a, cap);
    if (n <= 512) return launch_scan_board<BoardF32<8>>(ctx, a, cap);
    if (n <= 1024) return launch_scan_board<BoardF32<16>>(ctx, a, cap);
    if (n <= 2048) return launch_scan_board<BoardF32<32>>(ctx, a, cap);
    return launch_scan_board<BoardF32<64>>(ctx, a, cap);
}

static int launch_scan(ls_ctx *ctx, const NqArgs &a, long long k_max)
{
    const int n = a.n;
    // every wavefront should own >= 16 candidates; never more blocks than fit the chip at once
    long long want_waves = (k_max + 15) / 16;
    if (want_waves < 1) want_waves = 1;
    const long long bpr_cap = (want_waves + 3) / 4;
    // 0 = default: FP32 registers (n <= 4096), packed u16 registers (n <= 8192), LDS beyond;
    // 1 = packed u16 also for n <= 4096 (measurement knob, LS_NQ_SCAN_VARIANT)
    const int variant = ctx->nq_scan_variant;
    if (n <= 4096 && variant != 1) return launch_scan_f32(ctx, a, bpr_cap);
    if (n <= 8192 && variant == 1) return launch_scan_pk16(ctx, a, bpr_cap);
    const int n_pad = (n + 511) & ~511;
    const size_t lds = (size_t)n_pad * 4 + 16 * 8;
    LS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nq_scan_lds_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    // 1024-thread blocks: two per CU fill the wave slots; LDS may allow only one
    int bpc = (int)((160 * 1024) / (lds + 256));
    bpc = bpc < 1 ? 1 : (bpc > 2 ? 2 : bpc);
    const long long want_blocks = (want_waves + 15) / 16;
    long long bpr = ((long long)ctx->n_cus * bpc) / a.n_restarts;
    if (bpr < 1) bpr = 1;
    if (bpr > want_blocks) bpr = want_blocks;
    if (bpr < 1) bpr = 1;
    hipLaunchKernelGGL(nq_scan_lds_kernel, dim3((unsigned)(bpr * a.n_restarts)), dim3(NQ_LDS_BLOCK), lds, ctx->stream, a);
    LS_HIP(ctx, hipGetLastError());
    return LS_OK;
}

static int build_tables(ls_ctx *ctx, int n_restarts, int n, const int *rows_d, int **tabs_out, int *stride_out)
{
    const int stride = 5 * n + 8;
    const size_t bytes = (size_t)n_restarts * stride * sizeof(int);
    LS_RESERVE(ctx, ctx->nq_tabs, bytes);
    {   // cleared by a kernel, not hipMemsetAsync: keeps a captured hipGraph all-kernel (memset nodes
        // did not re-execute reliably on replay with ROCm 7.x) and avoids the memset path's latency
        const long long cnt = (long long)n_restarts * stride;
        long long g = (cnt + 255) / 256;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, ctx->nq_tabs.as<int>(), 0, cnt);
        LS_HIP(ctx, hipGetLastError());
    }
    int gx = (n + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(nq_tables_build_kernel, dim3(gx, n_restarts), dim3(256), 0, ctx->stream, rows_d, n,
                       ctx->nq_tabs.as<int>(), stride);
    LS_HIP(ctx, hipGetLastError());
    *tabs_out = ctx->nq_tabs.as<int>();
    *stride_out = stride;
    return LS_OK;
}

// NT threads per block; teams of TL lanes; PAIRS: two candidates per lane and look-up
template <int NT, int TL, bool PAIRS>
static int launch_table_teams_t(ls_ctx *ctx, const NqArgs &a, long long window)
{
    constexpr int TEAMS = NT / TL;
    const int n = a.n;
    auto lds_for = [](int nn) {
        const size_t tab_words = PAIRS ? ((((size_t)(6 * nn + 4) >> 1) + 3) & ~(size_t)3) : ((((size_t)(5 * nn + 1) / 2) + 3) & ~(size_t)3);
        return (tab_words + 4 * (size_t)NQ_REC_CAP) * 4;
    };
    const size_t lds = lds_for(n);
    const void *k_plain, *k_scores;
    if constexpr (PAIRS) {
        k_plain = reinterpret_cast<const void *>(&nq_table_pairs_kernel<NT, TL, false>);
        k_scores = reinterpret_cast<const void *>(&nq_table_pairs_kernel<NT, TL, true>);
    } else {
        static_assert(PAIRS || TL == 256, "the one-candidate-per-lane kernel has 256-lane teams");
        k_plain = reinterpret_cast<const void *>(&nq_table_teams_kernel<NT / 256, false>);
        k_scores = reinterpret_cast<const void *>(&nq_table_teams_kernel<NT / 256, true>);
    }
    static std::atomic<bool> attr_done_dev[64];
    if (!attr_done_dev[ctx->device & 63].load(std::memory_order_acquire)) {
        const int max_lds = (int)lds_for(NQ_TABLE_LDS_MAX_N);
        LS_HIP(ctx, hipFuncSetAttribute(k_plain, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        LS_HIP(ctx, hipFuncSetAttribute(k_scores, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        attr_done_dev[ctx->device & 63].store(true, std::memory_order_release);
    }
    // residency: LDS (160 KB per CU) and wave slots (32 per CU)
    long long per_cu = (160 * 1024) / (long long)(lds + 512);
    const long long by_waves = 32 / (NT / 64);
    per_cu = per_cu > by_waves ? by_waves : per_cu;
    per_cu = per_cu < 1 ? 1 : per_cu;
    long long bpr = ((long long)ctx->n_cus * per_cu) / a.n_restarts;
    if (bpr < 1) bpr = 1;
    // work units: whole slots x parts; a unit should keep a team busy for >= 4 steps (of TL values, or TL pairs)
    long long n_whole = window / (n - 1);
    if (n_whole > a.max_cols) n_whole = a.max_cols;
    const int per_step = PAIRS ? 2 * TL : TL;
    const int steps = (n + per_step - 1) / per_step;
    long long parts = 1;
    if (n_whole > 0 && n_whole < bpr * TEAMS) {
        parts = (bpr * TEAMS) / n_whole;
        const long long max_parts = steps >= 4 ? steps / 4 : 1;
        parts = parts > max_parts ? max_parts : parts;
        parts = parts < 1 ? 1 : parts;
    }
    long long want_blocks = (n_whole * parts + TEAMS - 1) / TEAMS;
    if (want_blocks < 1) want_blocks = 1;
    if (bpr > want_blocks) bpr = want_blocks;
    while (bpr > 1 && n_whole * parts * bpr >= (1ll << 32)) bpr /= 2; // the kernels split units x chunk in 32 bits
    if (n_whole * parts >= (1ll << 32)) return set_err(ctx, LS_ERR_UNSUPPORTED, "max_cols = %d is too large for the count-table kernels", a.max_cols);
    const dim3 grid((unsigned)(bpr * a.n_restarts));
    int parts_i = (int)parts, whole_max = (int)n_whole; // n_whole = min(window / (n - 1), max_cols) from above
    void *args[] = {const_cast<NqArgs *>(&a), &parts_i, &whole_max}; // (the teams kernel takes the first two)
    LS_HIP(ctx, hipLaunchKernel(a.scores_out ? k_scores : k_plain, grid, dim3(NT), args, lds, ctx->stream));
    return LS_OK;
}

// four candidates per lane and look-up, 8-bit counts (nq_table_quads_kernel): n a multiple of 256, no score dump
static int launch_table_quads(ls_ctx *ctx, const NqArgs &a, long long window)
{
    constexpr int NT = 1024, TEAMS = NT / 64;
    const int n = a.n;
    auto lds_for = [](int nn) { // the larger of the two layouts: a block may fall back to the u16 pairs (nq_pairs_block)
        const size_t pairs_words = (((size_t)(6 * nn + 4) >> 1) + 3) & ~(size_t)3;
        return (pairs_words + 4 * (size_t)NQ_REC_CAP) * 4;
    };
    const size_t lds = lds_for(n);
    const void *k = reinterpret_cast<const void *>(&nq_table_quads_kernel<NT>);
    static std::atomic<bool> attr_done_dev[64];
    if (!attr_done_dev[ctx->device & 63].load(std::memory_order_acquire)) {
        LS_HIP(ctx, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_for(NQ_TABLE_LDS_MAX_N)));
        attr_done_dev[ctx->device & 63].store(true, std::memory_order_release);
    }
    long long per_cu = (160 * 1024) / (long long)(lds + 512);
    per_cu = per_cu > 2 ? 2 : (per_cu < 1 ? 1 : per_cu); // wave slots: 32 per CU
    long long bpr = ((long long)ctx->n_cus * per_cu) / a.n_restarts;
    if (bpr < 1) bpr = 1;
    long long n_whole = window / (n - 1);
    if (n_whole > a.max_cols) n_whole = a.max_cols;
    const int steps = n >> 8;
    long long parts = 1;
    if (n_whole > 0 && n_whole < bpr * TEAMS) {
        parts = (bpr * TEAMS) / n_whole;
        const long long max_parts = steps >= 4 ? steps / 4 : 1;
        parts = parts > max_parts ? max_parts : parts;
        parts = parts < 1 ? 1 : parts;
    }
    long long want_blocks = (n_whole * parts + TEAMS - 1) / TEAMS;
    if (want_blocks < 1) want_blocks = 1;
    if (bpr > want_blocks) bpr = want_blocks;
    while (bpr > 1 && n_whole * parts * bpr >= (1ll << 32)) bpr /= 2;
    if (n_whole * parts >= (1ll << 32)) return set_err(ctx, LS_ERR_UNSUPPORTED, "max_cols = %d is too large for the count-table kernels", a.max_cols);
    int parts_i = (int)parts, whole_max = (int)n_whole;
    void *args[] = {const_cast<NqArgs *>(&a), &parts_i, &whole_max};
    LS_HIP(ctx, hipLaunchKernel(k, dim3((unsigned)(bpr * a.n_restarts)), dim3(NT), args, lds, ctx->stream));
    return LS_OK;
}

// n a multiple of 256 and no score dump: four candidates per lane and look-up (nq_table_quads_kernel); other even n >= 512: two
// (nq_table_pairs_kernel), one wavefront per column either way; else one candidate per lane, 256-lane teams (nq_table_teams_kernel)
static int launch_table_teams(ls_ctx *ctx, const NqArgs &a, long long window)
{
    if ((a.n & 255) == 0 && a.n >= 512 && !a.scores_out) return launch_table_quads(ctx, a, window);
    if ((a.n & 1) == 0 && a.n >= 512) return launch_table_teams_t<1024, 64, true>(ctx, a, window);
    return launch_table_teams_t<1024, 256, false>(ctx, a, window);
}

int nq_eval_batch_dev(ls_ctx *ctx, int n_restarts, int n, const int *rows_d, const long long *cur_d,
                      const int *cols_d, const int *n_cols_d, int max_cols, long long window,
                      ls_nq_result *results_d, long long *scores_out_d, int *next_rows_d)
{
    // (Small launches are NOT routed to the one-block-per-restart step kernel here: back to back on a stream the
    // table kernel + finalize pair is faster -- 10.7 vs 12.6 us for 8 restarts x window 5n -- because its launch
    // latency pipelines; the step kernel pays off where a host round trip follows every launch, see ls_api.hip.)
    {   // Invariant: all keys are EMPTY between calls: a fresh buffer is filled once, in stream order, and the finalize
        // kernel restores what a call touched.  The fill must really run, so a buffer may not be born while the stream
        // is being captured into a hipGraph (callers warm up before capturing, like for every other workspace).
        // (hipMemset is no alternative: it is asynchronous to the host and ordered on the NULL stream, which the
        // context's non-blocking stream does not wait for.)
        const void *before = ctx->nq_keys.p;
        LS_RESERVE(ctx, ctx->nq_keys, (size_t)n_restarts * 8);
        const bool fresh = ctx->nq_keys.p != before;
        if (fresh) {
            hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(ctx->stream, &st) != hipSuccess) { (void)hipGetLastError(); st = hipStreamCaptureStatusNone; }
            if (st != hipStreamCaptureStatusNone) {
                ctx->nq_keys.cap = 0; // keep the buffer (it is retired on the next reserve) but never use it unfilled
                return set_err(ctx, LS_ERR_STATE, "first N-queens launch of this size during stream capture: run one launch "
                                                  "before capturing so that the workspaces exist");
            }
        }
        if (fresh || ctx->nq_keys_dirty) { // dirty: an earlier call failed between its eval and finalize launches
            hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)((ctx->nq_keys.cap / 8 + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->nq_keys.as<long long>(), -1ll, (long long)(ctx->nq_keys.cap / 8)); // NQ_KEY_EMPTY
            LS_HIP(ctx, hipGetLastError());
        }
        ctx->nq_keys_dirty = true; // until this call's finalize is queued
    }
    NqArgs a;
    a.rows = rows_d; a.cur_scores = cur_d; a.cols = cols_d; a.n_cols = n_cols_d;
    a.n = n; a.n_restarts = n_restarts; a.max_cols = max_cols; a.window = window;
    a.best_keys = ctx->nq_keys.as<unsigned long long>();
    a.scores_out = scores_out_d;
    a.tabs = nullptr; a.tab_stride = 0;
    const long long k_all = (long long)max_cols * n;
    long long k_max = k_all;
    if (n > 1 && window / (n - 1) < max_cols) k_max = (window / (n - 1) + 1) * n; // reachable by take(window)
    if (scores_out_d) {
        const long long cnt = (long long)n_restarts * k_all;
        if (cnt > 0) {
            long long g = (cnt + 255) / 256;
            if (g > 4096) g = 4096;
            hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, scores_out_d,
                               (long long)INT64_MIN, cnt);
            LS_HIP(ctx, hipGetLastError());
        }
    }
    if (max_cols > 0) {
        if (ctx->nq_algo != LS_NQ_ALGO_SCAN && n >= 256 && n <= NQ_TABLE_LDS_MAX_N) {
            int rc = launch_table_teams(ctx, a, window);
            if (rc != LS_OK) return rc;
        } else if (ctx->nq_algo != LS_NQ_ALGO_SCAN && n <= NQ_TABLE_LDS_MAX_N) {
            const size_t lds = (size_t)((5 * n + 1) / 2) * 4 + 16;
            static std::atomic<bool> attr_done_dev[64];
            if (!attr_done_dev[ctx->device & 63].load(std::memory_order_acquire)) {
                LS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nq_table_lds_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (5 * NQ_TABLE_LDS_MAX_N + 1) / 2 * 4 + 16));
                attr_done_dev[ctx->device & 63].store(true, std::memory_order_release);
            }
            long long per_cu = (160 * 1024) / (long long)(lds + 64);
            per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu); // LDS-latency bound: residency beats wave-count parity here (measured)
            long long want_blocks = (k_max + 4095) / 4096; // >= 16 candidates per lane: amortises the table build
            long long bpr = ((long long)ctx->n_cus * per_cu) / n_restarts;
            if (bpr < 1) bpr = 1;
            if (bpr > want_blocks) bpr = want_blocks;
            if (bpr < 1) bpr = 1;
            hipLaunchKernelGGL(nq_table_lds_kernel, dim3((unsigned)(bpr * n_restarts)), dim3(256), lds, ctx->stream, a);
            LS_HIP(ctx, hipGetLastError());
        } else if (ctx->nq_algo != LS_NQ_ALGO_SCAN) {
            int *tabs = nullptr; int stride = 0;
            int rc = build_tables(ctx, n_restarts, n, rows_d, &tabs, &stride);
            if (rc != LS_OK) return rc;
            a.tabs = tabs; a.tab_stride = stride;
            long long want_blocks = (k_max + 2047) / 2048; // >= 8 candidates per lane
            long long total = (long long)ctx->n_cus * 8;
            long long bpr = total / n_restarts;
            if (bpr < 1) bpr = 1;
            if (bpr > want_blocks) bpr = want_blocks;
            if (bpr < 1) bpr = 1;
            hipLaunchKernelGGL(nq_table_eval_kernel, dim3((unsigned)(bpr * n_restarts)), dim3(256), 0,
                               ctx->stream, a);
            LS_HIP(ctx, hipGetLastError());
        } else {
            int rc = launch_scan(ctx, a, k_max);
            if (rc != LS_OK) return rc;
        }
    }
    hipLaunchKernelGGL(nq_finalize_kernel, dim3(n_restarts), dim3(256), 0, ctx->stream, a, results_d, next_rows_d);
    LS_HIP(ctx, hipGetLastError());
    ctx->nq_keys_dirty = false;
    return LS_OK;
}

// One restart, every pointer in device memory; excl_d: sorted tabu positions (see nq_table_excl_kernel).
int nq_eval_excl_dev(ls_ctx *ctx, int n, const int *rows_d, const long long *cur_d, const int *cols_d, const int *n_cols_d,
                     int max_cols, long long window, const long long *excl_d, int n_excl, ls_nq_result *results_d,
                     long long *scores_out_d)
{
    {   // keys: the same invariant as nq_eval_batch_dev (all EMPTY between calls)
        const void *before = ctx->nq_keys.p;
        LS_RESERVE(ctx, ctx->nq_keys, 8);
        if (ctx->nq_keys.p != before || ctx->nq_keys_dirty) {
            hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)((ctx->nq_keys.cap / 8 + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->nq_keys.as<long long>(), -1ll, (long long)(ctx->nq_keys.cap / 8));
            LS_HIP(ctx, hipGetLastError());
        }
        ctx->nq_keys_dirty = true;
    }
    LS_RESERVE(ctx, ctx->nq_totals, 8); // the scored-candidate counter (the buffer is shared with ls_nqueens_col_scores: cleared per call)
    hipLaunchKernelGGL(fill_i64_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->nq_totals.as<long long>(), 0ll, 1ll);
    LS_HIP(ctx, hipGetLastError());
    NqArgs a;
    a.rows = rows_d; a.cur_scores = cur_d; a.cols = cols_d; a.n_cols = n_cols_d;
    a.n = n; a.n_restarts = 1; a.max_cols = max_cols; a.window = window;
    a.best_keys = ctx->nq_keys.as<unsigned long long>();
    a.scores_out = scores_out_d;
    const long long k_all = (long long)max_cols * n;
    if (scores_out_d && k_all > 0) {
        long long g = (k_all + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, scores_out_d, (long long)INT64_MIN, k_all);
        LS_HIP(ctx, hipGetLastError());
    }
    int *tabs = nullptr; int stride = 0;
    int rc = build_tables(ctx, 1, n, rows_d, &tabs, &stride);
    if (rc != LS_OK) return rc;
    a.tabs = tabs; a.tab_stride = stride;
    if (max_cols > 0 && n > 1) {
        long long k_max = nq_clamp_k(k_all, n, max_cols, window + n_excl);
        long long g = (k_max + 255) / 256;
        const long long cap = (long long)ctx->n_cus * 8;
        g = g > cap ? cap : (g < 1 ? 1 : g);
        hipLaunchKernelGGL(nq_table_excl_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, a, excl_d, n_excl,
                           ctx->nq_totals.as<unsigned long long>());
        LS_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(nq_finalize_excl_kernel, dim3(1), dim3(1), 0, ctx->stream, a, results_d, ctx->nq_totals.as<unsigned long long>());
    LS_HIP(ctx, hipGetLastError());
    ctx->nq_keys_dirty = false;
    return LS_OK;
}

int nq_col_scores_dev(ls_ctx *ctx, int n_restarts, int n, const int *rows_d, long long *col_scores_d,
                      long long *scores_d)
{
    if (n <= NQ_CS_LDS_MAX_N) {
        const size_t lds = (size_t)5 * n * sizeof(int);
        static std::atomic<bool> attr_done_dev[64];
        if (!attr_done_dev[ctx->device & 63].load(std::memory_order_acquire)) {
            LS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nq_col_scores_lds_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 5 * NQ_CS_LDS_MAX_N * (int)sizeof(int)));
            attr_done_dev[ctx->device & 63].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL(nq_col_scores_lds_kernel, dim3(n_restarts), dim3(1024), lds, ctx->stream, rows_d, n, col_scores_d,
                           scores_d);
        LS_HIP(ctx, hipGetLastError());
        return LS_OK;
    }
    int *tabs = nullptr; int stride = 0;
    int rc = build_tables(ctx, n_restarts, n, rows_d, &tabs, &stride);
    if (rc != LS_OK) return rc;
    if (scores_d) {
        hipLaunchKernelGGL(fill_i64_kernel, dim3((n_restarts + 255) / 256), dim3(256), 0, ctx->stream, scores_d, 0ll,
                           (long long)n_restarts);
        LS_HIP(ctx, hipGetLastError());
    }
    int gx = (n + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(nq_col_scores_kernel, dim3(gx, n_restarts), dim3(256), 0, ctx->stream, rows_d, n, tabs,
                       stride, col_scores_d, scores_d);
    LS_HIP(ctx, hipGetLastError());
    return LS_OK;
}

bool nq_step_eligible(const ls_ctx *ctx, int n, int max_cols, long long window)
{
    if (ctx->nq_algo == LS_NQ_ALGO_SCAN || n < 2 || n > NQ_STEP_MAX_N || max_cols < 1) return false;
    long long k_max = (long long)max_cols * n;
    if (window / (n - 1) < max_cols) k_max = (window / (n - 1) + 1) * n;
    return k_max <= NQ_STEP_MAX_K && (k_max + n - 1) / n <= NQ_STEP_MAX_SLOTS;
}

// Pointers may be device memory or pinned host memory (each input is read once, each output written once).
int nq_step_dev(ls_ctx *ctx, int n_restarts, int n, const int *rows, const long long *cur, const int *cols,
                const int *n_cols, int max_cols, long long window, ls_nq_result *results, int *next_rows,
                long long *next_col_scores, long long *next_totals, const NqStepExtra *extra, const NqStepInline *inl)
{
    NqArgs a;
    a.rows = rows; a.cur_scores = cur; a.cols = cols; a.n_cols = n_cols;
    a.n = n; a.n_restarts = n_restarts; a.max_cols = max_cols; a.window = window;
    a.best_keys = nullptr; a.scores_out = nullptr; a.tabs = nullptr; a.tab_stride = 0;
    const bool pairs = (n & 1) == 0 && n >= 128;
    static std::atomic<bool> attr_done_dev[64];
    if (!attr_done_dev[ctx->device & 63].load(std::memory_order_acquire)) {
        const int max_lds = (((6 * NQ_STEP_MAX_N + 4) / 2 + 4) + NQ_STEP_MAX_N + 4 * NQ_STEP_MAX_SLOTS) * 4;
        LS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nq_step_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        LS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&nq_step_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        attr_done_dev[ctx->device & 63].store(true, std::memory_order_release);
    }
    long long slots = (long long)max_cols;
    if (window / (n - 1) < max_cols) slots = window / (n - 1) + 1;
    const size_t tab_words = pairs ? ((((size_t)(6 * n + 4) >> 1) + 3) & ~(size_t)3) : ((((size_t)(5 * n + 1) / 2) + 3) & ~(size_t)3);
    const size_t lds = (tab_words + (((size_t)n + 3) & ~(size_t)3) + 4 * (size_t)slots) * 4;
    NqStepExtra x;
    if (extra) x = *extra;
    else { x.fresh_mask = ~0ull; x.resident_rows = nullptr; x.resident_tabs = nullptr; x.tab_words = 0; x.next_cs16 = nullptr; x.want_cs_mask = ~0ull; x.done_flag = nullptr; x.done_seq = 0; x.block_counter = nullptr; }
    static const NqStepInline no_inline = {};
    const NqStepInline &in = inl && n_restarts == 1 && max_cols <= NQ_STEP_INLINE_COLS ? *inl : no_inline;
    if (pairs) hipLaunchKernelGGL(nq_step_kernel<true>, dim3(n_restarts), dim3(1024), lds, ctx->stream, in, a, results, next_rows, next_col_scores, next_totals, x);
    else hipLaunchKernelGGL(nq_step_kernel<false>, dim3(n_restarts), dim3(1024), lds, ctx->stream, in, a, results, next_rows, next_col_scores, next_totals, x);
    LS_HIP(ctx, hipGetLastError());
    return LS_OK;
}

} // namespace lsamd

#ifdef LS_NQ_STAMPS
extern "C" int ls_debug_nq_stamps(unsigned long long *out, int count)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lsamd::g_nq_stamps), (size_t)count * 8);
}
#endif
