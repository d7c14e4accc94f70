// host/rng.hpp — the random-number stack the reference's drivers sit on, restated for the
// host side of the boundary (move proposal, perturbation, acceptance stay on the host by
// design).  None of these crates is in /root/reference (Cargo.lock pins rand 0.8.5,
// rand_chacha 0.3.1, rand_core 0.6.3, blake2 0.10.4); the algorithms below follow their
// published sources as recorded in SURVEY.md Appendix A.  Stream equality with the Rust
// binaries is UNVERIFIED here (no Rust toolchain); Blake2b and the ChaCha20 block function
// are checked against RFC 7693 / RFC 7539 vectors in the tests.
//
// Call sites in the reference: examples/nqueens/src/main.rs:28-39,66 (seed hashing, two
// identically seeded ChaCha20Rng), nqueens lib.rs:158,198,202-203,297,303-311,
// employee-scheduling lib.rs:408,461,467-468,473,594,601-607,
// iterated_local_search.rs:61,70, local_search.rs:226.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace lshost {

// ------------------------------------------------------------------ Blake2b-256
// hash_str (nqueens main.rs:28-33): Blake2b with a 32-byte digest, no key.  RFC 7693.
inline void blake2b_256(const void *data, size_t len, uint8_t out[32])
{
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                   0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                   0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    static const uint8_t SIGMA[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    auto rotr = [](uint64_t x, int n) { return (x >> n) | (x << (64 - n)); };
    uint64_t h[8];
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010000ULL ^ 32ULL; // digest length 32, no key, fanout = depth = 1
    const uint8_t *p = static_cast<const uint8_t *>(data);
    uint64_t t = 0;
    auto compress = [&](const uint8_t block[128], bool last) {
        uint64_t m[16], v[16];
        for (int i = 0; i < 16; i++) {
            uint64_t w = 0;
            for (int b = 7; b >= 0; b--) w = (w << 8) | block[8 * i + b];
            m[i] = w;
        }
        for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
        v[12] ^= t; // low word of the 128-bit counter (inputs here are far below 2^64 bytes)
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; r++) {
            const uint8_t *s = SIGMA[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    };
    uint8_t block[128];
    size_t rem = len;
    while (rem > 128) {
        t += 128;
        compress(p, false);
        p += 128; rem -= 128;
    }
    std::memset(block, 0, sizeof block);
    std::memcpy(block, p, rem);
    t += rem;
    compress(block, true);
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(h[i] >> (8 * b));
}

// -------------------------------------------------------------------- ChaCha20
// One 64-byte block of the original (djb) layout rand_chacha uses: 64-bit block counter in
// words 12-13, 64-bit stream id in words 14-15.
inline void chacha20_block(const uint32_t key[8], uint32_t w12, uint32_t w13, uint32_t w14, uint32_t w15,
                           uint32_t out[16])
{
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4], key[5], key[6], key[7], w12, w13, w14, w15};
    uint32_t x[16];
    std::memcpy(x, s, sizeof x);
    auto rotl = [](uint32_t v, int n) { return (v << n) | (v >> (32 - n)); };
    auto QR = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    };
    for (int r = 0; r < 10; r++) {
        QR(0, 4, 8, 12); QR(1, 5, 9, 13); QR(2, 6, 10, 14); QR(3, 7, 11, 15);
        QR(0, 5, 10, 15); QR(1, 6, 11, 12); QR(2, 7, 8, 13); QR(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

// rand_chacha::ChaCha20Rng over rand_core::block::BlockRng.  BlockRng hands out the words of consecutive ChaCha20
// blocks in order -- next_u32 one word, next_u64 two consecutive words (low first), also across its 64-word buffer
// boundary (rand_core block.rs: the index == len - 1 case) -- so the generator is the continuous word stream of the
// block function, and this class keeps a position in that stream plus a pool of the blocks around it.
class ChaCha20Rng {
public:
    ChaCha20Rng() { std::memset(key_, 0, sizeof key_); }
    // SeedableRng::from_seed([u8; 32]): key words little-endian, counter = stream = 0
    explicit ChaCha20Rng(const uint8_t seed[32])
    {
        for (int i = 0; i < 8; i++)
            key_[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) |
                      ((uint32_t)seed[4 * i + 3] << 24);
    }
    static ChaCha20Rng from_seed_string(const std::string &s)
    {
        uint8_t d[32];
        blake2b_256(s.data(), s.size(), d);
        return ChaCha20Rng(d);
    }
    uint32_t next_u32() { return word(pos_++); }
    uint64_t next_u64()
    {
        const uint64_t lo = word(pos_), hi = word(pos_ + 1);
        pos_ += 2;
        return (hi << 32) | lo;
    }
    // Standard f64: 53 random bits scaled into [0, 1)
    double gen_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
    // n consecutive gen_f64() draws: eight per step wherever sixteen words of the position's group lie ahead
    void fill_f64(double *out, uint32_t n)
    {
        uint32_t i = 0;
#if defined(__x86_64__)
        if (have_avx512dq())
            while (i + 8 <= n) {
                const uint64_t g = pos_ >> 8;
                const unsigned w = (unsigned)(pos_ & 255);
                if (w + 16 > 256) { out[i++] = gen_f64(); continue; } // fewer than eight draws left in this group: one at a time up to the seam
                if (g >= hi_group_) { generate_group(g); hi_group_ = g + 1; }
                unsigned take = (256 - w) / 16;                        // steps of eight draws left in this group
                if (take > (n - i) / 8) take = (n - i) / 8;
                f64_from_words(pool_[(size_t)(g % GROUPS)] + w, out + i, take);
                i += 8 * take;
                pos_ += 16 * (uint64_t)take;
            }
#endif
        for (; i < n; i++) out[i] = gen_f64();
    }
    uint64_t words_drawn() const { return pos_; }
    // Compute one more group of the key stream ahead of the position (idle time of the caller, e.g. while a launch is in
    // flight): the stream is a pure function of (key, block number), so this changes nothing but when the work is done.
    // Returns false once GROUPS groups lie ahead.
    bool prefetch()
    {
        const uint64_t cur = pos_ >> 8;
        if (hi_group_ < cur) hi_group_ = cur; // nothing valid ahead
        if (hi_group_ - cur >= (uint64_t)GROUPS) return false;
        generate_group(hi_group_);
        hi_group_++;
        return true;
    }

private:
    // word k of the stream = word k mod 16 of block k / 16.  The pool is a ring of GROUPS groups of sixteen blocks
    // (group g = blocks 16 g .. 16 g + 15, in slot g mod GROUPS; valid: the position's group up to hi_group_), each in STREAM
    // order (slot[16 b + w] = word w of block 16 g + b: the vectorised block function's word-major lanes are transposed when a
    // group is generated -- mostly in the caller's idle time --, so that runs of draws are contiguous loads: fill_f64).
    static constexpr int GROUPS = 16;
    uint32_t word(uint64_t k)
    {
        const uint64_t g = k >> 8;
        if (g >= hi_group_) {
            generate_group(g);
            hi_group_ = g + 1;
        }
        return pool_[(size_t)(g % GROUPS)][(size_t)(k & 255)];
    }
    // One 32-bit lane per block (compiler vector extension): four blocks at a time, eight with AVX2, sixteen with
    // AVX-512 (single-instruction rotates).  The word sequence is that of chacha20_block called block by block
    // (tests/test_host_rng.py compares the three paths and pins the stream to OpenSSL's ChaCha20).
#define LS_CHACHA_LANES(L)                                                                           \
    typedef uint32_t vL __attribute__((vector_size(4 * L)));                                        \
    vL s[16], x[16];                                                                                 \
    const uint32_t sigma[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};                  \
    for (int i = 0; i < 4; i++) s[i] = vL{} + sigma[i];                                              \
    for (int i = 0; i < 8; i++) s[4 + i] = vL{} + key_[i];                                           \
    s[12] = vL{}; s[13] = vL{}; s[14] = vL{}; s[15] = vL{};                                          \
    for (int b = 0; b < L; b++) {                                                                    \
        const uint64_t c = first + (uint64_t)b;                                                      \
        s[12][b] = (uint32_t)c;                                                                      \
        s[13][b] = (uint32_t)(c >> 32);                                                              \
    }                                                                                                \
    for (int i = 0; i < 16; i++) x[i] = s[i];                                                        \
    for (int r = 0; r < 10; r++) {                                                                   \
        LS_QR(0, 4, 8, 12) LS_QR(1, 5, 9, 13) LS_QR(2, 6, 10, 14) LS_QR(3, 7, 11, 15)                \
        LS_QR(0, 5, 10, 15) LS_QR(1, 6, 11, 12) LS_QR(2, 7, 8, 13) LS_QR(3, 4, 9, 14)                \
    }                                                                                                \
    for (int i = 0; i < 16; i++) x[i] += s[i];                                                       \
    LS_CHACHA_STORE(L)
#define LS_QR(a, b, c, d)                                                            \
    x[a] += x[b]; x[d] = ((x[d] ^ x[a]) << 16) | ((x[d] ^ x[a]) >> 16);              \
    x[c] += x[d]; x[b] = ((x[b] ^ x[c]) << 12) | ((x[b] ^ x[c]) >> 20);              \
    x[a] += x[b]; x[d] = ((x[d] ^ x[a]) << 8) | ((x[d] ^ x[a]) >> 24);               \
    x[c] += x[d]; x[b] = ((x[b] ^ x[c]) << 7) | ((x[b] ^ x[c]) >> 25);
    // blocks first .. first + L - 1 in stream order (dst = the first block's sixteen words): x[i] holds word i of the L blocks
#define LS_CHACHA_STORE(L)                                                                           \
    for (int i = 0; i < 16; i++)                                                                     \
        for (int b = 0; b < L; b++) dst[16 * b + i] = x[i][b];
    void blocks4(uint64_t first, uint32_t *dst) { LS_CHACHA_LANES(4) }
#if defined(__x86_64__)
    __attribute__((target("avx2"))) void blocks8(uint64_t first, uint32_t *dst) { LS_CHACHA_LANES(8) }
#undef LS_CHACHA_STORE
    // sixteen blocks: a 16 x 16 transposition of 32-bit words in registers (unpack 32, unpack 64, two 128-bit lane shuffles)
#define LS_CHACHA_STORE(L)                                                                           \
    __m512i r[16], t[16];                                                                            \
    for (int i = 0; i < 16; i++) std::memcpy(&r[i], &x[i], 64);                                      \
    for (int i = 0; i < 16; i += 2) { t[i] = _mm512_unpacklo_epi32(r[i], r[i + 1]); t[i + 1] = _mm512_unpackhi_epi32(r[i], r[i + 1]); } \
    for (int i = 0; i < 16; i += 4) {                                                                \
        r[i] = _mm512_unpacklo_epi64(t[i], t[i + 2]); r[i + 1] = _mm512_unpackhi_epi64(t[i], t[i + 2]); \
        r[i + 2] = _mm512_unpacklo_epi64(t[i + 1], t[i + 3]); r[i + 3] = _mm512_unpackhi_epi64(t[i + 1], t[i + 3]); \
    }                                                                                                \
    for (int h = 0; h < 16; h += 8)                                                                  \
        for (int i = 0; i < 4; i++) {                                                                \
            t[h + i] = _mm512_shuffle_i32x4(r[h + i], r[h + i + 4], 0x88);                           \
            t[h + i + 4] = _mm512_shuffle_i32x4(r[h + i], r[h + i + 4], 0xdd);                       \
        }                                                                                            \
    for (int i = 0; i < 8; i++) {                                                                    \
        r[i] = _mm512_shuffle_i32x4(t[i], t[i + 8], 0x88);                                           \
        r[i + 8] = _mm512_shuffle_i32x4(t[i], t[i + 8], 0xdd);                                       \
    }                                                                                                \
    for (int b = 0; b < 16; b++) _mm512_store_si512(dst + 16 * b, r[b]);
    __attribute__((target("avx512f"))) void blocks16(uint64_t first, uint32_t *dst) { LS_CHACHA_LANES(16) }
#undef LS_CHACHA_STORE
    static bool have_avx2() // LS_HOST_NO_AVX2=1 forces the four-lane path (tests run all three)
    {
        static const bool v = __builtin_cpu_supports("avx2") && std::getenv("LS_HOST_NO_AVX2") == nullptr;
        return v;
    }
    static bool have_avx512() // LS_HOST_NO_AVX512=1 forces the eight-lane path
    {
        static const bool v = have_avx2() && __builtin_cpu_supports("avx512f") && std::getenv("LS_HOST_NO_AVX512") == nullptr;
        return v;
    }
    static bool have_avx512dq()
    {
        static const bool v = have_avx512() && __builtin_cpu_supports("avx512dq");
        return v;
    }
    // (double)(u64 >> 11) * 2^-53 for `steps` x 8 consecutive u64 of the stream (word pairs, low word first): gen_f64's value
    __attribute__((target("avx512f,avx512dq"))) static void f64_from_words(const uint32_t *w, double *out, unsigned steps)
    {
        const __m512d scale = _mm512_set1_pd(1.0 / 9007199254740992.0);
        for (unsigned s = 0; s < steps; s++) {
            const __m512i v = _mm512_loadu_si512(w + 16 * s);
            _mm512_storeu_pd(out + 8 * s, _mm512_mul_pd(_mm512_cvtepu64_pd(_mm512_srli_epi64(v, 11)), scale));
        }
    }
#endif
#undef LS_QR
#undef LS_CHACHA_LANES
    void generate_group(uint64_t g)
    {
        uint32_t *slot = pool_[(size_t)(g % GROUPS)];
        const uint64_t first = g << 4;
#if defined(__x86_64__)
        if (have_avx512()) { blocks16(first, slot); return; }
        if (have_avx2()) { blocks8(first, slot); blocks8(first + 8, slot + 8 * 16); return; }
#endif
        for (int q = 0; q < 4; q++) blocks4(first + 4 * (uint64_t)q, slot + 4 * q * 16);
    }
    uint32_t key_[8];
    uint64_t pos_ = 0;      // next word of the stream
    uint64_t hi_group_ = 0; // groups below this one (and not behind the position by GROUPS or more) are in the pool
    alignas(64) uint32_t pool_[GROUPS][256];
};

// ----------------------------------------------------------- rand 0.8.5 sampling
// UniformInt::sample_single_inclusive, 32-bit and 64-bit draws (widening multiply + zone).
template <class R> inline uint32_t gen_range_u32_incl(R &rng, uint32_t low, uint32_t high)
{
    const uint32_t range = high - low + 1;
    if (range == 0) return rng.next_u32();
    const uint32_t zone = (range << __builtin_clz(range)) - 1;
    for (;;) {
        const uint64_t m = (uint64_t)rng.next_u32() * range;
        if ((uint32_t)m <= zone) return low + (uint32_t)(m >> 32);
    }
}
template <class R> inline uint64_t gen_range_u64_incl(R &rng, uint64_t low, uint64_t high)
{
    const uint64_t range = high - low + 1;
    if (range == 0) return rng.next_u64();
    const uint64_t zone = (range << __builtin_clzll(range)) - 1;
    for (;;) {
        const unsigned __int128 m = (unsigned __int128)rng.next_u64() * range;
        if ((uint64_t)m <= zone) return low + (uint64_t)(m >> 64);
    }
}
// rng.gen_range(low..high) on u64 / usize / i64 (64-bit targets)
template <class R> inline uint64_t gen_range_u64(R &rng, uint64_t low, uint64_t high) { return gen_range_u64_incl(rng, low, high - 1); }

// Uniform::new(0, high).sample for the WeightedIndex draw: a DIFFERENT zone than sample_single
template <class R> inline uint64_t uniform_u64_below(R &rng, uint64_t high)
{
    const uint64_t range = high; // new_inclusive(0, high - 1)
    const uint64_t ints_to_reject = (UINT64_MAX - range + 1) % range;
    const uint64_t zone = UINT64_MAX - ints_to_reject;
    for (;;) {
        const unsigned __int128 m = (unsigned __int128)rng.next_u64() * range;
        if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
    }
}
template <class R> inline uint32_t uniform_u32_below(R &rng, uint32_t high)
{
    const uint32_t range = high;
    const uint32_t ints_to_reject = (UINT32_MAX - range + 1) % range;
    const uint32_t zone = UINT32_MAX - ints_to_reject;
    for (;;) {
        const uint64_t m = (uint64_t)rng.next_u32() * range;
        if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
    }
}

// seq::gen_index: u32 draws whenever the bound fits
template <class R> inline size_t gen_index(R &rng, size_t ubound)
{
    if (ubound <= (size_t)UINT32_MAX) return gen_range_u32_incl(rng, 0, (uint32_t)ubound - 1);
    return (size_t)gen_range_u64_incl(rng, 0, (uint64_t)ubound - 1);
}

// SliceRandom::shuffle
template <class R, class T> inline void shuffle(R &rng, std::vector<T> &v)
{
    for (size_t i = v.size(); i-- > 1;) std::swap(v[i], v[gen_index(rng, i + 1)]);
}

// SliceRandom::choose -> index
template <class R> inline size_t choose_index(R &rng, size_t len)
{
    if (len == 0) throw std::runtime_error("choose on an empty slice (the reference unwraps None)");
    return gen_index(rng, len);
}

// SliceRandom::choose_weighted with integer weights: WeightedIndex::new + sample.
// `wide` selects the weight type: u64 (strategy tables) or i32 (acceptance tuple literals).
template <class R> inline size_t choose_weighted_index(R &rng, const std::vector<uint64_t> &weights, bool wide)
{
    if (weights.empty()) throw std::runtime_error("choose_weighted: no items");
    uint64_t total = 0;
    std::vector<uint64_t> cumulative;
    for (size_t i = 0; i < weights.size(); i++) {
        if (i + 1 < weights.size()) cumulative.push_back(total + weights[i]);
        total += weights[i];
    }
    if (total == 0) throw std::runtime_error("choose_weighted: all weights zero");
    const uint64_t chosen = wide ? uniform_u64_below(rng, total) : uniform_u32_below(rng, (uint32_t)total);
    // partition_point(|w| w <= chosen)
    size_t idx = 0;
    while (idx < cumulative.size() && cumulative[idx] <= chosen) idx++;
    return idx;
}

// which of index::sample's three algorithms rand 0.8.5 picks (the selection at the end of index_sample, restated for callers
// that inline the Floyd path for small amounts)
inline bool index_sample_uses_floyd(uint32_t length, uint32_t amount)
{
    if (amount > length || amount >= 163) return false;
    const int j = length < 500000 ? 0 : 1;
    static const float C[2][2] = {{1.6f, 8.0f / 45.0f}, {10.0f, 70.0f / 9.0f}};
    const float amount_fp = (float)amount;
    return !(amount > 11 && (float)length < (C[1][j] + C[0][j] * amount_fp) * amount_fp);
}

// seq::index::sample(rng, length, amount) — indices in the order the reference yields them
template <class R> inline std::vector<uint32_t> index_sample(R &rng, uint32_t length, uint32_t amount)
{
    if (amount > length) throw std::runtime_error("`amount` of samples must be less than or equal to `length`");
    auto floyd = [&]() {
        const bool floyd_shuffle = amount < 50;
        std::vector<uint32_t> idx;
        idx.reserve(amount);
        // membership by bitmap (the reference scans the vector: the same answer; the N-queens proposer draws ~100 of ~200 here
        // every LS iteration, ~5000 compares per call otherwise); the position is only searched for when it is needed
        uint64_t small_map[64] = {0};
        std::vector<uint64_t> big_map;
        uint64_t *map = small_map;
        if (length > 64 * 64) { big_map.assign(((size_t)length + 63) / 64, 0); map = big_map.data(); }
        auto has = [&](uint32_t v) { return (map[v >> 6] >> (v & 63)) & 1u; };
        auto put = [&](uint32_t v) { map[v >> 6] |= 1ull << (v & 63); };
        for (uint32_t j = length - amount; j < length; j++) {
            const uint32_t t = gen_range_u32_incl(rng, 0, j);
            if (has(t)) {
                if (floyd_shuffle) idx.insert(std::find(idx.begin(), idx.end(), t), j);
                else idx.push_back(j);
                put(j);
                continue;
            }
            idx.push_back(t);
            put(t);
        }
        if (!floyd_shuffle)
            for (uint32_t i = amount; i-- > 1;) std::swap(idx[i], idx[gen_range_u32_incl(rng, 0, i)]);
        return idx;
    };
    auto inplace = [&]() {
        std::vector<uint32_t> idx(length);
        for (uint32_t i = 0; i < length; i++) idx[i] = i;
        for (uint32_t i = 0; i < amount; i++) std::swap(idx[i], idx[gen_range_u32_incl(rng, i, length - 1)]);
        idx.resize(amount);
        return idx;
    };
    auto rejection = [&]() {
        std::unordered_set<uint32_t> cache;
        std::vector<uint32_t> idx;
        idx.reserve(amount);
        for (uint32_t i = 0; i < amount; i++) {
            uint32_t pos = uniform_u32_below(rng, length);
            while (!cache.insert(pos).second) pos = uniform_u32_below(rng, length);
            idx.push_back(pos);
        }
        return idx;
    };
    const int j = length < 500000 ? 0 : 1;
    if (amount < 163) {
        static const float C[2][2] = {{1.6f, 8.0f / 45.0f}, {10.0f, 70.0f / 9.0f}};
        const float amount_fp = (float)amount;
        const float m4 = C[0][j] * amount_fp;
        if (amount > 11 && (float)length < (C[1][j] + m4) * amount_fp) return inplace();
        return floyd();
    }
    static const float C2[2] = {270.0f, 330.0f / 9.0f};
    if ((float)length < C2[j] * (float)amount) return inplace();
    return rejection();
}

// seq::index::sample_weighted -> Efraimidis-Spirakis with a binary max-heap (non-nightly path):
// key_i = u_i^(1/w_i), the `amount` largest keys in descending order.
// The reference pushes every key into a BinaryHeap and pops `amount` times.  index_sample_weighted_reference does the
// same by selection + sort (the round-1 implementation, kept as the yardstick of the unit test);
// index_sample_weighted returns the identical sequence by a 256-bucket histogram over the keys (they lie in [0, 1]):
// one counting pass finds the bucket the amount-th largest key falls into, everything above it is taken, the remainder
// comes from that bucket, and the taken elements are ordered bucket by bucket.  Ties between equal keys (probability
// ~0 for 53-bit random keys) are broken by index in both.  The N-queens proposer spends most of its host time here
// (about 1100 conflicting columns, 204 picks per LS iteration at n = 4096).
struct WeightedKey { double key; uint32_t index; };
inline bool weighted_key_before(const WeightedKey &a, const WeightedKey &b) { return a.key > b.key || (a.key == b.key && a.index < b.index); }

template <class R> inline void weighted_keys(R &rng, const std::vector<double> &weights, std::vector<WeightedKey> &keyed)
{
    const uint32_t length = (uint32_t)weights.size();
    keyed.resize(length);
    for (uint32_t i = 0; i < length; i++) {
        if (!(weights[i] >= 0.0)) throw std::runtime_error("invalid weight");
        keyed[i] = WeightedKey{std::pow(rng.gen_f64(), 1.0 / weights[i]), i};
    }
}

template <class R> inline std::vector<uint32_t> index_sample_weighted_reference(R &rng, const std::vector<double> &weights, uint32_t amount)
{
    const uint32_t length = (uint32_t)weights.size();
    if (amount == 0) return {};
    if (amount > length) throw std::runtime_error("`amount` of samples must be less than or equal to `length`");
    std::vector<WeightedKey> keyed;
    weighted_keys(rng, weights, keyed);
    if (amount < length) std::nth_element(keyed.begin(), keyed.begin() + amount, keyed.end(), weighted_key_before);
    std::sort(keyed.begin(), keyed.begin() + amount, weighted_key_before);
    std::vector<uint32_t> out;
    out.reserve(amount);
    for (uint32_t i = 0; i < amount; i++) out.push_back(keyed[i].index);
    return out;
}

// The `amount` best of `keyed` by (key descending, index ascending), in that order: a bucket histogram over the keys.
// `key_floor` is a hint -- the buckets divide [key_floor, 1], keys below it share bucket 0 -- so that a pre-filtered
// input (all keys in a narrow band under 1) still spreads over the buckets; short inputs use 64 buckets, not 256.
// `key_ceil`: the keys lie in [.., key_ceil] (1 for u^(1/w); 0 for the surrogate keys y log2(u)).  `guard` > 0: also report in
// *ambiguous whether any two keys whose order matters -- neighbours in the result, or the last taken and the best left out --
// are closer than `guard`.
// weighted_top's bucket of a key, as a function the vectorised key computation can evaluate as well: B = weighted_top_buckets(n)
inline int weighted_top_buckets(uint32_t length) { return length <= 128 ? 64 : 256; } // (256 from 129 keys on: ~280 survivors crowd 64 buckets unevenly)

inline void weighted_top(std::vector<WeightedKey> &keyed, uint32_t amount, std::vector<uint32_t> &out, double key_floor = 0.0,
                         double key_ceil = 1.0, double guard = 0.0, bool *ambiguous = nullptr, const uint8_t *pre_bucket = nullptr)
{
    static thread_local std::vector<WeightedKey> taken;
    static thread_local std::vector<uint8_t> bucket;
    const uint32_t length = (uint32_t)keyed.size();
    constexpr int BMAX = 256;
    const int B = weighted_top_buckets(length);
    if (!(key_floor < key_ceil)) key_floor = key_ceil - 1.0;
    const double scale = (double)B / (key_ceil - key_floor);
    uint32_t hist[BMAX];
    for (int b = 0; b < B; b++) hist[b] = 0;
    bucket.resize(length);
    if (pre_bucket) { // computed with the keys (weighted_log2_keys_avx512): any non-decreasing map of the key into [0, B) serves
        for (uint32_t i = 0; i < length; i++) {
            const uint8_t b = pre_bucket[i] < B ? pre_bucket[i] : (uint8_t)(B - 1);
            bucket[i] = b;
            hist[b]++;
        }
    } else
        for (uint32_t i = 0; i < length; i++) {
            const double k = keyed[i].key; // <= key_ceil; NaN cannot occur (w >= 0, 0 <= u < 1)
            int b = k <= key_floor ? 0 : (int)((k - key_floor) * scale);
            b = b < 0 ? 0 : (b > B - 1 ? B - 1 : b);
            bucket[i] = (uint8_t)b;
            hist[b]++;
        }
    int t = B - 1; // threshold bucket: the buckets above it hold fewer than `amount` keys, with it at least `amount`
    uint32_t above = 0;
    while (above + hist[t] < amount) { above += hist[t]; t--; }
    uint32_t start[BMAX]; // where every bucket's elements start in `taken` (descending buckets)
    {
        uint32_t off = 0;
        for (int b = B - 1; b > t; b--) { start[b] = off; off += hist[b]; }
        start[t] = off;
    }
    taken.resize((size_t)above + hist[t]);
    double best_below = -1.0e300; // the best key of the buckets under the threshold bucket (for the ambiguity check)
    {
        uint32_t fill[BMAX];
        for (int b = t; b < B; b++) fill[b] = start[b];
        for (uint32_t i = 0; i < length; i++) {
            const int b = bucket[i];
            if (b >= t) taken[fill[b]++] = keyed[i];
            else if (keyed[i].key > best_below) best_below = keyed[i].key;
        }
    }
    WeightedKey *tb = taken.data() + start[t]; // the threshold bucket: only its best (amount - above) survive
    const uint32_t need = amount - above;
    if (need < hist[t]) std::nth_element(tb, tb + need, tb + hist[t], weighted_key_before);
    {   // order inside the buckets: `taken` is in bucket order already, so ONE insertion pass over the `amount` survivors moves
        // an element only past its own bucket's earlier members (a bucket holds a handful; a loop per bucket cost more than the sorting)
        WeightedKey *p = taken.data();
        for (uint32_t i = 1; i < amount; i++) {
            const WeightedKey x = p[i];
            uint32_t j = i;
            while (j > 0 && weighted_key_before(x, p[j - 1])) { p[j] = p[j - 1]; j--; }
            p[j] = x;
        }
    }
    out.resize(amount);
    for (uint32_t i = 0; i < amount; i++) out[i] = taken[i].index;
    if (ambiguous) {
        bool amb = false;
        for (uint32_t i = 0; i + 1 < amount; i++) amb = amb || !(taken[i].key - taken[i + 1].key >= guard);
        if (amount > 0 && amount < length) { // the best key that was left out: in the threshold bucket's remainder or in a lower bucket
            double best_out = -1.0e300;
            for (uint32_t i = need; i < hist[t]; i++) best_out = tb[i].key > best_out ? tb[i].key : best_out;
            best_out = best_below > best_out ? best_below : best_out;
            amb = amb || !(taken[amount - 1].key - best_out >= guard);
        }
        *ambiguous = amb;
    }
}

// log2(u) for u = 2^e (1 + f) in (0, 1), f in [0, 1):  e + f + 0.345 f (1 - f), within NQ_LOG2_APPROX_ERR of the true
// value (|log2(1 + f) - f - 0.345 f (1 - f)| <= 0.00787 on a 4 M-point grid, slope <= 0.8: 0.0085 with slack).
constexpr double LOG2_APPROX_ERR = 0.0085;
inline double approx_log2_unit(double u)
{
    uint64_t bits;
    memcpy(&bits, &u, 8);
    const int e = (int)(bits >> 52) - 1023; // u > 0 and normal here (u >= 2^-53)
    const uint64_t fb = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m;
    memcpy(&m, &fb, 8);
    const double f = m - 1.0;
    return (double)e + f + 0.345 * f * (1.0 - f);
}

// One key's bracket of t = y log2(u): the upper end, and the histogram bucket of the lower end (1/32 wide over [-8, 0),
// everything below -8 in bucket 0).  u == 0: key 0, below everything.
inline void weighted_bracket(double u, double y, double &h, int &b)
{
    double l = -1.0e300;
    h = -1.0e300;
    if (u > 0.0) {
        const double a = y * approx_log2_unit(u);
        const double half = y * LOG2_APPROX_ERR + 1e-12 * (1.0 - a); // bracket of t, rounding of a included
        l = a - half;
        h = a + half;
    }
    double t = (l + 8.0) * 32.0;
    t = t > 0.0 ? t : 0.0;
    t = t < 255.0 ? t : 255.0;
    b = (int)t;
}

#if defined(__x86_64__)
// the same for eight keys at a time (the same IEEE operations in the same order: identical values)
__attribute__((target("avx512f,avx512dq,avx512vl"))) inline void weighted_brackets_avx512(const double *us, const double *ys, uint32_t length,
                                                                                       double *hi, int32_t *bkt)
{
    const __m512d v1 = _mm512_set1_pd(1.0), vk = _mm512_set1_pd(0.345), verr = _mm512_set1_pd(LOG2_APPROX_ERR), veps = _mm512_set1_pd(1e-12),
                  vneg = _mm512_set1_pd(-1.0e300), v8 = _mm512_set1_pd(8.0), v32 = _mm512_set1_pd(32.0), v0 = _mm512_setzero_pd(),
                  v255 = _mm512_set1_pd(255.0);
    const __m512i mant = _mm512_set1_epi64(0x000fffffffffffffll), onebits = _mm512_set1_epi64(0x3ff0000000000000ll), bias = _mm512_set1_epi64(1023);
    uint32_t i = 0;
    for (; i + 8 <= length; i += 8) {
        const __m512d u = _mm512_loadu_pd(us + i), y = _mm512_loadu_pd(ys + i);
        const __m512i bits = _mm512_castpd_si512(u);
        const __m512d e = _mm512_cvtepi64_pd(_mm512_sub_epi64(_mm512_srli_epi64(bits, 52), bias));
        const __m512d m = _mm512_castsi512_pd(_mm512_or_si512(_mm512_and_si512(bits, mant), onebits));
        const __m512d f = _mm512_sub_pd(m, v1);
        const __m512d lg = _mm512_add_pd(_mm512_add_pd(e, f), _mm512_mul_pd(_mm512_mul_pd(vk, f), _mm512_sub_pd(v1, f)));
        const __m512d a = _mm512_mul_pd(y, lg);
        const __m512d half = _mm512_add_pd(_mm512_mul_pd(y, verr), _mm512_mul_pd(veps, _mm512_sub_pd(v1, a)));
        const __mmask8 pos = _mm512_cmp_pd_mask(u, v0, _CMP_GT_OQ);
        const __m512d l = _mm512_mask_blend_pd(pos, vneg, _mm512_sub_pd(a, half));
        const __m512d h = _mm512_mask_blend_pd(pos, vneg, _mm512_add_pd(a, half));
        _mm512_storeu_pd(hi + i, h);
        __m512d t = _mm512_mul_pd(_mm512_add_pd(l, v8), v32);
        t = _mm512_mask_blend_pd(_mm512_cmp_pd_mask(t, v0, _CMP_GT_OQ), v0, t);
        t = _mm512_mask_blend_pd(_mm512_cmp_pd_mask(t, v255, _CMP_LT_OQ), v255, t);
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(bkt + i), _mm512_cvttpd_epi32(t));
    }
    for (; i < length; i++) {
        int b;
        weighted_bracket(us[i], ys[i], hi[i], b);
        bkt[i] = b;
    }
}

// indices i with hi[i] >= cut, ascending
__attribute__((target("avx512f,avx512dq,avx512vl"))) inline uint32_t weighted_filter_avx512(const double *hi, uint32_t length, double cut, uint32_t *idx)
{
    const __m512d vcut = _mm512_set1_pd(cut);
    __m256i cur = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
    const __m256i eight = _mm256_set1_epi32(8);
    uint32_t n = 0, i = 0;
    for (; i + 8 <= length; i += 8) {
        const __mmask8 keep = _mm512_cmp_pd_mask(_mm512_loadu_pd(hi + i), vcut, _CMP_GE_OQ);
        _mm256_mask_compressstoreu_epi32(idx + n, keep, cur);
        n += (uint32_t)__builtin_popcount((unsigned)keep);
        cur = _mm256_add_epi32(cur, eight);
    }
    for (; i < length; i++)
        if (hi[i] >= cut) idx[n++] = i;
    return n;
}

// t = y log2(u) for the kept keys, accurate to ~1e-15 absolute (u in [2^-53, 1), y in (0, 1]): u = 2^e m, m in [sqrt(1/2), sqrt 2),
// ln m = 2 atanh(f), f = (m - 1) / (m + 1) <= 0.1716, twelve terms of the series (truncation < 1e-20).  The ORDER of the keys
// u^y = 2^t is the order of t wherever two t differ by more than the guard of index_sample_weighted (1e-11 >> this error + pow's
// < 1 ulp): eight keys per instruction instead of one libm pow per key.
__attribute__((target("avx512f,avx512dq,avx512vl"))) inline void weighted_log2_keys_avx512(const double *us, const double *ys, const uint32_t *idx,
                                                                                        uint32_t n_keep, WeightedKey *keyed, uint8_t *bkt8 = nullptr,
                                                                                        double floor_t = 0.0, double scale = 0.0, int B = 0)
{
    const __m512d vfloor = _mm512_set1_pd(floor_t), vscale = _mm512_set1_pd(scale), vtop = _mm512_set1_pd((double)(B - 1)), vzero = _mm512_setzero_pd();
    const __m512d v1 = _mm512_set1_pd(1.0), vhalf = _mm512_set1_pd(0.5), vsqrt2 = _mm512_set1_pd(1.4142135623730951),
                  vlog2e2 = _mm512_set1_pd(2.0 * 1.4426950408889634);
    uint32_t j = 0;
    for (; j < n_keep; j += 8) {
        const uint32_t left = n_keep - j;
        const __mmask8 live = left >= 8 ? (__mmask8)0xff : (__mmask8)((1u << left) - 1u);
        const __m256i vi = _mm256_maskz_loadu_epi32(live, idx + j);
        const __m512d u = _mm512_mask_i32gather_pd(v1, live, vi, us, 8), y = _mm512_mask_i32gather_pd(v1, live, vi, ys, 8);
        __m512d e = _mm512_getexp_pd(u);
        __m512d m = _mm512_getmant_pd(u, _MM_MANT_NORM_1_2, _MM_MANT_SIGN_src);
        const __mmask8 big = _mm512_cmp_pd_mask(m, vsqrt2, _CMP_GT_OQ);
        m = _mm512_mask_mul_pd(m, big, m, vhalf);
        e = _mm512_mask_add_pd(e, big, e, v1);
        const __m512d f = _mm512_div_pd(_mm512_sub_pd(m, v1), _mm512_add_pd(m, v1));
        const __m512d z = _mm512_mul_pd(f, f);
        __m512d p = _mm512_set1_pd(1.0 / 23.0);
        p = _mm512_fmadd_pd(p, z, _mm512_set1_pd(1.0 / 21.0));
        p = _mm512_fmadd_pd(p, z, _mm512_set1_pd(1.0 / 19.0));
        p = _mm512_fmadd_pd(p, z, _mm512_set1_pd(1.0 / 17.0));
        p = _mm512_fmadd_pd(p, z, _mm512_set1_pd(1.0 / 15.0));
        p = _mm512_fmadd_pd(p, z, _mm512_set1_pd(1.0 / 13.0));
        p = _mm512_fmadd_pd(p, z, _mm512_set1_pd(1.0 / 11.0));
        p = _mm512_fmadd_pd(p, z, _mm512_set1_pd(1.0 / 9.0));
        p = _mm512_fmadd_pd(p, z, _mm512_set1_pd(1.0 / 7.0));
        p = _mm512_fmadd_pd(p, z, _mm512_set1_pd(1.0 / 5.0));
        p = _mm512_fmadd_pd(p, z, _mm512_set1_pd(1.0 / 3.0));
        p = _mm512_fmadd_pd(p, z, v1);
        const __m512d lg = _mm512_fmadd_pd(_mm512_mul_pd(f, p), vlog2e2, e); // e + 2 f p log2(e)
        const __m512d t = _mm512_mul_pd(y, lg);
        alignas(64) double tt[8];
        _mm512_store_pd(tt, t);
        for (uint32_t q = 0; q < 8 && q < left; q++) keyed[j + q] = WeightedKey{tt[q], idx[j + q]};
        if (bkt8) { // weighted_top's bucket of every key: clamp((t - floor) scale, 0, B - 1), non-decreasing in t
            __m512d b = _mm512_mul_pd(_mm512_sub_pd(t, vfloor), vscale);
            b = _mm512_min_pd(_mm512_max_pd(b, vzero), vtop);
            _mm_storel_epi64(reinterpret_cast<__m128i *>(bkt8 + j), _mm256_cvtepi32_epi8(_mm512_cvttpd_epi32(b))); // eight bytes: the caller pads by 8
        }
    }
}

inline bool weighted_have_avx512()
{
    static const bool v = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl") &&
                          std::getenv("LS_HOST_NO_AVX512") == nullptr && std::getenv("LS_HOST_NO_AVX2") == nullptr;
    return v;
}
#endif

#if defined(__x86_64__)
// y = 1 / w for every weight (IEEE division, the scalar path's quotient); false if a weight is negative or NaN
__attribute__((target("avx512f,avx512dq,avx512vl"))) inline bool weighted_exponents_avx512(const double *w, uint32_t length, double *ys, bool *all_ge1)
{
    const __m512d v1 = _mm512_set1_pd(1.0), v0 = _mm512_setzero_pd();
    __mmask8 bad = 0, small = 0;
    uint32_t i = 0;
    for (; i + 8 <= length; i += 8) {
        const __m512d x = _mm512_loadu_pd(w + i);
        bad |= _mm512_cmp_pd_mask(x, v0, _CMP_NGE_UQ);   // !(w >= 0)
        small |= _mm512_cmp_pd_mask(x, v1, _CMP_NGE_UQ); // !(w >= 1)
        _mm512_storeu_pd(ys + i, _mm512_div_pd(v1, x));
    }
    bool bad_tail = false, small_tail = false;
    for (; i < length; i++) {
        bad_tail = bad_tail || !(w[i] >= 0.0);
        small_tail = small_tail || !(w[i] >= 1.0);
        ys[i] = 1.0 / w[i];
    }
    *all_ge1 = !small && !small_tail;
    return !bad && !bad_tail;
}
#endif

inline bool weighted_surrogate_on() // LS_HOST_EXACT_POW=1: order the survivors by libm pow keys (the round-2 path; tests run both)
{
    static const bool v = std::getenv("LS_HOST_EXACT_POW") == nullptr;
    return v;
}
inline uint64_t &weighted_surrogate_fallbacks() // calls of this thread that found two keys within the guard and took the exact path
{
    static thread_local uint64_t n = 0;
    return n;
}

// -DLS_SAMPLER_PROFILE (tools/sampler_bench.cpp only): time-stamp-counter ticks per section of index_sample_weighted
#ifdef LS_SAMPLER_PROFILE
inline unsigned long long *sampler_prof() { static thread_local unsigned long long t[8] = {0}; return t; }
#define LS_SP_BEGIN unsigned long long sp_last_ = __builtin_ia32_rdtsc()
#define LS_SP(i) do { const unsigned long long now_ = __builtin_ia32_rdtsc(); sampler_prof()[i] += now_ - sp_last_; sp_last_ = now_; } while (0)
#else
#define LS_SP_BEGIN do { } while (0)
#define LS_SP(i) do { } while (0)
#endif

template <class R> inline std::vector<uint32_t> index_sample_weighted(R &rng, const double *weights, uint32_t length, uint32_t amount);
template <class R> inline std::vector<uint32_t> index_sample_weighted(R &rng, const std::vector<double> &weights, uint32_t amount)
{
    return index_sample_weighted(rng, weights.data(), (uint32_t)weights.size(), amount);
}
// (pointer + length: the proposer keeps its weights in a buffer of n entries of which the first `length` count)
template <class R> inline std::vector<uint32_t> index_sample_weighted(R &rng, const double *weights, uint32_t length, uint32_t amount)
{
    LS_SP_BEGIN;
    if (amount == 0) return {};
    if (amount > length) throw std::runtime_error("`amount` of samples must be less than or equal to `length`");
    static thread_local std::vector<WeightedKey> keyed;
    static thread_local std::vector<double> us, ys, hi;
    static thread_local std::vector<int32_t> bkt;
    static thread_local std::vector<uint32_t> idx;
    // Pre-filter (pays when few of many are wanted): most keys cannot reach the top `amount`, and that can be PROVEN
    // from a cheap bracket of t_i = y_i log2(u_i) (key_i = 2^t_i up to pow's < 1 ulp error): if hi_i lies below the
    // amount-th largest lo_j by more than 1e-9, at least `amount` keys are strictly larger than key_i whatever the
    // roundings.  pow() is then evaluated for the survivors only (about 240 of 1100 in the N-queens proposer); every
    // draw is still taken from the generator, in order.  Needs every exponent y = 1 / w <= 1 (no underflow: t >= -53).
    bool prefilter = length >= 64 && (uint64_t)amount * 2 <= length;
    us.resize(length);
    ys.resize(length);
    // 1 / w for small integer weights (column scores) from a table of the very same quotients: a division less per key
    static const std::vector<double> inv_small = [] {
        std::vector<double> t(1024);
        for (int k = 0; k < 1024; k++) t[(size_t)k] = 1.0 / (double)k;
        return t;
    }();
    bool bulk = false;
#if defined(__x86_64__)
    if constexpr (std::is_same<R, ChaCha20Rng>::value) {
        if (weighted_have_avx512()) { // the draws in one run of vector conversions, the exponents by vector division (the same IEEE quotients)
            bool all_ge1 = true;
            if (!weighted_exponents_avx512(weights, length, ys.data(), &all_ge1)) throw std::runtime_error("invalid weight");
            rng.fill_f64(us.data(), length);
            prefilter = prefilter && all_ge1;
            bulk = true;
        }
    }
#endif
    if (!bulk)
        for (uint32_t i = 0; i < length; i++) {
            const double w = weights[i];
            if (!(w >= 0.0)) throw std::runtime_error("invalid weight");
            us[i] = rng.gen_f64();
            const int wi = w < 1024.0 ? (int)w : -1;
            ys[i] = (wi >= 0 && (double)wi == w) ? inv_small[(size_t)wi] : 1.0 / w;
            prefilter = prefilter && w >= 1.0;
        }
    LS_SP(0); // draws + exponents
    double key_floor = 0.0; // where the surviving keys start (a hint for the bucket mapping of weighted_top)
    if (!prefilter) {
        keyed.resize(length);
        for (uint32_t i = 0; i < length; i++) keyed[i] = WeightedKey{std::pow(us[i], ys[i]), i};
    } else {
        hi.resize(length);
        bkt.resize(length);
        idx.resize(length);
        constexpr int B = 256; // histogram of the lower brackets over [-8, 0), 1/32 wide; everything below -8 in bucket 0
        uint32_t hist[B] = {0};
#if defined(__x86_64__)
        const bool wide = weighted_have_avx512();
        if (wide) weighted_brackets_avx512(us.data(), ys.data(), length, hi.data(), bkt.data());
        else
#endif
            for (uint32_t i = 0; i < length; i++) {
                int b;
                weighted_bracket(us[i], ys[i], hi[i], b);
                bkt[i] = b;
            }
        LS_SP(1); // brackets
        for (uint32_t i = 0; i < length; i++) hist[bkt[i]]++;
        int t = B - 1;
        uint32_t above = 0;
        while (t > 0 && above + hist[t] < amount) { above += hist[t]; t--; }
        // at least `amount` lower brackets are >= the lower edge of bucket t (bucket 0 has no lower edge: keep everything)
        const double cut = t == 0 ? -2.0e300 : (double)t / 32.0 - 8.0 - 1e-9;
        uint32_t n_keep = 0;
        LS_SP(2); // histogram + threshold
#if defined(__x86_64__)
        if (wide) n_keep = weighted_filter_avx512(hi.data(), length, cut, idx.data());
        else
#endif
            for (uint32_t i = 0; i < length; i++)
                if (hi[i] >= cut) idx[n_keep++] = i;
        keyed.resize(n_keep);
        LS_SP(3); // filter
#if defined(__x86_64__)
        if (wide && t > 0 && weighted_surrogate_on()) {
            // Order the survivors by t = y log2(u) (weighted_log2_keys_avx512) instead of by pow(u, y): the same order wherever two t
            // differ by more than 1e-11 (pow's < 1 ulp error and the surrogate's ~1e-15 are four orders of magnitude below it); if any
            // two keys whose order matters are closer than that (probability ~1e-7 per call), fall through to the exact keys.
            bool u_zero = false;
            for (uint32_t j = 0; j < n_keep; j++) u_zero = u_zero || !(us[idx[j]] > 0.0);
            if (!u_zero) {
                static thread_local std::vector<uint8_t> bkt8;
                bkt8.resize((size_t)n_keep + 8);
                const int TB = weighted_top_buckets(n_keep);
                weighted_log2_keys_avx512(us.data(), ys.data(), idx.data(), n_keep, keyed.data(), bkt8.data(), cut, (double)TB / (0.0 - cut), TB);
                LS_SP(4); // log2 keys of the survivors
                std::vector<uint32_t> out;
                bool ambiguous = true;
                weighted_top(keyed, amount, out, cut, 0.0, 1e-11, &ambiguous, bkt8.data());
                LS_SP(5); // top `amount` in order
                if (!ambiguous) return out;
                weighted_surrogate_fallbacks()++;
            }
        }
#endif
        for (uint32_t j = 0; j < n_keep; j++) {
            const uint32_t i = idx[j];
            keyed[j] = WeightedKey{std::pow(us[i], ys[i]), i};
        }
        if (t > 0) key_floor = std::exp2(cut);
    }
    std::vector<uint32_t> out;
    weighted_top(keyed, amount, out, key_floor);
    return out;
}

} // namespace lshost
