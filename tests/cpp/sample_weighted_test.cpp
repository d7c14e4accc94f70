// CPU unit test: the histogram-based top-k of index_sample_weighted (csrc/host/rng.hpp) returns exactly what the
// selection + sort formulation (the reference's BinaryHeap pops, rand 0.8.5 seq::index::sample_weighted) returns, and
// consumes the generator identically -- many shapes, including zero weights (keys tie at 0), huge weights (keys crowd just
// below 1: one histogram bucket) and amount == 0 / == length.
#include <cstdio>
#include <string>

#include "../../constraint-solver_amd/csrc/host/rng.hpp"

using namespace lshost;

int main()
{
    ChaCha20Rng g = ChaCha20Rng::from_seed_string("shapes");
    for (int rep = 0; rep < 60000; rep++) {
        const uint32_t L = 1 + (uint32_t)(g.next_u32() % (rep % 40 == 0 ? 3000 : 48));
        const uint32_t k = (uint32_t)(g.next_u32() % (L + 1));
        std::vector<double> w(L);
        const int mode = (int)(g.next_u32() % 5);
        for (uint32_t i = 0; i < L; i++) {
            switch (mode) {
            case 0: w[i] = 1 + (g.next_u32() % 8) + 1e-4; break;                                  // the N-queens proposer's weights
            case 1: w[i] = (g.next_u32() % 3 == 0) ? 0.0 : 1e-3 * (1 + g.next_u32() % 5); break;
            case 2: w[i] = 1e6 + g.next_u32() % 10; break;
            case 3: w[i] = 0.05 + 0.01 * (g.next_u32() % 100); break;
            default: w[i] = (double)(1 + g.next_u32() % 1000); break;
            }
        }
        ChaCha20Rng r1 = ChaCha20Rng::from_seed_string(("t" + std::to_string(rep)).c_str());
        ChaCha20Rng r2 = r1;
        const auto a = index_sample_weighted_reference(r1, w, k);
        const auto b = index_sample_weighted(r2, w, k);
        if (a != b || r1.next_u64() != r2.next_u64()) {
            printf("FAIL rep %d (length %u amount %u mode %d)\n", rep, L, k, mode);
            return 1;
        }
    }
    // the pre-filter's domain (length >= 64, amount <= length / 2, every weight >= 1): brackets from the cheap log2, pow
    // only for the keys that can still reach the top
    for (int rep = 0; rep < 12000; rep++) {
        const uint32_t L = 64 + (uint32_t)(g.next_u32() % 2400);
        const uint32_t k = 1 + (uint32_t)(g.next_u32() % (L / 2));
        std::vector<double> w(L);
        const int mode = (int)(g.next_u32() % 6);
        for (uint32_t i = 0; i < L; i++) {
            switch (mode) {
            case 0: w[i] = 1 + (g.next_u32() % 8) + 1e-4; break;          // the N-queens proposer's weights
            case 1: w[i] = 1.0; break;                                    // key == u: ties only where draws tie
            case 2: w[i] = 1e6 + g.next_u32() % 10; break;                // every key within 1e-5 of 1: the brackets overlap massively
            case 3: w[i] = 1.0 + 1e-3 * (g.next_u32() % 3); break;        // nearly equal exponents
            case 4: w[i] = (double)(1 + g.next_u32() % 1000); break;
            default: w[i] = (i % 7 == 0) ? 1e9 : 2.0001; break;           // a few keys pinned to ~1
            }
        }
        ChaCha20Rng r1 = ChaCha20Rng::from_seed_string(("p" + std::to_string(rep)).c_str());
        ChaCha20Rng r2 = r1;
        const auto a = index_sample_weighted_reference(r1, w, k);
        const auto b = index_sample_weighted(r2, w, k);
        if (a != b || r1.next_u64() != r2.next_u64()) {
            printf("FAIL (pre-filter) rep %d (length %u amount %u mode %d)\n", rep, L, k, mode);
            return 1;
        }
    }
    // the approximation the brackets rest on, on a dense sweep of mantissas and a few exponents
    for (int e = -53; e <= -1; e += 13)
        for (int i = 0; i < 2000000; i++) {
            const double u = std::ldexp(1.0 + (double)i / 2000000.0, e);
            if (!(u < 1.0)) continue;
            if (std::fabs(approx_log2_unit(u) - std::log2(u)) > LOG2_APPROX_ERR - 4e-4) { printf("FAIL log2 bound at %a\n", u); return 1; }
        }
#if defined(__x86_64__)
    // the surrogate keys t = y log2(u) that order the pre-filter's survivors (round 3): accurate to well under the 1e-11 guard --
    // against long-double log2 on a dense sweep of mantissas, every binade of gen_f64's range, exponents y = 1 / w as the proposer has them
    if (weighted_have_avx512()) {
        double worst = 0;
        std::vector<double> us, ys;
        std::vector<uint32_t> idx;
        for (int e = -53; e <= -1; e++)
            for (int i = 0; i < 60000; i++) {
                const double u = std::ldexp(1.0 + ((double)i + 0.37) / 60000.0, e);
                if (!(u < 1.0)) continue;
                us.push_back(u);
                ys.push_back(1.0 / (1.0 + (double)(i % 97) + 1e-4));
                idx.push_back((uint32_t)idx.size());
            }
        std::vector<WeightedKey> k(us.size());
        weighted_log2_keys_avx512(us.data(), ys.data(), idx.data(), (uint32_t)us.size(), k.data());
        for (size_t i = 0; i < us.size(); i++) {
            const long double t = (long double)ys[i] * log2l((long double)us[i]);
            const double err = (double)fabsl((long double)k[i].key - t);
            worst = err > worst ? err : worst;
            if (k[i].index != idx[i]) { printf("FAIL surrogate index\n"); return 1; }
        }
        printf("surrogate keys: worst |t - y log2 u| = %.3g over %zu keys; calls that fell back to exact pow keys: %llu\n", worst, us.size(),
               (unsigned long long)weighted_surrogate_fallbacks());
        if (!(worst < 2e-14)) { printf("FAIL surrogate accuracy\n"); return 1; }
        if (weighted_surrogate_on() && weighted_surrogate_fallbacks() == 0) { printf("FAIL: the guard never fired (mode 2 crowds keys within 1e-11)\n"); return 1; }
    }
#endif
    // fill_f64 = n gen_f64 draws, from any position of the stream (odd word positions, runs across group seams, n < 8)
    {
        ChaCha20Rng pick = ChaCha20Rng::from_seed_string("fill");
        for (int rep = 0; rep < 4000; rep++) {
            ChaCha20Rng a = ChaCha20Rng::from_seed_string(("f" + std::to_string(rep % 7)).c_str());
            const uint32_t skip = pick.next_u32() % 700, n = pick.next_u32() % (rep % 10 == 0 ? 5000 : 300);
            for (uint32_t i = 0; i < skip; i++) (void)a.next_u32();
            if (rep % 3 == 0) while (a.prefetch()) { }
            ChaCha20Rng b = a;
            std::vector<double> x(n), y(n);
            a.fill_f64(x.data(), n);
            for (uint32_t i = 0; i < n; i++) y[i] = b.gen_f64();
            if (x != y || a.words_drawn() != b.words_drawn() || a.next_u64() != b.next_u64()) {
                printf("fill_f64 differs from gen_f64: skip %u n %u\n", skip, n);
                return 1;
            }
        }
    }
    printf("OK\n");
    return 0;
}
