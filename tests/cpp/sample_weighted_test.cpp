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
    printf("OK\n");
    return 0;
}
