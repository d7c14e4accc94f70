// Diagnostic: where the N-queens proposer's weighted sample (1100 keys, 204 picks) spends its host time.
//   g++ -O3 -std=c++17 -I constraint-solver_amd/csrc -o /tmp/sampler_bench tools/sampler_bench.cpp && /tmp/sampler_bench
#define LS_SAMPLER_PROFILE 1
#include "host/rng.hpp"
#include <chrono>
#include <cstdio>
using namespace lshost;
int main(){
    ChaCha20Rng rng = ChaCha20Rng::from_seed_string("42");
    std::vector<double> w(1100);
    for (size_t i=0;i<w.size();i++) w[i] = 1 + (i*7)%5;
    const int reps=20000;
    auto t0=std::chrono::steady_clock::now();
    unsigned long long acc=0;
    for(int r=0;r<reps;r++){ auto v=index_sample_weighted(rng,w,204); acc+=v[0]; }
    auto t1=std::chrono::steady_clock::now();
    const double full_us = std::chrono::duration<double>(t1-t0).count()/reps*1e6;
    printf("full sampler %.2f us (acc %llu)\n", full_us, acc);
    {   // the sections of the real function (time-stamp-counter ticks, scaled to the measured total)
        const char *name[6] = {"draws + exponents", "brackets", "histogram + threshold", "filter", "log2 keys of the survivors", "top 204 in order"};
        unsigned long long sum = 0;
        for (int i = 0; i < 6; i++) sum += sampler_prof()[i];
        for (int i = 0; i < 6; i++) printf("  %-28s %5.2f us\n", name[i], full_us * (double)sampler_prof()[i] / (double)sum);
    }
    {   // the sampler's own sections (a copy of index_sample_weighted's pre-filter path with timers)
        const uint32_t amount = 204, length = 1100;
        std::vector<WeightedKey> keyed;
        std::vector<double> us(length), ys(length), lo(length), hi(length);
        double sec[5] = {0, 0, 0, 0, 0};
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto us_of = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count() * 1e6; };
        for (int r = 0; r < reps; r++) {
            auto a0 = now();
            for (uint32_t i = 0; i < length; i++) us[i] = rng.gen_f64();
            auto a1 = now();
            for (uint32_t i = 0; i < length; i++) ys[i] = 1.0 / w[i];
            auto a2 = now();
            constexpr int B = 256;
            uint32_t hist[B] = {0};
            for (uint32_t i = 0; i < length; i++) {
                double l = -1.0e300, h = -1.0e300;
                if (us[i] > 0.0) {
                    const double a = ys[i] * approx_log2_unit(us[i]);
                    const double half = ys[i] * LOG2_APPROX_ERR + 1e-12 * (1.0 - a);
                    l = a - half;
                    h = a + half;
                }
                lo[i] = l;
                hi[i] = h;
                int b = l <= -8.0 ? 0 : (int)((l + 8.0) * 32.0);
                b = b < 0 ? 0 : (b > B - 1 ? B - 1 : b);
                hist[b]++;
            }
            auto a3 = now();
            int t = B - 1;
            uint32_t above = 0;
            while (t > 0 && above + hist[t] < amount) { above += hist[t]; t--; }
            const double cut = t == 0 ? -2.0e300 : (double)t / 32.0 - 8.0 - 1e-9;
            keyed.clear();
            for (uint32_t i = 0; i < length; i++)
                if (hi[i] >= cut) keyed.push_back(WeightedKey{std::pow(us[i], ys[i]), i});
            auto a4 = now();
            std::vector<uint32_t> out;
            weighted_top(keyed, amount, out, t > 0 ? std::exp2(cut) : 0.0);
            auto a5 = now();
            acc += out[0] + keyed.size();
            sec[0] += us_of(a0, a1); sec[1] += us_of(a1, a2); sec[2] += us_of(a2, a3); sec[3] += us_of(a3, a4); sec[4] += us_of(a4, a5);
        }
        const char *names[5] = {"draws", "reciprocals", "brackets + histogram", "filter + pow of the survivors", "top 204 in order"};
        for (int i = 0; i < 5; i++) printf("  section %-30s %6.2f us\n", names[i], sec[i] / reps);
    }
    // pieces
    double s=0; t0=std::chrono::steady_clock::now();
    for(int r=0;r<reps;r++) for(int i=0;i<1100;i++) s+=rng.gen_f64();
    t1=std::chrono::steady_clock::now();
    printf("1100 gen_f64 %.2f us (%g)\n", std::chrono::duration<double>(t1-t0).count()/reps*1e6, s);
    std::vector<double> us(1100), ys(1100);
    for(int i=0;i<1100;i++){us[i]=rng.gen_f64(); ys[i]=1.0/w[i];}
    t0=std::chrono::steady_clock::now();
    for(int r=0;r<reps;r++) for(int i=0;i<240;i++) s+=std::pow(us[i]+r*1e-9,ys[i]);
    t1=std::chrono::steady_clock::now();
    printf("240 pow %.2f us (%g)\n", std::chrono::duration<double>(t1-t0).count()/reps*1e6, s);
    t0=std::chrono::steady_clock::now();
    for(int r=0;r<reps;r++) for(int i=0;i<1100;i++) s+=ys[i]*approx_log2_unit(us[i]);
    t1=std::chrono::steady_clock::now();
    printf("1100 approx log2 %.2f us (%g)\n", std::chrono::duration<double>(t1-t0).count()/reps*1e6, s);
    t0=std::chrono::steady_clock::now();
    for(int r=0;r<reps;r++) for(int i=0;i<1100;i++) s+=1.0/(w[i]+r*1e-9);
    t1=std::chrono::steady_clock::now();
    printf("1100 div %.2f us (%g)\n", std::chrono::duration<double>(t1-t0).count()/reps*1e6, s);
    std::vector<WeightedKey> keyed(240); std::vector<uint32_t> out;
    t0=std::chrono::steady_clock::now();
    for(int r=0;r<reps;r++){ for(int i=0;i<240;i++) keyed[i]=WeightedKey{0.5+0.5*us[(i+r)%1100],(uint32_t)i}; keyed.resize(240); weighted_top(keyed,204,out); s+=out[0]; }
    t1=std::chrono::steady_clock::now();
    printf("weighted_top(240->204) %.2f us (%g)\n", std::chrono::duration<double>(t1-t0).count()/reps*1e6, s);
}
