// Diagnostic: where the N-queens proposer's weighted sample (1100 keys, 204 picks) spends its host time.
//   g++ -O3 -std=c++17 -I constraint-solver_amd/csrc -o /tmp/sampler_bench tools/sampler_bench.cpp && /tmp/sampler_bench
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
    printf("full sampler %.2f us (acc %llu)\n", std::chrono::duration<double>(t1-t0).count()/reps*1e6, acc);
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
