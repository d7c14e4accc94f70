// CPU-side unit test of csrc/host/fibers.hpp (ls_multi_solver's M:N fibre pool; host logic only, no GPU, no HIP).
//  * every fibre runs to completion whatever the number of workers, yields resume where they left off (locals, floating-point
//    control state and callee-saved registers survive a switch, also onto another thread),
//  * an idle worker takes suspended fibres from a busy one (the restarts that are left spread over the cores),
//  * exceptions thrown and caught inside a fibre unwind on the fibre's own stack,
//  * deep frames fit the mmap'ed stack.
// Build + run: tests/test_host_batcher.py::test_fibre_pool.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <set>
#include <stdexcept>
#include <thread>
#include <vector>

#include <sys/syscall.h>
#include <unistd.h>

#include "../../constraint-solver_amd/csrc/host/fibers.hpp"

#if defined(__x86_64__)
asm(LS_FIBER_SWITCH_ASM);
#endif

using lshost::FiberPool;

static int fail(const char *what)
{
    printf("FAILED: %s\n", what);
    return 1;
}

static long deep(int depth, volatile char *sink)
{
    char buf[512];
    for (int i = 0; i < 512; i++) buf[i] = (char)(depth + i);
    *sink = buf[depth & 511];
    if (depth == 0) { FiberPool::yield(); return buf[7]; }
    return deep(depth - 1, sink) + buf[3];
}

int main()
{
    for (int n_workers : {1, 2, 4, 7}) {
        for (int n_fibres : {1, 3, 16, 61}) {
          for (int all_on_worker_0 = 0; all_on_worker_0 < 2; all_on_worker_0++) {
            FiberPool pool(n_workers);
            std::vector<long> sums((size_t)n_fibres, 0);
            std::vector<std::set<long>> seen((size_t)n_fibres); // kernel thread ids (pthread_self() is `const`: the compiler hoists it out of the loop)
            std::atomic<long> total_yields{0};
            for (int f = 0; f < n_fibres; f++)
                pool.add([&, f]() {
                    // unequal lengths: fibre f needs (f % 5 + 1) * 400 steps -- the long ones must migrate to idle workers
                    const int steps = (f % 5 + 1) * 400;
                    double acc = 1.0;
                    long local = 0;
                    for (int i = 0; i < steps; i++) {
                        local += i ^ f;
                        acc = std::sqrt(acc * 1.0001 + (double)i);
                        seen[(size_t)f].insert((long)syscall(SYS_gettid));
                        { volatile double w = 1; for (int k = 0; k < 100; k++) w = w * 1.0001 + k; } // the host work between two yields
                        if (i % 37 == 5) {
                            try { throw std::runtime_error("inside a fibre"); } catch (const std::exception &) { local += 1; }
                        }
                        FiberPool::yield();
                        total_yields.fetch_add(1, std::memory_order_relaxed);
                    }
                    if (f == 2) { volatile char sink = 0; local += deep(600, &sink); } // ~300 KB of frames, a yield at the bottom
                    sums[(size_t)f] = local + (acc > 0 ? 0 : 1);
                }, all_on_worker_0 ? 0 : f);
            pool.run();
            long want_yields = 0;
            for (int f = 0; f < n_fibres; f++) {
                const int steps = (f % 5 + 1) * 400;
                long want = 0;
                for (int i = 0; i < steps; i++) { want += i ^ f; if (i % 37 == 5) want += 1; }
                want_yields += steps;
                if (f == 2) { volatile char sink = 0; char b7 = (char)(0 + 7); long d = b7; for (int dep = 1; dep <= 600; dep++) d += (char)(dep + 3); want += d; (void)sink; }
                if (sums[(size_t)f] != want) { printf("workers %d fibres %d fibre %d: %ld != %ld\n", n_workers, n_fibres, f, sums[(size_t)f], want); return fail("fibre result"); }
            }
            if (total_yields.load() != want_yields) return fail("yield count");
            if (n_workers >= 2 && n_fibres >= 16 && all_on_worker_0) { // every other worker starts idle: it must take fibres from worker 0
                std::set<long> tids;
                for (auto &s : seen) tids.insert(s.begin(), s.end());
                if (pool.steals() == 0 || tids.size() < 2) return fail("no fibre ever ran on another worker than its home");
            }
          }
        }
    }
    // a fibre that never yields and an empty pool
    { FiberPool pool(3); int x = 0; pool.add([&]() { x = 42; }, 5); pool.run(); if (x != 42) return fail("single fibre"); }
    { FiberPool pool(2); pool.run(); }
    // priorities and readiness: a fibre is not resumed before what it waits for has arrived, and of the ready fibres the one
    // with the highest priority runs first
    {
        FiberPool pool(1);
        static volatile uint32_t line[2] = {0, 0};
        std::vector<int> order;
        auto waiter = [&](int id, uint64_t prio) {
            return [&order, id, prio]() {
                FiberPool::set_priority(prio);
                lshost::FiberWait w; w.words = line; w.n_words = 2; w.tag = 7;
                FiberPool::yield(&w);
                if ((line[0] >> 16) != 7 || (line[1] >> 16) != 7) order.push_back(-1); // resumed too early
                order.push_back(id);
            };
        };
        pool.add(waiter(1, 5), 0);
        pool.add(waiter(2, 9), 0);
        pool.add(waiter(3, 7), 0);
        pool.add([&]() { // priority 0: runs whenever nobody else can
            order.push_back(0);
            line[0] = 7u << 16;
            FiberPool::yield(); // half a line: still nobody else can run
            order.push_back(0);
            line[1] = (7u << 16) | 3u;
            FiberPool::yield(); // now the three waiters go first, highest priority first
            order.push_back(0);
        }, 0);
        pool.run();
        if (order != std::vector<int>({0, 0, 2, 3, 1, 0})) { for (int x : order) printf("%d ", x); return fail("priority / readiness order"); }
    }
    // an answer that only arrives once the waiter acts (a drained server is launched again by the waiter, srv_common.hpp
    // SrvHost::wait): a worker whose fibres all wait resumes them anyway after a while, the one that has waited longest first
    for (int workers : {1, 3}) {
        FiberPool pool(workers);
        static volatile uint32_t never[6];
        for (auto &x : never) x = 0;
        std::vector<int> resumed(6, 0);
        for (int id = 0; id < 6; id++)
            pool.add([&resumed, id]() {
                lshost::FiberWait w; w.words = never + id; w.n_words = 1; w.tag = 9;
                while ((never[id] >> 16) != 9) {
                    FiberPool::yield(&w);
                    if (++resumed[(size_t)id] == 5) never[id] = 9u << 16; // the waiter's own doing, on its fifth look
                }
            }, id);
        pool.run();
        for (int id = 0; id < 6; id++)
            if (resumed[(size_t)id] < 5) return fail("a fibre whose answer never came on its own was not resumed");
    }
    // yield() outside any fibre is a no-op
    FiberPool::yield();
    printf("OK\n");
    return 0;
}
