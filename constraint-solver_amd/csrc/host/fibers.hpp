// host/fibers.hpp — cooperative fibres for ls_multi_solver: many independent restarts share a few host threads.
//
// Why: one LS iteration of a restart is host work (move proposer, History: ~12 us at n = 4096) followed by a wait for the
// GPU's answer (~6 us).  A restart that waits hands its thread to another restart (the wait hook of the completion poll),
// so the cores stay busy.  Round 2 pinned every fibre to one thread; the run then ended with most threads idle while the
// thread that happened to own the longest restarts still juggled four of them.  Here the fibres of a run form one pool:
//  * every worker thread has its own set of fibres (a restart's working set -- boards, column scores, History: > 100 KB at
//    n = 4096 -- stays in that core's cache: a pool in which any worker took any ready fibre was measured 1.8 x SLOWER);
//  * at a scheduling point (its fibre yielded or finished) a worker takes, among its suspended fibres whose awaited answer
//    HAS ARRIVED (a fibre says what it waits for when it yields), the one with the highest priority -- ls_multi_solver uses
//    the LS iterations a restart has done so far: restart lengths are heavy-tailed (1.8 k to 62 k iterations at n = 4096),
//    the restart that has run longest is the one most likely to run longest still, and the run ends when the longest chain
//    ends, so that chain should wait for a core as little as possible;
//  * a worker with nothing left takes the highest-priority suspended fibre of the worker that holds most.
// Results do not depend on any of this: restarts are independent.  The reference has no counterpart (it is single-threaded;
// its only multi-run notion is the seed loop, examples/nqueens/src/main.rs:167).
//
// Switching is a dozen instructions (callee-saved registers + stack pointer; no signal-mask system call as in
// swapcontext), stacks are mmap'ed with a guard page below them (an overflow faults instead of corrupting the heap) and as
// large as a thread's (8 MB, committed lazily): a fibre calls into the HIP runtime, whose stack needs are not ours to bound.
// A fibre only ever switches inside FiberPool::yield(), never inside a runtime call; after a switch it may be on another
// thread, so nothing thread-local is cached across yield() (current_worker() is re-read, noinline).
#pragma once

#include <sched.h>
#include <sys/mman.h>
#include <time.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <thread>
#include <vector>

// spin-wait hint: `pause` on x86-64, nothing elsewhere (the ucontext fallback below is what a non-x86 host builds)
static inline void fiber_cpu_relax()
{
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

#if !defined(__x86_64__)
#include <ucontext.h>
#endif

namespace lshost {

#if defined(__x86_64__)
struct FiberRegs { void *rsp; };
extern "C" void ls_fiber_switch(FiberRegs *from, FiberRegs *to);
// saves rbp rbx r12-r15, MXCSR and the x87 control word on the current stack, stores rsp, loads the other side's
#define LS_FIBER_SWITCH_ASM                                                                                            \
    ".text\n.globl ls_fiber_switch\n.type ls_fiber_switch,@function\nls_fiber_switch:\n"                               \
    "pushq %rbp\npushq %rbx\npushq %r12\npushq %r13\npushq %r14\npushq %r15\nsubq $8, %rsp\n"                          \
    "stmxcsr (%rsp)\nfnstcw 4(%rsp)\nmovq %rsp, (%rdi)\nmovq (%rsi), %rsp\nldmxcsr (%rsp)\nfldcw 4(%rsp)\n"            \
    "addq $8, %rsp\npopq %r15\npopq %r14\npopq %r13\npopq %r12\npopq %rbx\npopq %rbp\nret\n"                           \
    ".size ls_fiber_switch, .-ls_fiber_switch\n"
#endif

// What a suspended fibre waits for: the n_words words at `words` all carry `tag` in their upper 16 bits (the tagged response
// line of the LS-step server, csrc/nq_server.hip).  words == nullptr: unknown -- the fibre is resumed to look for itself.
struct FiberWait {
    const volatile uint32_t *words = nullptr;
    int n_words = 0;
    uint32_t tag = 0;
    bool ready() const
    {
        for (int i = 0; i < n_words; i++)
            if ((words[i] >> 16) != tag) return false;
        return true;
    }
};

class FiberPool {
public:
    explicit FiberPool(int n_workers, size_t stack_bytes = (size_t)8 << 20) : stack_bytes_((stack_bytes + 4095) & ~(size_t)4095)
    {
        if (n_workers < 1) n_workers = 1;
        for (int i = 0; i < n_workers; i++) workers_.emplace_back(new Worker());
    }
    ~FiberPool()
    {
        for (auto &f : fibers_)
            if (f->stack_base) munmap(f->stack_base, f->stack_len);
    }
    FiberPool(const FiberPool &) = delete;
    FiberPool &operator=(const FiberPool &) = delete;

    // before run(): fibre `fn`, first run by preference on worker `home`
    void add(std::function<void()> fn, int home)
    {
        std::unique_ptr<Fiber> f(new Fiber());
        f->fn = std::move(fn);
        f->stack_len = stack_bytes_ + 4096;
        void *p = mmap(nullptr, f->stack_len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
        if (p == MAP_FAILED) throw std::bad_alloc();
        f->stack_base = static_cast<char *>(p);
        (void)mprotect(f->stack_base, 4096, PROT_NONE); // guard page: the stack grows down towards it
        prepare(*f);
        f->last_worker = home % (int)workers_.size();
        workers_[(size_t)f->last_worker]->mine.push_back(f.get());
        workers_[(size_t)f->last_worker]->n_mine.store((int)workers_[(size_t)f->last_worker]->mine.size(), std::memory_order_relaxed);
        fibers_.push_back(std::move(f));
    }

    // runs every fibre to completion on n_workers threads (the calling thread is worker 0)
    void run()
    {
        live_.store((int)fibers_.size(), std::memory_order_release);
        std::vector<std::thread> threads;
        for (size_t i = 1; i < workers_.size(); i++) threads.emplace_back([this, i]() { worker_loop((int)i); });
        worker_loop(0);
        for (auto &t : threads) t.join();
    }

    // From inside a fibre: give the thread up; some worker resumes the caller once `wait` (if given) has arrived and no
    // ready fibre of higher priority wants a core.
    static void yield(const FiberWait *wait = nullptr)
    {
        Worker *w = current_worker();
        if (!w || !w->current) return; // not on a fibre
        Fiber *f = w->current;
        f->wait = wait ? *wait : FiberWait();
        switch_to(f->regs, w->main);
    }
    static void yield_hook(void *) { yield(); }
    // of the calling fibre: larger runs first
    static void set_priority(uint64_t p)
    {
        Worker *w = current_worker();
        if (w && w->current) w->current->priority.store(p, std::memory_order_relaxed);
    }
    uint64_t steals() const { return steals_.load(std::memory_order_relaxed); } // fibres an idle worker took over

private:
    enum : int { SUSPENDED = 0, RUNNING = 1, DONE = 2 };
    enum : unsigned { STALL_SPINS = 256 }; // scans of a worker's fibres with none ready (~20 us) before one is resumed anyway
    struct alignas(64) Fiber {
        std::atomic<int> state{SUSPENDED};
        std::atomic<uint64_t> priority{0};
        FiberWait wait;              // written by the fibre before it yields, read by workers while it is SUSPENDED
        int last_worker = 0;
        uint64_t last_forced = 0;    // when it was last resumed without its answer (the worker's count)
        std::function<void()> fn;
        char *stack_base = nullptr;
        size_t stack_len = 0;
        bool done = false;
#if defined(__x86_64__)
        FiberRegs regs{nullptr};
#else
        ucontext_t regs;
#endif
    };
    struct alignas(64) Worker {
        std::atomic_flag lock = ATOMIC_FLAG_INIT;
        std::vector<Fiber *> mine;   // this worker's fibres that are not finished (running or suspended); guarded by `lock`
        std::atomic<int> n_mine{0};  // mine.size(), readable without the lock (thieves look for the fullest worker)
        void acquire() { while (lock.test_and_set(std::memory_order_acquire)) { fiber_cpu_relax(); } }
        void release() { lock.clear(std::memory_order_release); }
        Fiber *current = nullptr;    // the fibre this worker is running
        uint64_t forced = 0;         // fibres resumed without their answer
#if defined(__x86_64__)
        FiberRegs main{nullptr};
#else
        ucontext_t main;
#endif
    };

    // the worker the calling THREAD is; never cached across a switch (a fibre resumes on whichever thread took it)
    static Worker *&tl_slot()
    {
        static thread_local Worker *w = nullptr;
        return w;
    }
    __attribute__((noinline)) static Worker *current_worker()
    {
        asm volatile("" ::: "memory");
        return tl_slot();
    }
    __attribute__((noinline)) static void set_current_worker(Worker *w)
    {
        tl_slot() = w;
        asm volatile("" ::: "memory");
    }

#if defined(__x86_64__)
    static void switch_to(FiberRegs &from, FiberRegs &to) { ls_fiber_switch(&from, &to); }
    void prepare(Fiber &f)
    {
        // initial frame as ls_fiber_switch expects it: [mxcsr|fpucw][r15][r14][r13][r12][rbx][rbp][return address = entry][0]
        uintptr_t top = reinterpret_cast<uintptr_t>(f.stack_base + f.stack_len) & ~(uintptr_t)15;
        uint64_t *sp = reinterpret_cast<uint64_t *>(top);
        *--sp = 0;                                              // fake return address of entry(): never used
        *--sp = reinterpret_cast<uint64_t>(&FiberPool::entry);  // `ret` of the first switch lands here, rsp = 8 (mod 16) as after a call
        for (int i = 0; i < 6; i++) *--sp = 0;                  // rbp rbx r12 r13 r14 r15
        uint32_t csr[2];
        asm volatile("stmxcsr %0" : "=m"(csr[0]));
        uint16_t cw;
        asm volatile("fnstcw %0" : "=m"(cw));
        csr[1] = cw;
        uint64_t word;
        memcpy(&word, csr, 8);
        *--sp = word;
        f.regs.rsp = sp;
    }
#else
    static void switch_to(ucontext_t &from, ucontext_t &to) { swapcontext(&from, &to); }
    void prepare(Fiber &f)
    {
        getcontext(&f.regs);
        f.regs.uc_stack.ss_sp = f.stack_base + 4096;
        f.regs.uc_stack.ss_size = f.stack_len - 4096;
        f.regs.uc_link = nullptr;
        makecontext(&f.regs, reinterpret_cast<void (*)()>(&FiberPool::entry), 0);
    }
#endif

    static void entry()
    {
        {
            Worker *w = current_worker();
            Fiber *f = w->current;
            f->fn(); // catches everything itself
            f->done = true;
        }
        Worker *w = current_worker(); // possibly another thread than the one that started the fibre
        switch_to(w->current->regs, w->main);
        __builtin_trap(); // a finished fibre is never resumed
    }

    // The fibre this worker runs next: one of its own, suspended, whose awaited answer has arrived; highest priority first.
    // any_one: the worker has had nothing ready for a while -- the suspended fibre that has waited longest is resumed to look
    // for itself (an answer may never arrive unless the waiter acts: SrvHost::wait launches a drained server again).
    Fiber *pick_mine(Worker &w, bool any_one = false)
    {
        Fiber *best = nullptr;
        uint64_t best_p = 0;
        w.acquire();
        for (Fiber *f : w.mine) {
            if (f->state.load(std::memory_order_acquire) != SUSPENDED) continue;
            if (any_one) {
                if (!best || f->last_forced < best->last_forced) best = f;
                continue;
            }
            if (f->wait.words && !f->wait.ready()) continue;
            const uint64_t p = f->priority.load(std::memory_order_relaxed);
            if (!best || p > best_p) { best = f; best_p = p; }
        }
        if (best && any_one) best->last_forced = ++w.forced;
        if (best) best->state.store(RUNNING, std::memory_order_relaxed); // under the lock: a thief cannot take it any more
        w.release();
        return best;
    }
    // Nothing left here: the highest-priority suspended fibre of the worker that holds most (it is the one that would wait
    // longest for its turn there).  A worker's last fibre is left alone.
    Fiber *steal(int self)
    {
        const int n = (int)workers_.size();
        int victim = -1, most = 1;
        for (int i = 1; i < n; i++) {
            const int v = (self + i) % n;
            const int c = workers_[(size_t)v]->n_mine.load(std::memory_order_relaxed);
            if (c > most) { most = c; victim = v; }
        }
        if (victim < 0) return nullptr;
        Worker &v = *workers_[(size_t)victim];
        Fiber *best = nullptr;
        uint64_t best_p = 0;
        size_t at = 0;
        v.acquire();
        if (v.mine.size() >= 2)
            for (size_t i = 0; i < v.mine.size(); i++) {
                Fiber *f = v.mine[i];
                if (f->state.load(std::memory_order_acquire) != SUSPENDED) continue;
                const uint64_t p = f->priority.load(std::memory_order_relaxed);
                if (!best || p > best_p) { best = f; best_p = p; at = i; }
            }
        if (best) {
            v.mine.erase(v.mine.begin() + (long)at);
            v.n_mine.store((int)v.mine.size(), std::memory_order_relaxed);
        }
        v.release();
        if (!best) return nullptr;
        best->last_forced = 0;
        Worker &w = *workers_[(size_t)self];
        w.acquire();
        w.mine.push_back(best);
        w.n_mine.store((int)w.mine.size(), std::memory_order_relaxed);
        w.release();
        best->last_worker = self;
        steals_.fetch_add(1, std::memory_order_relaxed);
        return nullptr; // it is picked like any other of this worker's fibres (once its answer is there)
    }
    void worker_loop(int self)
    {
        Worker &w = *workers_[(size_t)self];
        set_current_worker(&w);
        unsigned idle = 0, stalled = 0;
        for (;;) {
            Fiber *f = pick_mine(w);
            if (!f && w.n_mine.load(std::memory_order_relaxed) != 0 && ++stalled >= STALL_SPINS) f = pick_mine(w, true);
            if (!f) {
                if (w.n_mine.load(std::memory_order_relaxed) == 0) {
                    if (live_.load(std::memory_order_acquire) == 0) break;
                    (void)steal(self);
                    if (w.n_mine.load(std::memory_order_relaxed) == 0) { // nothing to take: the others run one fibre each
                        // back off: spin, then yield, then sleep -- at the tail of a run most workers are here while the longest
                        // restarts finish on the others, and a spinning hyper-thread takes issue slots from its busy sibling
                        if (++idle <= 64) { for (int k = 0; k < 32; k++) fiber_cpu_relax(); }
                        else if (idle <= 64 + 16) sched_yield();
                        else {
                            struct timespec ts = {0, 20000}; // 20 us
                            nanosleep(&ts, nullptr);
                        }
                    } else idle = 0;
                } else {
                    fiber_cpu_relax(); // all of this worker's fibres wait for the GPU
                }
                continue;
            }
            idle = 0; stalled = 0;
            w.current = f;
            switch_to(w.main, f->regs);
            w.current = nullptr;
            if (f->done) {
                w.acquire();
                for (size_t i = 0; i < w.mine.size(); i++)
                    if (w.mine[i] == f) { w.mine.erase(w.mine.begin() + (long)i); break; }
                w.n_mine.store((int)w.mine.size(), std::memory_order_relaxed);
                w.release();
                f->state.store(DONE, std::memory_order_release);
                live_.fetch_sub(1, std::memory_order_acq_rel);
            } else {
                f->state.store(SUSPENDED, std::memory_order_release); // only now: its registers are saved
            }
        }
        set_current_worker(nullptr);
    }

    size_t stack_bytes_;
    std::vector<std::unique_ptr<Worker>> workers_;
    std::vector<std::unique_ptr<Fiber>> fibers_;
    std::atomic<int> live_{0};
    std::atomic<uint64_t> steals_{0};
};

} // namespace lshost
