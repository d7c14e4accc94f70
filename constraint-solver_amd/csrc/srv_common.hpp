// srv_common.hpp — what the resident servers (nq_server.hip: N-queens LS step; sched_kernels.hip: scheduling LS step) share:
// the mailbox protocol between a host thread and a resident workgroup, the rules by which a resident launch leaves, and the
// host object that owns the launch.
//
// Mailbox (no ordering assumption on the PCIe reads of one line, no atomics over PCIe): a line is 16 dwords in coherent
// pinned host memory, every dword (tag16 << 16 | payload16).  The writer gives all dwords of one message the same new tag;
// the reader takes a snapshot with one dword per lane and accepts it only if all tags are equal (a torn snapshot mixes two
// tags and is polled again).  Requests go host -> workgroup (16 dwords), answers workgroup -> host (8 dwords).  Bulk data
// lives elsewhere in the slot's pinned block and is written before the line that announces it.  Payload dword 0 of a
// request carries the flags; bit 2 (SRV_F_QUIT) asks the workgroup to leave.
//
// Leaving: a resident launch must never outlive its use.  Every workgroup leaves (a) on a QUIT request, (b) when the
// launch's quit word is raised, which any workgroup does once NO slot has been served for `idle_ticks` (default 2 ms,
// LS_NQ_SERVER_IDLE_US) or the launch is older than `limit_ticks` (2 s, LS_NQ_SERVER_LIMIT_MS) -- a dead or paused host
// cannot leave a kernel spinning -- and the last one out publishes the launch generation to a host word.  The host starts
// the launch again on demand; a request posted while it drains is found by the next launch (the last SERVED tag is read
// back from the answer line).
//
// The launch lives on its own hardware queue (a CU-masked stream; fallback: high priority): HIP multiplexes ordinary
// streams onto a few queues and a queue runs its packets in order, so a resident kernel on an ordinary stream blocks
// whatever stream shares its queue (tools/server_probe.hip: the 6th stream created afterwards waited 0.46 s; hipHostFree
// waits for resident kernels whatever the stream: the servers are asked to leave before their memory is released).
#pragma once

#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <sched.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#include "ls_common.hpp"

namespace lsamd {

constexpr unsigned SRV_F_QUIT = 4u; // request payload dword 0, bit 2 (the other bits belong to the server)

struct SrvCtl {                       // device memory, one per server
    unsigned long long quit_gen;      // == the running launch's generation: everybody leaves
    unsigned long long last_activity; // wall clock (100 MHz) of the last served request of any slot
    unsigned exited;                  // workgroups of the running launch that have left
    unsigned pad;
};

struct SrvLink {                      // kernel argument: the lines of slot 0 (slot s: + 16 s dwords) and the launch's rules
    const unsigned *req;
    unsigned *resp;
    SrvCtl *ctl;
    unsigned long long *exit_gen;     // pinned host
    unsigned long long gen, idle_ticks, limit_ticks;
};

#if defined(__HIPCC__)
template <class T> __device__ __forceinline__ T srv_sys_load(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// 16 bytes from host (or fine-grained device) memory past every cache: bulk data a workgroup fetches once (a slot's move stream)
typedef unsigned srv_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ srv_u32x4 srv_sys_load_b128(const void *p)
{
    srv_u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// A workgroup of 512 threads fetches one 512-move chunk (12 bytes a move: 384 16-byte reads instead of 1536 4-byte ones over PCIe)
// through `buf` (1536 dwords of LDS); afterwards thread t finds its move in buf[3 t .. 3 t + 2].  Block-uniform call.
__device__ __forceinline__ void srv_fetch_moves_512(const void *chunk, unsigned *buf, int tid)
{
    __syncthreads(); // nobody still reads the previous chunk
    if (tid < 384) reinterpret_cast<srv_u32x4 *>(buf)[tid] = srv_sys_load_b128(static_cast<const char *>(chunk) + (size_t)tid * 16);
    __syncthreads();
}

__device__ __forceinline__ unsigned long long srv_uniform_u64(unsigned long long x) // a wave-uniform value, into scalar registers
{
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x);
}

struct SrvPoll { // wave 0's poll state
    unsigned last_tag;
    unsigned long long t_start, t_served;
#ifdef LS_NQ_STAMPS
    unsigned n_polls = 0; // polls the last accepted request took (stamps builds: the poll period = idle gap / polls)
#endif
};

// wave 0, once per launch: the last request an earlier launch answered is read back from the answer line
__device__ __forceinline__ void srv_poll_begin(SrvPoll &p, const SrvLink &L, const unsigned *resp_line, int lane)
{
    p.t_start = p.t_served = wall_clock64();
    p.last_tag = __builtin_amdgcn_readfirstlane(srv_sys_load(resp_line) >> 16);
    if (lane == 0) atomicMax(&L.ctl->last_activity, p.t_start); // nobody reads a stale activity stamp of an earlier launch
}

// wave 0: wait for the slot's next request -- a message of ND dwords (one line: 16; up to four consecutive lines: 64).
// Returns 1 when the workgroup has to leave, else 0 with the ND payload halfwords in s_req (LDS) and, in `raw`, this lane's
// tagged dword as it was read.
template <int ND = 16>
__device__ __forceinline__ int srv_poll_wait(SrvPoll &p, const SrvLink &L, const unsigned *req_line, unsigned *s_req, int lane, unsigned *raw = nullptr)
{
    static_assert(ND == 16 || ND == 64, "one line or four");
    for (unsigned polls = 1;; polls++) {
        // ONE load in flight per poll, issued without a branch around it (every lane reads a dword of the message: lanes beyond it
        // read it again, the same line); the launch's quit word -- device memory, a round trip of its own -- only every fourth
        // poll, issued beside the line's read
        const unsigned v = srv_sys_load(req_line + (lane & (ND - 1)));
        unsigned long long q = 0;
        if ((polls & 3u) == 0) q = __hip_atomic_load(&L.ctl->quit_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (uniform address, uniform branch)
        const unsigned tag = v >> 16;
        const unsigned t0 = __builtin_amdgcn_readfirstlane(tag);
        const bool whole = __builtin_amdgcn_ballot_w64(tag != t0) == 0;
        if (whole && t0 != p.last_tag && t0 != 0) {
            if (lane < ND) s_req[lane] = v & 0xffffu;
            if (raw) *raw = v;
#ifdef LS_NQ_STAMPS
            p.n_polls = polls;
#endif
            p.last_tag = t0;
            return (__builtin_amdgcn_readfirstlane(v) & SRV_F_QUIT) ? 1 : 0;
        }
        if ((polls & 3u) == 0) {
            const unsigned qlo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)q);
            const unsigned qhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(q >> 32));
            if ((((unsigned long long)qhi << 32) | qlo) == L.gen) return 1;
        }
        const unsigned long long now = wall_clock64();
        if ((polls & 15u) == 0) { // nobody served for idle_ticks, or the launch has reached its age limit: everybody leaves
            unsigned long long la = 0;
            if (lane == 0) la = __hip_atomic_load(&L.ctl->last_activity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned llo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)la);
            const unsigned lhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(la >> 32));
            la = ((unsigned long long)lhi << 32) | llo;
            if ((now > la && now - la > L.idle_ticks) || now - srv_uniform_u64(p.t_start) > L.limit_ticks) {
                if (lane == 0) __hip_atomic_store(&L.ctl->quit_gen, L.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return 1;
            }
        }
        // a slot in the middle of a walk is asked again within microseconds; one that has been quiet for longer polls less often.
        // The quiet time as a SCALAR: with the stamp in vector registers the compiler turned the three-way choice into three
        // exec-masked regions without branches, and s_sleep ignores the exec mask -- every poll slept 1 + 8 + 64 units, 1.9 us
        // (one poll per 2.7 us, measured by the stamps build's poll counter, where the PCIe read takes 1.15 us; now 1.5 us per poll:
        // n = 4096 11.2 -> 10.3 us per LS iteration, n = 1024 6.7 -> 5.6, config 5 8.5 -> 8.0, same box).
        const unsigned long long quiet = now - srv_uniform_u64(p.t_served);
        if (quiet > 20000) __builtin_amdgcn_s_sleep(64);      // > 200 us quiet: ~2 us between polls
        else if (quiet > 2000) __builtin_amdgcn_s_sleep(8);   // > 20 us quiet
        else __builtin_amdgcn_s_sleep(1);
    }
}

// wave 0, after the answer line has been written
__device__ __forceinline__ void srv_served(SrvPoll &p, const SrvLink &L, int lane)
{
    p.t_served = wall_clock64();
    if (lane == 0) __hip_atomic_store(&L.ctl->last_activity, p.t_served, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// thread 0 of a workgroup that leaves: the last one out tells the host that this launch has drained
__device__ __forceinline__ void srv_block_leaves(const SrvLink &L)
{
    if (__hip_atomic_fetch_add(&L.ctl->exited, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        __hip_atomic_store(&L.ctl->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(L.exit_gen, L.gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#endif // __HIPCC__

// ---------------------------------------------------------------------------------------------------------------- host
// One 64-byte line of the mailbox, written by the host.  The device polls these lines over PCIe, so every separate store to a
// line it has just read costs the core a fresh ownership request: one 64-byte store per line where the CPU has one (the
// dwords' tags still decide what the reader accepts -- nothing here relies on the store being indivisible).
#if defined(__x86_64__)
__attribute__((target("avx512f"))) inline void srv_store_line_wide(volatile unsigned *dst, const unsigned *line)
{
    _mm512_store_si512(const_cast<unsigned *>(dst), _mm512_loadu_si512(line));
}
#endif
inline void srv_store_line(volatile unsigned *dst, const unsigned *line)
{
#if defined(__x86_64__)
#if !defined(__SANITIZE_THREAD__) // (ThreadSanitizer sees the dword-wise form below: every access of a line is an atomic one there)
    static const bool wide = __builtin_cpu_supports("avx512f") && getenv("LS_HOST_NO_AVX512") == nullptr;
    if (wide) { srv_store_line_wide(dst, line); return; }
#endif
#endif
    for (int d = 0; d < 16; d++) __atomic_store_n(const_cast<unsigned *>(dst) + d, line[d], __ATOMIC_RELAXED);
}

struct SrvHost {
    int device = 0, n_slots = 0;  // n_slots: mailbox lines (a reader owns `lines_per_msg` consecutive ones)
    int lines_per_msg = 1;
    const char *what = "server";
    hipStream_t stream = nullptr;
    char *pinned = nullptr;       // request lines | answer lines | exit word | the owner's per-server block
    size_t o_resp = 0, o_exit = 0, o_extra = 0;
    SrvCtl *ctl = nullptr;
    unsigned long long idle_ticks = 200000, limit_ticks = 200000000ull; // 2 ms, 2 s (100 MHz)
    std::mutex launch_mu;
    // Many host threads use one SrvHost at once, a slot each (ls_multi_solver): what a slot's thread writes on every request -- its tag,
    // its request count -- has a cache line of its own, and the launch generation every wait() reads is on a line nobody writes in
    // the steady state.  (Tags of 16 slots in one line and one shared request counter cost a 16-thread run 2.7 us per call.)
    struct alignas(64) PerSlot {
        unsigned tag = 0;                 // last tag posted
        unsigned long long requests = 0;  // requests posted (relaxed atomic accesses)
    };
    std::vector<PerSlot> per;
    alignas(64) std::atomic<unsigned long long> launch_gen{0};
    alignas(64) std::atomic<unsigned long long> relaunches{0};
    std::atomic<unsigned long long> slow_waits{0};
    unsigned long long requests_total() const
    {
        unsigned long long n = 0;
        for (const PerSlot &p : per) n += __atomic_load_n(&p.requests, __ATOMIC_RELAXED);
        return n;
    }
    bool failed = false;
    unsigned relaunch_limit = 0;  // wait(): give up (SRV_RC_GIVE_UP) after this many relaunches under one request; 0: never
    double no_answer_seconds = 10.0; // wait(): an error after this long without an answer from a launch that has not drained
    std::function<hipError_t(const SrvLink &)> launch; // set by the owner: starts the resident kernel on `stream`

    // Request lines: pinned host memory, polled by the device over PCIe -- or, when the owner sets req_dev, fine-grained DEVICE
    // memory the host writes through the PCIe BAR (write-combined: 64-byte stores, then sfence): a poll is then a local read, a
    // request crosses PCIe once instead of being fetched, and every workgroup of a grid can poll it without a relay.  (Used by the
    // scheduling grid.  For the one-line mailboxes of the step servers it was measured three times, rounds 3 and 4 -- the last time
    // after the poll loop's sleep bug was fixed, same box: config 3 5.0 against 4.6 us per LS iteration, N-queens n = 4096 10.15
    // against 10.2-10.4, n = 1024 5.4-5.7 against 5.6-5.8, config 4 end to end 0.84 against 0.78 s -- no gain: the write-combined
    // store + sfence costs the host what the local poll saves, and a request posted right after an answer is caught by the
    // pinned line's first poll anyway.)
    unsigned *req_dev = nullptr;
    unsigned *req(int s) const { return (req_dev ? req_dev : reinterpret_cast<unsigned *>(pinned)) + (size_t)s * 16; }
    void flush_lines() const
    {
#if defined(__x86_64__)
        if (req_dev) _mm_sfence();
#endif
    }
    unsigned *resp(int s) const { return reinterpret_cast<unsigned *>(pinned + o_resp) + (size_t)s * 16; }
    unsigned long long *exit_gen() const { return reinterpret_cast<unsigned long long *>(pinned + o_exit); }
    char *extra() const { return pinned + o_extra; }

    // the stream on a hardware queue of its own, the pinned block (lines + `extra_bytes` for the owner), the control block
    int init(ls_ctx *ctx, const char *what_, int slots, size_t extra_bytes)
    {
        what = what_;
        device = ctx->device;
        n_slots = slots;
        if (const char *v = getenv("LS_NQ_SERVER_IDLE_US")) { const long long us = atoll(v); if (us >= 50) idle_ticks = (unsigned long long)us * 100; }
        if (const char *v = getenv("LS_NQ_SERVER_LIMIT_MS")) { const long long ms = atoll(v); if (ms >= 1) limit_ticks = (unsigned long long)ms * 100000; }
        hipError_t e;
        {   // CU-masked streams are never pooled; fallback: the high-priority pool
            std::vector<uint32_t> mask(((size_t)ctx->n_cus + 31) / 32, 0xffffffffu);
            if (ctx->n_cus % 32) mask.back() = (1u << (ctx->n_cus % 32)) - 1u;
            e = hipExtStreamCreateWithCUMask(&stream, (uint32_t)mask.size(), mask.data());
            if (e != hipSuccess) {
                (void)hipGetLastError();
                int lo = 0, hi = 0;
                (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
                e = hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, hi);
            }
            if (e != hipSuccess) { stream = nullptr; return set_err(ctx, LS_ERR_HIP, "%s: stream creation failed: %s", what, hipGetErrorString(e)); }
        }
        size_t off = (size_t)slots * 64;
        o_resp = off; off += (size_t)slots * 64;
        o_exit = off; off += 64;
        o_extra = off; off += (extra_bytes + 63) & ~(size_t)63;
        e = hipHostMalloc(reinterpret_cast<void **>(&pinned), off, hipHostMallocCoherent);
        if (e != hipSuccess) { pinned = nullptr; release(); return set_err(ctx, LS_ERR_NOMEM, "%s: pinned allocation failed: %s", what, hipGetErrorString(e)); }
        memset(pinned, 0, off);
        e = hipMalloc(reinterpret_cast<void **>(&ctl), 256);
        if (e != hipSuccess) { ctl = nullptr; release(); return set_err(ctx, LS_ERR_NOMEM, "%s: device allocation failed: %s", what, hipGetErrorString(e)); }
        hipLaunchKernelGGL(fill_i64_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<long long *>(ctl), 0ll, 32ll);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(stream)) != hipSuccess) {
            release();
            return set_err(ctx, LS_ERR_HIP, "%s: control block set-up failed: %s", what, hipGetErrorString(e));
        }
        per.assign((size_t)slots, PerSlot());
        return LS_OK;
    }
    void release()
    {
        req_dev = nullptr; // (owned by whoever set it)
        if (stream) (void)hipStreamDestroy(stream);
        if (pinned) (void)hipHostFree(pinned);
        if (ctl) (void)hipFree(ctl);
        stream = nullptr; pinned = nullptr; ctl = nullptr;
    }
    bool drained() const { return __atomic_load_n(exit_gen(), __ATOMIC_ACQUIRE) == launch_gen.load(std::memory_order_acquire); }

    // (Re)launch when the previous launch has drained (or none has been made).  Cheap when the kernel is resident.
    int ensure_running(ls_ctx *ctx)
    {
        if (!drained()) return LS_OK;
        std::lock_guard<std::mutex> lk(launch_mu);
        if (failed) return set_err(ctx, LS_ERR_HIP, "%s: an earlier launch failed", what);
        if (!drained()) return LS_OK;
        hipError_t e = hipSetDevice(device); // the steady-state path makes no runtime call: this thread may never have bound the device
        if (e != hipSuccess) return set_err(ctx, LS_ERR_HIP, "hipSetDevice failed: %s", hipGetErrorString(e));
        SrvLink L;
        L.req = req(0); L.resp = resp(0); L.ctl = ctl; L.exit_gen = exit_gen();
        L.gen = launch_gen.load() + 1; L.idle_ticks = idle_ticks; L.limit_ticks = limit_ticks;
        e = launch(L);
        if (e != hipSuccess) {
            failed = true;
            return set_err(ctx, LS_ERR_HIP, "%s: launch failed: %s", what, hipGetErrorString(e));
        }
        launch_gen.store(L.gen, std::memory_order_release);
        relaunches.fetch_add(1, std::memory_order_relaxed);
        return LS_OK;
    }
    unsigned next_tag(int slot)
    {
        PerSlot &p = per[(size_t)slot];
        p.tag = p.tag >= 0xfffeu ? 1u : p.tag + 1u;
        __atomic_store_n(&p.requests, p.requests + 1, __ATOMIC_RELAXED); // (one writer: the slot's thread)
        return p.tag;
    }
    void post(int slot, const unsigned pay[16])
    {
        const unsigned tag = next_tag(slot);
        alignas(64) unsigned line[16];
        for (int d = 0; d < 16; d++) line[d] = (tag << 16) | (pay[d] & 0xffffu);
        __atomic_thread_fence(__ATOMIC_RELEASE); // the bulk data is written before the line that announces it
        srv_store_line(req(slot), line);
        flush_lines();
    }
    // A message of n_lines consecutive lines starting at line `slot` (the scheduling grid: header + state), every dword under
    // ONE new tag; the reader accepts a snapshot only when all tags agree, so the order of the line stores does not matter.
    void post_lines(int slot, const unsigned *pay, int n_lines)
    {
        const unsigned tag = next_tag(slot);
        alignas(64) unsigned msg[64];
        for (int d = 0; d < 16 * n_lines; d++) msg[d] = (tag << 16) | (pay[d] & 0xffffu);
        __atomic_thread_fence(__ATOMIC_RELEASE);
        for (int l = n_lines - 1; l >= 0; l--) srv_store_line(req(slot + l), msg + 16 * l);
        flush_lines();
    }
    // QUIT to every reader (one per `lines_per_msg` lines), as a WHOLE message of that reader's size under one new tag
    void post_quit_all()
    {
        unsigned pay[64] = {0};
        pay[0] = SRV_F_QUIT;
        for (int i = 0; i + lines_per_msg <= n_slots; i += lines_per_msg) {
            if (lines_per_msg == 1) post(i, pay);
            else post_lines(i, pay, lines_per_msg);
            __atomic_store_n(&per[(size_t)i].requests, per[(size_t)i].requests - 1, __ATOMIC_RELAXED); // not a request of a caller
        }
    }
    // the launch leaves now (its arguments are about to change, or memory is about to be released) and comes back on demand
    void pause()
    {
        if (!stream) return;
        (void)hipSetDevice(device);
        const bool asked = pinned && !drained();
        if (asked) post_quit_all();
        (void)hipStreamSynchronize(stream);
        // The QUIT messages stay in the request lines; a reader of the NEXT launch takes "the last request served" from its
        // answer line (srv_poll_begin), so the QUIT is marked served there -- else that reader would leave the moment it starts
        // and its slot would have nobody until the whole launch has drained.
        if (asked)
            for (int i = 0; i + lines_per_msg <= n_slots; i += lines_per_msg)
                __atomic_store_n(resp(i), per[(size_t)i].tag << 16, __ATOMIC_RELAXED);
    }
    // Poll the slot's answer line for the tag of the request just posted (the same hooks as the completion-word wait of a
    // launched step: the caller's idle work, another restart's fibre, sched_yield).
    int wait(ls_ctx *ctx, int slot, unsigned out[8])
    {
        const unsigned tag = per[(size_t)slot].tag;
        const volatile unsigned *r = resp(slot);
        struct Announce { // what this wait waits for, for a scheduler behind wait_hook
            ls_ctx *c;
            Announce(ls_ctx *c_, const volatile unsigned *w, unsigned t) : c(c_) { c->wait_words = w; c->n_wait_words = 8; c->wait_tag = t; }
            ~Announce() { c->wait_words = nullptr; c->n_wait_words = 0; }
        } announce(ctx, r, tag);
        bool idle_work = ctx->idle_hook != nullptr;
        std::chrono::steady_clock::time_point t0;
        bool timed = false;
        unsigned relaunched_here = 0;
        for (unsigned spins = 0;; spins++) {
            unsigned v[8];
            bool ok = true;
            for (int d = 0; d < 8; d++) { v[d] = __atomic_load_n(const_cast<const unsigned *>(r) + d, __ATOMIC_RELAXED); ok = ok && (v[d] >> 16) == tag; }
            if (ok) {
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                for (int d = 0; d < 8; d++) out[d] = v[d] & 0xffffu;
                return LS_OK;
            }
            if (idle_work) { idle_work = ctx->idle_hook(ctx->idle_arg); continue; }
            if (ctx->wait_hook) ctx->wait_hook(ctx->wait_hook_arg);
#if defined(__x86_64__)
            else __builtin_ia32_pause();
#endif
            if ((spins & 15u) == 15u || ctx->wait_hook) { // behind a hook every return here may have been a long absence
                // has the launch drained under the request (idle time-out, age limit)?  Then it is launched again and finds the request.
                if (drained()) {
                    slow_waits.fetch_add(1, std::memory_order_relaxed);
                    // a launch whose workgroups wait for each other (the scheduling grid) leaves unanswered when they cannot
                    // all be resident -- another context's grid holds the CUs: the owner then takes another route
                    if (relaunch_limit && ++relaunched_here > relaunch_limit) return SRV_RC_GIVE_UP;
                    const int rc = ensure_running(ctx);
                    if (rc != LS_OK) return rc;
                }
                if ((spins & 0x3ffu) == 0x3ffu) {
                    if (!timed) { t0 = std::chrono::steady_clock::now(); timed = true; }
                    else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > no_answer_seconds) {
                        const hipError_t e = hipStreamQuery(stream);
                        return set_err(ctx, LS_ERR_HIP, "%s: no answer from slot %d within %.1f s (stream: %s)", what, slot, no_answer_seconds, hipGetErrorString(e));
                    }
                }
            }
        }
    }
    // ask every workgroup to leave, wait for the launch, release everything
    void shutdown()
    {
        if (!stream) return;
        pause();
        release();
    }
};

} // namespace lsamd
