// CPU-side unit test of the resident servers' HOST state machine (csrc/srv_common.hpp: SrvHost -- post / wait / relaunch /
// give-up / pause / shutdown), no GPU and no HIP: the product header is compiled against tests/cpp/fake_hip/hip/hip_runtime.h
// and the "resident kernel" is a host thread that speaks the mailbox protocol (tagged dwords, torn-snapshot rule, QUIT, idle
// time-out, age limit, last-one-out publishes the launch generation) and can be told to answer, to drain without answering, or
// to stay resident and silent.  Run under AddressSanitizer and ThreadSanitizer by tests/test_host_batcher.py.
// Cases: many slots on many threads across dozens of relaunches; the 16-bit tag wrap (> 65 534 requests on one slot); a
// resident launch that never answers (the no-answer error); launches that leave unanswered (SRV_RC_GIVE_UP after
// relaunch_limit); a failing launch; multi-line messages with QUIT as a whole message; pause() of a busy launch.
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "../../constraint-solver_amd/csrc/srv_common.hpp"

namespace lsamd {
int set_err(const ls_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}
} // namespace lsamd
using namespace lsamd;

#define CHECK(cond)                                                                        \
    do {                                                                                   \
        if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); exit(1); } \
    } while (0)

static unsigned long long ticks_now() // the device's 100 MHz wall clock
{
    return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10;
}
static unsigned ld(const unsigned *p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
static void st(unsigned *p, unsigned v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }

enum Mode { ANSWER, LEAVE_AT_ONCE, SILENT };

// one fake resident launch: `readers` mailbox readers, reader i owns lines [i * lpm, (i + 1) * lpm) and answers on line i * lpm
static void fake_kernel(SrvLink L, int readers, int lpm, std::atomic<int> *mode)
{
    const unsigned long long t_start = ticks_now();
    unsigned long long last_activity = t_start;
    std::vector<unsigned> last_tag((size_t)readers);
    for (int i = 0; i < readers; i++) last_tag[(size_t)i] = ld(L.resp + (size_t)i * lpm * 16) >> 16; // srv_poll_begin
    bool leave = mode->load() == LEAVE_AT_ONCE;
    while (!leave) {
        for (int i = 0; i < readers && !leave; i++) {
            const unsigned *line = L.req + (size_t)i * lpm * 16;
            unsigned v[64];
            bool whole = true;
            for (int d = 0; d < 16 * lpm; d++) { v[d] = ld(line + d); whole = whole && (v[d] >> 16) == (v[0] >> 16); }
            const unsigned tag = v[0] >> 16;
            if (!whole || tag == 0 || tag == last_tag[(size_t)i]) continue;
            last_tag[(size_t)i] = tag;
            if (v[0] & SRV_F_QUIT) { leave = true; break; }
            if (mode->load() == SILENT) continue;
            unsigned sum = 0;
            for (int d = 1; d < 16 * lpm; d++) sum += v[d] & 0xffffu;
            unsigned *out = L.resp + (size_t)i * lpm * 16;
            for (int d = 7; d >= 0; d--) st(out + d, (tag << 16) | ((d == 0 ? (unsigned)i : d == 1 ? sum : (v[1] & 0xffffu) + (unsigned)d) & 0xffffu)); // last dword first: a torn answer must not pass
            last_activity = ticks_now();
        }
        const unsigned long long now = ticks_now();
        if (__atomic_load_n(&L.ctl->quit_gen, __ATOMIC_RELAXED) == L.gen) leave = true;
        if (now - last_activity > L.idle_ticks || now - t_start > L.limit_ticks) leave = true;
    }
    __atomic_store_n(L.exit_gen, L.gen, __ATOMIC_RELEASE); // srv_block_leaves: the last one out
}

struct Harness {
    ls_ctx ctx;
    SrvHost h;
    std::atomic<int> mode{ANSWER};
    std::atomic<int> fail_launch{0};
    int readers = 0, lpm = 1;
    void start(int readers_, int lpm_, unsigned long long idle_us)
    {
        readers = readers_; lpm = lpm_;
        ctx.device = 0; ctx.n_cus = 256;
        CHECK(h.init(&ctx, "fake server", readers * lpm, 0) == LS_OK);
        h.lines_per_msg = lpm;
        h.idle_ticks = idle_us * 100;
        h.launch = [this](const SrvLink &L) -> hipError_t {
            if (fail_launch.load()) return hipErrorLaunchFailure;
            (void)hipStreamSynchronize(h.stream); // a launch is only asked for once the previous one has drained: its thread has ended (or is ending)
            std::lock_guard<std::mutex> lk(h.stream->mu);
            h.stream->running.emplace_back(fake_kernel, L, readers, lpm, &mode);
            return hipSuccess;
        };
    }
};

static void request(Harness &H, ls_ctx *ctx, int reader, unsigned x, int expect_rc = LS_OK)
{
    unsigned pay[64] = {0};
    pay[0] = 1; pay[1] = x & 0xffffu; pay[2] = (x * 7u) & 0xffffu; // (the payload dwords of every line take part in the checksum)
    if (H.lpm > 1) { pay[16] = x & 0xffu; pay[63] = (x >> 3) & 0xffffu; }
    CHECK(H.h.ensure_running(ctx) == LS_OK);
    if (H.lpm == 1) H.h.post(reader, pay);
    else H.h.post_lines(reader * H.lpm, pay, H.lpm);
    unsigned r[8];
    const int rc = H.h.wait(ctx, reader * H.lpm, r);
    if (rc != expect_rc) fprintf(stderr, "request(reader %d, x %u): rc %d, expected %d (%s)\n", reader, x, rc, expect_rc, ctx->err.c_str());
    CHECK(rc == expect_rc);
    if (rc != LS_OK) return;
    unsigned sum = 0;
    for (int d = 1; d < 16 * H.lpm; d++) sum += pay[d] & 0xffffu;
    CHECK(r[0] == (unsigned)reader && r[1] == (sum & 0xffffu));
    for (int d = 2; d < 8; d++) CHECK(r[d] == (((x & 0xffffu) + (unsigned)d) & 0xffffu));
}

int main()
{
    {   // 1. many slots on many threads, the launch idles out between the phases and comes back: every answer right
        Harness H;
        H.start(6, 1, 200);
        std::vector<ls_ctx> ctxs(6);
        for (int phase = 0; phase < 4; phase++) {
            std::vector<std::thread> ts;
            for (int t = 0; t < 6; t++)
                ts.emplace_back([&, t]() {
                    for (unsigned i = 0; i < 750; i++) request(H, &ctxs[(size_t)t], t, (unsigned)phase * 9973u + i * 13u + (unsigned)t);
                });
            for (auto &t : ts) t.join();
            std::this_thread::sleep_for(std::chrono::milliseconds(5)); // nobody asks: the launch idles out, the next phase starts it again
            CHECK(H.h.drained());
        }
        CHECK(H.h.relaunches.load() >= 4);
        CHECK(H.h.requests_total() == 6 * 3000);
        H.h.shutdown();
    }
    {   // 2. the 16-bit tag wraps (tags run 1 .. 0xfffe): > 65 534 requests on ONE slot, then across a relaunch
        Harness H;
        H.start(1, 1, 2000);
        for (unsigned i = 0; i < 70000; i++) request(H, &H.ctx, 0, i);
        CHECK(H.h.per[0].tag >= 1 && H.h.per[0].tag <= 0xfffeu && H.h.per[0].tag == 70000u - 0xfffeu);
        H.h.pause(); // QUIT: the launch leaves at once, not after the idle time
        CHECK(H.h.drained());
        for (unsigned i = 0; i < 100; i++) request(H, &H.ctx, 0, i); // comes back; the last served tag is read back from the answer line
        H.h.shutdown();
    }
    {   // 3. a resident launch that never answers: an error after no_answer_seconds, not a hang; the next request works again
        Harness H;
        H.start(2, 1, 2000000);
        H.h.no_answer_seconds = 0.2;
        request(H, &H.ctx, 1, 5);
        H.mode.store(SILENT);
        const auto t0 = std::chrono::steady_clock::now();
        request(H, &H.ctx, 1, 6, LS_ERR_HIP);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        CHECK(dt > 0.15 && dt < 5.0);
        CHECK(H.ctx.err.find("no answer from slot 1") != std::string::npos);
        H.mode.store(ANSWER);
        request(H, &H.ctx, 0, 7);
        H.h.shutdown();
    }
    {   // 4. launches that leave unanswered (the grid whose workgroups cannot all be resident): give up after relaunch_limit
        Harness H;
        H.start(1, 4, 500);
        H.h.relaunch_limit = 2;
        request(H, &H.ctx, 0, 1);
        H.h.pause();
        H.mode.store(LEAVE_AT_ONCE);
        const unsigned long long before = H.h.relaunches.load();
        request(H, &H.ctx, 0, 2, SRV_RC_GIVE_UP);
        CHECK(H.h.relaunches.load() - before >= 2 && H.h.relaunches.load() - before <= 4);
        H.mode.store(ANSWER); // (an owner that does not give the route up would be served again)
        request(H, &H.ctx, 0, 3);
        H.h.shutdown();
    }
    {   // 5. a launch that fails: the error is reported, and stays reported
        Harness H;
        H.start(1, 1, 500);
        H.fail_launch.store(1);
        CHECK(H.h.ensure_running(&H.ctx) == LS_ERR_HIP && H.ctx.err.find("launch failed") != std::string::npos);
        H.fail_launch.store(0);
        CHECK(H.h.ensure_running(&H.ctx) == LS_ERR_HIP && H.ctx.err.find("an earlier launch failed") != std::string::npos);
        H.h.shutdown();
    }
    {   // 6. four-line messages for three readers on three threads; pause() of a BUSY launch returns long before the idle time
        Harness H;
        H.start(3, 4, 2000000);
        std::atomic<bool> stop{false};
        std::vector<ls_ctx> ctxs(3);
        std::vector<std::thread> ts;
        for (int t = 0; t < 3; t++)
            ts.emplace_back([&, t]() {
                for (unsigned i = 0; i < 4000 && !stop.load(); i++) request(H, &ctxs[(size_t)t], t, i + 100u * (unsigned)t);
            });
        for (auto &t : ts) t.join();
        CHECK(!H.h.drained());
        const auto t0 = std::chrono::steady_clock::now();
        H.h.pause(); // the QUIT reaches every reader as a whole message of ITS size
        CHECK(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1.0);
        CHECK(H.h.drained());
        request(H, &ctxs[0], 2, 9);
        H.h.shutdown();
    }
    printf("OK\n");
    return 0;
}
