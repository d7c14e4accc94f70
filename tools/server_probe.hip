// tools/server_probe.hip — facts the persistent LS-step server (csrc/nq_server.hip) is designed on.  Diagnostic only.
//   1. mailbox round trip: host writes a 64-byte request line whose 16 dwords each carry a 16-bit sequence tag, a
//      resident workgroup polls the line (one dword per lane), answers with a tagged response line in pinned host
//      memory; the host polls that.  Mailbox in (a) pinned coherent host memory (the GPU polls over PCIe) or (b) device
//      memory the host writes through the PCIe BAR, if the platform maps it.
//   2. does a never-ending kernel on its own stream (plain / high priority / CU-masked) block kernels launched on other
//      streams afterwards (HIP multiplexes streams onto a few hardware queues and a queue runs its packets in order)?
//   3. what hipMalloc / hipFree / hipHostMalloc cost while such a kernel is resident.
// Every kernel here leaves on a wall-clock limit whatever the host does.
// Build: hipcc --offload-arch=gfx950 -O2 tools/server_probe.hip -o tools/_build/server_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <immintrin.h>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e = (x);                                                               \
        if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } \
    } while (0)

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// One workgroup = one mailbox.  Line layout: dword i = (tag16 << 16) | payload16; tag 0xffff = quit.
__global__ __launch_bounds__(1024) void server_kernel(const unsigned *mail, unsigned *resp, unsigned long long *stats,
                                                      unsigned long long limit_ticks, int work_iters)
{
    __shared__ unsigned s_req[16];
    __shared__ int s_quit;
    const int tid = threadIdx.x;
    const unsigned *line = mail + blockIdx.x * 16;
    unsigned *out = resp + blockIdx.x * 16;
    const unsigned long long t_start = wall_clock64();
    unsigned last = 0;
    unsigned long long polls = 0, served = 0;
    for (;;) {
        if (tid < 64) { // wave 0 polls
            int quit = 0;
            for (;;) {
                unsigned v = 0;
                if (tid < 16) v = __hip_atomic_load(line + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned tag = v >> 16;
                const unsigned t0 = __builtin_amdgcn_readfirstlane(tag);
                const bool same = __builtin_amdgcn_ballot_w64(tid < 16 && tag != t0) == 0;
                polls++;
                if (same && t0 != last) {
                    if (tid < 16) s_req[tid] = v;
                    last = t0;
                    quit = t0 == 0xffffu;
                    break;
                }
                if (wall_clock64() - t_start > limit_ticks) { quit = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (tid == 0) s_quit = quit;
        }
        __syncthreads();
        if (s_quit) break;
        // stand-in for the LS step: some LDS traffic by all waves
        unsigned acc = s_req[tid & 15];
        for (int i = 0; i < work_iters; i++) acc = acc * 1664525u + s_req[(acc >> 8) & 15];
        __syncthreads();
        if (tid < 8) {
            const unsigned tag = s_req[0] >> 16;
            const unsigned pay = tid == 7 ? (acc & 0xffffu) : (s_req[tid] & 0xffffu) + 1u;
            __hip_atomic_store(out + tid, (tag << 16) | (pay & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        served++;
    }
    if (tid == 0) { stats[blockIdx.x * 2] = polls; stats[blockIdx.x * 2 + 1] = served; }
}

__global__ void tiny_kernel(int *p) { if (threadIdx.x == 0) atomicAdd(p, 1); }

static void write_line(volatile unsigned *line, unsigned tag, const unsigned short *payload, bool wc)
{
    alignas(64) unsigned tmp[16];
    for (int i = 0; i < 16; i++) tmp[i] = (tag << 16) | payload[i];
    _mm256_store_si256((__m256i *)line, _mm256_load_si256((const __m256i *)tmp));
    _mm256_store_si256((__m256i *)line + 1, _mm256_load_si256((const __m256i *)tmp + 1));
    if (wc) _mm_sfence(); // device memory through the BAR: push the write-combining buffer out
}

static bool resp_ready(const volatile unsigned *out, unsigned tag)
{
    for (int i = 0; i < 8; i++)
        if ((out[i] >> 16) != tag) return false;
    return true;
}

static sigjmp_buf g_jmp;
static void on_segv(int) { siglongjmp(g_jmp, 1); }

static bool host_can_write(void *p)
{
    struct sigaction sa, old_segv, old_bus;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_segv;
    sigaction(SIGSEGV, &sa, &old_segv);
    sigaction(SIGBUS, &sa, &old_bus);
    bool ok = false;
    if (sigsetjmp(g_jmp, 1) == 0) {
        volatile unsigned *q = (volatile unsigned *)p;
        q[0] = 0x12345678u;
        _mm_sfence();
        ok = q[0] == 0x12345678u;
    }
    sigaction(SIGSEGV, &old_segv, nullptr);
    sigaction(SIGBUS, &old_bus, nullptr);
    return ok;
}

static void pingpong(const char *label, unsigned *mail_host_view, const unsigned *mail_dev_view, bool wc, int n_blocks, int iters, int work_iters,
                     hipStream_t s)
{
    unsigned *resp = nullptr;
    unsigned long long *stats = nullptr;
    CHECK(hipHostMalloc((void **)&resp, 64 * n_blocks, hipHostMallocCoherent));
    CHECK(hipHostMalloc((void **)&stats, 16 * n_blocks, hipHostMallocCoherent));
    memset(resp, 0, 64 * n_blocks);
    unsigned short pay[16] = {0};
    for (int b = 0; b < n_blocks; b++) write_line(mail_host_view + 16 * b, 0, pay, wc);
    const double t_l0 = now_s();
    hipLaunchKernelGGL(server_kernel, dim3(n_blocks), dim3(1024), 0, s, mail_dev_view, resp, stats, 100000000ull * 3, work_iters); // 3 s limit
    CHECK(hipGetLastError());
    // first round trip includes the launch
    for (int i = 0; i < 16; i++) pay[i] = (unsigned short)(i * 3);
    write_line(mail_host_view, 1, pay, wc);
    while (!resp_ready(resp, 1)) _mm_pause();
    const double t_first = now_s() - t_l0;
    // steady state on block 0
    std::vector<double> lat;
    unsigned tag = 1;
    const double t0 = now_s();
    for (int it = 0; it < iters; it++) {
        tag = tag == 0xfffeu ? 1 : tag + 1;
        pay[0] = (unsigned short)it;
        const double a = now_s();
        write_line(mail_host_view, tag, pay, wc);
        while (!resp_ready(resp, tag)) _mm_pause();
        lat.push_back(now_s() - a);
        if ((resp[0] & 0xffffu) != ((unsigned)(pay[0] + 1) & 0xffffu)) { printf("%s: wrong payload echo\n", label); break; }
    }
    const double dt = now_s() - t0;
    std::sort(lat.begin(), lat.end());
    // all blocks at once, one host thread each would be the real thing; here: round-robin from one thread
    double t_rr = 0;
    if (n_blocks > 1) {
        const double r0 = now_s();
        unsigned tg = 100;
        for (int it = 0; it < 2000; it++) {
            tg++;
            for (int b = 0; b < n_blocks; b++) write_line(mail_host_view + 16 * b, tg, pay, wc);
            for (int b = 0; b < n_blocks; b++) while (!resp_ready(resp + 16 * b, tg)) _mm_pause();
        }
        t_rr = (now_s() - r0) / 2000;
    }
    for (int b = 0; b < n_blocks; b++) write_line(mail_host_view + 16 * b, 0xffffu, pay, wc);
    const double q0 = now_s();
    CHECK(hipStreamSynchronize(s));
    const double t_quit = now_s() - q0;
    printf("%-46s blocks %3d work %4d: first (launch incl.) %.1f us; round trip mean %.2f us, p50 %.2f, p99 %.2f, min %.2f; all-blocks sweep %.2f us; "
           "quit->drained %.1f us; polls/served of block 0: %llu / %llu\n",
           label, n_blocks, work_iters, t_first * 1e6, dt / iters * 1e6, lat[lat.size() / 2] * 1e6, lat[lat.size() * 99 / 100] * 1e6, lat[0] * 1e6,
           t_rr * 1e6, t_quit * 1e6, stats[0], stats[1]);
    CHECK(hipHostFree(resp));
    CHECK(hipHostFree(stats));
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    hipStream_t s_plain, s_prio, s_mask;
    CHECK(hipStreamCreateWithFlags(&s_plain, hipStreamNonBlocking));
    int lo = 0, hi = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d greatest %d\n", lo, hi);
    CHECK(hipStreamCreateWithPriority(&s_prio, hipStreamNonBlocking, hi));
    std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0xffffffffu);
    hipError_t em = hipExtStreamCreateWithCUMask(&s_mask, (uint32_t)mask.size(), mask.data());
    printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(em));

    // ---- 1. mailbox round trips
    unsigned *mail_h = nullptr;
    CHECK(hipHostMalloc((void **)&mail_h, 64 * 256, hipHostMallocCoherent));
    for (int blocks : {1, 64}) {
        for (int work : {0, 200}) pingpong("mailbox in pinned host memory (GPU polls PCIe)", mail_h, mail_h, false, blocks, iters, work, s_plain);
    }
    {
        unsigned *mail_d = nullptr;
        CHECK(hipMalloc((void **)&mail_d, 64 * 256));
        CHECK(hipMemset(mail_d, 0, 64 * 256));
        CHECK(hipDeviceSynchronize());
        const bool ok = host_can_write(mail_d);
        printf("host store to hipMalloc memory: %s\n", ok ? "works (BAR-mapped)" : "faults");
        if (ok) for (int blocks : {1, 64}) pingpong("mailbox in device memory (host writes the BAR)", mail_d, mail_d, true, blocks, iters, 0, s_plain);
        unsigned *mail_f = nullptr;
        hipError_t ef = hipExtMallocWithFlags((void **)&mail_f, 64 * 256, hipDeviceMallocFinegrained);
        printf("hipExtMallocWithFlags(fine-grained): %s\n", hipGetErrorString(ef));
        if (ef == hipSuccess) {
            CHECK(hipMemset(mail_f, 0, 64 * 256));
            CHECK(hipDeviceSynchronize());
            const bool okf = host_can_write(mail_f);
            printf("host store to fine-grained device memory: %s\n", okf ? "works" : "faults");
            if (okf) for (int blocks : {1, 64}) pingpong("mailbox in fine-grained device memory (BAR)", mail_f, mail_f, true, blocks, iters, 0, s_plain);
        }
    }

    // ---- 2. does a resident kernel block other streams?
    int *counter = nullptr;
    CHECK(hipMalloc((void **)&counter, 4));
    CHECK(hipMemset(counter, 0, 4));
    CHECK(hipDeviceSynchronize());
    struct Case { const char *name; hipStream_t s; } cases[] = {{"plain stream", s_plain}, {"high-priority stream", s_prio}, {"CU-masked stream", s_mask}};
    for (const Case &c : cases) {
        if (!c.s) continue;
        unsigned *resp = nullptr; unsigned long long *stats = nullptr;
        CHECK(hipHostMalloc((void **)&resp, 64 * 64, hipHostMallocCoherent));
        CHECK(hipHostMalloc((void **)&stats, 16 * 64, hipHostMallocCoherent));
        unsigned short pay[16] = {0};
        for (int b = 0; b < 64; b++) write_line(mail_h + 16 * b, 0, pay, false);
        hipLaunchKernelGGL(server_kernel, dim3(64), dim3(1024), 0, c.s, mail_h, resp, stats, 100000000ull / 2, 0); // leaves after 0.5 s by itself
        CHECK(hipGetLastError());
        write_line(mail_h, 1, pay, false);
        while (!resp_ready(resp, 1)) _mm_pause(); // resident now
        double worst = 0;
        std::vector<hipStream_t> others(12);
        for (auto &o : others) CHECK(hipStreamCreateWithFlags(&o, hipStreamNonBlocking));
        for (size_t i = 0; i < others.size(); i++) {
            const double a = now_s();
            hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, others[i], counter);
            CHECK(hipStreamSynchronize(others[i]));
            const double d = now_s() - a;
            worst = d > worst ? d : worst;
            printf("  resident on %-22s other stream %2zu: launch+sync %.1f us%s\n", c.name, i, d * 1e6, d > 0.05 ? "   <-- waited for the resident kernel" : "");
        }
        // 3. allocation calls while resident
        {
            void *p = nullptr; double a = now_s();
            CHECK(hipMalloc(&p, 1 << 20)); const double t_m = now_s() - a; a = now_s();
            void *h = nullptr; CHECK(hipHostMalloc(&h, 1 << 16, hipHostMallocCoherent)); const double t_h = now_s() - a; a = now_s();
            CHECK(hipHostFree(h)); const double t_hf = now_s() - a; a = now_s();
            CHECK(hipFree(p)); const double t_f = now_s() - a;
            printf("  while resident: hipMalloc %.1f us, hipHostMalloc %.1f us, hipHostFree %.1f us, hipFree %.1f us\n", t_m * 1e6, t_h * 1e6, t_hf * 1e6, t_f * 1e6);
        }
        for (int b = 0; b < 64; b++) write_line(mail_h + 16 * b, 0xffffu, pay, false);
        CHECK(hipStreamSynchronize(c.s));
        for (auto &o : others) CHECK(hipStreamDestroy(o));
        printf("resident kernel on a %s: worst launch+sync on 12 later streams %.1f us\n", c.name, worst * 1e6);
        CHECK(hipHostFree(resp)); CHECK(hipHostFree(stats));
    }
    printf("done\n");
    return 0;
}
