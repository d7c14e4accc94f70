// A FAKE hip_runtime.h for CPU-side unit tests of the host logic (tests/cpp/srv_host_test.cpp): just enough of the HIP API for
// csrc/ls_common.hpp and the host half of csrc/srv_common.hpp to compile with g++ and run under AddressSanitizer /
// ThreadSanitizer.  "Device" memory is host memory; a "stream" is a list of host threads (the fake resident kernels) that
// hipStreamSynchronize joins.  Never used by the product build.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#define __device__
#define __global__
#define __host__
#define __forceinline__ inline

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0, hipErrorNotSupported = 801, hipErrorLaunchFailure = 719, hipErrorNotReady = 600;
constexpr unsigned hipHostMallocCoherent = 0x40000000u, hipStreamNonBlocking = 1u;
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

struct FakeStream {
    std::mutex mu;
    std::vector<std::thread> running; // the fake resident kernels launched on this stream
};
typedef FakeStream *hipStream_t;

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake HIP error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return posix_memalign(p, 64, (n + 63) & ~(size_t)63) == 0 ? hipSuccess : 2; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t, const uint32_t *) { *s = new FakeStream(); return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = new FakeStream(); return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t s)
{
    std::vector<std::thread> ts;
    { std::lock_guard<std::mutex> lk(s->mu); ts.swap(s->running); }
    for (auto &t : ts) t.join();
    return hipSuccess;
}
inline hipError_t hipStreamQuery(hipStream_t) { return hipErrorNotReady; }
inline hipError_t hipStreamDestroy(hipStream_t s) { (void)hipStreamSynchronize(s); delete s; return hipSuccess; }
// the only kernels the host half launches are the fill kernels of ls_common.hpp: run them as plain functions
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) kernel##_host(__VA_ARGS__)
inline void fill_i64_kernel_host(long long *p, long long v, long long n) { for (long long i = 0; i < n; i++) p[i] = v; }
inline void fill_i32_kernel_host(int *p, int v, long long n) { for (long long i = 0; i < n; i++) p[i] = v; }
