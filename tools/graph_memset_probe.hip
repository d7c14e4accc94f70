// Root-cause probe for the hipGraph memset finding of round 1 (gpurun_out/gputests11.log: a captured
// hipMemsetAsync + dependent kernel did not behave on replay).  Captures, on one stream,
//     hipMemsetAsync(buf, 0xff, 64 B)  ->  consume(buf, in, out): out[i] = buf[i] + in[i]; buf[i] = 7 (dirty it)
// and replays the graph three times with different `in`.  If the memset node re-executes, every replay sees
// buf == 0xffffffff; if it is skipped (or its extent / value frozen wrong), later replays see the dirtied 7.
// Also tried: the same with hipMemsetD32Async and with a 8-byte memset (the size the argmin keys used: one u64 per
// restart), since small memsets take a different path (fill kernel vs. blit) in ROCclr.
//   hipcc --offload-arch=gfx950 tools/graph_memset_probe.hip -o tools/_build/graph_memset_probe && tools/_build/graph_memset_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

__global__ void consume(unsigned *buf, const unsigned *in, unsigned *out, int n)
{
    const int i = threadIdx.x;
    if (i < n) { out[i] = buf[i] + in[i]; buf[i] = 7u; }
}

static int probe(const char *name, int words, int kind)
{
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *buf, *in, *out;
    CK(hipMalloc(&buf, 64 * 4)); CK(hipMalloc(&in, 64 * 4)); CK(hipMalloc(&out, 64 * 4));
    unsigned h_in[64], h_out[64];
    CK(hipMemset(buf, 0, 64 * 4));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    if (kind == 0) CK(hipMemsetAsync(buf, 0xff, (size_t)words * 4, s));
    else CK(hipMemsetD32Async((hipDeviceptr_t)buf, 0xffffffffu, (size_t)words, s));
    hipLaunchKernelGGL(consume, dim3(1), dim3(64), 0, s, buf, in, out, words);
    CK(hipStreamEndCapture(s, &g));
    size_t nn = 0;
    CK(hipGraphGetNodes(g, nullptr, &nn));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    int bad = 0;
    for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < 64; i++) h_in[i] = 100u * (rep + 1) + i;
        CK(hipMemcpy(in, h_in, sizeof h_in, hipMemcpyHostToDevice));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h_out, out, sizeof h_out, hipMemcpyDeviceToHost));
        int wrong = 0;
        for (int i = 0; i < words; i++) wrong += h_out[i] != 0xffffffffu + h_in[i];
        printf("%-28s words %2d nodes %zu replay %d: %s (out[0] = 0x%08x, expected 0x%08x%s)\n", name, words, nn, rep,
               wrong ? "MEMSET NOT SEEN" : "ok", h_out[0], 0xffffffffu + h_in[0],
               wrong && h_out[0] == 7u + h_in[0] ? "; kernel saw the value it wrote in the previous replay" : "");
        bad += wrong != 0;
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    CK(hipFree(buf)); CK(hipFree(in)); CK(hipFree(out)); CK(hipStreamDestroy(s));
    return bad;
}

int main()
{
    int bad = 0;
    bad += probe("hipMemsetAsync 0xff", 16, 0);
    bad += probe("hipMemsetAsync 0xff", 2, 0);   // 8 bytes: one argmin key
    bad += probe("hipMemsetD32Async", 16, 1);
    bad += probe("hipMemsetD32Async", 2, 1);
    printf("summary: %d replay(s) did not see the captured memset\n", bad);
    return 0;
}
