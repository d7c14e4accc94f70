// Probe: do LDS reads and VALU work of DIFFERENT wavefronts overlap on a gfx950 CU for the instruction mix of the TABLE
// kernel's column walk (nq_pairs_unit: per four pair-steps 12 dword look-ups -- six ds_read2st64_b32 -- and 32 VALU)?
// Three kernels with the same loop skeleton: look-ups + arithmetic, look-ups only (one add per dword), arithmetic only
// (operands from registers).  If the units overlapped, the first would take the larger of the other two.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/lds_valu_overlap tools/lds_valu_overlap.hip && tools/_build/lds_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE> // 0 both, 1 look-ups only, 2 arithmetic only
__global__ __launch_bounds__(1024) void walk(unsigned *out, int iters, int n)
{
    extern __shared__ unsigned s_t[];
    const int tid = threadIdx.x, tv = tid & 63;
    for (int i = tid; i < 3 * n; i += 1024) s_t[i] = (unsigned)(i * 2654435761u) & 0x00ff00ffu;
    __syncthreads();
    const unsigned *pR = s_t + tv, *pA = s_t + n + tv + (blockIdx.x & 31), *pB = s_t + 2 * n + tv - (blockIdx.x & 15) + 16;
    unsigned m = 0xffffffffu, acc = 0;
    const int steps = n >> 6; // 64-dword strides inside one table
    for (int it = 0; it < iters; it++) {
        for (int i = 0; i + 4 <= steps; i += 4) {
            unsigned s[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (MODE == 2) s[j] = (unsigned)(i + j) * 0x01010101u + (unsigned)tv + acc;
                else s[j] = pR[(i + j) << 6] + pA[(i + j) << 6] + pB[(i + j) << 6];
            }
            if (MODE == 1) { acc += s[0] + s[1] + s[2] + s[3]; continue; }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned lo = ((s[j] << 15) & 0x7fffffffu) | (unsigned)(2 * (i + j));
                const unsigned hi = ((s[j] >> 16) << 15) | (unsigned)(2 * (i + j) + 1);
                const unsigned mm = lo < hi ? lo : hi;
                m = mm < m ? mm : m;
            }
            if (MODE == 2) acc += m & 1u;
        }
    }
    out[blockIdx.x * 1024 + tid] = m + acc;
}

template <int MODE> static int run(const char *what, unsigned *out, int blocks, int iters, int n, double *us)
{
    const size_t lds = (size_t)3 * n * 4 + 4096;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&walk<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(walk<MODE>, dim3(blocks), dim3(1024), lds, 0, out, iters, n);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(walk<MODE>, dim3(blocks), dim3(1024), lds, 0, out, iters, n);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    *us = ms * 1e3 / 5;
    printf("%-34s %9.1f us per launch (%d blocks x 1024 threads, %zu KB LDS each, %d walks of %d dwords per table)\n", what, *us, blocks, lds >> 10, iters, n);
    return 0;
}

int main(int argc, char **argv)
{
    const int n = 4096 * 3 / 2 / 3 * 2; // dwords per table: the pairs kernel keeps 6n + 4 u16 = 3n dwords at n = 4096
    const int iters = argc > 1 ? atoi(argv[1]) : 64;
    unsigned *out;
    CHECK(hipMalloc(&out, (size_t)1024 * 1024 * 4));
    for (int blocks : {256, 512}) {
        double both, lds, valu;
        if (run<0>("look-ups + arithmetic", out, blocks, iters, n, &both)) return 1;
        if (run<1>("look-ups only", out, blocks, iters, n, &lds)) return 1;
        if (run<2>("arithmetic only", out, blocks, iters, n, &valu)) return 1;
        printf("  => both / max(parts) = %.2f, both / sum(parts) = %.2f\n", both / (lds > valu ? lds : valu), both / (lds + valu));
    }
    return 0;
}
