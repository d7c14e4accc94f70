// tools/valu_rates.hip — issue-rate probe for the VALU instructions the scan kernels use.
// Diagnostic only (not part of the library): measures, on the GPU it runs on,
// cycles per wave-instruction per SIMD for independent instruction streams at
// 1..N waves per SIMD, and the shader clock actually held under that load
// (s_memtime ticks / wall time).  Build: hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o tools/_build/valu_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e = (x);                                                               \
        if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } \
    } while (0)

// 16 independent instructions per block; operands: %0..%15 = 16 VGPR accumulators, %16 %17 = VGPR sources
#define REP16(OP)                                                                         \
    OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")               \
    OP("%8") OP("%9") OP("%10") OP("%11") OP("%12") OP("%13") OP("%14") OP("%15")

#define OP_FMA(d) "v_fma_f32 " d ", %16, %17, " d "\n\t"
#define OP_SUB32(d) "v_sub_f32_e32 " d ", %16, " d "\n\t"
#define OP_SUB64ABS(d) "v_sub_f32_e64 " d ", |" d "|, %17\n\t"
#define OP_MULCLAMP(d) "v_mul_f32_e64 " d ", |" d "|, |%16| clamp\n\t"
#define OP_ADDF(d) "v_add_f32_e32 " d ", %16, " d "\n\t"
#define OP_XOR(d) "v_xor_b32_e32 " d ", %16, " d "\n\t"
#define OP_PKMINU16(d) "v_pk_min_u16 " d ", " d ", %16\n\t"
#define OP_PKADDU16(d) "v_pk_add_u16 " d ", " d ", %16\n\t"
#define OP_ADDU32(d) "v_add_u32_e32 " d ", %16, " d "\n\t"
#define OP_CMPEQ(d) "v_cmp_eq_u32_e64 s[20:21], " d ", %16\n\t"
#define OP_PKADDF32(d) "v_pk_add_f32 %[p0], %[p0], %[p1]\n\t"
#define OP_MIN3F(d) "v_min3_f32 " d ", " d ", %16, %17\n\t"
#define OP_SADU32(d) "v_sad_u32 " d ", " d ", %16, %17\n\t"
#define OP_MADU24(d) "v_mad_u32_u24 " d ", " d ", %16, %17\n\t"
#define OP_PKFMAF16(d) "v_pk_fma_f16 " d ", " d ", %16, %17\n\t"
#define OP_PKMULF16(d) "v_pk_mul_f16 " d ", " d ", %16\n\t"
#define OP_MED3F(d) "v_med3_f32 " d ", " d ", %16, %17\n\t"
#define OP_PKMAXI16(d) "v_pk_max_i16 " d ", " d ", %16\n\t"
#define OP_SUBU32(d) "v_sub_u32_e32 " d ", %16, " d "\n\t"
#define OP_MINU32(d) "v_min_u32_e32 " d ", %16, " d "\n\t"
#define OP_FMAMIX(d) "v_fma_mix_f32 " d ", " d ", %16, %17\n\t"
#define OP_DOT2F16(d) "v_dot2c_f32_f16 " d ", %16, %17\n\t"
#define OP_MIN3U32(d) "v_min3_u32 " d ", " d ", %16, %17\n\t"
#define OP_ANDOR(d) "v_and_or_b32 " d ", " d ", %16, %17\n\t"
#define OP_XAD(d) "v_xad_u32 " d ", " d ", %16, %17\n\t"
#define OP_ADD3(d) "v_add3_u32 " d ", " d ", %16, %17\n\t"
#define OP_LSHLADD(d) "v_lshl_add_u32 " d ", %16, 1, " d "\n\t"
#define OP_FMACLAMP(d) "v_fma_f32 " d ", |" d "|, |%16|, %17 clamp\n\t"

#define KERNEL(NAME, OPM, ...)                                                           \
    __global__ __launch_bounds__(256) void NAME(unsigned long long *out, float *sink, int iters)  \
    {                                                                                     \
        float a[16];                                                                      \
        for (int i = 0; i < 16; i++) a[i] = (float)(threadIdx.x + i);                     \
        float s0 = 3.0f, s1 = 5.0f;                                                       \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                             \
        for (int it = 0; it < iters; it++) {                                              \
            asm volatile(REP16(OPM) REP16(OPM) REP16(OPM) REP16(OPM)                      \
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),   \
                           "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]),           \
                           "+v"(a[13]), "+v"(a[14]), "+v"(a[15])                          \
                         : "v"(s0), "v"(s1)                                               \
                         : __VA_ARGS__);                                                       \
        }                                                                                 \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                             \
        float acc = 0;                                                                    \
        for (int i = 0; i < 16; i++) acc += a[i];                                         \
        if (acc == 12345.678f) sink[0] = acc;                                             \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0; \
    }

KERNEL(k_fma, OP_FMA, "memory")
KERNEL(k_sub32, OP_SUB32, "memory")
KERNEL(k_sub64abs, OP_SUB64ABS, "memory")
KERNEL(k_mulclamp, OP_MULCLAMP, "memory")
KERNEL(k_addf, OP_ADDF, "memory")
KERNEL(k_xor, OP_XOR, "memory")
KERNEL(k_pkminu16, OP_PKMINU16, "memory")
KERNEL(k_pkaddu16, OP_PKADDU16, "memory")
KERNEL(k_addu32, OP_ADDU32, "memory")
KERNEL(k_cmpeq, OP_CMPEQ, "s20", "s21", "memory")
KERNEL(k_min3f, OP_MIN3F, "memory")
KERNEL(k_sadu32, OP_SADU32, "memory")
KERNEL(k_madu24, OP_MADU24, "memory")
KERNEL(k_pkfmaf16, OP_PKFMAF16, "memory")
KERNEL(k_pkmulf16, OP_PKMULF16, "memory")
KERNEL(k_med3f, OP_MED3F, "memory")
KERNEL(k_pkmaxi16, OP_PKMAXI16, "memory")
KERNEL(k_subu32, OP_SUBU32, "memory")
KERNEL(k_minu32, OP_MINU32, "memory")
KERNEL(k_fmamix, OP_FMAMIX, "memory")
KERNEL(k_dot2f16, OP_DOT2F16, "memory")
KERNEL(k_min3u32, OP_MIN3U32, "memory")
KERNEL(k_andor, OP_ANDOR, "memory")
KERNEL(k_xad, OP_XAD, "memory")
KERNEL(k_add3, OP_ADD3, "memory")
KERNEL(k_lshladd, OP_LSHLADD, "memory")
KERNEL(k_fmaclamp, OP_FMACLAMP, "memory")

// packed f32 needs 64-bit register pairs
__global__ __launch_bounds__(256) void k_pkaddf32(unsigned long long *out, float *sink, int iters)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8];
    for (int i = 0; i < 8; i++) a[i] = f2{(float)threadIdx.x, (float)i};
    f2 s = f2{3.0f, 5.0f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#define P(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n\t"
#define P8 P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7)
        asm volatile(P8 P8 P8 P8 P8 P8 P8 P8
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(s)
                     : "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int i = 0; i < 8; i++) acc += a[i].x + a[i].y;
    if (acc == 12345.678f) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_pkfmaf32(unsigned long long *out, float *sink, int iters)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8];
    for (int i = 0; i < 8; i++) a[i] = f2{(float)threadIdx.x, (float)i};
    f2 s = f2{1.0f, 1.0f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#define Q(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %8\n\t"
#define Q8 Q(0) Q(1) Q(2) Q(3) Q(4) Q(5) Q(6) Q(7)
        asm volatile(Q8 Q8 Q8 Q8 Q8 Q8 Q8 Q8
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(s)
                     : "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int i = 0; i < 8; i++) acc += a[i].x + a[i].y;
    if (acc == 12345.678f) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

// The scan kernel's own instruction mix (sub e32, sub e64 |abs|, mul e64 |abs||abs| clamp, add e32; 4 candidates
// interleaved) as a SHORT loop body (64 instructions) and as a LONG straight-line body (1024 instructions, the
// unrolled 64-column step) — separates issue rate from instruction-fetch effects.
#define MIX_COL                                                                                         \
    "v_sub_f32_e32 %4, %16, %12\n\tv_sub_f32_e32 %5, %16, %13\n\tv_sub_f32_e32 %6, %16, %14\n\tv_sub_f32_e32 %7, %16, %15\n\t" \
    "v_sub_f32_e64 %8, |%4|, %17\n\tv_sub_f32_e64 %9, |%5|, %17\n\tv_sub_f32_e64 %10, |%6|, %17\n\tv_sub_f32_e64 %11, |%7|, %17\n\t" \
    "v_mul_f32_e64 %4, |%4|, |%8| clamp\n\tv_mul_f32_e64 %5, |%5|, |%9| clamp\n\tv_mul_f32_e64 %6, |%6|, |%10| clamp\n\tv_mul_f32_e64 %7, |%7|, |%11| clamp\n\t" \
    "v_add_f32_e32 %0, %0, %4\n\tv_add_f32_e32 %1, %1, %5\n\tv_add_f32_e32 %2, %2, %6\n\tv_add_f32_e32 %3, %3, %7\n\t"
#define MIX4 MIX_COL MIX_COL MIX_COL MIX_COL
#define MIX16 MIX4 MIX4 MIX4 MIX4
#define MIX64 MIX16 MIX16 MIX16 MIX16
#define MIX_KERNEL(NAME, BODY, DIV)                                                                     \
    __global__ __launch_bounds__(256) void NAME(unsigned long long *out, float *sink, int iters)        \
    {                                                                                                   \
        float a[16];                                                                                    \
        for (int i = 0; i < 16; i++) a[i] = (float)(threadIdx.x + i);                                   \
        float s0 = 3.0f, s1 = 5.0f;                                                                     \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                           \
        for (int it = 0; it < iters / DIV; it++) {                                                      \
            asm volatile(BODY                                                                           \
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),   \
                           "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]),           \
                           "+v"(a[13]), "+v"(a[14]), "+v"(a[15])                                        \
                         : "v"(s0), "v"(s1)                                                             \
                         : "memory");                                                                   \
        }                                                                                               \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                           \
        float acc = 0;                                                                                  \
        for (int i = 0; i < 16; i++) acc += a[i];                                                       \
        if (acc == 12345.678f) sink[0] = acc;                                                           \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;              \
    }
MIX_KERNEL(k_mix_short64, MIX4, 1)
MIX_KERNEL(k_mix_long1024, MIX64, 16)
// all-e32 control of the same length: 1024 x 4-byte encodings
#define E32_COL                                                                                         \
    "v_sub_f32_e32 %4, %16, %12\n\tv_sub_f32_e32 %5, %16, %13\n\tv_sub_f32_e32 %6, %16, %14\n\tv_sub_f32_e32 %7, %16, %15\n\t" \
    "v_sub_f32_e32 %8, %4, %17\n\tv_sub_f32_e32 %9, %5, %17\n\tv_sub_f32_e32 %10, %6, %17\n\tv_sub_f32_e32 %11, %7, %17\n\t" \
    "v_mul_f32_e32 %4, %4, %8\n\tv_mul_f32_e32 %5, %5, %9\n\tv_mul_f32_e32 %6, %6, %10\n\tv_mul_f32_e32 %7, %7, %11\n\t" \
    "v_add_f32_e32 %0, %0, %4\n\tv_add_f32_e32 %1, %1, %5\n\tv_add_f32_e32 %2, %2, %6\n\tv_add_f32_e32 %3, %3, %7\n\t"
#define E4 E32_COL E32_COL E32_COL E32_COL
#define E16 E4 E4 E4 E4
#define E64 E16 E16 E16 E16
MIX_KERNEL(k_e32_long1024, E64, 16)

// the two-column walk's mix with the two board columns' flags added by ONE v_add3_u32 per candidate (24 instructions per
// 8 (candidate, column) pairs: 4 sub e32, 8 sub e64, 8 mul e64, 4 add3) against the same with v_add_f32 (28)
#define ADD3_HALF(acc0, acc1, acc2, acc3)                                                                                       \
    "v_sub_f32_e32 %4, %16, %12\n\tv_sub_f32_e32 %5, %16, %13\n\tv_sub_f32_e32 %6, %16, %14\n\tv_sub_f32_e32 %7, %16, %15\n\t" \
    "v_sub_f32_e64 %8, |%4|, %17\n\tv_sub_f32_e64 %9, |%5|, %17\n\tv_sub_f32_e64 %10, |%6|, %17\n\tv_sub_f32_e64 %11, |%7|, %17\n\t" \
    "v_mul_f32_e64 %8, |%4|, |%8| clamp\n\tv_mul_f32_e64 %9, |%5|, |%9| clamp\n\tv_mul_f32_e64 %10, |%6|, |%10| clamp\n\tv_mul_f32_e64 %11, |%7|, |%11| clamp\n\t" \
    "v_sub_f32_e64 %4, |%4|, %16\n\tv_sub_f32_e64 %5, |%5|, %16\n\tv_sub_f32_e64 %6, |%6|, %16\n\tv_sub_f32_e64 %7, |%7|, %16\n\t" \
    "v_mul_f32_e64 %4, |%12|, |%4| clamp\n\tv_mul_f32_e64 %5, |%13|, |%5| clamp\n\tv_mul_f32_e64 %6, |%14|, |%6| clamp\n\tv_mul_f32_e64 %7, |%15|, |%7| clamp\n\t" \
    "v_add3_u32 " acc0 ", " acc0 ", %8, %4\n\tv_add3_u32 " acc1 ", " acc1 ", %9, %5\n\tv_add3_u32 " acc2 ", " acc2 ", %10, %6\n\tv_add3_u32 " acc3 ", " acc3 ", %11, %7\n\t"
#define ADD3_COL ADD3_HALF("%0", "%1", "%2", "%3")
#define A4 ADD3_COL ADD3_COL ADD3_COL ADD3_COL
#define A16 A4 A4 A4 A4
MIX_KERNEL(k_add3_mix_1536, A16 A16 A16 A16, 24)

// the 8 x 8 tile's mix: sub e64 |abs||abs|, mul e64 |abs||abs| clamp, add e32 (two thirds of the instructions are 8-byte encodings:
// 6.67 bytes per instruction against 6.0 for the scan mix) -- and the same with the add as an 8-byte encoding too (8.0)
#define T8_GRP(ADD)                                                                                      \
    "v_sub_f32_e64 %8, |%4|, |%16|\n\tv_sub_f32_e64 %9, |%5|, |%16|\n\tv_sub_f32_e64 %10, |%6|, |%16|\n\tv_sub_f32_e64 %11, |%7|, |%16|\n\t" \
    "v_mul_f32_e64 %8, |%4|, |%8| clamp\n\tv_mul_f32_e64 %9, |%5|, |%9| clamp\n\tv_mul_f32_e64 %10, |%6|, |%10| clamp\n\tv_mul_f32_e64 %11, |%7|, |%11| clamp\n\t" \
    ADD " %0, %0, %8\n\t" ADD " %1, %1, %9\n\t" ADD " %2, %2, %10\n\t" ADD " %3, %3, %11\n\t" \
    "v_sub_f32_e64 %8, |%12|, |%17|\n\tv_sub_f32_e64 %9, |%13|, |%17|\n\tv_sub_f32_e64 %10, |%14|, |%17|\n\tv_sub_f32_e64 %11, |%15|, |%17|\n\t" \
    "v_mul_f32_e64 %8, |%12|, |%8| clamp\n\tv_mul_f32_e64 %9, |%13|, |%9| clamp\n\tv_mul_f32_e64 %10, |%14|, |%10| clamp\n\tv_mul_f32_e64 %11, |%15|, |%11| clamp\n\t" \
    ADD " %0, %0, %8\n\t" ADD " %1, %1, %9\n\t" ADD " %2, %2, %10\n\t" ADD " %3, %3, %11\n\t"
#define T8A T8_GRP("v_add_f32_e32")
#define T8B T8_GRP("v_add_f32_e64")
#define T8A8 T8A T8A T8A T8A T8A T8A T8A T8A
#define T8B8 T8B T8B T8B T8B T8B T8B T8B T8B
MIX_KERNEL(k_t8_mix, T8A8 T8A8 T8A8 T8A8, 12)
MIX_KERNEL(k_t8_mix_e64, T8B8 T8B8 T8B8 T8B8, 12)

typedef void (*kern_t)(unsigned long long *, float *, int);

struct Entry { const char *name; kern_t fn; };

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    Entry tab[] = {
        {"scan mix, 64-instr loop", k_mix_short64}, {"scan mix, 1024 straight", k_mix_long1024},
        {"all-e32 mix, 1024 straight", k_e32_long1024},
        {"v_fma_f32", k_fma}, {"v_sub_f32_e32", k_sub32}, {"v_sub_f32_e64|abs|", k_sub64abs},
        {"v_mul_f32_e64|abs|clamp", k_mulclamp}, {"v_add_f32_e32", k_addf}, {"v_min3_f32", k_min3f},
        {"v_med3_f32", k_med3f}, {"v_pk_add_f32", k_pkaddf32}, {"v_pk_fma_f32", k_pkfmaf32},
        {"v_pk_fma_f16", k_pkfmaf16}, {"v_pk_mul_f16", k_pkmulf16}, {"v_fma_mix_f32", k_fmamix},
        {"v_dot2c_f32_f16", k_dot2f16},
        {"v_xor_b32", k_xor}, {"v_add_u32", k_addu32}, {"v_sub_u32", k_subu32}, {"v_min_u32", k_minu32},
        {"v_min3_u32", k_min3u32}, {"v_and_or_b32", k_andor}, {"v_xad_u32", k_xad},
        {"v_pk_min_u16", k_pkminu16}, {"v_pk_add_u16", k_pkaddu16}, {"v_pk_max_i16", k_pkmaxi16},
        {"v_sad_u32", k_sadu32}, {"v_mad_u32_u24", k_madu24}, {"v_cmp_eq_u32_e64(sdst)", k_cmpeq},
        {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshladd}, {"v_fma_f32|abs|clamp", k_fmaclamp},
        {"add3 mix (20 f32 + 4 add3), 1536 straight", k_add3_mix_1536},
        {"tile8 mix (6.67 B/instr), 768 straight", k_t8_mix}, {"tile8 mix, add e64 (8 B/instr), 768 straight", k_t8_mix_e64},
    };
    const char *only = argc > 2 ? argv[2] : nullptr; // substring filter on the names
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, iters %d (x64 instructions per wave per iteration; pk_f32 kernels also 64)\n",
           prop.gcnArchName, cus, iters);
    unsigned long long *out;
    float *sink;
    CHECK(hipMalloc(&out, sizeof(unsigned long long) * cus * 8 * 4));
    CHECK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-28s %5s %12s %12s %10s\n", "instruction", "w/SIMD", "cyc/instr/SIMD", "ns/instr/SIMD", "clock GHz");
    for (const Entry &en : tab) {
        if (only && !strstr(en.name, only)) continue;
        for (int wps : {1, 2, 3, 4}) {
            const int blocks = cus * wps; // 256-thread blocks: one wave per SIMD each
            hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, sink, iters / 10); // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, sink, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(blocks * 4);
            CHECK(hipMemcpy(h.data(), out, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost));
            double avg = 0;
            for (auto v : h) avg += (double)v;
            avg /= h.size();
            const double n_instr = (double)iters * 64.0;
            const double cyc = avg / (n_instr * wps);          // per instruction per SIMD (wps waves share a SIMD)
            const double ns = ms * 1e6 / (n_instr * wps);
            printf("%-28s %5d %12.2f %12.3f %10.2f\n", en.name, wps, cyc, ns, cyc / ns);
        }
    }
    return 0;
}
