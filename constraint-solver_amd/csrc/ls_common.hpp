// ls_common.hpp — context object, error plumbing and wavefront helpers shared by
// the gfx950 kernels.  Product code: never includes or links anything under
// oracle/.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/ls_amd.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the kernels are written for gfx950: wave64, stores counted in vmcnt (the completion words are published after s_waitcnt vmcnt(0)), DPP row operations"
#endif

namespace lsamd {

// ---------------------------------------------------------------- device bufs
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool owned = true; // false: the block belongs to another DevBuf (borrow()) and is never freed or retired here
    // Buffers this one outgrew.  Launches already queued on the context's stream may still use them, so they
    // are not freed on the spot (hipFree would also stall the whole device until it drains): they are handed
    // back by collect(), which the API calls right after a stream synchronisation, and by release().
    std::vector<void *> retired;
    // grow-only (geometric, so a slowly growing batch retires O(log) buffers); contents are not preserved
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        size_t want = bytes < 256 ? 256 : bytes;
        if (cap && want < cap + cap / 2) want = cap + cap / 2;
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, want);
        if (e != hipSuccess && want > bytes) { want = bytes; e = hipMalloc(&q, want); }
        if (e != hipSuccess) return e;
        if (p && owned) retired.push_back(p);
        p = q; cap = want; owned = true;
        return hipSuccess;
    }
    void collect() // only when the stream that used the retired buffers has drained
    {
        for (void *q : retired) (void)hipFree(q);
        retired.clear();
    }
    void release()
    {
        collect();
        if (p && owned) (void)hipFree(p);
        p = nullptr; cap = 0; owned = true;
    }
    // read-only view of a block another DevBuf owns (and keeps alive for as long as this view is used)
    void borrow(const DevBuf &o) { release(); p = o.p; cap = o.cap; owned = false; }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// Pinned host staging + one contiguous device block: a host-pointer call packs all its inputs, moves
// them with ONE host-to-device copy, and gets all its outputs back with ONE device-to-host copy
// (pageable hipMemcpyAsync calls cost several microseconds each and dominate small launches).
struct Staging {
    void *host = nullptr;   // hipHostMalloc
    void *dev = nullptr;    // hipMalloc
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        release();
        size_t want = bytes < 4096 ? 4096 : bytes + bytes / 4;
        // coherent (fine-grained) whatever HIP_HOST_COHERENT says: kernels read and write this block in place and publish
        // completion words next to it -- there is no non-coherent fallback (the polled routes would not be sound on it)
        hipError_t e = hipHostMalloc(&host, want, hipHostMallocCoherent);
        if (e != hipSuccess) { host = nullptr; return e; }
        e = hipMalloc(&dev, want);
        if (e != hipSuccess) { (void)hipHostFree(host); host = nullptr; dev = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (host) (void)hipHostFree(host);
        if (dev) (void)hipFree(dev);
        host = dev = nullptr; cap = 0;
    }
    template <class T> T *h(size_t off) const { return reinterpret_cast<T *>(static_cast<char *>(host) + off); }
    template <class T> T *d(size_t off) const { return reinterpret_cast<T *>(static_cast<char *>(dev) + off); }
};

// Optional extras of the host-pointer route (all nullable): boards resident on the device between calls, 16-bit column
// scores on the wire (a column score is < 3n <= 24576 for n <= NQ_STEP_MAX_N) and a completion word in coherent host
// memory that the caller polls instead of synchronising the stream.
struct NqStepExtra {
    unsigned long long fresh_mask; // bit r set: restart r's board comes from a.rows; clear: from resident_rows (n_restarts <= 64)
    int *resident_rows;           // [n_restarts][n] device copy; receives the winner board
    unsigned *resident_tabs;      // nullable [n_restarts][tab_words] the count tables of the resident boards (packed u16, the kernel's LDS layout)
    int tab_words;                // stride of resident_tabs in 32-bit words
    unsigned short *next_cs16;    // [n_restarts][n]
    unsigned long long want_cs_mask; // bit r set: restart r's column scores are written to next_cs16 (r >= 64: always)
    unsigned long long *done_flag; // host word <- done_seq once every block's outputs are visible to the host
    unsigned long long done_seq;
    unsigned *block_counter;      // device word, zero between launches
};

// One-restart LS step: the per-call inputs that are not the board travel INSIDE the dispatch packet's kernel arguments
// (which the runtime places in device memory) instead of being read from pinned host memory by the kernel: with the
// board resident on the device the kernel then starts without a single PCIe read (3.3 us of its 11 us, measured with
// in-kernel stamps).
constexpr int NQ_STEP_INLINE_COLS = 224;
struct NqStepInline {
    long long cur;
    int n_cols;
    int valid; // 0: read cur / n_cols / cols through the NqArgs pointers as before
    int cols[NQ_STEP_INLINE_COLS];
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Boards the fused LS step keeps on the device between host-pointer calls (LS_NQ_STEP=3): `shadow` is the host's copy
// of what the device holds, so a call whose board equals it uploads nothing; the kernel leaves the winner board in
// `dev_rows` and the host applies the same move to the shadow.
struct NqResident {
    double t_pre = 0, t_launch = 0, t_wait = 0, t_post = 0; // LS_AMD_TRACE: where a resident LS-step call spends its time
    unsigned long long calls = 0;
    int n = 0;
    std::vector<int32_t> shadow; // [slots][n]
    std::vector<char> valid;     // [slots]
    std::vector<int32_t> cs;     // [slots][n] column scores of the shadow boards (what the last call returned)
    std::vector<char> cs_valid;  // [slots]
    DevBuf dev_rows;             // [slots][n]
    DevBuf dev_tabs;             // [slots][tab_words] count tables of the resident boards, patched by every step
    int tab_words = 0;
    DevBuf counter;              // completion-word block counter (one u32, zero between launches)
    unsigned long long *flag = nullptr; // hipHostMalloc'd coherent word the kernel publishes done_seq to
    unsigned long long seq = 0;
};

// The resident LS-step server (nq_server.hip): one never-ending launch per device whose workgroups are the slots.
struct NqServer;
struct SchedStepServer; // sched_server.hip: the resident scheduling LS step, one workgroup per slot
struct SchedGrid;       // sched_grid.hip: the resident scheduling grid, one sub-grid per restart
constexpr int SRV_RC_GIVE_UP = 1; // internal (never crosses the ABI): a resident launch could not serve the request, the caller takes the launched route

struct SchedInstance {
    bool valid = false;
    int n_days_scored = 0, n_days_stored = 0, start_weekday = 0, n_employees = 0;
    DevBuf holiday_bits; // uint64[n_employees]
    DevBuf lane_tab;     // SchedLaneConst[64]
};

} // namespace lsamd

struct ls_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int nq_algo = LS_NQ_ALGO_AUTO;
    int nq_step_mode = 4; // LS_NQ_STEP: 4 (default) = host calls of up to 64 restarts go to the resident LS-step server (nq_server.hip): no launch per
                          // call; 3 = one fused LS-step launch per call (boards resident on the device, polled completion word): the launched fallback
    lsamd::NqServer *server = nullptr; // the server this context's one-restart LS steps go to (own: started on demand; or a slot of a shared one)
    int server_slot = 0;
    bool server_owned = false;
    unsigned long long server_retired[4] = {0, 0, 0, 0}; // counters of this context's earlier LS-step servers (one per board size)
    unsigned long long n_done_fallbacks = 0;   // completion-word waits that fell back to hipStreamSynchronize (2 ms without the word)
    unsigned long long n_resident_uploads = 0; // LS_NQ_STEP=3: boards uploaded because they were not the one resident on the device
    // Argmin keys: every entry of nq_keys holds NQ_KEY_EMPTY whenever no N-queens call is in flight (filled once, in
    // stream order, when the buffer is allocated; the finalize kernel puts back what the eval kernels touched).
    // nq_keys_dirty is raised while a call is being queued and stays up if it failed half-way: the next call then
    // clears the keys in stream order first.
    bool nq_keys_dirty = false;
    bool (*idle_hook)(void *) = nullptr; // completion-word poll: work to do while waiting (returns false when none is left); set per call by the solver
    void *idle_arg = nullptr;
    void (*wait_hook)(void *) = nullptr; // completion-word poll: called between polls (ls_multi_solver: switch to another restart's fibre on this thread)
    void *wait_hook_arg = nullptr;
    // What the wait in progress waits for, for a scheduler behind wait_hook (ls_multi_solver's fibre pool resumes a restart only
    // once its answer is there): n_wait_words words whose upper 16 bits all become wait_tag; wait_words == nullptr: not known.
    const volatile unsigned *wait_words = nullptr;
    int n_wait_words = 0;
    unsigned wait_tag = 0;
    unsigned long long steps_done = 0; // one-restart LS steps of this context (ls_multi_solver: the running restart's scheduling priority)
    bool sched_delta = true;     // LS_SCHED_DELTA: lane-per-candidate delta scorer (0: the wavefront-per-candidate kernels)
    int nq_scan_variant = 0; // register-scan formulation: 0 auto (fp32), 1 packed u16 (LS_NQ_SCAN_VARIANT)
    int n_cus = 0;
    int wave_size = 0;
    int clock_khz = 0;
    char dev_name[256] = {0};
    mutable std::string err;

    // persistent workspaces (grow-only)
    lsamd::DevBuf nq_rows, nq_cur, nq_cols, nq_ncols, nq_results, nq_scores, nq_keys, nq_tabs,
        nq_colscores, nq_totals, nq_next_rows;
    lsamd::DevBuf sc_states, sc_moves, sc_nmoves, sc_results, sc_hard, sc_soft, sc_keys, sc_terms, sc_cand_keys;
    lsamd::SchedInstance sched;
    // ls_sched_moves_upload: the resident list lives on the host (the resident servers read it from pinned memory, once, and the
    // winner is applied to the shadow state from it); the device copy is made when a launched pipeline first needs it
    std::vector<ls_sched_move> sc_resident_host;
    unsigned long long sc_resident_gen = 0;      // bumped by every upload: what a server slot holds is compared with it
    lsamd::DevBuf sc_resident_moves;
    unsigned long long sc_resident_dev_gen = 0;  // generation of the device copy
    // the resident routes of this context's scheduling calls: a server of its own (started on demand), or one slot / sub-grid
    // of a launch shared with other contexts (ls_multi_solver_create_sched)
    lsamd::SchedStepServer *sched_step = nullptr;
    int sched_step_slot = 0;
    bool sched_step_owned = false;
    lsamd::SchedGrid *sched_grid = nullptr;
    int sched_grid_sub = 0;
    bool sched_grid_owned = false;
    bool sched_grid_unavailable = false;         // its workgroups could not all be resident (another launch holds the CUs): not tried again
    unsigned long long sched_retired[3] = {0, 0, 0}; // counters of this context's earlier servers
    bool sched_server_on = true;                 // LS_SCHED_SERVER=0: one launch per call
    lsamd::Staging stage_in, stage_out;
    lsamd::NqResident nq_res;
    long long sc_resident_n = 0;
};

namespace lsamd {

int set_err(const ls_ctx *ctx, int code, const char *fmt, ...);

// nq_server.hip
bool nq_server_supports(const ls_ctx *ctx, int n);
int nq_server_max_slots(ls_ctx *ctx, int n); // resident workgroups of the server the device holds at once (less headroom)
int nq_server_create(ls_ctx *ctx, int n, int n_slots, NqServer **out);
void nq_server_destroy(NqServer *s);
int nq_server_n(const NqServer *s);
int nq_server_slots(const NqServer *s);
void nq_server_counters(const NqServer *s, unsigned long long out[4]); // requests, launches, waits that found the launch drained, boards uploaded
int nq_server_step(ls_ctx *ctx, NqServer *s, int slot, const int32_t *rows, int64_t cur_score, const int32_t *cols, int32_t n_cols,
                   long long window, ls_nq_result *res, int64_t *next_col_scores);
int nq_server_col_scores(ls_ctx *ctx, NqServer *s, int slot, const int32_t *rows, int64_t *col_scores, int64_t *total);
// sched_server.hip: the resident scheduling LS step
int sched_step_max_slots(ls_ctx *ctx);
int sched_step_create(ls_ctx *ctx, int n_slots, SchedStepServer **out);
void sched_step_destroy(SchedStepServer *s);
int sched_step_slots(const SchedStepServer *s);
void sched_step_pause(SchedStepServer *s);
void sched_step_counters(const SchedStepServer *s, unsigned long long out[3]); // requests, launches, waits that found the launch drained
int sched_step_score(ls_ctx *ctx, SchedStepServer *s, int slot, const int32_t *state, int64_t *hard, int64_t *soft);
int sched_step_eval(ls_ctx *ctx, SchedStepServer *s, int slot, const int32_t *state, long long n_moves, long long window, ls_sched_result *out);
// sched_grid.hip: the resident scheduling grid
bool sched_grid_fits(const ls_ctx *ctx, int n_sub, bool gen, long long n_moves);
int sched_grid_create(ls_ctx *ctx, int n_sub, bool gen, long long capacity, SchedGrid **out);
void sched_grid_destroy(SchedGrid *s);
int sched_grid_subs(const SchedGrid *s);
bool sched_grid_is_gen(const SchedGrid *s);
long long sched_grid_capacity(const SchedGrid *s);
void sched_grid_pause(SchedGrid *s);
void sched_grid_counters(const SchedGrid *s, unsigned long long out[3]);
int sched_grid_score(ls_ctx *ctx, SchedGrid *s, int sub, const int32_t *state, int64_t *hard, int64_t *soft);
int sched_grid_eval(ls_ctx *ctx, SchedGrid *s, int sub, const int32_t *state, long long n_moves, long long window, ls_sched_result *out);
// ls_api.hip
int ctx_clone_light(const ls_ctx *base, ls_ctx **out); // a context on the same device without a stream of its own yet (made on first use): what a slot of a shared resident launch needs
int nq_ctx_server(ls_ctx *ctx, int n, NqServer **out, int slots = 1);
int nq_server_post(ls_ctx *ctx, NqServer *s, int slot, const int32_t *rows, const int32_t *cols, int32_t n_cols, long long window,
                   bool has_next_col_scores);
int nq_server_finish(ls_ctx *ctx, NqServer *s, int slot, int64_t cur_score, const int32_t *cols, ls_nq_result *res, int64_t *next_col_scores);
int nq_ctx_prepare_server(ls_ctx *ctx, int n);
int sched_ctx_prepare(ls_ctx *ctx, long long want);

#define LS_HIP(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess)                                                              \
            return lsamd::set_err((ctx), LS_ERR_HIP, "%s failed: %s (%s:%d)", #call,        \
                                  hipGetErrorString(e__), __FILE__, __LINE__);              \
    } while (0)

#define LS_RESERVE(ctx, buf, bytes)                                                         \
    do {                                                                                    \
        hipError_t e__ = (buf).reserve(bytes);                                              \
        if (e__ != hipSuccess)                                                              \
            return lsamd::set_err((ctx), LS_ERR_NOMEM, "device allocation of %zu bytes failed: %s", \
                                  (size_t)(bytes), hipGetErrorString(e__));                 \
    } while (0)

#if defined(__HIPCC__) // (the host logic also compiles without a HIP compiler: tests/cpp/srv_host_test.cpp, against a fake runtime)
// one copy per translation unit that launches it
static __global__ void fill_i64_kernel(long long *p, long long value, long long count)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (long long)gridDim.x * blockDim.x)
        p[i] = value;
}

static __global__ void fill_i32_kernel(int *p, int value, long long count)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (long long)gridDim.x * blockDim.x)
        p[i] = value;
}

// ------------------------------------------------------------ wavefront utils
// DPP controls (gfx9 family, wave64)
#define LS_DPP_QUAD_1032 0xB1
#define LS_DPP_QUAD_2301 0x4E
#define LS_DPP_ROW_HALF_MIRROR 0x141
#define LS_DPP_ROW_MIRROR 0x140
#define LS_DPP_ROW_BCAST15 0x142
#define LS_DPP_ROW_BCAST31 0x143

// Sum over the 64 lanes of a wavefront; the total is returned wave-uniform
// (read from lane 63).  6 DPP adds, no LDS.
__device__ __forceinline__ int wave_sum_i32(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_QUAD_1032, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_QUAD_2301, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_ROW_HALF_MIRROR, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_ROW_MIRROR, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_ROW_BCAST15, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, LS_DPP_ROW_BCAST31, 0xc, 0xf, false);
    return __builtin_amdgcn_readlane(x, 63);
}

// Minimum over the 64 lanes of a wavefront, returned wave-uniform: the same six DPP steps as wave_sum_i32 (no LDS round trips:
// a __shfl_xor butterfly costs twelve dependent ds_bpermute operations); each step is one v_min_u32 with a DPP operand.
__device__ __forceinline__ unsigned wave_min_u32(unsigned x)
{
    // written out: hipcc expands the builtin form into v_mov + s_nop + v_mov_dpp + v_min per step.  (s_nop 1: a DPP operand may
    // not be read for two wait states after a VALU wrote it, and nothing inserts them inside an asm statement.)  Lanes a step
    // does not address (row_mask, a broadcast's missing source) keep their value.
    asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 0"
                 : "+v"(x));
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

// 64-bit minimum as two 32-bit ones: the high words, then the low words of the lanes that hold the smallest high word
// (twelve single-instruction steps instead of six steps of two DPP moves + a 64-bit compare + two selects)
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long x)
{
    const unsigned hi = (unsigned)(x >> 32);
    const unsigned mh = wave_min_u32(hi);
    const unsigned ml = wave_min_u32(hi == mh ? (unsigned)x : 0xffffffffu);
    return ((unsigned long long)mh << 32) | ml;
}

// The same over lanes 0..7 only (one value per wavefront of a 512-thread block, the other lanes hold the identity): three steps;
// returned wave-uniform from lane 0.
__device__ __forceinline__ unsigned wave_min8_u32(unsigned x)
{
    asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0"
                 : "+v"(x));
    return (unsigned)__builtin_amdgcn_readlane((int)x, 0);
}
__device__ __forceinline__ unsigned long long wave_min8_u64(unsigned long long x)
{
    const unsigned hi = (unsigned)(x >> 32);
    const unsigned mh = wave_min8_u32(hi);
    const unsigned ml = wave_min8_u32(hi == mh ? (unsigned)x : 0xffffffffu);
    return ((unsigned long long)mh << 32) | ml;
}

#endif // __HIPCC__

} // namespace lsamd
