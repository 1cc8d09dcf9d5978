// p2hot.hip -- context, pass planning and the C ABI of libp2hot (see include/p2hot.h).
// One translation unit: the kernels live in ntt.hpp / merkle.hpp / fri.hpp.
#include "../../include/p2hot.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "fri.hpp"
#include "plonk.hpp"
#include "merkle.hpp"
#include "ntt.hpp"
#include "nttl.hpp"

using gl::u32;
using gl::u64;

// ------------------------------------------------------------------ context
struct p2hot_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    u64 alpha_stage[2] = {0, 0};   // host staging of an extension challenge (outlives the asynchronous copy)
    std::atomic<bool> busy{false};  // a host-pointer entry point is running on this context (see CallGuard)
    u64 *tables = nullptr;  // fwd_lo, fwd_hi, inv_lo, inv_hi (65536 each)
    unsigned *pinned_oob = nullptr;  // host-pinned landing word of d_oob (p2hot_ctx_sync)
    unsigned *d_oob = nullptr;  // raised by gathers that were handed an out-of-range row / leaf index (device-resident indices)
    ntt::RootTable fwd{}, inv{};
    u64 *local_fwd = nullptr, *local_inv = nullptr;  // [2^m + e] = w_{2^m}^(+-e), m <= TILE_LOG
    bool use_regpass = true;
    bool use_limb = true;  // 4096-element tiles run the carry-free limb passes (nttl.hpp); P2HOT_NTT_LIMB=0 selects the round-2 kernels
    bool force_no_limb = false;  // P2HOT_NTT_LIMB=0 is sticky: a later p2hot_tune_ntt(ctx, 3) does not switch the limb passes back on
    struct LimbTables {
        nttl::W2 *tw_all = nullptr;
        u64 *ufac = nullptr;
    };
    std::map<std::pair<int, unsigned>, LimbTables> limb_tw_cache;  // (inverse, log_r) -> the round tables of a 2^log_r-row tile
    std::set<const void *> lds_opted;  // kernels whose large dynamic LDS request was registered (lds_opt_in)
    size_t host_block_cols = 0;     // > 0: p2hot_commit uploads / transforms this many columns per block whatever the size (tests)
    size_t host_tail_min_leaves = (size_t)1 << 18;  // ... and the tail per group of cap subtrees of at least this many leaves (P2HOT_HOST_TAIL_MIN_LEAVES)
    bool host_leaves_first = true;  // p2hot_commit with leaves_out: transforms, then the leaf matrix's copy beside the sponge (P2HOT_HOST_LEAVES_FIRST)
    bool host_chunked_hash = true;  // p2hot_commit: the leaf sponge absorbs each block's columns as soon as they are extended
    bool pp_streams = true;       // partial products: pairs of challenges through pp_quotients2_kernel (P2HOT_PP_STREAMS=0: one kernel per challenge)
    bool lde_reads_bitrev = true;  // from_values: fold the bit reversal between iNTT and LDE into the LDE's first pass (P2HOT_LDE_BITREV_SRC=0: off)
    bool limb_dual = true;       // P2HOT_LIMB_DUAL builds: pair tile groups into 1024-thread workgroups (P2HOT_LIMB_DUAL_OFF=1 turns it off)
    unsigned limb_tiles_log = 4;  // contiguous limb passes: a workgroup stages its tables once for 2^this tiles (P2HOT_LIMB_TILES_LOG)
    // second stream: the VALU-bound leaf sponge of coset block b runs beside the wait-bound NTT of block b+1
    hipStream_t side = nullptr;
    hipStream_t leaf_stream = nullptr;  // P2HOT_LEAVES_ASYNC: the leaf matrix's row blocks travel here, beside everything else (created on first use)
    hipStream_t xform_stream = nullptr; // P2HOT_LEAVES_ASYNC, several column blocks: the transforms + transposition lane beside the sponge (created on first use)
    bool host_async_split = true;       // ... (P2HOT_HOST_ASYNC_SPLIT=0: the single-stream leaves-first order)
    size_t host_async_early_blocks = (size_t)-1; // ... of the 64 leaf blocks, how many leave before the digests' copy is queued (P2HOT_HOST_ASYNC_EARLY_BLOCKS; unset: 0 beside the split lanes, a quarter otherwise)
    std::vector<hipEvent_t> fork_events;
    hipEvent_t join_event = nullptr;
    bool overlap = false;
    size_t quad_threshold = (size_t)1 << 15;  // launches with at most this many permutations use the quad kernels
    size_t row_threshold = (size_t)1 << 13;   // ... and with at most this many the word-per-lane kernels (16 lanes per permutation)
    unsigned ntt_radix_bits = 3;  // 3: radix-8 rounds / 512 threads, 4: radix-16 / 256 threads
    unsigned ntt_strided_bits = 10;  // most bits a strided pass takes (tile = 2^b rows x 2^(12-b) columns)
    bool ntt_xcd_remap = true;       // strided passes: neighbouring column groups (same 128-byte lines) on the same XCD
    // small device-to-host results of a host-pointer call land in a pinned arena (a truly asynchronous copy) and are moved
    // to the caller's pageable buffers when the call synchronises: a copy into pageable memory blocks the host ~20 us
    struct DeferredCopy {
        void *dst;
        const unsigned char *src;
        size_t width, height, dpitch;
    };
    unsigned char *pinned = nullptr;
    size_t pinned_cap = 0, pinned_used = 0;
    unsigned char *pinned_up = nullptr;  // upload staging: many short host columns -> one contiguous pinned block -> one copy
    size_t pinned_up_cap = 0;
    bool pinned_up_busy = false;  // an asynchronous copy may still be reading the staging block (set by every staged upload)
    hipEvent_t stage_ev[2] = {nullptr, nullptr};  // one per half of the staging block when an upload takes several slices
    std::vector<DeferredCopy> deferred;
    bool in_host_call = false;       // set by the host-pointer entry points (they end in stream_sync)
    size_t horner_two_level_min = 4096;  // divide_by_linear: more chunks than this -> carries in two levels (P2HOT_HORNER_2L_MIN)
    size_t zloop_min_groups = 2048;  // first LDE pass: one workgroup loops over the cosets when the launch has this many without
    struct Scratch {
        void *p = nullptr;
        size_t cap = 0;
    } scratch[4];  // 0: NTT temporary, 1: final_poly, 2: FRI commit phase, 3: all-gather staging of the sharded commit (grow-only, reused across calls)
    // coset scale tables keyed by (log_n, rate_bits, shift, first block, block count, first-pass log_r)
    std::map<std::tuple<unsigned, unsigned, u64, size_t, size_t, unsigned>, u64 *> scale_cache;
    std::map<std::tuple<int, unsigned, unsigned>, u64 *> twid_cache;
    // grow-only cache of device blocks for the host-pointer entry points: a fresh hipMalloc of the 9 GB LDE matrix
    // costs up to a second (the driver clears VRAM), so blocks go back to this list instead of hipFree
    std::vector<p2hot_ctx *> helpers;   // p2hot_prove_openings_many: sibling contexts on the same GPU, each with its own stream
    std::vector<hipStream_t> helper_streams;
    std::vector<struct p2hot_challenger *> helper_challengers;
    std::mutex pool_mu;  // pool_free / pool_live / scratch bookkeeping: frees may come from another thread (a Drop, a GC finaliser)
    std::vector<std::pair<void *, size_t>> pool_free;  // (pointer, capacity)
    std::map<void *, size_t> pool_live;  // (inverse, log_nblk, log_r) -> inter-pass twiddle table
    // the same for PINNED host blocks handed to the caller (p2hot_host_alloc): the flat leaf matrix behind MerkleTree::get
    std::vector<std::pair<void *, size_t>> host_pool_free;
    std::map<void *, size_t> host_pool_live;
    // live per-kernel timing (HIP events on the launch stream), off by default
    bool profiling = false;
    struct ProfRec {
        const char *name;
        hipEvent_t e0, e1;
    };
    std::vector<ProfRec> prof;
    std::map<std::string, std::pair<double, unsigned long long>> prof_acc;  // name -> (ms, launches)
    std::string prof_text;
};

// Makes the context's GPU the calling thread's current device for the scope (allocations, pinned memory and launches follow the
// current device) and restores the caller's on exit.  Every entry point that takes a context holds one: after a p2hot_group_*
// call -- which walks over the ranks' devices -- a single-context call on p2hot_group_ctx(group, r) must not inherit the last
// rank's device.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const p2hot_ctx *ctx) {
        if (!ctx) return;
#ifdef P2HOT_EMU
        if (emu::fault_no_device_guard) return;  // test hook of the emulator build: what a missing guard looks like (tests/test_emu_devices.py)
#endif
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != ctx->device) switched = hipSetDevice(ctx->device) == hipSuccess;
    }
    explicit DeviceGuard(int keep_current) { (void)keep_current; if (hipGetDevice(&prev) == hipSuccess) switched = true; }  // restore only
    ~DeviceGuard() {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// Brackets the launches of one kernel family with events when profiling is on.
struct ProfScope {
    p2hot_ctx *ctx;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const char *name;
    hipStream_t stream;
    ProfScope(p2hot_ctx *c, const char *n, hipStream_t st = nullptr, bool use_st = false)
        : ctx(c), name(n), stream(use_st ? st : c->stream) {
        if (!ctx->profiling) return;
        // events belong to the device that is current when they are made and are recorded on a stream of that device: the
        // multi-GPU loops construct and leave this scope with ANOTHER rank's device current (found by the emulated node,
        // tests/test_emu_devices.py: hipEventRecord on rank 0's communication stream while the last rank's device was current)
        DeviceGuard dev_guard_(ctx);
        if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) (void)hipEventRecord(e0, stream);
    }
    ~ProfScope() {
        if (e0 && e1) {
            DeviceGuard dev_guard_(ctx);
            (void)hipEventRecord(e1, stream);
            ctx->prof.push_back({name, e0, e1});
        }
    }
};

#define P2_FAIL(ctx, code, ...)                           \
    do {                                                  \
        char buf_[512];                                   \
        snprintf(buf_, sizeof buf_, __VA_ARGS__);         \
        (ctx)->err = buf_;                                \
        return (code);                                    \
    } while (0)

#define P2_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            P2_FAIL(ctx, (e_ == hipErrorOutOfMemory ? P2HOT_ENOMEM : P2HOT_EHIP), "%s: %s", #call, \
                    hipGetErrorString(e_));                                                       \
    } while (0)

#define P2_TRY(expr)                   \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != P2HOT_OK) return rc_; \
    } while (0)

#define P2_LAUNCH_CHECK(ctx) P2_HIP(ctx, hipGetLastError())

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// ---- small device-to-host results (see p2hot_ctx::deferred)
static void flush_deferred(p2hot_ctx *ctx) {
    for (auto &d : ctx->deferred)
        for (size_t r = 0; r < d.height; ++r) std::copy(d.src + r * d.width, d.src + (r + 1) * d.width, (unsigned char *)d.dst + r * d.dpitch);
    ctx->deferred.clear();
    ctx->pinned_used = 0;
}

// hipStreamSynchronize of the context's stream + delivery of the deferred results; every synchronisation of the context goes
// through here
static hipError_t stream_sync(p2hot_ctx *ctx) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) {
        flush_deferred(ctx);
    } else {
        ctx->deferred.clear();
        ctx->pinned_used = 0;
    }
    return e;
}

// `height` rows of `width` bytes, device pitch `spitch`, host pitch `dpitch`
static int d2h_2d(p2hot_ctx *ctx, void *dst, size_t dpitch, const void *d_src, size_t spitch, size_t width, size_t height) {
    const size_t bytes = width * height, kArena = (size_t)4 << 20;
    if (bytes == 0) return P2HOT_OK;
    if (ctx->in_host_call && bytes <= ((size_t)256 << 10)) {
        if (!ctx->pinned && hipHostMalloc((void **)&ctx->pinned, kArena, hipHostMallocDefault) == hipSuccess) ctx->pinned_cap = kArena;
        const size_t at = (ctx->pinned_used + 63) & ~(size_t)63;
        if (ctx->pinned && at + bytes <= ctx->pinned_cap) {
            unsigned char *slot = ctx->pinned + at;
            if (height == 1 || spitch == width)
                P2_HIP(ctx, hipMemcpyAsync(slot, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
            else
                P2_HIP(ctx, hipMemcpy2DAsync(slot, width, d_src, spitch, width, height, hipMemcpyDeviceToHost, ctx->stream));
            ctx->pinned_used = at + bytes;
            ctx->deferred.push_back({dst, slot, width, height, dpitch});
            return P2HOT_OK;
        }
    }
    if (height == 1 || (spitch == width && dpitch == width))
        P2_HIP(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    else
        P2_HIP(ctx, hipMemcpy2DAsync(dst, dpitch, d_src, spitch, width, height, hipMemcpyDeviceToHost, ctx->stream));
    return P2HOT_OK;
}
static int d2h(p2hot_ctx *ctx, void *dst, const void *d_src, size_t bytes) { return d2h_2d(ctx, dst, bytes, d_src, bytes, bytes, 1); }

// `count` host columns of `bytes` each -> d_dst (contiguous, column after column).  Short columns are gathered into the pinned
// upload block at byte offset `stage_off` and go up in one asynchronous copy (a copy from pageable memory costs ~15 us of host
// time whatever its size: 135 columns of a recursion-size trace were 1.5 ms); long ones are copied one by one as before.
// The staging block is only reused after the call's synchronisation, so distinct blocks of one call use distinct offsets.
static int h2d_columns(p2hot_ctx *ctx, void *d_dst, const uint64_t *const *cols, size_t count, size_t bytes, size_t stage_off,
                       size_t stage_total, hipStream_t stream) {
    const size_t kMaxStage = (size_t)64 << 20;
    if (ctx->in_host_call && count > 1 && bytes <= ((size_t)1 << 20) && stage_total <= kMaxStage) {
        if (ctx->pinned_up_cap < stage_total) {
            // growing the block: no copy staged earlier (this call or a previous one, either stream) may still be reading it
            if (ctx->pinned_up && ctx->pinned_up_busy) (void)hipDeviceSynchronize();
            ctx->pinned_up_busy = false;
            if (ctx->pinned_up) (void)hipHostFree(ctx->pinned_up);
            ctx->pinned_up = nullptr;
            ctx->pinned_up_cap = 0;
            const size_t want = std::max(stage_total, (size_t)8 << 20);
            if (hipHostMalloc((void **)&ctx->pinned_up, want, hipHostMallocDefault) == hipSuccess) ctx->pinned_up_cap = want;
        }
        if (ctx->pinned_up && stage_off + count * bytes <= ctx->pinned_up_cap) {
            unsigned char *slot = ctx->pinned_up + stage_off;
            for (size_t c = 0; c < count; ++c)
                std::copy((const unsigned char *)cols[c], (const unsigned char *)cols[c] + bytes, slot + c * bytes);
            ctx->pinned_up_busy = true;
            P2_HIP(ctx, hipMemcpyAsync(d_dst, slot, count * bytes, hipMemcpyHostToDevice, stream));
            return P2HOT_OK;
        }
    }
    // More short columns than the staging block holds (p2hot_commit_many of dozens of recursion-size proofs: thousands of 32 KB
    // vectors, each ~10 us of host time as a pageable copy): the block's two halves take slices in turn, a half being refilled
    // once the copy that read it has finished.
    // (only where the per-copy overhead outweighs the copy itself: beyond ~128 KB a pageable copy moves faster than one core stages it)
    if (ctx->in_host_call && count > 1 && bytes <= ((size_t)128 << 10) && stage_off == 0 && stage_total == count * bytes) {
        if (ctx->pinned_up_cap < kMaxStage) {
            if (ctx->pinned_up && ctx->pinned_up_busy) (void)hipDeviceSynchronize();
            ctx->pinned_up_busy = false;
            if (ctx->pinned_up) (void)hipHostFree(ctx->pinned_up);
            ctx->pinned_up = nullptr;
            ctx->pinned_up_cap = 0;
            if (hipHostMalloc((void **)&ctx->pinned_up, kMaxStage, hipHostMallocDefault) == hipSuccess) ctx->pinned_up_cap = kMaxStage;
        }
        for (int k = 0; k < 2 && ctx->pinned_up; ++k)
            if (!ctx->stage_ev[k] && hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming) != hipSuccess) ctx->stage_ev[k] = nullptr;
        if (ctx->pinned_up && ctx->stage_ev[0] && ctx->stage_ev[1]) {
            const size_t half = ctx->pinned_up_cap / 2, per = half / bytes;
            size_t k = 0;
            for (size_t c0 = 0; c0 < count; c0 += per, ++k) {
                const size_t cnt = std::min(per, count - c0);
                unsigned char *slot = ctx->pinned_up + (k & 1) * half;
                if (k >= 2) P2_HIP(ctx, hipEventSynchronize(ctx->stage_ev[k & 1]));
                for (size_t c = 0; c < cnt; ++c)
                    std::copy((const unsigned char *)cols[c0 + c], (const unsigned char *)cols[c0 + c] + bytes, slot + c * bytes);
                ctx->pinned_up_busy = true;
                P2_HIP(ctx, hipMemcpyAsync((unsigned char *)d_dst + c0 * bytes, slot, cnt * bytes, hipMemcpyHostToDevice, stream));
                P2_HIP(ctx, hipEventRecord(ctx->stage_ev[k & 1], stream));
            }
            return P2HOT_OK;
        }
    }
    for (size_t c = 0; c < count; ++c)
        P2_HIP(ctx, hipMemcpyAsync((unsigned char *)d_dst + c * bytes, cols[c], bytes, hipMemcpyHostToDevice, stream));
    return P2HOT_OK;
}

static int scratch_get(p2hot_ctx *ctx, int slot, size_t bytes, void **out) {
    DeviceGuard dev_guard_(ctx);
    std::lock_guard<std::mutex> pool_lock_(ctx->pool_mu);
    auto &s = ctx->scratch[slot];
    if (s.cap < bytes) {
        if (s.p) {
            P2_HIP(ctx, stream_sync(ctx));
            P2_HIP(ctx, hipFree(s.p));
            s.p = nullptr;
            s.cap = 0;
        }
        P2_HIP(ctx, hipMalloc(&s.p, bytes));
        s.cap = bytes;
    }
    *out = s.p;
    return P2HOT_OK;
}

// block cache of the host-pointer entry points (see p2hot_ctx::pool_free)
static int pool_alloc(p2hot_ctx *ctx, size_t bytes, void **out) {
    DeviceGuard dev_guard_(ctx);
    std::lock_guard<std::mutex> pool_lock_(ctx->pool_mu);
    if (bytes == 0) bytes = 8;
    size_t best = ctx->pool_free.size();
    for (size_t k = 0; k < ctx->pool_free.size(); ++k) {
        const size_t cap = ctx->pool_free[k].second;
        if (cap >= bytes && cap <= 2 * bytes + ((size_t)1 << 20) && (best == ctx->pool_free.size() || cap < ctx->pool_free[best].second))
            best = k;
    }
    if (best != ctx->pool_free.size()) {
        *out = ctx->pool_free[best].first;
        ctx->pool_live[*out] = ctx->pool_free[best].second;
        ctx->pool_free.erase(ctx->pool_free.begin() + best);
        return P2HOT_OK;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {  // give the cached blocks back and try once more
        (void)hipGetLastError();
        for (auto &b : ctx->pool_free) (void)hipFree(b.first);
        ctx->pool_free.clear();
        P2_HIP(ctx, hipMalloc(&p, bytes));
    }
    ctx->pool_live[p] = bytes;
    *out = p;
    return P2HOT_OK;
}
static void pool_release(p2hot_ctx *ctx, void *p) {
    DeviceGuard dev_guard_(ctx);
    std::lock_guard<std::mutex> pool_lock_(ctx->pool_mu);
    if (!p) return;
    auto it = ctx->pool_live.find(p);
    if (it == ctx->pool_live.end()) return;
    ctx->pool_free.emplace_back(p, it->second);
    ctx->pool_live.erase(it);
}
// ---- pinned host blocks for the caller (include/p2hot.h, p2hot_host_alloc)
extern "C" int p2hot_host_alloc(p2hot_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return P2HOT_EINVAL;
    *out = nullptr;
    DeviceGuard dev_guard_(ctx);
    std::lock_guard<std::mutex> pool_lock_(ctx->pool_mu);
    if (bytes == 0) bytes = 8;
    size_t best = ctx->host_pool_free.size();
    for (size_t k = 0; k < ctx->host_pool_free.size(); ++k) {
        const size_t cap = ctx->host_pool_free[k].second;
        if (cap >= bytes && cap <= 2 * bytes + ((size_t)1 << 20) && (best == ctx->host_pool_free.size() || cap < ctx->host_pool_free[best].second)) best = k;
    }
    if (best != ctx->host_pool_free.size()) {
        *out = ctx->host_pool_free[best].first;
        ctx->host_pool_live[*out] = ctx->host_pool_free[best].second;
        ctx->host_pool_free.erase(ctx->host_pool_free.begin() + best);
        return P2HOT_OK;
    }
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {  // give the cached blocks back and try once more
        (void)hipGetLastError();
        for (auto &b : ctx->host_pool_free) (void)hipHostFree(b.first);
        ctx->host_pool_free.clear();
        e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) P2_FAIL(ctx, P2HOT_ENOMEM, "host_alloc: %zu pinned bytes: %s", bytes, hipGetErrorString(e));
    }
    ctx->host_pool_live[p] = bytes;
    *out = p;
    return P2HOT_OK;
}
extern "C" void p2hot_host_free(p2hot_ctx *ctx, void *p) {
    if (!ctx || !p) return;
    std::lock_guard<std::mutex> pool_lock_(ctx->pool_mu);
    auto it = ctx->host_pool_live.find(p);
    if (it == ctx->host_pool_live.end()) return;  // not ours (or freed twice): left alone
    ctx->host_pool_free.emplace_back(p, it->second);
    ctx->host_pool_live.erase(it);
}

namespace {
struct PoolBuf {  // returns its block to the context's cache on scope exit (the owner syncs the stream first)
    p2hot_ctx *ctx;
    void *p = nullptr;
    explicit PoolBuf(p2hot_ctx *c) : ctx(c) {}
    PoolBuf(const PoolBuf &) = delete;
    ~PoolBuf() { pool_release(ctx, p); }
    u64 *u() const { return (u64 *)p; }
};
}  // namespace

extern "C" const char *p2hot_version(void) { return "p2hot 0.1 (gfx950)"; }
extern "C" int p2hot_is_emulated(void) {
#ifdef P2HOT_EMU
    return 1;
#else
    return 0;
#endif
}

extern "C" size_t p2hot_num_digests(unsigned log_leaves, unsigned cap_height) {
    if (cap_height > log_leaves) return 0;
    return 2 * (((size_t)1 << log_leaves) - ((size_t)1 << cap_height));
}

extern "C" int p2hot_ctx_create(int device, void *hip_stream, p2hot_ctx **out) {
    if (!out) return P2HOT_EINVAL;
    p2hot_ctx *ctx = new p2hot_ctx();
    ctx->device = device;
    if (const char *e = getenv("P2HOT_NTT_STRIDED_BITS")) {  // pass-planning experiments (tools/tune_strided.sh)
        const int b = atoi(e);
        if (b >= 6 && b <= 11) ctx->ntt_strided_bits = (unsigned)b;
    }
    if (const char *e = getenv("P2HOT_NTT_XCD_REMAP")) ctx->ntt_xcd_remap = atoi(e) != 0;
    if (const char *e = getenv("P2HOT_NTT_LIMB")) ctx->use_limb = atoi(e) != 0, ctx->force_no_limb = atoi(e) == 0;
    if (const char *e = getenv("P2HOT_LIMB_TILES_LOG")) ctx->limb_tiles_log = (unsigned)atoi(e);
    if (const char *e = getenv("P2HOT_LIMB_DUAL_OFF")) ctx->limb_dual = !(e[0] == '1');
    if (const char *e = getenv("P2HOT_LDE_BITREV_SRC")) ctx->lde_reads_bitrev = !(e[0] == '0');
    if (const char *e = getenv("P2HOT_NTT_ZLOOP_MIN")) ctx->zloop_min_groups = (size_t)strtoull(e, nullptr, 10);
    if (const char *e = getenv("P2HOT_HORNER_2L_MIN")) ctx->horner_two_level_min = (size_t)strtoull(e, nullptr, 10);
    if (const char *e = getenv("P2HOT_HOST_BLOCK_COLS")) ctx->host_block_cols = (size_t)strtoull(e, nullptr, 10);
    if (const char *e = getenv("P2HOT_PP_STREAMS")) ctx->pp_streams = atoi(e) != 0;
    if (const char *e = getenv("P2HOT_HOST_CHUNKED_HASH")) ctx->host_chunked_hash = atoi(e) != 0;
    if (const char *e = getenv("P2HOT_HOST_LEAVES_FIRST")) ctx->host_leaves_first = atoi(e) != 0;
    if (const char *e = getenv("P2HOT_HOST_ASYNC_SPLIT")) ctx->host_async_split = atoi(e) != 0;
    if (const char *e = getenv("P2HOT_HOST_ASYNC_EARLY_BLOCKS")) ctx->host_async_early_blocks = (size_t)strtoull(e, nullptr, 10);
    if (const char *e = getenv("P2HOT_HOST_TAIL_MIN_LEAVES")) ctx->host_tail_min_leaves = (size_t)strtoull(e, nullptr, 10);
    // start values of p2hot_tune_quad / p2hot_tune_row for every context of the process, the ones p2hot_group_create makes
    // included (the kernel emulator's test tier lowers them: emulated cross-lane exchanges are slow)
    if (const char *e = getenv("P2HOT_TUNE_QUAD")) ctx->quad_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char *e = getenv("P2HOT_TUNE_ROW")) ctx->row_threshold = (size_t)strtoull(e, nullptr, 10);
    *out = ctx;  // returned even on failure so the caller can read last_error, then destroy
    P2_HIP(ctx, hipSetDevice(device));
    ctx->stream = (hipStream_t)hip_stream;  // NULL = the legacy default stream
    P2_HIP(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
    P2_HIP(ctx, hipEventCreateWithFlags(&ctx->join_event, hipEventDisableTiming));
    P2_HIP(ctx, hipMalloc((void **)&ctx->tables, (4 * 65536 + 4 * (1u << ntt::TILE_LOG) + 1) * sizeof(u64)));
    ctx->d_oob = (unsigned *)(ctx->tables + 4 * 65536 + 4 * (1u << ntt::TILE_LOG));
    P2_HIP(ctx, hipMemsetAsync(ctx->d_oob, 0, 8, ctx->stream));
    u64 *t = ctx->tables;
    const u64 w = gl::ROOT_2_32, wi = gl::inv(gl::ROOT_2_32);
    P2HOT_LAUNCH(ntt::pow_table_kernel, dim3(256), dim3(256), 0, ctx->stream, t, (size_t)65536, w, (u64)1, (u64)0);
    P2HOT_LAUNCH(ntt::pow_table_kernel, dim3(256), dim3(256), 0, ctx->stream, t + 65536, (size_t)65536, w,
                 (u64)65536, (u64)0);
    P2HOT_LAUNCH(ntt::pow_table_kernel, dim3(256), dim3(256), 0, ctx->stream, t + 2 * 65536, (size_t)65536, wi,
                 (u64)1, (u64)0);
    P2HOT_LAUNCH(ntt::pow_table_kernel, dim3(256), dim3(256), 0, ctx->stream, t + 3 * 65536, (size_t)65536, wi,
                 (u64)65536, (u64)0);
    P2_LAUNCH_CHECK(ctx);
    ctx->fwd = ntt::RootTable{t, t + 65536};
    ctx->inv = ntt::RootTable{t + 2 * 65536, t + 3 * 65536};
    ctx->local_fwd = t + 4 * 65536;
    ctx->local_inv = ctx->local_fwd + 2 * (1u << ntt::TILE_LOG);
    for (unsigned m = 0; m <= ntt::TILE_LOG; ++m) {
        const size_t cnt = (size_t)1 << m;
        const u64 wm = gl::root_of_unity(m), wmi = gl::inv(wm);
        P2HOT_LAUNCH(ntt::pow_table_kernel, dim3(cdiv(cnt, 256)), dim3(256), 0, ctx->stream, ctx->local_fwd + cnt, cnt, wm,
                     (u64)1, (u64)0);
        P2HOT_LAUNCH(ntt::pow_table_kernel, dim3(cdiv(cnt, 256)), dim3(256), 0, ctx->stream, ctx->local_inv + cnt, cnt, wmi,
                     (u64)1, (u64)0);
    }
    P2_LAUNCH_CHECK(ctx);
    P2_HIP(ctx, stream_sync(ctx));
    return P2HOT_OK;
}

extern "C" void p2hot_challenger_destroy(struct p2hot_challenger *ch);
extern "C" void p2hot_ctx_destroy(p2hot_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)stream_sync(ctx);
    for (size_t k = 0; k < ctx->helpers.size(); ++k) {
        if (k < ctx->helper_challengers.size()) p2hot_challenger_destroy(ctx->helper_challengers[k]);
        p2hot_ctx_destroy(ctx->helpers[k]);
        if (k < ctx->helper_streams.size() && ctx->helper_streams[k]) (void)hipStreamDestroy(ctx->helper_streams[k]);
    }
    ctx->helpers.clear();
    if (ctx->side) {
        (void)hipStreamSynchronize(ctx->side);
        (void)hipStreamDestroy(ctx->side);
    }
    if (ctx->leaf_stream) {
        (void)hipStreamSynchronize(ctx->leaf_stream);
        (void)hipStreamDestroy(ctx->leaf_stream);
    }
    if (ctx->xform_stream) {
        (void)hipStreamSynchronize(ctx->xform_stream);
        (void)hipStreamDestroy(ctx->xform_stream);
    }
    for (auto e : ctx->fork_events) (void)hipEventDestroy(e);
    if (ctx->join_event) (void)hipEventDestroy(ctx->join_event);
    for (auto &kv : ctx->scale_cache) (void)hipFree(kv.second);
    for (auto &kv : ctx->twid_cache) (void)hipFree(kv.second);
    for (auto &kv : ctx->limb_tw_cache) {
        (void)hipFree(kv.second.tw_all);
        if (kv.second.ufac) (void)hipFree(kv.second.ufac);
    }
    for (auto &b : ctx->pool_free) (void)hipFree(b.first);
    for (auto &kv : ctx->pool_live) (void)hipFree(kv.first);
    for (auto &b : ctx->host_pool_free) (void)hipHostFree(b.first);
    for (auto &kv : ctx->host_pool_live) (void)hipHostFree(kv.first);
    for (auto &s : ctx->scratch)
        if (s.p) (void)hipFree(s.p);
    if (ctx->tables) (void)hipFree(ctx->tables);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->pinned_oob) (void)hipHostFree(ctx->pinned_oob);
    for (int k = 0; k < 2; ++k)
        if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]);
    if (ctx->pinned_up) (void)hipHostFree(ctx->pinned_up);
    delete ctx;
}

extern "C" int p2hot_ctx_set_stream(p2hot_ctx *ctx, void *hip_stream) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    P2_HIP(ctx, stream_sync(ctx));
    ctx->stream = (hipStream_t)hip_stream;
    return P2HOT_OK;
}

// p2hot_ctx_sync: ONE synchronisation -- the out-of-range flag of the device-index gathers travels to a pinned word by an
// asynchronous copy enqueued in front of it
extern "C" int p2hot_ctx_sync(p2hot_ctx *ctx) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    unsigned f = 0;
    unsigned *slot = &f;
    if (!ctx->pinned_oob && hipHostMalloc((void **)&ctx->pinned_oob, 64, hipHostMallocDefault) != hipSuccess) ctx->pinned_oob = nullptr;
    if (ctx->pinned_oob) slot = ctx->pinned_oob;
    P2_HIP(ctx, hipMemcpyAsync(slot, ctx->d_oob, 4, hipMemcpyDeviceToHost, ctx->stream));
    P2_HIP(ctx, stream_sync(ctx));
    f = *slot;
    if (f) {
        P2_HIP(ctx, hipMemsetAsync(ctx->d_oob, 0, 4, ctx->stream));
        P2_FAIL(ctx, P2HOT_EINVAL, "an earlier %s was given an index out of range (the reference panics on the slice index); its output rows are zero",
                (f & 1) ? "p2hot_gather_rows_dev" : "p2hot_merkle_paths_dev");
    }
    return P2HOT_OK;
}

extern "C" const char *p2hot_last_error(const p2hot_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// tuning knob: launches with at most `max_perms` leaves / nodes use the quad-cooperative Poseidon kernels (0 = never)
extern "C" int p2hot_tune_quad(p2hot_ctx *ctx, size_t max_perms) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    ctx->quad_threshold = max_perms;
    return P2HOT_OK;
}

// tuning knob: launches with at most `max_perms` permutations use the word-per-lane Poseidon kernels (0 = never)
extern "C" int p2hot_tune_row(p2hot_ctx *ctx, size_t max_perms) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    ctx->row_threshold = max_perms;
    return P2HOT_OK;
}

// tuning knob: overlap the leaf sponge of coset block b with the LDE of block b+1 on a second stream (default on)
extern "C" int p2hot_tune_overlap(p2hot_ctx *ctx, int on) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    ctx->overlap = on != 0;
    return P2HOT_OK;
}

// tuning knob (not part of the drop-in surface): 0 = LDS radix-2 layers, 3 = register radix 8, 4 = radix 16
extern "C" int p2hot_tune_ntt(p2hot_ctx *ctx, int radix_bits) {
    if (!ctx || (radix_bits != 0 && radix_bits != 3 && radix_bits != 4 && radix_bits != 8)) return P2HOT_EINVAL;
    ctx->use_regpass = radix_bits != 0;
    // 8: radix 8 on 64-bit words (the round-2 kernels), 3: radix 8 on 24-bit limbs -- unless the environment opted out of them
    ctx->use_limb = radix_bits == 3 && !ctx->force_no_limb;
    if (radix_bits) ctx->ntt_radix_bits = radix_bits == 8 ? 3u : (unsigned)radix_bits;
    return P2HOT_OK;
}

extern "C" int p2hot_profile_enable(p2hot_ctx *ctx, int on) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    ctx->profiling = on != 0;
    return P2HOT_OK;
}

// Drains the recorded events into the accumulators and returns them as one JSON object
// {"kernel": {"ms": total, "launches": count}, ...}; reset != 0 clears the accumulators afterwards.
extern "C" const char *p2hot_profile_json(p2hot_ctx *ctx, int reset) {
    if (!ctx) return "{}";
    DeviceGuard dev_guard_(ctx);
    (void)stream_sync(ctx);
    for (auto &r : ctx->prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            auto &acc = ctx->prof_acc[r.name];
            acc.first += ms;
            acc.second += 1;
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    ctx->prof.clear();
    std::string t = "{";
    bool first = true;
    for (auto &kv : ctx->prof_acc) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s\"%s\": {\"ms\": %.6f, \"launches\": %llu}", first ? "" : ", ", kv.first.c_str(),
                 kv.second.first, kv.second.second);
        t += buf;
        first = false;
    }
    t += "}";
    ctx->prof_text = t;
    if (reset) ctx->prof_acc.clear();
    return ctx->prof_text.c_str();
}

// ------------------------------------------------------------------ NTT pass planning
struct Pass {
    unsigned log_r, log_c;
};

static std::vector<Pass> plan_passes(unsigned log_n, unsigned maxb) {
    std::vector<Pass> p;
    if (log_n <= ntt::TILE_LOG) {
        p.push_back({log_n, 0});
        return p;
    }
    // strided passes take up to 10 bits each: a tile is then 1024 rows x 4 columns, i.e. 32-byte row segments.  Narrow
    // segments cost HBM efficiency (an 11-bit pass with 16-byte segments measured 1.6x slower than two passes), but the
    // passes are ALU-bound and the workgroups that share a 128-byte line run on the same XCD (xcd_remap), so 2^21 and 2^22
    // -- the per-GPU transforms of 2- and 4-GPU C3 jobs -- take two passes instead of three (2^22: 20.1 ms instead of
    // 2 x 11.6 per strided step; without the remap 28.0: profiles/r02_e_tune_strided.txt).  2^23 stays at three passes: a
    // 2^13-element contiguous tile (64 KiB of LDS, five radix rounds, half the waves per CU) was built and measured -- its
    // pass costs 35.4 ms instead of 29.2 and the single 10-bit strided pass 40.2 instead of 2 x 23.8: 559 ms either way.
    unsigned rem = log_n - ntt::TILE_LOG;
    unsigned k = (rem + maxb - 1) / maxb;
    for (unsigned i = 0; i < k; ++i) {
        unsigned part = rem / (k - i);
        if (rem % (k - i)) ++part;
        p.push_back({part, ntt::TILE_LOG - part});
        rem -= part;
    }
    p.push_back({ntt::TILE_LOG, 0});
    return p;
}


// ------------------------------------------------------------------ limb passes (nttl.hpp)
// the round tables of a 2^log_r-row tile, concatenated in the layout the kernel copies to LDS (nttl::round_table_off)
static int limb_tables(p2hot_ctx *ctx, bool inverse, unsigned log_r, p2hot_ctx::LimbTables *out) {
    auto key = std::make_pair((int)inverse, log_r);
    auto it = ctx->limb_tw_cache.find(key);
    if (it == ctx->limb_tw_cache.end()) {
        p2hot_ctx::LimbTables lt;
        const size_t total = (size_t)nttl::limb_tables_w2((int)log_r);
        P2_HIP(ctx, hipMalloc((void **)&lt.tw_all, (total ? total : 1) * sizeof(nttl::W2)));
        for (int r = 0; r < nttl::n_rounds((int)log_r); ++r) {
            const unsigned p = (unsigned)nttl::round_bits((int)log_r, r), log_rb = (unsigned)nttl::round_log_rb((int)log_r, r);
            if (nttl::round_table_w2((int)log_r, r) == 0) continue;  // no table twiddles (last round) or a borrowed table
            if (nttl::round_absorbs((int)log_r, r)) {  // this round's twiddles times the factor round 0 deferred
                const size_t count = (size_t)nttl::round_table_w2((int)log_r, r) / 2;
                P2HOT_LAUNCH(nttl::limb_twiddle_absorb_kernel, dim3(cdiv(count, 256)), dim3(256), 0, ctx->stream,
                             lt.tw_all + nttl::round_table_off((int)log_r, r), log_r, log_rb, p, inverse ? ctx->inv : ctx->fwd);
                P2_LAUNCH_CHECK(ctx);
                continue;
            }
            const size_t count = ((((size_t)1 << p) - 1) << (log_rb - p));
            P2HOT_LAUNCH(nttl::limb_twiddle_kernel, dim3(cdiv(count, 256)), dim3(256), 0, ctx->stream,
                         lt.tw_all + nttl::round_table_off((int)log_r, r), log_rb, p, inverse ? ctx->inv : ctx->fwd);
            P2_LAUNCH_CHECK(ctx);
        }
        if (nttl::uses_ufac((int)log_r)) {
            P2_HIP(ctx, hipMalloc((void **)&lt.ufac, (size_t)8 * nttl::UFAC_WORDS));
            P2HOT_LAUNCH(nttl::limb_ufac_kernel, dim3(1), dim3(64), 0, ctx->stream, lt.ufac, log_r, inverse ? ctx->inv : ctx->fwd);
            P2_LAUNCH_CHECK(ctx);
        }
        it = ctx->limb_tw_cache.emplace(key, lt).first;
    }
    *out = it->second;
    return P2HOT_OK;
}

// A kernel that asks for more than 48 KiB of dynamic LDS says so once per context (the contiguous limb pass sits at exactly
// 64 KiB, the most a launch gets without this; gfx950 has 160 KiB per workgroup)
static int lds_opt_in(p2hot_ctx *ctx, const void *kernel, size_t shm) {
#ifndef P2HOT_EMU
    if (shm <= ((size_t)48 << 10) || ctx->lds_opted.count(kernel)) return P2HOT_OK;
    P2_HIP(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    ctx->lds_opted.insert(kernel);
#else
    (void)ctx, (void)kernel, (void)shm;
#endif
    return P2HOT_OK;
}

static bool limb_supported(unsigned log_r, unsigned log_c) {
    if (log_r + log_c != (unsigned)nttl::TILE_LOG) return false;
    return log_r == 12 || (log_r >= 4 && log_r <= 10);
}

// launches one limb pass; `last_const` != 1 multiplies every output of this (last) pass by it (the 1/n of the inverse transform)
struct BrinArgs {  // the bit-reversed coefficient source of a coset LDE's first pass + where its natural-order copy goes (nttl.hpp BRIN)
    const u64 *src = nullptr;
    size_t src_stride = 0;
    u64 *nat = nullptr;
    size_t nat_stride = 0;
};
static int launch_limb_pass(p2hot_ctx *ctx, const ntt::PassArgs &a, bool inverse, const u64 *twid, unsigned xcd_remap,
                            unsigned zloop, dim3 grid, u64 last_const, const u64 *srow2, const nttl::W2 *sbase,
                            const BrinArgs *brin = nullptr) {
    nttl::LimbPassArgs ra{};
    ra.a = a;
    if (brin && brin->src) {
        ra.brin_src = brin->src;
        ra.brin_stride = brin->src_stride;
        ra.nat_out = brin->nat;
        ra.nat_stride = brin->nat_stride;
    }
    ra.twid = twid;
    ra.xcd_remap = xcd_remap;
    ra.zloop = zloop;
    p2hot_ctx::LimbTables lt;
    P2_TRY(limb_tables(ctx, inverse, a.log_r, &lt));
    ra.tw_all = lt.tw_all;
    ra.ufac = lt.ufac;
    // a workgroup of the contiguous pass keeps its staged tables for several tiles
    ra.tiles_log = 0;
    if (a.log_c == 0)
        while (ra.tiles_log < ctx->limb_tiles_log && (grid.x >> (ra.tiles_log + 1)) >= 1 &&
               (size_t)(grid.x >> (ra.tiles_log + 1)) * grid.y * grid.z >= 4096)
            ++ra.tiles_log;
    grid.x >>= ra.tiles_log;
    if (ra.xcd_remap) ra.xcd_remap -= ra.tiles_log;
    const bool wlast = gl::canon(last_const) != 1;
    ra.wlast[0] = gl::canon(last_const);
    ra.wlast[1] = gl::canon(gl::mul(last_const, nttl::B1));
    ra.wlast[2] = gl::canon(gl::mul(last_const, nttl::B2));
    ra.wlast[3] = gl::canon(gl::mul(last_const, nttl::B3));
    // two tiles per workgroup (1024 threads: every wave of a SIMD behind one barrier) when the build has that form and the
    // launch has an even number of tile groups to pair
    bool dual = false;
#if P2HOT_LIMB_DUAL
    dual = ctx->limb_dual && grid.x >= 2 && grid.x % 2 == 0 && !ra.brin_src;
    if (dual) {
        grid.x >>= 1;
        if (ra.xcd_remap) ra.xcd_remap = ra.xcd_remap > 3 ? ra.xcd_remap - 1 : 0;
    }
#endif
    const size_t shm = nttl::limb_shmem_bytes((int)a.log_r, dual ? 2 : 1);
    if (wlast && !(a.log_r == 12 && a.scale_mode == ntt::SCALE_NONE)) P2_FAIL(ctx, P2HOT_EINVAL, "limb pass: a final constant needs the contiguous pass");
    ra.srow2 = srow2;
    ra.sbase = sbase;
    // the last conversion multiplies by 1 (LAST_UNIT), by a constant (LAST_CONST: the inverse transform's 1/n) or -- the
    // strided first pass of a coset LDE -- by the tile's share of the coset scale (LAST_TILE)
#if P2HOT_LIMB_DUAL
#define P2_LIMB_DUAL_LAUNCH(INVF, LR, MODE, LASTM)                                                         \
    if (dual) {                                                                                            \
        auto kfn2_ = nttl::ntt_limbpass_kernel<INVF, LR, 12 - LR, MODE, LASTM, 2>;                         \
        P2_TRY(lds_opt_in(ctx, reinterpret_cast<const void *>(kfn2_), shm));                               \
        P2HOT_LAUNCH(kfn2_, grid, dim3(nttl::NT * 2), shm, ctx->stream, ra);                               \
        break;                                                                                             \
    }
#else
#define P2_LIMB_DUAL_LAUNCH(INVF, LR, MODE, LASTM)
#endif
#define P2_LIMB(INVF, LR, MODE, LASTM)                                                                     \
    do {                                                                                                   \
        P2_LIMB_DUAL_LAUNCH(INVF, LR, MODE, LASTM)                                                         \
        auto kfn_ = nttl::ntt_limbpass_kernel<INVF, LR, 12 - LR, MODE, LASTM>;                             \
        P2_TRY(lds_opt_in(ctx, reinterpret_cast<const void *>(kfn_), shm));                                \
        P2HOT_LAUNCH(kfn_, grid, dim3(nttl::NT), shm, ctx->stream, ra);                                    \
    } while (0)
#define P2_LIMB_BRIN(LR)                                                                                             \
    do {                                                                                                                 \
        auto kfnb_ = nttl::ntt_limbpass_kernel<false, (LR) == 12 ? 8 : (LR), (LR) == 12 ? 4 : 12 - (LR), ntt::SCALE_TABLE, nttl::LAST_TILE, 1, true>; \
        P2_TRY(lds_opt_in(ctx, reinterpret_cast<const void *>(kfnb_), shm));                                             \
        P2HOT_LAUNCH(kfnb_, grid, dim3(nttl::NT), shm, ctx->stream, ra);                                                 \
    } while (0)
#define P2_LIMB_DIR(LR, MODE, LASTM) do { if (inverse) P2_LIMB(true, LR, MODE, LASTM); else P2_LIMB(false, LR, MODE, LASTM); } while (0)
#define P2_LIMB_MODE(LR)                                                                                          \
    do {                                                                                                          \
        if (a.scale_mode == ntt::SCALE_TABLE) {                                                                   \
            if (LR == 12) P2_LIMB_DIR(LR, ntt::SCALE_TABLE, nttl::LAST_UNIT);                                     \
            else if (ra.brin_src) P2_LIMB_BRIN(LR);                                                               \
            else P2_LIMB_DIR(LR, ntt::SCALE_TABLE, nttl::LAST_TILE);                                              \
        } else if (a.scale_mode == ntt::SCALE_CONST) P2_LIMB_DIR(LR, ntt::SCALE_CONST, nttl::LAST_UNIT);          \
        else P2_LIMB_DIR(LR, ntt::SCALE_NONE, nttl::LAST_UNIT);                                                   \
    } while (0)
    if (a.scale_mode == ntt::SCALE_TABLE && a.log_r != 12 && (!srow2 || !sbase)) P2_FAIL(ctx, P2HOT_EINVAL, "limb pass: missing coset scale tables");
    if (a.canon_out && a.log_c) P2_FAIL(ctx, P2HOT_EINVAL, "limb pass: a strided pass is never the last one and stores no canonical representatives");
    if (ra.brin_src && (inverse || a.log_c == 0 || ra.tiles_log || a.scale_mode != ntt::SCALE_TABLE || a.log_nblk != a.log_r + 12))
        P2_FAIL(ctx, P2HOT_EINVAL, "limb pass: the bit-reversed source belongs to the first pass of a two-pass coset LDE");
    if (wlast) {
        if (inverse)
            P2_LIMB(true, 12, ntt::SCALE_NONE, nttl::LAST_CONST);
        else
            P2_LIMB(false, 12, ntt::SCALE_NONE, nttl::LAST_CONST);
        return P2HOT_OK;
    }
    switch (a.log_r) {
        case 12: P2_LIMB_MODE(12); break;
        case 10: P2_LIMB_MODE(10); break;
        case 9: P2_LIMB_MODE(9); break;
        case 8: P2_LIMB_MODE(8); break;
        case 7: P2_LIMB_MODE(7); break;
        case 6: P2_LIMB_MODE(6); break;
        case 5: P2_LIMB_MODE(5); break;
        case 4: P2_LIMB_MODE(4); break;
        default: P2_FAIL(ctx, P2HOT_EINVAL, "limb pass: unsupported tile 2^%u x 2^%u", a.log_r, a.log_c);
    }
#undef P2_LIMB_MODE
#undef P2_LIMB_BRIN
#undef P2_LIMB_DIR
#undef P2_LIMB
    return P2HOT_OK;
}

// Runs the DIF chain: natural-order input -> bit-reversed output (per polynomial, per z slice).
// The first pass reads `in` (no z offset: every z slice reads the same polynomials) and writes
// `out`; later passes run in place on `out`.
static int run_dif(p2hot_ctx *ctx, const u64 *in, size_t in_stride, u64 *out, size_t out_stride, size_t out_z_stride,
                   size_t batch, size_t zcount, unsigned log_n, const ntt::RootTable &roots, int scale_mode,
                   u64 scale_const, const u64 *srow, const u64 *scol, bool canon_last, const u64 *srow2 = nullptr,
                   const nttl::W2 *sbase = nullptr, const BrinArgs *brin = nullptr) {
    if (batch == 0 || zcount == 0) return P2HOT_OK;
    if (batch > 65535 || zcount > 65535) P2_FAIL(ctx, P2HOT_EINVAL, "batch %zu / z %zu exceed the grid limit", batch, zcount);
    std::vector<Pass> passes = plan_passes(log_n, ctx->ntt_strided_bits);
    unsigned log_nblk = log_n;
    bool limb_all = ctx->use_limb && ctx->use_regpass && ctx->ntt_radix_bits == 3;  // every pass of this chain is a limb pass
    for (const Pass &ps : passes) limb_all = limb_all && limb_supported(ps.log_r, ps.log_c);
    limb_all = limb_all && log_n <= 24;  // strided limb passes take their inter-pass twiddles from the table (blocks <= 2^24)
    for (size_t i = 0; i < passes.size(); ++i) {
        ntt::PassArgs a{};
        const bool first = i == 0;
        a.in = first ? in : out;
        a.out = out;
        a.in_poly_stride = first ? in_stride : out_stride;
        a.in_z_stride = first ? 0 : out_z_stride;
        a.out_poly_stride = out_stride;
        a.out_z_stride = out_z_stride;
        a.log_n = log_n;
        a.log_nblk = log_nblk;
        a.log_r = passes[i].log_r;
        a.log_c = passes[i].log_c;
        a.roots = roots;
        a.scale_mode = first ? scale_mode : ntt::SCALE_NONE;
        a.scale_const = scale_const;
        a.srow = srow;
        a.scol = scol;
        a.canon_out = (canon_last && i + 1 == passes.size()) ? 1 : 0;
        const unsigned tiles_log = log_n - a.log_r - a.log_c;
        dim3 grid(1u << tiles_log, (unsigned)batch, (unsigned)zcount);
        size_t shmem = ((size_t)8) << (a.log_r + a.log_c);
        // live timing per (transform, pass kind): the iNTT, the coset LDE (scale tables) and plain forward transforms apart
        const bool inv_t = roots.lo == ctx->inv.lo, lde_t = scale_mode == ntt::SCALE_TABLE;
        ProfScope ps(ctx, a.log_c ? (inv_t ? "ntt_intt_strided" : lde_t ? "ntt_lde_strided" : "ntt_fwd_strided")
                                  : (inv_t ? "ntt_intt_contig" : lde_t ? "ntt_lde_contig" : "ntt_fwd_contig"));
        if (ctx->use_regpass) {
            ntt::RegPassArgs ra{};
            ra.a = a;
            const bool inverse = roots.lo == ctx->inv.lo;
            // strided passes multiply by w_{n'}^(col * k1) on the way out: the same 2^log_nblk values for every block, polynomial
            // and coset, kept as a table (<= 128 MiB) laid out like a block so that the load is as coalesced as the store
            ra.twid = nullptr;
            if (log_nblk > a.log_r && log_nblk <= 24) {
                auto key = std::make_tuple((int)inverse, log_nblk, a.log_r);
                auto it = ctx->twid_cache.find(key);
                if (it == ctx->twid_cache.end()) {
                    u64 *t = nullptr;
                    P2_HIP(ctx, hipMalloc((void **)&t, (size_t)8 << log_nblk));
                    P2HOT_LAUNCH(ntt::interpass_twiddle_kernel, dim3(cdiv((size_t)1 << log_nblk, 256)), dim3(256), 0, ctx->stream,
                                 t, log_nblk, a.log_r, roots);
                    P2_LAUNCH_CHECK(ctx);
                    it = ctx->twid_cache.emplace(key, t).first;
                }
                ra.twid = it->second;
            }
            ra.local = inverse ? ctx->local_inv : ctx->local_fwd;
            ra.inverse = inverse;
            ra.xcd_remap = (ctx->ntt_xcd_remap && a.log_c && tiles_log >= 3) ? tiles_log : 0;
            // rounds of radix <= 2^maxp, remainder split as evenly as possible.  Radix 8 with 512 threads
            // (8 points per lane, <= 64 VGPRs, 8 waves/SIMD) measured faster than radix 16 with 256 threads:
            // the pass is mostly VALU issue, and the extra waves cover the global / LDS / barrier waits.
            const unsigned maxp = ctx->ntt_radix_bits;
            unsigned rem = a.log_r, k = 0;
            unsigned nr = (rem + maxp - 1) / maxp;
            for (unsigned q = 0; q < nr; ++q) {
                unsigned part = (rem + (nr - q) - 1) / (nr - q);
                ra.rounds[k++] = part;
                rem -= part;
            }
            size_t shm = (size_t)8 * ntt::TILE_WORDS_PADDED;
            // every z slice (coset) reads the same input tile: one workgroup produces them all from one fetch -- when the
            // launch has workgroups to spare.  A small launch keeps the cosets in grid.z instead: at 2^12 rows the loop made
            // the LDE a chain of 8 tile transforms on 2..135 workgroups (95 us whatever the width; 5 such launches per proof)
            if (first && zcount > 1 && a.in_z_stride == 0 && ((size_t)1 << tiles_log) * batch >= ctx->zloop_min_groups) {
                ra.zloop = (unsigned)zcount;
                grid.z = 1;
            }
            if (ctx->use_limb && maxp == 3 && limb_supported(a.log_r, a.log_c) && (a.log_c == 0 || ra.twid) &&
                !(a.scale_mode == ntt::SCALE_TABLE && a.log_c && !srow2)) {
                // the constant scale of an inverse transform moves from the first pass to the last conversion of the last
                // pass (every output of a tile's last round is converted by a constant 4-form anyway: it becomes c * B^i)
                u64 last_const = 1;
                if (limb_all && scale_mode == ntt::SCALE_CONST) {
                    ra.a.scale_mode = ntt::SCALE_NONE;
                    if (i + 1 == passes.size()) last_const = scale_const;
                }
                P2_TRY(launch_limb_pass(ctx, ra.a, inverse, ra.twid, ra.xcd_remap, ra.zloop, grid, last_const, srow2, sbase, first ? brin : nullptr));
                P2_LAUNCH_CHECK(ctx);
                log_nblk -= a.log_r;
                continue;
            }
            if (first && brin && brin->src) P2_FAIL(ctx, P2HOT_EINVAL, "run_dif: the bit-reversed source needs the limb passes");
            if (maxp == 3) {  // one kernel per (direction, scale mode): the per-point mode tests are compiled out
#define P2_NTT512C(INVF, MODE, CT) P2HOT_LAUNCH((ntt::ntt_regpass_kernel<INVF, 512, (CT) ? 8 : 6, MODE, CT>), grid, dim3(512), shm, ctx->stream, ra)
#define P2_NTT512(INVF, MODE) do { if (a.log_c == 0 && log_nblk == a.log_r) P2_NTT512C(INVF, MODE, true); else P2_NTT512C(INVF, MODE, false); } while (0)
                if (a.scale_mode == ntt::SCALE_TABLE) {
                    if (inverse) P2_NTT512(true, ntt::SCALE_TABLE); else P2_NTT512(false, ntt::SCALE_TABLE);
                } else if (a.scale_mode == ntt::SCALE_CONST) {
                    if (inverse) P2_NTT512(true, ntt::SCALE_CONST); else P2_NTT512(false, ntt::SCALE_CONST);
                } else {
                    if (inverse) P2_NTT512(true, ntt::SCALE_NONE); else P2_NTT512(false, ntt::SCALE_NONE);
                }
#undef P2_NTT512C
#undef P2_NTT512
            } else {
                if (inverse)
                    P2HOT_LAUNCH((ntt::ntt_regpass_kernel<true, 256, 4>), grid, dim3(256), shm, ctx->stream, ra);
                else
                    P2HOT_LAUNCH((ntt::ntt_regpass_kernel<false, 256, 4>), grid, dim3(256), shm, ctx->stream, ra);
            }
        } else {
            P2HOT_LAUNCH(ntt::ntt_pass_kernel, grid, dim3(ntt::THREADS), shmem, ctx->stream, a);
        }
        P2_LAUNCH_CHECK(ctx);
        log_nblk -= a.log_r;
    }
    return P2HOT_OK;
}

static unsigned first_pass_log_r(const p2hot_ctx *ctx, unsigned log_n) { return plan_passes(log_n, ctx->ntt_strided_bits)[0].log_r; }

static size_t bitrev_sz(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// ------------------------------------------------------------------ primitives
// out[bitrev(i)] = canon(in[i]) for `batch` arrays
static int launch_bitrev(p2hot_ctx *ctx, const u64 *in, u64 *out, size_t batch, size_t in_stride, size_t out_stride,
                         unsigned log_n) {
    const size_t n = (size_t)1 << log_n;
    ProfScope ps(ctx, "bitrev_permute");
    if (log_n >= 12) {
        P2HOT_LAUNCH(ntt::bitrev_tiled_kernel, dim3((unsigned)(n >> 10), (unsigned)batch), dim3(256), 0, ctx->stream, in, out,
                     in_stride, out_stride, log_n);
    } else {
        P2HOT_LAUNCH(ntt::bitrev_permute_kernel, dim3(cdiv(n, 256), (unsigned)batch), dim3(256), 0, ctx->stream, in, out,
                     in_stride, out_stride, log_n);
    }
    P2_LAUNCH_CHECK(ctx);
    return P2HOT_OK;
}

static int check_log(p2hot_ctx *ctx, unsigned log_n, const char *what) {
    if (log_n > 32) P2_FAIL(ctx, P2HOT_EINVAL, "%s: 2^%u exceeds the two-adicity of the field (fft.rs:171-177)", what, log_n);
    return P2HOT_OK;
}

static int ntt_natural(p2hot_ctx *ctx, u64 *d_data, size_t batch, size_t stride, unsigned log_n, bool inverse) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    P2_TRY(check_log(ctx, log_n, "ntt"));
    const size_t n = (size_t)1 << log_n;
    if (batch == 0) return P2HOT_OK;
    if (!d_data || stride < n) P2_FAIL(ctx, P2HOT_EINVAL, "ntt: null data or stride < n");
    u64 *tmp;
    P2_TRY(scratch_get(ctx, 0, batch * n * 8, (void **)&tmp));
    u64 n_inv = gl::inv(n % gl::P);
    P2_TRY(run_dif(ctx, d_data, stride, tmp, n, 0, batch, 1, log_n, inverse ? ctx->inv : ctx->fwd,
                   inverse ? ntt::SCALE_CONST : ntt::SCALE_NONE, n_inv, nullptr, nullptr, false));
    return launch_bitrev(ctx, tmp, d_data, batch, n, stride, log_n);
}

extern "C" int p2hot_fft_dev(p2hot_ctx *ctx, uint64_t *d_data, size_t batch, size_t poly_stride, unsigned log_n) {
    return ntt_natural(ctx, d_data, batch, poly_stride, log_n, false);
}

extern "C" int p2hot_ifft_dev(p2hot_ctx *ctx, uint64_t *d_data, size_t batch, size_t poly_stride, unsigned log_n) {
    return ntt_natural(ctx, d_data, batch, poly_stride, log_n, true);
}

extern "C" int p2hot_coset_ifft_dev(p2hot_ctx *ctx, uint64_t *d_data, size_t batch, size_t poly_stride, unsigned log_n,
                                    uint64_t shift) {
    P2_TRY(ntt_natural(ctx, d_data, batch, poly_stride, log_n, true));
    if (batch == 0) return P2HOT_OK;
    if (gl::canon(shift) == 0) P2_FAIL(ctx, P2HOT_EINVAL, "coset_ifft: shift must be nonzero");
    // coefficient i is multiplied by shift^-i: two-level table shift^-(i & mask), shift^-((i >> lo_bits) << lo_bits)
    const unsigned lo_bits = (log_n + 1) / 2, hi_bits = log_n - lo_bits;
    const size_t n_lo = (size_t)1 << lo_bits, n_hi = (size_t)1 << hi_bits, n = (size_t)1 << log_n;
    u64 *t;
    P2_TRY(scratch_get(ctx, 1, (n_lo + n_hi) * 8, (void **)&t));
    const u64 si = gl::inv(shift);
    P2HOT_LAUNCH(ntt::pow_table_kernel, dim3(cdiv(n_lo, 256)), dim3(256), 0, ctx->stream, t, n_lo, si, (u64)1, (u64)0);
    P2HOT_LAUNCH(ntt::pow_table_kernel, dim3(cdiv(n_hi, 256)), dim3(256), 0, ctx->stream, t + n_lo, n_hi, si, (u64)n_lo,
                 (u64)0);
    P2HOT_LAUNCH(ntt::scale_by_powers_kernel, dim3(cdiv(n, 256), (unsigned)batch), dim3(256), 0, ctx->stream, d_data,
                 poly_stride, log_n, (const u64 *)t, (const u64 *)(t + n_lo), lo_bits);
    P2_LAUNCH_CHECK(ctx);
    return P2HOT_OK;
}

// scale tables for row blocks [b0, b0 + zc): block b is coset j = bitrev_rb(b), s_b = shift * w_N^j;
// srow[z][i] = s_b^(i * stride), scol[z][base] = s_b^base
static int coset_scale_tables(p2hot_ctx *ctx, unsigned log_n, unsigned rate_bits, u64 shift, size_t b0, size_t zc,
                              const u64 **srow, const u64 **scol, const u64 **srow2, const nttl::W2 **sbase) {
    const Pass first = plan_passes(log_n, ctx->ntt_strided_bits)[0];
    const unsigned log_r = first.log_r;
    auto key = std::make_tuple(log_n, rate_bits, shift, b0, zc, log_r);
    const size_t R = (size_t)1 << log_r, stride = (size_t)1 << (log_n - log_r);
    // a strided limb first pass takes the scale as a tile-shaped (row, column-in-tile) table plus one 4-form per tile
    const bool limb = first.log_c > 0 && limb_supported(first.log_r, first.log_c);
    const size_t tiles = limb ? stride >> first.log_c : 0;
    const size_t words = zc * (R + stride) + (limb ? zc * ((size_t)4096 + tiles * 4) : 0);
    auto it = ctx->scale_cache.find(key);
    u64 *t;
    if (it != ctx->scale_cache.end()) {
        t = it->second;
    } else {
        P2_HIP(ctx, hipMalloc((void **)&t, words * 8));
        const u64 wN = gl::root_of_unity(log_n + rate_bits);
        for (size_t z = 0; z < zc; ++z) {
            u64 s = gl::mul(shift, gl::pow(wN, bitrev_sz(b0 + z, rate_bits)));
            P2HOT_LAUNCH(ntt::pow_table_kernel, dim3(cdiv(R, 256)), dim3(256), 0, ctx->stream, t + z * R, R, s,
                         (u64)stride, (u64)0);
            P2HOT_LAUNCH(ntt::pow_table_kernel, dim3(cdiv(stride, 256)), dim3(256), 0, ctx->stream,
                         t + zc * R + z * stride, stride, s, (u64)1, (u64)0);
            if (limb) {
                u64 *s2 = t + zc * (R + stride) + z * 4096;
                nttl::W2 *sb = reinterpret_cast<nttl::W2 *>(t + zc * (R + stride) + zc * 4096) + z * tiles * 2;
                P2HOT_LAUNCH(nttl::limb_scale_kernel, dim3(cdiv(std::max((size_t)4096, tiles), 256)), dim3(256), 0, ctx->stream, s2,
                             sb, first.log_c, log_n - log_r, s);
            }
        }
        P2_LAUNCH_CHECK(ctx);
        ctx->scale_cache[key] = t;
    }
    *srow = t;
    *scol = t + zc * R;
    *srow2 = limb ? t + zc * (R + stride) : nullptr;
    *sbase = limb ? reinterpret_cast<const nttl::W2 *>(t + zc * (R + stride) + zc * 4096) : nullptr;
    return P2HOT_OK;
}

// can the coset LDE of 2^log_n coefficients read them from the inverse transform's bit-reversed output (nttl.hpp BRIN)?  Two passes,
// both limb passes, inter-pass twiddles from the table: exactly the conditions under which run_dif launches the limb kernels
static bool lde_can_read_bitrev(const p2hot_ctx *ctx, unsigned log_n) {
    if (!ctx->lde_reads_bitrev || !(ctx->use_limb && ctx->use_regpass && ctx->ntt_radix_bits == 3) || log_n > 24) return false;
    const std::vector<Pass> passes = plan_passes(log_n, ctx->ntt_strided_bits);
    return passes.size() == 2 && passes[0].log_c > 0 && limb_supported(passes[0].log_r, passes[0].log_c) &&
           limb_supported(passes[1].log_r, passes[1].log_c);
}

static int coset_lde_impl(p2hot_ctx *ctx, const uint64_t *d_coeffs, size_t W, size_t coeff_stride, unsigned log_n, unsigned rate_bits,
                          uint64_t shift, size_t row_begin, size_t row_count, uint64_t *d_lde, size_t lde_stride, const BrinArgs *brin);

extern "C" int p2hot_coset_lde_dev(p2hot_ctx *ctx, const uint64_t *d_coeffs, size_t W, size_t coeff_stride,
                                   unsigned log_n, unsigned rate_bits, uint64_t shift, size_t row_begin,
                                   size_t row_count, uint64_t *d_lde, size_t lde_stride) {
    return coset_lde_impl(ctx, d_coeffs, W, coeff_stride, log_n, rate_bits, shift, row_begin, row_count, d_lde, lde_stride, nullptr);
}

// `brin` (optional): the coefficients come from brin->src in bit-reversed order instead of d_coeffs (which may then be the same
// pointer as brin->nat: nothing reads it), and the first pass leaves their natural-order, canonical copy in brin->nat
static int coset_lde_impl(p2hot_ctx *ctx, const uint64_t *d_coeffs, size_t W, size_t coeff_stride, unsigned log_n, unsigned rate_bits,
                          uint64_t shift, size_t row_begin, size_t row_count, uint64_t *d_lde, size_t lde_stride, const BrinArgs *brin) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    P2_TRY(check_log(ctx, log_n + rate_bits, "coset_lde"));
    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    if (W == 0 || row_count == 0) return P2HOT_OK;
    if (!d_coeffs || !d_lde || coeff_stride < n || lde_stride < row_count)
        P2_FAIL(ctx, P2HOT_EINVAL, "coset_lde: null pointer or stride too small");
    if (row_begin % n || row_count % n || row_begin + row_count > N)
        P2_FAIL(ctx, P2HOT_EINVAL, "coset_lde: rows [%zu,+%zu) are not whole coset blocks of %zu", row_begin, row_count, n);
    const size_t b0 = row_begin >> log_n, zc = row_count >> log_n;
    const u64 *srow, *scol, *srow2;
    const nttl::W2 *sbase;
    P2_TRY(coset_scale_tables(ctx, log_n, rate_bits, shift, b0, zc, &srow, &scol, &srow2, &sbase));
    return run_dif(ctx, d_coeffs, coeff_stride, d_lde, lde_stride, n, W, zc, log_n, ctx->fwd, ntt::SCALE_TABLE, 0, srow,
                   scol, true, srow2, sbase, brin);
}

// rev_bits > 0: row r of the column-major matrix lands in row reverse_bits(r, rev_bits) of the row-major one (rows == 2^rev_bits):
// the committed LDE matrix back in NATURAL order (P2HOT_LEAVES_NATURAL)
static int transpose_rows(p2hot_ctx *ctx, const uint64_t *d_colmajor, size_t col_stride, size_t W, size_t rows, uint64_t *d_rowmajor,
                          unsigned rev_bits) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    if (W == 0 || rows == 0) return P2HOT_OK;
    if (!d_colmajor || !d_rowmajor || col_stride < rows) P2_FAIL(ctx, P2HOT_EINVAL, "transpose: bad arguments");
    if (cdiv(W, 32) > 65535) P2_FAIL(ctx, P2HOT_EINVAL, "transpose: W too large");
    if (rev_bits && rows != (size_t)1 << rev_bits) P2_FAIL(ctx, P2HOT_EINVAL, "transpose: a bit-reversed destination needs 2^bits rows");
    ProfScope ps(ctx, "transpose");
    P2HOT_LAUNCH(ntt::transpose_kernel, dim3(cdiv(rows, 64), cdiv(W, 32)), dim3(256), 0, ctx->stream, d_colmajor,
                 col_stride, (unsigned)W, rows, d_rowmajor, rev_bits);
    P2_LAUNCH_CHECK(ctx);
    return P2HOT_OK;
}

extern "C" int p2hot_transpose_dev(p2hot_ctx *ctx, const uint64_t *d_colmajor, size_t col_stride, size_t W,
                                   size_t rows, uint64_t *d_rowmajor) {
    return transpose_rows(ctx, d_colmajor, col_stride, W, rows, d_rowmajor, 0);
}

extern "C" int p2hot_reverse_index_bits_dev(p2hot_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, size_t batch,
                                            size_t poly_stride, unsigned log_n) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    if (batch == 0) return P2HOT_OK;
    const size_t n = (size_t)1 << log_n;
    if (!d_in || !d_out || d_in == d_out || poly_stride < n || batch > 65535)
        P2_FAIL(ctx, P2HOT_EINVAL, "reverse_index_bits: bad arguments (out of place only)");
    return launch_bitrev(ctx, d_in, d_out, batch, poly_stride, poly_stride, log_n);
}

extern "C" int p2hot_poseidon_permute_dev(p2hot_ctx *ctx, uint64_t *d_states, size_t count) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    if (count == 0) return P2HOT_OK;
    if (!d_states) P2_FAIL(ctx, P2HOT_EINVAL, "poseidon_permute: null states");
    if (count <= ctx->row_threshold)  // the word-per-lane mapping (also what the challenger runs); larger batches: one permutation per lane
        P2HOT_LAUNCH(merkle::permute_batch_row_kernel, dim3(cdiv(16 * count, 256)), dim3(256), 0, ctx->stream, d_states, count);
    else
        P2HOT_LAUNCH(merkle::permute_batch_kernel, dim3(cdiv(count, 256)), dim3(256), 0, ctx->stream, d_states, count);
    P2_LAUNCH_CHECK(ctx);
    return P2HOT_OK;
}

extern "C" int p2hot_field_selftest_dev(p2hot_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, size_t count,
                                        uint64_t *d_out) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    if (count == 0) return P2HOT_OK;
    if (!d_a || !d_b || !d_out) P2_FAIL(ctx, P2HOT_EINVAL, "field_selftest: null pointer");
    P2HOT_LAUNCH(merkle::field_selftest_kernel, dim3(cdiv(count, 256)), dim3(256), 0, ctx->stream, d_a, d_b, count, d_out);
    P2_LAUNCH_CHECK(ctx);
    return P2HOT_OK;
}

extern "C" int p2hot_gather_rows_dev(p2hot_ctx *ctx, const uint64_t *d_colmajor, size_t col_stride, size_t rows, size_t W,
                                     const uint64_t *d_idx, size_t m, uint64_t *d_out) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    if (W == 0 || m == 0) return P2HOT_OK;
    if (!d_colmajor || !d_idx || !d_out) P2_FAIL(ctx, P2HOT_EINVAL, "gather_rows: null pointer");
    if (rows > col_stride) P2_FAIL(ctx, P2HOT_EINVAL, "gather_rows: rows > col_stride");
    P2HOT_LAUNCH(ntt::gather_rows_kernel, dim3(cdiv(m * W, 256)), dim3(256), 0, ctx->stream, d_colmajor, col_stride, rows,
                 (unsigned)W, d_idx, m, d_out, ctx->d_oob);
    P2_LAUNCH_CHECK(ctx);
    return P2HOT_OK;
}

// ------------------------------------------------------------------ Merkle
struct ForestGeom {
    unsigned h;    // subtree height = log2(leaves) - cap_height
    u64 *dig;      // digest slice of the first subtree of the range
    u64 *cap;      // cap entry of the first subtree of the range
};

static int forest_geom(p2hot_ctx *ctx, unsigned log_leaves, unsigned cap_height, size_t leaf_begin, size_t leaf_count,
                       u64 *d_digests, u64 *d_cap, ForestGeom *g) {
    if (cap_height > log_leaves)
        P2_FAIL(ctx, P2HOT_EINVAL, "cap_height %u > log2(leaves) %u (merkle_tree.rs:195-200)", cap_height, log_leaves);
    const unsigned h = log_leaves - cap_height;
    const size_t sub_leaves = (size_t)1 << h, n_leaves = (size_t)1 << log_leaves;
    if (leaf_begin % sub_leaves || leaf_count % sub_leaves || leaf_begin + leaf_count > n_leaves)
        P2_FAIL(ctx, P2HOT_EINVAL, "merkle: leaves [%zu,+%zu) are not whole cap subtrees of %zu", leaf_begin, leaf_count,
                sub_leaves);
    if (leaf_count && (!d_cap || (h > 0 && !d_digests))) P2_FAIL(ctx, P2HOT_EINVAL, "merkle: null output");
    const size_t s0 = leaf_begin >> h;
    g->h = h;
    g->dig = d_digests ? d_digests + 4 * s0 * (2 * (sub_leaves - 1)) : nullptr;
    g->cap = d_cap ? d_cap + 4 * s0 : nullptr;
    return P2HOT_OK;
}

// leaf sponge for the forest leaves [leaf_offset, leaf_offset + count) on `stream`
template <class Reader>
static int hash_leaves_range(p2hot_ctx *ctx, hipStream_t stream, Reader rd, size_t W, const ForestGeom &g,
                             size_t leaf_offset, size_t count) {
    if (count == 0) return P2HOT_OK;
    ProfScope ps(ctx, "hash_leaves", stream, true);
    if (count <= ctx->row_threshold) {  // a few thousand leaves at most: 16 lanes per leaf, the lowest latency per permutation
        P2HOT_LAUNCH((merkle::hash_leaves_row_kernel<Reader>), dim3(cdiv(16 * count, 256)), dim3(256), 0, stream, rd, (unsigned)W,
                     leaf_offset, count, g.h, g.dig, g.cap);
    } else if (count <= ctx->quad_threshold) {  // too few permutations to fill the chip: 4 lanes per leaf, ~3x lower latency
        P2HOT_LAUNCH((merkle::hash_leaves_quad_kernel<Reader>), dim3(cdiv(4 * count, 256)), dim3(256), 0, stream, rd,
                     (unsigned)W, leaf_offset, count, g.h, g.dig, g.cap);
    } else {
        P2HOT_LAUNCH((merkle::hash_leaves_kernel<Reader>), dim3(cdiv(count, 256)), dim3(256), 0, stream, rd, (unsigned)W,
                     leaf_offset, count, g.h, g.dig, g.cap);
    }
    P2_LAUNCH_CHECK(ctx);
    return P2HOT_OK;
}

// the chunks [off_begin, off_end) of the sponge of `count` leaves (reader, geometry and state all relative to the first of
// them; merkle::hash_leaves_chunks_kernel); word i of leaf t is parked at state[i * state_stride + t]
template <class Reader>
static int hash_leaves_chunks(p2hot_ctx *ctx, hipStream_t stream, Reader rd, size_t W, const ForestGeom &g, size_t count,
                              unsigned off_begin, unsigned off_end, u64 *state, size_t state_stride) {
    if (count == 0 || off_begin >= off_end) return P2HOT_OK;
    ProfScope ps(ctx, "hash_leaves", stream, true);
    P2HOT_LAUNCH((merkle::hash_leaves_chunks_kernel<Reader>), dim3(cdiv(count, 256)), dim3(256), 0, stream, rd, (unsigned)W,
                 (size_t)0, count, g.h, g.dig, g.cap, off_begin, off_end, state, state_stride);
    P2_LAUNCH_CHECK(ctx);
    return P2HOT_OK;
}

static int merkle_levels(p2hot_ctx *ctx, const ForestGeom &g, size_t leaf_count) {
    ProfScope ps(ctx, "merkle_levels");
    // One launch per level.  Walking the top levels of every cap subtree in one launch (a 1024-thread workgroup per subtree,
    // a barrier per level, word-per-lane permutations) was built and measured at recursion size: 146 us per tree against
    // 7 x 12.4 -- a level's nodes then share one CU (four waves per SIMD) instead of spreading over the chip
    for (unsigned level = 1; level <= g.h; ++level) {
        size_t nodes = leaf_count >> level;
        if (nodes <= ctx->row_threshold)
            P2HOT_LAUNCH(merkle::merkle_level_row_kernel, dim3(cdiv(16 * nodes, 256)), dim3(256), 0, ctx->stream, g.dig, g.cap, g.h, level,
                         nodes);
        else if (nodes <= ctx->quad_threshold)
            P2HOT_LAUNCH(merkle::merkle_level_quad_kernel, dim3(cdiv(4 * nodes, 256)), dim3(256), 0, ctx->stream, g.dig,
                         g.cap, g.h, level, nodes);
        else
            P2HOT_LAUNCH(merkle::merkle_level_kernel, dim3(cdiv(nodes, 256)), dim3(256), 0, ctx->stream, g.dig, g.cap, g.h,
                         level, nodes);
        P2_LAUNCH_CHECK(ctx);
    }
    return P2HOT_OK;
}

template <class Reader>
static int merkle_forest(p2hot_ctx *ctx, Reader rd, size_t W, unsigned log_leaves, unsigned cap_height,
                         size_t leaf_begin, size_t leaf_count, u64 *d_digests, u64 *d_cap) {
    ForestGeom g;
    P2_TRY(forest_geom(ctx, log_leaves, cap_height, leaf_begin, leaf_count, d_digests, d_cap, &g));
    if (leaf_count == 0) return P2HOT_OK;
    P2_TRY(hash_leaves_range(ctx, ctx->stream, rd, W, g, 0, leaf_count));
    return merkle_levels(ctx, g, leaf_count);
}

extern "C" int p2hot_merkle_dev(p2hot_ctx *ctx, const uint64_t *d_leaves, int layout, size_t leaf_stride, size_t W,
                                unsigned log_leaves, unsigned cap_height, size_t leaf_begin, size_t leaf_count,
                                uint64_t *d_digests, uint64_t *d_cap) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    if (W > 0 && !d_leaves) P2_FAIL(ctx, P2HOT_EINVAL, "merkle: null leaves");
    if (W > 0xFFFFFFFFull) P2_FAIL(ctx, P2HOT_EINVAL, "merkle: leaf too wide");
    if (layout == 0) {
        if (W > 0 && leaf_stride < leaf_count) P2_FAIL(ctx, P2HOT_EINVAL, "merkle: leaf_stride < leaf_count");
        return merkle_forest(ctx, merkle::ColMajorReader{d_leaves, leaf_stride}, W, log_leaves, cap_height, leaf_begin,
                             leaf_count, d_digests, d_cap);
    } else if (layout == 1) {
        return merkle_forest(ctx, merkle::RowMajorReader{d_leaves, W}, W, log_leaves, cap_height, leaf_begin, leaf_count,
                             d_digests, d_cap);
    }
    P2_FAIL(ctx, P2HOT_EINVAL, "merkle: unknown layout %d", layout);
}

// ------------------------------------------------------------------ PolynomialBatch
extern "C" int p2hot_commit_dev(p2hot_ctx *ctx, const uint64_t *d_cols, size_t col_stride, size_t W, unsigned log_n,
                                unsigned rate_bits, unsigned cap_height, int is_values, size_t row_begin,
                                size_t row_count, uint64_t *d_coeffs, size_t coeff_stride, uint64_t *d_lde,
                                size_t lde_stride, uint64_t *d_leaves, uint64_t *d_digests, uint64_t *d_cap) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    P2_TRY(check_log(ctx, log_n + rate_bits, "commit"));
    const size_t n = (size_t)1 << log_n;
    const unsigned log_N = log_n + rate_bits;
    if (W == 0) P2_FAIL(ctx, P2HOT_EINVAL, "commit: no polynomials (the reference panics on polynomials[0], fri/oracle.rs:90)");
    if (W > 0 && (!d_cols || col_stride < n)) P2_FAIL(ctx, P2HOT_EINVAL, "commit: null columns or stride < n");
    if (W > 0 && !d_lde) P2_FAIL(ctx, P2HOT_EINVAL, "commit: d_lde is required");
    const u64 *coeff_src = d_cols;
    size_t coeff_src_stride = col_stride;
    BrinArgs brin;
    if (is_values) {
        // "IFFT" (oracle.rs:65-69): DIF chain with inverse roots and n^-1, then bit-reverse into d_coeffs
        if (W > 0) {
            if (!d_coeffs || coeff_stride < n) P2_FAIL(ctx, P2HOT_EINVAL, "commit: d_coeffs is required for from_values");
            u64 *tmp;
            P2_TRY(scratch_get(ctx, 0, W * n * 8, (void **)&tmp));
            P2_TRY(run_dif(ctx, d_cols, col_stride, tmp, n, 0, W, 1, log_n, ctx->inv, ntt::SCALE_CONST,
                           gl::inv(n % gl::P), nullptr, nullptr, false));
            // the bit reversal into `polynomials` order: its own kernel, or -- when the LDE below is ONE launch sequence of two limb
            // passes -- folded into the LDE's first pass, which reads the bit-reversed array and writes the natural copy itself
            const bool one_lde = !(ctx->overlap && (row_count >> log_n) > 1 && row_count % n == 0);
            if (one_lde && row_count > 0 && lde_can_read_bitrev(ctx, log_n)) {
                brin.src = tmp;
                brin.src_stride = n;
                brin.nat = d_coeffs;
                brin.nat_stride = coeff_stride;
            } else {
                P2_TRY(launch_bitrev(ctx, tmp, d_coeffs, W, n, coeff_stride, log_n));
            }
        }
        coeff_src = d_coeffs;
        coeff_src_stride = coeff_stride;
    } else if (d_coeffs && d_coeffs != d_cols && W > 0) {
        if (coeff_stride < n) P2_FAIL(ctx, P2HOT_EINVAL, "commit: coeff_stride < n");
        for (size_t c = 0; c < W; ++c)
            P2_HIP(ctx, hipMemcpyAsync(d_coeffs + c * coeff_stride, d_cols + c * col_stride, n * 8,
                                       hipMemcpyDeviceToDevice, ctx->stream));
    }
    // "FFT + blinding" + "transpose LDEs" + reverse_index_bits (oracle.rs:91-98), "build Merkle tree" (oracle.rs:99-103)
    const size_t blocks = row_count >> log_n;
    if (ctx->overlap && W > 0 && blocks > 1 && row_count % n == 0) {
        // Coset blocks are independent: the LDE of block b+1 (wait-bound) runs on the main stream while the
        // Poseidon leaf sponge of block b (VALU-bound) runs on the side stream; the levels follow the join.
        if (W > 0xFFFFFFFFull) P2_FAIL(ctx, P2HOT_EINVAL, "commit: too many columns");
        ForestGeom g;
        P2_TRY(forest_geom(ctx, log_N, cap_height, row_begin, row_count, d_digests, d_cap, &g));
        while (ctx->fork_events.size() < blocks) {
            hipEvent_t e;
            P2_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->fork_events.push_back(e);
        }
        for (size_t b = 0; b < blocks; ++b) {
            P2_TRY(p2hot_coset_lde_dev(ctx, coeff_src, W, coeff_src_stride, log_n, rate_bits, gl::COSET_SHIFT,
                                       row_begin + b * n, n, d_lde + b * n, lde_stride));
            P2_HIP(ctx, hipEventRecord(ctx->fork_events[b], ctx->stream));
            P2_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->fork_events[b], 0));
            P2_TRY(hash_leaves_range(ctx, ctx->side, merkle::ColMajorReader{d_lde, lde_stride}, W, g, b * n, n));
        }
        P2_HIP(ctx, hipEventRecord(ctx->join_event, ctx->side));
        P2_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->join_event, 0));
        P2_TRY(merkle_levels(ctx, g, row_count));
    } else
    {
        (void)blocks;
        P2_TRY(coset_lde_impl(ctx, coeff_src, W, coeff_src_stride, log_n, rate_bits, gl::COSET_SHIFT, row_begin, row_count, d_lde,
                              lde_stride, brin.src ? &brin : nullptr));
        P2_TRY(p2hot_merkle_dev(ctx, d_lde, 0, lde_stride, W, log_N, cap_height, row_begin, row_count, d_digests, d_cap));
    }
    if (d_leaves) P2_TRY(p2hot_transpose_dev(ctx, d_lde, lde_stride, W, row_count, d_leaves));
    return P2HOT_OK;
}

// ------------------------------------------------------------------ Challenger
struct p2hot_challenger {
    p2hot_ctx *ctx;
    fri::Challenger *d;
    u64 *d_io;  // small staging buffer
    size_t io_cap;
};

static int challenger_io(p2hot_challenger *ch, size_t words, u64 **out) {
    p2hot_ctx *ctx = ch->ctx;
    if (ch->io_cap < words) {
        if (ch->d_io) {
            P2_HIP(ctx, stream_sync(ctx));
            P2_HIP(ctx, hipFree(ch->d_io));
            ch->d_io = nullptr;
            ch->io_cap = 0;
        }
        size_t cap = words < 1024 ? 1024 : words;
        P2_HIP(ctx, hipMalloc((void **)&ch->d_io, cap * 8));
        ch->io_cap = cap;
    }
    *out = ch->d_io;
    return P2HOT_OK;
}

extern "C" int p2hot_challenger_create(p2hot_ctx *ctx, p2hot_challenger **out) {
    if (!ctx || !out) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    p2hot_challenger *ch = new p2hot_challenger{ctx, nullptr, nullptr, 0};
    hipError_t e = hipMalloc((void **)&ch->d, sizeof(fri::Challenger));
    if (e != hipSuccess) {
        delete ch;
        P2_FAIL(ctx, P2HOT_ENOMEM, "challenger: hipMalloc failed");
    }
    e = hipMemsetAsync(ch->d, 0, sizeof(fri::Challenger), ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(ch->d);
        delete ch;
        P2_FAIL(ctx, P2HOT_EHIP, "challenger: memset failed");
    }
    *out = ch;
    return P2HOT_OK;
}

extern "C" void p2hot_challenger_destroy(p2hot_challenger *ch) {
    if (!ch) return;
    DeviceGuard dev_guard_(ch->ctx);
    (void)hipStreamSynchronize(ch->ctx->stream);
    if (ch->d_io) (void)hipFree(ch->d_io);
    (void)hipFree(ch->d);
    delete ch;
}

static_assert(sizeof(p2hot_challenger_state) == sizeof(fri::Challenger), "challenger state layout");

extern "C" int p2hot_challenger_load(p2hot_challenger *ch, const p2hot_challenger_state *st) {
    if (!ch || !st) return P2HOT_EINVAL;
    p2hot_ctx *ctx = ch->ctx;
    DeviceGuard dev_guard_(ctx);
    if (st->input_len >= 8 || st->output_len > 8) P2_FAIL(ctx, P2HOT_EINVAL, "challenger: buffer lengths out of range");
    P2_HIP(ctx, hipMemcpyAsync(ch->d, st, sizeof *st, hipMemcpyHostToDevice, ctx->stream));
    P2_HIP(ctx, stream_sync(ctx));
    return P2HOT_OK;
}

extern "C" int p2hot_challenger_store(p2hot_challenger *ch, p2hot_challenger_state *st) {
    if (!ch || !st) return P2HOT_EINVAL;
    p2hot_ctx *ctx = ch->ctx;
    DeviceGuard dev_guard_(ctx);
    P2_HIP(ctx, hipMemcpyAsync(st, ch->d, sizeof *st, hipMemcpyDeviceToHost, ctx->stream));
    P2_HIP(ctx, stream_sync(ctx));
    return P2HOT_OK;
}

// device-side step: observe d_obs[0..n_obs), squeeze n_get challenges into d_out
static int challenger_step_dev(p2hot_challenger *ch, const u64 *d_obs, size_t n_obs, u64 *d_out, size_t n_get) {
    p2hot_ctx *ctx = ch->ctx;
    if (n_obs == 0 && n_get == 0) return P2HOT_OK;
    P2HOT_LAUNCH(fri::challenger_kernel, dim3(1), dim3(64), 0, ctx->stream, ch->d, d_obs, n_obs, d_out, n_get);
    P2_LAUNCH_CHECK(ctx);
    return P2HOT_OK;
}

extern "C" int p2hot_challenger_step(p2hot_challenger *ch, const uint64_t *observe, size_t n_observe,
                                     uint64_t *challenges, size_t n_challenges) {
    if (!ch) return P2HOT_EINVAL;
    p2hot_ctx *ctx = ch->ctx;
    DeviceGuard dev_guard_(ctx);
    if ((n_observe && !observe) || (n_challenges && !challenges)) P2_FAIL(ctx, P2HOT_EINVAL, "challenger: null buffer");
    u64 *io;
    P2_TRY(challenger_io(ch, n_observe + n_challenges, &io));
    if (n_observe) P2_HIP(ctx, hipMemcpyAsync(io, observe, n_observe * 8, hipMemcpyHostToDevice, ctx->stream));
    P2_TRY(challenger_step_dev(ch, io, n_observe, io + n_observe, n_challenges));
    if (n_challenges)
        P2_HIP(ctx, hipMemcpyAsync(challenges, io + n_observe, n_challenges * 8, hipMemcpyDeviceToHost, ctx->stream));
    P2_HIP(ctx, stream_sync(ctx));
    return P2HOT_OK;
}

// ------------------------------------------------------------------ FRI commit phase
// coeffs: host [n][2] interleaved, or (d_planar != NULL) device planes [2][n]
// max_num_query_steps / final_poly_coeff_len: the Option<usize> arguments of fri_committed_trees (prover.rs:89-90), 0 = None
static int fri_commit_core(p2hot_ctx *ctx, const uint64_t *coeffs, const uint64_t *d_planar, unsigned log_n,
                           unsigned rate_bits, unsigned cap_height, const unsigned *arity_bits, unsigned n_rounds,
                           unsigned max_num_query_steps, size_t final_poly_coeff_len,
                           p2hot_challenger *challenger, uint64_t *leaves_out, bool leaves_on_device, uint64_t *digests_out,
                           bool digests_on_device, uint64_t *caps_out, uint64_t *betas_out, uint64_t *final_out,
                           bool defer_sync = false) {
    if (!ctx || !challenger || challenger->ctx != ctx) return P2HOT_EINVAL;
    P2_TRY(check_log(ctx, log_n + rate_bits, "fri_commit"));
    if ((!coeffs && !d_planar) || (n_rounds && !arity_bits)) P2_FAIL(ctx, P2HOT_EINVAL, "fri_commit: null input");
    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    // validate the schedule before touching the device
    {
        unsigned lm = log_n + rate_bits, ln = log_n;
        for (unsigned r = 0; r < n_rounds; ++r) {
            unsigned ab = arity_bits[r];
            if (ab == 0 || ab > ln) P2_FAIL(ctx, P2HOT_EINVAL, "fri_commit: round %u arity 2^%u does not divide the degree bound", r, ab);
            if (lm - ab < cap_height)
                P2_FAIL(ctx, P2HOT_EINVAL, "fri_commit: round %u tree has fewer leaves than the cap (merkle_tree.rs:195-200)", r);
            lm -= ab;
            ln -= ab;
        }
    }
    // one grow-only scratch block, carved up (no per-call hipMalloc / hipFree): stage [n][2], two coefficient plane
    // pairs [2][n], values [2][N], interleaved leaves [N][2], digests (<= N of them), cap, beta
    const size_t cap_words = (size_t)4 << cap_height;
    struct Part {
        u64 *p;
        u64 *u() const { return p; }
    } stage, planes_a, planes_b, values, leaves, digests, cap, beta;
    {
        const size_t words = 2 * n * 3 + 2 * N * 2 + 4 * (N > 1 ? N : 1) + cap_words + 2;
        u64 *base = nullptr;
        P2_TRY(scratch_get(ctx, 2, words * 8, (void **)&base));
        stage.p = base;
        planes_a.p = stage.p + 2 * n;
        planes_b.p = planes_a.p + 2 * n;
        values.p = planes_b.p + 2 * n;
        leaves.p = values.p + 2 * N;
        digests.p = leaves.p + 2 * N;
        cap.p = digests.p + 4 * (N > 1 ? N : 1);
        beta.p = cap.p + cap_words;
    }
    int rc = P2HOT_OK;
    auto body = [&]() -> int {
        u64 *cur = planes_a.u(), *nxt = planes_b.u();
        size_t cur_n = n;  // plane length (= plane stride)
        if (d_planar) {
            P2_HIP(ctx, hipMemcpyAsync(cur, d_planar, n * 16, hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            P2_HIP(ctx, hipMemcpyAsync(stage.p, coeffs, n * 16, hipMemcpyHostToDevice, ctx->stream));
            P2HOT_LAUNCH(fri::deinterleave_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, stage.u(), n, cur, cur + n);
            P2_LAUNCH_CHECK(ctx);
        }
        u64 shift = gl::COSET_SHIFT;
        size_t m = N;
        unsigned log_cur = log_n;
        for (unsigned r = 0; r < n_rounds; ++r) {
            const unsigned ab = arity_bits[r];
            // values = coeffs.lde(rate_bits).coset_fft(shift), rows in bit-reversed order
            // (oracle.rs:215-220 for round 0, prover.rs:119 afterwards; prover.rs:98 reverse_index_bits)
            u64 *v0 = values.u(), *v1 = v0 + m;
            P2_TRY(p2hot_coset_lde_dev(ctx, cur, 2, cur_n, log_cur, rate_bits, shift, 0, m, v0, m));
            if (leaves_out) {
                u64 *dst = leaves_on_device ? leaves_out : leaves.u();  // device-resident trees: write in place
                P2HOT_LAUNCH(fri::interleave_kernel, dim3(cdiv(m, 256)), dim3(256), 0, ctx->stream, v0, v1, m, dst);
                P2_LAUNCH_CHECK(ctx);
                if (!leaves_on_device)
                    P2_HIP(ctx, hipMemcpyAsync(leaves_out, leaves.p, m * 16, hipMemcpyDeviceToHost, ctx->stream));
                leaves_out += 2 * m;
            }
            // prover.rs:99-104: chunk(arity) + flatten -> MerkleTree::new
            const unsigned log_leaves = log_cur + rate_bits - ab;
            const size_t n_leaves = (size_t)1 << log_leaves;
            P2_TRY(merkle_forest(ctx, merkle::FriPlanarReader{v0, v1, ab}, (size_t)2 << ab, log_leaves, cap_height, 0,
                                 n_leaves, digests.u(), cap.u()));
            const size_t nd = p2hot_num_digests(log_leaves, cap_height);
            if (digests_out) {
                if (nd)
                    P2_HIP(ctx, hipMemcpyAsync(digests_out, digests.p, nd * 32,
                                               digests_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                               ctx->stream));
                digests_out += 4 * nd;
            }
            if (caps_out) {
                P2_TRY(d2h(ctx, caps_out, cap.p, cap_words * 8));
                caps_out += cap_words;
            }
            // prover.rs:106-109: observe_cap, beta = get_extension_challenge (stays on the device)
            P2_TRY(challenger_step_dev(challenger, cap.u(), cap_words, beta.u(), 2));
            if (betas_out) {
                P2_TRY(d2h(ctx, betas_out, beta.p, 16));
                betas_out += 2;
            }
            // prover.rs:111-118: fold the coefficients, shift <- shift^arity
            const size_t out_n = cur_n >> ab;
            P2HOT_LAUNCH(fri::fold_kernel, dim3(cdiv(out_n, 256)), dim3(256), 0, ctx->stream, cur, cur + cur_n, ab,
                         beta.u(), out_n, nxt, nxt + out_n);
            P2_LAUNCH_CHECK(ctx);
            u64 *t = cur;
            cur = nxt;
            nxt = t;
            cur_n = out_n;
            log_cur -= ab;
            m >>= ab;
            shift = gl::pow(shift, (u64)1 << ab);
        }
        // prover.rs:122-132: keep the transcript in sync with a verifier circuit that has more query steps:
        // observe an all-zero cap and draw a dummy challenge per missing step
        if (max_num_query_steps > n_rounds) {
            P2_HIP(ctx, hipMemsetAsync(cap.p, 0, cap_words * 8, ctx->stream));
            for (unsigned k = n_rounds; k < max_num_query_steps; ++k)
                P2_TRY(challenger_step_dev(challenger, cap.u(), cap_words, beta.u(), 2));
        }
        // prover.rs:135-139: final_poly = the remaining coefficients, observed by the challenger
        P2HOT_LAUNCH(fri::interleave_kernel, dim3(cdiv(cur_n, 256)), dim3(256), 0, ctx->stream, cur, cur + cur_n, cur_n,
                     stage.u());
        P2_LAUNCH_CHECK(ctx);
        P2_TRY(challenger_step_dev(challenger, stage.u(), 2 * cur_n, nullptr, 0));
        if (final_out) P2_TRY(d2h(ctx, final_out, stage.p, cur_n * 16));
        // prover.rs:140-147: observe zeros up to the padded final polynomial length
        if (final_poly_coeff_len > cur_n) {
            const size_t extra = 2 * (final_poly_coeff_len - cur_n);  // extension elements -> words
            P2_HIP(ctx, hipMemsetAsync(values.p, 0, (extra < 2 * N ? extra : 2 * N) * 8, ctx->stream));
            for (size_t done = 0; done < extra;) {
                size_t chunk = extra - done < 2 * N ? extra - done : 2 * N;
                P2_TRY(challenger_step_dev(challenger, values.u(), chunk, nullptr, 0));
                done += chunk;
            }
        }
        return P2HOT_OK;
    };
    rc = body();
    if (defer_sync) return rc;  // the caller synchronises once, after everything it enqueues behind this (p2hot_prove_openings)
    hipError_t e = stream_sync(ctx);  // host outputs are complete on return
    if (rc == P2HOT_OK && e != hipSuccess) P2_FAIL(ctx, P2HOT_EHIP, "fri_commit: %s", hipGetErrorString(e));
    return rc;
}

extern "C" int p2hot_fri_commit(p2hot_ctx *ctx, const uint64_t *coeffs, unsigned log_n, unsigned rate_bits,
                                unsigned cap_height, const unsigned *arity_bits, unsigned n_rounds,
                                unsigned max_num_query_steps, size_t final_poly_coeff_len,
                                p2hot_challenger *challenger, uint64_t *leaves_out, uint64_t *digests_out,
                                uint64_t *caps_out, uint64_t *betas_out, uint64_t *final_out) {
    if (ctx && !coeffs) P2_FAIL(ctx, P2HOT_EINVAL, "fri_commit: null coefficients");
    return fri_commit_core(ctx, coeffs, nullptr, log_n, rate_bits, cap_height, arity_bits, n_rounds, max_num_query_steps,
                           final_poly_coeff_len, challenger, leaves_out, false, digests_out, false, caps_out, betas_out,
                           final_out);
}

extern "C" int p2hot_fri_commit_dev(p2hot_ctx *ctx, const uint64_t *d_coeffs_planar, unsigned log_n, unsigned rate_bits,
                                    unsigned cap_height, const unsigned *arity_bits, unsigned n_rounds,
                                    unsigned max_num_query_steps, size_t final_poly_coeff_len,
                                    p2hot_challenger *challenger, uint64_t *d_leaves_out, uint64_t *digests_out,
                                    int digests_on_device, uint64_t *caps_out, uint64_t *betas_out, uint64_t *final_out) {
    if (ctx && !d_coeffs_planar) P2_FAIL(ctx, P2HOT_EINVAL, "fri_commit_dev: null coefficients");
    return fri_commit_core(ctx, nullptr, d_coeffs_planar, log_n, rate_bits, cap_height, arity_bits, n_rounds,
                           max_num_query_steps, final_poly_coeff_len, challenger, d_leaves_out, true, digests_out,
                           digests_on_device != 0, caps_out, betas_out, final_out);
}

// ------------------------------------------------------------------ prove_openings prelude
static gl::ext2 ext_pow(gl::ext2 b, u64 e) {
    gl::ext2 r{1, 0};
    while (e) {
        if (e & 1) r = gl::ext_mul(r, b);
        b = gl::ext_mul(b, b);
        e >>= 1;
    }
    return r;
}

// alpha: host words (alpha_host) or two device words (d_alpha); exactly one is non-null.  Nothing here waits for the GPU.
static int final_poly_core(p2hot_ctx *ctx, const uint64_t *const *d_poly_table, const size_t *batch_offsets, size_t n_batches,
                           const uint64_t *points, const uint64_t *alpha_host, const uint64_t *d_alpha, unsigned log_n,
                           uint64_t *d_final) {
    P2_TRY(check_log(ctx, log_n, "fri_final_poly"));
    if (!batch_offsets || !points || (!alpha_host && !d_alpha) || !d_final || (n_batches && !d_poly_table))
        P2_FAIL(ctx, P2HOT_EINVAL, "fri_final_poly: null argument");
    const size_t n = (size_t)1 << log_n;
    // Horner chunks of 64 coefficients for long polynomials; short ones (recursion size) get about 1024 chunks instead, so
    // the serial walks inside a chunk are 4 long at 2^12 (three launches of dependent 64-step walks cost 100 us there)
    const unsigned chunk_log = log_n < 2 ? log_n : std::min(6u, std::max(2u, log_n > 10 ? log_n - 10 : 0u));
    const size_t n_chunks = n >> chunk_log, per = (n_chunks + 1023) / 1024;
    size_t max_j = 1;
    for (size_t i = 0; i < n_batches; ++i) {
        if (batch_offsets[i + 1] < batch_offsets[i]) P2_FAIL(ctx, P2HOT_EINVAL, "fri_final_poly: offsets must ascend");
        max_j = std::max(max_j, batch_offsets[i + 1] - batch_offsets[i]);
    }
    // more than 4096 chunks: the carries in two levels (groups of 64 chunks), see horner_group_walk_kernel
    const unsigned group_log = 6;
    const bool two_level = n_chunks > ctx->horner_two_level_min && n_chunks >= ((size_t)1 << group_log);
    const size_t n_groups = two_level ? n_chunks >> group_log : 0, gper = (n_groups + 1023) / 1024;
    // scratch: composition planes [2][n], chunk totals [2][n_chunks], carries [2][n_chunks], group totals and carries
    // [4][n_groups], alpha powers [max_j + 1][2], alpha [2]
    u64 *sc = nullptr;
    P2_TRY(scratch_get(ctx, 1, (2 * n + 4 * n_chunks + 4 * n_groups + 2 * (max_j + 1) + 2) * 8, (void **)&sc));
    u64 *c0 = sc, *c1 = sc + n, *p0 = sc + 2 * n, *p1 = p0 + n_chunks, *t0 = p1 + n_chunks, *t1 = t0 + n_chunks;
    u64 *g0 = t1 + n_chunks, *g1 = g0 + n_groups, *T0 = g1 + n_groups, *T1 = T0 + n_groups;
    u64 *d_apow = T1 + n_groups, *d_a = d_apow + 2 * (max_j + 1);
    if (alpha_host) {
        ctx->alpha_stage[0] = gl::canon(alpha_host[0]);
        ctx->alpha_stage[1] = gl::canon(alpha_host[1]);
        P2_HIP(ctx, hipMemcpyAsync(d_a, ctx->alpha_stage, 16, hipMemcpyHostToDevice, ctx->stream));
        d_alpha = d_a;
    }
    if (n_batches == 0) P2_HIP(ctx, hipMemsetAsync(d_final, 0, n * 16, ctx->stream));
    // base.powers() restarts at 1 for every batch (reducing.rs:88-89): one table alpha^0 .. alpha^max_j serves all of them,
    // and entry J is the batch's shift_poly factor (reducing.rs:103-106)
    P2HOT_LAUNCH(fri::alpha_powers_kernel, dim3(cdiv(max_j + 1, 256)), dim3(256), 0, ctx->stream, d_alpha, max_j + 1, d_apow);
    P2_LAUNCH_CHECK(ctx);
    for (size_t i = 0; i < n_batches; ++i) {
        const size_t J = batch_offsets[i + 1] - batch_offsets[i];
        {
            ProfScope ps(ctx, "reduce_polys_base");
            if (log_n <= 16)
                P2HOT_LAUNCH(fri::reduce_polys_base_small_kernel, dim3(cdiv(n, 64)), dim3(1024), 0, ctx->stream,
                             d_poly_table + batch_offsets[i], J, (const u64 *)d_apow, n, c0, c1);
            else
                P2HOT_LAUNCH(fri::reduce_polys_base_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream,
                             d_poly_table + batch_offsets[i], J, (const u64 *)d_apow, n, c0, c1);
            P2_LAUNCH_CHECK(ctx);
        }
        ProfScope ps(ctx, "divide_by_linear");
        const gl::ext2 z{gl::canon(points[2 * i]), gl::canon(points[2 * i + 1])};
        const gl::ext2 zL = ext_pow(z, (u64)1 << chunk_log);
        P2HOT_LAUNCH(fri::horner_chunk_totals_kernel, dim3(cdiv(n_chunks, 256)), dim3(256), 0, ctx->stream, (const u64 *)c0,
                     (const u64 *)c1, chunk_log, n_chunks, z, p0, p1);
        if (two_level) {
            const gl::ext2 zG = ext_pow(zL, (u64)1 << group_log);
            P2HOT_LAUNCH(fri::horner_chunk_totals_kernel, dim3(cdiv(n_groups, 256)), dim3(256), 0, ctx->stream, (const u64 *)p0,
                         (const u64 *)p1, group_log, n_groups, zL, g0, g1);
            P2HOT_LAUNCH(fri::horner_carries_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const u64 *)g0, (const u64 *)g1, n_groups, gper,
                         zG, T0, T1);
            P2HOT_LAUNCH(fri::horner_group_walk_kernel, dim3(cdiv(n_groups, 256)), dim3(256), 0, ctx->stream, (const u64 *)p0,
                         (const u64 *)p1, group_log, n_groups, zL, (const u64 *)T0, (const u64 *)T1, t0, t1);
        } else {
            P2HOT_LAUNCH(fri::horner_carries_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const u64 *)p0, (const u64 *)p1,
                         n_chunks, per, zL, t0, t1);
        }
        P2HOT_LAUNCH(fri::horner_emit_kernel, dim3(cdiv(n_chunks, 256)), dim3(256), 0, ctx->stream, (const u64 *)c0,
                     (const u64 *)c1, chunk_log, n_chunks, z, (const u64 *)t0, (const u64 *)t1, (const u64 *)(d_apow + 2 * J),
                     i > 0 ? 1 : 0, d_final, d_final + n);
        P2_LAUNCH_CHECK(ctx);
    }
    return P2HOT_OK;
}

extern "C" int p2hot_fri_final_poly_dev(p2hot_ctx *ctx, const uint64_t *const *d_poly_table, const size_t *batch_offsets,
                                        size_t n_batches, const uint64_t *points, const uint64_t alpha[2], unsigned log_n,
                                        uint64_t *d_final) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    if (!alpha) P2_FAIL(ctx, P2HOT_EINVAL, "fri_final_poly: null argument");
    return final_poly_core(ctx, d_poly_table, batch_offsets, n_batches, points, alpha, nullptr, log_n, d_final);
}

extern "C" int p2hot_eval_polys_dev(p2hot_ctx *ctx, const uint64_t *const *d_poly_table, size_t n_polys, unsigned log_n,
                                    const uint64_t *points, size_t n_points, uint64_t *d_out) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    P2_TRY(check_log(ctx, log_n, "eval_polys"));
    if (n_polys == 0 || n_points == 0) return P2HOT_OK;
    if (!d_poly_table || !points || !d_out) P2_FAIL(ctx, P2HOT_EINVAL, "eval_polys: null argument");
    if (n_polys > 0x7FFFFFFFull) P2_FAIL(ctx, P2HOT_EINVAL, "eval_polys: too many polynomials");
    const size_t n = (size_t)1 << log_n;
    const unsigned seg_log = log_n < 12 ? log_n : 12;  // 4096-coefficient segments -> J * n/4096 workgroups
    const size_t n_seg = n >> seg_log;
    if (n_seg > 65535) P2_FAIL(ctx, P2HOT_EINVAL, "eval_polys: polynomial too long");
    u64 *part;
    const size_t seg = (size_t)1 << seg_log;
    P2_TRY(scratch_get(ctx, 1, (n_polys * n_seg * 2 + 2 * seg) * 8, (void **)&part));
    u64 *w0 = part + n_polys * n_seg * 2, *w1 = w0 + seg;  // z^u, u < seg: shared by all segments and polynomials
    ProfScope ps(ctx, "eval_polys");
    for (size_t p = 0; p < n_points; ++p) {
        const gl::ext2 z{gl::canon(points[2 * p]), gl::canon(points[2 * p + 1])};
        const gl::ext2 zs = ext_pow(z, (u64)1 << seg_log);
        P2HOT_LAUNCH(fri::ext_powers_kernel, dim3(cdiv(seg, 256)), dim3(256), 0, ctx->stream, z, (unsigned)seg, w0, w1);
        P2HOT_LAUNCH(fri::eval_polys_dot_kernel, dim3((unsigned)n_polys, (unsigned)n_seg), dim3(256), 0, ctx->stream,
                     d_poly_table, seg_log, (const u64 *)w0, (const u64 *)w1, part);
        P2HOT_LAUNCH(fri::eval_polys_stage2_kernel, dim3((unsigned)n_polys), dim3(256), 0, ctx->stream, (const u64 *)part,
                     n_seg, zs, ext_pow(zs, 256), d_out + 2 * p * n_polys);
        P2_LAUNCH_CHECK(ctx);
    }
    return P2HOT_OK;
}

extern "C" int p2hot_partial_products_dev(p2hot_ctx *ctx, const uint64_t *d_wires, size_t wires_stride,
                                          const uint64_t *d_sigmas, size_t sigmas_stride, const uint64_t *k_is,
                                          unsigned num_routed, unsigned log_n, unsigned degree, const uint64_t *betas,
                                          const uint64_t *gammas, unsigned num_challenges, uint64_t *d_out,
                                          size_t out_stride) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    P2_TRY(check_log(ctx, log_n, "partial_products"));
    if (num_challenges == 0) return P2HOT_OK;
    const size_t n = (size_t)1 << log_n;
    if (num_routed == 0 || degree < 2) P2_FAIL(ctx, P2HOT_EINVAL, "partial_products: need num_routed > 0 and degree > 1");
    if (!d_wires || !d_sigmas || !k_is || !betas || !gammas || !d_out)
        P2_FAIL(ctx, P2HOT_EINVAL, "partial_products: null argument");
    if (wires_stride < n || sigmas_stride < n || out_stride < n) P2_FAIL(ctx, P2HOT_EINVAL, "partial_products: stride < n");
    const unsigned num_chunks = (num_routed + degree - 1) / degree, num_prods = num_chunks - 1;
    const unsigned chunk_log = log_n < 6 ? log_n : 6;  // rows per scan chunk
    const size_t n_chunks = n >> chunk_log, per = (n_chunks + 1023) / 1024;
    // scratch: k_is | 2 x (chunk denominators [num_chunks][n] | row totals [n]) | chunk products | carries | flag
    // (two sets: a pair of challenges goes through pp_quotients2_kernel in one pass)
    const size_t set_words = (size_t)num_chunks * n + n;
    const size_t words = num_routed + 2 * set_words + 2 * n_chunks + 1;
    u64 *base = nullptr;
    P2_TRY(scratch_get(ctx, 0, words * 8, (void **)&base));
    u64 *d_k = base, *sets = d_k + num_routed, *prod = sets + 2 * set_words, *carry = prod + n_chunks;
    unsigned *flag = (unsigned *)(carry + n_chunks);
    std::vector<u64> kc(num_routed);
    for (unsigned j = 0; j < num_routed; ++j) kc[j] = gl::canon(k_is[j]);
    P2_HIP(ctx, hipMemcpyAsync(d_k, kc.data(), (size_t)num_routed * 8, hipMemcpyHostToDevice, ctx->stream));
    P2_HIP(ctx, hipMemsetAsync(flag, 0, 8, ctx->stream));
    ProfScope ps(ctx, "partial_products");
    std::vector<plonk::PPArgs> args(num_challenges);
    for (unsigned ch = 0; ch < num_challenges; ++ch) {
        plonk::PPArgs &a = args[ch];
        u64 *dchunk = sets + (size_t)(ch & 1) * set_words, *total = dchunk + (size_t)num_chunks * n;
        a.wires = d_wires;
        a.sigmas = d_sigmas;
        a.wires_stride = wires_stride;
        a.sigmas_stride = sigmas_stride;
        a.k_is = d_k;
        a.num_routed = num_routed;
        a.degree = degree;
        a.num_chunks = num_chunks;
        a.log_n = log_n;
        a.beta = gl::canon(betas[ch]);
        a.gamma = gl::canon(gammas[ch]);
        a.roots = ctx->fwd;
        a.pp = d_out + ((size_t)num_challenges + (size_t)ch * num_prods) * out_stride;
        a.pp_stride = out_stride;
        a.dchunk = dchunk;
        a.total = total;
        a.zero_flag = flag;
    }
    for (unsigned ch = 0; ch < num_challenges; ++ch) {
        const plonk::PPArgs &a = args[ch];
        const u64 *total = a.total;
        u64 *z = d_out + (size_t)ch * out_stride;
        if (ctx->pp_streams && (ch & 1) == 0 && ch + 1 < num_challenges) {  // a pair of challenges: one pass over the wires and sigmas
            plonk::PPArgs2 two{{args[ch], args[ch + 1]}};
            P2HOT_LAUNCH(plonk::pp_quotients2_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, two);
        } else if (!(ctx->pp_streams && (ch & 1))) {
            P2HOT_LAUNCH(plonk::pp_quotients_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, a);
        }
        P2HOT_LAUNCH(plonk::pp_chunk_totals_kernel, dim3(cdiv(n_chunks, 256)), dim3(256), 0, ctx->stream, (const u64 *)total,
                     chunk_log, n_chunks, prod);
        P2HOT_LAUNCH(plonk::pp_carries_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const u64 *)prod, n_chunks, per, carry);
        P2HOT_LAUNCH(plonk::pp_emit_z_kernel, dim3(cdiv(n_chunks, 256)), dim3(256), 0, ctx->stream, (const u64 *)total,
                     chunk_log, n_chunks, (const u64 *)carry, z);
        if (num_prods)
            P2HOT_LAUNCH(plonk::pp_scale_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, a.pp, out_stride, num_prods,
                         (const u64 *)z, n);
        P2_LAUNCH_CHECK(ctx);
    }
    unsigned zero = 0;
    P2_HIP(ctx, hipMemcpyAsync(&zero, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    P2_HIP(ctx, stream_sync(ctx));
    if (zero) P2_FAIL(ctx, P2HOT_EINVAL, "partial_products: tried to invert zero (a denominator wire + beta*sigma + gamma vanished)");
    return P2HOT_OK;
}

extern "C" int p2hot_merkle_paths_dev(p2hot_ctx *ctx, const uint64_t *d_digests, unsigned log_leaves,
                                      unsigned cap_height, const uint64_t *d_idx, size_t m, uint64_t *d_out) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    if (cap_height > log_leaves) P2_FAIL(ctx, P2HOT_EINVAL, "merkle_paths: cap_height > log2(leaves)");
    const unsigned layers = log_leaves - cap_height;
    if (m == 0 || layers == 0) return P2HOT_OK;
    if (!d_digests || !d_idx || !d_out) P2_FAIL(ctx, P2HOT_EINVAL, "merkle_paths: null pointer");
    P2HOT_LAUNCH(fri::merkle_paths_kernel, dim3(cdiv(m * layers, 256)), dim3(256), 0, ctx->stream, d_digests, log_leaves,
                 cap_height, d_idx, m, d_out, ctx->d_oob);
    P2_LAUNCH_CHECK(ctx);
    return P2HOT_OK;
}

// fri_proof_of_work (fri/prover.rs:153-202) without a host round trip: chunks of candidates (doubling from 2^14) are
// enqueued back to back up to 2^(pow_bits + 5) candidates; a chunk retires at once when an earlier one has found a
// witness.  d_best (8 bytes, device) receives the SMALLEST witness, or stays ~0 if none was found in that range
// (probability e^-32: the caller then continues with pow_continue_host).  Returns the first candidate NOT covered.
static int pow_search_dev(p2hot_ctx *ctx, p2hot_challenger *challenger, unsigned pow_bits, unsigned long long *d_best, u64 *next_start) {
    P2_HIP(ctx, hipMemsetAsync(d_best, 0xFF, 8, ctx->stream));
    // P2HOT_POW_RANGE_LOG (tests): log2 of the searched range relative to the expected 2^pow_bits trials, default 5
    int extra = 5;
    if (const char *e = getenv("P2HOT_POW_RANGE_LOG")) extra = atoi(e);
    // never more than 2^34 candidates (about 1100 launches) without looking at the result: a grind that large is
    // continued by pow_continue_host, which checks after every chunk
    const int lg = std::min((int)pow_bits + extra, 34);
    const u64 limit = (u64)1 << (lg < 0 ? 0 : lg);
    // the first chunk is the expected number of trials (a witness with probability 1 - 1/e), the following ones double and
    // retire at once when an earlier one found a witness.  2^16 lanes are one wave per SIMD (a lone wave's latency), 2^18 fill
    // every SIMD with four and take 2.5x as long (107 us measured) -- 4x the expected trials up front cost more than they saved
    u64 chunk = (u64)1 << std::min(18u, std::max(14u, pow_bits)), start = 0;
    while (start < limit) {
        const u64 count = std::min(limit - start, chunk);
        P2HOT_LAUNCH(fri::pow_kernel, dim3(cdiv(count, 256)), dim3(256), 0, ctx->stream, challenger->d, pow_bits, start, count, d_best);
        P2_LAUNCH_CHECK(ctx);
        start += count;
        if (chunk < ((u64)1 << 24)) chunk <<= 1;
    }
    *next_start = start;
    return P2HOT_OK;
}

// the rest of the candidate range, chunk by chunk with a host check after each (only reached when pow_search_dev's range held no witness)
static int pow_continue_host(p2hot_ctx *ctx, p2hot_challenger *challenger, unsigned pow_bits, unsigned long long *d_best, u64 start,
                             unsigned long long *best_out) {
    unsigned long long best = ~0ull;
    u64 chunk = (u64)1 << 16;
    while (start < gl::P) {  // candidates 0 ..= P-1 in the reference (prover.rs:182)
        const u64 count = std::min(gl::P - start, chunk);
        if (chunk < ((u64)1 << 24)) chunk <<= 1;
        P2HOT_LAUNCH(fri::pow_kernel, dim3(cdiv(count, 256)), dim3(256), 0, ctx->stream, challenger->d, pow_bits, start, count, d_best);
        P2_LAUNCH_CHECK(ctx);
        P2_HIP(ctx, hipMemcpyAsync(&best, d_best, 8, hipMemcpyDeviceToHost, ctx->stream));
        P2_HIP(ctx, stream_sync(ctx));
        if (best != ~0ull) break;
        start += count;
    }
    if (best == ~0ull) P2_FAIL(ctx, P2HOT_EUNSUPPORTED, "fri_pow: no witness found");
    *best_out = best;
    return P2HOT_OK;
}

extern "C" int p2hot_fri_pow(p2hot_ctx *ctx, p2hot_challenger *challenger, unsigned pow_bits, uint64_t *witness_out) {
    if (!ctx || !challenger || challenger->ctx != ctx || !witness_out) return P2HOT_EINVAL;
    if (pow_bits > 64) P2_FAIL(ctx, P2HOT_EINVAL, "fri_pow: pow_bits > 64");
    u64 *io;
    P2_TRY(challenger_io(challenger, 8, &io));
    unsigned long long best = ~0ull;
    u64 next = 0;
    P2_TRY(pow_search_dev(ctx, challenger, pow_bits, (unsigned long long *)io, &next));
    P2_HIP(ctx, hipMemcpyAsync(&best, io, 8, hipMemcpyDeviceToHost, ctx->stream));
    P2_HIP(ctx, stream_sync(ctx));
    if (best == ~0ull) P2_TRY(pow_continue_host(ctx, challenger, pow_bits, (unsigned long long *)io, next, &best));
    *witness_out = best;
    u64 w = best, resp;
    return p2hot_challenger_step(challenger, &w, 1, &resp, 1);  // prover.rs:197-198
}

// ------------------------------------------------------------------ PolynomialBatch, host pointers
// A device-resident PolynomialBatch (fri/oracle.rs:30-37): `polynomials` (coefficients), the LDE matrix (= merkle_tree.leaves,
// column-major, committed row order) and merkle_tree.digests; optionally the input values (P2HOT_KEEP_VALUES).
struct p2hot_batch {
    p2hot_ctx *ctx;
    u64 *d_lde;
    size_t W, N;
    u64 *d_dig;  // the tree's digest array (reference layout) stays on the device for p2hot_batch_paths
    unsigned log_N, cap_height;
    u64 *d_coef = nullptr;  // [W][n], stride n
    u64 *d_vals = nullptr;  // [W][n] values on H_n, kept on request
    unsigned log_n = 0, rate_bits = 0;
    bool owned = true;      // false: a view over caller-owned device buffers (p2hot_batch_wrap_dev)
    size_t S = 0;           // blinding (oracle.rs:123-137): salt columns W .. W+S-1 of d_lde; a leaf is W + S words wide
    // a member of a batched commitment (p2hot_commit_many): its columns are interleaved with the other proofs' ([W][M][N]), the
    // blocks are shared and go back to the pool with the last member
    size_t lde_stride = 0, coef_stride = 0;  // elements between consecutive columns; 0 = N / n
    struct SharedBlocks *shared = nullptr;
    // P2HOT_LEAVES_ASYNC: the row-major leaf matrix still travelling to the caller's buffer -- its device staging block, one event
    // per row block (rows [k * rows_per_block, ...) of the CALLER's row order), on the context's leaf stream
    struct LeafCopy {
        void *d_staging = nullptr;
        std::vector<hipEvent_t> ev;
        hipEvent_t aux = nullptr;  // "the matrix is transposed": what the leaf stream waits for before the first block
        hipEvent_t aux2 = nullptr; // "the digests are on their way home": what the later blocks queue behind
        size_t rows_per_block = 0, rows = 0;
    } *leafcopy = nullptr;
    size_t col_stride_lde() const { return lde_stride ? lde_stride : N; }
    size_t col_stride_coef() const { return coef_stride ? coef_stride : ((size_t)1 << log_n); }
};
struct SharedBlocks {
    std::atomic<int> refs{0};
    void *lde = nullptr, *dig = nullptr, *coef = nullptr;
};

// A device-resident column set [W][n] (Vec<PolynomialValues> / Vec<PolynomialCoeffs> that never visits the host)
struct p2hot_cols {
    p2hot_ctx *ctx;
    u64 *d;
    size_t W;
    unsigned log_n;
    bool owned;
};

#include "host_prover.hpp"
#include "host_multi.hpp"
