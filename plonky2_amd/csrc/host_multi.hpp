// host_multi.hpp -- the coset-sharded PolynomialBatch commit across the GPUs of one node, BEHIND the C ABI
// (include/p2hot.h "multi-GPU").  Included at the end of p2hot.hip (one TU).
//
// The reference has no multi-device mode; this is the MI355X design:
//   * the rate-1/B LDE is B independent coset transforms, and in the committed (bit-reversed) leaf order coset j is the
//     contiguous row block bitrev(j); with B = 8 and cap_height = 4 a block is two whole cap subtrees.  Rank r owns rows
//     [r*N/G, (r+1)*N/G): it runs the LDE, the Poseidon leaf sponge and the Merkle levels of its rows with NO data-path exchange;
//   * every rank needs all W*n coefficients: the iNTT is column-sharded (ceil(W/G) columns per rank) and followed by an
//     all-gather of coefficients (W*n*8 bytes in total), pipelined in column chunks on a second stream: chunk k travels
//     while chunk k+1 runs its iNTT and chunk k-1 its LDE;
//   * a rank's subtrees are a contiguous slice of the reference digest array, so one all-gather of the cap entries (and,
//     on request, of the digest slices) reassembles MerkleTree::cap (::digests).  No reductions anywhere.
// Transports (all move "rank r's slice at base + offsets[r]" to every other rank, in place):
//   RCCL, one process per GPU   p2hot_comm_create_rccl: ncclCommInitRank; a grouped ncclBroadcast per slice (= an
//                               all-gather with free placement) on the comm stream over xGMI
//   RCCL, one process, N GPUs   p2hot_group_create: ncclCommInitAll, the same grouped broadcasts for all local ranks
//   peer copies                 p2hot_group_create with a repeated device (one-GPU test boxes) or P2HOT_GROUP_PEER_COPY:
//                               hipMemcpyAsync between the ranks' buffers (xGMI is a full mesh: a direct all-gather)
//   caller-supplied             p2hot_comm_create_callback: the host application's own collective (the CPU tests use
//                               torch.distributed/gloo; a Rust prover may bring MPI or its own RCCL communicator)
// RCCL is bound at run time (dlopen: the copy PyTorch already loaded if there is one, else /opt/rocm's), so libp2hot.so
// carries no link-time dependency on it and two RCCL copies never meet in one process.
#pragma once
#include <chrono>

#ifdef P2HOT_EMU
#include "rccl_emu.h"  // the emulator's fake RCCL (tests/emu): same calls, device / stream / buffer identity enforced
#else
#include <dlfcn.h>
#endif

#include <cstring>
#include <functional>
#include <memory>

// ---- the few RCCL entry points used (rccl.h:40-43, :187, :220, :236, :260, :339, :460, :591 ncclBroadcast, :622 ncclAllGather, :923-929) ----
namespace rccl {
struct UniqueId {
    char internal[P2HOT_UNIQUE_ID_BYTES];
};
typedef void *Comm;
struct Api {
    void *lib = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommInitAll)(Comm *, int, const int *) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
    bool ok = false;
};
static const int kUint8 = 1;  // ncclUint8, rccl.h:460
static Api &api() {
    static Api a;
    static bool tried = false;
    if (tried) return a;
    tried = true;
#ifdef P2HOT_EMU
    a.GetUniqueId = [](UniqueId *id) { return emu_ncclGetUniqueId(id->internal); };
    a.CommInitRank = [](Comm *c, int n, UniqueId id, int r) { return emu_ncclCommInitRank(c, n, id.internal, r); };
    a.CommInitAll = [](Comm *c, int n, const int *d) { return emu_ncclCommInitAll(c, n, d); };
    a.CommDestroy = [](Comm c) { return emu_ncclCommDestroy(c); };
    a.Broadcast = [](const void *s, void *r, size_t n, int dt, int root, Comm c, hipStream_t st) { return emu_ncclBroadcast(s, r, n, dt, root, c, st); };
    a.AllGather = [](const void *s, void *r, size_t n, int dt, Comm c, hipStream_t st) { return emu_ncclAllGather(s, r, n, dt, c, st); };
    a.GroupStart = []() { return emu_ncclGroupStart(); };
    a.GroupEnd = []() { return emu_ncclGroupEnd(); };
    a.GetErrorString = [](int rc) { return emu_ncclGetErrorString(rc); };
    a.ok = true;
#else
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : names)  // the copy this process already uses (PyTorch's), so two RCCLs never meet
        if (!a.lib) a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    const char *paths[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *p : paths)
        if (!a.lib) a.lib = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
    if (!a.lib) {
        a.err = std::string("RCCL not found: ") + dlerror();
        return a;
    }
    bool all = true;
    auto sym = [&](const char *n) {
        void *s = dlsym(a.lib, n);
        if (!s) {
            all = false;
            a.err = std::string("RCCL symbol missing: ") + n;
        }
        return s;
    };
    a.GetUniqueId = (int (*)(UniqueId *))sym("ncclGetUniqueId");
    a.CommInitRank = (int (*)(Comm *, int, UniqueId, int))sym("ncclCommInitRank");
    a.CommInitAll = (int (*)(Comm *, int, const int *))sym("ncclCommInitAll");
    a.CommDestroy = (int (*)(Comm))sym("ncclCommDestroy");
    a.Broadcast = (int (*)(const void *, void *, size_t, int, int, Comm, hipStream_t))sym("ncclBroadcast");
    a.AllGather = (int (*)(const void *, void *, size_t, int, Comm, hipStream_t))sym("ncclAllGather");
    a.GroupStart = (int (*)())sym("ncclGroupStart");
    a.GroupEnd = (int (*)())sym("ncclGroupEnd");
    a.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
    a.ok = all;
#endif
    return a;
}
}  // namespace rccl

struct p2hot_group;

struct p2hot_comm {
    p2hot_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    enum Kind { RCCL, CALLBACK, GROUP } kind = CALLBACK;
    rccl::Comm nccl = nullptr;
    p2hot_allgather_fn fn = nullptr;
    void *user = nullptr;
    p2hot_group *group = nullptr;
    // how equal, contiguous slices travel on the RCCL transports: 0 = a grouped ncclBroadcast per slice (an all-gather with free
    // placement), 1 = ncclAllGather (RCCL's tuned path; pipelined coefficient chunks then go through a chunk-major staging block).
    // P2HOT_EXCHANGE=broadcast|allgather pins it, otherwise p2hot_comm_selftest / p2hot_group_create time both and pick.
    int exchange_mode = 0;
    hipStream_t comm_stream = nullptr;  // collectives run here, beside the transforms on the context's stream
    std::vector<hipEvent_t> ev_ready, ev_done;  // per pipeline slot: "slice written" (compute -> comm), "gathered" (comm -> compute)
};

struct p2hot_group {
    std::vector<p2hot_ctx *> ctx;
    std::vector<p2hot_comm *> comm;
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    bool use_rccl = false;
    std::string err;
};

#define P2_NCCL(ctx, call)                                                                                   \
    do {                                                                                                     \
        int r_ = (call);                                                                                     \
        if (r_ != 0) P2_FAIL(ctx, P2HOT_ECOMM, "%s: %s", #call, rccl::api().GetErrorString ? rccl::api().GetErrorString(r_) : "?"); \
    } while (0)

// P2HOT_EXCHANGE pins how equal contiguous slices travel: "broadcast" (0), "allgather" (1); anything else: measured (-1)
static int exchange_mode_from_env() {
    const char *e = getenv("P2HOT_EXCHANGE");
    if (e && !strcmp(e, "broadcast")) return 0;
    if (e && !strcmp(e, "allgather")) return 1;
    return -1;
}

static int comm_events(p2hot_comm *c, size_t slots) {
    p2hot_ctx *ctx = c->ctx;
    P2_HIP(ctx, hipSetDevice(ctx->device));
    if (!c->comm_stream) P2_HIP(ctx, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    while (c->ev_ready.size() < slots) {
        hipEvent_t a, b;
        P2_HIP(ctx, hipEventCreateWithFlags(&a, hipEventDisableTiming));
        P2_HIP(ctx, hipEventCreateWithFlags(&b, hipEventDisableTiming));
        c->ev_ready.push_back(a);
        c->ev_done.push_back(b);
    }
    return P2HOT_OK;
}

extern "C" int p2hot_comm_unique_id(uint8_t out[P2HOT_UNIQUE_ID_BYTES]) {
    if (!out) return P2HOT_EINVAL;
    rccl::Api &a = rccl::api();
    if (!a.ok) return P2HOT_ECOMM;
    rccl::UniqueId id;
    if (a.GetUniqueId(&id) != 0) return P2HOT_ECOMM;
    std::copy(id.internal, id.internal + P2HOT_UNIQUE_ID_BYTES, (char *)out);
    return P2HOT_OK;
}

extern "C" int p2hot_comm_create_rccl(p2hot_ctx *ctx, int rank, int world, const uint8_t id[P2HOT_UNIQUE_ID_BYTES], p2hot_comm **out) {
    if (!ctx || !out) return P2HOT_EINVAL;
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || !id) P2_FAIL(ctx, P2HOT_EINVAL, "comm_create_rccl: bad rank %d / world %d", rank, world);
    rccl::Api &a = rccl::api();
    if (!a.ok) P2_FAIL(ctx, P2HOT_ECOMM, "comm_create_rccl: %s", a.err.c_str());
    std::unique_ptr<p2hot_comm> c(new p2hot_comm());
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->kind = p2hot_comm::RCCL;
    rccl::UniqueId uid;
    std::copy(id, id + P2HOT_UNIQUE_ID_BYTES, (uint8_t *)uid.internal);
    P2_HIP(ctx, hipSetDevice(ctx->device));
    P2_NCCL(ctx, a.CommInitRank(&c->nccl, world, uid, rank));
    c->exchange_mode = exchange_mode_from_env() == 1 && a.AllGather ? 1 : 0;  // unpinned: broadcasts until p2hot_comm_selftest has timed both
    P2_TRY(comm_events(c.get(), 1));
    *out = c.release();
    return P2HOT_OK;
}

extern "C" int p2hot_comm_create_callback(p2hot_ctx *ctx, int rank, int world, p2hot_allgather_fn fn, void *user, p2hot_comm **out) {
    if (!ctx || !out) return P2HOT_EINVAL;
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || !fn) P2_FAIL(ctx, P2HOT_EINVAL, "comm_create_callback: bad rank %d / world %d or null transport", rank, world);
    p2hot_comm *c = new p2hot_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->kind = p2hot_comm::CALLBACK;
    c->fn = fn;
    c->user = user;
    *out = c;
    return P2HOT_OK;
}

extern "C" void p2hot_comm_destroy(p2hot_comm *c) {
    if (!c) return;
    DeviceGuard restore_caller_device_(0);
    (void)hipSetDevice(c->ctx->device);
    if (c->comm_stream) {
        (void)hipStreamSynchronize(c->comm_stream);
        (void)hipStreamDestroy(c->comm_stream);
    }
    for (auto e : c->ev_ready) (void)hipEventDestroy(e);
    for (auto e : c->ev_done) (void)hipEventDestroy(e);
    if (c->nccl && rccl::api().ok) (void)rccl::api().CommDestroy(c->nccl);
    delete c;
}

extern "C" int p2hot_comm_rank(const p2hot_comm *c) { return c ? c->rank : -1; }
extern "C" int p2hot_comm_world(const p2hot_comm *c) { return c ? c->world : 0; }

// ------------------------------------------------------------------ the exchange
// One entry per LOCAL rank (1 in the process-per-GPU mode, world in the single-process group): base[s] is that rank's copy
// of the buffer; rank r's slice lives at base[s] + offsets[r] (bytes) on every rank and is valid on rank r.
// slot: which (ev_ready, ev_done) pair orders this exchange against the compute stream; the caller makes the compute
// stream wait for it with gather_wait before touching the gathered slices.
static int gather_start(std::vector<p2hot_comm *> &cs, std::vector<u64 *> &base, const std::vector<size_t> &offsets, size_t bytes,
                        size_t slot) {
    p2hot_comm *c0 = cs[0];
    const int world = c0->world;
    // a one-rank RCCL communicator still issues its (trivial) collectives: the same calls, streams and events as at scale
    if (bytes == 0 || (world == 1 && c0->kind != p2hot_comm::RCCL)) return P2HOT_OK;
    if (c0->kind == p2hot_comm::CALLBACK) {  // the host application's collective: synchronous, after the slice is complete
        p2hot_ctx *ctx = c0->ctx;
        P2_HIP(ctx, stream_sync(ctx));
        const auto t0 = std::chrono::steady_clock::now();
        int rc = c0->fn(c0->user, base[0], offsets.data(), world, bytes, (void *)ctx->stream);
        if (ctx->profiling) {  // the host application's collective is synchronous: its wall time is the exchange span
            auto &acc = ctx->prof_acc["exchange"];
            acc.first += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            acc.second += 1;
        }
        if (rc != 0) P2_FAIL(ctx, P2HOT_ECOMM, "the caller-supplied all-gather failed (%d)", rc);
        return P2HOT_OK;
    }
    for (size_t s = 0; s < cs.size(); ++s) {  // slice written on the compute stream -> visible to the comm stream
        p2hot_ctx *ctx = cs[s]->ctx;
        P2_HIP(ctx, hipSetDevice(ctx->device));
        P2_TRY(comm_events(cs[s], slot + 1));
        P2_HIP(ctx, hipEventRecord(cs[s]->ev_ready[slot], ctx->stream));
    }
    // live timing (p2hot_profile_enable): the span of this exchange on rank 0's communication stream, waits for the peers'
    // slices included -- what a slow scaling curve is diagnosed with ("exchange" next to the compute kernels of the same rank)
    ProfScope exchange_scope(c0->ctx, "exchange", c0->comm_stream, true);
    const bool rccl_path = c0->kind == p2hot_comm::RCCL || (c0->kind == p2hot_comm::GROUP && c0->group->use_rccl);
    if (rccl_path) {
        rccl::Api &a = rccl::api();
        for (size_t s = 0; s < cs.size(); ++s) {
            P2_HIP(cs[s]->ctx, hipSetDevice(cs[s]->ctx->device));
            P2_HIP(cs[s]->ctx, hipStreamWaitEvent(cs[s]->comm_stream, cs[s]->ev_ready[slot], 0));
        }
        bool contiguous = true;  // rank r's slice directly behind rank r-1's: what ncclAllGather (in place) needs
        for (int r = 1; r < world; ++r) contiguous = contiguous && offsets[(size_t)r] == offsets[0] + (size_t)r * bytes;
        P2_NCCL(c0->ctx, a.GroupStart());
        if (c0->exchange_mode == 1 && contiguous && a.AllGather) {
            for (size_t s = 0; s < cs.size(); ++s) {
                char *recv = (char *)base[s] + offsets[0];
                int rc = a.AllGather(recv + (size_t)cs[s]->rank * bytes, recv, bytes, rccl::kUint8, cs[s]->nccl, cs[s]->comm_stream);
                if (rc != 0) {
                    (void)a.GroupEnd();
                    P2_FAIL(c0->ctx, P2HOT_ECOMM, "ncclAllGather: %s", a.GetErrorString(rc));
                }
            }
        } else
        for (size_t s = 0; s < cs.size(); ++s)
            for (int r = 0; r < world; ++r) {
                char *p = (char *)base[s] + offsets[r];
                int rc = a.Broadcast(p, p, bytes, rccl::kUint8, r, cs[s]->nccl, cs[s]->comm_stream);
                if (rc != 0) {
                    (void)a.GroupEnd();
                    P2_FAIL(c0->ctx, P2HOT_ECOMM, "ncclBroadcast: %s", a.GetErrorString(rc));
                }
            }
        P2_NCCL(c0->ctx, a.GroupEnd());
    } else {  // peer copies inside the single-process group: rank s pulls every other rank's slice
        for (size_t s = 0; s < cs.size(); ++s) {
            p2hot_ctx *ctx = cs[s]->ctx;
            P2_HIP(ctx, hipSetDevice(ctx->device));
            for (size_t r = 0; r < cs.size(); ++r) {
                if (r == s) continue;
                P2_HIP(ctx, hipStreamWaitEvent(cs[s]->comm_stream, cs[r]->ev_ready[slot], 0));
                P2_HIP(ctx, hipMemcpyAsync((char *)base[s] + offsets[r], (const char *)base[r] + offsets[r], bytes, hipMemcpyDefault,
                                           cs[s]->comm_stream));
            }
        }
    }
    for (size_t s = 0; s < cs.size(); ++s) {
        P2_HIP(cs[s]->ctx, hipSetDevice(cs[s]->ctx->device));
        P2_HIP(cs[s]->ctx, hipEventRecord(cs[s]->ev_done[slot], cs[s]->comm_stream));
    }
    return P2HOT_OK;
}

static int gather_wait(std::vector<p2hot_comm *> &cs, size_t slot) {
    if ((cs[0]->world == 1 && cs[0]->kind != p2hot_comm::RCCL) || cs[0]->kind == p2hot_comm::CALLBACK) return P2HOT_OK;
    for (auto *c : cs) {
        if (slot >= c->ev_done.size()) continue;
        P2_HIP(c->ctx, hipSetDevice(c->ctx->device));
        P2_HIP(c->ctx, hipStreamWaitEvent(c->ctx->stream, c->ev_done[slot], 0));
    }
    return P2HOT_OK;
}

// ------------------------------------------------------------------ the sharded commit
struct ShardArgs {  // per local rank
    const u64 *cols_local;  // [c1 - c0][n], stride col_stride: this rank's columns (values or coefficients)
    size_t col_stride;
    u64 *coeffs_all;        // [world * cols_per_rank][n]: row = global column index; complete on return
    u64 *lde;               // [W][rows_per_rank], stride lde_stride
    size_t lde_stride;
    u64 *leaves;            // [rows_per_rank][W] or NULL
    u64 *digests, *cap;     // FULL tree arrays
};

struct ShardPlan {
    size_t W, n, N, rows_per_rank, cols_per_rank, digests_per_rank, cap_per_rank;
    unsigned log_n, rate_bits, cap_height, log_N;
    int world;
};

static int shard_plan(p2hot_ctx *ctx, size_t W, unsigned log_n, unsigned rate_bits, unsigned cap_height, int world, ShardPlan *p,
                      bool by_columns = false) {
    P2_TRY(check_log(ctx, log_n + rate_bits, "commit_sharded"));
    if (W == 0) P2_FAIL(ctx, P2HOT_EINVAL, "commit_sharded: no polynomials (the reference panics on polynomials[0], fri/oracle.rs:90)");
    if (world < 1 || (world & (world - 1))) P2_FAIL(ctx, P2HOT_EINVAL, "commit_sharded: world size %d is not a power of two", world);
    p->W = W;
    p->log_n = log_n;
    p->rate_bits = rate_bits;
    p->cap_height = cap_height;
    p->log_N = log_n + rate_bits;
    p->n = (size_t)1 << log_n;
    p->N = p->n << rate_bits;
    p->world = world;
    if (cap_height > p->log_N) P2_FAIL(ctx, P2HOT_EINVAL, "commit_sharded: cap_height > log2(N) (merkle_tree.rs:195-200)");
    // more ranks than LDE cosets (starky's rate-1/2 traces on 4 or 8 GPUs): the cosets are split into sub-cosets of H_n (sub_bits > 0
    // in sharded_commit_core); what is left as a bound is one row block per rank and whole cap subtrees
    if (!by_columns && (size_t)world > p->N) P2_FAIL(ctx, P2HOT_EINVAL, "commit_sharded: world size %d exceeds the %zu LDE rows", world, p->N);
    if ((size_t)world > ((size_t)1 << cap_height)) P2_FAIL(ctx, P2HOT_EINVAL, "commit_sharded: world size %d exceeds the %u cap subtrees", world, 1u << cap_height);
    p->rows_per_rank = p->N / world;
    p->cols_per_rank = W ? (W + world - 1) / world : 0;
    p->digests_per_rank = p2hot_num_digests(p->log_N, cap_height) / world;
    p->cap_per_rank = ((size_t)1 << cap_height) / world;
    return P2HOT_OK;
}

// all local ranks advance phase by phase; everything is enqueued asynchronously on the ranks' streams
static int sharded_commit_core(std::vector<p2hot_comm *> &cs, std::vector<ShardArgs> &as, const ShardPlan &p, int is_values,
                               int gather_digests, unsigned pipeline_chunks) {
    const size_t L = cs.size(), n = p.n, cpr = p.cols_per_rank, W = p.W;
    const int world = p.world;
    auto setdev = [&](size_t s) -> int {
        P2_HIP(cs[s]->ctx, hipSetDevice(cs[s]->ctx->device));
        return P2HOT_OK;
    };
    if (world == 1 && cs[0]->kind != p2hot_comm::RCCL) {
        P2_TRY(setdev(0));
        return p2hot_commit_dev(cs[0]->ctx, as[0].cols_local, as[0].col_stride, W, p.log_n, p.rate_bits, p.cap_height, is_values, 0, p.N,
                                as[0].coeffs_all, n, as[0].lde, as[0].lde_stride, as[0].leaves, as[0].digests, as[0].cap);
    }
    // 1 + 2. column chunks: this rank's columns -> coefficient form in its slot, then the exchange of that chunk
    const size_t K = cpr ? std::max<size_t>(1, std::min<size_t>(pipeline_chunks ? pipeline_chunks : 1, cpr)) : 0;
    const size_t cpk = K ? (cpr + K - 1) / K : 0;
    // ncclAllGather wants the ranks' slices of one exchange back to back.  Whole-rank slices of coeffs_all are (K = 1); the
    // column chunks of the pipelined exchange are not (rank r's chunk k sits at row r*cpr + lo), so in the all-gather mode
    // they travel through a chunk-major staging block [chunk][rank][columns of the chunk][n] -- the iNTT writes there, the LDE
    // reads from there -- and one strided device copy per chunk files them into coeffs_all (`polynomials`, row = column index)
    const bool rccl_path = cs[0]->kind == p2hot_comm::RCCL || (cs[0]->kind == p2hot_comm::GROUP && cs[0]->group->use_rccl);
    const bool staged = rccl_path && cs[0]->exchange_mode == 1 && K > 1;
    std::vector<u64 *> stage(L, nullptr);
    if (staged)
        for (size_t s = 0; s < L; ++s) {
            P2_TRY(setdev(s));
            P2_TRY(scratch_get(cs[s]->ctx, 3, (size_t)world * cpr * n * 8, (void **)&stage[s]));
        }
    auto slice_at = [&](size_t s, size_t r, size_t lo, size_t hi) -> u64 * {  // rank r's slice of chunk [lo, hi) on local rank s
        return staged ? stage[s] + ((size_t)world * lo + r * (hi - lo)) * n : as[s].coeffs_all + (r * cpr + lo) * n;
    };
    std::vector<std::pair<size_t, size_t>> spans;
    for (size_t k = 0; k < K; ++k) {
        const size_t lo = k * cpk, hi = std::min((k + 1) * cpk, cpr);
        if (hi <= lo) continue;
        for (size_t s = 0; s < L; ++s) {
            P2_TRY(setdev(s));
            p2hot_ctx *ctx = cs[s]->ctx;
            const size_t rank = (size_t)cs[s]->rank;
            const size_t c0 = std::min(W, rank * cpr), c1 = std::min(W, c0 + cpr), mine = c1 - c0;
            const size_t valid = lo < mine ? std::min(hi, mine) - lo : 0;  // my real (non-padding) columns in the chunk
            u64 *slot = slice_at(s, rank, lo, hi);
            for (size_t j = 0; j < valid; ++j)
                P2_HIP(ctx, hipMemcpyAsync(slot + j * n, as[s].cols_local + (lo + j) * as[s].col_stride, n * 8, hipMemcpyDeviceToDevice,
                                           ctx->stream));
            if (valid && is_values) P2_TRY(ntt_natural(ctx, slot, valid, n, p.log_n, true));
        }
        std::vector<u64 *> base(L);
        std::vector<size_t> offs((size_t)world);
        for (size_t s = 0; s < L; ++s) base[s] = staged ? stage[s] : as[s].coeffs_all;
        for (int r = 0; r < world; ++r) offs[(size_t)r] = (size_t)(slice_at(0, (size_t)r, lo, hi) - base[0]) * 8;
        P2_TRY(gather_start(cs, base, offs, (hi - lo) * n * 8, spans.size()));
        spans.emplace_back(lo, hi);
    }
    // 3. LDE of every gathered chunk for this rank's coset rows.  world > 2^rate_bits: rows_per_rank = n' < n, one sub-coset per rank
    unsigned sub_bits = 0;
    while ((p.rows_per_rank << sub_bits) < n) ++sub_bits;
    const size_t np = n >> sub_bits;
    std::vector<u64> sub_c(L, 1);
    if (sub_bits)
        for (size_t s = 0; s < L; ++s) {  // block b of the committed order is the coset j = bitrev(b); c_j = (g * w_N^j)^n'
            const u64 shift = gl::mul(gl::COSET_SHIFT, gl::pow(gl::root_of_unity(p.log_N), bitrev_sz((size_t)cs[s]->rank, p.rate_bits + sub_bits)));
            sub_c[s] = gl::canon(gl::pow(shift, np));
        }
    for (size_t k = 0; k < spans.size(); ++k) {
        P2_TRY(gather_wait(cs, k));
        const size_t lo = spans[k].first, hi = spans[k].second;
        for (size_t s = 0; s < L; ++s) {
            P2_TRY(setdev(s));
            const size_t row_begin = (size_t)cs[s]->rank * p.rows_per_rank;
            for (int r = 0; r < world; ++r) {
                const size_t cb = (size_t)r * cpr + lo, ce = std::min(std::min((size_t)r * cpr + hi, W), ((size_t)r + 1) * cpr);
                if (ce <= cb) continue;
                if (sub_bits == 0) {
                    P2_TRY(p2hot_coset_lde_dev(cs[s]->ctx, slice_at(s, (size_t)r, lo, hi), ce - cb, n, p.log_n, p.rate_bits, gl::COSET_SHIFT, row_begin,
                                               p.rows_per_rank, as[s].lde + cb * as[s].lde_stride, as[s].lde_stride));
                    continue;
                }
                // sub-coset mode: this rank's one row block is the coset (g * w_N^j) * H_{n'}; fold the polynomials mod x^n' - c_j, then
                // the same LDE kernels at (log_n - sub_bits, rate_bits + sub_bits) place it (same N, same shift formula, same block order)
                p2hot_ctx *ctx = cs[s]->ctx;
                u64 *folded;
                P2_TRY(scratch_get(ctx, 0, (ce - cb) * np * 8, (void **)&folded));
                for (size_t c = 0; c < ce - cb; c += 65535) {
                    const size_t cnt = std::min<size_t>(65535, ce - cb - c);
                    P2HOT_LAUNCH(ntt::fold_mod_kernel, dim3((unsigned)cdiv(np, 256), (unsigned)cnt), dim3(256), 0, ctx->stream,
                                 slice_at(s, (size_t)r, lo, hi) + c * n, n, folded + c * np, np, p.log_n - sub_bits, 1u << sub_bits, sub_c[s]);
                    P2_LAUNCH_CHECK(ctx);
                }
                P2_TRY(p2hot_coset_lde_dev(ctx, folded, ce - cb, np, p.log_n - sub_bits, p.rate_bits + sub_bits, gl::COSET_SHIFT, row_begin, p.rows_per_rank,
                                           as[s].lde + cb * as[s].lde_stride, as[s].lde_stride));
            }
            if (staged)  // the chunk's `world` slices -> rows r*cpr + lo .. of coeffs_all: one strided copy
                P2_HIP(cs[s]->ctx, hipMemcpy2DAsync(as[s].coeffs_all + lo * n, cpr * n * 8, stage[s] + (size_t)world * lo * n, (hi - lo) * n * 8,
                                                    (hi - lo) * n * 8, (size_t)world, hipMemcpyDeviceToDevice, cs[s]->ctx->stream));
        }
    }
    // leaf sponge + Merkle levels of this rank's rows (whole cosets, whole cap subtrees), straight from the column-major LDE
    for (size_t s = 0; s < L; ++s) {
        P2_TRY(setdev(s));
        const size_t row_begin = (size_t)cs[s]->rank * p.rows_per_rank;
        P2_TRY(p2hot_merkle_dev(cs[s]->ctx, as[s].lde, 0, as[s].lde_stride, W, p.log_N, p.cap_height, row_begin, p.rows_per_rank,
                                as[s].digests, as[s].cap));
        if (as[s].leaves && W) P2_TRY(p2hot_transpose_dev(cs[s]->ctx, as[s].lde, as[s].lde_stride, W, p.rows_per_rank, as[s].leaves));
    }
    // 4. the cap entries (always) and the digest slices (on request): contiguous per-rank slices of the full arrays
    const size_t slot0 = spans.size();
    {
        std::vector<u64 *> base(L);
        std::vector<size_t> offs((size_t)world);
        for (size_t s = 0; s < L; ++s) base[s] = as[s].cap;
        for (int r = 0; r < world; ++r) offs[(size_t)r] = (size_t)r * p.cap_per_rank * 32;
        P2_TRY(gather_start(cs, base, offs, p.cap_per_rank * 32, slot0));
        if (gather_digests && p.digests_per_rank) {
            for (size_t s = 0; s < L; ++s) base[s] = as[s].digests;
            for (int r = 0; r < world; ++r) offs[(size_t)r] = (size_t)r * p.digests_per_rank * 32;
            P2_TRY(gather_start(cs, base, offs, p.digests_per_rank * 32, slot0 + 1));
            P2_TRY(gather_wait(cs, slot0 + 1));
        }
        P2_TRY(gather_wait(cs, slot0));
    }
    return P2HOT_OK;
}

// process-per-GPU entry: device pointers, asynchronous on the context's stream (+ the communicator's stream)
extern "C" int p2hot_commit_sharded_dev(p2hot_ctx *ctx, p2hot_comm *comm, const uint64_t *d_cols_local, size_t col_stride, size_t W,
                                        unsigned log_n, unsigned rate_bits, unsigned cap_height, int is_values, int gather_digests,
                                        unsigned pipeline_chunks, uint64_t *d_coeffs_all, uint64_t *d_lde, size_t lde_stride,
                                        uint64_t *d_leaves, uint64_t *d_digests, uint64_t *d_cap) {
    DeviceGuard restore_caller_device_(0);  // ranks' devices are visited below; the caller's comes back on return
    if (!ctx) return P2HOT_EINVAL;
    if (!comm || comm->ctx != ctx || comm->kind == p2hot_comm::GROUP) P2_FAIL(ctx, P2HOT_EINVAL, "commit_sharded: the communicator belongs to another context");
    ShardPlan p;
    P2_TRY(shard_plan(ctx, W, log_n, rate_bits, cap_height, comm->world, &p));
    const size_t c0 = std::min(W, (size_t)comm->rank * p.cols_per_rank), c1 = std::min(W, c0 + p.cols_per_rank);
    if (W && (!d_coeffs_all || !d_lde || (c1 > c0 && !d_cols_local) || col_stride < p.n || lde_stride < p.rows_per_rank))
        P2_FAIL(ctx, P2HOT_EINVAL, "commit_sharded: null buffer or stride too small");
    if (!d_cap || (p2hot_num_digests(p.log_N, cap_height) && !d_digests)) P2_FAIL(ctx, P2HOT_EINVAL, "commit_sharded: null tree output");
    std::vector<p2hot_comm *> cs{comm};
    std::vector<ShardArgs> as{ShardArgs{d_cols_local, col_stride, d_coeffs_all, d_lde, lde_stride, d_leaves, d_digests, d_cap}};
    return sharded_commit_core(cs, as, p, is_values, gather_digests, pipeline_chunks);
}

// Preflight of a process-per-GPU communicator: every rank contributes a `bytes`-sized slice of a known pattern, the slices
// are all-gathered exactly as the commit does it (grouped broadcasts / the caller's transport, the communication stream, the
// event hand-over), and every rank checks every slice.  A broken binding, a missing peer path or a transport that returns
// early fails HERE with a named cause instead of as a wrong cap (or a hang) inside the timed region.
// One pattern round trip through gather_start / gather_wait on `cs` (1 entry: a process-per-GPU communicator; world entries:
// the single-process group): every rank contributes a `bytes`-sized slice, every rank checks every slice.  *ms (optional): the
// best host-side time of `reps` exchanges (stream idle to stream idle).
static int exchange_roundtrip(std::vector<p2hot_comm *> &cs, std::vector<u64 *> &d, size_t bytes, int reps, double *ms) {
    const size_t world = (size_t)cs[0]->world, words = bytes / 8;
    auto pat = [](size_t r, size_t i) { return ((u64)(r + 1) << 40) ^ (i * 0x9E3779B97F4A7C15ull); };
    std::vector<size_t> offs(world);
    for (size_t r = 0; r < world; ++r) offs[r] = r * bytes;
    double best = 1e30;
    for (int rep = 0; rep < reps; ++rep) {
        std::vector<std::vector<u64>> host(cs.size());
        for (size_t s = 0; s < cs.size(); ++s) {
            p2hot_ctx *ctx = cs[s]->ctx;
            P2_HIP(ctx, hipSetDevice(ctx->device));
            host[s].assign(world * words, ~0ull);
            for (size_t i = 0; i < words; ++i) host[s][(size_t)cs[s]->rank * words + i] = pat((size_t)cs[s]->rank, i);
            P2_HIP(ctx, hipMemcpyAsync(d[s], host[s].data(), world * bytes, hipMemcpyHostToDevice, ctx->stream));
            P2_HIP(ctx, stream_sync(ctx));
        }
        const auto t0 = std::chrono::steady_clock::now();
        P2_TRY(gather_start(cs, d, offs, bytes, 0));
        P2_TRY(gather_wait(cs, 0));
        for (size_t s = 0; s < cs.size(); ++s) {
            P2_HIP(cs[s]->ctx, hipSetDevice(cs[s]->ctx->device));
            P2_HIP(cs[s]->ctx, stream_sync(cs[s]->ctx));
        }
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        for (size_t s = 0; s < cs.size(); ++s) {
            p2hot_ctx *ctx = cs[s]->ctx;
            P2_HIP(ctx, hipSetDevice(ctx->device));
            P2_HIP(ctx, hipMemcpyAsync(host[s].data(), d[s], world * bytes, hipMemcpyDeviceToHost, ctx->stream));
            P2_HIP(ctx, stream_sync(ctx));
            for (size_t r = 0; r < world; ++r)
                for (size_t i = 0; i < words; ++i)
                    if (host[s][r * words + i] != pat(r, i))
                        P2_FAIL(cs[0]->ctx, P2HOT_ECOMM, "comm_selftest: rank %d did not receive rank %zu's slice (word %zu) through %s: the %s transport delivered nothing or stale data",
                                cs[s]->rank, r, i, cs[0]->exchange_mode == 1 ? "ncclAllGather" : "the grouped broadcasts",
                                cs[0]->kind == p2hot_comm::CALLBACK ? "caller-supplied" : "RCCL");
        }
    }
    if (ms) *ms = best;
    return P2HOT_OK;
}

// Both RCCL exchange forms are checked, timed, and the faster one kept (unless P2HOT_EXCHANGE pins one).  `decide`: how the ranks
// agree on ONE answer when each of them measured (process-per-GPU: rank 0's verdict is broadcast); null in the group (one thread).
static int pick_exchange_mode(std::vector<p2hot_comm *> &cs, std::vector<u64 *> &d, size_t bytes, const std::function<int(int *)> &decide) {
    const bool rccl_path = cs[0]->kind == p2hot_comm::RCCL || (cs[0]->kind == p2hot_comm::GROUP && cs[0]->group->use_rccl);
    auto set_mode = [&](int m) {
        for (auto *c : cs) c->exchange_mode = m;
    };
    if (!rccl_path || !rccl::api().AllGather) {
        set_mode(0);
        return exchange_roundtrip(cs, d, bytes, 1, nullptr);
    }
    const int pinned = exchange_mode_from_env();
    double ms[2] = {0, 0};
    for (int m = 0; m < 2; ++m) {
        if (pinned >= 0 && m != pinned) continue;
        set_mode(m);
        P2_TRY(exchange_roundtrip(cs, d, bytes, pinned >= 0 ? 1 : 3, &ms[m]));
    }
    int mode = pinned >= 0 ? pinned : (ms[1] <= ms[0] ? 1 : 0);
    if (pinned < 0 && decide) P2_TRY(decide(&mode));
    set_mode(mode);
    return P2HOT_OK;
}

// Preflight of a process-per-GPU communicator: every rank contributes a `bytes`-sized slice of a known pattern, the slices
// are all-gathered exactly as the commit does it (grouped broadcasts and ncclAllGather / the caller's transport, the
// communication stream, the event hand-over), and every rank checks every slice.  A broken binding, a missing peer path or a
// transport that returns early fails HERE with a named cause instead of as a wrong cap (or a hang) inside the timed region.
// On an RCCL communicator both exchange forms are timed and the faster one is kept for the commits that follow
// (p2hot_comm_exchange_mode; rank 0's verdict is the communicator's, so every rank posts the same collectives).
extern "C" int p2hot_comm_selftest(p2hot_comm *comm, size_t bytes) {
    if (!comm || !comm->ctx) return P2HOT_EINVAL;
    p2hot_ctx *ctx = comm->ctx;
    DeviceGuard dev_guard_(ctx);
    if (comm->kind == p2hot_comm::GROUP) P2_FAIL(ctx, P2HOT_EINVAL, "comm_selftest: use it on a process-per-GPU communicator");
    if (bytes == 0 || bytes % 8) P2_FAIL(ctx, P2HOT_EINVAL, "comm_selftest: bytes must be a positive multiple of 8");
    const size_t world = (size_t)comm->world;
    u64 *d = nullptr;
    P2_HIP(ctx, hipMalloc((void **)&d, world * bytes + 64));
    std::vector<p2hot_comm *> cs{comm};
    std::vector<u64 *> base{d};
    auto decide = [&](int *mode) -> int {  // rank 0's measurement decides for everybody: 8 bytes broadcast from rank 0
        rccl::Api &a = rccl::api();
        u64 *flag = d + world * (bytes / 8);
        u64 v = (u64)*mode;
        P2_HIP(ctx, hipMemcpyAsync(flag, &v, 8, hipMemcpyHostToDevice, ctx->stream));
        P2_HIP(ctx, stream_sync(ctx));
        P2_NCCL(ctx, a.GroupStart());
        int rc = a.Broadcast(flag, flag, 8, rccl::kUint8, 0, comm->nccl, comm->comm_stream);
        P2_NCCL(ctx, a.GroupEnd());
        if (rc != 0) P2_FAIL(ctx, P2HOT_ECOMM, "comm_selftest: ncclBroadcast of the exchange mode: %s", a.GetErrorString(rc));
        P2_HIP(ctx, hipStreamSynchronize(comm->comm_stream));
        P2_HIP(ctx, hipMemcpyAsync(&v, flag, 8, hipMemcpyDeviceToHost, ctx->stream));
        P2_HIP(ctx, stream_sync(ctx));
        *mode = (int)v;
        return P2HOT_OK;
    };
    int rc = pick_exchange_mode(cs, base, bytes, comm->kind == p2hot_comm::RCCL && comm->world > 1 ? std::function<int(int *)>(decide) : nullptr);
    (void)hipFree(d);
    return rc;
}

extern "C" int p2hot_comm_exchange_mode(const p2hot_comm *c) { return c ? c->exchange_mode : -1; }

// Which RCCL the library is bound to in THIS process: the file and its ncclGetVersion code.  The two deployment modes bind
// different copies -- under torch.distributed.run the one PyTorch already loaded (RTLD_NOLOAD: two RCCLs never meet in one
// process), in a patched plonky2 (no torch) /opt/rocm's by path -- and a first multi-GPU run should say which one it used.
extern "C" int p2hot_rccl_info(char *path_out, size_t path_cap, int *version_out) {
    rccl::Api &a = rccl::api();
    if (path_out && path_cap) path_out[0] = 0;
    if (version_out) *version_out = 0;
    if (!a.ok) return P2HOT_ECOMM;
#ifdef P2HOT_EMU
    if (path_out && path_cap) snprintf(path_out, path_cap, "%s", "tests/emu fake RCCL");
#else
    Dl_info info{};
    if (path_out && path_cap && dladdr(reinterpret_cast<void *>(a.GetUniqueId), &info) && info.dli_fname)
        snprintf(path_out, path_cap, "%s", info.dli_fname);
    if (version_out)
        if (auto get_version = reinterpret_cast<int (*)(int *)>(dlsym(a.lib, "ncclGetVersion"))) (void)get_version(version_out);
#endif
    return P2HOT_OK;
}

extern "C" int p2hot_shard_columns(size_t W, int world, int rank, size_t *first, size_t *count) {
    if (world < 1 || rank < 0 || rank >= world || !first || !count) return P2HOT_EINVAL;
    const size_t cpr = W ? (W + (size_t)world - 1) / (size_t)world : 0;
    const size_t c0 = std::min(W, (size_t)rank * cpr), c1 = std::min(W, c0 + cpr);
    *first = c0;
    *count = c1 - c0;
    return P2HOT_OK;
}

// ------------------------------------------------------------------ the column-sharded fallback (last row)
// Columns are the unit, as BASELINE's north_star words it: rank r runs the iNTT AND the whole rate-1/B LDE (all N rows)
// of its ceil(W/G) columns, then the LDE matrix is re-partitioned to row blocks by an all-to-all (the leaf sponge chains
// across the columns of a row, hash/hashing.rs:118-141, so a rank needs every column of the rows it hashes) and each rank
// hashes its rows and builds its cap subtrees as in the coset scheme.  Exchange: W*N*8 bytes (2^rate_bits times the
// coset scheme's W*n*8 of coefficients) -- strictly more traffic, which is why it is the fallback, not the default.  In
// return it has no constraint G <= 2^rate_bits: starky's rate-1/2 traces (two cosets) can use 4 or 8 GPUs.
// Implemented for the single-process group, where the all-to-all is a set of strided peer copies over xGMI.
struct ColShardArgs {        // per local rank
    const u64 *cols_local;   // [cnt][n]
    u64 *coeffs_local;       // [cnt][n] out: this rank's columns in coefficient form
    u64 *lde_cols;           // [cnt][N] scratch: this rank's columns over ALL rows
    u64 *lde;                // [W][rows_per_rank] out: all columns over this rank's rows
    u64 *leaves, *digests, *cap;
};

static int sharded_commit_columns_core(std::vector<p2hot_comm *> &cs, std::vector<ColShardArgs> &as, const ShardPlan &p, int is_values) {
    const size_t L = cs.size(), n = p.n, N = p.N, cpr = p.cols_per_rank, W = p.W, rpr = p.rows_per_rank;
    auto cols_of = [&](size_t r, size_t *c0, size_t *cnt) {
        *c0 = std::min(W, r * cpr);
        *cnt = std::min(W, *c0 + cpr) - *c0;
    };
    for (size_t s = 0; s < L; ++s) {  // iNTT + full LDE of the rank's own columns
        p2hot_ctx *ctx = cs[s]->ctx;
        P2_HIP(ctx, hipSetDevice(ctx->device));
        size_t c0, cnt;
        cols_of(s, &c0, &cnt);
        if (cnt == 0) continue;
        P2_HIP(ctx, hipMemcpyAsync(as[s].coeffs_local, as[s].cols_local, cnt * n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (is_values) P2_TRY(ntt_natural(ctx, as[s].coeffs_local, cnt, n, p.log_n, true));
        P2_TRY(p2hot_coset_lde_dev(ctx, as[s].coeffs_local, cnt, n, p.log_n, p.rate_bits, gl::COSET_SHIFT, 0, N, as[s].lde_cols, N));
    }
    // all-to-all: rank d pulls, from every rank r, the rows [d*rpr, (d+1)*rpr) of r's columns into lde[c0_r ..][0 .. rpr)
    for (size_t s = 0; s < L; ++s) {
        P2_HIP(cs[s]->ctx, hipSetDevice(cs[s]->ctx->device));
        P2_TRY(comm_events(cs[s], 1));
        P2_HIP(cs[s]->ctx, hipEventRecord(cs[s]->ev_ready[0], cs[s]->ctx->stream));
    }
    for (size_t d = 0; d < L; ++d) {
        p2hot_ctx *ctx = cs[d]->ctx;
        P2_HIP(ctx, hipSetDevice(ctx->device));
        for (size_t r = 0; r < L; ++r) {
            size_t c0, cnt;
            cols_of(r, &c0, &cnt);
            if (cnt == 0) continue;
            hipStream_t st = cs[d]->comm_stream;
            P2_HIP(ctx, hipStreamWaitEvent(st, cs[r]->ev_ready[0], 0));
            P2_HIP(ctx, hipMemcpy2DAsync(as[d].lde + c0 * rpr, rpr * 8, as[r].lde_cols + d * rpr, N * 8, rpr * 8, cnt, hipMemcpyDefault, st));
        }
        P2_HIP(ctx, hipEventRecord(cs[d]->ev_done[0], cs[d]->comm_stream));
        P2_HIP(ctx, hipStreamWaitEvent(ctx->stream, cs[d]->ev_done[0], 0));
    }
    for (size_t s = 0; s < L; ++s) {  // leaf sponge + Merkle levels of the rank's rows
        p2hot_ctx *ctx = cs[s]->ctx;
        P2_HIP(ctx, hipSetDevice(ctx->device));
        P2_TRY(p2hot_merkle_dev(ctx, as[s].lde, 0, rpr, W, p.log_N, p.cap_height, s * rpr, rpr, as[s].digests, as[s].cap));
        if (as[s].leaves && W) P2_TRY(p2hot_transpose_dev(ctx, as[s].lde, rpr, W, rpr, as[s].leaves));
    }
    std::vector<u64 *> base(L);
    std::vector<size_t> offs(L);
    for (size_t s = 0; s < L; ++s) {
        base[s] = as[s].cap;
        offs[s] = s * p.cap_per_rank * 32;
    }
    P2_TRY(gather_start(cs, base, offs, p.cap_per_rank * 32, 1));
    return gather_wait(cs, 1);
}

// ------------------------------------------------------------------ single process, all GPUs of the node (the Rust prover's mode)
extern "C" int p2hot_group_create(int n_gpus, const int *devices, p2hot_group **out) {
    DeviceGuard restore_caller_device_(0);  // the loop below visits every rank's GPU
    if (!out || n_gpus < 1 || (n_gpus & (n_gpus - 1))) return P2HOT_EINVAL;
    *out = nullptr;
    std::unique_ptr<p2hot_group> g(new p2hot_group());
    bool distinct = true;
    for (int i = 0; i < n_gpus; ++i) {
        g->devices.push_back(devices ? devices[i] : i);
        for (int j = 0; j < i; ++j) distinct = distinct && g->devices[(size_t)j] != g->devices[(size_t)i];
    }
    auto fail = [&](int rc) {
        for (auto *c : g->comm) p2hot_comm_destroy(c);
        for (auto *c : g->ctx) p2hot_ctx_destroy(c);
        for (size_t i = 0; i < g->streams.size(); ++i)
            if (g->streams[i]) {
                (void)hipSetDevice(g->devices[i]);
                (void)hipStreamDestroy(g->streams[i]);
            }
        return rc;
    };
    for (int i = 0; i < n_gpus; ++i) {
        hipStream_t st = nullptr;
        if (hipSetDevice(g->devices[(size_t)i]) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return fail(P2HOT_EHIP);
        g->streams.push_back(st);
        p2hot_ctx *c = nullptr;
        int rc = p2hot_ctx_create(g->devices[(size_t)i], st, &c);
        if (c) g->ctx.push_back(c);
        if (rc != P2HOT_OK) return fail(rc);
        p2hot_comm *cm = new p2hot_comm();
        cm->ctx = c;
        cm->rank = i;
        cm->world = n_gpus;
        cm->kind = p2hot_comm::GROUP;
        cm->group = g.get();
        g->comm.push_back(cm);
    }
    // distinct devices: RCCL over xGMI (ncclCommInitAll); a repeated device (one-GPU test boxes) or P2HOT_GROUP_PEER_COPY=1: peer copies
    const char *env = getenv("P2HOT_GROUP_PEER_COPY");
    g->use_rccl = n_gpus > 1 && distinct && !(env && env[0] == '1');
    if (g->use_rccl) {
        rccl::Api &a = rccl::api();
        if (!a.ok) {
            g->ctx[0]->err = "group_create: " + a.err;
            return fail(P2HOT_ECOMM);
        }
        std::vector<rccl::Comm> comms((size_t)n_gpus);
        if (a.CommInitAll(comms.data(), n_gpus, g->devices.data()) != 0) return fail(P2HOT_ECOMM);
        for (int i = 0; i < n_gpus; ++i) g->comm[(size_t)i]->nccl = comms[(size_t)i];
        // both exchange forms once over every rank (1 MB slices): checked, timed, the faster one kept (P2HOT_EXCHANGE pins one)
        const size_t bytes = (size_t)1 << 20;
        std::vector<u64 *> d((size_t)n_gpus, nullptr);
        int rc = P2HOT_OK;
        for (int i = 0; i < n_gpus && rc == P2HOT_OK; ++i) {
            (void)hipSetDevice(g->devices[(size_t)i]);
            if (hipMalloc((void **)&d[(size_t)i], (size_t)n_gpus * bytes) != hipSuccess) rc = P2HOT_ENOMEM;
        }
        if (rc == P2HOT_OK) rc = pick_exchange_mode(g->comm, d, bytes, nullptr);
        for (int i = 0; i < n_gpus; ++i)
            if (d[(size_t)i]) {
                (void)hipSetDevice(g->devices[(size_t)i]);
                (void)hipFree(d[(size_t)i]);
            }
        if (rc != P2HOT_OK) return fail(rc);
    }
    if (!g->use_rccl && distinct)
        for (int i = 0; i < n_gpus; ++i) {  // peer copies between distinct devices need peer access
            (void)hipSetDevice(g->devices[(size_t)i]);
            for (int j = 0; j < n_gpus; ++j)
                if (j != i) (void)hipDeviceEnablePeerAccess(g->devices[(size_t)j], 0);
            (void)hipGetLastError();
        }
    *out = g.release();
    return P2HOT_OK;
}

extern "C" void p2hot_group_destroy(p2hot_group *g) {
    DeviceGuard restore_caller_device_(0);  // the loop below visits every rank's GPU
    if (!g) return;
    for (auto *c : g->comm) p2hot_comm_destroy(c);
    for (auto *c : g->ctx) p2hot_ctx_destroy(c);
    for (size_t i = 0; i < g->streams.size(); ++i)
        if (g->streams[i]) {
            (void)hipSetDevice(g->devices[i]);
            (void)hipStreamDestroy(g->streams[i]);
        }
    delete g;
}

extern "C" int p2hot_group_size(const p2hot_group *g) { return g ? (int)g->ctx.size() : 0; }
extern "C" p2hot_ctx *p2hot_group_ctx(p2hot_group *g, int i) { return (g && i >= 0 && (size_t)i < g->ctx.size()) ? g->ctx[(size_t)i] : nullptr; }
extern "C" int p2hot_group_uses_rccl(const p2hot_group *g) { return g && g->use_rccl ? 1 : 0; }
extern "C" int p2hot_group_exchange_mode(const p2hot_group *g) { return g && !g->comm.empty() ? g->comm[0]->exchange_mode : -1; }
extern "C" const char *p2hot_group_last_error(const p2hot_group *g) {
    if (!g) return "null group";
    for (auto *c : g->ctx)
        if (!c->err.empty()) return c->err.c_str();
    return "";
}

// a PolynomialBatch whose LDE rows and cap subtrees are spread over the group's GPUs; rank r = rows [r*N/G, (r+1)*N/G)
struct p2hot_sharded_batch {
    p2hot_group *g;
    ShardPlan plan;
    std::vector<u64 *> coeffs_all, lde, digests;  // per rank; in the coset mode every rank holds ALL coefficients
    bool by_columns = false;                       // column-sharded: rank r holds only its own columns' coefficients
};

extern "C" void p2hot_sharded_batch_free(p2hot_sharded_batch *b) {
    DeviceGuard restore_caller_device_(0);  // the loop below visits every rank's GPU
    if (!b) return;
    for (size_t s = 0; s < b->g->ctx.size(); ++s) {
        p2hot_ctx *ctx = b->g->ctx[s];
        (void)hipSetDevice(ctx->device);
        (void)stream_sync(ctx);
        pool_release(ctx, b->coeffs_all[s]);
        pool_release(ctx, b->lde[s]);
        pool_release(ctx, b->digests[s]);
    }
    delete b;
}

// from_values / from_coeffs over all GPUs of the group, HOST pointers (what p2hot_commit is for one GPU).
// coeffs_out [W][n], leaves_out [N][W], digests_out, cap_out: caller-allocated or NULL; assembled from the owning ranks.
extern "C" int p2hot_group_commit(p2hot_group *g, const uint64_t *const *cols, size_t W, unsigned log_n, unsigned rate_bits,
                                  unsigned cap_height, int is_values, int shard_mode, unsigned pipeline_chunks, uint64_t *coeffs_out,
                                  uint64_t *leaves_out, uint64_t *digests_out, uint64_t *cap_out, p2hot_sharded_batch **handle_out) {
    DeviceGuard restore_caller_device_(0);  // ranks' devices are visited below; the caller's comes back on return
    if (!g) return P2HOT_EINVAL;
    p2hot_ctx *ctx0 = g->ctx[0];
    P2_ENTER(ctx0);
    if (handle_out) *handle_out = nullptr;
    const int world = (int)g->ctx.size();
    if (shard_mode != P2HOT_SHARD_COSETS && shard_mode != P2HOT_SHARD_COLUMNS) P2_FAIL(ctx0, P2HOT_EINVAL, "group_commit: unknown shard mode %d", shard_mode);
    const bool by_columns = shard_mode == P2HOT_SHARD_COLUMNS && world > 1;
    ShardPlan p;
    P2_TRY(shard_plan(ctx0, W, log_n, rate_bits, cap_height, world, &p, by_columns));
    if (W && !cols) P2_FAIL(ctx0, P2HOT_EINVAL, "group_commit: null column table");
    for (size_t c = 0; c < W; ++c)
        if (!cols[c]) P2_FAIL(ctx0, P2HOT_EINVAL, "group_commit: column %zu is null", c);
    const size_t n = p.n, nd = p2hot_num_digests(p.log_N, cap_height), cap_words = (size_t)4 << cap_height;
    const size_t L = (size_t)world;
    std::vector<std::unique_ptr<PoolBuf>> b_cols, b_co, b_lde, b_leaves, b_dig, b_cap, b_ldecols;
    std::vector<ShardArgs> as(L);
    std::vector<ColShardArgs> cas(L);
    auto alloc = [&](std::vector<std::unique_ptr<PoolBuf>> &v, p2hot_ctx *ctx, size_t bytes) -> int {
        v.emplace_back(new PoolBuf(ctx));
        return pool_alloc(ctx, bytes, &v.back()->p);
    };
    int rc = P2HOT_OK;
    auto body = [&]() -> int {
        for (size_t s = 0; s < L; ++s) {
            p2hot_ctx *ctx = g->ctx[s];
            P2_HIP(ctx, hipSetDevice(ctx->device));
            const size_t c0 = std::min(W, s * p.cols_per_rank), c1 = std::min(W, c0 + p.cols_per_rank);
            P2_TRY(alloc(b_cols, ctx, std::max<size_t>(1, c1 - c0) * n * 8));
            P2_TRY(alloc(b_co, ctx, std::max<size_t>(1, by_columns ? p.cols_per_rank : L * p.cols_per_rank) * n * 8));
            P2_TRY(alloc(b_ldecols, ctx, by_columns ? std::max<size_t>(1, c1 - c0) * p.N * 8 : 8));
            P2_TRY(alloc(b_lde, ctx, std::max<size_t>(1, W) * p.rows_per_rank * 8));
            P2_TRY(alloc(b_leaves, ctx, leaves_out ? std::max<size_t>(1, W) * p.rows_per_rank * 8 : 8));
            P2_TRY(alloc(b_dig, ctx, std::max<size_t>(1, nd) * 32));
            P2_TRY(alloc(b_cap, ctx, cap_words * 8));
            for (size_t c = c0; c < c1; ++c)  // each GPU receives only the columns it transforms
                P2_HIP(ctx, hipMemcpyAsync(b_cols[s]->u() + (c - c0) * n, cols[c], n * 8, hipMemcpyHostToDevice, ctx->stream));
            as[s] = ShardArgs{b_cols[s]->u(), n, b_co[s]->u(), b_lde[s]->u(), p.rows_per_rank, leaves_out ? b_leaves[s]->u() : nullptr,
                              b_dig[s]->u(), b_cap[s]->u()};
            cas[s] = ColShardArgs{b_cols[s]->u(), b_co[s]->u(), b_ldecols[s]->u(), b_lde[s]->u(), leaves_out ? b_leaves[s]->u() : nullptr,
                                  b_dig[s]->u(), b_cap[s]->u()};
        }
        if (by_columns) {
            P2_TRY(sharded_commit_columns_core(g->comm, cas, p, is_values));
            // the coefficients live with the rank that owns the column
            for (size_t s = 0; s < L && coeffs_out; ++s) {
                p2hot_ctx *ctx = g->ctx[s];
                P2_HIP(ctx, hipSetDevice(ctx->device));
                const size_t c0 = std::min(W, s * p.cols_per_rank), c1 = std::min(W, c0 + p.cols_per_rank);
                if (c1 == c0) continue;
                if (!is_values) {
                    P2HOT_LAUNCH(ntt::canon_kernel, dim3(cdiv((c1 - c0) * n, 256)), dim3(256), 0, ctx->stream, cas[s].coeffs_local, (c1 - c0) * n);
                    P2_LAUNCH_CHECK(ctx);
                }
                P2_HIP(ctx, hipMemcpyAsync(coeffs_out + c0 * n, cas[s].coeffs_local, (c1 - c0) * n * 8, hipMemcpyDeviceToHost, ctx->stream));
            }
        } else {
            P2_TRY(sharded_commit_core(g->comm, as, p, is_values, /*gather_digests=*/0, pipeline_chunks ? pipeline_chunks : 8));
        }
        // results: the coefficients and the cap from rank 0 (complete everywhere), digests / leaves from their owners
        P2_HIP(ctx0, hipSetDevice(ctx0->device));
        if (coeffs_out && W && !by_columns) {
            if (!is_values) {
                P2HOT_LAUNCH(ntt::canon_kernel, dim3(cdiv(W * n, 256)), dim3(256), 0, ctx0->stream, as[0].coeffs_all, W * n);
                P2_LAUNCH_CHECK(ctx0);
            }
            P2_HIP(ctx0, hipMemcpyAsync(coeffs_out, as[0].coeffs_all, W * n * 8, hipMemcpyDeviceToHost, ctx0->stream));
        }
        if (cap_out) P2_HIP(ctx0, hipMemcpyAsync(cap_out, as[0].cap, cap_words * 8, hipMemcpyDeviceToHost, ctx0->stream));
        for (size_t s = 0; s < L; ++s) {
            p2hot_ctx *ctx = g->ctx[s];
            P2_HIP(ctx, hipSetDevice(ctx->device));
            if (digests_out && p.digests_per_rank)
                P2_HIP(ctx, hipMemcpyAsync(digests_out + 4 * s * p.digests_per_rank, as[s].digests + 4 * s * p.digests_per_rank,
                                           p.digests_per_rank * 32, hipMemcpyDeviceToHost, ctx->stream));
            if (leaves_out && W)
                P2_HIP(ctx, hipMemcpyAsync(leaves_out + s * p.rows_per_rank * W, as[s].leaves, p.rows_per_rank * W * 8, hipMemcpyDeviceToHost,
                                           ctx->stream));
        }
        return P2HOT_OK;
    };
    rc = body();
    for (size_t s = 0; s < L; ++s) {  // every rank's work is complete (and its pool blocks reusable) on return
        p2hot_ctx *ctx = g->ctx[s];
        (void)hipSetDevice(ctx->device);
        hipError_t e = stream_sync(ctx);
        if (g->comm[s]->comm_stream) (void)hipStreamSynchronize(g->comm[s]->comm_stream);
        if (rc == P2HOT_OK && e != hipSuccess) {
            ctx0->err = std::string("group_commit: ") + hipGetErrorString(e);
            rc = P2HOT_EHIP;
        }
        if (rc == P2HOT_OK && ctx != ctx0 && !ctx->err.empty()) ctx0->err = ctx->err;
    }
    if (rc != P2HOT_OK && ctx0->err.empty())
        for (auto *c : g->ctx)
            if (!c->err.empty()) ctx0->err = c->err;
    if (rc == P2HOT_OK && handle_out) {
        p2hot_sharded_batch *b = new p2hot_sharded_batch{g, p, {}, {}, {}, by_columns};
        for (size_t s = 0; s < L; ++s) {
            b->coeffs_all.push_back(b_co[s]->u());
            b->lde.push_back(b_lde[s]->u());
            b->digests.push_back(b_dig[s]->u());
            b_co[s]->p = b_lde[s]->p = b_dig[s]->p = nullptr;
        }
        *handle_out = b;
    }
    return rc;
}

// MerkleTree::get + merkle_tree_prove for m leaves of a sharded batch: every query is answered by the rank that owns the row
// (a Merkle path below the cap never leaves the cap subtree of its leaf).  One gather + one copy per owning rank.
// rows_out: row q at rows_out + q * row_pitch (W words); paths_out: path q at paths_out + q * path_pitch (layers * 4 words).
static int sharded_open(p2hot_sharded_batch *b, const u64 *leaf_idx, size_t m, u64 *rows_out, size_t row_pitch, u64 *paths_out,
                        size_t path_pitch) {
    p2hot_group *g = b->g;
    p2hot_ctx *ctx0 = g->ctx[0];
    const ShardPlan &p = b->plan;
    const unsigned layers = p.log_N - p.cap_height;
    for (size_t q = 0; q < m; ++q)
        if (leaf_idx[q] >= p.N) P2_FAIL(ctx0, P2HOT_EINVAL, "sharded_batch_open: index %llu out of range", (unsigned long long)leaf_idx[q]);
    for (size_t owner = 0; owner < g->ctx.size(); ++owner) {
        std::vector<size_t> mine;
        std::vector<u64> idx2;  // [local rows ...][global leaves ...]
        for (size_t q = 0; q < m; ++q)
            if (leaf_idx[q] / p.rows_per_rank == owner) mine.push_back(q);
        if (mine.empty()) continue;
        for (size_t q : mine) idx2.push_back(leaf_idx[q] - owner * p.rows_per_rank);
        for (size_t q : mine) idx2.push_back(leaf_idx[q]);
        const size_t k = mine.size();
        p2hot_ctx *ctx = g->ctx[owner];
        P2_HIP(ctx, hipSetDevice(ctx->device));
        PoolBuf d_idx(ctx), d_row(ctx), d_path(ctx);
        P2_TRY(pool_alloc(ctx, 2 * k * 8, &d_idx.p));
        P2_TRY(pool_alloc(ctx, std::max<size_t>(1, k * p.W) * 8, &d_row.p));
        P2_TRY(pool_alloc(ctx, std::max<size_t>(1, k * layers) * 32, &d_path.p));
        std::vector<u64> rows(k * p.W), paths(k * layers * 4);
        int rc = P2HOT_OK;
        auto body = [&]() -> int {
            P2_HIP(ctx, hipMemcpyAsync(d_idx.p, idx2.data(), 2 * k * 8, hipMemcpyHostToDevice, ctx->stream));
            if (rows_out && p.W) {
                P2_TRY(p2hot_gather_rows_dev(ctx, b->lde[owner], p.rows_per_rank, p.rows_per_rank, p.W, d_idx.u(), k, d_row.u()));
                P2_HIP(ctx, hipMemcpyAsync(rows.data(), d_row.p, k * p.W * 8, hipMemcpyDeviceToHost, ctx->stream));
            }
            if (paths_out && layers) {
                P2_TRY(p2hot_merkle_paths_dev(ctx, b->digests[owner], p.log_N, p.cap_height, d_idx.u() + k, k, d_path.u()));
                P2_HIP(ctx, hipMemcpyAsync(paths.data(), d_path.p, k * layers * 32, hipMemcpyDeviceToHost, ctx->stream));
            }
            return P2HOT_OK;
        };
        rc = body();
        hipError_t e = stream_sync(ctx);  // the local vectors and pool blocks are released below
        if (rc != P2HOT_OK) {
            if (ctx != ctx0) ctx0->err = ctx->err;
            return rc;
        }
        if (e != hipSuccess) P2_FAIL(ctx0, P2HOT_EHIP, "sharded_batch_open: %s", hipGetErrorString(e));
        for (size_t t = 0; t < k; ++t) {
            if (rows_out && p.W) std::copy(rows.begin() + t * p.W, rows.begin() + (t + 1) * p.W, rows_out + mine[t] * row_pitch);
            if (paths_out && layers)
                std::copy(paths.begin() + t * layers * 4, paths.begin() + (t + 1) * layers * 4, paths_out + mine[t] * path_pitch);
        }
    }
    return P2HOT_OK;
}

extern "C" int p2hot_sharded_batch_open(p2hot_sharded_batch *b, const uint64_t *leaf_idx, size_t m, uint64_t *rows_out, uint64_t *paths_out) {
    DeviceGuard restore_caller_device_(0);  // ranks' devices are visited below; the caller's comes back on return
    if (!b) return P2HOT_EINVAL;
    p2hot_ctx *ctx0 = b->g->ctx[0];
    P2_ENTER(ctx0);
    if (m == 0) return P2HOT_OK;
    if (!leaf_idx) P2_FAIL(ctx0, P2HOT_EINVAL, "sharded_batch_open: null indices");
    const unsigned layers = b->plan.log_N - b->plan.cap_height;
    return sharded_open(b, leaf_idx, m, rows_out, b->plan.W, paths_out, (size_t)layers * 4);
}

// ------------------------------------------------------------------ openings and prove_openings over sharded batches
// In the coset mode every rank holds all coefficients, so rank 0 runs what needs the polynomials -- the OpeningSet
// evaluations, final_poly, the FRI commit phase (after round 0 it is tiny) and the grind -- and the rows and paths
// of the initial trees come from the ranks that own them.
static int sharded_views(p2hot_group *g, const p2hot_sharded_batch *const *oracles, size_t n_oracles, std::vector<OracleView> *views,
                         unsigned *log_n) {
    p2hot_ctx *ctx0 = g->ctx[0];
    if (!oracles || n_oracles == 0) P2_FAIL(ctx0, P2HOT_EINVAL, "group openings: null oracle list");
    for (size_t o = 0; o < n_oracles; ++o) {
        const p2hot_sharded_batch *B = oracles[o];
        if (!B || B->g != g) P2_FAIL(ctx0, P2HOT_EINVAL, "group openings: oracle %zu is null or belongs to another group", o);
        if (B->by_columns) P2_FAIL(ctx0, P2HOT_EUNSUPPORTED, "group openings: oracle %zu was committed column-sharded (no rank holds all coefficients)", o);
        if (o == 0) *log_n = B->plan.log_n;
        if (B->plan.log_n != *log_n) P2_FAIL(ctx0, P2HOT_EINVAL, "group openings: all oracles must have the same degree");
        views->push_back(OracleView{B->coeffs_all[0], nullptr, nullptr, B->plan.W, B->plan.N});
    }
    return P2HOT_OK;
}

extern "C" int p2hot_group_eval_openings(p2hot_group *g, const p2hot_sharded_batch *const *oracles, size_t n_oracles, const uint64_t *points,
                                         size_t n_points, uint64_t *out) {
    DeviceGuard restore_caller_device_(0);  // ranks' devices are visited below; the caller's comes back on return
    if (!g) return P2HOT_EINVAL;
    p2hot_ctx *ctx = g->ctx[0];
    P2_ENTER(ctx);
    if (n_oracles == 0 || n_points == 0) return P2HOT_OK;
    if (!points || !out) P2_FAIL(ctx, P2HOT_EINVAL, "group_eval_openings: null argument");
    std::vector<OracleView> views;
    unsigned log_n = 0;
    P2_TRY(sharded_views(g, oracles, n_oracles, &views, &log_n));
    P2_HIP(ctx, hipSetDevice(ctx->device));
    return eval_openings_core(ctx, views, log_n, points, n_points, out);
}

extern "C" int p2hot_group_fri_proof_sizes(const p2hot_sharded_batch *const *oracles, size_t n_oracles, const p2hot_fri_params *fp,
                                           p2hot_fri_proof_layout *out) {
    if (!oracles || n_oracles == 0 || !oracles[0]) return P2HOT_EINVAL;
    std::vector<size_t> widths;
    for (size_t o = 0; o < n_oracles; ++o) {
        if (!oracles[o]) return P2HOT_EINVAL;
        widths.push_back(oracles[o]->plan.W);
    }
    return fri_proof_layout(widths.data(), n_oracles, oracles[0]->plan.log_n, fp, out);
}

// PolynomialBatch::prove_openings + fri_proof over sharded oracles; `challenger` belongs to p2hot_group_ctx(group, 0).
extern "C" int p2hot_group_prove_openings(p2hot_group *g, const p2hot_fri_batch_info *batches, size_t n_batches,
                                          const p2hot_sharded_batch *const *oracles, size_t n_oracles, p2hot_challenger *challenger,
                                          const p2hot_fri_params *fp, p2hot_fri_proof *proof) {
    DeviceGuard restore_caller_device_(0);  // ranks' devices are visited below; the caller's comes back on return
    if (!g) return P2HOT_EINVAL;
    p2hot_ctx *ctx = g->ctx[0];
    std::vector<OracleView> views;
    unsigned log_n = 0;
    {
        P2_ENTER(ctx);
        P2_TRY(sharded_views(g, oracles, n_oracles, &views, &log_n));
        for (size_t o = 0; o < n_oracles && fp; ++o)
            if (oracles[o]->plan.rate_bits != fp->rate_bits || oracles[o]->plan.cap_height != fp->cap_height)
                P2_FAIL(ctx, P2HOT_EINVAL, "group_prove_openings: oracle %zu was committed with another rate / cap height", o);
        P2_HIP(ctx, hipSetDevice(ctx->device));
        size_t w_sum = 0;
        for (auto &v : views) w_sum += v.W;
        const unsigned layers0 = fp ? log_n + fp->rate_bits - fp->cap_height : 0;
        InitialOpener opener = [&](const u64 *idx, size_t Q, u64 *leaves_out, u64 *paths_out) -> int {
            size_t w_off = 0;
            for (size_t o = 0; o < n_oracles; ++o) {  // query-major layout: row q of oracle o at [q][w_off ..), path at [q][o][..]
                P2_TRY(sharded_open(const_cast<p2hot_sharded_batch *>(oracles[o]), idx, Q, leaves_out + w_off, w_sum,
                                    paths_out + o * 4 * layers0, n_oracles * 4 * (size_t)layers0));
                w_off += views[o].W;
            }
            return P2HOT_OK;
        };
        return prove_openings_core(ctx, batches, n_batches, views, log_n, challenger, fp, proof, &opener);
    }
}
