// host_prover.hpp -- the HOST-POINTER layer of libp2hot: what a patched plonky2 crate calls from the prover's main thread
// with ordinary Vec<F> buffers (include/p2hot.h, "prover session" section).  Included at the end of p2hot.hip (one TU).
//
// Reference call sites this layer stands behind:
//   PolynomialBatch::from_values / from_coeffs   plonky2/src/fri/oracle.rs:57-112      p2hot_commit, p2hot_commit_cols
//   MerkleTree::get / ::prove                      hash/merkle_tree.rs:227, :231-237     p2hot_batch_rows, p2hot_batch_paths
//   OpeningSet::new                                plonk/proof.rs:314-327                p2hot_eval_openings
//   PolynomialBatch::prove_openings + fri_proof    fri/oracle.rs:176-237, fri/prover.rs:24-82, :204-258   p2hot_prove_openings
//   all_wires_permutation_partial_products         plonk/prover.rs:356-449               p2hot_partial_products
//   quotient coset_ifft + chunks                   plonk/prover.rs:274-289, :810-815     p2hot_quotient_chunks
// Everything between the calls stays on the GPU behind opaque handles (p2hot_batch, p2hot_cols, p2hot_challenger).
#pragma once

#include <atomic>
#include <functional>
#include <thread>

// one call at a time per context: a second thread entering gets P2HOT_EBUSY instead of a data race on the scratch blocks
struct CallGuard {
    p2hot_ctx *ctx;
    bool ok;
    explicit CallGuard(p2hot_ctx *c) : ctx(c), ok(false) {
        if (ctx) ok = !ctx->busy.exchange(true, std::memory_order_acquire);
        if (ok) ctx->in_host_call = true;
    }
    ~CallGuard() {
        if (!ok) return;
        ctx->in_host_call = false;
        ctx->deferred.clear();  // a call that failed before its synchronisation delivers nothing
        ctx->pinned_used = 0;
        ctx->busy.store(false, std::memory_order_release);
    }
};
#define P2_ENTER(ctx)                                                                                                  \
    CallGuard guard_(ctx);                                                                                             \
    DeviceGuard dev_guard_(ctx);                                                                                       \
    if (!(ctx)) return P2HOT_EINVAL;                                                                                   \
    if (!guard_.ok) return P2HOT_EBUSY /* the context's error text belongs to the call that is running */

static int sync_checked(p2hot_ctx *ctx, int rc, const char *what) {
    hipError_t e = stream_sync(ctx);
    if (rc == P2HOT_OK && e != hipSuccess) P2_FAIL(ctx, P2HOT_EHIP, "%s: %s", what, hipGetErrorString(e));
    return rc;
}

// ------------------------------------------------------------------ device-resident column sets
extern "C" int p2hot_cols_upload(p2hot_ctx *ctx, const uint64_t *const *cols, size_t W, unsigned log_n, p2hot_cols **out) {
    P2_ENTER(ctx);
    if (!out) P2_FAIL(ctx, P2HOT_EINVAL, "cols_upload: null output");
    *out = nullptr;
    P2_TRY(check_log(ctx, log_n, "cols_upload"));
    if (W > 0 && !cols) P2_FAIL(ctx, P2HOT_EINVAL, "cols_upload: null column table");
    const size_t n = (size_t)1 << log_n;
    PoolBuf d(ctx);
    P2_TRY(pool_alloc(ctx, (W ? W : 1) * n * 8, &d.p));
    for (size_t c = 0; c < W; ++c)
        if (!cols[c]) P2_FAIL(ctx, P2HOT_EINVAL, "cols_upload: column %zu is null", c);
    P2_TRY(h2d_columns(ctx, d.u(), cols, W, n * 8, 0, W * n * 8, ctx->stream));
    P2_TRY(sync_checked(ctx, P2HOT_OK, "cols_upload"));
    *out = new p2hot_cols{ctx, d.u(), W, log_n, true};
    d.p = nullptr;
    return P2HOT_OK;
}

extern "C" int p2hot_cols_download(p2hot_cols *c, size_t first, size_t count, uint64_t *out) {
    if (!c) return P2HOT_EINVAL;
    p2hot_ctx *ctx = c->ctx;
    P2_ENTER(ctx);
    if (first > c->W || count > c->W - first) P2_FAIL(ctx, P2HOT_EINVAL, "cols_download: columns [%zu,+%zu) of %zu", first, count, c->W);
    if (count == 0) return P2HOT_OK;
    if (!out) P2_FAIL(ctx, P2HOT_EINVAL, "cols_download: null output");
    const size_t n = (size_t)1 << c->log_n;
    P2_HIP(ctx, hipMemcpyAsync(out, c->d + first * n, count * n * 8, hipMemcpyDeviceToHost, ctx->stream));
    return sync_checked(ctx, P2HOT_OK, "cols_download");
}

extern "C" size_t p2hot_cols_width(const p2hot_cols *c) { return c ? c->W : 0; }
extern "C" unsigned p2hot_cols_degree_log(const p2hot_cols *c) { return c ? c->log_n : 0; }

extern "C" void p2hot_cols_free(p2hot_cols *c) {
    if (!c) return;
    if (c->owned) {
        (void)hipStreamSynchronize(c->ctx->stream);
        pool_release(c->ctx, c->d);
    }
    delete c;
}

// ------------------------------------------------------------------ from_values / from_coeffs
static p2hot_batch *make_batch(p2hot_ctx *ctx, PoolBuf &lde, PoolBuf &dig, PoolBuf &coef, PoolBuf *vals, size_t W, unsigned log_n,
                               unsigned rate_bits, unsigned cap_height, size_t S = 0) {
    p2hot_batch *b = new p2hot_batch{ctx, lde.u(), W, ((size_t)1 << log_n) << rate_bits, dig.u(), log_n + rate_bits, cap_height};
    b->S = S;
    b->d_coef = coef.u();
    b->d_vals = vals ? vals->u() : nullptr;
    b->log_n = log_n;
    b->rate_bits = rate_bits;
    lde.p = dig.p = coef.p = nullptr;  // ownership moves to the handle (still live blocks of the context's cache)
    if (vals) vals->p = nullptr;
    return b;
}

// from_values / from_coeffs with host pointers; S salt columns (blinding = true: oracle.rs:123-137) or none
static int commit_host(p2hot_ctx *ctx, const uint64_t *const *cols, size_t W, unsigned log_n, unsigned rate_bits,
                       unsigned cap_height, int is_values, unsigned flags, const uint64_t *const *salt_cols, size_t S,
                       uint64_t *coeffs_out, uint64_t *leaves_out, uint64_t *digests_out, uint64_t *cap_out,
                       p2hot_batch **handle_out) {
    P2_ENTER(ctx);
    if (handle_out) *handle_out = nullptr;
    if (S && !salt_cols) P2_FAIL(ctx, P2HOT_EINVAL, "commit: null salt column table");
    for (size_t j = 0; j < S; ++j)
        if (!salt_cols[j]) P2_FAIL(ctx, P2HOT_EINVAL, "commit: salt column %zu is null", j);
    P2_TRY(check_log(ctx, log_n + rate_bits, "commit"));
    if (W == 0) P2_FAIL(ctx, P2HOT_EINVAL, "commit: no polynomials (the reference panics on polynomials[0], fri/oracle.rs:90)");
    if (!cols) P2_FAIL(ctx, P2HOT_EINVAL, "commit: null column table");
    if (flags & ~(unsigned)(P2HOT_KEEP_VALUES | P2HOT_COEFFS_PER_COLUMN | P2HOT_LEAVES_ASYNC | P2HOT_LEAVES_NATURAL))
        P2_FAIL(ctx, P2HOT_EINVAL, "commit: unknown flags %#x", flags);
    if ((flags & P2HOT_LEAVES_ASYNC) && !(leaves_out && handle_out))
        P2_FAIL(ctx, P2HOT_EINVAL, "commit: P2HOT_LEAVES_ASYNC needs leaves_out and handle_out (the handle owns the copy in flight)");
    if ((flags & P2HOT_LEAVES_NATURAL) && !leaves_out) P2_FAIL(ctx, P2HOT_EINVAL, "commit: P2HOT_LEAVES_NATURAL without leaves_out");
    const bool async_leaves = (flags & P2HOT_LEAVES_ASYNC) != 0;
    // P2HOT_COEFFS_PER_COLUMN: coeffs_out is a table of W host pointers (one Vec per polynomial on the caller's side)
    uint64_t *const *coeffs_cols = (flags & P2HOT_COEFFS_PER_COLUMN) ? reinterpret_cast<uint64_t *const *>(coeffs_out) : nullptr;
    if (coeffs_cols)
        for (size_t c = 0; c < W; ++c)
            if (!coeffs_cols[c]) P2_FAIL(ctx, P2HOT_EINVAL, "commit: coefficient destination %zu is null", c);
    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    const unsigned log_N = log_n + rate_bits;
    if (cap_height > log_N) P2_FAIL(ctx, P2HOT_EINVAL, "commit: cap_height %u > log2(N) %u (merkle_tree.rs:195-200)", cap_height, log_N);
    for (size_t c = 0; c < W; ++c)
        if (!cols[c]) P2_FAIL(ctx, P2HOT_EINVAL, "commit: column %zu is null", c);
    const size_t nd = p2hot_num_digests(log_N, cap_height), cap_words = (size_t)4 << cap_height;
    const bool keep_vals = is_values && handle_out && (flags & P2HOT_KEEP_VALUES);
    PoolBuf d_work(ctx), d_vals(ctx), d_lde(ctx), d_leaves(ctx), d_dig(ctx), d_cap(ctx);
    const size_t LW = W + S;  // leaf width
    const size_t Wn = (W ? W : 1) * n * 8, WN = (LW ? LW : 1) * N * 8;
    PoolBuf d_salt(ctx);
    P2_TRY(pool_alloc(ctx, Wn, &d_work.p));  // uploaded columns; becomes the coefficients in place
    if (keep_vals) P2_TRY(pool_alloc(ctx, Wn, &d_vals.p));
    P2_TRY(pool_alloc(ctx, WN, &d_lde.p));
    if (leaves_out) P2_TRY(pool_alloc(ctx, WN, &d_leaves.p));
    if (S) P2_TRY(pool_alloc(ctx, S * N * 8, &d_salt.p));
    P2_TRY(pool_alloc(ctx, (nd ? nd : 1) * 32, &d_dig.p));
    P2_TRY(pool_alloc(ctx, cap_words * 8, &d_cap.p));
    // The PCIe copies run on the side stream beside the compute stream.  Column blocks are uploaded, transformed (iNTT)
    // and extended (LDE) one after the other -- the upload of block b+1 overlaps the transforms of block b -- and the
    // coefficient blocks go back to the host while the leaf sponge runs.  Small batches are one block.
    const size_t kBlockCols = ctx->host_block_cols ? ctx->host_block_cols : (W >= 32 && W * n >= ((size_t)1 << 22)) ? 16 : (W ? W : 1);
    // block b = columns [blk0[b], blk0[b + 1]).  The first upload has nothing to hide behind, so when there are several blocks
    // the first one is half as wide (a multiple of the sponge rate where it can be)
    std::vector<size_t> blk0;
    {
        size_t c = 0;
        const size_t first = (!ctx->host_block_cols && kBlockCols >= 16 && W > kBlockCols) ? kBlockCols / 2 : kBlockCols;
        while (c < W) {
            blk0.push_back(c);
            c += (c == 0 ? first : kBlockCols);
        }
        blk0.push_back(W);
    }
    const size_t nb = W ? blk0.size() - 1 : 0;
    std::vector<hipEvent_t> up, done;
    hipStream_t copy_stream = ctx->stream;
    // a large leaf matrix going back to the host gets the copy stream even when the columns are one block (W < 32: the Zs and
    // quotient commitments): its copy then runs beside the sponge as well
    const bool big_leaves = ctx->host_leaves_first && leaves_out && LW && nb >= 1 && LW * N * 8 >= ((size_t)64 << 20);
    const bool two_streams = nb > 1 || big_leaves;
    if (two_streams) copy_stream = ctx->side;
    // With more than one block the leaf sponge does not wait for the last column: after each block's LDE it absorbs the
    // 8-column chunks that are complete (its state parked in a scratch block between launches), so the hashing -- three
    // quarters of the commit -- runs beside the uploads still in flight instead of behind them.
    // (not when the row-major leaf matrix goes back as well: that copy -- 9 GB at C3 -- is the long pole and can start as soon
    // as every column is extended, so the transforms run first and the whole sponge runs beside the copy instead)
    // P2HOT_LEAVES_ASYNC with several column blocks: THREE compute / copy lanes.  The transforms (upload-bound: 25 ms for 10 ms of
    // kernels at the C3 wires shape) and the transposition run on a stream of their own, so the leaf matrix exists -- and starts
    // leaving on the leaf stream -- as early as in the leaves-first order; the leaf sponge absorbs each block's columns on the
    // context's stream BESIDE them (the chunked order: the hashing, three quarters of the call, hides the uploads), so the call
    // returns as early as a call without leaves.  Neither single-stream order gives both (profiles/r05_async_ab.txt).
    const bool xsplit = async_leaves && ctx->host_chunked_hash && ctx->host_async_split && nb > 1 && LW > 8 && LW <= 0xFFFFFFFFull;
    const bool leaves_first = ctx->host_leaves_first && leaves_out && LW && (nb > 1 || big_leaves) && !xsplit;
    const bool early_transpose = async_leaves && LW;  // the asynchronous copy: the matrix is transposed as soon as the last column is extended
    const bool chunked = ctx->host_chunked_hash && nb > 1 && LW > 8 && LW <= 0xFFFFFFFFull && !leaves_first;
    PoolBuf d_state(ctx);
    ForestGeom geom{};
    unsigned hashed = 0;  // columns the sponge has absorbed (a multiple of 8 until the end)
    if (chunked) {
        P2_TRY(pool_alloc(ctx, (size_t)12 * N * 8, &d_state.p));
        P2_TRY(forest_geom(ctx, log_N, cap_height, 0, N, d_dig.u(), d_cap.u(), &geom));
    }
    // When the digests go back to the host the tail -- the last chunk(s) of the sponge and the tree levels -- runs per group of
    // cap subtrees: a group's slice of the digest array is contiguous in the reference layout, so it can travel while the next
    // groups are hashed.  (A copy into pageable memory holds the calling thread until it is done: every group's kernels are
    // enqueued first, the copies are issued after them.)
    size_t tail_groups = 1;
    if (chunked && digests_out && nd)
        while (tail_groups < 8 && tail_groups * 2 <= ((size_t)1 << cap_height) && (N / (tail_groups * 2)) >= ctx->host_tail_min_leaves)
            tail_groups *= 2;
    std::vector<hipEvent_t> tail_ev;
    [[maybe_unused]] hipEvent_t leaves_ev = nullptr;
    const unsigned leaf_rev = (flags & P2HOT_LEAVES_NATURAL) ? log_N : 0u;  // (log_N == 0: one row, nothing to reverse)
    p2hot_batch::LeafCopy *leafcopy = nullptr;  // P2HOT_LEAVES_ASYNC: handed to the batch handle at the end
    const hipStream_t main_stream = ctx->stream;
    struct StreamRestore {  // the transform lane borrows `ctx->stream` (every transform launches "on the context's stream"): given back on every path
        p2hot_ctx *c;
        hipStream_t s;
        ~StreamRestore() { c->stream = s; }
    } stream_restore{ctx, main_stream};
    std::vector<hipEvent_t> lde_ev;  // xsplit: block b is extended (transform lane -> sponge lane)
    auto absorb_upto = [&](size_t cols_done, bool may_finish) -> int {  // cols_done leaf columns of d_lde are final
        const unsigned end = (cols_done >= LW && may_finish) ? (unsigned)LW : (unsigned)((cols_done < LW ? cols_done : LW - 1) / 8 * 8);
        if (end <= hashed) return P2HOT_OK;
        P2_TRY(hash_leaves_chunks(ctx, main_stream, merkle::ColMajorReader{d_lde.u(), N}, LW, geom, N, hashed, end, d_state.u(), N));
        hashed = end;
        return P2HOT_OK;
    };
    auto issue_leaf_blocks = [&](size_t k0, size_t k1) -> int {  // blocks [k0, k1) of the asynchronous leaf copy, one event each
        const size_t bw = leafcopy->rows_per_block * LW;           // words per block
        for (size_t k = k0; k < k1; ++k) {
            P2_HIP(ctx, hipMemcpyAsync(leaves_out + k * bw, d_leaves.u() + k * bw, bw * 8, hipMemcpyDeviceToHost, ctx->leaf_stream));
            hipEvent_t e;
            P2_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));
            leafcopy->ev.push_back(e);
            P2_HIP(ctx, hipEventRecord(e, ctx->leaf_stream));
        }
        return P2HOT_OK;
    };
    auto body = [&]() -> int {
        if (two_streams) {
            for (size_t b = 0; b < 2 * nb; ++b) {
                hipEvent_t e;
                P2_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                (b < nb ? up : done).push_back(e);
            }
            // the side stream must not start before work already queued on the compute stream has finished with the blocks
            P2_HIP(ctx, hipEventRecord(ctx->join_event, ctx->stream));
            P2_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->join_event, 0));
        }
        if (xsplit) {  // the transform lane: behind whatever the context's stream already holds, then on its own
            if (!ctx->xform_stream) P2_HIP(ctx, hipStreamCreateWithFlags(&ctx->xform_stream, hipStreamNonBlocking));
            P2_HIP(ctx, hipStreamWaitEvent(ctx->xform_stream, ctx->join_event, 0));
            for (size_t b = 0; b < nb; ++b) {
                hipEvent_t e;
                P2_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                lde_ev.push_back(e);
            }
            ctx->stream = ctx->xform_stream;  // (until the transposition below; StreamRestore covers the error paths)
        }
        for (size_t b = 0; b < nb; ++b) {
            const size_t c0 = blk0[b], cnt = blk0[b + 1] - c0;
            u64 *blk = d_work.u() + c0 * n;
            u64 *land = keep_vals ? d_vals.u() + c0 * n : blk;  // where the upload lands
            P2_TRY(h2d_columns(ctx, land, cols + c0, cnt, n * 8, c0 * n * 8, W * n * 8 + S * N * 8, copy_stream));
            if (two_streams) {
                P2_HIP(ctx, hipEventRecord(up[b], ctx->side));
                P2_HIP(ctx, hipStreamWaitEvent(ctx->stream, up[b], 0));
            }
            if (keep_vals) P2_HIP(ctx, hipMemcpyAsync(blk, land, cnt * n * 8, hipMemcpyDeviceToDevice, ctx->stream));
            if (is_values) {  // "IFFT" (oracle.rs:65-69): the block becomes its coefficients in place
                P2_TRY(ntt_natural(ctx, blk, cnt, n, log_n, true));
            } else if (coeffs_out) {
                P2HOT_LAUNCH(ntt::canon_kernel, dim3(cdiv(cnt * n, 256)), dim3(256), 0, ctx->stream, blk, cnt * n);
                P2_LAUNCH_CHECK(ctx);
            }
            if (two_streams) P2_HIP(ctx, hipEventRecord(done[b], ctx->stream));
            P2_TRY(p2hot_coset_lde_dev(ctx, blk, cnt, n, log_n, rate_bits, gl::COSET_SHIFT, 0, N, d_lde.u() + c0 * N, N));
            if (xsplit) {  // the sponge lane picks the block up when it is extended
                P2_HIP(ctx, hipEventRecord(lde_ev[b], ctx->stream));
                P2_HIP(ctx, hipStreamWaitEvent(main_stream, lde_ev[b], 0));
            }
            if (chunked) P2_TRY(absorb_upto(c0 + cnt, tail_groups == 1 && !S && !xsplit));  // (a grouped tail, and the split lanes, keep the last chunk)
        }
        if (S) {
            // the salt vectors are LDE-value columns in natural order (oracle.rs:133-137): like the LDE values they reach the
            // leaves through transpose + reverse_index_bits (:97-98), i.e. column W + j of the committed matrix is salt_j[bitrev(r)]
            // (the pinned staging block is shared with the column uploads still in flight: the salts take the slots behind them)
            P2_TRY(h2d_columns(ctx, d_salt.u(), salt_cols, S, N * 8, W * n * 8, W * n * 8 + S * N * 8, ctx->stream));
            P2_TRY(launch_bitrev(ctx, d_salt.u(), d_lde.u() + W * N, S, N, N, log_N));
        }
        if (leaves_first || early_transpose) {  // the row-major matrix before the sponge('s tail): its copy starts while the leaves are hashed
            P2_TRY(transpose_rows(ctx, d_lde.u(), N, LW, N, d_leaves.u(), leaf_rev));
            if (two_streams || async_leaves) {
                P2_HIP(ctx, hipEventCreateWithFlags(&leaves_ev, hipEventDisableTiming));
                P2_HIP(ctx, hipEventRecord(leaves_ev, ctx->stream));
            }
        }
        if (early_transpose) {
            // the matrix leaves in row blocks on a stream of its own, each block fenced by an event; this call does not wait for it.
            // The first quarter of the blocks goes now, beside the leaf sponge; the rest is queued BEHIND the digests' copy (below):
            // the 0.5 GB of digests are what this call still waits for, and they would share the link with 9 GB of leaves otherwise
            if (!ctx->leaf_stream) P2_HIP(ctx, hipStreamCreateWithFlags(&ctx->leaf_stream, hipStreamNonBlocking));
            P2_HIP(ctx, hipStreamWaitEvent(ctx->leaf_stream, leaves_ev, 0));
            leafcopy = new p2hot_batch::LeafCopy();
            leafcopy->ev.reserve(66);
            leafcopy->rows = N;
            leafcopy->aux = leaves_ev;  // the leaf stream's wait on it may not have run yet: destroyed with the copy, not with this call
            leaves_ev = nullptr;
            size_t blocks = 64;
            while (blocks > 1 && N / blocks < 1024) blocks >>= 1;  // (rows per block stays a power of two: N is one)
            leafcopy->rows_per_block = N / blocks;
            // how many blocks leave BEFORE the digests' copy is queued.  Measured at the C3 wires shape (profiles/r05_async_ab.txt): beside
            // the chunked sponge every early block costs the call ~7 ms (the copy and the sponge share the chip) and buys the last row
            // nothing -> none; in the single-stream leaves-first order a quarter of them moves the last row from 251 to 220 ms for 10 ms
            size_t early = ctx->host_async_early_blocks != (size_t)-1 ? ctx->host_async_early_blocks : (xsplit ? 0 : blocks / 4);
            if (early > blocks || !(digests_out && nd)) early = blocks;
            P2_TRY(issue_leaf_blocks(0, early));
        }
        if (xsplit) {  // the transform lane is done: the salts and the transposition were its last work; the sponge's tail waits for them
            P2_HIP(ctx, hipEventRecord(ctx->join_event, ctx->stream));
            ctx->stream = main_stream;
            P2_HIP(ctx, hipStreamWaitEvent(main_stream, ctx->join_event, 0));
        }
        if (chunked && tail_groups > 1) {
            const size_t cnt = N / tail_groups;
            for (size_t g = 0; g < tail_groups; ++g) {
                ForestGeom gg;
                P2_TRY(forest_geom(ctx, log_N, cap_height, g * cnt, cnt, d_dig.u(), d_cap.u(), &gg));
                P2_TRY(hash_leaves_chunks(ctx, ctx->stream, merkle::ColMajorReader{d_lde.u() + g * cnt, N}, LW, gg, cnt, hashed,
                                          (unsigned)LW, d_state.u() + g * cnt, N));
                P2_TRY(merkle_levels(ctx, gg, cnt));
                if (two_streams) {
                    hipEvent_t e;
                    P2_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    tail_ev.push_back(e);
                    P2_HIP(ctx, hipEventRecord(e, ctx->stream));
                }
            }
            hashed = (unsigned)LW;
        } else if (chunked) {
            P2_TRY(absorb_upto(LW, true));  // what is left (the salts' chunks)
            P2_TRY(merkle_levels(ctx, geom, N));
        } else {
            P2_TRY(p2hot_merkle_dev(ctx, d_lde.u(), 0, N, LW, log_N, cap_height, 0, N, d_dig.u(), d_cap.u()));
        }
        if (leaves_out && LW && !leaves_first && !early_transpose) P2_TRY(transpose_rows(ctx, d_lde.u(), N, LW, N, d_leaves.u(), leaf_rev));
        // coefficient blocks go back while the leaf sponge runs: queued behind the uploads on the copy stream, each
        // waiting for its block's transform only
        if (coeffs_out)
            for (size_t b = 0; b < nb; ++b) {
                const size_t c0 = blk0[b], cnt = blk0[b + 1] - c0;
                if (two_streams) P2_HIP(ctx, hipStreamWaitEvent(ctx->side, done[b], 0));
                if (coeffs_cols) {
                    for (size_t c = c0; c < c0 + cnt; ++c)
                        P2_HIP(ctx, hipMemcpyAsync(coeffs_cols[c], d_work.u() + c * n, n * 8, hipMemcpyDeviceToHost, copy_stream));
                } else {
                    P2_HIP(ctx, hipMemcpyAsync(coeffs_out + c0 * n, d_work.u() + c0 * n, cnt * n * 8, hipMemcpyDeviceToHost, copy_stream));
                }
            }
        if (leaves_out && LW && async_leaves) {
            // (issued above, right after the transposition; the later blocks follow the digests below)
        } else if (leaves_out && LW) {
            hipStream_t ls = ctx->stream;
            if (leaves_ev) {
                P2_HIP(ctx, hipStreamWaitEvent(copy_stream, leaves_ev, 0));
                ls = copy_stream;
            }
            P2_HIP(ctx, hipMemcpyAsync(leaves_out, d_leaves.p, LW * N * 8, hipMemcpyDeviceToHost, ls));
        }
        if (digests_out && nd && tail_groups > 1) {  // the groups' digest slices, each behind its group's levels only
            const size_t cnt = N / tail_groups, sub_leaves = N >> cap_height, sub_words = 8 * (sub_leaves - 1);
            for (size_t g = 0; g < tail_groups; ++g) {
                const size_t w0 = (g * cnt / sub_leaves) * sub_words, nw = (cnt / sub_leaves) * sub_words;
                if (two_streams) P2_HIP(ctx, hipStreamWaitEvent(copy_stream, tail_ev[g], 0));
                P2_HIP(ctx, hipMemcpyAsync(digests_out + w0, d_dig.u() + w0, nw * 8, hipMemcpyDeviceToHost, copy_stream));
            }
        } else if (digests_out && nd) {
            P2_HIP(ctx, hipMemcpyAsync(digests_out, d_dig.p, nd * 32, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (leafcopy && leafcopy->ev.size() < N / leafcopy->rows_per_block) {  // the rest of the leaf blocks, behind the digests' copy
            hipEvent_t e;
            P2_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            leafcopy->aux2 = e;
            P2_HIP(ctx, hipEventRecord(e, tail_groups > 1 ? copy_stream : ctx->stream));
            P2_HIP(ctx, hipStreamWaitEvent(ctx->leaf_stream, e, 0));
            P2_TRY(issue_leaf_blocks(leafcopy->ev.size(), N / leafcopy->rows_per_block));
        }
        if (cap_out) P2_TRY(d2h(ctx, cap_out, d_cap.p, cap_words * 8));
        return P2HOT_OK;
    };
    int rc = body();
    ctx->stream = main_stream;
    hipError_t e1 = hipSuccess;
    if (two_streams) e1 = hipStreamSynchronize(ctx->side);
    if (xsplit && ctx->xform_stream) {
        hipError_t e2 = hipStreamSynchronize(ctx->xform_stream);
        if (e1 == hipSuccess) e1 = e2;
    }
    for (hipEvent_t ev : lde_ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : up) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : done) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : tail_ev) (void)hipEventDestroy(ev);
    if (leaves_ev) (void)hipEventDestroy(leaves_ev);
    rc = sync_checked(ctx, rc, "commit");
    if (rc == P2HOT_OK && e1 != hipSuccess) P2_FAIL(ctx, P2HOT_EHIP, "commit: %s", hipGetErrorString(e1));
    if (rc == P2HOT_OK && handle_out) *handle_out = make_batch(ctx, d_lde, d_dig, d_work, keep_vals ? &d_vals : nullptr, W, log_n, rate_bits, cap_height, S);
    if (leafcopy) {
        if (rc == P2HOT_OK && handle_out && *handle_out) {
            leafcopy->d_staging = d_leaves.p;  // stays alive behind the handle until the last block has landed
            d_leaves.p = nullptr;
            (*handle_out)->leafcopy = leafcopy;
        } else {  // a failed call leaves nothing in flight
            (void)hipStreamSynchronize(ctx->leaf_stream);
            for (hipEvent_t ev : leafcopy->ev) (void)hipEventDestroy(ev);
            if (leafcopy->aux) (void)hipEventDestroy(leafcopy->aux);
            if (leafcopy->aux2) (void)hipEventDestroy(leafcopy->aux2);
            delete leafcopy;
        }
    }
    return rc;
}

static void leafcopy_finish(p2hot_batch *b) {  // waits for the copy in flight and returns its staging block
    if (!b || !b->leafcopy) return;
    // this batch's copy only: the leaf stream is the context's, and a later batch's 9 GB may be queued behind this one's.  The
    // block events are recorded in stream order, so the last one covers every block and both auxiliary waits
    const p2hot_batch::LeafCopy *lc = b->leafcopy;
    if (!lc->ev.empty() && lc->rows_per_block && lc->ev.size() == lc->rows / lc->rows_per_block)
        (void)hipEventSynchronize(lc->ev.back());
    else if (b->ctx->leaf_stream)
        (void)hipStreamSynchronize(b->ctx->leaf_stream);  // (a copy that was never fully issued: nothing else to wait on)
    for (hipEvent_t ev : b->leafcopy->ev) (void)hipEventDestroy(ev);
    if (b->leafcopy->aux) (void)hipEventDestroy(b->leafcopy->aux);
    if (b->leafcopy->aux2) (void)hipEventDestroy(b->leafcopy->aux2);
    pool_release(b->ctx, b->leafcopy->d_staging);
    delete b->leafcopy;
    b->leafcopy = nullptr;
}

// LOCK-FREE (rayon workers call it while another thread is inside a locked p2hot_* call of the same context): it therefore
// never writes the context's error text -- the code is all a failure reports
extern "C" int p2hot_batch_leaves_wait(p2hot_batch *b, size_t row_lo, size_t row_hi) {
    if (!b) return P2HOT_EINVAL;
    if (row_lo > row_hi || row_hi > b->N) return P2HOT_EINVAL;
    if (!b->leafcopy || row_lo == row_hi) return P2HOT_OK;
    DeviceGuard dev_guard_(b->ctx);
    const size_t rpb = b->leafcopy->rows_per_block;
    for (size_t k = row_lo / rpb; k <= (row_hi - 1) / rpb; ++k)
        if (hipEventSynchronize(b->leafcopy->ev[k]) != hipSuccess) return P2HOT_EHIP;
    return P2HOT_OK;
}
// rows per block of the pending asynchronous leaf copy (min(64, N / 1024) blocks, at least one), 0 when nothing is in flight:
// once row r has landed, so has every row below (r / rows_per_block + 1) * rows_per_block -- what the shim rounds its mark up to
extern "C" size_t p2hot_batch_leaves_block_rows(const p2hot_batch *b) { return b && b->leafcopy ? b->leafcopy->rows_per_block : 0; }

extern "C" int p2hot_commit(p2hot_ctx *ctx, const uint64_t *const *cols, size_t W, unsigned log_n, unsigned rate_bits,
                            unsigned cap_height, int is_values, unsigned flags, uint64_t *coeffs_out, uint64_t *leaves_out,
                            uint64_t *digests_out, uint64_t *cap_out, p2hot_batch **handle_out) {
    return commit_host(ctx, cols, W, log_n, rate_bits, cap_height, is_values, flags, nullptr, 0, coeffs_out, leaves_out, digests_out,
                       cap_out, handle_out);
}

extern "C" int p2hot_commit_salted(p2hot_ctx *ctx, const uint64_t *const *cols, size_t W, unsigned log_n, unsigned rate_bits,
                                   unsigned cap_height, int is_values, unsigned flags, const uint64_t *const *salt_cols,
                                   size_t n_salt, uint64_t *coeffs_out, uint64_t *leaves_out, uint64_t *digests_out,
                                   uint64_t *cap_out, p2hot_batch **handle_out) {
    return commit_host(ctx, cols, W, log_n, rate_bits, cap_height, is_values, flags, salt_cols, n_salt, coeffs_out, leaves_out,
                       digests_out, cap_out, handle_out);
}

// from_values / from_coeffs on a device-resident column set.  CONSUMES `cols` -- its block becomes the batch's coefficients
// (from_coeffs) or its kept values (from_values with P2HOT_KEEP_VALUES), or is released -- on success and on every failure
// EXCEPT P2HOT_EBUSY (another call is running on the context) and P2HOT_EINVAL (a null set, a set of another context, a borrowed
// view, an empty set, a bad rate / cap height / flag word): every argument is validated before the set is touched, so those two
// codes always mean "not consumed, the caller still owns the handle".
extern "C" int p2hot_commit_cols(p2hot_ctx *ctx, p2hot_cols *cols, unsigned rate_bits, unsigned cap_height, int is_values,
                                 unsigned flags, uint64_t *coeffs_out, uint64_t *leaves_out, uint64_t *digests_out,
                                 uint64_t *cap_out, p2hot_batch **handle_out) {
    P2_ENTER(ctx);
    if (handle_out) *handle_out = nullptr;
    if (!cols) P2_FAIL(ctx, P2HOT_EINVAL, "commit_cols: null column set");
    // every argument is checked BEFORE the set is touched: any P2HOT_EINVAL (and P2HOT_EBUSY) means "not consumed"
    if (cols->ctx != ctx || !cols->owned) P2_FAIL(ctx, P2HOT_EINVAL, "commit_cols: the column set belongs to another context or is a borrowed view");
    const size_t W = cols->W;
    const unsigned log_n = cols->log_n;
    P2_TRY(check_log(ctx, log_n + rate_bits, "commit_cols"));
    if (W == 0) P2_FAIL(ctx, P2HOT_EINVAL, "commit_cols: no polynomials (the reference panics on polynomials[0], fri/oracle.rs:90)");
    if (flags & ~(unsigned)(P2HOT_KEEP_VALUES | P2HOT_COEFFS_PER_COLUMN)) P2_FAIL(ctx, P2HOT_EINVAL, "commit_cols: unknown flags %#x", flags);
    uint64_t *const *coeffs_cols = (flags & P2HOT_COEFFS_PER_COLUMN) ? reinterpret_cast<uint64_t *const *>(coeffs_out) : nullptr;
    if (coeffs_cols)
        for (size_t c = 0; c < W; ++c)
            if (!coeffs_cols[c]) P2_FAIL(ctx, P2HOT_EINVAL, "commit_cols: coefficient destination %zu is null", c);
    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    const unsigned log_N = log_n + rate_bits;
    if (cap_height > log_N) P2_FAIL(ctx, P2HOT_EINVAL, "commit_cols: cap_height %u > log2(N) %u (merkle_tree.rs:195-200)", cap_height, log_N);
    // everything that can be refused for its ARGUMENTS is refused before the consume point, so that "EINVAL / EBUSY = not consumed"
    // (include/p2hot.h) holds for nested checks too: the transforms launch one grid row per polynomial (65535 at most)
    if (W > 65535) P2_FAIL(ctx, P2HOT_EINVAL, "commit_cols: %zu polynomials exceed the launch grid (65535 per call)", W);
    PoolBuf d_in(ctx);
    d_in.p = cols->d;
    delete cols;  // consumed from here on: the block is ours whatever happens below
    const size_t nd = p2hot_num_digests(log_N, cap_height), cap_words = (size_t)4 << cap_height;
    const bool keep_vals = is_values && handle_out && (flags & P2HOT_KEEP_VALUES);
    PoolBuf d_coef(ctx), d_lde(ctx), d_leaves(ctx), d_dig(ctx), d_cap(ctx);
    const size_t Wn = (W ? W : 1) * n * 8, WN = (W ? W : 1) * N * 8;
    if (keep_vals) P2_TRY(pool_alloc(ctx, Wn, &d_coef.p));
    P2_TRY(pool_alloc(ctx, WN, &d_lde.p));
    if (leaves_out) P2_TRY(pool_alloc(ctx, WN, &d_leaves.p));
    P2_TRY(pool_alloc(ctx, (nd ? nd : 1) * 32, &d_dig.p));
    P2_TRY(pool_alloc(ctx, cap_words * 8, &d_cap.p));
    auto body = [&]() -> int {
        u64 *co = d_in.u();
        if (keep_vals) {
            P2_HIP(ctx, hipMemcpyAsync(d_coef.p, d_in.p, W * n * 8, hipMemcpyDeviceToDevice, ctx->stream));
            co = d_coef.u();
        }
        if (is_values) {
            P2_TRY(ntt_natural(ctx, co, W, n, log_n, true));
        } else if (W) {
            P2HOT_LAUNCH(ntt::canon_kernel, dim3(cdiv(W * n, 256)), dim3(256), 0, ctx->stream, co, W * n);
            P2_LAUNCH_CHECK(ctx);
        }
        P2_TRY(p2hot_commit_dev(ctx, co, n, W, log_n, rate_bits, cap_height, 0, 0, N, nullptr, 0, d_lde.u(), N,
                                leaves_out ? d_leaves.u() : nullptr, d_dig.u(), d_cap.u()));
        if (coeffs_cols) {
            for (size_t c = 0; c < W; ++c) P2_HIP(ctx, hipMemcpyAsync(coeffs_cols[c], co + c * n, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        } else if (coeffs_out && W) {
            P2_HIP(ctx, hipMemcpyAsync(coeffs_out, co, W * n * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (leaves_out && W) P2_HIP(ctx, hipMemcpyAsync(leaves_out, d_leaves.p, W * N * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (digests_out && nd) P2_HIP(ctx, hipMemcpyAsync(digests_out, d_dig.p, nd * 32, hipMemcpyDeviceToHost, ctx->stream));
        if (cap_out) P2_TRY(d2h(ctx, cap_out, d_cap.p, cap_words * 8));
        return P2HOT_OK;
    };
    int rc = sync_checked(ctx, body(), "commit_cols");
    if (rc == P2HOT_OK && handle_out) {
        if (keep_vals)
            *handle_out = make_batch(ctx, d_lde, d_dig, d_coef, &d_in, W, log_n, rate_bits, cap_height);
        else
            *handle_out = make_batch(ctx, d_lde, d_dig, d_in, nullptr, W, log_n, rate_bits, cap_height);
    }
    return rc;
}

// ------------------------------------------------------------------ M commitments of one shape in one set of launches
// Recursion-size proofs (2^12 rows) are latency-bound: a tree level is one ~12 us permutation chain whatever its width.  M
// proofs of the same shape therefore share every launch: their columns are interleaved as [W][M][n] (column e of all proofs,
// then column e+1, ...), so the iNTT and the coset LDE see W*M polynomials, and the M leaf matrices form ONE forest of
// M * 2^cap_height subtrees (leaf L = m*N + r of width W, column stride M*N): the digest array is the M trees' arrays back to back,
// the cap array the M caps -- exactly what M separate p2hot_commit_dev calls produce (fri/oracle.rs:57-112 per proof).
extern "C" int p2hot_commit_many_dev(p2hot_ctx *ctx, uint64_t *d_cols, size_t M, size_t W, unsigned log_n, unsigned rate_bits,
                                     unsigned cap_height, int is_values, uint64_t *d_lde, uint64_t *d_digests, uint64_t *d_cap) {
    if (!ctx) return P2HOT_EINVAL;
    DeviceGuard dev_guard_(ctx);
    P2_TRY(check_log(ctx, log_n + rate_bits, "commit_many"));
    if (M == 0) return P2HOT_OK;
    if (W == 0) P2_FAIL(ctx, P2HOT_EINVAL, "commit_many: no polynomials (the reference panics on polynomials[0], fri/oracle.rs:90)");
    if (M & (M - 1)) P2_FAIL(ctx, P2HOT_EINVAL, "commit_many: the number of proofs (%zu) must be a power of two", M);
    unsigned lm = 0;
    while (((size_t)1 << lm) < M) ++lm;
    const unsigned log_N = log_n + rate_bits;
    if (cap_height > log_N) P2_FAIL(ctx, P2HOT_EINVAL, "commit_many: cap_height %u > log2(N) %u (merkle_tree.rs:195-200)", cap_height, log_N);
    if (!d_cols || !d_lde || !d_cap || (p2hot_num_digests(log_N, cap_height) && !d_digests)) P2_FAIL(ctx, P2HOT_EINVAL, "commit_many: null buffer");
    if (W * M > 65535) P2_FAIL(ctx, P2HOT_EINVAL, "commit_many: W * M = %zu polynomials exceed one launch (65535)", W * M);
    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    if (is_values) {
        P2_TRY(ntt_natural(ctx, d_cols, W * M, n, log_n, true));  // "IFFT": in place, the block becomes the coefficients
    } else {
        P2HOT_LAUNCH(ntt::canon_kernel, dim3(cdiv(W * M * n, 256)), dim3(256), 0, ctx->stream, d_cols, W * M * n);
        P2_LAUNCH_CHECK(ctx);
    }
    P2_TRY(p2hot_coset_lde_dev(ctx, d_cols, W * M, n, log_n, rate_bits, gl::COSET_SHIFT, 0, N, d_lde, N));
    // one forest: M * 2^cap_height subtrees of 2^(log_N - cap_height) leaves
    return p2hot_merkle_dev(ctx, d_lde, 0, M * N, W, log_N + lm, cap_height + lm, 0, M * N, d_digests, d_cap);
}

// the host-pointer form: cols[m * W + e] = column e of proof m (n words each).  coeffs_out [M][W][n], digests_out [M][nd][4],
// caps_out [M][2^cap_height][4] (any may be NULL); handles_out [M] (optional): one p2hot_batch per proof, for p2hot_batch_rows /
// _paths / _coeffs, p2hot_eval_openings and p2hot_prove_openings -- they share the device blocks, freed with the last handle.
extern "C" int p2hot_commit_many(p2hot_ctx *ctx, const uint64_t *const *cols, size_t M, size_t W, unsigned log_n, unsigned rate_bits,
                                 unsigned cap_height, int is_values, uint64_t *coeffs_out, uint64_t *digests_out, uint64_t *caps_out,
                                 p2hot_batch **handles_out) {
    P2_ENTER(ctx);
    if (handles_out)
        for (size_t m = 0; m < M; ++m) handles_out[m] = nullptr;
    P2_TRY(check_log(ctx, log_n + rate_bits, "commit_many"));
    if (M == 0) return P2HOT_OK;
    if (W == 0) P2_FAIL(ctx, P2HOT_EINVAL, "commit_many: no polynomials (the reference panics on polynomials[0], fri/oracle.rs:90)");
    if (M & (M - 1)) P2_FAIL(ctx, P2HOT_EINVAL, "commit_many: the number of proofs (%zu) must be a power of two", M);
    if (!cols) P2_FAIL(ctx, P2HOT_EINVAL, "commit_many: null column table");
    for (size_t i = 0; i < M * W; ++i)
        if (!cols[i]) P2_FAIL(ctx, P2HOT_EINVAL, "commit_many: column %zu of proof %zu is null", i % W, i / W);
    const unsigned log_N = log_n + rate_bits;
    if (cap_height > log_N) P2_FAIL(ctx, P2HOT_EINVAL, "commit_many: cap_height %u > log2(N) %u (merkle_tree.rs:195-200)", cap_height, log_N);
    const size_t n = (size_t)1 << log_n, N = n << rate_bits, nd = p2hot_num_digests(log_N, cap_height), cap_words = (size_t)4 << cap_height;
    PoolBuf d_co(ctx), d_lde(ctx), d_dig(ctx), d_cap(ctx);
    P2_TRY(pool_alloc(ctx, W * M * n * 8, &d_co.p));
    P2_TRY(pool_alloc(ctx, W * M * N * 8, &d_lde.p));
    P2_TRY(pool_alloc(ctx, (nd ? nd : 1) * M * 32, &d_dig.p));
    P2_TRY(pool_alloc(ctx, cap_words * M * 8, &d_cap.p));
    auto body = [&]() -> int {
        // interleave on the way in: column e of proof m lands in slot e * M + m (one staged copy for short columns)
        std::vector<const uint64_t *> order(W * M);
        for (size_t e = 0; e < W; ++e)
            for (size_t m = 0; m < M; ++m) order[e * M + m] = cols[m * W + e];
        P2_TRY(h2d_columns(ctx, d_co.u(), order.data(), W * M, n * 8, 0, W * M * n * 8, ctx->stream));
        P2_TRY(p2hot_commit_many_dev(ctx, d_co.u(), M, W, log_n, rate_bits, cap_height, is_values, d_lde.u(), d_dig.u(), d_cap.u()));
        if (coeffs_out)  // [M][W][n] <- [W][M][n]: one strided copy per proof
            for (size_t m = 0; m < M; ++m)
                P2_TRY(d2h_2d(ctx, coeffs_out + m * W * n, n * 8, d_co.u() + m * n, M * n * 8, n * 8, W));
        if (digests_out && nd) P2_TRY(d2h(ctx, digests_out, d_dig.p, nd * M * 32));
        if (caps_out) P2_TRY(d2h(ctx, caps_out, d_cap.p, cap_words * M * 8));
        return P2HOT_OK;
    };
    int rc = sync_checked(ctx, body(), "commit_many");
    if (rc == P2HOT_OK && handles_out) {
        SharedBlocks *sh = new SharedBlocks;
        sh->refs = (int)M;
        sh->lde = d_lde.p;
        sh->dig = d_dig.p;
        sh->coef = d_co.p;
        for (size_t m = 0; m < M; ++m) {
            p2hot_batch *b = new p2hot_batch{ctx, d_lde.u() + m * N, W, N, d_dig.u() + m * nd * 4, log_N, cap_height};
            b->d_coef = d_co.u() + m * n;
            b->log_n = log_n;
            b->rate_bits = rate_bits;
            b->lde_stride = M * N;
            b->coef_stride = M * n;
            b->shared = sh;
            handles_out[m] = b;
        }
        d_lde.p = d_dig.p = d_co.p = nullptr;  // owned by the handles now
    }
    return rc;
}

// a handle over device buffers the caller owns (the *_dev flow: p2hot_commit_dev outputs); nothing is copied or freed
extern "C" int p2hot_batch_wrap_dev(p2hot_ctx *ctx, const uint64_t *d_coeffs, const uint64_t *d_lde, const uint64_t *d_digests,
                                    size_t W, unsigned log_n, unsigned rate_bits, unsigned cap_height, p2hot_batch **out) {
    if (!ctx || !out) return P2HOT_EINVAL;
    *out = nullptr;
    P2_TRY(check_log(ctx, log_n + rate_bits, "batch_wrap"));
    if (cap_height > log_n + rate_bits) P2_FAIL(ctx, P2HOT_EINVAL, "batch_wrap: cap_height > log2(N)");
    if (W && (!d_coeffs || !d_lde)) P2_FAIL(ctx, P2HOT_EINVAL, "batch_wrap: null buffer");
    if (p2hot_num_digests(log_n + rate_bits, cap_height) && !d_digests) P2_FAIL(ctx, P2HOT_EINVAL, "batch_wrap: null digests");
    p2hot_batch *b = new p2hot_batch{ctx, (u64 *)d_lde, W, ((size_t)1 << log_n) << rate_bits, (u64 *)d_digests, log_n + rate_bits, cap_height};
    b->d_coef = (u64 *)d_coeffs;
    b->log_n = log_n;
    b->rate_bits = rate_bits;
    b->owned = false;
    *out = b;
    return P2HOT_OK;
}

extern "C" size_t p2hot_batch_width(const p2hot_batch *b) { return b ? b->W : 0; }
extern "C" size_t p2hot_batch_leaf_width(const p2hot_batch *b) { return b ? b->W + b->S : 0; }
extern "C" unsigned p2hot_batch_degree_log(const p2hot_batch *b) { return b ? b->log_n : 0; }

// the kept input values of a from_values batch as a BORROWED column set (valid while the batch lives; free the view
// with p2hot_cols_free, which leaves the batch's block alone)
extern "C" int p2hot_batch_values(p2hot_batch *b, p2hot_cols **out) {
    if (!b || !out) return P2HOT_EINVAL;
    *out = nullptr;
    if (!b->d_vals) P2_FAIL(b->ctx, P2HOT_EINVAL, "batch_values: the batch was not committed with P2HOT_KEEP_VALUES");
    *out = new p2hot_cols{b->ctx, b->d_vals, b->W, b->log_n, false};
    return P2HOT_OK;
}

// polynomials [first, first + count) of the batch as VALUES on the subgroup H_n (an owned column set): a forward NTT of a copy of
// their device-resident coefficients.  What plonk/prover.rs reads as `prover_data.sigmas` (the sigma polynomials' values, row
// by row: prover.rs:413) is this for the sigma range of the constants_sigmas commitment -- the input of p2hot_partial_products.
extern "C" int p2hot_batch_subgroup_values(p2hot_batch *b, size_t first, size_t count, p2hot_cols **out) {
    if (!b || !out) return P2HOT_EINVAL;
    p2hot_ctx *ctx = b->ctx;
    P2_ENTER(ctx);
    *out = nullptr;
    if (first > b->W || count > b->W - first || count == 0) P2_FAIL(ctx, P2HOT_EINVAL, "batch_subgroup_values: polynomials [%zu,+%zu) of %zu", first, count, b->W);
    const size_t n = (size_t)1 << b->log_n;
    PoolBuf d(ctx);
    P2_TRY(pool_alloc(ctx, count * n * 8, &d.p));
    auto body = [&]() -> int {
        P2_HIP(ctx, hipMemcpy2DAsync(d.p, n * 8, b->d_coef + first * b->col_stride_coef(), b->col_stride_coef() * 8, n * 8, count,
                                     hipMemcpyDeviceToDevice, ctx->stream));
        P2_TRY(p2hot_fft_dev(ctx, d.u(), count, n, b->log_n));
        P2HOT_LAUNCH(ntt::canon_kernel, dim3(cdiv(count * n, 256)), dim3(256), 0, ctx->stream, d.u(), count * n);
        P2_LAUNCH_CHECK(ctx);
        return P2HOT_OK;
    };
    int rc = sync_checked(ctx, body(), "batch_subgroup_values");
    if (rc != P2HOT_OK) return rc;
    *out = new p2hot_cols{ctx, d.u(), count, b->log_n, true};
    d.p = nullptr;
    return P2HOT_OK;
}

extern "C" int p2hot_batch_coeffs(p2hot_batch *b, size_t first, size_t count, uint64_t *out) {
    if (!b) return P2HOT_EINVAL;
    p2hot_ctx *ctx = b->ctx;
    P2_ENTER(ctx);
    if (first > b->W || count > b->W - first) P2_FAIL(ctx, P2HOT_EINVAL, "batch_coeffs: polynomials [%zu,+%zu) of %zu", first, count, b->W);
    if (count == 0) return P2HOT_OK;
    if (!out) P2_FAIL(ctx, P2HOT_EINVAL, "batch_coeffs: null output");
    const size_t n = (size_t)1 << b->log_n;
    PoolBuf tmp(ctx);
    P2_TRY(pool_alloc(ctx, count * n * 8, &tmp.p));
    P2_HIP(ctx, hipMemcpy2DAsync(tmp.p, n * 8, b->d_coef + first * b->col_stride_coef(), b->col_stride_coef() * 8, n * 8, count,
                                 hipMemcpyDeviceToDevice, ctx->stream));
    P2HOT_LAUNCH(ntt::canon_kernel, dim3(cdiv(count * n, 256)), dim3(256), 0, ctx->stream, tmp.u(), count * n);
    P2_LAUNCH_CHECK(ctx);
    P2_HIP(ctx, hipMemcpyAsync(out, tmp.p, count * n * 8, hipMemcpyDeviceToHost, ctx->stream));
    return sync_checked(ctx, P2HOT_OK, "batch_coeffs");
}

extern "C" int p2hot_batch_digests(p2hot_batch *b, uint64_t *out) {
    if (!b) return P2HOT_EINVAL;
    p2hot_ctx *ctx = b->ctx;
    P2_ENTER(ctx);
    const size_t nd = p2hot_num_digests(b->log_N, b->cap_height);
    if (nd == 0) return P2HOT_OK;
    if (!out) P2_FAIL(ctx, P2HOT_EINVAL, "batch_digests: null output");
    P2_HIP(ctx, hipMemcpyAsync(out, b->d_dig, nd * 32, hipMemcpyDeviceToHost, ctx->stream));
    return sync_checked(ctx, P2HOT_OK, "batch_digests");
}

extern "C" int p2hot_batch_rows(p2hot_batch *b, const uint64_t *row_idx, size_t m, uint64_t *out) {
    if (!b) return P2HOT_EINVAL;
    p2hot_ctx *ctx = b->ctx;
    P2_ENTER(ctx);
    const size_t LW = b->W + b->S;  // MerkleTree::get returns the whole leaf, salt included (get_lde_values strips it, oracle.rs:146)
    if (m == 0 || LW == 0) return P2HOT_OK;
    if (!row_idx || !out) P2_FAIL(ctx, P2HOT_EINVAL, "batch_rows: null buffer");
    for (size_t i = 0; i < m; ++i)
        if (row_idx[i] >= b->N) P2_FAIL(ctx, P2HOT_EINVAL, "batch_rows: index %llu out of range", (unsigned long long)row_idx[i]);
    PoolBuf d_idx(ctx), d_out(ctx);
    P2_TRY(pool_alloc(ctx, m * 8, &d_idx.p));
    P2_TRY(pool_alloc(ctx, m * LW * 8, &d_out.p));
    auto body = [&]() -> int {
        P2_HIP(ctx, hipMemcpyAsync(d_idx.p, row_idx, m * 8, hipMemcpyHostToDevice, ctx->stream));
        P2_TRY(p2hot_gather_rows_dev(ctx, b->d_lde, b->col_stride_lde(), b->N, LW, d_idx.u(), m, d_out.u()));
        P2_TRY(d2h(ctx, out, d_out.p, m * LW * 8));
        return P2HOT_OK;
    };
    return sync_checked(ctx, body(), "batch_rows");
}

extern "C" int p2hot_batch_paths(p2hot_batch *b, const uint64_t *leaf_idx, size_t m, uint64_t *out) {
    if (!b) return P2HOT_EINVAL;
    p2hot_ctx *ctx = b->ctx;
    P2_ENTER(ctx);
    const unsigned layers = b->log_N - b->cap_height;
    if (m == 0 || layers == 0) return P2HOT_OK;
    if (!leaf_idx || !out) P2_FAIL(ctx, P2HOT_EINVAL, "batch_paths: null buffer");
    for (size_t i = 0; i < m; ++i)
        if (leaf_idx[i] >= b->N) P2_FAIL(ctx, P2HOT_EINVAL, "batch_paths: index %llu out of range", (unsigned long long)leaf_idx[i]);
    PoolBuf d_idx(ctx), d_out(ctx);
    P2_TRY(pool_alloc(ctx, m * 8, &d_idx.p));
    P2_TRY(pool_alloc(ctx, m * layers * 32, &d_out.p));
    auto body = [&]() -> int {
        P2_HIP(ctx, hipMemcpyAsync(d_idx.p, leaf_idx, m * 8, hipMemcpyHostToDevice, ctx->stream));
        P2_TRY(p2hot_merkle_paths_dev(ctx, b->d_dig, b->log_N, b->cap_height, d_idx.u(), m, d_out.u()));
        P2_TRY(d2h(ctx, out, d_out.p, m * layers * 32));
        return P2HOT_OK;
    };
    return sync_checked(ctx, body(), "batch_paths");
}

extern "C" void p2hot_batch_free(p2hot_batch *b) {
    if (!b) return;
    leafcopy_finish(b);  // P2HOT_LEAVES_ASYNC: the caller's buffer is complete (and the staging block idle) when this returns
    if (b->shared) {  // a member of a batched commitment: the last one returns the blocks
        if (b->shared->refs.fetch_sub(1) == 1) {
            (void)hipStreamSynchronize(b->ctx->stream);
            pool_release(b->ctx, b->shared->lde);
            pool_release(b->ctx, b->shared->dig);
            pool_release(b->ctx, b->shared->coef);
            delete b->shared;
        }
    } else if (b->owned) {
        (void)hipStreamSynchronize(b->ctx->stream);
        pool_release(b->ctx, b->d_lde);
        pool_release(b->ctx, b->d_dig);
        pool_release(b->ctx, b->d_coef);
        pool_release(b->ctx, b->d_vals);
    }
    delete b;
}

// returns the cached free blocks of the host-pointer entry points to the driver (the cache is grow-only otherwise)
extern "C" int p2hot_ctx_trim(p2hot_ctx *ctx) {
    P2_ENTER(ctx);
    P2_HIP(ctx, stream_sync(ctx));
    {
        std::lock_guard<std::mutex> pool_lock_(ctx->pool_mu);
        for (auto &blk : ctx->pool_free) (void)hipFree(blk.first);
        ctx->pool_free.clear();
        for (auto &blk : ctx->host_pool_free) (void)hipHostFree(blk.first);
        ctx->host_pool_free.clear();
    }
    // the per-size L_0 denominator tables of p2hot_quotient_polys (8 bytes per point of the quotient coset: 64 MB at 2^20 gates);
    // rebuilt in 0.35 ms by the next call that needs one
    for (auto it = ctx->twid_cache.begin(); it != ctx->twid_cache.end();) {
        if (std::get<0>(it->first) == 100) {
            (void)hipFree(it->second);
            it = ctx->twid_cache.erase(it);
        } else {
            ++it;
        }
    }
    for (p2hot_ctx *h : ctx->helpers) {  // the sibling contexts of p2hot_prove_openings_many keep their own block caches
        P2_HIP(ctx, stream_sync(h));
        std::lock_guard<std::mutex> pool_lock_(h->pool_mu);
        for (auto &blk : h->pool_free) (void)hipFree(blk.first);
        h->pool_free.clear();
    }
    return P2HOT_OK;
}

// ------------------------------------------------------------------ OpeningSet::new (plonk/proof.rs:314-327)
// What prove_openings needs to know about an oracle: its coefficient polynomials on THIS context, and -- when the tree
// lives on this context too -- its LDE matrix and digest array.  d_lde == NULL: the rows and paths of the initial trees
// are served elsewhere (a sharded batch: by the rank that owns the row) through `open_initial`.
struct OracleView {
    const u64 *d_coef, *d_lde, *d_dig;
    size_t W, N;
    size_t S = 0;  // salt columns behind the W polynomial columns of d_lde (blinded oracles): a leaf is W + S words
    size_t lde_stride = 0, coef_stride = 0;  // elements between columns when they are interleaved with other proofs' (0: N / n)
    size_t leaf_width() const { return W + S; }
    size_t col_lde() const { return lde_stride ? lde_stride : N; }
    size_t col_coef(unsigned log_n) const { return coef_stride ? coef_stride : ((size_t)1 << log_n); }
};
// fills proof->initial_leaves / initial_paths (query-major layout) for the Q host-resident query indices
typedef std::function<int(const u64 *idx, size_t Q, u64 *leaves_out, u64 *paths_out)> InitialOpener;

// OpeningSet::new for oracles given as views: a device table of pointers to every polynomial in order, one
// p2hot_eval_polys_dev call, then per-oracle copies into the caller's layout ([n_points][W_o][2] per oracle)
static int eval_openings_core(p2hot_ctx *ctx, const std::vector<OracleView> &views, unsigned log_n, const uint64_t *points, size_t n_points,
                              uint64_t *out);

extern "C" int p2hot_eval_openings(p2hot_ctx *ctx, const p2hot_batch *const *batches, size_t n_batches, const uint64_t *points,
                                   size_t n_points, uint64_t *out) {
    P2_ENTER(ctx);
    if (n_batches == 0 || n_points == 0) return P2HOT_OK;
    if (!batches || !points || !out) P2_FAIL(ctx, P2HOT_EINVAL, "eval_openings: null argument");
    std::vector<OracleView> views;
    for (size_t b = 0; b < n_batches; ++b) {
        const p2hot_batch *B = batches[b];
        if (!B || B->ctx != ctx) P2_FAIL(ctx, P2HOT_EINVAL, "oracle %zu is null or belongs to another context", b);
        if (B->log_n != batches[0]->log_n)
            P2_FAIL(ctx, P2HOT_EINVAL, "all oracles must have the same degree (oracle %zu: 2^%u vs 2^%u)", b, B->log_n, batches[0]->log_n);
        views.push_back(OracleView{B->d_coef, B->d_lde, B->d_dig, B->W, B->N, B->S, B->lde_stride, B->coef_stride});
    }
    return eval_openings_core(ctx, views, batches[0]->log_n, points, n_points, out);
}

static int eval_openings_core(p2hot_ctx *ctx, const std::vector<OracleView> &views, unsigned log_n, const uint64_t *points, size_t n_points,
                              uint64_t *out) {
    std::vector<const u64 *> ptrs;
    for (auto &v : views)
        for (size_t j = 0; j < v.W; ++j) ptrs.push_back(v.d_coef + j * v.col_coef(log_n));
    const size_t total = ptrs.size();
    if (total == 0) return P2HOT_OK;
    PoolBuf d_table(ctx), d_res(ctx), d_out(ctx);
    P2_TRY(pool_alloc(ctx, total * sizeof(u64 *), &d_table.p));
    P2_TRY(pool_alloc(ctx, n_points * total * 16, &d_res.p));
    const bool one_copy = views.size() <= 8 && total <= 0xFFFFFFFFull;
    if (one_copy) P2_TRY(pool_alloc(ctx, n_points * total * 16, &d_out.p));
    auto body = [&]() -> int {
        P2_HIP(ctx, hipMemcpyAsync(d_table.p, ptrs.data(), total * sizeof(u64 *), hipMemcpyHostToDevice, ctx->stream));
        // device layout [n_points][total][2]; the caller's layout is per oracle [n_points][W_o][2]
        P2_TRY(p2hot_eval_polys_dev(ctx, (const uint64_t *const *)d_table.p, total, log_n, points, n_points, d_res.u()));
        if (one_copy) {  // reordered on the device, one copy back (each copy into pageable memory costs ~20 us of host time)
            fri::OpeningLayout lay{};
            lay.n_oracles = (unsigned)views.size();
            size_t first = 0;
            for (size_t o = 0; o < views.size(); ++o) {
                lay.width[o] = (unsigned)views[o].W;
                lay.first[o] = (unsigned)first;
                first += views[o].W;
            }
            P2HOT_LAUNCH(fri::reorder_openings_kernel, dim3(cdiv(n_points * total, 256)), dim3(256), 0, ctx->stream, (const u64 *)d_res.u(),
                         total, n_points, lay, d_out.u());
            P2_LAUNCH_CHECK(ctx);
            P2_TRY(d2h(ctx, out, d_out.p, n_points * total * 16));
            return P2HOT_OK;
        }
        size_t off = 0, out_off = 0;
        for (auto &v : views) {
            for (size_t p = 0; p < n_points && v.W; ++p)
                P2_HIP(ctx, hipMemcpyAsync(out + out_off + 2 * p * v.W, d_res.u() + 2 * (p * total + off), v.W * 16, hipMemcpyDeviceToHost,
                                           ctx->stream));
            off += v.W;
            out_off += 2 * n_points * v.W;
        }
        return P2HOT_OK;
    };
    return sync_checked(ctx, body(), "eval_openings");  // `ptrs` outlives the copy: the stream is idle on return
}

// ------------------------------------------------------------------ prove_openings + fri_proof
static int fri_check_params(p2hot_ctx *ctx, const p2hot_fri_params *fp, unsigned log_n) {
    if (!fp) P2_FAIL(ctx, P2HOT_EINVAL, "null fri params");
    if (fp->n_reduction_rounds && !fp->reduction_arity_bits) P2_FAIL(ctx, P2HOT_EINVAL, "null reduction_arity_bits");
    P2_TRY(check_log(ctx, log_n + fp->rate_bits, "fri"));
    unsigned lm = log_n + fp->rate_bits, ln = log_n;
    for (unsigned r = 0; r < fp->n_reduction_rounds; ++r) {
        const unsigned ab = fp->reduction_arity_bits[r];
        if (ab == 0 || ab > ln) P2_FAIL(ctx, P2HOT_EINVAL, "fri: round %u arity 2^%u does not divide the degree bound", r, ab);
        if (lm - ab < fp->cap_height) P2_FAIL(ctx, P2HOT_EINVAL, "fri: round %u tree has fewer leaves than the cap (merkle_tree.rs:195-200)", r);
        lm -= ab;
        ln -= ab;
    }
    return P2HOT_OK;
}

// sizes of the flat FriProof buffers for oracles of the given widths (include/p2hot.h, p2hot_fri_proof)
static int fri_proof_layout(const size_t *widths, size_t n_oracles, unsigned log_n, const p2hot_fri_params *fp, p2hot_fri_proof_layout *out) {
    if (!out || !fp || n_oracles == 0 || (fp->n_reduction_rounds && !fp->reduction_arity_bits)) return P2HOT_EINVAL;
    const unsigned log_N = log_n + fp->rate_bits;
    size_t w_sum = 0;
    for (size_t o = 0; o < n_oracles; ++o) w_sum += widths[o];
    const size_t q = fp->num_query_rounds, cap_words = (size_t)4 << fp->cap_height;
    size_t evals = 0, paths = 0;
    unsigned lm = log_N;
    for (unsigned r = 0; r < fp->n_reduction_rounds; ++r) {
        const unsigned ab = fp->reduction_arity_bits[r];
        if (ab > lm || lm - ab < fp->cap_height) return P2HOT_EINVAL;
        evals += (size_t)2 << ab;
        paths += 4 * (size_t)(lm - ab - fp->cap_height);
        lm -= ab;
    }
    if (lm < fp->rate_bits || log_N < fp->cap_height) return P2HOT_EINVAL;
    out->caps_words = fp->n_reduction_rounds * cap_words;
    out->final_poly_words = (size_t)2 << (lm - fp->rate_bits);
    out->initial_leaves_words = q * w_sum;
    out->initial_paths_words = q * n_oracles * 4 * (size_t)(log_N - fp->cap_height);
    out->step_evals_words = q * evals;
    out->step_paths_words = q * paths;
    return P2HOT_OK;
}

extern "C" int p2hot_fri_proof_sizes(const p2hot_batch *const *oracles, size_t n_oracles, const p2hot_fri_params *fp,
                                     p2hot_fri_proof_layout *out) {
    if (!oracles || n_oracles == 0 || !oracles[0]) return P2HOT_EINVAL;
    std::vector<size_t> widths;
    for (size_t o = 0; o < n_oracles; ++o) {
        if (!oracles[o]) return P2HOT_EINVAL;
        widths.push_back(oracles[o]->W + oracles[o]->S);  // evals_proofs carry whole leaves (fri/prover.rs:238-241)
    }
    return fri_proof_layout(widths.data(), n_oracles, oracles[0]->log_n, fp, out);
}

static int prove_openings_core(p2hot_ctx *ctx, const p2hot_fri_batch_info *batches, size_t n_batches, const std::vector<OracleView> &views,
                               unsigned log_n, p2hot_challenger *challenger, const p2hot_fri_params *fp, p2hot_fri_proof *proof,
                               const InitialOpener *open_initial);

extern "C" int p2hot_prove_openings(p2hot_ctx *ctx, const p2hot_fri_batch_info *batches, size_t n_batches,
                                    const p2hot_batch *const *oracles, size_t n_oracles, p2hot_challenger *challenger,
                                    const p2hot_fri_params *fp, p2hot_fri_proof *proof) {
    P2_ENTER(ctx);
    if (!oracles || n_oracles == 0) P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings: null argument");
    std::vector<OracleView> views;
    for (size_t o = 0; o < n_oracles; ++o) {
        if (!oracles[o] || oracles[o]->ctx != ctx) P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings: oracle %zu is null or belongs to another context", o);
        if (oracles[o]->log_n != oracles[0]->log_n || (fp && (oracles[o]->rate_bits != fp->rate_bits || oracles[o]->cap_height != fp->cap_height)))
            P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings: oracle %zu was committed with another degree / rate / cap height", o);
        views.push_back(OracleView{oracles[o]->d_coef, oracles[o]->d_lde, oracles[o]->d_dig, oracles[o]->W, oracles[o]->N, oracles[o]->S,
                                   oracles[o]->lde_stride, oracles[o]->coef_stride});
    }
    return prove_openings_core(ctx, batches, n_batches, views, oracles[0]->log_n, challenger, fp, proof, nullptr);
}

// ------------------------------------------------------------------ M opening proofs side by side
// A recursion-size opening proof (2^12 rows) is a chain of small launches -- about 36 dependent challenger permutations, tree
// levels of a few hundred nodes -- that leaves the chip almost idle.  M independent proofs therefore run on up to 4 sibling
// contexts of the same GPU (their own streams, scratch and block caches), each driven by its own host thread, so that the
// launches of different proofs overlap on the device.  Every proof is computed by the same prove_openings_core as a single
// p2hot_prove_openings call: results are identical, buffer by buffer.
static int helper_contexts(p2hot_ctx *ctx, size_t want) {
    while (ctx->helpers.size() < want) {
        hipStream_t st = nullptr;
        P2_HIP(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        p2hot_ctx *h = nullptr;
        int rc = p2hot_ctx_create(ctx->device, (void *)st, &h);
        if (rc != P2HOT_OK) {
            ctx->err = std::string("prove_openings_many: helper context: ") + (h ? h->err : "allocation failed");
            if (h) p2hot_ctx_destroy(h);
            return rc;
        }
        p2hot_challenger *hc = nullptr;
        rc = p2hot_challenger_create(h, &hc);
        if (rc != P2HOT_OK) {
            ctx->err = "prove_openings_many: helper challenger: " + h->err;
            p2hot_ctx_destroy(h);
            return rc;
        }
        ctx->helpers.push_back(h);
        ctx->helper_streams.push_back(st);
        ctx->helper_challengers.push_back(hc);
    }
    // the parent's tuning knobs and profiling switch as they are NOW (p2hot_tune_* / p2hot_profile_enable may have been called
    // since the helpers were made)
    for (p2hot_ctx *h : ctx->helpers) {
        h->quad_threshold = ctx->quad_threshold;
        h->row_threshold = ctx->row_threshold;
        h->use_regpass = ctx->use_regpass;
        h->use_limb = ctx->use_limb;
        h->ntt_radix_bits = ctx->ntt_radix_bits;
        h->overlap = ctx->overlap;
        h->horner_two_level_min = ctx->horner_two_level_min;
        h->profiling = ctx->profiling;
    }
    return P2HOT_OK;
}

extern "C" int p2hot_prove_openings_many(p2hot_ctx *ctx, size_t M, const p2hot_fri_batch_info *const *batches, const size_t *n_batches,
                                         const p2hot_batch *const *oracles, size_t n_oracles, p2hot_challenger *const *challengers,
                                         const p2hot_fri_params *fp, p2hot_fri_proof *proofs) {
    P2_ENTER(ctx);
    if (M == 0) return P2HOT_OK;
    if (!batches || !n_batches || !oracles || n_oracles == 0 || !challengers || !proofs) P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings_many: null argument");
    std::vector<std::vector<OracleView>> views(M);
    for (size_t j = 0; j < M; ++j) {
        if (!challengers[j] || challengers[j]->ctx != ctx) P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings_many: challenger %zu is null or belongs to another context", j);
        for (size_t o = 0; o < n_oracles; ++o) {
            const p2hot_batch *B = oracles[j * n_oracles + o];
            if (!B || B->ctx != ctx) P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings_many: oracle %zu of proof %zu is null or belongs to another context", o, j);
            if (B->log_n != oracles[0]->log_n || (fp && (B->rate_bits != fp->rate_bits || B->cap_height != fp->cap_height)))
                P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings_many: oracle %zu of proof %zu was committed with another degree / rate / cap height", o, j);
            views[j].push_back(OracleView{B->d_coef, B->d_lde, B->d_dig, B->W, B->N, B->S, B->lde_stride, B->coef_stride});
        }
    }
    size_t k_max = 4;  // sibling contexts: 2 and 4 measured 1.8-1.9x a single stream, 8+ no better (P2HOT_MANY_HELPERS overrides; tools/pom_trace.py)
    if (const char *e = getenv("P2HOT_MANY_HELPERS")) k_max = std::max<size_t>(1, std::min<size_t>(64, (size_t)strtoull(e, nullptr, 10)));
    const size_t K = std::min<size_t>(M, k_max);
    P2_TRY(helper_contexts(ctx, K));
    P2_HIP(ctx, stream_sync(ctx));  // the oracles and transcripts were produced on this context's stream
    const unsigned log_n = oracles[0]->log_n;
    std::vector<int> rcs(K, P2HOT_OK);
    std::vector<std::string> errs(K);
    auto work = [&](size_t k) {
        p2hot_ctx *h = ctx->helpers[k];
        p2hot_challenger *hc = ctx->helper_challengers[k];
        DeviceGuard dev_guard_(h);
        CallGuard guard_(h);
        for (size_t j = k; j < M && rcs[k] == P2HOT_OK; j += K) {
            auto one = [&]() -> int {
                P2_HIP(h, hipMemcpyAsync(hc->d, challengers[j]->d, sizeof(fri::Challenger), hipMemcpyDeviceToDevice, h->stream));
                P2_TRY(prove_openings_core(h, batches[j], n_batches[j], views[j], log_n, hc, fp, &proofs[j], nullptr));
                P2_HIP(h, hipMemcpyAsync(challengers[j]->d, hc->d, sizeof(fri::Challenger), hipMemcpyDeviceToDevice, h->stream));
                P2_HIP(h, stream_sync(h));
                return P2HOT_OK;
            };
            rcs[k] = one();
            if (rcs[k] != P2HOT_OK) errs[k] = "proof " + std::to_string(j) + ": " + h->err;
        }
    };
#ifdef P2HOT_EMU
    for (size_t k = 0; k < K; ++k) work(k);  // the kernel emulator is single-threaded
#else
    {
        std::vector<std::thread> th;
        for (size_t k = 1; k < K; ++k) th.emplace_back(work, k);
        work(0);
        for (auto &t : th) t.join();
    }
#endif
    if (ctx->profiling)  // the helpers' kernels belong to this call: their totals go to the parent's table
        for (size_t k = 0; k < K; ++k) {
            p2hot_ctx *h = ctx->helpers[k];
            (void)p2hot_profile_json(h, 0);  // drains h's recorded events into h->prof_acc
            for (auto &kv : h->prof_acc) {
                auto &acc = ctx->prof_acc[kv.first];
                acc.first += kv.second.first;
                acc.second += kv.second.second;
            }
            h->prof_acc.clear();
        }
    for (size_t k = 0; k < K; ++k)
        if (rcs[k] != P2HOT_OK) P2_FAIL(ctx, rcs[k], "prove_openings_many: %s", errs[k].c_str());
    return P2HOT_OK;
}

static int prove_openings_core(p2hot_ctx *ctx, const p2hot_fri_batch_info *batches, size_t n_batches, const std::vector<OracleView> &views,
                               unsigned log_n, p2hot_challenger *challenger, const p2hot_fri_params *fp, p2hot_fri_proof *proof,
                               const InitialOpener *open_initial) {
    const size_t n_oracles = views.size();
    if (!challenger || challenger->ctx != ctx) P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings: the challenger belongs to another context");
    if (!proof || (n_batches && !batches)) P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings: null argument");
    // FriParams::hiding changes nothing on the prover's FRI path (it is observed into the transcript by the caller,
    // fri/mod.rs:148, and tells the VERIFIER to strip the salts, fri/verifier.rs:149-151): blinded oracles carry their salt columns
    P2_TRY(fri_check_params(ctx, fp, log_n));
    const unsigned rate_bits = fp->rate_bits, cap_height = fp->cap_height, log_N = log_n + rate_bits, n_rounds = fp->n_reduction_rounds;
    const size_t n = (size_t)1 << log_n, N = n << rate_bits, Q = fp->num_query_rounds;
    p2hot_fri_proof_layout lay;
    {
        std::vector<size_t> widths;
        for (auto &v : views) widths.push_back(v.leaf_width());
        if (fri_proof_layout(widths.data(), n_oracles, log_n, fp, &lay) != P2HOT_OK) P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings: inconsistent parameters");
    }
    if ((lay.caps_words && !proof->commit_phase_merkle_caps) || !proof->final_poly ||
        (Q && ((lay.initial_leaves_words && !proof->initial_leaves) || (lay.initial_paths_words && !proof->initial_paths) ||
               (lay.step_evals_words && !proof->step_evals) || (lay.step_paths_words && !proof->step_paths))))
        P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings: a proof buffer is null (size them with p2hot_fri_proof_sizes)");
    const std::vector<OracleView> &oracles = views;
    // --- the polynomial table of the instance (FriInstanceInfo.batches, fri/structure.rs): device pointers in batch order
    std::vector<const u64 *> ptrs;
    std::vector<size_t> offsets(1, 0);
    std::vector<u64> points;
    for (size_t i = 0; i < n_batches; ++i) {
        const p2hot_fri_batch_info &bi = batches[i];
        if (bi.n_polys && (!bi.oracle_index || !bi.poly_index)) P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings: batch %zu has null index arrays", i);
        for (size_t j = 0; j < bi.n_polys; ++j) {
            const size_t oi = bi.oracle_index[j], pi = bi.poly_index[j];
            if (oi >= n_oracles || pi >= oracles[oi].W)
                P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings: batch %zu opens polynomial (%zu, %zu) which does not exist", i, oi, pi);
            ptrs.push_back(oracles[oi].d_coef + pi * oracles[oi].col_coef(log_n));
        }
        offsets.push_back(ptrs.size());
        points.push_back(bi.point[0]);
        points.push_back(bi.point[1]);
    }
    // --- device blocks: pointer table, final_poly planes, round trees (leaves + digests), query staging
    size_t leaf_words = 0, dig_words = 0;
    {
        size_t m = N;
        unsigned lm = log_N;
        for (unsigned r = 0; r < n_rounds; ++r) {
            const unsigned ab = fp->reduction_arity_bits[r];
            leaf_words += 2 * m;
            dig_words += 4 * p2hot_num_digests(lm - ab, cap_height);
            m >>= ab;
            lm -= ab;
        }
    }
    const unsigned layers0 = log_N - cap_height;
    PoolBuf d_table(ctx), d_planes(ctx), d_leaves(ctx), d_dig(ctx), d_q(ctx);
    P2_TRY(pool_alloc(ctx, (ptrs.size() ? ptrs.size() : 1) * sizeof(u64 *), &d_table.p));
    P2_TRY(pool_alloc(ctx, 2 * n * 8, &d_planes.p));
    P2_TRY(pool_alloc(ctx, (leaf_words ? leaf_words : 1) * 8, &d_leaves.p));
    P2_TRY(pool_alloc(ctx, (dig_words ? dig_words : 1) * 8, &d_dig.p));
    // query staging: indices [1 + R][Q] | initial leaves | initial paths | step evals | step paths | rand [Q] | alpha [2] | best | resp | saved challenger
    const size_t ch_words = (sizeof(fri::Challenger) + 7) / 8;
    const size_t q_words = Q * (1 + n_rounds) + lay.initial_leaves_words + lay.initial_paths_words + lay.step_evals_words +
                           lay.step_paths_words + Q + 2 + 1 + 1 + ch_words;
    P2_TRY(pool_alloc(ctx, q_words * 8, &d_q.p));
    u64 *d_idx = d_q.u(), *d_il = d_idx + Q * (1 + n_rounds), *d_ip = d_il + lay.initial_leaves_words,
        *d_se = d_ip + lay.initial_paths_words, *d_sp = d_se + lay.step_evals_words, *d_rand = d_sp + lay.step_paths_words,
        *d_alpha = d_rand + Q, *d_best = d_alpha + 2, *d_resp = d_best + 1, *d_chsave = d_resp + 1;
    fri::ArityBits ab{};
    if (n_rounds > 32) P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings: more than 32 reduction rounds");
    if (fp->proof_of_work_bits > 64) P2_FAIL(ctx, P2HOT_EINVAL, "prove_openings: proof_of_work_bits > 64");
    for (unsigned r = 0; r < n_rounds; ++r) ab.b[r] = (unsigned char)fp->reduction_arity_bits[r];
    size_t w_sum = 0;
    for (size_t o = 0; o < n_oracles; ++o) w_sum += oracles[o].leaf_width();
    unsigned long long best = ~0ull;
    u64 pow_next = 0;
    std::vector<u64> idx_for_owners;
    // Everything below is enqueued without waiting for the GPU: alpha, beta_i, the PoW witness and the query indices
    // stay on the device (the challenger is device-resident), results reach the caller's buffers by asynchronous copies
    // and ONE synchronisation ends the call.
    auto head = [&]() -> int {
        if (!ptrs.empty())
            P2_HIP(ctx, hipMemcpyAsync(d_table.p, ptrs.data(), ptrs.size() * sizeof(u64 *), hipMemcpyHostToDevice, ctx->stream));
        // oracle.rs:186: alpha = challenger.get_extension_challenge()
        P2_TRY(challenger_step_dev(challenger, nullptr, 0, d_alpha, 2));
        // oracle.rs:190-213: final_poly = sum_i alpha^(k_i) (F_i - F_i(z_i)) / (X - z_i)
        P2_TRY(final_poly_core(ctx, (const uint64_t *const *)d_table.p, offsets.data(), n_batches, points.data(), nullptr, d_alpha, log_n,
                               d_planes.u()));
        // oracle.rs:215-220 + fri/prover.rs:40-51: final FFT, commit phase; the round trees stay on the device
        P2_TRY(fri_commit_core(ctx, nullptr, d_planes.u(), log_n, rate_bits, cap_height, fp->reduction_arity_bits, n_rounds,
                               fp->max_num_query_steps, fp->final_poly_coeff_len, challenger, d_leaves.u(), true, d_dig.u(), true,
                               proof->commit_phase_merkle_caps, nullptr, proof->final_poly, /*defer_sync=*/true));
        // fri/prover.rs:53-58: proof of work, searched on the device; the transcript state before it is kept for the
        // (practically unreachable) case that the searched range holds no witness
        P2_HIP(ctx, hipMemcpyAsync(d_chsave, challenger->d, sizeof(fri::Challenger), hipMemcpyDeviceToDevice, ctx->stream));
        return pow_search_dev(ctx, challenger, fp->proof_of_work_bits, (unsigned long long *)d_best, &pow_next);
    };
    auto tail = [&]() -> int {
        // prover.rs:197-198: observe the witness, draw the response
        P2_TRY(challenger_step_dev(challenger, d_best, 1, d_resp, 1));
        P2_TRY(d2h(ctx, &best, d_best, 8));
        if (Q == 0) return P2HOT_OK;
        // fri/prover.rs:215-220: x_index = rand % n for num_query_rounds challenges; per round x_index >>= arity_bits (:243-253)
        P2_TRY(challenger_step_dev(challenger, nullptr, 0, d_rand, Q));
        P2HOT_LAUNCH(fri::query_indices_kernel, dim3(cdiv(Q, 256)), dim3(256), 0, ctx->stream, (const u64 *)d_rand, Q, log_N, ab, n_rounds,
                     d_idx);
        P2_LAUNCH_CHECK(ctx);
        if (proof->query_indices) P2_TRY(d2h(ctx, proof->query_indices, d_idx, Q * 8));
        if (open_initial) {  // the owners of the rows need the indices on the host
            idx_for_owners.resize(Q);
            P2_TRY(d2h(ctx, idx_for_owners.data(), d_idx, Q * 8));
        }
        // prover.rs:238-241: initial_trees_proof = for every oracle (tree.get(x), tree.prove(x)), all queries per launch.
        // Device staging is oracle-major ([oracle][q][...]); the host layout is query-major (see p2hot.h), fixed by the D2H copies.
        size_t w_off = 0;
        for (size_t o = 0; o < n_oracles && !open_initial; ++o) {
            const OracleView &B = oracles[o];
            P2_TRY(p2hot_gather_rows_dev(ctx, B.d_lde, B.col_lde(), B.N, B.leaf_width(), d_idx, Q, d_il + Q * w_off));
            P2_TRY(p2hot_merkle_paths_dev(ctx, B.d_dig, log_N, cap_height, d_idx, Q, d_ip + o * Q * 4 * layers0));
            w_off += B.leaf_width();
        }
        // prover.rs:242-253: per round (evals = unflatten(tree.get(x >> arity_bits)), tree.prove(x >> arity_bits))
        size_t ev_off = 0, pa_off = 0, lv = 0, dg = 0, m = N;
        unsigned lm = log_N;
        std::vector<size_t> ev_offs, pa_offs, ev_w, pa_w;
        for (unsigned r = 0; r < n_rounds; ++r) {
            const unsigned abr = fp->reduction_arity_bits[r];
            const size_t roww = (size_t)2 << abr, layers = lm - abr - cap_height;
            P2HOT_LAUNCH(fri::gather_rowmajor_kernel, dim3(cdiv(Q * roww, 256)), dim3(256), 0, ctx->stream, (const u64 *)(d_leaves.u() + lv),
                         roww, m >> abr, (const u64 *)(d_idx + (1 + (size_t)r) * Q), Q, d_se + Q * ev_off, ctx->d_oob);
            P2_LAUNCH_CHECK(ctx);
            if (layers)
                P2_TRY(p2hot_merkle_paths_dev(ctx, d_dig.u() + dg, lm - abr, cap_height, d_idx + (1 + (size_t)r) * Q, Q, d_sp + Q * pa_off));
            ev_offs.push_back(ev_off);
            pa_offs.push_back(pa_off);
            ev_w.push_back(roww);
            pa_w.push_back(4 * layers);
            ev_off += roww;
            pa_off += 4 * layers;
            lv += 2 * m;
            dg += 4 * p2hot_num_digests(lm - abr, cap_height);
            m >>= abr;
            lm -= abr;
        }
        // D2H: one strided copy per (oracle | round) turns the oracle-major staging into the query-major proof layout
        w_off = 0;
        for (size_t o = 0; o < n_oracles && !open_initial; ++o) {
            const size_t Wb = oracles[o].leaf_width();
            if (Wb)
                P2_TRY(d2h_2d(ctx, proof->initial_leaves + w_off, w_sum * 8, d_il + Q * w_off, Wb * 8, Wb * 8, Q));
            if (layers0)
                P2_TRY(d2h_2d(ctx, proof->initial_paths + o * 4 * layers0, n_oracles * 4 * layers0 * 8, d_ip + o * Q * 4 * layers0,
                              4 * layers0 * 8, 4 * layers0 * 8, Q));
            w_off += Wb;
        }
        for (unsigned r = 0; r < n_rounds; ++r) {
            P2_TRY(d2h_2d(ctx, proof->step_evals + ev_offs[r], ev_off * 8, d_se + Q * ev_offs[r], ev_w[r] * 8, ev_w[r] * 8, Q));
            if (pa_w[r])
                P2_TRY(d2h_2d(ctx, proof->step_paths + pa_offs[r], pa_off * 8, d_sp + Q * pa_offs[r], pa_w[r] * 8, pa_w[r] * 8, Q));
        }
        return P2HOT_OK;
    };
    int rc = head();
    if (rc == P2HOT_OK) rc = tail();
    rc = sync_checked(ctx, rc, "prove_openings");
    if (rc == P2HOT_OK && best == ~0ull) {
        // no witness among the first 2^(pow_bits + 5) candidates (probability e^-32): rewind the transcript to before the
        // grind, search the rest of the range with a host check per chunk, and replay the tail
        // (errors go through rc + sync_checked like the main path: the PoolBuf blocks must not return to the cache while
        // enqueued work still references them)
        auto fallback = [&]() -> int {
            P2_HIP(ctx, hipMemcpyAsync(challenger->d, d_chsave, sizeof(fri::Challenger), hipMemcpyDeviceToDevice, ctx->stream));
            P2_TRY(pow_continue_host(ctx, challenger, fp->proof_of_work_bits, (unsigned long long *)d_best, pow_next, &best));
            return tail();
        };
        rc = sync_checked(ctx, fallback(), "prove_openings");
    }
    if (rc == P2HOT_OK) proof->pow_witness = best;
    // a sharded batch: the initial trees' rows and paths come from the ranks that own them
    if (rc == P2HOT_OK && open_initial && Q) rc = (*open_initial)(idx_for_owners.data(), Q, proof->initial_leaves, proof->initial_paths);
    return rc;
}

// ------------------------------------------------------------------ permutation argument (plonk/prover.rs:356-449)
extern "C" int p2hot_partial_products(p2hot_ctx *ctx, const p2hot_cols *wires, size_t wires_first_col, const p2hot_cols *sigmas,
                                      size_t sigmas_first_col, const uint64_t *k_is, unsigned num_routed, unsigned degree,
                                      const uint64_t *betas, const uint64_t *gammas, unsigned num_challenges, uint64_t *out_host,
                                      p2hot_cols **out_cols) {
    P2_ENTER(ctx);
    if (out_cols) *out_cols = nullptr;
    if (!wires || !sigmas || wires->ctx != ctx || sigmas->ctx != ctx) P2_FAIL(ctx, P2HOT_EINVAL, "partial_products: null or foreign column set");
    if (wires->log_n != sigmas->log_n) P2_FAIL(ctx, P2HOT_EINVAL, "partial_products: wires and sigmas have different lengths");
    if (wires_first_col > wires->W || num_routed > wires->W - wires_first_col || sigmas_first_col > sigmas->W ||
        num_routed > sigmas->W - sigmas_first_col)
        P2_FAIL(ctx, P2HOT_EINVAL, "partial_products: %u routed columns do not fit the column sets", num_routed);
    // prover.rs:215-218: "When the number of routed wires is smaller that the degree, we should change the logic"
    if (degree < 2 || !(degree < num_routed)) P2_FAIL(ctx, P2HOT_EINVAL, "partial_products: need 2 <= degree < num_routed");
    const unsigned log_n = wires->log_n;
    const size_t n = (size_t)1 << log_n;
    const unsigned num_prods = (num_routed + degree - 1) / degree - 1;
    const size_t rows = (size_t)num_challenges * (num_prods + 1);
    PoolBuf d_out(ctx);
    P2_TRY(pool_alloc(ctx, (rows ? rows : 1) * n * 8, &d_out.p));
    auto body = [&]() -> int {
        P2_TRY(p2hot_partial_products_dev(ctx, wires->d + wires_first_col * n, n, sigmas->d + sigmas_first_col * n, n, k_is, num_routed,
                                          log_n, degree, betas, gammas, num_challenges, d_out.u(), n));
        if (out_host && rows) P2_HIP(ctx, hipMemcpyAsync(out_host, d_out.p, rows * n * 8, hipMemcpyDeviceToHost, ctx->stream));
        return P2HOT_OK;
    };
    int rc = sync_checked(ctx, body(), "partial_products");
    if (rc == P2HOT_OK && out_cols) {
        *out_cols = new p2hot_cols{ctx, d_out.u(), rows, log_n, true};
        d_out.p = nullptr;
    }
    return rc;
}

// ------------------------------------------------------------------ quotient polynomials -> chunks (plonk/prover.rs:274-289, :810-815)
// the tail both quotient entry points share: d_work holds num_challenges polynomials' VALUES on g*H of size m = n << qbits
// (natural order, one flag word behind them); coset_ifft, trim_to_len with its divisibility check, chunks of n
static int quotient_chunks_core(p2hot_ctx *ctx, PoolBuf &d_work, unsigned num_challenges, unsigned degree_bits, unsigned qbits,
                                unsigned quotient_degree_factor, const char *what, p2hot_cols **chunks_out) {
    const size_t n = (size_t)1 << degree_bits, m = n << qbits, keep = n * quotient_degree_factor;
    PoolBuf d_chunks(ctx);
    // (the caller's kernel or uploads into d_work may still be in flight: every early return synchronises before the PoolBufs go back)
    if (int rc0 = pool_alloc(ctx, (num_challenges ? (size_t)num_challenges * quotient_degree_factor : 1) * n * 8, &d_chunks.p))
        return sync_checked(ctx, rc0, what);
    unsigned nonzero = 0;
    auto body = [&]() -> int {
        // values.coset_ifft(F::coset_shift()) (prover.rs:810-814)
        P2_TRY(p2hot_coset_ifft_dev(ctx, d_work.u(), num_challenges, m, degree_bits + qbits, gl::COSET_SHIFT));
        // trim_to_len(quotient_degree_factor * n) (polynomial/mod.rs:164-178) with its divisibility check, then
        // chunks(degree) (mod.rs:136-142): the kept prefix of polynomial c is chunks c*qdf .. (c+1)*qdf-1
        unsigned *flag = (unsigned *)(d_work.u() + (size_t)num_challenges * m);
        P2_HIP(ctx, hipMemsetAsync(flag, 0, 8, ctx->stream));
        for (unsigned c = 0; c < num_challenges; ++c) {
            if (keep < m) {
                P2HOT_LAUNCH(plonk::any_nonzero_kernel, dim3(cdiv(m - keep, 256)), dim3(256), 0, ctx->stream,
                             (const u64 *)(d_work.u() + c * m + keep), m - keep, flag);
                P2_LAUNCH_CHECK(ctx);
            }
            P2_HIP(ctx, hipMemcpyAsync(d_chunks.u() + (size_t)c * keep, d_work.u() + c * m, keep * 8, hipMemcpyDeviceToDevice, ctx->stream));
        }
        if (num_challenges) {
            P2HOT_LAUNCH(ntt::canon_kernel, dim3(cdiv((size_t)num_challenges * keep, 256)), dim3(256), 0, ctx->stream, d_chunks.u(),
                         (size_t)num_challenges * keep);
            P2_LAUNCH_CHECK(ctx);
        }
        P2_HIP(ctx, hipMemcpyAsync(&nonzero, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
        return P2HOT_OK;
    };
    int rc = sync_checked(ctx, body(), what);
    if (rc != P2HOT_OK) return rc;
    if (nonzero) P2_FAIL(ctx, P2HOT_EINVAL, "Quotient has failed, the vanishing polynomial is not divisible by Z_H");
    *chunks_out = new p2hot_cols{ctx, d_chunks.u(), (size_t)num_challenges * quotient_degree_factor, degree_bits, true};
    d_chunks.p = nullptr;
    return P2HOT_OK;
}

extern "C" int p2hot_quotient_chunks(p2hot_ctx *ctx, const uint64_t *const *quotient_values, unsigned num_challenges,
                                     unsigned degree_bits, unsigned quotient_degree_factor, p2hot_cols **chunks_out) {
    P2_ENTER(ctx);
    if (!chunks_out) P2_FAIL(ctx, P2HOT_EINVAL, "quotient_chunks: null output");
    *chunks_out = nullptr;
    if (quotient_degree_factor == 0 || (num_challenges && !quotient_values)) P2_FAIL(ctx, P2HOT_EINVAL, "quotient_chunks: bad arguments");
    unsigned qbits = 0;
    while ((1u << qbits) < quotient_degree_factor) ++qbits;  // log2_ceil (circuit_data.rs quotient_degree_bits)
    P2_TRY(check_log(ctx, degree_bits + qbits, "quotient_chunks"));
    const size_t n = (size_t)1 << degree_bits, m = n << qbits;
    for (unsigned c = 0; c < num_challenges; ++c)  // every argument is checked before the first copy is queued
        if (!quotient_values[c]) P2_FAIL(ctx, P2HOT_EINVAL, "quotient_chunks: polynomial %u is null", c);
    PoolBuf d_work(ctx);
    P2_TRY(pool_alloc(ctx, (num_challenges ? num_challenges : 1) * m * 8 + 8, &d_work.p));
    auto upload = [&]() -> int {
        for (unsigned c = 0; c < num_challenges; ++c)
            P2_HIP(ctx, hipMemcpyAsync(d_work.u() + c * m, quotient_values[c], m * 8, hipMemcpyHostToDevice, ctx->stream));
        return P2HOT_OK;
    };
    if (int rc0 = upload()) return sync_checked(ctx, rc0, "quotient_chunks");  // earlier copies into d_work are drained before it goes back
    return quotient_chunks_core(ctx, d_work, num_challenges, degree_bits, qbits, quotient_degree_factor, "quotient_chunks", chunks_out);
}

// compute_quotient_polys (plonky2/src/plonk/prover.rs:609-815) without its gate evaluation: the permutation argument's vanishing
// terms on the quotient coset (plonk/vanishing_poly.rs:167-330; kernel: plonk::quotient_perm_kernel) from the three commitments'
// device-resident LDE matrices, plus the caller's reduced gate terms, over Z_H; then the same tail as p2hot_quotient_chunks.
extern "C" int p2hot_quotient_polys(p2hot_ctx *ctx, const p2hot_batch *wires, const p2hot_batch *constants_sigmas, size_t sigmas_first_col,
                                    const p2hot_batch *zs_partial_products, const uint64_t *k_is, unsigned num_routed,
                                    unsigned quotient_degree_factor, const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas,
                                    unsigned num_challenges, const uint64_t *const *gate_sums, uint64_t *values_out, p2hot_cols **chunks_out) {
    P2_ENTER(ctx);
    if (chunks_out) *chunks_out = nullptr;
    if (!wires || !constants_sigmas || !zs_partial_products || !k_is || !betas || !gammas || !alphas) P2_FAIL(ctx, P2HOT_EINVAL, "quotient_polys: null argument");
    if (!chunks_out && !values_out) P2_FAIL(ctx, P2HOT_EINVAL, "quotient_polys: nothing asked for");
    const p2hot_batch *bs[3] = {wires, constants_sigmas, zs_partial_products};
    for (const p2hot_batch *b : bs) {
        if (b->ctx != ctx) P2_FAIL(ctx, P2HOT_EINVAL, "quotient_polys: a commitment belongs to another context");
        if (b->log_n != wires->log_n || b->rate_bits != wires->rate_bits) P2_FAIL(ctx, P2HOT_EINVAL, "quotient_polys: the commitments differ in degree or rate");
    }
    if (num_challenges == 0 || num_challenges > 4) P2_FAIL(ctx, P2HOT_EUNSUPPORTED, "quotient_polys: %u challenges (1..4 supported; plonky2's configs use 2)", num_challenges);
    if (quotient_degree_factor < 2 || num_routed == 0) P2_FAIL(ctx, P2HOT_EINVAL, "quotient_polys: bad quotient degree factor or no routed wires");
    unsigned qbits = 0;
    while ((1u << qbits) < quotient_degree_factor) ++qbits;
    // "Having constraints of degree higher than the rate is not supported yet." (prover.rs:632-636)
    if (qbits > wires->rate_bits) P2_FAIL(ctx, P2HOT_EINVAL, "quotient_polys: quotient degree 2^%u above the rate 2^%u (prover.rs:632-636)", qbits, wires->rate_bits);
    const unsigned degree_bits = wires->log_n, log_nq = degree_bits + qbits;
    const unsigned num_chunks = (num_routed + quotient_degree_factor - 1) / quotient_degree_factor, num_prods = num_chunks - 1;
    if (wires->W < num_routed || constants_sigmas->W < sigmas_first_col + num_routed || zs_partial_products->W < (size_t)num_challenges * (1 + num_prods))
        P2_FAIL(ctx, P2HOT_EINVAL, "quotient_polys: a commitment is narrower than the permutation argument needs");
    const size_t n = (size_t)1 << degree_bits, m = n << qbits, rate = (size_t)1 << qbits;
    for (unsigned c = 0; c < num_challenges && gate_sums; ++c)
        if (!gate_sums[c]) P2_FAIL(ctx, P2HOT_EINVAL, "quotient_polys: gate_sums[%u] is null", c);
    PoolBuf d_work(ctx), d_small(ctx), d_gate(ctx);
    P2_TRY(pool_alloc(ctx, (size_t)num_challenges * m * 8 + 8, &d_work.p));
    const size_t nbk = (size_t)num_challenges * num_routed;
    P2_TRY(pool_alloc(ctx, (nbk + 2 * rate) * 8, &d_small.p));
    if (gate_sums) P2_TRY(pool_alloc(ctx, (size_t)num_challenges * m * 8, &d_gate.p));
    // ZeroPolyOnCoset::new(degree_bits, qbits) (field/src/zero_poly_coset.rs:21-34) and beta_c * k_j, on the host
    std::vector<u64> small(nbk + 2 * rate);
    for (unsigned c = 0; c < num_challenges; ++c)
        for (unsigned j = 0; j < num_routed; ++j) small[(size_t)c * num_routed + j] = gl::canon(gl::mul(betas[c], k_is[j]));
    const size_t num_routed_ = nbk;  // (offset of the Z_H table behind the beta * k table)
    const u64 g_pow_n = gl::pow(gl::COSET_SHIFT, n), v = gl::root_of_unity(qbits);
    for (size_t j = 0; j < rate; ++j) {
        const u64 e = gl::canon(gl::sub(gl::mul(g_pow_n, gl::pow(v, j)), 1));
        if (e == 0) P2_FAIL(ctx, P2HOT_EINVAL, "quotient_polys: Z_H vanishes on the coset");
        small[num_routed_ + j] = e;
        small[num_routed_ + rate + j] = gl::inv(e);
    }
    // 1 / (n (x - 1)) for every point of the quotient coset: sizes only, kept by the context
    const auto inv_key = std::make_tuple(100, log_nq, qbits);
    auto inv_it = ctx->twid_cache.find(inv_key);
    if (inv_it == ctx->twid_cache.end()) {
        u64 *t = nullptr;
        P2_HIP(ctx, hipMalloc((void **)&t, m * 8));
        P2HOT_LAUNCH(plonk::quot_inv_kernel, dim3(cdiv(m, 256)), dim3(256), 0, ctx->stream, t, log_nq, (u64)n % gl::P, ctx->fwd);
        if (hipGetLastError() != hipSuccess) {
            (void)hipFree(t);
            P2_FAIL(ctx, P2HOT_EHIP, "quotient_polys: the L_0 denominator table could not be launched");
        }
        inv_it = ctx->twid_cache.emplace(inv_key, t).first;
    }
    plonk::QuotArgs q{};
    q.wires = wires->d_lde, q.wires_stride = wires->col_stride_lde();
    q.sigmas = constants_sigmas->d_lde + sigmas_first_col * constants_sigmas->col_stride_lde(), q.sigmas_stride = constants_sigmas->col_stride_lde();
    q.zs = zs_partial_products->d_lde, q.zs_stride = zs_partial_products->col_stride_lde();
    q.bk = d_small.u(), q.zh = d_small.u() + nbk, q.inv_nx1 = inv_it->second;
    q.gate_sums = gate_sums ? d_gate.u() : nullptr;
    q.out = d_work.u();
    q.num_routed = num_routed, q.degree = quotient_degree_factor, q.num_chunks = num_chunks, q.log_nq = log_nq, q.qbits = qbits;
    const u64 K = (u64)num_challenges + (u64)num_challenges * num_chunks;
    for (unsigned a = 0; a < num_challenges; ++a) {
        q.betas[a] = betas[a], q.gammas[a] = gl::canon(gammas[a]), q.alphas[a] = alphas[a];
        q.alpha_k[a] = gl::pow(alphas[a], K);
        for (unsigned c = 0; c < num_challenges; ++c) q.base[a][c] = gl::pow(alphas[a], (u64)num_challenges + (u64)c * num_chunks);
    }
    q.roots = ctx->fwd;
    auto body = [&]() -> int {
        P2_HIP(ctx, hipMemcpyAsync(d_small.p, small.data(), small.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        for (unsigned c = 0; c < num_challenges && gate_sums; ++c)
            P2_HIP(ctx, hipMemcpyAsync(d_gate.u() + (size_t)c * m, gate_sums[c], m * 8, hipMemcpyHostToDevice, ctx->stream));
        ProfScope prof(ctx, "quotient_perm");
        const dim3 grid(cdiv(m, 256)), block(256);
        if (num_challenges == 2 && quotient_degree_factor == 8) {  // every plonky2 config (circuit_data.rs:101-119): the pipelined instantiation
            P2HOT_LAUNCH((plonk::quotient_perm_kernel<2, 8>), grid, block, 0, ctx->stream, q);
        } else {
            switch (num_challenges) {
                case 1: P2HOT_LAUNCH((plonk::quotient_perm_kernel<1, 0>), grid, block, 0, ctx->stream, q); break;
                case 2: P2HOT_LAUNCH((plonk::quotient_perm_kernel<2, 0>), grid, block, 0, ctx->stream, q); break;
                case 3: P2HOT_LAUNCH((plonk::quotient_perm_kernel<3, 0>), grid, block, 0, ctx->stream, q); break;
                default: P2HOT_LAUNCH((plonk::quotient_perm_kernel<4, 0>), grid, block, 0, ctx->stream, q); break;
            }
        }
        P2_LAUNCH_CHECK(ctx);
        if (values_out) P2_HIP(ctx, hipMemcpyAsync(values_out, d_work.p, (size_t)num_challenges * m * 8, hipMemcpyDeviceToHost, ctx->stream));
        return P2HOT_OK;
    };
    int rc = body();
    if (rc != P2HOT_OK || !chunks_out) return sync_checked(ctx, rc, "quotient_polys");
    return quotient_chunks_core(ctx, d_work, num_challenges, degree_bits, qbits, quotient_degree_factor, "quotient_polys", chunks_out);
}
