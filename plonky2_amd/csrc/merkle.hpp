// merkle.hpp -- Poseidon leaf sponge and Merkle levels, one permutation per lane.
//
// Replaces MerkleTree::new / fill_digests_buf / fill_subtree (plonky2/src/hash/merkle_tree.rs:193-224,
// :115-149, :86-113) with level-synchronous grid launches that write straight into the reference's
// digest layout (merkle_tree.rs:50-57): inside a cap subtree with 2^h leaves, node j of level i
// (level 0 = leaf digests) lives at digest index 2*(((j>>1) << (i+1)) + 2^i - 1) + (j&1); the
// level-h node is the cap entry.  The same closed form is what merkle_tree_prove (:151-190) walks.
//
// Leaves are read through a "reader" so the LDE matrix is consumed where it lies: column-major
// (lane L reads element (c, L): a wave reads 512 contiguous bytes per column), row-major
// (MerkleTree::new on caller-provided rows), or the FRI layout (two planes of extension components,
// leaf c = values[arity*c .. arity*(c+1)) flattened, fri/prover.rs:99-103).
#pragma once
#include "poseidon.hpp"
#include "poseidon4.hpp"
#include "poseidon16.hpp"
#include "ntt.hpp"

namespace merkle {
using gl::u32;
using gl::u64;

struct ColMajorReader {  // element (L, e) = m[e * stride + L]
    const u64 *m;
    size_t stride;
    __device__ __forceinline__ u64 operator()(size_t L, unsigned e) const { return m[(size_t)e * stride + L]; }
};
struct RowMajorReader {  // element (L, e) = m[L * W + e]
    const u64 *m;
    size_t W;
    __device__ __forceinline__ u64 operator()(size_t L, unsigned e) const { return m[L * W + e]; }
};
struct FriPlanarReader {  // element (L, e) = plane[e & 1][(L << arity_bits) + (e >> 1)]
    const u64 *p0, *p1;
    unsigned arity_bits;
    __device__ __forceinline__ u64 operator()(size_t L, unsigned e) const {
        const u64 *p = (e & 1) ? p1 : p0;
        return p[(L << arity_bits) + (e >> 1)];
    }
};

// where node j (global index over the forest of subtrees with 2^h leaves each) of `level` goes
__device__ __forceinline__ u64 *node_slot(u64 *digests, u64 *cap, unsigned h, unsigned level, size_t j) {
    if (level == h) return cap + 4 * j;
    size_t s = j >> (h - level);
    size_t jl = j & (((size_t)1 << (h - level)) - 1);
    size_t sub_digests = 2 * (((size_t)1 << h) - 1);
    size_t idx = 2 * (((jl >> 1) << (level + 1)) + ((size_t)1 << level) - 1) + (jl & 1);
    return digests + 4 * (s * sub_digests + idx);
}

// hash_or_noop of every leaf (plonk/config.rs:63-74; sponge hashing.rs:118-145: zero state,
// overwrite-mode absorb of <= 8 elements per permutation, no padding), digest -> level-0 slot.
template <class Reader>
__global__ void __launch_bounds__(256) hash_leaves_kernel(Reader rd, unsigned W, size_t leaf_offset, size_t leaf_count,
                                                         unsigned h, u64 *digests, u64 *cap) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= leaf_count) return;
    const size_t L = leaf_offset + t;  // leaf index inside the forest (reader and digest slots are forest-relative)
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = 0;
    if (W <= 4) {
        for (unsigned i = 0; i < W; ++i) s[i] = rd(L, i);
    } else {
        for (unsigned off = 0; off < W; off += 8) {
            unsigned cnt = W - off < 8 ? W - off : 8;
#pragma unroll
            for (unsigned i = 0; i < 8; ++i)
                if (i < cnt) s[i] = rd(L, off + i);
            // the next chunk overwrites words 0..next-1 (overwrite-mode sponge) and the digest is words 0..3:
            // only the word triples that survive are computed by the last MDS layer
            const unsigned next = off + 8 < W ? (W - off - 8 < 8 ? W - off - 8 : 8) : 0;
            // (a digest = triple 0 + word 3; the capacity behind a full absorb = word 8 + triple 3: four rows, not six)
            const unsigned keep = off + 8 < W ? (next == 8 ? 0x8u : (0xFu << (next / 3)) & 0xFu) : 0x1u;
            poseidon::permute(s, keep, off + 8 < W ? (next == 8 ? 8 : -1) : 3);
        }
    }
    u64 *dst = node_slot(digests, cap, h, 0, L);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = gl::canon(s[i]);
}

// The same sponge in column ranges, for the host-pointer commit whose columns arrive over PCIe block by block: one launch
// absorbs the chunks [off_begin, off_end) (multiples of 8, or off_end = W) of every leaf and parks the sponge state between
// launches in `state` (word i of leaf t at state[i * state_stride + t]; only the words the next absorb does not overwrite are
// kept, the ones the last MDS layer computed).  The launch that reaches W writes the digests.  W > 8 (a narrower leaf is one
// chunk: hash_leaves_kernel).
template <class Reader>
__global__ void __launch_bounds__(256) hash_leaves_chunks_kernel(Reader rd, unsigned W, size_t leaf_offset, size_t leaf_count,
                                                                unsigned h, u64 *digests, u64 *cap, unsigned off_begin,
                                                                unsigned off_end, u64 *state, size_t state_stride) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= leaf_count) return;
    const size_t L = leaf_offset + t;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = 0;
    if (off_begin) {  // words the chunk at off_begin overwrites (or that were never computed) stay 0 until it does
        const unsigned nxt = W - off_begin < 8 ? W - off_begin : 8;
#pragma unroll
        for (unsigned i = 0; i < 12; ++i)
            if (i >= 3 * (nxt / 3)) s[i] = state[(size_t)i * state_stride + t];
    }
    for (unsigned off = off_begin; off < off_end; off += 8) {
        unsigned cnt = W - off < 8 ? W - off : 8;
#pragma unroll
        for (unsigned i = 0; i < 8; ++i)
            if (i < cnt) s[i] = rd(L, off + i);
        const unsigned next = off + 8 < W ? (W - off - 8 < 8 ? W - off - 8 : 8) : 0;
        const unsigned keep = off + 8 < W ? (next == 8 ? 0x8u : (0xFu << (next / 3)) & 0xFu) : 0x1u;
        poseidon::permute(s, keep, off + 8 < W ? (next == 8 ? 8 : -1) : 3);
    }
    if (off_end >= W) {
        u64 *dst = node_slot(digests, cap, h, 0, L);
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = gl::canon(s[i]);
    } else {
        const unsigned nxt = W - off_end < 8 ? W - off_end : 8;
#pragma unroll
        for (unsigned i = 0; i < 12; ++i)
            if (i >= 3 * (nxt / 3)) state[(size_t)i * state_stride + t] = s[i];
    }
}

// one tree level: node j = two_to_one(children 2j, 2j+1 of level-1) (merkle_tree.rs:108-112)
__global__ void __launch_bounds__(256) merkle_level_kernel(u64 *digests, u64 *cap, unsigned h, unsigned level,
                                                          size_t n_nodes) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_nodes) return;
    const u64 *ch = node_slot(digests, cap, h, level - 1, 2 * j);  // siblings are adjacent
    u64 out[4];
    poseidon::two_to_one(ch, ch + 4, out);
    u64 *dst = node_slot(digests, cap, h, level, j);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = out[i];
}

// ---- quad-cooperative variants (4 lanes per leaf / node, poseidon4.hpp): same results, ~3x lower latency;
// used when a launch has too few permutations to fill the chip.  Every thread stays alive to the end (the quad
// exchanges need all four lanes); out-of-range quads compute on zeros and store nothing.
template <class Reader>
__global__ void __launch_bounds__(256) hash_leaves_quad_kernel(Reader rd, unsigned W, size_t leaf_offset, size_t leaf_count,
                                                              unsigned h, u64 *digests, u64 *cap) {
    const size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const unsigned q = threadIdx.x & 3;
    const bool live = t < leaf_count;
    const size_t L = leaf_offset + (live ? t : 0);
    u64 w[3] = {0, 0, 0};  // state words 3q, 3q+1, 3q+2
    if (W <= 4) {          // hash_or_noop: copy (plonk/config.rs:63-74)
#pragma unroll
        for (unsigned u = 0; u < 3; ++u)
            if (live && 3 * q + u < W) w[u] = rd(L, 3 * q + u);
    } else {
        for (unsigned off = 0; off < W; off += 8) {
            const unsigned cnt = W - off < 8 ? W - off : 8;
#pragma unroll
            for (unsigned u = 0; u < 3; ++u) {
                const unsigned e = 3 * q + u;  // word e of the rate portion is overwritten by input off + e
                if (live && e < cnt) w[u] = rd(L, off + e);
            }
            poseidon4::permute_quad(w, q);
        }
    }
    if (live) {  // digest = words 0..3: lane 0 holds 0,1,2 and lane 1 holds 3
        u64 *dst = node_slot(digests, cap, h, 0, L);
        if (q == 0) {
            dst[0] = gl::canon(w[0]);
            dst[1] = gl::canon(w[1]);
            dst[2] = gl::canon(w[2]);
        } else if (q == 1) {
            dst[3] = gl::canon(w[0]);
        }
    }
}

__global__ void __launch_bounds__(256) merkle_level_quad_kernel(u64 *digests, u64 *cap, unsigned h, unsigned level,
                                                               size_t n_nodes) {
    const size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const unsigned q = threadIdx.x & 3;
    const bool live = j < n_nodes;
    const u64 *ch = node_slot(digests, cap, h, level - 1, 2 * (live ? j : 0));  // 8 contiguous words [left, right]
    u64 w[3];
#pragma unroll
    for (unsigned u = 0; u < 3; ++u) w[u] = (live && 3 * q + u < 8) ? ch[3 * q + u] : 0;  // words 8..11 = 0
    poseidon4::permute_quad(w, q);
    if (live) {
        u64 *dst = node_slot(digests, cap, h, level, j);
        if (q == 0) {
            dst[0] = gl::canon(w[0]);
            dst[1] = gl::canon(w[1]);
            dst[2] = gl::canon(w[2]);
        } else if (q == 1) {
            dst[3] = gl::canon(w[0]);
        }
    }
}

// ---- word-per-lane variants (16 lanes per leaf / node, poseidon16.hpp): the lowest latency per permutation, used for
// launches with at most a few thousand permutations.  Every thread stays alive to the end (the row broadcasts need the lanes).
template <class Reader>
__global__ void __launch_bounds__(256) hash_leaves_row_kernel(Reader rd, unsigned W, size_t leaf_offset, size_t leaf_count,
                                                             unsigned h, u64 *digests, u64 *cap) {
    const size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const unsigned r = threadIdx.x & 15;
    const bool live = t < leaf_count;
    const size_t L = leaf_offset + (live ? t : 0);
    u64 w = 0;  // state word r
    if (W <= 4) {  // hash_or_noop: copy (plonk/config.rs:63-74)
        if (live && r < W) w = rd(L, r);
    } else {
        const poseidon16::RowConsts k = poseidon16::row_consts(r);
        for (unsigned off = 0; off < W; off += 8) {
            const unsigned cnt = W - off < 8 ? W - off : 8;
            if (live && r < cnt) w = rd(L, off + r);  // overwrite-mode absorb: word r of the rate portion <- input off + r
            poseidon16::permute_row(w, r, k);
        }
    }
    if (live && r < 4) node_slot(digests, cap, h, 0, L)[r] = gl::canon(w);
}

__global__ void __launch_bounds__(256) merkle_level_row_kernel(u64 *digests, u64 *cap, unsigned h, unsigned level, size_t n_nodes) {
    const size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const unsigned r = threadIdx.x & 15;
    const bool live = j < n_nodes;
    const u64 *ch = node_slot(digests, cap, h, level - 1, 2 * (live ? j : 0));  // 8 contiguous words [left, right]
    u64 w = (live && r < 8) ? ch[r] : 0;                                        // words 8..11 = 0
    const poseidon16::RowConsts k = poseidon16::row_consts(r);
    poseidon16::permute_row(w, r, k);
    if (live && r < 4) node_slot(digests, cap, h, level, j)[r] = gl::canon(w);
}

// raw permutations through the row mapping (parity primitive): states [count][12]
__global__ void __launch_bounds__(256) permute_batch_row_kernel(u64 *states, size_t count) {
    const size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const unsigned r = threadIdx.x & 15;
    const bool live = t < count && r < 12;
    u64 w = live ? states[12 * t + r] : 0;
    const poseidon16::RowConsts k = poseidon16::row_consts(r);
    poseidon16::permute_row(w, r, k);
    if (live) states[12 * t + r] = gl::canon(w);
}

// batch of raw permutations, states [count][12] (parity primitive for the reference KATs,
// poseidon_goldilocks.rs:455-490)
__global__ void permute_batch_kernel(u64 *states, size_t count) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = states[12 * t + i];
    poseidon::permute(s);
#pragma unroll
    for (int i = 0; i < 12; ++i) states[12 * t + i] = gl::canon(s[i]);
}

// field-arithmetic self-test primitive: out[0][t] = a*b by the compiler-scheduled multiply, out[1][t] by the
// hand-scheduled single stream (mul1), out[2][t] by the 3-way interleaved stream (mul3, lanes grouped in
// threes of consecutive elements), out[3][t] = a+b, out[4][t] = a-b, out[5][t] = 0 or a bit per hand-written stream that
// disagreed with the compiler's arithmetic.  All canonical.  Used by the edge-value
// grid test (the reference's field/src/prime_field_testing.rs:8-17 pattern).
__global__ void field_selftest_kernel(const u64 *a, const u64 *b, size_t count, u64 *out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const u64 x = a[t], y = b[t];
    out[t] = gl::canon(gl::mul(x, y));
    out[count + t] = gl::canon(gl::mul1(x, y));
    // three different products per lane through mul3: (x,y), (y,x), (x,x) -> check the first, fold the others in
    u64 aa[3] = {x, y, x}, bb[3] = {y, x, x}, rr[3];
    gl::mul3(aa, bb, rr);
    u64 ok = (gl::canon(rr[1]) == gl::canon(rr[0])) && (gl::canon(rr[2]) == gl::canon(gl::mul(x, x))) ? 0 : 1;
    // the other hand-written streams against the compiler's arithmetic, one flag bit each:
    // 2 = the low-register single stream, 4 = the power-of-two twiddle multiplies of the radix-8 butterflies (ntt.hpp),
    // 8 / 16 = the MDS row recombinations fold1 / fold3 on accumulators below 2^63, 32 / 64 = the two-stream mul2 / fold2, 128 = gl::mul_add, 256 = gl::mad3, 512 = gl::mul3cf / mul1cf / mul3cg / mul1cg
    if (gl::canon(gl::mul1_lowregs(x, y)) != gl::canon(gl::mul(x, y))) ok |= 2;
#define P2_CHK_POW2(S)                                                                                              \
    if (gl::canon(ntt::mul_pow2_asm<S>(x)) != gl::canon(gl::mul(x, (S) < 64 ? 1ull << ((S) & 63) : 0xFFFFFFFFull << (((S) - 64) & 31)))) \
        ok |= 4;
    P2_CHK_POW2(12) P2_CHK_POW2(24) P2_CHK_POW2(36) P2_CHK_POW2(48) P2_CHK_POW2(60) P2_CHK_POW2(72) P2_CHK_POW2(84)
    P2_CHK_POW2(1) P2_CHK_POW2(31) P2_CHK_POW2(33) P2_CHK_POW2(63) P2_CHK_POW2(65) P2_CHK_POW2(95)
#undef P2_CHK_POW2
    {
        const u64 al = x >> 1, ah = y >> 1;  // al + ah * 2^32
        const u64 want = gl::canon(gl::add(al, gl::mul(ah, 1ull << 32)));
        if (gl::canon(gl::fold1(al, ah)) != want) ok |= 8;
        const u64 al3[3] = {al, ah, al ^ 0x2AAAAAAAAAAAAAAAull}, ah3[3] = {ah, al, ah};
        u64 y3[3];
        gl::fold3(al3, ah3, y3);
        for (int k = 0; k < 3; ++k)
            if (gl::canon(y3[k]) != gl::canon(gl::add(al3[k], gl::mul(ah3[k], 1ull << 32)))) ok |= 16;
    }
    {
        const u64 a2[2] = {x, y ^ 0x9E3779B97F4A7C15ull}, b2[2] = {y, x};
        u64 r2[2];
        gl::mul2(a2, b2, r2);
        if (gl::canon(r2[0]) != gl::canon(gl::mul(x, y)) || gl::canon(r2[1]) != gl::canon(gl::mul(a2[1], x))) ok |= 32;
        const u64 l2[2] = {x >> 1, y >> 1}, h2[2] = {y >> 1, x >> 1};
        u64 f2[2];
        gl::fold2(l2, h2, f2);
        for (int k = 0; k < 2; ++k)
            if (gl::canon(f2[k]) != gl::canon(gl::add(l2[k], gl::mul(h2[k], 1ull << 32)))) ok |= 64;
    }
    // 128 = the fused multiply-add (gl::mul_add: the addend rides the multiply-add chain), with addends up to 2^64 - 1
    if (gl::canon(gl::mul_add(x, y, x ^ y)) != gl::canon(gl::add(gl::mul(x, y), x ^ y)) || gl::canon(gl::mul_add(y, x, ~0ull)) != gl::canon(gl::add(gl::mul(x, y), ~0ull)) ||
        gl::canon(gl::mul_add(x, x, y)) != gl::canon(gl::add(gl::mul(x, x), y)))
        ok |= 128;
    {  // 256 = the fused multiply-add streams (gl::mad3), addends up to 2^64 - 1
        const u64 a3[3] = {x, y, x}, b3[3] = {y, x, x}, c3[3] = {x ^ y, ~0ull, y};
        u64 r3[3];
        gl::mad3(a3, b3, c3, r3);
        for (int k = 0; k < 3; ++k)
            if (gl::canon(r3[k]) != gl::canon(gl::add(gl::mul(a3[k], b3[k]), c3[k]))) ok |= 256;
    }
    {  // 512 = the carry-free multiply streams of the Poseidon S-boxes (gl::mul3cf / mul1cf and mul3cg / mul1cg: partial products chained through the addend)
        const u64 a3[3] = {x, y, x ^ 0x9E3779B97F4A7C15ull}, b3[3] = {y, y, x};
        u64 r3[3];
        gl::mul3cf(a3, b3, r3);
        for (int k = 0; k < 3; ++k)
            if (gl::canon(r3[k]) != gl::canon(gl::mul(a3[k], b3[k]))) ok |= 512;
        if (gl::canon(gl::mul1cf(x, y)) != gl::canon(gl::mul(x, y)) || gl::canon(gl::mul1cf(y, y)) != gl::canon(gl::mul(y, y))) ok |= 512;
        gl::mul3cg(a3, b3, r3);
        for (int k = 0; k < 3; ++k)
            if (gl::canon(r3[k]) != gl::canon(gl::mul(a3[k], b3[k]))) ok |= 512;
        gl::mul3ch(a3, b3, r3);
        for (int k = 0; k < 3; ++k)
            if (gl::canon(r3[k]) != gl::canon(gl::mul(a3[k], b3[k]))) ok |= 512;
        if (gl::canon(gl::mul1cg(x, y)) != gl::canon(gl::mul(x, y)) || gl::canon(gl::mul1cg(y, y)) != gl::canon(gl::mul(y, y))) ok |= 512;
    }
    out[2 * count + t] = gl::canon(rr[0]);
    out[5 * count + t] = ok;
    out[3 * count + t] = gl::canon(gl::add(x, y));
    out[4 * count + t] = gl::canon(gl::sub(x, y));
}

}  // namespace merkle
