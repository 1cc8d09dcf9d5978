// fri.hpp -- FRI commit-phase kernels: coefficient fold, the device-resident Fiat-Shamir sponge,
// proof-of-work grind, extension (de)interleave.
//
// Replaces the loop body of fri_committed_trees (plonky2/src/fri/prover.rs:84-150):
//   reduce_with_powers fold (plonk/plonk_common.rs:120-132) with F^2 arithmetic
//   (field/src/extension/quadratic.rs:180-194, W = 7), Challenger::observe_cap /
//   get_extension_challenge / duplexing (plonky2/src/iop/challenger.rs:39-48, :76-92, :109-116,
//   :129-144) and fri_proof_of_work (fri/prover.rs:153-202).
//
// MI355X-first: extension polynomials live as two base-field planes so the F^2 NTT is the batch-of-2
// base NTT (extension/mod.rs:75-78: the roots of unity are base-field), and the challenger state is
// a 232-byte device object advanced by a single-wave kernel (state over one quad of lanes, MDS through
// DPP quad rotations) -- every FRI round is enqueued without a host round trip; beta never leaves the GPU.
#pragma once
#include "poseidon.hpp"
#include "poseidon4.hpp"
#include "poseidon16.hpp"

namespace fri {
using gl::u32;
using gl::u64;

struct Challenger {  // mirrors challenger.rs:16-24
    u64 state[12];
    u64 in[8];
    u64 out[8];
    u32 n_in, n_out;
};

// observe n_obs elements, then squeeze n_get challenges (popped from the back, challenger.rs:82-92).
// Launch with exactly one 64-thread block.  The sponge state lives in the first 16-lane row of the wave, one word per
// lane (poseidon16.hpp: the lowest-latency mapping -- a proof is a chain of dependent permutations here); the other
// lanes run the same instruction stream on zeros and store nothing.
__global__ void __launch_bounds__(64) challenger_kernel(Challenger *ch, const u64 *obs, size_t n_obs, u64 *out,
                                                       size_t n_get) {
    __shared__ u64 inbuf[8];
    __shared__ u64 outbuf[8];
    const unsigned lane = threadIdx.x, r = lane & 15;
    const bool owner = lane < 12;
    u32 n_in = ch->n_in, n_out = ch->n_out;
    u64 w = owner ? ch->state[lane] : 0;
    if (lane < 8) {
        inbuf[lane] = ch->in[lane];
        outbuf[lane] = ch->out[lane];
    }
    __syncthreads();
    const poseidon16::RowConsts k = poseidon16::row_consts(r);
    // duplexing (challenger.rs:129-144): overwrite the first n_in words with the buffered inputs, permute,
    // refill the output buffer with the rate portion
    auto duplex = [&]() {
        if (owner && lane < n_in) w = inbuf[lane];
        poseidon16::permute_row(w, r, k);
        w = gl::canon(w);
        if (lane < 8) outbuf[lane] = w;
        n_in = 0;
        n_out = 8;
        __syncthreads();
    };
    for (size_t i = 0; i < n_obs; ++i) {
        n_out = 0;
        if (lane == 0) inbuf[n_in] = gl::canon(obs[i]);
        ++n_in;
        __syncthreads();
        if (n_in == 8) duplex();
    }
    for (size_t i = 0; i < n_get; ++i) {
        if (n_in != 0 || n_out == 0) duplex();
        --n_out;
        if (lane == 0) out[i] = outbuf[n_out];
    }
    __syncthreads();
    if (owner) ch->state[lane] = w;
    if (lane < 8) {
        ch->in[lane] = inbuf[lane];
        ch->out[lane] = outbuf[lane];
    }
    if (lane == 0) {
        ch->n_in = n_in;
        ch->n_out = n_out;
    }
}

// coeffs'[j] = sum_{i < arity} beta^i * coeffs[arity*j + i]  (Horner from the top, plonk_common.rs:120-132)
__global__ void fold_kernel(const u64 *c0, const u64 *c1, unsigned arity_bits, const u64 *beta, size_t m_out,
                            u64 *o0, u64 *o1) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m_out) return;
    gl::ext2 b{beta[0], beta[1]};
    gl::ext2 acc{0, 0};
    const size_t base = j << arity_bits;
    for (unsigned i = 1u << arity_bits; i-- > 0;) {
        acc = gl::ext_mul(acc, b);
        acc.a0 = gl::add(acc.a0, c0[base + i]);
        acc.a1 = gl::add(acc.a1, c1[base + i]);
    }
    o0[j] = gl::canon(acc.a0);
    o1[j] = gl::canon(acc.a1);
}

// ---- prove_openings prelude: final_poly = sum_i alpha^(k_i) (F_i(X) - F_i(z_i)) / (X - z_i) ----
// ReducingFactor::reduce_polys_base (util/reducing.rs:83-95): o[t] = sum_j apow[j] * polys[j][t]
// (extension scalar times base coefficient = two base multiplies).  Lane = coefficient index, so
// every polynomial is streamed once with fully coalesced reads; apow / the pointer table are wave-uniform.
__global__ void __launch_bounds__(256) reduce_polys_base_kernel(const u64 *const *polys, size_t n_polys, const u64 *apow,
                                                               size_t n, u64 *o0, u64 *o1) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    u64 a0 = 0, a1 = 0;
    for (size_t j = 0; j < n_polys; ++j) {
        u64 p = polys[j][t];
        a0 = gl::add(a0, gl::mul(apow[2 * j], p));
        a1 = gl::add(a1, gl::mul(apow[2 * j + 1], p));
    }
    o0[t] = a0;
    o1[t] = a1;
}

// The same sum for short polynomials (recursion-size proofs: n = 2^12, ~170 polynomials): with one lane per coefficient the
// launch is 16 workgroups walking a chain of n_polys dependent multiply-adds.  Here a workgroup is 64 coefficients x 16
// polynomial groups (lane (t, g) sums the polynomials j = g mod 16) and the groups are folded through LDS in a fixed order,
// so the chain is n_polys / 16 long on 16x the lanes.  Field addition is exact: any order gives the same canonical value.
__global__ void __launch_bounds__(1024) reduce_polys_base_small_kernel(const u64 *const *polys, size_t n_polys, const u64 *apow,
                                                                      size_t n, u64 *o0, u64 *o1) {
    __shared__ u64 s0[16][64], s1[16][64];
    const unsigned lane = threadIdx.x & 63u, g = threadIdx.x >> 6;
    const size_t t = (size_t)blockIdx.x * 64 + lane;
    u64 a0 = 0, a1 = 0;
    if (t < n)
        for (size_t j = g; j < n_polys; j += 16) {
            u64 p = polys[j][t];
            a0 = gl::add(a0, gl::mul(apow[2 * j], p));
            a1 = gl::add(a1, gl::mul(apow[2 * j + 1], p));
        }
    s0[g][lane] = a0;
    s1[g][lane] = a1;
    __syncthreads();
    for (unsigned d = 8; d; d >>= 1) {
        if (g < d) {
            s0[g][lane] = gl::add(s0[g][lane], s0[g + d][lane]);
            s1[g][lane] = gl::add(s1[g][lane], s1[g + d][lane]);
        }
        __syncthreads();
    }
    if (g == 0 && t < n) {
        o0[t] = s0[0][lane];
        o1[t] = s1[0][lane];
    }
}

// divide_by_linear (field/src/polynomial/division.rs:79-92) is the Horner suffix scan
//   b_k = b_{k+1} * z + c_k,  quotient[k-1] = b_k (k >= 1), padded with a zero.
// Three steps over chunks of 2^chunk_log coefficients: (1) chunk totals, (2) chunk carries by a
// suffix scan (one workgroup), (3) replay each chunk from its carry and emit
//   acc[k] = acc[k] * shift + quotient[k]     (ReducingFactor::shift_poly + `final_poly += quotient`, oracle.rs:210-212)
__global__ void horner_chunk_totals_kernel(const u64 *c0, const u64 *c1, unsigned chunk_log, size_t n_chunks, gl::ext2 z,
                                           u64 *p0, u64 *p1) {
    size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_chunks) return;
    gl::ext2 acc{0, 0};
    const size_t base = m << chunk_log;
    for (size_t i = (size_t)1 << chunk_log; i-- > 0;) {
        acc = gl::ext_mul(acc, z);
        acc.a0 = gl::add(acc.a0, c0[base + i]);
        acc.a1 = gl::add(acc.a1, c1[base + i]);
    }
    p0[m] = acc.a0;
    p1[m] = acc.a1;
}

// carry[m] = sum_{m' > m} P[m'] * zL^(m' - m - 1)  (zL = z^(chunk length)); one 1024-thread block.
// Each thread owns `per` consecutive chunks; block-level Hillis-Steele suffix scan through LDS.
__global__ void __launch_bounds__(1024) horner_carries_kernel(const u64 *p0, const u64 *p1, size_t n_chunks, size_t per,
                                                             gl::ext2 zL, u64 *t0, u64 *t1) {
    __shared__ u64 s0[1024], s1[1024];
    const unsigned tid = threadIdx.x;
    const size_t lo = (size_t)tid * per, hi = lo + per < n_chunks ? lo + per : n_chunks;
    // local total of my chunks: L = sum_{g} P[lo + g] * zL^g ; and zL^per
    gl::ext2 loc{0, 0};
    for (size_t m = hi; m-- > lo;) {
        loc = gl::ext_mul(loc, zL);
        loc.a0 = gl::add(loc.a0, p0[m]);
        loc.a1 = gl::add(loc.a1, p1[m]);
    }
    gl::ext2 zP{1, 0};  // zL^per
    for (size_t g = 0; g < per; ++g) zP = gl::ext_mul(zP, zL);
    s0[tid] = lo < n_chunks ? loc.a0 : 0;
    s1[tid] = lo < n_chunks ? loc.a1 : 0;
    __syncthreads();
    // inclusive suffix scan: S[t] = sum_{t' >= t} loc[t'] * zP^(t' - t)
    gl::ext2 f = zP;
    for (unsigned d = 1; d < 1024; d <<= 1) {
        gl::ext2 add{0, 0};
        if (tid + d < 1024) add = gl::ext_mul(gl::ext2{s0[tid + d], s1[tid + d]}, f);
        __syncthreads();
        s0[tid] = gl::add(s0[tid], add.a0);
        s1[tid] = gl::add(s1[tid], add.a1);
        __syncthreads();
        f = gl::ext_mul(f, f);
    }
    // carry into my last chunk = S[tid + 1]; walk my chunks backwards
    gl::ext2 carry{0, 0};
    if (tid + 1 < 1024) carry = gl::ext2{s0[tid + 1], s1[tid + 1]};
    for (size_t m = hi; m-- > lo;) {
        t0[m] = carry.a0;
        t1[m] = carry.a1;
        carry = gl::ext_mul(carry, zL);
        carry.a0 = gl::add(carry.a0, p0[m]);
        carry.a1 = gl::add(carry.a1, p1[m]);
    }
}

// The chunk carries of a long polynomial in two levels (2^22 coefficients = 65536 chunks: one block walking 64 chunks per
// thread took 319 us): groups of 2^group_log chunks get their totals from horner_chunk_totals_kernel (ratio zL), the group
// carries T come from horner_carries_kernel over the groups, and this kernel walks each group down from its carry:
//   t[last chunk of g] = T[g],  t[m] = t[m + 1] * zL + P[m + 1].
__global__ void horner_group_walk_kernel(const u64 *p0, const u64 *p1, unsigned group_log, size_t n_groups, gl::ext2 zL, const u64 *T0,
                                         const u64 *T1, u64 *t0, u64 *t1) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    gl::ext2 carry{T0[g], T1[g]};
    const size_t base = g << group_log;
    for (size_t i = (size_t)1 << group_log; i-- > 0;) {
        const size_t m = base + i;
        t0[m] = carry.a0;
        t1[m] = carry.a1;
        carry = gl::ext_mul(carry, zL);
        carry.a0 = gl::add(carry.a0, p0[m]);
        carry.a1 = gl::add(carry.a1, p1[m]);
    }
}

__global__ void horner_emit_kernel(const u64 *c0, const u64 *c1, unsigned chunk_log, size_t n_chunks, gl::ext2 z,
                                   const u64 *t0, const u64 *t1, const u64 *shift_ptr, int accumulate, u64 *a0, u64 *a1) {
    size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_chunks) return;
    const gl::ext2 shift{shift_ptr[0], shift_ptr[1]};  // alpha^(#polys of the batch), device-resident (alpha never visits the host)
    gl::ext2 acc{t0[m], t1[m]};
    const size_t base = m << chunk_log, n = n_chunks << chunk_log;
    for (size_t i = (size_t)1 << chunk_log; i-- > 0;) {
        const size_t k = base + i;
        if (k == n - 1) {  // the padding coefficient of the quotient ("pad back to power of two", oracle.rs:209)
            gl::ext2 prev = accumulate ? gl::ext_mul(gl::ext2{a0[k], a1[k]}, shift) : gl::ext2{0, 0};
            a0[k] = gl::canon(prev.a0);
            a1[k] = gl::canon(prev.a1);
        }
        acc = gl::ext_mul(acc, z);
        acc.a0 = gl::add(acc.a0, c0[k]);
        acc.a1 = gl::add(acc.a1, c1[k]);
        if (k >= 1) {  // quotient[k-1] = b_k
            gl::ext2 prev = accumulate ? gl::ext_mul(gl::ext2{a0[k - 1], a1[k - 1]}, shift) : gl::ext2{0, 0};
            a0[k - 1] = gl::canon(gl::add(prev.a0, acc.a0));
            a1[k - 1] = gl::canon(gl::add(prev.a1, acc.a1));
        }
    }
}

// [n_points][total][2] (how the evaluation kernels write) -> per oracle [n_points][W_o][2], oracles back to back (how the
// caller's OpeningSet buffers are laid out), so the results go back in one copy.  Up to 8 oracles, described by value.
struct OpeningLayout {
    unsigned n_oracles;
    unsigned width[8];   // W_o
    unsigned first[8];   // index of the oracle's first polynomial among the `total`
};
__global__ void reorder_openings_kernel(const u64 *res, size_t total, size_t n_points, OpeningLayout lay, u64 *out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one (point, polynomial) pair of the device layout
    if (t >= n_points * total) return;
    const size_t p = t / total, j = t % total;
    size_t out_off = 0;
    unsigned o = 0;
    while (o + 1 < lay.n_oracles && j >= lay.first[o] + lay.width[o]) {
        out_off += 2 * n_points * lay.width[o];
        ++o;
    }
    const size_t dst = out_off + 2 * (p * lay.width[o] + (j - lay.first[o]));
    out[dst] = res[2 * t];
    out[dst + 1] = res[2 * t + 1];
}

// OpeningSet::new (plonky2/src/plonk/proof.rs:314-327): out[j] = polys[j](z) for an extension point z.
// Stage 1 (eval_polys_dot_kernel): workgroup (j, s) evaluates segment s (seg = 2^seg_log coefficients) of polynomial
//   j at z as a dot product with the table z^u, u < seg -> part[j][s].
// Stage 2: workgroup j evaluates the extension polynomial sum_s part[j][s] * (z^seg)^s: lane t Horner-folds the
//   partials t, t+256, ... with (z^seg)^256, the 256 results are combined with weights (z^seg)^t through LDS.
__device__ __forceinline__ gl::ext2 block_weighted_sum(gl::ext2 acc, gl::ext2 z, u64 *s0, u64 *s1) {
    const unsigned tid = threadIdx.x;
    gl::ext2 w{1, 0}, b = z;  // weight z^tid
    for (unsigned e = tid; e; e >>= 1) {
        if (e & 1) w = gl::ext_mul(w, b);
        b = gl::ext_mul(b, b);
    }
    acc = gl::ext_mul(acc, w);
    s0[tid] = acc.a0;
    s1[tid] = acc.a1;
    __syncthreads();
    for (unsigned d = 128; d; d >>= 1) {
        if (tid < d) {
            s0[tid] = gl::add(s0[tid], s0[tid + d]);
            s1[tid] = gl::add(s1[tid], s1[tid + d]);
        }
        __syncthreads();
    }
    return gl::ext2{s0[0], s1[0]};
}

// out[j] = alpha^j as [2] words for j < count, alpha read from device memory: base.powers() of ReducingFactor
// (util/reducing.rs:88-89) and, at index J, the shift_poly factor alpha^J (reducing.rs:103-106)
__global__ void alpha_powers_kernel(const u64 *alpha, size_t count, u64 *out) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    gl::ext2 w{1, 0}, b{gl::canon(alpha[0]), gl::canon(alpha[1])};
    for (size_t e = j; e; e >>= 1) {
        if (e & 1) w = gl::ext_mul(w, b);
        b = gl::ext_mul(b, b);
    }
    out[2 * j] = gl::canon(w.a0);
    out[2 * j + 1] = gl::canon(w.a1);
}

// fri_prover_query_rounds (fri/prover.rs:215-220, :243-253): x_index = rand % N per query, then x_index >>= arity_bits per
// round.  idx[0][q] = x_index, idx[1 + r][q] = x_index after round r's shift; rand are the challenger's outputs on the device.
struct ArityBits {
    unsigned char b[32];
};
__global__ void query_indices_kernel(const u64 *rand, size_t n_queries, unsigned log_n_total, ArityBits ab, unsigned n_rounds, u64 *idx) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_queries) return;
    u64 x = gl::canon(rand[q]) & ((((u64)1) << log_n_total) - 1);  // rand.to_canonical_u64() as usize % n, n = 2^log_n_total
    idx[q] = x;
    for (unsigned r = 0; r < n_rounds; ++r) {
        x >>= ab.b[r];
        idx[(1 + (size_t)r) * n_queries + q] = x;
    }
}

// w0[u] + X w1[u] = z^u for u < count (the power table of one segment, shared by every segment of every polynomial)
__global__ void ext_powers_kernel(gl::ext2 z, unsigned count, u64 *w0, u64 *w1) {
    const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= count) return;
    gl::ext2 w{1, 0}, b = z;
    for (unsigned e = u; e; e >>= 1) {
        if (e & 1) w = gl::ext_mul(w, b);
        b = gl::ext_mul(b, b);
    }
    w0[u] = w.a0;
    w1[u] = w.a1;
}

// Stage 1 as a dot product with the segment's power table: the coefficients are base-field elements, so a term is two
// base multiplications instead of the extension multiplication of a Horner step (about 40 instructions instead of 110).
__global__ void __launch_bounds__(256) eval_polys_dot_kernel(const u64 *const *polys, unsigned seg_log, const u64 *w0,
                                                            const u64 *w1, u64 *part /* [J][S][2] */) {
    __shared__ u64 s0[256], s1[256];
    const unsigned tid = threadIdx.x;
    const size_t seg = (size_t)1 << seg_log;
    const u64 *c = polys[blockIdx.x] + (size_t)blockIdx.y * seg;
    u64 a0 = 0, a1 = 0;
    for (size_t t = tid; t < seg; t += 256) {
        const u64 x = c[t];
        a0 = gl::add(a0, gl::mul1(x, w0[t]));
        a1 = gl::add(a1, gl::mul1(x, w1[t]));
    }
    s0[tid] = a0;
    s1[tid] = a1;
    __syncthreads();
    for (unsigned d = 128; d; d >>= 1) {
        if (tid < d) {
            s0[tid] = gl::add(s0[tid], s0[tid + d]);
            s1[tid] = gl::add(s1[tid], s1[tid + d]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        u64 *o = part + 2 * ((size_t)blockIdx.x * gridDim.y + blockIdx.y);
        o[0] = s0[0];
        o[1] = s1[0];
    }
}

__global__ void __launch_bounds__(256) eval_polys_stage2_kernel(const u64 *part, size_t n_seg, gl::ext2 zs, gl::ext2 zs256,
                                                               u64 *out /* [J][2] */) {
    __shared__ u64 s0[256], s1[256];
    const unsigned tid = threadIdx.x;
    const u64 *c = part + 2 * (size_t)blockIdx.x * n_seg;
    gl::ext2 acc{0, 0};
    if (tid < n_seg) {
        size_t last = tid + ((n_seg - 1 - tid) / 256) * 256;
        for (size_t t = last;; t -= 256) {
            acc = gl::ext_mul(acc, zs256);
            acc.a0 = gl::add(acc.a0, c[2 * t]);
            acc.a1 = gl::add(acc.a1, c[2 * t + 1]);
            if (t < 256) break;
        }
    }
    gl::ext2 r = block_weighted_sum(acc, zs, s0, s1);
    if (tid == 0) {
        out[2 * blockIdx.x] = gl::canon(r.a0);
        out[2 * blockIdx.x + 1] = gl::canon(r.a1);
    }
}

// merkle_tree_prove (hash/merkle_tree.rs:151-190) for m leaf indices straight from the device-resident
// digest array: out[q][i] = sibling at layer i.  Lane = (query, layer).
// A leaf index >= 2^log_leaves (the reference panics) yields a zero path and raises *oob.
__global__ void merkle_paths_kernel(const u64 *digests, unsigned log_leaves, unsigned cap_height, const u64 *idx,
                                    size_t m, u64 *out, unsigned *oob) {
    const unsigned layers = log_leaves - cap_height;
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (layers == 0 || e >= m * layers) return;
    const size_t q = e / layers;
    const unsigned i = (unsigned)(e % layers);
    const size_t leaf = idx[q];
    if (leaf >> log_leaves) {
#pragma unroll
        for (int w = 0; w < 4; ++w) out[4 * e + w] = 0;
        if (i == 0) atomicOr(oob, 2u);
        return;
    }
    const size_t tree_len = 2 * (((size_t)1 << layers) - 1);
    const size_t pair = (leaf & (((size_t)1 << layers) - 1)) >> i;  // pair_index before this layer's shift
    const size_t parity = pair & 1;
    const size_t siblings_index = ((pair >> 1) << (i + 1)) + ((size_t)1 << i) - 1;
    const u64 *src = digests + 4 * (tree_len * (leaf >> layers) + 2 * siblings_index + (1 - parity));
#pragma unroll
    for (int w = 0; w < 4; ++w) out[4 * e + w] = src[w];
}

// rows of a row-major matrix (the FRI round trees' leaves, MerkleTree::get, merkle_tree.rs:227): out[q][0..w) = in[idx[q]][0..w)
__global__ void gather_rowmajor_kernel(const u64 *in, size_t w, size_t n_rows, const u64 *idx, size_t m, u64 *out, unsigned *oob) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * w) return;
    const size_t q = e / w, c = e % w;
    const u64 r = idx[q];
    if (r >= n_rows) {
        out[e] = 0;
        if (c == 0) atomicOr(oob, 1u);
        return;
    }
    out[e] = gl::canon(in[r * w + c]);
}

// [count][2] <-> two planes (flatten order extension/mod.rs:128-135)
__global__ void deinterleave_kernel(const u64 *in, size_t count, u64 *p0, u64 *p1) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    p0[i] = in[2 * i];
    p1[i] = in[2 * i + 1];
}
__global__ void interleave_kernel(const u64 *p0, const u64 *p1, size_t count, u64 *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    out[2 * i] = gl::canon(p0[i]);
    out[2 * i + 1] = gl::canon(p1[i]);
}

// fri_proof_of_work (fri/prover.rs:153-202): candidate w is valid when the permutation of the
// challenger's duplex state with w in the next input slot has >= pow_bits leading zeros in word 7.
// Deterministic: atomicMin over valid candidates in [start, start + count) -> the smallest witness.
__global__ void __launch_bounds__(256) pow_kernel(const Challenger *ch, unsigned pow_bits, u64 start, u64 count,
                                                 unsigned long long *best) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    // a witness below `start` was found by an earlier chunk: the chunks are enqueued back to back without host round
    // trips, the later ones retire here (kernel boundaries order this load after the earlier chunk's atomicMin)
    if (*best < (unsigned long long)start) return;
    u64 s[12];
    const u32 n_in = ch->n_in;
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = ((u32)i < n_in && i < 8) ? ch->in[i < 8 ? i : 0] : ch->state[i];
    const u64 cand = start + t;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if ((u32)i == n_in) s[i] = cand;
    poseidon::permute(s);
    u64 resp = gl::canon(s[7]);
    unsigned lz = resp ? (unsigned)__clzll((long long)resp) : 64u;
    if (lz >= pow_bits) atomicMin(best, (unsigned long long)cand);
}

}  // namespace fri
