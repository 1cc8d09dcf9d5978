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
// a 232-byte device object advanced by a single-wave kernel (one state word per lane, MDS through
// LDS) -- every FRI round is enqueued without a host round trip; beta never leaves the GPU.
#pragma once
#include "poseidon.hpp"

namespace fri {
using gl::u32;
using gl::u64;

struct Challenger {  // mirrors challenger.rs:16-24
    u64 state[12];
    u64 in[8];
    u64 out[8];
    u32 n_in, n_out;
};

// One permutation spread over lanes 0..11 of a wave (x = this lane's state word).
// All 64 threads of the block must call it.
__device__ inline u64 permute_lanes(u64 x, unsigned lane, u64 *sh /* [12] */) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
#pragma unroll 1
    for (int round = 0; round < 30; ++round) {
        u64 rc = lane < 12 ? P2_POSEIDON_ALL_ROUND_CONSTANTS[12 * round + (lane < 12 ? lane : 0)] : 0;
        x = gl::add_canon(x, rc);
        const bool full = round < 4 || round >= 26;
        if (full || lane == 0) x = poseidon::sbox7(x);
        if (lane < 12) sh[lane] = x;
        __syncthreads();
        u32 a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            unsigned j = i + (lane < 12 ? lane : 0);
            if (j >= 12) j -= 12;
            u64 v = sh[j];
            u32 lo = (u32)v, hi = (u32)(v >> 32);
            a0 += C[i] * (lo & 0x3FFFFFu);
            a1 += C[i] * (((lo >> 22) | (hi << 10)) & 0x1FFFFFu);
            a2 += C[i] * (hi >> 11);
        }
        if (lane == 0) {
            u64 v = sh[0];
            u32 lo = (u32)v, hi = (u32)(v >> 32);
            a0 += 8u * (lo & 0x3FFFFFu);
            a1 += 8u * (((lo >> 22) | (hi << 10)) & 0x1FFFFFu);
            a2 += 8u * (hi >> 11);
        }
        u64 lo64 = (u64)a0 + ((u64)a1 << 22);
        u64 t = (u64)a2 << 43;
        lo64 += t;
        u64 hi = (u64)(a2 >> 21) + (u64)(lo64 < t);
        u64 e = (hi << 32) - hi;
        u64 y = lo64 + e;
        if (y < e) y += gl::EPS;
        __syncthreads();
        x = y;
    }
    return x;
}

// observe n_obs elements, then squeeze n_get challenges (popped from the back, challenger.rs:82-92).
// launch with exactly one 64-thread block.
__global__ void __launch_bounds__(64) challenger_kernel(Challenger *ch, const u64 *obs, size_t n_obs, u64 *out,
                                                       size_t n_get) {
    __shared__ u64 sh[12];
    __shared__ u64 inbuf[8];
    __shared__ u64 outbuf[8];
    const unsigned lane = threadIdx.x;
    u32 n_in = ch->n_in, n_out = ch->n_out;
    u64 x = lane < 12 ? ch->state[lane] : 0;
    if (lane < 8) {
        inbuf[lane] = ch->in[lane];
        outbuf[lane] = ch->out[lane];
    }
    __syncthreads();
    for (size_t k = 0; k < n_obs; ++k) {
        n_out = 0;
        if (lane == 0) inbuf[n_in] = gl::canon(obs[k]);
        ++n_in;
        __syncthreads();
        if (n_in == 8) {
            if (lane < 8) x = inbuf[lane];
            x = gl::canon(permute_lanes(x, lane, sh));
            if (lane < 8) outbuf[lane] = x;
            n_in = 0;
            n_out = 8;
            __syncthreads();
        }
    }
    for (size_t k = 0; k < n_get; ++k) {
        if (n_in != 0 || n_out == 0) {
            if (lane < n_in) x = inbuf[lane];
            x = gl::canon(permute_lanes(x, lane, sh));
            if (lane < 8) outbuf[lane] = x;
            n_in = 0;
            n_out = 8;
            __syncthreads();
        }
        --n_out;
        if (lane == 0) out[k] = outbuf[n_out];
    }
    __syncthreads();
    if (lane < 12) ch->state[lane] = x;
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
