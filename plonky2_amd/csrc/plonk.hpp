// plonk.hpp -- the permutation argument's partial products and Z polynomials on the device.
//
// Replaces wires_permutation_partial_products_and_zs (plonky2/src/plonk/prover.rs:392-449) with its helpers
// quotient_chunk_products / partial_products_and_z_gx (plonky2/src/util/partial_products.rs:13-39).  Per row i of
// the trace (x_i = w_n^i, prover_data.subgroup) and routed wire j the reference forms
//   num_j = wire + beta * k_j * x_i + gamma,   den_j = wire + beta * sigma_j(x_i) + gamma,
// multiplies num_j / den_j over chunks of `degree` wires, and walks the rows SEQUENTIALLY:
//   pp_p(x_i) = Z(x_i) * prod_{c <= p} chunk_c(i)   (p < num_prods),   Z(x_{i+1}) = Z(x_i) * prod_c chunk_c(i),  Z(x_0) = 1.
// Here the row walk is an exclusive prefix PRODUCT scan (chunk totals -> carries in one workgroup -> replay), and
// the per-row work is lane-parallel with one field inversion per row: with PN_c / PD_c the prefix products of the
// chunk numerators / denominators, prod_{k <= c} chunk_k = PN_c * PD_c^-1 and PD_{c-1}^-1 = PD_c^-1 * D_c
// (Montgomery's trick run backwards over the chunks).  Field arithmetic is exact, so the values equal the
// reference's element-wise batch_multiplicative_inverse route (field/src/types.rs:133) bit for bit once canonical.
// A zero denominator makes the reference panic ("Tried to invert zero"); here it raises a flag -> P2HOT_EINVAL.
#pragma once
#include "gl_mul3.hpp"
#include "gl.hpp"
#include "ntt.hpp"

namespace plonk {
using gl::u32;
using gl::u64;

struct PPArgs {
    const u64 *wires, *sigmas;  // [num_routed][n] column-major (element (j, i) at j * stride + i)
    size_t wires_stride, sigmas_stride;
    const u64 *k_is;            // device, [num_routed] coset shifts (common_data.k_is)
    unsigned num_routed, degree, num_chunks, log_n;
    u64 beta, gamma;
    ntt::RootTable roots;       // forward table: x_i = w_n^i
    u64 *pp;                    // [num_chunks - 1][n] (stride pp_stride): prefix quotients, later the partial products
    size_t pp_stride;
    u64 *dchunk;                // scratch [num_chunks][n]: chunk denominators
    u64 *total;                 // scratch [n]: T_i = prod over all chunks of row i
    unsigned *zero_flag;
};

// lane = row: prefix quotients Q_c(i) = prod_{k <= c} chunk_k(i) for c < num_chunks - 1, and T_i = Q_last(i)
__global__ void __launch_bounds__(256) pp_quotients_kernel(PPArgs a) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)1 << a.log_n;
    if (i >= n) return;
    const u64 x = a.log_n ? ntt::root_pow(a.roots, (u32)(i << (32 - a.log_n))) : 1;
    const u64 bx = gl::mul(a.beta, x);
    u64 pn = 1, pd = 1;
    for (unsigned c = 0; c < a.num_chunks; ++c) {
        u64 d = 1;
        const unsigned j_end = (c + 1) * a.degree < a.num_routed ? (c + 1) * a.degree : a.num_routed;
        for (unsigned j = c * a.degree; j < j_end; ++j) {
            const u64 w = a.wires[(size_t)j * a.wires_stride + i];
            const u64 num = gl::add(gl::add(w, gl::mul(bx, a.k_is[j])), a.gamma);
            const u64 den = gl::add(gl::add(w, gl::mul(a.beta, a.sigmas[(size_t)j * a.sigmas_stride + i])), a.gamma);
            pn = gl::mul(pn, num);
            d = gl::mul(d, den);
        }
        pd = gl::mul(pd, d);
        a.dchunk[(size_t)c * n + i] = d;
        if (c + 1 < a.num_chunks) a.pp[(size_t)c * a.pp_stride + i] = pn;
    }
    if (gl::canon(pd) == 0) atomicOr(a.zero_flag, 1u);
    u64 inv = gl::inv(pd);  // PD_last^-1
    a.total[i] = gl::mul(pn, inv);
    for (unsigned c = a.num_chunks; c-- > 0;) {
        if (c + 1 < a.num_chunks) {
            u64 *q = a.pp + (size_t)c * a.pp_stride + i;
            *q = gl::mul(*q, inv);
        }
        inv = gl::mul(inv, a.dchunk[(size_t)c * n + i]);
    }
}

// Both challenges of a plonky2 config in ONE pass over the wires and sigmas, all multiplications as gl_mul3.hpp streams (the
// additions ride the multiply-adds), the two Fermat inversions as two interleaved streams: the same results as two
// pp_quotients_kernel launches at less than half their instructions (the compiler's 64-bit multiply-reduce costs ~34 here).
struct PPArgs2 {
    PPArgs c[2];  // wires / sigmas / k_is / sizes / roots are taken from c[0]
};
__global__ void __launch_bounds__(256) pp_quotients2_kernel(PPArgs2 aa) {
    const PPArgs &a = aa.c[0], &b = aa.c[1];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)1 << a.log_n;
    if (i >= n) return;
    const u64 x = a.log_n ? ntt::root_pow(a.roots, (u32)(i << (32 - a.log_n))) : 1;
    const u64 beta[2] = {a.beta, b.beta}, gamma[2] = {a.gamma, b.gamma};  // canonical (the host canonicalises)
    u64 bx[2];
    {
        const u64 xx[2] = {x, x};
        gl::mul2(beta, xx, bx);
    }
    u64 pn[2] = {1, 1}, pd[2] = {1, 1};
    for (unsigned c = 0; c < a.num_chunks; ++c) {
        u64 d[2] = {1, 1};
        const unsigned j_end = (c + 1) * a.degree < a.num_routed ? (c + 1) * a.degree : a.num_routed;
        for (unsigned j = c * a.degree; j < j_end; ++j) {
            const u64 w = a.wires[(size_t)j * a.wires_stride + i], sg = a.sigmas[(size_t)j * a.sigmas_stride + i], kj = a.k_is[j];
            const u64 wg0 = gl::add_canon(w, gamma[0]), wg1 = gl::add_canon(w, gamma[1]);
            const u64 a1[3] = {bx[0], bx[1], beta[0]}, b1[3] = {kj, kj, sg}, c1[3] = {wg0, wg1, wg0};
            u64 r1[3];
            gl::mad3(a1, b1, c1, r1);  // num0, num1, den0
            const u64 a2[3] = {beta[1], pn[0], pn[1]}, b2[3] = {sg, r1[0], r1[1]}, c2[3] = {wg1, 0, 0};
            u64 r2[3];
            gl::mad3(a2, b2, c2, r2);  // den1, pn0 num0, pn1 num1
            pn[0] = r2[1], pn[1] = r2[2];
            const u64 b3[2] = {r1[2], r2[0]};
            u64 r3[2];
            gl::mul2(d, b3, r3);
            d[0] = r3[0], d[1] = r3[1];
        }
        gl::mul2(pd, d, pd);
        a.dchunk[(size_t)c * n + i] = d[0];
        b.dchunk[(size_t)c * n + i] = d[1];
        if (c + 1 < a.num_chunks) {
            a.pp[(size_t)c * a.pp_stride + i] = pn[0];
            b.pp[(size_t)c * b.pp_stride + i] = pn[1];
        }
    }
    if (gl::canon(pd[0]) == 0 || gl::canon(pd[1]) == 0) atomicOr(a.zero_flag, 1u);
    // pd^(P-2) for both challenges at once.  P - 2 = (2^31 - 1) * 2^33 + (2^32 - 1): an addition chain of 64 squarings and 9
    // multiplications (x^(2^k - 1) for k = 2, 3, 6, 12, 24, 30, 31, 32) instead of square-and-multiply's 63 + 62
    u64 inv[2];
    {
        auto sqn = [](u64 v[2], int k) {
            for (int t = 0; t < k; ++t) gl::mul2(v, v, v);
        };
        u64 t2[2] = {pd[0], pd[1]}, t3[2], t6[2], t12[2], t24[2], t[2];
        sqn(t2, 1), gl::mul2(t2, pd, t2);                                   // 2^2 - 1
        t3[0] = t2[0], t3[1] = t2[1], sqn(t3, 1), gl::mul2(t3, pd, t3);     // 2^3 - 1
        t6[0] = t3[0], t6[1] = t3[1], sqn(t6, 3), gl::mul2(t6, t3, t6);     // 2^6 - 1
        t12[0] = t6[0], t12[1] = t6[1], sqn(t12, 6), gl::mul2(t12, t6, t12);
        t24[0] = t12[0], t24[1] = t12[1], sqn(t24, 12), gl::mul2(t24, t12, t24);
        t[0] = t24[0], t[1] = t24[1], sqn(t, 6), gl::mul2(t, t6, t);        // 2^30 - 1
        sqn(t, 1), gl::mul2(t, pd, t);                                      // a = x^(2^31 - 1)
        u64 b[2] = {t[0], t[1]};
        sqn(b, 1), gl::mul2(b, pd, b);                                      // b = x^(2^32 - 1)
        sqn(t, 33);
        gl::mul2(t, b, inv);
    }
    {
        u64 t[2];
        gl::mul2(pn, inv, t);
        a.total[i] = t[0];
        b.total[i] = t[1];
    }
    for (unsigned c = a.num_chunks; c-- > 0;) {
        if (c + 1 < a.num_chunks) {
            u64 *q0 = a.pp + (size_t)c * a.pp_stride + i, *q1 = b.pp + (size_t)c * b.pp_stride + i;
            const u64 qq[2] = {*q0, *q1};
            u64 t[2];
            gl::mul2(qq, inv, t);
            *q0 = t[0], *q1 = t[1];
        }
        const u64 dd[2] = {a.dchunk[(size_t)c * n + i], b.dchunk[(size_t)c * n + i]};
        gl::mul2(inv, dd, inv);
    }
}

// lane = chunk of 2^chunk_log rows: its product
__global__ void pp_chunk_totals_kernel(const u64 *total, unsigned chunk_log, size_t n_chunks, u64 *prod) {
    const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_chunks) return;
    u64 acc = 1;
    const size_t base = m << chunk_log;
    for (size_t i = 0; i < ((size_t)1 << chunk_log); ++i) acc = gl::mul(acc, total[base + i]);
    prod[m] = acc;
}

// carry[m] = prod_{m' < m} P[m'] (exclusive prefix product); one 1024-thread block, `per` consecutive chunks per thread
__global__ void __launch_bounds__(1024) pp_carries_kernel(const u64 *prod, size_t n_chunks, size_t per, u64 *carry) {
    __shared__ u64 s[1024];
    const unsigned tid = threadIdx.x;
    const size_t lo = (size_t)tid * per, hi = lo + per < n_chunks ? lo + per : n_chunks;
    u64 loc = 1;
    for (size_t m = lo; m < hi; ++m) loc = gl::mul(loc, prod[m]);
    s[tid] = loc;
    __syncthreads();
    for (unsigned d = 1; d < 1024; d <<= 1) {  // inclusive Hillis-Steele scan
        u64 f = 1;
        if (tid >= d) f = s[tid - d];
        __syncthreads();
        s[tid] = gl::mul(s[tid], f);
        __syncthreads();
    }
    u64 c = tid ? s[tid - 1] : 1;
    for (size_t m = lo; m < hi; ++m) {
        carry[m] = c;
        c = gl::mul(c, prod[m]);
    }
}

// lane = chunk: Z(x_i) for its rows from the chunk's carry (Z(x_0) = 1)
__global__ void pp_emit_z_kernel(const u64 *total, unsigned chunk_log, size_t n_chunks, const u64 *carry, u64 *z) {
    const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_chunks) return;
    u64 acc = carry[m];
    const size_t base = m << chunk_log;
    for (size_t i = 0; i < ((size_t)1 << chunk_log); ++i) {
        z[base + i] = gl::canon(acc);
        acc = gl::mul(acc, total[base + i]);
    }
}

// lane = row: pp_p(x_i) = Z(x_i) * Q_p(i)
__global__ void pp_scale_kernel(u64 *pp, size_t pp_stride, unsigned num_prods, const u64 *z, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 zi = z[i];
    for (unsigned p = 0; p < num_prods; ++p) {
        u64 *q = pp + (size_t)p * pp_stride + i;
        *q = gl::canon(gl::mul(*q, zi));
    }
}

// trim_to_len's divisibility check (field/src/polynomial/mod.rs:164-178): raises *flag when any of v[0..count) is nonzero
__global__ void any_nonzero_kernel(const u64 *v, size_t count, unsigned *flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count && gl::canon(v[i]) != 0) atomicOr(flag, 1u);
}

// ------------------------------------------------------------------ the permutation argument's share of the quotient
// compute_quotient_polys (plonky2/src/plonk/prover.rs:609-815) evaluates, at every point x = g * w^i of the quotient coset
// (size Nq = n << qbits, qbits = log2_ceil(quotient_degree_factor)), eval_vanishing_poly_base_batch
// (plonk/vanishing_poly.rs:167-330): the terms  L_0(x) (Z_c(x) - 1)  and, per chunk of `degree` routed wires,
//   prev_acc * prod(wire_j + beta_c k_j x + gamma_c) - next_acc * prod(wire_j + beta_c sigma_j(x) + gamma_c)
// (check_partial_products, util/partial_products.rs:52-79: the accumulators run Z_c(x), the partial products, Z_c(g x)), then
// the gate constraint terms; reduces them with the powers of every alpha (plonk_common.rs:99-115) and multiplies by 1 / Z_H(x)
// (prover.rs:797-803).  Everything but the gate terms is circuit independent: this kernel computes it where the three
// commitments' LDE matrices already are.  The gate terms stay the caller's (out of scope): their own reduce_with_powers,
// `gate_sums[a][i]`, enters as alpha_a^K * gate_sums behind the K permutation terms.
// lane = row L of the LDE matrices (committed order).  The reference reads get_lde_values(i, step) = leaves[reverse_bits(i * step)]
// with step = 2^(rate_bits - qbits) (oracle.rs:142-147, prover.rs:640): those are exactly the rows L < Nq, i = bitrev_{log Nq}(L),
// so the column-major matrices are read coalesced; the "next" row is bitrev(i + 2^qbits mod Nq) (prover.rs:643, :708).
struct QuotArgs {
    const u64 *wires, *sigmas, *zs;  // column-major LDE matrices, element (col, L) at col * stride + L; sigmas points at sigma_0
    size_t wires_stride, sigmas_stride, zs_stride;
    const u64 *bk;         // device [nc][num_routed]: beta_c * k_j (wave-uniform)
    const u64 *zh;         // device [2 << qbits]: Z_H(g w^i) for i mod 2^qbits, then their inverses (field/src/zero_poly_coset.rs:21-34)
    const u64 *inv_nx1;    // device [Nq], committed order: 1 / (n (x_L - 1)), x_L = g w^bitrev(L)  (quot_inv_kernel; cached per size)
    const u64 *gate_sums;  // device [nc][Nq] natural order, or null
    u64 *out;              // device [nc][Nq] natural order (prover.rs:805-807: transpose(&quotient_values))
    unsigned num_routed, degree, num_chunks, log_nq, qbits;
    u64 betas[4], gammas[4], alphas[4];
    u64 base[4][4];        // base[a][c] = alpha_a^(nc + c * num_chunks): where challenge c's chunk terms start in the term list
    u64 alpha_k[4];        // alpha_a^K, K = nc + nc * num_chunks
    ntt::RootTable roots;  // forward table: w_Nq^i
};

// 1 / (n (x - 1)) on the quotient coset, in the committed (bit-reversed) order: the denominator of L_0 (zero_poly_coset.rs:58-61).
// Depends on the sizes only, so it is computed once per (log_nq, qbits) and kept by the context.
__global__ void __launch_bounds__(256) quot_inv_kernel(u64 *out, unsigned log_nq, u64 n_field, ntt::RootTable roots) {
    const size_t L = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (L >> log_nq) return;
    const size_t i = log_nq ? (size_t)(__brevll((unsigned long long)L) >> (64 - log_nq)) : 0;
    const u64 x = gl::mul(gl::COSET_SHIFT, log_nq ? ntt::root_pow(roots, (u32)(i << (32 - log_nq))) : (u64)1);
    out[L] = gl::canon(gl::inv(gl::mul(n_field, gl::sub(x, 1))));
}

// DEG > 0: quotient_degree_factor known at compile time (8 in every plonky2 config): the 2 * DEG loads of the NEXT chunk of routed
// wires are issued before the current chunk's 4 * NC * DEG multiplications, so the kernel streams its 13 GB instead of waiting
// for each pair of loads.  DEG == 0: any degree, plain loop.
template <int NC, int DEG>
__global__ void __launch_bounds__(256) quotient_perm_kernel(QuotArgs q) {
    const size_t L = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nq = (size_t)1 << q.log_nq;
    if (L >= nq) return;
    const size_t i = q.log_nq ? (size_t)(__brevll((unsigned long long)L) >> (64 - q.log_nq)) : 0;
    const size_t i_next = (i + ((size_t)1 << q.qbits)) & (nq - 1);
    const size_t L_next = q.log_nq ? (size_t)(__brevll((unsigned long long)i_next) >> (64 - q.log_nq)) : 0;
    const u64 x = gl::mul(gl::COSET_SHIFT, q.log_nq ? ntt::root_pow(q.roots, (u32)(i << (32 - q.log_nq))) : (u64)1);
    const size_t r = i & (((size_t)1 << q.qbits) - 1);
    // L_0(x) = Z_H(x) / (n (x - 1))  (zero_poly_coset.rs:58-61)
    const u64 l0 = gl::mul(q.zh[r], q.inv_nx1[L]);
    const unsigned num_prods = q.num_chunks - 1;
    const unsigned degree = DEG ? (unsigned)DEG : q.degree;
    u64 zx[NC], acc[NC][NC], pw[NC], res[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        zx[c] = q.zs[(size_t)c * q.zs_stride + L];
#pragma unroll
        for (int a = 0; a < NC; ++a) acc[a][c] = 0;
        pw[c] = 1;  // alpha_c^chunk
    }
    // terms 0 .. NC-1: L_0(x) (Z_c(x) - 1)
#pragma unroll
    for (int a = 0; a < NC; ++a) {
        u64 s = 0, p = 1;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            s = gl::mul_add(gl::mul(l0, gl::sub(zx[c], 1)), p, s);
            p = gl::mul(p, q.alphas[a]);
        }
        res[a] = s;
    }
    u64 prev[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) prev[c] = zx[c];
    constexpr int BUF = DEG ? DEG : 1;
    u64 wb[BUF], sb[BUF];  // the chunk in flight (DEG > 0)
    if (DEG) {
#pragma unroll
        for (int t = 0; t < BUF; ++t) {
            const unsigned j = (unsigned)t < q.num_routed ? (unsigned)t : q.num_routed - 1;
            wb[t] = q.wires[(size_t)j * q.wires_stride + L];
            sb[t] = q.sigmas[(size_t)j * q.sigmas_stride + L];
        }
    }
    for (unsigned ch = 0; ch < q.num_chunks; ++ch) {
        u64 pn[NC], pd[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) pn[c] = pd[c] = 1;
        const unsigned j0 = ch * degree, j_end = j0 + degree < q.num_routed ? j0 + degree : q.num_routed;
        // the "next" value of this chunk's check (a partial product of x, or Z(g x) for the last chunk): loaded early too
        u64 nextv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c)
            nextv[c] = ch == num_prods ? q.zs[(size_t)c * q.zs_stride + L_next] : q.zs[((size_t)NC + (size_t)c * num_prods + ch) * q.zs_stride + L];
        if (DEG) {
            u64 wc[BUF], sc[BUF];
#pragma unroll
            for (int t = 0; t < BUF; ++t) wc[t] = wb[t], sc[t] = sb[t];
            if (ch + 1 < q.num_chunks) {  // the next chunk's loads go out before this chunk's arithmetic
#pragma unroll
                for (int t = 0; t < BUF; ++t) {
                    const unsigned jn = j0 + degree + (unsigned)t, j = jn < q.num_routed ? jn : q.num_routed - 1;
                    wb[t] = q.wires[(size_t)j * q.wires_stride + L];
                    sb[t] = q.sigmas[(size_t)j * q.sigmas_stride + L];
                }
            }
            if constexpr (NC == 2) {
                // eight multiplications per routed wire as hand-scheduled streams (gl_mul3.hpp: 14 instructions each, where the
                // compiler's 64-bit code spends ~34 per multiply-reduce here): {bk0 x + wg0, bk1 x + wg1, beta0 sigma + wg0}, then
                // {beta1 sigma + wg1, pn0 n0, pn1 n1}, then {pd0 d0, pd1 d1}
#pragma unroll
                for (int t = 0; t < BUF; ++t) {
                    if (j0 + (unsigned)t < j_end) {
                        const u64 wg0 = gl::add_canon(wc[t], q.gammas[0]), wg1 = gl::add_canon(wc[t], q.gammas[1]);
                        // (the additions ride the streams' multiply-adds: gl::mad3)
                        const u64 a1[3] = {q.bk[j0 + t], q.bk[(size_t)q.num_routed + j0 + t], q.betas[0]}, b1[3] = {x, x, sc[t]}, c1[3] = {wg0, wg1, wg0};
                        u64 r1[3];
                        gl::mad3(a1, b1, c1, r1);  // n0, n1, d0
                        const u64 a2[3] = {q.betas[1], pn[0], pn[1]}, b2[3] = {sc[t], r1[0], r1[1]}, c2[3] = {wg1, 0, 0};
                        u64 r2[3];
                        gl::mad3(a2, b2, c2, r2);  // d1, pn0 n0, pn1 n1
                        pn[0] = r2[1], pn[1] = r2[2];
                        const u64 d0 = r1[2], d1 = r2[0];
                        const u64 a3[2] = {pd[0], pd[1]}, b3[2] = {d0, d1};
                        u64 r3[2];
                        gl::mul2(a3, b3, r3);
                        pd[0] = r3[0], pd[1] = r3[1];
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < BUF; ++t) {
                    if (j0 + (unsigned)t < j_end) {
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const u64 wg = gl::add(wc[t], q.gammas[c]);
                            pn[c] = gl::mul(pn[c], gl::mul_add(q.bk[(size_t)c * q.num_routed + j0 + t], x, wg));
                            pd[c] = gl::mul(pd[c], gl::mul_add(q.betas[c], sc[t], wg));
                        }
                    }
                }
            }
        } else {
            for (unsigned j = j0; j < j_end; ++j) {
                const u64 w = q.wires[(size_t)j * q.wires_stride + L], sg = q.sigmas[(size_t)j * q.sigmas_stride + L];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const u64 wg = gl::add(w, q.gammas[c]);
                    pn[c] = gl::mul(pn[c], gl::mul_add(q.bk[(size_t)c * q.num_routed + j], x, wg));
                    pd[c] = gl::mul(pd[c], gl::mul_add(q.betas[c], sg, wg));
                }
            }
        }
        if constexpr (NC == 2 && DEG != 0) {  // the chunk's check and its place in the alpha sums, as streams too
            const u64 a1[3] = {prev[0], prev[1], nextv[0]}, b1[3] = {pn[0], pn[1], pd[0]};
            u64 r1[3];
            gl::mul3(a1, b1, r1);
            const u64 a2[3] = {nextv[1], pw[0], pw[1]}, b2[3] = {pd[1], q.alphas[0], q.alphas[1]}, z3[3] = {0, 0, 0};
            u64 r2[3];
            gl::mad3(a2, b2, z3, r2);
            const u64 term0 = gl::sub(r1[0], r1[2]), term1 = gl::sub(r1[1], r2[0]);
            const u64 a3[3] = {term0, term0, term1}, b3[3] = {pw[0], pw[1], pw[0]}, c3[3] = {acc[0][0], acc[1][0], acc[0][1]};
            u64 r3[3];
            gl::mad3(a3, b3, c3, r3);
            acc[0][0] = r3[0], acc[1][0] = r3[1], acc[0][1] = r3[2];
            acc[1][1] = gl::mul_add(term1, pw[1], acc[1][1]);
            pw[0] = r2[1], pw[1] = r2[2];
            prev[0] = nextv[0], prev[1] = nextv[1];
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const u64 term = gl::sub(gl::mul(prev[c], pn[c]), gl::mul(nextv[c], pd[c]));
                prev[c] = nextv[c];
#pragma unroll
                for (int a = 0; a < NC; ++a) acc[a][c] = gl::mul_add(term, pw[a], acc[a][c]);
            }
#pragma unroll
            for (int a = 0; a < NC; ++a) pw[a] = gl::mul(pw[a], q.alphas[a]);
        }
    }
    const u64 zinv = q.zh[((size_t)1 << q.qbits) + r];
#pragma unroll
    for (int a = 0; a < NC; ++a) {
        u64 s = res[a];
#pragma unroll
        for (int c = 0; c < NC; ++c) s = gl::mul_add(acc[a][c], q.base[a][c], s);
        if (q.gate_sums) s = gl::mul_add(q.alpha_k[a], q.gate_sums[(size_t)a * nq + i], s);
        q.out[(size_t)a * nq + i] = gl::canon(gl::mul(s, zinv));
    }
}

}  // namespace plonk
