// gl.hpp -- Goldilocks (P = 2^64 - 2^32 + 1) arithmetic for CDNA4 VALU, 64-bit values kept in
// register pairs.  Replaces field/src/goldilocks_field.rs:245-320 (add/sub/mul), :402-415
// (reduce128) and extension/quadratic.rs:180-194 with W = 7 (goldilocks_extensions.rs:19).
//
// Representation: any u64 is a valid representative (like the reference, results may lie in
// [P, 2^64)); canon() is applied only where bytes leave the device.  Products are built from
// 32x32->64 multiply-adds (v_mad_u64_u32) and reduced with 2^64 = 2^32 - 1, 2^96 = -1 (mod P);
// there is no 128-bit type and no division anywhere.
#pragma once
#include "platform.h"

namespace gl {
typedef uint64_t u64;
typedef uint32_t u32;

constexpr u64 P = 0xFFFFFFFF00000001ULL;
constexpr u64 EPS = 0xFFFFFFFFULL;  // 2^32 - 1 = 2^64 mod P
constexpr u64 COSET_SHIFT = 14293326489335486720ULL;  // field/src/goldilocks_field.rs:80
constexpr u64 ROOT_2_32 = 7277203076849721926ULL;     // field/src/goldilocks_field.rs:87

__host__ __device__ __forceinline__ u64 canon(u64 x) { return x >= P ? x - P : x; }

// a + b (mod P), any representatives in, any representative out.
__host__ __device__ __forceinline__ u64 add(u64 a, u64 b) {
    u64 s = a + b;
    if (s < a) {  // wrapped: 2^64 = EPS
        s += EPS;
        if (s < EPS) s += EPS;
    }
    return s;
}

// a + c with c canonical (< P): one fold is enough.
__host__ __device__ __forceinline__ u64 add_canon(u64 a, u64 c) {
    u64 s = a + c;
    if (s < a) s += EPS;
    return s;
}

__host__ __device__ __forceinline__ u64 sub(u64 a, u64 b) {
    u64 d = a - b;
    if (a < b) {  // borrowed 2^64 = EPS too much
        u64 e = d;
        d -= EPS;
        if (e < EPS) d -= EPS;
    }
    return d;
}

// 32-bit add with carry in/out (v_add_co / v_addc_co chains)
__host__ __device__ __forceinline__ u32 addc32(u32 a, u32 b, u32 cin, u32 *cout) {
#if defined(__clang__)
    return __builtin_addc(a, b, cin, cout);
#else
    u64 s = (u64)a + b + cin;
    *cout = (u32)(s >> 32);
    return (u32)s;
#endif
}

// t + (carry ? EPS : 0) on the 32-bit halves; the caller guarantees it cannot wrap again
__host__ __device__ __forceinline__ u64 fold_carry(u64 t, bool carry) {
    u32 e = carry ? 0xFFFFFFFFu : 0u, k;
    u32 lo = addc32((u32)t, e, 0u, &k);
    u32 hi = (u32)(t >> 32) + k;
    return ((u64)hi << 32) | lo;
}

__host__ __device__ __forceinline__ u64 neg(u64 a) { return sub(0, a); }

// reduce lo + 2^64 * hi (goldilocks_field.rs:402-415): lo - hi_hi + hi_lo * EPS
__host__ __device__ __forceinline__ u64 reduce128(u64 lo, u64 hi) {
    u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= EPS;
    u64 t1 = ((u64)hl << 32) - hl;
    u64 t2 = t0 + t1;
    if (t2 < t1) t2 += EPS;
    return t2;
}

// 64x64 -> 128 from four 32x32+64 multiply-adds; every partial sum fits in 64 bits.
__host__ __device__ __forceinline__ void mul128(u64 a, u64 b, u64 &lo, u64 &hi) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    u64 p10 = (u64)a1 * b0 + (u32)p01;
    u64 p11 = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    lo = (p10 << 32) | (u32)p00;
    hi = p11;
}

// a^2: three multiplies (cross term doubled).
__host__ __device__ __forceinline__ void sqr128(u64 a, u64 &lo, u64 &hi) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32);
    u64 p00 = (u64)a0 * a0;
    u64 p01 = (u64)a0 * a1;  // < 2^64 - 2^33 + 1
    u64 p11 = (u64)a1 * a1;
    // a^2 = p00 + 2*p01*2^32 + p11*2^64
    u64 mid = (p01 << 1) + (p00 >> 32);       // may wrap once
    u64 c = (p01 >> 63) + (u64)(mid < (p01 << 1));  // bits 96.. contribution (in units of 2^96 -> 2^32 in hi)
    lo = (mid << 32) | (u32)p00;
    hi = p11 + (mid >> 32) + (c << 32);
}

__host__ __device__ __forceinline__ u64 mul(u64 a, u64 b) {
    u64 lo, hi;
    mul128(a, b, lo, hi);
    return reduce128(lo, hi);
}

// a * b + c (mod P), c any 64-bit word: the two halves of c ride the multiply-add chain of mul128 (p00 + c.lo and
// p01 + (p00 >> 32) + c.hi still fit in 64 bits), so the addition costs two instructions instead of a modular add
__host__ __device__ __forceinline__ u64 mul_add(u64 a, u64 b, u64 c) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0 + (u32)c;                          // <= (2^32-1)^2 + 2^32 - 1
    u64 p01 = (u64)a0 * b1 + ((p00 >> 32) + (c >> 32));      // <= (2^32-1)^2 + 2 (2^32-1) = 2^64 - 1
    u64 p10 = (u64)a1 * b0 + (u32)p01;
    u64 p11 = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    return reduce128((p10 << 32) | (u32)p00, p11);
}

__host__ __device__ __forceinline__ u64 sqr(u64 a) {
    u64 lo, hi;
    sqr128(a, lo, hi);
    return reduce128(lo, hi);
}

__host__ __device__ inline u64 pow(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = mul(r, a);
        a = sqr(a);
        e >>= 1;
    }
    return r;
}

// a^(P-2).  P - 2 = (2^31 - 1) * 2^33 + (2^32 - 1): an addition chain of 64 squarings and 9 multiplications through
// a^(2^k - 1), k = 2, 3, 6, 12, 24, 30, 31, 32 (square-and-multiply needs 63 + 62); inv(0) = 0 like pow(0, P - 2)
__host__ __device__ inline u64 inv(u64 a) {
    auto sqn = [](u64 v, int k) {
        for (int t = 0; t < k; ++t) v = sqr(v);
        return v;
    };
    const u64 t2 = mul(sqr(a), a), t3 = mul(sqr(t2), a), t6 = mul(sqn(t3, 3), t3), t12 = mul(sqn(t6, 6), t6), t24 = mul(sqn(t12, 12), t12);
    const u64 t31 = mul(sqr(mul(sqn(t24, 6), t6)), a);  // a^(2^31 - 1)
    const u64 t32 = mul(sqr(t31), a);                   // a^(2^32 - 1)
    return mul(sqn(t31, 33), t32);
}

// primitive 2^log_n-th root of unity (field/src/types.rs:268-272)
__host__ __device__ inline u64 root_of_unity(unsigned log_n) {
    u64 w = ROOT_2_32;
    for (unsigned i = log_n; i < 32; ++i) w = sqr(w);
    return w;
}

// F_p^2 = F_p[X]/(X^2 - 7)
struct ext2 {
    u64 a0, a1;
};

// a * c for a 32-bit constant c: two multiply-adds, hi word < 2^32 so the reduction is one fold
__host__ __device__ __forceinline__ u64 mul_small(u64 a, u32 c) {
    u64 p0 = (u64)(u32)a * c;
    u64 p1 = (u64)(u32)(a >> 32) * c + (p0 >> 32);
    u64 lo = (p1 << 32) | (u32)p0;
    u64 hl = p1 >> 32;
    u64 t1 = (hl << 32) - hl;
    u64 t2 = lo + t1;
    if (t2 < t1) t2 += EPS;
    return t2;
}

__host__ __device__ __forceinline__ ext2 ext_mul(ext2 x, ext2 y) {
    u64 t7 = mul_small(mul(x.a1, y.a1), 7);
    ext2 r;
    r.a0 = add(mul(x.a0, y.a0), t7);
    r.a1 = add(mul(x.a0, y.a1), mul(x.a1, y.a0));
    return r;
}

__host__ __device__ __forceinline__ ext2 ext_add(ext2 x, ext2 y) { return ext2{add(x.a0, y.a0), add(x.a1, y.a1)}; }

}  // namespace gl
