// ubench.hip -- VALU integer instruction-rate probe for gfx950 (tooling, not product).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run on the GPU box.
// Every probe issues 8 independent copies of one instruction per loop iteration from a single
// asm statement (so the compiler can neither fold nor pad them); 8 waves/SIMD resident.
// Output: cycles per wave64 instruction per SIMD (2.0 = full rate on a SIMD-32) "by wall at an ASSUMED 2.4 GHz" -- the chip
// clocks to its power budget, so that column is only comparable within one run.  The yardstick the bench uses comes from
// tools/ubench_pmc.sh: the same probes under `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU`, i.e. TRUE
// shader cycles per instruction and the clock each probe ran at (tools/ubench_summarize.py -> profiles/r04_ubench.json).
// usage: ubench [--waves N] [--only substring] [--json]   (N = resident waves per SIMD: 8 default, 4 = the leaf sponge's)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define ITERS 65536

#define R8(INS)                                                                                    \
    INS(0) "\n\t" INS(1) "\n\t" INS(2) "\n\t" INS(3) "\n\t" INS(4) "\n\t" INS(5) "\n\t" INS(6) "\n\t" INS(7)

// operand map: %0..%7 = 32-bit accumulators b[i] (+v); %8..%15 = 64-bit accumulators a[i] (+v); %16 = c (v), %17 = d (v)
#define OPS                                                                                         \
    : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), \
      "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])  \
    : "v"(c), "v"(d), "v"(e)                                                                         \
    : "vcc", "s20", "s21", "s22", "s23"

#define I_ADD(i) "v_add_u32 %" #i ", %" #i ", %16"
#define I_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %16"
#define I_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %16"
#define I_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", 41, %16"
#define I_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", 41"
#define I_MOV(i) "v_mov_b32 %" #i ", %16"
#define I_XOR(i) "v_xor_b32 %" #i ", %" #i ", %16"
#define I_ADDE64(i) "v_add_co_u32_e64 %" #i ", s[20:21], %" #i ", %16"
#define I_LSHL64(i) "v_lshlrev_b64 %" #i ", 3, %" #i
#define I_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %16, %17"
#define I_DOT2(i) "v_dot2_u32_u16 %" #i ", %16, %17, %" #i
#define I_DOT4(i) "v_dot4_u32_u8 %" #i ", %16, %17, %" #i
#define I_PERM(i) "v_perm_b32 %" #i ", %" #i ", %16, %17"
#define I_CNDM(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc"
#define I_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %16"
#define I_ADDC(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %16, vcc"
#define I_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %16, 22"
#define I_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 5, 21"
#define I_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 16, %16"
#define I_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %16, %17"
// 64-bit accumulators are operands %8..%15
#define J(i) "%" #i
#define I_MAD64(i) "v_mad_u64_u32 %" #i ", vcc, %16, %17, %" #i
#define I_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 3, %" #i
#define I_CMP64(i) "v_cmp_lt_u64 vcc, %" #i ", %" #i


// ---- round-3 additions: the instructions a carry-free limb NTT and the Poseidon alternatives are built from ----
#define I_SUB(i) "v_sub_u32 %" #i ", %" #i ", %16"
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %16"
#define I_LSHR(i) "v_lshrrev_b32 %" #i ", 8, %" #i
#define I_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i
#define I_SUBCO(i) "v_sub_co_u32 %" #i ", vcc, %" #i ", %16"
#define I_SUBB(i) "v_subb_co_u32 %" #i ", vcc, %" #i ", %16, vcc"
#define I_ADDCE64(i) "v_addc_co_u32_e64 %" #i ", s[20:21], %" #i ", %16, s[20:21]"
#define I_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %16, %17"
#define I_ADDF32(i) "v_add_f32 %" #i ", %" #i ", %16"
#define I_PKADD16(i) "v_pk_add_u16 %" #i ", %" #i ", %16"
#define I_LSHLADD32(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %16"
#define I_MAD64S(i) "v_mad_u64_u32 %" #i ", vcc, %16, s20, %" #i
#define I_MAD64I(i) "v_mad_u64_u32 %" #i ", vcc, %16, 41, %" #i
#define I_MADI64(i) "v_mad_i64_i32 %" #i ", vcc, %16, %17, %" #i
#define I_MAD64E(i) "v_mad_u64_u32 %" #i ", s[20:21], %16, %17, %" #i
#define I_ADDF64(i) "v_add_f64 %" #i ", %" #i ", %18"
#define I_MULF64(i) "v_mul_f64 %" #i ", %" #i ", %18"
#define I_FMAF64(i) "v_fma_f64 %" #i ", %" #i ", %18, %18"
#define I_PKFMA32(i) "v_pk_fma_f32 %" #i ", %" #i ", %18, %18"
#define I_CVTF64U(i) "v_cvt_f64_u32 %" #i ", %16"
#define I_LSHR64(i) "v_lshrrev_b64 %" #i ", 8, %" #i
// 4 cheap 32-bit ops + 4 multiply-adds per iteration: do the rates add or overlap?
#define MIX_ADD_MAD                                                                                   \
    "v_add_u32 %0, %0, %16\n\tv_mad_u64_u32 %8, vcc, %16, %17, %8\n\tv_add_u32 %1, %1, %16\n\t"       \
    "v_mad_u64_u32 %9, vcc, %16, %17, %9\n\tv_add_u32 %2, %2, %16\n\tv_mad_u64_u32 %10, vcc, %16, %17, %10\n\t" \
    "v_add_u32 %3, %3, %16\n\tv_mad_u64_u32 %11, vcc, %16, %17, %11"
// a dependent chain of one accumulator (latency, 8 waves/SIMD hide it or not)
#define CHAIN_ADD "v_add_u32 %0, %0, %16\n\tv_add_u32 %0, %0, %17\n\tv_add_u32 %0, %0, %16\n\tv_add_u32 %0, %0, %17\n\t" \
                  "v_add_u32 %0, %0, %16\n\tv_add_u32 %0, %0, %17\n\tv_add_u32 %0, %0, %16\n\tv_add_u32 %0, %0, %17"
#define CHAIN_MAD "v_mad_u64_u32 %8, vcc, %16, %17, %8\n\tv_mad_u64_u32 %8, vcc, %17, %16, %8\n\tv_mad_u64_u32 %8, vcc, %16, %17, %8\n\t" \
                  "v_mad_u64_u32 %8, vcc, %17, %16, %8\n\tv_mad_u64_u32 %8, vcc, %16, %17, %8\n\tv_mad_u64_u32 %8, vcc, %17, %16, %8\n\t" \
                  "v_mad_u64_u32 %8, vcc, %16, %17, %8\n\tv_mad_u64_u32 %8, vcc, %17, %16, %8"

// the leaf sponge's instruction mix as independent streams (32 slots: 19 multiply-adds, 4 v_subb, 2 v_addc, 2 v_add_co,
// 2 v_cndmask, 1 v_mov, 2 v_add_u32): what the SIMD can issue when nothing waits on anything.  As in the kernel the multiply-adds
// leave vcc alone (their carry-out goes to a scratch SGPR pair) and a carry consumer sits two instructions behind its producer
// (generated by the snippet in the commit that added it; keep the 19 / 4 / 2 / 2 / 2 / 1 / 2 split when editing)
#define MIX_HASH \
    "v_mad_u64_u32 %8, s[22:23], %16, %17, %8\n\t" \
    "v_mad_u64_u32 %9, s[22:23], %17, %16, %9\n\t" \
    "v_add_co_u32 %0, vcc, %0, %16\n\t" \
    "v_mad_u64_u32 %10, s[22:23], %16, %16, %10\n\t" \
    "v_mad_u64_u32 %11, s[22:23], %16, %17, %11\n\t" \
    "v_addc_co_u32 %1, vcc, %1, %16, vcc\n\t" \
    "v_mad_u64_u32 %12, s[22:23], %17, %16, %12\n\t" \
    "v_mad_u64_u32 %13, s[22:23], %16, %16, %13\n\t" \
    "v_subb_co_u32 %2, vcc, %2, %17, vcc\n\t" \
    "v_mad_u64_u32 %14, s[22:23], %16, %17, %14\n\t" \
    "v_mad_u64_u32 %15, s[22:23], %17, %16, %15\n\t" \
    "v_cndmask_b32 %3, %3, %17, vcc\n\t" \
    "v_mad_u64_u32 %8, s[22:23], %16, %16, %8\n\t" \
    "v_mad_u64_u32 %9, s[22:23], %16, %17, %9\n\t" \
    "v_add_co_u32 %4, vcc, %4, %16\n\t" \
    "v_mad_u64_u32 %10, s[22:23], %17, %16, %10\n\t" \
    "v_mad_u64_u32 %11, s[22:23], %16, %16, %11\n\t" \
    "v_subb_co_u32 %5, vcc, %5, %17, vcc\n\t" \
    "v_mad_u64_u32 %12, s[22:23], %16, %17, %12\n\t" \
    "v_mad_u64_u32 %13, s[22:23], %17, %16, %13\n\t" \
    "v_addc_co_u32 %6, vcc, %6, %16, vcc\n\t" \
    "v_mad_i64_i32 %14, s[22:23], %16, %16, %14\n\t" \
    "v_mad_u64_u32 %15, s[22:23], %16, %17, %15\n\t" \
    "v_subb_co_u32 %7, vcc, %7, %17, vcc\n\t" \
    "v_mad_u64_u32 %8, s[22:23], %17, %16, %8\n\t" \
    "v_add_u32 %0, %0, %16\n\t" \
    "v_cndmask_b32 %1, %1, %17, vcc\n\t" \
    "v_mad_u64_u32 %9, s[22:23], %16, %16, %9\n\t" \
    "v_mov_b32 %2, %17\n\t" \
    "v_subb_co_u32 %3, vcc, %3, %17, vcc\n\t" \
    "v_add_u32 %4, %4, %16\n\t" \
    "v_mad_u64_u32 %10, s[22:23], %16, %17, %10"
// the limb NTT passes' mix (16 slots: 5 multiply-adds, 7 plain 32-bit add / sub / and, 2 carry adds, 1 v_alignbit, 1 v_lshl_add_u64)
#define MIX_NTT                                                                                                   \
    "v_mad_u64_u32 %8, vcc, %16, %17, %8\n\tv_add_u32 %0, %0, %16\n\tv_sub_u32 %1, %1, %16\n\tv_mad_u64_u32 %9, vcc, %16, %17, %9\n\t" \
    "v_add_u32 %2, %2, %16\n\tv_and_b32 %3, %3, %16\n\tv_mad_u64_u32 %10, vcc, %16, %17, %10\n\tv_sub_u32 %4, %4, %16\n\t"            \
    "v_add_co_u32 %5, vcc, %5, %16\n\tv_mad_u64_u32 %11, vcc, %16, %17, %11\n\tv_add_u32 %6, %6, %16\n\tv_addc_co_u32 %7, vcc, %7, %16, vcc\n\t" \
    "v_mad_u64_u32 %12, vcc, %16, %17, %12\n\tv_sub_u32 %0, %0, %17\n\tv_alignbit_b32 %1, %1, %16, 24\n\tv_lshl_add_u64 %13, %13, 3, %13"

// ---- round 4: would the full-round MDS on the matrix pipe pay?  (north_star says no MFMA -- this prices the
// question instead of arguing it.)  One full-round MDS layer today is 338 VALU instructions (288 v_mad_u64_u32 + 12 x the
// 4-instruction row fold + 2).  On byte limbs with v_mfma_i32_4x4x4_16b_i8 it is 72 MFMAs + 48 v_perm_b32 + 24 v_xor_b32 +
// ~190 recombination instructions (carry-aware: multiply-adds and carry adds).  Both as dependency-free streams, HALF a layer per
// loop trip: MIX_MDS_VALU = 144 multiply-adds + 6 folds (169 slots); MIX_MDS_MFMA = 36 MFMAs + 24 perms + 12 xors + 48 multiply-adds
// + 24 v_add_co + 23 v_addc (167 slots).  %19..%22 are four 4-register accumulators of the MFMAs.
#define MFMA4(i, j) "v_mfma_i32_4x4x4_16b_i8 %" #i ", %16, %17, %" #i "\n\tv_mfma_i32_4x4x4_16b_i8 %" #j ", %17, %16, %" #j "\n\t"
#define MAD6 "v_mad_u64_u32 %8, s[22:23], %16, %17, %8\n\tv_mad_u64_u32 %9, s[22:23], %17, %16, %9\n\tv_mad_u64_u32 %10, s[22:23], %16, %16, %10\n\t" \
             "v_mad_u64_u32 %11, s[22:23], %16, %17, %11\n\tv_mad_u64_u32 %12, s[22:23], %17, %16, %12\n\tv_mad_u64_u32 %13, s[22:23], %16, %16, %13\n\t"
#define FOLD1 "v_mad_u64_u32 %14, s[22:23], %16, -1, %14\n\tv_add_co_u32 %0, vcc, %0, %16\n\tv_mov_b32 %2, %17\n\tv_cndmask_b32 %1, %1, %17, vcc\n\tv_mad_u64_u32 %15, s[22:23], %17, 1, %15\n\t"
#define MAD24 MAD6 MAD6 MAD6 MAD6
#define MIX_MDS_VALU MAD24 FOLD1 MAD24 FOLD1 MAD24 FOLD1 MAD24 FOLD1 MAD24 FOLD1 MAD24 "v_mad_u64_u32 %14, s[22:23], %16, -1, %14\n\tv_add_co_u32 %0, vcc, %0, %16\n\tv_cndmask_b32 %1, %1, %17, vcc\n\tv_mad_u64_u32 %15, s[22:23], %17, 1, %15"
#define PERM4 "v_perm_b32 %0, %0, %16, %17\n\tv_perm_b32 %1, %1, %17, %16\n\tv_perm_b32 %2, %2, %16, %17\n\tv_perm_b32 %3, %3, %17, %16\n\t"
#define XOR2 "v_xor_b32 %4, %4, %16\n\tv_xor_b32 %5, %5, %17\n\t"
#define CARRY4 "v_add_co_u32 %6, vcc, %6, %16\n\tv_mad_u64_u32 %8, s[22:23], %16, %17, %8\n\tv_addc_co_u32 %7, vcc, %7, %16, vcc\n\tv_mad_u64_u32 %9, s[22:23], %17, %16, %9\n\t"
// one sixth of the half layer: 6 MFMAs, 4 perms, 2 xors, 8 multiply-adds, 4 add_co, 4 addc (28 slots; x6 = 168, minus one v_addc)
#define MFMA_SIXTH MFMA4(19, 20) PERM4 MFMA4(21, 22) XOR2 CARRY4 MFMA4(19, 21) CARRY4 CARRY4 CARRY4
#define MIX_MDS_MFMA MFMA_SIXTH MFMA_SIXTH MFMA_SIXTH MFMA_SIXTH MFMA_SIXTH MFMA_SIXTH "v_mov_b32 %2, %17"
#define MFMA8 MFMA4(19, 20) MFMA4(21, 22) MFMA4(19, 20) MFMA4(21, 22) "v_mov_b32 %2, %17"
typedef int v4i __attribute__((ext_vector_type(4)));
#define OPS_M                                                                                       \
    : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), \
      "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])  \
    : "v"(c), "v"(d), "v"(e), "v"(m0), "v"(m1), "v"(m2), "v"(m3)                                     \
    : "vcc", "s20", "s21", "s22", "s23"

#define R8B(INS) INS(8) "\n\t" INS(9) "\n\t" INS(10) "\n\t" INS(11) "\n\t" INS(12) "\n\t" INS(13) "\n\t" INS(14) "\n\t" INS(15)

template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t *out, uint32_t seed, uint64_t *ticks) {
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    uint64_t a[8];
    uint32_t b[8];
    uint32_t c = t * 2654435761u + seed, d = (t ^ seed) * 40503u + 1;
    uint64_t e = 0x3FF0000000000001ull + t;  // a double near 1.0 (f64 probes)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (uint64_t)t * 0x9E3779B97F4A7C15ull + i + seed;
        b[i] = t * 2246822519u + i * 7 + seed;
    }
    v4i m0 = {(int)c, (int)d, 1, 2}, m1 = m0 + 1, m2 = m0 + 2, m3 = m0 + 3;  // MFMA accumulators (read-write through "v": the asm names them %19..%22)
    uint64_t t0 = __builtin_readcyclecounter();  // s_memtime: shader-clock ticks
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
        if (OP == 0) asm volatile(R8(I_ADD) OPS);
        if (OP == 1) asm volatile(R8(I_MULLO) OPS);
        if (OP == 2) asm volatile(R8(I_MULHI) OPS);
        if (OP == 3) asm volatile(R8(I_MAD24) OPS);
        if (OP == 4) asm volatile(R8(I_MUL24) OPS);
        if (OP == 5) asm volatile(R8(I_ADD3) OPS);
        if (OP == 6) asm volatile(R8(I_DOT2) OPS);
        if (OP == 7) asm volatile(R8(I_DOT4) OPS);
        if (OP == 8) asm volatile(R8(I_PERM) OPS);
        if (OP == 9) asm volatile(R8(I_CNDM) OPS);
        if (OP == 10) asm volatile(R8(I_ADDCO) OPS);
        if (OP == 11) asm volatile(R8(I_ADDC) OPS);
        if (OP == 12) asm volatile(R8(I_ALIGN) OPS);
        if (OP == 13) asm volatile(R8(I_BFE) OPS);
        if (OP == 14) asm volatile(R8(I_LSHLOR) OPS);
        if (OP == 15) asm volatile(R8(I_ANDOR) OPS);
        if (OP == 16) asm volatile(R8B(I_MAD64) OPS);
        if (OP == 17) asm volatile(R8B(I_LSHLADD64) OPS);
        if (OP == 18) asm volatile(R8B(I_CMP64) OPS);
        if (OP == 19) asm volatile(R8(I_MOV) OPS);
        if (OP == 20) asm volatile(R8(I_XOR) OPS);
        if (OP == 21) asm volatile(R8(I_ADDE64) OPS, "s20", "s21");
        if (OP == 22) asm volatile(R8B(I_LSHL64) OPS);
        if (OP == 23) asm volatile(R8(I_SUB) OPS);
        if (OP == 24) asm volatile(R8(I_AND) OPS);
        if (OP == 25) asm volatile(R8(I_LSHR) OPS);
        if (OP == 26) asm volatile(R8(I_LSHL) OPS);
        if (OP == 27) asm volatile(R8(I_SUBCO) OPS);
        if (OP == 28) asm volatile(R8(I_SUBB) OPS);
        if (OP == 29) asm volatile(R8(I_ADDCE64) OPS);
        if (OP == 30) asm volatile(R8(I_FMA32) OPS);
        if (OP == 31) asm volatile(R8(I_ADDF32) OPS);
        if (OP == 32) asm volatile(R8(I_PKADD16) OPS);
        if (OP == 33) asm volatile(R8(I_LSHLADD32) OPS);
        if (OP == 34) asm volatile(R8B(I_MAD64S) OPS);
        if (OP == 36) asm volatile(R8B(I_MAD64I) OPS);
        if (OP == 37) asm volatile(R8B(I_MADI64) OPS);
        if (OP == 38) asm volatile(R8B(I_MAD64E) OPS);
        if (OP == 39) asm volatile(R8B(I_ADDF64) OPS);
        if (OP == 40) asm volatile(R8B(I_MULF64) OPS);
        if (OP == 41) asm volatile(R8B(I_FMAF64) OPS);
        if (OP == 42) asm volatile(R8B(I_PKFMA32) OPS);
        if (OP == 43) asm volatile(R8B(I_CVTF64U) OPS);
        if (OP == 44) asm volatile(R8B(I_LSHR64) OPS);
        if (OP == 45) asm volatile(MIX_ADD_MAD OPS);
        if (OP == 46) asm volatile(CHAIN_ADD OPS);
        if (OP == 47) asm volatile(CHAIN_MAD OPS);
        if (OP == 48) asm volatile(MIX_HASH OPS);
        if (OP == 49) asm volatile(MIX_NTT OPS);
        if (OP == 50) asm volatile(MIX_HASH "\n\t" MIX_HASH "\n\t" MIX_HASH "\n\t" MIX_HASH OPS);  // 128 slots per loop trip: the loop's own SALU is < 3 %
        if (OP == 51) asm volatile(MFMA8 OPS_M);
        if (OP == 52) asm volatile(MIX_MDS_VALU OPS_M);
        if (OP == 53) asm volatile(MIX_MDS_MFMA OPS_M);
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + b[i];
    s += (uint64_t)(m0.x + m1.y + m2.z + m3.w);
    out[t] = s;
    if ((threadIdx.x & 63) == 0) ticks[t >> 6] = t1 - t0;
}

static int g_waves = 8;        // resident waves per SIMD (blocks of 256 threads per CU)
static const char *g_only = nullptr;
static bool g_json = false;

template <int OP>
void run(const char *name, int per_iter = 8) {
    if (g_only && !strstr(name, g_only)) return;
    const int blocks = 256 * g_waves, threads = 256;  // g_waves blocks/CU -> g_waves waves/SIMD
    uint64_t *d, *tk;
    (void)hipMalloc(&d, (size_t)blocks * threads * 8);
    (void)hipMalloc(&tk, (size_t)blocks * threads / 64 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(d, 1, tk);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k<OP><<<blocks, threads>>>(d, 2 + rep, tk);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double waves_per_simd = (double)blocks * threads / 64 / 1024;
    double inst_per_simd = waves_per_simd * ITERS * per_iter;
    double cycles = best * 1e-3 * 2.4e9;
    static uint64_t host_ticks[256 * 8 * 4];
    (void)hipMemcpy(host_ticks, tk, (size_t)blocks * threads / 64 * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < blocks * threads / 64; ++i) avg += (double)host_ticks[i];
    avg /= (blocks * threads / 64);
    if (g_json)
        printf("{\"op\": %d, \"name\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_insts_per_simd\": %.0f, \"per_iter\": %d}\n", OP, name,
               g_waves, best, inst_per_simd, per_iter);
    else
        printf("%-18s %7.3f ms  %5.2f cyc/inst/SIMD by wall@2.4GHz(assumed) | %5.2f ticks/inst by s_memtime (ticks/wave %.0f)\n", name, best,
               cycles / inst_per_simd, avg / (ITERS * (double)per_iter * waves_per_simd), avg);
    (void)hipFree(tk);
    (void)hipFree(d);
}

int main(int argc, char **argv) {
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--waves") && i + 1 < argc) g_waves = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--only") && i + 1 < argc) g_only = argv[++i];
        else if (!strcmp(argv[i], "--json")) g_json = true;
    }
    if (g_waves < 1 || g_waves > 8) g_waves = 8;
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    if (!g_json) printf("device %s  CUs %d  clock %d kHz  waves/SIMD %d\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, g_waves);
    run<0>("v_add_u32");
    run<19>("v_mov_b32");
    run<20>("v_xor_b32");
    run<21>("v_add_co_u32_e64");
    run<22>("v_lshlrev_b64");
    run<5>("v_add3_u32");
    run<10>("v_add_co_u32");
    run<11>("v_addc_co_u32");
    run<9>("v_cndmask_b32");
    run<1>("v_mul_lo_u32");
    run<2>("v_mul_hi_u32");
    run<16>("v_mad_u64_u32");
    run<3>("v_mad_u32_u24");
    run<4>("v_mul_u32_u24");
    run<6>("v_dot2_u32_u16");
    run<7>("v_dot4_u32_u8");
    run<8>("v_perm_b32");
    run<12>("v_alignbit_b32");
    run<13>("v_bfe_u32");
    run<14>("v_lshl_or_b32");
    run<15>("v_and_or_b32");
    run<17>("v_lshl_add_u64");
    run<18>("v_cmp_lt_u64");
    run<23>("v_sub_u32");
    run<24>("v_and_b32");
    run<25>("v_lshrrev_b32");
    run<26>("v_lshlrev_b32");
    run<33>("v_lshl_add_u32");
    run<27>("v_sub_co_u32");
    run<28>("v_subb_co_u32");
    run<29>("v_addc_co_u32_e64");
    run<30>("v_fma_f32");
    run<31>("v_add_f32");
    run<42>("v_pk_fma_f32");
    run<32>("v_pk_add_u16");
    run<34>("v_mad_u64_u32 sgpr");
    run<36>("v_mad_u64_u32 inl");
    run<38>("v_mad_u64_u32 sdst");
    run<37>("v_mad_i64_i32");
    run<39>("v_add_f64");
    run<40>("v_mul_f64");
    run<41>("v_fma_f64");
    run<43>("v_cvt_f64_u32");
    run<44>("v_lshrrev_b64");
    run<45>("mix 4add+4mad");
    run<46>("chain v_add_u32");
    run<47>("chain v_mad_u64");
    run<48>("mix hash_leaves", 32);
    run<50>("mix hash_leaves x4", 128);
    run<51>("v_mfma_i32_4x4x4_16b_i8", 9);   // 8 MFMAs + 1 v_mov per trip
    run<52>("mix mds valu (half layer)", 173);
    run<53>("mix mds mfma (half layer)", 169);
    run<49>("mix limb_ntt", 16);
    return 0;
}
