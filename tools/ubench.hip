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
// the leaf sponge's mix since round 6 (carry-free S-box products): 32 slots = 19 multiply-adds (one signed), 4 carry instructions, 2 selects,
// 5 v_mov_b32, 2 plain adds -- per permutation 59 % / 14 % / 6 % / 15 % / 6 % of 13.93 k instructions
#define MIX_HASH6 \
    "v_mad_u64_u32 %8, s[22:23], %16, %17, %8\n\t" \
    "v_mov_b32 %0, %16\n\t" \
    "v_mad_u64_u32 %9, s[22:23], %17, %16, %9\n\t" \
    "v_sub_co_u32 %1, vcc, %1, %16\n\t" \
    "v_mad_u64_u32 %10, s[22:23], %16, %16, %10\n\t" \
    "v_mov_b32 %2, %17\n\t" \
    "v_subb_co_u32 %3, vcc, %3, %17, vcc\n\t" \
    "v_mad_u64_u32 %11, s[22:23], %16, %17, %11\n\t" \
    "v_mad_u64_u32 %12, s[22:23], %17, %16, %12\n\t" \
    "v_cndmask_b32 %4, %4, %17, vcc\n\t" \
    "v_mad_u64_u32 %13, s[22:23], %16, %16, %13\n\t" \
    "v_mov_b32 %5, %16\n\t" \
    "v_mad_u64_u32 %14, s[22:23], %16, %17, %14\n\t" \
    "v_mad_u64_u32 %15, s[22:23], %17, %16, %15\n\t" \
    "v_add_co_u32 %6, vcc, %6, %16\n\t" \
    "v_mad_u64_u32 %8, s[22:23], %16, %16, %8\n\t" \
    "v_mov_b32 %7, %17\n\t" \
    "v_subb_co_u32 %0, vcc, %0, %17, vcc\n\t" \
    "v_mad_u64_u32 %9, s[22:23], %16, %17, %9\n\t" \
    "v_mad_u64_u32 %10, s[22:23], %17, %16, %10\n\t" \
    "v_cndmask_b32 %1, %1, %17, vcc\n\t" \
    "v_mad_i64_i32 %11, s[22:23], %16, %16, %11\n\t" \
    "v_mad_u64_u32 %12, s[22:23], %16, %17, %12\n\t" \
    "v_add_u32 %2, %2, %16\n\t" \
    "v_mad_u64_u32 %13, s[22:23], %17, %16, %13\n\t" \
    "v_mov_b32 %3, %16\n\t" \
    "v_mad_u64_u32 %14, s[22:23], %16, %16, %14\n\t" \
    "v_mad_u64_u32 %15, s[22:23], %16, %17, %15\n\t" \
    "v_add_u32 %4, %4, %17\n\t" \
    "v_mad_u64_u32 %8, s[22:23], %17, %16, %8\n\t" \
    "v_mad_u64_u32 %9, s[22:23], %16, %16, %9\n\t" \
    "v_mad_u64_u32 %10, s[22:23], %16, %17, %10"
// the leaf sponge's mix with the 14-instruction S-box products (gl::mul3cg: one chained addend less, the real carry left to the
// multiply-add's carry-out): 32 slots = 19 multiply-adds (one signed), 5 carry instructions, 2 selects, 4 v_mov_b32, 2 plain adds -- per
// permutation 60 % / 15 % / 6 % / 12 % / 6 % of 12.99 k instructions
#define MIX_HASH6B \
    "v_mad_u64_u32 %8, s[22:23], %16, %17, %8\n\t" \
    "v_mov_b32 %0, %16\n\t" \
    "v_mad_u64_u32 %9, s[22:23], %17, %16, %9\n\t" \
    "v_sub_co_u32 %1, vcc, %1, %16\n\t" \
    "v_mad_u64_u32 %10, s[22:23], %16, %16, %10\n\t" \
    "v_mov_b32 %2, %17\n\t" \
    "v_subb_co_u32 %3, vcc, %3, %17, vcc\n\t" \
    "v_mad_u64_u32 %11, s[22:23], %16, %17, %11\n\t" \
    "v_mad_u64_u32 %12, s[22:23], %17, %16, %12\n\t" \
    "v_cndmask_b32 %4, %4, %17, vcc\n\t" \
    "v_mad_u64_u32 %13, s[22:23], %16, %16, %13\n\t" \
    "v_mov_b32 %5, %16\n\t" \
    "v_mad_u64_u32 %14, s[22:23], %16, %17, %14\n\t" \
    "v_mad_u64_u32 %15, s[22:23], %17, %16, %15\n\t" \
    "v_add_co_u32 %6, vcc, %6, %16\n\t" \
    "v_mad_u64_u32 %8, s[22:23], %16, %16, %8\n\t" \
    "v_mov_b32 %7, %17\n\t" \
    "v_subb_co_u32 %0, vcc, %0, %17, vcc\n\t" \
    "v_mad_u64_u32 %9, s[22:23], %16, %17, %9\n\t" \
    "v_mad_u64_u32 %10, s[22:23], %17, %16, %10\n\t" \
    "v_cndmask_b32 %1, %1, %17, vcc\n\t" \
    "v_mad_i64_i32 %11, s[22:23], %16, %16, %11\n\t" \
    "v_mad_u64_u32 %12, s[22:23], %16, %17, %12\n\t" \
    "v_add_u32 %2, %2, %16\n\t" \
    "v_mad_u64_u32 %13, s[22:23], %17, %16, %13\n\t" \
    "v_subb_co_u32 %3, vcc, %3, %16, vcc\n\t" \
    "v_mad_u64_u32 %14, s[22:23], %16, %16, %14\n\t" \
    "v_mad_u64_u32 %15, s[22:23], %16, %17, %15\n\t" \
    "v_add_u32 %4, %4, %17\n\t" \
    "v_mad_u64_u32 %8, s[22:23], %17, %16, %8\n\t" \
    "v_mad_u64_u32 %9, s[22:23], %16, %16, %9\n\t" \
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
        if (OP == 50) asm volatile(MIX_HASH "\n\t" MIX_HASH "\n\t" MIX_HASH "\n\t" MIX_HASH OPS);
        if (OP == 54) asm volatile(MIX_HASH6 OPS);
        if (OP == 55) asm volatile(MIX_HASH6 "\n\t" MIX_HASH6 "\n\t" MIX_HASH6 "\n\t" MIX_HASH6 OPS);  // 128 slots per loop trip: the loop's own SALU is < 3 %
        if (OP == 56) asm volatile(MIX_HASH6B OPS);
        if (OP == 57) asm volatile(MIX_HASH6B "\n\t" MIX_HASH6B "\n\t" MIX_HASH6B "\n\t" MIX_HASH6B OPS);
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

// ---- round 6: what decides membership of the 2.4-cycle class?  Explicit physical registers (v16..v47 destinations, v48..v55
// sources: the VGPR bank = register number mod 4 is under control), 32 instructions per loop trip (the loop's three SALU
// instructions are < 10 % of the slots), one template instantiation per probe.  XR(M) expands M(dst) for the 32 destinations;
// XB(M) expands M(dst, src) with src in the SAME bank as dst, XD(M) with src in the NEXT bank.
#define XR(M) M(16) M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31) \
              M(32) M(33) M(34) M(35) M(36) M(37) M(38) M(39) M(40) M(41) M(42) M(43) M(44) M(45) M(46) M(47)
#define XB(M) M(16,48) M(17,49) M(18,50) M(19,51) M(20,48) M(21,49) M(22,50) M(23,51) M(24,48) M(25,49) M(26,50) M(27,51) M(28,48) M(29,49) M(30,50) M(31,51) \
              M(32,48) M(33,49) M(34,50) M(35,51) M(36,48) M(37,49) M(38,50) M(39,51) M(40,48) M(41,49) M(42,50) M(43,51) M(44,48) M(45,49) M(46,50) M(47,51)
#define XD(M) M(16,49) M(17,50) M(18,51) M(19,48) M(20,49) M(21,50) M(22,51) M(23,48) M(24,49) M(25,50) M(26,51) M(27,48) M(28,49) M(29,50) M(30,51) M(31,48) \
              M(32,49) M(33,50) M(34,51) M(35,48) M(36,49) M(37,50) M(38,51) M(39,48) M(40,49) M(41,50) M(42,51) M(43,48) M(44,49) M(45,50) M(46,51) M(47,48)
// 16 x (cheap op on an odd register pair's low word, multiply-add on an even pair): do the two classes overlap?
#define XP(M) M(16,18) M(20,22) M(24,26) M(28,30) M(32,34) M(36,38) M(40,42) M(44,46) M(17,18) M(21,22) M(25,26) M(29,30) M(33,34) M(37,38) M(41,42) M(45,46)
#define Y_RMW(OPC, r) OPC " v" #r ", v" #r ", v48\n\t"
#define Y_NODEP(OPC, r) OPC " v" #r ", v49, v50\n\t"
#define Y_INL(OPC, r) OPC " v" #r ", 1, v" #r "\n\t"
#define Y_SGPR(OPC, r) OPC " v" #r ", s20, v" #r "\n\t"
#define Y_2(OPC, r, s) OPC " v" #r ", v" #r ", v" #s "\n\t"
#define Y_ONE(OPC, r) OPC " v" #r ", v48\n\t"
#define Y_ONE_RMW(OPC, r) OPC " v" #r ", v" #r "\n\t"
#define P_ADDU_RMW(r) Y_RMW("v_add_u32", r)
#define P_ADDU_E64(r) Y_RMW("v_add_u32_e64", r)
#define P_ADDU_NODEP(r) Y_NODEP("v_add_u32", r)
#define P_ADDU_INL(r) Y_INL("v_add_u32", r)
#define P_ADDU_SGPR(r) Y_SGPR("v_add_u32", r)
#define P_ADDU_2(r, s) Y_2("v_add_u32", r, s)
#define P_ADDF_RMW(r) Y_RMW("v_add_f32", r)
#define P_ADDF_E64(r) Y_RMW("v_add_f32_e64", r)
#define P_ADDF_NODEP(r) Y_NODEP("v_add_f32", r)
#define P_ADDF_2(r, s) Y_2("v_add_f32", r, s)
#define P_MULF_RMW(r) Y_RMW("v_mul_f32", r)
#define P_FMAC(r) Y_NODEP("v_fmac_f32", r)
#define P_XOR_INL(r) Y_INL("v_xor_b32", r)
#define P_XOR_RMW(r) Y_RMW("v_xor_b32", r)
#define P_OR_RMW(r) Y_RMW("v_or_b32", r)
#define P_AND_RMW(r) Y_RMW("v_and_b32", r)
#define P_AND_NODEP(r) Y_NODEP("v_and_b32", r)
#define P_LSHR_INL(r) "v_lshrrev_b32 v" #r ", 8, v" #r "\n\t"
#define P_LSHR_VGPR(r) "v_lshrrev_b32 v" #r ", v48, v" #r "\n\t"
#define P_LSHL_INL(r) "v_lshlrev_b32 v" #r ", 3, v" #r "\n\t"
#define P_ASHR_INL(r) "v_ashrrev_i32 v" #r ", 8, v" #r "\n\t"
#define P_MAXU(r) Y_RMW("v_max_u32", r)
#define P_MINI(r) Y_RMW("v_min_i32", r)
#define P_SUBU_NODEP(r) Y_NODEP("v_sub_u32", r)
#define P_SUBU_RMW(r) Y_RMW("v_sub_u32", r)
#define P_SUBREV(r) Y_RMW("v_subrev_u32", r)
#define P_MOV(r) Y_ONE("v_mov_b32", r)
#define P_MOV_E64(r) Y_ONE("v_mov_b32_e64", r)
#define P_NOT(r) Y_ONE_RMW("v_not_b32", r)
#define P_CVTFU(r) Y_ONE_RMW("v_cvt_f32_u32", r)
#define P_CVTUF(r) Y_ONE_RMW("v_cvt_u32_f32", r)
#define P_MOV_DPP(r) "v_mov_b32_dpp v" #r ", v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define P_ADDCO(r) "v_add_co_u32 v" #r ", vcc, v" #r ", v48\n\t"
#define P_CNDM(r) "v_cndmask_b32 v" #r ", v49, v50, vcc\n\t"
#define P_MUL24(r) Y_RMW("v_mul_u32_u24", r)
#define P_MULHI24(r) Y_RMW("v_mul_hi_u32_u24", r)
#define P_MAD24(r) "v_mad_u32_u24 v" #r ", v" #r ", v48, v49\n\t"
#define P_ADD3(r) "v_add3_u32 v" #r ", v" #r ", v48, v49\n\t"
#define P_PKADDU16(r) Y_RMW("v_pk_add_u16", r)
#define P_ADDU16(r) Y_RMW("v_add_u16", r)
#define P_ADDF16(r) Y_RMW("v_add_f16", r)
#define P_MAD_ADDU(r, p) "v_add_u32 v" #r ", v" #r ", v48\n\tv_mad_u64_u32 v[" #p ":" #p "+1], s[22:23], v49, v50, v[" #p ":" #p "+1]\n\t"
#define P_MAD_ADDF(r, p) "v_add_f32 v" #r ", v" #r ", v48\n\tv_mad_u64_u32 v[" #p ":" #p "+1], s[22:23], v49, v50, v[" #p ":" #p "+1]\n\t"
#define P_MAD_MOV(r, p) "v_mov_b32 v" #r ", v48\n\tv_mad_u64_u32 v[" #p ":" #p "+1], s[22:23], v49, v50, v[" #p ":" #p "+1]\n\t"
#define P_MAD_ONLY(r, p) "v_mad_u64_u32 v[" #p ":" #p "+1], s[22:23], v49, v50, v[" #p ":" #p "+1]\n\tv_mad_u64_u32 v[" #p ":" #p "+1], s[22:23], v50, v49, v[" #p ":" #p "+1]\n\t"
#define P_ADDU_ADDF(r, p) "v_add_u32 v" #r ", v" #r ", v48\n\tv_add_f32 v" #p ", v" #p ", v49\n\t"
#define P_SUBCO_CNDM(r, p) "v_sub_co_u32 v" #r ", vcc, v" #r ", v48\n\tv_cndmask_b32 v" #p ", v49, v50, vcc\n\t"
// second batch: what co-issues with the 64-bit multiply-add?  (r = a 32-bit register of an odd pair, p = an even pair)
#define MADP(p) "v_mad_u64_u32 v[" #p ":" #p "+1], s[22:23], v49, v50, v[" #p ":" #p "+1]\n\t"
#define P_MAD_FMA64(r, p) MADP(p) "v_fma_f64 v[52:53], v[54:55], v[54:55], v[52:53]\n\t"
#define P_MAD_PKFMA(r, p) MADP(p) "v_pk_fma_f32 v[52:53], v[54:55], v[54:55], v[52:53]\n\t"
#define P_MAD_FMA32(r, p) MADP(p) "v_fma_f32 v" #r ", v" #r ", v50, v48\n\t"
#define P_MAD_XOR(r, p) MADP(p) "v_xor_b32 v" #r ", v" #r ", v48\n\t"
#define P_MAD_LSHR(r, p) MADP(p) "v_lshrrev_b32 v" #r ", 8, v" #r "\n\t"
#define P_MAD_AND(r, p) MADP(p) "v_and_b32 v" #r ", v" #r ", v48\n\t"
#define P_MAD_SUBU(r, p) MADP(p) "v_sub_u32 v" #r ", v" #r ", v48\n\t"
#define P_MAD_ADDCO(r, p) MADP(p) "v_add_co_u32 v" #r ", vcc, v" #r ", v48\n\t"
#define P_MAD_ADDCE(r, p) MADP(p) "v_addc_co_u32_e64 v" #r ", s[20:21], v" #r ", v48, s[20:21]\n\t"
#define P_MAD_MULF(r, p) MADP(p) "v_mul_f32 v" #r ", v" #r ", v50\n\t"
#define P_MAD_CVT64(r, p) MADP(p) "v_cvt_f64_u32 v[52:53], v48\n\t"
#define P_MAD_2ADDF(r, p) MADP(p) "v_add_f32 v" #r ", v" #r ", v48\n\tv_add_f32 v56, v56, v48\n\t"
#define P_MAD_MAD24(r, p) MADP(p) "v_mad_u32_u24 v" #r ", v" #r ", v48, v49\n\t"
#define P_MAD_MULLO(r, p) MADP(p) "v_mul_lo_u32 v" #r ", v" #r ", v48\n\t"
#define P_MAD_LSHLADD64(r, p) MADP(p) "v_lshl_add_u64 v[52:53], v[52:53], 0, v[54:55]\n\t"
#define P_MAD_CNDM(r, p) MADP(p) "v_cndmask_b32 v" #r ", v49, v50, s[20:21]\n\t"
#define P_FMA64_ONLY(r, p) "v_fma_f64 v[" #p ":" #p "+1], v[54:55], v[54:55], v[" #p ":" #p "+1]\n\tv_fma_f64 v[52:53], v[54:55], v[54:55], v[52:53]\n\t"
#define P_FMA64_ADDU(r, p) "v_fma_f64 v[" #p ":" #p "+1], v[54:55], v[54:55], v[" #p ":" #p "+1]\n\tv_add_u32 v" #r ", v" #r ", v48\n\t"
#define P_ADDCHAIN(r) "v_add_u32 v16, v16, v48\n\t"
#define P_MADRUN(r, p) MADP(p)
#define P_ADDRUN(r, p) "v_add_u32 v" #r ", v" #r ", v48\n\t"
#define P_MAD4(r, p) MADP(p) MADP(p)
#define P_ADD4(r, p) "v_add_u32 v" #r ", v" #r ", v48\n\tv_sub_u32 v" #r ", v" #r ", v49\n\t"
#define XP4A(M) M(16,18) M(20,22)
#define XP4B(M) M(24,26) M(28,30)
#define XP4C(M) M(32,34) M(36,38)
#define XP4D(M) M(40,42) M(44,46)
#define XP4E(M) M(17,18) M(21,22)
#define XP4F(M) M(25,26) M(29,30)
#define XP4G(M) M(33,34) M(37,38)
#define XP4H(M) M(41,42) M(45,46)
// ---- round 6, fourth batch: the S-box multiply's 128-bit assembly without carries?  Today: 4 multiply-adds + v_add_co + 2 v_addc (then 7
// reduction instructions).  Alternative: chain the partial products through the multiply-add's 64-bit addend -- t = a0*b0;
// M = a1*b0 + (t.hi, 0); M2 = a0*b1 + (M.lo, 0); Q = a1*b1 + (M.hi, 0); Q += M2.hi * 1 -- 5 multiply-adds + 3 v_mov_b32 (the moves
// build the (x, 0) pairs and ride the multiply-adds), same 7 reduction instructions.  Mixes only (no data dependence), x3 streams.
#define MUL_OLD(p, r) MADP(p) MADP(p) MADP(p) MADP(p) \
    "v_add_co_u32 v" #r ", s[20:21], v" #r ", v48\n\tv_addc_co_u32 v" #r ", s[20:21], v" #r ", v48, s[20:21]\n\tv_addc_co_u32 v" #r ", s[20:21], v" #r ", 0, s[20:21]\n\t" \
    "v_subb_co_u32 v" #r ", s[20:21], v" #r ", v48, s[20:21]\n\tv_subb_co_u32 v" #r ", s[20:21], v" #r ", 0, s[20:21]\n\t" MADP(p) \
    "v_cndmask_b32 v" #r ", 0, 1, s[20:21]\n\tv_subb_co_u32 v" #r ", s[20:21], v" #r ", 0, s[20:21]\n\tv_add_u32 v" #r ", v" #r ", v48\n\t" \
    "v_mad_i64_i32 v[" #p ":" #p "+1], s[22:23], v49, -1, v[" #p ":" #p "+1]\n\t"
#define MUL_NEW(p, r) MADP(p) "v_mov_b32 v" #r ", v48\n\t" MADP(p) "v_mov_b32 v" #r ", v49\n\t" MADP(p) "v_mov_b32 v" #r ", v50\n\t" MADP(p) MADP(p) \
    "v_sub_co_u32 v" #r ", s[20:21], v" #r ", v48\n\tv_subb_co_u32 v" #r ", s[20:21], v" #r ", 0, s[20:21]\n\t" MADP(p) \
    "v_cndmask_b32 v" #r ", 0, 1, s[20:21]\n\tv_subb_co_u32 v" #r ", s[20:21], v" #r ", 0, s[20:21]\n\tv_add_u32 v" #r ", v" #r ", v48\n\t" \
    "v_mad_i64_i32 v[" #p ":" #p "+1], s[22:23], v49, -1, v[" #p ":" #p "+1]\n\t"
#define MUL3_OLD MUL_OLD(18, 16) MUL_OLD(22, 20) MUL_OLD(26, 24)
#define MUL3_NEW MUL_NEW(18, 16) MUL_NEW(22, 20) MUL_NEW(26, 24)
#define XCLOB "vcc", "s20", "s21", "s22", "s23", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", \
              "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", \
              "v48", "v49", "v50", "v51"
#define XCLOB2 XCLOB, "v52", "v53", "v54", "v55", "v56"
#define P_INIT(r) "v_mov_b32 v" #r ", %0\n\t"
#define P_SUM(r) "v_xor_b32 %0, %0, v" #r "\n\t"

template <int OP>
__global__ void __launch_bounds__(256) kx(uint64_t *out, uint32_t seed, uint64_t *ticks) {
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    uint32_t c = t * 2654435761u + seed;
    asm volatile(XR(P_INIT) "v_mov_b32 v48, %0\n\tv_mov_b32 v49, %0\n\tv_mov_b32 v50, 0x3f800001\n\tv_mov_b32 v51, %0\n\ts_mov_b32 s20, 77\n\ts_mov_b32 s21, 0\n\tv_mov_b32 v52, 0\n\tv_mov_b32 v53, 0x3ff00000\n\tv_mov_b32 v54, 1\n\tv_mov_b32 v55, 0x3ff00000\n\tv_mov_b32 v56, %0" ::"v"(c) : XCLOB2);
    uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS / 4; ++it) {
        if (OP == 100) asm volatile(XR(P_ADDU_RMW) ::: XCLOB);
        if (OP == 101) asm volatile(XR(P_ADDU_E64) ::: XCLOB);
        if (OP == 102) asm volatile(XR(P_ADDU_NODEP) ::: XCLOB);
        if (OP == 103) asm volatile(XR(P_ADDU_INL) ::: XCLOB);
        if (OP == 104) asm volatile(XR(P_ADDU_SGPR) ::: XCLOB);
        if (OP == 105) asm volatile(XB(P_ADDU_2) ::: XCLOB);
        if (OP == 106) asm volatile(XD(P_ADDU_2) ::: XCLOB);
        if (OP == 107) asm volatile(XR(P_ADDF_RMW) ::: XCLOB);
        if (OP == 108) asm volatile(XR(P_ADDF_E64) ::: XCLOB);
        if (OP == 109) asm volatile(XR(P_ADDF_NODEP) ::: XCLOB);
        if (OP == 110) asm volatile(XR(P_MULF_RMW) ::: XCLOB);
        if (OP == 111) asm volatile(XR(P_FMAC) ::: XCLOB);
        if (OP == 112) asm volatile(XR(P_XOR_INL) ::: XCLOB);
        if (OP == 113) asm volatile(XR(P_OR_RMW) ::: XCLOB);
        if (OP == 114) asm volatile(XR(P_AND_NODEP) ::: XCLOB);
        if (OP == 115) asm volatile(XR(P_LSHR_VGPR) ::: XCLOB);
        if (OP == 116) asm volatile(XR(P_LSHL_INL) ::: XCLOB);
        if (OP == 117) asm volatile(XR(P_ASHR_INL) ::: XCLOB);
        if (OP == 118) asm volatile(XR(P_MAXU) ::: XCLOB);
        if (OP == 119) asm volatile(XR(P_MINI) ::: XCLOB);
        if (OP == 120) asm volatile(XR(P_SUBU_NODEP) ::: XCLOB);
        if (OP == 121) asm volatile(XR(P_SUBREV) ::: XCLOB);
        if (OP == 122) asm volatile(XR(P_MOV_E64) ::: XCLOB);
        if (OP == 123) asm volatile(XR(P_NOT) ::: XCLOB);
        if (OP == 124) asm volatile(XR(P_CVTFU) ::: XCLOB);
        if (OP == 125) asm volatile(XR(P_CVTUF) ::: XCLOB);
        if (OP == 126) asm volatile(XR(P_MOV_DPP) ::: XCLOB);
        if (OP == 127) asm volatile(XR(P_ADDCO) ::: XCLOB);
        if (OP == 128) asm volatile(XP(P_MAD_ADDU) ::: XCLOB);
        if (OP == 129) asm volatile(XP(P_MAD_ADDF) ::: XCLOB);
        if (OP == 130) asm volatile(XP(P_MAD_MOV) ::: XCLOB);
        if (OP == 131) asm volatile(XP(P_ADDU_ADDF) ::: XCLOB);
        if (OP == 132) asm volatile(XP(P_MAD_ONLY) ::: XCLOB);
        if (OP == 133) asm volatile(XR(P_CNDM) ::: XCLOB);
        if (OP == 134) asm volatile(XR(P_MUL24) ::: XCLOB);
        if (OP == 135) asm volatile(XR(P_MULHI24) ::: XCLOB);
        if (OP == 136) asm volatile(XR(P_MAD24) ::: XCLOB);
        if (OP == 137) asm volatile(XR(P_ADD3) ::: XCLOB);
        if (OP == 138) asm volatile(XR(P_AND_RMW) ::: XCLOB);
        if (OP == 139) asm volatile(XR(P_LSHR_INL) ::: XCLOB);
        if (OP == 140) asm volatile(XR(P_MOV) ::: XCLOB);
        if (OP == 141) asm volatile(XR(P_XOR_RMW) ::: XCLOB);
        if (OP == 142) asm volatile(XR(P_SUBU_RMW) ::: XCLOB);
        if (OP == 143) asm volatile(XD(P_ADDF_2) ::: XCLOB);
        if (OP == 144) asm volatile(XB(P_ADDF_2) ::: XCLOB);
        if (OP == 145) asm volatile(XR(P_PKADDU16) ::: XCLOB);
        if (OP == 146) asm volatile(XR(P_ADDU16) ::: XCLOB);
        if (OP == 147) asm volatile(XR(P_ADDF16) ::: XCLOB);
        if (OP == 148) asm volatile(XP(P_SUBCO_CNDM) ::: XCLOB);
        if (OP == 169) asm volatile(XR(P_ADDCHAIN) ::: XCLOB2);
        if (OP == 170) asm volatile(MUL3_OLD MUL3_OLD ::: XCLOB2);
        if (OP == 171) asm volatile(MUL3_NEW MUL3_NEW ::: XCLOB2);
        if (OP == 149) asm volatile(XP(P_MADRUN) XP(P_ADDRUN) ::: XCLOB2);
        if (OP == 150) asm volatile(XP(P_MAD_FMA64) ::: XCLOB2);
        if (OP == 151) asm volatile(XP(P_MAD_PKFMA) ::: XCLOB2);
        if (OP == 152) asm volatile(XP(P_MAD_FMA32) ::: XCLOB2);
        if (OP == 153) asm volatile(XP(P_MAD_XOR) ::: XCLOB2);
        if (OP == 154) asm volatile(XP(P_MAD_LSHR) ::: XCLOB2);
        if (OP == 155) asm volatile(XP(P_MAD_AND) ::: XCLOB2);
        if (OP == 156) asm volatile(XP(P_MAD_SUBU) ::: XCLOB2);
        if (OP == 157) asm volatile(XP(P_MAD_ADDCO) ::: XCLOB2);
        if (OP == 158) asm volatile(XP(P_MAD_ADDCE) ::: XCLOB2);
        if (OP == 160) asm volatile(XP(P_MAD_MULF) ::: XCLOB2);
        if (OP == 161) asm volatile(XP(P_MAD_CVT64) ::: XCLOB2);
        if (OP == 162) asm volatile(XP(P_MAD_2ADDF) ::: XCLOB2);
        if (OP == 163) asm volatile(XP(P_MAD_MAD24) ::: XCLOB2);
        if (OP == 164) asm volatile(XP(P_MAD_MULLO) ::: XCLOB2);
        if (OP == 165) asm volatile(XP(P_MAD_LSHLADD64) ::: XCLOB2);
        if (OP == 166) asm volatile(XP(P_MAD_CNDM) ::: XCLOB2);
        if (OP == 159) asm volatile(XP(P_FMA64_ONLY) ::: XCLOB2);
        if (OP == 168) asm volatile(XP(P_FMA64_ADDU) ::: XCLOB2);
        if (OP == 167) asm volatile(XP4A(P_MAD4) XP4A(P_ADD4) XP4B(P_MAD4) XP4B(P_ADD4) XP4C(P_MAD4) XP4C(P_ADD4) XP4D(P_MAD4) XP4D(P_ADD4) ::: XCLOB2);
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    asm volatile(XR(P_SUM) : "+v"(s)::XCLOB);
    out[t] = s;
    if ((threadIdx.x & 63) == 0) ticks[t >> 6] = t1 - t0;
}

static int g_waves = 8;        // resident waves per SIMD (blocks of 256 threads per CU)
static const char *g_only = nullptr;
static bool g_json = false;

// ---- round 6, third batch: does phase-synchronising the waves of a SIMD let the cheap integer class pair?  One 1024-thread
// workgroup per CU (16 waves = 4 per SIMD, all behind ONE s_barrier domain), explicit registers as above.
//   OP 200  every wave: 32 adds then 32 multiply-adds, no barrier          201  the same, s_barrier after each phase
//   OP 202  96 adds / 64 multiply-adds (the limb NTT's proportions), no barrier   203  the same with barriers
//   OP 204  even waves only adds, odd waves only multiply-adds              205  even waves v_add_u32, odd waves v_xor_b32
//   OP 206  waves 0-1 of each SIMD adds, waves 2-3 multiply-adds (wave >> 3 parity: waves w and w+4.. share a SIMD? see ubench output)
#define XADD32 XR(P_ADDU_RMW)
#define XMAD32 XP(P_MAD_ONLY)
template <int OP>
__global__ void __launch_bounds__(1024) ky(uint64_t *out, uint32_t seed, uint64_t *ticks) {
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    uint32_t c = t * 2654435761u + seed;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile(XR(P_INIT) "v_mov_b32 v48, %0\n\tv_mov_b32 v49, %0\n\tv_mov_b32 v50, 0x3f800001\n\tv_mov_b32 v51, %0\n\ts_mov_b32 s20, 77" ::"v"(c) : XCLOB);
    uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS / 8; ++it) {
        if (OP == 200) asm volatile(XADD32 XMAD32 ::: XCLOB);
        if (OP == 201) asm volatile(XADD32 "s_barrier\n\t" XMAD32 "s_barrier\n\t" ::: XCLOB);
        if (OP == 202) asm volatile(XADD32 XADD32 XADD32 XMAD32 XMAD32 ::: XCLOB);
        if (OP == 203) asm volatile(XADD32 XADD32 XADD32 "s_barrier\n\t" XMAD32 XMAD32 "s_barrier\n\t" ::: XCLOB);
        if (OP == 204) { if (wave & 1) asm volatile(XMAD32 XMAD32 ::: XCLOB); else asm volatile(XADD32 XADD32 ::: XCLOB); }
        if (OP == 205) { if (wave & 1) asm volatile(XR(P_XOR_RMW) XR(P_XOR_RMW) ::: XCLOB); else asm volatile(XADD32 XADD32 ::: XCLOB); }
        if (OP == 206) { if (wave & 8) asm volatile(XMAD32 XMAD32 ::: XCLOB); else asm volatile(XADD32 XADD32 ::: XCLOB); }
        if (OP == 207) asm volatile(XADD32 XADD32 ::: XCLOB);
        if (OP == 208) asm volatile(XMAD32 XMAD32 ::: XCLOB);
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    asm volatile(XR(P_SUM) : "+v"(s)::XCLOB);
    out[t] = s;
    if ((threadIdx.x & 63) == 0) ticks[t >> 6] = t1 - t0;
}

template <int OP>
void run_y(const char *name, int per_iter) {
    if (g_only && !strstr(name, g_only)) return;
    const int blocks = 256, threads = 1024;
    uint64_t *d, *tk;
    (void)hipMalloc(&d, (size_t)blocks * threads * 8);
    (void)hipMalloc(&tk, (size_t)blocks * threads / 64 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    ky<OP><<<blocks, threads>>>(d, 1, tk);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        ky<OP><<<blocks, threads>>>(d, 2 + rep, tk);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double inst_per_simd = 4.0 * (ITERS / 8) * per_iter;
    if (g_json)
        printf("{\"op\": %d, \"name\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_insts_per_simd\": %.0f, \"per_iter\": %d}\n", OP, name, 4, best, inst_per_simd, per_iter);
    else
        printf("%-44s %7.3f ms  %5.2f cyc/inst/SIMD by wall@2.4GHz(assumed)\n", name, best, best * 1e-3 * 2.4e9 / inst_per_simd);
    (void)hipFree(tk);
    (void)hipFree(d);
}

template <int OP, bool X = false>
void run(const char *name, int per_iter = 8) {
    if (g_only && !strstr(name, g_only)) return;
    const int blocks = 256 * g_waves, threads = 256;  // g_waves blocks/CU -> g_waves waves/SIMD
    uint64_t *d, *tk;
    (void)hipMalloc(&d, (size_t)blocks * threads * 8);
    (void)hipMalloc(&tk, (size_t)blocks * threads / 64 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    if (X) kx<OP><<<blocks, threads>>>(d, 1, tk); else k<OP><<<blocks, threads>>>(d, 1, tk);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        if (X) kx<OP><<<blocks, threads>>>(d, 2 + rep, tk); else k<OP><<<blocks, threads>>>(d, 2 + rep, tk);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double waves_per_simd = (double)blocks * threads / 64 / 1024;
    double inst_per_simd = waves_per_simd * (X ? ITERS / 4 : ITERS) * per_iter;
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
               cycles / inst_per_simd, avg / ((X ? ITERS / 4 : ITERS) * (double)per_iter * waves_per_simd), avg);
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
    run<54>("mix hash_leaves r06", 32);
    run<55>("mix hash_leaves r06 x4", 128);
    run<56>("mix hash_leaves r06b", 32);
    run<57>("mix hash_leaves r06b x4", 128);
    run<51>("v_mfma_i32_4x4x4_16b_i8", 9);   // 8 MFMAs + 1 v_mov per trip
    run<52>("mix mds valu (half layer)", 173);
    run<53>("mix mds mfma (half layer)", 169);
    run<49>("mix limb_ntt", 16);
    run<100, true>("x add_u32 rmw", 32);
    run<101, true>("x add_u32 e64", 32);
    run<102, true>("x add_u32 nodep", 32);
    run<103, true>("x add_u32 inline", 32);
    run<104, true>("x add_u32 sgpr", 32);
    run<105, true>("x add_u32 samebank", 32);
    run<106, true>("x add_u32 nextbank", 32);
    run<107, true>("x add_f32 rmw", 32);
    run<108, true>("x add_f32 e64", 32);
    run<109, true>("x add_f32 nodep", 32);
    run<143, true>("x add_f32 nextbank", 32);
    run<144, true>("x add_f32 samebank", 32);
    run<110, true>("x mul_f32 rmw", 32);
    run<111, true>("x fmac_f32", 32);
    run<112, true>("x xor inline", 32);
    run<141, true>("x xor rmw", 32);
    run<113, true>("x or rmw", 32);
    run<138, true>("x and rmw", 32);
    run<114, true>("x and nodep", 32);
    run<139, true>("x lshrrev inline", 32);
    run<115, true>("x lshrrev vgpr", 32);
    run<116, true>("x lshlrev inline", 32);
    run<117, true>("x ashrrev inline", 32);
    run<118, true>("x max_u32", 32);
    run<119, true>("x min_i32", 32);
    run<142, true>("x sub_u32 rmw", 32);
    run<120, true>("x sub_u32 nodep", 32);
    run<121, true>("x subrev_u32", 32);
    run<140, true>("x mov", 32);
    run<122, true>("x mov e64", 32);
    run<123, true>("x not", 32);
    run<124, true>("x cvt_f32_u32", 32);
    run<125, true>("x cvt_u32_f32", 32);
    run<126, true>("x mov dpp", 32);
    run<127, true>("x add_co vcc", 32);
    run<133, true>("x cndmask nodep", 32);
    run<134, true>("x mul_u32_u24 e32", 32);
    run<135, true>("x mul_hi_u32_u24 e32", 32);
    run<136, true>("x mad_u32_u24", 32);
    run<137, true>("x add3", 32);
    run<145, true>("x pk_add_u16", 32);
    run<146, true>("x add_u16", 32);
    run<147, true>("x add_f16", 32);
    run<132, true>("x mad64 only", 32);
    run<128, true>("x mad64+add_u32 1:1", 32);
    run<129, true>("x mad64+add_f32 1:1", 32);
    run<130, true>("x mad64+mov 1:1", 32);
    run<131, true>("x add_u32+add_f32 1:1", 32);
    run<148, true>("x sub_co+cndmask 1:1", 32);
    run<169, true>("x add_u32 dependent chain", 32);
    run<170, true>("x mul mix: 4 mad + 3 carry (14 instr) x6", 84);
    run<171, true>("x mul mix: 5 mad + 3 mov (15 instr) x6", 90);
    run<149, true>("x mad64 x16 then add_u32 x16", 32);
    run<150, true>("x mad64+fma_f64 1:1", 32);
    run<151, true>("x mad64+pk_fma_f32 1:1", 32);
    run<152, true>("x mad64+fma_f32 1:1", 32);
    run<153, true>("x mad64+xor 1:1", 32);
    run<154, true>("x mad64+lshrrev 1:1", 32);
    run<155, true>("x mad64+and 1:1", 32);
    run<156, true>("x mad64+sub_u32 1:1", 32);
    run<157, true>("x mad64+add_co vcc 1:1", 32);
    run<158, true>("x mad64+addc e64 1:1", 32);
    run<160, true>("x mad64+mul_f32 1:1", 32);
    run<161, true>("x mad64+cvt_f64_u32 1:1", 32);
    run<162, true>("x mad64+2 add_f32", 48);
    run<163, true>("x mad64+mad_u32_u24 1:1", 32);
    run<164, true>("x mad64+mul_lo_u32 1:1", 32);
    run<165, true>("x mad64+lshl_add_u64 1:1", 32);
    run<166, true>("x mad64+cndmask sgpr 1:1", 32);
    run<159, true>("x fma_f64 only", 32);
    run<168, true>("x fma_f64+add_u32 1:1", 32);
    run<167, true>("x mad64 x4 / add,sub x4 runs", 32);
    if (g_waves == 4) {
        run_y<207>("y adds only (1024-thread WG)", 64);
        run_y<208>("y mad64 only (1024-thread WG)", 64);
        run_y<200>("y 32 adds / 32 mad64, no barrier", 64);
        run_y<201>("y 32 adds / 32 mad64, barriers", 64);
        run_y<202>("y 96 adds / 64 mad64, no barrier", 160);
        run_y<203>("y 96 adds / 64 mad64, barriers", 160);
        run_y<204>("y even waves adds, odd waves mad64", 64);
        run_y<205>("y even waves add_u32, odd waves xor", 64);
        run_y<206>("y waves 0-7 adds, waves 8-15 mad64", 64);
    }
    return 0;
}
