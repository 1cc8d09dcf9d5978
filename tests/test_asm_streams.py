"""CPU tier: the product's inline-asm instruction streams, executed.

The kernel-source emulator normally runs the C fallbacks of the hand-written gfx950 blocks (they are ~30x faster).  Here
its instruction interpreter (tests/emu/gcn_asm.*) is switched on, so the very template strings hipcc assembles --
gl_mul3.hpp mul3 / mul1 / mul1_lowregs / fold3 / fold1, poseidon.hpp mds_term / mds_first, ntt.hpp mul_pow2_asm -- are
decoded, checked (gfx940+ "VALU writes SGPR -> VALU reads it" wait states, clobber lists, read-before-write) and executed
lane by lane under the same parity tests that run on the MI355X: field edge grid, the reference's Poseidon KATs through
all three lane mappings, sponge / two_to_one, NTTs through the word passes and the limb passes, a commitment, a FRI commit
phase.  A kernel edit made in the GPU-less container that breaks an instruction stream now fails here, not at the next gpurun.
"""
import ctypes as C

import numpy as np
import pytest

from tests import test_parity as tp
from tests.conftest import P


class AsmTier:
    def __init__(self, lib):
        self.lib = lib
        lib.p2hot_emu_asm_stats.restype = C.c_ulonglong
        lib.p2hot_emu_asm_stats.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_char_p, C.c_size_t]
        lib.p2hot_emu_asm.argtypes = [C.c_int]
        lib.p2hot_emu_asm_negative_tests.restype = C.c_uint

    def stats(self):
        blocks, errors, buf = C.c_ulonglong(), C.c_ulonglong(), C.create_string_buffer(1024)
        n = self.lib.p2hot_emu_asm_stats(C.byref(blocks), C.byref(errors), buf, 1024)
        return {"instructions": n, "blocks": blocks.value, "errors": errors.value, "first_error": buf.value.decode()}


@pytest.fixture(scope="module")
def tier(emu):
    return AsmTier(emu.lib)  # the library the emulator engine is bound to (tests/emu/libp2hot_emu.so, or an ASAN build of it)


@pytest.fixture
def emu_asm(emu, tier):
    """the emulator engine with the instruction interpreter on; the test must have executed asm blocks without a report"""
    before = tier.stats()
    was = tier.lib.p2hot_emu_asm(1)  # (already on when the whole tier runs under P2HOT_EMU_ASM=1)
    try:
        yield emu
    finally:
        tier.lib.p2hot_emu_asm(was)
    after = tier.stats()
    assert after["errors"] == before["errors"], after["first_error"]
    assert after["blocks"] > before["blocks"], "no asm block was interpreted: the test did not reach a hand-written stream"


def test_the_checker_catches_what_it_claims(tier):
    """deliberately broken blocks (carry read one wait state early, undeclared clobber, scratch register read before it is
    written, unmodelled instruction, input used as destination, output never written) are reported; a sound block is not"""
    assert tier.lib.p2hot_emu_asm_negative_tests() == 0x7F


def test_field_asm_streams_on_the_reference_edge_grid(emu_asm):
    """mul1 / mul3 / mul1_lowregs / mul_pow2_asm<S> (every shift the butterflies use and the boundary shifts) / fold1 / fold3
    against big-integer arithmetic and the compiler's streams (field/src/prime_field_testing.rs:8-17 grid)"""
    tp.test_field_ops_edge_grid(emu_asm)


def test_poseidon_kats_through_the_asm(emu_asm, kats, ora):
    """plonky2/src/hash/poseidon_goldilocks.rs:455-490 through mds_first / mds_term / fold3 / mul3 / sbox7_asm"""
    tp.test_poseidon_reference_kats(emu_asm, kats)
    tp.test_row_poseidon_kats_and_edges(emu_asm, ora, kats)


def test_poseidon_lane_mappings_through_the_asm(emu_asm, ora):
    from plonky2_amd.hash.merkle_tree import MerkleTree
    rng = np.random.default_rng(5)
    eng = emu_asm
    try:
        for (n, w, cap) in ((32, 9, 1), (16, 135, 2)):
            leaves = tp.rand_field(rng, n, w, noncanonical=True)
            digests, capv = ora.merkle_tree(leaves, cap)
            for (quad, row) in ((0, 0), (1 << 20, 0), (0, 8)):
                eng.check(eng.lib.p2hot_tune_quad(eng.ctx, quad))
                eng.check(eng.lib.p2hot_tune_row(eng.ctx, row))
                t = MerkleTree.new(leaves, cap, engine=eng)
                assert (t.cap.entries == capv).all(), (n, w, cap, quad, row)
                assert (np.asarray(t.digests).reshape(-1, 4) == digests).all(), (n, w, cap, quad, row)
    finally:
        eng.check(eng.lib.p2hot_tune_quad(eng.ctx, tp.EMU_TUNE_QUAD))
        eng.check(eng.lib.p2hot_tune_row(eng.ctx, tp.EMU_TUNE_ROW))


def test_sponge_and_two_to_one_through_the_asm(emu_asm, ora):
    tp.test_hash_no_pad_and_two_to_one(emu_asm, ora)


@pytest.mark.parametrize("log_n", [3, 6, 9, 11, 12, 13])
def test_ntt_passes_through_the_asm(emu_asm, ora, log_n):
    """word passes below 2^12 (mul_pow2_asm twiddles, mul1_lowregs table twiddles), limb passes from 2^12 (mul1, fold1)"""
    tp.test_fft_ifft_vs_oracle(emu_asm, ora, log_n)


def test_coset_lde_through_the_asm(emu_asm, ora):
    tp.test_coset_lde_vs_oracle_and_naive(emu_asm, ora)


def test_commit_and_fri_commit_through_the_asm(emu_asm, ora):
    from plonky2_amd.fri.oracle import PolynomialBatch
    from plonky2_amd.fri.prover import fri_committed_trees
    from plonky2_amd.iop.challenger import Challenger
    eng = emu_asm
    rng = np.random.default_rng(17)
    W, log_n, rb, cap = 11, 6, 3, 2
    vals = rng.integers(0, P, size=(W, 1 << log_n), dtype=np.uint64)
    batch = PolynomialBatch.from_values(vals, rb, False, cap, engine=eng)
    ref = ora.commit(vals, rb, cap, True)
    assert (batch.polynomials == ref["coeffs"]).all()
    assert (batch.merkle_tree.cap.entries == ref["cap"]).all()
    assert (batch.merkle_tree.digests == ref["digests"]).all()
    assert (batch.merkle_tree.leaves == ref["leaves"]).all()
    co = rng.integers(0, P, size=(1 << log_n, 2), dtype=np.uint64)
    pad = np.zeros((1 << (log_n + rb), 2), dtype=np.uint64)
    pad[:1 << log_n] = co
    ch, och = Challenger(eng), ora.Challenger()
    trees, final, betas = fri_committed_trees(co, ch, rb, cap, [2, 2], engine=eng)
    o = ora.fri_commit(pad, rb, cap, [2, 2], och)
    assert all((t.cap.entries == c).all() for t, c in zip(trees, o["caps"]))
    assert (final == o["final"]).all() and (betas == o["betas"]).all()


def test_prove_openings_through_the_asm(emu_asm, ora):
    """the whole opening proof (final_poly, FRI commit phase, grind, query rounds) with every hand-written stream interpreted"""
    from tests import test_prove_openings as tpo
    tpo.test_final_poly_and_prove_openings_vs_oracle(emu_asm, ora, 5, [3, 2], 3, 2, [2], None)
