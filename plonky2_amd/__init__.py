"""plonky2_amd -- the MI355X-native hot path of the plonky2 prover.

What is here is exactly the path of SURVEY.md section 8: PolynomialBatch LDE + Poseidon Merkle
commit and the FRI commit phase, as hand-written HIP kernels for gfx950 behind the C ABI of
include/p2hot.h, plus this thin host-side mirror of the reference's interface for that path:

  plonky2_amd.fri.oracle.PolynomialBatch      <- plonky2/src/fri/oracle.rs
  plonky2_amd.hash.merkle_tree.MerkleTree     <- plonky2/src/hash/merkle_tree.rs
  plonky2_amd.hash.poseidon                   <- plonky2/src/hash/poseidon.rs (permutation, hash_no_pad, two_to_one)
  plonky2_amd.iop.challenger.Challenger       <- plonky2/src/iop/challenger.rs
  plonky2_amd.fri.prover                      <- plonky2/src/fri/prover.rs (commit phase, proof of work)
  plonky2_amd.field.fft / .polynomial         <- field/src/fft.rs, field/src/polynomial/mod.rs
  plonky2_amd.distributed                     <- (new) coset-sharded commit over RCCL

There is no CPU fallback: the engine raises when libp2hot.so or the GPU is missing.
"""
from .engine import COSET_SHIFT, P, Engine, default_engine, set_default_engine  # noqa: F401
