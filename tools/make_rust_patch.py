#!/usr/bin/env python3
"""Regenerates integration/plonky2_p2hot.patch against the reference tree (/root/reference).

The patch is the Rust side of the drop-in boundary: Cargo feature `p2hot`, the module
`plonky2/src/p2hot.rs` (= integration/p2hot.rs, verbatim) and the feature-gated early returns in
  fri/oracle.rs        from_values / from_coeffs / get_lde_values / prove_openings
  fri/prover.rs        fri_committed_trees
  plonk/proof.rs       OpeningSet::new: eval_commitment through p2hot_eval_openings
  plonk/prover.rs      all_wires_permutation_partial_products through p2hot_partial_products;
                       compute_quotient_polys: the gate terms on the CPU, the permutation terms + coset_ifft on the GPU
  hash/merkle_tree.rs  `device` handle on MerkleTree, get / num_leaves / prove (the leaf matrix may be ONE flat buffer behind it)
  iop/challenger.rs    accessor for the transcript state
  util/serialization   the one other MerkleTree struct literal; write_merkle_tree reads rows through get / num_leaves
  fri/prover.rs        also: the grind takes the smallest witness under the feature; a test hook for the harness
  starky/Cargo.toml, starky/src/proof.rs   feature forwarding; StarkOpeningSet::new: eval_commitment through p2hot_eval_openings
plus plonky2/examples/p2hot_dump_goldens.rs (= integration/p2hot_dump_goldens.rs, verbatim): the golden dumper.
It is built by anchored edits of a scratch copy, so this script holds only the NEW lines and short
anchors -- no reference source is stored in the repository beyond the diff context of the patch itself.

    python tools/make_rust_patch.py        (build container only: needs /root/reference and `diff`)
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("P2_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "integration", "plonky2_p2hot.patch")

FILES = ["plonky2/Cargo.toml", "plonky2/src/lib.rs", "plonky2/src/fri/oracle.rs", "plonky2/src/fri/prover.rs", "plonky2/src/plonk/prover.rs",
         "plonky2/src/plonk/proof.rs",
         "plonky2/src/hash/merkle_tree.rs", "plonky2/src/iop/challenger.rs", "plonky2/src/util/serialization/mod.rs",
         "starky/Cargo.toml", "starky/src/proof.rs"]


def edit(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        if s.count(old) != 1:
            sys.exit("anchor not unique in %s (%d hits): %r" % (path, s.count(old), old[:60]))
        s = s.replace(old, new)
    open(path, "w").write(s)


GATE = '''        #[cfg(feature = "p2hot")]
        if crate::p2hot::applies::<F, C, D>(blinding) {
            // MI355X path (include/p2hot.h): iNTT + coset LDE + Poseidon leaf sponge + Merkle levels in one call; a blinded
            // commitment draws its salts on this side (F::rand_vec, as lde_values does) and hands them over
            return timed!(
                timing,
                "p2hot commit",
                crate::p2hot::commit::<F, C, D>(&crate::p2hot::%s(&%s), rate_bits, cap_height, %s, blinding)
            );
        }
'''


def main():
    top = tempfile.mkdtemp(prefix="p2hot_patch_")
    a, b = os.path.join(top, "a"), os.path.join(top, "b")
    for f in FILES:
        for d in (a, b):
            os.makedirs(os.path.dirname(os.path.join(d, f)), exist_ok=True)
            shutil.copy(os.path.join(REF, f), os.path.join(d, f))
    # ---- Cargo feature + build script + module
    edit(os.path.join(b, "plonky2/Cargo.toml"), [
        ('timing = ["std", "dep:web-time"]\n',
         'timing = ["std", "dep:web-time"]\n'
         '# MI355X hot path through libp2hot (include/p2hot.h); set P2HOT_LIB_DIR to the directory of libp2hot.so\n'
         'p2hot = ["std"]\n')])
    with open(os.path.join(b, "plonky2/build.rs"), "w") as f:
        f.write('''//! Link search path of libp2hot.so for the `p2hot` feature (the MI355X hot path, see src/p2hot.rs).
fn main() {
    println!("cargo:rerun-if-env-changed=P2HOT_LIB_DIR");
    if std::env::var_os("CARGO_FEATURE_P2HOT").is_some() {
        if let Some(dir) = std::env::var_os("P2HOT_LIB_DIR") {
            println!("cargo:rustc-link-search=native={}", dir.to_string_lossy());
            println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.to_string_lossy());
        }
    }
}
''')
    edit(os.path.join(b, "plonky2/src/lib.rs"), [
        ("pub mod iop;\n", "pub mod iop;\n#[cfg(feature = \"p2hot\")]\npub mod p2hot;\n")])
    shutil.copy(os.path.join(ROOT, "integration", "p2hot.rs"), os.path.join(b, "plonky2/src/p2hot.rs"))
    os.makedirs(os.path.join(b, "plonky2/examples"), exist_ok=True)
    shutil.copy(os.path.join(ROOT, "integration", "p2hot_dump_goldens.rs"), os.path.join(b, "plonky2/examples/p2hot_dump_goldens.rs"))
    # ---- PolynomialBatch
    edit(os.path.join(b, "plonky2/src/fri/oracle.rs"), [
        ('''        fft_root_table: Option<&FftRootTable<F>>,
    ) -> Self {
        let coeffs = timed!(
''', '''        fft_root_table: Option<&FftRootTable<F>>,
    ) -> Self {
''' + GATE % ("value_slices", "values", "true") + '''        let coeffs = timed!(
'''),
        ('''        fft_root_table: Option<&FftRootTable<F>>,
    ) -> Self {
        let degree = polynomials[0].len();
''', '''        fft_root_table: Option<&FftRootTable<F>>,
    ) -> Self {
''' + GATE % ("coeff_slices", "polynomials", "false") + '''        let degree = polynomials[0].len();
'''),
        ('''        let slice = &self.merkle_tree.leaves[index];
        &slice[..slice.len() - if self.blinding { SALT_SIZE } else { 0 }]
''', '''        // `get` instead of `leaves[index]`: with the `p2hot` feature the leaf matrix may live on the GPU
        let slice = self.merkle_tree.get(index);
        &slice[..slice.len() - if self.blinding { SALT_SIZE } else { 0 }]
'''),
        ('''        assert!(D > 1, "Not implemented for D=1.");
        let alpha = challenger.get_extension_challenge::<D>();
''', '''        assert!(D > 1, "Not implemented for D=1.");
        #[cfg(feature = "p2hot")]
        if let Some(proof) = timed!(
            timing,
            "p2hot prove_openings",
            crate::p2hot::prove_openings::<F, C, D>(
                instance,
                oracles,
                challenger,
                fri_params,
                final_poly_coeff_len,
                max_num_query_steps
            )
        ) {
            // alpha, final_poly, its LDE, the commit phase, the grind and the query rounds ran on the GPU
            return proof;
        }
        let alpha = challenger.get_extension_challenge::<D>();
''')])
    # ---- OpeningSet::new: the W Horner evaluations of a commitment at one point, on the coefficients the device already holds
    edit(os.path.join(b, "plonky2/src/plonk/proof.rs"), [
        ('''        let eval_commitment = |z: F::Extension, c: &PolynomialBatch<F, C, D>| {
            c.polynomials
''', '''        let eval_commitment = |z: F::Extension, c: &PolynomialBatch<F, C, D>| {
            #[cfg(feature = "p2hot")]
            if let Some(evals) = crate::p2hot::eval_commitment::<F, C, D>(z, c) {
                return evals;
            }
            c.polynomials
''')])
    # ---- starky: the feature is forwarded, and StarkOpeningSet::new (starky/src/proof.rs:237-242) evaluates its commitments the same
    # way (the trace and quotient commitments come from plonky2's PolynomialBatch::from_values / from_coeffs, so they carry handles)
    edit(os.path.join(b, "starky/Cargo.toml"), [
        ('timing = ["plonky2/timing"]\n',
         'timing = ["plonky2/timing"]\n'
         '# the MI355X hot path of the plonky2 crate (commitments, FRI, openings); see plonky2/src/p2hot.rs\n'
         'p2hot = ["std", "plonky2/p2hot"]\n')])
    edit(os.path.join(b, "starky/src/proof.rs"), [
        ('''        let eval_commitment = |z: F::Extension, c: &PolynomialBatch<F, C, D>| {
            c.polynomials
''', '''        let eval_commitment = |z: F::Extension, c: &PolynomialBatch<F, C, D>| {
            #[cfg(feature = "p2hot")]
            if let Some(evals) = plonky2::p2hot::eval_commitment::<F, C, D>(z, c) {
                return evals;
            }
            c.polynomials
''')])
    # ---- the quotient: the permutation argument's share of compute_quotient_polys on the GPU
    edit(os.path.join(b, "plonky2/src/plonk/prover.rs"), [
        ('''    common_data: &CommonCircuitData<F, D>,
) -> Vec<Vec<PolynomialValues<F>>> {
    (0..common_data.config.num_challenges)
''', '''    common_data: &CommonCircuitData<F, D>,
) -> Vec<Vec<PolynomialValues<F>>> {
    #[cfg(feature = "p2hot")]
    if let Some(polys) = crate::p2hot::all_wires_permutation_partial_products::<F, C, D>(
        witness,
        betas,
        gammas,
        prover_data,
        common_data,
    ) {
        // the row walk as a prefix-product scan on the GPU, for every (beta, gamma) at once
        return polys;
    }
    (0..common_data.config.num_challenges)
'''),
        ('''    alphas: &[F],
) -> Vec<PolynomialCoeffs<F>> {
    let num_challenges = common_data.config.num_challenges;

    let has_lookup = common_data.num_lookup_polys != 0;
''', '''    alphas: &[F],
) -> Vec<PolynomialCoeffs<F>> {
    #[cfg(feature = "p2hot")]
    if let Some(quotient_polys) = crate::p2hot::compute_quotient_polys::<F, C, D>(
        common_data,
        prover_data,
        public_inputs_hash,
        wires_commitment,
        zs_partial_products_and_lookup_commitment,
        betas,
        gammas,
        alphas,
    ) {
        // the permutation terms of the vanishing polynomial, / Z_H, coset_ifft ran on the GPU; the gate terms on the CPU
        return quotient_polys;
    }
    let num_challenges = common_data.config.num_challenges;

    let has_lookup = common_data.num_lookup_polys != 0;
''')])
    # ---- FRI commit phase
    edit(os.path.join(b, "plonky2/src/fri/prover.rs"), [
        ('''    max_num_query_steps: Option<usize>,
) -> FriCommitedTrees<F, C, D> {
    let mut trees = Vec::with_capacity(fri_params.reduction_arity_bits.len());
''', '''    max_num_query_steps: Option<usize>,
) -> FriCommitedTrees<F, C, D> {
    #[cfg(feature = "p2hot")]
    if crate::p2hot::applies::<F, C, D>(false) {
        // every round (tree, challenge, fold, coset NTT) on the GPU; `values` is recomputed there from `coeffs`
        return crate::p2hot::fri_committed_trees::<F, C, D>(
            &coeffs,
            challenger,
            fri_params,
            final_poly_coeff_len,
            max_num_query_steps,
        );
    }
    let mut trees = Vec::with_capacity(fri_params.reduction_arity_bits.len());
'''),
        # the grind: the GPU search returns the SMALLEST witness, rayon's find_any returns any.  With the feature on the CPU
        # body takes the smallest too, so that proofs are byte-identical whichever side ran (the bit-exact harness relies on it)
        ('''    let pow_witness = (0..=F::NEG_ONE.to_canonical_u64())
        .into_par_iter()
        .find_any(|&candidate| {
''', '''    let pow_candidates = (0..=F::NEG_ONE.to_canonical_u64()).into_par_iter();
    let pow_check = |&candidate: &u64| {
        {
'''),
        ('''            leading_zeros >= min_leading_zeros
        })
        .map(F::from_canonical_u64)
        .expect("Proof of work failed. This is highly unlikely!");
''', '''            leading_zeros >= min_leading_zeros
        }
    };
    // (without `parallel` the iterator is sequential and find_any already is the first match, maybe_rayon/src/lib.rs:254)
    #[cfg(all(feature = "p2hot", feature = "parallel"))]
    let pow_found = pow_candidates.find_first(pow_check); // the smallest witness, as libp2hot's grind returns it
    #[cfg(not(all(feature = "p2hot", feature = "parallel")))]
    let pow_found = pow_candidates.find_any(pow_check);
    let pow_witness = pow_found
        .map(F::from_canonical_u64)
        .expect("Proof of work failed. This is highly unlikely!");
'''),
        ('''fn fri_prover_query_rounds<
''', '''/// Test hook of the `p2hot` bit-exact harness (`crate::p2hot::tests`): `fri_committed_trees` is private to this module.
/// `lde_coeffs`: the zero-padded coefficients (length N); the values are recomputed as `fri_proof`'s caller does.
#[cfg(all(test, feature = "p2hot"))]
pub(crate) fn p2hot_fri_committed_trees_for_tests<
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F>,
    const D: usize,
>(
    lde_coeffs: &PolynomialCoeffs<F::Extension>,
    challenger: &mut Challenger<F, C::Hasher>,
    fri_params: &FriParams,
) -> FriCommitedTrees<F, C, D> {
    let lde_values = lde_coeffs.coset_fft(F::coset_shift().into());
    fri_committed_trees::<F, C, D>(lde_coeffs.clone(), lde_values, challenger, fri_params, None, None)
}

fn fri_prover_query_rounds<
''')])
    # ---- MerkleTree
    edit(os.path.join(b, "plonky2/src/hash/merkle_tree.rs"), [
        # equality is the VALUE (leaf rows, digests, cap).  Under the feature the leaf matrix / digests of a GPU-built tree live behind
        # `device` with the fields empty, so the derived field-by-field comparison would call such a tree unequal to its own
        # serialise / deserialise round trip (examples/square_root.rs:152 asserts that equality of CircuitData): written by hand
        ('''#[derive(Clone, Debug, Eq, PartialEq)]
pub struct MerkleTree<F: RichField, H: Hasher<F>> {
''', '''#[derive(Clone, Debug)]
#[cfg_attr(not(feature = "p2hot"), derive(Eq, PartialEq))]
pub struct MerkleTree<F: RichField, H: Hasher<F>> {
'''),
        ('''impl<F: RichField, H: Hasher<F>> Default for MerkleTree<F, H> {
''', '''/// Value equality with the `p2hot` feature: the leaf rows (through `get`: `leaves`, or the flat buffer / the GPU behind `device`),
/// the digests (the field, or the device's copy when the field was left empty) and the cap.
#[cfg(feature = "p2hot")]
impl<F: RichField, H: Hasher<F>> PartialEq for MerkleTree<F, H> {
    fn eq(&self, other: &Self) -> bool {
        if self.cap != other.cap || self.num_leaves() != other.num_leaves() {
            return false;
        }
        let digests_eq = match (self.digests.is_empty(), other.digests.is_empty()) {
            (false, false) => self.digests == other.digests,
            _ => self.digests_or_device() == other.digests_or_device(),
        };
        digests_eq && (0..self.num_leaves()).all(|i| self.get(i) == other.get(i))
    }
}
#[cfg(feature = "p2hot")]
impl<F: RichField, H: Hasher<F>> Eq for MerkleTree<F, H> {}

impl<F: RichField, H: Hasher<F>> Default for MerkleTree<F, H> {
'''),
        ('''    /// The Merkle cap.
    pub cap: MerkleCap<F, H>,
}
''', '''    /// The Merkle cap.
    pub cap: MerkleCap<F, H>,

    /// With the `p2hot` feature: the device-resident copy of this tree (LDE matrix = leaves, digests). When
    /// `leaves` / `digests` are empty (P2HOT_LEAVES=device), `get` / `prove` fetch from it.
    #[cfg(feature = "p2hot")]
    pub device: Option<std::sync::Arc<crate::p2hot::DeviceTree<F>>>,
}
'''),
        ('''            digests: Vec::new(),
            cap: MerkleCap::default(),
        }
''', '''            digests: Vec::new(),
            cap: MerkleCap::default(),
            #[cfg(feature = "p2hot")]
            device: None,
        }
'''),
        ('''            digests,
            cap: MerkleCap(cap),
        }
    }

    pub fn get(&self, i: usize) -> &[F] {
        &self.leaves[i]
    }
''', '''            digests,
            cap: MerkleCap(cap),
            #[cfg(feature = "p2hot")]
            device: None,
        }
    }

    pub fn get(&self, i: usize) -> &[F] {
        #[cfg(feature = "p2hot")]
        if self.leaves.is_empty() {
            if let Some(device) = &self.device {
                return device.row(i);
            }
        }
        &self.leaves[i]
    }
'''),
        ('''        let cap_height = log2_strict(self.cap.len());
        let siblings =
''', '''        let cap_height = log2_strict(self.cap.len());
        #[cfg(feature = "p2hot")]
        if self.digests.is_empty() {
            if let Some(device) = &self.device {
                return MerkleProof {
                    siblings: device.path::<H>(leaf_index),
                };
            }
        }
        let siblings =
'''),
        # `leaves` may be empty with the leaf matrix behind `device` (one flat host buffer, or the GPU): the leaf count comes
        # from `num_leaves`, which is `leaves.len()` in every other case
        ('''            merkle_tree_prove::<F, H>(leaf_index, self.leaves.len(), cap_height, &self.digests);
''', '''            merkle_tree_prove::<F, H>(leaf_index, self.num_leaves(), cap_height, &self.digests);
'''),
        ('''impl<F: RichField, H: Hasher<F>> MerkleTree<F, H> {
    pub fn new(leaves: Vec<Vec<F>>, cap_height: usize) -> Self {
''', '''impl<F: RichField, H: Hasher<F>> MerkleTree<F, H> {
    /// Number of leaves. With the `p2hot` feature the leaf matrix may live behind `device` (one flat host buffer, or
    /// the GPU) with `leaves` empty; otherwise this is `leaves.len()`.
    pub fn num_leaves(&self) -> usize {
        #[cfg(feature = "p2hot")]
        if self.leaves.is_empty() {
            if let Some(device) = &self.device {
                return device.num_leaves();
            }
        }
        self.leaves.len()
    }

    /// `digests`, or -- when the field was left empty because the digests stayed on the GPU (P2HOT_LEAVES=device) -- the
    /// device's copy of them, in the same (reference) layout
    #[cfg(feature = "p2hot")]
    pub fn digests_or_device(&self) -> alloc::borrow::Cow<'_, [H::Hash]> {
        if self.digests.is_empty() {
            if let Some(device) = &self.device {
                return alloc::borrow::Cow::Owned(device.digests::<H>());
            }
        }
        alloc::borrow::Cow::Borrowed(&self.digests)
    }

    pub fn new(leaves: Vec<Vec<F>>, cap_height: usize) -> Self {
''')])
    edit(os.path.join(b, "plonky2/src/util/serialization/mod.rs"), [
        # the serializer reads the rows through `get` / `num_leaves` (the same bytes: `get(i)` is `&leaves[i]` whenever `leaves`
        # is populated), so a tree whose leaf matrix is one flat buffer behind its device handle serialises unchanged
        ('''        self.write_usize(tree.leaves.len())?;
        for i in 0..tree.leaves.len() {
            self.write_usize(tree.leaves[i].len())?;
            self.write_field_vec(&tree.leaves[i])?;
        }
''', '''        self.write_usize(tree.num_leaves())?;
        for i in 0..tree.num_leaves() {
            self.write_usize(tree.get(i).len())?;
            self.write_field_vec(tree.get(i))?;
        }
'''),
        ('''        Ok(MerkleTree {
            leaves,
            digests,
            cap,
        })
''', '''        Ok(MerkleTree {
            leaves,
            digests,
            cap,
            #[cfg(feature = "p2hot")]
            device: None,
        })
''')])
    # ---- Challenger
    edit(os.path.join(b, "plonky2/src/iop/challenger.rs"), [
        ('''    pub fn compact(&mut self) -> H::Permutation {
''', '''    /// The transcript state for `crate::p2hot` (the sponge moves to the GPU for the FRI rounds and back).
    #[cfg(feature = "p2hot")]
    pub(crate) fn p2hot_parts(&mut self) -> (&mut H::Permutation, &mut Vec<F>, &mut Vec<F>) {
        (
            &mut self.sponge_state,
            &mut self.input_buffer,
            &mut self.output_buffer,
        )
    }

    pub fn compact(&mut self) -> H::Permutation {
''')])
    # ---- diff (a/ b/ prefixes, no timestamps, /dev/null for the two new files): `git apply` / `patch -p1` format
    r = subprocess.run(["diff", "-ruN", "a", "b"], capture_output=True, text=True, cwd=top)
    out = []
    for line in r.stdout.splitlines(keepends=True):
        if line.startswith("--- a/") or line.startswith("+++ b/"):
            name = line.split("\t")[0].rstrip("\n")
            if line.startswith("--- a/") and not os.path.exists(os.path.join(top, name[4:])):
                name = "--- /dev/null"
            out.append(name + "\n")
        else:
            out.append(line)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write("".join(out))
    shutil.rmtree(top)
    print("wrote", OUT, len(out), "lines")


if __name__ == "__main__":
    main()
