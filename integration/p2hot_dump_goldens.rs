//! Golden dumper of the GPU repository's parity protocol: runs the REFERENCE prover's own
//! `PolynomialBatch::from_values` / `from_coeffs` and FRI commit phase on the synthetic inputs of
//! `plonky2_amd/util/synthetic.py` and writes what they produce in the JSON schema of
//! `tests/golden/commit_caps.json` (cap, SHA-256 of the coefficient matrix, of `merkle_tree.digests`, of the LDE
//! matrix in the device's column-major committed order) plus the FRI caps / final_poly of a synthetic codeword.
//!
//! Added to the reference tree as `plonky2/examples/p2hot_dump_goldens.rs` by `integration/plonky2_p2hot.patch`.
//! Runs on any box with cargo; no GPU, no feature flag:
//!     cargo run --release --example p2hot_dump_goldens -- --out reference_run.json [--only c2_wires,small_a] [--big]
//! Copy the file to `tests/golden/reference_run.json` of the GPU repository: `pytest tests/test_oracle.py -k reference_run`
//! then compares the CPU oracle (and, through the existing goldens, every GPU result) with the real reference's bytes.
//! `--big` adds the 2^20 / 2^22-row shapes (BASELINE C3 / C4: tens of GB of RAM, minutes).
use std::io::Write as _;

use plonky2::field::extension::Extendable;
use plonky2::field::goldilocks_field::GoldilocksField;
use plonky2::field::polynomial::{PolynomialCoeffs, PolynomialValues};
use plonky2::field::types::{Field, PrimeField64};
use plonky2::fri::oracle::PolynomialBatch;
use plonky2::fri::prover::fri_proof;
use plonky2::fri::reduction_strategies::FriReductionStrategy;
use plonky2::fri::{FriConfig, FriParams};
use plonky2::hash::hash_types::HashOut;
use plonky2::iop::challenger::Challenger;
use plonky2::plonk::config::{GenericConfig, PoseidonGoldilocksConfig};
use plonky2::util::timing::TimingTree;

const D: usize = 2;
type C = PoseidonGoldilocksConfig;
type F = GoldilocksField;
type H = <C as GenericConfig<D>>::Hasher;
type FE = <F as Extendable<D>>::Extension;

const P: u64 = 0xFFFF_FFFF_0000_0001;
const SEED: u64 = 0x9E37_79B9_7F4A_7C15;

/// plonky2_amd/util/synthetic.py splitmix_columns_numpy: col[c][i] = splitmix64(SEED ^ (c << 32) ^ i) mod P
fn splitmix(c: u64, i: u64) -> u64 {
    let mut z = ((c << 32) ^ i ^ SEED).wrapping_add(0x9E37_79B9_7F4A_7C15);
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58_476D_1CE4_E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D0_49BB_1331_11EB);
    z ^= z >> 31;
    if z >= P {
        z - P
    } else {
        z
    }
}

fn splitmix_columns(w: usize, n: usize) -> Vec<Vec<F>> {
    (0..w).map(|c| (0..n).map(|i| F::from_canonical_u64(splitmix(c as u64, i as u64))).collect()).collect()
}

/// synthetic.py fibonacci_trace: column 0 = F_i, column 1 = F_{i+1} (starky/src/fibonacci_stark.rs:47-57 with x0 = 0, x1 = 1)
fn fibonacci_columns(n: usize) -> Vec<Vec<F>> {
    let (mut a, mut b) = (F::ZERO, F::ONE);
    let (mut c0, mut c1) = (Vec::with_capacity(n), Vec::with_capacity(n));
    for _ in 0..n {
        c0.push(a);
        c1.push(b);
        let t = a + b;
        a = b;
        b = t;
    }
    vec![c0, c1]
}

// ---- SHA-256 (FIPS 180-4), streaming: the crate has no SHA-2 dependency and the goldens are SHA-256
struct Sha256 {
    h: [u32; 8],
    buf: Vec<u8>,
    len: u64,
}
const K: [u32; 64] = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2,
];
impl Sha256 {
    fn new() -> Self {
        Sha256 { h: [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], buf: Vec::new(), len: 0 }
    }
    fn block(&mut self, b: &[u8]) {
        let mut w = [0u32; 64];
        for i in 0..16 {
            w[i] = u32::from_be_bytes([b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]]);
        }
        for i in 16..64 {
            let s0 = w[i - 15].rotate_right(7) ^ w[i - 15].rotate_right(18) ^ (w[i - 15] >> 3);
            let s1 = w[i - 2].rotate_right(17) ^ w[i - 2].rotate_right(19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16].wrapping_add(s0).wrapping_add(w[i - 7]).wrapping_add(s1);
        }
        let mut v = self.h;
        for i in 0..64 {
            let s1 = v[4].rotate_right(6) ^ v[4].rotate_right(11) ^ v[4].rotate_right(25);
            let ch = (v[4] & v[5]) ^ (!v[4] & v[6]);
            let t1 = v[7].wrapping_add(s1).wrapping_add(ch).wrapping_add(K[i]).wrapping_add(w[i]);
            let s0 = v[0].rotate_right(2) ^ v[0].rotate_right(13) ^ v[0].rotate_right(22);
            let maj = (v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]);
            let t2 = s0.wrapping_add(maj);
            v = [t1.wrapping_add(t2), v[0], v[1], v[2], v[3].wrapping_add(t1), v[4], v[5], v[6]];
        }
        for i in 0..8 {
            self.h[i] = self.h[i].wrapping_add(v[i]);
        }
    }
    fn update(&mut self, data: &[u8]) {
        self.len += data.len() as u64;
        let mut data = data;
        if !self.buf.is_empty() {
            let take = (64 - self.buf.len()).min(data.len());
            self.buf.extend_from_slice(&data[..take]);
            data = &data[take..];
            if self.buf.len() == 64 {
                let b = core::mem::take(&mut self.buf);
                self.block(&b);
            }
        }
        while data.len() >= 64 {
            let (b, rest) = data.split_at(64);
            self.block(b);
            data = rest;
        }
        self.buf.extend_from_slice(data);
    }
    fn update_words(&mut self, words: impl Iterator<Item = u64>) {
        let mut chunk = Vec::with_capacity(8 * 4096);
        for w in words {
            chunk.extend_from_slice(&w.to_le_bytes()); // u64 little endian, canonical: numpy's tobytes() of the goldens
            if chunk.len() == 8 * 4096 {
                self.update(&chunk);
                chunk.clear();
            }
        }
        self.update(&chunk);
    }
    fn hex(mut self) -> String {
        let bits = self.len * 8;
        self.update(&[0x80]);
        while self.buf.len() != 56 {
            self.update(&[0]);
        }
        self.len = 0;
        self.update(&bits.to_be_bytes());
        self.h.iter().map(|x| format!("{x:08x}")).collect()
    }
}

fn hash_words(h: &HashOut<F>) -> [u64; 4] {
    [h.elements[0].to_canonical_u64(), h.elements[1].to_canonical_u64(), h.elements[2].to_canonical_u64(), h.elements[3].to_canonical_u64()]
}

fn cap_json(cap: &[HashOut<F>]) -> String {
    let rows: Vec<String> = cap.iter().map(|h| format!("{:?}", hash_words(h))).collect();
    format!("[{}]", rows.join(", "))
}

/// one commitment in the schema of tests/golden/commit_caps.json
fn commit_record(w: usize, log_n: usize, rate_bits: usize, cap_height: usize, is_values: bool, input: &str) -> String {
    let n = 1usize << log_n;
    let cols = if input == "fibonacci" { fibonacci_columns(n) } else { splitmix_columns(w, n) };
    let batch: PolynomialBatch<F, C, D> = if is_values {
        PolynomialBatch::from_values(cols.into_iter().map(PolynomialValues::new).collect(), rate_bits, false, cap_height, &mut TimingTree::default(), None)
    } else {
        PolynomialBatch::from_coeffs(cols.into_iter().map(PolynomialCoeffs::new).collect(), rate_bits, false, cap_height, &mut TimingTree::default(), None)
    };
    let mut sc = Sha256::new(); // coefficient matrix [W][n]
    for p in &batch.polynomials {
        sc.update_words(p.coeffs.iter().map(|x| x.to_canonical_u64()));
    }
    let mut sd = Sha256::new(); // merkle_tree.digests in the reference layout (hash/merkle_tree.rs:50-57)
    sd.update_words(batch.merkle_tree.digests.iter().flat_map(|h| hash_words(h).into_iter()));
    let mut sl = Sha256::new(); // LDE matrix column-major [W][N], rows in committed (bit-reversed) order = `leaves` transposed
    for c in 0..w {
        sl.update_words(batch.merkle_tree.leaves.iter().map(|row| row[c].to_canonical_u64()));
    }
    format!(
        "{{\"W\": {w}, \"log_n\": {log_n}, \"rate_bits\": {rate_bits}, \"cap_height\": {cap_height}, \"is_values\": {is_values}, \"input\": \"{input}\", \"cap\": {}, \
         \"sha256_coeffs\": \"{}\", \"sha256_digests\": \"{}\", \"sha256_lde\": \"{}\"}}",
        cap_json(&batch.merkle_tree.cap.0),
        sc.hex(),
        sd.hex(),
        sl.hex()
    )
}

/// FRI commit phase + grind of the reference (`fri_proof` with no initial trees) on a synthetic extension polynomial:
/// coefficient i = (splitmix(0, i), splitmix(1, i)), transcript seeded by observing splitmix(2, 0..8)
fn fri_record(log_n: usize, rate_bits: usize, cap_height: usize, arity: &[usize], pow_bits: u32) -> String {
    let n = 1usize << log_n;
    let mut coeffs: Vec<FE> = (0..n)
        .map(|i| FE::from_basefield_array([F::from_canonical_u64(splitmix(0, i as u64)), F::from_canonical_u64(splitmix(1, i as u64))]))
        .collect();
    coeffs.resize(n << rate_bits, FE::ZERO);
    let lde_coeffs = PolynomialCoeffs::new(coeffs);
    let lde_values = lde_coeffs.coset_fft(F::coset_shift().into());
    let fri_params = FriParams {
        config: FriConfig { rate_bits, cap_height, proof_of_work_bits: pow_bits, reduction_strategy: FriReductionStrategy::Fixed(arity.to_vec()), num_query_rounds: 0 },
        hiding: false,
        degree_bits: log_n,
        reduction_arity_bits: arity.to_vec(),
    };
    let mut challenger = Challenger::<F, H>::new();
    let seed: Vec<F> = (0..8).map(|i| F::from_canonical_u64(splitmix(2, i))).collect();
    challenger.observe_elements(&seed);
    let proof = fri_proof::<F, C, D>(&[], lde_coeffs, lde_values, &mut challenger, &fri_params, None, None, &mut TimingTree::default());
    let caps: Vec<String> = proof.commit_phase_merkle_caps.iter().map(|c| cap_json(&c.0)).collect();
    let fin: Vec<String> = proof
        .final_poly
        .coeffs
        .iter()
        .map(|e| {
            let a = e.to_basefield_array();
            format!("[{}, {}]", a[0].to_canonical_u64(), a[1].to_canonical_u64())
        })
        .collect();
    format!(
        "{{\"log_n\": {log_n}, \"rate_bits\": {rate_bits}, \"cap_height\": {cap_height}, \"arity_bits\": {arity:?}, \"proof_of_work_bits\": {pow_bits}, \
         \"commit_phase_merkle_caps\": [{}], \"final_poly\": [{}], \"pow_witness\": {}, \"pow_witness_is_smallest\": {}}}",
        caps.join(", "),
        fin.join(", "),
        proof.pow_witness.to_canonical_u64(),
        cfg!(feature = "p2hot") // with the feature the grind takes the smallest witness (find_first); plain builds take any
    )
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    let arg = |name: &str| args.iter().position(|a| a == name).and_then(|i| args.get(i + 1).cloned());
    let out = arg("--out").unwrap_or_else(|| "reference_run.json".to_string());
    let only: Option<Vec<String>> = arg("--only").map(|s| s.split(',').map(|x| x.to_string()).collect());
    let big = args.iter().any(|a| a == "--big");
    // name -> (W, log_n, rate_bits, cap_height, is_values, input): the names and shapes of tools/gen_golden_caps.py, plus small ones
    let mut shapes: Vec<(&str, usize, usize, usize, usize, bool, &str)> = vec![
        ("small_values", 7, 5, 3, 4, true, "splitmix"),
        ("small_coeffs", 3, 8, 1, 0, false, "splitmix"),
        ("small_wide", 135, 12, 3, 4, true, "splitmix"),
        ("small_constants_sigmas", 84, 12, 3, 4, true, "splitmix"),
        ("small_fibonacci", 2, 10, 1, 4, true, "fibonacci"),
        ("c2_wires", 135, 16, 3, 4, true, "splitmix"),
    ];
    if big {
        shapes.extend([
            ("c3_wires", 135, 20, 3, 4, true, "splitmix"),
            ("c3_constants_sigmas", 84, 20, 3, 4, true, "splitmix"),
            ("c3_zs_partial_products", 20, 20, 3, 4, true, "splitmix"),
            ("c3_quotient_chunks", 16, 20, 3, 4, false, "splitmix"),
            ("c4_fibonacci_trace", 2, 22, 1, 4, true, "fibonacci"),
        ]);
    }
    let mut items = vec![format!(
        "\"source\": \"0xPolygonZero/plonky2 PolynomialBatch::from_values / from_coeffs and fri_proof on plonky2_amd/util/synthetic.py inputs; written by examples/p2hot_dump_goldens.rs (p2hot feature {})\"",
        if cfg!(feature = "p2hot") { "ON: make sure P2HOT_DISABLE=1 was set, or these are GPU results" } else { "off: the unmodified CPU prover" }
    )];
    for (name, w, log_n, rb, cap, is_values, input) in shapes {
        if only.as_ref().map_or(false, |o| !o.iter().any(|x| x == name)) {
            continue;
        }
        eprintln!("{name}: W = {w}, 2^{log_n} rows, rate 1/{}", 1 << rb);
        items.push(format!("\"{name}\": {}", commit_record(w, log_n, rb, cap, is_values, input)));
    }
    for (name, log_n, rb, cap, arity, pow) in [("fri_small", 8usize, 3usize, 2usize, vec![2usize, 1], 4u32), ("fri_plonky2_like", 12, 3, 4, vec![4, 4], 8), ("fri_starky_like", 10, 1, 3, vec![1, 2, 1], 6)] {
        if only.as_ref().map_or(false, |o| !o.iter().any(|x| x == name)) {
            continue;
        }
        eprintln!("{name}");
        items.push(format!("\"{name}\": {}", fri_record(log_n, rb, cap, &arity, pow)));
    }
    let mut f = std::fs::File::create(&out).expect("cannot create the output file");
    writeln!(f, "{{\n  {}\n}}", items.join(",\n  ")).unwrap();
    eprintln!("wrote {out}");
}
