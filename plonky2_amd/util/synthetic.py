"""Synthetic inputs of the BASELINE configurations, shared by bench.py, the golden-vector
generator (tools/gen_golden_caps.py) and the full-size parity tests so that all three hash the same bytes.

  splitmix columns: col[c][i] = splitmix64(SEED ^ (c << 32) ^ i) mod P   (C2 / C3 / C5 trace columns)
  fibonacci trace:  row i = (F_i, F_{i+1}) mod P, x0 = 0, x1 = 1          (C4, starky/src/fibonacci_stark.rs:47-57)
"""
import numpy as np

P = 0xFFFFFFFF00000001
SEED = 0x9E3779B97F4A7C15


def splitmix_columns_numpy(col_begin, col_count, n):
    with np.errstate(over="ignore"):
        c = np.arange(col_begin, col_begin + col_count, dtype=np.uint64)[:, None]
        i = np.arange(n, dtype=np.uint64)[None, :]
        z = (c << np.uint64(32)) ^ i ^ np.uint64(SEED)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        return np.where(z >= np.uint64(P), z - np.uint64(P), z)


def splitmix_columns_torch(torch, device, col_begin, col_count, n):
    """the same columns generated on the device (int64 bit patterns)"""
    i64 = torch.int64

    def k(v):  # python int -> wrapped int64 constant
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(z, s):
        return (z >> s) & ((1 << (64 - s)) - 1)

    c = torch.arange(col_begin, col_begin + col_count, dtype=i64, device=device).unsqueeze(1)
    i = torch.arange(n, dtype=i64, device=device).unsqueeze(0)
    z = (c << 32) ^ i ^ k(SEED)
    z = z + k(0x9E3779B97F4A7C15)
    z = (z ^ lsr(z, 30)) * k(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * k(0x94D049BB133111EB)
    z = z ^ lsr(z, 31)
    # unsigned z >= P  <=>  signed z in [-(2^32 - 1), -1]; subtract P == add 2^32 - 1 (mod 2^64)
    z = torch.where((z < 0) & (z >= -(2**32 - 1)), z + (2**32 - 1), z)
    return z.contiguous()


def fibonacci_trace(log_n):
    """[2][2^log_n]: column 0 = F_i, column 1 = F_{i+1} (mod P).  Doubling formulas on vectors, not a python loop:
    (F_{2k}, F_{2k+1}) from (F_k, F_{k+1}) is only needed per index, so the table is built by repeated block doubling
    with the 2x2 matrix power M^(2^j) applied to the rows already known."""
    n = 1 << log_n
    f0 = np.zeros(n, dtype=object)
    f1 = np.zeros(n, dtype=object)
    f0[0], f1[0] = 0, 1
    # M^m = [[F_{m-1}, F_m], [F_m, F_{m+1}]]; rows [m, 2m) = rows [0, m) advanced by m steps
    a, b = 1, 1  # (F_m, F_{m+1}) for m = 1
    m = 1
    while m < n:
        fm1 = (b - a) % P  # F_{m-1}
        # F_{i+m} = F_i * F_{m-1} + F_{i+1} * F_m ; F_{i+m+1} = F_i * F_m + F_{i+1} * F_{m+1}
        x0, x1 = f0[:m], f1[:m]
        f0[m:2 * m] = (x0 * fm1 + x1 * a) % P
        f1[m:2 * m] = (x0 * a + x1 * b) % P
        # (F_{2m}, F_{2m+1})
        a, b = (a * fm1 + b * a) % P, (a * a + b * b) % P
        m *= 2
    return np.stack([f0.astype(np.uint64), f1.astype(np.uint64)])


# ------------------------------------------------------------------ the per-proof paths of bench.py (`per_proof_path_*` lines)
# One definition shared by bench.py, tools/gen_golden_path.py (the CPU oracle's bytes for exactly these inputs,
# tests/golden/path_goldens.json) and tests/test_gpu_fullsize.py, so that all three prove the same instance.
GENERATOR = 14293326489335486720  # MULTIPLICATIVE_GROUP_GENERATOR, field/src/goldilocks_field.rs:80


def plonk_path_instance(log_n):
    """One standard_recursion_config proof of 2^log_n gates on the reference's FRI instance (get_fri_instance,
    plonk/circuit_data.rs:530-548, :578-664): oracles [constants_sigmas (4 + 80), wires (135), Zs + partial products (20),
    quotient chunks (16)], every polynomial at zeta, the num_challenges = 2 Z polynomials at the second point."""
    widths = (84, 135, 20, 16)
    return {
        "kind": "plonk", "log_n": log_n, "rate_bits": 3, "cap_height": 4, "num_queries": 28, "pow_bits": 16,
        "arity": {20: [4, 4, 4, 4], 16: [4, 4, 4], 12: [4, 4]}[log_n],      # reduction_strategies.rs:41-52
        "wires_seed": 0, "wires_width": 135, "cs_seed": 1000, "cs_width": 84, "num_constants": 4, "num_routed": 80,
        "k_is": [pow(GENERATOR, j, P) for j in range(80)],                 # get_unique_coset_shifts, field/src/cosets.rs:9-24
        "quotient_degree_factor": 8, "betas": [3, 5], "gammas": [11, 13], "alphas": [17, 19],
        "widths": widths, "transcript_seed": list(range(8)),
        "batch_zeta": [(oi, pi) for oi, W in enumerate(widths) for pi in range(W)],
        "batch_next": [(2, pi) for pi in range(2)],
    }


def starky_path_instance(log_n):
    """C4: one starky proof of the 2-column Fibonacci trace (StarkConfig::standard_fast_config: rate 1/2, cap 4, 84 queries,
    PoW 16 bits): trace + quotient at zeta, trace at the second point (starky/src/stark.rs:101-156)."""
    return {
        "kind": "starky", "log_n": log_n, "rate_bits": 1, "cap_height": 4, "num_queries": 84, "pow_bits": 16, "arity": [4, 4, 4, 4],
        "quotient_seed": 3000, "widths": (2, 2), "transcript_seed": list(range(8)),
        "batch_zeta": [(0, 0), (0, 1), (1, 0), (1, 1)], "batch_next": [(0, 0), (0, 1)],
    }


def path_instance(name):
    return {"per_proof_path_k20": lambda: plonk_path_instance(20), "per_proof_path_k12": lambda: plonk_path_instance(12),
            "per_proof_path_starky_k22": lambda: starky_path_instance(22)}[name]()


POWER_OF_TWO_GENERATOR = 7277203076849721926  # field/src/goldilocks_field.rs:87 (order 2^32)


def primitive_root_of_unity(log_n):
    """Field::primitive_root_of_unity (field/src/types.rs:268-272): POWER_OF_TWO_GENERATOR ^ (2^(32 - log_n))"""
    return pow(POWER_OF_TWO_GENERATOR, 1 << (32 - log_n), P)


def second_point(zeta, log_n):
    """g * zeta with g = primitive_root_of_unity(degree_bits) embedded in the extension field: the point the reference opens
    the Z polynomials (plonk/circuit_data.rs:537-539) and the starky trace (starky/src/stark.rs:101-156) at besides zeta"""
    g = primitive_root_of_unity(log_n)
    return [(int(zeta[0]) * g) % P, (int(zeta[1]) * g) % P]
