"""Pure-Python (big-int) twin of a few oracle functions, written independently of
oracle/p2oracle.c: O(n^2) polynomial evaluation, the 30-round *naive* Poseidon, closed-form
Merkle digest placement.  Small cases only; used to cross-check the C oracle.
Citations are into /root/reference.
"""
import os
import re

P = 0xFFFFFFFF00000001
G = 14293326489335486720          # field/src/goldilocks_field.rs:80 coset shift / generator
W32 = 7277203076849721926         # field/src/goldilocks_field.rs:87 2^32-th root of unity
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def root_of_unity(log_n):
    return pow(W32, 1 << (32 - log_n), P)


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def eval_poly(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % P
    return acc


def naive_ntt(coeffs):
    """out[i] = p(w^i) (field/src/fft.rs:259-282 evaluate_naive)"""
    n = len(coeffs)
    w = root_of_unity(n.bit_length() - 1)
    return [eval_poly(coeffs, pow(w, i, P)) for i in range(n)]


def naive_coset_lde_rows(coeffs, rate_bits):
    """leaves order: row L holds p(g * w_N^bitrev(L)) (plonky2/src/fri/oracle.rs:97-98, :142-147)"""
    n = len(coeffs)
    N = n << rate_bits
    lg = N.bit_length() - 1
    w = root_of_unity(lg)
    return [eval_poly(coeffs, G * pow(w, bitrev(L, lg), P) % P) for L in range(N)]


def _load_constants():
    t = open(os.path.join(ROOT, "oracle", "poseidon_constants.h")).read()
    out = {}
    for name, body in re.findall(r"(?:static const|P2_CONST_QUAL) uint64_t (\w+)\[\d+\] = \{(.*?)\};", t, re.S):
        out[name] = [int(x, 16) for x in re.findall(r"0x[0-9a-f]+", body)]
    return out


_C = _load_constants()
RC = _C["P2_POSEIDON_ALL_ROUND_CONSTANTS"]
CIRC = _C["P2_POSEIDON_MDS_CIRC"]
DIAG = _C["P2_POSEIDON_MDS_DIAG"]


def poseidon_naive(state):
    """plonky2/src/hash/poseidon.rs:781-801: ARK, S-box (all lanes in full rounds, lane 0 in
    rounds 4..25), circulant+diagonal MDS (poseidon.rs:180-199)."""
    s = [x % P for x in state]
    for rnd in range(30):
        s = [(s[i] + RC[12 * rnd + i]) % P for i in range(12)]
        if rnd < 4 or rnd >= 26:
            s = [pow(x, 7, P) for x in s]
        else:
            s[0] = pow(s[0], 7, P)
        s = [(sum(s[(i + r) % 12] * CIRC[i] for i in range(12)) + s[r] * DIAG[r]) % P for r in range(12)]
    return s


def hash_no_pad(xs):
    s = [0] * 12
    for off in range(0, len(xs), 8):
        chunk = xs[off:off + 8]
        s[:len(chunk)] = [x % P for x in chunk]
        s = poseidon_naive(s)
    return s[:4]


def hash_or_noop(xs):
    if len(xs) * 8 <= 32:
        return [x % P for x in xs] + [0] * (4 - len(xs))
    return hash_no_pad(xs)


def two_to_one(l, r):
    return poseidon_naive(list(l) + list(r) + [0] * 4)[:4]


def merkle_levels(leaves, cap_height):
    """level-major tree: levels[0] = leaf digests ... up to the cap level."""
    lv = [hash_or_noop(list(x)) for x in leaves]
    levels = [lv]
    while len(lv) > (1 << cap_height):
        lv = [two_to_one(lv[2 * i], lv[2 * i + 1]) for i in range(len(lv) // 2)]
        levels.append(lv)
    return levels


def digest_index(level, j):
    """closed form of the reference layout inside one cap subtree
    (plonky2/src/hash/merkle_tree.rs:50-57, :176-187)"""
    return 2 * (((j >> 1) << (level + 1)) + (1 << level) - 1) + (j & 1)


def ext_mul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def ext_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def ext_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = ext_mul(r, a)
        a = ext_mul(a, a)
        e >>= 1
    return r


def ext_inv(a):
    # norm = a0^2 - 7 a1^2
    nrm = (a[0] * a[0] - 7 * a[1] * a[1]) % P
    ni = pow(nrm, P - 2, P)
    return (a[0] * ni % P, (P - a[1]) * ni % P)


def ext_eval(coeffs, x):
    acc = (0, 0)
    for c in reversed(coeffs):
        acc = ext_add(ext_mul(acc, x), (int(c[0]), int(c[1])))
    return acc
