#!/usr/bin/env python3
"""Generate the Poseidon (width 12, Goldilocks) parameter headers.

The 360 round constants and the circulant/diagonal MDS rows are *parameters* of
the hash function plonky2 uses (reference: plonky2/src/hash/poseidon.rs:59-157
`ALL_ROUND_CONSTANTS`, plonky2/src/hash/poseidon_goldilocks.rs:24-25
`MDS_MATRIX_CIRC/DIAG`).  They cannot be re-derived (they came out of an RNG in
the upstream `hash-constants` repo), so this tool reads the numeric values from
the reference tree, and DERIVES everything else itself:

  * FAST_PARTIAL_FIRST_ROUND_CONSTANT, FAST_PARTIAL_ROUND_CONSTANTS,
    FAST_PARTIAL_ROUND_VS, FAST_PARTIAL_ROUND_W_HATS,
    FAST_PARTIAL_ROUND_INITIAL_MATRIX  (poseidon_goldilocks.rs:27-215)

from the round constants and the MDS matrix by the "equivalent constants /
equivalent matrices" transformation of the Poseidon paper (appendix B), then
checks the derived tables against the reference's own tables (when the
reference tree is present) so that the derivation itself is pinned.

Run in the build container only:  python tools/gen_poseidon_constants.py
Outputs (identical content, two consumers that must not depend on each other):
    oracle/poseidon_constants.h          (CPU oracle, test infrastructure)
    plonky2_amd/csrc/poseidon_constants.h (HIP product path)
"""
import os
import re
import sys

P = 0xFFFFFFFF00000001
W = 12
N_FULL = 8
N_PART = 22
HALF_FULL = 4
REF = os.environ.get("P2_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _array_after(text, name):
    """All integer literals of the first bracketed initializer after `name`."""
    i = text.index(name)
    j = text.index("=", i)
    depth = 0
    k = j
    start = None
    while True:
        ch = text[k]
        if ch == "[":
            if depth == 0:
                start = k
            depth += 1
        elif ch == "]":
            depth -= 1
            if depth == 0:
                break
        k += 1
    body = text[start:k + 1]
    body = re.sub(r"//[^\n]*", "", body)
    return [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\b\d+\b", body)]


def read_reference():
    with open(os.path.join(REF, "plonky2/src/hash/poseidon.rs")) as f:
        t1 = f.read()
    with open(os.path.join(REF, "plonky2/src/hash/poseidon_goldilocks.rs")) as f:
        t2 = f.read()
    # restrict to the width-12 impl (first impl block in the file)
    rc = _array_after(t1, "pub const ALL_ROUND_CONSTANTS")
    # the type annotation `[u64; MAX_WIDTH * N_ROUNDS]` precedes '=' so it is skipped
    assert len(rc) == W * (N_FULL + N_PART), len(rc)
    circ = _array_after(t2, "const MDS_MATRIX_CIRC")
    diag = _array_after(t2, "const MDS_MATRIX_DIAG")
    ref = {
        "first": _array_after(t2, "const FAST_PARTIAL_FIRST_ROUND_CONSTANT"),
        "rc": _array_after(t2, "const FAST_PARTIAL_ROUND_CONSTANTS"),
        "vs": _array_after(t2, "const FAST_PARTIAL_ROUND_VS"),
        "w_hats": _array_after(t2, "const FAST_PARTIAL_ROUND_W_HATS"),
        "init": _array_after(t2, "const FAST_PARTIAL_ROUND_INITIAL_MATRIX"),
    }
    return rc, circ[:W], diag[:W], ref


# ---------- tiny modular linear algebra ----------
def mat_mul(a, b):
    n, m, k = len(a), len(b[0]), len(b)
    return [[sum(a[i][t] * b[t][j] for t in range(k)) % P for j in range(m)] for i in range(n)]


def mat_vec(a, v):
    return [sum(a[i][j] * v[j] for j in range(len(v))) % P for i in range(len(a))]


def mat_inv(a):
    n = len(a)
    m = [row[:] + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(a)]
    for c in range(n):
        piv = next(r for r in range(c, n) if m[r][c] % P)
        m[c], m[piv] = m[piv], m[c]
        inv = pow(m[c][c], P - 2, P)
        m[c] = [x * inv % P for x in m[c]]
        for r in range(n):
            if r != c and m[r][c]:
                f = m[r][c]
                m[r] = [(x - f * y) % P for x, y in zip(m[r], m[c])]
    return [row[n:] for row in m]


def transpose(a):
    return [list(r) for r in zip(*a)]


def derive(rc, circ, diag):
    # new_state = M * state, M[r][c] = CIRC[(c - r) mod 12] + [r == c] DIAG[r]
    # (poseidon.rs:180-199 mds_row_shf: res += v[(i + r) % 12] * CIRC[i])
    M = [[(circ[(c - r) % W] + (diag[r] if r == c else 0)) % P for c in range(W)] for r in range(W)]
    Minv = mat_inv(M)
    # ---- equivalent round constants (move constants backwards through M) ----
    consts = [rc[W * r: W * (r + 1)] for r in range(N_FULL + N_PART)]
    first = HALF_FULL
    last = HALF_FULL + N_PART - 1
    scal = [0] * N_PART
    for r in range(last, first, -1):
        inv = mat_vec(Minv, consts[r])
        scal[r - 1 - first] = inv[0]
        for i in range(1, W):
            consts[r - 1][i] = (consts[r - 1][i] + inv[i]) % P
    first_round_constant = consts[first]
    # scal[N_PART-1] stays 0 (nothing is moved out of the following full round)
    # ---- equivalent matrices: A = Sp * [[1,0],[0,Ahat]], pushed backwards ----
    vs = [None] * N_PART      # column part  (reference name: FAST_PARTIAL_ROUND_VS)
    w_hats = [None] * N_PART  # row part     (reference name: FAST_PARTIAL_ROUND_W_HATS)
    A = M
    for r in range(N_PART - 1, -1, -1):
        Ahat = [row[1:] for row in A[1:]]
        v_row = A[0][1:]
        w_col = [A[i][0] for i in range(1, W)]
        Ahat_inv = mat_inv(Ahat)
        # row part v'^T = v^T * Ahat^{-1}
        w_hats[r] = [sum(v_row[t] * Ahat_inv[t][j] for t in range(W - 1)) % P for j in range(W - 1)]
        vs[r] = w_col
        Mp = [[1] + [0] * (W - 1)] + [[0] + Ahat[i] for i in range(W - 1)]
        A = mat_mul(Mp, M)  # the earlier round applies M, then the pushed-back block
        last_block = Ahat
    # stored transposed: result[c] += state[r] * INIT[r-1][c-1] (poseidon.rs:415-441)
    init = transpose(last_block)
    return first_round_constant, scal, vs, w_hats, init


def flat(a):
    out = []
    for x in a:
        if isinstance(x, list):
            out.extend(flat(x))
        else:
            out.append(x)
    return out


def pushed_constants(rc, circ, diag):
    """Dense-MDS form of the partial rounds with the constant layer reduced to one scalar: the passive part of
    round r's constants (positions 1..11 see no S-box) is pushed forward through the MDS into round r+1,
        c'_4 = c_4;  scalar_r = c'_r[0];  c'_{r+1} = c_{r+1} + M * (c'_r with position 0 cleared),  r = 4..25,
    so rounds 4..25 add only scalar_r to word 0 and round 26 adds c'_26.  Same function as
    poseidon.rs:781-801 (checked against the reference KATs by the tests)."""
    M = [[(circ[(c - r) % W] + (diag[r] if r == c else 0)) % P for c in range(W)] for r in range(W)]
    half = N_FULL // 2
    fused = [rc[W * r:W * (r + 1)] for r in range(N_FULL + N_PART)]
    cur = fused[half][:]
    for r in range(half, half + N_PART):
        scalar, passive = cur[0], [0] + cur[1:]
        fused[r] = [scalar] + [0] * (W - 1)
        nxt = mat_vec(M, passive)
        cur = [(a + b) % P for a, b in zip(rc[W * (r + 1):W * (r + 2)], nxt)]
    fused[half + N_PART] = cur
    return [v for row in fused for v in row]


def mds_power_tables(circ, diag):
    """Integer powers of the MDS matrix for the batched partial rounds (three rounds per dense pass):
        y3 = M^3 y + d0 * M^3 e0 + d1 * M^2 e0 + d2 * M e0,   d_k = sbox(y_k[0] + c_k) - y_k[0],
        y1[0] = (M y)[0] + d0 * M[0][0],   y2[0] = (M^2 y)[0] + d0 * (M^2)[0][0] + d1 * M[0][0].
    All entries are exact small integers (M^3 < 2^21, row sums < 2^25), so a row is still two 32x32+64
    multiply-add chains.  Returns (row0 of M, row0 of M^2, M^3 row-major, columns 0 of M, M^2, M^3)."""
    M = [[circ[(c - r) % W] + (diag[r] if r == c else 0) for c in range(W)] for r in range(W)]
    mul = lambda a, b: [[sum(a[i][t] * b[t][j] for t in range(W)) for j in range(W)] for i in range(W)]
    M2 = mul(M, M)
    M3 = mul(M2, M)
    assert max(max(r) for r in M3) < 1 << 21 and max(sum(r) for r in M3) < 1 << 25
    col = lambda A: [A[i][0] for i in range(W)]
    return M[0], M2[0], [v for r in M3 for v in r], col(M) + col(M2) + col(M3)


def emit(path, rc, circ, diag, first, scal, vs, w_hats, init):
    def arr(name, vals, per=4):
        s = "P2_CONST_QUAL uint64_t %s[%d] = {\n" % (name, len(vals))
        for i in range(0, len(vals), per):
            s += "    " + ", ".join("0x%016xULL" % v for v in vals[i:i + per]) + ",\n"
        return s + "};\n\n"
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_poseidon_constants.py -- do not edit.\n"
                "// Poseidon width-12 x^7 parameters over Goldilocks as used by plonky2\n"
                "// (values: plonky2/src/hash/poseidon.rs:59-157, poseidon_goldilocks.rs:24-25;\n"
                "//  fast-partial-round tables derived by the tool and checked against\n"
                "//  poseidon_goldilocks.rs:27-215).\n"
                "#pragma once\n#include <stdint.h>\n"
                "// P2_CONST_QUAL: storage qualifier (the HIP build defines it as __constant__)\n"
                "#ifndef P2_CONST_QUAL\n#define P2_CONST_QUAL static const\n#endif\n"
                "// P2_LITERAL_QUAL: tables whose entries must be visible to the compiler as literals (the HIP build\n"
                "// defines it as static constexpr: they become s_mov_b32 immediates, not constant-memory loads)\n"
                "#ifndef P2_LITERAL_QUAL\n#define P2_LITERAL_QUAL static const\n#endif\n\n")
        f.write(arr("P2_POSEIDON_ALL_ROUND_CONSTANTS", rc))
        f.write("// the same constants with every word split into {lo32, hi32} (two u64 words): the device\n"
                "// MDS rows start their two 32x32+64 multiply-add chains from them\n")
        f.write(arr("P2_POSEIDON_ALL_ROUND_CONSTANTS_SPLIT", [h for v in rc for h in (v & 0xFFFFFFFF, v >> 32)]))
        f.write("// round constants with the passive part of the partial rounds pushed forward through the MDS\n"
                "// (tools/gen_poseidon_constants.py pushed_constants): rounds 4..25 keep only word 0, round 26 absorbs the rest;\n"
                "// split {lo32, hi32} like the table above\n")
        f.write(arr("P2_POSEIDON_PUSHED_ROUND_CONSTANTS", pushed_constants(rc, circ, diag)))
        f.write(arr("P2_POSEIDON_PUSHED_ROUND_CONSTANTS_SPLIT",
                    [h for v in pushed_constants(rc, circ, diag) for h in (v & 0xFFFFFFFF, v >> 32)]))
        def arr32(name, vals, per=12):
            t = "P2_LITERAL_QUAL uint32_t %s[%d] = {\n" % (name, len(vals))
            for i in range(0, len(vals), per):
                t += "    " + ", ".join("%du" % v for v in vals[i:i + per]) + ",\n"
            return t + "};\n\n"
        m1r0, m2r0, m3, cols = mds_power_tables(circ, diag)
        f.write("// integer powers of the MDS matrix for the batched partial rounds (tools/gen_poseidon_constants.py\n"
                "// mds_power_tables): row 0 of M and M^2, M^3 row-major, then column 0 of M, M^2, M^3\n")
        f.write(arr32("P2_POSEIDON_M1_ROW0", m1r0))
        f.write(arr32("P2_POSEIDON_M2_ROW0", m2r0))
        f.write(arr32("P2_POSEIDON_M3", m3))
        f.write(arr32("P2_POSEIDON_MCOL0", cols))
        f.write(arr("P2_POSEIDON_MDS_CIRC", circ, 12))
        f.write(arr("P2_POSEIDON_MDS_DIAG", diag, 12))
        f.write(arr("P2_POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT", first))
        f.write(arr("P2_POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS", scal))
        f.write("// [round][i-1]: result[i] = state[i] + state[0] * VS[round][i-1]\n")
        f.write(arr("P2_POSEIDON_FAST_PARTIAL_ROUND_VS", flat(vs), 11))
        f.write("// [round][i-1]: result[0] = M00*state[0] + sum_i state[i] * W_HATS[round][i-1]\n")
        f.write(arr("P2_POSEIDON_FAST_PARTIAL_ROUND_W_HATS", flat(w_hats), 11))
        f.write("// [r-1][c-1]: result[c] += state[r] * INITIAL_MATRIX[r-1][c-1]\n")
        f.write(arr("P2_POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX", flat(init), 11))


def main():
    rc, circ, diag, ref = read_reference()
    first, scal, vs, w_hats, init = derive(rc, circ, diag)
    checks = [("first", first), ("rc", scal), ("vs", flat(vs)), ("w_hats", flat(w_hats)), ("init", flat(init))]
    for name, mine in checks:
        assert mine == ref[name], "derived table %s differs from the reference" % name
    print("derived fast-partial tables match the reference (%d values)" % sum(len(m) for _, m in checks))
    for rel in ("oracle/poseidon_constants.h", "plonky2_amd/csrc/poseidon_constants.h"):
        path = os.path.join(ROOT, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        emit(path, rc, circ, diag, first, scal, vs, w_hats, init)
        print("wrote", rel)


if __name__ == "__main__":
    sys.exit(main())
