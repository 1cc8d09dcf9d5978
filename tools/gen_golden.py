#!/usr/bin/env python3
"""Extract the reference's own known-answer vectors for the hot path into tests/golden/.

Run in the build container (needs /root/reference):  python tools/gen_golden.py
  - Poseidon width-12 permutation KATs: plonky2/src/hash/poseidon_goldilocks.rs:455-490
  - bit-reversal KATs: plonky2/src/util/mod.rs:61-126 (reverse_index_bits of 0..255 etc.)
Everything else on the path has no reference bytes.
"""
import json
import os
import re

REF = os.environ.get("P2_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0xFFFFFFFF00000001


def main():
    t = open(os.path.join(REF, "plonky2/src/hash/poseidon_goldilocks.rs")).read()
    body = t[t.index("let test_vectors12"):t.index("check_test_vectors::<F>(test_vectors12)")]
    body = body.replace("neg_one", str(P - 1))
    nums = [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\b\d+\b", body.split("=", 1)[1])]
    nums = [x for x in nums]
    # drop the "12" literals from the type annotation: they precede '=' so none remain
    assert len(nums) == 4 * 24, len(nums)
    kats = [{"input": nums[24 * i:24 * i + 12], "output": nums[24 * i + 12:24 * i + 24]} for i in range(4)]
    u = open(os.path.join(REF, "plonky2/src/util/mod.rs")).read()
    m = re.search(r"fn test_reverse_index_bits\(\)(.*?)\n    }\n", u, re.S)
    rb = m.group(1)
    # the big literal: reverse_index_bits(&(0..256).collect()) == vec![...]
    big = re.search(r"output256: Vec<u64> = vec!\[(.*?)\];", rb, re.S)
    table = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", big.group(1))]
    assert len(table) == 256 and sorted(table) == list(range(256))
    out = {"source": "plonky2/src/hash/poseidon_goldilocks.rs:455-490, plonky2/src/util/mod.rs:61-126",
           "poseidon12": kats, "reverse_index_bits_256": table}
    with open(os.path.join(ROOT, "tests/golden/reference_kats.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote tests/golden/reference_kats.json")


if __name__ == "__main__":
    main()
