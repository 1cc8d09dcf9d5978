"""The Rust side of the drop-in boundary (integration/): it cannot be compiled in this image (no cargo / rustc), so
it is checked mechanically --
  * every function include/p2hot.h declares has an `extern "C"` declaration in integration/p2hot.rs with the same
    parameter list (names, order, C type -> Rust type) and return type;
  * every #[repr(C)] struct mirrors its C struct field by field;
  * the module the patch adds as plonky2/src/p2hot.rs is integration/p2hot.rs verbatim;
  * the patch applies cleanly to the reference tree (when /root/reference is present: the build container)."""
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT

HDR = os.path.join(ROOT, "include", "p2hot.h")
RS = os.path.join(ROOT, "integration", "p2hot.rs")
PATCH = os.path.join(ROOT, "integration", "plonky2_p2hot.patch")

SCALAR = {"int": "c_int", "unsigned": "c_uint", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8",
          "char": "c_char", "void": "c_void"}


def _camel(name):
    return "".join(p.capitalize() for p in name.split("_"))


def _rust_type(ctype):
    """C parameter type (declarator stripped) -> the Rust FFI type"""
    t = " ".join(ctype.replace("*", " * ").split())
    if t == "p2hot_allgather_fn":
        return "P2hotAllgatherFn"
    toks = t.split()
    base_const = toks[0] == "const"
    if base_const:
        toks = toks[1:]
    base, rest = toks[0], toks[1:]
    rust = SCALAR.get(base) or _camel(base)
    const = base_const
    for tok in rest:
        if tok == "*":
            rust = ("*const " if const else "*mut ") + rust
            const = False
        elif tok == "const":
            const = True
        else:
            raise AssertionError("unexpected token %r in %r" % (tok, ctype))
    return rust


def _c_functions():
    h = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    h = re.sub(r"^\s*#.*$", "", h, flags=re.M)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(p2hot_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", h):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret or name == "p2hot_allgather_fn":
            continue
        plist = []
        if params and params != "void":
            for p in params.split(","):
                p = p.strip()
                arr = re.search(r"\[[^\]]*\]$", p)          # `const uint64_t alpha[2]` decays to a pointer
                if arr:
                    p = p[:arr.start()].strip()
                pm = re.match(r"(.*?)([A-Za-z_]\w*)$", p)
                ctype, pname = pm.group(1).strip(), pm.group(2)
                if arr:
                    ctype += " *"
                plist.append((pname, _rust_type(ctype)))
        out[name] = (plist, None if ret == "void" else _rust_type(ret))
    return out


def _rust_functions():
    s = open(RS).read()
    block = s[s.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    out = {}
    for m in re.finditer(r"pub fn (p2hot_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        name, params, ret = m.group(1), m.group(2), m.group(3)
        plist = []
        for p in [x.strip() for x in params.split(",") if x.strip()]:
            pname, ptype = [x.strip() for x in p.split(":", 1)]
            plist.append((pname.replace("r#", ""), " ".join(ptype.split())))
        out[name] = (plist, " ".join(ret.split()) if ret else None)
    return out


def test_every_header_function_has_a_matching_extern_declaration():
    c, r = _c_functions(), _rust_functions()
    assert len(c) >= 70
    assert sorted(c) == sorted(r), "symbols differ: %s" % sorted(set(c) ^ set(r))
    for name in c:
        assert c[name] == r[name], "%s:\n  header %s\n  rust   %s" % (name, c[name], r[name])


def _c_structs():
    h = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct \{(.*?)\}\s*(p2hot_[a-z0-9_]+)\s*;", h, flags=re.S):
        fields = []
        for decl in [d.strip() for d in m.group(1).split(";") if d.strip()]:
            tm = re.match(r"((?:const\s+)?[A-Za-z_]\w*)\s*(.*)$", decl, flags=re.S)
            base, rest = tm.group(1), tm.group(2)
            for item in [x.strip() for x in rest.split(",")]:
                stars = item.count("*")
                nm = item.replace("*", "").strip()
                arr = re.search(r"\[(\d+)\]$", nm)
                if arr:
                    nm = nm[:arr.start()]
                    fields.append((nm, "[%s; %s]" % (_rust_type(base), arr.group(1))))
                else:
                    fields.append((nm, _rust_type(base + " *" * stars)))
        out[m.group(2)] = fields
    return out


def test_repr_c_structs_mirror_the_header():
    s = open(RS).read()
    cs = _c_structs()
    assert set(cs) == {"p2hot_challenger_state", "p2hot_fri_batch_info", "p2hot_fri_params", "p2hot_fri_proof", "p2hot_fri_proof_layout"}
    for cname, fields in cs.items():
        m = re.search(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\)]*\)\]\s*)?pub struct %s \{(.*?)\n\}" % _camel(cname), s, flags=re.S)
        assert m, cname
        rf = [(a, " ".join(b.split())) for a, b in re.findall(r"pub (\w+): ([^,]+),", m.group(1))]
        assert rf == fields, "%s:\n  header %s\n  rust   %s" % (cname, fields, rf)


def test_patch_carries_the_module_verbatim():
    p = open(PATCH).read()
    start = p.index("+++ b/plonky2/src/p2hot.rs\n")
    body = p[start:].split("\n", 2)[2]                       # skip the +++ line and the @@ hunk header
    end = body.find("\ndiff -ruN ")
    body = body[:end + 1] if end >= 0 else body
    added = "".join(line[1:] + "\n" for line in body.splitlines() if line.startswith("+"))
    assert added == open(RS).read()
    for f in ("plonky2/Cargo.toml", "plonky2/build.rs", "plonky2/src/fri/oracle.rs", "plonky2/src/fri/prover.rs",
              "plonky2/src/hash/merkle_tree.rs", "plonky2/src/iop/challenger.rs", "plonky2/src/lib.rs"):
        assert "+++ b/%s\n" % f in p, f


def test_patch_applies_to_the_reference_tree():
    ref = os.environ.get("P2_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "plonky2", "src")):
        pytest.skip("the reference tree is only present in the build container")
    r = subprocess.run(["git", "apply", "--check", "--verbose", PATCH], cwd=ref, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # and it is what tools/make_rust_patch.py generates from the current integration/p2hot.rs
    before = open(PATCH).read()
    subprocess.check_call(["python", os.path.join(ROOT, "tools", "make_rust_patch.py")], stdout=subprocess.DEVNULL)
    assert open(PATCH).read() == before, "integration/plonky2_p2hot.patch is stale: run tools/make_rust_patch.py"
