"""A regex-level API-surface linter for the Rust files that cannot be compiled here (TEST INFRASTRUCTURE).

integration/p2hot.rs and integration/p2hot_dump_goldens.rs are written against the reference crate without a compiler in
the image.  Against a PATCHED copy of the reference tree (integration/plonky2_p2hot.patch applied) this module checks what
a first `cargo build` would otherwise find:
  uses       every `use crate::a::b::{X, Y}` / `use plonky2::a::b::X` names an item that exists in that module (or is re-exported
             there) with a visibility that reaches the user: `pub` for the example, `pub` / `pub(crate)` / `pub(super)` for the module
  calls      every call of a function or method DEFINED in the reference (free fn, associated fn, trait method) passes a number of
             arguments one of the definitions of that name accepts, and a turbofish carries a generic-argument count one of them has
  literals   every struct literal `Name { a, b: .., }` of a reference struct lists exactly the fields of the (patched) definition
             (unless it ends in `..`), and every `x.field` / destructuring of such fields is out of scope
It is deliberately conservative: names it cannot resolve are skipped, never guessed."""
import os
import re
import shutil
import subprocess
import tempfile

CRATES = {"crate": "plonky2/src", "plonky2": "plonky2/src", "plonky2_field": "field/src", "plonky2_util": "util/src",
          "plonky2_maybe_rayon": "maybe_rayon/src"}
# method / function names whose std / core / rayon / anyhow homonyms make an arity comparison meaningless
STD_NAMES = set("""new len map collect iter iter_mut into_iter unwrap expect clone push extend insert get get_mut lock contains_key as_ptr
as_mut_ptr as_ref as_deref as_slice to_vec into_boxed_slice with_capacity set_len from_raw_parts copy_nonoverlapping zip enumerate
min max sum all any rev chunks_exact par_chunks_exact par_iter to_string to_string_lossy into_owned from_ptr parse ok and_then unwrap_or
unwrap_or_default is_some is_none is_empty store load get_or_init var default null_mut null max_by min_by filter fold take skip step_by
concat join write fmt eq drop then ok_or map_err starts_with ends_with first last finish init add sub mul div neg inverse square exp
pow double from to into try_into try_from cast byref resize truncate clear remove pop append split_at chain flat_map cloned copied
count position find rfind nth last_mut swap sort sort_by dedup retain drain splice to_owned borrow borrow_mut deref index
is_power_of_two trailing_zeros leading_zeros next rand verify""".split())


def strip(src):
    """comments and string / char literals blanked out (same length is not needed)"""
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r'b?"(?:\\.|[^"\\])*"', '""', src, flags=re.S)
    src = re.sub(r"b?'(?:\\.|[^'\\])'", "' '", src)
    return src


def patched_tree(ref, patch):
    top = tempfile.mkdtemp(prefix="p2hot_lint_")
    for d in ("plonky2", "field", "util", "maybe_rayon", "starky"):
        shutil.copytree(os.path.join(ref, d), os.path.join(top, d), ignore=shutil.ignore_patterns("target", "*.md"))
    subprocess.check_call(["git", "init", "-q", top])
    subprocess.check_call(["git", "apply", patch], cwd=top)
    return top


def split_top(s, sep=","):
    """split on `sep` outside (), [], {}, <> (the last only when it is not part of -> or a comparison: good enough for argument lists)"""
    out, depth, cur = [], 0, ""
    i = 0
    while i < len(s):
        ch = s[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        elif ch == "<" and re.match(r"[\w:>)\]]", s[i - 1] if i else " ") and not s[i + 1:i + 2] in ("=", " "):
            depth += 1
        elif ch == ">" and depth > 0 and s[i - 1] not in "-=" and re.search(r"<", cur) is not None and _angle_open(cur):
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
        i += 1
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out if x.strip()]


def _angle_open(cur):
    opens = len(re.findall(r"(?<=[\w:>)\]])<(?![= ])", cur))
    closes = len(re.findall(r"(?<![-=])>", cur))
    return opens > closes


def matching(s, i, open_ch, close_ch):
    """index of the bracket closing s[i] == open_ch"""
    depth = 0
    for j in range(i, len(s)):
        if s[j] == open_ch:
            depth += 1
        elif s[j] == close_ch:
            depth -= 1
            if depth == 0:
                return j
    return -1


class Reference:
    def __init__(self, top):
        self.top = top
        self.fns = {}       # name -> set of (n_args_without_self, n_generics)
        self.structs = {}   # name -> set of field names (named-field structs only)
        self.sources = {}
        for crate_dir in set(CRATES.values()):
            for d, _, files in os.walk(os.path.join(top, crate_dir)):
                for f in files:
                    if f.endswith(".rs"):
                        p = os.path.join(d, f)
                        self.sources[p] = strip(open(p).read())
        for p, src in self.sources.items():
            self._scan_fns(src)
            self._scan_structs(src)

    def _scan_fns(self, src):
        for m in re.finditer(r"\bfn\s+(\w+)\s*(<)?", src):
            name, i = m.group(1), m.end()
            ngen = 0
            if m.group(2):
                j = self._angle_close(src, m.end() - 1)
                if j < 0:
                    continue
                gens = [g for g in split_top(src[m.end():j]) if not g.lstrip().startswith("'")]
                ngen = len(gens)
                i = j + 1
            k = src.find("(", i)
            if k < 0 or src[i:k].strip():
                continue
            e = matching(src, k, "(", ")")
            params = split_top(src[k + 1:e])
            if params and re.match(r"(&\s*('\w+\s+)?)?(mut\s+)?self\b", params[0]):
                params = params[1:]
            self.fns.setdefault(name, set()).add((len(params), ngen))

    @staticmethod
    def _angle_close(s, i):
        depth = 0
        for j in range(i, len(s)):
            if s[j] == "<":
                depth += 1
            elif s[j] == ">" and s[j - 1] != "-":
                depth -= 1
                if depth == 0:
                    return j
            elif s[j] in "{;" and depth == 0:
                return -1
        return -1

    def _scan_structs(self, src):
        for m in re.finditer(r"\bstruct\s+(\w+)\b([^;{(]*)\{", src):
            e = matching(src, m.end() - 1, "{", "}")
            body = src[m.end():e]
            body = re.sub(r"#\[[^\]]*\]", "", body)
            fields = set()
            for item in split_top(body):
                fm = re.match(r"(?:pub(?:\([^)]*\))?\s+)?(\w+)\s*:", item)
                if fm:
                    fields.add(fm.group(1))
            self.structs.setdefault(m.group(1), []).append(fields)

    def module_file(self, crate, path):
        base = os.path.join(self.top, CRATES[crate])
        if crate in ("crate", "plonky2") and path and path[0] == "field":  # pub use plonky2_field as field (lib.rs:12)
            base, path = os.path.join(self.top, CRATES["plonky2_field"]), path[1:]
        cand = [os.path.join(base, *path) + ".rs", os.path.join(base, *path, "mod.rs")] if path else [os.path.join(base, "lib.rs")]
        for c in cand:
            if os.path.exists(c):
                return c
        return None

    def item_visibility(self, crate, path, name):
        """'pub', 'crate', 'private' or None (not found) for `name` in module crate::path"""
        f = self.module_file(crate, path)
        if not f:
            return None
        src = self.sources.get(f) or strip(open(f).read())
        m = re.search(r"(?:^|\n)\s*(pub(?:\([^)]*\))?\s+)?(?:unsafe\s+)?(?:const\s+)?(?:struct|enum|trait|fn|const|type|static|mod|union)\s+%s\b" % re.escape(name), src)
        if m:
            v = (m.group(1) or "").strip()
            return "pub" if v == "pub" else "crate" if v.startswith("pub(") else "private"
        for um in re.finditer(r"(?:^|\n)\s*(pub(?:\([^)]*\))?)\s+use\s+([^;]+);", src):
            body = um.group(2)
            if re.search(r"\b%s\b" % re.escape(name), body) or body.rstrip().endswith("::*"):
                if re.search(r"\b%s\b" % re.escape(name), body):
                    return "pub" if um.group(1) == "pub" else "crate"
                # glob re-export: look the name up in the re-exported module
                gm = re.match(r"\s*(\w+)((?:::\w+)*)::\*", body)
                if gm and gm.group(1) in CRATES:
                    sub = [x for x in gm.group(2).split("::") if x]
                    v = self.item_visibility(gm.group(1), sub, name)
                    if v:
                        return v if um.group(1) == "pub" else "crate"
        return None


def uses(src):
    """[(crate, [module path], name)] of every `use crate::...` / `use plonky2::...` statement"""
    out = []
    for m in re.finditer(r"\buse\s+((?:crate|plonky2\w*)(?:::\w+)*)::(\{[^}]*\}|\w+|\*)\s*;", strip(src)):
        head = m.group(1).split("::")
        names = [n.strip() for n in m.group(2).strip("{}").split(",")] if m.group(2) != "*" else []
        for n in names:
            n = n.split(" as ")[0].strip()
            if n and n != "self":
                out.append((head[0], head[1:], n))
    return out


def calls(src):
    """[(name, n_args, n_turbofish_generics or None, is_method)] of every call expression `name(..)`, `name::<..>(..)`, `.name(..)`"""
    s = strip(src)
    out = []
    for m in re.finditer(r"(\.|::|\b)(\w+)\s*(::\s*<)?", s):
        name = m.group(2)
        i = m.end()
        ngen = None
        if m.group(3):
            j = Reference._angle_close(s, i - 1)
            if j < 0:
                continue
            ngen = len([g for g in split_top(s[i:j]) if not g.lstrip().startswith("'")])
            i = j + 1
        if s[i:i + 1] != "(":
            continue
        if re.match(r"(fn|if|while|for|match|return|in|as|let|mut|ref|move|loop|unsafe|where|impl|struct|enum|use|pub|mod|type|const|static|Some|Ok|Err|None)$", name):
            continue
        if re.search(r"\bfn\s+$", s[max(0, m.start() - 8):m.start(2)]):
            continue  # a definition, not a call
        e = matching(s, i, "(", ")")
        if e < 0:
            continue
        out.append((name, len(split_top(s[i + 1:e])), ngen, m.group(1) == "."))
    return out


def struct_literals(src, known):
    """[(Name, set(fields), has_rest)] of every `Name { field: expr, field, ..rest }` whose Name is a known reference struct"""
    s = strip(src)
    out = []
    for m in re.finditer(r"\b([A-Z]\w*)\s*(?:::\s*<[^{};]*?>\s*)?\{", s):
        name = m.group(1)
        if name not in known:
            continue
        before = s[max(0, m.start() - 40):m.start()]
        if re.search(r"\b(struct|enum|impl|trait|for|union)\s+$", before) or re.search(r"\bimpl\b[^;{}]*$", before) or re.search(r"->\s*$", before):
            continue
        e = matching(s, m.end() - 1, "{", "}")
        body = s[m.end():e]
        items = split_top(body)
        fields, rest, ok = set(), False, True
        for it in items:
            if it.startswith(".."):
                rest = True
                continue
            fm = re.match(r"(\w+)\s*(:|$)", it)
            if not fm:
                ok = False
                break
            fields.add(fm.group(1))
        if ok and (fields or rest):
            out.append((name, fields, rest))
    return out
