#!/usr/bin/env python3
"""Hard-wraps the prose of Markdown files at 120 columns (reviewable diffs): paragraphs, list items (hanging indent) and block
quotes are re-flowed; headings, tables, fenced code, HTML and reference definitions are left byte for byte.  Idempotent.

    python tools/wrap_md.py [--check] FILE...
"""
import re
import sys
import textwrap

WIDTH = 120
ITEM = re.compile(r"^(\s*)([*+-]|\d+[.)])\s+")


def flush(par, out):
    if not par:
        return
    lines = list(par)
    par.clear()
    first = lines[0]
    quote = ""
    m = re.match(r"^(\s*>\s?)", first)
    if m:
        quote = m.group(1)
        lines = [ln[len(quote):] if ln.startswith(quote) else re.sub(r"^\s*>\s?", "", ln) for ln in lines]
        first = lines[0]
    m = ITEM.match(first)
    if m:
        lead = m.group(0)
        hang = " " * len(lead)
        body = [first[len(lead):]] + [ln.strip() for ln in lines[1:]]
    else:
        lead = re.match(r"^\s*", first).group(0)
        hang = lead
        body = [ln.strip() for ln in lines]
    text = " ".join(b for b in body if b)
    # two spaces after a sentence end are the source's style: keep single spaces, textwrap would otherwise break inside `code`
    wrapped = textwrap.wrap(text, width=WIDTH - len(quote), initial_indent=lead, subsequent_indent=hang, break_long_words=False,
                            break_on_hyphens=False)
    out.extend(quote + w for w in wrapped)


def wrap(src):
    out, par, fence = [], [], None
    for ln in src.split("\n"):
        stripped = ln.strip()
        if fence:
            out.append(ln)
            if stripped.startswith(fence):
                fence = None
            continue
        m = re.match(r"^\s*(```|~~~)", ln)
        if m:
            flush(par, out)
            fence = m.group(1)
            out.append(ln)
            continue
        special = (not stripped or stripped.startswith("#") or stripped.startswith("|") or stripped.startswith("<") or
                   re.match(r"^(-{3,}|={3,}|\*{3,})$", stripped) or re.match(r"^\[[^\]]+\]:", stripped) or stripped.startswith("{\""))
        if special:
            flush(par, out)
            out.append(ln)
            continue
        if par and (ITEM.match(ln) or (re.match(r"^\s*>", ln) and not re.match(r"^\s*>", par[0]))):
            flush(par, out)  # a new list item / a quote starts a new paragraph
        par.append(ln)
    flush(par, out)
    return "\n".join(out)


def main():
    check = "--check" in sys.argv
    bad = 0
    for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
        src = open(path).read()
        new = wrap(src)
        if new != src:
            if check:
                print("not wrapped:", path)
                bad = 1
            else:
                open(path, "w").write(new)
                print("wrapped", path)
    sys.exit(bad)


if __name__ == "__main__":
    main()
