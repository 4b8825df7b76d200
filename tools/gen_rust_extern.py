#!/usr/bin/env python3
"""Generate the Rust `extern "C"` block of INTEGRATION.md from include/gftaylor.h — one line per entry point, so the
binding a maintainer would add to the reference is complete by construction.  `--check` verifies that INTEGRATION.md
holds exactly the generated block (tests/test_abi_symbols.py runs it); without it the file is rewritten in place."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gftaylor.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED extern block (tools/gen_rust_extern.py) -->", "<!-- END GENERATED extern block -->"

TYPES = {
    "int": "c_int", "long": "c_long", "size_t": "usize", "double": "f64", "float": "f32", "uint32_t": "u32",
    "const double*": "*const f64", "double*": "*mut f64", "const size_t*": "*const usize", "size_t*": "*mut usize",
    "const gft_poly*": "*const GftPoly", "gft_poly*": "*mut GftPoly", "void*": "*mut c_void", "const void*": "*const c_void",
    "const char*": "*const c_char", "char*": "*mut c_char",
}


def rust_type(c):
    c = re.sub(r"\s+", " ", c.strip()).replace(" *", "*")
    if c not in TYPES:
        raise SystemExit(f"gen_rust_extern: no Rust mapping for C type '{c}'")
    return TYPES[c]


def declarations():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = text[text.index('extern "C" {'):]
    out = []
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?\**)\s*\b(gfti?_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for i, a in enumerate(args.split(",")):
                a = a.strip()
                am = re.match(r"(.+?)\s*\b([A-Za-z_][A-Za-z0-9_]*)\s*(\[\d*\])?$", a)
                ctype, pname, arr = am.group(1), am.group(2), am.group(3)
                if arr:
                    ctype = ctype + "*"
                params.append(f"{pname}: {rust_type(ctype)}")
        r = "" if ret == "void" else f" -> {rust_type(ret)}"
        out.append(f"    pub fn {name}({', '.join(params)}){r};")
    return out


def block():
    lines = ["```rust", "// src/gft_sys.rs — generated from include/gftaylor.h by tools/gen_rust_extern.py: every entry point, nothing else",
             "use std::os::raw::{c_char, c_int, c_long, c_void};", "#[repr(C)] pub struct GftPoly { _private: [u8; 0] }", 'extern "C" {']
    lines += declarations()
    lines += ["}", "```"]
    return "\n".join(lines)


def main():
    doc = open(DOC).read()
    a, b = doc.index(BEGIN) + len(BEGIN), doc.index(END)
    new = doc[:a] + "\n" + block() + "\n" + doc[b:]
    if "--check" in sys.argv:
        if new != doc:
            raise SystemExit("INTEGRATION.md's extern block is out of date: run python tools/gen_rust_extern.py")
        return
    open(DOC, "w").write(new)
    print(f"wrote {len(declarations())} declarations")


if __name__ == "__main__":
    main()
