#!/usr/bin/env python3
"""Parses include/blsw.h (the C ABI) and generates the Rust `extern "C"` binding shown in INTEGRATION.md.

    tools/gen_bindings.py            prints the Rust block
    tools/gen_bindings.py --update   rewrites the block between the GENERATED markers of INTEGRATION.md

tests/test_abi_contract.py uses parse_header() to check that INTEGRATION.md, the ctypes binding of
bls-verify-gadget_amd/__init__.py and the symbols libblsw.so exports all agree with the header.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "blsw.h")
INTEGRATION = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED RUST BINDING (tools/gen_bindings.py) -->", "<!-- END GENERATED RUST BINDING -->"

SCALARS = {"uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "int": "i32", "float": "f32", "double": "f64", "void": "c_void"}
STRUCT_NAMES = {"blsw_layout_t": "BlswLayout", "blsw_engine_options_t": "BlswEngineOptions", "blsw_engine_t": "BlswEngine",
                "blsw_matrices_t": "BlswMatrices", "blsw_matrices_info_t": "BlswMatricesInfo"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def parse_header(path=HEADER):
    """-> dict(structs={name: [(ctype, field)]}, functions=[(name, ret, [(ctype, pname)])], defines={name: int})"""
    raw = open(path).read()
    text = strip_comments(raw)
    defines = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+(BLSW_\w+)\s+(-?\w+)\s*$", text, flags=re.M) if re.fullmatch(r"-?(0x)?[0-9a-fA-F]+", m.group(2))}
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ctype, names = decl.split(" ", 1)
            for nm in names.split(","):
                nm = nm.strip()
                am = re.fullmatch(r"(\w+)\[(\d+)\]", nm)  # fixed-size array member
                fields.append((ctype + "[%s]" % am.group(2), am.group(1)) if am else (ctype, nm))
        structs[m.group(2)] = fields
    opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text)
    functions = []
    for m in re.finditer(r"^\s*(int|void)\s+(blsw_\w+)\s*\((.*?)\)\s*;", text, flags=re.S | re.M):
        params = []
        body = " ".join(m.group(3).split())
        if body and body != "void":
            for p in body.split(","):
                p = p.strip()
                pm = re.fullmatch(r"(.*?)(\w+)", p)
                params.append((pm.group(1).strip(), pm.group(2)))
        functions.append((m.group(2), m.group(1), params))
    return {"structs": structs, "functions": functions, "defines": defines, "opaque": [o[1] for o in opaque]}


def rust_type(ctype):
    am = re.fullmatch(r"(.*)\[(\d+)\]", ctype)
    if am:
        return "[%s; %s]" % (rust_type(am.group(1)), am.group(2))
    c = ctype.replace(" *", "*").replace("* ", "*").strip()
    const = c.startswith("const ")
    if const:
        c = c[len("const "):]
    stars = len(c) - len(c.rstrip("*"))
    base = c.rstrip("*").strip()
    r = SCALARS.get(base) or STRUCT_NAMES.get(base)
    if r is None:
        raise ValueError("unknown C type %r" % ctype)
    for k in range(stars):
        r = ("*const " if (const and k == 0) else "*mut ") + r
    return r


def rust_block(h=None):
    h = h or parse_header()
    out = ["use std::os::raw::c_void;", ""]
    for name, v in sorted(h["defines"].items(), key=lambda kv: (kv[0].split("_")[1], kv[1])):
        out.append("pub const %s: i32 = %d;" % (name, v))
    out.append("")
    for sname, fields in h["structs"].items():
        has_ptr = any("*" in ctype for ctype, _ in fields)
        out.append("#[repr(C)] #[derive(%sClone, Copy, Debug)]" % ("" if has_ptr else "Default, "))
        out.append("pub struct %s {   // %s: %d fields" % (STRUCT_NAMES[sname], sname, len(fields)))
        line = "   "
        for ctype, f in fields:
            item = " pub %s: %s," % (f, rust_type(ctype))
            if len(line) + len(item) > 118:
                out.append(line)
                line = "   "
            line += item
        out.append(line)
        out.append("}")
    for o in h["opaque"]:
        out.append("#[repr(C)] pub struct %s { _p: [u8; 0] }   // opaque" % STRUCT_NAMES[o])
    out.append("")
    out.append('#[link(name = "blsw")]')
    out.append('extern "C" {')
    for name, ret, params in h["functions"]:
        args = ", ".join("%s: %s" % (p, rust_type(t)) for t, p in params)
        line = "    pub fn %s(%s)%s;" % (name, args, " -> i32" if ret == "int" else "")
        if len(line) > 150:  # wrap long signatures
            parts = ["%s: %s" % (p, rust_type(t)) for t, p in params]
            line = "    pub fn %s(\n        %s,\n    )%s;" % (name, ",\n        ".join(
                ", ".join(parts[i:i + 4]) for i in range(0, len(parts), 4)), " -> i32" if ret == "int" else "")
        out.append(line)
    out.append("}")
    return "\n".join(out)


def integration_block(path=INTEGRATION):
    text = open(path).read()
    a, b = text.index(BEGIN), text.index(END)
    inner = text[a + len(BEGIN):b].strip()
    assert inner.startswith("```rust") and inner.endswith("```")
    return inner[len("```rust"):-3].strip()


def update(path=INTEGRATION):
    text = open(path).read()
    a, b = text.index(BEGIN), text.index(END)
    new = text[:a + len(BEGIN)] + "\n```rust\n" + rust_block() + "\n```\n" + text[b:]
    open(path, "w").write(new)


if __name__ == "__main__":
    if "--update" in sys.argv:
        update()
    else:
        print(rust_block())
