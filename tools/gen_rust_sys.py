#!/usr/bin/env python3
"""Generates bindings/rend3-amd-sys/src/lib.rs from include/r3n.h: every #define constant, every plain struct and every
function of the C ABI, 1:1.  tests/test_rust_bindings.py regenerates it in memory and fails when the committed file is stale
or a symbol of the header is missing from the extern block.   usage: python tools/gen_rust_sys.py [--check]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "r3n.h")
OUT = os.path.join(ROOT, "bindings", "rend3-amd-sys", "src", "lib.rs")

SCALARS = {"int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "uint16_t": "u16", "int32_t": "i32",
           "float": "f32", "double": "f64", "r3n_camera": "u32", "char": "c_char", "void": "c_void",
           "unsigned long long": "u64", "long long": "i64", "unsigned": "u32", "unsigned int": "u32", "size_t": "usize"}
# what a type of the generated file may be made of: anything else is a C spelling that leaked through (rustc would reject it)
RUST_WORDS = {"c_int", "c_char", "c_void", "u8", "u16", "u32", "u64", "i32", "i64", "f32", "f64", "usize", "const", "mut"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def rust_type(ctype):
    """'const float *' -> '*const f32', 'r3n_ctx *' -> '*mut r3n_ctx', 'void **' -> '*mut *mut c_void'."""
    t = ctype.strip()
    const = False
    if t.startswith("const "):
        const, t = True, t[6:].strip()
    stars = t.count("*")
    base = t.replace("*", "").strip()
    if base.startswith("struct "):
        base = base[7:]
    r = SCALARS.get(base, base)
    if r not in RUST_WORDS and not re.fullmatch(r"r3n_\w+", r):
        raise ValueError(f"gen_rust_sys: no Rust spelling for the C type {ctype!r} (add it to SCALARS)")
    for k in range(stars):
        r = ("*const " if (const and k == 0) else "*mut ") + r
    return r


def parse(text):
    raw = text
    text = strip_comments(text)
    consts = []
    for m in re.finditer(r"^#define\s+(R3N_\w+)\s+(\(?-?[0-9xXa-fA-F]+u?\)?)\s*$", text, flags=re.M):
        name, val = m.group(1), m.group(2).strip("()")
        if name == "R3N_H":
            continue
        unsigned = val.endswith("u")
        val = val.rstrip("u")
        consts.append((name, val, "u32" if unsigned else "i32"))
    structs = []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            const = decl.startswith("const ")
            if const:
                decl = decl[6:]
            base, rest = decl.split(" ", 1)
            # 'uint32_t first_joint, n_joints' style lists; '*name' declarators are pointers
            for p in [q.strip() for q in rest.split(",")]:
                am = re.match(r"(\**)\s*(\w+)((?:\[\w+\])*)$", p)
                stars, name, dims = len(am.group(1)), am.group(2), re.findall(r"\[(\w+)\]", am.group(3))
                ty = SCALARS.get(base, base)
                for k in range(stars):
                    ty = ("*const " if (const and k == 0) else "*mut ") + ty
                for d in reversed(dims):
                    ty = f"[{ty}; {d} as usize]" if not d.isdigit() else f"[{ty}; {d}]"
                fields.append((name, ty))
        structs.append((m.group(3), fields))
    opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text)
    fnptrs = []  # typedef int (*name)(args);
    for m in re.finditer(r"typedef\s+(\w+)\s*\(\s*\*\s*(\w+)\s*\)\s*\(([^)]*)\)\s*;", text):
        params = []
        for a in m.group(3).split(","):
            am = re.match(r"(.+?)(\w+)$", a.strip())
            params.append((am.group(2), rust_type(am.group(1).strip())))
        fnptrs.append((m.group(2), rust_type(m.group(1)), params))
    funcs = []
    body = text[text.index('extern "C" {'):]
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?[\w]+(?:\s*\*+\s*|\s+))(r3n_\w+)\s*\(([^;{}]*?)\)\s*;", body):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                am = re.match(r"(.+?)(\w+)((?:\[\w*\])*)$", a)
                ctype, pname, arr = am.group(1).strip(), am.group(2), am.group(3)
                if arr:  # array parameter decays to a pointer
                    ctype = ctype + " *"
                params.append((pname if pname not in ("type", "fn", "ref", "box") else pname + "_", rust_type(ctype)))
        funcs.append((name, rust_type(ret) if ret != "void" else None, params))
    return consts, structs, opaque, funcs, fnptrs


def generate():
    consts, structs, opaque, funcs, fnptrs = parse(open(HEADER).read())
    out = ["// rend3-amd-sys: raw bindings of librend3_amd.so -- GENERATED from include/r3n.h by tools/gen_rust_sys.py, do not edit.",
           "// One item per #define, struct and function of the header; the documentation lives there (each entry point cites the",
           "// rend3 interface it replaces).  Source only in this repository: the build image has no Rust toolchain, so this crate is",
           "// kept honest by tests/test_rust_bindings.py (symbol-for-symbol diff against the header and the built library).",
           "#![allow(non_camel_case_types, non_upper_case_globals, clippy::too_many_arguments)]",
           "use std::os::raw::{c_char, c_int, c_void};", ""]
    for name, val, ty in consts:
        out.append(f"pub const {name}: {ty} = {val};")
    out.append("")
    for _tag, name in opaque:
        out += [f"#[repr(C)]", f"pub struct {name} {{", "    _private: [u8; 0],", "}", ""]
    for name, ret, params in fnptrs:
        ps = ", ".join(f"{p}: {t}" for p, t in params)
        out += [f'pub type {name} = Option<unsafe extern "C" fn({ps}) -> {ret}>;', ""]
    for name, fields in structs:
        out += ["#[repr(C)]", "#[derive(Clone, Copy)]", f"pub struct {name} {{"]
        out += [f"    pub {f}: {t}," for f, t in fields]
        out += ["}", ""]
    out += ['#[link(name = "rend3_amd")]', 'extern "C" {']
    for name, ret, params in funcs:
        ps = ", ".join(f"{p}: {t}" for p, t in params)
        out.append(f"    pub fn {name}({ps})" + (f" -> {ret};" if ret else ";"))
    out += ["}", ""]
    return "\n".join(out)


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(OUT, len(text.splitlines()), "lines")
