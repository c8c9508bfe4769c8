"""Keeps rust/needle-hip/src/ffi.rs honest without a Rust toolchain: every `pub fn` it declares is
compared, argument by argument and in its return type, with the C prototype of the same name in include/*.h (a C
declarator is translated to the Rust FFI type bindgen would emit); every #[repr(C)] struct's size and field offsets,
computed with repr(C)'s rules from the Rust field types, are compared with `offsetof` / `sizeof` of the C struct as gcc
lays it out; the NeedleError variants are compared with the header's enumerators, in order."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SCALARS = {"size_t": "usize", "bool": "bool", "float": "f32", "double": "f64", "int": "c_int", "char": "c_char",
             "void": "c_void", "uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64",
             "int16_t": "i16", "int32_t": "i32", "int64_t": "i64"}
RUST_SIZES = {"bool": 1, "u8": 1, "i8": 1, "u16": 2, "i16": 2, "u32": 4, "i32": 4, "f32": 4, "c_int": 4, "u64": 8,
              "i64": 8, "f64": 8, "usize": 8}


def strip_comments(text):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))


def c_type_to_rust(decl: str) -> str:
    """`const char *const *paths` / `uint64_t counts[4]` / `struct NeedleAudioAnalyzer **output` -> the Rust FFI type."""
    decl = decl.strip()
    array = re.search(r"\[[^\]]*\]\s*$", decl)
    if array:
        decl = decl[: array.start()]
    tokens = re.findall(r"[A-Za-z_]\w*|\*", decl)
    tokens = [t for t in tokens if t not in ("struct", "enum")]
    # the base type is the first identifier that is not `const`; a trailing identifier after it is the parameter name
    base_at = next(i for i, t in enumerate(tokens) if t not in ("const", "*"))
    base = tokens[base_at]
    base_const = "const" in tokens[:base_at] or (base_at + 1 < len(tokens) and tokens[base_at + 1] == "const")
    rest = tokens[base_at + 1:]
    if rest and rest[0] == "const":
        rest = rest[1:]
    if rest and rest[-1] not in ("*", "const"):
        rest = rest[:-1]                                        # the parameter's name
    ptrs = []                                                   # innermost first: (is the POINTER itself const-qualified)
    for t in rest:
        if t == "*":
            ptrs.append(False)
        elif t == "const" and ptrs:
            ptrs[-1] = True
    rust = C_SCALARS.get(base, base)
    pointee_const = base_const
    for ptr_const in ptrs:
        rust = ("*const " if pointee_const else "*mut ") + rust
        pointee_const = ptr_const
    if array:                                                   # T name[N] as a parameter is a pointer to T
        rust = ("*const " if pointee_const else "*mut ") + rust
    return rust


def c_prototypes():
    text = ""
    for h in ("needle.h", "needle_hip.h"):
        text += strip_comments(open(os.path.join(ROOT, "include", h)).read())
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(needle_\w+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [c_type_to_rust(a) for a in args.split(",")]
        ret_rust = None if ret == "void" else c_type_to_rust(ret + " x")
        out[name] = (params, ret_rust)
    return out


def rust_declarations():
    src = strip_comments(open(os.path.join(ROOT, "rust", "needle-hip", "src", "ffi.rs")).read())
    fns = {}
    for m in re.finditer(r"pub fn (needle_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(2).split(",") if a.strip()]
        params = [re.sub(r"\s+", " ", a.split(":", 1)[1].strip()) for a in args]
        fns[m.group(1)] = (params, m.group(3).strip() if m.group(3) else None)
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*pub struct (\w+)\s*\{(.*?)\}", src, flags=re.S):
        fields = re.findall(r"pub (\w+)\s*:\s*([\w\[\]; ]+?)\s*,", m.group(2))
        structs[m.group(1)] = fields
    enum = re.search(r"pub enum NeedleError\s*\{(.*?)\}", src, flags=re.S)
    variants = [v.split("=")[0].strip() for v in enum.group(1).split(",") if v.strip()]
    return fns, structs, variants


def rust_layout(fields):
    """repr(C): fields in order, each at the next multiple of its alignment; size rounded up to the largest alignment."""
    off, align, out = 0, 1, []
    for name, ty in fields:
        size = RUST_SIZES[ty]
        off = (off + size - 1) // size * size
        out.append((name, off))
        off += size
        align = max(align, size)
    return out, (off + align - 1) // align * align


def c_layout(struct_fields, workdir):
    """sizeof / offsetof of the header's structs, as gcc lays them out."""
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "needle_hip.h"', "int main(void) {"]
    for name, fields in struct_fields.items():
        lines.append(f'  printf("{name} size %zu\\n", sizeof({name}));')
        for f, _ in fields:
            lines.append(f'  printf("{name} {f} %zu\\n", offsetof({name}, {f}));')
    lines += ["  return 0;", "}"]
    src, exe = os.path.join(workdir, "layout.c"), os.path.join(workdir, "layout")
    open(src, "w").write("\n".join(lines) + "\n")
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", exe, src], check=True)
    out = {}
    for line in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines():
        s, f, v = line.split()
        out[(s, f)] = int(v)
    return out


def header_error_variants():
    text = strip_comments(open(os.path.join(ROOT, "include", "needle.h")).read())
    body = re.search(r"enum NeedleError\s*\{(.*?)\}", text, flags=re.S).group(1)
    return [v.strip().split("=")[0].strip().replace("NeedleError_", "") for v in body.split(",") if v.strip()]
