"""Drift guard of the Rust binding (integration/stepth_gpu.rs; never compiled: the image has no
rustc).  Parses its `extern "C"` block, `#[repr(C)]` structs and constants and compares them
with include/stepth_amd.h: same functions, argument counts and C types, same field order, types
and sizes, same constant values.  The reference signatures the binding sits behind:
/root/reference/src/depth_image.rs:76-80, :91-95."""
import os
import re

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "stepth_amd.h")
RUST = os.path.join(ROOT, "integration", "stepth_gpu.rs")

C_SCALARS = {"int": "c_int", "uint32_t": "u32", "uint8_t": "u8", "uint64_t": "u64", "size_t": "usize",
             "double": "f64", "char": "c_char", "void": "c_void"}
SIZES = {"u8": 1, "u32": 4, "u64": 8, "usize": 8, "c_int": 4, "f64": 8}


def c_type_to_rust(t):
    """`const stepth_image *` -> `*const stepth_image`; `stepth_group *const *` -> `*const *mut stepth_group`"""
    t = re.sub(r"\s+", " ", t.strip())
    stars = []
    while t.endswith("*") or t.endswith("* const") or t.endswith("*const"):
        ptr_const = t.endswith("const")
        t = re.sub(r"\*\s*(const)?$", "", t).strip()
        stars.append(ptr_const)
    base_const = t.startswith("const ") or t.endswith(" const")
    base = t.replace("const", "").strip()
    base = C_SCALARS.get(base, base)
    out = base
    # innermost pointer applies to the (possibly const) base; outer ones to the pointer's own constness
    consts = [base_const] + stars[::-1][:-1] if stars else []
    for c in consts:
        out = ("*const " if c else "*mut ") + out
    return out


def header_functions():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    fns = {}
    for m in re.finditer(r"^(int|void|uint32_t|const char \*)\s*(stepth_\w+)\s*\(([^;]*?)\);", text, flags=re.S | re.M):
        ret, name, args = m.group(1).strip(), m.group(2), re.sub(r"\s+", " ", m.group(3)).strip()
        params = []
        if args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = re.search(r"\[(\d+)\]$", a)
                a = re.sub(r"\[\d+\]$", "", a)
                mm = re.match(r"(.*?)(\w+)$", a)
                ctype, pname = mm.group(1).strip(), mm.group(2)
                if arr:  # `const uint8_t precision[3]` decays to a pointer
                    ctype += " *"
                params.append((pname, c_type_to_rust(ctype)))
        rret = {"int": "c_int", "void": None, "uint32_t": "u32", "const char *": "*const c_char"}[ret]
        fns[name] = (params, rret)
    return fns


def rust_functions():
    text = open(RUST).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    fns = {}
    for m in re.finditer(r"pub fn (stepth_\w+)\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        args = re.sub(r"\s+", " ", m.group(2)).strip()
        params = []
        if args:
            for a in args.split(","):
                pname, ptype = a.split(":", 1)
                params.append((pname.strip(), ptype.strip()))
        fns[m.group(1)] = (params, m.group(3).strip() if m.group(3) else None)
    return fns


def test_extern_block_matches_the_header():
    h, r = header_functions(), rust_functions()
    assert len(h) >= 30
    assert sorted(h) == sorted(r), (sorted(set(h) - set(r)), sorted(set(r) - set(h)))
    for name in h:
        hp, hr = h[name]
        rp, rr = r[name]
        assert hr == rr, (name, hr, rr)
        assert [t for _n, t in hp] == [t for _n, t in rp], (name, hp, rp)
        # `main`/`out` style names: the same words, so that call sites read alike
        assert [n for n, _t in hp] == [n for n, _t in rp], (name, [n for n, _ in hp], [n for n, _ in rp])


def header_structs():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct (\w+)?\s*\{(.*?)\}\s*(\w+);", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = re.sub(r"\s+", " ", decl).strip()
            if not decl:
                continue
            mm = re.match(r"(const void \*|\w+)\s*(.*)$", decl)
            ctype = mm.group(1).strip()
            for nm in mm.group(2).split(","):
                nm = nm.strip()
                arr = re.search(r"\[(\d+)\]$", nm)
                nm = re.sub(r"\[\d+\]$", "", nm)
                rt = "*const c_void" if ctype.startswith("const void") else C_SCALARS[ctype]
                fields.append((nm, f"[{rt}; {arr.group(1)}]" if arr else rt))
        out[m.group(3)] = fields
    return out


def rust_structs():
    text = open(RUST).read()
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", text, flags=re.S):
        fields = [(a.strip(), b.strip()) for a, b in re.findall(r"pub (\w+):\s*([^,\n]+),", m.group(2))]
        out[m.group(1)] = fields
    return out


def rust_size(fields):
    size = 0
    for _n, t in fields:
        arr = re.match(r"\[(\w+); (\d+)\]", t)
        n, al = (SIZES[arr.group(1)] * int(arr.group(2)), SIZES[arr.group(1)]) if arr else \
            ((8, 8) if t.startswith("*") else (SIZES[t], SIZES[t]))
        size = (size + al - 1) // al * al + n
    al = max(8 if t.startswith("*") else SIZES[re.sub(r"\[(\w+); \d+\]", r"\1", t)] for _n, t in fields)
    return (size + al - 1) // al * al


def test_repr_c_structs_match_the_header():
    h, r = header_structs(), rust_structs()
    for name in ("stepth_group", "stepth_image", "stepth_params", "stepth_stats", "stepth_zone"):
        assert h[name] == r[name], (name, h[name], r[name])
    # the sizes engine.hip static_asserts and stepth_amd/_lib.py mirrors
    assert rust_size(r["stepth_group"]) == 20 and rust_size(r["stepth_image"]) == 32
    assert rust_size(r["stepth_params"]) == 56 and rust_size(r["stepth_stats"]) == 56
    assert rust_size(r["stepth_zone"]) == 4
    assert r["stepth_engine"] == [] or "stepth_engine" in r  # opaque


def test_constants_match_the_header():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    rust = open(RUST).read()
    consts = dict(re.findall(r"\b(STEPTH_(?:OK|ERR_\w+|MODE_\w+|MEM_\w+|FLAG_\w+))\s*=\s*([^,\n}]+)", text))
    assert len(consts) >= 18
    for k, v in consts.items():
        want = eval(v.replace("u", ""))
        m = re.search(r"pub const " + k + r": \w+ = ([^;]+);", rust)
        assert m, f"{k} missing in stepth_gpu.rs"
        assert eval(m.group(1)) == want, (k, m.group(1), v)
    abi = int(re.search(r"#define STEPTH_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert int(re.search(r"pub const STEPTH_ABI_VERSION: u32 = (\d+);", rust).group(1)) == abi


def test_binding_keeps_the_reference_signatures():
    """the methods the binding sits behind keep the reference's signatures
    (src/depth_image.rs:91-95 and the consumers)"""
    rust = re.sub(r"\s+", " ", open(RUST).read())
    assert "pub fn load_depth_from_additional(&mut self, add_image: image::DynamicImage, precision: [u8; 3]) -> Result<(), std::io::Error>" in rust
    assert "pub fn depth_split(&self, zones: u8) -> Vec<(Option<u8>, Option<u8>)>" in rust
    assert "pub fn select_foreground(&mut self) -> MaskImage" in rust
    assert '"Sizes don\'t match"' in rust
