"""The Rust binding kept in the tree (bindings/takzero-hip-sys, bindings/takzero-hip) against include/takzero_hip.h.  No Rust
toolchain exists in the build image, so what can be checked is checked by parsing: every entry point of the header is declared in
the sys crate's `extern "C"` block with the same name, arity and parameter / return types (C -> Rust FFI type by a table written
here, independent of the generator tools/gen_rust_sys.py); every #define is a `pub const` of the same value; the two plain structs
have the same fields in the same order; and every `sys::tz_*` call in the adapter crate names a declared function with the number of
arguments it is called with."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "takzero_hip.h")
SYS = os.path.join(ROOT, "bindings", "takzero-hip-sys", "src", "lib.rs")
ADAPTER = os.path.join(ROOT, "bindings", "takzero-hip", "src", "lib.rs")

C2R = {"int": "c_int", "unsigned": "c_uint", "unsigned long long": "u64", "long long": "i64", "signed char": "i8", "float": "f32", "double": "f64", "char": "c_char", "unsigned char": "u8", "void": "c_void",
       "size_t": "usize", "int8_t": "i8", "uint8_t": "u8", "int16_t": "i16", "uint16_t": "u16", "int32_t": "i32", "uint32_t": "u32",
       "int64_t": "i64", "uint64_t": "u64"}


def camel(n):
    return "".join(p.capitalize() for p in n.split("_"))


def split_top(text):
    out, depth, cur = [], 0, ""
    for i, ch in enumerate(text):
        depth += ch in "(<["
        depth -= ch in ")]" or (ch == ">" and text[i - 1] != "-")
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def c_type_to_rust(c):
    c = " ".join(c.replace("*", " * ").split())
    m = re.match(r"^int \( \* \w* ?\) ?\((.*)\)$", c)
    if m:
        return 'Option<unsafe extern "C" fn(%s) -> c_int>' % ", ".join(c_type_to_rust(strip_name(a)) for a in split_top(m.group(1)) if a != "void")
    toks = c.split()
    stars = toks.count("*")
    const = "const" in toks
    base = " ".join(t for t in toks if t not in ("*", "const", "struct"))
    r = C2R.get(base) or (camel(base) if base.startswith("tz_") else None)
    assert r is not None, c
    for i in range(stars):
        r = ("*const " if const and i == 0 else "*mut ") + r
    return r


def strip_name(param):
    """'const tz_state* states' -> 'const tz_state*' (callbacks keep their shape)"""
    if "(*" in param.replace(" ", ""):
        return re.sub(r"\(\s*\*\s*\w+\s*\)", "( * )", param)
    param = param.strip()
    m = re.match(r"^(.*?)(\w+)\s*(\[\s*\])?$", param)
    if m is None or not m.group(1).strip() or m.group(2) in C2R or m.group(1).strip() in ("const", "unsigned", "const unsigned"):
        return param   # a bare type without a name (the parameters of a callback type)
    return m.group(1).strip() + ("*" if m.group(3) else "")


def header_items():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    consts = {k: int(v.strip("()")) for k, v in re.findall(r"^#define\s+(TZ_\w+)\s+(\(?-?\d+\)?)\s*$", src, flags=re.M)}
    flat = " ".join(re.sub(r"^#.*$", "", src, flags=re.M).split())
    funcs = {}
    for ret, name, args in re.findall(r"([A-Za-z_][\w\s\*]*?)\b(tz_\w+)\s*\(((?:[^()]|\([^()]*\))*)\)\s*;", flat):
        if ret.strip().startswith("typedef"):
            continue
        params = [c_type_to_rust(strip_name(a)) for a in split_top(args) if a != "void"]
        funcs[name] = (params, c_type_to_rust(ret.strip()))
    structs = {}
    body_src = re.sub(r"union\s*\{[^}]*\}\s*(\w+)\s*;", r"uint32_t \1_bits;", src)
    for name, body in re.findall(r"typedef struct (\w+)\s*\{(.*?)\}\s*\w+\s*;", body_src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if decl:
                m = re.match(r"^(.*?)(\w+)\s*(?:\[(\w+)\])?$", decl)
                t = c_type_to_rust(m.group(1).strip())
                if m.group(3):
                    t = "[%s; %d]" % (t, consts.get(m.group(3), None) or int(m.group(3)))
                fields.append((m.group(2), t))
        structs[camel(name)] = fields
    return consts, funcs, structs


def rust_items():
    src = open(SYS).read()
    src = re.sub(r"//[^\n]*", "", src)
    consts = {k: int(v) for k, v in re.findall(r"pub const (TZ_\w+): c_int = (-?\d+);", src)}
    block = re.search(r'extern "C" \{(.*)\}\s*$', src, flags=re.S).group(1)
    funcs = {}
    flat = " ".join(block.split())
    for m in re.finditer(r"pub fn (\w+)\(", flat):
        depth, i = 1, m.end()
        while depth:
            depth += flat[i] == "("
            depth -= flat[i] == ")"
            i += 1
        args = flat[m.end():i - 1]
        ret = re.match(r"\s*->\s*([^;]+);", flat[i:]).group(1)
        funcs[m.group(1)] = ([a.split(":", 1)[1].strip() for a in split_top(args)], ret.strip())
    structs = {}
    for name, body in re.findall(r"pub struct (\w+) \{(.*?)\n\}", src, flags=re.S):
        fields = [(n, " ".join(t.split())) for n, t in re.findall(r"pub (\w+): ([^,\n]+),", body)]
        if fields:
            structs[name] = fields
    return consts, funcs, structs


def test_every_entry_point_is_declared_with_the_headers_types():
    hc, hf, hs = header_items()
    rc, rf, rs = rust_items()
    assert len(hf) >= 111
    assert sorted(hf) == sorted(rf), (sorted(set(hf) - set(rf)), sorted(set(rf) - set(hf)))
    for name, (params, ret) in hf.items():
        assert rf[name][1] == ret, (name, rf[name][1], ret)
        assert len(rf[name][0]) == len(params), (name, len(rf[name][0]), len(params))
        for i, (a, b) in enumerate(zip(rf[name][0], params)):
            assert a == b, (name, i, a, b)
    assert rc == hc
    assert rs == hs, (rs, hs)
    assert [n for n, _ in hs["TzState"]][:3] == ["colors", "height", "top"]


def test_the_adapter_calls_declared_functions_with_their_arity():
    _, rf, _ = rust_items()
    src = re.sub(r"//[^\n]*", "", open(ADAPTER).read())
    calls = 0
    for m in re.finditer(r"sys::(tz_\w+)\s*\(", src):
        name = m.group(1)
        assert name in rf, name
        depth, i = 1, m.end()
        while depth:
            depth += src[i] in "([{"
            depth -= src[i] in ")]}"
            i += 1
        args = split_top(src[m.end():i - 1])
        assert len(args) == len(rf[name][0]), (name, len(args), len(rf[name][0]))
        calls += 1
    assert calls >= 20
    for const in set(re.findall(r"sys::(TZ_\w+)", src)):
        assert const in rust_items()[0], const


def test_the_generator_reproduces_the_committed_file(tmp_path, monkeypatch):
    """the committed lib.rs is what tools/gen_rust_sys.py writes from the header as it stands (a header change without regenerating fails here)"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_rust_sys", os.path.join(ROOT, "tools", "gen_rust_sys.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    out = tmp_path / "lib.rs"
    monkeypatch.setattr(gen, "OUT", str(out))
    gen.main()
    assert out.read_text() == open(SYS).read()
