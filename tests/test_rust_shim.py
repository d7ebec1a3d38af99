"""The Rust side of the boundary as files that can be checked without a Rust toolchain:
rust/nu_scaler_hip-sys (raw bindings), rust/nu_scaler_core_patch (impl Upscaler for HipUpscaler, the interpolator
pyclass).  No cargo / rustc in this image, so nothing here compiles Rust; what is checked instead:

* every prototype of include/nuscaler_hip.h has exactly one `extern "C"` declaration with the same name, arity,
  and per-parameter pointer depth, pointee constness and integer / float width (and vice versa: no declaration
  without a prototype);
* every enum value / #define of the header is a constant of the same value;
* every `sys::` item the wrapper files use exists in the -sys crate;
* the trait methods implemented are the ones the reference's `trait Upscaler` declares (names kept here as data).

The parsers below are written for this test and share nothing with tools/gen_rust_sys.py, which wrote the file."""
import os
import re

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "nuscaler_hip.h")
SYS = os.path.join(ROOT, "rust", "nu_scaler_hip-sys", "src", "lib.rs")
PATCH = os.path.join(ROOT, "rust", "nu_scaler_core_patch", "src")

# (pointer depth, constness of each level from the outside in, base) -- the comparison key of a type
C_BASE = {"int": ("i", 32), "int32_t": ("i", 32), "int64_t": ("i", 64), "uint32_t": ("u", 32), "uint64_t": ("u", 64),
          "size_t": ("u", "ptr"), "float": ("f", 32), "double": ("f", 64), "uint8_t": ("u", 8), "char": ("char", 8), "void": ("void", 0)}
RS_BASE = {"c_int": ("i", 32), "i32": ("i", 32), "i64": ("i", 64), "u32": ("u", 32), "u64": ("u", 64), "usize": ("u", "ptr"),
           "f32": ("f", 32), "f64": ("f", 64), "u8": ("u", 8), "c_char": ("char", 8), "c_void": ("void", 0)}


def c_key(t):
    toks = re.findall(r"\*|const|[A-Za-z_][A-Za-z0-9_]*", t)
    base, base_const, ptrs = None, False, []
    for tok in toks:
        if tok == "const":
            if ptrs:
                ptrs[-1] = True  # `* const`: the pointer itself is const (irrelevant for a by-value parameter)
            else:
                base_const = True
        elif tok == "*":
            ptrs.append(False)
        else:
            base = tok
    # constness of what each pointer level points AT, from the outermost pointer inwards
    pointee_const = []
    for lvl in range(len(ptrs) - 1, -1, -1):
        pointee_const.append(ptrs[lvl - 1] if lvl > 0 else base_const)
    return (len(ptrs), tuple(pointee_const), C_BASE.get(base, ("opaque", base)))


def rs_key(t):
    t = t.strip()
    consts = []
    while t.startswith("*"):
        m = re.match(r"^\*(const|mut)\s+(.*)$", t)
        consts.append(m.group(1) == "const")
        t = m.group(2).strip()
    return (len(consts), tuple(consts), RS_BASE.get(t, ("opaque", t)))


def header_protos():
    h = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for ret, name, args in re.findall(r"^([A-Za-z_][\w \*]*?)\b(nus_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", h, flags=re.M | re.S):
        args = " ".join(args.split())
        params = [] if args == "void" else [re.match(r"^(.*?)[A-Za-z_]\w*$", a.strip()).group(1) for a in args.split(",")]
        assert name not in out, name
        out[name] = (" ".join(ret.split()), params)
    consts = {k: int(v) for k, v in re.findall(r"^\s*(NUS_[A-Z0-9_]+)\s*=\s*(-?\d+)", h, flags=re.M)}
    consts.update({k: int(v) for k, v in re.findall(r"^#define\s+(NUS_[A-Z0-9_]+)\s+(-?\d+)\s*$", h, flags=re.M)})
    return out, consts


def rust_decls():
    src = open(SYS).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
    out = {}
    for name, args, ret in re.findall(r"pub fn (\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block, flags=re.S):
        params = [a.split(":", 1)[1].strip() for a in args.split(",") if a.strip()]
        assert name not in out, name
        out[name] = ((ret or "").strip(), params)
    consts = {k: int(v) for k, v in re.findall(r"pub const (NUS_[A-Z0-9_]+): c_int = (-?\d+);", src)}
    opaque = set(re.findall(r"pub struct (nus_\w+)", src))
    return out, consts, opaque


def test_every_prototype_has_a_matching_extern_declaration():
    protos, _ = header_protos()
    decls, _, opaque = rust_decls()
    assert len(protos) >= 70
    assert sorted(protos) == sorted(decls), (sorted(set(protos) - set(decls)), sorted(set(decls) - set(protos)))
    for name, (cret, cparams) in protos.items():
        rret, rparams = decls[name]
        assert len(cparams) == len(rparams), name
        for i, (ct, rt) in enumerate(zip(cparams, rparams)):
            assert c_key(ct) == rs_key(rt), (name, i, ct, rt)
        if cret == "void":
            assert rret == "", name
        else:
            assert c_key(cret) == rs_key(rret), (name, "return", cret, rret)
    used_opaque = {k[2][1] for n, (r, ps) in protos.items() for k in map(c_key, ps + [r]) if k[2][0] == "opaque"}
    assert used_opaque == opaque, (used_opaque, opaque)


def test_type_keys_tell_the_cases_apart():
    assert c_key("const uint8_t *const *") == rs_key("*const *const u8") == (2, (True, True), ("u", 8))
    assert c_key("uint8_t *const *") == rs_key("*const *mut u8")
    assert c_key("uint8_t *const *") != rs_key("*const *const u8")
    assert c_key("const char *") == rs_key("*const c_char") != rs_key("*mut c_char")
    assert c_key("size_t") == rs_key("usize") != rs_key("u32")
    assert c_key("int64_t") == rs_key("i64") != rs_key("c_int")
    assert c_key("const nus_upscaler *") == rs_key("*const nus_upscaler") != rs_key("*mut nus_upscaler")


def test_constants_match_the_header():
    _, cconsts = header_protos()
    _, rconsts, _ = rust_decls()
    assert cconsts == rconsts and len(cconsts) >= 40


def test_wrapper_files_only_use_what_the_sys_crate_declares():
    decls, consts, opaque = rust_decls()
    known = set(decls) | set(consts) | opaque
    for rel in ("upscale/hip.rs", "hip_interpolator.rs"):
        src = open(os.path.join(PATCH, rel)).read()
        used = set(re.findall(r"\bsys::(\w+)", src))
        assert used, rel
        assert used <= known, (rel, sorted(used - known))
    hip = open(os.path.join(PATCH, "upscale", "hip.rs")).read()
    # the methods of `trait Upscaler` (nu_scaler_core/src/upscale/mod.rs:67-88), all implemented in the impl block
    impl = hip[hip.index("impl Upscaler for HipUpscaler"):]
    for m in ("initialize", "upscale", "name", "quality", "set_quality", "as_any", "as_any_mut"):
        assert re.search(r"\bfn %s\(" % m, impl), m
    assert "fn upscale_batch" not in impl and "pub fn upscale_batch" in hip  # inherent in the reference too (mod.rs:609)
    interp = open(os.path.join(PATCH, "hip_interpolator.rs")).read()
    # the pyclass surface of wgpu_interpolator.rs:169-498
    assert '#[pyclass(name = "WgpuFrameInterpolator")]' in interp
    assert "#[pyo3(signature = (workgroup_preset_str=None))]" in interp
    assert "#[pyo3(signature = (frame_a_bytes, frame_b_bytes, width, height, *, time_t=0.5))]" in interp
    assert "fn get_last_gpu_duration_ms(&self) -> Option<f64>" in interp
    assert "Expected {} bytes per frame for {}x{}x4 RGBA, got frame_a: {} bytes, frame_b: {} bytes" in interp


def test_generator_is_idempotent(tmp_path):
    """tools/gen_rust_sys.py applied to the committed header reproduces the committed file."""
    import subprocess
    import sys

    before = open(SYS).read()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py")], check=True, capture_output=True)
    assert open(SYS).read() == before, "rust/nu_scaler_hip-sys/src/lib.rs is stale: run tools/gen_rust_sys.py"


def test_the_library_exports_what_the_sys_crate_links(nsc):
    import ctypes

    decls, _, _ = rust_decls()
    lib = ctypes.CDLL(nsc._capi.LIB_PATH)
    for name in decls:
        assert hasattr(lib, name), name
