"""CPU checks of the drop-in boundary: the shared library builds, loads, exports every symbol the header
declares, and fails loudly (error code + message, no abort, no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest


def test_library_present_and_exports_every_declared_symbol(dcn):
    N = dcn._native
    assert os.path.exists(N.LIB_PATH), "build with __graft_entry__.build()"
    L = C.CDLL(N.LIB_PATH)
    declared = N.declared_symbols()
    assert len(declared) >= 29
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/deacon_hip.h but not exported"
    # and the binding table covers exactly the declared surface
    assert sorted(N._SIGNATURES) == declared


def test_exports_are_c_abi(dcn):
    out = subprocess.check_output(["nm", "-D", "--defined-only", dcn._native.LIB_PATH], text=True)
    syms = {line.split()[-1] for line in out.splitlines() if " T " in line}
    for name in dcn._native.declared_symbols():
        assert name in syms  # unmangled => extern "C"


def test_params_layout_matches_header(dcn):
    P = dcn._native.Params
    assert C.sizeof(P) == 32
    assert (P.abs_threshold.offset, P.rel_threshold.offset, P.prefix_length.offset,
            P.deplete.offset, P.reserved.offset) == (0, 8, 16, 24, 28)


def test_header_compiles_as_plain_c(tmp_path, dcn):
    src = tmp_path / "t.c"
    src.write_text('#include "deacon_hip.h"\nint main(void){ dcn_params p; (void)p; return DCN_N_STATS == 6 ? 0 : 1; }\n')
    inc = os.path.dirname(dcn._native.HEADER_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", inc, str(src), "-o",
                           str(tmp_path / "t")])
    assert subprocess.call([str(tmp_path / "t")]) == 0


def test_version_and_argument_errors(dcn):
    L = dcn._native.lib()
    assert b"deacon-hip" in L.dcn_version()
    h = C.c_void_p()
    keys = np.arange(10, dtype=np.uint64)
    kp = keys.ctypes.data_as(C.c_void_p)
    assert L.dcn_index_from_keys(kp, 10, 31, 15, 0, None) == dcn._native.DCN_ERR_ARG
    assert L.dcn_index_from_keys(kp, 10, 31, 16, 0, C.byref(h)) == dcn._native.DCN_ERR_ARG  # k+w-1 even
    assert b"odd" in L.dcn_last_error()
    assert L.dcn_index_from_keys(kp, 10, 57, 15, 0, C.byref(h)) == dcn._native.DCN_ERR_ARG  # k > 56
    assert L.dcn_index_from_keys(None, 10, 31, 15, 0, C.byref(h)) == dcn._native.DCN_ERR_ARG
    assert L.dcn_ctx_create(None, 1 << 20, 1 << 10, C.byref(h)) == dcn._native.DCN_ERR_ARG
    assert L.dcn_ctx_synchronize(None) == dcn._native.DCN_ERR_ARG
    assert L.dcn_index_header(None, None, None, None) == dcn._native.DCN_ERR_ARG
    L.dcn_index_destroy(None)  # no-ops, must not crash
    L.dcn_ctx_destroy(None)


def test_no_gpu_means_loud_failure_not_fallback(dcn):
    L = dcn._native.lib()
    n = C.c_int(-1)
    rc = L.dcn_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    keys = np.arange(1, 100, dtype=np.uint64)
    with pytest.raises(dcn.DeaconHipError) as e:
        dcn.Index.from_keys(keys, 31, 15)
    assert e.value.code in (dcn._native.DCN_ERR_HIP, dcn._native.DCN_ERR_ARG)


def test_index_file_errors(dcn, tmp_path):
    L = dcn._native.lib()
    h = C.c_void_p()
    assert L.dcn_index_from_file(os.fsencode(tmp_path / "missing.idx"), 0, C.byref(h)) == dcn._native.DCN_ERR_IO
    bad = tmp_path / "bad.idx"
    bad.write_bytes(bytes([1, 31, 15, 0]))
    assert L.dcn_index_from_file(os.fsencode(bad), 0, C.byref(h)) == dcn._native.DCN_ERR_FORMAT
    assert b"format version" in L.dcn_last_error()
    trunc = tmp_path / "trunc.idx"
    trunc.write_bytes(bytes([2, 31, 15, 5, 0xFD, 1, 2]))
    assert L.dcn_index_from_file(os.fsencode(trunc), 0, C.byref(h)) == dcn._native.DCN_ERR_FORMAT


def test_product_does_not_reference_the_oracle(dcn):
    """The oracle is test infrastructure: nothing in the product package may import, link or mention it."""
    pkg = os.path.dirname(dcn._native.LIB_PATH)
    pkg = os.path.dirname(pkg)
    for root, _, files in os.walk(pkg):
        if os.path.basename(root) in ("build", "lib", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "liboracle" not in text and "deacon_oracle" not in text and "from oracle" not in text, f
    out = subprocess.check_output(["ldd", dcn._native.LIB_PATH], text=True)
    assert "oracle" not in out


def test_c_caller_links_and_runs(tmp_path, dcn):
    """the boundary is a C ABI: a C99 translation unit (tests/c/abi_smoke.c) includes the header, links the library and
    calls the entry points that need no GPU (version, the host-side packer, argument errors)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(dcn._native.LIB_PATH)
    exe = tmp_path / "abi_smoke"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c", "abi_smoke.c"), "-o", str(exe), "-L", libdir, "-ldeacon_hip",
                           f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert p.stdout.startswith("deacon-hip ")


# ---- INTEGRATION.md's extern "C" block against the header (21 of 38 exports were bound) -------------------------
def _c_declarations(header_text):
    """name -> (return type, [argument types]) of every function the header declares, types as normalised C strings"""
    import re
    text = re.sub(r"/\*.*?\*/", "", header_text, flags=re.S)
    out = {}
    for m in re.finditer(r"^([A-Za-z_][A-Za-z0-9_ \*]*?)\b(dcn_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.M | re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        argv = [] if args in ("void", "") else [a.strip() for a in args.split(",")]
        out[name] = (ret, argv)
    return out


def _rust_of_c(ctype):
    """'const uint64_t *keys' -> '*const u64' (the Rust FFI spelling of a C parameter or return type)"""
    import re
    t = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*\s*\[[^\]]*\]\s*$", "*", ctype.strip())  # `name[N]` parameter = pointer
    toks = re.findall(r"const|\*|[A-Za-z_][A-Za-z0-9_]*", t)
    base = {"uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "int": "c_int", "char": "c_char", "void": "c_void",
            "float": "f32", "double": "f64", "dcn_index": "dcn_index", "dcn_ctx": "dcn_ctx", "dcn_params": "dcn_params", "dcn_comm": "dcn_comm"}
    # drop a trailing parameter name (an identifier that is not a known type word, after the type is complete)
    while toks and toks[-1] not in base and toks[-1] not in ("const", "*"):
        toks.pop()
    const_next, cur = False, None
    for tk in toks:
        if tk == "const":
            const_next = True
        elif tk == "*":
            cur = ("*const " if const_next else "*mut ") + cur
            const_next = False
        else:
            cur = base[tk]
    return cur


def _rust_declarations(md_text):
    import re
    block = md_text[md_text.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    block = re.sub(r"//[^\n]*", "", block)
    out = {}
    for m in re.finditer(r"pub fn (dcn_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        args = [" ".join(a.split(":", 1)[1].split()) for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = (" ".join(m.group(3).split()) if m.group(3) else None, args)
    return out


def test_type_translator():
    assert _rust_of_c("const uint64_t *keys") == "*const u64"
    assert _rust_of_c("dcn_index **out") == "*mut *mut dcn_index"
    assert _rust_of_c("const dcn_index *const *inputs") == "*const *const dcn_index"
    assert _rust_of_c("dcn_ctx *const *ctxs") == "*const *mut dcn_ctx"
    assert _rust_of_c("uint64_t counters[DCN_N_STATS]") == "*mut u64"
    assert _rust_of_c("double stage_ms[DCN_N_STAGES]") == "*mut f64"
    assert _rust_of_c("void **out") == "*mut *mut c_void"
    assert _rust_of_c("const char *") == "*const c_char"
    assert _rust_of_c("float entropy_threshold") == "f32"
    assert _rust_of_c("int") == "c_int"


def test_integration_md_binds_every_export(dcn):
    root = os.path.dirname(os.path.dirname(dcn._native.HEADER_PATH))
    c = _c_declarations(open(dcn._native.HEADER_PATH).read())
    r = _rust_declarations(open(os.path.join(root, "INTEGRATION.md")).read())
    assert sorted(c) == dcn._native.declared_symbols()
    assert sorted(r) == sorted(c), {"unbound": sorted(set(c) - set(r)), "unknown": sorted(set(r) - set(c))}
    for name, (ret, args) in c.items():
        rret, rargs = r[name]
        assert rret == (None if ret == "void" else _rust_of_c(ret)), (name, ret, rret)
        assert rargs == [_rust_of_c(a) for a in args], (name, args, rargs)


def test_abi_version_of_header_library_and_bindings_agree(dcn):
    """DCN_ABI_MAJOR / DCN_ABI_MINOR of the header == what the built library reports == what the Python mirror was written
    against (it refuses to load a library of another major); INTEGRATION.md's Rust constants say the same."""
    import ctypes
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "deacon_hip.h")).read()
    major = int(re.search(r"#define DCN_ABI_MAJOR (\d+)", header).group(1))
    minor = int(re.search(r"#define DCN_ABI_MINOR (\d+)", header).group(1))
    a, b = ctypes.c_uint32(), ctypes.c_uint32()
    assert dcn._native.lib().dcn_abi_version(ctypes.byref(a), ctypes.byref(b)) == 0
    assert (a.value, b.value) == (major, minor) == tuple(dcn._native.ABI)
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    assert f"pub const DCN_ABI_MAJOR: u32 = {major};" in md and f"pub const DCN_ABI_MINOR: u32 = {minor};" in md
