"""CPU tests of the drop-in boundary: libsavont_hip.so loads without a GPU, exports exactly the symbols
include/savont_hip.h declares, and fails loudly (no CPU fallback) when asked to compute without a device."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "savont_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(svt_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported():
    from savont_amd import hip
    L = hip.load()
    names = _header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "libsavont_hip.so does not export %s" % n
    assert sorted(hip.SYMBOLS) == names


def test_only_c_abi_symbols_leak():
    """the dynamic symbol table of libsavont_hip.so IS the header: every defined symbol (functions, data, weak, kernel handles) is an
    extern "C" svt_* prototype of include/savont_hip.h and vice versa (-fvisibility=hidden + csrc/exports_hip.map)"""
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "savont_amd", "libsavont_hip.so")]).decode()
    rows = [l.split() for l in out.splitlines() if l.strip()]
    assert sorted(r[-1] for r in rows) == _header_symbols(), sorted(set(r[-1] for r in rows) ^ set(_header_symbols()))
    assert all(r[-2] == "T" for r in rows), [r for r in rows if r[-2] != "T"]
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "savont_amd", "libsavont_asv.so")]).decode()
    assert all(l.split()[-2] == "T" and l.split()[-1].startswith("svh_") for l in out.splitlines() if l.strip())


def test_no_device_is_a_loud_error():
    from savont_amd import hip
    L = hip.load()
    if L.svt_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert L.svt_create(0, C.byref(h)) == hip.SVT_ERR_NODEVICE and not h.value
    with pytest.raises(hip.SavontHipError):
        hip.Device(0)
    from savont_amd.pipeline import AsvPipeline
    with pytest.raises(hip.SavontHipError):
        AsvPipeline(0)


def test_product_never_references_the_oracle():
    """nothing under savont_amd/ or include/ may import, link or name the oracle"""
    bad = []
    for base in ("savont_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in d or "__pycache__" in d:
                continue
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                txt = open(os.path.join(d, f), errors="ignore").read()
                for m in re.finditer(r"(oracle_lib|savont_oracle|libsavont_oracle|orc_[a-z_]+\()", txt):
                    bad.append((f, m.group(0)))
    assert not bad, bad
    ldd = subprocess.check_output(["ldd", os.path.join(ROOT, "savont_amd", "libsavont_asv.so")]).decode()
    assert "oracle" not in ldd


def test_integration_rust_stub_matches_header():
    """the Rust `extern "C"` block INTEGRATION.md shows a savont maintainer is the header's: same symbols, arity, parameter and
    return types (a stale stub is undefined behaviour on the Rust side)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_stub as g
    header = open(os.path.join(ROOT, "include", "savont_hip.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    shown = doc[doc.index(g.BEGIN):doc.index(g.END)]
    got = g.rust_prototypes(shown)
    want = {name: ([g.rust_type(t) for t, _ in params], None if ret == "void" else g.rust_type(ret)) for name, ret, params in g.c_prototypes(header)}
    assert sorted(want) == _header_symbols()                      # the parser sees every prototype the symbol scan sees
    assert sorted(got) == sorted(want), sorted(set(got) ^ set(want))
    for name in want:
        assert got[name] == want[name], (name, got[name], want[name])
    assert "remaining svt_" not in doc                              # no elision
    # the opaque / plain structs of the header are declared
    for st in g.OPAQUE + g.STRUCTS:
        assert "pub struct %s" % st in shown
    # svt_seeds_out: field order of the Rust struct == field order of the C struct
    cfields = re.findall(r"\*\s*(\w+);", re.search(r"typedef struct svt_seeds_out \{(.*?)\} svt_seeds_out;", header, flags=re.S).group(1))
    rfields = re.findall(r"pub (\w+): \*mut", re.search(r"pub struct svt_seeds_out \{(.*?)\n\}", shown, flags=re.S).group(1))
    assert cfields == rfields


def test_library_reads_no_kernel_selection_environment():
    """kernel / engine selection is svt_set_option / svh_set_option state; the only environment the libraries read is SAVONT_TRACE and
    SAVONT_SAMPLE (diagnostics: host timers, the development CPU sampler), SAVONT_THREADS and LOCAL_WORLD_SIZE (worker-pool size)"""
    seen = set()
    for d, _, files in os.walk(os.path.join(ROOT, "savont_amd", "csrc")):
        if "build" in d:
            continue
        for f in files:
            seen |= set(re.findall(r'getenv\("([A-Z_]+)"\)', open(os.path.join(d, f), errors="ignore").read()))
    assert seen <= {"SAVONT_TRACE", "SAVONT_SAMPLE", "SAVONT_THREADS", "LOCAL_WORLD_SIZE"}, seen


# ---- the stage-level boundary: include/savont_asv.h <-> libsavont_asv.so <-> savont_amd/pipeline.py <-> INTEGRATION.md ----
def _asv_header():
    return open(os.path.join(ROOT, "include", "savont_asv.h")).read()


def test_stage_header_declares_exactly_what_the_library_exports():
    """every svh_* symbol of libsavont_asv.so is declared in include/savont_asv.h (citing the src/main.rs edge it stands for) and vice versa"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_stub as g
    protos = g.c_prototypes(_asv_header(), "svh_")
    names = sorted(p[0] for p in protos)
    assert len(names) == len(set(names)) and len(names) >= 80
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "savont_amd", "libsavont_asv.so")]).decode()
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T svh_" in l)
    assert exported == names, sorted(set(exported) ^ set(names))
    hdr = _asv_header()
    for edge in ("src/main.rs:501", "src/main.rs:520", "src/main.rs:537", "src/main.rs:83,87", "src/main.rs:142"):      # B1-B5 of SURVEY.md 8b
        assert edge in hdr, edge


def test_ctypes_table_matches_the_stage_header():
    """savont_amd/pipeline.py binds the stage entry points by hand: same arity, pointer-ness and integer widths as the header"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_stub as g
    from savont_amd import pipeline
    L = pipeline.load()
    width = {"int": 4, "int32_t": 4, "uint32_t": 4, "uint64_t": 8, "int64_t": 8, "double": 8, "uint8_t": 1}
    checked = 0
    for name, ret, params in g.c_prototypes(_asv_header(), "svh_"):
        fn = getattr(L, name)
        if fn.argtypes is None:
            continue                                                    # not bound by the Python harness (ctypes defaults are never used for it)
        assert len(fn.argtypes) == len(params), (name, len(fn.argtypes), len(params))
        for at, (ct, pn) in zip(fn.argtypes, params):
            if "*" in ct:
                assert at in (C.c_void_p, C.c_char_p) or hasattr(at, "_type_"), (name, pn, at)       # a pointer of some kind
            else:
                assert C.sizeof(at) == width[ct.replace("const ", "").strip()], (name, pn, at, ct)
                assert (at is C.c_double) == (ct.strip() == "double"), (name, pn)
        checked += 1
    assert checked >= 70


def test_integration_rust_stub_matches_stage_header():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_stub as g
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    shown = doc[doc.index(g.BEGIN_ASV):doc.index(g.END_ASV)]
    got = g.rust_prototypes(shown, "svh_")
    want = {name: ([g.rust_type(t) for t, _ in params], None if ret == "void" else g.rust_type(ret)) for name, ret, params in g.c_prototypes(_asv_header(), "svh_")}
    assert sorted(got) == sorted(want), sorted(set(got) ^ set(want))
    for name in want:
        assert got[name] == want[name], (name, got[name], want[name])
    # svh_args: field order and types of the Rust struct == the C struct
    rfields = re.findall(r"pub (\w+): (\w+),", re.search(r"pub struct svh_args \{(.*?)\n\}", shown, flags=re.S).group(1))
    assert rfields == [(n, g.SCALAR[t]) for t, n in g.svh_args_fields(_asv_header())]


def test_pipeline_loads_without_torch():
    """pipeline.load() binds every stage entry point, the pooled ones included, without importing torch"""
    code = "import sys; sys.modules['torch'] = None; from savont_amd import pipeline; L = pipeline.load(); assert L.svh_em_finish.argtypes is not None; print('ok')"
    out = subprocess.check_output(["python", "-c", code], cwd=ROOT).decode()
    assert out.strip().endswith("ok")
