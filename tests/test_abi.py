"""The C-ABI library loads on a machine without a GPU and exports every symbol include/bpltv.h
declares (no compute calls here)."""
import ctypes as C
import os
import re
import pytest
from conftest import ROOT


def header_functions():
    txt = open(os.path.join(ROOT, "include", "bpltv.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bpltv_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from bpldenoising_amd import _lib
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "libbpltv.so does not export %s" % n
    assert sorted(_lib.SYMBOLS) == names          # the binding covers exactly the header
    assert lib.bpltv_version() == 4


def test_default_params_match_reference():
    """/root/reference/src/TVLearningFunctionVec.jl:33-43 and :14 (Dt)."""
    from bpldenoising_amd import _lib
    lib = _lib.load()
    p = _lib.BpltvParams()
    assert lib.bpltv_default_params(C.byref(p)) == 0
    assert (p.rho, p.tau0, p.accel, p.maxiter, p.delta_t) == (0.0, 5.0, 1, 5000, 1e-6)
    assert p.sigma0 == 0.99 / 5
    assert p.gap_tol == 0.0 and p.check_every == 0      # fixed iteration count, as the reference
    assert (p.init, p.order, p.opnorm) == (0, 0, 0.0)    # the restatement
    assert lib.bpltv_default_params(None) != 0


def test_struct_layout_matches_header(tmp_path):
    """ctypes mirrors == the C structs of include/bpltv.h (sizes and offsets via gcc)."""
    import subprocess
    from bpldenoising_amd import _lib
    src = tmp_path / "lay.c"
    fields_p = [f for f, _ in _lib.BpltvParams._fields_]
    fields_s = [f for f, _ in _lib.BpltvStats._fields_]
    body = ['#include <stdio.h>', '#include <stddef.h>', '#include "bpltv.h"', 'int main(void){',
            'printf("%zu %zu\\n", sizeof(bpltv_params), sizeof(bpltv_stats_t));']
    for f in fields_p:
        body.append('printf("%%zu\\n", offsetof(bpltv_params, %s));' % f)
    for f in fields_s:
        body.append('printf("%%zu\\n", offsetof(bpltv_stats_t, %s));' % f)
    body.append('return 0;}')
    src.write_text("\n".join(body))
    exe = tmp_path / "lay"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) == C.sizeof(_lib.BpltvParams) and int(out[1]) == C.sizeof(_lib.BpltvStats)
    offs = [int(x) for x in out[2:]]
    exp = [getattr(_lib.BpltvParams, f).offset for f in fields_p] + [getattr(_lib.BpltvStats, f).offset for f in fields_s]
    assert offs == exp


def test_create_without_gpu_fails_cleanly_or_succeeds():
    from bpldenoising_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.bpltv_create(None, 8, 8, 1, -1, 64) != 0
    assert lib.bpltv_create(C.byref(h), 0, 8, 1, -1, 64) != 0      # bad shape
    assert lib.bpltv_create(C.byref(h), 8, 8, 1, -1, 32) != 0      # only Float64
    rc = lib.bpltv_create(C.byref(h), 8, 8, 1, -1, 64)
    if rc == 0:
        assert lib.bpltv_destroy(h) == 0
    else:
        assert rc == 2                                             # BPLTV_E_HIP: no device
    assert lib.bpltv_last_error(None) == b"null handle"
    assert lib.bpltv_destroy(None) == 0


def test_missing_library_raises(monkeypatch, tmp_path):
    from bpldenoising_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_path_does_not_import_oracle():
    """Nothing under bpldenoising_amd/ may reference the oracle."""
    pkg = os.path.join(ROOT, "bpldenoising_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(d, f)).read()
                for line in txt.splitlines():
                    s = line.strip()
                    if s.startswith(("#", "//", "*", '"""')) or "oracle/" in s and ("//" in s or "#" in s):
                        continue
                    assert "import oracle" not in s and "from oracle" not in s and "bplo_" not in s, (f, s)


def test_plain_c_program_links_against_the_abi(tmp_path):
    """examples/c_abi_demo.c (plain C11, no Python, no torch) compiles against include/bpltv.h and
    links against libbpltv.so -- what a Julia ccall or any other FFI sees."""
    import subprocess
    from bpldenoising_amd import _lib
    exe = tmp_path / "c_abi_demo"
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-o", str(exe), "-L", libdir, "-lbpltv",
           "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined", "-lm"]
    subprocess.check_call(cmd)
    assert exe.exists()


def test_julia_glue_struct_matches_the_header():
    """integration/TVLearningFunctionHIP.jl restates `bpltv_params`: same fields, same order, and it only
    calls symbols the header declares."""
    import re
    hdr = open(os.path.join(ROOT, "include", "bpltv.h")).read()
    body = re.search(r"typedef struct bpltv_params \{(.*?)\} bpltv_params;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    c_fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1]
        c_fields += [re.sub(r"\[.*\]", "", n).strip() for n in names.split(",")]
    jl = open(os.path.join(ROOT, "integration", "TVLearningFunctionHIP.jl")).read()
    jbody = re.search(r"struct BpltvParams\n(.*?)\nend", jl, re.S).group(1)
    j_fields = re.findall(r"(\w+)::", jbody)
    assert j_fields == c_fields, (j_fields, c_fields)
    declared = set(re.findall(r"\b(bpltv_\w+)\s*\(", hdr))
    for sym in re.findall(r":(bpltv_\w+), libbpltv", jl):
        assert sym in declared, sym
