"""CPU: the C-ABI library loads and exports every symbol include/snag_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "snag_hip.h")).read()
    return sorted(set(re.findall(r"SNAG_API\s+[\w\s\*]+?\b(snag_\w+)\s*\(", text)))


def test_header_declares_something():
    names = _declared()
    assert "snag_score_margin_fwd" in names and "snag_last_error" in names


def test_library_exports_every_declared_symbol():
    from snag_amd import _lib
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/snag_hip.h but not exported"
    assert lib.snag_version() == _lib.ABI_VERSION == 3
    # every declared entry point has a ctypes signature on the Python side
    bound = set(_lib.SIGNATURES) | set(_lib._SPECIAL)
    assert set(_declared()) == bound, sorted(set(_declared()) ^ bound)


def test_argument_refusal_needs_no_gpu():
    """NULL pointers are refused before any HIP call, with a message."""
    from snag_amd import _lib
    lib = _lib.load()
    rc = lib.snag_score_margin_fwd(None, 128, 10, None, 2, 128, None, None, None, None, 4, 8, 4, 1.0, None, None, None, None)
    assert rc == -1
    assert "null" in _lib.last_error().lower()


def test_ops_refuse_cpu_tensors():
    import torch
    from snag_amd import ops, _lib
    e = torch.zeros(4, 8)
    i = torch.zeros(2, dtype=torch.int64)
    with pytest.raises(_lib.SnagHipError):
        ops.score_margin_fwd(e, e, i, i, i, None, 1.0)


def _prototypes():
    """name -> list of C parameter types, parsed from include/snag_hip.h."""
    text = open(os.path.join(ROOT, "include", "snag_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"SNAG_API\s+([\w\s\*]+?)\b(snag_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = [a.strip() for a in m.group(3).replace("\n", " ").split(",")]
        if args == ["void"] or args == [""]:
            args = []
        types = []
        for a in args:
            a = re.sub(r"\b\w+$", "", a).strip() if not a.endswith("*") else a        # drop the parameter name
            types.append(re.sub(r"\s+", " ", a))
        out[m.group(2)] = (m.group(1).strip(), types)
    return out


def test_ctypes_signatures_match_the_header_prototypes():
    """Argument count and class (pointer / int / int64 / uint64 / float) of every ctypes binding equal the C prototype: a drifted
    binding would corrupt arguments silently."""
    from snag_amd import _lib

    def cls(ctype_or_c):
        if isinstance(ctype_or_c, str):
            t = ctype_or_c
            if "*" in t:
                return "ptr"
            t = t.replace("const", "").strip()
            return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "float": "f32", "double": "f64"}[t]
        c = ctype_or_c
        if c in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(c, "_type_") and isinstance(c._type_, type):
            return "ptr"
        return {ctypes.c_int: "i32", ctypes.c_int32: "i32", ctypes.c_int64: "i64", ctypes.c_uint64: "u64", ctypes.c_float: "f32",
                ctypes.c_double: "f64"}[c]

    protos = _prototypes()
    table = dict(_lib.SIGNATURES)
    table.update({k: v[0] for k, v in _lib._SPECIAL.items()})
    assert set(protos) == set(table)
    for name, (ret, ctypes_list) in protos.items():
        bound = table[name]
        assert len(bound) == len(ctypes_list), f"{name}: header has {len(ctypes_list)} parameters, the binding {len(bound)}"
        for i, (c_t, b_t) in enumerate(zip(ctypes_list, bound)):
            assert cls(c_t) == cls(b_t), f"{name} argument {i}: header '{c_t}' vs binding {b_t}"


def _structs():
    """struct name -> [(field, class)] in declaration order, parsed from include/snag_hip.h."""
    text = open(os.path.join(ROOT, "include", "snag_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct (snag_\w+) \{(.*?)\} \1;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r"((?:const\s+)?[\w]+(?:\s*\*)?)\s*(.*)", decl)
            base, names = mm.group(1).strip(), mm.group(2)
            for nm in names.split(","):
                nm = nm.strip()
                ptr = "*" in base or nm.startswith("*")
                nm = nm.lstrip("* ")
                arr = re.match(r"(\w+)\[(\d+)\]$", nm)           # a fixed-size array member: kind[n]
                if arr:
                    nm = arr.group(1)
                if ptr:
                    kind = "ptr"
                elif base.replace("const", "").strip().startswith("snag_"):
                    kind = "struct"
                else:
                    kind = {"int32_t": "i32", "int": "i32", "int64_t": "i64", "uint64_t": "u64", "float": "f32"}[base.replace("const", "").strip()]
                fields.append((nm, f"{kind}[{arr.group(2)}]" if arr else kind))
        out[m.group(1)] = fields
    return out


def test_ctypes_structures_match_the_header_structs():
    from snag_amd import _lib
    pairs = {"snag_gemm_desc": _lib.GemmDesc, "snag_fusion_layer": _lib.FusionLayer, "snag_fusion_layer_grad": _lib.FusionLayerGrad,
             "snag_fusion_desc": _lib.FusionDesc, "snag_fusion_bwd_desc": _lib.FusionBwdDesc, "snag_mask_region": _lib.MaskRegion, "snag_small_mm": _lib.SmallMM,
             "snag_ln_bwd_tail": _lib.LnBwdTail}
    structs = _structs()
    assert set(structs) == set(pairs)

    def kind(c):
        if isinstance(c, type) and issubclass(c, ctypes.Array):
            return f"{kind(c._type_)}[{c._length_}]"
        if isinstance(c, type) and issubclass(c, ctypes.Structure):
            return "struct"
        if c in (ctypes.c_void_p,) or hasattr(c, "contents") or (hasattr(c, "_type_") and not isinstance(c._type_, str)):
            return "ptr"
        return {ctypes.c_int32: "i32", ctypes.c_int: "i32", ctypes.c_int64: "i64", ctypes.c_uint64: "u64", ctypes.c_float: "f32"}[c]

    for name, cls in pairs.items():
        got = [(n, kind(t)) for n, t in cls._fields_]
        assert got == structs[name], f"{name}:\n header  {structs[name]}\n binding {got}"


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """The boundary is a C ABI: include/snag_hip.h compiles as C99 (-pedantic), and a C program linked against the shared library
    calls it (argument refusal needs no GPU)."""
    import shutil
    import subprocess
    from snag_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    _lib.load()
    src = tmp_path / "client.c"
    src.write_text(
        '#include "snag_hip.h"\n#include <stdio.h>\n#include <string.h>\n'
        "int main(void) {\n"
        "    if (snag_version() != SNAG_ABI_VERSION) return 2;\n"
        "    int rc = snag_score_margin_fwd(0, 128, 10, 0, 2, 128, 0, 0, 0, 0, 4, 8, 4, 1.0f, 0, 0, 0, 0);\n"
        "    if (rc != SNAG_ERR_ARG || strlen(snag_last_error()) == 0) return 3;\n"
        "    snag_gemm_desc d; memset(&d, 0, sizeof d); d.M = -1;\n"
        "    if (snag_gemm_f32(&d, 0) != SNAG_ERR_ARG) return 4;\n"
        '    printf("%s\\n", snag_last_error());\n'
        "    return 0;\n}\n")
    exe = tmp_path / "client"
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L", libdir, "-l:" + os.path.basename(_lib.LIB_PATH), f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "gemm" in r.stdout


def test_gemm_k_zero_does_not_crash_the_host():
    """K = 0 passes gemm_check (C = epilogue(0)); the dispatcher's split arithmetic once divided by zero on the HOST (SIGFPE, the
    process died).  Without a GPU the call must come back with the launch error, not kill the interpreter."""
    from snag_amd import _lib
    lib = _lib.load()
    assert lib.snag_gemm_workspace_floats(70, 132, 0) == 0
    buf = (ctypes.c_float * 64)()
    for with_ws in (False, True):
        d = _lib.GemmDesc()
        d.M, d.N, d.K, d.a_kcontig, d.b_kcontig = 4, 4, 0, 1, 1
        d.C, d.ldc, d.alpha = ctypes.addressof(buf), 4, 1.0
        if with_ws:
            d.workspace, d.workspace_floats = ctypes.addressof(buf), 64
        rc = lib.snag_gemm_f32(ctypes.byref(d), None)
        assert rc in (0, -3), (rc, _lib.last_error())


def test_ticket_words_constant_matches_header():
    from snag_amd import _lib
    text = open(os.path.join(ROOT, "include", "snag_hip.h")).read()
    assert int(re.search(r"#define\s+SNAG_SCORE_TICKET_WORDS\s+(\d+)", text).group(1)) == _lib.SCORE_TICKET_WORDS


def test_a_library_of_another_abi_version_is_refused(monkeypatch):
    """a stale or variant .so (SNAG_HIP_LIB) must not be read with this binding's struct offsets."""
    from snag_amd import _lib
    _lib.load()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", 99)
    with pytest.raises(_lib.SnagHipError, match="ABI version"):
        _lib.load()
