"""The C-ABI library builds for gfx950, loads, and exports every symbol include/graphik_amd.h
declares.  No compute (no GPU needed)."""
import os
import re

import pytest

from conftest import REPO


def test_header_symbols_are_bound_and_exported():
    from graphik_amd import _ffi, build
    build.build()
    hdr = open(os.path.join(REPO, "include", "graphik_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gik_[a-z_]+)\s*\(", hdr))
    assert declared == set(_ffi.SYMBOLS), declared ^ set(_ffi.SYMBOLS)
    L = _ffi.lib()
    for name in declared:
        assert hasattr(L, name)
    assert L.gik_abi_version() == _ffi.ABI_VERSION


def test_struct_layouts_match_header():
    import ctypes as C
    from graphik_amd import _ffi
    # gik_stats: the header says 48 bytes; parse its field list and compare with the ctypes mirror
    hdr = open(os.path.join(REPO, "include", "graphik_amd.h")).read()
    body = re.search(r"typedef struct \{([^}]*)\} gik_stats;", hdr).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(double|int32_t)\s+(\w+)\s*;", body)
    assert [n for _, n in fields] == [n for n, _ in _ffi.Stats._fields_]
    assert sum(8 if t == "double" else 4 for t, _ in fields) == 48 == C.sizeof(_ffi.Stats)
    assert _ffi.STATS_I32["inner_executed"] == 8 and _ffi.STATS_F64["gradnorm"] == 1 and _ffi.STATS_F64["stepsize"] == 5
    # descriptor structs: same fields in the same order with the same scalar types as the header
    ctype_of = {"double": C.c_double, "int32_t": C.c_int32}
    for cname, mirror in (("gik_template_desc", _ffi.TemplateDesc), ("gik_pipeline_desc", _ffi.PipelineDesc),
                          ("gik_prepare_diag", _ffi.PrepareDiag)):
        nocomment = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        body = re.search(r"typedef struct \{([^}]*)\} %s;" % cname, nocomment).group(1)
        decl = []
        for typ, names in re.findall(r"\b(?:const\s+)?(double|int32_t)\s+([^;]+);", body):
            for nm in names.split(","):
                nm = nm.strip()
                decl.append((nm.lstrip("*"), "ptr" if nm.startswith("*") else typ))
        assert [n for n, _ in decl] == [n for n, _ in mirror._fields_], (cname, decl)
        for (n, typ), (_, ct) in zip(decl, mirror._fields_):
            if typ == "ptr":
                assert C.sizeof(ct) == C.sizeof(C.c_void_p), (cname, n)
            else:
                assert ct is ctype_of[typ], (cname, n)
    assert C.sizeof(_ffi.Trace) == 8 + 6 * 8
    assert _ffi.TemplateDesc.N.offset == 4 and _ffi.TemplateDesc.term_i.offset == 16


def test_no_cpu_fallback_without_gpu():
    """Creating an engine object without a HIP device must fail loudly."""
    import torch
    from graphik_amd import _ffi
    from graphik_amd.engine import Template
    import numpy as np
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    om = np.zeros((4, 4))
    om[0, 1] = om[1, 0] = 1
    with pytest.raises(_ffi.GikError):
        Template.from_matrices(om, k=3, use_limits=False)


def test_product_does_not_import_oracle():
    """Nothing under graphik_amd/ may reference the CPU oracle."""
    pkg = os.path.join(REPO, "graphik_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and \
                    "gik_oracle" not in txt, f
