"""CPU: the C-ABI library loads and exports every symbol include/s2f.h declares, and the ctypes table in
spike2former_amd/_lib.py has the same arity (no compute calls here -- there is no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "s2f.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|int64_t|void\s*\*?|const char\s*\*)\s*(s2f_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def test_header_declares_the_path():
    d = _declared()
    for need in ("s2f_lif_fwd", "s2f_lif_bwd", "s2f_lif_seq_fwd", "s2f_lif_seq_bwd", "s2f_sdsa_fwd", "s2f_sdsa_bwd",
                 "s2f_dcnv3_fwd", "s2f_dcnv3_bwd", "s2f_version", "s2f_last_error"):
        assert need in d


def test_library_exports_every_declared_symbol():
    path = os.path.join(ROOT, "spike2former_amd", "libs2f_hip.so")
    assert os.path.exists(path), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(path)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/s2f.h but not exported"
    lib.s2f_version.restype = ctypes.c_int
    want = int(re.search(r"#define S2F_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "s2f.h")).read()).group(1))
    assert lib.s2f_version() == want
    lib.s2f_lif_mask_words.restype = ctypes.c_int64
    lib.s2f_lif_mask_words.argtypes = [ctypes.c_int64]
    assert [lib.s2f_lif_mask_words(n) for n in (0, 1, 256, 257, 1024)] == [0, 4, 4, 8, 16]


def test_ctypes_table_matches_header():
    from spike2former_amd import _lib
    d = _declared()
    assert set(_lib.SIGNATURES) == set(d)
    for name, (_, args) in _lib.SIGNATURES.items():
        assert len(args) == d[name], f"{name}: ctypes has {len(args)} args, header has {d[name]}"


def test_argument_errors_are_reported_without_a_gpu():
    """Validation happens before any launch, so the error convention can be checked on CPU."""
    from spike2former_amd._lib import lib
    assert lib.s2f_lif_fwd(None, None, None, None, None, None, None, 16, 1.0, 8, 0, None) == -1
    assert b"null" in lib.s2f_last_error()
    assert lib.s2f_dcnv3_fwd(1, 1, 1, 1, 0, 4, 4, 1, 4, 3, 3, 1, 1, 1, 1, 1, 1, 1.0, None) == -1
    assert lib.s2f_sdsa_fwd(1, 1, 1, 1, 1, 1, 8, 65, 4, 4, 1.0, None) == -1     # head dim > 64
