"""The C-ABI library loads on a CPU-only machine and exports every symbol include/lpi_hip.h declares; the ctypes table
covers exactly that set.  No compute call is made (no GPU here)."""
import os
import re

from lpi_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(REPO, "include", "lpi_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(lpi_[a-z0-9_]+)\s*\(", txt))


def test_header_and_binding_agree():
    syms = header_symbols()
    assert len(syms) >= 25
    assert syms == set(_lib.SIGNATURES), syms ^ set(_lib.SIGNATURES)


def test_library_exports_every_symbol():
    lib = _lib.load()
    for s in header_symbols():
        assert hasattr(lib, s), s
    assert lib.lpi_version() == _lib.EXPECTED_ABI          # a stale build must not load (load() raises on a mismatch)
    assert _lib.launch_count() >= 0


def test_argument_counts_match_header():
    txt = open(os.path.join(REPO, "include", "lpi_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for name, params in re.findall(r"\b(lpi_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
        n = 0 if params.strip() in ("", "void") else params.count(",") + 1
        assert n == len(_lib.SIGNATURES[name]), (name, n, len(_lib.SIGNATURES[name]))
