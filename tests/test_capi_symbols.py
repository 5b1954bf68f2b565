"""The C-ABI library loads on a CPU-only machine and exports every symbol include/lpi_hip.h declares; the ctypes table
covers exactly that set.  No compute call is made (no GPU here)."""
import os
import re
import subprocess

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


def test_plain_c_consumer(tmp_path):
    """include/lpi_hip.h is plain C (gcc -std=c99 -pedantic-errors) and a C program links liblpi_hip.so and calls it with no Python / torch / C++ in the
    process: the boundary another host language's FFI would bind.  Host-side and predicate entry points only — no GPU here."""
    lib_dir = os.path.join(REPO, "lpi_amd", "csrc")
    exe = str(tmp_path / "abi_consumer")
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic-errors", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "tests", "c", "abi_consumer.c"),
                         "-o", exe, "-L", lib_dir, "-llpi_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, (run.stdout, run.stderr)
    assert f"lpi_version={_lib.EXPECTED_ABI} " in run.stdout and "rows_ok=1 rows_bad=0 spool_ok=1 spool_long=0" in run.stdout, run.stdout
