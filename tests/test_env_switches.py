"""The SEVEN environment switches the package reads (round 6: the rest became EngineOptions fields, command-line flags or were deleted), each exercised at a
non-default value on the CPU: LPI_LIB and LPI_TUNING (the library loader: a child process, since a library is loaded once per process), LPI_TOKENIZER,
LPI_BPE_VOCAB, and the three engine fall-backs LPI_RESIDUAL / LPI_LN_FOLD / LPI_ROWSTATS through EngineOptions.from_env (their GPU behaviour:
tests/test_model_gpu.py).  And that there ARE only seven."""
import os
import re
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, **env):
    e = dict(os.environ)
    e.update(env)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, cwd=REPO)


def test_only_seven_environment_switches_are_read_by_the_package():
    names = set()
    for root, _, files in os.walk(os.path.join(REPO, "lpi_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                for m in re.finditer(r"environ(?:\.get\(|\[)\s*[\"'](LPI_[A-Z0-9_]+)[\"']", src):
                    names.add(m.group(1))
                for m in re.finditer(r"[\"'](LPI_[A-Z0-9_]+)[\"']\s+in\s+_?os\.environ", src):
                    names.add(m.group(1))
                for m in re.finditer(r"\(\(?[\"'](LPI_[A-Z0-9_]+)[\"'],\s*[\"'][a-z_]+[\"']\)", src):      # from_env's (name, field) table
                    names.add(m.group(1))
    assert names == {"LPI_LIB", "LPI_TUNING", "LPI_BPE_VOCAB", "LPI_TOKENIZER", "LPI_RESIDUAL", "LPI_LN_FOLD", "LPI_ROWSTATS"}, sorted(names)
    bench = open(os.path.join(REPO, "bench.py")).read()
    assert not re.findall(r"environ\.get\(\s*[\"']LPI_", bench), "bench.py's switches are command-line flags"


def test_lpi_lib_names_the_library_and_a_missing_one_fails_loudly(tmp_path):
    code = "from lpi_amd import _lib; L = _lib.load(); print('OK', _lib.LIB_PATH, L.lpi_version())"
    lib = os.path.join(REPO, "lpi_amd", "csrc", "liblpi_hip.so")
    copy = tmp_path / "elsewhere.so"
    copy.write_bytes(open(lib, "rb").read())
    p = _run(code, LPI_LIB=str(copy))
    assert p.returncode == 0 and f"OK {copy}" in p.stdout, p.stderr[-500:]
    p = _run(code, LPI_LIB=str(tmp_path / "missing.so"))
    assert p.returncode != 0 and "not found" in p.stderr and "no CPU fallback" in p.stderr


def test_lpi_tuning_sets_library_knobs_at_load():
    code = "from lpi_amd import _lib; L = _lib.load(); print('K', L.lpi_get_tuning(5), L.lpi_get_tuning(15))"
    p = _run(code, LPI_TUNING="5=96,15=-1")
    assert p.returncode == 0 and "K 96 -1" in p.stdout, p.stderr[-500:]
    p = _run(code)
    assert "K 160 0" in p.stdout
    p = _run(code, LPI_TUNING="99=1")
    assert p.returncode != 0 and "bad knob" in p.stderr


def test_lpi_tokenizer_selects_the_python_bpe(tmp_path, monkeypatch):
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import bpe_synth
    from lpi_amd.retrieval.models.clip import prompt_learner as PL
    from lpi_amd.retrieval.models.clip.simple_tokenizer import NativeTokenizer, SimpleTokenizer
    monkeypatch.setenv("LPI_BPE_VOCAB", bpe_synth.write_table(tmp_path / "t.txt.gz", seed=5))      # (LPI_BPE_VOCAB: the merge table's path)
    for value, cls in (("python", SimpleTokenizer), ("native", NativeTokenizer)):
        monkeypatch.setenv("LPI_TOKENIZER", value)
        monkeypatch.setattr(PL, "_tokenizer", None)
        tk = PL.get_tokenizer()
        assert type(tk) is cls
        assert tk.encode("a photo of two dogs") == SimpleTokenizer(os.environ["LPI_BPE_VOCAB"]).encode("a photo of two dogs")
    monkeypatch.setattr(PL, "_tokenizer", None)


def test_engine_options_from_env(monkeypatch):
    from lpi_amd.engine import EngineOptions
    for k in ("LPI_RESIDUAL", "LPI_LN_FOLD", "LPI_ROWSTATS"):
        monkeypatch.delenv(k, raising=False)
    assert EngineOptions.from_env() == EngineOptions()
    monkeypatch.setenv("LPI_RESIDUAL", "f32")
    monkeypatch.setenv("LPI_LN_FOLD", "1")
    monkeypatch.setenv("LPI_ROWSTATS", "0")
    o = EngineOptions.from_env(pooled_last=False)
    assert (o.residual_f16, o.ln_fold, o.rowstats, o.pooled_last, o.stream_pool) == (False, 1, 0, False, True)
    assert EngineOptions.from_env(ln_fold=2).ln_fold == 2          # an explicit argument wins over the environment
    monkeypatch.setenv("LPI_RESIDUAL", "bf16")
    with pytest.raises(ValueError):
        EngineOptions.from_env()
    monkeypatch.setenv("LPI_RESIDUAL", "f16")
    monkeypatch.setenv("LPI_LN_FOLD", "3")
    with pytest.raises(ValueError):
        EngineOptions.from_env()
