"""The product path never touches the oracle or a CPU fallback: static check over lpi_amd/ sources."""
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_never_imports_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "lpi_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "lpi_oracle" in txt:
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def test_only_allowed_files_import_oracle():
    allowed = {"bench.py", "__graft_entry__.py"}
    for f in os.listdir(REPO):
        if f.endswith(".py") and f not in allowed:
            assert "lpi_oracle" not in open(os.path.join(REPO, f)).read(), f
