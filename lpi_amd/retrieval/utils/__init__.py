# When this directory is used as the reference's ``retrieval/`` root (sys.path[0] = here, top-level ``utils`` / ``methods`` /
# ``models`` imports as in the reference's main.py / trainer.py), make the ``lpi_amd`` package importable as well.
import os
import sys

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)
