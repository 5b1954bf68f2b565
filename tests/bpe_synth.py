"""The synthetic BPE merge table lives in the package now (lpi_amd/synth_bpe.py: bench.py's plugin_step record needs it too); the tests keep this name."""
from lpi_amd.synth_bpe import fuzz_cases, write_table  # noqa: F401
