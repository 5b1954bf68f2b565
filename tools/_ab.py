# run bench.py with an alternate library: python tools/_ab.py <suffix|base> [bench args...]
import os, sys
sys.path.insert(0, os.getcwd())
import lpi_amd._lib as L
if sys.argv[1] != "base":
    L.LIB_PATH = os.path.join(os.getcwd(), "lpi_amd/csrc/liblpi_hip_%s.so" % sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import runpy
runpy.run_path("bench.py", run_name="__main__")
