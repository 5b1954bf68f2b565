#!/usr/bin/env python3
"""Run bench.py against an alternate build of the library (A/B of kernel variants inside one gpurun call):
   python tools/bench_variant.py base|<suffix> [bench.py arguments...]      (suffix -> lpi_amd/csrc/variants/liblpi_hip_<suffix>.so)"""
import os, sys
sys.path.insert(0, os.getcwd())
import lpi_amd._lib as L
if sys.argv[1] != "base":
    L.LIB_PATH = os.path.join(os.getcwd(), "lpi_amd/csrc/variants/liblpi_hip_%s.so" % sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import runpy
runpy.run_path("bench.py", run_name="__main__")
