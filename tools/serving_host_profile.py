#!/usr/bin/env python3
"""cProfile of the saturated serving run (host side): where the engine loop's Python time goes.
usage: serving_host_profile.py [serving_bench.py arguments]"""
import cProfile
import pstats
import runpy
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.argv = [str(ROOT / "serving_bench.py")] + sys.argv[1:]
prof = cProfile.Profile()
try:
    prof.runcall(runpy.run_path, sys.argv[0], run_name="__main__")
except SystemExit:
    pass
st = pstats.Stats(prof)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(40)
