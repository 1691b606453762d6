#!/usr/bin/env python3
"""Development: bench.py on an A/B build of the library (scripts/build_variant.py):  python scripts/dev/bench_lib.py <lib.so | -> [bench.py arguments]"""
import runpy
import sys
from pathlib import Path

root = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(root))
lib = sys.argv[1]
if lib != "-":
    from mtgs_amd import _lib
    _lib.use_library(lib)
sys.argv = [str(root / "bench.py")] + sys.argv[2:]
runpy.run_path(str(root / "bench.py"), run_name="__main__")
