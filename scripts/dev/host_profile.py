#!/usr/bin/env python3
"""Development: where the HOST time of the eager MTGS-like iteration goes (cProfile over 30 iterations, GPU work asynchronous)."""
import cProfile
import pstats
import runpy
import sys
from pathlib import Path

root = Path(__file__).resolve().parents[2]
sys.argv = [str(root / "scripts" / "mtgs_like_train.py"), "--shipped", "--only", "fused", "--reps", "30", "--visfirst", "--optimizer", "fused",
            "--row-lazy"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit:
    pass
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(r"mtgs_amd|mtgs_like_train|autograd|torch.empty|torch.zeros|_lib", 60)
