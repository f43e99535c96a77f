#!/usr/bin/env python3
"""cProfile of tools/bench_extract.py (host-side cost of the extraction loop): top functions by cumulative time."""
import cProfile, os, pstats, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
sys.argv = [os.path.join(ROOT, "tools", "bench_extract.py")] + sys.argv[2:]
cProfile.run('runpy.run_path(sys.argv[0], run_name="__main__")', out + ".prof")
with open(out, "w") as f:
    pstats.Stats(out + ".prof", stream=f).sort_stats("cumulative").print_stats(70)
