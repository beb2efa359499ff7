#!/usr/bin/env python3
"""cProfile of the end-to-end loop on the rendered sequence (where the host time of a frame goes)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import run_slam_loop
run_slam_loop.run(20)
pr = cProfile.Profile()
pr.enable()
out = run_slam_loop.run(60)
pr.disable()
print(out["frames_per_s"])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print("\n".join(l for l in s.getvalue().splitlines() if "synthetic.py" not in l and "render" not in l)[:6000])
