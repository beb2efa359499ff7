#!/usr/bin/env python3
"""cProfile of the rendered detect-track-pose-triangulate loop (tools/run_slam_loop.py's scene): where the host time of a
frame goes.  python tools/profile_slam_loop.py [frames]"""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import run_slam_loop

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
run_slam_loop.run(frames=frames)                       # warm
pr = cProfile.Profile()
pr.enable()
out = run_slam_loop.run(frames=frames)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
print({k: out[k] for k in ("frames", "frames_per_s")})
