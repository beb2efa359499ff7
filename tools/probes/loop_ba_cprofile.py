#!/usr/bin/env python3
"""Where the host time of the bundle adjustment per keyframe goes: cProfile over the second of two runs of the device loop."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import run_slam_loop
run_slam_loop.run_device(60, bundle_adjust="keyframe")
pr = cProfile.Profile()
pr.enable()
out = run_slam_loop.run_device(60, bundle_adjust="keyframe")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
print({k: out[k] for k in ("fps", "keyframes")} if "fps" in out else out.keys())
print(out["bundle_adjust_per_keyframe"])
