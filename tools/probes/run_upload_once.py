#!/usr/bin/env python3
"""One warmed run of the example sequence with the frames arriving inside the loop (for kernel / copy traces):
python tools/probes/run_upload_once.py [frames=80] [pageable|pinned]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import run_icl_nuim as R
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 80
up = sys.argv[2] if len(sys.argv) > 2 else "pageable"
R.run(frames, upload=up)
print(json.dumps({k: v for k, v in R.run(frames, upload=up).items() if k in ("frames", "frame_ingest", "frames_per_s")}))
