#!/usr/bin/env python3
"""Round 6: run-to-run spread of the loop legs inside one process (frames resident / arriving in ordinary memory), and what the box gives the
process to run on.    python tools/probes/loop_variability_probe.py [repeats=8]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import run_icl_nuim as R
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 8
print(json.dumps({"cpus_affinity": len(os.sched_getaffinity(0)), "cpu_count": os.cpu_count(),
                  "cpu_max": open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None}), flush=True)
R.run(80)
for rnd in range(2):
    for up in (None, "pageable"):
        fps = [R.run(80, upload=up)["frames_per_s"] for _ in range(rep)]
        print(json.dumps({"round": rnd, "upload": up, "frames_per_s": fps}), flush=True)
