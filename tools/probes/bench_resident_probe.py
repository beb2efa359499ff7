import os, sys, time, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import mqslam_amd
SD = mqslam_amd.slam_device.DeviceMonoSlam
orig = SD._img_ptr
acc = {"t": 0.0, "n": 0}
def timed(self, img, shape, sync=True):
    t = time.perf_counter(); r = orig(self, img, shape, sync); acc["t"] += time.perf_counter() - t; acc["n"] += 1; return r
SD._img_ptr = timed
import run_icl_nuim as R
R.run(80)
def leg(tag):
    for up in (None, "pageable"):
        acc["t"] = 0; acc["n"] = 0
        r = max((R.run(80, upload=up) for _ in range(3)), key=lambda r: r["frames_per_s"])
        print(tag, up, r["frames_per_s"], "img_ptr us/call", round(1e6 * acc["t"] / max(1, acc["n"]), 2), flush=True)
leg("fresh")
# what bench does before: the step legs
import subprocess
sys.argv = ["bench.py", "--no-frontend", "--no-cpu-baseline", "--no-asymptote", "--steps", "10", "--settle-steps", "10"]
import runpy
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
leg("after bench legs")
