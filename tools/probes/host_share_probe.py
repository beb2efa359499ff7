#!/usr/bin/env python3
"""Round 6: where a frame's wall time goes on the host -- inside mqs_slam_track (enqueue + wait) against the interpreter around it.
python tools/probes/host_share_probe.py [frames=200]"""
import os, sys, json, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import run_icl_nuim as R
import mqslam_amd
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
R.run(80, None, 0)
cls = mqslam_amd.slam_device.DeviceMonoSlam
orig_init = cls.__init__
acc = {"track_s": 0.0, "calls": 0}
def patched(self, *a, **k):
    orig_init(self, *a, **k)
    f = self._track
    def timed(*args):
        t = time.perf_counter(); rc = f(*args); acc["track_s"] += time.perf_counter() - t; acc["calls"] += 1
        return rc
    self._track = timed
cls.__init__ = patched
for pipe in (False, True):
    acc["track_s"] = 0.0; acc["calls"] = 0
    r = R.run(frames, None, 0, pipeline=pipe)
    per = 1e6 / r["frames_per_s"]
    print(json.dumps({"pipeline": pipe, "frames_per_s": r["frames_per_s"], "us_per_frame": round(per, 1),
                      "us_in_mqs_slam_track": round(1e6 * acc["track_s"] / max(1, acc["calls"]), 1)}))
