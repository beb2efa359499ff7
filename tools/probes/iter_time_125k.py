import sys, os, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch, mqslam_amd
u, P, pts = mqslam_amd.synthetic.triangulation_problem(125000, 4)
ba = mqslam_amd.bundle_adjustment.make_benchmark_problem(u, P, pts + 0.01, torch.device("cuda", 0), seed=1)
r = []
for _ in range(4):
    r.append(round(1e3 * mqslam_amd.bundle_adjustment.time_iterations(ba, 300), 2))
print(os.environ.get("MQS_EXPERIMENT_SKIP_FINALIZE", "0"), r)
