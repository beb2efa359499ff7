import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/tools") else os.getcwd())
import mqslam_amd
N = int(sys.argv[1])
u, P, pts = mqslam_amd.synthetic.triangulation_problem(N, 4)
ba = mqslam_amd.bundle_adjustment.make_benchmark_problem(u, P, pts + 0.01, torch.device("cuda", 0), seed=1)
ba.gauss_newton_iteration(0.0)
torch.cuda.synchronize()
np.savez(sys.argv[2], poses=ba.poses.cpu().numpy(), points=ba.points.cpu().numpy(), lin=ba.lin.cpu().numpy(), dpose=ba.dpose.cpu().numpy())
