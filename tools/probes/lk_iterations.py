import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, mqslam_amd
seq = mqslam_amd.synthetic.PlaneSequence(frames=60)
F = mqslam_amd.features
out = {}
for (i, j) in ((0, 1), (30, 31), (58, 59)):
    a, b = seq.render(i), seq.render(j)
    pts = F.goodFeaturesToTrack(a, 300, 0.01, 12).reshape(-1, 2).astype(np.float32)
    nxt, st, err = F.calcOpticalFlowPyrLK(a, b, pts)
    it = err.ravel()
    out["%d->%d" % (i, j)] = {"features": len(pts), "tracked": int(st.sum()), "iterations_mean": float(it.mean()), "median": float(np.median(it)), "max": float(it.max()), "flow_px_median": float(np.median(np.linalg.norm(nxt - pts, axis=1)))}
print(json.dumps(out))
