import os, sys, shutil, subprocess, tempfile
import numpy as np
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
import mqslam_amd
from util import ate_rmse
io=mqslam_amd.ba_io
SVO=ROOT+"/tests/golden/ba_svo"
work=tempfile.mkdtemp()+"/svo"; shutil.copytree(SVO, work)
for f in ("traj_out.cam0-slam2-BA.txt","map_out-slam2-BA.pcd"): os.remove(work+"/"+f)
out=subprocess.run([sys.executable, ROOT+"/tools/bundle_adjust.py", work, "slam2","1","50","0","1","0","1","0"],capture_output=True,text=True)
print(out.stdout[-600:], out.stderr[-300:])
ours=io.load_map(work+"/map_out-slam2-BA.pcd"); ref=io.load_map(SVO+"/map_out-slam2-BA.pcd"); inp=io.load_map(SVO+"/map_out-slam2.pcd")
d=np.linalg.norm(ours-ref,axis=1); di=np.linalg.norm(inp-ref,axis=1)
print("map: median d_ours %.5g p90 %.5g max %.5g | input median %.5g p90 %.5g" % (np.median(d), np.quantile(d,.9), d.max(), np.median(di), np.quantile(di,.9)))
tr=io.load_trajectory(work+"/traj_out.cam0-slam2-BA.txt"); rt=io.load_trajectory(SVO+"/traj_out.cam0-slam2-BA.txt"); it=io.load_trajectory(SVO+"/traj_out.cam0-slam2.txt")
e=np.array([np.linalg.norm(a[1][9:]-b[1][9:]) for a,b in zip(tr,rt)]); ei=np.array([np.linalg.norm(a[1][9:]-b[1][9:]) for a,b in zip(it,rt)])
r=np.array([np.abs(a[1][:9]-b[1][:9]).max() for a,b in zip(tr,rt)])
print("traj: median %.5g p90 %.5g max %.5g | input median %.5g ; rot max-abs median %.3g max %.3g" % (np.median(e), np.quantile(e,.9), e.max(), np.median(ei), np.median(r), r.max()))
gt=[(t,p[9:]) for t,p in io.load_trajectory(SVO+"/traj_groundtruth.txt")]
print("ATE ours %.6f ref %.6f" % (ate_rmse([(t,p[9:]) for t,p in tr],gt)[0], ate_rmse([(t,p[9:]) for t,p in rt],gt)[0]))
