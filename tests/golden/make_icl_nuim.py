#!/usr/bin/env python3
"""Fixture generator (runs in the build container, where /root/reference exists): the first frames of the reference's own SLAM example
run -- `slam2.py`'s ExampleUsage for the ICL-NUIM living-room sequence (slam2.py:924-933) -- with what the reference committed as
that run's OUTPUT.

  inputs   rgb/0.png .. rgb/<n-1>.png converted to 8-bit grey with OpenCV's BGR2GRAY fixed-point formula (slam2.py:1113:
           cv2.cvtColor(img, COLOR_BGR2GRAY) = (R*4899 + G*9617 + B*1868 + 8192) >> 14), camera_intrinsics.txt (fy NEGATIVE),
           init_pose.txt, init_points.pcd
  expected traj_out.cam0-slam2.txt (the trajectory slam2.py wrote for these images, OpenCV 2.4's goodFeaturesToTrack /
           calcOpticalFlowPyrLK / solvePnPRansac inside), traj_groundtruth3.txt (the renderer's exact trajectory),
           results_ate-slam2.txt (the reference's own error figures of that run)

Usage: python tests/golden/make_icl_nuim.py [n_frames=80]      -> icl_nuim_traj3n/sequence.npz          (frames 0 .. n-1 and all the rows)
       python tests/golden/make_icl_nuim.py rest [first=80]  -> icl_nuim_traj3n/sequence_rest.npz     (frames first .. 199: the part of the
                                                                run in which the reference's own trajectory drifts to 0.17 m)
The images are the ICL-NUIM living-room data set's (Handa, Whelan, McDonald, Davison: "A Benchmark for RGB-D Visual Odometry, 3D
Reconstruction and SLAM", ICRA 2014; http://www.doc.ic.ac.uk/~ahanda/VaFRIC/iclnuim.html; CC BY 3.0), as the reference repository
redistributes them (datasets/ICL_NUIM/living_room_traj3n_frei_png/rgb: "a subset (200 images)"); see icl_nuim_traj3n/NOTICE.
"""
import os, sys, numpy as np
from PIL import Image

SRC = "/root/reference/Work/SLAM/datasets/ICL_NUIM"
SEQ = os.path.join(SRC, "living_room_traj3n_frei_png")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "icl_nuim_traj3n")


def grey(path):
    rgb = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint32)
    return ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def load_tum(path, n):
    rows = [l.split() for l in open(path) if l.strip() and not l.startswith("#")]
    return np.array(rows[:n], dtype=np.float64)


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "rest":
        first = int(sys.argv[2]) if len(sys.argv) > 2 else 80
        frames = np.stack([grey(os.path.join(SEQ, "rgb", "%d.png" % k)) for k in range(first, 200)])
        np.savez_compressed(os.path.join(OUT, "sequence_rest.npz"), frames=frames, first=np.array(first))
        print("frames", frames.shape, "->", os.path.getsize(os.path.join(OUT, "sequence_rest.npz")) / 1e6, "MB")
        return
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    frames = np.stack([grey(os.path.join(SEQ, "rgb", "%d.png" % k)) for k in range(n)])
    pts = np.array([l.split() for l in open(os.path.join(SEQ, "init_points.pcd")).read().split("DATA ascii\n")[1].strip().split("\n")], dtype=np.float64)
    np.savez_compressed(
        os.path.join(OUT, "sequence.npz"), frames=frames,
        K=np.array([[481.20, 0.0, 319.50], [0.0, -480.00, 239.50], [0.0, 0.0, 1.0]]),       # ICL_NUIM/camera_intrinsics.txt
        dist=np.zeros(5), init_pose=np.loadtxt(os.path.join(SEQ, "init_pose.txt")), init_points=pts,
        traj_slam2=load_tum(os.path.join(SEQ, "traj_out.cam0-slam2.txt"), n),
        traj_groundtruth=load_tum(os.path.join(SEQ, "traj_groundtruth3.txt"), n),
        traj_slam2_all=load_tum(os.path.join(SEQ, "traj_out.cam0-slam2.txt"), 10 ** 9),
        traj_groundtruth_all=load_tum(os.path.join(SEQ, "traj_groundtruth3.txt"), 10 ** 9))
    print("frames", frames.shape, "->", os.path.getsize(os.path.join(OUT, "sequence.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
