#!/usr/bin/env python3
"""
Regenerates tests/golden/*.npz from the reference's committed known-answer DATA files
(only run in the build container, where /root/reference exists; the .npz travel).

  test_3_golden.npz     <- Work/triangulation_comparison/test_3.mat
  test_1and2_golden.npz <- Work/triangulation_comparison/test_1and2.mat  (k1=0.3 tier, summaries only)

Only numeric arrays (inputs: points_3D, noise_sigma_values, num_trials, rseed; expected
outputs: the *_summary statistics) are extracted; no reference source text is copied.
"""
import os
import numpy as np
import scipy.io as sio

REF = "/root/reference/Work/triangulation_comparison"
HERE = os.path.dirname(os.path.abspath(__file__))

def main():
    m = sio.loadmat(os.path.join(REF, "test_3.mat"))
    keys = ["err3D_mean_summary", "err3D_median_summary", "err2D_mean_summary", "err2D_median_summary",
            "false_pos_summary", "false_neg_summary", "points_3D", "noise_sigma_values"]
    out = {k: np.asarray(m[k], dtype=np.float64) for k in keys}
    out["noise_sigma_values"] = out["noise_sigma_values"].reshape(-1)
    out["num_trials"] = np.int64(m["num_trials"][0, 0])
    out["rseed"] = np.int64(m["rseed"][0, 0])
    np.savez_compressed(os.path.join(HERE, "test_3_golden.npz"), **out)

    m = sio.loadmat(os.path.join(REF, "test_1and2.mat"))
    keys = ["err3D_mean_summary", "err3D_median_summary", "err2D_mean_summary", "err2D_median_summary",
            "false_pos_summary", "false_neg_summary", "points_3D"]
    out = {k: np.asarray(m[k], dtype=np.float64) for k in keys}
    out["num_trials"] = np.int64(m["num_trials"][0, 0])
    out["rseed"] = np.int64(m["rseed"][0, 0])
    np.savez_compressed(os.path.join(HERE, "test_1and2_golden.npz"), **out)

if __name__ == "__main__":
    main()
