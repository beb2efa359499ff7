#!/usr/bin/env python3
"""
The reference's experiment `Work/triangulation_comparison/triangulation_comparison.py main()` (:643-657) on the GPU path:
runs Test 1and2 and Test 3 over the five trajectories and writes the summary arrays (same names and index order as the
reference's test_1and2.mat / test_3.mat) as .npz, and as .mat when scipy is importable.

    python tools/triangulation_comparison.py [out_dir]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(out_dir="."):
    import mqslam_amd
    tc = mqslam_amd.triangulation_comparison
    t0 = time.perf_counter()
    print("Running tests 1 and 2 ...")
    a = tc.test_1and2(filename=os.path.join(out_dir, "test_1and2.npz"))
    print("Running test 3 ...")
    b = tc.test_3(filename=os.path.join(out_dir, "test_3.npz"))
    print("%.1f s" % (time.perf_counter() - t0))
    try:
        import scipy.io as sio
        sio.savemat(os.path.join(out_dir, "test_1and2.mat"), {k: v for k, v in a.items()})
        sio.savemat(os.path.join(out_dir, "test_3.mat"), {k: v for k, v in b.items()})
    except ImportError:
        pass
    if not (a["is_inside_view"] and b["is_inside_view"]):
        print("Warning: some points fell out of view.")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
