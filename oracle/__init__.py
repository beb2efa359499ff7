"""
oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements (numpy + plain C) of the reference's hot-path algorithms.  They exist to
*check* the HIP product path and to be timed as the CPU baseline.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import, link or run
anything in this directory.  The product package (``multiple-quadrotor-slam_amd/``) never does:
it fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md section "Oracle"):
  * triangulation (T1/T2/T3): PINNED against the reference's committed known-answer file
    ``Work/triangulation_comparison/test_3.mat`` (tests/golden/test_3_golden.npz,
    tests/test_oracle_golden.py).
  * matcher (M1): parity unpinned (reference holds no fixture; OpenCV 2.4 absent).
  * bundle adjustment (B1-B4): parity unpinned per iteration (GTSAM 3.2.1 absent); the
    converged reference outputs under ``bundle_adjustment/example`` are a loose anchor only.
"""
