#!/usr/bin/env python3
"""
CLI-compatible counterpart of the reference's `bundle_adjust` tool
(Work/SLAM/tools/bundle_adjustment/bundle_adjust.cpp:454-533, file mode `runFromFiles` :425-448):

    bundle_adjust.py <baseDir> <baseName> <nrCameras> <fps> [useOdometry [fullOptimizeAtSecondPoints3DBatch
                     [startTime [firstFrameStartsAfterStartTime [iSAM_version [runFromGenerated]]]]]]

Reads the BA_info.* / traj_out.* / map_out-*.pcd file set, runs the full (batch Levenberg-Marquardt)
optimisation on the GPU -- the reference's iSAM_version = 0 mode, the one its ReadMe says works on slam2
data -- and writes traj_out.camC-<baseName>-BA.txt and map_out-<baseName>-BA.pcd.  useOdometry adds the
BetweenFactors of the odometry files (default 1, like the reference).  The damping is GTSAM's default additive form
(environment MQS_BA_DAMPING=marquardt: diagonal scaling).  iSAM1/iSAM2 and in-memory
generation are outside the accelerated path.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv):
    if len(argv) < 5 or argv[1] in ("-h", "--help"):
        print(__doc__)
        return 1
    base_dir, base_name, nr_cameras, fps = argv[1], argv[2], int(argv[3]), int(argv[4])
    opt = [float(a) for a in argv[5:]]
    use_odometry = bool(opt[0]) if len(opt) > 0 else True
    start_time = opt[2] if len(opt) > 2 else 0.0
    first_after = bool(opt[3]) if len(opt) > 3 else True
    isam = int(opt[4]) if len(opt) > 4 else 2
    if len(opt) > 5 and opt[5]:
        raise SystemExit("runFromGenerated is not supported (Boost.Random stream is version dependent)")
    import mqslam_amd
    io = mqslam_amd.ba_io
    if isam != 0:
        print("note: iSAM%d is not accelerated; running the full optimisation (iSAM_version = 0)" % isam)
    fn = io.create_filenames(base_dir, base_name, nr_cameras)
    data = io.load_data(fn, fps, start_time, first_after)
    io.validate_data_integrity(data, nr_cameras)
    ok, log = io.validate_sufficiently_constrained(data, use_odometry)
    if not ok:
        print("Warning: num_unknowns > num_constraints at some step")
    problem = io.build_sparse_problem(data, use_odometry=use_odometry)
    print("Running full optimization (Levenberg-Marquardt) on %d 3D points and %d camera(s) with each %d frames."
          % (len(problem.points), nr_cameras, len(data.point3DAddedIdxs)))
    ba = mqslam_amd.sparse_ba.SparseBundleAdjuster(problem)
    # Levenberg-Marquardt with GTSAM 3.2.1's default parameters, additive damping included (diagonalDamping = false): what
    # bundle_adjust.cpp:323-324 runs.  MQS_BA_DAMPING=marquardt selects the diagonal scaling instead (same optimum).
    hist = ba.optimize(mode="lm", verbose=False, damping=os.environ.get("MQS_BA_DAMPING", "gtsam"))
    print("cost %.6e -> %.6e in %d iterations" % (hist[0], hist[-1], len(hist) - 1))
    io.update_data_with_estimate(data, ba.problem, ba.poses.cpu().numpy(), ba.points.cpu().numpy())
    io.save_result(fn, data)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
