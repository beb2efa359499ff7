"""
On-disk bundle-adjustment problem format of the reference: readers,
writers, integrity / constraint-count validators, and the conversion to the flat sparse arrays
the solver consumes.

Follows (paths relative to the reference repository):
  file naming            Work/SLAM/tools/bundle_adjustment/IO.hpp:46-135   createFilenames
  ASCII list format      IO.hpp:141-185   loadAscii ('#' comment, empty line => next step / frame,
                                          fields separated by single spaces)
  record decoders        IO.hpp:188-296   (pose line "tx ty tz qx qy qz qw" = camera-to-world, used
                                          un-inverted; Cal3DS2 "fx fy s u0 v0 k1 k2 p1 p2"; PCD "x y z [rgb]";
                                          noise "Unit" | "Isotropic s" | "Diagonal s.." | "Constrained s..")
  fillHolesInTrajectories IO.hpp:302-363
  loadData / saveResult  IO.hpp:366-406, 412-475
  validateDataIntegrity  DataStructures.hpp:94-164
  validateDataSufficientlyConstrainted   bundle_adjust.cpp:42-177
  writer side            Work/SLAM/application/own/slam2.py:743-865 (BundleAdjustmentInfoContainer)
  graph construction     bundle_adjust.cpp:245-309 (-> SparseProblem below)
"""
import os
from collections import namedtuple

import numpy as np

Filenames = namedtuple("Filenames", [
    "map_in", "trajectories_in", "poseNoise", "odometryNoise", "point3DNoise", "point2DNoise", "calibrations",
    "odometry", "odometryAssocs", "point3DAddedIdxs", "points2D", "point2D3DAssocs", "map_out", "trajectories_out"])


def create_filenames(base_dir, base_name, nr_cameras):
    """IO.hpp:46-135."""
    j = lambda n: os.path.join(base_dir, n)
    cams = range(nr_cameras)
    return Filenames(
        map_in=j("map_out-%s.pcd" % base_name),
        trajectories_in=[j("traj_out.cam%d-%s.txt" % (c, base_name)) for c in cams],
        poseNoise=[j("BA_info.noise.pose.cam%d-%s.txt" % (c, base_name)) for c in cams],
        odometryNoise=j("BA_info.noise.odometry-%s.txt" % base_name),
        point3DNoise=j("BA_info.noise.point3D-%s.txt" % base_name),
        point2DNoise=[j("BA_info.noise.point2D.cam%d-%s.txt" % (c, base_name)) for c in cams],
        calibrations=[j("BA_info.calibrations.cam%d.txt" % c) for c in cams],
        odometry=j("BA_info.measurements.odometry-%s.txt" % base_name),
        odometryAssocs=j("BA_info.measurements.odometryAssocs-%s.txt" % base_name),
        point3DAddedIdxs=j("BA_info.measurements.point3DAddedIdxs-%s.txt" % base_name),
        points2D=[j("BA_info.measurements.points2D.cam%d-%s.txt" % (c, base_name)) for c in cams],
        point2D3DAssocs=[j("BA_info.measurements.point2D3DAssocs.cam%d-%s.txt" % (c, base_name)) for c in cams],
        map_out=j("map_out-%s-BA.pcd" % base_name),
        trajectories_out=[j("traj_out.cam%d-%s-BA.txt" % (c, base_name)) for c in cams])


def load_ascii(filename, decoder, empty_lines_trigger_new_list=True):
    """IO.hpp:141-185.  decoder(fields) returns a record, or None for a line that is not a record
    (PCD header lines, IO.hpp:249-262)."""
    out = [[]]
    with open(filename) as f:
        for line in f:
            line = line.rstrip("\n").rstrip("\r")
            if line and line[0] == "#":
                continue
            if empty_lines_trigger_new_list and line == "":
                out.append([])
                continue
            rec = decoder(line.split(" "))
            if rec is not None:
                out[-1].append(rec)
    return out


# ---- rotations ---------------------------------------------------------------------------

def quat_to_R(qx, qy, qz, qw):
    """Rot3::quaternion(w, x, y, z) (IO.hpp:224-225); the quaternion is normalised first."""
    n = np.sqrt(qx * qx + qy * qy + qz * qz + qw * qw)
    x, y, z, w = qx / n, qy / n, qz / n, qw / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_to_quat(R):
    """(qx, qy, qz, qw), qw >= 0 branch-stable (Rot3::toQuaternion, IO.hpp:432)."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = np.array([0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s])
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = np.array([(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s])
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = np.array([(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s])
    return q / np.linalg.norm(q)


def pose12_from_line(f):
    """"tx ty tz qx qy qz qw" -> [R row-major (9) | t (3)], camera-to-world (IO.hpp:221-227)."""
    t = np.array([float(f[0]), float(f[1]), float(f[2])])
    R = quat_to_R(float(f[3]), float(f[4]), float(f[5]), float(f[6]))
    return np.concatenate([R.reshape(-1), t])


# ---- decoders (IO.hpp:188-296) -----------------------------------------------------------

def _dec_size_t(f):
    assert len(f) == 1
    return int(f[0])


def _dec_assoc_odo(f):
    assert len(f) == 4
    return tuple(int(v) for v in f)              # from_cam, from_frame, to_cam, to_frame


def _dec_assoc_2d3d(f):
    assert len(f) == 3
    return tuple(int(v) for v in f)              # frame, point2D, point3D


def _dec_point2(f):
    assert len(f) == 2
    return (float(f[0]), float(f[1]))


def _dec_pose3(f):
    assert len(f) == 7
    return pose12_from_line(f)


def _dec_cal3ds2(f):
    assert len(f) == 9
    return np.array([float(v) for v in f])


def _dec_traj_node(f):
    assert len(f) == 8
    return (float(f[0]), pose12_from_line(f[1:]))


_PCD_HEADER = {"VERSION", "FIELDS", "SIZE", "TYPE", "COUNT", "WIDTH", "HEIGHT", "VIEWPOINT", "POINTS", "DATA"}


def _dec_map_point(f):
    if f[0] in _PCD_HEADER:
        if f[0] == "FIELDS":
            assert f[1:4] == ["x", "y", "z"] and (len(f) == 4 or f[4] == "rgb")
        return None
    assert len(f) >= 3
    col = float(f[3]) if len(f) > 3 else None
    return (float(f[0]), float(f[1]), float(f[2]), col)


def _noise_decoder(dim):
    def dec(f):
        kind, vals = f[0], [float(v) for v in f[1:]]
        if kind == "Unit":
            assert not vals
            return np.ones(dim)
        if kind == "Isotropic":
            assert len(vals) == 1
            return np.full(dim, vals[0])
        if kind in ("Diagonal", "Constrained"):
            assert len(vals) == dim
            return np.array(vals)
        raise ValueError("Noise-type '%s' unknown." % kind)
    return dec


# ---- in-memory problem (DataStructures.hpp:55-88) ------------------------------------------

class BAData:
    """Per-camera per-step ragged arrays exactly as the reference's BAdata."""

    def __init__(self):
        self.poseNoise = []          # [cam] sigmas (6)
        self.odometryNoise = []      # [from cam][to cam] sigmas (6)
        self.point3DNoise = None     # sigmas (3)
        self.point2DNoise = []       # [cam] sigmas (2)
        self.calibrations = []       # [cam] (9)
        self.poses = []              # [cam][frame] -> (t, pose12) or None
        self.odometry = []           # [step] list of pose12
        self.odometryAssocs = []     # [step] list of (from_cam, from_frame, to_cam, to_frame)
        self.points3D = None         # (N, 3)
        self.colors = None           # list of PCD colour floats (or None)
        self.point3DAddedIdxs = []   # [step] list of landmark idx
        self.points2D = []           # [cam][frame] list of (x, y)
        self.point2D3DAssocs = []    # [cam][step] list of (frame, point2D, point3D)


def fill_holes_in_trajectories(data, fps, start_time, first_frame_starts_after_start_time):
    """IO.hpp:302-363: map timestamps to frame slots at `fps`; missing => None."""
    nr_cameras = len(data.poses)
    nr_steps = len(data.point3DAddedIdxs)
    assert nr_cameras > 0
    end_time = start_time
    for c in range(nr_cameras):
        if data.poses[c] and data.poses[c][-1][0] > end_time:
            end_time = data.poses[c][-1][0]
    if fps > 0:
        nr_frames = int(round((end_time - start_time) * fps))
        if not first_frame_starts_after_start_time:
            nr_frames += 1
        new = []
        for c in range(nr_cameras):
            it = 0
            src = data.poses[c]
            row = []
            for f in range(nr_frames):
                t = start_time + (f + (1 if first_frame_starts_after_start_time else 0)) / float(fps)
                while it < len(src) and src[it][0] < t - 0.5 / fps:
                    it += 1
                if it < len(src) and (t - 0.5 / fps <= src[it][0] < t + 0.5 / fps):
                    row.append(src[it])
                else:
                    row.append(None)
            new.append(row)
        data.poses = new
    else:
        nr_frames = len(data.poses[0])
    assert nr_steps >= nr_frames
    for c in range(nr_cameras):
        data.poses[c] = list(data.poses[c]) + [None] * (nr_steps - nr_frames)


def load_data(filenames, fps=1, start_time=0.0, first_frame_starts_after_start_time=True):
    """IO.hpp:366-406."""
    d = BAData()
    for fn in filenames.poseNoise:
        d.poseNoise.append(load_ascii(fn, _noise_decoder(6))[0][0])
    d.odometryNoise = load_ascii(filenames.odometryNoise, _noise_decoder(6))
    d.point3DNoise = load_ascii(filenames.point3DNoise, _noise_decoder(3))[0][0]
    for fn in filenames.point2DNoise:
        d.point2DNoise.append(load_ascii(fn, _noise_decoder(2))[0][0])
    for fn in filenames.calibrations:
        d.calibrations.append(load_ascii(fn, _dec_cal3ds2)[0][0])
    d.odometry = load_ascii(filenames.odometry, _dec_pose3)
    d.odometryAssocs = load_ascii(filenames.odometryAssocs, _dec_assoc_odo)
    pts = load_ascii(filenames.map_in, _dec_map_point)[0]
    d.points3D = np.array([p[:3] for p in pts], dtype=np.float64).reshape(-1, 3)
    d.colors = [p[3] for p in pts]
    d.point3DAddedIdxs = load_ascii(filenames.point3DAddedIdxs, _dec_size_t)
    for fn in filenames.points2D:
        d.points2D.append(load_ascii(fn, _dec_point2))
    for fn in filenames.point2D3DAssocs:
        d.point2D3DAssocs.append(load_ascii(fn, _dec_assoc_2d3d))
    for fn in filenames.trajectories_in:
        d.poses.append(load_ascii(fn, _dec_traj_node)[0])
    fill_holes_in_trajectories(d, fps, start_time, first_frame_starts_after_start_time)
    return d


def validate_data_integrity(data, nr_cameras):
    """DataStructures.hpp:94-164 (asserts -> ValueError)."""
    def need(cond, msg):
        if not cond:
            raise ValueError("BA data integrity: " + msg)
    need(len(data.poses) == nr_cameras and len(data.calibrations) == nr_cameras, "camera list lengths")
    need(len(data.points2D) == nr_cameras and len(data.point2D3DAssocs) == nr_cameras, "camera list lengths")
    need(len(data.poseNoise) == nr_cameras and len(data.point2DNoise) == nr_cameras, "noise list lengths")
    need(len(data.odometryNoise) == nr_cameras and all(len(r) == nr_cameras for r in data.odometryNoise),
         "odometry noise must be a cam x cam matrix")
    need(nr_cameras > 0, "at least one camera")
    nr_frames = len(data.poses[0])
    need(all(len(p) == nr_frames for p in data.poses), "trajectories must have equal length")
    nr_steps = len(data.point3DAddedIdxs)
    need(nr_steps == nr_frames, "number of steps must equal number of frames")
    need(len(data.odometry) == nr_steps and len(data.odometryAssocs) == nr_steps, "odometry steps")
    for s in range(nr_steps):
        need(len(data.odometry[s]) == len(data.odometryAssocs[s]), "odometry / assocs per step")
    n = len(data.points3D)
    for s in range(nr_steps):
        for p in data.point3DAddedIdxs[s]:
            need(0 <= p < n, "point3DAddedIdxs out of range")
    for c in range(nr_cameras):
        need(len(data.point2D3DAssocs[c]) == nr_steps, "point2D3DAssocs steps")
        for s in range(nr_steps):
            for (f, p2, p3) in data.point2D3DAssocs[c][s]:
                need(0 <= f <= s, "association looks into the future")
                need(0 <= p2 < len(data.points2D[c][f]), "point2D index out of range")
                need(0 <= p3 < n, "point3D index out of range")
                need(data.poses[c][f] is not None, "association refers to a missing pose")
    for s in range(nr_steps):
        for (fc, ff, tc, tf) in data.odometryAssocs[s]:
            need(0 <= fc < nr_cameras and 0 <= tc < nr_cameras, "odometry camera out of range")
            need(0 <= ff <= s and 0 <= tf <= s, "odometry looks into the future")
            need(not (fc == tc and ff == tf), "odometry between the same image")
            need(data.poses[fc][ff] is not None and data.poses[tc][tf] is not None, "odometry refers to a missing pose")


def validate_sufficiently_constrained(data, use_odometry):
    """bundle_adjust.cpp:42-177: running count of unknowns (3/point, 6/pose) vs constraints (2/observation,
    6/pose prior, 3/point prior, 6/odometry).  Returns (valid, [(step, unknowns, constraints), ...])."""
    nr_cameras = len(data.calibrations)
    nr_steps = len(data.point3DAddedIdxs)
    n = len(data.points3D)
    est = np.zeros(n, dtype=bool)
    cnt = np.zeros(n, dtype=np.int64)
    pose_cnt = [[0] * nr_steps for _ in range(nr_cameras)]
    unknowns = constraints = 0
    valid = True
    log = []
    for s in range(nr_steps):
        for p in data.point3DAddedIdxs[s]:
            if est[p]:
                raise ValueError("3D point %d added twice" % p)
            est[p] = True
            unknowns += 3
        for c in range(nr_cameras):
            if data.poses[c][s] is not None:
                unknowns += 6
        if s == 0:
            for c in range(nr_cameras):
                if data.poses[c][0] is None:
                    raise ValueError("camera %d has no pose at the first frame" % c)
                pose_cnt[c][0] += 1
                constraints += 6
                for (f, p2, p3) in data.point2D3DAssocs[c][0]:
                    if f != 0 or not est[p3]:
                        raise ValueError("step-0 association must lie in frame 0 on an added point")
                    cnt[p3] += 1
                    constraints += 3
        for c in range(nr_cameras):
            for (f, p2, p3) in data.point2D3DAssocs[c][s]:
                if data.poses[c][f] is None or not est[p3]:
                    raise ValueError("association on a missing pose / point")
                pose_cnt[c][f] += 1
                cnt[p3] += 1
                constraints += 2
        if use_odometry:
            for (fc, ff, tc, tf) in data.odometryAssocs[s]:
                pose_cnt[fc][ff] += 1
                pose_cnt[tc][tf] += 1
                constraints += 6
        for p in data.point3DAddedIdxs[s]:
            if cnt[p] < 2:
                raise ValueError("3D point %d has fewer than 2 factors" % p)
        for c in range(nr_cameras):
            if data.poses[c][s] is not None and pose_cnt[c][s] < 1:
                raise ValueError("pose (%d, %d) has no factor" % (c, s))
        if unknowns > constraints:
            valid = False
        log.append((s, unknowns, constraints))
    return valid, log


# ---- flat sparse problem (graph of bundle_adjust.cpp:245-309) ------------------------------

SparseProblem = namedtuple("SparseProblem", [
    "poses", "pose_cam", "pose_key", "calib", "sigma", "points", "obs_ptr", "obs_pose", "obs_uv",
    "prior_w", "prior_xyz", "pose_prior_idx", "pose_prior_sigmas", "odo_from", "odo_to", "odo_meas", "odo_sigmas"])


def build_sparse_problem(data, use_odometry=False):
    """
    poses (P,12), pose_cam (P,), pose_key [(cam, frame)], calib (C,9), sigma (C,) isotropic pixel sigma,
    points (N,3); observations in CSR by landmark: obs_ptr (N+1,), obs_pose (M,), obs_uv (M,2);
    point priors on step-0 landmarks -- one PriorFactor<Point3> per step-0 association, so a point seen by
    two cameras at step 0 carries twice the weight (bundle_adjust.cpp:277-281); pose priors on the frame-0
    poses (:268-275); odometry BetweenFactors (:301-309) when requested.
    """
    nr_cameras = len(data.calibrations)
    nr_steps = len(data.point3DAddedIdxs)
    pose_index = {}
    poses, pose_cam, pose_key = [], [], []
    for c in range(nr_cameras):
        for f in range(nr_steps):
            if data.poses[c][f] is not None:
                pose_index[(c, f)] = len(poses)
                poses.append(data.poses[c][f][1])
                pose_cam.append(c)
                pose_key.append((c, f))
    n = len(data.points3D)
    per_lm = [[] for _ in range(n)]
    for c in range(nr_cameras):
        for s in range(nr_steps):
            for (f, p2, p3) in data.point2D3DAssocs[c][s]:
                per_lm[p3].append((pose_index[(c, f)], data.points2D[c][f][p2]))
    obs_ptr = np.zeros(n + 1, dtype=np.int64)
    for i in range(n):
        obs_ptr[i + 1] = obs_ptr[i] + len(per_lm[i])
    obs_pose = np.array([o[0] for l in per_lm for o in l], dtype=np.int32)
    obs_uv = np.array([o[1] for l in per_lm for o in l], dtype=np.float64).reshape(-1, 2)
    sig3 = data.point3DNoise
    if not np.all(sig3 == sig3[0]):
        raise NotImplementedError("only isotropic point3D noise is supported")
    prior_w = np.zeros(n)
    for c in range(nr_cameras):
        for (f, p2, p3) in data.point2D3DAssocs[c][0]:
            prior_w[p3] += 1.0 / sig3[0] ** 2
    sigma = []
    for c in range(nr_cameras):
        s2 = data.point2DNoise[c]
        if s2[0] != s2[1]:
            raise NotImplementedError("only isotropic point2D noise is supported")
        sigma.append(s2[0])
    ppi = [pose_index[(c, 0)] for c in range(nr_cameras) if (c, 0) in pose_index]
    pps = [data.poseNoise[c] for c in range(nr_cameras) if (c, 0) in pose_index]
    odo_from, odo_to, odo_meas, odo_sig = [], [], [], []
    if use_odometry:
        for s in range(nr_steps):
            for k, (fc, ff, tc, tf) in enumerate(data.odometryAssocs[s]):
                odo_from.append(pose_index[(fc, ff)])
                odo_to.append(pose_index[(tc, tf)])
                odo_meas.append(data.odometry[s][k])
                odo_sig.append(data.odometryNoise[fc][tc])
    return SparseProblem(
        poses=np.array(poses).reshape(-1, 12), pose_cam=np.array(pose_cam, dtype=np.int32), pose_key=pose_key,
        calib=np.array(data.calibrations).reshape(-1, 9), sigma=np.array(sigma), points=data.points3D.copy(),
        obs_ptr=obs_ptr, obs_pose=obs_pose, obs_uv=obs_uv, prior_w=prior_w, prior_xyz=data.points3D.copy(),
        pose_prior_idx=np.array(ppi, dtype=np.int32), pose_prior_sigmas=np.array(pps).reshape(-1, 6),
        odo_from=np.array(odo_from, dtype=np.int32), odo_to=np.array(odo_to, dtype=np.int32),
        odo_meas=np.array(odo_meas).reshape(-1, 12), odo_sigmas=np.array(odo_sig).reshape(-1, 6))


# ---- writers (IO.hpp:412-475) ---------------------------------------------------------------

def save_trajectory(filename, poses):
    """poses: list of (t, pose12) or None."""
    with open(filename, "w") as f:
        f.write("# Format: timestamp tx ty tz qx qy qz qw\n"
                "# Where translations and quaternions are defined in world coordinates (=> inverse of pose)\n")
        for node in poses:
            if node is None:
                continue
            t, p = node
            q = R_to_quat(p[:9].reshape(3, 3))
            f.write("%.16g %.16g %.16g %.16g %.16g %.16g %.16g %.16g\n" % (t, p[9], p[10], p[11], q[0], q[1], q[2], q[3]))


def save_map(filename, points3D, colors=None):
    n = len(points3D)
    with open(filename, "w") as f:
        f.write("# .PCD v.7 - Point Cloud Data file format\nVERSION .7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\n"
                "COUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA ascii\n" % (n, n))
        white = np.frombuffer(np.uint32(0xFF | (0xFF << 8) | (0xFF << 16) | (0xFD << 24)).tobytes(), dtype=np.float32)[0]
        for i in range(n):
            c = white if colors is None or colors[i] is None else colors[i]
            f.write("%.16g %.16g %.16g %.9g\n" % (points3D[i, 0], points3D[i, 1], points3D[i, 2], c))


def load_trajectory(filename):
    return load_ascii(filename, _dec_traj_node)[0]


def load_map(filename):
    pts = load_ascii(filename, _dec_map_point)[0]
    return np.array([p[:3] for p in pts], dtype=np.float64).reshape(-1, 3)


def update_data_with_estimate(data, problem, poses, points):
    """bundle_adjust.cpp:378-395."""
    data.points3D = np.asarray(points).copy()
    for k, (c, f) in enumerate(problem.pose_key):
        data.poses[c][f] = (data.poses[c][f][0], np.asarray(poses[k]).copy())


def save_result(filenames, data):
    save_map(filenames.map_out, data.points3D, data.colors)
    for c in range(len(data.poses)):
        save_trajectory(filenames.trajectories_out[c], data.poses[c])


# ---- recorder + writer of the BA problem on the SLAM side (slam2.py:743-865) --------------------------------

class BundleAdjustmentInfoContainer:
    """
    What the per-frame loop records for the bundle adjuster, step by step, and the writer of the `BA_info.*` file set --
    same member-function names, file names and line formats (the on-disk contract) as the reference's class of this name
    (Work/SLAM/application/own/slam2.py:743-865), so that `load_data` / tools/bundle_adjust.py (and the reference's own
    `bundle_adjust` binary) read what `slam_loop.MonoSlam` writes.  One "step" per frame (`next_step`).
    """

    def __init__(self, base_dir, base_name, num_cams=1):
        self.base_dir, self.base_name, self.num_cams = base_dir, base_name, num_cams
        self.calibrations = [None] * num_cams
        # Flat event logs, each row tagged with the step (or frame) it belongs to; the nested per-step lists of the file
        # formats are formed when writing (`_grouped`).  Rows:
        self._odometry = []                                   # (step, from_cam, from_frame, to_cam, to_frame, P 4x4)
        self._features = [[] for _ in range(num_cams)]        # per camera: (frame, x, y) in arrival order
        self._feature_count = [dict() for _ in range(num_cams)]    # per camera: frame -> features recorded so far
        self._assocs = [[] for _ in range(num_cams)]          # per camera: (step, frame, point2DIdx, point3DIdx)
        self._added = {}                                      # step -> landmark indices created in that step
        self.step = 0

    def next_step(self):                                      # slam2.py:761-768: one step per frame
        self.step += 1

    def set_calibration(self, K, distCoeffs, cam=0):          # :770-771
        self.calibrations[cam] = (np.asarray(K, dtype=np.float64), np.asarray(distCoeffs, dtype=np.float64).reshape(-1))

    def add_odometry(self, odometry, from_frame, to_frame, from_cam=0, to_cam=0):      # :773-775
        """odometry: 4x4 (or 3x4) transform P with P_to = P * P_from (world->camera matrices; trfm.delta_P)."""
        self._odometry.append((self.step, int(from_cam), int(from_frame), int(to_cam), int(to_frame),
                               np.asarray(odometry, dtype=np.float64)))

    def add_points2D_3Dassoc(self, points2D, point3DIdxs, frame, cam=0):               # :777-786
        """Appends features to `frame`'s list (their indices continue that frame's numbering) and ties each to its landmark."""
        points2D = np.asarray(points2D, dtype=np.float64).reshape(-1, 2)
        frame = int(frame)
        first = self._feature_count[cam].get(frame, 0)
        self._feature_count[cam][frame] = first + len(points2D)
        for k, ((x, y), lm) in enumerate(zip(points2D, np.asarray(point3DIdxs).reshape(-1))):
            self._features[cam].append((frame, float(x), float(y)))
            self._assocs[cam].append((self.step, frame, first + k, int(lm)))

    def set_point3DAddedIdxs(self, point3DAddedIdxs):         # :788-790
        self._added[self.step] = [int(i) for i in point3DAddedIdxs]

    def _grouped(self, rows, key, count):
        """rows -> `count` lists, row r in list key(r), arrival order kept inside a list."""
        out = [[] for _ in range(count)]
        for r in rows:
            out[key(r)].append(r)
        return out

    # ---- writers (:792-865) ----
    def _write(self, title, lines, cam=-1, omit_base_name=False):
        name = "BA_info.%s%s%s.txt" % (title, (".cam%s" % cam) if cam > -1 else "", "" if omit_base_name else "-%s" % self.base_name)
        with open(os.path.join(self.base_dir, name), "w") as f:
            f.write("\n".join(lines + [""]))

    def write_calibrations(self, cam):
        K, d = self.calibrations[cam]
        if len(d) > 4 and d[4] != 0.0:
            raise AttributeError("the optimiser's camera model (Cal3DS2) has no 6th-order radial coefficient")
        d4 = tuple(d[:4]) + (0.0,) * (4 - min(len(d), 4))
        self._write("calibrations", ["# Format: fx fy shear u0 v0 k1 k2 p1 p2",
                                     "%.16e %.16e %.16e %.16e %.16e %.16e %.16e %.16e %.16e"
                                     % ((K[0, 0], K[1, 1], K[0, 1], K[0, 2], K[1, 2]) + d4)], cam, omit_base_name=True)

    @staticmethod
    def _steps(lines, per_step, fmt):
        for step, items in enumerate(per_step):
            if step:
                lines.append("")
            lines.extend(fmt(x) for x in items)
        return lines

    def write_odometry(self):
        def fmt(P):                                                # trfm.pose_TUM_from_P: the INVERSE transform as location + quaternion
            R, t = P[:3, :3], P[:3, 3]
            q = R_to_quat(R.T)
            return "%.16e %.16e %.16e %.16e %.16e %.16e %.16e" % (tuple(-R.T @ t) + tuple(q))
        per_step = self._grouped(self._odometry, lambda r: r[0], self.step + 1)
        self._write("measurements.odometry", self._steps(
            ["# Format: tx ty tz qx qy qz qw", "# Newline means next odometry; Empty line means next step"], per_step,
            lambda r: fmt(r[5])))

    def write_odometryAssocs(self):
        self._write("measurements.odometryAssocs", self._steps(
            ["# Format: from_cam from_frame to_cam to_frame", "# Newline means next odometry; Empty line means next step"],
            self._grouped(self._odometry, lambda r: r[0], self.step + 1), lambda r: "%d %d %d %d" % r[1:5]))

    def write_points2D(self, cam):
        self._write("measurements.points2D", self._steps(
            ["# Format: x y", "# Newline means next feature; Empty line means next frame, first feature"],
            self._grouped(self._features[cam], lambda r: r[0], self.step + 1), lambda r: "%.16e %.16e" % r[1:3]), cam)

    def write_point2D3DAssocs(self, cam):
        self._write("measurements.point2D3DAssocs", self._steps(
            ["# Format: frameIdx point2DIdx point3DIdx", "# Newline means next feature; Empty line means next step, first feature"],
            self._grouped(self._assocs[cam], lambda r: r[0], self.step + 1), lambda r: "%d %d %d" % r[1:4]), cam)

    def write_point3DAddedIdxs(self):
        self._write("measurements.point3DAddedIdxs", self._steps(
            ["# Format: point3DIdx", "# Newline means next point; Empty line means next step"],
            [self._added.get(k, []) for k in range(self.step + 1)], str))

    def write_noise(self, pose=(0.002, 0.002, 0.002, 0.001, 0.001, 0.001), odometry=(0.05, 0.05, 0.05, 0.2, 0.2, 0.2),
                    point3D=0.25, point2D=5.0):
        """The four hand-written noise files of a data set (not produced by slam2.py; defaults = the values committed with
        the reference's SVO run, datasets/SVO/sin2_tex2_h1_v8_d/BA_info.noise.*-slam2.txt)."""
        head = lambda dim: ["# Format: noiseType noiseSpecificValues", "# The dimension of the noise is equal to %d." % dim]
        diag = lambda v: "Diagonal " + " ".join("%.16g" % x for x in v)
        for cam in range(self.num_cams):
            self._write("noise.pose", head(6) + [diag(pose)], cam)
            self._write("noise.point2D", head(2) + ["Isotropic %.16g" % point2D], cam)
        rows = []
        for a in range(self.num_cams):
            if a:
                rows.append("")
            rows.extend(diag(odometry) for _ in range(self.num_cams))
        self._write("noise.odometry", head(6) + rows)
        self._write("noise.point3D", head(3) + ["Isotropic %.16g" % point3D])

    def write_all(self):                                           # :856-865
        for cam in range(self.num_cams):
            self.write_calibrations(cam)
            self.write_points2D(cam)
            self.write_point2D3DAssocs(cam)
        self.write_odometry()
        self.write_odometryAssocs()
        self.write_point3DAddedIdxs()


def save_slam_output(filenames, fps, Ps, points3D, cam=0):
    """slam2.py:698-741 `write_output`: the trajectory in TUM format (world->camera matrices Ps, None for rejected frames;
    timestamp of frame i = (1 + i) / fps, dataset_tools.py:275-294) and the map as PCD."""
    nodes = []
    for i, P in enumerate(Ps):
        if P is None:
            continue
        R, t = np.asarray(P)[:3, :3], np.asarray(P)[:3, 3]
        nodes.append(((1.0 + i) / fps, np.concatenate([R.T.reshape(-1), -R.T @ t])))
    save_trajectory(filenames.trajectories_in[cam], nodes)
    save_map(filenames.map_in, np.asarray(points3D, dtype=np.float64))
