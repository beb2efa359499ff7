"""
Brute-force matcher facade over the gfx950 kernels (csrc/match.hip) -- counterpart of the
reference's `cv2_helpers.BFMatcher` (Work/python_libs/cv2_helpers.py:278-345): default
construction = NORM_L2 without cross-check, `radiusMatch(query, train, maxDistance)` returns per
query the (at most two) nearest train points within the radius, ascending, as DMatch-like tuples;
`knnMatch(query, train, k=2)` the two nearest.  Ties go to the lower train index.

`knn2(query, train)` is the array form used by the benchmark: idx (Nq,2) int32, dist (Nq,2) f32.
float32 inputs take the exact path; float16 inputs (binary descriptors expanded to {0,1}) take the
MFMA path.  No CPU fallback: a missing library raises RuntimeError.
"""
import ctypes
from collections import namedtuple

import numpy as np

from . import _lib
from ._lib import c_f32p, c_i32p, c_u16p, c_i64

DMatch = namedtuple("DMatch", ["queryIdx", "trainIdx", "distance"])

NORM_L2 = 4          # cv2.NORM_L2


def knn2(query, train):
    query = np.asarray(query)
    train = np.asarray(train)
    if query.ndim != 2 or train.ndim != 2 or query.shape[1] != train.shape[1]:
        raise ValueError("query (Nq, D) and train (Nt, D) must have the same D")
    if query.dtype != train.dtype:
        raise TypeError("query and train must have the same dtype")
    Nq, D = query.shape
    Nt = train.shape[0]
    idx = np.empty((Nq, 2), dtype=np.int32)
    dist = np.empty((Nq, 2), dtype=np.float32)
    if query.dtype not in (np.float32, np.float16):
        raise TypeError("descriptors must be float32 (exact path) or float16 {0,1} (MFMA path), got %s" % query.dtype)
    ctx = _lib.default_context().handle
    if query.dtype == np.float32:
        q = np.ascontiguousarray(query)
        t = np.ascontiguousarray(train)
        _lib.check(_lib.lib().mqs_match_knn2_f32(ctx, q.ctypes.data_as(c_f32p), c_i64(Nq), t.ctypes.data_as(c_f32p),
                                                 c_i64(Nt), int(D), idx.ctypes.data_as(c_i32p),
                                                 dist.ctypes.data_as(c_f32p)))
    elif query.dtype == np.float16:
        q = np.ascontiguousarray(query).view(np.uint16)
        t = np.ascontiguousarray(train).view(np.uint16)
        _lib.check(_lib.lib().mqs_match_knn2_f16(ctx, q.ctypes.data_as(c_u16p), c_i64(Nq), t.ctypes.data_as(c_u16p),
                                                 c_i64(Nt), int(D), idx.ctypes.data_as(c_i32p),
                                                 dist.ctypes.data_as(c_f32p)))
    else:
        raise TypeError("descriptors must be float32 (exact path) or float16 {0,1} (MFMA path), got %s" % query.dtype)
    return idx, dist


def knn2_dev(query, train, out_idx=None, out_dist=None, workspace=None):
    """Device-resident form: torch tensors (float32 or float16) in, torch tensors out."""
    import torch
    if not (query.is_cuda and train.is_cuda and query.is_contiguous() and train.is_contiguous()):
        raise ValueError("query/train must be contiguous device tensors")
    if query.dtype != train.dtype or query.dim() != 2 or train.dim() != 2 or query.shape[1] != train.shape[1]:
        raise ValueError("query (Nq, D) and train (Nt, D) must match in dtype and D")
    Nq, D = int(query.shape[0]), int(query.shape[1])
    Nt = int(train.shape[0])
    idx = out_idx if out_idx is not None else torch.empty((Nq, 2), dtype=torch.int32, device=query.device)
    dist = out_dist if out_dist is not None else torch.empty((Nq, 2), dtype=torch.float32, device=query.device)
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if query.dtype == torch.float32:
        _lib.check(_lib.lib().mqs_match_knn2_f32_dev(query.data_ptr(), Nq, train.data_ptr(), Nt, D, idx.data_ptr(),
                                                     dist.data_ptr(), sp))
    elif query.dtype == torch.float16:
        need = int(_lib.lib().mqs_match_knn2_f16_workspace_bytes(Nq, Nt))
        ws = workspace if workspace is not None else torch.empty(max(need, 16), dtype=torch.uint8, device=query.device)
        _lib.check(_lib.lib().mqs_match_knn2_f16_dev(query.data_ptr(), Nq, train.data_ptr(), Nt, D, idx.data_ptr(),
                                                     dist.data_ptr(), ws.data_ptr(), ws.numel(), sp))
    else:
        raise TypeError("descriptors must be float32 or float16")
    return idx, dist


def pack_bits(descriptors01):
    """(n, D) array of 0/1 -> (n, D / 8) uint8, bit k of a descriptor = bit (k & 7) of byte k >> 3 (the layout of
    OpenCV's binary descriptors read as little-endian bit strings)."""
    d = np.asarray(descriptors01)
    if d.ndim != 2 or d.shape[1] % 8:
        raise ValueError("descriptors must be (n, D) with D a multiple of 8")
    return np.packbits(d.astype(np.uint8), axis=1, bitorder="little")


def knn2_bits(query_bits, train_bits):
    """Two nearest train descriptors under the Hamming distance for packed binary descriptors ((n, D / 8) uint8,
    D in {128, 256, 512}); returns idx (Nq, 2) int32 and dist (Nq, 2) float32 = sqrt(Hamming distance), exactly what
    `knn2` returns for the same descriptors expanded to {0,1} float16."""
    q = np.ascontiguousarray(query_bits)
    t = np.ascontiguousarray(train_bits)
    if q.dtype != np.uint8 or t.dtype != np.uint8 or q.ndim != 2 or t.ndim != 2 or q.shape[1] != t.shape[1]:
        raise ValueError("query_bits (Nq, D / 8) and train_bits (Nt, D / 8) must be uint8 with the same width")
    D = 8 * q.shape[1]
    if D not in (128, 256, 512):
        raise ValueError("D must be 128, 256 or 512 bits")
    Nq, Nt = len(q), len(t)
    idx = np.empty((Nq, 2), dtype=np.int32)
    dist = np.empty((Nq, 2), dtype=np.float32)
    _lib.check(_lib.lib().mqs_match_knn2_bits(_lib.default_context().handle, q.ctypes.data_as(_lib.c_u8p), c_i64(Nq),
                                              t.ctypes.data_as(_lib.c_u8p), c_i64(Nt), D, idx.ctypes.data_as(c_i32p),
                                              dist.ctypes.data_as(c_f32p)))
    return idx, dist


def knn2_bits_dev(query_bits, train_bits, out_idx=None, out_dist=None, workspace=None):
    """Device-resident form: uint8 torch tensors (n, D / 8)."""
    import torch
    if not (query_bits.is_cuda and train_bits.is_cuda and query_bits.is_contiguous() and train_bits.is_contiguous()):
        raise ValueError("query_bits / train_bits must be contiguous device tensors")
    if query_bits.dtype != torch.uint8 or train_bits.dtype != torch.uint8 or query_bits.shape[1] != train_bits.shape[1]:
        raise ValueError("packed descriptors must be uint8 (n, D / 8) with the same width")
    Nq, Nt, D = int(query_bits.shape[0]), int(train_bits.shape[0]), 8 * int(query_bits.shape[1])
    idx = out_idx if out_idx is not None else torch.empty((Nq, 2), dtype=torch.int32, device=query_bits.device)
    dist = out_dist if out_dist is not None else torch.empty((Nq, 2), dtype=torch.float32, device=query_bits.device)
    need = int(_lib.lib().mqs_match_knn2_bits_workspace_bytes(Nq, Nt, D))
    ws = workspace if workspace is not None else torch.empty(max(need, 16), dtype=torch.uint8, device=query_bits.device)
    _lib.check(_lib.lib().mqs_match_knn2_bits_dev(query_bits.data_ptr(), Nq, train_bits.data_ptr(), Nt, D, idx.data_ptr(),
                                                  dist.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return idx, dist


class BFMatcher:
    """cv2.BFMatcher()-shaped object (NORM_L2, no cross-check), cv2_helpers.py:282-345."""

    def __init__(self, normType=NORM_L2, crossCheck=False):
        if normType != NORM_L2 or crossCheck:
            raise NotImplementedError("only the reference's configuration (NORM_L2, crossCheck=False) is accelerated")
        self.normType = normType

    def getInt(self, name):
        if name == "normType":
            return self.normType
        raise KeyError(name)

    def match(self, query_points, train_points):
        """cv2.BFMatcher.match (forwarded by the reference's wrapper, cv2_helpers.py:341-345): the nearest train point."""
        return [ms[0] for ms in self.knnMatch(query_points, train_points, k=1) if ms]

    def knnMatch(self, query_points, train_points, k=2):
        if k not in (1, 2):
            raise NotImplementedError("k must be 1 or 2")
        idx, dist = knn2(np.asarray(query_points, dtype=np.float32), np.asarray(train_points, dtype=np.float32))
        return [[DMatch(qi, int(idx[qi, j]), float(dist[qi, j])) for j in range(k) if idx[qi, j] >= 0]
                for qi in range(len(idx))]

    def radiusMatch(self, query_points, train_points, max_radius, **kwargs):
        """kNN radius match with k=2 (cv2_helpers.py:296-339)."""
        idx, dist = knn2(np.asarray(query_points, dtype=np.float32), np.asarray(train_points, dtype=np.float32))
        return [[DMatch(qi, int(idx[qi, j]), float(dist[qi, j])) for j in range(2)
                 if idx[qi, j] >= 0 and dist[qi, j] <= max_radius] for qi in range(len(idx))]


def match_radius_ratio_unique(query_points, train_points, max_radius, max_dist_ratio, priority=None):
    """
    The reference's whole matcher use in one call (Work/SLAM/application/own/slam.py:101-125): radiusMatch (k = 2),
    ratio test (a single match within the radius passes, two pass when dist0 / dist1 < max_dist_ratio), then at most one
    match per train point -- the query with the smallest `priority` (the reference's err_OF; default: the match
    distance), the earlier query on equal priority.  Returns {trainIdx: DMatch(queryIdx, trainIdx, distance)} like
    the reference's `best_dist_matches_by_trainIdx`.
    """
    q = np.ascontiguousarray(query_points, dtype=np.float32)
    t = np.ascontiguousarray(train_points, dtype=np.float32)
    if q.ndim != 2 or t.ndim != 2 or q.shape[1] != t.shape[1]:
        raise ValueError("query (Nq, D) and train (Nt, D) must have the same D")
    Nq, Nt, D = len(q), len(t), q.shape[1]
    pr = None
    if priority is not None:
        pr = np.ascontiguousarray(priority, dtype=np.float32)
        if pr.shape != (Nq,):
            raise ValueError("priority must have one entry per query")
    qot = np.full(Nt, -1, dtype=np.int32)
    dot = np.full(Nt, np.inf, dtype=np.float32)
    _lib.check(_lib.lib().mqs_match_radius_ratio_unique(
        _lib.default_context().handle, q.ctypes.data_as(c_f32p), c_i64(Nq), t.ctypes.data_as(c_f32p), c_i64(Nt), int(D),
        ctypes.c_float(max_radius), ctypes.c_double(max_dist_ratio), None if pr is None else pr.ctypes.data_as(c_f32p),
        qot.ctypes.data_as(c_i32p), dot.ctypes.data_as(c_f32p)))
    return {int(ti): DMatch(int(qot[ti]), int(ti), float(dot[ti])) for ti in np.nonzero(qot >= 0)[0]}


def ratio_unique_dev(idx, dist, n_train, max_radius, max_dist_ratio, priority=None, workspace=None):
    """Device-resident filter on the (idx, dist) of any knn2*_dev: returns (query_of_train (Nt,) int32 with -1 for
    unmatched train rows, dist_of_train (Nt,) float32)."""
    import torch
    if not (idx.is_cuda and dist.is_cuda and idx.is_contiguous() and dist.is_contiguous()):
        raise ValueError("idx / dist must be contiguous device tensors")
    if idx.dtype != torch.int32 or dist.dtype != torch.float32 or idx.shape != dist.shape or idx.dim() != 2 or idx.shape[1] != 2:
        raise ValueError("idx (Nq, 2) int32 and dist (Nq, 2) float32 as produced by knn2")
    if priority is not None and not (priority.is_cuda and priority.dtype == torch.float32 and priority.is_contiguous()
                                     and tuple(priority.shape) == (idx.shape[0],)):
        raise ValueError("priority must be a contiguous float32 device tensor with one entry per query")
    Nq, Nt = int(idx.shape[0]), int(n_train)
    qot = torch.empty(Nt, dtype=torch.int32, device=idx.device)
    dot = torch.empty(Nt, dtype=torch.float32, device=idx.device)
    need = int(_lib.lib().mqs_match_ratio_unique_workspace_bytes(Nt))
    ws = workspace if workspace is not None else torch.empty(max(need, 16) // 8 + 1, dtype=torch.int64, device=idx.device)
    _lib.check(_lib.lib().mqs_match_ratio_unique_dev(
        idx.data_ptr(), dist.data_ptr(), Nq, Nt, ctypes.c_float(max_radius), ctypes.c_double(max_dist_ratio),
        None if priority is None else priority.data_ptr(), qot.data_ptr(), dot.data_ptr(), ws.data_ptr(),
        ws.numel() * ws.element_size(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return qot, dot


def camera_pairs(n_cams):
    """The unordered camera pairs (a < b) of a cross-match, in the order they are dealt to ranks."""
    return [(a, b) for a in range(n_cams) for b in range(a + 1, n_cams)]


def cross_match_dev(descriptor_bits, rank=0, world=1, max_radius=float("inf"), max_dist_ratio=0.8):
    """
    BASELINE configs[2]: brute-force cross-match of the packed binary descriptors of every camera against every other
    camera (list of (n_c, D / 8) uint8 device tensors).  For the pair (a, b), a < b, camera a's descriptors are the
    queries and camera b's the train set.  The pairs are independent units: rank r of `world` handles the pairs
    `sharding.unit_shard` deals to it (pair-major, no collective).  Returns
    {(a, b): (idx (n_a, 2), dist (n_a, 2), query_of_train (n_b,), dist_of_train (n_b,))} for this rank's pairs.
    """
    from . import sharding
    pairs = camera_pairs(len(descriptor_bits))
    out = {}
    for k in sharding.unit_shard(len(pairs), rank, world):
        a, b = pairs[k]
        idx, dist = knn2_bits_dev(descriptor_bits[a], descriptor_bits[b])
        qot, dot = ratio_unique_dev(idx, dist, int(descriptor_bits[b].shape[0]), max_radius, max_dist_ratio)
        out[(a, b)] = (idx, dist, qot, dot)
    return out


def binary_descriptors(n, bits=256, seed=7, copies_of=None, copy_frac=0.5, flip_frac=0.1):
    """SURVEY.md 8(d) matcher workload: n descriptors of `bits` i.i.d. Bernoulli(0.5) bits (PCG64(seed)),
    optionally with `copy_frac` of the rows being copies of rows of `copies_of` with `flip_frac` of
    their bits flipped.  Returns a (n, bits) float16 array of {0,1}."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d = rng.integers(0, 2, size=(n, bits), dtype=np.uint8)
    if copies_of is not None:
        m = int(copy_frac * n)
        rows = rng.choice(n, m, replace=False)
        src = rng.integers(0, len(copies_of), m)
        flips = rng.random((m, bits)) < flip_frac
        d[rows] = np.asarray(copies_of[src], dtype=np.uint8) ^ flips.astype(np.uint8)
    return d.astype(np.float16)
