"""
Host-side mirror of the reference interface (triangulation.py / triangulation_c/__init__.py):
argument validation that must raise BEFORE the native call, dtype rules, output-dtype knob.
No GPU needed: every case here fails (or returns) before a kernel would be launched.
"""
import numpy as np
import pytest


def test_public_names(mqs):
    t = mqs.triangulation
    for name in ("linear_eigen_triangulation", "linear_LS_triangulation", "iterative_LS_triangulation",
                 "polynomial_triangulation", "set_triangl_output_dtype", "output_dtype"):
        assert hasattr(t, name)
    assert t.output_dtype is float


def test_P_is_not_cast(mqs):
    # the reference does not cast P: a float32 P is a TypeError at the weave boundary
    with pytest.raises(TypeError):
        mqs.triangulation.linear_LS_triangulation(np.zeros((3, 2)), np.eye(4, dtype=np.float32),
                                                  np.zeros((3, 2)), np.eye(4))


def test_shape_errors(mqs):
    with pytest.raises(ValueError):
        mqs.triangulation.linear_LS_triangulation(np.zeros((3, 3)), np.eye(4), np.zeros((3, 2)), np.eye(4))
    with pytest.raises(ValueError):
        mqs.triangulation.iterative_LS_triangulation(np.zeros((3, 2)), np.eye(4), np.zeros((4, 2)), np.eye(4))
    with pytest.raises(ValueError):
        mqs.triangulation.linear_LS_triangulation(np.zeros((3, 2)), np.eye(2, 4), np.zeros((3, 2)), np.eye(4))
    with pytest.raises(ValueError):
        mqs.triangulation_c.linear_LS_triangulation_nview(np.zeros((9, 3, 2)), np.zeros((9, 3, 4)))


def test_integer_u_raises_like_reference(mqs):
    # np.finfo(int dtype) raises ValueError in the reference's cast test (triangulation_c/__init__.py:32)
    with pytest.raises(ValueError):
        mqs.triangulation.linear_LS_triangulation(np.zeros((3, 2), dtype=np.int64), np.eye(4),
                                                  np.zeros((3, 2)), np.eye(4))


def test_polynomial_out_of_scope(mqs):
    with pytest.raises(NotImplementedError):
        mqs.triangulation.polynomial_triangulation(None, None, None, None)


def test_output_dtype_knob(mqs):
    t = mqs.triangulation
    try:
        t.set_triangl_output_dtype(np.float32)
        assert t._cast(np.zeros((2, 3))).dtype == np.float32
        t.set_triangl_output_dtype(float)
        x = np.zeros((2, 3))
        assert t._cast(x) is x
    finally:
        t.set_triangl_output_dtype(float)


def test_synthetic_generator_shapes(mqs):
    u, P, pts = mqs.synthetic.triangulation_problem(1000, 4)
    assert u.shape == (4, 1000, 2) and P.shape == (4, 3, 4) and pts.shape == (1000, 3)
    assert np.all(np.linalg.norm(pts, axis=1) <= 4.0 + 1e-12)
    # pixel-discretised: u*f + c are integers
    px = u * 480.0 + np.array([320.0, 240.0])
    assert np.allclose(px, np.rint(px), atol=1e-9)
    u2, P2, _ = mqs.synthetic.triangulation_problem(1000, 2)
    np.testing.assert_array_equal(P2, P[:2])


@pytest.mark.gpu
def test_float32_observations_through_the_facade(gpu):
    """slam2.py passes float32 image points (slam2.py:19): the facade sends them as float32 and the kernel widens them --
    bit for bit what the reference's rule (widen on the host, triangulation_c/__init__.py:32-33) gives."""
    u, P, _ = gpu.synthetic.triangulation_problem(5003, 2)
    u32 = u.astype(np.float32)
    wide = u32.astype(np.float64)
    t = gpu.triangulation
    for fn in (t.linear_LS_triangulation, t.iterative_LS_triangulation):
        x32, s32 = fn(u32[0], P[0], u32[1], P[1])
        xw, sw = fn(wide[0], P[0], wide[1], P[1])
        np.testing.assert_array_equal(x32, xw)
        np.testing.assert_array_equal(s32, sw)
    # mixed dtypes fall back to the reference's host widening
    xm, _ = t.linear_LS_triangulation(u32[0], P[0], wide[1], P[1])
    np.testing.assert_array_equal(xm, t.linear_LS_triangulation(wide[0], P[0], wide[1], P[1])[0])
