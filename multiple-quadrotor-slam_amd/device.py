"""
Device-resident entry points: torch tensors in, torch tensors out, no host round trip.
torch is used only for device memory and the current HIP stream; the arithmetic is the
library's HIP kernels, reached through the `*_dev` C-ABI functions with raw pointers.
"""
import ctypes

from . import _lib


def _torch():
    import torch
    return torch


def _stream_ptr():
    torch = _torch()
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_dev(t, dtype, name):
    torch = _torch()
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise ValueError("%s must be a device tensor" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def _tri_args(u, P):
    torch = _torch()
    _check_dev(u, torch.float64, "u")
    _check_dev(P, torch.float64, "P")
    if u.dim() != 3 or u.shape[2] != 2:
        raise ValueError("u must have shape (C, N, 2)")
    if tuple(P.shape) != (u.shape[0], 3, 4):
        raise ValueError("P must have shape (C, 3, 4)")
    return int(u.shape[0]), int(u.shape[1])


def linear_LS_triangulation(u, P, out=None):
    """u (C,N,2) f64, P (C,3,4) f64 device tensors -> x (N,3) f64 (status is all-True by definition)."""
    torch = _torch()
    C, N = _tri_args(u, P)
    x = out if out is not None else torch.empty((N, 3), dtype=torch.float64, device=u.device)
    _lib.check(_lib.lib().mqs_triangulate_linear_ls_dev(u.data_ptr(), P.data_ptr(), C, N, x.data_ptr(),
                                                        _stream_ptr()))
    return x


def iterative_LS_triangulation(u, P, tolerance=3.e-5, max_iter=10, out=None, out_status=None):
    """-> x (N,3) f64, status (N,) int32 (codes of triangulation.c:154-159, N-view generalised)."""
    torch = _torch()
    C, N = _tri_args(u, P)
    x = out if out is not None else torch.empty((N, 3), dtype=torch.float64, device=u.device)
    st = out_status if out_status is not None else torch.empty((N,), dtype=torch.int32, device=u.device)
    _lib.check(_lib.lib().mqs_triangulate_iterative_ls_dev(u.data_ptr(), P.data_ptr(), C, N, float(tolerance),
                                                           int(max_iter), x.data_ptr(), st.data_ptr(),
                                                           _stream_ptr()))
    return x, st


def linear_and_iterative_LS_triangulation(u, P, tolerance=3.e-5, max_iter=10, out_ls=None, out_it=None, out_status=None):
    """Both least-squares methods in ONE pass over the observations (mqs_triangulate_ls_and_iterative_dev): returns
    (x_linear_LS, x_iterative_LS, status)."""
    torch = _torch()
    C, N = _tri_args(u, P)
    x_ls = out_ls if out_ls is not None else torch.empty((N, 3), dtype=torch.float64, device=u.device)
    x_it = out_it if out_it is not None else torch.empty((N, 3), dtype=torch.float64, device=u.device)
    st = out_status if out_status is not None else torch.empty((N,), dtype=torch.int32, device=u.device)
    _lib.check(_lib.lib().mqs_triangulate_ls_and_iterative_dev(u.data_ptr(), P.data_ptr(), C, N, float(tolerance), int(max_iter),
                                                               x_ls.data_ptr(), x_it.data_ptr(), st.data_ptr(), _stream_ptr()))
    return x_ls, x_it, st


def linear_eigen_triangulation(u, P, max_coordinate_value=1.e16, out=None, out_ok=None):
    """-> x (N,3) f64, ok (N,) uint8."""
    torch = _torch()
    C, N = _tri_args(u, P)
    x = out if out is not None else torch.empty((N, 3), dtype=torch.float64, device=u.device)
    ok = out_ok if out_ok is not None else torch.empty((N,), dtype=torch.uint8, device=u.device)
    _lib.check(_lib.lib().mqs_triangulate_linear_eigen_dev(u.data_ptr(), P.data_ptr(), C, N,
                                                           float(max_coordinate_value), x.data_ptr(),
                                                           ok.data_ptr(), _stream_ptr()))
    return x, ok


def triangulate_f32(kind, u32, P, tolerance=3.e-5, max_iter=10, max_coordinate_value=1.e16):
    """Float32 observations (C,N,2) widened on load: returns (x, status | ok | None) exactly as the float64 functions
    return them for `u32.double()`."""
    torch = _torch()
    if not (u32.is_cuda and u32.dtype == torch.float32 and u32.is_contiguous() and u32.dim() == 3 and u32.shape[2] == 2):
        raise ValueError("u32 must be a contiguous float32 device tensor (C, N, 2)")
    _check_dev(P, torch.float64, "P")
    C, N = int(u32.shape[0]), int(u32.shape[1])
    if tuple(P.shape) != (C, 3, 4):
        raise ValueError("P must have shape (C, 3, 4)")
    k = {"linear_ls": 0, "iterative_ls": 1, "linear_eigen": 2}[kind]
    x = torch.empty((N, 3), dtype=torch.float64, device=u32.device)
    st = torch.empty((N,), dtype=torch.int32, device=u32.device) if k == 1 else None
    ok = torch.empty((N,), dtype=torch.uint8, device=u32.device) if k == 2 else None
    _lib.check(_lib.lib().mqs_triangulate_f32_dev(k, u32.data_ptr(), P.data_ptr(), C, N, float(tolerance), int(max_iter),
                                                  float(max_coordinate_value), x.data_ptr(),
                                                  None if st is None else st.data_ptr(), None if ok is None else ok.data_ptr(),
                                                  _stream_ptr()))
    return x, (st if k == 1 else ok)


def time_triangulation(kind, u, P, reps=20, tolerance=3.e-5, max_iter=10):
    """Average kernel duration in ms over `reps` back-to-back launches (hipEvents on the current stream)."""
    torch = _torch()
    f32 = u.dtype == torch.float32                       # float32 observations: the widen-on-load kernels
    if f32:
        C, N = int(u.shape[0]), int(u.shape[1])
    else:
        C, N = _tri_args(u, P)
    x = torch.empty((N, 3), dtype=torch.float64, device=u.device)
    st = torch.empty((N,), dtype=torch.int32, device=u.device)
    ok = torch.empty((N,), dtype=torch.uint8, device=u.device)
    ms = ctypes.c_float(0)
    k = {"linear_ls": 0, "iterative_ls": 1, "linear_eigen": 2}[kind] + (10 if f32 else 0)
    _lib.check(_lib.lib().mqs_time_triangulate_dev(k, u.data_ptr(), P.data_ptr(), C, N, float(tolerance),
                                                   int(max_iter), x.data_ptr(), st.data_ptr(), ok.data_ptr(),
                                                   int(reps), _stream_ptr(), ctypes.byref(ms)))
    return float(ms.value)


def triangulate_pixels(kind, pixels, intr, P, tolerance=3.e-5, max_iter=10, max_coordinate_value=1.e16):
    """Fused undistort + normalise + triangulate: pixels (C,N,2) f64, intr (C,9) f64 = fx fy cx cy k1 k2 p1 p2 k3,
    P (C,3,4).  Returns (x, status | ok | None) like the un-fused functions of the same `kind`."""
    torch = _torch()
    C, N = _tri_args(pixels, P)
    _check_dev(intr, torch.float64, "intr")
    if tuple(intr.shape) != (C, 9):
        raise ValueError("intr must have shape (C, 9)")
    k = {"linear_ls": 0, "iterative_ls": 1, "linear_eigen": 2}[kind]
    x = torch.empty((N, 3), dtype=torch.float64, device=pixels.device)
    st = torch.empty((N,), dtype=torch.int32, device=pixels.device) if k == 1 else None
    ok = torch.empty((N,), dtype=torch.uint8, device=pixels.device) if k == 2 else None
    _lib.check(_lib.lib().mqs_triangulate_pixels_dev(
        k, pixels.data_ptr(), intr.data_ptr(), P.data_ptr(), C, N, float(tolerance), int(max_iter),
        float(max_coordinate_value), x.data_ptr(), None if st is None else st.data_ptr(),
        None if ok is None else ok.data_ptr(), _stream_ptr()))
    return x, (st if k == 1 else ok)
