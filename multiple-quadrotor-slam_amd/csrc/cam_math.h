// Camera-model arithmetic either side of triangulation: the published
// OpenCV 2.4 pinhole + (k1, k2, p1, p2, k3) distortion model used by the reference through
//   cv2.undistortPoints   Work/SLAM/application/own/slam2.py:50-51,551-552,
//                         Work/triangulation_comparison/triangulation_comparison.py:173
//   cv2.projectPoints     Work/python_libs/calibration_tools.py:116-124 (reprojection_error),
//                         Work/triangulation_comparison/triangulation_comparison.py:136-142
// MQS_HD so that tests/host_math.cpp can check it on the CPU (test-only).
#pragma once
#include "tri_math.h"

namespace mqs {
namespace cam {

// intr[9] = fx, fy, cx, cy, k1, k2, p1, p2, k3
constexpr int kUndistortIters = 5;       // cvUndistortPoints in OpenCV 2.4.x: 5 fixed-point iterations

MQS_HD void distort(const double *intr, double x, double y, double &xd, double &yd)
{
    const double k1 = intr[4], k2 = intr[5], p1 = intr[6], p2 = intr[7], k3 = intr[8];
    const double r2 = x * x + y * y;
    const double g = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
    xd = x * g + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
    yd = y * g + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
}

// pixel -> normalised, undistorted
MQS_HD void undistort_pixel(const double *intr, double u, double v, double &x, double &y)
{
    const double k1 = intr[4], k2 = intr[5], p1 = intr[6], p2 = intr[7], k3 = intr[8];
    const double x0 = (u - intr[2]) / intr[0], y0 = (v - intr[3]) / intr[1];
    x = x0; y = y0;
    if (k1 == 0.0 && k2 == 0.0 && p1 == 0.0 && p2 == 0.0 && k3 == 0.0) return;
    for (int j = 0; j < kUndistortIters; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = rcp(1.0 + r2 * (k1 + r2 * (k2 + r2 * k3)));     // v_rcp_f64 + two Newton steps on the device (no division sequence)
        const double dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
        const double dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
        x = (x0 - dx) * icdist;
        y = (y0 - dy) * icdist;
    }
}

// world point -> pixel through P = [R | t] (3x4 row-major, world -> camera); returns depth
MQS_HD double project(const double *P, const double *intr, double px, double py, double pz, double &u, double &v)
{
    const double X = fma(P[0], px, fma(P[1], py, fma(P[2], pz, P[3])));
    const double Y = fma(P[4], px, fma(P[5], py, fma(P[6], pz, P[7])));
    const double Z = fma(P[8], px, fma(P[9], py, fma(P[10], pz, P[11])));
    // cvProjectPoints2 (OpenCV 2.4): `z = z ? 1./z : 1` -- a point in the camera's principal plane is projected with
    // unit depth instead of dividing by zero (the rank-deficient cells of the reference's experiment hit this)
    const double iz = (Z != 0.0) ? 1.0 / Z : 1.0;
    double xd, yd;
    distort(intr, X * iz, Y * iz, xd, yd);
    u = fma(intr[0], xd, intr[2]);
    v = fma(intr[1], yd, intr[3]);
    return Z;
}

}  // namespace cam
}  // namespace mqs
