// Per-landmark bundle-adjustment arithmetic shared by the gfx950 kernels in ba.hip:
// projection-factor linearisation (GTSAM GenericProjectionFactor<Pose3, Point3, Cal3DS2>
// conventions, see include/mqslam.h and SURVEY.md Appendix B), elimination of the landmark
// (Schur complement) and its back-substitution.  Replaces the GTSAM work behind
// Work/SLAM/tools/bundle_adjustment/bundle_adjust.cpp:289-298,323-324 (reference paths).
//
// Structure exploited (nothing is materialised per factor):
//   with  q = R^T (p - t) = Z (x, y, 1)  in the camera frame and the whitened 2x3 image Jacobian
//   D = E Pm,  Pm = [[1,0,-x],[0,1,-y]],  E = K_c * Ddist / (sigma Z)  (2x2):
//     J_point = D R^T,   J_pose = D [ [q]x | -I ] = E Jg,   Jg = Pm [ [q]x | -I ]  (2x6, closed form)
//   so every block of the normal equations is a 2x2 core sandwiched between the closed-form Jg's:
//     Hll  = sum_c PR_c^T F_c PR_c          PR_c = Pm_c R_c^T (2x3),  F_c = E_c^T E_c (2x2 sym)
//     S_cc = Jg_c^T (F_c - Uh_c Uh_c^T) Jg_c,    S_cd = -Jg_c^T (Uh_c Uh_d^T) Jg_d   (c != d)
//     Uh_c = F_c PR_c L^-T (2x3),  Hll = L L^T
//   A landmark therefore keeps 9 doubles per camera live (x, y, Z, Uh) instead of 6x3 + 6x6 blocks.
//
// MQS_HD: compiled for the host by tests/host_math.cpp to check the arithmetic against the
// oracle without a GPU (test-only).
#pragma once
#include "tri_math.h"

namespace mqs {
namespace ba {

// Slot layout of the per-landmark contributions (and of the reduced partial sums):
//   [camera c: 21 upper-triangle entries of S_cc (row-major, i <= j), then g_c (6), 5 unused]
//       c = 0..C-1 -- one full 32-slot reduction window per camera, so that no window is open
//       while the next camera's factor is being formed (register pressure)
//   [camera pair (c,d), c < d, in order (0,1),(0,2),..: 36 entries of S_cd row-major]
//   [cost] [number of valid factors]
template <int C>
struct Layout {
    static constexpr int kDiag = 32;
    static constexpr int kDiagUsed = 27;
    static constexpr int kPairs = C * (C - 1) / 2;
    static constexpr int diag_off(int c) { return kDiag * c; }
    static constexpr int pair_index(int c, int d) { return c * (2 * C - c - 1) / 2 + (d - c - 1); }
    static constexpr int pair_off(int c, int d) { return kDiag * C + 36 * pair_index(c, d); }
    static constexpr int kCost = kDiag * C + 36 * kPairs;
    static constexpr int kCount = kCost + 1;
    static constexpr int kSlots = kCount + 1;
    static constexpr int kChunks = (kSlots + 31) / 32;
};

constexpr int kCamStride = 24;     // doubles per camera in the staged parameter block
// staged camera block: R (9, row-major camera-to-world), t (3), fx fy s u0 v0 k1 k2 p1 p2 (9),
// 1/sigma (1), 2*fx/sigma (1: whitened cheirality residual), pad (1)

MQS_HD void stage_camera(double *dst, const double *pose12, const double *calib9, double sigma)
{
    for (int k = 0; k < 12; ++k) dst[k] = pose12[k];
    for (int k = 0; k < 9; ++k) dst[12 + k] = calib9[k];
    dst[21] = 1.0 / sigma;
    dst[22] = 2.0 * calib9[0] / sigma;
    // a camera without lens distortion (k1 = k2 = p1 = p2 = 0: the benchmark's Cal3DS2(480, 480, 0, 320, 240, 0, 0, 0, 0) and every
    // rectified data set): flagged once here, so that the kernels whose camera is wave-uniform can skip the distortion model
    // and its 2 x 2 Jacobian (~ 40 of a factor's ~ 100 instructions) -- see make_factor
    dst[23] = (calib9[5] == 0.0 && calib9[6] == 0.0 && calib9[7] == 0.0 && calib9[8] == 0.0) ? 1.0 : 0.0;
}

// The flag of a camera block as a wave-uniform value (a scalar branch on the device).  ONLY for kernels in which every lane of a
// wave works on the same camera (the dense-visibility kernels); the sparse kernels, where the camera follows the observation,
// pass false to make_factor.
MQS_HD bool camera_without_distortion(const double *cam)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane((int)(cam[23] != 0.0)) != 0;
#else
    return cam[23] != 0.0;
#endif
}

// One projection factor, reduced to what the elimination needs.
struct Factor {
    double x, y, Z;        // q = Z (x, y, 1)
    double F00, F01, F11;  // E^T E
    double f0, f1;         // E^T e   (e = whitened residual)
    double half_e2;        // 0.5 |e|^2
    bool valid;            // observed and in front of the camera
};

// cam: staged camera block.  uv: measurement.  observed == false -> contributes nothing.
// nodist: the camera has no lens distortion (must be wave-uniform on the device): g = 1 and the distortion Jacobian is the
// identity, so E = K / (sigma Z).  Same values as the general path (it multiplies by exact ones and zeros there).
MQS_HD Factor make_factor(const double *cam, const double px, const double py, const double pz, double u,
                          double v, bool observed, bool nodist = false)
{
    Factor o;
    const double dx = px - cam[9], dy = py - cam[10], dz = pz - cam[11];
    // q = R^T d
    const double X = fma(cam[0], dx, fma(cam[3], dy, cam[6] * dz));
    const double Y = fma(cam[1], dx, fma(cam[4], dy, cam[7] * dz));
    const double Z = fma(cam[2], dx, fma(cam[5], dy, cam[8] * dz));
    const bool front = Z > 0.0;
    const double iz = rcp(front ? Z : 1.0);
    const double x = X * iz, y = Y * iz;
    const double fx = cam[12], fy = cam[13], sk = cam[14], u0 = cam[15], v0 = cam[16];
    const double isig = cam[21];
    const double sc = isig * iz;
    double eu, ev, E00, E01, E10, E11;
    if (nodist) {
        eu = (fma(fx, x, fma(sk, y, u0)) - u) * isig;
        ev = (fma(fy, y, v0) - v) * isig;
        E00 = fx * sc; E01 = sk * sc; E10 = 0.0; E11 = fy * sc;
    } else {
        const double k1 = cam[17], k2 = cam[18], p1 = cam[19], p2 = cam[20];
        const double xx = x * x, yy = y * y, xy = x * y;
        const double r2 = xx + yy;
        const double g = fma(r2, fma(k2, r2, k1), 1.0);
        const double dg = fma(2.0 * k2, r2, k1);
        const double xd = fma(g, x, fma(2.0 * p1, xy, p2 * fma(2.0, xx, r2)));
        const double yd = fma(g, y, fma(2.0 * p2, xy, p1 * fma(2.0, yy, r2)));
        eu = (fma(fx, xd, fma(sk, yd, u0)) - u) * isig;
        ev = (fma(fy, yd, v0) - v) * isig;
        // 2x2 distortion Jacobian
        const double a = fma(2.0 * xx, dg, g) + 2.0 * p1 * y + 6.0 * p2 * x;
        const double b = fma(2.0 * xy, dg, 2.0 * p1 * x) + 2.0 * p2 * y;
        const double d = fma(2.0 * yy, dg, g) + 2.0 * p2 * x + 6.0 * p1 * y;
        // E = [[fx, sk],[0, fy]] * [[a, b],[b, d]] * (isig / Z)
        E00 = fma(fx, a, sk * b) * sc; E01 = fma(fx, b, sk * d) * sc;
        E10 = fy * b * sc; E11 = fy * d * sc;
    }
    const bool ok = observed && front;
    // selects, not multiplications by 0: masked / behind-camera slots may hold NaN measurements
    o.x = ok ? x : 0.0; o.y = ok ? y : 0.0; o.Z = ok ? Z : 1.0;
    o.F00 = ok ? fma(E00, E00, E10 * E10) : 0.0;
    o.F01 = ok ? fma(E00, E01, E10 * E11) : 0.0;
    o.F11 = ok ? fma(E01, E01, E11 * E11) : 0.0;
    o.f0 = ok ? fma(E00, eu, E10 * ev) : 0.0;
    o.f1 = ok ? fma(E01, eu, E11 * ev) : 0.0;
    // cheirality: constant residual 2*fx/sigma in both components, zero Jacobians
    const double ce = cam[22];
    o.half_e2 = observed ? (front ? 0.5 * fma(eu, eu, ev * ev) : ce * ce) : 0.0;
    o.valid = ok;
    return o;
}

// PR = Pm R^T (2x3):  rows  R[:,0]^T - x R[:,2]^T   and   R[:,1]^T - y R[:,2]^T   as vectors over
// the world axes k:  PR[0][k] = R[k][0] - x R[k][2],  PR[1][k] = R[k][1] - y R[k][2].
MQS_HD void make_PR(const double *cam, double x, double y, double PR[2][3])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        PR[0][k] = fma(-x, cam[3 * k + 2], cam[3 * k + 0]);
        PR[1][k] = fma(-y, cam[3 * k + 2], cam[3 * k + 1]);
    }
}

// Jg = Pm [ [q]x | -I ]  (2x6), q = Z (x, y, 1)
MQS_HD void make_Jg(double x, double y, double Z, double Jg[2][6])
{
    Jg[0][0] = Z * x * y;          Jg[0][1] = -Z * fma(x, x, 1.0); Jg[0][2] = Z * y;
    Jg[0][3] = -1.0;               Jg[0][4] = 0.0;                 Jg[0][5] = x;
    Jg[1][0] = Z * fma(y, y, 1.0); Jg[1][1] = -Z * x * y;          Jg[1][2] = -Z * x;
    Jg[1][3] = 0.0;                Jg[1][4] = -1.0;                Jg[1][5] = y;
}

// The geometric Jacobian is Jg = [A | B] with A = Z [[xy, -(1+x^2), y], [1+y^2, -xy, -x]] and
// B = [[-1, 0, x], [0, -1, y]]: columns 3 and 4 are minus unit vectors.  The block products below use
// that structure instead of multiplying by the literal 0 / -1 entries (which the compiler may not
// fold without fast-math): rows 3 and 4 of Jg^T T are just -T[0][:] and -T[1][:].
struct JgA { double a00, a01, a02, a10, a11, a12, x, y; };

MQS_HD JgA make_JgA(double x, double y, double Z)
{
    JgA j;
    j.a00 = Z * x * y;          j.a01 = -Z * fma(x, x, 1.0); j.a02 = Z * y;
    j.a10 = Z * fma(y, y, 1.0); j.a11 = -Z * x * y;          j.a12 = -Z * x;
    j.x = x; j.y = y;
    return j;
}

// T = k Jg (2x6) for a general 2x2 k = [[k00,k01],[k10,k11]]
MQS_HD void k_times_Jg(double k00, double k01, double k10, double k11, const JgA &j, double T[2][6])
{
    T[0][0] = fma(k00, j.a00, k01 * j.a10); T[0][1] = fma(k00, j.a01, k01 * j.a11); T[0][2] = fma(k00, j.a02, k01 * j.a12);
    T[1][0] = fma(k10, j.a00, k11 * j.a10); T[1][1] = fma(k10, j.a01, k11 * j.a11); T[1][2] = fma(k10, j.a02, k11 * j.a12);
    T[0][3] = -k00; T[0][4] = -k01; T[0][5] = fma(k00, j.x, k01 * j.y);
    T[1][3] = -k10; T[1][4] = -k11; T[1][5] = fma(k10, j.x, k11 * j.y);
}

// entry (i, jj) of Jg^T T
MQS_HD double JgT_T(const JgA &j, const double T[2][6], int i, int jj)
{
    switch (i) {
    case 0: return fma(j.a00, T[0][jj], j.a10 * T[1][jj]);
    case 1: return fma(j.a01, T[0][jj], j.a11 * T[1][jj]);
    case 2: return fma(j.a02, T[0][jj], j.a12 * T[1][jj]);
    case 3: return -T[0][jj];
    case 4: return -T[1][jj];
    default: return fma(j.x, T[0][jj], j.y * T[1][jj]);
    }
}

// entry i of Jg^T r for a 2-vector r
MQS_HD double JgT_r(const JgA &j, double r0, double r1, int i)
{
    switch (i) {
    case 0: return fma(j.a00, r0, j.a10 * r1);
    case 1: return fma(j.a01, r0, j.a11 * r1);
    case 2: return fma(j.a02, r0, j.a12 * r1);
    case 3: return -r0;
    case 4: return -r1;
    default: return fma(j.x, r0, j.y * r1);
    }
}

// Landmark block and its Cholesky factor.
struct PointSystem {
    Sym3 H;      // Hll (with prior and damping)
    Vec3 g;      // gl = -sum J_point^T e  (- prior gradient)
    // Cholesky H = L L^T:  L = [[l00,0,0],[l10,l11,0],[l20,l21,l22]], inverse diagonal kept
    double l10, l20, l21, i00, i11, i22;
    bool ok;
};

MQS_HD void point_add_factor(PointSystem &ps, const Factor &fc, const double PR[2][3])
{
    // W = F PR (2x3);  H += PR^T W;  g -= PR^T f
    double W[2][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        W[0][k] = fma(fc.F00, PR[0][k], fc.F01 * PR[1][k]);
        W[1][k] = fma(fc.F01, PR[0][k], fc.F11 * PR[1][k]);
    }
    ps.H.xx = fma(PR[0][0], W[0][0], fma(PR[1][0], W[1][0], ps.H.xx));
    ps.H.xy = fma(PR[0][0], W[0][1], fma(PR[1][0], W[1][1], ps.H.xy));
    ps.H.xz = fma(PR[0][0], W[0][2], fma(PR[1][0], W[1][2], ps.H.xz));
    ps.H.yy = fma(PR[0][1], W[0][1], fma(PR[1][1], W[1][1], ps.H.yy));
    ps.H.yz = fma(PR[0][1], W[0][2], fma(PR[1][1], W[1][2], ps.H.yz));
    ps.H.zz = fma(PR[0][2], W[0][2], fma(PR[1][2], W[1][2], ps.H.zz));
    ps.g.x -= fma(PR[0][0], fc.f0, PR[1][0] * fc.f1);
    ps.g.y -= fma(PR[0][1], fc.f0, PR[1][1] * fc.f1);
    ps.g.z -= fma(PR[0][2], fc.f0, PR[1][2] * fc.f1);
}

// Adds prior w*(p - p0) and the LM damping (lambda*diag(H), or |lambda|*I for lambda < 0), then factors.  A landmark without any
// constraint (all factors masked/behind, no prior) gets ok == false: it contributes nothing to
// the reduced system and is left unchanged by the back-substitution.
MQS_HD void point_finish(PointSystem &ps, double prior_w, double dpx, double dpy, double dpz, double lambda)
{
    ps.H.xx += prior_w; ps.H.yy += prior_w; ps.H.zz += prior_w;
    ps.g.x = fma(-prior_w, dpx, ps.g.x);
    ps.g.y = fma(-prior_w, dpy, ps.g.y);
    ps.g.z = fma(-prior_w, dpz, ps.g.z);
    // lambda >= 0: Marquardt scaling lambda * diag(H);  lambda < 0: Levenberg damping |lambda| * I, what GTSAM 3.2.1's
    // LevenbergMarquardtParams default (diagonalDamping = false) adds to every variable (include/mqslam.h, "damping")
    const double mul = lambda > 0.0 ? lambda : 0.0, add = lambda < 0.0 ? -lambda : 0.0;
    ps.H.xx = fma(mul, ps.H.xx, ps.H.xx) + add;
    ps.H.yy = fma(mul, ps.H.yy, ps.H.yy) + add;
    ps.H.zz = fma(mul, ps.H.zz, ps.H.zz) + add;
    const double thr = 1e-14 * (ps.H.xx + ps.H.yy + ps.H.zz);
    const double d0 = ps.H.xx;
    const double r0 = (d0 > 0.0) ? rsqrt_d(d0) : 0.0;      // 1/l00
    ps.i00 = r0;
    ps.l10 = ps.H.xy * r0;
    ps.l20 = ps.H.xz * r0;
    const double d1 = fma(-ps.l10, ps.l10, ps.H.yy);
    const double r1 = (d1 > 0.0) ? rsqrt_d(d1) : 0.0;
    ps.i11 = r1;
    ps.l21 = fma(-ps.l20, ps.l10, ps.H.yz) * r1;
    const double d2 = fma(-ps.l21, ps.l21, fma(-ps.l20, ps.l20, ps.H.zz));
    const double r2 = (d2 > 0.0) ? rsqrt_d(d2) : 0.0;
    ps.i22 = r2;
    ps.ok = (d0 > thr) && (d1 > thr) && (d2 > thr);
}

// row vector (r0,r1,r2) <- (r0,r1,r2) L^-T   i.e. solve  y L^T = r  <=>  L y^T = r^T
MQS_HD void apply_LinvT(const PointSystem &ps, double &r0, double &r1, double &r2)
{
    r0 = r0 * ps.i00;
    r1 = fma(-ps.l10, r0, r1) * ps.i11;
    r2 = fma(-ps.l21, r1, fma(-ps.l20, r0, r2)) * ps.i22;
}

// v <- L^-T v  (back substitution with L^T), used for  dp = L^-T (L^-1 rhs)
MQS_HD void apply_Lt_inv(const PointSystem &ps, double &v0, double &v1, double &v2)
{
    v2 = v2 * ps.i22;
    v1 = fma(-ps.l21, v2, v1) * ps.i11;
    v0 = fma(-ps.l20, v2, fma(-ps.l10, v1, v0)) * ps.i00;
}

// Observation access.  `Obs` provides  void get(int c, double &u, double &v, bool &seen) const;
// the kernels re-read the measurement from global memory (L2-resident) at every use instead of
// holding 2C doubles per landmark in registers; the host harness reads its arrays.

// Optional per-thread parking place for what the first pass over the cameras knows and the second needs again (F = E^T E
// and f = E^T e: 5 doubles per camera).  The second pass re-forms the factor from scratch without it (~100 instructions per
// camera, of which ~70 are the residual, the distortion Jacobian and E); the device emitter parks them in LDS.
struct NoFactorStash {
    static constexpr bool kEnabled = false;
    MQS_HD_MEMBER void put(int, double, double, double, double, double) const {}
    MQS_HD_MEMBER void get(int, double &, double &, double &, double &, double &) const {}
};

// Accumulates camera c's factor into the landmark block; returns the factor.
template <class Obs>
MQS_HD Factor add_camera(const double *cam, const Obs &obs, int c, double px, double py, double pz, bool live,
                         PointSystem &ps)
{
    double u, v;
    bool seen;
    obs.get(c, u, v, seen);
    const Factor fc = make_factor(cam, px, py, pz, u, v, seen && live, camera_without_distortion(cam));
    double PR[2][3];
    make_PR(cam, fc.x, fc.y, PR);
    point_add_factor(ps, fc, PR);
    return fc;
}

// Phase A shared by linearise / back-substitution: Hll, gl, cost, count over all cameras.
// The camera loop is deliberately NOT unrolled (small live state, high occupancy).
template <int C, class Obs, class Stash = NoFactorStash>
MQS_HD void landmark_point_system(const double *cams, const Obs &obs, double px, double py, double pz, bool live,
                                  double prior_w, double dpx, double dpy, double dpz, double lambda,
                                  PointSystem &ps, double &cost, double &count, const Stash &stash = Stash())
{
    ps.H = Sym3{0, 0, 0, 0, 0, 0};
    ps.g = Vec3{0, 0, 0};
    cost = 0.5 * prior_w * fma(dpx, dpx, fma(dpy, dpy, dpz * dpz));
    count = 0.0;
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
        const Factor fc = add_camera(cams + kCamStride * c, obs, c, px, py, pz, live, ps);
        if (Stash::kEnabled) stash.put(c, fc.F00, fc.F01, fc.F11, fc.f0, fc.f1);
        cost += fc.half_e2;
        count += fc.valid ? 1.0 : 0.0;
    }
    if (!live) cost = 0.0;
    point_finish(ps, live ? prior_w : 0.0, dpx, dpy, dpz, lambda);
}

// Everything the landmark contributes to the reduced camera system, emitted slot by slot.
// Emit must provide  void put(int slot, double value)  and  void flush(int window)  (slot / window
// are compile-time constants after unrolling; the device emitter relies on that to keep its
// 32-entry window in registers).
template <int C, class Obs, class Emit, class Stash = NoFactorStash>
MQS_HD void landmark_contribution(const double *cams, const Obs &obs, double px, double py, double pz,
                                  double prior_w, double dpx, double dpy, double dpz, double lambda, bool live,
                                  Emit &em, const Stash &stash = Stash())
{
    using L = Layout<C>;
    PointSystem ps;
    double cost, count;
    landmark_point_system<C, Obs, Stash>(cams, obs, px, py, pz, live, prior_w, dpx, dpy, dpz, lambda, ps, cost, count, stash);
    const double m = ps.ok ? 1.0 : 0.0;          // unconstrained landmark: pose blocks keep only J_pose^T J_pose
    // w = L^-1 gl
    double w0 = ps.g.x * ps.i00;
    double w1 = fma(-ps.l10, w0, ps.g.y) * ps.i11;
    double w2 = fma(-ps.l21, w1, fma(-ps.l20, w0, ps.g.z)) * ps.i22;
    w0 *= m; w1 *= m; w2 *= m;

    double cx[C], cy[C], cZ[C];
    double Uh[C][2][3];
    // ---- diagonal blocks and gradient: the factor is re-formed (cheaper than keeping it) ----
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double *cam = cams + kCamStride * c + opaque_zero();
        Factor fc;
        if (Stash::kEnabled) {
            // only the geometry is re-formed: x, y, Z exactly as make_factor computes them; F and f come back from the stash
            // (exact zeros for an unused factor, so x, y, Z of such a factor only ever multiply zeros)
            const double dx = px - cam[9], dy = py - cam[10], dz = pz - cam[11];
            const double X = fma(cam[0], dx, fma(cam[3], dy, cam[6] * dz));
            const double Y = fma(cam[1], dx, fma(cam[4], dy, cam[7] * dz));
            const double Z = fma(cam[2], dx, fma(cam[5], dy, cam[8] * dz));
            const bool front = Z > 0.0;
            const double iz = rcp(front ? Z : 1.0);
            stash.get(c, fc.F00, fc.F01, fc.F11, fc.f0, fc.f1);
            const bool used = (fc.F00 != 0.0) || (fc.F11 != 0.0);
            fc.x = used ? X * iz : 0.0; fc.y = used ? Y * iz : 0.0; fc.Z = used ? Z : 1.0;
        } else {
            double u, v;
            bool seen;
            obs.get(c, u, v, seen);
            fc = make_factor(cam, px, py, pz, u, v, seen && live, camera_without_distortion(cam));
        }
        cx[c] = fc.x; cy[c] = fc.y; cZ[c] = fc.Z;
        double PR[2][3];
        make_PR(cam, fc.x, fc.y, PR);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const double Fa = r ? fc.F01 : fc.F00, Fb = r ? fc.F11 : fc.F01;
            double u0 = fma(Fa, PR[0][0], Fb * PR[1][0]);
            double u1 = fma(Fa, PR[0][1], Fb * PR[1][1]);
            double u2 = fma(Fa, PR[0][2], Fb * PR[1][2]);
            apply_LinvT(ps, u0, u1, u2);
            Uh[c][r][0] = m * u0; Uh[c][r][1] = m * u1; Uh[c][r][2] = m * u2;
        }
        // k = F - Uh Uh^T (2x2 sym),  rh = -f - Uh w
        const double k00 = fc.F00 - fma(Uh[c][0][0], Uh[c][0][0], fma(Uh[c][0][1], Uh[c][0][1], Uh[c][0][2] * Uh[c][0][2]));
        const double k01 = fc.F01 - fma(Uh[c][0][0], Uh[c][1][0], fma(Uh[c][0][1], Uh[c][1][1], Uh[c][0][2] * Uh[c][1][2]));
        const double k11 = fc.F11 - fma(Uh[c][1][0], Uh[c][1][0], fma(Uh[c][1][1], Uh[c][1][1], Uh[c][1][2] * Uh[c][1][2]));
        const double rh0 = -fc.f0 - fma(Uh[c][0][0], w0, fma(Uh[c][0][1], w1, Uh[c][0][2] * w2));
        const double rh1 = -fc.f1 - fma(Uh[c][1][0], w0, fma(Uh[c][1][1], w1, Uh[c][1][2] * w2));
        MQS_SCHED_FENCE();
        const JgA jg = make_JgA(cx[c], cy[c], cZ[c]);
        double T[2][6];                           // T = k Jg
        k_times_Jg(k00, k01, k01, k11, jg, T);
        int slot = L::diag_off(c);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) em.put(slot++, JgT_T(jg, T, i, j));
#pragma unroll
        for (int i = 0; i < 6; ++i) em.put(slot++, JgT_r(jg, rh0, rh1, i));
        em.flush(L::diag_off(c) >> 5);            // the camera's window (5 unused entries)
        MQS_SCHED_FENCE();
    }
    // ---- off-diagonal blocks: S_cd = -Jg_c^T (Uh_c Uh_d^T) Jg_d ----
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const JgA jc = make_JgA(cx[c], cy[c], cZ[c]);
#pragma unroll
        for (int d = c + 1; d < C; ++d) {
            const double k00 = -fma(Uh[c][0][0], Uh[d][0][0], fma(Uh[c][0][1], Uh[d][0][1], Uh[c][0][2] * Uh[d][0][2]));
            const double k01 = -fma(Uh[c][0][0], Uh[d][1][0], fma(Uh[c][0][1], Uh[d][1][1], Uh[c][0][2] * Uh[d][1][2]));
            const double k10 = -fma(Uh[c][1][0], Uh[d][0][0], fma(Uh[c][1][1], Uh[d][0][1], Uh[c][1][2] * Uh[d][0][2]));
            const double k11 = -fma(Uh[c][1][0], Uh[d][1][0], fma(Uh[c][1][1], Uh[d][1][1], Uh[c][1][2] * Uh[d][1][2]));
            const JgA jd = make_JgA(cx[d], cy[d], cZ[d]);
            double T[2][6];
            k_times_Jg(k00, k01, k10, k11, jd, T);
            int slot = L::pair_off(c, d);
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) em.put(slot++, JgT_T(jc, T, i, j));
            MQS_SCHED_FENCE();
        }
    }
    em.put(L::kCost, cost);
    em.put(L::kCount, count);
    if ((L::kSlots & 31) != 0) em.flush(L::kChunks - 1);   // partial last window
}

// Landmark update dp = Hll^-1 (gl - sum_c Hpl_c^T dxi_c) at the same linearisation point.
template <int C, class Obs>
MQS_HD Vec3 landmark_backsub(const double *cams, const Obs &obs, double px, double py, double pz, double prior_w,
                             double dpx, double dpy, double dpz, double lambda, const double *dpose)
{
    PointSystem ps;
    ps.H = Sym3{0, 0, 0, 0, 0, 0};
    ps.g = Vec3{0, 0, 0};
    double rx = 0, ry = 0, rz = 0;                 // sum_c Hpl_c^T dxi_c = sum_c PR^T F (Jg dxi)
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
        const double *cam = cams + kCamStride * c;
        double u, v;
        bool seen;
        obs.get(c, u, v, seen);
        const Factor fc = make_factor(cam, px, py, pz, u, v, seen, camera_without_distortion(cam));
        double PR[2][3];
        make_PR(cam, fc.x, fc.y, PR);
        point_add_factor(ps, fc, PR);
        double Jg[2][6];
        make_Jg(fc.x, fc.y, fc.Z, Jg);
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            s0 = fma(Jg[0][j], dpose[6 * c + j], s0);
            s1 = fma(Jg[1][j], dpose[6 * c + j], s1);
        }
        const double t0 = fma(fc.F00, s0, fc.F01 * s1), t1 = fma(fc.F01, s0, fc.F11 * s1);
        rx = fma(PR[0][0], t0, fma(PR[1][0], t1, rx));
        ry = fma(PR[0][1], t0, fma(PR[1][1], t1, ry));
        rz = fma(PR[0][2], t0, fma(PR[1][2], t1, rz));
    }
    point_finish(ps, prior_w, dpx, dpy, dpz, lambda);
    double v0 = ps.g.x - rx, v1 = ps.g.y - ry, v2 = ps.g.z - rz;
    // L y = v, then L^T dp = y
    v0 = v0 * ps.i00;
    v1 = fma(-ps.l10, v0, v1) * ps.i11;
    v2 = fma(-ps.l21, v1, fma(-ps.l20, v0, v2)) * ps.i22;
    apply_Lt_inv(ps, v0, v1, v2);
    const double m = ps.ok ? 1.0 : 0.0;
    Vec3 dp = {m * v0, m * v1, m * v2};
    return dp;
}

template <int C, class Obs>
MQS_HD void landmark_cost(const double *cams, const Obs &obs, double px, double py, double pz, double prior_w,
                          double dpx, double dpy, double dpz, double &cost, double &count)
{
    cost = 0.5 * prior_w * fma(dpx, dpx, fma(dpy, dpy, dpz * dpz));
    count = 0.0;
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
        double u, v;
        bool seen;
        obs.get(c, u, v, seen);
        const Factor fc = make_factor(cams + kCamStride * c, px, py, pz, u, v, seen, camera_without_distortion(cams + kCamStride * c));
        cost += fc.half_e2;
        count += fc.valid ? 1.0 : 0.0;
    }
}

}  // namespace ba
}  // namespace mqs


namespace mqs {
namespace ba {

// Where a reduced slot lands in the public output vector
//   out = [ S (6C x 6C row-major) | g (6C) | cost | valid-factor count ].
// o2 is the mirrored entry of the symmetric matrix (or -1).
template <int C>
MQS_HD void slot_to_out(int slot, int &o1, int &o2)
{
    using L = Layout<C>;
    constexpr int n6 = 6 * C;
    o1 = -1;
    o2 = -1;
    if (slot < L::kDiag * C) {
        const int c = slot / L::kDiag, r = slot % L::kDiag;
        if (r >= L::kDiagUsed) return;
        if (r < 21) {
            int i = 0, rem = r;
            while (rem >= 6 - i) { rem -= 6 - i; ++i; }
            const int j = i + rem;
            o1 = (6 * c + i) * n6 + 6 * c + j;
            if (i != j) o2 = (6 * c + j) * n6 + 6 * c + i;
        } else {
            o1 = n6 * n6 + 6 * c + (r - 21);
        }
    } else if (slot < L::kCost) {
        int pidx = (slot - L::kDiag * C) / 36;
        const int r = (slot - L::kDiag * C) % 36;
        int c = 0;
        while (pidx >= C - 1 - c) { pidx -= C - 1 - c; ++c; }
        const int d = c + 1 + pidx;
        const int i = r / 6, j = r % 6;
        o1 = (6 * c + i) * n6 + 6 * d + j;
        o2 = (6 * d + j) * n6 + 6 * c + i;
    } else if (slot == L::kCost) {
        o1 = n6 * n6 + n6;
    } else if (slot == L::kCount) {
        o1 = n6 * n6 + n6 + 1;
    }
}

}  // namespace ba
}  // namespace mqs
