// Camera-pose-from-points arithmetic (perspective-n-point) shared by the gfx950 kernels in pnp.hip:
// the pose step of the reference's per-frame loop,
//   cv2.solvePnPRansac  Work/SLAM/application/own/slam2.py:453-454   (outlier rejection)
//   cv2.solvePnP        Work/SLAM/application/own/slam2.py:489-490, 576-577, 1156   (pose, refined pose)
// OpenCV 2.4 is not vendored; restated from its published method (CV_ITERATIVE): minimise the pixel
// reprojection error  sum_i | project(R X_i + t) - m_i |^2  over the 6 pose parameters with
// Levenberg-Marquardt (damping: diagonal scaled by 1 + lambda, lambda 1e-3, x10 / :10), started from the
// given pose or from a direct linear transform of the undistorted points.  The minimiser does not
// depend on the parametrisation; here the pose is updated on the left,  R <- Exp(w) R,  t <- Exp(w) t + v.
//
// MQS_HD: compiled for the host by tests/host_math.cpp (test-only).
#pragma once
#include "cam_math.h"
#include "so3_math.h"

namespace mqs {
namespace pnp {

constexpr int kAcc = 28;      // 21 upper-triangle entries of J^T J (row-major, i <= j), J^T r (6), |r|^2

// Adds one correspondence to acc.  P = [R | t] (3x4 row-major, world -> camera).
MQS_HD void accumulate_point(const double *P, const double *intr, double X, double Y, double Z, double u, double v,
                             double *acc)
{
    const double Xc = fma(P[0], X, fma(P[1], Y, fma(P[2], Z, P[3])));
    const double Yc = fma(P[4], X, fma(P[5], Y, fma(P[6], Z, P[7])));
    const double Zc = fma(P[8], X, fma(P[9], Y, fma(P[10], Z, P[11])));
    const double iz = mqs::rcp(Zc);                    // (device: v_rcp_f64 + two Newton steps, five dependent instructions where the IEEE division has ~15)
    const double x = Xc * iz, y = Yc * iz;
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    const double k1 = intr[4], k2 = intr[5], p1 = intr[6], p2 = intr[7], k3 = intr[8];
    const double xx = x * x, yy = y * y, xy = x * y, r2 = xx + yy;
    const double g = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
    const double dg = k1 + r2 * (2.0 * k2 + 3.0 * k3 * r2);
    const double xd = x * g + 2.0 * p1 * xy + p2 * (r2 + 2.0 * xx);
    const double yd = y * g + p1 * (r2 + 2.0 * yy) + 2.0 * p2 * xy;
    const double ru = fma(fx, xd, cx) - u, rv = fma(fy, yd, cy) - v;
    // d(xd, yd) / d(x, y)
    const double a = g + 2.0 * xx * dg + 2.0 * p1 * y + 6.0 * p2 * x;
    const double b = 2.0 * xy * dg + 2.0 * p1 * x + 2.0 * p2 * y;
    const double d = g + 2.0 * yy * dg + 6.0 * p1 * y + 2.0 * p2 * x;
    // E = diag(fx, fy) [[a, b], [b, d]] / Zc;   d(u, v)/d(Xc) = E [[1, 0, -x], [0, 1, -y]]
    const double E00 = fx * a * iz, E01 = fx * b * iz, E10 = fy * b * iz, E11 = fy * d * iz;
    // d(Xc)/d(w) = -[Xc]x, d(Xc)/d(v) = I:   Jg = [[1,0,-x],[0,1,-y]] [ -[Xc]x | I ]
    const double Jg[2][6] = {{-Zc * xy, Zc * (1.0 + xx), -Zc * y, 1.0, 0.0, -x},
                             {-Zc * (1.0 + yy), Zc * xy, Zc * x, 0.0, 1.0, -y}};
    double J[2][6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        J[0][k] = fma(E00, Jg[0][k], E01 * Jg[1][k]);
        J[1][k] = fma(E10, Jg[0][k], E11 * Jg[1][k]);
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) { acc[s] = fma(J[0][i], J[0][j], fma(J[1][i], J[1][j], acc[s])); ++s; }
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] = fma(J[0][i], ru, fma(J[1][i], rv, acc[21 + i]));
    acc[27] = fma(ru, ru, fma(rv, rv, acc[27]));
}

// Squared pixel reprojection error of one correspondence.
MQS_HD double reproj_sqerr(const double *P, const double *intr, double X, double Y, double Z, double u, double v)
{
    double pu, pv;
    cam::project(P, intr, X, Y, Z, pu, pv);
    const double du = pu - u, dv = pv - v;
    return fma(du, du, dv * dv);
}

// Solves (H + lambda diag(H)) delta = -g by Cholesky; false when not positive definite.
MQS_HD bool solve_step(const double *acc, double lambda, double *delta)
{
    double L[6][6];
    int s = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) { L[j][i] = acc[s] * (i == j ? 1.0 + lambda : 1.0); ++s; }   // lower triangle
    bool ok = true;
    double inv[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double dkk = L[k][k];
#pragma unroll
        for (int m = 0; m < k; ++m) dkk = fma(-L[k][m], L[k][m], dkk);
        ok = ok && (dkk > 0.0);
        const double r = mqs::rsqrt_d(dkk > 0.0 ? dkk : 1.0);      // (device: v_rsq_f64 + two Newton steps; the IEEE square root and division are ~25 dependent instructions per pivot)
        inv[k] = r;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            double v = L[i][k];
#pragma unroll
            for (int m = 0; m < k; ++m) v = fma(-L[i][m], L[k][m], v);
            L[i][k] = v * r;
        }
    }
    double yv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = -acc[21 + i];
#pragma unroll
        for (int m = 0; m < i; ++m) v = fma(-L[i][m], yv[m], v);
        yv[i] = v * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double v = yv[i];
#pragma unroll
        for (int m = i + 1; m < 6; ++m) v = fma(-L[m][i], delta[m], v);
        delta[i] = v * inv[i];
    }
    return ok;
}

MQS_HD void so3_exp(const double *w, double *E)
{
    const double th2 = fma(w[0], w[0], fma(w[1], w[1], w[2] * w[2]));
    double a, b;
    mqs::so3_exp_factors(th2, a, b);               // series for a step's small rotation (so3_math.h)
    const double K[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double k2 = fma(K[3 * i], K[j], fma(K[3 * i + 1], K[3 + j], K[3 * i + 2] * K[6 + j]));
            E[3 * i + j] = (i == j ? 1.0 : 0.0) + a * K[3 * i + j] + b * k2;
        }
}

// out = [Exp(w) R | Exp(w) t + v]
MQS_HD void retract(const double *P, const double *delta, double *out)
{
    double E[9];
    so3_exp(delta, E);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            out[4 * i + j] = fma(E[3 * i], P[j], fma(E[3 * i + 1], P[4 + j], E[3 * i + 2] * P[8 + j]));
        out[4 * i + 3] += delta[3 + i];
    }
}

struct LmResult { double sqerr; int iters; bool converged; };

// Levenberg-Marquardt on the pose.  eval(P, acc) must fill acc[kAcc] with the sums over the problem's
// correspondences at pose P (on the device: summed over the wavefront, identical in every lane).
template <class Eval>
MQS_HD LmResult lm_refine(Eval &eval, double *P, int max_iter, double eps)
{
    double acc[kAcc], trial[12], tacc[kAcc], delta[6];
    eval(P, acc);
    double lambda = 1e-3;
    LmResult res = {acc[27], 0, false};
    for (int it = 0; it < max_iter; ++it) {
        res.iters = it + 1;
        const bool pd = solve_step(acc, lambda, delta);
        bool accepted = false;
        if (pd) {
            // a step below the resolution asked for: at the minimum (also ends the lambda escalation that
            // follows the last successful step, where no trial can lower the cost any more)
            double d2 = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) d2 = fma(delta[k], delta[k], d2);
            const double t2 = fma(P[3], P[3], fma(P[7], P[7], P[11] * P[11]));
            if (d2 <= eps * eps * (1.0 + t2)) { res.converged = true; break; }
            retract(P, delta, trial);
            eval(trial, tacc);
            accepted = tacc[27] < acc[27];                 // NaN compares false: rejected
        }
        if (accepted) {
            const double gain = acc[27] - tacc[27];
#pragma unroll
            for (int k = 0; k < 12; ++k) P[k] = trial[k];
#pragma unroll
            for (int k = 0; k < kAcc; ++k) acc[k] = tacc[k];
            res.sqerr = acc[27];
            lambda = lambda > 1e-15 ? lambda * 0.1 : lambda;
            if (gain <= 1e-15 * acc[27]) { res.converged = true; break; }
        } else {
            lambda *= 10.0;
            if (lambda > 1e12) { res.converged = true; break; }     // no descent direction left: at the minimum
        }
    }
    return res;
}

// ---------------------------------------------------------------------------------------
// Direct linear transform start (>= 6 correspondences, undistorted normalised image points).
// World points are shifted to their centroid c and scaled by 1/sigma (mean distance), and the scale
// of the projection matrix is fixed by p34 = 1 (the centroid has positive depth), leaving 11 unknowns
//   p = [p11 p12 p13 p14  p21 p22 p23 p24  p31 p32 p33]
// with two equations per point:  p1.[X,1] - x p3.X = x,   p2.[X,1] - y p3.X = y.
// ---------------------------------------------------------------------------------------

// The normal equations A^T A p = A^T rhs have a fixed sparsity; their 51 distinct sums are accumulated
// per point (static indices: registers on the device):
//   [0..9]   sum Xh Xh^T  upper triangle, Xh = (X, Y, Z, 1)      (blocks (p1,p1) and (p2,p2))
//   [10..21] sum x Xh X^T  4x3 row-major                           (-block (p1,p3))
//   [22..33] sum y Xh X^T  4x3                                     (-block (p2,p3))
//   [34..39] sum (x^2 + y^2) X X^T  upper triangle                 (block (p3,p3))
//   [40..43] sum x Xh,  [44..47] sum y Xh,  [48..50] sum (x^2 + y^2) X   (right-hand side, last one negated)
constexpr int kDltAcc = 51;

MQS_HD void dlt_accumulate(double X, double Y, double Z, double x, double y, double *acc)
{
    const double Xh[4] = {X, Y, Z, 1.0};
    const double q = fma(x, x, y * y);
    int s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) { acc[s] = fma(Xh[i], Xh[j], acc[s]); ++s; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double t = Xh[i] * Xh[j];
            acc[10 + 3 * i + j] = fma(x, t, acc[10 + 3 * i + j]);
            acc[22 + 3 * i + j] = fma(y, t, acc[22 + 3 * i + j]);
        }
    s = 34;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) { acc[s] = fma(q * Xh[i], Xh[j], acc[s]); ++s; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[40 + i] = fma(x, Xh[i], acc[40 + i]);
        acc[44 + i] = fma(y, Xh[i], acc[44 + i]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[48 + i] = fma(q, Xh[i], acc[48 + i]);
}

// Expands the sums into the dense 11 x 11 system (A row-major, b).
MQS_HD void dlt_assemble(const double *acc, double *A, double *b)
{
#pragma unroll
    for (int k = 0; k < 121; ++k) A[k] = 0.0;
    int s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) {
            A[i * 11 + j] = A[j * 11 + i] = acc[s];
            A[(4 + i) * 11 + 4 + j] = A[(4 + j) * 11 + 4 + i] = acc[s];
            ++s;
        }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            A[i * 11 + 8 + j] = A[(8 + j) * 11 + i] = -acc[10 + 3 * i + j];
            A[(4 + i) * 11 + 8 + j] = A[(8 + j) * 11 + 4 + i] = -acc[22 + 3 * i + j];
        }
    s = 34;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) { A[(8 + i) * 11 + 8 + j] = A[(8 + j) * 11 + 8 + i] = acc[s]; ++s; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { b[i] = acc[40 + i]; b[4 + i] = acc[44 + i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) b[8 + i] = -acc[48 + i];
}

// In-place Cholesky solve of the n x n system A p = b (A row-major, symmetric; lower triangle used).
MQS_HD bool chol_solve_small(double *A, double *b, int n)
{
    bool ok = true;
    for (int k = 0; k < n; ++k) {
        double dkk = A[k * n + k];
        for (int m = 0; m < k; ++m) dkk -= A[k * n + m] * A[k * n + m];
        ok = ok && (dkk > 0.0);
        const double r = mqs::rsqrt_d(dkk > 0.0 ? dkk : 1.0);      // (device: v_rsq_f64 + two Newton steps; the IEEE square root and division are ~25 dependent instructions per pivot)
        A[k * n + k] = r;                                  // inverse diagonal
        for (int i = k + 1; i < n; ++i) {
            double v = A[i * n + k];
            for (int m = 0; m < k; ++m) v -= A[i * n + m] * A[k * n + m];
            A[i * n + k] = v * r;
        }
    }
    for (int i = 0; i < n; ++i) {
        double v = b[i];
        for (int m = 0; m < i; ++m) v -= A[i * n + m] * b[m];
        b[i] = v * A[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double v = b[i];
        for (int m = i + 1; m < n; ++m) v -= A[m * n + i] * b[m];
        b[i] = v * A[i * n + i];
    }
    return ok;
}

// The same solve for a size known at compile time, every loop unrolled: on the device the matrix then lives in registers (static
// indices), and every lane of a wavefront solves for itself -- the round-4 form handed the system to ONE lane that walked it in
// LDS with dynamic indices, a chain of dependent LDS round trips (most of a RANSAC hypothesis' 28 us).  Same operations in the
// same order as chol_solve_small: the same bits.
template <int N>
MQS_HD bool chol_solve_fixed(double (&A)[N * N], double (&b)[N])
{
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double dkk = A[k * N + k];
#pragma unroll
        for (int m = 0; m < k; ++m) dkk -= A[k * N + m] * A[k * N + m];
        ok = ok && (dkk > 0.0);
        const double r = mqs::rsqrt_d(dkk > 0.0 ? dkk : 1.0);      // (device: v_rsq_f64 + two Newton steps; the IEEE square root and division are ~25 dependent instructions per pivot)
        A[k * N + k] = r;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            double v = A[i * N + k];
#pragma unroll
            for (int m = 0; m < k; ++m) v -= A[i * N + m] * A[k * N + m];
            A[i * N + k] = v * r;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double v = b[i];
#pragma unroll
        for (int m = 0; m < i; ++m) v -= A[i * N + m] * b[m];
        b[i] = v * A[i * N + i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double v = b[i];
#pragma unroll
        for (int m = i + 1; m < N; ++m) v -= A[m * N + i] * b[m];
        b[i] = v * A[i * N + i];
    }
    return ok;
}

// Pose from the 11 DLT unknowns of the normalised problem: rotation by polar decomposition (Newton
// iteration R <- (R + R^-T) / 2), translation rescaled; false when the left 3x3 block is not a
// right-handed frame (degenerate sample).
MQS_HD bool pose_from_dlt(const double *p, const double *centroid, double sigma, double *P)
{
    double M[9] = {p[0], p[1], p[2], p[4], p[5], p[6], p[8], p[9], p[10]};
    const double n3sq = fma(M[6], M[6], fma(M[7], M[7], M[8] * M[8]));
    if (!(n3sq > 0.0)) return false;
    const double s = mqs::rsqrt_d(n3sq);
#pragma unroll
    for (int k = 0; k < 9; ++k) M[k] *= s;
    double tv[3] = {s * sigma * p[3], s * sigma * p[7], s * sigma};
    bool ok = true;
    for (int it = 0; it < 12; ++it) {
        const double c00 = fma(M[4], M[8], -M[5] * M[7]), c01 = fma(M[5], M[6], -M[3] * M[8]), c02 = fma(M[3], M[7], -M[4] * M[6]);
        const double det = fma(M[0], c00, fma(M[1], c01, M[2] * c02));
        if (!(det > 1e-12)) { ok = false; break; }
        const double id = 0.5 * mqs::rcp(det);
        // cofactor matrix = det * M^-T
        const double C[9] = {c00, c01, c02,
                             fma(M[2], M[7], -M[1] * M[8]), fma(M[0], M[8], -M[2] * M[6]), fma(M[1], M[6], -M[0] * M[7]),
                             fma(M[1], M[5], -M[2] * M[4]), fma(M[2], M[3], -M[0] * M[5]), fma(M[0], M[4], -M[1] * M[3])};
#pragma unroll
        for (int k = 0; k < 9; ++k) M[k] = fma(C[k], id, 0.5 * M[k]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        P[4 * i + 0] = M[3 * i]; P[4 * i + 1] = M[3 * i + 1]; P[4 * i + 2] = M[3 * i + 2];
        P[4 * i + 3] = tv[i] - fma(M[3 * i], centroid[0], fma(M[3 * i + 1], centroid[1], M[3 * i + 2] * centroid[2]));
    }
    return ok;
}

// ---------------------------------------------------------------------------------------
// Planar configurations (OpenCV: third singular value of the centred point covariance below 1e-3 of the
// second): the 3-D DLT is rank deficient there, the start comes from the plane-to-image homography instead.
// Plane coordinates (a, b) = (e1, e2) . (X - c) / sigma with e1, e2 the two leading principal axes; homography
// unknowns h = [h11 h12 h13 h21 h22 h23 h31 h32] (h33 = 1), two equations per point:
//   h1.(a,b,1) - x (h31 a + h32 b) = x,   h2.(a,b,1) - y (h31 a + h32 b) = y.
// ---------------------------------------------------------------------------------------

// Jacobi eigen-decomposition of the symmetric 3x3 S = (xx, xy, xz, yy, yz, zz): eigenvalues w[3] in DESCENDING
// order, eigenvectors as the ROWS of V (V[3 k + .] belongs to w[k]); V is a proper rotation.
MQS_HD void sym3_eigen(const double *S, double *w, double *V)
{
    double a[3][3] = {{S[0], S[1], S[2]}, {S[1], S[3], S[4]}, {S[2], S[4], S[5]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 16; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off <= 1e-300 || off <= 1e-17 * (fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]))) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) * mqs::rcp(2.0 * a[p][q]);
                // tan of the rotation angle: sgn / (|theta| + sqrt(theta^2 + 1)); beyond 1e100 the square root is |theta| to rounding
                const double q1 = fma(theta, theta, 1.0);
                const double t = fabs(theta) > 1e100 ? 0.5 * mqs::rcp(theta)
                                                     : (theta >= 0.0 ? 1.0 : -1.0) * mqs::rcp(fabs(theta) + q1 * mqs::rsqrt_d(q1));
                const double c = mqs::rsqrt_d(fma(t, t, 1.0)), sn = t * c;
                for (int k = 0; k < 3; ++k) {                       // A <- A G
                    const double kp = a[k][p], kq = a[k][q];
                    a[k][p] = c * kp - sn * kq; a[k][q] = sn * kp + c * kq;
                }
                for (int k = 0; k < 3; ++k) {                       // A <- G^T A
                    const double pk = a[p][k], qk = a[q][k];
                    a[p][k] = c * pk - sn * qk; a[q][k] = sn * pk + c * qk;
                }
                for (int k = 0; k < 3; ++k) {                       // columns of v rotate
                    const double kp = v[k][p], kq = v[k][q];
                    v[k][p] = c * kp - sn * kq; v[k][q] = sn * kp + c * kq;
                }
            }
    }
    // eigenvalues in descending order, the eigenvectors with them: three compare-and-swaps on values (static indices: registers on
    // the device; sorting an index array put `a` and `v` into scratch memory, a dependent round trip per comparison)
    double d[3] = {a[0][0], a[1][1], a[2][2]};
    double c[3][3] = {{v[0][0], v[1][0], v[2][0]}, {v[0][1], v[1][1], v[2][1]}, {v[0][2], v[1][2], v[2][2]}};
#define MQS_SWAP_IF_LESS(i, j)                                                                   \
    if (d[i] < d[j]) {                                                                           \
        const double td = d[i]; d[i] = d[j]; d[j] = td;                                          \
        for (int k = 0; k < 3; ++k) { const double tc = c[i][k]; c[i][k] = c[j][k]; c[j][k] = tc; } \
    }
    MQS_SWAP_IF_LESS(0, 1)
    MQS_SWAP_IF_LESS(1, 2)
    MQS_SWAP_IF_LESS(0, 1)
#undef MQS_SWAP_IF_LESS
    for (int k = 0; k < 3; ++k) {
        w[k] = d[k];
        for (int i = 0; i < 3; ++i) V[3 * k + i] = c[k][i];
    }
    // right-handed: third axis = first x second
    V[6] = V[1] * V[5] - V[2] * V[4];
    V[7] = V[2] * V[3] - V[0] * V[5];
    V[8] = V[0] * V[4] - V[1] * V[3];
}

// 29 distinct sums of the 8 x 8 normal equations:
//   [0..5]   sum q q^T upper triangle, q = (a, b, 1)            (blocks (h1,h1), (h2,h2))
//   [6..11]  sum x q (a, b)   3x2 row-major,  [12..17] sum y q (a, b)      (-blocks (h1,h3), (h2,h3))
//   [18..20] sum (x^2 + y^2) (a, b)(a, b)^T upper triangle               (block (h3,h3))
//   [21..23] sum x q,  [24..26] sum y q,  [27..28] sum (x^2 + y^2) (a, b)   (right-hand side, last negated)
constexpr int kHomAcc = 29;

MQS_HD void hom_accumulate(double a, double b, double x, double y, double *acc)
{
    const double q[3] = {a, b, 1.0};
    const double r = fma(x, x, y * y);
    int s = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) { acc[s] = fma(q[i], q[j], acc[s]); ++s; }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double t = q[i] * q[j];
            acc[6 + 2 * i + j] = fma(x, t, acc[6 + 2 * i + j]);
            acc[12 + 2 * i + j] = fma(y, t, acc[12 + 2 * i + j]);
        }
    acc[18] = fma(r * a, a, acc[18]); acc[19] = fma(r * a, b, acc[19]); acc[20] = fma(r * b, b, acc[20]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        acc[21 + i] = fma(x, q[i], acc[21 + i]);
        acc[24 + i] = fma(y, q[i], acc[24 + i]);
    }
    acc[27] = fma(r, a, acc[27]); acc[28] = fma(r, b, acc[28]);
}

MQS_HD void hom_assemble(const double *acc, double *A, double *rhs)
{
#pragma unroll
    for (int k = 0; k < 64; ++k) A[k] = 0.0;
    int s = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) {
            A[i * 8 + j] = A[j * 8 + i] = acc[s];
            A[(3 + i) * 8 + 3 + j] = A[(3 + j) * 8 + 3 + i] = acc[s];
            ++s;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            A[i * 8 + 6 + j] = A[(6 + j) * 8 + i] = -acc[6 + 2 * i + j];
            A[(3 + i) * 8 + 6 + j] = A[(6 + j) * 8 + 3 + i] = -acc[12 + 2 * i + j];
        }
    A[6 * 8 + 6] = acc[18]; A[6 * 8 + 7] = A[7 * 8 + 6] = acc[19]; A[7 * 8 + 7] = acc[20];
#pragma unroll
    for (int i = 0; i < 3; ++i) { rhs[i] = acc[21 + i]; rhs[3 + i] = acc[24 + i]; }
    rhs[6] = -acc[27]; rhs[7] = -acc[28];
}

// Pose from the homography of the normalised plane coordinates: H = lambda [sigma r1 | sigma r2 | t];
// E = principal axes as rows (e1, e2, n).
MQS_HD bool pose_from_homography(const double *h, const double *E, const double *centroid, double sigma, double *P)
{
    const double h1[3] = {h[0], h[3], h[6]}, h2[3] = {h[1], h[4], h[7]}, h3[3] = {h[2], h[5], 1.0};
    const double n1sq = h1[0] * h1[0] + h1[1] * h1[1] + h1[2] * h1[2], n2sq = h2[0] * h2[0] + h2[1] * h2[1] + h2[2] * h2[2];
    if (!(n1sq > 0.0) || !(n2sq > 0.0)) return false;
    const double i1 = mqs::rsqrt_d(n1sq), i2 = mqs::rsqrt_d(n2sq), n1 = n1sq * i1, n2 = n2sq * i2;      // (no division, no square root: Newton steps)
    double M[9];                                           // [r1 r2 r3] as columns
    for (int i = 0; i < 3; ++i) { M[3 * i] = h1[i] * i1; M[3 * i + 1] = h2[i] * i2; }
    M[2] = M[3] * M[7] - M[6] * M[4];                      // r3 = r1 x r2
    M[5] = M[6] * M[1] - M[0] * M[7];
    M[8] = M[0] * M[4] - M[3] * M[1];
    const double sc = 2.0 * sigma * mqs::rcp(n1 + n2);
    const double tp[3] = {h3[0] * sc, h3[1] * sc, h3[2] * sc};
    for (int it = 0; it < 12; ++it) {                      // polar decomposition (Newton)
        const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
        const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
        if (!(det > 1e-12)) return false;
        const double id = 0.5 * mqs::rcp(det);
        const double C[9] = {c00, c01, c02,
                             M[2] * M[7] - M[1] * M[8], M[0] * M[8] - M[2] * M[6], M[1] * M[6] - M[0] * M[7],
                             M[1] * M[5] - M[2] * M[4], M[2] * M[3] - M[0] * M[5], M[0] * M[4] - M[1] * M[3]};
        for (int k = 0; k < 9; ++k) M[k] = C[k] * id + 0.5 * M[k];
    }
    // X_cam = M E (X - c) + tp
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) P[4 * i + j] = M[3 * i] * E[j] + M[3 * i + 1] * E[3 + j] + M[3 * i + 2] * E[6 + j];
        P[4 * i + 3] = tp[i] - (P[4 * i] * centroid[0] + P[4 * i + 1] * centroid[1] + P[4 * i + 2] * centroid[2]);
    }
    return true;
}

}  // namespace pnp
}  // namespace mqs
