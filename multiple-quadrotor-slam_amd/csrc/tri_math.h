// Per-landmark triangulation arithmetic, shared by the gfx950 kernels in triangulate.hip.
// Everything is register-resident closed-form fp64: the reference solves each 2C x 3
// (resp. 3C x 4) system with a generic SVD (cvSolve DECOMP_SVD, triangulation.c:81,130;
// cv2.triangulatePoints); here the same least-squares / null-space problems are solved on the
// 3x3 (4x4) Gram matrix so that a landmark needs ~100 fp64 operations per solve instead of
// ~2000 and never leaves the VGPRs.
//
// The functions are MQS_HD so that tests/host_math.cpp can compile this header with g++ and
// check the arithmetic against the oracle on machines without a GPU (test-only; the
// product library has no CPU path).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MQS_HD __host__ __device__ __forceinline__
#define MQS_HD_MEMBER __host__ __device__ __forceinline__
#else
#define MQS_HD static inline
#define MQS_HD_MEMBER inline
#endif

// Scheduling fence between the per-camera sections of the BA arithmetic: stops the machine
// scheduler from interleaving all cameras' loads and temporaries (which multiplies the live
// register count by C and spills).
#if defined(__HIP_DEVICE_COMPILE__)
#define MQS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MQS_SCHED_FENCE() ((void)0)
#endif

namespace mqs {

// 1/d.  Device: v_rcp_f64 seed + two Newton steps (full double precision for normal inputs,
// no v_div_scale/v_div_fmas/v_div_fixup sequence); 0, inf and NaN behave like IEEE division
// for the uses below (a zero pivot fails the pivot test before its reciprocal is used).
MQS_HD double rcp(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
#else
    return 1.0 / d;
#endif
}

// 1/sqrt(d), d > 0.  Device: v_rsq_f64 seed + two Newton steps.
MQS_HD double rsqrt_d(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(d);
    const double hd = 0.5 * d;
    y = fma(y, fma(-hd * y, y, 0.5), y);
    y = fma(y, fma(-hd * y, y, 0.5), y);
    return y;
#else
    return 1.0 / sqrt(d);
#endif
}

// Returns 0 in a way the optimiser cannot see through: added to an LDS index inside a loop it
// keeps the loop-invariant camera-matrix reads IN the loop (re-read from LDS, ~free) instead of
// being hoisted into dozens of live VGPRs.
MQS_HD int opaque_zero()
{
    int z = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(z));
#endif
    return z;
}

constexpr double kEps = 2.220446049250313e-16;
// A pivot of the 3x3 LDL^T below kPivotRel * trace(G) switches to the eigen pseudo-inverse.
constexpr double kPivotRel = 1e-13;
// Eigenvalues of the Gram matrix <= kEigDropRel * lambda_max count as zero (minimum-norm
// solution, the role of the 2*eps*sum(w) singular-value cut of OpenCV's SVBkSb; a Gram matrix
// resolves singular values only down to ~sqrt(eps)*sigma_max).
constexpr double kEigDropRel = 64 * kEps;

struct Sym3 { double xx, xy, xz, yy, yz, zz; };
struct Vec3 { double x, y, z; };

// Row a = s*p2[0:3] - pk[0:3], rhs b = -(s*p2[3] - pk[3])   (triangulation.c:30-40);
// accumulates w2 * (a a^T) into G and w2 * (a b) into h.
MQS_HD void accum_row(Sym3 &G, Vec3 &h, double s, const double *pk, const double *p2, double w2)
{
    const double ax = fma(s, p2[0], -pk[0]);
    const double ay = fma(s, p2[1], -pk[1]);
    const double az = fma(s, p2[2], -pk[2]);
    const double b = -fma(s, p2[3], -pk[3]);
    const double wx = w2 * ax, wy = w2 * ay, wz = w2 * az;
    G.xx = fma(wx, ax, G.xx); G.xy = fma(wx, ay, G.xy); G.xz = fma(wx, az, G.xz);
    G.yy = fma(wy, ay, G.yy); G.yz = fma(wy, az, G.yz); G.zz = fma(wz, az, G.zz);
    h.x = fma(wx, b, h.x); h.y = fma(wy, b, h.y); h.z = fma(wz, b, h.z);
}

// Symmetric 3x3 Jacobi eigen-decomposition based minimum-norm solve (rare path).
MQS_HD Vec3 pinv_solve3(const Sym3 &G, const Vec3 &h)
{
    double a00 = G.xx, a01 = G.xy, a02 = G.xz, a11 = G.yy, a12 = G.yz, a22 = G.zz;
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
    for (int sweep = 0; sweep < 12; ++sweep) {
        bool rotated = false;
        // a rotation is skipped once |a_pq| <= eps*sqrt(a_pp*a_qq) (relative criterion for
        // positive semi-definite matrices); a sweep without rotations ends the iteration
#define MQS_ROT(app, aqq, apq, akp, akq, v0p, v0q, v1p, v1q, v2p, v2q)                         \
        if (fabs(apq) > kEps * sqrt(fabs(app * aqq))) {                                       \
            rotated = true;                                                                   \
            const double theta = (aqq - app) / (2.0 * apq);                                   \
            const double t = copysign(1.0, theta) / (fabs(theta) + sqrt(fma(theta, theta, 1.0))); \
            const double c = 1.0 / sqrt(fma(t, t, 1.0)), s = t * c;                           \
            app = fma(-t, apq, app); aqq = fma(t, apq, aqq); apq = 0.0;                       \
            { const double kp = akp, kq = akq; akp = c * kp - s * kq; akq = s * kp + c * kq; } \
            { const double p = v0p, q = v0q; v0p = c * p - s * q; v0q = s * p + c * q; }      \
            { const double p = v1p, q = v1q; v1p = c * p - s * q; v1q = s * p + c * q; }      \
            { const double p = v2p, q = v2q; v2p = c * p - s * q; v2q = s * p + c * q; }      \
        }
        MQS_ROT(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21)   // (p,q) = (0,1), k = 2
        MQS_ROT(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22)   // (0,2), k = 1
        MQS_ROT(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22)   // (1,2), k = 0
#undef MQS_ROT
        if (!rotated) break;
    }
    const double lmax = fmax(fmax(fabs(a00), fabs(a11)), fabs(a22));
    const double thr = kEigDropRel * lmax;
    const double i0 = (a00 > thr) ? 1.0 / a00 : 0.0;
    const double i1 = (a11 > thr) ? 1.0 / a11 : 0.0;
    const double i2 = (a22 > thr) ? 1.0 / a22 : 0.0;
    const double y0 = i0 * (v00 * h.x + v10 * h.y + v20 * h.z);
    const double y1 = i1 * (v01 * h.x + v11 * h.y + v21 * h.z);
    const double y2 = i2 * (v02 * h.x + v12 * h.y + v22 * h.z);
    Vec3 x;
    x.x = v00 * y0 + v01 * y1 + v02 * y2;
    x.y = v10 * y0 + v11 * y1 + v12 * y2;
    x.z = v20 * y0 + v21 * y1 + v22 * y2;
    return x;
}

// LDL^T factor of a symmetric positive (semi-)definite 3x3.  ok == false <=> a pivot fell
// below kPivotRel*trace (rank-deficient to working precision).
struct Ldlt3 { double i0, i1, i2, l10, l20, l21; bool ok; };

MQS_HD Ldlt3 ldlt3(const Sym3 &G)
{
    Ldlt3 f;
    const double thr = kPivotRel * (G.xx + G.yy + G.zz);
    const double d0 = G.xx;
    f.i0 = rcp(d0);
    f.l10 = G.xy * f.i0;
    f.l20 = G.xz * f.i0;
    const double d1 = fma(-f.l10, G.xy, G.yy);
    f.i1 = rcp(d1);
    const double t21 = fma(-f.l20, G.xy, G.yz);
    f.l21 = t21 * f.i1;
    const double d2 = fma(-f.l21, t21, fma(-f.l20, G.xz, G.zz));
    f.i2 = rcp(d2);
    f.ok = (d0 > thr) && (d1 > thr) && (d2 > thr);      // false for NaN as well
    return f;
}

MQS_HD Vec3 ldlt3_solve(const Ldlt3 &f, const Vec3 &h)
{
    const double y0 = h.x;
    const double y1 = fma(-f.l10, y0, h.y);
    const double y2 = fma(-f.l21, y1, fma(-f.l20, y0, h.z));
    Vec3 x;
    x.z = y2 * f.i2;
    x.y = fma(-f.l21, x.z, y1 * f.i1);
    x.x = fma(-f.l20, x.z, fma(-f.l10, x.y, y0 * f.i0));
    return x;
}

// Minimum-norm least-squares solve of G x = h: LDL^T, eigen pseudo-inverse when deficient.
MQS_HD Vec3 solve_normal3(const Sym3 &G, const Vec3 &h, Ldlt3 &f)
{
    f = ldlt3(G);
    if (f.ok) return ldlt3_solve(f, h);
    return pinv_solve3(G, h);
}

// ---------------------------------------------------------------------------------------
// 4x4 symmetric eigenproblem for the homogeneous DLT (T3): eigenvector of the smallest
// eigenvalue of N = A^T A by cyclic Jacobi.  n[] holds the upper triangle
// {00,01,02,03,11,12,13,22,23,33}.  Returns X (4).
// ---------------------------------------------------------------------------------------
MQS_HD void smallest_eigvec4(const double n[10], double X[4])
{
    double a00 = n[0], a01 = n[1], a02 = n[2], a03 = n[3], a11 = n[4], a12 = n[5], a13 = n[6],
           a22 = n[7], a23 = n[8], a33 = n[9];
    double v00 = 1, v01 = 0, v02 = 0, v03 = 0, v10 = 0, v11 = 1, v12 = 0, v13 = 0,
           v20 = 0, v21 = 0, v22 = 1, v23 = 0, v30 = 0, v31 = 0, v32 = 0, v33 = 1;
    for (int sweep = 0; sweep < 16; ++sweep) {
        bool rotated = false;
        // rotation in the (p,q) plane; (k,l) are the two other indices
#define MQS_ROT4(app, aqq, apq, akp, akq, alp, alq, v0p, v0q, v1p, v1q, v2p, v2q, v3p, v3q)    \
        if (fabs(apq) > kEps * sqrt(fabs(app * aqq))) {                                       \
            rotated = true;                                                                   \
            const double theta = (aqq - app) / (2.0 * apq);                                   \
            const double t = copysign(1.0, theta) / (fabs(theta) + sqrt(fma(theta, theta, 1.0))); \
            const double c = 1.0 / sqrt(fma(t, t, 1.0)), s = t * c;                           \
            app = fma(-t, apq, app); aqq = fma(t, apq, aqq); apq = 0.0;                       \
            { const double p = akp, q = akq; akp = c * p - s * q; akq = s * p + c * q; }      \
            { const double p = alp, q = alq; alp = c * p - s * q; alq = s * p + c * q; }      \
            { const double p = v0p, q = v0q; v0p = c * p - s * q; v0q = s * p + c * q; }      \
            { const double p = v1p, q = v1q; v1p = c * p - s * q; v1q = s * p + c * q; }      \
            { const double p = v2p, q = v2q; v2p = c * p - s * q; v2q = s * p + c * q; }      \
            { const double p = v3p, q = v3q; v3p = c * p - s * q; v3q = s * p + c * q; }      \
        }
        MQS_ROT4(a00, a11, a01, a02, a12, a03, a13, v00, v01, v10, v11, v20, v21, v30, v31)  // (0,1); k=2,l=3
        MQS_ROT4(a00, a22, a02, a01, a12, a03, a23, v00, v02, v10, v12, v20, v22, v30, v32)  // (0,2); k=1,l=3
        MQS_ROT4(a00, a33, a03, a01, a13, a02, a23, v00, v03, v10, v13, v20, v23, v30, v33)  // (0,3); k=1,l=2
        MQS_ROT4(a11, a22, a12, a01, a02, a13, a23, v01, v02, v11, v12, v21, v22, v31, v32)  // (1,2); k=0,l=3
        MQS_ROT4(a11, a33, a13, a01, a03, a12, a23, v01, v03, v11, v13, v21, v23, v31, v33)  // (1,3); k=0,l=2
        MQS_ROT4(a22, a33, a23, a02, a03, a12, a13, v02, v03, v12, v13, v22, v23, v32, v33)  // (2,3); k=0,l=1
#undef MQS_ROT4
        if (!rotated) break;
    }
    // column of V belonging to the smallest eigenvalue (ties: the later column, like the
    // last entry of a descending sort)
    double lam = a00;
    X[0] = v00; X[1] = v10; X[2] = v20; X[3] = v30;
    if (a11 <= lam) { lam = a11; X[0] = v01; X[1] = v11; X[2] = v21; X[3] = v31; }
    if (a22 <= lam) { lam = a22; X[0] = v02; X[1] = v12; X[2] = v22; X[3] = v32; }
    if (a33 <= lam) { lam = a33; X[0] = v03; X[1] = v13; X[2] = v23; X[3] = v33; }
}

// Fast path for the same eigenvector: LDL^T of N followed by inverse iteration
// (v <- N^-1 v).  N is positive semi-definite and, for a triangulation problem, nearly singular
// along the sought direction, which is exactly when inverse iteration converges in a few steps
// (rate rho = lambda_4 / lambda_3).  With four cameras rho is ~1e-3 and 4-6 steps reach 1e-11; with
// two or three the weakly constrained depth direction gives rho up to 0.1-0.9 for a percent of the
// landmarks, and since a wavefront runs as long as its slowest lane, those set the pace (and the
// ones past the step limit sent their whole wave through the Jacobi fallback).  So after
// kInvIterPlain plain steps a lane that has not settled shifts: mu = Rayleigh quotient of its
// iterate (>= lambda_4, off by rho^(2k) lambda_3), N - mu (1 - 2^-10) I is factored again and the
// rate becomes ~1e-3 rho or |mu - lambda_4| / lambda_3, whichever is larger -- two or three more steps.
// Returns false when the iterates have not settled to 1e-11 (clustered smallest eigenvalues:
// degenerate geometry) or a leading pivot is not positive; the caller then falls back to the Jacobi
// iteration above.
constexpr int kInvIterPlain = 5;
constexpr int kInvIterShifted = 8;
constexpr int kInvIterMax = kInvIterPlain + kInvIterShifted;

struct Ldlt4 {
    double l10, l20, l30, l21, l31, l32;    // unit lower-triangular factor
    double i0, i1, i2, i3;                  // reciprocal pivots
    bool ok;                                // leading pivots positive
};

// LDL^T of N - shift * I (no pivoting; the last pivot may have either sign and is kept away from zero)
MQS_HD Ldlt4 ldlt4_shifted(const double n[10], double shift, double tiny)
{
    Ldlt4 f;
    const double d0 = n[0] - shift;
    f.i0 = rcp(d0);
    f.l10 = n[1] * f.i0; f.l20 = n[2] * f.i0; f.l30 = n[3] * f.i0;
    const double d1 = fma(-f.l10, n[1], n[4] - shift);
    f.i1 = rcp(d1);
    const double t21 = fma(-f.l20, n[1], n[5]);
    const double t31 = fma(-f.l30, n[1], n[6]);
    f.l21 = t21 * f.i1; f.l31 = t31 * f.i1;
    const double d2 = fma(-f.l21, t21, fma(-f.l20, n[2], n[7] - shift));
    f.i2 = rcp(d2);
    const double t32 = fma(-f.l31, t21, fma(-f.l30, n[2], n[8]));
    f.l32 = t32 * f.i2;
    double d3 = fma(-f.l32, t32, fma(-f.l31, t31, fma(-f.l30, n[3], n[9] - shift)));
    f.ok = (d0 > 0.0) && (d1 > 0.0) && (d2 > 0.0);
    if (!(fabs(d3) > tiny)) d3 = tiny;                   // exact data: lambda_4 == 0 to rounding
    f.i3 = rcp(d3);
    return f;
}

// One inverse-iteration step on the power-of-two normalised iterate; returns true when the direction has settled.
MQS_HD bool invit4_step(const Ldlt4 &f, double &v0, double &v1, double &v2, double &v3)
{
    // normalise by an exact power of two (direction only matters), remember the iterate
    const double m = fmax(fmax(fabs(v0), fabs(v1)), fmax(fabs(v2), fabs(v3)));
    const double sc = ldexp(1.0, -ilogb(m));
    v0 *= sc; v1 *= sc; v2 *= sc; v3 *= sc;
    const double p0 = v0, p1 = v1, p2 = v2, p3 = v3;
    // solve L y = v, z = D^-1 y, L^T w = z
    const double y0 = v0;
    const double y1 = fma(-f.l10, y0, v1);
    const double y2 = fma(-f.l21, y1, fma(-f.l20, y0, v2));
    const double y3 = fma(-f.l32, y2, fma(-f.l31, y1, fma(-f.l30, y0, v3)));
    v3 = y3 * f.i3;
    v2 = fma(-f.l32, v3, y2 * f.i2);
    v1 = fma(-f.l31, v3, fma(-f.l21, v2, y1 * f.i1));
    v0 = fma(-f.l30, v3, fma(-f.l20, v2, fma(-f.l10, v1, y0 * f.i0)));
    // change of direction between the (scaled) previous iterate p and the new one v:
    // |v - (v.p / p.p) p|_inf relative to |v|_inf
    const double pp = fma(p0, p0, fma(p1, p1, fma(p2, p2, p3 * p3)));
    const double vp = fma(v0, p0, fma(v1, p1, fma(v2, p2, v3 * p3)));
    const double a = vp * rcp(pp);
    const double e0 = fabs(fma(-a, p0, v0)), e1 = fabs(fma(-a, p1, v1));
    const double e2 = fabs(fma(-a, p2, v2)), e3 = fabs(fma(-a, p3, v3));
    const double err = fmax(fmax(e0, e1), fmax(e2, e3));
    const double mag = fmax(fmax(fabs(v0), fabs(v1)), fmax(fabs(v2), fabs(v3)));
    return err <= 1e-11 * mag;
}

MQS_HD bool smallest_eigvec4_invit(const double n[10], double X[4])
{
    const double tr = n[0] + n[4] + n[7] + n[9];
    const double tiny = 1e-30 * tr;
    Ldlt4 f = ldlt4_shifted(n, 0.0, tiny);
    if (!f.ok) return false;
    // first iterate: N^-1 e_3 = L^-T D^-1 L^-1 e_3 = L^-T (e_3 / d3)  -> direction L^-T e_3
    double v3 = 1.0;
    double v2 = -f.l32;
    double v1 = -fma(f.l21, v2, f.l31);
    double v0 = -fma(f.l10, v1, fma(f.l20, v2, f.l30));
    bool done = false;
    for (int it = 0; it < kInvIterPlain && !done; ++it) done = invit4_step(f, v0, v1, v2, v3);
    if (!done) {
        // Rayleigh quotient of the current iterate, then the shifted factorisation
        const double w0 = fma(n[0], v0, fma(n[1], v1, fma(n[2], v2, n[3] * v3)));
        const double w1 = fma(n[1], v0, fma(n[4], v1, fma(n[5], v2, n[6] * v3)));
        const double w2 = fma(n[2], v0, fma(n[5], v1, fma(n[7], v2, n[8] * v3)));
        const double w3 = fma(n[3], v0, fma(n[6], v1, fma(n[8], v2, n[9] * v3)));
        const double vv = fma(v0, v0, fma(v1, v1, fma(v2, v2, v3 * v3)));
        const double mu = fma(v0, w0, fma(v1, w1, fma(v2, w2, v3 * w3))) * rcp(vv);
        const Ldlt4 g = ldlt4_shifted(n, mu * (1.0 - 0x1p-10), tiny);
        if (!g.ok) return false;
        for (int it = 0; it < kInvIterShifted && !done; ++it) done = invit4_step(g, v0, v1, v2, v3);
    }
    X[0] = v0; X[1] = v1; X[2] = v2; X[3] = v3;
    return done;
}

// Adds the three rows of cv2.triangulatePoints for one camera to the 4x4 Gram matrix:
// x*P[2]-P[0], y*P[2]-P[1], x*P[1]-y*P[0].  p0,p1,p2: rows of P (4 each).
MQS_HD void accum_eigen_rows(double n[10], double x, double y, const double *p0, const double *p1,
                             const double *p2)
{
    double r[4];
#define MQS_ACC()                                                                              \
    n[0] = fma(r[0], r[0], n[0]); n[1] = fma(r[0], r[1], n[1]); n[2] = fma(r[0], r[2], n[2]);  \
    n[3] = fma(r[0], r[3], n[3]); n[4] = fma(r[1], r[1], n[4]); n[5] = fma(r[1], r[2], n[5]);  \
    n[6] = fma(r[1], r[3], n[6]); n[7] = fma(r[2], r[2], n[7]); n[8] = fma(r[2], r[3], n[8]);  \
    n[9] = fma(r[3], r[3], n[9]);
    for (int l = 0; l < 4; ++l) r[l] = fma(x, p2[l], -p0[l]);
    MQS_ACC()
    for (int l = 0; l < 4; ++l) r[l] = fma(y, p2[l], -p1[l]);
    MQS_ACC()
    for (int l = 0; l < 4; ++l) r[l] = fma(x, p1[l], -(y * p0[l]));
    MQS_ACC()
#undef MQS_ACC
}

// ---------------------------------------------------------------------------------------
// Whole-landmark drivers.  uv[c] = (x, y) of camera c; P = [C][12].
// ---------------------------------------------------------------------------------------

// One corrected-semi-normal-equations refinement step: x += G^-1 A^T W (b - A x), with the
// residual formed from the rows themselves (restores ~eps*cond(A) accuracy that the Gram
// matrix alone would square away).
template <int C>
MQS_HD Vec3 refine(const Vec3 &x0, const double (*uv)[2], const double *P, const double *w2,
                   const Sym3 &G, const Ldlt3 &f)
{
    if (!f.ok) return x0;
    Vec3 r = {0, 0, 0};
    const double *Pl = P + opaque_zero();
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double *p0 = Pl + 12 * c, *p1 = p0 + 4, *p2 = p0 + 8;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double *pk = k ? p1 : p0;
            const double s = uv[c][k];
            const double ax = fma(s, p2[0], -pk[0]);
            const double ay = fma(s, p2[1], -pk[1]);
            const double az = fma(s, p2[2], -pk[2]);
            const double b = -fma(s, p2[3], -pk[3]);
            const double res = w2[c] * (b - fma(ax, x0.x, fma(ay, x0.y, az * x0.z)));
            r.x = fma(ax, res, r.x); r.y = fma(ay, res, r.y); r.z = fma(az, res, r.z);
        }
    }
    (void)G;
    const Vec3 dx = ldlt3_solve(f, r);
    Vec3 x = {x0.x + dx.x, x0.y + dx.y, x0.z + dx.z};
    return x;
}

template <int C>
MQS_HD Vec3 linear_ls_point(const double (*uv)[2], const double *P)
{
    Sym3 G = {0, 0, 0, 0, 0, 0};
    Vec3 h = {0, 0, 0};
    double w2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double *p0 = P + 12 * c, *p1 = p0 + 4, *p2 = p0 + 8;
        accum_row(G, h, uv[c][0], p0, p2, 1.0);
        accum_row(G, h, uv[c][1], p1, p2, 1.0);
        w2[c] = 1.0;
    }
    Ldlt3 f;
    const Vec3 x = solve_normal3(G, h, f);
    return refine<C>(x, uv, P, w2, G, f);
}

// T2 state handed from the iteration to the final refinement step.
template <int C>
struct IterResult {
    Vec3 x;
    double w2[C];      // squared cumulative row weights of the LAST solve
    Ldlt3 f;           // its factorisation
    int32_t status;    // triangulation.c:154-159 generalised to C cameras
    bool solved;       // at least one solve happened (max_iter > 0)
};

// T2 iteration.  Does not keep `uv` alive past the Gram set-up, so that a kernel can re-read
// the observations for refine<C>() instead of holding them in registers through the loop.
// Hook for callers that also want the FIRST solve of the iteration -- with unit weights it is the linear-LS system of
// triangulation.c:65-83, so linear-LS comes for free from an iterative-LS pass (the fused kernel parks it in LDS: holding
// it in registers through the loop would cost the loop its occupancy).
struct NoFirstSolve {
    MQS_HD_MEMBER void operator()(const Vec3 &, const Ldlt3 &) const {}
};

template <int C, class First = NoFirstSolve>
MQS_HD void iterative_ls_core(const double (*uv)[2], const double *P, const double *Pdepth, double tol,
                              int max_iter, IterResult<C> &out, const First &first = First())
{
    // P      : camera matrices for the Gram set-up (the kernels pass their LDS copy)
    // Pdepth : the same matrices for the depth rows read in every iteration (the kernels pass
    //          the read-only global pointer: wave-uniform address -> scalar loads, the rows live
    //          in SGPRs instead of being re-read from LDS or pinned in VGPRs)
    // Per-camera Gram pieces: the re-weighting of triangulation.c:143-146 multiplies camera
    // c's two rows (and b entries) by 1/d_c, i.e. its Gram contribution by 1/d_c^2, so the
    // pieces are built once and only re-combined per iteration.
    Sym3 M[C];
    Vec3 hc[C];
    double w2[C], d[C], dn[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double *p0 = P + 12 * c, *p1 = p0 + 4, *p2 = p0 + 8;
        M[c] = Sym3{0, 0, 0, 0, 0, 0};
        hc[c] = Vec3{0, 0, 0};
        accum_row(M[c], hc[c], uv[c][0], p0, p2, 1.0);
        accum_row(M[c], hc[c], uv[c][1], p1, p2, 1.0);
        w2[c] = 1.0; d[c] = 1.0; dn[c] = 1.0;             // triangulation.c:122
    }
    Vec3 x = {0, 0, 0};
    Sym3 G;
    Ldlt3 f = {0, 0, 0, 0, 0, 0, false};
    int i = 0;
    for (; i < max_iter; ++i) {
        if (i > 0) {
            // triangulation.c:143-150: camera c's rows are scaled by 1/d_c.  Only the RATIOS of
            // the weights matter to the solution, so 1/d_c is replaced by prod_{j != c} d_j
            // (= 1/d_c times the common factor prod_j d_j): no reciprocals.
            double pre[C], suf[C];
            pre[0] = 1.0;
#pragma unroll
            for (int c = 1; c < C; ++c) pre[c] = pre[c - 1] * dn[c - 1];
            suf[C - 1] = 1.0;
#pragma unroll
            for (int c = C - 2; c >= 0; --c) suf[c] = suf[c + 1] * dn[c + 1];
            // common factor removed again by an exact power of two (camera 0's scale -> [1,2))
            const double kappa = ldexp(1.0, -ilogb(suf[0]));
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double sc = (pre[c] * suf[c]) * kappa;
                w2[c] *= sc * sc;
                d[c] = dn[c];
            }
        }
        G = Sym3{0, 0, 0, 0, 0, 0};
        Vec3 h = {0, 0, 0};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            G.xx = fma(w2[c], M[c].xx, G.xx); G.xy = fma(w2[c], M[c].xy, G.xy);
            G.xz = fma(w2[c], M[c].xz, G.xz); G.yy = fma(w2[c], M[c].yy, G.yy);
            G.yz = fma(w2[c], M[c].yz, G.yz); G.zz = fma(w2[c], M[c].zz, G.zz);
            h.x = fma(w2[c], hc[c].x, h.x); h.y = fma(w2[c], hc[c].y, h.y); h.z = fma(w2[c], hc[c].z, h.z);
        }
        x = solve_normal3(G, h, f);                        // :130
        if (i == 0) first(x, f);
        bool conv = true, zero = false;
#pragma unroll
        for (int c = 0; c < C; ++c) {                      // :133-134
            const double *p2 = Pdepth + 12 * c + 8;
            dn[c] = fma(p2[0], x.x, fma(p2[1], x.y, fma(p2[2], x.z, p2[3])));
            conv = conv && (fabs(dn[c] - d[c]) <= tol);
            zero = zero || (dn[c] == 0.0);
        }
        if (conv || zero) break;                           // :137-140
    }
    bool front = true;
#pragma unroll
    for (int c = 0; c < C; ++c) front = front && (dn[c] > 0.0);
    int32_t s = ((i < max_iter) && front) ? 1 : 0;         // :154-155
#pragma unroll
    for (int c = 0; c < C; ++c)
        if (dn[c] <= 0.0) s -= (1 << c);                   // :156-159
    out.status = s;
    out.x = x;
    out.f = f;
    out.solved = max_iter > 0;
#pragma unroll
    for (int c = 0; c < C; ++c) out.w2[c] = w2[c];
}

// T2.  Returns x and the status code of triangulation.c:154-159 generalised to C cameras.
template <int C>
MQS_HD Vec3 iterative_ls_point(const double (*uv)[2], const double *P, double tol, int max_iter,
                               int32_t &status)
{
    IterResult<C> r;
    iterative_ls_core<C>(uv, P, P, tol, max_iter, r);
    status = r.status;
    if (!r.solved) return r.x;
    const Sym3 unused = {0, 0, 0, 0, 0, 0};
    return refine<C>(r.x, uv, P, r.w2, unused, r.f);
}

template <int C>
MQS_HD Vec3 linear_eigen_point(const double (*uv)[2], const double *P, double max_coord, bool &ok)
{
    double n[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double *p0 = P + 12 * c;
        accum_eigen_rows(n, uv[c][0], uv[c][1], p0, p0 + 4, p0 + 8);
    }
    double X[4];
    if (!smallest_eigvec4_invit(n, X)) smallest_eigvec4(n, X);
    Vec3 x = {X[0] / X[3], X[1] / X[3], X[2] / X[3]};      // triangulation.py:22
    ok = (fabs(x.x) <= max_coord) && (fabs(x.y) <= max_coord) && (fabs(x.z) <= max_coord);   // :23
    return x;
}

}  // namespace mqs
