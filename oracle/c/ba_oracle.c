/*
 * ORACLE (test infrastructure, not product code).
 *
 * Plain-C restatement of the bundle-adjustment arithmetic the reference delegates to GTSAM 3.2.1
 * (not vendored under /root/reference; unbuildable here -> per-iteration parity UNPINNED):
 * GenericProjectionFactor<Pose3, Point3, Cal3DS2> residual + Jacobians, PriorFactor<Point3>,
 * landmark elimination (Schur complement) and back-substitution.  Graph structure:
 * /root/reference/Work/SLAM/tools/bundle_adjustment/bundle_adjust.cpp:268-298; optimiser call
 * :323-324.  Same conventions as oracle/ba_np.py (which it is checked against in
 * tests/test_ba_oracle.py); written the straightforward way -- explicit 2x6 / 2x3 Jacobians per
 * factor, dense 6C x 3 / 6C x 6C blocks per landmark -- NOT the factored form the HIP kernel uses.
 * Timed by bench.py as the BA CPU baseline ("port", OpenMP over landmarks).
 *
 * Layouts: poses [C][12] (R row-major camera-to-world, t), calib [C][9], sigma [C],
 * points [N][3], obs [C][N][2], mask [C][N] u8 or NULL, prior_w [N] / prior_xyz [N][3] or NULL.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BA_MAX_CAMS 8
#define BA_MAX_N6 (6 * BA_MAX_CAMS)

/* whitened residual e[2], Jp[2][6], Jl[2][3]; returns 1 when the point is in front of the camera */
static int ba_factor(const double *pose, const double *K, double sigma, const double *p, const double *uv,
                     double *e, double Jp[2][6], double Jl[2][3])
{
    const double *R = pose, *t = pose + 9;
    const double d[3] = {p[0] - t[0], p[1] - t[1], p[2] - t[2]};
    double q[3];
    int i, j, k;
    for (i = 0; i < 3; i++) q[i] = R[0 + i] * d[0] + R[3 + i] * d[1] + R[6 + i] * d[2];     /* R^T d */
    memset(Jp, 0, sizeof(double) * 12);
    memset(Jl, 0, sizeof(double) * 6);
    if (!(q[2] > 0)) { e[0] = e[1] = 2.0 * K[0] / sigma; return 0; }
    {
        const double fx = K[0], fy = K[1], s = K[2], u0 = K[3], v0 = K[4], k1 = K[5], k2 = K[6], p1 = K[7], p2 = K[8];
        const double x = q[0] / q[2], y = q[1] / q[2], r2 = x * x + y * y;
        const double g = 1 + k1 * r2 + k2 * r2 * r2, dg = k1 + 2 * k2 * r2;
        const double xd = g * x + 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
        const double yd = g * y + 2 * p2 * x * y + p1 * (r2 + 2 * y * y);
        const double Dd[2][2] = {{g + 2 * x * x * dg + 2 * p1 * y + 6 * p2 * x, 2 * x * y * dg + 2 * p1 * x + 2 * p2 * y},
                                 {2 * x * y * dg + 2 * p2 * y + 2 * p1 * x, g + 2 * y * y * dg + 2 * p2 * x + 6 * p1 * y}};
        const double Kk[2][2] = {{fx, s}, {0, fy}};
        const double Dp[2][3] = {{1 / q[2], 0, -x / q[2]}, {0, 1 / q[2], -y / q[2]}};
        double KD[2][2], D[2][3];
        const double G[3][6] = {{0, -q[2], q[1], -1, 0, 0}, {q[2], 0, -q[0], 0, -1, 0}, {-q[1], q[0], 0, 0, 0, -1}};
        e[0] = (fx * xd + s * yd + u0 - uv[0]) / sigma;
        e[1] = (fy * yd + v0 - uv[1]) / sigma;
        for (i = 0; i < 2; i++) for (j = 0; j < 2; j++) KD[i][j] = Kk[i][0] * Dd[0][j] + Kk[i][1] * Dd[1][j];
        for (i = 0; i < 2; i++) for (j = 0; j < 3; j++) D[i][j] = (KD[i][0] * Dp[0][j] + KD[i][1] * Dp[1][j]) / sigma;
        for (i = 0; i < 2; i++) for (j = 0; j < 6; j++) for (k = 0; k < 3; k++) Jp[i][j] += D[i][k] * G[k][j];
        for (i = 0; i < 2; i++) for (j = 0; j < 3; j++) for (k = 0; k < 3; k++) Jl[i][j] += D[i][k] * R[3 * j + k];  /* D R^T */
    }
    return 1;
}

static int inv3_spd(const double H[3][3], double Hi[3][3])
{
    const double c00 = H[1][1] * H[2][2] - H[1][2] * H[2][1], c01 = H[1][2] * H[2][0] - H[1][0] * H[2][2],
                 c02 = H[1][0] * H[2][1] - H[1][1] * H[2][0];
    const double det = H[0][0] * c00 + H[0][1] * c01 + H[0][2] * c02;
    const double tr = H[0][0] + H[1][1] + H[2][2];
    /* positive definiteness by leading minors, relative threshold as oracle/ba_np.py */
    const double m2 = H[0][0] * H[1][1] - H[0][1] * H[1][0];
    if (!(H[0][0] > 1e-13 * tr) || !(m2 > 1e-13 * tr * H[0][0]) || !(det > 1e-13 * tr * m2)) return 0;
    Hi[0][0] = c00 / det; Hi[0][1] = (H[0][2] * H[2][1] - H[0][1] * H[2][2]) / det; Hi[0][2] = (H[0][1] * H[1][2] - H[0][2] * H[1][1]) / det;
    Hi[1][0] = c01 / det; Hi[1][1] = (H[0][0] * H[2][2] - H[0][2] * H[2][0]) / det; Hi[1][2] = (H[0][2] * H[1][0] - H[0][0] * H[1][2]) / det;
    Hi[2][0] = c02 / det; Hi[2][1] = (H[0][1] * H[2][0] - H[0][0] * H[2][1]) / det; Hi[2][2] = m2 / det;
    return 1;
}

/* per-landmark blocks; returns cost contribution, fills nvalid */
static double ba_landmark(const double *poses, const double *calib, const double *sigma, int C, const double *p,
                          const double *obs, const uint8_t *mask, int64_t N, int64_t i, double pw, const double *pxyz,
                          double lambda, double Hi[3][3], double gl[3], double Hpl[BA_MAX_N6][3],
                          double Hpp[BA_MAX_CAMS][6][6], double gp[BA_MAX_N6], int *nvalid, int *constrained)
{
    double H[3][3] = {{0}}, cost = 0;
    int c, a, b, k, ok = 0;
    memset(gl, 0, sizeof(double) * 3);
    memset(Hpl, 0, sizeof(double) * BA_MAX_N6 * 3);
    memset(Hpp, 0, sizeof(double) * BA_MAX_CAMS * 36);
    memset(gp, 0, sizeof(double) * BA_MAX_N6);
    *nvalid = 0;
    for (c = 0; c < C; c++) {
        double e[2], Jp[2][6], Jl[2][3];
        if (mask && !mask[(int64_t)c * N + i]) continue;
        const int valid = ba_factor(poses + 12 * c, calib + 9 * c, sigma[c], p, obs + ((int64_t)c * N + i) * 2, e, Jp, Jl);
        cost += 0.5 * (e[0] * e[0] + e[1] * e[1]);
        *nvalid += valid;
        ok |= valid;
        for (k = 0; k < 2; k++) {
            for (a = 0; a < 3; a++) { for (b = 0; b < 3; b++) H[a][b] += Jl[k][a] * Jl[k][b]; gl[a] -= Jl[k][a] * e[k]; }
            for (a = 0; a < 6; a++) {
                for (b = 0; b < 3; b++) Hpl[6 * c + a][b] += Jp[k][a] * Jl[k][b];
                for (b = 0; b < 6; b++) Hpp[c][a][b] += Jp[k][a] * Jp[k][b];
                gp[6 * c + a] -= Jp[k][a] * e[k];
            }
        }
    }
    if (pw > 0) {
        const double d[3] = {p[0] - pxyz[0], p[1] - pxyz[1], p[2] - pxyz[2]};
        for (a = 0; a < 3; a++) { H[a][a] += pw; gl[a] -= pw * d[a]; }
        cost += 0.5 * pw * (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        ok = 1;
    }
    for (a = 0; a < 3; a++) H[a][a] += lambda * H[a][a];
    *constrained = ok && inv3_spd(H, Hi);
    if (!*constrained) memset(Hi, 0, sizeof(double) * 9);
    return cost;
}

/* out[(6C)^2 + 6C + 2] = {S, g, cost, count} */
int orc_ba_linearize(const double *poses, const double *calib, const double *sigma, int C, const double *points,
                     const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz, int64_t N,
                     double lambda, double *out, int use_omp)
{
    const int n6 = 6 * C, nout = n6 * n6 + n6 + 2;
    int64_t i;
    int k;
    if (C < 1 || C > BA_MAX_CAMS) return -1;
    for (k = 0; k < nout; k++) out[k] = 0;
    #pragma omp parallel if (use_omp)
    {
        double *loc = (double *)calloc(nout, sizeof(double));
        #pragma omp for schedule(static)
        for (i = 0; i < N; i++) {
            double Hi[3][3], gl[3], Hpl[BA_MAX_N6][3], Hpp[BA_MAX_CAMS][6][6], gp[BA_MAX_N6], Y[BA_MAX_N6][3];
            int nv, con, a, b, c;
            loc[n6 * n6 + n6] += ba_landmark(poses, calib, sigma, C, points + 3 * i, obs, mask, N, i,
                                             prior_w ? prior_w[i] : 0.0, prior_xyz ? prior_xyz + 3 * i : points, lambda,
                                             Hi, gl, Hpl, Hpp, gp, &nv, &con);
            loc[n6 * n6 + n6 + 1] += nv;
            for (a = 0; a < n6; a++) for (b = 0; b < 3; b++) Y[a][b] = Hpl[a][0] * Hi[0][b] + Hpl[a][1] * Hi[1][b] + Hpl[a][2] * Hi[2][b];
            for (c = 0; c < C; c++) for (a = 0; a < 6; a++) for (b = 0; b < 6; b++) loc[(6 * c + a) * n6 + 6 * c + b] += Hpp[c][a][b];
            for (a = 0; a < n6; a++) {
                for (b = 0; b < n6; b++) loc[a * n6 + b] -= Y[a][0] * Hpl[b][0] + Y[a][1] * Hpl[b][1] + Y[a][2] * Hpl[b][2];
                loc[n6 * n6 + a] += gp[a] - (Y[a][0] * gl[0] + Y[a][1] * gl[1] + Y[a][2] * gl[2]);
            }
        }
        #pragma omp critical
        for (k = 0; k < nout; k++) out[k] += loc[k];
        free(loc);
    }
    return 0;
}

int orc_ba_backsub(const double *poses, const double *calib, const double *sigma, int C, const double *points,
                   const double *obs, const uint8_t *mask, const double *prior_w, const double *prior_xyz, int64_t N,
                   double lambda, const double *dpose, double *points_out, int use_omp)
{
    const int n6 = 6 * C;
    int64_t i;
    if (C < 1 || C > BA_MAX_CAMS) return -1;
    #pragma omp parallel for if (use_omp) schedule(static)
    for (i = 0; i < N; i++) {
        double Hi[3][3], gl[3], Hpl[BA_MAX_N6][3], Hpp[BA_MAX_CAMS][6][6], gp[BA_MAX_N6], r[3];
        int nv, con, a, b;
        ba_landmark(poses, calib, sigma, C, points + 3 * i, obs, mask, N, i, prior_w ? prior_w[i] : 0.0,
                    prior_xyz ? prior_xyz + 3 * i : points, lambda, Hi, gl, Hpl, Hpp, gp, &nv, &con);
        for (b = 0; b < 3; b++) { r[b] = gl[b]; for (a = 0; a < n6; a++) r[b] -= Hpl[a][b] * dpose[a]; }
        for (a = 0; a < 3; a++) points_out[3 * i + a] = points[3 * i + a] + Hi[a][0] * r[0] + Hi[a][1] * r[1] + Hi[a][2] * r[2];
    }
    return 0;
}
