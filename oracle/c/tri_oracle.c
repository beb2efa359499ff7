/*
 * ORACLE (test infrastructure, not product code).
 *
 * Plain-C restatement of the reference's native triangulation kernel
 *   /root/reference/Work/python_libs/triangulation_c/triangulation.c:24-42 (A, b construction),
 *   :65-83 (linear_LS_triangulation), :104-161 (iterative_LS_triangulation)
 * and of cv2.triangulatePoints as used by
 *   /root/reference/Work/python_libs/triangulation.py:6-25 (linear_eigen_triangulation),
 * generalised from 2 to C views (C == 2 is exactly the reference).
 *
 * The reference calls OpenCV 2.4.x `cvSolve(A, b, x, DECOMP_SVD)` (not vendored under
 * /root/reference, cannot be built here).  Its published algorithm is restated below:
 * one-sided (Hestenes) Jacobi SVD of the tall matrix, then back-substitution that drops
 * singular values w_i <= 2*DBL_EPSILON*sum(w).
 *
 * Pinned against the reference's known-answer file test_3.mat through
 * tests/test_oracle_golden.py.  Also timed by bench.py as the CPU baseline
 * ("cpu_baseline.kind": "port"), single-threaded like the shipped reference build
 * (triangulation_c/setup.py:12-13 passes openmp=False) or with OpenMP over landmarks
 * (the `#pragma omp parallel for` at triangulation.c:70,109).
 *
 * Layouts:  u [C][N][2] f64,  P [C][3][4] f64,  x [N][3] f64.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <string.h>

#define ORC_MAX_CAMS 8
#define ORC_MAX_ROWS (3 * ORC_MAX_CAMS)

/* One-sided Jacobi SVD of A (m x n, row-major, leading dim n), n <= 4, m <= ORC_MAX_ROWS.
 * On exit: A's columns are U*diag(w) (un-normalised left vectors), w[n] singular values
 * (unsorted), V (n x n, row-major, columns are right singular vectors). */
static void jacobi_svd(double *A, int m, int n, double *w, double *V)
{
    const double eps = DBL_EPSILON * 10;
    int i, j, k, iter;
    for (i = 0; i < n; i++)
        for (j = 0; j < n; j++)
            V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (i = 0; i < n; i++) {
        double s = 0;
        for (k = 0; k < m; k++) s += A[k * n + i] * A[k * n + i];
        w[i] = s;
    }
    for (iter = 0; iter < 30; iter++) {
        int changed = 0;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                double a = w[i], b = w[j], p = 0;
                for (k = 0; k < m; k++) p += A[k * n + i] * A[k * n + j];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = hypot(p, beta), c, s;
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (k = 0; k < m; k++) {
                    double t0 = c * A[k * n + i] + s * A[k * n + j];
                    double t1 = -s * A[k * n + i] + c * A[k * n + j];
                    A[k * n + i] = t0; A[k * n + j] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                w[i] = a; w[j] = b;
                changed = 1;
                for (k = 0; k < n; k++) {
                    double t0 = c * V[k * n + i] + s * V[k * n + j];
                    double t1 = -s * V[k * n + i] + c * V[k * n + j];
                    V[k * n + i] = t0; V[k * n + j] = t1;
                }
            }
        if (!changed) break;
    }
    for (i = 0; i < n; i++) {
        double s = 0;
        for (k = 0; k < m; k++) s += A[k * n + i] * A[k * n + i];
        w[i] = sqrt(s);
    }
}

/* x = argmin |A x - b| (minimum norm), A m x 3 row-major.  A is destroyed. */
static void svd_solve3(double *A, const double *b, int m, double *x)
{
    double w[3], V[9], thr = 0, y[3];
    int i, k;
    jacobi_svd(A, m, 3, w, V);
    for (i = 0; i < 3; i++) thr += w[i];
    thr *= 2 * DBL_EPSILON;
    for (i = 0; i < 3; i++) {
        if (w[i] > thr) {
            double s = 0;                       /* (U^T b)_i / w_i  with  U_i = A_i / w_i */
            for (k = 0; k < m; k++) s += A[k * 3 + i] * b[k];
            y[i] = s / (w[i] * w[i]);
        } else
            y[i] = 0;
    }
    for (i = 0; i < 3; i++)
        x[i] = V[i * 3 + 0] * y[0] + V[i * 3 + 1] * y[1] + V[i * 3 + 2] * y[2];
}

static void build_A_b(const double *u, const double *P, int C, int64_t N, int64_t xi, double *A, double *b)
{
    int c, k, l;
    for (c = 0; c < C; c++) {
        const double *Pc = P + 12 * c;
        const double *uc = u + ((int64_t)c * N + xi) * 2;
        for (k = 0; k < 2; k++) {                       /* triangulation.c:30-40 */
            for (l = 0; l < 3; l++)
                A[(2 * c + k) * 3 + l] = uc[k] * Pc[8 + l] - Pc[4 * k + l];
            b[2 * c + k] = -(uc[k] * Pc[11] - Pc[4 * k + 3]);
        }
    }
}

int orc_linear_ls(const double *u, const double *P, int C, int64_t N, double *x, int use_omp)
{
    int64_t xi;
    if (C < 2 || C > ORC_MAX_CAMS) return -1;
    #pragma omp parallel for if (use_omp) schedule(static)
    for (xi = 0; xi < N; xi++) {
        double A[2 * ORC_MAX_CAMS * 3], b[2 * ORC_MAX_CAMS];
        build_A_b(u, P, C, N, xi, A, b);
        svd_solve3(A, b, 2 * C, x + 3 * xi);            /* triangulation.c:81 */
    }
    return 0;
}

int orc_iterative_ls(const double *u, const double *P, int C, int64_t N, double tolerance, int max_iter,
                     double *x, int32_t *status, int use_omp)
{
    int64_t xi;
    if (C < 2 || C > ORC_MAX_CAMS) return -1;
    #pragma omp parallel for if (use_omp) schedule(dynamic, 1024)
    for (xi = 0; xi < N; xi++) {
        double A[2 * ORC_MAX_CAMS * 3], b[2 * ORC_MAX_CAMS], As[2 * ORC_MAX_CAMS * 3];
        double d[ORC_MAX_CAMS], dn[ORC_MAX_CAMS];
        double *xp = x + 3 * xi;
        int i, c, l;
        build_A_b(u, P, C, N, xi, A, b);
        for (c = 0; c < C; c++) d[c] = dn[c] = 1.0;     /* triangulation.c:122 */
        for (i = 0; i < max_iter; i++) {
            int conv = 1, zero = 0;
            memcpy(As, A, sizeof(double) * 6 * C);
            svd_solve3(As, b, 2 * C, xp);               /* :130 */
            for (c = 0; c < C; c++) {                   /* :133-134 */
                const double *Pc = P + 12 * c;
                dn[c] = Pc[8] * xp[0] + Pc[9] * xp[1] + Pc[10] * xp[2] + Pc[11];
                if (!(fabs(dn[c] - d[c]) <= tolerance)) conv = 0;
                if (dn[c] == 0) zero = 1;
            }
            if (conv || zero) break;                    /* :137-140 */
            for (c = 0; c < C; c++) {                   /* :143-146 */
                double s = 1. / dn[c];
                for (l = 0; l < 6; l++) A[6 * c + l] *= s;
                b[2 * c] *= s; b[2 * c + 1] *= s;
                d[c] = dn[c];                           /* :149-150 */
            }
        }
        {
            int front = 1, s;
            for (c = 0; c < C; c++) if (!(dn[c] > 0)) front = 0;
            s = (i < max_iter) && front;                /* :154-155 */
            for (c = 0; c < C; c++) if (dn[c] <= 0) s -= (1 << c);   /* :156-159 */
            status[xi] = s;
        }
    }
    return 0;
}

int orc_linear_eigen(const double *u, const double *P, int C, int64_t N, double max_coord,
                     double *x, uint8_t *ok, int use_omp)
{
    int64_t xi;
    if (C < 2 || C > ORC_MAX_CAMS) return -1;
    #pragma omp parallel for if (use_omp) schedule(static)
    for (xi = 0; xi < N; xi++) {
        double A[ORC_MAX_ROWS * 4], w[4], V[16];
        int c, l, imin = 0;
        for (c = 0; c < C; c++) {
            const double *Pc = P + 12 * c;
            const double *uc = u + ((int64_t)c * N + xi) * 2;
            for (l = 0; l < 4; l++) {
                A[(3 * c + 0) * 4 + l] = uc[0] * Pc[8 + l] - Pc[l];
                A[(3 * c + 1) * 4 + l] = uc[1] * Pc[8 + l] - Pc[4 + l];
                A[(3 * c + 2) * 4 + l] = uc[0] * Pc[4 + l] - uc[1] * Pc[l];
            }
        }
        jacobi_svd(A, 3 * C, 4, w, V);
        for (l = 1; l < 4; l++) if (w[l] < w[imin]) imin = l;
        {
            double X3 = V[3 * 4 + imin];
            int good = 1;
            for (l = 0; l < 3; l++) {
                double v = V[l * 4 + imin] / X3;        /* triangulation.py:22 */
                x[3 * xi + l] = v;
                if (!(fabs(v) <= max_coord)) good = 0;  /* NaN/Inf -> False, triangulation.py:23 */
            }
            ok[xi] = (uint8_t)good;
        }
    }
    return 0;
}
