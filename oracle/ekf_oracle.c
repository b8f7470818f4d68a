/*
 * oracle/ekf_oracle.c -- CPU ORACLE of HybVIO's EKF. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * nothing under hybvio_amd/ links, imports or calls it.
 *
 * A plain-C, f64, column-major restatement of src/odometry/ekf.cpp (the algorithm is entirely
 * in-tree; Eigen -- absent here, version unpinned -- only supplies dense products, a pivoted
 * LDLT, Matrix4d::exp() and sparse shift-matrix products). Every function cites the lines it
 * follows. Dense expressions are evaluated the way the reference writes them (HP, S, LDLT, K,
 * P -= K*HP; Joseph form with two full stateDim^3 products), so this file is also what the
 * HIP kernels -- which use algebraically equivalent but restructured forms -- are checked against.
 *
 * Pinned by the reference's own EKF tests (re-expressed in tests/test_oracle_ekf.py):
 *   test/ekf.cpp:19-71   chi-squared innovation value via LDLT  (Matlab 1.7626 +- 0.1)
 *   test/ekf.cpp:73-117  der_predict: analytic dydx vs forward differences  (< 1e-3)
 *   test/ekf.cpp:119-145 transformTo round trip on test/data/{P,m}.csv  (1e-6 / 1e-3)
 *   test/util.cpp:9-60   quat2rmat / quat2rmat_d Matlab goldens
 * No golden posterior (m, P) of predict / updateVisualTrack / augmentation exists upstream, so
 * the covariance numerics are additionally cross-checked against an independent numpy/scipy
 * implementation in the same test file.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "chi2inv95.h"

/* state layout: src/odometry/ekf.hpp:26-50 */
enum { POS = 0, VEL = 3, ORI = 6, BGA = 10, BAA = 13, BAT = 16, SFT = 19, CAM = 20, INER_DIM = 20,
       POSE_DIM = 7, MAP_POINT_DIM = 3 };
enum { Q_ACC = 0, Q_GYRO = 3, Q_BGA_DRIFT = 6, Q_BAA_DRIFT = 9, Q_DIM = 12 };
enum { VU_INLIER = 0, VU_NOT_COMPUTED = 1, VU_RMSE = 2, VU_CHI2 = 3 };   /* ekf.hpp:54-59 */

/* the odometry parameters the EKF reads (codegen/parameter_definitions.c:68-160) */
typedef struct orc_ekf_params {
    int cameraTrailLength, hybridMapSize;
    double noiseScale, gravity, augmentR, initZuptR, rotationZuptR;
    double noiseInitialPos, noiseInitialOri, noiseInitialVel, noiseInitialPosTrail, noiseInitialOriTrail;
    double noiseInitialBGA, noiseInitialBAA, noiseInitialBAT, noiseInitialSFT;
    double noiseProcessAcc, noiseProcessGyro, noiseProcessBAA, noiseProcessBGA;
    double noiseProcessBAARev, noiseProcessBGARev;
} orc_ekf_params;

typedef struct orc_ekf {
    orc_ekf_params par;
    double noiseScale;              /* = par.noiseScale^2 (ekf.cpp:154) */
    double gravity[3];
    int camPoseCount, hybridMapDim, n;   /* n = stateDim */
    double *m, *P;                  /* n, n*n column-major */
    double Q[Q_DIM * Q_DIM];
    double dydx[INER_DIM * INER_DIM], dydq[INER_DIM * Q_DIM];
    double *tmp0, *tmp1, *HP, *K, *S;    /* n*n work matrices */
    int *perm;
    int augmentCount;
    double *augmentTimes;
    double time, ZUPTtime, ZRUPTtime, initZUPTtime;
    int wasStationary;
    double prevSampleT, firstSampleT;
    int firstSample;
} orc_ekf;

#define M_(A, ld, i, j) ((A)[(size_t)(j) * (ld) + (i)])
static double pow2(double x) { return x * x; }

void orc_ekf_default_params(orc_ekf_params *p)
{
    p->cameraTrailLength = 20; p->hybridMapSize = 0;
    p->noiseScale = 100; p->gravity = 9.819; p->augmentR = 1e-9; p->initZuptR = 1e-4; p->rotationZuptR = 1e-6;
    p->noiseInitialPos = 1e-5; p->noiseInitialOri = 0.0316227766; p->noiseInitialVel = 0.1;
    p->noiseInitialPosTrail = 100; p->noiseInitialOriTrail = 3.16227766;
    p->noiseInitialBGA = 1e-3; p->noiseInitialBAA = 1e-6; p->noiseInitialBAT = 1e-5; p->noiseInitialSFT = 1e-5;
    p->noiseProcessAcc = 0.003; p->noiseProcessGyro = 0.00017; p->noiseProcessBAA = 1e-4; p->noiseProcessBGA = 0;
    p->noiseProcessBAARev = 0.1; p->noiseProcessBGARev = 0.1;
}

/* ---- small dense helpers (column-major) ---- */

/* C(m x n) = alpha * op(A) * op(B) + beta * C ; ta/tb: transpose flags; lda/ldb/ldc leading dims */
static void gemm(int ta, int tb, int m, int n, int k, double alpha, const double *A, int lda,
                 const double *B, int ldb, double beta, double *C, int ldc)
{
    for (int j = 0; j < n; j++)
        for (int i = 0; i < m; i++) {
            double s = 0.0;
            for (int p = 0; p < k; p++) {
                const double a = ta ? M_(A, lda, p, i) : M_(A, lda, i, p);
                const double b = tb ? M_(B, ldb, j, p) : M_(B, ldb, p, j);
                s += a * b;
            }
            M_(C, ldc, i, j) = alpha * s + (beta == 0.0 ? 0.0 : beta * M_(C, ldc, i, j));
        }
}

/* Pivoted LDL^T of a symmetric matrix, as Eigen::LDLT (used at ekf.cpp:60,73,782,865):
 * P A P^T = L D L^T, pivot = largest remaining |diagonal|. A (n x n, ld) is overwritten by L (unit
 * lower) and D (diagonal); perm[k] = row swapped with k at step k. */
static void ldlt_factor(int n, double *A, int ld, int *perm)
{
    for (int k = 0; k < n; k++) {
        int piv = k;
        double best = fabs(M_(A, ld, k, k));
        for (int i = k + 1; i < n; i++)
            if (fabs(M_(A, ld, i, i)) > best) { best = fabs(M_(A, ld, i, i)); piv = i; }
        perm[k] = piv;
        if (piv != k) {   /* symmetric row/column interchange on the lower triangle */
            for (int j = 0; j < k; j++) { double t = M_(A, ld, k, j); M_(A, ld, k, j) = M_(A, ld, piv, j); M_(A, ld, piv, j) = t; }
            for (int i = piv + 1; i < n; i++) { double t = M_(A, ld, i, k); M_(A, ld, i, k) = M_(A, ld, i, piv); M_(A, ld, i, piv) = t; }
            { double t = M_(A, ld, k, k); M_(A, ld, k, k) = M_(A, ld, piv, piv); M_(A, ld, piv, piv) = t; }
            for (int i = k + 1; i < piv; i++) { double t = M_(A, ld, i, k); M_(A, ld, i, k) = M_(A, ld, piv, i); M_(A, ld, piv, i) = t; }
        }
        double d = M_(A, ld, k, k);
        for (int j = 0; j < k; j++) d -= M_(A, ld, k, j) * M_(A, ld, k, j) * M_(A, ld, j, j);
        M_(A, ld, k, k) = d;
        for (int i = k + 1; i < n; i++) {
            double s = M_(A, ld, i, k);
            for (int j = 0; j < k; j++) s -= M_(A, ld, i, j) * M_(A, ld, k, j) * M_(A, ld, j, j);
            M_(A, ld, i, k) = (d != 0.0) ? s / d : 0.0;
        }
    }
}

/* Solve (P^T L D L^T P) X = B in place; B is n x nrhs, ldb. */
static void ldlt_solve(int n, const double *A, int ld, const int *perm, double *B, int ldb, int nrhs)
{
    for (int c = 0; c < nrhs; c++) {
        double *b = B + (size_t)c * ldb;
        for (int k = 0; k < n; k++) if (perm[k] != k) { double t = b[k]; b[k] = b[perm[k]]; b[perm[k]] = t; }
        for (int i = 0; i < n; i++) { double s = b[i]; for (int j = 0; j < i; j++) s -= M_(A, ld, i, j) * b[j]; b[i] = s; }
        for (int i = 0; i < n; i++) { const double d = M_(A, ld, i, i); b[i] = (fabs(d) > 1e-300) ? b[i] / d : 0.0; }
        for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int j = i + 1; j < n; j++) s -= M_(A, ld, j, i) * b[j]; b[i] = s; }
        for (int k = n - 1; k >= 0; k--) if (perm[k] != k) { double t = b[k]; b[k] = b[perm[k]]; b[perm[k]] = t; }
    }
}

/* exported for the solver pin of test/ekf.cpp:19-71: returns v' * (M \ v) via LDLT */
double orc_ldlt_quadratic_form(int n, const double *M, const double *v)
{
    double *A = (double *)malloc(sizeof(double) * n * n), *x = (double *)malloc(sizeof(double) * n);
    int *perm = (int *)malloc(sizeof(int) * n);
    memcpy(A, M, sizeof(double) * n * n); memcpy(x, v, sizeof(double) * n);
    ldlt_factor(n, A, n, perm);
    ldlt_solve(n, A, n, perm, x, n, 1);
    double t = 0; for (int i = 0; i < n; i++) t += x[i] * v[i];
    free(A); free(x); free(perm);
    return t;
}

/* odometry::util::quat2rmat / quat2rmat_d (src/odometry/util.cpp:10-47); R, dR[k] are 3x3 column-major */
void orc_quat2rmat(const double q[4], double R[9])
{
    M_(R, 3, 0, 0) = q[0]*q[0]+q[1]*q[1]-q[2]*q[2]-q[3]*q[3]; M_(R, 3, 0, 1) = 2*q[1]*q[2] - 2*q[0]*q[3]; M_(R, 3, 0, 2) = 2*q[1]*q[3] + 2*q[0]*q[2];
    M_(R, 3, 1, 0) = 2*q[1]*q[2] + 2*q[0]*q[3]; M_(R, 3, 1, 1) = q[0]*q[0]-q[1]*q[1]+q[2]*q[2]-q[3]*q[3]; M_(R, 3, 1, 2) = 2*q[2]*q[3] - 2*q[0]*q[1];
    M_(R, 3, 2, 0) = 2*q[1]*q[3] - 2*q[0]*q[2]; M_(R, 3, 2, 1) = 2*q[2]*q[3] + 2*q[0]*q[1]; M_(R, 3, 2, 2) = q[0]*q[0]-q[1]*q[1]-q[2]*q[2]+q[3]*q[3];
}

void orc_quat2rmat_d(const double q[4], double R[9], double dR[36])
{
    const double rows[4][9] = {   /* row-major literals exactly as written in util.cpp:30-45 */
        { 2*q[0], -2*q[3],  2*q[2],   2*q[3],  2*q[0], -2*q[1],  -2*q[2],  2*q[1],  2*q[0] },
        { 2*q[1],  2*q[2],  2*q[3],   2*q[2], -2*q[1], -2*q[0],   2*q[3],  2*q[0], -2*q[1] },
        {-2*q[2],  2*q[1],  2*q[0],   2*q[1],  2*q[2],  2*q[3],  -2*q[0],  2*q[3], -2*q[2] },
        {-2*q[3], -2*q[0],  2*q[1],   2*q[0], -2*q[3],  2*q[2],   2*q[1],  2*q[2],  2*q[3] } };
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) M_(dR + 9 * k, 3, i, j) = rows[k][3 * i + j];
    orc_quat2rmat(q, R);
}

static void normalize4(double *q)
{
    const double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
    if (n > 0.0) { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }   /* Eigen: zero vectors stay zero */
}

/* ---- construction (ekf.cpp:153-296) ---- */

orc_ekf *orc_ekf_create(const orc_ekf_params *par)
{
    orc_ekf *e = (orc_ekf *)calloc(1, sizeof(orc_ekf));
    e->par = *par;
    e->noiseScale = par->noiseScale * par->noiseScale;
    e->gravity[0] = 0; e->gravity[1] = 0; e->gravity[2] = -par->gravity;
    e->camPoseCount = par->cameraTrailLength;
    e->hybridMapDim = par->hybridMapSize * MAP_POINT_DIM;
    const int n = e->n = INER_DIM + e->camPoseCount * POSE_DIM + e->hybridMapDim;
    e->m = (double *)calloc(n, sizeof(double));
    e->P = (double *)calloc((size_t)n * n, sizeof(double));
    e->tmp0 = (double *)calloc((size_t)n * n, sizeof(double));
    e->tmp1 = (double *)calloc((size_t)n * n, sizeof(double));
    e->HP = (double *)calloc((size_t)n * n, sizeof(double));
    e->K = (double *)calloc((size_t)n * n, sizeof(double));
    e->S = (double *)calloc((size_t)n * n, sizeof(double));
    e->perm = (int *)calloc(n, sizeof(int));
    e->augmentTimes = (double *)calloc(e->camPoseCount + 2, sizeof(double));
    e->ZUPTtime = e->ZRUPTtime = e->initZUPTtime = -1.0;
    e->prevSampleT = e->firstSampleT = -1.0;
    e->firstSample = 1;

    e->m[ORI] = 1;
    e->m[BAT] = e->m[BAT + 1] = e->m[BAT + 2] = 1.0;
    double *P = e->P;
    for (int i = 0; i < 3; i++) {
        M_(P, n, POS + i, POS + i) = pow2(par->noiseInitialPos);
        M_(P, n, VEL + i, VEL + i) = pow2(par->noiseInitialVel);
        M_(P, n, BGA + i, BGA + i) = pow2(par->noiseInitialBGA);
        M_(P, n, BAA + i, BAA + i) = pow2(par->noiseInitialBAA);
        M_(P, n, BAT + i, BAT + i) = pow2(par->noiseInitialBAT);
    }
    for (int i = 0; i < 4; i++) M_(P, n, ORI + i, ORI + i) = 1.0;   /* placeholder */
    M_(P, n, SFT, SFT) = pow2(par->noiseInitialSFT);
    const double noisePos = pow2(par->noiseInitialPosTrail), noiseOri = pow2(par->noiseInitialOriTrail);
    for (int c = 0; c < e->camPoseCount; c++) {
        const int b = CAM + c * POSE_DIM;
        for (int i = 0; i < 3; i++) M_(P, n, b + i, b + i) = noisePos;
        for (int i = 3; i < 7; i++) M_(P, n, b + i, b + i) = noiseOri;
    }
    for (int i = 0; i < 3; i++) {
        M_(e->Q, Q_DIM, Q_ACC + i, Q_ACC + i) = pow2(par->noiseProcessAcc);
        M_(e->Q, Q_DIM, Q_GYRO + i, Q_GYRO + i) = pow2(par->noiseProcessGyro);
    }
    for (size_t i = 0; i < (size_t)n * n; i++) P[i] *= e->noiseScale;
    for (int i = 0; i < Q_DIM * Q_DIM; i++) e->Q[i] *= e->noiseScale;
    return e;
}

orc_ekf *orc_ekf_clone(const orc_ekf *o)
{
    orc_ekf *e = orc_ekf_create(&o->par);
    const int n = o->n;
    memcpy(e->m, o->m, sizeof(double) * n);
    memcpy(e->P, o->P, sizeof(double) * n * n);
    memcpy(e->Q, o->Q, sizeof e->Q);
    memcpy(e->dydx, o->dydx, sizeof e->dydx); memcpy(e->dydq, o->dydq, sizeof e->dydq);
    e->augmentCount = o->augmentCount;
    memcpy(e->augmentTimes, o->augmentTimes, sizeof(double) * (o->camPoseCount + 2));
    e->time = o->time; e->ZUPTtime = o->ZUPTtime; e->ZRUPTtime = o->ZRUPTtime; e->initZUPTtime = o->initZUPTtime;
    e->wasStationary = o->wasStationary; e->prevSampleT = o->prevSampleT; e->firstSampleT = o->firstSampleT;
    e->firstSample = o->firstSample;
    return e;
}

void orc_ekf_free(orc_ekf *e)
{
    if (!e) return;
    free(e->m); free(e->P); free(e->tmp0); free(e->tmp1); free(e->HP); free(e->K); free(e->S);
    free(e->perm); free(e->augmentTimes); free(e);
}

int orc_ekf_state_dim(const orc_ekf *e) { return e->n; }
double *orc_ekf_state(orc_ekf *e) { return e->m; }
double *orc_ekf_cov(orc_ekf *e) { return e->P; }
double *orc_ekf_process_noise(orc_ekf *e) { return e->Q; }
double *orc_ekf_dydx(orc_ekf *e) { return e->dydx; }
double orc_ekf_platform_time(const orc_ekf *e) { return e->firstSampleT + e->time; }
int orc_ekf_pose_count(const orc_ekf *e) { return e->augmentCount + 1; }
int orc_ekf_was_stationary(const orc_ekf *e) { return e->wasStationary; }
double orc_ekf_history_time(const orc_ekf *e, int i)   /* ekf.cpp:553-560 */
{
    if (i == -1) return orc_ekf_platform_time(e);
    return e->augmentTimes[e->augmentCount - i - 1];
}

void orc_ekf_set_first_sample_time(orc_ekf *e, double t)   /* ekf.cpp:1035-1041 */
{
    e->firstSample = 0; e->firstSampleT = t; e->prevSampleT = t; e->time = t;
}

void orc_ekf_normalize_quaternions(orc_ekf *e, int only_current)   /* ekf.cpp:1024-1032 */
{
    normalize4(e->m + ORI);
    if (only_current) return;
    for (int i = 0; i < e->camPoseCount; i++) normalize4(e->m + CAM + POSE_DIM * i + 3);
}

void orc_ekf_maintain_psd(orc_ekf *e)   /* ekf.cpp:1059-1067: P = (P + P')/2 */
{
    const int n = e->n;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) M_(e->tmp0, n, i, j) = 0.5 * (M_(e->P, n, i, j) + M_(e->P, n, j, i));
    memcpy(e->P, e->tmp0, sizeof(double) * n * n);
}

/* Eigen::Quaterniond::FromTwoVectors(-gravity, xa) + ekf.cpp:298-317 */
void orc_ekf_initialize_orientation(orc_ekf *e, const double xa[3])
{
    const int n = e->n;
    double a[3] = { -e->gravity[0], -e->gravity[1], -e->gravity[2] }, b[3] = { xa[0], xa[1], xa[2] };
    const double na = sqrt(a[0]*a[0]+a[1]*a[1]+a[2]*a[2]), nb = sqrt(b[0]*b[0]+b[1]*b[1]+b[2]*b[2]);
    for (int i = 0; i < 3; i++) { a[i] /= na; b[i] /= nb; }
    const double c = a[0]*b[0] + a[1]*b[1] + a[2]*b[2];
    double q[4];
    if (c < -1.0 + 1e-12) {
        /* antiparallel: Eigen picks an axis orthogonal to a via an SVD; any such axis gives a valid
         * half-turn. (Not reachable with a sane accelerometer sample.) */
        double ax[3] = { 1, 0, 0 };
        if (fabs(a[0]) > 0.9) { ax[0] = 0; ax[1] = 1; }
        double d = ax[0]*a[0] + ax[1]*a[1] + ax[2]*a[2];
        for (int i = 0; i < 3; i++) ax[i] -= d * a[i];
        const double nn = sqrt(ax[0]*ax[0]+ax[1]*ax[1]+ax[2]*ax[2]);
        q[0] = 0; q[1] = ax[0] / nn; q[2] = ax[1] / nn; q[3] = ax[2] / nn;
    } else {
        const double ax[3] = { a[1]*b[2]-a[2]*b[1], a[2]*b[0]-a[0]*b[2], a[0]*b[1]-a[1]*b[0] };
        const double s = sqrt((1.0 + c) * 2.0), invs = 1.0 / s;
        q[0] = s * 0.5; q[1] = ax[0] * invs; q[2] = ax[1] * invs; q[3] = ax[2] * invs;
    }
    memcpy(e->m + ORI, q, sizeof q);
    const double v = pow2(e->par.noiseInitialOri) * e->noiseScale;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) M_(e->P, n, ORI + i, ORI + j) = 0.0;
    for (int i = 0; i < 3; i++) M_(e->P, n, ORI + i, ORI + i) = v;   /* last component fixed (variance 0) */
}

/* ---- predict (ekf.cpp:320-514) ---- */

static void mat4_vec(const double A[16], const double x[4], double y[4])
{
    for (int i = 0; i < 4; i++) { double s = 0; for (int j = 0; j < 4; j++) s += M_(A, 4, i, j) * x[j]; y[i] = s; }
}

void orc_ekf_predict(orc_ekf *e, double t, const double xg[3], const double xa[3])
{
    const int n = e->n;
    double dt = 0.0;
    if (!e->firstSample) { dt = t - e->prevSampleT; e->time = t - e->firstSampleT; }
    else { e->firstSampleT = t; e->firstSample = 0; }
    e->prevSampleT = t;
    if (dt <= 0.0) return;

    double *m = e->m, *dydx = e->dydx, *dydq = e->dydq;
    memset(dydx, 0, sizeof e->dydx); memset(dydq, 0, sizeof e->dydq);
    for (int i = 0; i < INER_DIM; i++) M_(dydx, INER_DIM, i, i) = 1.0;

    /* bounded random walks on the biases (ekf.cpp:397-412) */
    if (e->par.noiseProcessBAA > 0.0) {
        double v = e->noiseScale * pow2(e->par.noiseProcessBAA);
        const double th = e->par.noiseProcessBAARev;
        if (th > 0.0) v *= (1 - exp(-2 * dt * th)) / (2 * th);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M_(e->Q, Q_DIM, Q_BAA_DRIFT + i, Q_BAA_DRIFT + j) = (i == j) ? v : 0.0;
    }
    if (e->par.noiseProcessBGA > 0.0) {
        double v = e->noiseScale * pow2(e->par.noiseProcessBGA);
        const double th = e->par.noiseProcessBGARev;
        if (th > 0.0) v *= (1 - exp(-2 * dt * th)) / (2 * th);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M_(e->Q, Q_DIM, Q_BGA_DRIFT + i, Q_BGA_DRIFT + j) = (i == j) ? v : 0.0;
    }

    /* A = exp(-dt/2 * Omega(w)) (ekf.cpp:415-425). Omega^2 = -|w|^2 I, so the exponential has the
     * closed form cos(th) I + sin(th)/th * S with th = |w| dt / 2 (Eigen evaluates a Pade
     * approximant of the same matrix; tests compare against scipy.linalg.expm). */
    const double w[3] = { xg[0] - m[BGA], xg[1] - m[BGA + 1], xg[2] - m[BGA + 2] };
    const double Srow[16] = { 0, -w[0], -w[1], -w[2],  w[0], 0, -w[2], w[1],  w[1], w[2], 0, -w[0],  w[2], -w[1], w[0], 0 };
    double S[16], A[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) M_(S, 4, i, j) = Srow[4 * i + j] * (-dt / 2);
    {
        const double th = sqrt(w[0]*w[0] + w[1]*w[1] + w[2]*w[2]) * dt / 2;
        const double c = cos(th), sc = th > 1e-8 ? sin(th) / th : 1.0 - th * th / 6.0;
        for (int i = 0; i < 16; i++) A[i] = sc * S[i];
        for (int i = 0; i < 4; i++) M_(A, 4, i, i) += c;
    }

    double qn[4], R[9], dR[36];
    mat4_vec(A, m + ORI, qn);
    orc_quat2rmat_d(qn, R, dR);

    for (int i = 0; i < 3; i++) m[POS + i] += m[VEL + i] * dt;                    /* position */
    double Txab[3];
    for (int i = 0; i < 3; i++) Txab[i] = m[BAT + i] * xa[i] - m[BAA + i];
    for (int i = 0; i < 3; i++) {                                                /* velocity */
        double s = 0; for (int j = 0; j < 3; j++) s += M_(R, 3, j, i) * Txab[j];  /* R' * Txab */
        m[VEL + i] += (s + e->gravity[i]) * dt;
    }
    double prevQuat[4]; memcpy(prevQuat, m + ORI, sizeof prevQuat);
    memcpy(m + ORI, qn, sizeof qn);                                               /* orientation */
    if (e->par.noiseProcessBAA > 0.0) { const double f = exp(-dt * e->par.noiseProcessBAARev); for (int i = 0; i < 3; i++) m[BAA + i] *= f; }
    if (e->par.noiseProcessBGA > 0.0) { const double f = exp(-dt * e->par.noiseProcessBGARev); for (int i = 0; i < 3; i++) m[BGA + i] *= f; }

    for (int i = 0; i < 3; i++) M_(dydx, INER_DIM, POS + i, VEL + i) = dt;
    /* d vel / d quat (ekf.cpp:457-461) */
    double T34[12];
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 3; i++) {
            double s = 0; for (int j = 0; j < 3; j++) s += M_(dR + 9 * k, 3, j, i) * Txab[j];
            M_(T34, 3, i, k) = s * dt;
        }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0; for (int k = 0; k < 4; k++) s += M_(T34, 3, i, k) * M_(A, 4, k, j);
            M_(dydx, INER_DIM, VEL + i, ORI + j) = s;
        }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) M_(dydx, INER_DIM, ORI + i, ORI + j) = M_(A, 4, i, j);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M_(dydq, INER_DIM, VEL + i, Q_ACC + j) = M_(R, 3, j, i) * dt;
    /* d quat / d gyro noise (ekf.cpp:469-476) */
    const double h = dt / 2;
    const double dS[3][16] = {
        { 0, h, 0, 0,  -h, 0, 0, 0,  0, 0, 0, h,  0, 0, -h, 0 },
        { 0, 0, h, 0,  0, 0, 0, -h,  -h, 0, 0, 0,  0, h, 0, 0 },
        { 0, 0, 0, h,  0, 0, h, 0,  0, -h, 0, 0,  -h, 0, 0, 0 } };
    for (int g = 0; g < 3; g++) {
        double t1[4], t2[4];
        for (int i = 0; i < 4; i++) { double s = 0; for (int j = 0; j < 4; j++) s += dS[g][4 * i + j] * prevQuat[j]; t1[i] = s; }
        mat4_vec(A, t1, t2);
        for (int i = 0; i < 4; i++) M_(dydq, INER_DIM, ORI + i, Q_GYRO + g) = t2[i];
    }
    for (int i = 0; i < 3; i++) { M_(dydq, INER_DIM, BGA + i, Q_BGA_DRIFT + i) = 1.0; M_(dydq, INER_DIM, BAA + i, Q_BAA_DRIFT + i) = 1.0; }
    /* d vel / d gyro noise = dydx(VEL,ORI) * dydq(ORI,GYRO) (ekf.cpp:485) */
    for (int i = 0; i < 3; i++)
        for (int g = 0; g < 3; g++) {
            double s = 0; for (int k = 0; k < 4; k++) s += M_(dydx, INER_DIM, VEL + i, ORI + k) * M_(dydq, INER_DIM, ORI + k, Q_GYRO + g);
            M_(dydq, INER_DIM, VEL + i, Q_GYRO + g) = s;
        }
    for (int i = 0; i < 3; i++) for (int g = 0; g < 3; g++) M_(dydx, INER_DIM, VEL + i, BGA + g) = -M_(dydq, INER_DIM, VEL + i, Q_GYRO + g);
    for (int i = 0; i < 4; i++) for (int g = 0; g < 3; g++) M_(dydx, INER_DIM, ORI + i, BGA + g) = -M_(dydq, INER_DIM, ORI + i, Q_GYRO + g);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        M_(dydx, INER_DIM, VEL + i, BAA + j) = -M_(R, 3, j, i) * dt;
        M_(dydx, INER_DIM, VEL + i, BAT + j) = M_(R, 3, j, i) * xa[j] * dt;
    }

    /* covariance (ekf.cpp:504-508): P00 = F P00 F' + L Q L'; P10 = P10 F'; P01 = F P01 */
    double FP[INER_DIM * INER_DIM], LQ[INER_DIM * Q_DIM], P00[INER_DIM * INER_DIM];
    gemm(0, 0, INER_DIM, INER_DIM, INER_DIM, 1.0, dydx, INER_DIM, e->P, n, 0.0, FP, INER_DIM);
    gemm(0, 1, INER_DIM, INER_DIM, INER_DIM, 1.0, FP, INER_DIM, dydx, INER_DIM, 0.0, P00, INER_DIM);
    gemm(0, 0, INER_DIM, Q_DIM, Q_DIM, 1.0, dydq, INER_DIM, e->Q, Q_DIM, 0.0, LQ, INER_DIM);
    gemm(0, 1, INER_DIM, INER_DIM, Q_DIM, 1.0, LQ, INER_DIM, dydq, INER_DIM, 1.0, P00, INER_DIM);
    const int r = n - INER_DIM;
    gemm(0, 1, r, INER_DIM, INER_DIM, 1.0, e->P + INER_DIM, n, dydx, INER_DIM, 0.0, e->tmp0, r);           /* P10 F' */
    gemm(0, 0, INER_DIM, r, INER_DIM, 1.0, dydx, INER_DIM, e->P + (size_t)INER_DIM * n, n, 0.0, e->tmp1, INER_DIM);  /* F P01 */
    for (int j = 0; j < INER_DIM; j++) for (int i = 0; i < INER_DIM; i++) M_(e->P, n, i, j) = M_(P00, INER_DIM, i, j);
    for (int j = 0; j < INER_DIM; j++) for (int i = 0; i < r; i++) M_(e->P, n, INER_DIM + i, j) = M_(e->tmp0, r, i, j);
    for (int j = 0; j < r; j++) for (int i = 0; i < INER_DIM; i++) M_(e->P, n, i, INER_DIM + j) = M_(e->tmp1, INER_DIM, i, j);
}

/* ---- generic update (ekf.cpp:25-32,57-82): H is nr x l (truncated), R = r_diag * I ---- */
static void kf_update(orc_ekf *e, int nr, int l, const double *H, const double *y, double r_diag)
{
    const int n = e->n;
    double *HP = e->HP, *S = e->S, *K = e->K, *X = e->tmp0;
    gemm(0, 0, nr, n, l, 1.0, H, nr, e->P, n, 0.0, HP, nr);                 /* HP = H * P.topRows(l) */
    gemm(0, 1, nr, nr, l, 1.0, HP, nr, H, nr, 0.0, S, nr);                  /* S = HP.leftCols(l) * H' */
    for (int i = 0; i < nr; i++) M_(S, nr, i, i) += r_diag;
    ldlt_factor(nr, S, nr, e->perm);
    memcpy(X, HP, sizeof(double) * nr * n);
    ldlt_solve(nr, S, nr, e->perm, X, nr, n);                               /* X = S^-1 HP ; K = X' */
    for (int i = 0; i < n; i++) for (int j = 0; j < nr; j++) M_(K, n, i, j) = M_(X, nr, j, i);
    double v[256];
    for (int i = 0; i < nr; i++) { double s = 0; for (int j = 0; j < l; j++) s += M_(H, nr, i, j) * e->m[j]; v[i] = y[i] - s; }
    for (int i = 0; i < n; i++) { double s = 0; for (int j = 0; j < nr; j++) s += M_(K, n, i, j) * v[j]; e->m[i] += s; }
    gemm(0, 0, n, n, nr, -1.0, K, n, HP, nr, 1.0, e->P, n);                 /* P -= K * HP */
    normalize4(e->m + ORI);
}

static void identity_block(double *H, int nr, int l, int col0)
{
    memset(H, 0, sizeof(double) * nr * l);
    for (int i = 0; i < nr; i++) M_(H, nr, i, col0 + i) = 1.0;
}

void orc_ekf_update_zupt(orc_ekf *e, double r)   /* ekf.cpp:573-590 */
{
    if (e->time - e->ZUPTtime < 0.25) return;
    e->ZUPTtime = e->time; e->wasStationary = 1;
    double H[3 * (VEL + 3)], y[3] = { 0, 0, 0 };
    identity_block(H, 3, VEL + 3, VEL);
    kf_update(e, 3, VEL + 3, H, y, r * e->noiseScale);
}

void orc_ekf_update_zupt_initialization(orc_ekf *e)   /* ekf.cpp:594-611 */
{
    if (e->wasStationary || e->time > 60 || e->time - e->initZUPTtime < 0.1) return;
    e->initZUPTtime = e->time;
    double H[3 * (VEL + 3)], y[3] = { 0, 0, 0 };
    identity_block(H, 3, VEL + 3, VEL);
    kf_update(e, 3, VEL + 3, H, y, e->par.initZuptR * e->noiseScale * exp(0.5 * e->time));
}

void orc_ekf_update_zrupt(orc_ekf *e, const double xg[3])   /* ekf.cpp:614-625 */
{
    if (e->time - e->ZRUPTtime < 0.25) return;
    e->ZRUPTtime = e->time;
    double H[3 * (BGA + 3)];
    identity_block(H, 3, BGA + 3, BGA);
    kf_update(e, 3, BGA + 3, H, xg, e->par.rotationZuptR * e->noiseScale);
}

void orc_ekf_update_pseudo_velocity(orc_ekf *e, double defaultSpeed, double r)   /* ekf.cpp:628-649 */
{
    const int n = e->n, l = VEL + 2;
    const double hn = sqrt(e->m[VEL] * e->m[VEL] + e->m[VEL + 1] * e->m[VEL + 1]);
    if (hn <= 1e-7) return;
    double H[VEL + 2]; memset(H, 0, sizeof H);
    for (int i = 0; i < 2; i++) H[VEL + i] = e->m[VEL + i] / hn;
    double *HP = e->HP, *K = e->K;
    gemm(0, 0, 1, n, l, 1.0, H, 1, e->P, n, 0.0, HP, 1);
    double s = 0; for (int j = 0; j < l; j++) s += HP[j] * H[j];
    s += r * e->noiseScale;
    for (int i = 0; i < n; i++) K[i] = HP[i] / s;
    for (int i = 0; i < n; i++) e->m[i] += K[i] * (defaultSpeed - hn);
    gemm(0, 0, n, n, 1, -1.0, K, n, HP, 1, 1.0, e->P, n);
    normalize4(e->m + ORI);
}

void orc_ekf_update_position(orc_ekf *e, const double pos[3], double r)   /* ekf.cpp:651-657 */
{
    double H[3 * (POS + 3)];
    identity_block(H, 3, POS + 3, POS);
    kf_update(e, 3, POS + 3, H, pos, r * e->noiseScale);
    orc_ekf_maintain_psd(e);
}

void orc_ekf_update_zero_height(orc_ekf *e, double r)   /* ekf.cpp:660-668 */
{
    double H[POS + 3] = { 0, 0, 1 }, y[1] = { 0 };
    kf_update(e, 1, POS + 3, H, y, r * e->noiseScale);
    orc_ekf_maintain_psd(e);
}

void orc_ekf_update_orientation(orc_ekf *e, const double q[4], double r)   /* ekf.cpp:670-677 */
{
    double H[4 * (ORI + 4)];
    identity_block(H, 4, ORI + 4, ORI);
    kf_update(e, 4, ORI + 4, H, q, r * e->noiseScale);
    orc_ekf_normalize_quaternions(e, 0);
    orc_ekf_maintain_psd(e);
}

void orc_ekf_get_inertial_state(const orc_ekf *e, double mean[INER_DIM], double cov[INER_DIM * INER_DIM])   /* ekf.cpp:679-682 */
{
    memcpy(mean, e->m, sizeof(double) * INER_DIM);
    for (int j = 0; j < INER_DIM; j++) for (int i = 0; i < INER_DIM; i++) M_(cov, INER_DIM, i, j) = M_(e->P, e->n, i, j);
}

void orc_ekf_set_inertial_state(orc_ekf *e, const double mean[INER_DIM], const double cov[INER_DIM * INER_DIM])   /* ekf.cpp:684-690 */
{
    memcpy(e->m, mean, sizeof(double) * INER_DIM);
    for (int j = 0; j < INER_DIM; j++) for (int i = 0; i < INER_DIM; i++) M_(e->P, e->n, i, j) = M_(cov, INER_DIM, i, j);
    e->augmentCount = 0;
}

void orc_ekf_translate_to(orc_ekf *e, const double pos[3])   /* ekf.cpp:696-702 */
{
    const double d[3] = { pos[0] - e->m[POS], pos[1] - e->m[POS + 1], pos[2] - e->m[POS + 2] };
    for (int i = 0; i < 3; i++) e->m[POS + i] += d[i];
    for (int c = 0; c < e->camPoseCount; c++) for (int i = 0; i < 3; i++) e->m[CAM + POSE_DIM * c + i] += d[i];
}

void orc_ekf_transform_to(orc_ekf *e, const double pos[3], const double q[4], int idx)   /* ekf.cpp:704-758 */
{
    const int n = e->n;
    const double *q0 = idx < 0 ? e->m + ORI : e->m + CAM + POSE_DIM * idx + 3;
    /* qChange = conj(quat0) * quat1 (Hamilton product, w first) */
    const double a[4] = { q0[0], -q0[1], -q0[2], -q0[3] };
    const double p1 = a[0]*q[0] - a[1]*q[1] - a[2]*q[2] - a[3]*q[3];
    const double p2 = a[0]*q[1] + a[1]*q[0] + a[2]*q[3] - a[3]*q[2];
    const double p3 = a[0]*q[2] - a[1]*q[3] + a[2]*q[0] + a[3]*q[1];
    const double p4 = a[0]*q[3] + a[1]*q[2] - a[2]*q[1] + a[3]*q[0];
    const double qrow[16] = { p1, -p2, -p3, -p4,  p2, p1, p4, -p3,  p3, -p4, p1, p2,  p4, p3, -p2, p1 };
    /* pChangeMat = qChange.toRotationMatrix()' ; Eigen's toRotationMatrix of (w,x,y,z) */
    double Rq[9];
    {
        const double w = p1, x = p2, y = p3, z = p4;
        const double tx = 2*x, ty = 2*y, tz = 2*z, twx = tx*w, twy = ty*w, twz = tz*w, txx = tx*x, txy = ty*x, txz = tz*x, tyy = ty*y, tyz = tz*y, tzz = tz*z;
        M_(Rq, 3, 0, 0) = 1 - (tyy + tzz); M_(Rq, 3, 0, 1) = txy - twz; M_(Rq, 3, 0, 2) = txz + twy;
        M_(Rq, 3, 1, 0) = txy + twz; M_(Rq, 3, 1, 1) = 1 - (txx + tzz); M_(Rq, 3, 1, 2) = tyz - twx;
        M_(Rq, 3, 2, 0) = txz - twy; M_(Rq, 3, 2, 1) = tyz + twx; M_(Rq, 3, 2, 2) = 1 - (txx + tyy);
    }
    double pC[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M_(pC, 3, i, j) = M_(Rq, 3, j, i);
    const double *refPos = idx < 0 ? e->m + POS : e->m + CAM + POSE_DIM * idx;
    double translation[3];
    for (int i = 0; i < 3; i++) { double s = 0; for (int j = 0; j < 3; j++) s += M_(pC, 3, i, j) * refPos[j]; translation[i] = pos[i] - s; }

    double *T = e->tmp1;   /* trailRotationA */
    memset(T, 0, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) M_(T, n, i, i) = 1.0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { M_(T, n, POS + i, POS + j) = M_(pC, 3, i, j); M_(T, n, VEL + i, VEL + j) = M_(pC, 3, i, j); }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) M_(T, n, ORI + i, ORI + j) = qrow[4 * i + j];
    for (int c = 0; c < e->camPoseCount; c++) {
        const int o = CAM + c * POSE_DIM;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M_(T, n, o + i, o + j) = M_(pC, 3, i, j);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) M_(T, n, o + 3 + i, o + 3 + j) = qrow[4 * i + j];
    }
    gemm(0, 0, n, 1, n, 1.0, T, n, e->m, n, 0.0, e->tmp0, n);
    memcpy(e->m, e->tmp0, sizeof(double) * n);
    gemm(0, 1, n, n, n, 1.0, e->P, n, T, n, 0.0, e->tmp0, n);      /* P * A' */
    gemm(0, 0, n, n, n, 1.0, T, n, e->tmp0, n, 0.0, e->P, n);      /* A * (P A') */
    const double target[3] = { e->m[POS] + translation[0], e->m[POS + 1] + translation[1], e->m[POS + 2] + translation[2] };
    orc_ekf_translate_to(e, target);
}

/* ---- visual updates (ekf.cpp:760-844): H is nr x l column-major ---- */
static void visual_common(orc_ekf *e, int nr, int l, const double *H, double r)
{
    const int n = e->n;
    gemm(0, 0, nr, n, l, 1.0, H, nr, e->P, n, 0.0, e->HP, nr);
    gemm(0, 1, nr, nr, l, 1.0, e->HP, nr, H, nr, 0.0, e->S, nr);
    const double rd = (r * r) * e->noiseScale;
    for (int i = 0; i < nr; i++) M_(e->S, nr, i, i) += rd;
    ldlt_factor(nr, e->S, nr, e->perm);
}

int orc_ekf_visual_track_outlier_check(orc_ekf *e, int nr, int l, const double *H, const double *f,
                                       const double *y, double r, double trackRmseThreshold, double *chi2_out)
{
    double v[512], x[512];
    for (int i = 0; i < nr; i++) v[i] = y[i] - f[i];
    if (chi2_out) *chi2_out = -1.0;
    if (trackRmseThreshold >= 0.0) {
        double s = 0; for (int i = 0; i < nr; i++) s += v[i] * v[i];
        if (sqrt(s / nr) > trackRmseThreshold) return VU_RMSE;
    }
    if (r < 0.0) return VU_INLIER;
    visual_common(e, nr, l, H, r);
    memcpy(x, v, sizeof(double) * nr);
    ldlt_solve(nr, e->S, nr, e->perm, x, nr, 1);
    double t = 0; for (int i = 0; i < nr; i++) t += x[i] * v[i];
    t *= e->noiseScale;
    if (chi2_out) *chi2_out = t;
    if (t > hv_chi2inv95[nr]) return VU_CHI2;
    return VU_INLIER;
}

void orc_ekf_update_visual_track(orc_ekf *e, int nr, int l, const double *H, const double *f, const double *y, double r)
{
    const int n = e->n;
    visual_common(e, nr, l, H, r);
    double *X = e->tmp0, *K = e->K;
    memcpy(X, e->HP, sizeof(double) * nr * n);
    ldlt_solve(nr, e->S, nr, e->perm, X, nr, n);
    for (int i = 0; i < n; i++) for (int j = 0; j < nr; j++) M_(K, n, i, j) = M_(X, nr, j, i);
    for (int i = 0; i < n; i++) { double s = 0; for (int j = 0; j < nr; j++) s += M_(K, n, i, j) * (y[j] - f[j]); e->m[i] += s; }
    gemm(0, 0, n, n, nr, -1.0, K, n, e->HP, nr, 1.0, e->P, n);
    orc_ekf_normalize_quaternions(e, 0);
}

/* The shift matrix visAugA[k] (ekf.cpp:230-248) as a source-index map: row i of A*m takes m[src[i]]
 * (src = -1: zero row). */
static void aug_src(const orc_ekf *e, int dropped, int *src)
{
    const int n = e->n;
    for (int i = 0; i < n; i++) src[i] = -1;
    for (int i = 0; i < CAM; i++) src[i] = i;
    for (int i = CAM; i < CAM + dropped * POSE_DIM; i++) src[i + POSE_DIM] = i;
    for (int i = CAM + (dropped + 1) * POSE_DIM; i < n; i++) src[i] = i;
}

static void apply_shift(orc_ekf *e, const int *src)   /* m = A m ; P = A P A' */
{
    const int n = e->n;
    for (int i = 0; i < n; i++) e->tmp0[i] = src[i] >= 0 ? e->m[src[i]] : 0.0;
    memcpy(e->m, e->tmp0, sizeof(double) * n);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++)
            M_(e->tmp0, n, i, j) = (src[i] >= 0 && src[j] >= 0) ? M_(e->P, n, src[i], src[j]) : 0.0;
    memcpy(e->P, e->tmp0, sizeof(double) * n * n);
}

void orc_ekf_update_visual_pose_augmentation(orc_ekf *e, int discarded)   /* ekf.cpp:848-885 */
{
    const int n = e->n;
    if (discarded == -1) discarded = e->camPoseCount - 1;
    int *src = (int *)malloc(sizeof(int) * n);
    aug_src(e, discarded, src);
    apply_shift(e, src);
    free(src);
    const double noisePos = pow2(e->par.noiseInitialPosTrail) * e->noiseScale, noiseOri = pow2(e->par.noiseInitialOriTrail) * e->noiseScale;
    for (int i = CAM; i < CAM + 3; i++) M_(e->P, n, i, i) += noisePos;          /* + visAugQ */
    for (int i = CAM + 3; i < CAM + POSE_DIM; i++) M_(e->P, n, i, i) += noiseOri;

    /* visAugH (ekf.cpp:267-278): 7 x n with +1 at (i, POS+i)/(3+i, ORI+i), -1 at (i, CAM+i) */
    double *H = e->tmp1;
    memset(H, 0, sizeof(double) * POSE_DIM * n);
    for (int i = 0; i < 3; i++) { M_(H, POSE_DIM, i, POS + i) = 1; M_(H, POSE_DIM, i, CAM + i) = -1; }
    for (int i = 0; i < 4; i++) { M_(H, POSE_DIM, 3 + i, ORI + i) = 1; M_(H, POSE_DIM, 3 + i, CAM + 3 + i) = -1; }
    const double rd = e->par.augmentR * e->noiseScale;
    double *HP = e->HP, *S = e->S, *K = e->K;
    gemm(0, 0, POSE_DIM, n, n, 1.0, H, POSE_DIM, e->P, n, 0.0, HP, POSE_DIM);
    gemm(0, 1, POSE_DIM, POSE_DIM, n, 1.0, HP, POSE_DIM, H, POSE_DIM, 0.0, S, POSE_DIM);
    for (int i = 0; i < POSE_DIM; i++) M_(S, POSE_DIM, i, i) += rd;
    ldlt_factor(POSE_DIM, S, POSE_DIM, e->perm);
    double *X = (double *)malloc(sizeof(double) * POSE_DIM * n);
    memcpy(X, HP, sizeof(double) * POSE_DIM * n);
    ldlt_solve(POSE_DIM, S, POSE_DIM, e->perm, X, POSE_DIM, n);
    for (int i = 0; i < n; i++) for (int j = 0; j < POSE_DIM; j++) M_(K, n, i, j) = M_(X, POSE_DIM, j, i);
    free(X);
    double v[POSE_DIM];
    for (int i = 0; i < POSE_DIM; i++) { double s = 0; for (int j = 0; j < n; j++) s += M_(H, POSE_DIM, i, j) * e->m[j]; v[i] = -s; }
    for (int i = 0; i < n; i++) { double s = 0; for (int j = 0; j < POSE_DIM; j++) s += M_(K, n, i, j) * v[j]; e->m[i] += s; }

    /* Joseph form (ekf.cpp:35-50): T = I - K H; P = T P T' + K R K' -- two dense n^3 products */
    double *T = (double *)malloc(sizeof(double) * n * n), *TP = e->tmp0;
    gemm(0, 0, n, n, POSE_DIM, -1.0, K, n, H, POSE_DIM, 0.0, T, n);
    for (int i = 0; i < n; i++) M_(T, n, i, i) += 1.0;
    gemm(0, 0, n, n, n, 1.0, T, n, e->P, n, 0.0, TP, n);
    gemm(0, 1, n, n, n, 1.0, TP, n, T, n, 0.0, e->P, n);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            double s = 0; for (int k = 0; k < POSE_DIM; k++) s += M_(K, n, i, k) * rd * M_(K, n, j, k);
            M_(e->P, n, i, j) += s;
        }
    free(T);
    orc_ekf_maintain_psd(e);
    orc_ekf_normalize_quaternions(e, 0);

    e->augmentTimes[e->augmentCount] = orc_ekf_platform_time(e);   /* push_back */
    if (e->augmentCount < e->camPoseCount) e->augmentCount++;
    else memmove(e->augmentTimes, e->augmentTimes + 1, sizeof(double) * e->camPoseCount);   /* erase front */
}

void orc_ekf_update_undo_augmentation(orc_ekf *e)   /* ekf.cpp:888-903, visUnaugmentA ekf.cpp:251-265 */
{
    const int n = e->n, poseTrailDim = n - e->hybridMapDim;
    int *src = (int *)malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) src[i] = -1;
    for (int i = 0; i < CAM; i++) src[i] = i;
    for (int i = CAM; i + POSE_DIM < poseTrailDim; i++) src[i] = i + POSE_DIM;
    for (int i = poseTrailDim; i < n; i++) src[i] = i;
    apply_shift(e, src);
    free(src);
    e->augmentCount--;
}

void orc_ekf_condition_on_last_pose(orc_ekf *e)   /* ekf.cpp:928-942 */
{
    const int n = e->n, mm = n - POSE_DIM;
    double Bm[POSE_DIM * POSE_DIM]; int perm[POSE_DIM];
    for (int j = 0; j < POSE_DIM; j++) for (int i = 0; i < POSE_DIM; i++) M_(Bm, POSE_DIM, i, j) = M_(e->P, n, mm + i, mm + j);
    ldlt_factor(POSE_DIM, Bm, POSE_DIM, perm);
    double *X = (double *)malloc(sizeof(double) * POSE_DIM * mm);   /* X = B^-1 * P(m.., 0..m) */
    for (int j = 0; j < mm; j++) for (int i = 0; i < POSE_DIM; i++) M_(X, POSE_DIM, i, j) = M_(e->P, n, mm + i, j);
    ldlt_solve(POSE_DIM, Bm, POSE_DIM, perm, X, POSE_DIM, mm);
    for (int j = 0; j < mm; j++)
        for (int i = 0; i < mm; i++) {
            double s = 0; for (int k = 0; k < POSE_DIM; k++) s += M_(e->P, n, i, mm + k) * M_(X, POSE_DIM, k, j);
            M_(e->tmp0, mm, i, j) = M_(e->P, n, i, j) - s;
        }
    for (int j = 0; j < mm; j++) for (int i = 0; i < mm; i++) M_(e->P, n, i, j) = M_(e->tmp0, mm, i, j);
    free(X);
    for (int k = 0; k < POSE_DIM; k++) for (int i = 0; i < mm; i++) { M_(e->P, n, i, mm + k) = 0; M_(e->P, n, mm + k, i) = 0; }
    for (int j = 0; j < POSE_DIM; j++) for (int i = 0; i < POSE_DIM; i++) M_(e->P, n, mm + i, mm + j) = (i == j) ? 1e6 : 0.0;
}

void orc_ekf_lock_biases(orc_ekf *e)   /* ekf.cpp:944-947 */
{
    const int n = e->n;
    for (int k = BGA; k < BGA + 9; k++) for (int i = 0; i < n; i++) { M_(e->P, n, k, i) = 0; M_(e->P, n, i, k) = 0; }
}

int orc_ekf_map_point_state_index(const orc_ekf *e, int idx)   /* ekf.cpp:923-926 */
{
    if (idx == -1) return -1;
    return e->n - e->hybridMapDim + idx * MAP_POINT_DIM;
}

void orc_ekf_insert_map_point(orc_ekf *e, int idx, const double pf[3])   /* ekf.cpp:911-921 */
{
    const int n = e->n, off = orc_ekf_map_point_state_index(e, idx);
    for (int k = 0; k < MAP_POINT_DIM; k++) for (int i = 0; i < n; i++) { M_(e->P, n, off + k, i) = 0; M_(e->P, n, i, off + k) = 0; }
    for (int k = 0; k < MAP_POINT_DIM; k++) { M_(e->P, n, off + k, off + k) = 1e6; e->m[off + k] = pf[k]; }
}
