/*
 * CPU oracle for SURVEY.md section 8(f) row f3: per-track triangulation and prepareVisualUpdate (the code that
 * builds the (H, f) handed to EKF::visualTrackOutlierCheck / updateVisualTrack).
 *
 * TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file; the product path (hybvio_amd/csrc, hybvio_amd/host) never does.
 *
 * Restated from the reference (all 3x3 matrices here are row-major double[9]):
 *   extractCameraPoseTrail        src/odometry/triangulation.cpp:65-103
 *   Triangulator::triangulate     src/odometry/triangulation.cpp:120-407   (iterative PIVO method, the default:
 *                                 useLinearTriangulation = false, useIndependentStereoTriangulation = false)
 *   triangulateLinear             src/odometry/triangulation.cpp:820-895   (useLinearTriangulation = true, :146-152; the
 *                                 reference's own test leaves this case as a TODO, test/triangulation.cpp:107: it is pinned here
 *                                 the way that test pins the default, by the numeric-vs-analytic derivative check on its data)
 *   triangulateWithTwoCameras     src/odometry/triangulation.cpp:612-716, dpinv :31-52, pinv :1000-1002
 *   inverseDepth                  src/odometry/triangulation.cpp:1004-1029
 *   prepareVisualUpdate           src/odometry/triangulation.cpp:897-987, getPosOriIndices :989-998
 *   the caller's glue             src/odometry/backend.cpp:1063-1148 (depth test, stereo derivative sum)
 *   quat2rmat_d                   src/odometry/util.cpp:10-47
 *
 * PINNED by the reference's own tests, re-expressed in tests/test_oracle_triangulation.py on the reference's
 * inline data: test/triangulation.cpp:56-246 ("visual": Matlab point pf_e, derivative checks of triangulate and of
 * prepareVisualUpdate), :248-474 ("stereo_visual"), :477-485 ("pinv" Matlab values), :487-519
 * ("triangulateWithTwoCameras"), :521-580 ("der_triangulateWithTwoCameras"), :582-601 ("der_inverseDepth").
 *
 * Two Eigen calls have no in-tree definition and are restated by their mathematical meaning (differences are at the
 * 1e-16 level, parity tolerance for this row is relative 1e-9):
 *   pinv(A) for the full-rank 3x2 A   = (A^T A)^-1 A^T            (Eigen: completeOrthogonalDecomposition)
 *   ETE.ldlt().solve / .rcond()       = exact 3x3 symmetric solve; rcond = 1 / (|A|_1 |A^-1|_1) (Eigen estimates
 *                                       |A^-1|_1 with Hager's method, exact or a slight under-estimate for 3x3;
 *                                       it is only compared with triangulationRcondThreshold = 1e-8)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { POS = 0, ORI = 6, SFT = 19, CAM = 20, POSE_DIM = 7 };
enum { TRI_OK = 0, TRI_HYBRID, TRI_BEHIND, TRI_BAD_COND, TRI_NO_CONVERGENCE, TRI_BAD_DEPTH, TRI_UNKNOWN_PROBLEM };
enum { PREPARE_VU_OK = 0, PREPARE_VU_ZERO_DEPTH = 1, PREPARE_VU_BEHIND = 2 };

typedef struct orc_campose {
    double p[3], R[9], dR[4][9], baseline[3];
} orc_campose;

typedef struct orc_tri_params {            /* codegen/parameter_definitions.c:37-44,163 */
    double triangulationConvergenceThreshold;      /* 1e-2 */
    double triangulationConvergenceR;              /* 11.0 */
    double triangulationRcondThreshold;            /* 1e-8 */
    unsigned triangulationGaussNewtonIterations;   /* 10 */
    double triangulationMinDist, triangulationMaxDist;   /* 0, 1e300 */
    int estimateImuCameraTimeShift;                /* true */
    int useLinearTriangulation;                    /* false (parameter_definitions.c:31) */
} orc_tri_params;

void orc_tri_default_params(orc_tri_params *p)
{
    p->triangulationConvergenceThreshold = 1e-2; p->triangulationConvergenceR = 11.0;
    p->triangulationRcondThreshold = 1e-8; p->triangulationGaussNewtonIterations = 10;
    p->triangulationMinDist = 0; p->triangulationMaxDist = 1e300; p->estimateImuCameraTimeShift = 1;
    p->useLinearTriangulation = 0;
}

/* ---- small dense helpers (row-major) ---- */
static void mm3(const double *A, const double *B, double *C)            /* C = A B */
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
static void mmT3(const double *A, const double *B, double *C)           /* C = A B^T */
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[3 * c] + A[3 * r + 1] * B[3 * c + 1] + A[3 * r + 2] * B[3 * c + 2];
}
static void mv3(const double *A, const double *x, double *y) { for (int r = 0; r < 3; ++r) y[r] = A[3 * r] * x[0] + A[3 * r + 1] * x[1] + A[3 * r + 2] * x[2]; }
static void mTv3(const double *A, const double *x, double *y) { for (int c = 0; c < 3; ++c) y[c] = A[c] * x[0] + A[3 + c] * x[1] + A[6 + c] * x[2]; }

static void quat2rmat_d(const double *q, double *R, double dR[4][9])     /* util.cpp:10-47 */
{
    const double r[9] = {q[0] * q[0] + q[1] * q[1] - q[2] * q[2] - q[3] * q[3], 2 * q[1] * q[2] - 2 * q[0] * q[3], 2 * q[1] * q[3] + 2 * q[0] * q[2],
                         2 * q[1] * q[2] + 2 * q[0] * q[3], q[0] * q[0] - q[1] * q[1] + q[2] * q[2] - q[3] * q[3], 2 * q[2] * q[3] - 2 * q[0] * q[1],
                         2 * q[1] * q[3] - 2 * q[0] * q[2], 2 * q[2] * q[3] + 2 * q[0] * q[1], q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3]};
    memcpy(R, r, sizeof r);
    const double d0[9] = {2 * q[0], -2 * q[3], 2 * q[2], 2 * q[3], 2 * q[0], -2 * q[1], -2 * q[2], 2 * q[1], 2 * q[0]};
    const double d1[9] = {2 * q[1], 2 * q[2], 2 * q[3], 2 * q[2], -2 * q[1], -2 * q[0], 2 * q[3], 2 * q[0], -2 * q[1]};
    const double d2[9] = {-2 * q[2], 2 * q[1], 2 * q[0], 2 * q[1], 2 * q[2], 2 * q[3], -2 * q[0], 2 * q[3], -2 * q[2]};
    const double d3[9] = {-2 * q[3], -2 * q[0], 2 * q[1], 2 * q[0], -2 * q[3], 2 * q[2], 2 * q[1], 2 * q[2], 2 * q[3]};
    memcpy(dR[0], d0, sizeof d0); memcpy(dR[1], d1, sizeof d1); memcpy(dR[2], d2, sizeof d2); memcpy(dR[3], d3, sizeof d3);
}

void orc_get_pos_ori_indices(int i, int *pos, int *ori)                  /* triangulation.cpp:989-998 */
{
    if (i == 0) { *pos = POS; *ori = ORI; }
    else { *pos = CAM + 7 * (i - 1); *ori = CAM + 7 * (i - 1) + 3; }
}

/* triangulation.cpp:65-103. imu_to_cam: 4x4 row-major; imu_to_cam2 NULL = mono. trail: n (mono) or 2n poses */
void orc_extract_camera_pose_trail(const double *m, const int *pose_trail_index, int n, const double *imu_to_cam,
                                   const double *imu_to_cam2, orc_campose *trail)
{
    const int ncam = imu_to_cam2 ? 2 : 1;
    for (int cam = 0; cam < ncam; ++cam) {
        const double *T = cam == 0 ? imu_to_cam : imu_to_cam2;
        const double Ric[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
        const double base[3] = {T[3], T[7], T[11]};
        for (int k = 0; k < n; ++k) {
            int ip, io;
            orc_get_pos_ori_indices(pose_trail_index[k], &ip, &io);     /* historyPosition(i - 1): ekf.cpp:544-554 */
            orc_campose *o = &trail[cam * n + k];
            double Rw[9], dRw[4][9], t[3];
            quat2rmat_d(m + io, Rw, dRw);
            mm3(Ric, Rw, o->R);
            mTv3(o->R, base, t);
            for (int a = 0; a < 3; ++a) { o->p[a] = m[ip + a] - t[a]; o->baseline[a] = base[a]; }
            for (int j = 0; j < 4; ++j) mm3(Ric, dRw[j], o->dR[j]);
        }
    }
}

/* triangulation.cpp:1004-1012 (first derivative only) */
void orc_inverse_depth(const double *p, double *ip, double *dip)
{
    ip[0] = p[0] / p[2]; ip[1] = p[1] / p[2]; ip[2] = 1 / p[2];
    const double d[9] = {1 / p[2], 0, -ip[0] / p[2], 0, 1 / p[2], -ip[1] / p[2], 0, 0, -ip[2] / p[2]};
    memcpy(dip, d, sizeof d);
}

/* pinv of a full-rank 3x2 matrix A (row-major 3x2) -> iA (row-major 2x3); triangulation.cpp:1000-1002 */
void orc_pinv32(const double *A, double *iA)
{
    const double a = A[0] * A[0] + A[2] * A[2] + A[4] * A[4], b = A[0] * A[1] + A[2] * A[3] + A[4] * A[5];
    const double d = A[1] * A[1] + A[3] * A[3] + A[5] * A[5], det = a * d - b * b;
    for (int c = 0; c < 3; ++c) {
        iA[c] = (d * A[2 * c] - b * A[2 * c + 1]) / det;
        iA[3 + c] = (-b * A[2 * c] + a * A[2 * c + 1]) / det;
    }
}

/* triangulation.cpp:31-52: derivative of the pseudo-inverse (Golub & Pereyra 4.12); A, dA 3x2, iA, out 2x3 */
static void dpinv(const double *A, const double *iA, const double *dA, double *out)
{
    double iAiAT[4], iATiA[9], AiA[9], iAA[4], t23[6], t23b[6];
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) {
        iAiAT[2 * r + c] = iA[3 * r] * iA[3 * c] + iA[3 * r + 1] * iA[3 * c + 1] + iA[3 * r + 2] * iA[3 * c + 2];
        iAA[2 * r + c] = iA[3 * r] * A[c] + iA[3 * r + 1] * A[2 + c] + iA[3 * r + 2] * A[4 + c];
    }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        iATiA[3 * r + c] = iA[r] * iA[c] + iA[3 + r] * iA[3 + c];
        AiA[3 * r + c] = A[2 * r] * iA[c] + A[2 * r + 1] * iA[3 + c];
    }
    /* term1 = -iA dA iA */
    double iAdA[4];
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) iAdA[2 * r + c] = iA[3 * r] * dA[c] + iA[3 * r + 1] * dA[2 + c] + iA[3 * r + 2] * dA[4 + c];
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) out[3 * r + c] = -(iAdA[2 * r] * iA[c] + iAdA[2 * r + 1] * iA[3 + c]);
    /* term2 = (iA iA^T) dA^T (I3 - A iA) */
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) t23[3 * r + c] = iAiAT[2 * r] * dA[2 * c] + iAiAT[2 * r + 1] * dA[2 * c + 1];
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += t23[3 * r + k] * ((k == c ? 1.0 : 0.0) - AiA[3 * k + c]);
        out[3 * r + c] += s;
    }
    /* term3 = (I2 - iA A) dA^T (iA^T iA) */
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c)
        t23b[3 * r + c] = ((r == 0 ? 1.0 : 0.0) - iAA[2 * r]) * dA[2 * c] + ((r == 1 ? 1.0 : 0.0) - iAA[2 * r + 1]) * dA[2 * c + 1];
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += t23b[3 * r + k] * iATiA[3 * k + c];
        out[3 * r + c] += s;
    }
}

/* triangulation.cpp:612-716. dpf: 3 x 15 row-major (p0 q0 p1 q1 t), may be NULL when !calc_derivatives */
void orc_triangulate_with_two_cameras(const orc_campose *pose0, const orc_campose *pose1, const double *ip0, const double *ip1,
                                      const double *vel0, const double *vel1, int calc_derivatives, int estimate_time_shift,
                                      int derivative_test, double time_shift, double *pf, double *dpf)
{
    const double *R0 = pose0->R, *R1 = pose1->R;
    double C[9], d01[3], b[3];
    mmT3(R0, R1, C);
    for (int a = 0; a < 3; ++a) d01[a] = pose1->p[a] - pose0->p[a];
    mv3(R0, d01, b);
    double v0[3] = {ip0[0], ip0[1], 1.0}, v1[3] = {ip1[0], ip1[1], 1.0};
    if (derivative_test && estimate_time_shift) {
        v0[0] += time_shift * vel0[0]; v0[1] += time_shift * vel0[1];
        v1[0] += time_shift * vel1[0]; v1[1] += time_shift * vel1[1];
    }
    const double n0 = sqrt(v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2]), n1 = sqrt(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]);
    const double vn0[3] = {v0[0] / n0, v0[1] / n0, v0[2] / n0}, vn1[3] = {v1[0] / n1, v1[1] / n1, v1[2] / n1};
    double Cvn1[3], A[6], iA[6];
    mv3(C, vn1, Cvn1);
    for (int r = 0; r < 3; ++r) { A[2 * r] = vn0[r]; A[2 * r + 1] = -Cvn1[r]; }
    orc_pinv32(A, iA);
    const double s0 = iA[0] * b[0] + iA[1] * b[1] + iA[2] * b[2];
    for (int a = 0; a < 3; ++a) pf[a] = s0 * vn0[a];
    if (!calc_derivatives) return;
    double dA[15][6], db[15][3];
    memset(dA, 0, sizeof dA); memset(db, 0, sizeof db);
    for (int i = 0; i < 4; ++i) {
        double dC0[9], dC1[9], t[3];
        mmT3(pose0->dR[i], R1, dC0);
        mmT3(R0, pose1->dR[i], dC1);
        mv3(dC0, vn1, t); for (int r = 0; r < 3; ++r) dA[3 + i][2 * r + 1] = -t[r];
        mv3(dC1, vn1, t); for (int r = 0; r < 3; ++r) dA[10 + i][2 * r + 1] = -t[r];
        mv3(pose0->dR[i], d01, db[3 + i]);
        if (i < 3) for (int r = 0; r < 3; ++r) { db[i][r] = -R0[3 * r + i]; db[7 + i][r] = R0[3 * r + i]; }
    }
    for (int i = 0; i < 14; ++i) {
        double diA[6];
        dpinv(A, iA, dA[i], diA);
        const double ds = (iA[0] * db[i][0] + iA[1] * db[i][1] + iA[2] * db[i][2]) + (diA[0] * b[0] + diA[1] * b[1] + diA[2] * b[2]);
        for (int r = 0; r < 3; ++r) dpf[15 * r + i] = ds * vn0[r];
    }
    if (estimate_time_shift) {
        double B0[9], B1[9], w0[3], w1[3], cw1[3], diA[6];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
            B0[3 * r + c] = ((r == c ? 1.0 : 0.0) - vn0[r] * vn0[c]) / n0;
            B1[3 * r + c] = ((r == c ? 1.0 : 0.0) - vn1[r] * vn1[c]) / n1;
        }
        const double u0[3] = {vel0[0], vel0[1], 0.0}, u1[3] = {vel1[0], vel1[1], 0.0};
        mv3(B0, u0, w0); mv3(B1, u1, w1); mv3(C, w1, cw1);
        for (int r = 0; r < 3; ++r) { dA[14][2 * r] = w0[r]; dA[14][2 * r + 1] = -cw1[r]; }
        dpinv(A, iA, dA[14], diA);
        const double ds0dt = diA[0] * b[0] + diA[1] * b[1] + diA[2] * b[2];
        for (int r = 0; r < 3; ++r) dpf[15 * r + 14] = s0 * w0[r] + vn0[r] * ds0dt;
    } else {
        for (int r = 0; r < 3; ++r) dpf[15 * r + 14] = 0.0;
    }
}

/* exact solve of the symmetric 3x3 system through the adjugate; returns det */
static double inv3sym(const double *M, double *inv)
{
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
    inv[0] = c00 / det; inv[1] = (M[2] * M[7] - M[1] * M[8]) / det; inv[2] = (M[1] * M[5] - M[2] * M[4]) / det;
    inv[3] = c01 / det; inv[4] = (M[0] * M[8] - M[2] * M[6]) / det; inv[5] = (M[2] * M[3] - M[0] * M[5]) / det;
    inv[6] = c02 / det; inv[7] = (M[1] * M[6] - M[0] * M[7]) / det; inv[8] = (M[0] * M[4] - M[1] * M[3]) / det;
    return det;
}
static double norm1_3(const double *M)
{
    double best = 0;
    for (int c = 0; c < 3; ++c) { const double s = fabs(M[c]) + fabs(M[3 + c]) + fabs(M[6 + c]); if (s > best) best = s; }
    return best;
}


/* 3x3 inverse through the cofactors of the first column (Eigen's fixed-size Matrix3d::inverse()) */
static void inv3(const double *M, double *inv)
{
    const double c0 = M[4] * M[8] - M[5] * M[7], c1 = M[2] * M[7] - M[1] * M[8], c2 = M[1] * M[5] - M[2] * M[4];
    const double invdet = 1.0 / (c0 * M[0] + c1 * M[3] + c2 * M[6]);
    inv[0] = c0 * invdet; inv[1] = c1 * invdet; inv[2] = c2 * invdet;
    inv[3] = (M[5] * M[6] - M[3] * M[8]) * invdet; inv[4] = (M[0] * M[8] - M[2] * M[6]) * invdet; inv[5] = (M[2] * M[3] - M[0] * M[5]) * invdet;
    inv[6] = (M[3] * M[7] - M[4] * M[6]) * invdet; inv[7] = (M[1] * M[6] - M[0] * M[7]) * invdet; inv[8] = (M[0] * M[4] - M[1] * M[3]) * invdet;
}

/* triangulateLinear (triangulation.cpp:820-895): the point closest to all camera rays, in closed form */
static int triangulate_linear(const orc_tri_params *par, int pose_count, const orc_campose *trail, const double *image_features,
                              const double *feature_velocities, int calc_derivatives, int derivative_test, double time_shift,
                              double *pf, double *dpfdp, double *dpfdq, double *dpfdt)
{
    double S0[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, S1[3] = {0, 0, 0}, S0inv[9];
    for (int i = 0; i < pose_count; ++i) {                                         /* :830-842 */
        double ip[3] = {image_features[2 * i], image_features[2 * i + 1], 1.0}, v[3], A[9], Ap[3];
        if (derivative_test) { ip[0] += time_shift * feature_velocities[2 * i]; ip[1] += time_shift * feature_velocities[2 * i + 1]; }
        mTv3(trail[i].R, ip, v);
        const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const double vn[3] = {v[0] / n, v[1] / n, v[2] / n};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[3 * r + c] = (r == c ? 1.0 : 0.0) - vn[r] * vn[c];
        mv3(A, trail[i].p, Ap);
        for (int k = 0; k < 9; ++k) S0[k] += A[k];
        for (int k = 0; k < 3; ++k) S1[k] += Ap[k];
    }
    inv3(S0, S0inv);
    mv3(S0inv, S1, pf);
    if (calc_derivatives) {                                                        /* :846-890 */
        dpfdt[0] = dpfdt[1] = dpfdt[2] = 0.0;
        for (int i = 0; i < pose_count; ++i) {
            double ip[3] = {image_features[2 * i], image_features[2 * i + 1], 1.0}, v[3], A[9];
            if (derivative_test) { ip[0] += time_shift * feature_velocities[2 * i]; ip[1] += time_shift * feature_velocities[2 * i + 1]; }
            mTv3(trail[i].R, ip, v);
            const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            const double vn[3] = {v[0] / n, v[1] / n, v[2] / n};
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[3 * r + c] = (r == c ? 1.0 : 0.0) - vn[r] * vn[c];
            mm3(S0inv, A, dpfdp + 9 * i);                                          /* dpfdp = S0inv * A */
            double dvdq[12];                                                       /* 3 x 4 row-major: column k = dR[k]^T ip */
            for (int k = 0; k < 4; ++k) { double t[3]; mTv3(trail[i].dR[k], ip, t); for (int r = 0; r < 3; ++r) dvdq[4 * r + k] = t[r]; }
            double dvndv[9];
            for (int k = 0; k < 9; ++k) dvndv[k] = A[k] / n;
            double dpfdvn[9];                                                      /* column k = S0inv Q S0inv S1 - S0inv Q p */
            for (int k = 0; k < 3; ++k) {
                double Q[9], SQ[9], t0[3], t1[3], t2[3];
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Q[3 * r + c] = (r == k ? vn[c] : 0.0) + (c == k ? vn[r] : 0.0);
                mm3(S0inv, Q, SQ);
                mv3(S0inv, S1, t0);
                mv3(SQ, t0, t1);
                mv3(SQ, trail[i].p, t2);
                for (int r = 0; r < 3; ++r) dpfdvn[3 * r + k] = t1[r] - t2[r];
            }
            double G[9];
            mm3(dpfdvn, dvndv, G);
            for (int r = 0; r < 3; ++r) for (int k = 0; k < 4; ++k)
                dpfdq[12 * i + 4 * r + k] = G[3 * r] * dvdq[k] + G[3 * r + 1] * dvdq[4 + k] + G[3 * r + 2] * dvdq[8 + k];
            if (par->estimateImuCameraTimeShift) {
                const double vel[3] = {feature_velocities[2 * i], feature_velocities[2 * i + 1], 0.0};
                double dvdt[3], t[3];
                mTv3(trail[i].R, vel, dvdt);
                mv3(G, dvdt, t);
                for (int r = 0; r < 3; ++r) dpfdt[r] += t[r];
            }
        }
    }
    for (int i = 0; i < pose_count; ++i) {                                          /* isBehind, :54-60 */
        double dd[3], a[3];
        for (int k = 0; k < 3; ++k) dd[k] = pf[k] - trail[i].p[k];
        mv3(trail[i].R, dd, a);
        if (a[2] < 0) return TRI_BEHIND;
    }
    return TRI_OK;
}

/* Triangulator::triangulate, iterative branch (triangulation.cpp:120-407).
 * image_features / feature_velocities: [pose_count][2]; outputs dpfdp [pose_count][9], dpfdq [pose_count][12] (3x4
 * row-major), dpfdt[3] are written when calc_derivatives and the status is not an early return. */
/* Test diagnostic (not part of the reference): how close the LAST orc_triangulate call of this thread came to a singular
 * Gauss-Newton step -- the smallest rcond of E'E over the iterations, the smallest |h_z| (the projective denominator of a pose)
 * relative to |h|, the iterations run and whether the loop converged. A track that drives either minimum to rounding level has no
 * well-defined failure status: which of BEHIND / BAD_COND / NO_CONVERGENCE comes out follows the last bit of every sum (tests use
 * this to say WHEN a device status may differ from the oracle's; see tests/test_gpu_visual_prepare.py). */
static _Thread_local double g_tri_diag[4];
void orc_tri_last_diag(double *out4) { for (int k = 0; k < 4; ++k) out4[k] = g_tri_diag[k]; }

int orc_triangulate(const orc_tri_params *par, int pose_count, const orc_campose *trail, const double *image_features,
                    const double *feature_velocities, int stereo, int calc_derivatives, int derivative_test, double time_shift,
                    double *pf, double *dpfdp, double *dpfdq, double *dpfdt)
{
    if (par->useLinearTriangulation)                                               /* :146-152 */
        return triangulate_linear(par, pose_count, trail, image_features, feature_velocities, calc_derivatives, derivative_test,
                                  time_shift, pf, dpfdp, dpfdq, dpfdt);
    const int est = par->estimateImuCameraTimeShift;
    const int ind1 = stereo ? pose_count / 2 - 1 : pose_count - 1;
    double dpf2[45];
    orc_triangulate_with_two_cameras(&trail[0], &trail[ind1], image_features, image_features + 2 * ind1, feature_velocities,
                                     feature_velocities + 2 * ind1, calc_derivatives, est, derivative_test, time_shift, pf, dpf2);
    double dpfi_dpf[9], pfi[3];
    orc_inverse_depth(pf, pfi, dpfi_dpf);
    double R0T[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R0T[3 * r + c] = trail[0].R[3 * c + r];
    const int dDim = calc_derivatives ? pose_count * POSE_DIM : 0, ncol = dDim + 1;
    double *dpfi = (double *)calloc((size_t)3 * ncol, sizeof(double));          /* 3 x (dDim + 1), row-major */
    double *dETE = (double *)calloc((size_t)9 * ncol, sizeof(double));          /* per column j a 3x3 block */
    double *dEerror = (double *)calloc((size_t)3 * ncol, sizeof(double));       /* per column j a 3-vector */
    if (calc_derivatives) {
        for (int r = 0; r < 3; ++r) for (int j = 0; j < POSE_DIM; ++j) {
            dpfi[r * ncol + j] = dpf2[15 * r + j];
            dpfi[r * ncol + POSE_DIM * ind1 + j] = dpf2[15 * r + POSE_DIM + j];     /* written second: wins when ind1 == 0 */
        }
        const double dt[3] = {dpf2[14], dpf2[29], dpf2[44]};
        double t[3];
        mv3(dpfi_dpf, dt, t);
        for (int r = 0; r < 3; ++r) dpfi[r * ncol + dDim] = t[r];
        const int which[2] = {0, ind1};
        for (int w = 0; w < 2; ++w) for (int j = 0; j < POSE_DIM; ++j) {         /* applied twice when ind1 == 0, like the reference */
            const int col = which[w] * POSE_DIM + j;
            const double cur[3] = {dpfi[col], dpfi[ncol + col], dpfi[2 * ncol + col]};
            mv3(dpfi_dpf, cur, t);
            for (int r = 0; r < 3; ++r) dpfi[r * ncol + col] = t[r];
        }
    }
    double rcond = 0.0, Jprev = 1e10;
    int converged = 0;
    const double *p0 = trail[0].p;
    g_tri_diag[0] = 1e300; g_tri_diag[1] = 1e300; g_tri_diag[2] = 0; g_tri_diag[3] = 0;
    for (unsigned it = 0; it < par->triangulationGaussNewtonIterations; ++it) {
        double ETE[9] = {0}, Eerror[3] = {0}, error2 = 0;
        memset(dETE, 0, sizeof(double) * 9 * ncol);
        memset(dEerror, 0, sizeof(double) * 3 * ncol);
        for (int i = 0; i < pose_count; ++i) {
            const orc_campose *cur = &trail[i];
            double C[9], t[3], d[3], h[3];
            mm3(cur->R, R0T, C);
            for (int a = 0; a < 3; ++a) d[a] = p0[a] - cur->p[a];
            mv3(cur->R, d, t);
            const double pfiab[3] = {pfi[0], pfi[1], 1.0};
            mv3(C, pfiab, h);
            for (int a = 0; a < 3; ++a) h[a] += pfi[2] * t[a];
            const double ih2sq = 1.0 / (h[2] * h[2]);
            {
                const double hn = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]), rz = hn > 0 ? fabs(h[2]) / hn : 0.0;
                if (!(rz >= g_tri_diag[1])) g_tri_diag[1] = rz;                  /* (NaN counts as degenerate) */
            }
            double err[2] = {image_features[2 * i] - h[0] / h[2], image_features[2 * i + 1] - h[1] / h[2]};
            if (derivative_test && est) { err[0] += time_shift * feature_velocities[2 * i]; err[1] += time_shift * feature_velocities[2 * i + 1]; }
            double E[6];                                                          /* 2 x 3 */
            for (int r = 0; r < 2; ++r) {
                for (int c = 0; c < 2; ++c) E[3 * r + c] = (-1 / h[2]) * C[3 * r + c] + h[r] * ih2sq * C[6 + c];
                E[3 * r + 2] = -t[r] / h[2] + h[r] * ih2sq * t[2];
            }
            error2 += err[0] * err[0] + err[1] * err[1];
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) ETE[3 * r + c] += E[r] * E[c] + E[3 + r] * E[3 + c];
                Eerror[r] += E[r] * err[0] + E[3 + r] * err[1];
            }
            for (int j = 0; j < (calc_derivatives ? ncol : 0); ++j) {
                const int is_t = j == dDim;
                if (is_t && !est) continue;
                double dC[9] = {0}, dt[3] = {0};
                if (!is_t) {
                    const int pose_idx = j / POSE_DIM, comp = j % POSE_DIM, current = pose_idx == i;
                    double dRi[9] = {0}, dR0[9] = {0}, dp0[3] = {0}, dpi[3] = {0};
                    if (comp < 3) {
                        if (current) dpi[comp] = 1;
                        if (pose_idx == 0) dp0[comp] = 1;
                    } else {
                        const int qi = comp - 3;
                        if (current) {
                            memcpy(dRi, trail[pose_idx].dR[qi], sizeof dRi);
                            mTv3(dRi, trail[pose_idx].baseline, dpi);
                            for (int a = 0; a < 3; ++a) dpi[a] = -dpi[a];
                        }
                        if (pose_idx == 0) {
                            memcpy(dR0, trail[0].dR[qi], sizeof dR0);
                            mTv3(dR0, trail[0].baseline, dp0);
                            for (int a = 0; a < 3; ++a) dp0[a] = -dp0[a];
                        }
                    }
                    double a1[9], a2[9], t1[3], t2[3], dd[3];
                    mm3(dRi, R0T, a1); mmT3(cur->R, dR0, a2);
                    for (int k = 0; k < 9; ++k) dC[k] = a1[k] + a2[k];
                    mv3(dRi, d, t1);
                    for (int a = 0; a < 3; ++a) dd[a] = dp0[a] - dpi[a];
                    mv3(cur->R, dd, t2);
                    for (int a = 0; a < 3; ++a) dt[a] = t1[a] + t2[a];
                }
                const double dpfiab[3] = {dpfi[j], dpfi[ncol + j], 0.0};
                double dh[3], u1[3], u2[3];
                mv3(dC, pfiab, u1); mv3(C, dpfiab, u2);
                for (int a = 0; a < 3; ++a) dh[a] = u1[a] + u2[a] + dpfi[2 * ncol + j] * t[a] + pfi[2] * dt[a];
                const double dih2 = -dh[2] / (h[2] * h[2]), dih2sq = -2 * dh[2] * ih2sq / h[2];
                double dErr[2], dE[6];
                for (int r = 0; r < 2; ++r) {
                    dErr[r] = (is_t ? feature_velocities[2 * i + r] : 0.0) - dh[r] / h[2] - dih2 * h[r];
                    for (int c = 0; c < 2; ++c)
                        dE[3 * r + c] = -dih2 * C[3 * r + c] + (-1 / h[2]) * dC[3 * r + c] + (dh[r] * ih2sq + dih2sq * h[r]) * C[6 + c]
                                        + h[r] * ih2sq * dC[6 + c];
                    dE[3 * r + 2] = -dt[r] / h[2] - t[r] * dih2 + dh[r] * ih2sq * t[2] + h[r] * dih2sq * t[2] + h[r] * ih2sq * dt[2];
                }
                for (int r = 0; r < 3; ++r) {
                    dEerror[3 * j + r] += dE[r] * err[0] + dE[3 + r] * err[1] + E[r] * dErr[0] + E[3 + r] * dErr[1];
                    for (int c = 0; c < 3; ++c)
                        dETE[9 * j + 3 * r + c] += dE[r] * E[c] + dE[3 + r] * E[3 + c] + E[r] * dE[c] + E[3 + r] * dE[3 + c];
                }
            }
        }
        double X[9], step[3];
        inv3sym(ETE, X);
        mv3(X, Eerror, step);
        for (int a = 0; a < 3; ++a) pfi[a] += -step[a];
        for (int j = 0; j < (calc_derivatives ? ncol : 0); ++j) {
            double t1[3], t2[3], t3[3];
            mv3(dETE + 9 * j, step, t1);            /* dETE_j * X.solve(Eerror) */
            mv3(X, t1, t2);                          /* dX_Eerror = -X.solve(...) */
            mv3(X, dEerror + 3 * j, t3);
            for (int a = 0; a < 3; ++a) dpfi[a * ncol + j] += -t3[a] - (-t2[a]);
        }
        rcond = 1.0 / (norm1_3(ETE) * norm1_3(X));
        if (!(rcond >= g_tri_diag[0])) g_tri_diag[0] = rcond;
        g_tri_diag[2] = (double)(it + 1);
        const double Rnoise = par->triangulationConvergenceR * par->triangulationConvergenceR;
        const double J = 0.5 * error2 / Rnoise, Jd = fabs((J - Jprev) / J);
        Jprev = J;
        if (Jd < par->triangulationConvergenceThreshold) { converged = 1; break; }
    }
    g_tri_diag[3] = (double)converged;
    int status = TRI_OK;
    if (!converged) status = TRI_NO_CONVERGENCE;
    else if (rcond < par->triangulationRcondThreshold) status = TRI_BAD_COND;
    if (status == TRI_OK) {
        double dpf0_dpfi[9], pf0[3], t[3];
        orc_inverse_depth(pfi, pf0, dpf0_dpfi);
        mv3(R0T, pf0, t);
        for (int a = 0; a < 3; ++a) pf[a] = t[a] + p0[a];
        if (pf[0] == p0[0] && pf[1] == p0[1] && pf[2] == p0[2]) status = TRI_UNKNOWN_PROBLEM;
        else {
            double M[9];
            mm3(R0T, dpf0_dpfi, M);
            for (int j = 0; j < (calc_derivatives ? ncol : 0); ++j) {
                double dp0[3] = {0, 0, 0}, u[3] = {0, 0, 0}, v[3];
                if (j < 3) dp0[j] = 1;
                if (j >= 3 && j < POSE_DIM) mTv3(trail[0].dR[j - 3], pf0, u);       /* dR0T * pf0 */
                const double cur[3] = {dpfi[j], dpfi[ncol + j], dpfi[2 * ncol + j]};
                mv3(M, cur, v);
                for (int a = 0; a < 3; ++a) dpfi[a * ncol + j] = u[a] + v[a] + dp0[a];
            }
            if (calc_derivatives) {
                for (int j = 0; j < pose_count; ++j)
                    for (int r = 0; r < 3; ++r) {
                        for (int c = 0; c < 3; ++c) dpfdp[9 * j + 3 * r + c] = dpfi[r * ncol + j * POSE_DIM + c];
                        for (int c = 0; c < 4; ++c) dpfdq[12 * j + 4 * r + c] = dpfi[r * ncol + j * POSE_DIM + 3 + c];
                    }
                for (int r = 0; r < 3; ++r) dpfdt[r] = dpfi[r * ncol + ncol - 1];
            }
            for (int i = 0; i < pose_count; ++i) {                                    /* isBehind, :54-60 */
                double dd[3], a[3];
                for (int k = 0; k < 3; ++k) dd[k] = pf[k] - trail[i].p[k];
                mv3(trail[i].R, dd, a);
                if (a[2] < 0) { status = TRI_BEHIND; break; }
            }
        }
    }
    free(dpfi); free(dETE); free(dEerror);
    return status;
}

/* prepareVisualUpdate (triangulation.cpp:897-987). trail: n_trail poses (n or 2n), pose_trail_index: n entries.
 * dpfdp / dpfdq: n blocks (already summed over the stereo pair) or NULL (map-point update). H: column-major
 * (2 * n_trail) x end_idx, zero-filled here; returns the status, *end_idx_out = columns of H. */
int orc_prepare_visual_update(const double *pf, const double *dpfdp, const double *dpfdq, const double *dpfdt,
                              const double *feature_velocities, const orc_campose *trail, int n_trail, const int *pose_trail_index,
                              int n, int state_dim, int truncated, int map_point_offset, int estimate_time_shift, int derivative_test,
                              double time_shift, double *H, double *f, int *end_idx_out)
{
    int end_idx = 0;
    if (truncated) {
        for (int k = 0; k < n; ++k) {
            int jp, jo;
            orc_get_pos_ori_indices(pose_trail_index[k], &jp, &jo);
            const int e = jp + 3 > jo + 4 ? jp + 3 : jo + 4;
            if (e > end_idx) end_idx = e;
        }
        if (map_point_offset > 0) end_idx = map_point_offset + 3;
    } else end_idx = state_dim;
    *end_idx_out = end_idx;
    const int rows = 2 * n_trail;
    memset(H, 0, sizeof(double) * (size_t)rows * end_idx);
    memset(f, 0, sizeof(double) * rows);
#define HH(r, c) H[(size_t)(c) * rows + (r)]
    for (int i = 0; i < n_trail; ++i) {
        const int ti = i % n;
        const orc_campose *pose = &trail[i];
        double pt[3], pfc[3], dipH[9], ipH[3];
        for (int a = 0; a < 3; ++a) pt[a] = pf[a] - pose->p[a];
        mv3(pose->R, pt, pfc);
        if (pfc[2] == 0) return PREPARE_VU_ZERO_DEPTH;
        else if (pfc[2] < 0) return PREPARE_VU_BEHIND;
        orc_inverse_depth(pfc, ipH, dipH);
        f[2 * i] = ipH[0]; f[2 * i + 1] = ipH[1];
        if (derivative_test && estimate_time_shift) { f[2 * i] -= time_shift * feature_velocities[2 * i]; f[2 * i + 1] -= time_shift * feature_velocities[2 * i + 1]; }
        int ip, io;
        orc_get_pos_ori_indices(pose_trail_index[ti], &ip, &io);
        double dipR[6];                                                   /* dip (2x3) * R */
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) dipR[3 * r + c] = dipH[3 * r] * pose->R[c] + dipH[3 * r + 1] * pose->R[3 + c] + dipH[3 * r + 2] * pose->R[6 + c];
        for (int j = 0; j < 4; ++j) {
            double a[3], b1[3], b2[3];
            mv3(pose->dR[j], pt, a);
            mTv3(pose->dR[j], pose->baseline, b1);
            mv3(pose->R, b1, b2);
            for (int r = 0; r < 2; ++r) HH(2 * i + r, io + j) = dipH[3 * r] * (a[0] + b2[0]) + dipH[3 * r + 1] * (a[1] + b2[1]) + dipH[3 * r + 2] * (a[2] + b2[2]);
        }
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) HH(2 * i + r, ip + c) = -dipR[3 * r + c];
        if (dpfdp) {
            for (int j = 0; j < n; ++j) {
                int jp, jo;
                orc_get_pos_ori_indices(pose_trail_index[j], &jp, &jo);
                for (int r = 0; r < 2; ++r) {
                    for (int c = 0; c < 3; ++c)
                        HH(2 * i + r, jp + c) += dipR[3 * r] * dpfdp[9 * j + c] + dipR[3 * r + 1] * dpfdp[9 * j + 3 + c] + dipR[3 * r + 2] * dpfdp[9 * j + 6 + c];
                    for (int c = 0; c < 4; ++c)
                        HH(2 * i + r, jo + c) += dipR[3 * r] * dpfdq[12 * j + c] + dipR[3 * r + 1] * dpfdq[12 * j + 4 + c] + dipR[3 * r + 2] * dpfdq[12 * j + 8 + c];
                }
            }
            if (estimate_time_shift)
                for (int r = 0; r < 2; ++r)
                    HH(2 * i + r, SFT) = dipR[3 * r] * dpfdt[0] + dipR[3 * r + 1] * dpfdt[1] + dipR[3 * r + 2] * dpfdt[2] - feature_velocities[2 * i + r];
        }
        if (map_point_offset > 0)
            for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) HH(2 * i + r, map_point_offset + c) += dipR[3 * r + c];
    }
#undef HH
    return PREPARE_VU_OK;
}

/* The per-track glue of the visual update loop for a pose-trail (non map point) track, backend.cpp:1063-1148:
 * extract the trail from the current mean, triangulate with derivatives, depth test against trail[0], sum the stereo
 * derivative blocks, prepareVisualUpdate (not truncated: batch or full-width H as the HIP path uses).
 * H column-major (2 * ncam * n) x state_dim. Returns the triangulation status; *prepare_status is written (and H, f
 * filled) whenever backend.cpp reaches prepareVisualUpdate, i.e. always -- with empty derivative blocks unless OK. */
int orc_visual_track_prepare(const orc_tri_params *par, const double *m, int state_dim, const int *pose_trail_index, int n,
                             const double *imu_to_cam, const double *imu_to_cam2, const double *image_features,
                             const double *feature_velocities, double *pf, double *H, double *f, int *prepare_status)
{
    const int stereo = imu_to_cam2 != NULL, nt = stereo ? 2 * n : n;
    orc_campose *trail = (orc_campose *)malloc(sizeof(orc_campose) * (size_t)nt);
    double *dpfdp = (double *)calloc((size_t)9 * nt, sizeof(double)), *dpfdq = (double *)calloc((size_t)12 * nt, sizeof(double));
    double dpfdt[3] = {0, 0, 0};
    orc_extract_camera_pose_trail(m, pose_trail_index, n, imu_to_cam, imu_to_cam2, trail);
    int status = orc_triangulate(par, nt, trail, image_features, feature_velocities, stereo, 1, 0, 0.0, pf, dpfdp, dpfdq, dpfdt);
    const double dx = pf[0] - trail[0].p[0], dy = pf[1] - trail[0].p[1], dz = pf[2] - trail[0].p[2];
    const double depth = sqrt(dx * dx + dy * dy + dz * dz);
    if (depth < par->triangulationMinDist || depth > par->triangulationMaxDist) status = TRI_BAD_DEPTH;
    if (stereo && status == TRI_OK)
        for (int i = 0; i < n; ++i) {
            for (int k = 0; k < 9; ++k) dpfdp[9 * i + k] += dpfdp[9 * (i + n) + k];
            for (int k = 0; k < 12; ++k) dpfdq[12 * i + k] += dpfdq[12 * (i + n) + k];
        }
    int end_idx;
    const int ok = status == TRI_OK;
    *prepare_status = orc_prepare_visual_update(pf, ok ? dpfdp : NULL, ok ? dpfdq : NULL, dpfdt, feature_velocities, trail, nt,
                                                pose_trail_index, n, state_dim, 0, -1, par->estimateImuCameraTimeShift, 0, 0.0, H, f, &end_idx);
    free(trail); free(dpfdp); free(dpfdq);
    return status;
}
