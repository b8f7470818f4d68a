// Per-track triangulation + prepareVisualUpdate on the device: builds the (H, y - f) of one
// feature track per filter straight from the device-resident EKF mean, so that the chi2 gate and the visual update
// (ekf.hip) run without the mean travelling to the host and a 2*nPoses x stateDim Jacobian travelling back.
//
// Reference: src/odometry/backend.cpp:1063-1148 (the per-track glue), src/odometry/triangulation.cpp:65-103
// (extractCameraPoseTrail), :120-407 (Triangulator::triangulate, iterative PIVO method with derivatives),
// :612-716 (triangulateWithTwoCameras), :31-52 (dpinv), :897-987 (prepareVisualUpdate), :1004-1012 (inverseDepth),
// src/odometry/util.cpp:10-47 (quat2rmat_d). 3x3 matrices are row-major double[9] (like oracle/triangulation_oracle.c).
//
// One workgroup per filter. The work is the differentiated Gauss-Newton loop: per iteration every one of the
// 7 * nPoses + 1 derivative columns visits every pose (<= 42 x 295 pairs, ~120 flops each, f64), so a thread owns a
// column and walks the poses, whose per-iteration quantities (C, t, h, E, error) are shared through LDS.
#include <stdlib.h>

#include "ekf_device.hpp"

#pragma clang fp contract(fast)

namespace hv {
// developer aid: s_memtime stamps of workgroup 0 at the phase boundaries (only with -DHV_EKF_PHASE_STAMPS)
__device__ long long g_vu_stamp[40];
__device__ long long g_tri_stamp[64];
#ifdef HV_EKF_PHASE_STAMPS
#define VU_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_vu_stamp[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define TRI_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_tri_stamp[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define VU_STAMP(i) do { } while (0)
#define TRI_STAMP(i) do { } while (0)
#endif
namespace {

// threads per workgroup (template parameter VT): G = 2 or 4 lanes per derivative column (<= 7 * 42 + 1 columns) + one wave.
// 768 gives the longest track (21 stereo poses) its G = 2 and a 10-pose stereo track G = 4: the shortest latency of ONE track.
// 384 threads cover tracks of up to 22 camera poses (see the kernel's template parameters).
constexpr int VT_LATENCY = 768, VT_THROUGHPUT = 384;
constexpr int MAXP_ALL = 42;            // 2 cameras x (cameraTrailLength + 1 <= 21) poses: the largest track
constexpr int MAXP_SMALL = 22;          // the 384-thread build: up to 22 camera poses (10 or 11 stereo poses, 21 mono poses)
constexpr int MAXP_REC = 24;            // the record-fed short-class gate (r06): up to 12 stereo poses = 48 rows, everything sparse_gate's 3 row tiles hold
constexpr int MAXNP = 21;              // poses per camera: s_dpf is [MAXNP][21], s_idx holds MAXNP (+3 spare) indices
constexpr int POSE_WORDS = 51;          // p[3] R[9] dR[4][9] baseline[3]
constexpr int ITER_WORDS = 26;          // C[9] t[3] h[3] E[6] err[2] d[3]

__device__ __forceinline__ void mm3(const double *A, const double *B, double *C)
{
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
__device__ __forceinline__ void mmT3(const double *A, const double *B, double *C)   // A B^T
{
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[3 * c] + A[3 * r + 1] * B[3 * c + 1] + A[3 * r + 2] * B[3 * c + 2];
}
__device__ __forceinline__ void mv3(const double *A, const double *x, double *y)
{
#pragma unroll
    for (int r = 0; r < 3; ++r) y[r] = A[3 * r] * x[0] + A[3 * r + 1] * x[1] + A[3 * r + 2] * x[2];
}
__device__ __forceinline__ void mTv3(const double *A, const double *x, double *y)
{
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = A[c] * x[0] + A[3 + c] * x[1] + A[6 + c] * x[2];
}

__device__ void quat2rmat_d(const double *q, double *R, double *dR /* [4][9] */)    // util.cpp:10-47
{
    R[0] = q[0] * q[0] + q[1] * q[1] - q[2] * q[2] - q[3] * q[3]; R[1] = 2 * q[1] * q[2] - 2 * q[0] * q[3]; R[2] = 2 * q[1] * q[3] + 2 * q[0] * q[2];
    R[3] = 2 * q[1] * q[2] + 2 * q[0] * q[3]; R[4] = q[0] * q[0] - q[1] * q[1] + q[2] * q[2] - q[3] * q[3]; R[5] = 2 * q[2] * q[3] - 2 * q[0] * q[1];
    R[6] = 2 * q[1] * q[3] - 2 * q[0] * q[2]; R[7] = 2 * q[2] * q[3] + 2 * q[0] * q[1]; R[8] = q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3];
    const double a = 2 * q[0], b = 2 * q[1], c = 2 * q[2], d = 2 * q[3];
    const double d0[9] = {a, -d, c, d, a, -b, -c, b, a}, d1[9] = {b, c, d, c, -b, -a, d, a, -b};
    const double d2[9] = {-c, b, a, b, c, d, -a, d, -c}, d3[9] = {-d, -a, b, a, -d, c, b, c, d};
#pragma unroll
    for (int k = 0; k < 9; ++k) { dR[k] = d0[k]; dR[9 + k] = d1[k]; dR[18 + k] = d2[k]; dR[27 + k] = d3[k]; }
}

__device__ __forceinline__ void pos_ori(int i, int &pos, int &ori)                   // triangulation.cpp:989-998
{
    pos = i == 0 ? 0 : 20 + 7 * (i - 1);
    ori = i == 0 ? 6 : 20 + 7 * (i - 1) + 3;
}

__device__ __forceinline__ void inverse_depth(const double *p, double *ip, double *dip)   // :1004-1012
{
    const double iz = 1 / p[2];
    ip[0] = p[0] * iz; ip[1] = p[1] * iz; ip[2] = iz;
    dip[0] = iz; dip[1] = 0; dip[2] = -ip[0] * iz; dip[3] = 0; dip[4] = iz; dip[5] = -ip[1] * iz; dip[6] = 0; dip[7] = 0; dip[8] = -ip[2] * iz;
}

// pseudo-inverse of the full-rank 3x2 A (row-major) and its derivative (Golub & Pereyra 4.12, triangulation.cpp:31-52)
__device__ void pinv32(const double *A, double *iA)
{
    const double a = A[0] * A[0] + A[2] * A[2] + A[4] * A[4], b = A[0] * A[1] + A[2] * A[3] + A[4] * A[5];
    const double d = A[1] * A[1] + A[3] * A[3] + A[5] * A[5], idet = 1.0 / (a * d - b * b);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        iA[c] = (d * A[2 * c] - b * A[2 * c + 1]) * idet;
        iA[3 + c] = (-b * A[2 * c] + a * A[2 * c + 1]) * idet;
    }
}
__device__ void dpinv(const double *A, const double *iA, const double *dA, double *out)
{
    double iAiAT[4], iATiA[9], AiA[9], iAA[4], iAdA[4], t23[6], t23b[6];
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) {
        iAiAT[2 * r + c] = iA[3 * r] * iA[3 * c] + iA[3 * r + 1] * iA[3 * c + 1] + iA[3 * r + 2] * iA[3 * c + 2];
        iAA[2 * r + c] = iA[3 * r] * A[c] + iA[3 * r + 1] * A[2 + c] + iA[3 * r + 2] * A[4 + c];
        iAdA[2 * r + c] = iA[3 * r] * dA[c] + iA[3 * r + 1] * dA[2 + c] + iA[3 * r + 2] * dA[4 + c];
    }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        iATiA[3 * r + c] = iA[r] * iA[c] + iA[3 + r] * iA[3 + c];
        AiA[3 * r + c] = A[2 * r] * iA[c] + A[2 * r + 1] * iA[3 + c];
    }
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
        out[3 * r + c] = -(iAdA[2 * r] * iA[c] + iAdA[2 * r + 1] * iA[3 + c]);
        t23[3 * r + c] = iAiAT[2 * r] * dA[2 * c] + iAiAT[2 * r + 1] * dA[2 * c + 1];
        t23b[3 * r + c] = ((r == 0 ? 1.0 : 0.0) - iAA[2 * r]) * dA[2 * c] + ((r == 1 ? 1.0 : 0.0) - iAA[2 * r + 1]) * dA[2 * c + 1];
    }
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += t23[3 * r + k] * ((k == c ? 1.0 : 0.0) - AiA[3 * k + c]) + t23b[3 * r + k] * iATiA[3 * k + c];
        out[3 * r + c] += s;
    }
}

__device__ __forceinline__ double inv3sym(const double *M, double *inv)
{
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02, id = 1.0 / det;
    inv[0] = c00 * id; inv[1] = (M[2] * M[7] - M[1] * M[8]) * id; inv[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    inv[3] = c01 * id; inv[4] = (M[0] * M[8] - M[2] * M[6]) * id; inv[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    inv[6] = c02 * id; inv[7] = (M[1] * M[6] - M[0] * M[7]) * id; inv[8] = (M[0] * M[4] - M[1] * M[3]) * id;
    return det;
}
__device__ __forceinline__ double norm1_3(const double *M)
{
    double best = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) best = fmax(best, fabs(M[c]) + fabs(M[3 + c]) + fabs(M[6 + c]));
    return best;
}

// One (pose, column) pair of the derivative sums (triangulation.cpp:264-311): given dh (the change of h), dC and dt
// (the change of C and t; zero unless HEAVY) it adds dEblock' * error + Eblock' * dErrorBlock to dEe and
// dEblock' * Eblock + Eblock' * dEblock to dM. o = the pose record of this iteration (C t h E err d).
// MODE 0: dC = dt = 0 (the plain part); 1: both given (false / true of r01 .. r05's bool parameter); 2 (r06): dt only -- a position
// component moves t of the pose and leaves C alone
// ih2_given (r06): 1 / h_z of the pose record as the per-pose phase computed it (the same value: one f64 division sequence per pair less)
template <int MODE>
__device__ __forceinline__ void pair_sums(const double *o, const double *dh, const double *dC, const double *dt, double vel0,
                                          double vel1, double *dEe, double *dM, const double *ih2_given = nullptr)
{
    const double *C = o, *t = o + 9, *h = o + 12, *E = o + 15, *er = o + 21;
    const double ih2 = ih2_given ? *ih2_given : 1.0 / h[2], ih2sq = ih2 * ih2;
    const double dih2 = -dh[2] * ih2sq, dih2sq = -2 * dh[2] * ih2sq * ih2;
    double dErr[2], dE[6];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        dErr[r] = (r == 0 ? vel0 : vel1) - dh[r] * ih2 - dih2 * h[r];
        const double g = dh[r] * ih2sq + dih2sq * h[r];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            dE[3 * r + c] = -dih2 * C[3 * r + c] + g * C[6 + c];
            if (MODE == 1) dE[3 * r + c] += -ih2 * dC[3 * r + c] + h[r] * ih2sq * dC[6 + c];
        }
        dE[3 * r + 2] = -t[r] * dih2 + g * t[2];
        if (MODE != 0) dE[3 * r + 2] += -dt[r] * ih2 + h[r] * ih2sq * dt[2];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        dEe[r] += dE[r] * er[0] + dE[3 + r] * er[1] + E[r] * dErr[0] + E[3 + r] * dErr[1];
        // dEblock' Eblock + Eblock' dEblock is symmetric: only the upper triangle is accumulated (6 of 9 entries, a seventh of the
        // pair's flops); sym3() completes the matrix where it is used
#pragma unroll
        for (int c = r; c < 3; ++c) dM[3 * r + c] += dE[r] * E[c] + dE[3 + r] * E[3 + c] + E[r] * dE[c] + E[3 + r] * dE[3 + c];
    }
}

// dC and dt of pose i for state component comp of pose pj (:269-292), as one instruction stream: the derivative
// matrices are scaled by 0 or 1 instead of branching on position / quaternion and own pose / pose 0.
// Nothing in here depends on the Gauss-Newton iterate (C, t and d = p0 - p_i are fixed by the pose trail): evaluated once per
// lane before the loop and parked in LDS (s_mot), ~250 of the ~600 f64 instructions a lane spent on its motion pair per iteration.
__device__ __forceinline__ void pose_motion(const double *trail, const double *R0T, int i, int pj, int comp, double *dC, double *dt)
{
    const double *cur = trail + i * POSE_WORDS;
    const double d[3] = {trail[0] - cur[0], trail[1] - cur[1], trail[2] - cur[2]};
    const bool current = pj == i, first = pj == 0;
    const int qi = comp >= 3 ? comp - 3 : 0;
    const double wc = current && comp >= 3 ? 1.0 : 0.0, wf = first && comp >= 3 ? 1.0 : 0.0;
    double dRi[9], dR0[9], a1[9], a2[9], dpi[3], dp0[3], t1[3], t2[3], dd[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) { dRi[k] = wc * cur[12 + 9 * qi + k]; dR0[k] = wf * trail[12 + 9 * qi + k]; }
    mm3(dRi, R0T, a1);
    mmT3(cur + 3, dR0, a2);
    mTv3(dRi, cur + 48, dpi);
    mTv3(dR0, trail + 48, dp0);
    mv3(dRi, d, t1);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        dpi[k] = (current && comp == k ? 1.0 : 0.0) - dpi[k];
        dp0[k] = (first && comp == k ? 1.0 : 0.0) - dp0[k];
        dd[k] = dp0[k] - dpi[k];
    }
    mv3(cur + 3, dd, t2);
#pragma unroll
    for (int k = 0; k < 9; ++k) dC[k] = a1[k] + a2[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) dt[k] = t1[k] + t2[k];
}

// ---- shared by the fused body (vu_prepare_body) and the split form's front (vu_tri_body) ----
// extractCameraPoseTrail (triangulation.cpp:65-103): the record of trail pose t (pose k = t % n of camera t / n) from the mean:
// p[3] R[9] dR[4][9] baseline[3] (POSE_WORDS doubles at o)
__device__ __forceinline__ void trail_pose_record(const VuPrepareArgs &a, const double *m, const int *s_idx, int t, int n, double *o)
{
    const int cam = t / n, k = t - cam * n;
    const double *T = a.imu_to_cam[cam];                 // 3x4 row-major [R | baseline]
    const double Ric[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]}, base[3] = {T[3], T[7], T[11]};
    int ip, io;
    pos_ori(s_idx[k], ip, io);
    const double q[4] = {m[io], m[io + 1], m[io + 2], m[io + 3]};
    double Rw[9], dRw[36], R[9], t3[3];
    quat2rmat_d(q, Rw, dRw);
    mm3(Ric, Rw, R);
    mTv3(R, base, t3);
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) { o[k2] = m[ip + k2] - t3[k2]; o[48 + k2] = base[k2]; }
#pragma unroll
    for (int k2 = 0; k2 < 9; ++k2) o[3 + k2] = R[k2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double d[9];
        mm3(Ric, dRw + 9 * j, d);
#pragma unroll
        for (int k2 = 0; k2 < 9; ++k2) o[12 + 9 * j + k2] = d[k2];
    }
}

// triangulateWithTwoCameras between pose 0 and pose ind1 (triangulation.cpp:154-173, 612-716), by lanes j = tid < 15: lane j owns
// derivative column j (p0 q0 p1 q1 t) of dpfTwoCameras, every one of them recomputes the small shared part; the columns land in
// s_dpfi (mapped through dpfi_dpf, :181-199), lane 0 publishes pfi, pfw, R0', the convergence scalars and the flag
__device__ __forceinline__ void two_camera_start(const VuPrepareArgs &a, int tid, int ind1, int ncol, int dDim, const double *s_trail,
                                                 const double *s_feat, double *s_dpfi, double *pfi, double *pfw, double *R0T, double *scal, int *s_flag)
{
    const double *P0 = s_trail, *P1 = s_trail + ind1 * POSE_WORDS;
    const double *R0 = P0 + 3, *R1 = P1 + 3;
    double C[9], d01[3], bb[3];
    mmT3(R0, R1, C);
#pragma unroll
    for (int k = 0; k < 3; ++k) d01[k] = P1[k] - P0[k];
    mv3(R0, d01, bb);
    const double v0[3] = {s_feat[0], s_feat[1], 1.0}, v1[3] = {s_feat[4 * ind1], s_feat[4 * ind1 + 1], 1.0};
    const double n0 = sqrt(v0[0] * v0[0] + v0[1] * v0[1] + 1.0), n1 = sqrt(v1[0] * v1[0] + v1[1] * v1[1] + 1.0);
    const double vn0[3] = {v0[0] / n0, v0[1] / n0, v0[2] / n0}, vn1[3] = {v1[0] / n1, v1[1] / n1, v1[2] / n1};
    double Cvn1[3], A[6], iA[6];
    mv3(C, vn1, Cvn1);
#pragma unroll
    for (int r = 0; r < 3; ++r) { A[2 * r] = vn0[r]; A[2 * r + 1] = -Cvn1[r]; }
    pinv32(A, iA);
    const double s0 = iA[0] * bb[0] + iA[1] * bb[1] + iA[2] * bb[2];
    double pf[3] = {s0 * vn0[0], s0 * vn0[1], s0 * vn0[2]};
    double ip3[3], dd[9];
    inverse_depth(pf, ip3, dd);
    // column tid of dpfTwoCameras
    double dA[6] = {0, 0, 0, 0, 0, 0}, db[3] = {0, 0, 0}, col[3];
    const int j = tid;
    if (j < 14) {
        const int second = j >= 7, comp = second ? j - 7 : j;
        if (comp < 3) {
#pragma unroll
            for (int r = 0; r < 3; ++r) db[r] = (second ? 1.0 : -1.0) * R0[3 * r + comp];
        } else {
            const int qi = comp - 3;
            double dC[9], t[3];
            if (!second) { mmT3(P0 + 12 + 9 * qi, R1, dC); mv3(P0 + 12 + 9 * qi, d01, db); }
            else mmT3(R0, P1 + 12 + 9 * qi, dC);
            mv3(dC, vn1, t);
#pragma unroll
            for (int r = 0; r < 3; ++r) dA[2 * r + 1] = -t[r];
        }
        double diA[6];
        dpinv(A, iA, dA, diA);
        const double ds = (iA[0] * db[0] + iA[1] * db[1] + iA[2] * db[2]) + (diA[0] * bb[0] + diA[1] * bb[1] + diA[2] * bb[2]);
#pragma unroll
        for (int r = 0; r < 3; ++r) col[r] = ds * vn0[r];
    } else if (a.est_shift) {
        double w0[3], w1[3], cw1[3], diA[6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            w0[r] = 0; w1[r] = 0;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                w0[r] += ((r == c ? 1.0 : 0.0) - vn0[r] * vn0[c]) / n0 * s_feat[2 + c];
                w1[r] += ((r == c ? 1.0 : 0.0) - vn1[r] * vn1[c]) / n1 * s_feat[4 * ind1 + 2 + c];
            }
        }
        mv3(C, w1, cw1);
#pragma unroll
        for (int r = 0; r < 3; ++r) { dA[2 * r] = w0[r]; dA[2 * r + 1] = -cw1[r]; }
        dpinv(A, iA, dA, diA);
        const double ds0dt = diA[0] * bb[0] + diA[1] * bb[1] + diA[2] * bb[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) col[r] = s0 * w0[r] + vn0[r] * ds0dt;
    } else { col[0] = col[1] = col[2] = 0.0; }
    // :181-199: place the two pose blocks and the time-shift column, all mapped through dpfi_dpf
    double mapped[3];
    mv3(dd, col, mapped);
    const int dst = j < 7 ? j : j < 14 ? 7 * ind1 + (j - 7) : dDim;
    if (!(j < 7 && ind1 == 0)) {                         // (a one-pose trail cannot occur: poseCount >= 2)
#pragma unroll
        for (int r = 0; r < 3; ++r) s_dpfi[r * ncol + dst] = mapped[r];
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { pfi[k] = ip3[k]; pfw[k] = pf[k]; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) R0T[3 * r + c] = R0[3 * c + r];
        scal[1] = 0.0; scal[2] = 1e10;
        s_flag[0] = 0;                                   // converged
    }
}

// LDS layout of the kernel in doubles, for MAXP camera poses
// LONG (the fused prepare + gate of the long class, FUSED = 3): the gate of a 49 .. 84-row track needs more room than the Gauss-Newton
// arrays leave in the FUSED = 1 carve -- [S; v'] up to 86 x 84 doubles in [0, LONG_T) and, behind it, LONG_HS doubles for the factors of
// the Jacobian, their products and A = P(a, a) Dp' (structured_S; until late r04 the staged compact Jacobian of up to 148 x 84 doubles,
// which sized the region). Everything the factors are copied FROM (pose records, dpf, dpfi, features, indices) lies below P0, the
// region lies over the dead motion / linear-map arrays, and the index arrays move behind both: 158.6 KB of the CU's 160.
template <int MAXP, bool LONG = false>
struct VuLds {
    static constexpr int MAXC = MAXP * 7 + 1, MOT_STRIDE = 13;
    static constexpr int MAXPAIRS = 14 * MAXP - 7;            // motion pairs: 7 nt with a pose-0 column + 7 (nt - 1) with the pose's own column
    static constexpr int LONG_T = 7224, LONG_HS = 12432;      // doubles: (84 + 2) x 84, 148 x 84
    static constexpr int TRAIL = 0, IT = TRAIL + MAXP * POSE_WORDS, DPFI = IT + MAXP * ITER_WORDS, FEAT = DPFI + 3 * MAXC,
                         SMALL = FEAT + MAXP * 4, DPF = SMALL + 64, P0 = DPF + MAXNP * 21, MOT = P0 + 7 * MAXP * 9 + 7 * 9,
                         OWN = MOT + MAXPAIRS * MOT_STRIDE, LIN = OWN + MAXC * 9, LIN_END = LIN + 3 * MAXP * 9 + 32,
                         INTS = (LONG && LONG_T + LONG_HS > LIN_END) ? LONG_T + LONG_HS : LIN_END,
                         TOTAL = INTS + (MAXNP + 3 + 4 + MAXC + 1) / 2 + 1;       // s_idx, s_flag, s_acol (fused gate)
    static_assert(!LONG || P0 <= LONG_T, "the sources of the Jacobian's factors must lie below the region they are copied to");
    // fused gate (FUSED = 1 builds): once H exists the Gauss-Newton work arrays are dead -- the compact Jacobian is staged in [P0, INTS),
    // the (rows + 1) x rows matrix [S; v'] in [0, P0)
    static constexpr int HS_DOUBLES = INTS - P0, T_DOUBLES = P0;
    static constexpr size_t BYTES = sizeof(double) * TOTAL;
};
static_assert(VuLds<42, true>::BYTES <= 160 * 1024, "the long build must fit one CU's LDS");

// LDS layout of the RECORD-FED gate builds (r06, VuPrepareArgs::from_rec): the front has run in vu_tri_kernel, so only the gate's own
// areas are left -- [S; v'] in [0, P0), the staged Jacobian / the factors and their products behind it -- and the record's copies (the
// per-pose values, features, dpf, the time-shift column) alias the front of [S; v']: they are dead before T is zeroed. Short class:
// 48 KB instead of 80, THREE workgroups per CU.
template <int MAXP, bool LONG = false>
struct VuRecLds {
    static constexpr int MAXC = MAXP * 7 + 1, MOT_STRIDE = 13;
    static constexpr int LONG_T = 7224, LONG_HS = 12432;
    static constexpr int TRAIL = 0, IT = 0, FEAT = IT + MAXP * ITER_WORDS, DPF = FEAT + MAXP * 4, DPFI = DPF + MAXP * 21, SMALL = DPFI + 4,
                         SMALL_END = SMALL + 64;
    static constexpr int P0 = LONG ? LONG_T : 2352;                      // short class: [S; v'] of up to 48 rows at stride 49 (44 rows: 47)
    static constexpr int MOT = P0, OWN = P0, LIN = P0;                   // (no Gauss-Newton arrays)
    static constexpr int INTS = LONG ? LONG_T + LONG_HS : P0 + 4224;     // short class: the staged Jacobian, 88 columns x 48 rows
    static constexpr int TOTAL = INTS + (MAXNP + 3 + 4 + MAXC + 1) / 2 + 1;
    static constexpr int HS_DOUBLES = INTS - P0, T_DOUBLES = P0;
    static_assert(SMALL_END <= P0, "the record's copies must fit in front of the staged Jacobian");
    static constexpr size_t BYTES = sizeof(double) * TOTAL;
};
// (LDS is handed out in granules of 1280 bytes, 128 per CU: three workgroups get 42 each; 32 bytes of static LDS ride along -- gate_turn_lds)
static_assert(VuRecLds<MAXP_REC>::BYTES + 32 <= 42 * 1280, "three record-fed short-class gates per CU");

// MAXP (camera poses the LDS arrays are sized for) is a template parameter next to VT: <768, 42> holds every track (151 KB of LDS,
// one workgroup per CU); <384, 22> holds the common sizes in 75 KB and 6 waves of 138 VGPRs, so TWO filters share a CU and one's
// serial sections (pose records, 3 x 3 solves, barriers: most of the kernel since r02 took the column work off the critical path)
// overlap the other's.
// FUSED (VuPrepareArgs::fused): 0 = the dense H of the public prepare entry point; 1 = compact Jacobian + the chi2 gate in this kernel
// (ekf_device.hpp sparse_gate; few filters: one launch per visit / speculative pass); 2 = compact Jacobian only, the gate follows as
// ekf_sparse_gate_kernel (many filters: three small workgroups per CU hide its Cholesky chain, which two of these cannot); 3 = like 1
// for the LONG class (49 .. 84 rows, VuLds<.., true>; r04 -- r03 ran 2 + the big gate kernel: two launches and a round trip of Hc
// through HBM on the critical path of every visit), with S formed from the FACTORS of the Jacobian on the vector unit (ekf_device.hpp
// structured_S) instead of sparse_gate's dense MFMA products. In 1 .. 3 the dense H is never written, only Hc / acol / v.
// MAP: the launch may hold hybrid-map tracks (VuPrepareArgs::map_index) -- its own instantiation, the others compile as before
// REC (r06): the gate half only -- the front has run as vu_tri_kernel and left a factor record per track (VuPrepareArgs::tri_rec)
template <int VT, int MAXP, int FUSED, bool MAP = false, bool REC = false>
__device__ __forceinline__ void vu_prepare_body(const VuPrepareArgs &a, const int bx /* filter: blockIdx.x, or the loop variable of a persistent launch */)
{
    // All LDS comes from the dynamic region (carved below): with static arrays the compiler derives the occupancy from their size
    // and stops honouring the register cap that lets two of the small workgroups share a CU.
    static_assert(!REC || (FUSED == 1 || FUSED == 3), "record-fed builds are gate builds");
    using Lay = std::conditional_t<REC, VuRecLds<MAXP, FUSED == 3>, VuLds<MAXP, FUSED == 3>>;
    constexpr int MOT_STRIDE = Lay::MOT_STRIDE;
    extern __shared__ __attribute__((aligned(16))) double vu_lds[];
    double *s_trail = vu_lds + Lay::TRAIL;       // [MAXP][POSE_WORDS]
    double *s_it = vu_lds + Lay::IT;             // [MAXP][ITER_WORDS]
    double *s_dpfi = vu_lds + Lay::DPFI;         // [3][ncol]
    double *s_feat = vu_lds + Lay::FEAT;         // image feature (2) + velocity (2) per pose
    double *s_small = vu_lds + Lay::SMALL;       // pfi[3] pf[3] X[9] step[3] ETE[9] Eerror[3] R0T[9] pf0 ... (see offsets)
    double *s_dpf = vu_lds + Lay::DPF;           // summed dpfdp [n][9] and dpfdq [n][12]
    double *s_p0 = vu_lds + Lay::P0;             // motion part of the 7 columns of pose 0: [7][pose][9] (dEe, upper dM), then their totals [7][9]
    double *s_mot = vu_lds + Lay::MOT;           // dC[9] dt[3] of every motion pair (+1: lanes 13 doubles apart hit the LDS banks two-way at worst)
    double *s_own = vu_lds + Lay::OWN;           // motion sums of the own pairs, by column (dEe[3], upper triangle of dM[6])
    double *s_lin = vu_lds + Lay::LIN;           // plain part: per (pose, unit vector) 3 + 6 numbers, then L[3][9] and c_t[3]
    int *s_idx = reinterpret_cast<int *>(vu_lds + Lay::INTS);      // [MAXNP + 3]
    int *s_flag = s_idx + MAXNP + 3;                               // [4]
    const int b = bx, tid = threadIdx.x;
    // rec is computed below; a.np is the record STRIDE (the longest track of the launch) when per-record lengths are given
    const size_t rec_ = a.spec_tracks > 0 ? (size_t)blockIdx.y * a.batch + bx : (size_t)bx;
    const int np_rec = a.np_rec ? a.np_rec[rec_] : a.np;
    const bool no_track = np_rec < 2 || np_rec > a.np;              // ragged batches: this filter has no (valid) track in this launch
    const int n = no_track ? a.np : np_rec, ncam = a.stereo ? 2 : 1, nt = n * ncam, N = a.n;
    const int nt_max = a.np * ncam, rows_max = 2 * nt_max;
    const int dDim = nt * 7, ncol = dDim + 1;
    const double *m = a.m + (size_t)b * N;
    double *pfi = s_small, *pfw = s_small + 3, *X = s_small + 6, *step = s_small + 15, *R0T = s_small + 18;
    double *scal = s_small + 36;                 // [0] error2, [1] rcond, [2] Jprev
    VU_STAMP(0);
    // rec: the (track, filter) record this workgroup reads its inputs from and writes its outputs to (the filter itself without speculation)
    const size_t rec = a.spec_tracks > 0 ? (size_t)blockIdx.y * a.batch + b : (size_t)b;
    const bool quota_used = a.success_counter && a.success_counter[b] >= a.max_successful;
    if (a.spec_tracks > 0) {
        if ((int)blockIdx.y < a.cursor[b]) return;                               // final already
        if (!quota_used && a.epoch[rec] == a.success_counter[b]) return;          // prepared against the current mean
    }
    if (a.np_hi > 0 && np_rec >= 2 && np_rec <= a.np && (np_rec < a.np_lo || np_rec > a.np_hi)) {       // another length class serves this record
        if (tid == 0) {
            if (a.class_inactive && a.active) a.active[rec] = 0;
            if (a.long_list && np_rec > a.np_hi) a.long_list[atomicAdd(a.long_count, 1)] = (int)rec;
        }
        return;
    }
    if (a.np_hi > 0 && a.np_lo > 2 && no_track) {                 // "no track" records belong to the class that starts at 2 poses
        if (a.class_inactive && a.active && tid == 0) a.active[rec] = 0;
        return;
    }
    if (quota_used) {
        // backend.cpp:1233-1238: the frame's quota of successful visual updates is used up, the loop does not visit this track
        if (tid == 0) {
            a.status[2 * rec] = HV_TRI_NOT_VISITED; a.status[2 * rec + 1] = HV_TRI_NOT_VISITED;
            if (a.active) a.active[rec] = 0;
            if (a.gate_status) a.gate_status[rec] = 1;
            // a speculative pass may have left a gate result of an earlier (m, P) here: the sequential loop never visits this track
            if (a.spec_tracks > 0) {
                if (a.chi2) a.chi2[rec] = 0.0;
                a.pf[3 * rec] = 0.0; a.pf[3 * rec + 1] = 0.0; a.pf[3 * rec + 2] = 0.0;
            }
        }
        return;
    }
    if (no_track) {
        if (tid == 0) {
            a.status[2 * rec] = HV_TRI_NOT_VISITED; a.status[2 * rec + 1] = HV_TRI_NOT_VISITED;
            if (a.active) a.active[rec] = 0;
            if (a.gate_status) a.gate_status[rec] = 1;
            if (a.rows_out) a.rows_out[rec] = 0;
            if (a.spec_tracks > 0) a.epoch[rec] = a.success_counter[b];
        }
        return;
    }
    if (tid == 0 && a.rows_out) a.rows_out[rec] = 2 * nt;
    if (tid < n) s_idx[tid] = a.pose_index[rec * a.np + tid];
    if (tid < nt) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            s_feat[4 * tid + k] = a.features[(rec * nt_max + tid) * 2 + k];
            s_feat[4 * tid + 2 + k] = a.velocities[(rec * nt_max + tid) * 2 + k];
        }
    }
    int status = HV_TRI_OK;
    int *st_out = a.status + 2 * rec;
    bool with_derivatives = false;
    const double *p0 = s_trail;
    bool map_track = false; int map_off = 0;
    if constexpr (REC) {
        // the record vu_tri_kernel left for this track: [nt_max][17] per-pose values, [np][21] dpf, the time-shift column, prep
        const double *recp = a.tri_rec + rec * (size_t)a.tri_stride;
        const int R_DPF = 17 * nt_max, R_SFT = R_DPF + 21 * a.np;
        status = a.status[2 * rec];
        for (int w = tid; w < 17 * nt; w += VT) { const int i = w / 17, k = w - 17 * i; s_it[i * ITER_WORDS + k] = recp[w]; }
        if (status == HV_TRI_OK) {
            for (int w = tid; w < 21 * n; w += VT) s_dpf[w] = recp[R_DPF + w];
            if (tid < 3) s_dpfi[tid] = recp[R_SFT + tid];
        }
        if (tid < 3) pfw[tid] = a.pf[3 * rec + tid];
        __syncthreads();
    } else {
    __syncthreads();
    // ---- extractCameraPoseTrail (triangulation.cpp:65-103): pose k of camera c from the mean ----
    if (tid < nt) trail_pose_record(a, m, s_idx, tid, n, s_trail + tid * POSE_WORDS);
    for (int i = tid; i < 3 * ncol; i += VT) s_dpfi[i] = 0.0;
    __syncthreads();
    VU_STAMP(1);
    if constexpr (MAP) {
        const int mi = a.map_index ? a.map_index[rec] : -1;
        map_track = mi >= 0; map_off = a.map_base + 3 * mi;
    }
    if (MAP && map_track) {
        // mapPointUpdate (backend.cpp:1075-1082): the point IS a state, nothing to triangulate, no derivative of it w.r.t. the poses
        if (tid < 3) pfw[tid] = m[map_off + tid];
        if (tid == 0) { s_flag[1] = HV_TRI_HYBRID; s_flag[2] = 0; }
    } else
    if (a.linear) {
        // ---- useLinearTriangulation (triangulation.cpp:146-152, triangulateLinear :820-895): the point closest to every camera
        // ray in closed form, pf = S0^-1 S1 with S0 = sum_i A_i, S1 = sum_i A_i p_i, A_i = I - vn_i vn_i', vn_i the normalised
        // world ray R_i' (ip_i, 1); derivatives d pf / d p_i = S0^-1 A_i, d pf / d q_i through vn_i, d pf / d t through the
        // feature velocities. World frame from the start: no inverse-depth map, no iteration, statuses OK / BEHIND only. ----
        double *Sinv = s_small + 40;                          // the slot the iterative branch uses for M
        if (tid < nt) {
            const double *pose = s_trail + tid * POSE_WORDS;
            double *o = s_it + tid * ITER_WORDS;              // vn[3] |v| A[9] (A p)[3], later the time-shift contribution in [16..18]
            const double ipv[3] = {s_feat[4 * tid], s_feat[4 * tid + 1], 1.0};
            double v[3], A[9], Ap[3];
            mTv3(pose + 3, ipv, v);
            const double nn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            const double vn[3] = {v[0] / nn, v[1] / nn, v[2] / nn};
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) A[3 * r + c] = (r == c ? 1.0 : 0.0) - vn[r] * vn[c];
            mv3(A, pose, Ap);
#pragma unroll
            for (int k = 0; k < 3; ++k) { o[k] = vn[k]; o[13 + k] = Ap[k]; }
            o[3] = nn;
#pragma unroll
            for (int k = 0; k < 9; ++k) o[4 + k] = A[k];
        }
        __syncthreads();
        if (tid == 0) {
            double S0[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, S1[3] = {0, 0, 0};
            for (int i = 0; i < nt; ++i) {
                const double *o = s_it + i * ITER_WORDS;
#pragma unroll
                for (int k = 0; k < 9; ++k) S0[k] += o[4 + k];
#pragma unroll
                for (int k = 0; k < 3; ++k) S1[k] += o[13 + k];
            }
            // 3 x 3 inverse through the cofactors of the first column (what Eigen's fixed-size inverse does)
            const double c0 = S0[4] * S0[8] - S0[5] * S0[7], c1 = S0[2] * S0[7] - S0[1] * S0[8], c2 = S0[1] * S0[5] - S0[2] * S0[4];
            const double invdet = 1.0 / (c0 * S0[0] + c1 * S0[3] + c2 * S0[6]);
            const double inv[9] = { c0 * invdet, c1 * invdet, c2 * invdet,
                                    (S0[5] * S0[6] - S0[3] * S0[8]) * invdet, (S0[0] * S0[8] - S0[2] * S0[6]) * invdet, (S0[2] * S0[3] - S0[0] * S0[5]) * invdet,
                                    (S0[3] * S0[7] - S0[4] * S0[6]) * invdet, (S0[1] * S0[6] - S0[0] * S0[7]) * invdet, (S0[0] * S0[4] - S0[1] * S0[3]) * invdet };
            double pf[3];
            mv3(inv, S1, pf);
#pragma unroll
            for (int k = 0; k < 9; ++k) Sinv[k] = inv[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) pfw[k] = pf[k];
        }
        __syncthreads();
        // item (pose i, c): c < 3 column c of S0^-1 A_i; c = 3..6 the quaternion components; c = 7 the pose's share of d pf / d t
        for (int w = tid; w < nt * 8; w += VT) {
            const int i = w >> 3, c = w & 7;
            const double *pose = s_trail + i * POSE_WORDS;
            double *o = s_it + i * ITER_WORDS;
            double col[3];
            if (c < 3) {
#pragma unroll
                for (int r = 0; r < 3; ++r) col[r] = Sinv[3 * r] * o[4 + c] + Sinv[3 * r + 1] * o[4 + 3 + c] + Sinv[3 * r + 2] * o[4 + 6 + c];
            } else {
                // dv: the change of the un-normalised ray; g = d vn = A dv / |v|; d pf = S0^-1 (sum_k g_k Q_k) (pf - p_i) with
                // Q_k x = e_k (vn . x) + vn x_k, i.e. (sum_k g_k Q_k) x = g (vn . x) + vn (g . x)
                double dv[3];
                if (c < 7) { const double ipv[3] = {s_feat[4 * i], s_feat[4 * i + 1], 1.0}; mTv3(pose + 12 + 9 * (c - 3), ipv, dv); }
                else { const double vel[3] = {s_feat[4 * i + 2], s_feat[4 * i + 3], 0.0}; mTv3(pose + 3, vel, dv); }
                double g[3];
                mv3(o + 4, dv, g);
#pragma unroll
                for (int k = 0; k < 3; ++k) g[k] /= o[3];
                const double x[3] = {pfw[0] - pose[0], pfw[1] - pose[1], pfw[2] - pose[2]};
                const double vx = o[0] * x[0] + o[1] * x[1] + o[2] * x[2], gx = g[0] * x[0] + g[1] * x[1] + g[2] * x[2];
                const double u[3] = {g[0] * vx + o[0] * gx, g[1] * vx + o[1] * gx, g[2] * vx + o[2] * gx};
                mv3(Sinv, u, col);
            }
            if (c < 7) {
#pragma unroll
                for (int r = 0; r < 3; ++r) s_dpfi[r * ncol + 7 * i + c] = col[r];
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) o[16 + r] = col[r];
            }
        }
        __syncthreads();
        if (tid < 3 && a.est_shift) {                         // d pf / d t: the poses' shares in pose order
            double t_ = 0.0;
            for (int i = 0; i < nt; ++i) t_ += s_it[i * ITER_WORDS + 16 + tid];
            s_dpfi[tid * ncol + dDim] = t_;
        }
        if (tid == 0) s_flag[0] = 1;
        __syncthreads();
    } else {
    // ---- triangulateWithTwoCameras between pose 0 and pose ind1 (triangulation.cpp:154-173, 612-716): thread j < 15
    // owns derivative column j (p0 q0 p1 q1 t); every one of them recomputes the small shared part ----
    const int ind1 = a.stereo ? nt / 2 - 1 : nt - 1;
    if (tid < 15) two_camera_start(a, tid, ind1, ncol, dDim, s_trail, s_feat, s_dpfi, pfi, pfw, R0T, scal, s_flag);
    __syncthreads();
    // ---- Gauss-Newton with derivatives (triangulation.cpp:206-343) ----
    // Lanes of the derivative-column phase. Every (pose i, column j) pair contributes through d(pfi)/dx_j (the plain
    // part); a pair also moves C and t of the pose when j belongs to pose i or to pose 0 (the motion part, ~2.5x the
    // flops). Both parts are linear in their inputs, so they are summed separately:
    //   plain part  : G = 2 or 4 adjacent lanes per column walk the poses (stride G), no branch in the loop
    //   motion part : the 7 (2 nt - 1) such pairs are dealt out one (or two) per lane in a single uniform step --
    //                 lane 0 of a column's group takes the column's own pair, the other lanes take the pairs of the
    //                 7 pose-0 columns, whose sums go through LDS in a fixed order.
    // (With a branch inside the loop the wave holding the pose-0 columns took 16.9 k of an iteration's 19 k cycles.)
    // The last wave (VT - 64 ..) forms ETE / Eerror / the step concurrently.
    // The motion pairs are dealt out densely, one per lane, to as few waves as hold them -- and to waves chosen by the SIMD they
    // sit on (wave w runs on SIMD w % 4): the column waves 0 .. 8 already load SIMD 0 with three waves of plain work, so the
    // pairs go to the two idle waves 9, 10 first, then to waves of SIMDs 1 - 3. (r01 gave every column lane a pair, own or
    // dummy: the motion code ran on all 9 column waves at 25 - 75 % lane use.) Pair slot ms -> pose, state component:
    //   ms < 7 nt: pose-0 column  (i = ms / 7, comp = ms % 7, sums through s_p0);  else the own pair of column j = ms - 7 nt + 7.
    constexpr unsigned long long WAVE_POS = VT == 768 ? 0xF10A43297658ull : 0xF42103ull;   // nibble w = position of wave w in that order
    const int wpos = (int)((WAVE_POS >> (4 * (tid >> 6))) & 0xF);
    const int ms = wpos * 64 + (tid & 63), npairs = 14 * nt - 7;
    const bool m_has = wpos != 0xF && ms < npairs, m_p0pair = ms < 7 * nt;
    const int m_col = m_p0pair ? ms : ms - 7 * nt + 7;                       // p0 pairs: u = 7 i + comp;  own pairs: the column
    const int m_i = m_col / 7, m_comp = m_col - 7 * m_i;
    if (m_has) {
        double dC[9], dt[3];
        pose_motion(s_trail, R0T, m_i, m_p0pair ? 0 : m_i, m_comp, dC, dt);
        double *dst = s_mot + ms * MOT_STRIDE;
#pragma unroll
        for (int k = 0; k < 9; ++k) dst[k] = dC[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) dst[9 + k] = dt[k];
    }
    VU_STAMP(2);
    for (int it = 0; it < a.gn_iters; ++it) {
        if (it < 6) VU_STAMP(3 + 4 * it);
        if (tid < nt) {                                       // per-pose quantities of this iteration
            const double *cur = s_trail + tid * POSE_WORDS;
            double *o = s_it + tid * ITER_WORDS;
            double C[9], t[3], d[3], h[3];
            mm3(cur + 3, R0T, C);
#pragma unroll
            for (int k = 0; k < 3; ++k) d[k] = p0[k] - cur[k];
            mv3(cur + 3, d, t);
            const double pfiab[3] = {pfi[0], pfi[1], 1.0};
            mv3(C, pfiab, h);
#pragma unroll
            for (int k = 0; k < 3; ++k) h[k] += pfi[2] * t[k];
            const double ih2 = 1.0 / h[2], ih2sq = ih2 * ih2;
#pragma unroll
            for (int k = 0; k < 9; ++k) o[k] = C[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) { o[9 + k] = t[k]; o[12 + k] = h[k]; o[23 + k] = d[k]; }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int c = 0; c < 2; ++c) o[15 + 3 * r + c] = -ih2 * C[3 * r + c] + h[r] * ih2sq * C[6 + c];
                o[15 + 3 * r + 2] = -t[r] * ih2 + h[r] * ih2sq * t[2];
                o[21 + r] = s_feat[4 * tid + r] - h[r] * ih2;
            }
        }
        __syncthreads();
        if (it < 6) VU_STAMP(4 + 4 * it);
        if (tid >= VT - 64) {                                 // the last wave owns no derivative column: it forms ETE (9),
            const int k = tid - (VT - 64);                    // Eerror (3) and error2 concurrently, lane k < 13 summing entry k
            // entry k = sum over poses of o[x0]*o[y0] + o[x0+dx]*o[y0+dy] on the pose record (E rows at 15 and 18, err at 21)
            const int r = k / 3, c = k - 3 * r;
            const int x0 = k < 9 ? 15 + r : k < 12 ? 15 + (k - 9) : 21, y0 = k < 9 ? 15 + c : 21;
            const int dx = k < 12 ? 3 : 1, dy = k < 9 ? 3 : 1;
            double acc = 0.0;
            if (k < 13)
                for (int i = 0; i < nt; ++i) {
                    const double *o = s_it + i * ITER_WORDS;
                    acc += o[x0] * o[y0] + o[x0 + dx] * o[y0 + dy];
                }
            double ETE[9], Ee[3];
#pragma unroll
            for (int q = 0; q < 9; ++q) ETE[q] = __shfl(acc, q);
#pragma unroll
            for (int q = 0; q < 3; ++q) Ee[q] = __shfl(acc, 9 + q);
            const double e2 = __shfl(acc, 12);
            if (k == 0) {
                double Xl[9], st[3];
                inv3sym(ETE, Xl);
                mv3(Xl, Ee, st);
#pragma unroll
                for (int q = 0; q < 9; ++q) X[q] = Xl[q];
#pragma unroll
                for (int q = 0; q < 3; ++q) step[q] = st[q];
                scal[0] = e2;
                scal[1] = 1.0 / (norm1_3(ETE) * norm1_3(Xl));
            }
        }
        if (it < 6) VU_STAMP(5 + 4 * it);
        // derivative columns: dEerror_j and dETE_j accumulated over the poses, with the OLD pfi (:236-312).
        // PLAIN part (the change of pfi seen by every pose): dh_i = [C_i(:, 0:2) | t_i] d_j with d_j = column j of dpfi, and everything
        // downstream of dh is linear in it -- so the sum over the poses is taken ONCE, of the linear maps, not per column:
        //   task (pose i, unit vector u)  ->  the 3 + 6 numbers (dEe, upper dM) pair_sums gives for dh = column u of [C_i | t_i]   (3 nt lanes)
        //   L[u][9] = sum over the poses                                                                                         (27 lanes)
        //   column j:  (dEe_j, dM_j) = L' d_j   (+ the velocity term c_t for the time-shift column)                             (27 FMAs)
        // r01 walked the poses per column: nt x ~90 f64 instructions in each of 9 waves, the critical path of the kernel.
        if (tid >= 64 && tid < 64 + 3 * nt) {
            const int task = tid - 64, i = task / 3, u = task - 3 * i;
            const double *o = s_it + i * ITER_WORDS;
            const double dh[3] = {u < 2 ? o[u] : o[9], u < 2 ? o[3 + u] : o[10], u < 2 ? o[6 + u] : o[11]};
            double e3[3] = {0, 0, 0}, m9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            pair_sums<0>(o, dh, nullptr, nullptr, 0.0, 0.0, e3, m9);
            double *dst = s_lin + task * 9;
            dst[0] = e3[0]; dst[1] = e3[1]; dst[2] = e3[2];
            dst[3] = m9[0]; dst[4] = m9[1]; dst[5] = m9[2]; dst[6] = m9[4]; dst[7] = m9[5]; dst[8] = m9[8];
        }
        if (m_has) {                                                                    // motion part: one pair per lane, one code path
            const double *o = s_it + m_i * ITER_WORDS;
            double dC[9], dt[3], dh[3], e3[3] = {0, 0, 0}, m9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            const double *mot = s_mot + ms * MOT_STRIDE;
#pragma unroll
            for (int k = 0; k < 9; ++k) dC[k] = mot[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) dt[k] = mot[9 + k];
#pragma unroll
            for (int r = 0; r < 3; ++r) dh[r] = (dC[3 * r] * pfi[0] + dC[3 * r + 1] * pfi[1] + dC[3 * r + 2]) + pfi[2] * dt[r];
            pair_sums<1>(o, dh, dC, dt, 0.0, 0.0, e3, m9);
            double *dst = m_p0pair ? s_p0 + (m_comp * MAXP + m_i) * 9 : s_own + m_col * 9;
            dst[0] = e3[0]; dst[1] = e3[1]; dst[2] = e3[2];
            dst[3] = m9[0]; dst[4] = m9[1]; dst[5] = m9[2]; dst[6] = m9[4]; dst[7] = m9[5]; dst[8] = m9[8];
        }
        __syncthreads();                                      // X, step, error2 are published; everybody is done with the old pfi
        if (it < 6) VU_STAMP(6 + 4 * it);
        // 93 sums over the poses, LPS adjacent lanes each (a lane per sum walked nt dependent LDS reads + adds: 1.4 k cycles):
        //   sigma < 63       pose-0 columns: entry e of column c, motion part            -> totals at s_p0 + 7 MAXP 9
        //   63 <= sigma < 90 L[u][e]: the linear maps of the plain part                  -> s_lin + 3 MAXP 9
        //   90 <= sigma < 93 c_t = sum_i E_i' vel_i: the constant of the time-shift column -> s_lin + 3 MAXP 9 + 27
        constexpr int LPS = VT >= 768 ? 4 : 2;
        if (tid >= 64 && tid < 64 + 93 * LPS) {
            const int g = tid - 64, sigma = g / LPS, part = g - sigma * LPS;
            double acc = 0.0;
            if (sigma < 63) {
                const int c = sigma / 9, e = sigma - 9 * c;
                for (int q = part; q < nt; q += LPS) acc += s_p0[(c * MAXP + q) * 9 + e];
            } else if (sigma < 90) {
                const int u = (sigma - 63) / 9, e = sigma - 63 - 9 * u;
                for (int q = part; q < nt; q += LPS) acc += s_lin[(3 * q + u) * 9 + e];
            } else {
                const int r = sigma - 90;
                for (int q = part; q < nt; q += LPS)
                    acc += s_it[q * ITER_WORDS + 15 + r] * s_feat[4 * q + 2] + s_it[q * ITER_WORDS + 18 + r] * s_feat[4 * q + 3];
            }
#pragma unroll
            for (int o = 1; o < LPS; o <<= 1) acc += __shfl_xor(acc, o);
            if (part == 0) {
                if (sigma < 63) s_p0[7 * MAXP * 9 + sigma] = acc;
                else s_lin[3 * MAXP * 9 + sigma - 63] = acc;
            }
        }
        __syncthreads();
        if (tid < ncol && !(tid == dDim && !a.est_shift)) {   // :324-328: d(A^-1) = -A^-1 dA A^-1, one lane per column
            const int j = tid;
            const double *Lm = s_lin + 3 * MAXP * 9;
            const double d0 = s_dpfi[j], d1 = s_dpfi[ncol + j], d2 = s_dpfi[2 * ncol + j];
            double v[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) v[e] = (Lm[e] * d0 + Lm[9 + e] * d1) + Lm[18 + e] * d2;
            double dEe[3] = {v[0], v[1], v[2]}, dM[9] = {v[3], v[4], v[5], v[4], v[6], v[7], v[5], v[7], v[8]};
            // + the motion sums: the pose-0 totals (columns 0 .. 6), the column's own pair (regular columns), the velocity term (time shift)
            const double *add = j < 7 ? s_p0 + 7 * MAXP * 9 + 9 * j : s_own + j * 9;
            if (j != dDim) {
#pragma unroll
                for (int k = 0; k < 3; ++k) dEe[k] += add[k];
                dM[0] += add[3]; dM[1] += add[4]; dM[2] += add[5]; dM[4] += add[6]; dM[5] += add[7]; dM[8] += add[8];
                dM[3] = dM[1]; dM[6] = dM[2]; dM[7] = dM[5];
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) dEe[k] += Lm[27 + k];
            }
            double t1[3], t2[3], t3[3];
            mv3(dM, step, t1);
            mv3(X, t1, t2);
            mv3(X, dEe, t3);
#pragma unroll
            for (int r = 0; r < 3; ++r) s_dpfi[r * ncol + j] += t2[r] - t3[r];
        }
        if (tid == 0) {                                       // :316-342
#pragma unroll
            for (int k = 0; k < 3; ++k) pfi[k] -= step[k];
            const double J = 0.5 * scal[0] / (a.conv_r * a.conv_r), Jd = fabs((J - scal[2]) / J);
            scal[2] = J;
            if (Jd < a.conv_threshold) s_flag[0] = 1;
        }
        __syncthreads();
        if (s_flag[0]) break;
    }
    }   // iterative branch
    VU_STAMP(27);
    // ---- status, back to world coordinates (:345-392) ----
    double *M = s_small + 40, *pf0 = s_small + 49;           // R0T * dpf0_dpfi, the point in the frame of pose 0
    if (tid == 0 && !(MAP && map_track)) {
        int status = HV_TRI_OK;
        if (a.linear) { /* pfw and the world-frame derivative columns are in place */ }
        else if (!s_flag[0]) status = HV_TRI_NO_CONVERGENCE;
        else if (scal[1] < a.rcond_threshold) status = HV_TRI_BAD_COND;
        if (status == HV_TRI_OK && !a.linear) {
            double d[9], q[3], w[3], Ml[9];
            inverse_depth(pfi, q, d);
            mv3(R0T, q, w);
#pragma unroll
            for (int k = 0; k < 3; ++k) { pfw[k] = w[k] + p0[k]; pf0[k] = q[k]; }
            if (pfw[0] == p0[0] && pfw[1] == p0[1] && pfw[2] == p0[2]) status = HV_TRI_UNKNOWN_PROBLEM;
            mm3(R0T, d, Ml);
#pragma unroll
            for (int k = 0; k < 9; ++k) M[k] = Ml[k];
        }
        s_flag[1] = status;
        s_flag[2] = 0;                                        // behind any camera
    }
    __syncthreads();
    status = s_flag[1];
    if (status == HV_TRI_OK) {
        if (tid < ncol && !a.linear) {
            const int j = tid;
            double u[3] = {0, 0, 0}, v[3];
            if (j >= 3 && j < 7) mTv3(s_trail + 12 + 9 * (j - 3), pf0, u);                // dR0T * pf0
            const double cur[3] = {s_dpfi[j], s_dpfi[ncol + j], s_dpfi[2 * ncol + j]};
            mv3(M, cur, v);
#pragma unroll
            for (int r = 0; r < 3; ++r) s_dpfi[r * ncol + j] = u[r] + v[r] + (j == r ? 1.0 : 0.0);      // dp0 = e_j for j < 3 (r < 3, so j == r implies it)
        }
        if (tid < nt) {                                       // isBehind (:54-60)
            const double *cur = s_trail + tid * POSE_WORDS;
            const double d[3] = {pfw[0] - cur[0], pfw[1] - cur[1], pfw[2] - cur[2]};
            if (cur[9] * d[0] + cur[10] * d[1] + cur[11] * d[2] < 0) atomicOr(&s_flag[2], 1);
        }
        __syncthreads();
        if (s_flag[2]) status = HV_TRI_BEHIND;
    }
    if (!(MAP && map_track)) {   // backend.cpp:1098-1102: depth window on whatever point the triangulation left behind
        const double dx = pfw[0] - p0[0], dy = pfw[1] - p0[1], dz = pfw[2] - p0[2], depth = sqrt(dx * dx + dy * dy + dz * dz);
        if (depth < a.min_dist || depth > a.max_dist) status = HV_TRI_BAD_DEPTH;
    }
    with_derivatives = status == HV_TRI_OK;
    // backend.cpp:1108-1119: per-pose derivative blocks, the two cameras of a pose summed
    if (with_derivatives)
        for (int i = tid; i < n * 21; i += VT) {
            const int k = i / 21, e = i - 21 * k, r = e / 7, c = e - 7 * r;              // [k][r][c]: c < 3 position, else quaternion
            double v = s_dpfi[r * ncol + 7 * k + c];
            if (a.stereo) v += s_dpfi[r * ncol + 7 * (k + n) + c];
            s_dpf[i] = v;
        }
    VU_STAMP(28);
    // ---- prepareVisualUpdate (triangulation.cpp:897-987), full-width H (batch layout of the update kernel) ----
    // per trail pose: dip*R (2x3), the own-orientation block dip*dRpt (2x4), f, depth class  -> s_it[i][0..16]
    if (tid < nt) {
        const double *pose = s_trail + tid * POSE_WORDS;
        double *o = s_it + tid * ITER_WORDS;
        const double pt[3] = {pfw[0] - pose[0], pfw[1] - pose[1], pfw[2] - pose[2]};
        double pfc[3], ipH[3], dip[9];
        mv3(pose + 3, pt, pfc);
        inverse_depth(pfc, ipH, dip);
        o[16] = pfc[2] == 0 ? 1.0 : pfc[2] < 0 ? 2.0 : 0.0;
        o[14] = ipH[0]; o[15] = ipH[1];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) o[3 * r + c] = dip[3 * r] * pose[3 + c] + dip[3 * r + 1] * pose[6 + c] + dip[3 * r + 2] * pose[9 + c];
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
            const double *dR = pose + 12 + 9 * jq;
            double a1[3], b1[3], b2[3];
            mv3(dR, pt, a1);
            mTv3(dR, pose + 48, b1);
            mv3(pose + 3, b1, b2);
#pragma unroll
            for (int r = 0; r < 2; ++r) o[6 + 4 * r + jq] = dip[3 * r] * (a1[0] + b2[0]) + dip[3 * r + 1] * (a1[1] + b2[1]) + dip[3 * r + 2] * (a1[2] + b2[2]);
        }
    }
    __syncthreads();
    VU_STAMP(29);
    }   // !REC: the front
    // the point's derivative w.r.t. the time shift: the last column of dpfi (record-fed builds keep just that column)
    auto sft_col = [&](int c) -> double { return REC ? s_dpfi[c] : s_dpfi[c * ncol + dDim]; };
    const int rows = 2 * nt;
    if constexpr (FUSED != 0) {
        // ---- prepareVisualUpdate in compact form + visualTrackOutlierCheck on the active columns (see VuPrepareArgs::fused) ----
        int prep = 0;
        for (int i = 0; i < nt && prep == 0; ++i) prep = (int)s_it[i * ITER_WORDS + 16];   // first failing pose decides (:920-927); uniform
        const double pf_out[3] = {pfw[0], pfw[1], pfw[2]};
        if (!(status == HV_TRI_OK && prep == 0)) {               // nothing to gate (uniform): the track is final
            if (tid == 0) {
                st_out[0] = status; st_out[1] = prep;
                if (a.active) a.active[rec] = 0;
                if (a.gate_status) a.gate_status[rec] = 1;        // VuOutlierStatus::NOT_COMPUTED
                if (a.chi2) a.chi2[rec] = 0.0;
                if (a.spec_tracks > 0) a.epoch[rec] = a.success_counter[b];
#pragma unroll
                for (int k = 0; k < 3; ++k) a.pf[3 * rec + k] = pf_out[k];
            }
            return;
        }
        if (tid == 0) {                                           // what does not depend on the gate leaves now (nothing to keep in registers)
            st_out[0] = HV_TRI_OK; st_out[1] = 0;
            if (a.active) a.active[rec] = 1;
#pragma unroll
            for (int k = 0; k < 3; ++k) a.pf[3 * rec + k] = pf_out[k];
        }
        // FUSED 3 (the long build, which also serves every record of a speculative pass over long tracks): the gate works from the FACTORS
        // of the Jacobian (ekf_device.hpp structured_S, r04), copied out of the Gauss-Newton arrays below before [S; v'] overwrites them;
        // the compact Jacobian itself only goes to HBM, for the update of an inlier. FUSED 1 (<= 48 rows) keeps r03's form -- the compact
        // Jacobian staged in LDS for sparse_gate's two dense MFMA products: measured on one box (scripts/r04_run21.sh, 1024 tracks per
        // launch) the factor form takes 321 against 372 us at 21 stereo poses and 231 against 240 at 16, but 152 against 148 at 10.
        constexpr bool STRUCT = FUSED == 3;
        constexpr bool STAGED = FUSED == 1;                       // the compact Jacobian is also staged in LDS for the gate of this launch
        const int na = 7 * n + 1, na4 = (na + 3) & ~3, ti = (rows + 15) >> 4;
        int Rs = rows + 1;                                        // column stride of [S; v']: 15 or 17 mod 32 doubles (bank spread)
        while ((Rs & 31) != 15 && (Rs & 31) != 17) Rs++;
        constexpr int T_CAP = FUSED == 3 ? Lay::LONG_T : Lay::T_DOUBLES, REGION = FUSED == 3 ? Lay::LONG_HS : Lay::HS_DOUBLES;
        if (STRUCT && Rs * rows > T_CAP) Rs = rows + 1 + (((rows + 1) & 1) ? 0 : 1);      // the longest tracks: any odd stride that fits (84 rows: 85)
        const int nrp = 16 * ti;                                  // (STAGED: rows per staged column)
        // region behind [S; v'] (the dead motion / linear-map arrays): the staged Jacobian, or the factors, their products and G
        double *Hs = vu_lds + (FUSED == 3 ? Lay::LONG_T : Lay::P0);
        const int f4s = na, nslots_g = VT / ((na + 63) & ~63);
        double *f_O4 = Hs, *f_DV = f_O4 + 4 * rows, *f_F4 = f_DV + 7 * rows, *f_WF = f_F4 + 4 * f4s, *f_FA = f_WF + 4 * na,
               *f_DWF = f_FA + 4 * rows, *f_FWF = f_DWF + 4 * rows, *f_WFp = f_FWF + 16, *f_G = f_WFp + 3 * na * nslots_g;
        const int g_cap = REGION - (int)(f_G - Hs);
        int *s_acol = s_flag + 4;
        if (tid < na) {                                           // compact column u -> state column: pose q's position / orientation, then SFT
            int col = SFT;
            if (tid < 7 * n) {
                const int q = tid / 7, comp = tid - 7 * q;
                int ip, io;
                pos_ori(s_idx[q], ip, io);
                col = comp < 3 ? ip + comp : io + comp - 3;
            }
            s_acol[tid] = col;                                    // (read by the gate, after the barriers below)
            a.acol[rec * a.na_max + tid] = col;
        }
        if constexpr (STRUCT) {
            // Hc = Dp + O4 F4 (see structured_S): observation tid's two rows of O4 and of Dp's 7 own-pose values, column tid of F4
            if (tid < nt) {
                const double *o = s_it + tid * ITER_WORDS;
                double *o4 = f_O4 + 8 * tid, *dv = f_DV + 14 * tid;
#pragma unroll
                for (int c = 0; c < 3; ++c) { o4[c] = o[c]; o4[4 + c] = o[3 + c]; }
                o4[3] = a.est_shift ? -s_feat[4 * tid + 2] : 0.0; o4[7] = a.est_shift ? -s_feat[4 * tid + 3] : 0.0;
#pragma unroll
                for (int comp = 0; comp < 7; ++comp) {
                    dv[comp] = comp < 3 ? -o[comp] : o[6 + comp - 3];
                    dv[7 + comp] = comp < 3 ? -o[3 + comp] : o[10 + comp - 3];
                }
            }
            if (tid < na) {
                const bool sft = tid == 7 * n;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double v = 0.0;
                    if (!sft) { const int k = tid / 7, comp = tid - 7 * k; v = s_dpf[21 * k + comp + 7 * c]; }
                    else if (a.est_shift) v = sft_col(c);
                    f_F4[c * f4s + tid] = v;
                }
                f_F4[3 * f4s + tid] = sft ? 1.0 : 0.0;
            }
        }
        double *Hc = a.Hc + rec * (size_t)rows_max * a.na_max;    // record stride: the longest track; leading dimension: this track's rows
        // (the measurement is requested in front of the Jacobian's stores: behind them its round trip would queue up with theirs)
        double yv[2] = {0.0, 0.0};
        if (tid < nt && a.y) { yv[0] = a.y[rec * rows_max + 2 * tid]; yv[1] = a.y[rec * rows_max + 2 * tid + 1]; }
        // work item = (compact column u, observation i), i fastest: the two rows of an observation are one 16-byte store. STAGED: u < na4,
        // i < nrp / 2 -- every element of the staged Hs is written exactly once, zero padding in rows >= 2 nt and columns >= na included
        // (STRUCT: the compact Jacobian is only needed by the update of an INLIER -- a quarter of the gates at the reference's inlier rate --
        //  and the factors it is made of survive the gate in LDS: it is built behind the gate, for inliers only; 24 k of a 21-pose gate's 176 k cycles)
        const int wi = STAGED ? nrp >> 1 : nt, n_items = (STRUCT && a.defer_h) ? 0 : (STAGED ? na4 : na) * wi;
        const unsigned inv_wi = (unsigned)((0x100000000ull + (unsigned)wi - 1) / (unsigned)wi);   // w / wi = umulhi(w, ceil(2^32 / wi))
        for (int w = tid; w < n_items; w += VT) {
            const int u = (int)__umulhi((unsigned)w, inv_wi), i = w - u * wi;
            double h0 = 0.0, h1 = 0.0;
            if (u < na && i < nt) {
                const double *o = s_it + i * ITER_WORDS;
                if (u < 7 * n) {
                    const int k = u / 7, comp = u - 7 * k;
                    if (k == (i >= n ? i - n : i)) {                                       // own pose: :946-953
                        if (comp < 3) { h0 = -o[comp]; h1 = -o[3 + comp]; }
                        else { h0 = o[6 + comp - 3]; h1 = o[10 + comp - 3]; }
                    }
                    const double *dp = s_dpf + 21 * k + comp;                              // :955-964
                    h0 += o[0] * dp[0] + o[1] * dp[7] + o[2] * dp[14];
                    h1 += o[3] * dp[0] + o[4] * dp[7] + o[5] * dp[14];
                } else if (a.est_shift) {                                                  // :965-967
                    const double sft_t0 = sft_col(0), sft_t1 = sft_col(1), sft_t2 = sft_col(2);
                    h0 = o[0] * sft_t0 + o[1] * sft_t1 + o[2] * sft_t2 - s_feat[4 * i + 2];
                    h1 = o[3] * sft_t0 + o[4] * sft_t1 + o[5] * sft_t2 - s_feat[4 * i + 3];
                }
                *reinterpret_cast<double2 *>(Hc + (size_t)u * rows + 2 * i) = double2{h0, h1};
            }
            if constexpr (STAGED) *reinterpret_cast<double2 *>(Hs + (size_t)u * nrp + 2 * i) = double2{h0, h1};
        }
        double vres[2] = {0.0, 0.0};
        if (tid < nt) {
            const double *o = s_it + tid * ITER_WORDS;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const size_t e = rec * rows_max + 2 * tid + r;
                if (a.f) a.f[e] = o[14 + r];
                vres[r] = yv[r] - o[14 + r];
                a.v[e] = vres[r];
            }
        }
        if constexpr (FUSED == 2) {                               // the gate is the next launch: this track is "prepared, not gated yet"
            if (tid == 0) {
                if (a.gate_status) a.gate_status[rec] = 1;        // VuOutlierStatus::NOT_COMPUTED until ekf_sparse_gate_kernel has run
                if (a.spec_tracks > 0) a.epoch[rec] = a.success_counter[b];
            }
            return;
        }
        lds_barrier();                                            // everything but the factor copies / s_acol is dead from here on (the stores above stay in flight)
        VU_STAMP(31);
        double *T = vu_lds;
        // (structured_S serves the poses in one group where A = na x rows fits g_cap, else in two: the half must fit)
        const int g_half = na * ((2 * ncam * ((n + 1) >> 1)) | 1), g_all = na * (rows | 1);
        if (STRUCT && (Rs * rows > T_CAP || g_cap < 832 + VT / 64 || (g_all > g_cap && g_half > g_cap))) {   // (cannot happen: the launcher admits what the carve holds) not gated, never applied
            if (tid == 0 && a.gate_status) a.gate_status[rec] = 1;
            return;
        }
        for (int i = tid; i < Rs * rows; i += VT) T[i] = 0.0;
        lds_barrier();
        if (tid < nt) { T[(size_t)(2 * tid) * Rs + rows] = vres[0]; T[(size_t)(2 * tid + 1) * Rs + rows] = vres[1]; }
        VU_STAMP(32);
        const double *Pb = a.P + (size_t)b * N * N;
        // adaptive thresholds of the frame loop (backend.cpp:1192-1193): this filter's multiplier of chiOutlierR and rmseThreshold
        const double gscale = a.gate_scale ? a.gate_scale[b] : 1.0;
        const double rd_eff = a.rd_gate * gscale * gscale;
        if (a.rmse_thr >= 0.0) {                                  // (uniform) visualTrackOutlierCheck's early RMSE test, ekf.cpp:797-801
            lds_barrier();                                        // v is in row `rows` of T
            double s2 = 0.0;
            for (int c = 0; c < rows; ++c) { const double vc = T[(size_t)c * Rs + rows]; s2 += vc * vc; }    // every thread: same order, same value
            if (sqrt(s2 / rows) > a.rmse_thr * gscale) {
                if (tid == 0) {
                    if (a.gate_status) a.gate_status[rec] = 2 /*RMSE*/;
                    if (a.chi2) a.chi2[rec] = 0.0;
                    if (a.gate_scale) a.gate_scale[b] = gscale * a.growth;
                    if (a.spec_tracks > 0) a.epoch[rec] = a.success_counter[b];
                }
                return;
            }
        }
        double chi;
        if constexpr (STRUCT) {
            lds_barrier();                                        // (v' is in row `rows` of the zeroed T)
            structured_S<VT, (VT > VT_THROUGHPUT)>(Pb, N, s_acol, na, n, ncam, rows, f_O4, f_DV, f_F4, f4s, f_WF, f_WFp, f_FA, f_DWF, f_FWF, f_G, g_cap, T, Rs, g_vu_stamp + 1);
            VU_STAMP(33);
            chi = gate_factor_chi2<VT>(T, Rs, rows, rd_eff, a.noise_scale, f_G, g_vu_stamp + 33);
            // every thread holds the same chi2: the inlier's compact Jacobian Hc = Dp + O4 F4, from the factor copies (intact: the gate's
            // scratch lies behind them)
            if (a.defer_h && chi < 1e300 && !((rows < HV_CHI2INV95_N) && chi > d_chi2inv95[rows])) {
                const unsigned inv_nt = (unsigned)((0x100000000ull + (unsigned)nt - 1) / (unsigned)nt);
                for (int w = tid; w < na * nt; w += VT) {
                    const int u = (int)__umulhi((unsigned)w, inv_nt), i = w - u * nt;
                    const int k = u / 7, comp = u - 7 * k;
                    const bool own = u < 7 * n && k == (i >= n ? i - n : i);
                    const double *o4 = f_O4 + 8 * i, *dv = f_DV + 14 * i;
                    double h0 = o4[0] * f_F4[u] + o4[1] * f_F4[f4s + u] + o4[2] * f_F4[2 * f4s + u] + o4[3] * f_F4[3 * f4s + u];
                    double h1 = o4[4] * f_F4[u] + o4[5] * f_F4[f4s + u] + o4[6] * f_F4[2 * f4s + u] + o4[7] * f_F4[3 * f4s + u];
                    if (own) { h0 += dv[comp]; h1 += dv[7 + comp]; }
                    *reinterpret_cast<double2 *>(Hc + (size_t)u * rows + 2 * i) = double2{h0, h1};
                }
            }
        } else {
            if (ti == 1)      chi = sparse_gate<1, VT, false>(Pb, N, s_acol, na, Hs, T, Rs, rows, rd_eff, a.noise_scale, Hs, g_vu_stamp + 33);
            else if (ti == 2) chi = sparse_gate<2, VT, false>(Pb, N, s_acol, na, Hs, T, Rs, rows, rd_eff, a.noise_scale, Hs, g_vu_stamp + 33);
            else              chi = sparse_gate<3, VT, false>(Pb, N, s_acol, na, Hs, T, Rs, rows, rd_eff, a.noise_scale, Hs, g_vu_stamp + 33);
        }
        if (tid == 0) {
            const bool broken = !(chi < 1e300);                   // non-positive pivot: reported as CHI2 (ekf_update_kernel phase D)
            const int outlier = broken || ((rows < HV_CHI2INV95_N) ? (chi > d_chi2inv95[rows]) : 0);
            if (a.gate_status) a.gate_status[rec] = outlier ? 3 /*CHI2*/ : 0 /*INLIER*/;
            if (!outlier && a.inl_list) a.inl_list[atomicAdd(a.inl_count, 1)] = (int)rec;
            if (outlier && a.gate_scale) a.gate_scale[b] = gscale * a.growth;
            if (a.chi2) a.chi2[rec] = chi;
            if (a.spec_tracks > 0) a.epoch[rec] = a.success_counter[b];
        }
        return;
    }
    double *H = a.H + rec * rows_max * N;                                    // record stride: the longest track; leading dimension: this track's rows
    // which pose of the track (if any) owns state column c, and which of its 7 components: once per column
    int *s_colmap = reinterpret_cast<int *>(s_p0);                          // s_p0 is free after the Gauss-Newton loop
    for (int c = tid; c < N; c += VT) {
        int code = -1;
        for (int q = 0; q < n; ++q) {
            int ip, io;
            pos_ori(s_idx[q], ip, io);
            if (c >= ip && c < ip + 3) code = 8 * q + (c - ip);
            else if (c >= io && c < io + 4) code = 8 * q + 3 + (c - io);
        }
        s_colmap[c] = code;
    }
    __syncthreads();
    // work item = (state column c, trail pose i), i fastest: the 2 * nt rows of a column are contiguous in the column-major
    // H, so consecutive lanes write consecutive 16-byte pairs (a thread per column wrote 16 bytes every 2 * nt * 8)
    const unsigned inv_nt = (unsigned)((0x100000000ull + (unsigned)nt - 1) / (unsigned)nt);   // w / nt = umulhi(w, ceil(2^32 / nt)) for w nt < 2^32
    for (int w = tid; w < N * nt; w += VT) {
        const int c = (int)__umulhi((unsigned)w, inv_nt), i = w - c * nt;
        const int code = s_colmap[c], k = code >> 3, comp = code & 7;
        const bool sft = c == 19 && with_derivatives && a.est_shift;
        const double *o = s_it + i * ITER_WORDS;
        double h0 = 0.0, h1 = 0.0;
        if (code >= 0) {
            if (k == i % n) {                                                              // own pose: :946-953
                if (comp < 3) { h0 = -o[comp]; h1 = -o[3 + comp]; }
                else { h0 = o[6 + comp - 3]; h1 = o[10 + comp - 3]; }
            }
            if (with_derivatives) {                                                        // :955-964
                const double *dp = s_dpf + 21 * k + comp;
                h0 += o[0] * dp[0] + o[1] * dp[7] + o[2] * dp[14];
                h1 += o[3] * dp[0] + o[4] * dp[7] + o[5] * dp[14];
            }
        } else if (sft) {                                                                  // :965-967
            const double t0 = s_dpfi[dDim], t1 = s_dpfi[ncol + dDim], t2 = s_dpfi[2 * ncol + dDim];
            h0 = o[0] * t0 + o[1] * t1 + o[2] * t2 - s_feat[4 * i + 2];
            h1 = o[3] * t0 + o[4] * t1 + o[5] * t2 - s_feat[4 * i + 3];
        }
        if (MAP && map_track && c >= map_off && c < map_off + 3) { h0 += o[c - map_off]; h1 += o[3 + c - map_off]; }   // :982-984: dip R
        *reinterpret_cast<double2 *>(H + (size_t)c * rows + 2 * i) = double2{h0, h1};
    }
    if (tid < nt) {
        const double *o = s_it + tid * ITER_WORDS;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const size_t e = rec * rows_max + 2 * tid + r;
            if (a.f) a.f[e] = o[14 + r];
            a.v[e] = (a.y ? a.y[e] : 0.0) - o[14 + r];
        }
    }
    VU_STAMP(30);
    if (tid == 0) {
        int prep = 0;
        for (int i = 0; i < nt && prep == 0; ++i) prep = (int)s_it[i * ITER_WORDS + 16];   // first failing pose decides (:920-927)
        st_out[0] = status; st_out[1] = prep;
        if (a.active) a.active[rec] = ((status == HV_TRI_OK || (MAP && map_track)) && prep == 0) ? 1 : 0;      // backend.cpp:1151-1152
        if (a.gate_status) a.gate_status[rec] = 1;                                     // VuOutlierStatus::NOT_COMPUTED
        if (a.spec_tracks > 0) a.epoch[rec] = a.success_counter[b];                    // (written last: every thread has read it by now)
#pragma unroll
        for (int k = 0; k < 3; ++k) a.pf[3 * rec + k] = pfw[k];
    }
}

// ---------------------------------------------------------------------------------------------
// Split form (r06): the triangulation front as its OWN kernel. r03 .. r05's fused kernels run pose trail ->
// two-camera start -> Gauss-Newton -> prepareVisualUpdate -> gate as ~50 barrier-separated phases of one 384- / 768-thread workgroup
// that holds 80 / 158 KB of LDS throughout (waves parked 70 %; for the typical 6-pose track 39 k of 74 k cycles pass before
// prepareVisualUpdate starts, profiles/r04/phase_stamps_one_track_visit.txt). Here NT = 64 threads -- ONE wavefront, whose LDS traffic
// is ordered by the hardware: the barriers below cost a wave nothing -- or NT = 256 (the long class: four wavefronts cut the longest
// track's chain) run everything up to the per-pose part of prepareVisualUpdate in 19 .. 49 KB of LDS, 3 .. 5 tracks per CU, and leave a
// FACTOR RECORD per track (VuPrepareArgs::tri_rec): the 17 per-pose values of prepareVisualUpdate, the summed point derivatives, the
// time-shift column. The gate kernels (from_rec builds of vu_prepare_body) start from it.
//
// The derivative sums are the same as vu_prepare_body's with the motion pairs taken by TYPE, each type a loop of its own (a wave's
// lanes run one code path) and each evaluated from what it really needs instead of through pose_motion's 0 / 1 weights:
//   P  own position column (pose i >= 1, component c):  dC = 0, dt = -R_i(:, c). The pose-0 column c has dt = +R_i(:, c) for the same
//      pose, and pair_sums is linear in (dh, dC, dt): ONE evaluation r serves both -- +r into the pose-0 total, -r for the own column
//      (exact: every operation of pair_sums is odd in its inputs) -- 11 nt - 7 pair evaluations per iteration instead of 14 nt - 7;
//   C  pose-0 quaternion column (pose i, q):  dC = R_i dR0_q' (+ dR0_q R0' for i = 0), dt = -R_i (dR0_q' base_0) (0 for i = 0);
//   Q  own quaternion column (pose i >= 1, q):  dC = dR_iq R0', dt = dR_iq (p_0 - p_i) + R_i (dR_iq' base_i);
// nothing is parked in LDS across the iterations (s_mot of the fused kernels: 13 doubles per pair). A lane that evaluates an own pair
// updates that column of dpfi right away (X, the step and the linear maps L of the plain part are in place by then), so the own sums
// never meet LDS either; the pose-0 sums and the plain part's maps go through one scratch area in a fixed order (deterministic).
// Reference: triangulation.cpp:65-103,120-407,612-716,897-947; backend.cpp:1098-1119.
// ---------------------------------------------------------------------------------------------
constexpr int TRI_ITW = 28;         // pose record of an iteration: ITER_WORDS + 1 / h_z (+ pad)
struct TriLds {                       // run-time carve of the dynamic LDS (doubles) for tracks of up to ntm camera poses
    int trail, it, dpfi, feat, small, scr, lsum, tot, ints, total;
    __host__ __device__ TriLds(int ntm)
    {
        trail = 0; it = trail + ntm * POSE_WORDS; dpfi = it + ntm * TRI_ITW; feat = dpfi + 3 * (7 * ntm + 1);
        small = feat + 4 * ntm; scr = small + 72; lsum = scr + 7 * ntm * 9; tot = lsum + 32; ints = tot + 64;
        total = ints + (MAXNP + 3 + 4 + 1) / 2 + 1;
    }
};

// sum of n terms base[(first + k) * stride], k = part, part + parts, ... -- four loads in flight per step (a lane per sum walking
// dependent LDS reads is a chain of ~130-cycle round trips: 5.6 k cycles for the 42 terms of a 21-pose track, profiles/r06/tri_stamps.txt)
__device__ __forceinline__ double strided_sum(const double *base, int stride, int n, int part, int parts)
{
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int k = part;
    for (; k + 3 * parts < n; k += 4 * parts) {
        a0 += base[(size_t)k * stride]; a1 += base[(size_t)(k + parts) * stride];
        a2 += base[(size_t)(k + 2 * parts) * stride]; a3 += base[(size_t)(k + 3 * parts) * stride];
    }
    for (; k < n; k += parts) a0 += base[(size_t)k * stride];
    return (a0 + a1) + (a2 + a3);
}

template <int NT>
__device__ __forceinline__ void vu_tri_body(const VuPrepareArgs &a, const int b)
{
    extern __shared__ __attribute__((aligned(16))) double vu_lds[];
    const int tid = threadIdx.x;
    const size_t rec = (size_t)b;
    // the records the gate launch of this class will look at (its early exits restated, without their side effects)
    const int np_rec = __builtin_amdgcn_readfirstlane(a.np_rec ? a.np_rec[rec] : a.np);
    if (np_rec < 2 || np_rec > a.np) return;
    if (a.np_hi > 0 && (np_rec < a.np_lo || np_rec > a.np_hi)) return;
    if (a.success_counter && a.success_counter[b] >= a.max_successful) return;
    const int n = np_rec, ncam = a.stereo ? 2 : 1, nt = n * ncam, N = a.n;
    const int nt_max = a.np * ncam;
    const int dDim = nt * 7, ncol = dDim + 1;
    const int np_carve = a.np_hi > 0 && a.np_hi < a.np ? a.np_hi : a.np;      // the longest track of THIS launch (launch_vu_tri sizes the LDS by it)
    const TriLds L(np_carve * ncam);
    double *s_trail = vu_lds + L.trail, *s_it = vu_lds + L.it, *s_dpfi = vu_lds + L.dpfi, *s_feat = vu_lds + L.feat;
    double *s_small = vu_lds + L.small, *s_scr = vu_lds + L.scr, *s_L = vu_lds + L.lsum, *s_tot = vu_lds + L.tot;
    int *s_idx = reinterpret_cast<int *>(vu_lds + L.ints), *s_flag = s_idx + MAXNP + 3;
    const double *m = a.m + (size_t)b * N;
    double *pfi = s_small, *pfw = s_small + 3, *R0T = s_small + 18, *scal = s_small + 36, *sums13 = s_small + 52;
    auto sync = [] { lds_barrier(); };
    TRI_STAMP(0);
    if (tid < n) s_idx[tid] = a.pose_index[rec * a.np + tid];
    if (tid < nt) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            s_feat[4 * tid + k] = a.features[(rec * nt_max + tid) * 2 + k];
            s_feat[4 * tid + 2 + k] = a.velocities[(rec * nt_max + tid) * 2 + k];
        }
    }
    sync();
    TRI_STAMP(1);
    // ---- extractCameraPoseTrail (triangulation.cpp:65-103) ----
    if (tid < nt) trail_pose_record(a, m, s_idx, tid, n, s_trail + tid * POSE_WORDS);
    for (int i = tid; i < 3 * ncol; i += NT) s_dpfi[i] = 0.0;
    sync();
    TRI_STAMP(2);
    const double *p0 = s_trail;
    // ---- triangulateWithTwoCameras between pose 0 and pose ind1 (:154-173, 612-716): lane j < 15 owns derivative column j ----
    const int ind1 = a.stereo ? nt / 2 - 1 : nt - 1;
    if (tid < 15) two_camera_start(a, tid, ind1, ncol, dDim, s_trail, s_feat, s_dpfi, pfi, pfw, R0T, scal, s_flag);
    sync();
    TRI_STAMP(3);
    // ---- Gauss-Newton with derivatives (:206-343) ----
    const double *Lm = s_L;
    // column j of dpfi takes its step: (dEe_j, dM_j) = L' d_j + the motion sums `own` (dEe[3], upper dM[6]) -- or, for the time-shift
    // column, the velocity term c_t -- then d(A^-1 b) = X (dM step) - X dEe (:324-328)
    auto update_col = [&](int j, const double *own, bool is_sft, const double (&X)[9], const double (&step)[3]) {
        const double d0 = s_dpfi[j], d1 = s_dpfi[ncol + j], d2 = s_dpfi[2 * ncol + j];
        double v[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) v[e] = (Lm[e] * d0 + Lm[9 + e] * d1) + Lm[18 + e] * d2;
        double dEe[3] = {v[0], v[1], v[2]}, dM[9] = {v[3], v[4], v[5], v[4], v[6], v[7], v[5], v[7], v[8]};
        if (!is_sft) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dEe[k] += own[k];
            dM[0] += own[3]; dM[1] += own[4]; dM[2] += own[5]; dM[4] += own[6]; dM[5] += own[7]; dM[8] += own[8];
            dM[3] = dM[1]; dM[6] = dM[2]; dM[7] = dM[5];
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) dEe[k] += Lm[27 + k];
        }
        double t1[3], t2[3], t3[3];
        mv3(dM, step, t1);
        mv3(X, t1, t2);
        mv3(X, dEe, t3);
#pragma unroll
        for (int r = 0; r < 3; ++r) s_dpfi[r * ncol + j] += t2[r] - t3[r];
    };
    // lanes per sum of the reductions below
    constexpr int PARTS_L = NT >= 240 ? 8 : NT >= 120 ? 4 : 2;          // 30 sums: L[u][e] and c_t
    constexpr int PARTS_T = NT >= 252 ? 4 : NT >= 126 ? 2 : 1;          // 63 sums: the pose-0 totals
    const int NC = 4 * nt, NPp = 3 * (nt - 1), NQ = 4 * (nt - 1);       // pair tasks by type
    for (int it = 0; it < a.gn_iters; ++it) {
#define TRI_IT_STAMP(k) do { if (it < 5) TRI_STAMP(4 + 10 * it + (k)); } while (0)
        TRI_IT_STAMP(0);
        if (tid < nt) {                                       // per-pose quantities of this iteration
            const double *cur = s_trail + tid * POSE_WORDS;
            double *o = s_it + tid * TRI_ITW;
            double C[9], t[3], d[3], h[3];
            mm3(cur + 3, R0T, C);
#pragma unroll
            for (int k = 0; k < 3; ++k) d[k] = p0[k] - cur[k];
            mv3(cur + 3, d, t);
            const double pfiab[3] = {pfi[0], pfi[1], 1.0};
            mv3(C, pfiab, h);
#pragma unroll
            for (int k = 0; k < 3; ++k) h[k] += pfi[2] * t[k];
            const double ih2 = 1.0 / h[2], ih2sq = ih2 * ih2;
#pragma unroll
            for (int k = 0; k < 9; ++k) o[k] = C[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) { o[9 + k] = t[k]; o[12 + k] = h[k]; o[23 + k] = d[k]; }
            o[26] = ih2;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int c = 0; c < 2; ++c) o[15 + 3 * r + c] = -ih2 * C[3 * r + c] + h[r] * ih2sq * C[6 + c];
                o[15 + 3 * r + 2] = -t[r] * ih2 + h[r] * ih2sq * t[2];
                o[21 + r] = s_feat[4 * tid + r] - h[r] * ih2;
            }
        }
        sync();
        TRI_IT_STAMP(1);
        // ETE (9), Eerror (3), error2: entry k = sum over the poses of o[x0] o[y0] + o[x0 + dx] o[y0 + dy], four adjacent lanes each
        if (tid < 52) {
            const int k = tid >> 2, part = tid & 3;
            const int r = k / 3, c = k - 3 * r;
            const int x0 = k < 9 ? 15 + r : k < 12 ? 15 + (k - 9) : 21, y0 = k < 9 ? 15 + c : 21;
            const int dx = k < 12 ? 3 : 1, dy = k < 9 ? 3 : 1;
            double acc = 0.0, acc2 = 0.0;
            int i = part;
            for (; i + 4 < nt; i += 8) {
                const double *o = s_it + i * TRI_ITW, *o2 = o + 4 * TRI_ITW;
                acc += o[x0] * o[y0] + o[x0 + dx] * o[y0 + dy];
                acc2 += o2[x0] * o2[y0] + o2[x0 + dx] * o2[y0 + dy];
            }
            if (i < nt) { const double *o = s_it + i * TRI_ITW; acc += o[x0] * o[y0] + o[x0 + dx] * o[y0 + dy]; }
            acc += acc2;
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            if (part == 0) sums13[k] = acc;
        }
        // PLAIN part: task (pose i, unit vector u) -> the 3 + 6 numbers pair_sums gives for dh = column u of [C_i(:, 0:2) | t_i]
        {
            const int t0 = NT > 64 ? tid - 64 : tid, tstep = NT > 64 ? NT - 64 : NT;
            if (t0 >= 0)
                for (int task = t0; task < 3 * nt; task += tstep) {
                    const int i = task / 3, u = task - 3 * i;
                    const double *o = s_it + i * TRI_ITW;
                    const double dh[3] = {u < 2 ? o[u] : o[9], u < 2 ? o[3 + u] : o[10], u < 2 ? o[6 + u] : o[11]};
                    double e3[3] = {0, 0, 0}, m9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                    pair_sums<0>(o, dh, nullptr, nullptr, 0.0, 0.0, e3, m9, o + 26);
                    double *dst = s_scr + task * 9;
                    dst[0] = e3[0]; dst[1] = e3[1]; dst[2] = e3[2];
                    dst[3] = m9[0]; dst[4] = m9[1]; dst[5] = m9[2]; dst[6] = m9[4]; dst[7] = m9[5]; dst[8] = m9[8];
                }
        }
        sync();
        TRI_IT_STAMP(2);
        // L[u][e] = sum over the poses of the plain part's maps (27), c_t = sum_i E_i' vel_i (3): PARTS_L adjacent lanes each
        if (tid < 30 * PARTS_L) {
            const int sigma = tid / PARTS_L, part = tid - sigma * PARTS_L;
            double acc;
            if (sigma < 27) {
                const int u = sigma / 9, e = sigma - 9 * u;
                acc = strided_sum(s_scr + u * 9 + e, 27, nt, part, PARTS_L);
            } else {
                const int r = sigma - 27;
                acc = 0.0;
                for (int q = part; q < nt; q += PARTS_L)
                    acc += s_it[q * TRI_ITW + 15 + r] * s_feat[4 * q + 2] + s_it[q * TRI_ITW + 18 + r] * s_feat[4 * q + 3];
            }
#pragma unroll
            for (int o = 1; o < PARTS_L; o <<= 1) acc += __shfl_xor(acc, o);
            if (part == 0) s_L[sigma] = acc;
        }
        // every lane: X = (E'E)^-1 and the step, in registers (the column updates below use them)
        double X[9], step[3];
        {
            double ETE[9], Ee[3];
#pragma unroll
            for (int q = 0; q < 9; ++q) ETE[q] = sums13[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) Ee[q] = sums13[9 + q];
            inv3sym(ETE, X);
            mv3(X, Ee, step);
        }
        sync();
        TRI_IT_STAMP(3);
        // ---- the motion pairs, one task per lane, by type (header): C -> scratch [3 + q][i][9]; P -> scratch [c][i][9] and the own
        // column's update with -r; Q -> the own column's update. (the plain part's maps in the scratch are dead: L is in s_L) ----
        // Tasks are dealt in GROUPS of 64 of ONE type (C groups, then P, then Q), a group per wavefront and round: every lane of a
        // wavefront runs the same code path (dealt flat, the waves that straddled a type boundary ran two paths: +15 % VALU work)
        const int gC = (NC + 63) >> 6, gP = (NPp + 63) >> 6, gQ = (NQ + 63) >> 6;
        for (int grp = tid >> 6; grp < gC + gP + gQ; grp += NT / 64) {
            const int lane_ = tid & 63;
            const int ty = grp < gC ? 0 : grp < gC + gP ? 1 : 2;                       // (wave-uniform)
            const int e_in = (grp - (ty == 0 ? 0 : ty == 1 ? gC : gC + gP)) * 64 + lane_;
            if (e_in >= (ty == 0 ? NC : ty == 1 ? NPp : NQ)) continue;
            const int e = e_in + (ty == 0 ? 0 : ty == 1 ? NC : NC + NPp);
            if (ty == 0) {
                const int q = e / nt, i = e - q * nt;
                const double *cur = s_trail + i * POSE_WORDS, *dR0 = s_trail + 12 + 9 * q, *o = s_it + i * TRI_ITW;
                double dC[9], dt[3], dp0[3], dh[3], e3[3] = {0, 0, 0}, m9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                mmT3(cur + 3, dR0, dC);
                mTv3(dR0, s_trail + 48, dp0);
                const double dd[3] = {-dp0[0], -dp0[1], -dp0[2]};
                mv3(cur + 3, dd, dt);
                if (i == 0) {                                     // pose 0 itself: both rotations move, the positions cancel
                    double a1[9];
                    mm3(dR0, R0T, a1);
#pragma unroll
                    for (int k = 0; k < 9; ++k) dC[k] += a1[k];
                    dt[0] = 0.0; dt[1] = 0.0; dt[2] = 0.0;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) dh[r] = (dC[3 * r] * pfi[0] + dC[3 * r + 1] * pfi[1] + dC[3 * r + 2]) + pfi[2] * dt[r];
                pair_sums<1>(o, dh, dC, dt, 0.0, 0.0, e3, m9, o + 26);
                double *dst = s_scr + ((3 + q) * nt + i) * 9;
                dst[0] = e3[0]; dst[1] = e3[1]; dst[2] = e3[2];
                dst[3] = m9[0]; dst[4] = m9[1]; dst[5] = m9[2]; dst[6] = m9[4]; dst[7] = m9[5]; dst[8] = m9[8];
            } else if (ty == 1) {
                const int e2 = e - NC, c = e2 / (nt - 1), i = 1 + e2 - c * (nt - 1);
                const double *cur = s_trail + i * POSE_WORDS, *o = s_it + i * TRI_ITW;
                const double dt[3] = {cur[3 + c], cur[6 + c], cur[9 + c]};          // R_i(:, c)
                const double dh[3] = {pfi[2] * dt[0], pfi[2] * dt[1], pfi[2] * dt[2]};
                double e3[3] = {0, 0, 0}, m9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                pair_sums<2>(o, dh, nullptr, dt, 0.0, 0.0, e3, m9, o + 26);
                const double r9[9] = {e3[0], e3[1], e3[2], m9[0], m9[1], m9[2], m9[4], m9[5], m9[8]};
                double *dst = s_scr + (c * nt + i) * 9;
                double own[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) { dst[k] = r9[k]; own[k] = -r9[k]; }
                update_col(7 * i + c, own, false, X, step);
            } else {
                const int e2 = e - NC - NPp, q = e2 / (nt - 1), i = 1 + e2 - q * (nt - 1);
                const double *cur = s_trail + i * POSE_WORDS, *dR = cur + 12 + 9 * q, *o = s_it + i * TRI_ITW;
                double dC[9], dt[3], dpi[3], t1[3], t2[3], dh[3], e3[3] = {0, 0, 0}, m9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                mm3(dR, R0T, dC);
                mTv3(dR, cur + 48, dpi);
                mv3(dR, o + 23, t1);
                mv3(cur + 3, dpi, t2);
#pragma unroll
                for (int k = 0; k < 3; ++k) dt[k] = t1[k] + t2[k];
#pragma unroll
                for (int r = 0; r < 3; ++r) dh[r] = (dC[3 * r] * pfi[0] + dC[3 * r + 1] * pfi[1] + dC[3 * r + 2]) + pfi[2] * dt[r];
                pair_sums<1>(o, dh, dC, dt, 0.0, 0.0, e3, m9, o + 26);
                const double own[9] = {e3[0], e3[1], e3[2], m9[0], m9[1], m9[2], m9[4], m9[5], m9[8]};
                update_col(7 * i + 3 + q, own, false, X, step);
            }
        }
        sync();
        TRI_IT_STAMP(4);
        // totals of the pose-0 columns (component, entry) over the poses, in a fixed order (position components: poses 1 ..)
        if (tid < 63 * PARTS_T) {
            const int sigma = tid / PARTS_T, part = tid - sigma * PARTS_T;
            const int comp = sigma / 9, en = sigma - 9 * comp, first = comp < 3 ? 1 : 0;
            double acc = strided_sum(s_scr + ((size_t)comp * nt + first) * 9 + en, 9, nt - first, part, PARTS_T);
#pragma unroll
            for (int o = 1; o < PARTS_T; o <<= 1) acc += __shfl_xor(acc, o);
            if (part == 0) s_tot[sigma] = acc;
        }
        sync();
        TRI_IT_STAMP(5);
        if (tid < 8) {                                        // the 7 columns of pose 0 and the time-shift column: one code path
            const bool sft = tid == 7;
            if (!sft || a.est_shift) update_col(sft ? dDim : tid, s_tot + 9 * (sft ? 0 : tid), sft, X, step);
        } else if (tid == NT - 1) {                             // :316-342 (everybody is done with the old pfi: the barrier above)
#pragma unroll
            for (int k = 0; k < 3; ++k) pfi[k] -= step[k];
            const double J = 0.5 * sums13[12] / (a.conv_r * a.conv_r), Jd = fabs((J - scal[2]) / J);
            scal[2] = J;
            if (Jd < a.conv_threshold) s_flag[0] = 1;
        }
        sync();
        TRI_IT_STAMP(6);
#undef TRI_IT_STAMP
        if (s_flag[0]) break;
    }
    TRI_STAMP(54);
    // ---- status, back to world coordinates (:345-392) ----
    double *M = s_small + 40, *pf0 = s_small + 49;
    if (tid == 0) {
        int status = HV_TRI_OK;
        if (a.gn_iters > 0) {                                 // reciprocal condition number of the last iteration's E'E (:313-315)
            double ETE[9], Xl[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) ETE[q] = sums13[q];
            inv3sym(ETE, Xl);
            scal[1] = 1.0 / (norm1_3(ETE) * norm1_3(Xl));
        }
        if (!s_flag[0]) status = HV_TRI_NO_CONVERGENCE;
        else if (scal[1] < a.rcond_threshold) status = HV_TRI_BAD_COND;
        if (status == HV_TRI_OK) {
            double d[9], q[3], w[3], Ml[9];
            inverse_depth(pfi, q, d);
            mv3(R0T, q, w);
#pragma unroll
            for (int k = 0; k < 3; ++k) { pfw[k] = w[k] + p0[k]; pf0[k] = q[k]; }
            if (pfw[0] == p0[0] && pfw[1] == p0[1] && pfw[2] == p0[2]) status = HV_TRI_UNKNOWN_PROBLEM;
            mm3(R0T, d, Ml);
#pragma unroll
            for (int k = 0; k < 9; ++k) M[k] = Ml[k];
        }
        s_flag[1] = status;
        s_flag[2] = 0;
    }
    sync();
    TRI_STAMP(55);
    int status = s_flag[1];
    double *recp = a.tri_rec + rec * (size_t)a.tri_stride;
    const int R_DPF = 17 * nt_max, R_SFT = R_DPF + 21 * a.np;
    // ---- prepareVisualUpdate, per-pose part (triangulation.cpp:897-947): dip R (2x3), dip dRpt (2x4), f, depth class; isBehind (:54-60).
    // The last wave takes the poses, the others the columns' way back to world coordinates ----
    {
        const int t0 = NT > 64 ? tid - (NT - 64) : tid;
        if (t0 >= 0 && t0 < nt) {
            const double *pose = s_trail + t0 * POSE_WORDS;
            double *o = recp + 17 * t0;
            const double pt[3] = {pfw[0] - pose[0], pfw[1] - pose[1], pfw[2] - pose[2]};
            if (status == HV_TRI_OK && pose[9] * pt[0] + pose[10] * pt[1] + pose[11] * pt[2] < 0) atomicOr(&s_flag[2], 1);
            double pfc[3], ipH[3], dip[9];
            mv3(pose + 3, pt, pfc);
            inverse_depth(pfc, ipH, dip);
            const double cls = pfc[2] == 0 ? 1.0 : pfc[2] < 0 ? 2.0 : 0.0;
            o[16] = cls;
            s_it[t0 * TRI_ITW + 16] = cls;
            o[14] = ipH[0]; o[15] = ipH[1];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) o[3 * r + c] = dip[3 * r] * pose[3 + c] + dip[3 * r + 1] * pose[6 + c] + dip[3 * r + 2] * pose[9 + c];
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) {
                const double *dR = pose + 12 + 9 * jq;
                double a1[3], b1[3], b2[3];
                mv3(dR, pt, a1);
                mTv3(dR, pose + 48, b1);
                mv3(pose + 3, b1, b2);
#pragma unroll
                for (int r = 0; r < 2; ++r) o[6 + 4 * r + jq] = dip[3 * r] * (a1[0] + b2[0]) + dip[3 * r + 1] * (a1[1] + b2[1]) + dip[3 * r + 2] * (a1[2] + b2[2]);
            }
        }
    }
    if (status == HV_TRI_OK) {
        // backend.cpp:1108-1119: per-pose derivative blocks in world coordinates (:345-392), the two cameras of a pose summed -- straight
        // into the record: entry (k, r, c) = world(column 7 k + c)[r] + world(column 7 (k + n) + c)[r]
        auto world = [&](int j, int r) -> double {
            double u = 0.0;
            if (j >= 3 && j < 7) { const double *dR = s_trail + 12 + 9 * (j - 3); u = dR[r] * pf0[0] + dR[3 + r] * pf0[1] + dR[6 + r] * pf0[2]; }   // (dR0' pf0)[r]
            const double v = M[3 * r] * s_dpfi[j] + M[3 * r + 1] * s_dpfi[ncol + j] + M[3 * r + 2] * s_dpfi[2 * ncol + j];
            return u + v + (j == r ? 1.0 : 0.0);
        };
        const int nlan = NT > 64 ? NT - 64 : NT;
        if (tid < nlan)
            for (int i = tid; i < n * 21; i += nlan) {
                const int k = i / 21, e = i - 21 * k, r = e / 7, c = e - 7 * r;
                double v = world(7 * k + c, r);
                if (a.stereo) v += world(7 * (k + n) + c, r);
                recp[R_DPF + i] = v;
            }
        if (tid < 3) recp[R_SFT + tid] = a.est_shift ? world(dDim, tid) : 0.0;
    }
    sync();
    TRI_STAMP(56);
    if (tid == 0) {
        if (status == HV_TRI_OK && s_flag[2]) status = HV_TRI_BEHIND;
        {   // backend.cpp:1098-1102: depth window on whatever point the triangulation left behind
            const double dx = pfw[0] - p0[0], dy = pfw[1] - p0[1], dz = pfw[2] - p0[2], depth = sqrt(dx * dx + dy * dy + dz * dz);
            if (depth < a.min_dist || depth > a.max_dist) status = HV_TRI_BAD_DEPTH;
        }
        int prep = 0;
        for (int i = 0; i < nt && prep == 0; ++i) prep = (int)s_it[i * TRI_ITW + 16];   // first failing pose decides (:920-927)
        recp[R_SFT + 3] = (double)prep;
        a.status[2 * rec] = status;
#pragma unroll
        for (int k = 0; k < 3; ++k) a.pf[3 * rec + k] = pfw[k];
    }
    TRI_STAMP(57);
}

// one wavefront per track: the short class (and every class of a small launch); listed / ordered like the gate launch it feeds
__global__ __launch_bounds__(64) void vu_tri_kernel(VuPrepareArgs a)
{
    int b = blockIdx.x;
    if (a.rec_list) { if (b >= *a.rec_count) return; b = a.rec_list[b]; }
    else if (a.order) b = __builtin_amdgcn_readfirstlane(a.order[blockIdx.x]);
    vu_tri_body<64>(a, b);
}
// four wavefronts per track: the long class, whose chain sets the visit's length
__global__ __launch_bounds__(256) void vu_tri_kernel_x4(VuPrepareArgs a)
{
    int b = blockIdx.x;
    if (a.rec_list) { if (b >= *a.rec_count) return; b = a.rec_list[b]; }
    else if (a.order) b = __builtin_amdgcn_readfirstlane(a.order[blockIdx.x]);
    vu_tri_body<256>(a, b);
}
// ... and two (knob vu_tri_threads)
__global__ __launch_bounds__(128) void vu_tri_kernel_x2(VuPrepareArgs a)
{
    int b = blockIdx.x;
    if (a.rec_list) { if (b >= *a.rec_count) return; b = a.rec_list[b]; }
    else if (a.order) b = __builtin_amdgcn_readfirstlane(a.order[blockIdx.x]);
    vu_tri_body<128>(a, b);
}

__global__ __launch_bounds__(VT_LATENCY, 3) void vu_prepare_kernel(VuPrepareArgs a) { vu_prepare_body<VT_LATENCY, MAXP_ALL, 0>(a, blockIdx.x); }
// 4 waves per SIMD = 128 VGPRs: two workgroups of 6 waves may put 4 waves on one SIMD (512 VGPRs per lane there)
__global__ __launch_bounds__(VT_THROUGHPUT, 4) void vu_prepare_kernel_2percu(VuPrepareArgs a)
{
    vu_prepare_body<VT_THROUGHPUT, MAXP_SMALL, 0>(a, blockIdx.x);
}
// the same two builds with the column-sparse chi2 gate fused in (VuPrepareArgs::fused)
__global__ __launch_bounds__(VT_LATENCY, 3) void vu_prepare_map_kernel(VuPrepareArgs a) { vu_prepare_body<VT_LATENCY, MAXP_ALL, 0, true>(a, blockIdx.x); }
__global__ __launch_bounds__(VT_LATENCY, 3) void vu_gate_kernel(VuPrepareArgs a) { vu_prepare_body<VT_LATENCY, MAXP_ALL, 1>(a, blockIdx.x); }
__global__ __launch_bounds__(VT_THROUGHPUT, 4) void vu_gate_kernel_2percu(VuPrepareArgs a)
{
    // a.order (ragged frame loops): the records arrive longest track first. A workgroup's time grows ~2x from 4 to 11 poses and the
    // launch is ~1.6 waves of workgroups on the chip's 512 slots: in filter order a long track that starts late sets the launch time
    const int b = a.order ? __builtin_amdgcn_readfirstlane(a.order[blockIdx.x]) : (int)blockIdx.x;
    vu_prepare_body<VT_THROUGHPUT, MAXP_SMALL, 1>(a, b);
}

// record-fed gate builds (r06): the front has run as vu_tri_kernel; three of the short class's workgroups share a CU (48 KB each)
constexpr int VT_REC = 256;     // (five wavefronts -- one per 16-row block of P(a, a) of an 11-pose track -- measured slower on four lanes: profiles/r06/split_tri_ab_v2.txt)
__global__ __launch_bounds__(VT_REC, 3) void vu_gate_rec_kernel(VuPrepareArgs a)
{
    const int b = a.order ? __builtin_amdgcn_readfirstlane(a.order[blockIdx.x]) : (int)blockIdx.x;
    vu_prepare_body<VT_REC, MAXP_REC, 1, false, true>(a, b);
}
__global__ __launch_bounds__(VT_LATENCY, 3) void vu_gate_long_rec_kernel(VuPrepareArgs a)
{
    int b = blockIdx.x;
    if (a.rec_list) { if (b >= *a.rec_count) return; b = a.rec_list[b]; }
    vu_prepare_body<VT_LATENCY, MAXP_ALL, 3, false, true>(a, b);
}

// launch_visit_order: counting sort of one visit's filters per workgroup (keys 0 .. 63: the pose count inside the class, else 0)
// Two sorts per visit: `order` = the whole batch, the class np_lo .. np_hi first and longest first; `long_list` / `long_count` = the
// records above np_hi (<= np_max), longest first -- the long class's launches then need not wait for the fused launch to collect them.
__global__ __launch_bounds__(1024) void visit_order_kernel(const int *np_rec, int batch, int np_lo, int np_hi, int np_max, int *order,
                                                           int *long_list, int *long_count)
{
    __shared__ int cnt[2][64], start[2][64];
    const int t = threadIdx.x;
    np_rec += (size_t)blockIdx.x * batch; order += (size_t)blockIdx.x * batch;
    if (long_list) long_list += (size_t)blockIdx.x * batch;
    if (t < 128) cnt[t >> 6][t & 63] = 0;
    __syncthreads();
    auto key = [&](int i) -> int { const int np = np_rec[i]; return (np >= np_lo && np <= np_hi) ? min(np, 63) : 0; };
    auto key_long = [&](int i) -> int { const int np = np_rec[i]; return (np > np_hi && np <= np_max) ? min(np, 63) : 0; };
    for (int i = t; i < batch; i += 1024) { atomicAdd(&cnt[0][key(i)], 1); atomicAdd(&cnt[1][key_long(i)], 1); }
    __syncthreads();
    if (t < 2) {
        int s = 0;
        for (int k = 63; k >= 0; --k) { start[t][k] = s; s += cnt[t][k]; }
        if (t == 1 && long_count) long_count[blockIdx.x] = batch - cnt[1][0];
    }
    __syncthreads();
    for (int i = t; i < batch; i += 1024) {
        order[atomicAdd(&start[0][key(i)], 1)] = i;
        const int kl = key_long(i);
        if (long_list && kl > 0) long_list[atomicAdd(&start[1][kl], 1)] = i;
    }
}
// the long class of a ragged visit: compact Jacobian + the chi2 gate (on the Jacobian's factors) in one launch (FUSED = 3), listed like
// vu_compact_kernel; unlisted over a (filters, tracks) grid it serves EVERY record of a speculative pass over long tracks
__global__ __launch_bounds__(VT_LATENCY, 3) void vu_gate_long_kernel(VuPrepareArgs a)
{
    int b = blockIdx.x;
    if (a.rec_list) { if (b >= *a.rec_count) return; b = a.rec_list[b]; }
    vu_prepare_body<VT_LATENCY, MAXP_ALL, 3>(a, b);
}
// ... and with the compact Jacobian only (the gate runs as its own launch)
__global__ __launch_bounds__(VT_LATENCY, 3) void vu_compact_kernel(VuPrepareArgs a)
{
    // listed launches (the long class of a ragged visit): workgroup i serves rec_list[i]; the workgroups beyond the list leave at once.
    // ONE instantiation of the body: r03 kept a queue-persistent copy and a listed copy beside the plain one, and the kernel spilled
    // 45 VGPRs (236 B of scratch per lane) at its 168-register cap
    int b = blockIdx.x;
    if (a.rec_list) { if (b >= *a.rec_count) return; b = a.rec_list[b]; }
    vu_prepare_body<VT_LATENCY, MAXP_ALL, 2>(a, b);
}
__global__ __launch_bounds__(VT_THROUGHPUT, 4) void vu_compact_kernel_2percu(VuPrepareArgs a)
{
    vu_prepare_body<VT_THROUGHPUT, MAXP_SMALL, 2>(a, blockIdx.x);
}

}  // namespace

int launch_visit_order(Ctx *c, int visits, int batch, const int *np_rec_dev, int np_lo, int np_hi, int np_max, int *order_dev,
                       int *long_list_dev, int *long_count_dev)
{
    if (visits < 1 || batch < 1 || !np_rec_dev || !order_dev) return HV_ERR_INVALID;
    hipLaunchKernelGGL(visit_order_kernel, dim3((unsigned)visits), dim3(1024), 0, c->stream, np_rec_dev, batch, np_lo, np_hi, np_max, order_dev,
                       long_list_dev, long_count_dev);
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

// The fused gate serves tracks of up to 48 rows (the 16 TI <= 48 MFMA tile template; 12 stereo / 24 mono poses) whose staged
// matrices fit the LDS regions the Gauss-Newton arrays leave free; everything else keeps the dense path.
static bool vu_small_build(const Ctx *c, int nt, int batch)
{
    const int forced = c->knob.vu_threads;
    // two filters per CU pay off once the launch holds more filters than the GPU has CUs; below that the big build's latency wins
    return nt <= MAXP_SMALL && (forced == VT_THROUGHPUT || (forced != VT_LATENCY && batch > 256));
}

bool vu_fused_supported(const Ctx *c, int n_state, int np, int stereo, int batch)
{
    if (c->knob.ekf_fused_gate == 0) return false;
    const int nt = np * (stereo ? 2 : 1), rows = 2 * nt, na4 = (7 * np + 1 + 3) & ~3, nrp = 16 * ((rows + 15) / 16);
    if (np < 2 || np > MAXNP || nt > MAXP_ALL || rows > 48 || rows >= HV_CHI2INV95_N || n_state < 1 || n_state > 160) return false;
    int Rs = rows + 1;
    while ((Rs & 31) != 15 && (Rs & 31) != 17) Rs++;
    const bool small = vu_small_build(c, nt, batch);
    const int hs_cap = small ? VuLds<MAXP_SMALL>::HS_DOUBLES : VuLds<MAXP_ALL>::HS_DOUBLES;
    const int t_cap = small ? VuLds<MAXP_SMALL>::T_DOUBLES : VuLds<MAXP_ALL>::T_DOUBLES;
    return na4 * nrp <= hs_cap && 816 + VT_LATENCY / 64 <= hs_cap && Rs * rows <= t_cap;
}

// shapes the split form serves: iterative triangulation of pose-trail tracks, no speculation; the record-fed short-class gate holds
// stereo tracks of up to 12 poses (its staged Jacobian is 88 x 48 doubles), the long-class gate everything vu_gate_long_kernel does
// vu_split_short_ok: what a caller knows before the arguments of a launch exist (ekf.hip visit_shape: where the short class ends)
bool vu_split_short_ok(const Ctx *c, int n_state, bool stereo, int batch, bool linear)
{
    // (knob value 2: at every batch size -- tests; 1: where the two-per-CU fused build would run)
    return c->knob.ekf_split_tri != 0 && stereo && !linear && n_state <= 160 && (batch > 256 || c->knob.ekf_split_tri == 2);
}
int vu_split_short_np(const Ctx *c) { return c->knob.ekf_short_np == 11 ? 11 : MAXP_REC / 2; }

bool vu_split_supported(const Ctx *c, const VuPrepareArgs &a, int fused)
{
    if (c->knob.ekf_split_tri == 0 || a.linear || a.map_index || a.spec_tracks > 0 || !a.P || a.n > 160) return false;
    const int ncam = a.stereo ? 2 : 1, np_sel = a.np_hi > 0 && a.np_hi < a.np ? a.np_hi : a.np;
    if (a.np > MAXNP || a.np * ncam > MAXP_ALL) return false;
    if (fused == 1) {
        if (!vu_split_short_ok(c, a.n, a.stereo != 0, a.batch, a.linear != 0) || np_sel * ncam > MAXP_REC) return false;
        const int rows = 2 * np_sel * ncam, na4 = (7 * np_sel + 1 + 3) & ~3, nrp = 16 * ((rows + 15) / 16);
        int Rs = rows + 1;
        while ((Rs & 31) != 15 && (Rs & 31) != 17) Rs++;
        return rows < HV_CHI2INV95_N && na4 * nrp <= VuRecLds<MAXP_REC>::HS_DOUBLES && Rs * rows <= VuRecLds<MAXP_REC>::T_DOUBLES &&
               c->knob.ekf_fused_gate != 0 && np_sel >= 2;
    }
    return fused == 3 && c->knob.ekf_split_tri != 3;        // (3: experiments -- the short class only, the long class keeps r05's fused launch)
}

int launch_vu_tri(Ctx *c, const VuPrepareArgs &a, hipStream_t stream)
{
    if (!stream) stream = c->stream;
    const int ncam = a.stereo ? 2 : 1;
    if (a.np < 2 || a.batch < 1 || a.np > MAXNP || a.np * ncam > MAXP_ALL) return HV_ERR_INVALID;
    if (!a.tri_rec || a.tri_stride < vu_tri_rec_stride(a.np, ncam) || a.spec_tracks > 0 || a.linear || a.map_index) return HV_ERR_INVALID;
    ScopedKernelTime tm(c, HV_K_VU_TRI, stream);
    const int np_sel = a.np_hi > 0 && a.np_hi < a.np ? a.np_hi : a.np;
    const TriLds L(np_sel * ncam);
    const size_t bytes = sizeof(double) * (size_t)L.total;
    // four wavefronts per track where the launch may hold long tracks (their chain sets the length of the visit), one otherwise
    // knob vu_tri_threads (experiments / tests): 64 / 128 / 256 force a build
    const int forced = c->knob.vu_tri_threads;
    // auto: four wavefronts per track where the launch may hold long tracks (their chain sets the length of the visit), two for the short
    // class (profiles/r06/split_tri_ab_v2.txt: 4 lanes 27.90 ms per step with 128 threads, 28.12 with 64, 28.2 with 256)
    const int nthr = forced == 64 || forced == 128 || forced == 256 ? forced : (np_sel * ncam > MAXP_REC ? 256 : 128);
    if (nthr == 256)      hipLaunchKernelGGL(vu_tri_kernel_x4, dim3((unsigned)a.batch), dim3(256), bytes, stream, a);
    else if (nthr == 128) hipLaunchKernelGGL(vu_tri_kernel_x2, dim3((unsigned)a.batch), dim3(128), bytes, stream, a);
    else                  hipLaunchKernelGGL(vu_tri_kernel, dim3((unsigned)a.batch), dim3(64), bytes, stream, a);
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

int launch_vu_prepare(Ctx *c, const VuPrepareArgs &a, hipStream_t stream)
{
    if (!stream) stream = c->stream;
    if (a.np < 2 || a.batch < 1) return HV_ERR_INVALID;
    // the kernel's LDS arrays hold at most MAXNP poses per camera (the reference's cameraTrailLength 20 + the current pose);
    // a longer trail (cameraTrailLength > 20) is a supported filter size but not a supported track length here
    if (a.np > MAXNP || a.np * (a.stereo ? 2 : 1) > MAXP_ALL) return HV_ERR_UNSUPPORTED;
    ScopedKernelTime tm(c, HV_K_VU_PREPARE, stream);
    const int np_sel = a.np_hi > 0 && a.np_hi < a.np ? a.np_hi : a.np;      // the longest track this launch processes (a.np stays the record stride)
    const int nt = np_sel * (a.stereo ? 2 : 1);
    // knob vu_threads (tests / experiments): 384 / 768 forces a build where it applies
    const bool small = vu_small_build(c, nt, a.batch);
    if (a.fused == 1 && (!vu_fused_supported(c, a.n, np_sel, a.stereo, a.batch) || !a.P)) return HV_ERR_INVALID;
    if (a.fused && (!a.Hc || !a.acol || a.na_max < 7 * a.np + 1)) return HV_ERR_INVALID;
    if (a.fused == 3) {                                      // long class: 12 .. 21 stereo poses (49 .. 84 rows; 48 rows ride along), n <= 160
        const int rows = 2 * nt;
        // (speculative frame loop, r04: grid (filters, tracks), never listed -- every pending record of whatever length on this build)
        if (!a.P || (a.spec_tracks > 0 && a.rec_list) || a.n > 160 || rows > HV_GATE_TIGHT_ROWS || rows >= HV_CHI2INV95_N) return HV_ERR_UNSUPPORTED;
    }
    static bool attr_set_dev[64] = {};                       // per device: the kernels need more than the default 64 KB of dynamic LDS
    bool &attr_set = attr_set_dev[c->p.device & 63];
    if (!attr_set) {
        HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(vu_prepare_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)VuLds<MAXP_ALL>::BYTES));
        HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(vu_prepare_kernel_2percu), hipFuncAttributeMaxDynamicSharedMemorySize, (int)VuLds<MAXP_SMALL>::BYTES));
        HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(vu_gate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)VuLds<MAXP_ALL>::BYTES));
        HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(vu_gate_kernel_2percu), hipFuncAttributeMaxDynamicSharedMemorySize, (int)VuLds<MAXP_SMALL>::BYTES));
        HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(vu_compact_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)VuLds<MAXP_ALL>::BYTES));
        HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(vu_compact_kernel_2percu), hipFuncAttributeMaxDynamicSharedMemorySize, (int)VuLds<MAXP_SMALL>::BYTES));
        constexpr int long_bytes_attr = (int)VuLds<MAXP_ALL, true>::BYTES;
        HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(vu_gate_long_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, long_bytes_attr));
        attr_set = true;
    }
    const dim3 grid((unsigned)a.batch, (unsigned)(a.spec_tracks > 0 ? a.spec_tracks : 1));
    if (a.from_rec) {                                        // the gate half of the split form (the caller has launched vu_tri_kernel)
        if (!a.tri_rec || !vu_split_supported(c, a, a.fused)) return HV_ERR_INVALID;
        static bool rec_attr_dev[64] = {};
        bool &rec_attr = rec_attr_dev[c->p.device & 63];
        if (!rec_attr) {
            constexpr int rec_long_attr = (int)VuRecLds<MAXP_ALL, true>::BYTES;
            HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(vu_gate_long_rec_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, rec_long_attr));
            rec_attr = true;
        }
        constexpr size_t rec_long_bytes = VuRecLds<MAXP_ALL, true>::BYTES, rec_short_bytes = VuRecLds<MAXP_REC>::BYTES;
        if (a.fused == 3) hipLaunchKernelGGL(vu_gate_long_rec_kernel, grid, dim3(VT_LATENCY), rec_long_bytes, stream, a);
        else              hipLaunchKernelGGL(vu_gate_rec_kernel, grid, dim3(VT_REC), rec_short_bytes, stream, a);
        HV_HIP(c, hipGetLastError());
        return HV_OK;
    }
    if (a.map_index) {                                       // hybrid-map tracks: the dense-H build with the map branch
        if (a.fused || a.map_base < 0) return HV_ERR_INVALID;
        static bool map_attr_dev[64] = {};
        bool &map_attr = map_attr_dev[c->p.device & 63];
        if (!map_attr) {
            HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(vu_prepare_map_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)VuLds<MAXP_ALL>::BYTES));
            map_attr = true;
        }
        hipLaunchKernelGGL(vu_prepare_map_kernel, grid, dim3(VT_LATENCY), VuLds<MAXP_ALL>::BYTES, stream, a);
    } else
    if (a.fused == 3) {
        constexpr size_t long_bytes = VuLds<MAXP_ALL, true>::BYTES;
        hipLaunchKernelGGL(vu_gate_long_kernel, grid, dim3(VT_LATENCY), long_bytes, stream, a);
    } else if (a.fused == 2) {
        if (small) hipLaunchKernelGGL(vu_compact_kernel_2percu, grid, dim3(VT_THROUGHPUT), VuLds<MAXP_SMALL>::BYTES, stream, a);
        else       hipLaunchKernelGGL(vu_compact_kernel, grid, dim3(VT_LATENCY), VuLds<MAXP_ALL>::BYTES, stream, a);
    } else if (a.fused) {
        if (small) hipLaunchKernelGGL(vu_gate_kernel_2percu, grid, dim3(VT_THROUGHPUT), VuLds<MAXP_SMALL>::BYTES, stream, a);
        else       hipLaunchKernelGGL(vu_gate_kernel, grid, dim3(VT_LATENCY), VuLds<MAXP_ALL>::BYTES, stream, a);
    } else {
        if (small) hipLaunchKernelGGL(vu_prepare_kernel_2percu, grid, dim3(VT_THROUGHPUT), VuLds<MAXP_SMALL>::BYTES, stream, a);
        else       hipLaunchKernelGGL(vu_prepare_kernel, grid, dim3(VT_LATENCY), VuLds<MAXP_ALL>::BYTES, stream, a);
    }
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

}  // namespace hv

extern "C" int hv_debug_tri_phase_stamps(hv_ctx *ctx, long long *out64 /* [64] */)
{
    hv::Ctx *c = hv::ctx_of(ctx);
    if (!c || !out64) return HV_ERR_INVALID;
    HV_HIP(c, hipStreamSynchronize(c->stream));
    HV_HIP(c, hipMemcpyFromSymbol(out64, HIP_SYMBOL(hv::g_tri_stamp), sizeof(long long) * 64));
    return HV_OK;
}

extern "C" int hv_debug_vu_phase_stamps(hv_ctx *ctx, long long *out32 /* [40] */)
{
    hv::Ctx *c = hv::ctx_of(ctx);
    if (!c || !out32) return HV_ERR_INVALID;
    HV_HIP(c, hipStreamSynchronize(c->stream));
    HV_HIP(c, hipMemcpyFromSymbol(out32, HIP_SYMBOL(hv::g_vu_stamp), sizeof(long long) * 40));
    return HV_OK;
}
